// What does page-locking cost against a pageable transfer?  (the CRAM reader's run buffers)  hipcc --offload-arch=gfx950 -O2 pinned_probe.hip -o pinned_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)1 << 30;
    void *d; hipMalloc(&d, N); hipMemset(d, 1, N); hipDeviceSynchronize();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); char *p = (char *)malloc(N); double t1 = now();
        hipMemcpyAsync(p, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t2 = now();
        hipMemcpyAsync(p, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t3 = now();
        hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t4 = now();
        printf("pageable 1 GiB: malloc %.1f ms, D2H first touch %.1f ms, D2H again %.1f ms, H2D %.1f ms\n", (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
        t = now(); hipHostRegister(p, N, hipHostRegisterDefault); t1 = now();
        hipMemcpyAsync(p, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t2 = now();
        hipHostUnregister(p); t3 = now();
        printf("hipHostRegister of the touched buffer %.1f ms, D2H %.1f ms, unregister %.1f ms\n", (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        free(p);
        void *h; t = now(); hipHostMalloc(&h, N, hipHostMallocDefault); t1 = now();
        hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t2 = now();
        hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t3 = now();
        hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t4 = now();
        double t5; hipHostFree(h); t5 = now();
        printf("pinned 1 GiB: hipHostMalloc %.1f ms, D2H %.1f ms, again %.1f ms, H2D %.1f ms, free %.1f ms\n", (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
        // consumer-side read of a pinned buffer
        hipHostMalloc(&h, N, hipHostMallocDefault); memset(h, 1, N); char *q = (char *)malloc(N);
        t = now(); memcpy(q, h, N); t1 = now(); memcpy(q, h, N); t2 = now();
        printf("host memcpy out of pinned memory 1 GiB: %.1f ms, again %.1f ms\n", (t1 - t) * 1e3, (t2 - t1) * 1e3);
        free(q); hipHostFree(h);
    }
    return 0;
}

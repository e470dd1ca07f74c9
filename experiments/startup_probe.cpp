// Probe (round 6): what a fresh process pays before its first device batch -- runtime load, hipInit, context, streams, pinned + device buffers,
// first launch of a kernel from libhtsgpu.so (code-object load).   startup_probe <path to libhtsgpu.so>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef int (*fn_i)(unsigned);
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    double t0 = now(), t;
    void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    t = now(); printf("dlopen libamdhip64.so      %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    if (!hip) { printf("no hip: %s\n", dlerror()); return 1; }
    auto hipInit = (int (*)(unsigned))dlsym(hip, "hipInit");
    auto hipSetDevice = (int (*)(int))dlsym(hip, "hipSetDevice");
    auto hipFree = (int (*)(void *))dlsym(hip, "hipFree");
    auto hipMalloc = (int (*)(void **, size_t))dlsym(hip, "hipMalloc");
    auto hipHostMalloc = (int (*)(void **, size_t, unsigned))dlsym(hip, "hipHostMalloc");
    auto hipStreamCreateWithFlags = (int (*)(void **, unsigned))dlsym(hip, "hipStreamCreateWithFlags");
    auto hipDeviceSynchronize = (int (*)())dlsym(hip, "hipDeviceSynchronize");
    hipInit(0);
    t = now(); printf("hipInit                    %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipSetDevice(0); hipFree(nullptr);
    t = now(); printf("hipSetDevice + hipFree(0)  %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    void *s[4]; for (int i = 0; i < 4; i++) hipStreamCreateWithFlags(&s[i], 1);
    t = now(); printf("4 streams                  %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    void *h; hipHostMalloc(&h, 64u << 20, 0);
    t = now(); printf("hipHostMalloc 64 MiB       %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    void *d; hipMalloc(&d, 256u << 20);
    t = now(); printf("hipMalloc 256 MiB          %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    if (argc > 1) {
        void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
        t = now(); printf("dlopen libhtsgpu.so        %7.1f ms\n", (t - t0) * 1e3); t0 = t;
        if (!lib) { printf("%s\n", dlerror()); return 1; }
        auto hg_init = (int (*)(int, void **))dlsym(lib, "hg_init");
        void *ctx; int rc = hg_init(0, &ctx);
        t = now(); printf("hg_init (rc %d)             %7.1f ms\n", rc, (t - t0) * 1e3); t0 = t;
        auto hg_crc = (int (*)(void *, const void *, size_t, uint32_t *))dlsym(lib, "hg_crc32_host");
        if (hg_crc) { std::vector<char> b(1 << 20, 'x'); uint32_t c = 0; rc = hg_crc(ctx, b.data(), b.size(), &c);
            t = now(); printf("first kernel (crc32, rc %d) %7.1f ms\n", rc, (t - t0) * 1e3); t0 = t; }
    }
    hipDeviceSynchronize();
    return 0;
}

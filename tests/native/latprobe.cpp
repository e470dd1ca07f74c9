// latprobe: latency of ONE small batch of BGZF blocks through (a) the device entry point (kernel only, HIP events and
// wall clock) and (b) the host pipe (H2D + kernel + D2H + wait), for the first n = 1, 4, 16, 64 blocks of a file.
//   latprobe <file.bgzf> [lib.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "htsgpu.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    size_t len = 8u << 20; std::vector<unsigned char> buf(len + 8); len = fread(buf.data(), 1, len, f); fclose(f);
    { size_t pos = 0; while (pos + 18 <= len) { size_t bs = (size_t)(buf[pos + 16] | (buf[pos + 17] << 8)) + 1; if (pos + bs > len) break; pos += bs; } len = pos; }   // whole blocks only
    void *h = dlopen(argc > 2 ? argv[2] : "htslib_amd/libhtsgpu.so", RTLD_NOW | RTLD_LOCAL); if (!h) { printf("%s\n", dlerror()); return 1; }
#define SYM(name) auto p_##name = (decltype(&name))dlsym(h, #name)
    SYM(hg_init); SYM(hg_destroy); SYM(hg_bgzf_scan); SYM(hg_bgzf_inflate_dev); SYM(hg_pipe_create); SYM(hg_pipe_destroy);
    SYM(hg_pipe_input); SYM(hg_pipe_inflate); SYM(hg_pipe_wait);
    hg_ctx *ctx; if (p_hg_init(0, &ctx)) return 1;
    uint64_t total = 0; long n = p_hg_bgzf_scan(buf.data(), len, nullptr, 0, &total);
    std::vector<hg_bgzf_desc> desc(n); p_hg_bgzf_scan(buf.data(), len, desc.data(), n, &total);
    void *dc, *dd, *dout; int32_t *dst;
    CK(hipMalloc(&dc, len + 256)); CK(hipMalloc(&dd, n * sizeof(hg_bgzf_desc))); CK(hipMalloc(&dout, total + 256)); CK(hipMalloc((void **)&dst, n * 4));
    CK(hipMemcpy(dc, buf.data(), len, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, desc.data(), n * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hg_pipe *pipe; if (p_hg_pipe_create(ctx, &pipe)) return 1;
    for (long nb : {1L, 4L, 16L, 64L}) {
        if (nb > n) break;
        const size_t clen = desc[nb - 1].coff + desc[nb - 1].clen, ulen = desc[nb - 1].uoff + desc[nb - 1].ulen;
        double k_ev = 1e30, k_wall = 1e30, p_sub = 1e30, p_all = 1e30;
        for (int r = 0; r < 12; r++) {
            CK(hipStreamSynchronize(s));
            double t0 = now();
            CK(hipEventRecord(e0, s)); p_hg_bgzf_inflate_dev(ctx, dc, clen, (hg_bgzf_desc *)dd, nb, dout, ulen, dst, s); CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1)); double t1 = now();
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { k_ev = ms * 1e3 < k_ev ? ms * 1e3 : k_ev; k_wall = t1 - t0 < k_wall ? t1 - t0 : k_wall; }
            t0 = now();
            void *in = p_hg_pipe_input(pipe, clen + 65536); memcpy(in, buf.data(), clen);
            double ta = now();
            if (p_hg_pipe_inflate(pipe, clen, desc.data(), nb)) return 2;
            double tb = now();
            const uint8_t *out; size_t ol; const int32_t *st;
            if (p_hg_pipe_wait(pipe, &out, &ol, &st, nullptr, nullptr)) return 3;
            t1 = now();
            if (r >= 2) { p_sub = tb - ta < p_sub ? tb - ta : p_sub; p_all = t1 - t0 < p_all ? t1 - t0 : p_all; }
        }
        printf("blocks %3ld  comp %7zu B  plain %8zu B : kernel %7.1f us (events) %7.1f us (wall)   pipe submit %6.1f us  input+submit+wait %7.1f us\n",
               nb, clen, ulen, k_ev, k_wall, p_sub, p_all);
    }
    p_hg_pipe_destroy(pipe); p_hg_destroy(ctx);
    return 0;
}

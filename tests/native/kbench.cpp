// kbench: time the BGZF inflate kernel of one or more builds of the library on a .bgzf file.
//   kbench <file.bgzf> <reps> <lib1.so> [lib2.so ...]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "htsgpu.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)
int main(int argc, char **argv) {
    if (argc < 4) { printf("usage\n"); return 1; }
    FILE *f = fopen(argv[1], "rb"); if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
    fseek(f, 0, SEEK_END); size_t len = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf(len + 8); if (fread(buf.data(), 1, len, f) != len) return 1; fclose(f);
    int reps = atoi(argv[2]);
    for (int a = 3; a < argc; a++) {
        void *h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL); if (!h) { printf("dlopen %s: %s\n", argv[a], dlerror()); continue; }
        auto p_init = (int (*)(int, hg_ctx **))dlsym(h, "hg_init");
        auto p_scan = (long (*)(const uint8_t *, size_t, hg_bgzf_desc *, size_t, uint64_t *))dlsym(h, "hg_bgzf_scan");
        auto p_inf = (int (*)(hg_ctx *, const void *, size_t, const hg_bgzf_desc *, size_t, void *, size_t, int32_t *, void *))dlsym(h, "hg_bgzf_inflate_dev");
        auto p_fini = (void (*)(hg_ctx *))dlsym(h, "hg_destroy");
        auto p_prof = (int (*)(unsigned long long *, int))dlsym(h, "hg_debug_get_profile");
        hg_ctx *ctx; if (p_init(0, &ctx)) { printf("init failed\n"); return 1; }
        uint64_t total = 0; long n = p_scan(buf.data(), len, nullptr, 0, &total);
        std::vector<hg_bgzf_desc> desc(n); p_scan(buf.data(), len, desc.data(), n, &total);
        void *dc, *dd, *dout; int32_t *dst;
        CK(hipMalloc(&dc, len + 256)); CK(hipMalloc(&dd, n * sizeof(hg_bgzf_desc))); CK(hipMalloc(&dout, total + 256)); CK(hipMalloc((void **)&dst, n * 4));
        CK(hipMemcpy(dc, buf.data(), len, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, desc.data(), n * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice));
        CK(hipMemset(dst, 0x7f, n * 4));
        hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        p_inf(ctx, dc, len, (hg_bgzf_desc *)dd, n, dout, total, dst, s); CK(hipStreamSynchronize(s));
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, s)); p_inf(ctx, dc, len, (hg_bgzf_desc *)dd, n, dout, total, dst, s); CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
        }
        std::vector<int32_t> st(n); CK(hipMemcpy(st.data(), dst, n * 4, hipMemcpyDeviceToHost));
        long bad = 0; for (long i = 0; i < n; i++) bad += st[i] != 0;
        printf("%-44s blocks %ld plain %.3f GB  best %.3f ms  mean %.3f ms  => %.2f GB/s (best)  bad=%ld\n", argv[a], n, total / 1e9, best, sum / reps, total / 1e6 / best, bad);
        if (p_prof) {
            unsigned long long pr[16]; p_prof(pr, 1);
            p_inf(ctx, dc, len, (hg_bgzf_desc *)dd, n, dout, total, dst, s); CK(hipStreamSynchronize(s)); p_prof(pr, 1);
            double tot = (double)pr[0];
            printf("   in-kernel wave time (s_memtime ticks, 100 MHz): blocks %llu  per-block total %.0f  header+tables %.1f%%  symbols %.1f%%  resolve %.1f%%  crc %.1f%%\n",
                   pr[5], tot / pr[5], 100 * pr[1] / tot, 100 * pr[2] / tot, 100 * pr[3] / tot, 100 * pr[4] / tot);
        }
        fflush(stdout);
        // ---- deflate the plain image that the inflate above produced --------------------------
        auto p_def = (int (*)(hg_ctx *, const void *, const hg_bgzf_desc *, size_t, int, void *, uint32_t *, void *))dlsym(h, "hg_bgzf_deflate_dev");
        auto p_dprof = (int (*)(unsigned long long *, int))dlsym(h, "hg_debug_get_deflate_profile");
        if (p_def && getenv("KBENCH_DEFLATE")) {
            std::vector<hg_bgzf_desc> dd2(desc); for (long i = 0; i < n; i++) dd2[i].coff = (uint64_t)i * 65536;
            void *dslots, *ddd; uint32_t *dclen;
            CK(hipMalloc(&dslots, (size_t)n * 65536 + 256)); CK(hipMalloc(&ddd, n * sizeof(hg_bgzf_desc))); CK(hipMalloc((void **)&dclen, n * 4));
            CK(hipMemcpy(ddd, dd2.data(), n * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice));
            const char *lv = getenv("KBENCH_LEVELS"); std::vector<int> levels;
            for (const char *q = lv ? lv : "6"; *q; q++) if (*q >= '0' && *q <= '9') levels.push_back(*q - '0');
            float ms = 0;
            for (int level : levels) {
            p_def(ctx, dout, (hg_bgzf_desc *)ddd, n, level, dslots, dclen, s); CK(hipStreamSynchronize(s));
            if (p_dprof) { unsigned long long pr[16]; p_dprof(pr, 1); }
            CK(hipEventRecord(e0, s)); p_def(ctx, dout, (hg_bgzf_desc *)ddd, n, level, dslots, dclen, s); CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint32_t> cl(n); CK(hipMemcpy(cl.data(), dclen, n * 4, hipMemcpyDeviceToHost));
            double csum = 0; for (long i = 0; i < n; i++) csum += cl[i];
            printf("   DEFLATE level %d: %.3f ms => %.2f GB/s, ratio %.3f (input stream ratio %.3f, size vs input stream %.4f)\n", level, ms, total / 1e6 / ms, total / csum, (double)total / len, csum / len);
            }
            if (p_dprof) { unsigned long long pr[16]; p_dprof(pr, 1); double tot = (double)pr[0];
                printf("   deflate in-kernel time per block %.0f ticks: stage+crc %.1f%%  match+parse %.1f%%  huffman %.1f%%  emit %.1f%%\n",
                       tot / pr[5], 100 * pr[1] / tot, 100 * pr[2] / tot, 100 * pr[3] / tot, 100 * pr[4] / tot);
                printf("      match detail (wave 0's clock): tier1 %.1f%%  wait %.1f%%  walk1 %.1f%%  tier-2 list %.1f%%  pairs %.1f%%  extend %.1f%%  merge %.1f%%  walk2+publish %.1f%%  emit %.1f%%\n", 100 * pr[6] / tot, 100 * pr[7] / tot, 100 * pr[8] / tot, 100 * pr[9] / tot,
                       100 * pr[10] / tot, 100 * pr[11] / tot, 100 * pr[12] / tot, 100 * pr[13] / tot, 100 * pr[14] / tot); }
            CK(hipFree(dslots)); CK(hipFree(ddd)); CK(hipFree(dclen));
        }
        fflush(stdout);
        CK(hipFree(dc)); CK(hipFree(dd)); CK(hipFree(dout)); CK(hipFree(dst)); p_fini(ctx); 
    }
    return 0;
}

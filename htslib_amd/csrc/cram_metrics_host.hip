// cram_metrics_host.hip -- batch form of cram_compress_block / cram_compress_block2 WITH the per-data-series
// method auto-tuner (reference cram/cram_io.c:1912-2325: cram_compress_block3, cram_new_metrics :2327-2339,
// TRIAL_SPAN 70 / NTRIALS 3 :118-119, meth_cost :2116-2154, methmap :1928-1943; struct cram_metrics
// cram/cram_structs.h:284-305).  Host logic only: every byte of compression work goes to the gfx950 encoders
// through the hg_*_encode_host entry points, one batched call per codec family and round.
//
// Batching: blocks that share a metrics object see exactly the state sequence of the reference's block-at-a-time
// loop.  A call is split into rounds at the points where a trial phase finishes (the blocks after it need the method
// it learns); everything decidable goes to the GPU together, across all data series -- usually two rounds per call.
#include <hip/hip_runtime.h>
#include <atomic>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <dlfcn.h>
#include <mutex>
#include <vector>
#include "htsgpu.h"
#include "hg_internal.h"

namespace {

constexpr int TRIAL_SPAN = 70, NTRIALS = 3, MAXM = HG_CRAM_MAX_METHOD;

// internal method id -> on-disk method id (methmap, cram_io.c:1928-1943)
const int8_t methmap[MAXM] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 0, 1, 1, 7, 7, 7, 4, 5, 5, 5, 5, 5, 5, 5, 8, 6, 6, 6, 6, 6, 6, 6};
// relative cost of the methods (meth_cost, cram_io.c:2116-2154)
const double meth_cost[MAXM] = {1, 1.04, 1.07, 1.08, 1.00, 1.00, 1.04, 1.05, 1.05, 1.00, 1.00, 1.01, 1.01, 1.05, 1.05, 1.05,
                                1.01, 1.01, 1.00, 1.03, 1.00, 1.01, 1.00, 1.01, 1.07, 1.04, 1.04, 1.04, 1.03, 1.04, 1.04, 1.04};
// the flag byte of the parameterised rANS Nx16 / arith methods ({1,64,9,128,129,192,193}, cram_io.c:1856,1877)
int pr_flags(int m) {
    static const int f[8] = {0, 1, 64, 9, 128, 129, 192, 193};
    if (m == HG_M_RANS_PR0 || m == HG_M_ARITH_PR0) return 0;
    if (m >= HG_M_RANS_PR1 && m <= HG_M_RANS_PR193) return f[m - HG_M_RANS_PR1 + 1];
    if (m >= HG_M_ARITH_PR1 && m <= HG_M_ARITH_PR193) return f[m - HG_M_ARITH_PR1 + 1];
    return -1;
}

struct Job { size_t blk; int m; };

// Test hook (tests/test_cram_metrics_reference.py): when set, no codec runs -- every (block, method) job "compresses" to the
// size the script names (0 = the method fails), so that the auto-tuner's decisions can be compared, on the CPU, with the
// reference's cram_compress_block3 driven by the same script.
uint32_t (*g_size_script)(int method, size_t blk, uint32_t in_len) = nullptr;
// counters of the auto-tuner since the process started (hg_debug_cram_tuner_counters): calls, rounds, trial blocks compressed ahead of time, ... of those
// folded from the cache, trial blocks that took the normal path
std::atomic<uint64_t> g_tuner[5];                                // (calls on different contexts run concurrently: sharded writers, several devices)

struct HostLibs {
    int (*bz2)(char *, unsigned int *, char *, unsigned int, int, int, int) = nullptr;                                                      // BZ2_bzBuffToBuffCompress
    int (*lzma_enc)(uint32_t, int, const void *, const uint8_t *, size_t, uint8_t *, size_t *, size_t) = nullptr;                           // lzma_easy_buffer_encode
    size_t (*lzma_bound)(size_t) = nullptr;                                                                                                // lzma_stream_buffer_bound
};
const HostLibs &host_libs() {
    static HostLibs H;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *n : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) { H.bz2 = (decltype(H.bz2))dlsym(h, "BZ2_bzBuffToBuffCompress"); if (H.bz2) break; }
        for (const char *n : {"liblzma.so.5", "liblzma.so"}) if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
            H.lzma_enc = (decltype(H.lzma_enc))dlsym(h, "lzma_easy_buffer_encode"); H.lzma_bound = (decltype(H.lzma_bound))dlsym(h, "lzma_stream_buffer_bound");
            if (H.lzma_enc && H.lzma_bound) break;
        }
    });
    return H;
}

// Runs every (block, method) job.  Jobs are grouped by CODEC FAMILY, not by method id: the entry points take a
// parameter per stream (order / flag byte / back-end), so e.g. all seven RANS_PR* trials of all blocks are ONE batched
// GPU call.  res[j] = the payload (or null when the method is not available / failed), rlen[j] = its size.  The payloads of a
// codec family live in one buffer of its sibling context (hg::host_slab: a malloc per job of the codecs' worst-case bounds was
// thousands of mmap / munmap pairs per round, and fresh pages fault); those of the host libraries are malloc'd and listed in
// `arenas`, which the caller frees when it has picked the winners.  Valid until the next run_jobs on this context.
int run_jobs(hg_ctx *ctx, const std::vector<Job> &jobs, int level, const uint8_t *const *in, const uint32_t *in_len,
             const hg_fqz_slice *const *fqz, std::vector<uint8_t *> &res, std::vector<uint32_t> &rlen, std::vector<uint8_t *> &arenas) {
    res.assign(jobs.size(), nullptr); rlen.assign(jobs.size(), 0);
    if (g_size_script) {
        for (size_t j = 0; j < jobs.size(); j++) {
            const uint32_t sz = g_size_script(jobs[j].m, jobs[j].blk, in_len[jobs[j].blk]);
            if (sz) { res[j] = (uint8_t *)calloc(sz, 1); rlen[j] = sz; arenas.push_back(res[j]); }
        }
        return HG_OK;
    }
    // bzip2 / lzma (cram_io.c:1781-1832): the system's libraries, looked up at first use like the read side does (cram_block_front.cpp) -- the
    // reference calls the very same functions when built with HAVE_LIBBZ2 / HAVE_LIBLZMA; absent libraries = the method fails, as there.
    for (size_t j = 0; j < jobs.size(); j++) {
        const int m = jobs[j].m;
        if (m != HG_M_BZIP2 && m != HG_M_LZMA) continue;
        const HostLibs &H = host_libs();
        const size_t b = jobs[j].blk;
        if (m == HG_M_BZIP2 && H.bz2) {
            unsigned int cap = (unsigned int)(in_len[b] * 1.01 + 600);
            char *p = (char *)malloc(cap);
            if (p && H.bz2(p, &cap, (char *)in[b], in_len[b], level > 0 ? level : 5, 0, 30) == 0) { res[j] = (uint8_t *)p; rlen[j] = cap; arenas.push_back(res[j]); } else free(p);
        } else if (m == HG_M_LZMA && H.lzma_enc && H.lzma_bound) {
            const size_t cap = H.lzma_bound(in_len[b]);
            uint8_t *p = (uint8_t *)malloc(cap ? cap : 1); size_t pos = 0;
            if (p && H.lzma_enc((uint32_t)(level > 0 ? level : 5), 1 /* LZMA_CHECK_CRC32 */, nullptr, in[b], in_len[b], p, &pos, cap) == 0) { res[j] = p; rlen[j] = (uint32_t)pos; arenas.push_back(p); } else free(p);
        }
    }
    enum Fam { F_GZ = 0, F_GZ1, F_R4, F_NX, F_AR, F_TK, F_FQ, F_N };
    auto family = [](int m) -> int {
        if (m == HG_M_GZIP) return F_GZ;
        if (m == HG_M_GZIP_RLE || m == HG_M_GZIP_1) return F_GZ1;
        if (m == HG_M_RANS0 || m == HG_M_RANS1) return F_R4;
        if (m == HG_M_RANS_PR0 || (m >= HG_M_RANS_PR1 && m <= HG_M_RANS_PR193)) return F_NX;
        if (m == HG_M_ARITH_PR0 || (m >= HG_M_ARITH_PR1 && m <= HG_M_ARITH_PR193)) return F_AR;
        if (m == HG_M_TOK3 || m == HG_M_TOKA) return F_TK;
        if (m == HG_M_FQZ || (m >= HG_M_FQZ_b && m <= HG_M_FQZ_d)) return F_FQ;
        return -1;                                                     // bzip2 / lzma: not in the engine -> "failed"
    };
    // The families are independent: each runs on its own thread, sibling context (own scratch, own HIP stream), so
    // their -- individually latency-bound -- kernels overlap on the GPU.
    int frc[F_N];
    static const char *const fam_name[F_N] = {"gzip", "gzip-1", "rans4x8", "ransNx16", "arith", "tok3", "fqzcomp"};
    const bool stats = getenv("HTS_GPU_STATS") != nullptr;
    double fam_ms[F_N] = {0}; size_t fam_jobs[F_N] = {0}; uint64_t fam_bytes[F_N] = {0};
    std::vector<std::thread> th;
    for (int fam = 0; fam < F_N; fam++) {
        frc[fam] = HG_OK;
        std::vector<size_t> idx;
        for (size_t j = 0; j < jobs.size(); j++) if (family(jobs[j].m) == fam) idx.push_back(j);
        if (idx.empty()) continue;
        if (!ctx->sub[fam] && hg_init(ctx->device, &ctx->sub[fam]) != HG_OK) { frc[fam] = HG_ENOMEM; continue; }
        hg_ctx *sub = ctx->sub[fam];
        th.emplace_back([&, fam, sub, idx]() {
            if (hipSetDevice(ctx->device) != hipSuccess) { frc[fam] = HG_ENODEV; return; }
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<const uint8_t *> sin; std::vector<uint8_t *> sout; std::vector<uint32_t> slen, solen(idx.size(), 0);
            std::vector<uint8_t> par(idx.size());
            std::vector<const hg_fqz_slice *> fsl; std::vector<int32_t> fstrat;
            std::vector<size_t> caps(idx.size() + 1, 0);
            for (size_t k = 0; k < idx.size(); k++) {
                const size_t b = jobs[idx[k]].blk; const int m = jobs[idx[k]].m;
                sin.push_back(in[b]); slen.push_back(in_len[b]);
                const size_t cap = fam <= F_GZ1 ? hg_gzip_compress_bound(in_len[b]) : fam == F_R4 ? hg_rans4x8_compress_bound(in_len[b])
                                 : fam == F_NX ? hg_ransnx16_compress_bound(in_len[b]) : fam == F_AR ? hg_arith_compress_bound(in_len[b])
                                 : fam == F_FQ ? hg_fqz_compress_bound(in_len[b], fqz && fqz[b] ? fqz[b]->num_records : 0)
                                 : hg_tok3_compress_bound(in_len[b]);
                if (fam == F_FQ) {                                     // strat = 0..3 for FQZ, FQZ_b, FQZ_c, FQZ_d (cram_io.c:2065-2068)
                    fsl.push_back(fqz ? fqz[b] : nullptr); fstrat.push_back(m == HG_M_FQZ ? 0 : m - HG_M_FQZ_b + 1);
                }
                caps[k + 1] = caps[k] + ((cap + 63) & ~(size_t)63);
                // RANS_ORDER_SIMD_AUTO (cram_io.c:1860): the 32-way layout for inputs big enough to fill it
                par[k] = fam == F_R4 ? (uint8_t)(m == HG_M_RANS1) : fam == F_NX ? (uint8_t)(pr_flags(m) | (in_len[b] >= 65536u ? 4 : 0))
                       : fam == F_AR ? (uint8_t)pr_flags(m) : (uint8_t)(m == HG_M_TOKA);
            }
            // the family's sibling context keeps the buffer between rounds and calls (untouched pages cost nothing: the bounds are worst cases; the
            // caller of this round holds the parent's lock, so nobody else is in here)
            uint8_t *arena = hg::host_slab(sub, 2, caps[idx.size()] + 64);
            if (!arena) { frc[fam] = HG_ENOMEM; return; }
            for (size_t k = 0; k < idx.size(); k++) sout.push_back(arena + caps[k]);
            int rc;
            // libdeflate has no Z_RLE strategy: GZIP_RLE is run as level 1, like GZIP_1 (cram_io.c:2057-2062)
            if (fam <= F_GZ1) rc = hg_gzip_deflate_host(sub, sin.data(), slen.data(), sin.size(), fam == F_GZ ? level : 1, sout.data(), solen.data());
            else if (fam == F_R4) rc = hg_rans4x8_encode_host(sub, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
            else if (fam == F_NX) rc = hg_ransnx16_encode_host(sub, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
            else if (fam == F_AR) rc = hg_arith_encode_host(sub, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
            else if (fam == F_FQ) rc = hg_fqz_encode_host(sub, sin.data(), slen.data(), fsl.data(), fstrat.data(), sin.size(), sout.data(), solen.data());
            else rc = hg_tok3_encode_host(sub, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
            if (rc != HG_OK) { frc[fam] = rc; return; }
            for (size_t k = 0; k < idx.size(); k++)                    // distinct jobs: no two threads touch the same slot
                if (solen[k]) { res[idx[k]] = sout[k]; rlen[idx[k]] = solen[k]; }
            fam_ms[fam] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            fam_jobs[fam] = idx.size(); for (uint32_t l : slen) fam_bytes[fam] += l;
        });
    }
    for (auto &t : th) t.join();
    if (stats) {
        fprintf(stderr, "[hts-gpu] cram auto-tuner round: %zu jobs:", jobs.size());
        for (int fam = 0; fam < F_N; fam++) if (fam_jobs[fam]) fprintf(stderr, " %s %zu jobs %.1f MB %.1f ms;", fam_name[fam], fam_jobs[fam], fam_bytes[fam] / 1e6, fam_ms[fam]);
        fprintf(stderr, "\n");
    }
    for (int fam = 0; fam < F_N; fam++) if (frc[fam] != HG_OK) return frc[fam];
    return HG_OK;
}

}  // namespace

extern "C" {

void hg_debug_set_cram_size_script(uint32_t (*fn)(int, size_t, uint32_t)) { g_size_script = fn; }
void hg_debug_cram_tuner_counters(uint64_t *out) { for (int i = 0; i < 5; i++) out[i] = g_tuner[i].load(std::memory_order_relaxed); }

hg_cram_metrics *hg_cram_metrics_new(void) {                            // cram_new_metrics, cram_io.c:2327-2339
    hg_cram_metrics *m = (hg_cram_metrics *)calloc(1, sizeof *m);
    if (!m) return nullptr;
    m->trial = NTRIALS - 1;
    m->next_trial = TRIAL_SPAN / 2;                                     // learn quicker at start
    m->method = HG_M_RAW;
    return m;
}
void hg_cram_metrics_free(hg_cram_metrics *m) { free(m); }

int hg_cram_compress_blocks_metrics_host(hg_ctx *ctx, size_t n, hg_cram_metrics *const *metrics, const uint32_t *method_set, int level,
                                         int version_major, const uint8_t *const *in, const uint32_t *in_len, uint8_t *const *out,
                                         uint32_t *out_len, int32_t *method_used) {
    return hg_cram_compress_blocks_metrics_fqz_host(ctx, n, metrics, method_set, level, version_major, in, in_len, nullptr, out, out_len, method_used);
}

int hg_cram_compress_blocks_metrics_fqz_host(hg_ctx *ctx, size_t n, hg_cram_metrics *const *metrics, const uint32_t *method_set, int level,
                                             int version_major, const uint8_t *const *in, const uint32_t *in_len, const hg_fqz_slice *const *fqz,
                                             uint8_t *const *out, uint32_t *out_len, int32_t *method_used) {
    if (!ctx || (n && (!method_set || !in || !in_len || !out || !out_len || !method_used))) return HG_EINVAL;
    std::unique_lock<std::recursive_mutex> whole_call;                     // the sibling contexts' result buffers are this call's until it returns
    if (!g_size_script) whole_call = std::unique_lock<std::recursive_mutex>(*ctx->mu);   // (the scripted CPU test has no context)
    struct Blk { bool trial, done, retry; uint32_t method, orig; size_t j0, j1; };
    std::vector<Blk> B(n);
    for (size_t i = 0; i < n; i++) {
        B[i] = {false, false, false, method_set[i], method_set[i], 0, 0};
        out_len[i] = in_len[i]; method_used[i] = HG_CRAM_RAW;               // RAW until something smaller turns up (copied at the end)
        if (method_set[i] == HG_M_RAW || level == 0 || in_len[i] == 0) B[i].done = true;   // cram_io.c:1967-1972
    }
    // ---- trial phases ahead of time.  A series' blocks after a trial phase need the method it learns, and the phase after that needs the span the
    // resolution sets: the reference's state sequence serialises a long call into one round per trial phase (a call of 1 250 slices: ~17 rounds, each as
    // long as its slowest codec chain, the GPU nearly idle).  But WHICH blocks the later phases trial is predictable -- the counters move with the blocks'
    // sizes, and the span only depends on whether the best method stays the same -- and WHAT a trial block needs is every method of a set that only ever
    // shrinks.  So the first round also compresses the predicted trial blocks of all later phases with the current set ("stable" prediction: every phase
    // confirms the method of the one before).  Their results (all sizes, the smallest payload) are kept; when the state machine, run exactly as before,
    // reaches such a block and finds everything it asks for, it folds the block at once and goes on -- through the resolution and into the blocks behind the
    // phase, in the same round.  A wrong prediction costs the wasted trials and nothing else: blocks the cache cannot answer take the normal path.
    struct Spec { uint32_t ran_mask = 0; uint32_t sz[MAXM]; int best_m = -1; uint8_t *best = nullptr; uint32_t best_len = 0; };
    std::vector<Spec *> spec(n, nullptr);
    struct SpecJobs { size_t blk, j0, j1; };
    bool speculate = metrics && n > (size_t)NTRIALS && !(getenv("HG_CRAM_SPECULATE") && atoi(getenv("HG_CRAM_SPECULATE")) == 0);
    struct SpecFree { std::vector<Spec *> &v; ~SpecFree() { for (Spec *p : v) if (p) { free(p->best); delete p; } } } spec_free{spec};
    const uint32_t fqz_bits_all = (1u << HG_M_FQZ) | (1u << HG_M_FQZ_b) | (1u << HG_M_FQZ_c) | (1u << HG_M_FQZ_d);
    uint32_t nolib_all = (1u << HG_M_BZIP2) | (1u << HG_M_LZMA);          // (the scripted test runs against a reference build without the libraries)
    if (!g_size_script) { const HostLibs &HL = host_libs(); nolib_all = (HL.bz2 ? 0u : 1u << HG_M_BZIP2) | (HL.lzma_enc && HL.lzma_bound ? 0u : 1u << HG_M_LZMA); }
    auto have_mask = [&](size_t i) { return ~(nolib_all | (fqz && fqz[i] ? 0u : fqz_bits_all) | (1u << 9) | (1u << 10)); };
    // picks the winner of a trial block and folds its sizes into the series' statistics (cram_io.c:2064-2244); result(m, p, len) = method m's output
    auto fold_trial = [&](size_t i, hg_cram_metrics *M, uint32_t method_set, auto &&result) {
        uint32_t sz[MAXM];
        for (int m = 0; m < MAXM; m++) sz[m] = UINT_MAX;                // arbitrarily worse than raw
        uint32_t sz_best = in_len[i]; int method_best = 0; const uint8_t *pbest = nullptr;
        for (int m = 0; m < MAXM; m++) {
            if (!(method_set & (1u << m))) continue;
            const uint8_t *p = nullptr; uint32_t len = 0;
            if (!result(m, p, len)) continue;
            sz[m] = len;
            if (sz_best > len) { sz_best = len; method_best = m; pbest = p; }
        }
        if (pbest) { memcpy(out[i], pbest, sz_best); out_len[i] = sz_best; method_used[i] = methmap[method_best]; }
        for (int m = 0; m < MAXM; m++) M->sz[m] = (int)((unsigned)M->sz[m] + sz[m] + 2000u);   // int arithmetic wraps as in the reference
        if (--M->trial == 0) {
            uint32_t method = method_set;
            int best_method = HG_M_RAW, best_sz = INT_MAX;
            const double div = level <= 1 ? 0.25 : level <= 3 ? 1 : level <= 6 ? 2 : level <= 7 ? 3 : 0;
            if (div > 0) for (int m = 0; m < MAXM; m++) M->sz[m] = (int)(M->sz[m] * (1 + (meth_cost[m] - 1) / div));
            M->sz[9] = M->sz[10] = INT_MAX;
            for (int m = 0; m < MAXM; m++) {
                if (!M->sz[m] || !(method & (1u << m))) continue;
                if (best_sz > M->sz[m]) { best_sz = M->sz[m]; best_method = m; }
            }
            if (best_method != M->method) M->consistency = 0;
            else { const double f = 1 + M->consistency / 4.0; M->next_trial = (int)(M->next_trial * (f < 2 ? f : 2)); M->consistency++; }
            M->method = best_method;
            // zlib strategy / fqzcomp preset / tokeniser back-end of the learnt method (cram_io.c:2193-2205): Z_FILTERED = 1, Z_RLE = 3
            M->strat = best_method == HG_M_GZIP ? 1 : best_method == HG_M_GZIP_RLE ? 3 : best_method == HG_M_TOKA ? 1 : 0;
            const double MAXDELTA = 0.20; const int MAXFAILS = 4, mul = 1 + (level >= 7);
            for (int m = 0; m < MAXM; m++) {
                if (best_method == m) { M->cnt[m] = 0; M->extra[m] = 0; }
                else if (best_sz < M->sz[m]) {
                    const double r = (double)M->sz[m] / best_sz - 1;
                    if (++M->cnt[m] >= MAXFAILS * mul && (M->extra[m] += r) >= MAXDELTA * mul) method &= ~(1u << m);
                    if ((m == HG_M_FQZ || (m >= 13 && m <= 15)) && M->sz[m] > best_sz) method &= ~(1u << m);
                }
            }
            M->revised_method = method;
        }
    };
    // Blocks that share a metrics object are handled in their order, exactly as the reference's one-at-a-time loop
    // would: a ROUND takes, per metrics object, every block whose branch is already decided by the current state --
    // cached-method blocks, then the blocks of the next trial phase -- and stops there, because the blocks after a
    // trial phase need the method that phase is about to learn.  Usually two rounds per call.
    g_tuner[0].fetch_add(1, std::memory_order_relaxed);
    for (;;) {
        std::vector<Job> jobs;
        std::vector<size_t> taken;
        std::vector<hg_cram_metrics *> seen; std::vector<int> pend; std::vector<char> blocked;
        std::vector<size_t> last_blk;                                    // the series' last block this round's walk has dealt with
        for (size_t i = 0; i < n; i++) {
            Blk &b = B[i];
            if (b.done) continue;
            hg_cram_metrics *M = metrics ? metrics[i] : nullptr;
            b.j0 = b.j1 = jobs.size(); b.trial = false;
            if (!M) { jobs.push_back({i, HG_M_GZIP}); b.j1 = jobs.size(); taken.push_back(i); continue; }   // cram_io.c:2282-2299
            size_t k = 0;
            while (k < seen.size() && seen[k] != M) k++;
            if (k == seen.size()) { seen.push_back(M); pend.push_back(0); blocked.push_back(0); last_blk.push_back(i); }
            if (blocked[k]) continue;
            last_blk[k] = i;
            const int sz = (int)in_len[i];
            auto size_check_and_avg = [&]() {
                // sudden changes in size trigger a retrial (cram_io.c:1988-1997)
                if (M->input_avg_sz && (sz / 4 - 750 > M->input_avg_sz || sz < M->input_avg_sz / 4 - 750) &&
                    abs(sz - M->input_avg_sz) / 10 > M->input_avg_delta)
                    M->next_trial = 0;
            };
            auto avg_update = [&]() {
                M->input_avg_delta = (int)(0.9 * (M->input_avg_delta + abs(sz - M->input_avg_sz)));
                M->input_avg_sz += (int)(sz * .2);
                M->input_avg_sz = (int)(M->input_avg_sz * 0.8);
            };
            bool trial;
            if (b.retry) {
                // second entry of a block whose cached method failed in the previous round (cram_io.c:2252-2268): the
                // metrics were re-armed there; cram_compress_block3 runs again from the top
                b.retry = false;
                size_check_and_avg();
                trial = M->trial - pend[k] > 0 || --M->next_trial <= 0;
                avg_update();
            } else {
                size_check_and_avg();
                trial = M->trial - pend[k] > 0 || --M->next_trial <= 0;
                avg_update();
                if (!trial && M->method == HG_M_RAW) {
                    // The learnt method is RAW: cram_compress_by_method(RAW) returns NULL (cram_io.c:1896-1903), which the
                    // reference takes as "cached method failed" -- it re-arms the trial counters and restores the caller's
                    // method set (cram_io.c:2252-2262) -- and then calls itself with method = RAW, which stores the block raw
                    // (cram_io.c:1967-1972).  So: this block stays RAW, the NEXT block of the series starts a trial phase.
                    M->trial = NTRIALS; M->next_trial = TRIAL_SPAN; M->revised_method = (int)b.orig;
                    pend[k] = 0;
                    taken.push_back(i);
                    continue;
                }
            }
            taken.push_back(i);
            if (!trial) { jobs.push_back({i, M->method}); b.j1 = jobs.size(); continue; }
            b.trial = true;
            // bzip2 / lzma stay in the set when the system has the libraries (else like an htslib built without them: they leave it); fqzcomp
            // needs the slice's record lengths (cram_compress_by_method gets them from its cram_slice, cram_io.c:1808-1820)
            const uint32_t have = have_mask(i);
            uint32_t method = b.method & have;
            if (M->revised_method) method = M->revised_method & have; else M->revised_method = method;
            if (M->next_trial <= 0) {
                M->next_trial = TRIAL_SPAN; M->trial = NTRIALS;
                for (int m = 0; m < MAXM; m++) M->sz[m] /= 2;
                M->unpackable = 0;
            }
            if (M->unpackable && version_major > 3) {                    // no point bit-packing 17+ symbols (cram_io.c:2026-2047)
                auto sw = [&](int from, uint32_t to) { if (method & (1u << from)) method = (method | to) & ~(1u << from); };
                sw(HG_M_RANS_PR128, 1u << HG_M_RANS_PR0); sw(HG_M_RANS_PR129, 1u << HG_M_RANS_PR1); sw(HG_M_RANS_PR192, 1u << HG_M_RANS_PR64);
                sw(HG_M_RANS_PR193, (1u << HG_M_RANS_PR64) | (1u << HG_M_RANS_PR1));
                sw(HG_M_ARITH_PR128, 1u << HG_M_ARITH_PR0); sw(HG_M_ARITH_PR129, 1u << HG_M_ARITH_PR1); sw(HG_M_ARITH_PR192, 1u << HG_M_ARITH_PR64);
                sw(HG_M_ARITH_PR193, (1u << HG_M_ARITH_PR64) | (1u << HG_M_ARITH_PR1));
            }
            if ((method & (1u << HG_M_GZIP_RLE)) && (method & (1u << HG_M_GZIP_1))) method &= ~(1u << HG_M_GZIP_RLE);   // cram_io.c:2057-2062
            b.method = method;
            if (spec[i] && pend[k] == 0 && (method & ~spec[i]->ran_mask) == 0) {
                // every method of the set was run ahead of time: if the winner among THESE methods is the payload that was kept, the block is done now
                const Spec &S = *spec[i];
                uint32_t bsz = in_len[i]; int bm = -1;
                for (int m = 0; m < MAXM; m++) if ((method & (1u << m)) && S.sz[m] && S.sz[m] < bsz) { bsz = S.sz[m]; bm = m; }
                if (bm < 0 || bm == S.best_m) {
                    fold_trial(i, M, method, [&](int m, const uint8_t *&p, uint32_t &len) { if (!S.sz[m]) return false; len = S.sz[m]; p = m == S.best_m ? S.best : nullptr; return true; });
                    b.done = true; b.trial = false;
                    taken.pop_back();
                    g_tuner[3].fetch_add(1, std::memory_order_relaxed);
                    continue;
                }
            }
            g_tuner[4].fetch_add(1, std::memory_order_relaxed);
            for (int m = 0; m < MAXM; m++) if (method & (1u << m)) jobs.push_back({i, m});
            b.j1 = jobs.size();
            if (M->trial - ++pend[k] <= 0) blocked[k] = 1;              // this block ends the trial phase: later blocks wait a round
        }
        // ---- the predicted trial blocks of the later phases join the first round
        std::vector<SpecJobs> spec_jobs;
        if (speculate && !taken.empty()) {
            // (every round: a series whose prediction failed -- the best method changed, a size jump re-armed the trial -- is predicted again from its new state;
            // blocks that already have their results are skipped)
            for (size_t k = 0; k < seen.size(); k++) {
                hg_cram_metrics S = *seen[k];                            // a copy: the walk below changes nothing
                // the trial blocks this round collected will have been folded; a finished phase resolves -- prediction: it confirms the method before it
                auto resolve = [&]() {
                    if (S.method == HG_M_RAW) { S.consistency = 0; S.method = HG_M_GZIP; }
                    else { const double f = 1 + S.consistency / 4.0; S.next_trial = (int)(S.next_trial * (f < 2 ? f : 2)); S.consistency++; }
                };
                if (pend[k]) { S.trial -= pend[k]; if (S.trial <= 0) { S.trial = 0; resolve(); } }
                for (size_t i = last_blk[k] + 1; i < n; i++) {
                    if (metrics[i] != seen[k] || B[i].done) continue;
                    const int sz = (int)in_len[i];
                    if (S.input_avg_sz && (sz / 4 - 750 > S.input_avg_sz || sz < S.input_avg_sz / 4 - 750) && abs(sz - S.input_avg_sz) / 10 > S.input_avg_delta) S.next_trial = 0;
                    const bool trial = S.trial > 0 || --S.next_trial <= 0;
                    S.input_avg_delta = (int)(0.9 * (S.input_avg_delta + abs(sz - S.input_avg_sz)));
                    S.input_avg_sz += (int)(sz * .2);
                    S.input_avg_sz = (int)(S.input_avg_sz * 0.8);
                    if (!trial) continue;
                    if (S.next_trial <= 0) { S.next_trial = TRIAL_SPAN; S.trial = NTRIALS; }
                    uint32_t method = (seen[k]->revised_method ? (uint32_t)seen[k]->revised_method : B[i].orig) & have_mask(i);
                    if ((method & (1u << HG_M_GZIP_RLE)) && (method & (1u << HG_M_GZIP_1))) method &= ~(1u << HG_M_GZIP_RLE);
                    if (method && !(spec[i] && (method & ~spec[i]->ran_mask) == 0)) {
                        if (spec[i]) { free(spec[i]->best); delete spec[i]; spec[i] = nullptr; }
                        SpecJobs sj{i, jobs.size(), 0};
                        for (int m = 0; m < MAXM; m++) if (method & (1u << m)) jobs.push_back({i, m});
                        sj.j1 = jobs.size();
                        spec_jobs.push_back(sj); g_tuner[2].fetch_add(1, std::memory_order_relaxed);
                    }
                    if (--S.trial == 0) resolve();
                }
            }
        }
        if (taken.empty()) break;
        g_tuner[1].fetch_add(1, std::memory_order_relaxed);
        // ---- compress -------------------------------------------------------------------------------------------
        std::vector<uint8_t *> res, arenas; std::vector<uint32_t> rlen;
        const auto t_round = std::chrono::steady_clock::now();
        int rc = run_jobs(ctx, jobs, level, in, in_len, fqz, res, rlen, arenas);
        const auto t_jobs = std::chrono::steady_clock::now();
        if (rc != HG_OK) { for (auto p : arenas) free(p); return rc; }
        // ---- select, then fold the statistics in block order (cram_io.c:2064-2244) ------------------------------
        for (size_t i : taken) {
            Blk &b = B[i];
            b.done = true;
            hg_cram_metrics *M = metrics ? metrics[i] : nullptr;
            if (!b.trial) {                                              // cached / default method: kept only if it shrinks the block
                if (b.j0 == b.j1) continue;
                const size_t j = b.j0;
                if (res[j] && rlen[j] < in_len[i]) { memcpy(out[i], res[j], rlen[j]); out_len[i] = rlen[j]; method_used[i] = methmap[jobs[j].m]; }
                else if (!res[j] && M) {
                    // the cached method failed on this block: re-arm the trial and run the block again next round
                    // (cram_io.c:2252-2268).  Later blocks of the same series that this round already handled with the
                    // cached method keep their result -- the reference, working block by block, would have trialled them.
                    M->trial = NTRIALS; M->next_trial = TRIAL_SPAN; M->revised_method = (int)b.orig;
                    b.done = false; b.retry = true;
                }
                continue;
            }
            fold_trial(i, M, b.method, [&](int m, const uint8_t *&p, uint32_t &len) {
                for (size_t j = b.j0; j < b.j1; j++) if (jobs[j].m == m) { if (!res[j]) return false; p = res[j]; len = rlen[j]; return true; }
                return false;
            });
        }
        // ---- keep what was compressed ahead of time: every size, the smallest payload
        for (const SpecJobs &sj : spec_jobs) {
            Spec *S = new Spec; memset(S->sz, 0, sizeof S->sz);
            uint32_t bsz = in_len[sj.blk]; size_t jb = (size_t)-1;
            for (size_t j = sj.j0; j < sj.j1; j++) {
                S->ran_mask |= 1u << jobs[j].m;
                if (!res[j]) continue;
                S->sz[jobs[j].m] = rlen[j];
                if (rlen[j] < bsz) { bsz = rlen[j]; jb = j; }
            }
            if (jb != (size_t)-1) { S->best = (uint8_t *)malloc(bsz); if (S->best) { memcpy(S->best, res[jb], bsz); S->best_m = jobs[jb].m; S->best_len = bsz; } else S->ran_mask = 0; }
            spec[sj.blk] = S;
        }
        for (auto p : arenas) free(p);
        if (getenv("HTS_GPU_STATS") && !g_size_script)
            fprintf(stderr, "[hts-gpu] cram auto-tuner round: codecs %.1f ms, picking winners + statistics %.1f ms\n", std::chrono::duration<double, std::milli>(t_jobs - t_round).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_jobs).count());
    }
    for (size_t i = 0; i < n; i++) if (method_used[i] == HG_CRAM_RAW && in_len[i]) memcpy(out[i], in[i], in_len[i]);
    return HG_OK;
}

}  // extern "C"

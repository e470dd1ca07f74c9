// ransnx16_enc.hip -- CRAM 3.1 "rANS Nx16" block ENCODER for MI355X (gfx950 / CDNA4).
//
// Replaces rans_compress_4x16() as called by cram_compress_by_method (reference
// cram/cram_io.c:1853-1866; htscodecs is an absent submodule -- format per
// oracle/ransnx16_oracle.c, PARITY UNPINNED).  The kernel reproduces the oracle's encoder byte for
// byte (same normalisation, same table layout), which is what the tests check.
//
// Mapping: like the decoder, the N (4 / 32) interleaved states of a stream live in N adjacent lanes (sixteen 4-way streams or ONE
// 32-way stream per wavefront).  rANS encodes backwards: each step every lane pushes one symbol into its state and the lanes whose
// state would overflow first spill their low 16 bits; the spilled words are laid out from the END of a per-stream buffer towards the
// front, the highest lane first, so the split of the shared word stream is again one ballot + popcount per step (of the lanes ABOVE me).
// A stream is one serial chain and a launch lasts as long as its longest stream, so the work is organised around the latency of a step
// and of the table build:
//   * x / f through a 4096-entry LDS table of reciprocals (exact for x < 2^31; no integer divide on gfx950);
//   * source bytes in 16-byte loads, one chunk AHEAD of the steps that use them; renormalisation words staged in an LDS ring and written
//     out 2 048 at a time (a load issued behind a scattered store waits for it: one counter orders both on gfx9);
//   * histograms from 16-byte reads into replicated LDS counter matrices (order 0, and order 1 for alphabets of <= 64 symbols), global
//     atomics into a 256 x 256 scratch matrix otherwise;
//   * order-1 tables one context row at a time by the whole lane group: coalesced load, oracle-identical normalisation, serialisation
//     from bit masks (what every alphabet entry emits and where), cumulative scan.
// The host sorts the streams of a launch by flags and length (lane groups of a wavefront run in lock step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hge {

// LDS traffic of one wavefront is processed in issue order: ordering LDS accesses between its lanes needs no wait, only the
// compiler kept from moving them (a full fence would also wait for every global store in flight, ~thousands of cycles)
#define LDS_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// -DHG_ENC_PROFILE: the first group of the first workgroup prints its phase times (s_memtime ticks, 100 MHz) per stream
#ifdef HG_ENC_PROFILE
#define HE_T(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tacc[slot] += n_ - tlast; tlast = n_; } while (0)
#else
#define HE_T(slot) do { } while (0)
#endif

constexpr uint32_t RANS_L = 1u << 15;
constexpr int waves_of(int) { return 2; }     // wavefronts per workgroup (LDS budget of the 32-way variant)
enum { F_ORDER = 1, F_X32 = 4, F_NOSZ = 16, F_CAT = 32 };

struct GroupLds { uint32_t H[256]; uint16_t C[258]; uint16_t pad[2]; };

// big-endian 7-bit groups, continuation bit on all but the last (no local array: that would live in scratch memory)
__device__ __forceinline__ int put_u7(uint8_t *cp, uint32_t v) {
    const int n = v < (1u << 7) ? 1 : v < (1u << 14) ? 2 : v < (1u << 21) ? 3 : v < (1u << 28) ? 4 : 5;
    for (int k = 0; k < n; k++) cp[k] = (uint8_t)(((v >> (7 * (n - 1 - k))) & 0x7fu) | (k + 1 < n ? 0x80u : 0u));
    return n;
}
__device__ __forceinline__ uint32_t round2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

// oracle-identical normalisation of 256 counts (in F, u32) to sum `tot`
template <typename Arr>
__device__ void normalise(Arr &F, uint32_t size, uint32_t tot) {
    unsigned long long sum = 0; int M = -1; uint32_t m = 0;
    if (!size) return;
    for (int j = 0; j < 256; j++) {
        const uint32_t c = F[j];
        if (!c) continue;
        if (c > m) { m = c; M = j; }
        unsigned long long f = ((unsigned long long)c * tot) / size;
        if (!f) f = 1;
        F[j] = (uint32_t)f; sum += f;
    }
    if (sum < tot) F[M] += (uint32_t)(tot - sum);
    while (sum > tot) {
        int b = -1;
        for (int j = 0; j < 256; j++) { const uint32_t fj = F[j]; if (fj > 1 && (b < 0 || fj > F[b])) b = j; }
        uint32_t take = (uint32_t)(sum - tot);
        if (take > F[b] - 1) take = F[b] - 1;
        F[b] -= take; sum -= take;
    }
}

template <typename Arr>
__device__ uint8_t *put_alphabet(uint8_t *cp, const Arr &F) {
    int rle = 0;
    for (int j = 0; j < 256; j++) {
        if (!F[j]) continue;
        if (rle) { rle--; continue; }
        *cp++ = (uint8_t)j;
        if (j && F[j - 1]) {
            for (rle = j + 1; rle < 256 && F[rle]; rle++) {}
            rle -= j + 1;
            *cp++ = (uint8_t)rle;
        }
    }
    *cp++ = 0;
    return cp;
}

// per-stream global scratch layout for order 1 (32-bit words):
//   [0, 65536)        F[ctx][sym] counts, later normalised+shifted frequencies
//   [65536, 65792)    T[ctx] row totals
//   [65792, 66048)    A[sym] alphabet flags
//   [66048, 66048+65792/2...)  C[ctx][sym] cumulative (u16 pairs packed as u32: 256*257 entries)
constexpr uint32_t O1_F = 0, O1_C = 66048;
constexpr uint32_t O1_WORDS = 66048 + (256 * 258) / 2 + 16;
constexpr uint32_t DENSE_MAX = 64;

__device__ __forceinline__ uint4 ld16(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }   // any alignment

// fn(c, prev) for every byte c = src[i] with prev = src[i - 1] (0 for i = 0), in no particular order: each lane takes 16
// contiguous bytes per step (one 16-byte load), the tail goes byte by byte.
template <int N, typename F>
__device__ __forceinline__ void for_each_pair(const uint8_t *src, uint32_t n, int sub, F fn) {
    const uint32_t blk = 16u * N;
    uint32_t i0 = 0;
    for (; i0 + blk <= n; i0 += blk) {
        const uint32_t p = i0 + 16u * (uint32_t)sub;
        const uint4 v = ld16(src + p);
        uint32_t prev = p ? src[p - 1] : 0u;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) { const uint32_t c = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu; fn(c, prev); prev = c; }
    }
    for (uint32_t i = i0 + (uint32_t)sub; i < n; i += N) fn((uint32_t)src[i], i ? (uint32_t)src[i - 1] : 0u);
}

// TABLES_ONLY (4-way streams, round 5): the kernel stops after the frequency tables are built and serialised, and leaves what the coder needs in the stream's
// scratch (the sizes of header and table, the flag byte as it goes out, the order-0 cumulative table; the order-1 tables are in scratch anyway):
// rans4_scalar_encode_kernel below does the coding, one stream per wavefront on the scalar ALU.
constexpr uint32_t O0_META = 0, O0_CUM = 16, O0_WORDS = 16 + 130 + 14;      // order-0 scratch: 3 meta words, 129 words of u16 pairs
// NS: the number of rANS states when it differs from the width N of the lane group that builds the tables (4-way streams, tables only: one stream per wavefront,
// 32 lanes on its histogram and rows instead of 4)
template <int N, bool TABLES_ONLY = false, int NS = N>
__global__ __launch_bounds__(waves_of(N) * 64)
void ransnx16_encode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc,
                            const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel, uint32_t nsel,
                            uint8_t *out, uint32_t *out_len, uint8_t *wbuf, uint32_t *scratch) {
    // 32-way: one stream per wavefront, lanes 32..63 idle (as in the decoder: streams that share a wavefront run in lock step)
    constexpr int GROUPS = N == 32 ? 1 : 64 / N, WAVES = waves_of(N);
    // renormalisation words are collected in an LDS ring and written out a ring at a time: a scattered global store per
    // step would sit in front of every later source load (one counter orders loads and stores on gfx9)
    constexpr uint32_t STAGE = N == 32 ? 2048u : 128u;           // 16-bit words per group
    __shared__ uint16_t wstage[WAVES * GROUPS][STAGE + 2];       // + a spare word for the lanes that emit nothing
    __shared__ __attribute__((aligned(4))) uint8_t alist[WAVES * GROUPS][260];
    __shared__ uint32_t rowbuf[WAVES * GROUPS][256];             // the context row being normalised / serialised (order 1)
    __shared__ GroupLds lds[WAVES * GROUPS];
    // 32-way streams (the big data series): order-1 counts, then (start << 16 | freq), for alphabets of <= 64 symbols
    __shared__ uint32_t dpool[N == 32 ? WAVES * GROUPS : 1][N == 32 ? DENSE_MAX * DENSE_MAX : 1];
    __shared__ uint8_t ipool[N == 32 ? WAVES * GROUPS : 1][64];
    // x / f for the state update without an integer division (no such instruction: ~40 VALU ops on the loop-carried chain):
    // rcp[f] = ceil(2^(31 + ceil(log2 f)) / f), q = mulhi(x, rcp[f]) >> (ceil(log2 f) - 1), exact for x < 2^31, 2 <= f <= 4096
    __shared__ uint32_t rcp_tab[TABLES_ONLY ? 1 : 4097];
    if constexpr (!TABLES_ONLY) {                                      // (the table-only launches never divide: 4 097 64-bit divisions per workgroup were most of their time
        for (uint32_t f = threadIdx.x; f <= 4096u; f += WAVES * 64) {  //  on the thousands of tiny token streams of a name block)
            uint32_t sh = 0;
            while (f > (1u << sh)) sh++;
            rcp_tab[f] = f < 2u ? 0u : (uint32_t)((((unsigned long long)1 << (sh + 31u)) + f - 1u) / f);
        }
        __syncthreads();
    }
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & (N - 1);
    const bool idle = lane / N >= GROUPS;
    const int grp = idle ? 0 : lane / N;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[(tid >> 6) * GROUPS + grp];
    const unsigned long long gmask = ((N == 32 ? 0xffffffffull : 0xfull)) << (grp * N);
    const int lane0 = grp * N;

    for (uint32_t k = g_global; __any(k < nsel); k += g_total) {
        const bool have = k < nsel && !idle;
        const uint32_t sidx = have ? sel[k] : 0;
#ifdef HG_ENC_PROFILE
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
        uint32_t n = 0, flags = 0, shift = 12;
        const uint8_t *src = nullptr;
        uint8_t *o = nullptr, *wb = nullptr;
        uint32_t *sc = nullptr;
        uint32_t wcap = 0;
        if (have) {
            const hg_stream_desc d = desc[sidx];
            src = in + d.in_off; n = d.in_len; o = out + d.out_off;
            flags = flags_in[sidx] & (F_ORDER | F_X32 | F_NOSZ | F_CAT);
            sc = scratch + d.scratch_off;
            wcap = 2u * n + 256u;
            wb = wbuf + (uint64_t)d.reserved * 16ull;          // word buffer offset (16-byte units)
            if ((flags & F_ORDER) && n < 2u * NS) flags &= ~(uint32_t)F_ORDER;  // tiny inputs: order 0
        }
        const uint32_t order = flags & F_ORDER;
        // ---- header: flags, size ---------------------------------------------------------------
        uint32_t hdr = 0;
        if (have && sub == 0) {
            uint8_t *cp = o;
            *cp++ = (uint8_t)flags;
            if (!(flags & F_NOSZ)) cp += put_u7(cp, n);
            hdr = (uint32_t)(cp - o);
        }
        hdr = (uint32_t)__shfl((int)hdr, lane0, 64);
        if (have && (flags & F_CAT)) {
            for (uint32_t i = (uint32_t)sub; i < n; i += N) o[hdr + i] = src[i];
            if (sub == 0) out_len[sidx] = hdr + n;
        }
        const bool core = have && !(flags & F_CAT) && n != 0;
        if (have && !core && !(flags & F_CAT) && sub == 0) out_len[sidx] = hdr;   // empty input
        uint32_t tab = 0;                                        // bytes of table written after hdr
        bool dense = false;                                      // order-1 tables held in LDS, indexed by alphabet rank
        uint32_t nsym_d = 0;
        uint32_t *D = dpool[N == 32 ? (tid >> 6) * GROUPS + grp : 0];
        uint8_t *isym = ipool[N == 32 ? (tid >> 6) * GROUPS + grp : 0];
        if (core && order == 0) {
            // ---- order-0 histogram in LDS --------------------------------------------------------
            if (N == 32) {
                // 16 copies of the 256 counters (the order-1 pool is free here): a few symbols shared by 32 lanes serialise on one
                for (uint32_t j = (uint32_t)sub; j < 4096u; j += N) D[j] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                uint32_t *Hc = D + ((uint32_t)sub & 15u) * 256u;
                for_each_pair<N>(src, n, sub, [&](uint32_t c, uint32_t) { atomicAdd(&Hc[c], 1u); });
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                for (uint32_t j = (uint32_t)sub; j < 256u; j += N) {
                    uint32_t t = 0;
#pragma unroll
                    for (uint32_t c = 0; c < 16u; c++) t += D[c * 256u + j];
                    G.H[j] = t;
                }
            } else {
                for (int j = sub; j < 256; j += N) G.H[j] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                for_each_pair<N>(src, n, sub, [&](uint32_t c, uint32_t) { atomicAdd(&G.H[c], 1u); });
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (sub == 0) {
                uint32_t tot = round2(n);
                if (tot > 4096u) tot = 4096u;
                normalise(G.H, n, tot);
                uint8_t *cp = put_alphabet(o + hdr, G.H);
                for (int j = 0; j < 256; j++) if (G.H[j]) cp += put_u7(cp, G.H[j]);
                tab = (uint32_t)(cp - (o + hdr));
                int sh = 0;
                while ((tot << sh) < 4096u) sh++;
                uint32_t x = 0;
                for (int j = 0; j < 256; j++) { G.C[j] = (uint16_t)x; x += G.H[j] << sh; }
                G.C[256] = (uint16_t)x;
            }
        } else if (core) {
            // ---- order 1.  Pass 0: which byte values occur (value 0 always: it is the start context) ---------------
            const uint32_t per = n / NS;
            for (int j = sub; j < 256; j += N) G.H[j] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            for_each_pair<N>(src, n, sub, [&](uint32_t c, uint32_t) { G.H[c] = 1; });
            if (sub == 0) G.H[0] = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            // dense numbering of the alphabet (G.C[value] = index), only the rows / columns that exist are touched below
            uint32_t nsym = 0;
            uint8_t *AL = alist[(tid >> 6) * GROUPS + grp];       // the alphabet, ascending
            if (sub == 0) { for (int j = 0; j < 256; j++) { G.C[j] = (uint16_t)nsym; if (G.H[j]) { if (N == 32) isym[nsym & 63u] = (uint8_t)j; AL[nsym] = (uint8_t)j; nsym++; } } }
            nsym = (uint32_t)__shfl((int)nsym, lane0, 64);
            dense = N == 32 && nsym <= DENSE_MAX;
            nsym_d = nsym;
            for (uint32_t i = 0; i < 256; i++) {
                if (!G.H[i]) continue;
                for (uint32_t j = (uint32_t)sub; j < 256; j += N) sc[O1_F + i * 256u + j] = 0;
            }
            // as many copies of the nsym x nsym counter matrix as the pool holds (a power of two <= 32): lanes that count the
            // same (context, symbol) pair -- the usual case for quality values -- would serialise on one LDS word
            uint32_t copies = 1;
            if (dense) { while (copies < 32u && 2u * copies * nsym * nsym <= DENSE_MAX * DENSE_MAX) copies <<= 1; }
            if (dense) for (uint32_t i = (uint32_t)sub; i < copies * nsym * nsym; i += N) D[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            HE_T(0);
            if (dense) {
                // histogram with LDS atomics on the nsym x nsym matrix, then written out in the 256 x 256 layout
                uint32_t *Dc = D + ((uint32_t)sub & (copies - 1u)) * nsym * nsym;
                for_each_pair<N>(src, n, sub, [&](uint32_t c, uint32_t l) { atomicAdd(&Dc[(uint32_t)G.C[l] * nsym + G.C[c]], 1u); });
                if (sub >= 1 && sub < NS) atomicAdd(&Dc[(uint32_t)G.C[0] * nsym + G.C[src[(uint32_t)sub * per]]], 1u);   // states start in ctx 0
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (copies > 1u) {
                    for (uint32_t i = (uint32_t)sub; i < nsym * nsym; i += N) {
                        uint32_t t = 0;
                        for (uint32_t c = 0; c < copies; c++) t += D[c * nsym * nsym + i];
                        D[i] = t;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (uint32_t a = (uint32_t)sub; a < nsym; a += N) {
                    uint32_t tsum = 0;
                    const uint32_t ia = isym[a];
                    for (uint32_t b = 0; b < nsym; b++) { const uint32_t v = D[a * nsym + b]; sc[O1_F + ia * 256u + isym[b]] = v; tsum += v; }
                    (void)tsum;
                }
            } else {
                for_each_pair<N>(src, n, sub, [&](uint32_t c, uint32_t l) {
                    atomicAdd(&sc[O1_F + l * 256u + c], 1u);
                });
                if (sub >= 1 && sub < NS) atomicAdd(&sc[O1_F + src[(uint32_t)sub * per]], 1u);   // states start in ctx 0
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            HE_T(1);
            // ---- one context row at a time, all lanes of the group on it: load (coalesced) -> normalise -> serialise -> cumulative.
            // (One row per LANE, as before round 2, meant 256-step loops of dependent global loads and 64-bit divisions per lane,
            // and the table was serialised by one lane reading global memory word by word: ~60 of the ~85 M cycles of a 40 kB
            // stream with a full alphabet.)  The row lives in an LDS buffer; only lane 0's byte emission is serial.
            constexpr uint32_t SEG = 256u / N;                     // entries of a row per lane
            uint32_t *RB = rowbuf[(tid >> 6) * GROUPS + grp];
            uint16_t *C16w = (uint16_t *)(sc + O1_C);
            uint8_t *cp = o + hdr;                                 // output cursor, the same in every lane of the group
            const int cnt_alpha = (int)nsym;
            {
                uint32_t used = 0;
                if (sub == 0) { uint8_t *e = cp; *e++ = (uint8_t)(12u << 4); e = put_alphabet(e, G.H); used = (uint32_t)(e - cp); }
                cp += (uint32_t)__shfl((int)used, lane0, 64);
            }
            auto gsum = [&](uint32_t v) { for (int m = 1; m < N; m <<= 1) v += (uint32_t)__shfl_xor((int)v, m, 64); return v; };
            // (value, index) -> the largest value, the lowest index among equals; every lane of the group gets the result
            auto gargmax = [&](uint32_t &v, uint32_t &ix) {
                for (int m = 1; m < N; m <<= 1) {
                    const uint32_t ov = (uint32_t)__shfl_xor((int)v, m, 64), oi = (uint32_t)__shfl_xor((int)ix, m, 64);
                    if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
                }
            };
            for (uint32_t i = 0; i < 256u; i++) {
                if (!G.H[i]) continue;
                uint32_t *row = sc + O1_F + i * 256u;
                const uint32_t j0 = (uint32_t)sub * SEG;
                // load my segment, row total and the first largest count
                uint32_t lsum = 0, lmax = 0, lidx = 0xffffffffu;
                {
                    uint4 v[SEG / 4];
#pragma unroll
                    for (uint32_t q = 0; q < SEG / 4; q++) v[q] = *(const uint4 *)(row + j0 + 4u * q);        // all loads in flight together
#pragma unroll
                    for (uint32_t q = 0; q < SEG / 4; q++) {
                        const uint32_t c4[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                        for (int k = 0; k < 4; k++) { RB[j0 + 4u * q + k] = c4[k]; lsum += c4[k]; if (c4[k] > lmax) { lmax = c4[k]; lidx = j0 + 4u * q + k; } }
                    }
                }
                const uint32_t T = gsum(lsum);
                if (!T) {                                          // a symbol that is never a context
                    if (sub == 0) { cp[0] = 0; cp[1] = (uint8_t)(cnt_alpha - 1); }
                    cp += 2;
                    continue;
                }
                gargmax(lmax, lidx);
                uint32_t tot = round2(T);
                if (tot > 4096u) tot = 4096u;
                // f = max(1, c * tot / T); the quotient via double (exact here: c * tot < 2^44, the fraction is >= 1/T away from
                // the next integer) and checked against the integer remainder all the same
                uint32_t fsum = 0;
                const double scale = (double)tot / (double)T;
                for (uint32_t q = 0; q < SEG; q++) {
                    const uint32_t c = RB[j0 + q];
                    if (!c) continue;
                    const unsigned long long x = (unsigned long long)c * tot;
                    uint32_t f = (uint32_t)((double)c * scale);       // <= 4096; off by at most one, corrected below
                    long long r = (long long)x - (long long)((unsigned long long)f * T);
                    if (r < 0) { f--; r += T; }
                    if (r >= (long long)T) f++;
                    if (!f) f = 1;
                    RB[j0 + q] = f; fsum += f;
                }
                unsigned long long sum = gsum(fsum);
                LDS_ORDER();
                if (sum < tot) { if (sub == 0) RB[lidx] += (uint32_t)(tot - sum); }
                while (sum > tot) {                                // take the excess from the largest entries (> 1), first one wins
                    uint32_t bv = 0, bi = 0xffffffffu;
                    for (uint32_t q = 0; q < SEG; q++) { const uint32_t fj = RB[j0 + q]; if (fj > 1u && fj > bv) { bv = fj; bi = j0 + q; } }
                    gargmax(bv, bi);
                    if (bi == 0xffffffffu) break;                   // nothing left to take from (cannot happen: sum <= tot then)
                    uint32_t take = (uint32_t)(sum - tot);
                    if (take > bv - 1u) take = bv - 1u;
                    if (sub == 0) RB[bi] = bv - take;
                    sum -= take;
                    LDS_ORDER();
                }
                LDS_ORDER();
                {
                    // Serialise the row over the alphabet: a non-zero frequency as a 7-bit varint (1 or 2 bytes), a zero as
                    // (0, how many MORE zeros follow) once per run of zeros.  Every lane takes SEG consecutive alphabet entries;
                    // what an entry emits and where follows from three bit masks (zero / non-zero / two-byte) of the lane's
                    // entries, the zero flag of the entry before them and the position of the next non-zero entry after them.
                    const uint32_t ka = (uint32_t)sub * SEG < nsym ? (uint32_t)sub * SEG : nsym;
                    const uint32_t kb = ka + SEG < nsym ? ka + SEG : nsym;
                    unsigned long long zm = 0, nzm = 0, big = 0;
                    for (uint32_t q = 0; ka + q < kb; q++) {
                        const uint32_t v = RB[AL[ka + q]];
                        if (v) { nzm |= 1ull << q; if (v >= 128u) big |= 1ull << q; } else zm |= 1ull << q;
                    }
                    const uint32_t cnt = kb - ka;
                    const uint32_t lastz = cnt ? (uint32_t)((zm >> (cnt - 1u)) & 1ull) : 0u;
                    uint32_t prevz = (uint32_t)__shfl_up((int)lastz, 1, N);
                    if (sub == 0) prevz = 0;
                    // first non-zero entry at or after the start of each lane's range -> suffix minimum over the lanes
                    uint32_t fnz = nzm ? ka + (uint32_t)__builtin_ctzll(nzm) : nsym;
                    for (int m = 1; m < N; m <<= 1) { const uint32_t t = (uint32_t)__shfl_down((int)fnz, m, N); if (sub + m < N && t < fnz) fnz = t; }
                    uint32_t after = (uint32_t)__shfl_down((int)fnz, 1, N);        // next non-zero entry beyond my range
                    if (sub == N - 1) after = nsym;
                    const unsigned long long first = zm & ~((zm << 1) | (unsigned long long)prevz);   // zeros that open a run
                    const uint32_t mybytes = 2u * (uint32_t)__popcll(first) + (uint32_t)__popcll(nzm) + (uint32_t)__popcll(big);
                    uint32_t base = mybytes;
                    for (int m = 1; m < N; m <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)base, m, N); if (sub >= m) base += t; }
                    const uint32_t rowbytes = (uint32_t)__shfl((int)base, lane0 + N - 1, 64);
                    base -= mybytes;
                    for (unsigned long long w = nzm | first; w; w &= w - 1ull) {
                        const uint32_t q = (uint32_t)__builtin_ctzll(w);
                        const unsigned long long below = (1ull << q) - 1ull;
                        uint8_t *e = cp + base + 2u * (uint32_t)__popcll(first & below) + (uint32_t)__popcll(nzm & below) + (uint32_t)__popcll(big & below);
                        if ((nzm >> q) & 1ull) {
                            const uint32_t v = RB[AL[ka + q]];
                            if (v >= 128u) { e[0] = (uint8_t)(0x80u | (v >> 7)); e[1] = (uint8_t)(v & 0x7fu); } else e[0] = (uint8_t)v;
                        } else {
                            const unsigned long long later = q == 63u ? 0ull : nzm >> (q + 1u);
                            const uint32_t nxt = later ? ka + q + 1u + (uint32_t)__builtin_ctzll(later) : after;
                            e[0] = 0; e[1] = (uint8_t)(nxt - (ka + q) - 1u);
                        }
                    }
                    cp += rowbytes;
                }
                // cumulative frequencies scaled up to 2^12
                int sh = 0;
                while ((tot << sh) < 4096u) sh++;
                uint32_t fl[SEG], mine = 0;
#pragma unroll
                for (uint32_t q = 0; q < SEG; q++) { fl[q] = G.H[j0 + q] ? RB[j0 + q] << sh : 0u; mine += fl[q]; }
                uint32_t x = mine;                                  // inclusive scan over the lanes of the group
                for (int m = 1; m < N; m <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)x, m, N); if (sub >= m) x += t; }
                const uint32_t total = (uint32_t)__shfl((int)x, lane0 + N - 1, 64);
                x -= mine;
#pragma unroll
                for (uint32_t q = 0; q < SEG; q++) {
                    const uint32_t j = j0 + q;
                    if (G.H[j]) { row[j] = fl[q]; if (dense) D[(uint32_t)G.C[i] * nsym + G.C[j]] = (x << 16) | fl[q]; }
                    C16w[i * 258u + j] = (uint16_t)x; x += fl[q];
                }
                if (sub == N - 1) C16w[i * 258u + 256] = (uint16_t)total;
                LDS_ORDER();
            }
            tab = (uint32_t)(cp - (o + hdr));
            HE_T(3);
        }
        tab = (uint32_t)__shfl((int)tab, lane0, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HE_T(4);
        if constexpr (TABLES_ONLY) {
            if (have) {
                // meta words: where the requested order put its scratch (order 1: the spare words behind the tables)
                uint32_t *meta = sc + ((flags_in[sidx] & F_ORDER) ? O1_WORDS - 16u : O0_META);
                if (core) {
                    if (order == 0) for (uint32_t j = (uint32_t)sub; j < 129u; j += N) sc[O0_CUM + j] = (uint32_t)G.C[2u * j] | (uint32_t)G.C[2u * j + 1u] << 16;
                    if (sub == 0) { meta[0] = hdr; meta[1] = tab; meta[2] = flags; }
                } else if (sub == 0) meta[0] = 0xffffffffu;          // CAT / empty: the stream is complete already
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // ---- encode backwards ------------------------------------------------------------------
        uint32_t R = RANS_L;
        uint32_t wpos = wcap;                                     // next free byte (from the end) in wb
        uint32_t whi = wcap;                                      // words in [wpos, whi) are still in the LDS ring
        uint16_t *WS = wstage[(tid >> 6) * GROUPS + grp];
        const uint16_t *C16 = (const uint16_t *)(sc + O1_C);
        // prep: everything that depends on the symbols only (table entry, reciprocal) -- issued for a whole chunk of steps
        // ahead of the state updates; step: the loop-carried part (renormalisation decision, word placement, x -> x').
        // The table flavour (MODE 0: order 0, 1: order 1 in LDS, 2: order 1 in global memory) and "every lane takes part" (ALL)
        // are compile-time so that the chunk loops are straight-line code whose loads the compiler can batch.
        auto prep = [&](auto mode, auto all, bool mine, uint32_t sym, uint32_t ctx, uint32_t &f, uint32_t &start, uint32_t &rc) {
            constexpr int MODE = decltype(mode)::value;
            constexpr bool ALL = decltype(all)::value;
            f = 1; start = 0; rc = 0;
            if (ALL || mine) {
                if (MODE == 0) { start = G.C[sym]; f = (uint32_t)G.C[sym + 1] - start; }
                else if (MODE == 1) { const uint32_t e = D[(uint32_t)G.C[ctx] * nsym_d + G.C[sym]]; start = e >> 16; f = e & 0xffffu; }
                else { start = C16[ctx * 258u + sym]; f = sc[O1_F + ctx * 256u + sym]; }
                rc = rcp_tab[f & 0x1fffu];
            }
        };
        auto step = [&](auto all, bool mine, uint32_t f, uint32_t start, uint32_t rc) {
            constexpr bool ALL = decltype(all)::value;
            const uint32_t x_max = ((RANS_L >> shift) << 16) * f;
            const bool emit = (ALL || mine) && R >= x_max;
            const unsigned long long b = __ballot(emit) & gmask;
            const uint32_t above = (uint32_t)__popcll(b & ~((2ull << lane) - 1ull));   // emitting lanes above me
            const uint32_t tot = (uint32_t)__popcll(b);
            // branch-free: lanes that do not emit write to a spare word behind the ring
            WS[emit ? (((wpos >> 1) - (above + 1u)) & (STAGE - 1u)) : STAGE] = (uint16_t)R;   // wb and wpos are even
            R = emit ? R >> 16 : R;
            wpos -= 2u * tot;
            const uint32_t q = f < 2u ? R : __umulhi(R, rc) >> (31u - (uint32_t)__builtin_clz(f - 1u));
            const uint32_t Rn = (q << shift) + (R - q * f) + start;
            R = (ALL || mine) ? Rn : R;
        };
        using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>;
        using YES = std::true_type; using NO = std::false_type;
        // staged words [wpos, whi) -> the word buffer; called when the next `room` steps might not fit in the ring
        auto flush_words = [&](uint32_t room) {
            if (((whi - wpos) >> 1) + room * (uint32_t)N <= STAGE) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            uint16_t *w16 = (uint16_t *)wb;
            for (uint32_t w = (wpos >> 1) + (uint32_t)sub; w < (whi >> 1); w += N) w16[w] = WS[w & (STAGE - 1u)];
            whi = wpos;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        };
        auto push = [&](bool mine, uint32_t sym, uint32_t ctx) {
            uint32_t f, start, rc;
            flush_words(1);
            if (order == 0) prep(M0{}, NO{}, mine, sym, ctx, f, start, rc);
            else if (dense) prep(M1{}, NO{}, mine, sym, ctx, f, start, rc);
            else prep(M2{}, NO{}, mine, sym, ctx, f, start, rc);
            step(NO{}, mine, f, start, rc);
        };
        constexpr int CH = 8;                                     // order 0: steps whose loads are issued together
        constexpr int CH1 = 16;                                   // order 1: one 16-byte load per lane and chunk
        if (core && order == 0) {
            const uint32_t rem = n & (uint32_t)(N - 1);
            {   // tail symbols belong to states 0..rem-1
                const bool mine = (uint32_t)sub < rem;
                push(mine, mine ? src[n - rem + sub] : 0u, 0u);
            }
            uint32_t i = n & ~(uint32_t)(N - 1);
            if (i >= (uint32_t)(CH * N)) {
                uint32_t sy[CH], nx[CH];
#pragma unroll
                for (int k = 0; k < CH; k++) sy[k] = src[i - (uint32_t)((k + 1) * N) + sub];
                for (; i >= (uint32_t)(CH * N); i -= CH * N) {
                    const bool more = i >= (uint32_t)(2 * CH * N);
#pragma unroll
                    for (int k = 0; k < CH; k++) nx[k] = more ? src[i - (uint32_t)((CH + k + 1) * N) + sub] : 0u;   // next chunk, ahead of the stores
                    flush_words(CH);
                    uint32_t f[CH], st[CH], rc[CH];
#pragma unroll
                    for (int k = 0; k < CH; k++) prep(M0{}, YES{}, true, sy[k], 0u, f[k], st[k], rc[k]);
#pragma unroll
                    for (int k = 0; k < CH; k++) step(YES{}, true, f[k], st[k], rc[k]);
#pragma unroll
                    for (int k = 0; k < CH; k++) sy[k] = nx[k];
                }
            }
            for (; i > 0; i -= N) push(true, src[i - N + sub], 0u);
        } else if (core) {
            const uint32_t per = n / N;
            // last state first eats the remainder [N*per, n)
            long idx = (long)((uint32_t)(sub + 1) * per) - 2;
            uint32_t l = src[(uint32_t)(sub + 1) * per - 1];
            if (sub == N - 1) { l = src[n - 1]; idx = (long)n - 2; }
            for (long t = (long)n - 2; t > (long)(N * per) - 2; t--) {
                const bool mine = sub == N - 1;
                const uint32_t c = mine ? src[idx] : 0u;
                push(mine, l, c);
                if (mine) { l = c; idx--; }
            }
            uint32_t s = 0;
            if (s + 1 + CH1 <= per) {
                uint4 cur = ld16(src + idx - (CH1 - 1));
                auto chunks = [&](auto mode) {
                    for (; s + 1 + CH1 <= per; s += CH1) {
                        const uint4 nxt = s + 1 + 2 * CH1 <= per ? ld16(src + idx - (2 * CH1 - 1)) : cur;       // next chunk, ahead of the stores
                        flush_words(CH1);
                        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
                        uint32_t f[CH1], st[CH1], rc[CH1], cs[CH1];
#pragma unroll
                        for (int k = 0; k < CH1; k++) cs[k] = (w[(15 - k) >> 2] >> (((15 - k) & 3) * 8)) & 0xffu;   // cs[k] = src[idx - k]
#pragma unroll
                        for (int k = 0; k < CH1; k++) prep(mode, YES{}, true, k ? cs[k - 1] : l, cs[k], f[k], st[k], rc[k]);
#pragma unroll
                        for (int k = 0; k < CH1; k++) step(YES{}, true, f[k], st[k], rc[k]);
                        l = cs[CH1 - 1]; idx -= CH1; cur = nxt;
                    }
                };
                if (dense) chunks(M1{}); else chunks(M2{});
            }
            for (; s + 1 < per; s++) {
                const uint32_t c = src[idx];
                push(true, l, c);
                l = c; idx--;
            }
            push(true, l, 0u);                                    // first symbol of each state: context 0
        }
        // non-core groups must still take part in the ballots above?  No: push() is only called by
        // core groups, and __ballot is masked with gmask, so groups proceed independently.
        if (core) {
            HE_T(5);
            flush_words(STAGE);                                   // everything still in the ring
            // flush the states: state z ends up at (final ptr) + 4 z
            wpos -= 4u * N;
            uint8_t *w = wb + wpos + 4u * sub;
            w[0] = (uint8_t)R; w[1] = (uint8_t)(R >> 8); w[2] = (uint8_t)(R >> 16); w[3] = (uint8_t)(R >> 24);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const uint32_t body = wcap - wpos;
            uint8_t *dst = o + hdr + tab;
            {   // word buffer -> output slot, four bytes per lane and step once dst is dword aligned
                const uint8_t *from = wb + wpos;
                uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
                if (head > body) head = body;
                if ((uint32_t)sub < head) dst[sub] = from[sub];
                const uint32_t nw = (body - head) >> 2;
                for (uint32_t i = (uint32_t)sub; i < nw; i += N) {
                    const uint8_t *q = from + head + 4u * i;
                    *(uint32_t *)(dst + head + 4u * i) = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                }
                for (uint32_t i = head + 4u * nw + (uint32_t)sub; i < body; i += N) dst[i] = from[i];
            }
            if (sub == 0) { o[0] = (uint8_t)flags; out_len[sidx] = hdr + tab + body; }
            HE_T(6);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef HG_ENC_PROFILE
        if (have && sub == 0 && ((blockIdx.x == 0 && tid == 0) || k + 2 >= nsel))
            printf("enc<%d> n=%u order=%u dense=%d nsym=%u ticks: setup %llu hist %llu norm %llu serialise %llu cum %llu encode %llu flush+copy %llu\n", N, n,
                   order, (int)dense, nsym_d, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], tacc[6]);
#endif
    }
}


// ================================================================================ the 4-way coder on the scalar ALU (round 5)
// A 4-way stream is four interleaved rANS states: in the lane-group form above a step costs a ballot -> popcount -> LDS round trip for the word placement
// (~1 us per step of four symbols), sixteen streams per wavefront in lock step.  Here ONE stream has the wavefront: the 64 lanes look up (start, freq,
// reciprocal) for 64 symbols at a time, and the step itself -- renormalisation test, the 16-bit word, x = (x / f << 12) + x % f + start by multiply-high --
// runs unrolled with constant lane numbers on the scalar ALU, the four states in four scalar registers (independent chains: their latencies overlap).
// Words go out through one VGPR, 64 per store, towards lower addresses exactly as the lane-group form places them (highest state first within a step),
// so the stream is byte-identical.  Tables, header and table bytes come from ransnx16_encode_kernel<4, true>.
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// rcp[f] = ceil(2^(31 + ceil(log2 f)) / f) for f = 0 .. 4096, built at compile time (the scalar coder copies it to LDS: a workgroup of four tiny streams used to spend
// longer on 4 097 divisions than on its symbols)
struct RcpTable { uint32_t v[4097]; };
constexpr RcpTable make_rcp_table() {
    RcpTable t{};
    for (uint32_t f = 0; f <= 4096u; f++) {
        uint32_t sh = 0;
        while (f > (1u << sh)) sh++;
        t.v[f] = f < 2u ? 0u : (uint32_t)((((unsigned long long)1 << (sh + 31u)) + f - 1u) / f);
    }
    return t;
}
__device__ const RcpTable g_rcp = make_rcp_table();

struct ScalarOut {
    uint16_t *w16; uint32_t wpos, cnt, buf;                              // wpos: byte position in the word buffer (words lie in [wpos, cap))
    __device__ __forceinline__ void put(uint32_t w, int lane) {
        buf = hg::writelane(w & 0xffffu, 63u - cnt, buf);
        if (++cnt == 64u) { w16[(wpos >> 1) - 64u + (uint32_t)lane] = (uint16_t)buf; wpos -= 128u; cnt = 0; }
    }
    __device__ __forceinline__ void finish(int lane) {
        if ((uint32_t)lane >= 64u - cnt) w16[(wpos >> 1) - 64u + (uint32_t)lane] = (uint16_t)buf;
        wpos -= 2u * cnt; cnt = 0;
    }
};
// one symbol into state x: A = start << 16 | freq, rc = reciprocal of freq (rcp_tab)
__device__ __forceinline__ void scalar_step(uint32_t &x, uint32_t A, uint32_t rc, ScalarOut &O, int lane) {
    const uint32_t f = A & 0xffffu, start = A >> 16;
    if (x >= (f << 19)) { O.put(x, lane); x >>= 16; }                    // x_max = ((RANS_L >> 12) << 16) * f
    const uint32_t qq = __umulhi(x, rc) >> (31u - (uint32_t)__builtin_clz((f - 1u) | 1u));   // (| 1: f = 1, 2 give the same shift; f = 1 takes x below)
    const uint32_t q = f < 2u ? x : qq;
    x = (q << 12) + (x - q * f) + start;
}

__global__ __launch_bounds__(256)
void rans4_scalar_encode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel,
                                uint32_t nsel, uint8_t *out, uint32_t *out_len, uint8_t *wbuf, const uint32_t *__restrict__ scratch) {
    __shared__ uint32_t rcp_tab[4097];
    __shared__ uint16_t ctab[4][260];
    for (uint32_t f = threadIdx.x; f <= 4096u; f += 256) rcp_tab[f] = g_rcp.v[f];
    __syncthreads();
    // (the wavefront's index is uniform, but only readfirstlane tells the compiler so: without it every value below -- the stream, its length, the four states --
    // counts as divergent and the coding loop lands on the vector ALU behind exec-mask branches)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t k = blockIdx.x * 4u + (uint32_t)wv;
    if (k >= nsel) return;
    const uint32_t sidx = sel[k];
    const hg_stream_desc d = desc[sidx];
    const uint32_t *sc = scratch + d.scratch_off;
    const uint32_t *meta = sc + ((flags_in[sidx] & F_ORDER) ? O1_WORDS - 16u : O0_META);
    const uint32_t hdr = meta[0], tab = meta[1], flags = meta[2];
    if (hdr == 0xffffffffu) return;
    const uint8_t *src = in + d.in_off;
    const uint32_t n = d.in_len, order = flags & F_ORDER, wcap = 2u * n + 256u;
    uint8_t *o = out + d.out_off, *wb = wbuf + (uint64_t)d.reserved * 16ull;
    ScalarOut O{(uint16_t *)wb, wcap, 0u, 0u};
    uint32_t R0 = RANS_L, R1 = RANS_L, R2 = RANS_L, R3 = RANS_L;
    auto var_step = [&](uint32_t s, uint32_t A, uint32_t rc) {           // a state chosen at run time (ragged ends only)
        uint32_t x = s == 0u ? R0 : s == 1u ? R1 : s == 2u ? R2 : R3;
        scalar_step(x, A, rc, O, lane);
        R0 = s == 0u ? x : R0; R1 = s == 1u ? x : R1; R2 = s == 2u ? x : R2; R3 = s == 3u ? x : R3;
    };
    if (order == 0) {
        uint16_t *C = ctab[wv];
        for (uint32_t j = (uint32_t)lane; j < 129u; j += 64) { const uint32_t w = sc[O0_CUM + j]; C[2u * j] = (uint16_t)w; C[2u * j + 1u] = (uint16_t)(w >> 16); }
        LDS_ORDER();
        auto look = [&](uint32_t sym, bool valid, uint32_t &A, uint32_t &B) {
            const uint32_t st = C[sym], f = (uint32_t)C[sym + 1u] - st;
            A = valid ? st << 16 | f : 1u; B = rcp_tab[f & 0x1fffu];
        };
        // symbol i belongs to state i & 3; tiles of 64 positions from the top, the ragged top tile first
        const uint32_t top = (n - 1u) & ~63u;
        auto sym_at = [&](uint32_t b) { const uint32_t i = b + (uint32_t)lane; return i < n ? (uint32_t)src[i] : 0u; };
        uint32_t sy = sym_at(top), sy_next = top ? sym_at(top - 64u) : 0u;
        uint32_t A, B; look(sy, top + (uint32_t)lane < n, A, B);
        for (uint32_t j = n - 1u - top + 1u; j-- > 0u;) var_step(j & 3u, rl(A, j), rl(B, j));
        for (uint32_t b = top; b >= 64u;) {
            b -= 64u;
            look(sy_next, true, A, B);                                   // (this tile's table entries; the tile after it is already on its way)
            sy_next = b ? sym_at(b - 64u) : 0u;
#pragma unroll
            for (int j = 63; j >= 0; j--) {
                const uint32_t a = rl(A, (uint32_t)j), r = rl(B, (uint32_t)j);
                if ((j & 3) == 0) scalar_step(R0, a, r, O, lane); else if ((j & 3) == 1) scalar_step(R1, a, r, O, lane);
                else if ((j & 3) == 2) scalar_step(R2, a, r, O, lane); else scalar_step(R3, a, r, O, lane);
            }
        }
    } else {
        const uint16_t *C16 = (const uint16_t *)(sc + O1_C);
        const uint32_t *F = sc + O1_F;
        const uint32_t per = n / 4u;
        auto look1 = [&](uint32_t sym, uint32_t ctx, bool valid, uint32_t &A, uint32_t &B) {
            const uint32_t st = valid ? C16[ctx * 258u + sym] : 0u, f = valid ? F[ctx * 256u + sym] : 1u;
            A = st << 16 | f; B = rcp_tab[f & 0x1fffu];
        };
        // the last state first eats the remainder [4 per, n): positions n - 1 ... 4 per, each in the context of the byte before it
        for (uint32_t pos = n; pos-- > 4u * per;) {
            uint32_t A, B; look1(src[pos], pos ? src[pos - 1u] : 0u, true, A, B);
            var_step(3u, hg::uni(A), hg::uni(B));
        }
        // then step k = 0 .. per - 1: state s codes position (s + 1) per - 1 - k (context: the byte before; 0 for the first byte of its quarter), states 3, 2, 1, 0
        auto gather = [&](uint32_t k0, uint32_t &sym, uint32_t &ctx, bool &valid) {
            const uint32_t kk = k0 + ((uint32_t)lane >> 2), s = 3u - ((uint32_t)lane & 3u);
            valid = kk < per;
            const uint32_t pos = (s + 1u) * per - 1u - (valid ? kk : 0u);
            sym = valid ? (uint32_t)src[pos] : 0u; ctx = valid && kk + 1u < per ? (uint32_t)src[pos - 1u] : 0u;
        };
        uint32_t sy, cx, sy_n = 0, cx_n = 0; bool va, va_n = false;
        gather(0u, sy, cx, va);
        for (uint32_t k0 = 0; k0 < per; k0 += 16u) {
            uint32_t A, B; look1(sy, cx, va, A, B);
            if (k0 + 16u < per) gather(k0 + 16u, sy_n, cx_n, va_n);
            if (per - k0 >= 16u) {
#pragma unroll
                for (int j = 0; j < 64; j++) {
                    const uint32_t a = rl(A, (uint32_t)j), r = rl(B, (uint32_t)j);
                    if ((j & 3) == 0) scalar_step(R3, a, r, O, lane); else if ((j & 3) == 1) scalar_step(R2, a, r, O, lane);
                    else if ((j & 3) == 2) scalar_step(R1, a, r, O, lane); else scalar_step(R0, a, r, O, lane);
                }
            } else {
                const uint32_t nn = 4u * (per - k0);
                for (uint32_t j = 0; j < nn; j++) var_step(3u - (j & 3u), rl(A, j), rl(B, j));
            }
            sy = sy_n; cx = cx_n; va = va_n;
        }
    }
    O.finish(lane);
    // the four states in front of the words (state z at + 4 z), then everything behind the tables
    uint32_t wpos = O.wpos - 16u;
    {
        const uint32_t Rz = lane == 0 ? R0 : lane == 1 ? R1 : lane == 2 ? R2 : R3;
        if (lane < 4) { uint8_t *w = wb + wpos + 4u * (uint32_t)lane; w[0] = (uint8_t)Rz; w[1] = (uint8_t)(Rz >> 8); w[2] = (uint8_t)(Rz >> 16); w[3] = (uint8_t)(Rz >> 24); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const uint32_t body = wcap - wpos;
    uint8_t *dst = o + hdr + tab;
    const uint8_t *from = wb + wpos;
    for (uint32_t i = (uint32_t)lane; i < body; i += 64) dst[i] = from[i];
    if (lane == 0) { o[0] = (uint8_t)flags; out_len[sidx] = hdr + tab + body; }
}

}  // namespace hge

namespace hg {
uint32_t ransnx16_enc_scratch_words(uint32_t flags) { return (flags & 1u) ? hge::O1_WORDS : hge::O0_WORDS; }

int launch_ransnx16_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags,
                           const uint32_t *d_sel4, size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out,
                           uint32_t *d_out_len, void *d_wbuf, uint32_t *d_scratch, hipStream_t s) {
    const size_t maxw = (size_t)ctx->cus * 8;
    const bool side = n4 != 0 && n32 != 0;                  // both variants present: overlap them
    hipStream_t s2 = side ? fork_side(ctx, s) : s;
    if (n4) {
        constexpr int W4 = hge::waves_of(4);
        size_t wgs = (n4 + W4 * 16 - 1) / (W4 * 16);
        if (wgs > maxw) wgs = maxw;
        // tables by lane groups (sixteen streams per wavefront), then the coding on the scalar ALU, one stream per wavefront.  HG_NX4_SCALAR=0: the lane-group
        // coder of rounds 1-4 (A/B runs)
        static const bool scalar = !(getenv("HG_NX4_SCALAR") && atoi(getenv("HG_NX4_SCALAR")) == 0);
        if (scalar) {
            constexpr int WT = hge::waves_of(32);                              // tables: 32 lanes per stream, one stream per wavefront
            size_t wgt = (n4 + WT - 1) / WT;
            if (wgt > maxw * 4) wgt = maxw * 4;
            hipLaunchKernelGGL((hge::ransnx16_encode_kernel<32, true, 4>), dim3((unsigned)wgt), dim3(WT * 64), 0, s, (const uint8_t *)d_in, d_desc, d_flags, d_sel4, (uint32_t)n4,
                               (uint8_t *)d_out, d_out_len, (uint8_t *)d_wbuf, d_scratch);
            hipLaunchKernelGGL(hge::rans4_scalar_encode_kernel, dim3((unsigned)((n4 + 3) / 4)), dim3(256), 0, s, (const uint8_t *)d_in, d_desc, d_flags, d_sel4, (uint32_t)n4,
                               (uint8_t *)d_out, d_out_len, (uint8_t *)d_wbuf, (const uint32_t *)d_scratch);
        } else
        hipLaunchKernelGGL(hge::ransnx16_encode_kernel<4>, dim3((unsigned)wgs), dim3(W4 * 64), 0, s,
                           (const uint8_t *)d_in, d_desc, d_flags, d_sel4, (uint32_t)n4, (uint8_t *)d_out, d_out_len,
                           (uint8_t *)d_wbuf, d_scratch);
    }
    if (n32) {
        constexpr int W32 = hge::waves_of(32);
        size_t wgs = (n32 + W32 - 1) / W32;                                   // one 32-way stream per wavefront
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL(hge::ransnx16_encode_kernel<32>, dim3((unsigned)wgs), dim3(W32 * 64), 0, s2,
                           (const uint8_t *)d_in, d_desc, d_flags, d_sel32, (uint32_t)n32, (uint8_t *)d_out, d_out_len,
                           (uint8_t *)d_wbuf, d_scratch);
        if (side) join_side(ctx, s);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

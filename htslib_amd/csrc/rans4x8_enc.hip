// rans4x8_enc.hip -- CRAM 3.0 "rANS 4x8" block ENCODER for MI355X (gfx950 / CDNA4).
//
// Replaces rans_compress() as called by cram_compress_by_method (reference
// cram/cram_io.c:1834-1848; htscodecs rANS_static.c is an absent submodule -- format per the CRAM
// v3.0 specification as restated in oracle/rans4x8_oracle.c, whose DECODER is pinned on the
// reference's CRAM fixtures).  The kernel reproduces the oracle's encoder byte for byte.
//
// Same mapping as the Nx16 encoder (ransnx16_enc.hip): the 4 states of a stream in 4 adjacent lanes, coding backwards; here a state
// spills 0, 1 or 2 BYTES per symbol, so the split of the shared byte stream is two ballots (">= 1 byte", "2 bytes") + popcounts of the
// lanes above.  Frequencies are normalised to 4095 (stock decoders require a total < 4096).  A stream is one chain of n / 4 steps and a
// launch lasts as long as its longest chain (round 2 rework, as for the Nx16 encoder): four streams per wavefront; x / f through an LDS
// table of reciprocals; source bytes fetched a chunk ahead (16-byte loads per lane in order 1); the spilled bytes staged in an LDS ring
// and written out 16 bytes per lane; order-1 alphabets of <= 48 symbols keep their whole table -- counts, then (start << 16 | freq) --
// in LDS and lane 0 of the quad normalises and serialises it there (larger alphabets keep the round-1 path through global scratch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgq {
using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>;

constexpr uint32_t TF_SHIFT = 12, RANS_L = 1u << 23;
constexpr int WAVES = 2, N = 4, GROUPS = 4;      // 4 streams per wavefront (lanes 16..63 idle), 8 per workgroup
constexpr uint32_t DMAX = 48;                    // order-1 alphabets up to this size live in LDS
constexpr uint32_t STAGE = 512;                  // bytes of the output ring per stream

struct GroupLds { uint32_t H[256]; uint16_t C[258]; uint16_t pad[2]; };

// oracle-identical normalisation (sum 4095): cnt[] -> F[] (in place)
template <typename Arr>
__device__ void normalise(Arr &F, uint32_t total) {
    const uint32_t target = 4095u;
    uint32_t fsum = 0, M = 0, m = 0;
    for (int j = 0; j < 256; j++) {
        const uint32_t c = F[j];
        if (!c) continue;
        unsigned long long f = ((unsigned long long)c * target) / total;
        if (f == 0) f = 1;
        F[j] = (uint32_t)f; fsum += (uint32_t)f;
        if (c > m) { m = c; M = j; }
    }
    if (fsum < target) F[M] += target - fsum;
    else if (fsum > target) {
        uint32_t over = fsum - target;
        while (over) {
            uint32_t best = 0;
            for (int j = 1; j < 256; j++) if (F[j] > F[best]) best = j;
            const uint32_t fb = F[best];
            const uint32_t take = fb - 1 < over ? fb - 1 : over;
            F[best] = fb - take; over -= take;
            if (!take) break;
        }
    }
}

template <typename Arr>
__device__ uint8_t *write_table0(uint8_t *cp, const Arr &F) {
    int rle = 0;
    for (int j = 0; j < 256; j++) {
        const uint32_t f = F[j];
        if (!f) continue;
        if (rle) rle--;
        else {
            *cp++ = (uint8_t)j;
            if (j && F[j - 1]) {
                for (rle = j + 1; rle < 256 && F[rle]; rle++) {}
                rle -= j + 1;
                *cp++ = (uint8_t)rle;
            }
        }
        if (f < 128) *cp++ = (uint8_t)f;
        else { *cp++ = (uint8_t)(128u | (f >> 8)); *cp++ = (uint8_t)(f & 0xffu); }
    }
    *cp++ = 0;
    return cp;
}

// order-1 scratch (32-bit words): F[256][256] counts -> frequencies, T[256], C16[256][258]
constexpr uint32_t O1_F = 0, O1_T = 65536, O1_C = 65792;
constexpr uint32_t O1_WORDS = 65792 + (256 * 258) / 2 + 16;

__device__ __forceinline__ uint4 ld16(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }   // any alignment

// fn(c, prev) for every byte c = src[i] with prev = src[i - 1] (0 for i = 0), 16 contiguous bytes per lane and step
template <typename F>
__device__ __forceinline__ void for_each_pair(const uint8_t *src, uint32_t n, int sub, F fn) {
    uint32_t i0 = 0;
    for (; i0 + 64u <= n; i0 += 64u) {
        const uint32_t p = i0 + 16u * (uint32_t)sub;
        const uint4 v = ld16(src + p);
        uint32_t prev = p ? src[p - 1] : 0u;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) { const uint32_t c = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu; fn(c, prev); prev = c; }
    }
    for (uint32_t i = i0 + (uint32_t)sub; i < n; i += N) fn((uint32_t)src[i], i ? (uint32_t)src[i - 1] : 0u);
}

__global__ __launch_bounds__(WAVES * 64)
void rans4x8_encode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc,
                           const uint8_t *__restrict__ order_in, uint32_t nstreams, uint8_t *out, uint32_t *out_len,
                           uint8_t *wbuf, uint32_t *scratch) {
    __shared__ GroupLds lds[WAVES * GROUPS];
    __shared__ uint32_t dpool[WAVES * GROUPS][DMAX * DMAX];      // order 1, <= DMAX symbols: counts, then start << 16 | freq
    __shared__ uint32_t rowx_s[WAVES * GROUPS][256];             // one context row in symbol space (normalise / write_table0 work on it)
    __shared__ uint32_t tt_s[WAVES * GROUPS][256];               // row totals by context value
    __shared__ __attribute__((aligned(16))) uint8_t wstage[WAVES * GROUPS][STAGE + 16];   // + a spare byte pair for lanes that emit nothing
    __shared__ uint8_t alist_s[WAVES * GROUPS][64];
    // x / f without an integer division: rcp[f] = ceil(2^(31 + ceil(log2 f)) / f), q = mulhi(x, rcp[f]) >> (ceil(log2 f) - 1), exact for x < 2^31
    __shared__ uint32_t rcp_tab[4097];
    for (uint32_t f = threadIdx.x; f <= 4096u; f += WAVES * 64) {
        uint32_t sh = 0;
        while (f > (1u << sh)) sh++;
        rcp_tab[f] = f < 2u ? 0u : (uint32_t)((((unsigned long long)1 << (sh + 31u)) + f - 1u) / f);
    }
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 3;
    const bool idle = (lane >> 2) >= GROUPS;
    const int grp = idle ? 0 : lane >> 2;
    const int slot = (tid >> 6) * GROUPS + grp;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[slot];
    uint32_t *D = dpool[slot], *RX = rowx_s[slot], *TT = tt_s[slot];
    uint8_t *WS = wstage[slot], *AL = alist_s[slot];
    const unsigned long long gmask = 0xfull << (grp * 4);
    const int lane0 = grp * 4;

    for (uint32_t sidx = g_global; __any(sidx < nstreams); sidx += g_total) {
        const bool have = sidx < nstreams && !idle;
        uint32_t n = 0, order = 0, wcap = 0;
        const uint8_t *src = nullptr;
        uint8_t *o = nullptr, *wb = nullptr;
        uint32_t *sc = nullptr;
        if (have) {
            const hg_stream_desc d = desc[sidx];
            src = in + d.in_off; n = d.in_len; o = out + d.out_off;
            order = order_in[sidx] & 1u;
            if (order && n < 4) order = 0;
            sc = scratch + d.scratch_off;
            wcap = (2u * n + 64u) & ~15u;                          // a multiple of 16 inside the stream's slice of the byte buffer
            wb = wbuf + (uint64_t)d.reserved * 16ull;
        }
        const bool core = have && n != 0;
        uint32_t tab = 0;
        bool dense = false;
        uint32_t nsym = 0;
        if (have && !core && sub == 0) { for (int i = 0; i < 9; i++) o[i] = 0; out_len[sidx] = 9; }   // empty input
        if (core && order == 0) {
            for (int j = sub; j < 256; j += N) G.H[j] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            for_each_pair(src, n, sub, [&](uint32_t c, uint32_t) { atomicAdd(&G.H[c], 1u); });
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (sub == 0) {
                normalise(G.H, n);
                uint8_t *cp = write_table0(o + 9, G.H);
                tab = (uint32_t)(cp - (o + 9));
                uint32_t x = 0;
                for (int j = 0; j < 256; j++) { G.C[j] = (uint16_t)x; x += G.H[j]; }
                G.C[256] = (uint16_t)x;
            }
        } else if (core) {
            const uint32_t q = n >> 2;
            // which byte values occur (0 always: the states start in context 0), dense numbering G.C[value] = rank
            for (int j = sub; j < 256; j += N) G.H[j] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            for_each_pair(src, n, sub, [&](uint32_t c, uint32_t) { G.H[c] = 1; });
            if (sub == 0) G.H[0] = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (sub == 0) for (int j = 0; j < 256; j++) { G.C[j] = (uint16_t)nsym; if (G.H[j]) { if (nsym < 64u) AL[nsym] = (uint8_t)j; nsym++; } }
            nsym = (uint32_t)__shfl((int)nsym, lane0, 64);
            dense = nsym <= DMAX;
            if (dense) {
                for (uint32_t i = (uint32_t)sub; i < nsym * nsym; i += N) D[i] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                for_each_pair(src, n, sub, [&](uint32_t c, uint32_t l) { atomicAdd(&D[(uint32_t)G.C[l] * nsym + G.C[c]], 1u); });
                if (sub >= 1) atomicAdd(&D[(uint32_t)G.C[0] * nsym + G.C[src[(uint32_t)sub * q]]], 1u);   // states start in ctx 0
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (sub == 0) {
                    // everything else about the table on LDS, by one lane: alphabets here are small
                    for (int i = 0; i < 256; i++) TT[i] = 0;
                    for (uint32_t a2 = 0; a2 < nsym; a2++) { uint32_t t = 0; for (uint32_t b2 = 0; b2 < nsym; b2++) t += D[a2 * nsym + b2]; TT[AL[a2]] = t; }
                    uint8_t *cp = o + 9;
                    int rle_i = 0;
                    for (int i = 0; i < 256; i++) {
                        if (!TT[i]) continue;
                        if (rle_i) rle_i--;
                        else {
                            *cp++ = (uint8_t)i;
                            if (i && TT[i - 1]) {
                                for (rle_i = i + 1; rle_i < 256 && TT[rle_i]; rle_i++) {}
                                rle_i -= i + 1;
                                *cp++ = (uint8_t)rle_i;
                            }
                        }
                        const uint32_t ri = G.C[i];
                        for (int j = 0; j < 256; j++) RX[j] = 0;
                        for (uint32_t b2 = 0; b2 < nsym; b2++) RX[AL[b2]] = D[ri * nsym + b2];
                        normalise(RX, TT[i]);
                        cp = write_table0(cp, RX);
                        uint32_t x = 0;
                        for (uint32_t b2 = 0; b2 < nsym; b2++) { const uint32_t f = RX[AL[b2]]; D[ri * nsym + b2] = (x << 16) | f; x += f; }
                    }
                    *cp++ = 0;
                    tab = (uint32_t)(cp - (o + 9));
                }
            } else {
                // ---- large alphabets: the round-1 path through global scratch --------------------------------
                for (uint32_t i = (uint32_t)sub; i < O1_C; i += N) sc[i] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                for_each_pair(src, n, sub, [&](uint32_t c, uint32_t l) { atomicAdd(&sc[O1_F + l * 256u + c], 1u); atomicAdd(&sc[O1_T + l], 1u); });
                if (sub >= 1) { atomicAdd(&sc[O1_F + src[(uint32_t)sub * q]], 1u); atomicAdd(&sc[O1_T], 1u); }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                for (uint32_t i = (uint32_t)sub; i < 256; i += N) {                   // one context row per lane
                    const uint32_t T = sc[O1_T + i];
                    if (!T) continue;
                    uint32_t *row = sc + O1_F + i * 256u;
                    normalise(row, T);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (sub == 0) {
                    uint8_t *cp = o + 9;
                    int rle_i = 0;
                    for (int i = 0; i < 256; i++) {
                        if (!sc[O1_T + i]) continue;
                        if (rle_i) rle_i--;
                        else {
                            *cp++ = (uint8_t)i;
                            if (i && sc[O1_T + i - 1]) {
                                for (rle_i = i + 1; rle_i < 256 && sc[O1_T + rle_i]; rle_i++) {}
                                rle_i -= i + 1;
                                *cp++ = (uint8_t)rle_i;
                            }
                        }
                        cp = write_table0(cp, *(const uint32_t(*)[256])(sc + O1_F + (uint32_t)i * 256u));
                    }
                    *cp++ = 0;
                    tab = (uint32_t)(cp - (o + 9));
                }
                uint16_t *C16w = (uint16_t *)(sc + O1_C);
                for (uint32_t i = (uint32_t)sub; i < 256; i += N) {
                    const uint32_t *row = sc + O1_F + i * 256u;
                    uint32_t x = 0;
                    for (int j = 0; j < 256; j++) { C16w[i * 258u + j] = (uint16_t)x; x += sc[O1_T + i] ? row[j] : 0u; }
                    C16w[i * 258u + 256] = (uint16_t)x;
                }
            }
        }
        tab = (uint32_t)__shfl((int)tab, lane0, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- encode backwards ------------------------------------------------------------------
        uint32_t R = RANS_L, wpos = wcap, whi = wcap;             // bytes in [wpos, whi) are still in the LDS ring
        const uint16_t *C16 = (const uint16_t *)(sc + O1_C);
        auto prep = [&](auto mode, bool mine, uint32_t sym, uint32_t ctx, uint32_t &f, uint32_t &start, uint32_t &rc) {
            constexpr int MODE = decltype(mode)::value;
            f = 1; start = 0; rc = 0;
            if (mine) {
                if (MODE == 0) { start = G.C[sym]; f = (uint32_t)G.C[sym + 1] - start; }
                else if (MODE == 1) { const uint32_t e = D[(uint32_t)G.C[ctx] * nsym + G.C[sym]]; start = e >> 16; f = e & 0xffffu; }
                else { start = C16[ctx * 258u + sym]; f = sc[O1_F + ctx * 256u + sym]; }
                rc = rcp_tab[f & 0x1fffu];
            }
        };
        auto step = [&](bool mine, uint32_t f, uint32_t start, uint32_t rc) {
            uint32_t cnt = 0;
            const uint32_t x_max = ((RANS_L >> TF_SHIFT) << 8) * f;
            if (mine && R >= x_max) { cnt = 1; if ((R >> 8) >= x_max) cnt = 2; }
            const unsigned long long b1 = __ballot(cnt >= 1) & gmask, b2 = __ballot(cnt >= 2) & gmask;
            const unsigned long long abv = ~((2ull << lane) - 1ull);
            const uint32_t above = (uint32_t)__popcll(b1 & abv) + (uint32_t)__popcll(b2 & abv);
            const uint32_t tot = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
            // lanes that emit nothing write to the spare bytes behind the ring
            WS[cnt >= 1 ? ((wpos - above - 1u) & (STAGE - 1u)) : STAGE] = (uint8_t)R;
            WS[cnt == 2 ? ((wpos - above - 2u) & (STAGE - 1u)) : STAGE + 1u] = (uint8_t)(R >> 8);
            R >>= 8u * cnt;
            wpos -= tot;
            const uint32_t qd = f < 2u ? R : __umulhi(R, rc) >> (31u - (uint32_t)__builtin_clz(f - 1u));
            const uint32_t Rn = (qd << TF_SHIFT) + (R - qd * f) + start;
            R = mine ? Rn : R;
        };
        // staged bytes down to the next multiple of 16 above wpos -> the byte buffer, 16 bytes per lane and store (both sides aligned)
        auto flush_bytes = [&](uint32_t room, bool all) {
            if (!all && (whi - wpos) + room * 8u <= STAGE) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const uint32_t lo = all ? wpos : (wpos + 15u) & ~15u;
            uint32_t a16 = (lo + 15u) & ~15u;
            if (a16 > whi) a16 = whi;
            for (uint32_t w = lo + (uint32_t)sub; w < a16; w += N) wb[w] = WS[w & (STAGE - 1u)];          // unaligned head (final flush only)
            for (uint32_t w = a16 + 16u * (uint32_t)sub; w + 16u <= whi; w += 16u * N) *(uint4 *)(wb + w) = *(const uint4 *)(WS + (w & (STAGE - 1u)));
            whi = lo;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        };
        auto push = [&](bool mine, uint32_t sym, uint32_t ctx) {
            uint32_t f, start, rc;
            flush_bytes(1, false);
            if (order == 0) prep(M0{}, mine, sym, ctx, f, start, rc);
            else if (dense) prep(M1{}, mine, sym, ctx, f, start, rc);
            else prep(M2{}, mine, sym, ctx, f, start, rc);
            step(mine, f, start, rc);
        };
        constexpr int CH = 8, CH1 = 16;
        if (core && order == 0) {
            const uint32_t rem = n & 3u;
            {   // tail: states 0..rem-1 take in[n-rem+z]
                const bool mine = (uint32_t)sub < rem;
                push(mine, mine ? src[n - rem + sub] : 0u, 0u);
            }
            uint32_t i = n & ~3u;
            if (i >= (uint32_t)(CH * N)) {
                uint32_t sy[CH], nx[CH];
#pragma unroll
                for (int k = 0; k < CH; k++) sy[k] = src[i - (uint32_t)((k + 1) * N) + sub];
                for (; i >= (uint32_t)(CH * N); i -= CH * N) {
                    const bool more = i >= (uint32_t)(2 * CH * N);
#pragma unroll
                    for (int k = 0; k < CH; k++) nx[k] = more ? src[i - (uint32_t)((CH + k + 1) * N) + sub] : 0u;
                    flush_bytes(CH, false);
                    uint32_t f[CH], st[CH], rc[CH];
#pragma unroll
                    for (int k = 0; k < CH; k++) prep(M0{}, true, sy[k], 0u, f[k], st[k], rc[k]);
#pragma unroll
                    for (int k = 0; k < CH; k++) step(true, f[k], st[k], rc[k]);
#pragma unroll
                    for (int k = 0; k < CH; k++) sy[k] = nx[k];
                }
            }
            for (; i > 0; i -= 4) push(true, src[i - 4 + sub], 0u);
        } else if (core) {
            const uint32_t q = n >> 2;
            long idx = (long)((uint32_t)(sub + 1) * q) - 2;
            uint32_t l = src[(uint32_t)(sub + 1) * q - 1];
            if (sub == 3) { l = src[n - 1]; idx = (long)n - 2; }
            for (long t = (long)n - 2; t > (long)(4 * q) - 2; t--) {
                const bool mine = sub == 3;
                const uint32_t c = mine ? src[idx] : 0u;
                push(mine, l, c);
                if (mine) { l = c; idx--; }
            }
            uint32_t s = 0;
            if (s + 1 + CH1 <= q) {
                uint4 cur = ld16(src + idx - (CH1 - 1));
                auto chunks = [&](auto mode) {
                    for (; s + 1 + CH1 <= q; s += CH1) {
                        const uint4 nxt = s + 1 + 2 * CH1 <= q ? ld16(src + idx - (2 * CH1 - 1)) : cur;
                        flush_bytes(CH1, false);
                        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
                        uint32_t f[CH1], st[CH1], rc[CH1], cs[CH1];
#pragma unroll
                        for (int k = 0; k < CH1; k++) cs[k] = (w[(15 - k) >> 2] >> (((15 - k) & 3) * 8)) & 0xffu;   // cs[k] = src[idx - k]
#pragma unroll
                        for (int k = 0; k < CH1; k++) prep(mode, true, k ? cs[k - 1] : l, cs[k], f[k], st[k], rc[k]);
#pragma unroll
                        for (int k = 0; k < CH1; k++) step(true, f[k], st[k], rc[k]);
                        l = cs[CH1 - 1]; idx -= CH1; cur = nxt;
                    }
                };
                if (dense) chunks(M1{}); else chunks(M2{});
            }
            for (; s + 1 < q; s++) {
                const uint32_t c = src[idx];
                push(true, l, c);
                l = c; idx--;
            }
            push(true, l, 0u);
        }
        if (core) {
            flush_bytes(0, true);
            wpos -= 16u;
            uint8_t *w = wb + wpos + 4u * sub;
            w[0] = (uint8_t)R; w[1] = (uint8_t)(R >> 8); w[2] = (uint8_t)(R >> 16); w[3] = (uint8_t)(R >> 24);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const uint32_t body = wcap - wpos;
            uint8_t *dst = o + 9 + tab;
            for (uint32_t i = (uint32_t)sub; i < body; i += N) dst[i] = wb[wpos + i];
            if (sub == 0) {
                const uint32_t csz = tab + body;
                o[0] = (uint8_t)order;
                for (int k = 0; k < 4; k++) { o[1 + k] = (uint8_t)(csz >> (8 * k)); o[5 + k] = (uint8_t)(n >> (8 * k)); }
                out_len[sidx] = 9 + csz;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace hgq

namespace hg {
uint32_t rans4x8_enc_scratch_words(uint32_t order) { return order ? hgq::O1_WORDS : 16u; }

int launch_rans4x8_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_order, size_t n,
                          void *d_out, uint32_t *d_out_len, void *d_wbuf, uint32_t *d_scratch, hipStream_t s) {
    if (n == 0) return HG_OK;
    size_t wgs = (n + hgq::WAVES * hgq::GROUPS - 1) / (hgq::WAVES * hgq::GROUPS);
    const size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgq::rans4x8_encode_kernel, dim3((unsigned)wgs), dim3(hgq::WAVES * 64), 0, s, (const uint8_t *)d_in,
                       d_desc, d_order, (uint32_t)n, (uint8_t *)d_out, d_out_len, (uint8_t *)d_wbuf, d_scratch);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

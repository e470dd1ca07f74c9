// ransnx16_enc.hip -- CRAM 3.1 "rANS Nx16" block ENCODER for MI355X (gfx950 / CDNA4).
//
// Replaces rans_compress_4x16() as called by cram_compress_by_method (reference
// cram/cram_io.c:1853-1866; htscodecs is an absent submodule -- format per
// oracle/ransnx16_oracle.c, PARITY UNPINNED).  The kernel reproduces the oracle's encoder byte for
// byte (same normalisation, same table layout), which is what the tests check.
//
// Mapping: like the decoder, the N (4 / 32) interleaved states of a stream live in N adjacent
// lanes.  rANS encodes backwards: each step every lane pushes one symbol into its state and the
// lanes whose state would overflow first spill their low 16 bits; the spilled words are laid out
// from the END of a per-stream buffer towards the front, the highest lane first, so the split of
// the shared word stream is again one ballot + popcount per step (of the lanes ABOVE me).
// Histograms are built with LDS atomics (order 0) or global atomics into a 256x256 scratch matrix
// (order 1); the small serial parts (normalisation, table serialisation) run on the group's
// first lane / are spread one context row per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hge {

constexpr uint32_t RANS_L = 1u << 15;
constexpr int WAVES = 4;
enum { F_ORDER = 1, F_X32 = 4, F_NOSZ = 16, F_CAT = 32 };

struct GroupLds { uint32_t H[256]; uint16_t C[258]; uint16_t pad[2]; };

__device__ __forceinline__ int put_u7(uint8_t *cp, uint32_t v) {
    uint8_t tmp[5]; int n = 0;
    do { tmp[n++] = v & 0x7f; v >>= 7; } while (v);
    for (int i = n - 1; i >= 0; i--) *cp++ = tmp[i] | (i ? 0x80 : 0);
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
constexpr uint32_t O1_F = 0, O1_T = 65536, O1_A = 65792, O1_C = 66048;
constexpr uint32_t O1_WORDS = 66048 + (256 * 258) / 2 + 16;
constexpr uint32_t DENSE_MAX = 64;

template <int N>
__global__ __launch_bounds__(WAVES * 64)
void ransnx16_encode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc,
                            const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel, uint32_t nsel,
                            uint8_t *out, uint32_t *out_len, uint8_t *wbuf, uint32_t *scratch) {
    constexpr int GROUPS = 64 / N;
    __shared__ GroupLds lds[WAVES * GROUPS];
    // 32-way streams (the big data series): order-1 counts, then (start << 16 | freq), for alphabets of <= 64 symbols
    __shared__ uint32_t dpool[N == 32 ? WAVES * GROUPS : 1][N == 32 ? DENSE_MAX * DENSE_MAX : 1];
    __shared__ uint8_t ipool[N == 32 ? WAVES * GROUPS : 1][64];
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & (N - 1), grp = lane / N;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[(tid >> 6) * GROUPS + grp];
    const unsigned long long gmask = ((N == 32 ? 0xffffffffull : 0xfull)) << (grp * N);
    const int lane0 = grp * N;

    for (uint32_t k = g_global; __any(k < nsel); k += g_total) {
        const bool have = k < nsel;
        const uint32_t sidx = have ? sel[k] : 0;
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
            if ((flags & F_ORDER) && n < 2u * N) flags &= ~(uint32_t)F_ORDER;   // tiny inputs: order 0
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
            for (int j = sub; j < 256; j += N) G.H[j] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (uint32_t i = (uint32_t)sub; i < n; i += N) atomicAdd(&G.H[src[i]], 1u);
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
            const uint32_t per = n / N;
            for (int j = sub; j < 256; j += N) G.H[j] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (uint32_t i = (uint32_t)sub; i < n; i += N) G.H[src[i]] = 1;
            if (sub == 0) G.H[0] = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            // dense numbering of the alphabet (G.C[value] = index), only the rows / columns that exist are touched below
            uint32_t nsym = 0;
            if (sub == 0) { for (int j = 0; j < 256; j++) { G.C[j] = (uint16_t)nsym; if (G.H[j]) { if (N == 32) isym[nsym & 63u] = (uint8_t)j; nsym++; } } }
            nsym = (uint32_t)__shfl((int)nsym, lane0, 64);
            dense = N == 32 && nsym <= DENSE_MAX;
            nsym_d = nsym;
            for (uint32_t i = 0; i < 256; i++) {
                if (!G.H[i]) continue;
                for (uint32_t j = (uint32_t)sub; j < 256; j += N) sc[O1_F + i * 256u + j] = 0;
            }
            for (uint32_t i = (uint32_t)sub; i < 256; i += N) { sc[O1_T + i] = 0; sc[O1_A + i] = G.H[i]; }
            if (dense) for (uint32_t i = (uint32_t)sub; i < nsym * nsym; i += N) D[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (dense) {
                // histogram with LDS atomics on the nsym x nsym matrix, then written out in the 256 x 256 layout
                for (uint32_t i = (uint32_t)sub; i < n; i += N) {
                    const uint32_t c = src[i], l = i ? src[i - 1] : 0u;
                    atomicAdd(&D[(uint32_t)G.C[l] * nsym + G.C[c]], 1u);
                }
                if (sub >= 1) atomicAdd(&D[(uint32_t)G.C[0] * nsym + G.C[src[(uint32_t)sub * per]]], 1u);   // states start in ctx 0
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                for (uint32_t a = (uint32_t)sub; a < nsym; a += N) {
                    uint32_t tsum = 0;
                    const uint32_t ia = isym[a];
                    for (uint32_t b = 0; b < nsym; b++) { const uint32_t v = D[a * nsym + b]; sc[O1_F + ia * 256u + isym[b]] = v; tsum += v; }
                    sc[O1_T + ia] = tsum;
                }
            } else {
                for (uint32_t i = (uint32_t)sub; i < n; i += N) {
                    const uint32_t c = src[i], l = i ? src[i - 1] : 0u;
                    atomicAdd(&sc[O1_F + l * 256u + c], 1u);
                    atomicAdd(&sc[O1_T + l], 1u);
                }
                if (sub >= 1) { atomicAdd(&sc[O1_F + src[(uint32_t)sub * per]], 1u); atomicAdd(&sc[O1_T], 1u); }   // states start in ctx 0
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            // normalise one context row per lane (rows are independent)
            for (uint32_t i = (uint32_t)sub; i < 256; i += N) {
                const uint32_t T = sc[O1_T + i];
                if (!sc[O1_A + i] || !T) continue;
                uint32_t tot = round2(T);
                if (tot > 4096u) tot = 4096u;
                uint32_t *row = sc + O1_F + i * 256u;
                normalise(row, T, tot);
                sc[O1_T + i] = tot;                               // remember the stored total
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (sub == 0) {                                       // serialise the table (oracle layout)
                uint8_t *cp = o + hdr;
                *cp++ = (uint8_t)(12u << 4);
                cp = put_alphabet(cp, G.H);
                int cnt_alpha = 0;
                for (int j = 0; j < 256; j++) cnt_alpha += G.H[j] != 0;
                for (int i = 0; i < 256; i++) {
                    if (!G.H[i]) continue;
                    if (sc[O1_T + i]) {
                        // a zero frequency is written as (0, how many MORE zero entries follow): streamed with a pending
                        // run byte instead of looking ahead, the row fetched four words at a time
                        const uint4 *row4 = (const uint4 *)(sc + O1_F + (uint32_t)i * 256u);
                        uint8_t *runbyte = nullptr; uint32_t runcnt = 0;
                        for (int j4 = 0; j4 < 64; j4++) {
                            const uint4 q = row4[j4];
                            const uint32_t v4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (!G.H[j4 * 4 + k]) continue;
                                const uint32_t v = v4[k];
                                if (runbyte) { if (!v) { runcnt++; continue; } *runbyte = (uint8_t)runcnt; runbyte = nullptr; }
                                cp += put_u7(cp, v);
                                if (!v) { runbyte = cp++; runcnt = 0; }
                            }
                        }
                        if (runbyte) *runbyte = (uint8_t)runcnt;
                    } else {
                        cp += put_u7(cp, 0); *cp++ = (uint8_t)(cnt_alpha - 1);
                    }
                }
                tab = (uint32_t)(cp - (o + hdr));
            }
            // cumulative tables, shifted up to 2^12, one row per lane (only rows that are contexts)
            uint16_t *C16 = (uint16_t *)(sc + O1_C);
            for (uint32_t i = (uint32_t)sub; i < 256; i += N) {
                if (!G.H[i]) continue;
                const uint32_t tot = sc[O1_T + i];
                uint32_t *row = sc + O1_F + i * 256u;
                int sh = 0;
                while (tot && (tot << sh) < 4096u) sh++;
                uint32_t x = 0;
                for (int j = 0; j < 256; j++) {
                    const uint32_t f = G.H[j] ? row[j] << sh : 0u;
                    if (G.H[j]) { row[j] = f; if (dense) D[(uint32_t)G.C[i] * nsym + G.C[j]] = (x << 16) | f; }
                    C16[i * 258u + j] = (uint16_t)x; x += f;
                }
                C16[i * 258u + 256] = (uint16_t)x;
            }
        }
        tab = (uint32_t)__shfl((int)tab, lane0, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- encode backwards ------------------------------------------------------------------
        uint32_t R = RANS_L;
        uint32_t wpos = wcap;                                     // next free byte (from the end) in wb
        const uint16_t *C16 = (const uint16_t *)(sc + O1_C);
        auto push = [&](bool mine, uint32_t sym, uint32_t ctx) {
            uint32_t emit = 0;
            uint32_t f = 1, start = 0;
            if (mine) {
                if (order == 0) { start = G.C[sym]; f = (uint32_t)G.C[sym + 1] - start; }
                else if (dense) { const uint32_t e = D[(uint32_t)G.C[ctx] * nsym_d + G.C[sym]]; start = e >> 16; f = e & 0xffffu; }
                else { start = C16[ctx * 258u + sym]; f = sc[O1_F + ctx * 256u + sym]; }
                const uint32_t x_max = ((RANS_L >> shift) << 16) * f;
                emit = R >= x_max ? 1u : 0u;
            }
            const unsigned long long b = __ballot(emit != 0) & gmask;
            const uint32_t above = (uint32_t)__popcll(b & ~((2ull << lane) - 1ull));   // emitting lanes above me
            const uint32_t tot = (uint32_t)__popcll(b);
            if (emit) {
                uint8_t *w = wb + wpos - 2u * (above + 1u);
                w[0] = (uint8_t)R; w[1] = (uint8_t)(R >> 8);
                R >>= 16;
            }
            wpos -= 2u * tot;
            if (mine) R = ((R / f) << shift) + (R % f) + start;
        };
        if (core && order == 0) {
            const uint32_t rem = n & (uint32_t)(N - 1);
            {   // tail symbols belong to states 0..rem-1
                const bool mine = (uint32_t)sub < rem;
                push(mine, mine ? src[n - rem + sub] : 0u, 0u);
            }
            for (uint32_t i = n & ~(uint32_t)(N - 1); i > 0; i -= N) push(true, src[i - N + sub], 0u);
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
            for (uint32_t s = 0; s + 1 < per; s++) {
                const uint32_t c = src[idx];
                push(true, l, c);
                l = c; idx--;
            }
            push(true, l, 0u);                                    // first symbol of each state: context 0
        }
        // non-core groups must still take part in the ballots above?  No: push() is only called by
        // core groups, and __ballot is masked with gmask, so groups proceed independently.
        if (core) {
            // flush the states: state z ends up at (final ptr) + 4 z
            wpos -= 4u * N;
            uint8_t *w = wb + wpos + 4u * sub;
            w[0] = (uint8_t)R; w[1] = (uint8_t)(R >> 8); w[2] = (uint8_t)(R >> 16); w[3] = (uint8_t)(R >> 24);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const uint32_t body = wcap - wpos;
            uint8_t *dst = o + hdr + tab;
            for (uint32_t i = (uint32_t)sub; i < body; i += N) dst[i] = wb[wpos + i];
            if (sub == 0) { o[0] = (uint8_t)flags; out_len[sidx] = hdr + tab + body; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace hge

namespace hg {
uint32_t ransnx16_enc_scratch_words(uint32_t flags) { return (flags & 1u) ? hge::O1_WORDS : 16u; }

int launch_ransnx16_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags,
                           const uint32_t *d_sel4, size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out,
                           uint32_t *d_out_len, void *d_wbuf, uint32_t *d_scratch, hipStream_t s) {
    const size_t maxw = (size_t)ctx->cus * 8;
    const bool side = n4 != 0 && n32 != 0;                  // both variants present: overlap them
    hipStream_t s2 = side ? fork_side(ctx, s) : s;
    if (n4) {
        size_t wgs = (n4 + hge::WAVES * 16 - 1) / (hge::WAVES * 16);
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL(hge::ransnx16_encode_kernel<4>, dim3((unsigned)wgs), dim3(hge::WAVES * 64), 0, s,
                           (const uint8_t *)d_in, d_desc, d_flags, d_sel4, (uint32_t)n4, (uint8_t *)d_out, d_out_len,
                           (uint8_t *)d_wbuf, d_scratch);
    }
    if (n32) {
        size_t wgs = (n32 + hge::WAVES * 2 - 1) / (hge::WAVES * 2);
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL(hge::ransnx16_encode_kernel<32>, dim3((unsigned)wgs), dim3(hge::WAVES * 64), 0, s2,
                           (const uint8_t *)d_in, d_desc, d_flags, d_sel32, (uint32_t)n32, (uint8_t *)d_out, d_out_len,
                           (uint8_t *)d_wbuf, d_scratch);
        if (side) join_side(ctx, s);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

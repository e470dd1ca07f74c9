// rans4x8_enc.hip -- CRAM 3.0 "rANS 4x8" block ENCODER for MI355X (gfx950 / CDNA4).
//
// Replaces rans_compress() as called by cram_compress_by_method (reference
// cram/cram_io.c:1834-1848; htscodecs rANS_static.c is an absent submodule -- format per the CRAM
// v3.0 specification as restated in oracle/rans4x8_oracle.c, whose DECODER is pinned on the
// reference's CRAM fixtures).  The kernel reproduces the oracle's encoder byte for byte.
//
// Same mapping as the Nx16 encoder (ransnx16_enc.hip): the 4 states of a stream in 4 adjacent lanes,
// 16 streams per wavefront, coding backwards; here a state spills 0, 1 or 2 BYTES per symbol, so the
// split of the shared byte stream is two ballots (">= 1 byte", "2 bytes") + popcounts of the lanes
// above.  Frequencies are normalised to 4095 (stock decoders require a total < 4096).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgq {

constexpr uint32_t TF_SHIFT = 12, RANS_L = 1u << 23;
constexpr int WAVES = 4, N = 4, GROUPS = 16;

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

__global__ __launch_bounds__(WAVES * 64)
void rans4x8_encode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc,
                           const uint8_t *__restrict__ order_in, uint32_t nstreams, uint8_t *out, uint32_t *out_len,
                           uint8_t *wbuf, uint32_t *scratch) {
    __shared__ GroupLds lds[WAVES * GROUPS];
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 3, grp = lane >> 2;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[(tid >> 6) * GROUPS + grp];
    const unsigned long long gmask = 0xfull << (grp * 4);
    const int lane0 = grp * 4;

    for (uint32_t sidx = g_global; __any(sidx < nstreams); sidx += g_total) {
        const bool have = sidx < nstreams;
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
            wcap = 2u * n + 64u;
            wb = wbuf + (uint64_t)d.reserved * 16ull;
        }
        const bool core = have && n != 0;
        uint32_t tab = 0;
        if (have && !core && sub == 0) { for (int i = 0; i < 9; i++) o[i] = 0; out_len[sidx] = 9; }   // empty input
        if (core && order == 0) {
            for (int j = sub; j < 256; j += N) G.H[j] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (uint32_t i = (uint32_t)sub; i < n; i += N) atomicAdd(&G.H[src[i]], 1u);
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
            for (uint32_t i = (uint32_t)sub; i < O1_C; i += N) sc[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (uint32_t i = (uint32_t)sub; i < n; i += N) {
                const uint32_t c = src[i], l = i ? src[i - 1] : 0u;
                atomicAdd(&sc[O1_F + l * 256u + c], 1u);
                atomicAdd(&sc[O1_T + l], 1u);
            }
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
            uint16_t *C16 = (uint16_t *)(sc + O1_C);
            for (uint32_t i = (uint32_t)sub; i < 256; i += N) {
                const uint32_t *row = sc + O1_F + i * 256u;
                uint32_t x = 0;
                for (int j = 0; j < 256; j++) { C16[i * 258u + j] = (uint16_t)x; x += sc[O1_T + i] ? row[j] : 0u; }
                C16[i * 258u + 256] = (uint16_t)x;
            }
        }
        tab = (uint32_t)__shfl((int)tab, lane0, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t R = RANS_L, wpos = wcap;
        const uint16_t *C16 = (const uint16_t *)(sc + O1_C);
        auto push = [&](bool mine, uint32_t sym, uint32_t ctx) {
            uint32_t f = 1, start = 0, cnt = 0;
            if (mine) {
                if (order == 0) { start = G.C[sym]; f = (uint32_t)G.C[sym + 1] - start; }
                else { start = C16[ctx * 258u + sym]; f = sc[O1_F + ctx * 256u + sym]; }
                const uint32_t x_max = ((RANS_L >> TF_SHIFT) << 8) * f;
                if (R >= x_max) { cnt = 1; if ((R >> 8) >= x_max) cnt = 2; }
            }
            const unsigned long long b1 = __ballot(cnt >= 1) & gmask, b2 = __ballot(cnt >= 2) & gmask;
            const unsigned long long abv = ~((2ull << lane) - 1ull);
            const uint32_t above = (uint32_t)__popcll(b1 & abv) + (uint32_t)__popcll(b2 & abv);
            const uint32_t tot = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
            if (cnt) {
                wb[wpos - above - 1u] = (uint8_t)R; R >>= 8;
                if (cnt == 2) { wb[wpos - above - 2u] = (uint8_t)R; R >>= 8; }
            }
            wpos -= tot;
            if (mine) R = ((R / f) << TF_SHIFT) + (R % f) + start;
        };
        if (core && order == 0) {
            const uint32_t rem = n & 3u;
            {   // tail: states 0..rem-1 take in[n-rem+z]
                const bool mine = (uint32_t)sub < rem;
                push(mine, mine ? src[n - rem + sub] : 0u, 0u);
            }
            for (uint32_t i = n & ~3u; i > 0; i -= 4) push(true, src[i - 4 + sub], 0u);
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
            for (uint32_t s = 0; s + 1 < q; s++) {
                const uint32_t c = src[idx];
                push(true, l, c);
                l = c; idx--;
            }
            push(true, l, 0u);
        }
        if (core) {
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

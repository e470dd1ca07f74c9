// ransnx16_xform.hip -- the byte transforms that follow entropy decoding in a CRAM 3.1 rANS Nx16
// stream (flags RLE 0x40, PACK 0x80, STRIPE 0x08), on MI355X (gfx950).
//
// Reference boundary: rans_uncompress_4x16 as called at cram/cram_io.c:1697-1714 (implementation:
// htscodecs rANS_static4x16pr.c -- absent submodule).  Semantics follow oracle/ransnx16_oracle.c
// (rle_decode, unpack, the STRIPE loop of uncompress_inner): PARITY UNPINNED like the rest of Nx16.
//
// One wavefront per job, three optional phases:
//   RLE    literals + meta (symbol list, then one uint7 run length per occurrence of a listed symbol).
//          64 literals per step: the run lengths are pulled from a small LDS FIFO that is refilled 64
//          meta bytes at a time (the bytes without a continuation bit mark the varint ends -- one
//          ballot), a wave scan turns the lengths into output offsets and the expanded bytes are
//          written OUTPUT-parallel (each lane finds its literal by binary search in the 64 offsets),
//          so one long run and 64 short ones cost the same.
//   PACK   1/2/4-bit symbol indices -> bytes through the <=16-entry map.
//   write  with a stride: STRIPE sub-stream k of S writes byte i to out[i*S + k], so the
//          de-interleave costs no extra pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgx {
using hg::wave_sync;

constexpr int WAVES = 4;
struct WaveLds {
    uint32_t fifo[256];
    uint32_t pre[64];
    uint8_t sym[64];
    uint8_t mb[64];
    uint8_t isr[256];
};

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_incl_scan64(uint32_t x, int lane) {
    unsigned long long v = x;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
        if (lane >= d) v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(WAVES * 64)
void nx16_xform_kernel(const uint8_t *__restrict__ in, uint8_t *work, uint8_t *out, const hg::nx16_xform *__restrict__ jobs,
                       uint32_t njobs, int32_t *status, uint32_t status_base) {
    __shared__ WaveLds lds[WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    WaveLds &S = lds[wv];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t j = blockIdx.x * WAVES + wv; j < njobs; j += gridDim.x * WAVES) {
        const hg::nx16_xform J = jobs[j];
        int err = 0;
        if (J.dep0 != 0xffffffffu && status[J.dep0] != 0) err = 1;
        if (J.dep1 != 0xffffffffu && status[J.dep1] != 0) err = 1;
        const uint8_t *src = work + J.s1_off;
        uint32_t src_len = J.lit_len;
        // ---------------------------------------------------------------- RLE
        if (!err && (J.ops & 1u)) {
            const uint8_t *meta = ((J.ops & 4u) ? (const uint8_t *)work : in) + J.meta_off;
            const bool to_out = !(J.ops & 2u);
            uint8_t *dst = to_out ? out + J.out_off : work + J.s2_off;
            const uint32_t dstride = to_out ? J.stride : 1u;
            uint32_t nr = 0, mp = 0;
            if (J.meta_len < 1) err = 1;
            else { nr = meta[0]; if (nr == 0) nr = 256; if (1u + nr > J.meta_len) err = 1; }
            for (int k = lane; k < 256; k += 64) S.isr[k] = 0;
            wave_sync();
            if (!err) for (uint32_t k = (uint32_t)lane; k < nr; k += 64) S.isr[meta[1 + k]] = 1;
            wave_sync();
            mp = 1u + nr;
            uint32_t head = 0, tail = 0, avail = 0;
            unsigned long long o = 0;
            for (uint32_t i0 = 0; !err && i0 < J.lit_len; i0 += 64) {
                const uint32_t nl = J.lit_len - i0 < 64u ? J.lit_len - i0 : 64u;
                const bool has = (uint32_t)lane < nl;
                const uint8_t c = has ? src[i0 + lane] : 0;
                const bool run = has && S.isr[c];
                const unsigned long long B = __ballot(run);
                const uint32_t need = (uint32_t)__popcll(B);
                while (!err && avail < need && mp < J.meta_len) {            // refill the run-length FIFO
                    const uint32_t nb = J.meta_len - mp < 64u ? J.meta_len - mp : 64u;
                    const uint8_t b = (uint32_t)lane < nb ? meta[mp + lane] : 0x80;
                    const unsigned long long T = __ballot(!(b & 0x80));
                    S.mb[lane] = b;
                    wave_sync();
                    if (T == 0) { if (nb == 64u) err = 1; break; }           // 64 continuation bytes / truncated tail
                    if ((T >> lane) & 1ull) {
                        const unsigned long long tb = T & below;
                        const int prev = tb ? 63 - __clzll(tb) : -1;
                        const int n = lane - prev;
                        if (n > 5) err = 1;
                        else {
                            uint32_t v = 0;
                            for (int q = prev + 1; q <= lane; q++) v = (v << 7) | (S.mb[q] & 0x7fu);
                            S.fifo[(tail + (uint32_t)__popcll(tb)) & 255u] = v;
                        }
                    }
                    err = __any(err) ? 1 : 0;
                    const uint32_t cnt = (uint32_t)__popcll(T);
                    tail += cnt; avail += cnt;
                    mp += 64u - (uint32_t)__clzll(T);
                    wave_sync();
                }
                if (!err && avail < need) err = 1;
                if (err) break;
                const uint32_t r = run ? S.fifo[(head + (uint32_t)__popcll(B & below)) & 255u] : 0u;
                head += need; avail -= need;
                // run lengths are bounded by the output size: anything larger is an overrun
                if (has && r >= J.plen) err = 1;
                err = __any(err) ? 1 : 0;
                if (err) break;
                const uint32_t len = has ? r + 1u : 0u;                             // <= plen < 2^32
                const unsigned long long incl = wave_incl_scan64(len, lane);
                const unsigned long long total64 = shfl64(incl, 63);
                if (total64 > (unsigned long long)J.plen - o) { err = 1; break; }
                const uint32_t total = (uint32_t)total64;
                S.pre[lane] = (uint32_t)incl - len; S.sym[lane] = c;
                wave_sync();
                for (uint32_t p = (uint32_t)lane; p < total; p += 64) {
                    uint32_t lo = 0, hi = nl;                                       // largest idx with pre[idx] <= p
                    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (S.pre[mid] <= p) lo = mid; else hi = mid; }
                    dst[(o + p) * dstride] = S.sym[lo];
                }
                o += total;
                wave_sync();
            }
            if (!err && !(J.ops & 8u) && o != (unsigned long long)J.plen) err = 1;
            if (J.ops & 8u) *(unsigned long long *)(work + J.len_off) = err ? 0ull : o;          // every lane stores the same word
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            wave_sync();
            src = work + J.s2_off; src_len = J.plen;
        }
        // ---------------------------------------------------------------- PACK, or a plain strided copy
        if (!err && (J.ops & 2u)) {
            uint8_t *dst = out + J.out_off;
            if (J.nsym <= 1) {
                const uint8_t v = J.nsym ? J.map[0] : 0;
                for (uint32_t i = (uint32_t)lane; i < J.ulen; i += 64) dst[(size_t)i * J.stride] = v;
            } else {
                const uint32_t bits = J.nsym <= 2 ? 1u : J.nsym <= 4 ? 2u : 4u, lg = J.nsym <= 2 ? 3u : J.nsym <= 4 ? 2u : 1u;
                const uint32_t per = 8u / bits, vm = (1u << bits) - 1u;
                if ((unsigned long long)src_len < ((unsigned long long)J.ulen + per - 1) / per) err = 1;
                else for (uint32_t i = (uint32_t)lane; i < J.ulen; i += 64) {
                    const uint32_t v = ((uint32_t)src[i >> lg] >> ((i & (per - 1u)) * bits)) & vm;
                    if (v >= J.nsym) err = 1;
                    else dst[(size_t)i * J.stride] = J.map[v];
                }
            }
        } else if (!err && !(J.ops & 1u)) {
            uint8_t *dst = out + J.out_off;
            for (uint32_t i = (uint32_t)lane; i < J.ulen; i += 64) dst[(size_t)i * J.stride] = src[i];
        }
        err = __any(err) ? 1 : 0;
        status[status_base + j] = err ? -1 : 0;                  // every lane stores the same word
        wave_sync();
    }
}

}  // namespace hgx

namespace hg {
int launch_ransnx16_xform(hg_ctx *ctx, const void *d_in, void *d_work, void *d_out, const nx16_xform *d_jobs, size_t njobs,
                          int32_t *d_status, uint32_t status_base, hipStream_t s) {
    if (!njobs) return HG_OK;
    size_t wgs = (njobs + hgx::WAVES - 1) / hgx::WAVES;
    const size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgx::nx16_xform_kernel, dim3((unsigned)wgs), dim3(hgx::WAVES * 64), 0, s, (const uint8_t *)d_in,
                       (uint8_t *)d_work, (uint8_t *)d_out, d_jobs, (uint32_t)njobs, d_status, status_base);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

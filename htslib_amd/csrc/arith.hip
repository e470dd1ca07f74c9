// arith.hip -- CRAM 3.1 adaptive arithmetic ("range") coder, block method 6, for MI355X (gfx950).
//
// Replaces arith_uncompress_to / arith_compress_to as called by cram_uncompress_block /
// cram_compress_by_method (reference cram/cram_io.c:1716-1733, 1869-1883; implementation = htscodecs
// arith_dynamic.c, an ABSENT submodule).  Format and arithmetic per oracle/arith_oracle.c -- PARITY
// UNPINNED; the kernels are bit-exact with that oracle.
//
// The coder is adaptive: symbol i+1 cannot be decoded before the model update of symbol i, so a stream
// is one serial dependency chain and the parallelism is ACROSS streams (a CRAM slice holds ~25 blocks, a
// batch of slices thousands; STRIPE adds 4 per block).  Mapping: one wavefront per stream.
//   * range-coder registers (low/code, range, input cursor) are wave-uniform scalars;
//   * a model is an array of (freq << 8 | symbol) words kept sorted by frequency in LDS (global scratch when
//     an order-1 alphabet is too big for the per-wave pool); the symbol search is done by all 64 lanes at
//     once: one coalesced LDS read, a DPP prefix sum, one ballot -- constant time instead of the CPU's walk
//     down the list; the halving of a model is lane-parallel as well;
//   * the compressed bytes are read 64 at a time into one VGPR and picked out with v_readlane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

#include "arith_dev.h"

namespace hga {

// Layout of a stream's model memory (32-bit words): nl literal models of m entries, their nl totals, then
// (RLE) 258 run models of 4 entries and their 258 totals.
// The totals are addressed through their own pointer TT: when the models live in global scratch (order-1 alphabets too big for the
// pool) the totals move into the otherwise idle LDS pool -- one global round trip less on every symbol's dependency chain.
struct Models {
    uint32_t *M; uint32_t *TT; uint32_t m, nl; bool split;
    __device__ uint32_t lit(uint32_t k) const { return k * m; }
    __device__ uint32_t lit_tot(uint32_t k) const { return split ? k : nl * m + k; }
    __device__ uint32_t run(uint32_t k) const { return nl * (m + 1) + k * 4; }
    __device__ uint32_t run_tot(uint32_t k) const { return split ? nl + k : nl * (m + 1) + 258 * 4 + k; }
    __device__ void place(uint32_t *pool, uint32_t poolw, uint32_t *global, uint32_t words) {
        if (words <= poolw) { M = pool; TT = pool; split = false; } else { M = global; TT = pool; split = true; }
    }
};
__host__ __device__ inline uint32_t model_words(uint32_t m, uint32_t order, uint32_t rle) { return (order ? m : 1u) * (m + 1u) + (rle ? 258u * 5u : 0u); }

__device__ void models_init(const Models &Q, bool rle, int lane) {
    for (uint32_t i = (uint32_t)lane; i < Q.nl * Q.m; i += 64) Q.M[i] = (1u << 8) | (i % Q.m);
    for (uint32_t i = (uint32_t)lane; i < Q.nl; i += 64) Q.TT[Q.lit_tot(i)] = Q.m;
    if (rle) {
        for (uint32_t i = (uint32_t)lane; i < 258 * 4; i += 64) Q.M[Q.run(0) + i] = (1u << 8) | (i & 3u);
        for (uint32_t i = (uint32_t)lane; i < 258; i += 64) Q.TT[Q.run_tot(i)] = 4u;
    }
    wave_sync();
}

// MODE 0: every stream of the list; 1: only streams whose models fit the pool; 2: only streams whose models do NOT fit HG_ARITH_POOL_BIG (global
// scratch; the pool then holds their totals only).  1 and 2 walk the same list ("big" streams): a stream with global models needs 3 KiB of LDS, not the
// 64 KiB of the big pool -- which kept its workgroups from sharing a CU with the 68 KiB workgroups of the rANS kernels (they ran one after the other).
template <int POOLW, int WAVES, int MODE = 0>
__global__ __launch_bounds__(WAVES * 64)
void arith_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint32_t *__restrict__ sel,
                         uint32_t nsel, uint8_t *out, int32_t *status, uint32_t *gscratch) {
    __shared__ uint32_t pool[WAVES][POOLW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t k = blockIdx.x * WAVES + wv; k < nsel; k += gridDim.x * WAVES) {
        const uint32_t sidx = sel[k];
        const hg_stream_desc d = desc[sidx];
        const uint32_t flags = d.reserved & 0xffu, order = flags & F_ORDER, rle = (flags & F_RLE) ? 1u : 0u;
        const uint8_t *cp = in + d.in_off;
        uint8_t *o = out + d.out_off;
        const uint32_t n = d.out_len;
        int err = 0;
        if (d.in_len < 1) err = 1;
        // The model pointer is LDS in one instantiation of the body and global memory in the other: with a pointer chosen at run
        // time every model access was a FLAT load (slow for LDS, and its wait also covers every global store in flight).
        auto body = [&](auto lds_models) {
            constexpr bool LM = decltype(lds_models)::value;
            Models Q;
            Q.m = cp[0] ? cp[0] : 256u; Q.nl = order ? Q.m : 1u;
            Q.TT = pool[wv]; Q.split = !LM;
            if (LM) Q.M = pool[wv]; else Q.M = gscratch + d.scratch_off;
            models_init(Q, rle != 0, lane);
            Decoder D;
            D.start(cp + 1, d.in_len - 1, lane);
            uint32_t last = 0, keep = 0;
            // every model goes through the lean step of arith_dev.h (the symbol is nearly always among a model's first 64 entries)
            WideO0 W;
            auto loops = [&](auto wide_c) {
            constexpr bool WIDE = decltype(wide_c)::value;           // order 0, 65..256 symbols, model in LDS: arith_dev.h WideO0
            if (WIDE) W.init(Q.M, nullptr, Q.m, false, lane);
            auto lit_sym = [&](uint32_t ctx) -> uint32_t {
                if (WIDE) return wide_decode(D, W, lane);
                return D.template symbol_lean<LM>(Q.M, Q.TT, Q.lit(ctx), Q.m, Q.lit_tot(ctx), lane);
            };
            auto run_sym = [&](uint32_t rctx) -> uint32_t {
                return D.template symbol_lean<LM>(Q.M, Q.TT, Q.run(rctx), 4, Q.run_tot(rctx), lane);
            };
            if (!rle) {
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t ctx = order ? last : 0u;
                    const uint32_t c = lit_sym(ctx);
                    if (D.err) break;
                    last = c;
                    if ((uint32_t)lane == (i & 63u)) keep = c;
                    if ((i & 63u) == 63u) o[i - 63u + (uint32_t)lane] = (uint8_t)keep;   // 64 symbols per store
                }
                if (!D.err && (n & 63u) && (uint32_t)lane < (n & 63u)) o[(n & ~63u) + (uint32_t)lane] = (uint8_t)keep;
            } else {
                for (uint32_t i = 0; i < n;) {
                    const uint32_t ctx = order ? last : 0u;
                    const uint32_t c = lit_sym(ctx);
                    if (D.err) break;
                    last = c;
                    unsigned long long r = 0; uint32_t rctx = c, part;
                    do {
                        part = run_sym(rctx);
                        if (D.err) break;
                        rctx = rctx == c ? 256u : 257u;
                        r += part;
                        if (r >= n) D.err = 1;
                    } while (part == 3 && !D.err);
                    if (D.err) break;
                    if ((unsigned long long)i + 1ull + r > n) { D.err = 1; break; }
                    for (uint32_t p = (uint32_t)lane; p <= (uint32_t)r; p += 64) o[i + p] = (uint8_t)c;
                    i += (uint32_t)r + 1u;
                }
            }
            };
            if (LM && !order && Q.m > 64u) loops(std::true_type{}); else loops(std::false_type{});
            if (D.err || D.in.overrun) err = 1;
#ifdef HG_ARITH_PROFILE
            if (blockIdx.x == 0 && threadIdx.x == 0)
                printf("arith dec n=%u m=%u order=%u rle=%u lds=%d ticks/symbol: load %.0f div %.0f search %.0f renorm %.0f update %.0f outside %.0f\n", n, Q.m, order, rle, (int)LM,
                       (double)D.g_tacc[0] / n, (double)D.g_tacc[1] / n, (double)D.g_tacc[2] / n, (double)D.g_tacc[3] / n, (double)D.g_tacc[4] / n, (double)D.g_tacc[5] / n);
#endif
        };
        if (!err && n) {
            const uint32_t m0 = cp[0] ? cp[0] : 256u, words = model_words(m0, order, rle);
            if (MODE == 1 && words > (uint32_t)POOLW) continue;      // (wave-uniform) the other kernel's stream
            if (MODE == 2 && words <= (uint32_t)HG_ARITH_POOL_BIG) continue;
            if (MODE != 2 && words <= (uint32_t)POOLW) body(std::true_type{}); else body(std::false_type{});
        }
        status[sidx] = err ? -1 : 0;                                 // every lane stores the same word
        wave_sync();
    }
}

}  // namespace hga

namespace hg {
uint32_t arith_model_words(uint32_t max_sym, uint32_t flags) { return hga::model_words(max_sym ? max_sym : 256u, flags & 1u, (flags & 64u) ? 1u : 0u); }

int launch_arith_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel_small, size_t nsmall,
                        const uint32_t *d_sel_big, size_t nbig, void *d_out, int32_t *d_status, uint32_t *d_scratch, hipStream_t s) {
    const size_t maxw = (size_t)ctx->cus * 8;
    const bool side = nsmall != 0 && nbig != 0;                  // both variants present: overlap them
    hipStream_t s2 = side ? fork_side(ctx, s) : s;
    hipStream_t s3 = nbig ? fork_side3(ctx, s) : s;              // (forked BEFORE anything of this call is queued on s: the three kernels start together)
    if (nsmall) {
        size_t wgs = (nsmall + 3) / 4;
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL((hga::arith_decode_kernel<HG_ARITH_POOL_SMALL, 4>), dim3((unsigned)wgs), dim3(256), 0, s, (const uint8_t *)d_in,
                           d_desc, d_sel_small, (uint32_t)nsmall, (uint8_t *)d_out, d_status, d_scratch);
    }
    if (nbig) {
        size_t wgs = nbig;
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL((hga::arith_decode_kernel<HG_ARITH_POOL_BIG, 1, 1>), dim3((unsigned)wgs), dim3(64), 0, s2, (const uint8_t *)d_in,
                           d_desc, d_sel_big, (uint32_t)nbig, (uint8_t *)d_out, d_status, d_scratch);
        // the same list again for the streams with global models, four per workgroup and 3 KiB of LDS each, beside the others
        size_t wg4 = (nbig + 3) / 4;
        if (wg4 > maxw) wg4 = maxw;
        hipLaunchKernelGGL((hga::arith_decode_kernel<HG_ARITH_POOL_TOTALS, 4, 2>), dim3((unsigned)wg4), dim3(256), 0, s3, (const uint8_t *)d_in,
                           d_desc, d_sel_big, (uint32_t)nbig, (uint8_t *)d_out, d_status, d_scratch);
        join_side3(ctx, s);
        if (side) join_side(ctx, s);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

// ================================================================================================
// Encoder
// ================================================================================================
namespace hga {

template <int POOLW, int WAVES, int MODE = 0>
__global__ __launch_bounds__(WAVES * 64)
void arith_encode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in,
                         const uint32_t *__restrict__ sel, uint32_t nsel, uint8_t *out, uint32_t *out_len, uint32_t *gscratch) {
    __shared__ uint32_t pool[WAVES][POOLW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t k = blockIdx.x * WAVES + wv; k < nsel; k += gridDim.x * WAVES) {
        const uint32_t sidx = sel[k];
        const hg_stream_desc d = desc[sidx];
        const uint32_t flags = flags_in[sidx], order = flags & F_ORDER, rle = (flags & F_RLE) ? 1u : 0u;
        const uint8_t *src = in + d.in_off;
        uint8_t *o = out + d.out_off;
        const uint32_t n = d.in_len;
        uint32_t total = 0;
        if (n) {
            uint32_t mx = 0;
            for (uint32_t i = (uint32_t)lane; i < n; i += 64) { const uint32_t c = src[i]; mx = c > mx ? c : mx; }
            for (int s = 32; s; s >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)mx, s, 64); mx = t > mx ? t : mx; }
            auto body = [&](auto lds_models) {
            constexpr bool LM = decltype(lds_models)::value;
            Models Q;
            Q.m = mx + 1u; Q.nl = order ? Q.m : 1u;
            Q.TT = pool[wv]; Q.split = !LM;
            if (LM) Q.M = pool[wv]; else Q.M = gscratch + d.scratch_off;
            models_init(Q, rle != 0, lane);
            o[0] = (uint8_t)Q.m;                                     // every lane, same byte (256 -> 0)
            Encoder E;
            E.start(o + 1);
            uint32_t last = 0;
            WideO0 W;
            auto loops = [&](auto wide_c) {
            constexpr bool WIDE = decltype(wide_c)::value;
            if (WIDE) W.init(Q.M, (uint8_t *)(pool[wv] + (POOLW - 64)), Q.m, true, lane);     // the symbol -> position map: the pool's last 256 bytes
            auto lit_sym = [&](uint32_t ctx, uint32_t c) {
                if (WIDE) { wide_encode(E, W, c, lane); return; }
                E.template symbol_lean<LM>(Q.M, Q.TT, Q.lit(ctx), Q.m, Q.lit_tot(ctx), c, lane);
            };
            auto run_sym = [&](uint32_t rctx, uint32_t part) {
                E.template symbol_lean<LM>(Q.M, Q.TT, Q.run(rctx), 4, Q.run_tot(rctx), part, lane);
            };
            if (!rle) {
                uint32_t win = 0;
                for (uint32_t i = 0; i < n; i++) {
                    if ((i & 63u) == 0) { const uint32_t p = i + (uint32_t)lane; win = p < n ? src[p] : 0u; }
                    const uint32_t c = rl(win, i & 63u), ctx = order ? last : 0u;
                    lit_sym(ctx, c);
                    last = c;
                }
            } else {
                for (uint32_t i = 0; i < n;) {
                    const uint32_t c = src[i], ctx = order ? last : 0u;
                    lit_sym(ctx, c);
                    last = c;
                    uint32_t r = 0;                                  // how many more copies of c follow
                    for (;;) {
                        const uint32_t p = i + 1u + r + (uint32_t)lane;
                        const unsigned long long ne = __ballot(!(p < n && src[p] == c));
                        if (ne) { r += (uint32_t)__builtin_ctzll(ne); break; }
                        r += 64;
                    }
                    i += r + 1u;
                    uint32_t rctx = c, part;
                    do {
                        part = r < 3u ? r : 3u;
                        run_sym(rctx, part);
                        rctx = rctx == c ? 256u : 257u;
                        r -= part;
                    } while (part == 3u);
                }
            }
            };
            if (LM && !order && Q.m > 64u && model_words(Q.m, order, rle) + 64u <= (uint32_t)POOLW) loops(std::true_type{}); else loops(std::false_type{});
            total = 1u + E.finish(lane);
            };
            const uint32_t words = model_words(mx + 1u, order, rle);
            if (MODE == 1 && words > (uint32_t)POOLW) continue;      // (wave-uniform) the other kernel's stream
            if (MODE == 2 && words <= (uint32_t)HG_ARITH_POOL_BIG) continue;
            if (MODE != 2 && words <= (uint32_t)POOLW) body(std::true_type{}); else body(std::false_type{});
        }
        out_len[sidx] = total;                                       // every lane stores the same word
        wave_sync();
    }
}

}  // namespace hga

namespace hg {
int launch_arith_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags, const uint32_t *d_sel_small,
                        size_t nsmall, const uint32_t *d_sel_big, size_t nbig, void *d_out, uint32_t *d_out_len, uint32_t *d_scratch,
                        hipStream_t s) {
    const size_t maxw = (size_t)ctx->cus * 8;
    const bool side = nsmall != 0 && nbig != 0;                  // both variants present: overlap them
    hipStream_t s2 = side ? fork_side(ctx, s) : s;
    hipStream_t s3 = nbig ? fork_side3(ctx, s) : s;
    if (nsmall) {
        size_t wgs = (nsmall + 3) / 4;
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL((hga::arith_encode_kernel<HG_ARITH_POOL_SMALL, 4>), dim3((unsigned)wgs), dim3(256), 0, s, (const uint8_t *)d_in,
                           d_desc, d_flags, d_sel_small, (uint32_t)nsmall, (uint8_t *)d_out, d_out_len, d_scratch);
    }
    if (nbig) {
        size_t wgs = nbig;
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL((hga::arith_encode_kernel<HG_ARITH_POOL_BIG, 1, 1>), dim3((unsigned)wgs), dim3(64), 0, s2, (const uint8_t *)d_in,
                           d_desc, d_flags, d_sel_big, (uint32_t)nbig, (uint8_t *)d_out, d_out_len, d_scratch);
        size_t wg4 = (nbig + 3) / 4;
        if (wg4 > maxw) wg4 = maxw;
        hipLaunchKernelGGL((hga::arith_encode_kernel<HG_ARITH_POOL_TOTALS, 4, 2>), dim3((unsigned)wg4), dim3(256), 0, s3, (const uint8_t *)d_in,
                           d_desc, d_flags, d_sel_big, (uint32_t)nbig, (uint8_t *)d_out, d_out_len, d_scratch);
        join_side3(ctx, s);
        if (side) join_side(ctx, s);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

// ---- self-check of the short divisions of arith_dev.h against the hardware's exact expansion, on the device (tests/test_arith.py): n random pairs per
// lane in the ranges the coder produces plus the corners; returns the number of disagreements
namespace hga {
__global__ void udiv_check_kernel(unsigned long long seed, uint32_t rounds, unsigned long long *bad) {
    unsigned long long x = seed + 0x9e3779b97f4a7c15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    unsigned long long miss = 0;
    for (uint32_t i = 0; i < rounds; i++) {
        const unsigned long long r = next();
        uint32_t tot = 2u + (uint32_t)(r % 65518u);                             // a model's total: 2 .. 65519
        if ((i & 31u) == 2) tot = 1u + (uint32_t)(r % 3u);                       // ... and the totals of tiny alphabets at their first symbols (r up to 2^32 - 1)
        uint32_t range = (uint32_t)(r >> 20) | (1u << 24);                       // >= 2^24 after renormalisation
        if ((i & 15u) == 0) range = 0xffffffffu;
        if ((i & 15u) == 1) range = 1u << 24;
        if (udiv_small_divisor(range, tot) != range / tot) miss++;
        const uint32_t rr = range / tot;
        const uint32_t code = (uint32_t)(next() % ((unsigned long long)range + ((i & 7u) == 0 ? 1u : 0u)));   // < range (sometimes == range at the start)
        if (udiv_small_quotient(code, rr) != code / rr) miss++;
        const uint32_t code2 = (i & 3u) == 0 ? range - 1u : code;               // the largest quotients
        if (udiv_small_quotient(code2, rr) != code2 / rr) miss++;
    }
    if (miss) atomicAdd(bad, miss);
}
}  // namespace hga
extern "C" long hg_debug_udiv_check(hg_ctx *ctx, unsigned long long seed, uint32_t rounds) {
    if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return -1;
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess || hipMemset(d, 0, 8) != hipSuccess) return -1;
    hipLaunchKernelGGL(hga::udiv_check_kernel, dim3(1024), dim3(256), 0, ctx->stream, seed, rounds, d);
    const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? (long)h : -1;
}

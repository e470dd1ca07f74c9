// rans4x16_big.hip -- CRAM 3.1 rANS Nx16, the FOUR-way variant ("rANS 4x16", north_star), for streams long enough that the decode CHAIN is all that
// matters (>= hgn::BIG4_MIN plain bytes; the small ones -- tok3's thousands of token streams -- stay with ransnx16_decode_kernel<4>, sixteen per
// wavefront).  Replaces rans_uncompress_4x16() as called by cram_uncompress_block (reference cram/cram_io.c:1697-1714; implementation = htscodecs
// rANS_static4x16pr.c, an ABSENT submodule).  Format per oracle/ransnx16_oracle.c -- PARITY UNPINNED; bit-exact with that oracle.
//
// A 4-way stream is ONE dependency chain of n / 4 steps; a launch lasts as long as its longest chain.  What a step costs is the number of DEPENDENT
// instructions and memory round trips between one rANS state and the next, so the kernel is shaped around that chain:
//   * ONE stream per wavefront (lanes 0..3 carry the four states): nothing runs in lock step with a neighbour, and everything that is the same for the four
//     states -- the stream position, the next renormalisation words -- is wave-uniform and lives on the SCALAR side;
//   * the next 8 renormalisation words sit in a 128-bit scalar window (the idea of the inflate kernel's bit buffer): a lane that renormalises picks its
//     word with one 64-bit shift, no memory access; the window is topped up four words at a time from a load issued one refill earlier;
//   * order 0: ONE LDS read per symbol -- a 4096-entry table indexed by the slot gives symbol | frequency - 1 | slot - cumulative;
//   * order 1: the LDS forms of the 32-way kernel (dense two-read form for <= 16 contexts of <= 16 symbols, bucket + list otherwise, global lists when
//     nothing fits), built by all 64 lanes;
//   * order 1 output: four bytes gathered per lane, one dword store every fourth step.
// Handled: flags ORDER, NOSZ (size from the descriptor).  CAT / X32 / small streams are left to ransnx16.hip; PACK / RLE / STRIPE are undone by
// ransnx16_xform.hip after this kernel (pre-parsed descriptors, desc.reserved bit 31).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include "ransnx16_dev.h"

namespace hgn {

constexpr int BW = 4;                                              // wavefronts (= streams) per workgroup
constexpr uint32_t BPOOL = 4352;                                   // words of table pool per stream: 68 KiB per workgroup, two workgroups per CU

typedef const uint32_t __attribute__((address_space(4))) *sptr_t;     // constant address space: a uniform address makes the load a scalar one (s_load)

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32); }

// The renormalisation words of one stream, all wave-uniform (scalar registers): q = the next nv (4..8) words, f = the four after them, fp = where the words
// after those are.  The stream is read with SCALAR loads of aligned dwords (most of them hit the scalar cache: eight refills per 64-byte line): the step's
// global stores and the stream's loads then count on different counters -- a vector load issued behind the byte stores of the decoded symbols would
// wait for them (one vmcnt on gfx9).
struct Window {
    uint64_t lo, hi, f; const uint8_t *fp, *first, *end; uint32_t wo;                      // the next words: lo | hi << 64 from word wo (0..3) on; f = the four after hi
    // the 8 stream bytes at p (p uniform).  Only aligned dwords that hold at least one byte of the stream are touched, so nothing is read beyond the
    // dword of the stream's last byte; bytes past the end come out as anything (overrun() notices a stream that needs them).
    __device__ __forceinline__ uint64_t fetch(const uint8_t *p) const {
        const uintptr_t a = (uintptr_t)p & ~(uintptr_t)3, last = ((uintptr_t)end - 1u) & ~(uintptr_t)3;
        const sptr_t s = (sptr_t)a;
        uint32_t d0 = 0, d1 = 0, d2 = 0;
        if (a + 8u <= last) { d0 = s[0]; d1 = s[1]; d2 = s[2]; }
        else { if (a <= last) d0 = s[0]; if (a + 4u <= last) d1 = s[1]; }
        const uint32_t sh = ((uint32_t)(uintptr_t)p & 3u) * 8u;
        const uint64_t lo = (uint64_t)d0 | ((uint64_t)d1 << 32);
        return sh ? (lo >> sh) | ((uint64_t)d2 << (64u - sh)) : lo;
    }
    __device__ __forceinline__ void start(const uint8_t *cp, const uint8_t *e) {
        end = e; first = cp; wo = 0;
        lo = fetch(cp); hi = fetch(cp + 8); f = fetch(cp + 16); fp = cp + 24;
    }
    // the four words the lanes of this step may take (lane order: the k-th lane that renormalises takes word k)
    __device__ __forceinline__ uint64_t cur() const { const uint32_t s = 16u * wo; return (lo >> s) | ((hi << 1) << (63u - s)); }
    // the lanes named in b (a ballot) have taken one word each
    __device__ __forceinline__ void advance(unsigned long long b) {
        wo += (uint32_t)__popcll(b);
        if (wo >= 4u) { lo = hi; hi = f; wo -= 4u; f = fetch(fp); fp += 8; }
    }
    __device__ __forceinline__ bool overrun() const { return (uint64_t)(fp - first) / 2u - 12u + wo > (uint64_t)(end - first) / 2u; }   // words taken > words there
};

// one renormalisation round of the (up to four) enabled lanes
__device__ __forceinline__ void renorm(uint32_t &R, Window &W) {
    const bool need = R < RANS_L;
    const unsigned long long b = __ballot(need);
    if (need) R = (R << 16) | ((uint32_t)(W.cur() >> (16u * __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))) & 0xffffu);
    W.advance(b);
}

// O1_DENSE1 (ransnx16_dev.h): one symbol of one state.  seen collects the table words (bit 30: a context / slot the encoder never produced)
__device__ __forceinline__ uint32_t dense1_step(const uint32_t *W1, const uint32_t *E, uint32_t &r, uint32_t &R, uint32_t mask, uint32_t shift, uint32_t &seen) {
    const uint32_t m = R & mask;
    const uint32_t w = W1[r * 256u + (m >> (shift - 8u))];
    seen |= w;
    uint32_t kb = (w >> 25) & 15u, cum = w & 0x1fffu, f = ((w >> 13) & 0xfffu) + 1u;
    const uint32_t *row = E + r * 17u;
    if (!(w & (1u << 29))) {                                         // a bucket that straddles symbols: walk the row
        uint32_t c1 = row[kb + 1] >> 16;
        while (c1 <= m && kb < 15u) { kb++; c1 = row[kb + 1] >> 16; }
        cum = row[kb] >> 16; f = c1 - cum;
    }
    const uint32_t e = row[kb];
    r = (e >> 8) & 31u;
    R = __umul24(f, R >> shift) + m - cum;                          // f <= 4096, R >> shift < 2^22
    return e & 0xffu;
}

// the other forms, FORM a compile-time constant so that lookup_o1's dispatch folds away
template <int FORM>
__device__ __forceinline__ uint32_t generic_step(const O1Forms &Fm, const uint32_t *P, const uint32_t *tabs, uint32_t &ctx, uint32_t &rctx, uint32_t &R, uint32_t mask,
                                                 uint32_t shift, uint32_t &seen) {
    O1Forms G = Fm; G.form = FORM;
    const uint32_t m = R & mask;
    uint32_t sym = 0, cum = 0, f = 1;
    if (!lookup_o1(G, P, tabs, ctx, rctx, m, shift, sym, cum, f)) { seen |= 1u << 30; return 0; }     // the state stays where it is: every later step fails alike
    ctx = sym;
    R = __umul24(f, R >> shift) + m - cum;
    return sym;
}

// Order 1, the four states (lanes 0..3 enabled): state s decodes out[s * per ..], the last one the remainder as well.  Four symbols are gathered per lane
// and stored as one dword.
template <int FORM>
__device__ __forceinline__ void o1_loop(const O1Forms &Fm, const uint32_t *P, const uint32_t *tabs, uint32_t shift, uint32_t per, uint32_t rem, uint8_t *o, int lane,
                                        uint32_t &R, Window &W, uint32_t &seen) {
    const uint32_t mask = (1u << shift) - 1u;
    const uint32_t *W1 = P + Fm.l_off, *E = P + Fm.d_off;
    uint32_t ctx = 0, rctx = Fm.drank0, acc = 0;
    uint8_t *op = o + (size_t)lane * per;
    auto step = [&]() -> uint32_t { return FORM == O1_DENSE1 ? dense1_step(W1, E, rctx, R, mask, shift, seen) : generic_step<FORM>(Fm, P, tabs, ctx, rctx, R, mask, shift, seen); };
    uint32_t it = 0;
    for (; it + 4 <= per; it += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) { acc = (acc >> 8) | (step() << 24); renorm(R, W); }
        __builtin_memcpy(op + it, &acc, 4);
    }
    for (; it < per; it++) { op[it] = (uint8_t)step(); renorm(R, W); }
    if (lane == 3) for (; it < per + rem; it++) { op[it] = (uint8_t)step(); renorm(R, W); }
}

__global__ __launch_bounds__(BW * 64)
void rans4x16_big_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint32_t *__restrict__ sel, uint32_t nsel,
                                uint8_t *out, int32_t *status, uint32_t *scratch) {
    __shared__ uint32_t pool[BW][BPOOL];
    __shared__ uint16_t Cs[BW][258];
    __shared__ uint8_t rank_s[BW][256];
    const int lane = threadIdx.x & 63, wv = (int)uni(threadIdx.x >> 6);
    uint32_t *P = pool[wv]; uint16_t *C = Cs[wv]; uint8_t *rk = rank_s[wv];
    const uint32_t w_global = blockIdx.x * BW + (uint32_t)wv, w_total = gridDim.x * BW;
    for (uint32_t k = w_global; k < nsel; k += w_total) {
        const uint32_t sidx = sel[k];
        const hg_stream_desc d = desc[sidx];
        if (!big4_takes(in, d)) continue;                            // wave-uniform
        const uint8_t *const base = in + d.in_off, *const end = base + d.in_len;
        uint8_t *o = out + d.out_off;
        uint32_t *tabs = scratch + d.scratch_off;
        int err = 0;
        uint32_t flags, usz = d.out_len, shift = 12, np_words = 0, hdr = 0;
        if (d.reserved & 0x80000000u) flags = d.reserved & (F_ORDER | F_X32 | F_CAT);
        else {
            const uint8_t *cp = base;
            flags = *cp++;
            if (!(flags & F_NOSZ)) { uint32_t v = 0; if (get_u7(cp, end, v) || v != usz) err = 1; }
            if (!err && (flags & (F_STRIPE | F_RLE | F_PACK))) err = 3;
            hdr = (uint32_t)(cp - base);
        }
        const uint32_t order = flags & F_ORDER;
        // ---- tables: lane 0 parses (the compressed form of an order-1 table is itself a serial rANS stream), all lanes build the LDS forms
        if (!err && lane == 0) {
            const uint8_t *cp = base + hdr;
            if (order == 0) err = parse_o0(cp, end, C);
            else err = parse_o1(cp, end, tabs, C, rk, shift, np_words);
            hdr = (uint32_t)(cp - base);
        }
        err = (int)uni((uint32_t)err); shift = uni(shift); np_words = uni(np_words); hdr = uni(hdr);     // lane 0 is the first active lane
        hg::wave_sync();
        O1Forms Fm;
        Fm.form = O1_LISTS_GLOBAL; Fm.bb = 0; Fm.l_off = 0; Fm.d_off = 0; Fm.drank0 = 0;
        if (!err && order) build_o1_forms<64, true>(P, BPOOL, tabs, np_words, shift, rk, lane, ~0ull, 0, Fm);
        else if (!err) {
            // order 0: slot -> symbol | (frequency - 1) << 8 | (slot - cumulative) << 20, one read per symbol
            for (uint32_t sy = (uint32_t)lane; sy < 256; sy += 64) {
                const uint32_t a = C[sy], b = C[sy + 1];
                for (uint32_t q = a; q < b; q++) P[q] = sy | ((b - a - 1u) << 8) | ((q - a) << 20);
            }
        }
        hg::wave_sync();
        const uint8_t *cp = base + hdr;
        if (!err && cp + 16 > end) err = 1;
        const uint32_t per = usz >> 2, rem = usz & 3u;
        int bad = 0;                                                  // per lane: a context / slot the encoder never produced
        bool over = false;
        if (!err && lane < 4) {                                       // the four states; everything below runs with these four lanes enabled
            uint32_t R = rd32(cp + 4 * lane);
            Window W;
            W.start(cp + 16, end);
            if (order == 0) {
                uint32_t pos = (uint32_t)lane;
#pragma unroll 2
                for (uint32_t it = 0; it < per; it++) {
                    const uint32_t e = P[R & 4095u];
                    o[pos] = (uint8_t)e; pos += 4;
                    R = (((e >> 8) & 4095u) + 1u) * (R >> 12) + (e >> 20);
                    renorm(R, W);
                }
                if ((uint32_t)lane < rem) o[per * 4u + (uint32_t)lane] = (uint8_t)P[R & 4095u];     // states 0..rem-1 give one more symbol each, without update
            } else {
                uint32_t seen = 0;
                switch (uni(Fm.form)) {                                // wave-uniform
                case O1_DENSE1: o1_loop<O1_DENSE1>(Fm, P, tabs, shift, per, rem, o, lane, R, W, seen); break;
                case O1_DENSE: o1_loop<O1_DENSE>(Fm, P, tabs, shift, per, rem, o, lane, R, W, seen); break;
                case O1_BUCKET: o1_loop<O1_BUCKET>(Fm, P, tabs, shift, per, rem, o, lane, R, W, seen); break;
                case O1_LISTS_LDS: o1_loop<O1_LISTS_LDS>(Fm, P, tabs, shift, per, rem, o, lane, R, W, seen); break;
                default: o1_loop<O1_LISTS_GLOBAL>(Fm, P, tabs, shift, per, rem, o, lane, R, W, seen); break;
                }
                bad = (seen >> 30) & 1u;
            }
            over = W.overrun();
        }
        if (__ballot(bad != 0 || over)) err = 1;
        if (lane == 0) status[sidx] = err == 0 ? 0 : (err == 3 ? HG_BLOCK_EUNSUPPORTED : -1);
        hg::wave_sync();
    }
}

}  // namespace hgn

namespace hg {
int launch_rans4x16_big_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4, size_t n4, void *d_out, int32_t *d_status,
                               uint32_t *d_scratch, hipStream_t s, bool leave_room) {
    if (!n4) return HG_OK;
    size_t wgs = (n4 + hgn::BW - 1) / hgn::BW;
    // two workgroups per CU are resident (72 KiB of LDS each); with a 32-way launch beside it (68 KiB workgroups) one per CU, or the two kernels take turns
    const size_t maxw = (size_t)ctx->cus * (leave_room ? 1 : 2);
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgn::rans4x16_big_decode_kernel, dim3((unsigned)wgs), dim3(hgn::BW * 64), 0, s, (const uint8_t *)d_in, d_desc, d_sel4, (uint32_t)n4,
                       (uint8_t *)d_out, d_status, d_scratch);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

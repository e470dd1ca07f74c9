// arith_dev.h -- device-side range coder and adaptive model shared by the CRAM 3.1 adaptive arithmetic coder (arith.hip, block
// method 6) and the fqzcomp quality codec (fqzcomp.hip, block method 7): the wave-cooperative restatement of htscodecs'
// c_range_coder.h / c_simple_model.h (absent submodule; arithmetic per oracle/range_model.h -- PARITY UNPINNED).
//
// One wavefront works on one stream.  The range-coder registers are wave-uniform; a model is an array of (freq << 8 | symbol)
// words kept sorted by frequency, searched by all 64 lanes at once (one read of up to 4 x 64 entries, a DPP prefix sum, one
// ballot); the compressed bytes are read 64 at a time into one VGPR and picked out with v_readlane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_device.h"

namespace hga {

// -DHG_ARITH_PROFILE: per-symbol phase times of the decoder (core-clock ticks), printed by the first wavefront per stream
#ifdef HG_ARITH_PROFILE
#define HA_T(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); g_tacc[slot] += n_ - g_tlast; g_tlast = n_; } while (0)
__device__ unsigned long long g_dummy;
#define HA_DECL unsigned long long g_tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g_tlast = __builtin_amdgcn_s_memtime()
#else
#define HA_T(slot) do { } while (0)
#endif
using hg::wave_sync;
using hg::wave_incl_scan_dpp;

constexpr uint32_t STEP = 16, MAX_FREQ = (1u << 16) - 17u, TOP = 1u << 24;
enum { F_ORDER = 1, F_RLE = 64 };

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// The two divisions of a coder step (range / total, code / r) sit on every symbol's chain, and the compiler's 32-bit division is ~22 instructions.
// Both have structure: (1) a quotient below 2^17 -- one single-precision estimate is within 1 of it, one correction either way;
__device__ __forceinline__ uint32_t udiv_small_quotient(uint32_t a, uint32_t b) {
    // (the sign test below reads a remainder of up to 2b - 1 as an int32: b < 2^30.  Larger divisors only come from models whose total is 1 .. 3 -- the
    // first symbols of an alphabet of one to three symbols -- and take the plain division; round 4: an all-zero block, alphabet {0}, failed to decode)
    if (__builtin_expect(b >= (1u << 30), 0)) return a / b;
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b));
    const uint32_t rem = a - q * b;                                  // (mod 2^32) -b <= rem < 2b
    if ((int32_t)rem < 0) q--; else if (rem >= b) q++;
    return q;
}
// (2) a divisor below 2^16 (a model's total): an estimate scaled to stay BELOW the quotient, the remainder's own estimate on top, one correction.
__device__ __forceinline__ uint32_t udiv_small_divisor(uint32_t a, uint32_t b) {
    const float rb = __builtin_amdgcn_rcpf((float)b) * 0.9999995f;
    const uint32_t q1 = (uint32_t)((float)a * rb);                  // <= a / b, short by a relative 2^-20 at most
    const uint32_t r1 = a - q1 * b;                                  // < 2^12 * b < 2^28
    uint32_t q = q1 + (uint32_t)((float)r1 * rb);
    if (a - q * b >= b) q++;
    return q;
}

// 64-byte window over the compressed stream: lane l holds byte pos0 + l
struct ByteWindow {
    const uint8_t *base; uint32_t len, pos0, idx, win; bool overrun;
    __device__ void init(const uint8_t *b, uint32_t n, int lane) { base = b; len = n; pos0 = 0; idx = 0; overrun = false; load(lane); }
    // (the wait belongs HERE, once per 64 bytes: left to the compiler, every later use of `win` inside the symbol loop gets a conservative s_waitcnt vmcnt(0)
    //  -- "it may have been loaded on the way round" -- which also waits for the model stores of the symbol before: a store round trip per symbol)
    __device__ void load(int lane) { const uint32_t p = pos0 + (uint32_t)lane; win = p < len ? base[p] : 0u; hg::wait_vm0(); }
    __device__ uint32_t next(int lane) {
        if (idx == 64) { pos0 += 64; idx = 0; load(lane); }
        if (pos0 + idx >= len) overrun = true;
        return rl(win, idx++);
    }
};

// After the coder step: bump entry x of the model at B (n entries, total at T), halve when due, keep sorted.
// f_x / f_prev are the frequencies of entries x and x-1 as read before the bump.
// e_prev = entry x - 1 as the symbol search saw it (HAVE_PREV) -- it came in the same 64-entry read, so the update needs no load.
// LDSM: the model is in LDS, which serves a wavefront's accesses in order -- the next symbol's read sees these writes without a fence.
template <bool LDSM = false>
__device__ __forceinline__ void model_update(uint32_t *M, uint32_t *TT, uint32_t B, uint32_t n, uint32_t T, uint32_t tot, uint32_t x,
                                             uint32_t e_x, bool have_prev, uint32_t e_prev, int lane) {
    uint32_t ex = e_x + (STEP << 8);
    tot += STEP;
    if (tot > MAX_FREQ) {                                            // halve every frequency (rare)
        if (lane == 0) M[B + x] = ex;
        wave_sync();
        uint32_t sum = 0;
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t i = b + (uint32_t)lane;
            uint32_t f = 0;
            if (i < n) { const uint32_t e = M[B + i]; f = e >> 8; f -= f >> 1; M[B + i] = (f << 8) | (e & 0xffu); }
            sum += rl(wave_incl_scan_dpp(f), 63);
        }
        tot = sum;
        wave_sync();
        ex = M[B + x];
        have_prev = false;
    }
    if (x > 0) {
        const uint32_t ep = have_prev ? e_prev : M[B + x - 1];
        if ((ex >> 8) > (ep >> 8)) { if (lane == 0) { M[B + x] = ep; M[B + x - 1] = ex; } }
        else if (lane == 0) M[B + x] = ex;
    } else if (lane == 0) M[B + x] = ex;
    if (lane == 0) TT[T] = tot;
    if (!LDSM) wave_sync();
}

struct Decoder {
    uint32_t code, range; ByteWindow in; int err;
#ifdef HG_ARITH_PROFILE
    unsigned long long g_tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g_tlast = 0;
#endif
    __device__ void start(const uint8_t *b, uint32_t n, int lane) {
#ifdef HG_ARITH_PROFILE
        g_tlast = __builtin_amdgcn_s_memtime();
#endif
        in.init(b, n, lane); err = 0; code = 0; range = 0xffffffffu;
        for (int i = 0; i < 5; i++) code = (code << 8) | in.next(lane);
    }
    // The LEAN step: models are kept sorted by frequency, so the symbol nearly always sits among the first 64 entries -- ONE 64-lane read of those, the two
    // short divisions above, one scan + ballot, and an update in which the (one or two) lanes whose entry changes write their own word.  ~70 instructions
    // per symbol against ~250 of the general routine below, which takes over when the symbol lies beyond entry 63 of a larger model (it starts again
    // from the untouched coder state) and for the halving of a model.  LDSM: the model is in LDS (no fence needed between the update and the next read).
    template <bool LDSM>
    __device__ __forceinline__ uint32_t symbol_lean(uint32_t *M, uint32_t *TT, uint32_t B, uint32_t n, uint32_t T, int lane) {
        const uint32_t e = (uint32_t)lane < n ? M[B + (uint32_t)lane] : 0u;
        uint32_t tot = hg::uni(TT[T]);
        const uint32_t r = udiv_small_divisor(range, tot), freq = udiv_small_quotient(code, r);
        if (freq >= tot) { err = 1; return 0; }
        const uint32_t incl = wave_incl_scan_dpp(e >> 8);            // lanes >= n carry the total: with n <= 64 the first lane above freq is inside the model
        const unsigned long long hit = __ballot(incl > freq);
        if (n > 64u && !hit) return symbol<LDSM>(M, TT, B, n, T, lane);
        const uint32_t l = (uint32_t)__builtin_ctzll(hit);
        const uint32_t ex = rl(e, l), f = ex >> 8;
        code -= (rl(incl, l) - f) * r; range = r * f;
        while (range < TOP) { code = (code << 8) | in.next(lane); range <<= 8; }
        tot += STEP;
        if (tot > MAX_FREQ) {                                          // halve every frequency (rare): the general routine
            model_update<LDSM>(M, TT, B, n, T, tot - STEP, l, ex, false, 0u, lane);
            return ex & 0xffu;
        }
        const uint32_t nex = ex + (STEP << 8), ep = l ? rl(e, l - 1u) : 0xffffffffu;
        const bool swap = (nex >> 8) > (ep >> 8);                     // keep the list sorted by frequency: at most one step towards the front
        if ((uint32_t)lane == l) M[B + l] = swap ? ep : nex;
        if (swap && (uint32_t)lane + 1u == l) M[B + l - 1u] = nex;
        if (lane == 0) TT[T] = tot;
        if (!LDSM) wave_sync();
        return ex & 0xffu;
    }
    // decodes one symbol with the model at B (n entries, total at T): the general routine (any n <= 256, the symbol anywhere in the list)
    template <bool LDSM = false>
    __device__ uint32_t symbol(uint32_t *M, uint32_t *TT, uint32_t B, uint32_t n, uint32_t T, int lane) {
        HA_T(5);
        // all (<= 4) 64-entry pieces of the model are requested at once, together with the total: the search below then runs
        // on registers -- with the pieces fetched one per probe a 256-symbol model in global memory cost up to 4 round trips
        uint32_t ev[4];
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) { const uint32_t i = c * 64u + (uint32_t)lane; ev[c] = (c * 64u < n && i < n) ? M[B + i] : 0u; }
        const uint32_t tot = TT[T];
        HA_T(0);
        const uint32_t r = udiv_small_divisor(range, tot), freq = udiv_small_quotient(code, r);
        if (freq >= tot) { err = 1; return 0; }
        HA_T(1);
        uint32_t acc0 = 0, x = 0, ex = 0, acc = 0, eprev = 0;
        bool have_prev = false;
#pragma unroll
        for (uint32_t b = 0; b < 256u; b += 64) {
            if (b >= n) break;
            const uint32_t i = b + (uint32_t)lane;
            const uint32_t e = ev[b >> 6];
            const uint32_t incl = acc0 + wave_incl_scan_dpp(e >> 8);
            const unsigned long long hit = __ballot(i < n && incl > freq);
            if (hit) {
                const uint32_t l = (uint32_t)__builtin_ctzll(hit);
                x = b + l; ex = rl(e, l); acc = rl(incl, l) - (ex >> 8);
                if (l) { have_prev = true; eprev = rl(e, l - 1u); }
                break;
            }
            acc0 = rl(incl, 63);
        }
        HA_T(2);
        const uint32_t f = ex >> 8;
        code -= acc * r; range = r * f;
        while (range < TOP) { code = (code << 8) | in.next(lane); range <<= 8; }
        HA_T(3);
        model_update<LDSM>(M, TT, B, n, T, tot, x, ex, have_prev, eprev, lane);
        HA_T(4);
        return ex & 0xffu;
    }
};


// ---- order 0 with a WIDE alphabet (65..256 symbols): one model, met by every symbol of the stream ---------------------------------------------------
// The lean step above looks at the first 64 entries only; data whose bytes are spread evenly (the byte planes of 32-bit integers) finds its symbol
// there one time in four and falls back to the general routine -- four reads, up to four scans.  With a single model per stream the bookkeeping that
// makes the step independent of WHERE the symbol sits fits in scalar registers: the totals of the four 64-entry pieces (sum[]) select the piece -- from
// the cumulative frequency when decoding, from a symbol -> position byte map in LDS when encoding -- and ONE read + ONE scan of that piece do the rest.
// The list itself is unchanged (sorted by frequency, one step towards the front per hit, halved when the total passes MAX_FREQ): same bytes as before.
struct WideO0 {
    uint32_t *M; uint8_t *pos; uint32_t n, tot, sum[4];
    __device__ __forceinline__ void init(uint32_t *M_, uint8_t *pos_, uint32_t n_, bool want_pos, int lane) {
        M = M_; pos = pos_; n = n_; tot = n_;
        for (uint32_t p = 0; p < 4; p++) sum[p] = n_ > 64u * p ? (n_ - 64u * p < 64u ? n_ - 64u * p : 64u) : 0u;     // every frequency starts at 1
        if (want_pos) for (uint32_t i = (uint32_t)lane; i < 256u; i += 64) pos_[i] = (uint8_t)i;                     // models_init: entry i holds symbol i
        wave_sync();
    }
    __device__ __forceinline__ uint32_t base(uint32_t p) const { return p == 0 ? 0u : p == 1 ? sum[0] : p == 2 ? sum[0] + sum[1] : sum[0] + sum[1] + sum[2]; }
    __device__ __forceinline__ uint32_t piece(uint32_t p, int lane) const { const uint32_t i = 64u * p + (uint32_t)lane; return i < n ? M[i] : 0u; }
    // entry l of piece p (word ex, as read) has been coded: bump it, keep the list sorted, keep sum[] / pos[] / tot in step
    template <bool POS>
    __device__ __forceinline__ void bump(uint32_t p, uint32_t l, uint32_t ex, uint32_t e, int lane) {
        uint32_t nex = ex + (STEP << 8);
        tot += STEP; sum[p] += STEP;
        if (tot > MAX_FREQ) {                                          // halve every frequency (rare)
            if ((uint32_t)lane == l) M[64u * p + l] = nex;
            uint32_t t = 0;
            for (uint32_t q = 0; q < 4 && 64u * q < n; q++) {
                const uint32_t i = 64u * q + (uint32_t)lane;
                uint32_t f = 0;
                if (i < n) { const uint32_t w = M[i]; f = w >> 8; f -= f >> 1; M[i] = (f << 8) | (w & 0xffu); }
                sum[q] = rl(wave_incl_scan_dpp(f), 63); t += sum[q];
            }
            tot = t;
            const uint32_t w = piece(p, lane);
            nex = rl(w, l); e = w;
        }
        const uint32_t x = 64u * p + l;
        if (x == 0) { if (lane == 0 && !(tot > MAX_FREQ)) M[0] = nex; return; }
        const uint32_t ep = l ? rl(e, l - 1u) : hg::uni(M[x - 1u]);   // the neighbour towards the front (the last entry of the piece before, when l = 0)
        if ((nex >> 8) > (ep >> 8)) {
            if (lane == 0) { M[x] = ep; M[x - 1u] = nex; if (POS) { pos[nex & 0xffu] = (uint8_t)(x - 1u); pos[ep & 0xffu] = (uint8_t)x; } }
            if (l == 0) { const uint32_t d = (nex >> 8) - (ep >> 8); sum[p] -= d; sum[p - 1u] += d; }     // the entries changed pieces
        } else if (lane == 0) M[x] = nex;
    }
};

struct Encoder {
    uint32_t low, range, carry, cache, ffnum;
    uint8_t *out; uint32_t opos, oidx, obuf;                         // 64 output bytes are gathered in one VGPR
    __device__ void start(uint8_t *o) { low = 0; range = 0xffffffffu; carry = 0; cache = 0; ffnum = 0; out = o; opos = 0; oidx = 0; obuf = 0; }
    __device__ void put(uint32_t b, int lane) {
        obuf = hg::writelane(b & 0xffu, oidx, obuf);
        if (++oidx == 64) { out[opos + (uint32_t)lane] = (uint8_t)obuf; opos += 64; oidx = 0; }
    }
    __device__ void shift_low(int lane) {
        if (low < 0xff000000u || carry) {
            put(cache + carry, lane);
            while (ffnum) { put(carry - 1u, lane); ffnum--; }
            cache = low >> 24; carry = 0;
        } else ffnum++;
        low <<= 8;
    }
    __device__ void encode(uint32_t cum, uint32_t freq, uint32_t tot, int lane) {
        const uint32_t old = low;
        range = udiv_small_divisor(range, tot);                        // tot <= MAX_FREQ < 2^16
        low += cum * range;
        range *= freq;
        if (low < old) carry = 1;
        while (range < TOP) { range <<= 8; shift_low(lane); }
    }
    __device__ uint32_t finish(int lane) {
        for (int i = 0; i < 5; i++) shift_low(lane);
        if ((uint32_t)lane < oidx) out[opos + (uint32_t)lane] = (uint8_t)obuf;
        return opos + oidx;
    }
    // the encoder's twin of Decoder::symbol_lean: the symbol is looked for among the first 64 entries; beyond them (larger models only) the general routine
    template <bool LDSM>
    __device__ __forceinline__ void symbol_lean(uint32_t *M, uint32_t *TT, uint32_t B, uint32_t n, uint32_t T, uint32_t sym, int lane) {
        const bool mine = (uint32_t)lane < n;
        const uint32_t e = mine ? M[B + (uint32_t)lane] : 0u;
        uint32_t tot = hg::uni(TT[T]);
        const unsigned long long hit = __ballot(mine && (e & 0xffu) == sym);
        if (!hit) { symbol<LDSM>(M, TT, B, n, T, sym, lane); return; }    // (n > 64: the symbol lies further back)
        const uint32_t incl = wave_incl_scan_dpp(e >> 8);
        const uint32_t l = (uint32_t)__builtin_ctzll(hit);
        const uint32_t ex = rl(e, l), f = ex >> 8;
        {                                                                  // encode(cum, f, tot)
            const uint32_t old = low;
            range = udiv_small_divisor(range, tot);
            low += (rl(incl, l) - f) * range;
            range *= f;
            if (low < old) carry = 1;
            while (range < TOP) { range <<= 8; shift_low(lane); }
        }
        tot += STEP;
        if (tot > MAX_FREQ) { model_update<LDSM>(M, TT, B, n, T, tot - STEP, l, ex, false, 0u, lane); return; }
        const uint32_t nex = ex + (STEP << 8), ep = l ? rl(e, l - 1u) : 0xffffffffu;
        const bool swap = (nex >> 8) > (ep >> 8);
        if ((uint32_t)lane == l) M[B + l] = swap ? ep : nex;
        if (swap && (uint32_t)lane + 1u == l) M[B + l - 1u] = nex;
        if (lane == 0) TT[T] = tot;
        if (!LDSM) wave_sync();
    }
    // codes `sym` with the model at B (n entries, total at T)
    template <bool LDSM = false>
    __device__ void symbol(uint32_t *M, uint32_t *TT, uint32_t B, uint32_t n, uint32_t T, uint32_t sym, int lane) {
        uint32_t ev[4];                                                    // the whole model at once (see the decoder)
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) { const uint32_t i = c * 64u + (uint32_t)lane; ev[c] = (c * 64u < n && i < n) ? M[B + i] : 0u; }
        const uint32_t tot = TT[T];
        uint32_t acc0 = 0, x = 0, ex = 0, acc = 0, eprev = 0;
        bool have_prev = false;
#pragma unroll
        for (uint32_t b = 0; b < 256u; b += 64) {
            if (b >= n) break;
            const uint32_t i = b + (uint32_t)lane;
            const uint32_t e = ev[b >> 6];
            const uint32_t incl = acc0 + wave_incl_scan_dpp(e >> 8);
            const unsigned long long hit = __ballot(i < n && (e & 0xffu) == sym);
            if (hit) {
                const uint32_t l = (uint32_t)__builtin_ctzll(hit);
                x = b + l; ex = rl(e, l); acc = rl(incl, l) - (ex >> 8);
                if (l) { have_prev = true; eprev = rl(e, l - 1u); }
                break;
            }
            acc0 = rl(incl, 63);
        }
        encode(acc, ex >> 8, tot, lane);
        model_update<LDSM>(M, TT, B, n, T, tot, x, ex, have_prev, eprev, lane);
    }
};


// one symbol through a WideO0 model (see there)
__device__ __forceinline__ uint32_t wide_decode(Decoder &D, WideO0 &W, int lane) {
    const uint32_t r = udiv_small_divisor(D.range, W.tot), freq = udiv_small_quotient(D.code, r);
    if (freq >= W.tot) { D.err = 1; return 0; }
    const uint32_t a = W.sum[0], b = a + W.sum[1], c = b + W.sum[2];
    const uint32_t p = (freq >= a ? 1u : 0u) + (freq >= b ? 1u : 0u) + (freq >= c ? 1u : 0u);
    const uint32_t e = W.piece(p, lane);
    const uint32_t incl = W.base(p) + wave_incl_scan_dpp(e >> 8);   // lanes past the model's end carry the total
    const uint32_t l = (uint32_t)__builtin_ctzll(__ballot(incl > freq));
    const uint32_t ex = rl(e, l), f = ex >> 8;
    D.code -= (rl(incl, l) - f) * r; D.range = r * f;
    while (D.range < TOP) { D.code = (D.code << 8) | D.in.next(lane); D.range <<= 8; }
    W.bump<false>(p, l, ex, e, lane);
    return ex & 0xffu;
}
__device__ __forceinline__ void wide_encode(Encoder &E, WideO0 &W, uint32_t sym, int lane) {
    const uint32_t x = hg::uni((uint32_t)W.pos[sym]), p = x >> 6, l = x & 63u;
    const uint32_t e = W.piece(p, lane);
    const uint32_t incl = wave_incl_scan_dpp(e >> 8);
    const uint32_t ex = rl(e, l), f = ex >> 8;
    E.encode(W.base(p) + rl(incl, l) - f, f, W.tot, lane);
    W.bump<true>(p, l, ex, e, lane);
}

}  // namespace hga

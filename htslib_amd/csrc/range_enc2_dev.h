// range_enc2_dev.h -- device pieces shared by the two-phase ENCODERS of the adaptive range coder (arith_enc2.hip, CRAM method 6) and of fqzcomp
// (fqzcomp.hip, method 7): the lane ranking used by the event sorts, the register model of phase A and the scalar coder of phase B.  Format and arithmetic per
// oracle/range_model.h -- PARITY UNPINNED (htscodecs absent); byte-identical to the one-pass kernels and to the oracles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_device.h"
#include "arith_dev.h"

namespace hga2 {
using namespace hga;

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// the lanes of `act` that hold the same BITS-bit key as this lane (valid on act lanes): one ballot per key bit
template <int BITS>
__device__ __forceinline__ unsigned long long match_bits(uint32_t key, bool act) {
    unsigned long long m = __ballot(act);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const bool bit = (key >> b) & 1u;
        const unsigned long long bal = __ballot(act && bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}
__device__ __forceinline__ unsigned long long match8(uint32_t key, bool act) { return match_bits<8>(key, act); }

// ---- a model of at most 64 symbols in registers: lane l holds entry l ((freq << 8) | symbol, sorted by frequency like every model of this coder) and the
//      inclusive prefix sum of the frequencies up to it.  A step stays on the VECTOR side: the lane that holds the symbol writes the event's record itself
//      (one predicated store), the lanes behind it add STEP to their prefix sums (v_mbcnt of the hit mask), and the "one step towards the front" swap is
//      two DPP wave shifts -- no v_readlane / scalar round trip on the chain from one event to the next (round 4's form: ~0.3 us per event).
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }   // lane i <- lane i - 1 (lane 0: 0)
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }   // lane i <- lane i + 1 (lane 63: 0)
// EPL entries per lane: lane l holds entries EPL l .. EPL l + EPL - 1 (1: alphabets up to 64 symbols, 2: up to 128, 4: up to 256), so that neighbours inside a
// lane are register moves and only the lane boundary needs the wave shift.
template <int EPL>
struct RegModel {
    uint32_t e[EPL], incl[EPL], tot, n;
    __device__ __forceinline__ void init(uint32_t m, int lane) {
        n = m; tot = m;
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            const uint32_t i = (uint32_t)lane * EPL + (uint32_t)k;
            e[k] = i < m ? (1u << 8) | i : 0u;
            incl[k] = i < m ? i + 1u : m;
        }
    }
    // codes `sym` (wave-uniform) into record `slot`, then the update (arith_dev.h model_update: bump by STEP, halve all when the total passes MAX_FREQ, one
    // step towards the front when the entry outgrew its neighbour)
    __device__ __forceinline__ void step(uint32_t sym, uint2 *R, uint32_t slot, int lane) {
        bool hit[EPL], any = false;
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            hit[k] = (uint32_t)lane * EPL + (uint32_t)k < n && (e[k] & 0xffu) == sym;      // exactly one entry of one lane
            const uint32_t f = e[k] >> 8;
            x = hit[k] ? (incl[k] - f) | f << 16 : x;
            any = any || hit[k];
        }
        // (device-scope store: the coder that follows the models may run on another XCD, whose L2 does not see this one's dirty lines)
        if (any) __hip_atomic_store((unsigned long long *)(R + slot), (unsigned long long)x | (unsigned long long)tot << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hm = __ballot(any);
        uint32_t acc = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));   // 1 on the lanes after the hit
        uint32_t eb[EPL], ib[EPL];
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            acc |= (uint32_t)hit[k];                                       // ... and on the hit entry and the entries after it in its lane
            eb[k] = hit[k] ? e[k] + (STEP << 8) : e[k];
            ib[k] = incl[k] + (acc ? STEP : 0u);
        }
        tot += STEP;
        if (tot > MAX_FREQ) {                                            // halve every frequency (rare)
            uint32_t fr[EPL], sum = 0;
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                const bool in = (uint32_t)lane * EPL + (uint32_t)k < n;
                fr[k] = eb[k] >> 8; fr[k] -= fr[k] >> 1;
                fr[k] = in ? fr[k] : 0u;
                eb[k] = in ? (fr[k] << 8) | (eb[k] & 0xffu) : 0u;
                sum += fr[k];
            }
            const uint32_t through = wave_incl_scan_dpp(sum);
            uint32_t run = through - sum;
#pragma unroll
            for (int k = 0; k < EPL; k++) { run += fr[k]; ib[k] = run; }
            tot = rl(through, 63);
        }
        uint32_t e_left[EPL], swv[EPL];
        bool swap_here[EPL];
        e_left[0] = wave_shr1(eb[EPL - 1]);
#pragma unroll
        for (int k = 1; k < EPL; k++) e_left[k] = eb[k - 1];
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            swap_here[k] = hit[k] && (lane > 0 || k > 0) && (eb[k] >> 8) > (e_left[k] >> 8);
            swv[k] = swap_here[k] ? eb[k] : 0u;
        }
        // the entry before a swapping hit receives the bumped entry, and the sum through the pair (which the swap does not change) minus its own frequency
        const uint32_t fr_next = wave_shl1(swv[0]), ir_next = wave_shl1(ib[0]);
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            const uint32_t from_right = k + 1 < EPL ? swv[k + 1 < EPL ? k + 1 : 0] : fr_next;
            const uint32_t i_right = k + 1 < EPL ? ib[k + 1 < EPL ? k + 1 : 0] : ir_next;
            const uint32_t fk = eb[k] >> 8;
            e[k] = swap_here[k] ? e_left[k] : from_right ? from_right : eb[k];
            incl[k] = from_right ? i_right - fk : ib[k];
        }
    }
};

// ---- phase B: one wavefront per stream over dense records; the coder registers are scalars
struct Coder {
    uint32_t low, range, carry, cache, ffnum;
    uint8_t *out; uint32_t opos, oidx, obuf;                         // 64 output bytes are gathered in one VGPR
    __device__ __forceinline__ void start(uint8_t *o) { low = 0; range = 0xffffffffu; carry = 0; cache = 0; ffnum = 0; out = o; opos = 0; oidx = 0; obuf = 0; }
    __device__ __forceinline__ void put(uint32_t b, int lane) {
        obuf = hg::writelane(b & 0xffu, oidx, obuf);
        if (++oidx == 64) { out[opos + (uint32_t)lane] = (uint8_t)obuf; opos += 64; oidx = 0; }
    }
    __device__ __forceinline__ void shift_low(int lane) {
        if (low < 0xff000000u || carry) {
            put(cache + carry, lane);
            while (ffnum) { put(carry - 1u, lane); ffnum--; }
            cache = low >> 24; carry = 0;
        } else ffnum++;
        low <<= 8;
    }
    // one event: x = cum | freq << 16, t = the model's total, inv = floor((2^32 - 1) / t)
    __device__ __forceinline__ void step(uint32_t x, uint32_t t, uint32_t inv, int lane) {
        const uint32_t cum = x & 0xffffu, f = x >> 16;
        uint32_t q = __umulhi(range, inv);
        const uint32_t r = range - q * t;                               // q is short by 2 at most: r < 3 t < 2^18
        q += ((t - 1u - r) >> 31) + ((2u * t - 1u - r) >> 31);            // + (r >= t) + (r >= 2 t), as sign bits: stays on the scalar ALU
        const unsigned long long s = (unsigned long long)low + (unsigned long long)cum * q;    // cum * q < 2^32: cum < t, q = range / t
        low = (uint32_t)s; carry |= (uint32_t)(s >> 32);
        range = q * f;
        while (__builtin_expect(range < TOP, 0)) { range <<= 8; shift_low(lane); }   // (the common case falls through: a taken branch per event is dear)
    }
    __device__ __forceinline__ uint32_t finish(int lane) {
        for (int i = 0; i < 5; i++) shift_low(lane);
        if ((uint32_t)lane < oidx) out[opos + (uint32_t)lane] = (uint8_t)obuf;
        return opos + oidx;
    }
};


// one tile of dense records through the coder: rec = lane j's record (x = cum | freq << 16, y = total), nn of them valid
__device__ __forceinline__ void code_tile(Coder &E, uint2 rec, uint32_t nn, int lane) {
    const uint32_t tv = rec.y & 0xffffu;
    // floor((2^32 - 1) / total), every lane for its own record: range / total is then a multiply-high, short by 2 at most
    const uint32_t iv = tv <= 1u ? 0xffffffffu : udiv_small_divisor(0xffffffffu, tv);
    if (nn == 64u) {
#pragma unroll
        for (int j = 0; j < 64; j++) E.step(rl(rec.x, (uint32_t)j), rl(tv, (uint32_t)j), rl(iv, (uint32_t)j), lane);
    } else for (uint32_t j = 0; j < nn; j++) E.step(rl(rec.x, j), rl(tv, j), rl(iv, j), lane);
}

}  // namespace hga2

// arith_enc2.hip -- the adaptive range coder's ENCODER in two phases, for streams long enough that the serial chain of arith.hip's
// one-wavefront-per-stream kernel is what a batch waits for (reference call site cram/cram_io.c:1869-1883, arith_compress_to; format
// and arithmetic per oracle/arith_oracle.c -- PARITY UNPINNED; byte-identical to arith.hip's encoder and to that oracle).
//
// An encoder -- unlike a decoder -- knows every symbol and therefore every CONTEXT up front.  Two dependency chains are tangled in the
// one-pass coder: (1) each adaptive model's state, which only the events of THAT model touch, and (2) the coder registers (low, range),
// which only need each event's (cumulative frequency, frequency, total).  Untangled:
//   phase A  one wavefront per (stream, model): it scans the stream for the events of its model (64 positions per ballot), keeps the
//            model in REGISTERS -- lane l holds entry l, the running prefix sums are maintained incrementally (one masked add per event,
//            no scan) -- and writes one 8-byte record (cum | freq << 16, total | valid) per event at the event's place in stream order.
//            An order-1 stream has up to 256 literal models (+ 258 run models with RLE): that many chains side by side, each as long as
//            its context is frequent.  Models of more than 64 symbols use arith_dev.h's WideO0 (LDS, symbol -> position map).
//   phase B  one wavefront per stream walks the records: range / total by a multiply-high with the reciprocal the 64 lanes computed for
//            64 records at a time, low += cum * r, range = r * freq, carry / renormalisation -- scalar work, ~20 instructions per event
//            against ~130 of the tangled step.
// Records: slot i = the literal at position i (no RLE); with RLE the run that starts at position i owns slots 2i (its literal) and
// 2i + 1 ... (its run-length parts): a run of r + 1 symbols has 2 + r / 3 events and 2 r + 2 slots.  Phase B skips empty slots by ballot.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

#include "arith_dev.h"

namespace hga2 {
using namespace hga;

constexpr uint32_t REC_VALID = 0x80000000u;
// per-stream words in the model scratch (hg_stream_desc::scratch_off): the alphabet size and which byte values occur
struct Info { uint32_t m, present[8], pad[7]; };

// ---- pre-pass: one workgroup per stream -- alphabet, presence bits; RLE streams get their record slots cleared
__global__ __launch_bounds__(256)
void prepass_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel,
                    uint32_t *gscratch, uint8_t *work) {
    __shared__ uint32_t pres[8];
    const uint32_t k = sel[blockIdx.x];
    const hg_stream_desc d = desc[k];
    const uint8_t *src = in + d.in_off;
    const uint32_t n = d.in_len, rle = flags_in[k] & F_RLE;
    if (threadIdx.x < 8) pres[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto mark = [&](uint32_t c) {
#pragma unroll
        for (int w = 0; w < 8; w++) mine[w] |= (c >> 5) == (uint32_t)w ? 1u << (c & 31u) : 0u;
    };
    const uint32_t head = (uint32_t)((4u - ((uintptr_t)src & 3u)) & 3u) < n ? (uint32_t)((4u - ((uintptr_t)src & 3u)) & 3u) : n;
    const uint32_t words = (n - head) / 4u, tail0 = head + words * 4u;
    if (threadIdx.x < head) mark(src[threadIdx.x]);
    const uint32_t *sw = (const uint32_t *)(src + head);
    for (uint32_t i = threadIdx.x; i < words; i += 256) { const uint32_t w = sw[i]; mark(w & 0xffu); mark((w >> 8) & 0xffu); mark((w >> 16) & 0xffu); mark(w >> 24); }
    if (tail0 + threadIdx.x < n) mark(src[tail0 + threadIdx.x]);
#pragma unroll
    for (int w = 0; w < 8; w++) if (mine[w]) atomicOr(&pres[w], mine[w]);
    if (rle) {
        uint4 *z = (uint4 *)(work + (uint64_t)d.reserved * 16u);
        const uint64_t nz = (uint64_t)n;                                 // 2 n slots of 8 bytes = n uint4
        for (uint64_t i = threadIdx.x; i < nz; i += 256) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Info *I = (Info *)(gscratch + d.scratch_off);
        uint32_t mx = 0;
        for (int w = 0; w < 8; w++) { I->present[w] = pres[w]; if (pres[w]) mx = 32u * (uint32_t)w + 31u - (uint32_t)__builtin_clz(pres[w]); }
        I->m = mx + 1u;
    }
}

// ---- a model of at most 64 symbols in registers: lane l holds entry l ((freq << 8) | symbol, sorted by frequency like every model of this coder) and the
//      inclusive prefix sum of the frequencies up to it
struct RegModel {
    uint32_t e, incl, tot, n;
    __device__ __forceinline__ void init(uint32_t m, int lane) {
        n = m; tot = m;
        e = (uint32_t)lane < m ? (1u << 8) | (uint32_t)lane : 0u;
        incl = (uint32_t)lane < m ? (uint32_t)lane + 1u : m;
    }
    // the triple of `sym` under the current state, then the update (arith_dev.h model_update: bump by STEP, halve all when the total passes MAX_FREQ, one
    // step towards the front when the entry outgrew its neighbour)
    __device__ __forceinline__ void step(uint32_t sym, uint32_t &cum, uint32_t &f, uint32_t &totb, int lane) {
        const unsigned long long hit = __ballot((uint32_t)lane < n && (e & 0xffu) == sym);
        const uint32_t l = (uint32_t)__builtin_ctzll(hit);
        const uint32_t ex = rl(e, l);
        f = ex >> 8; cum = rl(incl, l) - f; totb = tot;
        uint32_t nex = ex + (STEP << 8);
        tot += STEP;
        if (tot > MAX_FREQ) {                                            // halve every frequency (rare)
            e = (uint32_t)lane == l ? nex : e;
            uint32_t fr = e >> 8; fr -= fr >> 1;
            e = (uint32_t)lane < n ? (fr << 8) | (e & 0xffu) : 0u;
            incl = wave_incl_scan_dpp((uint32_t)lane < n ? fr : 0u);
            tot = rl(incl, 63);
            nex = rl(e, l);
        } else incl += (uint32_t)lane >= l ? STEP : 0u;
        if (l) {
            const uint32_t ep = rl(e, l - 1u);
            if ((nex >> 8) > (ep >> 8)) {
                const uint32_t through = rl(incl, l);                      // the sum through entry l does not change with the swap
                e = (uint32_t)lane == l ? ep : (uint32_t)lane + 1u == l ? nex : e;
                incl = (uint32_t)lane + 1u == l ? through - (ep >> 8) : incl;
                return;
            }
        }
        e = (uint32_t)lane == l ? nex : e;
    }
};

// records gathered 64 at a time, then one scattered store
struct RecOut {
    uint2 *R; uint32_t idx, lo, hi, cnt;
    __device__ __forceinline__ void start(uint2 *r) { R = r; idx = lo = hi = 0; cnt = 0; }
    __device__ __forceinline__ void put(uint32_t slot, uint32_t cum, uint32_t f, uint32_t tot, int lane) {
        idx = hg::writelane(slot, cnt, idx); lo = hg::writelane(cum | f << 16, cnt, lo); hi = hg::writelane(tot | REC_VALID, cnt, hi);
        if (++cnt == 64u) { R[idx] = make_uint2(lo, hi); cnt = 0; }
    }
    __device__ __forceinline__ void finish(int lane) { if ((uint32_t)lane < cnt) R[idx] = make_uint2(lo, hi); cnt = 0; }
};

// task word: model id (10 bits: 0..255 literal context, 256 + c run model of symbol c, 512 / 513 the run models 256 / 257) | index into sel << 10
constexpr uint32_t TASK_MODEL_BITS = 10;

// Every task walks its whole stream, 64 positions per step, and most steps find little to do: the loads of the next FOUR steps are in flight while four
// are worked on.  pre(i0, cur, prev) -> the step's event mask; body(i0, cur, mask) works through it.  The masks of all four steps are formed BEFORE the next
// loads are issued (and the scheduler is kept from moving the loads up): the first use of a loaded register makes the compiler wait for every load in
// flight -- vmcnt counts in order and nothing is known across the loop edge -- so loads issued ahead of that use were waited for at once and the walk
// ran at memory latency.  cur / prev: lane l holds the byte at i0 + l and the one before it (0 before the stream; 0x100 / 0 past its end).
template <class Pre, class Body>
__device__ __forceinline__ void scan_tiles(const uint8_t *src, uint32_t n, int lane, Pre pre, Body body) {
    auto load = [&](uint32_t i0, uint32_t &cur, uint32_t &prev) {
        const uint32_t p = i0 + (uint32_t)lane;
        cur = p < n ? (uint32_t)src[p] : 0x100u; prev = (p && p < n) ? (uint32_t)src[p - 1u] : 0u;
    };
    uint32_t nc[4], np[4];
#pragma unroll
    for (int t = 0; t < 4; t++) load(64u * (uint32_t)t, nc[t], np[t]);
    for (uint32_t g0 = 0; g0 < n; g0 += 256u) {
        uint32_t c[4]; unsigned long long m[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { c[t] = nc[t]; m[t] = pre(g0 + 64u * (uint32_t)t, nc[t], np[t]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; t++) load(g0 + 256u + 64u * (uint32_t)t, nc[t], np[t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; t++) { const uint32_t i0 = g0 + 64u * (uint32_t)t; if (i0 < n) body(i0, c[t], m[t]); }
    }
}

// The literal events of one context (order 1: the positions whose predecessor is `ctx`; order 0: every position).  With RLE only the first symbol of a run
// is a literal, and its context is the symbol of the run before -- the byte before it.
template <bool WIDE>
__device__ __forceinline__ void lit_task(const uint8_t *src, uint32_t n, uint32_t order, uint32_t rle, uint32_t ctx, uint32_t m, uint2 *R, uint32_t *wide_mem, int lane) {
    RegModel G; WideO0 W;
    if (WIDE) {
        for (uint32_t i = (uint32_t)lane; i < m; i += 64) wide_mem[i] = (1u << 8) | i;      // models_init: every frequency 1, entry i holds symbol i
        W.init(wide_mem, (uint8_t *)(wide_mem + 256), m, true, lane);
    } else G.init(m, lane);
    RecOut O; O.start(R);
    scan_tiles(src, n, lane, [&](uint32_t i0, uint32_t cur, uint32_t prev) -> unsigned long long {
        const uint32_t p = i0 + (uint32_t)lane;
        return __ballot(p < n && (!order || prev == ctx) && (!rle || p == 0 || cur != prev));
    }, [&](uint32_t i0, uint32_t cur, unsigned long long mask) {
        while (mask) {
            const uint32_t b = (uint32_t)__builtin_ctzll(mask); mask &= mask - 1ull;
            const uint32_t sym = rl(cur, b);
            uint32_t cum, f, t;
            if (WIDE) {
                const uint32_t x = hg::uni((uint32_t)W.pos[sym]), pc = x >> 6, l = x & 63u;
                const uint32_t e = W.piece(pc, lane);
                const uint32_t incl = wave_incl_scan_dpp(e >> 8);
                const uint32_t ex = rl(e, l);
                f = ex >> 8; cum = W.base(pc) + rl(incl, l) - f; t = W.tot;
                W.bump<true>(pc, l, ex, e, lane);
            } else G.step(sym, cum, f, t, lane);
            O.put(rle ? 2u * (i0 + b) : i0 + b, cum, f, t, lane);
        }
    });
    O.finish(lane);
}

// RLE: a run of r + 1 copies of c is coded as the literal c, then r in parts of at most 3 -- the first part with the run model of c, the second with run
// model 256, all further ones with run model 257; a part below 3 ends the list (arith.hip's encoder loop).
// This task: the first part of every run of symbol c (slot 2 s + 1 of the run starting at s).
__device__ __forceinline__ void run_first_task(const uint8_t *src, uint32_t n, uint32_t c, uint2 *R, int lane) {
    RegModel G; G.init(4, lane);
    RecOut O; O.start(R);
    auto emit = [&](uint32_t s, uint32_t len) {
        const uint32_t r = len - 1u;
        uint32_t cum, f, t;
        G.step(r < 3u ? r : 3u, cum, f, t, lane);
        O.put(2u * s + 1u, cum, f, t, lane);
    };
    bool open = false; uint32_t s = 0, len = 0;
    scan_tiles(src, n, lane, [&](uint32_t, uint32_t cur, uint32_t) -> unsigned long long {
        return __ballot(cur == c);                                        // the runs of c are the runs of set bits (positions past the end never match)
    }, [&](uint32_t i0, uint32_t, unsigned long long bits) {
        if (open) {
            const uint32_t ones = ~bits ? (uint32_t)__builtin_ctzll(~bits) : 64u;
            len += ones;
            if (ones == 64u) return;
            emit(s, len); open = false;
            bits &= ~((1ull << ones) - 1ull);
        }
        while (bits) {
            const uint32_t b = (uint32_t)__builtin_ctzll(bits);
            const unsigned long long rest = ~(bits >> b);                 // (the shift brings zeros in at the top: a zero bit is found unless b = 0 and all 64 are set)
            const uint32_t ones = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
            if (b + ones == 64u) { open = true; s = i0 + b; len = ones; break; }   // may go on in the next tile
            emit(i0 + b, ones);
            bits &= ~(((1ull << ones) - 1ull) << b);
        }
    });
    if (open) emit(s, len);
    O.finish(lane);
}
// This task: the second part (which = 0, run model 256, runs of 4 and more) or the third and further parts (which = 1, run model 257, runs of 7 and more) of the
// runs of ANY symbol.  Tiles without three non-starts in a row only have their first start looked at (it may end a long run from earlier tiles).
__device__ __forceinline__ void run_more_task(const uint8_t *src, uint32_t n, uint32_t which, uint2 *R, int lane) {
    RegModel G; G.init(4, lane);
    RecOut O; O.start(R);
    auto handle = [&](uint32_t s, uint32_t len) {
        const uint32_t r = len - 1u;
        uint32_t cum, f, t;
        if (!which) {
            if (r < 3u) return;
            G.step(r - 3u < 3u ? r - 3u : 3u, cum, f, t, lane);
            O.put(2u * s + 2u, cum, f, t, lane);
        } else {
            if (r < 6u) return;
            uint32_t rem = r - 6u, j = 3u, part;
            do {
                part = rem < 3u ? rem : 3u;
                G.step(part, cum, f, t, lane);
                O.put(2u * s + j, cum, f, t, lane);
                j++; rem -= part;
            } while (part == 3u);
        }
    };
    bool have = false; uint32_t s_last = 0;
    scan_tiles(src, n, lane, [&](uint32_t i0, uint32_t cur, uint32_t prev) -> unsigned long long {
        const uint32_t p = i0 + (uint32_t)lane;
        return __ballot(p < n && (p == 0 || cur != prev));                // the run starts
    }, [&](uint32_t i0, uint32_t, unsigned long long st) {
        const unsigned long long valid = n - i0 >= 64u ? ~0ull : (1ull << (n - i0)) - 1ull;
        if (!st) return;
        const unsigned long long nst = valid & ~st, longish = nst & (nst >> 1) & (nst >> 2);
        uint32_t a = (uint32_t)__builtin_ctzll(st);
        if (have) handle(s_last, i0 + a - s_last);
        if (longish) {
            unsigned long long bits = st & (st - 1ull);
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctzll(bits); bits &= bits - 1ull;
                handle(i0 + a, b - a);
                a = b;
            }
        }
        s_last = i0 + 63u - (uint32_t)__builtin_clzll(st); have = true;
    });
    if (have) handle(s_last, n - s_last);
    O.finish(lane);
}

// Four tasks per workgroup, one per wavefront (they never meet: no barrier): a quarter of the workgroups to dispatch.
__global__ __launch_bounds__(256)
void model_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel,
                  const uint32_t *__restrict__ tasks, uint32_t ntasks, const uint32_t *gscratch, uint8_t *work) {
    __shared__ uint32_t wide_mem_all[4][256 + 64];                       // per wavefront, WideO0: 256 entries + the symbol -> position bytes
    const int lane = threadIdx.x & 63;
    const uint32_t tix = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tix >= ntasks) return;
    uint32_t *wide_mem = wide_mem_all[threadIdx.x >> 6];
    const uint32_t task = tasks[tix], model = task & ((1u << TASK_MODEL_BITS) - 1u), k = sel[task >> TASK_MODEL_BITS];
    const hg_stream_desc d = desc[k];
    const Info *I = (const Info *)(gscratch + d.scratch_off);
    const uint32_t flags = flags_in[k], order = flags & F_ORDER, rle = flags & F_RLE, m = I->m, n = d.in_len;
    const uint8_t *src = in + d.in_off;
    uint2 *R = (uint2 *)(work + (uint64_t)d.reserved * 16u);
    auto present = [&](uint32_t c) { return c < m && ((I->present[c >> 5] >> (c & 31u)) & 1u) != 0u; };
    if (model < 256u) {
        if (order) { if (model ? !present(model) : m == 0u) return; }     // (context 0 also codes the first symbol)
        else if (model) return;
        if (m > 64u) lit_task<true>(src, n, order, rle, model, m, R, wide_mem, lane);
        else lit_task<false>(src, n, order, rle, model, m, R, wide_mem, lane);
    } else if (!rle) return;
    else if (model < 512u) { if (present(model - 256u)) run_first_task(src, n, model - 256u, R, lane); }
    else run_more_task(src, n, model - 512u, R, lane);
}

// ---- phase B
__global__ __launch_bounds__(64)
void code_kernel(const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel, const uint32_t *gscratch,
                 const uint8_t *work, uint8_t *out, uint32_t *out_len) {
    const int lane = threadIdx.x;
    const uint32_t k = sel[blockIdx.x];
    const hg_stream_desc d = desc[k];
    const Info *I = (const Info *)(gscratch + d.scratch_off);
    const uint32_t n = d.in_len, rle = flags_in[k] & F_RLE, nslots = rle ? 2u * n : n;
    const uint2 *R = (const uint2 *)(work + (uint64_t)d.reserved * 16u);
    uint8_t *o = out + d.out_off;
    o[0] = (uint8_t)I->m;                                                  // every lane, same byte (256 -> 0)
    Encoder E;
    E.start(o + 1);
    auto tile = [&](uint32_t s0) { const uint32_t s = s0 + (uint32_t)lane; return s < nslots ? R[s] : make_uint2(0u, 0u); };
    uint2 rec = tile(0), nrec = make_uint2(0u, 0u);
    for (uint32_t s0 = 0; s0 < nslots; s0 += 64) {
        if (s0 + 64u < nslots) nrec = tile(s0 + 64u);
        const uint32_t tv = rec.y & 0xffffu;
        // floor((2^32 - 1) / total), every lane for its own record: range / total is then a multiply-high, short by 2 at most
        const uint32_t iv = tv <= 1u ? 0xffffffffu : udiv_small_divisor(0xffffffffu, tv);
        unsigned long long mask = __ballot((rec.y & REC_VALID) != 0u);
        while (mask) {
            const uint32_t j = (uint32_t)__builtin_ctzll(mask); mask &= mask - 1ull;
            const uint32_t lo = rl(rec.x, j), t = rl(tv, j), inv = rl(iv, j);
            const uint32_t cum = lo & 0xffffu, f = lo >> 16;
            uint32_t q = __umulhi(E.range, inv), r = E.range - q * t;
            if (r >= t) { q++; r -= t; }
            if (r >= t) q++;
            const uint32_t old = E.low;
            E.low += cum * q;
            E.range = q * f;
            if (E.low < old) E.carry = 1;
            while (E.range < TOP) { E.range <<= 8; E.shift_low(lane); }
        }
        rec = nrec;
    }
    const uint32_t total = 1u + E.finish(lane);
    out_len[k] = total;                                                    // every lane stores the same word
}

}  // namespace hga2

namespace hg {
// sel2 / n2: the two-phase streams of this call (indices into desc); tasks: (model | position in sel2 << 10) words, ntasks of them, on the device
int launch_arith_encode2(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags, const uint32_t *d_sel2, size_t n2, const uint32_t *d_tasks,
                         size_t ntasks, void *d_out, uint32_t *d_out_len, uint32_t *d_scratch, void *d_work, hipStream_t s) {
    if (!n2) return HG_OK;
    hipLaunchKernelGGL(hga2::prepass_kernel, dim3((unsigned)n2), dim3(256), 0, s, (const uint8_t *)d_in, d_desc, d_flags, d_sel2, d_scratch, (uint8_t *)d_work);
    hipLaunchKernelGGL(hga2::model_kernel, dim3((unsigned)((ntasks + 3) / 4)), dim3(256), 0, s, (const uint8_t *)d_in, d_desc, d_flags, d_sel2, d_tasks, (uint32_t)ntasks, (const uint32_t *)d_scratch,
                       (uint8_t *)d_work);
    hipLaunchKernelGGL(hga2::code_kernel, dim3((unsigned)n2), dim3(64), 0, s, d_desc, d_flags, d_sel2, (const uint32_t *)d_scratch, (const uint8_t *)d_work, (uint8_t *)d_out, d_out_len);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

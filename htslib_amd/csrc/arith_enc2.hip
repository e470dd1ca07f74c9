// arith_enc2.hip -- the adaptive range coder's ENCODER in two phases (reference call site cram/cram_io.c:1869-1883, arith_compress_to; format and
// arithmetic per oracle/arith_oracle.c -- PARITY UNPINNED; byte-identical to arith.hip's one-pass encoder and to that oracle).
//
// An encoder -- unlike a decoder -- knows every symbol and therefore every CONTEXT up front.  Two dependency chains are tangled in the one-pass coder:
// (1) each adaptive model's state, which only the events of THAT model touch, and (2) the coder registers (low, range), which only need each event's
// (cumulative frequency, frequency, total).  Untangled, round 5 form (work proportional to the EVENTS; round 4's tasks each walked the whole stream):
//
//   sort     one wavefront per stream walks it twice, 64 positions per step: a stable counting sort of the events by model.  Literal events are keyed by
//            their context (order 1: the byte before; with RLE only the first symbol of a run is a literal), the first run-length part of a run by the
//            run's symbol; second / further parts are compacted in stream order.  Same-key lanes of a step rank themselves with 8 ballots (one per key
//            bit), so a step costs the same whatever the key distribution.  Every event gets its DENSE number in coding order (no RLE: its position;
//            RLE: a prefix sum of 2 + r / 3 events per run of r + 1 symbols), which is where its record goes.  The wave then appends the stream's tasks
//            to the call's task list: a model with many events is a task of its own (front of the list: long chains start first), small models of
//            a stream are bundled.
//   phase A  persistent wavefronts take tasks by ticket.  A task reads only ITS events (index + symbol lists), keeps the model in REGISTERS -- lane l
//            holds entry l, the running prefix sums are maintained incrementally (one masked add per event, no scan) -- and writes one 8-byte record
//            (cum | freq << 16, total) per event at the event's number.  Models of more than 64 symbols use arith_dev.h's WideO0 (LDS).
//   phase B  one wavefront per stream walks the dense records, 64 per load; floor((2^32 - 1) / total) is computed by the 64 lanes for 64 records at
//            once, and the coder step itself -- range / total as a multiply-high + correction, low += cum * r with the carry in a 64-bit add,
//            range = r * freq, renormalisation -- is unrolled over the tile with constant lane numbers, so that it compiles to SCALAR instructions
//            (s_mul_hi_u32, s_add_u32 / s_addc_u32, s_cselect): ~25 per event against ~55 with the records picked through VGPRs and ~130 in one pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

#include "arith_dev.h"

namespace hga2 {
using namespace hga;
using hg::Arith2pInfo;
using hg::arith2p_layout;

constexpr uint32_t BIG_TASK = 1024;      // events from which a model is a task of its own
constexpr uint32_t BUNDLE = 1024;        // small models of a stream are bundled up to about this many events
constexpr uint32_t M_R2 = 512, N_MODELS = 514;          // model ids: 0..255 literal context, 256 + c run model of symbol c, 512 / 513 the run models 256 / 257

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// the lanes of `act` that hold the same 8-bit key as this lane (valid on act lanes)
__device__ __forceinline__ unsigned long long match8(uint32_t key, bool act) {
    unsigned long long m = __ballot(act);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (key >> b) & 1u;
        const unsigned long long bal = __ballot(act && bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

struct SortOut { uint32_t *lidx; uint8_t *lsym; uint32_t *r1, *r2; uint2 *r3; };

// One walk over the stream.  PLACE = false: count (cL / cR = events per literal context / per first-part run model).  PLACE = true: cL / cR hold the
// cursors (exclusive prefix sums of the counts) and the lists are written.  Returns the number of events; nr2 / nr3 = entries of the second / further
// part lists, maxsym per lane (count pass).
template <bool PLACE>
__device__ __forceinline__ uint32_t sort_walk(const uint8_t *__restrict__ src, uint32_t n, uint32_t order, uint32_t rle, uint32_t *cL, uint32_t *cR, const SortOut &O,
                                              uint32_t &nr2, uint32_t &nr3, uint32_t &maxsym, int lane) {
    const unsigned long long lt = lanes_below(lane);
    uint32_t e_carry = 0, s_last = 0, c_last = 0;
    nr2 = nr3 = 0; maxsym = 0;
    auto load = [&](uint32_t i0, uint32_t &cur, uint32_t &prev) {
        const uint32_t p = i0 + (uint32_t)lane;
        cur = p < n ? (uint32_t)src[p] : 0u; prev = (p && p < n) ? (uint32_t)src[p - 1u] : 0u;
    };
    uint32_t ncur, nprev;
    load(0, ncur, nprev);
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const uint32_t cur = ncur, prev = nprev;
        if (i0 + 64u < n) load(i0 + 64u, ncur, nprev);
        const uint32_t p = i0 + (uint32_t)lane;
        const bool in = p < n;
        if (!PLACE) maxsym = cur > maxsym ? cur : maxsym;
        if (!rle) {
            const uint32_t key = order ? prev : 0u;
            const unsigned long long m = order ? match8(key, in) : __ballot(in);
            const uint32_t rank = (uint32_t)__popcll(m & lt), cnt = (uint32_t)__popcll(m);
            if (!PLACE) { if (in && rank == 0u) cL[key] += cnt; }
            else {
                const uint32_t base = in ? cL[key] : 0u;
                if (in && rank == 0u) cL[key] = base + cnt;
                if (in) { O.lidx[base + rank] = p; O.lsym[base + rank] = (uint8_t)cur; }
            }
            continue;
        }
        // ---- RLE: runs.  A lane that starts a run (other than the stream's first) CLOSES the run before it.
        const bool start = in && (p == 0u || cur != prev);
        const unsigned long long st = __ballot(start);
        if (!st) continue;                                                   // a long run goes on
        const bool closer = start && p != 0u;
        const unsigned long long below = st & lt;
        const int pl = below ? 63 - (int)__builtin_clzll(below) : -1;      // the start before this lane inside the step
        const uint32_t c_in = (uint32_t)__shfl((int)cur, pl < 0 ? 0 : pl);
        const uint32_t s_prev = pl >= 0 ? i0 + (uint32_t)pl : s_last, c_prev = pl >= 0 ? c_in : c_last;
        const uint32_t r = closer ? p - s_prev - 1u : 0u;                  // the closed run has r + 1 symbols
        const uint32_t nev = closer ? 2u + r / 3u : 0u;                    // its literal + its r / 3 + 1 run-length parts
        const uint32_t incl = wave_incl_scan_dpp(nev);
        const uint32_t e_closed = e_carry + incl - nev, e_new = e_carry + incl;   // first event of the closed run / of the run that starts here
        {   // literal of the run that starts here: context = the symbol before (order 1)
            const uint32_t key = order ? prev : 0u;
            const unsigned long long m = order ? match8(key, start) : st;
            const uint32_t rank = (uint32_t)__popcll(m & lt), cnt = (uint32_t)__popcll(m);
            if (!PLACE) { if (start && rank == 0u) cL[key] += cnt; }
            else {
                const uint32_t base = start ? cL[key] : 0u;
                if (start && rank == 0u) cL[key] = base + cnt;
                if (start) { O.lidx[base + rank] = e_new; O.lsym[base + rank] = (uint8_t)cur; }
            }
        }
        {   // first part of the closed run: the run model of its symbol
            const unsigned long long m = match8(c_prev, closer);
            const uint32_t rank = (uint32_t)__popcll(m & lt), cnt = (uint32_t)__popcll(m);
            if (!PLACE) { if (closer && rank == 0u) cR[c_prev] += cnt; }
            else {
                const uint32_t base = closer ? cR[c_prev] : 0u;
                if (closer && rank == 0u) cR[c_prev] = base + cnt;
                if (closer) O.r1[base + rank] = (e_closed + 1u) << 2 | (r < 3u ? r : 3u);
            }
        }
        {   // second part (runs of 4 and more) and further parts (7 and more): stream order
            const unsigned long long b2 = __ballot(closer && r >= 3u), b3 = __ballot(closer && r >= 6u);
            if (PLACE) {
                if (closer && r >= 3u) O.r2[nr2 + (uint32_t)__popcll(b2 & lt)] = (e_closed + 2u) << 2 | (r - 3u < 3u ? r - 3u : 3u);
                if (closer && r >= 6u) O.r3[nr3 + (uint32_t)__popcll(b3 & lt)] = make_uint2(e_closed + 3u, r - 6u);
            }
            nr2 += (uint32_t)__popcll(b2); nr3 += (uint32_t)__popcll(b3);
        }
        e_carry += rl(incl, 63);
        const uint32_t top = 63u - (uint32_t)__builtin_clzll(st);
        s_last = i0 + top; c_last = rl(cur, top);
    }
    if (!rle) return n;
    if (n) {                                                                 // the last run closes at the end of the stream
        const uint32_t r = n - s_last - 1u;
        if (!PLACE) { if (lane == 0) cR[c_last] += 1u; }
        else if (lane == 0) {
            const uint32_t base = cR[c_last]; cR[c_last] = base + 1u;
            O.r1[base] = (e_carry + 1u) << 2 | (r < 3u ? r : 3u);
            if (r >= 3u) O.r2[nr2] = (e_carry + 2u) << 2 | (r - 3u < 3u ? r - 3u : 3u);
            if (r >= 6u) O.r3[nr3] = make_uint2(e_carry + 3u, r - 6u);
        }
        nr2 += r >= 3u ? 1u : 0u; nr3 += r >= 6u ? 1u : 0u;
        e_carry += 2u + r / 3u;
    }
    return e_carry;
}

// task list of a call: ctr[0] = big tasks (stored from the front), ctr[1] = bundles (stored from the back), ctr[2] = the ticket counter of phase A
__device__ __forceinline__ void push_task(uint32_t *ctr, uint2 *tasks, uint32_t cap, bool big, uint32_t q, uint32_t first, uint32_t last) {
    const uint32_t i = atomicAdd(&ctr[big ? 0 : 1], 1u);
    tasks[big ? i : cap - 1u - i] = make_uint2(q, first | last << 10 | (big ? 0u : 1u << 20));
}

__global__ __launch_bounds__(64)
void sort_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel,
                 uint32_t *gscratch, uint8_t *work, uint32_t *ctr, uint2 *tasks, uint32_t task_cap) {
    __shared__ uint32_t cntL[256], cntR[256], curL[256], curR[256];
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x, k = sel[q];
    const hg_stream_desc d = desc[k];
    const uint8_t *src = in + d.in_off;
    const uint32_t n = d.in_len, flags = flags_in[k], order = flags & F_ORDER, rle = flags & F_RLE;
    Arith2pInfo *I = (Arith2pInfo *)(gscratch + d.scratch_off);
    uint8_t *W = work + (uint64_t)d.reserved * 16u;
    const hg::Arith2pLayout L = arith2p_layout(n, rle != 0u);
    SortOut O{(uint32_t *)(W + L.lidx), W + L.lsym, (uint32_t *)(W + L.r1), (uint32_t *)(W + L.r2), (uint2 *)(W + L.r3)};
    for (int i = lane; i < 256; i += 64) cntL[i] = cntR[i] = 0u;
    wave_sync();
    uint32_t nr2, nr3, mx;
    const uint32_t nev = sort_walk<false>(src, n, order, rle, cntL, cntR, O, nr2, nr3, mx, lane);
    wave_sync();
    // the largest symbol (wave maximum)
    for (int s = 32; s; s >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, s); mx = o > mx ? o : mx; }
    // exclusive prefix sums of the counts: lane l takes entries 4 l .. 4 l + 3
    auto scan256 = [&](const uint32_t *cnt, uint32_t *cur, uint32_t *off) {
        const uint32_t a0 = cnt[4 * lane], a1 = cnt[4 * lane + 1], a2 = cnt[4 * lane + 2], a3 = cnt[4 * lane + 3], s = a0 + a1 + a2 + a3;
        const uint32_t incl = wave_incl_scan_dpp(s), ex = incl - s;
        cur[4 * lane] = off[4 * lane] = ex; cur[4 * lane + 1] = off[4 * lane + 1] = ex + a0;
        cur[4 * lane + 2] = off[4 * lane + 2] = ex + a0 + a1; cur[4 * lane + 3] = off[4 * lane + 3] = ex + a0 + a1 + a2;
        if (lane == 63) off[256] = incl;
    };
    scan256(cntL, curL, I->offL);
    scan256(cntR, curR, I->offR);
    if (lane == 0) { I->m = n ? mx + 1u : 0u; I->nevents = nev; I->n_r2 = nr2; I->n_r3 = nr3; }
    wave_sync();
    uint32_t x2, x3, xm;
    (void)sort_walk<true>(src, n, order, rle, curL, curR, O, x2, x3, xm, lane);
    // ---- this stream's tasks
    if (lane == 0) {
        uint32_t acc = 0, first = 0, last = 0; bool open = false;
        for (uint32_t mdl = 0; mdl < N_MODELS; mdl++) {
            const uint32_t c = mdl < 256u ? cntL[mdl] : mdl < 512u ? cntR[mdl - 256u] : mdl == M_R2 ? nr2 : nr3;
            if (!c) continue;
            if (c >= BIG_TASK) { push_task(ctr, tasks, task_cap, true, q, mdl, mdl); continue; }
            if (!open) { first = mdl; open = true; acc = 0; }
            last = mdl; acc += c;
            if (acc >= BUNDLE) { push_task(ctr, tasks, task_cap, false, q, first, last); open = false; }
        }
        if (open) push_task(ctr, tasks, task_cap, false, q, first, last);
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
        idx = hg::writelane(slot, cnt, idx); lo = hg::writelane(cum | f << 16, cnt, lo); hi = hg::writelane(tot, cnt, hi);
        if (++cnt == 64u) { R[idx] = make_uint2(lo, hi); cnt = 0; }
    }
    __device__ __forceinline__ void finish(int lane) { if ((uint32_t)lane < cnt) R[idx] = make_uint2(lo, hi); cnt = 0; }
};

// the literal events of one context: (record number, symbol) lists
template <bool WIDE>
__device__ __forceinline__ void lit_model(const uint32_t *__restrict__ lidx, const uint8_t *__restrict__ lsym, uint32_t cnt, uint32_t m, RecOut &O, uint32_t *wide_mem, int lane) {
    RegModel G; WideO0 Wd;
    if (WIDE) {
        for (uint32_t i = (uint32_t)lane; i < m; i += 64) wide_mem[i] = (1u << 8) | i;      // models_init: every frequency 1, entry i holds symbol i
        Wd.init(wide_mem, (uint8_t *)(wide_mem + 256), m, true, lane);
    } else G.init(m, lane);
    uint32_t ni = (uint32_t)lane < cnt ? lidx[lane] : 0u, ns = (uint32_t)lane < cnt ? (uint32_t)lsym[lane] : 0u;
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
        const uint32_t iv = ni, sv = ns, nn = cnt - k0 < 64u ? cnt - k0 : 64u;
        { const uint32_t k = k0 + 64u + (uint32_t)lane; if (k < cnt) { ni = lidx[k]; ns = (uint32_t)lsym[k]; } }
        for (uint32_t b = 0; b < nn; b++) {
            const uint32_t sym = rl(sv, b);
            uint32_t cum, f, t;
            if (WIDE) {
                const uint32_t x = hg::uni((uint32_t)Wd.pos[sym]), pc = x >> 6, l = x & 63u;
                const uint32_t e = Wd.piece(pc, lane);
                const uint32_t incl = wave_incl_scan_dpp(e >> 8);
                const uint32_t ex = rl(e, l);
                f = ex >> 8; cum = Wd.base(pc) + rl(incl, l) - f; t = Wd.tot;
                Wd.bump<true>(pc, l, ex, e, lane);
            } else G.step(sym, cum, f, t, lane);
            O.put(rl(iv, b), cum, f, t, lane);
        }
    }
}

// RLE: a run of r + 1 copies of c is coded as the literal c, then r in parts of at most 3 -- the first part with the run model of c, the second with run
// model 256, all further ones with run model 257; a part below 3 ends the list (arith.hip's encoder loop).  First / second parts: one list word per
// event, record number << 2 | part.
__device__ __forceinline__ void part_model(const uint32_t *__restrict__ lst, uint32_t cnt, RecOut &O, int lane) {
    RegModel G; G.init(4, lane);
    uint32_t nw = (uint32_t)lane < cnt ? lst[lane] : 0u;
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
        const uint32_t wv = nw, nn = cnt - k0 < 64u ? cnt - k0 : 64u;
        { const uint32_t k = k0 + 64u + (uint32_t)lane; if (k < cnt) nw = lst[k]; }
        for (uint32_t b = 0; b < nn; b++) {
            const uint32_t w = rl(wv, b);
            uint32_t cum, f, t;
            G.step(w & 3u, cum, f, t, lane);
            O.put(w >> 2, cum, f, t, lane);
        }
    }
}
// third and further parts (run model 257) of the runs of 7 and more: (first record number, r - 6) per run
__device__ __forceinline__ void more_model(const uint2 *__restrict__ lst, uint32_t cnt, RecOut &O, int lane) {
    RegModel G; G.init(4, lane);
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
        const uint32_t k = k0 + (uint32_t)lane, nn = cnt - k0 < 64u ? cnt - k0 : 64u;
        const uint2 wv = k < cnt ? lst[k] : make_uint2(0u, 0u);
        for (uint32_t b = 0; b < nn; b++) {
            uint32_t at = rl(wv.x, b), rem = rl(wv.y, b), part;
            do {
                part = rem < 3u ? rem : 3u;
                uint32_t cum, f, t;
                G.step(part, cum, f, t, lane);
                O.put(at, cum, f, t, lane);
                at++; rem -= part;
            } while (part == 3u);
        }
    }
}

// ---- phase A: persistent wavefronts, tasks by ticket (big tasks first)
__global__ __launch_bounds__(256)
void model_kernel(const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel, const uint32_t *gscratch,
                  uint8_t *work, uint32_t *ctr, const uint2 *__restrict__ tasks, uint32_t task_cap) {
    __shared__ uint32_t wide_mem_all[4][256 + 64];                       // per wavefront, WideO0: 256 entries + the symbol -> position bytes
    const int lane = threadIdx.x & 63;
    uint32_t *wide_mem = wide_mem_all[threadIdx.x >> 6];
    const uint32_t nbig = ctr[0], nsmall = ctr[1];
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&ctr[2], 1u);
        t = hg::uni(t);
        if (t >= nbig + nsmall) break;
        const uint2 task = t < nbig ? tasks[t] : tasks[task_cap - 1u - (t - nbig)];
        const uint32_t k = sel[task.x], first = task.y & 1023u, last = (task.y >> 10) & 1023u, bundle = task.y >> 20;
        const hg_stream_desc d = desc[k];
        const Arith2pInfo *I = (const Arith2pInfo *)(gscratch + d.scratch_off);
        const uint32_t rle = flags_in[k] & F_RLE, m = I->m, n = d.in_len;
        uint8_t *W = work + (uint64_t)d.reserved * 16u;
        const hg::Arith2pLayout L = arith2p_layout(n, rle != 0u);
        RecOut O; O.start((uint2 *)(W + L.rec));
        for (uint32_t mdl = first; mdl <= last; mdl++) {
            if (mdl < 256u) {
                const uint32_t o0 = I->offL[mdl], cnt = I->offL[mdl + 1u] - o0;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                if (m > 64u) lit_model<true>((const uint32_t *)(W + L.lidx) + o0, W + L.lsym + o0, cnt, m, O, wide_mem, lane);
                else lit_model<false>((const uint32_t *)(W + L.lidx) + o0, W + L.lsym + o0, cnt, m, O, wide_mem, lane);
            } else if (mdl < 512u) {
                const uint32_t o0 = I->offR[mdl - 256u], cnt = I->offR[mdl - 255u] - o0;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                part_model((const uint32_t *)(W + L.r1) + o0, cnt, O, lane);
            } else if (mdl == M_R2) {
                const uint32_t cnt = I->n_r2;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                part_model((const uint32_t *)(W + L.r2), cnt, O, lane);
            } else {
                const uint32_t cnt = I->n_r3;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                more_model((const uint2 *)(W + L.r3), cnt, O, lane);
            }
        }
        O.finish(lane);
    }
}

// ---- phase B: one wavefront per stream over the dense records; the coder registers are scalars
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
        uint32_t q = __umulhi(range, inv), r = range - q * t;          // short by 2 at most
        if (r >= t) { q++; r -= t; }
        if (r >= t) q++;
        const unsigned long long s = (unsigned long long)low + (unsigned long long)cum * q;    // cum * q < 2^32: cum < t, q = range / t
        low = (uint32_t)s; carry |= (uint32_t)(s >> 32);
        range = q * f;
        while (range < TOP) { range <<= 8; shift_low(lane); }
    }
    __device__ __forceinline__ uint32_t finish(int lane) {
        for (int i = 0; i < 5; i++) shift_low(lane);
        if ((uint32_t)lane < oidx) out[opos + (uint32_t)lane] = (uint8_t)obuf;
        return opos + oidx;
    }
};

__global__ __launch_bounds__(64)
void code_kernel(const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel, const uint32_t *gscratch,
                 const uint8_t *work, uint8_t *out, uint32_t *out_len) {
    const int lane = threadIdx.x;
    const uint32_t k = sel[blockIdx.x];
    const hg_stream_desc d = desc[k];
    const Arith2pInfo *I = (const Arith2pInfo *)(gscratch + d.scratch_off);
    const uint32_t ne = I->nevents;
    const uint2 *R = (const uint2 *)(work + (uint64_t)d.reserved * 16u);   // (records lie first in the stream's work area)
    uint8_t *o = out + d.out_off;
    o[0] = (uint8_t)I->m;                                                  // every lane, same byte (256 -> 0)
    Coder E;
    E.start(o + 1);
    auto tile = [&](uint32_t s0) { const uint32_t s = s0 + (uint32_t)lane; return s < ne ? R[s] : make_uint2(0u, 1u); };
    uint2 rec = tile(0), nrec = make_uint2(0u, 1u);
    for (uint32_t s0 = 0; s0 < ne; s0 += 64) {
        if (s0 + 64u < ne) nrec = tile(s0 + 64u);
        const uint32_t tv = rec.y & 0xffffu;
        // floor((2^32 - 1) / total), every lane for its own record: range / total is then a multiply-high, short by 2 at most
        const uint32_t iv = tv <= 1u ? 0xffffffffu : udiv_small_divisor(0xffffffffu, tv);
        if (ne - s0 >= 64u) {
#pragma unroll
            for (int j = 0; j < 64; j++) E.step(rl(rec.x, (uint32_t)j), rl(tv, (uint32_t)j), rl(iv, (uint32_t)j), lane);
        } else {
            const uint32_t nn = ne - s0;
            for (uint32_t j = 0; j < nn; j++) E.step(rl(rec.x, j), rl(tv, j), rl(iv, j), lane);
        }
        rec = nrec;
    }
    const uint32_t total = 1u + E.finish(lane);
    out_len[k] = total;                                                    // every lane stores the same word
}

}  // namespace hga2

namespace hg {
// sel2 / n2: the two-phase streams of this call (indices into desc); d_tasks: 16 counter words, then task_cap task slots (uint2) filled by the sort kernel
int launch_arith_encode2(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags, const uint32_t *d_sel2, size_t n2, void *d_tasks,
                         size_t task_cap, void *d_out, uint32_t *d_out_len, uint32_t *d_scratch, void *d_work, hipStream_t s) {
    if (!n2) return HG_OK;
    uint32_t *ctr = (uint32_t *)d_tasks;
    uint2 *tasks = (uint2 *)(ctr + 16);
    if (hipMemsetAsync(ctr, 0, 64, s) != hipSuccess) return HG_ELAUNCH;
    hipLaunchKernelGGL(hga2::sort_kernel, dim3((unsigned)n2), dim3(64), 0, s, (const uint8_t *)d_in, d_desc, d_flags, d_sel2, d_scratch, (uint8_t *)d_work, ctr, tasks, (uint32_t)task_cap);
    // persistent grid: enough wavefronts to fill the chip, never more than there can be tasks
    const size_t waves = task_cap < 256u * 32u ? task_cap : 256u * 32u;
    hipLaunchKernelGGL(hga2::model_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d_desc, d_flags, d_sel2, (const uint32_t *)d_scratch, (uint8_t *)d_work, ctr,
                       (const uint2 *)tasks, (uint32_t)task_cap);
    hipLaunchKernelGGL(hga2::code_kernel, dim3((unsigned)n2), dim3(64), 0, s, d_desc, d_flags, d_sel2, (const uint32_t *)d_scratch, (const uint8_t *)d_work, (uint8_t *)d_out, d_out_len);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

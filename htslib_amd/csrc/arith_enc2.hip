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
//            (cum | freq << 16, total) per event at the event's number.  Alphabets of more than 64 symbols: 2 or 4 entries per lane.
//   phase B  one wavefront per stream walks the dense records, 64 per load; floor((2^32 - 1) / total) is computed by the 64 lanes for 64 records at
//            once, and the coder step itself -- range / total as a multiply-high + correction, low += cum * r with the carry in a 64-bit add,
//            range = r * freq, renormalisation -- is unrolled over the tile with constant lane numbers, so that it compiles to SCALAR instructions
//            (s_mul_hi_u32, s_add_u32 / s_addc_u32, s_cselect): ~25 per event against ~55 with the records picked through VGPRs and ~130 in one pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

#include "arith_dev.h"
#include "range_enc2_dev.h"

namespace hga2 {
using hg::Arith2pInfo;
using hg::arith2p_layout;

constexpr uint32_t BIG_TASK = 1024;      // events from which a model is a task of its own
constexpr uint32_t BUNDLE = 1024;        // small models of a stream are bundled up to about this many events
constexpr uint32_t M_R2 = 512, N_MODELS = 514;          // model ids: 0..255 literal context, 256 + c run model of symbol c, 512 / 513 the run models 256 / 257

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

// the events of one model, 64 per load: idx = record numbers, sym = symbols (lane k = event k0 + k); full tiles run with constant lane numbers
template <int EPL>
__device__ __forceinline__ void model_tile(RegModel<EPL> &G, uint2 *R, uint32_t iv, uint32_t sv, uint32_t nn, int lane) {
    if (nn == 64u) {
#pragma unroll
        for (int j = 0; j < 64; j++) G.step(rl(sv, (uint32_t)j), R, rl(iv, (uint32_t)j), lane);
    } else for (uint32_t j = 0; j < nn; j++) G.step(rl(sv, j), R, rl(iv, j), lane);
}

// the literal events of one context: (record number, symbol) lists
template <int EPL>
__device__ __forceinline__ void lit_model(const uint32_t *__restrict__ lidx, const uint8_t *__restrict__ lsym, uint32_t cnt, uint32_t m, uint2 *R, int lane) {
    RegModel<EPL> G; G.init(m, lane);
    uint32_t ni = (uint32_t)lane < cnt ? lidx[lane] : 0u, ns = (uint32_t)lane < cnt ? (uint32_t)lsym[lane] : 0u;
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
        const uint32_t iv = ni, sv = ns, nn = cnt - k0 < 64u ? cnt - k0 : 64u;
        { const uint32_t k = k0 + 64u + (uint32_t)lane; if (k < cnt) { ni = lidx[k]; ns = (uint32_t)lsym[k]; } }
        model_tile<EPL>(G, R, iv, sv, nn, lane);
    }
}

// RLE: a run of r + 1 copies of c is coded as the literal c, then r in parts of at most 3 -- the first part with the run model of c, the second with run
// model 256, all further ones with run model 257; a part below 3 ends the list (arith.hip's encoder loop).  First / second parts: one list word per
// event, record number << 2 | part.
__device__ __forceinline__ void part_model(const uint32_t *__restrict__ lst, uint32_t cnt, uint2 *R, int lane) {
    RegModel<1> G; G.init(4, lane);
    uint32_t nw = (uint32_t)lane < cnt ? lst[lane] : 0u;
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
        const uint32_t wv = nw, nn = cnt - k0 < 64u ? cnt - k0 : 64u;
        { const uint32_t k = k0 + 64u + (uint32_t)lane; if (k < cnt) nw = lst[k]; }
        model_tile<1>(G, R, wv >> 2, wv & 3u, nn, lane);
    }
}
// third and further parts (run model 257) of the runs of 7 and more: (first record number, r - 6) per run
__device__ __forceinline__ void more_model(const uint2 *__restrict__ lst, uint32_t cnt, uint2 *R, int lane) {
    RegModel<1> G; G.init(4, lane);
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
        const uint32_t k = k0 + (uint32_t)lane, nn = cnt - k0 < 64u ? cnt - k0 : 64u;
        const uint2 wv = k < cnt ? lst[k] : make_uint2(0u, 0u);
        for (uint32_t b = 0; b < nn; b++) {
            uint32_t at = rl(wv.x, b), rem = rl(wv.y, b), part;
            do {
                part = rem < 3u ? rem : 3u;
                G.step(part, R, at, lane);
                at++; rem -= part;
            } while (part == 3u);
        }
    }
}

// ---- phase B inside the persistent kernel: the coder of one stream FOLLOWS the models.  Records are zero until a model task writes them (total >= 1: the second
//      word marks a record valid), so the coder takes a tile as soon as its 64 records are there and sleeps a few cycles otherwise: the passes overlap, a stream
//      costs max(models, coder) instead of their sum.  Coder roles are tickets BEHIND the model tasks: when one is handed out every model task of the call has been
//      taken by a resident wavefront, so whatever a coder waits for is being produced.  (A poll budget turns a logic error into a failed stream, not a hang.)
__device__ __forceinline__ void coder_role(const hg_stream_desc &d, const Arith2pInfo *I, const uint8_t *work, uint8_t *out, uint32_t *out_len, uint32_t k, int lane) {
    const uint32_t ne = I->nevents;
    const unsigned long long *R = (const unsigned long long *)(work + (uint64_t)d.reserved * 16u);   // (records lie first in the stream's work area)
    uint8_t *o = out + d.out_off;
    o[0] = (uint8_t)I->m;                                                  // every lane, same byte (256 -> 0)
    Coder E;
    E.start(o + 1);
    auto peek = [&](uint32_t s0) -> unsigned long long {                   // lane j: record s0 + j as it is now (beyond the end: a harmless valid one)
        const uint32_t s = s0 + (uint32_t)lane;
        return s < ne ? __hip_atomic_load(R + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1ull << 32;
    };
    bool failed = false;
    auto settle = [&](uint32_t s0, unsigned long long v) -> unsigned long long {   // ... once all 64 are valid
        uint32_t polls = 0;
        while (__ballot((uint32_t)(v >> 32) == 0u)) {
            __builtin_amdgcn_s_sleep(16);
            if (++polls > (1u << 21)) { failed = true; break; }        // ~1 s
            v = peek(s0);
        }
        return v;
    };
    // three tiles in flight behind the one being coded: a device-scope load comes from memory, several microseconds away
    unsigned long long cur = peek(0u), n1 = peek(64u), n2 = peek(128u);
    cur = settle(0u, cur);
    for (uint32_t s0 = 0; s0 < ne && !failed; s0 += 64) {
        const unsigned long long n3 = peek(s0 + 192u);
        code_tile(E, make_uint2((uint32_t)cur, (uint32_t)(cur >> 32)), ne - s0 >= 64u ? 64u : ne - s0, lane);
        cur = settle(s0 + 64u, n1);
        n1 = n2; n2 = n3;
    }
    const uint32_t total = 1u + E.finish(lane);
    out_len[k] = failed ? 0u : total;                                      // every lane stores the same word (0: the host reports the stream as failed)
}

// ---- phase A (+ B): persistent wavefronts, tickets: the model tasks (big ones first), then one coder role per stream
__global__ __launch_bounds__(256)
void model_kernel(const hg_stream_desc *__restrict__ desc, const uint8_t *__restrict__ flags_in, const uint32_t *__restrict__ sel, const uint32_t *gscratch,
                  uint8_t *work, uint32_t *ctr, const uint2 *__restrict__ tasks, uint32_t task_cap, uint32_t ncoders, uint8_t *out, uint32_t *out_len) {
    const int lane = threadIdx.x & 63;
    const uint32_t nbig = ctr[0], nsmall = ctr[1];
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&ctr[2], 1u);
        t = hg::uni(t);
        if (t >= nbig + nsmall) {
            const uint32_t q = t - (nbig + nsmall);
            if (q >= ncoders) break;
            const uint32_t kc = sel[q];
            const hg_stream_desc dc = desc[kc];
            coder_role(dc, (const Arith2pInfo *)(gscratch + dc.scratch_off), work, out, out_len, kc, lane);
            continue;
        }
        const uint2 task = t < nbig ? tasks[t] : tasks[task_cap - 1u - (t - nbig)];
        const uint32_t k = sel[task.x], first = task.y & 1023u, last = (task.y >> 10) & 1023u, bundle = task.y >> 20;
        const hg_stream_desc d = desc[k];
        const Arith2pInfo *I = (const Arith2pInfo *)(gscratch + d.scratch_off);
        const uint32_t rle = flags_in[k] & F_RLE, m = I->m, n = d.in_len;
        uint8_t *W = work + (uint64_t)d.reserved * 16u;
        const hg::Arith2pLayout L = arith2p_layout(n, rle != 0u);
        uint2 *R = (uint2 *)(W + L.rec);
        for (uint32_t mdl = first; mdl <= last; mdl++) {
            if (mdl < 256u) {
                const uint32_t o0 = I->offL[mdl], cnt = I->offL[mdl + 1u] - o0;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                const uint32_t *li = (const uint32_t *)(W + L.lidx) + o0; const uint8_t *ls = W + L.lsym + o0;
                if (m > 128u) lit_model<4>(li, ls, cnt, m, R, lane);
                else if (m > 64u) lit_model<2>(li, ls, cnt, m, R, lane);
                else lit_model<1>(li, ls, cnt, m, R, lane);
            } else if (mdl < 512u) {
                const uint32_t o0 = I->offR[mdl - 256u], cnt = I->offR[mdl - 255u] - o0;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                part_model((const uint32_t *)(W + L.r1) + o0, cnt, R, lane);
            } else if (mdl == M_R2) {
                const uint32_t cnt = I->n_r2;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                part_model((const uint32_t *)(W + L.r2), cnt, R, lane);
            } else {
                const uint32_t cnt = I->n_r3;
                if (!cnt || (bundle && cnt >= BIG_TASK)) continue;
                more_model((const uint2 *)(W + L.r3), cnt, R, lane);
            }
        }
    }
}

// ---- phase B: one wavefront per stream over the dense records (range_enc2_dev.h: Coder, code_tile).  lead: the stream starts with the alphabet-size byte
//      (the range coder's own streams; fqzcomp's coder bytes follow a header the host writes)
__global__ __launch_bounds__(64)
void code_kernel(const hg_stream_desc *__restrict__ desc, const uint32_t *__restrict__ sel, const uint32_t *gscratch, const uint8_t *work, uint8_t *out, uint32_t *out_len,
                 uint32_t lead) {
    const int lane = threadIdx.x;
    const uint32_t k = sel ? sel[blockIdx.x] : blockIdx.x;
    const hg_stream_desc d = desc[k];
    const Arith2pInfo *I = (const Arith2pInfo *)(gscratch + d.scratch_off);
    const uint32_t ne = I->nevents;
    const uint2 *R = (const uint2 *)(work + (uint64_t)d.reserved * 16u);   // (records lie first in the stream's work area)
    uint8_t *o = out + d.out_off;
    if (lead) o[0] = (uint8_t)I->m;                                        // every lane, same byte (256 -> 0)
    Coder E;
    E.start(o + lead);
    auto tile = [&](uint32_t s0) { const uint32_t s = s0 + (uint32_t)lane; return s < ne ? R[s] : make_uint2(0u, 1u); };
    uint2 rec = tile(0), nrec = make_uint2(0u, 1u);
    for (uint32_t s0 = 0; s0 < ne; s0 += 64) {
        if (s0 + 64u < ne) nrec = tile(s0 + 64u);
        code_tile(E, rec, ne - s0 >= 64u ? 64u : ne - s0, lane);
        rec = nrec;
    }
    out_len[k] = lead + E.finish(lane);                                    // every lane stores the same word
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
    // HG_ARITH_2P_TIMES=1: the three kernels' durations on stderr (synchronises; measurement runs only)
    static const bool times = getenv("HG_ARITH_2P_TIMES") && atoi(getenv("HG_ARITH_2P_TIMES")) > 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    auto mark = [&](int i) { if (times) { if (!ev[i]) (void)hipEventCreate(&ev[i]); (void)hipEventRecord(ev[i], s); } };
    mark(0);
    hipLaunchKernelGGL(hga2::sort_kernel, dim3((unsigned)n2), dim3(64), 0, s, (const uint8_t *)d_in, d_desc, d_flags, d_sel2, d_scratch, (uint8_t *)d_work, ctr, tasks, (uint32_t)task_cap);
    mark(1);
    // persistent grid: enough wavefronts to fill the chip, never more than there can be tasks + coder roles.  HG_ARITH_2P_OVERLAP=0: the coder as a kernel of
    // its own behind the models (A/B runs)
    static const bool overlap = !(getenv("HG_ARITH_2P_OVERLAP") && atoi(getenv("HG_ARITH_2P_OVERLAP")) == 0);
    const size_t roles = task_cap + (overlap ? n2 : 0);
    const size_t waves = roles < 256u * 32u ? roles : 256u * 32u;
    hipLaunchKernelGGL(hga2::model_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d_desc, d_flags, d_sel2, (const uint32_t *)d_scratch, (uint8_t *)d_work, ctr,
                       (const uint2 *)tasks, (uint32_t)task_cap, (uint32_t)(overlap ? n2 : 0), (uint8_t *)d_out, d_out_len);
    mark(2);
    if (!overlap) hipLaunchKernelGGL(hga2::code_kernel, dim3((unsigned)n2), dim3(64), 0, s, d_desc, d_sel2, (const uint32_t *)d_scratch, (const uint8_t *)d_work, (uint8_t *)d_out, d_out_len, 1u);
    mark(3);
    if (times && hipStreamSynchronize(s) == hipSuccess) {
        float a = 0, b = 0, c = 0; uint32_t h[4] = {0, 0, 0, 0};
        (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]); (void)hipEventElapsedTime(&c, ev[2], ev[3]);
        (void)hipMemcpy(h, ctr, 16, hipMemcpyDeviceToHost);
        fprintf(stderr, "[arith 2p] %zu streams, %u + %u tasks: sort %.3f ms, models%s %.3f ms, coder %.3f ms\n", n2, h[0], h[1], a, overlap ? " + coder (overlapped)" : "", b, c);
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
// the coder pass alone over n streams' dense records (fqzcomp's two-phase encoder): desc[k].scratch_off -> Arith2pInfo (nevents), .reserved * 16 -> records, no lead byte
int launch_range_code(const hg_stream_desc *d_desc, size_t n, const uint32_t *d_scratch, const void *d_work, void *d_out, uint32_t *d_out_len, hipStream_t s) {
    if (!n) return HG_OK;
    hipLaunchKernelGGL(hga2::code_kernel, dim3((unsigned)n), dim3(64), 0, s, d_desc, (const uint32_t *)nullptr, d_scratch, (const uint8_t *)d_work, (uint8_t *)d_out, d_out_len, 0u);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

// ransnx16.hip -- CRAM 3.1 "rANS Nx16" block decoder for MI355X (gfx950 / CDNA4).
//
// Replaces rans_uncompress_4x16() as called by cram_uncompress_block (reference
// cram/cram_io.c:1697-1714; implementation = htscodecs rANS_static4x16pr.c / 32x16pr, an ABSENT
// submodule).  Format per the hts-specs CRAM-codecs document as restated in
// oracle/ransnx16_oracle.c -- PARITY UNPINNED (no stock-htslib stream exists in the reference to
// check against); the GPU decoder is bit-exact with that oracle.
//
// Mapping: the N (4 or 32) interleaved rANS states of a stream live in N adjacent lanes -- sixteen 4-way streams per wavefront, ONE
// 32-way stream per wavefront (lanes 32..63 idle: two streams side by side ran in lock step, so an order-0 and an order-1 neighbour
// cost the sum of both decode chains); every step each lane decodes one symbol and the lanes whose state drops below 2^15 pull the next
// 16-bit words of the SHARED stream in lane order -- a ballot and a prefix popcount per step, which is exactly what the 32-way SIMD CPU
// decoders emulate with shuffles.  A stream is one chain of n / N steps of dependent LDS reads, and a launch lasts as long as its longest
// chain, so the tables are shaped for few reads per step: order 0 -- a slot -> symbol byte table + the 257-entry cumulative array; order 1
// with a small alphabet (<= 16 contexts of <= 16 symbols) -- a dense form, two reads per symbol (a 256-bucket index per context, then the
// (cumulative, symbol, next-context rank) entry and its neighbour); order 1 otherwise -- sparse per-context lists with a 64-bucket index,
// in LDS when they fit, else in global scratch.  Tables are parsed by the first lane of the group.
// Handled here: flags ORDER, X32, NOSZ (size from the descriptor), CAT.  The PACK / RLE / STRIPE
// transforms are undone by ransnx16_xform.hip after this kernel: the host planner
// (cram_entropy_host.hip) parses their headers and hands this kernel pre-parsed core descriptors
// (desc.reserved bit 31).  Raw streams carrying those flags are reported -3 by THIS kernel, which is
// what the device-resident entry point (hg_ransnx16_decode_dev) returns for them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgn {

constexpr uint32_t RANS_L = 1u << 15;
constexpr int WAVES = 4;
enum { F_ORDER = 1, F_X32 = 4, F_STRIPE = 8, F_NOSZ = 16, F_CAT = 32, F_RLE = 64, F_PACK = 128 };

struct GroupLds { uint16_t C[258]; };
// 32-way streams (the big data series) keep their order-1 tables in LDS when they fit: a symbol lookup is a chain
// of 6-7 DEPENDENT table reads, which from global memory (~600 cycles each) capped a stream at ~23 MB/s.
#ifndef HG_O1_POOL
#define HG_O1_POOL 4352
#endif
constexpr uint32_t O1_LDS_WORDS = HG_O1_POOL;          // per stream; 4 streams (wavefronts) per workgroup -> 68 KiB, two workgroups per CU
// Order 0 uses the same pool as a direct slot -> symbol table (4096 one-byte entries) instead of a binary search.

__device__ __forceinline__ uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ int get_u7(const uint8_t *&cp, const uint8_t *end, uint32_t &v) {
    uint32_t x = 0;
    for (int n = 0; n < 5; n++) {
        if (cp >= end) return -1;
        const uint8_t c = *cp++;
        x = (x << 7) | (c & 0x7fu);
        if (!(c & 0x80u)) { v = x; return 0; }
    }
    return -1;
}
// alphabet (symbol run-length list) -> 256-bit presence mask in 8 words
__device__ __forceinline__ int get_alphabet(const uint8_t *&cp, const uint8_t *end, uint32_t *present) {
    for (int i = 0; i < 8; i++) present[i] = 0;
    if (cp >= end) return -1;
    uint32_t rle = 0, j = *cp++;
    for (int guard = 0; guard < 257; guard++) {
        present[j >> 5] |= 1u << (j & 31);
        if (cp >= end) return -1;
        if (!rle && j + 1 == *cp) {
            j = *cp++;
            if (cp >= end) return -1;
            rle = *cp++;
        } else if (rle) {
            rle--; j++;
            if (j > 255) return -1;
        } else {
            j = *cp++;
        }
        if (j == 0) return 0;
    }
    return -1;
}

// Single-lane order-0 Nx16 decoder (N = 4) for small side streams (compressed order-1 tables).
__device__ int serial_dec_o0_n4(const uint8_t *cp, const uint8_t *end, uint8_t *out, uint32_t out_sz, uint16_t *C /*258 LDS*/) {
    uint32_t present[8];
    if (get_alphabet(cp, end, present)) return -1;
    uint32_t tot = 0;
    // first pass: raw frequencies into C (as F), then convert to cumulative
    for (int j = 0; j < 256; j++) {
        uint32_t f = 0;
        if ((present[j >> 5] >> (j & 31)) & 1u) { if (get_u7(cp, end, f)) return -1; }
        C[j] = (uint16_t)f; tot += f;
        if (tot > 4096u) return -1;
    }
    if (!tot || (tot & (tot - 1))) return -1;
    int sh = 0;
    while ((tot << sh) < 4096u) sh++;
    uint32_t x = 0;
    for (int j = 0; j < 256; j++) { uint32_t f = (uint32_t)C[j] << sh; C[j] = (uint16_t)x; x += f; }
    C[256] = (uint16_t)x;
    if (cp + 16 > end) return -1;
    uint32_t R[4];
    for (int z = 0; z < 4; z++, cp += 4) R[z] = rd32(cp);
    const uint32_t out_end = out_sz & ~3u;
    for (uint32_t i = 0; i < out_sz; i++) {
        const int z = (int)(i & 3u);
        const uint32_t m = R[z] & 4095u;
        uint32_t lo = 0, hi = 256;
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (C[mid] <= m) lo = mid; else hi = mid; }
        out[i] = (uint8_t)lo;
        if (i < out_end) {
            const uint32_t cum = C[lo], f = (uint32_t)C[lo + 1] - cum;
            R[z] = f * (R[z] >> 12) + m - cum;
            if (R[z] < RANS_L) { if (cp + 2 > end) return -1; R[z] = (R[z] << 16) | cp[0] | ((uint32_t)cp[1] << 8); cp += 2; }
        }
    }
    return 0;
}

template <int N>
__global__ __launch_bounds__(WAVES * 64)
void ransnx16_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc,
                            const uint32_t *__restrict__ sel, uint32_t nsel, uint8_t *out, int32_t *status,
                            uint32_t *scratch) {
    // 32-way: ONE stream per wavefront (lanes 32..63 idle).  Two streams side by side ran in lock step: an order-0 and an order-1
    // neighbour cost the sum of both decode chains per step and a short stream waited for a long one; with the chain being LDS
    // latency, the idle lanes cost nothing and twice as many wavefronts hide more of it.
    constexpr int GROUPS = N == 32 ? 1 : 64 / N;
    __shared__ GroupLds lds[WAVES * GROUPS];
    // one pool per stream slot, used either as the order-1 table copy or as the order-0 lookup table
    __shared__ uint32_t pool[N == 32 ? WAVES * GROUPS : 1][N == 32 ? O1_LDS_WORDS : 1];
    // 32-way only: the next 128 renormalisation words of the stream (filled 64 at a time from a register prefetch, so a
    // step never waits on global memory) and the dense numbering of the order-1 contexts (for the bucket table below)
    __shared__ uint32_t ring_s[N == 32 ? WAVES * GROUPS : 1][N == 32 ? 64 : 1];
    __shared__ uint8_t rank_s[WAVES * GROUPS][256];            // alphabet list while the tables are parsed, then (32-way) the context ranks
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & (N - 1);
    const bool idle = lane / N >= GROUPS;                          // lanes beyond the groups in use
    const int grp = idle ? 0 : lane / N;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[(tid >> 6) * GROUPS + grp];
    const unsigned long long gmask = (N == 64 ? ~0ull : ((1ull << N) - 1ull)) << (grp * N);
    const int lane0 = grp * N;

    for (uint32_t k = g_global; __any(k < nsel); k += g_total) {
        const bool have = k < nsel && !idle;
        const uint32_t sidx = have ? sel[k] : 0;
        int err = have ? 0 : 2;
        uint32_t flags = 0, usz = 0, shift = 12, np_words = 0xffffffffu;
        const uint8_t *cp = nullptr, *end = nullptr;
        uint8_t *o = nullptr;
        uint32_t *tabs = nullptr;
        if (have) {
            const hg_stream_desc d = desc[sidx];
            cp = in + d.in_off; end = cp + d.in_len;
            o = out + d.out_off; tabs = scratch + d.scratch_off;
            if (d.reserved & 0x80000000u) {                       // header already parsed by the host planner
                flags = d.reserved & (F_ORDER | F_X32 | F_CAT);
                usz = d.out_len;
                if (((flags & F_X32) ? 32 : 4) != N) err = 1;
            } else if (d.in_len < 1) err = 1;
            else {
                flags = *cp++;
                if (flags & F_NOSZ) usz = d.out_len;
                else if (get_u7(cp, end, usz)) err = 1;
                if (!err && usz != d.out_len) err = 1;
                if (!err && (flags & (F_STRIPE | F_RLE | F_PACK))) err = 3;      // not handled yet
                if (!err && ((flags & F_X32) ? 32 : 4) != N) err = 1;
            }
        }
        const bool cat = !err && (flags & F_CAT);
        if (cat) {
            if (cp + usz > end) err = 1;
            else for (uint32_t i = (uint32_t)sub; i < usz; i += N) o[i] = cp[i];
        }
        const uint32_t order = flags & F_ORDER;
        const bool core = !err && !cat && usz != 0;
        // ---- tables (first lane of the group) ---------------------------------------------------
        if (core && sub == 0) {
            if (order == 0) {
                uint32_t present[8];
                if (get_alphabet(cp, end, present)) err = 1;
                uint32_t tot = 0;
                for (int j = 0; j < 256 && !err; j++) {
                    uint32_t f = 0;
                    if ((present[j >> 5] >> (j & 31)) & 1u) { if (get_u7(cp, end, f)) err = 1; }
                    G.C[j] = (uint16_t)f; tot += f;
                    if (tot > 4096u) err = 1;
                }
                if (!err && (!tot || (tot & (tot - 1)))) err = 1;
                if (!err) {
                    int sh = 0;
                    while ((tot << sh) < 4096u) sh++;
                    uint32_t x = 0;
                    for (int j = 0; j < 256; j++) { uint32_t f = (uint32_t)G.C[j] << sh; G.C[j] = (uint16_t)x; x += f; }
                    G.C[256] = (uint16_t)x;
                }
            } else {
                if (cp >= end) err = 1;
                uint32_t comp = 0;
                if (!err) { shift = *cp >> 4; comp = *cp & 1u; cp++; if (shift != 10 && shift != 12) err = 1; }
                const uint8_t *tp = cp, *tend = end;
                uint32_t np = 512;                                   // words used so far in tabs
                if (!err && comp) {
                    uint32_t ulen = 0, clen = 0;
                    if (get_u7(cp, end, ulen) || get_u7(cp, end, clen) || cp + clen > end || ulen > 262144u) err = 1;
                    else {
                        uint8_t *tb = (uint8_t *)(tabs + 512);
                        if (serial_dec_o0_n4(cp, cp + clen, tb, ulen, G.C)) err = 1;
                        tp = tb; tend = tb + ulen; cp += clen;
                        np = 512 + (ulen + 3) / 4;
                    }
                }
                uint32_t A[8];
                if (!err && get_alphabet(tp, tend, A)) err = 1;
                if (!err) {
                    for (int i = 0; i < 512; i++) tabs[i] = 0;
                    // the alphabet as a list (in the LDS bytes that hold the context ranks later): a row is walked over its TOKENS -- a
                    // frequency, or a zero with the number of further zeros to skip -- not over 256 symbols with a bit test each
                    // (a sparse 256-context table cost 65 k iterations of this single lane: 13 ms)
                    uint8_t *al = rank_s[(tid >> 6) * GROUPS + grp];
                    uint32_t nal = 0;
                    for (int i = 0; i < 256; i++) if ((A[i >> 5] >> (i & 31)) & 1u) al[nal++] = (uint8_t)i;
                    for (uint32_t ci = 0; ci < nal && !err; ci++) {
                        const int i = al[ci];
                        const uint32_t first = np;
                        uint32_t tot = 0, cnt = 0;
                        for (uint32_t k = 0; k < nal; k++) {
                            uint32_t f = 0;
                            if (get_u7(tp, tend, f)) { err = 1; break; }
                            if (f == 0) { if (tp >= tend) { err = 1; break; } k += *tp++; }       // the next *tp symbols are zero as well
                            else { tabs[np++] = (f << 8) | (uint32_t)al[k]; tot += f; cnt++; }
                        }
                        if (err) break;
                        if (tot > (1u << shift) || (tot & (tot - 1))) { err = 1; break; }
                        int sh = 0;
                        while (tot && (tot << sh) < (1u << shift)) sh++;
                        uint32_t x = 0;
                        for (uint32_t e = first; e < np; e++) {              // raw freq -> cumulative
                            const uint32_t f = (tabs[e] >> 8) << sh, s = tabs[e] & 0xffu;
                            tabs[e] = (x << 8) | s; x += f;
                        }
                        tabs[np++] = x << 8;                                 // sentinel = total
                        tabs[i] = first; tabs[256 + i] = cnt;
                    }
                    if (!comp) cp = tp;
                    np_words = np;
                }
            }
        }
        // broadcast parse results from the first lane of the group
        {
            err = __shfl(err, lane0, 64);
            shift = (uint32_t)__shfl((int)shift, lane0, 64);
            const unsigned long long cpv = (unsigned long long)(uintptr_t)cp;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)cpv, lane0, 64);
            const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(cpv >> 32), lane0, 64);
            cp = (const uint8_t *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t *T = tabs;                                    // where the decode loop reads the order-1 tables
        const uint8_t *lut = nullptr;                                // order-0 slot -> symbol
        const uint8_t *o1lut = nullptr;                              // order-1 bucket tables (64, 16 or 4 bytes per context)
        uint32_t o1bb = 0;                                           // log2 of the buckets per context
        const uint8_t *dL = nullptr; const uint32_t *dD = nullptr;   // order-1 dense form (small alphabets)
        uint32_t drank0 = 0;
        if constexpr (N == 32) {
            uint32_t *P = pool[(tid >> 6) * GROUPS + grp];
            const uint32_t npw = (uint32_t)__shfl((int)np_words, lane0, 64);
            if (core && !err && order && npw <= O1_LDS_WORDS) {
                for (uint32_t i = (uint32_t)sub; i < npw; i += N) P[i] = tabs[i];
                T = P;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // bucket table: for every context, which list entry holds slot b * (range / 64) -- replaces the binary
                // search (6-7 dependent LDS reads) by one read plus a short forward scan
                uint8_t *rk = rank_s[(tid >> 6) * GROUPS + grp];
                uint32_t nctx = 0;
                if (sub == 0) for (int i = 0; i < 256; i++) { rk[i] = (uint8_t)nctx; if (P[256 + i]) nctx++; }
                nctx = (uint32_t)__shfl((int)nctx, lane0, 64);
                // Small alphabets (<= 16 contexts of <= 16 symbols: binned qualities, bases, flags): a DENSE form that costs two
                // dependent LDS reads per symbol instead of four -- L256[context rank][slot >> (shift - 8)] = list index of the slot's
                // bucket, DD[context rank][index] = cumulative << 12 | symbol << 4 | rank of the symbol as the next context (bit 25:
                // that symbol never is a context).  The decode loop is a chain of such reads; with a few wavefronts per CU nothing
                // hides them.
                uint32_t big = 0;
                for (uint32_t i = (uint32_t)sub; i < 256; i += N) if (P[256 + i] > 16u) big = 1;
                big = (__ballot(big != 0) & gmask) ? 1u : 0u;
                if (!big && nctx <= 16u && npw + 1024u + 16u * 17u <= O1_LDS_WORDS && shift >= 8u) {
                    uint8_t *L = (uint8_t *)(P + npw);
                    uint32_t *DDw = P + npw + 1024u;
                    const uint32_t sh8 = shift - 8u;
                    for (uint32_t i = (uint32_t)sub; i < 256; i += N) {
                        const uint32_t cnt = P[256 + i], base = P[i];
                        if (!cnt) continue;
                        const uint32_t r = rk[i];
                        for (uint32_t k = 0; k <= cnt; k++) {
                            const uint32_t e = P[base + k], sy = e & 0xffu;
                            const uint32_t nx = k < cnt ? (P[256 + sy] ? (uint32_t)rk[sy] : (1u << 25)) : 0u;
                            DDw[r * 17u + k] = ((e >> 8) << 12) | (sy << 4) | nx;
                        }
                        uint32_t k = 0;
                        for (uint32_t bkt = 0; bkt < 256; bkt++) {
                            const uint32_t sl = bkt << sh8;
                            while (k + 1 < cnt && (P[base + k + 1] >> 8) <= sl) k++;
                            L[r * 256u + bkt] = (uint8_t)k;
                        }
                    }
                    dL = L; dD = DDw;
                    drank0 = P[256] ? (uint32_t)rk[0] : 0xffffffffu;            // the states start in context 0
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else
                // 64 buckets per context when they fit beside the lists, else 16, else 4 (sparse tables with many contexts)
                if (npw + nctx * 16u <= O1_LDS_WORDS) o1bb = 6; else if (npw + nctx * 4u <= O1_LDS_WORDS) o1bb = 4; else if (npw + nctx <= O1_LDS_WORDS) o1bb = 2;
                if (o1bb) {
                    const uint32_t sh6 = shift - o1bb, nb = 1u << o1bb;
                    for (uint32_t i = (uint32_t)sub; i < 256; i += N) {
                        const uint32_t cnt = P[256 + i], base = P[i];
                        if (!cnt) continue;
                        uint8_t *l8 = (uint8_t *)(P + npw) + nb * rk[i];
                        uint32_t k = 0;
                        for (uint32_t bkt = 0; bkt < nb; bkt++) {
                            const uint32_t sl = bkt << sh6;
                            while (k + 1 < cnt && (P[base + k + 1] >> 8) <= sl) k++;
                            l8[bkt] = (uint8_t)k;
                        }
                    }
                    o1lut = (const uint8_t *)(P + npw);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // one word per context for the decode loop: list start | bucket-table number << 13 | entries << 21
                    for (uint32_t i = (uint32_t)sub; i < 256; i += N) P[i] = P[i] | ((uint32_t)rk[i] << 13) | (P[256 + i] << 21);
                }
            } else if (core && !err && !order) {
                uint8_t *L8 = (uint8_t *)P;
                // lane l fills the slots of symbols l, l+32, ...
                for (uint32_t sy = (uint32_t)sub; sy < 256; sy += N) { const uint32_t a = G.C[sy], b = G.C[sy + 1]; for (uint32_t q = a; q < b; q++) L8[q] = (uint8_t)sy; }
                lut = L8;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const bool live = core && !err;
        uint32_t R = 0;
        if (live) {
            if (cp + 4 * N > end) err = 1;
            else { R = rd32(cp + 4 * sub); cp += 4 * N; }
        }
        // 32-way: renormalisation words come from the LDS ring, topped up from a register prefetch
        uint32_t *ring = ring_s[N == 32 ? (tid >> 6) * GROUPS + grp : 0];
        uint32_t wpos = 0, wfill = 0, wavail = 0, pre = 0;
        const uint8_t *wbase = cp;
        auto load_chunk = [&](uint32_t c) -> uint32_t {               // words 64c + 2*sub, +1 of the stream (0 past the end)
            const uint8_t *p = wbase + 2u * (64u * c + 2u * (uint32_t)sub);
            uint32_t v = 0;
            if (p + 4 <= end) v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            else for (int q = 0; q < 4; q++) if (p + q < end) v |= (uint32_t)p[q] << (8 * q);
            return v;
        };
        if constexpr (N == 32) {
            if (live && !err) {
                wavail = (uint32_t)((end - wbase) >> 1);
                ring[sub] = load_chunk(0);
                pre = load_chunk(1);
                wfill = 64;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t mask = (1u << shift) - 1u;
        const uint32_t per = usz / N;
        uint32_t pos = order == 0 ? (uint32_t)sub : (uint32_t)sub * per, ctx = 0;
        uint32_t rctx = drank0;                                      // dense form: rank of the current context
        const uint32_t steps = per, rem = usz - per * N;
        const uint32_t max_steps = (live && !err) ? steps + (order ? rem : 0u) : 0u;
        for (uint32_t it = 0; __any(it < max_steps); it++) {
            const bool act = live && !err && it < max_steps;
            const bool mine = act && (it < steps || (order && sub == N - 1));
            uint32_t need = 0;
            if (mine) {
                const uint32_t m = R & mask;
                uint32_t sym = 0, cum = 0, f = 1;
                if (order == 0) {
                    uint32_t lo = 0, hi = 256;
                    if (lut) lo = lut[m];
                    else while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (G.C[mid] <= m) lo = mid; else hi = mid; }
                    sym = lo; cum = G.C[lo]; f = (uint32_t)G.C[lo + 1] - cum;
                } else {
                    if (N == 32 && dD) {
                        if (rctx > 15u) err = 1;                      // context never seen by the encoder
                        else {
                            uint32_t kb = dL[rctx * 256u + (m >> (shift - 8u))];
                            const uint32_t *row = dD + rctx * 17u;
                            uint32_t e = row[kb], e1 = row[kb + 1];
                            while (((e1 >> 12) & 0x1fffu) <= m) { kb++; e = e1; e1 = row[kb + 1]; }   // the list ends with the total > m
                            sym = (e >> 4) & 0xffu; cum = (e >> 12) & 0x1fffu; f = ((e1 >> 12) & 0x1fffu) - cum;
                            rctx = (e >> 25) ? 0xffffu : (e & 15u);
                        }
                    } else if (N == 32 && o1lut) {
                        const uint32_t info = T[ctx], base = info & 0x1fffu;
                        if ((info >> 21) == 0) err = 1;               // context never seen by the encoder
                        else {
                            const uint32_t bk = m >> (shift - o1bb), bofs = ((info >> 13) & 0xffu) << o1bb;
                            uint32_t lo = o1lut[bofs + bk];
                            if (o1bb < 6u) {                          // few, wide buckets: binary search between this bucket's start and the next one's
                                uint32_t hi = bk + 1u < (1u << o1bb) ? (uint32_t)o1lut[bofs + bk + 1u] + 1u : (info >> 21);
                                while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if ((T[base + mid] >> 8) <= m) lo = mid; else hi = mid; }
                            }
                            uint32_t e = T[base + lo], e1 = T[base + lo + 1];
                            while ((e1 >> 8) <= m) { lo++; e = e1; e1 = T[base + lo + 1]; }   // the list ends with (range << 8) > m
                            sym = e & 0xffu; cum = e >> 8; f = (e1 >> 8) - cum;
                        }
                    } else if (const uint32_t n = T[256 + ctx], base = T[ctx]; n == 0 || (T[base + n] >> 8) <= m) err = 1;
                    else {
                        uint32_t lo = 0, hi = n;
                        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if ((T[base + mid] >> 8) <= m) lo = mid; else hi = mid; }
                        const uint32_t e = T[base + lo];
                        sym = e & 0xffu; cum = e >> 8; f = (T[base + lo + 1] >> 8) - cum;
                    }
                }
                if (!err) {
                    o[pos] = (uint8_t)sym;
                    pos += order == 0 ? (uint32_t)N : 1u;
                    ctx = sym;
                    R = f * (R >> shift) + m - cum;
                    need = R < RANS_L ? 1u : 0u;
                }
            }
            // the lanes that renormalise take consecutive 16-bit words in lane order
            const unsigned long long b = __ballot(need != 0) & gmask;
            const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            const uint32_t tot = (uint32_t)__popcll(b);
            if constexpr (N == 32) {
                if (need) {
                    const uint32_t k = wpos + before;
                    if (k >= wavail) err = 1;
                    else R = (R << 16) | (uint32_t)((const uint16_t *)ring)[k & 127u];
                }
                wpos += tot;
                if (act && wfill - wpos < 32u) {                      // whole group takes this branch together
                    ring[((wfill >> 1) + (uint32_t)sub) & 63u] = pre;
                    wfill += 64;
                    pre = load_chunk(wfill >> 6);
                }
            } else {
                if (need) {
                    const uint8_t *w = cp + 2u * before;
                    if (w + 2 > end) err = 1;
                    else R = (R << 16) | (uint32_t)w[0] | ((uint32_t)w[1] << 8);
                }
                cp += 2u * tot;
            }
            err = (__ballot(err == 1) & gmask) ? 1 : err;
        }
        // order-0 tail: states 0..rem-1 give one more symbol each, without update
        if (live && !err && order == 0 && (uint32_t)sub < rem) {
            const uint32_t m = R & mask;
            uint32_t lo = 0, hi = 256;
            if (lut) lo = lut[m];
            else while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (G.C[mid] <= m) lo = mid; else hi = mid; }
            o[per * N + sub] = (uint8_t)lo;
        }
        err = (__ballot(err == 1) & gmask) ? 1 : err;
        if (have && sub == 0) status[sidx] = err == 0 ? 0 : (err == 3 ? HG_BLOCK_EUNSUPPORTED : -1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace hgn

namespace hg {
int launch_ransnx16_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4,
                           size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out, int32_t *d_status,
                           uint32_t *d_scratch, hipStream_t s) {
    const size_t maxw = (size_t)ctx->cus * 8;
    const bool side = n4 != 0 && n32 != 0;                  // both variants present: overlap them
    hipStream_t s2 = side ? fork_side(ctx, s) : s;
    if (n4) {
        size_t wgs = (n4 + hgn::WAVES * 16 - 1) / (hgn::WAVES * 16);
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL(hgn::ransnx16_decode_kernel<4>, dim3((unsigned)wgs), dim3(hgn::WAVES * 64), 0, s,
                           (const uint8_t *)d_in, d_desc, d_sel4, (uint32_t)n4, (uint8_t *)d_out, d_status, d_scratch);
    }
    if (n32) {
        size_t wgs = (n32 + hgn::WAVES - 1) / hgn::WAVES;                  // one 32-way stream per wavefront
        if (wgs > maxw * 2) wgs = maxw * 2;
        hipLaunchKernelGGL(hgn::ransnx16_decode_kernel<32>, dim3((unsigned)wgs), dim3(hgn::WAVES * 64), 0, s2,
                           (const uint8_t *)d_in, d_desc, d_sel32, (uint32_t)n32, (uint8_t *)d_out, d_status, d_scratch);
        if (side) join_side(ctx, s);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

// ransnx16_dev.h -- device-side pieces shared by the rANS Nx16 decoders (ransnx16.hip: 32-way streams and small 4-way streams, sixteen per wavefront;
// rans4x16_big.hip: long 4-way streams, one per wavefront): header fields, frequency-table parsing (one lane), the LDS forms of an order-1 table
// (built by the COOP lanes that share a stream) and the symbol lookup over them.  Format per oracle/ransnx16_oracle.c (PARITY UNPINNED: htscodecs absent).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"

namespace hgn {

constexpr uint32_t RANS_L = 1u << 15;
enum { F_ORDER = 1, F_X32 = 4, F_STRIPE = 8, F_NOSZ = 16, F_CAT = 32, F_RLE = 64, F_PACK = 128 };
// 4-way streams of at least this many plain bytes go to rans4x16_big.hip (one per wavefront, chain-optimised); shorter ones share a wavefront
constexpr uint32_t BIG4_MIN = 16384;

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
// which of the two 4-way kernels decodes this stream (both walk the same list): the flags byte is in the descriptor (pre-parsed) or first in the stream
__device__ __forceinline__ bool big4_takes(const uint8_t *in, const hg_stream_desc &d) {
    if (d.out_len < BIG4_MIN) return false;
    const uint32_t fl = (d.reserved & 0x80000000u) ? d.reserved : (d.in_len ? in[d.in_off] : (uint32_t)F_CAT);
    return !(fl & (F_CAT | F_X32));
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

// Order-0 frequency table -> C[0..256] cumulative, scaled to 4096 (one lane).  0 / 1 (malformed)
__device__ inline int parse_o0(const uint8_t *&cp, const uint8_t *end, uint16_t *C) {
    uint32_t present[8];
    if (get_alphabet(cp, end, present)) return 1;
    uint32_t tot = 0;
    for (int j = 0; j < 256; j++) {
        uint32_t f = 0;
        if ((present[j >> 5] >> (j & 31)) & 1u) { if (get_u7(cp, end, f)) return 1; }
        C[j] = (uint16_t)f; tot += f;
        if (tot > 4096u) return 1;
    }
    if (!tot || (tot & (tot - 1))) return 1;
    int sh = 0;
    while ((tot << sh) < 4096u) sh++;
    uint32_t x = 0;
    for (int j = 0; j < 256; j++) { const uint32_t f = (uint32_t)C[j] << sh; C[j] = (uint16_t)x; x += f; }
    C[256] = (uint16_t)x;
    return 0;
}

// Single-lane order-0 Nx16 decoder (N = 4) for small side streams (compressed order-1 tables).
__device__ inline int serial_dec_o0_n4(const uint8_t *cp, const uint8_t *end, uint8_t *out, uint32_t out_sz, uint16_t *C /*258 LDS*/) {
    if (parse_o0(cp, end, C)) return -1;
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

// Order-1 frequency tables (one lane) -> sparse per-context lists in tabs (global scratch): tabs[c] = first entry of context c's list, tabs[256 + c] = its
// length, entries from word 512 on as cumulative << 8 | symbol, each list closed by total << 8.  C (258 x u16) and al (256 bytes) are LDS work areas.
// cp is left behind the tables.  0 / 1 (malformed)
__device__ inline int parse_o1(const uint8_t *&cp, const uint8_t *end, uint32_t *tabs, uint16_t *C, uint8_t *al, uint32_t &shift, uint32_t &np_words) {
    if (cp >= end) return 1;
    shift = *cp >> 4; const uint32_t comp = *cp & 1u; cp++;
    if (shift != 10 && shift != 12) return 1;
    const uint8_t *tp = cp, *tend = end;
    uint32_t np = 512;                                               // words used so far in tabs
    if (comp) {
        uint32_t ulen = 0, clen = 0;
        if (get_u7(cp, end, ulen) || get_u7(cp, end, clen) || cp + clen > end || ulen > 262144u) return 1;
        uint8_t *tb = (uint8_t *)(tabs + 512);
        if (serial_dec_o0_n4(cp, cp + clen, tb, ulen, C)) return 1;
        tp = tb; tend = tb + ulen; cp += clen;
        np = 512 + (ulen + 3) / 4;
    }
    uint32_t A[8];
    if (get_alphabet(tp, tend, A)) return 1;
    for (int i = 0; i < 512; i++) tabs[i] = 0;
    // the alphabet as a list: a row is walked over its TOKENS -- a frequency, or a zero with the number of further zeros to skip -- not over 256 symbols
    // with a bit test each (a sparse 256-context table cost 65 k iterations of this single lane: 13 ms)
    uint32_t nal = 0;
    for (int i = 0; i < 256; i++) if ((A[i >> 5] >> (i & 31)) & 1u) al[nal++] = (uint8_t)i;
    for (uint32_t ci = 0; ci < nal; ci++) {
        const int i = al[ci];
        const uint32_t first = np;
        uint32_t tot = 0, cnt = 0;
        for (uint32_t k = 0; k < nal; k++) {
            uint32_t f = 0;
            if (get_u7(tp, tend, f)) return 1;
            if (f == 0) { if (tp >= tend) return 1; k += *tp++; }       // the next *tp symbols are zero as well
            else { tabs[np++] = (f << 8) | (uint32_t)al[k]; tot += f; cnt++; }
        }
        if (tot > (1u << shift) || (tot & (tot - 1))) return 1;
        int sh = 0;
        while (tot && (tot << sh) < (1u << shift)) sh++;
        uint32_t x = 0;
        for (uint32_t e = first; e < np; e++) {                      // raw freq -> cumulative
            const uint32_t f = (tabs[e] >> 8) << sh, s = tabs[e] & 0xffu;
            tabs[e] = (x << 8) | s; x += f;
        }
        tabs[np++] = x << 8;                                         // sentinel = total
        tabs[i] = first; tabs[256 + i] = cnt;
    }
    if (!comp) cp = tp;
    np_words = np;
    return 0;
}

// Where the decode loop finds an order-1 table.  A symbol lookup is a chain of DEPENDENT reads, so the forms are shaped for few of them:
//   O1_DENSE : <= 16 contexts of <= 16 symbols (binned qualities, bases, flags) -- L[context rank][slot >> (shift - 8)] = list index of the slot's bucket
//              (bytes at pool word l_off), D[context rank][index] = cumulative << 12 | symbol << 4 | rank of the symbol as the next context (bit 25: that
//              symbol never is a context; pool word d_off): two reads per symbol;
//   O1_BUCKET: lists in the pool, P[context] = list start | bucket-table number << 13 | entries << 21, 2^bb buckets per context (bytes at pool word
//              l_off) -> one read + a short forward scan (or a short binary search for few, wide buckets);
//   O1_LISTS_LDS / O1_LISTS_GLOBAL: binary search in the lists (pool copy when it fits, else global scratch).
// The pool pointer and the global pointer are kept apart (and the form is a number, not a pointer) so that every access has a known address space: a
// generic pointer makes the loads FLAT instructions, which wait on the vector-memory counter as well -- i.e. on the byte stores of the decoded symbols.
//   O1_DENSE1: the dense case again when the pool has room for a word per (context, bucket) -- cumulative | (frequency - 1) << 13 | list index << 25 | bit 29:
//              the whole bucket lies inside one symbol's range ("pure").  For a pure bucket ONE read feeds the state update; the row entry (symbol, next
//              context) is read beside it, off the state's chain.  Impure buckets scan the row.  Only rans4x16_big.hip asks for it (ONE_READ) and walks
//              it (dense1_step there); lookup_o1 does not know it.
enum { O1_LISTS_GLOBAL = 0, O1_LISTS_LDS = 1, O1_BUCKET = 2, O1_DENSE = 3, O1_DENSE1 = 4 };
struct O1Forms { uint32_t form, bb, l_off, d_off, drank0; };

__device__ __forceinline__ void group_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The COOP lanes sharing a stream (cl = 0 .. COOP-1, the first of them is wave lane lane0, gmask = their bits in a ballot) copy the lists to the pool P
// (pool_words words) when they fit and build the best form beside them.  rk: 256 LDS bytes.  Must be called by all COOP lanes.
template <int COOP, bool ONE_READ = false>
__device__ __forceinline__ void build_o1_forms(uint32_t *P, uint32_t pool_words, const uint32_t *tabs, uint32_t npw, uint32_t shift, uint8_t *rk, int cl,
                                               unsigned long long gmask, int lane0, O1Forms &F) {
    F.form = O1_LISTS_GLOBAL; F.bb = 0; F.l_off = 0; F.d_off = 0; F.drank0 = 0;
    if (npw > pool_words) return;
    for (uint32_t i = (uint32_t)cl; i < npw; i += COOP) P[i] = tabs[i];
    F.form = O1_LISTS_LDS;
    group_sync();
    uint32_t nctx = 0;
    if (cl == 0) for (int i = 0; i < 256; i++) { rk[i] = (uint8_t)nctx; if (P[256 + i]) nctx++; }
    nctx = (uint32_t)__shfl((int)nctx, lane0, 64);
    group_sync();
    uint32_t big = 0;
    for (uint32_t i = (uint32_t)cl; i < 256; i += COOP) if (P[256 + i] > 16u) big = 1;
    big = (__ballot(big != 0) & gmask) ? 1u : 0u;
    if (ONE_READ && !big && nctx <= 16u && npw + (nctx + 1u) * (256u + 17u) <= pool_words && shift >= 8u) {
        // W1[rank][bucket] = cumulative | (frequency - 1) << 13 | list index << 25 | pure << 29 | invalid << 30; E[rank][index] = symbol | next rank << 8 |
        // cumulative << 16.  Rank nctx is a DUMMY context (every bucket "invalid", its symbols lead back to it): a symbol that never is a context sends
        // the state there, so the decode loop needs no test on the rank -- it ORs the words it reads and looks at bit 30 once, at the end.
        uint32_t *W1 = P + npw, *E = P + npw + (nctx + 1u) * 256u;
        const uint32_t sh8 = shift - 8u;
        for (uint32_t i = (uint32_t)cl; i < 256; i += COOP) {
            const uint32_t cnt = P[256 + i], base = P[i];
            if (!cnt) continue;
            const uint32_t r = rk[i];
            for (uint32_t k = 0; k <= cnt; k++) {
                const uint32_t e = P[base + k], sy = e & 0xffu;
                const uint32_t nx = k < cnt ? (P[256 + sy] ? (uint32_t)rk[sy] : nctx) : nctx;
                E[r * 17u + k] = sy | (nx << 8) | ((e >> 8) << 16);
            }
            uint32_t k = 0;
            for (uint32_t bkt = 0; bkt < 256; bkt++) {
                const uint32_t sl = bkt << sh8;
                while (k + 1 < cnt && (P[base + k + 1] >> 8) <= sl) k++;
                const uint32_t c0 = P[base + k] >> 8, c1 = P[base + k + 1] >> 8;          // the list ends with the total
                const uint32_t pure = c1 >= ((bkt + 1u) << sh8) ? 1u : 0u;
                W1[r * 256u + bkt] = c0 | ((c1 - c0 - 1u) << 13) | (k << 25) | (pure << 29);
            }
        }
        for (uint32_t i = (uint32_t)cl; i < 256; i += COOP) W1[nctx * 256u + i] = (1u << 30) | (1u << 29);
        for (uint32_t i = (uint32_t)cl; i < 17; i += COOP) E[nctx * 17u + i] = (nctx << 8) | ((i ? 1u : 0u) << 16);
        F.form = O1_DENSE1; F.l_off = npw; F.d_off = npw + (nctx + 1u) * 256u;
        F.drank0 = P[256] ? (uint32_t)rk[0] : nctx;                  // the states start in context 0
        group_sync();
        return;
    }
    if (!big && nctx <= 16u && npw + 1024u + 16u * 17u <= pool_words && shift >= 8u) {
        uint8_t *L = (uint8_t *)(P + npw);
        uint32_t *DDw = P + npw + 1024u;
        const uint32_t sh8 = shift - 8u;
        for (uint32_t i = (uint32_t)cl; i < 256; i += COOP) {
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
        F.form = O1_DENSE; F.l_off = npw; F.d_off = npw + 1024u;
        F.drank0 = P[256] ? (uint32_t)rk[0] : 0xffffffffu;           // the states start in context 0
        group_sync();
        return;
    }
    // 64 buckets per context when they fit beside the lists, else 16, else 4 (sparse tables with many contexts)
    uint32_t o1bb = 0;
    if (npw + nctx * 16u <= pool_words) o1bb = 6; else if (npw + nctx * 4u <= pool_words) o1bb = 4; else if (npw + nctx <= pool_words) o1bb = 2;
    if (o1bb) {
        const uint32_t sh6 = shift - o1bb, nb = 1u << o1bb;
        for (uint32_t i = (uint32_t)cl; i < 256; i += COOP) {
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
        F.form = O1_BUCKET; F.l_off = npw; F.bb = o1bb;
        group_sync();
        // one word per context for the decode loop: list start | bucket-table number << 13 | entries << 21
        for (uint32_t i = (uint32_t)cl; i < 256; i += COOP) P[i] = P[i] | ((uint32_t)rk[i] << 13) | (P[256 + i] << 21);
        group_sync();
    }
}

// binary search of slot m in the list of context ctx; T = the lists (pool or global scratch)
template <class Ptr>
__device__ __forceinline__ bool lookup_lists(Ptr T, uint32_t ctx, uint32_t m, uint32_t &sym, uint32_t &cum, uint32_t &f) {
    const uint32_t n = T[256 + ctx], base = T[ctx];
    if (n == 0 || (T[base + n] >> 8) <= m) return false;
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((T[base + mid] >> 8) <= m) lo = mid; else hi = mid; }
    const uint32_t e = T[base + lo];
    sym = e & 0xffu; cum = e >> 8; f = (T[base + lo + 1] >> 8) - cum;
    return true;
}

// slot m in context ctx (dense form: context rank rctx, updated here) -> symbol, its cumulative frequency and frequency; false = a context / slot the
// encoder never produced (malformed stream).  P = the stream's pool (LDS), tabs = its lists in global scratch.
__device__ __forceinline__ bool lookup_o1(const O1Forms &F, const uint32_t *P, const uint32_t *tabs, uint32_t ctx, uint32_t &rctx, uint32_t m, uint32_t shift,
                                          uint32_t &sym, uint32_t &cum, uint32_t &f) {
    if (F.form == O1_DENSE) {
        if (rctx > 15u) return false;
        uint32_t kb = ((const uint8_t *)(P + F.l_off))[rctx * 256u + (m >> (shift - 8u))];
        const uint32_t *row = P + F.d_off + rctx * 17u;
        uint32_t e = row[kb], e1 = row[kb + 1];
        while (((e1 >> 12) & 0x1fffu) <= m) { kb++; e = e1; e1 = row[kb + 1]; }   // the list ends with the total > m
        sym = (e >> 4) & 0xffu; cum = (e >> 12) & 0x1fffu; f = ((e1 >> 12) & 0x1fffu) - cum;
        rctx = (e >> 25) ? 0xffffu : (e & 15u);
        return true;
    }
    if (F.form == O1_BUCKET) {
        const uint8_t *lut = (const uint8_t *)(P + F.l_off);
        const uint32_t info = P[ctx], base = info & 0x1fffu;
        if ((info >> 21) == 0) return false;
        const uint32_t bk = m >> (shift - F.bb), bofs = ((info >> 13) & 0xffu) << F.bb;
        uint32_t lo = lut[bofs + bk];
        if (F.bb < 6u) {                                             // few, wide buckets: binary search between this bucket's start and the next one's
            uint32_t hi = bk + 1u < (1u << F.bb) ? (uint32_t)lut[bofs + bk + 1u] + 1u : (info >> 21);
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if ((P[base + mid] >> 8) <= m) lo = mid; else hi = mid; }
        }
        uint32_t e = P[base + lo], e1 = P[base + lo + 1];
        while ((e1 >> 8) <= m) { lo++; e = e1; e1 = P[base + lo + 1]; }   // the list ends with (range << 8) > m
        sym = e & 0xffu; cum = e >> 8; f = (e1 >> 8) - cum;
        return true;
    }
    if (F.form == O1_LISTS_LDS) return lookup_lists(P, ctx, m, sym, cum, f);
    return lookup_lists(tabs, ctx, m, sym, cum, f);
}

}  // namespace hgn

// deflate_huff_wg.h -- the Huffman phase of the deflate encoder as a WORKGROUP-collective routine.
//
// deflate_huff.h builds the two code-length sets and the RFC 1951 dynamic header on ONE lane (Moffat-Katajainen
// lengths + Kraft repair, a byte-wise run-length scan, a bit sink): 16 % of a block's time at level 6 and 31 % at level 1
// sat on that lane while 255 others waited.  Here every step is data parallel over the <= 286 + 30 + 19 symbols:
//
//   * code lengths by PACKAGE-MERGE (Larmore & Hirschberg 1990): level l's list is the merge of the sorted leaves with
//     the pairs ("packages") of level l-1's list.  Both inputs are sorted, so every item finds its place with one binary
//     search in the other list -- one thread per item, one barrier per level, 15 levels.  A leaf's code length is the
//     number of levels whose chosen prefix contains it; the prefixes come from a 15-step walk over per-level
//     "leaves among the first k items" tables.  The result is the OPTIMAL length-limited code (the serial version's
//     Kraft repair is a heuristic), and the litlen and distance trees share the same 14 barriers;
//   * canonical codes: histogram of the lengths by LDS atomics, per-symbol count of earlier symbols of the same length;
//   * the header's run-length coding: every run start computes its tokens in closed form (the greedy scan of
//     deflate_huff.h: write_dynamic_header re-stated per run), offsets from a prefix sum, the 19-symbol code by the same
//     package-merge with 7 levels;
//   * the header leaves as a list of (value, bit count) items for the kernel's parallel bit packer.
//
// Replaces what zlib's trees.c (build_tree / gen_bitlen / send_all_trees) or libdeflate do behind bgzf_compress
// (reference bgzf.c:561-683); written from RFC 1951 3.2.2 / 3.2.7 and the package-merge paper, not from those sources.
//
// The code compiles for gfx950 (barrier = __syncthreads, LDS atomics) and for the host, where tests/native/huffwg_host.cpp
// runs it on NT real threads with a pthread barrier (tests/test_deflate_huff.py) against zlib's decoder and an exact
// length-limited optimum.
#pragma once
#include <stdint.h>
#include "deflate_huff.h"

#if defined(__HIPCC__)
#define HGW_SYNC() __syncthreads()
#define HGW_ADD(p, v) atomicAdd((p), (v))
#define HGW_MAX(p, v) atomicMax((p), (v))
#define HGW_FN __device__ __forceinline__
#else
extern "C" void hgw_host_barrier(void);
#define HGW_SYNC() hgw_host_barrier()
#define HGW_ADD(p, v) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#define HGW_MAX(p, v) do { uint32_t o_ = __atomic_load_n((p), __ATOMIC_RELAXED); while (o_ < (uint32_t)(v) && !__atomic_compare_exchange_n((p), &o_, (uint32_t)(v), 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { } } while (0)
#define HGW_FN inline
#endif

namespace hgdef {

constexpr int PM_MAXA = 288;           // leaves of the big tree (litlen: 286, code-length code: 19)
constexpr int PM_MAXB = 32;            // leaves of the small tree (distances: 30)
constexpr int PM_LEVELS = 15;
constexpr int HDR_ITEMS = 1 + 19 + 320;

struct HuffWG {                        // ~27 KiB of scratch; the kernel overlays it on the staged input once matching is done
    uint32_t mA[2][2 * PM_MAXA];       // merged list of the previous / the current level (weights)
    uint32_t mB[2][2 * PM_MAXB];
    uint16_t cA[PM_LEVELS + 1][2 * PM_MAXA + 2];   // cA[l][k] = leaves among the first k items of level l's list (l >= 2)
    uint16_t cB[PM_LEVELS + 1][2 * PM_MAXB + 2];
    uint32_t wA[PM_MAXA], wB[PM_MAXB]; // leaf weights, ascending by (frequency, symbol)
    uint16_t oA[PM_MAXA], oB[PM_MAXB]; // ... and their symbols
    uint32_t hist[2][16], nxt[2][16];  // codes per length, first code per length
    uint8_t seq[320];                  // litlen lengths followed by distance lengths (hlit + hdist of them)
    uint8_t ntok[320];                 // tokens a run start emits (0 elsewhere)
    uint8_t cl_sym[320], cl_ext[320];  // the run-length coded sequence
    uint32_t clf[20];                  // frequencies of the code-length symbols
    uint8_t cl_len[20]; uint16_t cl_code[20];
    uint32_t nA, nB, hlit, hdist, m, hclen, hdr_bits, pad;
    uint16_t item_v[HDR_ITEMS]; uint8_t item_n[HDR_ITEMS];      // the header as (value, bit count) items, in order
    uint32_t nitems;
};

// ---- package-merge ---------------------------------------------------------------------------------------------------
// One level of one tree: Mp (mp items, ascending) is the previous level's list, w[0..n) the leaves.  Writes the merge of the
// leaves with the mp / 2 packages into Mn and the leaf counts of its prefixes into cnt.  Ties: leaf before package.
template <int NT>
HGW_FN void pm_level(const uint32_t *w, int n, const uint32_t *Mp, int mp, uint32_t *Mn, uint16_t *cnt, int tid) {
    const int np = mp >> 1;
    for (int i = tid; i < n; i += NT) {
        const uint32_t wi = w[i];
        int lo = 0, hi = np;                                   // packages strictly lighter than the leaf
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (Mp[2 * mid] + Mp[2 * mid + 1] < wi) lo = mid + 1; else hi = mid; }
        const int r = i + lo;
        Mn[r] = wi; cnt[r + 1] = (uint16_t)(i + 1);
    }
    for (int j = tid; j < np; j += NT) {
        const uint32_t pj = Mp[2 * j] + Mp[2 * j + 1];
        int lo = 0, hi = n;                                    // leaves not heavier than the package
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (w[mid] <= pj) lo = mid + 1; else hi = mid; }
        const int r = j + lo;
        Mn[r] = pj; cnt[r + 1] = (uint16_t)lo;
    }
    if (tid == 0) cnt[0] = 0;
}

// Code lengths of leaf i (rank order) for a tree of n >= 2 leaves after `levels` levels: every thread walks the levels
// (broadcast reads) and counts the levels whose prefix holds its leaf.
template <int LEVELS>
HGW_FN void pm_prefixes(const uint16_t (*cnt)[2 * PM_MAXA + 2], int n, int *c /* [LEVELS + 1] */) {
    int k = 2 * n - 2;
#pragma unroll
    for (int l = LEVELS; l >= 2; l--) { const int cl = cnt[l][k]; c[l] = cl; k = 2 * (k - cl); }
    c[1] = k;
}
template <int LEVELS>
HGW_FN void pm_prefixes_b(const uint16_t (*cnt)[2 * PM_MAXB + 2], int n, int *c) {
    int k = 2 * n - 2;
#pragma unroll
    for (int l = LEVELS; l >= 2; l--) { const int cl = cnt[l][k]; c[l] = cl; k = 2 * (k - cl); }
    c[1] = k;
}

// Length-limited code lengths for two alphabets at once (B may be empty: nsymB = 0).  freq* are the symbol frequencies,
// len* come back complete (a two-code set when fewer than two symbols occur, as deflate_huff.h: finish_lengths).
// Collective over NT threads; ends with a barrier.
template <int NT, int LEVELS>
HGW_FN void wg_code_lengths(HuffWG &W, const uint32_t *freqA, int nsymA, uint8_t *lenA, const uint32_t *freqB, int nsymB, uint8_t *lenB, int tid) {
    if (tid == 0) { W.nA = 0; W.nB = 0; }
    for (int i = tid; i < nsymA; i += NT) lenA[i] = 0;
    for (int i = tid; i < nsymB; i += NT) lenB[i] = 0;
    HGW_SYNC();
    // rank sort, one thread per symbol (the B symbols on the upper half of the workgroup)
    for (int i = tid; i < nsymA; i += NT) {
        const uint32_t f = freqA[i];
        if (f) { const int r = rank_symbol(freqA, nsymA, i); W.oA[r] = (uint16_t)i; W.wA[r] = f; HGW_ADD(&W.nA, 1u); }
    }
    for (int i = tid - NT / 2; i >= 0 && i < nsymB; i += NT) {
        const uint32_t f = freqB[i];
        if (f) { const int r = rank_symbol(freqB, nsymB, i); W.oB[r] = (uint16_t)i; W.wB[r] = f; HGW_ADD(&W.nB, 1u); }
    }
    HGW_SYNC();
    const int nA = (int)W.nA, nB = (int)W.nB;
    int mAp = nA, mBp = nB;                                    // level 1's list = the leaves themselves
    for (int l = 2; l <= LEVELS; l++) {
        const int cur = l & 1;
        if (nA >= 2) pm_level<NT>(W.wA, nA, l == 2 ? W.wA : W.mA[cur ^ 1], mAp, W.mA[cur], W.cA[l], tid);
        if (nB >= 2) pm_level<NT>(W.wB, nB, l == 2 ? W.wB : W.mB[cur ^ 1], mBp, W.mB[cur], W.cB[l], tid);
        mAp = nA + (mAp >> 1); mBp = nB + (mBp >> 1);
        HGW_SYNC();
    }
    int c[LEVELS + 1];
    if (nA >= 2) {
        pm_prefixes<LEVELS>(W.cA, nA, c);
        for (int i = tid; i < nA; i += NT) {
            int len = 0;
#pragma unroll
            for (int l = 1; l <= LEVELS; l++) len += i < c[l];
            lenA[W.oA[i]] = (uint8_t)len;
        }
    } else if (tid == 0) {
        if (nA == 0) { lenA[0] = 1; lenA[1] = 1; } else { const int s = W.oA[0]; lenA[s] = 1; lenA[s == 0 ? 1 : 0] = 1; }
    }
    if (nsymB > 0) {
        if (nB >= 2) {
            pm_prefixes_b<LEVELS>(W.cB, nB, c);
            for (int i = tid; i < nB; i += NT) {
                int len = 0;
#pragma unroll
                for (int l = 1; l <= LEVELS; l++) len += i < c[l];
                lenB[W.oB[i]] = (uint8_t)len;
            }
        } else if (tid == 0) {
            if (nB == 0) { lenB[0] = 1; lenB[1] = 1; } else { const int s = W.oB[0]; lenB[s] = 1; lenB[s == 0 ? 1 : 0] = 1; }
        }
    }
    HGW_SYNC();
}

// ---- canonical codes -------------------------------------------------------------------------------------------------
// Number of symbols j < i with len[j] == l, four lengths per LDS read (len is 4-byte aligned).
HGW_FN uint32_t earlier_same_length(const uint8_t *len, int i, uint32_t l) {
    const uint32_t *len32 = (const uint32_t *)len;
    const uint32_t pat = l * 0x01010101u;
    uint32_t k = 0;
    int j = 0;
    for (; j + 4 <= i; j += 4) {
        uint32_t x = len32[j >> 2] ^ pat;                      // a zero byte = an equal length
        // bytes that are zero -> 1 (lengths are < 128, so the classic test has no false positives from borrows... it has none
        // at all here: the high bit of every byte of x is clear)
        x = ~((x + 0x7f7f7f7fu) | 0x7f7f7f7fu);               // 0x80 in every byte that was zero
        k += (x >> 7 & 1u) + (x >> 15 & 1u) + (x >> 23 & 1u) + (x >> 31);
    }
    for (; j < i; j++) k += len[j] == l;
    return k;
}

// Canonical, bit-reversed codes of both alphabets.  Collective; ends with a barrier.
template <int NT>
HGW_FN void wg_assign_codes(HuffWG &W, const uint8_t *lenA, int nsymA, uint16_t *codeA, const uint8_t *lenB, int nsymB, uint16_t *codeB, int tid) {
    if (tid < 32) W.hist[tid >> 4][tid & 15] = 0;
    HGW_SYNC();
    for (int i = tid; i < nsymA; i += NT) if (lenA[i]) HGW_ADD(&W.hist[0][lenA[i]], 1u);
    for (int i = tid - NT / 2; i >= 0 && i < nsymB; i += NT) if (lenB[i]) HGW_ADD(&W.hist[1][lenB[i]], 1u);
    HGW_SYNC();
    if (tid == 0 || tid == NT / 2) {                           // first code of every length (RFC 1951 3.2.2)
        const int t = tid == 0 ? 0 : 1;
        uint32_t code = 0;
        W.nxt[t][0] = 0;
        for (int l = 1; l < 16; l++) { code = (code + (l == 1 ? 0u : W.hist[t][l - 1])) << 1; W.nxt[t][l] = code; }
    }
    HGW_SYNC();
    for (int i = tid; i < nsymA; i += NT) {
        const uint32_t l = lenA[i];
        codeA[i] = l ? (uint16_t)(rev16(W.nxt[0][l] + earlier_same_length(lenA, i, l)) >> (16 - l)) : (uint16_t)0;
    }
    for (int i = tid - NT / 2; i >= 0 && i < nsymB; i += NT) {
        const uint32_t l = lenB[i];
        codeB[i] = l ? (uint16_t)(rev16(W.nxt[1][l] + earlier_same_length(lenB, i, l)) >> (16 - l)) : (uint16_t)0;
    }
    HGW_SYNC();
}

// ---- the dynamic-block header ----------------------------------------------------------------------------------------
// Tokens of one run of `run` equal lengths `v` (the greedy scan of write_dynamic_header, per run): zero runs leave as
// 18(11..138) / 17(3..10) / plain zeros, other values as the value followed by 16(3..6) repeats and up to two plain values.
HGW_FN int run_token_count(uint32_t v, int run) {
    if (v == 0) { const int q = run / 138, rem = run % 138; return q + (rem >= 3 ? 1 : rem); }
    if (run < 4) return run;
    const int left = run - 1, full = left / 6, r2 = left % 6;
    return 1 + full + (r2 >= 3 ? 1 : r2);
}
// k-th token of that run
HGW_FN void run_token(uint32_t v, int run, int k, uint8_t &sym, uint8_t &ext) {
    if (v == 0) {
        const int q = run / 138, rem = run % 138;
        if (k < q) { sym = 18; ext = 127; }
        else if (rem >= 11) { sym = 18; ext = (uint8_t)(rem - 11); }
        else if (rem >= 3) { sym = 17; ext = (uint8_t)(rem - 3); }
        else { sym = 0; ext = 0; }
        return;
    }
    if (run < 4 || k == 0) { sym = (uint8_t)v; ext = 0; return; }
    const int left = run - 1, full = left / 6, r2 = left % 6;
    if (k - 1 < full) { sym = 16; ext = 3; }
    else if (r2 >= 3 && k - 1 == full) { sym = 16; ext = (uint8_t)(r2 - 3); }
    else { sym = (uint8_t)v; ext = 0; }
}

// sum of the bytes a[0..k)
HGW_FN uint32_t byte_prefix_sum(const uint8_t *a, int k) {
    const uint32_t *a32 = (const uint32_t *)a;
    uint32_t s = 0;
    int j = 0;
    for (; j + 4 <= k; j += 4) { const uint32_t x = a32[j >> 2]; const uint32_t y = (x & 0x00ff00ffu) + ((x >> 8) & 0x00ff00ffu); s += (y & 0xffffu) + (y >> 16); }
    for (; j < k; j++) s += a[j];
    return s;
}

// The header of a dynamic block for the given code lengths as a list of bit items (W.item_v / item_n / nitems), its length in
// bits in W.hdr_bits.  ll_len: 288 entries, d_len: 32 entries, both zero beyond the alphabets.  Collective; ends with a barrier.
template <int NT>
HGW_FN void wg_dynamic_header(HuffWG &W, const uint8_t *ll_len, const uint8_t *d_len, int bfinal, int tid) {
    if (tid == 0) { W.hlit = 257; W.hdist = 1; W.hdr_bits = 0; }
    if (tid < 20) W.clf[tid] = 0;
    HGW_SYNC();
    for (int i = tid; i < 286; i += NT) if (ll_len[i]) HGW_MAX(&W.hlit, (uint32_t)(i + 1));
    for (int i = tid - NT / 2; i >= 0 && i < 30; i += NT) if (d_len[i]) HGW_MAX(&W.hdist, (uint32_t)(i + 1));
    HGW_SYNC();
    const int hlit = (int)W.hlit, hdist = (int)W.hdist, n = hlit + hdist;
    for (int i = tid; i < 320; i += NT) { W.seq[i] = i < hlit ? ll_len[i] : i < n ? d_len[i - hlit] : (uint8_t)0xff; W.ntok[i] = 0; }
    HGW_SYNC();
    // run starts: length of the run, its token count
    int my_run[(320 + NT - 1) / NT] = {};
    for (int i = tid, s = 0; i < n; i += NT, s++) {
        const uint32_t v = W.seq[i];
        if (i == 0 || W.seq[i - 1] != v) {
            int run = 1;
            while (i + run < n && W.seq[i + run] == v) run++;
            my_run[s] = run;
            W.ntok[i] = (uint8_t)run_token_count(v, run);
        }
    }
    HGW_SYNC();
    for (int i = tid, s = 0; i < n; i += NT, s++) {
        const int run = my_run[s];
        if (run) {
            const uint32_t v = W.seq[i];
            const int cnt = W.ntok[i];
            const uint32_t off = byte_prefix_sum(W.ntok, i);
            for (int k = 0; k < cnt; k++) {
                uint8_t sym, ext;
                run_token(v, run, k, sym, ext);
                W.cl_sym[off + k] = sym; W.cl_ext[off + k] = ext;
                HGW_ADD(&W.clf[sym], 1u);
            }
            if (i + run == n) W.m = off + (uint32_t)cnt;
        }
    }
    HGW_SYNC();
    // the code-length code: 19 symbols, at most 7 bits
    wg_code_lengths<NT, 7>(W, W.clf, 19, W.cl_len, nullptr, 0, nullptr, tid);
    wg_assign_codes<NT>(W, W.cl_len, 19, W.cl_code, nullptr, 0, nullptr, tid);
    const uint8_t perm[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && W.cl_len[perm[hclen - 1]] == 0) hclen--;
    const int m = (int)W.m;
    // items: the 17 header bits in two items (a value has 16 bits), HCLEN x 3 bits, the m tokens
    uint32_t bits = 0;
    for (int it = tid; it < 2 + hclen + m; it += NT) {
        uint32_t v, nb;
        if (it == 0) { v = (uint32_t)(bfinal ? 1 : 0) | (2u << 1) | ((uint32_t)(hlit - 257) << 3); nb = 8; }
        else if (it == 1) { v = (uint32_t)(hdist - 1) | ((uint32_t)(hclen - 4) << 5); nb = 9; }
        else if (it < 2 + hclen) { v = W.cl_len[perm[it - 2]]; nb = 3; }
        else {
            const uint32_t s = W.cl_sym[it - 2 - hclen], e = W.cl_ext[it - 2 - hclen];
            nb = W.cl_len[s]; v = W.cl_code[s];
            const uint32_t xb = s == 16 ? 2u : s == 17 ? 3u : s == 18 ? 7u : 0u;
            v |= e << nb; nb += xb;
        }
        W.item_v[it] = (uint16_t)v; W.item_n[it] = (uint8_t)nb;
        bits += nb;
    }
    if (bits) HGW_ADD(&W.hdr_bits, bits);
    if (tid == 0) { W.nitems = (uint32_t)(2 + hclen + m); W.hclen = (uint32_t)hclen; }
    HGW_SYNC();
}

}  // namespace hgdef

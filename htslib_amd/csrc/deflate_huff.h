// deflate_huff.h -- Huffman code construction + dynamic-block header for the deflate encoder.
//
// Plain C++ that compiles for the host (unit-tested on CPU through tests/native/huff_host.cpp)
// and for gfx950 (called by one lane of the deflate kernel on LDS-resident arrays).  This is the
// small serial part of the encoder (<= 286 + 30 + 19 symbols); the data-parallel parts (match
// finding, parse, bit packing) are in bgzf_deflate.hip.
//
// Replaces what zlib's trees.c (build_tree / gen_bitlen / send_all_trees) or libdeflate do
// behind bgzf_compress (reference bgzf.c:561-683); written from the DEFLATE spec (RFC 1951
// 3.2.2, 3.2.7), not from those sources.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HG_HD __host__ __device__ __forceinline__
// serial tails: real calls on the device, so that they do not bloat the kernel's register budget (-DHG_SERIAL_INLINE=1 inlines them)
#ifndef HG_SERIAL_INLINE
#define HG_SERIAL_INLINE 0
#endif
#if HG_SERIAL_INLINE
#define HG_HD_SERIAL __host__ __device__ __forceinline__
#else
#define HG_HD_SERIAL __host__ __device__ __attribute__((noinline))
#endif
#else
#define HG_HD inline
#define HG_HD_SERIAL inline
#endif

namespace hgdef {

constexpr int MAX_SYMS = 288;

// In-place minimum-redundancy code lengths (Moffat & Katajainen 1995) for frequencies sorted in
// INCREASING order in A[0..n).  On return A[i] is the code length of the i-th smallest symbol.
HG_HD void min_redundancy_lengths(uint32_t *A, int n) {
    if (n == 0) return;
    if (n == 1) { A[0] = 1; return; }
    // phase 1: build the tree bottom-up; A[root..next) holds internal-node weights, A[..root)
    // parent indices of finished internal nodes
    A[0] += A[1];
    int root = 0, leaf = 2, next;
    for (next = 1; next < n - 1; next++) {
        if (leaf >= n || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = (uint32_t)next; }
        else A[next] = A[leaf++];
        if (leaf >= n || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = (uint32_t)next; }
        else A[next] += A[leaf++];
    }
    // phase 2: internal-node depths
    A[n - 2] = 0;
    for (next = n - 3; next >= 0; next--) A[next] = A[A[next]] + 1;
    // phase 3: leaf depths from the number of internal nodes at each depth
    int avbl = 1, used = 0, dpth = 0;
    root = n - 2; next = n - 1;
    while (avbl > 0) {
        while (root >= 0 && (int)A[root] == dpth) { used++; root--; }
        while (avbl > used) { A[next--] = (uint32_t)dpth; avbl--; }
        avbl = 2 * used; dpth++; used = 0;
    }
}

// Rank of symbol i among the used symbols ordered by (freq, symbol) ascending.  O(n) per
// symbol, independent per symbol: the kernel runs it with one thread per symbol.
HG_HD int rank_symbol(const uint32_t *freq, int n, int i) {
    int r = 0;
    const uint32_t fi = freq[i];
    for (int j = 0; j < n; j++) {
        const uint32_t fj = freq[j];
        if (fj && (fj < fi || (fj == fi && j < i))) r++;
    }
    return r;
}

// Serial tail of the length computation.  On entry order[r] = symbol of rank r and
// work[r] = its frequency for r in [0, used); `cnt` is a 33-entry scratch array.
// Writes len[sym] for the used symbols (the caller zeroed len[]).
HG_HD_SERIAL void finish_lengths(int used, int maxbits, uint8_t *len, const uint16_t *order, uint32_t *work, uint32_t *cnt) {
    if (used == 0) { len[0] = 1; len[1] = 1; return; }          // keep every decoder happy: a complete
    if (used == 1) {                                             // two-code set even if <2 symbols occur
        int s = order[0];
        len[s] = 1; len[s == 0 ? 1 : 0] = 1;
        return;
    }
    min_redundancy_lengths(work, used);
    // enforce the limit: count codes per length, clamp, then repair the Kraft sum
    for (int i = 0; i <= 32; i++) cnt[i] = 0;
    for (int i = 0; i < used; i++) cnt[work[i] > 32 ? 32 : work[i]]++;
    for (int i = maxbits + 1; i <= 32; i++) { cnt[maxbits] += cnt[i]; cnt[i] = 0; }
    uint32_t total = 0;
    for (int i = maxbits; i > 0; i--) total += cnt[i] << (maxbits - i);
    while (total != (1u << maxbits)) {               // over-subscribed after clamping
        cnt[maxbits]--;
        for (int i = maxbits - 1; i > 0; i--)
            if (cnt[i]) { cnt[i]--; cnt[i + 1] += 2; break; }
        total--;
    }
    // hand the lengths out: the most frequent symbols (end of `order`) get the shortest codes
    int idx = used - 1;
    for (int l = 1; l <= maxbits; l++)
        for (uint32_t c = cnt[l]; c > 0; c--) len[order[idx--]] = (uint8_t)l;
}

// Build length-limited canonical Huffman code lengths (whole thing, one thread).
//   freq[0..n)   symbol frequencies (may contain zeros)
//   len[0..n)    out: code length per symbol (0 = unused), all <= maxbits
//   order/work   scratch arrays of n entries each
HG_HD void build_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *len, uint16_t *order, uint32_t *work) {
    int used = 0;
    for (int i = 0; i < n; i++) { len[i] = 0; if (freq[i]) used++; }
    for (int i = 0; i < n; i++) {
        if (!freq[i]) continue;
        int r = rank_symbol(freq, n, i);
        order[r] = (uint16_t)i; work[r] = freq[i];
    }
    uint32_t cnt[33];
    finish_lengths(used, maxbits, len, order, work, cnt);
}

HG_HD uint32_t rev16(uint32_t v) {                      // reverse the low 16 bits
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0f0fu) << 4) | ((v >> 4) & 0x0f0fu);
    return ((v & 0x00ffu) << 8) | ((v >> 8) & 0x00ffu);
}

// First canonical code of every length (RFC 1951 3.2.2) into nxt[1..15]; cnt16 is scratch.
HG_HD_SERIAL void first_codes(const uint8_t *len, int n, uint32_t *cnt16, uint32_t *nxt16) {
    for (int i = 0; i < 16; i++) cnt16[i] = 0;
    for (int i = 0; i < n; i++) cnt16[len[i]]++;
    cnt16[0] = 0;
    uint32_t c = 0;
    nxt16[0] = 0;
    for (int l = 1; l < 16; l++) { c = (c + cnt16[l - 1]) << 1; nxt16[l] = c; }
}
// Code of symbol i given the first codes: first + number of earlier symbols of the same length.
// Bit-reversed so it can be OR-ed LSB-first into the stream.  Independent per symbol.
HG_HD uint16_t code_of(const uint8_t *len, int i, const uint32_t *nxt16) {
    const uint32_t l = len[i];
    if (!l) return 0;
    uint32_t k = 0;
    for (int j = 0; j < i; j++) k += len[j] == l;
    return (uint16_t)(rev16(nxt16[l] + k) >> (16 - l));
}
// Canonical codes for all symbols (one thread).
HG_HD void assign_codes(const uint8_t *len, int n, uint16_t *code) {
    uint32_t cnt[16], nxt[16];
    first_codes(len, n, cnt, nxt);
    for (int i = 0; i < n; i++) code[i] = code_of(len, i, nxt);
}

// LSB-first bit writer into a byte buffer (used for the <= ~200 byte block header only)
// (bits are collected in a register and leave a byte at a time: a read-modify-write of the buffer per BIT was ~40 % of the
// single-lane Huffman tail on the device)
struct BitSink {
    uint8_t *p; uint32_t nbits;
    unsigned long long acc = 0; uint32_t have = 0;
    HG_HD void put(uint32_t v, uint32_t n) {                     // n <= 16
        acc |= (unsigned long long)(v & ((1u << n) - 1u)) << have;
        have += n; nbits += n;
        while (have >= 8) { p[(nbits - have) >> 3] = (uint8_t)acc; acc >>= 8; have -= 8; }
    }
    HG_HD void finish() { if (have) p[(nbits - have) >> 3] = (uint8_t)acc; }   // the last, partial byte (upper bits zero)
    HG_HD void resume() { have = nbits & 7u; acc = have ? (p[nbits >> 3] & ((1u << have) - 1u)) : 0u; }   // continue behind nbits bits already in p
};

// Emit BFINAL=1, BTYPE=10 and the two trees (RFC 1951 3.2.7) into `dst` (>= 320 bytes).
// ll_len[0..286), d_len[0..30).  Returns the number of bits written.
// scratch: cl_sym/cl_ext hold the run-length coded length sequence (<= 316 entries each).
HG_HD_SERIAL uint32_t write_dynamic_header(const uint8_t *ll_len, const uint8_t *d_len, uint8_t *dst,
                                    uint8_t *cl_sym, uint8_t *cl_ext, uint32_t *work, uint16_t *order) {
    int hlit = 286; while (hlit > 257 && ll_len[hlit - 1] == 0) hlit--;
    int hdist = 30; while (hdist > 1 && d_len[hdist - 1] == 0) hdist--;
    // run-length code the concatenated lengths with symbols 16/17/18
    int n = hlit + hdist, m = 0;
    uint32_t clf[19];
    for (int i = 0; i < 19; i++) clf[i] = 0;
    int i = 0;
    while (i < n) {
        uint8_t v = i < hlit ? ll_len[i] : d_len[i - hlit];
        int run = 1;
        while (i + run < n && (i + run < hlit ? ll_len[i + run] : d_len[i + run - hlit]) == v) run++;
        if (v == 0 && run >= 3) {
            int r = run > 138 ? 138 : run;
            if (r <= 10) { cl_sym[m] = 17; cl_ext[m] = (uint8_t)(r - 3); } else { cl_sym[m] = 18; cl_ext[m] = (uint8_t)(r - 11); }
            clf[cl_sym[m]]++; m++; i += r;
        } else if (v != 0 && run >= 4) {
            cl_sym[m] = v; cl_ext[m] = 0; clf[v]++; m++; i++;          // the value itself, then repeats
            int left = run - 1;
            while (left >= 3) {
                int r = left > 6 ? 6 : left;
                cl_sym[m] = 16; cl_ext[m] = (uint8_t)(r - 3); clf[16]++; m++; i += r; left -= r;
            }
            // (a remainder of 1-2 is emitted by the next iterations as plain values)
        } else {
            cl_sym[m] = v; cl_ext[m] = 0; clf[v]++; m++; i++;
        }
    }
    uint8_t cl_len[19]; uint16_t cl_code[19];
    build_lengths(clf, 19, 7, cl_len, order, work);
    assign_codes(cl_len, 19, cl_code);
    const uint8_t perm[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19; while (hclen > 4 && cl_len[perm[hclen - 1]] == 0) hclen--;
    BitSink bs{dst, 0};
    bs.put(1, 1); bs.put(2, 2);
    bs.put((uint32_t)(hlit - 257), 5); bs.put((uint32_t)(hdist - 1), 5); bs.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; k++) bs.put(cl_len[perm[k]], 3);
    for (int k = 0; k < m; k++) {
        uint8_t s = cl_sym[k];
        bs.put(cl_code[s], cl_len[s]);
        if (s == 16) bs.put(cl_ext[k], 2); else if (s == 17) bs.put(cl_ext[k], 3); else if (s == 18) bs.put(cl_ext[k], 7);
    }
    bs.finish();
    return bs.nbits;
}

// length (3..258) -> litlen symbol index (0..28, add 257), extra-bit count and value
HG_HD void len_symbol(uint32_t len, uint32_t &sym, uint32_t &xb, uint32_t &xv) {
    uint32_t l = len - 3;
    if (l < 8) { sym = l; xb = 0; xv = 0; return; }
    if (l == 255) { sym = 28; xb = 0; xv = 0; return; }
    uint32_t lg = 31u - (uint32_t)__builtin_clz(l);
    xb = lg - 2;
    sym = 4 * (lg - 1) + ((l >> (lg - 2)) & 3u);
    xv = l & ((1u << xb) - 1u);
}
// distance (1..32768) -> distance symbol (0..29), extra-bit count and value
HG_HD void dist_symbol(uint32_t dist, uint32_t &sym, uint32_t &xb, uint32_t &xv) {
    uint32_t d = dist - 1;
    if (d < 4) { sym = d; xb = 0; xv = 0; return; }
    uint32_t lg = 31u - (uint32_t)__builtin_clz(d);
    xb = lg - 1;
    sym = 2 * lg + ((d >> (lg - 1)) & 1u);
    xv = d & ((1u << xb) - 1u);
}

}  // namespace hgdef

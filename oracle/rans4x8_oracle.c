/*
 * rans4x8_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C CPU restatement of the CRAM 3.0 "rANS 4x8" block codec (CRAM block method 4), the
 * codec behind
 *      rans_uncompress(in, in_size, &out_size)           cram/cram_io.c:1668
 *      rans_compress(in, in_size, &out_size, order)      cram/cram_io.c:1838
 * The implementation the reference links is htscodecs v1.6.6 (rANS_static.c), a git submodule
 * that is ABSENT from /root/reference (.gitmodules:1-4, htscodecs_bundled.mk:26-39), so this file
 * restates the published algorithm (CRAM format specification v3.0 section 13.4 "rANS codec",
 * hts-specs; Duda's rANS with 12-bit frequencies, 4 interleaved 32-bit states, byte-wise
 * renormalisation, lower bound 2^23) and is anchored on the reference's own fixtures:
 *
 * Pinning: tests/golden/rans4x8/ holds every rANS 4x8 block of the reference's CRAM v3.0 test
 * files (test/ce#5b_java.cram, auxf#values_java.cram, xx#large_aux_java.cram, range.cram --
 * the first three written by an independent Java implementation, range.cram by htslib).  For 28
 * of the 44 blocks, of both orders, the expected plaintext is derived from the .sam / .bam twins
 * WITHOUT any rANS code (QS, RN, SC, BF, RL, TS, AP, one-byte and string aux tags:
 * tests/golden/make_golden_rans.py); the other blocks are checked for their declared size and for
 * encode->decode identity.  The DECODER is pinned by those; the encoder through decode(encode(x)).
 *
 * Stream layout (all little endian):
 *   u8  order (0|1)   u32 compressed size (bytes after this 9-byte prefix)   u32 uncompressed size
 *   frequency table(s)   4 x u32 initial states (state 0 first)   renormalisation bytes
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))
#define TF_SHIFT 12
#define TOTFREQ (1u << TF_SHIFT)
#define RANS_L (1u << 23)

/* ------------------------------------------------------------------ decode */
/* one order-0 style table: symbol run-length list + 1/2-byte frequencies.  Returns new cursor or
 * NULL.  F[] and C[] (exclusive cumulative) are filled, lookup[] maps slot -> symbol. */
static const uint8_t *read_table0(const uint8_t *cp, const uint8_t *end, uint16_t *F, uint16_t *C, uint8_t *lookup,
                                  uint32_t *total)
{
    memset(F, 0, 256 * sizeof(uint16_t));
    memset(C, 0, 256 * sizeof(uint16_t));
    if (cp >= end) return NULL;
    unsigned rle = 0, x = 0, j = *cp++;
    do {
        if (cp + 2 > end) return NULL;
        unsigned f = *cp++;
        if (f >= 128) f = ((f & 127) << 8) | *cp++;
        if (x + f > TOTFREQ) return NULL;
        F[j] = (uint16_t)f; C[j] = (uint16_t)x;
        if (lookup) memset(lookup + x, (int)j, f);
        x += f;
        if (cp >= end) return NULL;
        if (!rle && j + 1 == *cp) {
            j = *cp++;
            if (cp >= end) return NULL;
            rle = *cp++;
        } else if (rle) {
            rle--; j++;
            if (j > 255) return NULL;
        } else {
            j = *cp++;
        }
    } while (j);
    *total = x;
    return cp;
}

/* A state whose slot lies beyond the table's total can only come from a corrupt stream; it is
 * reported as an error (the GPU decoder does the same).  Valid encoders never produce it. */
static int dec_o0(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_sz)
{
    uint16_t F[256], C[256];
    static __thread uint8_t lookup[TOTFREQ];
    const uint8_t *end = in + in_size;
    memset(lookup, 0, sizeof lookup);
    uint32_t total = 0;
    const uint8_t *cp = read_table0(in + 9, end, F, C, lookup, &total);
    if (!cp || cp + 16 > end) return -1;
    uint32_t R[4];
    for (int k = 0; k < 4; k++, cp += 4) R[k] = cp[0] | (cp[1] << 8) | (cp[2] << 16) | ((uint32_t)cp[3] << 24);
    size_t out_end = out_sz & ~(size_t)3;
    for (size_t i = 0; i < out_end; i += 4) {
        for (int k = 0; k < 4; k++) {
            uint32_t m = R[k] & (TOTFREQ - 1);
            if (m >= total) return -1;
            uint8_t c = lookup[m];
            out[i + k] = c;
            R[k] = F[c] * (R[k] >> TF_SHIFT) + m - C[c];
        }
        for (int k = 0; k < 4; k++)
            while (R[k] < RANS_L) { if (cp >= end) return -1; R[k] = (R[k] << 8) | *cp++; }
    }
    for (size_t k = 0; k < (out_sz & 3); k++) {
        if ((R[k] & (TOTFREQ - 1)) >= total) return -1;
        out[out_end + k] = lookup[R[k] & (TOTFREQ - 1)];
    }
    return 0;
}

static int dec_o1(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_sz)
{
    const uint8_t *end = in + in_size;
    uint16_t (*F)[256] = calloc(256, sizeof *F), (*C)[256] = calloc(256, sizeof *C);
    uint8_t (*lookup)[TOTFREQ] = calloc(256, TOTFREQ);
    uint32_t T[256];
    memset(T, 0, sizeof T);
    int rc = -1;
    if (!F || !C || !lookup) goto done;
    const uint8_t *cp = in + 9;
    if (cp >= end) goto done;
    unsigned rle_i = 0, i = *cp++;
    do {
        cp = read_table0(cp, end, F[i], C[i], lookup[i], &T[i]);
        if (!cp || cp >= end) goto done;
        if (!rle_i && i + 1 == *cp) {
            i = *cp++;
            if (cp >= end) goto done;
            rle_i = *cp++;
        } else if (rle_i) {
            rle_i--; i++;
            if (i > 255) goto done;
        } else {
            i = *cp++;
        }
    } while (i);
    if (cp + 16 > end) goto done;
    uint32_t R[4];
    for (int k = 0; k < 4; k++, cp += 4) R[k] = cp[0] | (cp[1] << 8) | (cp[2] << 16) | ((uint32_t)cp[3] << 24);
    size_t isz4 = out_sz >> 2, i4[4] = {0, isz4, 2 * isz4, 3 * isz4};
    unsigned l[4] = {0, 0, 0, 0};
    for (; i4[0] < isz4; i4[0]++, i4[1]++, i4[2]++, i4[3]++) {
        for (int k = 0; k < 4; k++) {
            uint32_t m = R[k] & (TOTFREQ - 1);
            if (m >= T[l[k]]) goto done;
            uint8_t c = lookup[l[k]][m];
            out[i4[k]] = c;
            R[k] = F[l[k]][c] * (R[k] >> TF_SHIFT) + m - C[l[k]][c];
            l[k] = c;
        }
        for (int k = 0; k < 4; k++)
            while (R[k] < RANS_L) { if (cp >= end) goto done; R[k] = (R[k] << 8) | *cp++; }
    }
    for (; i4[3] < out_sz; i4[3]++) {                     /* the tail belongs to the last quarter */
        uint32_t m = R[3] & (TOTFREQ - 1);
        if (m >= T[l[3]]) goto done;
        uint8_t c = lookup[l[3]][m];
        out[i4[3]] = c;
        R[3] = F[l[3]][c] * (R[3] >> TF_SHIFT) + m - C[l[3]][c];
        while (R[3] < RANS_L) { if (cp >= end) goto done; R[3] = (R[3] << 8) | *cp++; }
        l[3] = c;
    }
    rc = 0;
done:
    free(F); free(C); free(lookup);
    return rc;
}

/* Returns 0 and fills out[0..*out_size) (capacity out_cap), or -1 on a malformed stream. */
ORC_EXPORT int orc_rans4x8_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size)
{
    if (in_size < 9) return -1;
    uint32_t csz = in[1] | (in[2] << 8) | (in[3] << 16) | ((uint32_t)in[4] << 24);
    uint32_t usz = in[5] | (in[6] << 8) | (in[7] << 16) | ((uint32_t)in[8] << 24);
    if ((size_t)csz + 9 != in_size || usz > out_cap) return -1;
    *out_size = usz;
    if (usz == 0) return 0;
    if (in[0] == 0) return dec_o0(in, in_size, out, usz);
    if (in[0] == 1) return dec_o1(in, in_size, out, usz);
    return -1;
}

/* ------------------------------------------------------------------ encode */
typedef struct { uint32_t start, freq; } sym_t;

static void enc_put(uint32_t *r, uint8_t **pp, const sym_t *s)
{
    uint32_t x = *r, x_max = ((RANS_L >> TF_SHIFT) << 8) * s->freq;
    while (x >= x_max) { *--(*pp) = (uint8_t)x; x >>= 8; }
    *r = ((x / s->freq) << TF_SHIFT) + (x % s->freq) + s->start;
}
static void enc_flush(uint32_t r, uint8_t **pp)
{
    *pp -= 4;
    (*pp)[0] = (uint8_t)r; (*pp)[1] = (uint8_t)(r >> 8); (*pp)[2] = (uint8_t)(r >> 16); (*pp)[3] = (uint8_t)(r >> 24);
}

/* scale counts so that they sum to TOTFREQ-1 (stock decoders require a total < 4096) with every
 * present symbol >= 1 */
static void normalise(const uint32_t *cnt, uint32_t total, uint16_t *F)
{
    const uint32_t target = TOTFREQ - 1;
    uint32_t fsum = 0, M = 0, m = 0;
    for (int j = 0; j < 256; j++) {
        F[j] = 0;
        if (!cnt[j]) continue;
        uint64_t f = ((uint64_t)cnt[j] * target) / total;
        if (f == 0) f = 1;
        F[j] = (uint16_t)f; fsum += (uint32_t)f;
        if (cnt[j] > m) { m = cnt[j]; M = j; }
    }
    if (fsum < target) F[M] += target - fsum;
    else if (fsum > target) {
        uint32_t over = fsum - target;
        /* take the excess from the largest entries */
        while (over) {
            uint32_t best = 0;
            for (int j = 1; j < 256; j++) if (F[j] > F[best]) best = j;
            uint32_t take = F[best] - 1 < over ? F[best] - 1 : over;
            F[best] -= take; over -= take;
            if (!take) break;
        }
    }
}

static uint8_t *write_table0(uint8_t *cp, const uint16_t *F)
{
    int rle = 0;
    for (int j = 0; j < 256; j++) {
        if (!F[j]) continue;
        if (rle) rle--;
        else {
            *cp++ = (uint8_t)j;
            if (j && F[j - 1]) {
                for (rle = j + 1; rle < 256 && F[rle]; rle++) ;
                rle -= j + 1;
                *cp++ = (uint8_t)rle;
            }
        }
        if (F[j] < 128) *cp++ = (uint8_t)F[j];
        else { *cp++ = (uint8_t)(128 | (F[j] >> 8)); *cp++ = (uint8_t)(F[j] & 0xff); }
    }
    *cp++ = 0;
    return cp;
}

static size_t enc_o0(const uint8_t *in, size_t n, uint8_t *out)
{
    uint32_t cnt[256] = {0};
    uint16_t F[256];
    sym_t syms[256];
    for (size_t i = 0; i < n; i++) cnt[in[i]]++;
    normalise(cnt, (uint32_t)n, F);
    uint8_t *cp = write_table0(out + 9, F);
    uint32_t x = 0;
    for (int j = 0; j < 256; j++) { syms[j].start = x; syms[j].freq = F[j]; x += F[j]; }
    uint8_t *buf = malloc(n * 2 + 64), *ptr = buf + n * 2 + 64, *bend = ptr;
    uint32_t R[4] = {RANS_L, RANS_L, RANS_L, RANS_L};
    size_t i = n & 3;
    switch (i) {
    case 3: enc_put(&R[2], &ptr, &syms[in[n - (i - 2)]]); /* fall through */
    case 2: enc_put(&R[1], &ptr, &syms[in[n - (i - 1)]]); /* fall through */
    case 1: enc_put(&R[0], &ptr, &syms[in[n - i]]);
    }
    for (i = n & ~(size_t)3; i > 0; i -= 4) {
        enc_put(&R[3], &ptr, &syms[in[i - 1]]);
        enc_put(&R[2], &ptr, &syms[in[i - 2]]);
        enc_put(&R[1], &ptr, &syms[in[i - 3]]);
        enc_put(&R[0], &ptr, &syms[in[i - 4]]);
    }
    enc_flush(R[3], &ptr); enc_flush(R[2], &ptr); enc_flush(R[1], &ptr); enc_flush(R[0], &ptr);
    size_t body = (size_t)(bend - ptr);
    memcpy(cp, ptr, body);
    free(buf);
    return (size_t)(cp - out) + body;
}

static size_t enc_o1(const uint8_t *in, size_t n, uint8_t *out)
{
    uint32_t (*cnt)[256] = calloc(256, sizeof *cnt);
    uint32_t T[256] = {0};
    uint16_t (*F)[256] = calloc(256, sizeof *F);
    sym_t (*syms)[256] = calloc(256, sizeof *syms);
    size_t isz4 = n >> 2;
    unsigned last = 0;
    for (size_t i = 0; i < n; i++) { cnt[last][in[i]]++; T[last]++; last = in[i]; }
    cnt[0][in[1 * isz4]]++; cnt[0][in[2 * isz4]]++; cnt[0][in[3 * isz4]]++; T[0] += 3;
    uint8_t *cp = out + 9;
    int rle_i = 0;
    for (int i = 0; i < 256; i++) {
        if (!T[i]) continue;
        normalise(cnt[i], T[i], F[i]);
        if (rle_i) rle_i--;
        else {
            *cp++ = (uint8_t)i;
            if (i && T[i - 1]) {
                for (rle_i = i + 1; rle_i < 256 && T[rle_i]; rle_i++) ;
                rle_i -= i + 1;
                *cp++ = (uint8_t)rle_i;
            }
        }
        cp = write_table0(cp, F[i]);
        uint32_t x = 0;
        for (int j = 0; j < 256; j++) { syms[i][j].start = x; syms[i][j].freq = F[i][j]; x += F[i][j]; }
    }
    *cp++ = 0;
    uint8_t *buf = malloc(n * 2 + 64), *ptr = buf + n * 2 + 64, *bend = ptr;
    uint32_t R[4] = {RANS_L, RANS_L, RANS_L, RANS_L};
    long i0 = (long)(1 * isz4) - 2, i1 = (long)(2 * isz4) - 2, i2 = (long)(3 * isz4) - 2, i3;
    unsigned l0 = in[i0 + 1], l1 = in[i1 + 1], l2 = in[i2 + 1], l3 = in[n - 1];
    for (i3 = (long)n - 2; i3 > (long)(4 * isz4) - 2; i3--) {
        unsigned c3 = in[i3];
        enc_put(&R[3], &ptr, &syms[c3][l3]);
        l3 = c3;
    }
    for (; i0 >= 0; i0--, i1--, i2--, i3--) {
        unsigned c0 = in[i0], c1 = in[i1], c2 = in[i2], c3 = in[i3];
        enc_put(&R[3], &ptr, &syms[c3][l3]);
        enc_put(&R[2], &ptr, &syms[c2][l2]);
        enc_put(&R[1], &ptr, &syms[c1][l1]);
        enc_put(&R[0], &ptr, &syms[c0][l0]);
        l0 = c0; l1 = c1; l2 = c2; l3 = c3;
    }
    enc_put(&R[3], &ptr, &syms[0][l3]);
    enc_put(&R[2], &ptr, &syms[0][l2]);
    enc_put(&R[1], &ptr, &syms[0][l1]);
    enc_put(&R[0], &ptr, &syms[0][l0]);
    enc_flush(R[3], &ptr); enc_flush(R[2], &ptr); enc_flush(R[1], &ptr); enc_flush(R[0], &ptr);
    size_t body = (size_t)(bend - ptr);
    memcpy(cp, ptr, body);
    size_t tot = (size_t)(cp - out) + body;
    free(buf); free(cnt); free(F); free(syms);
    return tot;
}

ORC_EXPORT size_t orc_rans4x8_compress_bound(size_t n) { return (size_t)(1.05 * n) + 257 * 257 * 3 + 9 + 64; }

/* out must hold orc_rans4x8_compress_bound(n) bytes.  Returns the stream length (0 on error). */
ORC_EXPORT size_t orc_rans4x8_compress(const uint8_t *in, size_t n, uint8_t *out, int order)
{
    size_t tot;
    if (n == 0) {                                         /* empty input: header only, order 0 */
        memset(out, 0, 9);
        return 9;
    }
    if (order && n < 4) order = 0;
    tot = order ? enc_o1(in, n, out) : enc_o0(in, n, out);
    out[0] = (uint8_t)order;
    uint32_t csz = (uint32_t)(tot - 9), usz = (uint32_t)n;
    for (int k = 0; k < 4; k++) { out[1 + k] = (uint8_t)(csz >> (8 * k)); out[5 + k] = (uint8_t)(usz >> (8 * k)); }
    return tot;
}

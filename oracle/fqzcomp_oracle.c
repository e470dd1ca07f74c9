/*
 * fqzcomp_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * *** PARITY UNPINNED ***  Plain-C restatement of the CRAM 3.1 fqzcomp quality codec (block method 7), the codec behind
 *      fqz_decompress((char *)b->data, b->comp_size, &uncomp_size, NULL, 0)           cram/cram_io.c:1684-1695
 *      fqz_compress(vers, fqz_slice, in, in_size, &out_size, strat, NULL)             cram/cram_io.c:1801-1825
 * whose implementation (htscodecs v1.6.6 fqzcomp_qual.c) is an ABSENT git submodule of the reference, with no golden
 * stream in the reference's tests (all its CRAM fixtures are v3.0).  This file follows the published specification
 * (hts-specs "CRAM codecs" v3.1, chapter "FQZComp quality codec": FQZDecodeParams / ReadArray / FQZNewRecord /
 * FQZUpdateContext).  Byte-level agreement with htscodecs is UNVERIFIED; the encoder's CHOICE of parameters is free by
 * the format (any parameter block a decoder accepts is valid), ours is modelled on the four htscodecs presets but is not
 * claimed to reproduce them.
 *
 * Stream:  ulen:uint7   version:u8 (5)   gflags:u8 {1 MULTI_PARAM, 2 HAVE_STAB, 4 DO_REV}
 *          [nparam:u8 if MULTI_PARAM]    [max_sel:u8, stab:array(256) if HAVE_STAB]
 *          nparam x { context:u16le  pflags:u8 {2 DO_DEDUP, 4 DO_LEN (fixed length), 8 DO_SEL, 16 HAVE_QMAP, 32 HAVE_PTAB,
 *                     64 HAVE_DTAB, 128 HAVE_QTAB}  max_sym:u8  qbits<<4|qshift  qloc<<4|sloc  ploc<<4|dloc
 *                     [qmap: max_sym bytes] [qtab: array(256)] [ptab: array(1024)] [dtab: array(256)] }
 *          range-coder bytes (range_model.h).
 * array(n): the table is non-decreasing; it is stored as the run length of value 0, of value 1, ... (a run >= 255 continues
 * in the next byte), and that byte string is itself run-length coded: a byte equal to its predecessor is followed by a
 * repeat count.  The reader stops as soon as the run lengths add up to n, so when the LAST run is a multiple of 255 its
 * terminating 0 would be left unread in front of the next field: the writer here does not emit it and the reader accepts a
 * final 255 (an edge the specification's pseudocode leaves open).
 * Models: 65536 quality models of max_sym+1 symbols indexed by a 16-bit context, 4 length-byte models (256), reverse and
 * duplicate flags (2 each), the parameter selector (max_sel+1).
 * Per record: [selector] [4 length bytes unless fixed and already seen] [reverse flag if DO_REV] [duplicate flag if DO_DEDUP;
 * a duplicate copies the previous record].  Per quality: Q = decode(model[ctx]); out = qmap[Q];
 *   qctx = (qctx << qshift) + qtab[Q];  ctx = context + ((qctx & (2^qbits - 1)) << qloc) + (ptab[min(1023, remaining)] << ploc)
 *        + (dtab[min(255, delta)] << dloc) + (selector << sloc), 16 bits;  delta += (Q != previous Q); remaining -= 1.
 * Reversed records (flag set) are turned round after decoding / before encoding.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "range_model.h"

#define ORC_EXPORT __attribute__((visibility("default")))
#define FQZ_VERS 5
#define GF_MULTI 1
#define GF_STAB 2
#define GF_REV 4
#define PF_DEDUP 2
#define PF_LEN 4
#define PF_SEL 8
#define PF_QMAP 16
#define PF_PTAB 32
#define PF_DTAB 64
#define PF_QTAB 128
#define CTX_SIZE 65536u

typedef struct {
    uint32_t context, pflags, max_sym, qbits, qshift, qloc, sloc, ploc, dloc;
    uint32_t qmap[256], qtab[256], ptab[1024], dtab[256];
} fqz_param;
typedef struct {
    uint32_t gflags, nparam, max_sel, max_sym;
    uint32_t stab[256];
    fqz_param p[256];
} fqz_gparams;

/* ---- the two-level run-length array --------------------------------------------------------------- */
static int read_array(const uint8_t *in, size_t in_size, uint32_t *array, int size)
{
    uint8_t R[1024];
    int i, j, z, last = -1;
    for (i = j = z = 0; z < size && (size_t)i < in_size; i++) {
        int run = in[i];
        R[j++] = (uint8_t)run;
        z += run;
        if (run == last) {
            if ((size_t)i + 1 >= in_size) return -1;
            int copy = in[++i];
            z += run * copy;
            while (copy-- && z <= size && j < 1024) R[j++] = (uint8_t)run;
        }
        if (j >= 1024) return -1;
        last = run;
    }
    const int nb = i, rmax = j;
    for (i = j = z = 0; j < size; i++) {
        int len = 0, part;
        if (z >= rmax) return -1;
        do { part = R[z++]; len += part; } while (part == 255 && z < rmax);
        while (len && j < size) { len--; array[j++] = (uint32_t)i; }
    }
    return nb;
}
static int store_array(uint8_t *out, const uint32_t *array, int size)
{
    uint8_t tmp[2048];
    int i = 0, k = 0;
    for (uint32_t v = 0; i < size; v++) {
        int len = 0;
        while (i < size && array[i] == v) { i++; len++; }
        int r;
        do { r = len < 255 ? len : 255; tmp[k++] = (uint8_t)r; len -= r; } while (r == 255);
    }
    if (k >= 2 && tmp[k - 1] == 0 && tmp[k - 2] == 255) k--;          /* see the header: the reader stops once the table is full */
    int o = 0, last = -1;
    for (int j = 0; j < k;) {
        out[o] = tmp[j++];
        if (out[o] == last) {
            int n = 0;
            while (j < k && tmp[j] == last && n < 255) { j++; n++; }
            out[++o] = (uint8_t)n;
        } else last = out[o];
        o++;
    }
    return o;
}

/* ---- parameter block -------------------------------------------------------------------------------- */
static int read_params(fqz_gparams *g, const uint8_t *in, size_t n)
{
    size_t p = 0;
    if (n < 10 || in[p++] != FQZ_VERS) return -1;
    g->gflags = in[p++];
    g->nparam = (g->gflags & GF_MULTI) ? in[p++] : 1;
    if (g->nparam == 0) return -1;
    g->max_sel = g->nparam > 1 ? g->nparam - 1 : 0;
    if (g->gflags & GF_STAB) {
        g->max_sel = in[p++];
        int r = read_array(in + p, n - p, g->stab, 256);
        if (r < 0) return -1;
        p += (size_t)r;
    } else {
        for (uint32_t i = 0; i < g->nparam; i++) g->stab[i] = i;
        for (uint32_t i = g->nparam; i < 256; i++) g->stab[i] = g->nparam - 1;
    }
    g->max_sym = 0;
    for (uint32_t k = 0; k < g->nparam; k++) {
        fqz_param *m = &g->p[k];
        if (p + 7 > n) return -1;
        m->context = in[p] | (uint32_t)in[p + 1] << 8;
        m->pflags = in[p + 2];
        m->max_sym = in[p + 3];
        m->qbits = in[p + 4] >> 4; m->qshift = in[p + 4] & 15;
        m->qloc = in[p + 5] >> 4; m->sloc = in[p + 5] & 15;
        m->ploc = in[p + 6] >> 4; m->dloc = in[p + 6] & 15;
        p += 7;
        for (uint32_t i = 0; i < 256; i++) m->qmap[i] = i;
        if (m->pflags & PF_QMAP) {
            if (p + m->max_sym > n) return -1;
            for (uint32_t i = 0; i < m->max_sym; i++) m->qmap[i] = in[p++];
        }
        for (uint32_t i = 0; i < 256; i++) m->qtab[i] = i;
        if (m->pflags & PF_QTAB) { int r = read_array(in + p, n - p, m->qtab, 256); if (r < 0) return -1; p += (size_t)r; }
        memset(m->ptab, 0, sizeof m->ptab);
        if (m->pflags & PF_PTAB) { int r = read_array(in + p, n - p, m->ptab, 1024); if (r < 0) return -1; p += (size_t)r; }
        memset(m->dtab, 0, sizeof m->dtab);
        if (m->pflags & PF_DTAB) { int r = read_array(in + p, n - p, m->dtab, 256); if (r < 0) return -1; p += (size_t)r; }
        if (m->max_sym > g->max_sym) g->max_sym = m->max_sym;
    }
    for (uint32_t i = 0; i < 256; i++) if (g->stab[i] >= g->nparam) return -1;
    return (int)p;
}
static int write_params(const fqz_gparams *g, uint8_t *out)
{
    int p = 0;
    out[p++] = FQZ_VERS;
    out[p++] = (uint8_t)g->gflags;
    if (g->gflags & GF_MULTI) out[p++] = (uint8_t)g->nparam;
    if (g->gflags & GF_STAB) { out[p++] = (uint8_t)g->max_sel; p += store_array(out + p, g->stab, 256); }
    for (uint32_t k = 0; k < g->nparam; k++) {
        const fqz_param *m = &g->p[k];
        out[p++] = m->context & 0xff; out[p++] = (m->context >> 8) & 0xff;
        out[p++] = (uint8_t)m->pflags; out[p++] = (uint8_t)m->max_sym;
        out[p++] = (uint8_t)(m->qbits << 4 | m->qshift); out[p++] = (uint8_t)(m->qloc << 4 | m->sloc);
        out[p++] = (uint8_t)(m->ploc << 4 | m->dloc);
        if (m->pflags & PF_QMAP) for (uint32_t i = 0; i < m->max_sym; i++) out[p++] = (uint8_t)m->qmap[i];
        if (m->pflags & PF_QTAB) p += store_array(out + p, m->qtab, 256);
        if (m->pflags & PF_PTAB) p += store_array(out + p, m->ptab, 1024);
        if (m->pflags & PF_DTAB) p += store_array(out + p, m->dtab, 256);
    }
    return p;
}

typedef struct { uint32_t qctx, p, delta, prevq, s; } fqz_state;
static inline uint32_t update_ctx(const fqz_param *m, fqz_state *st, uint32_t q)
{
    uint32_t c = m->context;
    st->qctx = (st->qctx << m->qshift) + m->qtab[q];
    c += (st->qctx & ((1u << m->qbits) - 1u)) << m->qloc;
    if (m->pflags & PF_PTAB) c += m->ptab[st->p < 1023 ? st->p : 1023] << m->ploc;
    if (m->pflags & PF_DTAB) {
        c += m->dtab[st->delta < 255 ? st->delta : 255] << m->dloc;
        st->delta += st->prevq != q;
        st->prevq = q;
    }
    if (m->pflags & PF_SEL) c += st->s << m->sloc;
    st->p--;
    return c & (CTX_SIZE - 1);
}

typedef struct { model_t *qual, len[4], rev, dup, sel; } fqz_models;
static int models_new(fqz_models *M, const fqz_gparams *g)
{
    M->qual = malloc(CTX_SIZE * sizeof(model_t));
    if (!M->qual) return -1;
    for (uint32_t i = 0; i < CTX_SIZE; i++) model_init(&M->qual[i], g->max_sym + 1);
    for (int i = 0; i < 4; i++) model_init(&M->len[i], 256);
    model_init(&M->rev, 2); model_init(&M->dup, 2);
    model_init(&M->sel, g->max_sel + 1);
    return 0;
}
static void reverse_bytes(uint8_t *b, uint32_t n) { for (uint32_t i = 0, j = n; i + 1 < j; i++) { j--; uint8_t t = b[i]; b[i] = b[j]; b[j] = t; } }

/* returns 0 and the plaintext in out[0..*out_len) (out_cap bytes available), or -1.  When lens != NULL it receives the record
 * lengths found in the stream (at most max_rec; *nrec the count) */
ORC_EXPORT int orc_fqz_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *lens,
                              size_t max_rec, size_t *nrec)
{
    uint32_t ulen;
    int k = get_u7(in, in + in_len, &ulen);
    if (k < 0 || ulen > out_cap) return -1;
    fqz_gparams *g = malloc(sizeof *g);
    if (!g) return -1;
    int hp = read_params(g, in + k, in_len - (size_t)k);
    if (hp < 0) { free(g); return -1; }
    fqz_models M;
    if (models_new(&M, g) < 0) { free(g); return -1; }
    rc_t rc;
    rc_dec_start(&rc, in + k + hp, in + in_len);
    fqz_state st = {0, 0, 0, 0, 0};
    const fqz_param *pm = &g->p[0];
    uint32_t i = 0, last = 0, last_len = 0;
    int first_len = 1, err = 0;
    size_t nrev_cap = 1024, nr = 0;
    uint32_t *rlen = malloc(nrev_cap * 4);
    uint8_t *rrev = malloc(nrev_cap);
    while (i < ulen && !err) {
        if (st.p == 0) {
            int s = 0;
            if (g->max_sel > 0) { s = model_decode(&M.sel, &rc); if (s < 0) { err = 1; break; } }
            st.s = (uint32_t)s;
            pm = &g->p[g->stab[s]];
            uint32_t len;
            if (!(pm->pflags & PF_LEN) || first_len) {
                int b0 = model_decode(&M.len[0], &rc), b1 = model_decode(&M.len[1], &rc), b2 = model_decode(&M.len[2], &rc),
                    b3 = model_decode(&M.len[3], &rc);
                if ((b0 | b1 | b2 | b3) < 0) { err = 1; break; }
                len = (uint32_t)b0 | (uint32_t)b1 << 8 | (uint32_t)b2 << 16 | (uint32_t)b3 << 24;
                first_len = 0; last_len = len;
            } else len = last_len;
            if (len == 0 || len > ulen - i) { err = 1; break; }
            int rv = 0;
            if (g->gflags & GF_REV) { rv = model_decode(&M.rev, &rc); if (rv < 0) { err = 1; break; } }
            if (nr == nrev_cap) { nrev_cap *= 2; rlen = realloc(rlen, nrev_cap * 4); rrev = realloc(rrev, nrev_cap); }
            rlen[nr] = len; rrev[nr] = (uint8_t)rv; nr++;
            if (pm->pflags & PF_DEDUP) {
                int d = model_decode(&M.dup, &rc);
                if (d < 0) { err = 1; break; }
                if (d) {
                    if (i < len) { err = 1; break; }
                    memmove(out + i, out + i - len, len);      /* non-overlapping: the previous record has the same length */
                    i += len;
                    continue;
                }
            }
            st.p = len; st.delta = 0; st.qctx = 0; st.prevq = 0;
            last = pm->context;
        }
        int Q = model_decode(&M.qual[last], &rc);
        if (Q < 0) { err = 1; break; }
        out[i++] = (uint8_t)pm->qmap[Q];
        last = update_ctx(pm, &st, (uint32_t)Q);
    }
    if (!err && (st.p != 0 || rc.overrun)) err = 1;
    if (!err && (g->gflags & GF_REV)) {
        uint32_t at = 0;
        for (size_t r = 0; r < nr; r++) { if (rrev[r]) reverse_bytes(out + at, rlen[r]); at += rlen[r]; }
    }
    if (!err && lens) { for (size_t r = 0; r < nr && r < max_rec; r++) lens[r] = rlen[r]; }
    if (nrec) *nrec = nr;
    if (out_len) *out_len = ulen;
    free(rlen); free(rrev); free(M.qual); free(g);
    return err ? -1 : 0;
}

/* ---- encoder ---------------------------------------------------------------------------------------- */
/* strat 0..3: modelled on the htscodecs presets {qbits, qshift, pbits, pshift, dbits, dshift, qloc, sloc, ploc, dloc} */
static const int PRESET[4][10] = {
    {10, 5, 4, -1, 2, 1, 0, 14, 10, 14},
    {8, 5, 7, 0, 0, 0, 0, 14, 8, 14},
    {12, 6, 2, 0, 2, 3, 0, 9, 12, 14},
    {12, 6, 0, 0, 0, 0, 0, 12, 0, 0},
};
/* opts: bit 0 = use the READ2 flag of each record as the selector (two parameter sets), bit 1 = honour the reverse flags,
 * bit 2 = allow duplicate detection, bit 3 = force a selector table, bit 4 = never use a quality map; bits 5-7 = with bit 0: 2 + that many EXTRA
 * parameter sets, record r using set (READ2 flag + 2 * (r mod ...)) -- no htscodecs strategy writes more than two, but the format allows 256 and a
 * decoder must take them (the decoder tests need such streams); every extra set gets another preset and context seed so that they really differ.
 * rflags[i]: the record's BAM flags as in fqz_slice.flags (cram_io.c:1815): 16 = reverse strand, 128 = second read; NULL = none.
 * Returns the stream length or 0 on error. */
#define FQZ_FREVERSE 16u
#define FQZ_FREAD2 128u
ORC_EXPORT size_t orc_fqz_compress_bound(size_t n, size_t nrec) { return n + n / 4 + 8 * nrec + 16384; }
ORC_EXPORT size_t orc_fqz_encode(const uint8_t *in_, size_t n, const uint32_t *lens, const uint32_t *rflags, size_t nrec, int strat,
                                 int opts, uint8_t *out)
{
    if (strat < 0 || strat > 3) return 0;
    {
        uint64_t tot = 0;
        for (size_t r = 0; r < nrec; r++) { if (lens[r] == 0) return 0; tot += lens[r]; }
        if (tot != n) return 0;
    }
    uint8_t *in = malloc(n ? n : 1);
    fqz_gparams *g = calloc(1, sizeof *g);
    if (!in || !g) { free(in); free(g); return 0; }
    memcpy(in, in_, n);
    const int do_rev = (opts & 2) && rflags;
    if (do_rev) { size_t at = 0; for (size_t r = 0; r < nrec; r++) { if (rflags[r] & FQZ_FREVERSE) reverse_bytes(in + at, lens[r]); at += lens[r]; } }
    /* survey of the data */
    int seen[256] = {0}, nsym = 0, max_q = 0, fixed = 1;
    for (size_t i = 0; i < n; i++) seen[in[i]] = 1;
    for (int i = 0; i < 256; i++) if (seen[i]) { nsym++; max_q = i; }
    size_t dups = 0;
    {
        size_t at = 0;
        for (size_t r = 0; r < nrec; r++) {
            if (lens[r] != lens[0]) fixed = 0;
            if (r && lens[r] == lens[r - 1] && !memcmp(in + at, in + at - lens[r], lens[r])) dups++;
            at += lens[r];
        }
    }
    const int do_sel = (opts & 1) && rflags, do_dedup = (opts & 4) && dups * 10 >= nrec && nrec > 1;
    g->gflags = (do_sel ? GF_MULTI : 0) | (do_rev ? GF_REV : 0) | ((opts & 8) && do_sel ? GF_STAB : 0);
    const uint32_t extra = do_sel ? (uint32_t)(opts >> 5) & 7u : 0u;
    g->nparam = do_sel ? 2 + extra : 1;
    g->max_sel = do_sel ? g->nparam - 1 : 0;
    for (uint32_t i = 0; i < 256; i++) g->stab[i] = i < g->nparam ? i : g->nparam - 1;
    uint32_t inv[256] = {0};
    for (uint32_t k = 0; k < g->nparam; k++) {
        const int *P = PRESET[(strat + (int)(k >> 1)) & 3];
        fqz_param *m = &g->p[k];
        m->context = k < 2 ? 0 : k * 0x0101u;
        m->qbits = (uint32_t)P[0]; m->qshift = (uint32_t)P[1];
        m->qloc = (uint32_t)P[6]; m->sloc = (uint32_t)P[7]; m->ploc = (uint32_t)P[8]; m->dloc = (uint32_t)P[9];
        m->pflags = (do_dedup ? PF_DEDUP : 0) | (fixed ? PF_LEN : 0) | (do_sel ? PF_SEL : 0);
        for (uint32_t i = 0; i < 256; i++) { m->qmap[i] = i; m->qtab[i] = i; }
        m->max_sym = (uint32_t)max_q;
        if (nsym <= 8 && !(opts & 16) && nsym > 0) {                  /* few distinct values: code their ranks */
            m->pflags |= PF_QMAP;
            uint32_t j = 0;
            for (uint32_t i = 0; i < 256; i++) if (seen[i]) { m->qmap[j] = i; inv[i] = j; j++; }
            m->max_sym = (uint32_t)nsym;
            m->qshift = nsym <= 2 ? 1 : nsym <= 4 ? 2 : 3;
        } else for (uint32_t i = 0; i < 256; i++) inv[i] = i;
        if (m->qbits > 12) m->qbits = 12;
        const int pbits = P[2], pshift = P[3] < 0 ? (lens[0] > 511 ? 3 : lens[0] > 255 ? 2 : lens[0] > 127 ? 1 : 0) : P[3];
        if (pbits > 0) {
            m->pflags |= PF_PTAB;
            for (uint32_t i = 0; i < 1024; i++) { uint32_t v = i >> pshift, cap = (1u << pbits) - 1u; m->ptab[i] = v < cap ? v : cap; }
        }
        const int dbits = P[4], dshift = P[5];
        if (dbits > 0) {
            m->pflags |= PF_DTAB;
            for (uint32_t i = 0; i < 256; i++) { uint32_t v = i >> dshift, cap = (1u << dbits) - 1u; m->dtab[i] = v < cap ? v : cap; }
        }
        if (strat == 2) {                                             /* a coarser quality table for the history */
            m->pflags |= PF_QTAB;
            for (uint32_t i = 0; i < 256; i++) m->qtab[i] = i < 32 ? i : 32 + (i - 32) / 4 < 63 ? 32 + (i - 32) / 4 : 63;
        }
        if (m->max_sym > g->max_sym) g->max_sym = m->max_sym;
    }
    uint8_t *op = out;
    op += put_u7(op, (uint32_t)n);
    op += write_params(g, op);
    fqz_models M;
    if (models_new(&M, g) < 0) { free(in); free(g); return 0; }
    rc_t rc;
    rc_enc_start(&rc, op);
    fqz_state st = {0, 0, 0, 0, 0};
    int first_len = 1;
    size_t at = 0;
    for (size_t r = 0; r < nrec; r++) {
        const uint32_t len = lens[r];
        uint32_t s = do_sel ? (uint32_t)((rflags[r] & FQZ_FREAD2) != 0) + 2u * (uint32_t)(r % ((g->nparam + 1u) / 2u)) : 0;
        if (s >= g->nparam) s = g->nparam - 1;
        if (g->max_sel > 0) model_encode(&M.sel, &rc, s);
        st.s = s;
        const fqz_param *pm = &g->p[g->stab[s]];
        if (!(pm->pflags & PF_LEN) || first_len) {
            model_encode(&M.len[0], &rc, len & 0xff); model_encode(&M.len[1], &rc, (len >> 8) & 0xff);
            model_encode(&M.len[2], &rc, (len >> 16) & 0xff); model_encode(&M.len[3], &rc, len >> 24);
            first_len = 0;
        }
        if (g->gflags & GF_REV) model_encode(&M.rev, &rc, (rflags[r] & FQZ_FREVERSE) != 0);
        if (pm->pflags & PF_DEDUP) {
            const int d = r && lens[r - 1] == len && !memcmp(in + at, in + at - len, len);
            model_encode(&M.dup, &rc, (uint32_t)d);
            if (d) { at += len; continue; }
        }
        st.p = len; st.delta = 0; st.qctx = 0; st.prevq = 0;
        uint32_t last = pm->context;
        for (uint32_t i = 0; i < len; i++) {
            const uint32_t q = inv[in[at + i]];
            model_encode(&M.qual[last], &rc, q);
            last = update_ctx(pm, &st, q);
        }
        at += len;
    }
    op = rc_enc_finish(&rc);
    free(M.qual); free(in); free(g);
    return (size_t)(op - out);
}

/* the array coder alone, for the tests */
ORC_EXPORT int orc_fqz_store_array(const uint32_t *array, int size, uint8_t *out) { return store_array(out, array, size); }
ORC_EXPORT int orc_fqz_read_array(const uint8_t *in, size_t n, uint32_t *array, int size) { return read_array(in, n, array, size); }

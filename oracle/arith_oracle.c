/*
 * arith_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * *** PARITY UNPINNED ***  Plain-C restatement of the CRAM 3.1 adaptive arithmetic ("range") coder
 * (CRAM block method 6), the codec behind
 *      arith_uncompress_to(in, in_size, NULL, &out_size)          cram/cram_io.c:1718
 *      arith_compress_to(in, in_size, NULL, &out_size, flags)     cram/cram_io.c:1879
 * whose implementation (htscodecs v1.6.6: arith_dynamic.c, c_range_coder.h, c_simple_model.h, pack.c,
 * varint.h) is an ABSENT git submodule of the reference, with no golden stream in the reference's
 * tests.  This file follows the published specification (hts-specs "CRAM codecs" v3.1, chapter
 * "Adaptive arithmetic coding") as summarised in SURVEY.md Appendix A.5; the flag bits are confirmed by
 * the reference (cram/cram_external.c:628-638; flag sets {1,64,9,128,129,192,193} at cram_io.c:1877).
 * Byte-level agreement with htscodecs is UNVERIFIED.
 *
 * Stream:  flags:u8  [ulen:uint7 unless NOSZ 0x10]  then
 *   STRIPE 0x08: N:u8, N x clen:uint7, N complete sub-streams (byte j of sub-stream k = in[j*N+k])
 *   PACK   0x80: nsym:u8, nsym symbols, packed_len:uint7   (then the rest works on packed data)
 *   CAT    0x20: raw bytes
 *   EXT    0x04: payload is bzip2 -- not produced; decode returns -1 here
 *   else     : max_sym:u8 (0 = 256), then the range-coder bytes.  Order = flags & 1.
 *              RLE 0x40: every literal is followed by its run length, coded in parts of 0..3
 *              (3 = "more follows") with 258 four-symbol models: context = the literal for the first
 *              part, 256 for the second, 257 afterwards.
 * Range coder: 32-bit low/range with carry propagation (cache byte + pending 0xFF count); the first
 * output byte is always 0 and the decoder primes its code register with 5 bytes.  Models: symbols kept
 * roughly sorted by frequency, +16 per hit, halved when the total exceeds 2^16-17.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))
#define F_ORDER 0x01
#define F_EXT 0x04
#define F_STRIPE 0x08
#define F_NOSZ 0x10
#define F_CAT 0x20
#define F_RLE 0x40
#define F_PACK 0x80
#include "range_model.h"

/* ---- entropy cores ------------------------------------------------------------------------------ */
static size_t enc_core(const uint8_t *in, size_t n, uint8_t *out, int order, int rle)
{
    uint32_t m = 0;
    for (size_t i = 0; i < n; i++) if (in[i] > m) m = in[i];
    m++;
    out[0] = (uint8_t)m;                                  /* 256 -> 0 */
    model_t *lit = malloc((order ? 256 : 1) * sizeof(model_t)), *run = rle ? malloc(258 * sizeof(model_t)) : NULL;
    for (int i = 0; i < (order ? 256 : 1); i++) model_init(&lit[i], m);
    for (int i = 0; rle && i < 258; i++) model_init(&run[i], 4);
    rc_t rc;
    rc_enc_start(&rc, out + 1);
    uint32_t last = 0;
    for (size_t i = 0; i < n;) {
        uint32_t c = in[i];
        model_encode(&lit[order ? last : 0], &rc, c);
        last = c;
        if (!rle) { i++; continue; }
        size_t r = 0;
        while (i + 1 + r < n && in[i + 1 + r] == c) r++;
        i += r + 1;
        uint32_t ctx = c, part;
        do {
            part = r < 3 ? (uint32_t)r : 3;
            model_encode(&run[ctx], &rc, part);
            ctx = ctx == c ? 256 : 257;
            r -= part;
        } while (part == 3);
    }
    uint8_t *e = rc_enc_finish(&rc);
    free(lit); free(run);
    return (size_t)(e - out);
}

static int dec_core(const uint8_t *in, size_t in_size, uint8_t *out, size_t n, int order, int rle)
{
    if (in_size < 1) return -1;
    uint32_t m = in[0] ? in[0] : 256;
    model_t *lit = malloc((order ? 256 : 1) * sizeof(model_t)), *run = rle ? malloc(258 * sizeof(model_t)) : NULL;
    for (int i = 0; i < (order ? 256 : 1); i++) model_init(&lit[i], m);
    for (int i = 0; rle && i < 258; i++) model_init(&run[i], 4);
    rc_t rc;
    rc_dec_start(&rc, in + 1, in + in_size);
    int rcode = 0;
    uint32_t last = 0;
    for (size_t i = 0; i < n && !rcode;) {
        int c = model_decode(&lit[order ? last : 0], &rc);
        if (c < 0) { rcode = -1; break; }
        out[i] = (uint8_t)c; last = (uint32_t)c;
        if (!rle) { i++; continue; }
        uint64_t r = 0; uint32_t ctx = last; int part;
        do {
            part = model_decode(&run[ctx], &rc);
            if (part < 0) { rcode = -1; break; }
            ctx = ctx == last ? 256 : 257;
            r += (uint32_t)part;
            if (r >= n) { rcode = -1; break; }
        } while (part == 3);
        if (rcode) break;
        if (i + 1 + r > n) { rcode = -1; break; }
        memset(out + i + 1, c, r);
        i += r + 1;
    }
    if (rc.overrun) rcode = -1;                           /* ran off the end of the input */
    free(lit); free(run);
    return rcode;
}

/* ---- PACK (same transform as rANS Nx16) ------------------------------------------------------ */
static size_t pack(const uint8_t *in, size_t n, uint8_t *meta, int *meta_len, uint8_t *out)
{
    int used[256] = {0}, map[256], nsym = 0;
    for (size_t i = 0; i < n; i++) used[in[i]] = 1;
    for (int j = 0; j < 256; j++) nsym += used[j];
    if (nsym > 16) return (size_t)-1;
    nsym = 0;
    for (int j = 0; j < 256; j++) if (used[j]) { map[j] = nsym; meta[1 + nsym] = (uint8_t)j; nsym++; }
    meta[0] = (uint8_t)nsym; *meta_len = 1 + nsym;
    if (nsym <= 1) return 0;
    int bits = nsym <= 2 ? 1 : nsym <= 4 ? 2 : 4, per = 8 / bits;
    size_t o = 0;
    for (size_t i = 0; i < n; i += per) {
        unsigned v = 0;
        for (int k = 0; k < per && i + k < n; k++) v |= (unsigned)map[in[i + k]] << (k * bits);
        out[o++] = (uint8_t)v;
    }
    return o;
}
static int unpack(const uint8_t *in, size_t n, uint8_t *out, size_t out_sz, int nsym, const uint8_t *map)
{
    if (nsym <= 1) { memset(out, nsym ? map[0] : 0, out_sz); return 0; }
    int bits = nsym <= 2 ? 1 : nsym <= 4 ? 2 : 4, per = 8 / bits;
    if (n < (out_sz + per - 1) / per) return -1;
    for (size_t i = 0; i < out_sz; i++) {
        unsigned v = (in[i / per] >> ((i % per) * bits)) & ((1u << bits) - 1);
        if ((int)v >= nsym) return -1;
        out[i] = map[v];
    }
    return 0;
}

/* ---- top level ------------------------------------------------------------------------------ */
ORC_EXPORT size_t orc_arith_compress_bound(size_t n) { return n + n / 4 + 4096;   /* order-1 on incompressible bytes expands by >10% */ }

static size_t compress_inner(const uint8_t *in, size_t n, uint8_t *out, int flags)
{
    uint8_t *cp = out + 1;
    flags &= ~F_EXT;
    if (!(flags & F_NOSZ)) cp += put_u7(cp, (uint32_t)n);
    if (flags & F_STRIPE) {
        const int S = 4;
        flags &= ~(F_PACK | F_CAT);
        out[0] = (uint8_t)flags;
        *cp++ = (uint8_t)S;
        uint8_t *tmp = malloc(n / S + 8), *sub[4]; size_t slen[4];
        for (int k = 0; k < S; k++) {
            size_t m = n / S + ((n % S) > (size_t)k);
            for (size_t j = 0; j < m; j++) tmp[j] = in[j * S + k];
            sub[k] = malloc(orc_arith_compress_bound(m));
            slen[k] = compress_inner(tmp, m, sub[k], (flags & (F_ORDER | F_RLE)) | F_NOSZ);
        }
        for (int k = 0; k < S; k++) cp += put_u7(cp, (uint32_t)slen[k]);
        for (int k = 0; k < S; k++) { memcpy(cp, sub[k], slen[k]); cp += slen[k]; free(sub[k]); }
        free(tmp);
        return (size_t)(cp - out);
    }
    uint8_t *packed = NULL;
    const uint8_t *cur = in; size_t cur_n = n;
    if (flags & F_PACK) {
        uint8_t meta[20]; int ml = 0;
        packed = malloc(n + 8);
        size_t pl = n ? pack(in, n, meta, &ml, packed) : (size_t)-1;
        if (pl == (size_t)-1) flags &= ~F_PACK;
        else { memcpy(cp, meta, ml); cp += ml; cp += put_u7(cp, (uint32_t)pl); cur = packed; cur_n = pl; }
    }
    out[0] = (uint8_t)flags;
    if (flags & F_CAT) { memcpy(cp, cur, cur_n); cp += cur_n; }
    else if (cur_n) cp += enc_core(cur, cur_n, cp, flags & F_ORDER, flags & F_RLE);
    free(packed);
    return (size_t)(cp - out);
}

ORC_EXPORT size_t orc_arith_compress(const uint8_t *in, size_t n, uint8_t *out, int flags) { return compress_inner(in, n, out, flags & 0xff); }

static int uncompress_inner(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size, long known)
{
    const uint8_t *cp = in, *end = in + in_size;
    if (in_size < 1) return -1;
    const int flags = *cp++;
    uint32_t ulen;
    if (flags & F_NOSZ) { if (known < 0) return -1; ulen = (uint32_t)known; }
    else { int k = get_u7(cp, end, &ulen); if (k < 0) return -1; cp += k; }
    if (ulen > out_cap || (known >= 0 && ulen != (uint32_t)known)) return -1;
    *out_size = ulen;
    if (flags & F_STRIPE) {
        if (cp >= end) return -1;
        int S = *cp++;
        if (S < 1 || S > 32) return -1;
        uint32_t cl[32];
        for (int k = 0; k < S; k++) { int r = get_u7(cp, end, &cl[k]); if (r < 0) return -1; cp += r; }
        uint8_t *tmp = malloc(ulen / S + 8);
        for (int k = 0; k < S; k++) {
            size_t m = ulen / S + ((ulen % S) > (uint32_t)k), got = 0;
            if ((size_t)(end - cp) < cl[k] || uncompress_inner(cp, cl[k], tmp, m, &got, (long)m) || got != m) { free(tmp); return -1; }
            for (size_t j = 0; j < m; j++) out[j * S + k] = tmp[j];
            cp += cl[k];
        }
        free(tmp);
        return 0;
    }
    int nsym = 0; uint8_t map[16] = {0}; uint32_t plen = ulen;
    if (flags & F_PACK) {
        if (cp >= end) return -1;
        nsym = *cp++;
        if (nsym > 16 || cp + nsym > end) return -1;
        memcpy(map, cp, nsym); cp += nsym;
        int k = get_u7(cp, end, &plen); if (k < 0) return -1; cp += k;
        if (plen > ulen) return -1;                           /* resource guard: packing never grows the data */
    }
    int rc = 0;
    uint8_t *stage = (flags & F_PACK) ? malloc((size_t)plen + 8) : out;
    if (flags & F_CAT) { if ((size_t)(end - cp) < plen) rc = -1; else memcpy(stage, cp, plen); }
    else if (flags & F_EXT) rc = -1;                          /* bzip2 payload: not part of this restatement */
    else if (plen) rc = dec_core(cp, (size_t)(end - cp), stage, plen, flags & F_ORDER, flags & F_RLE);
    if (!rc && (flags & F_PACK)) rc = unpack(stage, plen, out, ulen, nsym, map);
    if (stage != out) free(stage);
    return rc;
}

ORC_EXPORT int orc_arith_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size, long known)
{
    return uncompress_inner(in, in_size, out, out_cap, out_size, known);
}

/* TEST INFRASTRUCTURE ONLY -- bodies behind the headers in oracle/htscodecs_stub/htscodecs/.
 *
 * samtools/htscodecs v1.6.6 is an un-vendored submodule of the reference (/root/reference/htscodecs is empty), so the reference's
 * CRAM reader / writer (cram/ *.c, sam.c, hts.c ...) cannot link as it stands.  This file supplies the symbols those sources call
 * (call sites: cram/cram_io.c:1668,1686,1699,1718,1737,1821,1838,1859,1879,1891; cram/cram_codecs.c:1399,1520,2106,2278; hts.c:149):
 *
 *   rans_compress / rans_uncompress (CRAM 3.0 method 4)  -> oracle/rans4x8_oracle.c, the restatement PINNED on the reference's own
 *                                                           CRAM v3.0 fixtures
 *   methods 5-8 (rANS Nx16, arith, fqzcomp, tok3)         -> NULL ("codec not available"), which is what makes oracle/_ref/ref_view a
 *                                                           CRAM <= 3.0 tool.  With ORC_STUB_CODECS31=1 in the environment they are
 *                                                           forwarded to the UNPINNED restatements under oracle/ instead: the files
 *                                                           such a run writes are "CRAM 3.1 in the builder's reading of the spec" --
 *                                                           good for timing the reference's record layer on 3.1-shaped work, never
 *                                                           evidence of parity with htscodecs.
 *   hts_pack / hts_unpack / hts_rle_* (CRAM 4.0 only)     -> oracle/hts_xform_oracle.c (unpinned; not reached by 2.x / 3.x files)
 *
 * Nothing here is linked into, loaded by or executed from the product (htslib_amd/). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "htscodecs/rANS_static.h"
#include "htscodecs/rANS_static4x16.h"
#include "htscodecs/arith_dynamic.h"
#include "htscodecs/tokenise_name3.h"
#include "htscodecs/fqzcomp_qual.h"
#include "htscodecs/pack.h"
#include "htscodecs/rle.h"
#include "htscodecs/htscodecs.h"

/* oracle/ *.c */
size_t orc_rans4x8_compress_bound(size_t n);
size_t orc_rans4x8_compress(const uint8_t *in, size_t n, uint8_t *out, int order);
int orc_rans4x8_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size);
size_t orc_ransnx16_compress(const uint8_t *in, size_t n, uint8_t *out, int flags);
int orc_ransnx16_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size);
size_t orc_ransnx16_compress_bound(size_t n);
size_t orc_arith_compress(const uint8_t *in, size_t n, uint8_t *out, int flags);
int orc_arith_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size, long known);
size_t orc_arith_compress_bound(size_t n);
size_t orc_tok3_compress_bound(size_t n);
size_t orc_tok3_encode(const uint8_t *in, size_t n, uint8_t *out, int use_arith);
int orc_tok3_decode(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size);
size_t orc_fqz_compress_bound(size_t n, size_t nrec);
size_t orc_fqz_encode(const uint8_t *in_, size_t n, const uint32_t *lens, const uint32_t *rflags, size_t nrec, int strat, int opts, uint8_t *out);
int orc_fqz_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *lens, size_t max_rec, size_t *nrec);
uint8_t *orc_hts_pack(const uint8_t *data, int64_t len, uint8_t *out_meta, int *out_meta_len, uint64_t *out_len);
uint8_t *orc_hts_unpack(const uint8_t *data, int64_t len, uint8_t *out, uint64_t out_len, int nsym, const uint8_t *p);
uint8_t *orc_hts_rle_encode(const uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms, int *rle_nsyms,
                            uint8_t *out, uint64_t *out_len);
uint8_t *orc_hts_rle_decode(const uint8_t *lit, uint64_t lit_len, const uint8_t *run, uint64_t run_len, const uint8_t *rle_syms,
                            int rle_nsyms, uint8_t *out, uint64_t *out_len);

const char *htscodecs_version(void) { return "absent (oracle stub: rANS 4x8 only)"; }

static int codecs31(void)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("ORC_STUB_CODECS31"); on = e && *e == '1'; }
    return on;
}

/* ---- CRAM 3.0 rANS 4x8: order byte, compressed size le32, plain size le32 ------------------------------------ */
unsigned char *rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order)
{
    uint8_t *out = malloc(orc_rans4x8_compress_bound(in_size));
    if (!out) return NULL;
    size_t n = orc_rans4x8_compress(in, in_size, out, order);
    if (!n) { free(out); return NULL; }
    *out_size = (unsigned int)n;
    return out;
}

unsigned char *rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size)
{
    if (in_size < 9) return NULL;
    uint32_t usz = in[5] | in[6] << 8 | in[7] << 16 | (uint32_t)in[8] << 24;
    if (usz > (1u << 31) - 64) return NULL;
    uint8_t *out = malloc((size_t)usz + 1);
    size_t got = 0;
    if (!out) return NULL;
    if (orc_rans4x8_uncompress(in, in_size, out, usz, &got) != 0 || got != usz) { free(out); return NULL; }
    *out_size = usz;
    return out;
}

/* ---- CRAM 3.1 methods: absent unless ORC_STUB_CODECS31=1 (then: the unpinned restatements) ---------------------------------- */
static int get_u7(const uint8_t *p, const uint8_t *e, uint32_t *v)
{
    uint32_t x = 0; int n = 0; uint8_t c;
    do { if (p >= e || n >= 5) return 0; c = *p++; x = (x << 7) | (c & 0x7f); n++; } while (c & 0x80);
    *v = x;
    return n;
}

unsigned char *rans_compress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order)
{
    if (!codecs31()) return NULL;
    int flags = order & 0xff;
    if ((order & RANS_ORDER_SIMD_AUTO) && in_size >= 20000) flags |= RANS_ORDER_X32;   /* large series go 32-way, small ones 4-way */
    uint8_t *out = malloc(orc_ransnx16_compress_bound(in_size));
    if (!out) return NULL;
    size_t n = orc_ransnx16_compress(in, in_size, out, flags);
    if (!n) { free(out); return NULL; }
    *out_size = (unsigned int)n;
    return out;
}

static unsigned char *sized_decode(unsigned char *in, unsigned int in_size, unsigned int *out_size, int arith)
{
    uint32_t usz = 0;
    if (!codecs31() || in_size < 2) return NULL;
    if (in[0] & RANS_ORDER_NOSZ) return NULL;                    /* only inside STRIPE / tok3 sub-streams, never a whole block */
    if (!get_u7(in + 1, in + in_size, &usz)) return NULL;
    uint8_t *out = malloc((size_t)usz + 64);
    size_t got = 0;
    if (!out) return NULL;
    int rc = arith ? orc_arith_uncompress(in, in_size, out, usz, &got, -1) : orc_ransnx16_uncompress(in, in_size, out, usz, &got);
    if (rc != 0 || got != usz) { free(out); return NULL; }
    *out_size = usz;
    return out;
}
unsigned char *rans_uncompress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size) { return sized_decode(in, in_size, out_size, 0); }

unsigned char *arith_compress_to(unsigned char *in, unsigned int in_size, unsigned char *out_, unsigned int *out_size, int order)
{
    if (!codecs31() || out_) return NULL;
    uint8_t *out = malloc(orc_arith_compress_bound(in_size));
    if (!out) return NULL;
    size_t n = orc_arith_compress(in, in_size, out, order & 0xff);
    if (!n) { free(out); return NULL; }
    *out_size = (unsigned int)n;
    return out;
}
unsigned char *arith_uncompress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size)
{
    if (out) return NULL;
    return sized_decode(in, in_size, out_size, 1);
}

/* orc_tok3_encode keeps its token streams in static storage: one caller at a time */
static pthread_mutex_t tok3_lock = PTHREAD_MUTEX_INITIALIZER;
uint8_t *tok3_encode_names(char *blk, int len, int level, int use_arith, int *out_len, int *last_start_p)
{
    (void)level;
    if (!codecs31() || last_start_p) return NULL;
    uint8_t *out = malloc(orc_tok3_compress_bound((size_t)len));
    if (!out) return NULL;
    pthread_mutex_lock(&tok3_lock);
    size_t n = orc_tok3_encode((const uint8_t *)blk, (size_t)len, out, use_arith);
    pthread_mutex_unlock(&tok3_lock);
    if (!n) { free(out); return NULL; }
    *out_len = (int)n;
    return out;
}
uint8_t *tok3_decode_names(uint8_t *in, uint32_t sz, uint32_t *out_len)
{
    if (!codecs31() || sz < 9) return NULL;
    uint32_t usz = in[0] | in[1] << 8 | in[2] << 16 | (uint32_t)in[3] << 24;
    if (usz > (1u << 30)) return NULL;
    uint8_t *out = malloc((size_t)usz + 64);
    size_t got = 0;
    if (!out) return NULL;
    pthread_mutex_lock(&tok3_lock);
    int rc = orc_tok3_decode(in, sz, out, usz, &got);
    pthread_mutex_unlock(&tok3_lock);
    if (rc != 0 || got != usz) { free(out); return NULL; }
    *out_len = usz;
    return out;
}

char *fqz_compress(int vers, fqz_slice *s, char *in, size_t uncomp_size, size_t *comp_size, int strat, fqz_gparams *gp)
{
    (void)vers;
    if (!codecs31() || gp) return NULL;
    uint8_t *out = malloc(orc_fqz_compress_bound(uncomp_size, (size_t)s->num_records));
    if (!out) return NULL;
    size_t n = orc_fqz_encode((const uint8_t *)in, uncomp_size, s->len, s->flags, (size_t)s->num_records, strat & 3, 0, out);
    if (!n) { free(out); return NULL; }
    *comp_size = n;
    return (char *)out;
}
char *fqz_decompress(char *in, size_t comp_size, size_t *uncomp_size, int *lengths, int nlengths)
{
    uint32_t usz = 0;
    (void)lengths; (void)nlengths;
    if (!codecs31() || !get_u7((const uint8_t *)in, (const uint8_t *)in + comp_size, &usz)) return NULL;
    uint8_t *out = malloc((size_t)usz + 64);
    size_t got = 0;
    if (!out) return NULL;
    if (orc_fqz_decode((const uint8_t *)in, comp_size, out, usz, &got, NULL, 0, NULL) != 0 || got != usz) { free(out); return NULL; }
    *uncomp_size = usz;
    return (char *)out;
}

/* ---- CRAM 4.0 transforms ------------------------------------------------------------------------------------------------------ */
uint8_t *hts_pack(uint8_t *data, int64_t len, uint8_t *out_meta, int *out_meta_len, uint64_t *out_len)
{ return orc_hts_pack(data, len, out_meta, out_meta_len, out_len); }
uint8_t *hts_unpack(uint8_t *data, int64_t len, uint8_t *out, uint64_t out_len, int nsym, uint8_t *p)
{ return orc_hts_unpack(data, len, out, out_len, nsym, p); }
uint8_t *hts_rle_encode(uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms, int *rle_nsyms,
                        uint8_t *out, uint64_t *out_len)
{ return orc_hts_rle_encode(data, data_len, run, run_len, rle_syms, rle_nsyms, out, out_len); }
uint8_t *hts_rle_decode(uint8_t *lit, uint64_t lit_len, uint8_t *run, uint64_t run_len, uint8_t *rle_syms, int rle_nsyms,
                        uint8_t *out, uint64_t *out_len)
{ return orc_hts_rle_decode(lit, lit_len, run, run_len, rle_syms, rle_nsyms, out, out_len); }

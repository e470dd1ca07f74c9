/*
 * bgzf_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C CPU restatement of the BGZF block-decode path of htslib:
 *
 *   reference                                   here
 *   ---------------------------------------------------------------------
 *   bgzf.c:896-903   check_header               orc_bgzf_check_header
 *   bgzf.c:730-804   bgzf_uncompress            orc_bgzf_uncompress_block
 *   bgzf.c:557-559   hts_crc32 (-> zlib crc32)  orc_crc32
 *   bgzf.c:1485-1539 bgzf_mt_read_block framing orc_bgzf_scan
 *   bgzf.c:566       EOF marker block           orc_bgzf_eof_block
 *
 * The deflate/CRC arithmetic itself is NOT in /root/reference: htslib calls
 * the third-party zlib (system 1.2.11 here) or libdeflate (1.8) at
 * bgzf.c:733-747 / 775-793.  The inflate below restates the published
 * algorithm, RFC 1951 (DEFLATE) section 3.2, and RFC 1952 section 8 (CRC-32),
 * in the most literal form (canonical-code counting decode, one bit at a
 * time) so that it is easy to audit; speed is irrelevant.
 *
 * Pinning: tests/test_oracle.py checks this file against (1) every BGZF
 * fixture the reference's own tests hold (tests/golden/, extracted from
 * /root/reference/test by tests/golden/make_golden.py, expected bytes produced
 * by the real reference built in oracle/_ref) and (2) the real reference
 * (oracle/_ref/libref_bgzf*.so) on seeded synthetic inputs, both zlib- and
 * libdeflate-compressed.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))
/* Huffman symbols decoded by this thread (literals + length/distance pairs): bench.py reports symbols/s beside GB/s,
 * because a more compressible input makes more bytes per symbol. */
static __thread unsigned long long orc_nlit, orc_nmatch;
void orc_symbol_counts(unsigned long long *lit, unsigned long long *match, int reset)
{ if (lit) *lit = orc_nlit; if (match) *match = orc_nmatch; if (reset) orc_nlit = orc_nmatch = 0; }
#ifndef TOKSTAT_MATCH            /* instrumentation hooks for scripts/tokstats.c */
#define TOKSTAT_MATCH(len, dist) do { orc_nmatch++; } while (0)
#define TOKSTAT_LIT() do { orc_nlit++; } while (0)
#endif

/* ------------------------------------------------------------------ CRC-32 */
/* RFC 1952 section 8: reflected polynomial 0xEDB88320, init/xorout ~0. */
static uint32_t crc_table[256];
static int crc_table_ready;

static void crc_make_table(void)
{
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_table_ready = 1;
}

ORC_EXPORT uint32_t orc_crc32(uint32_t crc, const uint8_t *buf, size_t len)
{
    if (!crc_table_ready) crc_make_table();
    uint32_t c = crc ^ 0xffffffffu;
    for (size_t i = 0; i < len; i++)
        c = crc_table[(c ^ buf[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

/* ----------------------------------------------------------------- inflate */
typedef struct {
    const uint8_t *in;
    size_t in_len, in_pos;
    uint32_t bitbuf;
    int bitcnt;
    uint8_t *out;
    size_t out_cap, out_pos;
    int err;
} orc_state;

/* RFC 1951 3.1.1: data elements are packed starting at the LSB of each byte */
static int getbits(orc_state *s, int need)
{
    uint32_t val = s->bitbuf;
    while (s->bitcnt < need) {
        if (s->in_pos == s->in_len) { s->err = -1; return 0; }
        val |= (uint32_t)s->in[s->in_pos++] << s->bitcnt;
        s->bitcnt += 8;
    }
    s->bitbuf = val >> need;
    s->bitcnt -= need;
    return (int)(val & ((1u << need) - 1));
}

#define MAXBITS 15
#define MAXLCODES 286
#define MAXDCODES 30
#define FIXLCODES 288

typedef struct { short count[MAXBITS + 1]; short symbol[FIXLCODES]; } orc_huff;

/* RFC 1951 3.2.2: canonical code construction.  Returns 0 complete,
 * <0 over-subscribed, >0 incomplete. */
static int build(orc_huff *h, const short *length, int n)
{
    short offs[MAXBITS + 1];
    for (int len = 0; len <= MAXBITS; len++) h->count[len] = 0;
    for (int sym = 0; sym < n; sym++) h->count[length[sym]]++;
    if (h->count[0] == n) return 0;
    int left = 1;
    for (int len = 1; len <= MAXBITS; len++) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int len = 1; len < MAXBITS; len++) offs[len + 1] = offs[len] + h->count[len];
    for (int sym = 0; sym < n; sym++)
        if (length[sym] != 0) h->symbol[offs[length[sym]]++] = (short)sym;
    return left;
}

/* Huffman codes are packed MSB first (RFC 1951 3.1.1): read bit by bit. */
static int decode_sym(orc_state *s, const orc_huff *h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= MAXBITS; len++) {
        code |= getbits(s, 1);
        if (s->err) return -1;
        int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    s->err = -1;
    return -1;
}

static const short len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,
                                   67,83,99,115,131,163,195,227,258};
static const short len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const short dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,
                                    1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const short dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,
                                     12,12,13,13};

static int codes(orc_state *s, const orc_huff *lc, const orc_huff *dc)
{
    for (;;) {
        int sym = decode_sym(s, lc);
        if (sym < 0) return -1;
        if (sym < 256) {
            if (s->out_pos == s->out_cap) return -1;
            s->out[s->out_pos++] = (uint8_t)sym;
            TOKSTAT_LIT();
        } else if (sym == 256) {
            return 0;
        } else {
            sym -= 257;
            if (sym >= 29) return -1;
            int len = len_base[sym] + getbits(s, len_extra[sym]);
            if (s->err) return -1;
            int dsym = decode_sym(s, dc);
            if (dsym < 0 || dsym >= 30) return -1;
            size_t dist = (size_t)dist_base[dsym] + (size_t)getbits(s, dist_extra[dsym]);
            if (s->err) return -1;
            if (dist > s->out_pos) return -1;           /* before start of block */
            if (s->out_pos + (size_t)len > s->out_cap) return -1;
            TOKSTAT_MATCH(len, dist);
            while (len--) { s->out[s->out_pos] = s->out[s->out_pos - dist]; s->out_pos++; }
        }
    }
}

static int do_stored(orc_state *s)
{
    s->bitbuf = 0; s->bitcnt = 0;                        /* skip to byte boundary */
    if (s->in_pos + 4 > s->in_len) return -1;
    unsigned len = s->in[s->in_pos] | (s->in[s->in_pos + 1] << 8);
    unsigned nlen = s->in[s->in_pos + 2] | (s->in[s->in_pos + 3] << 8);
    s->in_pos += 4;
    if ((len ^ 0xffffu) != nlen) return -1;
    if (s->in_pos + len > s->in_len) return -1;
    if (s->out_pos + len > s->out_cap) return -1;
    memcpy(s->out + s->out_pos, s->in + s->in_pos, len);
    s->in_pos += len; s->out_pos += len;
    return 0;
}

static int do_fixed(orc_state *s)
{
    orc_huff lc, dc; short lengths[FIXLCODES]; int sym;
    for (sym = 0; sym < 144; sym++) lengths[sym] = 8;
    for (; sym < 256; sym++) lengths[sym] = 9;
    for (; sym < 280; sym++) lengths[sym] = 7;
    for (; sym < FIXLCODES; sym++) lengths[sym] = 8;
    build(&lc, lengths, FIXLCODES);
    for (sym = 0; sym < MAXDCODES; sym++) lengths[sym] = 5;
    build(&dc, lengths, MAXDCODES);
    return codes(s, &lc, &dc);
}

static int do_dynamic(orc_state *s)
{
    static const short order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    short lengths[MAXLCODES + MAXDCODES];
    orc_huff lc, dc;
    int nlen = getbits(s, 5) + 257;
    int ndist = getbits(s, 5) + 1;
    int ncode = getbits(s, 4) + 4;
    if (s->err || nlen > MAXLCODES || ndist > MAXDCODES) return -1;
    int index;
    for (index = 0; index < ncode; index++) lengths[order[index]] = (short)getbits(s, 3);
    for (; index < 19; index++) lengths[order[index]] = 0;
    if (s->err) return -1;
    if (build(&lc, lengths, 19) != 0) return -1;         /* must be complete */
    index = 0;
    while (index < nlen + ndist) {
        int sym = decode_sym(s, &lc);
        if (sym < 0) return -1;
        if (sym < 16) lengths[index++] = (short)sym;
        else {
            int len = 0, rep;
            if (sym == 16) {
                if (index == 0) return -1;
                len = lengths[index - 1];
                rep = 3 + getbits(s, 2);
            } else if (sym == 17) rep = 3 + getbits(s, 3);
            else rep = 11 + getbits(s, 7);
            if (s->err || index + rep > nlen + ndist) return -1;
            while (rep--) lengths[index++] = (short)len;
        }
    }
    if (lengths[256] == 0) return -1;                    /* no end-of-block code */
    int err = build(&lc, lengths, nlen);
    if (err && (err < 0 || nlen != lc.count[0] + lc.count[1])) return -1;
    err = build(&dc, lengths + nlen, ndist);
    if (err && (err < 0 || ndist != dc.count[0] + dc.count[1])) return -1;
    return codes(s, &lc, &dc);
}

/* Raw DEFLATE stream (zlib windowBits -15 equivalent, bgzf.c:775).
 * Returns 0 ok, -1 on malformed/truncated input or output overflow. */
ORC_EXPORT int orc_inflate_raw(const uint8_t *in, size_t in_len,
                               uint8_t *out, size_t out_cap,
                               size_t *out_len, size_t *in_used)
{
    orc_state s;
    memset(&s, 0, sizeof(s));
    s.in = in; s.in_len = in_len; s.out = out; s.out_cap = out_cap;
    int last, rc = 0;
    do {
        last = getbits(&s, 1);
        int type = getbits(&s, 2);
        if (s.err) { rc = -1; break; }
        rc = type == 0 ? do_stored(&s) : type == 1 ? do_fixed(&s)
           : type == 2 ? do_dynamic(&s) : -1;
        if (rc) break;
    } while (!last);
    if (out_len) *out_len = s.out_pos;
    if (in_used) *in_used = s.in_pos;
    return rc ? -1 : 0;
}

/* ----------------------------------------------------------- BGZF framing */
/* bgzf.c:64-78: 1f 8b 08 04 | mtime(4) | xfl | os | xlen=6 | 'B' 'C' 02 00 | BSIZE-1 */
ORC_EXPORT int orc_bgzf_check_header(const uint8_t *h)
{
    /* bgzf.c:896-903: magic, FEXTRA flag set, XLEN == 6, subfield BC, SLEN == 2 */
    return (h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) != 0
            && (h[10] | (h[11] << 8)) == 6
            && h[12] == 'B' && h[13] == 'C' && (h[14] | (h[15] << 8)) == 2) ? 0 : -1;
}

static const uint8_t eof_block[28] =
    "\037\213\010\4\0\0\0\0\0\377\6\0\102\103\2\0\033\0\3\0\0\0\0\0\0\0\0\0";

ORC_EXPORT const uint8_t *orc_bgzf_eof_block(void) { return eof_block; }

/* Decode ONE BGZF block (header+payload+trailer, blen = BSIZE+1 bytes).
 * Return codes follow bgzf_uncompress (bgzf.c:730-804): 0 ok, -1 inflate
 * failure (also bad header / ISIZE mismatch), -2 CRC mismatch. */
ORC_EXPORT int orc_bgzf_uncompress_block(uint8_t *dst, size_t *dlen,
                                         const uint8_t *blk, size_t blen)
{
    if (blen < 26 || orc_bgzf_check_header(blk)) return -1;
    size_t bsize = (size_t)(blk[16] | (blk[17] << 8)) + 1;
    if (bsize != blen) return -1;
    uint32_t crc = blk[blen - 8] | (blk[blen - 7] << 8) | (blk[blen - 6] << 16) | ((uint32_t)blk[blen - 5] << 24);
    uint32_t isize = blk[blen - 4] | (blk[blen - 3] << 8) | (blk[blen - 2] << 16) | ((uint32_t)blk[blen - 1] << 24);
    size_t got = 0, used = 0;
    /* slen = block_length - 18: the 8 trailer bytes are visible to inflate, bgzf.c:813-816 */
    if (orc_inflate_raw(blk + 18, blen - 18, dst, *dlen, &got, &used)) return -1;
    if (got != isize) return -1;
    *dlen = got;
    if (orc_crc32(0, dst, got) != crc) return -2;
    return 0;
}

/* Walk a BGZF byte stream; record for block i its compressed offset, its
 * compressed length (BSIZE+1) and ISIZE.  Returns number of blocks or -1 on
 * a framing error.  Pass NULL arrays to count only. (bgzf.c:1485-1539) */
ORC_EXPORT long orc_bgzf_scan(const uint8_t *buf, size_t len, size_t max_blocks,
                              uint64_t *coff, uint32_t *clen, uint32_t *ulen)
{
    size_t pos = 0; long n = 0;
    while (pos < len) {
        if (pos + 18 > len || orc_bgzf_check_header(buf + pos)) return -1;
        size_t bs = (size_t)(buf[pos + 16] | (buf[pos + 17] << 8)) + 1;
        if (bs < 26 || pos + bs > len) return -1;
        if ((size_t)n < max_blocks) {
            if (coff) coff[n] = pos;
            if (clen) clen[n] = (uint32_t)bs;
            if (ulen) ulen[n] = buf[pos + bs - 4] | (buf[pos + bs - 3] << 8)
                              | (buf[pos + bs - 2] << 16) | ((uint32_t)buf[pos + bs - 1] << 24);
        }
        n++; pos += bs;
    }
    return n;
}

/* Whole-stream decode into one buffer; returns total bytes or <0. */
ORC_EXPORT long orc_bgzf_decompress_stream(const uint8_t *buf, size_t len,
                                           uint8_t *out, size_t out_cap)
{
    size_t pos = 0, opos = 0;
    while (pos < len) {
        if (pos + 18 > len || orc_bgzf_check_header(buf + pos)) return -1;
        size_t bs = (size_t)(buf[pos + 16] | (buf[pos + 17] << 8)) + 1;
        if (pos + bs > len) return -1;
        size_t dl = out_cap - opos;
        int rc = orc_bgzf_uncompress_block(out + opos, &dl, buf + pos, bs);
        if (rc) return rc;
        opos += dl; pos += bs;
    }
    return (long)opos;
}

/* TEST INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * hts_xform_oracle.c: scalar restatement of the four htscodecs byte transforms that cram/cram_codecs.c calls for the
 * CRAM 4.0 E_XPACK / E_XRLE encodings:
 *     hts_unpack      /root/reference/cram/cram_codecs.c:1399     hts_pack        :1520
 *     hts_rle_decode  /root/reference/cram/cram_codecs.c:2106     hts_rle_encode  :2278
 * and of var_put_u64 / var_get_u64 (:2276, :2103).
 *
 * PARITY UNPINNED: htscodecs (github.com/samtools/htscodecs, the submodule recorded in /root/reference/.gitmodules; no
 * commit is pinned in the checkout and the directory is empty) is absent, and the reference holds no golden vector for
 * these functions.  The semantics below are the published ones (htscodecs/pack.h, rle.h, varint.h and the CRAM 3.1
 * codec specification, sections "Bit packing" and "Run length encoding"), anchored on the call sites above: the
 * argument order, the meaning of `nsym` as symbols-per-byte (cram_codecs.c passes 8 / nbits), the separate run /
 * literal streams of the RLE pair and the caller-supplied symbol list.  Prefixed orc_ so that nothing can link to
 * them by accident.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define ORC_EXPORT __attribute__((visibility("default")))

ORC_EXPORT int orc_var_put_u64(uint8_t *cp, const uint8_t *endp, uint64_t v)
{
    int n = 1;
    for (uint64_t t = v >> 7; t; t >>= 7) n++;
    if (endp && endp - cp < n) return 0;
    for (int k = 0; k < n; k++) cp[k] = (uint8_t)(((v >> (7 * (n - 1 - k))) & 0x7f) | (k + 1 < n ? 0x80 : 0));
    return n;
}
ORC_EXPORT int orc_var_get_u64(const uint8_t *cp, const uint8_t *endp, uint64_t *v)
{
    uint64_t x = 0;
    int n = 0;
    if (endp && cp >= endp) { *v = 0; return 0; }
    for (;;) {
        uint8_t c = cp[n++];
        x = (x << 7) | (c & 0x7f);
        if (!(c & 0x80) || n == 10 || (endp && cp + n >= endp)) break;
    }
    *v = x;
    return n;
}
static int get_u32(const uint8_t *cp, const uint8_t *endp, uint32_t *v)
{
    uint32_t x = 0;
    int n = 0;
    while (cp + n < endp) {
        uint8_t c = cp[n++];
        x = (x << 7) | (c & 0x7f);
        if (!(c & 0x80)) { *v = x; return n; }
        if (n == 5) return -1;
    }
    return -1;
}

/* <= 16 distinct values -> 1 / 2 / 4 bits each; meta = [count][symbols ascending] */
ORC_EXPORT uint8_t *orc_hts_pack(const uint8_t *data, int64_t len, uint8_t *out_meta, int *out_meta_len, uint64_t *out_len)
{
    int used[256] = {0}, code[256], n = 0;
    uint8_t *out = malloc((size_t)len + 1);
    if (!out) return NULL;
    for (int64_t i = 0; i < len; i++) used[data[i]] = 1;
    for (int j = 0; j < 256; j++) if (used[j]) n++;
    out_meta[0] = (uint8_t)n;
    if (n > 16) { memcpy(out, data, (size_t)len); *out_meta_len = 1; *out_len = (uint64_t)len; return out; }
    n = 0;
    for (int j = 0; j < 256; j++) if (used[j]) { code[j] = n; out_meta[1 + n] = (uint8_t)j; n++; }
    *out_meta_len = 1 + n;
    if (n <= 1) { *out_len = 0; return out; }
    const int bits = n <= 2 ? 1 : n <= 4 ? 2 : 4, per = 8 / bits;
    uint64_t o = 0;
    for (int64_t i = 0; i < len; i += per) {
        unsigned v = 0;
        for (int k = 0; k < per && i + k < len; k++) v |= (unsigned)code[data[i + k]] << (k * bits);
        out[o++] = (uint8_t)v;
    }
    *out_len = o;
    return out;
}

/* nsym = symbols per byte (8, 4, 2), 1 = copy, 0 = constant */
ORC_EXPORT uint8_t *orc_hts_unpack(const uint8_t *data, int64_t len, uint8_t *out, uint64_t out_len, int nsym, const uint8_t *p)
{
    if (nsym == 1) { if ((uint64_t)len < out_len) return NULL; memcpy(out, data, out_len); return out; }
    if (nsym == 0) { memset(out, p[0], out_len); return out; }
    if (nsym != 8 && nsym != 4 && nsym != 2) return NULL;
    if ((out_len + nsym - 1) / nsym > (uint64_t)len) return NULL;
    const int bits = 8 / nsym;
    for (uint64_t i = 0; i < out_len; i++) out[i] = p[(data[i / nsym] >> ((i % nsym) * bits)) & ((1u << bits) - 1)];
    return out;
}

ORC_EXPORT uint8_t *orc_hts_rle_encode(const uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms,
                                      int *rle_nsyms, uint8_t *out, uint64_t *out_len)
{
    int isr[256] = {0};
    if (!out && !(out = malloc(data_len * 2 + 1))) return NULL;
    if (*rle_nsyms) for (int k = 0; k < *rle_nsyms; k++) isr[rle_syms[k]] = 1;
    else {
        /* a symbol is worth run-length coding when its repeats outnumber its run starts */
        int64_t score[256] = {0};
        int n = 0;
        for (uint64_t i = 0; i < data_len; i++) score[data[i]] += (i && data[i] == data[i - 1]) ? 1 : -1;
        for (int j = 0; j < 256; j++) if (score[j] > 0) { isr[j] = 1; rle_syms[n++] = (uint8_t)j; }
        *rle_nsyms = n;
    }
    uint64_t o = 0, r = 0;
    for (uint64_t i = 0; i < data_len;) {
        const uint8_t c = data[i];
        out[o++] = c;
        if (isr[c]) {
            uint64_t k = 1;
            while (i + k < data_len && data[i + k] == c) k++;
            r += (uint64_t)orc_var_put_u64(run + r, NULL, k - 1);
            i += k;
        } else i++;
    }
    *run_len = r; *out_len = o;
    return out;
}

ORC_EXPORT uint8_t *orc_hts_rle_decode(const uint8_t *lit, uint64_t lit_len, const uint8_t *run, uint64_t run_len, const uint8_t *rle_syms,
                                      int rle_nsyms, uint8_t *out, uint64_t *out_len)
{
    int isr[256] = {0};
    for (int k = 0; k < rle_nsyms; k++) isr[rle_syms[k]] = 1;
    const uint8_t *rp = run, *rend = run + run_len;
    uint64_t o = 0;
    for (uint64_t i = 0; i < lit_len; i++) {
        const uint8_t c = lit[i];
        uint32_t r = 0;
        if (isr[c]) { int k = get_u32(rp, rend, &r); if (k < 0) return NULL; rp += k; }
        if (o + r + 1 > *out_len) return NULL;
        memset(out + o, c, (size_t)r + 1);
        o += (uint64_t)r + 1;
    }
    *out_len = o;
    return out;
}

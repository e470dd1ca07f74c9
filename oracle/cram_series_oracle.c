/* TEST INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * cram_series_oracle.c: scalar restatement of the ITF8 integer coding of CRAM EXTERNAL blocks.
 *   orc_itf8_decode_block   a whole block -> int32 values, the loop cram_decode_slice runs one value at a time through
 *                           cram_external_decode_int (/root/reference/cram/cram_codecs.c:350-368) and safe_itf8_get
 *                           (/root/reference/cram/cram_io.c:644-673)
 *   orc_itf8_encode_block   the inverse: cram_external_encode_int (cram_codecs.c:523-527), itf8_put (cram_io.c:277-305)
 * PINNED: tests/test_cram_series.py runs the reference's OWN safe_itf8_get / itf8_put (spliced from cram_io.c at test time into
 * a scratch harness, tests/native/gen_itf8_ref.sh) over random and edge-case blocks and requires identical values, counts, byte
 * streams and error positions; the BF / RL / AP blocks of the reference's CRAM fixtures additionally decode to the values of the
 * .sam twins.
 */
#include <stdint.h>
#include <stddef.h>
#define ORC_EXPORT __attribute__((visibility("default")))

/* returns the number of values, or -1 when the block ends inside a value (count so far in *n_ok) or out[] is full */
ORC_EXPORT long orc_itf8_decode_block(const uint8_t *in, size_t len, int32_t *out, size_t cap, size_t *n_ok)
{
    size_t p = 0, n = 0;
    while (p < len) {
        const uint8_t b = in[p];
        const unsigned L = b < 0x80 ? 1 : b < 0xc0 ? 2 : b < 0xe0 ? 3 : b < 0xf0 ? 4 : 5;
        if (p + L > len || n >= cap) { if (n_ok) *n_ok = n; return -1; }
        uint32_t v;
        switch (L) {
        case 1: v = b; break;
        case 2: v = (((uint32_t)b << 8) | in[p + 1]) & 0x3fffu; break;
        case 3: v = (((uint32_t)b << 16) | ((uint32_t)in[p + 1] << 8) | in[p + 2]) & 0x1fffffu; break;
        case 4: v = (((uint32_t)b << 24) | ((uint32_t)in[p + 1] << 16) | ((uint32_t)in[p + 2] << 8) | in[p + 3]) & 0x0fffffffu; break;
        default: v = (((uint32_t)b & 0x0f) << 28) | ((uint32_t)in[p + 1] << 20) | ((uint32_t)in[p + 2] << 12) | ((uint32_t)in[p + 3] << 4) | (in[p + 4] & 0x0f);
        }
        out[n++] = (int32_t)v;
        p += L;
    }
    if (n_ok) *n_ok = n;
    return (long)n;
}

ORC_EXPORT size_t orc_itf8_encode_block(const int32_t *in, size_t n, uint8_t *out)
{
    size_t o = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t v = (uint32_t)in[i];
        if (!(v & ~0x7fu)) out[o++] = (uint8_t)v;
        else if (!(v & ~0x3fffu)) { out[o++] = (uint8_t)((v >> 8) | 0x80); out[o++] = (uint8_t)v; }
        else if (!(v & ~0x1fffffu)) { out[o++] = (uint8_t)((v >> 16) | 0xc0); out[o++] = (uint8_t)(v >> 8); out[o++] = (uint8_t)v; }
        else if (!(v & ~0x0fffffffu)) { out[o++] = (uint8_t)((v >> 24) | 0xe0); out[o++] = (uint8_t)(v >> 16); out[o++] = (uint8_t)(v >> 8); out[o++] = (uint8_t)v; }
        else { out[o++] = (uint8_t)(0xf0 | (v >> 28)); out[o++] = (uint8_t)(v >> 20); out[o++] = (uint8_t)(v >> 12); out[o++] = (uint8_t)(v >> 4); out[o++] = (uint8_t)(v & 0x0f); }
    }
    return o;
}

/* BYTE_ARRAY_STOP: the item loop of cram_byte_array_stop_decode_char (/root/reference/cram/cram_codecs.c:3586-3624) run over a whole
 * block.  off[k] = start of item k, off[count] = end of the last item + 1.  Returns the item count, or -1 when bytes follow the last
 * stop byte (the reference function's -1 on that item) or off[] is full.  PINNED the same way as the ITF8 pair
 * (tests/native/gen_bas_ref.sh runs the reference's own function). */
ORC_EXPORT long orc_byte_array_stop_split(const uint8_t *in, size_t len, uint8_t stop, uint32_t *off, size_t cap)
{
    size_t n = 0, p = 0;
    if (cap < 1) return -1;
    off[0] = 0;
    while (p < len) {
        size_t q = p;
        while (q < len && in[q] != stop) q++;
        if (q >= len) return -1;                      /* unterminated */
        if (n + 2 > cap) return -1;
        off[++n] = (uint32_t)(q + 1);
        p = q + 1;
    }
    return (long)n;
}

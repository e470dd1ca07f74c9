/*
 * bam_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the BAM framing layer that sits directly above bgzf_read:
 *   orc_bam_header   the walk of bam_hdr_read            (reference sam.c:229-335: "BAM\1", l_text, text, n_targets,
 *                    then per target l_name (> 0), name, l_ref)
 *   orc_bam_frame    the per-record framing and sanity checks of bam_read1 (sam.c:784-866): block_len >= 32, the 32
 *                    core bytes, l_qseq >= 0, l_qname >= 1, n_cigar*4 + l_qname + (l_qseq+1)/2 + l_qseq <= block_len-32;
 *                    return codes: record count, or -2 truncated / -4 invalid like bam_read1
 *   orc_nibble2base  nibble2base_default (htslib/sam.h seq_nt16_str "=ACMGRSVTWYHKDBN"; call site sam.c:1433)
 * Pinned by tests/test_bam_frame.py against the reference's own BAM fixtures and their .bai indexes (record
 * boundaries and per-reference mapped/unmapped counts written by reference htslib).
 */
#include <stdint.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

static uint32_t le32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }

/* returns 0 and fills n_ref / first record offset, or -1 */
ORC_EXPORT int orc_bam_header(const uint8_t *b, uint64_t len, int32_t *n_ref, uint64_t *first)
{
    if (len < 12 || memcmp(b, "BAM\1", 4)) return -1;
    uint64_t p = 8 + (uint64_t)le32(b + 4);
    if (p + 4 > len) return -1;
    int32_t n = (int32_t)le32(b + p); p += 4;
    if (n < 0) return -1;
    for (int32_t i = 0; i < n; i++) {
        if (p + 4 > len) return -1;
        int32_t l = (int32_t)le32(b + p); p += 4;
        if (l <= 0 || p + (uint64_t)l + 4 > len) return -1;
        p += (uint64_t)l + 4;
    }
    *n_ref = n; *first = p;
    return 0;
}

/* rec_off[i] = offset of record i's block_len field.  Returns the number of records, -2 if the stream ends inside a
 * record, -4 if a record fails bam_read1's checks; *bad = offset of the offending record. */
ORC_EXPORT long orc_bam_frame(const uint8_t *b, uint64_t len, uint64_t first, uint64_t *rec_off, long max_rec, uint64_t *bad)
{
    uint64_t p = first; long n = 0;
    while (p < len) {
        *bad = p;
        if (p + 4 > len) return -2;
        int32_t bl = (int32_t)le32(b + p);
        if (bl < 32) return -4;
        if (p + 4 + (uint64_t)bl > len) return (p + 36 > len) ? -2 : -2;
        const uint8_t *x = b + p + 4;
        uint32_t x2 = le32(x + 8), x3 = le32(x + 12);
        uint32_t l_qname = x2 & 0xff, n_cigar = x3 & 0xffff;
        int32_t l_qseq = (int32_t)le32(x + 16);
        if (l_qseq < 0 || l_qname < 1) return -4;
        if (((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_qseq + 1) >> 1) + (uint64_t)l_qseq > (uint64_t)(bl - 32)) return -4;
        if (rec_off && n < max_rec) rec_off[n] = p;
        n++;
        p += 4 + (uint64_t)bl;
    }
    return n;
}

ORC_EXPORT void orc_nibble2base(const uint8_t *nib, char *seq, int len)
{
    static const char code[] = "=ACMGRSVTWYHKDBN";
    for (int i = 0; i < len; i++) seq[i] = code[(nib[i >> 1] >> ((~i & 1) << 2)) & 0xf];
}

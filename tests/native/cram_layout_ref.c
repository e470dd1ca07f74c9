/* TEST INFRASTRUCTURE: the layout of struct cram_block / cram_metrics as the REFERENCE's own header defines it
 * (compiled with -I/root/reference; see cram_layout_ours.c and tests/test_cram_block_front.py). */
#include <stddef.h>
#include "cram/cram.h"
#define F(t, f) offsetof(struct t, f)
size_t ref_layout(size_t *o) {
    size_t n = 0;
    o[n++] = sizeof(struct cram_block); o[n++] = F(cram_block, method); o[n++] = F(cram_block, orig_method); o[n++] = F(cram_block, content_type);
    o[n++] = F(cram_block, content_id); o[n++] = F(cram_block, comp_size); o[n++] = F(cram_block, uncomp_size); o[n++] = F(cram_block, crc32);
    o[n++] = F(cram_block, idx); o[n++] = F(cram_block, data); o[n++] = F(cram_block, alloc); o[n++] = F(cram_block, byte); o[n++] = F(cram_block, bit);
    o[n++] = F(cram_block, m); o[n++] = F(cram_block, crc32_checked); o[n++] = F(cram_block, crc_part);
    o[n++] = sizeof(struct cram_metrics); o[n++] = F(cram_metrics, trial); o[n++] = F(cram_metrics, next_trial); o[n++] = F(cram_metrics, consistency);
    o[n++] = F(cram_metrics, sz); o[n++] = F(cram_metrics, input_avg_sz); o[n++] = F(cram_metrics, input_avg_delta); o[n++] = F(cram_metrics, method);
    o[n++] = F(cram_metrics, revised_method); o[n++] = F(cram_metrics, strat); o[n++] = F(cram_metrics, cnt); o[n++] = F(cram_metrics, extra);
    o[n++] = F(cram_metrics, unpackable);
    o[n++] = RAW; o[n++] = GZIP; o[n++] = RANS; o[n++] = RANSPR; o[n++] = ARITH; o[n++] = FQZ; o[n++] = TOK3; o[n++] = GZIP_RLE; o[n++] = GZIP_1; o[n++] = FQZ_d;
    o[n++] = RANS1; o[n++] = RANS_PR1; o[n++] = RANS_PR193; o[n++] = TOKA; o[n++] = ARITH_PR1; o[n++] = ARITH_PR193;
    o[n++] = EXTERNAL; o[n++] = CORE; o[n++] = CRAM_MAX_METHOD;
    /* cram_fd / cram_slice / cram_record fields the reference-named exports read (hts_cram_gpu.h: HG_CRAM_FD_* ...) */
    o[n++] = offsetof(cram_fd, fp); o[n++] = offsetof(cram_fd, version); o[n++] = offsetof(cram_fd, level); o[n++] = offsetof(cram_fd, ignore_md5);
    o[n++] = offsetof(cram_fd, use_bz2); o[n++] = offsetof(cram_fd, use_lzma); o[n++] = offsetof(cram_fd, metrics_lock);
    o[n++] = offsetof(cram_slice, hdr); o[n++] = offsetof(cram_slice, block); o[n++] = offsetof(cram_slice, crecs); o[n++] = offsetof(cram_block_slice_hdr, num_records);
    o[n++] = sizeof(cram_record); o[n++] = offsetof(cram_record, flags); o[n++] = offsetof(cram_record, qual); o[n++] = DS_QS;
    return n;
}

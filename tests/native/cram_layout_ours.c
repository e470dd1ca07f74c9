/* TEST INFRASTRUCTURE: the layout of struct cram_block / cram_metrics as the OUR header (include/hts_cram_gpu.h) defines it
 * (see cram_layout_ref.c). */
#include <stddef.h>
#include "hts_cram_gpu.h"
#define F(t, f) offsetof(struct t, f)
size_t our_layout(size_t *o) {
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
    o[n++] = HG_CRAM_FD_FP; o[n++] = HG_CRAM_FD_VERSION; o[n++] = HG_CRAM_FD_LEVEL; o[n++] = HG_CRAM_FD_IGNORE_MD5; o[n++] = HG_CRAM_FD_USE_BZ2; o[n++] = HG_CRAM_FD_USE_LZMA;
    o[n++] = HG_CRAM_FD_METRICS_LOCK; o[n++] = HG_CRAM_SLICE_HDR; o[n++] = HG_CRAM_SLICE_BLOCK; o[n++] = HG_CRAM_SLICE_CRECS; o[n++] = HG_CRAM_SLICE_HDR_NUM_RECORDS;
    o[n++] = HG_CRAM_RECORD_SIZE; o[n++] = HG_CRAM_RECORD_FLAGS; o[n++] = HG_CRAM_RECORD_QUAL; o[n++] = HG_CRAM_DS_QS;
    return n;
}

// cram_records_dev.h -- device-side tables and columns shared by the kernels of cram_records.hip (chain decoder, pack, cram_to_bam) and
// cram_records_fast.hip (the data-parallel passes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_internal.h"
#include "cram_records_plan.h"
#include "cram_records_fast.h"

namespace hgr {

struct DevTables {
    const PlanDev *plans; const Codec *codecs; const HuffCode *huff; const int32_t *tl_off, *tl_codec, *tl_tag;
    const SliceDev *slices; uint32_t *tab; const uint8_t *data; const RefSpan *refs;
};
struct DevCols {
    int32_t *flags, *cram_flags, *ref_id, *len, *rg, *mqual, *mate_ref_id, *ncigar, *name_len;
    int64_t *apos, *aend, *mate_pos, *tlen;
    uint64_t *cigar_off, *name_off;
    uint32_t *cigar; uint8_t *names;
    uint64_t *seq_off; uint8_t *seq, *qual; unsigned long long *seq_pool; uint64_t seq_cap;   // seq == nullptr: bases / qualities not wanted
    uint64_t *aux_off; int32_t *aux_len; uint8_t *aux;                                      // aux == nullptr: not wanted
    int32_t *mate_flags, *mate_line; int64_t *explicit_tlen; uint32_t *coff, *noff, *aoff;   // scratch columns
    uint32_t *totals;                                                                        // per slice: CIGAR words, name bytes, aux bytes written, copy jobs noted
    CopyJob *jobs;                                                                           // deferred bulk copies of all slices (SliceDev::job_off); nullptr = none
};


constexpr int32_t STATUS_RETRY = 0x7fff0002;       // the data-parallel passes gave the slice up: the chain decoder decides
constexpr int32_t STATUS_SKIP = 0x7fff0001;        // pre_status of a slice this launch of the chain decoder leaves alone

// what the passes of the data-parallel path read besides DevTables / DevCols (cram_records_fast.hip)
struct FastDev {
    const FSer *ser; const int32_t *tl_tagidx; const uint32_t *ser_off, *ntag;      // per slice: first FSer, number of distinct tags
    const uint32_t *pool; const uint64_t *col_off; const uint32_t *col_n;
    const uint32_t *chunk_slice, *chunk_r0; uint32_t nchunks;
    const uint32_t *fast_list; uint32_t nfast;
    FScr Z;
    int32_t *fail, *unclean;                       // per slice
    uint64_t *tot;                                 // per slice: sums the passes need as totals (TOT_* below)
    uint64_t *seq_base;                            // per slice: where its bases start in seq[] / qual[]
    uint64_t *seq_used;                            // one word: bases of all slices the passes decoded (the chain decoder's slices follow)
    int32_t nref, want_aux;
};
enum { TOT_SEQ, TOT_NAME, TOT_WORK, TOT_CIG, TOT_AUX, TOT_OVER, TOT_N };

int launch_fast_columns(hg_ctx *ctx, const uint8_t *d_data, const hg_stream_desc *d_itf8, size_t n_itf8, const hg_stream_desc *d_stop, size_t n_stop,
                        const uint32_t *d_sum_src, size_t n_sums, uint32_t *d_pool, const uint64_t *d_col_off, uint32_t *d_col_n, int32_t *d_col_status,
                        const uint32_t *d_col_slice, int32_t *d_fail, hipStream_t s);
// Fc: a FastDev whose fast_list names the slices the CHAIN decoder decoded (their bases sit in seq_tmp / qual_tmp at seq_off[])
int launch_chain_placement(hg_ctx *ctx, const DevTables &T, const DevCols &D, const FastDev &Fc, const int32_t *d_status, const uint8_t *seq_tmp, const uint8_t *qual_tmp, hipStream_t s);
int launch_seg_scan(hg_ctx *ctx, const SliceDev *d_slices, const uint32_t *d_list, uint32_t nlist, uint32_t *cols, int ncols, uint64_t N, uint64_t *d_tot, hipStream_t s);
int launch_fast_passes(hg_ctx *ctx, const DevTables &T, const DevCols &D, const FastDev &F, uint32_t nslices, int32_t *d_status, hipStream_t s);

}  // namespace hgr

// cram_records_fast.hip -- the data-parallel CRAM record decoder on MI355X (gfx950): kernels around the per-record passes of
// cram_records_fast.h (reference: the record loop of cram_decode_slice, cram/cram_decode.c:2553-2985; mates :2140-2307).
//
//   columns   every integer block -> int32 column, every BYTE_ARRAY_STOP block -> item table (cram_series.hip), running sums of length columns
//   pass k    one thread per record, CHUNK records of one slice per workgroup (the slice's tables are wave-uniform: scalar loads)
//   sums      one workgroup per slice: exclusive prefix sums, in place, of the counts a pass left (64-bit carries; a total that does not fit
//             32 bits gives the slice to the chain decoder)
//   finish    one thread per slice: capacities, the chain decoder's run-away guard, totals; one workgroup places the slices' bases
// The passes walk a few hundred bytes per record out of L2; the launch count (about twenty small kernels) matters more than any of them
// for small batches, the writing pass (pass 5: bases, qualities, names, tags, CIGAR) dominates large ones.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include "cram_records_dev.h"
#include "cram_records_fast_plan.h"

namespace hgr {

// ---- running sums of a length column (BYTE_ARRAY_LEN with an EXTERNAL length): sums[0] = 0, sums[i + 1] = sums[i] + len[i]; a negative
//      length or a sum beyond 32 bits poisons everything behind it (0xffffffff never passes f_item's bound check) ----
__global__ __launch_bounds__(64)
void fast_sums_kernel(const uint32_t *src_col, uint32_t n_sums, uint32_t first_col, uint32_t *pool, const uint64_t *col_off, uint32_t *col_n, int32_t *col_status) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < n_sums; q += gridDim.x) {
        const uint32_t src = src_col[q], n = col_n[src];
        const uint32_t *v = pool + col_off[src];
        uint32_t *o = pool + col_off[first_col + q];
        unsigned long long run = 0; bool poison = false;
        if (lane == 0) o[0] = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint32_t x = i < n ? v[i] : 0u;
            unsigned long long inc = x;
            for (int sft = 1; sft < 64; sft <<= 1) { const unsigned long long y = __shfl_up(inc, sft, 64); if ((int)lane >= sft) inc += y; }
            const unsigned long long neg = __ballot((int32_t)x < 0 || run + inc > 0xfffffff0ull);
            const bool mine = poison || (neg & ((2ull << lane) - 1ull)) != 0ull;        // a bad value at or before my lane
            if (i < n) o[i + 1] = mine ? 0xffffffffu : (uint32_t)(run + inc);
            poison = poison || neg != 0ull;
            run += __shfl(inc, 63, 64);
        }
        if (lane == 0) { col_n[first_col + q] = n; col_status[first_col + q] = col_status[src]; }
    }
}
// a column that did not decode cleanly (a value cut by the end of its block, bytes behind the last stop byte) hands its slice over
__global__ void fast_colfail_kernel(const int32_t *col_status, const uint32_t *col_slice, uint32_t ncols, int32_t *fail) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncols && col_status[c] != 0) fail[col_slice[c]] = 1;
}

// ---- the slice's context, built from wave-uniform tables ----
struct SliceCtx { Plan P; FCtx C; Cols O; };
__device__ __forceinline__ void make_ctx(const DevTables &T, const DevCols &D, const FastDev &F, uint32_t k, SliceCtx &X) {
    const SliceDev &d = T.slices[k];
    const PlanDev &pd = T.plans[d.plan];
    Plan &P = X.P;
    P.sm = &pd.sm[0][0]; P.rn_included = pd.rn_included; P.ap_delta = pd.ap_delta; P.qs_seq_orient = pd.qs_seq_orient; P.nslots = pd.nslots; P.nTL = pd.nTL;
    P.tl_off = T.tl_off + pd.tl_off_base; P.tl_codec = T.tl_codec + pd.tl_codec_base; P.tl_tag = T.tl_tag + pd.tl_codec_base;
    P.codecs = T.codecs + pd.codec_base; P.huff = T.huff + pd.huff_base;
    FCtx &C = X.C;
    C.P = &X.P; C.ser = F.ser + F.ser_off[k]; C.tl_tagidx = F.tl_tagidx + pd.tl_codec_base; C.ntag = F.ntag[k];
    C.V = FView{T.data, F.pool, F.col_off, F.col_n};
    C.Z = F.Z;
    C.rec_off = d.rec_off; C.nrec = d.nrec; C.ref_seq_id = d.ref_seq_id; C.nref = F.nref; C.ref_seq_start = d.ref_seq_start;
    C.refs = T.refs + d.ref_first; C.nrefs = (int32_t)d.nrefs; C.decode_md = d.decode_md;
    C.cig_cap = d.cig_cap; C.name_cap = d.name_cap; C.aux_cap = d.aux_cap; C.fail = F.fail + k; C.want_aux = F.want_aux != 0;
    const uint64_t r0 = d.rec_off;
    X.O = Cols{D.flags + r0, D.cram_flags + r0, D.ref_id + r0, D.len + r0, D.rg + r0, D.mqual + r0, D.mate_flags + r0, D.mate_ref_id + r0, D.mate_line + r0,
               D.ncigar + r0, D.name_len + r0, D.coff + r0, D.noff + r0, D.apos + r0, D.aend + r0, D.mate_pos + r0, D.tlen + r0, D.explicit_tlen + r0,
               D.cigar + d.cig_off, D.names + d.name_off, D.totals + 4 * (size_t)k, D.aux ? D.aux + d.aux_off : nullptr, D.aoff + r0, D.aux ? D.aux_len + r0 : nullptr, D.seq, D.qual,
               D.seq ? D.seq_off + r0 : nullptr, nullptr, D.seq_cap};
}

enum { PH_M1, PH_M2, PH_M3, PH_M4, PH_M5, PH_XA, PH_XB, PH_XC };
template <int PH>
__global__ __launch_bounds__(FastBatch::CHUNK)
void fast_pass_kernel(DevTables T, DevCols D, FastDev F) {
    // the slice's context is the same for every lane: one copy in LDS (as locals the ~800 bytes went to per-lane scratch)
    __shared__ SliceCtx X;
    __shared__ int skip;
    const uint32_t c = blockIdx.x;
    const uint32_t k = F.chunk_slice[c], r = F.chunk_r0[c] + threadIdx.x;
    if (threadIdx.x == 0) {
        skip = PH != PH_M1 && F.fail[k];                                    // given up already: the chain decoder will redo the slice.  Read ONCE per workgroup:
        if (!skip) make_ctx(T, D, F, k, X);                                 // other workgroups may set the flag while this one runs, and the barrier below needs every lane
    }
    __syncthreads();
    if (skip) return;
    if (r >= (uint32_t)X.C.nrec) return;
    if (PH == PH_M1) fast_m1(X.C, r);
    else if (PH == PH_M2) fast_m2(X.C, r, D.noff);
    else if (PH == PH_M3) fast_m3(X.C, r);
    else if (PH == PH_M4) fast_m4(X.C, r, D.coff, D.aoff, D.aux);
    else if (PH == PH_M5) fast_m5(X.C, r, X.O, F.seq_base[k]);
    else if (PH == PH_XA) fast_xa(X.C, r, X.O, F.unclean + k);
    else if (PH == PH_XB) {
        if (!F.unclean[k]) fast_xb(X.C, r, X.O);
        else if (r == 0 && xref(X.O, X.C.nrec)) F.fail[k] = 1;              // not a set of simple paths: the chain decoder's own serial pass, on one lane
    }
    else if (PH == PH_XC) { if (!F.unclean[k]) fast_xc(X.C, r, X.O); }
}

// ---- exclusive prefix sums per slice, in place; one workgroup per slice ----
constexpr int SCAN_NT = 1024;
struct ScanArgs {
    uint32_t *cols[8]; int32_t tot[8]; int ncols;      // plain columns; tot[i] >= 0: the column's total goes to F.tot[slice][tot[i]]
    uint32_t *fam; int fam_n;                          // a family of columns fam + j * N: fam_n of them, or (fam_n < 0) as many as the slice has tags
    int do_ap;                                         // AP deltas -> positions (inclusive, from the slice's start), for slices whose header says delta
    uint64_t *tot_all; int tot_stride;                 // optional: the total of EVERY column, tot_all[list position * tot_stride + column]
};
__device__ __forceinline__ unsigned long long block_exscan(unsigned long long x, unsigned long long *lds, unsigned long long &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long inc = x;
    for (int sft = 1; sft < 64; sft <<= 1) { const unsigned long long y = __shfl_up(inc, sft, 64); if (lane >= sft) inc += y; }
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (int w = 0; w < SCAN_NT / 64; w++) { const unsigned long long t = lds[w]; if (w < wv) before += t; all += t; }
    __syncthreads();
    total = all;
    return before + inc - x;
}
__global__ __launch_bounds__(SCAN_NT)
void fast_scan_kernel(DevTables T, FastDev F, ScanArgs A) {
    __shared__ unsigned long long lds[SCAN_NT / 64];
    for (uint32_t q = blockIdx.x; q < F.nfast; q += gridDim.x) {
        const uint32_t k = F.fast_list[q];
        const SliceDev &d = T.slices[k];
        const uint64_t base = d.rec_off; const uint32_t n = (uint32_t)d.nrec;
        const int nfam = A.fam ? (A.fam_n < 0 ? (int)F.ntag[k] : A.fam_n) : 0;
        bool over = false;
        for (int ci = 0; ci < A.ncols + nfam; ci++) {
            uint32_t *col = (ci < A.ncols ? A.cols[ci] : A.fam + (uint64_t)(ci - A.ncols) * F.Z.N) + base;
            unsigned long long carry = 0;
            for (uint32_t i0 = 0; i0 < n; i0 += SCAN_NT) {
                const uint32_t i = i0 + threadIdx.x;
                const unsigned long long x = i < n ? col[i] : 0ull;
                unsigned long long tot;
                const unsigned long long ex = block_exscan(x, lds, tot);
                if (i < n) col[i] = (uint32_t)(carry + ex);
                carry += tot;
            }
            if (carry > 0xffffffffull) over = true;
            if (ci < A.ncols && A.tot[ci] >= 0 && threadIdx.x == 0) F.tot[(size_t)k * TOT_N + A.tot[ci]] = carry;
            if (A.tot_all && threadIdx.x == 0) A.tot_all[(size_t)q * A.tot_stride + ci] = carry > 0xffffffffull ? ~0ull : carry;
        }
        if (A.do_ap && T.plans[d.plan].ap_delta) {
            long long carry = d.ref_seq_start;
            int64_t *col = F.Z.ap + base;
            for (uint32_t i0 = 0; i0 < n; i0 += SCAN_NT) {
                const uint32_t i = i0 + threadIdx.x;
                const long long x = i < n ? col[i] : 0ll;
                unsigned long long tot;
                const unsigned long long ex = block_exscan((unsigned long long)x, lds, tot);          // two's complement: sums of signed values wrap the same way
                if (i < n) col[i] = carry + (long long)ex + x;
                carry += (long long)tot;
            }
        }
        if (over && threadIdx.x == 0 && F.tot) F.tot[(size_t)k * TOT_N + TOT_OVER] = 1;
    }
}

// ---- after the sums of pass 4: does everything fit where the chain decoder would have put it?  (capacities and the run-away guard of
//      cram_records_core.h; a slice that fails here is decoded -- and judged -- by the chain decoder) ----
__global__ void fast_finish_kernel(DevTables T, DevCols D, FastDev F) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= F.nfast) return;
    const uint32_t k = F.fast_list[q];
    const SliceDev &d = T.slices[k];
    uint64_t *t = F.tot + (size_t)k * TOT_N;
    if (t[TOT_OVER] || t[TOT_NAME] > d.name_cap || t[TOT_CIG] > d.cig_cap || t[TOT_AUX] > d.aux_cap || t[TOT_WORK] > 16ull * d.cig_cap) F.fail[k] = 1;
    D.totals[4 * (size_t)k] = (uint32_t)t[TOT_CIG]; D.totals[4 * (size_t)k + 1] = (uint32_t)t[TOT_NAME]; D.totals[4 * (size_t)k + 2] = (uint32_t)t[TOT_AUX]; D.totals[4 * (size_t)k + 3] = 0;
    if (D.seq && t[TOT_SEQ] > D.seq_cap) F.fail[k] = 1;                    // more bases than the whole batch may hold: a damaged read length (must not push the neighbours out)
    if (F.fail[k] || !D.seq) t[TOT_SEQ] = 0;
}
// the slices' stretches of seq[] / qual[], in slice order (deterministic: a prefix sum, no atomics)
__global__ __launch_bounds__(SCAN_NT)
void fast_seqbase_kernel(FastDev F, int after_used) {
    __shared__ unsigned long long lds[SCAN_NT / 64];
    unsigned long long carry = after_used ? *F.seq_used : 0ull;             // the chain decoder's slices are placed behind the passes' (second call)
    __syncthreads();
    for (uint32_t i0 = 0; i0 < F.nfast; i0 += SCAN_NT) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t k = i < F.nfast ? F.fast_list[i] : 0u;
        const unsigned long long x = i < F.nfast ? F.tot[(size_t)k * TOT_N + TOT_SEQ] : 0ull;
        unsigned long long tot;
        const unsigned long long ex = block_exscan(x, lds, tot);
        if (i < F.nfast) F.seq_base[k] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *F.seq_used = carry;
}
__global__ void fast_status_kernel(FastDev F, int32_t *status) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < F.nfast) { const uint32_t k = F.fast_list[q]; status[k] = F.fail[k] ? STATUS_RETRY : 0; }
}

int launch_fast_columns(hg_ctx *ctx, const uint8_t *d_data, const hg_stream_desc *d_itf8, size_t n_itf8, const hg_stream_desc *d_stop, size_t n_stop,
                        const uint32_t *d_sum_src, size_t n_sums, uint32_t *d_pool, const uint64_t *d_col_off, uint32_t *d_col_n, int32_t *d_col_status,
                        const uint32_t *d_col_slice, int32_t *d_fail, hipStream_t s) {
    int rc;
    if ((rc = hg_cram_itf8_decode_dev(ctx, d_data, d_itf8, n_itf8, (int32_t *)d_pool, d_col_n, d_col_status, s)) != HG_OK) return rc;
    if ((rc = hg_cram_byte_array_stop_dev(ctx, d_data, d_stop, n_stop, d_pool, d_col_n + n_itf8, d_col_status + n_itf8, s)) != HG_OK) return rc;
    if (n_sums) hipLaunchKernelGGL(fast_sums_kernel, dim3((unsigned)std::min<size_t>(n_sums, (size_t)ctx->cus * 16)), dim3(64), 0, s, d_sum_src, (uint32_t)n_sums,
                                   (uint32_t)(n_itf8 + n_stop), d_pool, d_col_off, d_col_n, d_col_status);
    const size_t ncols = n_itf8 + n_stop + n_sums;
    if (ncols) hipLaunchKernelGGL(fast_colfail_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, d_col_status, d_col_slice, (uint32_t)ncols, d_fail);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

int launch_fast_passes(hg_ctx *ctx, const DevTables &T, const DevCols &D, const FastDev &F, uint32_t nslices, int32_t *d_status, hipStream_t s) {
    (void)nslices;
    if (!F.nfast) return HG_OK;
    const dim3 grid(F.nchunks), blk(FastBatch::CHUNK);
    const dim3 sgrid((unsigned)std::min<size_t>(F.nfast, (size_t)ctx->cus * 4)), sblk(SCAN_NT);
    const FScr &Z = F.Z;
    ScanArgs A;
    hipLaunchKernelGGL(fast_pass_kernel<PH_M1>, grid, blk, 0, s, T, D, F);
    A = ScanArgs{{Z.c_det, Z.c_down, Z.c_ts, Z.c_map, Z.seq_at}, {-1, -1, -1, -1, TOT_SEQ}, 5, nullptr, 0, 1, nullptr, 0};
    hipLaunchKernelGGL(fast_scan_kernel, sgrid, sblk, 0, s, T, F, A);
    hipLaunchKernelGGL(fast_pass_kernel<PH_M2>, grid, blk, 0, s, T, D, F);
    A = ScanArgs{{Z.fn, D.noff, Z.work}, {-1, TOT_NAME, TOT_WORK}, 3, Z.tag, -1, 0, nullptr, 0};
    hipLaunchKernelGGL(fast_scan_kernel, sgrid, sblk, 0, s, T, F, A);
    hipLaunchKernelGGL(fast_pass_kernel<PH_M3>, grid, blk, 0, s, T, D, F);
    A = ScanArgs{{}, {}, 0, Z.cls, NCLS, 0, nullptr, 0};
    hipLaunchKernelGGL(fast_scan_kernel, sgrid, sblk, 0, s, T, F, A);
    hipLaunchKernelGGL(fast_pass_kernel<PH_M4>, grid, blk, 0, s, T, D, F);
    A = ScanArgs{{D.coff, D.aoff}, {TOT_CIG, TOT_AUX}, 2, nullptr, 0, 0, nullptr, 0};
    hipLaunchKernelGGL(fast_scan_kernel, sgrid, sblk, 0, s, T, F, A);
    hipLaunchKernelGGL(fast_finish_kernel, dim3((F.nfast + 255) / 256), dim3(256), 0, s, T, D, F);
    hipLaunchKernelGGL(fast_seqbase_kernel, dim3(1), dim3(SCAN_NT), 0, s, F, 0);
    hipLaunchKernelGGL(fast_pass_kernel<PH_M5>, grid, blk, 0, s, T, D, F);
    hipLaunchKernelGGL(fast_pass_kernel<PH_XA>, grid, blk, 0, s, T, D, F);
    hipLaunchKernelGGL(fast_pass_kernel<PH_XB>, grid, blk, 0, s, T, D, F);
    hipLaunchKernelGGL(fast_pass_kernel<PH_XC>, grid, blk, 0, s, T, D, F);
    hipLaunchKernelGGL(fast_status_kernel, dim3((F.nfast + 255) / 256), dim3(256), 0, s, F, d_status);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

// ---- deterministic placement of the CHAIN decoder's bases: its slices take their bytes from one pool in whatever order the wavefronts get
//      there; afterwards every record is copied to slice order behind the passes' slices (prefix sum of the final read lengths) ----
__global__ __launch_bounds__(256)
void chain_len_kernel(DevTables T, DevCols D, FastDev F, const int32_t *status) {
    for (uint32_t q = blockIdx.x; q < F.nfast; q += gridDim.x) {
        const uint32_t k = F.fast_list[q];
        const SliceDev &d = T.slices[k];
        const bool ok = status[k] == 0;
        for (uint32_t r = threadIdx.x; r < (uint32_t)d.nrec; r += 256) F.Z.seq_at[d.rec_off + r] = ok && D.len[d.rec_off + r] > 0 ? (uint32_t)D.len[d.rec_off + r] : 0u;
    }
}
__global__ __launch_bounds__(256)
void chain_reorder_kernel(DevTables T, DevCols D, FastDev F, const int32_t *status, const uint8_t *seq_tmp, const uint8_t *qual_tmp) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t q = blockIdx.x; q < F.nfast; q += gridDim.x) {
        const uint32_t k = F.fast_list[q];
        if (status[k] != 0) continue;
        const SliceDev &d = T.slices[k];
        const uint64_t base = F.seq_base[k];
        for (uint32_t r = wv; r < (uint32_t)d.nrec; r += 4) {              // one wavefront per record
            const uint64_t g = d.rec_off + r, from = D.seq_off[g], to = base + F.Z.seq_at[g];
            const uint32_t len = D.len[g] > 0 ? (uint32_t)D.len[g] : 0u;
            if (to + len <= D.seq_cap) for (uint32_t i = lane; i < len; i += 64) { D.seq[to + i] = seq_tmp[from + i]; D.qual[to + i] = qual_tmp[from + i]; }
            hg::wave_sync();
            if (lane == 0) D.seq_off[g] = to;
        }
    }
}
int launch_chain_placement(hg_ctx *ctx, const DevTables &T, const DevCols &D, const FastDev &Fc, const int32_t *d_status, const uint8_t *seq_tmp, const uint8_t *qual_tmp, hipStream_t s) {
    if (!Fc.nfast) return HG_OK;
    const dim3 grid((unsigned)std::min<size_t>(Fc.nfast, (size_t)ctx->cus * 8));
    hipLaunchKernelGGL(chain_len_kernel, grid, dim3(256), 0, s, T, D, Fc, d_status);
    ScanArgs A{{Fc.Z.seq_at}, {TOT_SEQ}, 1, nullptr, 0, 0, nullptr, 0};
    hipLaunchKernelGGL(fast_scan_kernel, dim3((unsigned)std::min<size_t>(Fc.nfast, (size_t)ctx->cus * 4)), dim3(SCAN_NT), 0, s, T, Fc, A);
    hipLaunchKernelGGL(fast_seqbase_kernel, dim3(1), dim3(SCAN_NT), 0, s, Fc, 1);
    hipLaunchKernelGGL(chain_reorder_kernel, grid, dim3(256), 0, s, T, D, Fc, d_status, seq_tmp, qual_tmp);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

// exclusive prefix sums, per slice, of `ncols` columns cols + j * N (records of slice k = d_slices[list[k]].rec_off .. + nrec); the totals of
// every column of every listed slice go to d_tot[k * ncols + j] (~0 = does not fit 32 bits).  Used by the encoder (cram_encode.hip).
int launch_seg_scan(hg_ctx *ctx, const SliceDev *d_slices, const uint32_t *d_list, uint32_t nlist, uint32_t *cols, int ncols, uint64_t N, uint64_t *d_tot, hipStream_t s) {
    if (!nlist || !ncols) return HG_OK;
    DevTables T{}; T.slices = d_slices;
    FastDev F{}; F.fast_list = d_list; F.nfast = nlist; F.Z.N = N;
    ScanArgs A{{}, {}, 0, cols, ncols, 0, d_tot, ncols};
    hipLaunchKernelGGL(fast_scan_kernel, dim3((unsigned)std::min<size_t>(nlist, (size_t)ctx->cus * 4)), dim3(SCAN_NT), 0, s, T, F, A);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

}  // namespace hgr

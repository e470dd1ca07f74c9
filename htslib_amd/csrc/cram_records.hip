// cram_records.hip -- CRAM record decoding on MI355X (gfx950): the record loop of cram_decode_slice (reference
// cram/cram_decode.c:2346-3026) for batches of slices.  SURVEY.md 8f N2, second step (the first, cram_series.hip, decodes whole
// ITF8 / byte-array columns).
//
// What a slice is: nrec records whose fields are spread over ~28 data series; each series is coded by the codec the container's
// compression header names -- bits of the shared CORE block (HUFFMAN / BETA / GAMMA / SUBEXP) or items of an EXTERNAL block (ITF8
// integers, bytes, byte arrays), several series may share one block -- and WHICH series a record reads depends on values it has
// just read (flags, feature count, feature codes).  So a slice is one serial chain of variable-length reads; the parallelism is
// across slices (a 30x genome has ~10^5).  Mapping: one wavefront per slice; lane 0 walks the chain with cram_records_core.h (the
// same source the CPU harness checks against the reference's SAM twins), then all 64 lanes turn the slice-relative CIGAR / name
// offsets into offsets of the caller's arrays.  The compression / slice headers are parsed on the host (cram_records_plan.h): a
// few hundred bytes per container.
// Bases and qualities are rebuilt when the caller passes the reference spans (cram_decode_seq's copy-and-edit, MD:Z / NM regenerated
// when decode_md asks); each record takes its len bytes from one pool with an atomic add, so the order of records in seq[] / qual[] is not
// the record order -- seq_off[] says where each one is.
// cram_to_bam follows on the device (sizes, prefix sum, one lane per record writes the BAM bytes).
// Honest limits: the record loop is chain-bound (one serial chain per slice) -- the EXTERNAL-only
// fast path (prefix sums over per-record item counts, then the column kernels of cram_series.hip) is the next step and will be
// checked against this kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <chrono>
#include <string>
#include <vector>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include "cram_records_plan.h"
#include "cram_records_fast_plan.h"
#include "cram_records_dev.h"

namespace hgr {

// Runs the record loop of slice k (one thread).
__device__ __forceinline__ int decode_one(const DevTables &T, const DevCols &D, const SliceDev &d, int32_t nref, uint32_t k, bool defer, uint32_t *tab, const Codec *codecs, HGR_LDS uint8_t *wbuf, HGR_LDS uint32_t *wpos) {
    const PlanDev &pd = T.plans[d.plan];
    Plan P;
    for (int i = 0; i < S_N; i++) P.codec_of[i] = pd.codec_of[i];
    P.sm = &pd.sm[0][0];
    P.rn_included = pd.rn_included; P.ap_delta = pd.ap_delta; P.qs_seq_orient = pd.qs_seq_orient; P.nslots = pd.nslots; P.nTL = pd.nTL;
    P.tl_off = T.tl_off + pd.tl_off_base; P.tl_codec = T.tl_codec + pd.tl_codec_base; P.tl_tag = T.tl_tag + pd.tl_codec_base; P.codecs = codecs; P.huff = T.huff + pd.huff_base;
    Slice S;
    S.data = T.data; S.blk_off = tab; S.blk_len = tab + pd.nslots; S.cursor = tab + 2 * pd.nslots;     // tab: the slice's slot table, in LDS when it fits
    S.core_off = d.core_off; S.core_len = d.core_len; S.nrec = d.nrec; S.ref_seq_id = d.ref_seq_id; S.ref_seq_start = d.ref_seq_start; S.nref = nref;
    S.cigar_cap = d.cig_cap; S.name_cap = d.name_cap; S.aux_cap = d.aux_cap; S.refs = T.refs + d.ref_first; S.nrefs = (int32_t)d.nrefs; S.decode_md = d.decode_md;
    uint32_t *totals = D.totals + 4 * (size_t)k;
    totals[0] = totals[1] = totals[2] = totals[3] = 0;
    S.jobs = defer && D.jobs ? D.jobs + d.job_off : nullptr; S.job_cap = d.job_cap;
    S.wbuf = wbuf; S.wpos = wpos;
    const uint64_t r0 = d.rec_off;
    Cols O{D.flags + r0, D.cram_flags + r0, D.ref_id + r0, D.len + r0, D.rg + r0, D.mqual + r0, D.mate_flags + r0, D.mate_ref_id + r0, D.mate_line + r0,
           D.ncigar + r0, D.name_len + r0, D.coff + r0, D.noff + r0, D.apos + r0, D.aend + r0, D.mate_pos + r0, D.tlen + r0, D.explicit_tlen + r0,
           D.cigar + d.cig_off, D.names + d.name_off, totals, D.aux ? D.aux + d.aux_off : nullptr, D.aoff + r0, D.aux ? D.aux_len + r0 : nullptr, D.seq, D.qual,
           D.seq ? D.seq_off + r0 : nullptr, D.seq_pool, D.seq_cap};
    return decode_slice(&P, &S, O);
}
// Second pass: the slices wrote their CIGAR / name / aux bytes into capacity-sized regions; pack them back to back (dense[] = where
// each slice's part starts, from a host prefix sum over the totals) and turn the slice-relative offsets of the records into offsets
// of the packed arrays.  One wavefront per slice.
struct Dense { uint32_t *cigar; uint8_t *names, *aux; const uint64_t *base; };      // base: 3 per slice
__global__ __launch_bounds__(1024)
void cram_records_pack_kernel(DevTables T, DevCols D, Dense P, uint32_t nslices, const int32_t *status) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t k = blockIdx.x; k < nslices; k += gridDim.x) {
        if (status[k] != 0) continue;
        const SliceDev d = T.slices[k];
        const uint64_t bc = P.base[3 * (size_t)k], bn = P.base[3 * (size_t)k + 1], ba = P.base[3 * (size_t)k + 2];
        const uint32_t nc = D.totals[4 * (size_t)k], nn = D.totals[4 * (size_t)k + 1], na = D.totals[4 * (size_t)k + 2];
        for (uint32_t i = tid; i < nc; i += nt) P.cigar[bc + i] = D.cigar[d.cig_off + i];
        for (uint32_t i = tid; i < nn; i += nt) P.names[bn + i] = D.names[d.name_off + i];
        if (D.aux) for (uint32_t i = tid; i < na; i += nt) P.aux[ba + i] = D.aux[d.aux_off + i];
        for (uint32_t r = tid; r < (uint32_t)d.nrec; r += nt) {
            D.cigar_off[d.rec_off + r] = bc + D.coff[d.rec_off + r];
            D.name_off[d.rec_off + r] = bn + D.noff[d.rec_off + r];
            if (D.aux) D.aux_off[d.rec_off + r] = ba + D.aoff[d.rec_off + r];
        }
    }
}

// Two mappings of slices to the machine.  WAVE: one wavefront per slice, lane 0 walks the chain and the 64 lanes rebase the offsets --
// the right shape for a handful of large slices.  LANE: one slice per lane, 64 unlike chains per wavefront; the lanes diverge at every
// codec switch, yet all SIMD slots do useful work part of the time -- the right shape for batches of thousands of slices
// (the launcher picks by the slice count; HG_CRAM_RECORDS_MODE = wave | lane overrides, profiles/r02_cram_records_probe.txt).
__global__ __launch_bounds__(64)
void cram_records_kernel(DevTables T, DevCols D, uint32_t nslices, int32_t nref, const int32_t *pre_status, int32_t *status) {
    const int lane = threadIdx.x & 63;
    for (uint32_t k = blockIdx.x; k < nslices; k += gridDim.x) {
        if (pre_status[k] != 0) continue;                                   // failed while the batch was planned, or not this launch's (STATUS_SKIP): status[] holds the verdict
        const SliceDev d = T.slices[k];
        // block offsets / lengths / cursors of the slice: every value read starts with a look at them, so they live in LDS (a global
        // table cost one extra dependent round trip per value)
        // ... and so do the codec descriptions the container's compression header gave
        __shared__ uint32_t lds_tab[3 * 96];
        __shared__ Codec lds_codecs[192];
        __shared__ uint32_t lds_win[32 * 98];                               // 128-byte read-ahead window per block (cram_records_core.h, Reader::peek)
        __shared__ uint32_t lds_wpos[98];
        const PlanDev &pdk = T.plans[d.plan];
        const uint32_t ns = (uint32_t)pdk.nslots, ncd = pdk.ncodecs;
        if (ns <= 96u && ncd <= 192u) {
            for (uint32_t i = (uint32_t)lane; i < 3u * ns; i += 64) lds_tab[i] = i < 2u * ns ? T.tab[d.tab_off + i] : 0u;
            for (uint32_t i = (uint32_t)lane; i < ncd; i += 64) lds_codecs[i] = T.codecs[pdk.codec_base + i];
            hg::wave_sync();
            if (lane == 0) status[k] = decode_one(T, D, d, nref, k, true, lds_tab, lds_codecs, (HGR_LDS uint8_t *)lds_win, (HGR_LDS uint32_t *)lds_wpos);
        } else if (lane == 0) status[k] = decode_one(T, D, d, nref, k, true, T.tab + d.tab_off, T.codecs + pdk.codec_base, nullptr, nullptr);
        hg::wave_sync();
        if (D.jobs) {                                                      // the bulk copies lane 0 noted, one per lane
            const uint32_t nj = D.totals[4 * (size_t)k + 3];
            const CopyJob *J = D.jobs + d.job_off;
            for (uint32_t j = (uint32_t)lane; j < nj; j += 64) copy_bytes(J[j].dst, J[j].src, J[j].n);
        }
    }
}
__global__ __launch_bounds__(64)
void cram_records_lane_kernel(DevTables T, DevCols D, uint32_t nslices, int32_t nref, const int32_t *pre_status, int32_t *status) {
    for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < nslices; k += gridDim.x * 64u) {
        if (pre_status[k] != 0) continue;
        const SliceDev d = T.slices[k];
        status[k] = decode_one(T, D, d, nref, k, false, T.tab + d.tab_off, T.codecs + T.plans[d.plan].codec_base, nullptr, nullptr);                   // every lane is a chain of its own here: nothing to hand the copies to
    }
}

// ---- cram_to_bam (cram_decode.c:3100-3192) + bam_set1 (sam.c) + the on-disk layout of bam_write1: decoded columns -> BAM records ----
// One wavefront per slice, one record per lane.  sizes first (so that a prefix sum can place the records), then the bytes.
struct BamIn {
    const DevCols *D; const unsigned char *rg_names; const uint32_t *rg_off;   // read-group names back to back, rg_off[nrg + 1]
    int32_t nrg;
};
// Names of records stored without one (cram_to_bam, cram_decode.c:3113-3143): the mate's name when it has one, else "<prefix>:<number>" with the
// number of the record in the file (of the earlier record of the pair, so that mates agree).  code >= 0: that number; code <= -2: copy the
// name of record -(code + 2) of the batch.  Without a prefix the record is called "*".
__device__ __forceinline__ uint32_t dec_digits(uint64_t v) { uint32_t n = 1; while (v >= 10u) { v /= 10u; n++; } return n; }
// What the writer needs to know about a record, gathered ONCE by the sizing pass (one lane per record: every column read is coalesced) into 32
// words -- the writer, one wavefront per record, then pays one round trip for the descriptor and one for the bytes instead of a chain of a
// dozen dependent column reads.
enum { BD_W0 = 0 /* 9 words of block_size + core */, BD_NL = 9, BD_NC, BD_LEN, BD_NA, BD_RG, BD_NKIND /* 0 stored name, 1 "<prefix>:<number>", 2 "*" */, BD_NAME_LO = 16, BD_NAME_HI,
       BD_NUM_LO, BD_NUM_HI, BD_CIG_LO, BD_CIG_HI, BD_SEQ_LO, BD_SEQ_HI, BD_AUX_LO, BD_AUX_HI, BD_WORDS = 32 };
__device__ __forceinline__ uint32_t reg2bin_(int64_t beg, int64_t end);
__global__ __launch_bounds__(1024)
void cram_bam_size_kernel(DevTables T, DevCols D, Dense P, const uint32_t *rg_off, int32_t nrg, uint32_t plen, uint32_t nslices, int32_t *status, uint64_t *sizes, uint32_t *desc) {
    __shared__ int bad_any;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t k = blockIdx.x; k < nslices; k += gridDim.x) {
        const SliceDev d = T.slices[k];
        bool ok = status[k] == 0;
        if (tid == 0) bad_any = 0;
        __syncthreads();
        if (ok) {                                                          // what a BAM record cannot hold (bam_set1 / bam_write1 refuse or re-route these)
            int bad = 0;
            for (uint32_t r = tid; r < (uint32_t)d.nrec; r += nt) {
                const uint64_t g = d.rec_off + r;
                const int32_t rg = D.rg[g];
                if (rg < -1 || rg >= nrg) bad |= 2;                                            // cram_to_bam returns -1 (cram_decode.c:3146-3147)
                // the name: stored, the mate's, "<prefix>:<number>" or "*" (cram_decode.c:3113-3143)
                uint32_t nkind = 0, nl = D.name_len[g] > 0 ? (uint32_t)D.name_len[g] : 0u; uint64_t nsrc = g, num = 0;
                if (!nl) {
                    const int32_t m = D.mate_line[g];
                    if (!plen) { nkind = 2; nl = 1; }
                    else if (m >= 0 && m < d.nrec && D.name_len[d.rec_off + (uint32_t)m] > 0) { nsrc = d.rec_off + (uint32_t)m; nl = (uint32_t)D.name_len[nsrc]; }
                    else { nkind = 1; num = (uint64_t)(d.record_counter + (m >= 0 && m < (int32_t)r ? m : (int32_t)r) + 1); nl = plen + 1u + dec_digits(num); }
                }
                const int32_t len = D.len[g], nc = D.ncigar[g];
                if (nl > 254 || nc > 65535 || len < 0) bad |= 1;
                const uint32_t na = (uint32_t)D.aux_len[g], flag = (uint32_t)D.flags[g];
                const uint32_t rgb = rg >= 0 && rg < nrg ? rg_off[rg + 1] - rg_off[rg] + 4u : 0u;
                const uint32_t bytes = 4u + 32u + nl + 1u + 4u * (uint32_t)nc + ((uint32_t)len + 1u) / 2u + (uint32_t)len + na + rgb;
                sizes[g] = bytes;
                const uint64_t cig_off = D.cigar_off[g];
                long long rl = 0;                                                              // reference length of the alignment (bam_cigar2rlen) -> bin
                if (!(flag & BAM_FUNMAP)) for (int32_t i = 0; i < nc && i < 65536; i++) { const uint32_t c = P.cigar[cig_off + (uint32_t)i], op = c & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += c >> 4; }
                if (rl == 0) rl = 1;
                const int64_t pos = D.apos[g] - 1, mpos = D.mate_pos[g] - 1;
                uint32_t *w = desc + g * BD_WORDS;
                w[0] = bytes - 4u; w[1] = (uint32_t)D.ref_id[g]; w[2] = (uint32_t)pos;
                w[3] = reg2bin_(pos, pos + rl) << 16 | ((uint32_t)D.mqual[g] & 0xffu) << 8 | (nl + 1u);
                w[4] = flag << 16 | ((uint32_t)nc & 0xffffu); w[5] = (uint32_t)len; w[6] = (uint32_t)D.mate_ref_id[g]; w[7] = (uint32_t)mpos; w[8] = (uint32_t)D.tlen[g];
                w[BD_NL] = nl; w[BD_NC] = (uint32_t)nc; w[BD_LEN] = (uint32_t)len; w[BD_NA] = na; w[BD_RG] = (uint32_t)rg; w[BD_NKIND] = nkind; w[15] = 0;
                const uint64_t name_off = D.name_off[nsrc], seq_off = D.seq_off[g], aux_off = D.aux_off[g];
                w[BD_NAME_LO] = (uint32_t)name_off; w[BD_NAME_HI] = (uint32_t)(name_off >> 32); w[BD_NUM_LO] = (uint32_t)num; w[BD_NUM_HI] = (uint32_t)(num >> 32);
                w[BD_CIG_LO] = (uint32_t)cig_off; w[BD_CIG_HI] = (uint32_t)(cig_off >> 32); w[BD_SEQ_LO] = (uint32_t)seq_off; w[BD_SEQ_HI] = (uint32_t)(seq_off >> 32);
                w[BD_AUX_LO] = (uint32_t)aux_off; w[BD_AUX_HI] = (uint32_t)(aux_off >> 32);
            }
            if (bad) atomicOr(&bad_any, bad);
        }
        __syncthreads();
        if (ok && bad_any) { ok = false; if (tid == 0) status[k] = (bad_any & 2) ? ERR_MALFORMED : ERR_UNSUPPORTED; }
        if (!ok) for (uint32_t r = tid; r < (uint32_t)d.nrec; r += nt) sizes[d.rec_off + r] = 0u;
        __syncthreads();
    }
}
// exclusive prefix sum of n values in place, n + 1 outputs, over the whole device: tiles of SCAN_TILE values are summed (one workgroup each),
// the tile sums are scanned by one workgroup, and a third launch scans inside the tiles from their bases.  (Round 2 used ONE workgroup for
// everything: 1.9 ms for 1 M records, a quarter of cram_to_bam's time.)
constexpr uint32_t SCAN_TILE = 4096, SCAN_TPB = 256;
__device__ __forceinline__ uint64_t wg_excl_scan_u64(uint64_t x, uint64_t *lds, uint64_t &total) {     // 256 threads
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long inc = x;
    for (int sft = 1; sft < 64; sft <<= 1) { const unsigned long long y = __shfl_up(inc, sft, 64); if (lane >= sft) inc += y; }
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (int w = 0; w < 4; w++) { const uint64_t t = lds[w]; if (w < wv) before += t; all += t; }
    __syncthreads();
    total = all;
    return before + inc - x;
}
__global__ __launch_bounds__(SCAN_TPB)
void scan_tile_sums_kernel(const uint64_t *v, uint64_t n, uint64_t *tile_sum) {
    __shared__ uint64_t lds[4];
    const uint64_t t0 = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t sum = 0;
    for (uint32_t i = threadIdx.x; i < SCAN_TILE; i += SCAN_TPB) if (t0 + i < n) sum += v[t0 + i];
    uint64_t total;
    (void)wg_excl_scan_u64(sum, lds, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_TPB)
void scan_tile_bases_kernel(uint64_t *tile_sum, uint64_t ntiles, uint64_t *grand_total) {           // one workgroup
    __shared__ uint64_t lds[4];
    uint64_t carry = 0;
    for (uint64_t i0 = 0; i0 < ntiles; i0 += SCAN_TPB) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t x = i < ntiles ? tile_sum[i] : 0;
        uint64_t total;
        const uint64_t ex = wg_excl_scan_u64(x, lds, total);
        if (i < ntiles) tile_sum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
__global__ __launch_bounds__(SCAN_TPB)
void scan_tiles_kernel(uint64_t *v, uint64_t n, const uint64_t *tile_base) {
    __shared__ uint64_t lds[4];
    const uint64_t t0 = (uint64_t)blockIdx.x * SCAN_TILE;
    constexpr uint32_t PER = SCAN_TILE / SCAN_TPB;                         // consecutive values per thread
    uint64_t x[PER], sum = 0;
    const uint64_t a = t0 + (uint64_t)threadIdx.x * PER;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) { x[j] = a + j < n ? v[a + j] : 0; sum += x[j]; }
    uint64_t total;
    uint64_t run = tile_base[blockIdx.x] + wg_excl_scan_u64(sum, lds, total);
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) { if (a + j < n) v[a + j] = run; run += x[j]; }
}
__device__ __forceinline__ uint32_t nt16_switch(uint32_t c) {             // seq_nt16_table (hts.c)
    switch (c & ~0x20u) {
    case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5; case 'S': return 6; case 'V': return 7; case 'T': return 8;
    case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12; case 'D': return 13; case 'B': return 14;
    default: return c == '=' ? 0u : 15u;
    }
}
// the same table for the letters A..Z (either case) as 4-bit entries of two 64-bit words: no branches per base
__device__ __forceinline__ uint32_t nt16(uint32_t c) {
    if (c == '=') return 0u;
    const uint32_t u = (c & ~0x20u) - 'A';                                 // 0..25 for letters
    // A=1 B=14 C=2 D=13 E=15 F=15 G=4 H=11 I=15 J=15 K=12 L=15 M=3 N=15 O=15 P=15 | Q=15 R=5 S=6 T=8 U=15 V=7 W=9 X=15 Y=10 Z=15
    const unsigned long long lo = 0xfff3fcffb4ffd2e1ull, hi = 0xfaf97f865full;
    return u < 16u ? (uint32_t)(lo >> (4u * u)) & 15u : u < 26u ? (uint32_t)(hi >> (4u * (u - 16u))) & 15u : 15u;
}
__device__ __forceinline__ uint32_t reg2bin(int64_t beg, int64_t end) {  // bam_reg2bin (sam.h)
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
__device__ __forceinline__ uint32_t reg2bin_(int64_t beg, int64_t end) { return reg2bin(beg, end); }
// One WAVEFRONT per record: one coalesced load of the record's descriptor (its words go to scalar registers), then name, CIGAR, packed
// bases, qualities and tags a byte per lane -- consecutive lanes write consecutive bytes.  Records of failed slices have size 0.
// (Round 2: one LANE per record writing ~350 bytes one after the other, 7.5 ms per million records.)
__global__ __launch_bounds__(256)
void cram_bam_write_kernel(DevCols D, Dense P, const unsigned char *rg_names, const uint32_t *rg_off, const unsigned char *prefix, uint32_t plen, uint64_t nrec,
                           const uint64_t *off, const uint32_t *desc, uint8_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * 4u;
    for (uint64_t r = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); r < nrec; r += nw) {
        const uint64_t o0 = off[r];
        if (off[r + 1] == o0) continue;
        const uint32_t dw = desc[r * BD_WORDS + (lane & 31u)];
        auto W = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)dw, i); };
        auto W64 = [&](int i) -> uint64_t { return (uint64_t)W(i) | (uint64_t)W(i + 1) << 32; };
        uint8_t *o = out + o0;
        const uint32_t nl = W(BD_NL), nc = W(BD_NC), len = W(BD_LEN), na = W(BD_NA), nkind = W(BD_NKIND);
        const int32_t rg = (int32_t)W(BD_RG);
        if (lane < 9) { uint8_t *q = o + 4u * lane; q[0] = (uint8_t)dw; q[1] = (uint8_t)(dw >> 8); q[2] = (uint8_t)(dw >> 16); q[3] = (uint8_t)(dw >> 24); }      // block_size + the 32-byte core: lane i holds word i
        uint32_t at = 36;
        if (nkind == 0) { const uint8_t *nm = P.names + W64(BD_NAME_LO); for (uint32_t i = lane; i < nl; i += 64) o[at + i] = nm[i]; }
        else if (nkind == 1) {                                               // "<prefix>:<number>"
            for (uint32_t i = lane; i < plen; i += 64) o[at + i] = prefix[i];
            if (lane == 0) { o[at + plen] = ':'; uint64_t v = W64(BD_NUM_LO); for (uint32_t i = nl; i > plen + 1u; i--) { o[at + i - 1u] = (uint8_t)('0' + v % 10u); v /= 10u; } }
        }
        else if (lane == 0) o[at] = '*';
        if (lane == 0) o[at + nl] = 0;
        at += nl + 1u;
        const uint32_t *cig = P.cigar + W64(BD_CIG_LO);
        for (uint32_t i = lane; i < nc; i += 64) { const uint32_t w = cig[i]; uint8_t *q = o + at + 4u * i; q[0] = (uint8_t)w; q[1] = (uint8_t)(w >> 8); q[2] = (uint8_t)(w >> 16); q[3] = (uint8_t)(w >> 24); }
        at += 4u * nc;
        const uint64_t so = W64(BD_SEQ_LO);
        const uint8_t *sq = D.seq + so, *ql = D.qual + so;
        for (uint32_t i = lane; i < (len + 1u) / 2u; i += 64) o[at + i] = (uint8_t)(nt16(sq[2u * i]) << 4 | (2u * i + 1u < len ? nt16(sq[2u * i + 1u]) : 0u));
        at += (len + 1u) / 2u;
        for (uint32_t i = lane; i < len; i += 64) o[at + i] = ql[i];
        at += len;
        const uint8_t *ax = P.aux + W64(BD_AUX_LO);
        for (uint32_t i = lane; i < na; i += 64) o[at + i] = ax[i];
        at += na;
        if (rg >= 0) {                                                       // RG:Z: from the read-group series (cram_decode.c:3180-3189); the sizing pass checked the range
            const uint32_t a = rg_off[rg], n = rg_off[rg + 1] - a;
            if (lane < 3) o[at + lane] = lane == 0 ? 'R' : lane == 1 ? 'G' : 'Z';
            for (uint32_t i = lane; i < n; i += 64) o[at + 3u + i] = rg_names[a + i];
            if (lane == 0) o[at + 3u + n] = 0;
        }
    }
}

}  // namespace hgr

extern "C" int hg_cram_records_bound(size_t nslices, const hg_cram_slice_blocks *slices, int major_version, uint64_t *nrec, uint64_t *cigar_cap,
                                     uint64_t *name_cap, uint64_t *aux_cap) {
    if ((nslices && !slices) || !nrec || !cigar_cap || !name_cap || !aux_cap) return HG_EINVAL;
    static_assert(sizeof(hg_cram_slice_blocks) == sizeof(hgr::SliceIn), "hg_cram_slice_blocks layout");
    hgr::Batch B;
    const int rc = hgr::batch_build(B, (const hgr::SliceIn *)slices, nslices, major_version);
    if (rc) return rc == -3 ? HG_BLOCK_EUNSUPPORTED : HG_EINVAL;
    *nrec = B.nrec; *cigar_cap = B.cig_total; *name_cap = B.name_total; *aux_cap = B.aux_total;
    return HG_OK;
}

namespace {
// HG_CRAM_RECORDS_TIMING=1: phase times of a call on stderr (stream-synchronising, for probes only)
struct PhaseTimer {
    bool on; hipStream_t s; std::chrono::steady_clock::time_point t0; std::string log;
    PhaseTimer(hipStream_t st) : on(getenv("HG_CRAM_RECORDS_TIMING") != nullptr), s(st), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto t = std::chrono::steady_clock::now();
        char b[96]; snprintf(b, sizeof b, " %s %.2f ms", what, std::chrono::duration<double, std::milli>(t - t0).count()); log += b; t0 = t;
    }
    ~PhaseTimer() { if (on) fprintf(stderr, "cram records phases:%s\n", log.c_str()); }
};
struct BamSink { const char *const *rg_names; int nrg; uint8_t *out; size_t cap; uint64_t *rec_bam_off; uint64_t *total; const char *prefix; };

// Device memory of a batch: slots of the context's scratch (host entry points: grown on demand, kept between calls) or allocations of
// its own (a staged batch that outlives the call: hg_cram_batch)
enum { M_DATA, M_OUT, M_TAB, M_PACK, M_BSZ, M_BAM, M_JOBS, M_FAST, M_POOL, M_FSCR, M_TMPSEQ, M_N };
struct RecMem {
    hg_ctx *ctx = nullptr; bool own = false; void *p[M_N] = {}; size_t cap[M_N] = {};
    int need(int slot, size_t bytes) {
        static const int ctx_slot[M_N] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11};
        if (!own) { const int rc = hg::ensure_scratch(ctx, ctx_slot[slot], bytes); if (rc) return rc; p[slot] = ctx->d_scratch[ctx_slot[slot]]; return HG_OK; }
        if (cap[slot] >= bytes) return HG_OK;
        if (p[slot]) (void)hipFree(p[slot]);
        p[slot] = nullptr; cap[slot] = 0;
        if (hipMalloc(&p[slot], bytes + 256) != hipSuccess) return HG_ENOMEM;
        cap[slot] = bytes;
        return HG_OK;
    }
    void release() { if (own) for (int i = 0; i < M_N; i++) if (p[i]) { (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; } }
};
}  // namespace

// A batch of slices staged on the device (blocks, reference spans, header tables) and everything one decoding run leaves behind.
struct hg_cram_batch {
    hgr::Batch B; hgr::FastBatch F;
    RecMem M;
    size_t nslices = 0; int major = 3, nref = 0; bool want_seq = false, want_aux = false; size_t seq_cap = 0;
    // carved images
    size_t o32[12], o64[5], ou64[3], ou32[3], ocig, onam, ost, oaux, otot, obase, oso, oseq, oqual, opool, opre;
    size_t t_off[10];
    hgr::DevTables T; hgr::DevCols D; hgr::FastDev FD;
    int32_t *d_status = nullptr, *d_pre0 = nullptr, *d_pre1 = nullptr;
    uint32_t *d_list = nullptr;                         // device copy of a slice list (chain placement)
    // results of the last run (host)
    std::vector<int32_t> status; std::vector<uint32_t> tot; std::vector<uint64_t> base;
    uint64_t used_c = 0, used_n = 0, used_a = 0, pool_used = 0, bam_bytes = 0;
    hgr::Dense PK{};
    uint64_t *d_bam_off = nullptr; uint8_t *d_bam = nullptr;
    size_t n_chain0 = 0;                                // slices the passes do not take
    void *wait_ev = nullptr;                            // a hipEvent_t the device-resident blocks are complete behind (their decoders run on another stream)
    const uint8_t *dev_lo = nullptr, *dev_hi = nullptr; // block pointers inside [dev_lo, dev_hi) are DEVICE addresses (blocks decoded in place by the fused run
                                                        // decoder, cram_file_host.hip): rec_stage gathers them on the device instead of uploading them
    std::vector<uint8_t> retried;                       // slices the passes gave up (last run)
    // column descriptors of the passes (device)
    const hg_stream_desc *d_itf8 = nullptr, *d_stop = nullptr; const uint32_t *d_sum_src = nullptr, *d_col_slice = nullptr; int32_t *d_col_status = nullptr;
};

static int rec_stage(hg_ctx *ctx, hg_cram_batch &R, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, bool want_seq, bool want_aux, size_t seq_cap) {
    hgr::Batch &B = R.B;
    static const bool timing = getenv("HG_CRAM_RECORDS_TIMING") != nullptr;
    const auto ts0 = std::chrono::steady_clock::now();
    int rc = hgr::batch_build(B, (const hgr::SliceIn *)slices, nslices, major_version);
    if (rc) return rc == -3 ? HG_BLOCK_EUNSUPPORTED : HG_EINVAL;
    const char *fp = getenv("HG_CRAM_RECORDS_PATH");                   // "chain": every slice through the chain decoder (the checker of the data-parallel passes)
    hgr::fast_build(B, R.F, !(fp && fp[0] == 'c'));
    if (timing) fprintf(stderr, "cram records stage: host plan %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count());
    hgr::FastBatch &F = R.F;
    R.nslices = nslices; R.major = major_version; R.nref = nref; R.want_seq = want_seq; R.want_aux = want_aux; R.seq_cap = seq_cap;
    R.n_chain0 = 0;
    for (size_t i = 0; i < nslices; i++) if (B.status[i] == 0 && !F.is_fast[i]) R.n_chain0++;
    hipStream_t s = ctx->stream;
    // ---- header tables: one buffer, carved
    struct Part { const void *src; size_t bytes; };
    const Part parts[10] = {{B.plans.data(), B.plans.size() * sizeof(hgr::PlanDev)}, {B.codecs.data(), B.codecs.size() * sizeof(hgr::Codec)},
                            {B.huff.data(), B.huff.size() * sizeof(hgr::HuffCode)}, {B.tl_off.data(), B.tl_off.size() * 4},
                            {B.tl_codec.data(), B.tl_codec.size() * 4}, {B.slices.data(), B.slices.size() * sizeof(hgr::SliceDev)},
                            {B.tab.data(), B.tab.size() * 4}, {B.status.data(), B.status.size() * 4},
                            {B.refs.data(), B.refs.size() * sizeof(hgr::RefSpan)}, {B.tl_tag.data(), B.tl_tag.size() * 4}};
    size_t tbytes = 0;
    for (int i = 0; i < 10; i++) { R.t_off[i] = tbytes; tbytes += (parts[i].bytes + 63) & ~(size_t)63; }
    // ---- output image
    const size_t N = B.nrec ? B.nrec : 1;
    size_t obytes = 0;
    auto carve = [&](size_t bytes) { const size_t o = obytes; obytes += (bytes + 63) & ~(size_t)63; return o; };
    for (auto &o : R.o32) o = carve(N * 4);
    for (auto &o : R.o64) o = carve(N * 8);
    for (auto &o : R.ou64) o = carve(N * 8);
    for (auto &o : R.ou32) o = carve(N * 4);
    R.ocig = carve((B.cig_total ? B.cig_total : 1) * 4); R.onam = carve(B.name_total ? B.name_total : 1); R.ost = carve(nslices * 4);
    R.oaux = carve(want_aux ? B.aux_total + 1 : 1);
    R.otot = carve(nslices * 16); R.obase = carve(nslices * 24);
    R.oso = carve(N * 8); R.oseq = carve(want_seq ? seq_cap + 1 : 1); R.oqual = carve(want_seq ? seq_cap + 1 : 1); R.opool = carve(64);
    R.opre = carve(nslices * 8 + nslices * 4);                           // two pre-status vectors, a slice list
    RecMem &M = R.M;
    if ((rc = M.need(M_DATA, B.data_bytes + 64)) || (rc = M.need(M_OUT, obytes + 64)) || (rc = M.need(M_TAB, tbytes + 64))) return rc;
    uint8_t *d_data = (uint8_t *)M.p[M_DATA], *d_out = (uint8_t *)M.p[M_OUT], *d_tab = (uint8_t *)M.p[M_TAB];
    if (timing) fprintf(stderr, "cram records stage: + device memory (data %.0f MB, columns %.0f MB) %.1f ms\n", B.data_bytes / 1e6, obytes / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count());
    bool ok;
    if (R.dev_lo) {
        const size_t nsrc = B.src_ptr.size();
        std::vector<int32_t> skip(nsrc, 0);
        std::vector<uint64_t> g_src, g_dst; std::vector<uint32_t> g_len;
        for (size_t k = 0; k < nsrc; k++)
            if (B.src_ptr[k] >= R.dev_lo && B.src_ptr[k] < R.dev_hi) {
                skip[k] = 1;
                if (B.src_len[k]) { g_src.push_back((uint64_t)(B.src_ptr[k] - R.dev_lo)); g_dst.push_back(B.src_off[k]); g_len.push_back(B.src_len[k]); }
            }
        ok = hg::stage_upload(ctx, B.src_ptr.data(), B.src_len.data(), B.src_off.data(), skip.data(), nsrc, B.data_bytes, d_data, s) == HG_OK &&
             (!R.wait_ev || hipStreamWaitEvent(s, (hipEvent_t)R.wait_ev, 0) == hipSuccess) &&
             hg::stage_gather_dev(ctx, R.dev_lo, g_src.data(), g_len.data(), d_data, g_dst.data(), g_src.size(), s) == HG_OK;
    } else ok = hg::stage_upload(ctx, B.src_ptr.data(), B.src_len.data(), B.src_off.data(), nullptr, B.src_ptr.size(), B.data_bytes, d_data, s) == HG_OK;
    for (int i = 0; i < 10; i++) if (ok && parts[i].bytes) ok = hipMemcpyAsync(d_tab + R.t_off[i], parts[i].src, parts[i].bytes, hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    R.T = hgr::DevTables{(const hgr::PlanDev *)(d_tab + R.t_off[0]), (const hgr::Codec *)(d_tab + R.t_off[1]), (const hgr::HuffCode *)(d_tab + R.t_off[2]),
                         (const int32_t *)(d_tab + R.t_off[3]), (const int32_t *)(d_tab + R.t_off[4]), (const int32_t *)(d_tab + R.t_off[9]),
                         (const hgr::SliceDev *)(d_tab + R.t_off[5]), (uint32_t *)(d_tab + R.t_off[6]), d_data, (const hgr::RefSpan *)(d_tab + R.t_off[8])};
    hgr::DevCols &D = R.D;
    int32_t **p32[11] = {&D.flags, &D.cram_flags, &D.ref_id, &D.len, &D.rg, &D.mqual, &D.mate_ref_id, &D.ncigar, &D.name_len, &D.mate_flags, &D.mate_line};
    for (int i = 0; i < 11; i++) *p32[i] = (int32_t *)(d_out + R.o32[i]);
    int64_t **p64[5] = {&D.apos, &D.aend, &D.mate_pos, &D.tlen, &D.explicit_tlen};
    for (int i = 0; i < 5; i++) *p64[i] = (int64_t *)(d_out + R.o64[i]);
    D.cigar_off = (uint64_t *)(d_out + R.ou64[0]); D.name_off = (uint64_t *)(d_out + R.ou64[1]);
    D.coff = (uint32_t *)(d_out + R.ou32[0]); D.noff = (uint32_t *)(d_out + R.ou32[1]); D.aoff = (uint32_t *)(d_out + R.ou32[2]);
    D.aux_off = (uint64_t *)(d_out + R.ou64[2]); D.aux_len = (int32_t *)(d_out + R.o32[11]); D.aux = want_aux ? d_out + R.oaux : nullptr;
    D.cigar = (uint32_t *)(d_out + R.ocig); D.names = d_out + R.onam;
    D.seq_off = (uint64_t *)(d_out + R.oso); D.seq = want_seq ? d_out + R.oseq : nullptr; D.qual = want_seq ? d_out + R.oqual : nullptr;
    D.seq_pool = (unsigned long long *)(d_out + R.opool); D.seq_cap = seq_cap;
    D.totals = (uint32_t *)(d_out + R.otot);
    D.jobs = nullptr;
    R.d_status = (int32_t *)(d_out + R.ost);
    R.d_pre0 = (int32_t *)(d_out + R.opre); R.d_pre1 = R.d_pre0 + nslices; R.d_list = (uint32_t *)(R.d_pre1 + nslices);
    {   // the chain decoder's first launch leaves the passes' slices alone
        std::vector<int32_t> pre(B.status);
        for (size_t i = 0; i < nslices; i++) if (F.is_fast[i]) pre[i] = hgr::STATUS_SKIP;
        // (a pageable source is copied by the runtime before the call returns)
        if (nslices && hipMemcpyAsync(R.d_pre0, pre.data(), nslices * 4, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    }
    // ---- tables, columns and scratch of the data-parallel passes
    memset(&R.FD, 0, sizeof R.FD);
    if (!F.fast_list.empty()) {
        const size_t ncols = F.ncols(), ni = F.itf8.size(), ns = F.stop.size(), nsum = F.sums.size(), nch = F.chunk_slice.size(), nf = F.fast_list.size();
        std::vector<hg_stream_desc> di(ni), ds(ns);
        std::vector<uint64_t> col_off(ncols + 1); std::vector<uint32_t> col_slice(ncols + 1), sum_src(nsum + 1);
        for (size_t c = 0; c < ni; c++) { memset(&di[c], 0, sizeof di[c]); di[c].in_off = F.itf8[c].in_off; di[c].in_len = F.itf8[c].in_len; di[c].out_off = F.itf8[c].pool_off; di[c].out_len = F.itf8[c].cap; col_off[c] = F.itf8[c].pool_off; col_slice[c] = F.itf8[c].slice; }
        for (size_t c = 0; c < ns; c++) { memset(&ds[c], 0, sizeof ds[c]); ds[c].in_off = F.stop[c].in_off; ds[c].in_len = F.stop[c].in_len; ds[c].out_off = F.stop[c].pool_off; ds[c].out_len = F.stop[c].cap; ds[c].reserved = F.stop[c].stop; col_off[ni + c] = F.stop[c].pool_off; col_slice[ni + c] = F.stop[c].slice; }
        for (size_t c = 0; c < nsum; c++) { col_off[ni + ns + c] = F.sums[c].pool_off; col_slice[ni + ns + c] = F.sums[c].slice; sum_src[c] = F.sums[c].src; }
        struct FPart { const void *src; size_t bytes; size_t off; };
        FPart fpart[12] = {{F.ser.data(), F.ser.size() * sizeof(hgr::FSer), 0}, {F.tl_tagidx.data(), F.tl_tagidx.size() * 4, 0}, {F.ser_off.data(), nslices * 4, 0}, {F.ntag.data(), nslices * 4, 0},
                           {col_off.data(), ncols * 8, 0}, {col_slice.data(), ncols * 4, 0}, {sum_src.data(), nsum * 4, 0}, {di.data(), ni * sizeof(hg_stream_desc), 0},
                           {ds.data(), ns * sizeof(hg_stream_desc), 0}, {F.chunk_slice.data(), nch * 4, 0}, {F.chunk_r0.data(), nch * 4, 0}, {F.fast_list.data(), nf * 4, 0}};
        size_t fb = 0;
        for (auto &q : fpart) { q.off = fb; fb += (q.bytes + 63) & ~(size_t)63; }
        const size_t o_coln = fb; fb += (ncols * 4 + 63) & ~(size_t)63;
        const size_t o_colst = fb; fb += (ncols * 4 + 63) & ~(size_t)63;
        const size_t o_fail = fb; fb += (nslices * 4 + 63) & ~(size_t)63;
        const size_t o_uncl = fb; fb += (nslices * 4 + 63) & ~(size_t)63;
        const size_t o_tot = fb; fb += (nslices * hgr::TOT_N * 8 + 63) & ~(size_t)63;
        const size_t o_sbase = fb; fb += (nslices * 8 + 63) & ~(size_t)63;
        const size_t o_sused = fb; fb += 64;
        // scratch columns: 8 x u32, 1 x i64, tags, classes, bits, pred
        const size_t NN = (N + 15) & ~(size_t)15;
        const size_t ntagc = F.ntag_max ? F.ntag_max : 1;
        const size_t zbytes = NN * 4 * (9 + ntagc + hgr::NCLS) + NN * 8 + NN + 256;
        if ((rc = M.need(M_FAST, fb + 64)) || (rc = M.need(M_POOL, F.pool_words * 4 + 256)) || (rc = M.need(M_FSCR, zbytes))) return rc;
        uint8_t *d_f = (uint8_t *)M.p[M_FAST], *d_z = (uint8_t *)M.p[M_FSCR];
        for (auto &q : fpart) if (q.bytes && hipMemcpyAsync(d_f + q.off, q.src, q.bytes, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
        if (hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;    // the vectors above go out of scope
        hgr::FastDev &FD = R.FD;
        FD.ser = (const hgr::FSer *)(d_f + fpart[0].off); FD.tl_tagidx = (const int32_t *)(d_f + fpart[1].off); FD.ser_off = (const uint32_t *)(d_f + fpart[2].off);
        FD.ntag = (const uint32_t *)(d_f + fpart[3].off); FD.pool = (const uint32_t *)M.p[M_POOL]; FD.col_off = (const uint64_t *)(d_f + fpart[4].off);
        FD.col_n = (const uint32_t *)(d_f + o_coln);
        FD.chunk_slice = (const uint32_t *)(d_f + fpart[9].off); FD.chunk_r0 = (const uint32_t *)(d_f + fpart[10].off); FD.nchunks = (uint32_t)nch;
        FD.fast_list = (const uint32_t *)(d_f + fpart[11].off); FD.nfast = (uint32_t)nf;
        FD.fail = (int32_t *)(d_f + o_fail); FD.unclean = (int32_t *)(d_f + o_uncl); FD.tot = (uint64_t *)(d_f + o_tot);
        FD.seq_base = (uint64_t *)(d_f + o_sbase); FD.seq_used = (uint64_t *)(d_f + o_sused);
        FD.nref = nref; FD.want_aux = want_aux ? 1 : 0;
        hgr::FScr &Z = FD.Z;
        uint32_t *z32 = (uint32_t *)d_z;
        Z.c_det = z32; Z.c_down = z32 + NN; Z.c_ts = z32 + 2 * NN; Z.c_map = z32 + 3 * NN; Z.seq_at = z32 + 4 * NN; Z.fn = z32 + 5 * NN; Z.work = z32 + 6 * NN; Z.aux_stored = z32 + 7 * NN;
        Z.pred = (int32_t *)(z32 + 8 * NN); Z.tag = z32 + 9 * NN; Z.cls = Z.tag + ntagc * NN;
        Z.ap = (int64_t *)(Z.cls + (size_t)hgr::NCLS * NN); Z.bits = (uint8_t *)(Z.ap + NN); Z.N = NN;
        R.d_itf8 = (const hg_stream_desc *)(d_f + fpart[7].off); R.d_stop = (const hg_stream_desc *)(d_f + fpart[8].off); R.d_sum_src = (const uint32_t *)(d_f + fpart[6].off);
        R.d_col_status = (int32_t *)(d_f + o_colst); R.d_col_slice = (const uint32_t *)(d_f + fpart[5].off);
    } else if (R.n_chain0 && want_seq) {
        // chain-only batch: the placement pass still needs a few per-slice words and one scratch column
        const size_t NN = (N + 15) & ~(size_t)15;
        size_t fb = 0;
        const size_t o_tot = fb; fb += (nslices * hgr::TOT_N * 8 + 63) & ~(size_t)63;
        const size_t o_sbase = fb; fb += (nslices * 8 + 63) & ~(size_t)63;
        const size_t o_sused = fb; fb += 64;
        if ((rc = M.need(M_FAST, fb + 64)) || (rc = M.need(M_FSCR, NN * 4 + 256))) return rc;
        uint8_t *d_f = (uint8_t *)M.p[M_FAST];
        R.FD.tot = (uint64_t *)(d_f + o_tot); R.FD.seq_base = (uint64_t *)(d_f + o_sbase); R.FD.seq_used = (uint64_t *)(d_f + o_sused);
        R.FD.Z.seq_at = (uint32_t *)M.p[M_FSCR]; R.FD.Z.N = NN;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    return HG_OK;
}

// One decoding run over a staged batch: the columns, the packed CIGAR / name / aux arrays and -- with a sink -- the BAM stream, all left on
// the device.  Host work between the launches: one look at the per-slice verdicts and totals (a few bytes per slice).
static int rec_run(hg_ctx *ctx, hg_cram_batch &R, const BamSink *bam, size_t cigar_cap, size_t name_cap, size_t aux_cap) {
    hgr::Batch &B = R.B; hgr::FastBatch &F = R.F;
    const size_t nslices = R.nslices;
    hipStream_t s = ctx->stream;
    RecMem &M = R.M;
    uint8_t *d_out = (uint8_t *)M.p[M_OUT];
    hgr::DevTables &T = R.T; hgr::DevCols D = R.D; hgr::FastDev &FD = R.FD;
    const bool want_seq = R.want_seq, want_aux = R.want_aux;
    int rc;
    PhaseTimer PT(s);
    R.status.assign(nslices, 0); R.tot.assign(nslices * 4, 0u);
    bool ok = hipMemsetAsync(d_out + R.opool, 0, 64, s) == hipSuccess && hipMemcpyAsync(R.d_status, R.d_pre0, nslices * 4, hipMemcpyDeviceToDevice, s) == hipSuccess &&
              hipMemsetAsync(d_out + R.otot, 0, nslices * 16, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    // ---- the chain decoder: slices the data-parallel passes do not take.  Their bases go to a pool of their own (placed afterwards)
    hgr::DevCols Dc = D;
    auto chain_launch = [&](size_t count, const int32_t *d_pre) -> int {
        if (want_seq && !M.p[M_TMPSEQ]) { if ((rc = M.need(M_TMPSEQ, 2 * (R.seq_cap + 64)))) return rc; }
        if (want_seq) { Dc.seq = (uint8_t *)M.p[M_TMPSEQ]; Dc.qual = Dc.seq + R.seq_cap + 64; }
        bool lane_mode = count >= 1024;                                  // measured at 8192 slices: 19.9 ms per call against 32.6 ms (profiles/r02_cram_records_probe.txt)
        if (const char *m = getenv("HG_CRAM_RECORDS_MODE")) lane_mode = m[0] == 'l';
        Dc.jobs = nullptr;
        if (!lane_mode && (want_seq || want_aux)) {                       // room for the copies the chain hands over
            if ((rc = M.need(M_JOBS, (B.job_total + 1) * sizeof(hgr::CopyJob)))) return rc;
            Dc.jobs = (hgr::CopyJob *)M.p[M_JOBS];
        }
        if (lane_mode) {
            const unsigned grid = (unsigned)std::min<size_t>((nslices + 63) / 64, (size_t)ctx->cus * 16);
            hipLaunchKernelGGL(hgr::cram_records_lane_kernel, dim3(grid), dim3(64), 0, s, T, Dc, (uint32_t)nslices, (int32_t)R.nref, d_pre, R.d_status);
        } else {
            const unsigned grid = (unsigned)std::min<size_t>(nslices, (size_t)ctx->cus * 16);
            hipLaunchKernelGGL(hgr::cram_records_kernel, dim3(grid), dim3(64), 0, s, T, Dc, (uint32_t)nslices, (int32_t)R.nref, d_pre, R.d_status);
        }
        return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
    };
    if (R.n_chain0 && (rc = chain_launch(R.n_chain0, R.d_pre0))) return rc;
    PT.mark("chain");
    // ---- the data-parallel passes
    if (FD.nfast) {
        const size_t ncols = F.ncols();
        ok = hipMemsetAsync(FD.fail, 0, nslices * 4, s) == hipSuccess && hipMemsetAsync(FD.unclean, 0, nslices * 4, s) == hipSuccess &&
             hipMemsetAsync(FD.tot, 0, nslices * hgr::TOT_N * 8, s) == hipSuccess && hipMemsetAsync(FD.Z.pred, 0, FD.Z.N * 4, s) == hipSuccess &&
             hipMemsetAsync((void *)FD.col_n, 0, ncols * 4, s) == hipSuccess;
        if (!ok) return HG_ELAUNCH;
        if ((rc = hgr::launch_fast_columns(ctx, T.data, R.d_itf8, F.itf8.size(), R.d_stop, F.stop.size(), R.d_sum_src, F.sums.size(), (uint32_t *)M.p[M_POOL], FD.col_off,
                                           (uint32_t *)FD.col_n, R.d_col_status, R.d_col_slice, FD.fail, s))) return rc;
        PT.mark("columns");
        if ((rc = hgr::launch_fast_passes(ctx, T, D, FD, (uint32_t)nslices, R.d_status, s))) return rc;
        PT.mark("passes");
    } else if (want_seq && FD.seq_used && hipMemsetAsync(FD.seq_used, 0, 8, s) != hipSuccess) return HG_ELAUNCH;
    // ---- verdicts so far.  Slices the passes gave up go through the chain decoder; so do slices of the chain decoder that found its pool of
    //      bases used up -- by neighbours that have been moved to their final place by then, or by a damaged neighbour's read lengths
    std::vector<int32_t> &status = R.status; std::vector<uint32_t> &tot = R.tot;
    auto verdicts = [&]() {
        return hipMemcpyAsync(status.data(), R.d_status, nslices * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
               hipMemcpyAsync(tot.data(), d_out + R.otot, nslices * 16, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    };
    if (!verdicts()) return HG_ELAUNCH;
    R.retried.assign(nslices, 0);
    std::vector<uint32_t> round;                                         // slices the chain decoder took in the round just run
    for (size_t i = 0; i < nslices; i++) if (B.status[i] == 0 && !F.is_fast[i]) round.push_back((uint32_t)i);
    for (int rounds = 0;; rounds++) {
        // the round's good slices: bases from the pool to slice order behind everything placed so far (deterministic: prefix sums)
        std::vector<uint32_t> good, again;
        for (uint32_t i : round) { if (status[i] == 0) good.push_back(i); else if (status[i] == hgr::ERR_POOL) again.push_back(i); }
        if (want_seq && !good.empty()) {
            if (hipMemcpyAsync(R.d_list, good.data(), good.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
            hgr::FastDev Fc = FD; Fc.fast_list = R.d_list; Fc.nfast = (uint32_t)good.size();
            if ((rc = hgr::launch_chain_placement(ctx, T, D, Fc, R.d_status, Dc.seq, Dc.qual, s))) return rc;
        }
        if (rounds == 0) for (size_t i = 0; i < nslices; i++) if (status[i] == hgr::STATUS_RETRY) { again.push_back((uint32_t)i); R.retried[i] = 1; }
        const bool progress = rounds == 0 || again.size() < round.size();
        if (again.empty() || !progress) { for (uint32_t i : again) status[i] = hgr::ERR_UNSUPPORTED; break; }     // alone in the pool and still no room: the batch's seq_cap is too small for it
        std::vector<int32_t> pre(nslices, hgr::STATUS_SKIP);
        for (uint32_t i : again) pre[i] = 0;
        if (hipMemcpyAsync(R.d_pre1, pre.data(), nslices * 4, hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(d_out + R.opool, 0, 8, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        if ((rc = chain_launch(again.size(), R.d_pre1))) return rc;
        if (!verdicts()) return HG_ELAUNCH;
        round.swap(again);
        PT.mark("chain (again)");
    }
    if (hipMemcpyAsync(R.d_status, status.data(), nslices * 4, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;      // final verdicts (pool retries resolved) for the kernels below
    R.pool_used = 0;
    if (want_seq && FD.seq_used) {
        unsigned long long u = 0;
        if (hipMemcpyAsync(&u, FD.seq_used, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        R.pool_used = u;
    }
    PT.mark("placement");
    // ---- pack: prefix sums of the per-slice totals on the host (a few words per slice), second kernel
    std::vector<uint64_t> &base = R.base;
    base.assign(nslices * 3 + 1, 0);
    uint64_t used_c = 0, used_n = 0, used_a = 0;
    for (size_t i = 0; i < nslices; i++) {
        base[3 * i] = used_c; base[3 * i + 1] = used_n; base[3 * i + 2] = used_a;
        if (status[i] == 0) { used_c += tot[4 * i]; used_n += tot[4 * i + 1]; used_a += want_aux ? tot[4 * i + 2] : 0u; }
    }
    R.used_c = used_c; R.used_n = used_n; R.used_a = used_a;
    if (used_c > cigar_cap || used_n > name_cap || used_a > aux_cap) return HG_ENOMEM;      // the caller's arrays are too small: `used` says what is needed
    if (want_seq && R.pool_used > R.seq_cap) return HG_ENOMEM;
    const size_t pc = (used_c * 4 + 63) & ~(size_t)63, pn = (used_n + 63) & ~(size_t)63, pa = (used_a + 63) & ~(size_t)63;
    if ((rc = M.need(M_PACK, pc + pn + pa + 64))) return rc;
    uint8_t *d_pack = (uint8_t *)M.p[M_PACK];
    R.PK = hgr::Dense{(uint32_t *)d_pack, d_pack + pc, d_pack + pc + pn, (const uint64_t *)(d_out + R.obase)};
    if (hipMemcpyAsync(d_out + R.obase, base.data(), nslices * 24, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    hipLaunchKernelGGL(hgr::cram_records_pack_kernel, dim3((unsigned)std::min<size_t>(nslices, (size_t)ctx->cus * 8)), dim3(1024), 0, s, T, D, R.PK, (uint32_t)nslices, R.d_status);
    if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    PT.mark("pack");
    R.bam_bytes = 0;
    if (bam) {                                                           // cram_to_bam on the device: sizes, prefix sum, bytes
        std::vector<uint32_t> rgo((size_t)bam->nrg + 1, 0u); std::vector<unsigned char> rgn;
        for (int i = 0; i < bam->nrg; i++) { const size_t l = strlen(bam->rg_names[i]); rgn.insert(rgn.end(), bam->rg_names[i], bam->rg_names[i] + l); rgo[(size_t)i + 1] = (uint32_t)rgn.size(); }
        const uint32_t plen = bam->prefix ? (uint32_t)std::min<size_t>(strlen(bam->prefix), 200) : 0u;      // the read-group names are followed by the name prefix
        const size_t rg_bytes = rgn.size();
        if (plen) rgn.insert(rgn.end(), bam->prefix, bam->prefix + plen);
        const size_t NR = B.nrec ? B.nrec : 1, ntiles = (NR + hgr::SCAN_TILE - 1) / hgr::SCAN_TILE;
        const size_t szb = ((NR + 1) * 8 + 63) & ~(size_t)63, tlb = ((ntiles + 1) * 8 + 63) & ~(size_t)63, rgb = (rgo.size() * 4 + 63) & ~(size_t)63;
        const size_t dsb = NR * hgr::BD_WORDS * 4;
        if ((rc = M.need(M_BSZ, szb + tlb + rgb + ((rgn.size() + 63) & ~(size_t)63) + dsb + 256))) return rc;
        uint8_t *d_b = (uint8_t *)M.p[M_BSZ];
        uint32_t *d_desc = (uint32_t *)(d_b + szb + tlb + rgb + ((rgn.size() + 63) & ~(size_t)63) + 64);
        uint64_t *d_sz = (uint64_t *)d_b, *d_tile = (uint64_t *)(d_b + szb); uint32_t *d_rgo = (uint32_t *)(d_b + szb + tlb); unsigned char *d_rgn = (unsigned char *)d_rgo + rgb;
        ok = hipMemcpyAsync(d_rgo, rgo.data(), rgo.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
             (rgn.empty() || hipMemcpyAsync(d_rgn, rgn.data(), rgn.size(), hipMemcpyHostToDevice, s) == hipSuccess);
        if (!ok) return HG_ELAUNCH;
        const unsigned grid = (unsigned)std::min<size_t>(nslices, (size_t)ctx->cus * 8);
        hipLaunchKernelGGL(hgr::cram_bam_size_kernel, dim3(grid), dim3(1024), 0, s, T, D, R.PK, d_rgo, (int32_t)bam->nrg, plen, (uint32_t)nslices, R.d_status, d_sz, d_desc);
        hipLaunchKernelGGL(hgr::scan_tile_sums_kernel, dim3((unsigned)ntiles), dim3(hgr::SCAN_TPB), 0, s, d_sz, (uint64_t)B.nrec, d_tile);
        hipLaunchKernelGGL(hgr::scan_tile_bases_kernel, dim3(1), dim3(hgr::SCAN_TPB), 0, s, d_tile, (uint64_t)ntiles, d_sz + B.nrec);
        hipLaunchKernelGGL(hgr::scan_tiles_kernel, dim3((unsigned)ntiles), dim3(hgr::SCAN_TPB), 0, s, d_sz, (uint64_t)B.nrec, d_tile);
        uint64_t total = 0;
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&total, d_sz + B.nrec, 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(status.data(), R.d_status, nslices * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        if (!ok) return HG_ELAUNCH;
        PT.mark("bam sizes + scan");
        R.bam_bytes = total; R.d_bam_off = d_sz;
        if (bam->total) *bam->total = total;
        if (total > bam->cap) return HG_ENOMEM;
        if ((rc = M.need(M_BAM, total + 64))) return rc;
        R.d_bam = (uint8_t *)M.p[M_BAM];
        const unsigned wgrid = (unsigned)std::min<size_t>((B.nrec + 3) / 4 + 1, (size_t)ctx->cus * 64);
        hipLaunchKernelGGL(hgr::cram_bam_write_kernel, dim3(wgrid), dim3(256), 0, s, D, R.PK, d_rgn, d_rgo, d_rgn + rg_bytes, plen, (uint64_t)B.nrec, d_sz, d_desc, R.d_bam);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        PT.mark("bam write");
    }
    return HG_OK;
}
// the decoded columns of the last run -> the caller's arrays
static int rec_fetch(hg_ctx *ctx, hg_cram_batch &R, const hg_cram_record_cols *out) {
    hipStream_t s = ctx->stream;
    const hgr::Batch &B = R.B;
    const uint8_t *d_out = (const uint8_t *)R.M.p[M_OUT];
    bool ok = true;
    void *dst32[9] = {out->flags, out->cram_flags, out->ref_id, out->len, out->rg, out->mqual, out->mate_ref_id, out->ncigar, out->name_len};
    for (int i = 0; i < 9 && ok; i++) if (dst32[i] && B.nrec) ok = hipMemcpyAsync(dst32[i], d_out + R.o32[i], B.nrec * 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    void *dst64[4] = {out->apos, out->aend, out->mate_pos, out->tlen};
    for (int i = 0; i < 4 && ok; i++) if (dst64[i] && B.nrec) ok = hipMemcpyAsync(dst64[i], d_out + R.o64[i], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->cigar_off && B.nrec) ok = hipMemcpyAsync(out->cigar_off, d_out + R.ou64[0], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->name_off && B.nrec) ok = hipMemcpyAsync(out->name_off, d_out + R.ou64[1], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->cigar && R.used_c) ok = hipMemcpyAsync(out->cigar, R.PK.cigar, R.used_c * 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->names && R.used_n) ok = hipMemcpyAsync(out->names, R.PK.names, R.used_n, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && R.want_aux && out->aux && B.nrec) ok = hipMemcpyAsync(out->aux_off, d_out + R.ou64[2], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                                                 hipMemcpyAsync(out->aux_len, d_out + R.o32[11], B.nrec * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
                                                 (!R.used_a || hipMemcpyAsync(out->aux, R.PK.aux, R.used_a, hipMemcpyDeviceToHost, s) == hipSuccess);
    if (ok && R.want_seq && out->seq && B.nrec) {
        const uint64_t n = R.pool_used < R.seq_cap ? R.pool_used : R.seq_cap;
        ok = hipMemcpyAsync(out->seq_off, d_out + R.oso, B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
             (!n || (hipMemcpyAsync(out->seq, d_out + R.oseq, n, hipMemcpyDeviceToHost, s) == hipSuccess && hipMemcpyAsync(out->qual, d_out + R.oqual, n, hipMemcpyDeviceToHost, s) == hipSuccess));
    }
    return ok && hipStreamSynchronize(s) == hipSuccess ? HG_OK : HG_ELAUNCH;
}

extern "C" int hg_cram_decode_records_host(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, size_t rec_cap,
                                           size_t cigar_cap, size_t name_cap, size_t seq_cap, size_t aux_cap, const hg_cram_record_cols *out, uint64_t *rec_off, int32_t *status,
                                           uint64_t *used) {
    if (!ctx || (nslices && (!slices || !out || !rec_off || !status))) return HG_EINVAL;
    if (nslices == 0) { if (rec_off) rec_off[0] = 0; return HG_OK; }
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hg_cram_batch R; R.M.ctx = ctx;
    const bool want_aux = out->aux && out->aux_off && out->aux_len, want_seq = out->seq && out->qual && out->seq_off;
    int rc = rec_stage(ctx, R, nslices, slices, major_version, nref, want_seq, want_aux, seq_cap);
    if (rc) return rc;
    if (R.B.nrec > rec_cap) return HG_EINVAL;
    for (size_t i = 0; i < nslices; i++) rec_off[i] = R.B.slices[i].rec_off;
    rec_off[nslices] = R.B.nrec;
    rc = rec_run(ctx, R, nullptr, cigar_cap, name_cap, aux_cap);
    if (used) { used[0] = R.used_c; used[1] = R.used_n; used[2] = R.used_a; used[3] = R.pool_used; }
    for (size_t i = 0; i < nslices && i < R.status.size(); i++) status[i] = R.status[i];
    if (rc) return rc;
    if ((rc = rec_fetch(ctx, R, out))) return rc;
    for (size_t i = 0; i < nslices; i++) if (status[i] != 0) return HG_EBLOCK;
    return HG_OK;
}

// CRAM slices -> uncompressed BAM records (cram_decode_slice + cram_to_bam, cram_decode.c:2346-3192), everything on the device; only the
// BAM bytes come back.  rg_names: the @RG IDs in header order (the RG series indexes them).  Records of failed slices are left out.
extern "C" int hg_cram_decode_bam_host2(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, const char *const *rg_names,
                                        int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap, uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes,
                                        int32_t *status, const char *name_prefix);
extern "C" int hg_cram_decode_bam_host(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, const char *const *rg_names,
                                       int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap, uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes,
                                       int32_t *status) {
    return hg_cram_decode_bam_host2(ctx, nslices, slices, major_version, nref, rg_names, nrg, total_bases, bam_out, bam_cap, rec_off, rec_bam_off, bam_bytes, status, nullptr);
}
// ... with block pointers that may be device addresses (internal: the fused run decoder of cram_file_host.hip)
int hg_cram_decode_bam_devsrc(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, const char *const *rg_names,
                              int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap, uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes,
                              int32_t *status, const char *name_prefix, const uint8_t *dev_lo, const uint8_t *dev_hi, void *wait_ev);
extern "C" int hg_cram_decode_bam_host2(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, const char *const *rg_names,
                                        int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap, uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes,
                                        int32_t *status, const char *name_prefix) {
    return hg_cram_decode_bam_devsrc(ctx, nslices, slices, major_version, nref, rg_names, nrg, total_bases, bam_out, bam_cap, rec_off, rec_bam_off, bam_bytes, status, name_prefix,
                                     nullptr, nullptr, nullptr);
}
int hg_cram_decode_bam_devsrc(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, const char *const *rg_names,
                              int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap, uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes,
                              int32_t *status, const char *name_prefix, const uint8_t *dev_lo, const uint8_t *dev_hi, void *wait_ev) {
    if (!ctx || (nslices && (!slices || !bam_out || !rec_off || !status)) || (nrg && !rg_names)) return HG_EINVAL;
    if (nslices == 0) { if (rec_off) rec_off[0] = 0; if (bam_bytes) *bam_bytes = 0; return HG_OK; }
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hg_cram_batch R; R.M.ctx = ctx; R.dev_lo = dev_lo; R.dev_hi = dev_hi; R.wait_ev = wait_ev;
    static const bool timing = getenv("HG_CRAM_RECORDS_TIMING") != nullptr;
    const auto tt0 = std::chrono::steady_clock::now();
    int rc = rec_stage(ctx, R, nslices, slices, major_version, nref, true, true, (size_t)total_bases);
    if (rc) return rc;
    if (timing) (void)hipStreamSynchronize(ctx->stream);
    const auto tt1 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < nslices; i++) rec_off[i] = R.B.slices[i].rec_off;
    rec_off[nslices] = R.B.nrec;
    const BamSink sink{rg_names, nrg, bam_out, bam_cap, rec_bam_off, bam_bytes, name_prefix};
    rc = rec_run(ctx, R, &sink, (size_t)-1, (size_t)-1, (size_t)-1);
    for (size_t i = 0; i < nslices && i < R.status.size(); i++) status[i] = R.status[i];
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    const auto tt2 = std::chrono::steady_clock::now();
    const bool ok = (!R.bam_bytes || hipMemcpyAsync(bam_out, R.d_bam, R.bam_bytes, hipMemcpyDeviceToHost, s) == hipSuccess) &&
                    (!rec_bam_off || hipMemcpyAsync(rec_bam_off, R.d_bam_off, (R.B.nrec + 1) * 8, hipMemcpyDeviceToHost, s) == hipSuccess) && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "cram records call: stage %.1f ms, run %.1f ms, BAM to the host %.1f ms (%.1f MB)\n", ms(tt0, tt1), ms(tt1, tt2), ms(tt2, std::chrono::steady_clock::now()), R.bam_bytes / 1e6);
    }
    for (size_t i = 0; i < nslices; i++) if (status[i] != 0) return HG_EBLOCK;
    return HG_OK;
}

// ---- device-resident form: stage once, decode as often as wanted, the BAM stream stays in HBM (the next consumer is hg_bgzf_deflate_dev) ----
extern "C" int hg_cram_batch_stage(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, uint64_t total_bases, hg_cram_batch **out) {
    if (!ctx || !out || !nslices || !slices) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hg_cram_batch *R = new (std::nothrow) hg_cram_batch;
    if (!R) return HG_ENOMEM;
    R->M.ctx = ctx; R->M.own = true;
    const int rc = rec_stage(ctx, *R, nslices, slices, major_version, nref, true, true, (size_t)total_bases);
    if (rc) { R->M.release(); delete R; return rc; }
    *out = R;
    return HG_OK;
}
extern "C" int hg_cram_batch_decode_bam_dev(hg_ctx *ctx, hg_cram_batch *batch, const char *const *rg_names, int nrg, const char *name_prefix, void **d_bam, uint64_t *bam_bytes,
                                            uint64_t *nrec, uint64_t *fast_slices, int32_t *status) {
    if (!ctx || !batch || (nrg && !rg_names)) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    uint64_t total = 0;
    const BamSink sink{rg_names, nrg, nullptr, (size_t)-1, nullptr, &total, name_prefix};
    const int rc = rec_run(ctx, *batch, &sink, (size_t)-1, (size_t)-1, (size_t)-1);
    if (status) for (size_t i = 0; i < batch->nslices && i < batch->status.size(); i++) status[i] = batch->status[i];
    if (rc) return rc;
    if (d_bam) *d_bam = batch->d_bam;
    if (bam_bytes) *bam_bytes = total;
    if (nrec) *nrec = batch->B.nrec;
    if (fast_slices) { uint64_t n = 0; for (size_t i = 0; i < batch->nslices; i++) n += batch->F.is_fast[i] && !batch->retried[i]; *fast_slices = n; }
    for (size_t i = 0; i < batch->nslices; i++) if (batch->status[i] != 0) return HG_EBLOCK;
    return HG_OK;
}
extern "C" int hg_cram_batch_read_bam(hg_ctx *ctx, hg_cram_batch *batch, uint8_t *dst, size_t cap) {
    if (!ctx || !batch || (!dst && batch->bam_bytes)) return HG_EINVAL;
    if (batch->bam_bytes > cap) return HG_ENOMEM;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    if (batch->bam_bytes && (hipMemcpyAsync(dst, batch->d_bam, batch->bam_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) return HG_ELAUNCH;
    return HG_OK;
}
extern "C" void hg_cram_batch_free(hg_ctx *ctx, hg_cram_batch *batch) {
    if (!batch) return;
    if (ctx) { hg::CtxGuard guard_(ctx); (void)hipStreamSynchronize(ctx->stream); batch->M.release(); }
    delete batch;
}

// The .crai lines of one slice (cram_index_slice / cram_index_build_multiref, reference cram/cram_index.c:632-728): a single-reference
// slice is indexed from its header; a multi-reference slice (ref_seq_id == -2) gets one line per run of records on the same reference,
// with the run's first position and the span up to its furthest alignment end -- the ref_id / apos / aend columns of
// hg_cram_decode_records_host.  Host code: a few comparisons per record on columns that are already in host memory.
extern "C" long hg_cram_crai_slice(const uint8_t *slice_hdr, uint32_t slice_hdr_len, int major_version, const int32_t *ref_id, const int64_t *apos,
                                   const int64_t *aend, int64_t container_pos, int32_t landmark, int32_t slice_bytes, char *out, size_t cap) {
    if (!slice_hdr || !out) return HG_EINVAL;
    hgr::SliceHeader sh;
    if (hgr::parse_slice_header(slice_hdr, slice_hdr_len, major_version, sh)) return HG_EINVAL;
    size_t n = 0;
    auto line = [&](int32_t ref, int64_t start, int64_t span) {
        const int k = snprintf(out + n, cap - n, "%d\t%lld\t%lld\t%lld\t%d\t%d\n", ref, (long long)start, (long long)span, (long long)container_pos, landmark, slice_bytes);
        if (k < 0 || (size_t)k >= cap - n) return false;
        n += (size_t)k;
        return true;
    };
    if (sh.ref_seq_id != -2) return line(sh.ref_seq_id, sh.ref_seq_start, sh.ref_seq_span) ? (long)n : (long)HG_ENOMEM;
    if (sh.nrec && (!ref_id || !apos || !aend)) return HG_EINVAL;
    int32_t ref = -2, last_ref = -9; int64_t ref_start = 0, ref_end = INT32_MIN, last_pos = -9;
    for (int32_t i = 0; i < sh.nrec; i++) {
        if (ref_id[i] == last_ref && apos[i] < last_pos) return -2;      // "CRAM file is not sorted by chromosome / position"
        last_ref = ref_id[i]; last_pos = apos[i];
        if (ref_id[i] == ref) { if (ref_end < aend[i]) ref_end = aend[i]; continue; }
        if (ref != -2 && !line(ref, ref_start, ref_end - ref_start + 1)) return HG_ENOMEM;
        ref = ref_id[i]; ref_start = apos[i]; ref_end = aend[i];
    }
    if (ref != -2 && !line(ref, ref_start, ref_end - ref_start + 1)) return HG_ENOMEM;
    return (long)n;
}

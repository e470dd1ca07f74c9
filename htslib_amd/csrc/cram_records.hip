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
__global__ __launch_bounds__(64)
void cram_records_pack_kernel(DevTables T, DevCols D, Dense P, uint32_t nslices, const int32_t *status) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k = blockIdx.x; k < nslices; k += gridDim.x) {
        if (status[k] != 0) continue;
        const SliceDev d = T.slices[k];
        const uint64_t bc = P.base[3 * (size_t)k], bn = P.base[3 * (size_t)k + 1], ba = P.base[3 * (size_t)k + 2];
        const uint32_t nc = D.totals[4 * (size_t)k], nn = D.totals[4 * (size_t)k + 1], na = D.totals[4 * (size_t)k + 2];
        for (uint32_t i = lane; i < nc; i += 64) P.cigar[bc + i] = D.cigar[d.cig_off + i];
        for (uint32_t i = lane; i < nn; i += 64) P.names[bn + i] = D.names[d.name_off + i];
        if (D.aux) for (uint32_t i = lane; i < na; i += 64) P.aux[ba + i] = D.aux[d.aux_off + i];
        for (uint32_t r = lane; r < (uint32_t)d.nrec; r += 64) {
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
        if (pre_status[k] != 0) { if (lane == 0) status[k] = pre_status[k]; continue; }
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
        if (pre_status[k] != 0) { status[k] = pre_status[k]; continue; }
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
__device__ __forceinline__ uint32_t bam_record_bytes(const DevCols &D, uint64_t r, const uint32_t *rg_off, int32_t nrg) {
    const uint32_t nl = D.name_len[r] > 0 ? (uint32_t)D.name_len[r] : 1u, len = (uint32_t)D.len[r];
    const int32_t rg = D.rg[r];
    const uint32_t rgb = rg >= 0 && rg < nrg ? rg_off[rg + 1] - rg_off[rg] + 4u : 0u;
    return 4u + 32u + nl + 1u + 4u * (uint32_t)D.ncigar[r] + (len + 1u) / 2u + len + (uint32_t)D.aux_len[r] + rgb;
}
__global__ __launch_bounds__(64)
void cram_bam_size_kernel(DevTables T, DevCols D, const uint32_t *rg_off, int32_t nrg, uint32_t nslices, int32_t *status, uint64_t *sizes) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k = blockIdx.x; k < nslices; k += gridDim.x) {
        const SliceDev d = T.slices[k];
        bool ok = status[k] == 0;
        if (ok) {                                                          // what a BAM record cannot hold (bam_set1 / bam_write1 refuse or re-route these)
            bool bad = false;
            for (uint32_t r = lane; r < (uint32_t)d.nrec; r += 64) bad |= D.name_len[d.rec_off + r] > 254 || D.ncigar[d.rec_off + r] > 65535 || D.len[d.rec_off + r] < 0;
            if (__ballot(bad)) { ok = false; if (lane == 0) status[k] = ERR_UNSUPPORTED; }
        }
        for (uint32_t r = lane; r < (uint32_t)d.nrec; r += 64) sizes[d.rec_off + r] = ok ? bam_record_bytes(D, d.rec_off + r, rg_off, nrg) : 0u;
        hg::wave_sync();
    }
}
// exclusive prefix sum of n values in place, n + 1 outputs (one workgroup: each thread owns a contiguous piece)
__global__ __launch_bounds__(1024)
void scan_u64_kernel(uint64_t *v, uint64_t n) {
    __shared__ uint64_t part[1024];
    const uint64_t t = threadIdx.x, per = (n + 1023) / 1024, a = t * per < n ? t * per : n, b = a + per < n ? a + per : n;
    uint64_t sum = 0;
    for (uint64_t i = a; i < b; i++) sum += v[i];
    part[t] = sum;
    __syncthreads();
    if (t == 0) { uint64_t run = 0; for (int i = 0; i < 1024; i++) { const uint64_t x = part[i]; part[i] = run; run += x; } v[n] = run; }
    __syncthreads();
    uint64_t run = part[t];
    for (uint64_t i = a; i < b; i++) { const uint64_t x = v[i]; v[i] = run; run += x; }
}
__device__ __forceinline__ uint32_t nt16(uint32_t c) {                    // seq_nt16_table (hts.c)
    switch (c & ~0x20u) {
    case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5; case 'S': return 6; case 'V': return 7; case 'T': return 8;
    case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12; case 'D': return 13; case 'B': return 14;
    default: return c == '=' ? 0u : 15u;
    }
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
__global__ __launch_bounds__(64)
void cram_bam_write_kernel(DevTables T, DevCols D, Dense P, const unsigned char *rg_names, const uint32_t *rg_off, int32_t nrg, uint32_t nslices,
                           const int32_t *status, const uint64_t *off, uint8_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k = blockIdx.x; k < nslices; k += gridDim.x) {
        if (status[k] != 0) continue;
        const SliceDev d = T.slices[k];
        for (uint32_t q = lane; q < (uint32_t)d.nrec; q += 64) {
            const uint64_t r = d.rec_off + q;
            uint8_t *o = out + off[r];
            const uint32_t bytes = (uint32_t)(off[r + 1] - off[r]);
            const uint32_t nl = D.name_len[r] > 0 ? (uint32_t)D.name_len[r] : 0u, len = (uint32_t)D.len[r], nc = (uint32_t)D.ncigar[r], flag = (uint32_t)D.flags[r];
            const uint32_t *cig = P.cigar + D.cigar_off[r];
            int64_t rlen = 0;
            if (!(flag & BAM_FUNMAP)) for (uint32_t i = 0; i < nc; i++) { const uint32_t op = cig[i] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cig[i] >> 4; }
            if (rlen == 0) rlen = 1;
            const int64_t pos = D.apos[r] - 1, mpos = D.mate_pos[r] - 1;
            auto put32 = [&](uint32_t at, uint32_t v) { o[at] = (uint8_t)v; o[at + 1] = (uint8_t)(v >> 8); o[at + 2] = (uint8_t)(v >> 16); o[at + 3] = (uint8_t)(v >> 24); };
            put32(0, bytes - 4u);
            put32(4, (uint32_t)D.ref_id[r]); put32(8, (uint32_t)pos);
            put32(12, reg2bin(pos, pos + rlen) << 16 | ((uint32_t)D.mqual[r] & 0xffu) << 8 | ((nl ? nl : 1u) + 1u));
            put32(16, flag << 16 | (nc & 0xffffu)); put32(20, len);
            put32(24, (uint32_t)D.mate_ref_id[r]); put32(28, (uint32_t)mpos); put32(32, (uint32_t)D.tlen[r]);
            uint32_t at = 36;
            if (nl) { const uint8_t *nm = P.names + D.name_off[r]; for (uint32_t i = 0; i < nl; i++) o[at + i] = nm[i]; at += nl; } else o[at++] = '*';
            o[at++] = 0;
            for (uint32_t i = 0; i < nc; i++) { put32(at, cig[i]); at += 4; }
            const uint8_t *sq = D.seq + D.seq_off[r], *ql = D.qual + D.seq_off[r];
            for (uint32_t i = 0; i + 1 < len; i += 2) o[at + (i >> 1)] = (uint8_t)(nt16(sq[i]) << 4 | nt16(sq[i + 1]));
            if (len & 1u) o[at + (len >> 1)] = (uint8_t)(nt16(sq[len - 1]) << 4);
            at += (len + 1u) / 2u;
            for (uint32_t i = 0; i < len; i++) o[at + i] = ql[i];
            at += len;
            const uint32_t na = (uint32_t)D.aux_len[r];
            const uint8_t *ax = P.aux + D.aux_off[r];
            for (uint32_t i = 0; i < na; i++) o[at + i] = ax[i];
            at += na;
            const int32_t rg = D.rg[r];
            if (rg >= 0 && rg < nrg) {                                       // RG:Z: from the read-group series (cram_decode.c:3180-3189)
                o[at++] = 'R'; o[at++] = 'G'; o[at++] = 'Z';
                for (uint32_t i = rg_off[rg]; i < rg_off[rg + 1]; i++) o[at++] = rg_names[i];
                o[at++] = 0;
            }
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
        char b[96]; snprintf(b, sizeof b, " %s %.1f ms", what, std::chrono::duration<double, std::milli>(t - t0).count()); log += b; t0 = t;
    }
    ~PhaseTimer() { if (on) fprintf(stderr, "cram records phases:%s\n", log.c_str()); }
};
struct BamSink { const char *const *rg_names; int nrg; uint8_t *out; size_t cap; uint64_t *rec_bam_off; uint64_t *total; };
}
static int records_impl(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, size_t rec_cap, size_t cigar_cap,
                        size_t name_cap, size_t seq_cap, size_t aux_cap, const hg_cram_record_cols *out, uint64_t *rec_off, int32_t *status, uint64_t *used,
                        const BamSink *bam) {
    if (!ctx || (nslices && (!slices || !out || !rec_off || !status))) return HG_EINVAL;
    if (nslices == 0) { if (rec_off) rec_off[0] = 0; return HG_OK; }
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hgr::Batch B;
    int rc = hgr::batch_build(B, (const hgr::SliceIn *)slices, nslices, major_version);
    if (rc) return rc == -3 ? HG_BLOCK_EUNSUPPORTED : HG_EINVAL;
    const bool want_aux = bam || (out->aux && out->aux_off && out->aux_len);
    if (B.nrec > rec_cap) return HG_EINVAL;
    for (size_t i = 0; i < nslices; i++) rec_off[i] = B.slices[i].rec_off;
    rec_off[nslices] = B.nrec;
    // device image of the tables: one buffer, carved
    struct Part { const void *src; size_t bytes; size_t off; };
    std::vector<Part> parts = {{B.plans.data(), B.plans.size() * sizeof(hgr::PlanDev), 0}, {B.codecs.data(), B.codecs.size() * sizeof(hgr::Codec), 0},
                               {B.huff.data(), B.huff.size() * sizeof(hgr::HuffCode), 0}, {B.tl_off.data(), B.tl_off.size() * 4, 0},
                               {B.tl_codec.data(), B.tl_codec.size() * 4, 0}, {B.slices.data(), B.slices.size() * sizeof(hgr::SliceDev), 0},
                               {B.tab.data(), B.tab.size() * 4, 0}, {B.status.data(), B.status.size() * 4, 0},
                               {B.refs.data(), B.refs.size() * sizeof(hgr::RefSpan), 0}, {B.tl_tag.data(), B.tl_tag.size() * 4, 0}};
    size_t tbytes = 0;
    for (auto &p : parts) { p.off = tbytes; tbytes += (p.bytes + 63) & ~(size_t)63; }
    const size_t R = B.nrec ? B.nrec : 1;
    // output image: 9 + 2 int32, 4 + 1 int64, 2 uint64, 2 uint32 scratch columns, cigar, names, status
    size_t obytes = 0;
    auto carve = [&](size_t bytes) { const size_t o = obytes; obytes += (bytes + 63) & ~(size_t)63; return o; };
    size_t o32[12], o64[5], ou64[3], ou32[3];
    for (auto &o : o32) o = carve(R * 4);
    for (auto &o : o64) o = carve(R * 8);
    for (auto &o : ou64) o = carve(R * 8);
    for (auto &o : ou32) o = carve(R * 4);
    const size_t ocig = carve((B.cig_total ? B.cig_total : 1) * 4), onam = carve(B.name_total ? B.name_total : 1), ost = carve(nslices * 4);
    const bool want_seq = bam || (out->seq && out->qual && out->seq_off);
    const size_t oaux = carve(want_aux ? B.aux_total + 1 : 1);
    const size_t otot = carve(nslices * 16), obase = carve(nslices * 24);
    const size_t oso = carve(R * 8), oseq = carve(want_seq ? seq_cap + 1 : 1), oqual = carve(want_seq ? seq_cap + 1 : 1), opool = carve(8);
    if ((rc = hg::ensure_scratch(ctx, 0, B.data_bytes + 64)) || (rc = hg::ensure_scratch(ctx, 1, obytes + 64)) || (rc = hg::ensure_scratch(ctx, 2, tbytes + 64))) return rc;
    hipStream_t s = ctx->stream;
    PhaseTimer PT(s);
    PT.mark("plan");
    uint8_t *d_data = (uint8_t *)ctx->d_scratch[0], *d_out = (uint8_t *)ctx->d_scratch[1], *d_tab = (uint8_t *)ctx->d_scratch[2];
    bool ok = hg::stage_upload(ctx, B.src_ptr.data(), B.src_len.data(), B.src_off.data(), nullptr, B.src_ptr.size(), B.data_bytes, d_data, s) == HG_OK;
    for (auto &p : parts) if (ok && p.bytes) ok = hipMemcpyAsync(d_tab + p.off, p.src, p.bytes, hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    hgr::DevTables T{(const hgr::PlanDev *)(d_tab + parts[0].off), (const hgr::Codec *)(d_tab + parts[1].off), (const hgr::HuffCode *)(d_tab + parts[2].off),
                     (const int32_t *)(d_tab + parts[3].off), (const int32_t *)(d_tab + parts[4].off), (const int32_t *)(d_tab + parts[9].off),
                     (const hgr::SliceDev *)(d_tab + parts[5].off),
                     (uint32_t *)(d_tab + parts[6].off), d_data, (const hgr::RefSpan *)(d_tab + parts[8].off)};
    hgr::DevCols D;
    int32_t **p32[11] = {&D.flags, &D.cram_flags, &D.ref_id, &D.len, &D.rg, &D.mqual, &D.mate_ref_id, &D.ncigar, &D.name_len, &D.mate_flags, &D.mate_line};
    for (int i = 0; i < 11; i++) *p32[i] = (int32_t *)(d_out + o32[i]);
    int64_t **p64[5] = {&D.apos, &D.aend, &D.mate_pos, &D.tlen, &D.explicit_tlen};
    for (int i = 0; i < 5; i++) *p64[i] = (int64_t *)(d_out + o64[i]);
    D.cigar_off = (uint64_t *)(d_out + ou64[0]); D.name_off = (uint64_t *)(d_out + ou64[1]);
    D.coff = (uint32_t *)(d_out + ou32[0]); D.noff = (uint32_t *)(d_out + ou32[1]); D.aoff = (uint32_t *)(d_out + ou32[2]);
    D.aux_off = (uint64_t *)(d_out + ou64[2]); D.aux_len = (int32_t *)(d_out + o32[11]); D.aux = want_aux ? d_out + oaux : nullptr;
    D.cigar = (uint32_t *)(d_out + ocig); D.names = d_out + onam;
    D.seq_off = (uint64_t *)(d_out + oso); D.seq = want_seq ? d_out + oseq : nullptr; D.qual = want_seq ? d_out + oqual : nullptr;
    D.seq_pool = (unsigned long long *)(d_out + opool); D.seq_cap = seq_cap;
    if (hipMemsetAsync(d_out + opool, 0, 8, s) != hipSuccess) return HG_ELAUNCH;
    D.totals = (uint32_t *)(d_out + otot);
    D.jobs = nullptr;
    int32_t *d_status = (int32_t *)(d_out + ost);
    PT.mark("upload");
    // one wavefront per slice until the chip is full of them several times over, then one slice per lane
    bool lane_mode = nslices >= 1024;                                  // measured at 8192 slices: 19.9 ms per call against 32.6 ms (profiles/r02_cram_records_probe.txt)
    if (const char *m = getenv("HG_CRAM_RECORDS_MODE")) lane_mode = m[0] == 'l';
    if (!lane_mode && (want_seq || want_aux)) {                           // room for the copies the chain hands over
        if ((rc = hg::ensure_scratch(ctx, 7, (B.job_total + 1) * sizeof(hgr::CopyJob)))) return rc;
        D.jobs = (hgr::CopyJob *)ctx->d_scratch[7];
    }
    if (lane_mode) {
        const unsigned grid = (unsigned)std::min<size_t>((nslices + 63) / 64, (size_t)ctx->cus * 16);
        hipLaunchKernelGGL(hgr::cram_records_lane_kernel, dim3(grid), dim3(64), 0, s, T, D, (uint32_t)nslices, (int32_t)nref, (const int32_t *)(d_tab + parts[7].off), d_status);
    } else {
        const unsigned grid = (unsigned)std::min<size_t>(nslices, (size_t)ctx->cus * 16);
        hipLaunchKernelGGL(hgr::cram_records_kernel, dim3(grid), dim3(64), 0, s, T, D, (uint32_t)nslices, (int32_t)nref, (const int32_t *)(d_tab + parts[7].off), d_status);
    }
    if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    PT.mark("record loop");
    // pack: totals back, prefix sums on the host, second kernel
    std::vector<uint32_t> tot(nslices * 4);
    ok = hipMemcpyAsync(status, d_status, nslices * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(tot.data(), d_out + otot, nslices * 16, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    std::vector<uint64_t> base(nslices * 3);
    uint64_t used_c = 0, used_n = 0, used_a = 0;
    for (size_t i = 0; i < nslices; i++) {
        base[3 * i] = used_c; base[3 * i + 1] = used_n; base[3 * i + 2] = used_a;
        if (status[i] == 0) { used_c += tot[4 * i]; used_n += tot[4 * i + 1]; used_a += want_aux ? tot[4 * i + 2] : 0u; }
    }
    if (used) { used[0] = used_c; used[1] = used_n; used[2] = used_a; used[3] = 0; }
    if (used_c > cigar_cap || used_n > name_cap || used_a > aux_cap) return HG_ENOMEM;      // the caller's arrays are too small: `used` says what is needed
    const size_t pc = (used_c * 4 + 63) & ~(size_t)63, pn = (used_n + 63) & ~(size_t)63, pa = (used_a + 63) & ~(size_t)63;
    if ((rc = hg::ensure_scratch(ctx, 3, pc + pn + pa + 64))) return rc;
    uint8_t *d_pack = (uint8_t *)ctx->d_scratch[3];
    hgr::Dense PK{(uint32_t *)d_pack, d_pack + pc, d_pack + pc + pn, (const uint64_t *)(d_out + obase)};
    if (hipMemcpyAsync(d_out + obase, base.data(), nslices * 24, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    hipLaunchKernelGGL(hgr::cram_records_pack_kernel, dim3((unsigned)std::min<size_t>(nslices, (size_t)ctx->cus * 32)), dim3(64), 0, s, T, D, PK, (uint32_t)nslices, d_status);
    if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    PT.mark("pack");
    // results back: every column in one copy
    void *dst32[9] = {out->flags, out->cram_flags, out->ref_id, out->len, out->rg, out->mqual, out->mate_ref_id, out->ncigar, out->name_len};
    for (int i = 0; i < 9 && ok; i++) if (dst32[i] && B.nrec) ok = hipMemcpyAsync(dst32[i], d_out + o32[i], B.nrec * 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    void *dst64[4] = {out->apos, out->aend, out->mate_pos, out->tlen};
    for (int i = 0; i < 4 && ok; i++) if (dst64[i] && B.nrec) ok = hipMemcpyAsync(dst64[i], d_out + o64[i], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->cigar_off && B.nrec) ok = hipMemcpyAsync(out->cigar_off, d_out + ou64[0], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->name_off && B.nrec) ok = hipMemcpyAsync(out->name_off, d_out + ou64[1], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->cigar && used_c) ok = hipMemcpyAsync(out->cigar, PK.cigar, used_c * 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && out->names && used_n) ok = hipMemcpyAsync(out->names, PK.names, used_n, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (ok && want_aux && out->aux && B.nrec) ok = hipMemcpyAsync(out->aux_off, d_out + ou64[2], B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                                       hipMemcpyAsync(out->aux_len, d_out + o32[11], B.nrec * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
                                       (!used_a || hipMemcpyAsync(out->aux, PK.aux, used_a, hipMemcpyDeviceToHost, s) == hipSuccess);
    unsigned long long pool_used = 0;
    if (ok && want_seq) ok = hipMemcpyAsync(&pool_used, d_out + opool, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (ok && want_seq && B.nrec && used) used[3] = pool_used;
    if (ok && want_seq && pool_used > seq_cap) return HG_ENOMEM;
    if (ok && want_seq && out->seq && B.nrec) {
        if (pool_used > seq_cap) pool_used = seq_cap;
        ok = hipMemcpyAsync(out->seq_off, d_out + oso, B.nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
             (!pool_used || (hipMemcpyAsync(out->seq, d_out + oseq, pool_used, hipMemcpyDeviceToHost, s) == hipSuccess &&
                             hipMemcpyAsync(out->qual, d_out + oqual, pool_used, hipMemcpyDeviceToHost, s) == hipSuccess));
    }
    PT.mark("columns back");
    if (ok && bam) {                                                     // cram_to_bam on the device: sizes, prefix sum, bytes
        std::vector<uint32_t> rgo((size_t)bam->nrg + 1, 0u); std::vector<unsigned char> rgn;
        for (int i = 0; i < bam->nrg; i++) { const size_t l = strlen(bam->rg_names[i]); rgn.insert(rgn.end(), bam->rg_names[i], bam->rg_names[i] + l); rgo[(size_t)i + 1] = (uint32_t)rgn.size(); }
        const size_t szb = (R + 1) * 8, rgb = (rgo.size() * 4 + 63) & ~(size_t)63;
        if ((rc = hg::ensure_scratch(ctx, 4, szb + rgb + rgn.size() + 128))) return rc;
        uint8_t *d_b = (uint8_t *)ctx->d_scratch[4];
        uint64_t *d_sz = (uint64_t *)d_b; uint32_t *d_rgo = (uint32_t *)(d_b + ((szb + 63) & ~(size_t)63)); unsigned char *d_rgn = (unsigned char *)d_rgo + rgb;
        ok = hipMemcpyAsync(d_rgo, rgo.data(), rgo.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
             (rgn.empty() || hipMemcpyAsync(d_rgn, rgn.data(), rgn.size(), hipMemcpyHostToDevice, s) == hipSuccess);
        const unsigned grid = (unsigned)std::min<size_t>(nslices, (size_t)ctx->cus * 32);
        if (ok) {
            hipLaunchKernelGGL(hgr::cram_bam_size_kernel, dim3(grid), dim3(64), 0, s, T, D, d_rgo, (int32_t)bam->nrg, (uint32_t)nslices, d_status, d_sz);
            hipLaunchKernelGGL(hgr::scan_u64_kernel, dim3(1), dim3(1024), 0, s, d_sz, (uint64_t)B.nrec);
        }
        uint64_t total = 0;
        ok = ok && hipMemcpyAsync(&total, d_sz + B.nrec, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        if (!ok) return HG_ELAUNCH;
        PT.mark("bam sizes + scan");
        if (bam->total) *bam->total = total;
        if (total > bam->cap) return HG_ENOMEM;
        if ((rc = hg::ensure_scratch(ctx, 5, total + 64))) return rc;
        uint8_t *d_bam = (uint8_t *)ctx->d_scratch[5];
        hipLaunchKernelGGL(hgr::cram_bam_write_kernel, dim3(grid), dim3(64), 0, s, T, D, PK, d_rgn, d_rgo, (int32_t)bam->nrg, (uint32_t)nslices, d_status, d_sz, d_bam);
        ok = hipGetLastError() == hipSuccess && (!total || hipMemcpyAsync(bam->out, d_bam, total, hipMemcpyDeviceToHost, s) == hipSuccess) &&
             (!bam->rec_bam_off || hipMemcpyAsync(bam->rec_bam_off, d_sz, (B.nrec + 1) * 8, hipMemcpyDeviceToHost, s) == hipSuccess) &&
             hipMemcpyAsync(status, d_status, nslices * 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    PT.mark("bam write + back");
    if (!ok) return HG_ELAUNCH;
    for (size_t i = 0; i < nslices; i++) if (status[i] != 0) return HG_EBLOCK;
    return HG_OK;
}

extern "C" int hg_cram_decode_records_host(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, size_t rec_cap,
                                           size_t cigar_cap, size_t name_cap, size_t seq_cap, size_t aux_cap, const hg_cram_record_cols *out, uint64_t *rec_off, int32_t *status,
                                           uint64_t *used) {
    return records_impl(ctx, nslices, slices, major_version, nref, rec_cap, cigar_cap, name_cap, seq_cap, aux_cap, out, rec_off, status, used, nullptr);
}

// CRAM slices -> uncompressed BAM records (cram_decode_slice + cram_to_bam, cram_decode.c:2346-3192), everything on the device; only the
// BAM bytes come back.  rg_names: the @RG IDs in header order (the RG series indexes them).  Records of failed slices are left out.
extern "C" int hg_cram_decode_bam_host(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, const char *const *rg_names,
                                       int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap, uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes,
                                       int32_t *status) {
    if (!ctx || (nslices && (!slices || !bam_out || !rec_off || !status)) || (nrg && !rg_names)) return HG_EINVAL;
    hg_cram_record_cols none; memset(&none, 0, sizeof none);
    uint64_t nrec = 0, c0 = 0, c1 = 0, c2 = 0;
    int rc = hg_cram_records_bound(nslices, slices, major_version, &nrec, &c0, &c1, &c2);
    if (rc) return rc;
    const BamSink sink{rg_names, nrg, bam_out, bam_cap, rec_bam_off, bam_bytes};
    return records_impl(ctx, nslices, slices, major_version, nref, (size_t)nrec, (size_t)-1, (size_t)-1, (size_t)total_bases, (size_t)-1, &none, rec_off, status, nullptr, &sink);
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

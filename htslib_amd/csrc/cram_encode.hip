// cram_encode.hip -- the CRAM record ENCODER on MI355X (gfx950): BAM records -> the data series blocks and headers of CRAM slices (the
// write side; reference cram_encode_slice + process_one_read, cram/cram_encode.c:572-793, 1096-1209, 3382-3700).  What comes out is what
// cram_compress_slice takes (hg_cram_compress_slice / the block codecs) and what the record decoder (cram_records_fast.hip) reads back.
//
//   survey    one lane per record: which tag keys and tag lists does each slice hold (atomic tables), which references / positions does it span
//   host      a few hundred bytes per slice: sorted key list, tag dictionary in order of first appearance
//   count     one lane per record walks it (CIGAR against the reference, tags, fixed fields): bytes added to every series
//   sums      exclusive prefix sums per slice and series (fast_scan_kernel); the totals are the block sizes
//   write     the same walk again, storing at the record's offsets
// The per-record walks are cram_encode_core.h, one source for host and device (CPU compile: tests/native/cram_records_host.cpp).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <memory>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include "cram_records_dev.h"
#include "cram_encode_plan.h"

namespace {
// HG_CRAM_RECORDS_TIMING=1: phase times of a call on stderr (stream-synchronising, for probes only)
struct EncTimer {
    bool on; hipStream_t s; std::chrono::steady_clock::time_point t0; std::string log;
    EncTimer(hipStream_t st) : on(getenv("HG_CRAM_RECORDS_TIMING") != nullptr), s(st), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto t = std::chrono::steady_clock::now();
        char b[96]; snprintf(b, sizeof b, " %s %.2f ms", what, std::chrono::duration<double, std::milli>(t - t0).count()); log += b; t0 = t;
    }
    ~EncTimer() { if (on) fprintf(stderr, "cram encode phases:%s\n", log.c_str()); }
};
}  // namespace

namespace hgr {

struct SliceStat { int32_t min_ref, max_ref; long long min_pos, max_end; unsigned long long bases; };   // bases: sum of l_seq (the container header's base count)
struct EncDev {
    const uint8_t *bam; const uint64_t *rec_off; const uint8_t *data; const EncRef *refs; int32_t nref; const uint8_t *rg_names; const uint32_t *rg_off; int32_t nrg;
    const SliceDev *slices; const uint32_t *chunk_slice, *chunk_r0; uint32_t nchunks;
    EncSurvey V; int32_t *fail; SliceStat *stat;
    const uint32_t *keys, *key_off; const uint64_t *lhash; const uint32_t *line_off; const int64_t *start; const uint8_t *multi;
    uint32_t *col; uint64_t N; uint32_t ncmax; uint8_t *out; const uint64_t *base;
};
__device__ __forceinline__ void enc_ctx(const EncDev &E, uint32_t k, EncCtx &C) {
    const SliceDev &d = E.slices[k];
    C.bam = E.bam; C.rec_off = E.rec_off; C.data = E.data; C.refs = E.refs; C.nref = E.nref; C.rg_names = E.rg_names; C.rg_off = E.rg_off; C.nrg = E.nrg;
    C.r0 = d.rec_off; C.nrec = (uint32_t)d.nrec;
    C.keys = E.keys ? E.keys + E.key_off[k] : nullptr; C.nkeys = E.keys ? E.key_off[k + 1] - E.key_off[k] : 0u;
    C.line_hash = E.lhash ? E.lhash + E.line_off[k] : nullptr; C.nlines = E.lhash ? E.line_off[k + 1] - E.line_off[k] : 0u;
    C.fail = E.fail + k;
}
constexpr int ENC_CHUNK = 256;
__global__ __launch_bounds__(ENC_CHUNK)
void enc_survey_kernel(EncDev E) {
    __shared__ EncCtx C;
    const uint32_t k = E.chunk_slice[blockIdx.x], r = E.chunk_r0[blockIdx.x] + threadIdx.x;
    if (threadIdx.x == 0) enc_ctx(E, k, C);
    __syncthreads();
    if (r >= C.nrec) return;
    enc_survey_record(C, r, E.V, k);
    BamRec B;
    if (!bam_parse(C.bam, C.rec_off[C.r0 + r], C.rec_off[C.r0 + r + 1], B)) return;
    long long rl = 0;
    if (!(B.flag & BAM_FUNMAP)) for (uint32_t c = 0; c < B.n_cigar; c++) { const uint32_t cw = ld32(B.cigar + 4 * c), op = cw & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cw >> 4; }
    const long long ap = (long long)B.pos + 1, ae = rl ? ap + rl - 1 : ap;
    SliceStat *st = E.stat + k;
    atomicMin(&st->min_ref, B.ref_id); atomicMax(&st->max_ref, B.ref_id); atomicMin(&st->min_pos, ap); atomicMax(&st->max_end, ae);
    atomicAdd(&st->bases, (unsigned long long)B.l_seq);
}
template <bool WRITE>
__global__ __launch_bounds__(ENC_CHUNK)
void enc_walk_kernel(EncDev E) {
    __shared__ EncCtx C;
    __shared__ int skip;
    const uint32_t k = E.chunk_slice[blockIdx.x], r = E.chunk_r0[blockIdx.x] + threadIdx.x;
    if (threadIdx.x == 0) { skip = E.fail[k] != 0; if (!skip) enc_ctx(E, k, C); }
    __syncthreads();
    if (skip || r >= C.nrec) return;
    const uint64_t g = C.r0 + r;
    const int64_t prev = r ? (int64_t)(int32_t)ld32(C.bam + C.rec_off[g - 1] + 8) + 1 : E.start[k];
    Sink<WRITE> S;
    S.col = E.col; S.N = E.N; S.g = g; S.out = E.out; S.base = E.base ? E.base + (size_t)k * E.ncmax : nullptr;
    if (WRITE) {
#pragma unroll
        for (int s = 0; s < W_N; s++) S.p[s] = E.out + S.base[s] + E.col[(uint64_t)s * E.N + g];
    } else {
#pragma unroll
        for (int s = 0; s < W_N; s++) S.n[s] = 0;
    }
    if (!enc_record<WRITE>(C, r, prev, (int)E.multi[k], S)) return;
    if (!WRITE) {
#pragma unroll
        for (int s = 0; s < W_N; s++) E.col[(uint64_t)s * E.N + g] = S.n[s];
    }
}

}  // namespace hgr

// BAM records (bam_write1's layout, back to back, no header) -> CRAM slices of records_per_slice records.  Per slice a blob: u32 comp_len, compression
// header block, u32 len, slice header block, u32 nblocks, then per block i32 content id, u32 len, bytes.
extern "C" int hg_cram_encode_slices_host(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, size_t nrec, uint32_t records_per_slice, const hg_cram_ref_seq *refs, int nrefs,
                                          const char *const *rg_names, int nrg, int64_t record_counter0, uint8_t *out, size_t out_cap, uint64_t *slice_off, size_t max_slices,
                                          int32_t *status, uint64_t *out_bytes) {
    if (!nrec) { if (!ctx || !records_per_slice) return HG_EINVAL; if (out_bytes) *out_bytes = 0; if (slice_off) slice_off[0] = 0; return HG_OK; }
    size_t n = nrec;
    return hg_cram_encode_slices_host2(ctx, bam, bam_len, &n, records_per_slice, refs, nrefs, rg_names, nrg, record_counter0, out, out_cap, slice_off, max_slices, status, out_bytes, nullptr);
}

// *nrec_io = 0 on entry: the records are counted here (bam_read1's framing runs on the device anyway); on return the number of records.  slice_bases (may be
// NULL): per slice the sum of the records' l_seq -- what the container header of cram_write_container wants -- from the survey pass.
extern "C" int hg_cram_encode_slices_host2(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, size_t *nrec_io, uint32_t records_per_slice, const hg_cram_ref_seq *refs, int nrefs,
                                           const char *const *rg_names, int nrg, int64_t record_counter0, uint8_t *out, size_t out_cap, uint64_t *slice_off, size_t max_slices,
                                           int32_t *status, uint64_t *out_bytes, uint64_t *slice_bases) {
    using namespace hgr;
    if (!ctx || !nrec_io || !bam || !out || !slice_off || !status || !records_per_slice || (nrefs && !refs) || (nrg && !rg_names)) return HG_EINVAL;
    const size_t nrec_in = *nrec_io, nrec_bound = nrec_in ? nrec_in : bam_len / 36 + 1;       // (a record is at least 36 bytes: block_size + the fixed fields)
    if (out_bytes) *out_bytes = 0;
    if (bam_len < 36) { if (nrec_in) return HG_EINVAL; *nrec_io = 0; slice_off[0] = 0; return bam_len ? HG_EINVAL : HG_OK; }
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    // ---- lay out references and read-group names (the records are framed on the device: a host walk of the block_size fields is one cache miss
    //      per record, 50 ms for 640 k records -- more than everything else in this call)
    std::vector<EncRef> er((size_t)nrefs + 1); uint64_t dbytes = 0;
    for (int i = 0; i < nrefs; i++) { er[(size_t)i].off = dbytes; er[(size_t)i].len = refs[i].bases ? (int64_t)refs[i].len : 0; if (refs[i].bases) dbytes += (refs[i].len + 15) & ~15ull; }
    std::vector<uint8_t> rgn; std::vector<uint32_t> rgo((size_t)nrg + 1, 0);
    for (int i = 0; i < nrg; i++) { rgn.insert(rgn.end(), rg_names[i], rg_names[i] + strlen(rg_names[i])); rgo[(size_t)i + 1] = (uint32_t)rgn.size(); }
    hipStream_t s = ctx->stream;
    int rc;
    EncTimer PT(s);
    // ---- device image 1: BAM, offsets, references, small tables
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_bam = 0, o_off = al(bam_len + 64), o_data = o_off + al((nrec_bound + 1) * 8), o_end1 = o_data + al(dbytes + 64);
    if ((rc = hg::ensure_scratch(ctx, 0, o_end1))) return rc;
    uint8_t *d1 = (uint8_t *)ctx->d_scratch[0];
    bool ok = hipMemcpyAsync(d1 + o_bam, bam, bam_len, hipMemcpyHostToDevice, s) == hipSuccess;
    size_t nrec = 0;
    std::vector<uint64_t> rec_off;
    if (ok) {                                                            // bam_read1's framing, exact (hg_bam_frame_dev verifies its per-chunk guesses link by link)
        uint64_t bad = 0, end = bam_len;
        const long got = hg_bam_frame_dev(ctx, d1 + o_bam, bam_len, 0, 0x7fffffff, (uint64_t *)(d1 + o_off), nrec_bound, &bad, s);
        if (got < 0) return got == HG_BAM_ETRUNC || got == HG_BAM_EINVALID ? HG_EINVAL : (int)got;
        if ((nrec_in && (size_t)got != nrec_in) || (size_t)got > nrec_bound) return HG_EINVAL;
        nrec = (size_t)got; *nrec_io = nrec;
        if (!nrec) { slice_off[0] = 0; return HG_OK; }
        rec_off.resize(nrec + 1);
        ok = hipMemcpyAsync(d1 + o_off + nrec * 8, &end, 8, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(rec_off.data(), d1 + o_off, nrec * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipStreamSynchronize(s) == hipSuccess;
        rec_off[nrec] = bam_len;
    }
    if (!ok) return HG_ELAUNCH;
    PT.mark("upload + frame");
    const size_t ns = (nrec + records_per_slice - 1) / records_per_slice;
    if (ns > max_slices) return HG_ENOMEM;
    std::vector<SliceDev> sl(ns); std::vector<uint32_t> chunk_slice, chunk_r0, list(ns);
    std::vector<EncSlice> S(ns);
    for (size_t k = 0; k < ns; k++) {
        S[k].r0 = k * records_per_slice; S[k].nrec = (uint32_t)std::min<size_t>(records_per_slice, nrec - S[k].r0);
        memset(&sl[k], 0, sizeof sl[k]); sl[k].rec_off = S[k].r0; sl[k].nrec = (int32_t)S[k].nrec; list[k] = (uint32_t)k;
        for (uint32_t r0 = 0; r0 < S[k].nrec; r0 += ENC_CHUNK) { chunk_slice.push_back((uint32_t)k); chunk_r0.push_back(r0); }
    }
    const size_t N = (nrec + 16) & ~(size_t)15, nch = chunk_slice.size();
    for (int i = 0; i < nrefs && ok; i++) if (refs[i].bases && refs[i].len) ok = hipMemcpyAsync(d1 + o_data + er[(size_t)i].off, refs[i].bases, refs[i].len, hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    size_t tb = 0;
    auto carve = [&](size_t b) { const size_t o = tb; tb += al(b); return o; };
    const size_t t_refs = carve(er.size() * sizeof(EncRef)), t_rgn = carve(rgn.size() + 1), t_rgo = carve(rgo.size() * 4), t_sl = carve(ns * sizeof(SliceDev)), t_cs = carve(nch * 4), t_cr = carve(nch * 4),
                 t_list = carve(ns * 4), t_keys = carve(ns * ENC_KEY_SLOTS * 4), t_lh = carve(ns * ENC_LINE_SLOTS * 8), t_lf = carve(ns * ENC_LINE_SLOTS * 4), t_lc = carve(ns * ENC_LINE_SLOTS * 8), t_fail = carve(ns * 4), t_stat = carve(ns * sizeof(SliceStat)),
                 t_k2 = carve(ns * ENC_MAX_TAGS * 4 + 4), t_ko = carve((ns + 1) * 4), t_l2 = carve(ns * ENC_MAX_LINES * 8 + 8), t_lo = carve((ns + 1) * 4), t_start = carve(ns * 8), t_multi = carve(ns);
    if ((rc = hg::ensure_scratch(ctx, 2, tb + 64))) return rc;
    uint8_t *dt = (uint8_t *)ctx->d_scratch[2];
    std::vector<SliceStat> stat(ns, SliceStat{INT32_MAX, INT32_MIN, INT64_MAX, INT64_MIN, 0});
    ok = hipMemcpyAsync(dt + t_refs, er.data(), er.size() * sizeof(EncRef), hipMemcpyHostToDevice, s) == hipSuccess && (rgn.empty() || hipMemcpyAsync(dt + t_rgn, rgn.data(), rgn.size(), hipMemcpyHostToDevice, s) == hipSuccess) &&
         hipMemcpyAsync(dt + t_rgo, rgo.data(), rgo.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(dt + t_sl, sl.data(), ns * sizeof(SliceDev), hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(dt + t_cs, chunk_slice.data(), nch * 4, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(dt + t_cr, chunk_r0.data(), nch * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(dt + t_list, list.data(), ns * 4, hipMemcpyHostToDevice, s) == hipSuccess && hipMemsetAsync(dt + t_keys, 0xff, ns * ENC_KEY_SLOTS * 4, s) == hipSuccess &&
         hipMemsetAsync(dt + t_lh, 0, ns * ENC_LINE_SLOTS * 8, s) == hipSuccess && hipMemsetAsync(dt + t_lc, 0, ns * ENC_LINE_SLOTS * 8, s) == hipSuccess && hipMemsetAsync(dt + t_lf, 0xff, ns * ENC_LINE_SLOTS * 4, s) == hipSuccess && hipMemsetAsync(dt + t_fail, 0, ns * 4, s) == hipSuccess &&
         hipMemcpyAsync(dt + t_stat, stat.data(), ns * sizeof(SliceStat), hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    EncDev E; memset(&E, 0, sizeof E);
    E.bam = d1 + o_bam; E.rec_off = (const uint64_t *)(d1 + o_off); E.data = d1 + o_data; E.refs = (const EncRef *)(dt + t_refs); E.nref = nrefs; E.rg_names = dt + t_rgn; E.rg_off = (const uint32_t *)(dt + t_rgo); E.nrg = nrg;
    E.slices = (const SliceDev *)(dt + t_sl); E.chunk_slice = (const uint32_t *)(dt + t_cs); E.chunk_r0 = (const uint32_t *)(dt + t_cr); E.nchunks = (uint32_t)nch;
    E.V = EncSurvey{(uint32_t *)(dt + t_keys), (uint64_t *)(dt + t_lh), (uint64_t *)(dt + t_lc), (uint32_t *)(dt + t_lf)}; E.fail = (int32_t *)(dt + t_fail); E.stat = (SliceStat *)(dt + t_stat);
    E.N = N;
    PT.mark("upload");
    // ---- survey
    hipLaunchKernelGGL(enc_survey_kernel, dim3((unsigned)nch), dim3(ENC_CHUNK), 0, s, E);
    std::vector<uint32_t> keytab(ns * ENC_KEY_SLOTS), lfirst(ns * ENC_LINE_SLOTS); std::vector<uint64_t> lhash(ns * ENC_LINE_SLOTS); std::vector<int32_t> fail(ns);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(keytab.data(), dt + t_keys, keytab.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(lhash.data(), dt + t_lh, lhash.size() * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipMemcpyAsync(lfirst.data(), dt + t_lf, lfirst.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(fail.data(), dt + t_fail, ns * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipMemcpyAsync(stat.data(), dt + t_stat, ns * sizeof(SliceStat), hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipStreamSynchronize(s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    PT.mark("survey");
    std::vector<uint32_t> k2, ko(ns + 1, 0), lo(ns + 1, 0); std::vector<uint64_t> l2; std::vector<int64_t> start(ns); std::vector<uint8_t> multi(ns);
    size_t ncmax = W_N;
    for (size_t k = 0; k < ns; k++) {
        if (slice_bases) slice_bases[k] = stat[k].bases;
        S[k].fail = fail[k]; S[k].min_ref = stat[k].min_ref; S[k].max_ref = stat[k].max_ref; S[k].min_pos = stat[k].min_pos; S[k].max_end = stat[k].max_end;
        enc_survey_finish(keytab.data() + k * ENC_KEY_SLOTS, lhash.data() + k * ENC_LINE_SLOTS, lfirst.data() + k * ENC_LINE_SLOTS, S[k]);
        fail[k] = S[k].fail;
        if (fail[k]) { S[k].keys.clear(); S[k].lhash.clear(); S[k].lfirst.clear(); }
        k2.insert(k2.end(), S[k].keys.begin(), S[k].keys.end()); ko[k + 1] = (uint32_t)k2.size();
        l2.insert(l2.end(), S[k].lhash.begin(), S[k].lhash.end()); lo[k + 1] = (uint32_t)l2.size();
        enc_ref_policy(S[k], refs, nrefs, false);
        start[k] = S[k].start(); multi[k] = (uint8_t)S[k].walk_mode();
        ncmax = std::max<size_t>(ncmax, (size_t)S[k].ncols());
    }
    // the reference-span digests of the slice headers: host threads, beside the device's two walks (joined before the headers are written)
    std::vector<std::thread> md5_threads;
    {
        std::shared_ptr<std::atomic<size_t>> next = std::make_shared<std::atomic<size_t>>(0);
        const unsigned nt = (unsigned)std::min<size_t>(std::min<size_t>(ns, 8), std::max(1u, std::thread::hardware_concurrency()));
        EncSlice *Sp = S.data();
        for (unsigned t = 0; t < nt; t++)
            md5_threads.emplace_back([next, Sp, ns, refs] { for (size_t k; (k = next->fetch_add(1)) < ns;) if (!Sp[k].fail) enc_ref_md5(Sp[k], refs); });
    }
    struct Joiner { std::vector<std::thread> &v; ~Joiner() { for (auto &t : v) if (t.joinable()) t.join(); } } md5_join{md5_threads};
    ok = (k2.empty() || hipMemcpyAsync(dt + t_k2, k2.data(), k2.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess) && hipMemcpyAsync(dt + t_ko, ko.data(), ko.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
         (l2.empty() || hipMemcpyAsync(dt + t_l2, l2.data(), l2.size() * 8, hipMemcpyHostToDevice, s) == hipSuccess) && hipMemcpyAsync(dt + t_lo, lo.data(), lo.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(dt + t_start, start.data(), ns * 8, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(dt + t_multi, multi.data(), ns, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(dt + t_fail, fail.data(), ns * 4, hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    E.keys = (const uint32_t *)(dt + t_k2); E.key_off = (const uint32_t *)(dt + t_ko); E.lhash = (const uint64_t *)(dt + t_l2); E.line_off = (const uint32_t *)(dt + t_lo);
    E.start = (const int64_t *)(dt + t_start); E.multi = dt + t_multi; E.ncmax = (uint32_t)ncmax;
    // ---- counting walk + prefix sums
    const size_t colb = ncmax * N * 4, totb = al(ns * ncmax * 8), baseb = al(ns * ncmax * 8);
    if ((rc = hg::ensure_scratch(ctx, 1, al(colb) + totb + baseb + 64))) return rc;
    uint8_t *d3 = (uint8_t *)ctx->d_scratch[1];
    E.col = (uint32_t *)d3; uint64_t *d_tot = (uint64_t *)(d3 + al(colb)); uint64_t *d_base = (uint64_t *)(d3 + al(colb) + totb);
    if (hipMemsetAsync(d3, 0, colb, s) != hipSuccess || hipMemsetAsync(d_tot, 0, ns * ncmax * 8, s) != hipSuccess) return HG_ELAUNCH;
    hipLaunchKernelGGL(enc_walk_kernel<false>, dim3((unsigned)nch), dim3(ENC_CHUNK), 0, s, E);
    if ((rc = launch_seg_scan(ctx, E.slices, (const uint32_t *)(dt + t_list), (uint32_t)ns, E.col, (int)ncmax, N, d_tot, s))) return rc;
    std::vector<uint64_t> tot(ns * ncmax), base(ns * ncmax, 0);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(tot.data(), d_tot, tot.size() * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(fail.data(), dt + t_fail, ns * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    PT.mark("count + sums");
    uint64_t blk_bytes = 0;
    for (size_t k = 0; k < ns; k++) {
        if (fail[k]) continue;
        for (size_t c = 0; c < (size_t)S[k].ncols(); c++) {
            if (tot[k * ncmax + c] == ~0ull) { fail[k] = -3; break; }
            base[k * ncmax + c] = blk_bytes; blk_bytes += (tot[k * ncmax + c] + 15u) & ~15ull;
        }
    }
    if ((rc = hg::ensure_scratch(ctx, 3, blk_bytes + 64))) return rc;
    E.out = (uint8_t *)ctx->d_scratch[3]; E.base = d_base;
    ok = hipMemcpyAsync(d_base, base.data(), base.size() * 8, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(dt + t_fail, fail.data(), ns * 4, hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    // ---- writing walk, blocks back
    hipLaunchKernelGGL(enc_walk_kernel<true>, dim3((unsigned)nch), dim3(ENC_CHUNK), 0, s, E);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(fail.data(), dt + t_fail, ns * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    PT.mark("write");
    for (auto &t : md5_threads) t.join();
    md5_threads.clear();
    // ---- headers + blobs: the blocks go from the device straight to their places in the caller's buffer (one pinned transfer, hg_stage.hip)
    uint64_t o = 0; bool any_bad = false, too_small = false;
    std::vector<uint64_t> src_off; std::vector<uint32_t> src_len; std::vector<uint8_t *> dst;
    for (size_t k = 0; k < ns; k++) {
        slice_off[k] = o; status[k] = fail[k];
        if (fail[k]) { any_bad = true; continue; }
        std::vector<uint8_t> comp, sh; std::vector<std::pair<int32_t, uint32_t>> blocks;
        EncCtx H{}; H.bam = bam; H.rec_off = rec_off.data(); H.rg_names = rgn.data(); H.rg_off = rgo.data(); H.nrg = nrg;
        enc_headers(S[k], H, tot.data() + k * ncmax, record_counter0 + (int64_t)S[k].r0, comp, sh, blocks);
        uint64_t need = 12 + comp.size() + sh.size();
        for (auto &b : blocks) need += 8 + tot[k * ncmax + b.second];
        if (o + need > out_cap) { too_small = true; o += need; continue; }
        auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) out[o++] = (uint8_t)(v >> (8 * i)); };
        put32((uint32_t)comp.size()); memcpy(out + o, comp.data(), comp.size()); o += comp.size();
        put32((uint32_t)sh.size()); memcpy(out + o, sh.data(), sh.size()); o += sh.size();
        put32((uint32_t)blocks.size());
        for (auto &b : blocks) {
            const uint64_t n = tot[k * ncmax + b.second];
            put32((uint32_t)b.first); put32((uint32_t)n);
            src_off.push_back(base[k * ncmax + b.second]); src_len.push_back((uint32_t)n); dst.push_back(out + o);
            o += n;
        }
    }
    if (!src_off.empty() && (rc = hg::stage_download(ctx, E.out, src_off.data(), src_len.data(), dst.data(), src_off.size(), s)) != HG_OK) return rc;
    slice_off[ns] = o;
    if (out_bytes) *out_bytes = o;
    PT.mark("headers + blocks back");
    return too_small ? HG_ENOMEM : any_bad ? HG_EBLOCK : HG_OK;
}

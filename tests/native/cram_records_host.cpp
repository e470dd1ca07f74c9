// cram_records_host.cpp -- TEST INFRASTRUCTURE: compiles the record decoder of htslib_amd/csrc/cram_records_core.h for the CPU,
// with the same batch layout the device launcher uses (cram_records_plan.h), so that the logic can be checked against the
// reference's SAM twins on a box without a GPU (tests/test_cram_records.py).  The product library never runs this.
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "../../htslib_amd/csrc/cram_records_plan.h"

struct record_cols {            // = hg_cram_record_cols (include/htsgpu.h)
    int32_t *flags, *cram_flags, *ref_id, *len, *rg, *mqual, *mate_ref_id, *ncigar, *name_len;
    int64_t *apos, *aend, *mate_pos, *tlen;
    uint64_t *cigar_off, *name_off;
    uint32_t *cigar; uint8_t *names;
    uint64_t *seq_off; uint8_t *seq, *qual;
    uint64_t *aux_off; int32_t *aux_len; uint8_t *aux;
};

extern "C" int hgr_host_decode_records(size_t nslices, const hgr::SliceIn *in, int major, int nref, size_t rec_cap, size_t cigar_cap, size_t name_cap,
                                       size_t seq_cap, size_t aux_cap, const record_cols *out, uint64_t *rec_off, int32_t *status) {
    hgr::Batch B;
    const int rc = hgr::batch_build(B, in, nslices, major);
    if (rc) return rc;
    if (B.nrec > rec_cap || B.cig_total > cigar_cap || B.name_total > name_cap || (out->aux && B.aux_total > aux_cap)) return -5;
    std::vector<uint8_t> data(B.data_bytes + 16);
    for (size_t k = 0; k < B.src_ptr.size(); k++) if (B.src_len[k]) memcpy(data.data() + B.src_off[k], B.src_ptr[k], B.src_len[k]);
    unsigned long long pool = 0;
    std::vector<hgr::CopyJob> jobs(B.job_total + 1);
    std::vector<int32_t> mate_flags(B.nrec + 1), mate_line(B.nrec + 1); std::vector<int64_t> etlen(B.nrec + 1); std::vector<uint32_t> coff(B.nrec + 1), noff(B.nrec + 1), aoff(B.nrec + 1);
    for (size_t i = 0; i < nslices; i++) {
        const hgr::SliceDev &d = B.slices[i];
        rec_off[i] = d.rec_off;
        status[i] = B.status[i];
        if (status[i]) continue;
        const hgr::PlanDev &pd = B.plans[d.plan];
        hgr::Plan P; memcpy(P.codec_of, pd.codec_of, sizeof P.codec_of);
        P.sm = &pd.sm[0][0];
        P.rn_included = pd.rn_included; P.ap_delta = pd.ap_delta; P.qs_seq_orient = pd.qs_seq_orient; P.nslots = pd.nslots; P.nTL = pd.nTL;
        P.tl_off = B.tl_off.data() + pd.tl_off_base; P.tl_codec = B.tl_codec.data() + pd.tl_codec_base; P.tl_tag = B.tl_tag.data() + pd.tl_codec_base; P.codecs = B.codecs.data() + pd.codec_base; P.huff = B.huff.data() + pd.huff_base;
        hgr::Slice S; S.data = data.data(); S.blk_off = B.tab.data() + d.tab_off; S.blk_len = S.blk_off + pd.nslots; S.cursor = B.tab.data() + d.tab_off + 2 * pd.nslots;
        S.core_off = d.core_off; S.core_len = d.core_len; S.nrec = d.nrec; S.ref_seq_id = d.ref_seq_id; S.ref_seq_start = d.ref_seq_start; S.nref = nref;
        S.cigar_cap = d.cig_cap; S.name_cap = d.name_cap; S.aux_cap = d.aux_cap; S.refs = B.refs.data() + d.ref_first; S.nrefs = (int32_t)d.nrefs; S.decode_md = d.decode_md; S.jobs = jobs.data() + d.job_off; S.job_cap = d.job_cap;
        std::vector<uint32_t> wbuf32(32 * ((size_t)pd.nslots + 2)), wpos((size_t)pd.nslots + 2);      // the read-ahead windows the device keeps in LDS
        S.wbuf = (uint8_t *)wbuf32.data(); S.wpos = wpos.data();
        uint32_t totals[4] = {0, 0, 0, 0};
        const uint64_t r0 = d.rec_off;
        hgr::Cols O{out->flags + r0, out->cram_flags + r0, out->ref_id + r0, out->len + r0, out->rg + r0, out->mqual + r0, mate_flags.data() + r0, out->mate_ref_id + r0,
                    mate_line.data() + r0, out->ncigar + r0, out->name_len + r0, coff.data() + r0, noff.data() + r0, out->apos + r0, out->aend + r0, out->mate_pos + r0,
                    out->tlen + r0, etlen.data() + r0, out->cigar + d.cig_off, out->names + d.name_off, totals, out->aux ? out->aux + d.aux_off : nullptr, aoff.data() + r0, out->aux ? out->aux_len + r0 : nullptr,
                    out->seq, out->qual, out->seq ? out->seq_off + r0 : nullptr, &pool, seq_cap};
        status[i] = hgr::decode_slice(&P, &S, O);
        for (uint32_t j = 0; j < totals[3]; j++) hgr::copy_bytes(S.jobs[j].dst, S.jobs[j].src, S.jobs[j].n);     // the deferred bulk copies (one per lane on the device)
        for (int32_t r = 0; r < d.nrec; r++) { out->cigar_off[r0 + r] = d.cig_off + coff[r0 + r]; out->name_off[r0 + r] = d.name_off + noff[r0 + r]; if (out->aux) out->aux_off[r0 + r] = d.aux_off + aoff[r0 + r]; }
    }
    rec_off[nslices] = B.nrec;
    return 0;
}
// capacities a caller must provide for these slices (the product exports the same helper)
extern "C" int hgr_host_records_bound(size_t nslices, const hgr::SliceIn *in, int major, uint64_t *nrec, uint64_t *cigar_cap, uint64_t *name_cap, uint64_t *aux_cap) {
    hgr::Batch B;
    const int rc = hgr::batch_build(B, in, nslices, major);
    if (rc) return rc;
    *nrec = B.nrec; *cigar_cap = B.cig_total; *name_cap = B.name_total; *aux_cap = B.aux_total;
    return 0;
}

// ---- the data-parallel path (cram_records_fast.h) run from plain loops: the per-record passes are the very functions the kernels of
//      cram_records.hip call with one thread per record, the prefix sums and the column decodes are restated here as serial loops.
//      Slices the path does not take (or gives up on) go through the chain decoder above, exactly as the device launcher does. ----
#include "../../htslib_amd/csrc/cram_records_fast_plan.h"

namespace {
// exclusive prefix sum in place; the total comes back (the device kernel flags a total that does not fit 32 bits)
uint64_t exscan32(uint32_t *v, size_t n) { uint64_t run = 0; for (size_t i = 0; i < n; i++) { const uint32_t x = v[i]; v[i] = (uint32_t)run; run += x; } return run; }

// whole-block column decodes: what itf8_decode_kernel / byte_array_stop_kernel (cram_series.hip) and the running-sum kernel leave
void host_columns(const hgr::FastBatch &F, const uint8_t *data, std::vector<uint32_t> &pool, std::vector<uint32_t> &col_n, std::vector<int32_t> &col_st,
                  std::vector<uint64_t> &col_off) {
    pool.assign(F.pool_words + 4, 0u); col_n.assign(F.ncols() + 1, 0u); col_st.assign(F.ncols() + 1, 0); col_off.assign(F.ncols() + 1, 0);
    size_t c = 0;
    for (const hgr::FastCol &q : F.itf8) {
        col_off[c] = q.pool_off;
        hgr::Cursor cur{data + q.in_off, data + q.in_off + q.in_len};
        uint32_t n = 0; bool bad = false;
        while (cur.p < cur.end) { const int32_t v = cur.itf8(); if (cur.bad) { bad = true; break; } pool[q.pool_off + n++] = (uint32_t)v; }
        col_n[c] = bad ? 0u : n; col_st[c] = bad ? -1 : 0; c++;
    }
    for (const hgr::FastCol &q : F.stop) {
        col_off[c] = q.pool_off;
        uint32_t n = 0, last = 0; pool[q.pool_off] = 0;
        for (uint32_t i = 0; i < q.in_len; i++) if (data[q.in_off + i] == (uint8_t)q.stop) { pool[q.pool_off + ++n] = i + 1; last = i + 1; }
        const bool bad = last != q.in_len;
        col_n[c] = bad ? 0u : n; col_st[c] = bad ? -1 : 0; c++;
    }
    for (const hgr::FastCol &q : F.sums) {
        col_off[c] = q.pool_off;
        const uint32_t n = col_n[q.src]; const uint32_t *v = pool.data() + col_off[q.src];
        uint64_t run = 0; bool poison = false; pool[q.pool_off] = 0;
        for (uint32_t i = 0; i < n; i++) {
            if ((int32_t)v[i] < 0) poison = true;
            run += v[i]; if (run > 0xfffffff0ull) poison = true;
            pool[q.pool_off + i + 1] = poison ? 0xffffffffu : (uint32_t)run;
        }
        col_n[c] = n; col_st[c] = col_st[q.src]; c++;
    }
}
}  // namespace

extern "C" int hgr_host_decode_records_fast(size_t nslices, const hgr::SliceIn *in, int major, int nref, size_t rec_cap, size_t cigar_cap, size_t name_cap,
                                            size_t seq_cap, size_t aux_cap, const record_cols *out, uint64_t *rec_off, int32_t *status, int32_t *path) {
    using namespace hgr;
    Batch B;
    const int rc = batch_build(B, in, nslices, major);
    if (rc) return rc;
    if (B.nrec > rec_cap || B.cig_total > cigar_cap || B.name_total > name_cap || (out->aux && B.aux_total > aux_cap)) return -5;
    FastBatch F; fast_build(B, F, true);
    std::vector<uint8_t> data(B.data_bytes + 16);
    for (size_t k = 0; k < B.src_ptr.size(); k++) if (B.src_len[k]) memcpy(data.data() + B.src_off[k], B.src_ptr[k], B.src_len[k]);
    std::vector<uint32_t> pool, col_n; std::vector<int32_t> col_st; std::vector<uint64_t> col_off;
    host_columns(F, data.data(), pool, col_n, col_st, col_off);
    const size_t N = B.nrec + 1;
    std::vector<int32_t> mate_flags(N), mate_line(N), pred(N, 0), fail(nslices + 1, 0), unclean(nslices + 1, 0); std::vector<int64_t> etlen(N), ap(N);
    std::vector<uint32_t> coff(N), noff(N), aoff(N), c_det(N), c_down(N), c_ts(N), c_map(N), seq_at(N), fnc(N), work(N), aux_stored(N);
    std::vector<uint32_t> tag((size_t)(F.ntag_max ? F.ntag_max : 1) * N), cls((size_t)NCLS * N);
    std::vector<uint8_t> bits(N);
    std::vector<uint64_t> seq_total(nslices + 1, 0), seq_base(nslices + 1, 0);
    std::vector<uint32_t> totals(4 * nslices + 4, 0);
    const bool want_aux = out->aux != nullptr, want_seq = out->seq != nullptr;
    // columns that failed to decode hand their slice to the chain decoder
    { size_t c = 0; for (const auto *L : {&F.itf8, &F.stop, &F.sums}) for (const FastCol &q : *L) { if (col_st[c]) fail[q.slice] = 1; c++; } }
    struct Ctx { FCtx C; Plan P; Cols O; };
    auto make = [&](size_t i, Ctx &X) {
        const SliceDev &d = B.slices[i]; const PlanDev &pd = B.plans[d.plan];
        Plan &P = X.P; memset(&P, 0, sizeof P);
        P.sm = &pd.sm[0][0]; P.rn_included = pd.rn_included; P.ap_delta = pd.ap_delta; P.qs_seq_orient = pd.qs_seq_orient; P.nslots = pd.nslots; P.nTL = pd.nTL;
        P.tl_off = B.tl_off.data() + pd.tl_off_base; P.tl_codec = B.tl_codec.data() + pd.tl_codec_base; P.tl_tag = B.tl_tag.data() + pd.tl_codec_base;
        P.codecs = B.codecs.data() + pd.codec_base; P.huff = B.huff.data() + pd.huff_base;
        FCtx &C = X.C;
        C.P = &X.P; C.ser = F.ser.data() + F.ser_off[i]; C.tl_tagidx = F.tl_tagidx.data() + pd.tl_codec_base; C.ntag = F.ntag[i];
        C.V = FView{data.data(), pool.data(), col_off.data(), col_n.data()};
        C.Z = FScr{c_det.data(), c_down.data(), c_ts.data(), c_map.data(), seq_at.data(), ap.data(), fnc.data(), work.data(), tag.data(), cls.data(), aux_stored.data(),
                   bits.data(), pred.data(), (uint64_t)N};
        C.rec_off = d.rec_off; C.nrec = d.nrec; C.ref_seq_id = d.ref_seq_id; C.nref = nref; C.ref_seq_start = d.ref_seq_start;
        C.refs = B.refs.data() + d.ref_first; C.nrefs = (int32_t)d.nrefs; C.decode_md = d.decode_md;
        C.cig_cap = d.cig_cap; C.name_cap = d.name_cap; C.aux_cap = d.aux_cap; C.fail = &fail[i]; C.want_aux = want_aux;
        const uint64_t r0 = d.rec_off;
        X.O = Cols{out->flags + r0, out->cram_flags + r0, out->ref_id + r0, out->len + r0, out->rg + r0, out->mqual + r0, mate_flags.data() + r0, out->mate_ref_id + r0,
                   mate_line.data() + r0, out->ncigar + r0, out->name_len + r0, coff.data() + r0, noff.data() + r0, out->apos + r0, out->aend + r0, out->mate_pos + r0,
                   out->tlen + r0, etlen.data() + r0, out->cigar + d.cig_off, out->names + d.name_off, totals.data() + 4 * i, want_aux ? out->aux + d.aux_off : nullptr, aoff.data() + r0,
                   want_aux ? out->aux_len + r0 : nullptr, want_seq ? out->seq : nullptr, want_seq ? out->qual : nullptr, want_seq ? out->seq_off + r0 : nullptr, nullptr, seq_cap};
    };
    for (uint32_t i : F.fast_list) {                                      // passes 1-4 and their prefix sums, slice by slice
        Ctx X; make(i, X);
        const FCtx &C = X.C; const SliceDev &d = B.slices[i]; const size_t r0 = d.rec_off, n = (size_t)d.nrec;
        for (uint32_t r = 0; r < n; r++) fast_m1(C, r);
        bool over = false;
        for (uint32_t *col : {c_det.data(), c_down.data(), c_ts.data(), c_map.data()}) exscan32(col + r0, n);
        seq_total[i] = exscan32(seq_at.data() + r0, n); over |= seq_total[i] > 0xffffffffull;
        if (X.P.ap_delta) { int64_t run = d.ref_seq_start; for (size_t r = 0; r < n; r++) { run += ap[r0 + r]; ap[r0 + r] = run; } }
        for (uint32_t r = 0; r < n; r++) fast_m2(C, r, noff.data());
        over |= exscan32(fnc.data() + r0, n) > 0xffffffffull;
        const uint64_t name_total = exscan32(noff.data() + r0, n), work_total = exscan32(work.data() + r0, n);
        for (uint32_t k = 0; k < C.ntag; k++) exscan32(tag.data() + (size_t)k * N + r0, n);
        for (uint32_t r = 0; r < n; r++) fast_m3(C, r);
        for (int j = 0; j < NCLS; j++) over |= exscan32(cls.data() + (size_t)j * N + r0, n) > 0xffffffffull;
        for (uint32_t r = 0; r < n; r++) fast_m4(C, r, coff.data(), aoff.data(), want_aux ? out->aux : nullptr);
        const uint64_t cig_total = exscan32(coff.data() + r0, n), aux_total = exscan32(aoff.data() + r0, n);
        if (over || name_total > d.name_cap || cig_total > d.cig_cap || aux_total > d.aux_cap || work_total > 16ull * d.cig_cap) fail[i] = 1;
        totals[4 * i] = (uint32_t)cig_total; totals[4 * i + 1] = (uint32_t)name_total; totals[4 * i + 2] = (uint32_t)aux_total; totals[4 * i + 3] = 0;
        if (fail[i]) seq_total[i] = 0;
    }
    unsigned long long pool_at = 0;                                       // the fast slices' stretches of seq[] / qual[], in slice order; the chain decoder's pool follows
    for (uint32_t i : F.fast_list) { seq_base[i] = pool_at; pool_at += want_seq ? seq_total[i] : 0; }
    for (uint32_t i : F.fast_list) {                                      // pass 5 and the mates
        if (fail[i]) continue;
        Ctx X; make(i, X);
        const FCtx &C = X.C; const uint32_t n = (uint32_t)B.slices[i].nrec;
        for (uint32_t r = 0; r < n; r++) fast_m5(C, r, X.O, seq_base[i]);
        if (fail[i]) continue;
        for (uint32_t r = 0; r < n; r++) fast_xa(C, r, X.O, &unclean[i]);
        if (unclean[i]) { if (xref(X.O, (int32_t)n)) fail[i] = 1; continue; }
        for (uint32_t r = 0; r < n; r++) fast_xb(C, r, X.O);
        for (uint32_t r = 0; r < n; r++) fast_xc(C, r, X.O);
    }
    // everything else: the chain decoder
    std::vector<hgr::CopyJob> jobs(B.job_total + 1);
    for (size_t i = 0; i < nslices; i++) {
        const hgr::SliceDev &d = B.slices[i];
        rec_off[i] = d.rec_off;
        status[i] = B.status[i];
        path[i] = F.is_fast[i] && !fail[i] ? 1 : 0;
        const uint64_t r0 = d.rec_off;
        if (status[i] == 0 && !path[i]) {
            const hgr::PlanDev &pd = B.plans[d.plan];
            hgr::Plan P; memcpy(P.codec_of, pd.codec_of, sizeof P.codec_of);
            P.sm = &pd.sm[0][0];
            P.rn_included = pd.rn_included; P.ap_delta = pd.ap_delta; P.qs_seq_orient = pd.qs_seq_orient; P.nslots = pd.nslots; P.nTL = pd.nTL;
            P.tl_off = B.tl_off.data() + pd.tl_off_base; P.tl_codec = B.tl_codec.data() + pd.tl_codec_base; P.tl_tag = B.tl_tag.data() + pd.tl_codec_base; P.codecs = B.codecs.data() + pd.codec_base; P.huff = B.huff.data() + pd.huff_base;
            hgr::Slice S; S.data = data.data(); S.blk_off = B.tab.data() + d.tab_off; S.blk_len = S.blk_off + pd.nslots; S.cursor = B.tab.data() + d.tab_off + 2 * pd.nslots;
            S.core_off = d.core_off; S.core_len = d.core_len; S.nrec = d.nrec; S.ref_seq_id = d.ref_seq_id; S.ref_seq_start = d.ref_seq_start; S.nref = nref;
            S.cigar_cap = d.cig_cap; S.name_cap = d.name_cap; S.aux_cap = d.aux_cap; S.refs = B.refs.data() + d.ref_first; S.nrefs = (int32_t)d.nrefs; S.decode_md = d.decode_md; S.jobs = jobs.data() + d.job_off; S.job_cap = d.job_cap;
            S.wbuf = nullptr; S.wpos = nullptr;
            hgr::Cols O{out->flags + r0, out->cram_flags + r0, out->ref_id + r0, out->len + r0, out->rg + r0, out->mqual + r0, mate_flags.data() + r0, out->mate_ref_id + r0,
                        mate_line.data() + r0, out->ncigar + r0, out->name_len + r0, coff.data() + r0, noff.data() + r0, out->apos + r0, out->aend + r0, out->mate_pos + r0,
                        out->tlen + r0, etlen.data() + r0, out->cigar + d.cig_off, out->names + d.name_off, totals.data() + 4 * i, out->aux ? out->aux + d.aux_off : nullptr, aoff.data() + r0, out->aux ? out->aux_len + r0 : nullptr,
                        out->seq, out->qual, out->seq ? out->seq_off + r0 : nullptr, &pool_at, seq_cap};
            status[i] = hgr::decode_slice(&P, &S, O);
            for (uint32_t j = 0; j < totals[4 * i + 3]; j++) hgr::copy_bytes(S.jobs[j].dst, S.jobs[j].src, S.jobs[j].n);
        }
        for (int32_t r = 0; r < d.nrec; r++) { out->cigar_off[r0 + r] = d.cig_off + coff[r0 + r]; out->name_off[r0 + r] = d.name_off + noff[r0 + r]; if (out->aux) out->aux_off[r0 + r] = d.aux_off + aoff[r0 + r]; }
    }
    rec_off[nslices] = B.nrec;
    return 0;
}

// ---- the ENCODER (cram_encode_core.h / cram_encode_plan.h) run from plain loops, in the order of the kernels of cram_encode.hip: tag survey,
//      counting walk, prefix sums per slice, writing walk, headers.  Output: per slice a blob  u32 comp_len, comp, u32 slice_hdr_len, slice_hdr,
//      u32 nblocks, then per block i32 content id, u32 len, bytes  (the layout of tests/native/cram_encode_proto.cpp); slice_off[i] .. [i + 1]. ----
#include "../../htslib_amd/csrc/cram_encode_plan.h"

struct ref_seq_in { const uint8_t *bases; uint64_t len; };
extern "C" long hgr_host_encode_slices(const uint8_t *bam, size_t bam_len, size_t nrec_in, uint32_t per_slice, const ref_seq_in *refs, int nrefs, const char *const *rg_names, int nrg,
                                       int64_t record_counter0, uint8_t *out, size_t cap, uint64_t *slice_off, size_t max_slices, int32_t *status) {
    using namespace hgr;
    std::vector<uint64_t> rec_off;                                       // frame the records (bam_read1: block_size + bytes)
    for (uint64_t at = 0; at + 4 <= bam_len && rec_off.size() < nrec_in;) { rec_off.push_back(at); at += 4ull + ld32(bam + at); if (at > bam_len) return -1; }
    if (rec_off.size() != nrec_in) return -1;
    rec_off.push_back(rec_off.empty() ? 0 : rec_off.back() + 4ull + ld32(bam + rec_off.back()));
    const size_t n = nrec_in, ns = per_slice ? (n + per_slice - 1) / per_slice : 0;
    if (ns > max_slices) return -5;
    std::vector<uint8_t> data; std::vector<EncRef> er((size_t)nrefs);
    for (int i = 0; i < nrefs; i++) { er[(size_t)i].off = data.size(); er[(size_t)i].len = refs[i].bases ? (int64_t)refs[i].len : 0; if (refs[i].bases) data.insert(data.end(), refs[i].bases, refs[i].bases + refs[i].len); }
    std::vector<uint8_t> rgn; std::vector<uint32_t> rgo((size_t)nrg + 1, 0);
    for (int i = 0; i < nrg; i++) { rgn.insert(rgn.end(), rg_names[i], rg_names[i] + strlen(rg_names[i])); rgo[(size_t)i + 1] = (uint32_t)rgn.size(); }
    std::vector<EncSlice> S(ns);
    std::vector<uint32_t> keytab(ns * ENC_KEY_SLOTS, ENC_EMPTY), lfirst(ns * ENC_LINE_SLOTS, 0xffffffffu); std::vector<uint64_t> lhash(ns * ENC_LINE_SLOTS, 0), lcheck(ns * ENC_LINE_SLOTS, 0);
    std::vector<int32_t> fail(ns + 1, 0);
    auto ctx_of = [&](size_t k) {
        EncCtx C{}; C.bam = bam; C.rec_off = rec_off.data(); C.data = data.data(); C.refs = er.data(); C.nref = nrefs; C.rg_names = rgn.data(); C.rg_off = rgo.data(); C.nrg = nrg;
        C.r0 = S[k].r0; C.nrec = S[k].nrec; C.keys = S[k].keys.data(); C.nkeys = (uint32_t)S[k].keys.size(); C.line_hash = S[k].lhash.data(); C.nlines = (uint32_t)S[k].lhash.size(); C.fail = &fail[k];
        return C;
    };
    const EncSurvey V{keytab.data(), lhash.data(), lcheck.data(), lfirst.data()};
    for (size_t k = 0; k < ns; k++) {                                    // survey + slice statistics
        S[k].r0 = k * per_slice; S[k].nrec = (uint32_t)std::min<size_t>(per_slice, n - S[k].r0);
        const EncCtx C = ctx_of(k);
        int32_t lo = INT32_MAX, hi = INT32_MIN; int64_t p0 = INT64_MAX, p1 = INT64_MIN;
        for (uint32_t r = 0; r < S[k].nrec; r++) {
            enc_survey_record(C, r, V, (uint32_t)k);
            BamRec B;
            if (!bam_parse(bam, rec_off[S[k].r0 + r], rec_off[S[k].r0 + r + 1], B)) continue;
            int64_t rl = 0;
            if (!(B.flag & BAM_FUNMAP)) for (uint32_t c = 0; c < B.n_cigar; c++) { const uint32_t cw = ld32(B.cigar + 4 * c), op = cw & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cw >> 4; }
            const int64_t ap = (int64_t)B.pos + 1, ae = rl ? ap + rl - 1 : ap;
            lo = std::min(lo, B.ref_id); hi = std::max(hi, B.ref_id); p0 = std::min(p0, ap); p1 = std::max(p1, ae);
        }
        S[k].min_ref = lo; S[k].max_ref = hi; S[k].min_pos = p0; S[k].max_end = p1; S[k].fail = fail[k];
        enc_survey_finish(keytab.data() + k * ENC_KEY_SLOTS, lhash.data() + k * ENC_LINE_SLOTS, lfirst.data() + k * ENC_LINE_SLOTS, S[k]);
        fail[k] = S[k].fail;
        enc_ref_policy(S[k], refs, nrefs);
    }
    size_t ncmax = W_N; for (auto &s : S) ncmax = std::max<size_t>(ncmax, (size_t)s.ncols());
    const size_t N = n + 1;
    std::vector<uint32_t> col(ncmax * N, 0u);
    std::vector<std::vector<uint64_t>> tot(ns), base(ns);
    auto prev_of = [&](size_t k, uint32_t r) -> int64_t { if (r == 0) return S[k].start(); BamRec B; bam_parse(bam, rec_off[S[k].r0 + r - 1], rec_off[S[k].r0 + r], B); return (int64_t)B.pos + 1; };
    uint64_t out_bytes = 0;
    for (size_t k = 0; k < ns; k++) {                                    // counting walk + prefix sums
        if (fail[k]) continue;
        const EncCtx C = ctx_of(k);
        const int nc = S[k].ncols();
        for (uint32_t r = 0; r < S[k].nrec; r++) {
            Sink<false> K{}; K.col = col.data(); K.N = N; K.g = S[k].r0 + r;
            if (!enc_record<false>(C, r, prev_of(k, r), S[k].walk_mode(), K)) break;
            for (int s = 0; s < W_N; s++) col[(size_t)s * N + K.g] = K.n[s];
        }
        if (fail[k]) continue;
        tot[k].assign((size_t)nc, 0); base[k].assign((size_t)nc, 0);
        for (int c = 0; c < nc; c++) {
            uint64_t run = 0;
            for (uint32_t r = 0; r < S[k].nrec; r++) { uint32_t &x = col[(size_t)c * N + S[k].r0 + r]; const uint32_t v = x; x = (uint32_t)run; run += v; }
            if (run > 0xffffffffull) fail[k] = -3;
            tot[k][(size_t)c] = run; base[k][(size_t)c] = out_bytes; out_bytes += (run + 15u) & ~15ull;
        }
    }
    std::vector<uint8_t> blk(out_bytes + 64);
    for (size_t k = 0; k < ns; k++) {                                    // writing walk
        if (fail[k]) continue;
        const EncCtx C = ctx_of(k);
        for (uint32_t r = 0; r < S[k].nrec; r++) {
            Sink<true> K{}; K.col = col.data(); K.N = N; K.g = S[k].r0 + r; K.out = blk.data(); K.base = base[k].data();
            for (int s = 0; s < W_N; s++) K.p[s] = blk.data() + base[k][(size_t)s] + col[(size_t)s * N + K.g];
            if (!enc_record<true>(C, r, prev_of(k, r), S[k].walk_mode(), K)) break;
        }
    }
    uint64_t at = 0;
    for (size_t k = 0; k < ns; k++) {                                    // headers + blob
        slice_off[k] = at; status[k] = fail[k];
        if (fail[k]) continue;
        std::vector<uint8_t> comp, sh; std::vector<std::pair<int32_t, uint32_t>> blocks;
        enc_headers(S[k], ctx_of(k), tot[k].data(), record_counter0 + (int64_t)S[k].r0, comp, sh, blocks);
        uint64_t need = 12 + comp.size() + sh.size();
        for (auto &b : blocks) need += 8 + tot[k][b.second];
        if (at + need > cap) return -5;
        auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) out[at++] = (uint8_t)(v >> (8 * i)); };
        put32((uint32_t)comp.size()); memcpy(out + at, comp.data(), comp.size()); at += comp.size();
        put32((uint32_t)sh.size()); memcpy(out + at, sh.data(), sh.size()); at += sh.size();
        put32((uint32_t)blocks.size());
        for (auto &b : blocks) { put32((uint32_t)b.first); put32((uint32_t)tot[k][b.second]); memcpy(out + at, blk.data() + base[k][b.second], tot[k][b.second]); at += tot[k][b.second]; }
    }
    slice_off[ns] = at;
    return (long)ns;
}

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

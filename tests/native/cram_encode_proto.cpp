// cram_encode_proto.cpp -- TEST INFRASTRUCTURE, a CPU prototype of the ENCODE side of SURVEY 8f N2 (cram_encode_slice, reference
// cram/cram_encode.c:572-793, 1096-1209; feature generation process_one_read :3382-3700) in the column form the device will use:
// records (as the pinned decoder hands them out) -> one column per data series -> EXTERNAL blocks + compression / slice headers.
// Every step is a map over records (features of a record from its CIGAR, bases and the reference), a prefix sum, or a whole-column
// ITF8 pack (hg_cram_itf8_encode_dev on the device).  Choices the format leaves free are made the simple way: every series EXTERNAL in a
// block of its own, every record detached (mate fields stored, no in-slice mate links), default substitution matrix, names kept,
// qualities kept, tags as BYTE_ARRAY_LEN with EXTERNAL length and value blocks.
// Checked by tests/test_cram_records.py: the reference's fixtures and synthetic slices are decoded with the pinned chain decoder,
// re-encoded here, decoded again -- every field must survive.
#include <stdint.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "../../htslib_amd/csrc/cram_records_plan.h"

using namespace hgr;

namespace {

void itf8(std::vector<uint8_t> &o, int32_t sv) {
    const uint32_t v = (uint32_t)sv;
    if (v < 0x80) o.push_back((uint8_t)v);
    else if (v < 0x4000) { o.push_back((uint8_t)(0x80 | (v >> 8))); o.push_back((uint8_t)v); }
    else if (v < 0x200000) { o.push_back((uint8_t)(0xC0 | (v >> 16))); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
    else if (v < 0x10000000) { o.push_back((uint8_t)(0xE0 | (v >> 24))); o.push_back((uint8_t)(v >> 16)); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
    else { o.push_back((uint8_t)(0xF0 | (v >> 28))); o.push_back((uint8_t)(v >> 20)); o.push_back((uint8_t)(v >> 12)); o.push_back((uint8_t)(v >> 4)); o.push_back((uint8_t)(v & 0x0f)); }
}
std::vector<uint8_t> sized(const std::vector<uint8_t> &body) { std::vector<uint8_t> o; itf8(o, (int32_t)body.size()); o.insert(o.end(), body.begin(), body.end()); return o; }
void app(std::vector<uint8_t> &o, const std::vector<uint8_t> &x) { o.insert(o.end(), x.begin(), x.end()); }
std::vector<uint8_t> enc_external(int32_t cid) { std::vector<uint8_t> p, o; itf8(p, cid); itf8(o, E_EXTERNAL); app(o, sized(p)); return o; }
std::vector<uint8_t> enc_stop(uint8_t stop, int32_t cid) { std::vector<uint8_t> p{stop}, o; itf8(p, cid); itf8(o, E_BYTE_ARRAY_STOP); app(o, sized(p)); return o; }
std::vector<uint8_t> enc_len(int32_t len_cid, int32_t val_cid) { std::vector<uint8_t> p, o; app(p, enc_external(len_cid)); app(p, enc_external(val_cid)); itf8(o, E_BYTE_ARRAY_LEN); app(o, sized(p)); return o; }

// bytes of one BAM aux value of the given type at p (bounded by end); 0 = malformed
size_t aux_value_size(uint8_t type, const uint8_t *p, const uint8_t *end) {
    auto elem = [](uint8_t t) -> size_t { return (t == 'c' || t == 'C' || t == 'A') ? 1 : (t == 's' || t == 'S') ? 2 : (t == 'i' || t == 'I' || t == 'f') ? 4 : 0; };
    if (elem(type)) return elem(type);
    if (type == 'Z' || type == 'H') { const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p)); return z ? (size_t)(z - p) + 1 : 0; }
    if (type == 'B') { if (end - p < 5 || !elem(p[0])) return 0; const uint32_t n = (uint32_t)p[1] | (uint32_t)p[2] << 8 | (uint32_t)p[3] << 16 | (uint32_t)p[4] << 24; return 5 + (size_t)n * elem(p[0]); }
    return 0;
}

const char *SERIES_NAME[S_N] = {"BF", "CF", "RI", "RL", "AP", "RG", "RN", "MF", "NS", "NP", "TS", "NF", "TL", "FN", "FC", "FP", "DL", "BA", "BS", "IN", "SC", "HC", "PD", "RS", "MQ", "QS", "BB", "QQ"};
const char *SM[5] = {"CGTN", "AGTN", "ACTN", "ACGN", "ACGT"};

}  // namespace

// Decodes the slice with the chain decoder, re-encodes the records, writes: u32 comp_len, comp, u32 slice_hdr_len, slice_hdr, u32 nblocks, then per block
// i32 content id, u32 len, bytes.  Returns the blob size, or a negative code (-3 = records this prototype does not cover).
extern "C" long hgr_proto_reencode_slice(const SliceIn *in, int major, int nref, uint8_t *out, size_t cap) {
    // ---- the records, from the pinned decoder ----
    Batch B;
    if (batch_build(B, in, 1, major) || B.status[0]) return -1;
    const SliceDev &d = B.slices[0];
    const PlanDev &pd = B.plans[d.plan];
    std::vector<uint8_t> data(B.data_bytes + 16);
    for (size_t k = 0; k < B.src_ptr.size(); k++) if (B.src_len[k]) memcpy(data.data() + B.src_off[k], B.src_ptr[k], B.src_len[k]);
    Plan P; memcpy(P.codec_of, pd.codec_of, sizeof P.codec_of);
    P.sm = &pd.sm[0][0]; P.rn_included = pd.rn_included; P.ap_delta = pd.ap_delta; P.qs_seq_orient = pd.qs_seq_orient; P.nslots = pd.nslots; P.nTL = pd.nTL;
    P.tl_off = B.tl_off.data() + pd.tl_off_base; P.tl_codec = B.tl_codec.data() + pd.tl_codec_base; P.tl_tag = B.tl_tag.data() + pd.tl_codec_base;
    P.codecs = B.codecs.data() + pd.codec_base; P.huff = B.huff.data() + pd.huff_base;
    Slice S; S.data = data.data(); S.blk_off = B.tab.data() + d.tab_off; S.blk_len = S.blk_off + pd.nslots; S.cursor = B.tab.data() + d.tab_off + 2 * pd.nslots;
    S.core_off = d.core_off; S.core_len = d.core_len; S.nrec = d.nrec; S.ref_seq_id = d.ref_seq_id; S.ref_seq_start = d.ref_seq_start; S.nref = nref;
    S.cigar_cap = d.cig_cap; S.name_cap = d.name_cap; S.aux_cap = d.aux_cap; S.refs = B.refs.data() + d.ref_first; S.nrefs = (int32_t)d.nrefs; S.decode_md = 0;
    S.jobs = nullptr; S.job_cap = 0; S.wbuf = nullptr; S.wpos = nullptr;
    const size_t n = (size_t)d.nrec;
    std::vector<int32_t> flags(n + 1), cflags(n + 1), ref_id(n + 1), len(n + 1), rg(n + 1), mq(n + 1), mflags(n + 1), mref(n + 1), mline(n + 1), ncig(n + 1), nlen(n + 1), alen(n + 1);
    std::vector<uint32_t> coff(n + 1), noff(n + 1), aoff(n + 1), cig(d.cig_cap + 1);
    std::vector<int64_t> apos(n + 1), aend(n + 1), mpos(n + 1), tlen(n + 1), etlen(n + 1);
    std::vector<uint8_t> names(d.name_cap + 1), aux(d.aux_cap + 1);
    uint64_t bases = 0; { SliceHeader sh; (void)sh; }
    std::vector<uint64_t> soff(n + 1);
    unsigned long long pool = 0;
    // room for bases: decode lengths first (cheap second pass would do the same on the device)
    size_t seq_cap = 1 << 16;
    std::vector<uint8_t> seq, qual;
    int rc = 0;
    for (int attempt = 0; attempt < 12; attempt++, seq_cap *= 8) {
        seq.assign(seq_cap, 0); qual.assign(seq_cap, 0); pool = 0;
        uint32_t totals[4] = {0, 0, 0, 0};
        Cols O{flags.data(), cflags.data(), ref_id.data(), len.data(), rg.data(), mq.data(), mflags.data(), mref.data(), mline.data(), ncig.data(), nlen.data(), coff.data(), noff.data(),
               apos.data(), aend.data(), mpos.data(), tlen.data(), etlen.data(), cig.data(), names.data(), totals, aux.data(), aoff.data(), alen.data(), seq.data(), qual.data(), soff.data(), &pool, seq_cap};
        rc = decode_slice(&P, &S, O);
        if (rc != ERR_UNSUPPORTED) break;
    }
    if (rc) return rc;
    (void)bases;
    // ---- columns ----
    std::vector<uint8_t> col[S_N];
    std::map<int32_t, std::pair<std::vector<uint8_t>, std::vector<uint8_t>>> tagcol;       // tag key -> (length column, value bytes)
    std::vector<std::string> td; std::map<std::string, int32_t> td_index;
    bool multi = false;
    for (size_t r = 1; r < n; r++) if (ref_id[r] != ref_id[0]) multi = true;
    int64_t start = n ? apos[0] : 0, end = n ? aend[0] : 0;
    for (size_t r = 0; r < n; r++) { if (apos[r] < start) start = apos[r]; if (aend[r] > end) end = aend[r]; }
    const int32_t slice_ref = multi ? -2 : (n ? ref_id[0] : -1);
    if (multi || slice_ref < 0) { start = 0; end = -1; }
    int64_t last = start;
    for (size_t r = 0; r < n; r++) {                                     // MAP over records (+ scans for where each record's items go)
        const bool unmapped = (flags[r] & BAM_FUNMAP) != 0;
        const uint8_t *sq = seq.data() + soff[r], *ql = qual.data() + soff[r];
        const int32_t L = len[r];
        bool has_qual = false; for (int32_t i = 0; i < L; i++) if (ql[i] != 255) has_qual = true;
        if (L == 0 && !unmapped && ncig[r]) return -3;                   // CIGAR without bases (CF_NO_SEQ): not covered
        itf8(col[S_BF], flags[r]);
        itf8(col[S_CF], CF_DETACHED | (has_qual ? CF_PRESERVE_QUAL : 0));
        if (multi) itf8(col[S_RI], ref_id[r]);
        itf8(col[S_RL], L);
        itf8(col[S_AP], (int32_t)(apos[r] - last)); last = apos[r];
        itf8(col[S_RG], rg[r]);
        col[S_RN].insert(col[S_RN].end(), names.data() + noff[r], names.data() + noff[r] + nlen[r]); col[S_RN].push_back(0);
        itf8(col[S_MF], ((flags[r] & BAM_FMREVERSE) ? CRAM_M_REVERSE : 0) | ((flags[r] & BAM_FMUNMAP) ? CRAM_M_UNMAP : 0));
        itf8(col[S_NS], mref[r]); itf8(col[S_NP], (int32_t)mpos[r]); itf8(col[S_TS], (int32_t)tlen[r]);
        {   // tags: the record's list of (tag, type) keys picks its dictionary line; every value goes to its tag's two blocks
            std::string line;
            const uint8_t *a = aux.data() + aoff[r], *ae = a + alen[r];
            while (a < ae) {
                if (ae - a < 3) return -1;
                const size_t vs = aux_value_size(a[2], a + 3, ae);
                if (!vs || (size_t)(ae - a - 3) < vs) return -1;
                line.append((const char *)a, 3);
                auto &tc = tagcol[(a[0] << 16) | (a[1] << 8) | a[2]];
                itf8(tc.first, (int32_t)vs); tc.second.insert(tc.second.end(), a + 3, a + 3 + vs);
                a += 3 + vs;
            }
            auto it = td_index.find(line);
            if (it == td_index.end()) { it = td_index.emplace(line, (int32_t)td.size()).first; td.push_back(line); }
            itf8(col[S_TL], it->second);
        }
        if (unmapped) { col[S_BA].insert(col[S_BA].end(), sq, sq + L); if (has_qual) col[S_QS].insert(col[S_QS].end(), ql, ql + L); continue; }
        // features from CIGAR + bases + reference
        const RefSpan *ref = nullptr;
        for (int32_t k = 0; k < S.nrefs; k++) if (S.refs[k].ref_id == ref_id[r] && apos[r] >= S.refs[k].start) ref = &S.refs[k];
        std::vector<uint8_t> fc; std::vector<int32_t> fp;
        int32_t sp = 1, prev = 0; int64_t rp = apos[r];                 // read position (1-based), reference position (1-based)
        auto feature = [&](uint8_t code) { fc.push_back(code); fp.push_back(sp - prev); prev = sp; };
        for (int32_t c = 0; c < ncig[r]; c++) {
            const uint32_t op = cig[coff[r] + c] & 15u; const int32_t ol = (int32_t)(cig[coff[r] + c] >> 4);
            switch (op) {
            case C_MATCH:
                for (int32_t i = 0; i < ol; i++, sp++, rp++) {
                    const uint8_t b = sq[sp - 1];
                    const bool in_ref = ref && rp >= ref->start && rp < ref->start + (int64_t)ref->len && rp <= ref->sq_len;
                    if (in_ref) {
                        const uint8_t rb = S.data[ref_off(ref) + (rp - ref->start)];
                        if (b == rb) continue;
                        const int l1 = rb == 'A' ? 0 : rb == 'C' ? 1 : rb == 'G' ? 2 : rb == 'T' ? 3 : 4;
                        const char *hit = (const char *)memchr(SM[l1], b, 4);
                        if (hit) { feature('X'); col[S_BS].push_back((uint8_t)(hit - SM[l1])); continue; }
                    }
                    feature('B'); col[S_BA].push_back(b); col[S_QS].push_back(ql[sp - 1]);      // a base the substitution code cannot express, or no reference
                }
                break;
            case C_INS: feature('I'); col[S_IN].insert(col[S_IN].end(), sq + sp - 1, sq + sp - 1 + ol); col[S_IN].push_back('\t'); sp += ol; break;
            case C_SOFT_CLIP: feature('S'); col[S_SC].insert(col[S_SC].end(), sq + sp - 1, sq + sp - 1 + ol); col[S_SC].push_back('\t'); sp += ol; break;
            case C_DEL: feature('D'); itf8(col[S_DL], ol); rp += ol; break;
            case C_REF_SKIP: feature('N'); itf8(col[S_RS], ol); rp += ol; break;
            case C_HARD_CLIP: feature('H'); itf8(col[S_HC], ol); break;
            case C_PAD: feature('P'); itf8(col[S_PD], ol); break;
            default: return -3;                                          // = / X ops: the decoder always answers M
            }
        }
        if (sp - 1 != L) return -1;
        itf8(col[S_FN], (int32_t)fc.size());
        for (size_t f = 0; f < fc.size(); f++) { col[S_FC].push_back(fc[f]); itf8(col[S_FP], fp[f]); }
        itf8(col[S_MQ], mq[r]);
        if (has_qual) col[S_QS].insert(col[S_QS].end(), ql, ql + L);
    }
    // a base that is a tab would end an IN / SC item early: not covered (the reference picks another stop byte)
    // ---- compression header ----
    std::vector<uint8_t> pres, encm, tagm, comp;
    {
        std::vector<uint8_t> body; itf8(body, 5);
        const uint8_t kv[][3] = {{'R', 'N', 1}, {'A', 'P', 1}, {'R', 'R', 1}};
        for (auto &k : kv) body.insert(body.end(), k, k + 3);
        body.push_back('S'); body.push_back('M'); for (int i = 0; i < 5; i++) body.push_back(0x1B);
        std::vector<uint8_t> tdb; for (auto &l : td) { tdb.insert(tdb.end(), l.begin(), l.end()); tdb.push_back(0); }
        if (td.empty()) tdb.push_back(0);
        body.push_back('T'); body.push_back('D'); app(body, sized(tdb));
        pres = sized(body);
    }
    auto cid_of = [](int s) { return 10 + s; };
    {
        std::vector<uint8_t> body; int cnt = 0;
        std::vector<uint8_t> ents;
        for (int s = 0; s < S_N; s++) {
            if (s == S_NF || s == S_BB || s == S_QQ) continue;
            if (s == S_RI && !multi) continue;
            ents.push_back((uint8_t)SERIES_NAME[s][0]); ents.push_back((uint8_t)SERIES_NAME[s][1]);
            app(ents, s == S_RN ? enc_stop(0, cid_of(s)) : (s == S_IN || s == S_SC) ? enc_stop('\t', cid_of(s)) : enc_external(cid_of(s)));
            cnt++;
        }
        itf8(body, cnt); app(body, ents); encm = sized(body);
    }
    std::vector<std::pair<int32_t, std::vector<uint8_t>>> blocks;
    {
        std::vector<uint8_t> body; itf8(body, (int32_t)tagcol.size());
        int32_t next = 100;
        for (auto &t : tagcol) { itf8(body, t.first); app(body, enc_len(next, next + 1)); blocks.emplace_back(next, t.second.first); blocks.emplace_back(next + 1, t.second.second); next += 2; }
        tagm = sized(body);
    }
    app(comp, pres); app(comp, encm); app(comp, tagm);
    for (int s = 0; s < S_N; s++) if (!col[s].empty()) blocks.emplace_back(cid_of(s), col[s]);
    // ---- slice header ----
    std::vector<uint8_t> sh;
    itf8(sh, slice_ref); itf8(sh, (int32_t)start); itf8(sh, (int32_t)(end >= start ? end - start + 1 : 0)); itf8(sh, (int32_t)n); sh.push_back(0);
    itf8(sh, (int32_t)blocks.size() + 1); itf8(sh, (int32_t)blocks.size());
    for (auto &b : blocks) itf8(sh, b.first);
    itf8(sh, -1); sh.insert(sh.end(), 16, 0);
    // ---- blob ----
    std::vector<uint8_t> o;
    auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); };
    put32((uint32_t)comp.size()); app(o, comp); put32((uint32_t)sh.size()); app(o, sh); put32((uint32_t)blocks.size());
    for (auto &b : blocks) { put32((uint32_t)b.first); put32((uint32_t)b.second.size()); app(o, b.second); }
    if (o.size() > cap) return -5;
    memcpy(out, o.data(), o.size());
    return (long)o.size();
}

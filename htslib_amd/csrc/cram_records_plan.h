// cram_records_plan.h -- host side of the CRAM record decoder: the compression header block of a container
// (cram_decode_compression_header, reference cram/cram_decode.c:144-950; tag dictionary cram_decode_TD :70-140) and the slice
// header block (cram_decode_slice_header, :954-1060) flattened into the tables cram_records_core.h walks.  CRAM 2.x / 3.x (ITF8).
// Host-only C++; included by cram_records.hip (product) and by the CPU test harness.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include "cram_records_core.h"

namespace hgr {

struct PlanHost {
    Plan plan;                                   // pointers are filled in by finish() (host) or by the uploader (device)
    std::vector<Codec> codecs;
    std::vector<HuffCode> huff;
    std::vector<int32_t> tl_off, tl_codec, tl_tag;
    std::vector<uint8_t> sm = std::vector<uint8_t>(20);
    std::map<int32_t, int32_t> slot_of;          // content id -> slot
    std::vector<int32_t> slot_id;                // slot -> content id
    int unsupported = 0;                         // a codec this build does not decode is present (reported when a slice uses the plan)
    int no_ref = 0;                              // preservation map RR = 0: the slices were written without a reference (cram_decode.c:250-262)
    void finish() { plan.sm = sm.data(); plan.tl_off = tl_off.data(); plan.tl_codec = tl_codec.data(); plan.tl_tag = tl_tag.data(); plan.codecs = codecs.data(); plan.huff = huff.data(); plan.nslots = (int32_t)slot_id.size(); }
};

struct Cursor {
    const uint8_t *p, *end; bool bad = false;
    int32_t itf8() {
        if (p >= end) { bad = true; return 0; }
        const uint32_t b0 = *p;
        const int extra = b0 < 0x80 ? 0 : b0 < 0xc0 ? 1 : b0 < 0xe0 ? 2 : b0 < 0xf0 ? 3 : 4;
        if (end - p <= extra) { bad = true; p = end; return 0; }
        uint32_t v;
        if (extra == 0) v = b0;
        else if (extra == 1) v = ((b0 & 0x3f) << 8) | p[1];
        else if (extra == 2) v = ((b0 & 0x1f) << 16) | (p[1] << 8) | p[2];
        else if (extra == 3) v = ((b0 & 0x0f) << 24) | (p[1] << 16) | (p[2] << 8) | p[3];
        else v = ((b0 & 0x0f) << 28) | (p[1] << 20) | (p[2] << 12) | (p[3] << 4) | (p[4] & 0x0f);
        p += 1 + extra;
        return (int32_t)v;
    }
    int64_t ltf8() {                              // ltf8_get (cram_io.c:395-470)
        if (p >= end) { bad = true; return 0; }
        const uint32_t b0 = *p;
        int extra = 0; while (extra < 8 && (b0 & (0x80u >> extra))) extra++;
        if (end - p <= extra) { bad = true; p = end; return 0; }
        uint64_t v = extra >= 8 ? 0 : (b0 & (0xffu >> (extra + 1)));
        if (extra == 8) v = 0;
        for (int i = 1; i <= extra; i++) v = (v << 8) | p[i];
        p += 1 + extra;
        return (int64_t)v;
    }
    int byte() { if (p >= end) { bad = true; return 0; } return *p++; }
};

inline int32_t plan_slot(PlanHost &H, int32_t content_id) {
    auto it = H.slot_of.find(content_id);
    if (it != H.slot_of.end()) return it->second;
    const int32_t s = (int32_t)H.slot_id.size();
    H.slot_of[content_id] = s; H.slot_id.push_back(content_id);
    return s;
}

// One encoding (id, parameter bytes) -> index into H.codecs, or -1 on a malformed description.  cram_codecs.c *_decode_init.
inline int32_t plan_codec(PlanHost &H, int32_t encoding, const uint8_t *par, int32_t size) {
    Cursor c{par, par + size};
    Codec C{encoding, 0, 0, 0};
    switch (encoding) {
    case E_EXTERNAL: C.a = plan_slot(H, c.itf8()); break;
    case E_HUFFMAN: {
        const int32_t n = c.itf8();
        if (c.bad || n < 0 || n > 65536) return -1;
        std::vector<HuffCode> codes((size_t)n);
        for (int32_t i = 0; i < n; i++) codes[(size_t)i].symbol = c.itf8();
        if (c.itf8() != n || c.bad) return -1;
        int32_t max_len = 0;
        for (int32_t i = 0; i < n; i++) { const int32_t l = c.itf8(); if (l < 0) return -1; codes[(size_t)i].len = l; max_len = std::max(max_len, l); }
        if (c.bad || c.p != c.end || (n && max_len >= n) || max_len > 31) return -1;        // cram_codecs.c:2906-2915
        std::stable_sort(codes.begin(), codes.end(), [](const HuffCode &x, const HuffCode &y) { return x.len != y.len ? x.len < y.len : x.symbol < y.symbol; });
        int64_t val = -1; int32_t last_len = 0; uint32_t max_val = 0;
        for (int32_t i = 0; i < n; i++) {                               // canonical codes (cram_codecs.c:2920-2932)
            val++;
            if ((uint64_t)val > max_val) return -1;
            if (codes[(size_t)i].len > last_len) { val <<= (codes[(size_t)i].len - last_len); last_len = codes[(size_t)i].len; max_val = (1u << last_len) - 1u; }
            codes[(size_t)i].code = (uint32_t)val; codes[(size_t)i].pad = 0;
        }
        C.a = (int32_t)H.huff.size(); C.b = n;
        H.huff.insert(H.huff.end(), codes.begin(), codes.end());
        break;
    }
    case E_BETA: C.a = c.itf8(); C.b = c.itf8(); if (C.b < 0 || C.b > 32) return -1; break;
    case E_GAMMA: C.a = c.itf8(); break;
    case E_SUBEXP: C.a = c.itf8(); C.b = c.itf8(); if (C.b < 0) return -1; break;
    case E_BYTE_ARRAY_LEN: {
        const int32_t e1 = c.itf8(), s1 = c.itf8();
        if (c.bad || s1 < 0 || c.end - c.p < s1) return -1;
        C.a = plan_codec(H, e1, c.p, s1); c.p += s1;
        const int32_t e2 = c.itf8(), s2 = c.itf8();
        if (c.bad || s2 < 0 || c.end - c.p < s2) return -1;
        C.b = plan_codec(H, e2, c.p, s2); c.p += s2;
        if (C.a < 0 || C.b < 0) return -1;
        break;
    }
    case E_BYTE_ARRAY_STOP: C.b = c.byte(); C.a = plan_slot(H, c.itf8()); break;
    default: H.unsupported = 1; break;                                  // GOLOMB, GOLOMB_RICE, CRAM 4 transforms: kept as a codec that fails when used
    }
    if (c.bad) return -1;
    H.codecs.push_back(C);
    return (int32_t)H.codecs.size() - 1;
}

// The compression header block (decoded) -> plan.  0 or -1.
inline int plan_from_compression_header(PlanHost &H, const uint8_t *b, size_t n) {
    static const char *names[S_N] = {"BF", "CF", "RI", "RL", "AP", "RG", "RN", "MF", "NS", "NP", "TS", "NF", "TL", "FN", "FC", "FP", "DL", "BA", "BS", "IN",
                                     "SC", "HC", "PD", "RS", "MQ", "QS", "BB", "QQ"};
    H = PlanHost();
    for (int i = 0; i < S_N; i++) H.plan.codec_of[i] = -1;
    H.plan.rn_included = 0; H.plan.ap_delta = 1; H.plan.qs_seq_orient = 1;   // defaults (cram_decode.c:203-207)
    memcpy(H.sm.data(), "CGTNAGTNACTNACGNACGT", 20);
    Cursor c{b, b + n};
    std::vector<std::vector<uint8_t>> td;                               // tag dictionary lines: 3-byte (tag, type) triples
    {   // preservation map
        const int32_t size = c.itf8(); const uint8_t *start = c.p; const int32_t count = c.itf8();
        if (c.bad || size < 0 || c.end - start < size) return -1;
        for (int32_t i = 0; i < count; i++) {
            if (c.end - c.p < 3) return -1;
            const uint8_t k0 = c.p[0], k1 = c.p[1]; c.p += 2;
            if (k0 == 'R' && k1 == 'N') H.plan.rn_included = c.byte();
            else if (k0 == 'A' && k1 == 'P') H.plan.ap_delta = c.byte();
            else if (k0 == 'Q' && k1 == 'O') H.plan.qs_seq_orient = c.byte();
            else if (k0 == 'R' && k1 == 'R') H.no_ref = !c.byte();
            else if (k0 == 'S' && k1 == 'M') {                          // cram_decode.c:290-318: code -> base, per reference base
                if (c.end - c.p < 5) return -1;
                static const char *others[5] = {"CGTN", "AGTN", "ACTN", "ACGN", "ACGT"};
                for (int r = 0; r < 5; r++) for (int k = 0; k < 4; k++) H.sm[(size_t)(4 * r + ((c.p[r] >> (6 - 2 * k)) & 3))] = (uint8_t)others[r][k];
                c.p += 5;
            }
            else if (k0 == 'T' && k1 == 'D') {
                const int32_t sz = c.itf8();
                if (c.bad || sz < 0 || c.end - c.p < sz) return -1;
                const uint8_t *q = c.p, *qe = c.p + sz;
                if (sz && qe[-1] != 0) return -1;                       // cram_decode_TD: the dictionary ends with a NUL
                while (q < qe) { const uint8_t *z = (const uint8_t *)memchr(q, 0, (size_t)(qe - q)); if ((z - q) % 3) return -1; td.emplace_back(q, z); q = z + 1; }
                c.p += sz;
            } else (void)c.byte();                                      // RR, MI, UI, PI and unknown keys: one byte
        }
        if (c.bad || c.p - start != size) return -1;
    }
    {   // data series encodings
        const int32_t size = c.itf8(); const uint8_t *start = c.p; const int32_t count = c.itf8();
        if (c.bad || size < 0 || c.end - start < size) return -1;
        for (int32_t i = 0; i < count; i++) {
            if (c.end - c.p < 4) return -1;
            const char k0 = (char)c.p[0], k1 = (char)c.p[1]; c.p += 2;
            const int32_t enc = c.itf8(), sz = c.itf8();
            if (c.bad || sz < 0 || c.end - c.p < sz) return -1;
            if (enc != E_NULL)
                for (int s = 0; s < S_N; s++)
                    if (names[s][0] == k0 && names[s][1] == k1 && H.plan.codec_of[s] < 0) { const int32_t ci = plan_codec(H, enc, c.p, sz); if (ci < 0) return -1; H.plan.codec_of[s] = ci; }
            c.p += sz;
        }
        if (c.p - start != size) return -1;
    }
    std::map<int32_t, int32_t> tag_codec;
    {   // tag encodings
        const int32_t size = c.itf8(); const uint8_t *start = c.p; const int32_t count = c.itf8();
        if (c.bad || size < 0 || c.end - start < size) return -1;
        for (int32_t i = 0; i < count; i++) {
            const int32_t key = c.itf8(), enc = c.itf8(), sz = c.itf8();
            if (c.bad || sz < 0 || c.end - c.p < sz) return -1;
            if (enc != E_NULL && !tag_codec.count(key)) { const int32_t ci = plan_codec(H, enc, c.p, sz); if (ci < 0) return -1; tag_codec[key] = ci; }
            c.p += sz;
        }
        if (c.p - start != size) return -1;
    }
    H.tl_off.push_back(0);
    for (const auto &line : td) {
        for (size_t t = 0; t + 2 < line.size(); t += 3) {
            const int32_t key = (line[t] << 16) | (line[t + 1] << 8) | line[t + 2];
            auto it = tag_codec.find(key);
            H.tl_codec.push_back(it == tag_codec.end() ? -1 : it->second);
            H.tl_tag.push_back(key);
        }
        H.tl_off.push_back((int32_t)H.tl_codec.size());
    }
    H.plan.nTL = (int32_t)td.size();
    H.finish();
    return 0;
}

struct SliceHeader { int32_t ref_seq_id = -1; int64_t ref_seq_start = 0, ref_seq_span = 0; int32_t nrec = 0, nblocks = 0; int64_t record_counter = 0;
                     int32_t ref_base_id = -1; uint8_t md5[16] = {0}; bool has_md5 = false; };
// The slice header block (content type 2 = mapped / multi-reference, 3 = unmapped in CRAM 1; v2+ always writes type 2).  0 or -1.
inline int parse_slice_header(const uint8_t *b, size_t n, int major, SliceHeader &h) {
    Cursor c{b, b + n};
    h.ref_seq_id = c.itf8(); h.ref_seq_start = c.itf8(); h.ref_seq_span = c.itf8();
    if (h.ref_seq_start < 0 || h.ref_seq_span < 0) return -1;
    h.nrec = c.itf8();
    h.record_counter = major >= 3 ? c.ltf8() : c.itf8();
    h.nblocks = c.itf8();
    if (c.bad || h.nrec < 0) return -1;
    // the rest is optional for the record decoder: content ids, the embedded-reference block id, the MD5 of the reference span (v2+)
    const int32_t nids = c.itf8();
    for (int32_t k = 0; k < nids && !c.bad; k++) (void)c.itf8();
    if (!c.bad) { const int32_t e = c.itf8(); if (!c.bad) h.ref_base_id = e; }
    if (!c.bad && major >= 2 && c.end - c.p >= 16) { memcpy(h.md5, c.p, 16); h.has_md5 = true; }
    return 0;
}


// ---- a batch of slices laid out for the decoder (shared by the device launcher and the CPU test harness) -------------------
struct PlanDev { int32_t codec_of[S_N]; int32_t rn_included, ap_delta, qs_seq_orient, nslots, nTL; uint32_t tl_off_base, tl_codec_base, codec_base, huff_base; uint8_t sm[5][4]; uint32_t ncodecs, nhuff; };
struct SliceDev {
    uint32_t plan, tab_off;               // tab: nslots offsets, nslots lengths, nslots cursors (words) in the block table
    uint32_t core_off, core_len;
    int32_t nrec, ref_seq_id;
    int64_t ref_seq_start;
    uint64_t rec_off, cig_off, name_off, aux_off;
    uint32_t cig_cap, name_cap, aux_cap, pad;
    uint32_t ref_first, nrefs;            // reference spans of the slice in Batch::refs
    int32_t decode_md, pad2;
    uint64_t job_off; uint32_t job_cap, pad3;   // deferred bulk copies (Batch::job_total entries in all)
    int64_t record_counter;               // slice header: number of the slice's first record in the file (names of name-less records, cram_decode.c:3113-3143)
};
struct Batch {
    std::vector<PlanDev> plans;
    std::vector<Codec> codecs; std::vector<HuffCode> huff; std::vector<int32_t> tl_off, tl_codec, tl_tag;
    std::vector<SliceDev> slices;
    std::vector<RefSpan> refs;
    std::vector<uint32_t> tab;
    std::vector<uint64_t> src_off;        // where each staged buffer goes in the data image, in the order of src_ptr / src_len
    std::vector<const uint8_t *> src_ptr; std::vector<uint32_t> src_len;
    uint64_t data_bytes = 0, nrec = 0, cig_total = 0, name_total = 0, aux_total = 0, job_total = 0;
    std::vector<int32_t> status;          // per slice: 0 = goes to the decoder, else the status already known
    std::vector<PlanHost> hosts;          // the parsed compression headers, parallel to plans (cram_records_fast_plan.h reads them)
};

// One input slice as the caller hands it over (mirrors hg_cram_slice_blocks / hg_cram_ref_span)
struct RefIn { int32_t ref_id; int64_t start; const uint8_t *bases; uint32_t len; int64_t sq_len; };
struct SliceIn { const uint8_t *comp_hdr; uint32_t comp_hdr_len; const uint8_t *slice_hdr; uint32_t slice_hdr_len; const uint8_t *core; uint32_t core_len;
                 uint32_t nblocks; const int32_t *content_id; const uint8_t *const *data; const uint32_t *len; uint32_t nrefs; const RefIn *refs; int32_t decode_md; };

inline int batch_build(Batch &B, const SliceIn *in, size_t n, int major) {
    B = Batch();
    if (major != 2 && major != 3) return -3;
    std::map<std::pair<const uint8_t *, uint32_t>, int32_t> seen;        // containers share a compression header: parse it once
    std::vector<PlanHost> &hosts = B.hosts;
    B.status.assign(n, 0);
    auto stage = [&](const uint8_t *p, uint32_t len) { const uint64_t off = B.data_bytes; B.src_off.push_back(off); B.src_ptr.push_back(p); B.src_len.push_back(len); B.data_bytes += ((uint64_t)len + 15u) & ~15ull; return off; };
    // reference spans go behind all blocks (64-bit offsets), each distinct buffer once: the slices of a chromosome share its bases
    struct PendingRef { size_t ref_index; const uint8_t *p; uint32_t len; };
    std::vector<PendingRef> pending;
    for (size_t i = 0; i < n; i++) {
        const SliceIn &s = in[i];
        SliceDev d; memset(&d, 0, sizeof d);
        d.rec_off = B.nrec;                                               // a slice that fails early holds no records
        SliceHeader sh;
        if (!s.comp_hdr || !s.slice_hdr || parse_slice_header(s.slice_hdr, s.slice_hdr_len, major, sh)) { B.status[i] = -1; B.slices.push_back(d); continue; }
        const auto key = std::make_pair(s.comp_hdr, s.comp_hdr_len);
        auto it = seen.find(key);
        int32_t pi;
        if (it == seen.end()) {
            PlanHost H;
            if (plan_from_compression_header(H, s.comp_hdr, s.comp_hdr_len)) pi = -1;
            else {
                pi = (int32_t)B.plans.size();
                PlanDev pd; memcpy(pd.codec_of, H.plan.codec_of, sizeof pd.codec_of);
                pd.rn_included = H.plan.rn_included; pd.ap_delta = H.plan.ap_delta; pd.qs_seq_orient = H.plan.qs_seq_orient; pd.nslots = H.plan.nslots; pd.nTL = H.plan.nTL;
                memcpy(pd.sm, H.sm.data(), 20); pd.ncodecs = (uint32_t)H.codecs.size(); pd.nhuff = (uint32_t)H.huff.size();
                pd.tl_off_base = (uint32_t)B.tl_off.size(); pd.tl_codec_base = (uint32_t)B.tl_codec.size(); pd.codec_base = (uint32_t)B.codecs.size(); pd.huff_base = (uint32_t)B.huff.size();
                B.tl_off.insert(B.tl_off.end(), H.tl_off.begin(), H.tl_off.end()); B.tl_codec.insert(B.tl_codec.end(), H.tl_codec.begin(), H.tl_codec.end()); B.tl_tag.insert(B.tl_tag.end(), H.tl_tag.begin(), H.tl_tag.end());
                B.codecs.insert(B.codecs.end(), H.codecs.begin(), H.codecs.end()); B.huff.insert(B.huff.end(), H.huff.begin(), H.huff.end());
                B.plans.push_back(pd); hosts.push_back(std::move(H));
            }
            seen[key] = pi;
        } else pi = it->second;
        if (pi < 0) { B.status[i] = -1; B.slices.push_back(d); continue; }
        const PlanHost &H = hosts[(size_t)pi];
        d.plan = (uint32_t)pi; d.nrec = sh.nrec; d.ref_seq_id = sh.ref_seq_id; d.ref_seq_start = sh.ref_seq_start; d.record_counter = sh.record_counter;
        d.tab_off = (uint32_t)B.tab.size();
        const size_t ns = H.slot_id.size();
        B.tab.resize(B.tab.size() + 3 * ns, 0u);
        for (size_t k = 0; k < ns; k++) B.tab[d.tab_off + ns + k] = 0xffffffffu;      // absent until a block with that content id shows up
        uint64_t ext_bytes = 0;
        for (uint32_t k = 0; k < s.nblocks; k++) {
            auto sl = H.slot_of.find(s.content_id[k]);
            if (sl == H.slot_of.end()) continue;                       // a block no codec reads
            const uint64_t off = stage(s.data[k], s.len[k]);
            B.tab[d.tab_off + (size_t)sl->second] = (uint32_t)off; B.tab[d.tab_off + ns + (size_t)sl->second] = s.len[k];
            ext_bytes += s.len[k];
        }
        d.core_off = (uint32_t)stage(s.core, s.core ? s.core_len : 0); d.core_len = s.core ? s.core_len : 0;
        d.ref_first = (uint32_t)B.refs.size(); d.nrefs = s.refs ? s.nrefs : 0; d.decode_md = s.decode_md;
        for (uint32_t k = 0; k < d.nrefs; k++) {
            const RefIn &r = s.refs[k];
            pending.push_back(PendingRef{B.refs.size(), r.bases, r.len});
            B.refs.push_back(RefSpan{r.ref_id, 0u, r.len, 0u, r.start, r.sq_len});
        }
        // a header that claims more records than its blocks could possibly describe (16 per byte) is damage, not data: it must not turn
        // into a minutes-long walk over constant codecs or into gigabytes of columns
        if ((uint64_t)sh.nrec > 16ull * (ext_bytes + d.core_len) + 1024ull) { B.status[i] = -3; d.nrec = 0; B.slices.push_back(d); continue; }
        d.rec_off = B.nrec; B.nrec += (uint64_t)sh.nrec;
        // capacities: a read name is copied out of a block, a CIGAR op needs a feature; features that cost no bits at all (constant
        // codecs) are bounded by 4 ops per record on top of one op per byte of the slice
        // ... and, where the compression header says WHICH blocks feed an array (EXTERNAL / BYTE_ARRAY_* codecs over blocks of the slice: what every
        // writer emits for names, feature codes and tag values), by those blocks alone: names are copied out of the RN codec's blocks, a feature costs
        // one byte of the FC block (and makes at most two CIGAR operations: the match run before it and its own), a tag value costs at least one byte of
        // its codec's blocks.  (16 x cig_cap is also the decoder's run-away budget -- features + tag values walked per slice -- so a sixteenth of the slice's
        // bytes stays in it: every feature and every tag value costs at least one byte.)  The generic bounds (every block of the slice could be names) made the arrays of a 10 000-read slice 17 MB -- 16 GB for the
        // 1 000-slice batches of the whole-slice reader.  -1 = the codec reads the CORE block or is a constant: the generic bound stays.
        auto slot_len = [&](int32_t slot) -> int64_t {
            if (slot < 0 || (size_t)slot >= ns) return -1;
            const uint32_t l = B.tab[d.tab_off + ns + (size_t)slot];
            return l == 0xffffffffu ? 0 : (int64_t)l;
        };
        auto codec_bytes = [&](int32_t ci) -> int64_t {
            if (ci < 0 || (size_t)ci >= H.codecs.size()) return -1;
            const Codec &C = H.codecs[(size_t)ci];
            if (C.kind == E_EXTERNAL || C.kind == E_BYTE_ARRAY_STOP) return slot_len(C.a);
            if (C.kind == E_BYTE_ARRAY_LEN && C.a >= 0 && C.b >= 0 && (size_t)C.a < H.codecs.size() && (size_t)C.b < H.codecs.size()) {
                const Codec &V = H.codecs[(size_t)C.b];
                return V.kind == E_EXTERNAL ? slot_len(V.a) : -1;
            }
            return -1;
        };
        const uint64_t generic_names = ext_bytes + (uint64_t)d.core_len + 16u, generic_cig = 4ull * (uint64_t)sh.nrec + ext_bytes + 8ull * d.core_len + 16u;
        const int64_t rn_bytes = H.plan.codec_of[S_RN] < 0 ? 0 : codec_bytes(H.plan.codec_of[S_RN]);
        const int64_t fc_bytes = H.plan.codec_of[S_FC] < 0 ? 0 : (H.codecs[(size_t)H.plan.codec_of[S_FC]].kind == E_EXTERNAL ? codec_bytes(H.plan.codec_of[S_FC]) : -1);
        d.name_cap = (uint32_t)std::min<uint64_t>(rn_bytes >= 0 ? std::min<uint64_t>((uint64_t)rn_bytes + 16u, generic_names) : generic_names, 0xffffffffull);
        d.cig_cap = (uint32_t)std::min<uint64_t>(fc_bytes >= 0 ? std::min<uint64_t>(2ull * (uint64_t)fc_bytes + 4ull * (uint64_t)sh.nrec + 16u + (ext_bytes + 8ull * d.core_len) / 16u, generic_cig) : generic_cig, 0xffffffffull);
        d.cig_off = B.cig_total; B.cig_total += d.cig_cap;
        d.name_off = B.name_total; B.name_total += d.name_cap;
        // aux: the values are copied out of blocks, 3 bytes of tag + type are added per value; values that cost no bits bounded as above
        uint64_t tag_bytes = 0; bool tags_known = true;
        {
            std::vector<char> seen_slot(ns, 0);
            auto add_slot = [&](int32_t slot) { if (slot < 0 || (size_t)slot >= ns) { tags_known = false; return; } if (!seen_slot[(size_t)slot]) { seen_slot[(size_t)slot] = 1; tag_bytes += (uint64_t)slot_len(slot); } };
            for (size_t t = 0; t < H.tl_codec.size() && tags_known; t++) {
                const int32_t ci = H.tl_codec[t];
                if (ci < 0) continue;
                if ((size_t)ci >= H.codecs.size()) { tags_known = false; break; }
                const Codec &C = H.codecs[(size_t)ci];
                if (C.kind == E_EXTERNAL || C.kind == E_BYTE_ARRAY_STOP) add_slot(C.a);
                else if (C.kind == E_BYTE_ARRAY_LEN && C.a >= 0 && C.b >= 0 && (size_t)C.a < H.codecs.size() && (size_t)C.b < H.codecs.size()) {
                    const Codec &L = H.codecs[(size_t)C.a], &V = H.codecs[(size_t)C.b];
                    if (V.kind != E_EXTERNAL || (L.kind != E_EXTERNAL && !(L.kind == E_HUFFMAN && L.b == 1))) tags_known = false;
                    else { add_slot(V.a); if (L.kind == E_EXTERNAL) add_slot(L.a); else if (H.huff[(size_t)L.a].symbol <= 0) tags_known = false; }   // a constant length of 0 costs no byte at all
                } else tags_known = false;
            }
        }
        const uint64_t generic_aux = 4ull * (ext_bytes + d.core_len) + 64ull * (uint64_t)sh.nrec + 64u;
        // regenerated MD:Z / NM (decode_md with a reference): a mismatch costs a byte of the FC block and a few
        // characters of MD; a long deletion (few bytes in, its whole length out) can exceed this -- the slice then fails "no room" and the caller's fall-back takes it
        const uint64_t md_room = !(s.decode_md != 0 && s.refs && s.nrefs) ? 0ull
                               : fc_bytes >= 0 ? 8ull * (uint64_t)fc_bytes + 32ull * (uint64_t)sh.nrec          // per feature: a run length and a base; per record: "MD:Z", the last run, NM
                                               : 2ull * ext_bytes + 8ull * d.core_len;
        d.aux_cap = (uint32_t)std::min<uint64_t>(tags_known ? std::min<uint64_t>(4ull * tag_bytes + 64ull * (uint64_t)sh.nrec + 64u + md_room, generic_aux) : generic_aux, 0xffffffffull);
        d.aux_off = B.aux_total; B.aux_total += d.aux_cap;
        d.job_cap = (uint32_t)std::min<uint64_t>(4ull * (uint64_t)sh.nrec + 64u, 0x7fffffffull);     // a few per record; when the room runs out the copy is made at once
        d.job_off = B.job_total; B.job_total += d.job_cap;
        B.slices.push_back(d);
    }
    if (B.data_bytes > 0xfffffff0ull) return -4;                       // 32-bit offsets of the blocks in the data image: the caller splits the batch
    {
        std::map<std::pair<const uint8_t *, uint32_t>, uint64_t> placed;
        for (const PendingRef &q : pending) {
            const auto key = std::make_pair(q.p, q.len);
            auto it = placed.find(key);
            uint64_t off;
            if (it == placed.end()) { off = stage(q.p, q.len); placed[key] = off; } else off = it->second;
            B.refs[q.ref_index].off = (uint32_t)off; B.refs[q.ref_index].off_hi = (uint32_t)(off >> 32);
        }
    }
    return 0;
}

}  // namespace hgr

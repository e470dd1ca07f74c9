// cram_records_fast_plan.h -- host side of the data-parallel record decoder (cram_records_fast.h): which compression headers qualify, and
// the per-slice tables its passes read.  Host-only C++; shared by the device launcher (cram_records.hip) and the CPU test harness.
//
// A compression header (cram_decode_compression_header, reference cram/cram_decode.c:144-950) qualifies when every series and every tag
// value is EXTERNAL in a block of its own, a zero-bit HUFFMAN constant, a BYTE_ARRAY_STOP over such a block, or a BYTE_ARRAY_LEN whose length
// is a constant or EXTERNAL and whose bytes are EXTERNAL, all in blocks of their own -- cram_encode_compression_header's choices for
// position-sorted data (cram/cram_encode.c:2150-2450, tags :2900-3040).  What does not qualify (bits in the CORE block: htsjdk's HUFFMAN /
// BETA / GAMMA series, htslib's BETA-coded AP of unsorted data; two series in one block: 'B' array tags, htsjdk's shared blocks) keeps
// the chain decoder.
#pragma once
#include <vector>
#include "cram_records_plan.h"
#include "cram_records_fast.h"

namespace hgr {

struct FSerProto { int32_t kind = F_ABSENT, k = 0, slot = -1, len_slot = -1; };
struct FastPlanHost {
    bool eligible = false;
    std::vector<FSerProto> ser;           // S_N series, then the distinct tag codecs
    std::vector<int32_t> tl_tagidx;       // parallel to PlanHost::tl_codec
    uint32_t ntag = 0;
};

inline bool fast_is_const(const PlanHost &H, const Codec &C) { return C.kind == E_HUFFMAN && C.b == 1 && H.huff[(size_t)C.a].len == 0; }

inline void fast_plan(const PlanHost &H, FastPlanHost &F) {
    F = FastPlanHost();
    if (H.unsupported) return;
    static const int ints[] = {S_BF, S_CF, S_RI, S_RL, S_AP, S_RG, S_MF, S_NS, S_NP, S_TS, S_NF, S_TL, S_FN, S_FP, S_DL, S_HC, S_PD, S_RS, S_MQ};
    static const int bytes_[] = {S_FC, S_BS, S_BA, S_QS};
    static const int arrays[] = {S_RN, S_IN, S_SC, S_BB, S_QQ};
    std::vector<int> users(H.slot_id.size(), 0);
    F.ser.assign(S_N, FSerProto());
    auto scalar = [&](int s, bool as_byte) -> bool {
        const int32_t ci = H.plan.codec_of[s];
        if (ci < 0) return true;
        const Codec &C = H.codecs[(size_t)ci];
        FSerProto &e = F.ser[(size_t)s];
        if (C.kind == E_HUFFMAN && C.b <= 0) return true;                   // no symbols: an error when read, like the chain decoder's
        if (fast_is_const(H, C)) { e.kind = F_CONST; e.k = H.huff[(size_t)C.a].symbol; return true; }
        if (C.kind != E_EXTERNAL) return false;
        e.kind = as_byte ? F_BYTES : F_INT; e.slot = C.a; users[(size_t)C.a]++;
        return true;
    };
    auto array = [&](int32_t ci, FSerProto &e) -> bool {
        const Codec &C = H.codecs[(size_t)ci];
        if (C.kind == E_BYTE_ARRAY_STOP) { e.kind = F_STOP; e.k = C.b; e.slot = C.a; users[(size_t)C.a]++; return true; }
        if (C.kind == E_BYTE_ARRAY_LEN) {
            if (C.a < 0 || C.b < 0) return false;
            const Codec &L = H.codecs[(size_t)C.a], &V = H.codecs[(size_t)C.b];
            if (V.kind != E_EXTERNAL) return false;
            e.slot = V.a; users[(size_t)V.a]++;
            if (fast_is_const(H, L)) { e.kind = F_LENC; e.k = H.huff[(size_t)L.a].symbol; return e.k >= 0; }
            if (L.kind != E_EXTERNAL) return false;
            e.kind = F_LENV; e.len_slot = L.a; users[(size_t)L.a]++;
            return true;
        }
        return false;
    };
    for (int s : ints) if (!scalar(s, false)) return;
    for (int s : bytes_) if (!scalar(s, true)) return;
    for (int s : arrays) { const int32_t ci = H.plan.codec_of[s]; if (ci >= 0 && !array(ci, F.ser[(size_t)s])) return; }
    // tags: one entry per distinct codec of the tag encoding map that a dictionary line uses
    std::vector<int32_t> codec_of_tag;
    F.tl_tagidx.assign(H.tl_codec.size(), -1);
    for (size_t t = 0; t < H.tl_codec.size(); t++) {
        const int32_t ci = H.tl_codec[t];
        if (ci < 0) continue;                                               // a tag without an encoding: an error for the records that carry it
        size_t k = 0;
        while (k < codec_of_tag.size() && codec_of_tag[k] != ci) k++;
        if (k == codec_of_tag.size()) {
            if (k >= (size_t)FAST_MAX_TAGS) return;
            FSerProto e;
            const Codec &C = H.codecs[(size_t)ci];
            if (C.kind == E_EXTERNAL) { e.kind = F_LENC; e.k = 1; e.slot = C.a; users[(size_t)C.a]++; }      // a one-byte value (cram_decode_aux: out_sz = 1)
            else if (!array(ci, e)) return;
            codec_of_tag.push_back(ci); F.ser.push_back(e);
        }
        F.tl_tagidx[t] = (int32_t)k;
    }
    F.ntag = (uint32_t)codec_of_tag.size();
    for (size_t l = 0; l + 1 < H.tl_off.size(); l++)                        // a line naming one tag twice reads two values of it: chain decoder
        for (int32_t a = H.tl_off[l]; a < H.tl_off[l + 1]; a++)
            for (int32_t b = a + 1; b < H.tl_off[l + 1]; b++)
                if (F.tl_tagidx[(size_t)a] >= 0 && F.tl_tagidx[(size_t)a] == F.tl_tagidx[(size_t)b]) return;
    for (int u : users) if (u > 1) return;                                  // two readers of one block interleave record by record
    F.eligible = true;
}

// One column the device decodes before the passes run
struct FastCol { uint64_t in_off; uint32_t in_len, stop; uint64_t pool_off; uint32_t cap, slice, src; };
struct FastBatch {
    std::vector<uint8_t> is_fast;                 // per slice
    std::vector<uint32_t> ser_off, ntag;          // per slice: its FSer entries (S_N + ntag)
    std::vector<FSer> ser;
    std::vector<int32_t> tl_tagidx;               // parallel to Batch::tl_codec
    std::vector<FastCol> itf8, stop, sums;        // columns, numbered itf8 first, then stop tables, then running sums (src = the itf8 column summed)
    uint64_t pool_words = 0;
    uint32_t ntag_max = 0;
    std::vector<uint32_t> fast_list;              // the fast slices
    std::vector<uint32_t> chunk_slice, chunk_r0;  // work units of the per-record passes: CHUNK records of one slice each
    enum { CHUNK = 256 };
    size_t ncols() const { return itf8.size() + stop.size() + sums.size(); }
};

// Batch::hosts must still hold the parsed headers (batch_build keeps them)
inline void fast_build(const Batch &B, FastBatch &F, bool enable) {
    F = FastBatch();
    const size_t n = B.slices.size();
    F.is_fast.assign(n, 0); F.ser_off.assign(n, 0xffffffffu); F.ntag.assign(n, 0);
    F.tl_tagidx.assign(B.tl_codec.size(), -1);
    if (!enable) return;
    std::vector<FastPlanHost> fp(B.hosts.size());
    for (size_t p = 0; p < B.hosts.size(); p++) {
        fast_plan(B.hosts[p], fp[p]);
        if (fp[p].eligible) for (size_t t = 0; t < fp[p].tl_tagidx.size(); t++) F.tl_tagidx[B.plans[p].tl_codec_base + t] = fp[p].tl_tagidx[t];
    }
    struct Fix { size_t ser; int which; };        // FSer::col / col2 hold list positions until the lists are complete
    std::vector<Fix> fix_stop, fix_sum;
    for (size_t i = 0; i < n; i++) {
        if (B.status[i] != 0) continue;
        const SliceDev &d = B.slices[i];
        const FastPlanHost &P = fp[d.plan];
        if (!P.eligible || d.nrec <= 0) continue;
        const size_t ns = (size_t)B.plans[d.plan].nslots;
        F.is_fast[i] = 1; F.ser_off[i] = (uint32_t)F.ser.size(); F.ntag[i] = P.ntag;
        if (P.ntag > F.ntag_max) F.ntag_max = P.ntag;
        F.fast_list.push_back((uint32_t)i);
        for (int32_t r0 = 0; r0 < d.nrec; r0 += FastBatch::CHUNK) { F.chunk_slice.push_back((uint32_t)i); F.chunk_r0.push_back((uint32_t)r0); }
        auto blk = [&](int32_t slot, uint32_t &off, uint32_t &len) {
            off = B.tab[d.tab_off + (size_t)slot]; len = B.tab[d.tab_off + ns + (size_t)slot];
            if (len == 0xffffffffu) { off = 0; len = 0; }                   // block absent: nothing to read from it
        };
        for (const FSerProto &q : P.ser) {
            FSer e{q.kind, q.k, 0u, 0u, 0u, 0u};
            if (q.slot >= 0) blk(q.slot, e.off, e.len);
            if (q.kind == F_INT) {
                e.col = (uint32_t)F.itf8.size();
                F.itf8.push_back(FastCol{e.off, e.len, 0u, F.pool_words, e.len, (uint32_t)i, 0u});
                F.pool_words += ((uint64_t)e.len + 3u) & ~3ull;
                e.off = e.len = 0;
            } else if (q.kind == F_STOP) {
                e.col = (uint32_t)F.stop.size(); fix_stop.push_back(Fix{F.ser.size(), 0});
                F.stop.push_back(FastCol{e.off, e.len, (uint32_t)(q.k & 0xff), F.pool_words, e.len + 1u, (uint32_t)i, 0u});
                F.pool_words += ((uint64_t)e.len + 4u) & ~3ull;
            } else if (q.kind == F_LENV) {
                uint32_t lo = 0, ll = 0; blk(q.len_slot, lo, ll);
                e.col = (uint32_t)F.itf8.size();
                F.itf8.push_back(FastCol{lo, ll, 0u, F.pool_words, ll, (uint32_t)i, 0u});
                F.pool_words += ((uint64_t)ll + 3u) & ~3ull;
                e.col2 = (uint32_t)F.sums.size(); fix_sum.push_back(Fix{F.ser.size(), 1});
                F.sums.push_back(FastCol{0u, 0u, 0u, F.pool_words, ll + 1u, (uint32_t)i, e.col});
                F.pool_words += ((uint64_t)ll + 4u) & ~3ull;
            }
            F.ser.push_back(e);
        }
    }
    for (const Fix &f : fix_stop) F.ser[f.ser].col += (uint32_t)F.itf8.size();
    for (const Fix &f : fix_sum) F.ser[f.ser].col2 += (uint32_t)(F.itf8.size() + F.stop.size());
}

}  // namespace hgr

// cram_records_core.h -- the record loop of cram_decode_slice (reference cram/cram_decode.c:2346-3026) for CRAM 2.x / 3.x slices,
// every data series is CONSUMED exactly as the reference consumes it (so that series sharing an EXTERNAL block or the CORE bit stream
// stay in step), and the cram_record fields are produced -- flags, reference id, position, read length, read group, mapping quality,
// read name, mate fields, CIGAR and alignment end (cram_decode_seq's feature walk, cram_decode.c:1096-1900), the bases (reference span
// + edits through the substitution matrix) and qualities when the caller supplies the reference spans, the aux tags as stored
// (cram_decode_aux, :2008-2137), and, after the mate cross-referencing pass (cram_decode_slice_xref, :2140-2307), mate position /
// reference, template length and the mate bits of the flags; MD:Z / NM are regenerated like the reference's decode_md option does.
//
// Codecs (cram/cram_codecs.c): EXTERNAL (:350-410; ITF8 for integer series, bytes for byte series), HUFFMAN (canonical codes,
// :2641-2930), BETA (:1072-1130), GAMMA (:2546-2568), SUBEXP (:2452-2494), BYTE_ARRAY_LEN (:2937-3010), BYTE_ARRAY_STOP (:3180-3260).
// GOLOMB / GOLOMB_RICE (never written by htslib or htsjdk) and the CRAM 4 transforms are not handled: the slice reports -3.
//
// One slice is one serial chain -- a record's bits start where the previous record's end -- so the unit of parallelism is the
// slice.  The code is written once for host and device: the gfx950 kernel (cram_records.hip) runs it with one wavefront per slice,
// the test harness (tests/native/cram_records_host.cpp) compiles the same source for the CPU to check it against the reference's
// SAM twins without a GPU.  It is NOT a CPU fallback of the product: libhtsgpu exports only the device path.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define HGR_LDS __attribute__((address_space(3)))      /* the read-ahead windows live in LDS on the device: say so in the type, a generic pointer would make every access a FLAT one */
#else
#define HGR_LDS
#endif
#if defined(__HIPCC__)
#define HGR_FN __host__ __device__ __forceinline__      /* by-reference counters must stay in registers: an out-of-line call puts them in scratch */
#else
#define HGR_FN inline
#endif

namespace hgr {

enum { E_NULL = 0, E_EXTERNAL = 1, E_GOLOMB = 2, E_HUFFMAN = 3, E_BYTE_ARRAY_LEN = 4, E_BYTE_ARRAY_STOP = 5, E_BETA = 6, E_SUBEXP = 7,
       E_GOLOMB_RICE = 8, E_GAMMA = 9 };
// kind / a / b / c:  EXTERNAL: a = block slot.  HUFFMAN: a = first code in the code table, b = number of codes.  BETA: a = offset,
// b = bits.  GAMMA: a = offset.  SUBEXP: a = offset, b = k.  BYTE_ARRAY_LEN: a = codec of the length, b = codec of the bytes.
// BYTE_ARRAY_STOP: a = block slot, b = stop byte.
struct Codec { int32_t kind, a, b, c; };
struct HuffCode { int32_t symbol, len; uint32_t code; int32_t pad; };            // sorted by (len, symbol), canonical codes assigned
enum Series { S_BF, S_CF, S_RI, S_RL, S_AP, S_RG, S_RN, S_MF, S_NS, S_NP, S_TS, S_NF, S_TL, S_FN, S_FC, S_FP, S_DL, S_BA, S_BS, S_IN, S_SC,
              S_HC, S_PD, S_RS, S_MQ, S_QS, S_BB, S_QQ, S_N };

// What the compression header says (cram_decode_compression_header, cram_decode.c:144-950), flattened by the host
struct Plan {
    int32_t codec_of[S_N];                // index into codecs[] or -1 (series absent from the encoding map)
    int32_t rn_included, ap_delta, qs_seq_orient, nslots;
    const uint8_t *sm;                    // substitution matrix (preservation map SM), 5 x 4: sm[4 * (reference base A C G T N) + BS code]
    int32_t nTL;                          // tag dictionary lines; line t holds tags tl_off[t] .. tl_off[t+1]-1 of tl_codec[]
    const int32_t *tl_off;
    const int32_t *tl_codec;              // codec index of the tag's encoding (tag encoding map), -1 = not in the map
    const int32_t *tl_tag;                // tag[0] << 16 | tag[1] << 8 | type, parallel to tl_codec
    const Codec *codecs;
    const HuffCode *huff;
};
// A bulk copy taken off the chain: bases copied from the reference, the qualities of a record, the bases of an unmapped read, long
// tag values.  None of them is read again while the slice is decoded, so the walk only notes them; they are carried out afterwards --
// one job per lane on the device.
struct CopyJob { uint8_t *dst; const uint8_t *src; uint32_t n, pad; };

// A stretch of reference bases the caller supplies for a slice (upper case ASCII; an embedded-reference block is one of these)
struct RefSpan { int32_t ref_id; uint32_t off, len, off_hi; int64_t start, sq_len; };   // off_hi:off into Slice::data (reference spans may lie beyond 4 GiB: a genome staged once per batch); start = 1-based position of the first base; sq_len = @SQ LN
HGR_FN uint64_t ref_off(const RefSpan *r) { return (uint64_t)r->off_hi << 32 | r->off; }
// One slice: its blocks by slot (offset / length into `data`; length 0xffffffff = block absent), the CORE block, scratch cursors
struct Slice {
    const uint8_t *data;
    const uint32_t *blk_off, *blk_len;    // nslots each
    uint32_t *cursor;                     // nslots words of scratch, zeroed by the decoder
    uint32_t core_off, core_len;
    int32_t nrec, ref_seq_id;             // slice header: -2 = multi-reference slice (RI is read), -1 = unmapped
    int64_t ref_seq_start;
    int32_t nref;                         // number of @SQ lines (bounds of RI / NS)
    uint32_t cigar_cap, name_cap, aux_cap;
    CopyJob *jobs; uint32_t job_cap;      // room for deferred bulk copies (nullptr: copy while walking); the count comes back in totals[3]
    HGR_LDS uint8_t *wbuf; HGR_LDS uint32_t *wpos;   // read-ahead windows: 128 * (nslots + 2) bytes and nslots + 2 words (nullptr: read the blocks directly)
    const RefSpan *refs; int32_t nrefs;   // reference spans of this slice (none: bases come out as '=' plus the stored edits)
    int32_t decode_md;                    // fd->decode_md: non-zero = MD:Z / NM are generated for mapped records that do not store them (hts_open's default is -1)
};
// Per-record results (arrays of nrec), the CIGAR ops and the read names of the slice
struct Cols {
    int32_t *flags, *cram_flags, *ref_id, *len, *rg, *mqual, *mate_flags, *mate_ref_id, *mate_line, *ncigar, *name_len;
    uint32_t *cigar_off, *name_off;
    int64_t *apos, *aend, *mate_pos, *tlen, *explicit_tlen;
    uint32_t *cigar;                      // (len << 4 | op), BAM encoding
    uint8_t *names;
    uint32_t *totals;                     // [0] = cigar ops written, [1] = name bytes written, [2] = aux bytes written, [3] = copy jobs noted
    uint8_t *aux; uint32_t *aux_off; int32_t *aux_len;    // aux == nullptr: not wanted.  BAM encoding: tag[2] type value, back to back
    // bases and qualities (seq == nullptr: not wanted): len bytes each per record at seq_off[rec], handed out from one pool
    uint8_t *seq, *qual; uint64_t *seq_off; unsigned long long *seq_pool; uint64_t seq_cap;
};
enum { ERR_MALFORMED = -1, ERR_UNSUPPORTED = -3,
       ERR_POOL = -7 };      // device launcher only: no room left in the shared pool of bases -- the slice is decoded again once the others have been moved out
enum { BAM_FPAIRED = 1, BAM_FUNMAP = 4, BAM_FMUNMAP = 8, BAM_FREVERSE = 16, BAM_FMREVERSE = 32, BAM_FREAD1 = 64 };
enum { CF_PRESERVE_QUAL = 1, CF_DETACHED = 2, CF_MATE_DOWNSTREAM = 4, CF_NO_SEQ = 8, CF_EXPLICIT_TLEN = 16 };
enum { CRAM_M_REVERSE = 1, CRAM_M_UNMAP = 2 };
enum { C_MATCH = 0, C_INS = 1, C_DEL = 2, C_REF_SKIP = 3, C_SOFT_CLIP = 4, C_HARD_CLIP = 5, C_PAD = 6 };
constexpr int64_t TLEN_UNSET = INT64_MIN;

// Byte copies in groups of 16 independent loads followed by 16 stores: source and destination may alias as far as the compiler
// knows, so a plain byte loop pays one global round trip PER BYTE on the device (measured: ~70 us per 150-base record).
HGR_FN void copy_bytes(uint8_t *dst, const uint8_t *src, uint32_t n) {
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) {
        uint8_t t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = src[i + k];
#pragma unroll
        for (int k = 0; k < 16; k++) dst[i + k] = t[k];
    }
    for (; i < n; i++) dst[i] = src[i];
}

struct Reader {
    const Plan *P; const Slice *S;
    uint64_t bit;                         // position in the CORE block, MSB first
    uint64_t work;                        // features walked so far: a damaged count with zero-bit codecs must not spin for minutes
    int err;
    CopyJob *jobs; uint32_t njobs, job_cap;   // jobs == nullptr: copy at once
    uint32_t rec_job0;                    // first deferred job of the record being decoded
    const uint8_t *pend_lo[2], *pend_hi[2];   // what the record's deferred jobs cover in seq[] ([0]) and qual[] ([1]): a later direct access there runs them first
    // Read-ahead windows (device, wave mapping: in LDS): WIN bytes of every block -- slot s at wbuf + WIN * s, the CORE block after the
    // last slot -- so that the values of a series cost one global round trip per WIN bytes, not one per value.  wbuf == nullptr: none.
    HGR_LDS uint8_t *wbuf; HGR_LDS uint32_t *wpos;
    enum { WIN = 128 };

    // bytes [c, c + need) of the block that starts at data + off (len bytes, staged on a 16-byte boundary with >= 16 readable bytes of
    // slack): a pointer INTO THE WINDOW to them, at least `need` (<= 16) valid.  Only called when wbuf != nullptr -- callers keep the
    // window pointer and the global pointer apart so that neither becomes a generic (FLAT) access on the device.
    HGR_FN const HGR_LDS uint8_t *win(uint32_t w, uint32_t off, uint32_t len, uint32_t c, uint32_t need) {
        uint32_t st = wpos[w];
        if (c < st || c + need > st + (uint32_t)WIN) {
            st = c & ~3u;
            const uint32_t lim = (len + 15u) & ~15u;                       // the staged image is readable up to here
            const uint32_t *src = (const uint32_t *)(S->data + off + st);
            HGR_LDS uint32_t *dst = (HGR_LDS uint32_t *)(wbuf + (size_t)WIN * w);
            uint32_t t[WIN / 4];
#pragma unroll
            for (int k = 0; k < WIN / 4; k++) t[k] = st + 4u * (uint32_t)k < lim ? src[k] : 0u;   // all loads first: one round trip
#pragma unroll
            for (int k = 0; k < WIN / 4; k++) dst[k] = t[k];
            wpos[w] = st;
        }
        return wbuf + (size_t)WIN * w + (c - st);
    }

    // which: 0 = into seq[], 1 = into qual[], 2 = elsewhere (aux values: nothing writes there again)
    HGR_FN void bulk(uint8_t *dst, const uint8_t *src, uint32_t n, int which = 2) {
        if (jobs && n >= 32u && njobs < job_cap) {
            jobs[njobs].dst = dst; jobs[njobs].src = src; jobs[njobs].n = n; jobs[njobs].pad = 0; njobs++;
            if (which < 2) { if (!pend_lo[which] || dst < pend_lo[which]) pend_lo[which] = dst; if (!pend_hi[which] || dst + n > pend_hi[which]) pend_hi[which] = dst + n; }
        } else copy_bytes(dst, src, n);
    }
    HGR_FN void begin_record() { rec_job0 = njobs; pend_lo[0] = pend_lo[1] = pend_hi[0] = pend_hi[1] = nullptr; }
    // The reference applies features in order (cram_decode.c:1260-1700).  A direct access to bytes a deferred copy of THIS record will
    // write later would see (or be clobbered by) the wrong order: carry the record's pending copies out first.
    HGR_FN void note_access(int which, const uint8_t *p, uint32_t n) {
        if (!jobs || njobs == rec_job0 || !pend_lo[which] || p >= pend_hi[which] || p + n <= pend_lo[which]) return;
        for (uint32_t j = rec_job0; j < njobs; j++) copy_bytes(jobs[j].dst, jobs[j].src, jobs[j].n);
        njobs = rec_job0; pend_lo[0] = pend_lo[1] = pend_hi[0] = pend_hi[1] = nullptr;
    }
    // what decode_features / decode_body ask of a reader (the column reader of cram_records_fast.h answers the same questions)
    HGR_FN bool has(int s) const { return P->codec_of[s] >= 0; }
    HGR_FN uint32_t cigar_cap() const { return S->cigar_cap; }
    HGR_FN uint32_t aux_cap() const { return S->aux_cap; }
    HGR_FN int32_t decode_md() const { return S->decode_md; }
    HGR_FN const uint8_t *data() const { return S->data; }

    // ---- CORE bit stream (get_bit_MSB / get_bits_MSB, cram_codecs.c:73-200) ----
    HGR_FN bool need_bits(uint64_t n) { if (bit + n > (uint64_t)S->core_len * 8u) { if (!err) err = ERR_MALFORMED; return false; } return true; }
    HGR_FN uint32_t bit1() {
        const uint32_t at = (uint32_t)(bit >> 3);
        const uint32_t byte = wbuf ? *win((uint32_t)P->nslots, S->core_off, S->core_len, at, 1) : S->data[S->core_off + at];
        const uint32_t b = (byte >> (7u - (bit & 7u))) & 1u; bit++; return b;
    }
    HGR_FN uint32_t bits(int n) { uint32_t v = 0; for (int i = 0; i < n; i++) v = (v << 1) | bit1(); return v; }

    // ---- EXTERNAL blocks ----
    HGR_FN bool slot_ok(int32_t s) { if (s < 0 || s >= P->nslots || S->blk_len[s] == 0xffffffffu) { if (!err) err = ERR_MALFORMED; return false; } return true; }
    HGR_FN int32_t ext_itf8(int32_t s) {                          // itf8_get at the cursor (cram_external_decode_int)
        if (!slot_ok(s)) return 0;
        const uint32_t n = S->blk_len[s]; uint32_t c = S->cursor[s];
        if (c >= n) { if (!err) err = ERR_MALFORMED; return 0; }
        // all five possible bytes at once (from the window, or one round trip), then the length is read off the first
        uint32_t b0, b1, b2, b3, b4;
        if (wbuf) { const HGR_LDS uint8_t *p = win((uint32_t)s, S->blk_off[s], n, c, 5); b0 = p[0]; b1 = p[1]; b2 = p[2]; b3 = p[3]; b4 = p[4]; }
        else { const uint8_t *p = S->data + S->blk_off[s] + c; b0 = p[0]; b1 = c + 1 < n ? p[1] : 0u; b2 = c + 2 < n ? p[2] : 0u; b3 = c + 3 < n ? p[3] : 0u; b4 = c + 4 < n ? p[4] : 0u; }
        const int extra = b0 < 0x80 ? 0 : b0 < 0xc0 ? 1 : b0 < 0xe0 ? 2 : b0 < 0xf0 ? 3 : 4;
        if (c + (uint32_t)extra >= n) { if (!err) err = ERR_MALFORMED; return 0; }
        uint32_t v;
        if (extra == 0) v = b0;
        else if (extra == 1) v = ((b0 & 0x3f) << 8) | b1;
        else if (extra == 2) v = ((b0 & 0x1f) << 16) | (b1 << 8) | b2;
        else if (extra == 3) v = ((b0 & 0x0f) << 24) | (b1 << 16) | (b2 << 8) | b3;
        else v = ((b0 & 0x0f) << 28) | (b1 << 20) | (b2 << 12) | (b3 << 4) | (b4 & 0x0f);
        S->cursor[s] = c + 1u + (uint32_t)extra;
        return (int32_t)v;
    }
    HGR_FN void ext_bytes(int32_t s, uint8_t *out, uint32_t n, int which = 2) {      // cram_external_decode_char
        if (!slot_ok(s)) return;
        const uint32_t c = S->cursor[s];
        if (n > S->blk_len[s] || c > S->blk_len[s] - n) { if (!err) err = ERR_MALFORMED; return; }
        if (out && n == 1u) out[0] = wbuf ? *win((uint32_t)s, S->blk_off[s], S->blk_len[s], c, 1) : S->data[S->blk_off[s] + c];     // a byte series (FC, BS, BA, QS of a feature)
        else if (out) bulk(out, S->data + S->blk_off[s] + c, n, which);
        S->cursor[s] = c + n;
    }

    // ---- one value of an integer / byte series ----
    HGR_FN int32_t huffman(const Codec &C) {
        if (C.b <= 0) { if (!err) err = ERR_MALFORMED; return 0; }
        const HuffCode *h = P->huff + C.a;
        if (h[0].len == 0 && C.b == 1) return h[0].symbol;                // one symbol, no bits (cram_huffman_decode_int0)
        uint32_t val = 0; int len = 0;
        for (int i = 0; i < C.b; i++) {                                   // codes are sorted by length: extend, then compare
            const int d = h[i].len - len;
            if (d > 0) { if (!need_bits((uint64_t)d)) return 0; val = (val << d) | bits(d); len = h[i].len; }
            if (h[i].code == val) return h[i].symbol;
        }
        if (!err) err = ERR_MALFORMED;
        return 0;
    }
    HGR_FN int32_t value(int32_t ci, bool as_byte) {
        if (ci < 0) { if (!err) err = ERR_MALFORMED; return 0; }
        const Codec C = P->codecs[ci];
        switch (C.kind) {
        case E_EXTERNAL:
            if (as_byte) { uint8_t b = 0; ext_bytes(C.a, &b, 1); return b; }
            return ext_itf8(C.a);
        case E_HUFFMAN: return huffman(C);
        case E_BETA: if (C.b < 0 || C.b > 32 || !need_bits((uint64_t)C.b)) { if (!err) err = ERR_MALFORMED; return 0; } return (int32_t)(bits(C.b) - (uint32_t)C.a);
        case E_GAMMA: {
            int nz = 0;
            for (;;) { if (!need_bits(1)) return 0; if (bit1()) break; if (++nz > 31) { if (!err) err = ERR_MALFORMED; return 0; } }
            if (!need_bits((uint64_t)nz)) return 0;
            uint32_t v = 1; for (int i = 0; i < nz; i++) v = (v << 1) | bit1();
            return (int32_t)(v - (uint32_t)C.a);
        }
        case E_SUBEXP: {
            int i = 0;
            for (;;) { if (!need_bits(1)) return 0; if (!bit1()) break; if (++i > 31) { if (!err) err = ERR_MALFORMED; return 0; } }
            const int tail = i ? i + C.b - 1 : C.b;
            if (tail < 0 || tail > 31 || !need_bits((uint64_t)tail)) { if (!err) err = ERR_MALFORMED; return 0; }
            uint32_t v = bits(tail);
            if (i) v += 1u << (i + C.b - 1);
            return (int32_t)(v - (uint32_t)C.a);
        }
        default: if (!err) err = ERR_UNSUPPORTED; return 0;
        }
    }
    HGR_FN int32_t ival(int s) { return value(P->codec_of[s], false); }
    HGR_FN int32_t bval(int s) { return value(P->codec_of[s], true); }

    // ---- one item of a byte-array series: the bytes go to out (may be null), the length is returned ----
    HGR_FN int32_t array(int32_t ci, uint8_t *out, uint32_t cap, int which = 2) {
        if (ci < 0) { if (!err) err = ERR_MALFORMED; return 0; }
        const Codec C = P->codecs[ci];
        if (C.kind == E_BYTE_ARRAY_STOP) {                                // cram_byte_array_stop_decode_char
            if (!slot_ok(C.a)) return 0;
            const uint32_t n = S->blk_len[C.a]; uint32_t c = S->cursor[C.a], k = 0;
            bool found = false;
            while (c < n && !found) {                                     // 16 bytes per round trip, then the scan runs on registers
                uint8_t t[16];
                const uint32_t m = n - c < 16u ? n - c : 16u;
                if (wbuf) {
                    const HGR_LDS uint8_t *w = win((uint32_t)C.a, S->blk_off[C.a], n, c, 16);
#pragma unroll
                    for (int j = 0; j < 16; j++) t[j] = (uint32_t)j < m ? w[j] : 0;
                } else {
                    const uint8_t *g = S->data + S->blk_off[C.a] + c;
#pragma unroll
                    for (int j = 0; j < 16; j++) t[j] = (uint32_t)j < m ? g[j] : 0;
                }
                uint32_t j = m;                                           // first stop byte among the m (constant indices: t stays in registers)
#pragma unroll
                for (int q = 15; q >= 0; q--) if ((uint32_t)q < m && t[q] == (uint8_t)C.b) j = (uint32_t)q;
                found = j < m;
                if (out) {
                    if (k + j > cap) { if (!err) err = ERR_UNSUPPORTED; return 0; }
#pragma unroll
                    for (int q = 0; q < 16; q++) if ((uint32_t)q < j) out[k + q] = t[q];
                }
                k += j; c += j;
            }
            if (!found) { if (!err) err = ERR_MALFORMED; return 0; }      // no stop byte
            S->cursor[C.a] = c + 1u;
            return (int32_t)k;
        }
        if (C.kind == E_BYTE_ARRAY_LEN) {                                 // length from one codec, bytes from the other
            const int32_t len = value(C.a, false);
            if (err) return 0;
            if (len < 0) { err = ERR_MALFORMED; return 0; }
            if (out && (uint32_t)len > cap) { err = ERR_UNSUPPORTED; return 0; }
            if (C.b < 0) { err = ERR_MALFORMED; return 0; }
            const Codec V = P->codecs[C.b];
            if (V.kind == E_EXTERNAL) ext_bytes(V.a, out, (uint32_t)len, which);
            else {
                // bytes out of a bit codec (no writer we know does this): a damaged length paired with a zero-bit codec must not spin -- one
                // item is capped, and its bytes count against the same budget as the feature walk, whether or not anybody wants them
                work += ((uint64_t)len >> 4) + 1u;
                if ((uint32_t)len > (1u << 24) || work > 16ull * S->cigar_cap + (1ull << 16)) { err = ERR_UNSUPPORTED; return 0; }
                for (int32_t i = 0; i < len && !err; i++) { const int32_t b = value(C.b, true); if (out) out[i] = (uint8_t)b; }
            }
            return len;
        }
        if (!err) err = ERR_UNSUPPORTED;                                  // array series through a scalar codec: not written by any encoder we know
        return 0;
    }
    HGR_FN int32_t array_s(int s, uint8_t *out, uint32_t cap, int which) { return array(P->codec_of[s], out, cap, which); }
    // n bytes of a byte series in one go: the qualities of a record, the bases of an unmapped read (cram_decode.c:2917-2953)
    HGR_FN void bytes_bulk(int s, uint8_t *out, uint32_t n, int which) {
        if (P->codec_of[s] < 0) { if (!err) err = ERR_MALFORMED; return; }
        const Codec C = P->codecs[P->codec_of[s]];
        if (C.kind == E_EXTERNAL) ext_bytes(C.a, out, n, which);
        else for (uint32_t i = 0; i < n && !err; i++) { const int32_t b = bval(s); if (out) out[i] = (uint8_t)b; }
    }
};

// cram_decode_seq (cram_decode.c:1096-1900): features -> CIGAR, alignment end, and -- when the caller asked for them -- the bases
// (reference span + edits) and the qualities; MQ; MD:Z / NM regenerated like the reference's decode_md.
// RD: the reader that hands out the values of the series -- Reader above (one chain per slice) or the column reader of
// cram_records_fast.h (one lane per record, start positions from prefix sums).  DRY: count only -- the CIGAR ops and the generated
// aux bytes of the record are counted (ncig_total / naux advance), nothing is stored and no capacity is checked; the data-parallel
// path sizes its outputs with such a pass.
template <class RD, bool DRY>
HGR_FN void decode_features(RD &R, const Cols &O, int rec, int32_t cf, uint32_t &ncig_total, uint8_t *seq, uint8_t *qual, const RefSpan *ref,
                            uint32_t &naux, int has_md, int has_nm, int32_t len, int32_t ref_id, int64_t apos, uint32_t aux_stored) {
    const Plan *P = R.P;
    int64_t ref_pos = apos - 1;                                           // 0-based position of the next reference base (the record's fields come in as
                                                                          // arguments: reading a column back waits for every store in flight)
    int32_t prev_pos = 0, seq_pos = 1, cig_len = 0, cig_op = C_MATCH;
    const uint32_t cig0 = ncig_total;
    const uint8_t *refb = ref ? R.data() + ref_off(ref) : nullptr;            // refb[p - ref->start] = base at 1-based position p
    const int64_t ref_start = ref ? ref->start : 0, ref_end = ref ? ref->start + (int64_t)ref->len - 1 : 0, sq_len = ref ? ref->sq_len : 0;
    const bool have_ref = ref && ref_id >= 0;
    auto emit = [&](uint32_t l, int op) {
        if (!DRY) { if (ncig_total >= R.cigar_cap()) { if (!R.err) R.err = ERR_UNSUPPORTED; return; } O.cigar[ncig_total] = (l << 4) | (uint32_t)op; }
        ncig_total++;
    };
    auto flush_unless = [&](int op) { if (cig_len && cig_op != op) { emit((uint32_t)cig_len, cig_op); cig_len = 0; } };
    auto fill = [&](int32_t at, uint8_t c, int64_t n) { if (seq && n > 0) { R.note_access(0, seq + at, (uint32_t)n); for (int64_t i = 0; i < n; i++) seq[at + i] = c; } };
    auto copy_ref = [&](int32_t at, int64_t n) { if (seq && n > 0) R.bulk(seq + at, refb + (ref_pos + 1 - ref_start), (uint32_t)n, 0); };
    auto qual_touch = [&]() {                                             // "same as htsjdk"
        if (qual && !(cf & CF_PRESERVE_QUAL) && len > 0) { R.note_access(1, qual, (uint32_t)len); if (qual[0] == 255) for (int32_t i = 0; i < len; i++) qual[i] = 30; }
    };
    auto put_seq = [&](uint8_t *p, uint8_t c) { R.note_access(0, p, 1); *p = c; };
    auto put_qual = [&](uint8_t *p, uint8_t c) { R.note_access(1, p, 1); *p = c; };
    if (qual && !(cf & CF_PRESERVE_QUAL)) for (int32_t i = 0; i < len; i++) qual[i] = 255;
    // MD:Z / NM regeneration (decode_md, cram_decode.c:1111-1137): needs the reference and somewhere to put the tags
    const bool do_md = R.decode_md() != 0 && O.aux != nullptr;
    bool decode_md = do_md && ref && ref_id >= 0 && !has_md, decode_nm = do_md && ref && ref_id >= 0 && !has_nm;
    if (cf & CF_NO_SEQ) decode_md = decode_nm = false;
    uint32_t nm = 0; int32_t md_dist = 0;
    const uint32_t aux0 = naux;
    auto aux_char = [&](uint8_t c) {
        if (!DRY) { if (naux >= R.aux_cap()) { if (!R.err) R.err = ERR_UNSUPPORTED; return; } O.aux[naux] = c; }
        naux++;
    };
    auto aux_uint = [&](uint32_t v) { uint32_t div = 1; while (v / div >= 10u) div *= 10u; for (; div; div /= 10u) aux_char((uint8_t)('0' + (v / div) % 10u)); };   // BLOCK_APPEND_UINT
    auto md_char = [&](uint8_t c) { if (decode_md) { aux_uint((uint32_t)md_dist); aux_char(c); md_dist = 0; } };      // add_md_char
    // per-base look-ups go straight to the staged reference: a window in LDS was tried and LOST (151 ms against 131 ms for the bench
    // batch) -- the look-ups of one stretch are independent loads, sixteen per round trip, while every window access is a dependent LDS read
    auto ref_at = [&](int64_t p0) -> uint8_t { return refb[p0 + 1 - ref_start]; };                                      // base at 0-based position p0
    auto md_run = [&](int64_t n) {                                        // n reference bases copied as they are: only an N counts as a mismatch
        if (!(decode_md || decode_nm)) return;
        int64_t i = 0;
        for (; i + 16 <= n; i += 16) {                                    // 16 bases per round trip; a stretch without N just adds to the distance
            uint8_t t[16]; bool any = false;
#pragma unroll
            for (int k = 0; k < 16; k++) { t[k] = ref_at(ref_pos + i + k); any |= t[k] == 'N'; }
            if (!any) { md_dist += 16; continue; }
#pragma unroll
            for (int k = 0; k < 16; k++) { if (t[k] == 'N') { md_char('N'); nm++; } else md_dist++; }
        }
        for (; i < n; i++) { if (ref_at(ref_pos + i) == 'N') { md_char('N'); nm++; } else md_dist++; }
    };
    if (decode_md) { aux_char('M'); aux_char('D'); aux_char('Z'); }
    const int32_t fn = R.ival(S_FN);                                      // a series the walk needs and the map lacks is an error, as in the reference
    {
        for (int32_t f = 0; f < fn && !R.err; f++) {
            if (++R.work > 16ull * R.cigar_cap()) { R.err = ERR_UNSUPPORTED; break; }
            const int32_t op = R.bval(S_FC);
            int32_t pos = R.ival(S_FP) + prev_pos;
            if (R.err) break;
            if (pos <= 0) { R.err = ERR_MALFORMED; break; }
            if (len != 0 && pos > len) {
                const int32_t valid_end = (op == 'N' || op == 'P' || op == 'H' || op == 'D') ? len + 1 : len;
                if (pos > valid_end) { R.err = ERR_MALFORMED; break; }
            }
            if (pos > seq_pos) {
                if (have_ref) {                                           // the stretch up to the feature is a copy of the reference
                    if (ref_pos + pos - seq_pos > sq_len) {               // ... running off the end of the reference: pad with N
                        const int64_t rlen = sq_len - ref_pos;
                        if (rlen > 0) {
                            if (ref_pos + rlen > ref_end) { R.err = ERR_MALFORMED; break; }
                            if (len) { copy_ref(seq_pos - 1, rlen); if ((pos - seq_pos) - rlen > 0) fill(seq_pos - 1 + (int32_t)rlen, 'N', (pos - seq_pos) - rlen); }
                        } else if (len) fill(seq_pos - 1, 'N', len - seq_pos + 1);
                        if (md_dist >= 0) md_dist += pos - seq_pos;
                    } else {
                        if (ref_pos + pos - seq_pos > ref_end) { R.err = ERR_MALFORMED; break; }
                        md_run(pos - seq_pos);
                        if (len) copy_ref(seq_pos - 1, pos - seq_pos);
                    }
                }
                flush_unless(C_MATCH); cig_op = C_MATCH; cig_len += pos - seq_pos; ref_pos += pos - seq_pos; seq_pos = pos;
            }
            prev_pos = pos;
            uint8_t *sp = seq && len ? seq + (pos - 1) : nullptr;          // "cr->len ? &seq[pos-1] : NULL"
            uint8_t *qp = qual && len ? qual + (pos - 1) : nullptr;
            const uint32_t room = len ? (uint32_t)(len - (pos - 1)) : 0u;    // bytes a byte-array item may write
            switch (op) {
            case 'S': {
                if (cig_len) { emit((uint32_t)cig_len, cig_op); cig_len = 0; }
                int32_t n = 1;                                            // no SC codec: one unknown base (cram_decode.c:1317-1329)
                if (R.has(S_SC)) { if (sp) R.note_access(0, sp, room); n = R.array_s(S_SC, sp, room, 0); } else if (sp) put_seq(sp, 'N');
                emit((uint32_t)n, C_SOFT_CLIP); cig_op = C_SOFT_CLIP; seq_pos += n;
                break;
            }
            case 'X': {
                flush_unless(C_MATCH);
                const int32_t base = R.bval(S_BS) & 3;
                if (ref_id < 0 || ref_pos >= sq_len || !ref) {
                    if (seq && pos - 1 < len) put_seq(seq + (pos - 1), P->sm[16 + base]);
                    if (decode_md || decode_nm) { if (md_dist >= 0 && decode_md) aux_uint((uint32_t)md_dist); md_dist = -1; nm--; }
                } else {
                    const uint8_t rc = ref_pos < ref_end ? ref_at(ref_pos) : (uint8_t)'N';
                    const int l1 = (rc == 'A' || rc == 'a') ? 0 : (rc == 'C' || rc == 'c') ? 1 : (rc == 'G' || rc == 'g') ? 2 : (rc == 'T' || rc == 't') ? 3 : 4;
                    if (seq && pos - 1 < len) put_seq(seq + (pos - 1), P->sm[4 * l1 + base]);
                    md_char(rc);
                }
                nm++;
                cig_op = C_MATCH; cig_len++; seq_pos++; ref_pos++;
                break;
            }
            case 'D': {
                flush_unless(C_DEL);
                {
                    const int32_t v = R.ival(S_DL);
                    if (v < 0) { if (!R.err) R.err = ERR_MALFORMED; break; }
                    if (decode_md || decode_nm) {                         // ^ + the deleted reference bases (cram_decode.c:1417-1448)
                        if (ref_pos + v > ref_end) { R.err = ERR_MALFORMED; break; }
                        if (md_dist >= 0 && decode_md) aux_uint((uint32_t)md_dist);
                        if (ref_pos + v <= sq_len) {
                            if (decode_md) { aux_char('^'); for (int32_t i = 0; i < v; i++) aux_char(ref_at(ref_pos + i)); md_dist = 0; }
                            nm += (uint32_t)v;
                        } else {
                            if (sq_len >= ref_pos) {
                                if (decode_md) { aux_char('^'); for (int64_t i = 0; i < sq_len - ref_pos; i++) aux_char(ref_at(ref_pos + i)); aux_uint(0); }
                                nm += (uint32_t)(sq_len - ref_pos);
                            }
                            md_dist = -1;
                        }
                    }
                    cig_op = C_DEL; cig_len += v; ref_pos += v;
                }
                break;
            }
            case 'I': {
                flush_unless(C_INS);
                { if (sp) R.note_access(0, sp, room); const int32_t n = R.array_s(S_IN, sp, room, 0); cig_op = C_INS; cig_len += n; seq_pos += n; nm += (uint32_t)n; }
                break;
            }
            case 'i': { flush_unless(C_INS); const int32_t b = R.bval(S_BA); if (sp) put_seq(sp, (uint8_t)b); cig_op = C_INS; cig_len++; seq_pos++; nm++; break; }
            case 'b': {
                flush_unless(C_MATCH);
                if (sp) R.note_access(0, sp, room);
                const int32_t n = R.array_s(S_BB, sp, room, 0);
                if (decode_md || decode_nm) {                             // every stored base counts as a mismatch (cram_decode.c:1515-1541)
                    if (md_dist >= 0 && decode_md) aux_uint((uint32_t)md_dist);
                    int32_t x = 0;
                    for (; x < n; x++) {
                        if (x && decode_md) aux_uint(0);
                        if (ref_pos + x >= sq_len || !ref) { md_dist = -1; break; }
                        if (decode_md) { if (ref_pos + x >= ref_end) { R.err = ERR_MALFORMED; break; } aux_char(ref_at(ref_pos + x)); }
                    }
                    nm += (uint32_t)x;
                    md_dist = 0;
                }
                cig_op = C_MATCH; cig_len += n; seq_pos += n; ref_pos += n;
                break;
            }
            case 'q': flush_unless(C_MATCH); qual_touch(); if (qp) R.note_access(1, qp, room); (void)R.array_s(S_QQ, qp, room, 1); cig_op = C_MATCH; break;
            case 'B': {
                flush_unless(C_MATCH);
                const int32_t b = R.bval(S_BA); if (sp) put_seq(sp, (uint8_t)b);
                if (decode_md || decode_nm) {                             // cram_decode.c:1593-1610
                    if (md_dist >= 0 && decode_md) aux_uint((uint32_t)md_dist);
                    if (ref_pos >= sq_len || !ref) md_dist = -1;
                    else { if (decode_md) { if (ref_pos >= ref_end) { R.err = ERR_MALFORMED; break; } aux_char(ref_at(ref_pos)); } nm++; md_dist = 0; }
                }
                qual_touch();
                const int32_t q = R.bval(S_QS); if (qp) put_qual(qp, (uint8_t)q);
                cig_op = C_MATCH; cig_len++; seq_pos++; ref_pos++;
                break;
            }
            case 'Q': { qual_touch(); const int32_t q = R.bval(S_QS); if (qp) put_qual(qp, (uint8_t)q); break; }
            case 'H': {
                flush_unless(C_HARD_CLIP);
                { const int32_t v = R.ival(S_HC); if (v < 0) { if (!R.err) R.err = ERR_MALFORMED; break; } cig_op = C_HARD_CLIP; cig_len += v; }
                break;
            }
            case 'P': {
                flush_unless(C_PAD);
                { const int32_t v = R.ival(S_PD); if (v < 0) { if (!R.err) R.err = ERR_MALFORMED; break; } cig_op = C_PAD; cig_len += v; }
                break;
            }
            case 'N': {
                flush_unless(C_REF_SKIP);
                { const int32_t v = R.ival(S_RS); if (v < 0) { if (!R.err) R.err = ERR_MALFORMED; break; } cig_op = C_REF_SKIP; cig_len += v; ref_pos += v; }
                break;
            }
            default: if (!R.err) R.err = ERR_MALFORMED; break;
            }
        }
        // an implicit match for the bases no feature accounted for (cram_decode.c:1710-1795)
        if (!R.err && len >= seq_pos) {
            if (have_ref) {
                if (ref_pos + len - seq_pos + 1 > sq_len) {
                    const int64_t rlen = sq_len - ref_pos;
                    if (rlen > 0) {
                        if (ref_pos + rlen > ref_end) R.err = ERR_MALFORMED;
                        else { if (seq_pos - 1 + rlen < len) copy_ref(seq_pos - 1, rlen); if ((len - seq_pos + 1) - rlen > 0) fill(seq_pos - 1 + (int32_t)rlen, 'N', (len - seq_pos + 1) - rlen); }
                    } else if (len - seq_pos + 1 > 0) fill(seq_pos - 1, 'N', len - seq_pos + 1);
                    if (md_dist >= 0) md_dist += len - seq_pos + 1;
                } else {
                    if (len - seq_pos + 1 > 0) { if (ref_pos + len - seq_pos + 1 > ref_end) R.err = ERR_MALFORMED; else { md_run(len - (seq_pos - 1)); copy_ref(seq_pos - 1, len - (seq_pos - 1)); } }
                    ref_pos += len - seq_pos + 1;
                }
            } else if (ref_id >= 0) ref_pos += len - seq_pos + 1;
            flush_unless(C_MATCH); cig_op = C_MATCH; cig_len += len - seq_pos + 1;
        }
    }
    if (decode_md && md_dist >= 0) aux_uint((uint32_t)md_dist);
    if (cig_len) emit((uint32_t)cig_len, cig_op);
    const int32_t mq = R.ival(S_MQ);
    if (!DRY) {
        O.cigar_off[rec] = cig0; O.ncigar[rec] = (int32_t)(ncig_total - cig0);
        O.aend[rec] = ref_pos > apos ? ref_pos : apos;
        O.mqual[rec] = mq;
    }
    if ((cf & CF_PRESERVE_QUAL) && !R.err) R.bytes_bulk(S_QS, qual, (uint32_t)len, 1);      // len quality bytes
    if (!DRY && (cf & CF_NO_SEQ)) O.len[rec] = 0;
    if (decode_md) aux_char(0);                                           // MD:Z: is a NUL-terminated string
    if (decode_nm) {                                                      // NM in the narrowest unsigned type (cram_decode.c:1884-1908)
        aux_char('N'); aux_char('M');
        if (nm <= 0xffu) { aux_char('C'); aux_char((uint8_t)nm); }
        else if (nm <= 0xffffu) { aux_char('S'); aux_char((uint8_t)nm); aux_char((uint8_t)(nm >> 8)); }
        else { aux_char('I'); aux_char((uint8_t)nm); aux_char((uint8_t)(nm >> 8)); aux_char((uint8_t)(nm >> 16)); aux_char((uint8_t)(nm >> 24)); }
    }
    if (!DRY && O.aux && !R.err) O.aux_len[rec] = (int32_t)(aux_stored + (naux - aux0));
}

// What follows the fixed fields and the tags of a record in cram_decode_slice's loop (cram_decode.c:2890-2967): bases and qualities of a
// mapped record through the feature walk, of an unmapped one straight from BA / QS, and the quality reversal of QO = 0 files.
template <class RD, bool DRY>
HGR_FN void decode_body(RD &R, const Cols &O, int rec, int32_t bf, int32_t cf, int32_t len, int32_t ref_id, int64_t apos, uint8_t *seq, uint8_t *qual,
                        const RefSpan *ref, uint32_t &ncig, uint32_t &naux, int has_md, int has_nm, uint32_t aux_stored) {
    if (seq && !ref) for (int32_t i = 0; i < len; i++) seq[i] = '=';
    if (!(bf & BAM_FUNMAP)) {
        if (apos <= 0) { R.err = ERR_MALFORMED; return; }
        decode_features<RD, DRY>(R, O, rec, cf, ncig, seq, qual, ref, naux, has_md, has_nm, len, ref_id, apos, aux_stored);
    } else {
        if (!DRY) { O.cigar_off[rec] = ncig; O.ncigar[rec] = 0; O.aend[rec] = apos; O.mqual[rec] = 0; }
        if (len) R.bytes_bulk(S_BA, seq, (uint32_t)len, 0);
        if (R.err) return;
        if (cf & CF_PRESERVE_QUAL) R.bytes_bulk(S_QS, qual, (uint32_t)len, 1);
        else if (qual) for (int32_t i = 0; i < len; i++) qual[i] = 255;
    }
    if (qual && !R.err && !R.P->qs_seq_orient && (bf & BAM_FREVERSE)) {  // qualities stored in read orientation (cram_decode.c:2957-2965)
        R.note_access(1, qual, (uint32_t)len);
        for (int32_t i = 0, j = len - 1; i < j; i++, j--) { const uint8_t t = qual[i]; qual[i] = qual[j]; qual[j] = t; }
    }
}

// cram_decode_aux (cram_decode.c:2008-2137): tag list of the record (TL -> dictionary line), then one value per tag.  The values are
// stored in BAM encoding already; with O.aux the record's tags are written as tag[2] type value ..., else only consumed.
HGR_FN void decode_aux(Reader &R, const Cols &O, int rec, uint32_t &naux, int &has_md, int &has_nm) {
    const Plan *P = R.P;
    const int32_t tl = R.ival(S_TL);
    if (O.aux) { O.aux_off[rec] = naux; O.aux_len[rec] = 0; }
    if (R.err) return;
    if (tl < 0 || tl >= P->nTL) { R.err = ERR_MALFORMED; return; }
    const uint32_t start = naux;
    for (int32_t t = P->tl_off[tl]; t < P->tl_off[tl + 1] && !R.err; t++) {
        if (++R.work > 16ull * R.S->cigar_cap) { R.err = ERR_UNSUPPORTED; return; }
        const int32_t ci = P->tl_codec[t], tag = P->tl_tag[t];
        if ((tag >> 8) == (('M' << 8) | 'D')) has_md = 1;                 // stored: not regenerated (cram_decode.c:2044-2047)
        if ((tag >> 8) == (('N' << 8) | 'M')) has_nm = 1;
        if (ci < 0) { R.err = ERR_MALFORMED; return; }
        uint8_t *out = nullptr; uint32_t cap = 0;
        if (O.aux) {
            if (naux + 3u > R.S->aux_cap) { R.err = ERR_UNSUPPORTED; return; }
            O.aux[naux] = (uint8_t)(tag >> 16); O.aux[naux + 1] = (uint8_t)(tag >> 8); O.aux[naux + 2] = (uint8_t)tag;
            naux += 3; out = O.aux + naux; cap = R.S->aux_cap - naux;
        }
        const int32_t k = P->codecs[ci].kind;
        int32_t n;
        if (k == E_BYTE_ARRAY_LEN || k == E_BYTE_ARRAY_STOP) n = R.array(ci, out, cap);
        else { const int32_t b = R.value(ci, true); n = 1; if (out) { if (cap < 1) { R.err = ERR_UNSUPPORTED; return; } out[0] = (uint8_t)b; } }   // out_sz = 1
        if (R.err) return;
        if (O.aux) {
            naux += (uint32_t)n;
            if (tag == (('c' << 16) | ('F' << 8) | 'C') && n == 1) {          // cF:C is a note to the decoder, not a tag (cram_decode.c:2107-2118)
                const uint8_t cF = O.aux[naux - 1];
                naux -= 4;
                if ((cF & 1) && has_md == 0) has_md = 1;
                if ((cF & 2) && has_nm == 0) has_nm = 1;
            }
        }
    }
    if (O.aux) O.aux_len[rec] = (int32_t)(naux - start);
}

// cram_decode_slice_xref (cram_decode.c:2140-2307)
HGR_FN int xref(const Cols &O, int32_t nrec) {
    for (int32_t rec = 0; rec < nrec; rec++) {
        if (O.mate_line[rec] >= 0) {
            if (O.mate_line[rec] < nrec) {
                if (O.tlen[rec] == TLEN_UNSET) {
                    int32_t id1 = rec, id2 = rec, ref = O.ref_id[rec], left_cnt = 0, right_cnt = 0;
                    int64_t aleft = O.apos[rec], aright = O.aend[rec];
                    do {
                        if (aleft > O.apos[id2]) { aleft = O.apos[id2]; left_cnt = 1; } else if (aleft == O.apos[id2]) left_cnt++;
                        if (aright < O.aend[id2]) { aright = O.aend[id2]; right_cnt = 1; } else if (aright == O.aend[id2]) right_cnt++;
                        if (O.mate_line[id2] == -1) { O.mate_line[id2] = rec; break; }
                        if (O.mate_line[id2] <= id2 || O.mate_line[id2] >= nrec) return -1;
                        id2 = O.mate_line[id2];
                        if (O.ref_id[id2] != ref) ref = -1;
                    } while (id2 != id1);
                    if (ref != -1) {
                        int64_t tlen = aright - aleft + 1;
                        id1 = id2 = rec;
                        if (O.apos[id2] == aleft && (O.aend[id2] < aright || left_cnt <= 1)) { O.tlen[id2] = tlen; tlen = -tlen; }
                        else if (O.apos[id2] == aleft && O.aend[id2] == aright && left_cnt > 1 && right_cnt > 1) {
                            if (O.flags[id2] & BAM_FREAD1) { O.tlen[id2] = tlen; tlen = -tlen; } else O.tlen[id2] = -tlen;
                        } else O.tlen[id2] = -tlen;
                        id2 = O.mate_line[id2];
                        while (id2 != id1) { O.tlen[id2] = tlen; id2 = O.mate_line[id2]; }
                    } else {
                        id1 = id2 = rec;
                        O.tlen[id2] = 0;
                        id2 = O.mate_line[id2];
                        while (id2 != id1) { O.tlen[id2] = 0; id2 = O.mate_line[id2]; }
                    }
                }
                const int32_t m = O.mate_line[rec];
                O.mate_pos[rec] = O.apos[m];
                O.mate_ref_id[rec] = O.ref_id[m];
                O.flags[rec] |= BAM_FPAIRED;
                if (O.flags[m] & BAM_FUNMAP) { O.flags[rec] |= BAM_FMUNMAP; O.tlen[rec] = 0; }
                if (O.flags[rec] & BAM_FUNMAP) O.tlen[rec] = 0;
                if (O.flags[m] & BAM_FREVERSE) O.flags[rec] |= BAM_FMREVERSE;
            }
        } else {
            if (O.mate_flags[rec] & CRAM_M_REVERSE) O.flags[rec] |= BAM_FPAIRED | BAM_FMREVERSE;
            if (O.mate_flags[rec] & CRAM_M_UNMAP) O.flags[rec] |= BAM_FMUNMAP;
            if (!(O.flags[rec] & BAM_FPAIRED)) O.mate_ref_id[rec] = -1;
        }
        if (O.tlen[rec] == TLEN_UNSET) O.tlen[rec] = 0;
    }
    for (int32_t rec = 0; rec < nrec; rec++) if (O.explicit_tlen[rec] != TLEN_UNSET) O.tlen[rec] = O.explicit_tlen[rec];
    return 0;
}

// The record loop of cram_decode_slice (cram_decode.c:2553-2967).  Returns 0, ERR_MALFORMED or ERR_UNSUPPORTED.
HGR_FN int decode_slice(const Plan *P, const Slice *S, const Cols &O) {
    Reader R; R.P = P; R.S = S; R.bit = 0; R.work = 0; R.err = 0;
    R.wbuf = S->wbuf; R.wpos = S->wpos;
    if (R.wbuf) for (int32_t i = 0; i <= P->nslots; i++) R.wpos[i] = 0xffffff00u;          // windows empty (blocks, CORE)
    R.jobs = P->qs_seq_orient ? S->jobs : nullptr; R.njobs = 0; R.job_cap = S->job_cap; R.begin_record();      // the quality reversal of QO = 0 files reads the record back: no deferral there
    for (int32_t i = 0; i < P->nslots; i++) S->cursor[i] = 0;
    uint32_t ncig = 0, nname = 0, naux = 0;
    int64_t last_apos = S->ref_seq_start;
    for (int32_t rec = 0; rec < S->nrec && !R.err; rec++) {
        R.begin_record();
        const int32_t bf = R.ival(S_BF);
        if (!R.err && (bf < 0 || bf >= 0x1000)) R.err = ERR_MALFORMED;
        O.flags[rec] = bf;
        const int32_t cf = R.ival(S_CF);
        O.cram_flags[rec] = cf;
        int32_t ref_id = S->ref_seq_id;
        if (S->ref_seq_id == -2) ref_id = R.ival(S_RI);
        if (!R.err && (ref_id < -1 || ref_id >= S->nref)) R.err = ERR_MALFORMED;
        O.ref_id[rec] = ref_id;
        const int32_t len = R.ival(S_RL);
        if (!R.err && len < 0) R.err = ERR_MALFORMED;
        O.len[rec] = len;
        int64_t apos;
        {
            apos = R.ival(S_AP);
            if (P->ap_delta) apos += last_apos;
            last_apos = apos;
            if (!R.err && S->ref_seq_id >= 0 && apos < S->ref_seq_start) R.err = ERR_MALFORMED;
        }
        O.apos[rec] = apos;
        O.rg[rec] = R.ival(S_RG);
        O.name_off[rec] = nname; O.name_len[rec] = 0;
        auto read_name = [&]() {
            const int32_t n = R.array(P->codec_of[S_RN], O.names + nname, S->name_cap - nname);
            if (!R.err) { O.name_len[rec] = n; nname += (uint32_t)n; }
        };
        if (P->rn_included) read_name();
        O.mate_pos[rec] = 0; O.mate_line[rec] = -1; O.mate_ref_id[rec] = -1; O.explicit_tlen[rec] = TLEN_UNSET;
        O.mate_flags[rec] = 0; O.tlen[rec] = TLEN_UNSET;
        if (cf & CF_DETACHED) {
            O.mate_flags[rec] = R.ival(S_MF);
            if (!P->rn_included) { O.name_off[rec] = nname; read_name(); }
            { const int32_t v = R.ival(S_NS); if (!R.err && (v < -1 || v >= S->nref)) R.err = ERR_MALFORMED; O.mate_ref_id[rec] = v; }
            O.mate_pos[rec] = R.ival(S_NP);
            O.tlen[rec] = R.ival(S_TS);
        } else if (cf & CF_MATE_DOWNSTREAM) {
            O.mate_line[rec] = R.ival(S_NF) + rec + 1;
            if (cf & CF_EXPLICIT_TLEN) O.explicit_tlen[rec] = R.ival(S_TS);
        } else if (cf & CF_EXPLICIT_TLEN) {
            O.explicit_tlen[rec] = R.ival(S_TS);
        }
        int has_md = 0, has_nm = 0;
        const uint32_t aux_rec0 = naux;
        decode_aux(R, O, rec, naux, has_md, has_nm);
        if (R.err) break;
        // room for the bases / qualities of this record (cram_decode.c:2890-2906), and the reference span it aligns to
        uint8_t *seq = nullptr, *qual = nullptr;
        const RefSpan *ref = nullptr;
        if (O.seq) {
            uint64_t at;
            // room for len bases and qualities from the pool the slices of a launch share.  A request that does not fit takes NOTHING (a slice with
            // a damaged read length must not starve its neighbours); the launcher runs such a slice again when the pool has been emptied.
#if defined(__HIP_DEVICE_COMPILE__)
            {
                unsigned long long seen = *(volatile unsigned long long *)O.seq_pool, want;
                bool fits;
                do { want = seen; fits = want + (unsigned long long)len <= O.seq_cap; if (!fits) break; seen = atomicCAS(O.seq_pool, want, want + (unsigned long long)len); } while (seen != want);
                if (!fits) { R.err = ERR_POOL; break; }
                at = want;
            }
#else
            at = *O.seq_pool;
            if (at + (uint64_t)len > O.seq_cap) { R.err = ERR_UNSUPPORTED; break; }
            *O.seq_pool += (unsigned long long)len;
#endif
            O.seq_off[rec] = at; seq = O.seq + at; qual = O.qual + at;
        }
        for (int32_t i = 0; i < S->nrefs; i++) if (S->refs[i].ref_id == ref_id) { ref = &S->refs[i]; break; }
        if (ref && apos < ref->start) ref = nullptr;                      // the span does not reach back to this record: as if no reference had been given
        decode_body<Reader, false>(R, O, rec, bf, cf, len, ref_id, apos, seq, qual, ref, ncig, naux, has_md, has_nm, naux - aux_rec0);
    }
    O.totals[0] = ncig; O.totals[1] = nname; O.totals[2] = naux; O.totals[3] = R.njobs;
    if (R.err) return R.err;
    return xref(O, S->nrec) ? ERR_MALFORMED : 0;
}

}  // namespace hgr

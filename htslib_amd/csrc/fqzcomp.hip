// fqzcomp.hip -- CRAM 3.1 fqzcomp quality codec, block method 7, for MI355X (gfx950).
//
// Replaces fqz_decompress as called by cram_uncompress_block (reference cram/cram_io.c:1684-1695; implementation = htscodecs
// fqzcomp_qual.c, an ABSENT submodule).  Format and arithmetic per oracle/fqzcomp_oracle.c -- PARITY UNPINNED; the kernel is
// bit-exact with that oracle.
//
// A quality stream is ONE adaptive chain: every symbol is coded with the model its 16-bit context selects, and the context is
// a hash of the previous qualities, the position in the read and the number of changes so far, so symbol i+1 cannot start
// before symbol i is known.  The parallelism is ACROSS streams (one QS block per CRAM slice; a batch of slices gives hundreds
// to thousands).  Mapping: one wavefront per stream, as in arith.hip:
//   * the 65536 quality models (max_sym + 1 entries and their total each) live in a per-wavefront slot of global scratch --
//     5.5 to 67 MB, far beyond LDS; the model of a symbol is fetched with one 64-lane read (total + up to 4 x 64 entries at
//     once) and searched with a DPP prefix sum + ballot, so the dependent chain per quality is: model read -> divide -> search
//     -> context tables (LDS) -> next model read;
//   * the parameter block (qmap / qtab / ptab / dtab per parameter set, the selector table) is parsed ON THE HOST -- a few
//     hundred bytes per stream -- and shipped as a fixed-layout image that the wavefront copies to LDS; the record-level models
//     (4 length bytes, reverse, duplicate, selector) are LDS too;
//   * output bytes are gathered 64 per store; a record that is reversed or a duplicate of its predecessor is fixed up in
//     place when it ends (reversal is an involution, so "copy the predecessor as decoded, reverse at the very end" of the
//     oracle becomes "copy straight or mirrored, depending on whether the two reverse flags agree").
// Any number of parameter sets (the format allows 256; htscodecs writes one or two): the wavefront's LDS holds TWO at a time, the sets of a stream beyond
// the first two wait in a global overflow image and replace the less recently used slot when a record selects them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include "arith_dev.h"
#include "range_enc2_dev.h"

namespace hgq {
using hg::wave_sync;
using hga::rl;
// -DHG_FQZ_PROFILE: where the decoder's time per quality goes (core-clock ticks per phase, printed by the wavefront of stream 0; `make fqzprof`)
#ifdef HG_FQZ_PROFILE
#define FQ_T(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); fq_acc[slot] += n_ - fq_last; fq_last = n_; } while (0)
#else
#define FQ_T(slot) do { } while (0)
#endif

enum { GF_MULTI = 1, GF_STAB = 2, GF_REV = 4, PF_DEDUP = 2, PF_LEN = 4, PF_SEL = 8, PF_QMAP = 16, PF_PTAB = 32, PF_DTAB = 64, PF_QTAB = 128 };
constexpr uint32_t FQZ_VERS = 5, CTX_SIZE = 65536;
constexpr int HGQ_MAX_PARAM = 2;                                  // parameter sets resident in LDS (slots); a stream may have up to 256

// ---- the per-stream image the host builds (32-bit words) ------------------------------------------------------------
// head: [0] gflags [1] nparam [2] max_sel [3] ns = max_sym + 1 [4] offset of the range-coder bytes in the stream [5] ulen
//       [6] (encoder) number of records / (decoder) word offset of parameter set 2 in the overflow image [7] (encoder) have flags; then stab (256 bytes = 64 words); then per parameter set:
//       8 words {context, pflags, qmask, qshift, qloc, sloc, ploc, dloc}, qmap (256 bytes), qtab / dtab (256 x u16 each), ptab (1024 x u16)
constexpr uint32_t IMG_HEAD = 8, IMG_STAB = 64, IMG_PSCAL = 8, IMG_QMAP = 64, IMG_QTAB = 128, IMG_DTAB = 128, IMG_PTAB = 512;
constexpr uint32_t IMG_PARAM = IMG_PSCAL + IMG_QMAP + IMG_QTAB + IMG_DTAB + IMG_PTAB;          // 840 words
constexpr uint32_t IMG_WORDS = IMG_HEAD + IMG_STAB + HGQ_MAX_PARAM * IMG_PARAM;               // 1752 words
// record-level models in LDS: [total, entries...] each
constexpr uint32_t M_LEN = 0, M_REV = 4 * 257, M_DUP = M_REV + 3, M_SEL = M_DUP + 3, M_WORDS = M_SEL + 257;   // 1291 words
constexpr uint32_t POOLW = ((M_WORDS + 3) & ~3u) + IMG_WORDS;
constexpr int QPF = 4;                                           // decoder: models fetched ahead per quality (see fqz_decode_kernel)

// ---- host: parameter block -> image ----------------------------------------------------------------------------------
static int read_array_h(const uint8_t *in, size_t in_size, uint16_t *array, int size) {
    uint8_t R[1024];
    int i, j, z, last = -1;
    for (i = j = z = 0; z < size && (size_t)i < in_size; i++) {
        const int run = in[i];
        R[j++] = (uint8_t)run;
        z += run;
        if (run == last) {
            if ((size_t)i + 1 >= in_size) return -1;
            int copy = in[++i];
            z += run * copy;
            while (copy-- && z <= size && j < 1024) R[j++] = (uint8_t)run;
        }
        if (j >= 1024) return -1;
        last = run;
    }
    const int nb = i, rmax = j;
    for (i = j = z = 0; j < size; i++) {
        int len = 0, part;
        if (z >= rmax) return -1;
        do { part = R[z++]; len += part; } while (part == 255 && z < rmax);
        while (len && j < size) { len--; array[j++] = (uint16_t)i; }
    }
    return nb;
}

// 0 ok, -1 malformed, -3 more parameter sets than the kernel keeps in LDS
static int build_image(const uint8_t *in, uint32_t n, uint32_t want_ulen, uint32_t *img, std::vector<uint32_t> &overflow) {
    memset(img, 0, IMG_WORDS * 4);
    const size_t over0 = overflow.size();
    uint32_t ulen = 0, k = 0; uint8_t c;
    do { if (k >= n || k >= 5) return -1; c = in[k++]; ulen = (ulen << 7) | (c & 0x7fu); } while (c & 0x80u);
    if (ulen != want_ulen) return -1;
    size_t p = k;
    if ((size_t)n < p + 10 || in[p++] != FQZ_VERS) return -1;
    const uint32_t gflags = in[p++];
    const uint32_t nparam = (gflags & GF_MULTI) ? in[p++] : 1u;
    if (nparam == 0) return -1;
    uint32_t max_sel = nparam > 1 ? nparam - 1 : 0;
    uint8_t *stab = (uint8_t *)(img + IMG_HEAD);
    if (gflags & GF_STAB) {
        max_sel = in[p++];
        uint16_t t[256];
        const int r = read_array_h(in + p, n - p, t, 256);
        if (r < 0) return -1;
        p += (size_t)r;
        for (int i = 0; i < 256; i++) { if (t[i] >= nparam) return -1; stab[i] = (uint8_t)t[i]; }
    } else {
        for (uint32_t i = 0; i < 256; i++) stab[i] = (uint8_t)(i < nparam ? i : nparam - 1);
    }
    if (nparam > (uint32_t)HGQ_MAX_PARAM) overflow.resize(over0 + (size_t)(nparam - HGQ_MAX_PARAM) * IMG_PARAM, 0u);
    uint32_t max_sym = 0;
    for (uint32_t s = 0; s < nparam; s++) {
        uint32_t *P = s < (uint32_t)HGQ_MAX_PARAM ? img + IMG_HEAD + IMG_STAB + s * IMG_PARAM : overflow.data() + over0 + (size_t)(s - HGQ_MAX_PARAM) * IMG_PARAM;
        if (p + 7 > n) return -1;
        const uint32_t pflags = in[p + 2], msym = in[p + 3];
        P[0] = in[p] | (uint32_t)in[p + 1] << 8; P[1] = pflags;
        P[2] = (1u << (in[p + 4] >> 4)) - 1u; P[3] = in[p + 4] & 15u;
        P[4] = in[p + 5] >> 4; P[5] = in[p + 5] & 15u; P[6] = in[p + 6] >> 4; P[7] = in[p + 6] & 15u;
        p += 7;
        uint8_t *qmap = (uint8_t *)(P + IMG_PSCAL);
        uint16_t *qtab = (uint16_t *)(P + IMG_PSCAL + IMG_QMAP), *dtab = qtab + 256, *ptab = dtab + 256;
        for (uint32_t i = 0; i < 256; i++) { qmap[i] = (uint8_t)i; qtab[i] = (uint16_t)i; }
        if (pflags & PF_QMAP) { if (p + msym > n) return -1; for (uint32_t i = 0; i < msym; i++) qmap[i] = in[p++]; }
        if (pflags & PF_QTAB) { const int r = read_array_h(in + p, n - p, qtab, 256); if (r < 0) return -1; p += (size_t)r; }
        if (pflags & PF_PTAB) { const int r = read_array_h(in + p, n - p, ptab, 1024); if (r < 0) return -1; p += (size_t)r; }
        if (pflags & PF_DTAB) { const int r = read_array_h(in + p, n - p, dtab, 256); if (r < 0) return -1; p += (size_t)r; }
        if (msym > max_sym) max_sym = msym;
    }
    if (p > n) return -1;
    img[0] = gflags; img[1] = nparam; img[2] = max_sel; img[3] = max_sym + 1u; img[4] = (uint32_t)p; img[5] = ulen; img[6] = (uint32_t)over0;
    if (overflow.size() > 0xffffffffull) return -1;
    return 0;
}

// ---- host: choice of parameters for the encoder -> header bytes + image ------------------------------------------------
// The format leaves the choice free (any parameter block a decoder accepts is valid).  Ours is modelled on the four presets
// of htscodecs (strat 0..3 = CRAM methods FQZ, FQZ_b, FQZ_c, FQZ_d, cram_io.c:2065-2068): {qbits, qshift, pbits, pshift, dbits,
// dshift, qloc, sloc, ploc, dloc}; the READ2 flag selects one of two parameter sets in preset 1; reverse-strand records are
// turned round when flags are supplied; duplicates are flagged when at least one record in ten repeats its predecessor;
// eight or fewer distinct values are coded through a quality map.
static const int PRESET[4][10] = {
    {10, 5, 4, -1, 2, 1, 0, 14, 10, 14},
    {8, 5, 7, 0, 0, 0, 0, 14, 8, 14},
    {12, 6, 2, 0, 2, 3, 0, 9, 12, 14},
    {12, 6, 0, 0, 0, 0, 0, 12, 0, 0},
};
constexpr uint32_t FQZ_FREVERSE = 16, FQZ_FREAD2 = 128;

static size_t store_array_h(uint8_t *out, const uint16_t *array, int size) {
    uint8_t tmp[2048];
    int i = 0, k = 0;
    for (uint32_t v = 0; i < size; v++) {
        int len = 0;
        while (i < size && array[i] == v) { i++; len++; }
        int r;
        do { r = std::min(len, 255); tmp[k++] = (uint8_t)r; len -= r; } while (r == 255);
    }
    if (k >= 2 && tmp[k - 1] == 0 && tmp[k - 2] == 255) k--;          // the reader stops once the table is full
    size_t o = 0; int last = -1;
    for (int j = 0; j < k;) {
        out[o] = tmp[j++];
        if (out[o] == last) {
            int n = 0;
            while (j < k && tmp[j] == last && n < 255) { j++; n++; }
            out[++o] = (uint8_t)n;
        } else last = out[o];
        o++;
    }
    return o;
}

// records a and b (same length) equal once each is put in coding orientation (ra / rb = reversed)?
static bool same_record(const uint8_t *a, bool ra, const uint8_t *b, bool rb, uint32_t len) {
    if (ra == rb) return memcmp(a, b, len) == 0;
    for (uint32_t j = 0; j < len; j++) if (a[j] != b[len - 1 - j]) return false;
    return true;
}

// Fills the image (tables for the context, qual -> code map in the qmap slot) and the stream's header bytes.  0 or -1.
static int build_encode_image(const uint8_t *in, uint32_t n, const hg_fqz_slice *sl, int strat, uint32_t *img, std::vector<uint8_t> &hdr) {
    memset(img, 0, IMG_WORDS * 4);
    if (strat < 0 || strat > 3 || !sl || !sl->len || sl->num_records == 0) return -1;
    const uint32_t nrec = sl->num_records;
    const bool have_flags = sl->flags != nullptr;
    uint64_t tot = 0;
    for (uint32_t r = 0; r < nrec; r++) { if (sl->len[r] == 0) return -1; tot += sl->len[r]; }
    if (tot != n) return -1;
    bool seen[256] = {false};
    for (uint32_t i = 0; i < n; i++) seen[in[i]] = true;
    uint32_t nsym = 0, max_q = 0;
    for (uint32_t i = 0; i < 256; i++) if (seen[i]) { nsym++; max_q = i; }
    bool fixed = true; size_t dups = 0;
    {
        size_t at = 0;
        for (uint32_t r = 0; r < nrec; r++) {
            const uint32_t len = sl->len[r];
            if (len != sl->len[0]) fixed = false;
            if (r && len == sl->len[r - 1] &&
                same_record(in + at, have_flags && (sl->flags[r] & FQZ_FREVERSE), in + at - len, have_flags && (sl->flags[r - 1] & FQZ_FREVERSE), len)) dups++;
            at += len;
        }
    }
    const bool do_sel = strat == 1 && have_flags, do_rev = have_flags, do_dedup = dups * 10 >= nrec && nrec > 1;
    const uint32_t gflags = (do_sel ? GF_MULTI : 0u) | (do_rev ? GF_REV : 0u), nparam = do_sel ? 2u : 1u;
    const int *PR = PRESET[strat];
    hdr.clear();
    { uint8_t t[5]; int k = 0; uint32_t v = n; do { t[k++] = v & 0x7f; v >>= 7; } while (v); while (k--) hdr.push_back((uint8_t)(t[k] | (k ? 0x80 : 0))); }
    hdr.push_back((uint8_t)FQZ_VERS); hdr.push_back((uint8_t)gflags);
    if (gflags & GF_MULTI) hdr.push_back((uint8_t)nparam);
    uint8_t *stab = (uint8_t *)(img + IMG_HEAD);
    for (uint32_t i = 0; i < 256; i++) stab[i] = (uint8_t)(i < nparam ? i : nparam - 1);
    uint32_t max_sym = 0;
    for (uint32_t s = 0; s < nparam; s++) {
        uint32_t *P = img + IMG_HEAD + IMG_STAB + s * IMG_PARAM;
        uint8_t *inv = (uint8_t *)(P + IMG_PSCAL);
        uint16_t *qtab = (uint16_t *)(P + IMG_PSCAL + IMG_QMAP), *dtab = qtab + 256, *ptab = dtab + 256;
        uint32_t pflags = (do_dedup ? PF_DEDUP : 0u) | (fixed ? PF_LEN : 0u) | (do_sel ? PF_SEL : 0u);
        uint32_t qbits = (uint32_t)PR[0], qshift = (uint32_t)PR[1], msym = max_q;
        uint8_t qmap[256];
        for (uint32_t i = 0; i < 256; i++) { inv[i] = (uint8_t)i; qtab[i] = (uint16_t)i; }
        if (nsym <= 8) {
            pflags |= PF_QMAP;
            uint32_t j = 0;
            for (uint32_t i = 0; i < 256; i++) if (seen[i]) { qmap[j] = (uint8_t)i; inv[i] = (uint8_t)j; j++; }
            msym = nsym;
            qshift = nsym <= 2 ? 1 : nsym <= 4 ? 2 : 3;
        }
        if (qbits > 12) qbits = 12;
        const int pbits = PR[2], pshift = PR[3] < 0 ? (sl->len[0] > 511 ? 3 : sl->len[0] > 255 ? 2 : sl->len[0] > 127 ? 1 : 0) : PR[3];
        if (pbits > 0) { pflags |= PF_PTAB; for (uint32_t i = 0; i < 1024; i++) ptab[i] = (uint16_t)std::min<uint32_t>(i >> pshift, (1u << pbits) - 1u); }
        const int dbits = PR[4], dshift = PR[5];
        if (dbits > 0) { pflags |= PF_DTAB; for (uint32_t i = 0; i < 256; i++) dtab[i] = (uint16_t)std::min<uint32_t>(i >> dshift, (1u << dbits) - 1u); }
        if (strat == 2) { pflags |= PF_QTAB; for (uint32_t i = 0; i < 256; i++) qtab[i] = (uint16_t)(i < 32 ? i : std::min<uint32_t>(32 + (i - 32) / 4, 63)); }
        P[0] = 0; P[1] = pflags; P[2] = (1u << qbits) - 1u; P[3] = qshift;
        P[4] = (uint32_t)PR[6]; P[5] = (uint32_t)PR[7]; P[6] = (uint32_t)PR[8]; P[7] = (uint32_t)PR[9];
        hdr.push_back(0); hdr.push_back(0); hdr.push_back((uint8_t)pflags); hdr.push_back((uint8_t)msym);
        hdr.push_back((uint8_t)(qbits << 4 | qshift)); hdr.push_back((uint8_t)(P[4] << 4 | P[5])); hdr.push_back((uint8_t)(P[6] << 4 | P[7]));
        if (pflags & PF_QMAP) hdr.insert(hdr.end(), qmap, qmap + msym);
        uint8_t tmp[4096];
        if (pflags & PF_QTAB) { const size_t k = store_array_h(tmp, qtab, 256); hdr.insert(hdr.end(), tmp, tmp + k); }
        if (pflags & PF_PTAB) { const size_t k = store_array_h(tmp, ptab, 1024); hdr.insert(hdr.end(), tmp, tmp + k); }
        if (pflags & PF_DTAB) { const size_t k = store_array_h(tmp, dtab, 256); hdr.insert(hdr.end(), tmp, tmp + k); }
        max_sym = std::max(max_sym, msym);
    }
    img[0] = gflags; img[1] = nparam; img[2] = nparam - 1u; img[3] = max_sym + 1u; img[4] = (uint32_t)hdr.size(); img[5] = n; img[6] = nrec;
    img[7] = have_flags ? 1u : 0u;
    return 0;
}

// ---- device ----------------------------------------------------------------------------------------------------------
struct ParamRegs { uint32_t context, pflags, qmask, qshift, qloc, sloc, ploc, dloc; const uint32_t *base; };
__device__ __forceinline__ void load_param(ParamRegs &R, const uint32_t *img, uint32_t x) {
    const uint32_t *P = img + IMG_HEAD + IMG_STAB + x * IMG_PARAM;
    // (wave-uniform by construction; saying so moves the flag tests and shifts they feed to the scalar unit)
    R.context = hg::uni(P[0]); R.pflags = hg::uni(P[1]); R.qmask = hg::uni(P[2]); R.qshift = hg::uni(P[3]);
    R.qloc = hg::uni(P[4]); R.sloc = hg::uni(P[5]); R.ploc = hg::uni(P[6]); R.dloc = hg::uni(P[7]);
    R.base = P;
}
// Parameter set x of the stream in one of the two LDS slots: c0 / c1 = the sets they hold, victim = the slot replaced next.  gimg = the stream's base
// image in global memory (sets 0, 1), over = its further sets.  Wave-uniform throughout.
__device__ __forceinline__ void select_param(ParamRegs &R, uint32_t *img, const uint32_t *gimg, const uint32_t *over, uint32_t x, uint32_t &c0, uint32_t &c1,
                                             uint32_t &victim, int lane) {
    uint32_t slot;
    if (x == c0) slot = 0; else if (x == c1) slot = 1;
    else {
        slot = victim; victim ^= 1u;
        const uint32_t *src = x < (uint32_t)HGQ_MAX_PARAM ? gimg + IMG_HEAD + IMG_STAB + x * IMG_PARAM : over + (size_t)(x - HGQ_MAX_PARAM) * IMG_PARAM;
        uint32_t *dst = img + IMG_HEAD + IMG_STAB + slot * IMG_PARAM;
        hg::wave_sync();
        for (uint32_t i = (uint32_t)lane; i < IMG_PARAM; i += 64) dst[i] = src[i];
        hg::wave_sync();
        if (slot) c1 = x; else c0 = x;
    }
    load_param(R, img, slot);
}
struct State { uint32_t qctx, p, delta, prevq, s; };
__device__ __forceinline__ uint32_t update_ctx(const ParamRegs &R, State &st, uint32_t q) {
    const uint16_t *qtab = (const uint16_t *)(R.base + IMG_PSCAL + IMG_QMAP), *dtab = qtab + 256, *ptab = dtab + 256;
    const uint32_t tq = qtab[q], tp = ptab[st.p < 1023u ? st.p : 1023u], td = dtab[st.delta < 255u ? st.delta : 255u];   // three independent LDS reads
    uint32_t c = R.context;
    st.qctx = (st.qctx << R.qshift) + tq;
    c += (st.qctx & R.qmask) << R.qloc;
    if (R.pflags & PF_PTAB) c += tp << R.ploc;
    if (R.pflags & PF_DTAB) { c += td << R.dloc; st.delta += st.prevq != q; st.prevq = q; }
    if (R.pflags & PF_SEL) c += st.s << R.sloc;
    st.p--;
    return c & (CTX_SIZE - 1u);
}
__device__ __forceinline__ void lds_model_init(uint32_t *m, uint32_t n, int lane) {
    for (uint32_t i = (uint32_t)lane; i <= n; i += 64) m[i] = i ? ((1u << 8) | (i - 1u)) : n;
}

// One stream, start to finish (the kernel has set up its image and models).  FAST: see fqz_decode_kernel -- a template parameter rather than a branch inside the
// symbol loop: with both forms in one loop the compiler's wait insertion merges their pending loads (the general routine requests up to four 64-entry pieces
// of a model and may leave some unread), and the fast form then carries an s_waitcnt vmcnt(0) per quality for registers it never loads.
template <bool FAST>
__device__ __forceinline__ int decode_stream(const uint8_t *__restrict__ in, const hg_stream_desc &d, uint32_t *img, uint32_t *mdl, uint32_t *gq, const uint32_t *gimg,
                                             const uint32_t *over, uint8_t *o, uint32_t k, int lane) {
    const uint32_t gflags = hg::uni(img[0]), max_sel = hg::uni(img[2]), ns = hg::uni(img[3]), data_off = hg::uni(img[4]), ulen = hg::uni(img[5]);
    const uint8_t *stab = (const uint8_t *)(img + IMG_HEAD);
    uint32_t pc0 = 0, pc1 = 1, pvictim = 0;
    const uint32_t qw = ns + 1u;
    uint32_t cur = 0;                                              // fast form: the model of context `last` (valid iff have_cur)
    bool have_cur = false;
    hga::Decoder D;
    D.start(in + d.in_off + data_off, d.in_len - data_off, lane);
    ParamRegs R; load_param(R, img, 0);
    State st = {0, 0, 0, 0, 0};
    uint32_t i = 0, last = 0, last_len = 0, keep = 0, prev_rev = 0;
    bool first_len = true;
    int err = 0;
    auto lsym = [&](uint32_t base, uint32_t n) { return D.template symbol_lean<true>(mdl, mdl, base + 1u, n, base, lane); };
    auto flush_tail = [&]() { if ((uint32_t)lane < (i & 63u)) o[(i & ~63u) + (uint32_t)lane] = (uint8_t)keep; wave_sync(); };
    auto reload_tail = [&]() { wave_sync(); if ((uint32_t)lane < (i & 63u)) keep = o[(i & ~63u) + (uint32_t)lane]; hg::wait_vm0(); };
    uint32_t rec_start = 0, rec_len = 0, rec_rev = 0;
#ifdef HG_FQZ_PROFILE
    unsigned long long fq_acc[6] = {0, 0, 0, 0, 0, 0}, fq_last = __builtin_amdgcn_s_memtime();
#endif
    while (i < ulen) {
        FQ_T(4);
        if (st.p == 0) {
            uint32_t s = 0;
            if (max_sel > 0) { s = lsym(M_SEL, max_sel + 1u); if (D.err) break; }
            st.s = s;
            select_param(R, img, gimg, over, stab[s], pc0, pc1, pvictim, lane);
            uint32_t len;
            if (!(R.pflags & PF_LEN) || first_len) {
                len = lsym(M_LEN, 256);
                len |= lsym(M_LEN + 257, 256) << 8;
                len |= lsym(M_LEN + 2 * 257, 256) << 16;
                len |= lsym(M_LEN + 3 * 257, 256) << 24;
                if (D.err) break;
                first_len = false; last_len = len;
            } else len = last_len;
            if (len == 0 || len > ulen - i) { err = 1; break; }
            uint32_t rv = 0;
            if (gflags & GF_REV) { rv = lsym(M_REV, 2); if (D.err) break; }
            rec_start = i; rec_len = len; rec_rev = rv;
            if (R.pflags & PF_DEDUP) {
                const uint32_t dup = lsym(M_DUP, 2);
                if (D.err) break;
                if (dup) {
                    if (i < len) { err = 1; break; }
                    flush_tail();
                    // the predecessor sits in o[i - len, i) in its FINAL orientation: mirrored iff the two reverse flags differ
                    const bool mirror = rv != prev_rev;
                    for (uint32_t b = (uint32_t)lane; b < len; b += 64) o[i + b] = mirror ? o[i - 1u - b] : o[i - len + b];
                    i += len;
                    prev_rev = rv;
                    reload_tail();
                    continue;
                }
            }
            st.p = len; st.delta = 0; st.qctx = 0; st.prevq = 0;
            last = R.context;
            have_cur = false;
        }
        uint32_t Q;
        FQ_T(0);
        const size_t mb = (size_t)last * qw;
        uint32_t q;
        if (FAST) {
            // (every read of a model is 64 lanes wide, whatever its length: the words past it -- the next model's -- are masked where it matters.  A
            //  conditional read zeroes the other lanes first, and that write to a register with a fetch still outstanding costs a vmcnt(0) per quality)
            if (!have_cur) { cur = gq[mb + (uint32_t)lane]; hg::wait_vm0(); }
            FQ_T(1);
            // the part of the next context that does not depend on this quality, then the whole of it per candidate (lane j: entry j is the one decoded)
            const uint16_t *qtab = (const uint16_t *)(R.base + IMG_PSCAL + IMG_QMAP), *dtab = qtab + 256, *ptab = dtab + 256;
            const uint32_t tp = hg::uni(ptab[st.p < 1023u ? st.p : 1023u]), td = hg::uni(dtab[st.delta < 255u ? st.delta : 255u]);
            const uint32_t tqv = qtab[cur & 0xffu];
            uint32_t cb = R.context;
            if (R.pflags & PF_PTAB) cb += tp << R.ploc;
            if (R.pflags & PF_DTAB) cb += td << R.dloc;
            if (R.pflags & PF_SEL) cb += st.s << R.sloc;
            const uint32_t qnv = (st.qctx << R.qshift) + tqv;
            const uint32_t cv = (cb + ((qnv & R.qmask) << R.qloc)) & (CTX_SIZE - 1u);
            uint32_t pf[QPF];
#pragma unroll
            for (int c = 0; c < QPF; c++) {
                // (issued HERE, by hand: as plain loads the compiler sinks them towards their only use, after the coder step -- and it reuses the registers of
                //  the three that are never read, which costs a wait each.  The one s_waitcnt that covers them is wait_vm0() below.)
                const uint32_t *src = gq + (__umul24(rl(cv, (uint32_t)c), qw) + (uint32_t)lane);   // (a slot is 65 536 x <= 64 words: 32-bit index)
                asm volatile("global_load_dword %0, %1, off" : "=v"(pf[c]) : "v"(src) : "memory");
            }
            // coder step on the register model
            const uint32_t tot = rl(cur, ns);
            const uint32_t e = (uint32_t)lane < ns ? cur : 0u;
            const uint32_t r = hga::udiv_small_divisor(D.range, tot);
            const uint32_t incl = hg::wave_incl_scan_dpp(e >> 8);       // lanes >= ns carry the total
            const unsigned long long hit = __ballot(incl * r > D.code);   // incl > code / r  <=>  incl r > code;  incl r <= tot r <= range: no overflow
            if (!hit) { hg::wait_vm0(); D.err = 1; break; }             // code / r >= tot: not a valid stream (the four hand-issued loads must have landed before their registers are anyone else's)
            const uint32_t l = (uint32_t)__builtin_ctzll(hit);
            const uint32_t ex = rl(e, l), f = ex >> 8;
            D.code -= (rl(incl, l) - f) * r; D.range = r * f;
            while (D.range < hga::TOP) { D.code = (D.code << 8) | D.in.next(lane); D.range <<= 8; }
            Q = ex & 0xffu;
            bool changed = false;                                        // lanes whose word of the model is new
            if (tot + hga::STEP > hga::MAX_FREQ) {                      // halve every frequency (rare): the general routine on the stored model
                hga::model_update<false>(gq + mb, gq + mb, 0u, ns, ns, tot, l, ex, false, 0u, lane);
                cur = gq[mb + (uint32_t)lane];
                hg::wait_vm0();
            } else {
                const uint32_t nex = ex + (hga::STEP << 8), ep = l ? rl(e, l - 1u) : 0xffffffffu;
                const bool swap = (nex >> 8) > (ep >> 8);                // sorted by frequency: at most one step towards the front
                const bool w0 = (uint32_t)lane == l, w1 = swap && (uint32_t)lane + 1u == l, wt = (uint32_t)lane == ns;
                if (w0) cur = swap ? ep : nex;
                if (w1) cur = nex;
                if (wt) cur = tot + hga::STEP;
                changed = w0 || w1 || wt;
            }
            FQ_T(2);
            q = ((const uint8_t *)(R.base + IMG_PSCAL))[Q];
            st.qctx = (st.qctx << R.qshift) + rl(tqv, l);
            if (R.pflags & PF_DTAB) { st.delta += st.prevq != Q; st.prevq = Q; }
            st.p--;
            const uint32_t next = rl(cv, l), upd = cur;
            // the one wait for memory of a quality: the models requested before the coder step.  The four registers are operands of the wait, so that no
            // compiler can schedule the selects below above it (the loads were issued by hand: its own waitcnt bookkeeping does not know them)
            static_assert(QPF == 4, "the wait names the four prefetch registers");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]) : : "memory");
            {                                                            // (same context again: `cur` is already its updated model)
                const uint32_t p01 = (l & 1u) ? pf[1] : pf[0], p23 = (l & 1u) ? pf[3] : pf[2], psel = (l & 2u) ? p23 : p01;
                const bool same = next == last;
                cur = same ? cur : psel;
                have_cur = same || l < (uint32_t)QPF;
            }
            // the store goes LAST: vmcnt counts loads and stores together, so any wait after a store waits for its round trip as well -- the next such wait
            // is a whole coder step away
            if (changed) gq[mb + (uint32_t)lane] = upd;
            last = next;
        } else {
            FQ_T(1);
            Q = D.template symbol_lean<false>(gq + mb, gq + mb, 0u, ns, ns, lane);
            FQ_T(2);
            if (D.err) break;
            q = ((const uint8_t *)(R.base + IMG_PSCAL))[Q];
            last = update_ctx(R, st, Q);
        }
#ifdef HG_FQZ_PROFILE
        asm volatile("" :: "v"(last), "v"(q));                       // the phase ends when the next context is known
#endif
        FQ_T(3);
        if ((uint32_t)lane == (i & 63u)) keep = q;
        if ((i & 63u) == 63u) o[i - 63u + (uint32_t)lane] = (uint8_t)keep;       // 64 qualities per store
        i++;
        if (st.p == 0 && (gflags & GF_REV)) {                                // the record is complete
            if (rec_rev) {
                flush_tail();
                for (uint32_t b = (uint32_t)lane; b < rec_len / 2u; b += 64) {
                    const uint8_t x = o[rec_start + b], y = o[rec_start + rec_len - 1u - b];
                    o[rec_start + b] = y; o[rec_start + rec_len - 1u - b] = x;
                }
                reload_tail();
            }
            prev_rev = rec_rev;
        }
    }
    if (!err && !D.err && (ulen & 63u) && (uint32_t)lane < (ulen & 63u) && i == ulen) o[(ulen & ~63u) + (uint32_t)lane] = (uint8_t)keep;
    if (D.err || D.in.overrun || st.p != 0 || i != ulen) err = 1;
#ifdef HG_FQZ_PROFILE
    FQ_T(4);
    if (k < 4 && lane == 0)
        printf("[fqz-profile] stream %u: %u qualities, ns %u, fast form %d; ticks per quality: record %.1f, model fetch %.1f, coder step %.1f, next context %.1f, rest %.1f\n", k, ulen,
               ns, (int)FAST, (double)fq_acc[0] / ulen, (double)fq_acc[1] / ulen, (double)fq_acc[2] / ulen, (double)fq_acc[3] / ulen, (double)fq_acc[4] / ulen);
#endif
    return err;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64)
void fqz_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint32_t *__restrict__ images,
                       const uint32_t *__restrict__ overflow, uint32_t nstreams, uint8_t *out, int32_t *status, uint32_t *gscratch, unsigned long long slot_words,
                       int qc_on) {
    // The quality models of a stream (65 536 contexts x (ns + 1) words, ~11 MB; entries first, the total LAST) live in global memory.  Per quality the chain was
    // context -> model fetch (an L2 / MALL round trip, ~650 clocks) -> coder step (~950) -> next context (~90): r05_fqz_decode_phases.txt.  The fast form
    // (alphabets up to 63 symbols, i.e. every real quality block) keeps the model of the current context in one VGPR (lane j = entry j, lane n = total), and:
    //   * BEFORE the coder step it computes the next context for every symbol of the model at once (lane j: "if entry j is decoded") and requests the models of
    //     the first QPF candidates -- the list is sorted by frequency, so these are the likely ones; the fetch then overlaps the coder step;
    //   * the symbol search multiplies the cumulative frequencies by r and compares with the code (one multiply per lane) instead of dividing the code by r;
    //   * the update happens in the register, and only the two or three words that changed are stored.
    // An LDS cache of recently used models (tags + write-back) was tried first and dropped: hits were 14 .. 80 % depending on the parameter set, and the
    // stream with the fewest hits -- which sets the time of a batch -- got slower (same profile file).
    __shared__ uint32_t pool[WAVES][POOLW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * WAVES + wv;
    uint32_t *gq = gscratch + (size_t)slot * slot_words;
    uint32_t *mdl = pool[wv], *img = pool[wv] + ((M_WORDS + 3) & ~3u);
    for (uint32_t k = slot; k < nstreams; k += gridDim.x * WAVES) {
        const hg_stream_desc d = desc[k];
        for (uint32_t i = (uint32_t)lane; i < IMG_WORDS; i += 64) img[i] = images[(size_t)d.scratch_off * IMG_WORDS + i];
        wave_sync();
        const uint32_t max_sel = img[2], ns = img[3];
        const uint32_t *gimg = images + (size_t)d.scratch_off * IMG_WORDS, *over = overflow + img[6];
        uint8_t *o = out + d.out_off;
        // models
        {
            const uint32_t w = ns + 1u;
            uint32_t r = (uint32_t)lane % w;                           // word i of the slot is word (i mod w) of its model
            const uint32_t step = 64u % w;
            for (size_t i = (size_t)lane; i < (size_t)CTX_SIZE * w; i += 64) {
                gq[i] = r < ns ? ((1u << 8) | r) : ns;                  // (decoder layout: the total is the model's last word)
                r += step; if (r >= w) r -= w;
            }
            for (int j = 0; j < 4; j++) lds_model_init(mdl + M_LEN + j * 257, 256, lane);
            lds_model_init(mdl + M_REV, 2, lane); lds_model_init(mdl + M_DUP, 2, lane);
            lds_model_init(mdl + M_SEL, max_sel + 1u, lane);
            wave_sync();
        }
        const bool fast = qc_on && ns + 1u <= 64u;
        const int err = fast ? decode_stream<true>(in, d, img, mdl, gq, gimg, over, o, k, lane) : decode_stream<false>(in, d, img, mdl, gq, gimg, over, o, k, lane);
        status[k] = err ? -1 : 0;                                     // every lane stores the same word
        wave_sync();
    }
}

// Encoder: the same chain run forwards.  rec_len / rec_flags: the records of all streams back to back (desc.reserved = index of
// the stream's first record).  The qmap slot of the image holds the INVERSE map (quality -> code).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64)
void fqz_encode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint32_t *__restrict__ images,
                       const uint32_t *__restrict__ rec_len, const uint32_t *__restrict__ rec_flags, uint32_t nstreams, uint8_t *out,
                       uint32_t *out_len, uint32_t *gscratch, unsigned long long slot_words) {
    __shared__ uint32_t pool[WAVES][POOLW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * WAVES + wv;
    uint32_t *gq = gscratch + (size_t)slot * slot_words;
    uint32_t *mdl = pool[wv], *img = pool[wv] + ((M_WORDS + 3) & ~3u);
    for (uint32_t k = slot; k < nstreams; k += gridDim.x * WAVES) {
        const hg_stream_desc d = desc[k];
        for (uint32_t i = (uint32_t)lane; i < IMG_WORDS; i += 64) img[i] = images[(size_t)d.scratch_off * IMG_WORDS + i];
        wave_sync();
        const uint32_t gflags = img[0], max_sel = img[2], ns = img[3], nrec = img[6], have_flags = img[7];
        const uint8_t *stab = (const uint8_t *)(img + IMG_HEAD);
        const uint8_t *src = in + d.in_off;
        const uint32_t *lens = rec_len + d.reserved, *flg = rec_flags + d.reserved;
        {
            const uint32_t w = ns + 1u;
            uint32_t r = (uint32_t)lane % w;
            const uint32_t step = 64u % w;
            for (size_t i = (size_t)lane; i < (size_t)CTX_SIZE * w; i += 64) {
                gq[i] = r ? ((1u << 8) | (r - 1u)) : ns;
                r += step; if (r >= w) r -= w;
            }
            for (int j = 0; j < 4; j++) lds_model_init(mdl + M_LEN + j * 257, 256, lane);
            lds_model_init(mdl + M_REV, 2, lane); lds_model_init(mdl + M_DUP, 2, lane);
            lds_model_init(mdl + M_SEL, max_sel + 1u, lane);
            wave_sync();
        }
        hga::Encoder E;
        E.start(out + d.out_off);
        auto lsym = [&](uint32_t base, uint32_t n, uint32_t sym) { E.template symbol_lean<true>(mdl, mdl, base + 1u, n, base, sym, lane); };
        ParamRegs R; load_param(R, img, 0);
        State st = {0, 0, 0, 0, 0};
        bool first_len = true;
        uint32_t at = 0, prev_len = 0, prev_rev = 0;
        for (uint32_t r = 0; r < nrec; r++) {
            const uint32_t len = lens[r], fl = have_flags ? flg[r] : 0u;
            const uint32_t rv = (gflags & GF_REV) && (fl & FQZ_FREVERSE) ? 1u : 0u, s = max_sel && (fl & FQZ_FREAD2) ? 1u : 0u;
            if (max_sel > 0) lsym(M_SEL, max_sel + 1u, s);
            st.s = s;
            load_param(R, img, stab[s]);
            if (!(R.pflags & PF_LEN) || first_len) {
                lsym(M_LEN, 256, len & 0xffu); lsym(M_LEN + 257, 256, (len >> 8) & 0xffu);
                lsym(M_LEN + 2 * 257, 256, (len >> 16) & 0xffu); lsym(M_LEN + 3 * 257, 256, len >> 24);
                first_len = false;
            }
            if (gflags & GF_REV) lsym(M_REV, 2, rv);
            if (R.pflags & PF_DEDUP) {
                uint32_t dup = 0;
                if (r && prev_len == len) {                            // equal to the predecessor, both in coding orientation?
                    dup = 1;
                    for (uint32_t b = 0; b < len && dup; b += 64) {
                        const uint32_t j = b + (uint32_t)lane;
                        const bool ne = j < len && src[at + j] != (rv == prev_rev ? src[at - len + j] : src[at - 1u - j]);
                        if (__ballot(ne)) dup = 0;
                    }
                }
                lsym(M_DUP, 2, dup);
                if (dup) { at += len; prev_len = len; prev_rev = rv; continue; }
            }
            st.p = len; st.delta = 0; st.qctx = 0; st.prevq = 0;
            uint32_t last = R.context, win = 0;
            const uint8_t *inv = (const uint8_t *)(R.base + IMG_PSCAL);
            for (uint32_t j = 0; j < len; j++) {
                if ((j & 63u) == 0) { const uint32_t jj = j + (uint32_t)lane; win = jj < len ? src[rv ? at + len - 1u - jj : at + jj] : 0u; }
                const uint32_t q = inv[rl(win, j & 63u)];
                const size_t mb = (size_t)last * (ns + 1u);
                E.template symbol_lean<false>(gq + mb, gq + mb, 1u, ns, 0u, q, lane);
                last = update_ctx(R, st, q);
            }
            at += len; prev_len = len; prev_rev = rv;
        }
        out_len[k] = E.finish(lane);                                   // every lane stores the same word
        wave_sync();
    }
}


// ================================================================================ the ENCODER in two phases (round 5)
// An encoder knows every quality and therefore every CONTEXT before it codes anything: the 65 536 adaptive models only interact through the coder's
// registers.  So, as in arith_enc2.hip: (1) events_kernel -- one workgroup per block, one THREAD per record -- walks each record's context state machine (it
// restarts with every record) and writes the block's events in coding order: (model, symbol) with model = the 16-bit quality context, or one of the seven
// record-level models (selector, four length bytes, reverse, duplicate); record r's first event number is a prefix sum over the records.  (2) sort_kernel,
// one wavefront per block: a stable LSD radix sort of the events by model (8 + 9 bits, the lanes of a step ranking themselves with one ballot per key bit).
// (3) models_kernel: the sorted events cut every ENC2_CUT positions; wave t takes the models that START in its stretch, each in REGISTERS (range_enc2_dev.h
// RegModel: no model ever travels to HBM and back -- the one-pass kernel waits ~1 us for its model at every quality), and leaves an 8-byte record per event
// at the event's number.  (4) the scalar coder pass over the records (hga2::code_kernel).  Byte-identical to the one-pass encoder and the oracle.
constexpr uint32_t K_SEL = 65536, K_LEN = 65537, K_REV = 65541, K_DUP = 65542;
constexpr uint32_t ENC2_CUT = 1024;
struct Enc2Stream {            // per block, device: where its arrays lie (byte offsets into the work buffer) and what the events kernel found
    uint64_t key0, pay0, key1, pay1, rec, ebase;
    uint32_t ecap, nrec_first;   // nrec_first: index of the block's first record in the call's record arrays
};

__global__ __launch_bounds__(256)
void fqz_events_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, const uint32_t *__restrict__ images, const uint32_t *__restrict__ rec_len,
                       const uint32_t *__restrict__ rec_flags, const uint32_t *__restrict__ rec_off, const Enc2Stream *__restrict__ streams, uint8_t *work, uint32_t *info) {
    __shared__ uint32_t img[IMG_WORDS];
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry_s;
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    const hg_stream_desc d = desc[k];
    const Enc2Stream S = streams[k];
    for (uint32_t i = tid; i < IMG_WORDS; i += 256) img[i] = images[(size_t)d.scratch_off * IMG_WORDS + i];
    if (tid == 0) carry_s = 0;
    __syncthreads();
    const uint32_t gflags = img[0], max_sel = img[2], nrec = img[6], have_flags = img[7];
    const uint8_t *stab = (const uint8_t *)(img + IMG_HEAD);
    const uint8_t *src = in + d.in_off;
    const uint32_t *lens = rec_len + S.nrec_first, *flg = rec_flags + S.nrec_first, *offs = rec_off + S.nrec_first;
    uint32_t *ebase = (uint32_t *)(work + S.ebase);
    uint32_t *key0 = (uint32_t *)(work + S.key0), *pay0 = (uint32_t *)(work + S.pay0);
    // what record r codes before its qualities, and whether it is a duplicate of its predecessor (both in coding orientation)
    auto record = [&](uint32_t r, uint32_t &len, uint32_t &rv, uint32_t &s, uint32_t &pflags, bool &coded_len, bool &dup) {
        len = lens[r];
        const uint32_t fl = have_flags ? flg[r] : 0u;
        rv = (gflags & GF_REV) && (fl & FQZ_FREVERSE) ? 1u : 0u; s = max_sel && (fl & FQZ_FREAD2) ? 1u : 0u;
        pflags = img[IMG_HEAD + IMG_STAB + (uint32_t)stab[s] * IMG_PARAM + 1u];
        coded_len = !(pflags & PF_LEN) || r == 0u;
        dup = false;
        if ((pflags & PF_DEDUP) && r && lens[r - 1u] == len) {
            const uint32_t pfl = have_flags ? flg[r - 1u] : 0u, prv = (gflags & GF_REV) && (pfl & FQZ_FREVERSE) ? 1u : 0u, at = offs[r];
            dup = true;
            for (uint32_t j = 0; j < len; j++) if (src[at + j] != (rv == prv ? src[at - len + j] : src[at - 1u - j])) { dup = false; break; }
        }
    };
    // ---- pass 1: events per record, exclusive prefix sums (chunks of 256 records)
    for (uint32_t r0 = 0; r0 < nrec; r0 += 256) {
        const uint32_t r = r0 + tid;
        uint32_t nev = 0;
        if (r < nrec) {
            uint32_t len, rv, s, pflags; bool cl, dup;
            record(r, len, rv, s, pflags, cl, dup);
            nev = (max_sel ? 1u : 0u) + (cl ? 4u : 0u) + ((gflags & GF_REV) ? 1u : 0u) + ((pflags & PF_DEDUP) ? 1u : 0u) + (dup ? 0u : len);
        }
        part[tid] = nev;
        __syncthreads();
        for (uint32_t st = 1; st < 256; st <<= 1) {                 // Hillis-Steele over the chunk
            const uint32_t v = tid >= st ? part[tid - st] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (r < nrec) ebase[r] = carry_s + part[tid] - nev;
        __syncthreads();
        if (tid == 255) carry_s += part[255];
        __syncthreads();
    }
    if (tid == 0) { info[2 * k] = 0; info[2 * k + 1] = carry_s; }    // Arith2pInfo { m, nevents }: what the coder pass reads
    // ---- pass 2: the events
    for (uint32_t r = tid; r < nrec; r += 256) {
        uint32_t len, rv, s, pflags; bool cl, dup;
        record(r, len, rv, s, pflags, cl, dup);
        uint32_t e = ebase[r];
        auto ev = [&](uint32_t key, uint32_t sym) { key0[e] = key; pay0[e] = e << 8 | sym; e++; };
        if (max_sel) ev(K_SEL, s);
        if (cl) { ev(K_LEN, len & 0xffu); ev(K_LEN + 1u, (len >> 8) & 0xffu); ev(K_LEN + 2u, (len >> 16) & 0xffu); ev(K_LEN + 3u, len >> 24); }
        if (gflags & GF_REV) ev(K_REV, rv);
        if (pflags & PF_DEDUP) { ev(K_DUP, dup ? 1u : 0u); if (dup) continue; }
        ParamRegs R; load_param(R, img, stab[s]);
        State st = {0, len, 0, 0, s};
        const uint8_t *inv = (const uint8_t *)(R.base + IMG_PSCAL);
        const uint32_t at = offs[r];
        uint32_t last = R.context;
        for (uint32_t j = 0; j < len; j++) {
            const uint32_t q = inv[src[rv ? at + len - 1u - j : at + j]];
            ev(last, q);
            last = update_ctx(R, st, q);
        }
    }
}

// one LSD pass: events (key, pay) of [0, n) from `in` to `out`, stable, by BITS bits of the key at `shift`; cnt / cur: 1 << BITS LDS words each
template <int BITS>
__device__ __forceinline__ void radix_pass(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ pin, uint32_t *kout, uint32_t *pout, uint32_t n, uint32_t shift,
                                           uint32_t *cnt, uint32_t *cur, int lane) {
    constexpr uint32_t NB = 1u << BITS;
    const unsigned long long lt = hga2::lanes_below(lane);
    for (uint32_t i = (uint32_t)lane; i < NB; i += 64) cnt[i] = 0u;
    wave_sync();
    for (int place = 0; place < 2; place++) {
        uint32_t nk = (uint32_t)lane < n ? kin[lane] : 0u, np = place && (uint32_t)lane < n ? pin[lane] : 0u;
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            const uint32_t key = nk, pay = np, p = i0 + (uint32_t)lane;
            { const uint32_t q = p + 64u; if (q < n) { nk = kin[q]; if (place) np = pin[q]; } }
            const bool in = p < n;
            const uint32_t dgt = (key >> shift) & (NB - 1u);
            const unsigned long long m = hga2::match_bits<BITS>(dgt, in);
            const uint32_t rank = (uint32_t)__popcll(m & lt), c = (uint32_t)__popcll(m);
            if (!place) { if (in && rank == 0u) cnt[dgt] += c; }
            else {
                const uint32_t base = in ? cur[dgt] : 0u;
                if (in && rank == 0u) cur[dgt] = base + c;
                if (in) { kout[base + rank] = key; pout[base + rank] = pay; }
            }
        }
        wave_sync();
        if (!place) {                                               // exclusive prefix sums of the counts: lane l takes NB / 64 consecutive entries
            constexpr uint32_t PER = NB / 64u;
            uint32_t a[PER], sum = 0;
#pragma unroll
            for (uint32_t q = 0; q < PER; q++) { a[q] = cnt[PER * (uint32_t)lane + q]; sum += a[q]; }
            uint32_t ex = hg::wave_incl_scan_dpp(sum) - sum;
#pragma unroll
            for (uint32_t q = 0; q < PER; q++) { cur[PER * (uint32_t)lane + q] = ex; ex += a[q]; }
            wave_sync();
        }
    }
}

__global__ __launch_bounds__(64)
void fqz_sort_kernel(const Enc2Stream *__restrict__ streams, uint8_t *work, const uint32_t *__restrict__ info) {
    __shared__ uint32_t cnt[512], cur[512];
    const int lane = threadIdx.x;
    const uint32_t k = blockIdx.x, n = info[2 * k + 1];
    const Enc2Stream S = streams[k];
    uint32_t *key0 = (uint32_t *)(work + S.key0), *pay0 = (uint32_t *)(work + S.pay0), *key1 = (uint32_t *)(work + S.key1), *pay1 = (uint32_t *)(work + S.pay1);
    radix_pass<8>(key0, pay0, key1, pay1, n, 0u, cnt, cur, lane);
    radix_pass<9>(key1, pay1, key0, pay0, n, 8u, cnt, cur, lane);
}

// the events of the model that starts at sorted position i (key `key`, alphabet n_alpha): codes them, returns the position where the next model starts
template <int EPL>
__device__ __forceinline__ uint32_t enc2_model(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ pays, uint32_t i, uint32_t n, uint32_t key, uint32_t n_alpha, uint2 *R, int lane) {
    hga2::RegModel<EPL> G; G.init(n_alpha, lane);
    for (;;) {
        const uint32_t p = i + (uint32_t)lane;
        const uint32_t kv = p < n ? keys[p] : ~0u, pv = p < n ? pays[p] : 0u;
        const unsigned long long other = __ballot(kv != key);
        const uint32_t nn = other ? (uint32_t)__builtin_ctzll(other) : 64u;
        for (uint32_t j = 0; j < nn; j++) { const uint32_t w = rl(pv, j); G.step(w & 0xffu, R, w >> 8, lane); }
        i += nn;
        if (nn < 64u) return i;
    }
}

__global__ __launch_bounds__(256)
void fqz_models_kernel(const hg_stream_desc *__restrict__ desc, const uint32_t *__restrict__ images, const Enc2Stream *__restrict__ streams, uint8_t *work, const uint32_t *__restrict__ info) {
    const int lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.y, t = blockIdx.x * 4u + (threadIdx.x >> 6), n = info[2 * k + 1];
    if ((unsigned long long)t * ENC2_CUT >= n) return;
    const Enc2Stream S = streams[k];
    const uint32_t *keys = (const uint32_t *)(work + S.key0), *pays = (const uint32_t *)(work + S.pay0);
    uint2 *R = (uint2 *)(work + S.rec);
    const uint32_t *img = images + (size_t)desc[k].scratch_off * IMG_WORDS;
    const uint32_t max_sel = img[2], ns = img[3];
    uint32_t i = t * ENC2_CUT;
    const uint32_t stop = (t + 1u) * ENC2_CUT;                          // models starting at or behind this position belong to the next stretches
    if (t) {
        // the first model START in [i, ...): nothing when one model covers the stretch and beyond
        const uint32_t last = stop < n ? stop - 1u : n - 1u;
        if (keys[i - 1u] == keys[last]) return;
        for (;;) {
            const uint32_t p = i + (uint32_t)lane;
            const uint32_t a = p < n ? keys[p] : ~0u, b = keys[p < n ? p - 1u : n - 1u];
            const unsigned long long st = __ballot(p < n && a != b);
            if (st) { i += (uint32_t)__builtin_ctzll(st); break; }
            i += 64u;
            if (i >= n) return;
        }
        if (i >= stop) return;
    }
    while (i < n && i < stop) {
        const uint32_t key = hg::uni(keys[i]);
        if (key < CTX_SIZE) i = ns > 128u ? enc2_model<4>(keys, pays, i, n, key, ns, R, lane) : ns > 64u ? enc2_model<2>(keys, pays, i, n, key, ns, R, lane) : enc2_model<1>(keys, pays, i, n, key, ns, R, lane);
        else if (key >= K_LEN && key < K_LEN + 4u) i = enc2_model<4>(keys, pays, i, n, key, 256u, R, lane);
        else i = enc2_model<1>(keys, pays, i, n, key, key == K_SEL ? max_sel + 1u : 2u, R, lane);
    }
}

}  // namespace hgq

namespace hg {
// images: one IMG_WORDS image per stream (desc[k].scratch_off = its index); d_scratch: gridDim * WAVES slots of slot_words words
int launch_fqz_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_images, const uint32_t *d_overflow, size_t n, void *d_out,
                      int32_t *d_status, uint32_t *d_scratch, size_t slots, size_t slot_words, hipStream_t s) {
    (void)ctx;
    if (!n) return HG_OK;
    constexpr int WAVES = 1;
    const size_t wgs = (slots + WAVES - 1) / WAVES;
    static const int qc_on = [] { const char *e = getenv("HG_FQZ_FAST"); return (e && *e == '0') ? 0 : 1; }();   // 0: the general step for every stream (A/B)
    hipLaunchKernelGGL((hgq::fqz_decode_kernel<WAVES>), dim3((unsigned)wgs), dim3(WAVES * 64), 0, s, (const uint8_t *)d_in, d_desc, d_images, d_overflow,
                       (uint32_t)n, (uint8_t *)d_out, d_status, d_scratch, (unsigned long long)slot_words, qc_on);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
int launch_fqz_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_images, const uint32_t *d_rec_len,
                      const uint32_t *d_rec_flags, size_t n, void *d_out, uint32_t *d_out_len, uint32_t *d_scratch, size_t slots,
                      size_t slot_words, hipStream_t s) {
    (void)ctx;
    if (!n) return HG_OK;
    constexpr int WAVES = 2;
    const size_t wgs = (slots + WAVES - 1) / WAVES;
    hipLaunchKernelGGL((hgq::fqz_encode_kernel<WAVES>), dim3((unsigned)wgs), dim3(WAVES * 64), 0, s, (const uint8_t *)d_in, d_desc, d_images,
                       d_rec_len, d_rec_flags, (uint32_t)n, (uint8_t *)d_out, d_out_len, d_scratch, (unsigned long long)slot_words);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
// the two-phase encoder: events -> sort -> models -> coder (arith_enc2.hip's code kernel over the records); d_info: 2 words per stream {0, events}
int launch_fqz_encode2(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const hg_stream_desc *d_code_desc, const uint32_t *d_images, const uint32_t *d_rec_len,
                       const uint32_t *d_rec_flags, const uint32_t *d_rec_off, const void *d_streams, size_t n, size_t max_ecap, void *d_work, uint32_t *d_info, void *d_out,
                       uint32_t *d_out_len, hipStream_t s) {
    (void)ctx;
    if (!n) return HG_OK;
    const hgq::Enc2Stream *st = (const hgq::Enc2Stream *)d_streams;
    static const bool times = getenv("HG_FQZ_TIMES") && atoi(getenv("HG_FQZ_TIMES")) > 0;   // the four kernels' durations on stderr (synchronises; measurement runs only)
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto mark = [&](int i) { if (times) { (void)hipEventCreate(&ev[i]); (void)hipEventRecord(ev[i], s); } };
    mark(0);
    hipLaunchKernelGGL(hgq::fqz_events_kernel, dim3((unsigned)n), dim3(256), 0, s, (const uint8_t *)d_in, d_desc, d_images, d_rec_len, d_rec_flags, d_rec_off, st, (uint8_t *)d_work, d_info);
    mark(1);
    hipLaunchKernelGGL(hgq::fqz_sort_kernel, dim3((unsigned)n), dim3(64), 0, s, st, (uint8_t *)d_work, (const uint32_t *)d_info);
    mark(2);
    const unsigned stretches = (unsigned)((max_ecap + hgq::ENC2_CUT - 1) / hgq::ENC2_CUT);
    hipLaunchKernelGGL(hgq::fqz_models_kernel, dim3((stretches + 3) / 4, (unsigned)n), dim3(256), 0, s, d_desc, d_images, st, (uint8_t *)d_work, (const uint32_t *)d_info);
    mark(3);
    if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    const int rc = launch_range_code(d_code_desc, n, d_info, d_work, d_out, d_out_len, s);
    mark(4);
    if (times && hipStreamSynchronize(s) == hipSuccess) {
        float t[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&t[i], ev[i], ev[i + 1]);
        fprintf(stderr, "[fqz 2p] %zu blocks: events %.2f ms, sort %.2f ms, models %.2f ms, coder %.2f ms\n", n, t[0], t[1], t[2], t[3]);
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
    }
    return rc;
}
// resident wavefronts: every one owns 65536 models in scratch, so the count is bounded by memory as well as by the chip
static int fqz_slots(hg_ctx *ctx, size_t m, size_t slot_words, size_t *slots, int per_cu = 8) {
    size_t n = std::min<size_t>(m, (size_t)ctx->cus * per_cu);
    size_t freeb = 0, totalb = 0;
    if (hipMemGetInfo(&freeb, &totalb) == hipSuccess) {
        const size_t fit = ((freeb + ctx->d_scratch_cap[6]) / 2) / (slot_words * 4);
        if (fit < 2) return HG_ENOMEM;
        n = std::min(n, fit & ~(size_t)1);
    }
    *slots = (n + 1) & ~(size_t)1;
    return HG_OK;
}
}  // namespace hg

// Host convenience for n method-7 blocks: in[i]/in_len[i] -> out[i] (out_len[i] = the block's uncompressed size, which must
// equal the size stored in the stream).  status[i] = 0, -1 (malformed) or HG_BLOCK_EUNSUPPORTED.  Synchronous.
extern "C" int hg_fqz_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n, uint8_t *const *out,
                                  const uint32_t *out_len, int32_t *status) {
    if (!ctx || (n && (!in || !in_len || !out || !out_len || !status))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    std::vector<uint32_t> images; images.reserve(n * hgq::IMG_WORDS);
    std::vector<uint32_t> overflow(1, 0u);                            // parameter sets beyond the two that fit LDS (img[6] = a stream's offset)
    std::vector<size_t> live;                                         // streams that reach the device, longest first
    uint32_t max_ns = 0;
    for (size_t i = 0; i < n; i++) {
        status[i] = 0;
        if (out_len[i] == 0) { continue; }
        uint32_t img[hgq::IMG_WORDS];
        const size_t over0 = overflow.size();
        const int r = hgq::build_image(in[i], in_len[i], out_len[i], img, overflow);
        if (r) { overflow.resize(over0); status[i] = r == -3 ? HG_BLOCK_EUNSUPPORTED : -1; continue; }
        images.insert(images.end(), img, img + hgq::IMG_WORDS);
        live.push_back(i);
        max_ns = std::max(max_ns, img[3]);
    }
    int rc = HG_OK;
    const size_t m = live.size();
    if (m) {
        std::vector<size_t> ord(m);
        for (size_t k = 0; k < m; k++) ord[k] = k;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return out_len[live[a]] > out_len[live[b]]; });
        std::vector<hg_stream_desc> desc(m);
        std::vector<const uint8_t *> sp(m); std::vector<uint32_t> sl(m), ol(m); std::vector<uint64_t> so(m), oo(m); std::vector<uint8_t *> dp(m);
        uint64_t ioff = 0, ooff = 0;
        for (size_t k = 0; k < m; k++) {
            const size_t i = live[ord[k]];
            desc[k].in_off = ioff; desc[k].in_len = in_len[i]; desc[k].out_off = ooff; desc[k].out_len = out_len[i];
            desc[k].scratch_off = (uint32_t)ord[k]; desc[k].reserved = 0;
            sp[k] = in[i]; sl[k] = in_len[i]; so[k] = ioff; oo[k] = ooff; ol[k] = out_len[i]; dp[k] = out[i];
            ioff += ((uint64_t)in_len[i] + 15u) & ~15ull; ooff += ((uint64_t)out_len[i] + 63u) & ~63ull;
        }
        const size_t slot_words = (size_t)hgq::CTX_SIZE * (max_ns + 1u);
        size_t slots = 0;
        if ((rc = hg::fqz_slots(ctx, m, slot_words, &slots))) return rc;
        if ((rc = hg::ensure_scratch(ctx, 0, ioff + 64)) || (rc = hg::ensure_scratch(ctx, 1, ooff + 64)) ||
            (rc = hg::ensure_scratch(ctx, 2, m * sizeof(hg_stream_desc))) || (rc = hg::ensure_scratch(ctx, 3, m * 4 + 64)) ||
            (rc = hg::ensure_scratch(ctx, 4, images.size() * 4 + 64)) || (rc = hg::ensure_scratch(ctx, 5, overflow.size() * 4 + 64)) ||
            (rc = hg::ensure_scratch(ctx, 6, slots * slot_words * 4 + 512))) return rc;   // (+: the kernel reads a model with all 64 lanes, whatever its length)
        hipStream_t s = ctx->stream;
        bool ok = hg::stage_upload(ctx, sp.data(), sl.data(), so.data(), nullptr, m, ioff, (uint8_t *)ctx->d_scratch[0], s) == HG_OK;
        ok = ok && hipMemcpyAsync(ctx->d_scratch[2], desc.data(), m * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
             hipMemcpyAsync(ctx->d_scratch[4], images.data(), images.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
             hipMemcpyAsync(ctx->d_scratch[5], overflow.data(), overflow.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess;
        rc = ok ? hg::launch_fqz_decode(ctx, ctx->d_scratch[0], (const hg_stream_desc *)ctx->d_scratch[2], (const uint32_t *)ctx->d_scratch[4],
                                        (const uint32_t *)ctx->d_scratch[5], m,
                                        ctx->d_scratch[1], (int32_t *)ctx->d_scratch[3], (uint32_t *)ctx->d_scratch[6], slots, slot_words, s)
                : HG_ELAUNCH;
        if (rc == HG_OK) {
            std::vector<int32_t> st(m);
            ok = hipMemcpyAsync(st.data(), ctx->d_scratch[3], m * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
            for (size_t k = 0; k < m && ok; k++) { status[live[ord[k]]] = st[k]; if (st[k] != 0) ol[k] = 0; }
            ok = ok && hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], oo.data(), ol.data(), dp.data(), m, s) == HG_OK;
            if (!ok) rc = HG_ELAUNCH;
        }
    }
    if (rc != HG_OK) return rc;
    for (size_t i = 0; i < n; i++) if (status[i] != 0) return HG_EBLOCK;
    return HG_OK;
}

// Encoder (replaces fqz_compress as called by cram_compress_by_method, cram/cram_io.c:1801-1825).  slice[i] = the record
// lengths and BAM flags of block i (the reference's fqz_slice, built at cram_io.c:1808-1820), strat[i] = 0..3 (methods FQZ,
// FQZ_b, FQZ_c, FQZ_d, cram_io.c:2065-2068).  out[i] must hold hg_fqz_compress_bound(in_len[i], num_records).  A block whose
// record lengths do not add up to in_len[i] gets out_len[i] = 0 (the caller keeps another method).  Synchronous.
extern "C" size_t hg_fqz_compress_bound(size_t n, size_t nrec) { return n + n / 4 + 8 * nrec + 16384; }
extern "C" int hg_fqz_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const hg_fqz_slice *const *slice,
                                  const int32_t *strat, size_t n, uint8_t *const *out, uint32_t *out_len) {
    if (!ctx || (n && (!in || !in_len || !slice || !strat || !out || !out_len))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    const bool stats = getenv("HTS_GPU_STATS") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<uint32_t> images; images.reserve(n * hgq::IMG_WORDS);
    std::vector<std::vector<uint8_t>> hdrs;
    std::vector<size_t> live;
    uint32_t max_ns = 0;
    {
        // the parameter choice reads every quality once (census, duplicates): blocks are independent -> a few host threads
        std::vector<std::vector<uint32_t>> img_of(n); std::vector<std::vector<uint8_t>> hdr_of(n); std::vector<char> good(n, 0);
        auto work = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                out_len[i] = 0;
                if (in_len[i] == 0) continue;
                img_of[i].resize(hgq::IMG_WORDS);
                good[i] = hgq::build_encode_image(in[i], in_len[i], slice[i], strat[i], img_of[i].data(), hdr_of[i]) == 0;
            }
        };
        uint64_t total = 0; for (size_t i = 0; i < n; i++) total += in_len[i];
        const size_t nt = total < (8u << 20) || n < 2 ? 1 : std::min<size_t>({n, 16, std::max(1u, std::thread::hardware_concurrency())});
        if (nt <= 1) work(0, n);
        else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
            for (auto &t : th) t.join();
        }
        for (size_t i = 0; i < n; i++) {
            if (!good[i]) continue;
            images.insert(images.end(), img_of[i].begin(), img_of[i].end());
            hdrs.push_back(std::move(hdr_of[i]));
            live.push_back(i);
            max_ns = std::max(max_ns, img_of[i][3]);
        }
    }
    const size_t m = live.size();
    if (!m) return HG_OK;
    const double ms_images = ms_since(t_call);
    std::vector<size_t> ord(m);
    for (size_t k = 0; k < m; k++) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return in_len[live[a]] > in_len[live[b]]; });
    std::vector<hg_stream_desc> desc(m);
    std::vector<const uint8_t *> sp(m); std::vector<uint32_t> sl(m); std::vector<uint64_t> so(m), oo(m); std::vector<uint8_t *> dp(m);
    std::vector<uint32_t> rlen, rflg;
    uint64_t ioff = 0, ooff = 0;
    for (size_t k = 0; k < m; k++) {
        const size_t i = live[ord[k]];
        const hg_fqz_slice *f = slice[i];
        desc[k].in_off = ioff; desc[k].in_len = in_len[i]; desc[k].out_off = ooff; desc[k].out_len = 0;
        desc[k].scratch_off = (uint32_t)ord[k]; desc[k].reserved = (uint32_t)rlen.size();
        rlen.insert(rlen.end(), f->len, f->len + f->num_records);
        if (f->flags) rflg.insert(rflg.end(), f->flags, f->flags + f->num_records); else rflg.resize(rflg.size() + f->num_records, 0u);
        sp[k] = in[i]; sl[k] = in_len[i]; so[k] = ioff; oo[k] = ooff; dp[k] = out[i] + hdrs[ord[k]].size();
        ioff += ((uint64_t)in_len[i] + 15u) & ~15ull;
        ooff += (hg_fqz_compress_bound(in_len[i], f->num_records) + 63u) & ~63ull;
        if (rlen.size() > 0xffffffffull) return HG_EINVAL;
    }
    int rc;
    hipStream_t s = ctx->stream;
    // ---- two phases (events sorted by model, register models, scalar coder pass) when the work arrays fit: 24 bytes per event.  HG_FQZ_2P=0: the one-pass kernel
    //      (A/B runs); it also takes over when the device memory does not do, and for blocks beyond 2^24 events (event numbers travel in 24 bits).
    {
        const bool want = !(getenv("HG_FQZ_2P") && atoi(getenv("HG_FQZ_2P")) == 0);
        std::vector<hgq::Enc2Stream> st(m);
        std::vector<hg_stream_desc> cdesc(m);
        std::vector<uint32_t> roff(rlen.size());
        uint64_t woff = 0; size_t max_ecap = 0; bool fits = want;
        size_t rbase = 0;
        for (size_t k = 0; k < m && fits; k++) {
            const size_t i = live[ord[k]];
            const uint32_t nrec = slice[i]->num_records;
            const uint64_t ecap = (uint64_t)in_len[i] + 7ull * nrec;
            if (ecap >= (1u << 24)) { fits = false; break; }
            auto take = [&](uint64_t bytes) { const uint64_t at = woff; woff += (bytes + 15u) & ~15ull; return at; };
            st[k].rec = take(ecap * 8); st[k].key0 = take(ecap * 4); st[k].pay0 = take(ecap * 4); st[k].key1 = take(ecap * 4); st[k].pay1 = take(ecap * 4);
            st[k].ebase = take((uint64_t)nrec * 4);
            st[k].ecap = (uint32_t)ecap; st[k].nrec_first = (uint32_t)rbase;
            uint32_t at = 0;
            for (uint32_t r = 0; r < nrec; r++) { roff[rbase + r] = at; at += rlen[rbase + r]; }
            rbase += nrec;
            max_ecap = std::max<size_t>(max_ecap, (size_t)ecap);
            cdesc[k] = desc[k]; cdesc[k].scratch_off = (uint32_t)(2 * k); cdesc[k].reserved = (uint32_t)(st[k].rec / 16);
            if (st[k].rec / 16 > 0xffffffffull) fits = false;
        }
        if (fits) {
            size_t freeb = 0, totalb = 0;
            if (hipMemGetInfo(&freeb, &totalb) != hipSuccess || woff + ioff + ooff > freeb / 2 + ctx->d_scratch_cap[5] + ctx->d_scratch_cap[0] + ctx->d_scratch_cap[1]) fits = false;
        }
        if (fits) {
            const size_t recb = rlen.size() * 4;
            if ((rc = hg::ensure_scratch(ctx, 0, ioff + 64)) || (rc = hg::ensure_scratch(ctx, 1, ooff + 64)) || (rc = hg::ensure_scratch(ctx, 2, 2 * m * sizeof(hg_stream_desc) + 64)) ||
                (rc = hg::ensure_scratch(ctx, 3, m * 4 + 64)) || (rc = hg::ensure_scratch(ctx, 4, images.size() * 4 + 64)) || (rc = hg::ensure_scratch(ctx, 5, woff + 64)) ||
                (rc = hg::ensure_scratch(ctx, 7, 3 * recb + 64)) || (rc = hg::ensure_scratch(ctx, 8, m * sizeof(hgq::Enc2Stream) + m * 8 + 64))) return rc;
            uint32_t *d_rlen = (uint32_t *)ctx->d_scratch[7], *d_rflg = d_rlen + rlen.size(), *d_roff = d_rflg + rlen.size();
            hg_stream_desc *d_desc = (hg_stream_desc *)ctx->d_scratch[2], *d_cdesc = d_desc + m;
            uint8_t *d_st = (uint8_t *)ctx->d_scratch[8]; uint32_t *d_info = (uint32_t *)(d_st + ((m * sizeof(hgq::Enc2Stream) + 15) & ~(size_t)15));
            bool ok = hg::stage_upload(ctx, sp.data(), sl.data(), so.data(), nullptr, m, ioff, (uint8_t *)ctx->d_scratch[0], s) == HG_OK;
            ok = ok && hipMemcpyAsync(d_desc, desc.data(), m * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(d_cdesc, cdesc.data(), m * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(ctx->d_scratch[4], images.data(), images.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(d_rlen, rlen.data(), recb, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(d_rflg, rflg.data(), recb, hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(d_roff, roff.data(), recb, hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(d_st, st.data(), m * sizeof(hgq::Enc2Stream), hipMemcpyHostToDevice, s) == hipSuccess;
            const auto t_dev = std::chrono::steady_clock::now();
            if (stats) (void)hipStreamSynchronize(s);
            const double ms_up = ms_since(t_dev);
            rc = ok ? hg::launch_fqz_encode2(ctx, ctx->d_scratch[0], d_desc, d_cdesc, (const uint32_t *)ctx->d_scratch[4], d_rlen, d_rflg, d_roff, d_st, m, max_ecap, ctx->d_scratch[5], d_info,
                                             ctx->d_scratch[1], (uint32_t *)ctx->d_scratch[3], s) : HG_ELAUNCH;
            if (rc != HG_OK) return rc;
            std::vector<uint32_t> ol(m);
            ok = hipMemcpyAsync(ol.data(), ctx->d_scratch[3], m * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
            const double ms_kern = ms_since(t_dev) - ms_up;
            ok = ok && hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], oo.data(), ol.data(), dp.data(), m, s) == HG_OK;
            if (!ok) return HG_ELAUNCH;
            for (size_t k = 0; k < m; k++) {
                const size_t i = live[ord[k]];
                const std::vector<uint8_t> &h = hdrs[ord[k]];
                memcpy(out[i], h.data(), h.size());
                out_len[i] = (uint32_t)h.size() + ol[k];
            }
            if (stats) fprintf(stderr, "[hts-gpu] fqz encode, two phases: %zu blocks, %.1f MB: parameter choice (host) %.1f ms, staging + upload %.1f ms, kernels %.1f ms, whole call %.1f ms\n", m,
                               ioff / 1e6, ms_images, ms_up, ms_kern, ms_since(t_call));
            return HG_OK;
        }
    }
    const size_t slot_words = (size_t)hgq::CTX_SIZE * (max_ns + 1u);
    size_t slots = 0;
    if ((rc = hg::fqz_slots(ctx, m, slot_words, &slots))) return rc;
    const size_t recb = rlen.size() * 4;
    if ((rc = hg::ensure_scratch(ctx, 0, ioff + 64)) || (rc = hg::ensure_scratch(ctx, 1, ooff + 64)) ||
        (rc = hg::ensure_scratch(ctx, 2, m * sizeof(hg_stream_desc))) || (rc = hg::ensure_scratch(ctx, 3, m * 4 + 64)) ||
        (rc = hg::ensure_scratch(ctx, 4, images.size() * 4 + 64)) || (rc = hg::ensure_scratch(ctx, 5, 2 * recb + 64)) ||
        (rc = hg::ensure_scratch(ctx, 6, slots * slot_words * 4 + 64))) return rc;
    uint32_t *d_rlen = (uint32_t *)ctx->d_scratch[5], *d_rflg = d_rlen + rlen.size();
    bool ok = hg::stage_upload(ctx, sp.data(), sl.data(), so.data(), nullptr, m, ioff, (uint8_t *)ctx->d_scratch[0], s) == HG_OK;
    ok = ok && hipMemcpyAsync(ctx->d_scratch[2], desc.data(), m * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(ctx->d_scratch[4], images.data(), images.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(d_rlen, rlen.data(), recb, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(d_rflg, rflg.data(), recb, hipMemcpyHostToDevice, s) == hipSuccess;
    rc = ok ? hg::launch_fqz_encode(ctx, ctx->d_scratch[0], (const hg_stream_desc *)ctx->d_scratch[2], (const uint32_t *)ctx->d_scratch[4], d_rlen,
                                    d_rflg, m, ctx->d_scratch[1], (uint32_t *)ctx->d_scratch[3], (uint32_t *)ctx->d_scratch[6], slots, slot_words, s)
            : HG_ELAUNCH;
    if (rc != HG_OK) return rc;
    std::vector<uint32_t> ol(m);
    ok = hipMemcpyAsync(ol.data(), ctx->d_scratch[3], m * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    ok = ok && hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], oo.data(), ol.data(), dp.data(), m, s) == HG_OK;
    if (!ok) return HG_ELAUNCH;
    for (size_t k = 0; k < m; k++) {
        const size_t i = live[ord[k]];
        const std::vector<uint8_t> &h = hdrs[ord[k]];
        memcpy(out[i], h.data(), h.size());
        out_len[i] = (uint32_t)h.size() + ol[k];
    }
    return HG_OK;
}

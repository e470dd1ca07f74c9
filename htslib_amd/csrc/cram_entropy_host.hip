// cram_entropy_host.hip -- host planners for the CRAM 3.1 rANS Nx16 and adaptive-arithmetic block codecs
// (hg_ransnx16_decode_host / hg_ransnx16_encode_host / hg_arith_decode_host / hg_arith_encode_host).
//
// Replaces rans_uncompress_4x16 / arith_uncompress_to at their call sites in cram_uncompress_block (reference
// cram/cram_io.c:1697-1733).  Both formats share one container: a stream is a small tree, STRIPE splits it into S complete
// sub-streams, and every leaf is  [PACK header] [RLE header + meta stream] entropy-coded core.
// The planner walks the few header bytes of every stream on the host (no payload byte is touched
// here), and emits
//   * "core" jobs  -- entropy decode (rANS order 0/1, 4- or 32-way, or CAT) : ransnx16.hip
//                     (adaptive range coder order 0/1, with or without its run-length models) : arith.hip
//   * "xform" jobs -- RLE expand / bit unpack / strided (de-striping) write : ransnx16_xform.hip
// then runs the two kernels back to back on one HIP stream.  All payload work is on the GPU; there
// is no CPU decode path.  Header rules follow oracle/ransnx16_oracle.c (PARITY UNPINNED).
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <functional>
#include <vector>
#include "htsgpu.h"
#include "hg_internal.h"

using hg::ensure_scratch;
namespace {

enum { F_ORDER = 1, F_X32 = 4, F_EXT = 4, F_STRIPE = 8, F_NOSZ = 16, F_CAT = 32, F_RLE = 64, F_PACK = 128 };
constexpr uint32_t NONE = 0xffffffffu;
enum Codec { NX16 = 0, ARITH = 1 };
enum CoreClass { C_NX4 = 0, C_NX32 = 1, C_ARITH_SMALL = 2, C_ARITH_BIG = 3, C_ARITH_2P = 4 /* encoder only: arith_enc2.hip */, C_CLASSES = 5 };

struct Plan {
    std::vector<hg_stream_desc> core;      // out_off is into the work buffer, or (bit 63 set) the output buffer
    std::vector<uint32_t> core_top;
    std::vector<uint8_t> core_cls;
    std::vector<hg::nx16_xform> xf;
    std::vector<uint32_t> xf_top;
    uint64_t work = 0, scratch = 0;
    bool too_big = false;
};

int get_u7(const uint8_t *&cp, const uint8_t *end, uint32_t &v) {
    uint32_t x = 0;
    for (int n = 0; n < 5; n++) {
        if (cp >= end) return -1;
        const uint8_t c = *cp++;
        x = (x << 7) | (c & 0x7fu);
        if (!(c & 0x80u)) { v = x; return 0; }
    }
    return -1;
}

uint64_t work_alloc(Plan &P, uint64_t bytes) { const uint64_t o = P.work; P.work += (bytes + 31u) & ~15ull; return o; }

// order-1 table scratch (words) a core may need: 512 index words + the (possibly rANS-packed) table text +
// one word per (context, symbol) pair + sentinels
uint64_t o1_scratch_words(const uint8_t *cp, const uint8_t *end) {
    if (cp >= end) return 16;
    const uint32_t comp = *cp++ & 1u;
    uint64_t tab_bytes = (uint64_t)(end - cp), extra = 0;
    if (comp) {
        uint32_t ulen = 0;
        if (get_u7(cp, end, ulen)) return 16;
        if (ulen > 262144u) ulen = 262144u;
        tab_bytes = ulen; extra = (ulen + 3) / 4;
    }
    const uint64_t entries = tab_bytes < 65536u + 256u ? tab_bytes : 65536u + 256u;
    return 512 + extra + entries + 256 + 16;
}

// Adds the entropy-decode job for one payload; returns its status slot.
uint32_t add_core(Plan &P, Codec codec, uint32_t top, uint64_t in_base, const uint8_t *base, const uint8_t *cp, const uint8_t *end,
                  uint32_t flags, uint32_t out_len, uint64_t out_off) {
    hg_stream_desc d;
    memset(&d, 0, sizeof d);
    d.in_off = in_base + (uint64_t)(cp - base);
    d.in_len = (uint32_t)(end - cp);
    d.out_off = out_off; d.out_len = out_len;
    d.scratch_off = (uint32_t)P.scratch;
    uint8_t cls;
    if (codec == NX16 || (flags & F_CAT)) {                 // raw copies of either codec go through the Nx16 core kernel
        if (codec == ARITH) flags &= F_CAT;
        d.reserved = 0x80000000u | (flags & (F_ORDER | F_X32 | F_CAT));
        P.scratch += ((flags & F_ORDER) && !(flags & F_CAT)) ? o1_scratch_words(cp, end) : 16;
        cls = (flags & F_X32) ? C_NX32 : C_NX4;
    } else {
        d.reserved = 0x80000000u | (flags & (F_ORDER | F_RLE));
        const uint32_t words = hg::arith_model_words(cp < end ? *cp : 1u, flags);
        cls = words <= HG_ARITH_POOL_SMALL ? C_ARITH_SMALL : C_ARITH_BIG;
        P.scratch += words > HG_ARITH_POOL_BIG ? words + 16 : 16;
    }
    if (P.scratch > 0xffffffffull) P.too_big = true;
    P.core.push_back(d); P.core_top.push_back(top); P.core_cls.push_back(cls);
    return (uint32_t)P.core.size() - 1;
}

// Plans one (sub-)stream.  known = size when the caller knows it (NOSZ).  out_off/stride place byte i of this
// stream at output offset out_off + i*stride.  Returns 0, -1 (malformed).
int plan_stream(Plan &P, Codec codec, uint32_t top, uint64_t in_base, const uint8_t *base, const uint8_t *cp, const uint8_t *end,
                long long known, uint64_t out_off, uint32_t stride, int depth) {
    if (cp >= end || depth > 8) return -1;
    const uint32_t flags = *cp++;
    uint32_t ulen;
    if (flags & F_NOSZ) { if (known < 0) return -1; ulen = (uint32_t)known; }
    else if (get_u7(cp, end, ulen)) return -1;
    if (known >= 0 && ulen != (uint32_t)known) return -1;
    if (flags & F_STRIPE) {
        if (cp >= end) return -1;
        const uint32_t S = *cp++;
        if (S < 1 || S > 32) return -1;
        uint32_t cl[32];
        for (uint32_t k = 0; k < S; k++) if (get_u7(cp, end, cl[k])) return -1;
        for (uint32_t k = 0; k < S; k++) {
            const uint32_t m = ulen / S + ((ulen % S) > k ? 1u : 0u);
            if ((uint64_t)(end - cp) < cl[k]) return -1;
            if (int r = plan_stream(P, codec, top, in_base, base, cp, cp + cl[k], m, out_off + (uint64_t)k * stride, stride * S, depth + 1)) return r;
            cp += cl[k];
        }
        return 0;
    }
    uint32_t nsym = 0, plen = ulen; uint8_t map[16] = {0};
    if (flags & F_PACK) {
        if (cp >= end) return -1;
        nsym = *cp++;
        if (nsym > 16 || (uint64_t)(end - cp) < nsym) return -1;
        memcpy(map, cp, nsym); cp += nsym;
        if (get_u7(cp, end, plen) || plen > ulen) return -1;
    }
    uint32_t lit_len = plen, meta_len = 0, dep1 = NONE; uint64_t meta_off = 0; bool meta_in_work = false;
    if (codec == ARITH && (flags & F_EXT) && !(flags & F_CAT)) return HG_BLOCK_EUNSUPPORTED;      // bzip2 payload
    const bool rle_xform = codec == NX16 && (flags & F_RLE);         // the range coder's RLE lives inside its models
    if (rle_xform) {
        uint32_t v;
        if (get_u7(cp, end, v) || get_u7(cp, end, lit_len)) return -1;
        meta_len = v >> 1;
        if (lit_len > plen || meta_len > 5ull * lit_len + 257) return -1;
        if (v & 1u) {
            if ((uint64_t)(end - cp) < meta_len) return -1;
            meta_off = in_base + (uint64_t)(cp - base); cp += meta_len;
        } else {
            uint32_t cl;
            if (get_u7(cp, end, cl) || (uint64_t)(end - cp) < cl) return -1;
            meta_off = work_alloc(P, meta_len); meta_in_work = true;
            dep1 = add_core(P, codec, top, in_base, base, cp, cp + cl, 0, meta_len, meta_off);
            cp += cl;
        }
    }
    const bool xform = rle_xform || (flags & F_PACK) || stride != 1;
    if (!xform) {                                         // plain core straight into the output buffer
        if (ulen) add_core(P, codec, top, in_base, base, cp, end, flags, ulen, out_off | (1ull << 63));
        return 0;
    }
    hg::nx16_xform J;
    memset(&J, 0, sizeof J);
    J.s1_off = work_alloc(P, lit_len);
    J.dep0 = lit_len ? add_core(P, codec, top, in_base, base, cp, end, flags, lit_len, J.s1_off) : NONE;
    J.dep1 = dep1;
    J.meta_off = meta_off; J.meta_len = meta_len;
    J.lit_len = lit_len; J.plen = plen; J.ulen = ulen;
    J.ops = (rle_xform ? 1u : 0u) | ((flags & F_PACK) ? 2u : 0u) | (meta_in_work ? 4u : 0u);
    if (rle_xform && (flags & F_PACK)) J.s2_off = work_alloc(P, plen);
    J.out_off = out_off; J.stride = stride; J.nsym = nsym;
    memcpy(J.map, map, 16);
    P.xf.push_back(J); P.xf_top.push_back(top);
    return 0;
}

}  // namespace

// Uploads a plan and launches its kernels on stream s (no synchronisation).  Device layout: inputs in scratch
// slot 0 (the caller has copied them), outputs in slot 1 = [ obytes of output | P.work bytes of work area ],
// job statuses in slot 3: nc entropy jobs, then nx transform jobs.
static int launch_plan(hg_ctx *ctx, Plan &P, uint64_t in_bytes, uint64_t obytes, hipStream_t s) {
    const size_t nc = P.core.size(), nx = P.xf.size(), nst = nc + nx;
    std::vector<uint32_t> sel(nc);
    size_t cnt[C_CLASSES] = {0}, first[C_CLASSES + 1] = {0};
    for (size_t k = 0; k < nc; k++) {
        hg_stream_desc &d = P.core[k];
        if (d.out_off >> 63) d.out_off &= ~(1ull << 63); else d.out_off += obytes;
        cnt[P.core_cls[k]]++;
    }
    for (int c = 0; c < C_CLASSES; c++) first[c + 1] = first[c] + cnt[c];
    { size_t fill[C_CLASSES]; for (int c = 0; c < C_CLASSES; c++) fill[c] = first[c];
      for (size_t k = 0; k < nc; k++) sel[fill[P.core_cls[k]]++] = (uint32_t)k; }
    // streams that share a wavefront (16 four-way streams) run in lock step: neighbours alike -- same flags, then by length
    for (int c = 0; c < C_CLASSES; c++)
        std::stable_sort(sel.begin() + first[c], sel.begin() + first[c + 1], [&](uint32_t a, uint32_t b) {
            const uint32_t fa = P.core[a].reserved & 0xffu, fb = P.core[b].reserved & 0xffu;
            if (fa != fb) return fa < fb;
            return P.core[a].out_len > P.core[b].out_len;
        });
    int rc;
    if ((rc = ensure_scratch(ctx, 0, in_bytes + 64)) || (rc = ensure_scratch(ctx, 1, obytes + P.work + 64)) ||
        (rc = ensure_scratch(ctx, 2, nc * sizeof(hg_stream_desc) + 64)) || (rc = ensure_scratch(ctx, 3, nst * 4 + 64)) ||
        (rc = ensure_scratch(ctx, 4, nx * sizeof(hg::nx16_xform) + 64)) ||
        (rc = ensure_scratch(ctx, 6, P.scratch * 4 + 64)) || (rc = ensure_scratch(ctx, 7, nc * 4 + 64))) return rc;
    uint8_t *d_in = (uint8_t *)ctx->d_scratch[0], *d_out = (uint8_t *)ctx->d_scratch[1];
    int32_t *d_st = (int32_t *)ctx->d_scratch[3];
    bool ok = hipMemsetAsync(d_st, 0xff, nst * 4 + 4, s) == hipSuccess;
    if (nc) ok = ok && hipMemcpyAsync(ctx->d_scratch[2], P.core.data(), nc * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(ctx->d_scratch[7], sel.data(), nc * 4, hipMemcpyHostToDevice, s) == hipSuccess;
    if (nx) ok = ok && hipMemcpyAsync(ctx->d_scratch[4], P.xf.data(), nx * sizeof(hg::nx16_xform), hipMemcpyHostToDevice, s) == hipSuccess;
    rc = ok ? HG_OK : HG_ELAUNCH;
    const uint32_t *d_sel = (const uint32_t *)ctx->d_scratch[7];
    if (rc == HG_OK && cnt[C_NX4] + cnt[C_NX32])
        rc = hg::launch_ransnx16_decode(ctx, d_in, (const hg_stream_desc *)ctx->d_scratch[2], d_sel + first[C_NX4], cnt[C_NX4],
                                        d_sel + first[C_NX32], cnt[C_NX32], d_out, d_st, (uint32_t *)ctx->d_scratch[6], s);
    if (rc == HG_OK && cnt[C_ARITH_SMALL] + cnt[C_ARITH_BIG])
        rc = hg::launch_arith_decode(ctx, d_in, (const hg_stream_desc *)ctx->d_scratch[2], d_sel + first[C_ARITH_SMALL], cnt[C_ARITH_SMALL],
                                     d_sel + first[C_ARITH_BIG], cnt[C_ARITH_BIG], d_out, d_st, (uint32_t *)ctx->d_scratch[6], s);
    if (rc == HG_OK && nx)
        rc = hg::launch_ransnx16_xform(ctx, d_in, d_out + obytes, d_out, (const hg::nx16_xform *)ctx->d_scratch[4], nx, d_st, (uint32_t)nc, s);
    return rc;
}

// After the stream has been synchronised: folds the job statuses into per-top statuses st[].
static bool collect_plan_status(hg_ctx *ctx, const Plan &P, std::vector<int32_t> &st, hipStream_t s) {
    const size_t nc = P.core.size(), nx = P.xf.size(), nst = nc + nx;
    std::vector<int32_t> jst(nst + 1);
    if (nst && (hipMemcpyAsync(jst.data(), ctx->d_scratch[3], nst * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) return false;
    for (size_t k = 0; k < nc; k++) if (jst[k] != 0 && st[P.core_top[k]] == 0) st[P.core_top[k]] = jst[k];
    for (size_t k = 0; k < nx; k++) if (jst[nc + k] != 0 && st[P.xf_top[k]] == 0) st[P.xf_top[k]] = jst[nc + k];
    return true;
}

static int entropy_decode_host(Codec codec, hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                               uint8_t *const *out, const uint32_t *out_len, int32_t *status) {
    if (!ctx || (n && (!in || !in_len || !out || !out_len))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    Plan P;
    std::vector<uint64_t> ioffs(n), ooffs(n);
    std::vector<int32_t> st(n, 0);
    uint64_t ioff = 0, ooff = 0;
    for (size_t i = 0; i < n; i++) {
        ioffs[i] = ioff; ooffs[i] = ooff;
        const size_t c0 = P.core.size(), x0 = P.xf.size();
        const uint64_t w0 = P.work, s0 = P.scratch;
        if (int r = plan_stream(P, codec, (uint32_t)i, ioff, in[i], in[i], in[i] + in_len[i], out_len[i], ooff, 1, 0)) {
            st[i] = r;                                    // malformed header (-1) / bzip2 payload (-3): drop its jobs
            P.core.resize(c0); P.core_top.resize(c0); P.core_cls.resize(c0); P.xf.resize(x0); P.xf_top.resize(x0); P.work = w0; P.scratch = s0;
        }
        ioff += ((uint64_t)in_len[i] + 15u) & ~15ull;
        ooff += ((uint64_t)out_len[i] + 15u) & ~15ull;
    }
    if (P.too_big) return HG_EINVAL;
    const uint64_t obytes = (ooff + 63u) & ~63ull;
    int rc;
    if ((rc = ensure_scratch(ctx, 0, ioff + 64))) return rc;
    hipStream_t s = ctx->stream;
    uint8_t *d_in = (uint8_t *)ctx->d_scratch[0];
    static const bool stats = getenv("HTS_GPU_STATS") != nullptr;
    auto ms = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    const auto t0 = std::chrono::steady_clock::now();
    rc = hg::stage_upload(ctx, in, in_len, ioffs.data(), st.data(), n, ioff, d_in, s);
    const double m_up = ms(t0);
    if (rc == HG_OK) rc = launch_plan(ctx, P, ioff, obytes, s);
    if (rc == HG_OK) {
        bool ok = hipStreamSynchronize(s) == hipSuccess && collect_plan_status(ctx, P, st, s);
        const double m_k = ms(t0);
        if (ok) {
            std::vector<uint32_t> dl(n);
            for (size_t i = 0; i < n; i++) dl[i] = st[i] == 0 ? out_len[i] : 0u;
            rc = hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], ooffs.data(), dl.data(), out, n, s);
        } else rc = HG_ELAUNCH;
        if (stats) fprintf(stderr, "[htsgpu stats] entropy decode (%s, %zu streams, %.1f -> %.1f MB): staged in %.1f ms, kernels done at %.1f ms, out at %.1f ms\n",
                           codec == NX16 ? "nx16" : "arith", n, ioff / 1e6, ooff / 1e6, m_up, m_k, ms(t0));
    }
    if (rc == HG_OK)
        for (size_t i = 0; i < n; i++) { if (status) status[i] = st[i]; if (st[i] != 0) rc = HG_EBLOCK; }
    return rc;
}

// ================================================================================================
// The same plans for streams that are ALREADY ON THE DEVICE (the fused run decoder of cram_file_host.hip, round 6): stream i lies at in_off[i] of the
// context's input image (scratch slot 0, which the caller has filled -- the container bodies of a run as they lie in the file) and its plain bytes go to out_off[i]
// of the output image (slot 1, [0, obytes); the plan's work area follows behind).  Host bytes are needed only for the few header bytes the planner reads.
// launch: everything is queued on the context's stream, nothing is waited for; finish (after the caller's other work): statuses.  codec[i]: 0 Nx16, 1 range coder.
// ================================================================================================
struct hg_entropy_inplace { Plan P; std::vector<int32_t> st; };
int hg_entropy_decode_inplace_launch(hg_ctx *ctx, size_t n, const uint8_t *codec, const uint8_t *const *in, const uint32_t *in_len, const uint64_t *in_off,
                                     const uint32_t *out_len, const uint64_t *out_off, uint64_t in_bytes, uint64_t obytes, hg_entropy_inplace **handle) {
    if (!ctx || !handle || (n && (!codec || !in || !in_len || !in_off || !out_len || !out_off))) return HG_EINVAL;
    hg_entropy_inplace *H = new (std::nothrow) hg_entropy_inplace;
    if (!H) return HG_ENOMEM;
    H->st.assign(n, 0);
    Plan &P = H->P;
    for (size_t i = 0; i < n; i++) {
        const size_t c0 = P.core.size(), x0 = P.xf.size();
        const uint64_t w0 = P.work, s0 = P.scratch;
        if (int r = plan_stream(P, codec[i] ? ARITH : NX16, (uint32_t)i, in_off[i], in[i], in[i], in[i] + in_len[i], out_len[i], out_off[i], 1, 0)) {
            H->st[i] = r;
            P.core.resize(c0); P.core_top.resize(c0); P.core_cls.resize(c0); P.xf.resize(x0); P.xf_top.resize(x0); P.work = w0; P.scratch = s0;
        }
    }
    if (P.too_big) { delete H; return HG_EINVAL; }
    const int rc = launch_plan(ctx, P, in_bytes, obytes, ctx->stream);
    if (rc != HG_OK) { (void)hipStreamSynchronize(ctx->stream); delete H; return rc; }
    *handle = H;
    return HG_OK;
}
int hg_entropy_decode_inplace_finish(hg_ctx *ctx, hg_entropy_inplace *H, int32_t *status) {
    if (!ctx || !H) return HG_EINVAL;
    const bool ok = hipStreamSynchronize(ctx->stream) == hipSuccess && collect_plan_status(ctx, H->P, H->st, ctx->stream);
    int rc = ok ? HG_OK : HG_ELAUNCH;
    for (size_t i = 0; i < H->st.size(); i++) { if (status) status[i] = H->st[i]; if (H->st[i] != 0 && rc == HG_OK) rc = HG_EBLOCK; }
    delete H;
    return rc;
}

// ================================================================================================
// tok3 (name tokeniser, CRAM block method 8) decode: hg_tok3_decode_host replaces tok3_decode_names (call site
// cram/cram_io.c:1735-1749).  The container walk (a few bytes per token stream) is done here; every token stream
// is a complete Nx16 / range-coder stream and joins ONE entropy plan for the whole batch; the names are then
// rebuilt by tok3.hip, one wavefront per block, on the same HIP stream.
// ================================================================================================
extern "C" int hg_tok3_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                                   uint8_t *const *out, const uint32_t *out_len, int32_t *status) {
    if (!ctx || (n && (!in || !in_len || !out || !out_len))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    enum { MAX_TOK = 128, T_END = 12, T_MATCH = 10 };
    Plan P;
    std::vector<int32_t> st(n, 0);
    std::vector<uint64_t> ioffs(n);
    std::vector<hg::tok3_job> jobs;
    std::vector<uint32_t> job_top, tab;
    uint64_t ioff = 0, tboff = 0, ooff = 0, recs = 0, names = 0;
    for (size_t i = 0; i < n; i++) {
        ioffs[i] = ioff;
        const uint64_t my_ioff = ioff;
        ioff += ((uint64_t)in_len[i] + 15u) & ~15ull;
        const uint8_t *b = in[i], *end = b + in_len[i];
        if (in_len[i] < 9) { st[i] = -1; continue; }
        const uint32_t ulen = b[0] | b[1] << 8 | b[2] << 16 | (uint32_t)b[3] << 24, nn = b[4] | b[5] << 8 | b[6] << 16 | (uint32_t)b[7] << 24;
        const uint32_t use_arith = b[8];
        if (ulen != out_len[i] || use_arith > 1 || nn > ulen) { st[i] = -1; continue; }
        const size_t c0 = P.core.size(), x0 = P.xf.size(), t0 = tab.size();
        const uint64_t w0 = P.work, s0 = P.scratch, tb0 = tboff;
        struct Ent { uint32_t off, len; bool set; } E[MAX_TOK][16];
        memset(E, 0, sizeof E);
        int t = -1, bad = 0;
        const uint8_t *cp = b + 9;
        const uint64_t cap = (uint64_t)ulen * 5 + 64;
        while (cp < end && !bad) {
            const uint8_t tt = *cp++;
            const int type = tt & 15;
            if (tt & 0x80) {
                if (++t >= MAX_TOK) { bad = -1; break; }
                if (type != 0) { if (!nn) { bad = -1; break; } E[t][0] = {(uint32_t)type, nn | 0x80000000u, true}; }     // implied TYPE stream
            }
            if (t < 0 || type > T_END || E[t][type].set) { bad = -1; break; }
            if (tt & 0x40) {
                if (end - cp < 2) { bad = -1; break; }
                const int dp = cp[0], dt = cp[1]; cp += 2;
                if (dp > t || dt > T_END || !E[dp][dt].set) { bad = -1; break; }
                E[t][type] = E[dp][dt];
            } else {
                uint32_t clen, sz;
                if (get_u7(cp, end, clen) || (uint64_t)(end - cp) < clen || clen < 1) { bad = -1; break; }
                const uint8_t *sp = cp + 1;
                if ((cp[0] & F_NOSZ) || get_u7(sp, cp + clen, sz) || sz > cap) { bad = -1; break; }
                // resource guard: a name costs at most ~6 stream bytes per character (see the encoder's buffer bound)
                if (tboff - tb0 + sz > 16ull * ulen + 65536u) { bad = -1; break; }
                E[t][type] = {(uint32_t)(tboff - tb0), sz, true};
                if (int r = plan_stream(P, use_arith ? ARITH : NX16, (uint32_t)i, my_ioff, b, cp, cp + clen, -1, tboff, 1, 0)) { bad = r; break; }
                tboff += ((uint64_t)sz + 15u) & ~15ull;
                cp += clen;
            }
        }
        if (bad) {
            st[i] = bad;
            P.core.resize(c0); P.core_top.resize(c0); P.core_cls.resize(c0); P.xf.resize(x0); P.xf_top.resize(x0); P.work = w0; P.scratch = s0;
            tab.resize(t0); tboff = tb0;
            continue;
        }
        hg::tok3_job J;
        memset(&J, 0, sizeof J);
        J.tb_base = tb0; J.out_off = ooff; J.rec_off = recs; J.name_off = names; J.tab_off = (uint32_t)t0;
        J.ntp = (uint32_t)(t + 1); J.nn = nn; J.ulen = ulen;
        uint64_t rc_ = 0;
        for (int a = 0; a <= t; a++) {
            if (a >= 1) rc_ += E[a][0].len & 0x7fffffffu;
            for (int ty = 0; ty < 16; ty++) { tab.push_back(E[a][ty].off); tab.push_back(E[a][ty].set ? E[a][ty].len : 0u); }
        }
        if (rc_ > 0xffffffffull || tab.size() > 0x7fffffffull) return HG_EINVAL;
        J.rec_cap = (uint32_t)rc_;
        recs += rc_ + 64; names += 14ull * (nn + 2ull);             // tok3.hip: PAR_ARRAYS words per name for the position-major kernel (the serial one uses three)
        ooff += ((uint64_t)ulen + 15u) & ~15ull;
        jobs.push_back(J); job_top.push_back((uint32_t)i);
    }
    if (P.too_big) return HG_EINVAL;
    const size_t nj = jobs.size();
    const uint64_t tbytes = (tboff + 63u) & ~63ull;
    int rc;
    if ((rc = ensure_scratch(ctx, 0, ioff + 64)) || (rc = ensure_scratch(ctx, 8, ooff + 64)) || (rc = ensure_scratch(ctx, 9, recs * 16 + 64)) ||
        (rc = ensure_scratch(ctx, 10, names * 4 + 64)) || (rc = ensure_scratch(ctx, 11, tab.size() * 4 + 64)) ||
        (rc = ensure_scratch(ctx, 12, nj * sizeof(hg::tok3_job) + nj * 4 + 64))) return rc;
    hipStream_t s = ctx->stream;
    uint8_t *d_in = (uint8_t *)ctx->d_scratch[0];
    bool ok = true;
    rc = hg::stage_upload(ctx, in, in_len, ioffs.data(), st.data(), n, ioff, d_in, s);
    if (rc == HG_OK) rc = launch_plan(ctx, P, ioff, tbytes, s);
    hg::tok3_job *d_jobs = (hg::tok3_job *)ctx->d_scratch[12];
    int32_t *d_jst = (int32_t *)(d_jobs + nj);
    if (rc == HG_OK && nj) {
        ok = hipMemcpyAsync(d_jobs, jobs.data(), nj * sizeof(hg::tok3_job), hipMemcpyHostToDevice, s) == hipSuccess &&
             hipMemcpyAsync(ctx->d_scratch[11], tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
             hipMemsetAsync(d_jst, 0xff, nj * 4, s) == hipSuccess;
        rc = ok ? hg::launch_tok3_names(ctx, ctx->d_scratch[1], d_jobs, nj, (const uint32_t *)ctx->d_scratch[11], ctx->d_scratch[8],
                                        (uint32_t *)ctx->d_scratch[9], (uint32_t *)ctx->d_scratch[10], d_jst, s) : HG_ELAUNCH;
    }
    if (rc == HG_OK) {
        std::vector<int32_t> jst(nj + 1);
        ok = (!nj || hipMemcpyAsync(jst.data(), d_jst, nj * 4, hipMemcpyDeviceToHost, s) == hipSuccess) && hipStreamSynchronize(s) == hipSuccess &&
             collect_plan_status(ctx, P, st, s);
        std::vector<uint64_t> so(nj); std::vector<uint32_t> sl(nj); std::vector<uint8_t *> sd(nj);
        for (size_t k = 0; k < nj && ok; k++) {
            const uint32_t i = job_top[k];
            if (st[i] == 0 && jst[k] != 0) st[i] = jst[k];
            so[k] = jobs[k].out_off; sl[k] = st[i] == 0 ? out_len[i] : 0u; sd[k] = out[i];
        }
        if (!ok) rc = HG_ELAUNCH;
        else rc = hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[8], so.data(), sl.data(), sd.data(), nj, s);
    }
    if (rc == HG_OK)
        for (size_t i = 0; i < n; i++) { if (status) status[i] = st[i]; if (st[i] != 0) rc = HG_EBLOCK; }
    return rc;
}

extern "C" int hg_ransnx16_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                                       uint8_t *const *out, const uint32_t *out_len, int32_t *status) {
    return entropy_decode_host(NX16, ctx, in, in_len, n, out, out_len, status);
}
extern "C" int hg_arith_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                                    uint8_t *const *out, const uint32_t *out_len, int32_t *status) {
    return entropy_decode_host(ARITH, ctx, in, in_len, n, out, out_len, status);
}

// ================================================================================================
// Encoder: hg_ransnx16_encode_host (replaces rans_compress_4x16, call site cram/cram_io.c:1853-1866).
// Mirror image of the decoder plan: STRIPE / PACK / RLE run first (ransnx16_xenc.hip), their results
// (how many literals, which symbols...) come back in one small D2H, then every piece that needs entropy
// coding -- the data of each leaf and each RLE meta stream -- goes through the core encoder
// (ransnx16_enc.hip) in one launch, and the host stitches headers and pieces together in the layout
// of oracle/ransnx16_oracle.c compress_inner() (byte-identical output).
// ================================================================================================
namespace {

struct Leaf {
    uint32_t top, n, stride, flags;     // flags as requested for this leaf
    uint32_t max_sym;                   // upper bound of the byte values (sizes the range coder's models)
    uint64_t src_off;                   // in the device buffer
    const uint8_t *host_src;            // non-null when the leaf is the untouched host input
    int xjob, core, mcore;
    hg::nx16_xenc_res r;
};

// An order-1 table costs at most ~4 bytes per distinct (context, symbol) pair plus the alphabet; four stripes each pay
// their own fixed part.  (hg_ransnx16_compress_bound keeps the reference-style worst case for callers.)
uint64_t nx16_tight_bound(uint64_t len) { return 7 * len + 40000; }

int put_u7(uint8_t *cp, uint32_t v) {
    uint8_t tmp[5]; int n = 0;
    do { tmp[n++] = v & 0x7f; v >>= 7; } while (v);
    for (int i = n - 1; i >= 0; i--) *cp++ = tmp[i] | (i ? 0x80 : 0);
    return n;
}

}  // namespace

// When d_src is given the inputs are device-resident (input i = d_src + d_src_off[i], copied device-to-device into
// the encoder's own staging buffer) and `in` is not read.
static int entropy_encode_host(Codec codec, hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *flags, size_t n,
                               uint8_t *const *out, uint32_t *out_len, const uint8_t *d_src = nullptr, const uint64_t *d_src_off = nullptr,
                               const std::function<void()> *place = nullptr) {
    // `place`: every out_len[] is filled in FIRST (all lengths are known once the kernels are done), then (*place)() runs and may rewrite out[]
    // (the caller owns that array); items whose out[i] is then null are not fetched at all.  A caller that keeps one of several trial encodings
    // of a stream (tok3) pays the PCIe transfer, the host copies and the page faults of a bound-sized buffer only for the one it keeps.
    if (!ctx || (n && ((!in && !d_src) || !in_len || !flags || !out || !out_len))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    std::vector<Leaf> leaves;
    std::vector<hg::nx16_xenc> xj;
    std::vector<uint64_t> ioffs(n);
    std::vector<uint32_t> first_leaf(n + 1);
    std::vector<uint8_t> top_flags(n);
    uint64_t ioff = 0;
    for (size_t i = 0; i < n; i++) { ioffs[i] = ioff; ioff += ((uint64_t)in_len[i] + 15u) & ~15ull; }
    uint64_t work = (ioff + 63u) & ~63ull;                                    // work area follows the inputs
    auto walloc = [&](uint64_t bytes) { const uint64_t o = work; work += (bytes + 31u) & ~15ull; return o; };
    for (size_t i = 0; i < n; i++) {
        first_leaf[i] = (uint32_t)leaves.size();
        uint32_t f = flags[i];
        if (codec == ARITH) f &= ~(uint32_t)F_EXT;
        const uint32_t S = (f & F_STRIPE) ? 4u : 1u;
        if (f & F_STRIPE) f &= codec == NX16 ? ~(uint32_t)(F_PACK | F_RLE | F_CAT) : ~(uint32_t)(F_PACK | F_CAT);
        top_flags[i] = (uint8_t)f;
        uint32_t host_max = 255;
        if (codec == ARITH && (f & F_ORDER) && !d_src) { host_max = 0; for (uint32_t q = 0; q < in_len[i]; q++) if (in[i][q] > host_max) host_max = in[i][q]; }
        for (uint32_t k = 0; k < S; k++) {
            Leaf L;
            memset(&L, 0, sizeof L);
            L.top = (uint32_t)i; L.stride = S; L.src_off = ioffs[i] + k;
            L.n = S == 1 ? in_len[i] : in_len[i] / S + ((in_len[i] % S) > k ? 1u : 0u);
            L.flags = S == 1 ? f : ((f & (codec == NX16 ? (F_ORDER | F_X32) : (F_ORDER | F_RLE))) | F_NOSZ);
            L.max_sym = (L.flags & F_PACK) ? 255u : host_max;
            L.host_src = (S == 1 && !d_src) ? in[i] : nullptr;
            L.xjob = L.core = L.mcore = -1;
            const uint32_t xops = L.flags & (codec == NX16 ? (F_PACK | F_RLE) : F_PACK);     // the range coder's RLE is not a transform
            if (S != 1 || xops) {
                hg::nx16_xenc J;
                memset(&J, 0, sizeof J);
                J.src_off = L.src_off; J.n = L.n; J.stride = S; J.flags = xops;
                if (S != 1) J.g_off = walloc(L.n);
                if (xops & F_PACK) J.p_off = walloc(L.n / 2 + 8);
                if (xops & F_RLE) { J.l_off = walloc(L.n); J.m_off = walloc((uint64_t)L.n + 300); }
                L.xjob = (int)xj.size(); xj.push_back(J);
            }
            leaves.push_back(L);
        }
    }
    first_leaf[n] = (uint32_t)leaves.size();
    int rc;
    if ((rc = ensure_scratch(ctx, 0, work + 64)) || (rc = ensure_scratch(ctx, 4, xj.size() * (sizeof(hg::nx16_xenc) + sizeof(hg::nx16_xenc_res)) + 64))) return rc;
    hipStream_t s = ctx->stream;
    uint8_t *d_buf = (uint8_t *)ctx->d_scratch[0];
    bool ok = true;
    rc = d_src ? hg::stage_gather_dev(ctx, d_src, d_src_off, in_len, d_buf, ioffs.data(), n, s)
               : hg::stage_upload(ctx, in, in_len, ioffs.data(), nullptr, n, ioff, d_buf, s);
    if (rc) return rc;
    std::vector<hg::nx16_xenc_res> xr(xj.size());
    if (ok && !xj.empty()) {
        hg::nx16_xenc *d_j = (hg::nx16_xenc *)ctx->d_scratch[4];
        hg::nx16_xenc_res *d_r = (hg::nx16_xenc_res *)(d_j + xj.size());
        ok = hipMemcpyAsync(d_j, xj.data(), xj.size() * sizeof(hg::nx16_xenc), hipMemcpyHostToDevice, s) == hipSuccess;
        if (ok && (rc = hg::launch_ransnx16_xenc(ctx, d_buf, d_j, xj.size(), d_r, s, xj.data()))) return rc;
        ok = ok && hipMemcpyAsync(xr.data(), d_r, xj.size() * sizeof(hg::nx16_xenc_res), hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipStreamSynchronize(s) == hipSuccess;
    }
    if (!ok) return HG_ELAUNCH;
    // ---- entropy-coding jobs ------------------------------------------------------------------
    std::vector<hg_stream_desc> cd;
    std::vector<uint8_t> cfl;
    uint64_t ooff = 0, soff = 0, woff = 0;
    bool too_big = false;
    std::vector<uint8_t> ccls;
    // The range coder's encoder runs in two phases (arith_enc2.hip) from HG_ARITH_2P_MIN bytes on: the models' chains side by side (order 1, RLE), and even for
    // a single model (order 0) the register-model pass + the scalar coder pass are quicker than the tangled one-pass step (0.12 + 0.15 us per symbol against
    // 0.5 .. 0.95; profiles/r05_arith_two_phase.txt).  HG_ARITH_2P=0: one pass for everything (A/B runs); HG_ARITH_2P_MIN=<bytes>: the threshold (tests: 1).
    const bool two_phase = !(getenv("HG_ARITH_2P") && atoi(getenv("HG_ARITH_2P")) == 0);
    const bool two_phase_forced = getenv("HG_ARITH_2P_MIN") && atoi(getenv("HG_ARITH_2P_MIN")) > 0;
    const uint32_t two_phase_min = two_phase_forced ? (uint32_t)atoi(getenv("HG_ARITH_2P_MIN")) : HG_ARITH_2P_MIN;
    // Work memory of the two-phase form is 12 bytes per input byte (26 with RLE), per trial variant: a call gets a budget -- a quarter of the free device memory --
    // and the streams beyond it take the one-pass kernels (they need none).
    uint64_t budget_2p = 0;
    if (two_phase && codec == ARITH) { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget_2p = fr / 4 + ctx->d_scratch_cap[5]; }   // (+ what the context's work buffer already holds)
    size_t task_cap = 0;
    auto add_core = [&](uint64_t src_off, uint32_t len, uint32_t fl, Codec cc, uint32_t max_sym) {
        hg_stream_desc d;
        memset(&d, 0, sizeof d);
        const uint64_t cap = cc == ARITH ? (uint64_t)len + len / 4 + 4096
                           : (fl & F_ORDER) ? nx16_tight_bound(len) : (uint64_t)len + len / 16 + 4096;
        d.in_off = src_off; d.in_len = len; d.out_off = ooff; d.out_len = (uint32_t)(cap > 0xffffffffull ? 0xffffffffu : cap);
        d.scratch_off = (uint32_t)soff; d.reserved = (uint32_t)(woff / 16);
        ooff += (cap + 15u) & ~15ull;
        const uint64_t need_2p = cc == ARITH ? hg::arith2p_layout(len, (fl & F_RLE) != 0).end : 0;
        if (cc == ARITH && two_phase && len >= two_phase_min && len < HG_ARITH_2P_MAX && need_2p <= budget_2p) {
            ccls.push_back(C_ARITH_2P);
            budget_2p -= need_2p;
            soff += HG_ARITH_2P_INFO_WORDS;
            woff += need_2p;
            task_cap += hg::arith2p_max_tasks(fl);
        } else if (cc == ARITH) {
            const uint32_t words = hg::arith_model_words(max_sym + 1u, fl);
            ccls.push_back(words <= HG_ARITH_POOL_SMALL ? C_ARITH_SMALL : C_ARITH_BIG);
            soff += words > HG_ARITH_POOL_BIG ? words + 16 : 16;
        } else {
            ccls.push_back((fl & F_X32) ? C_NX32 : C_NX4);
            soff += hg::ransnx16_enc_scratch_words(fl);
            woff += (2ull * len + 256u + 15u) & ~15ull;
        }
        if (soff > 0xffffffffull || woff / 16 > 0xffffffffull) too_big = true;
        cd.push_back(d); cfl.push_back((uint8_t)fl);
        return (int)cd.size() - 1;
    };
    for (Leaf &L : leaves) {
        if (L.xjob >= 0) { const uint32_t keep = L.flags & ~(uint32_t)(codec == NX16 ? (F_PACK | F_RLE) : F_PACK); L.r = xr[L.xjob]; L.r.flags |= keep; }
        else { L.r.flags = L.flags; L.r.cur_off = L.src_off; L.r.cur_len = L.n; }
        if (codec == NX16) {
            const uint32_t N = (L.r.flags & F_X32) ? 32u : 4u;
            if ((L.r.flags & F_ORDER) && L.r.cur_len < 2u * N) L.r.flags &= ~(uint32_t)F_ORDER;
            if (L.r.flags & F_RLE) L.mcore = add_core(xj[L.xjob].m_off, L.r.meta_len, F_NOSZ, NX16, 255);
            if (!(L.r.flags & F_CAT) && L.r.cur_len) L.core = add_core(L.r.cur_off, L.r.cur_len, (L.r.flags & (F_ORDER | F_X32)) | F_NOSZ, NX16, 255);
        } else if (!(L.r.flags & F_CAT) && L.r.cur_len)
            L.core = add_core(L.r.cur_off, L.r.cur_len, L.r.flags & (F_ORDER | F_RLE), ARITH, (L.r.flags & F_PACK) ? 255u : L.max_sym);
    }
    if (too_big) return HG_EINVAL;
    const size_t nc = cd.size();
    std::vector<uint32_t> sel(nc), ol(nc, 0);
    size_t cnt[C_CLASSES] = {0}, first[C_CLASSES + 1] = {0};
    for (size_t k = 0; k < nc; k++) cnt[ccls[k]]++;
    for (int c = 0; c < C_CLASSES; c++) first[c + 1] = first[c] + cnt[c];
    { size_t fill[C_CLASSES]; for (int c = 0; c < C_CLASSES; c++) fill[c] = first[c];
      for (size_t k = 0; k < nc; k++) sel[fill[ccls[k]]++] = (uint32_t)k; }
    // Streams that share a wavefront (2 or 16 lane groups) run in lock step: groups that take different code paths (order 0 /
    // order 1) are executed one after the other and a short stream waits for a long neighbour.  Neighbours are therefore
    // made alike: same flags first, then by length.
    for (int c = 0; c < C_CLASSES; c++)
        std::stable_sort(sel.begin() + first[c], sel.begin() + first[c + 1], [&](uint32_t a, uint32_t b) {
            if (cfl[a] != cfl[b]) return cfl[a] < cfl[b];
            return cd[a].in_len > cd[b].in_len;
        });
    uint8_t *d_out = nullptr;
    if (nc) {
        if ((rc = ensure_scratch(ctx, 1, ooff + 64)) || (rc = ensure_scratch(ctx, 2, nc * sizeof(hg_stream_desc) + 64)) ||
            (rc = ensure_scratch(ctx, 3, nc * 5 + 64)) || (rc = ensure_scratch(ctx, 5, woff + 64)) ||
            (rc = ensure_scratch(ctx, 6, soff * 4 + 64)) || (rc = ensure_scratch(ctx, 7, nc * 4 + 64))) return rc;
        d_out = (uint8_t *)ctx->d_scratch[1];
        uint32_t *d_ol = (uint32_t *)ctx->d_scratch[3];
        uint8_t *d_fl = (uint8_t *)ctx->d_scratch[3] + nc * 4;
        ok = hipMemsetAsync(d_ol, 0, nc * 4, s) == hipSuccess &&
             hipMemcpyAsync(ctx->d_scratch[2], cd.data(), nc * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
             hipMemcpyAsync(ctx->d_scratch[7], sel.data(), nc * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
             hipMemcpyAsync(d_fl, cfl.data(), nc, hipMemcpyHostToDevice, s) == hipSuccess;
        if (!ok) return HG_ELAUNCH;
        const uint32_t *d_sel = (const uint32_t *)ctx->d_scratch[7];
        rc = HG_OK;
        if (cnt[C_NX4] + cnt[C_NX32])
            rc = hg::launch_ransnx16_encode(ctx, d_buf, (const hg_stream_desc *)ctx->d_scratch[2], d_fl, d_sel + first[C_NX4], cnt[C_NX4],
                                            d_sel + first[C_NX32], cnt[C_NX32], d_out, d_ol, ctx->d_scratch[5], (uint32_t *)ctx->d_scratch[6], s);
        bool forked4 = false;
        if (rc == HG_OK && cnt[C_ARITH_2P]) {
            // the device builds its own task list (sort kernel): 16 counter words + room for every model of every stream
            rc = ensure_scratch(ctx, 13, 64 + task_cap * 8 + 64);
            if (rc == HG_OK) {
                const bool others = cnt[C_NX4] + cnt[C_NX32] + cnt[C_ARITH_SMALL] + cnt[C_ARITH_BIG] != 0;
                hipStream_t s4 = others ? hg::fork_side4(ctx, s) : s;      // the one-pass kernels of this call run beside the two phases
                forked4 = others;
                // (the record areas start out zero: a record's second word, the model's total, doubles as its "written" mark for the coder that follows the models)
                if (hipMemsetAsync(ctx->d_scratch[5], 0, woff, s4) != hipSuccess) rc = HG_ELAUNCH;
                if (rc == HG_OK)
                rc = hg::launch_arith_encode2(ctx, d_buf, (const hg_stream_desc *)ctx->d_scratch[2], d_fl, d_sel + first[C_ARITH_2P], cnt[C_ARITH_2P], ctx->d_scratch[13], task_cap,
                                              d_out, d_ol, (uint32_t *)ctx->d_scratch[6], ctx->d_scratch[5], s4);
            }
        }
        if (rc == HG_OK && cnt[C_ARITH_SMALL] + cnt[C_ARITH_BIG])
            rc = hg::launch_arith_encode(ctx, d_buf, (const hg_stream_desc *)ctx->d_scratch[2], d_fl, d_sel + first[C_ARITH_SMALL], cnt[C_ARITH_SMALL],
                                         d_sel + first[C_ARITH_BIG], cnt[C_ARITH_BIG], d_out, d_ol, (uint32_t *)ctx->d_scratch[6], s);
        if (forked4) hg::join_side4(ctx, s);                                // (on every path after the fork)
        if (rc) return rc;
        if (hipMemcpyAsync(ol.data(), d_ol, nc * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        // a two-phase stream always has at least its alphabet byte: 0 = its coder gave up waiting for records (a logic error, reported loudly, never as bytes)
        for (size_t k = 0; k < nc; k++)
            if (ccls[k] == C_ARITH_2P && ol[k] == 0) {
                fprintf(stderr, "[hts-gpu] range coder: the coder pass of core stream %zu of %zu (two-phase encoder) ran out of its poll budget waiting for model records; the batch fails\n", k, nc);
                return HG_ELAUNCH;
            }
    }
    // ---- stitch --------------------------------------------------------------------------------
    // Payload pieces live in two device buffers (0: d_buf -- raw RLE meta and CAT data, 1: d_out -- entropy-coder output).
    // The headers are written now; the pieces are only RECORDED (every length is already known on the host) and
    // then fetched with one gather + one PCIe transfer per buffer.
    struct Fetch { std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint8_t *> dst; } F[2];
    auto fetch = [&](uint8_t *dst, int which, uint64_t off, size_t len) {
        if (len && dst) { F[which].off.push_back(off); F[which].len.push_back((uint32_t)len); F[which].dst.push_back(dst); }
    };
    // cp == nullptr: measure only.  Returns the leaf's length.
    auto emit_leaf = [&](uint8_t *cp, const Leaf &L) -> size_t {
        const bool wr = cp != nullptr;
        uint8_t hdr[64], *hp = hdr;
        const uint32_t f = L.r.flags;
        *hp++ = (uint8_t)f;
        if (!(f & F_NOSZ)) hp += put_u7(hp, L.n);
        if (f & F_PACK) { *hp++ = (uint8_t)L.r.nsym; memcpy(hp, L.r.map, L.r.nsym); hp += L.r.nsym; hp += put_u7(hp, L.r.plen); }
        size_t pos = 0;
        auto flush_hdr = [&]() { const size_t hl = (size_t)(hp - hdr); if (wr) memcpy(cp + pos, hdr, hl); pos += hl; hp = hdr; };
        if (codec == NX16 && (f & F_RLE)) {
            const uint32_t ml = L.r.meta_len, cl = ol[L.mcore] ? ol[L.mcore] - 1u : 0u;   // core output minus its flag byte
            if (cl + 5u < ml) {
                hp += put_u7(hp, ml * 2u); hp += put_u7(hp, L.r.lit_len); hp += put_u7(hp, cl);
                flush_hdr();
                fetch(wr ? cp + pos : nullptr, 1, cd[L.mcore].out_off + 1, cl); pos += cl;
            } else {
                hp += put_u7(hp, ml * 2u + 1u); hp += put_u7(hp, L.r.lit_len);
                flush_hdr();
                fetch(wr ? cp + pos : nullptr, 0, xj[L.xjob].m_off, ml); pos += ml;
            }
        } else flush_hdr();
        if (f & F_CAT) {
            if (L.xjob < 0 && L.host_src) { if (wr) memcpy(cp + pos, L.host_src, L.r.cur_len); }
            else fetch(wr ? cp + pos : nullptr, 0, L.r.cur_off, L.r.cur_len);
            pos += L.r.cur_len;
        } else if (L.core >= 0) {
            const uint32_t skip = codec == NX16 ? 1u : 0u;                    // the Nx16 core kernel writes its own flag byte first
            const uint32_t bl = ol[L.core] > skip ? ol[L.core] - skip : 0u;
            fetch(wr ? cp + pos : nullptr, 1, cd[L.core].out_off + skip, bl); pos += bl;
        }
        return pos;
    };
    if (place) {
        auto u7_len = [](uint32_t v) { size_t k = 1; while (v >>= 7) k++; return k; };
        for (size_t i = 0; i < n; i++) {
            const uint32_t l0 = first_leaf[i], nl = first_leaf[i + 1] - l0;
            size_t len = 0;
            if (nl == 1) len = emit_leaf(nullptr, leaves[l0]);
            else {
                len = 2 + ((top_flags[i] & F_NOSZ) ? 0 : u7_len(in_len[i]));
                for (uint32_t k = 0; k < nl; k++) { const size_t ll = emit_leaf(nullptr, leaves[l0 + k]); len += u7_len((uint32_t)ll) + ll; }
            }
            out_len[i] = (uint32_t)len;
        }
        (*place)();
    }
    for (size_t i = 0; i < n; i++) {
        uint8_t *cp = out[i];
        if (place && !cp) continue;
        const uint32_t l0 = first_leaf[i], nl = first_leaf[i + 1] - l0;
        if (nl == 1) cp += emit_leaf(cp, leaves[l0]);
        else {
            const uint32_t f = top_flags[i];
            *cp++ = (uint8_t)f;
            if (!(f & F_NOSZ)) cp += put_u7(cp, in_len[i]);
            *cp++ = (uint8_t)nl;
            for (uint32_t k = 0; k < nl; k++) cp += put_u7(cp, (uint32_t)emit_leaf(nullptr, leaves[l0 + k]));
            for (uint32_t k = 0; k < nl; k++) cp += emit_leaf(cp, leaves[l0 + k]);
        }
        out_len[i] = (uint32_t)(cp - out[i]);
    }
    if (!F[0].off.empty() && (rc = hg::stage_download(ctx, d_buf, F[0].off.data(), F[0].len.data(), F[0].dst.data(), F[0].off.size(), s))) return rc;
    if (!F[1].off.empty() && (rc = hg::stage_download(ctx, d_out, F[1].off.data(), F[1].len.data(), F[1].dst.data(), F[1].off.size(), s))) return rc;
    return ok ? HG_OK : HG_ELAUNCH;
}

extern "C" int hg_ransnx16_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *flags, size_t n,
                                       uint8_t *const *out, uint32_t *out_len) {
    return entropy_encode_host(NX16, ctx, in, in_len, flags, n, out, out_len);
}
extern "C" size_t hg_arith_compress_bound(size_t n) { return n + n / 4 + 4 * 4096 + 64; }
extern "C" int hg_arith_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *flags, size_t n,
                                    uint8_t *const *out, uint32_t *out_len) {
    return entropy_encode_host(ARITH, ctx, in, in_len, flags, n, out, out_len);
}

// ================================================================================================
// tok3 encode: hg_tok3_encode_host replaces tok3_encode_names (call site cram/cram_io.c:1885-1895).
// tok3.hip tokenises the names into (position, type) byte streams on the device; each stream is then entropy-coded
// with every setting of a short list (the same list as oracle/tok3_oracle.c best_entropy) through the Nx16 / range-coder
// encoders above -- inputs stay in HBM -- and the smallest result is kept.  The host only stitches the container.
// ================================================================================================
extern "C" size_t hg_tok3_compress_bound(size_t n) { return n * 2 + 65536 + 13 * 128 * 64; }

extern "C" int hg_tok3_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *use_arith, size_t n,
                                   uint8_t *const *out, uint32_t *out_len) {
    if (!ctx || (n && (!in || !in_len || !use_arith || !out || !out_len))) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hipStream_t s = ctx->stream;
    size_t i0 = 0;
    while (i0 < n) {
        // one group of blocks per round keeps the host-side trial buffers bounded
        size_t i1 = i0; uint64_t sum = 0;
        while (i1 < n && (i1 == i0 || sum + in_len[i1] <= (64u << 20))) { sum += in_len[i1]; i1++; }
        std::vector<hg::tok3_enc_job> jobs;
        std::vector<size_t> job_blk;
        uint64_t ioff = 0, sboff = 0;
        for (size_t i = i0; i < i1; i++) {
            out_len[i] = 0;
            if (in_len[i] == 0) {                                       // no names: header only
                memset(out[i], 0, 9); out[i][8] = use_arith[i] ? 1 : 0; out_len[i] = 9; continue;
            }
            if (in[i][in_len[i] - 1] != 0) continue;                    // not a list of NUL-terminated names: out_len 0
            hg::tok3_enc_job J;
            J.in_off = ioff; J.sb_off = sboff; J.n = in_len[i]; J.sb_cap = 8u * in_len[i] + 1024u;
            if ((uint64_t)in_len[i] * 8 + 1024 > 0x7fffffffull) return HG_EINVAL;
            ioff += ((uint64_t)in_len[i] + 15u) & ~15ull; sboff += ((uint64_t)J.sb_cap + 15u) & ~15ull;
            jobs.push_back(J); job_blk.push_back(i);
        }
        const size_t nj = jobs.size();
        if (nj) {
            int rc;
            if ((rc = ensure_scratch(ctx, 8, ioff + 64)) || (rc = ensure_scratch(ctx, 9, sboff + 64)) ||
                (rc = ensure_scratch(ctx, 10, nj * (size_t)HG_TOK3_MAX_STREAMS * sizeof(hg::tok3_enc_stream) + 64)) ||
                (rc = ensure_scratch(ctx, 11, nj * (sizeof(hg::tok3_enc_job) + sizeof(hg::tok3_enc_res)) + 64))) return rc;
            uint8_t *d_names = (uint8_t *)ctx->d_scratch[8], *d_sb = (uint8_t *)ctx->d_scratch[9];
            hg::tok3_enc_stream *d_list = (hg::tok3_enc_stream *)ctx->d_scratch[10];
            hg::tok3_enc_job *d_jobs = (hg::tok3_enc_job *)ctx->d_scratch[11];
            hg::tok3_enc_res *d_res = (hg::tok3_enc_res *)(d_jobs + nj);
            bool ok = hipMemcpyAsync(d_jobs, jobs.data(), nj * sizeof(hg::tok3_enc_job), hipMemcpyHostToDevice, s) == hipSuccess;
            for (size_t k = 0; k < nj && ok; k++)
                ok = hipMemcpyAsync(d_names + jobs[k].in_off, in[job_blk[k]], jobs[k].n, hipMemcpyHostToDevice, s) == hipSuccess;
            if (!ok) return HG_ELAUNCH;
            if ((rc = hg::launch_tok3_tokenise(ctx, d_names, d_jobs, nj, d_sb, d_list, d_res, s))) return rc;
            std::vector<hg::tok3_enc_res> res(nj);
            if (hipMemcpyAsync(res.data(), d_res, nj * sizeof(hg::tok3_enc_res), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
            std::vector<std::vector<hg::tok3_enc_stream>> lists(nj);
            {
                // the stream lists of all jobs in ONE transfer (a blocking copy per job was several ms per call of a slice batch)
                std::vector<hg::tok3_enc_stream> all(nj * (size_t)HG_TOK3_MAX_STREAMS);
                if (hipMemcpy(all.data(), d_list, all.size() * sizeof(hg::tok3_enc_stream), hipMemcpyDeviceToHost) != hipSuccess) return HG_ELAUNCH;
                for (size_t k = 0; k < nj; k++) {
                    if (res[k].total == 0xffffffffu || res[k].nstreams > HG_TOK3_MAX_STREAMS) return HG_ELAUNCH;   // cannot happen: 8 bytes/char is the worst case
                    lists[k].assign(all.begin() + k * (size_t)HG_TOK3_MAX_STREAMS, all.begin() + k * (size_t)HG_TOK3_MAX_STREAMS + res[k].nstreams);
                }
            }
            // ---- entropy trials: one item per (stream, setting) ----------------------------------------------
            static const uint8_t sets[8] = {0, 1, 64, 65, 128, 129, 8, 9};
            // The blocks of the rANS back-end (TOK3) and of the range coder's (TOKA) are independent: the second half runs on a thread and sibling context of
            // its own (the token streams lie in this context's device buffer, which both read), so their kernels and their host planning overlap.
            auto run_codec = [&](int codec, hg_ctx *cctx) -> int {
                int rc;
                std::vector<uint64_t> soff; std::vector<uint32_t> slen; std::vector<uint8_t> sfl;
                std::vector<std::pair<uint32_t, uint32_t>> owner;          // (job, stream index)
                for (size_t k = 0; k < nj; k++) {
                    if ((use_arith[job_blk[k]] ? 1 : 0) != codec) continue;
                    for (uint32_t q = 0; q < lists[k].size(); q++) {
                        const hg::tok3_enc_stream &E = lists[k][q];
                        if (E.ttype & 0x40) continue;
                        const int nsets = (E.type == 7 || E.type == 3 || E.type == 5 || E.type == 6) ? 8 : 6;
                        for (int v = 0; v < nsets; v++) { soff.push_back(jobs[k].sb_off + E.off); slen.push_back(E.len); sfl.push_back(sets[v]); owner.push_back({(uint32_t)k, q}); }
                    }
                }
                const size_t ni = soff.size();
                if (!ni) return HG_OK;
                // Only the smallest setting of every stream (the first one on ties) is fetched: the encoder reports all lengths, `keep` picks the winners and
                // gives them room in one compact buffer (a bound-sized slot per trial was ~1 GiB of fresh pages per slice batch, touched once each).
                uint8_t *arena = nullptr; bool nomem = false;
                std::vector<uint8_t *> optr(ni, nullptr); std::vector<uint32_t> olen(ni, 0);
                std::vector<std::vector<std::pair<const uint8_t *, uint32_t>>> best(nj);
                for (size_t k = 0; k < nj; k++) best[k].assign(lists[k].size(), {nullptr, 0});
                const std::function<void()> keep = [&]() {
                    std::vector<size_t> win;
                    uint64_t tot = 0;
                    for (size_t t = 0; t < ni;) {
                        size_t e = t, b = t;
                        while (e < ni && owner[e] == owner[t]) { if (olen[e] < olen[b]) b = e; e++; }
                        win.push_back(b); tot += olen[b];
                        t = e;
                    }
                    arena = (uint8_t *)malloc(tot + 64);
                    if (!arena) { nomem = true; return; }
                    uint64_t o = 0;
                    for (size_t b : win) { optr[b] = arena + o; best[owner[b].first][owner[b].second] = {optr[b], olen[b]}; o += olen[b]; }
                };
                rc = entropy_encode_host(codec ? ARITH : NX16, cctx, nullptr, slen.data(), sfl.data(), ni, optr.data(), olen.data(), d_sb, soff.data(), &keep);
                if (rc || nomem) { free(arena); return rc ? rc : HG_ENOMEM; }
                for (size_t k = 0; k < nj; k++) {
                    if ((use_arith[job_blk[k]] ? 1 : 0) != codec) continue;
                    const size_t i = job_blk[k];
                    uint8_t *cp = out[i];
                    for (int b = 0; b < 4; b++) *cp++ = (uint8_t)(in_len[i] >> (8 * b));
                    for (int b = 0; b < 4; b++) *cp++ = (uint8_t)(res[k].nn >> (8 * b));
                    *cp++ = (uint8_t)codec;
                    for (uint32_t q = 0; q < lists[k].size(); q++) {
                        const hg::tok3_enc_stream &E = lists[k][q];
                        *cp++ = E.ttype;
                        if (E.ttype & 0x40) { *cp++ = E.dup_pos; *cp++ = E.dup_type; continue; }
                        cp += put_u7(cp, best[k][q].second);
                        memcpy(cp, best[k][q].first, best[k][q].second); cp += best[k][q].second;
                    }
                    out_len[i] = (uint32_t)(cp - out[i]);
                }
                free(arena);
                return HG_OK;
            };
            bool have[2] = {false, false};
            for (size_t k = 0; k < nj; k++) have[use_arith[job_blk[k]] ? 1 : 0] = true;
            if (have[0] && have[1]) {
                if (!ctx->sub[0] && hg_init(ctx->device, &ctx->sub[0]) != HG_OK) return HG_ENOMEM;
                int rc1 = HG_OK;
                std::thread side([&] { rc1 = hipSetDevice(ctx->device) == hipSuccess ? run_codec(1, ctx->sub[0]) : HG_ENODEV; });
                const int rc0 = run_codec(0, ctx);
                side.join();
                if (rc0 || rc1) return rc0 ? rc0 : rc1;
            } else if ((rc = run_codec(have[1] ? 1 : 0, ctx))) return rc;
        }
        i0 = i1;
    }
    return HG_OK;
}

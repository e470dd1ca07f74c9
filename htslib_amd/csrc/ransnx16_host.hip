// ransnx16_host.hip -- host planner for CRAM 3.1 rANS Nx16 block decoding (hg_ransnx16_decode_host).
//
// Replaces rans_uncompress_4x16 at its call site in cram_uncompress_block (reference
// cram/cram_io.c:1697-1714).  An Nx16 stream is a small tree: STRIPE splits it into S complete
// sub-streams, and every leaf is  [PACK header] [RLE header + meta stream] entropy-coded core.
// The planner walks the few header bytes of every stream on the host (no payload byte is touched
// here), and emits
//   * "core" jobs  -- entropy decode (rANS order 0/1, 4- or 32-way, or CAT) : ransnx16.hip
//   * "xform" jobs -- RLE expand / bit unpack / strided (de-striping) write : ransnx16_xform.hip
// then runs the two kernels back to back on one HIP stream.  All payload work is on the GPU; there
// is no CPU decode path.  Header rules follow oracle/ransnx16_oracle.c (PARITY UNPINNED).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "htsgpu.h"
#include "hg_internal.h"

using hg::ensure_scratch;
namespace {

enum { F_ORDER = 1, F_X32 = 4, F_STRIPE = 8, F_NOSZ = 16, F_CAT = 32, F_RLE = 64, F_PACK = 128 };
constexpr uint32_t NONE = 0xffffffffu;

struct Plan {
    std::vector<hg_stream_desc> core;      // out_off is into the work buffer, or (bit 63 set) the output buffer
    std::vector<uint32_t> core_top;
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
uint32_t add_core(Plan &P, uint32_t top, uint64_t in_base, const uint8_t *base, const uint8_t *cp, const uint8_t *end,
                  uint32_t flags, uint32_t out_len, uint64_t out_off) {
    hg_stream_desc d;
    memset(&d, 0, sizeof d);
    d.in_off = in_base + (uint64_t)(cp - base);
    d.in_len = (uint32_t)(end - cp);
    d.out_off = out_off; d.out_len = out_len;
    d.reserved = 0x80000000u | (flags & (F_ORDER | F_X32 | F_CAT));
    d.scratch_off = (uint32_t)P.scratch;
    P.scratch += ((flags & F_ORDER) && !(flags & F_CAT)) ? o1_scratch_words(cp, end) : 16;
    if (P.scratch > 0xffffffffull) P.too_big = true;
    P.core.push_back(d); P.core_top.push_back(top);
    return (uint32_t)P.core.size() - 1;
}

// Plans one (sub-)stream.  known = size when the caller knows it (NOSZ).  out_off/stride place byte i of this
// stream at output offset out_off + i*stride.  Returns 0, -1 (malformed).
int plan_stream(Plan &P, uint32_t top, uint64_t in_base, const uint8_t *base, const uint8_t *cp, const uint8_t *end,
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
            if (plan_stream(P, top, in_base, base, cp, cp + cl[k], m, out_off + (uint64_t)k * stride, stride * S, depth + 1)) return -1;
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
    if (flags & F_RLE) {
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
            dep1 = add_core(P, top, in_base, base, cp, cp + cl, 0, meta_len, meta_off);
            cp += cl;
        }
    }
    const bool xform = (flags & (F_RLE | F_PACK)) || stride != 1;
    if (!xform) {                                         // plain core straight into the output buffer
        if (ulen) add_core(P, top, in_base, base, cp, end, flags, ulen, out_off | (1ull << 63));
        return 0;
    }
    hg::nx16_xform J;
    memset(&J, 0, sizeof J);
    J.s1_off = work_alloc(P, lit_len);
    J.dep0 = lit_len ? add_core(P, top, in_base, base, cp, end, flags, lit_len, J.s1_off) : NONE;
    J.dep1 = dep1;
    J.meta_off = meta_off; J.meta_len = meta_len;
    J.lit_len = lit_len; J.plen = plen; J.ulen = ulen;
    J.ops = ((flags & F_RLE) ? 1u : 0u) | ((flags & F_PACK) ? 2u : 0u) | (meta_in_work ? 4u : 0u);
    if ((flags & F_RLE) && (flags & F_PACK)) J.s2_off = work_alloc(P, plen);
    J.out_off = out_off; J.stride = stride; J.nsym = nsym;
    memcpy(J.map, map, 16);
    P.xf.push_back(J); P.xf_top.push_back(top);
    return 0;
}

}  // namespace

extern "C" int hg_ransnx16_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
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
        if (plan_stream(P, (uint32_t)i, ioff, in[i], in[i], in[i] + in_len[i], out_len[i], ooff, 1, 0)) {
            st[i] = -1;                                   // malformed header: drop whatever was planned for it
            P.core.resize(c0); P.core_top.resize(c0); P.xf.resize(x0); P.xf_top.resize(x0); P.work = w0; P.scratch = s0;
        }
        ioff += ((uint64_t)in_len[i] + 15u) & ~15ull;
        ooff += ((uint64_t)out_len[i] + 15u) & ~15ull;
    }
    if (P.too_big) return HG_EINVAL;
    const size_t nc = P.core.size(), nx = P.xf.size();
    // the core kernel addresses ONE output allocation: [ output buffer | work buffer ]
    const uint64_t obytes = (ooff + 63u) & ~63ull;
    std::vector<uint32_t> sel(nc);
    size_t n4 = 0, n32 = 0;
    for (size_t k = 0; k < nc; k++) {
        hg_stream_desc &d = P.core[k];
        if (d.out_off >> 63) d.out_off &= ~(1ull << 63); else d.out_off += obytes;
        if (d.reserved & F_X32) n32++; else n4++;
    }
    { size_t a = 0, b = n4; for (size_t k = 0; k < nc; k++) { if (P.core[k].reserved & F_X32) sel[b++] = (uint32_t)k; else sel[a++] = (uint32_t)k; } }
    int rc;
    const size_t nst = nc + nx;
    if ((rc = ensure_scratch(ctx, 0, ioff + 64)) || (rc = ensure_scratch(ctx, 1, obytes + P.work + 64)) ||
        (rc = ensure_scratch(ctx, 2, nc * sizeof(hg_stream_desc) + 64)) || (rc = ensure_scratch(ctx, 3, nst * 4 + 64)) ||
        (rc = ensure_scratch(ctx, 4, nx * sizeof(hg::nx16_xform) + 64)) ||
        (rc = ensure_scratch(ctx, 6, P.scratch * 4 + 64)) || (rc = ensure_scratch(ctx, 7, nc * 4 + 64))) return rc;
    hipStream_t s = nullptr;
    uint8_t *d_in = (uint8_t *)ctx->d_scratch[0], *d_out = (uint8_t *)ctx->d_scratch[1];
    int32_t *d_st = (int32_t *)ctx->d_scratch[3];
    bool ok = hipMemsetAsync(d_st, 0xff, nst * 4 + 4, s) == hipSuccess;
    for (size_t i = 0; i < n && ok; i++)
        if (in_len[i] && st[i] == 0) ok = hipMemcpyAsync(d_in + ioffs[i], in[i], in_len[i], hipMemcpyHostToDevice, s) == hipSuccess;
    if (nc) ok = ok && hipMemcpyAsync(ctx->d_scratch[2], P.core.data(), nc * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(ctx->d_scratch[7], sel.data(), nc * 4, hipMemcpyHostToDevice, s) == hipSuccess;
    if (nx) ok = ok && hipMemcpyAsync(ctx->d_scratch[4], P.xf.data(), nx * sizeof(hg::nx16_xform), hipMemcpyHostToDevice, s) == hipSuccess;
    rc = ok ? HG_OK : HG_ELAUNCH;
    if (rc == HG_OK && nc)
        rc = hg::launch_ransnx16_decode(ctx, d_in, (const hg_stream_desc *)ctx->d_scratch[2], (const uint32_t *)ctx->d_scratch[7], n4,
                                        (const uint32_t *)ctx->d_scratch[7] + n4, n32, d_out, d_st, (uint32_t *)ctx->d_scratch[6], s);
    if (rc == HG_OK && nx)
        rc = hg::launch_ransnx16_xform(ctx, d_in, d_out + obytes, d_out, (const hg::nx16_xform *)ctx->d_scratch[4], nx, d_st, (uint32_t)nc, s);
    if (rc == HG_OK) {
        std::vector<int32_t> jst(nst + 1);
        ok = hipMemcpyAsync(jst.data(), d_st, nst * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        for (size_t k = 0; k < nc && ok; k++) if (jst[k] != 0 && st[P.core_top[k]] == 0) st[P.core_top[k]] = jst[k];
        for (size_t k = 0; k < nx && ok; k++) if (jst[nc + k] != 0 && st[P.xf_top[k]] == 0) st[P.xf_top[k]] = jst[nc + k];
        for (size_t i = 0; i < n && ok; i++)
            if (out_len[i] && st[i] == 0) ok = hipMemcpy(out[i], d_out + ooffs[i], out_len[i], hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok) rc = HG_ELAUNCH;
    }
    if (rc == HG_OK)
        for (size_t i = 0; i < n; i++) { if (status) status[i] = st[i]; if (st[i] != 0) rc = HG_EBLOCK; }
    return rc;
}

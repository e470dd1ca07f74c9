// htsgpu_api.hip -- the C-ABI shim declared in include/htsgpu.h.
//
// Thin host layer: context lifetime, the BGZF framing scan that replaces the
// header walk of bgzf_mt_read_block (reference bgzf.c:1485-1539), and the
// batch entry points that launch the gfx950 kernels.  There is deliberately NO
// CPU implementation of any codec here: if HIP or the device is unavailable
// every entry point fails with HG_ENODEV.
#include <hip/hip_runtime.h>
#include <chrono>
#include <dlfcn.h>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "htsgpu.h"
#include "hg_internal.h"

#define HG_VERSION_STRING "htsgpu 0.1 (gfx950)"

namespace hg {
int ensure_scratch(hg_ctx *ctx, int slot, size_t bytes) {
    if (ctx->d_scratch_cap[slot] >= bytes) return HG_OK;
    if (ctx->d_scratch[slot]) (void)hipFree(ctx->d_scratch[slot]);
    ctx->d_scratch[slot] = nullptr; ctx->d_scratch_cap[slot] = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    if (hipMalloc(&ctx->d_scratch[slot], cap) != hipSuccess) return HG_ENOMEM;
    ctx->d_scratch_cap[slot] = cap;
    return HG_OK;
}
}  // namespace hg
using hg::ensure_scratch;

// The block layer runs the codec families of a batch on separate HIP streams (up to three per family context) so that their latency-bound kernels
// overlap.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and a hardware queue runs its kernels in order: with ~15
// streams on 4 queues the families took turns (kernel timeline: profiles/r04_cram_slices_256_timeline.txt; a decode call of 256 slices 62 -> 40 ms with
// 16 queues).  Asked for when this library is loaded, unless the user set the variable; without effect (and without harm) when the process initialised
// HIP earlier.
// HTS_GPU_NO_ENV=1 leaves the process environment alone (a host application that manages HIP's settings itself, or that loads this library with dlopen()
// while other threads run -- setenv is not thread-safe); the request can then be made by the deployment: GPU_MAX_HW_QUEUES=16.
__attribute__((constructor)) static void hg_more_hw_queues() {
    const char *no = getenv("HTS_GPU_NO_ENV");
    if (!(no && *no == '1')) (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

extern "C" {

const char *hg_version(void) { return HG_VERSION_STRING; }

const char *hg_strerror(int code) {
    switch (code) {
    case HG_OK: return "ok";
    case HG_EINVAL: return "invalid argument";
    case HG_ENODEV: return "no usable gfx950 device / HIP runtime failure";
    case HG_ENOMEM: return "out of (device) memory";
    case HG_EFORMAT: return "BGZF framing error";
    case HG_ELAUNCH: return "kernel launch or execution failed";
    case HG_EBLOCK: return "one or more blocks failed to decode";
    default: return "unknown error";
    }
}

int hg_init(int device, hg_ctx **out) {
    if (!out) return HG_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return HG_ENODEV;
    if (hipSetDevice(device) != hipSuccess) return HG_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HG_ENODEV;
    hg_ctx *ctx = (hg_ctx *)calloc(1, sizeof(hg_ctx));
    if (!ctx) return HG_ENOMEM;
    ctx->device = device;
    ctx->cus = prop.multiProcessorCount;
    // inflate sizes its own launches (26 one-wave workgroups per CU: bgzf_inflate.hip); this is the figure other callers ask for
    ctx->waves_per_launch = ctx->cus * 24;
    if (hipMalloc((void **)&ctx->d_ticket, HG_TICKETS * sizeof(unsigned int)) != hipSuccess) { free(ctx); return HG_ENOMEM; }
    ctx->launch_seq = new std::atomic<unsigned int>(0);
    ctx->mu = new std::recursive_mutex();
    ctx->tok_mu = new std::mutex();
    // (stream2 .. stream4 are created by the first call that forks onto them: hg_internal.h)
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork4, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join4, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_deflate, hipEventDisableTiming) != hipSuccess) { (void)hipFree(ctx->d_ticket); delete ctx->launch_seq; delete ctx->mu; delete ctx->tok_mu; free(ctx); return HG_ENODEV; }
    *out = ctx;
    return HG_OK;
}

void hg_destroy(hg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < HG_SCRATCH_SLOTS; i++) if (ctx->d_scratch[i]) (void)hipFree(ctx->d_scratch[i]);
    if (ctx->d_tok) (void)hipFree(ctx->d_tok);
    hg::stage_free(ctx);
    for (void *p : ctx->h_slab) free(p);
    for (hg_ctx *c : ctx->sub) if (c) hg_destroy(c);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->stream3) { (void)hipStreamSynchronize(ctx->stream3); (void)hipStreamDestroy(ctx->stream3); }
    if (ctx->stream4) { (void)hipStreamSynchronize(ctx->stream4); (void)hipStreamDestroy(ctx->stream4); }
    if (ctx->ev_fork4) (void)hipEventDestroy(ctx->ev_fork4);
    if (ctx->ev_join4) (void)hipEventDestroy(ctx->ev_join4);
    if (ctx->ev_fork3) (void)hipEventDestroy(ctx->ev_fork3);
    if (ctx->ev_join3) (void)hipEventDestroy(ctx->ev_join3);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->d_ticket) (void)hipFree(ctx->d_ticket);
    delete ctx->launch_seq;
    delete ctx->mu;
    delete ctx->tok_mu;
    if (ctx->ev_deflate) (void)hipEventDestroy(ctx->ev_deflate);
    free(ctx);
}

int hg_device_info(hg_ctx *ctx, int *cus, int *waves) {
    if (!ctx) return HG_EINVAL;
    if (cus) *cus = ctx->cus;
    if (waves) *waves = ctx->waves_per_launch;
    return HG_OK;
}

long hg_bgzf_scan(const uint8_t *buf, size_t len, hg_bgzf_desc *desc, size_t max_desc, uint64_t *total_ulen) {
    size_t pos = 0; long n = 0; uint64_t u = 0;
    while (pos < len) {
        if (len - pos < 18) return HG_EFORMAT;
        const uint8_t *h = buf + pos;
        if (!(h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 &&
              h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0))
            return HG_EFORMAT;
        size_t bs = (size_t)(h[16] | (h[17] << 8)) + 1;
        if (bs < 26 || bs > len - pos) return HG_EFORMAT;
        const uint8_t *t = h + bs - 4;
        uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > HG_BGZF_MAX_BLOCK_SIZE) return HG_EFORMAT;
        if (desc && (size_t)n < max_desc) {
            desc[n].coff = pos; desc[n].uoff = u; desc[n].clen = (uint32_t)bs; desc[n].ulen = isize;
        }
        u += isize; pos += bs; n++;
    }
    if (total_ulen) *total_ulen = u;
    return n;
}

int hg_bgzf_inflate_dev(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc,
                        size_t nblocks, void *d_out, size_t out_cap, int32_t *d_status, void *stream) {
    if (!ctx || (nblocks && (!d_comp || !d_desc || !d_status))) return HG_EINVAL;
    if (((uintptr_t)d_comp & 3u) != 0) return HG_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    return hg::launch_bgzf_inflate(ctx, d_comp, comp_len, d_desc, nblocks, d_out, out_cap, d_status,
                                   (hipStream_t)stream);
}

int hg_ransnx16_decode_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4, size_t n4,
                           const uint32_t *d_sel32, size_t n32, void *d_out, int32_t *d_status, uint32_t *d_scratch,
                           void *stream) {
    if (!ctx || ((n4 || n32) && (!d_in || !d_desc || !d_out || !d_status || !d_scratch))) return HG_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    return hg::launch_ransnx16_decode(ctx, d_in, d_desc, d_sel4, n4, d_sel32, n32, d_out, d_status, d_scratch, (hipStream_t)stream);
}

size_t hg_rans4x8_compress_bound(size_t n) { return n + n / 16 + 257 * 257 * 3 + 9 + 64; }

int hg_rans4x8_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *order, size_t n,
                           uint8_t *const *out, uint32_t *out_len) {
    if (!ctx || (n && (!in || !in_len || !order || !out || !out_len))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hg_stream_desc *desc = (hg_stream_desc *)calloc(n, sizeof(hg_stream_desc));
    uint32_t *ol = (uint32_t *)malloc(n * 4);
    if (!desc || !ol) { free(desc); free(ol); return HG_ENOMEM; }
    uint64_t ioff = 0, ooff = 0, soff = 0, woff = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t cap = hg_rans4x8_compress_bound(in_len[i]);
        desc[i].in_off = ioff; desc[i].in_len = in_len[i]; desc[i].out_off = ooff; desc[i].out_len = (uint32_t)cap;
        desc[i].scratch_off = (uint32_t)soff; desc[i].reserved = (uint32_t)(woff / 16);
        ioff += ((uint64_t)in_len[i] + 15u) & ~15ull;
        ooff += (cap + 15u) & ~15ull;
        soff += hg::rans4x8_enc_scratch_words(order[i] & 1u);
        woff += (2ull * in_len[i] + 64u + 15u) & ~15ull;
        if (soff > 0xffffffffull || woff / 16 > 0xffffffffull) { free(desc); free(ol); return HG_EINVAL; }
    }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, ioff + 64)) || (rc = ensure_scratch(ctx, 1, ooff + 64)) ||
        (rc = ensure_scratch(ctx, 2, n * sizeof(hg_stream_desc))) || (rc = ensure_scratch(ctx, 3, n * 4 + n + 64)) ||
        (rc = ensure_scratch(ctx, 4, woff + 64)) || (rc = ensure_scratch(ctx, 6, soff * 4 + 64))) { free(desc); free(ol); return rc; }
    hipStream_t s = ctx->stream;
    uint32_t *d_ol = (uint32_t *)ctx->d_scratch[3];
    uint8_t *d_ord = (uint8_t *)ctx->d_scratch[3] + n * 4;
    std::vector<uint64_t> ioffs(n), ooffs(n);
    for (size_t i = 0; i < n; i++) { ioffs[i] = desc[i].in_off; ooffs[i] = desc[i].out_off; }
    bool ok = hg::stage_upload(ctx, in, in_len, ioffs.data(), nullptr, n, ioff, (uint8_t *)ctx->d_scratch[0], s) == HG_OK &&
              hipMemsetAsync(d_ol, 0, n * 4, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(ctx->d_scratch[2], desc, n * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(d_ord, order, n, hipMemcpyHostToDevice, s) == hipSuccess;
    rc = ok ? hg::launch_rans4x8_encode(ctx, ctx->d_scratch[0], (const hg_stream_desc *)ctx->d_scratch[2], d_ord, n,
                                        ctx->d_scratch[1], d_ol, ctx->d_scratch[4], (uint32_t *)ctx->d_scratch[6], s) : HG_ELAUNCH;
    if (rc == HG_OK) {
        ok = hipMemcpyAsync(ol, d_ol, n * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        for (size_t i = 0; i < n && ok; i++) out_len[i] = ol[i];
        ok = ok && hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], ooffs.data(), ol, out, n, s) == HG_OK;
        if (!ok) rc = HG_ELAUNCH;
    }
    free(desc); free(ol);
    return rc;
}

size_t hg_ransnx16_compress_bound(size_t n) { return n + n / 16 + 4 * 257 * 257 * 3 + 8192; }

int hg_gzip_inflate_dev(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc, size_t n,
                        void *d_out, size_t out_cap, int32_t *d_status, void *stream) {
    if (!ctx || (n && (!d_comp || !d_desc || !d_status))) return HG_EINVAL;
    if (((uintptr_t)d_comp & 3u) != 0) return HG_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    return hg::launch_bgzf_inflate(ctx, d_comp, comp_len, d_desc, n, d_out, out_cap, d_status, (hipStream_t)stream, 1);
}

// CRAM methods 2 and 3 (bzip2, lzma; cram_io.c:1626-1664) are general-purpose CPU codecs the reference itself only reaches through libbz2 / liblzma: a block of
// either goes to the system's library, looked up at run time (no link-time dependency); absent library = HG_BLOCK_EUNSUPPORTED, the reference's "not compiled
// into this version".  Nothing of the GPU path runs through here.
extern "C++" {
namespace {
struct HostInflaters {
    int (*bz2)(char *, unsigned int *, char *, unsigned int, int, int) = nullptr;                                                    // BZ2_bzBuffToBuffDecompress
    int (*lzma)(uint64_t *, uint32_t, const void *, const uint8_t *, size_t *, size_t, uint8_t *, size_t *, size_t) = nullptr;      // lzma_stream_buffer_decode
};
const HostInflaters &host_inflaters() {
    static HostInflaters H;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *n : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) { H.bz2 = (decltype(H.bz2))dlsym(h, "BZ2_bzBuffToBuffDecompress"); if (H.bz2) break; }
        for (const char *n : {"liblzma.so.5", "liblzma.so"}) if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) { H.lzma = (decltype(H.lzma))dlsym(h, "lzma_stream_buffer_decode"); if (H.lzma) break; }
    });
    return H;
}
int32_t host_inflate(int32_t method, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len) {
    const HostInflaters &H = host_inflaters();
    if (method == HG_CRAM_BZIP2) {
        if (!H.bz2) return HG_BLOCK_EUNSUPPORTED;
        unsigned int got = out_len;
        return H.bz2((char *)out, &got, (char *)in, in_len, 0, 0) == 0 && got == out_len ? 0 : -1;
    }
    if (!H.lzma) return HG_BLOCK_EUNSUPPORTED;
    uint64_t memlimit = 1ull << 31; size_t ip = 0, op = 0;
    return H.lzma(&memlimit, 0, nullptr, in, &ip, in_len, out, &op, out_len) == 0 && op == out_len ? 0 : -1;
}
}  // namespace
}  // extern "C++"

int hg_cram_uncompress_blocks_host(hg_ctx *ctx, size_t n, const int32_t *method, const uint8_t *const *in,
                                   const uint32_t *in_len, uint8_t *const *out, const uint32_t *out_len,
                                   int32_t *status) {
    if (!ctx || (n && (!method || !in || !in_len || !out || !out_len || !status))) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    // partition by method
    size_t ng = 0, nr = 0, nx_ = 0, na = 0, nt = 0, nq = 0;
    for (size_t i = 0; i < n; i++) {
        status[i] = 0;
        if (out_len[i] == 0 || method[i] == HG_CRAM_RAW) {            // cram_io.c:1594-1603: nothing to do
            if (method[i] == HG_CRAM_RAW) { if (in_len[i] != out_len[i]) status[i] = -1; else if (out_len[i]) memcpy(out[i], in[i], out_len[i]); }
        } else if (method[i] == HG_CRAM_GZIP) ng++;
        else if (method[i] == HG_CRAM_RANS4x8) nr++;
        else if (method[i] == HG_CRAM_RANSNx16) nx_++;
        else if (method[i] == HG_CRAM_ARITH) na++;
        else if (method[i] == HG_CRAM_TOK3) nt++;
        else if (method[i] == HG_CRAM_FQZ) nq++;
        else if (method[i] == HG_CRAM_BZIP2 || method[i] == HG_CRAM_LZMA) status[i] = host_inflate(method[i], in[i], in_len[i], out[i], out_len[i]);
        else status[i] = HG_BLOCK_EUNSUPPORTED;
    }
    // Six independent codec families; each runs on its own thread / sibling context / HIP stream when more than one
    // is present, so their kernels (each latency-bound on its slowest block) overlap on the GPU.
    auto run_entropy = [&](hg_ctx *ctx, int pass) -> int {            // 0 Nx16, 1 the range coder, 2 the name tokeniser, 5 fqzcomp
        int rc = HG_OK;
        const int32_t meth = pass == 5 ? HG_CRAM_FQZ : pass == 2 ? HG_CRAM_TOK3 : pass ? HG_CRAM_ARITH : HG_CRAM_RANSNx16;
        const size_t nx = pass == 5 ? nq : pass == 2 ? nt : pass ? na : nx_;
        std::vector<const uint8_t *> xin_(nx); std::vector<uint8_t *> xout_(nx); std::vector<uint32_t> xl_(nx), xo_(nx);
        std::vector<int32_t> xs_(nx); std::vector<size_t> map_(nx);
        const uint8_t **xin = xin_.data(); uint8_t **xout = xout_.data(); uint32_t *xl = xl_.data(), *xo = xo_.data();
        int32_t *xs = xs_.data(); size_t *map = map_.data();
        size_t k = 0;
        for (size_t i = 0; i < n; i++)
            if (out_len[i] && method[i] == meth) { xin[k] = in[i]; xout[k] = out[i]; xl[k] = in_len[i]; xo[k] = out_len[i]; map[k] = i; k++; }
        int r = pass == 5 ? hg_fqz_decode_host(ctx, xin, xl, nx, xout, xo, xs) : pass == 2 ? hg_tok3_decode_host(ctx, xin, xl, nx, xout, xo, xs)
              : pass ? hg_arith_decode_host(ctx, xin, xl, nx, xout, xo, xs) : hg_ransnx16_decode_host(ctx, xin, xl, nx, xout, xo, xs);
        if (r != HG_OK && r != HG_EBLOCK) rc = r;
        for (k = 0; k < nx; k++) status[map[k]] = (r == HG_OK || r == HG_EBLOCK) ? xs[k] : -1;
        return rc;
    };
    auto run_rans4x8 = [&](hg_ctx *ctx) -> int {
        int rc = HG_OK;
        std::vector<const uint8_t *> rin_(nr); std::vector<uint8_t *> rout_(nr); std::vector<uint32_t> rl_(nr), rcap_(nr), ro_(nr);
        std::vector<int32_t> rs_(nr); std::vector<size_t> map_(nr);
        const uint8_t **rin = rin_.data(); uint8_t **rout = rout_.data(); uint32_t *rl = rl_.data(), *rc_ = rcap_.data(), *ro = ro_.data();
        int32_t *rs = rs_.data(); size_t *map = map_.data();
        size_t k = 0;
        for (size_t i = 0; i < n; i++)
            if (out_len[i] && method[i] == HG_CRAM_RANS4x8) { rin[k] = in[i]; rout[k] = out[i]; rl[k] = in_len[i]; rc_[k] = out_len[i]; map[k] = i; k++; }
        int r = hg_rans4x8_decode_host(ctx, rin, rl, nr, rout, rc_, ro, rs);
        if (r != HG_OK && r != HG_EBLOCK) rc = r;
        for (k = 0; k < nr; k++) status[map[k]] = (r == HG_OK || r == HG_EBLOCK) ? ((rs[k] == 0 && ro[k] == out_len[map[k]]) ? 0 : -1) : -1;
        return rc;
    };
    auto run_gzip = [&](hg_ctx *ctx) -> int {
        int rc = HG_OK;
        std::vector<hg_bgzf_desc> desc_(ng); std::vector<size_t> map_(ng); std::vector<int32_t> st_(ng);
        memset(desc_.data(), 0, ng * sizeof(hg_bgzf_desc));
        hg_bgzf_desc *desc = desc_.data(); size_t *map = map_.data(); int32_t *st = st_.data();
        uint64_t ioff = 0, ooff = 0; size_t k = 0;
        for (size_t i = 0; i < n; i++)
            if (out_len[i] && method[i] == HG_CRAM_GZIP) {
                desc[k].coff = ioff; desc[k].clen = in_len[i]; desc[k].uoff = ooff; desc[k].ulen = out_len[i]; map[k] = i; k++;
                ioff += ((uint64_t)in_len[i] + 15u) & ~15ull; ooff += ((uint64_t)out_len[i] + 15u) & ~15ull;
            }
        if ((rc = ensure_scratch(ctx, 0, ioff + 64)) == HG_OK && (rc = ensure_scratch(ctx, 1, ooff + 64)) == HG_OK &&
            (rc = ensure_scratch(ctx, 2, ng * sizeof(hg_bgzf_desc))) == HG_OK && (rc = ensure_scratch(ctx, 3, ng * 4)) == HG_OK) {
            hipStream_t s = ctx->stream;
            bool ok = true;
            {
                std::vector<const uint8_t *> gp(ng); std::vector<uint32_t> gl(ng); std::vector<uint64_t> go(ng);
                for (k = 0; k < ng; k++) { gp[k] = in[map[k]]; gl[k] = desc[k].clen; go[k] = desc[k].coff; }
                ok = hg::stage_upload(ctx, gp.data(), gl.data(), go.data(), nullptr, ng, ioff, (uint8_t *)ctx->d_scratch[0], s) == HG_OK;
            }
            ok = ok && hipMemcpyAsync(ctx->d_scratch[2], desc, ng * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice, s) == hipSuccess;
            rc = ok ? hg::launch_bgzf_inflate(ctx, ctx->d_scratch[0], (size_t)ioff, (const hg_bgzf_desc *)ctx->d_scratch[2], ng,
                                              ctx->d_scratch[1], (size_t)ooff, (int32_t *)ctx->d_scratch[3], s, 1) : HG_ELAUNCH;
            if (rc == HG_OK) {
                ok = hipMemcpyAsync(st, ctx->d_scratch[3], ng * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
                std::vector<uint64_t> go(ng); std::vector<uint32_t> gl(ng); std::vector<uint8_t *> gd(ng);
                for (k = 0; k < ng && ok; k++) {
                    status[map[k]] = st[k];
                    go[k] = desc[k].uoff; gl[k] = st[k] == 0 ? desc[k].ulen : 0u; gd[k] = out[map[k]];
                }
                ok = ok && hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], go.data(), gl.data(), gd.data(), ng, s) == HG_OK;
                if (!ok) rc = HG_ELAUNCH;
            }
        }
        return rc;
    };
    struct Task { int kind; size_t cnt; } tasks[6] = {{5, nq}, {0, nx_}, {1, na}, {2, nt}, {3, nr}, {4, ng}};
    int nact = 0;
    for (auto &t : tasks) nact += t.cnt != 0;
    int trc[6] = {HG_OK, HG_OK, HG_OK, HG_OK, HG_OK, HG_OK};
    static const bool stats = getenv("HTS_GPU_STATS") != nullptr;
    auto run_one = [&](int kind, hg_ctx *c) {
        const auto t0 = std::chrono::steady_clock::now();
        trc[kind] = (kind <= 2 || kind == 5) ? run_entropy(c, kind) : kind == 3 ? run_rans4x8(c) : run_gzip(c);
        if (stats) {
            static const char *nm[6] = {"nx16", "arith", "tok3", "rans4x8", "gzip", "fqz"};
            fprintf(stderr, "[htsgpu stats] uncompress family %s: %.1f ms\n", nm[kind], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    };
    if (nact <= 1) { for (auto &t : tasks) if (t.cnt) run_one(t.kind, ctx); }
    else {
        std::vector<std::thread> th;
        for (auto &t : tasks) {
            if (!t.cnt) continue;
            if (!ctx->sub[t.kind] && hg_init(ctx->device, &ctx->sub[t.kind]) != HG_OK) { trc[t.kind] = HG_ENOMEM; continue; }
            hg_ctx *c = ctx->sub[t.kind]; const int kind = t.kind;
            th.emplace_back([&, kind, c]() { if (hipSetDevice(ctx->device) != hipSuccess) { trc[kind] = HG_ENODEV; return; } run_one(kind, c); });
        }
        for (auto &t : th) t.join();
    }
    int rc = HG_OK;
    for (int k = 0; k < 6; k++) if (trc[k] != HG_OK && rc == HG_OK) rc = trc[k];
    if (rc != HG_OK) return rc;
    for (size_t i = 0; i < n; i++) if (status[i] != 0) return HG_EBLOCK;
    return HG_OK;
}

static uint32_t crc_combine_h(uint32_t crc1, uint32_t crc2, uint64_t len2);
// cram_uncompress_block's first step (cram_io.c:1585-1592): crc32(b->crc_part, payload) must equal the stored CRC.
int hg_cram_uncompress_blocks_crc_host(hg_ctx *ctx, size_t n, const int32_t *method, const uint8_t *const *in, const uint32_t *in_len,
                                       const uint32_t *crc_part, const uint32_t *crc32, uint8_t *const *out, const uint32_t *out_len,
                                       int32_t *status) {
    if (!ctx || (n && (!method || !in || !in_len || !crc_part || !crc32 || !out || !out_len || !status))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    // payload CRCs on the device, one wavefront per block
    std::vector<uint64_t> off(n); uint64_t tot = 0;
    for (size_t i = 0; i < n; i++) { off[i] = tot; tot += ((uint64_t)in_len[i] + 15u) & ~15ull; }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, tot + 64)) || (rc = ensure_scratch(ctx, 2, n * 8 + 64)) || (rc = ensure_scratch(ctx, 3, n * 8 + 64))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = hg::stage_upload(ctx, in, in_len, off.data(), nullptr, n, tot, (uint8_t *)ctx->d_scratch[0], s))) return rc;
    uint32_t *d_len = (uint32_t *)ctx->d_scratch[3], *d_crc = d_len + n;
    std::vector<uint32_t> crc(n);
    if (hipMemcpyAsync(ctx->d_scratch[2], off.data(), n * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_len, in_len, n * 4, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    if ((rc = hg::launch_crc32(ctx, ctx->d_scratch[0], (const uint64_t *)ctx->d_scratch[2], d_len, n, d_crc, s))) return rc;
    if (hipMemcpyAsync(crc.data(), d_crc, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    // blocks that fail the check are not decoded (the reference returns -1 before looking at the method)
    std::vector<int32_t> m2(method, method + n);
    std::vector<uint32_t> ol2(out_len, out_len + n);
    std::vector<char> bad(n, 0);
    for (size_t i = 0; i < n; i++) {
        const uint32_t c = in_len[i] ? crc_combine_h(crc_part[i], crc[i], in_len[i]) : crc_part[i];
        if (c != crc32[i]) { bad[i] = 1; m2[i] = HG_CRAM_RAW; ol2[i] = 0; }     // parked as an empty RAW block for the dispatcher
    }
    std::vector<uint32_t> il2(in_len, in_len + n);
    for (size_t i = 0; i < n; i++) if (bad[i]) il2[i] = 0;
    rc = hg_cram_uncompress_blocks_host(ctx, n, m2.data(), in, il2.data(), out, ol2.data(), status);
    if (rc != HG_OK && rc != HG_EBLOCK) return rc;
    for (size_t i = 0; i < n; i++) if (bad[i]) { status[i] = -1; rc = HG_EBLOCK; }
    return rc;
}

int hg_bgzf_inflate_host(hg_ctx *ctx, const uint8_t *comp, size_t comp_len, uint8_t *out, size_t out_cap,
                         size_t *out_len, int32_t *status, size_t max_status, long *first_bad_idx,
                         int *first_bad_code) {
    if (!ctx || (!comp && comp_len)) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    uint64_t total = 0;
    long n = hg_bgzf_scan(comp, comp_len, nullptr, 0, &total);
    if (n < 0) return (int)n;
    if (out_len) *out_len = (size_t)total;
    if (total > out_cap) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg_bgzf_desc *desc = (hg_bgzf_desc *)malloc((size_t)n * sizeof(hg_bgzf_desc));
    int32_t *st = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    if (!desc || !st) { free(desc); free(st); return HG_ENOMEM; }
    hg_bgzf_scan(comp, comp_len, desc, (size_t)n, nullptr);
    int rc;
    size_t comp_pad = (comp_len + 3) & ~(size_t)3;
    if ((rc = ensure_scratch(ctx, 0, comp_pad)) || (rc = ensure_scratch(ctx, 1, (size_t)total + 4)) ||
        (rc = ensure_scratch(ctx, 2, (size_t)n * sizeof(hg_bgzf_desc))) ||
        (rc = ensure_scratch(ctx, 3, (size_t)n * sizeof(int32_t)))) {
        free(desc); free(st); return rc;
    }
    hipStream_t s = ctx->stream;
    bool ok = hipMemcpyAsync(ctx->d_scratch[0], comp, comp_len, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(ctx->d_scratch[2], desc, (size_t)n * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice, s) == hipSuccess;
    rc = ok ? hg::launch_bgzf_inflate(ctx, ctx->d_scratch[0], comp_len, (const hg_bgzf_desc *)ctx->d_scratch[2],
                                      (size_t)n, ctx->d_scratch[1], (size_t)total, (int32_t *)ctx->d_scratch[3], s)
            : HG_ELAUNCH;
    if (rc == HG_OK) {
        ok = hipMemcpyAsync(st, ctx->d_scratch[3], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s) == hipSuccess &&
             (total == 0 || hipMemcpyAsync(out, ctx->d_scratch[1], (size_t)total, hipMemcpyDeviceToHost, s) == hipSuccess) &&
             hipStreamSynchronize(s) == hipSuccess;
        if (!ok) rc = HG_ELAUNCH;
    }
    if (rc == HG_OK) {
        for (long i = 0; i < n; i++) {
            if (status && (size_t)i < max_status) status[i] = st[i];
            if (st[i] != HG_BLOCK_OK && rc == HG_OK) {
                rc = HG_EBLOCK;
                if (first_bad_idx) *first_bad_idx = i;
                if (first_bad_code) *first_bad_code = st[i];
            }
        }
    }
    free(desc); free(st);
    return rc;
}

int hg_bgzf_deflate_dev(hg_ctx *ctx, const void *d_plain, const hg_bgzf_desc *d_desc, size_t nblocks, int level,
                        void *d_slots, uint32_t *d_clen, void *stream) {
    if (!ctx || (nblocks && (!d_plain || !d_desc || !d_slots || !d_clen)) || level < 0 || level > 9) return HG_EINVAL;
    if (((uintptr_t)d_slots & 3u) != 0) return HG_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    return hg::launch_bgzf_deflate(ctx, d_plain, d_desc, nblocks, level, d_slots, d_clen, (hipStream_t)stream);
}

int hg_bgzf_pack_dev(hg_ctx *ctx, const void *d_slots, const hg_bgzf_desc *d_desc, const uint32_t *d_clen,
                     size_t nblocks, void *d_packed, size_t packed_cap, uint64_t *d_packed_off, uint64_t *d_total,
                     int add_eof, void *stream) {
    if (!ctx || !d_packed || !d_packed_off || !d_total || (nblocks && (!d_slots || !d_desc || !d_clen))) return HG_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    return hg::launch_bgzf_pack(ctx, d_slots, d_desc, d_clen, nblocks, d_packed, packed_cap, d_packed_off, d_total,
                                add_eof, (hipStream_t)stream);
}

int hg_bgzf_deflate_host(hg_ctx *ctx, const uint8_t *plain, size_t len, const uint64_t *cuts, size_t ncuts, int level,
                         int add_eof, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!ctx || (!plain && len) || !out || level < 0 || level > 9) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    size_t nb = cuts ? ncuts : (len + HG_BGZF_BLOCK_SIZE - 1) / HG_BGZF_BLOCK_SIZE;
    hg_bgzf_desc *desc = (hg_bgzf_desc *)malloc((nb ? nb : 1) * sizeof(hg_bgzf_desc));
    if (!desc) return HG_ENOMEM;
    for (size_t i = 0; i < nb; i++) {
        uint64_t a = cuts ? cuts[i] : (uint64_t)i * HG_BGZF_BLOCK_SIZE;
        uint64_t b = cuts ? cuts[i + 1] : (a + HG_BGZF_BLOCK_SIZE < len ? a + HG_BGZF_BLOCK_SIZE : len);
        if (b < a || b > len || b - a > HG_BGZF_BLOCK_SIZE) { free(desc); return HG_EINVAL; }
        desc[i].uoff = a; desc[i].ulen = (uint32_t)(b - a); desc[i].coff = (uint64_t)i * HG_BGZF_MAX_BLOCK_SIZE; desc[i].clen = 0;
    }
    int rc;
    size_t slots = nb * (size_t)HG_BGZF_MAX_BLOCK_SIZE;
    if ((rc = ensure_scratch(ctx, 0, len + 64)) || (rc = ensure_scratch(ctx, 1, slots + 64)) ||
        (rc = ensure_scratch(ctx, 2, (nb + 1) * sizeof(hg_bgzf_desc))) || (rc = ensure_scratch(ctx, 3, (nb + 1) * 4)) ||
        (rc = ensure_scratch(ctx, 4, (nb + 2) * 8)) || (rc = ensure_scratch(ctx, 5, slots + 64))) {
        free(desc); return rc;
    }
    hipStream_t s = ctx->stream;
    uint64_t *d_poff = (uint64_t *)ctx->d_scratch[4];
    uint64_t *d_total = d_poff + nb;
    bool ok = (len == 0 || hipMemcpyAsync(ctx->d_scratch[0], plain, len, hipMemcpyHostToDevice, s) == hipSuccess) &&
              (nb == 0 || hipMemcpyAsync(ctx->d_scratch[2], desc, nb * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice, s) == hipSuccess);
    rc = ok ? hg::launch_bgzf_deflate(ctx, ctx->d_scratch[0], (const hg_bgzf_desc *)ctx->d_scratch[2], nb, level,
                                      ctx->d_scratch[1], (uint32_t *)ctx->d_scratch[3], s) : HG_ELAUNCH;
    if (rc == HG_OK)
        rc = hg::launch_bgzf_pack(ctx, ctx->d_scratch[1], (const hg_bgzf_desc *)ctx->d_scratch[2],
                                  (const uint32_t *)ctx->d_scratch[3], nb, ctx->d_scratch[5], slots + 64, d_poff, d_total,
                                  add_eof, s);
    uint64_t total = 0;
    if (rc == HG_OK) {
        ok = hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        if (!ok) rc = HG_ELAUNCH;
    }
    if (rc == HG_OK) {
        if (out_len) *out_len = (size_t)total;
        if (total > out_cap) rc = HG_EINVAL;
        else if (total && (hipMemcpy(out, ctx->d_scratch[5], (size_t)total, hipMemcpyDeviceToHost) != hipSuccess)) rc = HG_ELAUNCH;
    }
    free(desc);
    return rc;
}

int hg_rans4x8_decode_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, void *d_out,
                          int32_t *d_status, uint32_t *d_scratch, void *stream) {
    if (!ctx || (n && (!d_in || !d_desc || !d_out || !d_status || !d_scratch))) return HG_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    return hg::launch_rans4x8_decode(ctx, d_in, d_desc, n, d_out, d_status, d_scratch, (hipStream_t)stream);
}

int hg_rans4x8_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                           uint8_t *const *out, const uint32_t *out_cap, uint32_t *out_len, int32_t *status) {
    if (!ctx || (n && (!in || !in_len || !out || !out_cap || !out_len))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hg_stream_desc *desc = (hg_stream_desc *)calloc(n, sizeof(hg_stream_desc));
    int32_t *st = (int32_t *)malloc(n * sizeof(int32_t));
    if (!desc || !st) { free(desc); free(st); return HG_ENOMEM; }
    uint64_t ioff = 0, ooff = 0, soff = 0;
    // A stream whose header promises more bytes than the caller's buffer holds is that STREAM's failure (the size is
    // data, not an argument): it is parked as an empty descriptor, not uploaded, and reported -1; the rest of the batch
    // is decoded (cram_uncompress_block fails only the offending block, cram_io.c:1671-1674).
    std::vector<int32_t> skip(n, 0);
    std::vector<uint32_t> ilen(in_len, in_len + n);
    for (size_t i = 0; i < n; i++) {
        uint32_t usz = 0;
        if (in_len[i] >= 9) usz = (uint32_t)in[i][5] | ((uint32_t)in[i][6] << 8) | ((uint32_t)in[i][7] << 16) | ((uint32_t)in[i][8] << 24);
        if (usz > out_cap[i]) { skip[i] = 1; ilen[i] = 0; usz = 0; }
        desc[i].in_off = ioff; desc[i].in_len = ilen[i]; desc[i].out_off = ooff; desc[i].out_len = usz;
        desc[i].scratch_off = (uint32_t)soff;
        out_len[i] = usz;
        ioff += ((uint64_t)ilen[i] + 15u) & ~15ull;
        ooff += ((uint64_t)usz + 15u) & ~15ull;
        soff += HG_RANS4X8_SCRATCH_WORDS(ilen[i]);
        if (soff > 0xffffffffull) { free(desc); free(st); return HG_EINVAL; }
    }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, ioff + 64)) || (rc = ensure_scratch(ctx, 1, ooff + 64)) ||
        (rc = ensure_scratch(ctx, 2, n * sizeof(hg_stream_desc))) || (rc = ensure_scratch(ctx, 3, n * 4)) ||
        (rc = ensure_scratch(ctx, 6, soff * 4 + 64))) { free(desc); free(st); return rc; }
    hipStream_t s = ctx->stream;
    std::vector<uint64_t> ioffs(n), ooffs(n);
    for (size_t i = 0; i < n; i++) { ioffs[i] = desc[i].in_off; ooffs[i] = desc[i].out_off; }
    bool ok = hg::stage_upload(ctx, in, ilen.data(), ioffs.data(), skip.data(), n, ioff, (uint8_t *)ctx->d_scratch[0], s) == HG_OK;
    ok = ok && hipMemcpyAsync(ctx->d_scratch[2], desc, n * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) == hipSuccess;
    rc = ok ? hg::launch_rans4x8_decode(ctx, ctx->d_scratch[0], (const hg_stream_desc *)ctx->d_scratch[2], n, ctx->d_scratch[1],
                                        (int32_t *)ctx->d_scratch[3], (uint32_t *)ctx->d_scratch[6], s) : HG_ELAUNCH;
    if (rc == HG_OK) {
        ok = hipMemcpyAsync(st, ctx->d_scratch[3], n * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        ok = ok && hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], ooffs.data(), out_len, out, n, s) == HG_OK;
        if (!ok) rc = HG_ELAUNCH;
    }
    if (rc == HG_OK) for (size_t i = 0; i < n; i++) if (skip[i]) st[i] = -1;
    if (rc == HG_OK)
        for (size_t i = 0; i < n; i++) { if (status) status[i] = st[i]; if (st[i] != 0) rc = HG_EBLOCK; }
    free(desc); free(st);
    return rc;
}

// ---- host-side CRC-32 concatenation: crc(A||B) from crc(A), crc(B), |B| (GF(2) polynomial arithmetic) ----
static uint32_t crc_mulmod_h(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
static uint32_t crc_combine_h(uint32_t crc1, uint32_t crc2, uint64_t len2) {
    uint32_t xp = 0x00800000u, acc = 0x80000000u;                    // x^8, x^0
    for (; len2; len2 >>= 1) { if (len2 & 1) acc = crc_mulmod_h(acc, xp); xp = crc_mulmod_h(xp, xp); }
    return crc_mulmod_h(acc, crc1) ^ crc2;
}

size_t hg_gzip_compress_bound(size_t n) { return n + (n / HG_BGZF_BLOCK_SIZE + 2) * 16 + 64; }

int hg_gzip_deflate_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n, int level,
                         uint8_t *const *out, uint32_t *out_len) {
    if (!ctx || (n && (!in || !in_len || !out || !out_len)) || level < 0 || level > 9) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    size_t nchunks = 0; uint64_t total_in = 0;
    for (size_t i = 0; i < n; i++) { nchunks += in_len[i] ? (in_len[i] + HG_BGZF_BLOCK_SIZE - 1) / HG_BGZF_BLOCK_SIZE : 1; total_in += ((uint64_t)in_len[i] + 15u) & ~15ull; }
    hg_bgzf_desc *desc = (hg_bgzf_desc *)calloc(nchunks, sizeof(hg_bgzf_desc));
    uint32_t *clen = (uint32_t *)malloc(nchunks * 4), *crc = (uint32_t *)malloc(nchunks * 4);
    if (!desc || !clen || !crc) { free(desc); free(clen); free(crc); return HG_ENOMEM; }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, total_in + 64)) || (rc = ensure_scratch(ctx, 1, nchunks * (size_t)HG_BGZF_MAX_BLOCK_SIZE + 64)) ||
        (rc = ensure_scratch(ctx, 2, nchunks * sizeof(hg_bgzf_desc))) || (rc = ensure_scratch(ctx, 3, nchunks * 8 + 64))) {
        free(desc); free(clen); free(crc); return rc;
    }
    hipStream_t s = ctx->stream;
    bool ok = true;
    uint64_t ioff = 0; size_t k = 0;
    std::vector<uint64_t> ioffs(n);
    for (size_t i = 0; i < n && ok; i++) {
        ioffs[i] = ioff;
        const size_t nc = in_len[i] ? (in_len[i] + HG_BGZF_BLOCK_SIZE - 1) / HG_BGZF_BLOCK_SIZE : 1;
        for (size_t c = 0; c < nc; c++, k++) {
            const uint64_t a = (uint64_t)c * HG_BGZF_BLOCK_SIZE;
            desc[k].uoff = ioff + a;
            desc[k].ulen = (uint32_t)(in_len[i] - a < HG_BGZF_BLOCK_SIZE ? in_len[i] - a : HG_BGZF_BLOCK_SIZE);
            desc[k].coff = (uint64_t)k * HG_BGZF_MAX_BLOCK_SIZE;
            desc[k].clen = c + 1 == nc ? 1u : 0u;                     // last chunk of its member
        }
        ioff += ((uint64_t)in_len[i] + 15u) & ~15ull;
    }
    uint32_t *d_clen = (uint32_t *)ctx->d_scratch[3], *d_crc = d_clen + nchunks;
    ok = ok && hg::stage_upload(ctx, in, in_len, ioffs.data(), nullptr, n, ioff, (uint8_t *)ctx->d_scratch[0], s) == HG_OK;
    ok = ok && hipMemcpyAsync(ctx->d_scratch[2], desc, nchunks * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice, s) == hipSuccess;
    rc = ok ? hg::launch_bgzf_deflate(ctx, ctx->d_scratch[0], (const hg_bgzf_desc *)ctx->d_scratch[2], nchunks, level,
                                      ctx->d_scratch[1], d_clen, s, 1, d_crc) : HG_ELAUNCH;
    if (rc == HG_OK) {
        ok = hipMemcpyAsync(clen, d_clen, nchunks * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(crc, d_crc, nchunks * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        k = 0;
        std::vector<uint64_t> po(nchunks); std::vector<uint8_t *> pd(nchunks);
        for (size_t i = 0; i < n && ok; i++) {
            const size_t nc = in_len[i] ? (in_len[i] + HG_BGZF_BLOCK_SIZE - 1) / HG_BGZF_BLOCK_SIZE : 1;
            uint8_t *o = out[i];
            const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
            memcpy(o, hdr, 10);
            size_t pos = 10; uint32_t c32 = 0;
            for (size_t c = 0; c < nc && ok; c++, k++) {
                po[k] = desc[k].coff; pd[k] = o + pos;                // chunk payloads are fetched in one batch below
                pos += clen[k];
                c32 = c == 0 ? crc[k] : crc_combine_h(c32, crc[k], desc[k].ulen);
            }
            for (int b = 0; b < 4; b++) { o[pos + b] = (uint8_t)(c32 >> (8 * b)); o[pos + 4 + b] = (uint8_t)(in_len[i] >> (8 * b)); }
            out_len[i] = (uint32_t)(pos + 8);
        }
        ok = ok && hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], po.data(), clen, pd.data(), nchunks, s) == HG_OK;
        if (!ok) rc = HG_ELAUNCH;
    }
    free(desc); free(clen); free(crc);
    return rc;
}

size_t hg_cram_compress_bound(size_t n) {
    size_t a = hg_gzip_compress_bound(n), b = hg_rans4x8_compress_bound(n), c = hg_ransnx16_compress_bound(n), d = hg_arith_compress_bound(n);
    const size_t e = hg_tok3_compress_bound(n);
    if (e > a) a = e;
    if (b > a) a = b;
    if (c > a) a = c;
    return d > a ? d : a;
}

int hg_cram_compress_blocks_host(hg_ctx *ctx, size_t n, const uint32_t *method_mask, int level, const uint8_t *const *in,
                                 const uint32_t *in_len, uint8_t *const *out, uint32_t *out_len, int32_t *method_used) {
    if (!ctx || (n && (!method_mask || !in || !in_len || !out || !out_len || !method_used))) return HG_EINVAL;
    // start from RAW (sz_best = uncomp_size, cram_io.c:2001)
    for (size_t i = 0; i < n; i++) { out_len[i] = in_len[i]; method_used[i] = HG_CRAM_RAW; if (in_len[i]) memcpy(out[i], in[i], in_len[i]); }
    if (level == 0) return HG_OK;                                      // cram_io.c:1967-1972
    std::vector<uint8_t *> tmp(n, nullptr);
    std::vector<const uint8_t *> sin; std::vector<uint8_t *> sout; std::vector<uint32_t> slen, solen; std::vector<size_t> map;
    std::vector<uint8_t> par;
    auto keep = [&](size_t k) {                                        // candidate k beat the incumbent?
        const size_t i = map[k];
        if (solen[k] && solen[k] < out_len[i]) { memcpy(out[i], sout[k], solen[k]); out_len[i] = solen[k]; return true; }
        return false;
    };
    auto gather = [&](uint32_t bit) {
        sin.clear(); sout.clear(); slen.clear(); map.clear();
        for (size_t i = 0; i < n; i++) if (in_len[i] && (method_mask[i] >> bit) & 1u) {
            if (!tmp[i]) tmp[i] = (uint8_t *)malloc(hg_cram_compress_bound(in_len[i]));
            sin.push_back(in[i]); sout.push_back(tmp[i]); slen.push_back(in_len[i]); map.push_back(i);
        }
        solen.assign(sin.size(), 0);
    };
    int rc = HG_OK;
    gather(HG_CRAM_GZIP);
    if (!sin.empty()) {
        rc = hg_gzip_deflate_host(ctx, sin.data(), slen.data(), sin.size(), level, sout.data(), solen.data());
        for (size_t k = 0; rc == HG_OK && k < sin.size(); k++) if (keep(k)) method_used[map[k]] = HG_CRAM_GZIP;
    }
    for (int order = 0; order < 2 && rc == HG_OK; order++) {
        gather(HG_CRAM_RANS4x8);
        if (sin.empty()) break;
        par.assign(sin.size(), (uint8_t)order);
        rc = hg_rans4x8_encode_host(ctx, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
        for (size_t k = 0; rc == HG_OK && k < sin.size(); k++) if (keep(k)) method_used[map[k]] = HG_CRAM_RANS4x8;
    }
    // RANS_PR0, PR1 always; PR64, PR9, PR128, PR193 above level 1; PR129, PR192 above level 5 (cram_encode.c:818-826)
    static const uint8_t nx16_sets[] = {0, 1, 64, 9, 128, 193, 129, 192};
    const int nsets = level > 5 ? 8 : level > 1 ? 6 : 2;
    for (int v = 0; v < nsets && rc == HG_OK; v++) {
        gather(HG_CRAM_RANSNx16);
        if (sin.empty()) break;
        par.resize(sin.size());
        for (size_t k = 0; k < sin.size(); k++) par[k] = (uint8_t)(nx16_sets[v] | (slen[k] >= 65536u ? 4 : 0));   // 32-way for big inputs
        rc = hg_ransnx16_encode_host(ctx, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
        for (size_t k = 0; rc == HG_OK && k < sin.size(); k++) if (keep(k)) method_used[map[k]] = HG_CRAM_RANSNx16;
    }
    // arith: same flag sets as Nx16 (cram_io.c:1877), no 32-way
    for (int v = 0; v < nsets && rc == HG_OK; v++) {
        gather(HG_CRAM_ARITH);
        if (sin.empty()) break;
        par.assign(sin.size(), nx16_sets[v]);
        rc = hg_arith_encode_host(ctx, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
        for (size_t k = 0; rc == HG_OK && k < sin.size(); k++) if (keep(k)) method_used[map[k]] = HG_CRAM_ARITH;
    }
    // name tokeniser: rANS back-end (TOK3) and range-coder back-end (TOKA); an out_len of 0 means "not a list of names"
    for (int v = 0; v < 2 && rc == HG_OK; v++) {
        gather(v ? 9u : (uint32_t)HG_CRAM_TOK3);
        if (sin.empty()) continue;
        par.assign(sin.size(), (uint8_t)v);
        rc = hg_tok3_encode_host(ctx, sin.data(), slen.data(), par.data(), sin.size(), sout.data(), solen.data());
        for (size_t k = 0; rc == HG_OK && k < sin.size(); k++) if (keep(k)) method_used[map[k]] = HG_CRAM_TOK3;
    }
    for (auto p : tmp) free(p);
    return rc;
}

// crc[i] = CRC-32 of host buffer i: one staged upload, one wavefront per buffer, one download
int hg_crc32_batch_host(hg_ctx *ctx, const uint8_t *const *buf, const uint32_t *len, size_t n, uint32_t *crc) {
    if (!ctx || (n && (!buf || !len || !crc))) return HG_EINVAL;
    if (n == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    std::vector<uint64_t> off(n); uint64_t tot = 0;
    for (size_t i = 0; i < n; i++) { off[i] = tot; tot += ((uint64_t)len[i] + 15u) & ~15ull; }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, tot + 64)) || (rc = ensure_scratch(ctx, 2, n * 8 + 64)) || (rc = ensure_scratch(ctx, 3, n * 8 + 64))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = hg::stage_upload(ctx, buf, len, off.data(), nullptr, n, tot, (uint8_t *)ctx->d_scratch[0], s))) return rc;
    uint32_t *d_len = (uint32_t *)ctx->d_scratch[3], *d_crc = d_len + n;
    if (hipMemcpyAsync(ctx->d_scratch[2], off.data(), n * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_len, len, n * 4, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    if ((rc = hg::launch_crc32(ctx, ctx->d_scratch[0], (const uint64_t *)ctx->d_scratch[2], d_len, n, d_crc, s))) return rc;
    if (hipMemcpyAsync(crc, d_crc, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    return HG_OK;
}

// CRC-32 of one host buffer (hts_crc32, bgzf.c:557-559): upload, one wavefront per 1 MiB piece, combine on the host
int hg_crc32_host(hg_ctx *ctx, const void *buf, size_t len, uint32_t *crc) {
    if (!ctx || !crc || (!buf && len)) return HG_EINVAL;
    *crc = 0;
    if (len == 0) return HG_OK;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    const size_t piece = 1u << 20, n = (len + piece - 1) / piece;
    int rc;
    if ((rc = ensure_scratch(ctx, 0, len + 64)) || (rc = ensure_scratch(ctx, 2, n * 8 + 64)) || (rc = ensure_scratch(ctx, 3, n * 8 + 64))) return rc;
    std::vector<uint64_t> off(n); std::vector<uint32_t> ln(n), c(n);
    for (size_t i = 0; i < n; i++) { off[i] = i * piece; ln[i] = (uint32_t)(len - off[i] < piece ? len - off[i] : piece); }
    hipStream_t s = ctx->stream;
    uint32_t *d_len = (uint32_t *)ctx->d_scratch[3], *d_crc = d_len + n;
    if (hipMemcpyAsync(ctx->d_scratch[0], buf, len, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(ctx->d_scratch[2], off.data(), n * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_len, ln.data(), n * 4, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    if ((rc = hg::launch_crc32(ctx, ctx->d_scratch[0], (const uint64_t *)ctx->d_scratch[2], d_len, n, d_crc, s))) return rc;
    if (hipMemcpyAsync(c.data(), d_crc, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    uint32_t acc = c[0];
    for (size_t i = 1; i < n; i++) acc = crc_combine_h(acc, c[i], ln[i]);
    *crc = acc;
    return HG_OK;
}

int hg_crc32_dev(hg_ctx *ctx, const void *d_data, const uint64_t *d_off, const uint32_t *d_len, size_t n,
                 uint32_t *d_crc, void *stream) {
    if (!ctx || (n && (!d_data || !d_off || !d_len || !d_crc))) return HG_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    return hg::launch_crc32(ctx, d_data, d_off, d_len, n, d_crc, (hipStream_t)stream);
}

}  // extern "C"

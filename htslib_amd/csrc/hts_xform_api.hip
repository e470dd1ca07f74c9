// hts_xform_api.hip -- the byte transforms of CRAM 4.0's E_XPACK / E_XRLE encodings as callable entry points
// (SURVEY §8 a19).
//
// Reference boundary: the htscodecs functions cram/cram_codecs.c calls --
//     hts_unpack      cram_codecs.c:1399      hts_pack        cram_codecs.c:1520
//     hts_rle_decode  cram_codecs.c:2106      hts_rle_encode  cram_codecs.c:2278
// (prototypes: htscodecs/pack.h, htscodecs/rle.h of the un-vendored submodule; PARITY UNPINNED like rANS Nx16 --
// the semantics restated in oracle/hts_xform_oracle.c are the published ones).  The device work is done by the SAME
// two kernels that serve the PACK / RLE flags of rANS Nx16 streams (ransnx16_xform.hip, ransnx16_xenc.hip); this
// file only builds their job records around one host buffer.  The reference-named wrappers (hts_pack, ...) live in
// cram_block_front.cpp.
//
// One call = one buffer = one wavefront: these transforms are a few passes over a data series; a caller with
// many series should batch them (hg_hts_xform_batch below is the array form the single calls use).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "htsgpu.h"
#include "hg_internal.h"

using hg::ensure_scratch;

namespace {

constexpr uint64_t MAX_LEN = 0x7fffffffull;        // the kernels index with 32 bits

inline uint64_t al(uint64_t v) { return (v + 63u) & ~63ull; }

// run one encoder-side transform job over `data`; on success *res is its result record and the device buffer
// (scratch slot 0) holds the produced bytes at the offsets recorded in res / the job.
int run_xenc(hg_ctx *ctx, const uint8_t *data, uint64_t len, uint32_t flags, const uint8_t *preset, uint32_t npreset,
             hg::nx16_xenc *job, hg::nx16_xenc_res *res) {
    hg::nx16_xenc J;
    memset(&J, 0, sizeof J);
    J.src_off = 0; J.g_off = 0; J.p_off = al(len); J.l_off = J.p_off + al(len); J.m_off = J.l_off + al(len);
    J.n = (uint32_t)len; J.stride = 1; J.flags = flags;
    const uint64_t total = J.m_off + al(len + 257 + 8) + sizeof(hg::nx16_xenc) + sizeof(hg::nx16_xenc_res) + 128;
    int rc;
    if ((rc = ensure_scratch(ctx, 0, total)) != HG_OK) return rc;
    uint8_t *d = (uint8_t *)ctx->d_scratch[0];
    const uint64_t job_off = J.m_off + al(len + 257 + 8), res_off = job_off + al(sizeof J);
    hipStream_t s = ctx->stream;
    bool ok = hipMemcpyAsync(d, data, len, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d + job_off, &J, sizeof J, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok && npreset) {
        uint8_t hdr[257];
        hdr[0] = (uint8_t)npreset; memcpy(hdr + 1, preset, npreset);
        ok = hipMemcpyAsync(d + J.m_off, hdr, 1 + npreset, hipMemcpyHostToDevice, s) == hipSuccess;
        if (ok) ok = hipStreamSynchronize(s) == hipSuccess;              // hdr is a stack buffer
    }
    if (!ok) return HG_ELAUNCH;
    if ((rc = hg::launch_ransnx16_xenc(ctx, d, (const hg::nx16_xenc *)(d + job_off), 1, (hg::nx16_xenc_res *)(d + res_off), s)) != HG_OK) return rc;
    if (hipMemcpyAsync(res, d + res_off, sizeof *res, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return HG_ELAUNCH;
    *job = J;
    return HG_OK;
}

inline bool fetch(hg_ctx *ctx, void *dst, uint64_t off, uint64_t n) {
    if (!n) return true;
    return hipMemcpy(dst, (const uint8_t *)ctx->d_scratch[0] + off, n, hipMemcpyDeviceToHost) == hipSuccess;
}

}  // namespace

extern "C" {

// hts_pack (htscodecs/pack.h): <= 16 distinct byte values -> 1, 2 or 4 bits each, first value in the low bits.
// out_meta = [number of symbols][the symbols in ascending order] (out_meta_len bytes, <= 17); returns a malloc'd
// buffer of *out_len packed bytes.  More than 16 distinct values: the data is returned unpacked (copy) with
// out_meta = [count & 0xff], out_meta_len 1.  One distinct value: *out_len = 0.  NULL on error.
uint8_t *hg_hts_pack(hg_ctx *ctx, const uint8_t *data, int64_t len, uint8_t *out_meta, int *out_meta_len, uint64_t *out_len) {
    if (!ctx || !data || len < 0 || (uint64_t)len > MAX_LEN || !out_meta || !out_meta_len || !out_len) return nullptr;
    uint8_t *out = (uint8_t *)malloc((size_t)len + 1);
    if (!out) return nullptr;
    if (len == 0) { out_meta[0] = 0; *out_meta_len = 1; *out_len = 0; return out; }
    hg::CtxGuard guard_(ctx);
    if (guard_.rc) { free(out); return nullptr; }
    hg::nx16_xenc J; hg::nx16_xenc_res R;
    if (run_xenc(ctx, data, (uint64_t)len, 0x80u, nullptr, 0, &J, &R) != HG_OK) { free(out); return nullptr; }
    if (!(R.flags & 0x80u)) {                               // more than 16 symbols
        memcpy(out, data, (size_t)len);
        out_meta[0] = (uint8_t)R.nsym; *out_meta_len = 1; *out_len = (uint64_t)len;
        return out;
    }
    out_meta[0] = (uint8_t)R.nsym;
    memcpy(out_meta + 1, R.map, R.nsym);
    *out_meta_len = 1 + (int)R.nsym;
    *out_len = R.nsym > 1 ? R.plen : 0;
    if (!fetch(ctx, out, R.cur_off, *out_len)) { free(out); return nullptr; }
    return out;
}

// hts_unpack (htscodecs/pack.h): nsym is the number of symbols PER BYTE as hts_unpack_meta reports it -- 8, 4 or 2
// (1-, 2-, 4-bit codes), 1 = the data was not packed (copy), 0 = one constant symbol p[0].  Returns out, or NULL
// when `data` is too short for out_len symbols.
uint8_t *hg_hts_unpack(hg_ctx *ctx, const uint8_t *data, int64_t len, uint8_t *out, uint64_t out_len, int nsym, const uint8_t *p) {
    if (!ctx || !out || len < 0 || (uint64_t)len > MAX_LEN || out_len > MAX_LEN || (!data && len) || !p) return nullptr;
    if (nsym == 1) { if ((uint64_t)len < out_len) return nullptr; memcpy(out, data, out_len); return out; }   // identity
    if (nsym == 0) { memset(out, p[0], out_len); return out; }
    if (nsym != 8 && nsym != 4 && nsym != 2) return nullptr;
    if ((out_len + (uint64_t)nsym - 1) / (uint64_t)nsym > (uint64_t)len) return nullptr;
    if (!out_len) return out;
    hg::CtxGuard guard_(ctx);
    if (guard_.rc) return nullptr;
    hg::nx16_xform J;
    memset(&J, 0, sizeof J);
    J.s1_off = 0; J.lit_len = (uint32_t)len; J.ulen = (uint32_t)out_len; J.plen = (uint32_t)len;
    J.ops = 2u; J.stride = 1; J.dep0 = J.dep1 = 0xffffffffu;
    J.nsym = nsym == 8 ? 2u : nsym == 4 ? 4u : 16u;
    memcpy(J.map, p, J.nsym);
    const uint64_t job_off = al((uint64_t)len), st_off = job_off + al(sizeof J);
    if (ensure_scratch(ctx, 0, st_off + 64) != HG_OK || ensure_scratch(ctx, 1, out_len + 64) != HG_OK) return nullptr;
    uint8_t *d = (uint8_t *)ctx->d_scratch[0], *d_out = (uint8_t *)ctx->d_scratch[1];
    hipStream_t s = ctx->stream;
    int32_t st = -1;
    if (hipMemcpyAsync(d, data, (size_t)len, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d + job_off, &J, sizeof J, hipMemcpyHostToDevice, s) != hipSuccess ||
        hg::launch_ransnx16_xform(ctx, d, d, d_out, (const hg::nx16_xform *)(d + job_off), 1, (int32_t *)(d + st_off), 0, s) != HG_OK ||
        hipMemcpyAsync(&st, d + st_off, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(out, d_out, out_len, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return nullptr;
    return st == 0 ? out : nullptr;
}

// hts_rle_encode (htscodecs/rle.h): every symbol listed in rle_syms is written once per run to the literal stream
// and its run length - 1 as a 7-bit varint to `run`; other symbols are literals.  *rle_nsyms == 0 on entry: the
// symbols whose repeats outnumber their run starts are chosen and returned through rle_syms / rle_nsyms.  `run`
// needs room for data_len + 8 bytes; out == NULL: a buffer is malloc'd.  Returns the literal buffer (*out_len
// bytes), NULL on error.
uint8_t *hg_hts_rle_encode(hg_ctx *ctx, const uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms,
                           int *rle_nsyms, uint8_t *out, uint64_t *out_len) {
    if (!ctx || (!data && data_len) || data_len > MAX_LEN || !run || !run_len || !rle_syms || !rle_nsyms || !out_len ||
        *rle_nsyms < 0 || *rle_nsyms > 256) return nullptr;
    uint8_t *lit = out ? out : (uint8_t *)malloc((size_t)data_len * 2 + 1);
    if (!lit) return nullptr;
    *run_len = 0; *out_len = 0;
    if (!data_len) return lit;
    hg::CtxGuard guard_(ctx);
    hg::nx16_xenc J; hg::nx16_xenc_res R;
    const uint32_t np = (uint32_t)*rle_nsyms;
    if (guard_.rc || run_xenc(ctx, data, data_len, 0x40u | (np ? 0x100u : 0u), rle_syms, np, &J, &R) != HG_OK) {
        if (!out) free(lit);
        return nullptr;
    }
    if (!(R.flags & 0x40u)) {                               // no symbol is worth it: all literals, no run lengths
        memcpy(lit, data, (size_t)data_len);
        *out_len = data_len; *rle_nsyms = 0;
        return lit;
    }
    // device meta stream = [count][symbols][run lengths]
    uint8_t hdr[257];
    if (!fetch(ctx, hdr, J.m_off, 1)) { if (!out) free(lit); return nullptr; }
    const uint32_t nr = hdr[0] ? hdr[0] : 256u;
    if (!np) {
        if (!fetch(ctx, rle_syms, J.m_off + 1, nr)) { if (!out) free(lit); return nullptr; }
        *rle_nsyms = (int)nr;
    }
    *run_len = R.meta_len - (1u + nr);
    *out_len = R.lit_len;
    if (!fetch(ctx, run, J.m_off + 1 + nr, *run_len) || !fetch(ctx, lit, R.cur_off, *out_len)) { if (!out) free(lit); return nullptr; }
    return lit;
}

// hts_rle_decode (htscodecs/rle.h): the inverse.  *out_len is the room in `out` on entry and the number of bytes
// produced on return; NULL when a run does not fit, the run-length stream is malformed or runs short.
uint8_t *hg_hts_rle_decode(hg_ctx *ctx, const uint8_t *lit, uint64_t lit_len, const uint8_t *run, uint64_t run_len, const uint8_t *rle_syms,
                           uint32_t rle_nsyms, uint8_t *out, uint64_t *out_len) {
    if (!ctx || (!lit && lit_len) || (!run && run_len) || !out || !out_len || lit_len > MAX_LEN || run_len > MAX_LEN || *out_len > MAX_LEN ||
        rle_nsyms > 256 || (rle_nsyms && !rle_syms)) return nullptr;
    if (!lit_len) { *out_len = 0; return out; }
    if (!rle_nsyms) {                                       // nothing is run-length coded: the literals are the data
        if (lit_len > *out_len) return nullptr;
        memcpy(out, lit, lit_len); *out_len = lit_len;
        return out;
    }
    hg::CtxGuard guard_(ctx);
    if (guard_.rc) return nullptr;
    const uint64_t meta_len = 1 + (uint64_t)rle_nsyms + run_len;
    hg::nx16_xform J;
    memset(&J, 0, sizeof J);
    J.s1_off = 0; J.lit_len = (uint32_t)lit_len;
    J.meta_off = al(lit_len); J.meta_len = (uint32_t)meta_len;
    J.len_off = J.meta_off + al(meta_len);
    J.plen = (uint32_t)*out_len; J.ulen = J.plen;
    J.ops = 1u | 4u | 8u; J.stride = 1; J.dep0 = J.dep1 = 0xffffffffu;
    const uint64_t job_off = J.len_off + 64, st_off = job_off + al(sizeof J);
    if (ensure_scratch(ctx, 0, st_off + 64) != HG_OK || ensure_scratch(ctx, 1, *out_len + 64) != HG_OK) return nullptr;
    uint8_t *d = (uint8_t *)ctx->d_scratch[0], *d_out = (uint8_t *)ctx->d_scratch[1];
    hipStream_t s = ctx->stream;
    uint8_t hdr[257];
    hdr[0] = (uint8_t)rle_nsyms; memcpy(hdr + 1, rle_syms, rle_nsyms);
    int32_t st = -1; unsigned long long produced = 0;
    if (hipMemcpyAsync(d, lit, lit_len, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d + J.meta_off, hdr, 1 + rle_nsyms, hipMemcpyHostToDevice, s) != hipSuccess ||
        (run_len && hipMemcpyAsync(d + J.meta_off + 1 + rle_nsyms, run, run_len, hipMemcpyHostToDevice, s) != hipSuccess) ||
        hipMemcpyAsync(d + job_off, &J, sizeof J, hipMemcpyHostToDevice, s) != hipSuccess ||
        hg::launch_ransnx16_xform(ctx, d, d, d_out, (const hg::nx16_xform *)(d + job_off), 1, (int32_t *)(d + st_off), 0, s) != HG_OK ||
        hipMemcpyAsync(&st, d + st_off, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&produced, d + J.len_off, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return nullptr;
    if (st != 0 || produced > *out_len) return nullptr;
    if (produced && hipMemcpy(out, d_out, produced, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    *out_len = produced;
    return out;
}

}  // extern "C"

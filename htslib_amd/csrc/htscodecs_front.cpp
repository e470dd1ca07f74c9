// htscodecs_front.cpp -- the entry points of samtools/htscodecs that htslib calls, under htscodecs' own names, on the
// gfx950 engine.  Host C++ only; every codec runs in libhtsgpu.so.
//
// htslib reaches its CRAM entropy coders through ten functions of the htscodecs library (bundled submodule or, with
// HAVE_EXTERNAL_LIBHTSCODECS, a separate libhtscodecs.so -- htslib's configure --with-external-htscodecs).  Call sites:
//   cram/cram_io.c:1668  rans_uncompress            :1838  rans_compress             (CRAM 3.0 method 4, rANS 4x8)
//   cram/cram_io.c:1699  rans_uncompress_4x16       :1859  rans_compress_4x16        (method 5, rANS Nx16)
//   cram/cram_io.c:1718  arith_uncompress_to        :1879  arith_compress_to         (method 6, adaptive range coder)
//   cram/cram_io.c:1686  fqz_decompress             :1821  fqz_compress              (method 7, fqzcomp)
//   cram/cram_io.c:1737  tok3_decode_names          :1891  tok3_encode_names         (method 8, name tokeniser)
//   hts.c:149,225,229    htscodecs_version
// (hts_pack / hts_unpack / hts_rle_*, cram/cram_codecs.c:1399-2278, live in cram_block_front.cpp.)
// With these exported, libhts_bgzf.so also stands in for libhtscodecs.so: an htslib built --with-external-htscodecs links it instead
// and keeps its own cram_io.c.  One call = one stream = one engine launch, so this is the compatibility route; the block functions
// (cram_uncompress_block / cram_compress_block2, which coalesce concurrent callers) and the slice batches are the fast one.
// Buffers are returned malloc'd, as the callers free() them (cram_io.c:1675,1705 ...).  No CPU codec exists behind these names:
// without an engine every call returns NULL, which htslib reports as a failed block.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "hts_cram_gpu.h"
#include "htsgpu.h"

namespace hgfront { hg_ctx *shared_engine(); }

namespace {

// flag bits of the first byte of an Nx16 / arith stream (cram/cram_external.c:616-637)
constexpr int F_ORDER = 0x01, F_X32 = 0x04, F_NOSZ = 0x10;
constexpr int SIMD_AUTO = 0x8000;                       // RANS_ORDER_SIMD_AUTO: "pick 4-way or 32-way yourself" (cram_io.c:1859)
constexpr uint32_t X32_FROM = 64u << 10;                // the engine's choice, as in hg_cram_compress_blocks_host

// big-endian base-128 integer, top bit = "more" (the uint7 of the CRAM 3.1 codecs); 0 = malformed
int get_u7(const uint8_t *p, const uint8_t *end, uint32_t *v) {
    uint32_t x = 0;
    for (int n = 1; n <= 5 && p < end; n++) {
        const uint8_t c = *p++;
        x = (x << 7) | (c & 0x7f);
        if (!(c & 0x80)) { *v = x; return n; }
    }
    return 0;
}

uint8_t *room(size_t n) { return (uint8_t *)malloc(n ? n : 1); }

// method 5 / 6 streams carry their plain size after the flag byte unless NOSZ (only inside STRIPE / tok3 sub-streams)
template <class Decode>
uint8_t *sized_decode(const uint8_t *in, unsigned int in_size, unsigned int *out_size, Decode dec) {
    hg_ctx *ctx = hgfront::shared_engine();
    uint32_t usz = 0;
    if (!ctx || !in || in_size < 2 || (in[0] & F_NOSZ) || !get_u7(in + 1, in + in_size, &usz) || usz > 0x7fffffffu) return nullptr;
    uint8_t *out = room(usz);
    if (!out) return nullptr;
    int32_t st = -1;
    uint8_t *outs[1] = {out};
    const uint8_t *ins[1] = {in};
    const uint32_t il = in_size;
    if (dec(ctx, ins, &il, outs, &usz, &st) != HG_OK || st != 0) { free(out); return nullptr; }
    *out_size = usz;
    return out;
}

}  // namespace

extern "C" {

const char *htscodecs_version(void) { return "gfx950 engine (htslib_amd)"; }

// ---- CRAM 3.0 rANS 4x8: order byte, compressed size le32, plain size le32 -----------------------------------------------------------
unsigned char *rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order) {
    hg_ctx *ctx = hgfront::shared_engine();
    if (!ctx || !out_size) return nullptr;
    uint8_t *out = room(hg_rans4x8_compress_bound(in_size));
    if (!out) return nullptr;
    const uint8_t *ins[1] = {in}; uint8_t *outs[1] = {out};
    const uint32_t il = in_size; uint32_t ol = 0; const uint8_t ord = order ? 1 : 0;
    if (hg_rans4x8_encode_host(ctx, ins, &il, &ord, 1, outs, &ol) != HG_OK || !ol) { free(out); return nullptr; }
    *out_size = ol;
    return out;
}
unsigned char *rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size) {
    hg_ctx *ctx = hgfront::shared_engine();
    if (!ctx || !in || in_size < 9 || !out_size) return nullptr;
    const uint32_t usz = in[5] | in[6] << 8 | in[7] << 16 | (uint32_t)in[8] << 24;
    if (usz > 0x7fffffffu) return nullptr;
    uint8_t *out = room(usz);
    if (!out) return nullptr;
    const uint8_t *ins[1] = {in}; uint8_t *outs[1] = {out};
    const uint32_t il = in_size; uint32_t got = 0; int32_t st = -1;
    if (hg_rans4x8_decode_host(ctx, ins, &il, 1, outs, &usz, &got, &st) != HG_OK || st != 0 || got != usz) { free(out); return nullptr; }
    *out_size = usz;
    return out;
}

// ---- CRAM 3.1 rANS Nx16 ----------------------------------------------------------------------------------------------------------------
unsigned int rans_compress_bound_4x16(unsigned int size, int order) { (void)order; return (unsigned int)hg_ransnx16_compress_bound(size); }
unsigned char *rans_compress_to_4x16(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order) {
    hg_ctx *ctx = hgfront::shared_engine();
    if (!ctx || !out_size) return nullptr;
    uint8_t flags = (uint8_t)(order & 0xff);
    if ((order & SIMD_AUTO) && in_size >= X32_FROM) flags |= F_X32;
    const size_t need = hg_ransnx16_compress_bound(in_size);
    uint8_t *buf = out;
    if (buf) { if (*out_size < need) return nullptr; } else if (!(buf = room(need))) return nullptr;
    const uint8_t *ins[1] = {in}; uint8_t *outs[1] = {buf};
    const uint32_t il = in_size; uint32_t ol = 0;
    if (hg_ransnx16_encode_host(ctx, ins, &il, &flags, 1, outs, &ol) != HG_OK || !ol) { if (!out) free(buf); return nullptr; }
    *out_size = ol;
    return buf;
}
unsigned char *rans_compress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order) {
    return rans_compress_to_4x16(in, in_size, nullptr, out_size, order);
}
unsigned char *rans_uncompress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size) {
    if (!out_size) return nullptr;
    return sized_decode(in, in_size, out_size, [](hg_ctx *c, const uint8_t *const *i, const uint32_t *il, uint8_t *const *o, const uint32_t *ol, int32_t *st) {
        return hg_ransnx16_decode_host(c, i, il, 1, o, ol, st);
    });
}

// ---- CRAM 3.1 adaptive range coder -----------------------------------------------------------------------------------------------------
unsigned int arith_compress_bound(unsigned int size, int order) { (void)order; return (unsigned int)hg_arith_compress_bound(size); }
unsigned char *arith_compress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order) {
    hg_ctx *ctx = hgfront::shared_engine();
    if (!ctx || !out_size) return nullptr;
    const uint8_t flags = (uint8_t)(order & 0xff);
    const size_t need = hg_arith_compress_bound(in_size);
    uint8_t *buf = out;
    if (buf) { if (*out_size < need) return nullptr; } else if (!(buf = room(need))) return nullptr;
    const uint8_t *ins[1] = {in}; uint8_t *outs[1] = {buf};
    const uint32_t il = in_size; uint32_t ol = 0;
    if (hg_arith_encode_host(ctx, ins, &il, &flags, 1, outs, &ol) != HG_OK || !ol) { if (!out) free(buf); return nullptr; }
    *out_size = ol;
    return buf;
}
unsigned char *arith_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order) {
    return arith_compress_to(in, in_size, nullptr, out_size, order);
}
unsigned char *arith_uncompress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size) {
    if (!out_size) return nullptr;
    auto dec = [](hg_ctx *c, const uint8_t *const *i, const uint32_t *il, uint8_t *const *o, const uint32_t *ol, int32_t *st) {
        return hg_arith_decode_host(c, i, il, 1, o, ol, st);
    };
    if (!out) return sized_decode(in, in_size, out_size, dec);
    // caller's buffer: *out_size = its capacity.  The size the stream declares is checked against it BEFORE anything is allocated or launched (a few
    // bytes of hostile input must not buy a 2 GiB allocation and a device launch)
    {
        uint32_t usz = 0;
        if (!in || in_size < 2 || (in[0] & F_NOSZ) || !get_u7(in + 1, in + in_size, &usz) || usz > *out_size) return nullptr;
    }
    unsigned int got = 0;
    uint8_t *tmp = sized_decode(in, in_size, &got, dec);
    if (!tmp) return nullptr;
    if (got > *out_size) { free(tmp); return nullptr; }
    memcpy(out, tmp, got);
    free(tmp);
    *out_size = got;
    return out;
}
unsigned char *arith_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size) { return arith_uncompress_to(in, in_size, nullptr, out_size); }

// ---- CRAM 3.1 name tokeniser: plain size le32, number of names le32, use_arith byte, token streams ------------------------------------
uint8_t *tok3_encode_names(char *blk, int len, int level, int use_arith, int *out_len, int *last_start_p) {
    (void)level;                                                     // the level picks how many back-end flag sets are tried; the engine tries its own
    hg_ctx *ctx = hgfront::shared_engine();
    if (!ctx || len < 0 || !out_len || last_start_p) return nullptr; // last_start_p: partial-buffer mode of the tokeniser's own CLI, not used by htslib (NULL at cram_io.c:1891)
    uint8_t *out = room(hg_tok3_compress_bound((size_t)len));
    if (!out) return nullptr;
    const uint8_t *ins[1] = {(const uint8_t *)blk}; uint8_t *outs[1] = {out};
    const uint32_t il = (uint32_t)len; uint32_t ol = 0; const uint8_t ua = use_arith ? 1 : 0;
    if (hg_tok3_encode_host(ctx, ins, &il, &ua, 1, outs, &ol) != HG_OK || !ol) { free(out); return nullptr; }
    *out_len = (int)ol;
    return out;
}
uint8_t *tok3_decode_names(uint8_t *in, uint32_t sz, uint32_t *out_len) {
    hg_ctx *ctx = hgfront::shared_engine();
    if (!ctx || !in || sz < 9 || !out_len) return nullptr;
    const uint32_t usz = in[0] | in[1] << 8 | in[2] << 16 | (uint32_t)in[3] << 24;
    if (usz > 0x7fffffffu) return nullptr;
    uint8_t *out = room(usz);
    if (!out) return nullptr;
    const uint8_t *ins[1] = {in}; uint8_t *outs[1] = {out};
    int32_t st = -1;
    if (hg_tok3_decode_host(ctx, ins, &sz, 1, outs, &usz, &st) != HG_OK || st != 0) { free(out); return nullptr; }
    *out_len = usz;
    return out;
}

// ---- CRAM 3.1 fqzcomp ------------------------------------------------------------------------------------------------------------------
// fqz_slice as cram_io.c:1808-1820 fills it (num_records, len[], flags[]) = hg_fqz_slice; gp = caller-chosen parameters (htslib passes NULL)
char *fqz_compress(int vers, void *s, char *in, size_t uncomp_size, size_t *comp_size, int strat, void *gp) {
    (void)vers;
    hg_ctx *ctx = hgfront::shared_engine();
    const hg_fqz_slice *fs = (const hg_fqz_slice *)s;
    if (!ctx || !fs || !comp_size || gp || uncomp_size > 0x7fffffffu) return nullptr;
    uint8_t *out = room(hg_fqz_compress_bound(uncomp_size, fs->num_records));
    if (!out) return nullptr;
    const uint8_t *ins[1] = {(const uint8_t *)in}; uint8_t *outs[1] = {out};
    const hg_fqz_slice *sl[1] = {fs};
    const uint32_t il = (uint32_t)uncomp_size; uint32_t ol = 0; const int32_t st = strat & 3;
    if (hg_fqz_encode_host(ctx, ins, &il, sl, &st, 1, outs, &ol) != HG_OK || !ol) { free(out); return nullptr; }
    *comp_size = ol;
    return (char *)out;
}
char *fqz_decompress(char *in, size_t comp_size, size_t *uncomp_size, int *lengths, int nlengths) {
    (void)lengths; (void)nlengths;                                   // optional by-product (record lengths); htslib passes NULL, 0
    hg_ctx *ctx = hgfront::shared_engine();
    uint32_t usz = 0;
    if (!ctx || !in || !uncomp_size || comp_size > 0xffffffffu || !get_u7((const uint8_t *)in, (const uint8_t *)in + comp_size, &usz)) return nullptr;
    uint8_t *out = room(usz);
    if (!out) return nullptr;
    const uint8_t *ins[1] = {(const uint8_t *)in}; uint8_t *outs[1] = {out};
    const uint32_t il = (uint32_t)comp_size; int32_t st = -1;
    if (hg_fqz_decode_host(ctx, ins, &il, 1, outs, &usz, &st) != HG_OK || st != 0) { free(out); return nullptr; }
    *uncomp_size = usz;
    return (char *)out;
}

}  // extern "C"

/* TEST DOUBLE -- never built into, shipped with, or loaded by the product.
 *
 * A host-only stand-in for the handful of libhtsgpu entry points that the BGZF front-end
 * (htslib_amd/csrc/bgzf_front.cpp) calls, implemented with the system zlib.  It exists so that the
 * front-end's HOST logic -- state machine, I/O and output threads, seek / index / EOF handling -- can be
 * exercised by `pytest -m "not gpu"` on a machine without an MI355X, by running the reference's own
 * test/test_bgzf.c and bgzip.c against  bgzf_front.cpp + this file  (tests/test_front_host_logic.py).
 * The product library libhts_bgzf.so links the real engine and fails with ENODEV when there is no GPU.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <zlib.h>
#include "htsgpu.h"

/* "devices": the double has as many as HTS_GPU_DEVICES names; FAKE_ENGINE_REPORT=1 prints how many batches each one got when the process ends
 * (tests/test_front_host_logic.py: a handle spreads its windows over all devices and still delivers them in order) */
struct hg_ctx { int device; };
static long g_jobs[64];
static void report(void) { if (getenv("FAKE_ENGINE_REPORT")) { fprintf(stderr, "fake_engine jobs per device:"); for (int i = 0; i < 64; i++) if (g_jobs[i]) fprintf(stderr, " %d=%ld", i, g_jobs[i]); fprintf(stderr, "\n"); } }
struct hg_pipe {
    int device;
    uint8_t *in; size_t in_cap;
    uint8_t *out; size_t out_cap, out_len;
    int32_t *status; uint64_t *off; uint32_t *crc;
    size_t n; int kind;
};

int hg_init(int device, hg_ctx **ctx) {
    static int once; if (!once) { once = 1; atexit(report); }
    if (device < 0 || device >= 64) return HG_ENODEV;
    *ctx = calloc(1, sizeof(hg_ctx)); if (*ctx) (*ctx)->device = device;
    return *ctx ? HG_OK : HG_ENOMEM;
}
void hg_destroy(hg_ctx *ctx) { free(ctx); }
int hg_pipe_create(hg_ctx *ctx, hg_pipe **p) { *p = calloc(1, sizeof(hg_pipe)); if (*p) (*p)->device = ctx->device; return *p ? HG_OK : HG_ENOMEM; }
void hg_pipe_destroy(hg_pipe *p) { if (!p) return; free(p->in); free(p->out); free(p->status); free(p->off); free(p->crc); free(p); }

void *hg_pipe_input(hg_pipe *p, size_t bytes) {
    if (p->kind) return NULL;
    if (p->in_cap < bytes + 64) { free(p->in); p->in = malloc(bytes + 64); p->in_cap = p->in ? bytes + 64 : 0; }
    return p->in;
}
int hg_pipe_reserve(hg_pipe *p, size_t in_bytes, size_t out_bytes) {
    if (p->kind) return HG_EINVAL;
    if (!hg_pipe_input(p, in_bytes)) return HG_ENOMEM;
    if (p->out_cap < out_bytes) { free(p->out); p->out = malloc(out_bytes); p->out_cap = p->out ? out_bytes : 0; }
    return p->out ? HG_OK : HG_ENOMEM;
}
static int grow_out(hg_pipe *p, size_t need) {
    if (p->out_cap < need) { free(p->out); p->out = malloc(need); p->out_cap = p->out ? need : 0; }
    return p->out ? 0 : -1;
}

int hg_pipe_inflate(hg_pipe *p, size_t comp_len, const hg_bgzf_desc *desc, size_t n) {
    uint64_t plain = 0;
    (void)comp_len;
    __sync_fetch_and_add(&g_jobs[p->device], 1);
    for (size_t i = 0; i < n; i++) plain += desc[i].ulen;
    if (grow_out(p, plain + 64)) return HG_ENOMEM;
    free(p->status); p->status = calloc(n ? n : 1, sizeof(int32_t));
    for (size_t i = 0; i < n; i++) {
        const uint8_t *b = p->in + desc[i].coff;
        z_stream z; memset(&z, 0, sizeof z);
        inflateInit2(&z, -15);
        z.next_in = (Bytef *)b + 18; z.avail_in = desc[i].clen - 26;
        z.next_out = p->out + desc[i].uoff; z.avail_out = desc[i].ulen;
        int r = inflate(&z, Z_FINISH);
        uint32_t want_crc, isize;
        memcpy(&want_crc, b + desc[i].clen - 8, 4); memcpy(&isize, b + desc[i].clen - 4, 4);
        if (r != Z_STREAM_END || z.total_out != desc[i].ulen || isize != desc[i].ulen) p->status[i] = HG_BLOCK_EINFLATE;
        else if (crc32(0, p->out + desc[i].uoff, desc[i].ulen) != want_crc) p->status[i] = HG_BLOCK_ECRC;
        inflateEnd(&z);
    }
    p->n = n; p->out_len = plain; p->kind = 1;
    return HG_OK;
}

int hg_pipe_deflate(hg_pipe *p, size_t len, const uint64_t *cuts, size_t n, int level, int raw) {
    (void)len;
    __sync_fetch_and_add(&g_jobs[p->device], 1);
    if (grow_out(p, n * 65536 + 64)) return HG_ENOMEM;
    free(p->off); free(p->crc);
    p->off = calloc(n + 1, 8); p->crc = calloc(n ? n : 1, 4);
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *src = p->in + cuts[i]; const size_t sl = cuts[i + 1] - cuts[i];
        uint8_t *o = p->out + pos;
        const size_t hdr = raw ? 0 : 18;
        z_stream z; memset(&z, 0, sizeof z);
        deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        z.next_in = (Bytef *)src; z.avail_in = sl; z.next_out = o + hdr; z.avail_out = 65536 - hdr - 8;
        deflate(&z, raw ? Z_SYNC_FLUSH : Z_FINISH);
        size_t cl = z.total_out;
        deflateEnd(&z);
        p->off[i] = pos; p->crc[i] = crc32(0, src, sl);
        if (!raw) {
            const uint8_t h[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(o, h, 16);
            const size_t total = 18 + cl + 8;
            o[16] = (uint8_t)(total - 1); o[17] = (uint8_t)((total - 1) >> 8);
            const uint32_t c = p->crc[i], u = (uint32_t)sl;
            memcpy(o + 18 + cl, &c, 4); memcpy(o + 18 + cl + 4, &u, 4);
            cl = total;
        }
        pos += cl;
    }
    p->off[n] = pos; p->out_len = pos; p->n = n; p->kind = 2;
    return HG_OK;
}

int hg_pipe_wait(hg_pipe *p, const uint8_t **out, size_t *out_len, const int32_t **status, const uint64_t **blk_off, const uint32_t **crc) {
    const int kind = p->kind;
    p->kind = 0;
    if (out) *out = p->out;
    if (out_len) *out_len = p->out_len;
    if (status) *status = p->status;
    if (blk_off) *blk_off = p->off;
    if (crc) *crc = p->crc;
    if (kind == 1) for (size_t i = 0; i < p->n; i++) if (p->status[i]) return HG_EBLOCK;
    return kind ? HG_OK : HG_EINVAL;
}

int hg_bgzf_deflate_host(hg_ctx *ctx, const uint8_t *plain, size_t len, const uint64_t *cuts, size_t ncuts, int level, int add_eof,
                         uint8_t *out, size_t out_cap, size_t *out_len) {
    hg_pipe *p; (void)add_eof;
    hg_pipe_create(ctx, &p);
    memcpy(hg_pipe_input(p, len), plain, len);
    hg_pipe_deflate(p, len, cuts, ncuts, level, 0);
    const uint8_t *o; size_t ol;
    hg_pipe_wait(p, &o, &ol, NULL, NULL, NULL);
    int rc = ol <= out_cap ? HG_OK : HG_EINVAL;
    if (rc == HG_OK) { memcpy(out, o, ol); *out_len = ol; }
    hg_pipe_destroy(p);
    return rc;
}

int hg_crc32_host(hg_ctx *ctx, const void *buf, size_t len, uint32_t *crc) { (void)ctx; *crc = crc32(0, buf, len); return HG_OK; }

/* whole members only: enough for the front-end's plain-gzip fallback logic */
int hg_gzip_stream_inflate_host(hg_ctx *ctx, const uint8_t *comp, size_t comp_len, int comp_eof, hg_gz_state *st,
                                const uint8_t *hist, size_t hist_len, uint8_t *out, size_t out_cap, size_t soft_cap, size_t *out_len) {
    (void)ctx; (void)hist; (void)hist_len; (void)soft_cap;
    *out_len = 0;
    if (st->in_member || (st->in_bit & 7)) return HG_EINVAL;
    const size_t at = st->in_bit >> 3;
    z_stream z; memset(&z, 0, sizeof z);
    inflateInit2(&z, 15 + 16);
    z.next_in = (Bytef *)comp + at; z.avail_in = comp_len - at; z.next_out = out; z.avail_out = out_cap;
    const int r = inflate(&z, Z_FINISH);
    const size_t used = z.total_in, made = z.total_out;
    inflateEnd(&z);
    if (r == Z_STREAM_END) { st->in_bit += (uint64_t)used * 8; *out_len = made; return HG_GZ_MEMBER; }
    if ((r == Z_BUF_ERROR || r == Z_OK) && !comp_eof && z.avail_out) return HG_GZ_NEEDIN;
    return HG_EBLOCK;
}

// hg_pipe.hip -- asynchronous host-buffer job slots ("pipes") of the engine.
//
// The reference overlaps file I/O with (de)compression by running one pool job per block between an I/O thread
// and an ordered result queue (bgzf.c:1598-1738 reader, :1398-1473 writer, thread_pool.c:149-252 ordered results).
// Here the unit in flight is a BATCH of blocks: a pipe owns pinned host buffers, device buffers and one HIP
// stream; submitting a job queues  H2D -> kernel(s) -> D2H  on that stream and returns at once, so that with two
// or three pipes per BGZF handle the file read of batch n+1, the PCIe transfers of batch n and the kernel of batch
// n-1 overlap.  Results are consumed in submission order by the caller (hg_pipe_wait), which is the ordered-queue
// contract of hts_tpool_next_result_wait.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "htsgpu.h"
#include "hg_internal.h"

struct hg_pipe {
    hg_ctx *ctx;
    hipStream_t s;
    hipEvent_t ev;
    uint8_t *h_in, *h_out, *h_meta;                     // pinned host
    size_t h_in_cap, h_out_cap, h_meta_cap;
    uint8_t *d_in, *d_out, *d_meta, *d_slots, *d_tok;   // device (d_tok: the deflate kernel's token lists, the pipe's own so that pipes overlap)
    size_t d_in_cap, d_out_cap, d_meta_cap, d_slots_cap, d_tok_cap;
    int kind;                                            // 0 idle, 1 inflate, 2 deflate
    size_t n;                                            // blocks of the job in flight
    uint64_t out_len;
    int submit_rc;
};

namespace {

int grow_pinned(uint8_t **p, size_t *cap, size_t need) {
    if (*cap >= need) return HG_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 4 + 65536;
    if (hipHostMalloc((void **)p, want, hipHostMallocDefault) != hipSuccess) return HG_ENOMEM;
    *cap = want;
    return HG_OK;
}
int grow_dev(uint8_t **p, size_t *cap, size_t need) {
    if (*cap >= need) return HG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 4 + 65536;
    if (hipMalloc((void **)p, want) != hipSuccess) return HG_ENOMEM;
    *cap = want;
    return HG_OK;
}
inline size_t up16(size_t x) { return (x + 15u) & ~(size_t)15u; }

}  // namespace

extern "C" {

int hg_pipe_create(hg_ctx *ctx, hg_pipe **out) {
    if (!ctx || !out) return HG_EINVAL;
    *out = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    hg_pipe *p = (hg_pipe *)calloc(1, sizeof(hg_pipe));
    if (!p) return HG_ENOMEM;
    p->ctx = ctx;
    if (hipStreamCreateWithFlags(&p->s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev, hipEventDisableTiming) != hipSuccess) {
        if (p->s) (void)hipStreamDestroy(p->s);
        free(p);
        return HG_ENODEV;
    }
    *out = p;
    return HG_OK;
}

void hg_pipe_destroy(hg_pipe *p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->s);
    if (p->h_in) (void)hipHostFree(p->h_in);
    if (p->h_out) (void)hipHostFree(p->h_out);
    if (p->h_meta) (void)hipHostFree(p->h_meta);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    if (p->d_meta) (void)hipFree(p->d_meta);
    if (p->d_slots) (void)hipFree(p->d_slots);
    if (p->d_tok) (void)hipFree(p->d_tok);
    (void)hipEventDestroy(p->ev);
    (void)hipStreamDestroy(p->s);
    free(p);
}

void *hg_pipe_input(hg_pipe *p, size_t bytes) {
    if (!p || p->kind != 0) return nullptr;                           // a job is still in flight
    if (hipSetDevice(p->ctx->device) != hipSuccess) return nullptr;
    if (grow_pinned(&p->h_in, &p->h_in_cap, bytes + 64) != HG_OK) return nullptr;
    return p->h_in;
}

int hg_pipe_reserve(hg_pipe *p, size_t in_bytes, size_t out_bytes) {
    if (!p || p->kind != 0) return HG_EINVAL;
    if (hipSetDevice(p->ctx->device) != hipSuccess) return HG_ENODEV;
    int rc;
    if ((rc = grow_pinned(&p->h_in, &p->h_in_cap, in_bytes + 64)) || (rc = grow_pinned(&p->h_out, &p->h_out_cap, out_bytes + 64)) ||
        (rc = grow_dev(&p->d_in, &p->d_in_cap, in_bytes + 256)) || (rc = grow_dev(&p->d_out, &p->d_out_cap, out_bytes + 256))) return rc;
    return HG_OK;
}

int hg_pipe_inflate(hg_pipe *p, size_t comp_len, const hg_bgzf_desc *desc, size_t n) {
    if (!p || p->kind != 0 || (n && !desc) || comp_len + 64 > p->h_in_cap) return HG_EINVAL;
    if (n > 0xffffffffull) return HG_EINVAL;
    if (hipSetDevice(p->ctx->device) != hipSuccess) return HG_ENODEV;
    uint64_t plain = 0;
    for (size_t i = 0; i < n; i++) {
        if (desc[i].coff + desc[i].clen > comp_len || desc[i].uoff != plain) return HG_EINVAL;
        plain += desc[i].ulen;
    }
    p->n = n; p->out_len = plain; p->kind = 1; p->submit_rc = HG_OK;
    if (n == 0) return HG_OK;
    const size_t dsz = up16(n * sizeof(hg_bgzf_desc)), ssz = up16(n * sizeof(int32_t));
    int rc;
    if ((rc = grow_pinned(&p->h_meta, &p->h_meta_cap, dsz + ssz)) || (rc = grow_pinned(&p->h_out, &p->h_out_cap, (size_t)plain + 64)) ||
        (rc = grow_dev(&p->d_in, &p->d_in_cap, comp_len + 256)) || (rc = grow_dev(&p->d_out, &p->d_out_cap, (size_t)plain + 256)) ||
        (rc = grow_dev(&p->d_meta, &p->d_meta_cap, dsz + ssz))) { p->kind = 0; return rc; }
    memcpy(p->h_meta, desc, n * sizeof(hg_bgzf_desc));
    memset(p->h_in + comp_len, 0, 8);                                  // the kernel reads whole dwords
    // The kernel reads the compressed window and the block table straight out of the pinned host buffers and writes the block
    // verdicts there (hipHostMalloc memory is mapped into the device's address space): a wavefront fetches its input in 256-byte
    // windows, one ahead of the one being decoded, so the PCIe latency is off the decode chain -- and no host-to-device copy sits
    // in the copy engine's queue behind the previous job's device-to-host copy (which waits for ITS kernel): measured on 8 MiB
    // windows, the staged form ran D2H -> H2D -> kernel strictly one after the other across three pipes (17 GB/s of plain BAM).
    // HTS_GPU_STAGE_INPUT=1 restores the staged form.
    static const bool stage_in = getenv("HTS_GPU_STAGE_INPUT") != nullptr;
    bool ok = true;
    const uint8_t *k_in = p->h_in; const hg_bgzf_desc *k_desc = (const hg_bgzf_desc *)p->h_meta; int32_t *k_status = (int32_t *)(p->h_meta + dsz);
    if (stage_in) {
        const size_t comp_pad = (comp_len + 3) & ~(size_t)3;
        ok = hipMemcpyAsync(p->d_in, p->h_in, comp_pad, hipMemcpyHostToDevice, p->s) == hipSuccess &&
             hipMemcpyAsync(p->d_meta, p->h_meta, n * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice, p->s) == hipSuccess;
        k_in = p->d_in; k_desc = (const hg_bgzf_desc *)p->d_meta; k_status = (int32_t *)(p->d_meta + dsz);
    }
    rc = ok ? hg::launch_bgzf_inflate(p->ctx, k_in, comp_len, k_desc, n, p->d_out, (size_t)plain, k_status, p->s) : HG_ELAUNCH;
    if (rc == HG_OK) {
        ok = (!stage_in || hipMemcpyAsync(p->h_meta + dsz, p->d_meta + dsz, n * sizeof(int32_t), hipMemcpyDeviceToHost, p->s) == hipSuccess) &&
             (plain == 0 || hipMemcpyAsync(p->h_out, p->d_out, (size_t)plain, hipMemcpyDeviceToHost, p->s) == hipSuccess) &&
             hipEventRecord(p->ev, p->s) == hipSuccess;
        if (!ok) rc = HG_ELAUNCH;
    }
    if (rc != HG_OK) { (void)hipStreamSynchronize(p->s); p->kind = 0; }
    return rc;
}

int hg_pipe_deflate(hg_pipe *p, size_t len, const uint64_t *cuts, size_t n, int level, int raw) {
    if (!p || p->kind != 0 || !cuts || n == 0 || level < 0 || level > 9 || len + 64 > p->h_in_cap) return HG_EINVAL;
    if (hipSetDevice(p->ctx->device) != hipSuccess) return HG_ENODEV;
    if (cuts[0] != 0 || cuts[n] != len) return HG_EINVAL;
    // h_meta / d_meta layout: desc[n] | clen[n] | crc[n] | poff[n + 1] (the last one = stream length)
    const size_t dsz = up16(n * sizeof(hg_bgzf_desc)), csz = up16(n * 4), osz = up16((n + 1) * 8);
    const size_t slots = n * (size_t)HG_BGZF_MAX_BLOCK_SIZE;
    int rc;
    if ((rc = grow_pinned(&p->h_meta, &p->h_meta_cap, dsz + 2 * csz + osz)) || (rc = grow_dev(&p->d_meta, &p->d_meta_cap, dsz + 2 * csz + osz)) ||
        (rc = grow_dev(&p->d_in, &p->d_in_cap, len + 256)) || (rc = grow_dev(&p->d_slots, &p->d_slots_cap, slots + 256)) ||
        (rc = grow_dev(&p->d_out, &p->d_out_cap, slots + 256)) || (rc = grow_dev(&p->d_tok, &p->d_tok_cap, hg::bgzf_deflate_tok_bytes_for(p->ctx, n)))) return rc;
    hg_bgzf_desc *d = (hg_bgzf_desc *)p->h_meta;
    for (size_t i = 0; i < n; i++) {
        if (cuts[i + 1] < cuts[i] || cuts[i + 1] - cuts[i] > HG_BGZF_BLOCK_SIZE) return HG_EINVAL;
        d[i].uoff = cuts[i]; d[i].ulen = (uint32_t)(cuts[i + 1] - cuts[i]);
        d[i].coff = (uint64_t)i * HG_BGZF_MAX_BLOCK_SIZE; d[i].clen = 0;      // raw: no chunk carries BFINAL (the caller ends the member)
    }
    p->n = n; p->kind = 2; p->out_len = 0; p->submit_rc = HG_OK;
    uint32_t *d_clen = (uint32_t *)(p->d_meta + dsz), *d_crc = (uint32_t *)(p->d_meta + dsz + csz);
    uint64_t *d_poff = (uint64_t *)(p->d_meta + dsz + 2 * csz);
    bool ok = (len == 0 || hipMemcpyAsync(p->d_in, p->h_in, len, hipMemcpyHostToDevice, p->s) == hipSuccess) &&
              hipMemcpyAsync(p->d_meta, p->h_meta, n * sizeof(hg_bgzf_desc), hipMemcpyHostToDevice, p->s) == hipSuccess &&
              hipMemsetAsync(d_crc, 0, n * 4, p->s) == hipSuccess;
    rc = ok ? hg::launch_bgzf_deflate(p->ctx, p->d_in, (const hg_bgzf_desc *)p->d_meta, n, level, p->d_slots, d_clen, p->s,
                                      raw ? 1 : 0, d_crc, p->d_tok) : HG_ELAUNCH;
    if (rc == HG_OK)
        rc = hg::launch_bgzf_pack(p->ctx, p->d_slots, (const hg_bgzf_desc *)p->d_meta, d_clen, n, p->d_out, slots + 256, d_poff, d_poff + n,
                                  0, p->s);
    if (rc == HG_OK) {
        ok = hipMemcpyAsync(p->h_meta + dsz, p->d_meta + dsz, 2 * csz + osz, hipMemcpyDeviceToHost, p->s) == hipSuccess &&
             hipEventRecord(p->ev, p->s) == hipSuccess;
        if (!ok) rc = HG_ELAUNCH;
    }
    if (rc != HG_OK) { (void)hipStreamSynchronize(p->s); p->kind = 0; }
    return rc;
}

int hg_pipe_wait(hg_pipe *p, const uint8_t **out, size_t *out_len, const int32_t **status, const uint64_t **blk_off,
                 const uint32_t **crc) {
    if (!p || p->kind == 0) return HG_EINVAL;
    if (hipSetDevice(p->ctx->device) != hipSuccess) return HG_ENODEV;
    const int kind = p->kind;
    const size_t n = p->n;
    p->kind = 0;
    if (out) *out = nullptr;
    if (out_len) *out_len = 0;
    if (status) *status = nullptr;
    if (blk_off) *blk_off = nullptr;
    if (crc) *crc = nullptr;
    if (n == 0) return HG_OK;
    if (hipEventSynchronize(p->ev) != hipSuccess) return HG_ELAUNCH;
    const size_t dsz = up16(n * sizeof(hg_bgzf_desc));
    if (kind == 1) {
        const int32_t *st = (const int32_t *)(p->h_meta + dsz);
        if (out) *out = p->h_out;
        if (out_len) *out_len = (size_t)p->out_len;
        if (status) *status = st;
        for (size_t i = 0; i < n; i++) if (st[i] != HG_BLOCK_OK) return HG_EBLOCK;
        return HG_OK;
    }
    const size_t csz = up16(n * 4);
    const uint64_t *poff = (const uint64_t *)(p->h_meta + dsz + 2 * csz);
    const uint64_t total = poff[n];
    if (total > n * (uint64_t)HG_BGZF_MAX_BLOCK_SIZE) return HG_ELAUNCH;
    if (grow_pinned(&p->h_out, &p->h_out_cap, (size_t)total + 64) != HG_OK) return HG_ENOMEM;
    if (total && (hipMemcpyAsync(p->h_out, p->d_out, (size_t)total, hipMemcpyDeviceToHost, p->s) != hipSuccess ||
                  hipStreamSynchronize(p->s) != hipSuccess)) return HG_ELAUNCH;
    if (out) *out = p->h_out;
    if (out_len) *out_len = (size_t)total;
    if (blk_off) *blk_off = poff;
    if (crc) *crc = (const uint32_t *)(p->h_meta + dsz + csz);
    return HG_OK;
}

}  // extern "C"

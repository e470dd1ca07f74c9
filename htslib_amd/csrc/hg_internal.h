// hg_internal.h -- host-side internals shared by the C-ABI shim and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "htsgpu.h"

#define HG_SCRATCH_SLOTS 16
struct hg_ctx {
    int device;
    int cus;
    int waves_per_launch;     // resident wavefronts the persistent kernels are sized for
    unsigned int *d_ticket;   // ring of HG_TICKETS work-queue counters (device): one per launch, see next_ticket()
    std::atomic<unsigned int> *launch_seq;   // launches issued so far (any thread, any stream)
    std::mutex *tok_mu;       // orders deflate launches (shared token lists, see launch_bgzf_deflate)
    hipEvent_t ev_deflate; int ev_deflate_used;
    std::recursive_mutex *mu;           // serialises the host-buffer entry points of this context (they share scratch + staging)
    // scratch for the host-buffer convenience entry points (grown on demand)
    void *d_scratch[HG_SCRATCH_SLOTS];
    size_t d_scratch_cap[HG_SCRATCH_SLOTS];
    void *d_tok;              // deflate token lists, 256 KiB per resident workgroup
    size_t d_tok_cap;
    void *h_stage[2];         // pinned staging buffers of the host entry points (0: upload, 1: download)
    size_t h_stage_cap[2];
    hipStream_t stream;       // the host entry points run on this non-blocking stream (one per context)
    hipStream_t stream2;      // side stream: the second of two independent kernel variants of one call (fork_side / join_side)
    hipEvent_t ev_fork, ev_join;
    hipStream_t stream3;      // a second side stream (a third variant of one call: the range coder's global-model streams, the long 4-way rANS streams)
    hipEvent_t ev_fork3, ev_join3;
    hipStream_t stream4;      // a third side stream: the two-phase range-coder encoder (arith_enc2.hip) beside the one-pass kernels
    hipEvent_t ev_fork4, ev_join4;
    hg_ctx *sub[8];           // lazily created sibling contexts: independent codec families of one CRAM batch run concurrently
    void *h_slab[4];          // pageable host buffers the file-level entry points keep between calls (host_slab(): a writer comes back with the next
    size_t h_slab_cap[4];     // chunk, and first-touch page faults on fresh hundreds-of-MB buffers cost more than the device work they hold)
};

// Persistent kernels pull block indices from a counter that must start at 0.  Launches of one context may overlap
// (caller streams, double-buffered pipelines, the side stream), so every launch takes the next counter of a ring and
// zeroes it on its own stream; HG_TICKETS bounds the number of launches of one context that may be in flight at once.
#define HG_TICKETS 256
namespace hg {
// slot 0: series blocks of the record encoder, 1: compressed payloads of a CRAM writer call, 2: a codec family's trial results (in its sibling
// context).  Contents are NOT kept across a growth; the caller holds the context lock (CtxGuard) for as long as it uses the pointer.
inline uint8_t *host_slab(hg_ctx *ctx, int slot, size_t bytes) {
    if (ctx->h_slab_cap[slot] < bytes || !ctx->h_slab[slot]) {
        free(ctx->h_slab[slot]);
        const size_t cap = bytes + bytes / 4 + 4096;
        ctx->h_slab[slot] = malloc(cap);
        ctx->h_slab_cap[slot] = ctx->h_slab[slot] ? cap : 0;
    }
    return (uint8_t *)ctx->h_slab[slot];
}
inline unsigned int *next_ticket(hg_ctx *ctx) {
    return ctx->d_ticket + (ctx->launch_seq->fetch_add(1u, std::memory_order_relaxed) % HG_TICKETS);
}
// RAII guard of a host entry point: binds the device and holds the context lock
struct CtxGuard {
    std::unique_lock<std::recursive_mutex> lk;
    int rc;
    explicit CtxGuard(hg_ctx *ctx) : lk(*ctx->mu), rc(hipSetDevice(ctx->device) == hipSuccess ? HG_OK : HG_ENODEV) {}
};
int launch_bgzf_inflate(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc,
                        size_t nblocks, void *d_out, size_t out_cap, int32_t *d_status, hipStream_t s, int mode = 0);
int launch_bgzf_deflate(hg_ctx *ctx, const void *d_plain, const hg_bgzf_desc *d_desc, size_t nblocks, int level,
                        void *d_slots, uint32_t *d_clen, hipStream_t s, int mode = 0, uint32_t *d_crc = nullptr, void *own_tok = nullptr);
size_t bgzf_deflate_tok_bytes_for(const hg_ctx *ctx, size_t nblocks);   // ... of a launch over nblocks blocks (a writer pipe sizes its lists by its job)
size_t bgzf_deflate_tok_bytes(const hg_ctx *ctx);     // token lists of one launch (own_tok: a caller that brings its own may overlap its launches)
int launch_bgzf_pack(hg_ctx *ctx, const void *d_slots, const hg_bgzf_desc *d_desc, const uint32_t *d_clen,
                     size_t nblocks, void *d_packed, size_t cap, uint64_t *d_poff, uint64_t *d_total, int add_eof,
                     hipStream_t s);
int launch_rans4x8_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, void *d_out,
                          int32_t *d_status, uint32_t *d_scratch, hipStream_t s);
int launch_rans4x16_big_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4, size_t n4, void *d_out, int32_t *d_status,
                               uint32_t *d_scratch, hipStream_t s, bool leave_room);
int launch_ransnx16_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4,
                           size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out, int32_t *d_status,
                           uint32_t *d_scratch, hipStream_t s);
// One post-entropy-decode job of an Nx16 stream (ransnx16_xform.hip): [RLE expand] -> [bit unpack] -> strided write.
struct nx16_xform {
    uint64_t s1_off;       // work buffer: entropy-decoded bytes (lit_len of them)
    uint64_t meta_off;     // RLE meta stream: in the work buffer (ops & 4) or in the input buffer
    uint64_t s2_off;       // work buffer: RLE output when a PACK step follows
    uint64_t out_off;      // output buffer: first byte written
    uint32_t lit_len, meta_len, plen, ulen;
    uint32_t stride;       // output byte i goes to out_off + i * stride (STRIPE de-interleave)
    uint32_t ops;          // 1 RLE, 2 PACK, 4 meta lives in the work buffer, 8 plen is a CAPACITY (hts_rle_decode): the
                           //   number of bytes produced is stored as a uint64 at work + len_off
    uint32_t nsym;         // PACK symbol count (<= 16)
    uint32_t dep0, dep1;   // status slots of the entropy-decode jobs this one consumes (0xffffffff = none)
    uint8_t map[16];
    uint32_t pad;
    uint64_t len_off;
};
static_assert(sizeof(nx16_xform) == 96, "nx16_xform layout");
int launch_ransnx16_xform(hg_ctx *ctx, const void *d_in, void *d_work, void *d_out, const nx16_xform *d_jobs, size_t njobs,
                          int32_t *d_status, uint32_t status_base, hipStream_t s);
// Encoder-side transform job (ransnx16_xenc.hip): [gather stripe] -> [PACK] -> [RLE]; offsets into ONE buffer.
struct nx16_xenc {
    uint64_t src_off, g_off, p_off, l_off, m_off;
    uint32_t n, stride, flags, pad;     // flags: 0x80 PACK, 0x40 RLE, 0x100 the run symbols are PRESET at m_off ([count][symbols])
};
struct nx16_xenc_res {
    uint64_t cur_off;      // where the bytes to entropy-code ended up
    uint32_t cur_len, flags, nsym, plen, lit_len, meta_len;
    uint8_t map[16];
    uint64_t pad;
};
int launch_ransnx16_xenc(hg_ctx *ctx, void *d_buf, const nx16_xenc *d_jobs, size_t njobs, nx16_xenc_res *d_res, hipStream_t s, const nx16_xenc *h_jobs = nullptr);
// arith.hip: model memory is an LDS pool per wavefront; streams are sorted into a small-pool launch (many
// waves per CU) and a big-pool launch, larger models fall back to global scratch words.
#define HG_ARITH_POOL_SMALL 3072     /* words: order-0 (+RLE), order-1 up to 54 symbols (41 with RLE) */
#define HG_ARITH_POOL_BIG   16384    /* words: order-1 up to 127 symbols, or 120 with RLE */
#define HG_ARITH_POOL_TOTALS 768     /* words: the totals of a stream whose models live in global scratch (256 literal + 258 run models) */
uint32_t arith_model_words(uint32_t max_sym, uint32_t flags);
int launch_arith_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel_small, size_t nsmall,
                        const uint32_t *d_sel_big, size_t nbig, void *d_out, int32_t *d_status, uint32_t *d_scratch, hipStream_t s);
int launch_arith_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags, const uint32_t *d_sel_small,
                        size_t nsmall, const uint32_t *d_sel_big, size_t nbig, void *d_out, uint32_t *d_out_len, uint32_t *d_scratch,
                        hipStream_t s);
// arith_enc2.hip: the two-phase encoder (d_sel2: indices into d_desc).  Round 5: the events of a stream are sorted by model first, so the work is proportional to
// the events and the form pays from a few KiB on (round 4: every model's task walked its whole stream; two phases only from 256 KiB in big batches).
// HG_ARITH_2P_MIN overrides the threshold, HG_ARITH_2P=0 sends every stream through the one-pass kernels (A/B runs).
#define HG_ARITH_2P_MIN 1024u
#define HG_ARITH_2P_MAX (1u << 29)     /* record numbers are stored << 2 in 32-bit list words: at most 2 n < 2^30 events */
// per stream in the model scratch (hg_stream_desc::scratch_off): alphabet size, number of events, list sizes, list offsets per literal context / run symbol
struct Arith2pInfo { uint32_t m, nevents, n_r2, n_r3, pad[4]; uint32_t offL[260], offR[260]; };
#define HG_ARITH_2P_INFO_WORDS ((uint32_t)(sizeof(hg::Arith2pInfo) / 4))
// per stream in the work buffer (hg_stream_desc::reserved * 16): the dense records (8 bytes per event: n events, at most 2 n with RLE), the literal events'
// record numbers and symbols, and with RLE the first-part / second-part / further-part lists of the runs
struct Arith2pLayout { uint64_t rec, lidx, lsym, r1, r2, r3, end; };
__host__ __device__ inline Arith2pLayout arith2p_layout(uint32_t n, bool rle) {
    Arith2pLayout L;
    uint64_t o = 0;
    auto take = [&](uint64_t bytes) { const uint64_t at = o; o += (bytes + 15u) & ~15ull; return at; };
    L.rec = take((rle ? 2ull : 1ull) * n * 8u);
    L.lidx = take(4ull * n);
    L.lsym = take(n);
    L.r1 = L.r2 = L.r3 = o;
    if (rle) { L.r1 = take(4ull * n); L.r2 = take(4ull * (n / 4u + 2u)); L.r3 = take(8ull * (n / 7u + 2u)); }
    L.end = o;
    return L;
}
int launch_range_code(const hg_stream_desc *d_desc, size_t n, const uint32_t *d_scratch, const void *d_work, void *d_out, uint32_t *d_out_len, hipStream_t s);
inline uint32_t arith2p_max_tasks(uint32_t flags) { return (flags & 64u) ? 514u : (flags & 1u) ? 256u : 1u; }
int launch_arith_encode2(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags, const uint32_t *d_sel2, size_t n2, void *d_tasks,
                         size_t task_cap, void *d_out, uint32_t *d_out_len, uint32_t *d_scratch, void *d_work, hipStream_t s);
// tok3.hip: one name-reconstruction job per CRAM method-8 block
struct tok3_job {
    uint64_t tb_base;      // where this block's decoded token streams start in the token buffer
    uint64_t out_off;      // output buffer
    uint64_t rec_off;      // first token record (3 words each)
    uint64_t name_off;     // per-name arrays: noff[nn+1], first[nn+1], ntok[nn+1] (words)
    uint32_t tab_off;      // stream table: (offset, length) words for [position][16 types]
    uint32_t ntp, nn, ulen, rec_cap, pad;
};
int launch_tok3_names(hg_ctx *ctx, const void *d_tb, const tok3_job *d_jobs, size_t njobs, const uint32_t *d_tab, void *d_out,
                      uint32_t *d_rec, uint32_t *d_names, int32_t *d_status, hipStream_t s);
// tok3.hip encoder side: tokenise one block of NUL-terminated names into its token byte streams
#define HG_TOK3_MAX_STREAMS (13 * 128)
struct tok3_enc_job { uint64_t in_off, sb_off; uint32_t n, sb_cap; };
struct tok3_enc_stream { uint32_t off, len; uint8_t pos, type, ttype, dup_pos, dup_type, pad[3]; };   // in emission order
struct tok3_enc_res { uint32_t nn, nstreams, total, pad; };                                       // total = 0xffffffff: did not fit
int launch_tok3_tokenise(hg_ctx *ctx, const void *d_in, const tok3_enc_job *d_jobs, size_t njobs, void *d_sb, tok3_enc_stream *d_list,
                         tok3_enc_res *d_res, hipStream_t s);
int ensure_scratch(hg_ctx *ctx, int slot, size_t bytes);
// Two kernel variants of one call (4-way / 32-way rANS, small / big range-coder pool) work on disjoint streams of the
// batch and are each latency-bound: the second one goes to the context's side stream, ordered after everything already
// queued on s, and s then waits for it.
// The side streams are created when a call first forks onto them (round 6): a context that never runs two variants side by side keeps ONE stream.  Streams are
// not free: creating four costs ~40 ms of a process's start-up (experiments/startup_probe.cpp), and every live stream holds a place on one of the device's
// GPU_MAX_HW_QUEUES hardware queues -- a CRAM batch with six codec-family contexts of four streams each had more streams than queues, and which two families'
// long kernels ended up behind each other on one queue depended on what the process had created before (bench.py: cram_slices decode 21.9 GB/s alone, 15-17 inside
// the default run).  The caller holds the context lock.
inline hipStream_t side_stream(hipStream_t &st, hipStream_t fallback) {
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return fallback; }
    return st;
}
inline hipStream_t fork_side(hg_ctx *ctx, hipStream_t s) {
    hipStream_t t = side_stream(ctx->stream2, s);
    if (t == s) return s;                                                // (no side stream to be had: the variant runs in line)
    (void)hipEventRecord(ctx->ev_fork, s);
    (void)hipStreamWaitEvent(t, ctx->ev_fork, 0);
    return t;
}
inline void join_side(hg_ctx *ctx, hipStream_t s) {
    if (!ctx->stream2) return;
    (void)hipEventRecord(ctx->ev_join, ctx->stream2);
    (void)hipStreamWaitEvent(s, ctx->ev_join, 0);
}
inline hipStream_t fork_side3(hg_ctx *ctx, hipStream_t s) {
    hipStream_t t = side_stream(ctx->stream3, s);
    if (t == s) return s;
    (void)hipEventRecord(ctx->ev_fork3, s);
    (void)hipStreamWaitEvent(t, ctx->ev_fork3, 0);
    return t;
}
inline void join_side3(hg_ctx *ctx, hipStream_t s) {
    if (!ctx->stream3) return;
    (void)hipEventRecord(ctx->ev_join3, ctx->stream3);
    (void)hipStreamWaitEvent(s, ctx->ev_join3, 0);
}
inline hipStream_t fork_side4(hg_ctx *ctx, hipStream_t s) {
    hipStream_t t = side_stream(ctx->stream4, s);
    if (t == s) return s;
    (void)hipEventRecord(ctx->ev_fork4, s);
    (void)hipStreamWaitEvent(t, ctx->ev_fork4, 0);
    return t;
}
inline void join_side4(hg_ctx *ctx, hipStream_t s) {
    if (!ctx->stream4) return;
    (void)hipEventRecord(ctx->ev_join4, ctx->stream4);
    (void)hipStreamWaitEvent(s, ctx->ev_join4, 0);
}
// hg_stage.hip: many scattered host buffers <-> one device buffer, one PCIe transfer each way
int stage_upload(hg_ctx *ctx, const uint8_t *const *src, const uint32_t *len, const uint64_t *dst_off, const int32_t *skip, size_t n,
                 uint64_t total, uint8_t *d_base, hipStream_t s);
int stage_download(hg_ctx *ctx, const uint8_t *d_base, const uint64_t *src_off, const uint32_t *len, uint8_t *const *dst, size_t n,
                   hipStream_t s);
int stage_gather_dev(hg_ctx *ctx, const uint8_t *d_src, const uint64_t *src_off, const uint32_t *len, uint8_t *d_dst, const uint64_t *dst_off,
                     size_t n, hipStream_t s);
void stage_free(hg_ctx *ctx);
uint32_t ransnx16_enc_scratch_words(uint32_t flags);
int launch_ransnx16_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags,
                           const uint32_t *d_sel4, size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out,
                           uint32_t *d_out_len, void *d_wbuf, uint32_t *d_scratch, hipStream_t s);
uint32_t rans4x8_enc_scratch_words(uint32_t order);
int launch_rans4x8_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_order, size_t n,
                          void *d_out, uint32_t *d_out_len, void *d_wbuf, uint32_t *d_scratch, hipStream_t s);
int launch_crc32(hg_ctx *ctx, const void *d_data, const uint64_t *d_off, const uint32_t *d_len, size_t n,
                 uint32_t *d_crc, hipStream_t s);
}

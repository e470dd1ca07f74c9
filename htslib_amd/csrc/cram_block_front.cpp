// cram_block_front.cpp -- htslib's CRAM block layer (cram_uncompress_block / cram_compress_block, cram_block framing)
// on the gfx950 batch engine.  Host C++ only; every codec runs in libhtsgpu.so.  Interfaces and their reference
// anchors are in include/hts_cram_gpu.h.
//
// The reference calls these functions one block at a time from many pool workers (one slice each); a
// single block cannot fill a GPU.  The single-block entry points therefore COALESCE concurrent callers: the first
// caller to arrive becomes the leader, lingers for a moment so that the other workers can join, runs ONE engine batch
// for everybody and hands the results back.  A lone caller pays the linger (tens of microseconds) and gets its block
// alone -- correct, just not fast; the array forms (a whole slice per call) are the intended hook
// (cram_decode.c:624-627 loops over a slice's blocks; cram_encode.c:803-988 compresses them in one function).
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>
#include "hts_cram_gpu.h"
#include "hts_hfile_abi.h"
#include "htsgpu.h"

extern "C" void hts_log(int severity, const char *context, const char *format, ...) __attribute__((weak));

static_assert(sizeof(cram_metrics) == sizeof(hg_cram_metrics), "cram_metrics and the engine's view of it must agree");

namespace {

void logerr(const char *ctx, const char *fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (hts_log) hts_log(1, ctx, "%s", buf); else fprintf(stderr, "[E::%s] %s\n", ctx, buf);
}

hg_ctx *engine() {                                   // one context for the block layer; its host calls lock it
    static hg_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *d = getenv("HTS_GPU_DEVICE");
        if (hg_init(d ? atoi(d) : 0, &ctx) != HG_OK) ctx = nullptr;
    });
    return ctx;
}

// CRC-32 of the few framing bytes of a block header (<= 17 bytes; the payload's CRC is computed on the device)
uint32_t crc_small(uint32_t crc, const uint8_t *p, size_t n) {
    crc = ~crc;
    for (size_t i = 0; i < n; i++) { crc ^= p[i]; for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u))); }
    return ~crc;
}
uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
uint32_t crc_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    uint32_t xp = 0x00800000u, acc = 0x80000000u;
    for (; len_b; len_b >>= 1) { if (len_b & 1) acc = crc_mulmod(acc, xp); xp = crc_mulmod(xp, xp); }
    return crc_mulmod(acc, crc_a) ^ crc_b;
}

// ---- variable-length integers of the block header ------------------------------------------------------------
int itf8_put(uint8_t *p, int32_t sv) {
    const uint32_t v = (uint32_t)sv;
    if (v < 0x80u) { p[0] = (uint8_t)v; return 1; }
    if (v < 0x4000u) { p[0] = (uint8_t)(0x80 | (v >> 8)); p[1] = (uint8_t)v; return 2; }
    if (v < 0x200000u) { p[0] = (uint8_t)(0xc0 | (v >> 16)); p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)v; return 3; }
    if (v < 0x10000000u) { p[0] = (uint8_t)(0xe0 | (v >> 24)); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; return 4; }
    p[0] = (uint8_t)(0xf0 | (v >> 28)); p[1] = (uint8_t)(v >> 20); p[2] = (uint8_t)(v >> 12); p[3] = (uint8_t)(v >> 4); p[4] = (uint8_t)(v & 0x0f);
    return 5;
}
int uint7_put(uint8_t *p, uint32_t v) {                       // big-endian base 128, MSB = "more" (cram_io.c:892-990)
    int n = 1;
    for (uint32_t t = v >> 7; t; t >>= 7) n++;
    for (int i = 0; i < n; i++) p[i] = (uint8_t)(((v >> (7 * (n - 1 - i))) & 0x7f) | (i + 1 < n ? 0x80 : 0));
    return n;
}
int varint_put(uint8_t *p, int major, int32_t v) { return major >= 4 ? uint7_put(p, (uint32_t)v) : itf8_put(p, v); }

// reads one header integer, appending its raw bytes to hdr[] (for the CRC); -1 at EOF
int varint_get(hFILE *fp, int major, int32_t *out, uint8_t *hdr, size_t *hl) {
    auto next = [&]() -> int {
        const int c = fp->end > fp->begin ? (unsigned char)*fp->begin++ : hgetc2(fp);
        if (c >= 0) hdr[(*hl)++] = (uint8_t)c;
        return c;
    };
    int c = next();
    if (c < 0) return -1;
    if (major >= 4) {
        uint32_t v = (uint32_t)c & 0x7f;
        for (int k = 0; (c & 0x80) && k < 5; k++) { if ((c = next()) < 0) return -1; v = (v << 7) | ((uint32_t)c & 0x7f); }
        *out = (int32_t)v;
        return 0;
    }
    const int extra = c < 0x80 ? 0 : c < 0xc0 ? 1 : c < 0xe0 ? 2 : c < 0xf0 ? 3 : 4;
    uint32_t v = (uint32_t)c & (0xffu >> (extra + (extra < 4 ? 1 : 0)));
    if (extra == 4) v = (uint32_t)c & 0x0f;
    for (int k = 0; k < extra; k++) {
        if ((c = next()) < 0) return -1;
        v = (extra == 4 && k == 3) ? (v << 4) | ((uint32_t)c & 0x0f) : (v << 8) | (uint32_t)c;
    }
    *out = (int32_t)v;
    return 0;
}

// ================================================================================ bzip2 / lzma blocks
// CRAM methods 2 and 3 are general-purpose CPU codecs the reference itself only reaches through libbz2 / liblzma
// (cram_io.c:1626-1664, HAVE_LIBBZ2 / HAVE_LIBLZMA); there is no data-parallel form of either worth a kernel.  The front-end does what
// the reference does: hand the block to the system's library -- looked up at run time, so that the library has no link-time dependency --
// and fail with the reference's message when it is absent.  Nothing of the GPU path runs through here.
namespace hostlib {
typedef int (*bz2_fn)(char *, unsigned int *, char *, unsigned int, int, int);
typedef int (*lzma_fn)(uint64_t *, uint32_t, const void *, const uint8_t *, size_t *, size_t, uint8_t *, size_t *, size_t);
bz2_fn bz2 = nullptr; lzma_fn lzma = nullptr;
std::once_flag once;
void load() {
    for (const char *n : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) { bz2 = (bz2_fn)dlsym(h, "BZ2_bzBuffToBuffDecompress"); if (bz2) break; }
    for (const char *n : {"liblzma.so.5", "liblzma.so"}) if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) { lzma = (lzma_fn)dlsym(h, "lzma_stream_buffer_decode"); if (lzma) break; }
}
// 0 / -1, the block left untouched on failure
int inflate(cram_block *x, bool keep_input) {
    std::call_once(once, load);
    const bool is_bz2 = x->method == BZIP2;
    if (is_bz2 ? !bz2 : !lzma) {
        logerr("cram_uncompress_block", "%s compression is not compiled into this version. Please rebuild and try again", is_bz2 ? "Bzip2" : "Lzma");
        return -1;
    }
    uint8_t *out = (uint8_t *)malloc(x->uncomp_size ? (size_t)x->uncomp_size : 1);
    if (!out) return -1;
    size_t got = 0;
    bool ok;
    if (is_bz2) {
        unsigned int usize = (unsigned int)x->uncomp_size;
        ok = bz2((char *)out, &usize, (char *)x->data, (unsigned int)x->comp_size, 0, 0) == 0;     // BZ_OK
        got = usize;
    } else {
        uint64_t memlimit = UINT64_MAX; size_t in_pos = 0;
        ok = lzma(&memlimit, 0, nullptr, x->data, &in_pos, (size_t)x->comp_size, out, &got, (size_t)x->uncomp_size) == 0 && in_pos == (size_t)x->comp_size;   // LZMA_OK, whole input used
    }
    if (!ok || got != (size_t)x->uncomp_size) { free(out); return -1; }   // the reference's size check (cram_io.c:1639-1642, 1655-1658)
    if (!keep_input) free(x->data);
    x->data = out; x->alloc = got; x->method = RAW;
    return 0;
}
}  // namespace hostlib

// ================================================================================ batch workers
// Decode a set of blocks in one engine round.  rc[i] = 0 / -1.  keep_input: the compressed bytes a block came with are NOT freed when its data
// pointer is replaced (read-ahead works on shadow copies of the callers' blocks: the originals keep their payload until they are asked for).
void uncompress_batch(cram_block **b, int n, int *rc, bool keep_input = false) {
    hg_ctx *ctx = engine();
    std::vector<int> todo;                                     // blocks that need the device
    for (int i = 0; i < n; i++) rc[i] = 0;
    // ---- 1. block CRCs (cram_io.c:1585-1592), all unchecked blocks in one device batch
    std::vector<int> chk;
    for (int i = 0; i < n; i++) if (b[i]->crc32_checked == 0) chk.push_back(i);
    if (!chk.empty()) {
        std::vector<const uint8_t *> p(chk.size()); std::vector<uint32_t> len(chk.size()), crc(chk.size(), 0);
        static const uint8_t none[1] = {0};
        for (size_t k = 0; k < chk.size(); k++) { cram_block *x = b[chk[k]]; p[k] = x->data ? x->data : none; len[k] = x->data ? (uint32_t)x->alloc : 0u; }
        const int r = ctx ? hg_crc32_batch_host(ctx, p.data(), len.data(), chk.size(), crc.data()) : HG_ENODEV;
        for (size_t k = 0; k < chk.size(); k++) {
            cram_block *x = b[chk[k]];
            x->crc32_checked = 1;
            const uint32_t full = len[k] ? crc_concat(x->crc_part, crc[k], len[k]) : x->crc_part;
            if (r != HG_OK || full != x->crc32) { logerr("cram_uncompress_block", r != HG_OK ? "no usable GPU engine" : "Block CRC32 failure"); rc[chk[k]] = -1; }
        }
    }
    // ---- 2. method dispatch
    for (int i = 0; i < n; i++) {
        if (rc[i]) continue;
        cram_block *x = b[i];
        if (x->uncomp_size == 0) { x->method = RAW; continue; }            // blank block (cram_io.c:1594-1598)
        if (x->method == RAW) continue;
        if (x->uncomp_size < 0 || x->comp_size < 0 || (int)x->method < 0 || (int)x->method > TOK3) { rc[i] = -1; continue; }
        if (x->method == BZIP2 || x->method == LZMA) { rc[i] = hostlib::inflate(x, keep_input); continue; }    // the system's library, as in the reference
        todo.push_back(i);
    }
    if (todo.empty()) return;
    const size_t m = todo.size();
    std::vector<int32_t> meth(m), st(m, -1); std::vector<const uint8_t *> in(m); std::vector<uint8_t *> out(m, nullptr);
    std::vector<uint32_t> il(m), ol(m);
    bool oom = false;
    for (size_t k = 0; k < m; k++) {
        cram_block *x = b[todo[k]];
        meth[k] = (int32_t)x->method; in[k] = x->data; il[k] = (uint32_t)x->comp_size; ol[k] = (uint32_t)x->uncomp_size;
        out[k] = (uint8_t *)malloc(ol[k] ? ol[k] : 1);
        if (!out[k]) oom = true;
    }
    int r = oom ? HG_ENOMEM : ctx ? hg_cram_uncompress_blocks_host(ctx, m, meth.data(), in.data(), il.data(), out.data(), ol.data(), st.data()) : HG_ENODEV;
    if (r != HG_OK && r != HG_EBLOCK) { logerr("cram_uncompress_block", "engine failure: %s", hg_strerror(r)); for (auto &s : st) s = -1; }
    for (size_t k = 0; k < m; k++) {
        cram_block *x = b[todo[k]];
        if (st[k] != 0) {
            if (st[k] == HG_BLOCK_EUNSUPPORTED)
                logerr("cram_uncompress_block", "%s compression is not compiled into this version. Please rebuild and try again",
                       x->method == BZIP2 ? "Bzip2" : x->method == LZMA ? "Lzma" : "Fqzcomp");
            free(out[k]);
            rc[todo[k]] = -1;
            continue;
        }
        if (x->method == RANSPR || x->method == ARITH || x->method == TOK3) x->orig_method = x->method;   // cram_io.c:1706,1725,1740
        if (!keep_input) free(x->data);
        x->data = out[k];
        x->alloc = ol[k];
        x->method = RAW;
    }
}

struct CompJob { const hg_cram_opts *opts; cram_block *b; cram_metrics *m; int method, level; int rc; const hg_fqz_slice *fqz = nullptr; };

void compress_batch(CompJob *jobs, int n) {
    hg_ctx *ctx = engine();
    std::vector<int> todo;
    for (int i = 0; i < n; i++) {
        CompJob &j = jobs[i];
        j.rc = 0;
        cram_block *x = j.b;
        if (!x || x->method != RAW) continue;                                  // already compressed (cram_io.c:1945-1952)
        if (j.method == -1) j.method = 1 << GZIP | (j.opts && j.opts->use_bz2 ? 1 << BZIP2 : 0) | (j.opts && j.opts->use_lzma ? 1 << LZMA : 0);   // cram_io.c:1954-1960
        if (j.level == -1) j.level = j.opts ? j.opts->level : 5;
        if (j.method == RAW || j.level == 0 || x->uncomp_size == 0) { x->method = RAW; x->comp_size = x->uncomp_size; continue; }
        todo.push_back(i);
    }
    if (todo.empty()) return;
    // Jobs are grouped by (level, version): the engine call takes one of each.  Metrics objects are updated inside the
    // engine call in block order (the reference's state sequence), under the callers' metrics locks.
    std::sort(todo.begin(), todo.end(), [&](int a, int c) {
        const int va = jobs[a].opts ? jobs[a].opts->version : 0x301, vc = jobs[c].opts ? jobs[c].opts->version : 0x301;
        if (jobs[a].level != jobs[c].level) return jobs[a].level < jobs[c].level;
        if (va != vc) return va < vc;
        return a < c;
    });
    size_t g0 = 0;
    while (g0 < todo.size()) {
        size_t g1 = g0;
        const int level = jobs[todo[g0]].level, version = jobs[todo[g0]].opts ? jobs[todo[g0]].opts->version : 0x301;
        while (g1 < todo.size() && jobs[todo[g1]].level == level && (jobs[todo[g1]].opts ? jobs[todo[g1]].opts->version : 0x301) == version) g1++;
        const size_t m = g1 - g0;
        std::vector<hg_cram_metrics *> met(m); std::vector<uint32_t> set(m), il(m), ol(m, 0); std::vector<const uint8_t *> in(m);
        std::vector<uint8_t *> out(m, nullptr); std::vector<int32_t> used(m, 0);
        std::vector<const hg_fqz_slice *> fq(m, nullptr);
        std::vector<pthread_mutex_t *> locks;
        bool oom = false;
        for (size_t k = 0; k < m; k++) {
            CompJob &j = jobs[todo[g0 + k]];
            met[k] = reinterpret_cast<hg_cram_metrics *>(j.m); set[k] = (uint32_t)j.method;
            in[k] = j.b->data; il[k] = (uint32_t)j.b->uncomp_size; fq[k] = j.fqz;
            out[k] = (uint8_t *)malloc(std::max(hg_cram_compress_bound(il[k]), j.fqz ? hg_fqz_compress_bound(il[k], j.fqz->num_records) : (size_t)0));
            if (!out[k]) oom = true;
            if (j.m && j.opts && j.opts->metrics_lock) locks.push_back((pthread_mutex_t *)j.opts->metrics_lock);
        }
        std::sort(locks.begin(), locks.end());
        locks.erase(std::unique(locks.begin(), locks.end()), locks.end());
        // The callers' metrics locks are held while the metrics are READ (a private copy of every object the batch names) and while the results
        // are WRITTEN BACK, not across the engine call in between: cram_compress_slice takes fd->metrics_lock at its top (cram_encode.c:877), so a
        // lock held for the ~100 ms of a device batch kept every other pool thread from even recording its slice (round 6: one or two slices per
        // batch at -@16).  The reference does the same around its codec calls (cram_io.c:1979-2056, 2101-2236).  One batch at a time works on the
        // copies (batch_mu); `unpackable`, which cram_compress_slice sets from outside, survives a change made during the call.
        static std::mutex batch_mu;
        std::lock_guard<std::mutex> one_batch(batch_mu);
        std::vector<hg_cram_metrics *> uniq;
        for (size_t k = 0; k < m; k++) if (met[k] && std::find(uniq.begin(), uniq.end(), met[k]) == uniq.end()) uniq.push_back(met[k]);
        std::vector<hg_cram_metrics> copy(uniq.size());
        std::vector<int> unp0(uniq.size());
        for (auto l : locks) pthread_mutex_lock(l);
        for (size_t u = 0; u < uniq.size(); u++) { copy[u] = *uniq[u]; unp0[u] = uniq[u]->unpackable; }
        for (auto it = locks.rbegin(); it != locks.rend(); ++it) pthread_mutex_unlock(*it);
        std::vector<hg_cram_metrics *> metc(m, nullptr);
        for (size_t k = 0; k < m; k++) if (met[k]) metc[k] = &copy[(size_t)(std::find(uniq.begin(), uniq.end(), met[k]) - uniq.begin())];
        int r = oom ? HG_ENOMEM : ctx ? hg_cram_compress_blocks_metrics_fqz_host(ctx, m, metc.data(), set.data(), level, version >> 8, in.data(), il.data(),
                                                                                 fq.data(), out.data(), ol.data(), used.data()) : HG_ENODEV;
        for (auto l : locks) pthread_mutex_lock(l);
        for (size_t u = 0; u < uniq.size(); u++) {
            const int now = uniq[u]->unpackable;
            *uniq[u] = copy[u];
            if (now != unp0[u]) uniq[u]->unpackable = now;
        }
        for (auto it = locks.rbegin(); it != locks.rend(); ++it) pthread_mutex_unlock(*it);
        for (size_t k = 0; k < m; k++) {
            CompJob &j = jobs[todo[g0 + k]];
            cram_block *x = j.b;
            if (r != HG_OK) { free(out[k]); j.rc = -1; continue; }
            if (used[k] == HG_CRAM_RAW || ol[k] >= il[k]) {                     // nothing beat the raw bytes (cram_io.c:2271-2278)
                free(out[k]);
                x->method = RAW; x->comp_size = x->uncomp_size;
                continue;
            }
            uint8_t *fit = (uint8_t *)realloc(out[k], ol[k] ? ol[k] : 1);
            free(x->data);
            x->data = fit ? fit : out[k];
            x->alloc = ol[k];
            x->comp_size = (int32_t)ol[k];
            x->method = (enum cram_block_method_int)used[k];                   // already the on-disk id
        }
        if (r != HG_OK) logerr("cram_compress_block", "engine failure: %s", hg_strerror(r));
        g0 = g1;
    }
}

// ================================================================================ coalescing of single-block callers
// Leader / follower: requests queue up; whoever finds no leader becomes one, lingers, takes everything queued so far,
// runs the batch and publishes the results.
template <class Req>
struct Coalescer {
    std::mutex m;
    std::condition_variable cv;
    std::vector<Req *> queue;
    bool leader = false;
    void (*run)(Req **, int);

    // adaptive: a leader waits 1/8 of the previous batch's run time for company, between the fixed linger and 10 ms -- when a batch takes 100 ms (a slice's
    // 1.5 MB quality stream through a 4-way rANS coder is one chain), the pool threads it releases come back within a few ms of each other, and the
    // first of them used to leave alone with ONE slice while the other fourteen waited out its 100 ms (round 6: batches of 1, 14, 1, 14 ... slices)
    bool adaptive = false;
    int last_run_us = 0;

    void submit(Req *r) {
        std::unique_lock<std::mutex> lk(m);
        queue.push_back(r);
        while (!r->done) {
            if (!leader) {
                leader = true;
                const int wait_us = adaptive ? std::max(linger_us(), std::min(10000, last_run_us / 8)) : linger_us();
                lk.unlock();
                std::this_thread::sleep_for(std::chrono::microseconds(wait_us));
                lk.lock();
                std::vector<Req *> batch;
                batch.swap(queue);
                lk.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                run(batch.data(), (int)batch.size());
                const int took = (int)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                lk.lock();
                last_run_us = took;
                for (Req *q : batch) q->done = true;
                leader = false;
                cv.notify_all();
            } else cv.wait(lk);
        }
    }
    static int linger_us() {
        static const int us = [] { const char *e = getenv("HTS_GPU_LINGER_US"); return e ? atoi(e) : 50; }();
        return us;
    }
};

struct UncReq { cram_block *b; int rc; bool done; };
struct CompReq { CompJob job; bool done; };

void run_unc(UncReq **r, int n) {
    std::vector<cram_block *> b(n); std::vector<int> rc(n);
    for (int i = 0; i < n; i++) b[i] = r[i]->b;
    uncompress_batch(b.data(), n, rc.data());
    for (int i = 0; i < n; i++) r[i]->rc = rc[i];
}
void run_comp(CompReq **r, int n) {
    std::vector<CompJob> jobs(n);
    for (int i = 0; i < n; i++) jobs[i] = r[i]->job;
    compress_batch(jobs.data(), n);
    for (int i = 0; i < n; i++) r[i]->job.rc = jobs[i].rc;
}
Coalescer<UncReq> g_unc{{}, {}, {}, false, run_unc};
Coalescer<CompReq> g_comp{{}, {}, {}, false, run_comp};

// ================================================================================ read-ahead decode of the blocks cram_read_block hands out
// The reference decodes a slice's blocks one call at a time (cram_decode_slice's loop, cram/cram_decode.c:624-627), each call a device round trip
// of a millisecond or more here: 25 blocks per slice in a row made `test_view file.cram` on libhts_gpu.so 37 times slower than stock htslib
// (round 6's first libhts-level CRAM figure).  But the library SEES every block long before it is asked to decode it: cram_read_slice
// (cram/cram_io.c) reads all blocks of a slice through cram_read_block -- ours -- on the reading thread, which runs ahead of the pool's decode jobs.
// So a compressed block is queued for decoding the moment it is read; one thread drains the queue in batches (a whole slice, usually several, per
// engine round) into SHADOW copies of the blocks; cram_uncompress_block then only waits for its block's shadow and takes the result over.  The caller's
// block is not touched in between (it keeps its compressed payload: cram_write_block / cram_copy_slice style callers that never decode see what they
// read), cram_free_block drops a result nobody asked for.  HTS_GPU_CRAM_READAHEAD=0 turns it off.
struct Ahead { cram_block shadow; int rc; bool done; };
struct AheadState {
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::unordered_map<cram_block *, Ahead *> map;     // caller's block -> its shadow (until asked for or freed)
    std::vector<Ahead *> queue;
    bool started = false;
};
AheadState &ahead() { static AheadState *a = new AheadState(); return *a; }      // lives until exit (its thread is detached)
bool ahead_enabled() {
    static const bool on = [] { const char *e = getenv("HTS_GPU_CRAM_READAHEAD"); return !(e && e[0] == '0'); }();
    return on;
}
void ahead_main() {
    AheadState &A = ahead();
    for (;;) {
        std::vector<Ahead *> batch;
        {
            std::unique_lock<std::mutex> lk(A.m);
            A.cv_work.wait(lk, [&] { return !A.queue.empty(); });
            lk.unlock();
            std::this_thread::sleep_for(std::chrono::microseconds(150));      // the rest of the slice is microseconds behind its first block
            lk.lock();
            batch.swap(A.queue);
        }
        std::vector<cram_block *> b(batch.size()); std::vector<int> rc(batch.size());
        for (size_t i = 0; i < batch.size(); i++) b[i] = &batch[i]->shadow;
        static const bool stats = getenv("HTS_GPU_STATS") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        size_t cb = 0, ub = 0; int meth[16] = {0};
        if (stats) for (cram_block *x : b) { cb += (size_t)x->comp_size; ub += (size_t)x->uncomp_size; meth[(int)x->method & 15]++; }
        uncompress_batch(b.data(), (int)b.size(), rc.data(), true);
        if (stats) fprintf(stderr, "[htsgpu stats] cram read-ahead batch: %zu blocks (gzip %d bz2 %d lzma %d rans4x8 %d nx16 %d arith %d fqz %d tok3 %d), %.2f -> %.2f MB, %.2f ms\n", b.size(),
                           meth[1], meth[2], meth[3], meth[4], meth[5], meth[6], meth[7], meth[8], cb / 1e6, ub / 1e6,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        {
            std::lock_guard<std::mutex> lk(A.m);
            for (size_t i = 0; i < batch.size(); i++) { batch[i]->rc = rc[i]; batch[i]->done = true; }
        }
        A.cv_done.notify_all();
    }
}
void ahead_submit(cram_block *b) {
    if (!ahead_enabled() || !b || b->method == RAW || b->uncomp_size <= 0 || b->comp_size <= 0) return;
    AheadState &A = ahead();
    Ahead *a = new Ahead{*b, 0, false};
    std::lock_guard<std::mutex> lk(A.m);
    if (!A.started) { A.started = true; std::thread(ahead_main).detach(); }
    A.map[b] = a;
    A.queue.push_back(a);
    A.cv_work.notify_one();
}
// the shadow of a block, finished, removed from the table (nullptr: the block was never queued)
Ahead *ahead_take(cram_block *b) {
    AheadState &A = ahead();
    std::unique_lock<std::mutex> lk(A.m);
    if (A.map.empty()) return nullptr;
    auto it = A.map.find(b);
    if (it == A.map.end()) return nullptr;
    Ahead *a = it->second;
    A.cv_done.wait(lk, [&] { return a->done; });
    A.map.erase(it);
    return a;
}


// ================================================================================ write-behind compression of a slice's blocks
// The reference compresses a slice one cram_compress_block2 call at a time (cram_compress_slice, cram/cram_encode.c:803-988: ~25 calls in a row,
// then a sweep over whatever is still RAW), each call a device round trip here: `test_view -C` on libhts_gpu.so ran 28 times slower than stock
// htslib.  Nothing looks at a block between those calls and cram_encode_slice_header, which starts with cram_new_block -- ours.  So a call that
// names its slice only RECORDS the job (block, metrics, method set, level; a second call for a block of the same slice is the final sweep's: it
// applies if the block is still RAW afterwards); the recorded jobs run as ONE engine batch -- together with the slices other pool threads have
// recorded meanwhile (Coalescer) -- when the thread next enters the block layer with anything else (cram_new_block, cram_free_block,
// cram_write_block, cram_block_size, a block without a slice, another slice).  Metrics see the jobs in the reference's order.  A failed batch
// leaves the blocks RAW (a valid file) and logs the error: the call that could have returned -1 has returned.  HTS_GPU_CRAM_WRITEBEHIND=0: off.
struct ManyReq { CompJob *jobs; int n; bool done; };
void run_many(ManyReq **r, int n) {
    std::vector<CompJob> all;
    for (int i = 0; i < n; i++) all.insert(all.end(), r[i]->jobs, r[i]->jobs + r[i]->n);
    compress_batch(all.data(), (int)all.size());
    size_t at = 0;
    for (int i = 0; i < n; i++) for (int k = 0; k < r[i]->n; k++) r[i]->jobs[k].rc = all[at++].rc;
}
Coalescer<ManyReq> g_many{{}, {}, {}, false, run_many, true};
void compress_many(CompJob *jobs, int n) {
    if (n <= 0) return;
    ManyReq r{jobs, n, false};
    g_many.submit(&r);
}
struct FqzKeep { std::vector<uint32_t> len, flags; hg_fqz_slice fq; };
struct Behind {
    cram_slice *slice = nullptr;
    std::vector<CompJob> first, sweep;                 // the calls of the slice in order / second calls (final sweep)
    std::deque<hg_cram_opts> opts;                     // (jobs point at these)
    std::deque<std::unique_ptr<FqzKeep>> fqz;
    bool empty() const { return first.empty() && sweep.empty(); }
};
thread_local Behind tl_behind;
bool behind_enabled() {
    static const bool on = [] { const char *e = getenv("HTS_GPU_CRAM_WRITEBEHIND"); return !(e && e[0] == '0'); }();
    return on;
}
void behind_flush() {
    Behind &B = tl_behind;
    if (B.empty()) return;
    std::vector<CompJob> first; first.swap(B.first);
    std::vector<CompJob> sweep; sweep.swap(B.sweep);
    B.slice = nullptr;
    compress_many(first.data(), (int)first.size());
    bool bad = false;
    for (auto &j : first) bad |= j.rc != 0;
    std::vector<CompJob> rest;
    for (auto &j : sweep) if (j.b && j.b->method == RAW) rest.push_back(j);
    if (!bad) { compress_many(rest.data(), (int)rest.size()); for (auto &j : rest) bad |= j.rc != 0; }
    if (bad) logerr("cram_compress_block", "a deferred block batch failed: the slice's remaining blocks are written uncompressed");
    B.opts.clear(); B.fqz.clear();
}
inline void behind_sync() { if (!tl_behind.empty()) behind_flush(); }

}  // namespace

// the block layer's context, for the htscodecs-named entry points (htscodecs_front.cpp)
namespace hgfront { hg_ctx *shared_engine() { return engine(); } }
// ... and for the whole-slice reader under cram_get_bam_seq (cram_record_front.c, a C source)
extern "C" hg_ctx *hg_front_shared_engine(void) { return engine(); }

extern "C" {

cram_block *cram_new_block(enum cram_content_type content_type, int content_id) {
    behind_sync();                                                    // (cram_encode_slice_header right after cram_compress_slice comes through here)
    cram_block *b = (cram_block *)calloc(1, sizeof(cram_block));
    if (!b) return nullptr;
    b->method = b->orig_method = RAW;
    b->content_type = content_type;
    b->content_id = content_id;
    b->bit = 7;
    return b;
}

void cram_free_block(cram_block *b) {
    if (!b) return;
    behind_sync();
    if (Ahead *a = ahead_take(b)) {                                   // read ahead, never asked for
        if (a->shadow.data != b->data) free(a->shadow.data);
        delete a;
    }
    free(b->data);
    free(b);
}

cram_metrics *cram_new_metrics(void) { return reinterpret_cast<cram_metrics *>(hg_cram_metrics_new()); }

int cram_uncompress_block(cram_block *b) {
    if (!b) return -1;
    behind_sync();
    if (Ahead *a = ahead_take(b)) {                                   // decoded while it waited for this call: take the result over
        const int rc = a->rc;
        b->crc32_checked = a->shadow.crc32_checked;
        if (rc == 0) {
            if (a->shadow.data != b->data) { free(b->data); b->data = a->shadow.data; b->alloc = a->shadow.alloc; }
            b->method = a->shadow.method; b->orig_method = a->shadow.orig_method;
        } else if (a->shadow.data != b->data) free(a->shadow.data);
        delete a;
        return rc;
    }
    // nothing for the device to do: answer at once (cram_io.c:1594-1603) -- no linger for RAW blocks
    if (b->crc32_checked && (b->uncomp_size == 0 || b->method == RAW)) { if (b->uncomp_size == 0) b->method = RAW; return 0; }
    UncReq r{b, -1, false};
    g_unc.submit(&r);
    return r.rc;
}

int cram_uncompress_blocks(cram_block **b, int n, int *blk_rc) {
    if (n <= 0) return 0;
    std::vector<int> rc(n, 0);
    std::vector<cram_block *> rest; std::vector<int> at;
    for (int i = 0; i < n; i++) {
        if (b[i] && ahead().map.size()) {                               // (a block that was read ahead: its own call collects it)
            AheadState &A = ahead();
            bool queued;
            { std::lock_guard<std::mutex> lk(A.m); queued = A.map.count(b[i]) != 0; }
            if (queued) { rc[i] = cram_uncompress_block(b[i]); continue; }
        }
        rest.push_back(b[i]); at.push_back(i);
    }
    if (!rest.empty()) {
        std::vector<int> r2(rest.size());
        uncompress_batch(rest.data(), (int)rest.size(), r2.data());
        for (size_t k = 0; k < rest.size(); k++) rc[at[k]] = r2[k];
    }
    int any = 0;
    for (int i = 0; i < n; i++) { if (blk_rc) blk_rc[i] = rc[i]; if (rc[i]) any = -1; }
    return any;
}

int hg_cram_compress_block(const hg_cram_opts *opts, cram_block *b, cram_metrics *metrics, int method, int level) {
    if (!b) return 0;
    if (b->method != RAW) return 0;
    CompReq r{{opts, b, metrics, method, level, -1}, false};
    const int lv = level == -1 ? (opts ? opts->level : 5) : level;
    if (method == RAW || lv == 0 || b->uncomp_size == 0) { compress_batch(&r.job, 1); return r.job.rc; }   // no device work
    g_comp.submit(&r);
    return r.job.rc;
}

int hg_cram_compress_blocks(const hg_cram_opts *opts, cram_block **b, cram_metrics **metrics, const int *method, int level, int n) {
    if (n <= 0) return 0;
    std::vector<CompJob> jobs(n);
    for (int i = 0; i < n; i++) jobs[i] = CompJob{opts, b[i], metrics ? metrics[i] : nullptr, method ? method[i] : -1, level, -1};
    compress_batch(jobs.data(), n);
    for (int i = 0; i < n; i++) if (jobs[i].rc) return -1;
    return 0;
}

int hg_cram_compress_blocks_lv(const hg_cram_opts *opts, cram_block **b, cram_metrics **metrics, const int *method, const int *level, int n) {
    return hg_cram_compress_blocks_fqz(opts, b, metrics, method, level, nullptr, n);
}

int hg_cram_compress_block_fqz(const hg_cram_opts *opts, const hg_fqz_slice *fqz, cram_block *b, cram_metrics *metrics, int method, int level) {
    if (!b) return 0;
    if (b->method != RAW) return 0;
    CompReq r{{opts, b, metrics, method, level, -1, fqz}, false};
    const int lv = level == -1 ? (opts ? opts->level : 5) : level;
    if (method == RAW || lv == 0 || b->uncomp_size == 0) { compress_batch(&r.job, 1); return r.job.rc; }
    g_comp.submit(&r);
    return r.job.rc;
}

int hg_cram_compress_blocks_fqz(const hg_cram_opts *opts, cram_block **b, cram_metrics **metrics, const int *method, const int *level,
                                const hg_fqz_slice *const *fqz, int n) {
    if (n <= 0) return 0;
    std::vector<CompJob> jobs(n);
    for (int i = 0; i < n; i++)
        jobs[i] = CompJob{opts, b[i], metrics ? metrics[i] : nullptr, method ? method[i] : -1, level ? level[i] : -1, -1, fqz ? fqz[i] : nullptr};
    compress_batch(jobs.data(), n);
    for (int i = 0; i < n; i++) if (jobs[i].rc) return -1;
    return 0;
}

// ---- cram_compress_slice's policy (cram/cram_encode.c:803-988) ------------------------------------------------------
void hg_cram_slice_method_sets(const hg_cram_slice_opts *o, hg_cram_slice_sets *out) {
    auto bit = [](int m) { return (int)(1u << m); };
    const bool v31 = o->version >= (3 << 8) + 1;
    int method = bit(GZIP) | bit(GZIP_RLE);
    if (o->use_bz2) method |= bit(BZIP2);
    const int rans30 = bit(RANS0) | bit(RANS1);
    int ranspr = rans30;
    if (o->use_rans) {
        ranspr = bit(RANS_PR0) | bit(RANS_PR1);
        if (o->level > 1) ranspr |= bit(RANS_PR64) | bit(RANS_PR9) | bit(RANS_PR128) | bit(RANS_PR193);
        if (o->level > 5) ranspr |= bit(RANS_PR129) | bit(RANS_PR192);
        method |= v31 ? ranspr : rans30;
    }
    if (o->use_arith && v31) {
        method |= bit(ARITH_PR0) | bit(ARITH_PR1);
        if (o->level > 1) method |= bit(ARITH_PR64) | bit(ARITH_PR9) | bit(ARITH_PR128) | bit(ARITH_PR129) | bit(ARITH_PR192) | bit(ARITH_PR193);
    }
    if (o->use_lzma) method |= bit(LZMA);
    int methodF = method & ~(bit(GZIP) | bit(BZIP2) | bit(LZMA));      // entropy coders only, for the series nobody named
    if (o->level >= 5) { method |= bit(GZIP_1); methodF = method; }
    if (o->level == 1) { method = (method & ~bit(GZIP)) | bit(GZIP_1); methodF = method; }
    int q = method;
    if (v31 && o->use_fqz) {
        q |= bit(FQZ);
        if (o->level > 4) q |= bit(FQZ_b);
        if (o->level > 6) q |= bit(FQZ_c) | bit(FQZ_d);
    }
    int rn = method & ~(rans30 | ranspr | bit(GZIP_RLE));
    if (v31 && o->use_tok) rn |= o->use_arith ? bit(TOKA) : bit(TOK3);
    out->method = method; out->methodF = methodF; out->qmethod = q; out->qmethodF = q; out->method_rn = rn;
}

// The calls cram_compress_slice makes before its final sweep, in its order: (data series, method set, level).  Series
// ids >= HG_DS_END are the per-tag aux blocks.  present[ds] != 0: the slice has that block.
int hg_cram_slice_plan(const hg_cram_slice_opts *o, const uint8_t *present, int naux, int core_size, int *ds, int *set, int *lv, int max) {
    hg_cram_slice_sets S;
    hg_cram_slice_method_sets(o, &S);
    int n = 0;
    auto add = [&](int d, int s_, int l) { if ((d >= HG_DS_END || present[d]) && n < max) { ds[n] = d; set[n] = s_; lv[n] = l; n++; } };
    const int level = o->level;
    if (level > 5 && core_size > 500) add(HG_DS_CORE, 1 << GZIP, 1);
    add(HG_DS_IN, S.method, level);
    if (level == 1) {
        add(HG_DS_QS, S.qmethodF, 1);
        for (int i = HG_DS_aux; i <= HG_DS_aux_oz; i++) add(i, S.method, 1);
    } else if (level > 1) {
        const int l = level < 3 ? 1 : level;
        add(HG_DS_QS, S.qmethod, l);
        add(HG_DS_BA, S.method, l);
        add(HG_DS_BB, S.method, l);
        for (int i = HG_DS_aux; i <= HG_DS_aux_oz; i++) add(i, S.method, level);
    }
    add(HG_DS_RN, S.method_rn, level);
    add(HG_DS_NS, S.method, level);
    for (int i = 0; i < naux; i++) add(HG_DS_END + i, S.method, level);
    return n;
}

int hg_cram_compress_slice(const hg_cram_slice_opts *o, const hg_cram_opts *opts, cram_block **block, cram_metrics **metrics,
                           const int *nvals, cram_block **aux, int naux) {
    return hg_cram_compress_slice_fqz(o, opts, block, metrics, nvals, aux, naux, nullptr);
}

int hg_cram_compress_slice_fqz(const hg_cram_slice_opts *o, const hg_cram_opts *opts, cram_block **block, cram_metrics **metrics,
                               const int *nvals, cram_block **aux, int naux, const hg_fqz_slice *qs) {
    hg_cram_slice_sets S;
    hg_cram_slice_method_sets(o, &S);
    std::vector<cram_block *> jb; std::vector<cram_metrics *> jm; std::vector<int> jset, jlv;
    std::vector<const hg_fqz_slice *> jfq;
    cram_block *core = block[HG_DS_CORE];
    auto add = [&](cram_block *b, cram_metrics *m, int set, int lv) {
        if (!b || b->method != RAW) return;                                    // cram_io.c:1945-1952
        for (cram_block *x : jb) if (x == b) return;                           // aliased series: the first call wins
        jb.push_back(b); jm.push_back(m); jset.push_back(set); jlv.push_back(lv);
        jfq.push_back(b == block[HG_DS_QS] ? qs : nullptr);                    // only the quality block is fqzcomp material
    };
    if (nvals && metrics) {                                                    // cram_encode.c:877-881
        pthread_mutex_t *lk = opts ? (pthread_mutex_t *)opts->metrics_lock : nullptr;
        if (lk) pthread_mutex_lock(lk);
        for (int i = 0; i < HG_DS_END; i++) if (nvals[i] > 16 && metrics[i]) metrics[i]->unpackable = 1;
        if (lk) pthread_mutex_unlock(lk);
    }
    uint8_t present[HG_DS_END];
    for (int i = 0; i < HG_DS_END; i++) present[i] = block[i] != nullptr;
    std::vector<int> ds(HG_DS_END + naux + 8), set(ds.size()), lv(ds.size());
    const int n = hg_cram_slice_plan(o, present, naux, core ? core->uncomp_size : 0, ds.data(), set.data(), lv.data(), (int)ds.size());
    for (int k = 0; k < n; k++) {
        const int d = ds[k];
        if (d >= HG_DS_END) { cram_block *b = aux[d - HG_DS_END]; if (b && b != core) add(b, b->m, set[k], lv[k]); }
        else if (d == HG_DS_CORE) add(core, nullptr, set[k], lv[k]);
        else if (!(d == HG_DS_NS && block[d] == core)) add(block[d], metrics ? metrics[d] : nullptr, set[k], lv[k]);
    }
    if (!jb.empty() && hg_cram_compress_blocks_fqz(opts, jb.data(), jm.data(), jset.data(), jlv.data(), jfq.data(), (int)jb.size()) != 0) return -1;
    // final sweep: whatever is still RAW (never named above, or nothing beat the raw bytes) is tried with methodF
    jb.clear(); jm.clear(); jset.clear(); jlv.clear(); jfq.clear();
    for (int i = 1; i < HG_DS_END; i++) if (block[i] && block[i] != core) add(block[i], metrics ? metrics[i] : nullptr, S.methodF, o->level);
    if (!jb.empty() && hg_cram_compress_blocks_fqz(opts, jb.data(), jm.data(), jset.data(), jlv.data(), jfq.data(), (int)jb.size()) != 0) return -1;
    return 0;
}

// ---- the reference's entry points by name, over the real cram_fd / cram_slice of htslib 1.23 (fields read through the offsets of
//      hts_cram_gpu.h, which the layout test asserts against cram/cram_structs.h) ----
}  // extern "C" (helpers below have C++ linkage)
namespace {
template <class T> T field(const void *p, size_t off) { T v; memcpy(&v, (const char *)p + off, sizeof v); return v; }
hg_cram_opts opts_of(cram_fd *fd) {
    hg_cram_opts o; memset(&o, 0, sizeof o);
    o.level = field<int>(fd, HG_CRAM_FD_LEVEL); o.version = field<int>(fd, HG_CRAM_FD_VERSION);
    o.use_bz2 = field<int>(fd, HG_CRAM_FD_USE_BZ2); o.use_lzma = field<int>(fd, HG_CRAM_FD_USE_LZMA);
    o.metrics_lock = (char *)fd + HG_CRAM_FD_METRICS_LOCK;
    return o;
}
}  // namespace
extern "C" {
size_t hg_cram_fd_layout(size_t *o) {
    const size_t v[] = {HG_CRAM_FD_FP, HG_CRAM_FD_VERSION, HG_CRAM_FD_LEVEL, HG_CRAM_FD_IGNORE_MD5, HG_CRAM_FD_USE_BZ2, HG_CRAM_FD_USE_LZMA, HG_CRAM_FD_METRICS_LOCK, HG_CRAM_SLICE_HDR,
                        HG_CRAM_SLICE_BLOCK, HG_CRAM_SLICE_CRECS, HG_CRAM_SLICE_HDR_NUM_RECORDS, HG_CRAM_RECORD_SIZE, HG_CRAM_RECORD_FLAGS, HG_CRAM_RECORD_QUAL, HG_CRAM_DS_QS};
    for (size_t i = 0; i < sizeof v / sizeof v[0]; i++) o[i] = v[i];
    return sizeof v / sizeof v[0];
}
int cram_compress_block2(cram_fd *fd, cram_slice *s, cram_block *b, cram_metrics *metrics, int method, int level) {
    if (!b) return 0;                                    // cram_compress_slice passes the data series a slice does not have (cram_io.c:1917-1918)
    if (!fd) return -1;
    Behind &B = tl_behind;
    const bool defer = s && behind_enabled();
    if (!defer || (B.slice && B.slice != s)) behind_sync();
    const hg_cram_opts o = opts_of(fd);
    // the slice only matters for the FQZ methods: per-record quality lengths and flags, gathered as cram_io.c:1808-1820 does
    std::unique_ptr<FqzKeep> keep;
    const hg_fqz_slice *fqp = nullptr;
    const unsigned fqz_bits = 1u << FQZ | 1u << FQZ_b | 1u << FQZ_c | 1u << FQZ_d;
    if (s && method != -1 && ((unsigned)method & fqz_bits)) {
        const char *hdr = field<const char *>(s, HG_CRAM_SLICE_HDR);
        const char *crecs = field<const char *>(s, HG_CRAM_SLICE_CRECS);
        cram_block *const *blocks = field<cram_block *const *>(s, HG_CRAM_SLICE_BLOCK);
        const int32_t n = hdr ? field<int32_t>(hdr, HG_CRAM_SLICE_HDR_NUM_RECORDS) : 0;
        if (hdr && crecs && blocks && blocks[HG_CRAM_DS_QS] && n > 0) {
            keep.reset(new FqzKeep());
            keep->len.resize((size_t)n); keep->flags.resize((size_t)n);
            for (int32_t i = 0; i < n; i++) {
                const char *r = crecs + (size_t)i * HG_CRAM_RECORD_SIZE;
                keep->flags[(size_t)i] = (uint32_t)field<int32_t>(r, HG_CRAM_RECORD_FLAGS);
                const int32_t q = field<int32_t>(r, HG_CRAM_RECORD_QUAL);
                keep->len[(size_t)i] = (uint32_t)(i + 1 < n ? field<int32_t>(r + HG_CRAM_RECORD_SIZE, HG_CRAM_RECORD_QUAL) - q : blocks[HG_CRAM_DS_QS]->uncomp_size - q);
            }
            keep->fq.num_records = (uint32_t)n; keep->fq.len = keep->len.data(); keep->fq.flags = keep->flags.data(); fqp = &keep->fq;
        }
    }
    if (!defer) return hg_cram_compress_block_fqz(&o, fqp, b, metrics, method, level);
    if (b->method != RAW) return 0;                                           // cram_io.c:1945-1952
    B.slice = s;
    B.opts.push_back(o);
    if (keep) B.fqz.push_back(std::move(keep));
    CompJob j{&B.opts.back(), b, metrics, method, level, -1, fqp};
    bool again = false;
    for (const CompJob &x : B.first) if (x.b == b) { again = true; break; }
    (again ? B.sweep : B.first).push_back(j);
    return 0;
}
int cram_compress_block(cram_fd *fd, cram_block *b, cram_metrics *metrics, int method, int level) { return cram_compress_block2(fd, nullptr, b, metrics, method, level); }
cram_block *cram_read_block(cram_fd *fd) {
    if (!fd) return nullptr;
    cram_block *b = hg_cram_read_block(field<hFILE *>(fd, HG_CRAM_FD_FP), field<int>(fd, HG_CRAM_FD_VERSION) >> 8, field<int>(fd, HG_CRAM_FD_IGNORE_MD5));
    ahead_submit(b);
    return b;
}
int cram_write_block(cram_fd *fd, cram_block *b) {
    behind_sync();
    if (!fd || !b) return -1;
    return hg_cram_write_block(field<hFILE *>(fd, HG_CRAM_FD_FP), field<int>(fd, HG_CRAM_FD_VERSION) >> 8, b);
}

uint32_t cram_block_size(cram_block *b) {
    behind_sync();
    uint8_t tmp[32];
    size_t n = 2;
    n += (size_t)itf8_put(tmp, b->content_id) + (size_t)itf8_put(tmp, b->comp_size) + (size_t)itf8_put(tmp, b->uncomp_size);
    return (uint32_t)(n + 4 + (size_t)(b->method == RAW ? b->uncomp_size : b->comp_size));
}

// ---- the public accessors of struct cram_block (reference cram/cram_external.c:522-555; htslib.map:313-328,617): how tools outside libhts reach a block.
// "offset" / "size" is the fill level b->byte; growth follows block_resize (cram/cram_io.h:225-238): at least +25 % + 1000 bytes at a time.
int32_t cram_block_get_content_id(cram_block *b) { return b->content_type == CORE ? -1 : b->content_id; }
int32_t cram_block_get_comp_size(cram_block *b) { return b->comp_size; }
int32_t cram_block_get_uncomp_size(cram_block *b) { return b->uncomp_size; }
int32_t cram_block_get_crc32(cram_block *b) { return (int32_t)b->crc32; }
void *cram_block_get_data(cram_block *b) { return b->data; }
int32_t cram_block_get_size(cram_block *b) { return (int32_t)b->byte; }
enum cram_block_method cram_block_get_method(cram_block *b) { return (enum cram_block_method)b->orig_method; }
enum cram_content_type cram_block_get_content_type(cram_block *b) { return b->content_type; }
void cram_block_set_content_id(cram_block *b, int32_t id) { b->content_id = id; }
void cram_block_set_comp_size(cram_block *b, int32_t size) { b->comp_size = size; }
void cram_block_set_uncomp_size(cram_block *b, int32_t size) { b->uncomp_size = size; }
void cram_block_set_crc32(cram_block *b, int32_t crc) { b->crc32 = (uint32_t)crc; }
void cram_block_set_data(cram_block *b, void *data) { b->data = (unsigned char *)data; }
void cram_block_set_size(cram_block *b, int32_t size) { b->byte = (size_t)size; }
size_t cram_block_get_offset(cram_block *b) { return b->byte; }
void cram_block_set_offset(cram_block *b, size_t offset) { b->byte = offset; }
void cram_block_update_size(cram_block *b) { b->comp_size = b->uncomp_size = (int32_t)b->byte; }
int cram_block_append(cram_block *b, const void *data, int size) {
    if (size < 0) return -1;
    const size_t need = b->byte + (size_t)size;
    if (b->alloc <= need) {
        size_t alloc = b->alloc + 800;
        alloc += alloc >> 2;
        if (alloc < need) alloc = need;
        unsigned char *t = (unsigned char *)realloc(b->data, alloc ? alloc : 1);
        if (!t) return -1;
        b->data = t; b->alloc = alloc;
    }
    if (size) { memcpy(b->data + b->byte, data, (size_t)size); b->byte += (size_t)size; }
    return 0;
}

cram_block *hg_cram_read_block(hFILE *fp, int major, int ignore_crc) {
    cram_block *b = (cram_block *)calloc(1, sizeof(cram_block));
    if (!b) return nullptr;
    uint8_t hdr[40]; size_t hl = 0;
    auto byte = [&]() -> int { const int c = fp->end > fp->begin ? (unsigned char)*fp->begin++ : hgetc2(fp); if (c >= 0) hdr[hl++] = (uint8_t)c; return c; };
    int c = byte();
    if (c < 0) { free(b); return nullptr; }
    if (c > TOK3) { logerr("cram_read_block", "Unknown block compression method %d", c); free(b); return nullptr; }
    b->method = (enum cram_block_method_int)c;
    if ((c = byte()) < 0) { free(b); return nullptr; }
    b->content_type = (enum cram_content_type)c;
    if (varint_get(fp, major, &b->content_id, hdr, &hl) || varint_get(fp, major, &b->comp_size, hdr, &hl) ||
        varint_get(fp, major, &b->uncomp_size, hdr, &hl)) { free(b); return nullptr; }
    if (b->comp_size < 0 || b->uncomp_size < 0 || (b->method == RAW && b->comp_size != b->uncomp_size)) { free(b); return nullptr; }
    const size_t payload = (size_t)(b->method == RAW ? b->uncomp_size : b->comp_size);
    b->alloc = payload;
    if (!(b->data = (unsigned char *)malloc(payload ? payload : 1)) || hg_hread(fp, b->data, payload) != (ssize_t)payload) { cram_free_block(b); return nullptr; }
    if (major >= 3) {
        uint8_t t[4];
        if (hg_hread(fp, t, 4) != 4) { cram_free_block(b); return nullptr; }
        b->crc32 = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        b->crc32_checked = ignore_crc ? 1 : 0;
        b->crc_part = crc_small(0, hdr, hl);
    } else b->crc32_checked = 1;                                              // no CRC before v3
    b->orig_method = b->method;
    b->bit = 7;
    return b;
}

int hg_cram_write_block(hFILE *fp, int major, cram_block *b) {
    uint8_t hdr[40]; size_t hl = 0;
    if (b->method == RAW && b->comp_size != b->uncomp_size) return -1;
    hdr[hl++] = (uint8_t)b->method; hdr[hl++] = (uint8_t)b->content_type;
    hl += (size_t)varint_put(hdr + hl, major, b->content_id);
    hl += (size_t)varint_put(hdr + hl, major, b->comp_size);
    hl += (size_t)varint_put(hdr + hl, major, b->uncomp_size);
    if (hg_hwrite(fp, hdr, hl) != (ssize_t)hl) return -1;
    const size_t payload = b->data ? (size_t)(b->method == RAW ? b->uncomp_size : b->comp_size) : 0;
    if (payload && hg_hwrite(fp, b->data, payload) != (ssize_t)payload) return -1;
    if (major >= 3) {
        uint32_t crc = crc_small(0, hdr, hl);
        if (payload) {
            hg_ctx *ctx = engine();
            uint32_t pc = 0;
            if (!ctx || hg_crc32_host(ctx, b->data, payload, &pc) != HG_OK) { logerr("cram_write_block", "no usable GPU engine"); return -1; }
            crc = crc_concat(crc, pc, payload);
        }
        b->crc32 = crc;
        const uint8_t t[4] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24)};
        if (hg_hwrite(fp, t, 4) != 4) return -1;
    }
    return 0;
}

// ---- htscodecs' pack.h / rle.h functions that cram_codecs.c calls for CRAM 4.0 E_XPACK / E_XRLE, under the
// reference's own names and signatures so that cram_codecs.c links against them unchanged
// (cram/cram_codecs.c:1399, 1520, 2106, 2278).  var_put_u64 / var_get_u64 (:2276, :2103) are static inlines of
// htscodecs/varint.h, restated in hts_cram_gpu.h.
uint8_t *hts_pack(uint8_t *data, int64_t len, uint8_t *out_meta, int *out_meta_len, uint64_t *out_len) {
    hg_ctx *ctx = engine();
    if (!ctx) { logerr("hts_pack", "no usable GPU engine"); return nullptr; }
    return hg_hts_pack(ctx, data, len, out_meta, out_meta_len, out_len);
}
uint8_t *hts_unpack(uint8_t *data, int64_t len, uint8_t *out, uint64_t out_len, int nsym, uint8_t *p) {
    hg_ctx *ctx = engine();
    if (!ctx) { logerr("hts_unpack", "no usable GPU engine"); return nullptr; }
    return hg_hts_unpack(ctx, data, len, out, out_len, nsym, p);
}
uint8_t *hts_rle_encode(uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms, int *rle_nsyms,
                        uint8_t *out, uint64_t *out_len) {
    hg_ctx *ctx = engine();
    if (!ctx) { logerr("hts_rle_encode", "no usable GPU engine"); return nullptr; }
    return hg_hts_rle_encode(ctx, data, data_len, run, run_len, rle_syms, rle_nsyms, out, out_len);
}
uint8_t *hts_rle_decode(uint8_t *lit, uint64_t lit_len, uint8_t *run, uint64_t run_len, uint8_t *rle_syms, int rle_nsyms,
                        uint8_t *out, uint64_t *out_len) {
    hg_ctx *ctx = engine();
    if (!ctx) { logerr("hts_rle_decode", "no usable GPU engine"); return nullptr; }
    if (rle_nsyms < 0) return nullptr;
    return hg_hts_rle_decode(ctx, lit, lit_len, run, run_len, rle_syms, (uint32_t)rle_nsyms, out, out_len);
}

}  // extern "C"

// hg_stage.hip -- batched host<->device staging for the host-buffer entry points.
//
// The reference hands the codecs one malloc'd buffer per block (cram_block.data, cram/cram_structs.h:312-332).  A
// batch of CRAM blocks is therefore hundreds to thousands of small scattered host buffers; one hipMemcpy per buffer
// costs ~3 us each and used to dominate the host entry points (893 k copies in profiles/r01 codec probe).  Here the
// pieces are packed into ONE pinned staging buffer and moved with ONE transfer each way; on the way back a gather
// kernel first compacts the pieces (they sit in worst-case-sized slots) so that only payload crosses PCIe.
#include <hip/hip_runtime.h>
#include <utility>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <stdlib.h>
#include <thread>
#include <vector>
#include "htsgpu.h"
#include "hg_internal.h"

namespace hgs {

struct Piece { uint64_t src, dst; uint32_t len, pad; };
constexpr uint32_t CHUNK = 1u << 16;                                   // pieces are cut into <= 64 KiB work items

__global__ __launch_bounds__(256)
void gather_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const Piece *__restrict__ pc, uint32_t n) {
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const Piece p = pc[k];
        const uint8_t *s = src + p.src; uint8_t *d = dst + p.dst;
        // dword path when both ends are aligned the same way
        const uint32_t head = (uint32_t)((4u - ((uintptr_t)s & 3u)) & 3u);
        if ((((uintptr_t)s ^ (uintptr_t)d) & 3u) == 0 && p.len >= 64u) {
            if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
            const uint32_t nw = (p.len - head) >> 2;
            const uint32_t *s4 = (const uint32_t *)(s + head); uint32_t *d4 = (uint32_t *)(d + head);
            for (uint32_t i = threadIdx.x; i < nw; i += 256) d4[i] = s4[i];
            for (uint32_t i = head + (nw << 2) + threadIdx.x; i < p.len; i += 256) d[i] = s[i];
        } else for (uint32_t i = threadIdx.x; i < p.len; i += 256) d[i] = s[i];
    }
}

// ---- the host side of staging: hundreds of MB are memcpy'd between the callers' malloc'd blocks and the pinned buffer per batch;
// one thread does that at ~10 GB/s, which made the copies -- not the kernels, not PCIe -- the longest part of a CRAM batch.  A few
// process-wide helper threads share them (HTS_GPU_COPY_THREADS helpers, default 3, 0 = none); callers on different contexts may
// submit concurrently.
struct CopyJob { uint8_t *d; const uint8_t *s; size_t n; };
class CopyPool {
    struct Batch { std::atomic<size_t> left{0}; std::mutex m; std::condition_variable cv; };
    struct Task { const CopyJob *j; size_t n; Batch *b; };
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Task> q;
    bool stop = false;
    static void run(const Task &t) {
        for (size_t i = 0; i < t.n; i++) memcpy(t.j[i].d, t.j[i].s, t.j[i].n);
        if (t.b->left.fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(t.b->m); t.b->cv.notify_all(); }
    }
    void worker() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;
            const Task t = q.front(); q.pop_front();
            lk.unlock(); run(t); lk.lock();
        }
    }
public:
    explicit CopyPool(int n) { for (int i = 0; i < n; i++) th.emplace_back([this] { worker(); }); }
    ~CopyPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
    // jobs are cut into helpers + 1 runs of about equal bytes; the caller takes the first run itself
    void copy(const std::vector<CopyJob> &jobs) {
        size_t total = 0;
        for (const CopyJob &j : jobs) total += j.n;
        const size_t parts = th.size() + 1, target = total / parts + 1;
        std::vector<std::pair<size_t, size_t>> runs;                       // [first, count)
        size_t first = 0, acc = 0;
        for (size_t i = 0; i < jobs.size(); i++) {
            acc += jobs[i].n;
            if (acc >= target && runs.size() + 1 < parts) { runs.push_back({first, i + 1 - first}); first = i + 1; acc = 0; }
        }
        if (first < jobs.size()) runs.push_back({first, jobs.size() - first});
        if (runs.empty()) return;
        Batch b;
        b.left = runs.size();
        {
            std::lock_guard<std::mutex> lk(m);
            for (size_t r = 1; r < runs.size(); r++) q.push_back(Task{jobs.data() + runs[r].first, runs[r].second, &b});
        }
        cv.notify_all();
        run(Task{jobs.data() + runs[0].first, runs[0].second, &b});
        std::unique_lock<std::mutex> lk(b.m);
        b.cv.wait(lk, [&] { return b.left.load() == 0; });
    }
};
static void host_copies(const std::vector<CopyJob> &jobs) {
    static CopyPool *pool = [] {
        const char *v = getenv("HTS_GPU_COPY_THREADS");
        const int n = v ? atoi(v) : 3;
        return n > 0 ? new CopyPool(n > 16 ? 16 : n) : nullptr;
    }();
    size_t total = 0;
    for (const CopyJob &j : jobs) total += j.n;
    if (pool && total >= (4u << 20)) pool->copy(jobs);
    else for (const CopyJob &j : jobs) memcpy(j.d, j.s, j.n);
}

static int ensure_pinned(hg_ctx *ctx, int which, size_t bytes) {
    if (ctx->h_stage_cap[which] >= bytes) return HG_OK;
    if (ctx->h_stage[which]) (void)hipHostFree(ctx->h_stage[which]);
    ctx->h_stage[which] = nullptr; ctx->h_stage_cap[which] = 0;
    const size_t cap = bytes + bytes / 4 + (1u << 20);
    if (hipHostMalloc(&ctx->h_stage[which], cap, hipHostMallocDefault) != hipSuccess) return HG_ENOMEM;
    ctx->h_stage_cap[which] = cap;
    return HG_OK;
}

}  // namespace hgs

namespace hg {

// Host buffers src[i] (len[i] bytes) -> d_base + dst_off[i].  The device layout [0, total) is mirrored in a pinned
// buffer and sent in one transfer; src[i] may be null when len[i] is 0.  skip[i] != 0 leaves a piece out.
int stage_upload(hg_ctx *ctx, const uint8_t *const *src, const uint32_t *len, const uint64_t *dst_off, const int32_t *skip, size_t n,
                 uint64_t total, uint8_t *d_base, hipStream_t s) {
    if (!n || !total) return HG_OK;
    // the previous user of this pinned buffer may still be in flight on the stream
    if (hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    uint64_t hi = 0, lo = ~0ull;                             // only the stretch that holds pieces is mirrored and travels (skipped pieces: blocks that are already on the device)
    for (size_t i = 0; i < n; i++) {
        if (!len[i] || (skip && skip[i])) continue;
        if (dst_off[i] + len[i] > hi) hi = dst_off[i] + len[i];
        if (dst_off[i] < lo) lo = dst_off[i];
    }
    if (!hi) return HG_OK;
    if (hi > total) return HG_EINVAL;
    if (int rc = hgs::ensure_pinned(ctx, 0, hi - lo)) return rc;
    uint8_t *h = (uint8_t *)ctx->h_stage[0];
    std::vector<hgs::CopyJob> jobs;
    jobs.reserve(n);
    for (size_t i = 0; i < n; i++) if (len[i] && !(skip && skip[i])) jobs.push_back({h + (dst_off[i] - lo), src[i], len[i]});
    hgs::host_copies(jobs);
    return hipMemcpyAsync(d_base + lo, h, hi - lo, hipMemcpyHostToDevice, s) == hipSuccess ? HG_OK : HG_ELAUNCH;
}

// Device pieces d_base + src_off[i] (len[i] bytes) -> host buffers dst[i].  Synchronises the stream.
int stage_download(hg_ctx *ctx, const uint8_t *d_base, const uint64_t *src_off, const uint32_t *len, uint8_t *const *dst, size_t n,
                   hipStream_t s) {
    std::vector<hgs::Piece> pc;
    std::vector<uint64_t> hoff(n);
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) {
        hoff[i] = total;
        for (uint32_t o = 0; o < len[i]; o += hgs::CHUNK) {
            const uint32_t l = len[i] - o < hgs::CHUNK ? len[i] - o : hgs::CHUNK;
            pc.push_back({src_off[i] + o, total + o, l, 0});
        }
        total += ((uint64_t)len[i] + 3u) & ~3ull;
    }
    if (!total) return hipStreamSynchronize(s) == hipSuccess ? HG_OK : HG_ELAUNCH;
    int rc;
    if ((rc = ensure_scratch(ctx, 13, total + 64)) || (rc = ensure_scratch(ctx, 14, pc.size() * sizeof(hgs::Piece) + 64)) ||
        (rc = hgs::ensure_pinned(ctx, 1, total))) return rc;
    if (hipMemcpyAsync(ctx->d_scratch[14], pc.data(), pc.size() * sizeof(hgs::Piece), hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    size_t wgs = pc.size();
    const size_t maxw = (size_t)ctx->cus * 16;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgs::gather_kernel, dim3((unsigned)wgs), dim3(256), 0, s, d_base, (uint8_t *)ctx->d_scratch[13],
                       (const hgs::Piece *)ctx->d_scratch[14], (uint32_t)pc.size());
    if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    const uint8_t *h = (const uint8_t *)ctx->h_stage[1];
    // A large image comes over in a few transfers, and the host copies of one part (pinned buffer -> the callers' blocks) run while the next part is on
    // the wire: 786 MB of decoded CRAM blocks were 14 ms of PCIe and then 10 ms of memcpy, one after the other.
    constexpr uint64_t PART = 64ull << 20;
    constexpr size_t MAX_PARTS = 16;
    struct Part { uint64_t a, b; size_t i0, i1; };
    std::vector<Part> parts;
    if (total >= 2 * PART) {
        size_t i0 = 0; uint64_t a = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t end = i + 1 < n ? hoff[i + 1] : total;
            if (end - a >= PART && parts.size() + 1 < MAX_PARTS && i + 1 < n) { parts.push_back({a, end, i0, i + 1}); a = end; i0 = i + 1; }
        }
        parts.push_back({a, total, i0, n});
    } else parts.push_back({0, total, 0, n});
    static thread_local std::vector<std::pair<int, hipEvent_t>> ev_cache;     // (device, event): the waits below are on this thread
    std::vector<hipEvent_t> ev(parts.size(), nullptr);
    if (parts.size() > 1) {
        size_t got = 0;
        for (auto &e : ev_cache) if (e.first == ctx->device && got < ev.size()) ev[got++] = e.second;
        for (; got < ev.size(); got++) {
            if (hipEventCreateWithFlags(&ev[got], hipEventDisableTiming) != hipSuccess) return HG_ELAUNCH;
            ev_cache.push_back({ctx->device, ev[got]});
        }
    }
    for (size_t k = 0; k < parts.size(); k++) {
        if (hipMemcpyAsync(ctx->h_stage[1] ? (uint8_t *)ctx->h_stage[1] + parts[k].a : nullptr, (const uint8_t *)ctx->d_scratch[13] + parts[k].a, parts[k].b - parts[k].a,
                           hipMemcpyDeviceToHost, s) != hipSuccess) return HG_ELAUNCH;
        if (parts.size() > 1 && hipEventRecord(ev[k], s) != hipSuccess) return HG_ELAUNCH;
    }
    std::vector<hgs::CopyJob> jobs;
    for (size_t k = 0; k < parts.size(); k++) {
        if ((parts.size() > 1 ? hipEventSynchronize(ev[k]) : hipStreamSynchronize(s)) != hipSuccess) return HG_ELAUNCH;
        jobs.clear();
        for (size_t i = parts[k].i0; i < parts[k].i1; i++) if (len[i]) jobs.push_back({dst[i], h + hoff[i], len[i]});
        hgs::host_copies(jobs);
    }
    return HG_OK;
}

// Device-to-device form of the same gather: d_src + src_off[i] -> d_dst + dst_off[i], one launch.
int stage_gather_dev(hg_ctx *ctx, const uint8_t *d_src, const uint64_t *src_off, const uint32_t *len, uint8_t *d_dst, const uint64_t *dst_off,
                     size_t n, hipStream_t s) {
    std::vector<hgs::Piece> pc;
    for (size_t i = 0; i < n; i++)
        for (uint32_t o = 0; o < len[i]; o += hgs::CHUNK) pc.push_back({src_off[i] + o, dst_off[i] + o, len[i] - o < hgs::CHUNK ? len[i] - o : hgs::CHUNK, 0});
    if (pc.empty()) return HG_OK;
    if (int rc = ensure_scratch(ctx, 15, pc.size() * sizeof(hgs::Piece) + 64)) return rc;
    if (hipMemcpyAsync(ctx->d_scratch[15], pc.data(), pc.size() * sizeof(hgs::Piece), hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    size_t wgs = pc.size();
    const size_t maxw = (size_t)ctx->cus * 16;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgs::gather_kernel, dim3((unsigned)wgs), dim3(256), 0, s, d_src, d_dst, (const hgs::Piece *)ctx->d_scratch[15], (uint32_t)pc.size());
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

void stage_free(hg_ctx *ctx) {
    for (int k = 0; k < 2; k++) if (ctx->h_stage[k]) { (void)hipHostFree(ctx->h_stage[k]); ctx->h_stage[k] = nullptr; ctx->h_stage_cap[k] = 0; }
}

}  // namespace hg

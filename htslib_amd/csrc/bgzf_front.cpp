// bgzf_front.cpp -- htslib's BGZF front-end (bgzf_open / bgzf_read / bgzf_write / ...) written
// from scratch on top of the gfx950 batch engine.  Host C++ only; no codec arithmetic lives here.
//
// Behavioural contract = the reference's (bgzf.c; checklist in SURVEY.md Appendix C):
//   read state machine       bgzf.c:1004-1291   (block_length==0 <=> no block loaded, tell never
//                                                points at the end of a block, empty blocks skipped,
//                                                trailing empty block = EOF marker)
//   write state machine      bgzf.c:1927-2124   (blocks cut at 0xff00, bgzf_flush_try keeps records
//                                                whole, close appends the 28-byte EOF block)
//   seek / virtual offsets   bgzf.c:2175-2258   htslib/bgzf.h:261
//   .gzi index               bgzf.c:2336-2542   deferred index offsets bgzf.c:189-290
//   error bits               htslib/bgzf.h:53-58
//
// Engine.  Every compressed handle is the reference's *threaded* mode re-thought for a device that wants
// thousands of blocks per launch:
//   reader : an I/O thread hread()s a window of the file into a pipe's pinned buffer, frames the blocks and
//            submits ONE inflate job (H2D, kernel, D2H on the pipe's stream); three pipes rotate, so reading
//            window n+2, moving / inflating window n+1 and consuming window n overlap.  The consumer never
//            copies a block: fp->uncompressed_block points into the pipe's pinned plain image (the
//            reference steals the job's buffer the same way, bgzf.c:1077-1089), and because that image is
//            contiguous bgzf_read / bgzf_getline work on SPANS of blocks, not block by block.
//            The window starts small after open / seek (random access stays cheap) and grows 4x per batch.
//   writer : bgzf_write cuts blocks straight into a pipe's pinned input buffer; a full buffer is one
//            deflate job; an output thread collects jobs in order, resolves deferred index entries with
//            the now-known block addresses and hwrite()s each batch with one call.
//   gzip   : plain gzip streams (read) and mode "g" (write) use the engine's single-wavefront stream codec.
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "hts_bgzf_gpu.h"
#include "hts_hfile_abi.h"
#include "htsgpu.h"
#include "bgzf_host_codec.h"
// The host codec's state is ~450 KiB (Deflater) / ~32 KiB (Inflater): as plain `thread_local` objects that much static TLS would be reserved in EVERY thread
// of the host process, whether it ever touches a BGZF handle or not.  One heap object per thread that actually uses the codec instead.
static hgh::Deflater &tls_deflater() { static thread_local std::unique_ptr<hgh::Deflater> p; if (!p) p.reset(new hgh::Deflater); return *p; }
static hgh::Inflater &tls_inflater() { static thread_local std::unique_ptr<hgh::Inflater> p; if (!p) p.reset(new hgh::Inflater); return *p; }

// libhts symbols used when present (inside a libhts build they always are)
extern "C" {
void hts_log(int severity, const char *context, const char *format, ...) __attribute__((weak));
int hts_idx_push(void *idx, int tid, int64_t beg, int64_t end, uint64_t offset, int is_mapped) __attribute__((weak));
int hg_crc32_host(hg_ctx *ctx, const void *buf, size_t len, uint32_t *crc);
}

struct GziEntry { uint64_t caddr, uaddr; };
struct bgzidx_t {                      // behind fp->idx; opaque to callers (bgzf.c:169-174)
    std::vector<GziEntry> offs;        // offs[0] = {0, 0}
    uint64_t ublock_addr = 0;          // uncompressed offset of the block being read
    size_t recut = 0;                  // bgzf_block_write: which indexed block is being filled
    bool loaded = false;
};

namespace {

constexpr int NPIPES_MIN = 4, NPIPES_MAX = 16;     // pipes per handle: 4 on one device (one being consumed, three in flight: a job is ~2 ms of kernel
                                                   // latency + 0.8 ms of D2H for a 0.7 ms slot), 2 per device on several (HTS_GPU_DEVICES; HTS_GPU_PIPES overrides)
constexpr size_t WINDOW_MIN = 256u << 10;      // compressed bytes of the first batch after open / seek
constexpr size_t WINDOW_MAX = 8u << 20;        // 8 MiB of BGZF is 25-45 MiB of plain BAM per batch: the pinned buffers of a handle (3 pipes) stay
                                               // under ~200 MiB -- pinning costs ~0.3 ms per MiB at open and as much again at close (32 MiB
                                               // windows: 270 ms before the first GiB arrived, 170 ms to close; steady state is the same)
constexpr uint64_t PLAIN_MAX = 256ull << 20;   // plain bytes per batch (highly compressible input)
constexpr size_t WBLOCKS_MIN = 16, WBLOCKS_MAX = 768;    // = the deflate kernel's resident workgroups on a 256-CU device (3 per CU): a launch lasts as long as
                                                         // its slowest block whatever their number (trace: 1.68 ms for 512 blocks, back to back, the chip a third empty)
const uint8_t kEof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0,
                          0, 0, 0, 0, 0, 0, 0, 0};
enum { LOG_ERROR = 1, LOG_WARNING = 3 };

void logmsg(int level, const char *ctx, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (hts_log) hts_log(level, ctx, "%s", buf);
    else fprintf(stderr, "[%c::%s] %s\n", level == LOG_ERROR ? 'E' : 'W', ctx, buf);
}

uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
uint32_t crc_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {           // crc(A||B)
    uint32_t xp = 0x00800000u, acc = 0x80000000u;
    for (; len_b; len_b >>= 1) { if (len_b & 1) acc = crc_mulmod(acc, xp); xp = crc_mulmod(xp, xp); }
    return crc_mulmod(acc, crc_a) ^ crc_b;
}

bool bgzf_header_ok(const uint8_t *h) {   // check_header, bgzf.c:896-903
    return h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' &&
           h[13] == 'C' && h[14] == 2 && h[15] == 0;
}

int device_choice() { const char *d = getenv("HTS_GPU_DEVICE"); return d ? atoi(d) : 0; }
// HTS_GPU_DEVICES = "0-7" / "0,2,5" / "4": the devices a handle spreads its batches over (north_star: "independent blocks are sharded across the GPUs of
// one node with a trivial static split").  Each batch (window of whole blocks) is one job on one device's pipe; the pipes rotate, so consecutive windows go
// to consecutive devices and come back in submission order -- no collective, no cross-device traffic.  Unset: the one device of HTS_GPU_DEVICE.
std::vector<int> device_list() {
    std::vector<int> v;
    const char *d = getenv("HTS_GPU_DEVICES");
    if (d) {
        for (const char *p = d; *p;) {
            if (*p < '0' || *p > '9') { p++; continue; }
            char *q; const long a = strtol(p, &q, 10); long b = a;
            if (*q == '-') { b = strtol(q + 1, &q, 10); }
            for (long x = a; x <= b && v.size() < 64; x++) v.push_back((int)x);
            p = q;
        }
    }
    if (v.empty()) v.push_back(device_choice());
    return v;
}

// process-wide context for the stateless entry points (bgzf_compress, hts_crc32); its host calls lock it
hg_ctx *shared_ctx() {
    static hg_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] { if (hg_init(device_choice(), &ctx) != HG_OK) ctx = nullptr; });
    return ctx;
}

struct ReadBatch {
    hg_pipe *pipe = nullptr;
    std::vector<hg_bgzf_desc> desc;
    int64_t file_off = 0;          // file offset of desc[0]
    size_t comp_len = 0;           // bytes covered by desc
    int fail = 0;                  // BGZF_ERR_* met while reading / framing BEHIND the last good block
    bool submitted = false;
    bool reserved = false;         // the pipe's buffers have been sized for full windows (fill_batch)
    // after hg_pipe_wait:
    const uint8_t *plain = nullptr;
    const int32_t *status = nullptr;
    size_t good = 0;               // leading blocks with status 0
    // a HOST batch (the first blocks after bgzf_open / bgzf_seek, bgzf_host_codec.h): not submitted to the device; a block is decoded on the consumer's
    // thread when the consumer steps onto it (a region query that reads 100 bytes pays for one block, not for a device round trip)
    bool host = false;
    const uint8_t *hcomp = nullptr;    // the framed blocks (the pipe's pinned input buffer)
    std::vector<uint8_t> hplain;       // their plain image, filled block by block
    std::vector<int32_t> hstatus;      // HOST_PENDING until decoded, then 0 / -1 / -2
};
constexpr int32_t HOST_PENDING = 1;
constexpr size_t HOST_FIRST_BLOCKS = 4;

struct IdxPush { void *hidx; int tid; int64_t beg, end; uint32_t offset; int mapped; uint64_t block_number; };

struct WriteBatch {
    hg_pipe *pipe = nullptr;
    uint8_t *in = nullptr;         // the pipe's pinned input buffer
    size_t cap = 0, len = 0, target = 0;
    std::vector<uint64_t> cuts;
};

enum Kind { K_READ, K_WRITE, K_GZREAD, K_GZWRITE };

// ---- contexts and pipes outlive their handle.  Opening the HIP side of a handle costs ~40 ms and pinning its windows ~0.3 ms per MiB (a streaming
// reader: ~250 MiB); a program that opens one BGZF file after another (merge, cat, a region loop over many files) would pay that every time, and again
// to unpin at close.  A closed handle therefore parks its device context with the pipes (and their buffers) it grew, and the next handle on that
// device takes them over as they are.  At most two parked contexts per device (a reader and a writer); the rest is freed.  They live until exit.
// HTS_GPU_KEEP=0 turns parking off.
struct Parked { int dev; hg_ctx *ctx; std::vector<hg_pipe *> pipes; };
class EnginePark {
    std::mutex m;
    std::vector<Parked> v;
public:
    bool take(int dev, hg_ctx *&ctx, std::vector<hg_pipe *> &pipes) {
        std::lock_guard<std::mutex> lk(m);
        for (size_t i = v.size(); i-- > 0;)
            if (v[i].dev == dev) { ctx = v[i].ctx; pipes.swap(v[i].pipes); v.erase(v.begin() + (long)i); return true; }
        return false;
    }
    void give(int dev, hg_ctx *ctx, std::vector<hg_pipe *> &pipes) {
        static const bool keep = [] { const char *k = getenv("HTS_GPU_KEEP"); return !k || atoi(k) != 0; }();
        {
            std::lock_guard<std::mutex> lk(m);
            size_t same = 0;
            for (auto &p : v) same += p.dev == dev;
            if (keep && same < 2) { v.push_back(Parked{dev, ctx, std::move(pipes)}); return; }
        }
        for (hg_pipe *p : pipes) hg_pipe_destroy(p);
        hg_destroy(ctx);
    }
};
EnginePark &engine_park() { static EnginePark *p = new EnginePark(); return *p; }

// HTS_GPU_STATS=1: where a reader's wall time went, printed to stderr when the handle closes (seconds, per thread)
struct ReadStats { double io_read = 0, io_frame = 0, io_submit = 0, io_idle = 0, c_wait = 0, c_copy = 0; uint64_t batches = 0, copies = 0; };
// ... and a writer's: the caller's side (waiting for a free pipe, copying into the batch, submitting) and the output thread's (waiting for the device, hwrite)
struct WriteStats { double c_pipe = 0, c_copy = 0, c_submit = 0, o_wait = 0, o_write = 0; uint64_t batches = 0, blocks = 0, out_bytes = 0; };
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline bool stats_on() { static const bool on = getenv("HTS_GPU_STATS") != nullptr; return on; }

struct Engine {
    BGZF *fp = nullptr;
    ReadStats st;
    WriteStats wst;
    Kind kind = K_READ;
    hg_ctx *gpu = nullptr;                 // first device (stateless helpers: gzip streams, CRCs)
    std::vector<hg_ctx *> devs;            // every device of the handle; pipe i lives on devs[i % devs.size()]
    std::vector<int> dev_ids;              // their ordinals
    std::vector<std::vector<hg_pipe *>> spare;   // per device: pipes taken over from a closed handle, not yet in use
    int NPIPES = NPIPES_MIN;
    void *own_block = nullptr;     // the malloc'd 128 KiB block (fp->uncompressed_block is re-pointed by readers)
    std::thread th;
    bool started = false;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    // ---------------- reader
    ReadBatch rb[NPIPES_MAX];
    uint64_t fill_seq = 0, done_seq = 0;   // batches filled by the I/O thread / released by the consumer
    bool cur_loaded = false;               // rb[done_seq % e->NPIPES] has been waited for and is being consumed
    bool input_done = false;               // the I/O thread met EOF or a fatal input error
    bool pause_req = false, paused = false;
    size_t window = WINDOW_MIN;
    double ratio_seen = 0;                 // best plain / compressed ratio of a batch so far (sizes the pinned output buffers)
    size_t ahead = NPIPES_MIN;             // batches the I/O thread may run ahead: 1 after a seek until the consumer shows
                                           // that it is scanning (a random-access caller reads a few bytes and seeks again)
    std::vector<uint8_t> carry;            // bytes of a block cut by the window end
    int64_t read_off = 0;                  // file offset of carry[0] / of the next byte to hread
    int pfd = -1;                          // regular local file: a private descriptor for positional reads of the windows by several threads
    size_t blk = 0;                        // current block inside the current batch
    bool blk_pending = false;              // a seek positioned us BEFORE rb.desc[blk]
    int64_t next_addr = 0;                 // file offset of the block after the current one
    bool eof_seen = false;
    // ---------------- writer
    WriteBatch wb[NPIPES_MAX];
    uint64_t w_fill = 0, w_done = 0;       // batches submitted / written out
    bool w_open = false;                   // wb[w_fill % e->NPIPES] is being filled
    size_t w_target = WBLOCKS_MIN;
    int64_t block_address = 0;             // compressed bytes written so far (owned by the output thread)
    int w_err = 0;                         // BGZF_ERR_* raised by the output thread
    uint64_t block_number = 0;             // blocks queued so far (bgzf.c:1856)
    uint64_t block_written = 0;
    std::mutex idx_m;
    std::vector<IdxPush> pushes;
    // gzip member being written (mode "g")
    bool gz_header_done = false;
    uint32_t gz_crc = 0; uint64_t gz_isize = 0;
    // ---------------- plain gzip reader
    std::vector<uint8_t> gz_comp; int64_t gz_base = 0; bool gz_eof = false, gz_finished = false;
    hg_gz_state gz_st{};
    std::vector<uint8_t> gz_out; size_t gz_out_pos = 0;
    std::vector<uint8_t> gz_hist;
};

// The engine hangs on fp->cache (an opaque pointer the reference uses for its block cache, which has no role
// here).  fp->mt is the reference's "threaded semantics" marker that callers test (sam.c:942, vcf.c:4596): set for
// every compressed reader, and for a writer once bgzf_mt() / bgzf_thread_pool() has been called -- see queue_block().
inline Engine *E(BGZF *fp) { return reinterpret_cast<Engine *>(fp->cache); }

// ---- window reads.  One thread's read(2) out of the page cache moves 4-6 GB/s, i.e. ~25 GB/s of plain BAM -- less than the pipeline
// behind it takes -- so when the input is a regular local file (bgzf_open on a path, bgzf_dopen on a regular file's descriptor) a window
// is fetched with positional reads by a few threads at once (process-wide helpers, created on first use; HTS_GPU_READ_THREADS = readers
// per window, default 4, 1 = the calling thread alone).  Anything else (pipes, hFILE plugins) keeps the single hread of the reference.
class ReadPool {
    struct Job { int fd; uint8_t *d; size_t n; off_t off; ssize_t *res; };
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    std::vector<Job> q;
    size_t pending = 0;
    bool stop = false;
    static ssize_t pread_all(int fd, uint8_t *d, size_t n, off_t off) {
        size_t got = 0;
        while (got < n) {
            ssize_t r;
            do r = pread(fd, d + got, n - got, off + (off_t)got); while (r < 0 && errno == EINTR);
            if (r < 0) return got ? (ssize_t)got : -1;
            if (r == 0) break;
            got += (size_t)r;
        }
        return (ssize_t)got;
    }
    void worker() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (stop && q.empty()) return;
            const Job j = q.back(); q.pop_back();
            lk.unlock();
            *j.res = pread_all(j.fd, j.d, j.n, j.off);
            lk.lock();
            if (--pending == 0) done_cv.notify_all();
        }
    }
public:
    explicit ReadPool(int helpers) { for (int i = 0; i < helpers; i++) th.emplace_back([this] { worker(); }); }
    ~ReadPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
    std::mutex big;                                    // one window at a time per pool
    // bytes read into d (contiguous from `off`), < n only at end of file; -1 = I/O error before any byte
    ssize_t read(int fd, uint8_t *d, size_t n, off_t off) {
        constexpr size_t SLICE_MIN = 1u << 20;
        const size_t parts = th.size() + 1;
        if (n < 2 * SLICE_MIN || parts == 1) return pread_all(fd, d, n, off);
        std::lock_guard<std::mutex> one(big);
        const size_t each = (((n + parts - 1) / parts) + SLICE_MIN - 1) & ~(SLICE_MIN - 1);
        ssize_t res[17]; size_t len[17]; size_t k = 0;
        for (size_t o = 0; o < n; o += each, k++) { len[k] = n - o < each ? n - o : each; res[k] = 0; }
        {
            std::lock_guard<std::mutex> lk(m);
            for (size_t i = 1; i < k; i++) { q.push_back(Job{fd, d + i * each, len[i], off + (off_t)(i * each), &res[i]}); pending++; }
        }
        cv.notify_all();
        res[0] = pread_all(fd, d, len[0], off);
        { std::unique_lock<std::mutex> lk(m); done_cv.wait(lk, [&] { return pending == 0; }); }
        size_t total = 0;
        for (size_t i = 0; i < k; i++) {
            if (res[i] < 0) return total ? (ssize_t)total : -1;
            total += (size_t)res[i];
            if ((size_t)res[i] < len[i]) break;          // end of file inside this slice: what follows it is not contiguous
        }
        return (ssize_t)total;
    }
};
ReadPool *read_pool() {
    static ReadPool *pool = [] {
        const char *v = getenv("HTS_GPU_READ_THREADS");
        int n = v ? atoi(v) : 4;
        if (n < 1) n = 1;
        return new ReadPool((n > 16 ? 16 : n) - 1);      // lives until exit (helpers are parked on a condition variable)
    }();
    return pool;
}

// ================================================================================ reader: I/O thread
// Fill one batch: read a window, frame whole blocks, submit the inflate job.  Returns true when no further batch
// can follow (end of input or a fatal input error); the caller publishes that together with the batch.
bool fill_batch(Engine *e, ReadBatch &b, size_t host_blocks = 0) {
    bool finished = false;
    BGZF *fp = e->fp;
    b.desc.clear(); b.fail = 0; b.submitted = false; b.comp_len = 0; b.good = 0; b.host = false;
    b.file_off = e->read_off;
    const size_t want = e->window;
    // Past the first window after an open / seek the reader is scanning and the windows grow to WINDOW_MAX: size this pipe's buffers for
    // that once (input exactly, output from the best ratio seen so far) rather than re-pinning them at every step of the growth.
    if (want > WINDOW_MIN && !b.reserved) {
        const double r = e->ratio_seen > 1.0 ? e->ratio_seen * 1.15 : 4.0;
        const uint64_t out_est = (uint64_t)((double)WINDOW_MAX * r);
        (void)hg_pipe_reserve(b.pipe, WINDOW_MAX + 2 * BGZF_MAX_BLOCK_SIZE, (size_t)(out_est < PLAIN_MAX ? out_est : PLAIN_MAX));
        b.reserved = true;
    }
    uint8_t *buf = (uint8_t *)hg_pipe_input(b.pipe, e->carry.size() + want + BGZF_MAX_BLOCK_SIZE);
    if (!buf) { b.fail = BGZF_ERR_IO; return true; }
    size_t have = e->carry.size();
    if (have) memcpy(buf, e->carry.data(), have);
    e->carry.clear();
    bool at_eof = false;
    const double t_r0 = stats_on() ? now_s() : 0;
    {
        // regular local file: positional reads by several threads; the hFILE is left where the window ends, for whoever uses it next
        ssize_t n = e->pfd >= 0 ? read_pool()->read(e->pfd, buf + have, want, (off_t)(e->read_off + (int64_t)have))
                                : hg_hread(fp->fp, buf + have, want);
        if (n >= 0 && e->pfd >= 0 && hseek(fp->fp, (off_t)(e->read_off + (int64_t)have + n), SEEK_SET) < 0) n = -1;
        if (n < 0) { b.fail = BGZF_ERR_IO; return true; }
        if ((size_t)n < want) at_eof = true;
        have += (size_t)n;
    }
    const double t_r1 = stats_on() ? now_s() : 0;
    size_t pos = 0; uint64_t plain = 0;
    bool capped = false;
    while (have - pos >= 18) {
        const uint8_t *h = buf + pos;
        if (!bgzf_header_ok(h)) { b.fail = BGZF_ERR_HEADER; break; }
        const size_t bs = (size_t)(h[16] | (h[17] << 8)) + 1;
        if (bs < 26) { b.fail = BGZF_ERR_HEADER; break; }
        if (bs > have - pos) break;                                       // cut by the window end
        const uint8_t *t = h + bs - 4;
        const uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > BGZF_MAX_BLOCK_SIZE) { b.fail = BGZF_ERR_ZLIB; break; }
        if ((plain + isize > PLAIN_MAX && !b.desc.empty()) || (host_blocks && b.desc.size() >= host_blocks)) { capped = true; break; }
        hg_bgzf_desc d; d.coff = pos; d.uoff = plain; d.clen = (uint32_t)bs; d.ulen = isize;
        b.desc.push_back(d);
        plain += isize; pos += bs;
    }
    // the file ends inside a block header or payload (bgzf.c:1140-1163, 1206-1212)
    if (!b.fail && at_eof && !capped && pos < have) b.fail = have - pos >= 18 ? BGZF_ERR_IO : BGZF_ERR_HEADER;
    if (b.fail || (at_eof && !capped)) finished = true;
    else if (pos < have) e->carry.assign(buf + pos, buf + have);
    b.comp_len = pos;
    if (pos >= (64u << 10) && (double)plain / (double)pos > e->ratio_seen) e->ratio_seen = (double)plain / (double)pos;
    e->read_off += (int64_t)pos;
    const double t_r2 = stats_on() ? now_s() : 0;
    if (host_blocks) {                                                       // decoded later, block by block, on the consumer's thread
        b.host = true; b.hcomp = buf;
        b.hplain.resize((size_t)plain + 64); b.hstatus.assign(b.desc.size(), HOST_PENDING);
        if (stats_on()) { const double t = now_s(); e->st.io_read += t_r1 - t_r0; e->st.io_frame += t - t_r1; e->st.batches++; }
        return finished;
    }
    if (!b.desc.empty()) {
        if (hg_pipe_inflate(b.pipe, pos, b.desc.data(), b.desc.size()) != HG_OK) {
            b.desc.clear(); b.fail = BGZF_ERR_ZLIB; finished = true;
        } else b.submitted = true;
    }
    if (stats_on()) { const double t = now_s(); e->st.io_read += t_r1 - t_r0; e->st.io_frame += t_r2 - t_r1; e->st.io_submit += t - t_r2; e->st.batches++; }
    if (e->window < WINDOW_MAX) e->window = e->window * 4 > WINDOW_MAX ? WINDOW_MAX : e->window * 4;
    return finished;
}

void reader_main(Engine *e) {
    std::unique_lock<std::mutex> lk(e->m);
    for (;;) {
        e->cv.wait(lk, [&] { return e->stop || e->pause_req || (!e->input_done && e->fill_seq - e->done_seq < e->ahead); });
        if (e->stop) break;
        if (e->pause_req) {
            e->paused = true; e->cv.notify_all();
            e->cv.wait(lk, [&] { return !e->pause_req || e->stop; });
            e->paused = false;
            continue;
        }
        ReadBatch &b = e->rb[e->fill_seq % e->NPIPES];
        lk.unlock();
        const bool finished = fill_batch(e, b);
        lk.lock();
        if (finished) e->input_done = true;                               // published with the batch, never before it
        e->fill_seq++;
        e->cv.notify_all();
    }
}

// pipe number i of the handle: one left by a closed handle on that device, or a new one
bool get_pipe(Engine *e, int i, hg_pipe **out) {
    const size_t d = (size_t)i % e->devs.size();
    if (!e->spare[d].empty()) { *out = e->spare[d].back(); e->spare[d].pop_back(); return true; }
    return hg_pipe_create(e->devs[d], out) == HG_OK;
}

bool start_reader(Engine *e) {
    if (e->started) return true;
    for (int i = 0; i < e->NPIPES; i++) if (!e->rb[i].pipe && !get_pipe(e, i, &e->rb[i].pipe)) return false;
    e->read_off = hg_htell(e->fp->fp);
    // the first blocks of the file are framed here and decoded on the host as the consumer reaches them (first bytes without a device round trip);
    // the I/O thread takes over behind them once the consumer shows that it is scanning (widen_readahead)
    e->ahead = 1;
    if (fill_batch(e, e->rb[0], HOST_FIRST_BLOCKS)) e->input_done = true;
    e->fill_seq = 1;
    e->started = true;
    e->th = std::thread(reader_main, e);
    return true;
}

// Park the I/O thread (seek, EOF check); returns with e->m held by `lk`.
void pause_reader(Engine *e, std::unique_lock<std::mutex> &lk) {
    if (!e->started) return;
    e->pause_req = true; e->cv.notify_all();
    e->cv.wait(lk, [&] { return e->paused; });
}
void resume_reader(Engine *e) { e->pause_req = false; e->cv.notify_all(); }

// Drop every batch (filled or in flight) and restart reading at file offset `addr`.
int restart_reader_at(Engine *e, int64_t addr) {
    std::unique_lock<std::mutex> lk(e->m);
    pause_reader(e, lk);
    for (uint64_t s = e->done_seq + (e->cur_loaded ? 1 : 0); s < e->fill_seq; s++) {
        ReadBatch &b = e->rb[s % e->NPIPES];
        if (b.submitted) { (void)hg_pipe_wait(b.pipe, nullptr, nullptr, nullptr, nullptr, nullptr); b.submitted = false; }
    }
    e->fill_seq = e->done_seq = 0; e->cur_loaded = false;
    e->carry.clear(); e->input_done = false; e->window = WINDOW_MIN; e->eof_seen = false;
    e->blk = 0; e->blk_pending = false;
    int rc = 0;
    if (hseek(e->fp->fp, (off_t)addr, SEEK_SET) < 0) rc = -1;
    e->read_off = addr;
    if (rc != 0) e->input_done = true;
    else {
        // The first (small) window is read and submitted right here, on the caller's thread: no hand-over to the I/O
        // thread on the latency path of a random access.  Nothing is prefetched behind it until widen_readahead().
        e->ahead = 1;
        if (fill_batch(e, e->rb[0], HOST_FIRST_BLOCKS)) e->input_done = true;
        e->fill_seq = 1;
    }
    resume_reader(e);
    return rc;
}

// The consumer has moved past the first blocks after a seek: it is scanning, let the I/O thread run ahead again.
inline void widen_readahead(Engine *e) {
    if (e->ahead >= e->NPIPES) return;
    std::lock_guard<std::mutex> lk(e->m);
    e->ahead = e->NPIPES; e->cv.notify_all();
}

// Make the next batch current.  Returns 1 = batch loaded, 0 = end of input, -1 = error (errcode set).
int next_batch(Engine *e) {
    BGZF *fp = e->fp;
    if (!start_reader(e)) { fp->errcode |= BGZF_ERR_IO; return -1; }
    std::unique_lock<std::mutex> lk(e->m);
    if (e->cur_loaded) { e->cur_loaded = false; e->done_seq++; e->ahead = e->NPIPES; e->cv.notify_all(); }
    e->cv.wait(lk, [&] { return e->done_seq < e->fill_seq || e->input_done; });
    if (e->done_seq == e->fill_seq) return 0;
    ReadBatch &b = e->rb[e->done_seq % e->NPIPES];
    lk.unlock();
    b.plain = nullptr; b.status = nullptr; b.good = 0;
    if (b.submitted) {
        size_t plen = 0;
        const double t_w0 = stats_on() ? now_s() : 0;
        const int rc = hg_pipe_wait(b.pipe, &b.plain, &plen, &b.status, nullptr, nullptr);
        if (stats_on()) e->st.c_wait += now_s() - t_w0;
        b.submitted = false;
        if (rc != HG_OK && rc != HG_EBLOCK) { b.desc.clear(); b.fail = BGZF_ERR_ZLIB; }
        while (b.good < b.desc.size() && b.status[b.good] == 0) b.good++;
    }
    if (b.host) { b.plain = b.hplain.data(); b.status = b.hstatus.data(); b.good = 0; }
    lk.lock();
    e->cur_loaded = true;
    e->blk = 0; e->blk_pending = true;
    return 1;
}

inline ReadBatch &cur_batch(Engine *e) { return e->rb[e->done_seq % e->NPIPES]; }

// Bookkeeping for a block the consumer steps onto or over (bgzf.c:1066-1076).
inline void note_block(BGZF *fp, const ReadBatch &b, size_t i) {
    const hg_bgzf_desc &d = b.desc[i];
    fp->last_block_eof = d.ulen == 0;
    if (d.ulen && fp->idx_build_otf && fp->idx && !fp->idx->loaded) {
        fp->idx->offs.push_back(GziEntry{(uint64_t)(b.file_off + (int64_t)d.coff), fp->idx->ublock_addr});
        fp->idx->ublock_addr += d.ulen;
    }
}

// Engine reader: position on the next non-empty block.  0 ok (block_length == 0 at EOF), -1 error.
int engine_read_block(BGZF *fp) {
    Engine *e = E(fp);
    for (;;) {
        if (e->eof_seen) { fp->block_length = 0; return 0; }
        if (!e->cur_loaded) {
            const int r = next_batch(e);
            if (r < 0) return -1;
            if (r == 0) {                                                // end of input (bgzf.c:1044-1051)
                if (!fp->last_block_eof && !fp->no_eof_block) {
                    fp->no_eof_block = 1;
                    logmsg(LOG_WARNING, "bgzf_read_block", "EOF marker is absent. The input may be truncated");
                }
                e->eof_seen = true;
                fp->block_length = 0;
                return 0;
            }
        }
        ReadBatch &b = cur_batch(e);
        if (e->blk_pending) e->blk_pending = false; else e->blk++;
        if (e->blk >= 2) widen_readahead(e);
        if (e->blk >= b.desc.size()) {
            if (b.fail) {
                fp->errcode |= b.fail;
                logmsg(LOG_ERROR, "bgzf_read_block", "%s at offset %lld", b.fail == BGZF_ERR_HEADER ? "Invalid BGZF header" :
                       b.fail == BGZF_ERR_IO ? "Failed to read BGZF block data" : "Inflate block operation failed",
                       (long long)(b.file_off + (int64_t)b.comp_len));
                e->blk = b.desc.size(); e->blk_pending = true;          // stay here: later calls fail the same way
                return -1;
            }
            std::unique_lock<std::mutex> lk(e->m);
            e->cur_loaded = false; e->done_seq++; e->ahead = e->NPIPES; e->cv.notify_all();
            continue;
        }
        const hg_bgzf_desc &d = b.desc[e->blk];
        const int64_t addr = b.file_off + (int64_t)d.coff;
        if (b.host && b.hstatus[e->blk] == HOST_PENDING) {                  // bgzf_read_block's single-threaded branch: inflate_block on this thread (bgzf.c:1198-1205)
            hgh::Inflater &I = tls_inflater();
            b.hstatus[e->blk] = hgh::bgzf_block_inflate(I, b.hcomp + d.coff, d.clen, b.hplain.data() + d.uoff, d.ulen);
            while (b.good < b.desc.size() && b.hstatus[b.good] == 0) b.good++;
        }
        if (b.status[e->blk] != 0) {
            fp->errcode |= b.status[e->blk] == HG_BLOCK_ECRC ? BGZF_ERR_CRC : BGZF_ERR_ZLIB;
            logmsg(LOG_ERROR, "bgzf_read_block", "BGZF decode returned error %d for block offset %lld", (int)b.status[e->blk], (long long)addr);
            e->blk_pending = true;
            return -1;
        }
        e->next_addr = addr + d.clen;
        note_block(fp, b, e->blk);
        // empty blocks are skipped (bgzf.c:1054-1061) WITHOUT moving fp->block_address: it stays where bgzf_read left it when the previous block ran
        // out, the start of the empty block (bgzf.c:1064-1066 `if (!j->hit_eof)`, :1147-1155) -- so bgzf_tell() at the end of a file is the address of
        // the EOF block, which is what index builders store as the last chunk's end (hts_idx_finish(idx, bgzf_tell(fp)), test/index.bcf.csi)
        if (d.ulen == 0) continue;
        if (fp->block_length != 0) fp->block_offset = 0;                 // a seek's offset survives (bgzf.c:1064)
        fp->block_address = addr;
        fp->block_clength = (int)d.clen;
        fp->block_length = (int)d.ulen;
        fp->uncompressed_block = const_cast<uint8_t *>(b.plain) + d.uoff;   // no copy: the batch's pinned plain image
        return 0;
    }
}

// ---- large copies out of the pinned plain image.  A caller that reads megabytes per bgzf_read (bgzip -d, bulk loaders) is
// bounded by ONE thread's memcpy (~20 GB/s) while the pipeline behind it delivers several times that, so copies of
// COPY_SPLIT_MIN bytes and more are cut into slices for a few helper threads (process-wide, created on first use;
// HTS_GPU_COPY_THREADS = number of helpers, default 3, 0 = never).  Record-sized reads never get here.
constexpr size_t COPY_SPLIT_MIN = 2u << 20;
class CopyPool {
    struct Job { uint8_t *d; const uint8_t *s; size_t n; };
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    std::vector<Job> q;
    size_t pending = 0;
    bool stop = false;
    void worker() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (stop && q.empty()) return;
            const Job j = q.back(); q.pop_back();
            lk.unlock();
            memcpy(j.d, j.s, j.n);
            lk.lock();
            if (--pending == 0) done_cv.notify_all();
        }
    }
public:
    explicit CopyPool(int n) { for (int i = 0; i < n; i++) th.emplace_back([this] { worker(); }); }
    ~CopyPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
    size_t helpers() const { return th.size(); }
    // one caller at a time per pool use is enough here: concurrent callers serialise on `big`
    std::mutex big;
    void copy(uint8_t *d, const uint8_t *s, size_t n) {
        std::lock_guard<std::mutex> one(big);
        const size_t parts = th.size() + 1, each = ((n / parts) + 4095) & ~(size_t)4095;
        size_t off = each < n ? each : n;                                   // the caller takes the first slice itself
        {
            std::lock_guard<std::mutex> lk(m);
            for (size_t o = off; o < n; o += each) { q.push_back(Job{d + o, s + o, n - o < each ? n - o : each}); pending++; }
        }
        cv.notify_all();
        memcpy(d, s, off);
        std::unique_lock<std::mutex> lk(m);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
};
CopyPool *copy_pool() {
    static CopyPool *pool = [] {
        const char *v = getenv("HTS_GPU_COPY_THREADS");
        const int n = v ? atoi(v) : 3;
        return n > 0 ? new CopyPool(n > 16 ? 16 : n) : nullptr;             // lives until exit (threads are parked on a condition variable)
    }();
    return pool;
}
inline void copy_out(uint8_t *d, const uint8_t *s, size_t n) {
    CopyPool *p = n >= COPY_SPLIT_MIN ? copy_pool() : nullptr;
    if (p) p->copy(d, s, n); else memcpy(d, s, n);
}

// The current block is used up: "tell never points at the end of a block" (bgzf.c:1282-1285).
inline void block_consumed(BGZF *fp) {
    Engine *e = E(fp);
    fp->block_address = e && e->kind == K_READ ? e->next_addr : hg_htell(fp->fp);
    fp->block_offset = 0; fp->block_length = 0;
}

// Bytes readable without another bgzf_read_block: the rest of the current block and, for engine readers, every
// following block of the batch up to the first bad one -- they are adjacent in the plain image.
inline size_t span_avail(BGZF *fp, Engine *e) {
    const size_t in_block = (size_t)(fp->block_length - fp->block_offset);
    if (!e || e->kind != K_READ || !e->cur_loaded) return in_block;
    const ReadBatch &b = cur_batch(e);
    if (e->blk >= b.good) return in_block;
    const uint64_t here = b.desc[e->blk].uoff + (uint64_t)fp->block_offset;
    const uint64_t end = b.good < b.desc.size() ? b.desc[b.good].uoff : b.desc.back().uoff + b.desc.back().ulen;
    return (size_t)(end - here);
}

// Move the read position `n` bytes forward inside the span (n <= span_avail), stepping over whole blocks.
void span_advance(BGZF *fp, Engine *e, size_t n) {
    const size_t in_block = (size_t)(fp->block_length - fp->block_offset);
    if (n < in_block) { fp->block_offset += (int)n; return; }
    if (n == in_block) { block_consumed(fp); return; }
    ReadBatch &b = cur_batch(e);
    uint64_t target = b.desc[e->blk].uoff + (uint64_t)fp->block_offset + n;
    size_t j = e->blk + 1;
    for (;; j++) {                                                        // j < b.good by construction
        const hg_bgzf_desc &d = b.desc[j];
        note_block(fp, b, j);
        e->next_addr = b.file_off + (int64_t)d.coff + d.clen;
        if (target <= d.uoff + d.ulen && d.ulen) break;
    }
    const hg_bgzf_desc &d = b.desc[j];
    e->blk = j;
    if (j >= 2) widen_readahead(e);
    fp->block_address = b.file_off + (int64_t)d.coff;
    fp->block_clength = (int)d.clen;
    fp->block_length = (int)d.ulen;
    fp->block_offset = (int)(target - d.uoff);
    fp->uncompressed_block = const_cast<uint8_t *>(b.plain) + d.uoff;
    if (fp->block_offset == fp->block_length) block_consumed(fp);
}

// ================================================================================ plain gzip reader
int gz_read_block(BGZF *fp) {
    Engine *e = E(fp);
    const int64_t addr = e->gz_base + (int64_t)(e->gz_st.in_bit >> 3);
    while (e->gz_out_pos >= e->gz_out.size() && !e->gz_finished) {
        e->gz_out.clear(); e->gz_out_pos = 0;
        // top the compressed window up
        if (!e->gz_eof && e->gz_comp.size() - (size_t)(e->gz_st.in_bit >> 3) < (1u << 20)) {
            const size_t old = e->gz_comp.size(), add = 4u << 20;
            e->gz_comp.resize(old + add);
            ssize_t n = hg_hread(fp->fp, e->gz_comp.data() + old, add);
            if (n < 0) { fp->errcode |= BGZF_ERR_IO; return -1; }
            e->gz_comp.resize(old + (size_t)n);
            if ((size_t)n < add) e->gz_eof = true;
        }
        if (!e->gz_st.in_member && (size_t)(e->gz_st.in_bit >> 3) >= e->gz_comp.size() && e->gz_eof) { e->gz_finished = true; break; }
        const size_t soft = 8u << 20, cap = 40u << 20;
        e->gz_out.resize(cap);
        size_t made = 0;
        const bool fresh = e->gz_st.in_member == 0;
        const int rc = hg_gzip_stream_inflate_host(e->gpu, e->gz_comp.data(), e->gz_comp.size(), e->gz_eof ? 1 : 0, &e->gz_st,
                                                   fresh ? nullptr : e->gz_hist.data(), fresh ? 0 : e->gz_hist.size(),
                                                   e->gz_out.data(), cap, soft, &made);
        if (rc < 0) {
            logmsg(LOG_ERROR, "bgzf_read_block", "Reading GZIP stream failed at offset %lld", (long long)addr);
            fp->errcode |= (rc == HG_EBLOCK && e->gz_eof && e->gz_st.in_member) ? BGZF_ERR_ZLIB : BGZF_ERR_ZLIB;
            e->gz_out.clear();
            return -1;
        }
        e->gz_out.resize(made);
        if (rc == HG_GZ_NEEDIN) {
            if (e->gz_eof || e->gz_comp.size() > (1ull << 30)) {
                logmsg(LOG_ERROR, "bgzf_read_block", "Gzip file truncated");
                fp->errcode |= BGZF_ERR_IO;
                return -1;
            }
            const size_t old = e->gz_comp.size(), add = old < (4u << 20) ? (4u << 20) : old;
            e->gz_comp.resize(old + add);
            ssize_t n = hg_hread(fp->fp, e->gz_comp.data() + old, add);
            if (n < 0) { fp->errcode |= BGZF_ERR_IO; return -1; }
            e->gz_comp.resize(old + (size_t)n);
            if ((size_t)n < add) e->gz_eof = true;
            continue;
        }
        // history for the next call = the last 32 KiB of this member's output
        if (rc == HG_GZ_MEMBER) e->gz_hist.clear();
        else if (made >= 32768) e->gz_hist.assign(e->gz_out.end() - 32768, e->gz_out.end());
        else {
            e->gz_hist.insert(e->gz_hist.end(), e->gz_out.begin(), e->gz_out.end());
            if (e->gz_hist.size() > 32768) e->gz_hist.erase(e->gz_hist.begin(), e->gz_hist.end() - 32768);
        }
        // forget consumed input (dword granular so that bit offsets stay valid)
        const size_t drop = (size_t)(e->gz_st.in_bit >> 3) & ~(size_t)3;
        if (drop > (1u << 20) || drop == e->gz_comp.size()) {
            e->gz_comp.erase(e->gz_comp.begin(), e->gz_comp.begin() + drop);
            e->gz_base += (int64_t)drop; e->gz_st.in_bit -= (uint64_t)drop * 8u;
        }
    }
    const size_t left = e->gz_out.size() - e->gz_out_pos;
    const size_t take = left < BGZF_MAX_BLOCK_SIZE ? left : BGZF_MAX_BLOCK_SIZE;
    fp->uncompressed_block = e->own_block;
    if (take) memcpy(fp->uncompressed_block, e->gz_out.data() + e->gz_out_pos, take);
    e->gz_out_pos += take;
    if (fp->block_length != 0) fp->block_offset = 0;
    fp->block_address = addr;
    fp->block_length = (int)take;
    return 0;
}

// ================================================================================ writer: output thread
int flush_index_pushes(Engine *e, uint64_t block_addr, size_t ulen, size_t clen) {     // bgzf_idx_flush, bgzf.c:228-290
    std::lock_guard<std::mutex> g(e->idx_m);
    size_t i = 0;
    for (; i < e->pushes.size() && e->pushes[i].block_number == e->block_written; i++) {
        const IdxPush &p = e->pushes[i];
        if (!hts_idx_push) return -1;
        if (ulen > 0 && p.offset == ulen) {
            // an offset at the very end of a block means the start of the next one; it is this block's last entry
            if (hts_idx_push(p.hidx, p.tid, p.beg, p.end, (block_addr + clen) << 16, p.mapped) < 0) return -1;
            i++;
            break;
        }
        if (hts_idx_push(p.hidx, p.tid, p.beg, p.end, (block_addr << 16) + p.offset, p.mapped) < 0) return -1;
    }
    e->pushes.erase(e->pushes.begin(), e->pushes.begin() + (long)i);
    e->block_written++;
    return 0;
}

void writer_main(Engine *e) {
    BGZF *fp = e->fp;
    std::unique_lock<std::mutex> lk(e->m);
    for (;;) {
        e->cv.wait(lk, [&] { return e->stop || e->w_done < e->w_fill; });
        if (e->w_done >= e->w_fill) { if (e->stop) break; continue; }
        WriteBatch &b = e->wb[e->w_done % e->NPIPES];
        lk.unlock();
        const uint8_t *out = nullptr; size_t out_len = 0; const uint64_t *off = nullptr; const uint32_t *crc = nullptr;
        int err = 0;
        const size_t n = b.cuts.size() - 1;
        const double t_o0 = stats_on() ? now_s() : 0;
        if (hg_pipe_wait(b.pipe, &out, &out_len, nullptr, &off, &crc) != HG_OK) err = BGZF_ERR_ZLIB;
        const double t_o1 = stats_on() ? now_s() : 0;
        int64_t addr = e->block_address;
        if (!err) {
            if (e->kind == K_GZWRITE && !e->gz_header_done) {
                const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
                if (hg_hwrite(fp->fp, hdr, 10) != 10) err = BGZF_ERR_IO;
                e->gz_header_done = true; addr += 10;
            }
            for (size_t i = 0; i < n && !err; i++) {
                const size_t clen = (size_t)(off[i + 1] - off[i]), ulen = (size_t)(b.cuts[i + 1] - b.cuts[i]);
                if (e->kind == K_GZWRITE) { e->gz_crc = crc_concat(e->gz_crc, crc[i], ulen); e->gz_isize += ulen; }
                if (fp->idx_build_otf && fp->idx) {
                    const GziEntry last = fp->idx->offs.back();
                    fp->idx->offs.push_back(GziEntry{last.caddr + clen, last.uaddr + ulen});
                }
                if (flush_index_pushes(e, (uint64_t)(addr + (int64_t)off[i]), ulen, clen) != 0) err = BGZF_ERR_IO;
            }
            if (!err && out_len && hg_hwrite(fp->fp, out, out_len) != (ssize_t)out_len) err = BGZF_ERR_IO;
        }
        if (stats_on()) { e->wst.o_wait += t_o1 - t_o0; e->wst.o_write += now_s() - t_o1; e->wst.batches++; e->wst.blocks += n; e->wst.out_bytes += out_len; }
        lk.lock();
        if (err) e->w_err |= err; else e->block_address = addr + (int64_t)out_len;
        e->w_done++;
        e->cv.notify_all();
    }
}

bool start_writer(Engine *e) {
    if (e->started) return true;
    for (int i = 0; i < e->NPIPES; i++) if (!e->wb[i].pipe && !get_pipe(e, i, &e->wb[i].pipe)) return false;
    e->started = true;
    e->th = std::thread(writer_main, e);
    return true;
}

// Hand the batch being filled to the device.
int submit_write_batch(Engine *e) {
    BGZF *fp = e->fp;
    if (!e->w_open) return 0;
    WriteBatch &b = e->wb[e->w_fill % e->NPIPES];
    e->w_open = false;
    if (b.cuts.size() <= 1) return 0;
    int level = fp->compress_level < 0 ? 6 : fp->compress_level;
    const double t_s0 = stats_on() ? now_s() : 0;
    const int src = hg_pipe_deflate(b.pipe, b.len, b.cuts.data(), b.cuts.size() - 1, level, e->kind == K_GZWRITE);
    if (stats_on()) e->wst.c_submit += now_s() - t_s0;
    if (src != HG_OK) {
        fp->errcode |= BGZF_ERR_ZLIB;
        return -1;
    }
    std::lock_guard<std::mutex> g(e->m);
    e->w_fill++;
    e->cv.notify_all();
    return 0;
}

// Open the batch that the next blocks go to (waits for a free pipe).
int open_write_batch(BGZF *fp, Engine *e) {
    if (e->w_open) return 0;
    const double t_p0 = stats_on() ? now_s() : 0;
    std::unique_lock<std::mutex> lk(e->m);
    e->cv.wait(lk, [&] { return e->w_fill - e->w_done < e->NPIPES; });
    if (stats_on()) e->wst.c_pipe += now_s() - t_p0;
    if (e->w_err) { fp->errcode |= e->w_err; return -1; }
    lk.unlock();
    WriteBatch &b = e->wb[e->w_fill % e->NPIPES];
    b.target = fp->mt ? e->w_target : 1;                               // exact mode: one block per job
    if (fp->mt && e->w_target < WBLOCKS_MAX) e->w_target = e->w_target * 2 < WBLOCKS_MAX ? e->w_target * 2 : WBLOCKS_MAX;
    b.cap = b.target * (size_t)BGZF_BLOCK_SIZE;
    b.in = (uint8_t *)hg_pipe_input(b.pipe, b.cap);
    if (!b.in) { fp->errcode |= BGZF_ERR_IO; return -1; }
    b.len = 0; b.cuts.assign(1, 0);
    e->w_open = true;
    return 0;
}

// Queue fp->uncompressed_block[0 .. block_offset) as one block (mt_queue, bgzf.c:1852-1895).
int queue_block(BGZF *fp) {
    Engine *e = E(fp);
    if (fp->block_offset == 0) return 0;
    if (!start_writer(e)) { fp->errcode |= BGZF_ERR_IO; return -1; }
    if (open_write_batch(fp, e) != 0) return -1;
    WriteBatch &b = e->wb[e->w_fill % e->NPIPES];
    memcpy(b.in + b.len, fp->uncompressed_block, (size_t)fp->block_offset);
    b.len += (size_t)fp->block_offset;
    b.cuts.push_back(b.len);
    e->block_number++;
    fp->block_offset = 0;
    if (b.cuts.size() - 1 >= b.target || b.len + BGZF_BLOCK_SIZE > b.cap) return submit_write_batch(e);
    return 0;
}

int drain_writer(BGZF *fp);

// Whole blocks straight from the caller's buffer into the batches (after bgzf_mt(), at a block boundary): the same cuts as the block-by-block
// path, without the stop in fp->uncompressed_block -- a bulk writer (bgzip, a sorter's output) was bounded by the caller's thread copying every
// byte twice -- and with the copy helpers for spans of megabytes.  Returns the bytes taken (a multiple of the block size) or -1.
ssize_t queue_whole_blocks(BGZF *fp, const uint8_t *in, size_t length) {
    Engine *e = E(fp);
    size_t blocks = length / BGZF_BLOCK_SIZE, taken = 0;
    if (!blocks) return 0;
    if (!start_writer(e)) { fp->errcode |= BGZF_ERR_IO; return -1; }
    while (blocks) {
        if (open_write_batch(fp, e) != 0) return -1;
        WriteBatch &b = e->wb[e->w_fill % e->NPIPES];
        size_t room = b.target - (b.cuts.size() - 1);
        const size_t by_cap = (b.cap - b.len) / BGZF_BLOCK_SIZE;
        if (room > by_cap) room = by_cap;
        if (room > blocks) room = blocks;
        if (room) {
            const double t_c0 = stats_on() ? now_s() : 0;
            copy_out(b.in + b.len, in + taken, room * (size_t)BGZF_BLOCK_SIZE);
            if (stats_on()) e->wst.c_copy += now_s() - t_c0;
            for (size_t i = 0; i < room; i++) { b.len += BGZF_BLOCK_SIZE; b.cuts.push_back(b.len); }
            e->block_number += room;
            taken += room * (size_t)BGZF_BLOCK_SIZE; blocks -= room;
        }
        if (b.cuts.size() - 1 >= b.target || b.len + BGZF_BLOCK_SIZE > b.cap) { if (submit_write_batch(e) != 0) return -1; }
    }
    return (ssize_t)taken;
}

// A block has been cut (lazy_flush, bgzf.c:1927-1933).  Without bgzf_mt() the reference compresses it on the spot and
// callers may rely on an exact bgzf_tell() between writes (test/test_bgzf.c test_tell_seek_getc; sam.c:942-943 amends
// the index with it), so the block goes through the engine alone and is waited for: correct, but one device round
// trip per block.  After bgzf_mt() blocks are batched and fp->block_address is only valid after bgzf_flush(), exactly
// as with the reference's threads (bgzf.c:1953-1967).
int host_cut_block(BGZF *fp);
int cut_block(BGZF *fp) {
    if (!fp->mt && E(fp)->kind == K_WRITE) return host_cut_block(fp);
    if (queue_block(fp) != 0) return -1;
    return fp->mt ? 0 : drain_writer(fp);
}

// The reference's single-threaded writer (bgzf_flush -> deflate_block + hwrite, bgzf.c:1949-1994, 709-726): without bgzf_mt() a block is compressed
// when it is cut, on the caller's thread, and bgzf_tell() is exact right away.  A device job for ONE block costs a launch + a PCIe round trip (~4 ms:
// 16 MB/s); the host codec (bgzf_host_codec.h) does a block in ~0.6 ms.  The handle still owns a live engine: bgzf_mt() at any later point switches to
// batched device jobs, and e->block_address / block_number stay those of the output thread's bookkeeping.
int host_cut_block(BGZF *fp) {
    Engine *e = E(fp);
    if (fp->block_offset == 0) return 0;
    if (e->started && drain_writer(fp) != 0) return -1;                  // (device jobs of an earlier bgzf_mt phase first: order on disk)
    hgh::Deflater &D = tls_deflater();
    uint8_t *dst = (uint8_t *)fp->compressed_block;
    size_t dlen = BGZF_MAX_BLOCK_SIZE;
    const size_t ulen = (size_t)fp->block_offset;
    if (hgh::bgzf_block_deflate(D, dst, &dlen, (const uint8_t *)fp->uncompressed_block, ulen, fp->compress_level < 0 ? 6 : fp->compress_level) != 0) {
        fp->errcode |= BGZF_ERR_ZLIB;
        return -1;
    }
    if (fp->idx_build_otf && fp->idx) {
        const GziEntry last = fp->idx->offs.back();
        fp->idx->offs.push_back(GziEntry{last.caddr + dlen, last.uaddr + ulen});
    }
    if (hg_hwrite(fp->fp, dst, dlen) != (ssize_t)dlen) { fp->errcode |= BGZF_ERR_IO; return -1; }
    {
        std::lock_guard<std::mutex> g(e->m);
        e->block_address += (int64_t)dlen; e->block_number++; e->block_written++;
        fp->block_address = e->block_address;
    }
    fp->block_offset = 0;
    return 0;
}

int drain_writer(BGZF *fp) {
    Engine *e = E(fp);
    if (submit_write_batch(e) != 0) return -1;
    std::unique_lock<std::mutex> lk(e->m);
    e->cv.wait(lk, [&] { return e->w_done == e->w_fill; });
    if (e->w_err) { fp->errcode |= e->w_err; return -1; }
    fp->block_address = e->block_address;                                  // bgzf.c:1953-1967
    return 0;
}

// ================================================================================ handle lifetime
void stop_engine(Engine *e) {
    if (stats_on() && e->kind != K_READ && e->wst.batches)
        fprintf(stderr, "[htsgpu stats] writer: %llu jobs, %llu blocks, %.1f MB out; caller: wait for a pipe %.3f s, bulk copies %.3f s, submit %.3f s; output thread: "
                "wait for the device %.3f s, index + hwrite %.3f s\n", (unsigned long long)e->wst.batches, (unsigned long long)e->wst.blocks, e->wst.out_bytes / 1e6,
                e->wst.c_pipe, e->wst.c_copy, e->wst.c_submit, e->wst.o_wait, e->wst.o_write);
    e->wst.batches = 0;                                                     // (printed once)
    if (stats_on() && e->kind == K_READ && e->st.batches)
        fprintf(stderr, "[htsgpu stats] reader: %llu batches; I/O thread: read %.3f s, frame %.3f s, submit %.3f s; consumer: wait for batch %.3f s, "
                "large copies %.3f s (%llu)\n", (unsigned long long)e->st.batches, e->st.io_read, e->st.io_frame, e->st.io_submit, e->st.c_wait,
                e->st.c_copy, (unsigned long long)e->st.copies);
    if (e->started) {
        { std::lock_guard<std::mutex> g(e->m); e->stop = true; e->pause_req = false; e->cv.notify_all(); }
        e->th.join();
        e->started = false;
    }
    // idle pipes go back to their device's list; context + pipes are parked for the next handle (EnginePark)
    if (e->devs.empty()) { if (e->pfd >= 0) { close(e->pfd); e->pfd = -1; } return; }
    for (int i = 0; i < NPIPES_MAX; i++) {
        ReadBatch &b = e->rb[i];
        if (b.pipe) { if (b.submitted) (void)hg_pipe_wait(b.pipe, nullptr, nullptr, nullptr, nullptr, nullptr); b.submitted = false; e->spare[(size_t)i % e->devs.size()].push_back(b.pipe); b.pipe = nullptr; }
        WriteBatch &w = e->wb[i];
        if (w.pipe) { e->spare[(size_t)i % e->devs.size()].push_back(w.pipe); w.pipe = nullptr; }
    }
    for (size_t i = 0; i < e->devs.size(); i++) engine_park().give(e->dev_ids[i], e->devs[i], e->spare[i]);
    e->devs.clear(); e->dev_ids.clear(); e->spare.clear(); e->gpu = nullptr;
    if (e->pfd >= 0) { close(e->pfd); e->pfd = -1; }
}

void free_handle(BGZF *fp) {
    Engine *e = E(fp);
    if (e) { stop_engine(e); free(e->own_block); delete e; }
    else free(fp->uncompressed_block);
    delete fp->idx;
    free(fp);
}

int mode_level(const char *mode) {            // first digit = level, 'u' = no compression
    int level = -1;
    for (const char *m = mode; *m; m++) if (*m >= '0' && *m <= '9') { level = *m - '0'; break; }
    return strchr(mode, 'u') ? -2 : level;
}

Engine *new_engine(BGZF *fp, Kind kind) {
    Engine *e = new Engine();
    e->fp = fp; e->kind = kind;
    for (int d : device_list()) {
        hg_ctx *c = nullptr;
        std::vector<hg_pipe *> pipes;
        if (!engine_park().take(d, c, pipes) && hg_init(d, &c) != HG_OK) {
            for (size_t i = 0; i < e->devs.size(); i++) engine_park().give(e->dev_ids[i], e->devs[i], e->spare[i]);
            delete e; errno = ENODEV; return nullptr;
        }
        e->devs.push_back(c); e->dev_ids.push_back(d); e->spare.push_back(std::move(pipes));
    }
    e->gpu = e->devs[0];
    e->NPIPES = e->devs.size() == 1 ? NPIPES_MIN : (int)std::min<size_t>(2 * e->devs.size(), (size_t)NPIPES_MAX);
    if (const char *np = getenv("HTS_GPU_PIPES")) { const int v = atoi(np); if (v >= 2 && v <= NPIPES_MAX) e->NPIPES = v; }
    e->ahead = (size_t)e->NPIPES;
    e->own_block = fp->uncompressed_block;
    fp->cache = reinterpret_cast<bgzf_cache_t *>(e);
    if (kind == K_READ || kind == K_GZREAD) fp->mt = reinterpret_cast<struct bgzf_mtaux_t *>(e);
    return e;
}

BGZF *make_reader(hFILE *h) {
    uint8_t magic[18];
    const ssize_t n = hpeek(h, magic, 18);
    if (n < 0) return nullptr;
    BGZF *fp = (BGZF *)calloc(1, sizeof(BGZF));
    if (!fp) return nullptr;
    fp->uncompressed_block = malloc(2 * BGZF_MAX_BLOCK_SIZE);
    if (!fp->uncompressed_block) { free(fp); return nullptr; }
    fp->compressed_block = (uint8_t *)fp->uncompressed_block + BGZF_MAX_BLOCK_SIZE;
    fp->fp = h;
    fp->is_compressed = n == 18 && magic[0] == 0x1f && magic[1] == 0x8b;
    const bool extra = fp->is_compressed && (magic[3] & 4);
    fp->is_gzip = fp->is_compressed && !(extra && memcmp(magic + 12, "BC\2\0", 4) == 0);
    if (extra && memcmp(magic + 12, "RAZF", 4) == 0) {
        logmsg(LOG_ERROR, "bgzf_read_init", "Cannot decompress legacy RAZF format");
        free(fp->uncompressed_block); free(fp);
        errno = ENOEXEC;
        return nullptr;
    }
    if (fp->is_compressed && !new_engine(fp, fp->is_gzip ? K_GZREAD : K_READ)) { free(fp->uncompressed_block); free(fp); return nullptr; }
    return fp;
}

BGZF *make_writer(hFILE *h, const char *mode) {
    BGZF *fp = (BGZF *)calloc(1, sizeof(BGZF));
    if (!fp) return nullptr;
    fp->is_write = 1;
    fp->fp = h;
    const int level = mode_level(mode);
    if (level == -2) return fp;                                             // "u": pass-through, no buffers (bgzf.c:446-450)
    fp->is_compressed = 1;
    fp->uncompressed_block = malloc(2 * BGZF_MAX_BLOCK_SIZE);
    if (!fp->uncompressed_block) { free(fp); return nullptr; }
    fp->compressed_block = (uint8_t *)fp->uncompressed_block + BGZF_MAX_BLOCK_SIZE;
    fp->compress_level = level < 0 || level > 9 ? -1 : level;
    if (strchr(mode, 'g')) fp->is_gzip = 1;
    Engine *e = new_engine(fp, fp->is_gzip ? K_GZWRITE : K_WRITE);
    if (!e) { free(fp->uncompressed_block); free(fp); return nullptr; }
    return fp;
}

BGZF *finish_open(BGZF *fp) {
    if (fp) { const uint16_t one = 1; fp->is_be = *(const uint8_t *)&one == 0; }
    return fp;
}

}  // namespace

extern "C" {

// -------------------------------------------------------------------------------- open / close
BGZF *bgzf_hopen(hFILE *h, const char *mode) {
    if (strchr(mode, 'r')) return finish_open(make_reader(h));
    if (strchr(mode, 'w') || strchr(mode, 'a')) return finish_open(make_writer(h, mode));
    errno = EINVAL;
    return nullptr;
}

// A BGZF reader over a regular local file gets a descriptor of its own for positional window reads (ReadPool).
static void adopt_pread_fd(BGZF *fp, int fd) {
    Engine *e = fp ? E(fp) : nullptr;
    struct stat st;
    if (e && e->kind == K_READ && fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) e->pfd = fd;
    else if (fd >= 0) close(fd);
}

BGZF *bgzf_open(const char *path, const char *mode) {
    if (!strchr(mode, 'r') && !strchr(mode, 'w') && !strchr(mode, 'a')) { errno = EINVAL; return nullptr; }
    hFILE *h = hopen(path, mode);
    if (!h) return nullptr;
    BGZF *fp = bgzf_hopen(h, mode);
    if (!fp) hclose_abruptly(h);
    else if (fp->is_compressed && !fp->is_write && strcmp(path, "-") != 0 && !strstr(path, "://") && strncmp(path, "file:", 5) != 0 &&
             strncmp(path, "data:", 5) != 0 && strncmp(path, "preload:", 8) != 0) {
        const int keep = errno;
        adopt_pread_fd(fp, open(path, O_RDONLY | O_CLOEXEC));
        errno = keep;
    }
    return fp;
}

BGZF *bgzf_dopen(int fd, const char *mode) {
    if (!strchr(mode, 'r') && !strchr(mode, 'w') && !strchr(mode, 'a')) { errno = EINVAL; return nullptr; }
    // hdopen's logical offsets start at 0 wherever the descriptor stands (hfile.c:640-680), and the stream begins THERE: positional window reads
    // use file offsets, so they are only taken when the two agree (descriptor at the start of a regular file).  A descriptor that was read from,
    // lseek()ed, or is a pipe keeps the hread path, which -- like the reference -- reads from the current position.
    const int keep0 = errno;
    const off_t base = lseek(fd, 0, SEEK_CUR);
    errno = keep0;
    hFILE *h = hdopen(fd, mode);
    if (!h) return nullptr;
    BGZF *fp = bgzf_hopen(h, mode);
    if (!fp) hclose_abruptly(h);
    else if (fp->is_compressed && !fp->is_write && base == 0) { const int keep = errno; adopt_pread_fd(fp, dup(fd)); errno = keep; }
    return fp;
}

hFILE *bgzf_hfile(BGZF *fp) { return fp->fp; }

int bgzf_close(BGZF *fp) {
    if (!fp) return -1;
    Engine *e = E(fp);
    bool ok = true;
    if (fp->is_write && fp->is_compressed) {
        if (bgzf_flush(fp) != 0) ok = false;
        else if (e->kind == K_GZWRITE) {                                     // end the member: final empty stored block + trailer
            uint8_t tail[23]; size_t k = 0;
            if (!e->gz_header_done) { const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3}; memcpy(tail, hdr, 10); k = 10; }
            const uint8_t fin[5] = {1, 0, 0, 0xff, 0xff};
            memcpy(tail + k, fin, 5); k += 5;
            for (int i = 0; i < 4; i++) { tail[k + i] = (uint8_t)(e->gz_crc >> (8 * i)); tail[k + 4 + i] = (uint8_t)(e->gz_isize >> (8 * i)); }
            k += 8;
            if (hg_hwrite(fp->fp, tail, k) != (ssize_t)k || hflush(fp->fp) != 0) { fp->errcode |= BGZF_ERR_IO; ok = false; }
        } else if (hg_hwrite(fp->fp, kEof, 28) != 28 || hflush(fp->fp) != 0) {   // bgzf.c:2084-2101
            logmsg(LOG_ERROR, "bgzf_close", "File write failed");
            fp->errcode |= BGZF_ERR_IO; ok = false;
        }
    }
    if (e) stop_engine(e);
    if (hclose(fp->fp) != 0) ok = false;
    if (fp->errcode) ok = false;
    free_handle(fp);
    return ok ? 0 : -1;
}

// -------------------------------------------------------------------------------- reading
int bgzf_read_block(BGZF *fp) {
    if (fp->errcode) return -1;
    Engine *e = E(fp);
    if (e && e->kind == K_READ) return engine_read_block(fp);
    if (e && e->kind == K_GZREAD) return gz_read_block(fp);
    if (fp->is_write) { fp->errcode |= BGZF_ERR_MISUSE; return -1; }
    // uncompressed pass-through (bgzf.c:1110-1131)
    const int64_t at = hg_htell(fp->fp);
    const ssize_t n = hg_hread(fp->fp, fp->uncompressed_block, BGZF_MAX_BLOCK_SIZE);
    if (n < 0) {
        logmsg(LOG_ERROR, "bgzf_read_block", "Failed to read uncompressed data at offset %lld", (long long)at);
        fp->errcode |= BGZF_ERR_IO;
        return -1;
    }
    if (n == 0) { fp->block_length = 0; return 0; }
    if (fp->block_length != 0) fp->block_offset = 0;
    fp->block_address = at;
    fp->block_length = (int)n;
    return 0;
}

ssize_t bgzf_read(BGZF *fp, void *data, size_t length) {
    if (length == 0) return 0;
    if (fp->is_write) { fp->errcode |= BGZF_ERR_MISUSE; return -1; }
    Engine *e = E(fp);
    uint8_t *dst = (uint8_t *)data;
    size_t want = length;
    while (want) {
        if (fp->block_offset >= fp->block_length) {
            if (bgzf_read_block(fp) != 0) return -1;
            if (fp->block_length == 0) break;                                // end of file
            if (fp->block_offset > fp->block_length) { fp->errcode |= BGZF_ERR_MISUSE; return -1; }   // seek beyond the block
            if (fp->block_offset == fp->block_length) { block_consumed(fp); continue; }
        }
        size_t n = span_avail(fp, e);
        if (n > want) n = want;
        if (e && stats_on() && n >= COPY_SPLIT_MIN) {
            const double t_c0 = now_s();
            copy_out(dst, (const uint8_t *)fp->uncompressed_block + fp->block_offset, n);
            e->st.c_copy += now_s() - t_c0; e->st.copies++;
        } else
        copy_out(dst, (const uint8_t *)fp->uncompressed_block + fp->block_offset, n);
        span_advance(fp, e, n);
        dst += n; want -= n;
    }
    fp->uncompressed_address += (int64_t)(length - want);
    return (ssize_t)(length - want);
}

int bgzf_peek(BGZF *fp) {
    if (fp->block_offset >= fp->block_length) {
        if (bgzf_read_block(fp) != 0) { fp->errcode = BGZF_ERR_ZLIB; return -2; }
        if (fp->block_offset >= fp->block_length) return -1;
    }
    return ((const uint8_t *)fp->uncompressed_block)[fp->block_offset];
}

int bgzf_getc(BGZF *fp) {
    if (fp->block_offset >= fp->block_length) {
        if (bgzf_read_block(fp) != 0) return -2;
        if (fp->block_length == 0) return -1;
    }
    const int c = ((const uint8_t *)fp->uncompressed_block)[fp->block_offset++];
    if (fp->block_offset == fp->block_length) block_consumed(fp);
    fp->uncompressed_address++;
    return c;
}

// Lines are cut out of the batch's contiguous plain image: one memchr over the whole readable span, one copy.
int bgzf_getline(BGZF *fp, int delim, kstring_t *str) {
    Engine *e = E(fp);
    str->l = 0;
    bool found = false;
    int fail = 0;
    while (!found) {
        if (fp->block_offset >= fp->block_length) {
            if (bgzf_read_block(fp) != 0) { fail = -2; break; }
            if (fp->block_length == 0) { fail = -1; break; }
        }
        const uint8_t *p = (const uint8_t *)fp->uncompressed_block + fp->block_offset;
        const size_t span = span_avail(fp, e);
        const uint8_t *hit = (const uint8_t *)memchr(p, delim, span);
        const size_t take = hit ? (size_t)(hit - p) : span;
        if (str->l + take + 2 > str->m) {
            size_t m = str->l + take + 2;
            m += m >> 1;
            char *ns = (char *)realloc(str->s, m < 64 ? 64 : m);
            if (!ns) { fail = -3; break; }
            str->s = ns; str->m = m < 64 ? 64 : m;
        }
        memcpy(str->s + str->l, p, take);
        str->l += take;
        span_advance(fp, e, take + (hit ? 1 : 0));
        found = hit != nullptr;
    }
    if (fail < -1) return fail;
    if (fail == -1 && str->l == 0) return -1;
    fp->uncompressed_address += (int64_t)str->l + 1;
    if (delim == '\n' && str->l > 0 && str->s[str->l - 1] == '\r') str->l--;
    if (str->s) str->s[str->l] = '\0';
    return str->l <= 0x7fffffff ? (int)str->l : 0x7fffffff;
}

ssize_t bgzf_raw_read(BGZF *fp, void *data, size_t length) {
    const ssize_t n = hg_hread(fp->fp, data, length);
    if (n < 0) fp->errcode |= BGZF_ERR_IO;
    return n;
}

// -------------------------------------------------------------------------------- seeking
int64_t bgzf_seek(BGZF *fp, int64_t pos, int whence) {
    if (fp->is_write || whence != SEEK_SET || fp->is_gzip) { fp->errcode |= BGZF_ERR_MISUSE; return -1; }
    fp->seeked = pos;
    const int64_t addr = pos >> 16;
    Engine *e = E(fp);
    if (e) {
        bool inside = false;
        if (e->cur_loaded) {
            // a target inside the batch being consumed needs no I/O and no new launch
            const ReadBatch &b = cur_batch(e);
            if (addr >= b.file_off && addr < b.file_off + (int64_t)b.comp_len) {
                size_t lo = 0, hi = b.desc.size();
                while (lo < hi) { const size_t mid = (lo + hi) / 2; if (b.file_off + (int64_t)b.desc[mid].coff < addr) lo = mid + 1; else hi = mid; }
                if (lo < b.desc.size() && b.file_off + (int64_t)b.desc[lo].coff == addr) { e->blk = lo; e->blk_pending = true; inside = true; }
            }
        }
        e->eof_seen = false;
        if (!inside && (e->started ? restart_reader_at(e, addr) : (hseek(fp->fp, (off_t)addr, SEEK_SET) < 0 ? -1 : 0)) != 0) {
            fp->errcode |= BGZF_ERR_IO;
            return -1;
        }
    } else if (hseek(fp->fp, (off_t)addr, SEEK_SET) < 0) { fp->errcode |= BGZF_ERR_IO; return -1; }
    fp->block_length = 0;                                                    // "not loaded"
    fp->block_address = addr;
    fp->block_offset = (int)(pos & 0xFFFF);
    return 0;
}

int bgzf_useek(BGZF *fp, off_t uoffset, int where) {
    if (fp->is_write || where != SEEK_SET || fp->is_gzip) { fp->errcode |= BGZF_ERR_MISUSE; return -1; }
    const int64_t block_start = fp->uncompressed_address - fp->block_offset;
    if (uoffset >= block_start && uoffset < block_start + fp->block_length) {            // inside the loaded block
        fp->block_offset += (int)(uoffset - fp->uncompressed_address);
        fp->uncompressed_address = uoffset;
        return 0;
    }
    if (!fp->is_compressed) {
        if (hseek(fp->fp, uoffset, SEEK_SET) < 0) { fp->errcode |= BGZF_ERR_IO; return -1; }
        fp->block_length = 0; fp->block_address = uoffset; fp->block_offset = 0;
        if (bgzf_read_block(fp) < 0) { fp->errcode |= BGZF_ERR_IO; return -1; }
        fp->uncompressed_address = uoffset;
        return 0;
    }
    if (!fp->idx || fp->idx->offs.empty()) { fp->errcode |= BGZF_ERR_IO; return -1; }
    const std::vector<GziEntry> &o = fp->idx->offs;
    size_t lo = 0, hi = o.size();                                            // last entry with uaddr <= uoffset
    while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (o[mid].uaddr <= (uint64_t)uoffset) lo = mid; else hi = mid; }
    if (bgzf_seek(fp, (int64_t)(o[lo].caddr << 16), SEEK_SET) != 0) return -1;
    if (bgzf_read_block(fp) < 0) { fp->errcode |= BGZF_ERR_IO; return -1; }
    const int64_t inside = (int64_t)uoffset - (int64_t)o[lo].uaddr;
    if (inside > 0) {
        if (inside > fp->block_length) { fp->errcode |= BGZF_ERR_IO; return -1; }
        fp->block_offset = (int)inside;
    }
    fp->uncompressed_address = uoffset;
    return 0;
}

off_t bgzf_utell(BGZF *fp) { return (off_t)fp->uncompressed_address; }

int bgzf_check_EOF(BGZF *fp) {
    Engine *e = E(fp);
    std::unique_lock<std::mutex> lk;
    if (e && e->kind == K_READ && e->started) { lk = std::unique_lock<std::mutex>(e->m); pause_reader(e, lk); }
    int has = -1;
    const off_t here = hg_htell(fp->fp);
    if (hseek(fp->fp, -28, SEEK_END) < 0) {
        if (errno == ESPIPE) { fp->fp->has_errno = 0; has = 2; }
        else if (errno == EINVAL) { fp->fp->has_errno = 0; has = 0; }       // shorter than an EOF block
    } else {
        uint8_t buf[28];
        if (hg_hread(fp->fp, buf, 28) == 28 && hseek(fp->fp, here, SEEK_SET) >= 0) has = memcmp(buf, kEof, 28) == 0 ? 1 : 0;
    }
    if (lk.owns_lock()) resume_reader(e);
    fp->no_eof_block = has == 0;
    return has;
}

int bgzf_compression(BGZF *fp) { return !fp->is_compressed ? 0 /* no_compression */ : fp->is_gzip ? 1 /* gzip */ : 2 /* bgzf */; }

int bgzf_is_bgzf(const char *fn) {
    hFILE *h = hopen(fn, "r");
    if (!h) return 0;
    uint8_t buf[16];
    const ssize_t n = hg_hread(h, buf, 16);
    if (hclose(h) < 0) return 0;
    return n == 16 && bgzf_header_ok(buf);
}

void bgzf_set_cache_size(BGZF *fp, int size) { if (fp && !fp->mt) fp->cache_size = size; }   // no block cache: reads come from the prefetched batch
// bgzf_mt / bgzf_thread_pool: the engine is the pool, so no threads are created; for a writer the call switches on the
// reference's threaded contract (batched blocks, block addresses resolved later).  No-ops for uncompressed handles
// (bgzf.c:1742-1743, 1786-1787).
int bgzf_thread_pool(BGZF *fp, struct hts_tpool *, int) {
    Engine *e = E(fp);
    if (e && fp->is_write) fp->mt = reinterpret_cast<struct bgzf_mtaux_t *>(e);
    return 0;
}
int bgzf_mt(BGZF *fp, int, int) { return bgzf_thread_pool(fp, nullptr, 0); }

// -------------------------------------------------------------------------------- writing
ssize_t bgzf_write(BGZF *fp, const void *data, size_t length) {
    if (!fp->is_write) { fp->errcode |= BGZF_ERR_MISUSE; return -1; }
    if (!fp->is_compressed) {                                                // bgzf.c:2004-2009
        const size_t push = length + (size_t)fp->block_offset;
        fp->block_offset = (int)(push % BGZF_MAX_BLOCK_SIZE);
        fp->block_address += (int64_t)(push - (size_t)fp->block_offset);
        return hg_hwrite(fp->fp, data, length);
    }
    const uint8_t *in = (const uint8_t *)data;
    size_t left = length;
    while (left) {
        if (fp->mt && fp->block_offset == 0 && left >= (size_t)BGZF_BLOCK_SIZE && E(fp)) {
            const ssize_t took = queue_whole_blocks(fp, in, left);
            if (took < 0) return -1;
            in += took; left -= (size_t)took;
            if (!left) break;
        }
        size_t n = (size_t)(BGZF_BLOCK_SIZE - fp->block_offset);
        if (n > left) n = left;
        memcpy((uint8_t *)fp->uncompressed_block + fp->block_offset, in, n);
        fp->block_offset += (int)n; in += n; left -= n;
        if (fp->block_offset == BGZF_BLOCK_SIZE && cut_block(fp) != 0) return -1;
    }
    return (ssize_t)length;
}

// bgzip -g / --reindex: cut the blocks where a loaded .gzi says the original file cut them (bgzf.c:2029-2060)
ssize_t bgzf_block_write(BGZF *fp, const void *data, size_t length) {
    if (!fp->is_compressed || !fp->idx || !fp->idx->loaded) return bgzf_write(fp, data, length);
    const uint8_t *in = (const uint8_t *)data;
    size_t left = length;
    bgzidx_t *ix = fp->idx;
    while (left) {
        size_t limit = BGZF_BLOCK_SIZE;
        if (ix->recut + 1 < ix->offs.size()) {
            const uint64_t sz = ix->offs[ix->recut + 1].uaddr - ix->offs[ix->recut].uaddr;
            if (sz >= 1 && sz <= BGZF_BLOCK_SIZE) limit = (size_t)sz;
        }
        size_t n = limit > (size_t)fp->block_offset ? limit - (size_t)fp->block_offset : 0;
        if (n > left) n = left;
        memcpy((uint8_t *)fp->uncompressed_block + fp->block_offset, in, n);
        fp->block_offset += (int)n; in += n; left -= n;
        if ((size_t)fp->block_offset >= limit) {
            if (cut_block(fp) != 0) return -1;
            ix->recut++;
        }
    }
    return (ssize_t)length;
}

int bgzf_flush_try(BGZF *fp, ssize_t size) {
    if (fp->block_offset + size > BGZF_BLOCK_SIZE) return fp->is_compressed ? cut_block(fp) : bgzf_flush(fp);
    return 0;
}

int bgzf_flush(BGZF *fp) {
    if (!fp->is_write) return 0;
    if (!fp->is_compressed) return hflush(fp->fp);
    if (!fp->mt && E(fp)->kind == K_WRITE) { if (host_cut_block(fp) != 0) return -1; }
    else if (queue_block(fp) != 0) return -1;
    return drain_writer(fp);
}

ssize_t bgzf_raw_write(BGZF *fp, const void *data, size_t length) {
    const ssize_t n = hg_hwrite(fp->fp, data, length);
    if (n < 0) fp->errcode |= BGZF_ERR_IO;
    return n;
}

int bgzf_idx_push(BGZF *fp, void *hidx, int tid, int64_t beg, int64_t end, uint64_t offset, int is_mapped) {
    Engine *e = E(fp);
    if (!e || !fp->is_write || !fp->mt) return hts_idx_push ? hts_idx_push(hidx, tid, beg, end, offset, is_mapped) : -1;
    std::lock_guard<std::mutex> g(e->idx_m);
    e->pushes.push_back(IdxPush{hidx, tid, beg, end, (uint32_t)(offset & 0xffff), is_mapped, e->block_number});
    return 0;
}

int bgzf_compress(void *dst, size_t *dlen, const void *src, size_t slen, int level) {
    if (slen == 0) {                                                         // the canonical EOF block (bgzf.c:563-569)
        if (*dlen < 28) return -1;
        memcpy(dst, kEof, 28); *dlen = 28;
        return 0;
    }
    // One block, synchronously: the host codec (SURVEY 8b keeps bgzf_compress() on the scalar path; a device job for one block is a 4 ms round trip).
    // The engine must exist all the same -- this library has no life without a device.
    if (!shared_ctx() || slen > BGZF_BLOCK_SIZE) return -1;
    hgh::Deflater &D = tls_deflater();
    return hgh::bgzf_block_deflate(D, (uint8_t *)dst, dlen, (const uint8_t *)src, slen, level < 0 || level > 9 ? 6 : level);
}

// Short buffers (the 10-30 byte block headers cram_read_block / cram_write_block checksum, cram_io.c:1431-1470, 1547-1552) are not worth a
// device round trip, and a process without a usable GPU must still get a CORRECT checksum rather than die: a byte-wise table CRC on the
// host for buffers below HOST_CRC_MAX and whenever the engine is unavailable; everything else on the device (hg_crc32_host).
static uint32_t crc32_table_host(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t T[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u))); T[0][i] = c; }
        for (uint32_t i = 0; i < 256; i++) for (int k = 1; k < 8; k++) T[k][i] = (T[k - 1][i] >> 8) ^ T[0][T[k - 1][i] & 0xff];
    });
    uint32_t c = ~crc;
    for (; n >= 8; n -= 8, p += 8) {
        const uint32_t a = c ^ ((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24);
        c = T[7][a & 0xff] ^ T[6][(a >> 8) & 0xff] ^ T[5][(a >> 16) & 0xff] ^ T[4][a >> 24] ^ T[3][p[4]] ^ T[2][p[5]] ^ T[1][p[6]] ^ T[0][p[7]];
    }
    for (; n; n--, p++) c = T[0][(c ^ *p) & 0xff] ^ (c >> 8);
    return ~c;
}
uint32_t hts_crc32(uint32_t crc, const void *buf, size_t len) {
    if (len == 0) return crc;
    constexpr size_t HOST_CRC_MAX = 4096;
    if (len < HOST_CRC_MAX) return crc32_table_host(crc, (const uint8_t *)buf, len);
    hg_ctx *ctx = shared_ctx();
    uint32_t c = 0;
    if (!ctx || hg_crc32_host(ctx, buf, len, &c) != HG_OK) {
        // HTS_GPU_STRICT=1: a deployment that must never compute on the host without noticing gets the hard stop back
        static const bool strict = [] { const char *v = getenv("HTS_GPU_STRICT"); return v && atoi(v) != 0; }();
        if (strict) { logmsg(LOG_ERROR, "hts_crc32", "no usable GPU engine and HTS_GPU_STRICT is set"); abort(); }
        static std::once_flag warned;
        std::call_once(warned, [] { logmsg(LOG_WARNING, "hts_crc32", "no usable GPU engine: checksums are computed on the host"); });
        return crc32_table_host(crc, (const uint8_t *)buf, len);
    }
    return crc_concat(crc, c, len);
}

// -------------------------------------------------------------------------------- .gzi index
int bgzf_index_build_init(BGZF *fp) {
    delete fp->idx;
    fp->idx = new bgzidx_t();
    if (fp->is_write) fp->idx->offs.push_back(GziEntry{0, 0});
    fp->idx_build_otf = 1;
    return 0;
}

int bgzf_index_dump_hfile(BGZF *fp, hFILE *out, const char *name) {
    if (!fp->idx) { logmsg(LOG_ERROR, "bgzf_index_dump_hfile", "Called for BGZF handle with no index"); errno = EINVAL; return -1; }
    if (bgzf_flush(fp) != 0) return -1;
    // writers hold one entry per block END, readers one per block START: the file lists every block start but the
    // first (bgzf.c:2385-2411), so a writer's last entry (the end of the file) is left out
    const std::vector<GziEntry> &o = fp->idx->offs;
    size_t n = o.size();
    if (fp->is_write && n > 0) n--;
    size_t first = !o.empty() && o[0].caddr == 0 && o[0].uaddr == 0 ? 1 : 0;
    if (first > n) first = n;
    const uint64_t cnt = n - first;
    bool ok = hg_hwrite(out, &cnt, 8) == 8;
    for (size_t i = first; ok && i < n; i++) ok = hg_hwrite(out, &o[i].caddr, 8) == 8 && hg_hwrite(out, &o[i].uaddr, 8) == 8;
    if (!ok) logmsg(LOG_ERROR, "bgzf_index_dump_hfile", "Error writing to %s : %s", name ? name : "index", strerror(errno));
    return ok ? 0 : -1;
}

int bgzf_index_dump(BGZF *fp, const char *bname, const char *suffix) {
    if (!fp->idx) { logmsg(LOG_ERROR, "bgzf_index_dump", "Called for BGZF handle with no index"); errno = EINVAL; return -1; }
    const std::string name = std::string(bname) + (suffix ? suffix : "");
    hFILE *out = hopen(name.c_str(), "wb");
    if (!out) { logmsg(LOG_ERROR, "bgzf_index_dump", "Error opening %s : %s", name.c_str(), strerror(errno)); return -1; }
    if (bgzf_index_dump_hfile(fp, out, name.c_str()) != 0) { hclose_abruptly(out); return -1; }
    if (hclose(out) < 0) { logmsg(LOG_ERROR, "bgzf_index_dump", "Error on closing %s : %s", name.c_str(), strerror(errno)); return -1; }
    return 0;
}

int bgzf_index_load_hfile(BGZF *fp, hFILE *in, const char *name) {
    bgzidx_t *ix = new bgzidx_t();
    uint64_t n = 0;
    bool ok = hg_hread(in, &n, 8) == 8 && n < (1ull << 40);
    if (ok) {
        ix->offs.assign(1, GziEntry{0, 0});
        for (uint64_t i = 0; ok && i < n; i++) {
            GziEntry x;
            ok = hg_hread(in, &x.caddr, 8) == 8 && hg_hread(in, &x.uaddr, 8) == 8;
            if (ok) ix->offs.push_back(x);
        }
    }
    if (!ok) {
        logmsg(LOG_ERROR, "bgzf_index_load_hfile", "Error reading %s : %s", name ? name : "index", strerror(errno));
        delete ix;
        return -1;
    }
    ix->loaded = true;
    delete fp->idx;
    fp->idx = ix;
    return 0;
}

int bgzf_index_load(BGZF *fp, const char *bname, const char *suffix) {
    const std::string name = std::string(bname) + (suffix ? suffix : "");
    hFILE *in = hopen(name.c_str(), "rb");
    if (!in) { logmsg(LOG_ERROR, "bgzf_index_load", "Error opening %s : %s", name.c_str(), strerror(errno)); return -1; }
    if (bgzf_index_load_hfile(fp, in, name.c_str()) != 0) { hclose_abruptly(in); return -1; }
    if (hclose(in) != 0) { logmsg(LOG_ERROR, "bgzf_index_load", "Error closing %s : %s", name.c_str(), strerror(errno)); return -1; }
    return 0;
}

}  // extern "C"

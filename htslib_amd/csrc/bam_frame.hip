// bam_frame.hip -- BAM record framing and base decoding on the device, for consumers that keep
// the inflated stream in HBM instead of pulling it through bgzf_read one record at a time.
//
// Replaces the framing half of bam_read1 (reference sam.c:784-866: block_len, the 32 core bytes, the size sanity
// checks) and nibble2base (simd.c:119-161, call site sam.c:1433).  Parity: oracle/bam_oracle.c.
//
// A BAM stream is a linked list: record i+1 starts where record i's block_len says.  The chain is cut into 64 KiB
// chunks; one LANE per chunk
//   1. guesses the first record start at or after its chunk boundary: the smallest offset from which three
//      consecutive records pass bam_read1's checks plus the value ranges a real record has (reference ids inside
//      the header's range, NUL-terminated name) -- the classic guess-and-verify of parallel BAM readers;
//   2. walks its chunk, counts the records that start inside it and reports where the chain leaves the chunk.
// The host then checks that every chunk's exit is the next chunk's entry (one 16-byte row per chunk) -- that makes
// the result exact, not probabilistic: a wrong guess shows up as a broken link and that chunk is re-walked from the
// proven entry -- prefix-sums the counts, and a second launch writes the offsets.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgb {

constexpr uint64_t NONE = ~0ull;

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// bam_read1's own checks (sam.c:797-829).  Returns 0 ok (and the next record's offset), -2 truncated, -4 invalid.
__device__ __forceinline__ int check_record(const uint8_t *b, uint64_t len, uint64_t p, uint64_t &next) {
    if (p + 4 > len) return -2;
    const int32_t bl = (int32_t)ld32(b + p);
    if (bl < 32) return -4;
    if (p + 4 + (uint64_t)bl > len) return -2;
    const uint8_t *x = b + p + 4;
    const uint32_t x2 = ld32(x + 8), x3 = ld32(x + 12);
    const uint32_t l_qname = x2 & 0xffu, n_cigar = x3 & 0xffffu;
    const int32_t l_qseq = (int32_t)ld32(x + 16);
    if (l_qseq < 0 || l_qname < 1) return -4;
    if (((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_qseq + 1) >> 1) + (uint64_t)l_qseq > (uint64_t)(bl - 32)) return -4;
    next = p + 4 + (uint64_t)bl;
    return 0;
}
// what a real record additionally looks like (only used to GUESS an entry point, never to accept or reject data)
__device__ __forceinline__ bool plausible(const uint8_t *b, uint64_t len, uint64_t p, int32_t n_ref, uint64_t &next) {
    if (check_record(b, len, p, next)) return false;
    const uint8_t *x = b + p + 4;
    const int32_t tid = (int32_t)ld32(x), pos = (int32_t)ld32(x + 4), mtid = (int32_t)ld32(x + 20), mpos = (int32_t)ld32(x + 24);
    const uint32_t l_qname = ld32(x + 8) & 0xffu;
    if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1) return false;
    return x[32 + l_qname - 1] == 0;
}

struct ChunkRow { uint64_t entry, exit; };                           // exit: chain position when it left the chunk

// mode 0: guess entries (except chunk 0 / forced ones) and count; mode 1: walk from the given entries and write offsets
__global__ __launch_bounds__(256)
void bam_chunk_kernel(const uint8_t *__restrict__ b, uint64_t len, uint64_t first, int32_t n_ref, uint32_t chunk_bytes,
                      uint32_t nchunks, const uint32_t *__restrict__ only, uint32_t nonly, ChunkRow *rows, uint32_t *count,
                      int32_t *err, const uint64_t *__restrict__ base, uint64_t *rec_off, uint64_t max_rec, int mode) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= (only ? nonly : nchunks)) return;
    const uint32_t c = only ? only[t] : t;
    const uint64_t lo = (uint64_t)c * chunk_bytes, hi = lo + chunk_bytes < len ? lo + chunk_bytes : len;
    uint64_t p = rows[c].entry;
    if (mode == 0 && !only) {
        if (c == 0) p = first;
        else {
            p = NONE;
            for (uint64_t q = lo; q < hi; q++) {                       // smallest offset that starts a 3-record chain
                uint64_t a, e2, e3;
                if (!plausible(b, len, q, n_ref, a)) continue;
                if (a < len && !plausible(b, len, a, n_ref, e2)) continue;
                if (a < len && e2 < len && !plausible(b, len, e2, n_ref, e3)) continue;
                p = q; break;
            }
        }
        rows[c].entry = p;
    }
    uint32_t n = 0; int e = 0;
    uint64_t k = base ? base[c] : 0;
    if (p != NONE) {
        while (p < hi) {
            uint64_t nx;
            e = check_record(b, len, p, nx);
            if (e) break;
            if (mode == 1 && k < max_rec) rec_off[k] = p;
            k++; n++;
            p = nx;
        }
    }
    rows[c].exit = p;                                                // on error: the offending record
    count[c] = n; err[c] = e;
}

// ---- bases: nibble2base over all records ------------------------------------------------------------------
__global__ __launch_bounds__(256)
void bam_lseq_kernel(const uint8_t *__restrict__ b, const uint64_t *__restrict__ rec_off, uint64_t n, uint32_t *lseq) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) lseq[i] = ld32(b + rec_off[i] + 4 + 16);
}
// one wavefront per record: 64 bases per step, a 16-entry lookup held in a register pair
__global__ __launch_bounds__(256)
void bam_bases_kernel(const uint8_t *__restrict__ b, const uint64_t *__restrict__ rec_off, uint64_t n,
                      const uint64_t *__restrict__ base_off, uint8_t *bases) {
    const int lane = threadIdx.x & 63;
    const uint64_t w0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * 256u) >> 6;
    const unsigned long long lut_lo = 0x5652474d43413dull, lut_hi = 0x4e42444b48595754ull;   // "=ACMGRSV" "TWYHKDBN"
    for (uint64_t i = w0; i < n; i += nw) {
        const uint8_t *x = b + rec_off[i] + 4;
        const uint32_t l_qname = ld32(x + 8) & 0xffu, n_cigar = ld32(x + 12) & 0xffffu, l_qseq = ld32(x + 16);
        const uint8_t *nib = x + 32 + l_qname + 4u * n_cigar;
        uint8_t *o = bases + base_off[i];
        for (uint32_t j = (uint32_t)lane; j < l_qseq; j += 64) {
            const uint32_t v = (nib[j >> 1] >> ((~j & 1u) << 2)) & 0xfu;
            o[j] = (uint8_t)(((v & 8u) ? lut_hi : lut_lo) >> ((v & 7u) * 8u));
        }
    }
}
// the fixed fields of every record as columns (bam1_core_t, sam.c:808-821): one thread per record
__global__ __launch_bounds__(256)
void bam_core_kernel(const uint8_t *__restrict__ b, const uint64_t *__restrict__ rec_off, uint64_t n, hg_bam_core_cols c) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint8_t *x = b + rec_off[i] + 4;
    const uint32_t x2 = ld32(x + 8), x3 = ld32(x + 12);
    if (c.tid) c.tid[i] = (int32_t)ld32(x);
    if (c.pos) c.pos[i] = (int32_t)ld32(x + 4);
    if (c.bin) c.bin[i] = (uint16_t)(x2 >> 16);
    if (c.mapq) c.mapq[i] = (uint8_t)(x2 >> 8);
    if (c.l_qname) c.l_qname[i] = (uint8_t)x2;
    if (c.flag) c.flag[i] = (uint16_t)(x3 >> 16);
    if (c.n_cigar) c.n_cigar[i] = (uint16_t)x3;
    if (c.l_qseq) c.l_qseq[i] = (int32_t)ld32(x + 16);
    if (c.mtid) c.mtid[i] = (int32_t)ld32(x + 20);
    if (c.mpos) c.mpos[i] = (int32_t)ld32(x + 24);
    if (c.isize) c.isize[i] = (int32_t)ld32(x + 28);
}
// qualities as Phred+33 text at the same offsets as the bases (0xff = absent stays 0xff, as sam_format1 prints '*')
__global__ __launch_bounds__(256)
void bam_quals_kernel(const uint8_t *__restrict__ b, const uint64_t *__restrict__ rec_off, uint64_t n,
                      const uint64_t *__restrict__ base_off, uint8_t *quals) {
    const int lane = threadIdx.x & 63;
    const uint64_t w0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * 256u) >> 6;
    for (uint64_t i = w0; i < n; i += nw) {
        const uint8_t *x = b + rec_off[i] + 4;
        const uint32_t l_qname = ld32(x + 8) & 0xffu, n_cigar = ld32(x + 12) & 0xffffu, l_qseq = ld32(x + 16);
        const uint8_t *q = x + 32 + l_qname + 4u * n_cigar + ((l_qseq + 1u) >> 1);
        uint8_t *o = quals + base_off[i];
        for (uint32_t j = (uint32_t)lane; j < l_qseq; j += 64) { const uint32_t v = q[j]; o[j] = (uint8_t)(v == 0xffu ? 0xffu : v + 33u); }
    }
}

// Exclusive prefix sum of n 32-bit values into 64-bit offsets (n+1 outputs), three launches: per-tile sums, a scan of
// the tile sums by one workgroup, then the tiles again with their base added.
constexpr uint32_t TILE = 4096;
__device__ __forceinline__ unsigned long long wg_incl_scan(unsigned long long x, unsigned long long *wsum, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long s = x;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(s, d, 64); if (lane >= d) s += y; }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    unsigned long long pre = 0;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    __syncthreads();
    return pre + s;
}
__global__ __launch_bounds__(256)
void scan_tiles_kernel(const uint32_t *__restrict__ v, uint64_t n, const uint64_t *__restrict__ tile_base, uint64_t *tile_sum, uint64_t *out) {
    __shared__ unsigned long long wsum[4];
    const int tid = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;
    unsigned long long carry = tile_base ? tile_base[blockIdx.x] : 0ull;
    for (uint32_t k = 0; k < TILE; k += 256) {
        const uint64_t i = t0 + k + (uint64_t)tid;
        const unsigned long long x = i < n ? v[i] : 0ull;
        const unsigned long long incl = wg_incl_scan(x, wsum, tid);
        if (out && i < n) out[i] = carry + incl - x;
        __shared__ unsigned long long tot;
        if (tid == 255) tot = incl;
        __syncthreads();
        carry += tot;
        __syncthreads();
    }
    if (tile_sum && tid == 0) tile_sum[blockIdx.x] = carry;
    if (out && tid == 0 && t0 + TILE >= n) out[n] = carry;
}
__global__ __launch_bounds__(1024)
void scan_sums_kernel(uint64_t *sums, uint64_t nt) {                   // in place: exclusive scan of the tile sums
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s, tot;
    const int tid = threadIdx.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nt; b0 += 1024) {
        const uint64_t i = b0 + (uint64_t)tid;
        const unsigned long long x = i < nt ? sums[i] : 0ull;
        const unsigned long long incl = wg_incl_scan(x, wsum, tid);
        const unsigned long long c = carry_s;
        if (i < nt) sums[i] = c + incl - x;
        if (tid == 1023) tot = incl;
        __syncthreads();
        if (tid == 0) carry_s = c + tot;
        __syncthreads();
    }
}

}  // namespace hgb

namespace hg {
// exclusive prefix sum of n 32-bit values into 64-bit offsets (n + 1 outputs); uses scratch slot 14
int scan32_to64(hg_ctx *ctx, const uint32_t *d_v, uint64_t n, uint64_t *d_out, hipStream_t s) {
    if (!n) return hipMemsetAsync(d_out, 0, 8, s) == hipSuccess ? HG_OK : HG_ELAUNCH;
    const uint64_t nt = (n + hgb::TILE - 1) / hgb::TILE;
    if (int rc = ensure_scratch(ctx, 14, (size_t)nt * 8 + 64)) return rc;
    uint64_t *d_ts = (uint64_t *)ctx->d_scratch[14];
    hipLaunchKernelGGL(hgb::scan_tiles_kernel, dim3((unsigned)nt), dim3(256), 0, s, d_v, n, (const uint64_t *)nullptr, d_ts, (uint64_t *)nullptr);
    hipLaunchKernelGGL(hgb::scan_sums_kernel, dim3(1), dim3(1024), 0, s, d_ts, nt);
    hipLaunchKernelGGL(hgb::scan_tiles_kernel, dim3((unsigned)nt), dim3(256), 0, s, d_v, n, (const uint64_t *)d_ts, (uint64_t *)nullptr, d_out);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

extern "C" {

int hg_bam_header_host(const uint8_t *bam, size_t len, int32_t *n_ref, uint64_t *first_record_off) {
    if (!bam || !n_ref || !first_record_off) return HG_EINVAL;
    if (len < 12 || memcmp(bam, "BAM\1", 4)) return HG_EFORMAT;
    auto le = [&](uint64_t p) { return (uint32_t)bam[p] | ((uint32_t)bam[p + 1] << 8) | ((uint32_t)bam[p + 2] << 16) | ((uint32_t)bam[p + 3] << 24); };
    uint64_t p = 8 + (uint64_t)le(4);
    if (p + 4 > len) return HG_EFORMAT;
    const int32_t n = (int32_t)le(p); p += 4;
    if (n < 0) return HG_EFORMAT;
    for (int32_t i = 0; i < n; i++) {
        if (p + 4 > len) return HG_EFORMAT;
        const int32_t l = (int32_t)le(p); p += 4;
        if (l <= 0 || p + (uint64_t)l + 4 > len) return HG_EFORMAT;
        p += (uint64_t)l + 4;
    }
    *n_ref = n; *first_record_off = p;
    return HG_OK;
}

long hg_bam_frame_dev(hg_ctx *ctx, const void *d_bam, uint64_t len, uint64_t first_record_off, int32_t n_ref, uint64_t *d_rec_off,
                      uint64_t max_rec, uint64_t *bad_off, void *stream) {
    if (!ctx || !d_bam || first_record_off > len) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t CH = 1u << 16;
    const uint64_t nch64 = (len + CH - 1) / CH;
    if (nch64 == 0) return 0;
    if (nch64 > 0x7fffffffull) return HG_EINVAL;
    const uint32_t nch = (uint32_t)nch64;
    int rc;
    if ((rc = hg::ensure_scratch(ctx, 8, (size_t)nch * sizeof(hgb::ChunkRow) + 64)) || (rc = hg::ensure_scratch(ctx, 9, (size_t)nch * 8 + 64)) ||
        (rc = hg::ensure_scratch(ctx, 10, (size_t)nch * 8 + 64)) || (rc = hg::ensure_scratch(ctx, 11, (size_t)nch * 4 + 64))) return (long)rc;
    hgb::ChunkRow *d_rows = (hgb::ChunkRow *)ctx->d_scratch[8];
    uint32_t *d_count = (uint32_t *)ctx->d_scratch[9]; int32_t *d_err = (int32_t *)(d_count + nch);
    uint64_t *d_base = (uint64_t *)ctx->d_scratch[10];
    uint32_t *d_only = (uint32_t *)ctx->d_scratch[11];
    const uint8_t *b = (const uint8_t *)d_bam;
    std::vector<hgb::ChunkRow> rows(nch); std::vector<uint32_t> cnt(nch); std::vector<int32_t> err(nch);
    auto fetch = [&]() {
        return hipMemcpyAsync(rows.data(), d_rows, (size_t)nch * sizeof(hgb::ChunkRow), hipMemcpyDeviceToHost, s) == hipSuccess &&
               hipMemcpyAsync(cnt.data(), d_count, (size_t)nch * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
               hipMemcpyAsync(err.data(), d_err, (size_t)nch * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    };
    hipLaunchKernelGGL(hgb::bam_chunk_kernel, dim3((nch + 255) / 256), dim3(256), 0, s, b, len, first_record_off, n_ref, CH, nch,
                       (const uint32_t *)nullptr, 0u, d_rows, d_count, d_err, (const uint64_t *)nullptr, (uint64_t *)nullptr, 0ull, 0);
    if (hipGetLastError() != hipSuccess || !fetch()) return HG_ELAUNCH;
    // ---- link check: the chain must enter every chunk where the previous one left it -----------------------
    uint64_t expect = first_record_off; long result = 0; uint64_t total = 0;
    std::vector<uint64_t> base(nch);
    for (uint32_t c = 0; c < nch; c++) {
        const uint64_t lo = (uint64_t)c * CH, hi = lo + CH < len ? lo + CH : len;
        const uint64_t want = expect < hi ? expect : hgb::NONE;       // NONE: a long record covers the whole chunk
        if (rows[c].entry != want) {                                   // wrong guess: walk this chunk from the proven entry
            hgb::ChunkRow r = {want, want};
            if (hipMemcpyAsync(d_rows + c, &r, sizeof r, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipMemcpyAsync(d_only, &c, 4, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
            hipLaunchKernelGGL(hgb::bam_chunk_kernel, dim3(1), dim3(256), 0, s, b, len, first_record_off, n_ref, CH, nch, (const uint32_t *)d_only, 1u,
                               d_rows, d_count, d_err, (const uint64_t *)nullptr, (uint64_t *)nullptr, 0ull, 0);
            if (hipMemcpyAsync(&rows[c], d_rows + c, sizeof r, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(&cnt[c], d_count + c, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(&err[c], d_err + c, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        }
        base[c] = total; total += cnt[c];
        if (err[c]) { if (bad_off) *bad_off = rows[c].exit; result = err[c] == -2 ? HG_BAM_ETRUNC : HG_BAM_EINVALID; break; }
        if (want != hgb::NONE) expect = rows[c].exit;
    }
    if (result < 0) return result;
    if (d_rec_off && total) {
        if (hipMemcpyAsync(d_base, base.data(), (size_t)nch * 8, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
        hipLaunchKernelGGL(hgb::bam_chunk_kernel, dim3((nch + 255) / 256), dim3(256), 0, s, b, len, first_record_off, n_ref, CH, nch,
                           (const uint32_t *)nullptr, 0u, d_rows, d_count, d_err, (const uint64_t *)d_base, d_rec_off, max_rec, 1);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    }
    return (long)total;
}

int hg_bam_core_dev(hg_ctx *ctx, const void *d_bam, const uint64_t *d_rec_off, uint64_t n, const hg_bam_core_cols *cols, void *stream) {
    if (!ctx || !cols || (n && (!d_bam || !d_rec_off))) return HG_EINVAL;
    if (!n) return HG_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    hipLaunchKernelGGL(hgb::bam_core_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_bam, d_rec_off, n, *cols);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

int hg_bam_quals_dev(hg_ctx *ctx, const void *d_bam, const uint64_t *d_rec_off, uint64_t n, const uint64_t *d_base_off, void *d_quals, void *stream) {
    if (!ctx || (n && (!d_bam || !d_rec_off || !d_base_off || !d_quals))) return HG_EINVAL;
    if (!n) return HG_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    size_t wgs = (size_t)((n + 3) / 4);
    const size_t maxw = (size_t)ctx->cus * 32;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgb::bam_quals_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_bam, d_rec_off, n, d_base_off, (uint8_t *)d_quals);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

int hg_bam_bases_dev(hg_ctx *ctx, const void *d_bam, const uint64_t *d_rec_off, uint64_t n, uint64_t *d_base_off, void *d_bases,
                     uint64_t bases_cap, uint64_t *total_bases, void *stream) {
    if (!ctx || (n && (!d_bam || !d_rec_off || !d_base_off))) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hipStream_t s = (hipStream_t)stream;
    uint64_t total = 0;
    if (n) {
        int rc;
        if ((rc = hg::ensure_scratch(ctx, 8, (size_t)n * 4 + 64))) return rc;
        uint32_t *d_lseq = (uint32_t *)ctx->d_scratch[8];
        hipLaunchKernelGGL(hgb::bam_lseq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint8_t *)d_bam, d_rec_off, n, d_lseq);
        if ((rc = hg::scan32_to64(ctx, d_lseq, n, d_base_off, s))) return rc;
        if (hipMemcpyAsync(&total, d_base_off + n, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        if (d_bases && total) {
            if (total > bases_cap) { if (total_bases) *total_bases = total; return HG_EINVAL; }
            size_t wgs = (size_t)((n + 3) / 4);
            const size_t maxw = (size_t)ctx->cus * 32;
            if (wgs > maxw) wgs = maxw;
            hipLaunchKernelGGL(hgb::bam_bases_kernel, dim3((unsigned)wgs), dim3(256), 0, s, (const uint8_t *)d_bam, d_rec_off, n,
                               (const uint64_t *)d_base_off, (uint8_t *)d_bases);
            if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
        }
    }
    if (total_bases) *total_bases = total;
    return HG_OK;
}

}  // extern "C"

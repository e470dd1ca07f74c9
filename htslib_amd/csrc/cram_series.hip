// cram_series.hip -- CRAM integer data series <-> EXTERNAL blocks on MI355X (gfx950): whole-block ITF8 decode and encode.
// SURVEY §8f N2, first step: the per-record loop of cram_decode_slice pulls one integer at a time out of an EXTERNAL block
// (cram_external_decode_int, cram/cram_codecs.c:350-368, through safe_itf8_get, cram/cram_io.c:644-673); here a block becomes an
// int32 column in one pass, and back (cram_external_encode_int -> itf8_put, cram_codecs.c:523-527, cram_io.c:277-305).
//
// Decode.  Where a value starts depends on the length of every value before it (the length is in the first byte: 0xxxxxxx 1,
// 10xxxxxx 2, 110xxxxx 3, 1110xxxx 4, 1111xxxx 5 bytes), so the chain of starts is serial -- but it forgets quickly.  One
// wavefront takes a tile of 4 KiB, lane l the 64 bytes [64 l, 64 l + 64):
//   1. the lane turns its bytes into four 64-bit masks (first-byte class of every position) with SWAR compares -- the walks
//      below are then pure register arithmetic;
//   2. it walks its segment from each of the five possible entry offsets 0..4 (a walk that lands on a position an earlier walk
//      visited MERGES with it, so the five cost little more than one) and keeps, per entry, the exit offset into the next
//      segment, the number of values and the mask of value starts;
//   3. the per-lane maps entry -> exit (5 x 3 bits) are composed across the wavefront with a shuffle scan: every lane learns
//      its true entry, the last lane's exit is the entry of the next tile;
//   4. value counts are prefix-summed, every lane decodes the values at ITS starts (bytes re-read from the LDS copy of the
//      tile, which also serves values that straddle segments) into an LDS staging area, and the 64 lanes write the staged
//      values out coalesced.
// A value cut by the end of the block is the reference's error (safe_itf8_get sets *err): status -1.
//
// Encode.  64 values per step: byte count per value, wave prefix sum, every lane stores its 1..5 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgc {

constexpr int WAVES = 2;
constexpr uint32_t SEGB = 64, TILE = 64 * SEGB;
constexpr uint32_t ROW = 68;                        // LDS bytes per segment row: 17 dwords, odd -> no bank conflicts between lanes
struct WaveLds {
    uint32_t seg32[(65 * ROW + 16) / 4];            // the tile (+ 4 bytes of the next one), segment rows of ROW bytes
    int32_t vals[TILE];                             // decoded values of the tile, in order
};

__device__ __forceinline__ uint32_t tile_addr(uint32_t i) { return (i >> 6) * ROW + (i & 63u); }

// 3-bit fields: map entry e (0..4) -> get(f, e)
__device__ __forceinline__ uint32_t mget(uint32_t f, uint32_t e) { return (f >> (3u * e)) & 7u; }
__device__ __forceinline__ uint32_t mcompose(uint32_t first, uint32_t then) {      // e -> then(first(e))
    uint32_t r = 0;
#pragma unroll
    for (uint32_t e = 0; e < 5; e++) r |= mget(then, mget(first, e)) << (3u * e);
    return r;
}
constexpr uint32_t MAP_ID = 0 | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12);

__global__ __launch_bounds__(WAVES * 64)
void itf8_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, uint32_t n, int32_t *out,
                        uint32_t *count, int32_t *status) {
    __shared__ WaveLds lds[WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    WaveLds &S = lds[wv];
    uint8_t *seg8 = (uint8_t *)S.seg32;
    for (uint32_t k = blockIdx.x * WAVES + wv; k < n; k += gridDim.x * WAVES) {
        const hg_stream_desc d = desc[k];
        const uint8_t *src = in + d.in_off;
        int32_t *dst = out + d.out_off;
        const uint32_t len = d.in_len, cap = d.out_len;
        uint32_t entry = 0, nout = 0;
        int err = 0;
        for (uint32_t t0 = 0; t0 < len && !err; t0 += TILE) {
            // ---- stage the tile: 64 bytes per lane (+ the first 4 bytes of the next tile by lanes 0..3); zeros behind the end
            {
                const uint32_t base = t0 + (uint32_t)lane * SEGB;
                uint32_t w[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t p = base + 4u * (uint32_t)q;
                    uint32_t v = 0;
                    if (p + 4u <= len) __builtin_memcpy(&v, src + p, 4);
                    else for (uint32_t b = 0; b < 4 && p + b < len; b++) v |= (uint32_t)src[p + b] << (8u * b);
                    w[q] = v;
                }
#pragma unroll
                for (int q = 0; q < 16; q++) S.seg32[((uint32_t)lane * ROW) / 4 + q] = w[q];
                if (lane < 4) { const uint32_t p = t0 + TILE + (uint32_t)lane; seg8[64u * ROW + (uint32_t)lane] = p < len ? src[p] : 0; }
                // ---- first-byte classes of my 64 bytes as bit masks: C2 (10xxxxxx), C3, C4, C5; everything else is a 1-byte value
                unsigned long long c2 = 0, c3 = 0, c4 = 0, c5 = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t x = (w[q] >> (8 * b)) & 0xf0u;
                        const unsigned long long bit = 1ull << (4 * q + b);
                        if (x >= 0xf0u) c5 |= bit; else if (x >= 0xe0u) c4 |= bit; else if (x >= 0xc0u) c3 |= bit; else if (x >= 0x80u) c2 |= bit;
                    }
                }
                hg::wave_sync();
                const uint32_t lim_t = len - t0 < TILE ? len - t0 : TILE;              // bytes of the tile that exist
                const uint32_t seg0 = (uint32_t)lane * SEGB;
                const uint32_t lim_s = lim_t > seg0 ? (lim_t - seg0 < SEGB ? lim_t - seg0 : SEGB) : 0u;   // ... of my segment
                const unsigned long long live = lim_s >= 64u ? ~0ull : (1ull << lim_s) - 1ull;
                auto vlen = [&](uint32_t p) -> uint32_t {
                    const unsigned long long bit = 1ull << p;
                    return (c2 & bit) ? 2u : (c3 & bit) ? 3u : (c4 & bit) ? 4u : (c5 & bit) ? 5u : 1u;
                };
                // ---- the five walks
                unsigned long long starts[5];
                uint32_t fmap = 0;
#pragma unroll
                for (uint32_t e = 0; e < 5; e++) {
                    unsigned long long m = 0;
                    uint32_t p = e, ex = 0;
                    bool merged = false;
                    while (p < SEGB) {
#pragma unroll
                        for (uint32_t g = 0; g < e; g++) {
                            if (!merged && ((starts[g] >> p) & 1ull)) {             // joins walk g here
                                m |= starts[g] & ~((1ull << p) - 1ull);
                                ex = mget(fmap, g); merged = true;
                            }
                        }
                        if (merged) break;
                        m |= 1ull << p;
                        p += vlen(p);
                    }
                    if (!merged) ex = p - SEGB;
                    starts[e] = m;
                    fmap |= ex << (3u * e);
                }
                // ---- compose the maps: incl = f_lane o ... o f_0; my entry = (f_{lane-1} o ... o f_0)(entry of the tile)
                uint32_t incl = fmap;
                for (int s = 1; s < 64; s <<= 1) {
                    const uint32_t prev = (uint32_t)__shfl_up((int)incl, s, 64);
                    if (lane >= s) incl = mcompose(prev, incl);
                }
                uint32_t excl = (uint32_t)__shfl_up((int)incl, 1, 64);
                if (lane == 0) excl = MAP_ID;
                const uint32_t my_entry = mget(excl, entry);
                const uint32_t next_entry = mget((uint32_t)__shfl((int)incl, 63, 64), entry);
                unsigned long long mine = starts[0];
#pragma unroll
                for (uint32_t e = 1; e < 5; e++) mine = my_entry == e ? starts[e] : mine;
                mine &= live;                                                         // starts behind the end are not values
                // ---- counts, error check (a value that runs over the end), decode into the staging area
                const uint32_t cnt = (uint32_t)__popcll(mine);
                uint32_t pre = hg::wave_incl_scan_dpp(cnt);
                const uint32_t total = (uint32_t)__shfl((int)pre, 63, 64);
                pre -= cnt;
                uint32_t j = pre;
                for (unsigned long long w2 = mine; w2; w2 &= w2 - 1ull, j++) {
                    const uint32_t p = (uint32_t)__builtin_ctzll(w2), L = vlen(p), a = seg0 + p;
                    if (a + L > lim_t && t0 + a + L > len) err = 1;
                    const uint32_t b0 = seg8[tile_addr(a)];
                    uint32_t v = b0;
                    if (L == 2) v = ((b0 << 8) | seg8[tile_addr(a + 1)]) & 0x3fffu;
                    else if (L == 3) v = ((b0 << 16) | ((uint32_t)seg8[tile_addr(a + 1)] << 8) | seg8[tile_addr(a + 2)]) & 0x1fffffu;
                    else if (L == 4) v = ((b0 << 24) | ((uint32_t)seg8[tile_addr(a + 1)] << 16) | ((uint32_t)seg8[tile_addr(a + 2)] << 8) | seg8[tile_addr(a + 3)]) & 0x0fffffffu;
                    else if (L == 5) v = ((b0 & 0x0fu) << 28) | ((uint32_t)seg8[tile_addr(a + 1)] << 20) | ((uint32_t)seg8[tile_addr(a + 2)] << 12) |
                                         ((uint32_t)seg8[tile_addr(a + 3)] << 4) | (seg8[tile_addr(a + 4)] & 0x0fu);
                    S.vals[j] = (int32_t)v;
                }
                err = __any(err) ? 1 : 0;
                if (!err && nout + total > cap) err = 1;                              // more values than the caller has room for
                hg::wave_sync();
                if (!err) for (uint32_t q = (uint32_t)lane; q < total; q += 64) dst[nout + q] = S.vals[q];
                nout += total;
                entry = next_entry;
                hg::wave_sync();
            }
        }
        count[k] = err ? 0u : nout;                                                   // every lane stores the same word
        status[k] = err ? -1 : 0;
    }
}

__device__ __forceinline__ uint32_t itf8_len(uint32_t v) {
    return !(v & ~0x7fu) ? 1u : !(v & ~0x3fffu) ? 2u : !(v & ~0x1fffffu) ? 3u : !(v & ~0x0fffffffu) ? 4u : 5u;
}

__global__ __launch_bounds__(WAVES * 64)
void itf8_encode_kernel(const int32_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, uint32_t n, uint8_t *out,
                        uint32_t *out_len, int32_t *status) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t k = blockIdx.x * WAVES + wv; k < n; k += gridDim.x * WAVES) {
        const hg_stream_desc d = desc[k];
        const int32_t *src = in + d.in_off;
        uint8_t *dst = out + d.out_off;
        unsigned long long o = 0;
        int err = 0;
        for (uint32_t i0 = 0; i0 < d.in_len; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool has = i < d.in_len;
            const uint32_t v = has ? (uint32_t)src[i] : 0u;
            const uint32_t L = has ? itf8_len(v) : 0u;
            uint32_t pre = hg::wave_incl_scan_dpp(L);
            const uint32_t total = (uint32_t)__shfl((int)pre, 63, 64);
            pre -= L;
            if (o + total > d.out_len) { err = 1; break; }
            uint8_t *e = dst + o + pre;
            if (L == 1) e[0] = (uint8_t)v;
            else if (L == 2) { e[0] = (uint8_t)((v >> 8) | 0x80u); e[1] = (uint8_t)v; }
            else if (L == 3) { e[0] = (uint8_t)((v >> 16) | 0xc0u); e[1] = (uint8_t)(v >> 8); e[2] = (uint8_t)v; }
            else if (L == 4) { e[0] = (uint8_t)((v >> 24) | 0xe0u); e[1] = (uint8_t)(v >> 16); e[2] = (uint8_t)(v >> 8); e[3] = (uint8_t)v; }
            else if (L == 5) { e[0] = (uint8_t)(0xf0u | (v >> 28)); e[1] = (uint8_t)(v >> 20); e[2] = (uint8_t)(v >> 12); e[3] = (uint8_t)(v >> 4); e[4] = (uint8_t)(v & 0x0fu); }
            o += total;
        }
        out_len[k] = err ? 0u : (uint32_t)o;                                          // every lane stores the same word
        status[k] = err ? -1 : 0;
    }
}

// BYTE_ARRAY_STOP series (read names, string tags; cram_byte_array_stop_decode_char, cram/cram_codecs.c:3586-3624): the items of a
// block are separated by a stop byte.  One wavefront per block, 1 KiB per step: every lane compares its 16 bytes with the stop byte
// (a 16-bit mask), a wave prefix sum numbers the stops, and the lane writes for every stop the offset of the item that FOLLOWS it --
// off[0] = 0, off[k + 1] = position of the k-th stop + 1; off[count] is the end of the last complete item.  Bytes behind the last
// stop are an unterminated item: the reference's -1.
__global__ __launch_bounds__(WAVES * 64)
void byte_array_stop_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, uint32_t n, uint32_t *out,
                            uint32_t *count, int32_t *status) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t k = blockIdx.x * WAVES + wv; k < n; k += gridDim.x * WAVES) {
        const hg_stream_desc d = desc[k];
        const uint8_t *src = in + d.in_off;
        uint32_t *dst = out + d.out_off;
        const uint32_t len = d.in_len, cap = d.out_len, stop = d.reserved & 0xffu;
        uint32_t nitems = 0, last_end = 0;
        int err = cap < 1 ? 1 : 0;
        if (!err && lane == 0) dst[0] = 0;
        for (uint32_t t0 = 0; t0 < len && !err; t0 += 1024u) {
            const uint32_t p = t0 + 16u * (uint32_t)lane;
            uint32_t w[4] = {0, 0, 0, 0}, valid = 0;
            if (p + 16u <= len) { uint4 v; __builtin_memcpy(&v, src + p, 16); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; valid = 16; }
            else if (p < len) { valid = len - p; for (uint32_t b = 0; b < valid; b++) w[b >> 2] |= (uint32_t)src[p + b] << (8u * (b & 3u)); }
            uint32_t m = 0;
#pragma unroll
            for (uint32_t b = 0; b < 16; b++) if (b < valid && ((w[b >> 2] >> (8u * (b & 3u))) & 0xffu) == stop) m |= 1u << b;
            const uint32_t c = (uint32_t)__popc(m);
            uint32_t pre = hg::wave_incl_scan_dpp(c);
            const uint32_t total = (uint32_t)__shfl((int)pre, 63, 64);
            pre -= c;
            if (nitems + total + 1u > cap) { err = 1; break; }
            uint32_t j = nitems + pre;
            for (uint32_t mm = m; mm; mm &= mm - 1u) { j++; dst[j] = p + (uint32_t)__builtin_ctz(mm) + 1u; }
            const uint32_t my_last = m ? p + 32u - (uint32_t)__builtin_clz(m) : 0u;           // end of my last complete item
            uint32_t le = my_last;
            for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)le, sft, 64); le = o > le ? o : le; }
            if (le > last_end) last_end = le;
            nitems += total;
        }
        if (!err && last_end != len) err = 1;                                                // unterminated bytes at the end
        count[k] = err ? 0u : nitems;                                                        // every lane stores the same word
        status[k] = err ? -1 : 0;
    }
}

}  // namespace hgc

using hg::ensure_scratch;

extern "C" {

// d_desc[i]: in_off / in_len = the block's bytes, out_off = first int32 of its column in d_out (in VALUES), out_len = room
// (values).  d_count[i] = values decoded; d_status[i] = 0, or -1 for a value cut by the end of the block / no room.
int hg_cram_itf8_decode_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, int32_t *d_out, uint32_t *d_count,
                            int32_t *d_status, void *stream) {
    if (!ctx || (n && (!d_in || !d_desc || !d_out || !d_count || !d_status))) return HG_EINVAL;
    if (!n) return HG_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    size_t wgs = (n + hgc::WAVES - 1) / hgc::WAVES;
    const size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgc::itf8_decode_kernel, dim3((unsigned)wgs), dim3(hgc::WAVES * 64), 0, (hipStream_t)stream, (const uint8_t *)d_in, d_desc,
                       (uint32_t)n, d_out, d_count, d_status);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

// d_desc[i]: in_off / in_len = first value / number of values in d_in (int32), out_off / out_len = where the bytes go and the room
// there (5 bytes per value always suffice).  d_out_len[i] = bytes written; d_status[i] = 0 / -1 (no room).
int hg_cram_itf8_encode_dev(hg_ctx *ctx, const int32_t *d_in, const hg_stream_desc *d_desc, size_t n, void *d_out, uint32_t *d_out_len,
                            int32_t *d_status, void *stream) {
    if (!ctx || (n && (!d_in || !d_desc || !d_out || !d_out_len || !d_status))) return HG_EINVAL;
    if (!n) return HG_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    size_t wgs = (n + hgc::WAVES - 1) / hgc::WAVES;
    const size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgc::itf8_encode_kernel, dim3((unsigned)wgs), dim3(hgc::WAVES * 64), 0, (hipStream_t)stream, d_in, d_desc, (uint32_t)n,
                       (uint8_t *)d_out, d_out_len, d_status);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

// d_desc[i]: in_off / in_len = the block; reserved = the stop byte; out_off = first uint32 of its offset table in d_off (in WORDS),
// out_len = room there (words; items + 1 are written).  d_off[out_off + k] = offset of item k inside the block, [count] = end of
// the last item + 1 (= in_len).  d_status[i] = 0, or -1 for bytes behind the last stop byte / no room.
int hg_cram_byte_array_stop_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, uint32_t *d_off, uint32_t *d_count,
                                int32_t *d_status, void *stream) {
    if (!ctx || (n && (!d_in || !d_desc || !d_off || !d_count || !d_status))) return HG_EINVAL;
    if (!n) return HG_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HG_ENODEV;
    size_t wgs = (n + hgc::WAVES - 1) / hgc::WAVES;
    const size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgc::byte_array_stop_kernel, dim3((unsigned)wgs), dim3(hgc::WAVES * 64), 0, (hipStream_t)stream, (const uint8_t *)d_in, d_desc,
                       (uint32_t)n, d_off, d_count, d_status);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

int hg_cram_byte_array_stop_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *stop, size_t n, uint32_t *const *off,
                                 const uint32_t *cap, uint32_t *count, int32_t *status) {
    if (!ctx || (n && (!in || !in_len || !stop || !off || !cap || !count || !status))) return HG_EINVAL;
    if (!n) return HG_OK;
    hg::CtxGuard guard_(ctx);
    if (guard_.rc) return guard_.rc;
    std::vector<hg_stream_desc> d(n);
    std::vector<uint64_t> ioff(n), ooff(n);
    std::vector<uint32_t> olen(n);
    uint64_t ib = 0, ov = 0;
    for (size_t i = 0; i < n; i++) {
        memset(&d[i], 0, sizeof d[i]);
        d[i].in_off = ib; d[i].in_len = in_len[i]; d[i].out_off = ov; d[i].out_len = cap[i]; d[i].reserved = stop[i];
        ioff[i] = ib; ooff[i] = ov * 4;
        ib += ((uint64_t)in_len[i] + 15u) & ~15ull; ov += ((uint64_t)cap[i] + 3u) & ~3ull;
    }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, ib + 64)) || (rc = ensure_scratch(ctx, 1, ov * 4 + 64)) || (rc = ensure_scratch(ctx, 2, n * sizeof(hg_stream_desc) + 64)) ||
        (rc = ensure_scratch(ctx, 3, n * 8 + 64))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = hg::stage_upload(ctx, in, in_len, ioff.data(), nullptr, n, ib, (uint8_t *)ctx->d_scratch[0], s)) != HG_OK) return rc;
    uint32_t *d_cnt = (uint32_t *)ctx->d_scratch[3]; int32_t *d_st = (int32_t *)ctx->d_scratch[3] + n;
    if (hipMemcpyAsync(ctx->d_scratch[2], d.data(), n * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    if ((rc = hg_cram_byte_array_stop_dev(ctx, ctx->d_scratch[0], (const hg_stream_desc *)ctx->d_scratch[2], n, (uint32_t *)ctx->d_scratch[1], d_cnt, d_st, s)) != HG_OK) return rc;
    if (hipMemcpyAsync(count, d_cnt, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(status, d_st, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    bool bad = false;
    for (size_t i = 0; i < n; i++) { olen[i] = status[i] == 0 ? (count[i] + 1u) * 4u : 0u; bad |= status[i] != 0; }
    std::vector<uint8_t *> dst(n);
    for (size_t i = 0; i < n; i++) dst[i] = (uint8_t *)off[i];
    if ((rc = hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], ooff.data(), olen.data(), dst.data(), n, s)) != HG_OK) return rc;
    return bad ? HG_EBLOCK : HG_OK;
}

// Host-buffer forms: n blocks in, n int32 columns out (cap[i] values of room each), one device round trip.
int hg_cram_itf8_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n, int32_t *const *out, const uint32_t *cap,
                             uint32_t *count, int32_t *status) {
    if (!ctx || (n && (!in || !in_len || !out || !cap || !count || !status))) return HG_EINVAL;
    if (!n) return HG_OK;
    hg::CtxGuard guard_(ctx);
    if (guard_.rc) return guard_.rc;
    std::vector<hg_stream_desc> d(n);
    std::vector<uint64_t> ioff(n), ooff(n);
    std::vector<uint32_t> olen(n);
    uint64_t ib = 0, ov = 0;
    for (size_t i = 0; i < n; i++) {
        memset(&d[i], 0, sizeof d[i]);
        d[i].in_off = ib; d[i].in_len = in_len[i]; d[i].out_off = ov; d[i].out_len = cap[i];
        ioff[i] = ib; ooff[i] = ov * 4;
        ib += ((uint64_t)in_len[i] + 15u) & ~15ull; ov += ((uint64_t)cap[i] + 3u) & ~3ull;
    }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, ib + 64)) || (rc = ensure_scratch(ctx, 1, ov * 4 + 64)) || (rc = ensure_scratch(ctx, 2, n * sizeof(hg_stream_desc) + 64)) ||
        (rc = ensure_scratch(ctx, 3, n * 8 + 64))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = hg::stage_upload(ctx, in, in_len, ioff.data(), nullptr, n, ib, (uint8_t *)ctx->d_scratch[0], s)) != HG_OK) return rc;
    uint32_t *d_cnt = (uint32_t *)ctx->d_scratch[3]; int32_t *d_st = (int32_t *)ctx->d_scratch[3] + n;
    if (hipMemcpyAsync(ctx->d_scratch[2], d.data(), n * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    if ((rc = hg_cram_itf8_decode_dev(ctx, ctx->d_scratch[0], (const hg_stream_desc *)ctx->d_scratch[2], n, (int32_t *)ctx->d_scratch[1], d_cnt, d_st, s)) != HG_OK) return rc;
    if (hipMemcpyAsync(count, d_cnt, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(status, d_st, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    bool bad = false;
    for (size_t i = 0; i < n; i++) { olen[i] = status[i] == 0 ? count[i] * 4u : 0u; bad |= status[i] != 0; }
    std::vector<uint8_t *> dst(n);
    for (size_t i = 0; i < n; i++) dst[i] = (uint8_t *)out[i];
    if ((rc = hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], ooff.data(), olen.data(), dst.data(), n, s)) != HG_OK) return rc;
    return bad ? HG_EBLOCK : HG_OK;
}

int hg_cram_itf8_encode_host(hg_ctx *ctx, const int32_t *const *in, const uint32_t *nvals, size_t n, uint8_t *const *out, const uint32_t *cap,
                             uint32_t *out_len, int32_t *status) {
    if (!ctx || (n && (!in || !nvals || !out || !cap || !out_len || !status))) return HG_EINVAL;
    if (!n) return HG_OK;
    hg::CtxGuard guard_(ctx);
    if (guard_.rc) return guard_.rc;
    std::vector<hg_stream_desc> d(n);
    std::vector<uint64_t> ioff(n), ooff(n);
    std::vector<uint32_t> ibytes(n);
    std::vector<const uint8_t *> src(n);
    uint64_t iv = 0, ob = 0;
    for (size_t i = 0; i < n; i++) {
        if (nvals[i] > 0x3fffffffu) return HG_EINVAL;
        memset(&d[i], 0, sizeof d[i]);
        d[i].in_off = iv; d[i].in_len = nvals[i]; d[i].out_off = ob; d[i].out_len = cap[i];
        ioff[i] = iv * 4; ooff[i] = ob; ibytes[i] = nvals[i] * 4u; src[i] = (const uint8_t *)in[i];
        iv += ((uint64_t)nvals[i] + 3u) & ~3ull; ob += ((uint64_t)cap[i] + 15u) & ~15ull;
    }
    int rc;
    if ((rc = ensure_scratch(ctx, 0, iv * 4 + 64)) || (rc = ensure_scratch(ctx, 1, ob + 64)) || (rc = ensure_scratch(ctx, 2, n * sizeof(hg_stream_desc) + 64)) ||
        (rc = ensure_scratch(ctx, 3, n * 8 + 64))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = hg::stage_upload(ctx, src.data(), ibytes.data(), ioff.data(), nullptr, n, iv * 4, (uint8_t *)ctx->d_scratch[0], s)) != HG_OK) return rc;
    uint32_t *d_ol = (uint32_t *)ctx->d_scratch[3]; int32_t *d_st = (int32_t *)ctx->d_scratch[3] + n;
    if (hipMemcpyAsync(ctx->d_scratch[2], d.data(), n * sizeof(hg_stream_desc), hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    if ((rc = hg_cram_itf8_encode_dev(ctx, (const int32_t *)ctx->d_scratch[0], (const hg_stream_desc *)ctx->d_scratch[2], n, ctx->d_scratch[1], d_ol, d_st, s)) != HG_OK) return rc;
    if (hipMemcpyAsync(out_len, d_ol, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(status, d_st, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    bool bad = false;
    for (size_t i = 0; i < n; i++) bad |= status[i] != 0;
    if ((rc = hg::stage_download(ctx, (const uint8_t *)ctx->d_scratch[1], ooff.data(), out_len, out, n, s)) != HG_OK) return rc;
    return bad ? HG_EBLOCK : HG_OK;
}

}  // extern "C"

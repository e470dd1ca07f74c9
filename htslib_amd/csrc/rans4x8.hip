// rans4x8.hip -- CRAM 3.0 "rANS 4x8" block decoder for MI355X (gfx950 / CDNA4).
//
// Replaces rans_uncompress() as called by cram_uncompress_block (reference
// cram/cram_io.c:1666-1683; implementation = htscodecs rANS_static.c, an absent submodule).
// Format and arithmetic: CRAM v3.0 specification, rANS codec (restated and pinned against the
// reference's CRAM fixtures in oracle/rans4x8_oracle.c).
//
// Mapping: a stream has only 4 interleaved rANS states that share ONE renormalisation byte
// pointer, so the parallelism inside a stream is 4 lanes; the chip is filled with streams:
//   * one stream per 4-lane group, 16 streams per wavefront; the four lanes of a group hold the
//     four states and decode one symbol each per step;
//   * the data-dependent split of the shared byte stream between the four states is a prefix sum
//     of the per-lane renormalisation byte counts inside the group (two DPP-style shuffles);
//   * order-0 frequency tables live in LDS as a 257-entry cumulative array per group (514 B) and
//     the slot -> symbol map is a binary search over it (8 LDS reads) instead of a 4 KiB lookup
//     table per stream, which would limit a CU to a few streams;
//   * order-1 tables (256 contexts) are kept as sparse per-context (cumulative, symbol) lists in
//     a global scratch area sized from the stream itself (every table entry costs >= 1 stream byte),
//     L1/L2 resident for the few KiB typical of quality-value alphabets;
//   * tables are parsed by lane 0 of each group (serial, small), all 16 groups of a wave in parallel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgr {

constexpr uint32_t TF_SHIFT = 12, TOTFREQ = 1u << TF_SHIFT, RANS_L = 1u << 23;
constexpr int GROUPS = 16;                     // streams per wave
constexpr int WAVES = 4;                       // waves per workgroup

struct GroupLds { uint16_t C[258]; };          // exclusive cumulative frequencies, C[256] = total

__device__ __forceinline__ uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// Parse one order-0 style table starting at cp.  emit(sym, freq, cum) is called for every symbol in
// increasing symbol order.  Returns the new cursor or nullptr.
template <typename Emit>
__device__ __forceinline__ const uint8_t *parse_table0(const uint8_t *cp, const uint8_t *end, uint32_t &total, Emit emit) {
    if (cp >= end) return nullptr;
    uint32_t rle = 0, x = 0, j = *cp++;
    for (int guard = 0; guard < 257; guard++) {
        if (cp + 2 > end) return nullptr;
        uint32_t f = *cp++;
        if (f >= 128) f = ((f & 127u) << 8) | *cp++;
        if (x + f > TOTFREQ) return nullptr;
        emit(j, f, x);
        x += f;
        if (cp >= end) return nullptr;
        if (!rle && j + 1 == *cp) {
            j = *cp++;
            if (cp >= end) return nullptr;
            rle = *cp++;
        } else if (rle) {
            rle--; j++;
            if (j > 255) return nullptr;
        } else {
            j = *cp++;
        }
        if (j == 0) { total = x; return cp; }
    }
    return nullptr;
}

// inclusive prefix sum inside a 4-lane group
__device__ __forceinline__ uint32_t group_scan4(uint32_t v, int sub) {
    uint32_t t = (uint32_t)__shfl_up((int)v, 1, 4);
    if (sub >= 1) v += t;
    t = (uint32_t)__shfl_up((int)v, 2, 4);
    if (sub >= 2) v += t;
    return v;
}

__global__ __launch_bounds__(WAVES * 64)
void rans4x8_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, uint32_t nstreams,
                           uint8_t *out, int32_t *status, uint32_t *scratch) {
    __shared__ GroupLds lds[WAVES * GROUPS];
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 3, grp = lane >> 2;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[(tid >> 6) * GROUPS + grp];

    for (uint32_t sidx = g_global; __any(sidx < nstreams); sidx += g_total) {
        const bool have = sidx < nstreams;
        int err = 0;
        uint32_t order = 0, usz = 0;
        const uint8_t *cp = nullptr, *end = nullptr;
        uint8_t *o = nullptr;
        uint32_t *tabs = nullptr;                               // order-1: cbase[256], cn[256], pairs...
        if (have) {
            const hg_stream_desc d = desc[sidx];
            const uint8_t *s = in + d.in_off;
            o = out + d.out_off;
            end = s + d.in_len;
            tabs = scratch + d.scratch_off;
            if (d.in_len < 9) err = 1;
            else {
                order = s[0];
                const uint32_t csz = rd32(s + 1);
                usz = rd32(s + 5);
                if ((uint64_t)csz + 9u != d.in_len || usz != d.out_len || order > 1) err = 1;
                cp = s + 9;
            }
        } else err = 2;                                          // idle group
        // ---- frequency tables: parsed by lane 0 of the group ----------------------------------
        if (!err && usz && sub == 0) {
            if (order == 0) {
                for (int i = 0; i < 258; i++) G.C[i] = 0;
                uint32_t total = 0, prev = 0;
                const uint8_t *q = parse_table0(cp, end, total, [&](uint32_t sym, uint32_t f, uint32_t cum) {
                    // symbols between prev and sym are absent: their C equals cum
                    for (uint32_t k = prev; k <= sym; k++) G.C[k] = (uint16_t)cum;
                    prev = sym + 1; (void)f;
                });
                if (!q) err = 1;
                else { for (uint32_t k = prev; k <= 256; k++) G.C[k] = (uint16_t)total; cp = q; }
            } else {
                // sparse lists: tabs[0..255] = first pair index of context, tabs[256..511] = pair count
                for (int i = 0; i < 512; i++) tabs[i] = 0;
                uint32_t np = 512;                               // next free word
                const uint32_t cap_words = 512u + 2u * (uint32_t)(end - cp) + 512u;
                uint32_t rle_i = 0, i = *cp++;
                bool ok = cp < end;
                for (int guard = 0; ok && guard < 257; guard++) {
                    uint32_t total = 0, cnt = 0;
                    const uint32_t first = np;
                    const uint8_t *q = parse_table0(cp, end, total, [&](uint32_t sym, uint32_t f, uint32_t cum) {
                        if (np + 2 <= cap_words) tabs[np] = (cum << 8) | sym;
                        np++; cnt++; (void)f;
                    });
                    if (!q || np + 1 > cap_words) { ok = false; break; }
                    tabs[np++] = (total << 8);                   // sentinel: total of this context
                    tabs[i] = first; tabs[256 + i] = cnt;
                    cp = q;
                    if (cp >= end) { ok = false; break; }
                    if (!rle_i && i + 1 == *cp) {
                        i = *cp++;
                        if (cp >= end) { ok = false; break; }
                        rle_i = *cp++;
                    } else if (rle_i) {
                        rle_i--; i++;
                        if (i > 255) { ok = false; break; }
                    } else {
                        i = *cp++;
                    }
                    if (i == 0) break;
                }
                if (!ok) err = 1;
            }
        }
        // broadcast the parse result (cursor, error) from lane 0 of the group
        {
            const int src = lane & ~3;
            err = __shfl(err, src, 64);
            const unsigned long long cpv = (unsigned long long)(uintptr_t)cp;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)cpv, src, 64);
            const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(cpv >> 32), src, 64);
            cp = (const uint8_t *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t R = 0;
        if (!err && usz) {
            if (cp + 16 > end) err = 1;
            else { R = rd32(cp + 4 * sub); cp += 16; }
        }
        // ---- decode ----------------------------------------------------------------------------
        const uint32_t q4 = usz >> 2;
        uint32_t steps = 0, ctx = 0;
        uint32_t pos;                                            // my next output index
        if (order == 0) { steps = q4; pos = (uint32_t)sub; }
        else { steps = q4; pos = (uint32_t)sub * q4; }
        const bool live = !err && usz != 0;
        uint32_t max_steps = live ? steps + (order == 1 ? (usz & 3u) : 0u) : 0u;
        // the loop runs while any group of the wave has steps left
        for (uint32_t it = 0; __any(it < max_steps); it++) {
            const bool act = live && !err && it < max_steps;
            // order-1 tail: only state 3 continues past q4 steps
            const bool mine = act && (it < steps || (order == 1 && sub == 3));
            uint32_t nbytes = 0;
            if (mine) {
                const uint32_t m = R & (TOTFREQ - 1);
                uint32_t sym, cum, f;
                if (order == 0) {
                    uint32_t lo = 0, hi = 256;                   // last j with C[j] <= m
                    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (G.C[mid] <= m) lo = mid; else hi = mid; }
                    if (G.C[256] <= m) { err = 1; sym = 0; cum = 0; f = 1; }
                    else { sym = lo; cum = G.C[lo]; f = G.C[lo + 1] - cum; }
                } else {
                    const uint32_t n = tabs[256 + ctx], base = tabs[ctx];
                    if (n == 0 || (tabs[base + n] >> 8) <= m) { err = 1; sym = 0; cum = 0; f = 1; }
                    else {
                        uint32_t lo = 0, hi = n;
                        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if ((tabs[base + mid] >> 8) <= m) lo = mid; else hi = mid; }
                        const uint32_t e = tabs[base + lo];
                        sym = e & 0xffu; cum = e >> 8; f = (tabs[base + lo + 1] >> 8) - cum;
                    }
                }
                if (!err) {
                    o[pos] = (uint8_t)sym;
                    pos += order == 0 ? 4u : 1u;
                    ctx = sym;
                    R = f * (R >> TF_SHIFT) + m - cum;
                    nbytes = R < RANS_L ? (R < (1u << 15) ? 2u : 1u) : 0u;
                }
            }
            // split the shared byte stream: prefix sum of byte counts inside the group
            const uint32_t incl = group_scan4(nbytes, sub);
            const uint32_t tot = (uint32_t)__shfl((int)incl, (lane & ~3) | 3, 64);
            if (mine && !err && nbytes) {
                const uint8_t *b = cp + (incl - nbytes);
                if (b + nbytes > end) err = 1;
                else { R = (R << 8) | b[0]; if (nbytes == 2) R = (R << 8) | b[1]; }
            }
            cp += tot;
            // any lane of the group failing fails the stream
            err |= __shfl_xor(err, 1, 64);
            err |= __shfl_xor(err, 2, 64);
        }
        // order-0 tail: states 0..(usz&3)-1 give one more symbol each, without update
        if (live && !err && order == 0 && (uint32_t)sub < (usz & 3u)) {
            const uint32_t m = R & (TOTFREQ - 1);
            uint32_t lo = 0, hi = 256;
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (G.C[mid] <= m) lo = mid; else hi = mid; }
            if (G.C[256] <= m) err = 1; else o[(usz & ~3u) + sub] = (uint8_t)lo;
        }
        err |= __shfl_xor(err, 1, 64);
        err |= __shfl_xor(err, 2, 64);
        if (have && sub == 0) status[sidx] = (err == 0) ? 0 : -1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace hgr

namespace hg {
int launch_rans4x8_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, void *d_out,
                          int32_t *d_status, uint32_t *d_scratch, hipStream_t s) {
    if (n == 0) return HG_OK;
    size_t groups_per_wg = hgr::WAVES * hgr::GROUPS;
    size_t wgs = (n + groups_per_wg - 1) / groups_per_wg;
    size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgr::rans4x8_decode_kernel, dim3((unsigned)wgs), dim3(hgr::WAVES * 64), 0, s,
                       (const uint8_t *)d_in, d_desc, (uint32_t)n, (uint8_t *)d_out, d_status, d_scratch);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

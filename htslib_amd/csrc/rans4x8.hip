// rans4x8.hip -- CRAM 3.0 "rANS 4x8" block decoder for MI355X (gfx950 / CDNA4).
//
// Replaces rans_uncompress() as called by cram_uncompress_block (reference
// cram/cram_io.c:1666-1683; implementation = htscodecs rANS_static.c, an absent submodule).
// Format and arithmetic: CRAM v3.0 specification, rANS codec (restated and pinned against the
// reference's CRAM fixtures in oracle/rans4x8_oracle.c).
//
// Mapping: a stream has only 4 interleaved rANS states that share ONE renormalisation byte pointer, so the parallelism inside a
// stream is 4 lanes and a stream is one chain of n / 4 steps; a launch lasts as long as its longest chain, so everything a step
// touches is in LDS (round 2; before, the order-1 lists and the stream bytes were read from global memory inside the chain --
// ~10 dependent loads, 1.5 us per step):
//   * four streams per wavefront (lanes 0..15; the lanes of a quad hold the four states), 16 per workgroup, 9 KiB of LDS each;
//   * the stream bytes come from a 128-byte LDS ring per stream, topped up 64 bytes at a time from a register prefetch (the same
//     scheme as the 32-way Nx16 decoder); the data-dependent split of the bytes between the four states is a prefix sum inside
//     the quad on the DPP path (quad_perm, no LDS round trip);
//   * order 0: slot -> symbol byte table (4 KiB) + the 257-entry cumulative array;
//   * order 1, small alphabet (<= 16 contexts of <= 16 symbols): the dense two-read form of the Nx16 decoder (a 256-bucket index per
//     context, then the (cumulative, symbol, next-context rank) entry and its neighbour);
//   * order 1 otherwise: the sparse per-context (cumulative, symbol) lists copied into LDS when they fit (binary search there), else
//     left in the global scratch area they were parsed into;
//   * tables are parsed by lane 0 of each quad (serial, small).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgr {

constexpr uint32_t TF_SHIFT = 12, TOTFREQ = 1u << TF_SHIFT, RANS_L = 1u << 23;
constexpr int GROUPS = 4;                      // streams per wave (lanes 16..63 idle)
constexpr int WAVES = 2;                       // waves per workgroup
constexpr uint32_t POOLW = 4096;               // LDS words per stream for the tables (8 streams per workgroup: 138 KiB)

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

// inclusive prefix sum inside a quad, and the quad's total, on the DPP path (quad_perm)
__device__ __forceinline__ uint32_t group_scan4(uint32_t v, int sub) {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x90, 0xf, 0xf, false);      // lane i <- lane max(i - 1, 0)
    if (sub >= 1) v += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x44, 0xf, 0xf, false);               // lanes 2, 3 <- lanes 0, 1
    if (sub >= 2) v += t;
    return v;
}
__device__ __forceinline__ uint32_t quad_perm0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, false); }   // lane 0 of the quad
__device__ __forceinline__ uint32_t quad_perm1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false); }   // xor 1: [1,0,3,2]
__device__ __forceinline__ uint32_t quad_perm2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false); }   // xor 2: [2,3,0,1]
__device__ __forceinline__ uint32_t quad_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xff, 0xf, 0xf, false); }

__global__ __launch_bounds__(WAVES * 64)
void rans4x8_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc, uint32_t nstreams,
                           uint8_t *out, int32_t *status, uint32_t *scratch) {
    __shared__ GroupLds lds[WAVES * GROUPS];
    __shared__ uint32_t pool[WAVES * GROUPS][POOLW];
    __shared__ uint32_t ring_s[WAVES * GROUPS][32];             // 128 stream bytes
    __shared__ uint8_t rank_s[WAVES * GROUPS][256];
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 3;
    const bool idle = (lane >> 2) >= GROUPS;
    const int grp = idle ? 0 : lane >> 2;
    const int slot = (tid >> 6) * GROUPS + grp;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[slot];
    uint32_t *P = pool[slot];
    uint32_t *ring = ring_s[slot];
    uint8_t *rk = rank_s[slot];

    for (uint32_t sidx = g_global; __any(sidx < nstreams); sidx += g_total) {
        const bool have = sidx < nstreams && !idle;
        int err = 0;
        uint32_t order = 0, usz = 0;
        const uint8_t *cp = nullptr, *end = nullptr;
        uint8_t *o = nullptr;
        uint32_t *tabs = nullptr;                               // order-1: cbase[256], cn[256], pairs...
        uint32_t np_words = 0xffffffffu;
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
        } else err = 2;                                          // idle quad
        // ---- frequency tables: parsed by lane 0 of the quad -----------------------------------
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
                np_words = np;
            }
        }
        // broadcast the parse result (cursor, error, table size) from lane 0 of the quad
        {
            const int src = lane & ~3;
            err = __shfl(err, src, 64);
            np_words = (uint32_t)__shfl((int)np_words, src, 64);
            const unsigned long long cpv = (unsigned long long)(uintptr_t)cp;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)cpv, src, 64);
            const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(cpv >> 32), src, 64);
            cp = (const uint8_t *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- tables into LDS ---------------------------------------------------------------------
        const bool core = !err && usz != 0;
        const uint8_t *lut = nullptr;                            // order 0: slot -> symbol
        const uint8_t *dL = nullptr; const uint32_t *dD = nullptr, *dT = nullptr; uint32_t rctx = 0;   // order 1, dense form
        const uint32_t *bI = nullptr, *bL = nullptr; const uint8_t *bB = nullptr;           // order 1, bucket form in LDS
        const uint32_t *T = tabs;                                // order 1, lists in global scratch (tables too big for LDS)
        if (core && order == 0) {
            uint8_t *L8 = (uint8_t *)P;
            for (uint32_t sy = (uint32_t)sub; sy < 256; sy += 4) { const uint32_t a = G.C[sy], b2 = G.C[sy + 1]; for (uint32_t q = a; q < b2; q++) L8[q] = (uint8_t)sy; }
            lut = L8;
        } else if (core) {
            uint32_t nctx = 0, big = 0;
            if (sub == 0) for (int i = 0; i < 256; i++) { rk[i] = (uint8_t)nctx; const uint32_t c = tabs[256 + i]; if (c) nctx++; if (c > 16u) big = 1; }
            nctx = quad_perm0(nctx); big = quad_perm0(big);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!big && nctx <= 16u) {
                // dense: L[rank][slot >> 4] = list index of the bucket, DD[rank][index] = cum << 12 | sym << 4 | rank of sym as a context
                uint8_t *L = (uint8_t *)P;
                uint32_t *DDw = P + 1024u, *TOT = P + 1024u + 16u * 17u;
                for (uint32_t i = (uint32_t)sub; i < 256; i += 4) {
                    const uint32_t cnt = tabs[256 + i], base = tabs[i];
                    if (!cnt) continue;
                    const uint32_t r = rk[i];
                    TOT[r] = tabs[base + cnt] >> 8;                  // the context's total: slots at or beyond it are not decodable
                    for (uint32_t k = 0; k <= cnt; k++) {
                        const uint32_t e = tabs[base + k], sy = e & 0xffu;
                        const uint32_t nx = k < cnt ? (tabs[256 + sy] ? (uint32_t)rk[sy] : (1u << 25)) : 0u;
                        DDw[r * 17u + k] = ((e >> 8) << 12) | (sy << 4) | nx;
                    }
                    uint32_t k = 0;
                    for (uint32_t bkt = 0; bkt < 256; bkt++) {
                        const uint32_t sl = bkt << 4;
                        while (k + 1 < cnt && (tabs[base + k + 1] >> 8) <= sl) k++;
                        L[r * 256u + bkt] = (uint8_t)k;
                    }
                }
                dL = L; dD = DDw; dT = TOT;
                rctx = tabs[256] ? (uint32_t)rk[0] : 0xffffu;     // the states start in context 0
            } else if ((np_words - 512u) + nctx * 17u <= POOLW) {
                // bucket form: INFO[rank] = list start | entries << 16; BK[rank][slot >> 6] = list index of the bucket (64 buckets);
                // list entries cum << 16 | rank of the symbol as a context << 8 | symbol (rank 255 + flag bit 29: never a context)
                uint32_t *INFO = P, *LST = P + nctx * 17u;
                uint8_t *BK = (uint8_t *)(P + nctx);
                uint32_t off = 0;
                // list offsets in rank order: lane 0 of the quad walks the 256 contexts once
                if (sub == 0) for (int i = 0; i < 256; i++) { const uint32_t c = tabs[256 + i]; if (c) { INFO[rk[i]] = off | (c << 16); off += c + 1u; } }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = (uint32_t)sub; i < 256; i += 4) {
                    const uint32_t cnt = tabs[256 + i], base = tabs[i];
                    if (!cnt) continue;
                    const uint32_t r = rk[i], lo = INFO[r] & 0xffffu;
                    for (uint32_t k = 0; k <= cnt; k++) {
                        const uint32_t e = tabs[base + k], sy = e & 0xffu;
                        const uint32_t nx = k < cnt ? (tabs[256 + sy] ? ((uint32_t)rk[sy] << 8) : (1u << 29)) : 0u;
                        LST[lo + k] = ((e >> 8) << 16) | nx | sy;
                    }
                    uint32_t k = 0;
                    for (uint32_t bkt = 0; bkt < 64; bkt++) {
                        const uint32_t sl = bkt << 6;
                        while (k + 1 < cnt && (tabs[base + k + 1] >> 8) <= sl) k++;
                        BK[r * 64u + bkt] = (uint8_t)k;
                    }
                }
                bI = INFO; bB = BK; bL = LST;
                rctx = tabs[256] ? (uint32_t)rk[0] : 0xffffu;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t R = 0;
        if (core) {
            if (cp + 16 > end) err = 1;
            else { R = rd32(cp + 4 * sub); cp += 16; }
        }
        // ---- the stream bytes: a 128-byte LDS ring, 64 bytes per refill (16 per lane), one refill prefetched in registers
        const uint8_t *wbase = cp;
        const uint32_t wavail = core && !err ? (uint32_t)(end - wbase) : 0u;
        auto load_chunk = [&](uint32_t c) -> uint4 {             // bytes 64 c + 16 sub .. + 16 (zeros past the end)
            const uint8_t *p = wbase + 64u * c + 16u * (uint32_t)sub;
            uint4 v = {0, 0, 0, 0};
            if (p + 16 <= end) __builtin_memcpy(&v, p, 16);
            else if (p < end) { uint32_t w[4] = {0, 0, 0, 0}; for (uint32_t q = 0; p + q < end; q++) w[q >> 2] |= (uint32_t)p[q] << (8u * (q & 3u)); v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3]; }
            return v;
        };
        auto put_chunk = [&](uint32_t c, const uint4 &v) { uint32_t *d = ring + ((c & 1u) * 16u + 4u * (uint32_t)sub); d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; };
        uint32_t wpos = 0, wfill = 0;                            // bytes consumed / bytes present in the ring (multiples of 64)
        uint4 pre = {0, 0, 0, 0};
        if (core && !err) { put_chunk(0, load_chunk(0)); put_chunk(1, load_chunk(1)); wfill = 128; pre = load_chunk(2); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint8_t *ring8 = (const uint8_t *)ring;
        // ---- decode ----------------------------------------------------------------------------
        const uint32_t q4 = usz >> 2;
        uint32_t steps = q4, ctx = 0;
        uint32_t pos = order == 0 ? (uint32_t)sub : (uint32_t)sub * q4;      // my next output index
        const bool live = core && !err;
        uint32_t max_steps = live ? steps + (order == 1 ? (usz & 3u) : 0u) : 0u;
        // the loop runs while any quad of the wave has steps left
        for (uint32_t it = 0; __any(it < max_steps); it++) {
            const bool act = live && !err && it < max_steps;
            // order-1 tail: only state 3 continues past q4 steps
            const bool mine = act && (it < steps || (order == 1 && sub == 3));
            uint32_t nbytes = 0;
            if (mine) {
                const uint32_t m = R & (TOTFREQ - 1);
                uint32_t sym = 0, cum = 0, f = 1;
                if (order == 0) {
                    if (G.C[256] <= m) err = 1;
                    else { sym = lut[m]; cum = G.C[sym]; f = (uint32_t)G.C[sym + 1] - cum; }
                } else if (dD) {
                    if (rctx > 15u || dT[rctx] <= m) err = 1;     // context never seen by the encoder / slot beyond its total
                    else {
                        uint32_t kb = dL[rctx * 256u + (m >> 4)];
                        const uint32_t *row = dD + rctx * 17u;
                        uint32_t e = row[kb], e1 = row[kb + 1];
                        while (((e1 >> 12) & 0x1fffu) <= m) { kb++; e = e1; e1 = row[kb + 1]; }      // ends at the total (> m)
                        sym = (e >> 4) & 0xffu; cum = (e >> 12) & 0x1fffu; f = ((e1 >> 12) & 0x1fffu) - cum;
                        rctx = (e >> 25) ? 0xffffu : (e & 15u);
                    }
                } else if (bL) {
                    if (rctx > 255u) err = 1;                    // context never seen by the encoder
                    else {
                        const uint32_t info = bI[rctx], base = info & 0xffffu, n = info >> 16;
                        uint32_t kb = bB[rctx * 64u + (m >> 6)];
                        uint32_t e = bL[base + kb], e1 = bL[base + kb + 1];
                        if (((bL[base + n] >> 16) & 0x1fffu) <= m) err = 1;          // slot beyond the context's total
                        else {
                            while (((e1 >> 16) & 0x1fffu) <= m) { kb++; e = e1; e1 = bL[base + kb + 1]; }
                            sym = e & 0xffu; cum = (e >> 16) & 0x1fffu; f = ((e1 >> 16) & 0x1fffu) - cum;
                            rctx = (e >> 29) ? 0xffffu : ((e >> 8) & 0xffu);
                        }
                    }
                } else {
                    const uint32_t n = T[256 + ctx], base = T[ctx];
                    if (n == 0 || (T[base + n] >> 8) <= m) err = 1;
                    else {
                        uint32_t lo = 0, hi = n;
                        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if ((T[base + mid] >> 8) <= m) lo = mid; else hi = mid; }
                        const uint32_t e = T[base + lo];
                        sym = e & 0xffu; cum = e >> 8; f = (T[base + lo + 1] >> 8) - cum;
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
            // split the shared byte stream: prefix sum of byte counts inside the quad
            const uint32_t incl = group_scan4(nbytes, sub);
            const uint32_t tot = quad_last(incl);
            if (mine && !err && nbytes) {
                const uint32_t k = wpos + incl - nbytes;
                if (k + nbytes > wavail) err = 1;
                else { R = (R << 8) | ring8[k & 127u]; if (nbytes == 2) R = (R << 8) | ring8[(k + 1u) & 127u]; }
            }
            wpos += tot;
            if (act && wfill - wpos < 64u) {                     // the whole quad takes this branch together: at least 64 bytes are free
                put_chunk(wfill >> 6, pre);
                wfill += 64;
                pre = load_chunk(wfill >> 6);
            }
            // any lane of the quad failing fails the stream
            err |= (int)quad_perm1((uint32_t)err);
            err |= (int)quad_perm2((uint32_t)err);
        }
        // order-0 tail: states 0..(usz&3)-1 give one more symbol each, without update
        if (live && !err && order == 0 && (uint32_t)sub < (usz & 3u)) {
            const uint32_t m = R & (TOTFREQ - 1);
            if (G.C[256] <= m) err = 1; else o[(usz & ~3u) + sub] = lut[m];
        }
        err |= (int)quad_perm1((uint32_t)err);
        err |= (int)quad_perm2((uint32_t)err);
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

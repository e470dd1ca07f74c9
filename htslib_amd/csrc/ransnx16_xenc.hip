// ransnx16_xenc.hip -- the byte transforms that PRECEDE entropy coding in a CRAM 3.1 rANS Nx16
// stream (STRIPE de-interleave, PACK, RLE), encoder side, on MI355X (gfx950).
//
// Reference boundary: rans_compress_4x16 as called by cram_compress_by_method with the RANS_PR*
// flag sets {1,64,9,128,129,192,193} (cram/cram_io.c:1853-1866).  Choices (which symbols are run-length
// coded, the symbol map, the meta layout) follow oracle/ransnx16_oracle.c pack()/rle_encode() exactly, so
// the assembled streams are byte-identical to the oracle's.  PARITY UNPINNED like the rest of Nx16.
//
// One wavefront per leaf stream:
//   gather   byte i of stripe k = in[i*S + k] (only when striped)
//   PACK     256-entry "used" table in LDS -> rank = map; 2, 4 or 8 symbols per output byte
//   RLE      per-symbol score (repeats minus run starts, LDS atomics) picks the run symbols; then 64 input
//            bytes per step: a ballot of the "literal starts here" flags gives every literal its slot
//            (prefix popcount) and every run its end (next set bit); run lengths become uint7 varints whose
//            byte offsets come from a wave scan.  A run still open at the end of a step is carried.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgy {
using hg::wave_sync;

constexpr int WAVES = 4;
struct WaveLds { int32_t score[256]; uint8_t used[256]; uint8_t map[256]; };

__device__ __forceinline__ uint32_t u7_len(uint32_t v) { return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u; }
__device__ __forceinline__ void u7_put(uint8_t *cp, uint32_t v, uint32_t nb) {
    for (uint32_t k = 0; k < nb; k++) { const uint32_t sh = 7u * (nb - 1u - k); cp[k] = (uint8_t)(((v >> sh) & 0x7fu) | (k + 1u < nb ? 0x80u : 0u)); }
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t &total) {
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)x, d, 64); if (lane >= d) x += t; }
    total = (uint32_t)__shfl((int)x, 63, 64);
    return x - v;
}

__global__ __launch_bounds__(WAVES * 64)
void nx16_xenc_kernel(uint8_t *buf, const hg::nx16_xenc *__restrict__ jobs, uint32_t njobs, hg::nx16_xenc_res *res, uint32_t skip_from) {
    __shared__ WaveLds lds[WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    WaveLds &S = lds[wv];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t j = blockIdx.x * WAVES + wv; j < njobs; j += gridDim.x * WAVES) {
        const hg::nx16_xenc J = jobs[j];
        if (J.n >= skip_from) continue;                                    // long streams: nx16_xenc_big_kernel
        uint32_t flags = J.flags;
        const uint8_t *cur = buf + J.src_off;
        uint32_t n = J.n;
        if (J.stride != 1) {
            uint8_t *g = buf + J.g_off;
            for (uint32_t i = (uint32_t)lane; i < n; i += 64) g[i] = cur[(size_t)i * J.stride];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); wave_sync();
            cur = g;
        }
        uint32_t nsym = 0, plen = 0, lit_len = 0, meta_len = 0;
        uint8_t my_map = 0;                                                // lane k < nsym holds map symbol k
        // ------------------------------------------------------------------ PACK
        if ((flags & 0x80u) && n) {
            for (int k = lane; k < 256; k += 64) S.used[k] = 0;
            wave_sync();
            for (uint32_t i = (uint32_t)lane; i < n; i += 64) S.used[cur[i]] = 1;
            wave_sync();
            uint32_t base = 0;
            for (int q = 0; q < 4; q++) {
                const int sym = q * 64 + lane;
                const bool u = S.used[sym] != 0;
                const unsigned long long B = __ballot(u);
                const uint32_t rank = base + (uint32_t)__popcll(B & below);
                S.map[sym] = (uint8_t)rank;
                base += (uint32_t)__popcll(B);
            }
            nsym = base;
            wave_sync();
            if (nsym > 16) flags &= ~0x80u;
            else {
                // lane r learns the r-th used symbol
                for (int q = 0; q < 4; q++) { const int sym = q * 64 + lane; if (S.used[sym]) S.score[S.map[sym]] = sym; }
                wave_sync();
                my_map = (uint32_t)lane < nsym ? (uint8_t)S.score[lane] : 0;
                if (nsym > 1) {
                    const uint32_t bits = nsym <= 2 ? 1u : nsym <= 4 ? 2u : 4u, per = 8u / bits;
                    uint8_t *P = buf + J.p_off;
                    plen = (n + per - 1u) / per;
                    for (uint32_t o = (uint32_t)lane; o < plen; o += 64) {
                        uint32_t v = 0;
                        for (uint32_t k = 0; k < per; k++) { const uint32_t i = o * per + k; if (i < n) v |= (uint32_t)S.map[cur[i]] << (k * bits); }
                        P[o] = (uint8_t)v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); wave_sync();
                    cur = P;
                }
                n = plen;
            }
        } else flags &= ~0x80u;
        // ------------------------------------------------------------------ RLE
        if ((flags & 0x40u) && n) {
            uint8_t *meta = buf + J.m_off, *lit = buf + J.l_off;
            uint32_t nr = 0;
            if (flags & 0x100u) {                                          // the caller chose the run symbols (hts_rle_encode)
                nr = meta[0]; if (!nr) nr = 256;
                for (int k = lane; k < 256; k += 64) S.used[k] = 0;
                wave_sync();
                for (uint32_t k = (uint32_t)lane; k < nr; k += 64) S.used[meta[1u + k]] = 1;
                wave_sync();
            } else {
            for (int k = lane; k < 256; k += 64) S.score[k] = 0;
            wave_sync();
            for (uint32_t i = (uint32_t)lane; i < n; i += 64) atomicAdd(&S.score[cur[i]], (i && cur[i] == cur[i - 1]) ? 1 : -1);
            wave_sync();
            for (int q = 0; q < 4; q++) {
                const int sym = q * 64 + lane;
                const bool r = S.score[sym] > 0;
                const unsigned long long B = __ballot(r);
                if (r) meta[1u + nr + (uint32_t)__popcll(B & below)] = (uint8_t)sym;
                S.used[sym] = r ? 1 : 0;
                nr += (uint32_t)__popcll(B);
            }
            wave_sync();
            }
            if (!nr) flags &= ~0x40u;
            else {
                if (lane == 0) meta[0] = (uint8_t)nr;
                uint32_t mp = 1u + nr, lo = 0;
                bool pend = false; uint32_t pend_pos = 0;
                // the bytes of the NEXT step are requested before this step's scattered stores: loads issued behind stores
                // would wait for them (one counter orders both on gfx9)
                uint8_t cn = (uint32_t)lane < n ? cur[lane] : 0, carry = 0;          // carry = the last byte of the previous step
                for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                    const uint32_t i = i0 + (uint32_t)lane;
                    const bool has = i < n;
                    const uint8_t c = cn;
                    if (i + 64u < n) cn = cur[i + 64u];
                    const uint8_t up = (uint8_t)__shfl_up((int)c, 1, 64);
                    const uint8_t pc = lane == 0 ? carry : up;                           // byte i - 1
                    carry = (uint8_t)__shfl((int)c, 63, 64);
                    const bool isr = has && S.used[c];
                    const bool st = has && (i == 0 || !isr || pc != c);
                    const unsigned long long St = __ballot(st);
                    if (st) lit[lo + (uint32_t)__popcll(St & below)] = c;
                    lo += (uint32_t)__popcll(St);
                    if (pend && St) {                                      // the carried run ends at the first start
                        const uint32_t r = i0 + (uint32_t)__builtin_ctzll(St) - pend_pos, nb = u7_len(r - 1u);
                        if (lane == 0) u7_put(meta + mp, r - 1u, nb);
                        mp += nb; pend = false;
                    }
                    const unsigned long long above = lane == 63 ? 0ull : St & ~((2ull << lane) - 1ull);
                    const bool rs = st && isr;                             // a run starts in this lane
                    const bool closed = rs && above != 0;
                    const uint32_t r = closed ? (uint32_t)__builtin_ctzll(above) - (uint32_t)lane : 0u;
                    const uint32_t nb = closed ? u7_len(r - 1u) : 0u;
                    uint32_t tot;
                    const uint32_t off = wave_excl_scan(nb, lane, tot);
                    if (closed) u7_put(meta + mp + off, r - 1u, nb);
                    mp += tot;
                    const unsigned long long open = __ballot(rs && !closed);   // at most the last start
                    if (open) { pend = true; pend_pos = i0 + (uint32_t)__builtin_ctzll(open); }
                }
                if (pend) { const uint32_t r = n - pend_pos, nb = u7_len(r - 1u); if (lane == 0) u7_put(meta + mp, r - 1u, nb); mp += nb; }
                meta_len = mp; lit_len = lo;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); wave_sync();
                cur = lit; n = lo;
            }
        } else flags &= ~0x40u;
        hg::nx16_xenc_res R;
        R.flags = flags; R.nsym = nsym; R.plen = plen; R.lit_len = lit_len; R.meta_len = meta_len;
        R.cur_off = (uint64_t)(cur - buf); R.cur_len = n;
        // the map travels through lanes 0..15
        for (int k = 0; k < 16; k++) R.map[k] = (uint8_t)__shfl((int)my_map, k, 64);
        R.pad = 0;
        res[j] = R;                                                        // every lane stores the same record
        wave_sync();
    }
}


// ---- long streams (round 5): ONE WORKGROUP of 16 wavefronts per stream.  A 1.5 MB quality series through the one-wave kernel above takes 20-28 ms -- on the
// critical path of every slice batch, ahead of the entropy coder that waits for it.  Here the gather, the "used" table, PACK and the RLE scores are spread over all
// threads, and the RLE emission runs per CHUNK (one wavefront each) in two walks: count (literals and meta bytes of the chunk), a prefix over the chunks, emit.
// A run belongs to the chunk it STARTS in; its owner looks past the chunk's end for the next start (the bytes a later chunk begins with, if they continue a run,
// are neither literals nor run starts there: nothing to do for them).  Same choices, same bytes as the one-wave kernel.
constexpr int BIG_WAVES = 16;
constexpr uint32_t BIG_MIN = 64u << 10;

// the RLE walk of positions [a, b) of cur[0, n): WRITE = false counts, true emits at lit + lo / meta + mp.  Returns through lo / mp the advanced cursors.
template <bool WRITE>
__device__ __forceinline__ void rle_chunk(const uint8_t *__restrict__ cur, uint32_t n, uint32_t a, uint32_t b, const uint8_t *used, uint8_t *lit, uint8_t *meta, uint32_t &lo, uint32_t &mp, int lane) {
    const unsigned long long below = (1ull << lane) - 1ull;
    bool pend = false; uint32_t pend_pos = 0;
    uint8_t cn = a + (uint32_t)lane < n ? cur[a + (uint32_t)lane] : 0;
    uint8_t carry = a ? cur[a - 1u] : 0;                                    // the byte before the chunk (every lane reads the same address)
    for (uint32_t i0 = a; i0 < b; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool has = i < b;                                              // (b <= n; chunks are multiples of 64 except the last)
        const uint8_t c = cn;
        if (i + 64u < n) cn = cur[i + 64u];
        const uint8_t up = (uint8_t)__shfl_up((int)c, 1, 64);
        const uint8_t pc = lane == 0 ? carry : up;
        carry = (uint8_t)__shfl((int)c, 63, 64);
        const bool isr = has && used[c];
        const bool st = has && (i == 0 || !isr || pc != c);
        const unsigned long long St = __ballot(st);
        if (WRITE) { if (st) lit[lo + (uint32_t)__popcll(St & below)] = c; }
        lo += (uint32_t)__popcll(St);
        if (pend && St) {
            const uint32_t r = i0 + (uint32_t)__builtin_ctzll(St) - pend_pos, nb = u7_len(r - 1u);
            if (WRITE && lane == 0) u7_put(meta + mp, r - 1u, nb);
            mp += nb; pend = false;
        }
        const unsigned long long above = lane == 63 ? 0ull : St & ~((2ull << lane) - 1ull);
        const bool rs = st && isr;
        const bool closed = rs && above != 0;
        const uint32_t r = closed ? (uint32_t)__builtin_ctzll(above) - (uint32_t)lane : 0u;
        const uint32_t nb = closed ? u7_len(r - 1u) : 0u;
        uint32_t tot;
        const uint32_t off = wave_excl_scan(nb, lane, tot);
        if (WRITE && closed) u7_put(meta + mp + off, r - 1u, nb);
        mp += tot;
        const unsigned long long open = __ballot(rs && !closed);
        if (open) { pend = true; pend_pos = i0 + (uint32_t)__builtin_ctzll(open); }
    }
    if (pend) {
        // the run is still open at the end of the chunk: it ends at the first position >= b that starts something (a different byte), or at n
        uint32_t end = n;
        const uint8_t sym = cur[pend_pos];
        for (uint32_t i0 = b; i0 < n; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const unsigned long long diff = __ballot(i < n && cur[i] != sym);
            if (diff) { end = i0 + (uint32_t)__builtin_ctzll(diff); break; }
        }
        const uint32_t r = end - pend_pos, nb = u7_len(r - 1u);
        if (WRITE && lane == 0) u7_put(meta + mp, r - 1u, nb);
        mp += nb;
    }
}

__global__ __launch_bounds__(BIG_WAVES * 64)
void nx16_xenc_big_kernel(uint8_t *buf, const hg::nx16_xenc *__restrict__ jobs, const uint32_t *__restrict__ big, hg::nx16_xenc_res *res) {
    __shared__ uint8_t used[256], map[256];
    __shared__ int32_t score[BIG_WAVES][256];
    __shared__ uint32_t wlit[BIG_WAVES + 1], wmeta[BIG_WAVES + 1];
    __shared__ uint32_t sh_nsym, sh_nr;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t j = big[blockIdx.x];
    const hg::nx16_xenc J = jobs[j];
    uint32_t flags = J.flags;
    const uint8_t *cur = buf + J.src_off;
    uint32_t n = J.n;
    constexpr uint32_t T = BIG_WAVES * 64;
    if (J.stride != 1) {
        uint8_t *g = buf + J.g_off;
        for (uint32_t i = (uint32_t)tid; i < n; i += T) g[i] = cur[(size_t)i * J.stride];
        __threadfence_block(); __syncthreads();
        cur = g;
    }
    uint32_t nsym = 0, plen = 0, lit_len = 0, meta_len = 0;
    uint8_t my_map = 0;
    // ------------------------------------------------------------------ PACK
    if ((flags & 0x80u) && n) {
        for (int k = tid; k < 256; k += T) used[k] = 0;
        __syncthreads();
        for (uint32_t i = (uint32_t)tid; i < n; i += T) used[cur[i]] = 1;
        __syncthreads();
        if (wv == 0) {
            uint32_t base = 0;
            for (int q = 0; q < 4; q++) {
                const int sym = q * 64 + lane;
                const bool u = used[sym] != 0;
                const unsigned long long B = __ballot(u);
                map[sym] = (uint8_t)(base + (uint32_t)__popcll(B & below));
                base += (uint32_t)__popcll(B);
            }
            if (lane == 0) sh_nsym = base;
        }
        __syncthreads();
        nsym = sh_nsym;
        if (nsym > 16) flags &= ~0x80u;
        else {
            // lane r of every wavefront learns the r-th used symbol (through wave 0's score row)
            if (wv == 0) for (int q = 0; q < 4; q++) { const int sym = q * 64 + lane; if (used[sym]) score[0][map[sym]] = sym; }
            __syncthreads();
            my_map = (uint32_t)lane < nsym ? (uint8_t)score[0][lane] : 0;
            if (nsym > 1) {
                const uint32_t bits = nsym <= 2 ? 1u : nsym <= 4 ? 2u : 4u, per = 8u / bits;
                uint8_t *P = buf + J.p_off;
                plen = (n + per - 1u) / per;
                for (uint32_t o = (uint32_t)tid; o < plen; o += T) {
                    uint32_t v = 0;
                    for (uint32_t k = 0; k < per; k++) { const uint32_t i = o * per + k; if (i < n) v |= (uint32_t)map[cur[i]] << (k * bits); }
                    P[o] = (uint8_t)v;
                }
                __threadfence_block();
                cur = P;
            }
            n = plen;
            __syncthreads();
        }
    } else flags &= ~0x80u;
    // ------------------------------------------------------------------ RLE
    if ((flags & 0x40u) && n) {
        uint8_t *meta = buf + J.m_off, *lit = buf + J.l_off;
        uint32_t nr = 0;
        __syncthreads();
        if (flags & 0x100u) {                                              // the caller chose the run symbols (hts_rle_encode)
            nr = meta[0]; if (!nr) nr = 256;
            for (int k = tid; k < 256; k += T) used[k] = 0;
            __syncthreads();
            for (uint32_t k = (uint32_t)tid; k < nr; k += T) used[meta[1u + k]] = 1;
            __syncthreads();
        } else {
            for (int k = lane; k < 256; k += 64) score[wv][k] = 0;
            __syncthreads();
            for (uint32_t i = (uint32_t)tid; i < n; i += T) atomicAdd(&score[wv][cur[i]], (i && cur[i] == cur[i - 1]) ? 1 : -1);
            __syncthreads();
            if (wv == 0) {
                for (int q = 0; q < 4; q++) {
                    const int sym = q * 64 + lane;
                    int32_t t = 0;
                    for (int w = 0; w < BIG_WAVES; w++) t += score[w][sym];
                    const bool r = t > 0;
                    const unsigned long long B = __ballot(r);
                    if (r) meta[1u + nr + (uint32_t)__popcll(B & below)] = (uint8_t)sym;
                    used[sym] = r ? 1 : 0;
                    nr += (uint32_t)__popcll(B);
                }
                if (lane == 0) sh_nr = nr;
            }
            __syncthreads();
            nr = sh_nr;
        }
        if (!nr) flags &= ~0x40u;
        else {
            if (tid == 0) meta[0] = (uint8_t)nr;
            // chunks: multiples of 64 positions, one per wavefront
            const uint32_t cs = (((n + BIG_WAVES - 1u) / BIG_WAVES) + 63u) & ~63u;
            const uint32_t a = (uint32_t)wv * cs < n ? (uint32_t)wv * cs : n, b = a + cs < n ? a + cs : n;
            uint32_t lo = 0, mp = 0;
            if (a < b) rle_chunk<false>(cur, n, a, b, used, lit, meta, lo, mp, lane);
            if (lane == 0) { wlit[wv + 1] = lo; wmeta[wv + 1] = mp; }
            __syncthreads();
            if (tid == 0) { wlit[0] = 0; wmeta[0] = 1u + nr; for (int w = 0; w < BIG_WAVES; w++) { wlit[w + 1] += wlit[w]; wmeta[w + 1] += wmeta[w]; } }
            __syncthreads();
            lo = wlit[wv]; mp = wmeta[wv];
            if (a < b) rle_chunk<true>(cur, n, a, b, used, lit, meta, lo, mp, lane);
            lit_len = wlit[BIG_WAVES]; meta_len = wmeta[BIG_WAVES];
            __threadfence_block(); __syncthreads();
            cur = lit; n = lit_len;
        }
    } else flags &= ~0x40u;
    if (wv == 0) {
        hg::nx16_xenc_res R;
        R.flags = flags; R.nsym = nsym; R.plen = plen; R.lit_len = lit_len; R.meta_len = meta_len;
        R.cur_off = (uint64_t)(cur - buf); R.cur_len = n;
        for (int k = 0; k < 16; k++) R.map[k] = (uint8_t)__shfl((int)my_map, k, 64);
        R.pad = 0;
        res[j] = R;                                                        // every lane of wave 0 stores the same record
    }
}

}  // namespace hgy

namespace hg {
// h_jobs (optional): the jobs as the host holds them -- streams of 64 KiB and more then go to the workgroup-wide kernel (their indices travel in scratch slot 14)
int launch_ransnx16_xenc(hg_ctx *ctx, void *d_buf, const nx16_xenc *d_jobs, size_t njobs, nx16_xenc_res *d_res, hipStream_t s, const nx16_xenc *h_jobs) {
    if (!njobs) return HG_OK;
    static const bool want_big = !(getenv("HG_XENC_BIG") && atoi(getenv("HG_XENC_BIG")) == 0);
    std::vector<uint32_t> big;
    if (h_jobs && want_big) for (size_t j = 0; j < njobs; j++) if (h_jobs[j].n >= hgy::BIG_MIN) big.push_back((uint32_t)j);
    if (!big.empty()) {
        const int rc = ensure_scratch(ctx, 14, big.size() * 4 + 64);
        if (rc) return rc;
        if (hipMemcpyAsync(ctx->d_scratch[14], big.data(), big.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        hipLaunchKernelGGL(hgy::nx16_xenc_big_kernel, dim3((unsigned)big.size()), dim3(hgy::BIG_WAVES * 64), 0, s, (uint8_t *)d_buf, d_jobs, (const uint32_t *)ctx->d_scratch[14], d_res);
    }
    if (big.size() < njobs) {
        size_t wgs = (njobs + hgy::WAVES - 1) / hgy::WAVES;
        const size_t maxw = (size_t)ctx->cus * 8;
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL(hgy::nx16_xenc_kernel, dim3((unsigned)wgs), dim3(hgy::WAVES * 64), 0, s, (uint8_t *)d_buf, d_jobs, (uint32_t)njobs, d_res,
                           big.empty() ? 0xffffffffu : hgy::BIG_MIN);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

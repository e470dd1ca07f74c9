// bgzf_deflate.hip -- BGZF block deflate for MI355X (gfx950 / CDNA4).
//
// Replaces the writer-side worker of htslib, bgzf_encode_func -> bgzf_compress (reference
// bgzf.c:1330-1341, 561-683): raw DEFLATE of one <= 0xff00-byte block, BGZF header, CRC-32 and
// ISIZE trailer, stored-block fallback when deflate does not shrink the block (bgzf.c:652-667),
// the canonical 28-byte EOF block for empty input (bgzf.c:563-569).  Compressed bytes are not
// expected to equal zlib's or libdeflate's (the reference's own tests accept any valid stream,
// test/test.pl:1238-1260); they must decode bit-exactly with stock htslib and stay within a few
// per cent of zlib level 6 in size.
//
// Mapping (CDNA4 first, not a port of a CPU deflate):
//   * one BGZF block per 256-thread workgroup; the input is staged in LDS as a RING of the last 36 KiB (32 KiB window + look-ahead, topped
//     up 2 KiB at a time while the chunks advance), every later access (hashing, match extension, literals, CRC) is an LDS access.  53 KiB of
//     LDS and <= 168 VGPRs per workgroup -> 3 workgroups (12 waves) per CU, persistent workgroups pull block tickets from a global counter;
//   * match finding is position-parallel: the block is walked in chunks of 256 positions, one position per lane; a lane hashes its 4 bytes,
//     reads the WAYS (4 / 8 / 12 by level) most recent earlier positions with that hash from a set-associative table in LDS and compares
//     each with its own bytes over 32 (24) bytes in straight-line code (aligned dword reads, v_alignbyte, xor, v_ffbl, one unsigned minimum:
//     the kernel is bound by instruction ISSUE, so the instruction count per candidate is what matters), then the nearest survivor alone,
//     16 bytes per round; then the chunk's positions are inserted (LDS atomics pick the way).  Candidates are always from earlier chunks;
//     distance 1 is probed directly;
//   * the lazy parse (take a match unless the next position has a longer one) is a chain over positions; it is resolved per chunk with 8
//     rounds of pointer jumping in LDS, then the chosen tokens are compacted in order with ballots and appended to a per-workgroup token
//     list in HBM while LDS atomics build the litlen/distance histograms;
//   * the Huffman phase (<= 286 + 30 + 19 symbols) is a workgroup-collective routine (deflate_huff_wg.h): package-merge code lengths, one
//     binary search per item and level; canonical codes; the header's run-length coding per run start -- unit-tested on the host;
//   * bit packing is a prefix scan of code lengths over 256 tokens per step; lanes OR their <= 48 bits into an LDS staging window with
//     ds_or and the finished dwords leave with coalesced stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include "deflate_huff.h"
#include "deflate_huff_wg.h"
#include <type_traits>

namespace hgd {

#ifdef HG_PROFILE
__device__ unsigned long long g_dprof[16];   // 0 total, 1 stage+crc, 2 match+parse, 3 huffman, 4 emit, 5 blocks
#define HD_T0(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define HD_TACC(slot, var) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); dacc[slot] += n_ - var; var = n_; } while (0)
#ifdef HG_PROFILE_HUFF   /* slots 6..9 = sub-phases of the Huffman construction instead of the match detail */
#define HD_TACCM(slot, var) do { } while (0)
#define HD_TACCH(slot, var) HD_TACC(slot, var)
#else
#define HD_TACCM(slot, var) HD_TACC(slot, var)
#define HD_TACCH(slot, var) do { } while (0)
#endif
#else
#define HD_T0(var) do { } while (0)
#define HD_TACC(slot, var) do { } while (0)
#define HD_TACCM(slot, var) do { } while (0)
#define HD_TACCH(slot, var) do { } while (0)
#endif

#ifndef HG_DEF_HB
#define HG_DEF_HB 9
#endif
constexpr int WG = 256;
constexpr int HB = HG_DEF_HB;                 // hash buckets = 2^HB
constexpr int MAX_WAYS = 12;                   // most recent positions kept per bucket (the level picks 4, 8 or 12 of them)
#ifndef HG_DEF_KERNEL_ATTR
#define HG_DEF_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))   // 168 VGPRs: three wavefronts per SIMD (the LDS allows three workgroups per CU)
#endif
#ifndef HG_DEF_WGS_PER_CU
#define HG_DEF_WGS_PER_CU 3
#endif
constexpr uint32_t MAX_IN = 0xff00u;           // BGZF_BLOCK_SIZE (htslib/bgzf.h:50)
#ifndef HG_LS_G0
#define HG_LS_G0 32      // bytes over which the candidates of the first group are compared in lock step
#endif
#ifndef HG_LS_G1
#define HG_LS_G1 24      // ... of the second group (12 ways): 24 costs 0.02 % of size and gives 3.5 % of speed (r02 sweep)
#endif
constexpr uint32_t LOCKSTEP = (uint32_t)HG_LS_G0;
constexpr uint32_t TOO_FAR = 4096u;            // a 3-byte match this far away costs more than 3 literals

struct Huff {                                  // overlays the hash table once matching is done
    uint32_t obuf[520];                        // bit-packing staging window (dwords)
    uint16_t ll_code[288];
    uint16_t d_code[32];
    alignas(4) uint8_t ll_len[288];
    alignas(4) uint8_t d_len[32];
};

// The staged input is a RING of the last 36 KiB (DEFLATE looks back 32 KiB): position x of the block lives at byte x mod RING, the ring's first
// MIRROR bytes are kept twice (again behind its end) so that a read that starts near the end runs straight on.  With the whole 64 KiB block
// staged the kernel held 80 KiB of LDS and two workgroups per CU; with the ring it holds 53 KiB: three per CU, +33 % (measured on a build that
// simply staged half blocks).  A block of up to 36 KiB never wraps; behind that the ring is topped up 2 KiB at a time while the chunks advance.
constexpr uint32_t RING = 36864u, MIRROR = 64u, REFILL = 2048u, AHEAD = 544u;   // AHEAD: a chunk reads up to 256 + 258 + 19 bytes past its start
static_assert(RING % 16 == 0 && REFILL == WG * 8 && RING >= 32768u + WG + AHEAD + REFILL, "ring geometry");
__device__ __forceinline__ uint32_t ro(uint32_t pos) { const uint32_t w = pos - RING; return w < pos ? w : pos; }   // pos mod RING for pos < 2 * RING

struct Lds {
    uint32_t in32[(RING + MIRROR) / 4];
    union {
        uint16_t tab[(1 << HB) * MAX_WAYS];
        Huff h;
    } u;
    uint32_t cnt32[(1 << HB) / 4];             // 8-bit insertion counters, 4 per dword
    uint32_t lfreq[288];
    uint32_t dfreq[32];
    uint16_t mlen[WG + 2];
    uint16_t mdist[WG];
    uint16_t jump[WG];
    uint16_t jump2[WG];
    uint8_t mark[WG];
    uint32_t wsum[8];
    uint32_t carry_next;
    uint32_t misc[7];
};
static_assert(sizeof(Huff) <= sizeof(uint16_t) * (1 << HB) * MAX_WAYS, "Huff scratch must fit in the hash table");
static_assert(sizeof(hgdef::HuffWG) <= RING, "the collective Huffman phase works in the staged input's LDS once matching is done");
static_assert(sizeof(Lds) <= 53 * 1024, "three workgroups per CU");

// Unaligned reads of the staged input: aligned dword reads glued with v_alignbyte.  (Tried in round 3: gfx950 runs the LDS in unaligned-access
// mode and the compiler emits ONE ds_read_b64 / b128 for a byte-aligned 8 / 16-byte read -- a third of the LDS instructions -- but a misaligned
// wide read is slow in the LDS itself: level 6 went from 9.3 to 8.2 GB/s, level 1 from 22.2 to 20.6.)
__device__ __forceinline__ uint32_t load4(const uint32_t *in32, uint32_t off) {
    uint32_t lo = in32[off >> 2], hi = in32[(off >> 2) + 1];
    return __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
}
__device__ __forceinline__ unsigned long long load8(const uint32_t *in32, uint32_t off) {
    const uint32_t w0 = in32[off >> 2], w1 = in32[(off >> 2) + 1], w2 = in32[(off >> 2) + 2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, off & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, off & 3u);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void load16(const uint32_t *in32, uint32_t off, unsigned long long &a, unsigned long long &b) {
    const uint32_t *q = in32 + (off >> 2);
    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], sh = off & 3u;
    const uint32_t r0 = __builtin_amdgcn_alignbyte(w1, w0, sh), r1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
    const uint32_t r2 = __builtin_amdgcn_alignbyte(w3, w2, sh), r3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
    a = ((unsigned long long)r1 << 32) | r0; b = ((unsigned long long)r3 << 32) | r2;
}
// NB / 4 dwords of the string at byte offset `off`
template <int NB>
__device__ __forceinline__ void load_string(const uint32_t *in32, uint32_t off, uint32_t (&out)[NB / 4]) {
    const uint32_t *q = in32 + (off >> 2);
    const uint32_t sh = off & 3u;
    uint32_t w[NB / 4 + 1];
#pragma unroll
    for (int j = 0; j <= NB / 4; j++) w[j] = q[j];
#pragma unroll
    for (int j = 0; j < NB / 4; j++) out[j] = __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh);
}
// Length of the common prefix of the string at `c` and the NB bytes in own[], at most NB: straight-line code, no branches -- NB + 4 bytes read
// as aligned dwords, every dword shifted into place and compared, the first difference picked with selects from the last dword down.
// (The lock-step rounds of rounds 1-2 did this 8 or 16 bytes at a time inside a divergent loop; the compiler turned every candidate's
// bookkeeping into exec-mask branches -- about 330 issue slots per round of nine candidates, a third of them scalar.  The kernel is bound by
// instruction issue (PMC: 1600 VALU + 800 SALU + 200 LDS instructions per wavefront and chunk), so the instructions are what had to go.)
template <int NB>
__device__ __forceinline__ uint32_t common_prefix_fixed(const uint32_t *in32, uint32_t c, const uint32_t (&own)[8]) {
    const uint32_t *q = in32 + (c >> 2);
    const uint32_t sh = c & 3u;
    uint32_t w[NB / 4 + 1];
#pragma unroll
    for (int j = 0; j <= NB / 4; j++) w[j] = q[j];
    // position of the first differing bit: v_ffbl_b32 answers -1 for "no bit set", OR-ing the dword's bit offset into that leaves it -1, so a
    // plain unsigned minimum over the dwords finds the first difference (4.5 instructions per dword, no condition registers)
    uint32_t m = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < NB / 4; j++) {
        const uint32_t x = __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh) ^ own[j];
        uint32_t f;
        asm("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(x));
        f |= (uint32_t)(32 * j);
        m = f < m ? f : m;
    }
    m >>= 3;
    return m < (uint32_t)NB ? m : (uint32_t)NB;
}
__device__ __forceinline__ uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32 - HB); }

// longest common prefix of the strings at a and b (a < b), at most maxl bytes
__device__ __forceinline__ uint32_t match_len(const uint32_t *in32, uint32_t a, uint32_t b, uint32_t maxl) {
    uint32_t l = 0;
    while (l < maxl) {
        uint32_t x = load4(in32, a + l) ^ load4(in32, b + l);
        if (x) { l += (uint32_t)__builtin_ctz(x) >> 3; break; }
        l += 4;
    }
    return l < maxl ? l : maxl;
}

// CRC-32 register after the n bytes at the start of the ring (no wrap: n <= RING), all 256 threads; valid in thread 0.  `first`: the bytes are the
// start of the message (register preset to all ones); the final inversion is the caller's.
__device__ uint32_t wg_crc32_raw(Lds &S, uint32_t n, int tid, bool first) {
    using namespace hg;
    if (n == 0) return first ? 0xffffffffu : 0u;
    uint32_t per = (n + 255u) >> 8;
    int k = 2;
    while ((1u << k) < per) k++;
    const uint32_t K = 1u << k;
    long long beg = (long long)n - (long long)(256 - tid) * (long long)K;
    long long end = beg + (long long)K;
    if (beg < 0) beg = 0;
    if (end < 0) end = 0;
    uint32_t len = (uint32_t)(end - beg), q = (uint32_t)beg;
    uint32_t c = (beg == 0 && end > 0 && first) ? 0xffffffffu : 0u;
    const uint8_t *in8 = (const uint8_t *)S.in32;
    uint32_t head = len & 3u;
    for (uint32_t i = 0; i < head; i++) c = crc_byte(c, in8[q + i]);
    q += head; len -= head;
    for (uint32_t i = 0; i < len; i += 4) c = crc_word(c, load4(S.in32, q + i));
    const int lane = tid & 63;
#pragma unroll
    for (int s = 0; s < 6; s++) {
        uint32_t other = (uint32_t)__shfl_xor((int)c, 1 << s, 64);
        uint32_t m = g_crc.xpow[k + s];
        bool left = ((lane >> s) & 1) == 0;
        uint32_t a = left ? c : other, b = left ? other : c;
        c = crc_mulmod(a, m) ^ b;
    }
    if (lane == 0) S.wsum[4 + (tid >> 6)] = c;
    __syncthreads();
    uint32_t r = 0;
    if (tid == 0) {
        uint32_t X = g_crc.xpow[k + 6];
        r = S.wsum[4];
        r = crc_mulmod(r, X) ^ S.wsum[5];
        r = crc_mulmod(r, X) ^ S.wsum[6];
        r = crc_mulmod(r, X) ^ S.wsum[7];
    }
    __syncthreads();
    return r;
}

// Bit packer: every thread contributes `nb` (<= 56) bits `v` (LSB first) in thread order.
// *bitpos is the absolute bit offset in the output slot; complete dwords are flushed.
__device__ __forceinline__ void pack_bits(Lds &S, uint32_t *out32, uint32_t &bitpos, uint64_t v, uint32_t nb, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    // inclusive scan of nb inside the wave
    uint32_t x = nb;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint32_t y = (uint32_t)__shfl_up((int)x, s, 64);
        if (lane >= s) x += y;
    }
    if (lane == 63) S.wsum[wave] = x;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { uint32_t t = S.wsum[w]; if (w < wave) base += t; total += t; }
    const uint32_t my = bitpos + base + x - nb;            // my first bit
    const uint32_t w0 = bitpos >> 5;                        // first dword of the staging window
    uint32_t *ob = S.u.h.obuf;
    if (nb) {
        uint32_t wi = (my >> 5) - w0, sh = my & 31u;
        uint64_t lo = v << sh;
        atomicOr(&ob[wi], (uint32_t)lo);
        if (sh + nb > 32) atomicOr(&ob[wi + 1], (uint32_t)(lo >> 32));
        if (sh + nb > 64) atomicOr(&ob[wi + 2], (uint32_t)(v >> (64 - sh)));
    }
    __syncthreads();
    const uint32_t endbit = bitpos + total;
    const uint32_t ndone = (endbit >> 5) - w0;              // complete dwords
    for (uint32_t i = tid; i < ndone; i += WG) out32[w0 + i] = ob[i];
    uint32_t carry = ob[ndone];
    __syncthreads();
    for (uint32_t i = tid; i <= ndone + 1 && i < 520; i += WG) ob[i] = 0;
    __syncthreads();
    if (tid == 0) ob[0] = carry;
    bitpos = endbit;
    __syncthreads();
}

// Compression levels (bgzf.c:583-585 maps 1..9 onto libdeflate levels; zlib uses them directly): the level picks the
// search effort -- WAYS candidates per hash bucket and the parse --
//   1-3: 4 candidates, greedy parse            (fastest, largest)
//   4-5: 8 candidates, one-step lazy parse
//   6-9: 12 candidates, two-step lazy parse, 4-byte matches farther than 2 KiB dropped (they cost more than literals)
template <int WAYS, int LAZY>
__global__ __launch_bounds__(WG) HG_DEF_KERNEL_ATTR
void bgzf_deflate_kernel(const uint8_t *__restrict__ plain, const hg_bgzf_desc *__restrict__ desc, uint32_t nblocks,
                         uint8_t *slots, uint32_t *clen_out, uint32_t *tokbuf, unsigned int *ticket, int level,
                         int mode, uint32_t *crc_out) {
    // mode 0: complete BGZF blocks.  mode 1: bare deflate blocks that concatenate into ONE stream
    // (CRAM GZIP blocks, zlib_mem_deflate cram/cram_io.c:1222-1277): desc[b].clen bit 0 = this chunk is
    // the last of its stream (BFINAL); every other chunk ends with an empty stored block (the zlib
    // "sync flush" marker 00 00 FF FF) so that chunks are byte aligned; crc_out[b] = CRC-32 of the chunk.
    __shared__ Lds S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *tok = tokbuf + (size_t)blockIdx.x * 65536u;
    const uint8_t *in8 = (const uint8_t *)S.in32;

    for (;;) {
        __syncthreads();
        if (tid == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t b = S.misc[0];
        if (b >= nblocks) break;
        const hg_bgzf_desc dsc = desc[b];
        const uint32_t n = dsc.ulen > MAX_IN ? MAX_IN : dsc.ulen;     // API guarantees ulen <= 0xff00
        const uint8_t *src = plain + dsc.uoff;
        uint8_t *o8 = slots + dsc.coff;
        uint32_t *o32 = (uint32_t *)o8;

        const bool last_chunk = (dsc.clen & 1u) != 0;
        if (n == 0 && mode == 1) {
            const uint8_t fin[5] = {1, 0, 0, 0xff, 0xff};
            if (last_chunk && tid < 5) o8[tid] = fin[tid];
            if (tid == 0) { clen_out[b] = last_chunk ? 5u : 0u; if (crc_out) crc_out[b] = 0; }
            continue;
        }
        if (n == 0) {                                      // canonical EOF block (bgzf.c:566)
            const uint8_t eofb[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0,
                                      0, 0, 0, 0, 0, 0, 0, 0};
            if (tid < 28) o8[tid] = eofb[tid];
            if (tid == 0) clen_out[b] = 28;
            continue;
        }
#ifdef HG_PROFILE
        unsigned long long dacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
        HD_T0(tp); HD_T0(tb);
        // ---- stage the block's first RING bytes (coalesced 16-byte loads) and take the CRC from LDS ----------------
        // bytes [from, to) of the block into the ring, zeros behind the block's end; from and to are multiples of 16
        auto stage = [&](uint32_t from, uint32_t to) {
            for (uint32_t i = from + (uint32_t)tid * 16u; i < to; i += WG * 16u) {
                uint4 w = {0, 0, 0, 0};
                if (i + 16u <= n) __builtin_memcpy(&w, src + i, 16);
                else if (i < n) {
                    uint8_t t[16];
                    for (int k = 0; k < 16; k++) t[k] = i + k < n ? src[i + k] : 0;
                    __builtin_memcpy(&w, t, 16);
                }
                const uint32_t r = ro(i) >> 2;
                S.in32[r] = w.x; S.in32[r + 1] = w.y; S.in32[r + 2] = w.z; S.in32[r + 3] = w.w;
                if (r < MIRROR / 4) { S.in32[RING / 4 + r] = w.x; S.in32[RING / 4 + r + 1] = w.y; S.in32[RING / 4 + r + 2] = w.z; S.in32[RING / 4 + r + 3] = w.w; }
            }
        };
        const uint32_t pad_end = (n + 48u + 15u) & ~15u;                  // the reads behind the last position find zeros
        uint32_t hi = pad_end < RING ? pad_end : RING;                    // bytes of the block staged so far
        uint32_t crc;
        if (n <= RING) {
            stage(0, hi);
            __syncthreads();
            crc = ~wg_crc32_raw(S, n, tid, true);                         // valid in thread 0
        } else {
            // a block longer than the ring: CRC of its tail first (staged at the ring's start), then of its head, which stays for the matching
            stage(RING, pad_end);
            __syncthreads();
            const uint32_t tail = wg_crc32_raw(S, n - RING, tid, false);
            stage(0, RING);
            __syncthreads();
            uint32_t head = wg_crc32_raw(S, RING, tid, true);
            if (tid == 0) {                                                // head * x^(8 * tail bytes) + tail
                const uint32_t lt = n - RING;
                for (int j = 0; (lt >> j) != 0u; j++) if ((lt >> j) & 1u) head = hg::crc_mulmod(head, hg::g_crc.xpow[j]);
            }
            crc = ~(head ^ tail);
        }
        for (int i = tid; i < (1 << HB) * MAX_WAYS / 2; i += WG) ((uint32_t *)S.u.tab)[i] = 0xffffffffu;
        for (int i = tid; i < (1 << HB) / 4; i += WG) S.cnt32[i] = 0;
        for (int i = tid; i < 288; i += WG) S.lfreq[i] = 0;
        if (tid < 32) S.dfreq[tid] = 0;
        if (tid == 0) { S.mlen[WG] = 0; S.mlen[WG + 1] = 0; }
        __syncthreads();

        HD_TACC(1, tp);
        uint32_t ntok = 0;
        if (level != 0) {
            // ---- match finding + lazy parse, 256 positions per step -----------------------
            uint32_t carry = 0;                            // chunk-relative position of the next token
            for (uint32_t c0 = 0; c0 < n; c0 += WG) {
                HD_T0(tq);
                const uint32_t p = c0 + (uint32_t)tid;
                // top up the ring for the NEXT chunk: 8 bytes per thread, requested now, stored behind this chunk's search (the bytes they
                // replace lie more than 32 KiB before the next chunk)
                const bool refill = hi < pad_end && hi < c0 + WG + AHEAD;
                uint2 rf = {0, 0};
                if (refill) {
                    const uint32_t q = hi + (uint32_t)tid * 8u;
                    if (q + 8u <= n) __builtin_memcpy(&rf, src + q, 8);
                    else if (q < n) {
                        uint8_t t[8];
                        for (int k = 0; k < 8; k++) t[k] = q + k < n ? src[q + k] : 0;
                        __builtin_memcpy(&rf, t, 8);
                    }
                }
                uint32_t best = 0, bd = 0, h = 0;
                const bool hashable = p + 4u <= n;
                {   // every lane walks through the search (the long-match phase is a wavefront's joint work); a position that cannot start a
                    // match (the block's last three bytes, the tail of the last chunk) simply has no candidates
                    const uint32_t maxl = !hashable ? 0u : n - p < 258u ? n - p : 258u;
                    uint32_t own[8];
                    load_string<32>(S.in32, hashable ? ro(p) : 0u, own);
                    static_assert(LOCKSTEP <= 32 && HG_LS_G1 <= 32 && LOCKSTEP % 4 == 0 && HG_LS_G1 % 4 == 0, "own[] holds 32 bytes");
                    const uint32_t cur = own[0];
                    h = hashable ? hash4(cur) : 0u;
                    // Candidates are evaluated in groups of up to 8 table entries (+ distance 1 with the first group: runs are never in this
                    // chunk's table) to bound the registers held per lane: one group for 4 or 8 ways, 8 + 4 for 12.  All of this is straight-line
                    // code -- the kernel is bound by instruction issue, every exec-mask branch costs scalar instructions on top of both sides.
                    // The best match so far is one key: length << 16 | (32768 - distance): a maximum picks the longer match and, among equals,
                    // the nearer one.
                    const uint32_t *row32 = (const uint32_t *)&S.u.tab[h * MAX_WAYS];           // 24-byte rows, 8-byte aligned
                    uint32_t bestkey = 0;
                    auto group = [&](auto nc_, auto ls_, auto first_, auto d1_) {
                        constexpr int NC = decltype(nc_)::value, LS = decltype(ls_)::value, FIRST = decltype(first_)::value;
                        constexpr bool D1 = decltype(d1_)::value;
                        constexpr int G = NC + (D1 ? 1 : 0);
                        static_assert(NC % 4 == 0 && FIRST % 4 == 0, "the ways are read eight bytes at a time");
                        uint32_t cw[NC / 2];
#pragma unroll
                        for (int k = 0; k < NC / 2; k += 2) {
                            const uint2 rw = *(const uint2 *)(row32 + FIRST / 2 + k);
                            cw[k] = rw.x; cw[k + 1] = rw.y;
                        }
                        uint32_t cand[G], dist[G];                                               // dist 0 = no candidate
#pragma unroll
                        for (int w = 0; w < NC; w++) {
                            const uint32_t c = (cw[w >> 1] >> ((w & 1) * 16)) & 0xffffu;
                            const bool ok = hashable && c != 0xffffu && p - c <= 32768u;         // 32 KiB window
                            cand[w] = ok ? c : 0u;
                            dist[w] = ok ? p - c : 0u;
                        }
                        if constexpr (D1) { const bool ok = hashable && p >= 1u; cand[NC] = ok ? p - 1u : 0u; dist[NC] = ok ? 1u : 0u; }
                        // every candidate over the first LS bytes (common_prefix_fixed); the nearest one that gets through them goes on alone
                        uint32_t near = 0xffffffffu;
#pragma unroll
                        for (int w = 0; w < G; w++) {
                            uint32_t l = common_prefix_fixed<LS>(S.in32, ro(cand[w]), own);
                            l = dist[w] ? l : 0u;
                            if (l >= (uint32_t)LS) near = dist[w] < near ? dist[w] : near;
                            l = l < maxl ? l : maxl;
                            const uint32_t key = (l << 16) | (32768u - dist[w]);
                            bestkey = key > bestkey ? key : bestkey;
                        }
                        // long-match phase: only the nearest survivor is extended (16 bytes per dependent round); the others keep the LS bytes
                        // they have proven
                        if (near != 0xffffffffu && (uint32_t)LS < maxl) {
                            const uint32_t c = p - near;
                            uint32_t l = (uint32_t)LS;
                            while (l < maxl) {
                                unsigned long long a0, a1, b0, b1;
                                load16(S.in32, ro(c + l), a0, a1); load16(S.in32, ro(p + l), b0, b1);
                                const unsigned long long x0 = a0 ^ b0, x1 = a1 ^ b1;
                                if (x0) { l += (uint32_t)__builtin_ctzll(x0) >> 3; break; }
                                if (x1) { l += 8u + ((uint32_t)__builtin_ctzll(x1) >> 3); break; }
                                l += 16;
                            }
                            l = l < maxl ? l : maxl;
                            const uint32_t key = (l << 16) | (32768u - near);
                            bestkey = key > bestkey ? key : bestkey;
                        }
                    };
                    constexpr int GW = WAYS < 8 ? WAYS : 8;
                    group(std::integral_constant<int, GW>{}, std::integral_constant<int, (int)LOCKSTEP>{}, std::integral_constant<int, 0>{}, std::true_type{});
                    if constexpr (WAYS > 8)
                        group(std::integral_constant<int, WAYS - 8>{}, std::integral_constant<int, HG_LS_G1>{}, std::integral_constant<int, 8>{}, std::false_type{});
                    best = bestkey >> 16; bd = 32768u - (bestkey & 0xffffu);
                    if (best < 3u || (best == 3u && bd > TOO_FAR) || (LAZY >= 2 && best == 4u && bd > 2048u)) best = 0;
                }
                S.mlen[tid] = (uint16_t)best;
                S.mdist[tid] = (uint16_t)bd;
                HD_TACCM(6, tq);
                __syncthreads();
                HD_TACCM(7, tq);
                // Publish this chunk's positions.  The slot a position gets inside its bucket comes from an atomic
                // counter, so the four waves insert one after the other (a wave's own LDS atomics resolve in lane
                // order): the table -- and with it the compressed bytes -- is the same on every run.
                auto publish = [&]() {
                    if (hashable) {
                        const uint32_t sh = (h & 3u) * 8u;
                        const uint32_t old = atomicAdd(&S.cnt32[h >> 2], 1u << sh);
                        S.u.tab[h * MAX_WAYS + ((old >> sh) & 0xffu) % (uint32_t)WAYS] = (uint16_t)p;
                    }
                };
                if (wave == 0) publish();
                if (refill) {                                              // (every wave is past its search: the barrier above)
                    const uint32_t r = ro(hi + (uint32_t)tid * 8u) >> 2;
                    S.in32[r] = rf.x; S.in32[r + 1] = rf.y;
                    if (r < MIRROR / 4) { S.in32[RING / 4 + r] = rf.x; S.in32[RING / 4 + r + 1] = rf.y; }
                    hi += REFILL;
                }
                __syncthreads();
                if (wave == 1) publish();
                // ---- lazy parse by pointer jumping -----------------------------------------
                const bool live = p < n;
                // lazy parse: a match yields to a longer one starting at the next position (and, two-step, to one longer by
                // two or more at the position after that); mlen[WG], mlen[WG + 1] = 0: the chunk's last positions cannot look ahead
                const bool take = best >= 3u && !(LAZY >= 1 && (uint32_t)S.mlen[tid + 1] > best) &&
                                  !(LAZY >= 2 && (uint32_t)S.mlen[tid + 2] > best + 1u);
                const uint32_t step = take ? best : 1u;
                uint32_t nx = (uint32_t)tid + step;
                if (nx > WG) nx = WG;
                bool marked = false;
                if (carry < WG) {
                    // one barrier per round: the doubled pointers alternate between two arrays (a round reads one and writes the other), the marks
                    // only ever go from 0 to 1 and a round's marks are complete at its barrier
                    S.jump[tid] = (uint16_t)nx;
                    S.mark[tid] = (uint8_t)(((uint32_t)tid == carry) && live);
                    __syncthreads();
#pragma unroll 1
                    for (int r = 0; r < 8; r += 2) {
                        {
                            const uint32_t j = S.jump[tid];
                            if (S.mark[tid] && j < WG) S.mark[j] = 1;
                            S.jump2[tid] = (uint16_t)(j < WG ? (uint32_t)S.jump[j] : (uint32_t)WG);
                            __syncthreads();
                        }
                        {
                            const uint32_t j = S.jump2[tid];
                            if (S.mark[tid] && j < WG) S.mark[j] = 1;
                            S.jump[tid] = (uint16_t)(j < WG ? (uint32_t)S.jump2[j] : (uint32_t)WG);
                            __syncthreads();
                        }
                    }
                    marked = S.mark[tid] != 0 && live;
                    if (marked && (uint32_t)tid + step >= WG) S.carry_next = (uint32_t)tid + step - WG;
                    if (tid == 0 && c0 + WG >= n) S.carry_next = 0;   // last chunk: value unused
                } else if (tid == 0) {
                    S.carry_next = carry - WG;
                }
                HD_TACCM(8, tq);
                // ---- compact the chosen tokens, in order -----------------------------------
                const unsigned long long bal = __ballot(marked);
                if (lane == 0) S.wsum[wave] = (uint32_t)__popcll(bal);
                __syncthreads();
                if (wave == 2) publish();
                uint32_t base = ntok, total = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) { uint32_t t = S.wsum[w]; if (w < wave) base += t; total += t; }
                if (marked) {
                    const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    if (take) {
                        tok[idx] = 0x80000000u | ((best - 3u) << 16) | (bd - 1u);
                        uint32_t s, xb, xv;
                        hgdef::len_symbol(best, s, xb, xv); atomicAdd(&S.lfreq[257 + s], 1u);
                        hgdef::dist_symbol(bd, s, xb, xv); atomicAdd(&S.dfreq[s], 1u);
                    } else {
                        const uint32_t byte = in8[ro(p)];
                        tok[idx] = byte;
                        atomicAdd(&S.lfreq[byte], 1u);
                    }
                }
                ntok += total;
                carry = S.carry_next;
                __syncthreads();
                if (wave == 3) publish();
                __syncthreads();
                HD_TACCM(9, tq);
            }
        }
        // ---- choose the block type and build the codes -------------------------------------
        __syncthreads();
        HD_TACC(2, tp);
        uint32_t dyn_bits = 0;
        if (level != 0) {
            // the staged input and the hash table are dead now: the collective Huffman phase (deflate_huff_wg.h) works in the input's LDS,
            // the codes and the bit-packing window in the table's
            Huff &H = S.u.h;
            hgdef::HuffWG &W = *reinterpret_cast<hgdef::HuffWG *>(S.in32);
            if (tid == 0) S.lfreq[256] = 1;                                       // end of block
            HD_T0(th);
            hgdef::wg_code_lengths<WG, 15>(W, S.lfreq, 286, H.ll_len, S.dfreq, 30, H.d_len, tid);
            HD_TACCH(6, th);
            hgdef::wg_assign_codes<WG>(W, H.ll_len, 286, H.ll_code, H.d_len, 30, H.d_code, tid);
            HD_TACCH(7, th);
            hgdef::wg_dynamic_header<WG>(W, H.ll_len, H.d_len, !(mode == 1 && !last_chunk), tid);
            HD_TACCH(8, th);
            if (tid == 0) S.misc[2] = W.hdr_bits;
            __syncthreads();
            {   // size of the dynamic block: one symbol per thread, summed with an LDS atomic
                uint32_t part = 0;
                for (int s = tid; s < 286; s += WG) {
                    uint32_t xb = 0;
                    if (s > 264 && s < 285) xb = (uint32_t)(s - 261) >> 2;
                    part += S.lfreq[s] * (H.ll_len[s] + xb);
                }
                if (tid < 30) part += S.dfreq[tid] * (H.d_len[tid] + (tid < 4 ? 0u : (uint32_t)(tid - 2) >> 1));
                if (part) atomicAdd(&S.misc[2], part);
            }
            __syncthreads();
            dyn_bits = S.misc[2];
        }
        HD_TACC(3, tp);
        const uint32_t dyn_bytes = (dyn_bits + 7u) >> 3;
        const bool stored = level == 0 || dyn_bytes >= n + 5u;
        // ---- BGZF header (BSIZE patched at the end) ------------------------------------------
        const uint32_t hoff = mode == 1 ? 0u : 18u;
        if (mode == 0 && tid < 18) {
            const uint8_t h18[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0, 0};
            o8[tid] = h18[tid];
        }
        uint32_t total_len;
        if (stored) {
            // 01 LEN NLEN data (bgzf.c:573-580, 652-667)
            if (tid == 0) {
                o8[hoff] = (mode == 1 && !last_chunk) ? 0 : 1; o8[hoff + 1] = (uint8_t)n; o8[hoff + 2] = (uint8_t)(n >> 8);
                o8[hoff + 3] = (uint8_t)~n; o8[hoff + 4] = (uint8_t)(~n >> 8);
            }
            for (uint32_t i = tid; i < n; i += WG) o8[hoff + 5 + i] = src[i];      // (the staged copy may hold the Huffman scratch by now)
            total_len = hoff + 5u + n + (mode == 1 ? 0u : 8u);
        } else {
            Huff &H = S.u.h;
            for (int i = tid; i < 520; i += WG) H.obuf[i] = 0;
            __syncthreads();
            uint32_t bitpos = hoff * 8u;
            {   // the dynamic-block header: (value, bit count) items of wg_dynamic_header, one per thread
                const hgdef::HuffWG &W = *reinterpret_cast<const hgdef::HuffWG *>(S.in32);
                const uint32_t nitems = W.nitems;
                for (uint32_t i0 = 0; i0 < nitems; i0 += WG) {
                    const uint32_t i = i0 + (uint32_t)tid;
                    uint32_t nb = 0; uint64_t v = 0;
                    if (i < nitems) { v = W.item_v[i]; nb = W.item_n[i]; }
                    pack_bits(S, o32, bitpos, v, nb, tid);
                }
            }
            // the tokens (+ end-of-block after the last one)
            for (uint32_t i0 = 0; i0 <= ntok; i0 += WG) {
                const uint32_t i = i0 + (uint32_t)tid;
                uint32_t nb = 0; uint64_t v = 0;
                if (i < ntok) {
                    const uint32_t t = tok[i];
                    if (t & 0x80000000u) {
                        uint32_t s, xb, xv;
                        hgdef::len_symbol(((t >> 16) & 0xffu) + 3u, s, xb, xv);
                        v = H.ll_code[257 + s]; nb = H.ll_len[257 + s];
                        v |= (uint64_t)xv << nb; nb += xb;
                        hgdef::dist_symbol((t & 0x7fffu) + 1u, s, xb, xv);
                        v |= (uint64_t)H.d_code[s] << nb; nb += H.d_len[s];
                        v |= (uint64_t)xv << nb; nb += xb;
                    } else {
                        v = H.ll_code[t & 0xffu]; nb = H.ll_len[t & 0xffu];
                    }
                } else if (i == ntok) {
                    v = H.ll_code[256]; nb = H.ll_len[256];
                }
                pack_bits(S, o32, bitpos, v, nb, tid);
            }
            // a chunk that is not the last of its stream continues with an empty stored block:
            // 000 (BFINAL=0, BTYPE=00), pad to a byte, LEN=0000 NLEN=FFFF
            if (mode == 1 && !last_chunk) { uint32_t nb = tid == 0 ? 3u : 0u; pack_bits(S, o32, bitpos, 0, nb, tid); }
            // pad to a byte boundary, then CRC32 + ISIZE as 8 single bytes
            {
                uint32_t nb = 0; uint64_t v = 0;
                if (tid == 0) nb = (8u - (bitpos & 7u)) & 7u;
                pack_bits(S, o32, bitpos, v, nb, tid);
            }
            if (mode == 1 && !last_chunk) {
                { uint32_t nb = tid < 4 ? 8u : 0u; uint64_t v = tid >= 2 ? 0xffu : 0u; pack_bits(S, o32, bitpos, v, nb, tid); }
            }
            total_len = (bitpos >> 3) + (mode == 1 ? 0u : 8u);
            // flush the partial dword that is still in the staging window
            if (tid == 0 && (bitpos & 31u)) o32[bitpos >> 5] = H.obuf[0];
            __syncthreads();
        }
        if (tid == 0) {
            if (mode == 0) {
                uint8_t *t = o8 + total_len - 8;
                for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(n >> (8 * k)); }
                o8[16] = (uint8_t)(total_len - 1u); o8[17] = (uint8_t)((total_len - 1u) >> 8);
            }
            clen_out[b] = total_len;
            if (crc_out) crc_out[b] = crc;
        }
        HD_TACC(4, tp); HD_TACC(0, tb);
#ifdef HG_PROFILE
        if (tid == 0) { atomicAdd(&g_dprof[5], 1ull); for (int k = 0; k < 10; k++) if (k != 5) atomicAdd(&g_dprof[k], dacc[k]); }
#endif
    }
}

// exclusive prefix sum of clen -> packed offsets; one workgroup, sequential over 1024-element tiles
__global__ __launch_bounds__(1024)
void scan_clen_kernel(const uint32_t *__restrict__ clen, uint32_t n, uint64_t *poff, uint64_t *total, int add_eof) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) run = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        uint32_t i = i0 + tid;
        unsigned long long v = i < n ? clen[i] : 0, x = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            unsigned long long y = __shfl_up(x, s, 64);
            if (lane >= s) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        unsigned long long base = run;
        for (int w = 0; w < wave; w++) base += wsum[w];
        if (i < n) poff[i] = base + x - v;
        __syncthreads();
        if (tid == 1023) run = base + x;
        __syncthreads();
    }
    if (tid == 0) *total = run + (add_eof ? 28 : 0);
}

// gather the slots into one contiguous BGZF stream (+ optional EOF block)
__global__ __launch_bounds__(256)
void pack_slots_kernel(const uint8_t *__restrict__ slots, const hg_bgzf_desc *__restrict__ desc,
                       const uint32_t *__restrict__ clen, const uint64_t *__restrict__ poff, uint32_t n,
                       uint8_t *packed, uint64_t cap, const uint64_t *total, int add_eof) {
    for (uint32_t b = blockIdx.x; b < n + (add_eof ? 1u : 0u); b += gridDim.x) {
        if (b == n) {
            const uint8_t eofb[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0,
                                      0, 0, 0, 0, 0, 0, 0, 0};
            uint64_t at = *total - 28;
            if (threadIdx.x < 28 && at + 28 <= cap) packed[at + threadIdx.x] = eofb[threadIdx.x];
            continue;
        }
        const uint8_t *s = slots + desc[b].coff;
        uint8_t *d = packed + poff[b];
        const uint32_t len = clen[b];
        if (poff[b] + len > cap) continue;
        for (uint32_t i = threadIdx.x; i < len; i += 256) d[i] = s[i];
    }
}

}  // namespace hgd

#ifdef HG_PROFILE
extern "C" int hg_debug_get_deflate_profile(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(hgd::g_dprof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hgd::g_dprof), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

namespace hg {

size_t bgzf_deflate_tok_bytes(const hg_ctx *ctx) { return (size_t)ctx->cus * HG_DEF_WGS_PER_CU * 65536 * sizeof(uint32_t); }
// ... of a launch over nblocks blocks: the lists are indexed by workgroup and a launch has min(resident workgroups, blocks) of them
size_t bgzf_deflate_tok_bytes_for(const hg_ctx *ctx, size_t nblocks) {
    const size_t wgs = std::min<size_t>((size_t)ctx->cus * HG_DEF_WGS_PER_CU, nblocks ? nblocks : 1);
    return wgs * 65536 * sizeof(uint32_t);
}

int launch_bgzf_deflate(hg_ctx *ctx, const void *d_plain, const hg_bgzf_desc *d_desc, size_t nblocks, int level,
                        void *d_slots, uint32_t *d_clen, hipStream_t s, int mode, uint32_t *d_crc, void *own_tok) {
    if (nblocks == 0) return HG_OK;
    if (nblocks > 0xffffffffull) return HG_EINVAL;
    size_t wgs = (size_t)ctx->cus * HG_DEF_WGS_PER_CU;
    if (wgs > nblocks) wgs = nblocks;
    const size_t need = bgzf_deflate_tok_bytes(ctx);
    auto kern = level <= 3 ? hgd::bgzf_deflate_kernel<4, 0> : level <= 5 ? hgd::bgzf_deflate_kernel<8, 1> : hgd::bgzf_deflate_kernel<12, 2>;
    if (own_tok) {
        // the caller's own token lists (a pipe of the BGZF writer): nothing is shared with other launches, so launches of different pipes overlap
        // and the next job's workgroups take the CUs the slowest blocks of this one leave idle
        unsigned int *ticket;
        { std::lock_guard<std::mutex> order(*ctx->tok_mu); ticket = next_ticket(ctx); }
        if (hipMemsetAsync(ticket, 0, sizeof(unsigned int), s) != hipSuccess) return HG_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(hgd::WG), 0, s, (const uint8_t *)d_plain,
                           d_desc, (uint32_t)nblocks, (uint8_t *)d_slots, d_clen, (uint32_t *)own_tok, ticket, level, mode, d_crc);
        return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
    }
    {
        std::lock_guard<std::mutex> order(*ctx->tok_mu);
        if (ctx->d_tok_cap < need) {                                   // allocated once (the size depends on the device only)
            if (hipMalloc(&ctx->d_tok, need) != hipSuccess) return HG_ENOMEM;
            ctx->d_tok_cap = need;
        }
    }
    // The context's token lists (ctx->d_tok) are indexed by workgroup, so two launches that use them must not overlap:
    // each waits for the previous one's completion event, whatever streams the callers use.
    {
        std::lock_guard<std::mutex> order(*ctx->tok_mu);
        if (ctx->ev_deflate_used && hipStreamWaitEvent(s, ctx->ev_deflate, 0) != hipSuccess) return HG_ELAUNCH;
        unsigned int *ticket = next_ticket(ctx);
        if (hipMemsetAsync(ticket, 0, sizeof(unsigned int), s) != hipSuccess) return HG_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(hgd::WG), 0, s, (const uint8_t *)d_plain,
                           d_desc, (uint32_t)nblocks, (uint8_t *)d_slots, d_clen, (uint32_t *)ctx->d_tok,
                           ticket, level, mode, d_crc);
        if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
        if (hipEventRecord(ctx->ev_deflate, s) != hipSuccess) return HG_ELAUNCH;
        ctx->ev_deflate_used = 1;
    }
    return HG_OK;
}

int launch_bgzf_pack(hg_ctx *ctx, const void *d_slots, const hg_bgzf_desc *d_desc, const uint32_t *d_clen,
                     size_t nblocks, void *d_packed, size_t cap, uint64_t *d_poff, uint64_t *d_total, int add_eof,
                     hipStream_t s) {
    hipLaunchKernelGGL(hgd::scan_clen_kernel, dim3(1), dim3(1024), 0, s, d_clen, (uint32_t)nblocks, d_poff, d_total,
                       add_eof);
    size_t wgs = nblocks + 1;
    size_t maxw = (size_t)ctx->cus * 16;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgd::pack_slots_kernel, dim3((unsigned)wgs), dim3(256), 0, s, (const uint8_t *)d_slots, d_desc,
                       d_clen, d_poff, (uint32_t)nblocks, (uint8_t *)d_packed, (uint64_t)cap, d_total, add_eof);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

}  // namespace hg

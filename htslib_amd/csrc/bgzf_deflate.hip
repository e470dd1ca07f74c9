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
//   * one BGZF block per 256-thread workgroup; the input is staged in LDS as a RING of the last 22 KiB (window + look-ahead, topped up 2 KiB at a
//     time while the chunks advance), every later access (hashing, match extension, literals, CRC) is an LDS access.  The window is 19 KiB, not
//     DEFLATE's 32: the table below remembers the 12 most recent positions of 512 buckets -- about 6 000 positions -- so nothing older than that
//     is ever proposed (scripts/lzsim3.c: the same size to the fourth digit down to an 8 KiB window on every test input).  38 KiB of LDS and
//     <= 128 VGPRs per workgroup -> 4 workgroups (16 waves) per CU, persistent workgroups pull block tickets from a global counter;
//   * match finding is position-parallel: the block is walked in chunks of 256 positions, one position per lane; a lane hashes its 4 bytes,
//     reads the WAYS (4 / 8 / 12 by level) most recent earlier positions with that hash from a set-associative table in LDS and compares
//     each with its own bytes over 32 bytes in straight-line code (aligned dword reads, v_alignbyte, xor, v_ffbl, one unsigned minimum:
//     the kernel is bound by instruction ISSUE, so the instruction count per candidate is what matters), then the nearest survivor alone,
//     16 bytes per round; then one wavefront inserts the chunk's positions (LDS atomics pick the way).  Candidates are always from earlier
//     chunks; distance 1 is probed directly;
//   * the lazy parse (take a match unless the next position has a longer one) is a chain over positions; every wavefront resolves it for its own
//     64 positions and EVERY entry point at once -- six rounds of pointer doubling in registers (ds_bpermute) -- then the wavefronts' exits are
//     chained through LDS (two barriers per chunk; rounds 1-5: pointer jumping across the workgroup, fourteen); the chosen tokens are appended
//     in order to a per-workgroup token list in HBM while LDS atomics build the litlen/distance histograms;
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
#define HG_DEF_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))   // <= 128 VGPRs: four wavefronts per SIMD (the LDS allows four workgroups per CU)
#endif
#ifndef HG_DEF_WGS_PER_CU
#define HG_DEF_WGS_PER_CU 4
#endif
constexpr uint32_t MAX_IN = 0xff00u;           // BGZF_BLOCK_SIZE (htslib/bgzf.h:50)
constexpr uint32_t TOO_FAR = 4096u;            // a 3-byte match this far away costs more than 3 literals

struct Huff {                                  // codes + bit-packing window of the emit phase
    uint32_t obuf[2][520];                     // bit-packing staging windows (dwords): one being filled, one flushed and zeroed behind it
    uint16_t ll_code[288];
    uint16_t d_code[32];
    alignas(4) uint8_t ll_len[288];
    alignas(4) uint8_t d_len[32];
};

// The staged input is a RING of the last 22 KiB: position x of the block lives at byte x mod RING, the ring's first MIRROR bytes are kept twice
// (again behind its end) so that a read that starts near the end runs straight on.  A block of up to 22 KiB never wraps; behind that the ring is
// topped up 2 KiB at a time while the chunks advance.  History: the whole 64 KiB block staged (rounds 1-2: 80 KiB of LDS, two workgroups per CU),
// a 36 KiB ring for DEFLATE's full 32 KiB look-back (rounds 3-5: 53 KiB, three per CU, +33 %), and now a ring for the look-back the hash table can
// actually offer (WINDOW): the table keeps 12 positions in each of 512 buckets, ~6 000 positions in all, so a candidate farther than ~10 KiB back
// practically never survives in it -- scripts/lzsim3.c gives the same compressed size to the fourth digit for windows from 32 KiB down to 8 KiB on
// the BAM, the 41-level-quality BAM and the FASTQ inputs.  38 KiB of LDS: four workgroups per CU.
constexpr uint32_t RING = 22528u, MIRROR = 64u, REFILL = 2048u, AHEAD = 544u;   // AHEAD: a chunk reads up to 256 + 258 + 19 bytes past its start
constexpr uint32_t WINDOW = RING - (WG + AHEAD + REFILL);                        // how far back a match may start: 19 680
static_assert(RING % 32 == 0 && REFILL == WG * 8 && WINDOW >= 16384u && WINDOW <= 32768u && 3u * RING >= MAX_IN + 64u,
              "ring geometry (a top-up is stored in 32-byte pieces that must not straddle the ring's end; a block is at most three ring fills)");
__device__ __forceinline__ uint32_t ro(uint32_t pos) {                           // pos mod RING for pos < 3 * RING
    uint32_t w = pos - RING; pos = w < pos ? w : pos;
    w = pos - RING; return w < pos ? w : pos;
}

struct Lds {
    union {
        struct {                               // matching phase
            uint32_t in32[(RING + MIRROR) / 4];
            uint16_t tab[(1 << HB) * MAX_WAYS];
        } m;
        struct {                               // Huffman + emit phases (the staged input and the hash table are dead by then)
            hgdef::HuffWG w;
            Huff h;
        } e;
    } u;
    uint32_t cnt32[(1 << HB) / 4];             // 8-bit insertion counters, 4 per dword
    uint32_t lfreq[288];
    uint32_t dfreq[32];
    uint16_t mlen[WG + 2];                     // match length a position offers to the parse (0 = literal)
    uint16_t hk[WG];                           // hash bucket of a position (for the wavefront that publishes the chunk)
    uint16_t exitp[WG];                        // final parse: where the walk from a position leaves its wavefront's 64 | tokens on the way << 9
    uint32_t wsum[8];
    uint32_t misc[8];
};
// gfx950 hands out its 160 KiB of LDS in granules of 1280 bytes (128 per CU): four workgroups per CU get 32 granules each.  (Measured in round 6: a
// 53 892-byte build -- 43 granules where three workgroups have 42 each -- ran two per CU, SQ_WAVE_CYCLES / SQ_BUSY_CYCLES 15.8 instead of 23.5, and
// lost a quarter of its speed.)
static_assert(sizeof(Lds) <= 32 * 1280, "four workgroups per CU");

// Unaligned reads of the staged input: aligned dword reads glued with v_alignbyte.  v_alignbyte_b32 reads only bits [1:0] of its shift operand on gfx950
// (experiments/alignbyte_probe.hip, run in round 6), so the byte offset itself is passed: no `& 3` per string.  (Tried in round 3: gfx950 runs the LDS in unaligned-access
// mode and the compiler emits ONE ds_read_b64 / b128 for a byte-aligned 8 / 16-byte read -- a third of the LDS instructions -- but a misaligned
// wide read is slow in the LDS itself: level 6 went from 9.3 to 8.2 GB/s, level 1 from 22.2 to 20.6.)
__device__ __forceinline__ uint32_t load4(const uint32_t *in32, uint32_t off) {
    uint32_t lo = in32[off >> 2], hi = in32[(off >> 2) + 1];
    return __builtin_amdgcn_alignbyte(hi, lo, off);
}
__device__ __forceinline__ unsigned long long load8(const uint32_t *in32, uint32_t off) {
    const uint32_t w0 = in32[off >> 2], w1 = in32[(off >> 2) + 1], w2 = in32[(off >> 2) + 2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, off), hi = __builtin_amdgcn_alignbyte(w2, w1, off);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void load16(const uint32_t *in32, uint32_t off, unsigned long long &a, unsigned long long &b) {
    const uint32_t *q = in32 + (off >> 2);
    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], sh = off;
    const uint32_t r0 = __builtin_amdgcn_alignbyte(w1, w0, sh), r1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
    const uint32_t r2 = __builtin_amdgcn_alignbyte(w3, w2, sh), r3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
    a = ((unsigned long long)r1 << 32) | r0; b = ((unsigned long long)r3 << 32) | r2;
}
// NB / 4 dwords of the string at byte offset `off`
template <int NB>
__device__ __forceinline__ void load_string(const uint32_t *in32, uint32_t off, uint32_t (&out)[NB / 4]) {
    const uint32_t *q = in32 + (off >> 2);
    const uint32_t sh = off;
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
    const uint32_t sh = c;
    uint32_t w[NB / 4 + 1];
#pragma unroll
    for (int j = 0; j <= NB / 4; j++) w[j] = q[j];
    // position of the first differing bit: v_ffbl_b32 answers -1 for "no bit set", OR-ing the dword's bit offset into that leaves it -1, so a
    // plain unsigned minimum over the dwords finds the first difference (4.5 instructions per dword, no condition registers).
    // (Round 6 tried ONE asm statement for all 35 instructions -- the compiler pads every asm statement with an s_nop, four per candidate -- and lost
    // 3 %: the statement needs all nine dwords before it starts, the per-dword form overlaps the LDS returns with the arithmetic.)
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
    const uint8_t *in8 = (const uint8_t *)S.u.m.in32;
    uint32_t head = len & 3u;
    for (uint32_t i = 0; i < head; i++) c = crc_byte(c, in8[q + i]);
    q += head; len -= head;
    for (uint32_t i = 0; i < len; i += 4) c = crc_word(c, load4(S.u.m.in32, q + i));
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
// *bitpos is the absolute bit offset in the output slot; complete dwords are flushed.  Two staging windows alternate (`win`): a step ORs its bits into
// one, flushes its complete dwords and zeroes them behind the flush, and moves the partial last dword to the head of the OTHER window with one atomic OR --
// two barriers per 256 tokens (rounds 1-5: one window, five barriers: flush, carry, clear, restore).
__device__ __forceinline__ void pack_bits(Lds &S, uint32_t *out32, uint32_t &bitpos, uint32_t &win, uint64_t v, uint32_t nb, int tid) {
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
    uint32_t *ob = S.u.e.h.obuf[win], *nx = S.u.e.h.obuf[win ^ 1u];
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
    for (uint32_t i = tid; i < ndone; i += WG) { out32[w0 + i] = ob[i]; ob[i] = 0; }
    if (tid == 0) { const uint32_t carry = ob[ndone]; ob[ndone] = 0; if (carry) atomicOr(&nx[0], carry); }
    // (the other window is all zero but for that carry: it was cleared behind its own flush, a step ago; the next step's first barrier orders the rest)
    bitpos = endbit;
    win ^= 1u;
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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (the wavefront's number in a scalar register)
    uint32_t *tok = tokbuf + (size_t)blockIdx.x * 65536u;
    const uint8_t *in8 = (const uint8_t *)S.u.m.in32;

    for (;;) {
        __syncthreads();
        if (tid == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t b = S.misc[0];
        if (b >= nblocks) break;
        const hg_bgzf_desc dsc = desc[b];
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dsc.ulen > MAX_IN ? MAX_IN : dsc.ulen));     // API guarantees ulen <= 0xff00 (in a scalar register: the parse walk loops on it)
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
        unsigned long long dacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
        HD_T0(tp); HD_T0(tb);
        // ---- stage the block through the ring, one ring fill at a time from the LAST piece to the first (coalesced 16-byte loads), taking each
        //      piece's CRC from LDS; the first piece stays for the matching ----------------------------------------------------------------------
        // bytes [from, to) of the block to ring offset (position - base), zeros behind the block's end; from, to and base are multiples of 16
        auto stage = [&](uint32_t from, uint32_t to, uint32_t base) {
            for (uint32_t i = from + (uint32_t)tid * 16u; i < to; i += WG * 16u) {
                uint4 w = {0, 0, 0, 0};
                if (i + 16u <= n) __builtin_memcpy(&w, src + i, 16);
                else if (i < n) {
                    uint8_t t[16];
                    for (int k = 0; k < 16; k++) t[k] = i + k < n ? src[i + k] : 0;
                    __builtin_memcpy(&w, t, 16);
                }
                const uint32_t r = (i - base) >> 2;
                uint32_t *d = S.u.m.in32 + r;
                d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
                if (r < MIRROR / 4) { d = S.u.m.in32 + RING / 4 + r; d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w; }
            }
        };
        const uint32_t pad_end = (n + 48u + 15u) & ~15u;                  // the reads behind the last position find zeros
        uint32_t hi = pad_end < RING ? pad_end : RING;                    // bytes of the block staged so far
        uint32_t crc = 0;                                                  // (thread 0's is the one that counts)
        for (uint32_t base = n > RING ? ((n - 1u) / RING) * RING : 0u;; base -= RING) {
            // CRC of the whole = sum over the pieces of (piece's register) * x^(8 * bytes behind the piece); the all-ones preset goes into the first piece
            const uint32_t len = n - base < RING ? n - base : RING;
            const uint32_t upto = base == 0u ? hi : (base + len + 15u) & ~15u;
            __syncthreads();
            stage(base, upto, base);
            __syncthreads();
            uint32_t part = wg_crc32_raw(S, len, tid, base == 0u);
            if (tid == 0) {
                const uint32_t behind = n - base - len;
                for (int j = 0; (behind >> j) != 0u; j++) if ((behind >> j) & 1u) part = hg::crc_mulmod(part, hg::g_crc.xpow[j]);
                crc ^= part;
            }
            if (base == 0u) break;
        }
        crc = ~crc;
        for (int i = tid; i < (1 << HB) * MAX_WAYS / 2; i += WG) ((uint32_t *)S.u.m.tab)[i] = 0xffffffffu;
        for (int i = tid; i < (1 << HB) / 4; i += WG) S.cnt32[i] = 0;
        for (int i = tid; i < 288; i += WG) S.lfreq[i] = 0;
        if (tid < 32) S.dfreq[tid] = 0;
        if (tid == 0) { S.mlen[WG] = 0; S.mlen[WG + 1] = 0; }
        __syncthreads();

        HD_TACC(1, tp);
        uint32_t ntok = 0;
        if (level != 0) {
            // ---- match finding + parse, 256 positions per step; two barriers per step -----------------------------------------------------------
            uint32_t carry = 0;                            // chunk-relative position of the next token
            uint32_t chunk = 0;
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t *in32 = S.u.m.in32;
            for (uint32_t c0 = 0; c0 < n; c0 += WG, chunk++) {
                HD_T0(tq);
                const uint32_t p = c0 + (uint32_t)tid;
                const int w_pub = (int)(chunk & 3u), w_fill = (w_pub + 2) & 3;     // the wavefront that publishes this chunk / tops up the ring
                // top up the ring for the NEXT chunk: 32 bytes per lane of one wavefront, requested now, stored behind this chunk's searches (the
                // bytes they replace lie more than WINDOW before the next chunk)
                const bool refill = hi < pad_end && hi < c0 + WG + AHEAD;
                uint4 rf0 = {0, 0, 0, 0}, rf1 = {0, 0, 0, 0};
                if (refill && wave == w_fill) {
                    const uint32_t q = hi + (uint32_t)lane * 32u;
                    if (q + 32u <= n) { __builtin_memcpy(&rf0, src + q, 16); __builtin_memcpy(&rf1, src + q + 16, 16); }
                    else if (q < n) {
                        uint8_t t[32];
#pragma unroll 1
                        for (int k = 0; k < 32; k++) t[k] = q + k < n ? src[q + k] : 0;
                        __builtin_memcpy(&rf0, t, 16); __builtin_memcpy(&rf1, t + 16, 16);
                    }
                }
                // ---- the candidates ------------------------------------------------------------------------------------------------------------
                // A position that cannot start a match (the block's first byte, its last three, the tail of the last chunk) has maxl = 0: every length
                // clamps to 0.
                const bool hashable = p + 4u <= n;
                const uint32_t maxl = !hashable || p == 0u ? 0u : n - p < 258u ? n - p : 258u;
                const uint32_t rp = hashable ? ro(p) : 0u;
                uint32_t own[8];
                load_string<32>(in32, rp, own);
                const uint32_t h = hashable ? hash4(own[0]) : 0u;
                // The best match so far is one key: length << 16 | (32768 - distance): a maximum picks the longer match and, among equals, the nearer
                // one -- so a key of length 32 names the NEAREST candidate that got through all 32 bytes.  All of this is straight-line code: the
                // kernel is bound by instruction issue, every exec-mask branch costs scalar instructions on top of both sides.
                uint32_t key = 0;
                {
                    static_assert(WAYS % 4 == 0, "the ways are read eight bytes at a time");
                    const uint32_t *row32 = (const uint32_t *)&S.u.m.tab[h * MAX_WAYS];         // 24-byte rows, 8-byte aligned
                    const uint32_t dcap = p < WINDOW + 1u ? p : WINDOW + 1u;
#pragma unroll
                    for (int k = 0; k < WAYS / 2; k += 2) {
                        const uint2 rw = *(const uint2 *)(row32 + k);
                        const uint32_t cw[2] = {rw.x, rw.y};
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const uint32_t c = (cw[w >> 1] >> ((w & 1) * 16)) & 0xffffu;
                            // No validity test: an empty slot (0xffff lies "ahead" of p: the distance wraps to a huge number) or an entry that has left
                            // the window is CLAMPED to the farthest position the ring still holds (or the block's first byte) -- a real position, whatever
                            // it matches is a real match.  One v_min instead of a compare and two selects (and the wait states between them).
                            const uint32_t d = p - c;
                            const uint32_t dd = d < dcap ? d : dcap;
                            const uint32_t rc0 = rp - dd, rc1 = rc0 + RING;           // ring offset of the candidate: rp - d, + RING when that wrapped
                            uint32_t l = common_prefix_fixed<32>(in32, rc0 < rc1 ? rc0 : rc1, own);
                            l = l < maxl ? l : maxl;
                            const uint32_t k1 = ((l << 16) | 32768u) - dd;             // (v_lshl_or + v_sub)
                            key = k1 > key ? k1 : key;
                        }
                    }
                }
                {   // distance 1 (runs are never in the table: candidates come from earlier chunks).  A match of length L at distance 1 means the L bytes
                    // from p on all equal the byte before p: compare the position's own bytes with that byte, no second string to fetch
                    const uint32_t splat = (uint32_t)in8[p >= 1u ? ro(p - 1u) : 0u] * 0x01010101u;      // (p = 0: maxl = 0)
                    uint32_t m = 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t x = own[j] ^ splat;
                        uint32_t f;
                        asm("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(x));
                        f |= (uint32_t)(32 * j);
                        m = f < m ? f : m;
                    }
                    m >>= 3;
                    uint32_t l = m < 32u ? m : 32u;
                    l = l < maxl ? l : maxl;
                    const uint32_t k1 = (l << 16) | 32767u;
                    key = k1 > key ? k1 : key;
                }
                HD_TACCM(6, tq);
                // ---- the nearest candidate that got through all 32 bytes goes on alone, 16 bytes per dependent round; the others keep what they have proven
                if ((key >> 16) == 32u && maxl > 32u) {
                    const uint32_t dist = 32768u - (key & 0xffffu);
                    const uint32_t rc0 = rp - dist, rc1 = rc0 + RING, rc = rc0 < rc1 ? rc0 : rc1;      // ring offsets of both strings; + l stays below 2 * RING
                    uint32_t l = 32u;
                    while (l < maxl) {
                        unsigned long long a0, a1, b0, b1;
                        const uint32_t qa = rc + l, qb = rp + l;
                        load16(in32, qa - RING < qa ? qa - RING : qa, a0, a1); load16(in32, qb - RING < qb ? qb - RING : qb, b0, b1);
                        const unsigned long long x0 = a0 ^ b0, x1 = a1 ^ b1;
                        if (x0) { l += (uint32_t)__builtin_ctzll(x0) >> 3; break; }
                        if (x1) { l += 8u + ((uint32_t)__builtin_ctzll(x1) >> 3); break; }
                        l += 16;
                    }
                    l = l < maxl ? l : maxl;
                    key = (l << 16) | (key & 0xffffu);
                }
                uint32_t best = key >> 16;
                const uint32_t bd = 32768u - (key & 0xffffu);
                if (best < 3u || (best == 3u && bd > TOO_FAR) || (LAZY >= 2 && best == 4u && bd > 2048u)) best = 0;
                S.mlen[tid] = (uint16_t)best;
                S.hk[tid] = (uint16_t)h;
                HD_TACCM(7, tq);
                __syncthreads();
                HD_TACCM(8, tq);
                // lazy parse: a match yields to a longer one starting at the next position (and, two-step, to one longer by two or more at the position
                // after that); mlen[WG], mlen[WG + 1] = 0: the chunk's last positions cannot look ahead
                const bool take = (best >= 3u) & !(LAZY >= 1 && (uint32_t)S.mlen[tid + 1] > best) & !(LAZY >= 2 && (uint32_t)S.mlen[tid + 2] > best + 1u);
                // ---- one wavefront publishes the chunk's positions, one stores the ring's top-up --------------------------------------------------------
                if (wave == w_pub) {
                    // The slot a position gets inside its bucket comes from an atomic counter; ONE wavefront inserts the four groups of 64 one after the
                    // other (a wave's own LDS atomics resolve in lane order): the table -- and with it the compressed bytes -- is the same on every run.
                    // (the four groups' reads, atomics and stores are issued back to back -- LDS operations of one wavefront complete in order)
                    uint32_t hh[4], old[4];
#pragma unroll
                    for (int s = 0; s < 4; s++) hh[s] = (uint32_t)S.hk[64u * s + (uint32_t)lane];
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const bool v = c0 + 64u * s + (uint32_t)lane + 4u <= n;
                        old[s] = atomicAdd(&S.cnt32[hh[s] >> 2], v ? 1u << ((hh[s] & 3u) * 8u) : 0u);
                    }
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const uint32_t pq = c0 + 64u * s + (uint32_t)lane;
                        if (pq + 4u <= n) S.u.m.tab[hh[s] * MAX_WAYS + ((old[s] >> ((hh[s] & 3u) * 8u)) & 0xffu) % (uint32_t)WAYS] = (uint16_t)pq;
                    }
                } else if (wave == w_fill && refill) {
                    const uint32_t r = ro(hi + (uint32_t)lane * 32u) >> 2;
                    uint32_t *d = S.u.m.in32 + r;
                    d[0] = rf0.x; d[1] = rf0.y; d[2] = rf0.z; d[3] = rf0.w; d[4] = rf1.x; d[5] = rf1.y; d[6] = rf1.z; d[7] = rf1.w;
                    if (r < MIRROR / 4) {
                        d = S.u.m.in32 + RING / 4 + r;
                        d[0] = rf0.x; d[1] = rf0.y; d[2] = rf0.z; d[3] = rf0.w; d[4] = rf1.x; d[5] = rf1.y; d[6] = rf1.z; d[7] = rf1.w;
                    }
                }
                if (refill) hi += REFILL;
                // ---- the parse of this wavefront's 64 positions for every entry at once: step = the token length a walk standing at a position takes;
                //      R = mask of the token starts on the walk from this lane, J = where that walk leaves the 64 (64 .. 321).  Round t doubles the
                //      tokens covered: R |= R[J], J = J[J] ------------------------------------------------------------------------------------------------
                unsigned long long R = p < n ? 1ull << lane : 0ull;
                uint32_t J = (uint32_t)lane + (take ? best : 1u);
#pragma unroll 1
                for (int t = 0; t < 6; t++) {
                    const bool in = J < 64u;
                    if (__ballot(in) == 0ull) break;
                    const int idx = (int)((J & 63u) << 2);
                    const uint32_t rl = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)(uint32_t)R);
                    const uint32_t rh = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)(uint32_t)(R >> 32));
                    const uint32_t j2 = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)J);
                    if (in) { R |= ((unsigned long long)rh << 32) | rl; J = j2; }
                }
                S.exitp[tid] = (uint16_t)(J | ((uint32_t)__popcll(R) << 9));           // exit position (<= 321) and number of tokens of the walk from here
                HD_TACCM(9, tq);
                __syncthreads();
                HD_TACCM(10, tq);
                // ---- chain the wavefronts' walks: entry of the next = exit of this one - 64 ---------------------------------------------------------------
                // (lane i fetches the four wavefronts' records of entry i in one go; the chain itself is then four v_readlane steps instead of four
                // dependent LDS round trips)
                uint32_t xs[4];
#pragma unroll
                for (int s = 0; s < 4; s++) xs[s] = (uint32_t)S.exitp[64u * s + (uint32_t)lane];
                uint32_t e = carry, cnt = ntok, my_e = 0, my_base = 0;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (s == wave) { my_e = e; my_base = cnt; }
                    if (e < 64u) {
                        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)xs[s], (int)e);
                        cnt += v >> 9; e = (v & 511u) - 64u;
                    } else e -= 64u;
                }
                carry = e; ntok = cnt;
                unsigned long long mine = 0;
                if (my_e < 64u) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)R, (int)my_e);
                    const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(R >> 32), (int)my_e);
                    mine = ((unsigned long long)hi32 << 32) | lo;
                }
                // ---- emit the chosen tokens, in order -------------------------------------------------------------------------------------------
                if ((mine >> lane) & 1ull) {
                    const uint32_t idx = my_base + (uint32_t)__popcll(mine & below);
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
                HD_TACCM(11, tq);
            }
        }
        // ---- choose the block type and build the codes -------------------------------------
        __syncthreads();
        HD_TACC(2, tp);
        uint32_t dyn_bits = 0;
        if (level != 0) {
            // the staged input and the hash table are dead now: the collective Huffman phase (deflate_huff_wg.h), the codes and the bit-packing window
            // take their LDS (the union in Lds)
            Huff &H = S.u.e.h;
            hgdef::HuffWG &W = S.u.e.w;
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
            Huff &H = S.u.e.h;
            for (int i = tid; i < 2 * 520; i += WG) (&H.obuf[0][0])[i] = 0;
            __syncthreads();
            uint32_t bitpos = hoff * 8u, win = 0;
            {   // the dynamic-block header: (value, bit count) items of wg_dynamic_header, one per thread
                const hgdef::HuffWG &W = S.u.e.w;
                const uint32_t nitems = W.nitems;
                for (uint32_t i0 = 0; i0 < nitems; i0 += WG) {
                    const uint32_t i = i0 + (uint32_t)tid;
                    uint32_t nb = 0; uint64_t v = 0;
                    if (i < nitems) { v = W.item_v[i]; nb = W.item_n[i]; }
                    pack_bits(S, o32, bitpos, win, v, nb, tid);
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
                pack_bits(S, o32, bitpos, win, v, nb, tid);
            }
            // a chunk that is not the last of its stream continues with an empty stored block:
            // 000 (BFINAL=0, BTYPE=00), pad to a byte, LEN=0000 NLEN=FFFF
            if (mode == 1 && !last_chunk) { uint32_t nb = tid == 0 ? 3u : 0u; pack_bits(S, o32, bitpos, win, 0, nb, tid); }
            // pad to a byte boundary, then CRC32 + ISIZE as 8 single bytes
            {
                uint32_t nb = 0; uint64_t v = 0;
                if (tid == 0) nb = (8u - (bitpos & 7u)) & 7u;
                pack_bits(S, o32, bitpos, win, v, nb, tid);
            }
            if (mode == 1 && !last_chunk) {
                { uint32_t nb = tid < 4 ? 8u : 0u; uint64_t v = tid >= 2 ? 0xffu : 0u; pack_bits(S, o32, bitpos, win, v, nb, tid); }
            }
            total_len = (bitpos >> 3) + (mode == 1 ? 0u : 8u);
            // flush the partial dword that is still in the staging window
            __syncthreads();
            if (tid == 0 && (bitpos & 31u)) o32[bitpos >> 5] = H.obuf[win][0];
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
        if (tid == 0) { atomicAdd(&g_dprof[5], 1ull); for (int k = 0; k < 16; k++) if (k != 5) atomicAdd(&g_dprof[k], dacc[k]); }
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

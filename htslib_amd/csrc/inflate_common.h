// inflate_common.h -- decode tables, the table builder and the wave-window bit reader of the BGZF inflate kernel (bgzf_inflate.hip:
// one block per wavefront; also used by the shelved two-kernel pipeline under experiments/).
// Everything lives in namespace hg; the includer may define HG_TRACE / HG_T0 / HG_TACC / HG_CNT before including.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_device.h"
#ifndef HG_TRACE
#define HG_TRACE(slot, val) do { } while (0)
#endif

namespace hg {

#ifndef HG_LIT_RB
#define HG_LIT_RB 9
#endif
constexpr int LIT_RB = HG_LIT_RB;
constexpr int DIST_RB = 8;
// zlib's ENOUGH bound for 286 symbols / 15-bit codes: root 10 -> 1332 entries, root 9 -> 852
constexpr int LIT_TAB = HG_LIT_RB == 10 ? 1344 : 864;
constexpr int DIST_TAB = 416;   // >= ENOUGH(30 symbols, root 8, max 15)  = 402
constexpr int PRE_RB = 7;
#ifndef HG_WAVES_PER_WG
#define HG_WAVES_PER_WG 1      // one wavefront per workgroup: LDS is handed out per workgroup, and 26 x 6 KiB fit a CU where 6 x 24 KiB leave a hole
#endif
constexpr int WAVES_PER_WG = HG_WAVES_PER_WG;
#ifndef HG_RING
#define HG_RING 1024
#endif
constexpr uint32_t RING = HG_RING;          // bytes of most recent output mirrored in LDS per wave
constexpr uint32_t RING_NEAR = RING - 64u;  // look-back spans up to this are served from LDS

// table entry: [3:0] code bits to drop, [4] literal, [5] length/distance,
// [6] end of block, [7] second-level pointer, [11:8] extra bits (or sub-table
// bits), [31:16] value (literal / base / sub-table offset)
constexpr uint32_t F_LIT = 0x10u, F_BASE = 0x20u, F_EOB = 0x40u, F_SUB = 0x80u;
// bit 31: "fast literal" -- a literal whose whole code sits in the root table, so the hot loop
// needs ONE sign test to know it can emit a byte and drop e&15 bits.
constexpr uint32_t F_FAST = 0x80000000u;

struct BuildScratch {               // only live while the Huffman tables of a deflate block are built
    uint32_t cnt[16];
    uint32_t nc[16];
    uint32_t alloc;
    uint32_t pad[3];
    uint8_t lens[352];
};
// One wavefront's LDS (= one workgroup's: WAVES_PER_WG is 1, so the struct sits at LDS address 0 and the ring on a 1 KiB boundary, which the
// symbol loop's ring addressing (pos & 1023) | base needs).  4 608 bytes: under the 5 KiB step that 32 wavefronts per CU need.
struct WaveLds {
    union {                         // the output ring shares its LDS with the table-build scratch;
        BuildScratch b;             // it is re-filled from the wave's own output after each build
        uint8_t ring[RING > sizeof(BuildScratch) ? RING : sizeof(BuildScratch)];
    } u;
    uint32_t lit[1 << LIT_RB];      // litlen root table (32-bit entries); also the workspace in which the precode table and the distance table are built
    uint16_t litsub[LIT_TAB - (1 << LIT_RB)];   // litlen second level, 16-bit entries: [3:0] code bits, 0x10 literal in [15:8], 0x20 length symbol
                                    // (0..28) in [12:8], 0x40 end of block; entry k of the table = litsub[k - 512]
    uint16_t dist[DIST_TAB];        // distance table, 16-bit entries: [3:0] code bits, 0x20 distance symbol in [12:8] (base and extra bits come
                                    // from the symbol: a 30-entry table in one VGPR, read with v_readlane), 0x80 second level: [3:0] its index
                                    // bits, [15:8] its first entry / 2
};
static_assert(WAVES_PER_WG == 1, "the ring must sit on a 1 KiB LDS boundary: one WaveLds per workgroup");

enum { KIND_LITLEN = 0, KIND_DIST = 1, KIND_PRE = 2 };

__device__ __forceinline__ uint32_t make_entry(int kind, uint32_t sym, uint32_t nb, bool root) {
    if (kind == KIND_PRE) return (sym << 16) | F_LIT | nb;
    if (kind == KIND_LITLEN && !root) {                        // second level: 16-bit form (WaveLds::litsub); len_base_extra() has the rest
        if (sym < 256) return (sym << 8) | F_LIT | nb;
        if (sym == 256) return F_EOB | nb;
        return sym - 257 > 28 ? 0u : ((sym - 257) << 8) | F_BASE | nb;
    }
    if (kind == KIND_LITLEN) {
        if (sym < 256) return (sym << 16) | F_LIT | nb | (root ? F_FAST : 0u);
        if (sym == 256) return F_EOB | nb;
        uint32_t s = sym - 257;
        if (s > 28) return 0;                                  // 286, 287: invalid
        uint32_t extra, base;
        if (s < 8) { extra = 0; base = 3 + s; }
        else if (s == 28) { extra = 0; base = 258; }
        else { extra = (s - 4) >> 2; base = 3 + ((4 + (s & 3)) << extra); }
        return (base << 16) | (extra << 8) | F_BASE | nb;
    }
    if (sym > 29) return 0;                                    // 30, 31: invalid
    return (sym << 8) | F_BASE | nb;                           // 16-bit form (WaveLds::dist); dist_base_extra() has the rest
}

// Lane k of ONE VGPR: what the 16-bit entries leave out (RFC 1951 3.2.5) -- distance symbol k: base in [14:0], extra bit count in [18:15];
// length symbol 257 + k: base in [27:19], extra bit count in [30:28].  Read with v_readlane.
__device__ __forceinline__ uint32_t sym_base_extra(uint32_t k) {
    uint32_t v = 0;
    if (k <= 29) {
        const uint32_t extra = k < 4 ? 0 : (k - 2) >> 1;
        const uint32_t base = k < 4 ? 1 + k : 1 + ((2 + (k & 1)) << extra);
        v = base | (extra << 15);
    }
    if (k <= 28) {
        const uint32_t extra = k < 8 || k == 28 ? 0 : (k - 4) >> 2;
        const uint32_t base = k < 8 ? 3 + k : k == 28 ? 258 : 3 + ((4 + (k & 3)) << extra);
        v |= (base << 19) | (extra << 28);
    }
    return v;
}

// Build a root+subtable decode table from code lengths S.u.b.lens[lens_off ..+n).
// Returns 0 ok, 1 invalid code set (over-subscribed / illegal incomplete).
// All 64 lanes participate; result uniform.
template <int KIND, int RB, int CAP, int NCHUNK>
// The table builder and the CRC pass are real calls: inlined into the kernel they set its register budget (105 VGPRs wanted, 80 granted with
// spills to scratch); as functions the kernel needs 63 and spills nothing, and the calls cost nothing measurable (one per deflate block).
#ifndef HG_PHASE_FN
#define HG_PHASE_FN __attribute__((noinline))
#endif
__device__ HG_PHASE_FN int build_table(WaveLds &S, uint32_t *tab, int lens_off, int n, int lane) {
    // ---- zero root + count code lengths --------------------------------
    HG_TRACE(4, 100 + KIND);
    if (lane < 16) { S.u.b.cnt[lane] = 0; }
    for (int i = lane; i < (1 << RB); i += 64) tab[i] = 0;
    if (lane == 0) S.u.b.alloc = 1u << RB;
    wave_sync();
    uint32_t L[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
        int sym = c * 64 + lane;
        L[c] = sym < n ? S.u.b.lens[lens_off + sym] : 0;
        if (L[c]) atomicAdd(&S.u.b.cnt[L[c]], 1u);
    }
    wave_sync();
    // ---- canonical first codes (RFC 1951 3.2.2), uniform -----------------
    uint32_t code = 0;
    int left = 1, total = 0, maxlen = 0;
    uint32_t mycnt = lane < 16 ? S.u.b.cnt[lane] : 0;
    uint32_t first_code = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
        uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, l);
        if (lane == l) first_code = code;
        code = (code + c) << 1;
        left = (left << 1) - (int)c;
        if (left < 0) return 1;                                  // over-subscribed
        total += (int)c;
        if (c) maxlen = l;
    }
    if (left > 0) {
        // incomplete: legal only for "no codes" or "a single 1-bit code"
        // (same rule as the oracle / zlib inftrees: max != 1 -> error)
        if (!(total == 0 || (total == 1 && maxlen == 1))) return 1;
    }
    if (lane >= 1 && lane < 16) S.u.b.nc[lane] = first_code;
    wave_sync();
    HG_TRACE(4, 200 + KIND);
    // ---- assign codes in symbol order, fill root entries -----------------
    bool any_long = maxlen > RB;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
        uint32_t len = L[c];
        uint32_t mycode = 0;
        unsigned long long todo = __ballot(len != 0);
        int guard = 0;
        while (todo) {                                           // one pass per distinct length
            if (++guard > 16) return 1;                          // cannot happen (<= 15 lengths)
            HG_TRACE(8, guard); HG_TRACE(9, (uint32_t)todo); HG_TRACE(10, (uint32_t)(todo >> 32));
            int leader = __builtin_ctzll(todo);
            uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, leader);
            unsigned long long m = __ballot(len == l);
            uint32_t base = S.u.b.nc[l];                             // uniform LDS read
            uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (len == l) mycode = base + rank;
            wave_sync();
            if (lane == leader) S.u.b.nc[l] = base + (uint32_t)__popcll(m);
            wave_sync();
            todo &= ~m;
        }
        L[c] = len | (mycode << 8);
        if (len) {
            uint32_t sym = (uint32_t)(c * 64 + lane);
            uint32_t rev = __brev(mycode) >> (32 - len);
            if (len <= (uint32_t)RB) {
                uint32_t e = make_entry(KIND, sym, len, true);
                for (uint32_t idx = rev; idx < (1u << RB); idx += 1u << len) tab[idx] = e;
            } else {
                atomicMax(&tab[rev & ((1u << RB) - 1)], len);     // longest code under this root slot
            }
        }
    }
    HG_TRACE(4, 300 + KIND);
    if (!any_long) { wave_sync(); return 0; }
    wave_sync();
    // ---- size and place second-level tables ------------------------------
    int bad = 0;
    for (int i = lane; i < (1 << RB); i += 64) {
        uint32_t v = tab[i];
        if (v != 0 && v < 16) {
            uint32_t sb = v - RB;
            uint32_t off = atomicAdd(&S.u.b.alloc, 1u << sb);
            if (off + (1u << sb) > (uint32_t)CAP) { bad = 1; tab[i] = 0; }
            else tab[i] = KIND == KIND_DIST ? ((off >> 1) << 8) | F_SUB | sb : (off << 16) | (sb << 8) | F_SUB | RB;
        }
    }
    if (__ballot(bad)) return 1;
    wave_sync();
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
        uint32_t len = L[c] & 0xff, mycode = L[c] >> 8;
        if (len > (uint32_t)RB) {
            uint32_t sym = (uint32_t)(c * 64 + lane);
            uint32_t rev = __brev(mycode) >> (32 - len);
            uint32_t root = tab[rev & ((1u << RB) - 1)];
            const uint32_t off = KIND == KIND_DIST ? ((root >> 8) & 0xffu) << 1 : root >> 16, sb = KIND == KIND_DIST ? root & 0xfu : (root >> 8) & 0xfu;
            uint32_t e = make_entry(KIND, sym, len - RB, false);
            if (KIND == KIND_LITLEN) for (uint32_t idx = rev >> RB; idx < (1u << sb); idx += 1u << (len - RB)) S.litsub[off - (1u << RB) + idx] = (uint16_t)e;
            else for (uint32_t idx = rev >> RB; idx < (1u << sb); idx += 1u << (len - RB)) tab[off + idx] = e;
        }
    }
    wave_sync();
    return 0;
}

// ------------------------------------------------------------------ bit reader
struct BitReader {
    const uint32_t *g;     // dword view of the stream, aligned down from the block start
    uint32_t max_dw;       // last readable dword index (clamp)
    uint32_t wbase;        // dword index held by lane 0 of `win`
    uint32_t win, win_next;
    uint32_t next_dw;      // next dword to append to the bit buffer
    uint64_t bb;
    uint32_t bc;
};

__device__ __forceinline__ uint32_t br_gload(const BitReader &br, uint32_t idx) {
    idx = idx < br.max_dw ? idx : br.max_dw;
    return br.g[idx];
}

__device__ __forceinline__ uint32_t br_fetch(BitReader &br, uint32_t idx, int lane) {
    uint32_t rel = idx - br.wbase;
    if (rel >= 64u) {
        if (rel < 128u) { br.win = br.win_next; br.wbase += 64u; }
        else { br.wbase = idx; br.win = br_gload(br, idx + lane); }
        br.win_next = br_gload(br, br.wbase + 64u + lane);
        rel = idx - br.wbase;
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)br.win, (int)rel);
}

// position the reader at byte offset `byte_pos` (relative to br.g)
__device__ __forceinline__ void br_seek(BitReader &br, uint32_t byte_pos, int lane) {
    uint32_t dw = byte_pos >> 2, sh = (byte_pos & 3u) * 8u;
    uint32_t w = br_fetch(br, dw, lane);
    br.bb = (uint64_t)(w >> sh);
    br.bc = 32u - sh;
    br.next_dw = dw + 1;
}

// position the reader at BIT offset `bit_pos` (relative to br.g)
__device__ __forceinline__ void br_seek_bits(BitReader &br, uint32_t bit_pos, int lane) {
    uint32_t dw = bit_pos >> 5, sh = bit_pos & 31u;
    uint32_t w = br_fetch(br, dw, lane);
    br.bb = (uint64_t)(w >> sh);
    br.bc = 32u - sh;
    br.next_dw = dw + 1;
}

__device__ __forceinline__ void br_refill(BitReader &br, int lane) {
    if (br.bc <= 32u) {
        uint32_t w = br_fetch(br, br.next_dw, lane);
        br.next_dw++;
        br.bb |= (uint64_t)w << br.bc;
        br.bc += 32u;
    }
}
__device__ __forceinline__ uint32_t br_peek(const BitReader &br, uint32_t n) {
    return (uint32_t)br.bb & ((1u << n) - 1u);
}
__device__ __forceinline__ void br_drop(BitReader &br, uint32_t n) { br.bb >>= n; br.bc -= n; }
__device__ __forceinline__ uint32_t br_bits(BitReader &br, uint32_t n) {
    uint32_t v = br_peek(br, n);
    br_drop(br, n);
    return v;
}
// bytes consumed so far, relative to br.g
__device__ __forceinline__ uint32_t br_byte_pos(const BitReader &br) {
    return br.next_dw * 4u - (br.bc >> 3);
}

__device__ __forceinline__ uint32_t lds_uniform(const uint32_t *p) {
    return uni(*p);
}


}  // namespace hg

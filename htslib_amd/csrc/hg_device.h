// hg_device.h -- shared device-side helpers for the gfx950 block-codec kernels.
//
// Everything here is written for CDNA4 wave64: one wavefront owns one block,
// wave-uniform state lives in SGPRs (values are made provably uniform with
// readfirstlane/readlane so hipcc keeps the bit-twiddling on the scalar ALU),
// and the 64 lanes are used for table construction, byte copies and CRC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HG_WAVE 64

// v_writelane_b32 has no clang builtin in ROCm 7.2; bind the LLVM intrinsic directly
// (the compiler then manages M0 for the lane select itself).
extern "C" __device__ int hg_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

namespace hg {

__device__ __forceinline__ uint32_t writelane(uint32_t val, uint32_t lane, uint32_t old) {
    return (uint32_t)hg_llvm_writelane((int)val, (int)lane, (int)old);
}
__device__ __forceinline__ uint32_t uni(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// A zero the compiler cannot see through, living in a VGPR: OR-ing it into a wave-uniform
// value moves the arithmetic that follows from the scalar ALU to the vector ALUs.
__device__ __forceinline__ uint32_t vzero() {
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
// true in every lane iff the predicate holds in some lane (predicates here are wave-uniform)
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// Orders LDS / global traffic between lanes of ONE wave.  Within a wave the
// hardware keeps DS and VMEM operations in program order, so this only has to
// stop the compiler from moving accesses across it (wavefront-scope fences are
// no-ops in the gfx9 memory model).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15), as an instruction the compiler's own wait insertion sees: placed right after a RARE load
// whose register is read in a hot loop, it keeps the per-iteration uses free of conservative waits.
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Inclusive prefix sum over the 64 lanes on the DPP data path (no LDS round trips): shifts by
// 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1, out-of-row lanes read 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return (uint32_t)v;
}

// ---------------------------------------------------------------------------
// CRC-32 (RFC 1952 section 8; zlib's crc32(), called at bgzf.c:612,747,793).
// Slice-by-4 tables and the x^(8*2^j) mod P table used to concatenate per-lane
// partial CRCs are built at compile time.
// ---------------------------------------------------------------------------
struct CrcTables {
    uint32_t t[4][256];
    uint32_t xpow[40];   // xpow[j] = x^(8 * 2^j) mod P, reflected representation
};

constexpr uint32_t kCrcPoly = 0xEDB88320u;

constexpr uint32_t crc_mulmod_host(uint32_t a, uint32_t b) {
    // product of two polynomials mod P; bit 31 is the x^0 coefficient
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

constexpr CrcTables make_crc_tables() {
    CrcTables T{};
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? kCrcPoly ^ (c >> 1) : c >> 1;
        T.t[0][n] = c;
    }
    for (uint32_t n = 0; n < 256; n++)
        for (int k = 1; k < 4; k++)
            T.t[k][n] = (T.t[k - 1][n] >> 8) ^ T.t[0][T.t[k - 1][n] & 0xff];
    uint32_t p = 0x00800000u;           // x^8
    for (int j = 0; j < 40; j++) { T.xpow[j] = p; p = crc_mulmod_host(p, p); }
    return T;
}

static __device__ __constant__ const CrcTables g_crc = make_crc_tables();

__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        p ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = (b >> 1) ^ (kCrcPoly & (0u - (b & 1u)));
    }
    return p;
}

__device__ __forceinline__ uint32_t crc_byte(uint32_t c, uint32_t b) {
    return g_crc.t[0][(c ^ b) & 0xff] ^ (c >> 8);
}
__device__ __forceinline__ uint32_t crc_word(uint32_t c, uint32_t w) {
    c ^= w;
    return g_crc.t[3][c & 0xff] ^ g_crc.t[2][(c >> 8) & 0xff] ^
           g_crc.t[1][(c >> 16) & 0xff] ^ g_crc.t[0][c >> 24];
}

// CRC-32 of p[0..n) computed by one wave.  The buffer is cut into 64 chunks of
// K = 2^k bytes aligned to the END of the buffer; lane i owns chunk i, runs a
// slice-by-4 CRC over it (init 0, or ~0 for the chunk holding byte 0) and the
// partial states are concatenated with a 6-step butterfly:
//     crc(A||B) = crc(A) * x^(8|B|) mod P  xor  crc(B)
// Returns the finalised CRC in every lane.
#ifndef HG_PHASE_FN
#define HG_PHASE_FN __attribute__((noinline))      /* see inflate_common.h: build_table */
#endif
__device__ HG_PHASE_FN uint32_t wave_crc32(const uint8_t *p, uint32_t n, int lane) {
    if (n == 0) return 0;
    uint32_t per = (n + 63u) >> 6;
    int k = 2;
    while ((1u << k) < per) k++;
    const uint32_t K = 1u << k;
    long long beg = (long long)n - (long long)(64 - lane) * (long long)K;
    long long end = beg + (long long)K;
    if (beg < 0) beg = 0;
    if (end < 0) end = 0;
    uint32_t len = (uint32_t)(end - beg);
    uint32_t c = (beg == 0 && end > 0) ? 0xffffffffu : 0u;
    const uint8_t *q = p + beg;
    uint32_t head = len & 3u;               // only the chunk holding byte 0
    for (uint32_t i = 0; i < head; i++) c = crc_byte(c, q[i]);
    q += head; len -= head;
    uint32_t i = 0;
    for (; i + 16 <= len; i += 16) {
        uint4 w;
        __builtin_memcpy(&w, q + i, 16);
        c = crc_word(c, w.x); c = crc_word(c, w.y);
        c = crc_word(c, w.z); c = crc_word(c, w.w);
    }
    for (; i < len; i += 4) {
        uint32_t w;
        __builtin_memcpy(&w, q + i, 4);
        c = crc_word(c, w);
    }
#pragma unroll
    for (int s = 0; s < 6; s++) {
        uint32_t other = (uint32_t)__shfl_xor((int)c, 1 << s, 64);
        uint32_t m = g_crc.xpow[k + s];
        bool left = ((lane >> s) & 1) == 0;
        uint32_t a = left ? c : other;      // the half that comes first in memory
        uint32_t b = left ? other : c;
        c = crc_mulmod(a, m) ^ b;
    }
    return c ^ 0xffffffffu;
}

}  // namespace hg

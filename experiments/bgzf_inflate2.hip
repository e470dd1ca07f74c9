// bgzf_inflate2.hip -- BGZF inflate as a two-kernel pipeline ("v2"): a lane-parallel PARSE kernel and a
// one-wavefront-per-block RESOLVE kernel that exchange LZ77 tokens through HBM.  OPT-IN (HG_INFLATE_V2=1): it is
// bit-exact on every fixture and test (tests/test_bgzf_inflate_gpu.py runs them through it), but it measured 65 GB/s against
// 106 GB/s for the one-kernel path on the same file -- profiles/r02_inflate_v2_steps.txt has the step-by-step numbers.
//
// Idea.  The one-block-per-wavefront kernel (bgzf_inflate.hip) spends ~40 vector instructions per Huffman symbol with all
// 64 lanes computing the same value.  Decoding can use the lanes: the bits of a deflate block are cut into segments, every
// lane parses the tokens that START in its segment from a guessed entry bit, and entries are corrected until they chain
// (lane 0 is always right, so the fixpoint is the true parse).  What cannot be parallelised inside a block is the LZ77 copy
// chain (a sorted BAM copies each record from the previous one: scripts/inflate_sim.c) -- so that part stays sequential,
// one wavefront per block, but it no longer decodes: it reads ready-made 32-bit tokens.
//
//   parse_kernel    one wavefront per block (256 lanes per block were tried: 20 correction passes instead of 3).
//                   Deflate block header + decode tables (shared code, inflate_common.h), speculative parse passes over the
//                   LDS tables, prefix sum of token counts, token emission.
//                   token = literal byte | 0x80000000 | (len-3) << 16 | (dist-1).
//   resolve_kernel  one wavefront per block: 64 tokens per step, prefix sum of lengths, literals and matches go to a 4 KiB
//                   LDS ring that is flushed to HBM 1 KiB at a time with 16-byte stores; CRC-32 as in v1.
// Tokens: 4 B per symbol, ~0.5 B per plain byte on the bench BAM, written once and read once.
// What it taught: 2.0x fewer vector instructions per block than v1, but the resolve loop is bound by the CU's single
// scalar ALU and the parse loop by per-lane dependent-instruction latency (see the profile file).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include "inflate_common.h"

namespace hg2 {
using namespace hg;

// In-kernel phase timing (s_memtime ticks), compiled in only with -DHG_PROFILE (tests/native/kbench prints it).
// parse: 0 block total, 1 header + tables, 2 speculative passes, 3 prefix + emission, 4 blocks, 5 passes, 6 rounds;
// resolve: 8 block total, 9 token loop, 10 flush, 11 crc, 12 blocks
#ifdef HG_PROFILE
__device__ unsigned long long g_prof2[16];
#define P_T0(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define P_ACC(slot, var) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_prof2[slot], n_ - var); var = n_; } while (0)
#define P_ACCW(slot, var) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) atomicAdd(&g_prof2[slot], n_ - var); var = n_; } while (0)
#define P_CNT(slot, v) do { if (threadIdx.x == 0) atomicAdd(&g_prof2[slot], (unsigned long long)(v)); } while (0)
#define P_CNTW(slot, v) do { if (lane_id() == 0) atomicAdd(&g_prof2[slot], (unsigned long long)(v)); } while (0)
#else
#define P_T0(var) do { } while (0)
#define P_ACC(slot, var) do { } while (0)
#define P_ACCW(slot, var) do { } while (0)
#define P_CNT(slot, v) do { } while (0)
#define P_CNTW(slot, v) do { } while (0)
#endif

#ifndef HG2_NL
#define HG2_NL 64
#endif
constexpr int NL = HG2_NL;                    // lanes (threads) per block in the parse kernel: 64 (measured) or 256
constexpr int NW = NL / 64;                   // wavefronts per block
constexpr uint32_t SEG = 131072u / NL;        // bits per lane segment: one round covers 128 Kibit, the usual block in one round
constexpr uint32_t TOK_CAP = 65536u + 256u;   // tokens per block: at most one per output byte
constexpr uint32_t IDLE = 0xffffffffu;
enum { P_OK = 0, P_EOB = 1, P_ERR = 2, P_IDLE = 3 };
enum { ST_OK = 0, ST_HEADER = 1, ST_INFLATE = 2, ST_SIZE = 3, ST_CRC = 4 };

constexpr uint32_t ROUND_DW = (uint32_t)NL * SEG / 32u;           // dwords per round (4096)
constexpr uint32_t STAGE_DW = ROUND_DW + 64u;                      // + what the last lane may read past its segment
constexpr uint32_t STAGE_WORDS __attribute__((unused)) = STAGE_DW + (STAGE_DW >> 6) + 2u;

struct ParseLds {
    WaveLds T;                                // tables + build scratch (the ring is not used here)
    uint32_t entry[NL];
    uint32_t exitb[NL];
    uint32_t ntok[NL];
    uint32_t nbyte[NL];
    uint32_t flag[NL];
#if HG2_STAGE
    uint32_t inb[STAGE_WORDS];                // the round's dwords (see LaneBits)
#endif
    uint32_t wsum[8];
    // wave 0 -> everybody: [0] first symbol bit of the deflate block / round  [1] status  [2] bfinal  [3] btype
    // [4] stored length  [5] stored source byte  [6] tokens so far  [7] bytes so far  [8] ticket
    uint32_t bc[12];
};

// ---- per-lane bit reader over the round's bits, staged in LDS ----------------------------------------------------------
// A round covers NL segments of SEG bits = 16 KiB of the stream.  Lanes walk their own segments, i.e. 64 different cache
// lines per load instruction if they read global memory (measured: the L2 -> L1 traffic of that pattern, ~9 MB per 12 KiB
// block, bounded the kernel).  So the round's dwords are copied to LDS once, coalesced, and read from there.  Dword d of
// the round sits at index d + (d >> 6): one padding word per 64 keeps lanes that are at the same offset of their segments
// on different banks.
// Measured (profiles/r02_inflate_v2_steps.txt): staging cuts a block's parse latency by 40 % but its 17 KiB of LDS leave 6
// wavefronts per CU instead of 20, and the kernel as a whole gets slower (4.7 ms vs 3.9 ms per 8192 blocks): HG2_STAGE=0,
// the default, lets the lanes read global memory (clamped to the stream's last dword).
#ifndef HG2_STAGE
#define HG2_STAGE 0
#endif
struct LaneBits { uint64_t bb; uint32_t bc, dw, max_rel; };
__device__ __forceinline__ uint32_t lb_load_(const uint32_t *inb, uint32_t dw, uint32_t max_rel) {
#if HG2_STAGE
    (void)max_rel;
    const uint32_t d = dw < STAGE_DW - 1u ? dw : STAGE_DW - 1u;    // (a lane never needs more; the clamp keeps garbage lanes in bounds)
    return inb[d + (d >> 6)];
#else
    return inb[dw < max_rel ? dw : max_rel];                      // inb = global base of the round, clamped to the image's last dword
#endif
}
#define lb_load(inb, dw) lb_load_(inb, dw, r.max_rel)
__device__ __forceinline__ void lb_seek(LaneBits &r, const uint32_t *inb, uint32_t bit, uint32_t max_rel) {   // bit: relative to the round's base
    r.max_rel = max_rel;
    r.dw = bit >> 5;
    const uint32_t sh = bit & 31u;
    r.bb = (uint64_t)(lb_load(inb, r.dw) >> sh);
    r.bc = 32u - sh;
    r.dw++;
}
__device__ __forceinline__ void lb_refill(LaneBits &r, const uint32_t *inb) {
    if (r.bc <= 32u) { r.bb |= (uint64_t)lb_load(inb, r.dw) << r.bc; r.bc += 32u; r.dw++; }
}
__device__ __forceinline__ uint32_t lb_pos(const LaneBits &r) { return r.dw * 32u - r.bc; }

// Parse the tokens that start in [entry, lim).  EMIT: also write them to tok_out.
// (A variant that stopped correction parses at positions an earlier parse of the lane had visited was tried: the lanes on
// the correction chain are by definition the ones that do NOT re-synchronise, so it saved nothing.)
template <bool EMIT>
__device__ __forceinline__ void lane_parse(const uint32_t *inb, uint32_t base_bit, uint32_t max_rel, const uint32_t *lit, const uint32_t *dist, uint32_t entry,
                                           uint32_t lim, uint32_t end_bit, uint32_t &x, uint32_t &flag, uint32_t &nt, uint32_t &nb,
                                           uint32_t *tok_out) {
    // positions are bit offsets relative to br.g as everywhere else; the reader works relative to the staged base
    LaneBits r;
    lb_seek(r, inb, entry - base_bit, max_rel);
    nt = 0; nb = 0; flag = P_OK;
    for (uint32_t guard = 0; guard < SEG + 128u; guard++) {
        const uint32_t start = lb_pos(r) + base_bit;
        if (start >= lim) break;
        if (start >= end_bit) { flag = P_ERR; break; }
        lb_refill(r, inb);
        uint32_t e = lit[(uint32_t)r.bb & ((1u << LIT_RB) - 1u)];
        if (e & F_SUB) {
            r.bb >>= LIT_RB; r.bc -= LIT_RB;
            e = lit[(e >> 16) + ((uint32_t)r.bb & ((1u << ((e >> 8) & 15u)) - 1u))];
        }
        const uint32_t nbits = e & 15u;
        if (e & F_LIT) {
            r.bb >>= nbits; r.bc -= nbits;
            if (EMIT) tok_out[nt] = (e >> 16) & 0xffu;
            nt++; nb++;
            continue;
        }
        if (!(e & F_BASE)) {
            r.bb >>= nbits; r.bc -= nbits;
            flag = (e & F_EOB) ? P_EOB : P_ERR;
            break;
        }
        const uint32_t xb = (e >> 8) & 15u;
        const uint32_t len = (e >> 16) + (((uint32_t)r.bb >> nbits) & ((1u << xb) - 1u));
        r.bb >>= (nbits + xb); r.bc -= nbits + xb;
        lb_refill(r, inb);
        uint32_t d = dist[(uint32_t)r.bb & ((1u << DIST_RB) - 1u)];
        if (d & F_SUB) {
            r.bb >>= DIST_RB; r.bc -= DIST_RB;
            d = dist[(d >> 16) + ((uint32_t)r.bb & ((1u << ((d >> 8) & 15u)) - 1u))];
        }
        if (!(d & F_BASE)) { flag = P_ERR; break; }
        const uint32_t nb2 = d & 15u, xb2 = (d >> 8) & 15u;
        const uint32_t dv = (d >> 16) + (((uint32_t)r.bb >> nb2) & ((1u << xb2) - 1u));
        r.bb >>= (nb2 + xb2); r.bc -= nb2 + xb2;
        if (len < 3u || len > 258u || dv < 1u || dv > 32768u) { flag = P_ERR; break; }
        if (EMIT) tok_out[nt] = 0x80000000u | ((len - 3u) << 16) | (dv - 1u);
        nt++; nb += len;
    }
    x = lb_pos(r) + base_bit;
}

// Deflate block header + tables, wave 0 only (the code of inflate_stream in bgzf_inflate.hip up to its symbol loop).
// Returns ST_*; for a stored block slen / ssrc describe the raw bytes.
__device__ int block_prologue(WaveLds &S, BitReader &br, uint32_t in_end, int lane, uint32_t &bfinal, uint32_t &btype, uint32_t &slen,
                              uint32_t &ssrc) {
    br_refill(br, lane);
    if (br_byte_pos(br) > in_end) return ST_INFLATE;
    bfinal = br_bits(br, 1);
    btype = br_bits(br, 2);
    slen = ssrc = 0;
    if (btype == 0) {                                              // stored (RFC 1951 3.2.4)
        br_drop(br, br.bc & 7u);
        br_refill(br, lane);
        const uint32_t len = br_bits(br, 16);
        br_refill(br, lane);
        const uint32_t nlen = br_bits(br, 16);
        if ((len ^ 0xffffu) != nlen) return ST_INFLATE;
        ssrc = br_byte_pos(br); slen = len;
        if (ssrc + len > in_end) return ST_INFLATE;
        return ST_OK;
    }
    if (btype == 3) return ST_INFLATE;
    if (btype == 1) {                                              // fixed codes (RFC 1951 3.2.6)
        for (int i = lane; i < 288; i += 64) S.u.b.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        if (lane < 32) S.u.b.lens[288 + lane] = 5;
        wave_sync();
        if (build_table<KIND_LITLEN, LIT_RB, LIT_TAB, 5>(S, S.lit, 0, 288, lane)) return ST_INFLATE;
        if (build_table<KIND_DIST, DIST_RB, DIST_TAB, 1>(S, S.dist, 288, 32, lane)) return ST_INFLATE;
        return ST_OK;
    }
    // dynamic codes (RFC 1951 3.2.7)
    br_refill(br, lane);
    const uint32_t nlen = br_bits(br, 5) + 257, ndist = br_bits(br, 5) + 1, ncode = br_bits(br, 4) + 4;
    if (nlen > 286 || ndist > 30) return ST_INFLATE;
    if (lane < 19) S.u.b.lens[lane] = 0;
    wave_sync();
    for (uint32_t i = 0; i < ncode; i++) {
        br_refill(br, lane);
        const uint32_t v = br_bits(br, 3);
        const uint64_t ord_lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) |
                                (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
        const uint64_t ord_hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
        const uint32_t sym = i < 12 ? (uint32_t)(ord_lo >> (5 * i)) & 31u : (uint32_t)(ord_hi >> (5 * (i - 12))) & 31u;
        if (lane == 0) S.u.b.lens[sym] = (uint8_t)v;
    }
    wave_sync();
    if (build_table<KIND_PRE, PRE_RB, DIST_TAB, 1>(S, S.dist, 0, 19, lane)) return ST_INFLATE;
    {
        const uint32_t c = lane < 16 ? S.u.b.cnt[lane] : 0;        // the code-length code must be complete
        int left = 1;
#pragma unroll
        for (int l = 1; l <= 7; l++) left = (left << 1) - (int)__builtin_amdgcn_readlane((int)c, l);
        if (left != 0) return ST_INFLATE;
    }
    uint32_t idx = 0, total = nlen + ndist, prev = 0;
    while (idx < total) {
        br_refill(br, lane);
        const uint32_t e = lds_uniform(&S.dist[br_peek(br, PRE_RB)]);
        if (!(e & F_LIT)) return ST_INFLATE;
        br_drop(br, e & 15u);
        const uint32_t sym = e >> 16;
        if (sym < 16) {
            if (lane == 0) S.u.b.lens[32 + idx] = (uint8_t)sym;
            prev = sym; idx++;
        } else {
            uint32_t rep, val = 0;
            if (sym == 16) {
                if (idx == 0) return ST_INFLATE;
                val = prev; rep = 3 + br_bits(br, 2);
            } else if (sym == 17) rep = 3 + br_bits(br, 3);
            else rep = 11 + br_bits(br, 7);
            if (idx + rep > total) return ST_INFLATE;
            for (uint32_t j = lane; j < rep; j += 64) S.u.b.lens[32 + idx + j] = (uint8_t)val;
            idx += rep; prev = val;
        }
    }
    wave_sync();
    if (S.u.b.lens[32 + 256] == 0) return ST_INFLATE;             // no end-of-block code
    if (build_table<KIND_LITLEN, LIT_RB, LIT_TAB, 5>(S, S.lit, 32, (int)nlen, lane)) return ST_INFLATE;
    if (build_table<KIND_DIST, DIST_RB, DIST_TAB, 1>(S, S.dist, 32 + (int)nlen, (int)ndist, lane)) return ST_INFLATE;
    return ST_OK;
}

__global__ __launch_bounds__(NL)
void parse_kernel(const uint8_t *__restrict__ comp, uint64_t comp_len, const hg_bgzf_desc *__restrict__ desc, uint32_t block0, uint32_t nblocks,
                  uint32_t *__restrict__ tokbuf, uint32_t *__restrict__ ntok_out, int32_t *__restrict__ pst_out, unsigned int *ticket) {
    __shared__ ParseLds L;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t max_dw_abs = (comp_len + 3) / 4 - 1;
    for (;;) {
        __syncthreads();
        if (tid == 0) L.bc[8] = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t bi = L.bc[8];
        if (bi >= nblocks) break;
        const hg_bgzf_desc dsc = desc[block0 + bi];
        const uint64_t coff = dsc.coff;
        const uint32_t clen = dsc.clen, ulen = dsc.ulen;
        uint32_t *tok = tokbuf + (size_t)bi * TOK_CAP;
        if (clen < 26u || coff + clen > comp_len || ulen > 65536u) {
            if (tid == 0) { ntok_out[bi] = 0; pst_out[bi] = ST_HEADER; }
            continue;
        }
        const uint64_t base_dw = coff >> 2;
        const uint32_t skew = (uint32_t)(coff & 3u);
        const uint32_t *g = (const uint32_t *)comp + base_dw;
        const uint32_t max_dw = (uint32_t)(max_dw_abs - base_dw);
        const uint32_t in_end = skew + clen, end_bit = in_end * 8u;
        P_T0(tblk); P_T0(tph);
        BitReader br;                                              // used by wave 0 only
        br.g = g; br.max_dw = max_dw; br.wbase = 0; br.win = 0; br.win_next = 0; br.next_dw = 0; br.bb = 0; br.bc = 0;
        if (wave == 0) {
            br.win = br_gload(br, (uint32_t)lane);
            br.win_next = br_gload(br, 64u + (uint32_t)lane);
            // BGZF header (bgzf.c:896-903 check_header + BSIZE)
            const uint8_t *hb = comp + coff;
            const bool ok = hb[0] == 31 && hb[1] == 139 && hb[2] == 8 && (hb[3] & 4) && hb[10] == 6 && hb[11] == 0 && hb[12] == 'B' && hb[13] == 'C' &&
                            hb[14] == 2 && hb[15] == 0 && ((uint32_t)hb[16] | ((uint32_t)hb[17] << 8)) + 1u == clen;
            if (lane == 0) { L.bc[1] = ok ? ST_OK : ST_HEADER; L.bc[6] = 0; L.bc[7] = 0; }
            if (ok) br_seek(br, skew + 18u, lane);
        }
        __syncthreads();
        // ---- deflate blocks ------------------------------------------------------------------------------------------
        for (;;) {
            if (L.bc[1] != ST_OK) break;
            __syncthreads();
            if (wave == 0) {
                uint32_t bfinal = 0, btype = 0, slen = 0, ssrc = 0;
                const int st = block_prologue(L.T, br, in_end, lane, bfinal, btype, slen, ssrc);
                if (lane == 0) {
                    L.bc[1] = (uint32_t)st; L.bc[2] = bfinal; L.bc[3] = btype; L.bc[4] = slen; L.bc[5] = ssrc;
                    L.bc[0] = br.next_dw * 32u - br.bc;           // first symbol bit
                }
            }
            __syncthreads();
            P_ACC(1, tph);
            if (L.bc[1] != ST_OK) break;
            const uint32_t bfinal = L.bc[2], btype = L.bc[3];
            if (btype == 0) {
                // stored block: its bytes become literal tokens
                const uint32_t slen = L.bc[4], ssrc = L.bc[5], t0 = L.bc[6];
                const bool fits = t0 + slen <= TOK_CAP;
                const uint8_t *sp = (const uint8_t *)g + ssrc;
                if (fits) for (uint32_t i = (uint32_t)tid; i < slen; i += NL) tok[t0 + i] = sp[i];
                __syncthreads();
                if (tid == 0) { if (fits) { L.bc[6] = t0 + slen; L.bc[7] += slen; } else L.bc[1] = ST_INFLATE; }
                if (wave == 0 && fits) br_seek(br, ssrc + slen, lane);
            } else {
                // ---- rounds of NL segments until the end-of-block code -----------------------------------------------------
                for (;;) {
                    const uint32_t B = L.bc[0];
                    const uint32_t base_dw = B >> 5, base_bit = base_dw << 5;
#if HG2_STAGE
                    for (uint32_t d = (uint32_t)tid; d < STAGE_DW; d += NL) {          // coalesced copy of the round's dwords
                        const uint32_t a_dw = base_dw + d;
                        L.inb[d + (d >> 6)] = g[a_dw < max_dw ? a_dw : max_dw];
                    }
                    const uint32_t *inb = L.inb;
#else
                    const uint32_t *inb = g + base_dw;              // reads stay below end_bit + 64 bits <= the padded end of the image
#endif
                    const uint32_t Ni = B + (uint32_t)tid * SEG, lim = Ni + SEG;
                    L.entry[tid] = tid == 0 ? B : (Ni < end_bit ? Ni : IDLE);
                    uint32_t last = ~L.entry[tid];                 // "never parsed"
                    for (;;) {
                        __syncthreads();
                        const uint32_t e = L.entry[tid];
#ifdef HG_PROFILE
                        { const unsigned long long dm = __ballot(e != last && e != IDLE && e < end_bit);
                          if (lane == 0 && dm) { atomicAdd(&g_prof2[7], 1ull); atomicAdd(&g_prof2[13], (unsigned long long)__popcll(dm)); } }
#endif
                        if (e != last) {
                            last = e;
                            uint32_t x = IDLE, f = P_IDLE, nt = 0, nb = 0;
                            if (e != IDLE && e < end_bit) lane_parse<false>(inb, base_bit, max_dw - base_dw, L.T.lit, L.T.dist, e, lim, end_bit, x, f, nt, nb, nullptr);
                            L.exitb[tid] = x; L.flag[tid] = f; L.ntok[tid] = nt; L.nbyte[tid] = nb;
                        }
                        __syncthreads();
                        const uint32_t ne = tid == 0 ? B : (L.flag[tid - 1] == P_OK ? L.exitb[tid - 1] : IDLE);
                        const int changed = ne != e;
                        if (changed) L.entry[tid] = ne;
                        P_CNT(5, 1);
                        if (!__syncthreads_or(changed)) break;
                    }
                    P_ACC(2, tph); P_CNT(6, 1);
                    // first lane that did not run to the end of its segment: end-of-block, error, or out of input
                    const uint32_t f = L.flag[tid];
                    const unsigned long long bal = __ballot(f != P_OK);
                    if (lane == 0) L.wsum[wave] = bal ? (uint32_t)(wave * 64 + __builtin_ctzll(bal)) : (uint32_t)NL;
                    __syncthreads();
                    uint32_t k = L.wsum[0];
#pragma unroll
                    for (int w = 1; w < NW; w++) k = k < L.wsum[w] ? k : L.wsum[w];
                    const bool flagged = k < (uint32_t)NL;
                    const uint32_t fk = flagged ? L.flag[k] : (uint32_t)P_OK;
                    const uint32_t last_lane = flagged ? k : (uint32_t)NL - 1u;
                    __syncthreads();                               // wsum is reused below
                    // token offsets: prefix sum of the counts of lanes 0..last_lane
                    const uint32_t mine = (uint32_t)tid <= last_lane ? L.ntok[tid] : 0u, myb = (uint32_t)tid <= last_lane ? L.nbyte[tid] : 0u;
                    const uint32_t incl = wave_incl_scan_dpp(mine), inclb = wave_incl_scan_dpp(myb);
                    if (lane == 63) { L.wsum[wave] = incl; L.wsum[4 + wave] = inclb; }
                    __syncthreads();
                    uint32_t off = incl - mine, tot = 0, totb = 0;
#pragma unroll
                    for (int w = 0; w < NW; w++) { const uint32_t t = L.wsum[w]; if (w < wave) off += t; tot += t; totb += L.wsum[4 + w]; }
                    const uint32_t t0 = L.bc[6];
                    const bool bad = (flagged && fk != P_EOB) || t0 + tot > TOK_CAP;
                    if (!bad && (uint32_t)tid <= last_lane && mine) {
                        uint32_t x, ff, nt, nb;
                        lane_parse<true>(inb, base_bit, max_dw - base_dw, L.T.lit, L.T.dist, L.entry[tid], lim, end_bit, x, ff, nt, nb, tok + t0 + off);
                    }
                    const uint32_t next_bit = L.exitb[last_lane];
                    __syncthreads();
                    if (tid == 0) {
                        if (bad) L.bc[1] = ST_INFLATE;
                        else { L.bc[6] = t0 + tot; L.bc[7] += totb; L.bc[0] = next_bit; }
                    }
                    __syncthreads();
                    P_ACC(3, tph);
                    if (bad || flagged) break;                     // error, or end of this deflate block
                }
                if (wave == 0 && L.bc[1] == ST_OK) br_seek_bits(br, L.bc[0], lane);
            }
            __syncthreads();
            if (L.bc[1] != ST_OK || bfinal) break;
        }
        __syncthreads();
        if (wave == 0) {
            int st = (int)L.bc[1];
            if (st == ST_OK && br_byte_pos(br) > in_end) st = ST_INFLATE;
            if (st == ST_OK && L.bc[7] != ulen) st = ST_SIZE;
            if (lane == 0) { ntok_out[bi] = L.bc[6]; pst_out[bi] = st; }
        }
        P_ACC(0, tblk); P_CNT(4, 1);
    }
}

// The resolve kernel keeps the most recent RING2 bytes of output in LDS and lets ALL output reach HBM through it:
// literals and matches are written to the ring only; whenever the write position passes a FLUSH-byte boundary the finished
// FLUSH bytes leave with one 16-byte store per lane.  (v1 stored every match with its own partially filled store instruction:
// ~3.8 k vector-memory instructions per block; here ~70.)  Ring indices are positions shifted by `a` = the misalignment of
// the block's output address, so that flush units are 16-byte aligned in both LDS and HBM.
constexpr uint32_t RING2 = 4096u, FLUSH = 1024u;
constexpr uint32_t RING2_NEAR = RING2 - FLUSH - 64u;   // a source this close is certainly still in the ring

struct ResolveLds { uint4 ring[4][RING2 / 16]; };

// store ring bytes [q0, q1) (q = position + a; both multiples of 16 except at the very ends of the block) to o_al + q
__device__ __forceinline__ void flush_span(const uint8_t *ring, uint8_t *o_al, uint32_t q0, uint32_t q1, uint32_t qmin, uint32_t qmax, int lane) {
    // q0 is a multiple of FLUSH or qmin rounded down; bytes outside [qmin, qmax) do not belong to this block
    for (uint32_t q = q0 + (uint32_t)lane * 16u; q < q1; q += 1024u) {
        const uint4 v = *(const uint4 *)(ring + (q & (RING2 - 1u)));
        if (q >= qmin && q + 16u <= qmax) *(uint4 *)(o_al + q) = v;
        else {
            const uint8_t *pb = (const uint8_t *)&v;
#pragma unroll
            for (int k = 0; k < 16; k++) if (q + (uint32_t)k >= qmin && q + (uint32_t)k < qmax) o_al[q + k] = pb[k];
        }
    }
}

__global__ __launch_bounds__(256)
void resolve_kernel(const uint8_t *__restrict__ comp, const hg_bgzf_desc *__restrict__ desc, uint32_t block0, uint32_t nblocks,
                    const uint32_t *__restrict__ tokbuf, const uint32_t *__restrict__ ntok_in, const int32_t *__restrict__ pst_in, uint8_t *out,
                    uint64_t out_cap, int32_t *status, unsigned int *ticket) {
    __shared__ ResolveLds RL;
    const int lane = lane_id();
    uint8_t *ring = (uint8_t *)RL.ring[uni(threadIdx.x >> 6)];
    for (;;) {
        uint32_t bi = atomicAdd(ticket, lane == 0 ? 1u : 0u);
        bi = (uint32_t)__builtin_amdgcn_readlane((int)bi, 0);
        if (bi >= nblocks) break;
        P_T0(tblk); P_T0(tph);
        const hg_bgzf_desc dsc = desc[block0 + bi];
        const uint64_t coff = ((uint64_t)uni((uint32_t)(dsc.coff >> 32)) << 32) | uni((uint32_t)dsc.coff);
        const uint64_t uoff = ((uint64_t)uni((uint32_t)(dsc.uoff >> 32)) << 32) | uni((uint32_t)dsc.uoff);
        const uint32_t clen = uni(dsc.clen), ulen = uni(dsc.ulen);
        int st = (int)uni((uint32_t)pst_in[bi]);
        if (st == ST_OK && uoff + ulen > out_cap) st = ST_HEADER;
        if (st == ST_OK) {
            const uint32_t n = uni(ntok_in[bi]);
            const uint32_t *tok = tokbuf + (size_t)bi * TOK_CAP;
            uint8_t *o = out + uoff;
            const uint32_t a = (uint32_t)((uintptr_t)o & 15u);      // q = position + a
            uint8_t *o_al = o - a;
            const uint32_t qmin = a, qmax = a + ulen;
            uint32_t pos = 0, flushed = 0;                          // flushed: q up to which the ring has been stored (multiple of FLUSH)
            for (uint32_t t0 = 0; t0 < n && st == ST_OK; t0 += 64u) {
                const bool act = t0 + (uint32_t)lane < n;
                const uint32_t tk = act ? tok[t0 + (uint32_t)lane] : 0u;
                const bool ism = act && (tk >> 31) != 0u;
                const uint32_t len = !act ? 0u : ism ? ((tk >> 16) & 0xffu) + 3u : 1u;
                const uint32_t incl = wave_incl_scan_dpp(len);
                const uint32_t mypos = pos + incl - len;
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (pos + total > ulen) { st = ST_INFLATE; break; }
                unsigned long long mm = __ballot(ism);
                uint32_t lit_from = 0;
                while (mm) {
                    const uint32_t Lm = (uint32_t)__builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    // the literals between the previous match and this one enter the ring first (ring order = output order)
                    if ((uint32_t)lane >= lit_from && (uint32_t)lane < Lm) ring[(mypos + a) & (RING2 - 1u)] = (uint8_t)tk;
                    lit_from = Lm + 1u;
                    const uint32_t mpos = (uint32_t)__builtin_amdgcn_readlane((int)mypos, (int)Lm);
                    const uint32_t mlen = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)Lm);
                    const uint32_t mdist = ((uint32_t)__builtin_amdgcn_readlane((int)tk, (int)Lm) & 0x7fffu) + 1u;
                    if (mdist > mpos) { st = ST_INFLATE; break; }
                    // everything older than FLUSH + 64 bytes is in HBM: make room before this match can pass a flush boundary
                    if (((mpos + a) & ~(FLUSH - 1u)) > flushed) {
                        const uint32_t upto = (mpos + a) & ~(FLUSH - 1u);
                        flush_span(ring, o_al, flushed, upto, qmin, qmax, lane);
                        flushed = upto;
                    }
#ifndef HG2_NO_FAST
                    if (mlen <= 64u && mdist >= mlen && mdist <= RING2_NEAR) {
                        // the common match: one LDS read, one LDS write, no loop
                        const uint8_t v = ring[(mpos - mdist + (uint32_t)lane + a) & (RING2 - 1u)];
                        if ((uint32_t)lane < mlen) ring[(mpos + (uint32_t)lane + a) & (RING2 - 1u)] = v;
                        continue;
                    }
#endif
                    uint32_t done = 0, span = mdist;
                    do {                                              // spans double for overlapping matches
                        uint32_t c = mlen - done;
                        c = c < 64u ? c : 64u;
                        c = c < span ? c : span;
                        const uint32_t dst = mpos + done;
                        uint8_t v;
                        if (span <= RING2_NEAR) v = ring[(dst - span + (uint32_t)lane + a) & (RING2 - 1u)];
                        else v = o[dst - span + ((uint32_t)lane < c ? (uint32_t)lane : 0u)];   // far: flushed long ago
                        if ((uint32_t)lane < c) ring[(dst + (uint32_t)lane + a) & (RING2 - 1u)] = v;
                        done += c;
                        span = c == span ? span << 1 : span;
                    } while (done < mlen);
                }
                if (st != ST_OK) break;
                if ((uint32_t)lane >= lit_from && act) ring[(mypos + a) & (RING2 - 1u)] = (uint8_t)tk;
                pos += total;
                if (((pos + a) & ~(FLUSH - 1u)) > flushed) {
                    const uint32_t upto = (pos + a) & ~(FLUSH - 1u);
                    flush_span(ring, o_al, flushed, upto, qmin, qmax, lane);
                    flushed = upto;
                }
            }
            P_ACCW(9, tph);
            if (st == ST_OK && pos != ulen) st = ST_SIZE;
            if (st == ST_OK) {
                flush_span(ring, o_al, flushed, (qmax + 15u) & ~15u, qmin, qmax, lane);          // the tail
                P_ACCW(10, tph);
                const uint8_t *t = comp + coff + clen - 8;
                uint32_t c = 0, z = 0;
                for (int k = 0; k < 4; k++) { c |= (uint32_t)t[k] << (8 * k); z |= (uint32_t)t[4 + k] << (8 * k); }
                if (uni(z) != ulen) st = ST_SIZE;
                else if (uni(wave_crc32(o, ulen, lane)) != uni(c)) st = ST_CRC;
                P_ACCW(11, tph);
            }
        }
        status[block0 + bi] = st == ST_OK ? HG_BLOCK_OK : st == ST_CRC ? HG_BLOCK_ECRC : HG_BLOCK_EINFLATE;
        P_ACCW(8, tblk); P_CNTW(12, 1);
    }
}

}  // namespace hg2

#ifdef HG_PROFILE
extern "C" int hg_debug_get_profile2(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(hg2::g_prof2), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hg2::g_prof2), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

namespace hg {

// Blocks are processed in chunks of CHUNK: parse (tokens -> HBM) then resolve, back to back on the caller's stream.
int launch_bgzf_inflate_v2(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc, size_t nblocks, void *d_out,
                           size_t out_cap, int32_t *d_status, hipStream_t s) {
    constexpr size_t CHUNK = 8192;
    if (nblocks == 0) return HG_OK;
    if (nblocks > 0xffffffffull) return HG_EINVAL;
    const size_t chunk = nblocks < CHUNK ? nblocks : CHUNK;
    std::lock_guard<std::mutex> order(*ctx->tok_mu);                // the token buffer belongs to the context: launches are ordered
    const size_t need = chunk * (size_t)hg2::TOK_CAP * 4 + chunk * 8 + 256;
    if (ctx->d_tok2_cap < need) {
        if (ctx->ev_inflate2_used && hipEventSynchronize(ctx->ev_inflate2) != hipSuccess) return HG_ELAUNCH;
        if (ctx->d_tok2) (void)hipFree(ctx->d_tok2);
        ctx->d_tok2 = nullptr; ctx->d_tok2_cap = 0;
        if (hipMalloc(&ctx->d_tok2, need) != hipSuccess) return HG_ENOMEM;
        ctx->d_tok2_cap = need;
    }
    if (ctx->ev_inflate2_used && hipStreamWaitEvent(s, ctx->ev_inflate2, 0) != hipSuccess) return HG_ELAUNCH;
    uint32_t *tokbuf = (uint32_t *)ctx->d_tok2;
    uint32_t *ntok = tokbuf + chunk * (size_t)hg2::TOK_CAP;
    int32_t *pst = (int32_t *)(ntok + chunk);
    for (size_t c0 = 0; c0 < nblocks; c0 += chunk) {
        const size_t n = nblocks - c0 < chunk ? nblocks - c0 : chunk;
        unsigned int *t1 = next_ticket(ctx), *t2 = next_ticket(ctx);
        if (hipMemsetAsync(t1, 0, sizeof(unsigned int), s) != hipSuccess || hipMemsetAsync(t2, 0, sizeof(unsigned int), s) != hipSuccess) return HG_ELAUNCH;
        size_t wg1 = (size_t)ctx->cus * (HG2_STAGE ? (hg2::NL == 64 ? 6 : 3) : (hg2::NL == 64 ? 20 : 8)); if (wg1 > n) wg1 = n;
        hipLaunchKernelGGL(hg2::parse_kernel, dim3((unsigned)wg1), dim3(hg2::NL), 0, s, (const uint8_t *)d_comp, (uint64_t)comp_len, d_desc,
                           (uint32_t)c0, (uint32_t)n, tokbuf, ntok, pst, t1);
        size_t wg2 = (size_t)ctx->cus * 8; const size_t need2 = (n + 3) / 4; if (wg2 > need2) wg2 = need2;
        hipLaunchKernelGGL(hg2::resolve_kernel, dim3((unsigned)wg2), dim3(256), 0, s, (const uint8_t *)d_comp, d_desc, (uint32_t)c0, (uint32_t)n,
                           (const uint32_t *)tokbuf, (const uint32_t *)ntok, (const int32_t *)pst, (uint8_t *)d_out, (uint64_t)out_cap, d_status, t2);
        if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    }
    if (hipEventRecord(ctx->ev_inflate2, s) != hipSuccess) return HG_ELAUNCH;
    ctx->ev_inflate2_used = 1;
    return HG_OK;
}

}  // namespace hg

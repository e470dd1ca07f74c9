// bgzf_inflate.hip -- BGZF block inflate for MI355X (gfx950 / CDNA4).
//
// Replaces the worker hot loop of htslib's reader, bgzf_decode_func ->
// bgzf_uncompress (reference bgzf.c:1373-1384, 730-804): raw DEFLATE decode
// (RFC 1951) of one <=64 KiB block + CRC-32 check against the trailer.
//
// Mapping (designed for CDNA4, not translated from a CPU inflate):
//   * one BGZF block per 64-lane wavefront, one wavefront per workgroup; wavefronts are persistent and pull block indices from
//     a per-launch ticket counter, so the 168 k blocks of a 10 GiB BAM load-balance over 256 CUs without a tail;
//   * the compressed stream is read with coalesced 256-byte wave loads: the wave keeps a 64-dword window of the input in ONE
//     VGPR (+ the next window prefetched in a second one) and the bit reader pulls dwords out of it with v_readlane;
//   * decode tables live in LDS (9-bit litlen root + zlib-style second level, 8-bit distance root: 3.5 KiB per wave) and are
//     built by all 64 lanes (LDS-atomic histogram, ballot-ranked canonical codes); root-table literals carry a sign bit so
//     that the hot path is one compare;
//   * the symbol loop of a deflate block is hand-scheduled assembly (inflate_loop_asm.inc): every lane computes the same
//     decode state, so what bounds the loop is instruction issue -- one vector and one scalar instruction per SIMD per
//     quad-cycle -- and the work is split evenly between the two sides by hand (17.9 VALU + 15.9 SALU per symbol against the
//     compiler's 39 + 15; profiles/r03_inflate_instruction_mix.txt);
//   * output goes into a 1 KiB LDS ring per wave and nowhere else: a literal is one ds_write of the table entry, a match is
//     ring -> ring (or global -> ring for a look-back beyond the ring: the wave's own flushed output, whose load is not waited
//     for until the next copy needs the ring); every 256 finished bytes leave the ring as one coalesced dword store per lane;
//   * the CRC-32 is fused: after the last deflate block the wave re-reads its output (L2-resident), 64 lanes x slice-by-4, and
//     folds the partials with a 6-step butterfly;
//   * 64 VGPRs, 4.5 KiB of LDS per wave (16-bit distance / second-level entries; table builder and CRC pass as calls) -> 32
//     wavefronts per CU.  gzip members of any length (CRAM GZIP blocks, plain .gz files)
//     use the same decoder in a resumable form (mode 1, gzip_stream_kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hg {

// Bring-up tracing (tests/native/diag.cpp): compiled in only with -DHG_DEBUG_TRACE.
#ifdef HG_DEBUG_TRACE
__device__ volatile uint32_t *g_trace = nullptr;    // host-pinned, 64 words per block
__device__ uint32_t g_trace_blk = 0;
#define HG_TRACE(slot, val) do { if (g_trace && lane_id() == 0) { g_trace[16 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + (slot)] = (uint32_t)(val); __threadfence_system(); } } while (0)
#else
#define HG_TRACE(slot, val) do { } while (0)
#endif

// In-kernel phase timing (s_memtime), compiled in only with -DHG_PROFILE (kbench).
#ifdef HG_PROFILE
__device__ unsigned long long g_prof[16];   // 0 total, 1 header+tables, 2 symbols, 3 resolve, 4 crc, 5 blocks, 6 stored
#define HG_T0(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define HG_TACC(slot, var) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) atomicAdd(&g_prof[slot], n_ - var); var = n_; } while (0)
#define HG_CNT(slot, v) do { if (lane_id() == 0) atomicAdd(&g_prof[slot], (unsigned long long)(v)); } while (0)
#else
#define HG_T0(var) do { } while (0)
#define HG_TACC(slot, var) do { } while (0)
#define HG_CNT(slot, v) do { } while (0)
#endif

}  // namespace hg
#include "inflate_common.h"
namespace hg {

// status codes internal to the kernel
enum { ST_OK = 0, ST_HEADER = 1, ST_INFLATE = 2, ST_SIZE = 3, ST_CRC = 4, ST_PAUSE = 5 };

// Resumable decoding (plain gzip streams, hg_gzip_stream_inflate_host): where to start, when to pause, and the
// last deflate-block boundary reached -- the only places a decode can be suspended without saving Huffman tables.
struct StreamCtl {
    uint32_t start_bit;      // in: first bit of a deflate block, relative to br.g
    uint32_t pos0;           // in: bytes of history already present at out[0..pos0)
    uint32_t soft_cap;       // in: pause at the first block boundary with pos - pos0 >= soft_cap
    uint32_t good_bit;       // out: bit position of the last block boundary reached
    uint32_t good_pos;       // out: output position at that boundary
};

// Decode one raw deflate stream.  `in_end` = first byte (relative to br.g)
// that is NOT part of the payload.  Returns ST_*; *out_len receives bytes made.
template <bool RESUMABLE>
__device__ __forceinline__ int inflate_stream(WaveLds &S, BitReader &br, uint32_t in_start, uint32_t in_end,
                              uint8_t *out, uint32_t cap, uint32_t *out_len, int lane, StreamCtl *ctl = nullptr) {
    uint32_t pos = 0;
    HG_T0(tph);
    HG_TRACE(2, 1);
    if (RESUMABLE) { br_seek_bits(br, ctl->start_bit, lane); pos = ctl->pos0; }
    else br_seek(br, in_start, lane);
    HG_TRACE(2, 2);
    for (;;) {
        if (RESUMABLE) {
            ctl->good_bit = br.next_dw * 32u - br.bc; ctl->good_pos = pos;
            if (pos - ctl->pos0 >= ctl->soft_cap) { *out_len = pos; return ST_PAUSE; }
        }
        br_refill(br, lane);
        HG_TRACE(2, 3);
        if (br_byte_pos(br) > in_end) return ST_INFLATE;
        uint32_t bfinal = br_bits(br, 1);
        uint32_t btype = br_bits(br, 2);
        if (btype == 0) {
            // ---- stored (RFC 1951 3.2.4) ----
            br_drop(br, br.bc & 7u);
            br_refill(br, lane);
            uint32_t len = br_bits(br, 16);
            br_refill(br, lane);
            uint32_t nlen = br_bits(br, 16);
            if ((len ^ 0xffffu) != nlen) return ST_INFLATE;
            uint32_t src = br_byte_pos(br);
            if (src + len > in_end || pos + len > cap) return ST_INFLATE;
            const uint8_t *sp = (const uint8_t *)br.g + src;
            for (uint32_t i = lane; i < len; i += 64) out[pos + i] = sp[i];
            pos += len;
            if (!bfinal) {
                // later blocks may reach back into these bytes: mirror the tail into the LDS ring
                uint32_t lo = pos > RING ? pos - RING : 0u;
                for (uint32_t p = lo + lane; p < pos; p += 64) S.u.ring[p & (RING - 1u)] = out[p];
                wave_sync();
            }
            br_seek(br, src + len, lane);
        } else if (btype == 3) {
            return ST_INFLATE;
        } else {
            if (btype == 1) {
                // ---- fixed codes (RFC 1951 3.2.6) ----
                for (int i = lane; i < 288; i += 64)
                    S.u.b.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                if (lane < 32) S.u.b.lens[288 + lane] = 5;
                wave_sync();
                if (uni((uint32_t)build_table<KIND_DIST, DIST_RB, DIST_TAB, 1>(S, S.lit, 288, 32, lane))) return ST_INFLATE;
                for (int i = lane; i < DIST_TAB; i += 64) S.dist[i] = (uint16_t)S.lit[i];      // built in the litlen area, kept as 16-bit entries
                wave_sync();
                if (uni((uint32_t)build_table<KIND_LITLEN, LIT_RB, LIT_TAB, 5>(S, S.lit, 0, 288, lane))) return ST_INFLATE;
            } else {
                // ---- dynamic codes (RFC 1951 3.2.7) ----
                br_refill(br, lane);
                uint32_t nlen = br_bits(br, 5) + 257;
                uint32_t ndist = br_bits(br, 5) + 1;
                uint32_t ncode = br_bits(br, 4) + 4;
                if (nlen > 286 || ndist > 30) return ST_INFLATE;
                // code-length code lengths: 3 bits each, permuted order
                if (lane < 19) S.u.b.lens[lane] = 0;
                wave_sync();
                for (uint32_t i = 0; i < ncode; i++) {
                    br_refill(br, lane);
                    uint32_t v = br_bits(br, 3);
                    // order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 packed 5 bits each
                    const uint64_t ord_lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) |
                                            (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) |
                                            (5ull << 45) | (11ull << 50) | (4ull << 55);
                    const uint64_t ord_hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) |
                                            (1ull << 25) | (15ull << 30);
                    uint32_t sym = i < 12 ? (uint32_t)(ord_lo >> (5 * i)) & 31u
                                          : (uint32_t)(ord_hi >> (5 * (i - 12))) & 31u;
                    if (lane == 0) S.u.b.lens[sym] = (uint8_t)v;
                }
                wave_sync();
                if (uni((uint32_t)build_table<KIND_PRE, PRE_RB, DIST_TAB, 1>(S, S.lit, 0, 19, lane))) return ST_INFLATE;
                // the precode must be complete unless trivially small: zlib rejects incomplete
                // code-length codes outright; build_table allows the 1-code case, which a
                // conforming encoder never emits -- keep oracle behaviour (complete only).
                {
                    uint32_t c = lane < 16 ? S.u.b.cnt[lane] : 0;
                    int left = 1;
#pragma unroll
                    for (int l = 1; l <= 7; l++)
                        left = (left << 1) - (int)__builtin_amdgcn_readlane((int)c, l);
                    if (left != 0) return ST_INFLATE;
                }
                // literal/length + distance code lengths, run-length coded
                uint32_t idx = 0, total = nlen + ndist, prev = 0;
                while (idx < total) {
                    br_refill(br, lane);
                    uint32_t e = lds_uniform(&S.lit[br_peek(br, PRE_RB)]);
                    if (!(e & F_LIT)) return ST_INFLATE;
                    br_drop(br, e & 15u);
                    uint32_t sym = e >> 16;
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
                if (S.u.b.lens[32 + 256] == 0) return ST_INFLATE;          // no end-of-block code
                // lengths sit at lens[32 .. 32+nlen+ndist): the precode table (in the litlen area) is dead now
                if (uni((uint32_t)build_table<KIND_DIST, DIST_RB, DIST_TAB, 1>(S, S.lit, 32 + (int)nlen, (int)ndist, lane))) return ST_INFLATE;
                for (int i = lane; i < DIST_TAB; i += 64) S.dist[i] = (uint16_t)S.lit[i];      // built in the litlen area, kept as 16-bit entries
                wave_sync();
                if (uni((uint32_t)build_table<KIND_LITLEN, LIT_RB, LIT_TAB, 5>(S, S.lit, 32, (int)nlen, lane))) return ST_INFLATE;
            }
            HG_TACC(1, tph);
            // the table-build scratch overlays the output ring: restore the ring from the wave's own
            // output (visible to it in program order) when this is not the first deflate block
            if (pos) {
                const uint32_t lo = pos > RING ? pos - RING : 0u;
                for (uint32_t p = lo + (uint32_t)lane; p < pos; p += 64) S.u.ring[p & (RING - 1u)] = out[p];
            }
            wave_sync();
#if defined(HG_LOOP_VEC)            /* A/B builds only (scripts/mk_variant.sh ... -DHG_LOOP_VEC -Iexperiments): the compiled loops the assembly replaced */
#include "inflate_loop_vec.inc"
#elif defined(HG_LOOP_MIX)
#include "inflate_loop_mix.inc"
#else
#include "inflate_loop_asm.inc"
#endif
            if (br_byte_pos(br) > in_end) return ST_INFLATE;
        }
        if (bfinal) break;
    }
    if (RESUMABLE) { ctl->good_bit = br.next_dw * 32u - br.bc; ctl->good_pos = pos; }
    *out_len = pos;
    return ST_OK;
}

#ifndef HG_INFLATE_MIN_WAVES
#define HG_INFLATE_MIN_WAVES 8      // <= 64 VGPRs: 8 waves per SIMD, with the LDS at 4.5 KiB per wavefront
#endif
__global__ __launch_bounds__(WAVES_PER_WG * 64, HG_INFLATE_MIN_WAVES)
void bgzf_inflate_kernel(const uint8_t *__restrict__ comp, uint64_t comp_len,
                         const hg_bgzf_desc *__restrict__ desc, uint32_t nblocks,
                         uint8_t *out, uint64_t out_cap, int32_t *status,
                         unsigned int *ticket, int mode) {
    // mode 0: BGZF blocks (strict check_header + BSIZE);  mode 1: generic gzip members of any
    // length (CRAM block method GZIP, zlib_mem_inflate, cram/cram_io.c:1068-1110)
    __shared__ WaveLds lds[WAVES_PER_WG];
    const int lane = lane_id();
    const int wave = (int)uni(threadIdx.x >> 6);
    WaveLds &S = lds[wave];
    const uint64_t max_dw_abs = (comp_len + 3) / 4 - 1;   // buffer is padded to a dword multiple

    uint32_t iter = 0; (void)iter;
    for (;;) {
        uint32_t b = 0;
        iter++;
        HG_TRACE(12, iter);
        // every lane takes part (lane 0 adds 1, the others add 0): no lane-0-only control
        // flow at the loop back-edge, which hipcc (ROCm 7.2) was seen to merge with the
        // lane-0-only status store of the previous iteration and hang the wave.
        b = atomicAdd(ticket, lane == 0 ? 1u : 0u);
        b = (uint32_t)__builtin_amdgcn_readlane((int)b, 0);
        HG_TRACE(13, b + 1000);
        if (b >= nblocks) break;
        HG_TRACE(0, b + 1);
        HG_T0(tblk);
        const hg_bgzf_desc dsc = desc[b];
        const uint64_t coff = ((uint64_t)uni((uint32_t)(dsc.coff >> 32)) << 32) | uni((uint32_t)dsc.coff);
        const uint64_t uoff = ((uint64_t)uni((uint32_t)(dsc.uoff >> 32)) << 32) | uni((uint32_t)dsc.uoff);
        const uint32_t clen = uni(dsc.clen), ulen = uni(dsc.ulen);
        int st = ST_OK;
        if (clen < (mode == 1 ? 18u : 26u) || coff + clen > comp_len || uoff + ulen > out_cap) {
            st = ST_HEADER;
        } else {
            BitReader br;
            const uint64_t base_dw = coff >> 2;
            const uint32_t skew = (uint32_t)(coff & 3u);
            br.g = (const uint32_t *)comp + base_dw;
            br.max_dw = (uint32_t)(max_dw_abs - base_dw);
            br.wbase = 0;
            br.win = br_gload(br, (uint32_t)lane);
            br.win_next = br_gload(br, 64u + (uint32_t)lane);
            br.next_dw = 0; br.bb = 0; br.bc = 0;
            // header (bgzf.c:896-903 check_header + BSIZE) and trailer
            const uint8_t *hb = comp + coff;
            uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
            {
                // 18 header bytes via the window (dword granular, skewed by coff&3)
                uint32_t w[6];
#pragma unroll
                for (int i = 0; i < 6; i++) w[i] = (uint32_t)__builtin_amdgcn_readlane((int)br.win, i);
                auto byte_at = [&](int k) -> uint32_t {
                    uint32_t p = skew + (uint32_t)k;
                    return (w[p >> 2] >> ((p & 3u) * 8u)) & 0xffu;
                };
                h0 = byte_at(0) | (byte_at(1) << 8) | (byte_at(2) << 16) | (byte_at(3) << 24);
                h1 = byte_at(10) | (byte_at(11) << 8);
                h2 = byte_at(12) | (byte_at(13) << 8);
                h3 = byte_at(14) | (byte_at(15) << 8);
                h4 = byte_at(16) | (byte_at(17) << 8);
            }
            bool hdr_ok = (h0 & 0x04ffffffu) == 0x04088b1fu && h1 == 6 && h2 == 0x4342u && h3 == 2 &&
                          h4 + 1u == clen;
            uint32_t pay = 18u;                                  // payload offset inside the member
            if (mode == 1) {
                // RFC 1952 member header: ID1 ID2 CM FLG MTIME(4) XFL OS [XLEN+extra] [name] [comment] [hcrc]
                hdr_ok = (h0 & 0x00ffffffu) == 0x00088b1fu && (h0 >> 29) == 0;
                const uint32_t flg = h0 >> 24;
                uint32_t q = 10;
                if (hdr_ok && (flg & 4u)) { q += 2u + h1; }
                if (hdr_ok && (flg & 8u)) { while (q < clen && uni(hb[q]) != 0) q++; q++; }
                if (hdr_ok && (flg & 16u)) { while (q < clen && uni(hb[q]) != 0) q++; q++; }
                if (hdr_ok && (flg & 2u)) q += 2;
                if (q + 8u > clen) hdr_ok = false;
                pay = q;
            }
            if (!hdr_ok) {
                st = ST_HEADER;
            } else {
                uint32_t tr[2];
                {
                    const uint8_t *t = hb + clen - 8;
                    uint32_t c = 0, z = 0;
                    for (int k = 0; k < 4; k++) { c |= (uint32_t)t[k] << (8 * k); z |= (uint32_t)t[4 + k] << (8 * k); }
                    tr[0] = uni(c); tr[1] = uni(z);
                }
                uint32_t made = 0;
                uint8_t *o = out + uoff;
                // payload handed to inflate = block[18 .. clen) like the reference (slen = block_length-18)
                st = inflate_stream<false>(S, br, skew + pay, skew + clen, o, ulen, &made, lane);
                HG_TRACE(1, 50 + st);
                if (st == ST_OK && made != ulen) st = ST_SIZE;
                if (st == ST_OK && tr[1] != ulen) st = ST_SIZE;
#ifndef HG_AB_NO_CRC_PASS        /* measurement variant only (scripts/pmc_ab_crc.sh): how much of the FETCH_SIZE counter is the CRC pass re-reading the output */
                if (st == ST_OK) {
                    HG_TRACE(11, 1);
                    HG_T0(tcrc);
                    uint32_t crc = wave_crc32(o, ulen, lane);
                    HG_TACC(4, tcrc);
                    HG_TRACE(11, 2);
                    if (uni(crc) != tr[0]) st = ST_CRC;
                }
#endif
            }
        }
        HG_TRACE(1, 90 + st);
        HG_TACC(0, tblk); HG_CNT(5, 1);
        // all lanes store the same word (one coalesced write)
        status[b] = st == ST_OK ? HG_BLOCK_OK : st == ST_CRC ? HG_BLOCK_ECRC : HG_BLOCK_EINFLATE;
        HG_TRACE(14, 555);
    }
    HG_TRACE(15, 777);
}

// ---- plain gzip stream, one wavefront, resumable at deflate-block boundaries (hg_gzip_stream_inflate_host) ----
struct GzJob {               // device copy of the request / result
    uint64_t in_bit;         // in: where to start; out: where the next call starts
    uint32_t in_member;      // in/out
    uint32_t hist_len;       // in: history bytes at out[0..hist_len)
    uint32_t soft_cap, out_cap;   // in: out_cap counts NEW bytes after the history
    uint32_t comp_eof;       // in
    int32_t status;          // out: HG_GZ_* or -1 (corrupt) / -2 (trailer CRC is checked on the host; unused here)
    uint32_t made;           // out: new bytes at out[hist_len ..)
    uint32_t crc_new;        // out: CRC-32 of the new bytes
    uint32_t trailer_crc, trailer_isize;   // out: when status == HG_GZ_MEMBER
};

__global__ __launch_bounds__(64)
void gzip_stream_kernel(const uint8_t *__restrict__ comp, uint64_t comp_len, uint8_t *out, GzJob *job) {
    __shared__ WaveLds S;
    const int lane = lane_id();
    const uint64_t in_bit = ((uint64_t)uni((uint32_t)(job->in_bit >> 32)) << 32) | uni((uint32_t)job->in_bit);
    const uint32_t hist = uni(job->hist_len), soft = uni(job->soft_cap), ocap = uni(job->out_cap), eof = uni(job->comp_eof);
    uint32_t in_member = uni(job->in_member);
    int32_t status = -1;
    uint32_t made = 0, crc_new = 0, tcrc = 0, tisz = 0;
    uint64_t next_bit = in_bit;
    const uint32_t clen = comp_len > 0x1ffffff0ull ? 0x1ffffff0u : (uint32_t)comp_len;     // bit offsets are 32-bit
    uint32_t q = (uint32_t)(in_bit >> 3);
    bool go = true;
    if (!in_member) {
        // RFC 1952 member header at byte q: ID1 ID2 CM FLG MTIME(4) XFL OS [XLEN + extra] [name] [comment] [hcrc]
        if ((in_bit & 7u) != 0 || q > clen) { go = false; }
        else if (q + 10u > clen) { status = eof ? -1 : HG_GZ_NEEDIN; go = false; }
        else {
            const uint8_t *h = comp + q;
            const uint32_t id = uni((uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16)), flg = uni(h[3]);
            bool short_in = false;
            if (id != 0x088b1fu || (flg >> 5) != 0) go = false;
            uint32_t r = q + 10u;
            if (go && (flg & 4u)) {
                if (r + 2u > clen) short_in = true;
                else r += 2u + uni((uint32_t)comp[r] | ((uint32_t)comp[r + 1] << 8));
            }
            for (int pass = 0; pass < 2 && go && !short_in; pass++)
                if (flg & (pass ? 16u : 8u)) {
                    while (r < clen && uni(comp[r]) != 0) r++;
                    if (r >= clen) short_in = true; else r++;
                }
            if (go && !short_in && (flg & 2u)) r += 2u;
            if (go && (short_in || r > clen)) { status = eof ? -1 : HG_GZ_NEEDIN; go = false; }
            if (go) { next_bit = (uint64_t)r * 8u; in_member = 1; }
        }
    }
    if (go) {
        BitReader br;
        br.g = (const uint32_t *)comp;
        br.max_dw = (uint32_t)((comp_len + 3) / 4 - 1);
        br.wbase = 0;
        br.win = br_gload(br, (uint32_t)lane);
        br.win_next = br_gload(br, 64u + (uint32_t)lane);
        br.next_dw = 0; br.bb = 0; br.bc = 0;
        StreamCtl ctl;
        ctl.start_bit = (uint32_t)next_bit; ctl.pos0 = hist; ctl.soft_cap = soft; ctl.good_bit = ctl.start_bit; ctl.good_pos = hist;
        uint32_t pos_end = hist;
        const int st = inflate_stream<true>(S, br, 0, clen, out, hist + ocap, &pos_end, lane, &ctl);
        uint32_t keep_pos = hist;                                       // output that stands
        if (st == ST_PAUSE) { status = HG_GZ_MORE; keep_pos = pos_end; next_bit = ctl.good_bit; }
        else if (st == ST_OK) {
            const uint32_t t = (ctl.good_bit + 7u) >> 3;                // trailer: CRC32, ISIZE after the final block, byte aligned
            if (t + 8u <= clen) {
                for (int k = 0; k < 4; k++) { tcrc |= (uint32_t)uni(comp[t + k]) << (8 * k); tisz |= (uint32_t)uni(comp[t + 4 + k]) << (8 * k); }
                status = HG_GZ_MEMBER; keep_pos = pos_end; next_bit = (uint64_t)(t + 8u) * 8u; in_member = 0;
            } else status = eof ? -1 : HG_GZ_NEEDIN;                     // trailer not here yet: this call is redone with more input
        } else if (ctl.good_pos > hist) {
            // a block failed or ran past the input: everything up to the last block boundary stands
            status = HG_GZ_MORE; keep_pos = ctl.good_pos; next_bit = ctl.good_bit;
        } else status = eof ? -1 : HG_GZ_NEEDIN;
        made = keep_pos - hist;
        if (made) crc_new = uni(wave_crc32(out + hist, made, lane));
    }
    if (lane == 0) {
        job->in_bit = next_bit; job->in_member = in_member; job->status = status; job->made = made; job->crc_new = crc_new;
        job->trailer_crc = tcrc; job->trailer_isize = tisz;
    }
}

__global__ __launch_bounds__(256)
void crc32_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ off,
                  const uint32_t *__restrict__ len, uint32_t n, uint32_t *crc) {
    const int lane = lane_id();
    uint32_t w = uni((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = w; i < n; i += nw) {
        uint32_t c = wave_crc32(data + off[i], len[i], lane);
        if (lane == 0) crc[i] = c;
    }
}

#ifdef HG_PROFILE
extern "C" int hg_debug_get_profile(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifdef HG_DEBUG_TRACE
extern "C" int hg_debug_set_trace(void *pinned_host_words) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &pinned_host_words, sizeof(void *)) == hipSuccess ? 0 : -1;
}
#endif

// ---------------------------------------------------------------- host launchers
int launch_bgzf_inflate(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc,
                        size_t nblocks, void *d_out, size_t out_cap, int32_t *d_status, hipStream_t s, int mode) {
    if (nblocks == 0) return HG_OK;
    if (nblocks > 0xffffffffull) return HG_EINVAL;
    unsigned int *ticket = next_ticket(ctx);
    if (hipMemsetAsync(ticket, 0, sizeof(unsigned int), s) != hipSuccess) return HG_ELAUNCH;
#ifndef HG_INFLATE_WAVES_PER_CU
#define HG_INFLATE_WAVES_PER_CU 32     // 4.5 KiB of LDS and 64 VGPRs per wavefront: the 8 wavefronts per SIMD the hardware has slots for
#endif
    size_t waves = (size_t)ctx->cus * HG_INFLATE_WAVES_PER_CU;
    size_t wgs = (waves + WAVES_PER_WG - 1) / WAVES_PER_WG;
    size_t need = (nblocks + WAVES_PER_WG - 1) / WAVES_PER_WG;
    if (wgs > need) wgs = need;
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)wgs), dim3(WAVES_PER_WG * 64), 0, s,
                       (const uint8_t *)d_comp, (uint64_t)comp_len, d_desc, (uint32_t)nblocks,
                       (uint8_t *)d_out, (uint64_t)out_cap, d_status, ticket, mode);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

int launch_crc32(hg_ctx *ctx, const void *d_data, const uint64_t *d_off, const uint32_t *d_len, size_t n,
                 uint32_t *d_crc, hipStream_t s) {
    if (n == 0) return HG_OK;
    size_t wgs = (n + 3) / 4;
    size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(crc32_kernel, dim3((unsigned)wgs), dim3(256), 0, s, (const uint8_t *)d_data, d_off, d_len,
                       (uint32_t)n, d_crc);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}

}  // namespace hg

// ---- host-side CRC-32 concatenation (GF(2) polynomial arithmetic): crc(A||B) from crc(A), crc(B), |B| ----
static uint32_t gz_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
static uint32_t gz_crc_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    uint32_t xp = 0x00800000u, acc = 0x80000000u;                    // x^8, x^0
    for (; len_b; len_b >>= 1) { if (len_b & 1) acc = gz_mulmod(acc, xp); xp = gz_mulmod(xp, xp); }
    return gz_mulmod(acc, crc_a) ^ crc_b;
}

extern "C" int hg_gzip_stream_inflate_host(hg_ctx *ctx, const uint8_t *comp, size_t comp_len, int comp_eof, hg_gz_state *state,
                                           const uint8_t *hist, size_t hist_len, uint8_t *out, size_t out_cap, size_t soft_cap,
                                           size_t *out_len) {
    if (!ctx || !state || !out || !out_len || (!comp && comp_len) || (hist_len && !hist) || hist_len > 32768 ||
        out_cap == 0 || out_cap > 0x7fffffffu || comp_len > 0x1ffffff0ull || (state->in_bit >> 3) > comp_len) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    *out_len = 0;
    int rc;
    if ((rc = hg::ensure_scratch(ctx, 0, comp_len + 512)) || (rc = hg::ensure_scratch(ctx, 1, hist_len + out_cap + 256)) ||
        (rc = hg::ensure_scratch(ctx, 2, sizeof(hg::GzJob)))) return rc;
    hg::GzJob job;
    memset(&job, 0, sizeof job);
    job.in_bit = state->in_bit; job.in_member = state->in_member; job.hist_len = (uint32_t)hist_len;
    job.soft_cap = (uint32_t)(soft_cap < out_cap ? soft_cap : out_cap); job.out_cap = (uint32_t)out_cap; job.comp_eof = comp_eof ? 1u : 0u;
    job.status = -1;
    hipStream_t s = ctx->stream;
    uint8_t *d_comp = (uint8_t *)ctx->d_scratch[0], *d_out = (uint8_t *)ctx->d_scratch[1];
    bool ok = hipMemsetAsync(d_comp + (comp_len & ~(size_t)3), 0, 256, s) == hipSuccess &&
              (comp_len == 0 || hipMemcpyAsync(d_comp, comp, comp_len, hipMemcpyHostToDevice, s) == hipSuccess) &&
              (hist_len == 0 || hipMemcpyAsync(d_out, hist, hist_len, hipMemcpyHostToDevice, s) == hipSuccess) &&
              hipMemcpyAsync(ctx->d_scratch[2], &job, sizeof job, hipMemcpyHostToDevice, s) == hipSuccess;
    if (!ok) return HG_ELAUNCH;
    hipLaunchKernelGGL(hg::gzip_stream_kernel, dim3(1), dim3(64), 0, s, (const uint8_t *)d_comp, (uint64_t)comp_len, d_out,
                       (hg::GzJob *)ctx->d_scratch[2]);
    if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    if (hipMemcpyAsync(&job, ctx->d_scratch[2], sizeof job, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    if (job.status < 0) return HG_EBLOCK;
    if (job.made) {
        if (job.made > out_cap) return HG_ELAUNCH;
        if (hipMemcpy(out, d_out + hist_len, job.made, hipMemcpyDeviceToHost) != hipSuccess) return HG_ELAUNCH;
        state->crc = gz_crc_concat(state->crc, job.crc_new, job.made);      // crc(empty) = 0 is the identity
        state->isize += job.made;
    }
    *out_len = job.made;
    state->in_bit = job.in_bit; state->in_member = job.in_member;
    if (job.status == HG_GZ_MEMBER) {
        // trailer check (RFC 1952 2.3.1): CRC-32 and ISIZE of the member that just ended
        const bool good = state->crc == job.trailer_crc && state->isize == job.trailer_isize;
        state->crc = 0; state->isize = 0;
        if (!good) return HG_EBLOCK;
    }
    return job.status;
}


// tok3.hip -- CRAM 3.1 read-name tokeniser (block method 8): name reconstruction / tokenisation kernels
// for MI355X (gfx950).
//
// Replaces tok3_decode_names / tok3_encode_names as called at reference cram/cram_io.c:1735-1749 and
// 1885-1895 (implementation = htscodecs tokenise_name3.c, an ABSENT submodule).  Container and token
// semantics per oracle/tok3_oracle.c -- PARITY UNPINNED; the kernels are bit-exact with that oracle.
//
// A tok3 block is ~20-60 small entropy-coded byte streams, one per (token position, token type), plus
// the rule that rebuilds every name from the previous one.  The streams are decoded by the rANS Nx16 /
// range-coder kernels (planned by cram_entropy_host.hip, all streams of all blocks in one launch).
// What is left is inherently ordered -- name n refers to name n-d -- so the mapping is one wavefront
// per block with the 64 lanes spread over the TOKEN POSITIONS of the current name:
//   lane t owns the read cursors of every stream of position t+1 (kept in LDS), peeks its next TYPE
//   byte, one ballot finds the END token, the lanes before it fetch their token (text, number, delta or
//   a copy of the reference name's token), a DPP prefix sum of the token lengths places them, and every
//   lane writes its own few bytes.  Names with more than 64 tokens take a second pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgt {
using hg::wave_sync;
using hg::wave_incl_scan_dpp;

enum { T_TYPE = 0, T_STRING = 1, T_CHAR = 2, T_DIGITS0 = 3, T_DZLEN = 4, T_DUP = 5, T_DIFF = 6, T_DIGITS = 7,
       T_DELTA = 8, T_DELTA0 = 9, T_MATCH = 10, T_NOP = 11, T_END = 12, NTYPES = 13, MAX_TOK = 128 };

struct WaveLds { uint32_t off[NTYPES][MAX_TOK], len[NTYPES][MAX_TOK], cur[NTYPES][MAX_TOK]; };

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t ndigits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u
         : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}

__global__ __launch_bounds__(64)
void tok3_names_kernel(const uint8_t *__restrict__ tb, const hg::tok3_job *__restrict__ jobs, uint32_t njobs,
                       const uint32_t *__restrict__ tab, uint8_t *out, uint32_t *rec, uint32_t *names, int32_t *status, int only_redo) {
    __shared__ WaveLds S;
    const int lane = threadIdx.x;
    for (uint32_t j = blockIdx.x; j < njobs; j += gridDim.x) {
        if (only_redo && status[j] != 1) continue;                  // (after the position-major kernel: the jobs it handed back)
        const hg::tok3_job J = jobs[j];
        const uint8_t *B = tb + J.tb_base;
        uint8_t *o = out + J.out_off;
        uint32_t *R = rec + J.rec_off * 4ull;                      // (off, len | numeric << 31, val) per token; the job's region is sized for the position-major kernel's 4-word records
        uint32_t *noff = names + J.name_off, *first = noff + (J.nn + 1u), *ntok = first + (J.nn + 1u);
        int err = (J.nn && J.ntp < 1) ? 1 : 0;
        // stream table -> LDS; positions beyond ntp are empty
        for (uint32_t i = (uint32_t)lane; i < NTYPES * MAX_TOK; i += 64) {
            const uint32_t ty = i / MAX_TOK, tp = i % MAX_TOK;
            const bool have = tp < J.ntp;
            S.off[ty][tp] = have ? tab[J.tab_off + (tp * 16u + ty) * 2u] : 0u;
            S.len[ty][tp] = have ? tab[J.tab_off + (tp * 16u + ty) * 2u + 1u] : 0u;
            S.cur[ty][tp] = 0;
        }
        wave_sync();
        uint32_t opos = 0, nrec = 0;
        if (lane == 0) noff[0] = 0;
        for (uint32_t n = 0; n < J.nn && !err; n++) {
            // ---- position 0: DUP / DIFF and the distance ------------------------------------------
            uint32_t ty0 = 255, dist = 0;
            {
                const uint32_t c = S.cur[T_TYPE][0], l = S.len[T_TYPE][0];
                if (c < (l & 0x7fffffffu)) ty0 = (l >> 31) ? (c == 0 ? S.off[T_TYPE][0] : (uint32_t)T_MATCH) : B[S.off[T_TYPE][0] + c];
                if (ty0 != T_DUP && ty0 != T_DIFF) { err = 1; break; }
                const uint32_t dc = S.cur[ty0][0];
                if (dc + 4u > S.len[ty0][0]) { err = 1; break; }
                const uint8_t *p = B + S.off[ty0][0] + dc;
                dist = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                wave_sync();
                if (lane == 0) { S.cur[T_TYPE][0] = c + 1u; S.cur[ty0][0] = dc + 4u; }
                wave_sync();
            }
            if (dist > n) { err = 1; break; }
            const uint32_t m = n - dist;
            if (ty0 == T_DUP) {
                if (m == n) { err = 1; break; }
                const uint32_t s = noff[m], len = noff[m + 1] - s;
                if (opos + len > J.ulen) { err = 1; break; }
                for (uint32_t k = (uint32_t)lane; k < len; k += 64) o[opos + k] = o[s + k];
                opos += len;
                if (lane == 0) { first[n] = first[m]; ntok[n] = ntok[m]; noff[n + 1] = opos; }
                wave_sync();
                continue;
            }
            const uint32_t pfirst = m != n ? first[m] : 0u, pntok = m != n ? ntok[m] : 0u;
            const uint32_t rec0 = nrec;
            bool ended = false;
            for (uint32_t pass = 0; pass < 2 && !ended && !err; pass++) {
                const uint32_t tp = 1u + 64u * pass + (uint32_t)lane;
                // peek my TYPE byte
                uint32_t ty = 255;
                const uint32_t tc = tp < MAX_TOK ? S.cur[T_TYPE][tp] : 0u, tl = tp < MAX_TOK ? S.len[T_TYPE][tp] : 0u;
                if (tp < MAX_TOK && tc < (tl & 0x7fffffffu))
                    ty = (tl >> 31) ? (tc == 0 ? S.off[T_TYPE][tp] : (uint32_t)T_MATCH) : B[S.off[T_TYPE][tp] + tc];
                const unsigned long long stop = __ballot(ty == T_END || ty > T_END);       // END, or nothing left to read
                const uint32_t E = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                if (stop && rl(ty, E) != T_END) { err = 1; break; }                        // ran out of TYPE bytes before END
                ended = stop != 0;
                const bool act = (uint32_t)lane <= E && (uint32_t)lane < 64u;
                // ---- fetch my token ----------------------------------------------------------------
                uint32_t len = 0, val = 0, numeric = 0, width = 0, src = 0;                 // src: 1 stream text, 2 previous output, 3 number, 4 char
                uint32_t src_off = 0, chr = 0;
                int bad = 0;
                if (act) {
                    S.cur[T_TYPE][tp] = tc + 1u;
                    const uint32_t pi = tp - 1u;                                            // token index inside a name
                    uint32_t P0 = 0, P1 = 0, P2 = 0; bool haveP = false;
                    if ((ty == T_DELTA || ty == T_DELTA0 || ty == T_MATCH)) {
                        if (pi < pntok) { const uint32_t *q = R + (size_t)(pfirst + pi) * 3u; P0 = q[0]; P1 = q[1]; P2 = q[2]; haveP = true; }
                        else bad = 1;
                    }
                    if (ty == T_TYPE || ty == T_DZLEN || ty == T_DUP || ty == T_DIFF) bad = 1;    // not token types
                    else if (ty == T_STRING) {
                        const uint32_t c = S.cur[T_STRING][tp], l = S.len[T_STRING][tp];
                        const uint8_t *p = B + S.off[T_STRING][tp];
                        uint32_t e = c;
                        while (e < l && p[e]) e++;
                        if (e >= l) bad = 1;
                        else { len = e - c; src = 1; src_off = S.off[T_STRING][tp] + c; S.cur[T_STRING][tp] = e + 1u; }
                    } else if (ty == T_CHAR) {
                        const uint32_t c = S.cur[T_CHAR][tp];
                        if (c >= S.len[T_CHAR][tp]) bad = 1;
                        else { chr = B[S.off[T_CHAR][tp] + c]; len = 1; src = 4; S.cur[T_CHAR][tp] = c + 1u; }
                    } else if (ty == T_DIGITS || ty == T_DIGITS0) {
                        const uint32_t c = S.cur[ty][tp];
                        if (c + 4u > S.len[ty][tp]) bad = 1;
                        else {
                            const uint8_t *p = B + S.off[ty][tp] + c;
                            val = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                            S.cur[ty][tp] = c + 4u; numeric = 1; src = 3;
                            if (ty == T_DIGITS0) {
                                const uint32_t z = S.cur[T_DZLEN][tp];
                                if (z >= S.len[T_DZLEN][tp]) bad = 1;
                                else { width = B[S.off[T_DZLEN][tp] + z]; S.cur[T_DZLEN][tp] = z + 1u; }
                            }
                        }
                    } else if (ty == T_DELTA || ty == T_DELTA0) {
                        const uint32_t c = S.cur[ty][tp];
                        if (!haveP || c >= S.len[ty][tp]) bad = 1;
                        else {
                            val = P2 + B[S.off[ty][tp] + c]; S.cur[ty][tp] = c + 1u; numeric = 1; src = 3;
                            if (ty == T_DELTA0) width = P1 & 0x7fffffffu;
                        }
                    } else if (ty == T_MATCH) {
                        if (haveP) { len = P1 & 0x7fffffffu; numeric = P1 >> 31; val = P2; src = 2; src_off = P0; }
                    }
                    if (src == 3) { if (width > 15u) width = 15u; const uint32_t nd = ndigits(val); len = nd > width ? nd : width; }
                }
                if (__any(bad)) { err = 1; break; }
                // ---- place and write -----------------------------------------------------------------
                const uint32_t incl = wave_incl_scan_dpp(act ? len : 0u);
                const uint32_t total = rl(incl, 63), at = opos + incl - (act ? len : 0u);
                if ((unsigned long long)opos + total > J.ulen) { err = 1; break; }
                if (act) {
                    uint8_t *w = o + at;
                    if (src == 1) { const uint8_t *p = B + src_off; for (uint32_t k = 0; k < len; k++) w[k] = p[k]; }
                    else if (src == 2) { const uint8_t *p = o + src_off; for (uint32_t k = 0; k < len; k++) w[k] = p[k]; }
                    else if (src == 4) w[0] = (uint8_t)chr;
                    else if (src == 3) { uint32_t v = val; for (uint32_t k = len; k-- > 0;) { w[k] = (uint8_t)('0' + v % 10u); v /= 10u; } }
                    uint32_t *q = R + (size_t)(nrec + (uint32_t)lane) * 3u;
                    q[0] = at; q[1] = len | (numeric << 31); q[2] = val;
                }
                const uint32_t nact = E < 64u ? E + 1u : 64u;
                if (nrec + nact > J.rec_cap) { err = 1; break; }
                nrec += nact;
                opos += total;
                wave_sync();
            }
            if (err) break;
            if (!ended) { err = 1; break; }
            if (opos + 1u > J.ulen) { err = 1; break; }
            if (lane == 0) { o[opos] = 0; first[n] = rec0; ntok[n] = nrec - rec0; noff[n + 1] = opos + 1u; }
            opos += 1u;
            wave_sync();
        }
        if (!err && opos != J.ulen) err = 1;
        status[j] = err ? -1 : 0;                                   // every lane stores the same word
        wave_sync();
    }
}


// ================================================================================================
// The same reconstruction, POSITION-MAJOR (round 6): one workgroup of 16 wavefronts per block, lanes over the NAMES.
//
// The serial form above walks the names in order because name n refers to name n - d; 10 000 names x ~3 us were the long pole of every CRAM 3.1 slice
// decode (27-33 ms per 256 slices beside 15-19 ms of entropy decoding).  Nothing in the format needs that order:
//   * which stream entry a name reads is a COUNT: at token position p the k-th name still alive reads TYPE[p][k], and the k-th name of type t reads the k-th
//     entry of stream (p, t) -- exclusive prefix sums over the names (the k-th STRING starts behind the k-th NUL: a prefix count over the stream's bytes);
//   * what DELTA / MATCH take from the earlier name is the same position's token of that name: per position a forest over the names with edges n -> n - d,
//     and along a path the tokens compose -- (value, length) -> (value + D, length rule) with three length rules (MATCH keeps, DELTA takes the digit count,
//     DELTA0 pads to the earlier length).  Pointer jumping collapses every path to its root (a literal token) in log2(depth) rounds; a counter field that
//     ticks through the whole block is a path of 10 000 names and takes 14 rounds;
//   * where a token lands is the name's start + the lengths of the tokens before it: a running per-name sum over the positions, and one prefix sum over the
//     finished name lengths; DUP names copy their (resolved) source afterwards.
// Per position: two prefix-sum sweeps, the token fetch, the jumping rounds, the record sweep -- every one a coalesced pass over per-name arrays that stay in L2.
// One thing is NOT reproduced: 32-bit wrap-around INSIDE a DELTA0 path changes the digit counts the serial walk would see on the way; such a block (and one with
// 2^28 names) reports status 1 and the serial kernel decodes it (launch_tok3_names runs it over the status-1 jobs).  Error verdicts are those of the serial walk.
// ================================================================================================
constexpr int PT = 1024, PW = PT / 64;
enum { K_NONE = 0, K_STR = 1, K_CHR = 2, K_NUM = 3 };
enum { M_ID = 0, M_NUM = 1, M_PAD = 2 };
constexpr uint32_t NOPE = 0xffffffffu, ROOTP = 0x3fffffffu;


struct ParLds { uint32_t wsum[PW][8]; uint32_t soff[16], slen[16]; };

template <int K>
__device__ __forceinline__ void block_scan(uint32_t (&v)[K], uint32_t (&tot)[K], ParLds &L, int lane, int wave) {
    uint32_t incl[K];
#pragma unroll
    for (int k = 0; k < K; k++) incl[k] = wave_incl_scan_dpp(v[k]);
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < K; k++) L.wsum[wave][k] = incl[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint32_t base = 0, t = 0;
#pragma unroll
        for (int w = 0; w < PW; w++) { const uint32_t x = L.wsum[w][k]; if (w < wave) base += x; t += x; }
        tot[k] = t; v[k] = base + incl[k] - v[k];
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t rd32u(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

__global__ __launch_bounds__(PT)
void tok3_names_par_kernel(const uint8_t *__restrict__ tb, const hg::tok3_job *__restrict__ jobs, uint32_t njobs,
                           const uint32_t *__restrict__ tab, uint8_t *out, uint32_t *rec, uint32_t *names, int32_t *status) {
    __shared__ ParLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t j = blockIdx.x; j < njobs; j += gridDim.x) {
        const hg::tok3_job J = jobs[j];
        const uint8_t *B = tb + J.tb_base;
        uint8_t *o = out + J.out_off;
        uint32_t *R = rec + J.rec_off * 4ull;                       // (name, offset inside the name, length | kind << 28, value / stream offset / character) per token
        const uint32_t nn = J.nn;
        const size_t N2 = (size_t)nn + 2u;
        uint32_t *W = names + J.name_off;
        uint32_t *noff = W, *mraw = W + N2, *eff = W + 2 * N2, *pm = W + 3 * N2, *state = W + 4 * N2, *acc = W + 5 * N2, *tval = W + 6 * N2, *tlk = W + 7 * N2,
                 *tsrc = W + 8 * N2, *rank = W + 9 * N2, *strs = W + 10 * N2;
        unsigned long long *link = (unsigned long long *)(W + 12 * N2);     // par | mode << 30 in the low word, D in the high one
        // (verdicts are gathered in registers and agreed on at barriers -- __syncthreads_or -- so that every thread takes every exit together)
        int myerr = (nn && J.ntp < 1) ? 1 : 0, myredo = nn >= (1u << 28) ? 1 : 0;
        bool stop = myerr || myredo;
        // ---- position 0: DUP / DIFF and the distance -------------------------------------------------------------
        if (!stop && nn) {
            const uint32_t t0off = tab[J.tab_off + T_TYPE * 2u], t0raw = tab[J.tab_off + T_TYPE * 2u + 1u];
            const uint32_t doff[2] = {tab[J.tab_off + T_DUP * 2u], tab[J.tab_off + T_DIFF * 2u]}, dlen[2] = {tab[J.tab_off + T_DUP * 2u + 1u], tab[J.tab_off + T_DIFF * 2u + 1u]};
            uint32_t carry[2] = {0, 0};
            for (uint32_t base = 0; base < nn; base += PT) {
                const uint32_t n = base + (uint32_t)tid;
                const bool has = n < nn;
                uint32_t ty0 = 255;
                if (has && n < (t0raw & 0x7fffffffu)) ty0 = (t0raw >> 31) ? (n == 0 ? t0off : (uint32_t)T_MATCH) : B[t0off + n];
                bool bad = has && ty0 != T_DUP && ty0 != T_DIFF;
                uint32_t v[2] = {has && ty0 == T_DUP ? 1u : 0u, has && ty0 == T_DIFF ? 1u : 0u}, tot[2];
                block_scan<2>(v, tot, L, lane, wave);
                if (has && !bad) {
                    const int w = ty0 == T_DUP ? 0 : 1;
                    const unsigned long long dc = 4ull * ((unsigned long long)carry[w] + v[w]);
                    if (dc + 4u > dlen[w]) bad = true;
                    else {
                        const uint32_t dist = rd32u(B + doff[w] + dc);
                        if (dist > n || (w == 0 && dist == 0)) bad = true;
                        else { mraw[n] = n - dist; state[n] = w == 0 ? 1u : 0u; acc[n] = 0; eff[n] = w == 0 ? n - dist : n; }
                    }
                }
                carry[0] += tot[0]; carry[1] += tot[1];
                if (bad) myerr = 1;
            }
            stop = __syncthreads_or(myerr) != 0;
            // a DUP name stands for the name its chain of DUPs ends in
            while (!stop) {
                int changed = 0;
                for (uint32_t n = (uint32_t)tid; n < nn; n += PT) {
                    const uint32_t e = eff[n];
                    if (e != n && (state[e] & 1u)) { eff[n] = eff[e]; changed = 1; }
                }
                if (!__syncthreads_or(changed)) break;
            }
            if (!stop) for (uint32_t n = (uint32_t)tid; n < nn; n += PT) pm[n] = mraw[n] == n ? NOPE : eff[mraw[n]];
            __syncthreads();
        }
        // ---- token positions ----------------------------------------------------------------------------------
        uint32_t posbase = 0;
        for (uint32_t tp = 1; !stop && nn && tp < J.ntp && tp < MAX_TOK; tp++) {
            if (tid < 16) { L.soff[tid] = tab[J.tab_off + (tp * 16u + (uint32_t)tid) * 2u]; L.slen[tid] = tab[J.tab_off + (tp * 16u + (uint32_t)tid) * 2u + 1u]; }
            __syncthreads();
            int myany = 0;
            const uint32_t toff = L.soff[T_TYPE], tcount = L.slen[T_TYPE] & 0x7fffffffu; const bool implied = (L.slen[T_TYPE] >> 31) != 0;
            // sweep A: who is alive, its TYPE byte, its entry number in the stream of that type
            uint32_t calive = 0, ccur[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t base = 0; base < nn; base += PT) {
                const uint32_t n = base + (uint32_t)tid;
                const bool has = n < nn;
                const bool alive = has && !(state[n] & 3u);
                uint32_t a[1] = {alive ? 1u : 0u}, at[1];
                block_scan<1>(a, at, L, lane, wave);
                const uint32_t r = calive + a[0];
                uint32_t ty = 255;
                if (alive && r < tcount) ty = implied ? (r == 0 ? toff : (uint32_t)T_MATCH) : B[toff + r];
                if (alive && (ty > T_END || ty == T_TYPE || ty == T_DZLEN || ty == T_DUP || ty == T_DIFF)) myerr = 1;      // out of TYPE bytes before END, or not a token type
                uint32_t v[6] = {alive && ty == T_STRING, alive && ty == T_CHAR, alive && ty == T_DIGITS, alive && ty == T_DIGITS0, alive && ty == T_DELTA, alive && ty == T_DELTA0}, tot[6];
                block_scan<6>(v, tot, L, lane, wave);
                if (has) {
                    rank[n] = alive ? r : NOPE;
                    if (alive) {
                        tlk[n] = ty;
                        tval[n] = ty == T_STRING ? ccur[0] + v[0] : ty == T_CHAR ? ccur[1] + v[1] : ty == T_DIGITS ? ccur[2] + v[2] : ty == T_DIGITS0 ? ccur[3] + v[3]
                                : ty == T_DELTA ? ccur[4] + v[4] : ty == T_DELTA0 ? ccur[5] + v[5] : 0u;
                    }
                }
                calive += at[0];
#pragma unroll
                for (int k = 0; k < 6; k++) ccur[k] += tot[k];
            }
            if (__syncthreads_or(myerr)) { stop = true; break; }
            if (calive == 0) break;                                  // every name has had its END
            // the k-th STRING of this position starts behind the k-th NUL of the stream
            const uint32_t nstr = ccur[0];
            if (nstr) {
                const uint32_t so = L.soff[T_STRING], sl = L.slen[T_STRING];
                uint32_t cn = 0;
                if (tid == 0) strs[0] = 0;
                for (uint32_t base = 0; base < sl && cn <= nstr; base += PT * 16u) {
                    const uint32_t i0 = base + (uint32_t)tid * 16u;
                    uint32_t mask = 0;
                    for (uint32_t k = 0; k < 16u && i0 + k < sl; k++) if (B[so + i0 + k] == 0) mask |= 1u << k;
                    uint32_t c[1] = {(uint32_t)__popc(mask)}, ct[1];
                    block_scan<1>(c, ct, L, lane, wave);
                    uint32_t ord = cn + c[0];
                    while (mask) { const uint32_t k = (uint32_t)__ffs((int)mask) - 1u; mask &= mask - 1u; if (ord + 1u <= nstr) strs[ord + 1u] = i0 + k + 1u; ord++; }
                    cn += ct[0];
                }
                if (cn < nstr) myerr = 1;                             // a STRING without its terminator
                if (__syncthreads_or(myerr)) { stop = true; break; }
            }
            // sweep B: fetch the tokens; literal ones are final, DELTA / MATCH note their edge
            for (uint32_t n = (uint32_t)tid; n < nn; n += PT) {
                if (rank[n] == NOPE) continue;
                const uint32_t ty = tlk[n], k = tval[n];
                uint32_t len = 0, kind = K_NONE, val = 0, src = 0, par = ROOTP, mode = M_ID, D = 0;
                bool bad = false;
                if (ty == T_STRING) { const uint32_t a = strs[k], e = strs[k + 1u] - 1u; len = e - a; kind = K_STR; src = L.soff[T_STRING] + a; }
                else if (ty == T_CHAR) { if (k >= L.slen[T_CHAR]) bad = true; else { src = B[L.soff[T_CHAR] + k]; len = 1; kind = K_CHR; } }
                else if (ty == T_DIGITS || ty == T_DIGITS0) {
                    if (4ull * k + 4u > L.slen[ty]) bad = true;
                    else {
                        val = rd32u(B + L.soff[ty] + 4ull * k); kind = K_NUM;
                        uint32_t width = 0;
                        if (ty == T_DIGITS0) { if (k >= L.slen[T_DZLEN]) bad = true; else width = B[L.soff[T_DZLEN] + k]; }
                        if (width > 15u) width = 15u;
                        const uint32_t nd = ndigits(val); len = nd > width ? nd : width;
                    }
                } else if (ty == T_DELTA || ty == T_DELTA0 || ty == T_MATCH) {
                    const uint32_t p = pm[n];
                    if (p == NOPE || rank[p] == NOPE) bad = true;    // no earlier name, or it has no token at this position
                    else if (ty == T_MATCH) par = p;
                    else if (k >= L.slen[ty]) bad = true;
                    else { par = p; D = B[L.soff[ty] + k]; mode = ty == T_DELTA ? M_NUM : M_PAD; }
                } else if (ty == T_END) state[n] |= 2u;
                if (bad) { myerr = 1; continue; }
                if (par != ROOTP) myany = 1;
                else { tval[n] = val; tlk[n] = len | (kind << 28); tsrc[n] = src; }
                link[n] = (unsigned long long)(par | (mode << 30)) | ((unsigned long long)D << 32);
            }
            if (__syncthreads_or(myerr)) { stop = true; break; }
            const bool any = __syncthreads_or(myany) != 0;
            // pointer jumping: every edge ends at a literal token, carrying the composition of the steps on its path
            while (any) {
                int changed = 0;
                for (uint32_t base = 0; base < nn; base += 4u * PT) {          // four independent edges per thread in flight: the loads are a dependent pair each
                    unsigned long long me[4], up[4]; uint32_t idx[4]; bool go[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        idx[u] = base + (uint32_t)u * PT + (uint32_t)tid;
                        go[u] = idx[u] < nn && rank[idx[u]] != NOPE;
                        me[u] = go[u] ? link[idx[u]] : (unsigned long long)ROOTP;
                        go[u] = go[u] && ((uint32_t)me[u] & ROOTP) != ROOTP;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) up[u] = go[u] ? link[(uint32_t)me[u] & ROOTP] : (unsigned long long)ROOTP;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t gpar = (uint32_t)up[u] & ROOTP;
                        if (!go[u] || gpar == ROOTP) continue;       // my parent is a literal: done
                        const uint32_t m1 = ((uint32_t)me[u] >> 30) & 3u, m0 = ((uint32_t)up[u] >> 30) & 3u;       // m0 applies first
                        const unsigned long long Ds = (me[u] >> 32) + (up[u] >> 32);
                        if (Ds > 0xffffffffull) myredo = 1;
                        const uint32_t mo = m1 == M_ID ? m0 : m1 == M_NUM ? (uint32_t)M_NUM : (m0 == M_NUM ? (uint32_t)M_NUM : (uint32_t)M_PAD);
                        link[idx[u]] = (unsigned long long)(gpar | (mo << 30)) | ((Ds & 0xffffffffull) << 32);
                        changed = 1;
                    }
                }
                if (!__syncthreads_or(changed)) break;
            }
            // sweep C: resolve, record, advance the names
            for (uint32_t n = (uint32_t)tid; n < nn; n += PT) {
                const uint32_t r = rank[n];
                if (r == NOPE) continue;
                const unsigned long long me = link[n];
                const uint32_t par = (uint32_t)me & ROOTP;
                uint32_t val, lk, src;
                if (par == ROOTP) { val = tval[n]; lk = tlk[n]; src = tsrc[n]; }
                else {
                    const uint32_t mode = ((uint32_t)me >> 30) & 3u;
                    const uint32_t rv = tval[par], rl = tlk[par];
                    const unsigned long long sum = (unsigned long long)rv + (me >> 32);
                    if (sum > 0xffffffffull) myredo = 1;
                    val = (uint32_t)sum;
                    if (mode == M_ID) { lk = rl; src = tsrc[par]; }
                    else {
                        const uint32_t nd = ndigits(val);
                        uint32_t width = mode == M_PAD ? (rl & 0x0fffffffu) : 0u;
                        if (width > 15u) width = 15u;
                        lk = (nd > width ? nd : width) | ((uint32_t)K_NUM << 28); src = 0;
                    }
                }
                const uint32_t a = acc[n], len = lk & 0x0fffffffu;
                uint32_t *q = R + ((size_t)posbase + r) * 4u;
                q[0] = n; q[1] = a; q[2] = lk; q[3] = (lk >> 28) == K_NUM ? val : src;
                if ((unsigned long long)a + len > J.ulen) myerr = 1; else acc[n] = a + len;
            }
            if (__syncthreads_or(myerr)) { stop = true; break; }
            posbase += calive;
        }
        // ---- name lengths -> offsets; tokens -> bytes ------------------------------------------------------------
        const bool redo = __syncthreads_or(myredo) != 0;
        if (!stop && !redo && nn) {
            unsigned long long total = 0;
            for (uint32_t base = 0; base < nn; base += PT) {
                const uint32_t n = base + (uint32_t)tid;
                const bool has = n < nn;
                uint32_t len = 0;
                if (has) {
                    const uint32_t st = state[n];
                    if (!(st & 1u) && !(st & 2u)) myerr = 1;            // a name without its END
                    len = acc[(st & 1u) ? eff[n] : n] + 1u;
                }
                uint32_t v[1] = {len}, t[1];
                block_scan<1>(v, t, L, lane, wave);
                if (has) noff[n] = (uint32_t)(total + v[0]);
                total += t[0];
            }
            if (total != J.ulen) myerr = 1;
            if (!__syncthreads_or(myerr)) {
                for (uint32_t t = (uint32_t)tid; t < posbase; t += PT) {
                    const uint32_t *q = R + (size_t)t * 4u;
                    const uint32_t lk = q[2], len = lk & 0x0fffffffu, kind = lk >> 28, x = q[3];
                    uint8_t *w = o + noff[q[0]] + q[1];
                    if (kind == K_STR) { const uint8_t *p = B + x; for (uint32_t k = 0; k < len; k++) w[k] = p[k]; }
                    else if (kind == K_CHR) w[0] = (uint8_t)x;
                    else if (kind == K_NUM) { uint32_t v = x; for (uint32_t k = len; k-- > 0;) { w[k] = (uint8_t)('0' + v % 10u); v /= 10u; } }
                }
                for (uint32_t n = (uint32_t)tid; n < nn; n += PT) if (!(state[n] & 1u)) o[noff[n] + acc[n]] = 0;
                __syncthreads();
                for (uint32_t n = (uint32_t)tid; n < nn; n += PT)
                    if (state[n] & 1u) { const uint32_t e = eff[n], len = acc[e] + 1u; const uint8_t *p = o + noff[e]; uint8_t *w = o + noff[n]; for (uint32_t k = 0; k < len; k++) w[k] = p[k]; }
            }
        } else if (!stop && !redo && !nn && J.ulen != 0) myerr = 1;
        const int err = __syncthreads_or(myerr);
        if (tid == 0) status[j] = err ? -1 : redo ? 1 : 0;
        __syncthreads();
    }
}

}  // namespace hgt

namespace hg {
int launch_tok3_names(hg_ctx *ctx, const void *d_tb, const tok3_job *d_jobs, size_t njobs, const uint32_t *d_tab, void *d_out,
                      uint32_t *d_rec, uint32_t *d_names, int32_t *d_status, hipStream_t s) {
    if (!njobs) return HG_OK;
    static const bool par = [] { const char *e = getenv("HG_TOK3_PAR"); return !(e && e[0] == '0'); }();      // 0: the serial walk for every block (A/B)
    size_t wgs = njobs;
    if (par) {
        const size_t maxp = (size_t)ctx->cus * 4;
        hipLaunchKernelGGL(hgt::tok3_names_par_kernel, dim3((unsigned)(wgs > maxp ? maxp : wgs)), dim3(hgt::PT), 0, s, (const uint8_t *)d_tb, d_jobs, (uint32_t)njobs,
                           d_tab, (uint8_t *)d_out, d_rec, d_names, d_status);
        if (hipGetLastError() != hipSuccess) return HG_ELAUNCH;
    }
    const size_t maxw = (size_t)ctx->cus * 8;
    if (wgs > maxw) wgs = maxw;
    hipLaunchKernelGGL(hgt::tok3_names_kernel, dim3((unsigned)wgs), dim3(64), 0, s, (const uint8_t *)d_tb, d_jobs, (uint32_t)njobs,
                       d_tab, (uint8_t *)d_out, d_rec, d_names, d_status, par ? 1 : 0);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

// ================================================================================================
// Encoder side: tokenise every name against its predecessor and append to the (position, type) byte
// streams -- the choices of oracle/tok3_oracle.c orc_tok3_encode(), so that the assembled block is
// byte-identical to the oracle's.  Sixteen wavefronts per block, each taking a contiguous share of the names (a name
// only needs its predecessor, which the wave re-tokenises for itself); two sweeps: the first only counts stream
// sizes per wave, a layout kernel turns them into per-wave write offsets, the second writes.  Lanes = the characters
// of the current name (64 per step): token boundaries come from one ballot of the "class changes here"
// flags; the lane sitting on a token's first character owns that token (length, value, comparison with the
// previous name's token, append to the streams of its position -- no two tokens of a name share a position).
// ================================================================================================
namespace hgt {

constexpr uint32_t NAME_LDS = 256;                                  // a name up to this long is tokenised out of LDS (longer ones out of global memory)
struct EncLds {
    uint32_t base[NTYPES][MAX_TOK], cur[NTYPES][MAX_TOK];
    uint32_t toff[2][MAX_TOK], tlen[2][MAX_TOK], tval[2][MAX_TOK];   // toff: offset of the token INSIDE its name
    uint8_t tcls[2][MAX_TOK];
    uint8_t nm[2][NAME_LDS];                                          // the characters of the current and of the previous tokenised name
};

__device__ __forceinline__ int char_class(uint32_t c) {          // 0 digit, 1 letter, 2 anything else
    return (c - '0') < 10u ? 0 : (((c | 0x20u) - 'a') < 26u ? 1 : 2);
}
__device__ __forceinline__ void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

// Tokenises the names that START in [begin, end) of one block and appends them to the streams (sizes only unless wr).
// The name before `begin` (if any) is tokenised first, without emitting, so that the first name of the range sees its
// predecessor exactly as a single sweep over the whole block would.  Returns the number of names; maxpos by reference.
__device__ uint32_t tok_sweep(const uint8_t *__restrict__ src, uint32_t n, uint32_t begin, uint32_t end, bool wr, uint8_t *B, EncLds &S,
                              int lane, uint32_t &maxpos_out) {
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t pos = begin, ppos = 0, plen = 0, pn = 0, rb = 0, nn = 0, maxpos = 0;
    bool have_prev = false;
    // The tokeniser reads a name many times, a byte per lane and step (digit runs, letter runs, the comparison with the previous name's token): out of global
    // memory every such read is a round trip on some lane's chain, and in the WRITE sweep each of them also waits for the byte stores issued before it (vmcnt
    // counts loads and stores together) -- 30 us per name.  A name is therefore copied into LDS once (S.nm[rb]; the previous tokenised name stays in S.nm[rb ^ 1])
    // and everything below reads the copy: C(i) = character i of the current name, P(i) = of the previous tokenised one.
    uint32_t cabs = 0, pabs = 0;                                      // where the two names start in src (the fallback for names longer than NAME_LDS)
    bool clds = false, plds = false;
    auto stage = [&](uint32_t p0, uint32_t len) {
        cabs = p0; clds = len <= NAME_LDS;
        if (clds) for (uint32_t k = (uint32_t)lane; k < len; k += 64) S.nm[rb][k] = src[p0 + k];
        wave_sync();
    };
    auto C = [&](uint32_t i) -> uint32_t { return clds ? (uint32_t)S.nm[rb][i] : (uint32_t)src[cabs + i]; };
    auto P = [&](uint32_t i) -> uint32_t { return plds ? (uint32_t)S.nm[rb ^ 1u][i] : (uint32_t)src[pabs + i]; };
    auto put8 = [&](uint32_t ty, uint32_t tp, uint32_t v) {
        const uint32_t c = S.cur[ty][tp];
        if (wr) B[S.base[ty][tp] + c] = (uint8_t)v;
        S.cur[ty][tp] = c + 1u;
    };
    auto put32s = [&](uint32_t ty, uint32_t tp, uint32_t v) {
        const uint32_t c = S.cur[ty][tp];
        if (wr) put32(B + S.base[ty][tp] + c, v);
        S.cur[ty][tp] = c + 4u;
    };
    // tokenises the staged name (len characters): emit == false only records the tokens (for the predecessor of the range)
    auto do_name = [&](uint32_t len, bool emit) {
        uint32_t nb = 0, pc = 3, prun = 0;                          // boundaries so far, class / run position of the previous char
        for (uint32_t c0 = 0; c0 < len; c0 += 64) {
            const uint32_t i = c0 + (uint32_t)lane;
            const bool has = i < len;
            const uint32_t ch = has ? C(i) : 0u;
            const int cls = has ? char_class(ch) : 3;
            const uint32_t pcl = (uint32_t)__shfl_up(cls, 1, 64);
            const bool chg = has && (cls == 2 || (uint32_t)cls != (lane == 0 ? pc : pcl));
            const unsigned long long CH = __ballot(chg);
            const unsigned long long upto = CH & (below | (1ull << lane));
            const uint32_t runpos = upto ? (uint32_t)lane - (63u - (uint32_t)__clzll(upto)) : prun + 1u + (uint32_t)lane;
            const bool bnd0 = has && (chg || (cls == 0 && runpos % 9u == 0u));
            const unsigned long long BD0 = __ballot(bnd0);
            const uint32_t idx0 = nb + (uint32_t)__popcll(BD0 & (below | (1ull << lane))) - 1u;   // token index of my char
            const bool bnd = bnd0 && idx0 <= (uint32_t)(MAX_TOK - 3);       // later boundaries fold into the last token
            if (bnd) {
                const uint32_t t = idx0, tp = t + 1u;
                uint32_t tl = 1, val = 0; uint32_t tc;
                if (t == (uint32_t)(MAX_TOK - 3)) { tc = T_STRING; tl = len - i; }
                else if (cls == 0) {
                    uint32_t e = i;
                    while (e < len && e - i < 9u && (C(e) - '0') < 10u) { val = val * 10u + (C(e) - '0'); e++; }
                    tl = e - i;
                    tc = (ch == '0' && tl > 1u) ? T_DIGITS0 : T_DIGITS;
                } else if (cls == 1) {
                    uint32_t e = i;
                    while (e < len && char_class(C(e)) == 1) e++;
                    tl = e - i; tc = tl == 1u ? T_CHAR : T_STRING;
                } else tc = T_CHAR;
                const uint32_t off = i;
                S.toff[rb][t] = off; S.tlen[rb][t] = tl; S.tval[rb][t] = val; S.tcls[rb][t] = (uint8_t)tc;
                if (emit) {
                    const bool haveP = have_prev && t < pn;
                    const uint32_t po = S.toff[rb ^ 1][t], pl = S.tlen[rb ^ 1][t], pv = S.tval[rb ^ 1][t], pcs = S.tcls[rb ^ 1][t];
                    bool same = haveP && pcs == tc && pl == tl;
                    for (uint32_t k = 0; same && k < tl; k++) same = P(po + k) == C(off + k);
                    if (same) put8(T_TYPE, tp, T_MATCH);
                    else if (haveP && tc == T_DIGITS && pcs == T_DIGITS && val >= pv && val - pv < 256u) { put8(T_TYPE, tp, T_DELTA); put8(T_DELTA, tp, val - pv); }
                    else if (haveP && tc == T_DIGITS0 && pcs == T_DIGITS0 && tl == pl && val >= pv && val - pv < 256u) { put8(T_TYPE, tp, T_DELTA0); put8(T_DELTA0, tp, val - pv); }
                    else {
                        put8(T_TYPE, tp, tc);
                        if (tc == T_STRING) {
                            const uint32_t c = S.cur[T_STRING][tp];
                            if (wr) { uint8_t *w = B + S.base[T_STRING][tp] + c; for (uint32_t k = 0; k < tl; k++) w[k] = (uint8_t)C(off + k); w[tl] = 0; }
                            S.cur[T_STRING][tp] = c + tl + 1u;
                        } else if (tc == T_CHAR) put8(T_CHAR, tp, ch);
                        else { put32s(tc, tp, val); if (tc == T_DIGITS0) put8(T_DZLEN, tp, tl); }
                    }
                }
            }
            const uint32_t nbd = (uint32_t)__popcll(BD0);
            nb = nb + nbd > (uint32_t)(MAX_TOK - 2) ? (uint32_t)(MAX_TOK - 2) : nb + nbd;
            pc = (uint32_t)__shfl(cls, 63, 64); prun = (uint32_t)__shfl((int)runpos, 63, 64);
            wave_sync();
        }
        const uint32_t nt = nb;
        if (emit) {
            if (lane == 0) put8(T_TYPE, nt + 1u, T_END);
            if (maxpos < nt + 2u) maxpos = nt + 2u;
        }
        rb ^= 1u; pn = nt; pabs = cabs; plds = clds;                  // the name just tokenised is the "previous" one from here on
        wave_sync();
    };
    if (begin > 0 && begin < end) {                                   // predecessor: the name that ends at begin - 1
        uint32_t q = begin - 1u;                                      // its NUL
        for (;;) {                                                    // walk back to the byte after the previous NUL (or 0)
            if (q == 0) break;
            const uint32_t back = q < 64u ? q : 64u;
            const uint32_t p = q - 1u - (uint32_t)lane;               // lanes look at q-1, q-2, ...
            const unsigned long long z = __ballot((uint32_t)lane < back && src[p] == 0);
            if (z) { q -= (uint32_t)__builtin_ctzll(z); break; }
            q -= back;
        }
        ppos = q; plen = begin - 1u - q; have_prev = true;
        stage(ppos, plen);
        do_name(plen, false);
    }
    while (pos < end) {
        uint32_t len = 0;                                             // distance to the terminating NUL
        for (;;) {
            const uint32_t p = pos + len + (uint32_t)lane;
            const unsigned long long z = __ballot(p >= n || src[p] == 0);
            if (z) { len += (uint32_t)__builtin_ctzll(z); break; }
            len += 64;
        }
        stage(pos, len);
        bool dup = false;
        if (have_prev && len == plen) {                               // (a duplicate of a duplicate equals the last TOKENISED name too: that is the one P() reads)
            dup = true;
            for (uint32_t k = 0; k < len && dup; k += 64) {
                const uint32_t q = k + (uint32_t)lane;
                if (__any(q < len && C(q) != P(q))) dup = false;
            }
        }
        if (dup) {
            if (lane == 0) { put8(T_TYPE, 0, T_DUP); put32s(T_DUP, 0, 1); }
            if (maxpos < 1) maxpos = 1;
            wave_sync();
        } else {
            if (lane == 0) { put8(T_TYPE, 0, T_DIFF); put32s(T_DIFF, 0, have_prev ? 1u : 0u); }
            do_name(len, true);
        }
        ppos = pos; plen = len; have_prev = true; nn++;
        pos += len + 1u;
    }
    maxpos_out = maxpos;
    return nn;
}

constexpr uint32_t TOKW = 16;                                          // wavefronts that share one block
constexpr uint32_t NSTR = NTYPES * MAX_TOK;

// first name start at or after byte offset lo (a name starts at 0 or right after a NUL)
__device__ uint32_t name_start_at(const uint8_t *__restrict__ src, uint32_t n, uint32_t lo, int lane) {
    if (lo == 0) return 0;
    uint32_t p = lo;
    for (;;) {
        if (p >= n) return n;
        const uint32_t q = p + (uint32_t)lane;
        const unsigned long long z = __ballot(q <= n && src[q - 1u] == 0);
        if (z) return p + (uint32_t)__builtin_ctzll(z);
        p += 64;
    }
}

// pass 0 (wr = 0): sizes of every (position, type) stream for this wave's share of the names; pass 1 (wr = 1): write them
__global__ __launch_bounds__(64)
void tok3_sweep_kernel(const uint8_t *__restrict__ in, const hg::tok3_enc_job *__restrict__ jobs, uint32_t njobs, uint8_t *sb,
                       uint32_t *counts, uint32_t *meta, const hg::tok3_enc_res *res, int wr) {
    __shared__ EncLds S;
    const int lane = threadIdx.x;
    for (uint32_t u = blockIdx.x; u < njobs * TOKW; u += gridDim.x) {
        const uint32_t j = u / TOKW, w = u % TOKW;
        if (wr && res[j].total == 0xffffffffu) continue;               // the streams would not fit their buffer (cannot happen: <= 6 bytes per input byte)
        const hg::tok3_enc_job J = jobs[j];
        const uint8_t *src = in + J.in_off;
        const uint32_t n = J.n;
        const uint32_t begin = name_start_at(src, n, (uint32_t)((uint64_t)n * w / TOKW), lane);
        const uint32_t end = w + 1 == TOKW ? n : name_start_at(src, n, (uint32_t)((uint64_t)n * (w + 1) / TOKW), lane);
        uint32_t *C = counts + ((size_t)j * TOKW + w) * NSTR;
        for (uint32_t i = (uint32_t)lane; i < NSTR; i += 64) { (&S.base[0][0])[i] = wr ? C[i] : 0u; (&S.cur[0][0])[i] = 0; }
        wave_sync();
        uint32_t maxpos = 0;
        const uint32_t nn = tok_sweep(src, n, begin, end, wr != 0, sb + J.sb_off, S, lane, maxpos);
        if (!wr) {
            for (uint32_t i = (uint32_t)lane; i < NSTR; i += 64) C[i] = (&S.cur[0][0])[i];
            if (lane == 0) { meta[((size_t)j * TOKW + w) * 2] = nn; meta[((size_t)j * TOKW + w) * 2 + 1] = maxpos; }
        }
        wave_sync();
    }
}

// between the passes: turn the per-wave sizes into per-wave write offsets (in place) and the per-stream totals
__global__ __launch_bounds__(64)
void tok3_layout_kernel(const hg::tok3_enc_job *__restrict__ jobs, uint32_t njobs, uint32_t *counts, const uint32_t *__restrict__ meta,
                        uint32_t *fin, hg::tok3_enc_res *res) {
    const int lane = threadIdx.x;
    for (uint32_t j = blockIdx.x; j < njobs; j += gridDim.x) {
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < NSTR; b0 += 64) {
            const uint32_t i = b0 + (uint32_t)lane;
            uint32_t tot = 0;
            for (uint32_t w = 0; w < TOKW; w++) tot += counts[((size_t)j * TOKW + w) * NSTR + i];
            const uint32_t incl = wave_incl_scan_dpp(tot);
            uint32_t off = carry + incl - tot;
            fin[(size_t)j * NSTR * 2 + i] = off; fin[(size_t)j * NSTR * 2 + NSTR + i] = tot;
            for (uint32_t w = 0; w < TOKW; w++) { uint32_t *c = &counts[((size_t)j * TOKW + w) * NSTR + i]; const uint32_t v = *c; *c = off; off += v; }
            carry += rl(incl, 63);
        }
        uint32_t nn = 0, maxpos = 0;
        for (uint32_t w = 0; w < TOKW; w++) { nn += meta[((size_t)j * TOKW + w) * 2]; const uint32_t m = meta[((size_t)j * TOKW + w) * 2 + 1]; maxpos = m > maxpos ? m : maxpos; }
        hg::tok3_enc_res R;
        R.nn = nn; R.nstreams = 0; R.total = carry > jobs[j].sb_cap ? 0xffffffffu : carry; R.pad = maxpos;
        res[j] = R;
        wave_sync();
    }
}

// after the write pass: emission order, implied TYPE streams, duplicate streams
__global__ __launch_bounds__(64)
void tok3_emit_kernel(const hg::tok3_enc_job *__restrict__ jobs, uint32_t njobs, const uint8_t *__restrict__ sb, const uint32_t *__restrict__ fin,
                      hg::tok3_enc_stream *list, hg::tok3_enc_res *res) {
    __shared__ EncLds S;
    const int lane = threadIdx.x;
    for (uint32_t j = blockIdx.x; j < njobs; j += gridDim.x) {
        const uint8_t *B = sb + jobs[j].sb_off;
        for (uint32_t i = (uint32_t)lane; i < NSTR; i += 64) { (&S.base[0][0])[i] = fin[(size_t)j * NSTR * 2 + i]; (&S.cur[0][0])[i] = fin[(size_t)j * NSTR * 2 + NSTR + i]; }
        wave_sync();
        hg::tok3_enc_res R = res[j];
        const uint32_t nn = R.nn, maxpos = R.pad, total = R.total;
        uint32_t nem = 0;
        hg::tok3_enc_stream *L = list + (size_t)j * NSTR;
        if (total != 0xffffffffu) {
            for (uint32_t t = 0; t < maxpos; t++) {
                int implied = -1;
                const uint32_t tyn = S.cur[T_TYPE][t];
                if (t > 0 && tyn == nn && tyn) {
                    const uint8_t *ty = B + S.base[T_TYPE][t];
                    const uint32_t f = ty[0];
                    if (f != T_TYPE && f != T_MATCH && f <= T_DELTA0 && S.cur[f][t]) {
                        bool all = true;
                        for (uint32_t k = 1; k < nn && all; k += 64) { const uint32_t q = k + (uint32_t)lane; if (__any(q < nn && ty[q] != T_MATCH)) all = false; }
                        if (all) implied = (int)f;
                    }
                }
                bool firsts = true;
                for (int pass = 0; pass < 2; pass++)
                for (uint32_t ty = 0; ty <= T_END; ty++) {
                    if (implied >= 0) { if (ty == T_TYPE) continue; if ((pass == 0) != ((int)ty == implied)) continue; }
                    else if (pass) continue;
                    const uint32_t sl = S.cur[ty][t], so = S.base[ty][t];
                    if (!sl) continue;
                    uint32_t dupk = 0xffffffffu;
                    for (uint32_t k = 0; k < nem && dupk == 0xffffffffu; k++) {
                        if (L[k].len != sl) continue;
                        bool eq = true;
                        for (uint32_t q0 = 0; q0 < sl && eq; q0 += 64) { const uint32_t q = q0 + (uint32_t)lane; if (__any(q < sl && B[L[k].off + q] != B[so + q])) eq = false; }
                        if (eq) dupk = k;
                    }
                    hg::tok3_enc_stream E;
                    E.off = so; E.len = sl; E.pos = (uint8_t)t; E.type = (uint8_t)ty;
                    E.ttype = (uint8_t)(ty | (firsts ? 0x80u : 0u) | (dupk != 0xffffffffu ? 0x40u : 0u));
                    E.dup_pos = dupk != 0xffffffffu ? L[dupk].pos : 0; E.dup_type = dupk != 0xffffffffu ? L[dupk].type : 0;
                    E.pad[0] = E.pad[1] = E.pad[2] = 0;
                    firsts = false;
                    L[nem] = E;                                     // every lane stores the same record
                    wave_sync();
                    nem++;
                }
            }
        }
        R.nstreams = nem; R.pad = 0;
        res[j] = R;
        wave_sync();
    }
}

}  // namespace hgt

namespace hg {
int launch_tok3_tokenise(hg_ctx *ctx, const void *d_in, const tok3_enc_job *d_jobs, size_t njobs, void *d_sb, tok3_enc_stream *d_list,
                         tok3_enc_res *d_res, hipStream_t s) {
    if (!njobs) return HG_OK;
    // per job: TOKW x 1664 stream sizes / offsets, (names, positions) per wave, 2 x 1664 final (offset, size) rows
    const size_t words = njobs * ((size_t)hgt::TOKW * hgt::NSTR + hgt::TOKW * 2 + 2 * (size_t)hgt::NSTR);
    if (int rc = ensure_scratch(ctx, 12, words * 4 + 64)) return rc;
    uint32_t *d_counts = (uint32_t *)ctx->d_scratch[12], *d_meta = d_counts + njobs * (size_t)hgt::TOKW * hgt::NSTR;
    uint32_t *d_fin = d_meta + njobs * (size_t)hgt::TOKW * 2;
    const size_t maxw = (size_t)ctx->cus * 8;
    size_t wg1 = njobs * hgt::TOKW, wg0 = njobs;
    if (wg1 > maxw) wg1 = maxw;
    if (wg0 > maxw) wg0 = maxw;
    hipLaunchKernelGGL(hgt::tok3_sweep_kernel, dim3((unsigned)wg1), dim3(64), 0, s, (const uint8_t *)d_in, d_jobs, (uint32_t)njobs, (uint8_t *)d_sb, d_counts, d_meta, (const tok3_enc_res *)d_res, 0);
    hipLaunchKernelGGL(hgt::tok3_layout_kernel, dim3((unsigned)wg0), dim3(64), 0, s, d_jobs, (uint32_t)njobs, d_counts, (const uint32_t *)d_meta, d_fin, d_res);
    hipLaunchKernelGGL(hgt::tok3_sweep_kernel, dim3((unsigned)wg1), dim3(64), 0, s, (const uint8_t *)d_in, d_jobs, (uint32_t)njobs, (uint8_t *)d_sb, d_counts, d_meta, (const tok3_enc_res *)d_res, 1);
    hipLaunchKernelGGL(hgt::tok3_emit_kernel, dim3((unsigned)wg0), dim3(64), 0, s, d_jobs, (uint32_t)njobs, (const uint8_t *)d_sb, (const uint32_t *)d_fin, d_list, d_res);
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

// ransnx16.hip -- CRAM 3.1 "rANS Nx16" block decoder for MI355X (gfx950 / CDNA4).
//
// Replaces rans_uncompress_4x16() as called by cram_uncompress_block (reference
// cram/cram_io.c:1697-1714; implementation = htscodecs rANS_static4x16pr.c / 32x16pr, an ABSENT
// submodule).  Format per the hts-specs CRAM-codecs document as restated in
// oracle/ransnx16_oracle.c -- PARITY UNPINNED (no stock-htslib stream exists in the reference to
// check against); the GPU decoder is bit-exact with that oracle.
//
// Mapping: the N (4 or 32) interleaved rANS states of a stream live in N adjacent lanes -- sixteen 4-way streams per wavefront, ONE
// 32-way stream per wavefront (lanes 32..63 idle: two streams side by side ran in lock step, so an order-0 and an order-1 neighbour
// cost the sum of both decode chains); every step each lane decodes one symbol and the lanes whose state drops below 2^15 pull the next
// 16-bit words of the SHARED stream in lane order -- a ballot and a prefix popcount per step, which is exactly what the 32-way SIMD CPU
// decoders emulate with shuffles.  A stream is one chain of n / N steps of dependent LDS reads, and a launch lasts as long as its longest
// chain, so the tables are shaped for few reads per step: order 0 -- a slot -> symbol byte table + the 257-entry cumulative array; order 1
// with a small alphabet (<= 16 contexts of <= 16 symbols) -- a dense form, two reads per symbol (a 256-bucket index per context, then the
// (cumulative, symbol, next-context rank) entry and its neighbour); order 1 otherwise -- sparse per-context lists with a 64-bucket index,
// in LDS when they fit, else in global scratch.  Tables are parsed by the first lane of the group.
// Handled here: flags ORDER, X32, NOSZ (size from the descriptor), CAT.  The PACK / RLE / STRIPE
// transforms are undone by ransnx16_xform.hip after this kernel: the host planner
// (cram_entropy_host.hip) parses their headers and hands this kernel pre-parsed core descriptors
// (desc.reserved bit 31).  Raw streams carrying those flags are reported -3 by THIS kernel, which is
// what the device-resident entry point (hg_ransnx16_decode_dev) returns for them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"
#include <type_traits>
#include "ransnx16_dev.h"

namespace hgn {

constexpr int WAVES = 4;

struct GroupLds { uint16_t C[258]; };
// 32-way streams (the big data series) keep their order-1 tables in LDS when they fit: a symbol lookup is a chain
// of 6-7 DEPENDENT table reads, which from global memory (~600 cycles each) capped a stream at ~23 MB/s.
#ifndef HG_O1_POOL
#define HG_O1_POOL 4352
#endif
constexpr uint32_t O1_LDS_WORDS = HG_O1_POOL;          // per stream; 4 streams (wavefronts) per workgroup -> 68 KiB, two workgroups per CU
// Order 0 uses the same pool as a direct slot -> symbol table (4096 one-byte entries) instead of a binary search.

template <int N>
__global__ __launch_bounds__(WAVES * 64)
void ransnx16_decode_kernel(const uint8_t *__restrict__ in, const hg_stream_desc *__restrict__ desc,
                            const uint32_t *__restrict__ sel, uint32_t nsel, uint8_t *out, int32_t *status,
                            uint32_t *scratch) {
    // 32-way: ONE stream per wavefront (lanes 32..63 idle).  Two streams side by side ran in lock step: an order-0 and an order-1
    // neighbour cost the sum of both decode chains per step and a short stream waited for a long one; with the chain being LDS
    // latency, the idle lanes cost nothing and twice as many wavefronts hide more of it.
    constexpr int GROUPS = N == 32 ? 1 : 64 / N;
    __shared__ GroupLds lds[WAVES * GROUPS];
    // one pool per stream slot, used either as the order-1 table copy or as the order-0 lookup table
    __shared__ uint32_t pool[N == 32 ? WAVES * GROUPS : 1][N == 32 ? O1_LDS_WORDS : 1];
    // 32-way only: the next 128 renormalisation words of the stream (filled 64 at a time from a register prefetch, so a
    // step never waits on global memory) and the dense numbering of the order-1 contexts (for the bucket table below)
    __shared__ uint32_t ring_s[N == 32 ? WAVES * GROUPS : 1][N == 32 ? 64 : 1];
    __shared__ uint8_t rank_s[WAVES * GROUPS][256];            // alphabet list while the tables are parsed, then (32-way) the context ranks
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & (N - 1);
    const bool idle = lane / N >= GROUPS;                          // lanes beyond the groups in use
    const int grp = idle ? 0 : lane / N;
    const uint32_t g_global = (blockIdx.x * WAVES + (tid >> 6)) * GROUPS + grp;
    const uint32_t g_total = gridDim.x * WAVES * GROUPS;
    GroupLds &G = lds[(tid >> 6) * GROUPS + grp];
    const unsigned long long gmask = (N == 64 ? ~0ull : ((1ull << N) - 1ull)) << (grp * N);
    const int lane0 = grp * N;

    for (uint32_t k = g_global; __any(k < nsel); k += g_total) {
        bool have = k < nsel && !idle;
        const uint32_t sidx = have ? sel[k] : 0;
        if (N == 4 && have && big4_takes(in, desc[sidx])) have = false;   // long 4-way streams: rans4x16_big.hip, one per wavefront
        int err = have ? 0 : 2;
        uint32_t flags = 0, usz = 0, shift = 12, np_words = 0xffffffffu;
        const uint8_t *cp = nullptr, *end = nullptr;
        uint8_t *o = nullptr;
        uint32_t *tabs = nullptr;
        if (have) {
            const hg_stream_desc d = desc[sidx];
            cp = in + d.in_off; end = cp + d.in_len;
            o = out + d.out_off; tabs = scratch + d.scratch_off;
            if (d.reserved & 0x80000000u) {                       // header already parsed by the host planner
                flags = d.reserved & (F_ORDER | F_X32 | F_CAT);
                usz = d.out_len;
                if (((flags & F_X32) ? 32 : 4) != N) err = 1;
            } else if (d.in_len < 1) err = 1;
            else {
                flags = *cp++;
                if (flags & F_NOSZ) usz = d.out_len;
                else if (get_u7(cp, end, usz)) err = 1;
                if (!err && usz != d.out_len) err = 1;
                if (!err && (flags & (F_STRIPE | F_RLE | F_PACK))) err = 3;      // not handled yet
                if (!err && ((flags & F_X32) ? 32 : 4) != N) err = 1;
            }
        }
        const bool cat = !err && (flags & F_CAT);
        if (cat) {
            if (cp + usz > end) err = 1;
            else for (uint32_t i = (uint32_t)sub; i < usz; i += N) o[i] = cp[i];
        }
        const uint32_t order = flags & F_ORDER;
        const bool core = !err && !cat && usz != 0;
        // ---- tables (first lane of the group) ---------------------------------------------------
        if (core && sub == 0) {
            if (order == 0) err = parse_o0(cp, end, G.C);
            else err = parse_o1(cp, end, tabs, G.C, rank_s[(tid >> 6) * GROUPS + grp], shift, np_words);
        }
        // broadcast parse results from the first lane of the group
        {
            err = __shfl(err, lane0, 64);
            shift = (uint32_t)__shfl((int)shift, lane0, 64);
            const unsigned long long cpv = (unsigned long long)(uintptr_t)cp;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)cpv, lane0, 64);
            const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(cpv >> 32), lane0, 64);
            cp = (const uint8_t *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        O1Forms Fm;
        Fm.form = O1_LISTS_GLOBAL; Fm.bb = 0; Fm.l_off = 0; Fm.d_off = 0; Fm.drank0 = 0;
        const uint8_t *lut = nullptr;                                // order-0 slot -> symbol
        if constexpr (N == 32) {
            uint32_t *P = pool[(tid >> 6) * GROUPS + grp];
            const uint32_t npw = (uint32_t)__shfl((int)np_words, lane0, 64);
            if (core && !err && order) build_o1_forms<N>(P, O1_LDS_WORDS, tabs, npw, shift, rank_s[(tid >> 6) * GROUPS + grp], sub, gmask, lane0, Fm);
            else if (core && !err && !order) {
                uint8_t *L8 = (uint8_t *)P;
                // lane l fills the slots of symbols l, l+32, ...
                for (uint32_t sy = (uint32_t)sub; sy < 256; sy += N) { const uint32_t a = G.C[sy], b = G.C[sy + 1]; for (uint32_t q = a; q < b; q++) L8[q] = (uint8_t)sy; }
                lut = L8;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const bool live = core && !err;
        uint32_t R = 0;
        if (live) {
            if (cp + 4 * N > end) err = 1;
            else { R = rd32(cp + 4 * sub); cp += 4 * N; }
        }
        // 32-way: renormalisation words come from the LDS ring, topped up from a register prefetch
        uint32_t *ring = ring_s[N == 32 ? (tid >> 6) * GROUPS + grp : 0];
        uint32_t wpos = 0, wfill = 0, wavail = 0, pre = 0;
        const uint8_t *wbase = cp;
        auto load_chunk = [&](uint32_t c) -> uint32_t {               // words 64c + 2*sub, +1 of the stream (0 past the end)
            const uint8_t *p = wbase + 2u * (64u * c + 2u * (uint32_t)sub);
            uint32_t v = 0;
            if (p + 4 <= end) v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            else for (int q = 0; q < 4; q++) if (p + q < end) v |= (uint32_t)p[q] << (8 * q);
            return v;
        };
        if constexpr (N == 32) {
            if (live && !err) {
                wavail = (uint32_t)((end - wbase) >> 1);
                ring[sub] = load_chunk(0);
                pre = load_chunk(1);
                wfill = 64;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t mask = (1u << shift) - 1u;
        const uint32_t per = usz / N;
        uint32_t pos = order == 0 ? (uint32_t)sub : (uint32_t)sub * per, ctx = 0;
        uint32_t rctx = Fm.drank0;                                      // dense form: rank of the current context
        const uint32_t steps = per, rem = usz - per * N;
        const uint32_t max_steps = (live && !err) ? steps + (order ? rem : 0u) : 0u;
        // The decode loop, once per table form: the form is the same for every lane of a 32-way launch's wavefront (one stream per wavefront), so the
        // choice is made ONCE, outside the loop -- inside it, a per-step dispatch on a value the compiler cannot prove uniform became nested exec-mask branches.
        auto run = [&](auto formc) {
            O1Forms Gf = Fm; Gf.form = decltype(formc)::value;
            for (uint32_t it = 0; __any(it < max_steps); it++) {
                const bool act = live && !err && it < max_steps;
                const bool mine = act && (it < steps || (order && sub == N - 1));
                uint32_t need = 0;
                if (mine) {
                    const uint32_t m = R & mask;
                    uint32_t sym = 0, cum = 0, f = 1;
                    if (order == 0) {
                        uint32_t lo = 0, hi = 256;
                        if (lut) lo = lut[m];
                        else while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (G.C[mid] <= m) lo = mid; else hi = mid; }
                        sym = lo; cum = G.C[lo]; f = (uint32_t)G.C[lo + 1] - cum;
                    } else {
                        if (!lookup_o1(Gf, pool[N == 32 ? (tid >> 6) * GROUPS + grp : 0], tabs, ctx, rctx, m, shift, sym, cum, f)) err = 1;
                    }
                    if (!err) {
                        o[pos] = (uint8_t)sym;
                        pos += order == 0 ? (uint32_t)N : 1u;
                        ctx = sym;
                        R = f * (R >> shift) + m - cum;
                        need = R < RANS_L ? 1u : 0u;
                    }
                }
                // the lanes that renormalise take consecutive 16-bit words in lane order
                const unsigned long long b = __ballot(need != 0) & gmask;
                const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                const uint32_t tot = (uint32_t)__popcll(b);
                if constexpr (N == 32) {
                    if (need) {
                        const uint32_t k = wpos + before;
                        if (k >= wavail) err = 1;
                        else R = (R << 16) | (uint32_t)((const uint16_t *)ring)[k & 127u];
                    }
                    wpos += tot;
                    if (act && wfill - wpos < 32u) {                      // whole group takes this branch together
                        ring[((wfill >> 1) + (uint32_t)sub) & 63u] = pre;
                        wfill += 64;
                        pre = load_chunk(wfill >> 6);
                    }
                } else {
                    if (need) {
                        const uint8_t *w = cp + 2u * before;
                        if (w + 2 > end) err = 1;
                        else R = (R << 16) | (uint32_t)w[0] | ((uint32_t)w[1] << 8);
                    }
                    cp += 2u * tot;
                }
                err = (__ballot(err == 1) & gmask) ? 1 : err;
            }
        };
        if constexpr (N == 32) {
            switch (__builtin_amdgcn_readfirstlane((int)Fm.form)) {
            case O1_DENSE: run(std::integral_constant<uint32_t, O1_DENSE>()); break;
            case O1_BUCKET: run(std::integral_constant<uint32_t, O1_BUCKET>()); break;
            case O1_LISTS_LDS: run(std::integral_constant<uint32_t, O1_LISTS_LDS>()); break;
            default: run(std::integral_constant<uint32_t, O1_LISTS_GLOBAL>()); break;
            }
        } else run(std::integral_constant<uint32_t, O1_LISTS_GLOBAL>());       // sixteen 4-way streams per wavefront: no pool, lists in global scratch
        // order-0 tail: states 0..rem-1 give one more symbol each, without update
        if (live && !err && order == 0 && (uint32_t)sub < rem) {
            const uint32_t m = R & mask;
            uint32_t lo = 0, hi = 256;
            if (lut) lo = lut[m];
            else while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (G.C[mid] <= m) lo = mid; else hi = mid; }
            o[per * N + sub] = (uint8_t)lo;
        }
        err = (__ballot(err == 1) & gmask) ? 1 : err;
        if (have && sub == 0) status[sidx] = err == 0 ? 0 : (err == 3 ? HG_BLOCK_EUNSUPPORTED : -1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace hgn

namespace hg {
int launch_ransnx16_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4,
                           size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out, int32_t *d_status,
                           uint32_t *d_scratch, hipStream_t s) {
    const size_t maxw = (size_t)ctx->cus * 8;
    const bool side = n4 != 0 && n32 != 0;                  // both variants present: overlap them
    hipStream_t s2 = side ? fork_side(ctx, s) : s;
    hipStream_t s3 = n4 ? fork_side3(ctx, s) : s;               // (forked before anything of this call is queued on s)
    if (n4) {
        size_t wgs = (n4 + hgn::WAVES * 16 - 1) / (hgn::WAVES * 16);
        if (wgs > maxw) wgs = maxw;
        hipLaunchKernelGGL(hgn::ransnx16_decode_kernel<4>, dim3((unsigned)wgs), dim3(hgn::WAVES * 64), 0, s,
                           (const uint8_t *)d_in, d_desc, d_sel4, (uint32_t)n4, (uint8_t *)d_out, d_status, d_scratch);
        // the long 4-way streams of the same list, one per wavefront (the kernel above skipped them; this one skips the others)
        const int rc = launch_rans4x16_big_decode(ctx, d_in, d_desc, d_sel4, n4, d_out, d_status, d_scratch, s3, n32 != 0);
        join_side3(ctx, s);
        if (rc != HG_OK) return rc;
    }
    if (n32) {
        size_t wgs = (n32 + hgn::WAVES - 1) / hgn::WAVES;                  // one 32-way stream per wavefront
        if (wgs > maxw * 2) wgs = maxw * 2;
        hipLaunchKernelGGL(hgn::ransnx16_decode_kernel<32>, dim3((unsigned)wgs), dim3(hgn::WAVES * 64), 0, s2,
                           (const uint8_t *)d_in, d_desc, d_sel32, (uint32_t)n32, (uint8_t *)d_out, d_status, d_scratch);
        if (side) join_side(ctx, s);
    }
    return hipGetLastError() == hipSuccess ? HG_OK : HG_ELAUNCH;
}
}  // namespace hg

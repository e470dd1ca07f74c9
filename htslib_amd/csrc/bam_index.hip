// bam_index.hip -- BAI index construction from a device-resident BAM stream.
//
// Replaces the read-and-push loop of `samtools index` (reference sam.c:994-1031 sam_index: bam_read1 + hts_idx_push per
// record) and the index finalisation / serialisation (hts.c:2320-2365 insert_to_b / insert_to_l, 2431-2536 update_loff /
// compress_binning / hts_idx_finish, 2558-2640 hts_idx_push, 2759-2822 idx_save_core; hts_reg2bin htslib/hts.h:1516-1523;
// bam_endpos sam.c:673-678).  Parity: oracle/bam_oracle.c orc_bai_build, which is pinned to the .bai files reference
// htslib wrote for its own fixtures.
//
// hts_idx_push is written as a sequential state machine, but everything it does per record is a function of the record
// and its predecessor:
//   * a chunk of the binning index = a maximal run of consecutive records with the same (reference, bin); its ends are
//     the virtual offsets of the first record of the run and of the first record after it      -> flags + prefix sum
//   * linear index slot w of a reference = virtual offset of the FIRST record overlapping window w  -> atomicMin
//   * the meta bin = per-reference first / last offsets and mapped / unmapped counts                -> atomics
//   * "unsorted" / "chromosome blocks not continuous" / "NO_COOR reads not at the end"              -> neighbour tests
// so the per-record work (CIGAR walk for the end position, bin, virtual offset by binary search in the BGZF block table,
// run detection) runs one thread per record; the host only folds the few thousand runs into bins, applies
// compress_binning and writes the file.  Bins of a reference are written in ascending order (htslib: hash-table order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>
#include "htsgpu.h"
#include "hg_device.h"
#include "hg_internal.h"

namespace hgi {

struct Blk { uint64_t coff, uoff; uint32_t ulen, pad; };
struct Run { uint64_t u; int32_t tid; uint32_t bin; };

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
// bgzf_tell at uncompressed offset u: the first block that starts at u if there is one (bgzf_read steps to the next block as
// soon as the current one is used up, bgzf.c:1276-1281), else the block containing u
__host__ __device__ inline uint64_t voff_of(const Blk *blk, uint64_t nb, uint64_t file_size, uint64_t u) {
    uint64_t lo = 0, hi = nb;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (blk[mid].uoff < u) lo = mid + 1; else hi = mid; }
    if (lo < nb && blk[lo].uoff == u) return blk[lo].coff << 16;
    if (lo == 0) return 0;
    if (lo == nb && u >= blk[nb - 1].uoff + blk[nb - 1].ulen) return file_size << 16;
    return (blk[lo - 1].coff << 16) | (u - blk[lo - 1].uoff);
}
__host__ __device__ inline int reg2bin(long long beg, long long end, int min_shift, int n_lvls) {       // hts_reg2bin
    int l, s = min_shift, t = ((1 << (3 * n_lvls)) - 1) / 7;
    for (--end, l = n_lvls; l > 0; --l, s += 3, t -= 1 << (3 * l))
        if (beg >> s == end >> s) return t + (int)(beg >> s);
    return 0;
}

enum { E_TID = 1, E_UNSORTED = 2, E_NOCOOR = 4, E_LIN = 8 };

// per record: what hts_idx_push is called with, plus the run-start flag
__global__ __launch_bounds__(256)
void idx_record_kernel(const uint8_t *__restrict__ b, const uint64_t *__restrict__ rec_off, uint64_t n, int32_t n_ref, const Blk *__restrict__ blk,
                       uint64_t nb, uint64_t file_size, int32_t *tid_o, uint32_t *bin_o, uint64_t *voff_o, uint32_t *flag_o,
                       const uint64_t *__restrict__ lin_base, const uint32_t *__restrict__ lin_cap, unsigned long long *lin, uint32_t *lin_n,
                       uint32_t *cnt, uint32_t *err, int min_shift, int n_lvls) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    long long raw_beg = 0;                                            // the position as stored, before hts_idx_push clamps it
    auto fields = [&](uint64_t r, int32_t &tid, long long &beg, long long &end, bool &mapped) {
        const uint8_t *x = b + rec_off[r] + 4;
        tid = (int32_t)ld32(x); beg = (int32_t)ld32(x + 4);
        const uint32_t l_qname = ld32(x + 8) & 0xffu, x3 = ld32(x + 12), n_cigar = x3 & 0xffffu, flag = x3 >> 16;
        mapped = !(flag & 4u);
        long long rlen = 0;
        if (mapped) for (uint32_t k = 0; k < n_cigar; k++) {           // bam_cigar2rlen: M D N = X consume the reference
            const uint32_t c = ld32(x + 32 + l_qname + 4u * k), op = c & 0xfu;
            if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) rlen += c >> 4;
        }
        if (rlen == 0) rlen = 1;
        end = beg + rlen;
        if (tid < 0) { beg = -1; end = 0; }
        raw_beg = beg;
        if (tid >= 0) { if (beg < 0) beg = 0; if (end <= 0) end = 1; }
    };
    int32_t tid; long long beg, end; bool mapped;
    fields(i, tid, beg, end, mapped);
    const long long my_raw = raw_beg;
    uint32_t e = 0;
    if (tid >= n_ref) { e |= E_TID; tid = -1; beg = -1; end = 0; }
    const int bin = reg2bin(beg, end, min_shift, n_lvls);
    const uint64_t vo = voff_of(blk, nb, file_size, rec_off[i]);
    bool start = true;
    if (i) {
        int32_t ptid; long long pbeg, pend; bool pm;
        fields(i - 1, ptid, pbeg, pend, pm);
        if (ptid >= n_ref) ptid = -1;
        if (ptid == tid) {
            if (tid >= 0 && pbeg > my_raw) e |= E_UNSORTED;        // last_coor (clamped) against the raw position, as the reference does
            start = reg2bin(pbeg, pend, min_shift, n_lvls) != bin;
        } else if (ptid < 0 && tid >= 0) e |= E_NOCOOR;
    }
    tid_o[i] = tid; bin_o[i] = (uint32_t)bin; voff_o[i] = vo; flag_o[i] = start ? 1u : 0u;
    if (tid >= 0) {
        const long long wb = beg >> min_shift, we = (end - 1) >> min_shift;
        if ((unsigned long long)we >= lin_cap[tid]) e |= E_LIN;
        else {
            for (long long w = wb; w <= we; w++) atomicMin(&lin[lin_base[tid] + (uint64_t)w], (unsigned long long)vo);
            atomicMax(&lin_n[tid], (uint32_t)(we + 1));
        }
        atomicAdd(&cnt[2 * tid + (mapped ? 0 : 1)], 1u);
        // a reference's records must be one contiguous block: count its run starts (tid changes)
        if (i == 0 || ((int32_t)ld32(b + rec_off[i - 1] + 4) != tid)) atomicAdd(&cnt[2 * n_ref + tid], 1u);
    } else atomicAdd(&cnt[3 * n_ref], 1u);
    if (e) atomicOr(err, e);
}
// compaction of the run starts (rank = exclusive prefix sum of the flags)
__global__ __launch_bounds__(256)
void idx_runs_kernel(const int32_t *__restrict__ tid, const uint32_t *__restrict__ bin, const uint64_t *__restrict__ voff,
                     const uint32_t *__restrict__ flag, const uint64_t *__restrict__ rank, uint64_t n, Run *runs) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n || !flag[i]) return;
    Run r; r.u = voff[i]; r.tid = tid[i]; r.bin = bin[i];
    runs[rank[i]] = r;
}

}  // namespace hgi

namespace hg { int scan32_to64(hg_ctx *ctx, const uint32_t *d_v, uint64_t n, uint64_t *d_out, hipStream_t s); }

extern "C" int hg_csi_levels(uint64_t max_ref_len, int min_shift) {      // hts_adjust_csi_settings with n_lvls = 0 (sam_index, sam.c:1003-1012)
    int n_lvls = 0;
    unsigned long long maxpos = 1ull << min_shift;
    while (max_ref_len + 256 > maxpos && n_lvls < 9) { n_lvls++; maxpos *= 8; }
    return n_lvls;
}

extern "C" long hg_bai_build_dev(hg_ctx *ctx, const void *d_bam, uint64_t len, uint64_t first_record_off, int32_t n_ref, const uint32_t *ref_len,
                                 const uint64_t *d_rec_off, uint64_t nrec, const hg_bgzf_desc *blocks, uint64_t nblocks, uint64_t file_size,
                                 uint8_t *out, size_t out_cap, void *stream) {
    return hg_idx_build_dev(ctx, d_bam, len, first_record_off, n_ref, ref_len, d_rec_off, nrec, blocks, nblocks, file_size, 0, 14, 5, out, out_cap, stream);
}

extern "C" long hg_idx_build_dev(hg_ctx *ctx, const void *d_bam, uint64_t len, uint64_t first_record_off, int32_t n_ref, const uint32_t *ref_len,
                                 const uint64_t *d_rec_off, uint64_t nrec, const hg_bgzf_desc *blocks, uint64_t nblocks, uint64_t file_size,
                                 int csi, int min_shift, int n_lvls, uint8_t *out, size_t out_cap, void *stream) {
    if (min_shift < 1 || min_shift > 30 || n_lvls < 1 || n_lvls > 9) return HG_EINVAL;
    if (!ctx || n_ref < 0 || (n_ref && !ref_len) || (nrec && (!d_bam || !d_rec_off)) || (nblocks && !blocks) || !out) return HG_EINVAL;
    hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;
    hipStream_t s = (hipStream_t)stream;
    using hgi::Blk; using hgi::Run;
    std::vector<Blk> blk(nblocks);
    for (uint64_t k = 0; k < nblocks; k++) { blk[k].coff = blocks[k].coff; blk[k].uoff = blocks[k].uoff; blk[k].ulen = blocks[k].ulen; blk[k].pad = 0; }
    // linear-index windows per reference: what the header's length needs, plus slack for reads hanging over the end
    std::vector<uint64_t> lin_base(n_ref + 1, 0); std::vector<uint32_t> lin_cap(n_ref);
    for (int32_t t = 0; t < n_ref; t++) { lin_cap[t] = (ref_len[t] >> min_shift) + 64u; lin_base[t + 1] = lin_base[t] + lin_cap[t]; }
    const uint64_t nlin = lin_base[n_ref];
    const size_t ncnt = 3 * (size_t)n_ref + 1;
    int rc;
    if ((rc = hg::ensure_scratch(ctx, 8, nrec * 24 + 64)) || (rc = hg::ensure_scratch(ctx, 9, nrec * 8 + 64)) ||
        (rc = hg::ensure_scratch(ctx, 10, nblocks * sizeof(Blk) + 64)) ||
        (rc = hg::ensure_scratch(ctx, 11, nlin * 8 + (size_t)n_ref * 16 + ncnt * 4 + 256)) ||
        (rc = hg::ensure_scratch(ctx, 12, nrec * 4 + 64))) return rc;
    uint64_t *d_voff = (uint64_t *)ctx->d_scratch[8]; int32_t *d_tid = (int32_t *)(d_voff + nrec); uint32_t *d_bin = (uint32_t *)(d_tid + nrec);
    uint64_t *d_rank = (uint64_t *)ctx->d_scratch[9];
    Blk *d_blk = (Blk *)ctx->d_scratch[10];
    unsigned long long *d_lin = (unsigned long long *)ctx->d_scratch[11];
    uint64_t *d_lbase = (uint64_t *)(d_lin + nlin); uint32_t *d_lcap = (uint32_t *)(d_lbase + n_ref + 1);
    uint32_t *d_linn = d_lcap + n_ref, *d_cnt = d_linn + n_ref, *d_err = d_cnt + ncnt;
    uint32_t *d_flag = (uint32_t *)ctx->d_scratch[12];
    bool ok = hipMemsetAsync(d_lin, 0xff, nlin * 8, s) == hipSuccess &&
              hipMemsetAsync(d_linn, 0, ((size_t)n_ref + ncnt + 1) * 4, s) == hipSuccess &&
              (!nblocks || hipMemcpyAsync(d_blk, blk.data(), nblocks * sizeof(Blk), hipMemcpyHostToDevice, s) == hipSuccess) &&
              hipMemcpyAsync(d_lbase, lin_base.data(), ((size_t)n_ref + 1) * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              (!n_ref || hipMemcpyAsync(d_lcap, lin_cap.data(), (size_t)n_ref * 4, hipMemcpyHostToDevice, s) == hipSuccess);
    if (!ok) return HG_ELAUNCH;
    uint64_t nruns = 0;
    std::vector<Run> runs;
    if (nrec) {
        hipLaunchKernelGGL(hgi::idx_record_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, (const uint8_t *)d_bam, d_rec_off, nrec, n_ref,
                           (const Blk *)d_blk, nblocks, file_size, d_tid, d_bin, d_voff, d_flag, (const uint64_t *)d_lbase, (const uint32_t *)d_lcap,
                           d_lin, d_linn, d_cnt, d_err, min_shift, n_lvls);
        if ((rc = hg::scan32_to64(ctx, d_flag, nrec, d_rank, s))) return rc;
        if (hipMemcpyAsync(&nruns, d_rank + nrec, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
        if ((rc = hg::ensure_scratch(ctx, 13, nruns * sizeof(Run) + 64))) return rc;
        hipLaunchKernelGGL(hgi::idx_runs_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, (const int32_t *)d_tid, (const uint32_t *)d_bin,
                           (const uint64_t *)d_voff, (const uint32_t *)d_flag, (const uint64_t *)d_rank, nrec, (Run *)ctx->d_scratch[13]);
        runs.resize(nruns);
        if (hipMemcpyAsync(runs.data(), ctx->d_scratch[13], nruns * sizeof(Run), hipMemcpyDeviceToHost, s) != hipSuccess) return HG_ELAUNCH;
    }
    std::vector<uint64_t> lin(nlin); std::vector<uint32_t> linn(n_ref), cnt(ncnt); uint32_t err = 0;
    if ((nlin && hipMemcpyAsync(lin.data(), d_lin, nlin * 8, hipMemcpyDeviceToHost, s) != hipSuccess) ||
        (n_ref && hipMemcpyAsync(linn.data(), d_linn, (size_t)n_ref * 4, hipMemcpyDeviceToHost, s) != hipSuccess) ||
        hipMemcpyAsync(cnt.data(), d_cnt, ncnt * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return HG_ELAUNCH;
    if (err & hgi::E_LIN) return HG_EINVAL;                               // a read far beyond its reference's declared length
    if (err) return HG_BAM_EUNSORTED;
    for (int32_t t = 0; t < n_ref; t++) if (cnt[2 * n_ref + t] > 1) return HG_BAM_EUNSORTED;   // "Chromosome blocks not continuous"
    // ---- fold the runs into bins (insert_to_b in hts_idx_push order) and the meta bins -------------------------------
    const uint32_t N_BINS = (uint32_t)(((1ull << (3 * n_lvls + 3)) - 1) / 7), META = N_BINS + 1;
    const uint64_t final_off = hgi::voff_of(blk.data(), nblocks, file_size, len);
    typedef std::pair<uint64_t, uint64_t> Chunk;
    std::vector<std::map<uint32_t, std::vector<Chunk>>> bins(n_ref);
    for (uint64_t r = 0; r < nruns; r++) {
        const uint64_t v = r + 1 < nruns ? runs[r + 1].u : final_off;
        if (runs[r].tid >= 0) bins[runs[r].tid][runs[r].bin].push_back({runs[r].u, v});
        if (runs[r].tid >= 0 && (r + 1 == nruns || runs[r + 1].tid != runs[r].tid)) {   // last run of its reference: the meta bin
            uint64_t rb = r;
            while (rb > 0 && runs[rb - 1].tid == runs[r].tid) rb--;
            auto &m = bins[runs[r].tid][META];
            m.push_back({runs[rb].u, v});
            m.push_back({cnt[2 * runs[r].tid], cnt[2 * runs[r].tid + 1]});
        }
    }
    auto level = [](uint32_t b) { int l = 0; while (b) { b = (b - 1) >> 3; l++; } return l; };
    auto by_u = [](const Chunk &a, const Chunk &b) { return a.first < b.first; };
    for (int32_t t = 0; t < n_ref; t++) {
        // update_loff: the linear index is back-filled from the right
        uint64_t *L = lin.data() + lin_base[t];
        for (long long l = (long long)linn[t] - 2; l >= 0; l--) if (L[l] == ~0ull) L[l] = L[l + 1];
        // compress_binning: small bins move into their parent, deepest level first
        auto &B = bins[t];
        for (int l = n_lvls; l > 0; l--) {
            std::vector<uint32_t> keys;
            for (auto &kv : B) if (kv.first < N_BINS && level(kv.first) == l) keys.push_back(kv.first);
            for (uint32_t k : keys) {
                auto &p = B[k];
                if (l < n_lvls && p.size() > 1) std::sort(p.begin(), p.end(), by_u);
                if ((p.back().second >> 16) - (p.front().first >> 16) < 0x10000u) {
                    auto q = B.find((k - 1) >> 3);
                    if (q == B.end()) continue;
                    q->second.insert(q->second.end(), p.begin(), p.end());
                    B.erase(k);
                }
            }
        }
        auto b0 = B.find(0);
        if (b0 != B.end()) std::sort(b0->second.begin(), b0->second.end(), by_u);
        for (auto &kv : B) {                                              // chunks touching the same BGZF block merge
            if (kv.first >= N_BINS) continue;
            auto &p = kv.second; size_t m = 0;
            for (size_t l = 1; l < p.size(); l++) {
                if (p[m].second >> 16 >= p[l].first >> 16) { if (p[m].second < p[l].second) p[m].second = p[l].second; }
                else p[++m] = p[l];
            }
            p.resize(m + 1);
        }
    }
    // ---- idx_save_core ---------------------------------------------------------------------------------------------------
    std::vector<uint8_t> o;
    auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) o.push_back((uint8_t)(v >> (8 * k))); };
    auto put64 = [&](uint64_t v) { put32((uint32_t)v); put32((uint32_t)(v >> 32)); };
    o.insert(o.end(), {(uint8_t)(csi ? 'C' : 'B'), (uint8_t)(csi ? 'S' : 'A'), 'I', 1});
    if (csi) { put32((uint32_t)min_shift); put32((uint32_t)n_lvls); put32(0); }         // hts_idx_write_out: no meta block
    put32((uint32_t)n_ref);
    for (int32_t t = 0; t < n_ref; t++) {
        put32((uint32_t)bins[t].size());
        for (auto &kv : bins[t]) {
            put32(kv.first);
            if (csi) {                                                    // update_loff (hts.c:2443-2453): the linear index, folded into the bins
                uint64_t loff = 0;
                if (kv.first < N_BINS) {
                    const int l = level(kv.first);
                    const uint64_t bot = (uint64_t)(kv.first - (uint32_t)(((1ull << (3 * l)) - 1) / 7)) << ((n_lvls - l) * 3);
                    loff = bot < linn[t] ? lin[lin_base[t] + bot] : 0;
                }
                put64(loff);
            }
            put32((uint32_t)kv.second.size());
            for (auto &c : kv.second) { put64(c.first); put64(c.second); }
        }
        if (!csi) {
            put32(linn[t]);
            for (uint32_t w = 0; w < linn[t]; w++) put64(lin[lin_base[t] + w]);
        }
    }
    put64(cnt[3 * n_ref]);                                                // n_no_coor
    if (o.size() > out_cap) return HG_EINVAL;
    memcpy(out, o.data(), o.size());
    return (long)o.size();
}

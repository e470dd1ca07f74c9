// cram_file_host.hip -- a whole CRAM 2.x / 3.x file to an uncompressed BAM stream: the job of `samtools view -u -b in.cram` without the
// BGZF layer, as a composition of the pieces of this library.  Host code only (no kernel of its own):
//   1. the container / block walk of cram_read_container and cram_read_block (reference cram/cram_io.c:3590-3760, 1414-1500): file
//      definition, file-header container (the SAM header text), data containers = compression header block + slices;
//   2. every compressed block of the file through hg_cram_uncompress_blocks_crc_host (cram_uncompress_block incl. its CRC check) in ONE
//      batch;
//   3. every slice through hg_cram_decode_bam_host (cram_decode_slice + cram_to_bam on the device) in ONE batch;
//   4. the BAM header of bam_hdr_write (sam.c): magic, the header text, the @SQ names and lengths.
// The caller supplies the reference sequences the file was written against (or none: bases come out as '=' plus the stored edits, as
// the reference does with no_ref files); embedded references are taken from the file.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <tuple>
#include <thread>
#include <atomic>
#include <chrono>
#include <vector>
#include "htsgpu.h"
#include "hg_internal.h"
#include "cram_records_plan.h"
#include "hg_md5.h"

namespace {

struct Blk { int32_t method, ctype, cid; uint32_t csz, usz; const uint8_t *data; uint32_t crc_part, crc; size_t out; };   // out: index into the decoded buffers
struct Sl { size_t hdr; std::vector<size_t> body; int32_t ref_seq_id; int64_t start, span; int32_t embedded; size_t comp; };

uint32_t crc32_small(const uint8_t *p, size_t n) {                      // CRC-32 of a block header (a few bytes): bitwise is enough
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u))); }
    return ~c;
}

}  // namespace

// ---- several GPUs ("independent blocks / slices are sharded across the GPUs of one node with a trivial static split").  HTS_GPU_DEVICES
// ("0-7", "0,2,5") names the devices, as for the BGZF front-end.  A whole-file job keeps the caller's context as its first shard and opens one
// sibling context per FURTHER entry of the list (process-wide, created on first use); blocks and slices are cut into that many contiguous
// ranges of about equal bytes, every range runs its own batches on its own device from its own host thread, and the outputs are joined in
// file order -- no collective, no cross-device traffic.  An entry may repeat a device ("0,0": two contexts on one GPU), which is how the
// sharded path is tested on a one-GPU box.
namespace {
std::vector<hg_ctx *> job_contexts(hg_ctx *ctx) {
    static std::mutex m;
    static std::vector<int> devs;
    static std::vector<hg_ctx *> peers;
    static bool parsed = false;
    std::lock_guard<std::mutex> lk(m);
    if (!parsed) {
        parsed = true;
        if (const char *d = getenv("HTS_GPU_DEVICES")) {
            for (const char *p = d; *p;) {
                if (*p < '0' || *p > '9') { p++; continue; }
                char *q; const long a = strtol(p, &q, 10); long b = a;
                if (*q == '-') b = strtol(q + 1, &q, 10);
                for (long x = a; x <= b && devs.size() < 64; x++) devs.push_back((int)x);
                p = q;
            }
        }
        peers.assign(devs.size(), nullptr);
    }
    std::vector<hg_ctx *> out{ctx};
    for (size_t i = 1; i < devs.size(); i++) {
        if (!peers[i] && hg_init(devs[i], &peers[i]) != HG_OK) peers[i] = nullptr;
        if (peers[i]) out.push_back(peers[i]);
    }
    return out;
}
// [0, n) cut into `parts` contiguous ranges of about equal weight; cut[k] .. cut[k + 1] is range k
template <class W> std::vector<size_t> cut_ranges(size_t n, size_t parts, W weight) {
    std::vector<size_t> cut{0};
    uint64_t total = 0; for (size_t i = 0; i < n; i++) total += weight(i);
    uint64_t acc = 0; size_t i = 0;
    for (size_t k = 1; k < parts; k++) {
        const uint64_t want = total * k / parts;
        while (i < n && acc < want) acc += weight(i++);
        cut.push_back(i);
    }
    cut.push_back(n);
    return cut;
}
}  // namespace

int hg_cram_decode_bam_devsrc(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref, const char *const *rg_names,
                              int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap, uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes,
                              int32_t *status, const char *name_prefix, const uint8_t *dev_lo, const uint8_t *dev_hi, void *wait_ev);     // cram_records.hip
struct hg_entropy_inplace;                                                                                              // cram_entropy_host.hip
int hg_entropy_decode_inplace_launch(hg_ctx *ctx, size_t n, const uint8_t *codec, const uint8_t *const *in, const uint32_t *in_len, const uint64_t *in_off,
                                     const uint32_t *out_len, const uint64_t *out_off, uint64_t in_bytes, uint64_t obytes, hg_entropy_inplace **handle);
int hg_entropy_decode_inplace_finish(hg_ctx *ctx, hg_entropy_inplace *H, int32_t *status);

namespace {
constexpr int NOT_FUSABLE = 0x7f01;       // internal: this run goes through the host-buffer composition instead
struct Walk { std::vector<Blk> blocks; std::vector<Sl> slices; size_t file_hdr = (size_t)-1; uint64_t bases = 0; };

// the blocks of one container body (cram_read_container's successor calls: cram_read_block for the compression header, cram_read_slice for each
// slice, cram/cram_io.c:1414-1500, cram_decode.c:3386-3416): block headers parsed in place, slices = header block + the blocks that follow it
int walk_body(Walk &W, const uint8_t *p, const uint8_t *cend, int32_t nblk, int major) {
    std::vector<Blk> &blocks = W.blocks; std::vector<Sl> &slices = W.slices;
    hgr::Cursor b{p, cend};
    size_t comp = (size_t)-1;
    bool in_slice = false;                                           // blocks before the container's first slice header belong to no slice
    for (int32_t k = 0; k < nblk && b.p < b.end; k++) {
        const uint8_t *h0 = b.p;
        Blk x; memset(&x, 0, sizeof x);
        x.method = b.byte(); x.ctype = b.byte(); x.cid = b.itf8(); x.csz = (uint32_t)b.itf8(); x.usz = (uint32_t)b.itf8();
        if (b.bad || (size_t)(b.end - b.p) < (size_t)x.csz + (major >= 3 ? 4u : 0u)) return HG_EINVAL;
        x.crc_part = crc32_small(h0, (size_t)(b.p - h0));
        x.data = b.p; b.p += x.csz;
        if (major >= 3) { x.crc = (uint32_t)b.p[0] | (uint32_t)b.p[1] << 8 | (uint32_t)b.p[2] << 16 | (uint32_t)b.p[3] << 24; b.p += 4; }
        blocks.push_back(x);
        const size_t me = blocks.size() - 1;
        if (x.ctype == 0 && W.file_hdr == (size_t)-1) W.file_hdr = me;                     // FILE_HEADER
        else if (x.ctype == 1) comp = me;                                              // COMPRESSION_HEADER
        else if (x.ctype == 2 || x.ctype == 3) { Sl s; s.hdr = me; s.comp = comp; s.ref_seq_id = -1; s.start = s.span = 0; s.embedded = -1; slices.push_back(s); in_slice = true; }
        else if ((x.ctype == 4 || x.ctype == 5) && in_slice) slices.back().body.push_back(me);
    }
    return 0;
}

// every block through cram_uncompress_block in one batch (one batch per device when HTS_GPU_DEVICES names several)
int uncompress_walk(const std::vector<hg_ctx *> &ctxs, int major, const std::vector<Blk> &blocks, std::vector<std::vector<uint8_t>> &dec) {
    const size_t nb = blocks.size();
    dec.assign(nb, std::vector<uint8_t>());
    std::vector<int32_t> method(nb), status(nb); std::vector<const uint8_t *> in(nb); std::vector<uint32_t> il(nb), ol(nb), part(nb), crc(nb); std::vector<uint8_t *> out(nb);
    for (size_t i = 0; i < nb; i++) {
        dec[i].resize(blocks[i].usz ? blocks[i].usz : 1);
        method[i] = blocks[i].method; in[i] = blocks[i].data; il[i] = blocks[i].csz; ol[i] = blocks[i].usz; out[i] = dec[i].data(); part[i] = blocks[i].crc_part; crc[i] = blocks[i].crc;
    }
    const std::vector<size_t> bcut = cut_ranges(nb, nb >= 64 ? ctxs.size() : 1, [&](size_t i) { return (uint64_t)blocks[i].csz + blocks[i].usz + 64; });
    std::vector<int> brc(bcut.size() - 1, HG_OK);
    auto unc = [&](size_t k) {
        const size_t a = bcut[k], n_ = bcut[k + 1] - a;
        hg_ctx *c = ctxs[k];
        brc[k] = !n_ ? HG_OK : major >= 3
            ? hg_cram_uncompress_blocks_crc_host(c, n_, method.data() + a, in.data() + a, il.data() + a, part.data() + a, crc.data() + a, out.data() + a, ol.data() + a, status.data() + a)
            : hg_cram_uncompress_blocks_host(c, n_, method.data() + a, in.data() + a, il.data() + a, out.data() + a, ol.data() + a, status.data() + a);
    };
    {
        std::vector<std::thread> th;
        for (size_t k = 1; k + 1 < bcut.size(); k++) th.emplace_back(unc, k);
        unc(0);
        for (auto &t : th) t.join();
    }
    for (int rc : brc) if (rc != HG_OK) return rc;                                      // a block that fails (CRC, malformed, bzip2 / lzma) fails the file, like cram_read_slice / cram_decode_slice
    return HG_OK;
}

// crc(A || B) from crc(A), crc(B), |B| (GF(2) polynomial arithmetic; x^8 is 0x00800000 in the reflected form)
uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
uint32_t crc_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    uint32_t xp = 0x00800000u, acc = 0x80000000u;
    for (; len_b; len_b >>= 1) { if (len_b & 1) acc = crc_mulmod(acc, xp); xp = crc_mulmod(xp, xp); }
    return crc_mulmod(acc, crc_a) ^ crc_b;
}

// The blocks of a run decoded WHERE THE RECORD DECODER WILL READ THEM: the container bodies go to the device as they lie in the caller's buffer (one transfer),
// the payload CRCs, the gzip members (inflate kernel) and the rANS 4x8 streams are taken straight from that image into one decoded image beside it -- both in a
// sibling context's scratch, the two codec families on two streams.  bptr[k] = where block k's plain bytes are: a device address for data blocks (RAW ones
// point into the uploaded image itself), the host bytes for the header blocks the planner parses.  NOT_FUSABLE: a method of CRAM 3.1, bzip2 / lzma, a
// compressed header block, bodies scattered over the address space -- the caller takes the host-buffer composition.  HG_EBLOCK: a block failed its CRC or its codec.
// (The call returns with the kernels and the read-back of their verdicts IN FLIGHT: the caller plans the record decoder, fetches and digests references and
// allocates its device memory meanwhile -- on a first run that is as long as the 150 ms rANS kernel -- and orders its gather behind P.done; blocks_verify then
// reads the verdicts.)
struct BlocksPending {
    hg_ctx *A = nullptr; hipEvent_t done = nullptr; int major = 3;
    uint8_t *res = nullptr;                                              // PAGE-LOCKED: a read-back into pageable memory does not return before the kernels ahead of it have finished
    size_t ng = 0, nr = 0, nb = 0, r_gz = 0, r_rs = 0, r_crc = 0;
    hg_entropy_inplace *ent = nullptr; size_t nent = 0;                  // the rANS Nx16 / range-coder streams of the run (cram_entropy_host.hip)
    std::vector<std::vector<uint8_t>> tok_out; int tok_rc = HG_OK;        // tok3 / fqzcomp / bzip2 / lzma blocks: decoded through the host entry point, handed to the record decoder as host blocks
    ~BlocksPending() {
        if (A && (done || res || ent)) (void)hipStreamSynchronize(A->stream);
        if (ent) (void)hg_entropy_decode_inplace_finish(A, ent, nullptr);
        if (done) (void)hipEventDestroy(done);
        if (res) (void)hipHostFree(res);
    }
};
int blocks_verify(BlocksPending &P, const Walk &W) {
    if (!P.A || !P.res) return HG_EINVAL;
    if (hipStreamSynchronize(P.A->stream) != hipSuccess) return HG_ELAUNCH;
    int rc = HG_OK;
    if (P.ent) { rc = hg_entropy_decode_inplace_finish(P.A, P.ent, nullptr); P.ent = nullptr; }
    if (rc != HG_OK) return rc;
    if (P.tok_rc != HG_OK) return P.tok_rc;
    const int32_t *st_gz = (const int32_t *)(P.res + P.r_gz), *st_rs = (const int32_t *)(P.res + P.r_rs); const uint32_t *crc = (const uint32_t *)(P.res + P.r_crc);
    for (size_t g = 0; g < P.ng; g++) if (st_gz[g] != 0) return HG_EBLOCK;
    for (size_t r = 0; r < P.nr; r++) if (st_rs[r] != 0) return HG_EBLOCK;
    if (P.major >= 3)
        for (size_t k = 0; k < P.nb; k++) {
            const Blk &b = W.blocks[k];
            if ((b.csz ? crc_join(b.crc_part, crc[k], b.csz) : b.crc_part) != b.crc) return HG_EBLOCK;       // cram_uncompress_block's first step (cram_io.c:1585-1592)
        }
    return HG_OK;
}
int blocks_on_device(hg_ctx *ctx, int major, const Walk &W, std::vector<const uint8_t *> &bptr, const uint8_t *&dev_lo, const uint8_t *&dev_hi, BlocksPending &P) {
    const std::vector<Blk> &blocks = W.blocks;
    const size_t nb = blocks.size();
    if (!nb) return NOT_FUSABLE;
    const uint8_t *lo = nullptr, *hi = nullptr;
    uint64_t csum = 0;
    size_t ng = 0, nr = 0, ne = 0, nt = 0, nraw = 0;
    for (const Blk &b : blocks) {
        const bool header = b.ctype <= 3;
        if (header ? b.method != HG_CRAM_RAW : (b.method < 0 || b.method > HG_CRAM_TOK3)) return NOT_FUSABLE;                     // a compressed header block, an unknown method
        if (b.method == HG_CRAM_RAW && b.csz != b.usz) return HG_EBLOCK;
        if (!lo || b.data < lo) lo = b.data;
        if (!hi || b.data + b.csz > hi) hi = b.data + b.csz;
        csum += b.csz;
        if (!b.usz || header) continue;
        if (b.method == HG_CRAM_GZIP) ng++;
        else if (b.method == HG_CRAM_RANS4x8) nr++;
        else if (b.method == HG_CRAM_RANSNx16 || b.method == HG_CRAM_ARITH) ne++;
        else if (b.method == HG_CRAM_RAW) nraw++;
        else nt++;                                                       // tok3, fqzcomp, bzip2, lzma: through the host entry point (below)
    }
    const uint64_t span = (uint64_t)(hi - lo);
    if (span > 0xe0000000ull || span > 2 * csum + (64u << 20)) return NOT_FUSABLE;
    if (!ctx->sub[3] && hg_init(ctx->device, &ctx->sub[3]) != HG_OK) return HG_ENOMEM;
    hg_ctx *A = ctx->sub[3];
    hg::CtxGuard ga(A); if (ga.rc) return ga.rc;
    // ---- layout: the uploaded image in scratch slot 0, the decoded image at the front of slot 1 (the entropy plans' work area behind it); tables and results further back
    const uint64_t raw_bytes = (span + 255u) & ~255ull;
    std::vector<uint64_t> out_off(nb, 0);
    uint64_t dec_bytes = 0, scratch_words = 0;
    auto on_host = [](int32_t m) { return m == HG_CRAM_TOK3 || m == HG_CRAM_FQZ || m == HG_CRAM_BZIP2 || m == HG_CRAM_LZMA; };
    for (size_t k = 0; k < nb; k++) if (blocks[k].usz && blocks[k].ctype > 3 && !on_host(blocks[k].method)) { out_off[k] = dec_bytes; dec_bytes += ((uint64_t)blocks[k].usz + 15u) & ~15ull; }
    dec_bytes = (dec_bytes + 255u) & ~255ull;
    if (raw_bytes > 0xfffffff0ull || dec_bytes > 0xfffffff0ull * 2) return NOT_FUSABLE;
    std::vector<hg_bgzf_desc> gz(ng); std::vector<hg_stream_desc> rs(nr);
    std::vector<uint64_t> c_off(nb); std::vector<uint32_t> c_len(nb);
    std::vector<uint8_t> e_codec(ne); std::vector<const uint8_t *> e_in(ne); std::vector<uint32_t> e_il(ne), e_ol(ne); std::vector<uint64_t> e_io(ne), e_oo(ne);
    std::vector<uint64_t> w_src(nraw), w_dst(nraw); std::vector<uint32_t> w_len(nraw);
    std::vector<size_t> tok_of; tok_of.reserve(nt);
    {
        size_t g = 0, r = 0, e = 0, w = 0;
        for (size_t k = 0; k < nb; k++) {
            const Blk &b = blocks[k];
            c_off[k] = (uint64_t)(b.data - lo); c_len[k] = b.csz;
            if (!b.usz || b.ctype <= 3) continue;
            if (b.method == HG_CRAM_GZIP) gz[g++] = hg_bgzf_desc{c_off[k], out_off[k], b.csz, b.usz};
            else if (b.method == HG_CRAM_RANS4x8) {
                uint32_t usz = 0;
                if (b.csz >= 9) usz = (uint32_t)b.data[5] | (uint32_t)b.data[6] << 8 | (uint32_t)b.data[7] << 16 | (uint32_t)b.data[8] << 24;
                if (usz != b.usz) return HG_EBLOCK;                              // cram_uncompress_block: "uncompressed size mismatch" (cram_io.c:1671-1680)
                memset(&rs[r], 0, sizeof rs[r]);
                rs[r].in_off = c_off[k]; rs[r].in_len = b.csz; rs[r].out_off = out_off[k]; rs[r].out_len = usz; rs[r].scratch_off = (uint32_t)scratch_words;
                scratch_words += HG_RANS4X8_SCRATCH_WORDS(b.csz);
                if (scratch_words > 0xffffffffull) return NOT_FUSABLE;
                r++;
            } else if (b.method == HG_CRAM_RANSNx16 || b.method == HG_CRAM_ARITH) {
                e_codec[e] = b.method == HG_CRAM_ARITH; e_in[e] = b.data; e_il[e] = b.csz; e_ol[e] = b.usz; e_io[e] = c_off[k]; e_oo[e] = out_off[k]; e++;
            } else if (on_host(b.method)) tok_of.push_back(k);
            else { w_src[w] = c_off[k]; w_dst[w] = out_off[k]; w_len[w] = b.usz; w++; }
        }
    }
    // Longest streams first: a stream is one chain on one lane group and a launch lasts as long as its longest group, so the groups start with one long
    // stream each (the quality blocks) and pick up the short ones behind them -- the kernels hand out streams in descriptor order.
    std::stable_sort(rs.begin(), rs.end(), [](const hg_stream_desc &a, const hg_stream_desc &b) { return a.in_len > b.in_len; });
    std::stable_sort(gz.begin(), gz.end(), [](const hg_bgzf_desc &a, const hg_bgzf_desc &b) { return a.clen > b.clen; });
    auto up64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_gz = 0, o_rs = o_gz + up64(ng * sizeof(hg_bgzf_desc)), o_coff = o_rs + up64(nr * sizeof(hg_stream_desc)), o_clen = o_coff + up64(nb * 8), tab_bytes = o_clen + up64(nb * 4);
    const size_t r_gz = 0, r_rs = r_gz + up64(ng * 4), r_crc = r_rs + up64(nr * 4), res_bytes = r_crc + up64(nb * 4);
    int rc;
    if ((rc = hg::ensure_scratch(A, 0, (size_t)raw_bytes + 256)) || (rc = hg::ensure_scratch(A, 8, tab_bytes + 64)) || (rc = hg::ensure_scratch(A, 9, res_bytes + 64)) ||
        (rc = hg::ensure_scratch(A, 10, (size_t)scratch_words * 4 + 64))) return rc;
    uint8_t *d_img = (uint8_t *)A->d_scratch[0], *d_tab = (uint8_t *)A->d_scratch[8], *d_res = (uint8_t *)A->d_scratch[9];
    hipStream_t s = A->stream;
    std::vector<uint8_t> tab(tab_bytes, 0);
    if (ng) memcpy(tab.data() + o_gz, gz.data(), ng * sizeof(hg_bgzf_desc));
    if (nr) memcpy(tab.data() + o_rs, rs.data(), nr * sizeof(hg_stream_desc));
    memcpy(tab.data() + o_coff, c_off.data(), nb * 8); memcpy(tab.data() + o_clen, c_len.data(), nb * 4);
    if (hipMemcpyAsync(d_img, lo, (size_t)span, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(d_tab, tab.data(), tab_bytes, hipMemcpyHostToDevice, s) != hipSuccess) return HG_ELAUNCH;
    P.A = A; P.major = major; P.ng = ng; P.nr = nr; P.nb = nb; P.r_gz = r_gz; P.r_rs = r_rs; P.r_crc = r_crc;
    // the CRAM 3.1 entropy coders first: their plan sizes slot 1 (decoded image + work area), the other decoders then write into the same image
    if (ne) {
        if ((rc = hg_entropy_decode_inplace_launch(A, ne, e_codec.data(), e_in.data(), e_il.data(), e_io.data(), e_ol.data(), e_oo.data(), raw_bytes, dec_bytes, &P.ent))) { (void)hipStreamSynchronize(s); return rc; }
        P.nent = ne;
    } else if ((rc = hg::ensure_scratch(A, 1, (size_t)dec_bytes + 256))) { (void)hipStreamSynchronize(s); return rc; }
    uint8_t *d_dec = (uint8_t *)A->d_scratch[1];
    // gzip members on the side stream, rANS 4x8 + the CRCs + the RAW blocks on this one
    if (ng) {
        hipStream_t s2 = hg::fork_side(A, s);
        rc = hg::launch_bgzf_inflate(A, d_img, (size_t)span, (const hg_bgzf_desc *)(d_tab + o_gz), ng, d_dec, (size_t)dec_bytes, (int32_t *)(d_res + r_gz), s2, 1);
        hg::join_side(A, s);
        if (rc) { (void)hipStreamSynchronize(s); return rc; }
    }
    if (nr && (rc = hg::launch_rans4x8_decode(A, d_img, (const hg_stream_desc *)(d_tab + o_rs), nr, d_dec, (int32_t *)(d_res + r_rs), (uint32_t *)A->d_scratch[10], s))) { (void)hipStreamSynchronize(s); return rc; }
    if (major >= 3 && (rc = hg::launch_crc32(A, d_img, (const uint64_t *)(d_tab + o_coff), (const uint32_t *)(d_tab + o_clen), nb, (uint32_t *)(d_res + r_crc), s))) { (void)hipStreamSynchronize(s); return rc; }
    if (nraw && (rc = hg::stage_gather_dev(A, d_img, w_src.data(), w_len.data(), d_dec, w_dst.data(), nraw, s))) { (void)hipStreamSynchronize(s); return rc; }
    if (hipHostMalloc((void **)&P.res, res_bytes + 64, hipHostMallocDefault) != hipSuccess) { P.res = nullptr; (void)hipStreamSynchronize(s); return HG_ENOMEM; }
    if (hipEventCreateWithFlags(&P.done, hipEventDisableTiming) != hipSuccess) { P.done = nullptr; (void)hipStreamSynchronize(s); return HG_ELAUNCH; }
    if (hipMemcpyAsync(P.res, d_res, res_bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(P.done, s) != hipSuccess) { (void)hipStreamSynchronize(s); return HG_ELAUNCH; }
    // The blocks whose decoders have no in-place form -- the name tokeniser's (method 8: ~60 token streams per block, a plan of its own), fqzcomp's, and the host
    // libraries' bzip2 / lzma -- go through the ordinary host entry point, beside all of the above on their own family contexts, and reach the record decoder as host
    // blocks: a few hundred KB of names per slice, or the qualities of an fqzcomp file (whose decoder, not the round trip, is what such a file waits for).
    if (nt) {
        P.tok_out.resize(nt);
        std::vector<int32_t> tm(nt); std::vector<const uint8_t *> ti(nt); std::vector<uint32_t> til(nt), tol(nt); std::vector<uint8_t *> to(nt); std::vector<int32_t> tst(nt, 0);
        for (size_t q = 0; q < nt; q++) { const Blk &b = blocks[tok_of[q]]; P.tok_out[q].resize(b.usz ? b.usz : 1); tm[q] = b.method; ti[q] = b.data; til[q] = b.csz; tol[q] = b.usz; to[q] = P.tok_out[q].data(); }
        P.tok_rc = hg_cram_uncompress_blocks_host(ctx, nt, tm.data(), ti.data(), til.data(), to.data(), tol.data(), tst.data());
    }
    bptr.assign(nb, nullptr);
    {
        size_t q = 0;
        for (size_t k = 0; k < nb; k++) {
            const Blk &b = blocks[k];
            if (b.ctype <= 3) bptr[k] = b.data;
            else if (b.usz && on_host(b.method)) bptr[k] = P.tok_out[q++].data();
            else bptr[k] = d_dec + out_off[k];
        }
    }
    dev_lo = d_dec; dev_hi = d_dec + dec_bytes + 256;
    return HG_OK;
}

int slices_to_bam(hg_ctx *ctx, const std::vector<hg_ctx *> &ctxs, int major, Walk &W, const uint8_t *const *dec, const uint8_t *dev_lo, const uint8_t *dev_hi, void *wait_ev, int nref, const int64_t *sq_len,
                  const char *const *rg_names, int nrg, const hg_cram_ref_seq *refs, int nrefs_given, hg_cram_get_ref_fn get_ref, void *get_ref_ud, int flags, int decode_md,
                  const char *name_prefix, uint8_t *rec_out, size_t rec_cap, uint64_t *rec_bytes_out, uint64_t *nrecords);
}  // namespace

extern "C" int hg_cram_file_to_bam_host2(hg_ctx *ctx, const uint8_t *cram, size_t cram_len, const hg_cram_ref_seq *refs, int nrefs_given, uint8_t *bam_out,
                                         size_t bam_cap, uint64_t *bam_bytes, uint64_t *nrecords, int flags, const char *name_prefix);
extern "C" int hg_cram_file_to_bam_host(hg_ctx *ctx, const uint8_t *cram, size_t cram_len, const hg_cram_ref_seq *refs, int nrefs_given, uint8_t *bam_out,
                                        size_t bam_cap, uint64_t *bam_bytes, uint64_t *nrecords) {
    return hg_cram_file_to_bam_host2(ctx, cram, cram_len, refs, nrefs_given, bam_out, bam_cap, bam_bytes, nrecords, 0, nullptr);
}
extern "C" int hg_cram_file_to_bam_host2(hg_ctx *ctx, const uint8_t *cram, size_t cram_len, const hg_cram_ref_seq *refs, int nrefs_given, uint8_t *bam_out,
                                         size_t bam_cap, uint64_t *bam_bytes, uint64_t *nrecords, int flags, const char *name_prefix) {
    if (!ctx || !cram || !bam_out || !bam_bytes || (nrefs_given && !refs)) return HG_EINVAL;
    if (cram_len < 26 || memcmp(cram, "CRAM", 4) != 0) return HG_EINVAL;
    const int major = cram[4];
    if (major != 2 && major != 3) return HG_BLOCK_EUNSUPPORTED;
    // ---- 1. walk ----
    Walk W;
    hgr::Cursor c{cram + 26, cram + cram_len};
    while (c.p < c.end) {
        if (c.end - c.p < 4) return HG_EINVAL;
        const uint32_t clen = (uint32_t)c.p[0] | (uint32_t)c.p[1] << 8 | (uint32_t)c.p[2] << 16 | (uint32_t)c.p[3] << 24; c.p += 4;
        (void)c.itf8(); (void)c.itf8(); (void)c.itf8();                  // reference id, start, span of the container
        (void)c.itf8();                                                  // number of records
        if (major >= 3) (void)c.ltf8(); else (void)c.itf8();             // record counter
        W.bases += (uint64_t)c.ltf8();
        const int32_t nblk = c.itf8(), nland = c.itf8();
        for (int32_t i = 0; i < nland; i++) (void)c.itf8();
        if (major >= 3) c.p += 4;                                        // CRC of the container header
        if (c.bad || nblk < 0 || c.p > c.end || (size_t)(c.end - c.p) < clen) return HG_EINVAL;
        const uint8_t *cend = c.p + clen;
        if (const int wrc = walk_body(W, c.p, cend, nblk, major)) return wrc;
        c.p = cend;
    }
    const size_t file_hdr = W.file_hdr;
    if (file_hdr == (size_t)-1) return HG_EINVAL;
    // ---- 2. every block through cram_uncompress_block in one batch (one batch per device when HTS_GPU_DEVICES names several) ----
    const std::vector<hg_ctx *> ctxs = job_contexts(ctx);
    std::vector<std::vector<uint8_t>> dec;
    if (const int urc = uncompress_walk(ctxs, major, W.blocks, dec)) return urc;
    // ---- SAM header: text, @SQ, @RG ----
    const std::vector<uint8_t> &fh = dec[file_hdr];
    if (fh.size() < 4) return HG_EINVAL;
    const uint32_t tl = (uint32_t)fh[0] | (uint32_t)fh[1] << 8 | (uint32_t)fh[2] << 16 | (uint32_t)fh[3] << 24;
    if ((size_t)tl + 4 > fh.size()) return HG_EINVAL;
    const std::string text((const char *)fh.data() + 4, tl);
    std::vector<std::string> sq_name, rg_id; std::vector<int64_t> sq_len;
    for (size_t at = 0; at < text.size();) {
        size_t e = text.find('\n', at); if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(at, e - at); at = e + 1;
        const bool sq = line.compare(0, 3, "@SQ") == 0, rg = line.compare(0, 3, "@RG") == 0;
        if (!sq && !rg) continue;
        std::string name; int64_t ln = 0;
        for (size_t f = 3; f < line.size();) {
            size_t t = line.find('\t', f + 1); if (t == std::string::npos) t = line.size();
            const std::string fld = line.substr(f + 1, t - f - 1); f = t;
            if (sq && fld.compare(0, 3, "SN:") == 0) name = fld.substr(3);
            if (sq && fld.compare(0, 3, "LN:") == 0) ln = strtoll(fld.c_str() + 3, nullptr, 10);
            if (rg && fld.compare(0, 3, "ID:") == 0) name = fld.substr(3);
        }
        if (sq) { sq_name.push_back(name); sq_len.push_back(ln); } else rg_id.push_back(name);
    }
    const int nref = (int)sq_name.size();
    // ---- BAM header (bam_hdr_write) ----
    size_t hb = 4 + 4 + text.size() + 4;
    for (int i = 0; i < nref; i++) hb += 4 + sq_name[(size_t)i].size() + 1 + 4;
    if (hb > bam_cap) { *bam_bytes = hb; return HG_ENOMEM; }
    {
        uint8_t *o = bam_out;
        auto put32 = [&](uint32_t v) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24); o += 4; };
        memcpy(o, "BAM\1", 4); o += 4;
        put32((uint32_t)text.size()); memcpy(o, text.data(), text.size()); o += text.size();
        put32((uint32_t)nref);
        for (int i = 0; i < nref; i++) { const std::string &nm = sq_name[(size_t)i]; put32((uint32_t)nm.size() + 1); memcpy(o, nm.c_str(), nm.size() + 1); o += nm.size() + 1; put32((uint32_t)sq_len[(size_t)i]); }
    }
    // ---- 3. slices -> BAM records ----
    std::vector<const char *> rgp; for (auto &r : rg_id) rgp.push_back(r.c_str());
    uint64_t rec_bytes = 0;
    std::vector<const uint8_t *> dptr(dec.size()); for (size_t i = 0; i < dec.size(); i++) dptr[i] = dec[i].data();
    const int rc = slices_to_bam(ctx, ctxs, major, W, dptr.data(), nullptr, nullptr, nullptr, nref, sq_len.data(), rgp.empty() ? nullptr : rgp.data(), (int)rgp.size(), refs, nrefs_given, nullptr, nullptr, flags, -1 /* hts_open's default */,
                                 name_prefix, bam_out + hb, bam_cap - hb, &rec_bytes, nrecords);
    *bam_bytes = hb + rec_bytes;
    return rc;
}

// The same path for a reader that walks the file itself (cram_read_container on the caller's side: cram_record_front.c under cram_get_bam_seq): the BODIES of a
// run of data containers -> their BAM records, back to back, no BAM header.
extern "C" int hg_cram_containers_to_bam_host(hg_ctx *ctx, int major, size_t ncontainers, const hg_cram_container *cont, int nref, const int64_t *sq_len,
                                              const char *const *rg_names, int nrg, const hg_cram_ref_seq *refs, int nrefs_given, hg_cram_get_ref_fn get_ref, void *get_ref_ud,
                                              int flags, int decode_md, const char *name_prefix, uint8_t *bam_out, size_t bam_cap, uint64_t *bam_bytes, uint64_t *nrecords) {
    if (!ctx || (ncontainers && !cont) || !bam_out || !bam_bytes || (nrefs_given && !refs) || (nref && !sq_len) || (nrg && !rg_names)) return HG_EINVAL;
    if (major != 2 && major != 3) return HG_BLOCK_EUNSUPPORTED;
    static const bool stats = getenv("HTS_GPU_STATS") != nullptr;
    static const bool fuse = [] { const char *e = getenv("HTS_GPU_CRAM_FUSED"); return !(e && e[0] == '0'); }();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = std::chrono::steady_clock::now();
    Walk W;
    for (size_t i = 0; i < ncontainers; i++) {
        if (!cont[i].body || cont[i].num_blocks < 0) return HG_EINVAL;
        if (const int wrc = walk_body(W, cont[i].body, cont[i].body + cont[i].body_len, cont[i].num_blocks, major)) return wrc;
        W.bases += cont[i].bases;
    }
    const std::vector<hg_ctx *> ctxs = job_contexts(ctx);
    uint64_t cb = 0, ub = 0; for (const Blk &b : W.blocks) { cb += b.csz; ub += b.usz; }
    const auto t1 = std::chrono::steady_clock::now();
    uint64_t rec_bytes = 0;
    // ---- the fused form: the blocks never come back to the host between the block codecs and the record decoder ----
    if (fuse && ctxs.size() == 1) {
        hg::CtxGuard guard_(ctx); if (guard_.rc) return guard_.rc;               // the sibling context's buffers hold this run's blocks until the records are out
        std::vector<const uint8_t *> bptr; const uint8_t *dev_lo = nullptr, *dev_hi = nullptr;
        BlocksPending pend;
        int rc = blocks_on_device(ctx, major, W, bptr, dev_lo, dev_hi, pend);
        const auto t2 = std::chrono::steady_clock::now();
        if (rc == HG_OK) {
            rc = slices_to_bam(ctx, ctxs, major, W, bptr.data(), dev_lo, dev_hi, pend.done, nref, sq_len, rg_names, nrg, refs, nrefs_given, get_ref, get_ref_ud, flags, decode_md, name_prefix,
                               bam_out, bam_cap, &rec_bytes, nrecords);
            const int vrc = blocks_verify(pend, W);                       // the blocks' own verdicts (CRC, codec status): a run with a bad block is not delivered, whatever the records looked like
            if (vrc != HG_OK && rc != NOT_FUSABLE) rc = vrc;
        }
        if (rc != NOT_FUSABLE) {
            *bam_bytes = rec_bytes;
            if (stats) fprintf(stderr, "[htsgpu stats] cram run (fused): %zu containers, %zu slices, %zu blocks %.1f -> %.1f MB, %.1f MB of BAM; walk %.1f ms, blocks %.1f ms, slices -> BAM %.1f ms (rc %d)\n",
                               ncontainers, W.slices.size(), W.blocks.size(), cb / 1e6, ub / 1e6, rec_bytes / 1e6, ms(t0, t1), ms(t1, t2), ms(t2, std::chrono::steady_clock::now()), rc);
            return rc;
        }
        rec_bytes = 0;
    }
    std::vector<std::vector<uint8_t>> dec;
    const auto t1b = std::chrono::steady_clock::now();
    if (const int urc = uncompress_walk(ctxs, major, W.blocks, dec)) return urc;
    const auto t2 = std::chrono::steady_clock::now();
    std::vector<const uint8_t *> dptr(dec.size()); for (size_t i = 0; i < dec.size(); i++) dptr[i] = dec[i].data();
    const int rc = slices_to_bam(ctx, ctxs, major, W, dptr.data(), nullptr, nullptr, nullptr, nref, sq_len, rg_names, nrg, refs, nrefs_given, get_ref, get_ref_ud, flags, decode_md, name_prefix, bam_out, bam_cap,
                                 &rec_bytes, nrecords);
    *bam_bytes = rec_bytes;
    if (stats) fprintf(stderr, "[htsgpu stats] cram run: %zu containers, %zu slices, %zu blocks %.1f -> %.1f MB, %.1f MB of BAM; walk %.1f ms, blocks %.1f ms, slices -> BAM %.1f ms (rc %d)\n",
                       ncontainers, W.slices.size(), W.blocks.size(), cb / 1e6, ub / 1e6, rec_bytes / 1e6, ms(t0, t1), ms(t1b, t2), ms(t2, std::chrono::steady_clock::now()), rc);
    return rc;
}

namespace {
int slices_to_bam(hg_ctx *ctx, const std::vector<hg_ctx *> &ctxs, int major, Walk &W, const uint8_t *const *dec, const uint8_t *dev_lo, const uint8_t *dev_hi, void *wait_ev, int nref, const int64_t *sq_len,
                  const char *const *rg_names, int nrg, const hg_cram_ref_seq *refs, int nrefs_given, hg_cram_get_ref_fn get_ref, void *get_ref_ud, int flags, int decode_md,
                  const char *name_prefix, uint8_t *rec_out, size_t rec_cap, uint64_t *rec_bytes_out, uint64_t *nrecords) {
    std::vector<Blk> &blocks = W.blocks; std::vector<Sl> &slices = W.slices;
    const uint64_t bases = W.bases;
    const size_t ns = slices.size();
    const auto t_prep = std::chrono::steady_clock::now();
    // references on demand (get_ref): a slice that brings its own bases (embedded block, RR = 0) never asks; the others ask for the sequence their header
    // names, multi-reference slices for the ids their RI series holds -- what cram_decode_slice does record by record (cram_decode.c:2436-2446, 2610-2650)
    std::vector<hg_cram_ref_seq> lazy; std::vector<char> asked;
    if (get_ref) { lazy.assign((size_t)std::max(nref, 0), hg_cram_ref_seq{nullptr, 0}); asked.assign(lazy.size(), 0); refs = lazy.data(); nrefs_given = (int)lazy.size(); }
    auto need = [&](int r) {
        if (!get_ref || r < 0 || r >= nrefs_given || asked[(size_t)r]) return;
        asked[(size_t)r] = 1;
        hg_cram_ref_seq q{nullptr, 0};
        if (get_ref(get_ref_ud, r, &q) == 0 && q.bases) lazy[(size_t)r] = q;
    };
    std::vector<size_t> md5_jobs;
    std::vector<hgr::SliceHeader> shs(ns);
    std::vector<hg_cram_slice_blocks> sb(ns);
    std::vector<std::vector<int32_t>> ids(ns); std::vector<std::vector<const uint8_t *>> ptr(ns); std::vector<std::vector<uint32_t>> len(ns);
    std::vector<std::vector<hg_cram_ref_span>> spans(ns);
    for (size_t i = 0; i < ns; i++) {
        Sl &s = slices[i];
        if (s.comp == (size_t)-1) return HG_EINVAL;
        hgr::SliceHeader sh;
        const uint8_t *hd = dec[s.hdr];
        if (hgr::parse_slice_header(hd, blocks[s.hdr].usz, major, sh)) return HG_EINVAL;
        if (dev_lo && sh.ref_base_id >= 0) return NOT_FUSABLE;                // an embedded reference is digested on the host
        shs[i] = sh;
        s.embedded = sh.ref_base_id;
        s.ref_seq_id = sh.ref_seq_id; s.start = sh.ref_seq_start; s.span = sh.ref_seq_span;
        memset(&sb[i], 0, sizeof sb[i]);
        sb[i].comp_hdr = dec[s.comp]; sb[i].comp_hdr_len = blocks[s.comp].usz;
        sb[i].slice_hdr = hd; sb[i].slice_hdr_len = blocks[s.hdr].usz;
        size_t taken = 0;
        for (size_t k : s.body) {
            if ((int32_t)taken >= sh.nblocks) break;                     // blocks beyond the slice's count belong to nobody
            taken++;
            if (blocks[k].ctype == 5) { sb[i].core = dec[k]; sb[i].core_len = blocks[k].usz; continue; }
            ids[i].push_back(blocks[k].cid); ptr[i].push_back(dec[k]); len[i].push_back(blocks[k].usz);
            if (s.embedded >= 0 && blocks[k].cid == s.embedded && sh.ref_seq_id >= 0)
                spans[i].push_back(hg_cram_ref_span{sh.ref_seq_id, sh.ref_seq_start, dec[k], blocks[k].usz, sh.ref_seq_id < nref ? sq_len[sh.ref_seq_id] : (int64_t)blocks[k].usz});
        }
        sb[i].nblocks = (uint32_t)ids[i].size(); sb[i].content_id = ids[i].data(); sb[i].data = ptr[i].data(); sb[i].len = len[i].data();
        // RR = 0 in the container's preservation map: written without a reference -- none is attached (cram_decode.c:2436-2442)
        hgr::PlanHost ph;
        const bool no_ref = hgr::plan_from_compression_header(ph, dec[s.comp], blocks[s.comp].usz) == 0 && ph.no_ref;
        if (spans[i].empty() && !no_ref) {                               // the caller's references: the slice's stretch (s->ref_start .. ref_end of the reference), or whole sequences for a multi-reference slice (each staged once per batch)
            if (get_ref) {
                if (sh.ref_seq_id >= 0) need(sh.ref_seq_id);
                else if (sh.ref_seq_id == -2) {
                    // the reference ids of a multi-reference slice: its RI series, when that is an EXTERNAL block of ITF8 values (what every writer emits) or a
                    // one-symbol Huffman code; anything else: every reference of the header
                    bool known = false;
                    const int ci = ph.plan.codec_of[hgr::S_RI];
                    if (ci >= 0 && (size_t)ci < ph.codecs.size()) {
                        const hgr::Codec &cd = ph.codecs[(size_t)ci];
                        if (cd.kind == hgr::E_EXTERNAL && cd.a >= 0 && (size_t)cd.a < ph.slot_id.size()) {
                            const int32_t cid = ph.slot_id[(size_t)cd.a];
                            for (size_t k = 0; k < ids[i].size(); k++) if (ids[i][k] == cid) {
                                const uint8_t *ri = ptr[i][k];
                                std::vector<uint8_t> fetched;
                                if (dev_lo && ri >= dev_lo && ri < dev_hi) {
                                    // the block is on the device (fused run): its few KB come over once its decoder has finished -- a multi-reference slice costs the run the overlap, not the fused path
                                    fetched.resize(len[i][k] ? len[i][k] : 1);
                                    if ((wait_ev && hipEventSynchronize((hipEvent_t)wait_ev) != hipSuccess) ||
                                        (len[i][k] && hipMemcpy(fetched.data(), ri, len[i][k], hipMemcpyDeviceToHost) != hipSuccess)) return HG_ELAUNCH;
                                    ri = fetched.data();
                                }
                                hgr::Cursor rc{ri, ri + len[i][k]};
                                while (rc.p < rc.end && !rc.bad) need(rc.itf8());
                                known = true; break;
                            }
                        } else if (cd.kind == hgr::E_HUFFMAN && cd.b == 1 && (size_t)cd.a < ph.huff.size()) { need(ph.huff[(size_t)cd.a].symbol); known = true; }
                    }
                    if (!known) for (int r = 0; r < nrefs_given; r++) need(r);
                }
            }
            auto whole = [&](int r) { if (r >= 0 && r < nrefs_given && refs[r].bases) spans[i].push_back(hg_cram_ref_span{r, 1, refs[r].bases, (uint32_t)std::min<uint64_t>(refs[r].len, 0xfffffff0ull), r < nref ? sq_len[r] : (int64_t)refs[r].len}); };
            if (sh.ref_seq_id >= 0) {
                const int r = sh.ref_seq_id;
                if (r < nrefs_given && refs[r].bases && sh.ref_seq_start >= 1 && (uint64_t)sh.ref_seq_start <= refs[r].len) {
                    const uint64_t avail = refs[r].len - (uint64_t)sh.ref_seq_start + 1;
                    const uint64_t want = sh.ref_seq_span > 0 ? (uint64_t)sh.ref_seq_span : avail;      // cram_get_ref(fd, id, start, start + span - 1)
                    spans[i].push_back(hg_cram_ref_span{r, sh.ref_seq_start, refs[r].bases + sh.ref_seq_start - 1, (uint32_t)std::min<uint64_t>(std::min(avail, want), 0xfffffff0ull),
                                                        r < nref ? sq_len[r] : (int64_t)refs[r].len});
                }
            } else if (sh.ref_seq_id == -2) for (int r = 0; r < nrefs_given; r++) whole(r);
        }
        // "MD5 checksum reference mismatch" (cram_decode.c:2480-2540): the slice header's digest against the span about to be used
        static const uint8_t zero16[16] = {0};
        if (major >= 2 && sh.ref_seq_id >= 0 && !(flags & HG_CRAM_IGNORE_MD5) && sh.has_md5 && memcmp(sh.md5, zero16, 16) != 0 && !no_ref) {
            if (spans[i].empty()) return HG_EBLOCK;                      // no reference and no embedded block: "Unable to fetch reference"
            md5_jobs.push_back(i);
        }
        sb[i].nrefs = (uint32_t)spans[i].size(); sb[i].refs = spans[i].data(); sb[i].decode_md = decode_md;
    }
    if (!md5_jobs.empty()) {                                             // host threads: the digests are independent (the reference computes them in its slice workers)
        std::atomic<size_t> next{0}; std::atomic<int> bad{0};
        auto work = [&]() {
            for (size_t j; (j = next.fetch_add(1)) < md5_jobs.size();) {
                const size_t i = md5_jobs[j];
                const hg_cram_ref_span &sp = spans[i][0];
                const hgr::SliceHeader &sh = shs[i];
                uint64_t start = sh.ref_seq_start >= sp.start ? (uint64_t)(sh.ref_seq_start - sp.start) : 0u, len = (uint64_t)sh.ref_seq_span;
                const bool embedded = slices[i].embedded >= 0 && sp.bases != (refs && sh.ref_seq_id < nrefs_given && refs[sh.ref_seq_id].bases ? refs[sh.ref_seq_id].bases + sh.ref_seq_start - 1 : nullptr);
                if (embedded) { start = 0; len = sp.len; }               // an embedded reference block is digested whole
                if (start > sp.len) start = sp.len;
                if (start + len > sp.len) len = sp.len - start;
                uint8_t dg[16]; hgr::md5_of(sp.bases + start, len, dg);
                if (memcmp(dg, sh.md5, 16) != 0) bad = 1;
            }
        };
        const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::min<size_t>(md5_jobs.size(), 16));
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        if (bad) return HG_EBLOCK;                                       // the reference: error "MD5 checksum reference mismatch", cram_decode_slice returns -1
    }
    static const bool timing = getenv("HG_CRAM_RECORDS_TIMING") != nullptr;
    if (timing) fprintf(stderr, "cram run prep (slice headers, references, MD5 of %zu spans): %.1f ms\n", md5_jobs.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prep).count());
    std::vector<uint64_t> rec_off(ns + 1, 0); std::vector<int32_t> status(ns, 0);
    auto slice_bytes = [&](size_t i) { uint64_t b = (uint64_t)sb[i].core_len + 64; for (uint32_t k = 0; k < sb[i].nblocks; k++) b += ((uint64_t)sb[i].len[k] + 15) & ~15ull; return b; };
    uint64_t all_bytes = 0; for (size_t i = 0; i < ns; i++) all_bytes += slice_bytes(i);
    // slices s0 .. s1 on context c into dst (cap bytes): records back to back, ro[i - s0] = records before slice i
    auto decode_range = [&](hg_ctx *c, size_t s0, size_t s1, uint8_t *dst, size_t cap, uint64_t &bytes_out, uint64_t *ro_out) -> int {
        int rc = HG_OK;
        uint64_t rec_bytes = 0; ro_out[0] = 0;
        // the blocks of one device batch are addressed with 32 bits (cram_records_plan.h): a large range goes through in several batches
        for (size_t i0 = s0; i0 < s1 && rc == HG_OK;) {
            uint64_t bytes = 0; size_t i1 = i0;
            while (i1 < s1) {
                const uint64_t b = slice_bytes(i1);
                if (i1 > i0 && bytes + b > 0xc0000000ull) break;
                bytes += b; i1++;
            }
            std::vector<uint64_t> ro(i1 - i0 + 1, 0);
            uint64_t got = 0;
            // Room for bases + qualities: the container headers' `bases` fields say how much, but they are a writer's claim, not a fact (a
            // writer may leave 0 there): when a slice comes back "no room" the batch is decoded again with more, instead of failing the file.
            uint64_t seq_cap = std::min<uint64_t>(s1 - s0 == ns ? bases : (uint64_t)((double)bases * 1.25 * (double)bytes / (double)(all_bytes ? all_bytes : 1)), 1ull << 36) + 4096;
            for (int attempt = 0;; attempt++) {
                got = 0;
                rc = hg_cram_decode_bam_devsrc(c, i1 - i0, sb.data() + i0, major, nref, rg_names, nrg, seq_cap, dst + rec_bytes,
                                               cap - rec_bytes, ro.data(), nullptr, &got, status.data() + i0, name_prefix, dev_lo, dev_hi, wait_ev);
                bool no_room = false;
                for (size_t k = i0; k < i1; k++) if (status[k] == hgr::ERR_UNSUPPORTED) no_room = true;    // "does not fit" is one of its meanings
                if ((rc != HG_OK && rc != HG_EBLOCK) || !no_room || attempt == 1 || seq_cap >= (1ull << 34)) break;   // once: a slice the decoder does not support says -3, too
                seq_cap = seq_cap * 4 + (64ull << 20);
            }
            rec_bytes += got;
            for (size_t k = 0; k <= i1 - i0; k++) ro_out[i0 - s0 + k] = ro_out[i0 - s0] + ro[k];
            i0 = i1;
        }
        bytes_out = rec_bytes;
        return rc;
    };
    uint64_t rec_bytes = 0;
    int rc = HG_OK;
    const std::vector<size_t> scut = cut_ranges(ns, ns >= 2 * ctxs.size() ? ctxs.size() : 1, slice_bytes);
    if (scut.size() == 2) {
        rc = decode_range(ctx, 0, ns, rec_out, rec_cap, rec_bytes, rec_off.data());
    } else {
        // every further range decodes into a buffer of its own (its place in the stream is only known when the ranges before it are done);
        // the first one writes in place
        const size_t nr = scut.size() - 1;
        std::vector<std::vector<uint8_t>> tmp(nr); std::vector<uint64_t> got(nr, 0); std::vector<int> rrc(nr, HG_OK);
        std::vector<std::vector<uint64_t>> ro(nr);
        for (size_t k = 0; k < nr; k++) ro[k].assign(scut[k + 1] - scut[k] + 1, 0);
        const uint64_t room = rec_cap;
        auto run = [&](size_t k) {
            if (k == 0) { rrc[0] = decode_range(ctxs[0], scut[0], scut[1], rec_out, (size_t)room, got[0], ro[0].data()); return; }
            uint64_t w = 0; for (size_t i = scut[k]; i < scut[k + 1]; i++) w += slice_bytes(i);
            const uint64_t cap = std::min<uint64_t>(room, (uint64_t)((double)room * 1.5 * (double)w / (double)(all_bytes ? all_bytes : 1)) + (4u << 20));
            tmp[k].resize((size_t)cap);
            rrc[k] = decode_range(ctxs[k], scut[k], scut[k + 1], tmp[k].data(), (size_t)cap, got[k], ro[k].data());
        };
        {
            std::vector<std::thread> th;
            for (size_t k = 1; k < nr; k++) th.emplace_back(run, k);
            run(0);
            for (auto &t : th) t.join();
        }
        bool retry_serial = false;
        for (size_t k = 0; k < nr; k++) if (rrc[k] == HG_ENOMEM) retry_serial = true;       // a range's private buffer was too small: the plain way
        if (retry_serial) {
            std::fill(status.begin(), status.end(), 0);
            rc = decode_range(ctx, 0, ns, rec_out, rec_cap, rec_bytes, rec_off.data());
        } else {
            for (size_t k = 0; k < nr; k++) {
                if (rrc[k] != HG_OK && (rc == HG_OK || rc == HG_EBLOCK)) rc = rrc[k];
                if (rec_bytes + got[k] > room) { rc = HG_ENOMEM; break; }
                if (k) memcpy(rec_out + rec_bytes, tmp[k].data(), (size_t)got[k]);
                for (size_t i = 0; i + 1 < ro[k].size() + 0; i++) rec_off[scut[k] + i] = rec_off[scut[k]] + ro[k][i];
                rec_off[scut[k + 1]] = rec_off[scut[k]] + ro[k].back();
                rec_bytes += got[k];
            }
        }
    }
    *rec_bytes_out = rec_bytes;
    if (nrecords) *nrecords = rec_off[ns];
    return rc;
}
}  // namespace

// ---- the other direction: an uncompressed BAM stream (header + records, what hg_cram_file_to_bam_host returns / bam_hdr_write + bam_write1 produce) ->
//      a CRAM 3.0 file.  Host composition of library pieces, like the reader above:
//        1. bam_hdr_read's walk of the header (sam.c): text, reference names / lengths; @RG ids from the text;
//        2. hg_cram_encode_slices_host: records -> the series blocks and headers of slices of `records_per_slice` reads (device);
//        3. every block through the method auto-tuner (hg_cram_compress_blocks_metrics_host, one cram_metrics per content id as cram_encode.c keeps
//           one per data series) with the CRAM 3.0 method set GZIP | RANS0 | RANS1 -- rANS 4x8 is the codec of this library that is PINNED against
//           the reference's fixtures, so the files need nothing unpinned to be read;
//        4. framing (cram_write_file_def, cram_write_SAM_hdr, cram_write_container, cram_write_block, cram/cram_io.c:1511-1560, 3890-4060, 4380-4500):
//           file definition, header container, one container per slice (compression header block RAW, slice header block RAW, the series blocks),
//           the EOF container; block and container CRC-32s on the device.
namespace {
template <class O> void put_itf8(O &o, int32_t sv) {
    const uint32_t v = (uint32_t)sv;
    if (v < 0x80) o.push_back((uint8_t)v);
    else if (v < 0x4000) { o.push_back((uint8_t)(0x80 | (v >> 8))); o.push_back((uint8_t)v); }
    else if (v < 0x200000) { o.push_back((uint8_t)(0xc0 | (v >> 16))); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
    else if (v < 0x10000000) { o.push_back((uint8_t)(0xe0 | (v >> 24))); o.push_back((uint8_t)(v >> 16)); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
    else { o.push_back((uint8_t)(0xf0 | (v >> 28))); o.push_back((uint8_t)(v >> 20)); o.push_back((uint8_t)(v >> 12)); o.push_back((uint8_t)(v >> 4)); o.push_back((uint8_t)(v & 0x0f)); }
}
template <class O> void put_ltf8(O &o, uint64_t v) {                     // ltf8_put (cram_io.c:475-560), values below 2^56
    int extra = 0;
    while (extra < 7 && v >= (1ull << (7 * (extra + 1)))) extra++;
    if (extra == 0) { o.push_back((uint8_t)v); return; }
    o.push_back((uint8_t)((0xff00u >> extra) & 0xffu) | (uint8_t)(v >> (8 * extra)));
    for (int i = extra - 1; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i)));
}
template <class O> void put32le(O &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
}  // namespace

extern "C" int hg_bam_to_cram_host(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, const hg_cram_ref_seq *refs, int nrefs_given, uint32_t records_per_slice, int level,
                                   uint8_t *cram_out, size_t cram_cap, uint64_t *cram_bytes, uint64_t *nrecords) {
    return hg_bam_to_cram_host2(ctx, bam, bam_len, refs, nrefs_given, records_per_slice, level, 0, cram_out, cram_cap, cram_bytes, nrecords);
}

// flags: HG_CRAM_WRITE_V31 = a CRAM 3.1 file -- the block auto-tuner is offered what cram_compress_slice offers a 3.1 writer with use_rans and use_tok
// (htslib's "normal" profile): GZIP, GZIP_RLE, (level >= 5) GZIP_1, rANS Nx16 PR0 / PR1 (+ PR64 / PR9 / PR128 / PR193 above level 1, + PR129 / PR192 above
// level 5), the read names TOK3 instead of rANS (cram_encode.c:818-826, 937-942); HG_CRAM_WRITE_ARITH adds the range coder's sets (use_arith, :833-845; names: TOKA).
// Without V31: the CRAM 3.0 set GZIP | rANS 4x8.  fqzcomp is not offered (the writer would have to hand the quality block's per-record lengths over).
// A writer that comes back with more records (the whole-slice writer under cram_put_bam_seq, cram_record_front.c) keeps what a cram_fd keeps between slices: one
// cram_metrics per data series -- the method trials of the first slices are not repeated -- and the record counter of the container headers.
struct hg_cram_writer {
    std::vector<std::map<int32_t, hg_cram_metrics *>> met;              // per device range (one range on one device)
    uint64_t counter = 0;
    uint32_t records_per_slice = 10000; int level = 5, flags = 0;
};
static int bam_to_cram_impl(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, const hg_cram_ref_seq *refs, int nrefs_given, uint32_t records_per_slice, int level,
                            int flags, uint8_t *cram_out, size_t cram_cap, uint64_t *cram_bytes, uint64_t *nrecords, hg_cram_writer *W, const char *const *rg_names, int nrg);
extern "C" int hg_bam_to_cram_host2(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, const hg_cram_ref_seq *refs, int nrefs_given, uint32_t records_per_slice, int level,
                                    int flags, uint8_t *cram_out, size_t cram_cap, uint64_t *cram_bytes, uint64_t *nrecords) {
    return bam_to_cram_impl(ctx, bam, bam_len, refs, nrefs_given, records_per_slice, level, flags, cram_out, cram_cap, cram_bytes, nrecords, nullptr, nullptr, 0);
}
extern "C" hg_cram_writer *hg_cram_writer_new(uint32_t records_per_slice, int level, int flags) {
    hg_cram_writer *W = new (std::nothrow) hg_cram_writer;
    if (W) { W->records_per_slice = records_per_slice ? records_per_slice : 10000; W->level = level <= 0 ? 5 : level; W->flags = flags; }
    return W;
}
extern "C" void hg_cram_writer_free(hg_cram_writer *W) {
    if (!W) return;
    for (auto &mm : W->met) for (auto &m : mm) hg_cram_metrics_free(m.second);
    delete W;
}
extern "C" int hg_cram_writer_containers_host(hg_ctx *ctx, hg_cram_writer *W, const uint8_t *bam_records, size_t len, const hg_cram_ref_seq *refs, int nrefs_given,
                                              const char *const *rg_names, int nrg, uint8_t *out, size_t cap, uint64_t *out_bytes, uint64_t *nrecords) {
    if (!W || (nrg && !rg_names)) return HG_EINVAL;
    return bam_to_cram_impl(ctx, bam_records, len, refs, nrefs_given, W->records_per_slice, W->level, W->flags, out, cap, out_bytes, nrecords, W, rg_names, nrg);
}
static int bam_to_cram_impl(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, const hg_cram_ref_seq *refs, int nrefs_given, uint32_t records_per_slice, int level,
                            int flags, uint8_t *cram_out, size_t cram_cap, uint64_t *cram_bytes, uint64_t *nrecords, hg_cram_writer *W, const char *const *rg_names, int nrg) {
    if (!ctx || !bam || !cram_out || !cram_bytes || (nrefs_given && !refs)) return HG_EINVAL;
    const bool v31 = (flags & HG_CRAM_WRITE_V31) != 0, arith = v31 && (flags & HG_CRAM_WRITE_ARITH) != 0;
    if (!records_per_slice) records_per_slice = 10000;                   // the reference's default (cram/cram_structs.h:87-89)
    if (level <= 0) level = 5;
    // HTS_GPU_STATS=1: where the call's wall time went, on stderr (the stages below)
    const bool stats = getenv("HTS_GPU_STATS") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    double t_stage[5] = {0, 0, 0, 0, 0};
    auto lap = [&](int k) { const auto t = now(); t_stage[k] += std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; };
    // ---- 1. BAM header (a writer object brings records only, and the @RG ids with them)
    std::string text;
    size_t p = 0;
    std::vector<std::string> rg_id;
    if (W) for (int i = 0; i < nrg; i++) rg_id.push_back(rg_names[i] ? rg_names[i] : "");
    else {
    if (bam_len < 12 || memcmp(bam, "BAM\1", 4) != 0) return HG_EINVAL;
    auto rd32 = [&](size_t at) { return (uint32_t)bam[at] | (uint32_t)bam[at + 1] << 8 | (uint32_t)bam[at + 2] << 16 | (uint32_t)bam[at + 3] << 24; };
    const uint32_t l_text = rd32(4);
    if ((size_t)l_text + 12 > bam_len) return HG_EINVAL;
    text.assign((const char *)bam + 8, l_text);
    p = 8 + (size_t)l_text;
    const uint32_t n_ref = rd32(p); p += 4;
    for (uint32_t i = 0; i < n_ref; i++) { if (p + 4 > bam_len) return HG_EINVAL; const uint32_t ln = rd32(p); p += 4 + (size_t)ln + 4; if (p > bam_len) return HG_EINVAL; }
    }
    for (size_t at = 0; !W && at < text.size();) {
        size_t e = text.find('\n', at); if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(at, e - at); at = e + 1;
        if (line.compare(0, 3, "@RG") != 0) continue;
        for (size_t f = 3; f < line.size();) { size_t t = line.find('\t', f + 1); if (t == std::string::npos) t = line.size(); const std::string fld = line.substr(f + 1, t - f - 1); f = t; if (fld.compare(0, 3, "ID:") == 0) rg_id.push_back(fld.substr(3)); }
    }
    // ---- 2. records -> slices.  The records are counted (and their bases summed per slice, for the container headers) by the encoder's own device passes:
    //      a host walk of the block_size fields is one cache miss per record -- two such walks were 60 % of this call for 640 k records.
    const size_t ns_max = (bam_len - p) / 36 / records_per_slice + 2;
    // (no zero-filled vector: filling this size costs more host time than the device needs for the records)
    size_t blob_cap = (bam_len - p) * 2 + 65536 * ns_max + 4096;
    hg::CtxGuard whole_call(ctx); if (whole_call.rc) return whole_call.rc;   // the buffers below are the context's (hg::host_slab) for as long as this call runs
    uint8_t *blob = hg::host_slab(ctx, 0, blob_cap);
    if (!blob) return HG_ENOMEM;
    blob_cap = ctx->h_slab_cap[0];
    std::vector<uint64_t> soff(ns_max + 1, 0), sbases(ns_max, 0); std::vector<int32_t> sst(ns_max, 0);
    std::vector<const char *> rgp; for (auto &r : rg_id) rgp.push_back(r.c_str());
    int rc = HG_OK;
    size_t nrec = 0;
    const uint64_t counter0 = W ? W->counter : 0;
    if (bam_len > p) {
        uint64_t need = 0;
        rc = hg_cram_encode_slices_host2(ctx, bam + p, bam_len - p, &nrec, records_per_slice, refs, nrefs_given, rgp.empty() ? nullptr : rgp.data(), (int)rgp.size(), (int64_t)counter0, blob, blob_cap,
                                         soff.data(), ns_max, sst.data(), &need, sbases.data());
        if (rc == HG_ENOMEM && need > blob_cap) {                       // reads far from their reference (or no reference at all) store every base: several bytes per base
            blob = hg::host_slab(ctx, 0, need + 64);
            if (!blob) return HG_ENOMEM;
            blob_cap = ctx->h_slab_cap[0];
            rc = hg_cram_encode_slices_host2(ctx, bam + p, bam_len - p, &nrec, records_per_slice, refs, nrefs_given, rgp.empty() ? nullptr : rgp.data(), (int)rgp.size(), (int64_t)counter0, blob, blob_cap,
                                             soff.data(), ns_max, sst.data(), &need, sbases.data());
        }
        if (rc != HG_OK) return rc;                                     // a slice the encoder does not cover fails the file
    }
    if (nrecords) *nrecords = nrec;
    const size_t ns = (nrec + records_per_slice - 1) / records_per_slice;
    lap(0);
    // ---- 3. every series block through the auto-tuner
    struct Blk { int32_t cid; const uint8_t *p; uint32_t n; size_t slice; };
    struct SliceParts { const uint8_t *comp; uint32_t comp_len; const uint8_t *sh; uint32_t sh_len; size_t b0, b1; hgr::SliceHeader hdr; uint64_t bases; };
    std::vector<Blk> blks; std::vector<SliceParts> parts(ns);
    for (size_t k = 0; k < ns; k++) {
        const uint8_t *b = blob + soff[k];
        auto r32 = [&](const uint8_t *q) { return (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24; };
        SliceParts &P = parts[k];
        P.comp_len = r32(b); P.comp = b + 4; b += 4 + P.comp_len;
        P.sh_len = r32(b); P.sh = b + 4; b += 4 + P.sh_len;
        const uint32_t nb = r32(b); b += 4;
        P.b0 = blks.size();
        for (uint32_t i = 0; i < nb; i++) { const int32_t cid = (int32_t)r32(b); const uint32_t n = r32(b + 4); blks.push_back(Blk{cid, b + 8, n, k}); b += 8 + n; }
        P.b1 = blks.size();
        if (hgr::parse_slice_header(P.sh, P.sh_len, 3, P.hdr)) return HG_EINVAL;
        P.bases = sbases[k];
    }
    const size_t nb = blks.size();
    // compressed payloads: one arena, a block's slot as long as the block (a block that does not shrink is stored RAW, so nothing longer comes back)
    std::vector<uint64_t> coff(nb + 1, 0);
    for (size_t i = 0; i < nb; i++) coff[i + 1] = coff[i] + ((blks[i].n + 15u) & ~(uint64_t)15);
    uint8_t *cdata = hg::host_slab(ctx, 1, coff[nb] + 16);
    if (!cdata) return HG_ENOMEM;
    std::vector<uint32_t> clen(nb, 0); std::vector<int32_t> cmeth(nb, 0);
    if (nb) {
        // method sets per block (internal method ids of cram/cram_structs.h:215-266: GZIP 1, RANS0 4, RANS_PR0 5, ARITH_PR0 6, TOK3 8, GZIP_RLE 11, GZIP_1 12, RANS1 16,
        // RANS_PR1 / 64 / 9 / 128 / 129 / 192 / 193 = 17..23, TOKA 24, ARITH_PR1 / 64 / 9 / 128 / 129 / 192 / 193 = 25..31)
        uint32_t set_gen = (1u << 1) | (1u << 4) | (1u << 16), set_rn = set_gen;
        if (v31) {
            uint32_t ranspr = (1u << 5) | (1u << 17);
            if (level > 1) ranspr |= (1u << 18) | (1u << 19) | (1u << 20) | (1u << 23);
            if (level > 5) ranspr |= (1u << 21) | (1u << 22);
            set_gen = (1u << 1) | (1u << 11) | ranspr;
            if (arith) { set_gen |= (1u << 6) | (1u << 25); if (level > 1) set_gen |= (1u << 26) | (1u << 27) | (1u << 28) | (1u << 29) | (1u << 30) | (1u << 31); }
            if (level >= 5) set_gen |= 1u << 12;
            if (level == 1) set_gen = (set_gen & ~(1u << 1)) | (1u << 12);
            set_rn = (set_gen & ~(ranspr | (1u << 11))) | (arith ? 1u << 24 : 1u << 8);
        }
        const int32_t cid_rn = 10 + 6;                                                                                   // the RN series (cram_encode_plan.h: content id 10 + series)
        std::vector<hg_cram_metrics *> mp(nb); std::vector<uint32_t> sets(nb, set_gen);
        for (size_t i = 0; i < nb; i++) if (blks[i].cid == cid_rn) sets[i] = set_rn;
        std::vector<const uint8_t *> in(nb); std::vector<uint32_t> il(nb); std::vector<uint8_t *> out(nb);
        // Several devices (HTS_GPU_DEVICES): the slices are cut into one contiguous range per device; a range has its OWN set of cram_metrics
        // (one per content id, as cram_encode.c keeps one per data series) and learns its methods from its own blocks in file order, on its own
        // context and host thread -- the file stays valid CRAM, only the method chosen for a block may differ from a one-device run.
        const std::vector<hg_ctx *> ctxs = job_contexts(ctx);
        const std::vector<size_t> scut = cut_ranges(ns, ns >= 2 * ctxs.size() ? ctxs.size() : 1, [&](size_t k) { uint64_t w = 64; for (size_t i = parts[k].b0; i < parts[k].b1; i++) w += blks[i].n; return w; });
        const size_t nr = scut.size() - 1;
        std::vector<std::map<int32_t, hg_cram_metrics *>> own(W ? 0 : nr);
        if (W && W->met.size() < nr) W->met.resize(nr);
        std::vector<std::map<int32_t, hg_cram_metrics *>> &met = W ? W->met : own;
        for (size_t r = 0; r < nr; r++)
            for (size_t k = scut[r]; k < scut[r + 1]; k++)
                for (size_t i = parts[k].b0; i < parts[k].b1; i++) {
                    auto it = met[r].find(blks[i].cid);
                    if (it == met[r].end()) it = met[r].emplace(blks[i].cid, hg_cram_metrics_new()).first;
                    mp[i] = it->second; in[i] = blks[i].p; il[i] = blks[i].n;
                    out[i] = cdata + coff[i];
                }
        std::vector<int> rrc(nr, HG_OK);
        auto run = [&](size_t r) {
            // ALL blocks of the range in one call, in file order: the auto-tuner works through the blocks that share a metrics object in their order and
            // splits the call into rounds where a trial phase ends (hg_cram_compress_blocks_metrics_host), so the decisions are those of a slice-by-slice
            // walk -- which is what this was until round 4: one call per slice = ~30 latency-bound launches per slice, 12.7 s for 64 slices with the
            // fourteen-method sets of a 3.1 writer.  A few rounds now, each one batch over every slice.
            const size_t a = parts[scut[r]].b0, n = parts[scut[r + 1] - 1].b1 - a;
            if (n) rrc[r] = hg_cram_compress_blocks_metrics_host(ctxs[r], n, mp.data() + a, sets.data() + a, level, 3, in.data() + a, il.data() + a, out.data() + a, clen.data() + a, cmeth.data() + a);
        };
        {
            std::vector<std::thread> th;
            for (size_t r = 1; r < nr; r++) th.emplace_back(run, r);
            run(0);
            for (auto &t : th) t.join();
        }
        for (auto &mm : own) for (auto &m : mm) hg_cram_metrics_free(m.second);
        for (int r : rrc) if (r != HG_OK) return r;
    }
    lap(1);
    // ---- 4. framing, straight into the caller's buffer.  Every size is known by now (a block is 2 + itf8(cid) + itf8(csz) + itf8(usz) + csz + 4 bytes), so a
    //      container's header goes out before its blocks; past the end of the buffer the sink only counts (cram_bytes = what the file needs).
    struct Sink {
        uint8_t *p; size_t cap, n;
        void push_back(uint8_t b) { if (n < cap) p[n] = b; n++; }
        void bytes(const uint8_t *d, size_t k) { if (k && n + k <= cap) memcpy(p + n, d, k); n += k; }
        size_t size() const { return n; }
    } o{cram_out, cram_cap, 0};
    auto itf8_len = [](int32_t sv) { const uint32_t v = (uint32_t)sv; return v < 0x80 ? 1u : v < 0x4000 ? 2u : v < 0x200000 ? 3u : v < 0x10000000 ? 4u : 5u; };
    auto block_size = [&](int32_t cid, uint32_t csz, uint32_t usz) { return (uint64_t)2 + itf8_len(cid) + itf8_len((int32_t)csz) + itf8_len((int32_t)usz) + csz + 4; };
    std::vector<uint64_t> crc_from, crc_at;                              // byte ranges [from, at) whose CRC-32 goes to o[at .. at + 4)
    crc_from.reserve(nb + 4 * ns + 8); crc_at.reserve(nb + 4 * ns + 8);
    auto block = [&](int method, int ctype, int32_t cid, const uint8_t *data, uint32_t csz, uint32_t usz) {
        const uint64_t from = o.size();
        o.push_back((uint8_t)method); o.push_back((uint8_t)ctype); put_itf8(o, cid); put_itf8(o, (int32_t)csz); put_itf8(o, (int32_t)usz);
        o.bytes(data, csz);
        crc_from.push_back(from); crc_at.push_back(o.size()); put32le(o, 0);
    };
    auto container = [&](uint64_t body_bytes, int32_t ref, int64_t start, int64_t span, int32_t nrecs, uint64_t counter, uint64_t bases, int32_t nblocks, int32_t landmark) {
        const uint64_t from = o.size();
        put32le(o, (uint32_t)body_bytes);
        put_itf8(o, ref); put_itf8(o, (int32_t)start); put_itf8(o, (int32_t)span); put_itf8(o, nrecs); put_ltf8(o, counter); put_ltf8(o, bases); put_itf8(o, nblocks);
        put_itf8(o, 1); put_itf8(o, landmark);
        crc_from.push_back(from); crc_at.push_back(o.size()); put32le(o, 0);
    };
    // file definition: "CRAM", 3.0 / 3.1, 20-byte file id
    if (!W) { const uint8_t def[6] = {'C', 'R', 'A', 'M', 3, (uint8_t)(v31 ? 1 : 0)}; o.bytes(def, 6); const char id[20] = "htslib_amd"; o.bytes((const uint8_t *)id, 20); }
    if (!W) {   // header container: one FILE_HEADER block = int32 text length + text (cram_write_SAM_hdr)
        std::vector<uint8_t> h; put32le(h, (uint32_t)text.size()); h.insert(h.end(), text.begin(), text.end());
        container(block_size(0, (uint32_t)h.size(), (uint32_t)h.size()), 0, 0, 0, 0, 0, 0, 1, 0);
        block(0, 0, 0, h.data(), (uint32_t)h.size(), (uint32_t)h.size());
    }
    for (size_t k = 0; k < ns; k++) {
        const SliceParts &P = parts[k];
        const uint64_t comp_bytes = block_size(0, P.comp_len, P.comp_len);
        uint64_t body = comp_bytes + block_size(0, P.sh_len, P.sh_len) + block_size(0, 0, 0);
        for (size_t i = P.b0; i < P.b1; i++) body += block_size(blks[i].cid, cmeth[i] == 0 ? blks[i].n : clen[i], blks[i].n);
        // (the slice header's block count includes the CORE block: the encoder counted blocks + 1 already)
        container(body, P.hdr.ref_seq_id, P.hdr.ref_seq_start, P.hdr.ref_seq_span, P.hdr.nrec, counter0 + (uint64_t)k * records_per_slice, P.bases, (int32_t)(3 + (P.b1 - P.b0)), (int32_t)comp_bytes);
        block(0, 1, 0, P.comp, P.comp_len, P.comp_len);                    // compression header
        block(0, 2, 0, P.sh, P.sh_len, P.sh_len);                          // slice header (MAPPED_SLICE); the landmark points here
        { const uint8_t none = 0; block(0, 5, 0, &none, 0, 0); }           // the CORE block: empty (every series is EXTERNAL)
        for (size_t i = P.b0; i < P.b1; i++) block(cmeth[i], 4, blks[i].cid, cmeth[i] == 0 ? blks[i].p : cdata + coff[i], cmeth[i] == 0 ? blks[i].n : clen[i], blks[i].n);
    }
    static const uint8_t eof3[38] = {0x0f, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0x0f, 0xe0, 0x45, 0x4f, 0x46, 0x00, 0x00, 0x00, 0x00, 0x01, 0x00, 0x05, 0xbd, 0xd9, 0x4f, 0x00, 0x01, 0x00, 0x06, 0x06,
                                     0x01, 0x00, 0x01, 0x00, 0x01, 0x00, 0xee, 0x63, 0x01, 0x4b};      // cram_write_eof_block, CRAM 3 (cram_io.c:4320-4370)
    if (!W) o.bytes(eof3, 38);
    *cram_bytes = o.size();
    if (o.size() > cram_cap) return HG_ENOMEM;
    lap(2);
    {   // checksums, one batched device call (block CRCs cover header + payload, container CRCs the container header, cram_io.c:1547-1552, 3990-4010)
        const size_t nc = crc_from.size();
        std::vector<const uint8_t *> bp(nc); std::vector<uint32_t> bl(nc), crc(nc);
        for (size_t i = 0; i < nc; i++) { bp[i] = cram_out + crc_from[i]; bl[i] = (uint32_t)(crc_at[i] - crc_from[i]); }
        if (nc && (rc = hg_crc32_batch_host(ctx, bp.data(), bl.data(), nc, crc.data())) != HG_OK) return rc;
        for (size_t i = 0; i < nc; i++) for (int b = 0; b < 4; b++) cram_out[crc_at[i] + (size_t)b] = (uint8_t)(crc[i] >> (8 * b));
    }
    lap(3);
    if (stats) fprintf(stderr, "[hts-gpu] bam_to_cram: %zu records, %zu slices, %zu blocks: header walk + record encoder %.1f ms, block auto-tuner %.1f ms, framing %.1f ms, CRC-32s + copy out %.1f ms\n",
                       nrec, ns, nb, t_stage[0], t_stage[1], t_stage[2], t_stage[3]);
    if (W) W->counter += nrec;
    return HG_OK;
}

// ---- cram_index_build (reference cram/cram_index.c:779-870 over cram_index_container / cram_index_slice / cram_index_build_multiref :632-760): the .crai
//      text of a whole CRAM file -- one line per slice ("ref start span container_offset slice_offset slice_bytes"), or one per run of records on the
//      same reference for a multi-reference slice.  The container / block walk is host code (a few bytes per block); only multi-reference slices are
//      decoded: their blocks through cram_uncompress_block and their records through the record decoder (ref_id / apos / aend columns) in one batch
//      each.  The reference writes the text through bgzf_open(fn, "wg") (gzip); this returns the text, the caller compresses it (hg_gzip_deflate_host).
extern "C" long hg_cram_index_build_host(hg_ctx *ctx, const uint8_t *cram, size_t cram_len, char *out, size_t cap) {
    if (!ctx || !cram || !out) return HG_EINVAL;
    if (cram_len < 26 || memcmp(cram, "CRAM", 4) != 0) return HG_EINVAL;
    const int major = cram[4];
    if (major != 2 && major != 3) return HG_BLOCK_EUNSUPPORTED;
    struct SliceAt { int64_t cpos; int32_t landmark, bytes; size_t hdr; std::vector<size_t> body; size_t comp; };
    std::vector<Blk> blocks; std::vector<SliceAt> slices;
    hgr::Cursor c{cram + 26, cram + cram_len};
    bool first = true;
    while (c.p < c.end) {
        const int64_t cpos = c.p - cram;
        if (c.end - c.p < 4) return HG_EINVAL;
        const uint32_t clen = (uint32_t)c.p[0] | (uint32_t)c.p[1] << 8 | (uint32_t)c.p[2] << 16 | (uint32_t)c.p[3] << 24; c.p += 4;
        (void)c.itf8(); (void)c.itf8(); (void)c.itf8(); (void)c.itf8();
        if (major >= 3) (void)c.ltf8(); else (void)c.itf8();
        (void)c.ltf8();
        const int32_t nblk = c.itf8(), nland = c.itf8();
        for (int32_t i = 0; i < nland; i++) (void)c.itf8();
        if (major >= 3) c.p += 4;
        if (c.bad || nblk < 0 || c.p > c.end || (size_t)(c.end - c.p) < clen) return HG_EINVAL;
        const uint8_t *region = c.p, *cend = c.p + clen;
        hgr::Cursor b{c.p, cend};
        size_t comp = (size_t)-1; bool in_slice = false;
        for (int32_t k = 0; k < nblk && b.p < b.end && !first; k++) {      // (the first container holds the SAM header)
            const uint8_t *h0 = b.p;
            Blk x; memset(&x, 0, sizeof x);
            x.method = b.byte(); x.ctype = b.byte(); x.cid = b.itf8(); x.csz = (uint32_t)b.itf8(); x.usz = (uint32_t)b.itf8();
            if (b.bad || (size_t)(b.end - b.p) < (size_t)x.csz + (major >= 3 ? 4u : 0u)) return HG_EINVAL;
            x.crc_part = crc32_small(h0, (size_t)(b.p - h0));
            x.data = b.p; b.p += x.csz;
            if (major >= 3) { x.crc = (uint32_t)b.p[0] | (uint32_t)b.p[1] << 8 | (uint32_t)b.p[2] << 16 | (uint32_t)b.p[3] << 24; b.p += 4; }
            blocks.push_back(x);
            const size_t me = blocks.size() - 1;
            const int32_t nbytes = (int32_t)(b.p - h0);
            if (x.ctype == 1) comp = me;
            else if (x.ctype == 2 || x.ctype == 3) { slices.push_back(SliceAt{cpos, (int32_t)(h0 - region), nbytes, me, {}, comp}); in_slice = true; }
            else if ((x.ctype == 4 || x.ctype == 5) && in_slice) { slices.back().body.push_back(me); slices.back().bytes += nbytes; }
        }
        first = false;
        c.p = cend;
    }
    // slice headers are small RAW / gzip blocks: decode them (and, for multi-reference slices, everything they need) in one batch
    std::vector<size_t> want;
    for (auto &s : slices) { want.push_back(s.hdr); }
    std::vector<std::vector<uint8_t>> dec(blocks.size());
    auto decode_blocks = [&](const std::vector<size_t> &idx) -> int {
        const size_t n = idx.size();
        if (!n) return HG_OK;
        std::vector<int32_t> method(n), status(n); std::vector<const uint8_t *> in(n); std::vector<uint32_t> il(n), ol(n), part(n), crc(n); std::vector<uint8_t *> o(n);
        for (size_t i = 0; i < n; i++) {
            const Blk &x = blocks[idx[i]];
            dec[idx[i]].resize(x.usz ? x.usz : 1);
            method[i] = x.method; in[i] = x.data; il[i] = x.csz; ol[i] = x.usz; o[i] = dec[idx[i]].data(); part[i] = x.crc_part; crc[i] = x.crc;
        }
        return major >= 3 ? hg_cram_uncompress_blocks_crc_host(ctx, n, method.data(), in.data(), il.data(), part.data(), crc.data(), o.data(), ol.data(), status.data())
                          : hg_cram_uncompress_blocks_host(ctx, n, method.data(), in.data(), il.data(), o.data(), ol.data(), status.data());
    };
    int rc = decode_blocks(want);
    if (rc != HG_OK) return rc;
    std::vector<hgr::SliceHeader> sh(slices.size());
    std::vector<size_t> multi;
    want.clear();
    for (size_t i = 0; i < slices.size(); i++) {
        if (hgr::parse_slice_header(dec[slices[i].hdr].data(), blocks[slices[i].hdr].usz, major, sh[i])) return HG_EINVAL;
        if (sh[i].ref_seq_id == -2) {
            multi.push_back(i);
            if (slices[i].comp == (size_t)-1) return HG_EINVAL;
            want.push_back(slices[i].comp);
            for (size_t k : slices[i].body) want.push_back(k);
        }
    }
    std::sort(want.begin(), want.end()); want.erase(std::unique(want.begin(), want.end()), want.end());
    if ((rc = decode_blocks(want)) != HG_OK) return rc;
    // records of the multi-reference slices: ref_id / apos / aend
    std::vector<int32_t> ref_id; std::vector<int64_t> apos, aend; std::vector<uint64_t> rec_off(multi.size() + 1, 0);
    if (!multi.empty()) {
        const size_t nm = multi.size();
        std::vector<hg_cram_slice_blocks> sb(nm);
        std::vector<std::vector<int32_t>> ids(nm); std::vector<std::vector<const uint8_t *>> ptr(nm); std::vector<std::vector<uint32_t>> len(nm);
        for (size_t j = 0; j < nm; j++) {
            const SliceAt &s = slices[multi[j]];
            memset(&sb[j], 0, sizeof sb[j]);
            sb[j].comp_hdr = dec[s.comp].data(); sb[j].comp_hdr_len = blocks[s.comp].usz; sb[j].slice_hdr = dec[s.hdr].data(); sb[j].slice_hdr_len = blocks[s.hdr].usz;
            size_t taken = 0;
            for (size_t k : s.body) {
                if ((int32_t)taken >= sh[multi[j]].nblocks) break;
                taken++;
                if (blocks[k].ctype == 5) { sb[j].core = dec[k].data(); sb[j].core_len = blocks[k].usz; continue; }
                ids[j].push_back(blocks[k].cid); ptr[j].push_back(dec[k].data()); len[j].push_back(blocks[k].usz);
            }
            sb[j].nblocks = (uint32_t)ids[j].size(); sb[j].content_id = ids[j].data(); sb[j].data = ptr[j].data(); sb[j].len = len[j].data(); sb[j].decode_md = 0;
        }
        uint64_t nrec = 0, cc = 0, nc = 0, ac = 0;
        if ((rc = hg_cram_records_bound(nm, sb.data(), major, &nrec, &cc, &nc, &ac)) != HG_OK) return rc;
        ref_id.resize(nrec + 1); apos.resize(nrec + 1); aend.resize(nrec + 1);
        hg_cram_record_cols cols; memset(&cols, 0, sizeof cols);
        cols.ref_id = ref_id.data(); cols.apos = apos.data(); cols.aend = aend.data();
        std::vector<int32_t> st(nm);
        rc = hg_cram_decode_records_host(ctx, nm, sb.data(), major, 0x7fffffff, (size_t)nrec + 1, (size_t)-1, (size_t)-1, 0, 0, &cols, rec_off.data(), st.data(), nullptr);
        if (rc != HG_OK) return rc;
    }
    // the lines, in file order; "CRAM file is not sorted by chromosome / position" like the reference (per container there, per slice here)
    size_t n = 0, mj = 0;
    int64_t last_ref = -9, last_start = -9;
    for (size_t i = 0; i < slices.size(); i++) {
        const SliceAt &s = slices[i];
        if (sh[i].ref_seq_id == last_ref && sh[i].ref_seq_start < last_start) return -2;
        last_ref = sh[i].ref_seq_id; last_start = sh[i].ref_seq_start;
        const bool is_multi = sh[i].ref_seq_id == -2;
        const uint64_t r0 = is_multi ? rec_off[mj] : 0;
        const long got = hg_cram_crai_slice(dec[s.hdr].data(), blocks[s.hdr].usz, major, is_multi ? ref_id.data() + r0 : nullptr, is_multi ? apos.data() + r0 : nullptr,
                                            is_multi ? aend.data() + r0 : nullptr, s.cpos, s.landmark, s.bytes, out + n, cap - n);
        if (is_multi) mj++;
        if (got < 0) return got;
        n += (size_t)got;
    }
    return (long)n;
}

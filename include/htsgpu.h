/*
 * htsgpu.h -- C ABI of the MI355X (gfx950) block-codec engine.
 *
 * This is the drop-in boundary for htslib's block-codec hot path.  There is no
 * codec plugin API in htslib (the only plugin interface is hFILE transports,
 * hfile_internal.h:65-154), so the boundary is a plain C library that a libhts
 * build links and calls from the places listed next to each entry point.  All
 * signatures are extern "C", plain pointers and sizes; no torch / C++ types.
 *
 * Unit of work = a BATCH of independent blocks (BGZF blocks or CRAM
 * data-series buffers).  One block is decoded/encoded by one 64-lane wavefront.
 *
 * Conventions
 *   - "dev" entry points take DEVICE pointers, are asynchronous on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) and never
 *     synchronise.  "host" entry points take host pointers and are synchronous.
 *   - per-block status codes mirror bgzf_uncompress (bgzf.c:730-804):
 *        0 ok, -1 inflate/format failure, -2 CRC mismatch.
 *   - functions return 0 on success or a negative HG_E* code.
 */
#ifndef HTSGPU_H
#define HTSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_OK            0
#define HG_EINVAL       -1   /* bad argument                                  */
#define HG_ENODEV       -2   /* no usable gfx950 device / HIP runtime failure */
#define HG_ENOMEM       -3
#define HG_EFORMAT      -4   /* framing error found by a host-side scan       */
#define HG_ELAUNCH      -5   /* kernel launch / execution failure             */
#define HG_EBLOCK       -6   /* at least one block of the batch failed        */

/* per-block status, identical meaning to bgzf_uncompress()'s return value */
#define HG_BLOCK_OK      0
#define HG_BLOCK_EINFLATE (-1)
#define HG_BLOCK_ECRC    (-2)

#define HG_BGZF_BLOCK_SIZE     0xff00   /* htslib/bgzf.h:50 */
#define HG_BGZF_MAX_BLOCK_SIZE 0x10000  /* htslib/bgzf.h:51 */

typedef struct hg_ctx hg_ctx;

/* One descriptor per BGZF block.  Built by the host framing scan
 * (replaces the header walk of bgzf_mt_read_block, bgzf.c:1485-1539). */
typedef struct hg_bgzf_desc {
    uint64_t coff;   /* byte offset of the block's first header byte          */
    uint64_t uoff;   /* byte offset of the block's payload in the plain image */
    uint32_t clen;   /* BSIZE+1: whole block incl. 18 B header + 8 B trailer  */
    uint32_t ulen;   /* ISIZE                                                  */
} hg_bgzf_desc;

/* ---- lifetime ---------------------------------------------------------- */
int  hg_init(int device, hg_ctx **ctx);      /* binds ctx to HIP device `device` */
void hg_destroy(hg_ctx *ctx);
const char *hg_strerror(int code);
const char *hg_version(void);
/* number of compute units / resident wavefronts the engine will launch */
int  hg_device_info(hg_ctx *ctx, int *cus, int *waves_per_launch);

/* ---- BGZF framing (host) ----------------------------------------------- */
/* Walk `len` bytes of a BGZF stream (check_header, bgzf.c:896-903) and fill
 * up to `max_desc` descriptors.  Returns the number of blocks found
 * (may exceed max_desc: call again with a larger array) or HG_EFORMAT.
 * *total_ulen receives the sum of ISIZE over all blocks. */
long hg_bgzf_scan(const uint8_t *buf, size_t len,
                  hg_bgzf_desc *desc, size_t max_desc, uint64_t *total_ulen);

/* ---- BGZF inflate (replaces bgzf_uncompress / bgzf_decode_func,
 *      bgzf.c:730-804, 1373-1384) ----------------------------------------- */
/* d_comp: the compressed stream image in HBM (comp_len bytes, any alignment
 *         >= 4); d_desc: nblocks descriptors in HBM; d_out: plain image
 *         (out_cap bytes); d_status: nblocks int32 (HG_BLOCK_*).
 * Each block's raw deflate payload is decoded, its length checked against
 * ISIZE and its CRC-32 against the trailer, all inside one kernel. */
int hg_bgzf_inflate_dev(hg_ctx *ctx,
                        const void *d_comp, size_t comp_len,
                        const hg_bgzf_desc *d_desc, size_t nblocks,
                        void *d_out, size_t out_cap,
                        int32_t *d_status, void *stream);

/* Host-buffer convenience: scan + H2D + inflate + D2H, synchronous.
 * status may be NULL.  Returns 0, or HG_EBLOCK if any block failed (the first
 * failing block's code is stored in *first_bad_code, its index in
 * *first_bad_idx when non-NULL). */
int hg_bgzf_inflate_host(hg_ctx *ctx,
                         const uint8_t *comp, size_t comp_len,
                         uint8_t *out, size_t out_cap, size_t *out_len,
                         int32_t *status, size_t max_status,
                         long *first_bad_idx, int *first_bad_code);

/* Same kernel for generic gzip members of any length (RFC 1952; 32 KiB window, several deflate
 * blocks): the CRAM block method GZIP path, zlib_mem_inflate (cram/cram_io.c:1068-1110, called from
 * cram_uncompress_block :1605-1624).  desc[i]: coff/clen = the member, uoff/ulen = where its
 * plaintext goes and the size the CRAM block header promises (checked, with ISIZE and CRC-32). */
int hg_gzip_inflate_dev(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc,
                        size_t nmembers, void *d_out, size_t out_cap, int32_t *d_status, void *stream);

/* ---- BGZF deflate (replaces bgzf_compress / bgzf_encode_func / bgzf_encode_level0_func,
 *      bgzf.c:561-683, 1330-1368) ------------------------------------------ */
/* d_plain: the uncompressed image in HBM; d_desc[i].uoff/.ulen = the bytes of block i
 * (ulen <= 0xff00, as cut by bgzf_write / bgzf_flush_try, bgzf.c:1996-2024) and d_desc[i].coff =
 * byte offset of that block's 64 KiB output slot inside d_slots (normally i * 65536).  Every block
 * becomes one complete BGZF block (18 B header, raw deflate or stored payload, CRC32, ISIZE) at
 * the start of its slot; its length (BSIZE+1) is written to d_clen[i].  ulen == 0 yields the
 * 28-byte EOF marker block.  level 0 = stored blocks (bgzf_encode_level0_func), 1..9 = deflate. */
int hg_bgzf_deflate_dev(hg_ctx *ctx, const void *d_plain, const hg_bgzf_desc *d_desc, size_t nblocks,
                        int level, void *d_slots, uint32_t *d_clen, void *stream);

/* Gather the slots into one contiguous BGZF stream (what bgzf_mt_writer hwrite()s in order,
 * bgzf.c:1398-1473).  d_packed_off (nblocks u64) receives each block's offset in the stream (the
 * block_address needed by the .gzi / BAI index code, bgzf.c:228-290), *d_total the stream length.
 * add_eof != 0 appends the EOF marker block (bgzf_close, bgzf.c:2084-2101). */
int hg_bgzf_pack_dev(hg_ctx *ctx, const void *d_slots, const hg_bgzf_desc *d_desc, const uint32_t *d_clen,
                     size_t nblocks, void *d_packed, size_t packed_cap, uint64_t *d_packed_off,
                     uint64_t *d_total, int add_eof, void *stream);

/* Host-buffer convenience: cut `plain` into blocks (at cuts[0..ncuts] if given -- cuts[0] = 0,
 * cuts[ncuts] = len, every piece <= 0xff00 -- otherwise every 0xff00 bytes), H2D, deflate, pack,
 * D2H; appends the EOF block when add_eof != 0.  Synchronous. */
int hg_bgzf_deflate_host(hg_ctx *ctx, const uint8_t *plain, size_t len, const uint64_t *cuts, size_t ncuts,
                         int level, int add_eof, uint8_t *out, size_t out_cap, size_t *out_len);

/* ---- asynchronous host-buffer jobs ("pipes"): the batching engine behind bgzf_read / bgzf_write.
 *      Replaces the I/O-thread + pool + ordered-result-queue pipeline of bgzf_mt_reader / bgzf_mt_writer
 *      (bgzf.c:1598-1738, 1398-1473; hts_tpool ordered results thread_pool.c:149-252): a pipe owns pinned host
 *      buffers, device buffers and a HIP stream; a submitted job (H2D -> kernels -> D2H) runs asynchronously and
 *      the caller collects results in submission order.  Two or three pipes per stream overlap file I/O, PCIe and
 *      the kernels.  A pipe carries ONE job at a time; different pipes may be driven from different threads. ---- */
typedef struct hg_pipe hg_pipe;
int  hg_pipe_create(hg_ctx *ctx, hg_pipe **pipe);
void hg_pipe_destroy(hg_pipe *pipe);
/* Pinned staging buffer for the NEXT job's input, at least `bytes` long (contents are lost when it grows).
 * NULL while a job is in flight or on allocation failure. */
void *hg_pipe_input(hg_pipe *pipe, size_t bytes);
/* Size the pipe's buffers for jobs of up to in_bytes of input and out_bytes of output in one step (pinning host memory costs
 * ~0.3 ms per MiB, so a reader that knows where its windows are heading asks once instead of growing job by job).  Only
 * between jobs; contents of the buffers are lost. */
int hg_pipe_reserve(hg_pipe *pipe, size_t in_bytes, size_t out_bytes);
/* Inflate job: the first comp_len bytes of the input buffer are n whole BGZF blocks; desc[i].coff is relative to
 * the buffer, desc[i].uoff cumulative from 0.  Returns after queueing the work. */
int hg_pipe_inflate(hg_pipe *pipe, size_t comp_len, const hg_bgzf_desc *desc, size_t n);
/* Deflate job: the input buffer holds len plain bytes cut into n blocks at cuts[0..n] (cuts[0] = 0, cuts[n] = len,
 * pieces <= 0xff00).  raw = 0: complete BGZF blocks back to back.  raw != 0: bare byte-aligned deflate chunks
 * without BFINAL that concatenate into one deflate stream (bgzf mode "g", bgzf.c:686-706); crc[i] = CRC-32 of chunk i. */
int hg_pipe_deflate(hg_pipe *pipe, size_t len, const uint64_t *cuts, size_t n, int level, int raw);
/* Wait for the job.  Inflate: *out / *out_len = the plain image (pinned host memory owned by the pipe, valid until
 * the next job), status[i] = HG_BLOCK_*; returns HG_OK or HG_EBLOCK.  Deflate: *out = the compressed stream,
 * blk_off[0..n] = offset of every block in it (blk_off[n] = *out_len), crc[i] as above.  Unused outputs may be NULL. */
int hg_pipe_wait(hg_pipe *pipe, const uint8_t **out, size_t *out_len, const int32_t **status, const uint64_t **blk_off,
                 const uint32_t **crc);

/* ---- plain gzip streams (not BGZF): the fallback of bgzf_read_block for ordinary .gz input
 *      (inflate_gzip_block, bgzf.c:826-893, 1165-1196).  One deflate stream is one dependency chain, so this runs
 *      on a single wavefront; it exists so that the drop-in reads what stock htslib reads.  Resumable at deflate-block
 *      boundaries: each call decodes whole deflate blocks from state->in_bit until `soft_cap` bytes have been made,
 *      the member ends, or the input runs out. ---- */
typedef struct hg_gz_state {
    uint64_t in_bit;       /* next bit to decode, relative to comp[0]                                   */
    uint32_t in_member;    /* 0: comp + in_bit/8 is a member header; 1: inside a member's deflate data   */
    uint32_t crc;          /* CRC-32 of the current member's output so far                              */
    uint32_t isize;        /* bytes of the current member so far (mod 2^32)                             */
    uint32_t reserved;
} hg_gz_state;
#define HG_GZ_MORE     1   /* soft_cap reached at a deflate-block boundary: call again                  */
#define HG_GZ_MEMBER   2   /* a member ended and its CRC-32 / ISIZE trailer matched; in_bit is past it  */
#define HG_GZ_NEEDIN   3   /* the next deflate block is not complete in comp[0..comp_len): supply more   */
/* hist/hist_len: the last <= 32768 bytes produced before this call (the LZ77 window), NULL at a member start.
 * out receives the new bytes (*out_len of them, <= out_cap; out_cap should exceed soft_cap by the size of one
 * deflate block's output).  comp_eof != 0: no more input exists (an incomplete block is then an error).
 * Returns HG_GZ_* (> 0), or a negative HG_E* / HG_EBLOCK on a corrupt stream. */
int hg_gzip_stream_inflate_host(hg_ctx *ctx, const uint8_t *comp, size_t comp_len, int comp_eof, hg_gz_state *state,
                                const uint8_t *hist, size_t hist_len, uint8_t *out, size_t out_cap, size_t soft_cap,
                                size_t *out_len);

/* ---- CRAM 3.0 rANS 4x8 (replaces rans_uncompress as called by cram_uncompress_block,
 *      cram/cram_io.c:1666-1683; CRAM block method 4) ------------------------------ */
/* One descriptor per entropy-coded stream (= the payload of one CRAM block). */
typedef struct hg_stream_desc {
    uint64_t in_off;        /* byte offset of the stream in the input image              */
    uint64_t out_off;       /* byte offset of its plaintext in the output image          */
    uint32_t in_len;        /* compressed bytes (cram_block.comp_size)                   */
    uint32_t out_len;       /* expected plaintext bytes (cram_block.uncomp_size)         */
    uint32_t scratch_off;   /* offset (32-bit words) of this stream's table scratch      */
    uint32_t reserved;
} hg_stream_desc;

/* words of table scratch a stream of in_len bytes may need (order-1 context tables) */
#define HG_RANS4X8_SCRATCH_WORDS(in_len) (1024u + 2u * (uint32_t)(in_len))

/* status[i] = 0 ok, -1 malformed stream / size mismatch (cram_uncompress_block returns -1). */
int hg_rans4x8_decode_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t nstreams,
                          void *d_out, int32_t *d_status, uint32_t *d_scratch, void *stream);

/* Host convenience for n streams: in[i]/in_len[i] -> out[i] (capacity out_cap[i], actual size in
 * out_len[i] = the size stored in the stream header).  Synchronous.  Returns 0 or HG_EBLOCK. */
int hg_rans4x8_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                           uint8_t *const *out, const uint32_t *out_cap, uint32_t *out_len, int32_t *status);

/* Encoder (replaces rans_compress as called by cram_compress_by_method, cram/cram_io.c:1834-1848).
 * order[i] = 0 | 1.  out[i] must hold hg_rans4x8_compress_bound(in_len[i]) bytes.  Synchronous. */
size_t hg_rans4x8_compress_bound(size_t in_len);
int hg_rans4x8_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *order,
                           size_t n, uint8_t *const *out, uint32_t *out_len);

/* ---- CRAM 3.1 rANS Nx16 (replaces rans_uncompress_4x16 as called by cram_uncompress_block,
 *      cram/cram_io.c:1697-1714; CRAM block method 5).  PARITY UNPINNED: htscodecs is absent from
 *      the reference and no stock stream exists to check against (oracle/ransnx16_oracle.c). ---- */
/* Decode n streams; out_len[i] = expected plaintext size (cram_block.uncomp_size, also the size
 * for NOSZ streams).  Handles every flag: order 0/1, 4-way and 32-way (X32), NOSZ, CAT, and the PACK /
 * RLE / STRIPE transforms (headers planned on the host, all payload work in the gfx950 kernels
 * ransnx16.hip + ransnx16_xform.hip).  Synchronous.  Returns 0 or HG_EBLOCK. */
int hg_ransnx16_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                            uint8_t *const *out, const uint32_t *out_len, int32_t *status);
/* device form (entropy core only): d_sel4 / d_sel32 list the descriptor indices of the 4-way / 32-way
 * streams; a stream whose flag byte carries PACK, RLE or STRIPE gets status -3 here -- use the host form,
 * which plans those transforms. */
int hg_ransnx16_decode_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4,
                           size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out, int32_t *d_status,
                           uint32_t *d_scratch, void *stream);

/* Encoder (replaces rans_compress_4x16 as called by cram_compress_by_method,
 * cram/cram_io.c:1853-1866).  flags[i]: bit 0x01 order-1, 0x04 32-way, 0x08 STRIPE (4 stripes), 0x10 NOSZ,
 * 0x20 CAT, 0x40 RLE, 0x80 PACK (the RANS_ORDER_* bits of cram/cram_external.c:616-624), i.e. every
 * RANS_PR* set of cram_io.c:1856.  PACK / RLE are dropped from the stream's flag byte when they do not
 * apply (more than 16 distinct symbols / no symbol worth run-length coding).
 * out[i] must hold hg_ransnx16_compress_bound(in_len[i]) bytes.  Synchronous. */
size_t hg_ransnx16_compress_bound(size_t in_len);
int hg_ransnx16_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *flags,
                            size_t n, uint8_t *const *out, uint32_t *out_len);

/* ---- CRAM 3.1 adaptive arithmetic ("range") coder (replaces arith_uncompress_to / arith_compress_to as
 *      called at cram/cram_io.c:1716-1733 and 1869-1883; CRAM block method 6).  PARITY UNPINNED like Nx16
 *      (oracle/arith_oracle.c).  flags[i] = the stream's first byte: 0x01 order-1, 0x08 STRIPE, 0x10 NOSZ,
 *      0x20 CAT, 0x40 RLE (run-length models), 0x80 PACK (cram/cram_external.c:628-638).  Streams whose
 *      payload is bzip2 (EXT 0x04) get status -3.  One wavefront per stream; see arith.hip. ---- */
int hg_arith_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                         uint8_t *const *out, const uint32_t *out_len, int32_t *status);
size_t hg_arith_compress_bound(size_t in_len);
int hg_arith_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *flags,
                         size_t n, uint8_t *const *out, uint32_t *out_len);

/* ---- CRAM 3.1 read-name tokeniser "tok3" (replaces tok3_decode_names / tok3_encode_names as called at
 *      cram/cram_io.c:1735-1749 and 1885-1895; CRAM block method 8).  PARITY UNPINNED (oracle/tok3_oracle.c).
 *      Names are NUL-terminated in the plain buffer (NEWS:278).  The token byte streams inside a block are
 *      rANS Nx16 or range-coder streams and go through those kernels; tok3.hip rebuilds / tokenises the
 *      names, one wavefront per block. ---- */
int hg_tok3_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                        uint8_t *const *out, const uint32_t *out_len, int32_t *status);
/* use_arith[i]: 0 = rANS Nx16 back-end, 1 = range coder (the use_arith argument of tok3_encode_names).  A buffer that
 * is not a list of NUL-terminated names gets out_len[i] = 0 (the caller keeps another method).  out[i] must hold
 * hg_tok3_compress_bound(in_len[i]).  Synchronous. */
size_t hg_tok3_compress_bound(size_t in_len);
int hg_tok3_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *use_arith,
                        size_t n, uint8_t *const *out, uint32_t *out_len);

/* ---- CRAM 3.1 fqzcomp quality codec (replaces fqz_decompress as called at cram/cram_io.c:1684-1695; CRAM block
 *      method 7).  PARITY UNPINNED (oracle/fqzcomp_oracle.c).  The stream carries its own record lengths, so the
 *      call needs nothing but the block (the reference passes lengths = NULL as well).  out_len[i] = the block's
 *      uncompressed size; a stream that stores a different size gets status -1, one with more than two parameter
 *      sets HG_BLOCK_EUNSUPPORTED.  One wavefront per stream, 65536 adaptive models per wavefront in device
 *      scratch (up to 67 MB each: the number of resident streams is sized to the free memory); see fqzcomp.hip. ---- */
int hg_fqz_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n,
                       uint8_t *const *out, const uint32_t *out_len, int32_t *status);
/* Encoder (replaces fqz_compress as called by cram_compress_by_method, cram/cram_io.c:1801-1825).  hg_fqz_slice is the
 * reference's fqz_slice as cram_io.c:1808-1820 fills it: the records' quality lengths and BAM flags (16 = reverse strand,
 * 128 = second read; flags may be NULL).  strat[i] = 0..3 for the methods FQZ, FQZ_b, FQZ_c, FQZ_d (cram_io.c:2065-2068).
 * The parameter block is chosen on the host (modelled on the four htscodecs presets; any choice is valid by the format),
 * the adaptive coding runs one wavefront per block.  A block whose record lengths do not add up to in_len[i] gets
 * out_len[i] = 0.  out[i] must hold hg_fqz_compress_bound(in_len[i], num_records).  Synchronous. */
#ifndef HG_FQZ_SLICE_DEFINED
#define HG_FQZ_SLICE_DEFINED
typedef struct hg_fqz_slice { uint32_t num_records; const uint32_t *len; const uint32_t *flags; } hg_fqz_slice;
#endif
size_t hg_fqz_compress_bound(size_t in_len, size_t num_records);
int hg_fqz_encode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const hg_fqz_slice *const *slice,
                       const int32_t *strat, size_t n, uint8_t *const *out, uint32_t *out_len);

/* ---- CRAM block layer (replaces cram_uncompress_block, cram/cram_io.c:1576-1754) ------------ */
/* on-disk method ids, htslib/cram.h:84-101 */
#define HG_CRAM_RAW      0
#define HG_CRAM_GZIP     1
#define HG_CRAM_BZIP2    2
#define HG_CRAM_LZMA     3
#define HG_CRAM_RANS4x8  4
#define HG_CRAM_RANSNx16 5
#define HG_CRAM_ARITH    6
#define HG_CRAM_FQZ      7
#define HG_CRAM_TOK3     8
#define HG_CRAM_MASK_TOKA (1u << 9) /* method_mask bit for hg_cram_compress_blocks_host only */
#define HG_BLOCK_EUNSUPPORTED (-3)   /* method not implemented by the engine (yet): caller keeps its CPU codec */

/* Uncompress n CRAM blocks in one batch: block i has on-disk method method[i], compressed payload
 * in[i] (in_len[i] = comp_size) and must produce exactly out_len[i] = uncomp_size bytes into out[i]
 * (cram_uncompress_block's size check, cram_io.c:1611-1614).  RAW blocks are copied, GZIP, RANS4x8,
 * RANSNx16, ARITH, FQZ and TOK3 blocks go to the gfx950 kernels, BZIP2 / LZMA blocks to the system libraries the reference itself delegates them to (looked up at run
 * time; absent = -3); status[i] = 0 / -1 / -2 (CRC) / -3 (unsupported method).
 * Synchronous; returns 0, or HG_EBLOCK if any status is non-zero. */
int hg_cram_uncompress_blocks_host(hg_ctx *ctx, size_t n, const int32_t *method, const uint8_t *const *in,
                                   const uint32_t *in_len, uint8_t *const *out, const uint32_t *out_len,
                                   int32_t *status);

/* Same, preceded by cram_uncompress_block's CRC check (cram_io.c:1585-1592): crc_part[i] = CRC-32 of block i's header
 * bytes as cram_read_block leaves it in b->crc_part, crc32[i] = the CRC stored after the payload.  The payload CRCs
 * are computed on the device; a block whose CRC does not match gets status -1 ("Block CRC32 failure") and is not
 * decoded. */
int hg_cram_uncompress_blocks_crc_host(hg_ctx *ctx, size_t n, const int32_t *method, const uint8_t *const *in,
                                       const uint32_t *in_len, const uint32_t *crc_part, const uint32_t *crc32,
                                       uint8_t *const *out, const uint32_t *out_len, int32_t *status);

/* gzip-wrapped deflate of whole buffers (CRAM block method GZIP on the write side: zlib_mem_deflate /
 * libdeflate_deflate, cram/cram_io.c:1113-1148,1222-1277).  Each buffer becomes ONE gzip member: it is
 * deflated in 0xff00-byte chunks by the BGZF deflate kernel (matches do not cross chunks, chunks are
 * joined with empty stored blocks).  out[i] must hold hg_gzip_compress_bound(in_len[i]).  Synchronous. */
size_t hg_gzip_compress_bound(size_t in_len);
int hg_gzip_deflate_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n, int level,
                         uint8_t *const *out, uint32_t *out_len);

/* Batch form of cram_compress_block's trial phase (cram/cram_io.c:1912-2325): block i is compressed
 * with EVERY method whose bit is set in method_mask[i] and the smallest result is kept; RAW is kept
 * when nothing beats it (cram_io.c:2001,2271-2278).  Bits: 1<<HG_CRAM_GZIP, 1<<HG_CRAM_RANS4x8 (orders 0
 * and 1 are both tried), 1<<HG_CRAM_RANSNx16 (flag sets {0,1} / +{64,9,128,193} above level 1 /
 * +{129,192} above level 5 as cram_compress_slice builds them, cram/cram_encode.c:818-826; 32-way for
 * inputs >= 64 KiB like RANS_ORDER_SIMD_AUTO), 1<<HG_CRAM_ARITH (the same flag sets, cram_io.c:1877),
 * 1<<HG_CRAM_TOK3 (name tokeniser over rANS; the reference's TOK3) and HG_CRAM_MASK_TOKA (name tokeniser over the
 * range coder; the reference's TOKA, cram/cram_structs.h:215-266 -- both are written as on-disk method 8).  method_used[i] = on-disk method id of the winner; out[i] must hold
 * hg_cram_compress_bound(in_len[i]).  No cross-slice metrics are kept (every call is a trial). */
size_t hg_cram_compress_bound(size_t in_len);
int hg_cram_compress_blocks_host(hg_ctx *ctx, size_t n, const uint32_t *method_mask, int level,
                                 const uint8_t *const *in, const uint32_t *in_len, uint8_t *const *out,
                                 uint32_t *out_len, int32_t *method_used);

/* ---- cram_compress_block / cram_compress_block2 with the per-data-series method auto-tuner
 *      (cram/cram_io.c:1912-2325).  Method sets use the reference's INTERNAL method ids
 *      (enum cram_block_method_int, cram/cram_structs.h:215-266): bit m of a set = method m. ---- */
#define HG_CRAM_MAX_METHOD 32
enum hg_cram_method_int {
    HG_M_RAW = 0, HG_M_GZIP = 1, HG_M_BZIP2 = 2, HG_M_LZMA = 3, HG_M_RANS0 = 4, HG_M_RANS_PR0 = 5, HG_M_ARITH_PR0 = 6,
    HG_M_FQZ = 7, HG_M_TOK3 = 8, HG_M_GZIP_RLE = 11, HG_M_GZIP_1 = 12, HG_M_FQZ_b = 13, HG_M_FQZ_c = 14, HG_M_FQZ_d = 15,
    HG_M_RANS1 = 16, HG_M_RANS_PR1 = 17, HG_M_RANS_PR64 = 18, HG_M_RANS_PR9 = 19, HG_M_RANS_PR128 = 20, HG_M_RANS_PR129 = 21,
    HG_M_RANS_PR192 = 22, HG_M_RANS_PR193 = 23, HG_M_TOKA = 24, HG_M_ARITH_PR1 = 25, HG_M_ARITH_PR64 = 26, HG_M_ARITH_PR9 = 27,
    HG_M_ARITH_PR128 = 28, HG_M_ARITH_PR129 = 29, HG_M_ARITH_PR192 = 30, HG_M_ARITH_PR193 = 31
};
/* same fields as struct cram_metrics (cram/cram_structs.h:284-305) */
typedef struct hg_cram_metrics {
    int trial, next_trial, consistency;
    int sz[HG_CRAM_MAX_METHOD];
    int input_avg_sz, input_avg_delta;
    int method, revised_method, strat;
    int cnt[HG_CRAM_MAX_METHOD];
    double extra[HG_CRAM_MAX_METHOD];
    int unpackable;
} hg_cram_metrics;
hg_cram_metrics *hg_cram_metrics_new(void);             /* cram_new_metrics, cram_io.c:2327-2339 */
void hg_cram_metrics_free(hg_cram_metrics *m);
/* Block i is compressed under metrics[i] (one object per data series; NULL metrics / NULL entry = plain GZIP as
 * cram_io.c:2282-2299) starting from method_set[i]: while the series is in a trial phase every method of the set is
 * run and the smallest output kept, afterwards only the learnt method; statistics, method costs, retrial spans and
 * the culling of persistently bad methods follow the reference.  Blocks of one call that share a metrics object are
 * handled in their order with the reference's state sequence (the call runs in rounds, split where a trial phase
 * ends).  bzip2 / lzma bits (and the fqzcomp bits of a block without slice information) are dropped from the set, as in an
 * htslib built without those libraries.  method_used[i] = on-disk method id; out[i] must hold in_len[i] bytes: a block that no method
 * shrinks is stored RAW (cram_io.c:2229-2244), so out_len[i] <= in_len[i] always. */
int hg_cram_compress_blocks_metrics_host(hg_ctx *ctx, size_t n, hg_cram_metrics *const *metrics, const uint32_t *method_set,
                                         int level, int version_major, const uint8_t *const *in, const uint32_t *in_len,
                                         uint8_t *const *out, uint32_t *out_len, int32_t *method_used);
/* The same with the cram_slice argument of cram_compress_block2/3 reduced to what the codecs read from it: fqz[i] = the record
 * lengths and flags of block i when it is a quality block (cram_io.c:1808-1820), NULL otherwise; fqz itself may be NULL.
 * Blocks with a slice keep the FQZ / FQZ_b / FQZ_c / FQZ_d bits of their method set (fqzcomp.hip). */
int hg_cram_compress_blocks_metrics_fqz_host(hg_ctx *ctx, size_t n, hg_cram_metrics *const *metrics, const uint32_t *method_set,
                                             int level, int version_major, const uint8_t *const *in, const uint32_t *in_len,
                                             const hg_fqz_slice *const *fqz, uint8_t *const *out, uint32_t *out_len, int32_t *method_used);

/* ---- CRAM record decoding: the record loop of cram_decode_slice (cram/cram_decode.c:2346-3026) for batches of
 *      slices, CRAM 2.x / 3.x.  Input per slice: the DECODED blocks (after cram_uncompress_block) -- the container's compression
 *      header block, the slice header block, the CORE block and the EXTERNAL blocks with their content ids.  All codecs htslib and
 *      htsjdk write are handled (EXTERNAL, HUFFMAN, BETA, GAMMA, SUBEXP, BYTE_ARRAY_LEN, BYTE_ARRAY_STOP); a slice that needs
 *      GOLOMB / GOLOMB_RICE reports HG_BLOCK_EUNSUPPORTED.  Output per record: the cram_record fields that do not need the reference
 *      sequence, after cram_decode_slice_xref -- flags (with the mate bits), cram_flags, ref_id, len, apos, aend, rg, mqual, the
 *      CIGAR (BAM encoding, cram_decode_seq's feature walk), the read name, mate_ref_id, mate_pos, tlen, and -- given the reference
 *      spans -- the bases and qualities (cram_decode_seq's reconstruction), the aux tags as stored (cram_decode_aux) and, with
 *      decode_md, regenerated MD:Z / NM.  Pinned on the reference's 34 CRAM fixtures against their SAM / BAM twins
 *      (tests/test_cram_records.py).  Data-parallel passes where the compression header allows, else one wavefront per slice
 *      (cram_records_fast.hip, cram_records.hip).  Bases / qualities are placed by prefix sums: slice order, record order. ---- */
typedef struct hg_cram_slice_blocks {
    const uint8_t *comp_hdr; uint32_t comp_hdr_len;     /* compression header block of the slice's container (slices of one container may share the pointer) */
    const uint8_t *slice_hdr; uint32_t slice_hdr_len;   /* slice header block */
    const uint8_t *core; uint32_t core_len;             /* CORE block (content type 5) */
    uint32_t nblocks;                                   /* EXTERNAL blocks (content type 4): */
    const int32_t *content_id; const uint8_t *const *data; const uint32_t *len;
    uint32_t nrefs; const struct hg_cram_ref_span *refs; /* reference bases the slice aligns to (only needed for SEQ; may be 0 / NULL) */
    int32_t decode_md;                                  /* fd->decode_md: non-zero = regenerate MD:Z / NM for mapped records that do not store them
                                                           (hts_open sets -1, "auto" = on below CRAM 4); needs refs and the aux output */
} hg_cram_slice_blocks;
/* A stretch of one reference sequence, upper case ASCII: what cram_get_ref hands cram_decode_slice (s->ref, ref_start, ref_end), or the
 * slice's embedded-reference block.  start = 1-based position of bases[0]; sq_len = the @SQ LN of that reference. */
typedef struct hg_cram_ref_span { int32_t ref_id; int64_t start; const uint8_t *bases; uint32_t len; int64_t sq_len; } hg_cram_ref_span;
typedef struct hg_cram_record_cols {                    /* arrays of rec_cap entries; a NULL column is not copied back */
    int32_t *flags, *cram_flags, *ref_id, *len, *rg, *mqual, *mate_ref_id, *ncigar, *name_len;
    int64_t *apos, *aend, *mate_pos, *tlen;
    uint64_t *cigar_off, *name_off;                     /* first CIGAR op / name byte of the record in cigar[] / names[] */
    uint32_t *cigar;                                    /* cigar_cap words: len << 4 | op */
    uint8_t *names;                                     /* name_cap bytes, names are not terminated */
    uint64_t *seq_off; uint8_t *seq, *qual;             /* bases (ASCII, '=' where no reference span was given) and qualities (255 = absent):
                                                           len[r] bytes each at seq_off[r]; all three NULL = not wanted.  seq_cap >= the number
                                                           of bases of the slices (the containers' `bases` header field). */
    uint64_t *aux_off; int32_t *aux_len; uint8_t *aux;  /* the record's tags in BAM encoding (tag[2] type value ...), aux_len[r] bytes at
                                                           aux_off[r]; all three NULL = not wanted.  As stored, plus MD:Z / NM when decode_md asks; RG:Z is not added here. */
} hg_cram_record_cols;
/* For these slices: the number of records (exact) and WORST-CASE sizes of the CIGAR / name / aux arrays (what a slice could produce
 * given the size of its blocks; typical slices need a few per cent of that).  Host only. */
int hg_cram_records_bound(size_t nslices, const hg_cram_slice_blocks *slices, int major_version, uint64_t *nrec, uint64_t *cigar_cap,
                          uint64_t *name_cap, uint64_t *aux_cap);
/* nref = number of @SQ lines (bounds of RI / NS).  rec_off[i] .. rec_off[i+1] = the records of slice i (nslices + 1 entries).
 * status[i] = 0, -1 (malformed slice, as cram_decode_slice returning -1) or HG_BLOCK_EUNSUPPORTED -- which includes "seq_cap has no room for this slice's
 * bases + qualities even with the other slices out of the way" (hg_cram_file_to_bam_host then decodes the batch once more with four times the room).  cigar[] / names[] / aux[] come back
 * PACKED (slice after slice, no gaps); the capacities may be any estimate: if one is too small the call returns HG_ENOMEM and used[]
 * (optional, 4 entries: CIGAR words, name bytes, aux bytes, sequence bytes) says what the batch needs -- hg_cram_records_bound's
 * figures always suffice.  Returns HG_OK / HG_EBLOCK / HG_ENOMEM. */
int hg_cram_decode_records_host(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref,
                                size_t rec_cap, size_t cigar_cap, size_t name_cap, size_t seq_cap, size_t aux_cap,
                                const hg_cram_record_cols *out, uint64_t *rec_off, int32_t *status, uint64_t *used);

/* CRAM slices -> uncompressed BAM records: cram_decode_slice + cram_to_bam (cram/cram_decode.c:2346-3192; bam_set1 and the on-disk
 * layout of bam_write1, sam.c) entirely on the device -- the record loop above, then one lane per record sizes and writes
 * block_size, the 32-byte core (bin from the CIGAR's reference length), QNAME, CIGAR, 4-bit bases, qualities, the stored tags and RG:Z
 * from the read-group series.  Only the BAM bytes cross PCIe; they are what bgzf_write / hg_bgzf_deflate take.  rg_names = the @RG IDs
 * in header order; total_bases >= the bases of the slices (sum of the containers' `bases` fields).  rec_off as above; rec_bam_off
 * (optional, records + 1 entries) = where each record starts in bam_out; *bam_bytes = bytes written, or needed when the call returns
 * HG_ENOMEM.  Records of failed slices are left out; a read-group index outside the header's @RG lines fails its slice with -1 like
 * cram_to_bam.  Records stored without a name get "*"; hg_cram_decode_bam_host2 with a name_prefix (the reference uses the file's base
 * name, cram/cram_io.c:5346) gives them the reference's names instead: the mate's name when it has one, else "<prefix>:<number of the
 * record in the file>" (cram_decode.c:3113-3143).  Not done: CIGARs of more than 65535 operations. */
int hg_cram_decode_bam_host(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref,
                            const char *const *rg_names, int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap,
                            uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes, int32_t *status);

/* Device-resident form of the same path: stage a batch of slices in HBM once (headers parsed, blocks and reference spans uploaded),
 * then decode it -- cram_decode_slice + cram_to_bam -- with the BAM stream left on the device, where hg_bgzf_deflate_dev or the BAM
 * kernels take it.  A run costs kernel launches and two small read-backs (per-slice verdicts and totals, the BAM size).
 * How a slice is decoded (cram_records_fast.h): slices whose compression header gives every series a block of its own (what htslib
 * writes for sorted data) go through data-parallel passes -- column decodes, per-record passes, prefix sums, one lane per record for
 * the feature walk; slices with CORE-coded or shared series, and slices the passes find irregular, go through the serial chain
 * decoder (one wavefront per slice), whose verdict is the one reported.  Both produce the same bytes (tests/test_cram_records_fast.py).
 * *d_bam stays valid until the next run or hg_cram_batch_free; *fast_slices = slices decoded by the passes. */
typedef struct hg_cram_batch hg_cram_batch;
int hg_cram_batch_stage(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref,
                        uint64_t total_bases, hg_cram_batch **out);
int hg_cram_batch_decode_bam_dev(hg_ctx *ctx, hg_cram_batch *batch, const char *const *rg_names, int nrg, const char *name_prefix,
                                 void **d_bam, uint64_t *bam_bytes, uint64_t *nrec, uint64_t *fast_slices, int32_t *status);
int hg_cram_batch_read_bam(hg_ctx *ctx, hg_cram_batch *batch, uint8_t *dst, size_t cap);   /* the last run's stream -> host */
void hg_cram_batch_free(hg_ctx *ctx, hg_cram_batch *batch);

typedef struct hg_cram_ref_seq { const uint8_t *bases; uint64_t len; } hg_cram_ref_seq;
/* ---- The write side of the record layer: BAM records -> CRAM slices, i.e. cram_encode_slice fed by process_one_read
 * (cram/cram_encode.c:572-793, 1096-1209, 3382-3700) -- per-record walks (CIGAR against the reference -> features, fixed fields, tags) and
 * prefix sums on the device (cram_encode.hip / cram_encode_core.h).  bam = nrec records in bam_write1's layout, back to back (no header);
 * refs[i] = reference sequence i in upper case (bases NULL = not available: every base is then stored); rg_names = the @RG ids (an RG:Z tag
 * becomes the RG series).  Slices of records_per_slice records; a slice spanning several references is a multi-reference slice.  Per slice
 * i, out + slice_off[i] holds: u32 length + the compression header block, u32 length + the slice header block, u32 nblocks, then per block
 * i32 content id, u32 length, bytes -- uncompressed EXTERNAL blocks, ready for cram_compress_slice / hg_cram_compress_blocks_metrics_host
 * and decodable by hg_cram_decode_records_host (all of it by the data-parallel passes).  Writer's choices: cram_encode_core.h's header.
 * status[i] = 0, -1 (malformed BAM record) or HG_BLOCK_EUNSUPPORTED (a record with a CIGAR but no bases; more than 64 distinct tags or 256
 * distinct tag lists in the slice; an RG:Z naming no @RG line).  Returns HG_OK / HG_EBLOCK / HG_ENOMEM (*out_bytes = bytes needed). */
int hg_cram_encode_slices_host(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, size_t nrec, uint32_t records_per_slice,
                               const hg_cram_ref_seq *refs, int nrefs, const char *const *rg_names, int nrg, int64_t record_counter0,
                               uint8_t *out, size_t out_cap, uint64_t *slice_off, size_t max_slices, int32_t *status, uint64_t *out_bytes);
/* The same for a caller that has not walked the records: *nrec_io = 0 on entry -> they are counted here (bam_read1's framing, sam.c:784-866, runs on the
 * device in any case; a host walk costs one cache miss per record), on return the number of records; slice_off / status need room for
 * bam_len / 36 / records_per_slice + 2 slices then.  slice_bases (may be NULL): per slice the sum of the records' l_seq -- the base count of the
 * container header (cram_write_container, cram/cram_io.c:3890-4010). */
int hg_cram_encode_slices_host2(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, size_t *nrec_io, uint32_t records_per_slice,
                                const hg_cram_ref_seq *refs, int nrefs, const char *const *rg_names, int nrg, int64_t record_counter0,
                                uint8_t *out, size_t out_cap, uint64_t *slice_off, size_t max_slices, int32_t *status, uint64_t *out_bytes,
                                uint64_t *slice_bases);

/* A whole CRAM 2.x / 3.x file -> the uncompressed BAM stream `samtools view -u -b` would hand to bgzf_write: the container / block walk
 * of cram_read_container / cram_read_block on the host, every block through cram_uncompress_block (CRC check included) in one batch,
 * every slice through hg_cram_decode_bam_host in one batch, and bam_hdr_write's header in front.  refs[i] = reference sequence i of the
 * @SQ lines in upper case (bases == NULL or nrefs_given == 0: not available -- bases come out as '=' plus the stored edits); embedded
 * reference blocks are used when a slice has one.  *bam_bytes = bytes written, or needed when the call returns HG_ENOMEM.  A block that
 * fails to decode fails the file (HG_EBLOCK), as in the reference. */
int hg_cram_decode_bam_host2(hg_ctx *ctx, size_t nslices, const hg_cram_slice_blocks *slices, int major_version, int nref,
                             const char *const *rg_names, int nrg, uint64_t total_bases, uint8_t *bam_out, size_t bam_cap,
                             uint64_t *rec_off, uint64_t *rec_bam_off, uint64_t *bam_bytes, int32_t *status, const char *name_prefix);
int hg_cram_file_to_bam_host(hg_ctx *ctx, const uint8_t *cram, size_t cram_len, const hg_cram_ref_seq *refs, int nrefs_given,
                             uint8_t *bam_out, size_t bam_cap, uint64_t *bam_bytes, uint64_t *nrecords);
/* The same with options.  By default the MD5 of the reference span in every slice header is checked against the bases about to be used,
 * as cram_decode_slice does (cram/cram_decode.c:2480-2540; a mismatch fails the file with HG_EBLOCK -- the reference's "MD5 checksum
 * reference mismatch"); HG_CRAM_IGNORE_MD5 = the reference's ignore_md5 option.  Containers written without a reference (RR = 0) get none. */
#define HG_CRAM_IGNORE_MD5 1
int hg_cram_file_to_bam_host2(hg_ctx *ctx, const uint8_t *cram, size_t cram_len, const hg_cram_ref_seq *refs, int nrefs_given,
                              uint8_t *bam_out, size_t bam_cap, uint64_t *bam_bytes, uint64_t *nrecords, int flags, const char *name_prefix);

/* The same for a reader that walks the file itself -- cram_get_bam_seq's whole-slice path (cram/cram_decode.c:3268-3627: cram_next_slice -> cram_decode_slice ->
 * cram_to_bam; htslib_amd/csrc/cram_record_front.c calls this under that name inside a libhts build): the BODIES of a run of data containers, as they
 * lie in the file after each container header (compression header block, then per slice the slice header block and its blocks), become the
 * BAM records of those containers back to back -- no BAM header in front.  num_blocks / bases = the container header's fields.  sq_len[nref] = the
 * @SQ LN values; rg_names = the @RG IDs in header order; refs as above; decode_md = the cram_fd's decode_md (-1 = hts_open's default: MD / NM are
 * made when the file does not store them).  flags: HG_CRAM_IGNORE_MD5.  *bam_bytes = bytes written, or needed when the call returns HG_ENOMEM.
 * HG_EBLOCK: a block or slice failed as it would fail cram_read_slice / cram_decode_slice (the caller lets the reference's decoder report it). */
typedef struct hg_cram_container { const uint8_t *body; uint32_t body_len; int32_t num_blocks; uint64_t bases; } hg_cram_container;
/* References on demand: called (on the calling thread, once per id and call) only for the sequences a slice actually needs -- cram_get_ref's role
 * (cram/cram_io.c:3409).  Returns 0 and fills *out (upper case bases, valid until the decode call returns), or non-zero: not available.
 * With get_ref != NULL, refs / nrefs_given are ignored. */
typedef int (*hg_cram_get_ref_fn)(void *user, int ref_id, hg_cram_ref_seq *out);
int hg_cram_containers_to_bam_host(hg_ctx *ctx, int major_version, size_t ncontainers, const hg_cram_container *containers, int nref, const int64_t *sq_len,
                                   const char *const *rg_names, int nrg, const hg_cram_ref_seq *refs, int nrefs_given, hg_cram_get_ref_fn get_ref, void *user,
                                   int flags, int decode_md, const char *name_prefix, uint8_t *bam_out, size_t bam_cap, uint64_t *bam_bytes, uint64_t *nrecords);

/* cram_index_build (cram/cram_index.c:779-870): the .crai text of a whole CRAM 2.x / 3.x file, one line per slice -- or per run of records on one
 * reference for multi-reference slices, which are the only slices decoded (blocks + the ref_id / apos / aend columns, one batch each).
 * Returns the number of bytes written, -2 for a file that is not sorted (the reference's error), or a negative HG_E* code.  The reference
 * gzips the text (bgzf "wg"); compress with hg_gzip_deflate_host when a .crai file is wanted. */
long hg_cram_index_build_host(hg_ctx *ctx, const uint8_t *cram, size_t cram_len, char *out, size_t cap);

/* The other direction: an uncompressed BAM stream (header + records, as bam_hdr_write / bam_write1 lay it out and as
 * hg_cram_file_to_bam_host returns it) -> a CRAM 3.0 file: hg_cram_encode_slices_host for the records, every series block through the
 * method auto-tuner with the CRAM 3.0 set GZIP | rANS 4x8 (the codec of this library that is pinned against the reference's fixtures),
 * then cram_write_file_def / cram_write_SAM_hdr / cram_write_container / cram_write_block's framing with one slice per container and the
 * EOF container.  records_per_slice 0 = 10 000, level <= 0 = 5.  Read back by hg_cram_file_to_bam_host (tests/test_cram_encode.py); the
 * container / block layout follows the reference's writer, but no stock htslib was available to read these files here. */
int hg_bam_to_cram_host(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, const hg_cram_ref_seq *refs, int nrefs_given,
                        uint32_t records_per_slice, int level, uint8_t *cram_out, size_t cram_cap, uint64_t *cram_bytes, uint64_t *nrecords);
/* ... with flags: HG_CRAM_WRITE_V31 = a CRAM 3.1 file whose blocks go through the auto-tuner with the method sets cram_compress_slice gives a 3.1 writer with
 * use_rans + use_tok (htslib's default profile: rANS Nx16 with PACK / RLE / STRIPE / order-1 variants by level, TOK3 for the read names, GZIP);
 * HG_CRAM_WRITE_ARITH adds the range coder's sets (use_arith; names: TOKA).  Codec format parity with htscodecs is UNPINNED. */
#define HG_CRAM_WRITE_V31   1
#define HG_CRAM_WRITE_ARITH 2
int hg_bam_to_cram_host2(hg_ctx *ctx, const uint8_t *bam, size_t bam_len, const hg_cram_ref_seq *refs, int nrefs_given,
                         uint32_t records_per_slice, int level, int flags, uint8_t *cram_out, size_t cram_cap, uint64_t *cram_bytes, uint64_t *nrecords);

/* A writer that hands its records over in RUNS (the whole-slice writer under cram_put_bam_seq, htslib_amd/csrc/cram_record_front.c: cram/cram_encode.c:4042
 * cram_put_bam_seq -> cram_encode_container -> cram_flush_container, a run of slices at a time): the object keeps what a cram_fd keeps between slices -- one
 * cram_metrics per data series, so that the method trials of the first slices are not repeated, and the record counter of the container headers.
 * hg_cram_writer_containers_host: bam_records = records in bam_write1's layout back to back (no BAM header); out receives the data containers of these records
 * (one slice per container, as hg_bam_to_cram_host2 writes them), nothing else -- file definition, header container and EOF container are the caller's.
 * flags / level / records_per_slice as hg_bam_to_cram_host2.  HG_ENOMEM: *out_bytes = bytes needed (nothing of the run was kept: call again). */
typedef struct hg_cram_writer hg_cram_writer;
hg_cram_writer *hg_cram_writer_new(uint32_t records_per_slice, int level, int flags);
void hg_cram_writer_free(hg_cram_writer *w);
int hg_cram_writer_containers_host(hg_ctx *ctx, hg_cram_writer *w, const uint8_t *bam_records, size_t len, const hg_cram_ref_seq *refs, int nrefs_given,
                                   const char *const *rg_names, int nrg, uint8_t *out, size_t cap, uint64_t *out_bytes, uint64_t *nrecords);

/* The .crai text of one slice (cram_index_slice / cram_index_build_multiref, cram/cram_index.c:632-728): "ref start span container_pos
 * landmark slice_bytes" -- one line from the slice header, or, for a multi-reference slice, one line per run of records on the same
 * reference from the ref_id / apos / aend columns of hg_cram_decode_records_host (the arrays of THIS slice).  Returns the number of
 * bytes written (no terminator), -2 if the records are not sorted (as the reference), or a negative HG_E* code.  Host only. */
long hg_cram_crai_slice(const uint8_t *slice_hdr, uint32_t slice_hdr_len, int major_version, const int32_t *ref_id, const int64_t *apos,
                        const int64_t *aend, int64_t container_pos, int32_t landmark, int32_t slice_bytes, char *out, size_t cap);

/* ---- BAM record framing on the device: the framing half of bam_read1 (sam.c:784-866) and
 *      nibble2base (simd.c:119-161) for consumers that keep the inflated stream in HBM. ---- */
#define HG_BAM_ETRUNC   (-2)   /* the stream ends inside a record (bam_read1 returns -2 / -3) */
#define HG_BAM_EINVALID (-4)   /* a record fails bam_read1's sanity checks (it returns -4) */
/* bam_hdr_read's walk (sam.c:229-335) over a host copy of the start of the stream: number of reference sequences and
 * the offset of the first alignment record. */
int hg_bam_header_host(const uint8_t *bam, size_t len, int32_t *n_ref, uint64_t *first_record_off);
/* Offsets of every record's block_len field in a device-resident uncompressed BAM stream, in order.  Returns the
 * record count (even when d_rec_off is NULL or smaller than the count -- only max_rec offsets are written), or
 * HG_BAM_ETRUNC / HG_BAM_EINVALID with *bad_off = offset of the offending record, or a negative HG_E* code.
 * Exact: per-chunk guesses are verified link by link on the host.  Synchronises the stream. */
long hg_bam_frame_dev(hg_ctx *ctx, const void *d_bam, uint64_t len, uint64_t first_record_off, int32_t n_ref,
                      uint64_t *d_rec_off, uint64_t max_rec, uint64_t *bad_off, void *stream);
/* The fixed fields of every record as device columns (bam1_core_t as bam_read1 fills it, sam.c:808-821); NULL
 * columns are skipped.  Asynchronous on `stream`. */
typedef struct hg_bam_core_cols {
    int32_t *tid, *pos; uint16_t *bin; uint8_t *mapq, *l_qname; uint16_t *flag, *n_cigar; int32_t *l_qseq, *mtid, *mpos, *isize;
} hg_bam_core_cols;
int hg_bam_core_dev(hg_ctx *ctx, const void *d_bam, const uint64_t *d_rec_off, uint64_t n, const hg_bam_core_cols *cols, void *stream);
/* Qualities as Phred+33 text (what sam_format1 prints, sam.c:4368-4376; 0xff = absent is kept) at the SAME offsets as the
 * bases of hg_bam_bases_dev.  Asynchronous on `stream`. */
int hg_bam_quals_dev(hg_ctx *ctx, const void *d_bam, const uint64_t *d_rec_off, uint64_t n, const uint64_t *d_base_off,
                     void *d_quals, void *stream);
/* d_base_off[i] (n+1 entries) = start of record i's bases in d_bases; d_bases = "=ACMGRSVTWYHKDBN" text of every
 * record back to back (pass d_bases NULL to get only the offsets and *total_bases).  Synchronises the stream. */
int hg_bam_bases_dev(hg_ctx *ctx, const void *d_bam, const uint64_t *d_rec_off, uint64_t n, uint64_t *d_base_off,
                     void *d_bases, uint64_t bases_cap, uint64_t *total_bases, void *stream);

/* ---- BAI index from a device-resident BAM stream: what `samtools index` computes -- the
 *      bam_read1 + hts_idx_push loop of sam_index (sam.c:994-1031), hts_idx_finish / compress_binning and idx_save_core
 *      (hts.c:2431-2640, 2759-2822).  d_rec_off / nrec come from hg_bam_frame_dev; blocks[] = EVERY BGZF block of the file
 *      in order as hg_bgzf_scan reports them (empty blocks and the EOF block included), file_size = compressed size;
 *      ref_len[] = the header's reference lengths.  Writes the .bai bytes to out and returns their count; the bins of a
 *      reference are written in ascending order (htslib writes its hash-table order; the content is identical).
 *      HG_BAM_EUNSORTED = the conditions `samtools index` refuses (unsorted positions, a reference in several blocks,
 *      unplaced reads before placed ones).  Synchronises the stream. ---- */
#define HG_BAM_EUNSORTED (-5)
long hg_bai_build_dev(hg_ctx *ctx, const void *d_bam, uint64_t len, uint64_t first_record_off, int32_t n_ref,
                      const uint32_t *ref_len, const uint64_t *d_rec_off, uint64_t nrec, const hg_bgzf_desc *blocks,
                      uint64_t nblocks, uint64_t file_size, uint8_t *out, size_t out_cap, void *stream);
/* General form: csi = 0 -> BAI (min_shift 14, n_lvls 5); csi = 1 -> the CSI layout of `samtools index -c` (min_shift 14
 * by default, n_lvls = hg_csi_levels(longest reference, min_shift), i.e. hts_adjust_csi_settings as sam_index calls it,
 * sam.c:1003-1012), uncompressed: the caller BGZF-compresses it as hts_idx_save_as does. */
int hg_csi_levels(uint64_t max_ref_len, int min_shift);
long hg_idx_build_dev(hg_ctx *ctx, const void *d_bam, uint64_t len, uint64_t first_record_off, int32_t n_ref,
                      const uint32_t *ref_len, const uint64_t *d_rec_off, uint64_t nrec, const hg_bgzf_desc *blocks,
                      uint64_t nblocks, uint64_t file_size, int csi, int min_shift, int n_lvls, uint8_t *out, size_t out_cap,
                      void *stream);

/* ---- CRC-32 (replaces hts_crc32, bgzf.c:557-559 / 620-622) -------------- */
/* crc[i] = crc32(0, d_data + off[i], len[i]) for n independent buffers. */
int hg_crc32_dev(hg_ctx *ctx, const void *d_data,
                 const uint64_t *d_off, const uint32_t *d_len, size_t n,
                 uint32_t *d_crc, void *stream);

/* crc[i] = crc32(0, buf[i], len[i]) for n host buffers in one device round trip (the block CRCs of
 * cram_uncompress_block / cram_write_block, cram_io.c:1585-1592, 1538-1554).  Synchronous. */
int hg_crc32_batch_host(hg_ctx *ctx, const uint8_t *const *buf, const uint32_t *len, size_t n, uint32_t *crc);
/* CRC-32 of one host buffer (upload + device CRC + host combine).  Synchronous. */
int hg_crc32_host(hg_ctx *ctx, const void *buf, size_t len, uint32_t *crc);

/* ---- CRAM 4.0 E_XPACK / E_XRLE byte transforms: the htscodecs functions cram_codecs.c calls --
 * hts_unpack (cram/cram_codecs.c:1399), hts_pack (:1520), hts_rle_decode (:2106), hts_rle_encode (:2278) -- with the
 * context as an extra first argument; arguments and results otherwise as htscodecs/pack.h and rle.h define them.
 * The same kernels serve the PACK / RLE flags of rANS Nx16 streams.  Synchronous, one wavefront per call; lengths
 * up to 2^31-1.  hts_cram_gpu.h has the wrappers under the reference's own names. ---- */
/* <= 16 distinct byte values -> 1-, 2- or 4-bit codes, first value in the low bits.  out_meta (room for 17 bytes) =
 * [number of symbols][symbols ascending]; returns a malloc'd buffer of *out_len bytes.  More than 16 values: a copy,
 * out_meta_len 1.  One value: *out_len 0.  NULL on error. */
uint8_t *hg_hts_pack(hg_ctx *ctx, const uint8_t *data, int64_t len, uint8_t *out_meta, int *out_meta_len, uint64_t *out_len);
/* nsym = symbols PER BYTE as hts_unpack_meta reports them: 8, 4, 2; 1 = not packed (copy), 0 = constant p[0].
 * p = the symbol map (>= 16 entries).  Returns out, NULL when data is too short for out_len symbols. */
uint8_t *hg_hts_unpack(hg_ctx *ctx, const uint8_t *data, int64_t len, uint8_t *out, uint64_t out_len, int nsym, const uint8_t *p);
/* Symbols in rle_syms are written once per run to the literal stream, run length - 1 as a 7-bit varint to `run`
 * (room for data_len + 8 bytes).  *rle_nsyms == 0 on entry: the symbols whose repeats outnumber their run starts are
 * chosen and returned.  out == NULL: malloc'd (2 * data_len).  Returns the literals (*out_len bytes). */
uint8_t *hg_hts_rle_encode(hg_ctx *ctx, const uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms,
                           int *rle_nsyms, uint8_t *out, uint64_t *out_len);
/* The inverse; *out_len = room in `out` on entry, bytes produced on return.  NULL on overrun / malformed run lengths. */
uint8_t *hg_hts_rle_decode(hg_ctx *ctx, const uint8_t *lit, uint64_t lit_len, const uint8_t *run, uint64_t run_len, const uint8_t *rle_syms,
                           uint32_t rle_nsyms, uint8_t *out, uint64_t *out_len);

/* ---- CRAM integer data series <-> EXTERNAL blocks (first step): a whole block of ITF8 values becomes an
 * int32 column in one pass and back, instead of cram_decode_slice pulling one value per record out of the block
 * (cram_external_decode_int, cram/cram_codecs.c:350-368 -> safe_itf8_get, cram/cram_io.c:644-673; encode side
 * cram_external_encode_int, cram_codecs.c:523-527 -> itf8_put, cram_io.c:277-305). ---- */
/* d_desc[i]: in_off / in_len = the block's bytes in d_in; out_off = first int32 of its column in d_out, counted in VALUES;
 * out_len = room there, in values.  d_count[i] = values decoded.  d_status[i] = 0, or -1 when the block ends inside a
 * value (the reference's *err) or the column has no room.  No sync inside. */
int hg_cram_itf8_decode_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, int32_t *d_out, uint32_t *d_count,
                            int32_t *d_status, void *stream);
/* d_desc[i]: in_off / in_len = first value / number of values in d_in; out_off / out_len = where the bytes go in d_out and
 * the room there (5 bytes per value always suffice).  d_out_len[i] = bytes written; d_status[i] = 0 / -1 (no room). */
int hg_cram_itf8_encode_dev(hg_ctx *ctx, const int32_t *d_in, const hg_stream_desc *d_desc, size_t n, void *d_out, uint32_t *d_out_len,
                            int32_t *d_status, void *stream);
/* BYTE_ARRAY_STOP series (read names, string tags; cram_byte_array_stop_decode_char, cram/cram_codecs.c:3586-3624): the offset of every
 * item of a block.  d_desc[i]: in_off / in_len = the block, reserved = the stop byte, out_off = first word of its offset table in
 * d_off, out_len = room there in words.  d_off[out_off + k] = start of item k (k < count), [count] = in_len; item k is the bytes
 * [off[k], off[k + 1] - 1).  d_status[i] = 0, or -1 when bytes follow the last stop byte (the reference's -1) or the table has no room. */
int hg_cram_byte_array_stop_dev(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, uint32_t *d_off, uint32_t *d_count,
                                int32_t *d_status, void *stream);
int hg_cram_byte_array_stop_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, const uint8_t *stop, size_t n,
                                 uint32_t *const *off, const uint32_t *cap, uint32_t *count, int32_t *status);
/* Host-buffer forms (one PCIe round trip for the batch); HG_EBLOCK when some status[i] != 0. */
int hg_cram_itf8_decode_host(hg_ctx *ctx, const uint8_t *const *in, const uint32_t *in_len, size_t n, int32_t *const *out, const uint32_t *cap,
                             uint32_t *count, int32_t *status);
int hg_cram_itf8_encode_host(hg_ctx *ctx, const int32_t *const *in, const uint32_t *nvals, size_t n, uint8_t *const *out, const uint32_t *cap,
                             uint32_t *out_len, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* HTSGPU_H */

/*
 * hts_bgzf_gpu.h -- the BGZF front-end of htslib re-implemented on top of the gfx950 engine.
 *
 * Same names, argument meaning, return values and error bits as htslib/bgzf.h (reference
 * htslib/bgzf.h:50-58,68-84,110-495) so that code written against libhts' BGZF API compiles and
 * behaves the same; `struct BGZF` has the reference's public layout because callers read
 * block_offset / block_length / block_address / uncompressed_block directly (bgzf_tell macro,
 * bgzf_read_small/bgzf_write_small inlines, sam.c:800-803).  The reference's own test/test_bgzf.c
 * and bgzip.c are built against this library, unmodified, by tests/test_reference_programs.py.
 *
 * How it differs from bgzf.c, by design:
 *   - blocks are (de)compressed in BATCHES on the GPU (libhtsgpu.so, hg_pipe_*).  A reader always behaves like the
 *     reference's threaded mode: an I/O thread prefetches and inflates windows of blocks, `fp->mt` is non-NULL,
 *     bgzf_set_cache_size() is ignored as it is with threads (bgzf.c:2126-2130).  The first blocks after bgzf_open / bgzf_seek are
 *     inflated on the calling thread by the host codec (bgzf_host_codec.h): random access costs what it costs the reference.
 *   - a writer WITHOUT bgzf_mt() keeps bgzf_tell() exact between writes like the reference's single-threaded writer
 *     (each block is compressed where it is cut, on the caller's thread, by the host codec).  bgzf_mt() / bgzf_thread_pool() create no
 *     threads but switch the writer to the reference's threaded contract: blocks are batched, `fp->mt` becomes non-NULL,
 *     fp->block_address is valid after bgzf_flush() only and index offsets go through bgzf_idx_push
 *     (bgzf.c:189-290, 1953-1967; sam.c:942).
 *   - `fp->fp` is a real hFILE; all I/O goes through the exported hFILE functions, so the library works with libhts'
 *     hfile.c (every transport) or with the bundled local-file provider (hfile_min.cpp).  The engine state hangs on
 *     `fp->cache` (the reference's block cache has no role here).
 *   - plain gzip input and mode "g" output are supported through the engine as well (one wavefront per stream:
 *     it works, slowly; BGZF is the fast path).
 *   - there is no CPU codec: without a usable MI355X bgzf_open() fails (ENODEV) for compressed streams.
 */
#ifndef HTS_BGZF_GPU_H
#define HTS_BGZF_GPU_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGZF_BLOCK_SIZE     0xff00
#define BGZF_MAX_BLOCK_SIZE 0x10000

#define BGZF_ERR_ZLIB   1
#define BGZF_ERR_HEADER 2
#define BGZF_ERR_IO     4
#define BGZF_ERR_MISUSE 8
#define BGZF_ERR_MT     16
#define BGZF_ERR_CRC    32

struct hFILE;
struct hts_tpool;
struct bgzf_mtaux_t;
typedef struct bgzidx_t bgzidx_t;
typedef struct bgzf_cache_t bgzf_cache_t;
struct z_stream_s;

/* htslib/kstring.h layout */
#ifndef KSTRING_T
#define KSTRING_T kstring_t
typedef struct kstring_t { size_t l, m; char *s; } kstring_t;
#endif

struct BGZF {
    unsigned errcode:16, reserved:1, is_write:1, no_eof_block:1, is_be:1;
    signed compress_level:9;
    unsigned last_block_eof:1, is_compressed:1, is_gzip:1;
    int cache_size;
    int block_length, block_clength, block_offset;
    int64_t block_address, uncompressed_address;
    void *uncompressed_block, *compressed_block;
    bgzf_cache_t *cache;            /* the batch engine of this handle (NULL for uncompressed handles); opaque */
    struct hFILE *fp;               /* the transport, as in the reference (bgzf_hfile) */
    struct bgzf_mtaux_t *mt;        /* non-NULL = "threaded" contract in force (see above); opaque */
    bgzidx_t *idx;
    int idx_build_otf;
    struct z_stream_s *gz_stream;   /* unused: gzip streams go through the engine too */
    int64_t seeked;
};
typedef struct BGZF BGZF;

BGZF *bgzf_dopen(int fd, const char *mode);
BGZF *bgzf_open(const char *path, const char *mode);
BGZF *bgzf_hopen(struct hFILE *fp, const char *mode);          /* bgzf.c:539 */
struct hFILE *bgzf_hfile(BGZF *fp);                            /* bgzf.c:2619 */
int bgzf_close(BGZF *fp);
ssize_t bgzf_read(BGZF *fp, void *data, size_t length);
ssize_t bgzf_write(BGZF *fp, const void *data, size_t length);
ssize_t bgzf_block_write(BGZF *fp, const void *data, size_t length);
int bgzf_peek(BGZF *fp);
ssize_t bgzf_raw_read(BGZF *fp, void *data, size_t length);
ssize_t bgzf_raw_write(BGZF *fp, const void *data, size_t length);
int bgzf_flush(BGZF *fp);
#define bgzf_tell(fp) (((fp)->block_address << 16) | ((fp)->block_offset & 0xFFFF))
int64_t bgzf_seek(BGZF *fp, int64_t pos, int whence);
int bgzf_check_EOF(BGZF *fp);
int bgzf_compression(BGZF *fp);
int bgzf_is_bgzf(const char *fn);
void bgzf_set_cache_size(BGZF *fp, int size);
int bgzf_flush_try(BGZF *fp, ssize_t size);
int bgzf_getc(BGZF *fp);
int bgzf_getline(BGZF *fp, int delim, kstring_t *str);
int bgzf_read_block(BGZF *fp);
int bgzf_thread_pool(BGZF *fp, struct hts_tpool *pool, int qsize);
int bgzf_mt(BGZF *fp, int n_threads, int n_sub_blks);
int bgzf_compress(void *dst, size_t *dlen, const void *src, size_t slen, int level);
int bgzf_useek(BGZF *fp, off_t uoffset, int where);
off_t bgzf_utell(BGZF *fp);
int bgzf_index_build_init(BGZF *fp);
int bgzf_index_load(BGZF *fp, const char *bname, const char *suffix);
int bgzf_index_load_hfile(BGZF *fp, struct hFILE *idx, const char *name);   /* bgzf.c:2472 */
int bgzf_index_dump(BGZF *fp, const char *bname, const char *suffix);
int bgzf_index_dump_hfile(BGZF *fp, struct hFILE *idx, const char *name);   /* bgzf.c:2385 */
/* hts_idx_push with the block address resolved once the block has been compressed (bgzf.c:189-290);
 * hidx is the caller's hts_idx_t, handed on to libhts' hts_idx_push. */
int bgzf_idx_push(BGZF *fp, void *hidx, int tid, int64_t beg, int64_t end, uint64_t offset, int is_mapped);
/* CRC-32 of a host buffer, computed on the device (bgzf.c:557-559).  One PCIe round trip per call: hot paths
 * use the batch entry points of htsgpu.h instead. */
uint32_t hts_crc32(uint32_t crc, const void *buf, size_t len);

static inline ssize_t bgzf_read_small(BGZF *fp, void *data, size_t length) {
    if ((ssize_t)length < fp->block_length - fp->block_offset) {
        memcpy((uint8_t *)data, (uint8_t *)fp->uncompressed_block + fp->block_offset, length);
        fp->block_offset += (int)length;
        fp->uncompressed_address += (int64_t)length;
        return (ssize_t)length;
    }
    return bgzf_read(fp, data, length);
}
static inline ssize_t bgzf_write_small(BGZF *fp, const void *data, size_t length) {
    if (fp->is_compressed && (size_t)(BGZF_BLOCK_SIZE - fp->block_offset) > length) {
        memcpy((uint8_t *)fp->uncompressed_block + fp->block_offset, data, length);
        fp->block_offset += (int)length;
        return (ssize_t)length;
    }
    return bgzf_write(fp, data, length);
}

#ifdef __cplusplus
}
#endif
#endif

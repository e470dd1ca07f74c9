/*
 * hts_bgzf_gpu.h -- the BGZF front-end of htslib re-implemented on top of the gfx950 engine.
 *
 * Same names, argument meaning, return values and error bits as htslib/bgzf.h (reference
 * htslib/bgzf.h:50-58,68-84,110-495) so that code written against libhts' BGZF API compiles and
 * behaves the same; `struct BGZF` has the reference's public layout because callers read
 * block_offset / block_length / block_address / uncompressed_block directly (bgzf_tell macro,
 * bgzf_read_small/bgzf_write_small inlines, sam.c:800-803).
 *
 * Differences (by design):
 *   - blocks are (de)compressed in BATCHES on the GPU (libhtsgpu.so): the reader reads ahead and
 *     inflates a window of blocks per kernel launch, the writer collects blocks and deflates them
 *     per launch.  bgzf_mt()/bgzf_thread_pool() are accepted and are no-ops (the batch engine
 *     replaces the per-block pool jobs of bgzf.c:1598-1738 and 1852-1925).
 *   - transport is a plain POSIX fd (no hFILE plugins); `fp->fp` is private.
 *   - plain gzip input (not BGZF) and mode "g" are not supported (bgzf_open returns NULL);
 *     uncompressed pass-through ("u", or non-gzip input) is.
 *   - there is no CPU codec: without a usable MI355X bgzf_open() fails for compressed streams.
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
    bgzf_cache_t *cache;
    struct hFILE *fp;               /* private: the fd-based engine state */
    struct bgzf_mtaux_t *mt;
    bgzidx_t *idx;
    int idx_build_otf;
    struct z_stream_s *gz_stream;
    int64_t seeked;
};
typedef struct BGZF BGZF;

BGZF *bgzf_dopen(int fd, const char *mode);
BGZF *bgzf_open(const char *path, const char *mode);
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
int64_t bgzf_useek(BGZF *fp, off_t uoffset, int where);
off_t bgzf_utell(BGZF *fp);
int bgzf_index_build_init(BGZF *fp);
int bgzf_index_load(BGZF *fp, const char *bname, const char *suffix);
int bgzf_index_dump(BGZF *fp, const char *bname, const char *suffix);

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

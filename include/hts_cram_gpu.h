/*
 * hts_cram_gpu.h -- the CRAM block layer of htslib on the gfx950 engine: cram_uncompress_block / cram_compress_block
 * with the reference's own `struct cram_block` calling convention (one malloc'd buffer per block, replaced in place).
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   struct cram_block, enum cram_block_method_int, struct cram_metrics   cram/cram_structs.h:215-266,284-305,312-332
 *   int  cram_uncompress_block(cram_block *b)                            cram/cram_io.c:1576-1754   (htslib.map:164)
 *   int  cram_compress_block / cram_compress_block2 / 3                  cram/cram_io.c:1912-2325   (htslib.map:149)
 *   cram_block *cram_new_block / void cram_free_block                    cram/cram_io.c:1388, 1565
 *   cram_block *cram_read_block(cram_fd*) / int cram_write_block         cram/cram_io.c:1414-1483, 1511-1560
 *   cram_metrics *cram_new_metrics(void)                                 cram/cram_io.c:2327-2339
 *
 * The two structs below have the reference's layout (tests/native/cram_layout_check.c asserts every offset against
 * the real cram/cram_structs.h), so objects compiled against htslib's headers can call these functions directly.
 * `cram_fd` / `cram_slice` are large private structs; the block layer needs five scalars of the former and nothing
 * of the latter (fqzcomp's per-record lengths aside), so those travel in `hg_cram_opts`.  INTEGRATION.md shows the
 * three-line bodies that make cram_io.c's own cram_compress_block2 / cram_read_block forward here.
 *
 * Batching.  One block is far too little work for a GPU, and htslib calls these functions one block at a time from
 * many pool workers (one slice per worker).  Two answers:
 *   - the ARRAY forms (cram_uncompress_blocks, hg_cram_compress_blocks) take all blocks of a slice in one call;
 *   - the single-block forms COALESCE: concurrent callers are gathered for a short window by a leader thread and
 *     sent to the device as one batch (each caller still gets its own result and its own malloc'd buffer).
 * Results follow the reference's ownership rule: b->data is free()d and replaced by a malloc()ed buffer
 * (cram_io.c:1615-1617, 2093-2097).
 */
#ifndef HTS_CRAM_GPU_H
#define HTS_CRAM_GPU_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

struct hFILE;

/* cram/cram_structs.h:215-266 (values are the on-disk ids up to TOK3, internal parameterised ones above) */
enum cram_block_method_int {
    BM_ERROR = -1,
    RAW = 0, GZIP = 1, BZIP2 = 2, LZMA = 3, RANS = 4, RANS0 = RANS,
    RANSPR = 5, RANS_PR0 = RANSPR, ARITH = 6, ARITH_PR0 = ARITH, FQZ = 7, TOK3 = 8,
    GZIP_RLE = 11, GZIP_1, FQZ_b, FQZ_c, FQZ_d,
    RANS1,
    RANS_PR1, RANS_PR64, RANS_PR9, RANS_PR128, RANS_PR129, RANS_PR192, RANS_PR193,
    TOKA,
    ARITH_PR1, ARITH_PR64, ARITH_PR9, ARITH_PR128, ARITH_PR129, ARITH_PR192, ARITH_PR193
};

/* htslib/cram.h:84-101: the public (on-disk) method ids, what cram_block_get_method answers */
enum cram_block_method {
    CRAM_COMP_UNKNOWN = -1, CRAM_COMP_RAW = 0, CRAM_COMP_GZIP = 1, CRAM_COMP_BZIP2 = 2, CRAM_COMP_LZMA = 3, CRAM_COMP_RANS4x8 = 4,
    CRAM_COMP_RANSNx16 = 5, CRAM_COMP_ARITH = 6, CRAM_COMP_FQZ = 7, CRAM_COMP_TOK3 = 8
};

/* htslib/cram.h:103-111 */
enum cram_content_type {
    CT_ERROR = -1, FILE_HEADER = 0, COMPRESSION_HEADER = 1, MAPPED_SLICE = 2, UNMAPPED_SLICE = 3, EXTERNAL = 4, CORE = 5
};

#define CRAM_MAX_METHOD 32
typedef struct cram_metrics {
    int trial, next_trial, consistency;
    int sz[CRAM_MAX_METHOD];
    int input_avg_sz, input_avg_delta;
    int method, revised_method;
    int strat;
    int cnt[CRAM_MAX_METHOD];
    double extra[CRAM_MAX_METHOD];
    int unpackable;
} cram_metrics;

typedef struct cram_block {
    enum cram_block_method_int method, orig_method;
    enum cram_content_type content_type;
    int32_t content_id;
    int32_t comp_size;
    int32_t uncomp_size;
    uint32_t crc32;
    int32_t idx;
    unsigned char *data;
    size_t alloc;
    size_t byte;
    int bit;
    cram_metrics *m;
    int crc32_checked;
    uint32_t crc_part;
} cram_block;

cram_block *cram_new_block(enum cram_content_type content_type, int content_id);
void cram_free_block(cram_block *b);
cram_metrics *cram_new_metrics(void);

/* Exactly the reference's function: CRC check (once), method dispatch, b->data replaced, method = RAW; 0 / -1.
 * bzip2 and lzma blocks (methods 2, 3) go to the system's libbz2 / liblzma, looked up with dlopen at first use -- what the reference does
 * with HAVE_LIBBZ2 / HAVE_LIBLZMA (cram_io.c:1626-1664); when the library is absent they fail with -1 and the reference's message. */
int cram_uncompress_block(cram_block *b);
/* All blocks of a slice / container at once; returns 0 or -1 if any block failed (each block is left either fully
 * decoded or untouched; blk_rc, if given, receives the per-block 0 / -1). */
int cram_uncompress_blocks(cram_block **b, int n, int *blk_rc);

/* What cram_compress_block3 reads from its cram_fd (cram_io.c:1951-1965, 1979, 2282). */
typedef struct hg_cram_opts {
    int level;            /* fd->level                                                              */
    int version;          /* fd->version (major << 8 | minor)                                       */
    int use_bz2, use_lzma;/* fd->use_bz2 / use_lzma: offered through the system's libbz2 / liblzma when present     */
    void *metrics_lock;   /* &fd->metrics_lock (pthread_mutex_t *) or NULL                          */
} hg_cram_opts;

/* cram_compress_block2(fd, s, b, metrics, method, level) with fd reduced to opts: `method` is a bit set of
 * cram_block_method_int values (-1 = default GZIP), level -1 = opts->level.  The block is compressed with every
 * method of the set while `metrics` is in a trial phase, else with the method it learnt; statistics, costs and
 * retrial spans follow cram_io.c:1978-2278.  b->data / comp_size / method are replaced as the reference does. */
int hg_cram_compress_block(const hg_cram_opts *opts, cram_block *b, cram_metrics *metrics, int method, int level);
int hg_cram_compress_blocks(const hg_cram_opts *opts, cram_block **b, cram_metrics **metrics, const int *method, int level, int n);

/* Same with one level per block (cram_compress_slice mixes level 1 and fd->level, cram_encode.c:886-935). */
int hg_cram_compress_blocks_lv(const hg_cram_opts *opts, cram_block **b, cram_metrics **metrics, const int *method, const int *level, int n);

/* The `cram_slice *s` argument of cram_compress_block2/3 reduced to what a codec reads from it: the per-record quality lengths
 * and BAM flags that cram_compress_by_method copies into an fqz_slice for the FQZ* methods (cram_io.c:1808-1820; same layout as
 * htscodecs' fqz_slice, so that one can be passed with a cast).  With it a quality block keeps the FQZ / FQZ_b / FQZ_c / FQZ_d
 * bits of its method set (fqzcomp.hip); without it those bits are dropped, as for a NULL slice. */
#ifndef HG_FQZ_SLICE_DEFINED
#define HG_FQZ_SLICE_DEFINED
typedef struct hg_fqz_slice { uint32_t num_records; const uint32_t *len; const uint32_t *flags; } hg_fqz_slice;
#endif
int hg_cram_compress_block_fqz(const hg_cram_opts *opts, const hg_fqz_slice *fqz, cram_block *b, cram_metrics *metrics, int method, int level);
int hg_cram_compress_blocks_fqz(const hg_cram_opts *opts, cram_block **b, cram_metrics **metrics, const int *method, const int *level,
                                const hg_fqz_slice *const *fqz, int n);

/* ---- cram_compress_slice's method-set policy (cram/cram_encode.c:803-988) as data.  Which codecs a block may be tried
 *      with depends on the file version, the compression level, the use_* options and the data series; the reference
 *      computes five bit sets of cram_block_method_int values and hands one of them, with a level, to cram_compress_block2
 *      per data series.  Pinned by running the reference's own function with a recording cram_compress_block2
 *      (tests/test_cram_slice_policy.py). ---- */
typedef struct hg_cram_slice_opts {
    int level, version;                                   /* fd->level, fd->version (major << 8 | minor)            */
    int use_bz2, use_lzma, use_rans, use_arith, use_fqz, use_tok;
} hg_cram_slice_opts;
typedef struct hg_cram_slice_sets {
    int method;        /* general set                                                    */
    int methodF;       /* the final sweep over blocks that are still RAW                 */
    int qmethod;       /* quality values: + fqzcomp variants                             */
    int qmethodF;
    int method_rn;     /* read names: no rANS / GZIP_RLE, + TOK3 or TOKA                 */
} hg_cram_slice_sets;
void hg_cram_slice_method_sets(const hg_cram_slice_opts *o, hg_cram_slice_sets *sets);
/* Data-series ids (enum cram_DS_ID, cram/cram_structs.h:143-197) of the series the policy names. */
enum { HG_DS_CORE = 0, HG_DS_aux = 1, HG_DS_aux_oz = 9, HG_DS_RN = 11, HG_DS_QS = 12, HG_DS_IN = 13, HG_DS_NS = 20, HG_DS_BA = 30,
       HG_DS_BB = 37, HG_DS_END = 47 };
/* The calls cram_compress_slice makes before its final sweep, in its order: ds[k] = data series (ids >= HG_DS_END = the
 * per-tag aux blocks), set[k] = method set, lv[k] = level.  present[ds] != 0: the slice has that block; core_size = bytes of
 * the CORE block.  Returns the number of calls.  (The final sweep then offers methodF at the slice level to every block
 * that is still RAW.) */
int hg_cram_slice_plan(const hg_cram_slice_opts *o, const uint8_t *present, int naux, int core_size, int *ds, int *set, int *lv, int max);
/* cram_compress_slice for one slice in ONE engine batch (two when a block is still RAW after its first attempt, as the
 * reference's final sweep re-tries it with methodF).  block[ds] / metrics[ds] for ds < HG_DS_END (NULL = absent), nvals[ds] =
 * c->stats[ds]->nvals or 0 (series with > 16 distinct values are marked unpackable, cram_encode.c:877-881); aux[] = the
 * per-tag blocks beyond DS_END with their own b->m metrics.  opts carries the level / version / metrics lock as for
 * hg_cram_compress_block.  Returns 0 / -1. */
int hg_cram_compress_slice(const hg_cram_slice_opts *o, const hg_cram_opts *opts, cram_block **block, cram_metrics **metrics,
                           const int *nvals, cram_block **aux, int naux);
/* ... with the slice's fqz_slice for the DS_QS block (use_fqz puts the FQZ* methods into its set, cram_encode.c:864-869) */
int hg_cram_compress_slice_fqz(const hg_cram_slice_opts *o, const hg_cram_opts *opts, cram_block **block, cram_metrics **metrics,
                               const int *nvals, cram_block **aux, int naux, const hg_fqz_slice *qs);

/* Block framing (cram_read_block / cram_write_block with fd reduced to the transport and the file's major version):
 * method u8, content_type u8, content_id / comp_size / uncomp_size as ITF8 (v2, v3) or uint7 varints (v4), payload,
 * CRC-32 (v3+).  read: b->crc_part = CRC of the header bytes, crc32_checked = ignore_crc. */
cram_block *hg_cram_read_block(struct hFILE *fp, int major_version, int ignore_crc);
int hg_cram_write_block(struct hFILE *fp, int major_version, cram_block *b);

/* ---- The reference's own entry points, by name (cram/cram_io.c:2316-2325, 1414, 1511; htslib.map:149) -- for callers that hold a real
 * `cram_fd *` / `cram_slice *` of htslib 1.23 (LP64).  Those structs are large and private; the functions below read exactly these
 * fields through the byte offsets listed here (tests/native/cram_layout_*.c assert every one of them against the reference's
 * cram/cram_structs.h with offsetof):
 *   cram_fd:    fp (hFILE *), version, level, ignore_md5, use_bz2, use_lzma, metrics_lock
 *   cram_slice: hdr -> num_records, block[DS_QS], crecs[i].flags / .qual   (only for the FQZ methods, as cram_io.c:1808-1820)
 * A libhts built from another release must regenerate the offsets (the layout test prints them) or use the hg_* forms above. */
typedef struct cram_fd cram_fd;
typedef struct cram_slice cram_slice;
enum { HG_CRAM_FD_FP = 0, HG_CRAM_FD_VERSION = 12, HG_CRAM_FD_LEVEL = 136, HG_CRAM_FD_IGNORE_MD5 = 556, HG_CRAM_FD_USE_BZ2 = 560, HG_CRAM_FD_USE_LZMA = 568,
       HG_CRAM_FD_METRICS_LOCK = 35016, HG_CRAM_SLICE_HDR = 0, HG_CRAM_SLICE_BLOCK = 16, HG_CRAM_SLICE_CRECS = 48, HG_CRAM_SLICE_HDR_NUM_RECORDS = 24,
       HG_CRAM_RECORD_SIZE = 144, HG_CRAM_RECORD_FLAGS = 12, HG_CRAM_RECORD_QUAL = 104, HG_CRAM_DS_QS = 12 };
int cram_compress_block(cram_fd *fd, cram_block *b, cram_metrics *metrics, int method, int level);
int cram_compress_block2(cram_fd *fd, cram_slice *s, cram_block *b, cram_metrics *metrics, int method, int level);
cram_block *cram_read_block(cram_fd *fd);
int cram_write_block(cram_fd *fd, cram_block *b);
size_t hg_cram_fd_layout(size_t *offsets);            /* the offsets above in that order (for the layout test) */
uint32_t cram_block_size(cram_block *b);                                   /* cram_io.c:1490-1505 */
/* The public accessors of a block (cram/cram_external.c:522-555; htslib/cram.h:230-262; htslib.map:313-328,617).  cram_block_get_content_id
 * answers -1 for the CORE block; get_method answers the method the block had when it was read (orig_method); "size" / "offset" is the fill
 * level of a block under construction; cram_block_append grows it as block_resize does (cram/cram_io.h:225-238), 0 / -1. */
int32_t cram_block_get_content_id(cram_block *b);
int32_t cram_block_get_comp_size(cram_block *b);
int32_t cram_block_get_uncomp_size(cram_block *b);
int32_t cram_block_get_crc32(cram_block *b);
void *cram_block_get_data(cram_block *b);
int32_t cram_block_get_size(cram_block *b);
enum cram_block_method cram_block_get_method(cram_block *b);
enum cram_content_type cram_block_get_content_type(cram_block *b);
void cram_block_set_content_id(cram_block *b, int32_t id);
void cram_block_set_comp_size(cram_block *b, int32_t size);
void cram_block_set_uncomp_size(cram_block *b, int32_t size);
void cram_block_set_crc32(cram_block *b, int32_t crc);
void cram_block_set_data(cram_block *b, void *data);
void cram_block_set_size(cram_block *b, int32_t size);
size_t cram_block_get_offset(cram_block *b);
void cram_block_set_offset(cram_block *b, size_t offset);
int cram_block_append(cram_block *b, const void *data, int size);
void cram_block_update_size(cram_block *b);

/* ---- CRAM 4.0 E_XPACK / E_XRLE transforms: the htscodecs functions cram_codecs.c calls (cram/cram_codecs.c:1399, 1520,
 *      2106, 2278), same names and signatures as htscodecs/pack.h and rle.h, computed by the engine (htsgpu.h:
 *      hg_hts_pack ... document the argument conventions).  htscodecs is an un-vendored submodule of the reference:
 *      semantics follow its published headers, parity unpinned. ---- */
uint8_t *hts_pack(uint8_t *data, int64_t len, uint8_t *out_meta, int *out_meta_len, uint64_t *out_len);
uint8_t *hts_unpack(uint8_t *data, int64_t len, uint8_t *out, uint64_t out_len, int nsym, uint8_t *p);
uint8_t *hts_rle_encode(uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms, int *rle_nsyms,
                        uint8_t *out, uint64_t *out_len);
uint8_t *hts_rle_decode(uint8_t *lit, uint64_t lit_len, uint8_t *run, uint64_t run_len, uint8_t *rle_syms, int rle_nsyms,
                        uint8_t *out, uint64_t *out_len);

/* ---- htscodecs' codec entry points, under htscodecs' own names and signatures (htscodecs_front.cpp): what cram_io.c calls for methods 4-8
 *      (cram/cram_io.c:1668 rans_uncompress, :1838 rans_compress, :1699 rans_uncompress_4x16, :1859 rans_compress_4x16, :1718 arith_uncompress_to,
 *      :1879 arith_compress_to, :1737 tok3_decode_names, :1891 tok3_encode_names, :1686 fqz_decompress, :1821 fqz_compress; hts.c:149 htscodecs_version),
 *      so that an htslib built --with-external-htscodecs can link libhts_bgzf.so in place of libhtscodecs.so.  Results are malloc'd, the caller frees them
 *      (cram_io.c:1675 ...); NULL on failure or when no engine can be had.  `order` of the *_4x16 / arith functions = the RANS_ORDER_* flag byte of
 *      cram/cram_external.c:616-637 (+ 0x8000 RANS_ORDER_SIMD_AUTO: the 32-way layout from 64 KiB); fqz_slice has htscodecs' layout
 *      { int num_records; uint32_t *len; uint32_t *flags; } = hg_fqz_slice.  htscodecs is an un-vendored submodule: parity unpinned. ---- */
const char *htscodecs_version(void);
unsigned char *rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order);
unsigned char *rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size);
unsigned int rans_compress_bound_4x16(unsigned int size, int order);
unsigned char *rans_compress_to_4x16(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order);
unsigned char *rans_compress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order);
unsigned char *rans_uncompress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size);
unsigned int arith_compress_bound(unsigned int size, int order);
unsigned char *arith_compress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order);
unsigned char *arith_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order);
unsigned char *arith_uncompress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size);
unsigned char *arith_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size);
uint8_t *tok3_encode_names(char *blk, int len, int level, int use_arith, int *out_len, int *last_start_p);
uint8_t *tok3_decode_names(uint8_t *in, uint32_t sz, uint32_t *out_len);
char *fqz_compress(int vers, void *fqz_slice, char *in, size_t uncomp_size, size_t *comp_size, int strat, void *gparams);
char *fqz_decompress(char *in, size_t comp_size, size_t *uncomp_size, int *lengths, int nlengths);

/* htscodecs/varint.h (static inlines there too; cram_codecs.c:2103, 2276): big-endian 7 bits per byte, continuation in
 * bit 7.  endp may be NULL for put (no bound).  Return the number of bytes used, 0 when out of room / input. */
#ifndef VARINT_H
static inline int var_put_u64(uint8_t *cp, const uint8_t *endp, uint64_t v) {
    int n = 1, k;
    uint64_t t = v;
    while (t >>= 7) n++;
    if (endp && endp - cp < n) return 0;
    for (k = 0; k < n; k++) cp[k] = (uint8_t)(((v >> (7 * (n - 1 - k))) & 0x7f) | (k + 1 < n ? 0x80 : 0));
    return n;
}
static inline int var_get_u64(uint8_t *cp, const uint8_t *endp, uint64_t *v) {
    uint64_t x = 0;
    int n = 0;
    if (endp && cp >= endp) { *v = 0; return 0; }
    for (;;) {
        const uint8_t c = cp[n++];
        x = (x << 7) | (c & 0x7f);
        if (!(c & 0x80) || n == 10 || (endp && cp + n >= endp)) break;
    }
    *v = x;
    return n;
}
#endif

#ifdef __cplusplus
}
#endif
#endif

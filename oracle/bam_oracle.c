/*
 * bam_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the BAM framing layer that sits directly above bgzf_read:
 *   orc_bam_header   the walk of bam_hdr_read            (reference sam.c:229-335: "BAM\1", l_text, text, n_targets,
 *                    then per target l_name (> 0), name, l_ref)
 *   orc_bam_frame    the per-record framing and sanity checks of bam_read1 (sam.c:784-866): block_len >= 32, the 32
 *                    core bytes, l_qseq >= 0, l_qname >= 1, n_cigar*4 + l_qname + (l_qseq+1)/2 + l_qseq <= block_len-32;
 *                    return codes: record count, or -2 truncated / -4 invalid like bam_read1
 *   orc_nibble2base  nibble2base_default (htslib/sam.h seq_nt16_str "=ACMGRSVTWYHKDBN"; call site sam.c:1433)
 * Pinned by tests/test_bam_frame.py against the reference's own BAM fixtures and their .bai indexes (record
 * boundaries and per-reference mapped/unmapped counts written by reference htslib).
 */
#include <stdint.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

static uint32_t le32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }

/* returns 0 and fills n_ref / first record offset, or -1 */
ORC_EXPORT int orc_bam_header(const uint8_t *b, uint64_t len, int32_t *n_ref, uint64_t *first)
{
    if (len < 12 || memcmp(b, "BAM\1", 4)) return -1;
    uint64_t p = 8 + (uint64_t)le32(b + 4);
    if (p + 4 > len) return -1;
    int32_t n = (int32_t)le32(b + p); p += 4;
    if (n < 0) return -1;
    for (int32_t i = 0; i < n; i++) {
        if (p + 4 > len) return -1;
        int32_t l = (int32_t)le32(b + p); p += 4;
        if (l <= 0 || p + (uint64_t)l + 4 > len) return -1;
        p += (uint64_t)l + 4;
    }
    *n_ref = n; *first = p;
    return 0;
}

/* rec_off[i] = offset of record i's block_len field.  Returns the number of records, -2 if the stream ends inside a
 * record, -4 if a record fails bam_read1's checks; *bad = offset of the offending record. */
ORC_EXPORT long orc_bam_frame(const uint8_t *b, uint64_t len, uint64_t first, uint64_t *rec_off, long max_rec, uint64_t *bad)
{
    uint64_t p = first; long n = 0;
    while (p < len) {
        *bad = p;
        if (p + 4 > len) return -2;
        int32_t bl = (int32_t)le32(b + p);
        if (bl < 32) return -4;
        if (p + 4 + (uint64_t)bl > len) return (p + 36 > len) ? -2 : -2;
        const uint8_t *x = b + p + 4;
        uint32_t x2 = le32(x + 8), x3 = le32(x + 12);
        uint32_t l_qname = x2 & 0xff, n_cigar = x3 & 0xffff;
        int32_t l_qseq = (int32_t)le32(x + 16);
        if (l_qseq < 0 || l_qname < 1) return -4;
        if (((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_qseq + 1) >> 1) + (uint64_t)l_qseq > (uint64_t)(bl - 32)) return -4;
        if (rec_off && n < max_rec) rec_off[n] = p;
        n++;
        p += 4 + (uint64_t)bl;
    }
    return n;
}

ORC_EXPORT void orc_nibble2base(const uint8_t *nib, char *seq, int len)
{
    static const char code[] = "=ACMGRSVTWYHKDBN";
    for (int i = 0; i < len; i++) seq[i] = code[(nib[i >> 1] >> ((~i & 1) << 2)) & 0xf];
}

/* ------------------------------------------------------------------------------------------------------------------
 * BAI construction: a restatement of what `samtools index` does --
 *   sam_index (sam.c:994-1031): for every record hts_idx_push(tid, pos, bam_endpos, bgzf_tell AFTER the record, mapped)
 *   hts_idx_push / insert_to_b / insert_to_l (hts.c:2320-2365, 2558-2640), hts_idx_finish, update_loff,
 *   compress_binning (hts.c:2431-2536), idx_save_core (hts.c:2759-2822), hts_reg2bin (htslib/hts.h:1516-1523),
 *   bam_endpos (sam.c:673-678).
 * Output: the .bai byte stream with ONE normalisation -- the bins of a reference are written in ascending bin order
 * (htslib writes them in the iteration order of its hash table, which carries no meaning).  Pinned by
 * tests/test_bam_index.py: equal, after the same normalisation, to the .bai files reference htslib wrote for its own
 * BAM fixtures.
 * blk[] = every BGZF block of the file in order: (compressed offset, uncompressed offset, uncompressed length),
 * including empty ones and the EOF block; file_size = compressed size (the virtual offset after the last block).
 * ------------------------------------------------------------------------------------------------------------------ */
#include <stdlib.h>
typedef struct { uint64_t coff, uoff; uint32_t ulen; uint32_t pad; } oblk_t;
typedef struct { uint64_t u, v; } opair_t;
typedef struct { uint32_t bin; opair_t *list; uint32_t n, m; } obin_t;
typedef struct { obin_t *bins; uint32_t n, m; uint64_t *lin; int64_t lin_n, lin_m; } oref_t;

static uint64_t voff_of(const oblk_t *blk, long nb, uint64_t file_size, uint64_t u)
{
    /* the block bgzf_tell reports: the first block starting at u if there is one (bgzf_read moves to the next block as
     * soon as the current one is used up, bgzf.c:1276-1281), else the block containing u */
    long lo = 0, hi = nb;                                   /* lower_bound on uoff */
    while (lo < hi) { long mid = (lo + hi) / 2; if (blk[mid].uoff < u) lo = mid + 1; else hi = mid; }
    if (lo < nb && blk[lo].uoff == u) return blk[lo].coff << 16;
    if (lo == 0) return 0;
    if (lo == nb && u >= blk[nb - 1].uoff + blk[nb - 1].ulen) return file_size << 16;
    return (blk[lo - 1].coff << 16) | (u - blk[lo - 1].uoff);
}
static int reg2bin(int64_t beg, int64_t end, int min_shift, int n_lvls)
{
    int l, s = min_shift, t = ((1 << (3 * n_lvls)) - 1) / 7;
    for (--end, l = n_lvls; l > 0; --l, s += 3, t -= 1 << (3 * l))
        if (beg >> s == end >> s) return t + (int)(beg >> s);
    return 0;
}
static obin_t *bin_get(oref_t *r, uint32_t bin, int create)
{
    for (uint32_t i = 0; i < r->n; i++) if (r->bins[i].bin == bin) return &r->bins[i];
    if (!create) return NULL;
    if (r->n == r->m) { r->m = r->m ? r->m * 2 : 16; r->bins = realloc(r->bins, r->m * sizeof(obin_t)); }
    obin_t *b = &r->bins[r->n++];
    b->bin = bin; b->list = NULL; b->n = b->m = 0;
    return b;
}
static void bin_add(oref_t *r, uint32_t bin, uint64_t u, uint64_t v)
{
    obin_t *b = bin_get(r, bin, 1);
    if (b->n == b->m) { b->m = b->m ? b->m * 2 : 4; b->list = realloc(b->list, b->m * sizeof(opair_t)); }
    b->list[b->n].u = u; b->list[b->n++].v = v;
}
static int cmp_pair(const void *a, const void *b) { uint64_t x = ((const opair_t *)a)->u, y = ((const opair_t *)b)->u; return x < y ? -1 : x > y; }
static int cmp_bin(const void *a, const void *b) { uint32_t x = ((const obin_t *)a)->bin, y = ((const obin_t *)b)->bin; return x < y ? -1 : x > y; }
static int bin_level(uint32_t bin) { int l = 0; while (bin) { bin = (bin - 1) >> 3; l++; } return l; }

/* returns the number of bytes written to out (<= cap), or -1 (unsorted input / chromosome blocks not continuous / bad
 * record) like `samtools index` failing */
ORC_EXPORT long orc_idx_build(const uint8_t *b, uint64_t len, uint64_t first, int32_t n_ref, const oblk_t *blk, long nb,
                              uint64_t file_size, int csi, int min_shift, int n_lvls, uint8_t *out, long cap);
ORC_EXPORT long orc_bai_build(const uint8_t *b, uint64_t len, uint64_t first, int32_t n_ref, const oblk_t *blk, long nb,
                              uint64_t file_size, uint8_t *out, long cap)
{
    return orc_idx_build(b, len, first, n_ref, blk, nb, file_size, 0, 14, 5, out, cap);
}
/* csi != 0: the CSI layout (hts_idx_write_out / idx_save_core with HTS_FMT_CSI: min_shift, depth, l_meta = 0, a
 * linear-index-derived `loff` per bin instead of the linear index itself), uncompressed -- samtools BGZF-compresses it.
 * n_lvls as sam_index derives it with hts_adjust_csi_settings (hts.c:2367-2403). */
ORC_EXPORT long orc_idx_build(const uint8_t *b, uint64_t len, uint64_t first, int32_t n_ref, const oblk_t *blk, long nb,
                              uint64_t file_size, int csi, int min_shift, int n_lvls, uint8_t *out, long cap)
{
    const uint32_t N_BINS = (uint32_t)(((1ull << (3 * n_lvls + 3)) - 1) / 7), META = N_BINS + 1;
    oref_t *R = calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(oref_t));
    int32_t save_tid = -1, last_tid = -1; uint32_t save_bin = 0xffffffffu, last_bin = 0xffffffffu;
    uint64_t offset0 = voff_of(blk, nb, file_size, first);
    uint64_t save_off = offset0, last_off = offset0, off_beg = offset0, off_end = offset0, n_mapped = 0, n_unmapped = 0, n_no_coor = 0;
    int64_t last_coor = 0xffffffffu;
    int rc = 0;
    uint64_t p = first;
    while (p < len && !rc) {
        if (p + 4 > len) { rc = -1; break; }
        int32_t bl = (int32_t)le32(b + p);
        if (bl < 32 || p + 4 + (uint64_t)bl > len) { rc = -1; break; }
        const uint8_t *x = b + p + 4;
        int32_t tid = (int32_t)le32(x); int64_t beg = (int32_t)le32(x + 4);
        uint32_t l_qname = le32(x + 8) & 0xff, x3 = le32(x + 12), n_cigar = x3 & 0xffff, flag = x3 >> 16;
        int64_t rlen = 0;
        if (!(flag & 4)) for (uint32_t k = 0; k < n_cigar; k++) {      /* bam_cigar2rlen: M D N = X consume the reference */
            uint32_t c = le32(x + 32 + l_qname + 4 * k), op = c & 0xf;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;
        }
        if (rlen == 0) rlen = 1;
        int64_t end = beg + rlen;
        uint64_t offset = voff_of(blk, nb, file_size, p + 4 + (uint64_t)bl);     /* bgzf_tell after the record */
        int is_mapped = !(flag & 4);
        /* ---- hts_idx_push ---- */
        if (tid < 0) { beg = -1; end = 0; }
        if (tid >= n_ref) { rc = -1; break; }
        if (last_tid != tid || (last_tid >= 0 && tid < 0)) {
            if (tid >= 0 && n_no_coor) { rc = -1; break; }
            if (tid >= 0 && (R[tid].n || R[tid].lin_n)) { rc = -1; break; }       /* chromosome blocks not continuous */
            last_tid = tid; last_bin = 0xffffffffu;
        } else if (tid >= 0 && last_coor > beg) { rc = -1; break; }               /* unsorted positions */
        if (end < beg) { rc = -1; break; }
        if (tid >= 0) {
            if (beg < 0) beg = 0;
            if (end <= 0) end = 1;
            oref_t *r = &R[tid];
            int64_t wb = beg >> min_shift, we = (end - 1) >> min_shift;
            if (r->lin_m < we + 1) {
                int64_t nm = r->lin_m * 2 > we + 1 ? r->lin_m * 2 : we + 1;
                r->lin = realloc(r->lin, (size_t)nm * 8);
                for (int64_t i = r->lin_m; i < nm; i++) r->lin[i] = ~0ull;
                r->lin_m = nm;
            }
            for (int64_t i = wb; i <= we; i++) if (r->lin[i] == ~0ull) r->lin[i] = last_off;
            if (r->lin_n < we + 1) r->lin_n = we + 1;
        } else n_no_coor++;
        int bin = reg2bin(beg, end, min_shift, n_lvls);
        if ((int)last_bin != bin) {
            if (save_bin != 0xffffffffu) bin_add(&R[save_tid], save_bin, save_off, last_off);
            if (last_bin == 0xffffffffu && save_bin != 0xffffffffu) {              /* change of reference: its meta bin */
                off_end = last_off;
                bin_add(&R[save_tid], META, off_beg, off_end);
                bin_add(&R[save_tid], META, n_mapped, n_unmapped);
                n_mapped = n_unmapped = 0; off_beg = off_end;
            }
            save_off = last_off; save_bin = last_bin = (uint32_t)bin; save_tid = tid;
        }
        if (is_mapped) n_mapped++; else n_unmapped++;
        last_off = offset; last_coor = beg;
        p += 4 + (uint64_t)bl;
    }
    long written = -1;
    if (!rc) {
        /* ---- hts_idx_finish ---- */
        uint64_t final_offset = voff_of(blk, nb, file_size, len);
        if (save_tid >= 0) {
            bin_add(&R[save_tid], save_bin, save_off, final_offset);
            bin_add(&R[save_tid], META, off_beg, final_offset);
            bin_add(&R[save_tid], META, n_mapped, n_unmapped);
        }
        for (int32_t t = 0; t < n_ref; t++) {
            oref_t *r = &R[t];
            for (int64_t l = r->lin_n - 2; l >= 0; l--) if (r->lin[l] == ~0ull) r->lin[l] = r->lin[l + 1];
            /* compress_binning: a bin spanning < 0x10000 compressed bytes moves into its parent, bottom level first */
            for (int l = n_lvls; l > 0; l--)
                for (uint32_t i = 0; i < r->n; i++) {
                    obin_t *pb = &r->bins[i];
                    if (pb->bin >= N_BINS || !pb->n || bin_level(pb->bin) != l) continue;
                    if (l < n_lvls && pb->n > 1) qsort(pb->list, pb->n, sizeof(opair_t), cmp_pair);
                    if ((pb->list[pb->n - 1].v >> 16) - (pb->list[0].u >> 16) < 0x10000) {
                        obin_t *q = bin_get(r, (pb->bin - 1) >> 3, 0);
                        if (!q || !q->n) continue;
                        pb = &r->bins[i];
                        for (uint32_t k = 0; k < pb->n; k++) bin_add(r, q->bin, pb->list[k].u, pb->list[k].v);
                        r->bins[i].n = 0;                                           /* deleted */
                    }
                }
            obin_t *b0 = bin_get(r, 0, 0);
            if (b0 && b0->n > 1) qsort(b0->list, b0->n, sizeof(opair_t), cmp_pair);
            for (uint32_t i = 0; i < r->n; i++) {                                   /* merge chunks that touch the same block */
                obin_t *pb = &r->bins[i];
                if (pb->bin >= N_BINS || !pb->n) continue;
                uint32_t m = 0;
                for (uint32_t l = 1; l < pb->n; l++) {
                    if (pb->list[m].v >> 16 >= pb->list[l].u >> 16) { if (pb->list[m].v < pb->list[l].v) pb->list[m].v = pb->list[l].v; }
                    else pb->list[++m] = pb->list[l];
                }
                pb->n = m + 1;
            }
        }
        /* ---- idx_save_core, bins in ascending order ---- */
        uint8_t *o = out, *oe = out + cap;
#define PUT32(v) do { if (o + 4 > oe) goto full; uint32_t _v = (uint32_t)(v); o[0] = _v; o[1] = _v >> 8; o[2] = _v >> 16; o[3] = _v >> 24; o += 4; } while (0)
#define PUT64(v) do { uint64_t _w = (v); PUT32(_w); PUT32(_w >> 32); } while (0)
        if (o + 4 > oe) goto full;
        memcpy(o, csi ? "CSI\1" : "BAI\1", 4); o += 4;
        if (csi) { PUT32(min_shift); PUT32(n_lvls); PUT32(0); }
        PUT32(n_ref);
        for (int32_t t = 0; t < n_ref; t++) {
            oref_t *r = &R[t];
            uint32_t live = 0;
            qsort(r->bins, r->n, sizeof(obin_t), cmp_bin);
            for (uint32_t i = 0; i < r->n; i++) live += r->bins[i].n != 0;
            PUT32(live);
            for (uint32_t i = 0; i < r->n; i++) {
                obin_t *pb = &r->bins[i];
                if (!pb->n) continue;
                PUT32(pb->bin);
                if (csi) {                                                          /* update_loff, hts.c:2443-2453 */
                    uint64_t loff = 0;
                    if (pb->bin < N_BINS) {
                        int l = bin_level(pb->bin);
                        int64_t bot = (int64_t)(pb->bin - (uint32_t)(((1ull << (3 * l)) - 1) / 7)) << ((n_lvls - l) * 3);
                        loff = bot < r->lin_n ? r->lin[bot] : 0;
                    }
                    PUT64(loff);
                }
                PUT32(pb->n);
                for (uint32_t k = 0; k < pb->n; k++) { PUT64(pb->list[k].u); PUT64(pb->list[k].v); }
            }
            if (!csi) {
                PUT32((uint32_t)r->lin_n);
                for (int64_t i = 0; i < r->lin_n; i++) PUT64(r->lin[i]);
            }
        }
        PUT64(n_no_coor);
        written = (long)(o - out);
    }
full:
    for (int32_t t = 0; t < n_ref; t++) { for (uint32_t i = 0; i < R[t].n; i++) free(R[t].bins[i].list); free(R[t].bins); free(R[t].lin); }
    free(R);
    return written;
}

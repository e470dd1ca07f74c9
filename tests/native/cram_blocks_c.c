/* TEST DRIVER (plain C, the way code inside htslib would call the block layer; include/hts_cram_gpu.h).
 *   cram_blocks_c <blocks.bin> <out.bin> <array|single|threads>
 * blocks.bin = CRAM v3 blocks back to back exactly as they sit in a .cram file (header, payload, CRC-32).
 * Reads them with hg_cram_read_block, re-writes them with hg_cram_write_block (must reproduce the input bytes),
 * decodes them with cram_uncompress_block -- all at once, one by one, or from 8 threads at the same time (the calling
 * pattern of htslib's pool workers, which the block layer coalesces into device batches) -- then compresses every
 * decoded block again through the auto-tuner and decodes that.  out.bin: per block  int32 rc, int32 size, bytes. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hts_cram_gpu.h"
#include "hts_hfile_abi.h"

static cram_block **blk; static int nblk; static int *rcs;
struct share { int from, to; };
static void *worker(void *v) {
    struct share *s = v;
    for (int i = s->from; i < s->to; i++) rcs[i] = cram_uncompress_block(blk[i]);
    return NULL;
}
static int fail(const char *m) { fprintf(stderr, "cram_blocks_c: %s\n", m); return 1; }

int main(int argc, char **argv) {
    if (argc != 4) return fail("usage");
    hFILE *in = hopen(argv[1], "r");
    if (!in) return fail("cannot open input");
    int cap = 0;
    for (;;) {
        cram_block *b = hg_cram_read_block(in, 3, 0);
        if (!b) break;
        if (nblk == cap) { cap = cap ? cap * 2 : 256; blk = realloc(blk, cap * sizeof(*blk)); }
        blk[nblk++] = b;
    }
    if (hclose(in) != 0 || nblk == 0) return fail("no blocks read");
    /* framing round trip: byte-identical re-serialisation, CRCs recomputed on the device */
    char tmpn[4096]; snprintf(tmpn, sizeof tmpn, "%s.rewrite", argv[2]);
    hFILE *rw = hopen(tmpn, "w");
    if (!rw) return fail("cannot open rewrite file");
    for (int i = 0; i < nblk; i++) {
        uint32_t want = blk[i]->crc32;
        if (hg_cram_write_block(rw, 3, blk[i]) != 0) return fail("hg_cram_write_block failed");
        if (blk[i]->crc32 != want) { if (!getenv("CRAMC_ALLOW_BAD_CRC")) return fail("recomputed block CRC differs from the file's"); blk[i]->crc32 = want; }
    }
    if (hclose(rw) != 0) return fail("close");
    rcs = calloc(nblk, sizeof(int));
    if (!strcmp(argv[3], "array")) {
        cram_uncompress_blocks(blk, nblk, rcs);
    } else if (!strcmp(argv[3], "single")) {
        for (int i = 0; i < nblk; i++) rcs[i] = cram_uncompress_block(blk[i]);
    } else {
        enum { NT = 8 };
        pthread_t th[NT]; struct share sh[NT];
        for (int t = 0; t < NT; t++) { sh[t].from = (int)((long)nblk * t / NT); sh[t].to = (int)((long)nblk * (t + 1) / NT); pthread_create(&th[t], NULL, worker, &sh[t]); }
        for (int t = 0; t < NT; t++) pthread_join(th[t], NULL);
    }
    FILE *out = fopen(argv[2], "wb");
    if (!out) return fail("cannot open output");
    for (int i = 0; i < nblk; i++) {
        int32_t hdr[2] = {rcs[i], rcs[i] == 0 ? blk[i]->uncomp_size : 0};
        fwrite(hdr, 4, 2, out);
        if (rcs[i] == 0) {
            if (blk[i]->method != RAW) return fail("decoded block is not RAW");
            if (blk[i]->uncomp_size) fwrite(blk[i]->data, 1, (size_t)blk[i]->uncomp_size, out);
        }
        /* idempotence (cram_io.c:1594-1603) */
        if (rcs[i] == 0 && cram_uncompress_block(blk[i]) != 0) return fail("second cram_uncompress_block on a RAW block failed");
    }
    fclose(out);
    /* compress every decoded block through the auto-tuner (one metrics object per content id), decode again */
    hg_cram_opts opts = {5, (3 << 8) | 1, 0, 0, NULL};
    cram_metrics *met[64]; for (int k = 0; k < 64; k++) met[k] = cram_new_metrics();
    int ngood = 0;
    cram_block **cb = calloc(nblk, sizeof(*cb)); cram_metrics **cm = calloc(nblk, sizeof(*cm)); int *set = calloc(nblk, sizeof(int));
    unsigned char **orig = calloc(nblk, sizeof(*orig));
    for (int i = 0; i < nblk; i++) {
        if (rcs[i] != 0) continue;
        orig[ngood] = malloc(blk[i]->uncomp_size ? blk[i]->uncomp_size : 1);
        memcpy(orig[ngood], blk[i]->data, blk[i]->uncomp_size);
        blk[i]->comp_size = blk[i]->uncomp_size;
        cb[ngood] = blk[i]; cm[ngood] = met[(unsigned)blk[i]->content_id % 64];
        set[ngood] = (1 << GZIP) | (1 << RANS_PR0) | (1 << RANS_PR1) | (1 << RANS_PR64) | (1 << RANS0);
        ngood++;
    }
    if (hg_cram_compress_blocks(&opts, cb, cm, set, -1, ngood) != 0) return fail("hg_cram_compress_blocks failed");
    long raw_bytes = 0, comp_bytes = 0; int nraw = 0;
    for (int i = 0; i < ngood; i++) {
        raw_bytes += cb[i]->uncomp_size; comp_bytes += cb[i]->comp_size; nraw += cb[i]->method == RAW;
        if (cb[i]->method != RAW && cb[i]->method != GZIP && cb[i]->method != RANS && cb[i]->method != RANSPR) return fail("unexpected on-disk method");
        if (cb[i]->comp_size > cb[i]->uncomp_size) return fail("compressed block larger than raw");
        cb[i]->crc32_checked = 1;
    }
    int *rc2 = calloc(ngood, sizeof(int));
    if (cram_uncompress_blocks(cb, ngood, rc2) != 0) return fail("decode of re-compressed blocks failed");
    for (int i = 0; i < ngood; i++)
        if (cb[i]->method != RAW || memcmp(cb[i]->data, orig[i], (size_t)cb[i]->uncomp_size)) return fail("compress -> uncompress round trip differs");
    printf("blocks %d decoded_ok %d recompressed %ld -> %ld bytes (%d kept raw)\n", nblk, ngood, raw_bytes, comp_bytes, nraw);
    for (int i = 0; i < nblk; i++) cram_free_block(blk[i]);
    return 0;
}

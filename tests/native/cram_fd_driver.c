/* TEST DRIVER compiled against the REFERENCE's headers (-I/root/reference: the real struct cram_fd / cram_slice / cram_record) and linked
 * to OUR libhts_bgzf.so: the reference-named entry points cram_write_block / cram_read_block / cram_compress_block / cram_compress_block2 /
 * cram_uncompress_block (cram/cram_io.c:1414, 1511, 1576, 2316-2325) called the way code inside htslib calls them, on a real cram_fd.
 *   cram_fd_driver <scratch file>
 * Writes three blocks through cram_fd, reads them back, compresses them (block 2 with the FQZ methods and a real cram_slice), writes, reads,
 * uncompresses, compares.  Built by oracle/Makefile (dropin) into oracle/_ref/, run by tests/test_cram_block_front.py on the GPU. */
#include <config.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "cram/cram.h"
#include "htslib/hfile.h"

static int fail(const char *m) { fprintf(stderr, "cram_fd_driver: %s\n", m); return 1; }

int main(int argc, char **argv) {
    if (argc != 2) return fail("usage");
    cram_fd *fd = calloc(1, sizeof(*fd));
    fd->version = 3 << 8 | 1; fd->level = 5; fd->use_rans = 1; fd->use_fqz = 1;
    pthread_mutex_init(&fd->metrics_lock, NULL);
    enum { NREC = 500, RL = 80 };
    static unsigned char qs[NREC * RL], small[4000], names[NREC * 12];
    unsigned x = 12345; int q = 30, nn = 0;
    for (int i = 0; i < NREC * RL; i++) { x = x * 1103515245u + 12345u; if ((x >> 16) % 9 == 0) q = 2 + (x >> 20) % 40; qs[i] = (unsigned char)q; }
    for (int i = 0; i < 4000; i++) { x = x * 1103515245u + 12345u; small[i] = (x >> 16) % 5; }
    for (int r = 0; r < NREC; r++) nn += sprintf((char *)names + nn, "read%05d", r) + 1;
    const unsigned char *src[3] = {small, qs, names}; const int len[3] = {sizeof small, sizeof qs, nn}; const int cid[3] = {DS_BF, DS_QS, DS_RN};
    cram_block *b[3];
    for (int i = 0; i < 3; i++) {
        b[i] = cram_new_block(EXTERNAL, cid[i]);
        b[i]->data = malloc(len[i]); memcpy(b[i]->data, src[i], len[i]); b[i]->uncomp_size = b[i]->comp_size = len[i]; b[i]->byte = b[i]->alloc = len[i];
    }
    /* a real cram_slice for the quality block's FQZ methods: records with their qual offsets and flags */
    cram_slice s; memset(&s, 0, sizeof s);
    cram_block_slice_hdr hdr; memset(&hdr, 0, sizeof hdr); hdr.num_records = NREC; s.hdr = &hdr;
    s.crecs = calloc(NREC, sizeof(cram_record));
    for (int r = 0; r < NREC; r++) { s.crecs[r].qual = r * RL; s.crecs[r].len = RL; s.crecs[r].flags = (r & 1) ? 16 : 0; }
    cram_block *blocks[DS_END]; memset(blocks, 0, sizeof blocks); blocks[DS_QS] = b[1]; s.block = blocks;
    cram_metrics *m[3] = {cram_new_metrics(), cram_new_metrics(), cram_new_metrics()};
    if (cram_compress_block(fd, b[0], m[0], 1 << GZIP | 1 << RANS_PR0 | 1 << RANS_PR1, -1) != 0) return fail("cram_compress_block");
    if (cram_compress_block2(fd, &s, b[1], m[1], 1 << GZIP | 1 << RANS_PR1 | 1 << FQZ | 1 << FQZ_b, -1) != 0) return fail("cram_compress_block2 (FQZ)");
    if (cram_compress_block2(fd, &s, b[2], m[2], 1 << GZIP | 1 << TOK3, -1) != 0) return fail("cram_compress_block2 (TOK3)");
    /* a data series the slice does not have: cram_compress_slice passes NULL and expects "nothing to do" (cram_io.c:1917-1918) */
    if (cram_compress_block2(fd, &s, NULL, m[0], 1 << GZIP, -1) != 0 || cram_compress_block(fd, NULL, NULL, -1, -1) != 0) return fail("NULL block must be a no-op returning 0");
    /* an already compressed block is left alone (cram_io.c:1945-1952) */
    { int cs = b[0]->comp_size, me = b[0]->method; if (cram_compress_block(fd, b[0], m[0], 1 << GZIP, -1) != 0 || b[0]->comp_size != cs || b[0]->method != me) return fail("compressed block touched"); }
    printf("methods %d %d %d sizes %d %d %d\n", b[0]->method, b[1]->method, b[2]->method, b[0]->comp_size, b[1]->comp_size, b[2]->comp_size);
    fd->fp = hopen(argv[1], "w");
    if (!fd->fp) return fail("hopen w");
    for (int i = 0; i < 3; i++) if (cram_write_block(fd, b[i]) != 0) return fail("cram_write_block");
    if (hclose(fd->fp) != 0) return fail("hclose");
    fd->fp = hopen(argv[1], "r");
    if (!fd->fp) return fail("hopen r");
    for (int i = 0; i < 3; i++) {
        cram_block *r = cram_read_block(fd);
        if (!r) return fail("cram_read_block");
        if (r->content_id != cid[i] || r->uncomp_size != len[i] || r->method != b[i]->method) return fail("block header differs");
        if (cram_uncompress_block(r) != 0) return fail("cram_uncompress_block");
        if (r->uncomp_size != len[i] || memcmp(r->data, src[i], len[i]) != 0) return fail("plaintext differs");
        cram_free_block(r);
    }
    if (cram_read_block(fd) != NULL) return fail("a fourth block?");
    hclose_abruptly(fd->fp);
    printf("cram_fd entry points ok\n");
    return 0;
}

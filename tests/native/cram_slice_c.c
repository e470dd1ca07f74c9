/* TEST DRIVER (plain C): cram_compress_slice for ONE slice in one engine batch -- hg_cram_compress_slice_fqz (include/hts_cram_gpu.h; reference
 * cram/cram_encode.c:803-988), the way cram_encode_slice would call it: the blocks of the data series by DS id, two per-tag aux blocks, the
 * slice's fqz_slice for DS_QS, one cram_metrics per series.
 *   cram_slice_c <level> <version major> <use_fqz> <use_arith> <nslices>
 * Synthetic series (qualities with record structure, names, bases, small integer columns) are compressed slice after slice with the same
 * metrics objects (the auto-tuner learns across slices), every block is then decoded with cram_uncompress_block and compared with what
 * went in.  Prints per slice and series: ds, on-disk method, sizes; "methods_offered_ok" when every chosen method belongs to the set
 * hg_cram_slice_plan names for that series (or to methodF of the final sweep). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hts_cram_gpu.h"

static unsigned long long rs = 88172645463325252ull;
static unsigned rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 11); }
static int fail(const char *m) { fprintf(stderr, "cram_slice_c: %s\n", m); return 1; }

static cram_block *make(int ds, const unsigned char *p, int n) {
    cram_block *b = cram_new_block(EXTERNAL, ds);
    if (!b) return NULL;
    b->data = malloc(n ? n : 1); memcpy(b->data, p, n); b->alloc = n; b->byte = n; b->uncomp_size = n; b->comp_size = n; b->method = RAW; b->orig_method = RAW;
    return b;
}
/* on-disk method of an internal one (cram_io.c:1928-1943) */
static int disk(int m) {
    if (m == RAW || m == GZIP || m == BZIP2 || m == LZMA || m == RANS0) return m;
    if (m == GZIP_RLE || m == GZIP_1) return GZIP;
    if (m == RANS1) return RANS0;
    if (m == FQZ || m == FQZ_b || m == FQZ_c || m == FQZ_d) return FQZ;
    if (m == TOK3 || m == TOKA) return TOK3;
    if (m >= RANS_PR0 && m <= RANS_PR0) return RANSPR;
    if (m >= RANS_PR1 && m <= RANS_PR193) return RANSPR;
    if (m == ARITH_PR0 || (m >= ARITH_PR1 && m <= ARITH_PR193)) return ARITH;
    return -1;
}
static int set_allows(int set, int on_disk) { for (int m = 0; m < 32; m++) if ((set >> m & 1) && disk(m) == on_disk) return 1; return on_disk == RAW; }

int main(int argc, char **argv) {
    if (argc != 6) return fail("usage");
    hg_cram_slice_opts so; memset(&so, 0, sizeof so);
    so.level = atoi(argv[1]); so.version = atoi(argv[2]) << 8 | (atoi(argv[2]) >= 3 ? 1 : 0); so.use_rans = 1; so.use_tok = atoi(argv[2]) >= 3;
    so.use_fqz = atoi(argv[3]) & 1; so.use_arith = atoi(argv[4]); so.use_bz2 = so.use_lzma = (atoi(argv[3]) >> 1) & 1;     /* use_fqz argument: bit 1 = also bzip2 + lzma */
    const int nslices = atoi(argv[5]);
    hg_cram_opts op; memset(&op, 0, sizeof op); op.level = so.level; op.version = so.version; op.use_bz2 = so.use_bz2; op.use_lzma = so.use_lzma;
    cram_metrics *metrics[HG_DS_END]; memset(metrics, 0, sizeof metrics);
    for (int i = 1; i < HG_DS_END; i++) metrics[i] = cram_new_metrics();
    cram_metrics *auxm[2] = {cram_new_metrics(), cram_new_metrics()};
    hg_cram_slice_sets sets; hg_cram_slice_method_sets(&so, &sets);
    int bad = 0, offered_ok = 1;
    enum { NREC = 3000, RL = 100 };
    for (int sl = 0; sl < nslices; sl++) {
        /* the series of one slice */
        static unsigned char qs[NREC * RL], ba[NREC * RL], rn[NREC * 24], small[NREC], tagv[NREC * 4], tagz[NREC * 12];
        static unsigned lens[NREC], flags[NREC];
        int q = 30, nn = 0, nz = 0;
        for (int r = 0; r < NREC; r++) {
            lens[r] = RL; flags[r] = (rnd() & 1 ? 16 : 0) | (r & 1 ? 128 : 64);
            for (int i = 0; i < RL; i++) { if (rnd() % 10 == 0) q = "\2\14\27\45"[rnd() & 3]; qs[r * RL + i] = (unsigned char)q; ba[r * RL + i] = "ACGT"[rnd() & 3]; }
            nn += sprintf((char *)rn + nn, "SIM:1:FC%02d:%d:%d", sl, 1101 + r / 100, 1000 + (int)(rnd() % 9000)) + 1;
            small[r] = (unsigned char)(rnd() % 3);
            tagv[4 * r] = (unsigned char)(rnd() % 7); tagv[4 * r + 1] = tagv[4 * r + 2] = tagv[4 * r + 3] = 0;
            nz += sprintf((char *)tagz + nz, "%dM", 50 + (int)(rnd() % 50)) + 1;
        }
        cram_block *block[HG_DS_END]; memset(block, 0, sizeof block);
        int nvals[HG_DS_END]; memset(nvals, 0, sizeof nvals);
        block[HG_DS_CORE] = cram_new_block(CORE, 0);
        block[HG_DS_QS] = make(HG_DS_QS, qs, sizeof qs); nvals[HG_DS_QS] = 4;
        block[HG_DS_BA] = make(HG_DS_BA, ba, sizeof ba); nvals[HG_DS_BA] = 4;
        block[HG_DS_RN] = make(HG_DS_RN, rn, nn); nvals[HG_DS_RN] = 64;
        block[HG_DS_IN] = make(HG_DS_IN, small, sizeof small); nvals[HG_DS_IN] = 3;
        block[HG_DS_NS] = make(HG_DS_NS, small, 200); nvals[HG_DS_NS] = 3;
        block[HG_DS_BB] = make(HG_DS_BB, ba, 5000); nvals[HG_DS_BB] = 4;
        cram_block *aux[2] = {make(0x4e4d69 /* NMi */, tagv, sizeof tagv), make(0x4d445a /* MDZ */, tagz, nz)};
        aux[0]->m = auxm[0]; aux[1]->m = auxm[1];
        const hg_fqz_slice fq = {NREC, lens, flags};
        /* what went in, and what the policy offers */
        struct { cram_block *b; unsigned char *copy; int n, ds, set; } chk[16]; int nchk = 0;
        unsigned char present[HG_DS_END]; for (int i = 0; i < HG_DS_END; i++) present[i] = block[i] != NULL;
        int pds[HG_DS_END + 16], pset[HG_DS_END + 16], plv[HG_DS_END + 16];
        const int np = hg_cram_slice_plan(&so, present, 2, 0, pds, pset, plv, HG_DS_END + 16);
        for (int i = 1; i < HG_DS_END + 2; i++) {
            cram_block *b = i < HG_DS_END ? block[i] : aux[i - HG_DS_END];
            if (!b) continue;
            int set = sets.methodF;
            for (int k = 0; k < np; k++) if (pds[k] == i) set |= pset[k];
            chk[nchk].b = b; chk[nchk].n = b->uncomp_size; chk[nchk].copy = malloc(b->uncomp_size + 1); memcpy(chk[nchk].copy, b->data, b->uncomp_size); chk[nchk].ds = i; chk[nchk].set = set; nchk++;
        }
        if (hg_cram_compress_slice_fqz(&so, &op, block, metrics, nvals, aux, 2, so.use_fqz ? &fq : NULL) != 0) return fail("hg_cram_compress_slice_fqz failed");
        for (int k = 0; k < nchk; k++) {
            cram_block *b = chk[k].b;
            const int method = b->method, csz = b->comp_size;
            if (!set_allows(chk[k].set, method)) { offered_ok = 0; fprintf(stderr, "ds %d: method %d not in set %x\n", chk[k].ds, method, chk[k].set); }
            printf("slice %d ds %d method %d size %d -> %d\n", sl, chk[k].ds, method, chk[k].n, csz);
            b->crc32_checked = 1;                                            /* not read from a file: no stored CRC to check (cram_io.c:1585-1592) */
            if (cram_uncompress_block(b) != 0 || b->uncomp_size != chk[k].n || memcmp(b->data, chk[k].copy, chk[k].n) != 0) { bad++; fprintf(stderr, "ds %d: does not decode back\n", chk[k].ds); }
            free(chk[k].copy); cram_free_block(b);
        }
        cram_free_block(block[HG_DS_CORE]);
    }
    printf("undecodable %d\n%s\n", bad, offered_ok ? "methods_offered_ok" : "methods_offered_BAD");
    return bad || !offered_ok;
}

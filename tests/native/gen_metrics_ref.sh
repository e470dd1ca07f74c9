#!/bin/bash
# TEST INFRASTRUCTURE.  Runs the REFERENCE's own cram_compress_block3 (the method auto-tuner, cram/cram_io.c) over a
# scripted sequence of blocks with a cram_compress_by_method that does not compress but returns the size a script names.
# The static function's text is spliced from the reference source at build time into a scratch file (never into the
# repository).   usage: gen_metrics_ref.sh <scratch dir> <level> <version> <nblocks> <method set>  -> <scratch dir>/ref_metrics.txt
set -e
REF=${REF:-/root/reference}; OUT=$1; ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$REF/cram/cram_io.c
START=$(grep -n '^static int cram_compress_block3(cram_fd \*fd, cram_slice \*s,' $SRC | cut -d: -f1)
END=$(awk -v s=$START 'NR>s && /^}/ {print NR; exit}' $SRC)
[ -n "$START" ] && [ -n "$END" ] || { echo "cram_compress_block3 not found" >&2; exit 1; }
{
cat <<'C1'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include <zlib.h>
#include "cram/cram.h"
#include "htslib/hts_log.h"
#define HAVE_LIBDEFLATE 1
C1
grep -E '^#define (TRIAL_SPAN|NTRIALS|CRAM_DEFAULT_LEVEL) ' $SRC
cat <<'C2'
static long cur_k;
static uint32_t script(int method, long k, uint32_t in_len) {          /* the same function as tests/test_cram_metrics_reference.py */
    uint32_t h = (uint32_t)method * 2654435761u + (uint32_t)k * 40503u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    uint32_t frac = 250u + h % 900u;
    if (k % 4 == 2) frac = 1000u + h % 200u;                          /* series 2 never shrinks: its learnt method is RAW */
    uint64_t sz = (uint64_t)in_len * frac / 1000u;
    return sz ? (uint32_t)sz : 1u;
}
static char *cram_compress_by_method(cram_slice *s, char *in, size_t in_size, int content_id, size_t *out_size,
                                     enum cram_block_method_int method, int level, int strat) {
    (void)s; (void)in; (void)content_id; (void)level; (void)strat;
    if (method == RAW) return NULL;                                   /* as the real function (cram_io.c:1896-1903) */
    *out_size = script((int)method, cur_k, (uint32_t)in_size);
    return calloc(*out_size, 1);
}
char *cram_block_method2str(enum cram_block_method_int m) { (void)m; return "?"; }
C2
sed -n "${START},${END}p" $SRC
cat <<'C3'
int main(int argc, char **argv) {
    static cram_fd fd; static cram_metrics met[4];
    int level = atoi(argv[1]), version = atoi(argv[2]), n = atoi(argv[3]), set = atoi(argv[4]);
    fd.level = level; fd.version = version; pthread_mutex_init(&fd.metrics_lock, NULL);
    for (int i = 0; i < 4; i++) { memset(&met[i], 0, sizeof met[i]); met[i].trial = NTRIALS - 1; met[i].next_trial = TRIAL_SPAN / 2; met[i].method = RAW; }
    met[3].unpackable = 1;
    hts_set_log_level(HTS_LOG_OFF);
    for (long i = 0; i < n; i++) {
        cram_block b; memset(&b, 0, sizeof b);
        uint32_t len = 20000u + (uint32_t)((i * 7919) % 5000) + (uint32_t)(i % 4) * 30000u;
        if (i >= 400 && i < 520 && i % 4 == 0) len *= 20;             /* a sudden change of size on series 0 */
        b.method = RAW; b.uncomp_size = b.comp_size = (int32_t)len; b.data = calloc(len, 1); b.content_id = (int)(i % 4);
        cur_k = i;
        if (cram_compress_block3(&fd, NULL, &b, &met[i % 4], set, level, 0) != 0) { printf("ERR\n"); return 1; }
        printf("B %ld %d %d\n", i, (int)b.method, b.comp_size);
        free(b.data);
    }
    for (int q = 0; q < 4; q++) {
        cram_metrics *m = &met[q];
        printf("M %d %d %d %d %d %d %d %d %d %d", q, m->trial, m->next_trial, m->consistency, m->input_avg_sz, m->input_avg_delta, m->method,
               m->revised_method, m->strat, m->unpackable);
        for (int k = 0; k < CRAM_MAX_METHOD; k++) printf(" %d %d %.6f", m->sz[k], m->cnt[k], m->extra[k]);
        printf("\n");
    }
    return 0;
}
C3
} > $OUT/metrics_ref.c
[ -f $ROOT/oracle/_ref/config.h ] || make -C $ROOT/oracle _ref/config.h >/dev/null
gcc -O1 -w -I$ROOT/oracle/_ref -I$REF $OUT/metrics_ref.c $ROOT/oracle/ref_stubs.c -o $OUT/metrics_ref -lpthread
$OUT/metrics_ref $2 $3 $4 $5 > $OUT/ref_metrics.txt

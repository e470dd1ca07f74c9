#!/bin/bash
# TEST INFRASTRUCTURE.  Runs the REFERENCE's own cram_compress_slice (cram/cram_encode.c) over a grid of options with a
# recording cram_compress_block2, printing every call it makes: "<level> <version> <flags> : <ds> <method set> <level> <own metrics>".
# The function is `static`, so its text is spliced from the reference source at build time into a scratch file (never into
# the repository).   usage: gen_slice_policy_ref.sh <scratch dir>  -> writes <scratch dir>/ref_policy.txt
set -e
REF=${REF:-/root/reference}; OUT=$1; ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$REF/cram/cram_encode.c
START=$(grep -n '^static int cram_compress_slice(cram_fd \*fd, cram_container \*c, cram_slice \*s) {' $SRC | cut -d: -f1)
END=$(awk -v s=$START 'NR>s && /^}/ {print NR; exit}' $SRC)
[ -n "$START" ] && [ -n "$END" ] || { echo "cram_compress_slice not found in $SRC" >&2; exit 1; }
{
cat <<'C1'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "cram/cram.h"
static cram_slice *cur_s; static int naux_blocks;
int cram_compress_block2(cram_fd *fd, cram_slice *s, cram_block *b, cram_metrics *m, int method, int level) {
    int ds = -1;
    for (int i = 0; i < s->hdr->num_blocks; i++) if (s->block[i] == b) { ds = i; break; }
    printf(" %d:%d:%d:%d", ds, method, level, (m && ds >= 0 && ds < DS_END && m == fd->m[ds]) ? 1 : (m ? 2 : 0));
    return 0;
}
C1
sed -n "${START},${END}p" $SRC
cat <<'C2'
int main(void) {
    static cram_fd fd; static cram_container c; static cram_slice s; static cram_block_slice_hdr hdr;
    static cram_block blk[DS_END + 2]; static cram_block *bp[DS_END + 2]; static cram_metrics met[DS_END + 2];
    pthread_mutex_init(&fd.metrics_lock, NULL);
    s.hdr = &hdr; s.block = bp; hdr.num_blocks = DS_END + 2;
    for (int i = 0; i < DS_END + 2; i++) { bp[i] = &blk[i]; blk[i].method = RAW; blk[i].uncomp_size = 1000; blk[i].m = &met[i]; if (i < DS_END) fd.m[i] = &met[i]; }
    int versions[3] = {(2 << 8) | 1, (3 << 8) | 0, (3 << 8) | 1};
    for (int level = 0; level <= 9; level++) for (int vi = 0; vi < 3; vi++) for (int flags = 0; flags < 64; flags++) {
        fd.level = level; fd.version = versions[vi];
        fd.use_bz2 = flags & 1; fd.use_lzma = (flags >> 1) & 1; fd.use_rans = (flags >> 2) & 1; fd.use_arith = (flags >> 3) & 1;
        fd.use_fqz = (flags >> 4) & 1; fd.use_tok = (flags >> 5) & 1;
        printf("%d %d %d :", level, fd.version, flags);
        if (cram_compress_slice(&fd, &c, &s) != 0) return 1;
        printf("\n");
    }
    return 0;
}
C2
} > $OUT/slice_policy_ref.c
[ -f $ROOT/oracle/_ref/config.h ] || make -C $ROOT/oracle _ref/config.h >/dev/null
gcc -O1 -w -I$ROOT/oracle/_ref -I$REF $OUT/slice_policy_ref.c -o $OUT/slice_policy_ref -lpthread
$OUT/slice_policy_ref > $OUT/ref_policy.txt

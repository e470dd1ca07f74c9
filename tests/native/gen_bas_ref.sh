#!/bin/bash
# TEST INFRASTRUCTURE.  Builds a scratch program around the REFERENCE's own cram_byte_array_stop_decode_char (a static function of
# cram/cram_codecs.c, spliced from the reference source at build time into a scratch file -- never into the repository) and runs it
# item by item over a block, as cram_decode_slice does:
#   bas_ref <stop byte> < block    -> one item length per line, then "END <bytes consumed> <rc of the last call>"
# usage: gen_bas_ref.sh <scratch dir>   -> <scratch dir>/bas_ref
set -e
REF=${REF:-/root/reference}; OUT=$1; ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$REF/cram/cram_codecs.c
START=$(grep -n '^static int cram_byte_array_stop_decode_char(cram_slice \*slice, cram_codec \*c,' $SRC | cut -d: -f1)
END=$(awk -v s=$START 'NR>s && /^}/ {print NR; exit}' $SRC)
[ -n "$START" ] && [ -n "$END" ] || { echo "cram_byte_array_stop_decode_char not found" >&2; exit 1; }
mkdir -p $OUT
[ -f $ROOT/oracle/_ref/config.h ] || make -C $ROOT/oracle ref >/dev/null
{
cat <<'C1'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include "cram/cram.h"
C1
sed -n "${START},${END}p" $SRC
cat <<'C2'
int main(int argc, char **argv) {
    size_t cap = 1 << 20, len = 0, k; unsigned char *buf = malloc(cap);
    while ((k = fread(buf + len, 1, cap - len, stdin)) > 0) { len += k; if (len == cap) buf = realloc(buf, cap *= 2); }
    static cram_slice s; static cram_block b; static cram_codec c; static cram_block *by_id[1024];
    memset(&b, 0, sizeof b); b.data = buf; b.uncomp_size = (int32_t)len; b.content_type = EXTERNAL; b.content_id = 7;
    by_id[7] = &b; s.block_by_id = by_id;
    c.u.byte_array_stop.stop = (unsigned char)atoi(argv[1]); c.u.byte_array_stop.content_id = 7;
    int rc = 0;
    while ((size_t)b.idx < len) {
        int sz = 0x7fffffff;
        rc = cram_byte_array_stop_decode_char(&s, &c, NULL, NULL, &sz);
        if (rc) break;
        printf("%d\n", sz);
    }
    printf("END %ld %d\n", (long)b.idx, rc);
    return 0;
}
C2
} > $OUT/bas_ref.c
gcc -O1 -w -I$ROOT/oracle/_ref -I$REF -o $OUT/bas_ref $OUT/bas_ref.c

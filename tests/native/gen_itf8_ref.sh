#!/bin/bash
# TEST INFRASTRUCTURE.  Builds a scratch program around the REFERENCE's own safe_itf8_get and itf8_put (static functions of
# cram/cram_io.c, spliced from the reference source at build time into a scratch file -- never into the repository):
#   itf8_ref d < block      -> one decoded value per line, then "END <bytes consumed> <err>"   (the loop cram_decode_slice runs
#                              through cram_external_decode_int, one safe_itf8_get per value)
#   itf8_ref e < int32 LE   -> the bytes itf8_put writes for the values, on stdout
# usage: gen_itf8_ref.sh <scratch dir>   -> <scratch dir>/itf8_ref
set -e
REF=${REF:-/root/reference}; OUT=$1
SRC=$REF/cram/cram_io.c
cut_fn() {   # $1 = regex of the first line; prints the function up to its closing brace
    local s e
    s=$(grep -n "$1" $SRC | head -1 | cut -d: -f1)
    e=$(awk -v s=$s 'NR>s && /^}/ {print NR; exit}' $SRC)
    [ -n "$s" ] && [ -n "$e" ] || { echo "not found: $1" >&2; exit 1; }
    sed -n "${s},${e}p" $SRC
}
mkdir -p $OUT
{
cat <<'C1'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
C1
s=$(grep -n '^const int itf8_bytes\[16\]' $SRC | cut -d: -f1)
e=$(awk -v s=$s 'NR>=s && /};/ {print NR; exit}' $SRC)
sed -n "${s},${e}p" $SRC
cut_fn '^static inline int itf8_put(char \*cp, int32_t val)'
cut_fn '^static int64_t safe_itf8_get(char \*\*cp, const char \*endp, int \*err)'
cat <<'C2'
int main(int argc, char **argv) {
    size_t cap = 1 << 20, len = 0; char *buf = malloc(cap); size_t k;
    while ((k = fread(buf + len, 1, cap - len, stdin)) > 0) { len += k; if (len == cap) buf = realloc(buf, cap *= 2); }
    if (argc > 1 && argv[1][0] == 'd') {
        char *cp = buf, *endp = buf + len; int err = 0;
        while (cp < endp) {
            int64_t v = safe_itf8_get(&cp, endp, &err);
            if (err) break;
            printf("%d\n", (int32_t)v);
        }
        printf("END %ld %d\n", (long)(cp - buf), err);
    } else {
        size_t n = len / 4, i; char tmp[8];
        for (i = 0; i < n; i++) { int32_t v; memcpy(&v, buf + 4 * i, 4); int l = itf8_put(tmp, v); fwrite(tmp, 1, l, stdout); }
    }
    return 0;
}
C2
} > $OUT/itf8_ref.c
gcc -O1 -w -o $OUT/itf8_ref $OUT/itf8_ref.c

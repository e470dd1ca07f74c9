#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hts_bgzf_gpu.h"
int main(int argc, char **argv) {
    BGZF *fp = bgzf_open(argv[1], "r"); if (!fp) return 1;
    size_t cap = 8u << 20, tot = 0; char *buf = malloc(cap); unsigned long long sum = 0;
    for (;;) { ssize_t n = bgzf_read(fp, buf, cap); if (n < 0) return 2; if (n == 0) break; for (ssize_t i = 0; i < n; i += 4099) sum += (unsigned char)buf[i]; tot += n; fwrite(buf, 1, n, stdout); }
    bgzf_close(fp); fprintf(stderr, "read %zu\n", tot); return 0; }

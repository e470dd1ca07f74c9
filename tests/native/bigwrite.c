/* bgzf_write in calls of very different sizes after bgzf_mt(): whole blocks go straight from the caller's buffer into the batches (helpers copy the
 * megabyte spans), the rest through fp->uncompressed_block.  usage: bigwrite in out mode(0 = mixed sizes, 1 = small calls only) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hts_bgzf_gpu.h"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = malloc(n); if (fread(buf, 1, n, f) != n) return 1; fclose(f);
    BGZF *fp = bgzf_open(argv[2], "w"); if (!fp) return 2;
    if (bgzf_index_build_init(fp) != 0) return 3;
    bgzf_mt(fp, 4, 256);
    const int mode = atoi(argv[3]);
    static const size_t sizes[] = {8u << 20, 100, 70000, 0xff00, 3, 0xff00 * 3u, 1u << 20, 65279, 65281, 5u << 20};
    size_t pos = 0; int k = 0;
    while (pos < n) {
        size_t c = mode ? 1000 : sizes[k++ % 10];
        if (c > n - pos) c = n - pos;
        if (bgzf_write(fp, buf + pos, c) != (ssize_t)c) return 4;
        pos += c;
    }
    if (bgzf_index_dump(fp, argv[2], ".gzi") != 0) return 5;
    if (bgzf_close(fp) != 0) return 6;
    return 0; }

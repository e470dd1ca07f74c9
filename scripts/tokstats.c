/* tokstats.c -- token statistics of the deflate streams in a BGZF file (design aid, not product).
 * Includes the oracle source and instruments codes(). gcc -O2 -o tokstats scripts/tokstats.c */
#include <stdio.h>
#include <stdlib.h>
static long g_dist_hist[16], g_len_hist[9], g_nlit, g_nmatch, g_litrun_hist[12], g_cur_run, g_overlap, g_matchbytes;
#define TOKSTAT_MATCH(len, dist) do { g_nmatch++; g_matchbytes += (len); int b = 0; while ((1 << (b + 1)) <= (int)(dist)) b++; g_dist_hist[b]++; \
    int lb = (len) <= 4 ? 0 : (len) <= 8 ? 1 : (len) <= 16 ? 2 : (len) <= 32 ? 3 : (len) <= 64 ? 4 : (len) <= 128 ? 5 : 6; g_len_hist[lb]++; \
    if ((int)(dist) < (len)) g_overlap++; int r = 0; while ((1 << r) <= g_cur_run && r < 11) r++; g_litrun_hist[r]++; g_cur_run = 0; } while (0)
#define TOKSTAT_LIT() do { g_nlit++; g_cur_run++; } while (0)
#include "../oracle/bgzf_oracle.c"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *b = malloc(n); if (fread(b, 1, n, f) != n) return 1;
    uint8_t *o = malloc(1 << 30); long r = orc_bgzf_decompress_stream(b, n, o, 1 << 30);
    printf("decoded %ld bytes; literals %ld matches %ld (match bytes %ld, %.1f%% of output), overlapping(dist<len) %ld\n", r, g_nlit, g_nmatch, g_matchbytes, 100.0 * g_matchbytes / r, g_overlap);
    printf("dist histogram (log2 bucket: count):"); for (int i = 0; i < 16; i++) printf(" [%d]=%.1f%%", i, 100.0 * g_dist_hist[i] / g_nmatch); printf("\n");
    printf("len  histogram (<=4,8,16,32,64,128,258):"); for (int i = 0; i < 7; i++) printf(" %.1f%%", 100.0 * g_len_hist[i] / g_nmatch); printf("\n");
    printf("literal-run-before-match histogram (0,1,2-3,4-7,...):"); for (int i = 0; i < 12; i++) printf(" %.1f%%", 100.0 * g_litrun_hist[i] / g_nmatch); printf("\n");
    return 0; }

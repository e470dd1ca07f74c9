/* TEST INFRASTRUCTURE ONLY.
 * Link-time stubs that let the BGZF-only subset of the reference
 * (/root/reference: bgzf.c hfile.c thread_pool.c ...) link without the rest
 * of libhts.  Signatures follow htslib/hts.h:491,555,938, htslib/hts_log.h:62,75
 * and hts_internal.h:93.  None of these are on the codec path.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include "htslib/hts.h"
#include "htslib/hts_log.h"

int hts_verbose = 3;

void hts_log(enum htsLogLevel severity, const char *context, const char *format, ...)
{
    if ((int)severity > hts_verbose) return;
    va_list ap;
    va_start(ap, format);
    fprintf(stderr, "[ref:%s] ", context ? context : "?");
    vfprintf(stderr, format, ap);
    fputc('\n', stderr);
    va_end(ap);
}

/* the accessor pair of hts.c:5160-5168 (test/test_bgzf.c silences expected errors with them) */
void hts_set_log_level(enum htsLogLevel level) { hts_verbose = level; }
enum htsLogLevel hts_get_log_level(void) { return hts_verbose; }

int hts_idx_push(hts_idx_t *idx, int tid, hts_pos_t beg, hts_pos_t end, uint64_t offset, int is_mapped)
{ (void)idx; (void)tid; (void)beg; (void)end; (void)offset; (void)is_mapped; return -1; }

int hts_idx_check_range(hts_idx_t *idx, int tid, hts_pos_t beg, hts_pos_t end)
{ (void)idx; (void)tid; (void)beg; (void)end; return -1; }

int hts_detect_format(struct hFILE *fp, htsFormat *fmt)
{ (void)fp; memset(fmt, 0, sizeof(*fmt)); fmt->compression = bgzf; return 0; }

const char *hts_version(void) { return "1.23.1-oracle"; }

size_t hts_realloc_or_die(size_t n, size_t m, size_t m_sz, size_t size,
                          int clear, void **ptr, const char *func)
{
    size_t new_m = n; (void)m_sz;
    /* round up to a power of two like a growable array would */
    size_t p = 1; while (p < new_m) p <<= 1; new_m = p;
    void *np = realloc(*ptr, new_m * size);
    if (!np) { fprintf(stderr, "[ref:%s] out of memory\n", func); exit(1); }
    if (clear && new_m > m) memset((char *)np + m * size, 0, (new_m - m) * size);
    *ptr = np;
    return new_m;
}

/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_FQZCOMP_QUAL_H
#define ORC_STUB_FQZCOMP_QUAL_H
#include <stdint.h>
#include <stddef.h>
/* cram/cram_io.c:1808-1820 fills exactly these three fields */
typedef struct { int num_records; uint32_t *len; uint32_t *flags; } fqz_slice;
typedef struct fqz_gparams fqz_gparams;
#define FQZ_FREVERSE 16
#define FQZ_FREAD2 128
/* cram/cram_io.c:1821 fqz_compress(vers, f, in, in_size, out_size, strat >> 8, NULL); :1686 fqz_decompress(data, comp_size, &uncomp_size, NULL, 0) */
char *fqz_compress(int vers, fqz_slice *s, char *in, size_t uncomp_size, size_t *comp_size, int strat, fqz_gparams *gp);
char *fqz_decompress(char *in, size_t comp_size, size_t *uncomp_size, int *lengths, int nlengths);
#endif

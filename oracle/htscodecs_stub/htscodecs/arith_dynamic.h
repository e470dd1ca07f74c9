/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_ARITH_DYNAMIC_H
#define ORC_STUB_ARITH_DYNAMIC_H
/* cram/cram_io.c:1718 arith_uncompress_to(b->data, b->comp_size, NULL, &usize2); :1879 arith_compress_to(in, in_size, NULL, &out_size_i, flags) */
unsigned char *arith_compress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order);
unsigned char *arith_uncompress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size);
#endif

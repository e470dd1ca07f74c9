/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_RANS_STATIC_H
#define ORC_STUB_RANS_STATIC_H
/* cram/cram_io.c:1668 rans_uncompress(b->data, b->comp_size, &usize2); :1838 rans_compress(in, in_size, &out_size_i, order) */
unsigned char *rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order);
unsigned char *rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size);
#endif

/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_RANS_STATIC4X16_H
#define ORC_STUB_RANS_STATIC4X16_H
/* flag bits: cram/cram_external.c:616-637 reads them out of the first byte of a stream; values from the CRAM codecs
 * specification (hts-specs CRAMcodecs, "rANS Nx16" first byte) */
#define RANS_ORDER_X32    0x04
#define RANS_ORDER_STRIPE 0x08
#define RANS_ORDER_NOSZ   0x10
#define RANS_ORDER_CAT    0x20
#define RANS_ORDER_RLE    0x40
#define RANS_ORDER_PACK   0x80
/* cram/cram_io.c:1860 ORs this into the order: "pick 4-way or 32-way yourself"; not a stream bit */
#define RANS_ORDER_SIMD_AUTO 0x100
/* cram/cram_io.c:1699, :1859 */
unsigned char *rans_compress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order);
unsigned char *rans_uncompress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size);
#endif

/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_RLE_H
#define ORC_STUB_RLE_H
#include <stdint.h>
/* cram/cram_codecs.c:2106 hts_rle_decode(lit, lit_sz, len, len_sz, rle_syms, rle_nsyms, out, &out_sz); :2278 hts_rle_encode(data, len, run, &run_len, rle_syms, &rle_nsyms, NULL, &out_len) */
uint8_t *hts_rle_encode(uint8_t *data, uint64_t data_len, uint8_t *run, uint64_t *run_len, uint8_t *rle_syms, int *rle_nsyms,
                        uint8_t *out, uint64_t *out_len);
uint8_t *hts_rle_decode(uint8_t *lit, uint64_t lit_len, uint8_t *run, uint64_t run_len, uint8_t *rle_syms, int rle_nsyms,
                        uint8_t *out, uint64_t *out_len);
#endif

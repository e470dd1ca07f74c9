/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_TOKENISE_NAME3_H
#define ORC_STUB_TOKENISE_NAME3_H
#include <stdint.h>
/* cram/cram_io.c:1737 tok3_decode_names(b->data, b->comp_size, &out_len); :1891 tok3_encode_names(in, in_size, lev, strat, &out_len, NULL) */
uint8_t *tok3_encode_names(char *blk, int len, int level, int use_arith, int *out_len, int *last_start_p);
uint8_t *tok3_decode_names(uint8_t *in, uint32_t sz, uint32_t *out_len);
#endif

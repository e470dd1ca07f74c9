/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_HTSCODECS_H
#define ORC_STUB_HTSCODECS_H
/* hts.c:149,225 */
const char *htscodecs_version(void);
#endif

/* TEST INFRASTRUCTURE ONLY -- stand-in for a header of samtools/htscodecs v1.6.6 (an un-vendored submodule of the
 * reference: /root/reference/htscodecs is empty, .gitmodules).  NOT htscodecs code: the prototypes are inferred from the
 * reference's call sites (file:line below) so that the reference's own cram/ *.c, sam.c, hts.c compile from where they lie
 * into oracle/_ref/ (oracle/Makefile target ref_cram).  Bodies: oracle/htscodecs_stub/htscodecs_stub.c. */
#ifndef ORC_STUB_VARINT_H
#define ORC_STUB_VARINT_H
#include <stdint.h>
/* CRAM 4.0 integers (cram/cram_io.c:768-985 wraps these; "Big endian, see also htscodecs/varint.h" :892): 7 bits per byte, most
 * significant group first, bit 7 = another byte follows; signed values zig-zag folded.  put: bytes written, 0 when they do not fit
 * (endp may be NULL = no limit).  get: bytes consumed, 0 when the buffer ends first (endp may be NULL).  Not used by CRAM 2.x / 3.x. */
static inline int var_size_u64(uint64_t v) { int n = 1; while (v >>= 7) n++; return n; }
static inline int var_put_u64(uint8_t *cp, const uint8_t *endp, uint64_t v) {
    int n = var_size_u64(v), s;
    if (endp && endp - cp < n) return 0;
    for (s = 7 * (n - 1); s > 0; s -= 7) *cp++ = (uint8_t)(((v >> s) & 0x7f) | 0x80);
    *cp = (uint8_t)(v & 0x7f);
    return n;
}
static inline int var_get_u64(uint8_t *cp, const uint8_t *endp, uint64_t *v) {
    uint64_t x = 0; int n = 0; uint8_t c;
    do {
        if ((endp && cp >= endp) || n >= 11) { *v = x; return 0; }
        c = *cp++; x = (x << 7) | (c & 0x7f); n++;
    } while (c & 0x80);
    *v = x;
    return n;
}
static inline int var_put_u32(uint8_t *cp, const uint8_t *endp, uint32_t v) { return var_put_u64(cp, endp, v); }
static inline int var_get_u32(uint8_t *cp, const uint8_t *endp, uint32_t *v) {
    uint64_t x; int n = var_get_u64(cp, endp, &x); *v = (uint32_t)x; return n;
}
static inline int var_put_s64(uint8_t *cp, const uint8_t *endp, int64_t v) {
    return var_put_u64(cp, endp, ((uint64_t)v << 1) ^ (uint64_t)(v >> 63));
}
static inline int var_get_s64(uint8_t *cp, const uint8_t *endp, int64_t *v) {
    uint64_t x; int n = var_get_u64(cp, endp, &x); *v = (int64_t)(x >> 1) ^ -(int64_t)(x & 1); return n;
}
static inline int var_put_s32(uint8_t *cp, const uint8_t *endp, int32_t v) {
    return var_put_u64(cp, endp, (uint32_t)(((uint32_t)v << 1) ^ (uint32_t)(v >> 31)));
}
static inline int var_get_s32(uint8_t *cp, const uint8_t *endp, int32_t *v) {
    uint64_t x; int n = var_get_u64(cp, endp, &x); *v = (int32_t)((uint32_t)x >> 1) ^ -(int32_t)(x & 1); return n;
}
#endif

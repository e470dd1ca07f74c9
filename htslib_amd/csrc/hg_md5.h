// hg_md5.h -- MD5 (RFC 1321), host code.  Used for the reference-span digest of a CRAM slice header: checked on reading
// (cram_decode_slice, reference cram/cram_decode.c:2480-2540) and stored on writing (cram_encode_slice, cram/cram_encode.c:1700-1760).
#pragma once
#include <stdint.h>
#include <string.h>

namespace hgr {

// MD5 (RFC 1321) of a reference span: cram_decode_slice compares it with the slice header's (cram_decode.c:2480-2540).  Written from the RFC.
inline void md5_of(const uint8_t *p, uint64_t n, uint8_t out[16]) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
        0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
        0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
        0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    auto block = [&](const uint8_t *b) {
        uint32_t w[16];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] | (uint32_t)b[4 * i + 1] << 8 | (uint32_t)b[4 * i + 2] << 16 | (uint32_t)b[4 * i + 3] << 24;
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3];
        for (int i = 0; i < 64; i++) {
            uint32_t f; int g;
            if (i < 16) { f = (bb & c) | (~bb & d); g = i; }
            else if (i < 32) { f = (d & bb) | (~d & c); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = bb ^ c ^ d; g = (3 * i + 5) & 15; }
            else { f = c ^ (bb | ~d); g = (7 * i) & 15; }
            const uint32_t t = a + f + K[i] + w[g];
            a = d; d = c; c = bb; bb += (t << S[i]) | (t >> (32 - S[i]));
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d;
    };
    uint64_t i = 0;
    for (; i + 64 <= n; i += 64) block(p + i);
    uint8_t tail[128]; const size_t r = (size_t)(n - i);
    memset(tail, 0, sizeof tail);
    if (r) memcpy(tail, p + i, r);
    tail[r] = 0x80;
    const size_t tl = r < 56 ? 64 : 128;
    const uint64_t bits = n * 8;
    for (int k = 0; k < 8; k++) tail[tl - 8 + k] = (uint8_t)(bits >> (8 * k));
    block(tail); if (tl == 128) block(tail + 64);
    for (int k = 0; k < 16; k++) out[k] = (uint8_t)(h[k >> 2] >> (8 * (k & 3)));
}

}  // namespace hgr

// Test harness: the front-end's host block codec (htslib_amd/csrc/bgzf_host_codec.h) behind a plain C ABI, so that tests/test_host_codec.py can hold it against
// the oracle, python's zlib and the real reference without a GPU.  Test infrastructure.
#include "bgzf_host_codec.h"
extern "C" {
int hc_block_inflate(const uint8_t *block, size_t clen, uint8_t *out, uint32_t ulen) { static thread_local hgh::Inflater I; return hgh::bgzf_block_inflate(I, block, clen, out, ulen); }
int hc_block_deflate(uint8_t *dst, size_t *dlen, const uint8_t *src, size_t slen, int level) { static thread_local hgh::Deflater D; return hgh::bgzf_block_deflate(D, dst, dlen, src, slen, level); }
uint32_t hc_crc32(uint32_t crc, const uint8_t *p, size_t n) { return hgh::crc32(crc, p, n); }
}

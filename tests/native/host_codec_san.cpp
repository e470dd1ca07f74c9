// TEST INFRASTRUCTURE (tests/test_host_codec.py): the front-end's host block codec under AddressSanitizer + UBSan -- every block of the input files is cut at
// random sizes, deflated at levels 0 / 1 / 6 / 9 into exact-size heap buffers, inflated back, and then damaged six ways (bit flips anywhere behind the BGZF
// header, a truncation): the inflater must stay inside its buffers whatever its verdict, must never accept a truncated block, and may accept a flipped bit only
// when the bytes it returns are still the right ones (padding bits behind the end-of-block code or before a stored block's length).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "bgzf_host_codec.h"
static uint64_t x = 88172645463325252ull;
static uint64_t rnd() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
int main(int argc, char **argv) {
    static hgh::Deflater D; static hgh::Inflater I;
    size_t blocks = 0, damaged_ok = 0, damaged_bad = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb"); if (!f) continue;
        std::vector<uint8_t> d(8 << 20); size_t n = fread(d.data(), 1, d.size(), f); fclose(f); d.resize(n);
        for (int level : {0, 1, 6, 9}) {
            for (size_t o = 0; o < n; ) {
                size_t len = 1 + rnd() % 0xff00; if (len > n - o) len = n - o;
                std::vector<uint8_t> blk(len + 1024);   // exact-size heap buffers: the sanitizer sees every overrun
                size_t dl = blk.size();
                std::vector<uint8_t> src(d.begin() + o, d.begin() + o + len);
                if (hgh::bgzf_block_deflate(D, blk.data(), &dl, src.data(), len, level) != 0) { printf("deflate failed\n"); return 1; }
                std::vector<uint8_t> comp(blk.begin(), blk.begin() + dl), out(len);
                if (hgh::bgzf_block_inflate(I, comp.data(), comp.size(), out.data(), (uint32_t)len) != 0 || memcmp(out.data(), src.data(), len)) { printf("round trip failed at %zu level %d\n", o, level); return 1; }
                blocks++;
                for (int k = 0; k < 6; k++) {               // damage: the inflater must stay inside its buffers whatever the verdict
                    std::vector<uint8_t> bad(comp);
                    const size_t at = 18 + rnd() % (bad.size() - 18);
                    bad[at] ^= (uint8_t)(1u << (rnd() & 7));
                    if (k == 5) bad.resize(18 + rnd() % (bad.size() - 18));
                    std::vector<uint8_t> o2(len);
                    const int rc = hgh::bgzf_block_inflate(I, bad.data(), bad.size(), o2.data(), (uint32_t)len);
                    if (rc == 0 && (k == 5 || memcmp(o2.data(), src.data(), len))) { printf("ACCEPTED WRONG: level %d len %zu flip at %zu of %zu k %d\n", level, len, at, comp.size(), k); return 3; }
                    (rc == 0 ? damaged_ok : damaged_bad)++;
                }
                o += len;
            }
        }
    }
    printf("%zu blocks round-tripped, damaged: %zu rejected, %zu accepted (CRC-equal flips cannot happen: must be 0)\n", blocks, damaged_bad, damaged_ok);
    return 0;
}

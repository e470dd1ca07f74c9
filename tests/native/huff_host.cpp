// Host build of htslib_amd/csrc/deflate_huff.h for CPU unit tests (tests/test_deflate_huff.py).
#include <stdint.h>
#include <string.h>
#include <vector>
#include "deflate_huff.h"
using namespace hgdef;
extern "C" {
void hh_build_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *len) {
    std::vector<uint16_t> order(n + 2); std::vector<uint32_t> work(n + 2);
    build_lengths(freq, n, maxbits, len, order.data(), work.data());
}
void hh_assign_codes(const uint8_t *len, int n, uint16_t *code) { assign_codes(len, n, code); }
void hh_len_symbol(uint32_t len, uint32_t *s, uint32_t *xb, uint32_t *xv) { len_symbol(len, *s, *xb, *xv); }
void hh_dist_symbol(uint32_t d, uint32_t *s, uint32_t *xb, uint32_t *xv) { dist_symbol(d, *s, *xb, *xv); }
// A complete raw-deflate stream of literals + optional (len,dist) tokens given as u32
// (0x80000000 | (len-3)<<16 | (dist-1)) -- same token format as the kernel.  Returns bytes.
long hh_encode_tokens(const uint32_t *tok, long ntok, uint8_t *out, long cap) {
    uint32_t lf[288] = {0}, df[32] = {0};
    for (long i = 0; i < ntok; i++) {
        uint32_t t = tok[i];
        if (t & 0x80000000u) { uint32_t s, xb, xv; len_symbol(((t >> 16) & 0xff) + 3, s, xb, xv); lf[257 + s]++;
                               dist_symbol((t & 0x7fff) + 1, s, xb, xv); df[s]++; }
        else lf[t & 0xff]++;
    }
    lf[256] = 1;
    uint8_t ll[288], dl[32]; uint16_t lc[288], dc[32]; uint16_t order[320]; uint32_t work[320];
    build_lengths(lf, 286, 15, ll, order, work); build_lengths(df, 30, 15, dl, order, work);
    assign_codes(ll, 286, lc); assign_codes(dl, 30, dc);
    uint8_t cs[320], ce[320];
    memset(out, 0, cap);
    uint32_t nb = write_dynamic_header(ll, dl, out, cs, ce, work, order);
    BitSink bs{out, nb};
    bs.resume();
    for (long i = 0; i < ntok; i++) {
        uint32_t t = tok[i];
        if ((long)(bs.nbits >> 3) + 16 > cap) return -1;
        if (t & 0x80000000u) {
            uint32_t s, xb, xv; len_symbol(((t >> 16) & 0xff) + 3, s, xb, xv);
            bs.put(lc[257 + s], ll[257 + s]); bs.put(xv, xb);
            dist_symbol((t & 0x7fff) + 1, s, xb, xv); bs.put(dc[s], dl[s]); bs.put(xv, xb);
        } else bs.put(lc[t & 0xff], ll[t & 0xff]);
    }
    bs.put(lc[256], ll[256]);
    bs.finish();
    return (bs.nbits + 7) >> 3;
}
}

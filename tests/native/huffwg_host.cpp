// Host build of htslib_amd/csrc/deflate_huff_wg.h (the workgroup-collective Huffman phase of the deflate kernel) for CPU
// unit tests (tests/test_deflate_huff.py): the "workgroup" is NT real threads and a pthread barrier.
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>
#include "deflate_huff_wg.h"
using namespace hgdef;

static pthread_barrier_t g_bar;
extern "C" void hgw_host_barrier(void) { pthread_barrier_wait(&g_bar); }

template <class F> static void run_wg(int nt, F f) {
    pthread_barrier_init(&g_bar, nullptr, (unsigned)nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t] { f(t); });
    for (auto &x : th) x.join();
    pthread_barrier_destroy(&g_bar);
}

struct Shared {
    HuffWG W;
    alignas(4) uint32_t lf[288]; alignas(4) uint32_t df[32];
    alignas(4) uint8_t ll[288]; alignas(4) uint8_t dl[32];
    uint16_t lc[288], dc[32];
};

template <int NT> static void whole(Shared &S, int bfinal) {
    run_wg(NT, [&](int tid) {
        wg_code_lengths<NT, 15>(S.W, S.lf, 286, S.ll, S.df, 30, S.dl, tid);
        wg_assign_codes<NT>(S.W, S.ll, 286, S.lc, S.dl, 30, S.dc, tid);
        wg_dynamic_header<NT>(S.W, S.ll, S.dl, bfinal, tid);
    });
}

extern "C" {
// code lengths of one alphabet (maxbits 15 or 7) on nt = 32 or 256 threads
void hhw_build_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *len, int nt) {
    static Shared S;
    memset(&S, 0xa5, sizeof S);
    for (int i = 0; i < 288; i++) S.lf[i] = i < n ? freq[i] : 0;
    auto body = [&](auto NTc, int tid) {
        constexpr int NT = decltype(NTc)::value;
        if (maxbits == 15) wg_code_lengths<NT, 15>(S.W, S.lf, n, S.ll, nullptr, 0, nullptr, tid);
        else wg_code_lengths<NT, 7>(S.W, S.lf, n, S.ll, nullptr, 0, nullptr, tid);
    };
    if (nt == 256) run_wg(256, [&](int tid) { body(std::integral_constant<int, 256>{}, tid); });
    else run_wg(32, [&](int tid) { body(std::integral_constant<int, 32>{}, tid); });
    memcpy(len, S.ll, (size_t)n);
}
// A complete raw-deflate stream for kernel-format tokens (see huff_host.cpp: hh_encode_tokens), the Huffman phase run collectively.
long hhw_encode_tokens(const uint32_t *tok, long ntok, uint8_t *out, long cap, int nt) {
    static Shared S;
    memset(&S, 0x5a, sizeof S);
    memset(S.lf, 0, sizeof S.lf); memset(S.df, 0, sizeof S.df);
    for (long i = 0; i < ntok; i++) {
        uint32_t t = tok[i];
        if (t & 0x80000000u) { uint32_t s, xb, xv; len_symbol(((t >> 16) & 0xff) + 3, s, xb, xv); S.lf[257 + s]++;
                               dist_symbol((t & 0x7fff) + 1, s, xb, xv); S.df[s]++; }
        else S.lf[t & 0xff]++;
    }
    S.lf[256] = 1;
    if (nt == 256) whole<256>(S, 1); else whole<32>(S, 1);
    memset(out, 0, cap);
    BitSink bs{out, 0};
    for (uint32_t k = 0; k < S.W.nitems; k++) bs.put(S.W.item_v[k], S.W.item_n[k]);
    if (bs.nbits != S.W.hdr_bits) return -2;
    for (long i = 0; i < ntok; i++) {
        uint32_t t = tok[i];
        if ((long)(bs.nbits >> 3) + 16 > cap) return -1;
        if (t & 0x80000000u) {
            uint32_t s, xb, xv; len_symbol(((t >> 16) & 0xff) + 3, s, xb, xv);
            bs.put(S.lc[257 + s], S.ll[257 + s]); bs.put(xv, xb);
            dist_symbol((t & 0x7fff) + 1, s, xb, xv); bs.put(S.dc[s], S.dl[s]); bs.put(xv, xb);
        } else bs.put(S.lc[t & 0xff], S.ll[t & 0xff]);
    }
    bs.put(S.lc[256], S.ll[256]);
    bs.finish();
    return (bs.nbits + 7) >> 3;
}
}

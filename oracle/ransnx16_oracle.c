/*
 * ransnx16_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * *** PARITY UNPINNED ***  Plain-C restatement of the CRAM 3.1 "rANS Nx16" block codec (CRAM
 * block method 5), the codec behind
 *      rans_uncompress_4x16(in, in_size, &out_size)          cram/cram_io.c:1699
 *      rans_compress_4x16(in, in_size, &out_size, flags)     cram/cram_io.c:1859
 * whose implementation (htscodecs v1.6.6: rANS_static4x16pr.c, rANS_static32x16pr*.c, pack.c,
 * rle.c, varint.h) is an ABSENT git submodule of the reference, and for which the reference holds
 * NO golden stream (its tests for CRAM 3.1 are self round trips only, test/test.pl:792-816).
 * This file follows the published specification (hts-specs "CRAM codecs" v3.1, rANS Nx16 chapter)
 * as restated in SURVEY.md Appendix A.4; flag bits are confirmed by the reference
 * (cram/cram_external.c:616-637, cram/cram_io.c:1856).  Until a stream written by stock htslib is
 * available, agreement with htscodecs on the byte level is UNVERIFIED; what the tests establish is
 * encoder/decoder self-consistency of this restatement and bit-exactness of the GPU decoder
 * against it.
 *
 * Stream:  flags:u8  [ulen:uint7 unless NOSZ]  then by flag
 *   STRIPE 0x08: N:u8, N x clen:uint7, N complete sub-streams (byte j of sub-stream k = in[j*N+k])
 *   CAT    0x20: ulen raw bytes
 *   PACK   0x80: nsym:u8, nsym symbols, packed_len:uint7          (then RLE / entropy on packed data)
 *   RLE    0x40: (2*meta_len + raw?1:0):uint7, lit_len:uint7, [cmeta_len:uint7], meta (raw or rANS o0)
 *   entropy core: order 0/1, N = 4 or 32 (X32 0x04) interleaved 32-bit states, 16-bit renormalisation,
 *                 lower bound 2^15, 12-bit (order-1: 10 or 12-bit) frequencies.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))
#define F_ORDER 0x01
#define F_X32 0x04
#define F_STRIPE 0x08
#define F_NOSZ 0x10
#define F_CAT 0x20
#define F_RLE 0x40
#define F_PACK 0x80
#define TF_SHIFT 12
#define RANS_L (1u << 15)

/* ---- uint7: big-endian base-128, MSB = continue --------------------------------------------- */
static int put_u7(uint8_t *cp, uint32_t v)
{
    int n = 0;
    uint8_t tmp[5];
    do { tmp[n++] = v & 0x7f; v >>= 7; } while (v);
    for (int i = n - 1; i >= 0; i--) *cp++ = tmp[i] | (i ? 0x80 : 0);
    return n;
}
static int get_u7(const uint8_t *cp, const uint8_t *end, uint32_t *v)
{
    uint32_t x = 0; int n = 0; uint8_t c;
    do {
        if (cp + n >= end || n >= 5) return -1;
        c = cp[n++];
        x = (x << 7) | (c & 0x7f);
    } while (c & 0x80);
    *v = x;
    return n;
}

/* ---- alphabet: symbol run-length list (same scheme as the 4x8 tables) ------------------------ */
static uint8_t *put_alphabet(uint8_t *cp, const uint32_t *F)
{
    int rle = 0;
    for (int j = 0; j < 256; j++) {
        if (!F[j]) continue;
        if (rle) { rle--; continue; }
        *cp++ = (uint8_t)j;
        if (j && F[j - 1]) {
            for (rle = j + 1; rle < 256 && F[rle]; rle++) ;
            rle -= j + 1;
            *cp++ = (uint8_t)rle;
        }
    }
    *cp++ = 0;
    return cp;
}
static const uint8_t *get_alphabet(const uint8_t *cp, const uint8_t *end, uint8_t *present)
{
    memset(present, 0, 256);
    if (cp >= end) return NULL;
    unsigned rle = 0, j = *cp++;
    for (int guard = 0; guard < 257; guard++) {
        present[j] = 1;
        if (cp >= end) return NULL;
        if (!rle && j + 1 == *cp) {
            j = *cp++;
            if (cp >= end) return NULL;
            rle = *cp++;
        } else if (rle) {
            rle--; j++;
            if (j > 255) return NULL;
        } else {
            j = *cp++;
        }
        if (j == 0) return cp;
    }
    return NULL;
}

/* scale counts to sum exactly `tot` (a power of two), every present symbol >= 1 */
static void normalise(uint32_t *F, uint32_t size, uint32_t tot)
{
    uint64_t sum = 0; int M = -1; uint32_t m = 0;
    if (!size) return;
    for (int j = 0; j < 256; j++) {
        if (!F[j]) continue;
        if (F[j] > m) { m = F[j]; M = j; }
        uint64_t f = ((uint64_t)F[j] * tot) / size;
        if (!f) f = 1;
        F[j] = (uint32_t)f; sum += f;
    }
    if (sum < tot) F[M] += (uint32_t)(tot - sum);
    while (sum > tot) {                                   /* shave the largest entries */
        int b = -1;
        for (int j = 0; j < 256; j++) if (F[j] > 1 && (b < 0 || F[j] > F[b])) b = j;
        uint32_t take = (uint32_t)(sum - tot);
        if (take > F[b] - 1) take = F[b] - 1;
        F[b] -= take; sum -= take;
    }
}
static uint32_t round2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }
static void shift_up(uint32_t *F, uint32_t tot, uint32_t target)
{
    if (!tot || tot == target) return;
    int sh = 0;
    while (tot < target) { tot <<= 1; sh++; }
    for (int j = 0; j < 256; j++) F[j] <<= sh;
}

/* ---- encoder primitives ------------------------------------------------------------------ */
typedef struct { uint32_t start, freq; } sym_t;
static void enc_put(uint32_t *r, uint8_t **pp, const sym_t *s, int shift)
{
    uint32_t x = *r, x_max = ((RANS_L >> shift) << 16) * s->freq;
    if (x >= x_max) { *pp -= 2; (*pp)[0] = (uint8_t)x; (*pp)[1] = (uint8_t)(x >> 8); x >>= 16; }
    *r = ((x / s->freq) << shift) + (x % s->freq) + s->start;
}
static void enc_flush(uint32_t r, uint8_t **pp)
{
    *pp -= 4;
    (*pp)[0] = (uint8_t)r; (*pp)[1] = (uint8_t)(r >> 8); (*pp)[2] = (uint8_t)(r >> 16); (*pp)[3] = (uint8_t)(r >> 24);
}

/* entropy core, order 0.  Returns bytes written to out. */
static size_t enc_o0(const uint8_t *in, size_t n, uint8_t *out, int N)
{
    uint32_t F[256] = {0};
    sym_t syms[256];
    for (size_t i = 0; i < n; i++) F[in[i]]++;
    uint32_t tot = round2((uint32_t)n);
    if (tot > (1u << TF_SHIFT)) tot = 1u << TF_SHIFT;
    normalise(F, (uint32_t)n, tot);
    uint8_t *cp = put_alphabet(out, F);
    for (int j = 0; j < 256; j++) if (F[j]) cp += put_u7(cp, F[j]);
    shift_up(F, tot, 1u << TF_SHIFT);
    uint32_t x = 0;
    for (int j = 0; j < 256; j++) { syms[j].start = x; syms[j].freq = F[j]; x += F[j]; }
    uint8_t *buf = malloc(n * 2 + 256), *ptr = buf + n * 2 + 256, *bend = ptr;
    uint32_t R[32];
    for (int z = 0; z < N; z++) R[z] = RANS_L;
    size_t rem = n & (size_t)(N - 1);
    for (size_t z = rem; z-- > 0;) enc_put(&R[z], &ptr, &syms[in[n - (rem - z)]], TF_SHIFT);
    for (size_t i = n & ~(size_t)(N - 1); i > 0; i -= N)
        for (int z = N - 1; z >= 0; z--) enc_put(&R[z], &ptr, &syms[in[i - (N - z)]], TF_SHIFT);
    for (int z = N - 1; z >= 0; z--) enc_flush(R[z], &ptr);
    size_t body = (size_t)(bend - ptr);
    memcpy(cp, ptr, body);
    free(buf);
    return (size_t)(cp - out) + body;
}

static int dec_o0(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_sz, int N)
{
    const uint8_t *cp = in, *end = in + in_size;
    uint8_t present[256];
    uint32_t F[256] = {0}, C[257];
    static __thread uint8_t lookup[1 << TF_SHIFT];
    if (!(cp = get_alphabet(cp, end, present))) return -1;
    uint32_t tot = 0;
    for (int j = 0; j < 256; j++) {
        if (!present[j]) continue;
        int k = get_u7(cp, end, &F[j]);
        if (k < 0) return -1;
        cp += k; tot += F[j];
        if (tot > (1u << TF_SHIFT)) return -1;
    }
    if (!tot || (tot & (tot - 1))) return -1;             /* must be a power of two */
    shift_up(F, tot, 1u << TF_SHIFT);
    uint32_t x = 0;
    for (int j = 0; j < 256; j++) { C[j] = x; if (F[j]) memset(lookup + x, j, F[j]); x += F[j]; }
    C[256] = x;
    if (cp + 4 * N > end) return -1;
    uint32_t R[32];
    for (int z = 0; z < N; z++, cp += 4) R[z] = cp[0] | (cp[1] << 8) | (cp[2] << 16) | ((uint32_t)cp[3] << 24);
    size_t out_end = out_sz & ~(size_t)(N - 1);
    const uint32_t mask = (1u << TF_SHIFT) - 1;
    for (size_t i = 0; i < out_end; i += N)
        for (int z = 0; z < N; z++) {
            uint32_t m = R[z] & mask; uint8_t c = lookup[m];
            out[i + z] = c;
            R[z] = F[c] * (R[z] >> TF_SHIFT) + m - C[c];
            if (R[z] < RANS_L) { if (cp + 2 > end) return -1; R[z] = (R[z] << 16) | cp[0] | (cp[1] << 8); cp += 2; }
        }
    for (size_t z = out_sz & (size_t)(N - 1); z-- > 0;) out[out_end + z] = lookup[R[z] & mask];
    return 0;
}

/* order-1 table: alphabet, then for every context of the alphabet the frequencies of every symbol of
 * the alphabet as uint7, a zero frequency being followed by one byte = number of further zeros */
static size_t enc_o1(const uint8_t *in, size_t n, uint8_t *out, int N)
{
    uint32_t (*F)[256] = calloc(256, sizeof *F);
    uint32_t T[256] = {0}, A[256] = {0};
    sym_t (*syms)[256] = calloc(256, sizeof *syms);
    size_t isz = n / N;
    unsigned last = 0;
    for (size_t i = 0; i < n; i++) { F[last][in[i]]++; T[last]++; A[last] = 1; A[in[i]] = 1; last = in[i]; }
    for (int z = 1; z < N; z++) { F[0][in[z * isz]]++; T[0]++; }   /* every state starts in context 0 */
    /* the first symbol of states 1..N-1 was also counted under its true predecessor: harmless */
    const int shift = TF_SHIFT;
    uint8_t *cp = out;
    *cp++ = (uint8_t)(shift << 4);                            /* table stored uncompressed */
    cp = put_alphabet(cp, A);
    for (int i = 0; i < 256; i++) {
        if (!A[i]) continue;
        if (T[i]) {
            uint32_t tot = round2(T[i]);
            if (tot > (1u << shift)) tot = 1u << shift;
            normalise(F[i], T[i], tot);
            int run = 0;
            for (int j = 0; j < 256; j++) {
                if (!A[j]) continue;
                if (run) { run--; continue; }
                cp += put_u7(cp, F[i][j]);
                if (!F[i][j]) {
                    for (int k = j + 1; k < 256; k++) { if (!A[k]) continue; if (F[i][k] == 0) run++; else break; }
                    *cp++ = (uint8_t)run;
                }
            }
            shift_up(F[i], tot, 1u << shift);
        } else {                                              /* context never used: all zeros */
            int cnt = 0;
            for (int j = 0; j < 256; j++) cnt += A[j] != 0;
            cp += put_u7(cp, 0); *cp++ = (uint8_t)(cnt - 1);
        }
        uint32_t x = 0;
        for (int j = 0; j < 256; j++) { syms[i][j].start = x; syms[i][j].freq = F[i][j]; x += F[i][j]; }
    }
    uint8_t *buf = malloc(n * 2 + 256), *ptr = buf + n * 2 + 256, *bend = ptr;
    uint32_t R[32]; long idx[32]; unsigned l[32];
    for (int z = 0; z < N; z++) { R[z] = RANS_L; idx[z] = (long)((z + 1) * isz) - 2; l[z] = in[idx[z] + 1]; }
    l[N - 1] = in[n - 1];
    for (idx[N - 1] = (long)n - 2; idx[N - 1] > (long)(N * isz) - 2; idx[N - 1]--) {
        unsigned c = in[idx[N - 1]];
        enc_put(&R[N - 1], &ptr, &syms[c][l[N - 1]], shift);
        l[N - 1] = c;
    }
    for (; idx[0] >= 0;) {
        for (int z = N - 1; z >= 0; z--) {
            unsigned c = in[idx[z]];
            enc_put(&R[z], &ptr, &syms[c][l[z]], shift);
            l[z] = c; idx[z]--;
        }
    }
    for (int z = N - 1; z >= 0; z--) enc_put(&R[z], &ptr, &syms[0][l[z]], shift);
    for (int z = N - 1; z >= 0; z--) enc_flush(R[z], &ptr);
    size_t body = (size_t)(bend - ptr);
    memcpy(cp, ptr, body);
    size_t tot = (size_t)(cp - out) + body;
    free(buf); free(F); free(syms);
    return tot;
}

static int dec_o0_alloc(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_sz, int N);

static int dec_o1(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_sz, int N)
{
    const uint8_t *cp = in, *end = in + in_size;
    uint8_t *tab_free = NULL;
    int rc = -1;
    uint32_t (*F)[256] = calloc(256, sizeof *F), (*C)[257] = calloc(256, sizeof *C);
    uint8_t *lookup = NULL;
    if (cp >= end) goto done;
    int shift = *cp >> 4, comp = *cp & 1; cp++;
    if (shift != 10 && shift != 12) goto done;
    const uint8_t *tp = cp, *tend = end;
    if (comp) {                                               /* table itself rANS order-0 (N=4) compressed */
        uint32_t ulen, clen; int k;
        if ((k = get_u7(cp, end, &ulen)) < 0) goto done;
        cp += k;
        if ((k = get_u7(cp, end, &clen)) < 0) goto done;
        cp += k;
        if (cp + clen > end || ulen > 256 * 256 * 6) goto done;
        tab_free = malloc(ulen + 1);
        if (dec_o0_alloc(cp, clen, tab_free, ulen, 4)) goto done;
        tp = tab_free; tend = tab_free + ulen; cp += clen;
    }
    uint8_t A[256];
    if (!(tp = get_alphabet(tp, tend, A))) goto done;
    lookup = malloc((size_t)256 << shift);
    for (int i = 0; i < 256; i++) {
        if (!A[i]) continue;
        uint32_t tot = 0; int run = 0;
        for (int j = 0; j < 256; j++) {
            if (!A[j]) continue;
            if (run) { run--; continue; }
            int k = get_u7(tp, tend, &F[i][j]);
            if (k < 0) goto done;
            tp += k; tot += F[i][j];
            if (!F[i][j]) { if (tp >= tend) goto done; run = *tp++; }
        }
        if (tot > (1u << shift) || (tot & (tot - 1))) goto done;
        shift_up(F[i], tot, 1u << shift);
        uint32_t x = 0;
        for (int j = 0; j < 256; j++) { C[i][j] = x; if (F[i][j]) memset(lookup + ((size_t)i << shift) + x, j, F[i][j]); x += F[i][j]; }
        C[i][256] = x;
    }
    if (!comp) cp = tp;
    if (cp + 4 * N > end) goto done;
    uint32_t R[32]; size_t idx[32]; unsigned l[32];
    size_t isz = out_sz / N;
    for (int z = 0; z < N; z++, cp += 4) { R[z] = cp[0] | (cp[1] << 8) | (cp[2] << 16) | ((uint32_t)cp[3] << 24); idx[z] = z * isz; l[z] = 0; }
    const uint32_t mask = (1u << shift) - 1;
    for (size_t s = 0; s < isz; s++)
        for (int z = 0; z < N; z++) {
            uint32_t m = R[z] & mask;
            if (m >= C[l[z]][256]) goto done;
            uint8_t c = lookup[((size_t)l[z] << shift) + m];
            out[idx[z]++] = c;
            R[z] = F[l[z]][c] * (R[z] >> shift) + m - C[l[z]][c];
            if (R[z] < RANS_L) { if (cp + 2 > end) goto done; R[z] = (R[z] << 16) | cp[0] | (cp[1] << 8); cp += 2; }
            l[z] = c;
        }
    for (; idx[N - 1] < out_sz;) {
        int z = N - 1;
        uint32_t m = R[z] & mask;
        if (m >= C[l[z]][256]) goto done;
        uint8_t c = lookup[((size_t)l[z] << shift) + m];
        out[idx[z]++] = c;
        R[z] = F[l[z]][c] * (R[z] >> shift) + m - C[l[z]][c];
        if (R[z] < RANS_L) { if (cp + 2 > end) goto done; R[z] = (R[z] << 16) | cp[0] | (cp[1] << 8); cp += 2; }
        l[z] = c;
    }
    rc = 0;
done:
    free(F); free(C); free(lookup); free(tab_free);
    return rc;
}
static int dec_o0_alloc(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_sz, int N) { return dec_o0(in, in_size, out, out_sz, N); }

/* ---- PACK (bit packing of <= 16 distinct symbols) ----------------------------------------- */
static size_t pack(const uint8_t *in, size_t n, uint8_t *meta, int *meta_len, uint8_t *out)
{
    int used[256] = {0}, map[256], nsym = 0;
    for (size_t i = 0; i < n; i++) used[in[i]] = 1;
    for (int j = 0; j < 256; j++) nsym += used[j];
    if (nsym > 16) return (size_t)-1;
    nsym = 0;
    for (int j = 0; j < 256; j++) if (used[j]) { map[j] = nsym; meta[1 + nsym] = (uint8_t)j; nsym++; }
    meta[0] = (uint8_t)nsym; *meta_len = 1 + nsym;
    if (nsym <= 1) return 0;
    int bits = nsym <= 2 ? 1 : nsym <= 4 ? 2 : 4, per = 8 / bits;
    size_t o = 0;
    for (size_t i = 0; i < n; i += per) {
        unsigned v = 0;
        for (int k = 0; k < per && i + k < n; k++) v |= (unsigned)map[in[i + k]] << (k * bits);
        out[o++] = (uint8_t)v;
    }
    return o;
}
static int unpack(const uint8_t *in, size_t n, uint8_t *out, size_t out_sz, int nsym, const uint8_t *map)
{
    if (nsym <= 1) { memset(out, nsym ? map[0] : 0, out_sz); return 0; }
    int bits = nsym <= 2 ? 1 : nsym <= 4 ? 2 : 4, per = 8 / bits;
    if (n < (out_sz + per - 1) / per) return -1;
    for (size_t i = 0; i < out_sz; i++) {
        unsigned v = (in[i / per] >> ((i % per) * bits)) & ((1u << bits) - 1);
        if ((int)v >= nsym) return -1;
        out[i] = map[v];
    }
    return 0;
}

/* ---- RLE: runs of selected symbols -> literal once + run length (uint7) in the meta stream ------ */
static size_t rle_encode(const uint8_t *in, size_t n, uint8_t *meta, size_t *meta_len, uint8_t *lit)
{
    /* a symbol is run-length coded when its repeats outnumber its run starts */
    long score[256] = {0};
    for (size_t i = 0; i < n; i++) score[in[i]] += (i && in[i] == in[i - 1]) ? 1 : -1;
    int nr = 0;
    uint8_t *mp = meta + 1;
    int isr[256];
    for (int j = 0; j < 256; j++) { isr[j] = score[j] > 0; if (isr[j]) { *mp++ = (uint8_t)j; nr++; } }
    meta[0] = (uint8_t)nr;                                    /* 0 would mean 256: never produced here */
    if (!nr) { *meta_len = 0; return (size_t)-1; }
    size_t o = 0;
    for (size_t i = 0; i < n;) {
        uint8_t c = in[i];
        lit[o++] = c;
        if (isr[c]) {
            size_t r = 1;
            while (i + r < n && in[i + r] == c) r++;
            mp += put_u7(mp, (uint32_t)(r - 1));
            i += r;
        } else i++;
    }
    *meta_len = (size_t)(mp - meta);
    return o;
}
static int rle_decode(const uint8_t *lit, size_t lit_len, const uint8_t *meta, size_t meta_len, uint8_t *out, size_t out_sz)
{
    const uint8_t *mp = meta, *mend = meta + meta_len;
    if (mp >= mend) return -1;
    int nr = *mp++, isr[256] = {0};
    if (nr == 0) nr = 256;
    for (int k = 0; k < nr; k++) { if (mp >= mend) return -1; isr[*mp++] = 1; }
    size_t o = 0;
    for (size_t i = 0; i < lit_len; i++) {
        uint8_t c = lit[i];
        uint32_t r = 0;
        if (isr[c]) { int k = get_u7(mp, mend, &r); if (k < 0) return -1; mp += k; }
        if (o + r + 1 > out_sz) return -1;
        memset(out + o, c, r + 1); o += r + 1;
    }
    return o == out_sz ? 0 : -1;
}

/* ---- top level ------------------------------------------------------------------------------ */
ORC_EXPORT size_t orc_ransnx16_compress_bound(size_t n) { return (size_t)(1.05 * n) + 4 * 257 * 257 * 3 + 8192; }

static size_t compress_inner(const uint8_t *in, size_t n, uint8_t *out, int flags);

ORC_EXPORT size_t orc_ransnx16_compress(const uint8_t *in, size_t n, uint8_t *out, int flags)
{
    return compress_inner(in, n, out, flags & 0xff);
}

static size_t compress_inner(const uint8_t *in, size_t n, uint8_t *out, int flags)
{
    const int N = (flags & F_X32) ? 32 : 4;
    uint8_t *cp = out + 1;
    if (!(flags & F_NOSZ)) cp += put_u7(cp, (uint32_t)n);
    if (flags & F_STRIPE) {
        const int S = 4;
        flags &= ~(F_PACK | F_RLE | F_CAT);
        out[0] = (uint8_t)flags;
        *cp++ = (uint8_t)S;
        uint8_t *lens = cp;                                   /* reserve: written after the sub-streams are known */
        uint8_t *tmp = malloc(n / S + 8), *sub[4]; size_t slen[4];
        for (int k = 0; k < S; k++) {
            size_t m = n / S + ((n % S) > (size_t)k);
            for (size_t j = 0; j < m; j++) tmp[j] = in[j * S + k];
            sub[k] = malloc(orc_ransnx16_compress_bound(m));
            slen[k] = compress_inner(tmp, m, sub[k], (flags & (F_ORDER | F_X32)) | F_NOSZ);
        }
        for (int k = 0; k < S; k++) lens += put_u7(lens, (uint32_t)slen[k]);
        cp = lens;
        for (int k = 0; k < S; k++) { memcpy(cp, sub[k], slen[k]); cp += slen[k]; free(sub[k]); }
        free(tmp);
        return (size_t)(cp - out);
    }
    uint8_t *packed = NULL, *lit = NULL, *rmeta = NULL;
    const uint8_t *cur = in; size_t cur_n = n;
    if (flags & F_PACK) {
        uint8_t meta[20]; int ml = 0;
        packed = malloc(n + 8);
        size_t pl = n ? pack(in, n, meta, &ml, packed) : (size_t)-1;
        if (pl == (size_t)-1) flags &= ~F_PACK;
        else { memcpy(cp, meta, ml); cp += ml; cp += put_u7(cp, (uint32_t)pl); cur = packed; cur_n = pl; }
    }
    if (flags & F_RLE) {
        lit = malloc(cur_n + 8); rmeta = malloc(cur_n * 5 + 300);
        size_t ml = 0, ll = cur_n ? rle_encode(cur, cur_n, rmeta, &ml, lit) : (size_t)-1;
        if (ll == (size_t)-1) flags &= ~F_RLE;
        else {
            /* the meta stream is itself order-0 4-way coded when that is smaller than storing it raw */
            uint8_t *cm = malloc(orc_ransnx16_compress_bound(ml));
            size_t cl = enc_o0(rmeta, ml, cm, 4);
            if (cl + 5 < ml) {
                cp += put_u7(cp, (uint32_t)(ml * 2));
                cp += put_u7(cp, (uint32_t)ll);
                cp += put_u7(cp, (uint32_t)cl);
                memcpy(cp, cm, cl); cp += cl;
            } else {
                cp += put_u7(cp, (uint32_t)(ml * 2 + 1));    /* meta stored raw */
                cp += put_u7(cp, (uint32_t)ll);
                memcpy(cp, rmeta, ml); cp += ml;
            }
            free(cm);
            cur = lit; cur_n = ll;
        }
    }
    if ((flags & F_ORDER) && cur_n < (size_t)N * 2) flags &= ~F_ORDER;   /* tiny inputs: order 0 */
    out[0] = (uint8_t)flags;
    /* CAT replaces only the entropy coder: the PACK / RLE transforms above still apply */
    if (flags & F_CAT) { memcpy(cp, cur, cur_n); cp += cur_n; }
    else if (cur_n) cp += (flags & F_ORDER) ? enc_o1(cur, cur_n, cp, N) : enc_o0(cur, cur_n, cp, N);
    free(packed); free(lit); free(rmeta);
    return (size_t)(cp - out);
}

static int uncompress_inner(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size, long known);

ORC_EXPORT int orc_ransnx16_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size)
{
    return uncompress_inner(in, in_size, out, out_cap, out_size, -1);
}

static int uncompress_inner(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size, long known)
{
    const uint8_t *cp = in, *end = in + in_size;
    if (in_size < 1) return -1;
    const int flags = *cp++, N = (flags & F_X32) ? 32 : 4;
    uint32_t ulen;
    if (flags & F_NOSZ) { if (known < 0) return -1; ulen = (uint32_t)known; }
    else { int k = get_u7(cp, end, &ulen); if (k < 0) return -1; cp += k; }
    if (ulen > out_cap) return -1;
    *out_size = ulen;
    if (flags & F_STRIPE) {
        if (cp >= end) return -1;
        int S = *cp++;
        if (S < 1 || S > 32) return -1;
        uint32_t cl[32];
        for (int k = 0; k < S; k++) { int r = get_u7(cp, end, &cl[k]); if (r < 0) return -1; cp += r; }
        uint8_t *tmp = malloc(ulen / S + 8);
        for (int k = 0; k < S; k++) {
            size_t m = ulen / S + ((ulen % S) > (uint32_t)k), got = 0;
            if (cp + cl[k] > end || uncompress_inner(cp, cl[k], tmp, m, &got, (long)m) || got != m) { free(tmp); return -1; }
            for (size_t j = 0; j < m; j++) out[j * S + k] = tmp[j];
            cp += cl[k];
        }
        free(tmp);
        return 0;
    }
    int nsym = 0; uint8_t map[16] = {0}; uint32_t plen = ulen;
    if (flags & F_PACK) {
        if (cp >= end) return -1;
        nsym = *cp++;
        if (nsym > 16 || cp + nsym > end) return -1;
        memcpy(map, cp, nsym); cp += nsym;
        int k = get_u7(cp, end, &plen); if (k < 0) return -1; cp += k;
        if (plen > ulen) return -1;                           /* resource guard: packing never grows the data */
    }
    uint32_t rmeta_len = 0, lit_len = plen; const uint8_t *rmeta = NULL; uint8_t *rmeta_free = NULL;
    if (flags & F_RLE) {
        uint32_t v; int k = get_u7(cp, end, &v); if (k < 0) return -1; cp += k;
        k = get_u7(cp, end, &lit_len); if (k < 0) return -1; cp += k;
        rmeta_len = v >> 1;
        /* resource guards: every literal yields >= 1 byte; the meta stream holds at most a 257-byte symbol
         * list and one 5-byte run length per literal */
        if (lit_len > plen || rmeta_len > 5ull * lit_len + 257) return -1;
        if (v & 1) { if (cp + rmeta_len > end) return -1; rmeta = cp; cp += rmeta_len; }
        else {
            uint32_t cl; k = get_u7(cp, end, &cl); if (k < 0 || cp + k + cl > end) return -1; cp += k;
            rmeta_free = malloc(rmeta_len + 1);
            if (dec_o0(cp, cl, rmeta_free, rmeta_len, 4)) { free(rmeta_free); return -1; }
            rmeta = rmeta_free; cp += cl;
        }
    }
    int rc = 0;
    uint8_t *stage1 = (flags & (F_RLE | F_PACK)) ? malloc((size_t)lit_len + 8) : out;
    if (flags & F_CAT) { if ((size_t)(end - cp) < lit_len) rc = -1; else memcpy(stage1, cp, lit_len); }
    else if (lit_len) rc = (flags & F_ORDER) ? dec_o1(cp, (size_t)(end - cp), stage1, lit_len, N) : dec_o0(cp, (size_t)(end - cp), stage1, lit_len, N);
    uint8_t *stage2 = stage1;
    if (!rc && (flags & F_RLE)) {
        stage2 = (flags & F_PACK) ? malloc((size_t)plen + 8) : out;
        rc = rle_decode(stage1, lit_len, rmeta, rmeta_len, stage2, plen);
    }
    if (!rc && (flags & F_PACK)) rc = unpack(stage2, plen, out, ulen, nsym, map);
    if (stage2 != stage1 && stage2 != out) free(stage2);
    if (stage1 != out) free(stage1);
    free(rmeta_free);
    return rc;
}

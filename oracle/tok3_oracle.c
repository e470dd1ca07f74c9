/*
 * tok3_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * *** PARITY UNPINNED ***  Plain-C restatement of the CRAM 3.1 read-name tokeniser (CRAM block
 * method 8), the codec behind
 *      tok3_decode_names(in, in_size, &out_len)                          cram/cram_io.c:1737
 *      tok3_encode_names(in, in_size, level, use_arith, &out_len, NULL)  cram/cram_io.c:1891
 * whose implementation (htscodecs v1.6.6 tokenise_name3.c) is an ABSENT git submodule of the reference,
 * with no golden stream in the reference's tests.  This file follows the published specification
 * (hts-specs "CRAM codecs" v3.1, chapter "Name tokenisation codec") as summarised in SURVEY.md Appendix
 * A.6; the header layout (byte 8 = entropy back-end) is confirmed by cram/cram_external.c:641-647, the NUL
 * separator by NEWS:278.  Byte-level agreement with htscodecs is UNVERIFIED.  The DECODER below is the
 * contract (it accepts any stream the specification allows); the ENCODER's tokenisation choices are
 * this restatement's own (the specification leaves them free).
 *
 * Container:  ulen:le32  nnames:le32  use_arith:u8  then token byte streams until the end of input:
 *     ttype:u8   bit7 = first stream of the next token position, bit6 = duplicate, low 4 bits = type
 *     duplicate: dup_pos:u8 dup_type:u8                   (stream content = that earlier stream)
 *     else     : clen:uint7, clen bytes = a complete rANS Nx16 (use_arith 0) or range-coder (1) stream
 *   When the first stream of a position is not TYPE(0), that position's TYPE stream is implied:
 *   [that type, MATCH, MATCH, ...] (nnames entries).
 * Name n:  position 0 holds DUP(5) or DIFF(6) with a le32 distance d -> reference name m = n - d.
 *   DUP copies name m.  DIFF reads one token per position until END(12):
 *   STRING(1) NUL-terminated text, CHAR(2) one byte, DIGITS(7) le32 number, DIGITS0(3) le32 number
 *   left-padded with zeros to DZLEN(4) u8 width, DELTA(8) = number of name m's token + u8,
 *   DELTA0(9) same but zero-padded to the width of name m's token, MATCH(10) = name m's token,
 *   NOP(11) nothing.  Every name is written followed by a NUL.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define ORC_EXPORT __attribute__((visibility("default")))
enum { T_TYPE = 0, T_STRING = 1, T_CHAR = 2, T_DIGITS0 = 3, T_DZLEN = 4, T_DUP = 5, T_DIFF = 6, T_DIGITS = 7,
       T_DELTA = 8, T_DELTA0 = 9, T_MATCH = 10, T_NOP = 11, T_END = 12 };
#define MAX_TOK 128

/* the entropy back-ends live in the sibling oracle files */
size_t orc_ransnx16_compress(const uint8_t *in, size_t n, uint8_t *out, int flags);
int orc_ransnx16_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size);
size_t orc_ransnx16_compress_bound(size_t n);
size_t orc_arith_compress(const uint8_t *in, size_t n, uint8_t *out, int flags);
int orc_arith_uncompress(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size, long known);
size_t orc_arith_compress_bound(size_t n);

typedef struct { uint8_t *b; size_t n, cap, pos; } buf_t;
static void buf_put(buf_t *s, const void *p, size_t n)
{
    if (s->n + n > s->cap) { s->cap = (s->n + n) * 2 + 64; s->b = realloc(s->b, s->cap); }
    memcpy(s->b + s->n, p, n); s->n += n;
}
static void buf_put8(buf_t *s, uint8_t v) { buf_put(s, &v, 1); }
static void buf_put32(buf_t *s, uint32_t v) { uint8_t t[4] = {v, v >> 8, v >> 16, v >> 24}; buf_put(s, t, 4); }

static int put_u7(uint8_t *cp, uint32_t v)
{
    int n = 0; uint8_t tmp[5];
    do { tmp[n++] = v & 0x7f; v >>= 7; } while (v);
    for (int i = n - 1; i >= 0; i--) *cp++ = tmp[i] | (i ? 0x80 : 0);
    return n;
}
static int get_u7(const uint8_t *cp, const uint8_t *end, uint32_t *v)
{
    uint32_t x = 0; int n = 0; uint8_t c;
    do { if (cp + n >= end || n >= 5) return -1; c = cp[n++]; x = (x << 7) | (c & 0x7f); } while (c & 0x80);
    *v = x; return n;
}

/* one token of a finished name, as later names may refer to it */
typedef struct { uint32_t off, len, val; uint8_t numeric; } tokrec_t;

/* ------------------------------------------------------------------------------------------ decoder */
ORC_EXPORT int orc_tok3_decode(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap, size_t *out_size)
{
    if (in_size < 9) return -1;
    const uint32_t ulen = in[0] | in[1] << 8 | in[2] << 16 | (uint32_t)in[3] << 24;
    const uint32_t nn = in[4] | in[5] << 8 | in[6] << 16 | (uint32_t)in[7] << 24;
    const int use_arith = in[8];
    if (ulen > out_cap || use_arith > 1) return -1;
    if (nn > ulen) return -1;                                   /* every name costs at least its NUL */
    *out_size = ulen;
    static buf_t S[MAX_TOK][16];                                /* not re-entrant: test infrastructure */
    uint8_t owned[MAX_TOK][16];
    memset(S, 0, sizeof S); memset(owned, 0, sizeof owned);
    int rc = 0, t = -1;
    unsigned long long total_stream = 0;
    const uint8_t *cp = in + 9, *end = in + in_size;
    while (cp < end && !rc) {
        const uint8_t tt = *cp++;
        const int type = tt & 15;
        if (tt & 0x80) {
            if (++t >= MAX_TOK) { rc = -1; break; }
            if (type != T_TYPE) {                               /* implied TYPE stream */
                if (!nn) { rc = -1; break; }
                S[t][0].b = malloc(nn); S[t][0].n = nn; owned[t][0] = 1;
                memset(S[t][0].b, T_MATCH, nn); S[t][0].b[0] = (uint8_t)type;
            }
        }
        if (t < 0 || type > T_END || S[t][type].b) { rc = -1; break; }      /* stream given twice / before any position */
        if (tt & 0x40) {
            if (end - cp < 2) { rc = -1; break; }
            const int dp = cp[0], dt = cp[1]; cp += 2;
            if (dp > t || dt > T_END || !S[dp][dt].b || (dp == t && dt == type)) { rc = -1; break; }
            S[t][type].b = S[dp][dt].b; S[t][type].n = S[dp][dt].n;
        } else {
            uint32_t clen; int k = get_u7(cp, end, &clen);
            if (k < 0 || (size_t)(end - cp - k) < clen) { rc = -1; break; }
            cp += k;
            /* the stream's own header carries its size; bound it by what a name block can need */
            const size_t cap = (size_t)ulen * 5 + 64;
            uint8_t *b = malloc(cap + 1); size_t got = 0;
            int r = use_arith ? orc_arith_uncompress(cp, clen, b, cap, &got, -1) : orc_ransnx16_uncompress(cp, clen, b, cap, &got);
            if (r) { free(b); rc = -1; break; }
            total_stream += got;
            if (total_stream > 16ull * ulen + 65536u) { free(b); rc = -1; break; }   /* resource guard, as in the GPU planner */
            S[t][type].b = b; S[t][type].n = got; owned[t][type] = 1;
            cp += clen;
        }
    }
    const int ntokpos = t + 1;
    /* per-name records: first token index, token count, output offset */
    uint32_t *first = malloc(((size_t)nn + 1) * 4), *ntok = malloc(((size_t)nn + 1) * 4), *noff = malloc(((size_t)nn + 1) * 4);
    tokrec_t *rec = NULL; size_t nrec = 0, caprec = 0;
    size_t o = 0;
    for (uint32_t n = 0; n < nn && !rc; n++) {
        buf_t *T0 = &S[0][T_TYPE];
        if (ntokpos < 1 || T0->pos >= T0->n) { rc = -1; break; }
        const int ty0 = T0->b[T0->pos++];
        if (ty0 != T_DUP && ty0 != T_DIFF) { rc = -1; break; }
        buf_t *D = &S[0][ty0];
        if (D->pos + 4 > D->n) { rc = -1; break; }
        const uint32_t dist = D->b[D->pos] | D->b[D->pos + 1] << 8 | D->b[D->pos + 2] << 16 | (uint32_t)D->b[D->pos + 3] << 24;
        D->pos += 4;
        if (dist > n) { rc = -1; break; }
        const uint32_t m = n - dist;                            /* m == n: no reference name */
        noff[n] = (uint32_t)o; first[n] = (uint32_t)nrec; ntok[n] = 0;
        if (ty0 == T_DUP) {
            if (m == n) { rc = -1; break; }
            const uint32_t len = (m + 1 < n + 1 ? noff[m + 1] : 0) - noff[m];   /* includes the NUL */
            if (o + len > ulen) { rc = -1; break; }
            memmove(out + o, out + noff[m], len); o += len;
            first[n] = first[m]; ntok[n] = ntok[m];             /* later names see the same tokens */
            noff[n + 1] = (uint32_t)o;
            continue;
        }
        for (int tp = 1;; tp++) {
            if (tp >= ntokpos || tp >= MAX_TOK) { rc = -1; break; }
            buf_t *TY = &S[tp][T_TYPE];
            if (TY->pos >= TY->n) { rc = -1; break; }
            const int ty = TY->b[TY->pos++];
            tokrec_t R = {(uint32_t)o, 0, 0, 0};
            const tokrec_t *P = (m != n && (uint32_t)(tp - 1) < ntok[m]) ? &rec[first[m] + tp - 1] : NULL;
            char num[16]; int nl = 0;
            buf_t *V = ty <= T_END ? &S[tp][ty] : NULL;
            switch (ty) {
            case T_STRING: {
                size_t e = V->pos;
                while (e < V->n && V->b[e]) e++;
                if (e >= V->n) { rc = -1; break; }
                R.len = (uint32_t)(e - V->pos);
                if (o + R.len > ulen) { rc = -1; break; }
                memcpy(out + o, V->b + V->pos, R.len); V->pos = e + 1;
                break; }
            case T_CHAR:
                if (V->pos >= V->n || o + 1 > ulen) { rc = -1; break; }
                out[o] = V->b[V->pos++]; R.len = 1;
                break;
            case T_DIGITS: case T_DIGITS0: {
                if (V->pos + 4 > V->n) { rc = -1; break; }
                R.val = V->b[V->pos] | V->b[V->pos + 1] << 8 | V->b[V->pos + 2] << 16 | (uint32_t)V->b[V->pos + 3] << 24;
                V->pos += 4; R.numeric = 1;
                int width = 0;
                if (ty == T_DIGITS0) { buf_t *Z = &S[tp][T_DZLEN]; if (Z->pos >= Z->n) { rc = -1; break; } width = Z->b[Z->pos++]; }
                nl = snprintf(num, sizeof num, "%0*u", width > 15 ? 15 : width, R.val);
                break; }
            case T_DELTA: case T_DELTA0: {
                if (!P || V->pos >= V->n) { rc = -1; break; }
                R.val = P->val + V->b[V->pos++]; R.numeric = 1;
                const int width = ty == T_DELTA0 ? (int)(P->len > 15 ? 15 : P->len) : 0;
                nl = snprintf(num, sizeof num, "%0*u", width, R.val);
                break; }
            case T_MATCH:
                if (!P) { rc = -1; break; }
                R.len = P->len; R.val = P->val; R.numeric = P->numeric;
                if (o + R.len > ulen) { rc = -1; break; }
                memmove(out + o, out + P->off, R.len);
                break;
            case T_NOP: case T_END: break;
            default: rc = -1;
            }
            if (rc) break;
            if (nl) { if (o + (size_t)nl > ulen) { rc = -1; break; } memcpy(out + o, num, (size_t)nl); R.len = (uint32_t)nl; }
            o += R.len;
            if (nrec == caprec) { caprec = caprec * 2 + 1024; rec = realloc(rec, caprec * sizeof *rec); }
            rec[nrec++] = R; ntok[n]++;
            if (ty == T_END) break;
        }
        if (rc) break;
        if (o + 1 > ulen) { rc = -1; break; }
        out[o++] = 0;
        noff[n + 1] = (uint32_t)o;
    }
    if (!rc && o != ulen) rc = -1;
    for (int a = 0; a < MAX_TOK; a++) for (int b = 0; b < 16; b++) if (owned[a][b]) free(S[a][b].b);
    free(first); free(ntok); free(noff); free(rec);
    return rc;
}

/* ------------------------------------------------------------------------------------------ encoder */
typedef struct { uint8_t cls; uint32_t off, len, val; } etok_t;     /* cls: T_STRING / T_CHAR / T_DIGITS / T_DIGITS0 */

static int is_alpha(int c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
static int is_digit(int c) { return c >= '0' && c <= '9'; }

static int tokenise(const uint8_t *s, uint32_t base, uint32_t len, etok_t *T)
{
    int nt = 0; uint32_t i = 0;
    while (i < len) {
        etok_t *E = &T[nt];
        E->off = base + i; E->val = 0;
        if (nt == MAX_TOK - 3) { E->cls = T_STRING; E->len = len - i; nt++; break; }   /* out of positions: the rest is text */
        if (is_digit(s[i])) {
            uint32_t e = i;
            while (e < len && e - i < 9 && is_digit(s[e])) e++;
            E->len = e - i;
            for (uint32_t k = i; k < e; k++) E->val = E->val * 10 + (s[k] - '0');
            E->cls = (s[i] == '0' && E->len > 1) ? T_DIGITS0 : T_DIGITS;
            i = e;
        } else if (is_alpha(s[i])) {
            uint32_t e = i;
            while (e < len && is_alpha(s[e])) e++;
            E->len = e - i; E->cls = E->len == 1 ? T_CHAR : T_STRING;
            i = e;
        } else { E->len = 1; E->cls = T_CHAR; i++; }
        nt++;
    }
    return nt;
}

/* smallest of a fixed list of back-end settings; ties keep the earlier one */
static size_t best_entropy(const uint8_t *d, size_t n, int type, int use_arith, uint8_t *out)
{
    static const int sets[] = {0, 1, 64, 65, 128, 129, 8, 9};
    const int nsets = (type == T_DIGITS || type == T_DIGITS0 || type == T_DUP || type == T_DIFF) ? 8 : 6;
    size_t bound = use_arith ? orc_arith_compress_bound(n) : orc_ransnx16_compress_bound(n), best = (size_t)-1;
    uint8_t *tmp = malloc(bound);
    for (int k = 0; k < nsets; k++) {
        size_t l = use_arith ? orc_arith_compress(d, n, tmp, sets[k]) : orc_ransnx16_compress(d, n, tmp, sets[k]);
        if (l < best) { best = l; memcpy(out, tmp, l); }
    }
    free(tmp);
    return best;
}

ORC_EXPORT size_t orc_tok3_compress_bound(size_t n) { return n * 2 + 65536 + 13 * MAX_TOK * 64; }

/* returns the stream size, or 0 when the input is not a list of NUL-terminated names */
ORC_EXPORT size_t orc_tok3_encode(const uint8_t *in, size_t n, uint8_t *out, int use_arith)
{
    if (n && in[n - 1] != 0) return 0;
    static buf_t S[MAX_TOK][16];
    memset(S, 0, sizeof S);
    etok_t *cur = malloc(MAX_TOK * sizeof *cur), *prev = malloc(MAX_TOK * sizeof *prev);
    int pn = 0; uint32_t nn = 0, poff = 0, plen = 0; int maxpos = 0;
    for (size_t i = 0; i < n;) {
        size_t e = i;
        while (in[e]) e++;
        const uint32_t len = (uint32_t)(e - i);
        if (nn && len == plen && !memcmp(in + i, in + poff, len)) {
            buf_put8(&S[0][T_TYPE], T_DUP); buf_put32(&S[0][T_DUP], 1);
            if (maxpos < 1) maxpos = 1;
        } else {
            buf_put8(&S[0][T_TYPE], T_DIFF); buf_put32(&S[0][T_DIFF], nn ? 1 : 0);
            const int nt = tokenise(in + i, (uint32_t)i, len, cur);
            for (int t = 0; t <= nt; t++) {
                const int tp = t + 1;
                if (t == nt) { buf_put8(&S[tp][T_TYPE], T_END); break; }
                const etok_t *C = &cur[t], *P = (nn && t < pn) ? &prev[t] : NULL;
                if (P && P->cls == C->cls && P->len == C->len && !memcmp(in + P->off, in + C->off, C->len)) buf_put8(&S[tp][T_TYPE], T_MATCH);
                else if (P && C->cls == T_DIGITS && P->cls == T_DIGITS && C->val >= P->val && C->val - P->val < 256) {
                    buf_put8(&S[tp][T_TYPE], T_DELTA); buf_put8(&S[tp][T_DELTA], (uint8_t)(C->val - P->val));
                } else if (P && C->cls == T_DIGITS0 && P->cls == T_DIGITS0 && C->len == P->len && C->val >= P->val && C->val - P->val < 256) {
                    buf_put8(&S[tp][T_TYPE], T_DELTA0); buf_put8(&S[tp][T_DELTA0], (uint8_t)(C->val - P->val));
                } else {
                    buf_put8(&S[tp][T_TYPE], C->cls);
                    if (C->cls == T_STRING) { buf_put(&S[tp][T_STRING], in + C->off, C->len); buf_put8(&S[tp][T_STRING], 0); }
                    else if (C->cls == T_CHAR) buf_put8(&S[tp][T_CHAR], in[C->off]);
                    else { buf_put32(&S[tp][C->cls], C->val); if (C->cls == T_DIGITS0) buf_put8(&S[tp][T_DZLEN], (uint8_t)C->len); }
                }
            }
            if (maxpos < nt + 2) maxpos = nt + 2;
            etok_t *sw = prev; prev = cur; cur = sw; pn = nt;
        }
        poff = (uint32_t)i; plen = len; nn++;
        i = e + 1;
    }
    uint8_t *cp = out;
    for (int k = 0; k < 4; k++) *cp++ = (uint8_t)((uint32_t)n >> (8 * k));
    for (int k = 0; k < 4; k++) *cp++ = (uint8_t)(nn >> (8 * k));
    *cp++ = (uint8_t)(use_arith ? 1 : 0);
    int *em = malloc(MAX_TOK * 16 * sizeof *em), nem = 0;           /* streams written so far, (pos << 4 | type) */
    for (int t = 0; t < maxpos; t++) {
        int first = 1;
        /* TYPE stream [X, MATCH...] over all names is implied by the first value stream of type X */
        int implied = -1;
        buf_t *TY = &S[t][T_TYPE];
        if (t > 0 && TY->n == nn && TY->b[0] != T_TYPE && TY->b[0] != T_MATCH && TY->b[0] <= T_DELTA0 && S[t][TY->b[0]].n) {
            implied = TY->b[0];
            for (uint32_t k = 1; k < nn; k++) if (TY->b[k] != T_MATCH) { implied = -1; break; }
        }
        for (int pass = 0; pass < 2; pass++)
        for (int ty = 0; ty <= T_END; ty++) {
            /* with an implied TYPE stream the stream of that type goes first (pass 0), the rest follow */
            if (implied >= 0) { if (ty == T_TYPE) continue; if ((pass == 0) != (ty == implied)) continue; }
            else if (pass) continue;
            buf_t *B = &S[t][ty];
            if (!B->n) continue;
            uint8_t tt = (uint8_t)ty | (first ? 0x80 : 0);
            first = 0;
            int dp = -1, dt = -1;                                  /* identical to a stream already written? */
            for (int k = 0; k < nem && dp < 0; k++) {
                const buf_t *E = &S[em[k] >> 4][em[k] & 15];
                if (E->n == B->n && !memcmp(E->b, B->b, B->n)) { dp = em[k] >> 4; dt = em[k] & 15; }
            }
            em[nem++] = t << 4 | ty;
            if (dp >= 0) { *cp++ = tt | 0x40; *cp++ = (uint8_t)dp; *cp++ = (uint8_t)dt; continue; }
            *cp++ = tt;
            uint8_t *tmp = malloc(use_arith ? orc_arith_compress_bound(B->n) : orc_ransnx16_compress_bound(B->n));
            size_t l = best_entropy(B->b, B->n, ty, use_arith, tmp);
            cp += put_u7(cp, (uint32_t)l);
            memcpy(cp, tmp, l); cp += l;
            free(tmp);
        }
    }
    for (int a = 0; a < MAX_TOK; a++) for (int b = 0; b < 16; b++) free(S[a][b].b);
    free(cur); free(prev); free(em);
    return (size_t)(cp - out);
}

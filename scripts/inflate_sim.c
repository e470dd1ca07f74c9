/* inflate_sim.c -- CPU model of the lane-parallel BGZF inflate (design tool, not product, not oracle).
 *
 * Models what the gfx950 kernel "v2" does inside one deflate block, with NL lanes (4 wavefronts) per BGZF block:
 *   P: speculative parse.  The block's bits are cut into NL segments of S bits; lane i parses the tokens that START in
 *      its segment, beginning at an entry bit e[i] (guess: the segment start; truth: where lane i-1's last token ended).
 *      Passes repeat until the entries are consistent -- Huffman streams re-synchronise within a few symbols, so most
 *      lanes land on true token boundaries in the first pass.
 *   W: every lane re-parses its tokens and writes them: literals directly, matches when their source bytes exist (the
 *      source may belong to an earlier lane that is still working: lanes publish their write cursors).
 * The model runs the lanes in lock step and reports how many passes / iterations the data needs, and checks the result
 * byte for byte against the serial decode.     gcc -O2 -o inflate_sim scripts/inflate_sim.c && ./inflate_sim file.bgzf
 *
 * Finding (10 GiB-config BAM, 256 lanes x 512 bits, profiles/r02_inflate_design_model.txt): the parse needs ~20 passes per
 * block (re-synchronisation is slower than hoped: a chain of ~20 consecutive lanes has to be corrected one after the
 * other) and -- decisive -- the write phase degenerates into a systolic pipeline: a sorted BAM copies every record from
 * the previous one (~1 lane back at the same relative offset), so each lane waits for its predecessor (34 stalls per
 * token, 2575 lock-step iterations instead of 73).  The design was therefore NOT built; see DESIGN.md section 9.
 * Limitation: the write-phase model is only complete for blocks that fit one round (lanes x segment bits >= block bits).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/bgzf_oracle.c"

static int NL = 256, SBITS = 512;
static unsigned long long st_blocks, st_dblocks, st_rounds, st_passes, st_parse_iters, st_w_iters, st_w_copy, st_tokens, st_stalls,
    st_w_iters_ideal, st_lane_parses;

typedef struct { int kind; int len; int dist; uint32_t next; } tok_t;      /* kind 0 literal (len = byte), 1 match, 2 EOB, -1 error */

static void seek_bits(orc_state *t, const orc_state *s, uint32_t bit) {
    *t = *s; t->err = 0;
    t->in_pos = bit >> 3; t->bitbuf = 0; t->bitcnt = 0;
    if (bit & 7) getbits(t, bit & 7);
}
static uint32_t bitpos(const orc_state *s) { return (uint32_t)(s->in_pos * 8 - s->bitcnt); }

static tok_t next_token(const orc_state *s, uint32_t bit, const orc_huff *lc, const orc_huff *dc) {
    orc_state t; tok_t k; memset(&k, 0, sizeof k);
    seek_bits(&t, s, bit);
    int sym = decode_sym(&t, lc);
    if (sym < 0) { k.kind = -1; return k; }
    if (sym < 256) { k.kind = 0; k.len = sym; }
    else if (sym == 256) k.kind = 2;
    else {
        sym -= 257;
        if (sym >= 29) { k.kind = -1; return k; }
        k.kind = 1;
        k.len = len_base[sym] + getbits(&t, len_extra[sym]);
        int ds = decode_sym(&t, dc);
        if (t.err || ds < 0 || ds >= 30) { k.kind = -1; return k; }
        k.dist = dist_base[ds] + getbits(&t, dist_extra[ds]);
        if (t.err) { k.kind = -1; return k; }
    }
    k.next = bitpos(&t);
    return k;
}

#define IDLE 0xffffffffu
static int sim_codes(orc_state *s, const orc_huff *lc, const orc_huff *dc) {
    const uint32_t end_bit = (uint32_t)s->in_len * 8;
    uint32_t B = bitpos(s);
    uint32_t *e = malloc((NL + 1) * 4), *x = malloc(NL * 4), *nb = malloc(NL * 4), *nt = malloc(NL * 4), *pe = malloc((NL + 1) * 4);
    int *flag = malloc(NL * 4);
    st_dblocks++;
    for (int round = 0;; round++) {                                /* rounds of NL * S bits */
        if (round > 4096) { fprintf(stderr, "model: too many rounds\n"); return -1; }
        st_rounds++;
        for (int i = 0; i <= NL; i++) { e[i] = B + (uint32_t)i * SBITS; pe[i] = ~e[i]; }
        int passes = 0;
        for (;;) {
            if (passes > 4 * NL) { fprintf(stderr, "model: parse passes do not converge\n"); return -1; }
            int changed = 0, worst = 0;
            for (int i = 0; i < NL; i++) {
                if (e[i] == pe[i]) continue;                       /* clean: keeps its results */
                pe[i] = e[i]; changed = 1;
                x[i] = IDLE; nb[i] = 0; nt[i] = 0; flag[i] = 0;
                if (e[i] == IDLE || e[i] >= end_bit) { flag[i] = 3; continue; }
                st_lane_parses++;
                uint32_t b = e[i], lim = B + (uint32_t)(i + 1) * SBITS; int it = 0;
                while (b < lim) {
                    tok_t k = next_token(s, b, lc, dc); it++;
                    if (k.kind < 0) { flag[i] = 2; break; }
                    if (k.kind == 2) { flag[i] = 1; b = k.next; break; }
                    nb[i] += k.kind == 0 ? 1 : (uint32_t)k.len; nt[i]++;
                    b = k.next;
                }
                x[i] = b;
                if (it > worst) worst = it;
            }
            if (!changed) break;
            passes++; st_parse_iters += (unsigned long long)worst;
            for (int i = 0; i < NL; i++) { uint32_t v = flag[i] ? IDLE : x[i]; if (i + 1 <= NL && e[i + 1] != v && i + 1 < NL) e[i + 1] = v; }
        }
        st_passes += (unsigned long long)passes;
        /* valid lanes: 0..k where k is the first lane with a flag */
        int k = NL - 1, eob = 0;
        for (int i = 0; i < NL; i++) if (flag[i]) { k = i; eob = flag[i] == 1; if (flag[i] != 1) { if (flag[i] == 3 && i > 0) { k = i - 1; } else return -1; } break; }
        if (!eob && flag[k] == 3) return -1;
        /* W phase */
        uint32_t *start = malloc((k + 2) * 4), *cur = malloc((k + 1) * 4), *snap = malloc((k + 1) * 4), *bit = malloc((k + 1) * 4);
        uint32_t base = (uint32_t)s->out_pos;
        start[0] = base;
        for (int i = 0; i <= k; i++) { start[i + 1] = start[i] + nb[i]; cur[i] = start[i]; bit[i] = e[i]; }
        if (start[k + 1] > s->out_cap) return -1;
        int *left = malloc((k + 1) * 4); int maxtok = 0;
        for (int i = 0; i <= k; i++) { left[i] = (int)nt[i]; st_tokens += nt[i]; if ((int)nt[i] > maxtok) maxtok = (int)nt[i]; }
        st_w_iters_ideal += (unsigned long long)maxtok;
        int remaining = k + 1;
        for (int i = 0; i <= k; i++) if (!left[i]) remaining--;
        for (unsigned long long guard = 0; remaining > 0; guard++) {
            if (guard > 10000000ull) { fprintf(stderr, "model: W phase does not finish\n"); return -1; }
            memcpy(snap, cur, (k + 1) * 4);
            int maxcopy = 0;
            st_w_iters++;
            for (int i = 0; i <= k; i++) {
                if (!left[i]) continue;
                tok_t t = next_token(s, bit[i], lc, dc);
                if (t.kind == 0) { s->out[cur[i]++] = (uint8_t)t.len; bit[i] = t.next; if (--left[i] == 0) remaining--; continue; }
                uint32_t p = cur[i];
                if ((uint32_t)t.dist > p) return -1;
                uint32_t src = p - (uint32_t)t.dist, fe = src + (uint32_t)t.len;      /* foreign part: bytes below start[i] */
                int ready = 1;
                if (src < start[i]) {
                    if (fe > start[i]) fe = start[i];
                    /* owner lanes of [src, fe): every one must have written past the bytes we need (snapshot = what
                       the lane can see of the others) */
                    int j = i - 1;
                    while (j > 0 && start[j] > src) j--;
                    for (int q = j; q < i && ready; q++) {
                        uint32_t need = fe < start[q + 1] ? fe : start[q + 1];
                        if (start[q] < need && snap[q] < need) ready = 0;
                    }
                }
                if (!ready) { st_stalls++; continue; }
                for (int c = 0; c < t.len; c++) s->out[p + c] = s->out[src + c];
                cur[i] = p + (uint32_t)t.len; bit[i] = t.next;
                if (t.len > maxcopy) maxcopy = t.len;
                if (--left[i] == 0) remaining--;
            }
            st_w_copy += (unsigned long long)maxcopy;
        }
        s->out_pos = start[k + 1];
        free(start); free(cur); free(snap); free(bit); free(left);
        if (eob) { seek_bits(s, s, x[k]); break; }
        fprintf(stderr, "model: block needs more than one round of %d x %d bits -- not modelled\n", NL, SBITS);
        return -1;
    }
    free(e); free(x); free(nb); free(nt); free(pe); free(flag);
    return 0;
}

static int sim_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t cap, size_t *out_len) {
    orc_state s; memset(&s, 0, sizeof s);
    s.in = in; s.in_len = in_len; s.out = out; s.out_cap = cap;
    int last, rc = 0;
    do {
        last = getbits(&s, 1);
        int type = getbits(&s, 2);
        if (s.err) return -1;
        if (type == 0) rc = do_stored(&s);
        else if (type == 1 || type == 2) {
            orc_huff lc, dc; short lengths[MAXLCODES + MAXDCODES + 32];
            if (type == 1) {
                int sym;
                for (sym = 0; sym < 144; sym++) lengths[sym] = 8;
                for (; sym < 256; sym++) lengths[sym] = 9;
                for (; sym < 280; sym++) lengths[sym] = 7;
                for (; sym < FIXLCODES; sym++) lengths[sym] = 8;
                build(&lc, lengths, FIXLCODES);
                for (sym = 0; sym < MAXDCODES; sym++) lengths[sym] = 5;
                build(&dc, lengths, MAXDCODES);
            } else {
                static const short order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                int nlen = getbits(&s, 5) + 257, ndist = getbits(&s, 5) + 1, ncode = getbits(&s, 4) + 4, index;
                if (s.err || nlen > MAXLCODES || ndist > MAXDCODES) return -1;
                for (index = 0; index < ncode; index++) lengths[order[index]] = (short)getbits(&s, 3);
                for (; index < 19; index++) lengths[order[index]] = 0;
                if (build(&lc, lengths, 19) != 0) return -1;
                index = 0;
                while (index < nlen + ndist) {
                    int sym = decode_sym(&s, &lc);
                    if (sym < 0) return -1;
                    if (sym < 16) lengths[index++] = (short)sym;
                    else {
                        int len = 0, rep;
                        if (sym == 16) { if (index == 0) return -1; len = lengths[index - 1]; rep = 3 + getbits(&s, 2); }
                        else if (sym == 17) rep = 3 + getbits(&s, 3);
                        else rep = 11 + getbits(&s, 7);
                        if (s.err || index + rep > nlen + ndist) return -1;
                        while (rep--) lengths[index++] = (short)len;
                    }
                }
                int err = build(&lc, lengths, nlen);
                if (err && (err < 0 || nlen != lc.count[0] + lc.count[1])) return -1;
                err = build(&dc, lengths + nlen, ndist);
                if (err && (err < 0 || ndist != dc.count[0] + dc.count[1])) return -1;
            }
            rc = sim_codes(&s, &lc, &dc);
        } else rc = -1;
        if (rc) return -1;
    } while (!last);
    *out_len = s.out_pos;
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: inflate_sim file.bgzf [lanes] [segment bits] [max blocks]\n"); return 1; }
    if (argc > 2) NL = atoi(argv[2]);
    if (argc > 3) SBITS = atoi(argv[3]);
    long maxb = argc > 4 ? atol(argv[4]) : 400;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = malloc(n); if (fread(buf, 1, n, f) != n) return 1; fclose(f);
    uint8_t *a = malloc(65536 + 64), *b = malloc(65536 + 64);
    size_t pos = 0; unsigned long long bytes = 0;
    while (pos + 26 <= n && (long)st_blocks < maxb) {
        size_t bs = (size_t)(buf[pos + 16] | (buf[pos + 17] << 8)) + 1;
        size_t la = 0, lb = 0, used = 0;
        if (orc_inflate_raw(buf + pos + 18, bs - 18, a, 65536, &la, &used)) { fprintf(stderr, "serial decode failed\n"); return 1; }
        if (sim_inflate_raw(buf + pos + 18, bs - 18, b, 65536, &lb) || la != lb || memcmp(a, b, la)) {
            fprintf(stderr, "MODEL MISMATCH at block %llu (file offset %zu): %zu vs %zu bytes\n", st_blocks, pos, la, lb); return 2;
        }
        bytes += la; st_blocks++; pos += bs;
    }
    printf("lanes %d, segment %d bits: %llu BGZF blocks, %.1f KB plain and %.1f deflate blocks per block -- model output identical\n",
           NL, SBITS, st_blocks, bytes / 1e3 / st_blocks, (double)st_dblocks / st_blocks);
    printf("  rounds/deflate block %.2f   parse passes/round %.2f   lane parses per valid lane-round ~%.2f   lockstep parse iterations/round %.1f\n",
           (double)st_rounds / st_dblocks, (double)st_passes / st_rounds, (double)st_lane_parses / ((double)st_rounds * NL),
           (double)st_parse_iters / st_rounds);
    printf("  tokens/block %.0f   W iterations/round %.1f (ideal = longest lane: %.1f)   stalls/token %.3f   sum of per-iteration max copy length/round %.0f\n",
           (double)st_tokens / st_blocks, (double)st_w_iters / st_rounds, (double)st_w_iters_ideal / st_rounds,
           (double)st_stalls / st_tokens, (double)st_w_copy / st_rounds);
    return 0;
}

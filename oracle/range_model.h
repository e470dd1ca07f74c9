/*
 * range_model.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The byte-wise range coder and the adaptive frequency-sorted model shared by the CRAM 3.1 adaptive arithmetic coder
 * (arith_oracle.c) and the fqzcomp quality codec (fqzcomp_oracle.c): htscodecs' c_range_coder.h / c_simple_model.h (absent
 * submodule), restated from the hts-specs "CRAM codecs" v3.1 pseudocode (RangeDecodeCreate / RangeGetFrequency /
 * RangeDecode / ModelDecode / ModelRenormalise and their encode twins).  *** PARITY UNPINNED *** -- see the two users.
 */
#ifndef ORC_RANGE_MODEL_H
#define ORC_RANGE_MODEL_H
#include <stdint.h>
#include <string.h>

#define STEP 16u
#define MAX_FREQ ((1u << 16) - 17u)
#define TOP (1u << 24)

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

/* ---- range coder ------------------------------------------------------------------------------ */
typedef struct {
    uint32_t low, range, code, ffnum, carry, cache;
    uint8_t *out;                       /* encoder */
    const uint8_t *in, *in_end;         /* decoder */
    int overrun;
} rc_t;

static void rc_enc_start(rc_t *rc, uint8_t *out) { memset(rc, 0, sizeof *rc); rc->range = 0xffffffffu; rc->out = out; }
static void rc_shift_low(rc_t *rc)
{
    if (rc->low < 0xff000000u || rc->carry) {
        *rc->out++ = (uint8_t)(rc->cache + rc->carry);
        while (rc->ffnum) { *rc->out++ = (uint8_t)(rc->carry - 1); rc->ffnum--; }
        rc->cache = rc->low >> 24;
        rc->carry = 0;
    } else rc->ffnum++;
    rc->low <<= 8;
}
static void rc_encode(rc_t *rc, uint32_t cum, uint32_t freq, uint32_t tot)
{
    uint32_t old = rc->low;
    rc->range /= tot;
    rc->low += cum * rc->range;
    rc->range *= freq;
    if (rc->low < old) rc->carry = 1;
    while (rc->range < TOP) { rc->range <<= 8; rc_shift_low(rc); }
}
static uint8_t *rc_enc_finish(rc_t *rc) { for (int i = 0; i < 5; i++) rc_shift_low(rc); return rc->out; }

static void rc_dec_start(rc_t *rc, const uint8_t *in, const uint8_t *end)
{
    memset(rc, 0, sizeof *rc);
    rc->range = 0xffffffffu; rc->in = in; rc->in_end = end;
    for (int i = 0; i < 5; i++) { if (rc->in < end) rc->code = (rc->code << 8) | *rc->in++; else { rc->overrun = 1; rc->code <<= 8; } }
}
static uint32_t rc_get_freq(rc_t *rc, uint32_t tot) { rc->range /= tot; return rc->code / rc->range; }
static void rc_decode(rc_t *rc, uint32_t cum, uint32_t freq)
{
    rc->code -= cum * rc->range;
    rc->range *= freq;
    while (rc->range < TOP) {
        if (rc->in < rc->in_end) rc->code = (rc->code << 8) | *rc->in++; else { rc->overrun = 1; rc->code <<= 8; }
        rc->range <<= 8;
    }
}

/* ---- adaptive model ------------------------------------------------------------------------------ */
typedef struct { uint32_t tot, nsym; uint16_t F[256]; uint8_t S[256]; } model_t;

static void model_init(model_t *m, uint32_t nsym)
{
    m->tot = nsym; m->nsym = nsym;
    for (uint32_t i = 0; i < nsym; i++) { m->F[i] = 1; m->S[i] = (uint8_t)i; }
}
static void model_renorm(model_t *m)
{
    m->tot = 0;
    for (uint32_t i = 0; i < m->nsym; i++) { m->F[i] -= m->F[i] >> 1; m->tot += m->F[i]; }
}
static void model_update(model_t *m, uint32_t x)
{
    m->F[x] += STEP; m->tot += STEP;
    if (m->tot > MAX_FREQ) model_renorm(m);
    if (x > 0 && m->F[x] > m->F[x - 1]) {
        uint16_t f = m->F[x]; m->F[x] = m->F[x - 1]; m->F[x - 1] = f;
        uint8_t s = m->S[x]; m->S[x] = m->S[x - 1]; m->S[x - 1] = s;
    }
}
static void model_encode(model_t *m, rc_t *rc, uint32_t sym)
{
    uint32_t x = 0, acc = 0;
    while (m->S[x] != sym) acc += m->F[x++];
    rc_encode(rc, acc, m->F[x], m->tot);
    model_update(m, x);
}
static int model_decode(model_t *m, rc_t *rc)
{
    uint32_t freq = rc_get_freq(rc, m->tot), x = 0, acc = 0;
    if (freq >= m->tot) return -1;                       /* cannot happen in a valid stream */
    while (acc + m->F[x] <= freq) acc += m->F[x++];
    int sym = m->S[x];
    rc_decode(rc, acc, m->F[x]);
    model_update(m, x);
    return sym;
}

#endif

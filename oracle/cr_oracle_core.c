/*
 * oracle/cr_oracle_core.c — range coder, order-2 node, PPM model, u16 side model.
 * TEST INFRASTRUCTURE ONLY (see cr_oracle.h). Restated from the behaviour of the reference;
 * citations are /root/reference/src/<file>:<lines>.
 */
#include "cr_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ buffer */

void cro_buf_init(cro_buf* b) { b->data = NULL; b->size = 0; b->cap = 0; }
void cro_buf_free(cro_buf* b) { free(b->data); cro_buf_init(b); }
void cro_buf_clear(cro_buf* b) { b->size = 0; }

static void buf_need(cro_buf* b, uint32_t want) {
    if (want <= b->cap) return;
    uint32_t cap = b->cap ? b->cap : 256;
    while (cap < want) cap = cap + cap / 2 + 64;
    b->data = (uint8_t*)realloc(b->data, cap);
    b->cap = cap;
}
void cro_buf_put(cro_buf* b, uint8_t v) { buf_need(b, b->size + 1); b->data[b->size++] = v; }
void cro_buf_append(cro_buf* b, const uint8_t* p, uint32_t n) {
    buf_need(b, b->size + n);
    memcpy(b->data + b->size, p, n);
    b->size += n;
}
void cro_buf_resize(cro_buf* b, uint32_t n) { buf_need(b, n); b->size = n; }

/* ------------------------------------------------------------------ range coder */

#define RC_TOP    0x01000000u   /* cr-rangecoder.c:31 */
#define RC_NOCARRY 0xFF000000u  /* cr-rangecoder.c:32 "thresold" */

/* cr-rangecoder.c:34-41 */
void cro_rc_enc_init(cro_rc* rc) {
    rc->low = 0; rc->range = 0xFFFFFFFFu; rc->follow = 0; rc->carry = 0; rc->cache = 0; rc->end = NULL;
}

/* cr-rangecoder.c:44-58: shift one byte out of low; a byte is only released once it is known
 * whether a carry will still reach it (low < 0xFF000000 or the carry already happened). */
static void rc_shift(cro_rc* rc, cro_buf* out) {
    if (rc->low < RC_NOCARRY || rc->carry) {
        cro_buf_put(out, (uint8_t)(rc->cache + rc->carry));
        for (; rc->follow; rc->follow--) cro_buf_put(out, (uint8_t)(rc->carry - 1));
        rc->cache = rc->low >> 24;
        rc->carry = 0;
    } else {
        rc->follow++;
    }
    rc->low <<= 8;
}

/* cr-rangecoder.c:60-70 */
void cro_rc_enc_step(cro_rc* rc, uint32_t cum, uint32_t frq, uint32_t sum, cro_buf* out) {
    uint32_t unit = rc->range / sum;
    uint32_t moved = rc->low + cum * unit;
    rc->carry += moved < rc->low;
    rc->low = moved;
    rc->range = unit * frq;
    while (rc->range < RC_TOP) {
        rc->range <<= 8;
        rc_shift(rc, out);
    }
}

/* cr-rangecoder.c:72-79 */
void cro_rc_enc_flush(cro_rc* rc, cro_buf* out) {
    for (int i = 0; i < 5; i++) rc_shift(rc, out);
}

/* cr-rangecoder.c:81-89: five bytes through a 32-bit register, i.e. the first one falls out */
static uint32_t rc_next(const cro_rc* rc, const uint8_t** in) {
    /* The reference reads whatever follows its buffer when a damaged stream makes it consume more than was written; the
     * block decoders here stop at the end of the coded block and read zeros behind it (what the GPU decoders do). */
    if (rc->end && *in >= rc->end) return 0;
    return *(*in)++;
}

void cro_rc_dec_init_end(cro_rc* rc, const uint8_t** in, const uint8_t* end) {
    cro_rc_enc_init(rc);
    rc->end = end;
    for (int i = 0; i < 5; i++) rc->cache = (rc->cache << 8) + rc_next(rc, in);
}

void cro_rc_dec_init(cro_rc* rc, const uint8_t** in) { cro_rc_dec_init_end(rc, in, NULL); }

/* cr-rangecoder.c:101-104 */
uint32_t cro_rc_dec_target(cro_rc* rc, uint32_t sum) {
    /* A well-formed stream keeps the decoder inside the coded interval: cache < sum * (range / sum), because the encoder
     * only ever adds cum * unit with cum < sum. A damaged stream can leave it (the reference then indexes whatever its
     * search loops run into); that state is recorded in `carry`, which a decoder does not use otherwise, and the block
     * decoders report the block as corrupt. */
    if (sum == 0 || rc->range / sum == 0) { rc->carry = 1; rc->range = 1; return 0; }
    rc->range /= sum;
    const uint32_t t = rc->cache / rc->range;
    if (t >= sum) rc->carry = 1;
    return t;
}

int cro_rc_dec_left_interval(const cro_rc* rc) { return rc->carry != 0; }

/* cr-rangecoder.c:91-99 (the reference's `sum` argument is unused there) */
void cro_rc_dec_consume(cro_rc* rc, uint32_t cum, uint32_t frq, const uint8_t** in) {
    if (frq == 0) {      /* only a damaged stream selects a symbol nobody counted: the reference's loop below would never end */
        rc->carry = 1;
        rc->range = RC_TOP;
        return;
    }
    rc->cache -= cum * rc->range;
    rc->range *= frq;
    while (rc->range < RC_TOP) {
        rc->cache = (rc->cache << 8) + rc_next(rc, in);
        rc->range <<= 8;
    }
}

uint32_t cro_kat_rangecoder(const uint32_t* t, uint32_t n, uint8_t* out, uint32_t cap) {
    cro_rc rc; cro_buf b; cro_buf_init(&b);
    cro_rc_enc_init(&rc);
    for (uint32_t i = 0; i < n; i++) cro_rc_enc_step(&rc, t[3 * i], t[3 * i + 1], t[3 * i + 2], &b);
    cro_rc_enc_flush(&rc, &b);
    uint32_t w = b.size < cap ? b.size : cap;
    memcpy(out, b.data, w);
    uint32_t total = b.size;
    cro_buf_free(&b);
    return total;
}

/* ------------------------------------------------------------------ order-2 node */
/* Symbols 0..255 are bytes, 256 = "the o3 prediction was right", 257 = escape to order 1.
 * The reference caches nine 32-symbol group sums (cr-o2model.h:38-39); they are always equal to
 * the prefix sums of f[0..255] (cr-o2model.c:50-52,56-62), so this restatement recomputes them. */

#define SYM_HIT 256
#define SYM_ESC 257

/* cr-o2model.c:31-41 */
static void o2_fresh(cro_o2* nd) {
    memset(nd->f, 0, sizeof nd->f);
    nd->f[SYM_HIT] = 1;
    nd->f[SYM_ESC] = 1;
}

/* cr-o2model.c:75-84 (o2_model_cum for any symbol 0..257) */
static uint32_t o2_below(const cro_o2* nd, int sym) {
    uint32_t acc = 0;
    for (int i = 0; i < sym; i++) acc += nd->f[i];
    return acc;
}

/* cr-o2model.c:90-92 */
static uint32_t o2_total(const cro_o2* nd) { return o2_below(nd, 258); }

/* cr-o2model.c:43-73. Counts are u8 and wrap like the reference's `+=` on uint8_t; the halving
 * pass fires when the touched count exceeds 250 and returns 1. */
static int o2_bump(cro_o2* nd, int sym, int inc) {
    nd->f[sym] = (uint8_t)(nd->f[sym] + inc);
    if (nd->f[sym] <= 250) return 0;
    unsigned singles = 1;
    for (int i = 0; i < 256; i++) {
        nd->f[i] >>= 1;                       /* may drop to zero: the byte leaves the node */
        singles += nd->f[i] == 1;             /* PPMX-style escape estimate: 1 + #singletons */
    }
    nd->f[SYM_HIT] = (uint8_t)((nd->f[SYM_HIT] + 1) >> 1);
    nd->f[SYM_ESC] = (uint8_t)singles;
    return 1;
}

/* ------------------------------------------------------------------ PPM model */

#define O3_KEYS (1u << 22)

struct cro_ppm {
    uint8_t   o1[256][256];      /* cr-ppm.h:37 — count of byte j after byte i, weight 8c-7 */
    cro_o2*   o2[65536];         /* cr-ppm.h:38 — node per last-two-bytes context, on demand */
    uint8_t*  o3_byte;           /* cr-ppm.h:39 — predicted next byte per 22-bit key ...      */
    uint8_t*  o3_conf;           /*               ... and its 4-bit confidence (kept unpacked) */
    uint32_t  ctx;               /* cr-ppm.h:40 — last four bytes, newest in the low byte      */
    /* undo logs so that a reset costs O(touched), not 8 MiB: purely an oracle speed-up */
    uint32_t* o3_used; uint32_t o3_nused, o3_cap;
    uint16_t* o2_used; uint32_t o2_nused;
};

cro_ppm* cro_ppm_new(void) {
    cro_ppm* m = (cro_ppm*)calloc(1, sizeof *m);
    m->o3_byte = (uint8_t*)calloc(O3_KEYS, 1);
    m->o3_conf = (uint8_t*)calloc(O3_KEYS, 1);
    m->o3_cap = 1u << 16;
    m->o3_used = (uint32_t*)malloc(m->o3_cap * sizeof(uint32_t));
    m->o2_used = (uint16_t*)malloc(65536 * sizeof(uint16_t));
    memset(m->o1, 1, sizeof m->o1);
    return m;
}

/* cr-ppm.c:34-57 (ppm_model_free followed by ppm_model_init) */
void cro_ppm_reset(cro_ppm* m) {
    for (uint32_t i = 0; i < m->o2_nused; i++) { free(m->o2[m->o2_used[i]]); m->o2[m->o2_used[i]] = NULL; }
    m->o2_nused = 0;
    for (uint32_t i = 0; i < m->o3_nused; i++) { m->o3_byte[m->o3_used[i]] = 0; m->o3_conf[m->o3_used[i]] = 0; }
    m->o3_nused = 0;
    memset(m->o1, 1, sizeof m->o1);
    m->ctx = 0;
}

void cro_ppm_free(cro_ppm* m) {
    if (!m) return;
    cro_ppm_reset(m);
    free(m->o3_byte); free(m->o3_conf); free(m->o3_used); free(m->o2_used); free(m);
}

uint32_t cro_ppm_nodes(const cro_ppm* m) { return m->o2_nused; }

/* cr-ppm.c:60-64 */
void cro_ppm_push(cro_ppm* m, int byte) { m->ctx = (m->ctx << 8) | (uint32_t)(byte & 0xff); }

/* cr-ppm.c:66 M_predctx3_: 22-bit key folding the last ~3 bytes */
static uint32_t o3_key(uint32_t ctx) { return (ctx ^ (ctx >> 2)) & (O3_KEYS - 1); }

static void o3_touch(cro_ppm* m, uint32_t k) {
    if (m->o3_nused == m->o3_cap) {
        m->o3_cap *= 2;
        m->o3_used = (uint32_t*)realloc(m->o3_used, m->o3_cap * sizeof(uint32_t));
    }
    m->o3_used[m->o3_nused++] = k;
}

/* cr-ppm.c:69-88 with c < 0: the prediction was right, confidence saturates at 15 */
static void o3_hit(cro_ppm* m, uint32_t k) {
    if (!m->o3_conf[k] && !m->o3_byte[k]) o3_touch(m, k);
    if (m->o3_conf[k] < 15) m->o3_conf[k]++;
}

/* cr-ppm.c:69-88 with c >= 0: confidence decays 15..9→4, 8..5→3, 4..3→2, 2→1, 1..0→0, and a
 * prediction whose confidence hit zero is replaced by the byte just seen (confidence 1). */
static void o3_miss(cro_ppm* m, uint32_t k, int seen) {
    if (!m->o3_conf[k] && !m->o3_byte[k]) o3_touch(m, k);
    unsigned c = m->o3_conf[k];
    c = (c > 1) + (c > 2) + (c > 4) + (c > 8);
    if (c == 0) { m->o3_byte[k] = (uint8_t)seen; c = 1; }
    m->o3_conf[k] = (uint8_t)c;
}

/* cr-ppm.c:90-97 */
static void o1_bump(uint8_t* row, int sym) {
    if (++row[sym] >= 255)
        for (int i = 0; i < 256; i++) row[i] -= row[i] >> 1;
}
/* cr-ppm.c:98 M_freq_o1 */
static uint32_t o1_weight(const uint8_t* row, int i) { return (uint32_t)row[i] * 8u - 7u; }

/* cr-ppm.c:104-107,171-174 */
static cro_o2* o2_get(cro_ppm* m) {
    uint32_t key = m->ctx & 0xffff;
    if (!m->o2[key]) {
        m->o2[key] = (cro_o2*)malloc(sizeof(cro_o2));
        o2_fresh(m->o2[key]);
        m->o2_used[m->o2_nused++] = (uint16_t)key;
    }
    return m->o2[key];
}

/* cr-ppm.c:103-167 */
void cro_ppm_encode(cro_ppm* m, cro_rc* rc, int sym, cro_buf* out) {
    cro_o2*  nd   = o2_get(m);
    uint8_t* row  = m->o1[m->ctx & 0xff];
    uint32_t key  = o3_key(m->ctx);
    int      pred = m->o3_byte[key];
    uint32_t pf   = nd->f[pred];               /* the predicted byte is priced as symbol 256, */
    uint32_t tot  = o2_total(nd) - pf;         /* so its own count is taken out of the node   */

    if (sym == pred) {                                                   /* cr-ppm.c:119-126 */
        cro_rc_enc_step(rc, o2_below(nd, SYM_HIT) - pf, nd->f[SYM_HIT], tot, out);
        o2_bump(nd, SYM_HIT, 1);
        o3_hit(m, key);
        return;
    }
    if (nd->f[sym]) {                                                    /* cr-ppm.c:129-139 */
        cro_rc_enc_step(rc, o2_below(nd, sym) - (sym > pred ? pf : 0), nd->f[sym], tot, out);
        int halved = o2_bump(nd, sym, 1);
        if (!halved && nd->f[sym] == 2) o2_bump(nd, SYM_ESC, -1);        /* no longer a singleton */
    } else {                                                             /* cr-ppm.c:141-163 */
        cro_rc_enc_step(rc, o2_below(nd, SYM_ESC) - pf, nd->f[SYM_ESC], tot, out);
        int halved = o2_bump(nd, SYM_ESC, 1);
        if (row[sym] > 0) {
            /* order 1 with exclusion of the prediction and of every byte the node knows NOW,
             * i.e. after the escape update above (which may have halved counts to zero) */
            uint32_t lo = 0, all = 0;
            for (int i = 0; i < 256; i++) {
                if (nd->f[i] || i == pred) continue;
                uint32_t w = o1_weight(row, i);
                if (i < sym) lo += w;
                all += w;
            }
            cro_rc_enc_step(rc, lo, o1_weight(row, sym), all, out);
            o1_bump(row, sym);
        }
        if (!halved) o2_bump(nd, sym, 1);
    }
    o3_miss(m, key, sym);
}

/* cr-o2model.c:94-113: the symbol whose interval [below, below+f) holds `target`, with the
 * predicted byte's count treated as zero. */
static int o2_find(const cro_o2* nd, uint32_t target, int pred, uint32_t* below) {
    uint32_t acc = 0;
    for (int s = 0; s < 258; s++) {
        uint32_t f = (s == pred) ? 0 : nd->f[s];
        if (acc + f > target) { *below = acc; return s; }
        acc += f;
    }
    *below = acc;
    return 257; /* unreachable on a well-formed stream */
}

/* cr-ppm.c:169-235 */
int cro_ppm_decode(cro_ppm* m, cro_rc* rc, const uint8_t** in) {
    cro_o2*  nd   = o2_get(m);
    uint8_t* row  = m->o1[m->ctx & 0xff];
    uint32_t key  = o3_key(m->ctx);
    int      pred = m->o3_byte[key];
    uint32_t pf   = nd->f[pred];

    uint32_t target = cro_rc_dec_target(rc, o2_total(nd) - pf);
    uint32_t below;
    int s = o2_find(nd, target, pred, &below);
    cro_rc_dec_consume(rc, below, nd->f[s], in);
    int halved = o2_bump(nd, s, 1);

    if (s == SYM_HIT) {                                                  /* cr-ppm.c:199-201 */
        o3_hit(m, key);
        return pred;
    }
    if (s < 256) {                                                       /* cr-ppm.c:203-207 */
        if (!halved && nd->f[s] == 2) o2_bump(nd, SYM_ESC, -1);
        o3_miss(m, key, s);
        return s;
    }
    /* escape: cr-ppm.c:209-232 */
    uint32_t all = 0;
    for (int i = 0; i < 256; i++)
        if (!nd->f[i] && i != pred) all += o1_weight(row, i);
    target = cro_rc_dec_target(rc, all);
    uint32_t acc = 0;
    int got = SYM_ESC;                        /* the reference leaves 257 here if nothing fits */
    for (int i = 0; i < 256; i++) {
        if (nd->f[i] || i == pred) continue;
        uint32_t w = o1_weight(row, i);
        if (acc + w > target) { got = i; break; }
        acc += w;
    }
    if (got == SYM_ESC) got = 0;              /* corrupt stream: stay inside the tables */
    cro_rc_dec_consume(rc, acc, o1_weight(row, got), in);
    o1_bump(row, got);
    if (!halved) o2_bump(nd, got, 1);
    o3_miss(m, key, got);
    return got;
}

/* ------------------------------------------------------------------ raw PPM harness */

uint32_t cro_ppm_encode_raw(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap, uint32_t* preflush) {
    cro_ppm* m = cro_ppm_new();
    cro_rc rc; cro_buf b; cro_buf_init(&b);
    cro_rc_enc_init(&rc);
    for (uint32_t i = 0; i < n; i++) {
        cro_ppm_encode(m, &rc, in[i], &b);
        cro_ppm_push(m, in[i]);
    }
    if (preflush) *preflush = b.size;
    cro_rc_enc_flush(&rc, &b);
    uint32_t total = b.size;
    memcpy(out, b.data, total < cap ? total : cap);
    cro_buf_free(&b);
    cro_ppm_free(m);
    return total;
}

uint32_t cro_ppm_decode_raw(const uint8_t* in, uint32_t n_in, uint8_t* out, uint32_t n_out) {
    (void)n_in;
    cro_ppm* m = cro_ppm_new();
    cro_rc rc;
    const uint8_t* p = in;
    cro_rc_dec_init(&rc, &p);
    for (uint32_t i = 0; i < n_out; i++) {
        out[i] = (uint8_t)cro_ppm_decode(m, &rc, &p);
        cro_ppm_push(m, out[i]);
    }
    cro_ppm_free(m);
    return n_out;
}

/* ------------------------------------------------------------------ u16 side model */
/* cr-model.c: 256 u16 counts; the reference's nine group sums are again plain prefix sums. */

/* cr-model.c:44-54 */
void cro_model_recount(cro_model* m) {
    uint32_t t = 0;
    for (int i = 0; i < 256; i++) t += m->f[i];
    m->total = t;
}
/* cr-model.c:33-42 */
void cro_model_init_flat(cro_model* m) {
    for (int i = 0; i < 256; i++) m->f[i] = 1;
    m->total = 256;
}
/* cr-model.c:56-78: add, then halve (rounding up) once the total passes 32000 */
static void model_bump(cro_model* m, int sym, int inc) {
    m->f[sym] = (uint16_t)(m->f[sym] + inc);
    m->total = (uint16_t)(m->total + inc);    /* reference total is a u16 cell (cr-model.h:44) */
    if (m->total > 32000) {
        for (int i = 0; i < 256; i++) m->f[i] = (uint16_t)((m->f[i] + 1) >> 1);
        cro_model_recount(m);
    }
}
static uint32_t model_below(const cro_model* m, int sym) {
    uint32_t a = 0;
    for (int i = 0; i < sym; i++) a += m->f[i];
    return a;
}
/* cr-model.h:58-64 M_my_enc_ */
void cro_model_encode(cro_model* m, cro_rc* rc, int sym, int inc, cro_buf* out) {
    cro_rc_enc_step(rc, model_below(m, sym), m->f[sym], m->total, out);
    if (inc) model_bump(m, sym, inc);
}
/* cr-model.h:66-74 M_my_dec_ with cr-model.c:98-115 */
int cro_model_decode(cro_model* m, cro_rc* rc, int inc, const uint8_t** in) {
    uint32_t target = cro_rc_dec_target(rc, m->total);
    uint32_t acc = 0;
    int s = 0;
    while (s < 255 && acc + m->f[s] <= target) acc += m->f[s++];
    cro_rc_dec_consume(rc, acc, m->f[s], in);
    if (inc) model_bump(m, s, inc);
    return s;
}

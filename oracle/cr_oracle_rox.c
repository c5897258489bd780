/*
 * oracle/cr_oracle_rox.c — comprox block codec: LZ77 parse (hash chains, lazy evaluation, repeat
 * match, short-distance cache) + PPM main stream + three side streams coded with u16 models.
 * TEST INFRASTRUCTURE ONLY (see cr_oracle.h). Restated from the behaviour of
 * /root/reference/src/roxmain/cr-coder.c and cr-matcher.c; citations are to those files.
 * Both parsers: the default lazy one and the -f "flexible parsing" one (cr-matcher.c:253-289).
 *
 * Block layout (cr-coder.c:69-81, sizeof == 32): [0] coded flag, [1] match_min, [2] esc, [3] pad,
 * then u32 LE: original size, #short-distance codes, #distance codes, #length codes, offsets of the
 * short-distance / distance / length streams; body = main PPM stream, then those three streams.
 * Stored form: 32 zero bytes + the raw input.
 */
#include "cr_oracle.h"
#include <stdlib.h>
#include <string.h>

#define ROX_NEAR_MIN   6u            /* match_min_near, cr-matcher.c:36 */
#define ROX_MAX        255u          /* match_max, cr-matcher.c:38 */
#define ROX_TAIL       1024u         /* cr-coder.c:136 */
#define ROX_NONE       0xFFFFFFFFu

typedef struct rox_match { uint32_t pos, len; } rox_match;

struct cro_rox {
    cro_ppm*  ppm;
    cro_model len_model, pos_model[6], spos_model;       /* cr-coder.c:55-60 */
    uint32_t  chain_limit;                               /* match_limit, cr-matcher.c:39 (-m switch) */
    /* matcher state for the block being parsed */
    uint32_t* prev;                                      /* m_next: earlier position of the same hash class */
    uint32_t  prev_cap;
    uint32_t  near[65536];                               /* m_short_cache */
    uint32_t  repeat;                                    /* m_last_match (a distance) */
    uint32_t  long_min;                                  /* match_min: 10, or 11 above 16 MiB */
    int       flexible;                                  /* flexible_parsing, cr-matcher.c:32 (-f switch) */
};

cro_rox* cro_rox_new(void) {
    cro_rox* c = (cro_rox*)calloc(1, sizeof *c);
    c->ppm = cro_ppm_new();
    c->chain_limit = 40;
    cro_rox_reset(c);
    return c;
}
void cro_rox_free(cro_rox* c) { if (c) { cro_ppm_free(c->ppm); free(c->prev); free(c); } }
void cro_rox_set_chain_limit(cro_rox* c, uint32_t limit) { c->chain_limit = limit; }
void cro_rox_set_flexible(cro_rox* c, int on) { c->flexible = on != 0; }

/* reset_models, cr-coder.c:88-114 */
void cro_rox_reset(cro_rox* c) {
    cro_ppm_reset(c->ppm);
    for (int k = 0; k < 256; k++) {
        c->pos_model[0].f[k] = (k % 8 == 0);             /* distances are coded times 8 */
        c->pos_model[1].f[k] = 1;
        c->pos_model[2].f[k] = c->pos_model[3].f[k] = c->pos_model[4].f[k] = (k < 128);
        c->len_model.f[k] = (k == 0) || (k >= (int)ROX_NEAR_MIN);
    }
    for (int i = 0; i < 5; i++) cro_model_recount(&c->pos_model[i]);
    cro_model_recount(&c->len_model);
    cro_model_init_flat(&c->pos_model[5]);
    cro_model_init_flat(&c->spos_model);
}

/* ------------------------------------------------------------------ matcher */

static uint32_t mix_bytes(const uint8_t* s, uint32_t n) {       /* cr-matcher.c:45-53,203-211 */
    uint32_t h = 0;
    for (uint32_t i = 0; i < n; i++) h = (h * 123456791u) ^ s[i];
    return h;
}

/* matcher_init, cr-matcher.c:89-148. The two bucket passes leave, for every position p with
 * p + 255 < len, the largest earlier position with the same (s[0]+s[1]) % 20 and the same
 * hash-of-match_min-bytes % (20 + len/25); positions closer to the end are never linked. */
static void chains_build(cro_rox* c, const uint8_t* d, uint32_t len) {
    if (len > c->prev_cap) { free(c->prev); c->prev = (uint32_t*)malloc((size_t)len * 4); c->prev_cap = len; }
    for (uint32_t i = 0; i < len; i++) c->prev[i] = ROX_NONE;
    memset(c->near, 0, sizeof c->near);
    c->repeat = 0;
    const uint32_t classes = 20u + len / 25u;
    uint32_t* last = (uint32_t*)malloc((size_t)20 * classes * 4);
    for (size_t i = 0; i < (size_t)20 * classes; i++) last[i] = ROX_NONE;
    for (uint32_t p = 0; p + ROX_MAX < len; p++) {
        size_t cls = (size_t)((d[p] + d[p + 1]) % 20u) * classes + mix_bytes(d + p, c->long_min) % classes;
        c->prev[p] = last[cls];
        last[cls] = p;
    }
    free(last);
}

/* match(), cr-matcher.c:156-201 */
static rox_match chain_search(const cro_rox* c, const uint8_t* d, uint32_t pos, uint32_t want, uint32_t budget, uint32_t eager) {
    rox_match best = {0, want - 1};
    uint32_t at = c->prev[pos];
    for (uint32_t i = 0; i < budget && at != ROX_NONE; i++, at = c->prev[at]) {
        uint32_t len = best.len;
        while (len < ROX_MAX && d[at + len] == d[pos + len]) len++;
        /* a farther candidate has to be longer by up to 3 to displace the current one */
        uint32_t far = pos - at, cur = pos - best.pos, toll = 0;
        toll += far / 1048576u > cur;
        toll += far / 4096u > cur;
        toll += far / 64u > cur;
        if (len > best.len + toll && memcmp(d + pos, d + at, best.len) == 0) {
            best.pos = at; best.len = len;
            if ((eager && eager < best.pos) || best.len == ROX_MAX) return best;
        }
    }
    if (best.len < want) { best.pos = ROX_NONE; best.len = 1; }
    return best;
}

static uint32_t same_run(const uint8_t* d, uint32_t a, uint32_t b) {
    uint32_t n = 0;
    while (n < ROX_MAX && d[a + n] == d[b + n]) n++;
    return n;
}

/* fast_log2, cr-matcher.c:218-235: floor(log2(x)) for x >= 1 */
static int ilog2(uint32_t x) { int l = -1; while (x) { l++; x >>= 1; } return l; }
/* M_price, cr-matcher.c:269-271: 3 per matched byte beyond the first minus 4/5 log2(distance); 9 for a literal */
static uint32_t flex_price(const cro_rox* c, uint32_t pos, uint32_t from, uint32_t len) {
    return len >= c->long_min ? (len - 1u) * 3u - (uint32_t)(ilog2(pos - from) * 4 / 5) : 9u;
}
static uint32_t flex_price_at(const cro_rox* c, const uint8_t* d, uint32_t pos, uint32_t at) {
    const rox_match r = chain_search(c, d, at, c->long_min, c->chain_limit, 0);
    return flex_price(c, pos, r.pos, r.len);
}

/* matcher_lookup, cr-matcher.c:237-340 */
static rox_match parse_at(cro_rox* c, const uint8_t* d, uint32_t pos) {
    const uint32_t lim = c->chain_limit;
    rox_match rep = {pos - c->repeat, 0};                /* cr-matcher.c:246-251: the previous distance again */
    if (rep.pos < pos) rep.len = same_run(d, pos, rep.pos);

    rox_match m = chain_search(c, d, pos, c->long_min, lim, 0);
    if (c->flexible) {
        /* cr-matcher.c:253-289: cut the match at the length that leaves the best-priced pair "this match,
         * then whatever starts right behind it". (The reference memoises match(); it is a pure function of
         * the position. Its price macro measures BOTH distances from the current position.) */
        if (m.len >= c->long_min) {
            uint32_t best = flex_price(c, pos, m.pos, m.len) + flex_price_at(c, d, pos, pos + m.len);
            const uint32_t whole = m.len;
            for (uint32_t i = whole - 1; i >= 1; i--) {
                const uint32_t v = flex_price(c, pos, m.pos, i) + flex_price_at(c, d, pos, pos + i);
                if (best < v) { m.len = i; best = v; }
            }
            if (m.len < c->long_min) { m.pos = ROX_NONE; m.len = 1; }
        }
    } else if (m.len >= c->long_min) {                   /* cr-matcher.c:292-310: would waiting pay off? */
        rox_match n1 = chain_search(c, d, pos + 1, m.len + 1, lim / 4, 1);
        int defer = n1.len > m.len + (n1.pos < m.pos)
                 || chain_search(c, d, pos + 2, m.len + 1, lim / 8, 1).len > 1
                 || chain_search(c, d, pos + 3, m.len + 2, lim / 8, 1).len > 1
                 || chain_search(c, d, pos + 4, m.len + 2, lim / 8, 1).len > 1
                 || chain_search(c, d, pos + 5, m.len + 2, lim / 8, 1).len > 1
                 || chain_search(c, d, pos + 6, m.len + 3, lim / 8, 1).len > 1;
        if (defer) { m.pos = ROX_NONE; m.len = 1; }
    }
    if (m.pos != ROX_NONE &&                             /* cr-matcher.c:312-317: a repeat saves the distance */
        m.len < rep.len + 3 + (m.pos + 64u < pos) + (m.pos + 4096u < pos) + (m.pos + 1048576u < pos))
        m = rep;
    if (m.len < ROX_NEAR_MIN) {                          /* cr-matcher.c:319-331: short, nearby match */
        m.pos = c->near[mix_bytes(d + pos, ROX_NEAR_MIN) % 65536u];
        m.len = 0;
        if (m.pos < pos && m.pos + 256u > pos) m.len = same_run(d, m.pos, pos);
    }
    if (m.len < ROX_NEAR_MIN || (m.len < c->long_min && m.pos + 256u <= pos)) {     /* cr-matcher.c:333-338 */
        m.pos = ROX_NONE; m.len = 1;
    } else {
        c->repeat = pos - m.pos;
    }
    return m;
}

/* lzmatch_thread, cr-coder.c:126-151, for the whole block */
uint32_t cro_rox_parse(cro_rox* c, const uint8_t* in, uint32_t n, uint32_t* pos_out, uint32_t* len_out) {
    c->long_min = 10u + (n > 16777216u);                 /* cr-coder.c:192 */
    chains_build(c, in, n);
    uint32_t pos = 0, nt = 0;
    while (pos < n) {
        rox_match m = {ROX_NONE, 1};
        if (pos + ROX_TAIL < n) {
            m = parse_at(c, in, pos);
            for (uint32_t i = 0; i < m.len; i++)          /* matcher_update_cache, cr-matcher.c:213-216 */
                c->near[mix_bytes(in + pos + i, ROX_NEAR_MIN) % 65536u] = pos + i;
        }
        pos_out[nt] = m.pos; len_out[nt] = m.len; nt++;
        pos += m.len;
    }
    return nt;
}

/* ------------------------------------------------------------------ coder */

static void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

static uint32_t rox_stored(const uint8_t* in, uint32_t n, uint8_t* out) {          /* cr-coder.c:309-313 */
    memset(out, 0, CRO_ROX_HEADER);
    memcpy(out + CRO_ROX_HEADER, in, n);
    return CRO_ROX_HEADER + n;
}

/* lzencode, cr-coder.c:153-318 */
uint32_t cro_rox_encode(cro_rox* c, const uint8_t* in, uint32_t n, uint8_t* out) {
    uint32_t hist[256] = {0};
    for (uint32_t i = 0; i < n; i++) hist[in[i]]++;
    int esc = 0;
    for (int v = 1; v < 256; v++) if (hist[v] < hist[esc]) esc = v;       /* cr-coder.c:181-189 */

    uint32_t* mpos = (uint32_t*)malloc((size_t)(n + 1) * 4);
    uint32_t* mlen = (uint32_t*)malloc((size_t)(n + 1) * 4);
    cro_rox_parse(c, in, n, mpos, mlen);

    cro_buf main_s, spos_s, pos_s, len_s;
    cro_buf_init(&main_s); cro_buf_init(&spos_s); cro_buf_init(&pos_s); cro_buf_init(&len_s);
    cro_rc rc_main, rc_spos, rc_pos, rc_len;
    cro_rc_enc_init(&rc_main); cro_rc_enc_init(&rc_spos); cro_rc_enc_init(&rc_pos); cro_rc_enc_init(&rc_len);
    uint32_t n_spos = 0, n_pos = 0, n_len = 0, pos = 0, t = 0, prev_dist = 0;
    int stored = 0;
    while (pos < n) {                                                    /* cr-coder.c:213-276 */
        uint32_t from = mpos[t], len = mlen[t]; t++;
        if (from != ROX_NONE) {
            cro_ppm_encode(c->ppm, &rc_main, esc, &main_s);
            uint32_t dist = pos - from;
            if (dist == prev_dist) dist = 0;                             /* cr-coder.c:232-234: "same as last" */
            cro_model_encode(&c->len_model, &rc_len, (int)len, 30, &len_s); n_len++;
            if (len < c->long_min) {
                cro_model_encode(&c->spos_model, &rc_spos, (int)dist, 1, &spos_s); n_spos++;
            } else {                                                     /* cr-coder.c:243-258: dist*8 in digits */
                uint32_t j = dist * 8u; int i = 0;
                while (j >= 128 && i < 2) { cro_model_encode(&c->pos_model[i], &rc_pos, (int)(j % 128 + 128), 1 << (2 * i), &pos_s); i++; j /= 128; }
                if (i >= 2) while (j >= 64 && i < 5) { cro_model_encode(&c->pos_model[i], &rc_pos, (int)(j % 64 + 64), 1 << (2 * i), &pos_s); i++; j /= 64; }
                cro_model_encode(&c->pos_model[i], &rc_pos, (int)j, 1 << (2 * i), &pos_s);
                n_pos++;
            }
            prev_dist = dist;                                            /* cr-coder.c:260 (0 after a repeat) */
        } else {
            cro_ppm_encode(c->ppm, &rc_main, in[pos], &main_s);
            if (in[pos] == esc) { cro_model_encode(&c->len_model, &rc_len, 0, 30, &len_s); n_len++; }
        }
        for (uint32_t i = 0; i < len; i++) cro_ppm_push(c->ppm, in[pos++]);
        if (CRO_ROX_HEADER + main_s.size >= n) { stored = 1; break; }    /* cr-coder.c:273-275 */
    }
    free(mpos); free(mlen);
    uint32_t total;
    if (stored) {
        total = rox_stored(in, n, out);
    } else {
        cro_rc_enc_flush(&rc_main, &main_s); cro_rc_enc_flush(&rc_spos, &spos_s);
        cro_rc_enc_flush(&rc_pos, &pos_s); cro_rc_enc_flush(&rc_len, &len_s);
        memset(out, 0, CRO_ROX_HEADER);
        out[0] = 1; out[1] = (uint8_t)c->long_min; out[2] = (uint8_t)esc;
        put32(out + 4, n); put32(out + 8, n_spos); put32(out + 12, n_pos); put32(out + 16, n_len);
        uint32_t o = CRO_ROX_HEADER + main_s.size;
        put32(out + 20, o); memcpy(out + CRO_ROX_HEADER, main_s.data, main_s.size);
        memcpy(out + o, spos_s.data, spos_s.size); o += spos_s.size;
        put32(out + 24, o); memcpy(out + o, pos_s.data, pos_s.size); o += pos_s.size;
        put32(out + 28, o); memcpy(out + o, len_s.data, len_s.size); o += len_s.size;
        total = o;
    }
    cro_buf_free(&main_s); cro_buf_free(&spos_s); cro_buf_free(&pos_s); cro_buf_free(&len_s);
    return total;
}

/* lzdecode, cr-coder.c:390-526; the reference pre-decodes the side streams into queues, which is
 * the same as decoding each stream on demand because every stream is sequential on its own */
uint32_t cro_rox_decode(cro_rox* c, const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap) {
    if (n < CRO_ROX_HEADER) return 0xFFFFFFFFu;
    if (!in[0]) {
        uint32_t raw = n - CRO_ROX_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        memcpy(out, in + CRO_ROX_HEADER, raw);
        return raw;
    }
    const uint32_t long_min = in[1], total = get32(in + 4);
    const int esc = in[2];
    if (total > cap) return 0xFFFFFFFFu;
    {   /* a header whose stream offsets do not lie inside the block in order is corrupt (the reference trusts them and
         * reads wherever they point) */
        const uint32_t o_spos = get32(in + 20), o_pos = get32(in + 24), o_len = get32(in + 28);
        if (o_spos < CRO_ROX_HEADER || o_spos > o_pos || o_pos > o_len || o_len > n) return 0xFFFFFFFFu;
    }
    const uint8_t *s_main = in + CRO_ROX_HEADER, *s_spos = in + get32(in + 20), *s_pos = in + get32(in + 24), *s_len = in + get32(in + 28);
    cro_rc rc_main, rc_spos, rc_pos, rc_len;
    cro_rc_dec_init_end(&rc_main, &s_main, in + n); cro_rc_dec_init_end(&rc_spos, &s_spos, in + n);
    cro_rc_dec_init_end(&rc_pos, &s_pos, in + n); cro_rc_dec_init_end(&rc_len, &s_len, in + n);
    uint32_t have = 0, prev_dist = 0;
    while (have < total) {                                               /* cr-coder.c:459-523 */
        uint32_t len = 1, from = 0;
        int s = cro_ppm_decode(c->ppm, &rc_main, &s_main);
        int lit = s;
        if (s == esc) {
            len = (uint32_t)cro_model_decode(&c->len_model, &rc_len, 30, &s_len);
            if (len == 0) {
                len = 1; lit = esc;
            } else {
                uint32_t dist;
                if (len < long_min) {
                    dist = (uint32_t)cro_model_decode(&c->spos_model, &rc_spos, 1, &s_spos);
                } else {                                                 /* cr-coder.c:347-368 */
                    uint32_t v = 0, sym = 0; int j = 0;
                    while (j < 2 && (sym = (uint32_t)cro_model_decode(&c->pos_model[j], &rc_pos, 1 << (2 * j), &s_pos)) >= 128) { v += (sym - 128) << (7 * j); j++; }
                    if (j < 2) {
                        dist = (v + (sym << (7 * j))) / 8;
                    } else {
                        while (j < 5 && (sym = (uint32_t)cro_model_decode(&c->pos_model[j], &rc_pos, 1 << (2 * j), &s_pos)) >= 64) { v += (sym - 64) << (6 * j + 2); j++; }
                        dist = (v + (sym << (6 * j + 2))) / 8;
                    }
                }
                if (len > 1) {                                           /* cr-coder.c:503-506 */
                    uint32_t d = dist > 0 ? dist : prev_dist;
                    if (d == 0 || d > have || have + len > cap || have + len > total) return 0xFFFFFFFFu;
                    from = have - d;
                    prev_dist = d;
                }
            }
        }
        if (len > 1) { for (uint32_t i = 0; i < len; i++) out[have + i] = out[from + i]; }
        else { if (have >= cap) return 0xFFFFFFFFu; out[have] = (uint8_t)lit; }
        for (uint32_t i = 0; i < len; i++) cro_ppm_push(c->ppm, out[have + i]);
        have += len;
    }
    if (cro_rc_dec_left_interval(&rc_main) || cro_rc_dec_left_interval(&rc_spos) || cro_rc_dec_left_interval(&rc_pos) ||
        cro_rc_dec_left_interval(&rc_len)) return 0xFFFFFFFFu;           /* corrupt stream (cr_oracle_core.c, cro_rc_dec_target) */
    return have;
}

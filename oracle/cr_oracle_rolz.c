/*
 * oracle/cr_oracle_rolz.c — comprolz block codec: ROLZ parse (reduced-offset LZ: a match is named by its
 * rank among the most recent positions that followed the same hashed context) + PPM main stream + one
 * side stream (lengths, ranks) coded with two u16 models.
 * TEST INFRASTRUCTURE ONLY (see cr_oracle.h). Restated from the behaviour of
 * /root/reference/src/rolzmain/cr-coder.c and cr-matcher.c; citations are to those files.
 * Both parsers: the default lazy one and the -f "flexible parsing" one (cr-matcher.c:143-167).
 *
 * Block layout (cr-coder.c:63-71, sizeof == 16): [0] first byte of the block, [1] coded flag, [2] esc,
 * [3] pad, then u32 LE: original size, number of side-stream codes, offset of the side stream;
 * body = main PPM stream, then the side stream. Stored form: 16 zero bytes + the raw input.
 *
 * The reference keeps 262 144 rings of 64 positions (+ a hash byte each) and a 256 x 16 table of the
 * positions that followed each byte value, 85 MB re-initialised per block (cr-matcher.h:43-52,
 * cr-matcher.c:45-59). A block of n bytes feeds every position once, so the same information is two
 * link arrays here: the previous position fed to the same ring / the same row. Rank i of a ring is the
 * i-th link from its newest entry; ranks past the end of a ring read -1 (the reference's memset), ranks past
 * the end of a row read position 0 (its zero fill) — both kept.
 */
#include "cr_oracle.h"
#include <stdlib.h>
#include <string.h>

#define ROLZ_BUCKETS 262144u          /* M_rolz_buckets */
#define ROLZ_RING    64u              /* M_rolz_indices */
#define ROLZ_ROW     16u              /* M_rolz_indices_short */
#define ROLZ_MIN     5u               /* M_rolz_minlength */
#define ROLZ_MAX     255u             /* M_rolz_maxlength */
#define ROLZ_TAIL    1024u            /* cr-coder.c:121 */
#define ROLZ_WARM    16u              /* cr-matcher.c:68,148: nothing is fed or looked up below position 16 */
#define ROLZ_NONE    0xFFFFFFFFu

struct cro_rolz {
    cro_ppm*  ppm;
    cro_model idx_model, len_model;                       /* cr-coder.c:52-56 */
    /* matcher state for the block in hand */
    uint32_t* ring_prev;                                  /* previous position fed to the same ring */
    uint32_t* row_prev;                                   /* previous position fed to the same row */
    uint32_t  cap;
    uint32_t* ring_head;                                  /* newest position of each ring, ROLZ_NONE = empty */
    uint32_t  row_head[256];                              /* newest position of each row, ROLZ_NONE = empty */
    uint32_t  ring_now;                                   /* m_context */
    uint32_t  row_now;                                    /* m_short_context */
    int       ctx4;                                       /* using_ctx4: blocks of 4 MiB and more hash four bytes */
    int       flexible;                                   /* flexible_parsing, cr-matcher.c:30 (-f switch) */
};

cro_rolz* cro_rolz_new(void) {
    cro_rolz* c = (cro_rolz*)calloc(1, sizeof *c);
    c->ppm = cro_ppm_new();
    c->ring_head = (uint32_t*)malloc(ROLZ_BUCKETS * sizeof(uint32_t));
    cro_rolz_reset(c);
    return c;
}
void cro_rolz_free(cro_rolz* c) { if (c) { cro_ppm_free(c->ppm); free(c->ring_prev); free(c->row_prev); free(c->ring_head); free(c); } }

void cro_rolz_set_flexible(cro_rolz* c, int on) { c->flexible = on != 0; }

/* reset_models, cr-coder.c:78-97 */
void cro_rolz_reset(cro_rolz* c) {
    cro_ppm_reset(c->ppm);
    for (int k = 0; k < 256; k++) {
        c->idx_model.f[k] = (k < (int)(ROLZ_RING + ROLZ_ROW));
        c->len_model.f[k] = (k == 0) || (k >= (int)ROLZ_MIN && k <= (int)ROLZ_MAX);
    }
    cro_model_recount(&c->idx_model);
    cro_model_recount(&c->len_model);
}

/* M_rolz_hash_ctx, cr-matcher.c:37-41: the three (four) bytes ending at x */
static uint32_t ring_of(const cro_rolz* c, const uint8_t* x) {
    uint32_t h = (uint32_t)x[0] * 1313131u + (uint32_t)x[-1] * 13131u + (uint32_t)x[-2] * 131u;
    if (c->ctx4) h += x[-3];
    return h % ROLZ_BUCKETS;
}

/* matcher_init, cr-matcher.c:43-59 */
static void matcher_start(cro_rolz* c, uint32_t n) {
    if (n + 1u > c->cap) {
        c->cap = n + 1u;
        c->ring_prev = (uint32_t*)realloc(c->ring_prev, c->cap * sizeof(uint32_t));
        c->row_prev = (uint32_t*)realloc(c->row_prev, c->cap * sizeof(uint32_t));
    }
    memset(c->ring_head, 0xff, ROLZ_BUCKETS * sizeof(uint32_t));
    memset(c->row_head, 0xff, sizeof c->row_head);
    c->ring_now = 0;
    c->row_now = 0;
    c->ctx4 = n >= 4194304u;
}

/* matcher_update, cr-matcher.c:66-84: position `pos` joins the ring of the context in front of it and
 * the row of the byte in front of it; both then move on to the context / byte ending at pos.
 * (Nothing happens below position 16, so position 16 itself joins ring 0 and row 0: kept.) */
static void matcher_feed(cro_rolz* c, const uint8_t* d, uint32_t pos) {
    if (pos < ROLZ_WARM) return;
    c->ring_prev[pos] = c->ring_head[c->ring_now];
    c->ring_head[c->ring_now] = pos;
    c->ring_now = ring_of(c, d + pos);
    c->row_prev[pos] = c->row_head[c->row_now];
    c->row_head[c->row_now] = pos;
    c->row_now = d[pos];
}

/* matcher_getpos, cr-matcher.c:86-91 */
static uint32_t matcher_at(const cro_rolz* c, uint32_t rank) {
    if (rank < ROLZ_RING) {
        uint32_t p = c->ring_head[c->ring_now];
        while (rank-- && p != ROLZ_NONE) p = c->ring_prev[p];
        return p;
    }
    rank -= ROLZ_RING;
    uint32_t p = c->row_head[c->row_now];
    while (rank-- && p != ROLZ_NONE) p = c->row_prev[p];
    return p == ROLZ_NONE ? 0u : p;                        /* the reference's rows are zero-filled */
}

typedef struct { uint32_t rank, len; } rolz_hit;

/* match(), cr-matcher.c:93-124: the ring `ring`, seen as it stood before position `floor` was fed
 * (lazy evaluation looks ahead without feeding), newest first; first strictly longer agreement wins */
static rolz_hit ring_search(const cro_rolz* c, const uint8_t* d, uint32_t pos, uint32_t ring, uint32_t floor) {
    rolz_hit best = { ROLZ_NONE, ROLZ_MIN - 1u };
    uint32_t p = c->ring_head[ring];
    while (p != ROLZ_NONE && p >= floor) p = c->ring_prev[p];
    for (uint32_t i = 0; i < ROLZ_RING && best.len < ROLZ_MAX && p != ROLZ_NONE; i++, p = c->ring_prev[p]) {
        if (d[p] != d[pos]) continue;                      /* the ring's hash byte is the position's first byte */
        uint32_t j = 0;
        while (j < ROLZ_MAX && d[pos + j] == d[p + j]) j++;
        if (j > best.len) { best.rank = i; best.len = j; }
    }
    if (best.len < ROLZ_MIN) { best.rank = ROLZ_NONE; best.len = 1; }
    return best;
}

/* M_price, cr-matcher.c:150-152 */
static uint32_t price(rolz_hit h) {
    return h.len >= ROLZ_MIN ? (h.len - 1u) * 3u * ROLZ_RING - 3u * h.rank : 9u * ROLZ_RING;
}

/* matcher_lookup, cr-matcher.c:126-197 */
static rolz_hit matcher_find(const cro_rolz* c, const uint8_t* d, uint32_t pos) {
    rolz_hit r = { ROLZ_NONE, 1 };
    if (pos < ROLZ_WARM) return r;
    r = ring_search(c, d, pos, c->ring_now, pos);
    const int fell_short = r.len < ROLZ_MIN;
    if (c->flexible && !fell_short) {                      /* :143-167: cut the match where "this match + what follows" prices best */
        uint32_t best = 0, keep = r.len;
        for (uint32_t i = r.len; i >= 1; i--) {
            const rolz_hit ahead = ring_search(c, d, pos + i, ring_of(c, d + pos + i - 1), pos);
            rolz_hit cut = r; cut.len = i;
            const uint32_t v = price(cut) + price(ahead);
            if (i == r.len) best = v;
            else if (v > best) { keep = i; best = v; }
        }
        r.len = keep;
    }
    if (fell_short) {                                      /* the 16 newest positions behind the same byte, :171-186 */
        r.len = ROLZ_MIN - 1u;
        r.rank = ROLZ_NONE;
        uint32_t p = c->row_head[c->row_now];
        for (uint32_t i = 0; i < ROLZ_ROW; i++) {
            const uint32_t at = p == ROLZ_NONE ? 0u : p;
            uint32_t j = 0;
            while (j < ROLZ_MAX && d[pos + j] == d[at + j]) j++;
            if (j > r.len) { r.rank = ROLZ_RING + i; r.len = j; }
            if (p != ROLZ_NONE) p = c->row_prev[p];
        }
    }
    if (r.len < ROLZ_MIN) { r.rank = ROLZ_NONE; r.len = 1; }
    if ((!c->flexible || fell_short) && r.len > 1) {       /* lazy evaluation, :188-196 */
        for (uint32_t i = 1; i < ROLZ_MIN; i++) {
            const rolz_hit ahead = ring_search(c, d, pos + i, ring_of(c, d + pos + i - 1), pos);
            if (price(ahead) > price(r) + i * ROLZ_RING) { r.rank = ROLZ_NONE; r.len = 1; break; }
        }
    }
    return r;
}

static void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* ROLZ parse only: per token the rank (ROLZ_NONE = literal) and length; returns the number of tokens */
uint32_t cro_rolz_parse(cro_rolz* c, const uint8_t* in, uint32_t n, uint32_t* rank_out, uint32_t* len_out) {
    uint32_t pos = 1, t = 0;
    if (n == 0) return 0;
    matcher_start(c, n);
    while (pos < n) {                                      /* lzmatch_thread, cr-coder.c:110-136 */
        rolz_hit r = { ROLZ_NONE, 1 };
        if (pos + ROLZ_TAIL < n) r = matcher_find(c, in, pos);
        for (uint32_t i = 0; i < r.len; i++) matcher_feed(c, in, pos + i);
        rank_out[t] = r.rank; len_out[t] = r.len; t++;
        pos += r.len;
    }
    return t;
}

/* lzencode, cr-coder.c:138-258. An empty block is not defined by the reference (it reads m_data[0] and its
 * decoder then produces one byte); the block loop never passes one. Written here as an empty stored block. */
uint32_t cro_rolz_encode(cro_rolz* c, const uint8_t* in, uint32_t n, uint8_t* out) {
    memset(out, 0, CRO_ROLZ_HEADER);
    if (n == 0) return CRO_ROLZ_HEADER;
    uint32_t count[256] = {0};
    for (uint32_t i = 0; i < n; i++) count[in[i]]++;
    int esc = 0;
    for (int i = 1; i < 256; i++) if (count[esc] > count[i]) esc = i;      /* :167-175 */

    cro_buf main_s, side_s;
    cro_buf_init(&main_s); cro_buf_init(&side_s);
    cro_buf_resize(&main_s, CRO_ROLZ_HEADER);
    cro_rc rc, rc_side;
    cro_rc_enc_init(&rc); cro_rc_enc_init(&rc_side);
    matcher_start(c, n);
    uint32_t pos = 1, codes = 0;
    int stored = 0;
    while (pos < n) {                                      /* :196-236 with the matching thread's loop :110-136 folded in */
        rolz_hit r = { ROLZ_NONE, 1 };
        if (pos + ROLZ_TAIL < n) r = matcher_find(c, in, pos);
        for (uint32_t i = 0; i < r.len; i++) matcher_feed(c, in, pos + i);
        if (r.rank != ROLZ_NONE) {
            cro_ppm_encode(c->ppm, &rc, esc, &main_s);
            cro_model_encode(&c->len_model, &rc_side, (int)r.len, 4, &side_s);
            cro_model_encode(&c->idx_model, &rc_side, (int)r.rank, 4, &side_s);
            codes++;
        } else {
            cro_ppm_encode(c->ppm, &rc, in[pos], &main_s);
            if (in[pos] == esc) { cro_model_encode(&c->len_model, &rc_side, 0, 4, &side_s); codes++; }
        }
        for (uint32_t i = 0; i < r.len; i++) cro_ppm_push(c->ppm, in[pos++]);
        if (main_s.size >= n) { stored = 1; break; }       /* :233-235 */
    }
    uint32_t total;
    if (stored) {                                          /* :247-257 */
        memcpy(out + CRO_ROLZ_HEADER, in, n);
        total = CRO_ROLZ_HEADER + n;
    } else {
        cro_rc_enc_flush(&rc, &main_s);
        cro_rc_enc_flush(&rc_side, &side_s);
        memcpy(out, main_s.data, main_s.size);
        memset(out, 0, CRO_ROLZ_HEADER);
        out[0] = in[0]; out[1] = 1; out[2] = (uint8_t)esc;
        put32(out + 4, n); put32(out + 8, codes); put32(out + 12, main_s.size);
        memcpy(out + main_s.size, side_s.data, side_s.size);
        total = main_s.size + side_s.size;
    }
    cro_buf_free(&main_s); cro_buf_free(&side_s);
    return total;
}

/* lzdecode, cr-coder.c:283-379 (the side-stream queue thread only pre-decodes the same sequential stream) */
uint32_t cro_rolz_decode(cro_rolz* c, const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap) {
    if (n < CRO_ROLZ_HEADER) return 0xFFFFFFFFu;
    if (!in[1]) {                                          /* :303-308 */
        if (n - CRO_ROLZ_HEADER > cap) return 0xFFFFFFFFu;
        memcpy(out, in + CRO_ROLZ_HEADER, n - CRO_ROLZ_HEADER);
        return n - CRO_ROLZ_HEADER;
    }
    const uint32_t total = get32(in + 4), side_off = get32(in + 12);
    uint32_t codes = get32(in + 8);
    const int esc = in[2];
    if (total > cap || total == 0 || side_off > n) return 0xFFFFFFFFu;
    out[0] = in[0];
    matcher_start(c, total);
    const uint8_t* p_main = in + CRO_ROLZ_HEADER;
    const uint8_t* p_side = in + side_off;
    cro_rc rc, rc_side;
    cro_rc_dec_init_end(&rc, &p_main, in + n);
    cro_rc_dec_init_end(&rc_side, &p_side, in + n);
    uint32_t have = 1;
    while (have < total) {                                 /* :334-375 */
        const int sym = cro_ppm_decode(c->ppm, &rc, &p_main);
        uint32_t len = 1;
        if (sym == esc) {
            uint32_t l = 0, rank = 0;
            if (codes > 0) {                               /* :265-277: an exhausted stream yields zeros */
                codes--;
                l = (uint32_t)cro_model_decode(&c->len_model, &rc_side, 4, &p_side);
                if (l > 0) rank = (uint32_t)cro_model_decode(&c->idx_model, &rc_side, 4, &p_side);
            }
            if (l == 0) {
                out[have++] = (uint8_t)esc;
            } else {
                const uint32_t from = matcher_at(c, rank);
                /* corrupt stream (no encoder writes these; the reference trusts its input and would copy from an empty
                 * ring / the zero-filled rows): a match before the matcher's warm-up (cr-matcher.c:68), a rank nothing
                 * was fed for, a source that is not in front of the write position, a length past the block */
                if (from == ROLZ_NONE || from >= have || have < ROLZ_WARM || have + l > total) return 0xFFFFFFFFu;
                for (uint32_t i = 0; i < l; i++) out[have + i] = out[from + i];
                have += l;
                len = l;
            }
        } else {
            out[have++] = (uint8_t)sym;
        }
        for (uint32_t i = len; i > 0; i--) {               /* :369-373 */
            matcher_feed(c, out, have - i);
            cro_ppm_push(c->ppm, out[have - i]);
        }
    }
    if (cro_rc_dec_left_interval(&rc) || cro_rc_dec_left_interval(&rc_side)) return 0xFFFFFFFFu;   /* corrupt stream */
    return have;
}

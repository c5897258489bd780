/*
 * oracle/cr_oracle_rop.c — comprop block codec: LZP predictor + single PPM/range stream.
 * TEST INFRASTRUCTURE ONLY (see cr_oracle.h). Restated from the behaviour of
 * /root/reference/src/ropmain/cr-coder.c and cr-matcher.c; citations are to those files.
 *
 * Block layout (cr-coder.c:59-66, sizeof == 20 on LP64):
 *   [0] 1 = coded, 0 = stored     [4..7] original size, LE     [8] escape byte
 *   [9..17] the first nine input bytes (never modelled)        [1..3],[18..19] zero padding
 * followed by the range-coder bytes, or — stored form — 20 zero bytes followed by the raw input.
 */
#include "cr_oracle.h"
#include <stdlib.h>
#include <string.h>

#define LZP_MIN 4u           /* cr-matcher.h:36 */
#define LZP_MAX 255u         /* cr-matcher.h:37 */
#define LZP_TAIL 1024u       /* cr-coder.c:103: no prediction is tried this close to the end */
#define LZP_SKIP 9u          /* cr-coder.c:143-145: coding starts after nine raw bytes */

struct cro_rop {
    cro_ppm*  ppm;
    /* three "where did this context last occur" tables (cr-matcher.c:35-50), dense like the
     * reference, with an undo log so a per-block reset is O(block) instead of 68 MB */
    uint32_t* t8; uint32_t* t4; uint32_t* t2;
    uint32_t* log8; uint32_t* log4; uint32_t* log2;
    uint32_t  nlog, logcap;
};

static uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

/* cr-matcher.c:31-33 (little-endian loads of the 8 / 4 / 2 bytes in front of the position) */
static uint32_t key8(const uint8_t* at) { uint64_t x = ld64(at - 8); return (uint32_t)((x ^ (x >> 20) ^ (x >> 40)) & 0xffffff); }
static uint32_t key4(const uint8_t* at) { uint32_t x = ld32(at - 4); return (x ^ (x >> 6) ^ (x >> 12)) & 0xfffff; }
static uint32_t key2(const uint8_t* at) { return ld16(at - 2); }

cro_rop* cro_rop_new(void) {
    cro_rop* c = (cro_rop*)calloc(1, sizeof *c);
    c->ppm = cro_ppm_new();
    c->t8 = (uint32_t*)malloc(sizeof(uint32_t) << 24);
    c->t4 = (uint32_t*)malloc(sizeof(uint32_t) << 20);
    c->t2 = (uint32_t*)malloc(sizeof(uint32_t) << 16);
    for (uint32_t i = 0; i < (1u << 24); i++) c->t8[i] = 8;    /* cr-matcher.c:41-49: an empty  */
    for (uint32_t i = 0; i < (1u << 20); i++) c->t4[i] = 4;    /* slot points just behind its   */
    for (uint32_t i = 0; i < (1u << 16); i++) c->t2[i] = 2;    /* own context length            */
    c->logcap = 1u << 16;
    c->log8 = (uint32_t*)malloc(c->logcap * 4);
    c->log4 = (uint32_t*)malloc(c->logcap * 4);
    c->log2 = (uint32_t*)malloc(c->logcap * 4);
    return c;
}

void cro_rop_free(cro_rop* c) {
    if (!c) return;
    cro_ppm_free(c->ppm);
    free(c->t8); free(c->t4); free(c->t2); free(c->log8); free(c->log4); free(c->log2); free(c);
}

/* cr-coder.c:73-83 */
void cro_rop_reset(cro_rop* c) { cro_ppm_reset(c->ppm); }

/* matcher_init (cr-matcher.c:35-50) by undoing the previous block's writes */
static void lzp_clear(cro_rop* c) {
    for (uint32_t i = 0; i < c->nlog; i++) { c->t8[c->log8[i]] = 8; c->t4[c->log4[i]] = 4; c->t2[c->log2[i]] = 2; }
    c->nlog = 0;
}

/* cr-matcher.c:91-96 */
static void lzp_learn(cro_rop* c, const uint8_t* d, uint32_t pos) {
    if (c->nlog == c->logcap) {
        c->logcap *= 2;
        c->log8 = (uint32_t*)realloc(c->log8, c->logcap * 4);
        c->log4 = (uint32_t*)realloc(c->log4, c->logcap * 4);
        c->log2 = (uint32_t*)realloc(c->log2, c->logcap * 4);
    }
    uint32_t a = key8(d + pos), b = key4(d + pos), e = key2(d + pos);
    c->log8[c->nlog] = a; c->log4[c->nlog] = b; c->log2[c->nlog] = e; c->nlog++;
    c->t8[a] = pos; c->t4[b] = pos; c->t2[e] = pos;
}

/* cr-matcher.c:59-73: longest verified context wins; the 2-byte table is taken on trust */
static uint32_t lzp_predict(const cro_rop* c, const uint8_t* d, uint32_t pos) {
    uint32_t p8 = c->t8[key8(d + pos)];
    if (!memcmp(d + p8 - 8, d + pos - 8, 8)) return p8;
    uint32_t p4 = c->t4[key4(d + pos)];
    if (!memcmp(d + p4 - 4, d + pos - 4, 4)) return p4;
    return c->t2[key2(d + pos)];
}

/* cr-matcher.c:75-89: length of the agreement between prediction and reality, 4..255, else 1 */
static uint32_t lzp_length(const cro_rop* c, const uint8_t* d, uint32_t pos) {
    uint32_t from = lzp_predict(c, d, pos), len = 0;
    if (from != 0)
        while (len < LZP_MAX && d[from + len] == d[pos + len]) len++;
    return len < LZP_MIN ? 1 : len;
}

/* cr-coder.c:95-118 without the 32000-entry hand-off: token lengths from `start` to the end */
uint32_t cro_rop_parse(cro_rop* c, const uint8_t* in, uint32_t n, uint32_t start, uint32_t* lens) {
    uint32_t pos = start, nt = 0;
    lzp_clear(c);
    while (pos < n) {
        uint32_t len = 1;
        if (pos + LZP_TAIL < n) {
            len = lzp_length(c, in, pos);
            for (uint32_t i = 0; i < len; i++) lzp_learn(c, in, pos + i);
        }
        lens[nt++] = len;
        pos += len;
    }
    return nt;
}

static void put_stored(const uint8_t* in, uint32_t n, uint8_t* out) {    /* cr-coder.c:222-228 */
    memset(out, 0, CRO_ROP_HEADER);
    memcpy(out + CRO_ROP_HEADER, in, n);
}

/* cr-coder.c:119-229 */
uint32_t cro_rop_encode(cro_rop* c, const uint8_t* in, uint32_t n, uint8_t* out) {
    if (n < 16) { put_stored(in, n, out); return CRO_ROP_HEADER + n; }   /* cr-coder.c:140-142 */

    /* escape byte = least frequent value, lowest value on ties (cr-coder.c:147-156) */
    uint32_t hist[256] = {0};
    for (uint32_t i = 0; i < n; i++) hist[in[i]]++;
    int esc = 0;
    for (int v = 1; v < 256; v++) if (hist[v] < hist[esc]) esc = v;

    uint32_t* lens = (uint32_t*)malloc((size_t)(n + 1) * sizeof(uint32_t));
    cro_rop_parse(c, in, n, LZP_SKIP, lens);

    cro_buf b; cro_buf_init(&b);
    cro_rc rc; cro_rc_enc_init(&rc);
    cro_ppm* m = c->ppm;
    uint32_t pos = LZP_SKIP, t = 0;
    int stored = 0;
    while (pos < n) {
        uint32_t len = lens[t++];
        if (len > 1) {                                  /* cr-coder.c:185-188: esc, then length */
            cro_ppm_encode(m, &rc, esc, &b);            /* coded in the context ending in esc  */
            cro_ppm_push(m, esc);
            cro_ppm_encode(m, &rc, (int)len, &b);
        } else {                                        /* cr-coder.c:190-196 */
            cro_ppm_encode(m, &rc, in[pos], &b);
            if (in[pos] == esc) {                       /* a literal escape byte is "length 0" */
                cro_ppm_push(m, esc);
                cro_ppm_encode(m, &rc, 0, &b);
            }
        }
        for (; len; len--) cro_ppm_push(m, in[pos++]);  /* cr-coder.c:198-202 */
        if (CRO_ROP_HEADER + b.size >= n) { stored = 1; break; }    /* cr-coder.c:204-206 */
    }
    free(lens);
    if (stored) { cro_buf_free(&b); put_stored(in, n, out); return CRO_ROP_HEADER + n; }

    cro_rc_enc_flush(&rc, &b);                          /* cr-coder.c:210 */
    memset(out, 0, CRO_ROP_HEADER);                     /* cr-coder.c:213-216 */
    out[0] = 1;
    out[4] = (uint8_t)n; out[5] = (uint8_t)(n >> 8); out[6] = (uint8_t)(n >> 16); out[7] = (uint8_t)(n >> 24);
    out[8] = (uint8_t)esc;
    memcpy(out + 9, in, 9);
    memcpy(out + CRO_ROP_HEADER, b.data, b.size);
    uint32_t total = CRO_ROP_HEADER + b.size;
    cro_buf_free(&b);
    return total;
}

/* cr-coder.c:231-292 */
uint32_t cro_rop_decode(cro_rop* c, const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap) {
    if (n < CRO_ROP_HEADER) return 0xFFFFFFFFu;
    if (!in[0]) {                                                        /* cr-coder.c:243-248 */
        uint32_t raw = n - CRO_ROP_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        memcpy(out, in + CRO_ROP_HEADER, raw);
        return raw;
    }
    uint32_t total = (uint32_t)in[4] | ((uint32_t)in[5] << 8) | ((uint32_t)in[6] << 16) | ((uint32_t)in[7] << 24);
    int esc = in[8];
    if (total > cap || total < LZP_SKIP) return 0xFFFFFFFFu;
    memcpy(out, in + 9, LZP_SKIP);                                       /* cr-coder.c:251-254 */

    const uint8_t* src = in + CRO_ROP_HEADER;
    cro_rc rc; cro_rc_dec_init_end(&rc, &src, in + n);
    cro_ppm* m = c->ppm;
    lzp_clear(c);
    uint32_t have = LZP_SKIP;
    while (have < total) {                                               /* cr-coder.c:259-290 */
        uint32_t len = 1;
        int s = cro_ppm_decode(m, &rc, &src);
        if (s != esc) {
            out[have] = (uint8_t)s;
        } else {
            cro_ppm_push(m, esc);
            len = (uint32_t)cro_ppm_decode(m, &rc, &src);
            if (len == 0) {
                len = 1;
                out[have] = (uint8_t)esc;
            } else {
                if (have + len > cap) return 0xFFFFFFFFu;
                uint32_t from = lzp_predict(c, out, have);
                for (uint32_t i = 0; i < len; i++) out[have + i] = out[from + i];
            }
        }
        for (uint32_t i = 0; i < len; i++) {                             /* cr-coder.c:284-288 */
            cro_ppm_push(m, out[have + i]);
            lzp_learn(c, out, have + i);
        }
        have += len;
    }
    if (cro_rc_dec_left_interval(&rc)) return 0xFFFFFFFFu;               /* corrupt stream (cr_oracle_core.c, cro_rc_dec_target) */
    return have;
}

void cro_rop_encode_blocks(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size, uint32_t nblocks,
                           uint8_t* out, const uint64_t* out_off, uint32_t* out_size) {
    cro_rop* c = cro_rop_new();
    for (uint32_t b = 0; b < nblocks; b++) {
        cro_rop_reset(c);
        out_size[b] = cro_rop_encode(c, in + in_off[b], in_size[b], out + out_off[b]);
    }
    cro_rop_free(c);
}

void cro_rop_decode_blocks(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size, uint32_t nblocks,
                           uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_size) {
    cro_rop* c = cro_rop_new();
    for (uint32_t b = 0; b < nblocks; b++) {
        cro_rop_reset(c);
        out_size[b] = cro_rop_decode(c, in + in_off[b], in_size[b], out + out_off[b], out_cap[b]);
    }
    cro_rop_free(c);
}

/*
 * comprox_amd/csrc/crgpu_rox2.h — comprox lzencode for the batched API on the comprop encoder's kernel pipeline.
 *
 * Reference: /root/reference/src/roxmain/cr-coder.c:153-318 (lzencode).
 *
 * The main stream is a sequence of ppm_encode calls (one per token: the literal byte, or the escape byte for a
 * match), so it fits the per-context pipeline of crgpu_rop2.h unchanged: k_rox_events walks the token loop, writes
 * one event {context, symbol} per token and codes the three side streams (small adaptive models, their own range
 * coders) into a per-block staging area; k_rop_links / _o3 / _o2 / _o1 turn the events into range-coder triples;
 * k_rox_rc codes them and assembles the block (32-byte header, main stream, side streams).
 */
#ifndef CRGPU_ROX2_H
#define CRGPU_ROX2_H

#include "crgpu_rox.h"
#include "crgpu_rop2.h"

/* V.ctr words of a comprox block: [0] events, [3] escape byte | 0x200 (block too large), then */
#define CR_ROXC_NSPOS 4
#define CR_ROXC_NPOS  5
#define CR_ROXC_NLEN  6
#define CR_ROXC_BSPOS 7     /* bytes of the three side streams */
#define CR_ROXC_BPOS  8
#define CR_ROXC_BLEN  9
#define CR_ROXC_LIST  10    /* symbols listed for the three side streams (spos, pos, len) */

/* token loop of lzencode (cr-coder.c:213-276) without the main stream's coding.
 * The context a token is coded in is the four bytes in front of it (a literal pushes its byte, a match leaves its
 * last four bytes, the escape byte is not pushed: cr-coder.c:222-229,262-270), empty at position 0: the loop only
 * records token positions and symbols, contexts are gathered 64 tokens at a time. The only memory the decisions wait
 * for is the 256 bytes at the previous distance (matcher_lookup's repeat test, cr-matcher.c:246-251); they are
 * requested as soon as the next token's position is known, ahead of the side coders. */
CR_DEV void cr_rox_flush_events(const uint8_t* src, uint32_t n, CrEvViews& V, uint32_t nev0, uint32_t count, uint32_t ppos, uint32_t psym) {
    const uint32_t lane = cr_lane();
    if (lane < count) {
        uint32_t ctx;
        if (ppos >= 4u) ctx = __builtin_bswap32(*reinterpret_cast<const cr_u32u*>(src + ppos - 4u));
        else {
            ctx = 0;
            for (uint32_t i = 0; i < ppos; i++) ctx = (ctx << 8) | src[i];
        }
        V.ev_ctx[nev0 + lane] = ctx;
        V.ev_sym[nev0 + lane] = (uint16_t)psym;
    }
    (void)n;
}

/* The three side streams only depend on their own symbols (own models, own range coder), so the token loop just
 * lists them in the upper half of each stream's staging area (n + 128 bytes: one byte per length / short-distance
 * symbol, at most one per token; u16 {digit, symbol} per distance digit, at most five per match of >= 10 bytes) and
 * three waves code the lists side by side afterwards (cr_rox_code_side). */
#define CR_ROX_LIST(w_) (side + (u64)(w_) * side_stride + ((side_stride / 2u) & ~(u64)1u))
#define CR_ROX_PUT(w_, cnt_, m_, sym_) do { \
        if (lane == 0) { if ((w_) == 1) reinterpret_cast<uint16_t*>(CR_ROX_LIST(1))[cnt_] = (uint16_t)(((m_) << 8) | (sym_)); else CR_ROX_LIST(w_)[cnt_] = (uint8_t)(sym_); } \
        cnt_++; } while (0)

CR_DEV void cr_rox_emit_events(const uint8_t* src, uint32_t n, const CrRoxTables& T, uint8_t* side, u64 side_stride,
                               CrEvViews& V, CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    const uint32_t long_min = 10u + (n > 16777216u ? 1u : 0u);           /* cr-coder.c:192 */
    const uint32_t esc = cr_pick_escape(src, n, sh.hist);
    uint32_t c_spos = 0, c_pos = 0, c_len = 0;                           /* list lengths: stream 0 = spos, 1 = pos, 2 = len */
    CrWindow win;
    cr_window_init(win, src, n, 0);

    uint32_t pos = 0, repeat = 0, prev_dist = 0, n_spos = 0, n_pos = 0, n_len = 0, nev = 0;
    /* events are written 64 at a time: lane j of (ppos, psym) holds event nev0 + j */
    uint32_t ppos = 0, psym = 0, nev0 = 0;
    /* the match kernel's answers for 64 positions at a time: lane j holds those of position tbase + j */
    uint32_t tbase = 0, t_mp = T.ml_pos[lane], t_np = T.nprev[lane], t_len = (uint32_t)T.ml_len[lane] | ((uint32_t)T.nl_len[lane] << 8);
    uint32_t rep_x = 0, cur_x = 0;                                       /* lane l: bytes 4l .. 4l+3 at pos - repeat and at pos (valid when repeat != 0) */
    while (pos < n) {                                                    /* cr-coder.c:213-276 */
        uint32_t from = CR_ROX_NONE, len = 1;
        if (pos - tbase >= CRGPU_WAVE && pos + CR_ROX_TAIL < n) {
            tbase = pos;
            t_mp = T.ml_pos[tbase + lane]; t_np = T.nprev[tbase + lane];
            t_len = (uint32_t)T.ml_len[tbase + lane] | ((uint32_t)T.nl_len[tbase + lane] << 8);
        }
        /* Runs of plain literals, up to 64 positions at a time. Whether a position needs the parser's state (the previous
         * distance) only depends on the match kernel's answers: without a long candidate (ml_len < 2) matcher_lookup goes
         * straight to the short cache (:319-338), whose verdict is a function of the position alone; and behind
         * n - 1024 every position is a literal (cr-coder.c:136). Literals change neither `repeat` nor `prev_dist`, so a run
         * of them is emitted in one step: events by position, a zero length symbol for every escape byte among them. */
        {
            bool plain;
            const uint32_t q = (pos + CR_ROX_TAIL < n ? tbase : pos) + lane;
            if (q + CR_ROX_TAIL >= n) plain = q < n;
            else {
                const uint32_t ml_q = t_len & 0xffu, nl_q = t_len >> 8;
                plain = ml_q < 2u && (nl_q < CR_ROX_NEAR_MIN || (nl_q < long_min && t_np + 256u <= q));
            }
            const uint32_t first_lane = pos + CR_ROX_TAIL < n ? pos - tbase : 0u;
            const u64 pm = cr_ballot(plain) >> first_lane;
            const uint32_t run = pm == ~0ull ? 64u - first_lane : (uint32_t)__builtin_ctzll(~pm);
            if (run > 0u) {
                cr_rox_flush_events(src, n, V, nev0, nev - nev0, ppos, psym);          /* what the token-by-token part has buffered */
                const bool mine = lane < run;
                const uint32_t c = mine ? (uint32_t)src[pos + lane] : 0u;
                const u64 em = cr_ballot(mine && c == esc);
                if (mine && c == esc) CR_ROX_LIST(2)[c_len + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = 0;   /* cr-coder.c:264-267 */
                c_len += (uint32_t)__builtin_popcountll(em); n_len += (uint32_t)__builtin_popcountll(em);
                cr_rox_flush_events(src, n, V, nev, run, pos + lane, c | CR_EV_LAST);
                nev += run; nev0 = nev;
                pos += run;
                /* the next token's bytes at the previous distance (the sequential part expects them loaded) */
                if (repeat != 0u && pos + CR_ROX_TAIL < n) {
                    rep_x = *reinterpret_cast<const cr_u32u*>(src + pos - repeat + lane * 4u);
                    cur_x = *reinterpret_cast<const cr_u32u*>(src + pos + lane * 4u);
                }
                continue;
            }
        }
        if (pos + CR_ROX_TAIL < n) {                                     /* matcher_lookup, cr-matcher.c:237-340 */
            const uint32_t tl = pos - tbase, tlen = cr_lane_get(t_len, tl);
            uint32_t mp = cr_lane_get(t_mp, tl), ml = tlen & 0xffu;
            if (ml < 2u) mp = CR_ROX_NONE;                               /* (flexible parsing keeps the uncut position in ml_pos) */
            if (mp != CR_ROX_NONE) {
                uint32_t rp = pos - repeat, rl = 0;                      /* the previous distance again (:246-251) */
                if (rp < pos) {
                    const uint32_t x = rep_x ^ cur_x;
                    const u64 diff = cr_ballot(x != 0u);
                    rl = CR_ROX_MAX;
                    if (diff) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(diff);
                        rl = l * 4u + ((uint32_t)__builtin_ctz(cr_lane_get(x, l)) >> 3);
                        if (rl > CR_ROX_MAX) rl = CR_ROX_MAX;
                    }
                }
                if (ml < rl + 3u + (mp + 64u < pos ? 1u : 0u) + (mp + 4096u < pos ? 1u : 0u) + (mp + 1048576u < pos ? 1u : 0u)) { mp = rp; ml = rl; }
            }
            if (ml < CR_ROX_NEAR_MIN) { mp = cr_lane_get(t_np, tl); ml = tlen >> 8; }                 /* (:319-331) */
            if (!(ml < CR_ROX_NEAR_MIN || (ml < long_min && mp + 256u <= pos))) {                     /* (:333-338) */
                from = mp; len = ml; repeat = pos - mp;
            }
        }
        /* the next token starts at pos + len: its bytes at the (possibly new) previous distance */
        if (repeat != 0u && pos + len + CR_ROX_TAIL < n) {
            rep_x = *reinterpret_cast<const cr_u32u*>(src + pos + len - repeat + lane * 4u);
            cur_x = *reinterpret_cast<const cr_u32u*>(src + pos + len + lane * 4u);
        }
        uint32_t sym;
        if (from != CR_ROX_NONE) {
            sym = esc;
            uint32_t dist = pos - from;
            if (dist == prev_dist) dist = 0;                             /* cr-coder.c:232-234 */
            CR_ROX_PUT(2, c_len, CR_SIDE_LEN, len); n_len++;
            if (len < long_min) {
                CR_ROX_PUT(0, c_spos, CR_SIDE_SPOS, dist); n_spos++;
            } else {                                                     /* cr-coder.c:243-258 */
                uint32_t j = dist * 8u, i = 0;
                while (j >= 128u && i < 2u) { CR_ROX_PUT(1, c_pos, CR_SIDE_POS + i, j % 128u + 128u); i++; j /= 128u; }
                if (i >= 2u) while (j >= 64u && i < 5u) { CR_ROX_PUT(1, c_pos, CR_SIDE_POS + i, j % 64u + 64u); i++; j /= 64u; }
                CR_ROX_PUT(1, c_pos, CR_SIDE_POS + i, j);
                n_pos++;
            }
            prev_dist = dist;
        } else {
            const uint32_t c = cr_window_at(win, pos);
            sym = c;
            if (c == esc) { CR_ROX_PUT(2, c_len, CR_SIDE_LEN, 0u); n_len++; }
        }
        if (lane == nev - nev0) { ppos = pos; psym = sym | CR_EV_LAST; }
        nev++;
        if (nev - nev0 == CRGPU_WAVE) {
            cr_rox_flush_events(src, n, V, nev0, CRGPU_WAVE, ppos, psym);
            nev0 = nev;
        }
        pos += len;
    }
    cr_rox_flush_events(src, n, V, nev0, nev - nev0, ppos, psym);
    if (lane == 0) {
        V.ctr[0] = nev; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = esc;
        V.ctr[CR_ROXC_NSPOS] = n_spos; V.ctr[CR_ROXC_NPOS] = n_pos; V.ctr[CR_ROXC_NLEN] = n_len;
        V.ctr[CR_ROXC_LIST + 0] = c_spos; V.ctr[CR_ROXC_LIST + 1] = c_pos; V.ctr[CR_ROXC_LIST + 2] = c_len;
    }
}

/* one wave per side stream (w = 0 spos, 1 pos, 2 len): the listed symbols through their models (cr-model.c, increments
 * roxmain/cr-coder.c:52) and the stream's own range coder */
CR_DEV void cr_rox_code_side(uint32_t w, uint8_t* side, u64 side_stride, CrEvViews& V, CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    const uint32_t count = cr_uni(V.ctr[CR_ROXC_LIST + w]);
    const uint8_t* list = CR_ROX_LIST(w);
    CrSink s; s.dst = side + (u64)w * side_stride; s.n = 0;
    CrRc rc; cr_rc_init(rc);
    for (uint32_t k0 = 0; k0 < count; k0 += CRGPU_WAVE) {
        uint32_t mine = 0;
        if (k0 + lane < count) mine = w == 1u ? reinterpret_cast<const uint16_t*>(list)[k0 + lane]
                                              : (uint32_t)list[k0 + lane] | ((w == 0u ? CR_SIDE_SPOS : CR_SIDE_LEN) << 8);
        const uint32_t lim = count - k0 < CRGPU_WAVE ? count - k0 : CRGPU_WAVE;
        for (uint32_t l = 0; l < lim; l++) {
            const uint32_t v = cr_lane_get(mine, l), m = v >> 8, sym = v & 0xffu;
            const uint32_t inc = m == CR_SIDE_LEN ? 30u : m == CR_SIDE_SPOS ? 1u : 1u << (2u * (m - CR_SIDE_POS));
            cr_side_encode(sh, m, sym, inc, rc, s);
        }
    }
    cr_rc_pin(rc); cr_rc_flush(rc, s);
    if (lane == 0) V.ctr[CR_ROXC_BSPOS + w] = s.n;
}
#undef CR_ROX_PUT

/* main stream from the triples, then the block: cr-coder.c:273-275 (stored when the main stream alone reaches the
 * input size), :280-318 (streams back to back behind the header) */
CR_DEV uint32_t cr_rox_finish(const uint8_t* src, uint32_t n, uint8_t* dst, const uint8_t* side, u64 side_stride,
                              CrEvViews& V, u64* ring) {
    const uint32_t lane = cr_lane();
    const uint32_t long_min = 10u + (n > 16777216u ? 1u : 0u);
    const uint32_t esc = cr_uni(V.ctr[3]) & 0xffu;
    uint32_t got = cr_code_events_fast(n, dst + CR_ROX_HEADER, CR_ROX_HEADER, V, ring);
    if (got == 0u) got = cr_code_events(n, dst + CR_ROX_HEADER, CR_ROX_HEADER, V);
    if (got == 0xFFFFFFFFu) {
        cr_wave_sync();
        cr_rox_store_raw(src, n, dst);
        return CR_ROX_HEADER + n;
    }
    cr_wave_sync();
    const uint32_t n_spos = cr_uni(V.ctr[CR_ROXC_NSPOS]), n_pos = cr_uni(V.ctr[CR_ROXC_NPOS]), n_len = cr_uni(V.ctr[CR_ROXC_NLEN]);
    const uint32_t b_spos = cr_uni(V.ctr[CR_ROXC_BSPOS]), b_pos = cr_uni(V.ctr[CR_ROXC_BPOS]), b_len = cr_uni(V.ctr[CR_ROXC_BLEN]);
    const uint32_t o_spos = CR_ROX_HEADER + got, o_pos = o_spos + b_spos, o_len = o_pos + b_pos;
    if ((u64)o_len + b_len > cr_bound_rox(n)) return 0xFFFFFFFFu;           /* cannot happen (crgpu_device.h); never write past the slot */
    for (uint32_t i = lane; i < b_spos; i += CRGPU_WAVE) dst[o_spos + i] = side[i];
    for (uint32_t i = lane; i < b_pos; i += CRGPU_WAVE) dst[o_pos + i] = side[side_stride + i];
    for (uint32_t i = lane; i < b_len; i += CRGPU_WAVE) dst[o_len + i] = side[2u * side_stride + i];
    if (lane < CR_ROX_HEADER) {                                          /* cr-coder.c:289-297 */
        uint32_t word = lane >> 2, v = 0;
        const uint32_t fields[8] = {1u | (long_min << 8) | (esc << 16), n, n_spos, n_pos, n_len, o_spos, o_pos, o_len};
        v = fields[word];
        dst[lane] = (uint8_t)(v >> (8u * (lane & 3u)));
    }
    return o_len + b_len;
}

#endif

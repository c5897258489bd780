/*
 * comprox_amd/csrc/crgpu_dict.h — static-dictionary word substitution on gfx950 (one wavefront
 * per datablock).
 *
 * Reference: /root/reference/src/cr-diccode.c — dictionary_encode (:142-221),
 * dictionary_encode_imp (:285-362), dictionary_decode (:223-283), dictionary_decode_imp (:364-425).
 *
 * Device dictionary (built once per file on the host, crgpu_dict_create):
 *   next   u32[nnodes][128]  child index of the reference's trie (cr-diccode.c:38-41,47-70), with the
 *                            upper-case root links and the '.' ',' ':' ';' aliases of ' ' already
 *                            applied (:107-117); bit 31 set when the child is a terminal node
 *   ids    i32[nnodes]       word number of a terminal node
 *   words  u8[nwords][24]    word text (with its trailing ' '), wlen u8[nwords]
 *
 * Encoding walks the trie for 64 consecutive positions at once (one lane per position; the walk is
 * at most 22 dependent loads), resolves "a match swallows the positions it covers" in lane order,
 * and places the variable-length codes with a DPP prefix sum. Decoding parses a piece from its end,
 * one token per step, with the word body copied by the lanes in parallel; the deferred
 * sentence-case fix-up (:415-419) is decided from the bytes captured while they are written.
 */
#ifndef CRGPU_DICT_H
#define CRGPU_DICT_H

#include "crgpu_wave.h"

#define CR_DIC_WORD_MAX    20u                          /* cr-diccode.h:42 */
#define CR_DIC_PIECE       1000000u                     /* cr-diccode.c:176-178 */
#define CR_DIC_WORD_STRIDE 24u
#define CR_DIC_TERMINAL    0x80000000u

struct CrDict {
    const uint32_t* next;
    const int32_t*  ids;
    const uint8_t*  words;
    const uint8_t*  wlen;
    uint32_t        nwords;       /* dic_len */
    uint32_t        level1;       /* LEVEL1_WORD_NUM(dic_len), cr-diccode.h:40 */
};

struct CrDictShared {
    uint32_t hist[256];
    uint8_t  escmap[256];
    uint8_t  esc[16];
};

CR_DEV bool cr_is_alpha(uint32_t c) { return ((c | 0x20u) - 'a') < 26u; }
CR_DEV bool cr_is_upper(uint32_t c) { return (c - 'A') < 26u; }

/* inclusive prefix maximum over the 64 lanes */
CR_DEV uint32_t cr_scan_max_incl(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;
    return v;
}

/* the ten least frequent byte values, in the reference's order (cr-diccode.c:160-171) */
CR_DEV void cr_dict_pick_escapes(const uint8_t* d, uint32_t n, CrDictShared& sh) {
    const uint32_t lane = cr_lane();
    for (uint32_t i = lane; i < 256u; i += CRGPU_WAVE) { sh.hist[i] = 0; sh.escmap[i] = 0; }
    cr_wave_sync();
    const uint32_t body = n & ~3u;
    for (uint32_t i = lane * 4u; i < body; i += 4u * CRGPU_WAVE) {
        uint32_t v = *reinterpret_cast<const cr_u32u*>(d + i);
        atomicAdd(&sh.hist[v & 0xffu], 1u);
        atomicAdd(&sh.hist[(v >> 8) & 0xffu], 1u);
        atomicAdd(&sh.hist[(v >> 16) & 0xffu], 1u);
        atomicAdd(&sh.hist[v >> 24], 1u);
    }
    if (lane < (n & 3u)) atomicAdd(&sh.hist[d[body + lane]], 1u);
    cr_wave_sync();
    u64 c0 = ((u64)sh.hist[lane * 4u] << 8) | (lane * 4u), c1 = ((u64)sh.hist[lane * 4u + 1u] << 8) | (lane * 4u + 1u);
    u64 c2 = ((u64)sh.hist[lane * 4u + 2u] << 8) | (lane * 4u + 2u), c3 = ((u64)sh.hist[lane * 4u + 3u] << 8) | (lane * 4u + 3u);
    for (uint32_t k = 0; k < 10u; k++) {
        u64 a = c0 < c1 ? c0 : c1, b = c2 < c3 ? c2 : c3;
        u64 best = a < b ? a : b;
        for (int dlt = 32; dlt; dlt >>= 1) {
            u64 o = __shfl_xor(best, dlt);
            best = o < best ? o : best;
        }
        uint32_t v = (uint32_t)best & 0xffu;
        if (lane == 0) { sh.esc[k] = (uint8_t)v; sh.escmap[v] = (uint8_t)(k + 1u); }
        /* counter[esc] = -1: takes the value out of the running (no count reaches 2^32 - 1) */
        if (c0 == best) c0 = ~0ull;
        if (c1 == best) c1 = ~0ull;
        if (c2 == best) c2 = ~0ull;
        if (c3 == best) c3 = ~0ull;
    }
    cr_wave_sync();
}

/* cr-diccode.c:309 M_check_reverse_case on bytes given explicitly: b1 = s[i-1], b2 = s[i-2], b3 = s[i-3] */
CR_DEV bool cr_sentence_start(uint32_t i, uint32_t b1, uint32_t b2, uint32_t b3) {
    return i >= 3u && b1 == ' ' && (b2 == '.' || (b2 == ' ' && b3 == '.'));
}

/* dictionary_encode_imp, cr-diccode.c:285-362. Returns the bytes written at `out`. */
CR_DEV uint32_t cr_dict_encode_piece(const CrDict& D, const CrDictShared& sh, const uint8_t* s, uint32_t n, uint8_t* out) {
    const uint32_t lane = cr_lane();
    const uint32_t l1 = D.level1, wide = 256u - l1;
    const uint32_t lit_hi = D.nwords / wide, lit_lo = D.nwords % wide + l1;       /* code of word #dic_len */
    uint32_t o = 0;              /* output cursor (uniform) */
    uint32_t skip = 0;           /* positions below this are covered by an accepted word */
    for (uint32_t i0 = 0; i0 < n; i0 += CRGPU_WAVE) {
        const uint32_t p = i0 + lane;
        const bool live = p < n;
        uint32_t c = 0, cm1 = 0, cm2 = 0, cm3 = 0;
        if (live) {
            c = s[p];
            if (p >= 1) cm1 = s[p - 1];
            if (p >= 2) cm2 = s[p - 2];
            if (p >= 3) cm3 = s[p - 3];
        }
        /* trie walk for word starts (cr-diccode.c:305-308) */
        bool found = false;
        uint32_t j = p, id = 0, endc = 0;
        if (live && p > 0 && p + 2u * CR_DIC_WORD_MAX < n && cr_is_alpha(c) && !cr_is_alpha(cm1)) {
            uint32_t node = 0;
            for (;;) {
                uint32_t ch = s[j];
                if (ch >= 128u) break;
                uint32_t e = D.next[node * 128u + ch];
                if (e == 0u) break;
                node = e & ~CR_DIC_TERMINAL;
                if (e & CR_DIC_TERMINAL) { found = true; id = (uint32_t)D.ids[node]; endc = ch; break; }
                j++;
            }
        }
        /* a word swallows everything up to its terminator (i = j, cr-diccode.c:331): walk this
         * step's candidates in position order; one is accepted iff it is not already covered */
        const uint32_t skip_in = skip;
        u64 cand = cr_ballot(found);
        bool take = false;
        while (cand) {
            uint32_t l = (uint32_t)__builtin_ctzll(cand);
            cand &= cand - 1ull;
            if (i0 + l >= skip) {
                skip = cr_lane_get(j, l) + 1u;
                if (lane == l) take = true;
            }
        }
        /* covered = inside a word accepted in an earlier step, or at a lower lane of this one */
        uint32_t reach = cr_scan_max_incl(take ? j + 1u : 0u);
        uint32_t reach_before = cr_shift_up1(reach, 0u);
        const bool covered = live && (p < skip_in || p < reach_before);
        /* emit */
        uint32_t nout = 0, b0 = 0, b1 = 0, b2 = 0;
        if (live && !covered) {
            if (take) {
                bool flip = cr_is_upper(c) != cr_sentence_start(p, cm1, cm2, cm3);
                uint32_t tail = endc == ':' ? 4u : endc == ';' ? 3u : endc == ',' ? 2u : endc == '.' ? 1u : 0u;
                uint32_t e = sh.esc[(flip ? 5u : 0u) + tail];
                if (id < l1) { b0 = id; b1 = e; nout = 2; }
                else { b0 = id / wide; b1 = id % wide + l1; b2 = e; nout = 3; }
            } else if (sh.escmap[c]) {
                b0 = lit_hi; b1 = lit_lo; b2 = c; nout = 3;
            } else {
                b0 = c; nout = 1;
            }
        }
        uint32_t incl = cr_scan_incl(nout);
        uint32_t at = o + incl - nout;
        if (nout >= 1) out[at] = (uint8_t)b0;
        if (nout >= 2) out[at + 1] = (uint8_t)b1;
        if (nout >= 3) out[at + 2] = (uint8_t)b2;
        o += cr_lane_get(incl, 63);
    }
    if (lane < 4u) out[o + lane] = (uint8_t)(n >> (8u * lane));       /* cr-diccode.c:358-360 */
    return o + 4u;
}

/* dictionary_encode, cr-diccode.c:142-221. `out` must hold n + 1 bytes... plus scratch: the coded
 * form is built in `tmp` (capacity >= 3n + 64) and copied when it is smaller than the input. */
CR_DEV uint32_t cr_dict_encode_block(const CrDict& D, CrDictShared& sh, const uint8_t* src, uint32_t n,
                                     uint8_t* out, uint8_t* tmp) {
    const uint32_t lane = cr_lane();
    cr_dict_pick_escapes(src, n, sh);
    uint32_t o = 0, pos = 0;
    while (pos < n) {
        uint32_t a = pos + CR_DIC_PIECE < n ? CR_DIC_PIECE : n - pos; pos += a;
        uint32_t c = pos + CR_DIC_PIECE < n ? CR_DIC_PIECE : n - pos; pos += c;
        uint32_t s1 = cr_dict_encode_piece(D, sh, src + pos - c - a, a, tmp + o + 8u);
        uint32_t s2 = cr_dict_encode_piece(D, sh, src + pos - c, c, tmp + o + 8u + s1);
        if (lane < 4u) { tmp[o + lane] = (uint8_t)(s1 >> (8u * lane)); tmp[o + 4u + lane] = (uint8_t)(s2 >> (8u * lane)); }
        o += 8u + s1 + s2;
    }
    if (lane < 10u) tmp[o + lane] = sh.esc[lane];
    if (lane == 10u) tmp[o + 10u] = 1;
    o += 11u;
    cr_wave_sync();
    if (o >= n) {                                          /* cr-diccode.c:212-217 */
        for (uint32_t i = lane; i < n; i += CRGPU_WAVE) out[i] = src[i];
        if (lane == 0) out[n] = 0;
        return n + 1u;
    }
    for (uint32_t i = lane; i < o; i += CRGPU_WAVE) out[i] = tmp[i];
    return o;
}

/* 256-byte register window for reading a byte stream from its end */
struct CrBackWindow {
    const uint8_t* p;
    uint32_t size, base, word;
};
CR_DEV void cr_back_fill(CrBackWindow& w, uint32_t upto) {      /* make [upto-256, upto) resident */
    w.base = upto >= 256u ? upto - 256u : 0u;
    uint32_t o = w.base + cr_lane() * 4u, v = 0;
    if (o + 4u <= w.size) v = *reinterpret_cast<const cr_u32u*>(w.p + o);
    else for (uint32_t k = 0; k < 4; k++) if (o + k < w.size) v |= (uint32_t)w.p[o + k] << (8 * k);
    w.word = v;
}
CR_DEV uint32_t cr_back_at(CrBackWindow& w, uint32_t pos) {
    if (pos < w.base || pos >= w.base + 256u) cr_back_fill(w, pos + 1u);
    return cr_table_byte(w.word, pos - w.base);
}

/* dictionary_decode_imp, cr-diccode.c:364-425. Returns the piece's decoded size or 0xFFFFFFFF. */
CR_DEV uint32_t cr_dict_decode_piece(const CrDict& D, const CrDictShared& sh, const uint8_t* s, uint32_t n,
                                     uint8_t* out, uint32_t cap) {
    const uint32_t lane = cr_lane();
    const uint32_t l1 = D.level1, wide = 256u - l1;
    if (n < 4u) return 0xFFFFFFFFu;
    const uint32_t total = (uint32_t)s[n - 4] | ((uint32_t)s[n - 3] << 8) | ((uint32_t)s[n - 2] << 16) | ((uint32_t)s[n - 1] << 24);
    if (total > cap) return 0xFFFFFFFFu;
    CrBackWindow win; win.p = s; win.size = n;
    cr_back_fill(win, n - 4u);
    uint32_t w = total, r = n - 4u;
    /* the word decoded last (to the right): where it starts, what was written there, and the three
     * bytes to its left as they get written (cr-diccode.c:415-419 reads them back from memory) */
    uint32_t fix = 0xFFFFFFFFu, fix_byte = 0, seen = 0, n1 = 0, n2 = 0, n3 = 0;
#define CR_DIC_NOTE(byte_) do { if (seen == 0) n1 = (byte_); else if (seen == 1) n2 = (byte_); else if (seen == 2) n3 = (byte_); seen++; } while (0)
    while (w > 0) {
        if (r == 0) return 0xFFFFFFFFu;
        uint32_t ch = cr_back_at(win, --r);
        uint32_t kind = sh.escmap[ch];
        if (!kind) {
            w--;
            if (lane == 0) out[w] = (uint8_t)ch;
            CR_DIC_NOTE(ch);
            continue;
        }
        if (r == 0) return 0xFFFFFFFFu;
        uint32_t id = cr_back_at(win, --r);
        if (id >= l1) {
            if (r == 0) return 0xFFFFFFFFu;
            id = cr_back_at(win, --r) * wide + (id - l1);
            if (id == D.nwords) {                                        /* escaped literal */
                w--;
                if (lane == 0) out[w] = (uint8_t)ch;
                CR_DIC_NOTE(ch);
                continue;
            }
        }
        if (id >= D.nwords) return 0xFFFFFFFFu;
        const uint32_t len = D.wlen[id];
        if (len > w || len == 0) return 0xFFFFFFFFu;
        w -= len;
        uint32_t mine = 0;
        if (lane < len) {
            mine = D.words[id * CR_DIC_WORD_STRIDE + lane];
            if (lane == len - 1u) {                                      /* cr-diccode.c:405-410 */
                uint32_t t = kind > 5u ? kind - 5u : kind;
                if (t == 2u) mine = '.'; else if (t == 3u) mine = ','; else if (t == 4u) mine = ';'; else if (t == 5u) mine = ':';
            }
            if (lane == 0 && kind >= 6u) mine ^= 0x20u;
            out[w + lane] = (uint8_t)mine;
        }
        /* bytes to the left of the previous word, right to left: this word's tail */
        if (fix != 0xFFFFFFFFu) {
            for (uint32_t k = 0; k < 3u && seen < 3u && k < len; k++) CR_DIC_NOTE(cr_lane_get(mine, len - 1u - k));
            if (cr_sentence_start(fix, n1, n2, n3) && lane == 0) out[fix] = (uint8_t)(fix_byte ^ 0x20u);
        }
        fix = w; fix_byte = cr_lane_get(mine, 0); seen = 0; n1 = n2 = n3 = 0;
    }
    if (fix != 0xFFFFFFFFu && cr_sentence_start(fix, n1, n2, n3) && lane == 0) out[fix] = (uint8_t)(fix_byte ^ 0x20u);
#undef CR_DIC_NOTE
    return total;
}

/* dictionary_decode, cr-diccode.c:223-283. Returns the decoded size or 0xFFFFFFFF. */
CR_DEV uint32_t cr_dict_decode_block(const CrDict& D, CrDictShared& sh, const uint8_t* src, uint32_t n,
                                     uint8_t* out, uint32_t cap) {
    const uint32_t lane = cr_lane();
    if (n == 0) return 0xFFFFFFFFu;
    if (src[n - 1] == 0) {
        if (n - 1u > cap) return 0xFFFFFFFFu;
        for (uint32_t i = lane; i < n - 1u; i += CRGPU_WAVE) out[i] = src[i];
        return n - 1u;
    }
    if (n < 11u) return 0xFFFFFFFFu;
    for (uint32_t i = lane; i < 256u; i += CRGPU_WAVE) sh.escmap[i] = 0;
    cr_wave_sync();
    if (lane < 10u) { sh.esc[lane] = src[n - 11u + lane]; }
    cr_wave_sync();
    if (lane == 0) for (uint32_t k = 0; k < 10u; k++) sh.escmap[sh.esc[k]] = (uint8_t)(k + 1u);   /* later entries win, cr-diccode.c:376-378 */
    cr_wave_sync();
    uint32_t pos = 0, w = 0;
    while (pos + 11u < n) {
        if (pos + 8u > n) return 0xFFFFFFFFu;
        uint32_t a = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
        uint32_t c = (uint32_t)src[pos + 4] | ((uint32_t)src[pos + 5] << 8) | ((uint32_t)src[pos + 6] << 16) | ((uint32_t)src[pos + 7] << 24);
        pos += 8u;
        if ((u64)pos + a + c + 11u > n) return 0xFFFFFFFFu;
        uint32_t g = cr_dict_decode_piece(D, sh, src + pos, cr_uni(a), out + w, cap - w);
        if (g == 0xFFFFFFFFu) return g;
        w += g;
        g = cr_dict_decode_piece(D, sh, src + pos + a, cr_uni(c), out + w, cap - w);
        if (g == 0xFFFFFFFFu) return g;
        w += g;
        pos += a + c;
    }
    return w;
}

#endif

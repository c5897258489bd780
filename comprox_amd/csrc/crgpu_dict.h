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
 * Encoding is two kernels. What the trie says about a position — "a dictionary word starts here, it is word #id,
 * ends at j, with this terminator and this case" — depends on the data alone, not on what the encoder did with the
 * positions before (cr-diccode.c:303-310), so k_dict_match answers it for EVERY position of every block at once (one
 * thread per position; the walk is at most 22 dependent loads into a ~40 MB trie, i.e. Infinity Cache latency, which
 * only thousands of resident waves hide). k_dict_encode (one wave per block) then runs the part that IS sequential:
 * "a word swallows the positions it covers" (i = j, :331) resolved 64 positions per step in lane order, codes placed
 * with a DPP prefix sum. Decoding parses a piece from its end 64 coded bytes per step: which bytes start a token is a
 * three-state automaton run as a wave scan, word bodies are fetched per token lane, and the deferred sentence-case
 * fix-up (:415-419) of a step's words runs once the step to their left has written its bytes.
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

/* What the trie holds for position p of a piece (cr-diccode.c:303-322): 0, or CR_DM_FOUND | word number | span (bytes
 * up to and including the terminator) | terminator class | reverse-case flag */
#define CR_DM_FOUND     0x80000000u
#define CR_DM_ID(m)     ((m) & 0x7fffu)
#define CR_DM_SPAN(m)   (((m) >> 15) & 31u)
#define CR_DM_TAIL(m)   (((m) >> 20) & 7u)
#define CR_DM_FLIP(m)   (((m) >> 23) & 1u)
/* a word can only start at an alpha-after-non-alpha position that leaves 40 bytes behind it (cr-diccode.c:300,305) */
CR_DEV bool cr_dict_word_start(const uint8_t* s, uint32_t n, uint32_t p) {
    if (p == 0u || p + 2u * CR_DIC_WORD_MAX >= n) return false;
    return cr_is_alpha(s[p]) && !cr_is_alpha(s[p - 1u]);
}
/* the trie walk from such a position */
CR_DEV uint32_t cr_dict_match_at(const CrDict& D, const uint8_t* s, uint32_t n, uint32_t p) {
    (void)n;            /* (walks start at least 40 bytes in front of the piece's end and a word has at most CR_DIC_WORD_STRIDE bytes) */
    const uint32_t c = s[p], cm1 = s[p - 1u];
    uint32_t node = 0, j = p;
    for (;;) {
        const uint32_t ch = s[j];
        if (ch >= 128u) return 0u;
        const uint32_t e = D.next[node * 128u + ch];
        if (e == 0u) return 0u;
        node = e & ~CR_DIC_TERMINAL;
        if (e & CR_DIC_TERMINAL) {
            const uint32_t cm2 = p >= 2u ? s[p - 2u] : 0u, cm3 = p >= 3u ? s[p - 3u] : 0u;
            const uint32_t flip = (cr_is_upper(c) != cr_sentence_start(p, cm1, cm2, cm3)) ? 1u : 0u;
            const uint32_t tail = ch == ':' ? 4u : ch == ';' ? 3u : ch == ',' ? 2u : ch == '.' ? 1u : 0u;
            return CR_DM_FOUND | (uint32_t)D.ids[node] | ((j - p + 1u) << 15) | (tail << 20) | (flip << 23);
        }
        j++;
    }
}

/* dictionary_encode_imp, cr-diccode.c:285-362, given the trie's answers `match` for every position of the piece.
 * Returns the bytes written at `out`. */
CR_DEV uint32_t cr_dict_encode_piece(const CrDict& D, const CrDictShared& sh, const uint8_t* s, uint32_t n, const uint32_t* match, uint8_t* out) {
    const uint32_t lane = cr_lane();
    const uint32_t l1 = D.level1, wide = 256u - l1;
    const uint32_t lit_hi = D.nwords / wide, lit_lo = D.nwords % wide + l1;       /* code of word #dic_len */
    uint32_t o = 0;              /* output cursor (uniform) */
    uint32_t skip = 0;           /* positions below this are covered by an accepted word */
    if (n == 0u) { if (lane < 4u) out[lane] = 0; return 4u; }         /* (an empty piece: nothing may be read) */
    /* The byte and the trie's answer of every position are fetched FOUR steps ahead, and the order of a step's memory operations
     * is what makes that real (round 4; the kernel was parked 72 % of its time, one memory round trip per step). This target
     * counts loads and stores in one counter, in issue order: a load behind `if (p < n)` or a store behind `if (nout >= 2)` is a
     * branch, the compiler cannot know whether it was issued, and where the next step needs its operands it waits for
     * everything outstanding — the loads issued a moment ago included. So: the loads are unconditional (clamped index), four
     * register slots are reloaded for the step after next-but-two as soon as their values are taken out (no copies that would
     * wait for the newest loads), and a step's output bytes are stored at the START of the next step, before its reload:
     * whatever a step waits for is at least a step old. */
    struct Slot { uint32_t c, mt; };
    const auto fetch = [&](Slot& sl, uint32_t i0) __attribute__((always_inline)) {
        const uint32_t p = i0 + lane, q = p < n ? p : n - 1u;
        sl.c = s[q]; sl.mt = match[q];
    };
    uint32_t w_nout = 0, w_at = 0, w_b0 = 0, w_b1 = 0, w_b2 = 0;      /* the previous step's output, still to be stored */
    const auto step = [&](Slot& sl, uint32_t i0) __attribute__((always_inline)) {
        const uint32_t p = i0 + lane;
        const bool live = p < n;
        const uint32_t c = live ? sl.c : 0u, mt = live ? sl.mt : 0u;
        if (w_nout >= 1u) out[w_at] = (uint8_t)w_b0;
        if (w_nout >= 2u) out[w_at + 1u] = (uint8_t)w_b1;
        if (w_nout >= 3u) out[w_at + 2u] = (uint8_t)w_b2;
        fetch(sl, i0 + 4u * CRGPU_WAVE);
        const bool found = (mt & CR_DM_FOUND) != 0u;
        const uint32_t j = p + (found ? CR_DM_SPAN(mt) - 1u : 0u), id = CR_DM_ID(mt);
        /* a word swallows everything up to its terminator (i = j, cr-diccode.c:331): walk this
         * step's candidates in position order; one is accepted iff it is not already covered */
        const uint32_t skip_in = skip;
        /* Almost always no candidate of a step starts inside another one (words end at their terminator, candidates sit
         * at word starts): if none does even with ALL of them accepted, the greedy walk accepts all of them, and the
         * walk - a dependent chain of one lane hop per candidate - is not needed. */
        bool take = found && live;
        uint32_t reach = cr_scan_max_incl(take ? j + 1u : 0u);
        uint32_t reach_before = cr_shift_up1(reach, 0u);
        if (cr_ballot(take && (p < skip_in || p < reach_before)) == 0ull) {
            const uint32_t top = cr_lane_get(reach, 63);
            skip = top > skip_in ? top : skip_in;
        } else {
            u64 cand = cr_ballot(found && live);
            take = false;
            while (cand) {
                uint32_t l = (uint32_t)__builtin_ctzll(cand);
                cand &= cand - 1ull;
                if (i0 + l >= skip) {
                    skip = cr_lane_get(j, l) + 1u;
                    if (lane == l) take = true;
                }
            }
            /* covered = inside a word accepted in an earlier step, or at a lower lane of this one */
            reach = cr_scan_max_incl(take ? j + 1u : 0u);
            reach_before = cr_shift_up1(reach, 0u);
        }
        const bool covered = live && (p < skip_in || p < reach_before);
        /* emit */
        uint32_t nout = 0, b0 = 0, b1 = 0, b2 = 0;
        if (live && !covered) {
            if (take) {
                uint32_t e = sh.esc[(CR_DM_FLIP(mt) ? 5u : 0u) + CR_DM_TAIL(mt)];
                if (id < l1) { b0 = id; b1 = e; nout = 2; }
                else { b0 = id / wide; b1 = id % wide + l1; b2 = e; nout = 3; }
            } else if (sh.escmap[c]) {
                b0 = lit_hi; b1 = lit_lo; b2 = c; nout = 3;
            } else {
                b0 = c; nout = 1;
            }
        }
        uint32_t incl = cr_scan_incl(nout);
        w_nout = nout; w_at = o + incl - nout; w_b0 = b0; w_b1 = b1; w_b2 = b2;
        o += cr_lane_get(incl, 63);
    };
    Slot sa, sb, sc, sd;
    fetch(sa, 0u); fetch(sb, CRGPU_WAVE); fetch(sc, 2u * CRGPU_WAVE); fetch(sd, 3u * CRGPU_WAVE);
    for (uint32_t i0 = 0; i0 < n; i0 += 4u * CRGPU_WAVE) {            /* (a step past the end has no live position: it only stores what is pending) */
        cr_take_turns<0>(i0 >> 13);                                      /* (two of these waves on a SIMD take turns: crgpu_wave.h) */
        step(sa, i0); step(sb, i0 + CRGPU_WAVE); step(sc, i0 + 2u * CRGPU_WAVE); step(sd, i0 + 3u * CRGPU_WAVE);
    }
    if (w_nout >= 1u) out[w_at] = (uint8_t)w_b0;
    if (w_nout >= 2u) out[w_at + 1u] = (uint8_t)w_b1;
    if (w_nout >= 3u) out[w_at + 2u] = (uint8_t)w_b2;
    if (lane < 4u) out[o + lane] = (uint8_t)(n >> (8u * lane));       /* cr-diccode.c:358-360 */
    return o + 4u;
}

/* one wave copies n bytes, any alignment on both sides: 16 bytes per lane and round, four rounds in flight */
CR_DEV void cr_wave_copy(uint8_t* dst, const uint8_t* src, uint32_t n) {
    const uint32_t lane = cr_lane();
    const uint32_t lead = (uint32_t)((16u - ((u64)(uintptr_t)dst & 15u)) & 15u);
    const uint32_t head = lead < n ? lead : n;
    if (lane < head) dst[lane] = src[lane];
    const uint32_t body = (n - head) / 16u;
    uint32_t i = lane;
    for (; i + 3u * CRGPU_WAVE < body; i += 4u * CRGPU_WAVE) {
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, src + head + (u64)i * 16u, 16);
        __builtin_memcpy(&v1, src + head + (u64)(i + CRGPU_WAVE) * 16u, 16);
        __builtin_memcpy(&v2, src + head + (u64)(i + 2u * CRGPU_WAVE) * 16u, 16);
        __builtin_memcpy(&v3, src + head + (u64)(i + 3u * CRGPU_WAVE) * 16u, 16);
        *reinterpret_cast<uint4*>(dst + head + (u64)i * 16u) = v0;
        *reinterpret_cast<uint4*>(dst + head + (u64)(i + CRGPU_WAVE) * 16u) = v1;
        *reinterpret_cast<uint4*>(dst + head + (u64)(i + 2u * CRGPU_WAVE) * 16u) = v2;
        *reinterpret_cast<uint4*>(dst + head + (u64)(i + 3u * CRGPU_WAVE) * 16u) = v3;
    }
    for (; i < body; i += CRGPU_WAVE) {
        uint4 v;
        __builtin_memcpy(&v, src + head + (u64)i * 16u, 16);
        *reinterpret_cast<uint4*>(dst + head + (u64)i * 16u) = v;
    }
    const uint32_t done = head + body * 16u;
    if (lane < n - done) dst[done + lane] = src[done + lane];
}

/* dictionary_encode, cr-diccode.c:142-221. `out` must hold n + 1 bytes... plus scratch: the coded
 * form is built in `tmp` (capacity >= 3n + 64) and copied when it is smaller than the input. */
CR_DEV uint32_t cr_dict_encode_block(const CrDict& D, CrDictShared& sh, const uint8_t* src, uint32_t n, const uint32_t* match,
                                     uint8_t* out, uint8_t* tmp) {
    const uint32_t lane = cr_lane();
    cr_dict_pick_escapes(src, n, sh);
    uint32_t o = 0, pos = 0;
    while (pos < n) {
        uint32_t a = pos + CR_DIC_PIECE < n ? CR_DIC_PIECE : n - pos; pos += a;
        uint32_t c = pos + CR_DIC_PIECE < n ? CR_DIC_PIECE : n - pos; pos += c;
        uint32_t s1 = cr_dict_encode_piece(D, sh, src + pos - c - a, a, match + pos - c - a, tmp + o + 8u);
        uint32_t s2 = cr_dict_encode_piece(D, sh, src + pos - c, c, match + pos - c, tmp + o + 8u + s1);
        if (lane < 4u) { tmp[o + lane] = (uint8_t)(s1 >> (8u * lane)); tmp[o + 4u + lane] = (uint8_t)(s2 >> (8u * lane)); }
        o += 8u + s1 + s2;
    }
    if (lane < 10u) tmp[o + lane] = sh.esc[lane];
    if (lane == 10u) tmp[o + 10u] = 1;
    o += 11u;
    cr_wave_sync();
    if (o >= n) {                                          /* cr-diccode.c:212-217 */
        cr_wave_copy(out, src, n);
        if (lane == 0) out[n] = 0;
        return n + 1u;
    }
    cr_wave_copy(out, tmp, o);
    return o;
}

/* Which coded bytes start a token when a piece is read from its end (cr-diccode.c:386-396)? A token is 1 byte (a
 * literal), 2 (escape byte, 1-byte word number) or 3 (escape byte, 2-byte number) long, so reading position r in state
 * "k more bytes of the current token to skip" is a map on {0, 1, 2}: 0 -> (length of the token starting at r) - 1,
 * 1 -> 0, 2 -> 1. Maps are packed 2 bits per argument and composed with a wave scan. */
#define CR_DT_IDENT 36u                                   /* 0 -> 0, 1 -> 1, 2 -> 2 */
CR_DEV uint32_t cr_dt_compose(uint32_t g, uint32_t f) {   /* g after f */
    const uint32_t r0 = (g >> (2u * (f & 3u))) & 3u, r1 = (g >> (2u * ((f >> 2) & 3u))) & 3u, r2 = (g >> (2u * ((f >> 4) & 3u))) & 3u;
    return r0 | (r1 << 2) | (r2 << 4);
}
CR_DEV uint32_t cr_dt_scan_incl(uint32_t v) {             /* lane l: map of lanes 0..l applied in that order */
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)CR_DT_IDENT, (int)v, 0x111, 0xf, 0xf, false); v = cr_dt_compose(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)CR_DT_IDENT, (int)v, 0x112, 0xf, 0xf, false); v = cr_dt_compose(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)CR_DT_IDENT, (int)v, 0x114, 0xf, 0xf, false); v = cr_dt_compose(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)CR_DT_IDENT, (int)v, 0x118, 0xf, 0xf, false); v = cr_dt_compose(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)CR_DT_IDENT, (int)v, 0x142, 0xa, 0xf, false); v = cr_dt_compose(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)CR_DT_IDENT, (int)v, 0x143, 0xc, 0xf, false); v = cr_dt_compose(v, t);
    return v;
}

/* the deferred sentence-case fix-up (cr-diccode.c:415-419) of the words a step decoded: lane's word starts at piece
 * offset `at` (0xFFFFFFFF: none) with first byte `first`; the three bytes to its left have been written by now */
/* The decoder's output goes through a ring of LDS bytes per wave: a step's tokens are up to 64 x 24 bytes at byte-granular,
 * data-dependent places, and as global byte stores (one 64-lane store instruction per byte column, ~20 per step) they kept
 * the CU's one address path busy for the whole kernel (1.23 ms on the bench shard, six waves per CU queueing for it).
 * Bytes are written to the ring instead (index = output offset + the output's address modulo 4, so that aligned global
 * dwords are aligned ring dwords), the sentence-case fix of the previous step's words is applied there, and what is final —
 * everything from the top of this step's window upwards — leaves as whole dwords, coalesced. The ring holds two steps. */
#define CR_DD_RING 4096u
CR_DEV void cr_dd_lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
struct CrDictOut {
    uint8_t* ring;        /* LDS, CR_DD_RING bytes of this wave */
    uint8_t* out;         /* the piece's output */
    uint32_t shift;       /* (address of out) & 3 */
    CR_DEV uint32_t at(uint32_t off) const { return (off + shift) & (CR_DD_RING - 1u); }
    CR_DEV uint32_t get(uint32_t off) const { return ring[at(off)]; }
    CR_DEV void put(uint32_t off, uint32_t v) const { ring[at(off)] = (uint8_t)v; }
};
/* bytes [lo, hi) of the output from the ring to memory: whole dwords where the address allows, every lane of the wave calls */
CR_DEV void cr_dd_flush(const CrDictOut& o, uint32_t lo, uint32_t hi) {
    if (hi <= lo) return;
    const uint32_t lane = cr_lane();
    uint32_t head = (4u - ((lo + o.shift) & 3u)) & 3u;
    if (head > hi - lo) head = hi - lo;
    if (lane < head) o.out[lo + lane] = (uint8_t)o.get(lo + lane);
    const uint32_t a = lo + head, nd = (hi - a) >> 2;
    for (uint32_t i = lane; i < nd; i += CRGPU_WAVE) {
        const uint32_t off = a + 4u * i;
        *reinterpret_cast<uint32_t*>(o.out + off) = *reinterpret_cast<const uint32_t*>(o.ring + o.at(off));
    }
    const uint32_t t0 = a + 4u * nd;
    if (lane < hi - t0) o.out[t0 + lane] = (uint8_t)o.get(t0 + lane);
}
CR_DEV void cr_dict_fix_case(const CrDictOut& o, uint32_t at, uint32_t first) {
    if (at != 0xFFFFFFFFu && at >= 3u) {
        const uint32_t b1 = o.get(at - 1u), b2 = o.get(at - 2u), b3 = o.get(at - 3u);
        if (cr_sentence_start(at, b1, b2, b3)) o.put(at, first ^ 0x20u);
    }
}

/* dictionary_decode_imp, cr-diccode.c:364-425. Returns the piece's decoded size or 0xFFFFFFFF. 64 coded bytes per step,
 * read from the end: lane l looks at byte hi - 1 - l. */
CR_DEV uint32_t cr_dict_decode_piece(const CrDict& D, const CrDictShared& sh, const uint8_t* s, uint32_t n,
                                     uint8_t* out, uint32_t cap, uint8_t* ring) {
    const uint32_t lane = cr_lane();
    const uint32_t l1 = D.level1, wide = 256u - l1;
    if (n < 4u) return 0xFFFFFFFFu;
    const uint32_t total = (uint32_t)s[n - 4] | ((uint32_t)s[n - 3] << 8) | ((uint32_t)s[n - 2] << 16) | ((uint32_t)s[n - 1] << 24);
    if (total > cap) return 0xFFFFFFFFu;
    CrDictOut o;
    o.ring = ring; o.out = out; o.shift = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 3u);
    uint32_t w = total;                   /* output bytes still to produce: the next token ends at out[w - 1] */
    uint32_t flushed = total;             /* output bytes [flushed, total) are in memory, [w, flushed) in the ring */
    uint32_t hi = n - 4u;                 /* coded bytes [0, hi) not read yet */
    uint32_t state = 0;                   /* bytes at the top of [0, hi) that belong to a token of the previous step */
    uint32_t fix_at = 0xFFFFFFFFu, fix_first = 0;      /* this lane's word of the previous step, waiting for its left neighbours */
    /* the coded bytes are fetched one step ahead; a token's second and third byte (read backwards: s[r-1], s[r-2]) are
     * the next two lanes' bytes, the last two lanes take them from the step that follows */
    /* (round 4: the loads of a step are unconditional — clamped index, word 0 for a lane without a word —: behind `if` they are
     * branches, the compiler cannot count them and waits for the byte fetched "ahead" on the spot, a second memory round trip
     * per step beside the word's) */
    /* the window of step k + j: lane l looks at byte hi - 64 j - 1 - l */
    const auto window = [&](uint32_t hi_now, uint32_t j) __attribute__((always_inline)) -> uint32_t {
        const uint32_t back = j * CRGPU_WAVE + lane;
        return s[back < hi_now ? hi_now - 1u - back : 0u];
    };
    /* A step's tokens are PARSED a step early (round 4): which word a token names only depends on the coded bytes, not on where
     * the tokens start, so the parse of step k + 1 and the fetch of its words (a dependent load: the number comes out of the bytes)
     * run at the top of step k and have the whole step to arrive. The bytes themselves are fetched three steps ahead. */
    struct Tok { uint32_t ch, kind, id, tl, wlen; uint32_t dw[6]; bool bad, word; };
    const auto parse = [&](Tok& t, uint32_t hi_j, uint32_t ch_raw, uint32_t ch_raw_next) __attribute__((always_inline)) {
        const bool live = lane < hi_j;
        const uint32_t r = hi_j - 1u - lane;                                /* (meaningless when !live) */
        const uint32_t ch = live ? ch_raw : 0u;
        const uint32_t ch_next = lane + CRGPU_WAVE < hi_j ? ch_raw_next : 0u;
        const uint32_t ch1 = cr_shift_down1(ch, cr_lane_get(ch_next, 0));
        const uint32_t ch2 = cr_shift_down1(ch1, cr_lane_get(ch_next, 1));
        uint32_t kind = 0, id = 0, tl = 1;
        bool bad = false, word = false;
        if (live) {
            kind = sh.escmap[ch];
            if (kind) {
                if (r < 1u) bad = true;
                else {
                    id = ch1; tl = 2u;
                    if (id >= l1) {
                        if (r < 2u) bad = true;
                        else { id = ch2 * wide + (id - l1); tl = 3u; }
                    }
                    if (!bad) {
                        if (tl == 3u && id == D.nwords) word = false;       /* escaped literal: the escape byte itself */
                        else if (id >= D.nwords) bad = true;
                        else word = true;
                    }
                }
            }
        }
        /* the word itself and its length (a byte in the middle of a token that looks like an escape code asks for a word nobody
         * uses: its number is in range, the load is harmless; a lane without a word asks for word 0) */
        const uint32_t idw = word && !bad ? id : 0u;
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(D.words + idw * CR_DIC_WORD_STRIDE);
#pragma unroll
        for (int k = 0; k < 6; k++) t.dw[k] = wp[k];
        t.wlen = D.wlen[idw];
        t.ch = ch; t.kind = kind; t.id = id; t.tl = tl; t.bad = bad; t.word = word;
    };
    if (hi == 0u && w > 0u) return 0xFFFFFFFFu;                             /* no coded bytes at all: nothing to read (the windows below read s[0]) */
    uint32_t ch_n1 = window(hi, 1u), ch_n2 = window(hi, 2u);               /* (raw: a lane past the start reads byte 0 and is masked when used) */
    Tok T;
    parse(T, hi, window(hi, 0u), ch_n1);
    while (w > 0u) {
        cr_take_turns<0>(hi >> 12);                                         /* (two of these waves on a SIMD take turns: crgpu_wave.h) */
        if (hi == 0u) return 0xFFFFFFFFu;                                   /* ran out of coded bytes */
        const bool live = lane < hi;
        const uint32_t hi_1 = hi > CRGPU_WAVE ? hi - CRGPU_WAVE : 0u;
        const uint32_t ch_n3 = window(hi, 3u);
        Tok Tn;
        parse(Tn, hi_1, ch_n1, ch_n2);                                      /* the next step's tokens; their words load during this step */
        const uint32_t ch = T.ch, kind = T.kind, tl = T.tl;
        const bool word = T.word;
        bool bad = T.bad;
        const uint32_t wlen_id = T.wlen;
        uint32_t dw[6];
#pragma unroll
        for (int k = 0; k < 6; k++) dw[k] = T.dw[k];
        /* token starts: the automaton run over the lanes, entered in `state` */
        const uint32_t f = live ? ((tl - 1u) | (0u << 2) | (1u << 4)) : CR_DT_IDENT;
        const uint32_t incl_f = cr_dt_scan_incl(f);
        const uint32_t before = cr_shift_up1(incl_f, CR_DT_IDENT);
        const bool start = live && ((before >> (2u * state)) & 3u) == 0u;
        const uint32_t state_out = (cr_lane_get(incl_f, 63) >> (2u * state)) & 3u;
        /* output bytes per token, right to left */
        uint32_t len = 0;
        if (start) {
            len = 1u;
            if (word && !bad) { len = wlen_id; if (len == 0u) bad = true; }
        }
        const uint32_t incl = cr_scan_incl(len);
        const uint32_t excl = incl - len;
        const bool reached = start && excl < w;                            /* while(srcpos > 0) */
        if (cr_ballot(reached && (bad || incl > w))) return 0xFFFFFFFFu;   /* truncated token, unknown word, word longer than what is left */
        const uint32_t dst = w - incl;                                     /* where this token's bytes go (reached lanes) */
        uint32_t first = 0;
        if (reached && !word) o.put(dst, ch);
        const bool wr = reached && word;
        if (cr_ballot(wr)) {
            const uint32_t t = kind > 5u ? kind - 5u : kind;
            const uint32_t punct = t == 2u ? '.' : t == 3u ? ',' : t == 4u ? ';' : t == 5u ? ':' : 0u;   /* cr-diccode.c:405-410 */
            const uint32_t longest = cr_uni(cr_lane_get(cr_scan_max_incl(wr ? len : 0u), 63));
#pragma unroll
            for (int k = 0; k < (int)CR_DIC_WORD_STRIDE; k++) {
                if ((uint32_t)k >= longest) break;
                uint32_t c = (dw[k >> 2] >> (8 * (k & 3))) & 0xffu;
                if ((uint32_t)k == len - 1u && punct) c = punct;
                if (k == 0) { if (kind >= 6u) c ^= 0x20u; first = c; }      /* M_reverse_case */
                if (wr && (uint32_t)k < len) o.put(dst + (uint32_t)k, c);
            }
        }
        cr_dd_lds_order();
        /* the words of the previous step have their left neighbours now (every step writes >= 21 bytes unless it is the last) */
        cr_dict_fix_case(o, fix_at, fix_first);
        cr_dd_lds_order();
        fix_at = wr ? dst : 0xFFFFFFFFu;
        fix_first = first;
        /* everything from the top of this step's window upwards is final: it leaves in whole dwords (the up to three bytes
         * between the window's top and the next dword boundary of the output wait for the next step) */
        {
            uint32_t lo = w + ((4u - ((w + o.shift) & 3u)) & 3u);
            if (lo > flushed) lo = flushed;
            cr_dd_flush(o, lo, flushed);
            flushed = lo;
        }
        const uint32_t produced = cr_lane_get(incl, 63);
        w = cr_uni(produced >= w ? 0u : w - produced);
        hi = cr_uni(hi_1);
        T = Tn; ch_n1 = ch_n2; ch_n2 = ch_n3;
        state = cr_uni(state_out);
        cr_dd_lds_order();                                                  /* (the flush has read the ring before the next step writes it) */
    }
    cr_dict_fix_case(o, fix_at, fix_first);
    cr_dd_lds_order();
    cr_dd_flush(o, 0u, flushed);
    cr_wave_sync();
    return total;
}

/* dictionary_decode, cr-diccode.c:223-283. Returns the decoded size or 0xFFFFFFFF. */
CR_DEV uint32_t cr_dict_decode_block(const CrDict& D, CrDictShared& sh, const uint8_t* src, uint32_t n,
                                     uint8_t* out, uint32_t cap, uint8_t* ring /* LDS, CR_DD_RING bytes */) {
    const uint32_t lane = cr_lane();
    if (n == 0) return 0xFFFFFFFFu;
    if (src[n - 1] == 0) {
        if (n - 1u > cap) return 0xFFFFFFFFu;
        cr_wave_copy(out, src, n - 1u);
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
        uint32_t g = cr_dict_decode_piece(D, sh, src + pos, cr_uni(a), out + w, cap - w, ring);
        if (g == 0xFFFFFFFFu) return g;
        w += g;
        g = cr_dict_decode_piece(D, sh, src + pos + a, cr_uni(c), out + w, cap - w, ring);
        if (g == 0xFFFFFFFFu) return g;
        w += g;
        pos += a + c;
    }
    return w;
}

#endif

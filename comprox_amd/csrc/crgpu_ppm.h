/*
 * comprox_amd/csrc/crgpu_ppm.h — wave-parallel PPM model + range coder for gfx950.
 *
 * Mirrors, bit for bit, the arithmetic of the reference's
 *   range coder     /root/reference/src/cr-rangecoder.c:34-104
 *   order-2 node    /root/reference/src/cr-o2model.c:31-113
 *   PPM driver      /root/reference/src/cr-ppm.c:60-235
 * re-laid-out for one wavefront per datablock:
 *   - the current order-2 node lives in registers, four byte counts per lane (lane l holds
 *     symbols 4l..4l+3) plus one uniform word for the two flag symbols; a cumulative frequency is
 *     one packed-byte SAD per lane and one DPP sum, the decoder's symbol search one DPP scan;
 *   - the reference's cached 32-symbol group sums are not stored: they always equal the prefix
 *     sums of the counts (cr-o2model.c:50-52,56-62);
 *   - order-1 rows are handled the same way (256 u8 = one word per lane);
 *   - the order-3 predictor is an exact-keyed open-addressing table probed 64 slots at a time.
 * The range-coder state is wave-uniform; its bytes go through a 256-byte LDS stage.
 */
#ifndef CRGPU_PPM_H
#define CRGPU_PPM_H

#include "crgpu_wave.h"

/* Diagnostic build only (-DCRGPU_PROF): per-segment shader-clock sums of the coding step, kept in
 * registers and written to the block's stats slots 8..15. The product build compiles none of it. */
#ifdef CRGPU_PROF
struct CrProf { u64 last; u64 acc[8]; };
__device__ CrProf g_prof_dummy;
#define CR_PROF_ARG , CrProf& prof
#define CR_PROF_PASS , prof
#define CR_PROF_MARK(slot) do { u64 t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                                prof.acc[slot] += t_ - prof.last; prof.last = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CR_PROF_ARG
#define CR_PROF_PASS
#define CR_PROF_MARK(slot) do { } while (0)
#endif

/* ------------------------------------------------------------------ encoder byte sink */

struct CrSink {
    uint8_t* dst;        /* global */
    uint32_t n;          /* bytes emitted so far */
};

/* One byte store by lane 0, fire and forget: the coder emits ~0.3 bytes per input byte, so staging
 * them through LDS for wider stores only bought a flush routine inlined at every call site (the
 * encode kernel's hot loop no longer fitted the instruction cache). */
CR_DEV void cr_sink_put(CrSink& s, uint32_t byte) {
    if (cr_lane() == 0) s.dst[s.n] = (uint8_t)byte;
    s.n++;
}
CR_DEV void cr_sink_finish(CrSink& s) { (void)s; }

/* ------------------------------------------------------------------ range coder */

struct CrRc {
    uint32_t low, range, follow, carry, cache;
};

#define CR_RC_TOP     0x01000000u
#define CR_RC_NOCARRY 0xFF000000u

CR_DEV void cr_rc_init(CrRc& rc) { rc.low = 0; rc.range = 0xFFFFFFFFu; rc.follow = 0; rc.carry = 0; rc.cache = 0; }

/* cr-rangecoder.c:44-58 */
CR_DEV void cr_rc_shift(CrRc& rc, CrSink& out) {
    if (rc.low < CR_RC_NOCARRY || rc.carry) {
        cr_sink_put(out, (rc.cache + rc.carry) & 0xffu);
        while (rc.follow) { cr_sink_put(out, (rc.carry - 1u) & 0xffu); rc.follow--; }
        rc.cache = rc.low >> 24;
        rc.carry = 0;
    } else {
        rc.follow++;
    }
    rc.low <<= 8;
}

/* cr-rangecoder.c:60-70 */
CR_DEV void cr_rc_encode(CrRc& rc, uint32_t cum, uint32_t frq, uint32_t sum, CrSink& out) {
    uint32_t unit = cr_uni(rc.range / sum);
    uint32_t moved = rc.low + cum * unit;
    rc.carry += (moved < rc.low) ? 1u : 0u;
    rc.low = moved;
    rc.range = unit * frq;
    while (rc.range < CR_RC_TOP) {
        rc.range <<= 8;
        cr_rc_shift(rc, out);
    }
}

/* cr-rangecoder.c:72-79 */
CR_DEV void cr_rc_flush(CrRc& rc, CrSink& out) {
    for (int i = 0; i < 5; i++) cr_rc_shift(rc, out);
}

/* ------------------------------------------------------------------ decoder byte source */

struct CrSource {
    const uint8_t* src;  /* global, block payload */
    uint32_t size;       /* bytes available */
    uint32_t pos;        /* next byte */
    uint32_t base;       /* window covers [base, base+256) */
    uint32_t word;       /* lane l: bytes base+4l .. base+4l+3 */
};

CR_DEV void cr_source_fill(CrSource& s, uint32_t at) {
    s.base = at;
    uint32_t o = at + cr_lane() * 4u;
    uint32_t w = 0;
    if (o + 4u <= s.size) {
        w = *reinterpret_cast<const cr_u32u*>(s.src + o);
    } else {
        for (uint32_t j = 0; j < 4; j++)
            if (o + j < s.size) w |= (uint32_t)s.src[o + j] << (8 * j);
    }
    s.word = w;
    cr_drain_loads();
}
CR_DEV void cr_source_init(CrSource& s, const uint8_t* src, uint32_t size) {
    s.src = src; s.size = size; s.pos = 0;
    cr_source_fill(s, 0);
}
CR_DEV uint32_t cr_source_get(CrSource& s) {           /* reads past the end yield zero bytes */
    uint32_t rel = s.pos - s.base;
    if (rel >= 256u) { cr_source_fill(s, s.pos); rel = 0; }
    s.pos++;
    return cr_table_byte(s.word, rel);
}

/* cr-rangecoder.c:81-89 */
CR_DEV void cr_rc_dec_init(CrRc& rc, CrSource& in) {
    cr_rc_init(rc);
    for (int i = 0; i < 5; i++) rc.cache = (rc.cache << 8) + cr_source_get(in);
}
/* cr-rangecoder.c:101-104 */
CR_DEV uint32_t cr_rc_dec_target(CrRc& rc, uint32_t sum) {
    rc.range = cr_uni(rc.range / sum);
    return cr_uni(rc.cache / rc.range);
}
/* cr-rangecoder.c:91-99 */
CR_DEV void cr_rc_dec_consume(CrRc& rc, uint32_t cum, uint32_t frq, CrSource& in) {
    rc.cache -= cum * rc.range;
    rc.range *= frq;
    while (rc.range < CR_RC_TOP) {
        rc.cache = (rc.cache << 8) + cr_source_get(in);
        rc.range <<= 8;
    }
}

/* ------------------------------------------------------------------ PPM model state */

struct CrPpm {
    /* arena views */
    uint32_t* dir;
    uint32_t* nodes;
    u64*      o3;
    uint8_t*  o1;
    uint32_t  o3_mask;
    uint32_t  o3_shift;      /* 32 - log2(capacity) */
    uint32_t  max_nodes;
    /* model registers */
    uint32_t  ctx;           /* cr-ppm.h:40 */
    uint32_t  nnodes;
    /* the order-2 node currently held in registers */
    uint32_t  nd_key;        /* 0xFFFFFFFF = none */
    uint32_t  nd_idx;
    uint32_t  nd_w;          /* per lane: counts of symbols 4l..4l+3 */
    uint32_t  nd_x;          /* uniform: count(256) | count(257) << 8 */
    uint32_t  nd_dirty;
    uint32_t  nd_w0;         /* per lane: the word as it stands in memory (only changed words are written back) */
    uint32_t  nd_all;        /* 1: memory holds nothing valid for this node yet, write every word */
    uint32_t  gen;           /* generation tag of this block's nodes (16 bits, never 0) */
    /* the node that was in registers before the current one: its write-back is deferred until the
     * next step's loads have been issued (stores must not sit in front of loads in the in-order
     * vmcnt queue), and its content doubles as a one-entry cache for a context that comes straight back */
    uint32_t  defer;         /* 1 (encoder): use the victim registers; 0 (decoder): write a node back when it is left */
    uint32_t  vk_key;        /* 0xFFFFFFFF = none */
    uint32_t  vk_w, vk_w0, vk_x, vk_all, vk_dirty;
    /* what the last step stored, for patching loads that were issued before those stores */
    uint32_t  o3_ls;         /* slot of the last order-3 store (0xFFFFFFFF: none) */
    u64       o3_lv;         /* its value */
    uint32_t  lr_idx;        /* order-1 row last modified (0xFFFFFFFF: none) */
    uint32_t  lr_row;        /* per lane: that row's current word */
    uint32_t  row_here;      /* the current step's fetch already carries its order-1 row */
};

/* Loads for one coding step, issued as early as the context is known (software pipelining):
 * while they are in flight the previous step finishes its updates and stores. vmcnt retires in
 * order, so waiting for these loads never waits for stores issued after them. */
struct CrFetch {
    uint32_t ctx, valid, sw, key, h, row_idx;
    uint32_t with_row;       /* 1: fetch the order-1 row with every step (decoder: an escape would otherwise add a
                                round trip to the serial chain), 0: on escapes only (encoder) */
    uint32_t nw, nx;         /* node words (when the context's node is not the one in registers) */
    u64      v0;             /* first order-3 probe window */
    uint32_t row;            /* order-1 row word */
};

CR_DEV void cr_ppm_attach(CrPpm& m, uint8_t* arena, const CrArenaLayout& L, uint32_t o3_cap) {
    m.dir = reinterpret_cast<uint32_t*>(arena + L.off_dir);
    m.nodes = reinterpret_cast<uint32_t*>(arena + L.off_nodes);
    m.o3 = reinterpret_cast<u64*>(arena + L.off_o3);
    m.o1 = arena + L.off_o1;
    m.o3_mask = o3_cap - 1u;
    m.o3_shift = 32u - (uint32_t)__builtin_ctz(o3_cap);
    m.max_nodes = L.max_nodes;
}

/* ppm_model_free + ppm_model_init (cr-ppm.c:34-57) on the sparse tables */
CR_DEV void cr_ppm_reset(CrPpm& m) {
    /* Order-2 nodes are direct-indexed by the 16-bit context and validated by a generation tag, so
     * "free every node" is one increment. dir[0] keeps the slot's generation across launches; when
     * the 16-bit tag wraps, the node area is wiped once. */
    uint32_t g = cr_uni(m.dir[0]) + 1u;
    if (g > 0xffffu) {
        cr_fill(reinterpret_cast<uint8_t*>(m.nodes), (u64)CRGPU_NODE_AREA, 0u);
        g = 1u;
    }
    cr_wave_sync();
    if (cr_lane() == 0) m.dir[0] = g;
    m.gen = g;
    m.defer = 0;
    cr_fill(reinterpret_cast<uint8_t*>(m.o3), (u64)(m.o3_mask + 1u) * 8u, 0u);
    cr_fill(m.o1, 65536u, 0x01010101u);
    m.ctx = 0; m.nnodes = 0;
    m.nd_key = 0xFFFFFFFFu; m.nd_idx = 0; m.nd_w = 0; m.nd_x = 0; m.nd_dirty = 0; m.nd_w0 = 0; m.nd_all = 0;
    m.vk_key = 0xFFFFFFFFu; m.vk_w = 0; m.vk_w0 = 0; m.vk_x = 0; m.vk_all = 0; m.vk_dirty = 0;
    m.o3_ls = 0xFFFFFFFFu; m.o3_lv = 0; m.lr_idx = 0xFFFFFFFFu; m.lr_row = 0; m.row_here = 0;
}

/* persist mode: pick the model up where the previous call left it / leave it for the next call.
 * dir[0] = generation, dir[1] = context register, dir[2] = node count (the directory area is
 * otherwise unused since nodes are direct-indexed). */
CR_DEV void cr_ppm_resume(CrPpm& m) {
    m.gen = cr_uni(m.dir[0]);
    m.ctx = cr_uni(m.dir[1]);
    m.nnodes = cr_uni(m.dir[2]);
    m.defer = 0;
    m.nd_key = 0xFFFFFFFFu; m.nd_idx = 0; m.nd_w = 0; m.nd_x = 0; m.nd_dirty = 0; m.nd_w0 = 0; m.nd_all = 0;
    m.vk_key = 0xFFFFFFFFu; m.vk_w = 0; m.vk_w0 = 0; m.vk_x = 0; m.vk_all = 0; m.vk_dirty = 0;
    m.o3_ls = 0xFFFFFFFFu; m.o3_lv = 0; m.lr_idx = 0xFFFFFFFFu; m.lr_row = 0; m.row_here = 0;
}

CR_DEV void cr_ppm_push(CrPpm& m, uint32_t byte) { m.ctx = (m.ctx << 8) | (byte & 0xffu); }   /* cr-ppm.c:60-64 */

/* store the deferred (previous) node's changed words; its content stays valid as a cache entry */
CR_DEV void cr_victim_flush(CrPpm& m) {
    if (m.vk_dirty) {
        uint32_t* p = m.nodes + (u64)m.vk_key * CRGPU_NODE_WORDS;
        if (m.vk_all || m.vk_w != m.vk_w0) p[cr_lane()] = m.vk_w;
        if (cr_lane() == 0) m.nodes[CRGPU_FLAGS_WORD + m.vk_key] = m.vk_x | (m.gen << 16);
        m.vk_w0 = m.vk_w; m.vk_all = 0; m.vk_dirty = 0;
    }
}

CR_DEV void cr_node_writeback(CrPpm& m) {
    cr_victim_flush(m);
    if (m.nd_dirty) {
        uint32_t* p = m.nodes + (u64)m.nd_idx * CRGPU_NODE_WORDS;
        /* a coding step changes one or two counts: store only the words that differ from memory
         * (all of them for a new or a halved node) instead of the whole record */
        if (m.nd_all || m.nd_w != m.nd_w0) p[cr_lane()] = m.nd_w;
        if (cr_lane() == 0) m.nodes[CRGPU_FLAGS_WORD + m.nd_idx] = m.nd_x | (m.gen << 16);
        m.nd_w0 = m.nd_w; m.nd_all = 0;
        m.nd_dirty = 0;
    }
}

CR_DEV void cr_ppm_suspend(CrPpm& m) {
    cr_node_writeback(m);
    if (cr_lane() == 0) { m.dir[1] = m.ctx; m.dir[2] = m.nnodes; }
}

/* bring the node of the current context into registers (cr-ppm.c:104-107: allocate on demand).
 * `w`,`x` are the caller's early loads of the node's words; a stale generation means "never
 * allocated in this block" and yields o2_model_init's state (cr-o2model.c:31-41). */
CR_DEV void cr_node_install(CrPpm& m, uint32_t key, uint32_t w, uint32_t x) {
    if (!m.defer) {
        /* decoder: the next context is only known once this symbol is, there is nothing to overlap the
         * write-back with, and the extra register traffic of the victim costs more than it saves */
        cr_node_writeback(m);
        m.nd_key = key;
        m.nd_idx = key;
        if ((x >> 16) != m.gen) { m.nnodes++; m.nd_w = 0; m.nd_x = 0x0101u; m.nd_dirty = 1; m.nd_all = 1; }
        else { m.nd_w = w; m.nd_x = x & 0xffffu; m.nd_dirty = 0; m.nd_all = 0; }
        m.nd_w0 = m.nd_w;
        return;
    }
    /* the node two steps back must be in memory before its slot in the victim registers is reused */
    cr_victim_flush(m);
    const uint32_t back = (m.vk_key == key) ? 1u : 0u;       /* the context we just left comes straight back */
    const uint32_t ow = m.vk_w, ow0 = m.vk_w0, ox = m.vk_x;
    if (m.nd_key != 0xFFFFFFFFu) {
        m.vk_key = m.nd_key; m.vk_w = m.nd_w; m.vk_w0 = m.nd_w0; m.vk_x = m.nd_x; m.vk_all = m.nd_all; m.vk_dirty = m.nd_dirty;
    }
    m.nd_key = key;
    m.nd_idx = key;
    if (back) {
        /* (its loaded copy may predate the write-back that was just issued) */
        m.nd_w = ow; m.nd_x = ox; m.nd_w0 = ow0; m.nd_dirty = 0; m.nd_all = 0;
        return;
    }
    if ((x >> 16) != m.gen) {
        m.nnodes++;
        m.nd_w = 0;
        m.nd_x = 0x0101u;
        m.nd_dirty = 1;
        m.nd_all = 1;
    } else {
        m.nd_w = w;
        m.nd_x = x & 0xffffu;
        m.nd_dirty = 0;
        m.nd_all = 0;
    }
    m.nd_w0 = m.nd_w;
}

/* o2_model_update's halving pass, cr-o2model.c:54-71 */
CR_DEV void cr_node_halve(CrPpm& m) {
    m.nd_w = (m.nd_w >> 1) & 0x7f7f7f7fu;
    uint32_t singles = 1u + cr_sum(cr_count_ones_bytes(m.nd_w));
    uint32_t hit = ((m.nd_x & 0xffu) + 1u) >> 1;
    m.nd_x = hit | ((singles & 0xffu) << 8);
}

/* o2_model_update(node, sym, +1) for a byte symbol whose current count is `cur`; returns 1 if halved */
CR_DEV uint32_t cr_node_bump_byte(CrPpm& m, uint32_t sym, uint32_t cur) {
    if (cr_lane() == (sym >> 2)) m.nd_w += 1u << ((sym & 3u) * 8u);
    m.nd_dirty = 1;
    if (cur + 1u > 250u) { cr_node_halve(m); return 1u; }
    return 0u;
}
/* o2_model_update(node, 256, +1) */
CR_DEV uint32_t cr_node_bump_hit(CrPpm& m) {
    uint32_t v = ((m.nd_x & 0xffu) + 1u) & 0xffu;
    m.nd_x = (m.nd_x & 0xff00u) | v;
    m.nd_dirty = 1;
    if (v > 250u) { cr_node_halve(m); return 1u; }
    return 0u;
}
/* o2_model_update(node, 257, inc) with inc = +1 or -1 (u8 wrap like the reference) */
CR_DEV uint32_t cr_node_bump_esc(CrPpm& m, int inc) {
    uint32_t v = (((m.nd_x >> 8) & 0xffu) + (uint32_t)inc) & 0xffu;
    m.nd_x = (m.nd_x & 0x00ffu) | (v << 8);
    m.nd_dirty = 1;
    if (v > 250u) { cr_node_halve(m); return 1u; }
    return 0u;
}

/* ------------------------------------------------------------------ order-3 predictor */
/* entry: bit 63 valid, bits 53..32 key, bits 15..8 predicted byte, bits 3..0 confidence */

struct CrO3 {
    uint32_t slot;
    uint32_t key;
    uint32_t byte;
    uint32_t conf;
};

CR_DEV uint32_t cr_o3_key(uint32_t ctx) { return (ctx ^ (ctx >> 2)) & 0x3fffffu; }   /* cr-ppm.c:66 */

/* probe the table one 8-slot group (= one 64-byte line, lanes 0..7) at a time; `v0` is the
 * caller's early load of the home group */
#define CR_O3_GROUP 8u
CR_DEV uint32_t cr_o3_home(const CrPpm& m, uint32_t key) { return ((key * 2654435761u) >> m.o3_shift) & ~(CR_O3_GROUP - 1u); }
CR_DEV u64 cr_o3_load_group(const CrPpm& m, uint32_t first_slot) {
    u64 v = ~0ull;                                   /* lanes 8..63: neither empty nor a possible key */
    if (cr_lane() < CR_O3_GROUP) v = m.o3[(first_slot + cr_lane()) & m.o3_mask];
    return v;
}
CR_DEV void cr_o3_find(const CrPpm& m, CrO3& e, uint32_t h, u64 v0) {
    const u64 want = (u64)(e.key | 0x80000000u);
    u64 v = v0;
    for (uint32_t probe = 0;; probe += CR_O3_GROUP) {
        if (probe) { v = cr_o3_load_group(m, h + probe); cr_drain_loads(); }
        u64 hits = cr_ballot(v == 0ull || (v >> 32) == want);
        if (hits) {
            uint32_t first = (uint32_t)__builtin_ctzll(hits);
            u64 got = cr_lane_get64(v, first);
            e.slot = (h + probe + first) & m.o3_mask;
            e.byte = (uint32_t)(got >> 8) & 0xffu;       /* empty slot: byte 0, confidence 0 == */
            e.conf = (uint32_t)got & 0xfu;               /* the reference's zero-filled table   */
            return;
        }
    }
}
/* One symbol's model fetch with every independent load in flight together: directory word,
 * order-3 window and order-1 row go out first, the node follows as soon as the directory word is
 * back. */
/* Re-assert that the wave-uniform model/coder registers are uniform (one v_readfirstlane each):
 * lets the compiler keep them in SGPRs and branch on them with scalar branches. */
CR_DEV void cr_ppm_pin(CrPpm& m) {
    m.ctx = cr_uni(m.ctx); m.nnodes = cr_uni(m.nnodes); m.nd_key = cr_uni(m.nd_key);
    m.nd_idx = cr_uni(m.nd_idx); m.nd_x = cr_uni(m.nd_x); m.nd_dirty = cr_uni(m.nd_dirty); m.gen = cr_uni(m.gen); m.nd_all = cr_uni(m.nd_all);
    m.vk_key = cr_uni(m.vk_key); m.vk_x = cr_uni(m.vk_x); m.vk_all = cr_uni(m.vk_all); m.vk_dirty = cr_uni(m.vk_dirty);
}
CR_DEV void cr_rc_pin(CrRc& rc) {
    rc.low = cr_uni(rc.low); rc.range = cr_uni(rc.range); rc.follow = cr_uni(rc.follow);
    rc.carry = cr_uni(rc.carry); rc.cache = cr_uni(rc.cache);
}

CR_DEV void cr_ppm_issue(const CrPpm& m, CrFetch& F, uint32_t ctx) {
    F.ctx = ctx; F.valid = 1;
    F.key = ctx & 0xffffu;
    F.sw = (F.key != m.nd_key) ? 1u : 0u;
    F.nw = 0; F.nx = 0;
    if (F.sw) {
        const uint32_t* p = m.nodes + (u64)F.key * CRGPU_NODE_WORDS;
        F.nw = p[cr_lane()];
        F.nx = m.nodes[CRGPU_FLAGS_WORD + F.key];
    }
    F.h = cr_o3_home(m, cr_o3_key(ctx));
    F.v0 = cr_o3_load_group(m, F.h);
    F.row_idx = ctx & 0xffu;
    F.row = 0;
    if (F.with_row) F.row = reinterpret_cast<const uint32_t*>(m.o1 + (F.row_idx << 8))[cr_lane()];
}

/* consume the fetch for the current context (issuing it now if nobody prefetched it) */
CR_DEV void cr_ppm_take(CrPpm& m, CrFetch& F, CrO3& e, uint8_t*& rowp, uint32_t& row) {
    cr_ppm_pin(m);
    F.ctx = cr_uni(F.ctx); F.valid = cr_uni(F.valid);
    if (!F.valid || F.ctx != m.ctx) {
        /* vmcnt retires in order: a store issued between a load and its wait would put a full
         * write round trip on the critical path, so a pending write-back goes out BEFORE the loads */
        if ((m.ctx & 0xffffu) != m.nd_key) { if (m.defer) cr_victim_flush(m); else cr_node_writeback(m); }
        cr_ppm_issue(m, F, m.ctx);
    }
    F.valid = 0;
    F.sw = cr_uni(F.sw); F.key = cr_uni(F.key); F.h = cr_uni(F.h); F.row_idx = cr_uni(F.row_idx);
    if (F.sw) cr_node_install(m, F.key, F.nw, cr_uni(F.nx));
    /* the window may have been loaded before the previous step's order-3 store: patch that slot */
    u64 v = F.v0;
    if (cr_lane() < CR_O3_GROUP && ((F.h + cr_lane()) & m.o3_mask) == m.o3_ls) v = m.o3_lv;
    e.key = cr_o3_key(m.ctx);
    cr_o3_find(m, e, F.h, v);
    rowp = m.o1 + (F.row_idx << 8);
    row = F.row;
    m.row_here = F.with_row;
}
/* the order-1 row of the current context, needed by ~1 step in 5 (escapes): loaded on demand
 * instead of with every step's fetch (256 B saved per step); the row modified last is kept in
 * registers, which also covers a load overtaken by that row's store */
CR_DEV uint32_t cr_o1_row(const CrPpm& m, const uint8_t* rowp, uint32_t fetched) {
    const uint32_t idx = m.ctx & 0xffu;
    if (idx == m.lr_idx) return m.lr_row;
    if (m.row_here) return fetched;
    return reinterpret_cast<const uint32_t*>(rowp)[cr_lane()];
}
CR_DEV void cr_o3_store(CrPpm& m, const CrO3& e) {
    const u64 val = ((u64)(e.key | 0x80000000u) << 32) | (u64)(e.byte << 8) | (u64)e.conf;
    if (cr_lane() == 0) m.o3[e.slot] = val;
    m.o3_ls = e.slot; m.o3_lv = val;
}
/* ppm_update_o3(model, -1), cr-ppm.c:81-83 */
CR_DEV void cr_o3_hit(CrPpm& m, CrO3& e) {
    e.conf += (e.conf < 15u) ? 1u : 0u;
    cr_o3_store(m, e);
}
/* ppm_update_o3(model, c), cr-ppm.c:75-80 */
CR_DEV void cr_o3_miss(CrPpm& m, CrO3& e, uint32_t seen) {
    uint32_t c = e.conf;
    c = (uint32_t)(c > 1u) + (uint32_t)(c > 2u) + (uint32_t)(c > 4u) + (uint32_t)(c > 8u);
    if (c == 0u) { e.byte = seen; c = 1u; }
    e.conf = c;
    cr_o3_store(m, e);
}

/* ------------------------------------------------------------------ order-1 row helpers */

/* per-lane sum of the weights 8c-7 (cr-ppm.c:98) of the bytes of `row` selected by `keep`
 * (keep has 0xff in every selected byte) */
CR_DEV uint32_t cr_o1_weight_sum(uint32_t row, uint32_t keep) {
    uint32_t cnt = (uint32_t)__builtin_popcount(keep) >> 3;
    return 8u * cr_bytesum(row & keep) - 7u * cnt;
}
/* 0xff in every byte of w that is zero */
CR_DEV uint32_t cr_zero_bytes(uint32_t w) {
    uint32_t m = 0;
    if ((w & 0x000000ffu) == 0) m |= 0x000000ffu;
    if ((w & 0x0000ff00u) == 0) m |= 0x0000ff00u;
    if ((w & 0x00ff0000u) == 0) m |= 0x00ff0000u;
    if ((w & 0xff000000u) == 0) m |= 0xff000000u;
    return m;
}
/* candidates of the order-1 step: bytes absent from the order-2 node and different from the
 * prediction (cr-ppm.c:150-155) */
CR_DEV uint32_t cr_o1_keep(const CrPpm& m, uint32_t pred) {
    uint32_t keep = cr_zero_bytes(m.nd_w);
    if (cr_lane() == (pred >> 2)) keep &= ~(0xffu << ((pred & 3u) * 8u));
    return keep;
}
/* ppm_update_o1, cr-ppm.c:90-97; `row` is the lane's word of the row, returns the new word */
CR_DEV uint32_t cr_o1_bump(CrPpm& m, uint32_t row_idx, uint8_t* rowp, uint32_t row, uint32_t sym) {
    uint32_t cur = cr_table_byte(row, sym);
    if (cr_lane() == (sym >> 2)) row += 1u << ((sym & 3u) * 8u);
    if (cur + 1u >= 255u) {
        row -= (row >> 1) & 0x7f7f7f7fu;
        reinterpret_cast<uint32_t*>(rowp)[cr_lane()] = row;
    } else if (cr_lane() == (sym >> 2)) {
        reinterpret_cast<uint32_t*>(rowp)[cr_lane()] = row;
    }
    m.lr_idx = row_idx; m.lr_row = row;
    return row;
}

/* ------------------------------------------------------------------ ppm_encode, cr-ppm.c:103-167 */

/* `next_ctx`: the context of the NEXT ppm_encode call (the encoder knows it in advance), whose
 * loads are issued here, before this step's arithmetic and stores; pass has_next = 0 at the end. */
CR_DEV void cr_ppm_encode(CrPpm& m, CrRc& rc, uint32_t sym, CrSink& out, CrFetch& F, uint32_t next_ctx, uint32_t has_next CR_PROF_ARG) {
    CR_PROF_MARK(0);
    sym = cr_uni(sym);
    cr_rc_pin(rc);
    out.n = cr_uni(out.n);
    uint8_t* rowp; uint32_t row;
    CrO3 e;
    cr_ppm_take(m, F, e, rowp, row);
    CR_PROF_MARK(1);
    (void)has_next;
    cr_ppm_issue(m, F, cr_uni(next_ctx));             /* unconditional: a single definition site per loop pass */
    cr_victim_flush(m);                               /* the previous node's stores go out behind those loads */
    CR_PROF_MARK(2);
    const uint32_t pred = e.byte;
    const uint32_t pf = cr_table_byte(m.nd_w, pred);
    const uint32_t f_hit = m.nd_x & 0xffu, f_esc = (m.nd_x >> 8) & 0xffu;
    const uint32_t lane = cr_lane();

    if (sym == pred) {                                                   /* cr-ppm.c:119-126 */
        uint32_t bytes = cr_sum(cr_bytesum(m.nd_w));
        cr_rc_encode(rc, bytes - pf, f_hit, bytes + f_hit + f_esc - pf, out);
        CR_PROF_MARK(3);
        cr_node_bump_hit(m);
        cr_o3_hit(m, e);
        CR_PROF_MARK(4);
        return;
    }
    const uint32_t fs = cr_table_byte(m.nd_w, sym);
    uint32_t packed = cr_sum((cr_bytesum(m.nd_w & cr_mask_below(lane, sym)) << 16) | cr_bytesum(m.nd_w));
    const uint32_t below = packed >> 16, bytes = packed & 0xffffu;
    const uint32_t tot = bytes + f_hit + f_esc - pf;
    if (fs) {                                                            /* cr-ppm.c:129-139 */
        cr_rc_encode(rc, below - (sym > pred ? pf : 0u), fs, tot, out);
        CR_PROF_MARK(3);
        uint32_t halved = cr_node_bump_byte(m, sym, fs);
        if (!halved && fs + 1u == 2u) cr_node_bump_esc(m, -1);
    } else {                                                             /* cr-ppm.c:141-163 */
        cr_rc_encode(rc, bytes + f_hit - pf, f_esc, tot, out);
        uint32_t halved = cr_node_bump_esc(m, +1);
        row = cr_o1_row(m, rowp, row);
        uint32_t keep = cr_o1_keep(m, pred);
        uint32_t all = cr_sum(cr_o1_weight_sum(row, keep));
        uint32_t lo = cr_sum(cr_o1_weight_sum(row, keep & cr_mask_below(lane, sym)));
        uint32_t fo = cr_table_byte(row, sym) * 8u - 7u;
        cr_rc_encode(rc, lo, fo, all, out);
        CR_PROF_MARK(5);
        cr_o1_bump(m, m.ctx & 0xffu, rowp, row, sym);
        if (!halved) cr_node_bump_byte(m, sym, 0u);
    }
    cr_o3_miss(m, e, sym);
    CR_PROF_MARK(4);
}

/* ------------------------------------------------------------------ ppm_decode, cr-ppm.c:169-235 */

/* index (0..3) of the byte of `w` whose running interval holds `target`, given the count of
 * everything before this word; also returns that byte's lower bound */
CR_DEV uint32_t cr_pick_in_word(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t before,
                                uint32_t target, uint32_t& lower) {
    uint32_t a0 = before, a1 = a0 + w0, a2 = a1 + w1, a3 = a2 + w2;
    (void)w3;
    if (target < a1) { lower = a0; return 0; }
    if (target < a2) { lower = a1; return 1; }
    if (target < a3) { lower = a2; return 2; }
    lower = a3; return 3;
}

/* As soon as the symbol is known the next context is too (ctx<<8 | symbol, also right after an
 * escape byte), so the next step's loads go out before this step's updates; after a match copy the
 * caller's context differs and the fetch is simply re-issued. */
CR_DEV uint32_t cr_ppm_decode(CrPpm& m, CrRc& rc, CrSource& in, CrFetch& F CR_PROF_ARG) {
    CR_PROF_MARK(0);
    cr_rc_pin(rc);
    in.pos = cr_uni(in.pos); in.base = cr_uni(in.base);
    uint8_t* rowp; uint32_t row;
    CrO3 e;
    cr_ppm_take(m, F, e, rowp, row);
    CR_PROF_MARK(1);
    const uint32_t pred = e.byte;
    const uint32_t lane = cr_lane();
    const uint32_t f_hit = m.nd_x & 0xffu, f_esc = (m.nd_x >> 8) & 0xffu;

    /* counts with the predicted byte taken out (o2_model_get_decode_symbol's M_frq, cr-o2model.c:97) */
    uint32_t w = m.nd_w;
    if (lane == (pred >> 2)) w &= ~(0xffu << ((pred & 3u) * 8u));
    uint32_t incl = cr_scan_incl(cr_bytesum(w));
    const uint32_t bytes = cr_lane_get(incl, 63);
    CR_PROF_MARK(2);
    const uint32_t target = cr_rc_dec_target(rc, bytes + f_hit + f_esc);
    CR_PROF_MARK(3);

    uint32_t s, lower, frq;
    if (target < bytes) {
        uint32_t excl = incl - cr_bytesum(w);
        u64 owner = cr_ballot(excl <= target && target < incl);
        uint32_t ol = (uint32_t)__builtin_ctzll(owner);
        uint32_t ww = cr_lane_get(w, ol), before = cr_lane_get(excl, ol);
        uint32_t j = cr_pick_in_word(ww & 0xffu, (ww >> 8) & 0xffu, (ww >> 16) & 0xffu, ww >> 24, before, target, lower);
        s = ol * 4u + j;
        frq = (ww >> (8u * j)) & 0xffu;
    } else if (target < bytes + f_hit) {
        s = 256u; lower = bytes; frq = f_hit;
    } else {
        s = 257u; lower = bytes + f_hit; frq = f_esc;
    }
    CR_PROF_MARK(4);
    cr_rc_dec_consume(rc, lower, frq, in);                               /* cr-ppm.c:190-195 */
    CR_PROF_MARK(5);

    /* resolve an escape first (register-only work), so that ONE unconditional prefetch site follows:
     * a prefetch issued on some paths only would make the compiler copy the loaded registers at the
     * join, and a copy is a use — it would wait for the loads right there */
    uint32_t sym = s == 256u ? pred : s;
    uint32_t halved = 0;
    if (s == 257u) {                                                     /* cr-ppm.c:209-232 */
        halved = cr_node_bump_esc(m, +1);
        row = cr_o1_row(m, rowp, row);
        uint32_t keep = cr_o1_keep(m, pred);
        uint32_t mine = cr_o1_weight_sum(row, keep);
        uint32_t incl1 = cr_scan_incl(mine);
        uint32_t all = cr_lane_get(incl1, 63);
        uint32_t t1 = cr_rc_dec_target(rc, all);
        uint32_t excl1 = incl1 - mine;
        u64 owner = cr_ballot(excl1 <= t1 && t1 < incl1);
        uint32_t got = 0, lo = 0, fo = 1;
        if (owner) {
            uint32_t ol = (uint32_t)__builtin_ctzll(owner);
            uint32_t rw = cr_lane_get(row, ol), kp = cr_lane_get(keep, ol), before = cr_lane_get(excl1, ol);
            uint32_t q0 = (kp & 0x000000ffu) ? ((rw & 0xffu) * 8u - 7u) : 0u;
            uint32_t q1 = (kp & 0x0000ff00u) ? (((rw >> 8) & 0xffu) * 8u - 7u) : 0u;
            uint32_t q2 = (kp & 0x00ff0000u) ? (((rw >> 16) & 0xffu) * 8u - 7u) : 0u;
            uint32_t q3 = (kp & 0xff000000u) ? ((rw >> 24) * 8u - 7u) : 0u;
            uint32_t j = cr_pick_in_word(q0, q1, q2, q3, before, t1, lo);
            got = ol * 4u + j;
            fo = ((rw >> (8u * j)) & 0xffu) * 8u - 7u;
        }
        cr_rc_dec_consume(rc, lo, fo, in);
        sym = got;
    }
    sym = cr_uni(sym);
    cr_ppm_issue(m, F, (m.ctx << 8) | sym);             /* next step's loads, before this step's stores */

    if (s == 256u) {                                                     /* cr-ppm.c:199-201 */
        cr_node_bump_hit(m);
        cr_o3_hit(m, e);
    } else if (s < 256u) {                                               /* cr-ppm.c:203-207 */
        uint32_t hv = cr_node_bump_byte(m, s, frq);
        if (!hv && frq + 1u == 2u) cr_node_bump_esc(m, -1);
        cr_o3_miss(m, e, s);
    } else {
        cr_o1_bump(m, m.ctx & 0xffu, rowp, row, sym);
        if (!halved) cr_node_bump_byte(m, sym, 0u);
        cr_o3_miss(m, e, sym);
    }
    CR_PROF_MARK(6);
    return sym;
}

#endif

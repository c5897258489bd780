/*
 * comprox_amd/csrc/crgpu_device.h — shared device-side definitions for the gfx950 block codec.
 *
 * Execution model: ONE wavefront (64 lanes) owns one datablock from start to finish. A workgroup
 * is exactly one wave, so every "barrier" is free and all cross-lane traffic is DPP / readlane;
 * many such waves share a CU and hide each other's memory latency. Workgroups are persistent:
 * they pull block indices from an atomic ticket until the batch is drained, so the model arena
 * is sized by the number of resident workgroups, not by the number of blocks.
 *
 * Per-workgroup arena in HBM (all tables exact-keyed, so collision behaviour equals the
 * reference's, whose only collisions are in the keys themselves — SURVEY.md §7):
 *   dir     65536 x u32      last-two-bytes context -> node index + 1   (cr-ppm.h:38)
 *   nodes   up to 65536 x 272 B   258 u8 counts per node                 (cr-o2model.h:37-40)
 *   o3      cap_o3 x u64     22-bit key -> predicted byte + confidence   (cr-ppm.h:39)
 *   o1      256 x 256 u8                                                (cr-ppm.h:37)
 *   lzp8/4/2  cap_lz x u64   hashed 8/4/2-byte context -> last position  (ropmain/cr-matcher.c:35-50)
 *   lens    n x u8           LZP agreement length at every position
 */
#ifndef CRGPU_DEVICE_H
#define CRGPU_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

#define CRGPU_WAVE        64
/* order-2 nodes: the 256 byte counts of context c at c * 256 (two aligned 128-byte lines), the flag words {generation << 16
 * | count(257) << 8 | count(256)} of all contexts in an array of their own behind the counts: 32 neighbouring contexts share
 * a line there, so the flag traffic mostly stays in L2 (round 2; before, a node was 272 bytes: counts + flag word + pad,
 * three lines per fetch). The area keeps its old size. */
#define CRGPU_NODE_WORDS  64u
#define CRGPU_NODE_BYTES  (CRGPU_NODE_WORDS * 4u)
#define CRGPU_NODE_AREA   (65536u * 272u)
#define CRGPU_FLAGS_WORD  (65536u * CRGPU_NODE_WORDS)     /* first flag word, in words from the node area's start */
#define CRGPU_EMPTY64     0xFFFFFFFFFFFFFFFFull

/* Room an encoded block may need (crgpu_bound, include/crgpu.h); the assembling kernels test against the same value.
 * comprop: its one stream is tested against the input size after every token (ropmain/cr-coder.c:204), so header + n.
 * comprox / comprolz only test the MAIN stream (roxmain/cr-coder.c:273, rolzmain/cr-coder.c:233), the side streams
 * come on top. A side symbol costs at most log2(32 256) + 0.01 bits = 1.873 bytes (model_t keeps every count >= 1 and
 * the total <= 32 000 + one increment, cr-model.c:58-70; the coder's truncated range / total loses < 2^-9), and
 *   comprox: a match covers >= 6 bytes and carries 1 length + <= 2 distance symbols, >= 10 bytes from 3 distance
 *            digits on (<= 5), an escaped literal 1 symbol per occurrence of the LEAST frequent byte value (<= n / 256):
 *            <= (6/10 + 1/256) n symbols = 1.131 n bytes + 3 x 5 flush bytes            -> n + n / 4 + 128
 *   comprolz: a match covers >= 5 bytes and carries 2 symbols: <= (2/5 + 1/256) n symbols = 0.757 n bytes -> n - n / 8 + 128 */
static __host__ __device__ inline uint32_t cr_bound_rox(uint32_t n) { return 32u + n + n + n / 4u + 128u; }
static __host__ __device__ inline uint32_t cr_bound_rolz(uint32_t n) { return 16u + n + (n - n / 8u) + 128u; }

/* fixed head of the per-workgroup arena (make_layout keeps this order): dir, nodes, order-1 rows, direct order-3 table */
#define CRGPU_OFF_DIR     0u
#define CRGPU_OFF_SCRATCH 4096u                                   /* 1 KiB inside the directory area nobody reads */
#define CRGPU_OFF_NODES   262144u
#define CRGPU_OFF_O1      (CRGPU_OFF_NODES + CRGPU_NODE_AREA)
#define CRGPU_OFF_FLAGS   (CRGPU_OFF_NODES + 65536u * CRGPU_NODE_BYTES)
#define CRGPU_OFF_O3D     (CRGPU_OFF_O1 + 65536u)
/* The batched decoders (crgpu_rop5.h) keep an order-2 node as ONE 128-byte line at context * 128: up to 62 {symbol, count}
 * pairs (u16 = symbol << 8 | count, in symbol order; an unused pair reads 0xff00) and, in its last four bytes, the flag word
 * {generation << 16 | count(257) << 8 | count(256)}. A node that outgrows its line moves into a slot of 256 count bytes in a
 * small dense area (at most one slot per 63 coded symbols); its line then holds the slot number in pairs 0 / 1 and the
 * mark 0xffff in pair 61. Their arena's head: directory, lines, order-1 rows, direct order-3 table. */
#define CRGPU_LINE_BYTES    128u
#define CRGPU_LINE_PAIRS    62u
#define CRGPU_LINE_AREA     (65536u * CRGPU_LINE_BYTES)
#define CRGPU_DEC_OFF_O1    (CRGPU_OFF_NODES + CRGPU_LINE_AREA)
#define CRGPU_DEC_OFF_O3D   (CRGPU_DEC_OFF_O1 + 65536u)

struct CrArenaLayout {
    u64      stride;        /* bytes per workgroup                                   */
    u64      off_dir;       /* u32[65536]                                            */
    u64      off_nodes;     /* u32[max_nodes * CRGPU_NODE_WORDS]                     */
    u64      off_o3;        /* u64[cap_o3]                                           */
    u64      off_o1;        /* u8[65536]                                             */
    u64      off_o3d;       /* u16[1 << 22]: direct-indexed order-3 predictor (k_rop_decode_v3) */
    u64      off_lz8;       /* u64[cap_lz]                                           */
    u64      off_lz4;       /* u64[cap_lz]                                           */
    u64      off_lz2;       /* u64[cap_lz2]                                          */
    u64      off_lens;      /* u8[max_block]                                         */
    u64      off_cand;      /* u32[3][max_block]: LZP candidates per table (k_rop_lzp); comprolz decoder: ring / row links */
    u64      off_hist;      /* comprolz decoder: u32[8] per position, the eight entries of its ring before it (0 = not laid out) */
    u64      off_rox_cls;   /* u32[20 * (20 + max_block/25)]: hash-class heads (k_rox_match) */
    u64      off_rox_near;  /* u32[65536]: short-cache heads (k_rox_match) */
    u64      off_rolz_head; /* u32[262144]: newest position + 1 of every ROLZ ring (k_rolz_match, k_rolz_decode) */
    u64      off_keep;      /* u8[8192]: state kept between calls in persist mode (side models of comprox) */
    u64      off_side;      /* u8[3][side_stride]: side streams before concatenation (k_rox_encode) */
    u64      side_stride;
    u64      off_dense;     /* batched decoders: dense_slots x 256 count bytes for the nodes that outgrew their line */
    u64      node_area;     /* bytes at off_nodes: 65536 x 272 (one-wave coders) or 65536 x 128 (batched decoders' lines) */
    uint32_t dense_slots;
    uint32_t cap_o3;        /* power of two                                          */
    uint32_t cap_lz;        /* power of two                                          */
    uint32_t cap_lz2;       /* power of two (<= 131072: only 65536 distinct keys)    */
    uint32_t max_nodes;
    uint32_t max_block;
};

struct CrBatch {
    const uint8_t*  in;
    const u64*      in_off;
    const uint32_t* in_size;
    uint8_t*        out;
    const u64*      out_off;
    const uint32_t* out_cap;    /* decode only */
    uint32_t*       out_size;
    uint32_t        nblocks;
    uint32_t*       ticket;     /* zeroed before launch */
    uint8_t*        arena;
    uint32_t        fresh;      /* 1: reset_models() before every block */
    uint32_t        persist;    /* 1: single-slot mode of the reference-signature shims: the model outlives the call
                                   (fixed table capacities, context saved in the arena); fresh then says whether
                                   reset_models() was called since the previous block */
    uint8_t*        ev;         /* comprop chain encoder: per-block event scratch, block b at ev + b * ev_stride */
    u64             ev_stride;
    uint32_t        ev_cap;
    uint8_t*        side;       /* comprox chain encoder: per-block staging of the three side streams, block b at side + b * 3 * L.side_stride */
    uint8_t*        rox;        /* comprox encode: per-block match tables, block b at rox + b * rox_stride */
    u64             rox_stride;
    uint32_t        rox_limit;  /* match_limit: chain nodes examined per search (the reference's -m switch) */
    uint32_t        flexible;   /* flexible_parsing (the reference's -f switch; comprox and comprolz) */
    uint32_t        lzp_lds;    /* 1: the LDS pre-pass kernel (k_rop_lzp_lds / k_rolz_match_lds) has taken the blocks of up to 28 672 bytes,
                                 * the table-sweeping one (k_rop_lzp / k_rolz_match) skips them */
    uint32_t        links_lds;  /* 1: k_rop_links_lds has sorted the blocks of up to 28 672 events, k_rop_links skips them */
    uint32_t        o2_tickets; /* 1 (CRGPU_OPT_LZP_TABLES, the switch for the older kernels): k_rop_o2 walks its chains by tickets whatever the block's size */
    uint8_t*        pre_done;   /* encode: one byte per block, zeroed before the launch: which pre-pass kernel has done the block
                                 * (0 none yet = the table sweep takes it, 1 the 28 KiB LDS kernel, 2 the 64 KiB LDS kernel) */
    uint8_t*        lens;       /* encode: LZP agreement lengths, block b at lens + b * lens_stride (k_rop_lzp -> k_rop_encode) */
    u64             lens_stride;
    u64*            stats;      /* optional: 16 x u64 per block of phase stamps (100 MHz ticks, counts) */
};

#endif

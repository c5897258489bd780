/*
 * comprox_amd/csrc/crgpu_dec.h — what the three batched decoders (crgpu_rop5.h, crgpu_rox5.h, crgpu_rolz5.h)
 * share outside the assembly step: the per-block reset of the direct-indexed model tables and the
 * register window the coded bytes arrive through.
 *
 * Reference: /root/reference/src/cr-ppm.c:34-57 (ppm_model_free + ppm_model_init), src/cr-rangecoder.c:81-89
 * (range_decoder_init reads the coded bytes front to back).
 *
 * Tables: order-2 nodes direct-indexed by the 16-bit context — one 128-byte line of {symbol, count} pairs + flag word each,
 * 256 count bytes in a dense slot for the few that outgrow it (crgpu_device.h) — the flag word carries a generation tag (a stale
 * generation = "not allocated in this block" = the o2_model_init state); order-3 predictor direct-indexed by
 * the reference's 22-bit key (cr-ppm.c:66), u16 {byte, 4-bit generation, confidence}, wiped every 15th block
 * of a workgroup; order-1 rows dense, reset to 1 per block.
 */
#ifndef CRGPU_DEC_H
#define CRGPU_DEC_H

#include "crgpu_rop.h"

#define CR_O3D_ENTRIES (1u << 22)        /* cr-ppm.c:66: 22-bit key */

#define CR_LIKELY(x)   __builtin_expect(!!(x), 1)
#define CR_UNLIKELY(x) __builtin_expect(!!(x), 0)

/* ppm_model_free + ppm_model_init (cr-ppm.c:34-57) for the direct-indexed tables: node generation,
 * order-3 generation, order-1 rows = 1. Returns the node generation; o3gen by reference. */
CR_DEV uint32_t cr_v3_reset(uint8_t* arena, const CrArenaLayout& L, uint32_t& o3gen) {
    uint32_t* dir = reinterpret_cast<uint32_t*>(arena + L.off_dir);
    uint32_t g = cr_uni(dir[0]) + 1u;
    uint32_t g3 = cr_uni(dir[3]) + 1u;
    if (g > 0xffffu) {
        cr_fill(arena + L.off_nodes, L.node_area, 0u);
        g = 1u;
    }
    if (g3 > 15u) {
        cr_fill(arena + L.off_o3d, (u64)CR_O3D_ENTRIES * 2u, 0u);
        g3 = 1u;
    }
    cr_wave_sync();
    if (cr_lane() == 0) { dir[0] = g; dir[3] = g3; }
    cr_fill(arena + L.off_o1, 65536u, 0x01010101u);
    o3gen = g3;
    return g;
}

/* big-endian view of the coded bytes: lane l holds payload bytes [1 + base + 4l, +4), first byte in the
 * top bits; bytes past the end read as zero (CrSource does the same for the other decoders) */
CR_DEV uint32_t cr_v4_window(const uint8_t* payload, uint32_t size, uint32_t base) {
    const uint32_t o = 1u + base + cr_lane() * 4u;
    uint32_t v = 0;
    if (o + 4u <= size) v = *reinterpret_cast<const cr_u32u*>(payload + o);
    else for (uint32_t j = 0; j < 4; j++) if (o + j < size) v |= (uint32_t)payload[o + j] << (8 * j);
    cr_drain_loads();
    return __builtin_bswap32(v);
}

/* a wave-uniform pointer the compiler knows to be uniform (it came out of a vector load of one address) */
template <typename T>
CR_DEV T* cr_uni_ptr(T* p) {
    const u64 v = reinterpret_cast<u64>(p);
    return reinterpret_cast<T*>(((u64)cr_uni((uint32_t)(v >> 32)) << 32) | cr_uni((uint32_t)v));
}

#endif

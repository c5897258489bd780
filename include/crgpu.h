/*
 * include/crgpu.h — C-ABI of libcrgpu.so, the MI355X (gfx950) block codec behind comprox's
 * data_block_t entry points.
 *
 * Drop-in boundary. Every symbol below is plain C (pointers and sizes, no C++
 * or torch types) and names the reference interface it stands in for:
 *
 *   data_block_t, data_block_*    /root/reference/src/cr-datablock.h:35-46, cr-datablock.c:31-56
 *   reset_models/lzencode/lzdecode /root/reference/src/main.c:55-59 (extern decls the block loop
 *                                  links against), bodies in src/ropmain/cr-coder.c:73-83,119-292
 *                                  (comprop) and src/roxmain/cr-coder.c:88-114,153-318,390-526
 *   crgpu_*_blocks*               the batched form of the block loop src/main.c:174-206 (encode)
 *                                  and src/main.c:263-292 (decode) run with reset_models() before
 *                                  every block, i.e. independent datablocks
 *
 * The codec work happens in hand-written HIP kernels (comprox_amd/csrc/crgpu.hip + crgpu_*.h); there
 * is no CPU fallback: every entry point returns CRGPU_E_NODEVICE when no gfx950 device is usable.
 */
#ifndef CRGPU_H
#define CRGPU_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference's entry points are void; the batched API reports) ---- */
#define CRGPU_OK            0
#define CRGPU_E_NODEVICE   -1   /* no HIP device / wrong architecture / HIP runtime error       */
#define CRGPU_E_ARG        -2   /* NULL pointer, bad codec id, block larger than CRGPU_MAX_BLOCK */
#define CRGPU_E_NOMEM      -3   /* device or host allocation failed                             */
#define CRGPU_E_CORRUPT    -4   /* decode: a block header is inconsistent with its capacity     */

/* ---- codecs: which reference binary's lzencode/lzdecode is mirrored ---- */
#define CRGPU_CODEC_ROP     1   /* comprop: LZP + one PPM/range stream (src/ropmain/)           */
#define CRGPU_CODEC_ROX     2   /* comprox: LZ77 + PPM + three side streams (src/roxmain/)      */
#define CRGPU_CODEC_ROLZ    3   /* comprolz: ROLZ + PPM + one side stream (src/rolzmain/)       */

#define CRGPU_ROP_HEADER   20u  /* sizeof(block_header), src/ropmain/cr-coder.c:59-66           */
#define CRGPU_ROX_HEADER   32u  /* sizeof(block_header), src/roxmain/cr-coder.c:69-81           */
#define CRGPU_ROLZ_HEADER  16u  /* sizeof(block_header), src/rolzmain/cr-coder.c:63-71          */
#define CRGPU_MAX_BLOCK    (16u << 20)  /* largest datablock one call accepts (reference default
                                           -b16, src/main.c:62)                                 */

/* Worst-case encoded size of an n-byte block for `codec` (stored form: header + raw bytes). */
uint32_t crgpu_bound(int codec, uint32_t n);

/* ---- context: one per (process, GPU); owns the stream and the per-workgroup model arena ---- */
typedef struct crgpu_ctx crgpu_ctx;

int  crgpu_create(crgpu_ctx** out, int device);
void crgpu_destroy(crgpu_ctx* ctx);
const char* crgpu_last_error(const crgpu_ctx* ctx);   /* text of the last HIP failure, or ""   */
/* comprox codec: chain nodes examined per match search — the reference's -m switch
 * (match_limit, src/roxmain/cr-matcher.c:39, default 40). */
int  crgpu_rox_set_chain_limit(crgpu_ctx* ctx, uint32_t limit);
/* flexible_parsing (`-f`, src/roxmain/cr-matcher.c:32,253-289 and src/rolzmain/cr-matcher.c:30,143-167): the
 * parser of comprox / comprolz cuts a match where "this match + what follows it" prices best. Off by default. */
int  crgpu_set_flexible_parsing(crgpu_ctx* ctx, int on);
/* Route work to a caller-owned hipStream_t (NULL = the context's own stream). */
int  crgpu_set_stream(crgpu_ctx* ctx, void* hip_stream);

/*
 * Batched independent-datablock codec, DEVICE pointers (inputs already resident in HBM).
 * Block b reads  in[in_off[b] .. in_off[b]+in_size[b])  and writes its result at
 * out[out_off[b] ..], storing the produced byte count in out_size[b].
 *   encode: out must hold crgpu_bound(codec, in_size[b]) bytes per block.
 *   decode: out_cap[b] is the room at out_off[b]; a block that would overflow it, or whose
 *           header is malformed, yields out_size[b] = 0xFFFFFFFF and the call returns
 *           CRGPU_E_CORRUPT after all other blocks have been decoded.
 * Each block is coded with freshly reset models (reset_models() semantics).
 * All arrays (in_off, in_size, out_off, out_cap, out_size) are device arrays of nblocks entries.
 * The call enqueues on the context's stream and returns without synchronising unless
 * `sync` is non-zero.
 */
int crgpu_encode_blocks_dev(crgpu_ctx* ctx, int codec,
                            const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                            uint32_t nblocks, uint32_t max_block,
                            uint8_t* out, const uint64_t* out_off, uint32_t* out_size, int sync);
int crgpu_decode_blocks_dev(crgpu_ctx* ctx, int codec,
                            const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                            uint32_t nblocks, uint32_t max_block,
                            uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                            uint32_t* out_size, int sync);

/* Same contract with HOST pointers: stages through device buffers (H2D, kernels, D2H). */
int crgpu_encode_blocks(crgpu_ctx* ctx, int codec,
                        const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                        uint32_t nblocks, uint8_t* out, const uint64_t* out_off, uint32_t* out_size);
int crgpu_decode_blocks(crgpu_ctx* ctx, int codec,
                        const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                        uint32_t nblocks, uint8_t* out, const uint64_t* out_off,
                        const uint32_t* out_cap, uint32_t* out_size);

/* Timing of the most recent *_dev / host call on this context, from HIP events recorded on the
 * stream the kernels ran on: milliseconds spent in the dominant codec kernel. */
float crgpu_last_kernel_ms(const crgpu_ctx* ctx);
/* An encode call runs a matching pre-pass (k_rop_lzp / k_rox_match) and then the coding kernels;
 * crgpu_last_kernel_ms covers all of them, this returns the share of the pre-pass (-1 after a decode call). */
float crgpu_last_lzp_ms(const crgpu_ctx* ctx);
/* Every kernel of the most recent call, in launch order: names[i] (static strings) and ms[i] for
 * i < min(room, return value). Returns the number of kernels the call launched, -1 on error. */
int crgpu_last_stage_ms(const crgpu_ctx* ctx, const char** names, float* ms, int room);

/* ---- static-dictionary stage (reference: src/cr-diccode.c) ------------------------------------
 * crgpu_dict_create   == dictionary_load(text, 1) (src/cr-diccode.c:76-118): parses the dictionary
 *                        text produced by dicpick() (one word per line, NUL-terminated), builds the
 *                        reference's trie on the host and uploads it. One dictionary per file.
 * crgpu_dict_*_blocks == dictionary_encode (src/cr-diccode.c:142-221) / dictionary_decode
 *                        (:223-283) per datablock. encode: out must hold in_size[b] + 1 bytes per block
 *                        (raw copy + flag byte when substitution does not shrink the block);
 *                        decode: as crgpu_decode_blocks (out_cap, 0xFFFFFFFF on malformed input). */
typedef struct crgpu_dict crgpu_dict;
int  crgpu_dict_create(crgpu_ctx* ctx, const char* dictionary_text, crgpu_dict** out);
void crgpu_dict_destroy(crgpu_dict* dict);
int  crgpu_dict_words(const crgpu_dict* dict);          /* dictionary_load's return value */
int crgpu_dict_encode_blocks_dev(crgpu_ctx* ctx, crgpu_dict* dict,
                                 const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                                 uint32_t nblocks, uint32_t max_block,
                                 uint8_t* out, const uint64_t* out_off, uint32_t* out_size, int sync);
int crgpu_dict_decode_blocks_dev(crgpu_ctx* ctx, crgpu_dict* dict,
                                 const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                                 uint32_t nblocks, uint32_t max_block,
                                 uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                                 uint32_t* out_size, int sync);
int crgpu_dict_encode_blocks(crgpu_ctx* ctx, crgpu_dict* dict,
                             const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                             uint32_t nblocks, uint8_t* out, const uint64_t* out_off, uint32_t* out_size);
int crgpu_dict_decode_blocks(crgpu_ctx* ctx, crgpu_dict* dict,
                             const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                             uint32_t nblocks, uint8_t* out, const uint64_t* out_off,
                             const uint32_t* out_cap, uint32_t* out_size);

/* Diagnostics: when dev_stats (device memory, 16 x uint64 per block of the next batches) is set,
 * every block records 100 MHz phase stamps [start, lzp-reset, lzp-scan, lzp-done, model-ready,
 * coded] plus its order-2 node and token counts. NULL switches it off again. */
int crgpu_debug_stats(crgpu_ctx* ctx, uint64_t* dev_stats);

/* Wave-primitive self test used by tests/: in = 66 uint32 (64 lane values, mask limit, table
 * index), out = 448 uint32 (scan, sum, byte-sum, mask, previous-equal-lane, table byte,
 * agreement flags of the three previous-equal-lane implementations). */
int crgpu_selftest(crgpu_ctx* ctx, const uint32_t* in, uint32_t* out);

/* ---- drop-in data_block_t + codec entry points (reference signatures, void, global state) ---- */
typedef struct data_block_t {
    uint8_t* m_data;
    uint32_t m_size;
    uint32_t m_capacity;
} data_block_t;

void data_block_reserve(data_block_t* block, uint32_t size);
void data_block_resize(data_block_t* block, uint32_t size);
void data_block_add(data_block_t* block, uint8_t byte);
void data_block_destroy(data_block_t* block);

/* Select which reference binary the three shims below mirror (default CRGPU_CODEC_ROP) and on
 * which device they run (default 0). Not part of the reference; call before the first shim. */
int  crgpu_shim_config(int codec, int device);
int  crgpu_shim_rox_chain_limit(uint32_t limit);      /* -m for the comprox shims */
int  crgpu_shim_flexible_parsing(int on);             /* -f for the comprox / comprolz shims */

void reset_models(void);
void lzencode(data_block_t* ib, data_block_t* ob, int print_information);
void lzdecode(data_block_t* ib, data_block_t* ob, int print_information);

/* Static-dictionary entry points with the reference's signatures (src/cr-diccode.h:44-48,
 * src/cr-dicpick.h:40-42). dictionary_* run on the GPU through one process-wide crgpu_dict;
 * dicpick / dic_lcp_* are the once-per-file host passes and stay on the CPU. */
int  dictionary_load(const char* dicstr, int init_trie);
void dictionary_encode(data_block_t* i_block, data_block_t* o_block);
void dictionary_decode(data_block_t* i_block, data_block_t* o_block, FILE* fpout_sync);
void dicpick(FILE* fp, data_block_t* dic_block);
void dic_lcp_encode(data_block_t* dic_block);
void dic_lcp_decode(data_block_t* dic_block);

/* `-F` pre-filters with the reference's signature (src/cr-filter.h:35-38; called per datablock by the
 * block loop, src/main.c:183-185 before dictionary_encode and :284-286 after dictionary_decode).
 * E8/E9 call-target conversion inside PE / ELF i386 images and colour + row + column deltas of 24/32-bit
 * BMP pixel arrays; host C (stateful across the blocks of a file, one cheap pass). Returns 1 when a
 * filter touched the block (the block header's m_filt). The state that carries an image from one block to
 * the next lives for the process, as in the reference; crgpu_filter_reset() (new) clears it. */
#define FILTER_ENC 0
#define FILTER_DEC 1
int  filter_inplace(unsigned char* buf, uint32_t len, int en_de);
void crgpu_filter_reset(void);

#ifdef __cplusplus
}
#endif
#endif

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
#define CRGPU_MAX_BLOCK    (16u << 20)  /* largest datablock (reference default -b16, src/main.c:62). The
                                           codec stage accepts CRGPU_MAX_BLOCK + 1 bytes: dictionary_encode hands
                                           a block it could not shrink on as raw copy + flag byte
                                           (src/cr-diccode.c:208-217)                                        */

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
/* Diagnostic switches (the defaults are the product). crgpu_create reads the environment variable of the same
 * name ONCE; nothing on the call path looks at the environment. */
#define CRGPU_OPT_WG_PER_CU        1   /* resident workgroups per CU, 1..32 (default 16); each owns one model arena     */
#define CRGPU_OPT_ONE_WAVE_ENCODER 2   /* 1: the model-carrying one-wave coders (k_rop_encode / k_rox_encode /
                                          k_rolz_encode, what the shims run) instead of the kernel pipeline            */
#define CRGPU_OPT_ONE_WAVE_DECODER 3   /* 1: the model-carrying C++ decoders (k_rop_decode / k_rox_decode /
                                          k_rolz_decode) instead of the assembly step                                  */
#define CRGPU_OPT_LZP_GRID         4   /* at most this many workgroups for k_rop_lzp (0 = no limit)                    */
#define CRGPU_OPT_MATCH_GRID       5   /* the same for k_rox_match / k_rolz_match                                      */
#define CRGPU_OPT_LZP_TABLES       6   /* 1: the older kernels — every block through the table sweeps (k_rop_lzp, k_rop_links,
                                          the sweeps inside k_rox_match / k_rolz_match) instead of the sorts in LDS, and
                                          k_rop_o2 / k_rop_o3 walking their chains by tickets instead of slot ranges. Same
                                          bytes either way: the parity tests run both                                   */
#define CRGPU_OPT_STAGE_LOG        7   /* 1: keep the HIP-event boundaries of every kernel of every call until
                                          crgpu_stage_log_read folds them up (a timed loop then needs no event wait
                                          between its calls); 0: off, log dropped                                      */
#define CRGPU_OPT_DECODER_HELPER    8   /* 1: comprop's batched decoder as k_rop_decode_v5h — a workgroup of two waves per block, the
                                          second one preparing the escape step's order-1 sums (cr-ppm.c:209-211) from what the
                                          coder wave posts in LDS. Same bytes; measured slower than the one-wave kernel
                                          (DESIGN.md), kept for the parity tests and the record                              */
#define CRGPU_OPT_DECODER_LDS_NODES  9   /* 1 (default): comprop's batched decoder keeps its first dense order-2 nodes in LDS as well
                                          (k_rop_decode_v5, 8.7 KB of LDS per block in flight); 0: k_rop_decode_v5s, 272 bytes — for a
                                          caller that runs another context's encode calls beside this one's decodes, so that the
                                          decoder's workgroups leave room on a CU for a sorting kernel's 152 KB. Same bytes          */
int  crgpu_set_option(crgpu_ctx* ctx, int option, int value);

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

/* Device-side concatenation of a batch (`k_pack`; the write loop of src/main.c:198-205 as one contiguous run):
 * payload b = in[in_off[b] .. +in_size[b]) is copied to out[out_off[b] ..) where out_off is the exclusive sum of the
 * slots in front of it. with_headers != 0: every non-empty block is preceded by the container's packed
 * {u32 m_size, u8 m_filt, u8 m_prec} header (src/main.c:90-94; m_filt from filt[b] or 0, m_prec = prec) and empty
 * blocks are skipped, as `if(yb->m_size > 0)` does; out_off[b] is the PAYLOAD's position. total[0] = bytes laid out,
 * total[1] = blocks whose size is 0xFFFFFFFF (failed; skipped). All pointers are device pointers; out needs
 * sum(in_size) + 6 * nblocks bytes at most. */
int crgpu_pack_blocks_dev(crgpu_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                          uint32_t nblocks, const uint8_t* filt, int prec, int with_headers,
                          uint8_t* out, uint64_t* out_off, uint64_t* total, int sync);

/* The scan alone: out_off[b] = sum of sizes[k] for k < b, total[0] = the sum, total[1] = entries equal to 0xFFFFFFFF
 * (counted as 0). */
int crgpu_offsets_dev(crgpu_ctx* ctx, const uint32_t* sizes, uint32_t nblocks, uint64_t* out_off, uint64_t* total, int sync);

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

/* With CRGPU_OPT_STAGE_LOG on: for every kernel name launched since the log was switched on or last read, the summed
 * milliseconds (HIP events on the kernels' stream) and the number of launches; waits for the stream once and empties
 * the log (also when it fails). names / total_ms / launches have `room` entries; returns the number of distinct kernels
 * — a value above `room` means that many were seen and only the first `room` stored — or -1 on error. */
int crgpu_stage_log_read(crgpu_ctx* ctx, const char** names, float* total_ms, uint32_t* launches, int room);

/* Which match pre-pass took the blocks of the most recent encode call on this context: counts[0] = the table sweep in HBM
 * (k_rop_lzp / k_rox_match / k_rolz_match), counts[1] = the LDS sort for blocks of up to 28 672 bytes, counts[2] = the LDS sort
 * in groups by key for blocks of up to 65 537 bytes (a block whose keys do not split falls back to the sweep). They stand
 * for the reference's dense "last position with this key" tables (src/ropmain/cr-matcher.c:35-50 and siblings). */
int crgpu_last_prepass_paths(crgpu_ctx* ctx, uint32_t counts[3]);

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
/* Bytes dictionary_decode will produce for each dictionary-stage block (the sizes recorded at the end of its pieces,
 * src/cr-diccode.c:359-360, or n - 1 for the raw form): what a decoder needs to lay its output out before it runs.
 * out_size[b] = 0xFFFFFFFF for a malformed block. Device pointers. */
int crgpu_dict_decoded_sizes_dev(crgpu_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                                 uint32_t nblocks, uint32_t* out_size, int sync);
int crgpu_dict_encode_blocks(crgpu_ctx* ctx, crgpu_dict* dict,
                             const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                             uint32_t nblocks, uint8_t* out, const uint64_t* out_off, uint32_t* out_size);
int crgpu_dict_decode_blocks(crgpu_ctx* ctx, crgpu_dict* dict,
                             const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                             uint32_t nblocks, uint8_t* out, const uint64_t* out_off,
                             const uint32_t* out_cap, uint32_t* out_size);

/* ---- several GPUs of one node (csrc/crgpu_multi.hip) -------------------------------------------------
 * The block loop of src/main.c:174-206 (encode) / :263-292 (decode) for INDEPENDENT datablocks, sharded over G devices:
 * one host thread, context and stream per device; rank r codes the contiguous block range [r*ceil(n/G), (r+1)*ceil(n/G));
 * each rank lays its run of the container out on its device (k_pack) and copies it to its offset of the output. The one
 * exchange is the per-block size table: ncclAllGather (RCCL) over a communicator of the devices, or — when the device
 * list names a GPU twice, holds a single device, or CRGPU_MULTI_HOST_GATHER is given — through host memory (the ranks
 * are threads). A rank whose range is empty (nblocks < G, or the tail of ceil(n/G) ranges) contributes zeros. */
typedef struct crgpu_multi crgpu_multi;
#define CRGPU_MULTI_DICT        1   /* run the dictionary stage (needs crgpu_multi_set_dictionary)                   */
#define CRGPU_MULTI_PREC        2   /* encode: dictionary stage only, no codec (the reference's -p; needs _DICT)     */
#define CRGPU_MULTI_HEADERS     4   /* encode: write the container's 6-byte block headers, skip empty blocks         */
#define CRGPU_MULTI_HOST_GATHER 8   /* crgpu_multi_create: exchange the size table through host memory, not RCCL     */
#define CRGPU_MULTI_RCCL       16   /* crgpu_multi_create with ONE device: still form the (one-rank) RCCL communicator;
                                       without it a single device exchanges nothing and librccl is not loaded        */
#define CRGPU_MULTI_PINNED_OUT 32   /* crgpu_multi_create: *out of the encode / decode calls lies in the context's page-locked pool
                                       (grown as needed, kept for the context's life): the ranks copy their runs out as DMA at the
                                       link's rate instead of staging them into fresh pageable memory. The result then belongs
                                       to the context: valid until the next job on it, crgpu_multi_free is a no-op for it.
                                       What the write loop src/main.c:198-205 / :281-292 needs: the bytes once, to fwrite them */
int  crgpu_multi_create(crgpu_multi** out, const int* devices, int ndev, int flags);
int  crgpu_multi_reserve_output(crgpu_multi* m, uint64_t bytes);   /* CRGPU_MULTI_PINNED_OUT: page-lock the pool ahead of the first job */
void crgpu_multi_destroy(crgpu_multi* m);
const char* crgpu_multi_last_error(const crgpu_multi* m);
int  crgpu_multi_devices(const crgpu_multi* m);
int  crgpu_multi_uses_rccl(const crgpu_multi* m);       /* 1: the size table travels by ncclAllGather                 */
/* Wall-clock marks of `rank` in the most recent job, in seconds since the job was handed to the ranks: [1] input on the
 * device (H2D done), [2] stages done (dictionary stage, codec, k_pack), [3] sizes exchanged, [4] output allocated,
 * [5] this rank's run copied out (D2H done). Returns CRGPU_MULTI_TIMES. What `comp*-gpu -t` prints. */
#define CRGPU_MULTI_TIMES 6
int  crgpu_multi_timing(const crgpu_multi* m, int rank, double* seconds, int room);
/* The deadline of a job (crgpu_multi_encode_blocks / _decode_blocks), in seconds; default 120, $CRGPU_MULTI_DEADLINE_S read
 * at crgpu_multi_create, 0 = wait for ever. A rank that never reaches the size exchange (a kernel that does not end, a device
 * that dropped out, a peer missing from ncclAllGather) would leave the others waiting in the collective for ever; instead the
 * call returns CRGPU_E_NODEVICE when the deadline passes, crgpu_multi_last_error names the ranks that did not arrive and
 * where they are stuck, and the context is abandoned: every later call on it fails, crgpu_multi_destroy leaves its threads
 * alone — they may still be reading the job's input, which therefore has to stay valid (a process that gets this error is
 * expected to report it and end, as the reference's only error path does: `perror + return -1`, src/main.c:207-214). */
int  crgpu_multi_set_deadline(crgpu_multi* m, double seconds);
int  crgpu_multi_set_dictionary(crgpu_multi* m, const char* dictionary_text);   /* dictionary_load on every device    */
int  crgpu_multi_configure(crgpu_multi* m, uint32_t rox_chain_limit, int flexible);   /* -m (0 = keep) / -f          */
/* HOST pointers. Block b = in[in_off[b] .. +in_size[b]). *out is allocated by the library (release it with
 * crgpu_multi_free) and holds the blocks' results in block order, *out_total bytes; out_off[b] / out_size[b]
 * (optional, nblocks entries) = position and size of block b's result in it.
 *   encode: result = lzencode(dictionary_encode(block)) (flags pick the stages), behind its block header with
 *           CRGPU_MULTI_HEADERS (filt[b] = the header's m_filt, NULL = 0);
 *   decode: in = the coded blocks (payloads, without container headers), prec[b] != 0 marks a block that only went
 *           through the dictionary stage; result = dictionary_decode(lzdecode(block)). */
int  crgpu_multi_encode_blocks(crgpu_multi* m, int codec, int flags,
                               const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size, uint32_t nblocks,
                               const uint8_t* filt, uint8_t** out, uint64_t* out_total, uint64_t* out_off, uint32_t* out_size);
int  crgpu_multi_decode_blocks(crgpu_multi* m, int codec, int flags,
                               const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size, uint32_t nblocks,
                               const uint8_t* prec, uint8_t** out, uint64_t* out_total, uint64_t* out_off, uint32_t* out_size);
void crgpu_multi_free(void* p);
/* Host twins of the planning, used by the driver itself: rank's block range, and the exclusive sum of the slots
 * (6-byte header + payload for non-empty blocks when with_headers; entries of 0xFFFFFFFF count as 0). Returns the total. */
void crgpu_shard_range(uint32_t nblocks, int nranks, int rank, uint32_t* first, uint32_t* count);
uint64_t crgpu_container_offsets(const uint32_t* sizes, uint32_t nblocks, int with_headers, uint64_t* out_off);

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

/* Which reference binary the three shims below mirror. Without a call to crgpu_shim_config the codec follows the
 * front-end the library is linked into: every front-end defines `const char* cr_magic_header` (src/main.c:47;
 * src/roxmain/main.c:35 "...-comprox", src/rolzmain/main.c:35 "...-comprolz", src/ropmain/main.c:35 "...-comprop"),
 * the library holds a weak reference to it and picks the codec from its tail at the first shim call (comprop when
 * no such symbol exists). crgpu_shim_config (new, optional) overrides that and picks the device (default 0). */
int  crgpu_shim_config(int codec, int device);
int  crgpu_shim_codec(void);                          /* the codec the shims use (CRGPU_CODEC_*) */
/* Optional: bring the shims' device side up now (HIP runtime, context) rather than inside the first lzencode / lzdecode —
 * a tool calls it from the thread that is idle while dicpick() runs. Returns CRGPU_OK or the failure's code. */
int  crgpu_shim_prepare(void);
/* Optional hint: more blocks will follow WITHOUT reset_models() (the stock loop on a file of several blocks,
 * src/main.c:174-206). A block that starts from fresh models then runs on the model-carrying coder at once, so that its
 * successor continues from its models instead of rebuilding them from a second pass over its input. Same bytes either way. */
void crgpu_shim_expect_dependent_blocks(int on);
float crgpu_shim_last_kernel_ms(void);                /* kernel time of the most recent lzencode / lzdecode call, -1 if none */
/* Page-locked host memory for buffers handed to the host-pointer entry points (DMA at the link's rate instead of the
 * runtime's staging of pageable memory); release with crgpu_host_free. NULL when it cannot be had. */
void* crgpu_host_alloc(size_t bytes);
void  crgpu_host_free(void* p);
/* The switches the reference's front-ends assign directly (src/roxmain/main.c:88,99, src/rolzmain/main.c:87;
 * extern in src/roxmain/cr-matcher.h:52,56 and src/rolzmain/cr-matcher.h:43): data symbols of the library,
 * read by the shims at every call. */
extern int      flexible_parsing;                     /* -f, comprox / comprolz; default 0 */
extern uint32_t match_limit;                          /* -m, comprox; default 40 */
int  crgpu_shim_rox_chain_limit(uint32_t limit);      /* = match_limit = limit */
int  crgpu_shim_flexible_parsing(int on);             /* = flexible_parsing = on */
/* Failures of the void entry points below (no gfx950 device, HIP error, malformed block): recorded, handed to the
 * handler, and the entry point returns with an empty output block. Without a handler: message to stderr and
 * exit(EXIT_FAILURE) — there is no CPU fallback, and carrying on would write a broken file. */
typedef void (*crgpu_error_fn)(int code, const char* message, void* user);
void crgpu_shim_set_error_handler(crgpu_error_fn fn, void* user);
int  crgpu_shim_status(void);                         /* CRGPU_OK or the code of the last failing shim call */
const char* crgpu_shim_last_error(void);

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
/* filter_inplace reproduces the reference's bytes by default, including elf_i386_transform's never-reset byte counter
 * (src/filter_x86_elf.c:131-134), which converts every ELF image after the first of a run in a way its own FILTER_DEC
 * cannot undo. CRGPU_FILTER_RESTART_ELF (new, NOT the reference's format) restarts the counter per image so that the
 * transform round-trips; comp*-gpu -FF selects it and marks such blocks with m_filt = 2 so that the decoder follows. */
#define CRGPU_FILTER_REFERENCE   0
#define CRGPU_FILTER_RESTART_ELF 1
int  crgpu_filter_set_mode(int mode);
int  crgpu_filter_mode(void);
/* How many ELF images filter_inplace has converted with the reference's stale counter (CRGPU_FILTER_REFERENCE, a second or
 * later ELF image of the run) since the last crgpu_filter_reset(): a stream for which this is not 0 does NOT come back
 * from FILTER_DEC — neither here nor in the reference. comp*-gpu -F prints a warning when it happens (also under -q). */
int  crgpu_filter_lossy(void);

#ifdef __cplusplus
}
#endif
#endif

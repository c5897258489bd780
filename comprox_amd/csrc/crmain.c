/*
 * comprox_amd/csrc/crmain.c — container writer/reader and command line of `comprop-gpu`, plain C
 * over the C-ABI of libcrgpu.so.
 *
 * Re-states the reference's driver: cr_main (src/main.c:89-331), cr_process_arguments and the
 * banner/usage/magic strings of the comprop front-end (src/ropmain/main.c:35-104). File layout
 * (src/main.c:153-205):
 *     magic (no NUL) | u32 dict_csize | lzencode(dic_lcp_encode(dicpick(file)))
 *     then per block:  packed {u32 size, u8 filt, u8 prec} | payload
 * where payload = lzencode(dictionary_encode(block)), or dictionary_encode(block) alone with -p.
 *
 * Switches kept from the reference: -b<MB> block size (default 16), -p precompressor only, -q quiet,
 * -f flexible parsing (comprox-gpu, comprolz-gpu), -m<n> chain depth (comprox-gpu),
 * -F PE/ELF/BMP filters (crhost_filter.c; run per block in file order, also with -k). The reference's
 * DECODER loses the inverse filter: dictionary_decode() has already flushed the block to the output when
 * filter_inplace(FILTER_DEC) is called on the now empty buffer (src/main.c:281-286 with
 * src/cr-diccode.c:275-278), so `comprox -F e` followed by `d` does not reproduce the input. This tool
 * writes the same compressed file as the reference for -F and restores the input when it decodes.
 * New switch: -k<KiB> independent datablocks of that size, coded in ONE batched GPU call per stage
 * (reset_models() per block — the mode BASELINE.json's configs 2/3/5 describe). Files written
 * with -k carry format byte 2 in the magic so that the stock decoder refuses them instead of
 * mis-decoding. Without -k the loop is the stock one: the per-block entry points carry the models
 * from block to block exactly like the reference (reset_models() only after the dictionary blob),
 * so the output is the stock tool's byte for byte.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#include "../../include/crgpu.h"

/* -DCR_FRONTEND_ROX builds comprox-gpu (src/roxmain/main.c), -DCR_FRONTEND_ROLZ comprolz-gpu
 * (src/rolzmain/main.c), the default is comprop-gpu (src/ropmain/main.c) */
#ifdef CR_FRONTEND_ROLZ
#define CR_NAME  "comprolz-gpu"
#define CR_CODEC CRGPU_CODEC_ROLZ
#define CR_HEADER_BYTES CRGPU_ROLZ_HEADER
static const char MAGIC_STOCK[] = "\x1f\x9d\x01\x01::0.11.0-comprolz";    /* src/rolzmain/main.c:35 */
static const char MAGIC_INDEP[] = "\x1f\x9d\x01\x02::0.11.0-comprolz";
static const char BANNER[] =
    "============================================\n"
    " comprolz-gpu: rolz-ari compressor, MI355X  \n"
    " (format of comprolz by Zhang Li)           \n"
    "============================================\n";
#elif defined(CR_FRONTEND_ROX)
#define CR_NAME  "comprox-gpu"
#define CR_CODEC CRGPU_CODEC_ROX
#define CR_HEADER_BYTES CRGPU_ROX_HEADER
static const char MAGIC_STOCK[] = "\x1f\x9d\x01\x01::0.11.0-comprox";     /* src/roxmain/main.c:35 */
static const char MAGIC_INDEP[] = "\x1f\x9d\x01\x02::0.11.0-comprox";
static const char BANNER[] =
    "============================================\n"
    " comprox-gpu: lz77-ari compressor, MI355X   \n"
    " (format of comprox by Zhang Li)            \n"
    "============================================\n";
#else
#define CR_NAME  "comprop-gpu"
#define CR_CODEC CRGPU_CODEC_ROP
#define CR_HEADER_BYTES CRGPU_ROP_HEADER
static const char MAGIC_STOCK[] = "\x1f\x9d\x01\x01::0.11.0-comprop";     /* src/ropmain/main.c:35 */
static const char MAGIC_INDEP[] = "\x1f\x9d\x01\x02::0.11.0-comprop";
static const char BANNER[] =
    "============================================\n"
    " comprop-gpu: lzp-ari compressor, MI355X    \n"
    " (format of comprop by Zhang Li)            \n"
    "============================================\n";
#endif
static const char USAGE[] =
    "to compress:   " CR_NAME " [SWITCH] e [input] [output]\n"
    "to decompress: " CR_NAME "          d [input] [output]\n"
    "work with standard I/O streams if filenames are not given.\n"
    "\n"
    "optional SWITCH:\n"
    "   -b  set block size(MB), default = 16.\n"
    "   -k  independent blocks of this many KiB (1..16384), coded as one GPU batch.\n"
    "   -g  with -k: shard the blocks over this many GPUs (-G0,1,.. names them).\n"
    "   -p  work as a precompressor.\n"
    "   -F  use PE/ELF/BMP filter.\n"
    "   -FF the same with the ELF offset restarted per image (round-trips streams with several\n"
    "       ELF images; not the reference's bytes; blocks are marked m_filt = 2).\n"
#if defined(CR_FRONTEND_ROX) || defined(CR_FRONTEND_ROLZ)
    "   -f  use flexible parsing.\n"
#endif
#ifdef CR_FRONTEND_ROX
    "   -m  set maximum searching depth for LZ77 matching, default = 40.\n"
#endif
    "   -q  quiet mode.\n"
    "   -t  print a wall-clock breakdown of the run (JSON, stderr).\n";

static uint32_t opt_block = 16u * 1048576u;      /* cr_split_size, src/main.c:62 */
static uint32_t opt_indep_kib = 0;
static int opt_prec = 0;
static int opt_filt = 0;      /* cr_filt_enable, src/main.c:63 */
static int opt_flex = 0;      /* flexible_parsing */
static int opt_quiet = 0;
static int opt_times = 0;     /* -t: wall-clock breakdown of the run on stderr (one JSON object) */
static uint32_t opt_depth = 40;     /* match_limit, src/roxmain/cr-matcher.c:39 */
static int opt_devices[16];         /* -g<n> / -G<list>: the GPUs the -k batches are sharded over */
static int opt_ndev = 0;            /* 0: one GPU (device 0), size table exchanged in host memory */

#define SAY(...) do { if (!opt_quiet) fprintf(stderr, __VA_ARGS__); } while (0)

/* -t: named marks, seconds since main() started */
static struct { const char* name; double t; } g_marks[48];
static int g_nmarks = 0;
static double g_t0 = 0.0;
static double wall(void) { struct timeval tv; gettimeofday(&tv, NULL); return (double)tv.tv_sec + (double)tv.tv_usec * 1e-6; }
static void mark(const char* name) { if (opt_times && g_nmarks < 48) { g_marks[g_nmarks].name = name; g_marks[g_nmarks].t = wall() - g_t0; g_nmarks++; } }
static void mark_at(const char* name, double t) { if (opt_times && g_nmarks < 48) { g_marks[g_nmarks].name = name; g_marks[g_nmarks].t = t; g_nmarks++; } }
static void print_marks(void) {
    if (!opt_times) return;
    fprintf(stderr, "{\"tool\": \"%s\", \"seconds_since_start\": {", CR_NAME);
    for (int i = 0; i < g_nmarks; i++) fprintf(stderr, "%s\"%s\": %.4f", i ? ", " : "", g_marks[i].name, g_marks[i].t);
    fprintf(stderr, "}}\n");
}

#pragma pack(push, 1)
typedef struct { uint32_t m_size; uint8_t m_filt; uint8_t m_prec; } block_head_t;   /* src/main.c:90-94 */
#pragma pack(pop)

/* src/ropmain/main.c:59-104 */
static int process_arguments(int argc, char** argv) {
    while (argc >= 2 && argv[1][0] == '-') {
        const char* a = argv[1];
        switch (a[1]) {
            case 'b': { int mb = atoi(a + 2); if (mb <= 0 || mb > 16) goto bad; opt_block = (uint32_t)mb * 1048576u; break; }
            case 'k': { int kb = atoi(a + 2); if (kb <= 0 || kb > 16384) goto bad; opt_indep_kib = (uint32_t)kb; break; }
            case 'g': { int g = atoi(a + 2); if (g <= 0 || g > 16) goto bad; opt_ndev = g; for (int i = 0; i < g; i++) opt_devices[i] = i; break; }
            case 'G': {                                          /* explicit device list, e.g. -G0,2,4,6 (a GPU may be named twice) */
                const char* q = a + 2; opt_ndev = 0;
                while (*q) {
                    char* end; long d = strtol(q, &end, 10);
                    if (end == q || d < 0 || opt_ndev >= 16) goto bad;
                    opt_devices[opt_ndev++] = (int)d;
                    q = *end == ',' ? end + 1 : end;
                    if (*end && *end != ',') goto bad;
                }
                if (!opt_ndev) goto bad;
                break;
            }
            case 'p': if (a[2]) goto bad; opt_prec = 1; break;
            case 'q': if (a[2]) goto bad; opt_quiet = 1; break;
            case 't': if (a[2]) goto bad; opt_times = 1; break;
            case 'F':                                            /* -F: the reference's filters; -FF: with the ELF counter restarted per image */
                if (a[2] == 'F' && !a[3]) { opt_filt = 2; if (crgpu_filter_set_mode(CRGPU_FILTER_RESTART_ELF) != CRGPU_OK) goto bad; break; }
                if (a[2]) goto bad;
                opt_filt = 1; break;
#if defined(CR_FRONTEND_ROX) || defined(CR_FRONTEND_ROLZ)
            case 'f': if (a[2]) goto bad; opt_flex = 1; break;           /* src/roxmain/main.c:86-91, src/rolzmain/main.c:84-89 */
#endif
#ifdef CR_FRONTEND_ROX
            case 'm': { int d = atoi(a + 2); if (d <= 0) goto bad; opt_depth = (uint32_t)d; break; }
#endif
            default: bad: fprintf(stderr, "invalid switch '%s'.\n", a); return 0;
        }
        memmove(argv + 1, argv + 2, (size_t)(argc - 2) * sizeof(char*));
        argc--;
    }
    return argc;
}

static FILE* spool_stdin(void) {                 /* src/main.c:141-150: two passes need a seekable file */
    FILE* t = tmpfile();
    char buf[65536];
    size_t n;
    if (!t) return NULL;
    while ((n = fread(buf, 1, sizeof buf, stdin)) > 0) fwrite(buf, 1, n, t);
    rewind(t);
    return t;
}

static int die(const char* what) { perror(what); return -1; }

/* ---- encode ------------------------------------------------------------------------------- */

/* ---- the sharded GPU path of -k and what is prepared for it while the dictionary is picked ------------------ */
static crgpu_multi* g_mg;           /* created early (contexts, streams, RCCL), the dictionary is set when it exists */
static uint8_t* g_slice;            /* the slice of the input a job works on: page-locked when that can be had */
static size_t g_slice_cap;
static int g_slice_pinned;
static uint64_t g_first_bytes;      /* bytes of the first slice already read into g_slice (by pread, behind dicpick's back) */
static int g_first_read;
static int g_src_fd = -1;
static uint64_t g_src_size;

static uint64_t g_pool_hint;         /* bytes to page-lock for the results ahead of the first job (0: let the first job do it) */
/* a guess is never allowed more than a quarter of the memory that is free right now: page-locking costs about a second per
 * GiB on a host that is short of it, and a reservation that fails is simply skipped (the job then allocates what it needs) */
static uint64_t pool_hint_capped(uint64_t want) {
    const long pages = sysconf(_SC_AVPHYS_PAGES), psize = sysconf(_SC_PAGESIZE);
    if (pages > 0 && psize > 0) { const uint64_t quarter = (uint64_t)pages * (uint64_t)psize / 4u; if (want > quarter) want = quarter; }
    return want;
}
static crgpu_multi* create_multi(void) {
    static const int one[1] = {0};
    crgpu_multi* mg = NULL;
    /* without -g: one GPU and nothing to exchange, so RCCL is not even loaded; -g1 / -G<one device> asks for the sharded
     * path by name: it keeps its (one-rank) RCCL communicator */
    /* Results in the context's page-locked pool (every slice's run is written to the file before the next job starts) only
     * for files of several slices: page-locking ~100 MB costs ~50 ms once (profiles/r04i_cli_breakdown.txt), a staged copy of
     * one slice's run into pageable memory ~8 ms — a one-slice run is better off without the pool, a long one with it. */
    const int pooled = g_pool_hint ? CRGPU_MULTI_PINNED_OUT : 0;
    int rc = opt_ndev ? crgpu_multi_create(&mg, opt_devices, opt_ndev, CRGPU_MULTI_RCCL | pooled)
                      : crgpu_multi_create(&mg, one, 1, CRGPU_MULTI_HOST_GATHER | pooled);
    if (rc != CRGPU_OK) { fprintf(stderr, "no usable MI355X (gfx950) device for the requested GPU list (%d); there is no CPU fallback\n", rc); return NULL; }
    rc = crgpu_multi_configure(mg, opt_depth, opt_flex);
    if (rc != CRGPU_OK) { fprintf(stderr, "GPU setup failed (%d): %s\n", rc, crgpu_multi_last_error(mg)); crgpu_multi_destroy(mg); return NULL; }
    /* the pool the results come back in, page-locked now (this runs beside dicpick / the dictionary blob's decode): an encoder's
     * slice shrinks, a decoder's grows — a guess; a job that needs more grows the pool itself */
    if (g_pool_hint) (void)crgpu_multi_reserve_output(mg, g_pool_hint);
    return mg;
}

static void* create_multi_main(void* unused) { (void)unused; g_mg = create_multi(); return NULL; }

static uint64_t slice_blocks_for(uint64_t nb_all, uint32_t block) {
    /* the file goes through in slices of whole blocks (at most 65 536 of them or 1 GiB, like decode_batched): a rank
     * holds about four times its share of a slice in HBM, the host one slice and its output */
    uint64_t per = ((uint64_t)1 << 30) / block;
    if (per > 65536u) per = 65536u;
    if (per == 0) per = 1;
    return nb_all < per ? nb_all : per;
}

static int slice_buffer(size_t bytes) {
    if (g_slice && g_slice_cap >= bytes) return 0;
    if (g_slice) { if (g_slice_pinned) crgpu_host_free(g_slice); else free(g_slice); }
    g_slice = (uint8_t*)crgpu_host_alloc(bytes ? bytes : 1u);
    g_slice_pinned = g_slice != NULL;
    if (!g_slice) g_slice = (uint8_t*)malloc(bytes ? bytes : 1u);
    g_slice_cap = g_slice ? bytes : 0;
    return g_slice ? 0 : -1;
}

typedef struct { FILE* src; data_block_t* dic; } dicpick_job;
static void* dicpick_main(void* p) { dicpick_job* j = (dicpick_job*)p; dicpick(j->src, j->dic); return NULL; }

/* What the main thread does while the dictionary is picked: everything that does not need the dictionary. */
static void while_picking(void) {
    if (crgpu_shim_prepare() != CRGPU_OK) return;             /* (reported again, with its message, by the first shim call) */
    mark("hip_ready");
    if (!opt_indep_kib) return;
    g_mg = create_multi();
    mark("gpu_contexts_ready");
    if (!g_mg || g_src_fd < 0) return;
    const uint32_t block = opt_indep_kib * 1024u;
    const uint64_t nb_all = g_src_size / block + 1u;
    const uint64_t sb = slice_blocks_for(nb_all, block);
    if (slice_buffer((size_t)(sb * block)) != 0) return;
    const uint64_t want = g_src_size < sb * block ? g_src_size : sb * block;
    uint64_t got = 0;
    while (got < want) {                                       /* pread: dicpick's position in the same file is not touched */
        const ssize_t r = pread(g_src_fd, g_slice + got, (size_t)(want - got), (off_t)got);
        if (r <= 0) break;
        got += (uint64_t)r;
    }
    if (got == want) { g_first_bytes = got; g_first_read = 1; }
    mark("first_slice_read");
}

static int write_dictionary(FILE* src, FILE* dst, char** text_out) {          /* src/main.c:156-171 */
    data_block_t dic = {0, 0, 0}, packed = {0, 0, 0};
    SAY("-> building static dictionary...\n");
    /* the reference overlaps its census with the file read on a helper thread (src/cr-dicpick.c:188-216); here the
     * census runs on the helper while this thread brings the GPU up (HIP runtime, contexts, staging memory, first read) */
    dicpick_job job = {src, &dic};
    pthread_t th;
    if (pthread_create(&th, NULL, dicpick_main, &job) == 0) {
        while_picking();
        pthread_join(th, NULL);
    } else {
        dicpick(src, &dic);
    }
    mark("dicpick_done");
    rewind(src);
    if (text_out) {                              /* the batched calls take the dictionary as their own object */
        *text_out = (char*)malloc((size_t)dic.m_size + 1u);
        if (!*text_out) return -1;
        memcpy(*text_out, dic.m_data, dic.m_size);
        (*text_out)[dic.m_size] = 0;
    }
    /* the per-block entry points of the stock loop need the process-wide dictionary (dictionary_load); the sharded path
     * of -k has one copy per GPU instead (crgpu_multi_set_dictionary) and reports the same count */
    int nword = 0;
    if (text_out) { for (uint32_t i = 0; i < dic.m_size; i++) nword += dic.m_data[i] == '\n'; }
    else nword = dictionary_load((const char*)dic.m_data, 1);
    mark("dictionary_loaded");
    dic_lcp_encode(&dic);
    lzencode(&dic, &packed, 0);
    mark("dictionary_blob_coded");
    mark_at("(dictionary_blob_kernel_seconds)", (double)crgpu_shim_last_kernel_ms() * 1e-3);
    reset_models();
    SAY("added %d words to dictionary, compressed size = %u bytes\n", nword, packed.m_size);
    fwrite(&packed.m_size, sizeof packed.m_size, 1, dst);
    fwrite(packed.m_data, 1, packed.m_size, dst);
    data_block_destroy(&dic);
    data_block_destroy(&packed);
    return 0;
}

static void put_block(FILE* dst, const uint8_t* p, uint32_t n, int filt) {          /* src/main.c:198-205 */
    block_head_t h;
    if (n == 0) return;
    h.m_size = n; h.m_filt = (uint8_t)filt; h.m_prec = (uint8_t)opt_prec;
    fwrite(&h, sizeof h, 1, dst);
    fwrite(p, 1, n, dst);
}

/* block loop of src/main.c:174-206, one block at a time through the per-block entry points */
static int encode_sequential(FILE* src, FILE* dst) {
    data_block_t x = {0, 0, 0}, y = {0, 0, 0};
    int filt = 0;                                /* src/main.c:106: keeps its last value once a filter has matched */
    while (!ferror(src) && !ferror(dst) && !feof(src)) {
        data_block_resize(&x, opt_block);
        x.m_size = (uint32_t)fread(x.m_data, 1, opt_block, src);
        if (opt_filt) { SAY("-> running filters...\n"); filt = filter_inplace(x.m_data, x.m_size, FILTER_ENC) ? opt_filt : 0; }   /* src/main.c:183-185 */
        data_block_resize(&y, 0);
        dictionary_encode(&x, &y);
        if (!opt_prec) {
            data_block_resize(&x, 0);
            lzencode(&y, &x, 0);                 /* no reset_models() here: block k starts from block k-1's models */
            put_block(dst, x.m_data, x.m_size, filt);
        } else {
            put_block(dst, y.m_data, y.m_size, filt);
        }
    }
    data_block_destroy(&x);
    data_block_destroy(&y);
    return (ferror(src) || ferror(dst)) ? -1 : 0;
}

/* -k: the same loop body for every block of the file, sharded over the GPUs of -g (csrc/crgpu_multi.hip): each GPU
 * runs dictionary stage, codec and k_pack on its contiguous range of blocks and hands back its run of the file */
static int encode_batched(crgpu_multi* mg, FILE* src, FILE* dst, uint64_t size) {
    const uint32_t block = opt_indep_kib * 1024u;
    /* the reference reads until a short read, so a file that is a multiple of the block size gets a
     * trailing empty block (src/main.c:174-180) */
    const uint64_t nb_all = size / block + 1u;
    const uint64_t slice_blocks = slice_blocks_for(nb_all, block);
    if (slice_buffer((size_t)(slice_blocks * block)) != 0) return -1;
    uint8_t* const data = g_slice;
    uint64_t* off = (uint64_t*)malloc((size_t)slice_blocks * sizeof *off);
    uint32_t* len = (uint32_t*)malloc((size_t)slice_blocks * sizeof *len);
    uint8_t* filt = (uint8_t*)calloc((size_t)slice_blocks, 1);
    if (!off || !len || !filt) return -1;
    int rc = 0;
    for (uint64_t first = 0; first < nb_all && rc == 0; first += slice_blocks) {
        const uint32_t nb = (uint32_t)(nb_all - first < slice_blocks ? nb_all - first : slice_blocks);
        const uint64_t at = first * block;
        const uint64_t bytes = size - at < (uint64_t)nb * block ? size - at : (uint64_t)nb * block;
        if (first == 0 && g_first_read && g_first_bytes == bytes) {
            if (fseek(src, (long)bytes, SEEK_CUR) != 0) return die("fseek()");     /* read while the dictionary was picked */
        } else if (fread(data, 1, (size_t)bytes, src) != bytes) return die("fread()");
        mark("slice_read");
        for (uint32_t b = 0; b < nb; b++) {
            off[b] = (uint64_t)b * block;
            len[b] = (uint32_t)(bytes - off[b] < block ? bytes - off[b] : block);
            /* the filters are a sequential host pass over the blocks in file order, like the stock loop's */
            filt[b] = opt_filt && filter_inplace(data + off[b], len[b], FILTER_ENC) ? (uint8_t)opt_filt : (uint8_t)0;
        }
        SAY("-> dictionary stage + %s on %u blocks, %d GPU(s)%s...\n", opt_prec ? "no codec (-p)" : "LZ/ARI encoding", nb, crgpu_multi_devices(mg),
            crgpu_multi_uses_rccl(mg) ? ", sizes by RCCL all-gather" : "");
        uint8_t* body = NULL;
        uint64_t total = 0;
        const int e = crgpu_multi_encode_blocks(mg, CR_CODEC, CRGPU_MULTI_DICT | CRGPU_MULTI_HEADERS | (opt_prec ? CRGPU_MULTI_PREC : 0),
                                                data, off, len, nb, filt, &body, &total, NULL, NULL);
        if (e != CRGPU_OK) { fprintf(stderr, "GPU codec failed (%d): %s\n", e, crgpu_multi_last_error(mg)); rc = -1; break; }
        if (opt_times) {                                    /* rank 0's marks inside the job */
            double tm[CRGPU_MULTI_TIMES] = {0};
            const double end = wall() - g_t0;
            static const char* const names[CRGPU_MULTI_TIMES] = {NULL, "job_h2d_done", "job_kernels_done", "job_sizes_exchanged", "job_output_allocated", "job_d2h_done"};
            if (crgpu_multi_timing(mg, 0, tm, CRGPU_MULTI_TIMES) == CRGPU_MULTI_TIMES)
                for (int i = 1; i < CRGPU_MULTI_TIMES; i++) mark_at(names[i], end - (tm[CRGPU_MULTI_TIMES - 1] - tm[i]));
            mark("job_returned");
        }
        if (total) fwrite(body, 1, total, dst);             /* headers and payloads already lie as src/main.c:198-205 writes them */
        crgpu_multi_free(body);
        if (ferror(dst)) rc = -1;
    }
    free(filt); free(off); free(len);
    return rc;
}

/* ---- decode ------------------------------------------------------------------------------- */

static int read_dictionary(FILE* src, char** text_out) {     /* src/main.c:244-259 */
    data_block_t packed = {0, 0, 0}, dic = {0, 0, 0};
    uint32_t csize = 0;
    SAY("-> decoding static dictionary...\n");
    if (fread(&csize, sizeof csize, 1, src) != 1) return -1;
    {                                            /* a size field larger than what is left of the file is not a dictionary */
        const long at = ftell(src);
        if (at >= 0 && fseek(src, 0, SEEK_END) == 0) {
            const long end = ftell(src);
            if (fseek(src, at, SEEK_SET) != 0 || (end >= at && (uint64_t)csize > (uint64_t)(end - at))) { errno = EINVAL; return -1; }
        }
    }
    if (csize > crgpu_bound(CR_CODEC, CRGPU_MAX_BLOCK)) { errno = EINVAL; return -1; }
    data_block_resize(&packed, csize);
    if (fread(packed.m_data, 1, csize, src) != csize) return -1;
    lzdecode(&packed, &dic, 0);
    mark("dictionary_blob_decoded");
    mark_at("(dictionary_blob_kernel_seconds)", (double)crgpu_shim_last_kernel_ms() * 1e-3);
    mark_at("(dictionary_blob_coded_bytes)", (double)csize);
    if (crgpu_shim_status() != CRGPU_OK) return -1;
    reset_models();
    dic_lcp_decode(&dic);
    if (dic.m_size == 0) { fprintf(stderr, "malformed dictionary.\n"); errno = EINVAL; return -1; }
    if (!text_out) dictionary_load((const char*)dic.m_data, 0);     /* (-k files: one copy per GPU instead, open_multi) */
    mark("dictionary_loaded");
    if (crgpu_shim_status() != CRGPU_OK) return -1;
    if (text_out) {                              /* the batched calls take the dictionary as their own object */
        *text_out = (char*)malloc((size_t)dic.m_size + 1u);
        if (!*text_out) return -1;
        memcpy(*text_out, dic.m_data, dic.m_size);
        (*text_out)[dic.m_size] = 0;
    }
    data_block_destroy(&packed);
    data_block_destroy(&dic);
    return 0;
}

/* -k files: every block is independent, so the file goes through the sharded GPU path in slices (blocks as they lie in
 * the file, at most 65 536 of them or 1 GiB) instead of one launch per block */
static int decode_batched(crgpu_multi* mg, FILE* src, FILE* dst) {
    enum { MAXB = 65536 };
    uint64_t* off = (uint64_t*)malloc(8u * MAXB), *ooff = (uint64_t*)malloc(8u * MAXB);
    uint32_t* len = (uint32_t*)malloc(4u * MAXB), *olen = (uint32_t*)malloc(4u * MAXB);
    uint8_t* prec = (uint8_t*)malloc(MAXB), *filt = (uint8_t*)malloc(MAXB);
    if (!off || !ooff || !len || !olen || !prec || !filt) return -1;
    uint8_t* pk = NULL;
    size_t pk_cap = 0;
    int rc = 0, more = 1;
    while (more && rc == 0) {
        uint32_t nb = 0;
        size_t used = 0;
        while (nb < MAXB && used < ((size_t)1 << 30)) {
            block_head_t h;
            if (fread(&h, sizeof h, 1, src) != 1) { more = 0; break; }
            if (h.m_size > crgpu_bound(CR_CODEC, CRGPU_MAX_BLOCK + 1u)) { rc = -1; break; }
            if (used + h.m_size > pk_cap) {
                pk_cap = (used + h.m_size) * 2u + 65536u;
                pk = (uint8_t*)realloc(pk, pk_cap);
                if (!pk) return -1;
            }
            if (fread(pk + used, 1, h.m_size, src) != h.m_size) { rc = -1; break; }
            off[nb] = used; len[nb] = h.m_size; prec[nb] = h.m_prec; filt[nb] = h.m_filt;
            used += h.m_size; nb++;
        }
        if (rc || nb == 0) break;
        mark("slice_read");
        SAY("-> LZ/ARI + dictionary decoding (%u blocks, %d GPU(s))...\n", nb, crgpu_multi_devices(mg));
        uint8_t* body = NULL;
        uint64_t total = 0;
        const int e = crgpu_multi_decode_blocks(mg, CR_CODEC, CRGPU_MULTI_DICT, pk, off, len, nb, prec, &body, &total, ooff, olen);
        if (e != CRGPU_OK) { fprintf(stderr, "GPU codec failed (%d): %s\n", e, crgpu_multi_last_error(mg)); rc = -1; break; }
        if (opt_times) {
            double tm[CRGPU_MULTI_TIMES] = {0};
            const double end = wall() - g_t0;
            static const char* const names[CRGPU_MULTI_TIMES] = {NULL, "job_h2d_done", "job_kernels_done", "job_sizes_exchanged", "job_output_allocated", "job_d2h_done"};
            if (crgpu_multi_timing(mg, 0, tm, CRGPU_MULTI_TIMES) == CRGPU_MULTI_TIMES)
                for (int i = 1; i < CRGPU_MULTI_TIMES; i++) mark_at(names[i], end - (tm[CRGPU_MULTI_TIMES - 1] - tm[i]));
            mark("job_returned");
        }
        for (uint32_t b = 0; b < nb; b++)                   /* the inverse filters: a sequential host pass in file order */
            if (filt[b]) {
                (void)crgpu_filter_set_mode(filt[b] == 2 ? CRGPU_FILTER_RESTART_ELF : CRGPU_FILTER_REFERENCE);    /* 2: written by -FF */
                filter_inplace(body + ooff[b], olen[b], FILTER_DEC);
            }
        if (total) fwrite(body, 1, total, dst);
        crgpu_multi_free(body);
        if (ferror(dst)) rc = -1;
    }
    free(pk); free(off); free(ooff); free(len); free(olen); free(prec); free(filt);
    return rc;
}

static int decode_stream(FILE* src, FILE* dst, int stock) {   /* src/main.c:263-292 */
    data_block_t x = {0, 0, 0}, y = {0, 0, 0};
    block_head_t h;
    uint32_t seen = 0;
    while (!ferror(src) && !ferror(dst) && fread(&h, sizeof h, 1, src) == 1) {
        data_block_resize(&y, h.m_size);
        if (fread(y.m_data, 1, h.m_size, src) != h.m_size) return -1;
        (void)seen;
        data_block_resize(&x, 0);
        /* a filtered block has to come back into memory whole: the inverse filter runs on it before it is
         * written (the reference streams it out of dictionary_decode first and filters an empty buffer) */
        FILE* const sync = h.m_filt ? NULL : dst;
        data_block_t* out = &y;
        if (!h.m_prec) {
            if (!stock) reset_models();          /* -k files: independent blocks; stock files carry the models over */
            lzdecode(&y, &x, 0);
            data_block_resize(&y, 0);
            dictionary_decode(&x, &y, sync);
        } else {
            dictionary_decode(&y, &x, sync);
            out = &x;
        }
        if (h.m_filt) {
            SAY("-> running filters...\n");
            (void)crgpu_filter_set_mode(h.m_filt == 2 ? CRGPU_FILTER_RESTART_ELF : CRGPU_FILTER_REFERENCE);    /* 2: written by -FF */
            filter_inplace(out->m_data, out->m_size, FILTER_DEC);
        }
        if (out->m_size > 0) fwrite(out->m_data, 1, out->m_size, dst);
    }
    data_block_destroy(&x);
    data_block_destroy(&y);
    return (ferror(src) || ferror(dst)) ? -1 : 0;
}

/* ---- driver ------------------------------------------------------------------------------- */

static crgpu_multi* open_multi(const char* dictionary_text) {
    crgpu_multi* mg = g_mg ? g_mg : create_multi();          /* (the encoder has made it while the dictionary was picked) */
    g_mg = NULL;
    if (!mg) return NULL;
    const int rc = crgpu_multi_set_dictionary(mg, dictionary_text);
    if (rc != CRGPU_OK) { fprintf(stderr, "GPU setup failed (%d): %s\n", rc, crgpu_multi_last_error(mg)); crgpu_multi_destroy(mg); return NULL; }
    return mg;
}

int main(int argc, char** argv) {
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    g_t0 = wall();
    if ((argc = process_arguments(argc, argv)) == 0) return -1;
    if (opt_ndev && !opt_indep_kib && argc >= 2 && strcmp(argv[1], "e") == 0) {
        /* the stock loop's blocks depend on each other: there is nothing to shard */
        fprintf(stderr, "-g / -G need -k (independent blocks).\n");
        return -1;
    }
    SAY("%s\n", BANNER);
    const int enc = argc >= 2 && argc <= 4 && strcmp(argv[1], "e") == 0;
    const int dec = argc >= 2 && argc <= 4 && strcmp(argv[1], "d") == 0;
    if (!enc && !dec) { fprintf(stderr, "%s\n", USAGE); return -1; }
    const char* src_name = argc >= 3 ? argv[2] : "<stdin>";
    const char* dst_name = argc >= 4 ? argv[3] : "<stdout>";
    FILE* src = argc >= 3 ? fopen(argv[2], "rb") : spool_stdin();
    FILE* dst = argc >= 4 ? fopen(argv[3], "wb") : stdout;
    if (!src || !dst) return die("fopen()");
    if (crgpu_shim_config(CR_CODEC, 0) != CRGPU_OK || crgpu_shim_rox_chain_limit(opt_depth) != CRGPU_OK ||
        crgpu_shim_flexible_parsing(opt_flex) != CRGPU_OK) return -1;

    int rc = 0;
    if (enc) {
        fseek(src, 0, SEEK_END);
        const uint64_t size = (uint64_t)ftell(src);
        rewind(src);
        g_src_fd = fileno(src);
        g_src_size = size;
        {   /* a slice's coded run: text comes out at a quarter, nothing at more than its own size + headers */
            const uint64_t slice = (uint64_t)1 << 30;
            g_pool_hint = opt_indep_kib && size > slice ? pool_hint_capped(slice / 2u + (1u << 20)) : 0u;
        }
        /* without -k the block loop is the stock one (models carried from block to block), so the
         * file is the stock tool's, byte for byte; -k files are marked with format byte 2 */
        fwrite(opt_indep_kib ? MAGIC_INDEP : MAGIC_STOCK, 1, sizeof MAGIC_STOCK - 1, dst);
        SAY("compressing %s to %s, block_size = %s%u%s...\n", src_name, dst_name, "",
            opt_indep_kib ? opt_indep_kib : opt_block / 1048576u, opt_indep_kib ? "KiB (independent)" : "MB");
        char* text = NULL;
        mark("files_open");
        if (write_dictionary(src, dst, opt_indep_kib ? &text : NULL)) return die("dictionary");
        mark("dictionary_written");
        if (opt_indep_kib) {
            /* the shims own a context and the process-wide dictionary (they coded the dictionary blob); the sharded
             * path has one context + dictionary copy per GPU, created from the same text */
            crgpu_multi* mg = open_multi(text);
            if (!mg) return -1;
            mark("dictionary_on_gpus");
            rc = encode_batched(mg, src, dst, size);
            mark("blocks_written");
            crgpu_multi_destroy(mg);
            free(text);
            mark("gpu_contexts_closed");
        } else {
            crgpu_shim_expect_dependent_blocks(size > opt_block);      /* several blocks: they continue each other's models */
            rc = encode_sequential(src, dst);
        }
    } else {
        char magic[64] = {0};
        if (fread(magic, 1, sizeof MAGIC_STOCK - 1, src) != sizeof MAGIC_STOCK - 1 ||
            (memcmp(magic, MAGIC_STOCK, sizeof MAGIC_STOCK - 1) && memcmp(magic, MAGIC_INDEP, sizeof MAGIC_INDEP - 1))) {
            fprintf(stderr, "check_magic() failed.\n");
            return -1;
        }
        SAY("decompressing %s to %s...\n", src_name, dst_name);
        const int stock = memcmp(magic, MAGIC_STOCK, sizeof MAGIC_STOCK - 1) == 0;
        char* text = NULL;
        mark("files_open");
        /* -k files: the per-GPU contexts of the sharded path come up on a helper thread while this one decodes the
         * dictionary blob (one block, one dependent chain: ~0.1 s whatever the file's size) */
        pthread_t th;
        int helper = 0;
        if (!stock) {
            /* a slice's decoded run: coded text decodes to four or five times its size (a slice holds at most 1 GiB of coded blocks) */
            const long at = ftell(src);
            if (at >= 0 && fseek(src, 0, SEEK_END) == 0) {
                const uint64_t coded = (uint64_t)ftell(src);
                g_pool_hint = coded > ((uint64_t)1 << 30) ? pool_hint_capped((uint64_t)2 << 30) : 0u;      /* more than one slice of coded blocks */
                if (fseek(src, at, SEEK_SET) != 0) return die("fseek()");
            }
            helper = pthread_create(&th, NULL, create_multi_main, NULL) == 0;
        }
        const int drc = read_dictionary(src, stock ? NULL : &text);
        if (helper) pthread_join(th, NULL);
        if (drc) return die("dictionary");
        mark("dictionary_read");
        if (stock) {
            {   /* more than one block in the file? then they continue each other's models (the stock loop never resets them) */
                const long at = ftell(src);
                block_head_t h;
                if (at >= 0 && fread(&h, sizeof h, 1, src) == 1 && fseek(src, 0, SEEK_END) == 0) {
                    const long end = ftell(src);
                    crgpu_shim_expect_dependent_blocks(end > at + (long)sizeof h + (long)h.m_size + (long)sizeof h);
                }
                if (at < 0 || fseek(src, at, SEEK_SET) != 0) return die("fseek()");
            }
            rc = decode_stream(src, dst, 1);
        } else {
            crgpu_multi* mg = open_multi(text);
            if (!mg) return -1;
            mark("gpu_contexts_ready");
            rc = decode_batched(mg, src, dst);
            mark("blocks_written");
            crgpu_multi_destroy(mg);
            free(text);
            mark("gpu_contexts_closed");
        }
    }
    if (rc) { fprintf(stderr, "failed.\n"); return -1; }
    if (crgpu_filter_lossy()) {
        /* the reference's ELF filter never resets its byte counter (src/filter_x86_elf.c:131-134): every ELF image after the
         * first of a run is converted in a way FILTER_DEC cannot undo. The bytes are the reference's; say what they cost
         * (not silenced by -q: the exit status stays 0 and nothing else would tell) */
        fprintf(stderr, enc ? "warning: -F converted %d ELF image(s) with the reference's never-reset byte counter; this file will NOT decode "
                              "back to the input (the reference's own does not either). Use -FF for a filter that round-trips.\n"
                            : "warning: %d ELF image(s) of this file were filtered with the reference's never-reset byte counter and could not be "
                              "restored; the output differs from the original input there.\n", crgpu_filter_lossy());
    }
    const long src_size = ftell(src), dst_size = ftell(dst);
    fclose(src);
    if (fclose(dst) != 0) return die("fclose()");         /* ENOSPC / EIO can surface only here */
    mark("files_closed");
    print_marks();
    const int exit_fast = 1;                                 /* see the end of main */
    gettimeofday(&t1, NULL);
    const double secs = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_usec - t0.tv_usec) / 1e6;
    SAY("%ld bytes => %ld bytes\n\n", src_size, dst_size);                /* src/main.c:318-329 */
    if (enc) {
        SAY("encode-speed:   %.3lf MB/s\n", (double)(src_size / 1048576) / secs);
        SAY("cost-time:      %.3lf s\n", secs);
        SAY("compress-ratio: %.3lf\n", (double)dst_size / (double)(src_size ? src_size : 1));
    } else {
        SAY("decode-speed:   %.3lf MB/s\n", (double)(dst_size / 1048576) / secs);
        SAY("cost-time:      %.3lf s\n", secs);
    }
    /* both files are closed and flushed: leave without the HIP runtime's orderly teardown (tens of milliseconds of
     * unloading and freeing what the driver reclaims at process exit anyway) */
    if (exit_fast) { fflush(stderr); _exit(0); }
    return 0;
}

/*
 * comprox_amd/csrc/crgpu_multi.hip — the block loop of the reference's driver on several GPUs of one node.
 *
 * Replaces: the encode loop /root/reference/src/main.c:174-206 (read block, filter, dictionary_encode, lzencode,
 * write header + payload) and the decode loop src/main.c:263-292, for INDEPENDENT datablocks (reset_models() per
 * block): SURVEY.md §8e, BASELINE.json config 3. Built on the public C-ABI of include/crgpu.h only (one crgpu_ctx
 * per device), the HIP runtime for the staging copies and RCCL for the one exchange the path has.
 *
 * Shape (MI355X node: 8 GPUs, xGMI point to point):
 *   - one host thread per GPU, each with its own context, stream and device buffers;
 *   - rank r owns the contiguous block range [r * ceil(nb / G), (r + 1) * ceil(nb / G)) — blocks are independent, so
 *     nothing else is shared but the read-only dictionary (built once on the host, uploaded to every GPU);
 *   - on its range a rank runs dictionary stage -> codec -> k_pack (headers + payloads back to back, on the device),
 *     so its share of the container is ONE contiguous device buffer and one D2H copy;
 *   - the only exchange: the per-block output sizes (4 B each), ncclAllGather over the communicator of the devices
 *     (RCCL; 8 x 1 908 x 4 B for enwik9 — latency, not bandwidth). Every rank then knows every size, takes the
 *     exclusive sum in front of its range as the file offset of its run and copies it there. Payload bytes never
 *     cross GPUs;
 *   - a device list that names one GPU twice (a one-GPU box rehearsing two ranks) cannot form a communicator: the
 *     ranks are threads of one process, so the size table is then exchanged through host memory behind a barrier.
 *     CRGPU_MULTI_HOST_GATHER asks for that explicitly.
 *
 * librccl.so is loaded with dlopen at crgpu_multi_create — a single-GPU user of libcrgpu.so never loads it.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <rccl/rccl.h>

#include "../../include/crgpu.h"

#define MULTI_MAX 16

/* ---- host twins of the device planning (exported: the CPU tests call them, the driver uses them) ---------------- */

extern "C" void crgpu_shard_range(uint32_t nblocks, int nranks, int rank, uint32_t* first, uint32_t* count) {
    if (nranks < 1) nranks = 1;
    const uint32_t per = (nblocks + (uint32_t)nranks - 1u) / (uint32_t)nranks;
    uint64_t lo = (uint64_t)per * (uint32_t)rank;
    if (lo > nblocks) lo = nblocks;
    uint64_t hi = lo + per;
    if (hi > nblocks) hi = nblocks;
    if (first) *first = (uint32_t)lo;
    if (count) *count = (uint32_t)(hi - lo);
}

extern "C" uint64_t crgpu_container_offsets(const uint32_t* sizes, uint32_t nblocks, int with_headers, uint64_t* out_off) {
    uint64_t at = 0;
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t s = sizes[b] == 0xFFFFFFFFu ? 0u : sizes[b];
        const uint32_t head = (with_headers && s) ? 6u : 0u;         /* src/main.c:198: empty blocks are not written */
        if (out_off) out_off[b] = at + head;
        at += (uint64_t)s + head;
    }
    return at;
}

/* ---- RCCL through dlopen ------------------------------------------------------------------------------------ */

struct rccl_api {
    void* lib;
    decltype(&ncclCommInitAll)     CommInitAll;
    decltype(&ncclCommDestroy)     CommDestroy;
    decltype(&ncclAllGather)       AllGather;
    decltype(&ncclGetErrorString)  GetErrorString;
};

static int rccl_open(rccl_api* a, char* err, size_t errlen) {
    memset(a, 0, sizeof *a);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (size_t i = 0; i < sizeof names / sizeof names[0] && !a->lib; i++) a->lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!a->lib) { snprintf(err, errlen, "librccl.so not found: %s", dlerror()); return CRGPU_E_NODEVICE; }
    a->CommInitAll = (decltype(a->CommInitAll))dlsym(a->lib, "ncclCommInitAll");
    a->CommDestroy = (decltype(a->CommDestroy))dlsym(a->lib, "ncclCommDestroy");
    a->AllGather = (decltype(a->AllGather))dlsym(a->lib, "ncclAllGather");
    a->GetErrorString = (decltype(a->GetErrorString))dlsym(a->lib, "ncclGetErrorString");
    if (!a->CommInitAll || !a->CommDestroy || !a->AllGather || !a->GetErrorString) {
        snprintf(err, errlen, "librccl.so lacks an expected symbol");
        dlclose(a->lib); a->lib = NULL;
        return CRGPU_E_NODEVICE;
    }
    return CRGPU_OK;
}

/* ---- the driver --------------------------------------------------------------------------------------------- */

/* results handed out from a context's page-locked pool are not the caller's to free: crgpu_multi_free looks them up here */
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static void* g_pools[64];
static int pool_remember(void* p) {              /* 0: the registry is full (64 live pools): the caller must not hand this pointer out */
    int ok = 0;
    pthread_mutex_lock(&g_pool_mu);
    for (int i = 0; i < 64 && !ok; i++) if (!g_pools[i]) { g_pools[i] = p; ok = 1; }
    pthread_mutex_unlock(&g_pool_mu);
    return ok;
}
static void pool_forget(void* p) { pthread_mutex_lock(&g_pool_mu); for (int i = 0; i < 64; i++) if (g_pools[i] == p) g_pools[i] = NULL; pthread_mutex_unlock(&g_pool_mu); }
static int pool_known(void* p) { int k = 0; pthread_mutex_lock(&g_pool_mu); for (int i = 0; i < 64; i++) if (g_pools[i] == p) k = 1; pthread_mutex_unlock(&g_pool_mu); return k; }

struct dev_buf { uint8_t* p; size_t cap; };

struct rank_state {
    int         device;
    crgpu_ctx*  ctx;
    crgpu_dict* dict;
    hipStream_t stream;
    ncclComm_t  comm;
    dev_buf     in, st1, enc, pack, meta, sizes, all;
    uint32_t*   h_all;          /* this rank's copy of the gathered size table */
    size_t      h_all_cap;
    int         rc;
    char        err[256];
    double      t_mark[CRGPU_MULTI_TIMES];   /* seconds since the job started: see crgpu_multi_timing */
    int         where;          /* progress of the running job (RANK_*): what the deadline report names */
};

/* where a rank of the running job is; a rank below RANK_AT_BARRIER when the deadline passes has not arrived */
enum { RANK_IDLE = 0, RANK_STAGES, RANK_EXCHANGE, RANK_AT_BARRIER, RANK_COPY_OUT, RANK_DONE };
static const char* const rank_where[] = {"not started", "its stages (upload / kernels)", "the size exchange", "the barrier", "the copy-out", "done"};

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + (double)ts.tv_nsec * 1e-9; }

struct job {
    int            decode, codec, flags;
    const uint8_t* in;
    const uint64_t* in_off;
    const uint32_t* in_size;
    const uint8_t* per_block;   /* encode: m_filt per block; decode: m_prec per block */
    uint32_t       nblocks;
    uint8_t*       out;         /* allocated by rank 0 between the two barriers */
    uint64_t       out_total;
    uint64_t*      out_off;
    uint32_t*      out_size;
    uint32_t*      host_all;    /* host exchange: the shared size table */
    double         t0;          /* when the job was handed to the ranks */
    int            failed;      /* any rank failed before the allocation: nobody copies */
};

struct thread_arg { struct crgpu_multi* m; int r; };

struct crgpu_multi {
    int          ndev;
    int          use_rccl;
    rccl_api     rccl;
    rank_state   rank[MULTI_MAX];
    pthread_barrier_t bar;
    int          bar_ok;
    /* one worker thread per rank lives as long as the context: a job is handed over under `mu` */
    pthread_t    thread[MULTI_MAX];
    thread_arg   targ[MULTI_MAX];
    int          nthreads;
    pthread_mutex_t mu;
    pthread_cond_t  cv_go, cv_done;
    int          sync_ok;
    uint64_t     seq;
    int          done, quit;
    job          j;
    char         err[320];
    int          pinned_out;    /* CRGPU_MULTI_PINNED_OUT: results live in the context's page-locked pool until the next job */
    uint8_t*     pool;          /* hipHostMalloc, grown as needed, kept for the context's life */
    size_t       pool_cap;
    double       deadline_s;    /* a job that has not finished after this many seconds is abandoned (0: wait for ever) */
    int          broken;        /* a job missed its deadline: its threads may still sit in a collective or behind a kernel */
    int          test_stall;    /* tests: this rank never starts its job (crgpu_multi_test_stall_rank), -1 = none */
};

static uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

static int dgrow(rank_state* R, dev_buf* b, size_t want) {
    if (b->cap >= want) return CRGPU_OK;
    if (b->p) (void)hipFree(b->p);
    b->p = NULL; b->cap = 0;
    const size_t sz = want + want / 8 + 4096;
    if (hipMalloc((void**)&b->p, sz) != hipSuccess) { snprintf(R->err, sizeof R->err, "hipMalloc(%zu) failed on device %d", sz, R->device); return CRGPU_E_NOMEM; }
    b->cap = sz;
    return CRGPU_OK;
}

#define M_HIP(R, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf((R)->err, sizeof (R)->err, "%s: %s", #call, hipGetErrorString(e_)); return CRGPU_E_NODEVICE; } } while (0)
#define M_RC(R, call) do { int rc_ = (call); if (rc_ != CRGPU_OK) { if (!(R)->err[0]) snprintf((R)->err, sizeof (R)->err, "%s failed (%d): %s", #call, rc_, crgpu_last_error((R)->ctx)); return rc_; } } while (0)

static uint32_t header_bytes(int codec) {
    return codec == CRGPU_CODEC_ROX ? CRGPU_ROX_HEADER : codec == CRGPU_CODEC_ROLZ ? CRGPU_ROLZ_HEADER : CRGPU_ROP_HEADER;
}

/* exchange of the per-block sizes: d_mine = this rank's `per` entries (zero padded) on its device; afterwards
 * R->h_all (RCCL) or the shared table (host exchange) holds all G * per entries in rank order. Every rank takes part,
 * a rank whose stages failed with a table of zeros — nobody is left waiting in the collective. */
static int exchange_sizes(crgpu_multi* m, int r, const uint32_t* d_mine, uint32_t per) {
    rank_state* R = &m->rank[r];
    const size_t total = (size_t)per * (size_t)m->ndev;
    if (total == 0) return CRGPU_OK;
    if (m->use_rccl) {
        const ncclResult_t e = m->rccl.AllGather(d_mine, R->all.p, per, ncclUint32, R->comm, R->stream);
        if (e != ncclSuccess) { snprintf(R->err, sizeof R->err, "ncclAllGather: %s", m->rccl.GetErrorString(e)); return CRGPU_E_NODEVICE; }
        M_HIP(R, hipMemcpyAsync(R->h_all, R->all.p, total * 4u, hipMemcpyDeviceToHost, R->stream));
        M_HIP(R, hipStreamSynchronize(R->stream));
    } else {
        M_HIP(R, hipMemcpyAsync(m->j.host_all + (size_t)r * per, d_mine, (size_t)per * 4u, hipMemcpyDeviceToHost, R->stream));
        M_HIP(R, hipStreamSynchronize(R->stream));
    }
    return CRGPU_OK;
}

/* H2D of the blocks [first, first + count): blocks that lie back to back in `in` travel in one copy */
static uint64_t plan_range(const job* J, uint32_t first, uint32_t count, uint64_t* h_off) {
    uint64_t at = 0;
    for (uint32_t k = 0; k < count; k++) {
        /* the decoders read their input at any alignment, so a packed run of coded blocks (what the encode call returns: 1 526
         * blocks of ~14 KB) stays one run on the device and is one copy, not one per block; the encoders' inputs start on 16 bytes */
        const bool follows = J->decode && k > 0 && J->in_off[first + k] == J->in_off[first + k - 1u] + J->in_size[first + k - 1u];
        if (!follows) at = up(at, 16);
        h_off[k] = at;
        at += J->in_size[first + k];
    }
    return up(at, 16);
}

static int upload_range(rank_state* R, const job* J, uint32_t first, uint32_t count, const uint64_t* h_off) {
    for (uint32_t k = 0; k < count;) {
        uint32_t last = k;
        while (last + 1u < count && J->in_off[first + last + 1u] == J->in_off[first + last] + J->in_size[first + last] &&
               h_off[last + 1u] == h_off[last] + J->in_size[first + last]) last++;
        const uint64_t bytes = J->in_off[first + last] + J->in_size[first + last] - J->in_off[first + k];
        if (bytes) M_HIP(R, hipMemcpyAsync(R->in.p + h_off[k], J->in + J->in_off[first + k], (size_t)bytes, hipMemcpyHostToDevice, R->stream));
        k = last + 1u;
    }
    return CRGPU_OK;
}

static int encode_rank(crgpu_multi* m, int r, uint32_t first, uint32_t count, uint32_t per, uint64_t* my_total, const uint32_t** d_mine_out) {
    rank_state* R = &m->rank[r];
    const job* J = &m->j;
    const int use_dict = (J->flags & CRGPU_MULTI_DICT) != 0, prec = (J->flags & CRGPU_MULTI_PREC) != 0;
    const int headers = (J->flags & CRGPU_MULTI_HEADERS) != 0;
    M_HIP(R, hipSetDevice(R->device));
    if (!use_dict && prec) {                                    /* no stage selected (check_job refuses such flags) */
        snprintf(R->err, sizeof R->err, "nothing to do: neither dictionary stage nor codec selected");
        return CRGPU_E_ARG;
    }
    if (count == 0) {                                           /* ceil(nb / G) leaves trailing ranks without blocks (always when nb < G) */
        *my_total = 0;
        *d_mine_out = (const uint32_t*)R->sizes.p;              /* the zeroed table run_job prepared */
        return CRGPU_OK;
    }
    /* meta (host, then device): [in_off | st1_off | enc_off | pack_off] u64 x count, [in_size | len1 | len2 (per, padded)] u32, filt u8 */
    const size_t n8 = (size_t)(count ? count : 1u);
    uint64_t* h = (uint64_t*)malloc(n8 * 8u * 3u);
    if (!h) return CRGPU_E_NOMEM;
    uint64_t *h_in = h, *h_st1 = h + n8, *h_enc = h + 2 * n8;
    uint64_t st1_bytes = 0, enc_bytes = 0;
    uint32_t max_block = 0;
    const uint64_t in_bytes = plan_range(J, first, count, h_in);
    int rc = dgrow(R, &R->in, (size_t)in_bytes + 16u);
    if (rc == CRGPU_OK) rc = upload_range(R, J, first, count, h_in);
    if (rc != CRGPU_OK) { free(h); return rc; }
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t n = J->in_size[first + k];
        if (n > max_block) max_block = n;
        h_st1[k] = st1_bytes; st1_bytes = up(st1_bytes + (uint64_t)n + 1u, 16);
        h_enc[k] = enc_bytes; enc_bytes = up(enc_bytes + crgpu_bound(J->codec, n + (use_dict ? 1u : 0u)), 64);
    }
    if (max_block > CRGPU_MAX_BLOCK) { free(h); snprintf(R->err, sizeof R->err, "block larger than CRGPU_MAX_BLOCK"); return CRGPU_E_ARG; }
    const size_t meta_bytes = n8 * 8u * 4u + ((size_t)count + 2u * per + 8u) * 4u + n8 + 64u;
    rc = dgrow(R, &R->meta, meta_bytes);
    if (rc == CRGPU_OK && use_dict) rc = dgrow(R, &R->st1, (size_t)st1_bytes + 16u);
    if (rc == CRGPU_OK && !prec) rc = dgrow(R, &R->enc, (size_t)enc_bytes + 64u);
    if (rc == CRGPU_OK) rc = dgrow(R, &R->pack, (size_t)((prec ? st1_bytes : enc_bytes) + 6u * (uint64_t)count + 64u));
    if (rc != CRGPU_OK) { free(h); return rc; }
    uint64_t* d_in_off = (uint64_t*)R->meta.p;
    uint64_t* d_st1_off = d_in_off + n8;
    uint64_t* d_enc_off = d_st1_off + n8;
    uint64_t* d_pack_off = d_enc_off + n8;
    uint64_t* d_total = d_pack_off + n8;                         /* 2 x u64 */
    uint32_t* d_in_size = (uint32_t*)(d_total + 2);
    uint32_t* d_len1 = d_in_size + count;                       /* padded to per entries */
    uint32_t* d_len2 = d_len1 + per + 2u;                       /* padded to per entries */
    uint8_t* d_filt = (uint8_t*)(d_len2 + per + 2u);
    hipError_t e = hipMemcpyAsync(d_in_off, h, n8 * 8u * 3u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(d_in_size, J->in_size + first, (size_t)count * 4u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_len1, 0, ((size_t)2u * per + 4u) * 4u, R->stream);
    if (e == hipSuccess && J->per_block && count) e = hipMemcpyAsync(d_filt, J->per_block + first, count, hipMemcpyHostToDevice, R->stream);
    if (e != hipSuccess) { free(h); snprintf(R->err, sizeof R->err, "H2D: %s", hipGetErrorString(e)); return CRGPU_E_NODEVICE; }
    /* the staged copies read h until they have run */
    e = hipStreamSynchronize(R->stream);
    R->t_mark[1] = now_s() - J->t0;                             /* input and tables are on the device */
    free(h);
    if (e != hipSuccess) { snprintf(R->err, sizeof R->err, "H2D: %s", hipGetErrorString(e)); return CRGPU_E_NODEVICE; }

    const uint8_t* cur = R->in.p; const uint64_t* cur_off = d_in_off; const uint32_t* cur_size = d_in_size;
    uint32_t cur_max = max_block;
    if (use_dict && count) {                                    /* src/main.c:189 */
        M_RC(R, crgpu_dict_encode_blocks_dev(R->ctx, R->dict, cur, cur_off, cur_size, count, cur_max, R->st1.p, d_st1_off, d_len1, 0));
        cur = R->st1.p; cur_off = d_st1_off; cur_size = d_len1; cur_max = max_block + 1u;
    }
    if (!prec && count) {                                       /* src/main.c:191-195 */
        M_RC(R, crgpu_encode_blocks_dev(R->ctx, J->codec, cur, cur_off, cur_size, count, cur_max, R->enc.p, d_enc_off, d_len2, 0));
        cur = R->enc.p; cur_off = d_enc_off; cur_size = d_len2;
    }
    /* k_pack: this rank's run of the container, src/main.c:198-205 */
    M_RC(R, crgpu_pack_blocks_dev(R->ctx, cur, cur_off, cur_size, count, J->per_block ? d_filt : NULL, prec, headers, R->pack.p, d_pack_off, d_total, 0));
    uint64_t tot[2] = {0, 0};
    M_HIP(R, hipMemcpyAsync(tot, d_total, 16, hipMemcpyDeviceToHost, R->stream));
    M_HIP(R, hipStreamSynchronize(R->stream));
    if (tot[1]) { snprintf(R->err, sizeof R->err, "%llu block(s) could not be encoded", (unsigned long long)tot[1]); return CRGPU_E_ARG; }
    *my_total = tot[0];
    *d_mine_out = cur_size;                                     /* what the ranks exchange: the size of every block */
    return CRGPU_OK;
}

static int decode_rank(crgpu_multi* m, int r, uint32_t first, uint32_t count, uint32_t per, uint64_t* my_total, const uint32_t** d_mine_out) {
    rank_state* R = &m->rank[r];
    const job* J = &m->j;
    const int use_dict = (J->flags & CRGPU_MULTI_DICT) != 0;
    const uint32_t hdr = header_bytes(J->codec);
    M_HIP(R, hipSetDevice(R->device));
    const size_t n8 = (size_t)(count ? count : 1u);
    uint64_t* h = (uint64_t*)malloc(n8 * (8u * 4u + 4u * 3u));
    if (!h) return CRGPU_E_NOMEM;
    uint64_t *h_in = h, *h_lz_in = h + n8, *h_st1 = h + 2 * n8, *h_d_in = h + 3 * n8;
    uint32_t *h_lz_size = (uint32_t*)(h + 4 * n8), *h_cap1 = h_lz_size + n8, *h_d_size = h_cap1 + n8;
    uint64_t st1_bytes = 0;
    const uint64_t in_bytes = plan_range(J, first, count, h_in);
    int rc = CRGPU_OK;
    /* stage 1 list: the blocks that went through the codec; their decoded size sits in the block header (bytes 4..7,
     * zero for a stored block), src/ropmain/cr-coder.c:59-66 and its siblings */
    uint32_t n1 = 0, max1 = 0;
    for (uint32_t k = 0; k < count && rc == CRGPU_OK; k++) {
        const uint32_t len = J->in_size[first + k];
        if (J->per_block && J->per_block[first + k]) { h_d_in[k] = h_in[k]; h_d_size[k] = len; continue; }   /* m_prec: dictionary stage only */
        if (len < hdr) { rc = CRGPU_E_CORRUPT; break; }
        uint32_t field;
        memcpy(&field, J->in + J->in_off[first + k] + 4, 4);
        const uint32_t want = field ? field : len - hdr;
        if (want > CRGPU_MAX_BLOCK + 1u) { rc = CRGPU_E_CORRUPT; break; }
        if (want > max1) max1 = want;
        h_lz_in[n1] = h_in[k]; h_lz_size[n1] = len; h_cap1[n1] = want;
        h_st1[n1] = in_bytes + st1_bytes;                      /* stage-1 outputs live behind the input in ONE buffer */
        h_d_in[k] = h_st1[n1]; h_d_size[k] = 0xFFFFFFFFu;      /* filled from the device sizes below */
        st1_bytes = up(st1_bytes + want, 16);
        n1++;
    }
    if (rc != CRGPU_OK) { free(h); snprintf(R->err, sizeof R->err, "malformed block header"); return rc; }
    /* one work buffer: [input | stage-1 outputs] */
    rc = dgrow(R, &R->in, (size_t)(in_bytes + st1_bytes + 64u));
    if (rc == CRGPU_OK) rc = upload_range(R, J, first, count, h_in);
    if (rc != CRGPU_OK) { free(h); return rc; }
    const size_t meta_bytes = n8 * 8u * 6u + ((size_t)4u * count + 2u * per + 16u) * 4u + 64u;
    rc = dgrow(R, &R->meta, meta_bytes);
    if (rc != CRGPU_OK) { free(h); return rc; }
    uint64_t* d_lz_in = (uint64_t*)R->meta.p;
    uint64_t* d_st1_off = d_lz_in + n8;
    uint64_t* d_d_in = d_st1_off + n8;
    uint64_t* d_out_off = d_d_in + n8;
    uint64_t* d_total = d_out_off + n8;                         /* 2 x u64 */
    uint32_t* d_lz_size = (uint32_t*)(d_total + 2);
    uint32_t* d_cap1 = d_lz_size + count;
    uint32_t* d_len1 = d_cap1 + count;
    uint32_t* d_d_size = d_len1 + count;
    uint32_t* d_cap2 = d_d_size + count + 2u;                   /* padded to per entries: this is what is exchanged */
    uint32_t* d_len2 = d_cap2 + per + 2u;
    hipError_t e = hipMemcpyAsync(d_lz_in, h_lz_in, n8 * 8u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_st1_off, h_st1, n8 * 8u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d_in, h_d_in, n8 * 8u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(d_lz_size, h_lz_size, (size_t)count * 4u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(d_cap1, h_cap1, (size_t)count * 4u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(d_d_size, h_d_size, (size_t)count * 4u, hipMemcpyHostToDevice, R->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cap2, 0, ((size_t)per + 2u) * 4u, R->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(R->stream);
    R->t_mark[1] = now_s() - J->t0;                             /* input and tables are on the device */
    if (e != hipSuccess) { free(h); snprintf(R->err, sizeof R->err, "H2D: %s", hipGetErrorString(e)); return CRGPU_E_NODEVICE; }
    if (n1) {                                                   /* src/main.c:277 */
        rc = crgpu_decode_blocks_dev(R->ctx, J->codec, R->in.p, d_lz_in, d_lz_size, n1, max1, R->in.p, d_st1_off, d_cap1, d_len1, 0);
        if (rc != CRGPU_OK) { free(h); snprintf(R->err, sizeof R->err, "crgpu_decode_blocks_dev failed (%d): %s", rc, crgpu_last_error(R->ctx)); return rc; }
        /* the sizes stage 1 produced become the input sizes of stage 2 (the codec blocks keep their order) */
        uint32_t* h_len1 = (uint32_t*)malloc((size_t)n1 * 4u);
        if (!h_len1) { free(h); return CRGPU_E_NOMEM; }
        e = hipMemcpyAsync(h_len1, d_len1, (size_t)n1 * 4u, hipMemcpyDeviceToHost, R->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(R->stream);
        uint32_t k1 = 0;
        for (uint32_t k = 0; k < count && e == hipSuccess; k++) {
            if (h_d_size[k] != 0xFFFFFFFFu) continue;
            if (h_len1[k1] == 0xFFFFFFFFu) rc = CRGPU_E_CORRUPT;
            h_d_size[k] = h_len1[k1++];
        }
        free(h_len1);
        if (e == hipSuccess && rc == CRGPU_OK) e = hipMemcpyAsync(d_d_size, h_d_size, (size_t)count * 4u, hipMemcpyHostToDevice, R->stream);
        if (e == hipSuccess && rc == CRGPU_OK) e = hipStreamSynchronize(R->stream);
        if (e != hipSuccess) { free(h); snprintf(R->err, sizeof R->err, "sizes: %s", hipGetErrorString(e)); return CRGPU_E_NODEVICE; }
        if (rc != CRGPU_OK) { free(h); snprintf(R->err, sizeof R->err, "a block did not decode"); return rc; }
    }
    free(h);
    if (use_dict) {
        /* src/main.c:281: sizes first (recorded at the end of the pieces), so that the outputs can be laid back to back */
        M_RC(R, crgpu_dict_decoded_sizes_dev(R->ctx, R->in.p, d_d_in, d_d_size, count, d_cap2, 0));
    } else if (count) {
        M_HIP(R, hipMemcpyAsync(d_cap2, d_d_size, (size_t)count * 4u, hipMemcpyDeviceToDevice, R->stream));
    }
    M_RC(R, crgpu_offsets_dev(R->ctx, d_cap2, count, d_out_off, d_total, 0));
    uint64_t tot[2] = {0, 0};
    M_HIP(R, hipMemcpyAsync(tot, d_total, 16, hipMemcpyDeviceToHost, R->stream));
    M_HIP(R, hipStreamSynchronize(R->stream));
    if (tot[1]) { snprintf(R->err, sizeof R->err, "%llu malformed dictionary-stage block(s)", (unsigned long long)tot[1]); return CRGPU_E_CORRUPT; }
    M_RC(R, dgrow(R, &R->pack, (size_t)tot[0] + 64u));
    if (use_dict && count) {
        uint32_t max2 = CRGPU_MAX_BLOCK;                        /* the device tables are laid out per launch; the sizes are on the device */
        int drc = crgpu_dict_decode_blocks_dev(R->ctx, R->dict, R->in.p, d_d_in, d_d_size, count, max2, R->pack.p, d_out_off, d_cap2, d_len2, 0);
        if (drc != CRGPU_OK) { snprintf(R->err, sizeof R->err, "crgpu_dict_decode_blocks_dev failed (%d): %s", drc, crgpu_last_error(R->ctx)); return drc; }
        /* every block must have produced exactly the size its pieces recorded */
        uint32_t* chk = (uint32_t*)malloc((size_t)count * 8u);
        if (!chk) return CRGPU_E_NOMEM;
        hipError_t e3 = hipMemcpyAsync(chk, d_cap2, (size_t)count * 4u, hipMemcpyDeviceToHost, R->stream);
        if (e3 == hipSuccess) e3 = hipMemcpyAsync(chk + count, d_len2, (size_t)count * 4u, hipMemcpyDeviceToHost, R->stream);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(R->stream);
        int bad = e3 != hipSuccess;
        for (uint32_t k = 0; k < count && !bad; k++) bad = chk[k] != chk[count + k];
        free(chk);
        if (bad) { snprintf(R->err, sizeof R->err, "a dictionary-stage block did not decode to its recorded size"); return CRGPU_E_CORRUPT; }
    } else if (count) {
        M_RC(R, crgpu_pack_blocks_dev(R->ctx, R->in.p, d_d_in, d_d_size, count, NULL, 0, 0, R->pack.p, d_out_off, d_total, 0));
    }
    *my_total = tot[0];
    *d_mine_out = d_cap2;
    return CRGPU_OK;
}

static void run_rank(crgpu_multi* m, int r) {
    rank_state* R = &m->rank[r];
    job* J = &m->j;
    R->err[0] = 0;
    uint32_t first = 0, count = 0;
    crgpu_shard_range(J->nblocks, m->ndev, r, &first, &count);
    const uint32_t per = (J->nblocks + (uint32_t)m->ndev - 1u) / (uint32_t)m->ndev;
    uint64_t my_total = 0;
    const uint32_t* d_mine = NULL;
    for (int i = 0; i < CRGPU_MULTI_TIMES; i++) R->t_mark[i] = 0.0;
    if (m->test_stall == r) for (;;) { struct timespec ts = {1, 0}; nanosleep(&ts, NULL); }    /* (tests) the rank that never arrives */
    __atomic_store_n(&R->where, RANK_STAGES, __ATOMIC_SEQ_CST);
    R->rc = J->decode ? decode_rank(m, r, first, count, per, &my_total, &d_mine) : encode_rank(m, r, first, count, per, &my_total, &d_mine);
    if (R->rc != CRGPU_OK) {
        __atomic_store_n(&J->failed, 1, __ATOMIC_SEQ_CST);
        my_total = 0;
        d_mine = (const uint32_t*)R->sizes.p;                    /* zeros, prepared by run_job */
    }
    R->t_mark[2] = now_s() - J->t0;                             /* stages done (their last synchronisation) */
    __atomic_store_n(&R->where, RANK_EXCHANGE, __ATOMIC_SEQ_CST);
    const int xrc = exchange_sizes(m, r, d_mine, per);
    if (xrc != CRGPU_OK && R->rc == CRGPU_OK) { R->rc = xrc; __atomic_store_n(&J->failed, 1, __ATOMIC_SEQ_CST); }
    __atomic_store_n(&R->where, RANK_AT_BARRIER, __ATOMIC_SEQ_CST);
    pthread_barrier_wait(&m->bar);                              /* every size is known everywhere */
    R->t_mark[3] = now_s() - J->t0;
    const uint32_t* all = m->use_rccl ? R->h_all : J->host_all;
    const int headers = !J->decode && (J->flags & CRGPU_MULTI_HEADERS);
    if (r == 0 && !__atomic_load_n(&J->failed, __ATOMIC_SEQ_CST)) {
        /* the table is rank-major and a rank's entries are its contiguous block range: that IS block order */
        J->out_total = crgpu_container_offsets(all, J->nblocks, headers, J->out_off);
        if (J->out_size) memcpy(J->out_size, all, (size_t)J->nblocks * 4u);
        if (m->pinned_out) {
            /* the context's page-locked pool: the copy-out below is then one DMA at the link's rate instead of a staged copy
             * into fresh pageable memory (1e8 bytes: ~2 ms instead of ~8); the result stays valid until the next job */
            if (m->pool_cap < J->out_total) {
                if (m->pool) { pool_forget(m->pool); (void)hipHostFree(m->pool); m->pool = NULL; m->pool_cap = 0; }
                const size_t want = (size_t)(J->out_total + J->out_total / 4u + 65536u);
                void* p = NULL;
                if (hipSetDevice(R->device) == hipSuccess && hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess) {
                    if (pool_remember(p)) { m->pool = (uint8_t*)p; m->pool_cap = want; } else (void)hipHostFree(p);
                }
            }
            J->out = m->pool_cap >= J->out_total && m->pool ? m->pool : (uint8_t*)malloc(J->out_total ? J->out_total : 1u);   /* (no pool to be had: pageable, the caller's to free) */
        } else {
            J->out = (uint8_t*)malloc(J->out_total ? J->out_total : 1u);
        }
        if (!J->out) {
            R->rc = CRGPU_E_NOMEM;
            snprintf(R->err, sizeof R->err, "%s(%llu) failed", m->pinned_out ? "hipHostMalloc" : "malloc", (unsigned long long)J->out_total);
            __atomic_store_n(&J->failed, 1, __ATOMIC_SEQ_CST);
        }
    }
    pthread_barrier_wait(&m->bar);                              /* the output exists */
    __atomic_store_n(&R->where, RANK_COPY_OUT, __ATOMIC_SEQ_CST);
    R->t_mark[4] = now_s() - J->t0;
    if (!__atomic_load_n(&J->failed, __ATOMIC_SEQ_CST) && my_total) {
        const uint64_t base = crgpu_container_offsets(all, first, headers, NULL);
        hipError_t e = hipSetDevice(R->device);
        if (e == hipSuccess) e = hipMemcpyAsync(J->out + base, R->pack.p, (size_t)my_total, hipMemcpyDeviceToHost, R->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(R->stream);
        if (e != hipSuccess) { R->rc = CRGPU_E_NODEVICE; snprintf(R->err, sizeof R->err, "D2H: %s", hipGetErrorString(e)); }
    }
    R->t_mark[5] = now_s() - J->t0;
    __atomic_store_n(&R->where, RANK_DONE, __ATOMIC_SEQ_CST);
}

static void* worker_main(void* argp) {
    thread_arg* a = (thread_arg*)argp;
    crgpu_multi* m = a->m;
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&m->mu);
        while (m->seq == seen && !m->quit) pthread_cond_wait(&m->cv_go, &m->mu);
        if (m->quit) { pthread_mutex_unlock(&m->mu); break; }
        seen = m->seq;
        pthread_mutex_unlock(&m->mu);
        run_rank(m, a->r);
        pthread_mutex_lock(&m->mu);
        if (++m->done == m->ndev) pthread_cond_signal(&m->cv_done);
        pthread_mutex_unlock(&m->mu);
    }
    return NULL;
}

static int run_job(crgpu_multi* m) {
    job* J = &m->j;
    const uint32_t per = (J->nblocks + (uint32_t)m->ndev - 1u) / (uint32_t)m->ndev;
    const size_t table = (size_t)per * (size_t)m->ndev;
    J->host_all = NULL; J->out = NULL; J->out_total = 0; J->failed = 0;
    if (m->broken) { snprintf(m->err, sizeof m->err, "this multi-GPU context was abandoned after a job missed its deadline"); return CRGPU_E_NODEVICE; }
    m->err[0] = 0;
    int caller_device = -1;                                     /* the preparation below changes the calling thread's device */
    if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
    struct restore_device { int d; ~restore_device() { if (d >= 0) (void)hipSetDevice(d); } } restore_{caller_device};
    /* everything the exchange needs exists before a rank starts, so that a rank can always take part in it */
    if (!m->use_rccl) {
        J->host_all = (uint32_t*)calloc(table + 1u, 4);
        if (!J->host_all) return CRGPU_E_NOMEM;
    }
    for (int r = 0; r < m->ndev; r++) {
        rank_state* R = &m->rank[r];
        R->rc = CRGPU_OK;
        int rc = CRGPU_OK;
        if (hipSetDevice(R->device) != hipSuccess) rc = CRGPU_E_NODEVICE;
        if (rc == CRGPU_OK) rc = dgrow(R, &R->sizes, (size_t)per * 4u + 16u);
        if (rc == CRGPU_OK && hipMemsetAsync(R->sizes.p, 0, (size_t)per * 4u + 16u, R->stream) != hipSuccess) rc = CRGPU_E_NODEVICE;
        if (rc == CRGPU_OK && m->use_rccl) {
            rc = dgrow(R, &R->all, table * 4u + 16u);
            if (rc == CRGPU_OK && R->h_all_cap < table) {
                free(R->h_all);
                R->h_all = (uint32_t*)malloc((table ? table : 1u) * 4u);
                R->h_all_cap = R->h_all ? table : 0;
                if (!R->h_all) rc = CRGPU_E_NOMEM;
            }
        }
        if (rc != CRGPU_OK) {
            snprintf(m->err, sizeof m->err, "device %d: could not prepare the size exchange (%d)", R->device, rc);
            free(J->host_all); J->host_all = NULL;
            return rc;
        }
    }
    J->t0 = now_s();
    for (int r = 0; r < m->ndev; r++) __atomic_store_n(&m->rank[r].where, RANK_IDLE, __ATOMIC_SEQ_CST);
    pthread_mutex_lock(&m->mu);
    m->done = 0;
    m->seq++;
    pthread_cond_broadcast(&m->cv_go);
    /* The deadline: a rank that never reaches the exchange (a kernel that does not end, a device that fell off the bus, a
     * peer missing from the collective) leaves every other rank waiting in ncclAllGather or at the barrier for ever. The
     * caller does not wait with them: after deadline_s the job is given up, the report names the ranks that did not arrive,
     * and the context is marked broken — its threads may never come back, nothing of it is touched again. */
    int timed_out = 0;
    if (m->deadline_s > 0.0) {
        struct timespec until;
        clock_gettime(CLOCK_MONOTONIC, &until);                 /* (cv_done is created on CLOCK_MONOTONIC: a wall-clock step must not end a healthy job) */
        const double whole = (double)(long)m->deadline_s;
        until.tv_sec += (time_t)whole;
        until.tv_nsec += (long)((m->deadline_s - whole) * 1e9);
        if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
        while (m->done < m->ndev && !timed_out) timed_out = pthread_cond_timedwait(&m->cv_done, &m->mu, &until) != 0 && m->done < m->ndev;
    } else {
        while (m->done < m->ndev) pthread_cond_wait(&m->cv_done, &m->mu);
    }
    if (timed_out) m->broken = 1;
    pthread_mutex_unlock(&m->mu);
    if (timed_out) {
        /* J->host_all and the ranks' buffers stay allocated: the abandoned threads may still write to them */
        int n = snprintf(m->err, sizeof m->err, "deadline of %.0f s passed:", m->deadline_s);
        int late = 0;
        for (int r = 0; r < m->ndev && n < (int)sizeof m->err - 1; r++) {
            const int w = __atomic_load_n(&m->rank[r].where, __ATOMIC_SEQ_CST);
            if (w < RANK_AT_BARRIER) { n += snprintf(m->err + n, sizeof m->err - (size_t)n, " rank %d (device %d) did not arrive, stuck in %s;", r, m->rank[r].device, rank_where[w]); late++; }
        }
        if (!late && n < (int)sizeof m->err - 1) snprintf(m->err + n, sizeof m->err - (size_t)n, " every rank reached the barrier, the copy-out did not end");
        J->out = NULL;
        return CRGPU_E_NODEVICE;
    }
    free(J->host_all); J->host_all = NULL;
    int rc = CRGPU_OK;
    for (int r = 0; r < m->ndev; r++) if (m->rank[r].rc != CRGPU_OK && rc == CRGPU_OK) {
        rc = m->rank[r].rc;
        snprintf(m->err, sizeof m->err, "device %d (rank %d): %s", m->rank[r].device, r, m->rank[r].err);
    }
    if (rc != CRGPU_OK) { if (J->out != m->pool) free(J->out); J->out = NULL; }
    return rc;
}

/* tests only (not declared in include/crgpu.h): rank `rank` never starts its next jobs, -1 = none */
extern "C" void crgpu_multi_test_stall_rank(crgpu_multi* m, int rank) { if (m) m->test_stall = rank; }

extern "C" const char* crgpu_multi_last_error(const crgpu_multi* m) { return m ? m->err : "no multi-GPU context"; }
extern "C" int crgpu_multi_devices(const crgpu_multi* m) { return m ? m->ndev : 0; }
extern "C" int crgpu_multi_timing(const crgpu_multi* m, int rank, double* seconds, int room) {
    if (!m || rank < 0 || rank >= m->ndev || !seconds || room < 0) return -1;
    for (int i = 0; i < room && i < CRGPU_MULTI_TIMES; i++) seconds[i] = m->rank[rank].t_mark[i];
    return CRGPU_MULTI_TIMES;
}
extern "C" int crgpu_multi_uses_rccl(const crgpu_multi* m) { return m ? m->use_rccl : 0; }

/* grow the page-locked pool of a CRGPU_MULTI_PINNED_OUT context ahead of the first job (page-locking 100 MB takes tens of
 * milliseconds: a command line does it while it is busy elsewhere) */
extern "C" int crgpu_multi_reserve_output(crgpu_multi* m, uint64_t bytes) {
    if (!m) return CRGPU_E_ARG;
    if (!m->pinned_out || m->pool_cap >= bytes) return CRGPU_OK;
    if (m->pool) { pool_forget(m->pool); (void)hipHostFree(m->pool); m->pool = NULL; m->pool_cap = 0; }
    void* p = NULL;
    if (hipSetDevice(m->rank[0].device) != hipSuccess || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return CRGPU_E_NOMEM;
    if (!pool_remember(p)) { (void)hipHostFree(p); return CRGPU_E_NOMEM; }
    m->pool = (uint8_t*)p; m->pool_cap = (size_t)bytes;
    return CRGPU_OK;
}

extern "C" int crgpu_multi_set_deadline(crgpu_multi* m, double seconds) {
    if (!m || !(seconds >= 0.0)) return CRGPU_E_ARG;
    m->deadline_s = seconds;
    return CRGPU_OK;
}

extern "C" void crgpu_multi_destroy(crgpu_multi* m) {
    if (!m) return;
    if (m->broken) {
        /* worker threads of the abandoned job may sit in a collective or behind a kernel that never ends: joining them, or
         * freeing what they use, would hang or crash. They are left alone and the context's memory is leaked on purpose. */
        for (int r = 0; r < m->nthreads; r++) (void)pthread_detach(m->thread[r]);
        return;
    }
    if (m->sync_ok) {
        pthread_mutex_lock(&m->mu);
        m->quit = 1;
        pthread_cond_broadcast(&m->cv_go);
        pthread_mutex_unlock(&m->mu);
        for (int r = 0; r < m->nthreads; r++) pthread_join(m->thread[r], NULL);
        pthread_mutex_destroy(&m->mu); pthread_cond_destroy(&m->cv_go); pthread_cond_destroy(&m->cv_done);
    }
    for (int r = 0; r < m->ndev; r++) {
        rank_state* R = &m->rank[r];
        if (R->ctx) (void)hipSetDevice(R->device);
        if (R->stream) (void)hipStreamSynchronize(R->stream);
        if (m->use_rccl && R->comm) (void)m->rccl.CommDestroy(R->comm);
        dev_buf* bufs[] = {&R->in, &R->st1, &R->enc, &R->pack, &R->meta, &R->sizes, &R->all};
        for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; i++) if (bufs[i]->p) (void)hipFree(bufs[i]->p);
        if (R->dict) crgpu_dict_destroy(R->dict);
        if (R->ctx) crgpu_destroy(R->ctx);
        if (R->stream) (void)hipStreamDestroy(R->stream);
        free(R->h_all);
    }
    if (m->pool) { pool_forget(m->pool); (void)hipHostFree(m->pool); }
    if (m->bar_ok) pthread_barrier_destroy(&m->bar);
    if (m->rccl.lib) dlclose(m->rccl.lib);
    free(m);
}

extern "C" int crgpu_multi_create(crgpu_multi** out, const int* devices, int ndev, int flags) {
    if (!out) return CRGPU_E_ARG;
    *out = NULL;
    if (!devices || ndev < 1 || ndev > MULTI_MAX) return CRGPU_E_ARG;
    crgpu_multi* m = (crgpu_multi*)calloc(1, sizeof *m);
    if (!m) return CRGPU_E_NOMEM;
    m->ndev = ndev;
    m->pinned_out = (flags & CRGPU_MULTI_PINNED_OUT) != 0;
    m->deadline_s = 120.0;                                      /* per job; crgpu_multi_set_deadline / CRGPU_MULTI_DEADLINE_S */
    m->test_stall = -1;
    if (const char* e = getenv("CRGPU_MULTI_DEADLINE_S")) { char* end = NULL; const double v = strtod(e, &end); if (end != e && v >= 0.0) m->deadline_s = v; }
    /* (the rank that never starts — the deadline tests' hook — is set through crgpu_multi_test_stall_rank, not the environment:
     * a process that merely inherits a variable must not get a spinning rank) */
    int distinct = 1;
    for (int r = 0; r < ndev; r++) for (int q = 0; q < r; q++) if (devices[q] == devices[r]) distinct = 0;
    int rc = CRGPU_OK;
    for (int r = 0; r < ndev && rc == CRGPU_OK; r++) {
        rank_state* R = &m->rank[r];
        R->device = devices[r];
        rc = crgpu_create(&R->ctx, devices[r]);
        if (rc != CRGPU_OK) { R->ctx = NULL; break; }
        if (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&R->stream, hipStreamNonBlocking) != hipSuccess) { rc = CRGPU_E_NODEVICE; break; }
        rc = crgpu_set_stream(R->ctx, R->stream);               /* kernels, copies and the collective share one stream per rank */
    }
    if (rc == CRGPU_OK && pthread_barrier_init(&m->bar, NULL, (unsigned)ndev) != 0) rc = CRGPU_E_NOMEM;
    if (rc == CRGPU_OK) m->bar_ok = 1;
    /* one device has nobody to exchange with: the table stays in host memory and librccl is not loaded, unless
     * CRGPU_MULTI_RCCL asks for the communicator of one (tests: the collective's code path on a one-GPU box) */
    if (rc == CRGPU_OK && distinct && !(flags & CRGPU_MULTI_HOST_GATHER) && (ndev > 1 || (flags & CRGPU_MULTI_RCCL))) {
        rc = rccl_open(&m->rccl, m->err, sizeof m->err);
        if (rc == CRGPU_OK) {
            ncclComm_t comms[MULTI_MAX];
            const ncclResult_t e = m->rccl.CommInitAll(comms, ndev, devices);
            if (e != ncclSuccess) { snprintf(m->err, sizeof m->err, "ncclCommInitAll: %s", m->rccl.GetErrorString(e)); rc = CRGPU_E_NODEVICE; }
            else { for (int r = 0; r < ndev; r++) m->rank[r].comm = comms[r]; m->use_rccl = 1; }
        }
    }
    if (rc == CRGPU_OK) {
        pthread_condattr_t ca;                                   /* the deadline wait runs on the monotonic clock */
        const int ca_ok = pthread_condattr_init(&ca) == 0 && pthread_condattr_setclock(&ca, CLOCK_MONOTONIC) == 0;
        if (!ca_ok || pthread_mutex_init(&m->mu, NULL) != 0 || pthread_cond_init(&m->cv_go, NULL) != 0 || pthread_cond_init(&m->cv_done, &ca) != 0) rc = CRGPU_E_NOMEM;
        else m->sync_ok = 1;
        if (ca_ok) pthread_condattr_destroy(&ca);
    }
    for (int r = 0; r < ndev && rc == CRGPU_OK; r++) {
        m->targ[r].m = m; m->targ[r].r = r;
        if (pthread_create(&m->thread[r], NULL, worker_main, &m->targ[r]) != 0) { snprintf(m->err, sizeof m->err, "pthread_create failed"); rc = CRGPU_E_NOMEM; }
        else m->nthreads++;
    }
    if (rc != CRGPU_OK) { crgpu_multi_destroy(m); return rc; }
    *out = m;
    return CRGPU_OK;
}

extern "C" int crgpu_multi_set_dictionary(crgpu_multi* m, const char* dictionary_text) {
    if (!m || !dictionary_text) return CRGPU_E_ARG;
    for (int r = 0; r < m->ndev; r++) {                         /* read-only per-file input: built on the host, uploaded to every GPU */
        rank_state* R = &m->rank[r];
        if (R->dict) { crgpu_dict_destroy(R->dict); R->dict = NULL; }
        const int rc = crgpu_dict_create(R->ctx, dictionary_text, &R->dict);
        if (rc != CRGPU_OK) { snprintf(m->err, sizeof m->err, "crgpu_dict_create on device %d failed (%d): %s", R->device, rc, crgpu_last_error(R->ctx)); return rc; }
    }
    return CRGPU_OK;
}

extern "C" int crgpu_multi_configure(crgpu_multi* m, uint32_t rox_chain_limit, int flexible) {
    if (!m) return CRGPU_E_ARG;
    for (int r = 0; r < m->ndev; r++) {
        int rc = rox_chain_limit ? crgpu_rox_set_chain_limit(m->rank[r].ctx, rox_chain_limit) : CRGPU_OK;
        if (rc == CRGPU_OK) rc = crgpu_set_flexible_parsing(m->rank[r].ctx, flexible);
        if (rc != CRGPU_OK) return rc;
    }
    return CRGPU_OK;
}

static int check_job(crgpu_multi* m, int codec, int flags, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                     uint32_t nblocks, uint8_t** out, uint64_t* out_total) {
    if (!m || !out || !out_total) return CRGPU_E_ARG;
    *out = NULL; *out_total = 0;
    /* a context that gave a job up keeps that job's struct and buffers for its abandoned threads (they may only be slow, not
     * stuck): nothing of it is written again — checked here, in front of the entry points' memset of the job */
    if (m->broken) { snprintf(m->err, sizeof m->err, "this multi-GPU context was abandoned after a job missed its deadline"); return CRGPU_E_NODEVICE; }
    if (nblocks && (!in || !in_off || !in_size)) return CRGPU_E_ARG;
    if (codec != CRGPU_CODEC_ROP && codec != CRGPU_CODEC_ROX && codec != CRGPU_CODEC_ROLZ) return CRGPU_E_ARG;
    if ((flags & CRGPU_MULTI_DICT) && !m->rank[0].dict) { snprintf(m->err, sizeof m->err, "CRGPU_MULTI_DICT without crgpu_multi_set_dictionary"); return CRGPU_E_ARG; }
    if ((flags & CRGPU_MULTI_PREC) && !(flags & CRGPU_MULTI_DICT)) { snprintf(m->err, sizeof m->err, "CRGPU_MULTI_PREC needs CRGPU_MULTI_DICT"); return CRGPU_E_ARG; }
    return CRGPU_OK;
}

extern "C" int crgpu_multi_encode_blocks(crgpu_multi* m, int codec, int flags,
                                         const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size, uint32_t nblocks,
                                         const uint8_t* filt, uint8_t** out, uint64_t* out_total, uint64_t* out_off, uint32_t* out_size) {
    int rc = check_job(m, codec, flags, in, in_off, in_size, nblocks, out, out_total);
    if (rc != CRGPU_OK) return rc;
    job* J = &m->j;
    memset(J, 0, sizeof *J);
    J->decode = 0; J->codec = codec; J->flags = flags; J->in = in; J->in_off = in_off; J->in_size = in_size;
    J->per_block = filt; J->nblocks = nblocks; J->out_off = out_off; J->out_size = out_size;
    rc = run_job(m);
    if (rc == CRGPU_OK) { *out = J->out; *out_total = J->out_total; }
    return rc;
}

extern "C" int crgpu_multi_decode_blocks(crgpu_multi* m, int codec, int flags,
                                         const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size, uint32_t nblocks,
                                         const uint8_t* prec, uint8_t** out, uint64_t* out_total, uint64_t* out_off, uint32_t* out_size) {
    int rc = check_job(m, codec, flags & ~CRGPU_MULTI_PREC, in, in_off, in_size, nblocks, out, out_total);
    if (rc != CRGPU_OK) return rc;
    job* J = &m->j;
    memset(J, 0, sizeof *J);
    J->decode = 1; J->codec = codec; J->flags = flags; J->in = in; J->in_off = in_off; J->in_size = in_size;
    J->per_block = prec; J->nblocks = nblocks; J->out_off = out_off; J->out_size = out_size;
    rc = run_job(m);
    if (rc == CRGPU_OK) { *out = J->out; *out_total = J->out_total; }
    return rc;
}

extern "C" void crgpu_multi_free(void* p) { if (p && !pool_known(p)) free(p); }

/*
 * comprox_amd/csrc/crhost_dict.c — host-side (CPU) parts of the static-dictionary stage, plain C.
 *
 * These run once per FILE, not per datablock, and stay on the host exactly as SURVEY.md §8 a16
 * prescribes: the whole-file word census that picks the dictionary (reference src/cr-dicpick.c
 * :164-273) and the front-coding of the dictionary blob (:275-346). Their output feeds the GPU:
 * the dictionary text goes to crgpu_dict_create(), the front-coded blob through lzencode().
 * Exported with the reference's own names and signatures (src/cr-dicpick.h:40-42).
 */
#include <ctype.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/crgpu.h"

#define CRH_MAXWORDS   25000                               /* cr-diccode.h:39 */
#define CRH_LEVEL1(n)  ((65535 - (int)(n)) / 255 - 1)      /* cr-diccode.h:40 */
#define CRH_WORD_MIN   2
#define CRH_WORD_MAX   20
#define CRH_MAP_FULL   (CRH_MAXWORDS * 13 + 1)             /* cr-dicpick.c:33 */
#define CRH_MAP_SLOTS  (CRH_MAXWORDS * 23 + 3)             /* cr-dicpick.c:34 */
#define CRH_KEEP_ABOVE 5                                   /* cr-dicpick.c:35 */
#define CRH_CHUNK      200000                              /* cr-dicpick.c:162 */

typedef struct { char text[CRH_WORD_MAX + 1]; int hits; } crh_cell;

/* The reference classifies bytes with isalpha / islower / tolower and never calls setlocale, i.e. in the "C" locale:
 * ASCII letters only. The census looks at every byte of the file, so the classes are table lookups here (bit 0 letter,
 * bit 1 lower-case letter) and a word's hash is taken while its letters are copied. */
static unsigned char crh_class[256];
static unsigned char crh_low[256];
static void crh_tables(void) {
    if (crh_class['a']) return;
    for (int c = 0; c < 256; c++) {
        const int up = c >= 'A' && c <= 'Z', lo = c >= 'a' && c <= 'z';
        crh_low[c] = (unsigned char)(up ? c + 32 : c);
        crh_class[c] = (unsigned char)((up || lo ? 1 : 0) | (lo ? 2 : 0));
    }
}
static int crh_letter(int c) { return crh_class[(unsigned char)c] & 1; }

static unsigned crh_hash(const char* w) {                  /* cr-dicpick.c:71-78 */
    unsigned h = 0;
    for (; crh_letter(*w); w++) h = h * 131313131u + (unsigned)crh_low[(unsigned char)*w];
    return h & 0x7fffffffu;
}

static void crh_lower(char* dst, const char* src) {        /* cr-dicpick.c:89-95 */
    for (; crh_letter(*src); src++) *dst++ = (char)crh_low[(unsigned char)*src];
    *dst = 0;
}

/* `w` is a lower-cased word (letters only, NUL-terminated), as every text in the map is: cr-dicpick.c:79-88's
 * case-blind comparison up to the first non-letter is strcmp on such strings */
static unsigned crh_slot_hashed(const crh_cell* map, const char* w, unsigned hash) {
    unsigned at = hash % CRH_MAP_SLOTS;
    while (map[at].hits > 0 && strcmp(map[at].text, w) != 0) { at++; if (at == CRH_MAP_SLOTS) at = 0; }
    return at;
}
static unsigned crh_slot(const crh_cell* map, const char* w) { return crh_slot_hashed(map, w, crh_hash(w)); }

/* cr-dicpick.c:96-143: when the map reaches its fill limit, everything within 5 hits of the least
 * used word is forgotten and the survivors are re-inserted in the reference's order */
static void crh_count(crh_cell* map, int* live, const char* w, unsigned hash) {
    unsigned at = crh_slot_hashed(map, w, hash);
    if (map[at].hits > 0) { map[at].hits++; return; }
    strcpy(map[at].text, w);
    map[at].hits = 1;
    *live += 1;
    if (*live != CRH_MAP_FULL) return;

    crh_cell* saved = (crh_cell*)malloc(sizeof(crh_cell) * CRH_MAP_FULL);
    int floor_hits = INT_MAX, top = *live;
    for (unsigned i = 0; i < CRH_MAP_SLOTS; i++) {
        if (map[i].hits > 0) {
            if (map[i].hits < floor_hits) floor_hits = map[i].hits;
            saved[--top] = map[i];
        }
        map[i].hits = 0;
    }
    *live = 0;
    for (int i = 0; i < CRH_MAP_FULL; i++) {
        if (saved[i].hits <= floor_hits + 5) continue;
        unsigned t = crh_slot(map, saved[i].text);
        strcpy(map[t].text, saved[i].text);
        map[t].hits = saved[i].hits;
        *live += 1;
    }
    free(saved);
}

static int crh_by_hits(const void* pa, const void* pb) {   /* cr-dicpick.c:58-65 */
    const crh_cell* a = (const crh_cell*)pa;
    const crh_cell* b = (const crh_cell*)pb;
    return a->hits != b->hits ? b->hits - a->hits : strcmp(b->text, a->text);
}
static int crh_by_text(const void* pa, const void* pb) {   /* cr-dicpick.c:53-57 */
    return strcmp(((const crh_cell*)pa)->text, ((const crh_cell*)pb)->text);
}

/* dicpick(), cr-dicpick.c:164-273: reads `fp` to its end in 200 000-byte chunks */
void dicpick(FILE* fp, data_block_t* dic_block) {
    crh_cell* map = (crh_cell*)calloc(CRH_MAP_SLOTS, sizeof(crh_cell));
    unsigned char* buf = (unsigned char*)malloc(CRH_CHUNK);
    unsigned char closes_word[256] = {0};
    int live = 0, got;
    closes_word[' '] = closes_word[','] = closes_word['.'] = closes_word[':'] = closes_word[';'] = 1;
    crh_tables();
    /* the map is 16 MB and a word's home cell is a cache miss more often than not: words wait in a short queue with
     * their cell's line requested, and are counted in the order they were found (the census is order-dependent) */
    enum { CRH_QUEUE = 8 };
    struct { char w[CRH_WORD_MAX + 2]; unsigned hash; } queue[CRH_QUEUE];
    unsigned q_head = 0, q_count = 0;

    while ((got = (int)fread(buf, 1, CRH_CHUNK, fp)) > 0) {
        buf[got - 1] = 0;                                  /* cr-dicpick.c:192 */
        for (int x = 1; x < got; x++) {
            if (!(crh_class[buf[x]] & 1) || (crh_class[buf[x - 1]] & 1)) continue;
            int y = x + 1;
            while (y < got && (crh_class[buf[y]] & 2)) y++;
            if (y - x >= CRH_WORD_MIN && y - x <= CRH_WORD_MAX && closes_word[buf[y]]) {
                /* the word: a letter and lower-case letters up to a closing byte; lower-cased and hashed in one pass
                 * (cr-dicpick.c:71-78 hashes the lower-cased letters) */
                if (q_count == CRH_QUEUE) {
                    crh_count(map, &live, queue[q_head].w, queue[q_head].hash);
                    q_head = (q_head + 1) % CRH_QUEUE;
                    q_count--;
                }
                char* w = queue[(q_head + q_count) % CRH_QUEUE].w;
                unsigned h = 0;
                for (int k = x; k < y; k++) { const unsigned char c = crh_low[buf[k]]; w[k - x] = (char)c; h = h * 131313131u + c; }
                w[y - x] = 0;
                h &= 0x7fffffffu;
                queue[(q_head + q_count) % CRH_QUEUE].hash = h;
                q_count++;
                __builtin_prefetch(&map[h % CRH_MAP_SLOTS]);
            }
            x = y;
        }
    }
    for (; q_count; q_count--, q_head = (q_head + 1) % CRH_QUEUE) crh_count(map, &live, queue[q_head].w, queue[q_head].hash);

    int kept = 0;                                          /* cr-dicpick.c:219-228 */
    for (unsigned i = 0; i < CRH_MAP_SLOTS; i++) {
        if (map[i].hits > CRH_KEEP_ABOVE) {
            crh_cell c = map[i];
            crh_lower(map[kept].text, c.text);
            map[kept].hits = c.hits;
            kept++;
        }
    }
    qsort(map, (size_t)kept, sizeof(crh_cell), crh_by_hits);
    const char* reserved[2] = {"\x20\x20", "http://www."}; /* cr-dicpick.c:38-41 */
    const int nres = 2;
    if (kept > CRH_MAXWORDS - nres) kept = CRH_MAXWORDS - nres;
    if (kept > CRH_LEVEL1(kept) - nres) {                  /* words with 2-byte codes go in name order */
        int first = CRH_LEVEL1(kept) - nres;
        qsort(map + first, (size_t)(kept - first), sizeof(crh_cell), crh_by_text);
    }
    data_block_reserve(dic_block, (uint32_t)((live + nres) * (CRH_WORD_MAX + 3)));
    for (int r = 0; r < nres; r++) {
        for (const char* p = reserved[r]; *p; p++) data_block_add(dic_block, (uint8_t)*p);
        data_block_add(dic_block, '\n');
    }
    for (int i = 0; i < kept; i++) {
        if (i >= CRH_LEVEL1(kept) && strlen(map[i].text) < CRH_WORD_MIN + 1) continue;   /* cr-dicpick.c:254 */
        for (const char* p = map[i].text; *p; p++) data_block_add(dic_block, (uint8_t)*p);
        data_block_add(dic_block, '\n');
    }
    data_block_add(dic_block, 0);
    free(buf);
    free(map);
}

/* dic_lcp_encode(), cr-dicpick.c:275-316 */
void dic_lcp_encode(data_block_t* dic_block) {
    data_block_t out = {0, 0, 0};
    const uint8_t* t = dic_block->m_data;
    uint32_t prev = 0, cur = 0;
    while (t[cur] != '\n') data_block_add(&out, t[cur++]);
    cur++;
    data_block_add(&out, '\n');
    while (t[cur] != 0) {
        uint32_t shared = 0;
        while (t[prev + shared] == t[cur + shared]) shared++;
        data_block_add(&out, (uint8_t)shared);
        prev = cur;
        for (cur += shared; t[cur] != '\n'; cur++) data_block_add(&out, t[cur]);
        cur++;
        data_block_add(&out, '\n');
    }
    data_block_add(&out, 255);
    data_block_resize(dic_block, out.m_size);
    memcpy(dic_block->m_data, out.m_data, out.m_size);
    data_block_destroy(&out);
}

/* dic_lcp_decode(), cr-dicpick.c:318-346. The blob comes out of a file: every scan is bounded by the block's size
 * (the reference trusts the terminators). A malformed blob leaves an EMPTY block (m_size 0) for the caller to refuse. */
void dic_lcp_decode(data_block_t* dic_block) {
    data_block_t out = {0, 0, 0};
    const uint8_t* b = dic_block->m_data;
    const uint32_t n = dic_block->m_size;
    uint32_t r = 0, prev = 0;
    int ok = 0;
    while (r < n && b[r] != '\n') data_block_add(&out, b[r++]);
    if (r >= n) goto done;
    r++;
    data_block_add(&out, '\n');
    for (;;) {
        if (r >= n) goto done;
        if (b[r] == 255) break;
        for (uint32_t shared = b[r++]; shared; shared--) {
            if (prev >= out.m_size || out.m_data[prev] == '\n') goto done;      /* shares more than the previous word has */
            data_block_add(&out, out.m_data[prev++]);
        }
        while (r < n && b[r] != '\n') data_block_add(&out, b[r++]);
        if (r >= n) goto done;
        r++;
        data_block_add(&out, '\n');
        while (prev < out.m_size && out.m_data[prev] != '\n') prev++;
        prev++;                                                              /* start of the word just written */
    }
    data_block_add(&out, 0);
    ok = 1;
done:
    if (!ok) out.m_size = 0;
    data_block_resize(dic_block, out.m_size);
    if (out.m_size) memcpy(dic_block->m_data, out.m_data, out.m_size);
    data_block_destroy(&out);
}

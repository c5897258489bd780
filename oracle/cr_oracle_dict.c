/*
 * oracle/cr_oracle_dict.c — static-dictionary stage: word census, dictionary blob coding, per-block
 * word substitution and its inverse. TEST INFRASTRUCTURE ONLY (see cr_oracle.h).
 * Restated from the behaviour of /root/reference/src/cr-dicpick.c and cr-diccode.c; citations are to
 * those files. State that the reference keeps in file-scope statics lives in a cro_dict object.
 */
#include "cr_oracle.h"
#include <ctype.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define DIC_MAXWORDS   25000                              /* cr-diccode.h:39 TOTAL_WORD_NUM */
#define DIC_LEVEL1(n)  ((65535 - (int)(n)) / 255 - 1)     /* cr-diccode.h:40 */
#define WORD_MIN       2                                  /* cr-diccode.h:41 */
#define WORD_MAX       20                                 /* cr-diccode.h:42 */
#define PIECE_BYTES    1000000u                           /* cr-diccode.c:176-178 */

typedef struct trie_node {
    int id;                /* -1 internal, otherwise the word number */
    int next[128];
} trie_node;

struct cro_dict {
    char      word[DIC_MAXWORDS][WORD_MAX + 2];           /* cr-diccode.c:33 */
    uint8_t   wlen[DIC_MAXWORDS];
    int       nwords;                                     /* dic_len */
    trie_node* nodes; uint32_t nnodes, ncap;
    uint32_t  ntrie_words;
};

cro_dict* cro_dict_new(void) { return (cro_dict*)calloc(1, sizeof(cro_dict)); }
void cro_dict_free(cro_dict* d) { if (d) { free(d->nodes); free(d); } }
int cro_dict_words(const cro_dict* d) { return d->nwords; }
uint32_t cro_dict_trie_nodes(const cro_dict* d) { return d->nnodes; }
const char* cro_dict_word(const cro_dict* d, int i) { return d->word[i]; }

/* cr-diccode.c:47-70 */
static void trie_add(cro_dict* d, const char* w) {
    uint32_t at = 0;
    for (uint32_t i = 0; w[i]; ) {
        unsigned char ch = (unsigned char)w[i];
        if (d->nodes[at].next[ch] == 0) {
            if (d->nnodes >= d->ncap) {
                d->ncap = (uint32_t)(d->nnodes * 1.33 + 1);
                d->nodes = (trie_node*)realloc(d->nodes, d->ncap * sizeof(trie_node));
            }
            memset(&d->nodes[d->nnodes], 0, sizeof(trie_node));
            d->nodes[at].id = -1;                          /* a node with children is never terminal */
            d->nodes[at].next[ch] = (int)d->nnodes++;
        }
        at = (uint32_t)d->nodes[at].next[ch];
        i++;
    }
    d->nodes[at].id = (int)d->ntrie_words++;
}

/* cr-diccode.c:76-118: one word per line; words ending in a letter get " \0" appended */
int cro_dict_load(cro_dict* d, const char* text, int with_trie) {
    int p = 0;
    for (size_t i = 0; text[i]; i++) {
        if (text[i] == '\n') {
            if (p > 0 && isalpha((unsigned char)d->word[d->nwords][p - 1])) {
                d->word[d->nwords][p++] = ' ';
                d->word[d->nwords][p++] = 0;
            }
            p = 0;
            d->nwords++;
        } else {
            d->word[d->nwords][p++] = text[i];
        }
    }
    for (int i = 0; i < d->nwords; i++) d->wlen[i] = (uint8_t)strlen(d->word[i]);
    if (!with_trie) return 0;

    d->ncap = 4096; d->nnodes = 1; d->ntrie_words = 0;
    d->nodes = (trie_node*)calloc(d->ncap, sizeof(trie_node));
    for (int i = 0; i < d->nwords; i++) trie_add(d, d->word[i]);
    for (int c = 'A'; c < 'Z'; c++)                         /* cr-diccode.c:107-109: 'Z' is left out */
        d->nodes[0].next[c] = d->nodes[0].next[tolower(c)];
    for (uint32_t i = 0; i < d->nnodes; i++) {              /* cr-diccode.c:110-117 */
        int sp = d->nodes[i].next[' '];
        if (sp > 0) {
            if (!d->nodes[i].next['.']) d->nodes[i].next['.'] = sp;
            if (!d->nodes[i].next[',']) d->nodes[i].next[','] = sp;
            if (!d->nodes[i].next[':']) d->nodes[i].next[':'] = sp;
            if (!d->nodes[i].next[';']) d->nodes[i].next[';'] = sp;
        }
    }
    return (int)d->ntrie_words;
}

/* cr-diccode.c:309: a word that follows ". " or ".  " is expected to be capitalised */
static int sentence_start(const uint8_t* s, uint32_t i) {
    return i >= 3 && s[i - 1] == ' ' && (s[i - 2] == '.' || (s[i - 2] == ' ' && s[i - 3] == '.'));
}

static void put_literal(const cro_dict* d, const uint8_t escmap[256], uint8_t c, cro_buf* out) {
    int l1 = DIC_LEVEL1(d->nwords);
    if (escmap[c]) {                                        /* cr-diccode.c:337-341: code of word #nwords */
        cro_buf_put(out, (uint8_t)(d->nwords / (256 - l1)));
        cro_buf_put(out, (uint8_t)(d->nwords % (256 - l1) + l1));
    }
    cro_buf_put(out, c);
}

/* cr-diccode.c:285-362 */
static void encode_piece(const cro_dict* d, const uint8_t* s, uint32_t n, const uint8_t esc[10], cro_buf* out) {
    uint8_t escmap[256] = {0};
    for (int i = 0; i < 10; i++) escmap[esc[i]] = (uint8_t)(i + 1);
    const int l1 = DIC_LEVEL1(d->nwords);
    uint32_t i = 0;
    for (; i + WORD_MAX * 2 < n; i++) {
        uint32_t j = i;
        const trie_node* at = d->nodes;
        if (i > 0 && isalpha(s[i]) && !isalpha(s[i - 1])) {
            while (s[j] < 128 && (at = d->nodes + at->next[s[j]]) != d->nodes && at->id == -1) j++;
        }
        if (s[j] < 128 && at != d->nodes) {
            int flip = (isupper(s[i]) != 0) ^ sentence_start(s, i);
            int tail = s[j] == ':' ? 4 : s[j] == ';' ? 3 : s[j] == ',' ? 2 : s[j] == '.' ? 1 : 0;
            if (at->id < l1) {
                cro_buf_put(out, (uint8_t)at->id);
            } else {
                cro_buf_put(out, (uint8_t)(at->id / (256 - l1)));
                cro_buf_put(out, (uint8_t)(at->id % (256 - l1) + l1));
            }
            cro_buf_put(out, esc[flip * 5 + tail]);
            i = j;
        } else {
            put_literal(d, escmap, s[i], out);
        }
    }
    for (; i < n; i++) put_literal(d, escmap, s[i], out);   /* cr-diccode.c:347-356 */
    uint8_t sz[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    cro_buf_append(out, sz, 4);
}

/* cr-diccode.c:142-221 */
uint32_t cro_dict_encode(const cro_dict* d, const uint8_t* in, uint32_t n, uint8_t* out) {
    uint32_t hist[256] = {0};
    uint8_t esc[10] = {0};
    for (uint32_t i = 0; i < n; i++) hist[in[i]]++;
    for (int k = 0; k < 10; k++) {                          /* ten least frequent values, in order */
        for (int v = 0; v < 256; v++) if (hist[v] < hist[esc[k]]) esc[k] = (uint8_t)v;
        hist[esc[k]] = 0xFFFFFFFFu;
    }
    cro_buf b, p1, p2; cro_buf_init(&b); cro_buf_init(&p1); cro_buf_init(&p2);
    uint32_t pos = 0;
    while (pos < n) {
        uint32_t a = pos + PIECE_BYTES < n ? PIECE_BYTES : n - pos; pos += a;
        uint32_t c = pos + PIECE_BYTES < n ? PIECE_BYTES : n - pos; pos += c;
        cro_buf_clear(&p1); cro_buf_clear(&p2);
        encode_piece(d, in + pos - c - a, a, esc, &p1);
        encode_piece(d, in + pos - c, c, esc, &p2);
        uint8_t hdr[8];
        for (int k = 0; k < 4; k++) { hdr[k] = (uint8_t)(p1.size >> (8 * k)); hdr[4 + k] = (uint8_t)(p2.size >> (8 * k)); }
        cro_buf_append(&b, hdr, 8);
        cro_buf_append(&b, p1.data, p1.size);
        cro_buf_append(&b, p2.data, p2.size);
    }
    cro_buf_append(&b, esc, 10);
    cro_buf_put(&b, 1);
    uint32_t total;
    if (b.size >= n) {                                      /* cr-diccode.c:212-217 */
        memcpy(out, in, n);
        out[n] = 0;
        total = n + 1;
    } else {
        memcpy(out, b.data, b.size);
        total = b.size;
    }
    cro_buf_free(&b); cro_buf_free(&p1); cro_buf_free(&p2);
    return total;
}

/* cr-diccode.c:364-425: pieces are parsed from their end */
static uint32_t decode_piece(const cro_dict* d, const uint8_t* s, uint32_t n, const uint8_t esc[10], uint8_t* out, uint32_t cap) {
    uint8_t escmap[256] = {0};
    for (int i = 0; i < 10; i++) escmap[esc[i]] = (uint8_t)(i + 1);
    const int l1 = DIC_LEVEL1(d->nwords);
    if (n < 4) return 0xFFFFFFFFu;
    uint32_t total = (uint32_t)s[n - 4] | ((uint32_t)s[n - 3] << 8) | ((uint32_t)s[n - 2] << 16) | ((uint32_t)s[n - 1] << 24);
    if (total > cap) return 0xFFFFFFFFu;
    uint32_t w = total, r = n - 4, fix = 0xFFFFFFFFu;
    while (w > 0) {
        if (r == 0) return 0xFFFFFFFFu;
        uint8_t ch = s[--r];
        if (!escmap[ch]) { out[--w] = ch; continue; }
        if (r == 0) return 0xFFFFFFFFu;
        int id = s[--r];
        if (id >= l1) {
            if (r == 0) return 0xFFFFFFFFu;
            id = s[--r] * (256 - l1) + (id - l1);
            if (id == d->nwords) { out[--w] = ch; continue; }
        }
        if (id >= d->nwords || d->wlen[id] > w) return 0xFFFFFFFFu;
        uint32_t len = d->wlen[id];
        w -= len;
        memcpy(out + w, d->word[id], len);
        switch (escmap[ch]) {                               /* cr-diccode.c:405-410 */
            case 2: case 7:  out[w + len - 1] = '.'; break;
            case 3: case 8:  out[w + len - 1] = ','; break;
            case 4: case 9:  out[w + len - 1] = ';'; break;
            case 5: case 10: out[w + len - 1] = ':'; break;
        }
        if (escmap[ch] >= 6) out[w] ^= 0x20;
        /* the word decoded just before (further right) can be case-checked now that the bytes in
         * front of it exist (cr-diccode.c:415-419) */
        if (fix != 0xFFFFFFFFu && sentence_start(out, fix)) out[fix] ^= 0x20;
        fix = w;
    }
    if (fix != 0xFFFFFFFFu && sentence_start(out, fix)) out[fix] ^= 0x20;
    return total;
}

/* cr-diccode.c:223-283 */
uint32_t cro_dict_decode(const cro_dict* d, const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap) {
    if (n == 0) return 0xFFFFFFFFu;
    if (in[n - 1] == 0) {
        if (n - 1 > cap) return 0xFFFFFFFFu;
        memcpy(out, in, n - 1);
        return n - 1;
    }
    if (n < 11) return 0xFFFFFFFFu;
    const uint8_t* esc = in + n - 11;
    uint32_t pos = 0, w = 0;
    while (pos + 11 < n) {
        if (pos + 8 > n) return 0xFFFFFFFFu;
        uint32_t a = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
        uint32_t c = (uint32_t)in[pos + 4] | ((uint32_t)in[pos + 5] << 8) | ((uint32_t)in[pos + 6] << 16) | ((uint32_t)in[pos + 7] << 24);
        pos += 8;
        if ((uint64_t)pos + a + c + 11 > n) return 0xFFFFFFFFu;
        uint32_t g = decode_piece(d, in + pos, a, esc, out + w, cap - w);
        if (g == 0xFFFFFFFFu) return g;
        w += g;
        g = decode_piece(d, in + pos + a, c, esc, out + w, cap - w);
        if (g == 0xFFFFFFFFu) return g;
        w += g;
        pos += a + c;
    }
    return w;
}

/* ------------------------------------------------------------------ census (cr-dicpick.c) */

#define MAP_LIMIT  (DIC_MAXWORDS * 13 + 1)                  /* cr-dicpick.c:33 */
#define MAP_SLOTS  (DIC_MAXWORDS * 23 + 3)                  /* cr-dicpick.c:34 */
#define MIN_COUNT  5                                        /* cr-dicpick.c:35 */
#define CENSUS_CHUNK 200000                                 /* cr-dicpick.c:162 */

typedef struct census_cell { char w[WORD_MAX + 1]; int count; } census_cell;

/* cr-dicpick.c:71-78 (two's-complement wrap, sign bit dropped) */
static int word_hash(const char* s) {
    uint32_t h = 0;
    while (isalpha((unsigned char)*s)) h = h * 131313131u + (uint32_t)tolower((unsigned char)*s++);
    return (int)(h & 0x7fffffffu);
}
/* cr-dicpick.c:79-88 */
static int word_differs(const char* a, const char* b) {
    while (isalpha((unsigned char)*a) && isalpha((unsigned char)*b)) {
        if (tolower((unsigned char)*a) != tolower((unsigned char)*b)) return 1;
        a++; b++;
    }
    return (!isalpha((unsigned char)*a)) != (!isalpha((unsigned char)*b));
}
/* cr-dicpick.c:89-95 */
static void word_copy(char* dst, const char* src) {
    while (isalpha((unsigned char)*src)) *dst++ = (char)tolower((unsigned char)*src++);
    *dst = 0;
}
static int by_count_desc(const void* pa, const void* pb) {   /* cr-dicpick.c:58-65 */
    const census_cell* a = (const census_cell*)pa; const census_cell* b = (const census_cell*)pb;
    if (a->count != b->count) return b->count - a->count;
    return strcmp(b->w, a->w);
}
static int by_word(const void* pa, const void* pb) {         /* cr-dicpick.c:53-57 */
    return strcmp(((const census_cell*)pa)->w, ((const census_cell*)pb)->w);
}

static uint32_t map_find(const census_cell* map, const char* w) {
    uint32_t at = (uint32_t)word_hash(w) % MAP_SLOTS;
    while (map[at].count > 0 && word_differs(map[at].w, w)) at = (at + 1) % MAP_SLOTS;
    return at;
}

/* cr-dicpick.c:96-143: count, and when the map fills up drop everything within 5 of the minimum */
static void census_add(census_cell* map, int* used, const char* w) {
    uint32_t at = map_find(map, w);
    if (map[at].count > 0) { map[at].count++; return; }
    word_copy(map[at].w, w);
    map[at].count = 1;
    if (++*used != MAP_LIMIT) return;
    census_cell* keep = (census_cell*)malloc(sizeof(census_cell) * MAP_LIMIT);
    int lo = INT_MAX, k = *used;
    for (uint32_t i = 0; i < MAP_SLOTS; i++) {
        if (map[i].count > 0) {
            if (map[i].count < lo) lo = map[i].count;
            keep[--k] = map[i];
        }
        map[i].count = 0;
    }
    *used = 0;
    for (int i = 0; i < MAP_LIMIT; i++) {
        if (keep[i].count > lo + 5) {
            uint32_t t = map_find(map, keep[i].w);
            strcpy(map[t].w, keep[i].w);
            map[t].count = keep[i].count;
            ++*used;
        }
    }
    free(keep);
}

/* cr-dicpick.c:164-273. `data` is the whole file; returns the dictionary text (NUL-terminated,
 * one word per line) in `out` (capacity >= 26000*23) and its size including the NUL. */
uint32_t cro_dicpick(const uint8_t* data, uint64_t n, uint8_t* out) {
    census_cell* map = (census_cell*)calloc(MAP_SLOTS, sizeof(census_cell));
    uint8_t* chunk = (uint8_t*)malloc(CENSUS_CHUNK);
    int used = 0;
    uint8_t ok_after[256] = {0};
    ok_after[' '] = ok_after[','] = ok_after['.'] = ok_after[':'] = ok_after[';'] = 1;
    for (uint64_t base = 0; base < n; base += CENSUS_CHUNK) {
        int len = (int)(n - base < CENSUS_CHUNK ? n - base : CENSUS_CHUNK);
        memcpy(chunk, data + base, (size_t)len);
        chunk[len - 1] = 0;                                  /* cr-dicpick.c:192: last byte is sacrificed */
        for (int x = 1; x < len; x++) {
            if (isalpha(chunk[x]) && !isalpha(chunk[x - 1])) {
                int y = x + 1;
                while (y < len && islower(chunk[y])) y++;
                if (y >= x + WORD_MIN && y <= x + WORD_MAX && ok_after[chunk[y]]) {
                    char w[WORD_MAX + 2];
                    word_copy(w, (const char*)chunk + x);
                    census_add(map, &used, w);
                }
                x = y;
            }
        }
    }
    /* keep count > 5, most frequent first (cr-dicpick.c:219-228) */
    int y = 0;
    for (uint32_t x = 0; x < MAP_SLOTS; x++) {
        if (map[x].count > MIN_COUNT) {
            census_cell c = map[x];
            word_copy(map[y].w, c.w);
            map[y].count = c.count;
            y++;
        }
    }
    qsort(map, (size_t)y, sizeof(census_cell), by_count_desc);
    const int reserved = 2;
    if (y > DIC_MAXWORDS - reserved) y = DIC_MAXWORDS - reserved;
    if (y > DIC_LEVEL1(y) - reserved) {                      /* two-byte-code words sorted by name */
        int x = DIC_LEVEL1(y) - reserved;
        qsort(map + x, (size_t)(y - x), sizeof(census_cell), by_word);
    }
    uint32_t o = 0;
    const char* fixed[2] = {"\x20\x20", "http://www."};      /* cr-dicpick.c:38-41 */
    for (int r = 0; r < reserved; r++) { size_t l = strlen(fixed[r]); memcpy(out + o, fixed[r], l); o += (uint32_t)l; out[o++] = '\n'; }
    for (int x = 0; x < y; x++) {
        if (x < DIC_LEVEL1(y) || strlen(map[x].w) >= WORD_MIN + 1) {   /* cr-dicpick.c:254: two-letter words only pay off with a 1-byte code */
            size_t l = strlen(map[x].w);
            memcpy(out + o, map[x].w, l); o += (uint32_t)l; out[o++] = '\n';
        }
    }
    out[o++] = 0;
    free(chunk); free(map);
    return o;
}

/* cr-dicpick.c:275-316: front coding — each word after the first is (shared-prefix length, rest) */
uint32_t cro_dic_lcp_encode(const uint8_t* text, uint8_t* out) {
    uint32_t prev = 0, cur = 0, o = 0;
    while (text[cur] != '\n') out[o++] = text[cur++];
    cur++; out[o++] = '\n';
    while (text[cur] != 0) {
        uint32_t lcp = 0;
        while (text[prev + lcp] == text[cur + lcp]) lcp++;
        out[o++] = (uint8_t)lcp;
        prev = cur;
        cur += lcp;
        while (text[cur] != '\n') out[o++] = text[cur++];
        cur++; out[o++] = '\n';
    }
    out[o++] = 255;
    return o;
}

/* cr-dicpick.c:318-346 */
uint32_t cro_dic_lcp_decode(const uint8_t* blob, uint8_t* out) {
    uint32_t r = 0, o = 0, prev = 0;
    while (blob[r] != '\n') out[o++] = blob[r++];
    r++; out[o++] = '\n';
    while (blob[r] != 255) {
        uint32_t lcp = blob[r++];
        for (; lcp; lcp--) out[o++] = out[prev++];
        while (blob[r] != '\n') out[o++] = blob[r++];
        r++; out[o++] = '\n';
        while (out[prev] != '\n') prev++;
        prev++;
    }
    out[o++] = 0;
    return o;
}

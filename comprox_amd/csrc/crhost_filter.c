/*
 * comprox_amd/csrc/crhost_filter.c — the `-F` pre-filters (host C, one pass per datablock before the
 * dictionary stage; stateful across the blocks of a file, so they stay on the CPU — SURVEY.md §8f #3).
 *
 * Restates the behaviour of the reference's
 *   filter_inplace                 /root/reference/src/cr-filter.c:33-73   (scan, remember the filter that matched last)
 *   i386_e8e9                      src/filter_x86opcode.h:38-62           (CALL/JMP rel32 <-> absolute inside an image)
 *   pe_i386_transform              src/filter_x86_pe.c:126-159 (+ header analysis :71-124)
 *   elf_i386_transform             src/filter_x86_elf.c:127-156 (+ :102-125)
 *   bmp_transform                  src/filter_bmp.c:151-204 (+ delta passes :57-149)
 * as three small state machines behind one dispatcher. Everything that decides bytes is kept, including
 * the reference's quirks (each marked "quirk" below) — also the ELF byte counter the reference never resets,
 * which makes its transform lossy from the second ELF image of a run on: that is the DEFAULT, because the bar
 * is the reference encoder's bytes. crgpu_filter_set_mode(CRGPU_FILTER_RESTART_ELF) selects a transform that
 * restarts the counter per image and can be undone; it is not the reference's format (see elf_step). Where the reference reads or writes PAST the block
 * it was given (header fields beyond the block, an ELF code range that is 52 bytes longer than the block,
 * a code range shorter than 8 bytes) its result depends on stale heap bytes or it crashes; there this file
 * stays inside the block, and says so. The process-lifetime state of the reference (function statics) is
 * one struct here; crgpu_filter_reset() clears it (a new entry point: a library, unlike a one-shot
 * command line, filters more than one stream per process).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/crgpu.h"

static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* ---- CALL / JMP operands, filter_x86opcode.h:38-62 ------------------------------------------------
 * Every byte 0xE8 / 0xE9 below limit - 8 is followed by a 32-bit operand; encoding turns a relative
 * target that lands inside [0, image_end) into an absolute one, decoding turns it back. `cur` is the
 * offset of buf[0] inside the image. */
static void e8e9(uint8_t* buf, uint32_t limit, uint32_t avail, int decode, int32_t cur, int32_t end) {
    if (limit < 8u) return;                       /* the reference's unsigned `limit - 8` wraps and runs off the block */
    uint32_t i = 0;
    while (i < limit - 8u && i < avail) {         /* avail < limit only for the ELF range that overhangs its block */
        if ((buf[i++] & 0xfe) != 0xe8) continue;
        if (i + 4u > avail) break;                /* the operand is not (wholly) inside the block */
        const int32_t at = cur + (int32_t)i;      /* image offset of the operand */
        int32_t v = (int32_t)rd32(buf + i);
        if (!decode) {
            if (v >= -at && v < end - at) v = (int32_t)((uint32_t)v + (uint32_t)at);
            else if (v > 0 && v < end) v = (int32_t)((uint32_t)v - (uint32_t)end);
        } else {
            if (v < 0) { if ((int32_t)((uint32_t)v + (uint32_t)at) >= 0) v = (int32_t)((uint32_t)v + (uint32_t)end); }
            else if (v < end) v = (int32_t)((uint32_t)v - (uint32_t)at);
        }
        wr32(buf + i, (uint32_t)v);
        i += 4;
    }
}

/* state of an image that continues into the next call (and the next datablock) */
typedef struct { int open; uint32_t cur, size; } image_state;

typedef struct {
    int last;                     /* filter_inplace's `lastproc`: 0 none, 1 PE, 2 ELF, 3 BMP */
    image_state pe, elf;
    struct { int open; int cur, size, row, bpp, width, height, skip; } bmp;
} filter_state;

static filter_state g_fs;
static int g_mode = CRGPU_FILTER_REFERENCE;
static int g_stale_elf;           /* ELF images converted with the reference's stale byte counter since the last reset */

void crgpu_filter_reset(void) { memset(&g_fs, 0, sizeof g_fs); g_stale_elf = 0; }

int crgpu_filter_lossy(void) { return g_stale_elf; }

int crgpu_filter_set_mode(int mode) {
    if (mode != CRGPU_FILTER_REFERENCE && mode != CRGPU_FILTER_RESTART_ELF) return CRGPU_E_ARG;
    g_mode = mode;
    return CRGPU_OK;
}

int crgpu_filter_mode(void) { return g_mode; }

/* ---- PE / COFF i386, filter_x86_pe.c ---------------------------------------------------------------- */
static uint32_t pe_step(uint8_t* buf, uint32_t len, int decode) {
    image_state* s = &g_fs.pe;
    uint8_t* start = buf;
    uint32_t size = umin(s->size - s->cur, len), ret = size;
    if (!s->open) {
        s->cur = 0;
        /* "MZ", e_lfanew at 0x3c, "PE\0\0" there (:105-124). The reference only checks e_lfanew < len before
         * reading the signature and the headers; here every field read must lie inside the block. */
        if (len < 0x40u || rd16(buf) != 0x5a4du) return 0;
        const uint32_t hdr = rd32(buf + 0x3c);
        if (hdr == 0 || hdr >= len || (uint64_t)hdr + 24u > len || rd32(buf + hdr) != 0x00004550u) return 0;
        const uint16_t machine = rd16(buf + hdr + 4), nsec = rd16(buf + hdr + 6), optsz = rd16(buf + hdr + 20), flags = rd16(buf + hdr + 22);
        if (machine != 0x14c && (flags & 0x0002)) return 0;      /* quirk (:81): only a non-i386 EXECUTABLE image is turned away */
        const uint32_t sec_tbl = 24u + optsz;                      /* both relative to the PE signature */
        const uint32_t hdr_size = sec_tbl + (uint32_t)nsec * 40u;
        if ((uint64_t)hdr + hdr_size > len) return 0;             /* section table inside the block */
        uint32_t est = hdr_size;
        for (uint32_t k = 0; k < nsec; k++) est += rd32(buf + hdr + sec_tbl + k * 40u + 16u);   /* SizeOfRawData */
        if (est > (1u << 28)) return 0;
        /* quirk (:147-150): the code range is taken to start hdr_size bytes into the FILE — the e_lfanew
         * offset is not added — and to be est - hdr_size long */
        start = buf + hdr_size;
        s->size = est - hdr_size;
        size = umin(s->size, len - hdr_size);
        ret = size + hdr_size;
    }
    e8e9(start, size, size, decode, (int32_t)s->cur, (int32_t)s->size);
    s->cur += size;
    s->open = s->cur < s->size;
    return ret;
}

/* ---- ELF32 i386, filter_x86_elf.c ------------------------------------------------------------------- */
static uint32_t elf_step(uint8_t* buf, uint32_t len, int decode) {
    image_state* s = &g_fs.elf;
    uint8_t* start = buf;
    uint32_t size = umin(s->size - s->cur, len), ret = size;
    if (!s->open) {
        if (len < 52u || rd32(buf) != 0x464c457fu || rd16(buf + 18) != 3) return 0;      /* "\x7fELF", EM_386 */
        const uint32_t shoff = rd32(buf + 32);
        if (shoff < 52u || shoff - 52u >= (1u << 30)) return 0;
        /* quirk (:143-146): the range starts behind the 52-byte header, its length is e_shoff - 52 - 52, and
         * the length is capped by len, not by len - 52. The reference's pass therefore runs up to 52 bytes
         * past the block when the image does not end inside it; here it converts every operand that lies
         * inside the block and stops there (the returned length, which positions the scan, is the reference's). */
        s->size = shoff - 52u - 52u;
        /* quirk (:131-134, function-scope statics): the reference never resets its ELF byte counter (`curr`), so a
         * second ELF image in one run is converted with the first image's length as its start offset — and is
         * taken to be complete as soon as that stale counter passes its length. With offset > length the two
         * operand ranges of i386_e8e9 overlap and the transform cannot be undone (tests/golden/golden_filter.json,
         * cases two_elf / tar_like: the reference's own FILTER_DEC does not restore them). Kept by default: the
         * encoder's bytes are the reference's. CRGPU_FILTER_RESTART_ELF restarts the counter with every image —
         * a transform that round-trips, marked m_filt = 2 in the files of comp*-gpu -FF. */
        if (g_mode == CRGPU_FILTER_RESTART_ELF) s->cur = 0;
        else if (s->cur != 0) g_stale_elf++;     /* the caller can see it (crgpu_filter_lossy): comp*-gpu -F warns */
        start = buf + 52;
        size = umin(s->size, len);
        ret = size;
        e8e9(start, size, len - 52u, decode, (int32_t)s->cur, (int32_t)s->size);
    } else {
        e8e9(start, size, size, decode, (int32_t)s->cur, (int32_t)s->size);
    }
    s->cur += size;
    s->open = s->cur < s->size;
    return ret;
}

/* ---- BMP 24 / 32 bpp, filter_bmp.c ------------------------------------------------------------------- */
/* colour (B -= G, R -= G), then left-neighbour and upper-neighbour deltas over the complete rows of the
 * range; decoding undoes them in the opposite order (:57-149) */
static uint32_t bmp_rows(uint8_t* buf, uint32_t len, int width, int row, int bpp, int decode) {
    const int rows = (int)(len / (uint32_t)row), px = bpp / 8;
    if (!decode) {
        for (int y = 0; y < rows; y++) {
            uint8_t* r = buf + (size_t)y * row;
            for (int x = 0; x < width; x++) { r[x * px + 0] -= r[x * px + 1]; r[x * px + 2] -= r[x * px + 1]; }
        }
        for (int y = 0; y < rows; y++) {
            uint8_t* r = buf + (size_t)y * row;
            for (int x = width - 1; x > 0; x--) for (int c = 0; c < px; c++) r[x * px + c] -= r[(x - 1) * px + c];
        }
        for (int y = rows - 1; y > 0; y--) {
            uint8_t *r = buf + (size_t)y * row, *u = r - row;
            for (int x = 0; x < width * px; x++) r[x] -= u[x];
        }
    } else {
        for (int y = 0; y < rows; y++) {
            uint8_t* r = buf + (size_t)y * row;
            for (int x = 1; x < width; x++) for (int c = 0; c < px; c++) r[x * px + c] += r[(x - 1) * px + c];
        }
        for (int y = 1; y < rows; y++) {
            uint8_t *r = buf + (size_t)y * row, *u = r - row;
            for (int x = 0; x < width * px; x++) r[x] += u[x];
        }
        for (int y = 0; y < rows; y++) {
            uint8_t* r = buf + (size_t)y * row;
            for (int x = 0; x < width; x++) { r[x * px + 0] += r[x * px + 1]; r[x * px + 2] += r[x * px + 1]; }
        }
    }
    return (uint32_t)row * (uint32_t)rows;
}

static uint32_t bmp_step(uint8_t* buf, uint32_t len, int decode) {
    if (!g_fs.bmp.open) {
        if (len < 54u || rd16(buf) != 0x4d42u || rd16(buf + 26) != 1 || rd32(buf + 30) != 0) return 0;   /* "BM", 1 plane, BI_RGB */
        const uint32_t file_size = rd32(buf + 2), image_off = rd32(buf + 10), image_size = rd32(buf + 34);
        const uint16_t bpp = rd16(buf + 28);
        if (image_size != 0 && image_off + image_size != file_size) return 0;
        if (bpp != 24 && bpp != 32) return 0;
        const int width = abs((int)rd32(buf + 18)), height = abs((int)rd32(buf + 22));
        if (width < 4 || height < 4 || width >= (1 << 20) || height >= (1 << 20)) return 0;
        g_fs.bmp.width = width; g_fs.bmp.height = height; g_fs.bmp.bpp = bpp;
        g_fs.bmp.row = (bpp * width + 31) / 32 * 4;
        /* quirk (:183-188): the header is skipped in one step and the byte counter starts at the pixel
         * offset, not at 0, so the last image_off bytes of the pixel array are left alone */
        g_fs.bmp.cur = (int)image_off;
        g_fs.bmp.size = height * g_fs.bmp.row;
        g_fs.bmp.skip = 0;
        g_fs.bmp.open = 1;
        return image_off;
    }
    if (g_fs.bmp.skip > 0) {                                   /* the row that straddled the previous range (:190-195) */
        const uint32_t n = umin((uint32_t)g_fs.bmp.skip, len);
        g_fs.bmp.cur += (int)n;
        g_fs.bmp.skip -= (int)n;
        return n;
    }
    const uint32_t done = bmp_rows(buf, umin(len, (uint32_t)(g_fs.bmp.size - g_fs.bmp.cur)), g_fs.bmp.width, g_fs.bmp.row, g_fs.bmp.bpp, decode);
    g_fs.bmp.cur += (int)done;
    if (g_fs.bmp.cur < g_fs.bmp.size) g_fs.bmp.skip = (int)umin((uint32_t)g_fs.bmp.row, (uint32_t)(g_fs.bmp.size - g_fs.bmp.cur));
    else g_fs.bmp.open = 0;
    return done;
}

/* ---- dispatcher, cr-filter.c:33-73 -------------------------------------------------------------------- */
static uint32_t run(int which, uint8_t* buf, uint32_t len, int decode) {
    return which == 1 ? pe_step(buf, len, decode) : which == 2 ? elf_step(buf, len, decode) : bmp_step(buf, len, decode);
}

/* Try the filter that matched last, then PE, ELF, BMP, at every position a filter has not claimed.
 * Returns 1 when any filter touched the block (the block header's m_filt, src/main.c:183-185,200). */
int filter_inplace(unsigned char* buf, uint32_t len, int en_de) {
    int touched = 0;
    const int decode = en_de != 0;
    for (uint32_t pos = 0; pos < len; pos++) {
        if (g_fs.last) {
            const uint32_t n = run(g_fs.last, buf + pos, len - pos, decode);
            if (n == 0) g_fs.last = 0;
            else { touched = 1; pos += n - 1u; continue; }
        }
        for (int k = 1; k <= 3; k++) {
            const uint32_t n = run(k, buf + pos, len - pos, decode);
            if (n > 0) { touched = 1; g_fs.last = k; pos += n - 1u; break; }
        }
    }
    return touched;
}

// imread.cpp — the file front-end of the reference's entry points: `imgcodecs::imread(path,
// IMREAD_UNCHANGED)` (utils.rs:110-117, 132) and keypoint_match / ecc_match in the reference's own call shape, a list of
// paths (lib.rs:129-137, 702-710). Decoded here: binary PNM (P5 grey, P6 colour; 8 or 16 bit) without any library; PNG
// (8- and 16-bit grey / RGB, palette -> BGR, 1/2/4-bit grey -> 8 bit: libpng's row API, samples as stored, never
// gamma-converted), JPEG (grey / YCbCr: libjpeg-turbo at its default settings, OpenCV's decoder family) and stripped 8/16-bit
// grey / RGB TIFF (libtiff's handle API) through libpng16.so.16 / libjpeg.so.8 / libtiff.so.5 loaded at run time — the image
// has the libraries but not their headers, so the entry points used are declared below. Round 4: PNG with alpha or tRNS and
// RGBA TIFF decode to four channels as the reference's imread does, tiled TIFF, uncompressed BMP, still WebP (libwebp.so.7).
// CMYK / YCCK JPEG comes out as B G R through OpenCV's own conversion. What the build does not take (separate-plane TIFF;
// animated WebP; EXR / JPEG 2000) returns
// STK_NOT_IMPLEMENTED and the caller decodes it itself (the frame-based entry points are the boundary). A file that is
// missing or not an image behaves as in the reference: imread gives an empty Mat and the following cvtColor raises ->
// STK_BACKEND_ERROR (OpenCvError).
#include <dlfcn.h>
#include <setjmp.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "context.h"

namespace {

struct Pnm { int w = 0, h = 0, cn = 0, depth = 0; size_t data_ofs = 0; };

// parse "P5|P6 <w> <h> <maxval>\n" with '#' comments; returns 0 ok, 1 not a PNM, 2 unsupported variant
int pnm_header(const unsigned char* b, size_t n, Pnm& p) {
    if (n < 3 || b[0] != 'P' || (b[1] != '5' && b[1] != '6')) return 1;
    p.cn = b[1] == '6' ? 3 : 1;
    size_t i = 2;
    long v[3];
    for (int k = 0; k < 3; k++) {
        for (;;) {                                            // whitespace and comments
            while (i < n && (b[i] == ' ' || b[i] == '\t' || b[i] == '\n' || b[i] == '\r')) i++;
            if (i < n && b[i] == '#') { while (i < n && b[i] != '\n') i++; continue; }
            break;
        }
        if (i >= n || b[i] < '0' || b[i] > '9') return 1;
        long x = 0;
        while (i < n && b[i] >= '0' && b[i] <= '9') { x = x * 10 + (b[i] - '0'); if (x > 1000000) return 1; i++; }
        v[k] = x;
    }
    if (i >= n) return 1;
    i++;                                                      // the single whitespace byte before the raster
    if (v[0] <= 0 || v[1] <= 0) return 1;
    if (v[2] != 255 && v[2] != 65535) return 2;               // other maxvals: OpenCV's handling is not restated here
    p.w = (int)v[0]; p.h = (int)v[1]; p.depth = v[2] == 255 ? 8 : 16; p.data_ofs = i;
    return 0;
}

bool read_file(const char* path, std::vector<unsigned char>& out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 0) { std::fclose(f); return false; }
    out.resize((size_t)sz);
    const size_t got = sz ? std::fread(out.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    return got == (size_t)sz;
}

bool has_ext(const char* path, const char* ext) {
    const size_t lp = std::strlen(path), le = std::strlen(ext);
    if (lp < le) return false;
    for (size_t i = 0; i < le; i++) {
        char c = path[lp - le + i];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        if (c != ext[i]) return false;
    }
    return true;
}

// R G B -> B G R of `px` 8-bit pixels. The byte loop below runs at ~1 GB/s per thread, which made the decoder pool the
// slowest stage of the path-based entry points on uncompressed files (round 4: 256 x 4K from tmpfs, 16 threads, 8-10 GB/s
// against the 55 GB/s of the PCIe link behind it); with SSSE3 five pixels go through one byte shuffle per 16-byte load
// (the 16th byte is written too and overwritten by the next store; the last 16 bytes take the byte loop).
#if defined(__x86_64__)
__attribute__((target("ssse3"))) static size_t swap_rb_ssse3(const unsigned char* src, unsigned char* dst, size_t px) {
    typedef char v16 __attribute__((vector_size(16)));
    typedef char v16u __attribute__((vector_size(16), aligned(1)));
    const v16 mask = {2, 1, 0, 5, 4, 3, 8, 7, 6, 11, 10, 9, 14, 13, 12, 15};
    size_t i = 0;
    for (; (i + 5) * 3 + 1 <= px * 3 && (i + 5) * 3 + 16 <= px * 3; i += 5)
        *(v16u*)(dst + 3 * i) = __builtin_ia32_pshufb128(*(const v16u*)(src + 3 * i), mask);
    return i;
}
#endif
void swap_rb_u8(const unsigned char* src, unsigned char* dst, size_t px) {
    size_t i = 0;
#if defined(__x86_64__)
    if (px >= 16 && __builtin_cpu_supports("ssse3")) i = swap_rb_ssse3(src, dst, px);
#endif
    for (; i < px; i++) { dst[3 * i] = src[3 * i + 2]; dst[3 * i + 1] = src[3 * i + 1]; dst[3 * i + 2] = src[3 * i]; }
}

// decode into `dst` (w*h*cn samples of depth/8 bytes): RGB -> BGR, 16-bit big-endian -> native
void pnm_decode(const unsigned char* raster, const Pnm& p, void* dst) {
    const size_t px = (size_t)p.w * p.h;
    if (p.depth == 8) {
        unsigned char* o = (unsigned char*)dst;
        if (p.cn == 1) std::memcpy(o, raster, px);
        else swap_rb_u8(raster, o, px);
    } else {
        unsigned short* o = (unsigned short*)dst;
        auto be = [&](size_t s) { return (unsigned short)((raster[2 * s] << 8) | raster[2 * s + 1]); };
        if (p.cn == 1) for (size_t i = 0; i < px; i++) o[i] = be(i);
        else for (size_t i = 0; i < px; i++) { o[3 * i] = be(3 * i + 2); o[3 * i + 1] = be(3 * i + 1); o[3 * i + 2] = be(3 * i); }
    }
}

// ---- PNG through libpng 1.6's row API, resolved at run time (opaque handles only: no struct layouts involved) ----------
// What OpenCV's PngDecoder does for IMREAD_UNCHANGED (grfmt_png.cpp) [OCV-RECALL]: samples as stored — 8 bit, or 16 bit
// byte-swapped to native order, never gamma-converted (which is why libpng's simplified API, whose 16-bit output is "linear
// light", is not used) —, palette expanded to RGB, 1/2/4-bit grey expanded to 8, colour delivered as BGR.
struct PngApi {
    const char* (*get_libpng_ver)(const void*) = nullptr;
    void* (*create_read_struct)(const char*, void*, void (*)(void*, const char*), void (*)(void*, const char*)) = nullptr;
    void* (*create_info_struct)(void*) = nullptr;
    void (*destroy_read_struct)(void**, void**, void**) = nullptr;
    jmp_buf* (*set_longjmp_fn)(void*, void (*)(jmp_buf, int), size_t) = nullptr;
    void (*longjmp_)(void*, int) = nullptr;
    void (*init_io)(void*, FILE*) = nullptr;
    void (*read_info)(void*, void*) = nullptr;
    uint32_t (*get_IHDR)(void*, void*, uint32_t*, uint32_t*, int*, int*, int*, int*, int*) = nullptr;
    uint32_t (*get_valid)(void*, void*, uint32_t) = nullptr;
    void (*set_palette_to_rgb)(void*) = nullptr;
    void (*set_expand_gray_1_2_4_to_8)(void*) = nullptr;
    void (*set_swap)(void*) = nullptr;
    void (*set_bgr)(void*) = nullptr;
    void (*set_gray_to_rgb)(void*) = nullptr;
    void (*set_tRNS_to_alpha)(void*) = nullptr;
    int (*set_interlace_handling)(void*) = nullptr;
    void (*read_update_info)(void*, void*) = nullptr;
    size_t (*get_rowbytes)(void*, void*) = nullptr;
    void (*read_row)(void*, unsigned char*, unsigned char*) = nullptr;
    bool ok = false;
};

const PngApi& png_api() {
    static const PngApi api = []() {
        PngApi a;
        void* h = dlopen("libpng16.so.16", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
#define STK_PNG_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
        STK_PNG_SYM(get_libpng_ver, "png_get_libpng_ver"); STK_PNG_SYM(create_read_struct, "png_create_read_struct");
        STK_PNG_SYM(create_info_struct, "png_create_info_struct"); STK_PNG_SYM(destroy_read_struct, "png_destroy_read_struct");
        STK_PNG_SYM(set_longjmp_fn, "png_set_longjmp_fn"); STK_PNG_SYM(longjmp_, "png_longjmp"); STK_PNG_SYM(init_io, "png_init_io");
        STK_PNG_SYM(read_info, "png_read_info"); STK_PNG_SYM(get_IHDR, "png_get_IHDR"); STK_PNG_SYM(get_valid, "png_get_valid");
        STK_PNG_SYM(set_palette_to_rgb, "png_set_palette_to_rgb"); STK_PNG_SYM(set_expand_gray_1_2_4_to_8, "png_set_expand_gray_1_2_4_to_8");
        STK_PNG_SYM(set_swap, "png_set_swap"); STK_PNG_SYM(set_bgr, "png_set_bgr"); STK_PNG_SYM(set_interlace_handling, "png_set_interlace_handling");
        STK_PNG_SYM(set_gray_to_rgb, "png_set_gray_to_rgb"); STK_PNG_SYM(set_tRNS_to_alpha, "png_set_tRNS_to_alpha");
        STK_PNG_SYM(read_update_info, "png_read_update_info"); STK_PNG_SYM(get_rowbytes, "png_get_rowbytes"); STK_PNG_SYM(read_row, "png_read_row");
#undef STK_PNG_SYM
        a.ok = a.get_libpng_ver && a.create_read_struct && a.create_info_struct && a.destroy_read_struct && a.set_longjmp_fn && a.longjmp_ &&
               a.init_io && a.read_info && a.get_IHDR && a.get_valid && a.set_palette_to_rgb && a.set_expand_gray_1_2_4_to_8 && a.set_swap &&
               a.set_bgr && a.set_gray_to_rgb && a.set_tRNS_to_alpha && a.set_interlace_handling && a.read_update_info && a.get_rowbytes && a.read_row;
        return a;
    }();
    return api;
}

static void png_quiet_warning(void*, const char*) {}
static void png_trap_error(void* png, const char*) { png_api().longjmp_(png, 1); }   // must not return

// One pass over the file. pix == nullptr: geometry only (w, h, cn, depth out). Else the decoded rows go to pix, which holds
// w * h * cn * depth / 8 bytes. libpng reports errors by longjmp into this frame: it owns no object with a destructor and
// touches nothing but PODs, the FILE* and the caller's buffer (the reason for the two passes: the buffer exists beforehand).
// 0 ok; 1 not decodable; 2 a flavour this build does not take.
static int png_decode_raw(const PngApi& api, const char* path, int* w, int* h, int* cn, int* depth, unsigned char* pix) noexcept {
    FILE* volatile f = std::fopen(path, "rb");
    if (!f) return 1;
    void* volatile png = api.create_read_struct(api.get_libpng_ver(nullptr), nullptr, png_trap_error, png_quiet_warning);
    if (!png) { std::fclose(f); return 2; }
    void* volatile info = api.create_info_struct(png);
    if (!info) { void* p0 = png; api.destroy_read_struct(&p0, nullptr, nullptr); std::fclose(f); return 2; }
    jmp_buf* jb = api.set_longjmp_fn(png, longjmp, sizeof(jmp_buf));
    if (!jb || setjmp(*jb)) {
        void* p0 = png; void* i0 = info;
        api.destroy_read_struct(&p0, &i0, nullptr);
        std::fclose(f);
        return 1;
    }
    api.init_io(png, f);
    api.read_info(png, info);
    uint32_t iw = 0, ih = 0;
    int bits = 0, color = 0, interlace = 0, comp = 0, filter = 0;
    api.get_IHDR(png, info, &iw, &ih, &bits, &color, &interlace, &comp, &filter);
    int rc = 0;
    const bool has_trns = api.get_valid(png, info, 0x10u /* PNG_INFO_tRNS */) != 0;
    const bool has_alpha = (color & 4) != 0 || has_trns;
    if (iw == 0 || ih == 0 || iw > 65500 || ih > 65500) rc = 1;
    else {
        *w = (int)iw; *h = (int)ih;
        // IMREAD_UNCHANGED keeps an alpha plane: every PNG with alpha (RGBA, grey + alpha, a tRNS chunk) comes out of
        // OpenCV's PngDecoder as FOUR channels, B G R A (grey replicated) [OCV-RECALL]; the reference then stacks all four
        *cn = has_alpha ? 4 : (color & 2) ? 3 : 1;           // PNG_COLOR_MASK_COLOR (palette images carry it too)
        *depth = bits == 16 ? 16 : 8;
        if (pix) {
            if (color == 3) api.set_palette_to_rgb(png);     // PNG_COLOR_TYPE_PALETTE
            if ((color & 2) == 0 && bits < 8) api.set_expand_gray_1_2_4_to_8(png);
            if (has_trns) api.set_tRNS_to_alpha(png);
            if (has_alpha && (color & 2) == 0) api.set_gray_to_rgb(png);
            if (bits == 16) api.set_swap(png);               // big-endian file order -> native (x86-64)
            if (*cn >= 3) api.set_bgr(png);
            const int passes = api.set_interlace_handling(png);
            api.read_update_info(png, info);
            const size_t row = (size_t)iw * *cn * (*depth / 8);
            if (api.get_rowbytes(png, info) != row) rc = 2;
            else
                for (volatile int pass = 0; pass < passes; pass = pass + 1)
                    for (volatile uint32_t y = 0; y < ih; y = y + 1) api.read_row(png, pix + row * y, nullptr);
        }
    }
    void* p0 = png; void* i0 = info;
    api.destroy_read_struct(&p0, &i0, nullptr);
    std::fclose(f);
    return rc;
}

// 0 decoded into pix (BGR or grey, 8 or 16 bit); 1 not decodable (read error); 2 a PNG flavour this build does not take
int png_load(const char* path, Pnm& p, std::vector<unsigned char>& pix) {
    const PngApi& api = png_api();
    if (!api.ok) return 2;
    int w = 0, h = 0, cn = 0, depth = 0;
    int rc = png_decode_raw(api, path, &w, &h, &cn, &depth, nullptr);
    if (rc) return rc;
    pix.resize((size_t)w * h * cn * (depth / 8));
    rc = png_decode_raw(api, path, &w, &h, &cn, &depth, pix.data());
    if (rc) return rc;
    p.w = w; p.h = h; p.cn = cn; p.depth = depth; p.data_ofs = 0;
    return 0;
}

// ---- TIFF through libtiff's handle-based API, resolved at run time (no struct layouts involved) --------------------
struct TiffApi {
    void* (*open)(const char*, const char*) = nullptr;
    void (*close)(void*) = nullptr;
    int (*get_field)(void*, uint32_t, ...) = nullptr;
    int (*read_scanline)(void*, void*, uint32_t, uint16_t) = nullptr;
    long (*scanline_size)(void*) = nullptr;
    int (*is_tiled)(void*) = nullptr;
    long (*read_tile)(void*, void*, uint32_t, uint32_t, uint32_t, uint16_t) = nullptr;   // optional: tiled files
    long (*tile_size)(void*) = nullptr;
    void* (*set_error_handler)(void*) = nullptr;
    void* (*set_warning_handler)(void*) = nullptr;
    bool ok = false;
};

const TiffApi& tiff_api() {
    static const TiffApi api = []() {
        TiffApi a;
        void* h = dlopen("libtiff.so.5", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libtiff.so.6", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.open = reinterpret_cast<void* (*)(const char*, const char*)>(dlsym(h, "TIFFOpen"));
        a.close = reinterpret_cast<void (*)(void*)>(dlsym(h, "TIFFClose"));
        a.get_field = reinterpret_cast<int (*)(void*, uint32_t, ...)>(dlsym(h, "TIFFGetField"));
        a.read_scanline = reinterpret_cast<int (*)(void*, void*, uint32_t, uint16_t)>(dlsym(h, "TIFFReadScanline"));
        a.scanline_size = reinterpret_cast<long (*)(void*)>(dlsym(h, "TIFFScanlineSize"));
        a.is_tiled = reinterpret_cast<int (*)(void*)>(dlsym(h, "TIFFIsTiled"));
        a.read_tile = reinterpret_cast<long (*)(void*, void*, uint32_t, uint32_t, uint32_t, uint16_t)>(dlsym(h, "TIFFReadTile"));
        a.tile_size = reinterpret_cast<long (*)(void*)>(dlsym(h, "TIFFTileSize"));
        a.set_error_handler = reinterpret_cast<void* (*)(void*)>(dlsym(h, "TIFFSetErrorHandler"));
        a.set_warning_handler = reinterpret_cast<void* (*)(void*)>(dlsym(h, "TIFFSetWarningHandler"));
        a.ok = a.open && a.close && a.get_field && a.read_scanline && a.scanline_size && a.is_tiled;
        if (a.ok && a.set_error_handler && a.set_warning_handler) { a.set_error_handler(nullptr); a.set_warning_handler(nullptr); }   // quiet
        return a;
    }();
    return api;
}

// swap channels 0 and 2 of `px` pixels of CN channels of T (R G B [A] -> B G R [A])
template <typename T, int CN>
static void swap_02(const T* src, T* dst, size_t px) {
    for (size_t i = 0; i < px; i++) {
        const T a = src[CN * i], c = src[CN * i + 2];
        dst[CN * i] = c; dst[CN * i + 1] = src[CN * i + 1]; dst[CN * i + 2] = a;
        if (CN == 4) dst[CN * i + 3] = src[CN * i + 3];
    }
}
// one decoded row (RGB / RGBA / grey samples in native byte order) -> B G R [A] / grey
static void tiff_row_out(const unsigned char* line, unsigned char* o, uint32_t w, int cn, int bps) {
    if (cn == 1) std::memcpy(o, line, (size_t)w * (bps / 8));
    else if (cn == 3 && bps == 8) swap_rb_u8(line, o, w);
    else if (cn == 3) swap_02<uint16_t, 3>(reinterpret_cast<const uint16_t*>(line), reinterpret_cast<uint16_t*>(o), w);
    else if (bps == 8) swap_02<unsigned char, 4>(line, o, w);
    else swap_02<uint16_t, 4>(reinterpret_cast<const uint16_t*>(line), reinterpret_cast<uint16_t*>(o), w);
}

// 8- or 16-bit, grey (MINISBLACK), RGB or RGBA (IMREAD_UNCHANGED keeps the fourth sample: B G R A), contiguous; stripped or
// — round 4 — tiled (any compression libtiff handles) -> grey / BGR / BGRA
// 0 decoded; 1 not decodable; 2 a flavour this build does not take
int tiff_load(const char* path, Pnm& p, std::vector<unsigned char>& pix) {
    const TiffApi& api = tiff_api();
    if (!api.ok) return 2;
    void* t = api.open(path, "r");
    if (!t) return 1;
    uint32_t w = 0, h = 0;
    uint16_t bps = 1, spp = 1, photo = 0, planar = 1;
    api.get_field(t, 256, &w); api.get_field(t, 257, &h);                  // IMAGEWIDTH, IMAGELENGTH
    api.get_field(t, 258, &bps); api.get_field(t, 277, &spp);              // BITSPERSAMPLE, SAMPLESPERPIXEL
    api.get_field(t, 262, &photo); api.get_field(t, 284, &planar);         // PHOTOMETRIC, PLANARCONFIG
    const bool grey = spp == 1 && photo == 1, rgb = (spp == 3 || spp == 4) && photo == 2;
    const bool tiled = api.is_tiled(t) != 0;
    if (w == 0 || h == 0 || (bps != 8 && bps != 16) || (!grey && !rgb) || planar != 1 || (tiled && !(api.read_tile && api.tile_size))) { api.close(t); return 2; }
    p.w = (int)w; p.h = (int)h; p.cn = rgb ? spp : 1; p.depth = bps; p.data_ofs = 0;
    const size_t esz = (size_t)p.cn * (bps / 8), row = (size_t)w * esz;
    if (!tiled) {
        if ((size_t)api.scanline_size(t) < row) { api.close(t); return 1; }
        pix.resize(row * h);
        std::vector<unsigned char> line((size_t)api.scanline_size(t));
        for (uint32_t y = 0; y < h; y++) {
            if (api.read_scanline(t, line.data(), y, 0) < 0) { api.close(t); return 1; }
            tiff_row_out(line.data(), pix.data() + row * y, w, p.cn, bps);
        }
    } else {
        uint32_t tw = 0, th = 0;
        api.get_field(t, 322, &tw); api.get_field(t, 323, &th);            // TILEWIDTH, TILELENGTH
        const long tsz = api.tile_size(t);
        if (tw == 0 || th == 0 || tsz <= 0 || (size_t)tsz < (size_t)tw * th * esz) { api.close(t); return 1; }
        pix.resize(row * h);
        std::vector<unsigned char> tile((size_t)tsz), band(row);
        for (uint32_t y0 = 0; y0 < h; y0 += th) {
            const uint32_t rows = std::min(th, h - y0);
            // a band of tiles: decode each tile once, hand its rows over one image row at a time
            std::vector<std::vector<unsigned char>> tiles;
            for (uint32_t x0 = 0; x0 < w; x0 += tw) {
                if (api.read_tile(t, tile.data(), x0, y0, 0, 0) < 0) { api.close(t); return 1; }
                tiles.push_back(tile);
            }
            for (uint32_t y = 0; y < rows; y++) {
                for (uint32_t x0 = 0, k = 0; x0 < w; x0 += tw, k++)
                    std::memcpy(band.data() + (size_t)x0 * esz, tiles[k].data() + (size_t)y * tw * esz, (size_t)std::min(tw, w - x0) * esz);
                tiff_row_out(band.data(), pix.data() + row * (y0 + y), w, p.cn, bps);
            }
        }
    }
    api.close(t);
    return 0;
}


// ---- BMP (no library): uncompressed 24-bit -> BGR (the file's own order), 32-bit -> B G R A (IMREAD_UNCHANGED keeps the
// fourth byte), 8-bit palette -> BGR, or one grey channel when every palette entry is grey [OCV-RECALL: BmpDecoder's
// IsColorPalette]; bottom-up or top-down rows, rows padded to 4 bytes. 1 / 4 / 16-bit and RLE files: not taken.
// 0 decoded; 1 not decodable; 2 a flavour this build does not take
int bmp_load(const std::vector<unsigned char>& f, Pnm& p, std::vector<unsigned char>& pix) {
    auto u16 = [&](size_t o) { return (uint32_t)f[o] | ((uint32_t)f[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (f.size() < 54 || f[0] != 'B' || f[1] != 'M') return 1;
    const uint32_t data_ofs = u32(10), hsz = u32(14);
    if (hsz < 40 || 14 + (size_t)hsz > f.size()) return hsz == 12 ? 2 : 1;          // (OS/2 core headers: not taken)
    const int32_t w = (int32_t)u32(18), hraw = (int32_t)u32(22);
    const uint32_t planes = u16(26), bpp = u16(28), comp = u32(30);
    uint32_t ncol = u32(46);
    const bool top_down = hraw < 0;
    const int64_t h = top_down ? -(int64_t)hraw : hraw;
    if (w <= 0 || h <= 0 || w > 65500 || h > 65500 || planes != 1) return 1;
    if (!(bpp == 8 || bpp == 24 || bpp == 32) || !(comp == 0 || (comp == 3 && bpp == 32))) return 2;   // BI_RGB; BI_BITFIELDS only as 32-bit
    if (comp == 3) {                       // the usual masks only: B G R (A) in memory order
        const size_t mo = 14 + 40;
        if (hsz < 52 && f.size() < mo + 12) return 1;
        if (u32(mo) != 0x00ff0000u || u32(mo + 4) != 0x0000ff00u || u32(mo + 8) != 0x000000ffu) return 2;
    }
    const size_t src_row = (((size_t)w * bpp + 31) / 32) * 4;
    if ((size_t)data_ofs + src_row * (size_t)h > f.size()) return 1;
    const unsigned char* pal = nullptr;
    bool grey_pal = false;
    if (bpp == 8) {
        if (ncol == 0 || ncol > 256) ncol = 256;
        const size_t po = 14 + (size_t)hsz;
        if (po + 4 * (size_t)ncol > f.size()) return 1;
        pal = f.data() + po;                                  // B G R 0 per entry
        grey_pal = true;
        for (uint32_t i = 0; i < ncol; i++) grey_pal = grey_pal && pal[4 * i] == pal[4 * i + 1] && pal[4 * i] == pal[4 * i + 2];
    }
    const int cn = bpp == 32 ? 4 : (bpp == 24 || !grey_pal) ? 3 : 1;
    pix.resize((size_t)w * h * cn);
    for (int64_t y = 0; y < h; y++) {
        const unsigned char* s = f.data() + data_ofs + src_row * (size_t)(top_down ? y : h - 1 - y);
        unsigned char* o = pix.data() + (size_t)w * cn * (size_t)y;
        if (bpp != 8) std::memcpy(o, s, (size_t)w * cn);
        else if (grey_pal) for (int32_t x = 0; x < w; x++) o[x] = s[x] < ncol ? pal[4 * s[x]] : 0;
        else for (int32_t x = 0; x < w; x++) { const unsigned char* e = pal + 4 * (s[x] < ncol ? s[x] : 0); o[3 * x] = e[0]; o[3 * x + 1] = e[1]; o[3 * x + 2] = e[2]; }
    }
    p.w = w; p.h = (int)h; p.cn = cn; p.depth = 8; p.data_ofs = 0;
    return 0;
}

// ---- WebP through libwebp's simple decoding API (libwebp.so.7), resolved at run time -------------------------------------
// OpenCV's WebPDecoder [OCV-RECALL: grfmt_webp.cpp] asks WebPGetFeatures for the geometry, gives the Mat four channels when
// the bitstream has alpha and three otherwise (IMREAD_UNCHANGED), and decodes with WebPDecodeBGRAInto / WebPDecodeBGRInto:
// the same two calls here, so the pixels are the library's own. Animated files: not taken.
struct WebPFeatures { int width, height, has_alpha, has_animation, format; uint32_t pad[5]; };   // WebPBitstreamFeatures (decode.h, ABI 0x02xx)
static_assert(sizeof(WebPFeatures) == 40, "WebPBitstreamFeatures layout");
struct WebPApi {
    int (*get_features)(const uint8_t*, size_t, WebPFeatures*, int) = nullptr;                  // WebPGetFeaturesInternal -> VP8StatusCode
    uint8_t* (*bgr_into)(const uint8_t*, size_t, uint8_t*, size_t, int) = nullptr;
    uint8_t* (*bgra_into)(const uint8_t*, size_t, uint8_t*, size_t, int) = nullptr;
    bool ok = false;
};
const WebPApi& webp_api() {
    static const WebPApi api = [] {
        WebPApi a;
        void* h = dlopen("libwebp.so.7", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.get_features = reinterpret_cast<decltype(a.get_features)>(dlsym(h, "WebPGetFeaturesInternal"));
        a.bgr_into = reinterpret_cast<decltype(a.bgr_into)>(dlsym(h, "WebPDecodeBGRInto"));
        a.bgra_into = reinterpret_cast<decltype(a.bgra_into)>(dlsym(h, "WebPDecodeBGRAInto"));
        a.ok = a.get_features && a.bgr_into && a.bgra_into;
        return a;
    }();
    return api;
}

// 0 decoded (BGR or BGRA, 8 bit); 1 not decodable; 2 a flavour / library this build does not take
int webp_load(const std::vector<unsigned char>& f, Pnm& p, std::vector<unsigned char>& pix) {
    const WebPApi& api = webp_api();
    if (!api.ok) return 2;
    WebPFeatures ft{};
    const int st = api.get_features(f.data(), f.size(), &ft, 0x0209);           // WEBP_DECODER_ABI_VERSION of the 1.x series
    if (st == 2) return 2;                                                      // VP8_STATUS_INVALID_PARAM: another ABI
    if (st != 0 || ft.width <= 0 || ft.height <= 0 || ft.width > 16383 || ft.height > 16383) return 1;
    if (ft.has_animation) return 2;
    const int cn = ft.has_alpha ? 4 : 3;
    pix.resize((size_t)ft.width * ft.height * cn);
    uint8_t* got = (cn == 4 ? api.bgra_into : api.bgr_into)(f.data(), f.size(), pix.data(), pix.size(), ft.width * cn);
    if (got != pix.data()) return 1;
    p.w = ft.width; p.h = ft.height; p.cn = cn; p.depth = 8; p.data_ofs = 0;
    return 0;
}

// ---- JPEG through libjpeg-turbo's classic API (libjpeg.so.8), resolved at run time --------------------------------------
// The reference's own data set is JPEG (README.md:18, examples/main.rs:35) and OpenCV decodes it with this very library
// family at its default settings (JDCT_ISLOW, fancy upsampling), which is what is requested here, so the pixels are
// OpenCV's. The image ships the library but not jpeglib.h: the leading part of jpeg_decompress_struct (stable since
// libjpeg 6b) and jpeg_error_mgr are declared below, and nothing is taken on trust —
//   * the struct size comes from the library itself: jpeg_CreateDecompress is first called with a guess, and on a mismatch
//     the library's error (JERR_BAD_STRUCT_SIZE) carries the size it expects;
//   * the field offsets are checked against an independent parse of the file's SOF marker (width, height, components)
//     after jpeg_read_header, and output_scanline must advance exactly as rows are delivered;
//   * any disagreement ends in STK_NOT_IMPLEMENTED ("decode on the caller's side"), never in a wrong image.
struct JpegErrorMgr {                       // struct jpeg_error_mgr
    void (*error_exit)(void*);
    void (*emit_message)(void*, int);
    void (*output_message)(void*);
    void (*format_message)(void*, char*);
    void (*reset_error_mgr)(void*);
    int msg_code;
    union { int i[8]; char s[80]; } msg_parm;
    int trace_level;
    long num_warnings;
    const char* const* jpeg_message_table;
    int last_jpeg_message;
    const char* const* addon_message_table;
    int first_addon_message, last_addon_message;
    char slack[128];                        // room for a larger library-side struct
};
struct JpegDecompressHead {                 // the leading fields of struct jpeg_decompress_struct (x86-64 layout)
    JpegErrorMgr* err; void* mem; void* progress; void* client_data;
    int is_decompressor, global_state;
    void* src;
    unsigned image_width, image_height;
    int num_components, jpeg_color_space, out_color_space;
    unsigned scale_num, scale_denom;
    double output_gamma;
    int buffered_image, raw_data_out, dct_method, do_fancy_upsampling, do_block_smoothing, quantize_colors, dither_mode,
        two_pass_quantize, desired_number_of_colors, enable_1pass_quant, enable_external_quant, enable_2pass_quant;
    unsigned output_width, output_height;
    int out_color_components, output_components, rec_outbuf_height, actual_number_of_colors;
    void* colormap;
    unsigned output_scanline;
};
static_assert(offsetof(JpegDecompressHead, image_width) == 48 && offsetof(JpegDecompressHead, output_gamma) == 80 &&
              offsetof(JpegDecompressHead, output_width) == 136 && offsetof(JpegDecompressHead, output_scanline) == 168, "jpeglib.h layout");

struct JpegApi {
    JpegErrorMgr* (*std_error)(JpegErrorMgr*) = nullptr;
    void (*create)(void*, int version, size_t structsize) = nullptr;
    void (*mem_src)(void*, const unsigned char*, unsigned long) = nullptr;
    int (*read_header)(void*, int require_image) = nullptr;
    int (*start)(void*) = nullptr;
    unsigned (*read_scanlines)(void*, unsigned char**, unsigned) = nullptr;
    int (*finish)(void*) = nullptr;
    void (*destroy)(void*) = nullptr;
    bool ok = false;
};
const JpegApi& jpeg_api() {
    static const JpegApi api = []() {
        JpegApi a;
        void* h = dlopen("libjpeg.so.8", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.std_error = reinterpret_cast<decltype(a.std_error)>(dlsym(h, "jpeg_std_error"));
        a.create = reinterpret_cast<decltype(a.create)>(dlsym(h, "jpeg_CreateDecompress"));
        a.mem_src = reinterpret_cast<decltype(a.mem_src)>(dlsym(h, "jpeg_mem_src"));
        a.read_header = reinterpret_cast<decltype(a.read_header)>(dlsym(h, "jpeg_read_header"));
        a.start = reinterpret_cast<decltype(a.start)>(dlsym(h, "jpeg_start_decompress"));
        a.read_scanlines = reinterpret_cast<decltype(a.read_scanlines)>(dlsym(h, "jpeg_read_scanlines"));
        a.finish = reinterpret_cast<decltype(a.finish)>(dlsym(h, "jpeg_finish_decompress"));
        a.destroy = reinterpret_cast<decltype(a.destroy)>(dlsym(h, "jpeg_destroy_decompress"));
        a.ok = a.std_error && a.create && a.mem_src && a.read_header && a.start && a.read_scanlines && a.finish && a.destroy;
        return a;
    }();
    return api;
}

struct JpegTrap { jmp_buf jb; int code = 0, parm0 = 0; };
void jpeg_trap_exit(void* cinfo) {          // error_exit: libjpeg must not return from it
    JpegDecompressHead* c = static_cast<JpegDecompressHead*>(cinfo);
    JpegTrap* t = static_cast<JpegTrap*>(c->client_data);
    t->code = c->err->msg_code; t->parm0 = c->err->msg_parm.i[0];
    longjmp(t->jb, 1);
}
void jpeg_quiet(void*) {}
void jpeg_quiet_emit(void*, int) {}

// width / height / components of the first SOFn marker, parsed independently of the library. False if none is found.
bool jpeg_sof(const unsigned char* b, size_t n, unsigned& w, unsigned& h, int& comps) {
    if (n < 4 || b[0] != 0xff || b[1] != 0xd8) return false;
    size_t i = 2;
    while (i + 4 <= n) {
        if (b[i] != 0xff) return false;
        const unsigned m = b[i + 1];
        if (m == 0xff) { i++; continue; }                                   // fill byte
        if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) { i += 2; continue; }
        const size_t len = ((size_t)b[i + 2] << 8) | b[i + 3];
        if (len < 2 || i + 2 + len > n) return false;
        if (m >= 0xc0 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            if (len < 8) return false;
            h = ((unsigned)b[i + 5] << 8) | b[i + 6]; w = ((unsigned)b[i + 7] << 8) | b[i + 8]; comps = b[i + 9];
            return true;
        }
        if (m == 0xda) return false;                                        // scan data before any frame header
        i += 2 + len;
    }
    return false;
}

// The libjpeg calls proper. longjmp lands back in this frame, so it owns nothing with a destructor and changes no
// automatic object after setjmp except through `volatile` PODs and caller-provided raw buffers (ISO C++ [support.runtime]:
// anything else would be indeterminate after the jump): `store` holds the decompress object (capacity >= 4096 + 256),
// `pix` the sw * shh * (1 or 3) output bytes, `line` one decoded row (scomp samples per pixel). 0 decoded; 1 not decodable; 2 not taken by this build.
static int jpeg_decode_raw(const JpegApi& api, const unsigned char* file, size_t file_size, unsigned sw, unsigned shh, int scomp,
                           unsigned char* store, size_t store_cap, unsigned char* pix, unsigned char* line, int* struct_size_io) noexcept {
    JpegErrorMgr err;
    JpegTrap trap;
    JpegDecompressHead* const c = reinterpret_cast<JpegDecompressHead*>(store);
    volatile bool created = false;
    volatile int attempts = 0;
    volatile int struct_size = *struct_size_io;
    if (setjmp(trap.jb)) {
        // JERR_BAD_STRUCT_SIZE is the only error that can arrive before the object exists: msg_parm.i[0] = expected size
        if (!created && attempts < 2 && trap.parm0 > (int)sizeof(JpegDecompressHead) && (size_t)trap.parm0 + 256 <= store_cap &&
            trap.parm0 != struct_size) {
            struct_size = trap.parm0;                       // and once more from the top, through the same setjmp
        } else {
            if (created) api.destroy(c);
            return created ? 1 : 2;
        }
    }
    if (!created) {
        std::memset(store, 0, store_cap);
        std::memset(&err, 0, sizeof err);
        c->err = api.std_error(&err);
        err.error_exit = jpeg_trap_exit; err.output_message = jpeg_quiet; err.emit_message = jpeg_quiet_emit;
        c->client_data = &trap;
        attempts = attempts + 1;
        api.create(c, 80, (size_t)struct_size);             // may longjmp with the size the library expects
        c->client_data = &trap;                             // jpeg_CreateDecompress zeroes the struct but keeps err
        created = true;
        *struct_size_io = struct_size;
    }
    api.mem_src(c, file, (unsigned long)file_size);
    if (api.read_header(c, 1) != 1) { api.destroy(c); return 1; }
    // the declared layout must agree with the file: otherwise hands off
    if (c->image_width != sw || c->image_height != shh || c->num_components != scomp) { api.destroy(c); return 2; }
    // JCS_GRAYSCALE / JCS_RGB / JCS_CMYK (library defaults otherwise). Four components (CMYK, or YCCK which the library turns
    // into CMYK): OpenCV's JpegDecoder asks for JCS_CMYK and converts to three channels itself [OCV-RECALL: grfmt_jpeg.cpp,
    // icvCvt_CMYK2BGR_8u_C4C3R] — the file's samples are Adobe's inverted CMYK: c' = k - ((255 - c) * k >> 8), B G R = y' m' c'.
    c->out_color_space = scomp == 4 ? 4 : scomp == 3 ? 2 : 1;
    api.start(c);
    if (c->output_width != sw || c->output_height != shh || c->output_components != scomp || c->output_scanline != 0) { api.destroy(c); return 2; }
    const size_t row = (size_t)sw * (scomp == 1 ? 1 : 3);
    for (volatile unsigned y = 0; y < shh; y = y + 1) {
        unsigned char* lp = line;
        const unsigned got = api.read_scanlines(c, &lp, 1);
        const unsigned at = c->output_scanline;             // read before the object goes away
        if (got != 1 || at != y + 1) { api.destroy(c); return at != y + 1 ? 2 : 1; }
        unsigned char* o = pix + row * y;
        if (scomp == 1) std::memcpy(o, lp, row);
        else if (scomp == 3) for (unsigned x = 0; x < sw; x++) { o[3 * x] = lp[3 * x + 2]; o[3 * x + 1] = lp[3 * x + 1]; o[3 * x + 2] = lp[3 * x]; }   // RGB -> BGR
        else for (unsigned x = 0; x < sw; x++) {
            const int k = lp[4 * x + 3];
            o[3 * x + 2] = (unsigned char)(k - (((255 - lp[4 * x]) * k) >> 8));
            o[3 * x + 1] = (unsigned char)(k - (((255 - lp[4 * x + 1]) * k) >> 8));
            o[3 * x] = (unsigned char)(k - (((255 - lp[4 * x + 2]) * k) >> 8));
        }
    }
    api.finish(c);
    api.destroy(c);
    return 0;
}

// 0 decoded (BGR or grey, 8 bit; CMYK / YCCK files as BGR); 1 not decodable; 2 a flavour / library this build does not take
int jpeg_load(const std::vector<unsigned char>& file, Pnm& p, std::vector<unsigned char>& pix) {
    const JpegApi& api = jpeg_api();
    if (!api.ok) return 2;
    unsigned sw = 0, shh = 0; int scomp = 0;
    if (!jpeg_sof(file.data(), file.size(), sw, shh, scomp)) return 1;
    if (sw == 0 || shh == 0 || sw > 65500 || shh > 65500) return 1;
    if (scomp != 1 && scomp != 3 && scomp != 4) return 2;
    static std::atomic<int> struct_size_shared{656};                       // libjpeg-turbo 2.x, v8 ABI, x86-64; corrected by the handshake
    int struct_size = struct_size_shared.load();
    // every buffer exists before the first libjpeg call: the geometry comes from the independent SOF parse above
    std::vector<unsigned char> store(4096 + 256), line((size_t)sw * scomp);
    const int ocomp = scomp == 1 ? 1 : 3;                                    // (four components come out as B G R, like OpenCV's)
    pix.resize((size_t)sw * shh * ocomp);
    const int rc = jpeg_decode_raw(api, file.data(), file.size(), sw, shh, scomp, store.data(), store.size(), pix.data(), line.data(), &struct_size);
    struct_size_shared.store(struct_size);
    if (rc == 0) { p.w = (int)sw; p.h = (int)shh; p.cn = ocomp; p.depth = 8; p.data_ofs = 0; }
    return rc;
}

// 0 ok; else a status with the message set. PNM: `file` holds the file, raster at p.data_ofs; PNG: `file` holds the decoded pixels.
stk_status load_image(stk_ctx* ctx, const char* path, std::vector<unsigned char>& file, Pnm& p) {
    if (!path) return fail(ctx, STK_INVALID_PARAMS, "null path");
    if (has_ext(path, ".png")) {
        const int rc = png_load(path, p, file);
        if (rc == 0) { p.data_ofs = (size_t)-1; return STK_OK; }                      // already decoded
        if (rc == 1) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot decode '") + path + "' (empty Mat -> cvtColor fails)");
        return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: '") + path + "': a PNG flavour this build does not decode, or libpng is "
                                              "missing (libpng16.so.16 " + (png_api().ok ? "loaded" : "not found") + ")");
    }
    if (has_ext(path, ".bmp") || has_ext(path, ".dib")) {
        std::vector<unsigned char> raw;
        if (!read_file(path, raw)) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot read '") + path + "' (empty Mat -> cvtColor fails)");
        const int rc = bmp_load(raw, p, file);
        if (rc == 0) { p.data_ofs = (size_t)-1; return STK_OK; }
        if (rc == 1) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot decode '") + path + "' (empty Mat -> cvtColor fails)");
        return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: '") + path + "': only uncompressed 8-bit palette, 24- and 32-bit BMP is decoded in this build");
    }
    if (has_ext(path, ".tif") || has_ext(path, ".tiff")) {
        const int rc = tiff_load(path, p, file);
        if (rc == 0) { p.data_ofs = (size_t)-1; return STK_OK; }
        if (rc == 1) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot decode '") + path + "' (empty Mat -> cvtColor fails)");
        return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: '") + path + "': only contiguous 8/16-bit grey, RGB or RGBA TIFF is decoded in this "
                                              "build (libtiff " + (tiff_api().ok ? "loaded" : "not found") + ")");
    }
    if (has_ext(path, ".jpg") || has_ext(path, ".jpeg") || has_ext(path, ".jpe")) {
        std::vector<unsigned char> raw;
        if (!read_file(path, raw)) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot read '") + path + "' (empty Mat -> cvtColor fails)");
        const int rc = jpeg_load(raw, p, file);
        if (rc == 0) { p.data_ofs = (size_t)-1; return STK_OK; }
        if (rc == 1) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot decode '") + path + "' (empty Mat -> cvtColor fails)");
        return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: '") + path + "': only grey / YCbCr / CMYK JPEG through libjpeg-turbo's libjpeg.so.8 is decoded "
                                              "in this build (library " + (jpeg_api().ok ? "loaded" : "not found") + "); decode it on the caller's side");
    }
    if (has_ext(path, ".webp")) {
        std::vector<unsigned char> raw;
        if (!read_file(path, raw)) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot read '") + path + "' (empty Mat -> cvtColor fails)");
        const int rc = webp_load(raw, p, file);
        if (rc == 0) { p.data_ofs = (size_t)-1; return STK_OK; }
        if (rc == 1) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot decode '") + path + "' (empty Mat -> cvtColor fails)");
        return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: '") + path + "': only still WebP through libwebp.so.7 is decoded in this build (library " +
                                              (webp_api().ok ? "loaded" : "not found") + ")");
    }
    for (const char* e : {".exr", ".jp2", ".hdr"})
        if (has_ext(path, e))
            return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: no codec for '") + path + "' in this build; decode it "
                                                  "on the caller's side and use the frame-based entry points");
    if (!read_file(path, file)) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot read '") + path + "' (empty Mat -> cvtColor fails)");
    const int rc = pnm_header(file.data(), file.size(), p);
    if (rc == 2) return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: PNM maxval other than 255 / 65535 in '") + path + "'");
    if (rc != 0) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: '") + path + "' is not an image this build can decode");
    const size_t need = (size_t)p.w * p.h * p.cn * (p.depth / 8);
    if (file.size() - p.data_ofs < need) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: '") + path + "' is truncated");
    return STK_OK;
}

// The reference decodes inside its Rayon fold (read_grey_and_f32 at lib.rs:200, 756 runs on every worker): files are
// decoded here by a pool of host threads as well, straight into ONE page-locked block (frame i at i * frame_bytes), so
// that the host-fed pipeline behind the frame-based entry points moves them by DMA at link rate. Frame 0 is decoded first
// (it fixes the geometry every other file must match, lib.rs:166).
// `on_mixed_sizes`: what to do when the files decode to frames of differing SIZE (same channels and depth). The reference's
// keypoint path takes such a stack (every frame is read on its own, lib.rs:200-204, and warped into the first frame's
// size, lib.rs:290-299); its ECC path fails on it in cv::add (lib.rs:809). Null: the mismatch is reported as that failure.
template <typename Call>
stk_status match_files(stk_ctx* ctx, const char* const* paths, int32_t n, Call call, const std::function<stk_status()>* on_mixed_sizes = nullptr) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (n <= 0 || !paths) return fail(ctx, STK_NOT_ENOUGH_FILES, "Not enough files");      // lib.rs:155-157, 725-727
    Pnm first;
    std::vector<unsigned char> file0;
    stk_status st = load_image(ctx, paths[0], file0, first);
    if (st) return st;
    const size_t fbytes = (size_t)first.w * first.h * first.cn * (first.depth / 8);
    // page-locked when the runtime grants it, plain memory otherwise; kept by the context and re-used by the next call
    // (round 4: allocating and locking the block per call was a third of a 256-frame 4K call)
    if (ctx->files_block_cap < fbytes * (size_t)n) {
        if (ctx->files_block) { if (ctx->files_block_pinned) (void)hipHostFree(ctx->files_block); else std::free(ctx->files_block); }
        ctx->files_block = nullptr; ctx->files_block_cap = 0;
        if (hipHostMalloc((void**)&ctx->files_block, fbytes * (size_t)n, hipHostMallocDefault) == hipSuccess) ctx->files_block_pinned = true;
        else { (void)hipGetLastError(); ctx->files_block = (unsigned char*)std::malloc(fbytes * (size_t)n); ctx->files_block_pinned = false; }
        if (!ctx->files_block) return fail(ctx, STK_PROCESSING_ERROR, "out of host memory for the decoded stack");
        ctx->files_block_cap = fbytes * (size_t)n;
    }
    struct { unsigned char* p; } block{ctx->files_block};
    auto place = [&](int i, std::vector<unsigned char>& file, const Pnm& p) {
        unsigned char* dst = block.p + fbytes * (size_t)i;
        if (p.data_ofs == (size_t)-1) std::memcpy(dst, file.data(), fbytes);             // decoded by a codec library
        else pnm_decode(file.data() + p.data_ofs, p, dst);
    };
    place(0, file0, first);
    file0.clear(); file0.shrink_to_fit();
    // Frames 1 .. n-1 on a pool of threads WHILE the engine already runs: the call below starts at once, its uploader asks
    // the gate before it copies a frame (AsyncUpload), so decode, PCIe transfer, template preparation and alignment of
    // different frames overlap. The first failure (lowest index) is the one reported, like a sequential loop.
    const int workers = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16, (size_t)std::max(n - 1, 1)});
    std::atomic<int> next{1};
    std::atomic<bool> size_mismatch{false};
    std::mutex em;
    std::condition_variable ecv;
    std::vector<char> state(n, 0);                       // 0 pending, 1 decoded, 2 failed (guarded by em)
    state[0] = 1;
    int err_index = n;
    stk_status err_status = STK_OK;
    std::string err_msg;
    auto work = [&]() {
        stk_ctx local;                                   // per-thread error text (load_image writes into the context it is given)
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            std::vector<unsigned char> file;
            Pnm p;
            stk_status s2 = load_image(&local, paths[i], file, p);
            if (!s2 && (p.cn != first.cn || p.depth != first.depth))
                s2 = fail(&local, STK_BACKEND_ERROR, std::string("'") + paths[i] + "' differs in size or type from the first frame (channels / depth)");
            else if (!s2 && (p.w != first.w || p.h != first.h)) {
                size_mismatch.store(true);
                s2 = fail(&local, STK_BACKEND_ERROR, std::string("'") + paths[i] + "' differs in size or type from the first frame (" +
                                  std::to_string(p.w) + " x " + std::to_string(p.h) + " against " + std::to_string(first.w) + " x " +
                                  std::to_string(first.h) + "): cv::add needs one size (lib.rs:809)");
            }
            if (!s2) place(i, file, p);
            {
                std::lock_guard<std::mutex> lk(em);
                state[i] = s2 ? 2 : 1;
                if (s2 && i < err_index) { err_index = i; err_status = s2; err_msg = local.err; }
            }
            ecv.notify_all();
        }
    };
    // (joined on every way out of this function: a thread that cannot be started — std::system_error — or a throwing
    // allocation below must not leave joinable threads behind, whose destructors would terminate the process)
    struct Joiner {
        std::vector<std::thread> threads;
        std::atomic<int>* next; int n;
        ~Joiner() { next->store(n); for (auto& t : threads) if (t.joinable()) t.join(); }
    } pool{{}, &next, n};
    pool.threads.reserve(workers);
    try {
        for (int t = 0; t < workers; t++) pool.threads.emplace_back(work);
    } catch (const std::exception&) {
        if (pool.threads.empty()) return fail(ctx, STK_PROCESSING_ERROR, "could not start a decoder thread");
    }                                                     // fewer threads than planned still decode everything
    FrameGate gate;
    gate.wait = [&](const void* ptr) -> bool {
        const size_t i = ((const unsigned char*)ptr - block.p) / fbytes;
        std::unique_lock<std::mutex> lk(em);
        ecv.wait(lk, [&]() { return state[i] != 0 || err_status != STK_OK; });
        return state[i] == 1;
    };
    std::vector<void*> ptrs(n);
    for (int i = 0; i < n; i++) ptrs[i] = block.p + fbytes * (size_t)i;
    stk_frames fr{};
    fr.data = ptrs.data(); fr.n = n; fr.width = first.w; fr.height = first.h; fr.channels = first.cn; fr.depth = first.depth;
    fr.location = STK_HOST; fr.row_stride_bytes = 0;
    ctx->frame_gate = &gate;
    st = call(&fr);
    ctx->frame_gate = nullptr;
    for (auto& t : pool.threads) t.join();
    if (err_status && size_mismatch.load() && on_mixed_sizes) return (*on_mixed_sizes)();   // (an unreadable file shows up there again)
    if (err_status) return fail(ctx, err_status, err_msg);   // a file that could not be decoded outranks whatever the engine made of it
    return st;
}

// keypoint_match on files that decode to frames of differing size: every file decoded into a buffer of its own (a pool of
// host threads), then stk_keypoint_match_mixed on the host frames
stk_status keypoint_files_mixed(stk_ctx* ctx, const char* const* paths, int32_t n, const stk_keypoint_params* params, float scale_down_width,
                                stk_image_f32* out, int32_t* dropped, stk_frame_stats* stats) {
    std::vector<std::vector<unsigned char>> pix(n);
    std::vector<stk_frame_geometry> geo(n);
    std::vector<Pnm> heads(n);
    std::vector<stk_status> sts(n, STK_OK);
    std::vector<std::string> msgs(n);
    std::atomic<int> next{0};
    auto work = [&]() {
        stk_ctx local;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            std::vector<unsigned char> file;
            sts[i] = load_image(&local, paths[i], file, heads[i]);
            if (sts[i]) { msgs[i] = local.err; continue; }
            const Pnm& p = heads[i];
            if (p.data_ofs == (size_t)-1) pix[i].swap(file);
            else { pix[i].resize((size_t)p.w * p.h * p.cn * (p.depth / 8)); pnm_decode(file.data() + p.data_ofs, p, pix[i].data()); }
            geo[i] = stk_frame_geometry{p.w, p.h, 0};
        }
    };
    {
        const int workers = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16, (size_t)n});
        struct Joiner { std::vector<std::thread> t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } pool;
        try { for (int t = 1; t < workers; t++) pool.t.emplace_back(work); } catch (const std::exception&) {}
        work();
    }
    for (int i = 0; i < n; i++) if (sts[i]) return fail(ctx, sts[i], msgs[i]);
    for (int i = 1; i < n; i++)
        if (heads[i].cn != heads[0].cn || heads[i].depth != heads[0].depth)
            return fail(ctx, STK_BACKEND_ERROR, std::string("'") + paths[i] + "' differs in type from the first frame (channels / depth)");
    std::vector<const void*> ptrs(n);
    for (int i = 0; i < n; i++) ptrs[i] = pix[i].data();
    stk_frames fr{};
    fr.data = ptrs.data(); fr.n = n; fr.width = heads[0].w; fr.height = heads[0].h; fr.channels = heads[0].cn; fr.depth = heads[0].depth;
    fr.location = STK_HOST;
    return stk_keypoint_match_mixed(ctx, &fr, geo.data(), params, scale_down_width, out, dropped, stats);
}

}  // namespace

extern "C" {

stk_status stk_imread(stk_ctx* ctx, const char* path, void* data, size_t capacity_bytes, int32_t* width, int32_t* height,
                      int32_t* channels, int32_t* depth) {
    std::vector<unsigned char> file;
    Pnm p;
    stk_status st = load_image(ctx, path, file, p);
    if (st) return st;
    if (width) *width = p.w;
    if (height) *height = p.h;
    if (channels) *channels = p.cn;
    if (depth) *depth = p.depth;
    if (!data) return STK_OK;                                  // geometry query
    const size_t need = (size_t)p.w * p.h * p.cn * (p.depth / 8);
    if (capacity_bytes < need) return fail(ctx, STK_INVALID_PARAMS, "imread: output buffer too small");
    if (p.data_ofs == (size_t)-1) std::memcpy(data, file.data(), need);   // PNG: already decoded
    else pnm_decode(file.data() + p.data_ofs, p, data);
    return STK_OK;
}

stk_status stk_keypoint_match_files(stk_ctx* ctx, const char* const* paths, int32_t n, const stk_keypoint_params* params,
                                    float scale_down_width, stk_image_f32* out, int32_t* dropped, stk_frame_stats* stats) {
    const std::function<stk_status()> mixed = [&]() { return keypoint_files_mixed(ctx, paths, n, params, scale_down_width, out, dropped, stats); };
    return match_files(ctx, paths, n, [&](const stk_frames* fr) { return stk_keypoint_match(ctx, fr, params, scale_down_width, out, dropped, stats); }, &mixed);
}

stk_status stk_ecc_match_files(stk_ctx* ctx, const char* const* paths, int32_t n, const stk_ecc_params* params,
                               float scale_down_width, stk_image_f32* out, stk_frame_stats* stats) {
    return match_files(ctx, paths, n, [&](const stk_frames* fr) { return stk_ecc_match(ctx, fr, params, scale_down_width, out, stats); });
}


stk_status stk_hybrid_match_files(stk_ctx* ctx, const char* const* paths, int32_t n, const stk_keypoint_params* kp_params,
                                  const stk_ecc_params* ecc_params, stk_image_f32* out, stk_frame_stats* stats) {
    return match_files(ctx, paths, n, [&](const stk_frames* fr) { return stk_hybrid_match(ctx, fr, kp_params, ecc_params, out, stats); });
}

}  // extern "C"

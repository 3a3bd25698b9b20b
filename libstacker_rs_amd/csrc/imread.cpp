// imread.cpp — the file front-end of the reference's entry points: `imgcodecs::imread(path,
// IMREAD_UNCHANGED)` (utils.rs:110-117, 132) for the one family of formats this build can decode without external codec
// libraries — binary PNM (P5 grey, P6 colour; 8 or 16 bit) — and keypoint_match / ecc_match in the reference's own call
// shape, a list of paths (lib.rs:129-137, 702-710). 8-bit RGB / grey PNG is decoded through libpng's simplified API when
// libpng16.so.16 can be loaded at run time (it is installed in the image, its headers are not: the four entry points and
// the png_image struct of png.h 1.6 are declared below). JPEG / TIFF / other PNG flavours (alpha, 16-bit, palette) return
// STK_NOT_IMPLEMENTED and the caller decodes them itself (the frame-based entry points are the boundary). Stripped 8/16-bit
// grey / RGB TIFF goes through libtiff's handle-based API (TIFFOpen / TIFFGetField / TIFFReadScanline), loaded the same way. A file that is missing or not an image behaves as in the reference: imread gives an
// empty Mat and the following cvtColor raises -> STK_BACKEND_ERROR (OpenCvError).
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
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

// decode into `dst` (w*h*cn samples of depth/8 bytes): RGB -> BGR, 16-bit big-endian -> native
void pnm_decode(const unsigned char* raster, const Pnm& p, void* dst) {
    const size_t px = (size_t)p.w * p.h;
    if (p.depth == 8) {
        unsigned char* o = (unsigned char*)dst;
        if (p.cn == 1) std::memcpy(o, raster, px);
        else for (size_t i = 0; i < px; i++) { o[3 * i] = raster[3 * i + 2]; o[3 * i + 1] = raster[3 * i + 1]; o[3 * i + 2] = raster[3 * i]; }
    } else {
        unsigned short* o = (unsigned short*)dst;
        auto be = [&](size_t s) { return (unsigned short)((raster[2 * s] << 8) | raster[2 * s + 1]); };
        if (p.cn == 1) for (size_t i = 0; i < px; i++) o[i] = be(i);
        else for (size_t i = 0; i < px; i++) { o[3 * i] = be(3 * i + 2); o[3 * i + 1] = be(3 * i + 1); o[3 * i + 2] = be(3 * i); }
    }
}

// ---- PNG through libpng 1.6's simplified API, resolved at run time ------------------------------------------
struct PngImage {            // png_image of png.h (libpng 1.6, PNG_IMAGE_VERSION 1)
    void* opaque;
    uint32_t version, width, height, format, flags, colormap_entries, warning_or_error;
    char message[64];
};
constexpr uint32_t PNG_FMT_GRAY = 0x00, PNG_FMT_RGB = 0x02, PNG_FMT_BGR = 0x12;   // COLOR = 0x02, BGR = 0x10

struct PngApi {
    int (*begin_read_from_file)(PngImage*, const char*) = nullptr;
    int (*finish_read)(PngImage*, const void* background, void* buffer, int32_t row_stride, void* colormap) = nullptr;
    void (*image_free)(PngImage*) = nullptr;
    bool ok = false;
};

const PngApi& png_api() {
    static const PngApi api = []() {
        PngApi a;
        void* h = dlopen("libpng16.so.16", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.begin_read_from_file = reinterpret_cast<int (*)(PngImage*, const char*)>(dlsym(h, "png_image_begin_read_from_file"));
        a.finish_read = reinterpret_cast<int (*)(PngImage*, const void*, void*, int32_t, void*)>(dlsym(h, "png_image_finish_read"));
        a.image_free = reinterpret_cast<void (*)(PngImage*)>(dlsym(h, "png_image_free"));
        a.ok = a.begin_read_from_file && a.finish_read && a.image_free;
        return a;
    }();
    return api;
}

// 0 decoded into pix (BGR or grey, 8 bit); 1 not decodable (read error); 2 a PNG flavour this build does not take
int png_load(const char* path, Pnm& p, std::vector<unsigned char>& pix) {
    const PngApi& api = png_api();
    if (!api.ok) return 2;
    PngImage im;
    std::memset(&im, 0, sizeof im);
    im.version = 1;
    if (!api.begin_read_from_file(&im, path)) { api.image_free(&im); return 1; }
    if (im.format != PNG_FMT_RGB && im.format != PNG_FMT_GRAY) { api.image_free(&im); return 2; }   // alpha / 16-bit / palette
    p.w = (int)im.width; p.h = (int)im.height; p.cn = im.format == PNG_FMT_RGB ? 3 : 1; p.depth = 8; p.data_ofs = 0;
    if (im.format == PNG_FMT_RGB) im.format = PNG_FMT_BGR;      // imread returns BGR
    pix.resize((size_t)p.w * p.h * p.cn);
    if (!api.finish_read(&im, nullptr, pix.data(), 0, nullptr)) { api.image_free(&im); return 1; }
    return 0;                                                     // finish_read frees the image on success
}

// ---- TIFF through libtiff's handle-based API, resolved at run time (no struct layouts involved) --------------------
struct TiffApi {
    void* (*open)(const char*, const char*) = nullptr;
    void (*close)(void*) = nullptr;
    int (*get_field)(void*, uint32_t, ...) = nullptr;
    int (*read_scanline)(void*, void*, uint32_t, uint16_t) = nullptr;
    long (*scanline_size)(void*) = nullptr;
    int (*is_tiled)(void*) = nullptr;
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
        a.set_error_handler = reinterpret_cast<void* (*)(void*)>(dlsym(h, "TIFFSetErrorHandler"));
        a.set_warning_handler = reinterpret_cast<void* (*)(void*)>(dlsym(h, "TIFFSetWarningHandler"));
        a.ok = a.open && a.close && a.get_field && a.read_scanline && a.scanline_size && a.is_tiled;
        if (a.ok && a.set_error_handler && a.set_warning_handler) { a.set_error_handler(nullptr); a.set_warning_handler(nullptr); }   // quiet
        return a;
    }();
    return api;
}

// 8- or 16-bit, grey (MINISBLACK) or RGB, contiguous, stripped TIFF (any compression libtiff handles) -> grey / BGR
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
    const bool grey = spp == 1 && photo == 1, rgb = spp == 3 && photo == 2;
    if (w == 0 || h == 0 || (bps != 8 && bps != 16) || (!grey && !rgb) || planar != 1 || api.is_tiled(t)) { api.close(t); return 2; }
    p.w = (int)w; p.h = (int)h; p.cn = rgb ? 3 : 1; p.depth = bps; p.data_ofs = 0;
    const size_t row = (size_t)w * p.cn * (bps / 8);
    if ((size_t)api.scanline_size(t) < row) { api.close(t); return 1; }
    pix.resize(row * h);
    std::vector<unsigned char> line((size_t)api.scanline_size(t));
    for (uint32_t y = 0; y < h; y++) {
        if (api.read_scanline(t, line.data(), y, 0) < 0) { api.close(t); return 1; }
        unsigned char* o = pix.data() + row * y;
        if (!rgb) std::memcpy(o, line.data(), row);
        else if (bps == 8) for (uint32_t x = 0; x < w; x++) { o[3 * x] = line[3 * x + 2]; o[3 * x + 1] = line[3 * x + 1]; o[3 * x + 2] = line[3 * x]; }
        else {
            const uint16_t* s16 = reinterpret_cast<const uint16_t*>(line.data());   // libtiff returns native byte order
            uint16_t* o16 = reinterpret_cast<uint16_t*>(o);
            for (uint32_t x = 0; x < w; x++) { o16[3 * x] = s16[3 * x + 2]; o16[3 * x + 1] = s16[3 * x + 1]; o16[3 * x + 2] = s16[3 * x]; }
        }
    }
    api.close(t);
    return 0;
}

// 0 ok; else a status with the message set. PNM: `file` holds the file, raster at p.data_ofs; PNG: `file` holds the decoded pixels.
stk_status load_image(stk_ctx* ctx, const char* path, std::vector<unsigned char>& file, Pnm& p) {
    if (!path) return fail(ctx, STK_INVALID_PARAMS, "null path");
    if (has_ext(path, ".png")) {
        const int rc = png_load(path, p, file);
        if (rc == 0) { p.data_ofs = (size_t)-1; return STK_OK; }                      // already decoded
        if (rc == 1) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot decode '") + path + "' (empty Mat -> cvtColor fails)");
        return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: '") + path + "': only 8-bit RGB / grey PNG without alpha is decoded in this "
                                              "build (libpng16.so.16 " + (png_api().ok ? "loaded" : "not found") + ")");
    }
    if (has_ext(path, ".tif") || has_ext(path, ".tiff")) {
        const int rc = tiff_load(path, p, file);
        if (rc == 0) { p.data_ofs = (size_t)-1; return STK_OK; }
        if (rc == 1) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot decode '") + path + "' (empty Mat -> cvtColor fails)");
        return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: '") + path + "': only stripped 8/16-bit grey or RGB TIFF is decoded in this "
                                              "build (libtiff " + (tiff_api().ok ? "loaded" : "not found") + ")");
    }
    for (const char* e : {".jpg", ".jpeg", ".jpe", ".bmp", ".webp", ".exr"})
        if (has_ext(path, e))
            return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: no codec for '") + path + "' in this build (binary PNM only); decode it "
                                                  "on the caller's side and use the frame-based entry points");
    if (!read_file(path, file)) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: cannot read '") + path + "' (empty Mat -> cvtColor fails)");
    const int rc = pnm_header(file.data(), file.size(), p);
    if (rc == 2) return fail(ctx, STK_NOT_IMPLEMENTED, std::string("imread: PNM maxval other than 255 / 65535 in '") + path + "'");
    if (rc != 0) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: '") + path + "' is not an image this build can decode");
    const size_t need = (size_t)p.w * p.h * p.cn * (p.depth / 8);
    if (file.size() - p.data_ofs < need) return fail(ctx, STK_BACKEND_ERROR, std::string("imread: '") + path + "' is truncated");
    return STK_OK;
}

template <typename Call>
stk_status match_files(stk_ctx* ctx, const char* const* paths, int32_t n, Call call) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (n <= 0 || !paths) return fail(ctx, STK_NOT_ENOUGH_FILES, "Not enough files");      // lib.rs:155-157, 725-727
    std::vector<std::vector<unsigned char>> pix(n);
    std::vector<void*> ptrs(n);
    Pnm first;
    for (int i = 0; i < n; i++) {
        std::vector<unsigned char> file;
        Pnm p;
        stk_status st = load_image(ctx, paths[i], file, p);
        if (st) return st;
        if (i == 0) first = p;
        else if (p.w != first.w || p.h != first.h || p.cn != first.cn || p.depth != first.depth)
            return fail(ctx, STK_INVALID_PARAMS, std::string("'") + paths[i] + "' differs in size or type from the first frame");
        if (p.data_ofs == (size_t)-1) pix[i].swap(file);             // PNG: already decoded
        else {
            pix[i].resize((size_t)p.w * p.h * p.cn * (p.depth / 8));
            pnm_decode(file.data() + p.data_ofs, p, pix[i].data());
        }
        ptrs[i] = pix[i].data();
    }
    stk_frames fr{};
    fr.data = ptrs.data(); fr.n = n; fr.width = first.w; fr.height = first.h; fr.channels = first.cn; fr.depth = first.depth;
    fr.location = STK_HOST; fr.row_stride_bytes = 0;
    return call(&fr);
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
    return match_files(ctx, paths, n, [&](const stk_frames* fr) { return stk_keypoint_match(ctx, fr, params, scale_down_width, out, dropped, stats); });
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

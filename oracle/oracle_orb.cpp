// oracle_orb.cpp — CPU restatement of ORB::create_def().detect_and_compute (utils.rs:174-183;
// OpenCV features2d/src/orb.cpp, fast.cpp, fast_score.cpp, keypoint.cpp).
// TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
//
// Defaults: nfeatures 500, scaleFactor 1.2f, nlevels 8, edgeThreshold 31, firstLevel 0, WTA_K 2,
// HARRIS_SCORE, patchSize 31, fastThreshold 20.
//
// Declared deviation: KeyPointsFilter::retainBest leaves the surviving keypoints in the
// implementation-defined order of std::nth_element/std::partition. The SET it keeps is well
// defined (everything with response >= the n-th best) and is reproduced exactly; the ORDER used
// here is (response descending, then raster order y, x), level after level.
#include "oracle_common.h"
#include "../libstacker_rs_amd/csrc/orb_pattern.h"   // shared data table (not logic)

using namespace orc;

namespace {

struct KP { float x, y, size, angle, response; int octave; int lx, ly; };   // lx,ly = integer level coordinates

// INTER_LINEAR_EXACT resize of an 8-bit image (imgproc/src/resize.cpp resize_bitExact):
// coefficients in 8.8 fixed point, horizontal then vertical, result (v + 2^15) >> 16.
struct LinCoef { int ofs; int c0, c1; };
static void lin_coeffs(int src, int dst, std::vector<LinCoef>& out) {
    out.resize(dst);
    const double inv_scale = (double)dst / src;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; d++) {
        const double fval = scale * (d + 0.5) - 0.5;
        const int ival = cv_floor(fval);
        LinCoef c;
        if (ival >= 0 && src > 1) {
            if (ival < src - 1) {
                c.ofs = ival;
                c.c1 = cv_round((fval - ival) * 256.0);
                c.c0 = 256 - c.c1;
            } else { c.ofs = src - 1; c.c0 = 256; c.c1 = 0; }
        } else { c.ofs = 0; c.c0 = 256; c.c1 = 0; }
        out[d] = c;
    }
}
static void resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    std::vector<LinCoef> cx, cy;
    lin_coeffs(sw, dw, cx); lin_coeffs(sh, dh, cy);
    for (int y = 0; y < dh; y++) {
        const uint8_t* r0 = src + (size_t)cy[y].ofs * sw;
        const uint8_t* r1 = src + (size_t)std::min(cy[y].ofs + 1, sh - 1) * sw;
        for (int x = 0; x < dw; x++) {
            const int o = cx[x].ofs, o1 = std::min(o + 1, sw - 1);
            const uint32_t h0 = (uint32_t)cx[x].c0 * r0[o] + (uint32_t)cx[x].c1 * r0[o1];
            const uint32_t h1 = (uint32_t)cx[x].c0 * r1[o] + (uint32_t)cx[x].c1 * r1[o1];
            const uint32_t v = (uint32_t)cy[y].c0 * h0 + (uint32_t)cy[y].c1 * h1;
            dst[(size_t)y * dw + x] = (uint8_t)std::min<uint32_t>((v + (1u << 15)) >> 16, 255u);
        }
    }
}

// FAST-9/16: circle offsets (x, y) of fast_score.cpp makeOffsets(patternSize 16)
static const int CIRCLE[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// corner strength: the largest t for which the pixel is still a FAST-9 corner (cornerScore<16>);
// returns 0 when the pixel is not a corner at `threshold`.
static inline int fast_score(const uint8_t* p, int stride, int threshold) {
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = v - p[CIRCLE[k][1] * stride + CIRCLE[k][0]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best = 0;   // max over arcs of min(d) (darker ring) and of min(-d) (brighter ring)
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) { mn = std::min(mn, d[k + j]); mx = std::max(mx, d[k + j]); }
        best = std::max(best, std::max(mn, -mx));
    }
    return best > threshold ? best - 1 : 0;
}

// cv::fastAtan2 (degrees), core/src/mathfuncs_core.simd.hpp atan_f32
static inline float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// GaussianBlur(level, 7x7, sigma 2, REFLECT_101) on 8-bit data as the generic separable filter
// runs it for a sub-matrix (float kernel; row pass in index order, symmetric column pass, cvRound).
static void gauss7_u8(const uint8_t* src, int w, int h, uint8_t* dst) {
    double kd[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; kd[i] = std::exp(-0.5 * x * x / 4.0); sum += kd[i]; }
    float k[7];
    for (int i = 0; i < 7; i++) k[i] = (float)(kd[i] * (1.0 / sum));
    std::vector<float> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float acc = k[0] * (float)s[border_interpolate(x - 3, w, BORDER_REFLECT_101)];
            for (int i = 1; i < 7; i++) acc += k[i] * (float)s[border_interpolate(x - 3 + i, w, BORDER_REFLECT_101)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        const float* c = tmp.data() + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float acc = k[3] * c[x];
            for (int i = 1; i <= 3; i++) {
                const float* a = tmp.data() + (size_t)border_interpolate(y - i, h, BORDER_REFLECT_101) * w;
                const float* b = tmp.data() + (size_t)border_interpolate(y + i, h, BORDER_REFLECT_101) * w;
                acc += k[3 + i] * (a[x] + b[x]);
            }
            int r = cv_round(acc);
            dst[(size_t)y * w + x] = (uint8_t)std::min(std::max(r, 0), 255);
        }
    }
}

static inline uint8_t at_reflect(const uint8_t* img, int w, int h, int x, int y) {
    return img[(size_t)border_interpolate(y, h, BORDER_REFLECT_101) * w + border_interpolate(x, w, BORDER_REFLECT_101)];
}

}  // namespace

extern "C" {

// level geometry exactly as ORB_Impl::detectAndCompute computes it
int orc_orb_level_sizes(int w, int h, int nlevels, int* ws, int* hs, float* scales) {
    for (int l = 0; l < nlevels; l++) {
        const float scale = (float)std::pow((double)1.2f, (double)l);
        const float inv = 1.0f / scale;
        ws[l] = cv_round((float)w * inv); hs[l] = cv_round((float)h * inv);
        scales[l] = scale;
    }
    return 0;
}

int orc_orb_features_per_level(int nfeatures, int nlevels, int* out) {
    const float factor = (float)(1.0 / (double)1.2f);
    float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { out[l] = cv_round(nd); sum += out[l]; nd *= factor; }
    out[nlevels - 1] = std::max(nfeatures - sum, 0);
    return 0;
}

int orc_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    resize_linear_exact(src, sw, sh, dst, dw, dh);
    return 0;
}

int orc_fast_score_map(const uint8_t* img, int w, int h, int threshold, uint8_t* out) {
    std::memset(out, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) out[(size_t)y * w + x] = (uint8_t)fast_score(img + (size_t)y * w + x, w, threshold);
    return 0;
}

int orc_gauss7_u8(const uint8_t* src, int w, int h, uint8_t* dst) { gauss7_u8(src, w, h, dst); return 0; }

// ORB detectAndCompute on an 8-bit grey image. keypoints: rows of 7 floats
// {x, y, size, angle, response, octave, class_id(-1)}; descriptors: rows of 32 bytes.
int orc_orb_detect_and_compute(const uint8_t* grey, int w, int h, int max_keypoints, float* kp_out,
                               uint8_t* desc_out, int* n_out) {
    const int nfeatures = 500, nlevels = 8, edge = 31, patch = 31, half = 15, fast_thr = 20;
    const float harris_k = 0.04f;
    int ws[8], hs[8], nfl[8];
    float scales[8];
    orc_orb_level_sizes(w, h, nlevels, ws, hs, scales);
    orc_orb_features_per_level(nfeatures, nlevels, nfl);

    // umax: row half-widths of the circular patch
    int umax[17];
    {
        const int vmax = cv_floor(half * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
        for (int v = 0; v <= vmax; v++) umax[v] = cv_round(std::sqrt((double)half * half - v * v));
        for (int v = half, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
    }

    std::vector<std::vector<uint8_t>> pyr(nlevels);
    for (int l = 0; l < nlevels; l++) {
        pyr[l].resize((size_t)ws[l] * hs[l]);
        if (l == 0) std::memcpy(pyr[0].data(), grey, (size_t)w * h);
        else resize_linear_exact(pyr[l - 1].data(), ws[l - 1], hs[l - 1], pyr[l].data(), ws[l], hs[l]);
    }

    std::vector<KP> all;
    for (int l = 0; l < nlevels; l++) {
        const int lw = ws[l], lh = hs[l];
        const uint8_t* img = pyr[l].data();
        std::vector<KP> kps;
        if (lw > 2 * edge && lh > 2 * edge && lw > 6 && lh > 6) {
            std::vector<uint8_t> score((size_t)lw * lh);
            orc_fast_score_map(img, lw, lh, fast_thr, score.data());
            // 3x3 non-maximum suppression (strict) + runByImageBorder(edgeThreshold), raster order
            for (int y = edge; y < lh - edge; y++)
                for (int x = edge; x < lw - edge; x++) {
                    const int s = score[(size_t)y * lw + x];
                    if (!s) continue;
                    const uint8_t* c = score.data() + (size_t)y * lw + x;
                    if (s > c[-1] && s > c[1] && s > c[-lw - 1] && s > c[-lw] && s > c[-lw + 1] && s > c[lw - 1] && s > c[lw] && s > c[lw + 1]) {
                        KP k{}; k.lx = x; k.ly = y; k.response = (float)s; k.octave = l;
                        kps.push_back(k);
                    }
                }
        }
        // retainBest(2 * n_l) by FAST score: keep everything >= the n-th best response
        auto retain = [](std::vector<KP>& v, int n) {
            if (n >= 0 && (int)v.size() > n) {
                if (n == 0) { v.clear(); return; }
                std::vector<float> r(v.size());
                for (size_t i = 0; i < v.size(); i++) r[i] = v[i].response;
                std::nth_element(r.begin(), r.begin() + (n - 1), r.end(), std::greater<float>());
                const float thr = r[n - 1];
                std::vector<KP> o;
                for (auto& k : v) if (k.response >= thr) o.push_back(k);
                v.swap(o);
            }
        };
        retain(kps, 2 * nfl[l]);
        // Harris response (blockSize 7) on the un-blurred level, reading through the REFLECT_101 border
        const float scale = 1.f / ((1 << 2) * 7 * 255.f);
        const float scale4 = scale * scale * scale * scale;
        for (auto& k : kps) {
            int a = 0, b = 0, c = 0;
            for (int dy = -3; dy <= 3; dy++)
                for (int dx = -3; dx <= 3; dx++) {
                    const int x = k.lx + dx, y = k.ly + dy;
                    auto P = [&](int xx, int yy) { return (int)at_reflect(img, lw, lh, xx, yy); };
                    const int Ix = (P(x + 1, y) - P(x - 1, y)) * 2 + (P(x + 1, y - 1) - P(x - 1, y - 1)) + (P(x + 1, y + 1) - P(x - 1, y + 1));
                    const int Iy = (P(x, y + 1) - P(x, y - 1)) * 2 + (P(x - 1, y + 1) - P(x - 1, y - 1)) + (P(x + 1, y + 1) - P(x + 1, y - 1));
                    a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
                }
            k.response = ((float)a * (float)b - (float)c * (float)c - harris_k * ((float)a + (float)b) * ((float)a + (float)b)) * scale4;
        }
        retain(kps, nfl[l]);
        std::stable_sort(kps.begin(), kps.end(), [](const KP& p, const KP& q) {
            if (p.response != q.response) return p.response > q.response;
            if (p.ly != q.ly) return p.ly < q.ly;
            return p.lx < q.lx;
        });
        // intensity-centroid angle over the circular patch (un-blurred level)
        for (auto& k : kps) {
            int m01 = 0, m10 = 0;
            for (int u = -half; u <= half; u++) m10 += u * at_reflect(img, lw, lh, k.lx + u, k.ly);
            for (int v = 1; v <= half; v++) {
                int vsum = 0;
                const int d = umax[v];
                for (int u = -d; u <= d; u++) {
                    const int vp = at_reflect(img, lw, lh, k.lx + u, k.ly + v), vm = at_reflect(img, lw, lh, k.lx + u, k.ly - v);
                    vsum += vp - vm;
                    m10 += u * (vp + vm);
                }
                m01 += v * vsum;
            }
            k.angle = fast_atan2((float)m01, (float)m10);
            k.size = patch * scales[l];
            k.x = (float)k.lx * scales[l];
            k.y = (float)k.ly * scales[l];
        }
        all.insert(all.end(), kps.begin(), kps.end());
    }

    // descriptors on the blurred levels
    std::vector<std::vector<uint8_t>> blurred(nlevels);
    std::vector<char> have(nlevels, 0);
    int n = 0;
    for (const KP& k : all) {
        if (n >= max_keypoints) break;
        const int l = k.octave;
        if (!have[l]) { blurred[l].resize(pyr[l].size()); gauss7_u8(pyr[l].data(), ws[l], hs[l], blurred[l].data()); have[l] = 1; }
        const uint8_t* img = blurred[l].data();
        const int lw = ws[l], lh = hs[l];
        const float inv = 1.f / scales[l];
        float angle = k.angle;
        angle *= (float)(3.14159265358979323846 / 180.f);
        const float a = (float)std::cos(angle), b = (float)std::sin(angle);
        const int cx = cv_round(k.x * inv), cy = cv_round(k.y * inv);
        uint8_t* desc = desc_out + (size_t)n * 32;
        for (int i = 0; i < 32; i++) {
            int val = 0;
            for (int bit = 0; bit < 8; bit++) {
                const signed char* p = ORB_BIT_PATTERN_31 + (i * 8 + bit) * 4;
                const float x0 = p[0] * a - p[1] * b, y0 = p[0] * b + p[1] * a;
                const float x1 = p[2] * a - p[3] * b, y1 = p[2] * b + p[3] * a;
                const int t0 = at_reflect(img, lw, lh, cx + cv_round(x0), cy + cv_round(y0));
                const int t1 = at_reflect(img, lw, lh, cx + cv_round(x1), cy + cv_round(y1));
                val |= (t0 < t1) << bit;
            }
            desc[i] = (uint8_t)val;
        }
        float* o = kp_out + (size_t)n * 7;
        o[0] = k.x; o[1] = k.y; o[2] = k.size; o[3] = k.angle; o[4] = k.response; o[5] = (float)k.octave; o[6] = -1.f;
        n++;
    }
    *n_out = n;
    return 0;
}

}  // extern "C"

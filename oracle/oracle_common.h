// oracle_common.h — shared helpers of the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// PARITY UNPINNED: the reference's arithmetic lives in OpenCV 4.12.0 (un-vendored,
// absent from this container) and the reference holds no golden
// vectors for the path. This oracle restates OpenCV's published algorithms from
// the notes in SURVEY.md §8a; it is anchored on the reference's call sites
// (src/lib.rs, src/utils.rs), on analytic ground truth from the synthetic
// generator and on closed-form known-answer tests — not on OpenCV outputs.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cfloat>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3,
       BORDER_REFLECT_101 = 4, BORDER_TRANSPARENT = 5 };
enum { MOTION_TRANSLATION = 0, MOTION_EUCLIDEAN = 1, MOTION_AFFINE = 2, MOTION_HOMOGRAPHY = 3 };

// cv::borderInterpolate (core/src/copy.cpp) — returns -1 for BORDER_CONSTANT outside.
inline int border_interpolate(int p, int len, int type) {
    if ((unsigned)p < (unsigned)len) return p;
    if (type == BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (type == BORDER_REFLECT || type == BORDER_REFLECT_101) {
        int delta = type == BORDER_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (type == BORDER_WRAP) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1;
}

// cvRound(double): round half to even (lrint under the default rounding mode).
inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_round(float v) { return (int)std::lrintf(v); }
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }

// saturate_cast<int>(double): cvRound with clamping to the int range.
inline int sat_int(double v) {
    if (!(v > -2147483648.0)) return std::numeric_limits<int>::min();
    if (!(v < 2147483647.0)) return std::numeric_limits<int>::max();
    return cv_round(v);
}

// 3x3 inverse in double by the adjugate (what cv::invert does for 3x3 CV_64F, DECOMP_LU).
// Returns false when det == 0 (OpenCV then leaves a zero matrix).
inline bool invert3x3(const double* m, double* out) {
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.0) { for (int i = 0; i < 9; i++) out[i] = 0; return false; }
    d = 1.0 / d;
    double t[9];
    t[0] = (m[4] * m[8] - m[5] * m[7]) * d;
    t[1] = (m[2] * m[7] - m[1] * m[8]) * d;
    t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    t[3] = (m[5] * m[6] - m[3] * m[8]) * d;
    t[4] = (m[0] * m[8] - m[2] * m[6]) * d;
    t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    t[6] = (m[3] * m[7] - m[4] * m[6]) * d;
    t[7] = (m[1] * m[6] - m[0] * m[7]) * d;
    t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    for (int i = 0; i < 9; i++) out[i] = t[i];
    return true;
}

// cv::invertAffineTransform: 2x3 double.
inline void invert_affine(const double* m, double* out) {
    double D = m[0] * m[4] - m[1] * m[3];
    D = D != 0 ? 1.0 / D : 0;
    double A11 = m[4] * D, A22 = m[0] * D, A12 = -m[1] * D, A21 = -m[3] * D;
    double b1 = -A11 * m[2] - A12 * m[5];
    double b2 = -A21 * m[2] - A22 * m[5];
    out[0] = A11; out[1] = A12; out[2] = b1;
    out[3] = A21; out[4] = A22; out[5] = b2;
    out[6] = 0; out[7] = 0; out[8] = 1;
}

}  // namespace orc

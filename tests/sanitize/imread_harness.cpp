// imread_harness.cpp — stk_imread (hand-written PNM parser, dlopen'ed libpng / libtiff / libjpeg with hand-declared
// structs) over every file named on the command line, under ASan + UBSan. Prints "<status> <w> <h> <c> <depth> <checksum>".
#include "hip_stubs.h"
#include "../../libstacker_rs_amd/csrc/imread.cpp"

// the path-based entry points of imread.cpp call into the engine; they are not exercised here
extern "C" {
stk_status stk_keypoint_match(stk_ctx*, const stk_frames*, const stk_keypoint_params*, float, stk_image_f32*, int32_t*, stk_frame_stats*) { return STK_HIP_ERROR; }
stk_status stk_ecc_match(stk_ctx*, const stk_frames*, const stk_ecc_params*, float, stk_image_f32*, stk_frame_stats*) { return STK_HIP_ERROR; }
stk_status stk_hybrid_match(stk_ctx*, const stk_frames*, const stk_keypoint_params*, const stk_ecc_params*, stk_image_f32*, stk_frame_stats*) { return STK_HIP_ERROR; }
stk_status stk_keypoint_match_mixed(stk_ctx*, const stk_frames*, const stk_frame_geometry*, const stk_keypoint_params*, float, stk_image_f32*, int32_t*, stk_frame_stats*) { return STK_HIP_ERROR; }
}

int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        int32_t w = 0, h = 0, c = 0, d = 0;
        stk_status st = stk_imread(nullptr, argv[i], nullptr, 0, &w, &h, &c, &d);
        unsigned long sum = 0;
        if (st == STK_OK) {
            std::vector<unsigned char> buf((size_t)w * h * c * (d / 8));
            st = stk_imread(nullptr, argv[i], buf.data(), buf.size(), nullptr, nullptr, nullptr, nullptr);
            for (unsigned char b : buf) sum = sum * 131 + b;
            // one byte short: must be refused, not overrun
            if (!buf.empty() && stk_imread(nullptr, argv[i], buf.data(), buf.size() - 1, nullptr, nullptr, nullptr, nullptr) != STK_INVALID_PARAMS) return 3;
        }
        std::printf("%d %d %d %d %d %lu\n", (int)st, w, h, c, d, sum);
    }
    return 0;
}

// homography.cpp — host driver of calib3d::findHomography (lib.rs:267-276) for a batch of frames.
//
// What stays on the host is only what is inherently sequential and tiny: cv::RNG's multiply-with-carry stream, the
// sample admissibility test (collinearity + orientation, HomographyEstimatorCallback::checkSubset), and the replay of
// RANSAC's best-so-far / RANSACUpdateNumIters (or LMEDS's least-median) decision over the per-model scores that
// kernels_homography.hip computes. Models are evaluated speculatively in rounds (32, then 256, then the rest of the 2000):
// with the usual > 80 % inlier ratios the adaptive iteration count collapses below 32 after the first good model, so one
// launch covers every frame of the shard; frames that need more iterations simply take part in the next round.
// OpenCV sources this follows [OCV-RECALL]: calib3d/src/ptsetreg.cpp (RANSACPointSetRegistrator, LMeDSPointSetRegistrator),
// fundam.cpp (checkSubset, findHomography's final DLT + LM on the inliers), core/src/rand.cpp.
#include "homography.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "context.h"

namespace stk {
namespace geom {

struct HgWorkspace {
    DevBuf pts, frames, samples, scores, err, jobs, results, masks;
    void* pinned = nullptr;          // host staging for everything that crosses PCIe in a round
    size_t pinned_cap = 0;
};

HgWorkspace* hg_workspace_create() { return new HgWorkspace(); }
void hg_workspace_destroy(HgWorkspace* w) {
    if (!w) return;
    for (DevBuf* b : {&w->pts, &w->frames, &w->samples, &w->scores, &w->err, &w->jobs, &w->results, &w->masks}) b->release();
    if (w->pinned) (void)hipHostFree(w->pinned);
    delete w;
}

namespace {

// cv::RNG: 64-bit multiply-with-carry state, 32-bit outputs; uniform(a, b) = a + next() % (b - a)
class MwcStream {
public:
    MwcStream() : s_(0xffffffffffffffffull) {}               // RNG(-1), the seed both registrators use
    int below(int n) {
        s_ = (uint64_t)(uint32_t)s_ * 4164903690u + (uint32_t)(s_ >> 32);
        return (int)((uint32_t)s_ % (uint32_t)n);
    }
private:
    uint64_t s_;
};

// Is the last of the 4 points (nearly) on a line through two of the others? (haveCollinearPoints: only triples that
// contain the newest point are looked at.)
bool newest_point_collinear(const float* xy /* 4 x 2 */) {
    const double px = xy[6], py = xy[7];
    for (int j = 0; j < 3; j++) {
        const double ax = xy[2 * j] - px, ay = xy[2 * j + 1] - py;
        for (int k = 0; k < j; k++) {
            const double bx = xy[2 * k] - px, by = xy[2 * k + 1] - py;
            if (std::fabs(bx * ay - by * ax) <= FLT_EPSILON * (std::fabs(ax) + std::fabs(ay) + std::fabs(bx) + std::fabs(by))) return true;
        }
    }
    return false;
}
// determinant of [p q r] with a third coordinate of 1, expanded along the first row like cv::determinant on a Matx33d
double orientation(const float* p, const float* q, const float* r) {
    const double a = p[0], b = p[1], c = q[0], d = q[1], e = r[0], f = r[1];
    return a * (d * 1. - 1. * f) - b * (c * 1. - 1. * e) + 1. * (c * f - d * e);
}
// A sample is admissible when neither point set has its newest point on a line with two others and the four
// point triples keep (or all flip) their orientation under the mapping.
bool sample_admissible(const float* from4, const float* to4) {
    if (newest_point_collinear(from4) || newest_point_collinear(to4)) return false;
    static const int tri[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int flipped = 0;
    for (const auto& t : tri)
        flipped += orientation(from4 + 2 * t[0], from4 + 2 * t[1], from4 + 2 * t[2]) *
                   orientation(to4 + 2 * t[0], to4 + 2 * t[1], to4 + 2 * t[2]) < 0;
    return flipped == 0 || flipped == 4;
}

// RANSACUpdateNumIters(confidence p, outlier ratio ep, 4 model points, current cap)
int iterations_needed(double p, double ep, int cap) {
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    const double miss = std::max(1. - p, DBL_MIN);
    const double all_in = 1. - std::pow(1. - ep, 4);
    if (all_in < DBL_MIN) return 0;
    const double ln_miss = std::log(miss), ln_bad = std::log(all_in);
    return ln_bad >= 0 || -ln_miss >= cap * (-ln_bad) ? cap : (int)std::lrint(ln_miss / ln_bad);
}

struct Track {                       // the sequential state of one robust estimation
    int problem = -1;                // index into probs / out
    int n = 0, pt_ofs = 0;
    MwcStream rng;
    std::vector<HgSample> samples;   // every sample drawn so far, in iteration order
    bool stream_ended = false;       // no admissible sample within 1000 attempts: the loop ends there
    int replayed = 0;                // iterations whose score has been consumed
    int limit = 2000;                // RANSAC: current niters; LMEDS: fixed count
    int best = -1, best_count = 0;   // RANSAC
    double best_median = DBL_MAX;    // LMEDS
    bool finished = false;
    int round_first = 0, round_count = 0, hyp_ofs = 0, err_ofs = 0;
};

// draw the next sample of `t` (getSubset); false when 1000 attempts gave nothing admissible
bool draw_sample(Track& t, const HgProblem& pr, HgSample& s) {
    float from4[8], to4[8];
    for (int attempt = 0; attempt < 1000; attempt++) {
        for (int i = 0; i < 4; i++) {
            int pick = t.rng.below(t.n);
            for (;;) {
                bool repeated = false;
                for (int k = 0; k < i; k++) repeated |= s.idx[k] == pick;
                if (!repeated) break;
                pick = t.rng.below(t.n);
            }
            s.idx[i] = pick;
            from4[2 * i] = pr.from_pts[2 * pick]; from4[2 * i + 1] = pr.from_pts[2 * pick + 1];
            to4[2 * i] = pr.to_pts[2 * pick]; to4[2 * i + 1] = pr.to_pts[2 * pick + 1];
        }
        if (sample_admissible(from4, to4)) return true;
    }
    return false;
}

hipError_t ensure_pinned(HgWorkspace* ws, size_t bytes) {
    if (bytes <= ws->pinned_cap) return hipSuccess;
    if (ws->pinned) (void)hipHostFree(ws->pinned);
    ws->pinned = nullptr; ws->pinned_cap = 0;
    const size_t want = bytes + (bytes >> 2) + 4096;
    hipError_t e = hipHostMalloc(&ws->pinned, want, hipHostMallocDefault);
    if (e == hipSuccess) ws->pinned_cap = want;
    return e;
}

}  // namespace

int find_homography_batch(stk_ctx* ctx, hipStream_t stream, HgWorkspace* ws, const HgProblem* probs, int count, int method,
                          double thr, HgOutcome* out) {
    if (count <= 0) return STK_OK;
    if (thr <= 0) thr = 3;           // findHomography: ransacReprojThreshold <= 0 -> default 3
    const bool ransac = method == 8, lmeds = method == 4;
    // ---- admission + point packing --------------------------------------------------------------------------------
    std::vector<int> live;           // problems that reach the device
    size_t n_points = 0;
    for (int i = 0; i < count; i++) {
        out[i] = HgOutcome{};
        if (probs[i].n < 4 || (method != 0 && !ransac && !lmeds)) {
            out[i].rc = probs[i].n >= 4 && method == 16 ? 7 : 3;
            if (probs[i].mask_or_null && probs[i].n > 0) std::memset(probs[i].mask_or_null, 0, probs[i].n);
            continue;
        }
        if (probs[i].n > HG_MAX_POINTS) return fail(ctx, STK_NOT_IMPLEMENTED, "findHomography: more than 4096 correspondences");
        live.push_back(i);
        n_points += (size_t)probs[i].n;
    }
    if (live.empty()) return STK_OK;
    const int L = (int)live.size();
    HIP_TRY(ws->pts.reserve(n_points * sizeof(HgPoint)));
    HIP_TRY(ws->masks.reserve(n_points));
    HIP_TRY(ws->jobs.reserve(sizeof(HgJob) * L));
    HIP_TRY(ws->results.reserve(sizeof(HgResult) * L));
    HIP_TRY(ws->frames.reserve(sizeof(HgFrame) * L));
    // pinned staging, laid out once for the whole call so that no region is reused while a copy may still read it:
    // [points][jobs][results][masks][per-round: samples, scores, frames — worst case 2000 samples per problem]
    const size_t stage_fixed = n_points * sizeof(HgPoint) + n_points + (sizeof(HgJob) + sizeof(HgResult)) * (size_t)L + 64;
    const size_t round_max = ((sizeof(HgSample) + sizeof(int)) * (size_t)2000 + sizeof(HgFrame)) * (size_t)L + 64;
    HIP_TRY(ensure_pinned(ws, stage_fixed + round_max));
    uint8_t* const pin = (uint8_t*)ws->pinned;
    uint8_t* const round_area = pin + ((stage_fixed + 63) & ~(size_t)63);
    std::vector<Track> tracks(L);
    {
        HgPoint* hp = (HgPoint*)pin;
        size_t ofs = 0;
        for (int k = 0; k < L; k++) {
            const HgProblem& pr = probs[live[k]];
            Track& t = tracks[k];
            t.problem = live[k]; t.n = pr.n; t.pt_ofs = (int)ofs;
            for (int i = 0; i < pr.n; i++) hp[ofs + i] = HgPoint{pr.from_pts[2 * i], pr.from_pts[2 * i + 1], pr.to_pts[2 * i], pr.to_pts[2 * i + 1]};
            ofs += (size_t)pr.n;
        }
        HIP_TRY(hipMemcpyAsync(ws->pts.p, hp, n_points * sizeof(HgPoint), hipMemcpyHostToDevice, stream));
    }

    // ---- robust stage: rounds of speculative model evaluation -------------------------------------------------------
    std::vector<HgJob> jobs(L);
    for (int k = 0; k < L; k++) {
        Track& t = tracks[k];
        jobs[k] = HgJob{t.pt_ofs, t.n, {0, 0, 0, 0}, 0.f, 0};
        if (method == 0 || t.n == 4) { t.finished = true; continue; }       // plain DLT (+ LM when n > 4)
        t.limit = lmeds ? iterations_needed(0.995, 0.45, 2000) : 2000;      // LMEDS: fixed, from an assumed 45 % of outliers
    }
    const float thr2 = (float)(thr * thr);
    for (int round = 0;; round++) {
        const int round_cap = round == 0 ? 32 : round == 1 ? 256 : 2000;
        // draw this round's samples
        int n_active = 0, hyp_total = 0, max_hyp = 0;
        size_t err_total = 0;
        for (Track& t : tracks) {
            t.round_count = 0;
            if (t.finished) continue;
            const HgProblem& pr = probs[t.problem];
            t.round_first = (int)t.samples.size();
            const int want = std::min(lmeds ? t.limit : round_cap, t.limit - t.round_first);
            for (int k = 0; k < want && !t.stream_ended; k++) {
                HgSample s{};
                if (draw_sample(t, pr, s)) t.samples.push_back(s);
                else t.stream_ended = true;
            }
            t.round_count = (int)t.samples.size() - t.round_first;
            if (t.round_count == 0) { t.finished = true; continue; }        // the sample stream ended: loop over
            t.hyp_ofs = hyp_total; t.err_ofs = (int)err_total;
            hyp_total += t.round_count;
            max_hyp = std::max(max_hyp, t.round_count);
            if (lmeds) err_total += (size_t)t.round_count * t.n;
            n_active++;
        }
        if (n_active == 0) break;
        if (err_total > (size_t)1 << 30) return fail(ctx, STK_NOT_IMPLEMENTED, "findHomography(LMEDS): batch too large");
        // upload frames + samples, evaluate, fetch scores
        HIP_TRY(ws->samples.reserve(sizeof(HgSample) * (size_t)hyp_total));
        HIP_TRY(ws->scores.reserve(sizeof(int) * (size_t)hyp_total));
        if (lmeds) HIP_TRY(ws->err.reserve(sizeof(float) * err_total));
        HgSample* hs = (HgSample*)round_area;                     // the previous round ended with a stream synchronisation
        int* hscore = (int*)(hs + hyp_total);
        HgFrame* hf = (HgFrame*)(hscore + hyp_total);
        int fi = 0;
        for (Track& t : tracks) {
            if (t.finished || t.round_count == 0) continue;
            std::memcpy(hs + t.hyp_ofs, t.samples.data() + t.round_first, sizeof(HgSample) * (size_t)t.round_count);
            hf[fi++] = HgFrame{t.pt_ofs, t.n, t.hyp_ofs, t.round_count, t.err_ofs, thr2};
        }
        HIP_TRY(hipMemcpyAsync(ws->samples.p, hs, sizeof(HgSample) * (size_t)hyp_total, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(ws->frames.p, hf, sizeof(HgFrame) * (size_t)n_active, hipMemcpyHostToDevice, stream));
        HIP_TRY(launch_hg_models(ws->pts.as<HgPoint>(), ws->frames.as<HgFrame>(), n_active, max_hyp, ws->samples.as<HgSample>(),
                                 lmeds ? 1 : 0, ws->err.as<float>(), ws->scores.as<int>(), stream));
        HIP_TRY(hipMemcpyAsync(hscore, ws->scores.p, sizeof(int) * (size_t)hyp_total, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        // replay the sequential decisions
        for (Track& t : tracks) {
            if (t.finished || t.round_count == 0) continue;
            const int* sc = hscore + t.hyp_ofs;
            int it = t.round_first;
            const int end = t.round_first + t.round_count;
            for (; it < end && it < t.limit; it++) {
                const int s = sc[it - t.round_first];
                if (lmeds) {
                    float med; std::memcpy(&med, &s, 4);
                    if (s != -1 && (double)med < t.best_median) { t.best_median = med; t.best = it; }
                } else if (s > std::max(t.best_count, 3)) {                 // `good > max(maxGoodCount, modelPoints - 1)`
                    t.best = it; t.best_count = s;
                    t.limit = iterations_needed(0.995, (double)(t.n - s) / t.n, t.limit);
                }
            }
            t.replayed = it;
            out[t.problem].models_evaluated = end;
            if (it >= t.limit || t.stream_ended) t.finished = true;
        }
    }
    // ---- jobs for the refinement launch ---------------------------------------------------------------------------------
    for (int k = 0; k < L; k++) {
        Track& t = tracks[k];
        HgJob& j = jobs[k];
        if (method == 0 || t.n == 4) { j.mode = 0; continue; }
        if (t.best < 0) { j.mode = -1; continue; }                           // no admissible sample / no model with > 3 inliers
        j.mode = 1;
        for (int q = 0; q < 4; q++) j.idx[q] = t.samples[t.best].idx[q];
        if (lmeds) {
            // sigma = 2.5 * 1.4826 * (1 + 5 / (n - 4)) * sqrt(min median), at least 0.001; inliers: err <= sigma^2 (f32)
            double sigma = 2.5 * 1.4826 * (1 + 5. / (t.n - 4)) * std::sqrt(t.best_median);
            sigma = std::max(sigma, 0.001);
            j.thr2 = (float)(sigma * sigma);
        } else j.thr2 = thr2;
    }
    HgJob* hj = (HgJob*)(pin + n_points * sizeof(HgPoint));
    HgResult* hr = (HgResult*)(hj + L);
    uint8_t* hm = (uint8_t*)(hr + L);
    std::memcpy(hj, jobs.data(), sizeof(HgJob) * L);
    HIP_TRY(hipMemcpyAsync(ws->jobs.p, hj, sizeof(HgJob) * L, hipMemcpyHostToDevice, stream));
    HIP_TRY(launch_hg_refine(ws->pts.as<HgPoint>(), ws->jobs.as<HgJob>(), L, ws->results.as<HgResult>(), ws->masks.as<uint8_t>(), stream));
    HIP_TRY(hipMemcpyAsync(hr, ws->results.p, sizeof(HgResult) * L, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(hm, ws->masks.p, n_points, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int k = 0; k < L; k++) {
        const Track& t = tracks[k];
        HgOutcome& o = out[t.problem];
        o.found = hr[k].found;
        o.n_inliers = hr[k].found ? hr[k].n_inliers : 0;
        for (int q = 0; q < 9; q++) o.H[q] = hr[k].H[q];
        if (probs[t.problem].mask_or_null) std::memcpy(probs[t.problem].mask_or_null, hm + t.pt_ofs, (size_t)t.n);
    }
    return STK_OK;
}

}  // namespace geom
}  // namespace stk

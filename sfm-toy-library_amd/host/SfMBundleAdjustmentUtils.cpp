// SfMBundleAdjustmentUtils.cpp -- drop-in replacement of the reference translation unit of the same name
// (SfMToyLib/SfMBundleAdjustmentUtils.cpp): same function, same in-out contract, but ceres::Problem +
// ceres::Solve are replaced by ONE call into the MI355X back end through the C ABI of include/sfmba.h.
//
// What is kept from the reference, line for line in behaviour (not in code):
//   BA.cpp:118-122,196-199  "empty" poses (zero diagonal of R) take no part and are never written back
//   BA.cpp:123-134          R (float) -> angle-axis computed IN FLOAT, then widened; t widened from float
//   BA.cpp:138              focal = K(0,0);  BA.cpp:149-153  obs = feature - (K(0,2), K(1,2)) in float
//   BA.cpp:142-166          residual blocks in point-major order, ascending view (std::map order)
//   BA.cpp:171-177          500 iterations, 10 s; linear solver = exact Schur + dense Cholesky (== DENSE_SCHUR) unless
//                           SFMBA_LINEAR opts into the inexact block-Jacobi PCG
//   BA.cpp:180              one-line report on stdout
//   BA.cpp:182-185          anything but CONVERGENCE: "Bundle adjustment failed." on stderr, outputs untouched
//   BA.cpp:187-221          K(0,0)=K(1,1)=focal; angle-axis -> R; t; points; all narrowed to float
// Environment overrides (reference options are hard-coded, BA.cpp:171-177):
//   SFMBA_LINEAR=cholesky|pcg|auto (default cholesky; auto = cholesky up to 256 reduced unknowns, PCG above)
//   SFMBA_PRECISION=f64|f32j  SFMBA_MAX_SECONDS=<s>  SFMBA_VERBOSE=1
//   SFMBA_DUMP=<path>  writes the marshalled problem (format: sfm-toy-library_amd/problem_io.py)
//   SFMBA_SHIM_CACHE=0  disables the resident-problem cache described below
//
// The reference re-runs BA from scratch after every added view (SfM.cpp:464-466), and its cloud only ever grows (new
// points, new views of existing points: SfM.cpp:530-629).  The signature carries no incremental information, so the shim
// finds it: the marshalled observation list of the previous call is kept (host) next to the device-resident problem it was
// solved on; a call whose list CONTAINS the previous one -- checked observation by observation, coordinates included --
// uploads only the difference (sfmba_problem_append), anything else rebuilds.  The result never depends on the cache.
#include "SfMBundleAdjustmentUtils.h"

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <algorithm>
#include <iostream>
#include <mutex>
#include <vector>

#include "../../include/sfmba.h"

namespace sfmtoylib {

namespace {

// ceres::RotationMatrixToAngleAxis<float> [Ceres-upstream rotation.h]: via the unit quaternion,
// evaluated in float exactly like the template instantiation the reference uses (BA.cpp:126).
void rotationMatrixToAngleAxisFloat(const cv::Matx33f& R, float aa[3]) {
    float q0, q1, q2, q3;
    const float trace = R(0, 0) + R(1, 1) + R(2, 2);
    if (trace >= 0.0f) {
        float t = std::sqrt(trace + 1.0f);
        q0 = 0.5f * t;
        t = 0.5f / t;
        q1 = (R(2, 1) - R(1, 2)) * t;
        q2 = (R(0, 2) - R(2, 0)) * t;
        q3 = (R(1, 0) - R(0, 1)) * t;
    } else {
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        float t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0f);
        float q[4];
        q[i + 1] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (R(k, j) - R(j, k)) * t;
        q[j + 1] = (R(j, i) + R(i, j)) * t;
        q[k + 1] = (R(k, i) + R(i, k)) * t;
        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
    }
    const float sin2 = q1 * q1 + q2 * q2 + q3 * q3;
    if (sin2 > 0.0f) {
        const float s = std::sqrt(sin2);
        const float two_theta = 2.0f * ((q0 < 0.0f) ? std::atan2(-s, -q0) : std::atan2(s, q0));
        const float k = two_theta / s;
        aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
    } else {
        aa[0] = q1 * 2.0f; aa[1] = q2 * 2.0f; aa[2] = q3 * 2.0f;
    }
}

// ceres::AngleAxisToRotationMatrix<double> [Ceres-upstream], result as R(row, col).
void angleAxisToRotationMatrix(const double aa[3], double R[3][3]) {
    const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > DBL_EPSILON) {
        const double theta = std::sqrt(theta2);
        const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
        const double c = std::cos(theta), s = std::sin(theta);
        R[0][0] = c + wx * wx * (1.0 - c);       R[1][0] = wz * s + wx * wy * (1.0 - c);  R[2][0] = -wy * s + wx * wz * (1.0 - c);
        R[0][1] = wx * wy * (1.0 - c) - wz * s;  R[1][1] = c + wy * wy * (1.0 - c);       R[2][1] = wx * s + wy * wz * (1.0 - c);
        R[0][2] = wy * s + wx * wz * (1.0 - c);  R[1][2] = -wx * s + wy * wz * (1.0 - c); R[2][2] = c + wz * wz * (1.0 - c);
    } else {
        R[0][0] = 1.0;     R[1][0] = aa[2];   R[2][0] = -aa[1];
        R[0][1] = -aa[2];  R[1][1] = 1.0;     R[2][1] = aa[0];
        R[0][2] = aa[1];   R[1][2] = -aa[0];  R[2][2] = 1.0;
    }
}

const char* terminationName(int t) {
    return t == SFMBA_CONVERGENCE ? "CONVERGENCE" : t == SFMBA_NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE";
}

void dumpProblem(const char* path, int n_cam, const std::vector<double>& cam6, int n_pt, const std::vector<double>& pt3,
                 const std::vector<int32_t>& oc, const std::vector<int32_t>& op, const std::vector<double>& oxy, double focal) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return;
    const int64_t n_obs = (int64_t)oc.size();
    std::fwrite("SFMBA001", 1, 8, f);
    std::fwrite(&n_cam, 4, 1, f); std::fwrite(&n_pt, 4, 1, f); std::fwrite(&n_obs, 8, 1, f); std::fwrite(&focal, 8, 1, f);
    std::fwrite(cam6.data(), 8, cam6.size(), f); std::fwrite(pt3.data(), 8, pt3.size(), f);
    std::fwrite(oc.data(), 4, oc.size(), f); std::fwrite(op.data(), 4, op.size(), f); std::fwrite(oxy.data(), 8, oxy.size(), f);
    std::fclose(f);
}

// ---- resident-problem cache (see the header comment) ----
struct ShimCache {
    std::mutex mu;
    sfmba_problem* problem = nullptr;
    int n_cam = 0, n_pt = 0, precision = -1;
    std::vector<size_t> first;            // CSR over points of the observation list of the previous call (point-major)
    std::vector<int32_t> obs_cam;
    std::vector<double> obs_xy;
    ~ShimCache() { if (problem) sfmba_problem_destroy(problem); }
    void drop() { if (problem) sfmba_problem_destroy(problem); problem = nullptr; first.clear(); obs_cam.clear(); obs_xy.clear(); }
};
ShimCache g_cache;

// threads for the host loops over 10^5..10^6 containers
template <typename F>
void parallelRanges(size_t n, size_t grain, F fn) {
    unsigned n_thr = n >= grain ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (n_thr <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(fn, n * t / n_thr, n * (t + 1) / n_thr);
    for (auto& th : pool) th.join();
}

// Is the previous observation list contained in the new one (same (point, view) entries with the same coordinates)?  If so,
// collects the observations that are new.  Both lists are point-major with ascending view inside a point.
bool diffObservations(const ShimCache& c, int n_pt, const std::vector<size_t>& first, const std::vector<int32_t>& obs_cam,
                      const std::vector<int32_t>& obs_pt, const std::vector<double>& obs_xy, std::vector<int32_t>* new_cam,
                      std::vector<int32_t>* new_pt, std::vector<double>* new_xy) {
    if (n_pt < c.n_pt) return false;
    const size_t np = (size_t)c.n_pt;
    const unsigned n_thr = np >= 50000 ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    std::vector<std::vector<size_t>> added(n_thr);
    std::vector<char> ok(n_thr, 1);
    auto scan = [&](unsigned t) {
        for (size_t i = np * t / n_thr; i < np * (t + 1) / n_thr && ok[t]; ++i) {
            size_t a = c.first[i], a1 = c.first[i + 1];
            for (size_t b = first[i]; b < first[i + 1]; ++b) {
                if (a < a1 && c.obs_cam[a] == obs_cam[b]) {
                    if (c.obs_xy[2 * a] != obs_xy[2 * b] || c.obs_xy[2 * a + 1] != obs_xy[2 * b + 1]) { ok[t] = 0; break; }
                    ++a;
                } else if (a < a1 && c.obs_cam[a] < obs_cam[b]) { ok[t] = 0; break; }      // an old observation disappeared
                else added[t].push_back(b);
            }
            if (a != a1) ok[t] = 0;
        }
    };
    if (n_thr <= 1) scan(0);
    else { std::vector<std::thread> pool; for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(scan, t); for (auto& th : pool) th.join(); }
    for (unsigned t = 0; t < n_thr; ++t) if (!ok[t]) return false;
    for (unsigned t = 0; t < n_thr; ++t)
        for (size_t b : added[t]) { new_cam->push_back(obs_cam[b]); new_pt->push_back(obs_pt[b]); new_xy->push_back(obs_xy[2 * b]); new_xy->push_back(obs_xy[2 * b + 1]); }
    for (size_t b = first[np]; b < first[(size_t)n_pt]; ++b) { new_cam->push_back(obs_cam[b]); new_pt->push_back(obs_pt[b]); new_xy->push_back(obs_xy[2 * b]); new_xy->push_back(obs_xy[2 * b + 1]); }
    return true;
}

}  // namespace

void SfMBundleAdjustmentUtils::adjustBundle(
        PointCloud&                  pointCloud,
        std::vector<Pose>&           cameraPoses,
        Intrinsics&                  intrinsics,
        const std::vector<Features>& image2dFeatures) {

    // ---- marshal in (BA.cpp:111-166) ----
    const bool timing = std::getenv("SFMBA_SHIM_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    const int n_cam = (int)cameraPoses.size();
    std::vector<double> cam6((size_t)6 * n_cam, 0.0);
    std::vector<unsigned char> empty((size_t)n_cam, 0);
    for (int i = 0; i < n_cam; i++) {
        const Pose& pose = cameraPoses[i];
        if (pose(0, 0) == 0 && pose(1, 1) == 0 && pose(2, 2) == 0) {
            empty[i] = 1;            // not part of the optimisation unless a point references it, never written back
            continue;
        }
        const cv::Matx33f R = pose.get_minor<3, 3>(0, 0);
        float aa[3];
        rotationMatrixToAngleAxisFloat(R, aa);
        cam6[6 * i + 0] = aa[0]; cam6[6 * i + 1] = aa[1]; cam6[6 * i + 2] = aa[2];
        cam6[6 * i + 3] = pose(0, 3); cam6[6 * i + 4] = pose(1, 3); cam6[6 * i + 5] = pose(2, 3);
    }
    double focal = intrinsics.K.at<float>(0, 0);
    const float cx = intrinsics.K.at<float>(0, 2), cy = intrinsics.K.at<float>(1, 2);

    // Residual blocks in the reference's order: point-major, ascending view inside a point (std::map iteration,
    // BA.cpp:142-166).  The walk over 10^5..10^6 map nodes is the dominant host cost of the call at BASELINE config 3,
    // so it is split over a few threads: offsets first (map::size() is O(1)), then every thread fills its own range.
    const int n_pt = (int)pointCloud.size();
    std::vector<double> pt3((size_t)3 * n_pt);
    std::vector<size_t> first((size_t)n_pt + 1, 0);
    for (int i = 0; i < n_pt; i++) first[(size_t)i + 1] = first[i] + pointCloud[i].originatingViews.size();
    const size_t n_obs = first[n_pt];
    std::vector<int32_t> obs_cam(n_obs), obs_pt(n_obs);
    std::vector<double> obs_xy(2 * n_obs);
    auto fill = [&](int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            const Point3DInMap& p = pointCloud[i];
            pt3[3 * (size_t)i] = p.p.x; pt3[3 * (size_t)i + 1] = p.p.y; pt3[3 * (size_t)i + 2] = p.p.z;
            size_t k = first[i];
            for (const auto& kv : p.originatingViews) {
                cv::Point2f p2d = image2dFeatures[kv.first].points[kv.second];
                p2d.x -= cx;             // float subtraction, as the reference
                p2d.y -= cy;
                obs_cam[k] = kv.first;
                obs_pt[k] = i;
                obs_xy[2 * k] = p2d.x;
                obs_xy[2 * k + 1] = p2d.y;
                ++k;
            }
        }
    };
    {
        unsigned n_thr = n_obs >= 200000 ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (n_thr <= 1) fill(0, n_pt);
        else {
            std::vector<std::thread> pool;
            int i0 = 0;
            for (unsigned t = 0; t < n_thr; ++t) {       // equal shares of observations, not of points
                const size_t want = n_obs * (t + 1) / n_thr;
                int i1 = (t + 1 == n_thr) ? n_pt : (int)(std::upper_bound(first.begin(), first.end(), want) - first.begin() - 1);
                if (i1 < i0) i1 = i0;
                pool.emplace_back(fill, i0, i1);
                i0 = i1;
            }
            for (auto& th : pool) th.join();
        }
    }

    // ---- options (BA.cpp:171-177) ----
    sfmba_options opt;
    sfmba_options_default(&opt);
    // The reference solves the reduced camera system exactly (DENSE_SCHUR, BA.cpp:172): so does the drop-in by default.
    // The block-Jacobi PCG (faster above a few dozen views; poses and points agree with the exact solve to ~2e-7, the cost
    // to 1e-12) is an explicit opt-in: SFMBA_LINEAR=pcg, or =auto for Cholesky up to 256 reduced unknowns and PCG above.
    opt.linear_solver = SFMBA_LINEAR_CHOLESKY;
    if (const char* e = std::getenv("SFMBA_LINEAR"))
        opt.linear_solver = std::strcmp(e, "pcg") == 0 ? SFMBA_LINEAR_PCG : std::strcmp(e, "auto") == 0 ? SFMBA_LINEAR_AUTO : SFMBA_LINEAR_CHOLESKY;
    if (const char* e = std::getenv("SFMBA_PRECISION")) opt.precision = std::strcmp(e, "f32j") == 0 ? SFMBA_PRECISION_F32J : SFMBA_PRECISION_F64;
    if (const char* e = std::getenv("SFMBA_MAX_SECONDS")) opt.max_seconds = std::atof(e);
    if (const char* e = std::getenv("SFMBA_VERBOSE")) opt.verbose = std::atoi(e);
    if (const char* e = std::getenv("SFMBA_DUMP")) dumpProblem(e, n_cam, cam6, n_pt, pt3, obs_cam, obs_pt, obs_xy, focal);

    // ---- solve on the MI355X (replaces ceres::Solve, BA.cpp:179) ----
    const double t_marshalled = now();
    sfmba_summary summary;
    std::memset(&summary, 0, sizeof(summary));
    int rc;
    const char* cache_env = std::getenv("SFMBA_SHIM_CACHE");
    const bool use_cache = !(cache_env && cache_env[0] == '0');
    const char* how = "one-shot";
    if (!use_cache) {
        rc = sfmba_solve(n_cam, cam6.data(), n_pt, pt3.data(), (int64_t)obs_cam.size(), obs_cam.data(), obs_pt.data(),
                         obs_xy.data(), &focal, &opt, &summary, nullptr, 0, nullptr);
    } else {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        ShimCache& c = g_cache;
        std::vector<int32_t> new_cam, new_pt;
        std::vector<double> new_xy;
        bool grown = c.problem && c.precision == opt.precision && c.n_cam == n_cam &&
                     diffObservations(c, n_pt, first, obs_cam, obs_pt, obs_xy, &new_cam, &new_pt, &new_xy);
        const double t_diff = now();
        rc = SFMBA_OK;
        if (grown) {
            how = new_cam.empty() ? "resident" : "append";
            if (new_cam.empty() && n_pt == c.n_pt) rc = sfmba_problem_set_params(c.problem, cam6.data(), pt3.data(), focal);
            else rc = sfmba_problem_append(c.problem, n_cam, cam6.data(), n_pt, pt3.data(), (int64_t)new_cam.size(), new_cam.data(), new_pt.data(),
                                           new_xy.data(), focal);
            if (rc != SFMBA_OK) { c.drop(); grown = false; }
        }
        if (!grown) {
            how = "rebuild";
            c.drop();
            rc = sfmba_problem_create(0, opt.precision, n_cam, cam6.data(), n_pt, pt3.data(), (int64_t)obs_cam.size(), obs_cam.data(), obs_pt.data(),
                                      obs_xy.data(), focal, &c.problem);
        }
        const double setup = now() - t_diff;
        if (rc == SFMBA_OK) rc = sfmba_problem_solve(c.problem, &opt, &summary, nullptr, 0, nullptr);
        summary.setup_seconds = setup;
        // Ceres leaves the parameter blocks alone on FAILURE; every other termination hands back the best point
        if (rc == SFMBA_OK && summary.termination != SFMBA_FAILURE) rc = sfmba_problem_get_params(c.problem, cam6.data(), pt3.data(), &focal);
        if (rc == SFMBA_OK) {
            c.n_cam = n_cam; c.n_pt = n_pt; c.precision = opt.precision;
            c.first = first; c.obs_cam.swap(obs_cam); c.obs_xy.swap(obs_xy);
        } else c.drop();
    }
    const double t_solved = now();
    if (timing) std::fprintf(stderr, "[sfmba shim] path: %s\n", how);
    if (timing)
        std::fprintf(stderr, "[sfmba shim] marshal %.2f ms, sfmba_solve %.2f ms (setup %.2f + LM %.2f)\n", 1e3 * (t_marshalled - t_begin),
                     1e3 * (t_solved - t_marshalled), 1e3 * summary.setup_seconds, 1e3 * summary.seconds);
    if (rc != SFMBA_OK) {
        std::cerr << "Bundle adjustment failed. (sfmba rc=" << rc << ": " << sfmba_last_error() << ")" << std::endl;
        return;
    }
    char report[256];
    std::snprintf(report, sizeof(report), "Ceres Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
                  summary.iterations + 1, summary.initial_cost, summary.final_cost, terminationName(summary.termination));
    std::cout << report << "\n";

    if (summary.termination != SFMBA_CONVERGENCE) {
        std::cerr << "Bundle adjustment failed." << std::endl;
        return;
    }

    // ---- write back (BA.cpp:187-221) ----
    intrinsics.K.at<float>(0, 0) = (float)focal;
    intrinsics.K.at<float>(1, 1) = (float)focal;
    for (int i = 0; i < n_cam; i++) {
        if (empty[i]) continue;
        Pose& pose = cameraPoses[i];
        double R[3][3];
        angleAxisToRotationMatrix(&cam6[6 * (size_t)i], R);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) pose(r, c) = (float)R[r][c];
        pose(0, 3) = (float)cam6[6 * (size_t)i + 3];
        pose(1, 3) = (float)cam6[6 * (size_t)i + 4];
        pose(2, 3) = (float)cam6[6 * (size_t)i + 5];
    }
    parallelRanges((size_t)n_pt, 200000, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            pointCloud[i].p.x = (float)pt3[3 * i];
            pointCloud[i].p.y = (float)pt3[3 * i + 1];
            pointCloud[i].p.z = (float)pt3[3 * i + 2];
        }
    });
}

} /* namespace sfmtoylib */

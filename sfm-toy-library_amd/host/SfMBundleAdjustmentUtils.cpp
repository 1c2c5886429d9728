// SfMBundleAdjustmentUtils.cpp -- drop-in replacement of the reference translation unit of the same name
// (SfMToyLib/SfMBundleAdjustmentUtils.cpp): same function, same in-out contract, but ceres::Problem +
// ceres::Solve are replaced by ONE call into the MI355X back end through the C ABI of include/sfmba.h.
//
// What is kept from the reference, line for line in behaviour (not in code):
//   BA.cpp:118-122,196-199  "empty" poses (zero diagonal of R) take no part and are never written back
//   BA.cpp:123-134          R (float) -> angle-axis computed IN FLOAT, then widened; t widened from float
//   BA.cpp:138              focal = K(0,0);  BA.cpp:149-153  obs = feature - (K(0,2), K(1,2)) in float
//   BA.cpp:142-166          residual blocks in point-major order, ascending view (std::map order)
//   BA.cpp:171-177          500 iterations, 10 s; linear solver = the DENSE_SCHUR result (exact Schur complement; dense Cholesky, or
//                           above 256 reduced unknowns the CG run to 1e-12 with the Cholesky as fallback) unless SFMBA_LINEAR says otherwise
//   BA.cpp:180              one-line report on stdout
//   BA.cpp:182-185          anything but CONVERGENCE: "Bundle adjustment failed." on stderr, outputs untouched
//   BA.cpp:187-221          K(0,0)=K(1,1)=focal; angle-axis -> R; t; points; all narrowed to float
// Environment overrides (reference options are hard-coded, BA.cpp:171-177):
//   SFMBA_LINEAR=cholesky|pcg|auto (default auto, see sfmba.h SFMBA_LINEAR_AUTO)
//   SFMBA_PCG_TOL=<t>           CG tolerance with SFMBA_LINEAR=pcg (library default 1e-8; 1e-3 is the measured time-to-solution optimum
//                               with the LM trajectory length and the final RMS of the exact solve)
//   SFMBA_PRECISION=f64|f32j  SFMBA_MAX_SECONDS=<s>  SFMBA_VERBOSE=1
//   SFMBA_DUMP=<path>  writes the marshalled problem (format: sfm-toy-library_amd/problem_io.py)
//   SFMBA_SHIM_CACHE=0  disables the resident-problem cache described below;  SFMBA_SHIM_OVERLAP=0  its overlapped comparison
//
// The reference re-runs BA from scratch after every added view (SfM.cpp:464-466), and its cloud only ever grows (new
// points, new views of existing points: SfM.cpp:530-629).  The signature carries no incremental information, so the shim
// finds it: the marshalled observation list of the previous call is kept (host) next to the device-resident problem it was
// solved on; a call whose list CONTAINS the previous one -- checked observation by observation, coordinates included --
// uploads only the difference (sfmba_problem_append), anything else rebuilds.  The result never depends on the cache.
// The comparison of the lists that kept their length runs on worker threads while the GPU already solves (see "Overlapped
// marshalling" below; SFMBA_SHIM_OVERLAP=0 compares first, solves second); the marshalling buffers and the worker threads live
// across calls.
#include "SfMBundleAdjustmentUtils.h"

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <algorithm>
#include <atomic>
#include <iostream>
#include <condition_variable>
#include <functional>
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

template <typename VD, typename VI>
void dumpProblem(const char* path, int n_cam, const std::vector<double>& cam6, int n_pt, const VD& pt3,
                 const VI& oc, const VI& op, const VD& oxy, double focal) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return;
    const int64_t n_obs = (int64_t)oc.size();
    std::fwrite("SFMBA001", 1, 8, f);
    std::fwrite(&n_cam, 4, 1, f); std::fwrite(&n_pt, 4, 1, f); std::fwrite(&n_obs, 8, 1, f); std::fwrite(&focal, 8, 1, f);
    std::fwrite(cam6.data(), 8, cam6.size(), f); std::fwrite(pt3.data(), 8, pt3.size(), f);
    std::fwrite(oc.data(), 4, oc.size(), f); std::fwrite(op.data(), 4, op.size(), f); std::fwrite(oxy.data(), 8, oxy.size(), f);
    std::fclose(f);
}

// ---- a few resident worker threads for the host loops over 10^5..10^6 containers ----
// adjustBundle() is called once per added view (SfM.cpp:464-466): starting 16 threads three times per call cost more than the
// loops they ran (~0.3 ms per batch).  The pool is created on first use and lives until the process exits.
class WorkerPool {
public:
    static WorkerPool& instance() { static WorkerPool pool; return pool; }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // fn(t) for t = 0 .. n_tasks - 1, the caller takes part; returns when all are done
    void run(unsigned n_tasks, const std::function<void(unsigned)>& fn) {
        if (n_tasks <= 1 || workers_.empty()) { for (unsigned t = 0; t < n_tasks; ++t) fn(t); return; }
        begin(n_tasks, fn);
        end();
    }
    // the same in two halves: begin() hands the batch to the workers and returns, end() joins in and waits for the rest
    void begin(unsigned n_tasks, std::function<void(unsigned)> fn) {
        run_mu_.lock();                                      // callers on different threads take turns (released by end())
        const uint64_t g = generation_.load(std::memory_order_relaxed) + 1;
        slot_fn_[g & 1] = std::move(fn);
        slot_n_[g & 1].store(n_tasks, std::memory_order_relaxed);
        pending_.store(n_tasks, std::memory_order_relaxed);
        ticket_.store(g << 32, std::memory_order_release);
        generation_.store(g, std::memory_order_release);
        wake_sleepers();
    }
    void end() {
        work();
        for (unsigned spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin) {
            if (spin < 20000) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lk(mu_);
#if defined(__SANITIZE_THREAD__)
            // (make tsan: gcc 11's libtsan does not intercept pthread_cond_clockwait -- what wait_for compiles to -- and then believes mu_ is still held:
            // every later lock is reported as a double lock / lock-order inversion.  The untimed wait is equivalent here: the task that takes pending_ to
            // zero notifies done_ under mu_, and the predicate is tested under mu_.)
            done_.wait(lk, [&] { return pending_.load() == 0; });
#else
            done_.wait_for(lk, std::chrono::microseconds(200), [&] { return pending_.load() == 0; });
#endif
        }
        run_mu_.unlock();
    }
    // workers that went to sleep between two calls take ~0.3 ms to come back: called first thing in adjustBundle(), so that they
    // are polling again by the time the first batch is ready
    void wake() {
        std::lock_guard<std::mutex> one_batch(run_mu_);
        const uint64_t g = generation_.load(std::memory_order_relaxed) + 1;
        slot_n_[g & 1].store(0, std::memory_order_relaxed);
        pending_.store(0, std::memory_order_relaxed);
        ticket_.store(g << 32, std::memory_order_release);
        generation_.store(g, std::memory_order_release);
        wake_sleepers();
    }
private:
    WorkerPool() {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned n = std::min(64u, std::max(2u, hw / 2));      // the map walks are cache misses: threads buy memory-level parallelism
        for (unsigned t = 1; t < n; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true); }
        cv_.notify_all();
        for (auto& th : workers_) th.join();
    }
    void wake_sleepers() {
        { std::lock_guard<std::mutex> lk(mu_); if (sleepers_ == 0) return; }
        cv_.notify_all();
    }
    // Tasks are handed out through one atomic ticket (generation << 32 | next index): no lock on the way to a task.  A worker that
    // draws a ticket of a generation that is over (or an index past the end) goes back to waiting; the slot of generation g is
    // rewritten by generation g + 2 at the earliest, and only after g + 1 was published, so "still generation g" after reading
    // the slot means the slot is the one the ticket belongs to.
    void work() {
        for (;;) {
            const uint64_t v = ticket_.fetch_add(1, std::memory_order_acq_rel);
            const uint64_t g = v >> 32;
            const unsigned idx = (unsigned)(v & 0xffffffffu);
            // begin() publishes the ticket word (g << 32) BEFORE generation_ = g: a worker still in this loop from batch g - 1 can
            // draw index 0 of batch g between the two stores.  Returning then would lose that task for good (pending_ never
            // reaches 0, end() spins forever -- ADVICE r2, reproduced under CPU contention): wait for generation_ to catch up
            // with the ticket, and give up only if it has moved PAST it.
            uint64_t cur = generation_.load(std::memory_order_acquire);
            while (cur < g) { __builtin_ia32_pause(); cur = generation_.load(std::memory_order_acquire); }
            if (cur != g) return;
            const unsigned n = slot_n_[g & 1].load(std::memory_order_relaxed);
            if (idx >= n || g != generation_.load(std::memory_order_acquire)) return;
            slot_fn_[g & 1](idx);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(mu_); done_.notify_all(); }
        }
    }
    void loop() {
        // glibc gives a thread its malloc arena at its FIRST allocation (an mmap + mprotect under a process-wide lock): the first batch in which
        // the workers allocate -- the classification of the first append of a process -- paid ~1.4 ms for sixty of those.  Here, once, off every path.
        { void* warm = std::malloc(1 << 16); if (warm) { std::memset(warm, 0, 1 << 16); std::free(warm); } }
        uint64_t seen = 0;
        for (;;) {
            // A call runs several batches a few hundred microseconds apart: poll for the next one for a while before blocking
            // (waking 60 sleeping threads through a condition variable was ~0.2 ms per batch, more than the batches themselves)
            const auto t0 = std::chrono::steady_clock::now();
            bool got = false;
            for (unsigned spin = 0; !got; ++spin) {
                if (generation_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_relaxed)) { got = true; break; }
                if ((spin & 127) == 127 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1500)) break;
                __builtin_ia32_pause();
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(mu_);
                ++sleepers_;
                cv_.wait(lk, [&] { return stop_.load() || generation_.load() != seen; });
                --sleepers_;
            }
            if (stop_.load()) return;
            seen = generation_.load(std::memory_order_acquire);
            work();
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    std::function<void(unsigned)> slot_fn_[2];
    std::atomic<unsigned> slot_n_[2] = { {0}, {0} };
    std::atomic<uint64_t> ticket_{0}, generation_{0};
    std::atomic<unsigned> pending_{0};
    std::atomic<bool> stop_{false};
    unsigned sleepers_ = 0;
};

template <typename F>
void parallelRanges(size_t n, size_t grain, F fn) {
    WorkerPool& pool = WorkerPool::instance();
    const unsigned n_tasks = n >= grain ? pool.size() : 1u;
    pool.run(n_tasks, [&](unsigned t) { fn(n * t / n_tasks, n * (t + 1) / n_tasks); });
}

// ---- resident-problem cache (see the header comment) ----
// A flat array that is neither zero-filled when it grows nor given back when it shrinks (std::vector::resize() value-initialises:
// 24 MB of zeros per call at BASELINE config 3, written once more right after), with half as much headroom again so that a
// reconstruction that grows view by view reallocates a handful of times, not every call.
template <typename T>
struct FlatBuffer {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    FlatBuffer() = default;
    FlatBuffer(const FlatBuffer&) = delete;
    FlatBuffer& operator=(const FlatBuffer&) = delete;
    ~FlatBuffer() { std::free(p); }
    void resize(size_t m) {
        if (m > cap) {
            const size_t want = m + m / 2 + 64;
            T* q = static_cast<T*>(std::realloc(p, want * sizeof(T)));
            if (!q) throw std::bad_alloc();
            p = q; cap = want;
        }
        n = m;
    }
    void clear() { n = 0; }
    // capacity without contents, and share `part` of `parts` of its pages written once (first touch = the page fault), for pre-faulting off the critical path
    void reserve(size_t m) { const size_t keep = n; if (m > cap) { resize(m); n = keep; } }
    void touch(unsigned part, unsigned parts) {
        const size_t bytes = cap * sizeof(T), pages = (bytes + 4095) / 4096;
        volatile char* b = reinterpret_cast<volatile char*>(p);
        for (size_t g = pages * part / parts; g < pages * (part + 1) / parts; ++g) b[g * 4096] = b[g * 4096];
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    const T* begin() const { return p; }
    const T* end() const { return p + n; }
    void swap(FlatBuffer& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
};

// the marshalled problem of one call: flat arrays in the reference's residual-block order
struct Marshalled {
    FlatBuffer<size_t> first;             // CSR over points of the observation list (point-major, ascending view)
    FlatBuffer<int32_t> obs_cam, obs_pt;
    FlatBuffer<double> obs_xy, pt3;
    void swap(Marshalled& o) { first.swap(o.first); obs_cam.swap(o.obs_cam); obs_pt.swap(o.obs_pt); obs_xy.swap(o.obs_xy); pt3.swap(o.pt3); }
    // room for what `o` holds, every page touched by share `part` of `parts` (see ShimCache: the FIRST append of a process used to pay ~2.6 ms of
    // page faults in the set of buffers it was the first to write)
    void reserve_like(const Marshalled& o) { first.reserve(o.first.cap); obs_cam.reserve(o.obs_cam.cap); obs_pt.reserve(o.obs_pt.cap); obs_xy.reserve(o.obs_xy.cap); pt3.reserve(o.pt3.cap); }
    void touch(unsigned part, unsigned parts) { first.touch(part, parts); obs_cam.touch(part, parts); obs_pt.touch(part, parts); obs_xy.touch(part, parts); pt3.touch(part, parts); }
};
struct ShimCache {
    std::mutex mu;
    sfmba_problem* problem = nullptr;
    int n_cam = 0, n_pt = 0, precision = -1;
    // Two marshalling buffers that trade places every call: `prev` is what the resident problem was built from, `cur` is filled by
    // the running call.  Keeping them alive matters as much as the cache itself: 24 MB of fresh vectors per call are ~6000 page
    // faults going in and an munmap going out (1.5 of the 2.3 ms the marshalling took at BASELINE config 3).
    Marshalled prev, cur;
    std::vector<int32_t> new_cam, new_pt;
    std::vector<double> new_xy;
    std::vector<int> grown_pts;           // points whose list is longer than in the previous call
    std::vector<char> is_grown;
    ~ShimCache() { if (problem) sfmba_problem_destroy(problem); }
    void drop() { if (problem) sfmba_problem_destroy(problem); problem = nullptr; prev.first.clear(); prev.obs_cam.clear(); prev.obs_xy.clear(); }
};
ShimCache g_cache;

// Is the previous observation list contained in the new one (same (point, view) entries with the same coordinates)?  If so,
// collects the observations that are new.  Both lists are point-major with ascending view inside a point.
bool diffObservations(const ShimCache& c, int n_pt, const Marshalled& m, std::vector<int32_t>* new_cam,
                      std::vector<int32_t>* new_pt, std::vector<double>* new_xy) {
    new_cam->clear(); new_pt->clear(); new_xy->clear();
    if (n_pt < c.n_pt) return false;
    const FlatBuffer<size_t>& first = m.first;
    const FlatBuffer<int32_t>& obs_cam = m.obs_cam;
    const FlatBuffer<double>& obs_xy = m.obs_xy;
    const size_t np = (size_t)c.n_pt;
    WorkerPool& pool = WorkerPool::instance();
    const unsigned n_thr = np >= 50000 ? pool.size() : 1u;
    std::vector<std::vector<size_t>> added(n_thr);
    std::vector<char> ok(n_thr, 1);
    pool.run(n_thr, [&](unsigned t) {
        for (size_t i = np * t / n_thr; i < np * (t + 1) / n_thr && ok[t]; ++i) {
            size_t a = c.prev.first[i], a1 = c.prev.first[i + 1];
            for (size_t b = first[i]; b < first[i + 1]; ++b) {
                if (a < a1 && c.prev.obs_cam[a] == obs_cam[b]) {
                    if (c.prev.obs_xy[2 * a] != obs_xy[2 * b] || c.prev.obs_xy[2 * a + 1] != obs_xy[2 * b + 1]) { ok[t] = 0; break; }
                    ++a;
                } else if (a < a1 && c.prev.obs_cam[a] < obs_cam[b]) { ok[t] = 0; break; }      // an old observation disappeared
                else added[t].push_back(b);
            }
            if (a != a1) ok[t] = 0;
        }
    });
    for (unsigned t = 0; t < n_thr; ++t) if (!ok[t]) return false;
    auto take = [&](size_t b) { new_cam->push_back(obs_cam[b]); new_pt->push_back(m.obs_pt[b]); new_xy->push_back(obs_xy[2 * b]); new_xy->push_back(obs_xy[2 * b + 1]); };
    for (unsigned t = 0; t < n_thr; ++t) for (size_t b : added[t]) take(b);
    for (size_t b = first[np]; b < first[(size_t)n_pt]; ++b) take(b);
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
    WorkerPool::instance().wake();
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

    // ---- options (BA.cpp:171-177) ----
    sfmba_options opt;
    sfmba_options_default(&opt);
    // The reference solves the reduced camera system exactly (DENSE_SCHUR, BA.cpp:172).  The library default (SFMBA_LINEAR_AUTO)
    // delivers that result with the cheapest solver: Cholesky up to 256 reduced unknowns; above, the two-level CG run to a relative
    // residual of 1e-12 (step within ~1e-10 of the factorised one: far below the float containers this function writes back to),
    // with the Cholesky as fallback on the same linearisation if the CG does not get there.  SFMBA_LINEAR=cholesky forces the
    // factorisation, =pcg the inexact CG (tolerance 1e-8: poses and points agree with the exact solve to ~1e-8, the cost to 1e-12).
    if (const char* e = std::getenv("SFMBA_LINEAR"))
        opt.linear_solver = std::strcmp(e, "pcg") == 0 ? SFMBA_LINEAR_PCG : std::strcmp(e, "auto") == 0 ? SFMBA_LINEAR_AUTO : SFMBA_LINEAR_CHOLESKY;
    if (const char* e = std::getenv("SFMBA_PCG_TOL")) { const double t = std::atof(e); if (t > 0.0 && t < 1.0) opt.pcg_tolerance = t; }   // SFMBA_LINEAR=pcg only
    if (const char* e = std::getenv("SFMBA_PRECISION")) opt.precision = std::strcmp(e, "f32j") == 0 ? SFMBA_PRECISION_F32J : SFMBA_PRECISION_F64;
    if (const char* e = std::getenv("SFMBA_MAX_SECONDS")) opt.max_seconds = std::atof(e);
    if (const char* e = std::getenv("SFMBA_VERBOSE")) opt.verbose = std::atoi(e);

    // Residual blocks in the reference's order: point-major, ascending view inside a point (std::map iteration,
    // BA.cpp:142-166).  The walk over 10^5..10^6 map nodes is the dominant host cost of the call at BASELINE config 3,
    // so it is split over a few threads: offsets first (map::size() is O(1)), then every thread fills its own range.
    const int n_pt = (int)pointCloud.size();
    const char* cache_env = std::getenv("SFMBA_SHIM_CACHE");
    const bool use_cache = !(cache_env && cache_env[0] == '0');
    // the marshalling buffers live in the cache (one call at a time goes through it); without the cache they are local
    std::unique_lock<std::mutex> cache_lock(g_cache.mu, std::defer_lock);
    if (use_cache) cache_lock.lock();
    Marshalled local;
    Marshalled& m = use_cache ? g_cache.cur : local;
    FlatBuffer<double>& pt3 = m.pt3;
    FlatBuffer<size_t>& first = m.first;
    FlatBuffer<int32_t>& obs_cam = m.obs_cam;
    FlatBuffer<int32_t>& obs_pt = m.obs_pt;
    FlatBuffer<double>& obs_xy = m.obs_xy;
    pt3.resize((size_t)3 * n_pt);
    first.resize((size_t)n_pt + 1);
    first[0] = 0;
    {
        // list lengths (std::map::size() is O(1)) and their prefix sum, in two pooled passes over ranges of points
        WorkerPool& sizes_pool = WorkerPool::instance();
        const unsigned n_rng = n_pt >= 20000 ? sizes_pool.size() : 1u;
        std::vector<size_t> rng_sum(n_rng + 1, 0);
        sizes_pool.run(n_rng, [&](unsigned t) {
            size_t sum = 0;
            for (size_t i = (size_t)n_pt * t / n_rng; i < (size_t)n_pt * (t + 1) / n_rng; i++) { const size_t k = pointCloud[i].originatingViews.size(); first[i + 1] = k; sum += k; }
            rng_sum[t + 1] = sum;
        });
        for (unsigned t = 0; t < n_rng; ++t) rng_sum[t + 1] += rng_sum[t];
        sizes_pool.run(n_rng, [&](unsigned t) {
            size_t run = rng_sum[t];
            for (size_t i = (size_t)n_pt * t / n_rng; i < (size_t)n_pt * (t + 1) / n_rng; i++) { run += first[i + 1]; first[i + 1] = run; }
        });
    }
    const size_t n_obs = first[n_pt];
    obs_cam.resize(n_obs); obs_pt.resize(n_obs);
    obs_xy.resize(2 * n_obs);
    const double t_sizes = now();
    auto fill_points = [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) { const cv::Point3f& q = pointCloud[i].p; pt3[3 * i] = q.x; pt3[3 * i + 1] = q.y; pt3[3 * i + 2] = q.z; }
    };
    auto fill_lists = [&](int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            size_t k = first[i];
            for (const auto& kv : pointCloud[i].originatingViews) {
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
    WorkerPool& pool = WorkerPool::instance();
    const unsigned n_shares = n_obs >= 100000 ? 2 * pool.size() : 1u;      // equal shares of observations, not of points
    std::vector<int> cut(n_shares + 1, 0);
    for (unsigned t = 1; t < n_shares; ++t) {
        const size_t want = n_obs * t / n_shares;
        cut[t] = std::max(cut[t - 1], (int)(std::upper_bound(first.begin(), first.end(), want) - first.begin() - 1));
    }
    cut[n_shares] = n_pt;

    // Overlapped marshalling.  Walking 10^6 map nodes (cache misses) and comparing them with the previous call's list is most of
    // the host time of a call that only adds a view, and the GPU work of that call does not need it: std::map::size() is O(1), so
    // the points whose lists GREW (and the new points) are known at once, their new observations are all the device needs, and
    // the walk over the lists that kept their length -- which must still be compared, entry by entry, with what the resident
    // problem was built from -- runs on the worker threads WHILE the GPU solves.  If that comparison fails (a caller that replaced
    // an observation without changing the count), the result is thrown away and the call starts over on the rebuild path: the
    // outcome never depends on the overlap, only the time does.  SFMBA_SHIM_OVERLAP=0 switches it off.
    ShimCache& c = g_cache;
    const char* overlap_env = std::getenv("SFMBA_SHIM_OVERLAP");
    bool overlap = use_cache && c.problem && c.precision == opt.precision && c.n_cam == n_cam && c.n_pt > 0 && n_pt >= c.n_pt &&
                   c.prev.first.size() == (size_t)c.n_pt + 1 && !std::getenv("SFMBA_DUMP") && !(overlap_env && overlap_env[0] == '0');
    std::vector<int>& grown_pts = c.grown_pts;
    grown_pts.clear();
    if (overlap) {
        // pooled over ranges of points; the per-range lists are concatenated in order
        WorkerPool& cls_pool = WorkerPool::instance();
        const unsigned n_rng = c.n_pt >= 20000 ? cls_pool.size() : 1u;
        std::vector<std::vector<int>> grown_rng(n_rng);
        std::vector<char> shrunk(n_rng, 0);
        cls_pool.run(n_rng, [&](unsigned t) {
            for (size_t i = (size_t)c.n_pt * t / n_rng; i < (size_t)c.n_pt * (t + 1) / n_rng; i++) {
                const size_t k_new = first[i + 1] - first[i], k_old = c.prev.first[i + 1] - c.prev.first[i];
                if (k_new < k_old) { shrunk[t] = 1; break; }
                if (k_new != k_old) grown_rng[t].push_back((int)i);
            }
        });
        for (unsigned t = 0; t < n_rng; ++t) {
            if (shrunk[t]) overlap = false;
            grown_pts.insert(grown_pts.end(), grown_rng[t].begin(), grown_rng[t].end());
        }
    }
    std::atomic<bool> lists_differ(false);
    bool walk_pending = false;
    const double t_classified = now();
    double t_points = t_classified, t_grown = t_classified;
    if (overlap) {
        parallelRanges((size_t)n_pt, 20000, fill_points);
        t_points = now();
        // lists of the grown and the new points now (pooled: every map node is a cache miss); what they add to the previous list
        // is what the device gets
        parallelRanges(grown_pts.size(), 256, [&](size_t g0, size_t g1) { for (size_t g = g0; g < g1; ++g) fill_lists(grown_pts[g], grown_pts[g] + 1); });
        parallelRanges((size_t)(n_pt - c.n_pt), 256, [&](size_t i0, size_t i1) { fill_lists(c.n_pt + (int)i0, c.n_pt + (int)i1); });
        t_grown = now();
        c.new_cam.clear(); c.new_pt.clear(); c.new_xy.clear();
        auto take = [&](size_t b) { c.new_cam.push_back(obs_cam[b]); c.new_pt.push_back(obs_pt[b]); c.new_xy.push_back(obs_xy[2 * b]); c.new_xy.push_back(obs_xy[2 * b + 1]); };
        // which entries of the grown lists are new (pooled over ranges of grown points, gathered in order afterwards)
        const unsigned n_parts = grown_pts.size() >= 1024 ? pool.size() : 1u;
        std::vector<std::vector<size_t>> added(n_parts);
        std::vector<char> part_ok(n_parts, 1);
        pool.run(n_parts, [&](unsigned t) {
            for (size_t g = grown_pts.size() * t / n_parts; g < grown_pts.size() * (t + 1) / n_parts && part_ok[t]; ++g) {
                const int i = grown_pts[g];
                size_t a = c.prev.first[i];
                const size_t a1 = c.prev.first[(size_t)i + 1];
                for (size_t b = first[i]; b < first[(size_t)i + 1]; ++b) {
                    if (a < a1 && c.prev.obs_cam[a] == obs_cam[b]) {
                        if (c.prev.obs_xy[2 * a] != obs_xy[2 * b] || c.prev.obs_xy[2 * a + 1] != obs_xy[2 * b + 1]) { part_ok[t] = 0; break; }
                        ++a;
                    } else if (a < a1 && c.prev.obs_cam[a] < obs_cam[b]) { part_ok[t] = 0; break; }      // an old observation disappeared
                    else added[t].push_back(b);
                }
                if (a != a1) part_ok[t] = 0;
            }
        });
        for (unsigned t = 0; t < n_parts; ++t) overlap = overlap && part_ok[t];
        if (overlap) {
            for (unsigned t = 0; t < n_parts; ++t) for (size_t b : added[t]) take(b);
            for (size_t b = first[(size_t)c.n_pt]; b < first[(size_t)n_pt]; ++b) take(b);
        }
    }
    if (overlap) {
        // the lists that kept their length: walked and compared on the worker threads from here on
        c.is_grown.assign((size_t)c.n_pt, 0);
        for (int i : grown_pts) c.is_grown[(size_t)i] = 1;
        const int n_old_pt = c.n_pt;
        pool.begin(n_shares, [&, n_old_pt](unsigned t) {
            const int i1 = std::min(cut[t + 1], n_old_pt);
            for (int i = cut[t]; i < i1; i++) {          // (no early exit: the rebuild path needs every list)
                if (c.is_grown[(size_t)i]) continue;
                fill_lists(i, i + 1);
                const size_t a0 = c.prev.first[i], b0 = first[i], k = first[(size_t)i + 1] - b0;
                if (std::memcmp(&c.prev.obs_cam[a0], &obs_cam[b0], sizeof(int32_t) * k) != 0 ||
                    std::memcmp(&c.prev.obs_xy[2 * a0], &obs_xy[2 * b0], sizeof(double) * 2 * k) != 0) lists_differ.store(true);
            }
        });
        walk_pending = true;
    } else {
        pool.run(n_shares, [&](unsigned t) { fill_points((size_t)cut[t], (size_t)cut[t + 1]); fill_lists(cut[t], cut[t + 1]); });
    }
    if (const char* e = std::getenv("SFMBA_DUMP")) dumpProblem(e, n_cam, cam6, n_pt, pt3, obs_cam, obs_pt, obs_xy, focal);

    // ---- solve on the MI355X (replaces ceres::Solve, BA.cpp:179) ----
    const double t_marshalled = now();
    sfmba_summary summary;
    std::memset(&summary, 0, sizeof(summary));
    int rc;
    const char* how = "one-shot";
    double t_diff = t_marshalled, t_setup = t_marshalled, t_lm = t_marshalled;
    if (!use_cache) {
        rc = sfmba_solve(n_cam, cam6.data(), n_pt, pt3.data(), (int64_t)obs_cam.size(), obs_cam.data(), obs_pt.data(),
                         obs_xy.data(), &focal, &opt, &summary, nullptr, 0, nullptr);
    } else {
        bool grown = overlap || (c.problem && c.precision == opt.precision && c.n_cam == n_cam &&
                                 diffObservations(c, n_pt, m, &c.new_cam, &c.new_pt, &c.new_xy));
        t_diff = now();
        rc = SFMBA_OK;
        if (grown) {
            how = c.new_cam.empty() ? "resident" : "append";
            if (c.new_cam.empty() && n_pt == c.n_pt) rc = sfmba_problem_set_params(c.problem, cam6.data(), pt3.data(), focal);
            else rc = sfmba_problem_append(c.problem, n_cam, cam6.data(), n_pt, pt3.data(), (int64_t)c.new_cam.size(), c.new_cam.data(), c.new_pt.data(),
                                           c.new_xy.data(), focal);
            if (rc != SFMBA_OK) {
                if (walk_pending) { pool.end(); walk_pending = false; }       // (it reads the lists drop() clears)
                c.drop(); grown = false;
            }
        }
        if (!grown) {
            if (walk_pending) { pool.end(); walk_pending = false; }       // (the lists are complete after this)
            how = "rebuild";
            c.drop();
            rc = sfmba_problem_create(0, opt.precision, n_cam, cam6.data(), n_pt, pt3.data(), (int64_t)obs_cam.size(), obs_cam.data(), obs_pt.data(),
                                      obs_xy.data(), focal, &c.problem);
        }
        t_setup = now();
        // A rebuild is (also) the first call of a process: the OTHER set of marshalling buffers -- the one the next call fills -- has never been
        // written.  It gets its capacity and its page faults on the worker threads while the GPU solves (the first append of a
        // process paid 1.98 + 0.63 ms for them on its critical path).
        bool prefault_pending = false;
        if (rc == SFMBA_OK && !grown && !walk_pending) {
            c.prev.reserve_like(c.cur);
            const unsigned parts = pool.size();
            pool.begin(parts, [&c, parts](unsigned t) { c.prev.touch(t, parts); });
            prefault_pending = true;
        }
        if (rc == SFMBA_OK) rc = sfmba_problem_solve(c.problem, &opt, &summary, nullptr, 0, nullptr);
        if (prefault_pending) pool.end();
        t_lm = now();
        summary.setup_seconds = t_setup - t_diff;
        // Ceres leaves the parameter blocks alone on FAILURE; every other termination hands back the best point
        if (walk_pending) {
            pool.end();
            walk_pending = false;
            if (lists_differ.load()) {
                // a list changed behind an unchanged length: what was solved is not this call's problem.  Start over from the complete lists.
                how = "rebuild (a list changed in place)";
                c.drop();
                focal = intrinsics.K.at<float>(0, 0);
                fill_points(0, (size_t)n_pt);
                rc = sfmba_problem_create(0, opt.precision, n_cam, cam6.data(), n_pt, pt3.data(), (int64_t)obs_cam.size(), obs_cam.data(), obs_pt.data(),
                                          obs_xy.data(), focal, &c.problem);
                if (rc == SFMBA_OK) rc = sfmba_problem_solve(c.problem, &opt, &summary, nullptr, 0, nullptr);
            }
        }
        if (rc == SFMBA_OK && summary.termination != SFMBA_FAILURE) rc = sfmba_problem_get_params(c.problem, cam6.data(), pt3.data(), &focal);
        if (rc == SFMBA_OK) {
            c.n_cam = n_cam; c.n_pt = n_pt; c.precision = opt.precision;
            c.prev.swap(c.cur);          // this call's lists are what the resident problem now holds (pt3: the solved points)
        } else c.drop();
    }
    const double t_solved = now();
    if (timing) std::fprintf(stderr, "[sfmba shim] path: %s\n", how);
    if (timing) std::fprintf(stderr, "[sfmba shim] marshal split: poses+sizes %.2f, classify %.2f, points %.2f, grown lists %.2f, new obs + hand-over %.2f ms\n",
                             1e3 * (t_sizes - t_begin), 1e3 * (t_classified - t_sizes), 1e3 * (t_points - t_classified), 1e3 * (t_grown - t_points), 1e3 * (t_marshalled - t_grown));
    struct ExitTimer { bool on; double t0; decltype(now)& clk; ~ExitTimer() { if (on) std::fprintf(stderr, "[sfmba shim] write back %.2f ms\n", 1e3 * (clk() - t0)); } } exit_timer{ timing, t_solved, now };
    if (timing)
        std::fprintf(stderr, "[sfmba shim] marshal %.2f ms, sfmba_solve %.2f ms (setup %.2f + LM %.2f); diff %.2f, build/append %.2f, solve call %.2f, read back %.2f\n",
                     1e3 * (t_marshalled - t_begin), 1e3 * (t_solved - t_marshalled), 1e3 * summary.setup_seconds, 1e3 * summary.seconds,
                     1e3 * (t_diff - t_marshalled), 1e3 * (t_setup - t_diff), 1e3 * (t_lm - t_setup), 1e3 * (t_solved - t_lm));
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
    const FlatBuffer<double>& pt3_out = use_cache ? g_cache.prev.pt3 : pt3;        // (the buffers traded places after the solve)
    parallelRanges((size_t)n_pt, 20000, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            pointCloud[i].p.x = (float)pt3_out[3 * i];
            pointCloud[i].p.y = (float)pt3_out[3 * i + 1];
            pointCloud[i].p.z = (float)pt3_out[3 * i + 2];
        }
    });
}

struct WorkerPoolAccess {
    static int selftest(int batches) {
        WorkerPool& pool = WorkerPool::instance();
        int wrong = 0;
        for (int b = 0; b < batches; ++b) {
            const unsigned n = 1u + (unsigned)((b * 2654435761u) % 517u);
            std::atomic<unsigned long long> sum(0);
            std::atomic<unsigned> calls(0);
            auto task = [&](unsigned t) { sum.fetch_add(t + 1ull, std::memory_order_relaxed); calls.fetch_add(1, std::memory_order_relaxed); };
            if (b % 3 == 0) { pool.begin(n, task); volatile unsigned spin = 0; for (unsigned k = 0; k < (unsigned)(b % 7) * 100u; ++k) spin = spin + 1; pool.end(); }
            else if (b % 3 == 1) pool.run(n, task);
            else { pool.wake(); pool.run(n, task); }
            if (sum.load() != (unsigned long long)n * (n + 1) / 2 || calls.load() != n) ++wrong;
        }
        return wrong;
    }
    // many SMALL batches back to back (the regime where a worker of batch g - 1 is still drawing tickets when batch g is published)
    static int stress(int batches, int max_tasks) {
        WorkerPool& pool = WorkerPool::instance();
        int wrong = 0;
        std::atomic<unsigned> calls(0);
        for (int b = 0; b < batches; ++b) {
            const unsigned n = 2u + (unsigned)((b * 2654435761u) % (unsigned)(max_tasks > 1 ? max_tasks - 1 : 1));
            calls.store(0, std::memory_order_relaxed);
            pool.run(n, [&](unsigned) { calls.fetch_add(1, std::memory_order_relaxed); });
            if (calls.load() != n) ++wrong;
        }
        return wrong;
    }
};

} /* namespace sfmtoylib */

// Self-test of the worker pool (tests/test_shim_pool_cpu.py; no GPU involved): `batches` batches of varying size, run() and the
// begin() / end() pair alternating, every task adds its index + 1 to a sum.  Returns the number of batches whose sum was wrong.
extern "C" __attribute__((visibility("default"))) int sfmba_shim_pool_selftest(int batches) {
    using sfmtoylib::WorkerPoolAccess;
    return WorkerPoolAccess::selftest(batches);
}
// Contended variant (tests/test_shim_pool_cpu.py runs several PROCESSES of it at once under a watchdog): small batches only.
extern "C" __attribute__((visibility("default"))) int sfmba_shim_pool_stress(int batches, int max_tasks) {
    return sfmtoylib::WorkerPoolAccess::stress(batches, max_tasks);
}

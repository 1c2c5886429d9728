// sfmba_api.hip -- C ABI of include/sfmba.h: problem structure build, HBM residency, LM driver.
//
// Host-side counterpart of what adjustBundle() does between BA.cpp:142 (AddResidualBlock loop) and
// BA.cpp:179 (ceres::Solve): the observation list is regrouped once per problem (point-major CSR
// for the point pass, camera-major CSR for the reduced-system pass), uploaded, and every LM
// iteration is then a fixed sequence of kernels on one private HIP stream.  The accept/reject
// logic runs on the device (k_lm_control); the host only reads the small LMState back once per
// iteration to learn whether to stop.
//
// There is NO CPU fallback in this file: without a HIP device every entry point returns
// SFMBA_ERR_NO_DEVICE.
#include "../../include/sfmba.h"
#include "association.h"
#include "ba_kernels.h"
#include "dense_solver.h"
#include "dist_cg.h"
#include "device_arena.h"
#include "profiler.h"
#include "sfmba_device.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <vector>

using namespace sfmba;

namespace {

thread_local std::string g_last_error;

int fail(int rc, const std::string& msg) { g_last_error = msg; return rc; }

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(SFMBA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

const char* message_text(int id) {
    switch (id) {
        case MSG_GRADIENT_TOL: return "Gradient tolerance reached.";
        case MSG_PARAMETER_TOL: return "Parameter tolerance reached.";
        case MSG_FUNCTION_TOL: return "Function tolerance reached.";
        case MSG_MIN_RADIUS: return "Minimum trust region radius reached.";
        case MSG_INVALID_STEPS: return "Number of consecutive invalid steps more than max_num_consecutive_invalid_steps.";
        case MSG_INITIAL_EVAL_FAILED: return "Initial residual and Jacobian evaluation failed.";
        case MSG_EVAL_FAILED: return "Residual and Jacobian evaluation failed.";
        case MSG_MAX_ITERS: return "Maximum number of iterations reached.";
        case MSG_MAX_TIME: return "Maximum solver time reached.";
        default: return "";
    }
}

// Device arrays of a problem come from its arena (set for the duration of create_impl); API-call temporaries from HIP.
thread_local DeviceArena* t_arena = nullptr;
struct ArenaScope { DeviceArena* prev; explicit ArenaScope(DeviceArena* a) : prev(t_arena) { t_arena = a; } ~ArenaScope() { t_arena = prev; } };

template <typename T> hipError_t dev_alloc(T** p, size_t n) {
    if (t_arena) { *p = t_arena->alloc_n<T>(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
    return hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (n ? n : 1));
}

template <typename T> hipError_t dev_upload(T** p, const std::vector<T>& v) {
    hipError_t e = dev_alloc(p, v.size());
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpy(*p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice);
    return e;
}

// Host loops over the observation list of a large problem, split over a few threads (structure build of the one-shot call).
constexpr int PARALLEL_FOR_MIN = 200000;
template <typename F>
void parallel_for(int n, F fn) {
    unsigned nt = n >= PARALLEL_FOR_MIN ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (nt <= 1) { fn(0, n); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) pool.emplace_back(fn, (int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt));
    for (auto& th : pool) th.join();
}

// Host side of the one meeting point per LM iteration: wait until the device has posted sequence number `want` to a
// host-mapped mailbox word.  Three phases so that a solve does not pin a core: a short pause-spin (the common case: the
// post arrives within microseconds of the host getting here), then yielding, then short sleeps.  While waiting, the
// stream is queried now and then: a stream that has drained WITHOUT the post means a launch failed or was lost --
// reported instead of waiting on a wall clock.  Returns 0 = posted, 1 = stream idle and nothing posted, 2 = HIP error.
int wait_mailbox(volatile int* word, int want, hipStream_t stream) {
    const double t0 = now_seconds();
    unsigned spins = 0;
    double next_query = 2e-3;
    for (;;) {
        if (*word >= want) { __sync_synchronize(); return 0; }
        ++spins;
        if ((spins & 63u) != 0u) { __builtin_ia32_pause(); continue; }
        const double waited = now_seconds() - t0;
        if (waited < 30e-6) continue;
        if (waited >= next_query) {
            next_query = waited * 1.5;
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {
                __sync_synchronize();
                return *word >= want ? 0 : 1;
            }
            if (q != hipErrorNotReady) return 2;
        }
        if (waited < 1e-3) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// relative residual AUTO runs the CG to (see run_solve)
constexpr double auto_cg_tol() { return 1e-12; }
// A behaviour switch of sfmba_options: the field (1 on, -1 off), otherwise the library default.  (ABI v4 let an environment variable of the
// name given here override the field; since ABI v5 nothing below sfmba_problem_create* reads the environment: the name documents the switch.)
bool option_switch(int field, const char* /*name*/, bool dflt) { return field > 0 ? true : field < 0 ? false : dflt; }

// ---- roctx ranges around the phases of an LM iteration: `rocprofv3 --marker-trace` then shows linearise / reduce / solve / update
// per iteration on the host timeline next to the kernels they enqueue.  Bound at run time like RCCL (no link-time dependency); switched on when a
// problem is BUILT with SFMBA_ROCTX=1 in the environment (nothing below sfmba_problem_create* reads the environment). ----
struct RoctxApi { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const RoctxApi* roctx_api() {
    static RoctxApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" }) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                api.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                api.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (api.push && api.pop) break;
                api.push = nullptr; api.pop = nullptr;
            }
        }
    });
    return api.push ? &api : nullptr;
}
// The back-substitution is ONE step in two launches that share a convention: in F32J k_cam_update writes the fp32 camera records (pu32) AND adds
// the gradient . step term of the model cost change, and k_point_update<float> then forms no residual term -- pairing a cam_update that saw pu32 with a
// point_update that did not (or the other way round) would count that term twice or not at all and skew the step quality rho.  Every LM loop goes through
// this helper: both kernels see the SAME DeviceBuffers, and an fp64 point pass can never meet a pu32 camera pass.  ds_points: the point pass's structure
// (a row-sharded rank passes the view with its own point range).
template <typename T>
void launch_back_substitution(hipStream_t s, const DeviceStructure& ds, const DeviceStructure& ds_points, const DeviceBuffers& dbu, Profiler* prof) {
    DeviceBuffers b = dbu;
    if (sizeof(T) == 8) b.pu32 = nullptr;
    { ProfScope ps(prof, KID_CAM_UPDATE, s); launch_cam_update(s, ds, b); }
    { ProfScope ps(prof, KID_POINT_UPDATE, s); launch_point_update<T>(s, ds_points, b); }
}

struct RoctxRange {
    const RoctxApi* a;
    RoctxRange(bool on, const char* name) : a(on ? roctx_api() : nullptr) { if (a) a->push(name); }
    void end() { if (a) { a->pop(); a = nullptr; } }
    ~RoctxRange() { end(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

__global__ void k_fill(double* p, size_t n, double v) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) p[e] = v;
}

}  // namespace

// One resident helper thread per process: the host half of a structure build (descriptor loops, uploads) runs on it while the calling
// thread enqueues the device half.  A build takes the helper for its duration (Lease); a second build at the same time does both
// halves itself.  Never destroyed: the thread outlives static destruction.
class BuildHelper {
public:
    static BuildHelper& instance() { static BuildHelper* h = new BuildHelper(); return *h; }
    class Lease {
    public:
        explicit Lease(BuildHelper& h) : h_(h), granted_(h.alive_ && h.busy_.try_lock()) {}
        ~Lease() { if (granted_) { wait(); h_.busy_.unlock(); } }
        bool granted() const { return granted_; }
        void post(std::function<void()> fn) {
            h_.fn_ = std::move(fn);
            pending_ = true;
            h_.done_.store(false, std::memory_order_relaxed);
            h_.ready_.store(true, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(h_.mu_); h_.wake_ = true; }
            h_.cv_.notify_one();
        }
        void wait() {
            if (!pending_) return;
            while (!h_.done_.load(std::memory_order_acquire)) __builtin_ia32_pause();
            pending_ = false;
        }
    private:
        BuildHelper& h_;
        bool granted_, pending_ = false;
    };
private:
    BuildHelper() {
        try { std::thread([this] { loop(); }).detach(); alive_ = true; }
        catch (...) { alive_ = false; }          // no thread to be had: every build does both halves itself
    }
    void loop() {
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return wake_; }); wake_ = false; }
            if (ready_.load(std::memory_order_acquire)) {
                ready_.store(false, std::memory_order_relaxed);
                fn_();
                fn_ = nullptr;
                done_.store(true, std::memory_order_release);
            }
        }
    }
    std::mutex busy_, mu_;
    std::condition_variable cv_;
    bool wake_ = false, alive_ = false;
    std::atomic<bool> ready_{ false }, done_{ true };
    std::function<void()> fn_;
};

struct sfmba_problem {
    int device = 0;
    int precision = SFMBA_PRECISION_F64;
    hipStream_t stream = nullptr;
    int n_cam_full = 0, n_pt_full = 0;
    int64_t n_obs = 0;
    std::vector<int> acam_id, apt_id;     // active slot -> caller index
    std::vector<int> h_pt_cnt, h_cam_cnt; // observations per point / camera slot (host mirror: CSR pointers without a device round trip)
    std::vector<int> cam_slot, pt_slot;   // caller index -> slot (-1: not observed)
    bool sharded = false;
    DeviceStructure ds = {};
    DeviceBuffers db = {};
    DenseSolver solver;
    DeviceArena arena;                    // every device array below except db.trace
    HostKit kit;                          // stream + pinned block (recycled)
    // owned device arrays behind ds
    int *d_pt_ptr = nullptr, *d_obs_cam = nullptr, *d_cam_ptr = nullptr, *d_cam_obs = nullptr, *d_cam_obs_pt = nullptr;   // (all in `arena`)
    int *d_obs_pt = nullptr, *d_perm = nullptr;   // contiguous [2*nobs]: point slot, perm
    void* d_obs_xy = nullptr;
    int4* d_chunks = nullptr, *d_chunks_coarse = nullptr, *d_pwg_desc = nullptr;
    float* d_pu32 = nullptr;                // fp32 camera records of the back-substitution's first sweep (F32J, every LM loop; launch_back_substitution)
    int2* d_pwg_chunk = nullptr; int* d_multi_slots = nullptr; int* d_build_counters = nullptr; int* d_pt_order = nullptr;
    int *d_chunk_order = nullptr, *d_coarse_order = nullptr;
    double block_fill = 1.0;              // non-empty off-diagonal blocks of the reduced matrix / all of them
    double block_band = 0.0;              // ... and the share of those that couple cameras within a quarter of the cyclic camera order
    int* d_blk_ptr = nullptr;
    unsigned* d_blk_mask = nullptr;       // per camera: cameras with a non-empty block in common (block-sparse CG product)
    int* d_cam_chunk_ptr = nullptr;
    bool deterministic = false;             // SFMBA_DETERMINISTIC=1 at build time
    bool roctx = false;                     // SFMBA_ROCTX=1 at build time: roctx ranges around the phases of an LM iteration
    bool cam_identity = false, pt_identity = false;   // slot == caller index for every camera / point (arrays copied as they are)
    bool reset_pending = false;             // sfmba_problem_reset() was called: the initial parameters are restored by the next solve's first kernel
                                            // (or by flush_reset() if anything else looks at the problem first)
    int2 *d_blk_cams = nullptr, *d_pwg_blocks = nullptr, *d_dup_blocks = nullptr;
    int* d_pair_pt = nullptr;
    void* d_cam_obs_xy = nullptr;
    double* d_facc = nullptr;
    double *d_cam0 = nullptr, *d_pts0 = nullptr;  // parameters given at create time
    double focal0 = 0.0;
    double *d_sys = nullptr;                      // S | rhs | udiag | bc (contiguous)
    double *d_red = nullptr;                      // sharded mode: packed upper triangle of S + the same tail (the all-reduce buffer)
    int* d_info = nullptr;
    LMState* h_state = nullptr;                   // pinned
    volatile int* h_lm_mail = nullptr;            // host-mapped mailbox written by k_lm_control
    char* d_pinned = nullptr;                     // device address of kit.pinned
    bool trace_mapped = false;                    // db.trace points into the pinned block
    int cur = 0;                                  // which buffer holds the current parameters
    double focal = 0.0;
    bool empty = false;                           // no observations
    bool poisoned = false;                        // an append failed half way: only sfmba_problem_destroy is valid (include/sfmba.h)
    // sharded-mode state
    sfmba_options shard_opt;
    bool shard_active = false;
    double shard_t0 = 0.0;
    int shard_phase = 0;
    int shard_rank = 0, shard_world = 1;
    double* d_scal = nullptr;                     // tail of d_sys: SFMBA_SHARD_SCALARS doubles
    int shard_host_iter = 0;
    int64_t shard_exchange[4] = { 0, 0, 0, 0 };       // bytes of exchanges (A), (B), (C) per linearisation of the last sharded solve; (B) in fp32?
    sfmba_allreduce_f32_fn allreduce_f32 = nullptr;   // optional: exchange (B) in fp32 where the CG stores S~ in fp32
    sfmba_reduce_scatter_fn reduce_scatter = nullptr; // optional: the distributed CG's exchange (B)
    // row-sharded problem (SFMBA_CREATE_ROW_SHARDED: every rank holds the whole problem; options.shard_distributed_cg = 3)
    // no pair list (SFMBA_CREATE_NO_PAIR_LIST, or more pairs of observations than a list can hold): sfmba_problem_solve runs the CG with the
    // reduced matrix applied implicitly (implicit_schur.hip) -- O(observations) memory whatever the track lengths
    bool no_pairs = false;
    bool row_sharded = false;
    int own_pt0 = 0, own_pt1 = 0, own_pt_stride = 0;  // own range of point slots; slots per rank (the per-point arrays are padded to world * stride)
    int own_chunk0 = 0, own_chunk1 = 0;               // own share of the camera-major chunks (k_cam_diag_f) ...
    int own_coarse0 = 0, own_coarse1 = 0;             // ... and of the coarse ones (column norms)
    sfmba_allgather_fn allgather = nullptr;
    DistCg dcg;                                       // distributed CG workspace (created by the first solve that asks for it)
    double *imp_dtab = nullptr, *imp_spt = nullptr, *imp_acc = nullptr, *imp_part = nullptr;   // implicit Schur product workspace (shard_distributed_cg = 2; allocated by the first solve that asks)
    long long shard_blocks_off = 0;                   // doubles: where the block region of d_red starts (behind the region of exchange (A))
    int dcg_last_f32 = -1;
    sfmba_summary shard_sum;
    Profiler prof;
};

namespace {

int check_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SFMBA_ERR_NO_DEVICE, "no HIP device available: the MI355X back end has no CPU fallback");
    if (device < 0 || device >= n) return fail(SFMBA_ERR_INVALID_ARG, "device index out of range");
    return SFMBA_OK;
}

// What the passes of a ROW-SHARDED rank see (include/sfmba.h, SFMBA_CREATE_ROW_SHARDED): the point passes its own points (pt_order lists
// them), the camera-major passes its share of the chunks; everything else the whole problem.
DeviceStructure ds_points(const sfmba_problem* p) {
    DeviceStructure ds = p->ds;
    if (p->row_sharded) { ds.npt = p->own_pt1 - p->own_pt0; ds.pt_base = p->own_pt0; }
    return ds;
}
DeviceStructure ds_cams(const sfmba_problem* p) {
    DeviceStructure ds = p->ds;
    if (p->row_sharded) {
        // (a contiguous share of the LAUNCH order: the rank's workgroups stay inside one window of the point table at a time)
        ds.chunk_order += p->own_chunk0; ds.nchunk = p->own_chunk1 - p->own_chunk0;
        ds.coarse_order += p->own_coarse0; ds.nchunk_coarse = p->own_coarse1 - p->own_coarse0;
    }
    return ds;
}
DeviceBuffers db_cams(const sfmba_problem* p) {
    return p->db;      // (deterministic mode: a chunk's slot in cd_part is its index in the chunk LIST, whatever the launch order; the slots of the others' chunks stay zero)
}

void init_state(sfmba_problem* p, LMState& st, const sfmba_options& o) {
    std::memset(&st, 0, sizeof(st));
    st.cur = p->cur;
    st.iter = 0;
    st.termination = -1;
    st.x_is_new = 1;
    st.radius = o.initial_radius;
    st.decrease_factor = 2.0;
    st.focal[p->cur] = p->focal;
    st.focal[p->cur ^ 1] = p->focal;
    st.fscale = 1.0;
    st.function_tolerance = o.function_tolerance;
    st.gradient_tolerance = o.gradient_tolerance;
    st.parameter_tolerance = o.parameter_tolerance;
    st.max_radius = o.max_radius;
    st.min_radius = o.min_radius;
    st.min_relative_decrease = o.min_relative_decrease;
    st.min_diag = o.min_lm_diagonal;
    st.max_diag = o.max_lm_diagonal;
    st.max_consecutive_invalid = o.max_consecutive_invalid_steps;
    // An invalid step (the linear solver failed) halves the radius [Ceres-upstream: LevenbergMarquardtStrategy::StepIsInvalid].  In fp64 that is all there
    // is to it -- it does not happen on the problems the oracle solves.  With fp32 Jacobians it is what a trust region that has grown past ~1e7 looks
    // like: the damping diag / radius is then below the rounding of the blocks and the reduced matrix is no longer positive definite; five halvings
    // (a factor 32) do not bring it back and the run ends in FAILURE where the reference converges (tests/fuzz_parity.py: a weakly determined problem,
    // radius 8.6e8).  F32J therefore divides by eight: five in a row cover 3e4.  Runs without invalid steps -- every parity fixture -- are untouched.
    st.invalid_shrink = p->precision == SFMBA_PRECISION_F32J ? 0.125 : 0.5;
}

int upload_state(sfmba_problem* p, const LMState& st) {
    *p->h_state = st;
    HIP_TRY(hipMemcpyAsync(p->db.st, p->h_state, sizeof(LMState), hipMemcpyHostToDevice, p->stream));
    return SFMBA_OK;
}

int download_state(sfmba_problem* p) {
    HIP_TRY(hipMemcpyAsync(p->h_state, p->db.st, sizeof(LMState), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return SFMBA_OK;
}

int ensure_trace(sfmba_problem* p, int rows) {
    if (rows <= p->db.trace_cap) return SFMBA_OK;
    if (p->db.trace && !p->trace_mapped) (void)hipFree(p->db.trace);
    p->db.trace = nullptr;
    p->db.trace_cap = 0;
    p->trace_mapped = false;
    if (p->d_pinned && sizeof(TraceRow) * (size_t)rows <= HOSTKIT_PINNED_BYTES - 4096) {
        // the rows live in the handle's host-mapped block: k_lm_control writes them over PCIe, nothing is copied back
        p->db.trace = reinterpret_cast<TraceRow*>(p->d_pinned + 4096);
        p->trace_mapped = true;
    } else {
        HIP_TRY(dev_alloc(&p->db.trace, (size_t)rows));
    }
    p->db.trace_cap = rows;
    return SFMBA_OK;
}

template <typename T>
void launch_linearise_setup(sfmba_problem* p, int jacobi, bool begun = false) {
    if (begun) {
        // inside a solve: k_begin has written the LM state, cleared the accumulators and built the camera tables; the point
        // scales are formed by the first k_point_build; ||x|| is finished by k_colnorm_finish
        // (||x||^2 rides in the camera pass of the column norms where that pass runs: k_xnorm as a launch of its own was 4.6 us per solve)
        const bool fold = jacobi && p->ds.nchunk_coarse > 0;
        if (!fold) launch_xnorm(p->stream, p->ds, p->db);
        launch_colnorm<T>(p->stream, p->ds, p->db, jacobi, /*clear_udiag=*/false, /*points=*/false, /*finish_xnorm=*/true, /*with_xnorm=*/fold);
        return;
    }
    const size_t n = 6 * (size_t)p->ds.ncam;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, p->db.cscale, n, 1.0);
    launch_cam_setup<T>(p->stream, p->ds, p->db, p->cur);
    launch_xnorm(p->stream, p->ds, p->db);
    launch_iter0(p->stream, p->ds, p->db);
    launch_colnorm<T>(p->stream, p->ds, p->db, jacobi);
}

template <typename T>
int run_solve(sfmba_problem* p, const sfmba_options& o, sfmba_summary* summary, sfmba_iteration* trace, int trace_cap, int* trace_len) {
    sfmba_summary sum;
    std::memset(&sum, 0, sizeof(sum));
    if (trace_len) *trace_len = 0;
    if (p->empty) {
        sum.termination = SFMBA_CONVERGENCE;
        std::snprintf(sum.message, sizeof(sum.message), "Function tolerance reached. No non-constant parameter blocks found.");
        if (summary) *summary = sum;
        return SFMBA_OK;
    }
    HIP_TRY(hipSetDevice(p->device));
    const int want_rows = std::min(std::max(o.max_iters, 0) + 2, 1 << 16);
    int rc = ensure_trace(p, want_rows);
    if (rc) return rc;
    // no stream synchronisation here: whatever the caller enqueued before (reset, set_params) is ordered by the stream
    const double t0 = now_seconds();

    LMState st;
    init_state(p, st, o);
    *p->h_state = st;
    { ProfScope ps(p->prof.on ? &p->prof : nullptr, KID_SETUP, p->stream);
      launch_begin(p->stream, p->ds, p->db, st, p->reset_pending ? p->d_cam0 : nullptr, p->reset_pending ? p->d_pts0 : nullptr);
      p->reset_pending = false;
      launch_linearise_setup<T>(p, o.jacobi_scaling, /*begun=*/true); }

    int term = -1, msg = MSG_NONE;
    int host_iter = 0;
    bool state_mirrored = false;
    bool first_linearisation = true;
    bool first_linear_solve = true;
    const bool f32_matrix = option_switch(o.pcg_f32_matrix, "SFMBA_PCG_F32_MATRIX", true);
    // two-level preconditioner (8 gauge vectors as a coarse space, dense_solver.hip)
    const bool coarse_cg = option_switch(o.pcg_coarse_space, "SFMBA_PCG_COARSE", true);
    // ... and for a sparsely filled reduced matrix (a camera graph of large diameter) the same vectors restricted to eight segments of the
    // camera order (dense_solver.hip "Segmented coarse space"); SFMBA_PCG_SEGMENTS=0|1 forces it off / on wherever it applies
    // -- where the structure says the camera order IS that path: a sparsely filled matrix whose blocks sit near the (cyclic) diagonal
    // (options.pcg_coarse_space: 0 = by structure, 1 = the eight global vectors only, 2 = the segments wherever they apply)
    const bool segments_cg = coarse_cg && (dense_pcg_segments_applicable(&p->solver) || dense_pcg_segments_streaming_applicable(&p->solver)) &&
                             option_switch(o.pcg_coarse_space == 2 ? 1 : o.pcg_coarse_space == 1 ? -1 : 0, "SFMBA_PCG_SEGMENTS", p->block_fill < 0.5 && p->block_band >= 0.9);
    const bool symmetric_cg = option_switch(o.pcg_symmetric, "SFMBA_PCG_SYMMETRIC", true) && !p->deterministic &&
                              !(segments_cg && dense_pcg_segments_streaming_applicable(&p->solver));
    const bool pcg_mode = o.linear_solver == SFMBA_LINEAR_PCG || (o.linear_solver == SFMBA_LINEAR_AUTO && p->ds.d > 256);
    // AUTO above 256 unknowns = the DENSE_SCHUR result through the CG: plain relative residual <= 1e-12, bounded iteration count,
    // Cholesky on the same linearisation if the CG does not get there (include/sfmba.h)
    const bool exact_pcg = pcg_mode && o.linear_solver == SFMBA_LINEAR_AUTO;
    const double cg_tol = exact_pcg ? std::min(o.pcg_tolerance > 0.0 ? o.pcg_tolerance : auto_cg_tol(), auto_cg_tol()) : o.pcg_tolerance;
    const int cg_max_iters = exact_pcg ? (o.pcg_max_iters > 0 ? o.pcg_max_iters : std::min(4 * p->ds.d, 200)) : o.pcg_max_iters;
    const bool anchored_cg = !exact_pcg && o.pcg_anchored != 0;
    const bool gated_cg = true;        // (the fallback is decided where the gated loop learns that the batch was too short)
    // AUTO picks per LM iteration: the CG while it is the cheaper way to the DENSE_SCHUR result, the factorisation once a linearisation
    // has needed more CG iterations than a factorisation costs (measured at d = 1201: 0.41 ms against 6.4 us per iteration = ~64
    // iterations; uniform co-visibility needs 14 per LM iteration, a banded reduced system ~160 -- profiles/r03_*_banded_*).
    const int nblk64 = p->ds.ld / 64;
    const int cg_break_even = std::max(30, (nblk64 <= 40 ? 33 : 20) * nblk64 / 10);
    int launched_controls = 0;
    const bool speculate = option_switch(o.early_linearise, "SFMBA_EARLY_LINEARISE", true) && !p->prof.on;
    bool build_enqueued = false;
    std::vector<int> lin_hist;
    int cholesky_fallbacks = 0;
    // AUTO: a linearisation of this solve cost more CG iterations than a factorisation would have.  Local to the solve: a
    // resident problem solved twice from the same point takes the same path twice -- solve / reset / solve is bitwise repeatable in
    // deterministic mode.
    bool auto_prefers_cholesky = false;
    // ... and what the STRUCTURE says before the first iteration: a sparsely filled reduced matrix is a camera graph of large diameter
    // (views along a path, tracks of neighbouring cameras: SfM.cpp:366-469 builds exactly that) -- block-Jacobi CG then needs hundreds of
    // iterations per linearisation (cfg3_banded, fill 0.29: ~160 at 1e-12) and the factorisation is the cheaper way to the DENSE_SCHUR
    // result from the first linearisation on.  A property of the problem, not of the call history: deterministic.
    // (With the segmented coarse space the CG needs ~49 iterations per linearisation at 1e-12 on cfg3_banded: 1 566 LM iterations/s against the
    // factorisation's 1 639 -- the factorisation stays AUTO's choice there; the PCG mode is where the segments pay: 1 224 -> 2 220.)
    // On the streaming path (d > 1280) it is the other way round: a 600-camera path costs 4.5 ms per factorisation against ~1.7 ms for the
    // segmented CG run to 1e-12 (tools/large_banded_check.py) -- AUTO keeps the CG there.
    // (fill alone is not the signature of a path: an unordered but well-connected collection is sparsely filled too and its CG converges in a
    // few iterations -- there the per-solve break-even below decides)
    if (exact_pcg && p->block_fill < 0.5 && p->block_band >= 0.9 && !(segments_cg && dense_pcg_segments_streaming_applicable(&p->solver))) auto_prefers_cholesky = true;
    p->h_lm_mail[0] = 0; p->h_lm_mail[1] = -1;
    for (;;) {
        if (o.max_seconds > 0.0 && now_seconds() - t0 >= o.max_seconds) { term = SFMBA_NO_CONVERGENCE; msg = MSG_MAX_TIME; break; }
        if (host_iter >= o.max_iters) {
            term = SFMBA_NO_CONVERGENCE; msg = MSG_MAX_ITERS;
            if (host_iter == 0) {
                // max_iters <= 0: Ceres still evaluates iteration 0 [Ceres-upstream: TrustRegionMinimizer::IterationZero] -- a non-finite evaluation ends the run
                // there as FAILURE -- and reports that cost as the final one; then the iteration limit is looked at BEFORE the gradient tolerance
                // (FinalizeIterationAndCheckIfMinimizerCanContinue: run time, iteration count, gradient, radius): NO_CONVERGENCE even at a stationary point
                // (tests/fuzz_parity.py --options: the summary used to come back with a cost of 0)
                launch_point_build<T>(p->stream, p->ds, p->db, o.jacobi_scaling ? 1 : 2);
                launch_cam_diag<T>(p->stream, p->ds, p->db);
                launch_schur_pairs<T>(p->stream, p->ds, p->db, 2);
                launch_schur_pairs<T>(p->stream, p->ds, p->db, 0);
                launch_finalize(p->stream, p->ds, p->db, 0);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(p->stream));
                rc = download_state(p);
                if (rc) return rc;
                if (p->h_state->termination == SFMBA_FAILURE) { term = p->h_state->termination; msg = p->h_state->message; }
            }
            break;
        }
        Profiler* prof = p->prof.on ? &p->prof : nullptr;
        const bool pcg = pcg_mode && !(exact_pcg && auto_prefers_cholesky);
        if (pcg) {
            if (dense_pcg_ensure_workspace(&p->solver)) return fail(SFMBA_ERR_ALLOC, "PCG workspace allocation failed");
            p->db.pcg_F = p->solver.Sfull;
            p->db.pcg_W = coarse_cg ? p->solver.W : nullptr;
            // fp32 Jacobian mode + streaming CG path: the preconditioned matrix is stored in fp32 (halves the HBM-bound matvec)
            p->db.pcg_F32 = (p->precision == SFMBA_PRECISION_F32J && f32_matrix) ? dense_pcg_want_f32(&p->solver) : nullptr;
            p->solver.use_f32 = p->db.pcg_F32 != nullptr;
            // the streaming CG on ONE triangle of S~ (dense_solver.hip "Symmetric streaming path"); its sums arrive through atomics: not for deterministic handles
            p->solver.symmetric = symmetric_cg && dense_pcg_symmetric_applicable(&p->solver);
            p->db.pcg_upper_only = p->solver.symmetric ? 1 : 0;      // ... and the pair pass then writes that triangle only
            p->db.pcg_zero = p->solver.symmetric ? p->solver.sym_zero : nullptr;        // S~ W~, the CG's products and partial sums: added into with atomics
            p->db.pcg_zero_n = p->solver.symmetric ? (int)p->solver.sym_zero_n : 0;
        }
        RoctxRange rx_iter(p->roctx, "sfmba LM iteration");
        {
        RoctxRange rx(p->roctx, "linearise: point_build + cam_diag");
        if (!build_enqueued) {
            ProfScope ps(prof, KID_POINT_BUILD, p->stream);
            launch_point_build<T>(p->stream, p->ds, p->db, first_linearisation ? (o.jacobi_scaling ? 1 : 2) : 0);
        }
        build_enqueued = false;
        first_linearisation = false;
        { ProfScope ps(prof, KID_CAM_DIAG, p->stream); launch_cam_diag<T>(p->stream, p->ds, p->db); }
        launch_schur_pairs<T>(p->stream, p->ds, p->db, 2);      // duplicate pairs inside diagonal blocks (usually none)
        }
        RoctxRange rx_reduce(p->roctx, "reduce: finalize + schur_pairs (reduced camera system)");
        if (pcg) {
            // the preconditioner (Linv of the damped diagonal blocks) is known before the pair pass, which then writes the
            // preconditioned matrix directly
            { ProfScope ps(prof, KID_FINALIZE, p->stream); launch_finalize(p->stream, p->ds, p->db, 1); }
            { ProfScope ps(prof, KID_SCHUR_PAIRS, p->stream); launch_schur_pairs<T>(p->stream, p->ds, p->db, 1); }
        } else {
            { ProfScope ps(prof, KID_SCHUR_PAIRS, p->stream); launch_schur_pairs<T>(p->stream, p->ds, p->db, 0); }
            { ProfScope ps(prof, KID_FINALIZE, p->stream); launch_finalize(p->stream, p->ds, p->db, 0); }
        }
        rx_reduce.end();
        DeviceBuffers dbu = p->db;
        if (sizeof(T) == 4) dbu.pu32 = p->d_pu32;       // F32J: the back-substitution's first sweep gathers fp32 camera records (ba_kernels.hip, k_cam_update / k_point_update)
        bool pcg_gated = false;
        RoctxRange rx_solve(p->roctx, pcg ? "solve: two-level CG on the reduced system" : "solve: Cholesky of the reduced system");
        if (pcg) {
            const int anchor = anchored_cg ? (first_linear_solve ? 1 : 2) : 0;
            first_linear_solve = false;
            {
                // Launch-per-iteration CG: a batch of the length the previous solve needed (+2) goes into the queue together
                // with the three kernels that consume the solution; those are GATED on the CG's done flag, so the host does
                // not wait for the linear solve.  If the batch was too short k_lm_control says so and more is enqueued.
                pcg_gated = gated_cg;
                const int it = dense_pcg_solve(p->stream, &p->solver, p->db.S, p->db.rhs, cg_tol, cg_max_iters, p->d_info, prof,
                                               /*finish=*/false, /*hist_key=*/host_iter, /*pretransformed=*/true, anchor, /*no_wait=*/pcg_gated, /*coarse=*/coarse_cg, /*segments=*/segments_cg);
                if (it < 0) return fail(SFMBA_ERR_ALLOC, "PCG workspace allocation failed");
                if (!pcg_gated) { sum.linear_iters += it; lin_hist.push_back(it); }
            }
            dbu.pcg_vec = p->solver.vec; dbu.pcg_linv = p->solver.binv; dbu.pcg_flags = p->solver.flags;
            dbu.cg_gate = pcg_gated ? p->solver.flags : nullptr;
            dbu.cg_force = 0;
        } else {
            dense_cholesky_solve(p->stream, &p->solver, p->db.S, p->db.rhs, p->d_info, prof);
            lin_hist.push_back(0);
        }
        rx_solve.end();
        RoctxRange rx_update(p->roctx, "update: cam_update + point_update + lm_control (and the wait for its verdict)");
        bool lm_done = false;
        while (!lm_done) {
            launch_back_substitution<T>(p->stream, p->ds, p->ds, dbu, prof);
            { ProfScope ps(prof, KID_CONTROL, p->stream); launch_control(p->stream, p->ds, dbu); }
            { ProfScope ps(prof, KID_EMPTY, p->stream); }   // two back-to-back event records: the bracketing overhead itself
            ++launched_controls;
            // every launch of this LM iteration is in the queue: a failed launch must not leave the host waiting for a post
            if (hipError_t le = hipGetLastError(); le != hipSuccess)
                return fail(SFMBA_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(le));
            // The next linearisation's first kernel goes into the queue BEFORE the host waits for the control kernel's verdict: the
            // mailbox round trip plus the launch latency into an empty queue were 16-19 us of idle GPU per LM iteration.  The kernel
            // looks at the LM state itself and returns at once if the solve ended or the iteration wants more CG first (bit 2).
            if (speculate && host_iter + 2 <= o.max_iters) {
                ProfScope ps(prof, KID_POINT_BUILD, p->stream);
                launch_point_build<T>(p->stream, p->ds, p->db, 4);
                build_enqueued = true;
            }
            // wait for k_lm_control's mailbox post (system-scope stores to host-mapped memory)
            volatile int* mb = p->h_lm_mail;
            const int wrc = wait_mailbox(mb, launched_controls, p->stream);
            if (wrc != 0) {
                const hipError_t se = hipStreamSynchronize(p->stream);
                return fail(SFMBA_ERR_HIP, std::string("LM iteration did not complete on the device: ") +
                                           (se != hipSuccess ? hipGetErrorString(se) : "stream drained without the control kernel's post"));
            }
            if (mb[1] == -2) {
                // the CG batch was too short: enqueue more iterations (or force the step once max_iters are spent), then the trio again
                // (the early linearisation kernel behind that control kernel has returned without doing anything)
                if (dense_pcg_more(p->stream, &p->solver, 8, prof) == 0) {
                    if (exact_pcg) {
                        // AUTO: the CG has spent its iterations without reaching 1e-12 -- solve THIS linearisation exactly instead.  The
                        // damped diagonal blocks, the focal column and the right-hand side are in db.S / db.rhs already (camera pass +
                        // k_finalize); the pair pass writes the off-diagonal blocks again, unpreconditioned, and the Cholesky takes over.
                        { ProfScope ps(prof, KID_SCHUR_PAIRS, p->stream); launch_schur_pairs<T>(p->stream, p->ds, p->db, 0); }
                        dense_cholesky_solve(p->stream, &p->solver, p->db.S, p->db.rhs, p->d_info, prof);
                        dbu.pcg_vec = nullptr; dbu.pcg_linv = nullptr; dbu.pcg_flags = nullptr; dbu.cg_gate = nullptr;
                        pcg_gated = false;
                        ++cholesky_fallbacks;
                    } else {
                        dbu.cg_force = 1;
                    }
                }
                build_enqueued = false;
                continue;
            }
            lm_done = true;
            state_mirrored = true;
            host_iter = mb[3];
            if (mb[1] != -1) { term = mb[1]; msg = mb[2]; }
            if (pcg_gated || (pcg && exact_pcg && dbu.cg_gate == nullptr)) {
                const int it = (pcg_gated || dbu.pcg_vec) ? mb[4] : p->solver.run.launched;      // after a fallback: the launches that were spent
                if (exact_pcg && it > cg_break_even) auto_prefers_cholesky = true;                 // the next linearisations of THIS solve are factorised
                dense_pcg_note(&p->solver, (int)lin_hist.size(), it);
                sum.linear_iters += it;
                lin_hist.push_back(it);
            }
        }
        if (term != -1) break;
        if (o.verbose) {
            rc = download_state(p);
            if (rc) return rc;
            std::fprintf(stderr, "[sfmba] it %3d cost %.12e |g|inf %.3e radius %.3e term %d\n", p->h_state->iter, p->h_state->cost,
                         p->h_state->gmax, p->h_state->radius, p->h_state->termination);
        }
    }
    // k_lm_control mirrored the LM state into the pinned block before its last mailbox post, and it is the last kernel of
    // an iteration: unless that post was missed (or events have to be collected) there is nothing to wait for or to copy
    if (!state_mirrored || p->prof.on) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        rc = download_state(p);
        if (rc) return rc;
    }
    if (p->prof.on) p->prof.collect();
    const LMState& hs = *p->h_state;
    p->cur = hs.cur;
    p->focal = hs.focal[hs.cur];
    sum.termination = term;
    sum.iterations = hs.iter;
    sum.successful_steps = hs.successful;
    sum.unsuccessful_steps = hs.unsuccessful;
    sum.residual_evals = hs.residual_evals;
    sum.jacobian_evals = hs.jacobian_evals;
    sum.final_cost = hs.cost;
    sum.seconds = now_seconds() - t0;
    sum.cholesky_fallbacks = cholesky_fallbacks;
    std::snprintf(sum.message, sizeof(sum.message), "%s", message_text(msg));
    // trace rows
    const int rows = std::min(hs.iter + 1, p->db.trace_cap);
    std::vector<TraceRow> tr((size_t)std::max(rows, 1));
    if (p->trace_mapped) std::memcpy(tr.data(), p->kit.pinned + 4096, sizeof(TraceRow) * (size_t)rows);
    else HIP_TRY(hipMemcpy(tr.data(), p->db.trace, sizeof(TraceRow) * (size_t)rows, hipMemcpyDeviceToHost));
    sum.initial_cost = rows > 0 ? tr[0].cost : hs.cost;
    for (int r = 1; r < rows && r - 1 < (int)lin_hist.size(); ++r) tr[(size_t)r].linear_iters = lin_hist[(size_t)r - 1];
    if (trace && trace_cap > 0) {
        const int n = std::min(rows, trace_cap);
        static_assert(sizeof(TraceRow) == sizeof(sfmba_iteration), "trace row layout");
        std::memcpy(trace, tr.data(), sizeof(TraceRow) * (size_t)n);
    }
    if (trace_len) *trace_len = trace ? std::min(rows, std::max(trace_cap, 0)) : rows;     // rows written (all rows if only the count was asked for)
    if (summary) *summary = sum;
    return SFMBA_OK;
}

}  // namespace

extern "C" {

void sfmba_options_default(sfmba_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->max_iters = 500;               // BA.cpp:174
    o->max_seconds = 10.0;            // BA.cpp:176
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_radius = 1e4;
    o->max_radius = 1e16;
    o->min_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->jacobi_scaling = 1;
    o->max_consecutive_invalid_steps = 5;
    o->linear_solver = SFMBA_LINEAR_AUTO;       // DENSE_SCHUR-equivalent result (BA.cpp:172), cheapest solver that delivers it
    o->precision = SFMBA_PRECISION_F64;
    o->pcg_tolerance = 1e-8;
    o->pcg_max_iters = 0;
    o->verbose = 0;
    o->pcg_anchored = 1;
}

int sfmba_abi_version(void) { return SFMBA_ABI_VERSION; }

const char* sfmba_last_error(void) { return g_last_error.c_str(); }

long long sfmba_release_cache(void) { return (long long)arena_cache_release(); }

int sfmba_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace { __global__ void k_warm(int* p) { if (p && threadIdx.x == 0) *p = 1; } }

// What the first call of a process pays once (measured at BASELINE config 3, profiles/r05_b_shim_incremental.txt: 138 ms of which 113 in
// sfmba_problem_create -- the HIP context, the first pinned allocation, the first device chunks -- against 2.6 - 3 ms for every later adjustBundle()):
// a host can pay it at start-up instead.  Creates the context, one stream + pinned block (cached for the first problem), a pinned upload buffer and
// device chunks for a problem of `expected_obs` observations (0: contexts and the fixed-size pieces only), launches one kernel.  Idempotent.
int sfmba_device_warmup(int device, int64_t expected_obs) {
    if (expected_obs < 0) return fail(SFMBA_ERR_INVALID_ARG, "negative size");
    int rc = check_device(device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));
    HostKit kit;
    if (!hostkit_acquire(device, &kit)) return fail(SFMBA_ERR_HIP, "stream / pinned memory creation failed");
    struct KitGuard { HostKit* k; ~KitGuard() { if (k->stream) (void)hipStreamSynchronize(k->stream); hostkit_release(*k); } } kg{ &kit };
    if (expected_obs > 0) (void)hostkit_upload(&kit, std::min((size_t)HOSTKIT_UPLOAD_MAX, (size_t)expected_obs * 24));      // (camera, point, xy as doubles: the packed upload of a build)
    {
        DeviceArena arena(device);
        // a resident problem holds ~150 bytes per observation in structure, tables and sort temporaries (18 + 20 + 12 + 18 MB ... at one million: DESIGN.md section 3)
        const size_t want = (size_t)(4 << 20) + (size_t)expected_obs * 160;
        int* flag = static_cast<int*>(arena.alloc(want));
        if (!flag) return fail(SFMBA_ERR_ALLOC, "device allocation failed");
        hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, kit.stream, flag);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(kit.stream));
        arena.release();          // -> the chunk cache the first problem draws from
    }
    return SFMBA_OK;
}

void sfmba_problem_destroy(sfmba_problem* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    dense_solver_destroy(&p->solver);
    if (p->db.trace && !p->trace_mapped) (void)hipFree(p->db.trace);
    p->arena.release();
    p->prof.destroy();
    hostkit_release(p->kit);
    delete p;
}

static int create_impl(int device, int precision, int flags, int n_cam, const double* cam6, const unsigned char* cam_active, int n_pt, const double* pt3,
                       int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                       double focal, int rank, int world, sfmba_problem** out);

int sfmba_problem_create(int device, int precision, int n_cam, const double* cam6, int n_pt, const double* pt3,
                         int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                         double focal, sfmba_problem** out) {
    return create_impl(device, precision, 0, n_cam, cam6, nullptr, n_pt, pt3, n_obs, obs_cam, obs_pt, obs_xy, focal, 0, 1, out);
}

int sfmba_problem_create_ex(int device, int precision, int flags, int n_cam, const double* cam6, const unsigned char* cam_active, int n_pt,
                            const double* pt3, int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                            double focal, int rank, int world, sfmba_problem** out) {
    if (flags & ~(SFMBA_CREATE_DETERMINISTIC | SFMBA_CREATE_ROW_SHARDED | SFMBA_CREATE_NO_PAIR_LIST)) return fail(SFMBA_ERR_INVALID_ARG, "unknown create flag");
    if ((flags & SFMBA_CREATE_NO_PAIR_LIST) && (cam_active || (flags & SFMBA_CREATE_ROW_SHARDED))) return fail(SFMBA_ERR_INVALID_ARG, "SFMBA_CREATE_NO_PAIR_LIST applies to unsharded problems");
    const bool sharded = cam_active != nullptr || (flags & SFMBA_CREATE_ROW_SHARDED) != 0;
    if (sharded && (world < 1 || rank < 0 || rank >= world || world > SFMBA_SHARD_SCALARS - 16)) return fail(SFMBA_ERR_INVALID_ARG, "bad rank/world");
    return create_impl(device, precision, flags, n_cam, cam6, cam_active, n_pt, pt3, n_obs, obs_cam, obs_pt, obs_xy, focal,
                       sharded ? rank : 0, sharded ? world : 1, out);
}

int sfmba_problem_create_sharded(int device, int precision, int n_cam, const double* cam6, const unsigned char* cam_active,
                                 int n_pt, const double* pt3, int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt,
                                 const double* obs_xy, double focal, int rank, int world, sfmba_problem** out) {
    if (world < 1 || rank < 0 || rank >= world || world > SFMBA_SHARD_SCALARS - 16) return fail(SFMBA_ERR_INVALID_ARG, "bad rank/world");
    return create_impl(device, precision, 0, n_cam, cam6, cam_active, n_pt, pt3, n_obs, obs_cam, obs_pt, obs_xy, focal, rank, world, out);
}

int sfmba_problem_reset(sfmba_problem* p);
static int flush_reset(sfmba_problem* p);

// Observations a (re)build starts from: the point-major arrays of the previous structure (device, old arena) and / or new
// observations on the host (caller indices, mapped through the slot tables of the problem).
struct ObsSource {
    int n_old = 0;
    const int* d_old_pt = nullptr; const int* d_old_cam = nullptr; const void* d_old_xy = nullptr; const int* d_old_perm = nullptr;
    int n_new = 0;
    const int32_t* cam = nullptr; const int32_t* pt = nullptr; const double* xy = nullptr;
};

// Builds everything that depends on the observation list into p->arena (which must be empty of structure): point-major order,
// camera-major index, pair lists, launch descriptors, parameter / record / reduced-system buffers, dense-solver workspace;
// uploads the parameters.  All sorting and list building runs on the device (structure_build.hip); the host derives the
// launch descriptors from its own per-point / per-camera observation counts (no device round trip).  The counterpart of the reference's AddResidualBlock loop
// (BA.cpp:142-166) -- and, for sfmba_problem_append, of re-running it after a view was added (SfM.cpp:464-466).
static int build_structure(sfmba_problem* p, const ObsSource& src, const double* cam6, const double* pt3, double focal, bool sharded) {
    const int device = p->device, precision = p->precision;
    const bool bt_on = std::getenv("SFMBA_BUILD_TIMING") != nullptr;
    auto bt_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double bt_t = bt_now();
    auto bt_mark = [&](const char* what) { if (bt_on) { const double t = bt_now(); std::fprintf(stderr, "[sfmba build] %-18s %.3f ms\n", what, 1e3 * (t - bt_t)); bt_t = bt_now(); } };
    ArenaScope arena_scope(&p->arena);
    const int ncam = (int)p->acam_id.size(), npt = (int)p->apt_id.size();
    const long long nobs64 = (long long)src.n_old + src.n_new;
    if (nobs64 >= ((long long)1 << 31)) return fail(SFMBA_ERR_INVALID_ARG, "too many observations");
    const int nobs = (int)nobs64;
    const bool f32 = precision == SFMBA_PRECISION_F32J;
    const int xy_bytes = f32 ? 8 : 16;

    // ---- per-point / per-camera observation counts on the HOST (slot order), kept across appends: the CSR pointers, the pair total
    // and every launch descriptor that depends on them are known without asking the device, so the whole structure build is one
    // enqueue with a single wait at its end (it used to be sort -> wait -> copy back -> host loop, three times over) ----
    DeviceArena staging(device);           // unsorted observations + sort temporaries: released once the stream has drained
    staging.set_zeroing(false);
    p->h_pt_cnt.resize((size_t)npt, 0); p->h_cam_cnt.resize((size_t)ncam, 0);
    // new observations as ONE packed host buffer: point slots | camera slots | caller indices | coordinates
    const size_t nn = (size_t)src.n_new;
    // (in the kit's pinned block, grown to fit and kept: the upload is a plain asynchronous copy, and a rebuild does not fault in
    // 16 MB of fresh pageable memory first -- that alone was 5 ms of the shim's rebuild path)
    const size_t h_new_bytes = (3 * sizeof(int) + (size_t)xy_bytes) * nn + 16;
    char* h_new = hostkit_upload(&p->kit, h_new_bytes);
    std::vector<char> h_new_heap(h_new ? 0 : h_new_bytes);
    if (!h_new) h_new = h_new_heap.data();
    int* h_pt = reinterpret_cast<int*>(h_new);
    int* h_cam = h_pt + nn;
    int* h_perm = h_cam + nn;
    char* h_xy = h_new + (3 * sizeof(int) * nn + 15) / 16 * 16;                  // 16-byte aligned (double2 loads on the device)
    if (src.n_new > 0) {
        const int perm0 = (int)p->n_obs - src.n_new;           // caller index of the first new observation
        std::mutex cam_mu;
        int* pt_cnt = p->h_pt_cnt.data();
        // (one thread below PARALLEL_FOR_MIN items: plain increments then -- a locked add is a full fence on x86 and would serialise the
        // cache misses of the scattered point slots, which are all this loop costs when a view is appended)
        const bool one_thread = src.n_new < PARALLEL_FOR_MIN;
        parallel_for(src.n_new, [&](int k0, int k1) {
            std::vector<int> cam_local((size_t)ncam, 0);
            for (int k = k0; k < k1; ++k) {
                if (k + 16 < k1) { const int pf = p->pt_slot[(size_t)src.pt[k + 16]]; __builtin_prefetch(pt_cnt + pf, 1); }
                const int ps = p->pt_slot[(size_t)src.pt[k]], cs = p->cam_slot[(size_t)src.cam[k]];
                h_pt[(size_t)k] = ps; h_cam[(size_t)k] = cs; h_perm[(size_t)k] = perm0 + k;
                if (one_thread) ++pt_cnt[ps]; else __atomic_fetch_add(pt_cnt + ps, 1, __ATOMIC_RELAXED);
                ++cam_local[(size_t)cs];
                if (f32) { float* d = reinterpret_cast<float*>(h_xy) + 2 * (size_t)k; d[0] = (float)src.xy[2 * (size_t)k]; d[1] = (float)src.xy[2 * (size_t)k + 1]; }
                else { double* d = reinterpret_cast<double*>(h_xy) + 2 * (size_t)k; d[0] = src.xy[2 * (size_t)k]; d[1] = src.xy[2 * (size_t)k + 1]; }
            }
            std::lock_guard<std::mutex> lk(cam_mu);
            for (int j = 0; j < ncam; ++j) p->h_cam_cnt[(size_t)j] += cam_local[(size_t)j];
        });
    }
    bt_mark("counts");
    // row-sharded: this rank's range of point slots (ceil(n / world) each: the per-point arrays are padded so that the table goes through an
    // in-place all-gather of equal slices) and its block rows of the reduced matrix (the partition of the distributed CG)
    const bool rowsh = p->row_sharded && sharded;
    int own0 = 0, own1 = npt, pt_stride = npt, brow0 = 0, brow1 = ncam;
    if (rowsh) {
        pt_stride = (npt + p->shard_world - 1) / p->shard_world;
        own0 = std::min(npt, p->shard_rank * pt_stride); own1 = std::min(npt, own0 + pt_stride);
        std::vector<int> rows; long long chunk_blocks = 0;
        dcg_partition(ncam, p->shard_world, &rows, &chunk_blocks);
        brow0 = rows[(size_t)p->shard_rank]; brow1 = rows[(size_t)p->shard_rank + 1];
    }
    p->own_pt0 = own0; p->own_pt1 = own1; p->own_pt_stride = pt_stride;
    const size_t npt_alloc = rowsh ? (size_t)pt_stride * (size_t)p->shard_world : (size_t)npt;
    PointMajor pm;
    const int64_t nblock64 = (int64_t)ncam * (ncam + 1) / 2;
    if (nblock64 >= ((int64_t)1 << 31)) return fail(SFMBA_ERR_INVALID_ARG, "too many cameras");
    const int nblock = (int)nblock64;
    auto block_of = [ncam](int ja, int jb) { return (int)((int64_t)ja * ncam - (int64_t)ja * (ja - 1) / 2 + (jb - ja)); };

    // ---- host half of the build: CSR pointers from the counts, launch descriptors, their upload, the parameters.  It runs on the
    // process's helper thread WHILE this thread enqueues the sorts (some forty launches: the enqueue is what the build costs now);
    // the two halves meet once, at the pair total (needed to size the pair list), and join before the last two launches. ----
    std::vector<int> pt_ptr((size_t)npt + 1, 0), cam_ptr((size_t)ncam + 1, 0);
    long long npair_total = 0;
    std::atomic<int> counts_state(0);            // 1 = pointers and pair total ready, -1 = counts inconsistent
    std::vector<int4> chunks, chunks_coarse;
    std::vector<int> cam_chunk_ptr((size_t)ncam + 1, 0), pt_order, chunk_order, coarse_order;
    std::vector<int2> blk_cams, pwg_blocks;
    std::vector<double> cam0, pts0;
    std::vector<char> blob;
    int pair_lpb = 64, blocks_per_wg = 1;
    size_t pair_slot_cap = 0;
    // (SFMBA_PAIR_LIMIT: a test hook that lowers the threshold so that the matrix-free path runs at test size; read when a problem is built)
    long long pair_limit = (long long)1 << 31;
    if (const char* e = std::getenv("SFMBA_PAIR_LIMIT")) { const long long v = std::atoll(e); if (v > 0 && v < pair_limit) pair_limit = v; }
    auto host_half = [&]() -> int {
        for (int i = 0; i < npt; ++i) { const long long m = p->h_pt_cnt[(size_t)i]; pt_ptr[(size_t)i + 1] = pt_ptr[(size_t)i] + (int)m; npair_total += m * (m - 1) / 2; }
        for (int j = 0; j < ncam; ++j) cam_ptr[(size_t)j + 1] = cam_ptr[(size_t)j] + p->h_cam_cnt[(size_t)j];
        // More pairs of observations than a pair list can hold (32-bit positions; 16 bytes of sort workspace each) -- a hundred cameras that all
        // see 440k points, say; the reference adds a residual block per (view, point) without any bound on the track length, BA.cpp:142-166 --
        // or a caller that asked for it: no pair list is built and the solve applies the reduced matrix implicitly (include/sfmba.h).
        if (!sharded && npair_total >= pair_limit) p->no_pairs = true;
        const bool counts_ok = pt_ptr[(size_t)npt] == nobs && cam_ptr[(size_t)ncam] == nobs && (p->no_pairs || npair_total < ((long long)1 << 31));
        counts_state.store(counts_ok ? 1 : -1, std::memory_order_release);
        if (!counts_ok) return SFMBA_OK;         // (reported by the other half)
        // ---- launch descriptors (from the CSR pointer arrays) ----
        // chunks of the camera-major list: (camera, entry range)
        const int chunk_len = SFMBA_CAM_CHUNK;   // k_cam_diag: one lane per entry, one workgroup per chunk
        const int coarse_len = p->deterministic ? (1 << 30) : 1024;      // deterministic mode: one column-norm workgroup per camera (single writer)
        for (int j = 0; j < ncam; ++j) {
            cam_chunk_ptr[(size_t)j] = (int)chunks.size();
            for (int e0 = cam_ptr[j]; e0 < cam_ptr[(size_t)j + 1]; e0 += chunk_len) {
                int4 c; c.x = j; c.y = e0; c.z = std::min(e0 + chunk_len, cam_ptr[(size_t)j + 1]); c.w = 0;
                chunks.push_back(c);
            }
            for (int e0 = cam_ptr[j]; e0 < cam_ptr[(size_t)j + 1]; e0 += coarse_len) {
                int4 c; c.x = j; c.y = e0; c.z = (int)std::min<long long>((long long)e0 + coarse_len, cam_ptr[(size_t)j + 1]); c.w = 0;
                chunks_coarse.push_back(c);
            }
        }
        cam_chunk_ptr[(size_t)ncam] = (int)chunks.size();
        // launch order of the chunks (ba_kernels.h, chunk_order): by the chunk's relative position in its camera's list (a camera's entries ascend
        // in point slot, so that is -- for any co-visibility that samples the points evenly -- the window of the per-point table it gathers from),
        // cameras side by side inside a window (1024 windows).  The chunk list itself stays camera-major (deterministic mode adds a camera's chunks in list order).
        auto order_of = [&](const std::vector<int4>& ch, std::vector<int>* order) {
            // (a counting sort over 1024 windows, stable in list order: this runs on the append path of the incremental caller)
            constexpr int NWIN = 1024;
            order->resize(ch.size());
            std::vector<int> win(ch.size()), start(NWIN + 1, 0);
            for (size_t c = 0; c < ch.size(); ++c) {
                const int j = ch[c].x;
                const long long cnt = std::max(1, cam_ptr[(size_t)j + 1] - cam_ptr[(size_t)j]);
                const long long mid = (long long)(ch[c].y - cam_ptr[(size_t)j]) + (ch[c].z - ch[c].y) / 2;
                win[c] = (int)std::min<long long>(NWIN - 1, mid * NWIN / cnt);
                ++start[(size_t)win[c] + 1];
            }
            for (int w = 0; w < NWIN; ++w) start[(size_t)w + 1] += start[(size_t)w];
            for (size_t c = 0; c < ch.size(); ++c) (*order)[(size_t)start[(size_t)win[c]]++] = (int)c;
        };
        order_of(chunks, &chunk_order);
        order_of(chunks_coarse, &coarse_order);
        blk_cams.resize((size_t)nblock);
        for (int ja = 0; ja < ncam; ++ja)
            for (int jb = ja; jb < ncam; ++jb) { int2 c; c.x = ja; c.y = jb; blk_cams[(size_t)block_of(ja, jb)] = c; }
        // workgroups of the pair pass: consecutive blocks of ONE block-row each; rows are dealt to the 8
        // XCDs (blockIdx % 8, the observed dispatch order) so a row's records stay in one L2.  Performance
        // only: any placement gives the same result.
        // Lanes per block of the pair pass, from the mean number of pairs of an off-diagonal block: a whole wave (64 pairs per
        // round) or 16 lanes (4 blocks per wave).  Measured on MI355X: 16 lanes win at 56 pairs per block (110 vs 139 us) and
        // below (280 vs 738 us at 5.6), the whole wave wins at 226 (77 vs 105 us).  SFMBA_PAIR_LPB overrides.
        // ... per NON-EMPTY block where that is known: a handle that is rebuilt (sfmba_problem_append: the incremental loop) remembers the fill of its
        // previous structure (1.0 on the first build).  A long camera path -- 600 cameras, 6 % of the blocks non-empty, ~1 250 pairs in each of those but
        // 75 over all blocks -- runs the wave-per-chunk pass 627 -> 370 us faster than sixteen lanes per block (tools/large_banded_check.py).
        const double mean_pairs = (double)npair_total / std::max(1.0, p->block_fill * (double)std::max(1, nblock - ncam));
        pair_lpb = mean_pairs >= 128.0 ? 64 : 16;
        // A FIRST build does not know the fill.  Where the plain mean says "sixteen lanes" for a problem with many cameras, a sample decides: the
        // pairs of ~1000 points (every 2^s-th slot), their blocks sorted, m_b sampled pairs in block b -- pairs of one point never share a block, so
        // sum_b m_b (m_b - 1) estimates q^2 sum_b n_b^2 (q: the sampled share of the points) and  sum n_b^2 / sum n_b = C2 / (q * keys) + 1  is the
        // number of pairs in the block an average PAIR lives in: mean + 1 for uniform co-visibility, ~800 on the 600-camera path (mean 75).
        if (!p->no_pairs && pair_lpb == 16 && src.n_old == 0 && ncam > 64 && nn > 100000 && mean_pairs >= 8.0) {
            int shift = 0;
            while ((npt >> shift) > 1024) ++shift;
            const int mask = (1 << shift) - 1;
            std::vector<std::pair<int, int>> smp;
            { std::mutex mu;
              parallel_for((int)nn, [&](int k0, int k1) {
                  std::vector<std::pair<int, int>> loc;
                  for (int k = k0; k < k1; ++k) if ((h_pt[(size_t)k] & mask) == 0) loc.emplace_back(h_pt[(size_t)k], h_cam[(size_t)k]);
                  std::lock_guard<std::mutex> lk(mu);
                  smp.insert(smp.end(), loc.begin(), loc.end());
              }); }
            std::sort(smp.begin(), smp.end());
            std::vector<int> keys;
            size_t npts_s = 0;
            for (size_t a = 0; a < smp.size() && keys.size() < ((size_t)1 << 22);) {      // (bounded: long tracks make m (m - 1) / 2 keys per point)
                size_t b = a;
                while (b < smp.size() && smp[b].first == smp[a].first) ++b;
                ++npts_s;
                for (size_t u = a; u < b; ++u)
                    for (size_t v = u + 1; v < b; ++v) if (smp[u].second != smp[v].second) keys.push_back(block_of(std::min(smp[u].second, smp[v].second), std::max(smp[u].second, smp[v].second)));
                a = b;
            }
            std::sort(keys.begin(), keys.end());
            double c2 = 0.0;
            for (size_t a = 0; a < keys.size();) { size_t b = a; while (b < keys.size() && keys[b] == keys[a]) ++b; c2 += (double)(b - a) * (double)(b - a - 1); a = b; }
            const double q = (double)npts_s / (double)std::max(1, npt);
            const double weighted = (keys.empty() || q <= 0.0) ? mean_pairs : c2 / (q * (double)keys.size()) + 1.0;
            if (weighted >= 4.0 * 128.0) pair_lpb = 64;         // (well beyond the uniform crossover: the empty blocks of such a structure cost a wave each)
        }
        if (const char* e = std::getenv("SFMBA_PAIR_LPB")) { const int v = std::atoi(e); if (v == 64 || v == 16) pair_lpb = v; }
        blocks_per_wg = pair_lpb == 64 ? 1 : 64 / pair_lpb;
        {
            std::vector<std::vector<int2>> per_xcd(8);
            for (int ja = brow0; ja < (p->no_pairs ? brow0 : brow1); ++ja) {       // (a row-sharded rank: its own block rows only -- all pairs of each of their blocks; no pair list: no pair pass)
                const int b0 = block_of(ja, ja), nb = ncam - ja;
                for (int o = 0; o < nb; o += blocks_per_wg) { int2 w; w.x = b0 + o; w.y = std::min(blocks_per_wg, nb - o); per_xcd[ja % 8].push_back(w); }
            }
            size_t longest = 0;
            for (auto& v : per_xcd) longest = std::max(longest, v.size());
            for (size_t m = 0; m < longest; ++m)
                for (int x = 0; x < 8; ++x) {
                    int2 w; w.x = 0; w.y = 0;
                    if (m < per_xcd[x].size()) w = per_xcd[x][m];
                    pwg_blocks.push_back(w);
                }
        }
        // order of the points in the point passes (k_point_build / k_point_update: four lanes per point, sixteen points per
        // wave): by number of rounds of four observations, stable -- the quads of a wave then loop alike whatever the track lengths
        // (tracks of 2..30 views: a wave of unsorted points runs to its longest track).  Uniform track length: slot order, no array.
        {
            constexpr int NB = 18;                 // rounds 1 .. 16, > 16 in one bucket (+ the empty bucket 0)
            int cnt_b[NB + 1] = {};
            auto bucket = [&](int i) { const int k = pt_ptr[(size_t)i + 1] - pt_ptr[i]; return std::min((k + 3) / 4, NB - 1); };
            // (a row-sharded rank: the list of its OWN points -- always materialised, position and slot differ)
            for (int i = own0; i < own1; ++i) ++cnt_b[bucket(i) + 1];
            int used = 0;
            for (int b = 0; b < NB; ++b) used += cnt_b[b + 1] > 0;
            if (used > 1 || rowsh) {
                for (int b = 0; b < NB; ++b) cnt_b[b + 1] += cnt_b[b];
                pt_order.resize((size_t)std::max(own1 - own0, 1));
                for (int i = own0; i < own1; ++i) pt_order[(size_t)cnt_b[bucket(i)]++] = i;
            }
        }
        // upload: ONE synchronous copy on the NULL stream (the problem's stream is non-blocking: it runs beside the sorts) of the
        // arrays laid out back to back, 256-byte aligned
        {
            size_t off = 0;
            auto place = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
            const size_t o_chunks = place(sizeof(int4) * chunks.size()), o_coarse = place(sizeof(int4) * chunks_coarse.size()),
                         o_ccp = place(sizeof(int) * cam_chunk_ptr.size()), o_bc = place(sizeof(int2) * blk_cams.size()),
                         o_pwg = place(sizeof(int2) * pwg_blocks.size()), o_pto = place(sizeof(int) * pt_order.size()),
                         o_co = place(sizeof(int) * chunk_order.size()), o_cco = place(sizeof(int) * coarse_order.size());
            blob.resize(off ? off : 1);
            auto put = [&](size_t o, const void* src, size_t bytes) { if (bytes) std::memcpy(blob.data() + o, src, bytes); };
            put(o_chunks, chunks.data(), sizeof(int4) * chunks.size()); put(o_coarse, chunks_coarse.data(), sizeof(int4) * chunks_coarse.size());
            put(o_ccp, cam_chunk_ptr.data(), sizeof(int) * cam_chunk_ptr.size()); put(o_bc, blk_cams.data(), sizeof(int2) * blk_cams.size());
            put(o_pwg, pwg_blocks.data(), sizeof(int2) * pwg_blocks.size());
            put(o_pto, pt_order.data(), sizeof(int) * pt_order.size());
            put(o_co, chunk_order.data(), sizeof(int) * chunk_order.size()); put(o_cco, coarse_order.data(), sizeof(int) * coarse_order.size());
            char* d_blob = nullptr;
            HIP_TRY(dev_alloc(&d_blob, blob.size()));
            HIP_TRY(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
            p->d_chunks = reinterpret_cast<int4*>(d_blob + o_chunks); p->d_chunks_coarse = reinterpret_cast<int4*>(d_blob + o_coarse);
            p->d_cam_chunk_ptr = reinterpret_cast<int*>(d_blob + o_ccp); p->d_blk_cams = reinterpret_cast<int2*>(d_blob + o_bc);
            p->d_pwg_blocks = reinterpret_cast<int2*>(d_blob + o_pwg);
            p->d_pt_order = pt_order.empty() ? nullptr : reinterpret_cast<int*>(d_blob + o_pto);
            p->d_chunk_order = reinterpret_cast<int*>(d_blob + o_co); p->d_coarse_order = reinterpret_cast<int*>(d_blob + o_cco);
        }
        // (wave-per-block pass: one descriptor per chunk of SFMBA_PAIR_CHUNK pairs -- as many as the pair total allows at most)
        pair_slot_cap = p->no_pairs ? 1 : pair_lpb == 64 ? pwg_blocks.size() + (size_t)(npair_total / SFMBA_PAIR_CHUNK) + 1 : std::max<size_t>(pwg_blocks.size() * (size_t)blocks_per_wg, 1);
        HIP_TRY(dev_alloc(&p->d_pwg_desc, pair_slot_cap));
        if (pair_lpb == 64) {
            HIP_TRY(dev_alloc(&p->d_pwg_chunk, pair_slot_cap));
            HIP_TRY(dev_alloc(&p->d_multi_slots, pwg_blocks.size() + 1));
        }
        HIP_TRY(dev_alloc(&p->d_build_counters, (size_t)4));
        HIP_TRY(dev_alloc(&p->d_dup_blocks, (size_t)ncam));
        return SFMBA_OK;
    };
    // ---- parameters.  Slot order = order of first observation; when every camera / point of the caller's arrays is observed and the slots came
    // out in index order (the usual case: point-major observation lists), the arrays go up as they are.  Runs on the helper thread behind the host
    // half (round 6: the helper had 0.24 ms to spare at BASELINE config 3 + one view while this thread enqueued; the copies are synchronous, NULL stream) ----
    auto params_half = [&]() -> int {
        bool cam_identity = (size_t)ncam == p->cam_slot.size(), pt_identity = (size_t)npt == p->pt_slot.size();
        for (int j = 0; j < ncam && cam_identity; ++j) cam_identity = p->acam_id[j] == j;
        for (int i = 0; i < npt && pt_identity; ++i) pt_identity = p->apt_id[i] == i;
        p->cam_identity = cam_identity; p->pt_identity = pt_identity;
        const double* cam_src = cam6;
        const double* pts_src = pt3;
        if (!cam_identity) {
            cam0.resize((size_t)6 * ncam);
            for (int j = 0; j < ncam; ++j) std::memcpy(&cam0[6 * (size_t)j], cam6 + 6 * (size_t)p->acam_id[j], 6 * sizeof(double));
            cam_src = cam0.data();
        }
        if (!pt_identity) {
            pts0.resize((size_t)3 * npt);
            for (int i = 0; i < npt; ++i) std::memcpy(&pts0[3 * (size_t)i], pt3 + 3 * (size_t)p->apt_id[i], 3 * sizeof(double));
            pts_src = pts0.data();
        }
        HIP_TRY(dev_alloc(&p->d_cam0, (size_t)6 * ncam));
        HIP_TRY(dev_alloc(&p->d_pts0, (size_t)3 * npt));
        if (ncam > 0) HIP_TRY(hipMemcpy(p->d_cam0, cam_src, sizeof(double) * 6 * (size_t)ncam, hipMemcpyHostToDevice));
        if (npt > 0) HIP_TRY(hipMemcpy(p->d_pts0, pts_src, sizeof(double) * 3 * (size_t)npt, hipMemcpyHostToDevice));
        return SFMBA_OK;
    };
    int host_rc = SFMBA_OK;
    std::string host_msg;
    BuildHelper::Lease helper(BuildHelper::instance());
    auto host_task = [&] {
        (void)hipSetDevice(device);
        ArenaScope helper_scope(&p->arena);
        const double th0 = bt_now();
        host_rc = host_half();
        const double th1 = bt_now();
        if (host_rc == SFMBA_OK && counts_state.load() > 0) host_rc = params_half();
        if (host_rc != SFMBA_OK) host_msg = g_last_error;
        if (bt_on) std::fprintf(stderr, "[sfmba build] (helper) host half   %.3f ms + parameters %.3f ms\n", 1e3 * (th1 - th0), 1e3 * (bt_now() - th1));
    };
    if (helper.granted()) helper.post(host_task); else host_task();        // (another build has the helper: everything on this thread)
    bt_mark("post");
    // ---- unsorted observations on the device: one upload of the new ones, one kernel that lays old (device to device) and new
    // ones out behind each other; then the point-major sort ----
    {
        int* u_pt = staging.alloc_n<int>((size_t)nobs);
        int* u_cam = staging.alloc_n<int>((size_t)nobs);
        int* u_perm = staging.alloc_n<int>((size_t)nobs);
        char* u_xy = static_cast<char*>(staging.alloc((size_t)xy_bytes * std::max(nobs, 1)));
        char* d_new = static_cast<char*>(staging.alloc(h_new_bytes));
        if (!u_pt || !u_cam || !u_perm || !u_xy || !d_new) { helper.wait(); return fail(SFMBA_ERR_ALLOC, "device allocation failed"); }
        hipError_t ce = hipSuccess;
        if (src.n_new > 0) ce = hipMemcpyAsync(d_new, h_new, h_new_bytes, hipMemcpyHostToDevice, p->stream);
        if (ce == hipSuccess) {
            StageObs so;
            so.n_old = src.n_old; so.n_new = src.n_new; so.old_pt = src.d_old_pt; so.old_cam = src.d_old_cam; so.old_perm = src.d_old_perm; so.old_xy = src.d_old_xy;
            so.new_pt = reinterpret_cast<const int*>(d_new); so.new_cam = so.new_pt + nn; so.new_perm = so.new_cam + nn; so.new_xy = d_new + (h_xy - h_new);
            so.u_pt = u_pt; so.u_cam = u_cam; so.u_perm = u_perm; so.u_xy = u_xy;
            launch_stage_obs(p->stream, so, xy_bytes);
            ce = hipGetLastError();
        }
        if (ce != hipSuccess) { helper.wait(); return fail(SFMBA_ERR_HIP, std::string("staging the observations: ") + hipGetErrorString(ce)); }
        bt_mark("upload obs");
        const int brc = build_point_major(p->stream, &p->arena, &staging, nobs, npt, ncam, xy_bytes, u_pt, u_cam, u_perm, u_xy, &pm);
        if (brc) { helper.wait(); return fail(SFMBA_ERR_HIP, std::string("point-major build: ") + hipGetErrorString((hipError_t)brc)); }
    }
    p->d_pt_ptr = pm.pt_ptr; p->d_obs_cam = pm.obs_cam; p->d_obs_pt = pm.obs_pt; p->d_perm = pm.obs_pt + nobs; p->d_obs_xy = pm.obs_xy;
    bt_mark("enqueue point-major");
    while (counts_state.load(std::memory_order_acquire) == 0) __builtin_ia32_pause();
    if (counts_state.load() < 0) {
        helper.wait();
        if (npair_total >= ((long long)1 << 31)) return fail(SFMBA_ERR_INVALID_ARG, "too many observation pairs for a sharded problem (2^31: the matrix-free path is unsharded)");
        return fail(SFMBA_ERR_HIP, "observation counts out of step with the observation list");
    }
    const bool no_pairs = p->no_pairs;      // (decided by the host half before it published the counts)
    // camera-pair lists: for every point, every pair of its observations (qa < qb, cameras ascending; the self pairs are folded
    // into the camera-diagonal pass) goes to block (ja, jb) of the upper triangle of S -- listed as the pair's POINT, which is all the
    // re-evaluating pair pass reads per pair
    if (no_pairs) {
        p->d_blk_ptr = p->arena.alloc_n<int>((size_t)nblock + 1);          // (all empty: the arena hands out zeroed memory)
        p->d_pair_pt = p->arena.alloc_n<int>(1);
        if (!p->d_blk_ptr || !p->d_pair_pt) { helper.wait(); return fail(SFMBA_ERR_ALLOC, "device allocation failed"); }
    } else {
        const int brc = build_pair_lists(p->stream, &p->arena, &staging, npt, nobs, ncam, nblock, p->d_pt_ptr, p->d_obs_pt, p->d_obs_cam, pm.pair_off, npair_total,
                                         &p->d_blk_ptr, &p->d_pair_pt);
        if (brc) { helper.wait(); return fail(SFMBA_ERR_HIP, std::string("pair-list build: ") + hipGetErrorString((hipError_t)brc)); }
    }
    {
        const int crc = build_camera_major(p->stream, &p->arena, &staging, nobs, ncam, p->d_obs_cam, p->d_obs_pt, &p->d_cam_obs, &p->d_cam_obs_pt, &p->d_cam_ptr);
        if (crc) { helper.wait(); return fail(SFMBA_ERR_HIP, std::string("camera-major build: ") + hipGetErrorString((hipError_t)crc)); }
    }
    {
        void* cxy = nullptr;
        const int xrc = build_camera_major_xy(p->stream, &p->arena, nobs, xy_bytes, p->d_cam_obs, p->d_obs_xy, &cxy);
        if (xrc) { helper.wait(); return fail(SFMBA_ERR_HIP, std::string("camera-major coordinates: ") + hipGetErrorString((hipError_t)xrc)); }
        p->d_cam_obs_xy = cxy;
    }
    bt_mark("enqueue sorts");
    // ---- the buffers of the solve: sized by (cameras, points, observations) alone, carved out of the arena and zeroed WHILE the helper thread finishes
    // its half (descriptors, parameters); what depends on that half -- chunk counts, the pair-pass descriptors -- follows the join ----
    DeviceStructure& ds = p->ds;
    ds = DeviceStructure{};
    ds.ncam = ncam; ds.npt = npt; ds.nobs = nobs;
    ds.d = 6 * ncam + 1;
    ds.ld = dense_padded_dim(ds.d);
    ds.nblock = nblock;          // (shard_offdiag_len below sizes the exchange buffer from it)
    DeviceBuffers& db = p->db;
    db = DeviceBuffers{};
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(dev_alloc(&db.cam[b], (size_t)6 * ncam));
        HIP_TRY(dev_alloc(&db.pts[b], (size_t)3 * npt_alloc));
        HIP_TRY(dev_alloc(&db.camtab[b], (size_t)CT_STRIDE * ncam));
    }
    HIP_TRY(dev_alloc(&db.steptab, (size_t)ST_STRIDE * ncam));
    HIP_TRY(dev_alloc(&p->d_pu32, (size_t)20 * std::max(ncam, 1)));
    HIP_TRY(dev_alloc(&db.cscale, (size_t)6 * ncam));
    HIP_TRY(dev_alloc(&db.pscale, (size_t)3 * npt));
    // Nothing is stored per observation: the reduced-system passes and the back-substitution re-evaluate every observation from the
    // camera row and a per-point table (PA / PB, 64 + 24 bytes per point in fp32-Jacobian mode) plus a camera-major copy of the
    // observation coordinates (rounds 1 - 3 kept a 64-byte record per observation: 64 MB at BASELINE config 3).
    db.PA = p->arena.alloc(std::max<size_t>(npt_alloc, 1) * (f32 ? sizeof(PtRecA<float>) : sizeof(PtRecA<double>)));
    db.PB = p->arena.alloc(std::max<size_t>(npt_alloc, 1) * (f32 ? sizeof(PtRecB<float>) : sizeof(PtRecB<double>)));
    if (!db.PA || !db.PB) return fail(SFMBA_ERR_ALLOC, "device allocation failed");
    HIP_TRY(dev_alloc(&db.pt_t, (size_t)3 * npt));
    HIP_TRY(dev_alloc(&db.pt_M, (size_t)6 * npt));
    const size_t sys_len = (size_t)ds.ld * ds.ld + 3 * (size_t)ds.ld + SFMBA_SHARD_SCALARS;
    HIP_TRY(dev_alloc(&p->d_sys, sys_len));
    p->d_red = nullptr;
    if (no_pairs) {
        // the matrix-free solve runs the CG loop of the sharded path on one rank: the buffer of exchange (A)
        p->shard_blocks_off = (shard_diag_len(ds) + 63) / 64 * 64;
        HIP_TRY(dev_alloc(&p->d_red, (size_t)p->shard_blocks_off));
        p->dcg = DistCg(); p->dcg_last_f32 = -1;
        p->imp_dtab = p->imp_spt = p->imp_acc = p->imp_part = nullptr;
    }
    if (sharded) {
        // the all-reduce buffer: packed triangle of S + tail (exact solver), or the two blocks of the CG path (ba_kernels.hip, k_shard_diag)
        const size_t tri = (size_t)ds.ld * (ds.ld + 1) / 2 + 3 * (size_t)ds.ld + SFMBA_SHARD_SCALARS;
        // distributed CG: the blocks go behind the region of exchange (A), in `world` equal chunks (the padding stays zero)
        std::vector<int> rows; long long chunk = 0;
        dcg_partition(ncam, p->shard_world, &rows, &chunk);
        p->shard_blocks_off = (shard_diag_len(ds) + 63) / 64 * 64;
        const size_t dist = (size_t)p->shard_blocks_off + (size_t)p->shard_world * (size_t)chunk * 36;
        HIP_TRY(dev_alloc(&p->d_red, std::max(std::max(tri, dist), (size_t)std::max(shard_diag_len(ds), shard_offdiag_len(ds)))));
        p->dcg = DistCg(); p->dcg_last_f32 = -1;
        p->imp_dtab = p->imp_spt = p->imp_acc = p->imp_part = nullptr;
    }
    db.S = p->d_sys;
    db.rhs = db.S + (size_t)ds.ld * ds.ld;
    db.udiag = db.rhs + ds.ld;
    db.bc = db.udiag + ds.ld;
    p->d_scal = db.bc + ds.ld;
    db.shared_weight = 1.0;
    db.shard_blocks = nullptr; db.shard_blocks32 = nullptr; db.shard_scal = nullptr;
    HIP_TRY(dev_alloc(&db.st, 1));
    // padding of the reduced system (rows/columns >= d) is zero apart from the identity diagonal set by k_finalize
    HIP_TRY(hipMemsetAsync(p->d_sys, 0, sizeof(double) * sys_len, p->stream));
    HIP_TRY(dev_alloc(&p->d_info, 2));
    HIP_TRY(hipMemsetAsync(p->d_info, 0, 2 * sizeof(int), p->stream));
    db.lin_info = p->d_info;
    db.fin_counter = p->d_info + 1;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_pinned), p->kit.pinned, 0));
    db.lm_mailbox = reinterpret_cast<int*>(p->d_pinned + 1024);
    db.st_mirror = reinterpret_cast<LMState*>(p->d_pinned);
    db.trace = nullptr; db.trace_cap = 0; p->trace_mapped = false;
    dense_solver_destroy(&p->solver);
    if (dense_solver_create(&p->solver, ds.d, ds.ld, &p->arena, p->kit.pinned + 2048)) return fail(SFMBA_ERR_ALLOC, "dense solver workspace allocation failed");
    db.pcg_bt = p->solver.vec + (size_t)8 * ds.ld;
    db.pcg_binv = p->solver.binv;
    HIP_TRY(dev_alloc(&db.pair_G, (size_t)36 * std::max(ncam, 1)));
    bt_mark("alloc buffers");
    helper.wait();
    if (host_rc != SFMBA_OK) return fail(host_rc, host_msg);
    bt_mark("join host half");
    // what depends on the block CSR is filled in by the device, behind the pair sort: the pair-pass descriptors and the list of
    // diagonal blocks that contain pairs (the same camera observing a point twice; handled by a separate pass).  The number of
    // those and the device's own pair total come back through host-mapped memory and are read after the one wait below.
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_pinned), p->kit.pinned, 0));
    volatile int* build_report = reinterpret_cast<volatile int*>(p->kit.pinned + 1536);      // [1536, 1568) of the mailbox slice
    build_report[0] = -1;
    int* d_report = reinterpret_cast<int*>(p->d_pinned + 1536);
    build_report[1] = -1; build_report[4] = -1; build_report[5] = -1; build_report[6] = -1; build_report[7] = -1;
    HIP_TRY(hipMemsetAsync(p->d_build_counters, 0, 4 * sizeof(int), p->stream));
    if (pwg_blocks.empty()) {
        build_report[4] = 0; build_report[5] = 0; build_report[7] = 0;      // (a row-sharded rank without a block row, or no pair list: no pair pass)
    } else if (pair_lpb == 64) {
        const int crc = build_pair_chunks(p->stream, &staging, (int)pwg_blocks.size(), SFMBA_PAIR_CHUNK, p->d_pwg_blocks, p->d_blk_cams, p->d_blk_ptr,
                                          p->d_pwg_desc, p->d_pwg_chunk, p->d_multi_slots, p->d_build_counters, d_report);
        if (crc) return fail(SFMBA_ERR_HIP, std::string("pair-chunk descriptors: ") + hipGetErrorString((hipError_t)crc));
    } else {
        // sixteen lanes per block: the blocks of a row grouped by rounds of sixteen pairs (a wave holds four of them and loops to the longest)
        int* d_perm = staging.alloc_n<int>((size_t)nblock);
        if (!d_perm) return fail(SFMBA_ERR_ALLOC, "device allocation failed");
        launch_row_order(p->stream, ncam, pair_lpb, p->d_blk_ptr, d_perm);
        launch_pair_desc(p->stream, (int)pwg_blocks.size(), blocks_per_wg, p->d_pwg_blocks, p->d_blk_cams, p->d_blk_ptr, d_perm, p->d_pwg_desc);
    }
    launch_block_fill(p->stream, nblock, ncam, p->d_blk_cams, p->d_blk_ptr, p->d_build_counters, d_report);
    p->d_blk_mask = nullptr;
    if (6 * ncam + 1 > 1280 && !sharded && !no_pairs) {       // (the streaming CG kernels: a sparsely filled reduced matrix is multiplied block-sparse there)
        HIP_TRY(dev_alloc(&p->d_blk_mask, (size_t)ncam * (size_t)((ncam + 31) / 32)));
        launch_block_mask(p->stream, ncam, p->d_blk_ptr, p->d_blk_mask);
    }
    launch_dup_blocks(p->stream, ncam, p->d_blk_ptr, pm.pair_off + npt, p->d_dup_blocks, d_report);
    HIP_TRY(hipGetLastError());
    p->focal0 = p->focal = focal;

    ds.pt_ptr = p->d_pt_ptr; ds.pt_order = p->d_pt_order; ds.obs_cam = p->d_obs_cam; ds.obs_xy = p->d_obs_xy;
    ds.cam_ptr = p->d_cam_ptr; ds.cam_obs = p->d_cam_obs; ds.cam_obs_pt = p->d_cam_obs_pt; ds.cam_obs_xy = p->d_cam_obs_xy;
    // row-sharded: a contiguous, equally sized share of the camera-major chunks per rank (every chunk is <= 256 entries of one camera)
    p->own_chunk0 = 0; p->own_chunk1 = (int)chunks.size(); p->own_coarse0 = 0; p->own_coarse1 = (int)chunks_coarse.size();
    if (rowsh) {
        const long long nc_ = (long long)chunks.size(), ncc = (long long)chunks_coarse.size(), r_ = p->shard_rank, w_ = p->shard_world;
        p->own_chunk0 = (int)(nc_ * r_ / w_); p->own_chunk1 = (int)(nc_ * (r_ + 1) / w_);
        p->own_coarse0 = (int)(ncc * r_ / w_); p->own_coarse1 = (int)(ncc * (r_ + 1) / w_);
    }
    ds.nchunk = (int)chunks.size(); ds.chunks = p->d_chunks; ds.chunk_order = p->d_chunk_order; ds.coarse_order = p->d_coarse_order;
    ds.nchunk_coarse = (int)chunks_coarse.size(); ds.chunks_coarse = p->d_chunks_coarse; ds.cam_chunk_ptr = p->d_cam_chunk_ptr;
    ds.obs_pt = p->d_obs_pt;
    ds.nblock = nblock; ds.blk_cams = p->d_blk_cams; ds.blk_ptr = p->d_blk_ptr; ds.pair_pt = p->d_pair_pt;
    ds.npairwg = (int)pwg_blocks.size(); ds.pwg_blocks = p->d_pwg_blocks; ds.pair_lpb = pair_lpb;
    ds.pwg_group = blocks_per_wg; ds.pwg_desc = p->d_pwg_desc;
    ds.pwg_chunk = pair_lpb == 64 ? p->d_pwg_chunk : nullptr; ds.nmulti = 0; ds.multi_slots = p->d_multi_slots;      // counts: after the wait at the end
    ds.ndupwg = 0; ds.dup_blocks = p->d_dup_blocks;          // count: after the wait at the end

    // Opt-in deterministic accumulation (SFMBA_DETERMINISTIC=1 when the problem is built): every workgroup of a launch owns its
    // accumulator slot and the multi-chunk camera sums are added in chunk order, so that no result depends on the order in which
    // fp64 atomics arrive -- two runs give bit-identical trajectories.  Costs ~10 % (longer slot sweeps, a serial chunk loop).
    db.nslot = NSLOT;
    db.cd_part = nullptr;
    if (p->deterministic) {
        const int grid = std::max(std::max(1024, (ds.npt + 31) / 32 + 1), std::max(ds.nchunk, ds.nchunk_coarse));      // (point passes: 32 points per workgroup)
        db.nslot = (grid + 63) / 64 * 64;
        HIP_TRY(dev_alloc(&db.cd_part, (size_t)std::max(ds.nchunk, 1) * 48));
    }
    HIP_TRY(dev_alloc(&p->d_facc, (size_t)db.nslot * SLOT_W));
    HIP_TRY(hipMemsetAsync(p->d_facc, 0, sizeof(double) * (size_t)db.nslot * SLOT_W, p->stream));
    db.slots = p->d_facc;
    p->solver.blk_mask = p->d_blk_mask;
    bt_mark("descriptors");
    // the one wait of the build: sorts, lists and descriptors are in place; the staging arena and the host vectors may go
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (build_report[0] < 0 || (((long long)build_report[3] << 32) | (unsigned)build_report[2]) != npair_total)
        return fail(SFMBA_ERR_HIP, "structure build: the device's pair count differs from the host's");
    ds.ndupwg = no_pairs ? 0 : build_report[0];
    if (pair_lpb == 64) {
        if (build_report[4] < 0 || (size_t)build_report[4] > pair_slot_cap || build_report[5] < 0 || build_report[7] < 0 || build_report[7] > build_report[4])
            return fail(SFMBA_ERR_HIP, "structure build: pair-chunk descriptors out of range");
        ds.npairwg = build_report[4]; ds.nmulti = build_report[5];
        // partial sums: one row per chunk of the blocks that HAVE several chunks (the build counted them), not one per descriptor slot
        if (!no_pairs) HIP_TRY(dev_alloc(&db.pair_partial, (size_t)36 * std::max((int)build_report[7], 1)));
    }
    // fill of the reduced matrix: non-empty off-diagonal blocks / all of them (what SFMBA_LINEAR_AUTO reads the co-visibility from)
    p->block_fill = ncam > 1 ? (double)std::max((int)build_report[1], 0) / ((double)ncam * (ncam - 1) / 2.0) : 1.0;
    p->block_band = build_report[1] > 0 ? (double)std::max((int)build_report[6], 0) / (double)build_report[1] : 0.0;
    if (no_pairs) { p->block_fill = 1.0; p->block_band = 0.0; }      // (unknown without the list)
    p->solver.blk_fill = p->block_fill;
    bt_mark("wait for device");
    return sfmba_problem_reset(p);
}

static int create_impl(int device, int precision, int flags, int n_cam, const double* cam6, const unsigned char* cam_active, int n_pt, const double* pt3,
                       int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                       double focal, int rank, int world, sfmba_problem** out) {
    if (!out) return fail(SFMBA_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_cam < 0 || n_pt < 0 || n_obs < 0 || n_obs >= (int64_t)1 << 31) return fail(SFMBA_ERR_INVALID_ARG, "bad sizes");
    if (precision != SFMBA_PRECISION_F64 && precision != SFMBA_PRECISION_F32J) return fail(SFMBA_ERR_INVALID_ARG, "bad precision");
    if ((n_cam > 0 && !cam6) || (n_pt > 0 && !pt3) || (n_obs > 0 && (!obs_cam || !obs_pt || !obs_xy)))
        return fail(SFMBA_ERR_INVALID_ARG, "NULL array");
    int rc = check_device(device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));

    sfmba_problem* p = new sfmba_problem();
    p->device = device;
    p->arena.set_device(device);
    p->precision = precision;
    p->n_cam_full = n_cam; p->n_pt_full = n_pt; p->n_obs = n_obs;
    p->focal0 = p->focal = focal;
    p->shard_rank = rank; p->shard_world = world;
    p->no_pairs = (flags & SFMBA_CREATE_NO_PAIR_LIST) != 0;
    p->row_sharded = (flags & SFMBA_CREATE_ROW_SHARDED) != 0;
    p->sharded = cam_active != nullptr || p->row_sharded;
    // create flag, or the environment override (kept across appends: the structure is rebuilt in the same mode)
    { const char* e = std::getenv("SFMBA_DETERMINISTIC"); p->deterministic = e ? e[0] == '1' : (flags & SFMBA_CREATE_DETERMINISTIC) != 0; }
    { const char* e = std::getenv("SFMBA_ROCTX"); p->roctx = e && e[0] == '1'; }
    struct Guard { sfmba_problem* p; ~Guard() { if (p) sfmba_problem_destroy(p); } } guard{ p };

    // active (observed) cameras / points -> slots, ascending caller index
    p->cam_slot.assign((size_t)n_cam, -1); p->pt_slot.assign((size_t)n_pt, -1);
    {
        std::atomic<bool> bad(false);
        int* cs = p->cam_slot.data();
        int* ps = p->pt_slot.data();
        parallel_for((int)n_obs, [&](int k0, int k1) {
            for (int k = k0; k < k1; ++k) {
                if (obs_cam[k] < 0 || obs_cam[k] >= n_cam || obs_pt[k] < 0 || obs_pt[k] >= n_pt) { bad.store(true, std::memory_order_relaxed); return; }
                // (several threads may mark the same slot; test first: unconditional stores make the 200 camera entries' cache lines
                // bounce between the cores -- 45 ms for 10^6 observations when tried)
                if (__atomic_load_n(cs + obs_cam[k], __ATOMIC_RELAXED) != 0) __atomic_store_n(cs + obs_cam[k], 0, __ATOMIC_RELAXED);
                if (__atomic_load_n(ps + obs_pt[k], __ATOMIC_RELAXED) != 0) __atomic_store_n(ps + obs_pt[k], 0, __ATOMIC_RELAXED);
            }
        });
        if (bad.load()) return fail(SFMBA_ERR_INVALID_ARG, "observation index out of range");
    }
    if (cam_active)   // sharded: every globally observed camera is part of every rank's reduced system
        for (int j = 0; j < n_cam; ++j) if (cam_active[j]) p->cam_slot[j] = 0;
    for (int j = 0; j < n_cam; ++j) if (p->cam_slot[j] == 0) { p->cam_slot[j] = (int)p->acam_id.size(); p->acam_id.push_back(j); }
    for (int i = 0; i < n_pt; ++i) if (p->pt_slot[i] == 0) { p->pt_slot[i] = (int)p->apt_id.size(); p->apt_id.push_back(i); }
    static_assert(sizeof(LMState) <= 1024, "LMState must fit its slice of the pinned block");
    if (!hostkit_acquire(device, &p->kit)) return fail(SFMBA_ERR_HIP, "stream / pinned memory creation failed");
    p->stream = p->kit.stream;
    p->arena.set_ordering_stream(p->stream);          // zeroing of its chunks: ordered in front of the stream's work, not waited for
    p->h_state = reinterpret_cast<LMState*>(p->kit.pinned);                        // [0, 1024)
    p->h_lm_mail = reinterpret_cast<volatile int*>(p->kit.pinned + 1024);          // [1024, 1088)
    if (n_obs == 0 && !p->sharded) {
        p->empty = true;
        guard.p = nullptr;
        *out = p;
        return SFMBA_OK;
    }
    ObsSource src;
    src.n_new = (int)n_obs; src.cam = obs_cam; src.pt = obs_pt; src.xy = obs_xy;
    rc = build_structure(p, src, cam6, pt3, focal, p->sharded);
    if (rc) return rc;
    guard.p = nullptr;
    *out = p;
    return SFMBA_OK;
}

int sfmba_problem_append(sfmba_problem* p, int n_cam, const double* cam6, int n_pt, const double* pt3,
                         int64_t n_obs_new, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy, double focal) {
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    if (p->sharded) return fail(SFMBA_ERR_INVALID_ARG, "a sharded problem cannot grow in place");
    if (p->no_pairs) return fail(SFMBA_ERR_INVALID_ARG, "a problem without a pair list cannot grow in place: create it anew");
    if (n_cam < p->n_cam_full || n_pt < p->n_pt_full || n_obs_new < 0 || p->n_obs + n_obs_new >= (int64_t)1 << 31)
        return fail(SFMBA_ERR_INVALID_ARG, "bad sizes: cameras and points can only be added at the end");
    if ((n_cam > 0 && !cam6) || (n_pt > 0 && !pt3) || (n_obs_new > 0 && (!obs_cam || !obs_pt || !obs_xy)))
        return fail(SFMBA_ERR_INVALID_ARG, "NULL array");
    HIP_TRY(hipSetDevice(p->device));
    const bool at_on = std::getenv("SFMBA_BUILD_TIMING") != nullptr;
    auto at_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double at_t = at_now();
    auto at_mark = [&](const char* what) { if (at_on) { const double t = at_now(); std::fprintf(stderr, "[sfmba append] %-18s %.3f ms\n", what, 1e3 * (t - at_t)); at_t = at_now(); } };
    // (the observations of an added view are scattered over the points: the three loops over them cost their cache misses.  The
    // check pulls the slot entries in, the slot loop the counters of build_structure's loop.)
    p->cam_slot.resize((size_t)n_cam, -1); p->pt_slot.resize((size_t)n_pt, -1);
    for (int64_t k = 0; k < n_obs_new; ++k) {
        if (obs_cam[k] < 0 || obs_cam[k] >= n_cam || obs_pt[k] < 0 || obs_pt[k] >= n_pt)
            return fail(SFMBA_ERR_INVALID_ARG, "observation index out of range");
        __builtin_prefetch(&p->pt_slot[(size_t)obs_pt[k]], 1);
    }
    at_mark("check indices");
    HIP_TRY(hipStreamSynchronize(p->stream));
    at_mark("sync (idle stream)");
    // (argument errors are behind us: from here on the problem is being replaced)
    p->reset_pending = false;           // the parameters are replaced by the caller's below
    // cameras / points that become observed get the next free slot (slot order = order of first observation)
    p->h_pt_cnt.reserve((size_t)n_pt);
    for (int64_t k = 0; k < n_obs_new; ++k) {
        int& cs = p->cam_slot[(size_t)obs_cam[k]];
        if (cs < 0) { cs = (int)p->acam_id.size(); p->acam_id.push_back(obs_cam[k]); }
        int& ps = p->pt_slot[(size_t)obs_pt[k]];
        if (ps < 0) { ps = (int)p->apt_id.size(); p->apt_id.push_back(obs_pt[k]); }
        else if ((size_t)ps < p->h_pt_cnt.size()) __builtin_prefetch(&p->h_pt_cnt[(size_t)ps], 1);
    }
    p->n_cam_full = n_cam; p->n_pt_full = n_pt;
    ObsSource src;
    src.n_old = p->empty ? 0 : p->ds.nobs;
    src.d_old_pt = p->d_obs_pt; src.d_old_cam = p->d_obs_cam; src.d_old_xy = p->d_obs_xy; src.d_old_perm = p->d_perm;
    src.n_new = (int)n_obs_new; src.cam = obs_cam; src.pt = obs_pt; src.xy = obs_xy;
    p->n_obs += n_obs_new;
    if (p->n_obs == 0) { p->focal0 = p->focal = focal; return SFMBA_OK; }
    p->empty = false;
    // the new structure is built in a fresh arena (reading the old point-major arrays); the old arena then goes back to the chunk
    // cache, where the next append finds it: steady state performs no hipMalloc / hipFree
    DeviceArena old(p->device);
    old.swap(p->arena);
    if (p->db.trace && !p->trace_mapped) { (void)hipFree(p->db.trace); }
    p->db.trace = nullptr; p->db.trace_cap = 0; p->trace_mapped = false;
    p->cur = 0;
    at_mark("slots, arena swap");
    const int rc = build_structure(p, src, cam6, pt3, focal, false);
    at_mark("build_structure");
    // not failure-atomic (the old structure is gone, counts and slot tables are already the new ones): a failed build leaves the
    // handle POISONED -- every entry point refuses it from here on, only sfmba_problem_destroy is valid (include/sfmba.h)
    if (rc != SFMBA_OK) { p->poisoned = true; (void)hipStreamSynchronize(p->stream); return rc; }
    if (hipStreamSynchronize(p->stream) != hipSuccess) { p->poisoned = true; return fail(SFMBA_ERR_HIP, "structure build did not complete"); }
    at_mark("final sync");
    return rc;      // `old` releases the previous structure here
}

// the device side of a reset, for the callers that are not a solve
static int flush_reset(sfmba_problem* p) {
    if (!p || !p->reset_pending) return SFMBA_OK;
    p->reset_pending = false;
    if (p->empty) return SFMBA_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(p->db.cam[0], p->d_cam0, sizeof(double) * 6 * (size_t)p->ds.ncam, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->db.pts[0], p->d_pts0, sizeof(double) * 3 * (size_t)p->ds.npt, hipMemcpyDeviceToDevice, p->stream));
    sfmba_options o;
    sfmba_options_default(&o);
    LMState st;
    init_state(p, st, o);
    return upload_state(p, st);
}

int sfmba_problem_reset(sfmba_problem* p) {
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    p->focal = p->focal0;
    p->cur = 0;
    // nothing is enqueued here: the next solve's first kernel copies the initial parameters itself (k_begin); any other entry point
    // that looks at the parameters flushes the reset first
    p->reset_pending = !p->empty;
    return SFMBA_OK;
}

int sfmba_problem_set_params(sfmba_problem* p, const double* cam6, const double* pt3, double focal) {
    if (!p || !cam6 || !pt3) return fail(SFMBA_ERR_INVALID_ARG, "NULL argument");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    p->reset_pending = false;           // everything a reset would restore is overwritten here
    p->focal = focal;
    p->cur = 0;
    if (p->empty) return SFMBA_OK;
    HIP_TRY(hipSetDevice(p->device));
    std::vector<double> cam((size_t)6 * p->ds.ncam), pts((size_t)3 * p->ds.npt);
    for (int j = 0; j < p->ds.ncam; ++j) std::memcpy(&cam[6 * (size_t)j], cam6 + 6 * (size_t)p->acam_id[j], 6 * sizeof(double));
    for (int i = 0; i < p->ds.npt; ++i) std::memcpy(&pts[3 * (size_t)i], pt3 + 3 * (size_t)p->apt_id[i], 3 * sizeof(double));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(p->db.cam[0], cam.data(), sizeof(double) * cam.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->db.pts[0], pts.data(), sizeof(double) * pts.size(), hipMemcpyHostToDevice));
    sfmba_options o;
    sfmba_options_default(&o);
    LMState st;
    init_state(p, st, o);
    return upload_state(p, st);
}

int sfmba_problem_get_params(sfmba_problem* p, double* cam6, double* pt3, double* focal) {
    if (p && !p->poisoned) { const int frc = flush_reset(p); if (frc) return frc; }
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    if (focal) *focal = p->focal;
    if (p->empty) return SFMBA_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    // slot == caller index everywhere (the usual case): straight into the caller's arrays, no gathered copy
    if (cam6) {
        if (p->cam_identity) HIP_TRY(hipMemcpy(cam6, p->db.cam[p->cur], sizeof(double) * 6 * (size_t)p->ds.ncam, hipMemcpyDeviceToHost));
        else {
            std::vector<double> cam((size_t)6 * p->ds.ncam);
            HIP_TRY(hipMemcpy(cam.data(), p->db.cam[p->cur], sizeof(double) * cam.size(), hipMemcpyDeviceToHost));
            for (int j = 0; j < p->ds.ncam; ++j) std::memcpy(cam6 + 6 * (size_t)p->acam_id[j], &cam[6 * (size_t)j], 6 * sizeof(double));
        }
    }
    if (pt3) {
        if (p->pt_identity) HIP_TRY(hipMemcpy(pt3, p->db.pts[p->cur], sizeof(double) * 3 * (size_t)p->ds.npt, hipMemcpyDeviceToHost));
        else {
            std::vector<double> pts((size_t)3 * p->ds.npt);
            HIP_TRY(hipMemcpy(pts.data(), p->db.pts[p->cur], sizeof(double) * pts.size(), hipMemcpyDeviceToHost));
            for (int i = 0; i < p->ds.npt; ++i) std::memcpy(pt3 + 3 * (size_t)p->apt_id[i], &pts[3 * (size_t)i], 3 * sizeof(double));
        }
    }
    return SFMBA_OK;
}

static int solve_matrix_free(sfmba_problem* p, const sfmba_options& o, sfmba_summary* summary, sfmba_iteration* trace, int trace_cap, int* trace_len);

int sfmba_problem_solve(sfmba_problem* p, const sfmba_options* opt, sfmba_summary* summary,
                        sfmba_iteration* trace, int trace_cap, int* trace_len) {
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    if (p->row_sharded) return fail(SFMBA_ERR_INVALID_ARG, "a row-sharded problem is solved with sfmba_problem_solve_sharded only");
    sfmba_options o;
    if (opt) o = *opt; else sfmba_options_default(&o);
    if (p->no_pairs && !p->empty) return solve_matrix_free(p, o, summary, trace, trace_cap, trace_len);
    if (p->precision == SFMBA_PRECISION_F32J) return run_solve<float>(p, o, summary, trace, trace_cap, trace_len);
    return run_solve<double>(p, o, summary, trace, trace_cap, trace_len);
}

void* sfmba_problem_stream(sfmba_problem* p) { return p ? (void*)p->stream : nullptr; }

int sfmba_problem_reduced_dim(const sfmba_problem* p) { return p && !p->empty ? p->ds.d : 0; }

int sfmba_solve(int n_cam, double* cam6, int n_pt, double* pt3, int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt,
                const double* obs_xy, double* focal, const sfmba_options* opt, sfmba_summary* summary,
                sfmba_iteration* trace, int trace_cap, int* trace_len) {
    if (!focal) return fail(SFMBA_ERR_INVALID_ARG, "focal is NULL");
    sfmba_options o;
    if (opt) o = *opt; else sfmba_options_default(&o);
    const double t0 = now_seconds();
    sfmba_problem* p = nullptr;
    int rc = sfmba_problem_create(0, o.precision, n_cam, cam6, n_pt, pt3, n_obs, obs_cam, obs_pt, obs_xy, *focal, &p);
    if (rc) return rc;
    const double setup = now_seconds() - t0;
    sfmba_summary sum;
    std::memset(&sum, 0, sizeof(sum));
    rc = sfmba_problem_solve(p, &o, &sum, trace, trace_cap, trace_len);
    // Ceres leaves the user's parameter blocks untouched when the solve terminates with FAILURE (solver.cc: "do not update
    // user state" [Ceres-upstream]); every other termination writes the best accepted point back
    if (rc == SFMBA_OK && sum.termination != SFMBA_FAILURE) rc = sfmba_problem_get_params(p, cam6, pt3, focal);
    sum.setup_seconds = setup;
    if (summary && rc == SFMBA_OK) *summary = sum;
    sfmba_problem_destroy(p);
    return rc;
}

int sfmba_problem_set_profiling(sfmba_problem* p, int enable) {
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    p->prof.reset();
    p->prof.on = enable != 0;
    return SFMBA_OK;
}

int sfmba_problem_get_profile(sfmba_problem* p, sfmba_kernel_time* out, int cap, int* n) {
    if (!p || !n) return fail(SFMBA_ERR_INVALID_ARG, "NULL argument");
    int k = 0;
    for (int id = 0; id < KID_COUNT; ++id) {
        if (p->prof.count[id] == 0) continue;
        if (out && k < cap) {
            std::snprintf(out[k].name, sizeof(out[k].name), "%s", kernel_name(id));
            out[k].total_us = 1e3 * p->prof.total_ms[id];
            out[k].launches = p->prof.count[id];
        }
        ++k;
    }
    *n = k;
    return SFMBA_OK;
}

// ---- kernel-level entry points ----------------------------------------------------------------
int sfmba_problem_eval_residuals(sfmba_problem* p, double* residuals_out, double* cost_out) {
    if (p && !p->poisoned) { const int frc = flush_reset(p); if (frc) return frc; }
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    if (cost_out) *cost_out = 0.0;
    if (p->empty) return SFMBA_OK;
    HIP_TRY(hipSetDevice(p->device));
    double *d_res = nullptr, *d_cost = nullptr;
    HIP_TRY(dev_alloc(&d_res, (size_t)2 * p->ds.nobs));
    HIP_TRY(dev_alloc(&d_cost, 1));
    HIP_TRY(hipMemsetAsync(d_cost, 0, sizeof(double), p->stream));
    const size_t n = 6 * (size_t)p->ds.ncam;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, p->db.cscale, n, 1.0);
    launch_cam_setup<double>(p->stream, p->ds, p->db, p->cur);
    if (p->precision == SFMBA_PRECISION_F32J) launch_eval_residuals<float>(p->stream, p->ds, p->db, p->d_obs_pt, d_res, d_cost);
    else launch_eval_residuals<double>(p->stream, p->ds, p->db, p->d_obs_pt, d_res, d_cost);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (residuals_out) HIP_TRY(hipMemcpy(residuals_out, d_res, sizeof(double) * 2 * (size_t)p->ds.nobs, hipMemcpyDeviceToHost));
    if (cost_out) HIP_TRY(hipMemcpy(cost_out, d_cost, sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d_res); (void)hipFree(d_cost);
    return SFMBA_OK;
}

int sfmba_problem_eval_jacobian(sfmba_problem* p, double* jc, double* jp, double* jf) {
    if (p && !p->poisoned) { const int frc = flush_reset(p); if (frc) return frc; }
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    if (p->empty) return SFMBA_OK;
    HIP_TRY(hipSetDevice(p->device));
    const size_t n = (size_t)p->ds.nobs;
    double *d_jc = nullptr, *d_jp = nullptr, *d_jf = nullptr;
    if (jc) HIP_TRY(dev_alloc(&d_jc, 12 * n));
    if (jp) HIP_TRY(dev_alloc(&d_jp, 6 * n));
    if (jf) HIP_TRY(dev_alloc(&d_jf, 2 * n));
    const size_t nc = 6 * (size_t)p->ds.ncam;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, p->stream, p->db.cscale, nc, 1.0);
    launch_cam_setup<double>(p->stream, p->ds, p->db, p->cur);
    if (p->precision == SFMBA_PRECISION_F32J) launch_eval_jacobian<float>(p->stream, p->ds, p->db, p->d_obs_pt, p->d_perm, d_jc, d_jp, d_jf);
    else launch_eval_jacobian<double>(p->stream, p->ds, p->db, p->d_obs_pt, p->d_perm, d_jc, d_jp, d_jf);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (jc) HIP_TRY(hipMemcpy(jc, d_jc, sizeof(double) * 12 * n, hipMemcpyDeviceToHost));
    if (jp) HIP_TRY(hipMemcpy(jp, d_jp, sizeof(double) * 6 * n, hipMemcpyDeviceToHost));
    if (jf) HIP_TRY(hipMemcpy(jf, d_jf, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    if (d_jc) (void)hipFree(d_jc);
    if (d_jp) (void)hipFree(d_jp);
    if (d_jf) (void)hipFree(d_jf);
    return SFMBA_OK;
}

int sfmba_problem_build_reduced(sfmba_problem* p, const sfmba_options* opt, double radius, double* S, double* rhs, double* scale) {
    if (p && !p->poisoned) { const int frc = flush_reset(p); if (frc) return frc; }
    if (!p || p->empty || p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, p && p->poisoned ? "poisoned problem (a failed sfmba_problem_append): destroy it" : "NULL or empty problem");
    if (p->row_sharded) return fail(SFMBA_ERR_INVALID_ARG, "a row-sharded problem is solved with sfmba_problem_solve_sharded only");
    if (p->no_pairs) return fail(SFMBA_ERR_INVALID_ARG, "this problem has no pair list (SFMBA_CREATE_NO_PAIR_LIST / too many pairs): its reduced matrix is never formed");
    sfmba_options o;
    if (opt) o = *opt; else sfmba_options_default(&o);
    HIP_TRY(hipSetDevice(p->device));
    LMState st;
    init_state(p, st, o);
    st.radius = radius;
    int rc = upload_state(p, st);
    if (rc) return rc;
    rc = ensure_trace(p, 4);
    if (rc) return rc;
    if (p->precision == SFMBA_PRECISION_F32J) {
        launch_linearise_setup<float>(p, o.jacobi_scaling);
        launch_point_build<float>(p->stream, p->ds, p->db);
        launch_cam_diag<float>(p->stream, p->ds, p->db);
        launch_schur_pairs<float>(p->stream, p->ds, p->db, 2);
        launch_schur_pairs<float>(p->stream, p->ds, p->db, 0);
    } else {
        launch_linearise_setup<double>(p, o.jacobi_scaling);
        launch_point_build<double>(p->stream, p->ds, p->db);
        launch_cam_diag<double>(p->stream, p->ds, p->db);
        launch_schur_pairs<double>(p->stream, p->ds, p->db, 2);
        launch_schur_pairs<double>(p->stream, p->ds, p->db, 0);
    }
    launch_finalize(p->stream, p->ds, p->db, 0);
    const int d = p->ds.d;
    double *d_full = nullptr, *d_scale = nullptr;
    HIP_TRY(dev_alloc(&d_full, (size_t)d * d));
    HIP_TRY(dev_alloc(&d_scale, (size_t)d));
    launch_mirror_scale(p->stream, p->ds, p->db, d_full, d_scale);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (S) HIP_TRY(hipMemcpy(S, d_full, sizeof(double) * (size_t)d * d, hipMemcpyDeviceToHost));
    if (rhs) HIP_TRY(hipMemcpy(rhs, p->db.rhs, sizeof(double) * (size_t)d, hipMemcpyDeviceToHost));
    if (scale) HIP_TRY(hipMemcpy(scale, d_scale, sizeof(double) * (size_t)d, hipMemcpyDeviceToHost));
    (void)hipFree(d_full); (void)hipFree(d_scale);
    return SFMBA_OK;
}

int sfmba_dense_spd_solve(int device, int n, const double* A, const double* b, double* x, int method,
                          double pcg_tol, int pcg_max_iters, int* info, int* iters) {
    if (n <= 0 || !A || !b || !x) return fail(SFMBA_ERR_INVALID_ARG, "bad arguments");
    int rc = check_device(device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    const int ld = dense_padded_dim(n);
    std::vector<double> Ap((size_t)ld * ld, 0.0), bp((size_t)ld, 0.0);
    for (int r = 0; r < n; ++r) for (int c = r; c < n; ++c) Ap[(size_t)r * ld + c] = A[(size_t)r * n + c];
    for (int e = n; e < ld; ++e) Ap[(size_t)e * ld + e] = 1.0;
    std::memcpy(bp.data(), b, sizeof(double) * (size_t)n);
    double *dA = nullptr, *db_ = nullptr;
    int* dinfo = nullptr;
    HIP_TRY(dev_upload(&dA, Ap));
    HIP_TRY(dev_upload(&db_, bp));
    HIP_TRY(dev_alloc(&dinfo, 1));
    HIP_TRY(hipMemset(dinfo, 0, sizeof(int)));
    DenseSolver ws;
    if (dense_solver_create(&ws, n, ld)) return fail(SFMBA_ERR_ALLOC, "dense solver workspace allocation failed");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int it = 0;
    if (method == SFMBA_LINEAR_PCG) it = dense_pcg_solve(s, &ws, dA, db_, pcg_tol > 0 ? pcg_tol : 1e-10, pcg_max_iters, dinfo);
    else dense_cholesky_solve(s, &ws, dA, db_, dinfo);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(x, db_, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    int hinfo = 0;
    HIP_TRY(hipMemcpy(&hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost));
    if (info) *info = hinfo;
    if (iters) *iters = it;
    dense_solver_destroy(&ws);
    (void)hipStreamDestroy(s);
    (void)hipFree(dA); (void)hipFree(db_); (void)hipFree(dinfo);
    return SFMBA_OK;
}

// ---- sharded (multi-GPU) API --------------------------------------------------------------------
// reduce_buf = [S (ld*ld) | rhs (ld) | udiag (ld) | bc (ld) | scalars]: one all-reduce(SUM) per LM iteration
// carries the partial reduced camera system, its right-hand side, the undamped diagonal, the scaled
// gradient and the linearisation scalars.  See include/sfmba.h for the protocol.
int64_t sfmba_shard_reduce_len(const sfmba_problem* p) {
    return p && !p->empty ? (int64_t)p->ds.ld * (p->ds.ld + 1) / 2 + 3 * (int64_t)p->ds.ld + SFMBA_SHARD_SCALARS : 0;
}
void* sfmba_shard_reduce_buf(sfmba_problem* p) { return p ? (void*)p->d_red : nullptr; }
int64_t sfmba_shard_setup_len(const sfmba_problem* p) { return p && !p->empty ? 2 * (int64_t)p->ds.ld + SFMBA_SHARD_SCALARS : 0; }
void* sfmba_shard_setup_buf(sfmba_problem* p) { return p ? (void*)p->db.udiag : nullptr; }
void* sfmba_shard_scalars_buf(sfmba_problem* p) { return p ? (void*)p->d_scal : nullptr; }

// fused = true (the C loop of sfmba_problem_solve_sharded, CG path): one k_begin launch carries the LM state and clears / builds what
// five launches and two copies do otherwise, and the point scales are left to the first k_point_build, as in run_solve
static int shard_begin_impl(sfmba_problem* p, const sfmba_options* opt, bool fused) {
    if (!p || p->empty || p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, p && p->poisoned ? "poisoned problem (a failed sfmba_problem_append): destroy it" : "NULL or empty problem");
    if (opt) p->shard_opt = *opt; else sfmba_options_default(&p->shard_opt);
    HIP_TRY(hipSetDevice(p->device));
    int rc = ensure_trace(p, std::min(std::max(p->shard_opt.max_iters, 0) + 2, 1 << 16));
    if (rc) return rc;
    if (!fused) HIP_TRY(hipStreamSynchronize(p->stream));
    p->shard_t0 = now_seconds();
    p->shard_active = true;
    p->shard_host_iter = 0;
    std::memset(&p->shard_sum, 0, sizeof(p->shard_sum));
    p->db.shared_weight = p->shard_rank == 0 ? 1.0 : 0.0;
    LMState st;
    init_state(p, st, p->shard_opt);
    const int f32 = p->precision == SFMBA_PRECISION_F32J;
    if (p->row_sharded && !fused) return fail(SFMBA_ERR_INVALID_ARG, "a row-sharded problem runs the loop of sfmba_problem_solve_sharded only");
    if (fused) {
        *p->h_state = st;
        launch_begin(p->stream, p->ds, p->db, st, p->reset_pending ? p->d_cam0 : nullptr, p->reset_pending ? p->d_pts0 : nullptr);
        p->reset_pending = false;
        launch_xnorm(p->stream, ds_points(p), p->db);
        launch_colnorm_cams_only(p->stream, ds_cams(p), p->db, p->shard_opt.jacobi_scaling, f32, /*clear_udiag=*/false);
    } else {
        if ((rc = flush_reset(p))) return rc;
        rc = upload_state(p, st);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(p->d_info, 0, sizeof(int), p->stream));
        const size_t n = 6 * (size_t)p->ds.ncam;
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, p->db.cscale, n, 1.0);
        launch_clear_slots(p->stream, p->db);
        launch_cam_setup<double>(p->stream, p->ds, p->db, p->cur);
        launch_xnorm(p->stream, p->ds, p->db);
        launch_colnorm_points_only(p->stream, p->ds, p->db, p->shard_opt.jacobi_scaling, f32);
        launch_colnorm_cams_only(p->stream, p->ds, p->db, p->shard_opt.jacobi_scaling, f32);
    }
    HIP_TRY(hipMemsetAsync(p->db.bc, 0, sizeof(double) * p->ds.ld, p->stream));
    launch_shard_pack(p->stream, p->db, p->d_scal, 0, p->shard_rank);
    return SFMBA_OK;
}
int sfmba_shard_begin(sfmba_problem* p, const sfmba_options* opt) { return shard_begin_impl(p, opt, false); }

int sfmba_shard_setup_finish(sfmba_problem* p) {
    if (!p || !p->shard_active) return fail(SFMBA_ERR_INVALID_ARG, "shard_begin was not called");
    HIP_TRY(hipSetDevice(p->device));
    launch_shard_unpack(p->stream, p->db, p->d_scal, 0, p->shard_world);
    launch_shard_xnorm_finish(p->stream, p->db);
    launch_colnorm_finish(p->stream, p->ds, p->db, p->shard_opt.jacobi_scaling);
    launch_cam_setup<double>(p->stream, p->ds, p->db, p->cur);
    return SFMBA_OK;
}

int sfmba_shard_partial_build(sfmba_problem* p) {
    if (!p || !p->shard_active) return fail(SFMBA_ERR_INVALID_ARG, "shard_begin was not called");
    HIP_TRY(hipSetDevice(p->device));
    if (p->precision == SFMBA_PRECISION_F32J) {
        launch_point_build<float>(p->stream, p->ds, p->db);
        launch_cam_diag<float>(p->stream, p->ds, p->db);
        launch_schur_pairs<float>(p->stream, p->ds, p->db, 2);
        launch_schur_pairs<float>(p->stream, p->ds, p->db, 0);
    } else {
        launch_point_build<double>(p->stream, p->ds, p->db);
        launch_cam_diag<double>(p->stream, p->ds, p->db);
        launch_schur_pairs<double>(p->stream, p->ds, p->db, 2);
        launch_schur_pairs<double>(p->stream, p->ds, p->db, 0);
    }
    launch_cd_fold(p->stream, p->ds, p->db);       // deterministic mode: chunk sums in chunk order, before the exchange
    launch_shard_pack(p->stream, p->db, p->d_scal, 1, p->shard_rank);
    launch_shard_tri(p->stream, p->d_sys, p->d_red, p->ds.ld, 3 * (long long)p->ds.ld + SFMBA_SHARD_SCALARS, /*unpack=*/false);
    return SFMBA_OK;
}

int sfmba_shard_solve_update(sfmba_problem* p) {
    if (!p || !p->shard_active) return fail(SFMBA_ERR_INVALID_ARG, "shard_begin was not called");
    HIP_TRY(hipSetDevice(p->device));
    const sfmba_options& o = p->shard_opt;
    launch_shard_tri(p->stream, p->d_sys, p->d_red, p->ds.ld, 3 * (long long)p->ds.ld + SFMBA_SHARD_SCALARS, /*unpack=*/true);
    launch_shard_unpack(p->stream, p->db, p->d_scal, 1, p->shard_world);
    { DeviceBuffers dbf = p->db; dbf.cd_part = nullptr; launch_finalize(p->stream, p->ds, dbf, 0); }     // (chunk sums: folded before the exchange)
    DeviceBuffers dbu = p->db;
    if (p->precision == SFMBA_PRECISION_F32J) dbu.pu32 = p->d_pu32;      // (the gradient . step term is added by rank 0 only: shared_weight)
    if (o.linear_solver == SFMBA_LINEAR_PCG || (o.linear_solver == SFMBA_LINEAR_AUTO && p->ds.d > 256)) {
        // fp32 Jacobian mode: the streaming CG path keeps the preconditioned matrix in fp32 (k_pcg_transform writes it)
        p->solver.use_f32 = p->precision == SFMBA_PRECISION_F32J && dense_pcg_want_f32(&p->solver) != nullptr;
        p->solver.symmetric = false; p->db.pcg_upper_only = 0; p->db.pcg_zero = nullptr; p->db.pcg_zero_n = 0;      // (the step-wise sharded protocol keeps both triangles)
        // block factors and S~ from the all-reduced system, then the gauge vectors from those factors (two-level preconditioner)
        if (dense_pcg_transform(p->stream, &p->solver, p->db.S, p->db.rhs, p->d_info, nullptr)) return fail(SFMBA_ERR_ALLOC, "PCG workspace allocation failed");
        const bool coarse_cg = option_switch(o.pcg_coarse_space, "SFMBA_PCG_COARSE", true);
        if (coarse_cg) { p->db.pcg_W = p->solver.W; launch_gauge(p->stream, p->ds, p->db); }
        const bool exact_pcg = o.linear_solver == SFMBA_LINEAR_AUTO;
        const double cg_tol = exact_pcg ? std::min(o.pcg_tolerance > 0.0 ? o.pcg_tolerance : auto_cg_tol(), auto_cg_tol()) : o.pcg_tolerance;
        const int it = dense_pcg_solve(p->stream, &p->solver, p->db.S, p->db.rhs, cg_tol, o.pcg_max_iters, p->d_info, nullptr,
                                       false, p->shard_host_iter, /*pretransformed=*/true, /*anchor=*/(!o.pcg_anchored || exact_pcg) ? 0 : p->shard_host_iter == 0 ? 1 : 2,
                                       /*no_wait=*/false, /*coarse=*/coarse_cg);
        if (it < 0) return fail(SFMBA_ERR_ALLOC, "PCG workspace allocation failed");
        p->shard_sum.linear_iters += it;
        dbu.pcg_vec = p->solver.vec; dbu.pcg_linv = p->solver.binv; dbu.pcg_flags = p->solver.flags;
    } else {
        dense_cholesky_solve(p->stream, &p->solver, p->db.S, p->db.rhs, p->d_info, nullptr);
    }
    if (p->precision == SFMBA_PRECISION_F32J) launch_back_substitution<float>(p->stream, p->ds, p->ds, dbu, nullptr);      // (one DeviceBuffers for both: sum u . r is formed in ONE of the two)
    else launch_back_substitution<double>(p->stream, p->ds, p->ds, dbu, nullptr);
    launch_shard_pack(p->stream, p->db, p->d_scal, 2, p->shard_rank);
    return SFMBA_OK;
}

int sfmba_shard_finish(sfmba_problem* p, int* done) {
    if (!p || !p->shard_active || !done) return fail(SFMBA_ERR_INVALID_ARG, "shard_begin was not called");
    HIP_TRY(hipSetDevice(p->device));
    const sfmba_options& o = p->shard_opt;
    launch_shard_unpack(p->stream, p->db, p->d_scal, 2, p->shard_world);
    launch_control(p->stream, p->ds, p->db);
    int rc = download_state(p);
    if (rc) return rc;
    p->shard_host_iter = p->h_state->iter;
    *done = 0;
    if (p->h_state->termination != -1) {
        p->shard_sum.termination = p->h_state->termination;
        std::snprintf(p->shard_sum.message, sizeof(p->shard_sum.message), "%s", message_text(p->h_state->message));
        *done = 1;
    } else if (p->shard_host_iter >= o.max_iters) {
        p->shard_sum.termination = SFMBA_NO_CONVERGENCE;
        std::snprintf(p->shard_sum.message, sizeof(p->shard_sum.message), "%s", message_text(MSG_MAX_ITERS));
        *done = 1;
    }
    // the wall-clock limit is deliberately not applied here: ranks would disagree on it
    return SFMBA_OK;
}

int sfmba_shard_end(sfmba_problem* p, sfmba_summary* summary) {
    if (!p || !p->shard_active) return fail(SFMBA_ERR_INVALID_ARG, "shard_begin was not called");
    HIP_TRY(hipSetDevice(p->device));
    int rc = download_state(p);
    if (rc) return rc;
    const LMState& hs = *p->h_state;
    p->cur = hs.cur;
    p->focal = hs.focal[hs.cur];
    p->db.shared_weight = 1.0;
    p->shard_active = false;
    sfmba_summary& sum = p->shard_sum;
    sum.iterations = hs.iter;
    sum.successful_steps = hs.successful;
    sum.unsuccessful_steps = hs.unsuccessful;
    sum.residual_evals = hs.residual_evals;
    sum.jacobian_evals = hs.jacobian_evals;
    sum.final_cost = hs.cost;
    sum.seconds = now_seconds() - p->shard_t0;
    TraceRow row0;
    if (p->trace_mapped) std::memcpy(&row0, p->kit.pinned + 4096, sizeof(TraceRow));
    else HIP_TRY(hipMemcpy(&row0, p->db.trace, sizeof(TraceRow), hipMemcpyDeviceToHost));
    sum.initial_cost = row0.cost;
    if (summary) *summary = sum;
    return SFMBA_OK;
}

// ---- the sharded LM loop in one call: collectives through a callback (RCCL below, or the caller's) ----
static int solve_sharded_impl(sfmba_problem* p, const sfmba_options* opt, sfmba_allreduce_fn allreduce, void* ctx, sfmba_summary* summary);

int sfmba_problem_solve_sharded(sfmba_problem* p, const sfmba_options* opt, sfmba_allreduce_fn allreduce, void* ctx, sfmba_summary* summary) {
    if (!p || p->empty || p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, p && p->poisoned ? "poisoned problem (a failed sfmba_problem_append): destroy it" : "NULL or empty problem");
    if (!p->sharded) return fail(SFMBA_ERR_INVALID_ARG, "not a sharded problem (sfmba_problem_create_sharded)");
    return solve_sharded_impl(p, opt, allreduce, ctx, summary);
}

// The matrix-free solve of an UNSHARDED problem without a pair list (include/sfmba.h, SFMBA_CREATE_NO_PAIR_LIST): the CG loop of the sharded
// path on one rank with the reduced matrix applied implicitly (implicit_schur.hip) -- two passes over the observations per CG iteration,
// nothing of size (pairs) or (cameras^2) is ever formed.  The trace rows are the ones k_lm_control wrote.
static int solve_matrix_free(sfmba_problem* p, const sfmba_options& o, sfmba_summary* summary, sfmba_iteration* trace, int trace_cap, int* trace_len) {
    if (trace_len) *trace_len = 0;
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    sfmba_summary sum;
    const int rc = solve_sharded_impl(p, &o, nullptr, nullptr, &sum);
    if (rc) return rc;
    const int rows = std::min(sum.iterations + 1, p->db.trace_cap);
    if (rows > 0 && (trace || trace_len)) {
        std::vector<TraceRow> tr((size_t)rows);
        if (p->trace_mapped) std::memcpy(tr.data(), p->kit.pinned + 4096, sizeof(TraceRow) * (size_t)rows);
        else HIP_TRY(hipMemcpy(tr.data(), p->db.trace, sizeof(TraceRow) * (size_t)rows, hipMemcpyDeviceToHost));
        if (trace && trace_cap > 0) std::memcpy(trace, tr.data(), sizeof(TraceRow) * (size_t)std::min(rows, trace_cap));
        if (trace_len) *trace_len = trace ? std::min(rows, std::max(trace_cap, 0)) : rows;
    }
    if (summary) *summary = sum;
    return SFMBA_OK;
}

static int solve_sharded_impl(sfmba_problem* p, const sfmba_options* opt, sfmba_allreduce_fn allreduce, void* ctx, sfmba_summary* summary) {
    if (p->shard_world > 1 && !allreduce) return fail(SFMBA_ERR_INVALID_ARG, "world > 1 needs an all-reduce");
    const double t_shard0 = now_seconds();
    auto reduce = [&](void* buf, int64_t n) -> int {
        if (!allreduce) return SFMBA_OK;          // (a communicator of one rank is still called: the RCCL path is exercised on a one-GPU box)
        const int arc = allreduce(ctx, buf, n, (void*)p->stream);
        return arc == 0 ? SFMBA_OK : fail(SFMBA_ERR_HIP, "all-reduce failed (rc " + std::to_string(arc) + ")");
    };
    sfmba_options o_in;
    if (opt) o_in = *opt; else sfmba_options_default(&o_in);
    // (a row-sharded problem: always the CG loop -- a rank holds all observations, the single all-reduce of partial systems does not apply)
    const bool two_phase = p->row_sharded || p->no_pairs || ((o_in.linear_solver == SFMBA_LINEAR_PCG || (o_in.linear_solver == SFMBA_LINEAR_AUTO && p->ds.d > 256)) &&
                                              option_switch(o_in.shard_two_phase, "SFMBA_SHARD_TWO_PHASE", true));
    if (p->row_sharded && p->shard_world > 1 && !p->allgather) return fail(SFMBA_ERR_INVALID_ARG, "a row-sharded solve with world > 1 needs sfmba_problem_set_allgather");
    // fail-stop rule (include/sfmba.h): everything the loop allocates is allocated before this rank's first collective
    if (two_phase) {
        if (dense_pcg_ensure_workspace(&p->solver)) return fail(SFMBA_ERR_ALLOC, "PCG workspace allocation failed");
        if (p->precision == SFMBA_PRECISION_F32J) (void)dense_pcg_want_f32(&p->solver);     // (null = not applicable at this size)
    }
    int rc = shard_begin_impl(p, opt, /*fused=*/two_phase);
    if (rc) return rc;
    p->h_lm_mail[0] = 0; p->h_lm_mail[1] = -1;
    if ((rc = reduce(sfmba_shard_setup_buf(p), sfmba_shard_setup_len(p)))) return rc;
    if ((rc = sfmba_shard_setup_finish(p))) return rc;
    const sfmba_options& o = p->shard_opt;
    int controls = 0;
    if (two_phase) {
        // CG solver: two all-reduces per linearisation (diagonal blocks + vectors, then the off-diagonal blocks of the preconditioned
        // matrix, ba_kernels.hip k_shard_diag / k_shard_offdiag) and the same fused kernels as the one-GPU loop in run_solve:
        // block factors and gauge vectors in k_finalize, transform in the pair pass, gated CG batches (no host wait on the solve).
        const bool coarse_cg = option_switch(o.pcg_coarse_space, "SFMBA_PCG_COARSE", true);
        const bool f32 = p->precision == SFMBA_PRECISION_F32J;
        bool first_linear_solve = true;
        const bool speculate = option_switch(o.early_linearise, "SFMBA_EARLY_LINEARISE", true);
        bool build_enqueued = false;
        const bool exchange_f32_off = !option_switch(o.shard_f32_exchange, "SFMBA_SHARD_F32_EXCHANGE", true);
        // AUTO here = the CG run to a plain relative 1e-12 (no Cholesky fallback in the sharded loop: the factorisation would need the
        // unpreconditioned matrix exchanged as well; at max_iters the step is forced, as with PCG)
        const bool exact_pcg = o.linear_solver != SFMBA_LINEAR_PCG;
        const double cg_tol = exact_pcg ? std::min(o.pcg_tolerance > 0.0 ? o.pcg_tolerance : auto_cg_tol(), auto_cg_tol()) : o.pcg_tolerance;
        // the CG without the redundant solve (dist_cg.h): reduce-scatter of the blocks, products from the owned blocks, one small
        // all-reduce per CG iteration
        // ... or WITHOUT any exchange of the reduced matrix (shard_distributed_cg = 2): the product of a CG iteration is formed implicitly from
        // every rank's own points (implicit_schur.hip)
        int dist_mode = o.shard_distributed_cg > 0 ? o.shard_distributed_cg : 0;
        // ... or with the block ROWS of S~ sharded (shard_distributed_cg = 3): a property of the problem handle -- every rank was given the whole problem
        if (p->no_pairs) dist_mode = 2;          // no pair list: the implicit product is the only one there is
        else if (p->row_sharded) dist_mode = 3;
        else if (dist_mode == 3) return fail(SFMBA_ERR_INVALID_ARG, "shard_distributed_cg = 3 needs a problem created with SFMBA_CREATE_ROW_SHARDED");
        // (duplicate (camera, point) observations: the implicit product sums over ALL pairs of a point's observations, those of one camera
        // included, so in that form their cross terms are simply not added to the diagonal blocks -- which stay what they are for: a
        // preconditioner.  Nothing is decided from a rank's own duplicate count: every rank takes the form the options name)
        const bool implicit_cg = dist_mode == 2;
        const bool row_cg = dist_mode == 3;
        const bool dist_cg = dist_mode > 0 && (implicit_cg || row_cg || p->shard_world == 1 || p->reduce_scatter != nullptr);
        const DeviceStructure dsp = ds_points(p), dsc = ds_cams(p);
        const size_t pa_bytes = f32 ? sizeof(PtRecA<float>) : sizeof(PtRecA<double>), pb_bytes = f32 ? sizeof(PtRecB<float>) : sizeof(PtRecB<double>);
        auto gather = [&](void* buf, size_t bytes_per_rank) -> int {
            if (!p->allgather) return SFMBA_OK;       // (one rank; a communicator of one rank is still called)
            const int grc = p->allgather(ctx, buf, (int64_t)bytes_per_rank, (void*)p->stream);
            return grc == 0 ? SFMBA_OK : fail(SFMBA_ERR_HIP, "all-gather failed (rc " + std::to_string(grc) + ")");
        };
        struct ArCtx { sfmba_allreduce_fn fn; void* ctx; } arctx{ allreduce, ctx };
        auto ar_thunk = [](void* c, void* buf, long long n, hipStream_t st) -> int { ArCtx* a = static_cast<ArCtx*>(c); return a->fn ? a->fn(a->ctx, buf, (int64_t)n, (void*)st) : 0; };
        int dcg_launched = 0, dcg_max = o.pcg_max_iters > 0 ? o.pcg_max_iters : 4 * p->ds.d;
        if (dist_cg && !p->dcg.ready) {
            ArenaScope as(&p->arena);
            if (dcg_create(&p->dcg, p->ds.d, p->ds.ld, p->ds.ncam, p->shard_rank, p->shard_world, &p->arena)) return fail(SFMBA_ERR_ALLOC, "distributed CG workspace allocation failed");
        }
        if (implicit_cg && !p->imp_dtab) {
            ArenaScope as(&p->arena);
            p->imp_dtab = p->arena.alloc_n<double>((size_t)8 * std::max(p->ds.ncam, 1));
            p->imp_spt = p->arena.alloc_n<double>((size_t)3 * std::max(p->ds.npt, 1));
            p->imp_acc = p->arena.alloc_n<double>((size_t)6 * std::max(p->ds.ncam, 1));
            if (p->deterministic) p->imp_part = p->arena.alloc_n<double>((size_t)6 * std::max(p->ds.nchunk, 1));
            if (!p->imp_dtab || !p->imp_spt || !p->imp_acc || (p->deterministic && !p->imp_part)) { p->imp_dtab = nullptr; return fail(SFMBA_ERR_ALLOC, "implicit Schur workspace allocation failed"); }
        }
        ImplicitProduct ip;
        int first_build = o.jacobi_scaling ? 1 : 2;          // the first point pass also forms the point scales
        for (;;) {
            if (dense_pcg_ensure_workspace(&p->solver)) return fail(SFMBA_ERR_ALLOC, "PCG workspace allocation failed");
            p->db.pcg_F = p->solver.Sfull;
            p->db.pcg_W = coarse_cg ? p->solver.W : nullptr;
            float* F32 = (f32 && !implicit_cg) ? dense_pcg_want_f32(&p->solver) : nullptr;      // (implicit product: there is no matrix to store)
            p->solver.use_f32 = F32 != nullptr;
            p->solver.symmetric = false; p->db.pcg_upper_only = 0; p->db.pcg_zero = nullptr; p->db.pcg_zero_n = 0;      // (sharded solves keep both triangles of the replicated matrix)
            // the streaming CG path stores S~ in fp32: with a single-precision all-reduce the partial blocks are exchanged in fp32 and
            // the sum is the CG's matrix (what the camera pass and the pair epilogue write directly -- diagonal blocks, focal column --
            // goes to the fp32 matrix as on one GPU); otherwise everything is summed in fp64 and narrowed after the sum
            // (one rank without a communicator: nothing is exchanged, the blocks stay in the CG's own precision)
            const bool x32 = F32 != nullptr && !exchange_f32_off && (row_cg || (p->allreduce_f32 != nullptr && allreduce != nullptr) || (p->shard_world == 1 && allreduce == nullptr));
            p->db.pcg_F32 = x32 ? F32 : nullptr;
            p->shard_exchange[0] = 8 * shard_diag_len(p->ds); p->shard_exchange[1] = (x32 ? 4 : 8) * shard_offdiag_len(p->ds);
            p->shard_exchange[2] = 8 * SFMBA_SHARD_SCALARS; p->shard_exchange[3] = x32 ? 1 : 0;
            if (!build_enqueued) { if (f32) launch_point_build<float>(p->stream, dsp, p->db, first_build); else launch_point_build<double>(p->stream, dsp, p->db, first_build); }
            build_enqueued = false;
            if (row_cg) {
                // the per-point table of every rank's points: what the camera pass and the pair pass re-evaluate the observations from
                if ((rc = gather(p->db.PA, (size_t)p->own_pt_stride * pa_bytes))) return rc;
                if ((rc = gather(p->db.PB, (size_t)p->own_pt_stride * pb_bytes))) return rc;
            }
            { const DeviceBuffers dbc = db_cams(p); if (f32) launch_cam_diag<float>(p->stream, dsc, dbc); else launch_cam_diag<double>(p->stream, dsc, dbc); }
            // pairs inside diagonal blocks: part of the implicit product; a row-sharded rank holds ALL of them -- rank 0 adds them
            if (!implicit_cg && (!row_cg || p->shard_rank == 0)) { if (f32) launch_schur_pairs<float>(p->stream, p->ds, p->db, 2); else launch_schur_pairs<double>(p->stream, p->ds, p->db, 2); }
            first_build = 0;
            launch_cd_fold(p->stream, p->ds, p->db);       // deterministic mode: chunk sums in chunk order, before the exchange
            launch_shard_diag(p->stream, p->ds, p->db, p->d_red, /*unpack=*/false, p->shard_rank, p->shard_world);
            if ((rc = reduce(p->d_red, shard_diag_len(p->ds)))) return rc;
            launch_shard_diag(p->stream, p->ds, p->db, p->d_red, /*unpack=*/true, p->shard_rank, p->shard_world);
            { DeviceBuffers dbf = p->db; dbf.cd_part = nullptr; launch_finalize(p->stream, p->ds, dbf, 1); }
            // the pair pass stores its transformed blocks straight into the all-reduce buffer
            DcgSolveArgs da;
            if (dist_cg) {
              const int fo = p->ds.d - 1;
              if (implicit_cg) {
                // no pair pass, no exchange (B): the glue the pair pass does on the way (focal row of S~, b~, post-linearisation) on its own
                launch_pcg_glue(p->stream, p->ds, p->db);
                ip.ds = p->ds; ip.db = p->db; ip.dtab = p->imp_dtab; ip.spt = p->imp_spt; ip.acc = p->imp_acc; ip.part = p->imp_part;
                ip.focal_row = p->solver.Sfull + (size_t)fo * p->ds.ld; ip.focal_row32 = nullptr; ip.rank = p->shard_rank; ip.f32 = f32;
                da.implicit = &ip;
                p->shard_exchange[1] = 0; p->shard_exchange[3] = 2 | 4;
              } else {
                // row-sharded: the glue for ALL cameras on its own (the pair pass below only visits the diagonal blocks of the rank's rows)
                if (row_cg) launch_pcg_glue(p->stream, p->ds, p->db);
                // ... in the reduce-scatter layout (`world` equal chunks of whole block rows behind the region of exchange (A); padding zero)
                double* blocks = p->d_red + p->shard_blocks_off;
                const long long cv = dcg_chunk_values(p->dcg), total = cv * p->shard_world;
                if (p->dcg_last_f32 != (x32 ? 1 : 0)) { HIP_TRY(hipMemsetAsync(blocks, 0, (size_t)total * (x32 ? 4 : 8), p->stream)); p->dcg_last_f32 = x32 ? 1 : 0; }
                p->db.shard_row_shift = p->dcg.d_row_shift;
                if (x32) p->db.shard_blocks32 = reinterpret_cast<float*>(blocks); else p->db.shard_blocks = blocks;
                if (f32) launch_schur_pairs<float>(p->stream, p->ds, p->db, 1); else launch_schur_pairs<double>(p->stream, p->ds, p->db, 1);
                p->db.shard_blocks = nullptr; p->db.shard_blocks32 = nullptr; p->db.shard_row_shift = nullptr;
                char* mine = reinterpret_cast<char*>(blocks) + (size_t)p->shard_rank * (size_t)cv * (x32 ? 4 : 8);
                if (p->shard_world > 1 && !row_cg) {
                    const int rrc = p->reduce_scatter(ctx, blocks, mine, (int64_t)cv, x32 ? 1 : 0, (void*)p->stream);
                    if (rrc != 0) return fail(SFMBA_ERR_HIP, "reduce-scatter failed (rc " + std::to_string(rrc) + ")");
                }
                p->shard_exchange[1] = (x32 ? 4 : 8) * total; p->shard_exchange[3] = (x32 ? 1 : 0) | 2;
                // (row-sharded: the rank's blocks are complete as they leave the pair pass -- what crosses the ranks is the per-point table)
                if (row_cg) { p->shard_exchange[1] = (long long)(p->shard_world - 1) * p->own_pt_stride * (long long)(pa_bytes + pb_bytes); p->shard_exchange[3] = (x32 ? 1 : 0) | 2 | 8; }
                da.owned = mine; da.owned_f32 = x32;
                if (p->db.pcg_F32) da.focal_row32 = p->db.pcg_F32 + (size_t)fo * p->ds.ld; else da.focal_row = p->solver.Sfull + (size_t)fo * p->ds.ld;
              }
                da.bt = p->solver.vec + (size_t)8 * p->ds.ld; da.W = coarse_cg ? p->solver.W : nullptr;
                da.flags = p->solver.flags; da.info = p->d_info; da.tol = cg_tol;
                da.anchor = (!o.pcg_anchored || exact_pcg) ? 0 : first_linear_solve ? 1 : 2;
                { const double t2 = cg_tol * cg_tol; da.cap = t2 > 0.0 ? std::max(t2, 1e-8) / t2 : 1.0; }
                first_linear_solve = false;
                p->dcg.x = p->solver.vec;
                int drc = dcg_begin(p->stream, &p->dcg, da, ar_thunk, &arctx);
                int batch = 24;
                // (every launch of a batch issues its collective, converged or not: one launch in reserve, not two -- a surplus iteration is a whole
                // all-reduce at N > 1; a batch one short costs one host round trip)
                if (p->shard_host_iter < (int)p->solver.hist.size() && p->solver.hist[(size_t)p->shard_host_iter] > 0) batch = p->solver.hist[(size_t)p->shard_host_iter] + 1;
                batch = std::min(batch, dcg_max);
                if (!drc) drc = dcg_iterate(p->stream, &p->dcg, da, batch, ar_thunk, &arctx);
                if (drc) return fail(SFMBA_ERR_HIP, "distributed CG: collective failed (rc " + std::to_string(drc) + ")");
                dcg_launched = batch;
            } else {
            if (x32) p->db.shard_blocks32 = reinterpret_cast<float*>(p->d_red); else p->db.shard_blocks = p->d_red;
            if (f32) launch_schur_pairs<float>(p->stream, p->ds, p->db, 1); else launch_schur_pairs<double>(p->stream, p->ds, p->db, 1);
            p->db.shard_blocks = nullptr; p->db.shard_blocks32 = nullptr;
            if (x32) {
                const int arc = allreduce ? p->allreduce_f32(ctx, p->d_red, shard_offdiag_len(p->ds), (void*)p->stream) : 0;
                if (arc != 0) return fail(SFMBA_ERR_HIP, "all-reduce (fp32) failed (rc " + std::to_string(arc) + ")");
                launch_shard_offdiag_f32(p->stream, p->ds, F32, reinterpret_cast<const float*>(p->d_red));
            } else {
                if ((rc = reduce(p->d_red, shard_offdiag_len(p->ds)))) return rc;
                launch_shard_offdiag(p->stream, p->ds, p->solver.Sfull, p->d_red, /*unpack=*/true);
                if (F32) launch_narrow_matrix(p->stream, p->solver.Sfull, F32, (long long)p->ds.d * p->ds.ld);
            }
            const int anchor = (!o.pcg_anchored || exact_pcg) ? 0 : first_linear_solve ? 1 : 2;
            first_linear_solve = false;
            const int it0 = dense_pcg_solve(p->stream, &p->solver, p->db.S, p->db.rhs, cg_tol, o.pcg_max_iters, p->d_info, nullptr,
                                            /*finish=*/false, /*hist_key=*/p->shard_host_iter, /*pretransformed=*/true, anchor, /*no_wait=*/true, coarse_cg);
            if (it0 < 0) return fail(SFMBA_ERR_ALLOC, "PCG workspace allocation failed");
            }
            DeviceBuffers dbu = p->db;
            if (f32) dbu.pu32 = p->d_pu32;
            dbu.pcg_vec = p->solver.vec; dbu.pcg_linv = p->solver.binv; dbu.pcg_flags = p->solver.flags;
            dbu.cg_gate = p->solver.flags; dbu.cg_force = 0;
            volatile int* mb = p->h_lm_mail;
            for (;;) {
                if (f32) launch_back_substitution<float>(p->stream, p->ds, dsp, dbu, nullptr); else launch_back_substitution<double>(p->stream, p->ds, dsp, dbu, nullptr);
                launch_shard_pack(p->stream, p->db, p->d_scal, 2, p->shard_rank);
                if ((rc = reduce(sfmba_shard_scalars_buf(p), SFMBA_SHARD_SCALARS))) return rc;
                dbu.shard_scal = p->d_scal;        // k_lm_control reads the sums from the all-reduced block
                launch_control(p->stream, p->ds, dbu);
                if (hipError_t le = hipGetLastError(); le != hipSuccess) return fail(SFMBA_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(le));
                ++controls;
                // the next linearisation's first kernel before the host waits for the verdict, as in the one-GPU loop (run_solve): it
                // looks at the LM state itself and returns at once if the solve ended or the iteration wants more CG first
                if (speculate && p->shard_host_iter + 2 <= o.max_iters) {
                    if (f32) launch_point_build<float>(p->stream, dsp, p->db, 4); else launch_point_build<double>(p->stream, dsp, p->db, 4);
                    build_enqueued = true;
                }
                if (wait_mailbox(mb, controls, p->stream) != 0) {
                    const hipError_t se = hipStreamSynchronize(p->stream);
                    return fail(SFMBA_ERR_HIP, std::string("sharded LM iteration did not complete: ") + (se != hipSuccess ? hipGetErrorString(se) : "no control post"));
                }
                if (mb[1] != -2) break;
                if (o.verbose) std::fprintf(stderr, "[sfmba shard %d/%d] CG batch too short after %d launches\n", p->shard_rank, p->shard_world, dist_cg ? dcg_launched : -1);
                // the CG batch was too short (identically on every rank: same matrix, same arithmetic): more iterations, then the trio again
                // (the early linearisation kernel behind that control kernel has returned without doing anything)
                build_enqueued = false;
                if (dist_cg) {
                    const int more = std::min(8, dcg_max - dcg_launched);
                    if (more <= 0) dbu.cg_force = 1;
                    else {
                        const int drc = dcg_iterate(p->stream, &p->dcg, da, more, ar_thunk, &arctx);
                        if (drc) return fail(SFMBA_ERR_HIP, "distributed CG: collective failed (rc " + std::to_string(drc) + ")");
                        dcg_launched += more;
                    }
                } else if (dense_pcg_more(p->stream, &p->solver, 8, nullptr) == 0) dbu.cg_force = 1;
            }
            dense_pcg_note(&p->solver, p->shard_host_iter, mb[4]);
            p->shard_sum.linear_iters += mb[4];
            if (o.verbose) std::fprintf(stderr, "[sfmba shard %d/%d] LM iteration %d: %d CG iterations (%d launched), termination %d\n", p->shard_rank, p->shard_world, mb[3], mb[4], dist_cg ? dcg_launched : 0, mb[1]);
            p->shard_host_iter = mb[3];
            if (mb[1] != -1) {
                p->shard_sum.termination = mb[1];
                std::snprintf(p->shard_sum.message, sizeof(p->shard_sum.message), "%s", message_text(mb[2]));
                break;
            }
            if (p->shard_host_iter >= o.max_iters) {
                p->shard_sum.termination = SFMBA_NO_CONVERGENCE;
                std::snprintf(p->shard_sum.message, sizeof(p->shard_sum.message), "%s", message_text(MSG_MAX_ITERS));
                break;
            }
            // one rank (the matrix-free solve of an unsharded handle): the wall-clock limit of the reference (BA.cpp:176) applies; between
            // ranks it cannot (they would disagree on it)
            if (p->shard_world == 1 && o.max_seconds > 0.0 && now_seconds() - t_shard0 >= o.max_seconds) {
                p->shard_sum.termination = SFMBA_NO_CONVERGENCE;
                std::snprintf(p->shard_sum.message, sizeof(p->shard_sum.message), "%s", message_text(MSG_MAX_TIME));
                break;
            }
        }
        if (row_cg) {
            // every rank has moved its own points only: the final points of all ranks, so that the handle holds the whole solution
            if ((rc = download_state(p))) return rc;
            if ((rc = gather(p->db.pts[p->h_state->cur], (size_t)p->own_pt_stride * 3 * sizeof(double)))) return rc;
        }
        return sfmba_shard_end(p, summary);
    }
    for (;;) {
        if ((rc = sfmba_shard_partial_build(p))) return rc;
        if ((rc = reduce(sfmba_shard_reduce_buf(p), sfmba_shard_reduce_len(p)))) return rc;
        if ((rc = sfmba_shard_solve_update(p))) return rc;
        if ((rc = reduce(sfmba_shard_scalars_buf(p), SFMBA_SHARD_SCALARS))) return rc;
        // accept / reject on the device; the host meets the GPU at the control kernel's mailbox post (no copy, no stream sync)
        launch_shard_unpack(p->stream, p->db, p->d_scal, 2, p->shard_world);
        launch_control(p->stream, p->ds, p->db);
        if (hipError_t le = hipGetLastError(); le != hipSuccess) return fail(SFMBA_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(le));
        ++controls;
        volatile int* mb = p->h_lm_mail;
        if (wait_mailbox(mb, controls, p->stream) != 0) {
            const hipError_t se = hipStreamSynchronize(p->stream);
            return fail(SFMBA_ERR_HIP, std::string("sharded LM iteration did not complete: ") + (se != hipSuccess ? hipGetErrorString(se) : "no control post"));
        }
        p->shard_host_iter = mb[3];
        if (mb[1] != -1) {
            p->shard_sum.termination = mb[1];
            std::snprintf(p->shard_sum.message, sizeof(p->shard_sum.message), "%s", message_text(mb[2]));
            break;
        }
        if (p->shard_host_iter >= o.max_iters) {
            p->shard_sum.termination = SFMBA_NO_CONVERGENCE;
            std::snprintf(p->shard_sum.message, sizeof(p->shard_sum.message), "%s", message_text(MSG_MAX_ITERS));
            break;
        }
    }
    return sfmba_shard_end(p, summary);
}

// ---- RCCL (ncclAllReduce over xGMI) bound at run time: the library has no link-time dependency on librccl ----
struct sfmba_comm { ncclComm_t comm = nullptr; int rank = 0, world = 1; };
namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (api.handle) break; }
        if (api.handle) {
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
            api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
            api.ReduceScatter = reinterpret_cast<decltype(api.ReduceScatter)>(dlsym(api.handle, "ncclReduceScatter"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
            api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
            api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.handle, "ncclCommUserRank"));
            if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) api.handle = nullptr;
        }
    }
    return api.handle ? &api : nullptr;
}
}  // namespace

int sfmba_comm_unique_id(unsigned char id[SFMBA_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == SFMBA_COMM_ID_BYTES, "ncclUniqueId size");
    RcclApi* a = rccl();
    if (!a || !id) return fail(SFMBA_ERR_HIP, "RCCL (librccl.so) is not available");
    ncclUniqueId u;
    const ncclResult_t r = a->GetUniqueId(&u);
    if (r != ncclSuccess) return fail(SFMBA_ERR_HIP, std::string("ncclGetUniqueId: ") + (a->GetErrorString ? a->GetErrorString(r) : "error"));
    std::memcpy(id, &u, sizeof(u));
    return SFMBA_OK;
}

int sfmba_comm_create(const unsigned char id[SFMBA_COMM_ID_BYTES], int rank, int world, int device, sfmba_comm** out) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(SFMBA_ERR_INVALID_ARG, "bad argument");
    *out = nullptr;
    RcclApi* a = rccl();
    if (!a) return fail(SFMBA_ERR_HIP, "RCCL (librccl.so) is not available");
    int rc = check_device(device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    sfmba_comm* c = new sfmba_comm();
    c->rank = rank; c->world = world;
    const ncclResult_t r = a->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return fail(SFMBA_ERR_HIP, std::string("ncclCommInitRank: ") + (a->GetErrorString ? a->GetErrorString(r) : "error")); }
    *out = c;
    return SFMBA_OK;
}

void sfmba_comm_destroy(sfmba_comm* c) {
    if (!c) return;
    RcclApi* a = rccl();
    if (a && c->comm) (void)a->CommDestroy(c->comm);
    delete c;
}

int sfmba_comm_size(const sfmba_comm* c, int* world, int* rank) {
    RcclApi* a = rccl();
    if (!c || !c->comm || !a || !a->CommCount || !a->CommUserRank) return fail(SFMBA_ERR_HIP, "ncclCommCount is not available");
    int n = 0, r = -1;
    ncclResult_t e = a->CommCount(c->comm, &n);
    if (e == ncclSuccess) e = a->CommUserRank(c->comm, &r);
    if (e != ncclSuccess) return fail(SFMBA_ERR_HIP, std::string("ncclCommCount: ") + (a->GetErrorString ? a->GetErrorString(e) : "error"));
    if (world) *world = n;
    if (rank) *rank = r;
    return SFMBA_OK;
}

int sfmba_comm_abort(sfmba_comm* c) {
    RcclApi* a = rccl();
    if (!c || !a || !a->CommAbort) return fail(SFMBA_ERR_HIP, "ncclCommAbort is not available");
    if (c->comm) { (void)a->CommAbort(c->comm); c->comm = nullptr; }
    return SFMBA_OK;
}

int sfmba_comm_allreduce(void* comm, void* device_buf, int64_t n_doubles, void* hip_stream) {
    sfmba_comm* c = static_cast<sfmba_comm*>(comm);
    RcclApi* a = rccl();
    if (!c || !a) return -1;
    const ncclResult_t r = a->AllReduce(device_buf, device_buf, (size_t)n_doubles, ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(hip_stream));
    return r == ncclSuccess ? 0 : (int)r;
}

int sfmba_comm_allreduce_f32(void* comm, void* device_buf, int64_t n_floats, void* hip_stream) {
    sfmba_comm* c = static_cast<sfmba_comm*>(comm);
    RcclApi* a = rccl();
    if (!c || !a) return -1;
    const ncclResult_t r = a->AllReduce(device_buf, device_buf, (size_t)n_floats, ncclFloat, ncclSum, c->comm, static_cast<hipStream_t>(hip_stream));
    return r == ncclSuccess ? 0 : (int)r;
}

int sfmba_comm_reduce_scatter(void* comm, void* send_buf, void* recv_buf, int64_t n_values, int is_f32, void* hip_stream) {
    sfmba_comm* c = static_cast<sfmba_comm*>(comm);
    RcclApi* a = rccl();
    if (!c || !a || !a->ReduceScatter) return -1;
    const ncclResult_t r = a->ReduceScatter(send_buf, recv_buf, (size_t)n_values, is_f32 ? ncclFloat : ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(hip_stream));
    return r == ncclSuccess ? 0 : (int)r;
}

int sfmba_comm_allgather(void* comm, void* buf, int64_t bytes_per_rank, void* hip_stream) {
    sfmba_comm* c = static_cast<sfmba_comm*>(comm);
    RcclApi* a = rccl();
    if (!c || !a || !a->AllGather) return -1;
    const ncclResult_t r = a->AllGather(static_cast<char*>(buf) + (size_t)c->rank * (size_t)bytes_per_rank, buf, (size_t)bytes_per_rank, ncclChar, c->comm,
                                        static_cast<hipStream_t>(hip_stream));
    return r == ncclSuccess ? 0 : (int)r;
}

int sfmba_problem_set_allgather(sfmba_problem* p, sfmba_allgather_fn allgather) {
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    p->allgather = allgather;
    return SFMBA_OK;
}

int sfmba_problem_set_reduce_scatter(sfmba_problem* p, sfmba_reduce_scatter_fn reduce_scatter) {
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    p->reduce_scatter = reduce_scatter;
    return SFMBA_OK;
}

int sfmba_shard_last_exchange(const sfmba_problem* p, int64_t out[4]) {
    if (!p || !out) return fail(SFMBA_ERR_INVALID_ARG, "NULL argument");
    for (int k = 0; k < 4; ++k) out[k] = p->shard_exchange[k];
    return SFMBA_OK;
}

int sfmba_problem_set_allreduce_f32(sfmba_problem* p, sfmba_allreduce_f32_fn allreduce_f32) {
    if (!p) return fail(SFMBA_ERR_INVALID_ARG, "NULL problem");
    if (p->poisoned) return fail(SFMBA_ERR_INVALID_ARG, "poisoned problem (a failed sfmba_problem_append): destroy it");
    p->allreduce_f32 = allreduce_f32;
    return SFMBA_OK;
}

int sfmba_triangulate(int device, int64_t n, const float* left_xy, const float* right_xy, const float* K, const float* P_left,
                      const float* P_right, float max_reproj_px, float* points3d, unsigned char* keep, float* reproj_err) {
    if (n < 0 || !K || !P_left || !P_right || (n > 0 && (!left_xy || !right_xy || !points3d || !keep)))
        return fail(SFMBA_ERR_INVALID_ARG, "bad argument");
    int rc = check_device(device);
    if (rc) return rc;
    if (n == 0) return SFMBA_OK;
    HIP_TRY(hipSetDevice(device));
    DeviceArena arena(device);
    HostKit kit;
    if (!hostkit_acquire(device, &kit)) return fail(SFMBA_ERR_HIP, "stream creation failed");
    struct KitGuard { HostKit k; ~KitGuard() { if (k.stream) (void)hipStreamSynchronize(k.stream); hostkit_release(k); } } kg{ kit };
    float* d_l = arena.alloc_n<float>((size_t)2 * n);
    float* d_r = arena.alloc_n<float>((size_t)2 * n);
    float* d_x = arena.alloc_n<float>((size_t)3 * n);
    float* d_e = reproj_err ? arena.alloc_n<float>((size_t)2 * n) : nullptr;
    unsigned char* d_k = arena.alloc_n<unsigned char>((size_t)n);
    if (!d_l || !d_r || !d_x || !d_k || (reproj_err && !d_e)) return fail(SFMBA_ERR_ALLOC, "device allocation failed");
    HIP_TRY(hipMemcpyAsync(d_l, left_xy, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, kit.stream));
    HIP_TRY(hipMemcpyAsync(d_r, right_xy, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, kit.stream));
    launch_triangulate(kit.stream, (long long)n, d_l, d_r, K, P_left, P_right, max_reproj_px, d_x, d_k, d_e);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(points3d, d_x, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, kit.stream));
    HIP_TRY(hipMemcpyAsync(keep, d_k, (size_t)n, hipMemcpyDeviceToHost, kit.stream));
    if (reproj_err) HIP_TRY(hipMemcpyAsync(reproj_err, d_e, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, kit.stream));
    HIP_TRY(hipStreamSynchronize(kit.stream));
    return SFMBA_OK;
}

// ---- association joins -----------------------------------------------------------------------
static int assoc_result(int rc, const char* what) {
    if (rc == 0) return SFMBA_OK;
    if (rc == ASSOC_ERR_CAPACITY) return fail(SFMBA_ERR_CAPACITY, std::string(what) + ": output capacity too small");
    if (rc == ASSOC_ERR_TOO_LARGE) return fail(SFMBA_ERR_INVALID_ARG, std::string(what) + ": problem too large for 32-bit indices");
    if (rc == (int)hipErrorOutOfMemory) return fail(SFMBA_ERR_ALLOC, std::string(what) + ": device allocation failed");
    return fail(SFMBA_ERR_HIP, std::string(what) + ": " + hipGetErrorString((hipError_t)rc));
}

int sfmba_find_2d3d_matches(int device, int n_views, const unsigned char* view_done, int n_pt, const int64_t* view_ptr,
                            const int32_t* view_idx, const int32_t* feat_idx, int n_pairs, const int32_t* pair_left,
                            const int32_t* pair_right, const int64_t* pair_ptr, const int32_t* query_idx, const int32_t* train_idx,
                            int64_t* out_ptr, int32_t* out_point, int32_t* out_feature, int64_t cap, int64_t* total) {
    if (n_views < 0 || n_pt < 0 || n_pairs < 0 || cap < 0 || !out_ptr || !total || (n_views > 0 && !view_done) || !view_ptr ||
        (n_pairs > 0 && (!pair_left || !pair_right || !pair_ptr)) || (cap > 0 && (!out_point || !out_feature)))
        return fail(SFMBA_ERR_INVALID_ARG, "bad argument");
    if (view_ptr[0] != 0 || (n_pairs > 0 && pair_ptr[0] != 0)) return fail(SFMBA_ERR_INVALID_ARG, "CSR pointers must start at 0");
    for (int i = 0; i < n_pt; ++i) if (view_ptr[i + 1] < view_ptr[i]) return fail(SFMBA_ERR_INVALID_ARG, "view_ptr not monotone");
    for (int p = 0; p < n_pairs; ++p) if (pair_ptr[p + 1] < pair_ptr[p]) return fail(SFMBA_ERR_INVALID_ARG, "pair_ptr not monotone");
    if ((view_ptr[n_pt] > 0 && (!view_idx || !feat_idx)) || (n_pairs > 0 && pair_ptr[n_pairs] > 0 && (!query_idx || !train_idx)))
        return fail(SFMBA_ERR_INVALID_ARG, "NULL array");
    int rc = check_device(device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    HostKit kit;
    if (!hostkit_acquire(device, &kit)) return fail(SFMBA_ERR_HIP, "stream creation failed");
    struct KitGuard { HostKit k; ~KitGuard() { if (k.stream) (void)hipStreamSynchronize(k.stream); hostkit_release(k); } } kg{ kit };
    return assoc_result(assoc_find_2d3d(kit.stream, device, n_views, view_done, n_pt, view_ptr, view_idx, feat_idx, n_pairs, pair_left, pair_right,
                                        pair_ptr, query_idx, train_idx, out_ptr, out_point, out_feature, cap, total), "find_2d3d_matches");
}

int sfmba_merge_candidates(int device, int n_exist, const float* exist_xyz, int n_new, const float* new_xyz, float max_dist,
                           int64_t* cand_ptr, int32_t* cand_idx, int64_t cap, int64_t* total) {
    if (n_exist < 0 || n_new < 0 || cap < 0 || !cand_ptr || !total || (n_exist > 0 && !exist_xyz) || (n_new > 0 && !new_xyz) || (cap > 0 && !cand_idx))
        return fail(SFMBA_ERR_INVALID_ARG, "bad argument");
    int rc = check_device(device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    HostKit kit;
    if (!hostkit_acquire(device, &kit)) return fail(SFMBA_ERR_HIP, "stream creation failed");
    struct KitGuard { HostKit k; ~KitGuard() { if (k.stream) (void)hipStreamSynchronize(k.stream); hostkit_release(k); } } kg{ kit };
    return assoc_result(assoc_radius_candidates(kit.stream, device, n_exist, exist_xyz, n_new, new_xyz, max_dist, cand_ptr, cand_idx, cap, total),
                        "merge_candidates");
}

}  // extern "C"

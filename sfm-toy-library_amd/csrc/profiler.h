// profiler.h -- optional per-kernel HIP-event timing on the solver's own stream.
// bench.py turns it on for the timed region: every launch is bracketed by two hipEventRecord calls
// on the stream the kernel runs on (no host synchronisation until the solve has finished), so the
// average launch duration of each kernel class can be reported next to rocprofv3's.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace sfmba {

enum KernelId {
    KID_SETUP = 0, KID_POINT_BUILD, KID_SCHUR_PAIRS, KID_CAM_DIAG, KID_FINALIZE,
    KID_CHOL_AUGMENT, KID_CHOL_PANEL, KID_CHOL_UPDATE, KID_CHOL_EXTRACT, KID_CHOL_BACKSTEP,
    KID_PCG_SETUP, KID_PCG_ITER, KID_PCG_FINISH,
    KID_CAM_UPDATE, KID_POINT_UPDATE, KID_CONTROL, KID_EMPTY, KID_COUNT
};

inline const char* kernel_name(int id) {
    static const char* names[KID_COUNT] = {
        "setup", "point_build", "schur_pairs", "cam_diag", "finalize",
        "chol_augment", "chol_panel", "chol_update", "chol_extract", "chol_backstep",
        "pcg_setup", "pcg_iter", "pcg_finish",
        "cam_update", "point_update", "lm_control", "empty_bracket" };
    return (id >= 0 && id < KID_COUNT) ? names[id] : "?";
}

struct Profiler {
    bool on = false;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    struct Rec { int id; size_t e0, e1; int launches; };
    std::vector<Rec> recs;
    double total_ms[KID_COUNT] = {};
    long long count[KID_COUNT] = {};

    hipEvent_t next() {
        if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
        return pool[used++];
    }
    void begin(int id, hipStream_t s) {
        if (!on) return;
        Rec r; r.id = id; r.e0 = used; (void)hipEventRecord(next(), s); r.e1 = 0; r.launches = 1; recs.push_back(r);
    }
    void end(hipStream_t s, int launches = 1) {
        if (!on) return;
        recs.back().e1 = used; recs.back().launches = launches; (void)hipEventRecord(next(), s);
    }
    // call with the stream idle
    void collect() {
        for (const Rec& r : recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pool[r.e0], pool[r.e1]) == hipSuccess) { total_ms[r.id] += ms; count[r.id] += r.launches; }
        }
        recs.clear();
        used = 0;
    }
    void reset() { for (int i = 0; i < KID_COUNT; ++i) { total_ms[i] = 0; count[i] = 0; } recs.clear(); used = 0; }
    void destroy() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); pool.clear(); }
};

struct ProfScope {
    Profiler* p; hipStream_t s; int launches;
    // launches > 1: one event pair around a batch of back-to-back launches of the same kernel (per-launch average
    // then includes the launch boundary, but not the ~5 us cost of an event pair per few-microsecond kernel)
    ProfScope(Profiler* p_, int id, hipStream_t s_, int launches_ = 1) : p(p_), s(s_), launches(launches_) { if (p) p->begin(id, s); }
    ~ProfScope() { if (p) p->end(s, launches); }
};

}  // namespace sfmba

// pool_sanitize.cpp -- the shim's host-side concurrency under a sanitizer (make -C host tsan / asan; tests/test_sanitizers_cpu.py).  No GPU involved:
// the resident WorkerPool of SfMBundleAdjustmentUtils.cpp (one atomic ticket word + a generation counter, two std::function slots, callers on several
// threads taking turns) is driven through the self-test entry points the shim exports, from ONE instrumented executable -- ThreadSanitizer needs the
// whole program instrumented, which a Python test process loading libsfmba_shim.so is not.  SURVEY section 5 asked for sanitizer runs of the shim.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

extern "C" int sfmba_shim_pool_selftest(int batches);
extern "C" int sfmba_shim_pool_stress(int batches, int max_tasks);

int main(int argc, char** argv) {
    const int scale = argc > 1 ? std::atoi(argv[1]) : 1;
    int bad = 0;
    bad += sfmba_shim_pool_selftest(400 * scale);                 // run() and begin() / end() alternating, varying batch sizes
    bad += sfmba_shim_pool_stress(20000 * scale, 6);              // many small batches: the window between the ticket word and the generation store
    std::vector<int> out(4, -1);                                  // callers on different threads take turns on the one pool
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t) th.emplace_back([&out, t, scale] { out[t] = sfmba_shim_pool_selftest(150 * scale); });
    for (auto& t : th) t.join();
    for (int v : out) bad += v;
    std::printf("pool_sanitize: %d wrong batch(es)\n", bad);
    return bad == 0 ? 0 : 1;
}

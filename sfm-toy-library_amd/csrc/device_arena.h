// device_arena.h -- per-problem device memory arena with a process-wide chunk cache.
//
// A problem needs ~45 device arrays; hipMalloc/hipFree of each cost ~0.1-0.2 ms, which is 2/3 of a resident solve at
// BASELINE config 3 and matters for the one-shot boundary (adjustBundle() re-creates the problem on every call,
// SfMToyLib/SfM.cpp:464-466).  The arena carves the arrays out of a few large chunks; chunks of destroyed problems go
// to a cache (bounded) and are handed to the next problem on the same device, so a steady stream of one-shot calls
// performs no hipMalloc/hipFree at all.  sfmba_release_cache() (include/sfmba.h) returns the cached memory to HIP.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>
#include <utility>
#include <vector>

namespace sfmba {

struct ArenaChunk {
    char* base = nullptr;
    size_t cap = 0;
    int device = 0;
};

class DeviceArena {
public:
    explicit DeviceArena(int device = 0) : device_(device) {}
    DeviceArena(const DeviceArena&) = delete;
    DeviceArena& operator=(const DeviceArena&) = delete;
    ~DeviceArena();
    void set_device(int device) { device_ = device; }
    // Memory is handed out ZEROED (fresh or recycled chunks alike).  By default the zeroing is waited for; with an ordering stream
    // the host does not wait: the fill runs on the NULL stream (later NULL-stream copies are ordered behind it) and `s` is made to
    // wait for it, so work enqueued on `s` after alloc() returned sees zeroed memory.  Nothing else may touch the memory.
    void set_ordering_stream(hipStream_t s) { order_stream_ = s; }
    // scratch arenas (sort temporaries): recycled chunks are handed out as they are
    void set_zeroing(bool on) { zero_ = on; }
    // 256-byte aligned block of `bytes` (>= 1); nullptr when the device allocation fails.  May be called from two threads at once
    // (the two halves of a structure build); everything else is single-threaded.
    void* alloc(size_t bytes);
    template <typename T> T* alloc_n(size_t n) { return static_cast<T*>(alloc(sizeof(T) * (n ? n : 1))); }
    // Returns every chunk to the cache (or to HIP when the cache is full).  All pointers handed out become invalid.
    void release();
    size_t bytes_reserved() const;
    // exchange the contents (chunks) of two arenas
    void swap(DeviceArena& o) { std::swap(device_, o.device_); chunks_.swap(o.chunks_); std::swap(off_, o.off_); }      // (settings stay)

private:
    int device_;
    std::vector<ArenaChunk> chunks_;
    size_t off_ = 0;   // fill of chunks_.back()
    std::mutex mu_;    // alloc() only
    hipStream_t order_stream_ = nullptr;
    hipEvent_t order_event_ = nullptr;
    bool zero_ = true;
};

// Host-side resources of a problem that are slow to create and destroy (a private non-blocking stream and one 64 KB
// block of pinned, host-mapped memory for the LM state mirror and the mailboxes); recycled through the same cache.
struct HostKit {
    int device = 0;
    hipStream_t stream = nullptr;
    char* pinned = nullptr;            // HOSTKIT_PINNED_BYTES, hipHostMallocMapped
    char* upload = nullptr;            // pinned memory for uploads (no staging copy, no page faults): upload_bytes, grown on demand by hostkit_upload
    size_t upload_bytes = 0;
};
constexpr size_t HOSTKIT_PINNED_BYTES = 65536;    // [0,1024) LM state mirror, [1024,2048) LM mailbox, [2048,4096) solver, [4096,..) trace rows
constexpr size_t HOSTKIT_UPLOAD_BYTES = (size_t)1 << 20;          // initial size
constexpr size_t HOSTKIT_UPLOAD_MAX = (size_t)256 << 20;         // larger uploads go through pageable memory
// Pinned upload buffer of at least `bytes` (the kit's, grown if need be; contents are not kept); nullptr if bytes > HOSTKIT_UPLOAD_MAX or
// the allocation fails.  Nothing enqueued from the old buffer may still be in flight.
char* hostkit_upload(HostKit* kit, size_t bytes);
bool hostkit_acquire(int device, HostKit* kit);     // false on HIP failure
void hostkit_release(const HostKit& kit);           // the stream must be idle

// Frees every cached chunk and host kit (all devices).  Returns the number of device bytes released.
size_t arena_cache_release();

}  // namespace sfmba

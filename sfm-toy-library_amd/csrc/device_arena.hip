// device_arena.hip -- see device_arena.h
#include "device_arena.h"

#include <algorithm>
#include <mutex>

namespace sfmba {

namespace {

constexpr size_t ALIGN = 256;
constexpr size_t MIN_CHUNK = (size_t)32 << 20;          // small arrays share 32 MB chunks
constexpr size_t CACHE_LIMIT = (size_t)8 << 30;         // at most 8 GB of idle chunks (of 288 GB)

std::mutex g_mu;
std::vector<ArenaChunk> g_cache;
size_t g_cached_bytes = 0;
std::vector<HostKit> g_kits;
constexpr size_t KIT_LIMIT = 16;

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// smallest cached chunk of this device with cap >= need (and not absurdly larger than the request)
bool cache_take(int device, size_t need, ArenaChunk* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    int best = -1;
    for (int i = 0; i < (int)g_cache.size(); ++i) {
        const ArenaChunk& c = g_cache[i];
        if (c.device != device || c.cap < need || c.cap > 2 * need + MIN_CHUNK) continue;
        if (best < 0 || c.cap < g_cache[best].cap) best = i;
    }
    if (best < 0) return false;
    *out = g_cache[best];
    g_cached_bytes -= out->cap;
    g_cache.erase(g_cache.begin() + best);
    return true;
}

}  // namespace

void* DeviceArena::alloc(size_t bytes) {
    std::lock_guard<std::mutex> lk(mu_);
    bytes = round_up(bytes ? bytes : 1, ALIGN);
    if (!chunks_.empty() && off_ + bytes <= chunks_.back().cap) {
        void* p = chunks_.back().base + off_;
        off_ += bytes;
        return p;
    }
    const size_t need = std::max(bytes, MIN_CHUNK);
    ArenaChunk c;
    if (!cache_take(device_, need, &c)) {
        void* base = nullptr;
        if (hipMalloc(&base, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        c.base = static_cast<char*>(base); c.cap = need; c.device = device_;
    }
    if (zero_) {
        bool ok = hipMemsetAsync(c.base, 0, c.cap, nullptr) == hipSuccess;
        if (ok && order_stream_) {
            if (!order_event_) ok = hipEventCreateWithFlags(&order_event_, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventRecord(order_event_, nullptr) == hipSuccess && hipStreamWaitEvent(order_stream_, order_event_, 0) == hipSuccess;
        } else if (ok) ok = hipStreamSynchronize(nullptr) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); (void)hipFree(c.base); return nullptr; }
    }
    // keep a partially filled small chunk usable: put the new chunk last only if it has more room left
    chunks_.push_back(c);
    off_ = bytes;
    return c.base;
}

DeviceArena::~DeviceArena() {
    release();
    if (order_event_) (void)hipEventDestroy(order_event_);
}

void DeviceArena::release() {
    std::vector<ArenaChunk> drop;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (const ArenaChunk& c : chunks_) {
            if (g_cached_bytes + c.cap <= CACHE_LIMIT) { g_cache.push_back(c); g_cached_bytes += c.cap; }
            else drop.push_back(c);
        }
    }
    for (const ArenaChunk& c : drop) { (void)hipSetDevice(c.device); (void)hipFree(c.base); }
    chunks_.clear();
    off_ = 0;
}

size_t DeviceArena::bytes_reserved() const {
    size_t n = 0;
    for (const ArenaChunk& c : chunks_) n += c.cap;
    return n;
}

bool hostkit_acquire(int device, HostKit* kit) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_kits.size(); ++i)
            if (g_kits[i].device == device) { *kit = g_kits[i]; g_kits.erase(g_kits.begin() + i); return true; }
    }
    HostKit k;
    k.device = device;
    if (hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess) return false;
    void* hm = nullptr;
    if (hipHostMalloc(&hm, HOSTKIT_PINNED_BYTES, hipHostMallocMapped) != hipSuccess) { (void)hipStreamDestroy(k.stream); return false; }
    k.pinned = static_cast<char*>(hm);
    void* up = nullptr;
    if (hipHostMalloc(&up, HOSTKIT_UPLOAD_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipHostFree(hm); (void)hipStreamDestroy(k.stream); return false; }
    k.upload = static_cast<char*>(up);
    k.upload_bytes = HOSTKIT_UPLOAD_BYTES;
    *kit = k;
    return true;
}

char* hostkit_upload(HostKit* kit, size_t bytes) {
    if (bytes <= kit->upload_bytes && kit->upload) return kit->upload;
    if (bytes > HOSTKIT_UPLOAD_MAX) return nullptr;
    const size_t want = std::min(HOSTKIT_UPLOAD_MAX, std::max(bytes + bytes / 4, HOSTKIT_UPLOAD_BYTES));
    void* up = nullptr;
    if (hipHostMalloc(&up, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (kit->upload) (void)hipHostFree(kit->upload);
    kit->upload = static_cast<char*>(up);
    kit->upload_bytes = want;
    return kit->upload;
}

static void hostkit_destroy(const HostKit& k) {
    (void)hipSetDevice(k.device);
    if (k.pinned) (void)hipHostFree(k.pinned);
    if (k.upload) (void)hipHostFree(k.upload);
    if (k.stream) (void)hipStreamDestroy(k.stream);
}

void hostkit_release(const HostKit& kit) {
    if (!kit.stream && !kit.pinned) return;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_kits.size() < KIT_LIMIT) { g_kits.push_back(kit); return; }
    }
    hostkit_destroy(kit);
}

size_t arena_cache_release() {
    std::vector<ArenaChunk> all;
    std::vector<HostKit> kits;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        all.swap(g_cache);
        kits.swap(g_kits);
        g_cached_bytes = 0;
    }
    for (const HostKit& k : kits) hostkit_destroy(k);
    size_t n = 0;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (const ArenaChunk& c : all) { (void)hipSetDevice(c.device); (void)hipFree(c.base); n += c.cap; }
    if (cur >= 0) (void)hipSetDevice(cur);
    return n;
}

}  // namespace sfmba

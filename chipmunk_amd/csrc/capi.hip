// Error plumbing and version of the C ABI (include/chipmunk_hip.h).
#include "common.h"

static thread_local char g_last_error[512] = "";

void chipmunk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *chipmunk_last_error(void) { return g_last_error; }
extern "C" int chipmunk_abi_version(void) { return 1; }

// ---- tuning knobs (kernel variant selection for A/B measurement; defaults are the shipped choices) ----
#include <string.h>
namespace {
struct Option { const char *name; int value; };
Option g_options[] = {{"mm1_variant", 0}, {"mm2_variant", 0}, {"attn_variant", 0}, {"m2i_variant", 0}, {"topk_variant", 0}, {"mm1_nr", 0}, {"mm2_nr", 0}, {"mm1_probe", 0}, {"attn_xcd_chunks", 0}, {"attn_no_order", 0}, {"attn_no_tail", 0}, {"mm1_no_split", 0}, {"mm2_no_split", 0}, {"attn_no_split", 0}, {"attn_split_gather", 0}, {"attn_pp", 0}, {"attn_dense64", 0}, {"attn_csp64", 0}, {"attn_colsum64", 0}, {"attn_csp96", 0}, {"attn_nomax", 0}, {"attn_fused_colsum", 0}, {"attn_cs_probe", 0}, {"big_scratch_gb", 0}, {"attn_balanced", 0}, {"attn_row_split", 0}, {"mm2_order", 0}};
}
int chipmunk_get_option(const char *name) {
    for (auto &o : g_options) if (strcmp(o.name, name) == 0) return o.value;
    return 0;
}
extern "C" int chipmunk_set_option(const char *name, int value) {
    for (auto &o : g_options) if (strcmp(o.name, name) == 0) { o.value = value; return CHIPMUNK_OK; }
    chipmunk_set_error("unknown option '%s'", name);
    return CHIPMUNK_ERR_INVALID;
}

// ---- random keys: per-launch salt ----
// topk_indices / topk_delta_indices / topk_mask add `random_amount` of the columns through a counter-based hash of
// (row, column, salt).  The salt is splitmix64(seed + launch counter): every launch -- layer, step, generation -- draws
// a different set (the reference reseeds cuRAND per call, topk_indices.cu:47-49, and calls torch.randint per layer,
// modules/attn.py:77).  chipmunk_set_random_seed(seed) restarts the sequence: same seed + same launch order = same sets.
// A launch captured into a hipGraph replays with the salt it was captured with; the kernels also mix in the first
// element of each row, so replays on new data still draw new sets.
#include <atomic>
namespace {
std::atomic<uint64_t> g_rng_seed{0x243F6A8885A308D3ull};
std::atomic<uint64_t> g_rng_counter{0};
}
uint32_t chipmunk_next_random_salt() {
    uint64_t z = g_rng_seed.load(std::memory_order_relaxed) + 0x9E3779B97F4A7C15ull * (g_rng_counter.fetch_add(1) + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 16);
}
extern "C" int chipmunk_set_random_seed(uint64_t seed) {
    g_rng_seed.store(seed * 0xD6E8FEB86659FD93ull + 0x243F6A8885A308D3ull);
    g_rng_counter.store(0);
    return CHIPMUNK_OK;
}

// ---- per-(device, stream) scratch: schedule arrays, split-K partials, arrival tickets ----
// Grow-only, zero-filled when (re)allocated, contents persist between launches (kernels that use tickets leave them at
// zero).  Keyed by stream so that launches on different streams never share a buffer; launches on one stream are
// ordered, which is all the users need.  A stream-ordered hipMallocAsync/hipFreeAsync pair per launch measured tens of
// microseconds of host time -- more than the kernels it served.
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
namespace {
struct Scratch { void *ptr; size_t bytes; };
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, Scratch> g_scratch;
}
void *chipmunk_scratch(hipStream_t stream, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch &s = g_scratch[{dev, stream}];
    if (s.bytes >= bytes) return s.ptr;
    if (s.ptr) {
        (void)hipStreamSynchronize(stream);  // earlier launches on this stream may still be using the old buffer
        (void)hipFree(s.ptr);
        s.ptr = nullptr, s.bytes = 0;
    }
    const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    void *ptr = nullptr;
    if (hipMalloc(&ptr, want) != hipSuccess) return nullptr;
    if (hipMemsetAsync(ptr, 0, want, stream) != hipSuccess) {
        (void)hipFree(ptr);
        return nullptr;
    }
    s.ptr = ptr, s.bytes = want;
    return ptr;
}
// A second, separately grown buffer for the one multi-GB user (the per-wave partial column sums of the fused
// dense_colsum_attn pass: 10.6 GB of bf16 at HunyuanVideo size).  Not zeroed; nullptr if the device cannot spare it (the
// caller then takes smaller head chunks or the two-pass route).  The buffer lives outside torch's caching allocator, so it is
// bounded: never more than option `big_scratch_gb` GiB (default 24) and never more than the free memory minus a 4 GiB
// reserve at the time of the request; a size that failed is remembered for a while, so failing hipMallocs are not retried on every call;
// chipmunk_release_scratch() gives everything back.
namespace {
std::map<std::pair<int, hipStream_t>, Scratch> g_big;
struct BigFail { size_t want, free_then; int calls_left; };
std::map<int, BigFail> g_big_failed;   // per device: smallest request that could not be served, and when to try again
int g_big_fallbacks = 0;               // requests answered with nullptr since the last release (bench.py reports it)
}
void *chipmunk_big_scratch(hipStream_t stream, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch &s = g_big[{dev, stream}];
    if (s.bytes >= bytes) return s.ptr;
    const int cap_gb = chipmunk_get_option("big_scratch_gb");
    const size_t cap = (size_t)(cap_gb > 0 ? cap_gb : 24) << 30;
    const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    if (want > cap) return ++g_big_fallbacks, nullptr;
    size_t free_b = 0, total_b = 0;
    const bool have_info = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    auto f = g_big_failed.find(dev);
    if (f != g_big_failed.end() && want >= f->second.want) {
        // a remembered failure is not for ever: hipMemGetInfo does not see torch's cached blocks, so one transient low
        // moment must not switch the one-pass column-sum route off for good -- try again once noticeably more memory is free than
        // at the time of the failure, or every 64th request
        const bool more_free = have_info && free_b > f->second.free_then + ((size_t)1 << 30);
        if (!more_free && --f->second.calls_left > 0) return ++g_big_fallbacks, nullptr;
        g_big_failed.erase(f);
        f = g_big_failed.end();
    }
    auto failed = [&]() {
        g_big_failed[dev] = BigFail{f == g_big_failed.end() ? want : std::min(f->second.want, want), free_b, 64};
        ++g_big_fallbacks;
        return nullptr;
    };
    // the old buffer is given back BEFORE the larger one is requested (the headroom test below counts its bytes as free, so the
    // request has to be able to use them); contents are scratch, nothing is lost
    if (have_info && want > free_b + s.bytes - std::min(free_b + s.bytes, (size_t)4 << 30)) return failed();
    if (s.ptr) {
        (void)hipStreamSynchronize(stream);   // earlier launches on this stream may still be using the old buffer
        (void)hipFree(s.ptr);
        s.ptr = nullptr, s.bytes = 0;
    }
    void *ptr = nullptr;
    if (hipMalloc(&ptr, want) != hipSuccess) {
        (void)hipGetLastError();
        return failed();
    }
    s.ptr = ptr, s.bytes = want;
    return ptr;
}
extern "C" int chipmunk_big_scratch_fallbacks(void) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    return g_big_fallbacks;
}
// ---- pinned host offload pool (include/chipmunk_hip.h): hipHostMalloc + hipMemcpyAsync on the caller's side stream ----
namespace {
std::mutex g_host_mu;
std::map<void *, size_t> g_host_allocs;
size_t g_host_total = 0;
}
extern "C" int chipmunk_host_alloc(size_t bytes, void **host_ptr) {
    CM_CHECK(host_ptr != nullptr && bytes > 0, "chipmunk_host_alloc: null result pointer or zero bytes");
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess || !p) {
        chipmunk_set_error("chipmunk_host_alloc: hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return CHIPMUNK_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_host_mu);
    g_host_allocs[p] = bytes;
    g_host_total += bytes;
    *host_ptr = p;
    return CHIPMUNK_OK;
}
extern "C" int chipmunk_host_free(void *host_ptr) {
    if (!host_ptr) return CHIPMUNK_OK;
    {
        std::lock_guard<std::mutex> lock(g_host_mu);
        auto it = g_host_allocs.find(host_ptr);
        CM_CHECK(it != g_host_allocs.end(), "chipmunk_host_free: %p was not allocated by chipmunk_host_alloc", host_ptr);
        g_host_total -= it->second;
        g_host_allocs.erase(it);
    }
    (void)hipHostFree(host_ptr);
    return CHIPMUNK_OK;
}
extern "C" size_t chipmunk_host_bytes(void) {
    std::lock_guard<std::mutex> lock(g_host_mu);
    return g_host_total;
}
static int copy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, void *stream, const char *what) {
    CM_CHECK(dst && src, "%s: null pointer", what);
    if (bytes == 0) return CHIPMUNK_OK;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)stream);
    if (e != hipSuccess) {
        chipmunk_set_error("%s: hipMemcpyAsync of %zu bytes failed: %s", what, bytes, hipGetErrorString(e));
        return CHIPMUNK_ERR_LAUNCH;
    }
    return CHIPMUNK_OK;
}
extern "C" int chipmunk_copy_d2h_async(void *host_dst, const void *dev_src, size_t bytes, void *stream) {
    return copy_async(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, stream, "chipmunk_copy_d2h_async");
}
extern "C" int chipmunk_copy_h2d_async(void *dev_dst, const void *host_src, size_t bytes, void *stream) {
    return copy_async(dev_dst, host_src, bytes, hipMemcpyHostToDevice, stream, "chipmunk_copy_h2d_async");
}

extern "C" int chipmunk_release_scratch(void) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    (void)hipDeviceSynchronize();
    for (auto &kv : g_big)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    g_big.clear();
    g_big_failed.clear();
    g_big_fallbacks = 0;
    for (auto &kv : g_scratch)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    g_scratch.clear();
    return CHIPMUNK_OK;
}

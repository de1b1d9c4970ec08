// mvx_api.hip -- device/runtime helpers of the C ABI (no pixel work here).
#include "mvx_common.h"

extern "C" __attribute__((visibility("default"))) const char *mvx_version(void) { return "mvtools_amd 0.1 (gfx950; mvtools v24 hot path)"; }

extern "C" __attribute__((visibility("default"))) int mvx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { mvx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return MVX_E_DEVICE; }
    return n;
}

extern "C" __attribute__((visibility("default"))) int mvx_set_device(int ordinal) { HIP_CHECK(hipSetDevice(ordinal)); return MVX_OK; }

// ---- device memory for hosts without HIP headers.  hipMalloc / hipFree are slow and hipFree synchronises the whole device, which
// would serialise every concurrently running filter of a frame server; released buffers therefore go to a free list keyed by their
// exact size (a frame server asks for the same few sizes over and over) and are handed out again.  The list is bounded
// (mvx_dev_pool_limit, default 24 GiB); beyond it buffers are really freed.
#include <map>
#include <unordered_map>
static std::mutex g_pool_mu;
typedef std::pair<int, size_t> PoolKey; // (device ordinal, bytes): a host that switches devices (mvx_set_device) never gets another GPU's buffer
static std::multimap<PoolKey, void *> g_pool_free;
static std::unordered_map<void *, PoolKey> g_pool_size;
static size_t g_pool_held = 0, g_pool_limit = (size_t)24 << 30;
static int pool_device() { int d = 0; (void)hipGetDevice(&d); return d; }

static void *pool_take(size_t bytes) {
    const PoolKey key(pool_device(), bytes);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_free.find(key);
    if (it == g_pool_free.end()) return nullptr;
    void *p = it->second;
    g_pool_free.erase(it);
    g_pool_held -= bytes;
    return p;
}
extern "C" __attribute__((visibility("default"))) void *mvx_dev_alloc_uninit(size_t bytes) {
    if (!bytes) bytes = 1;
    void *p = pool_take(bytes);
    if (p) return p;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        { // give the pool back and retry once
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (auto &e : g_pool_free) { (void)hipFree(e.second); g_pool_size.erase(e.second); }
            g_pool_free.clear(); g_pool_held = 0;
        }
        if (hipMalloc(&p, bytes) != hipSuccess) { mvx_set_error("hipMalloc(%zu) failed", bytes); return nullptr; }
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_size[p] = PoolKey(pool_device(), bytes);
    return p;
}
extern "C" __attribute__((visibility("default"))) void *mvx_dev_alloc(size_t bytes) { // zero-filled
    void *p = mvx_dev_alloc_uninit(bytes);
    if (!p) return nullptr;
    // (on the legacy default stream and waited for: the caller may touch the buffer from any stream next)
    if (hipMemsetAsync(p, 0, bytes ? bytes : 1, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { mvx_set_error("hipMemset failed"); mvx_dev_free(p); return nullptr; }
    return p;
}
// The caller must have waited for the work that uses the buffer (every host in this repository synchronises its stream before it
// releases per-frame buffers).
extern "C" __attribute__((visibility("default"))) void mvx_dev_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_size.find(p);
        if (it != g_pool_size.end() && g_pool_held + it->second.second <= g_pool_limit) {
            g_pool_free.emplace(it->second, p);
            g_pool_held += it->second.second;
            return;
        }
        if (it != g_pool_size.end()) g_pool_size.erase(it);
    }
    (void)hipFree(p);
}
// free / total bytes of the current device, the free list counted as free (a host sizes its caches from this)
extern "C" __attribute__((visibility("default"))) int mvx_dev_mem_info(size_t *free_bytes, size_t *total_bytes) {
    size_t f = 0, t = 0;
    HIP_CHECK(hipMemGetInfo(&f, &t));
    { std::lock_guard<std::mutex> lk(g_pool_mu); f += g_pool_held; }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MVX_OK;
}
// gives every buffer of the free list back to the driver (a host calls this before it retries a failed allocation of another size)
extern "C" __attribute__((visibility("default"))) void mvx_dev_pool_trim(void) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &e : g_pool_free) { (void)hipFree(e.second); g_pool_size.erase(e.second); }
    g_pool_free.clear(); g_pool_held = 0;
}
extern "C" __attribute__((visibility("default"))) void mvx_dev_pool_limit(size_t bytes) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_limit = bytes;
    while (g_pool_held > g_pool_limit && !g_pool_free.empty()) {
        auto it = std::prev(g_pool_free.end());
        g_pool_held -= it->first.second;
        g_pool_size.erase(it->second);
        (void)hipFree(it->second);
        g_pool_free.erase(it);
    }
}

// a stream of its own per filter instance lets the launches of different instances overlap (non-blocking: no implicit
// synchronisation with the default stream -- the caller orders its uploads with mvx_stream_sync)
extern "C" __attribute__((visibility("default"))) void *mvx_stream_create(void) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { mvx_set_error("hipStreamCreate failed"); return nullptr; }
    return (void *)s;
}
// level < 0: lowest priority, > 0: highest, 0: default.  Streams of different priorities do not share a hardware queue, so the short
// per-frame kernels and copies of a frame server (high) are not stuck behind a search launch that runs for hundreds of milliseconds (low).
extern "C" __attribute__((visibility("default"))) void *mvx_stream_create_priority(int level) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
    hipStream_t s = nullptr;
    const int prio = level < 0 ? least : (level > 0 ? greatest : (least + greatest) / 2);
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) { mvx_set_error("hipStreamCreateWithPriority failed"); return nullptr; }
    return (void *)s;
}
extern "C" __attribute__((visibility("default"))) void mvx_stream_destroy(void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }

// page-locked host memory: the target of asynchronous device-to-host copies (mvx_copy_to_host) that must not stall the caller
extern "C" __attribute__((visibility("default"))) void *mvx_host_alloc_pinned(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { mvx_set_error("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
extern "C" __attribute__((visibility("default"))) void mvx_host_free_pinned(void *p) { if (p) (void)hipHostFree(p); }

extern "C" __attribute__((visibility("default"))) int mvx_copy_to_device(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t row_bytes, size_t rows, void *stream) {
    HIP_CHECK(hipMemcpy2DAsync(dst, dp, src, sp, row_bytes, rows, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) int mvx_copy_to_host(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t row_bytes, size_t rows, void *stream) {
    HIP_CHECK(hipMemcpy2DAsync(dst, dp, src, sp, row_bytes, rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MVX_OK;
}
// ---- synchronous 2-D transfers between PAGEABLE host memory (a frame server owns its frames) and the device, through pinned staging
// buffers: hipMemcpy2D on pageable memory moves a 131 MB super frame at a few GB/s, a linear copy into pinned memory runs at PCIe
// speed and the row-by-row repacking is an ordinary memcpy that the caller's threads do in parallel.  r3: the staging buffers have ONE
// fixed size (16 MiB) and a transfer goes through two of them in turn, chunk by chunk -- the DMA of one chunk runs under the memcpy of the
// next.  (Buffers that grew to the largest request were re-allocated until all 64 of them held 88 MB: 5 s of hipHostMalloc / hipHostFree
// once per process, during which every other HIP call of the frame server stalled.)  The pool is bounded; callers wait for a free buffer.
#include <condition_variable>
#include <cstring>
namespace {
constexpr size_t kStageBytes = (size_t)16 << 20;
struct Stage { void *p = nullptr; bool busy = false; };
constexpr int kStages = 96; // two per transfer in flight: a frame server runs tens of request threads plus the window builders
Stage g_stage[kStages];
std::mutex g_stage_mu;
std::condition_variable g_stage_cv;
// one or two buffers at once: a transfer that took its first buffer and then waited for the second could wait forever once every buffer
// of the pool was some transfer's first
bool stage_acquire_n(Stage **out, int n) {
    out[0] = out[1] = nullptr;
    {
        std::unique_lock<std::mutex> lk(g_stage_mu);
        g_stage_cv.wait(lk, [&] {
            int have = 0;
            Stage *pick[2] = { nullptr, nullptr };
            for (auto &t : g_stage) if (!t.busy && t.p && have < n) pick[have++] = &t;   // allocated ones first
            for (auto &t : g_stage) if (!t.busy && !t.p && have < n) pick[have++] = &t;
            if (have < n) return false;
            for (int i = 0; i < n; i++) { pick[i]->busy = true; out[i] = pick[i]; }
            return true;
        });
    }
    for (int i = 0; i < n; i++)
        if (!out[i]->p && hipHostMalloc(&out[i]->p, kStageBytes, hipHostMallocDefault) != hipSuccess) {
            out[i]->p = nullptr;
            mvx_set_error("hipHostMalloc(%zu) failed", kStageBytes);
            { std::lock_guard<std::mutex> g(g_stage_mu); for (int k = 0; k < n; k++) out[k]->busy = false; }
            g_stage_cv.notify_all();
            out[0] = out[1] = nullptr;
            return false;
        }
    return true;
}
void stage_release(Stage *s) {
    if (!s) return;
    { std::lock_guard<std::mutex> g(g_stage_mu); s->busy = false; }
    g_stage_cv.notify_all(); // (waiters need one OR two buffers: a single wake-up may go to one that cannot proceed)
}
// rows of a transfer that fit one staging buffer (device pitch dp; the last row of a chunk needs row_bytes only)
size_t chunk_rows(ptrdiff_t dp, size_t row_bytes) { const size_t n = (kStageBytes - row_bytes) / (size_t)dp + 1; return n ? n : 1; }
}
__global__ void mvx_noop_kernel(int *p) { if (p) *p = 1; }
// r6: see mvtools_amd.h.  (The staging buffers it page-locks are the pool's own: taken and given back one by one, so a transfer that runs meanwhile is not starved)
extern "C" __attribute__((visibility("default"))) int mvx_warmup(int staging_buffers) {
    if (hipFree(nullptr) != hipSuccess) { mvx_set_error("mvx_warmup: no usable HIP device"); return MVX_E_DEVICE; }
    if (staging_buffers < 0) return MVX_OK; // (the runtime only)
    hipLaunchKernelGGL(mvx_noop_kernel, dim3(1), dim3(64), 0, 0, (int *)nullptr); // (the first launch loads the library's code objects)
    if (hipDeviceSynchronize() != hipSuccess) { mvx_set_error("mvx_warmup: first launch failed"); return MVX_E_DEVICE; }
    if (staging_buffers > kStages) staging_buffers = kStages;
    for (int i = 0; i < staging_buffers; i++) {
        Stage *s = nullptr;
        {
            std::lock_guard<std::mutex> g(g_stage_mu);
            for (auto &t : g_stage) if (!t.busy && !t.p) { t.busy = true; s = &t; break; }
        }
        if (!s) break; // every buffer exists (or is in use)
        if (hipHostMalloc(&s->p, kStageBytes, hipHostMallocDefault) != hipSuccess) s->p = nullptr;
        stage_release(s);
    }
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) int mvx_upload_2d(void *dev, ptrdiff_t dp, const void *host, ptrdiff_t hp, size_t row_bytes, size_t rows, void *stream) {
    if (!rows || !row_bytes) return MVX_OK;
    if (dp < (ptrdiff_t)row_bytes) { mvx_set_error("mvx_upload_2d: device pitch smaller than a row"); return MVX_E_ARG; }
    if (row_bytes > kStageBytes) { // a row longer than a staging buffer (the shell moves whole vector blobs as one row): segment by segment
        for (size_t r = 0; r < rows; r++) {
            char *d = (char *)dev + r * (size_t)dp;
            const char *h = (const char *)host + (ptrdiff_t)r * hp;
            const size_t full = row_bytes / kStageBytes, tail = row_bytes % kStageBytes;
            int rc = mvx_upload_2d(d, (ptrdiff_t)kStageBytes, h, (ptrdiff_t)kStageBytes, kStageBytes, full, stream);
            if (rc == MVX_OK && tail) rc = mvx_upload_2d(d + full * kStageBytes, (ptrdiff_t)tail, h + full * kStageBytes, (ptrdiff_t)tail, tail, 1, stream);
            if (rc != MVX_OK) return rc;
        }
        return MVX_OK;
    }
    const size_t per = chunk_rows(dp, row_bytes);
    Stage *st[2];
    if (!stage_acquire_n(st, rows > per ? 2 : 1)) return MVX_E_DEVICE;
    auto fill = [&](Stage *s, size_t r0, size_t n) {
        for (size_t r = 0; r < n; r++) {
            memcpy((char *)s->p + r * (size_t)dp, (const char *)host + (ptrdiff_t)(r0 + r) * hp, row_bytes);
            // the pitch padding travels with the linear copy: zero it (a staging buffer holds whatever an earlier transfer left)
            if (r + 1 < n && (size_t)dp > row_bytes) memset((char *)s->p + r * (size_t)dp + row_bytes, 0, (size_t)dp - row_bytes);
        }
    };
    hipError_t e = hipSuccess;
    int cur = 0;
    size_t r0 = 0, n = rows < per ? rows : per;
    fill(st[0], 0, n);
    while (e == hipSuccess && r0 < rows) {
        e = hipMemcpyAsync((char *)dev + r0 * (size_t)dp, st[cur]->p, (n - 1) * (size_t)dp + row_bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
        const size_t r1 = r0 + n, n1 = r1 < rows ? (rows - r1 < per ? rows - r1 : per) : 0;
        if (n1) fill(st[cur ^ 1], r1, n1); // (under the DMA of the chunk just queued)
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        r0 = r1; n = n1; cur ^= 1;
    }
    if (e != hipSuccess) (void)hipStreamSynchronize((hipStream_t)stream); // (a copy queued from a staging buffer must be over before the buffer is handed on)
    stage_release(st[0]); stage_release(st[1]);
    if (e != hipSuccess) { mvx_set_error("mvx_upload_2d: %s", hipGetErrorString(e)); return MVX_E_DEVICE; }
    return MVX_OK;
}
// (work enqueued on `stream` before the call -- the kernels that produce the data -- is complete when the copy runs)
extern "C" __attribute__((visibility("default"))) int mvx_download_2d(void *host, ptrdiff_t hp, const void *dev, ptrdiff_t dp, size_t row_bytes, size_t rows, void *stream) {
    if (!rows || !row_bytes) return MVX_OK;
    if (dp < (ptrdiff_t)row_bytes) { mvx_set_error("mvx_download_2d: device pitch smaller than a row"); return MVX_E_ARG; }
    if (row_bytes > kStageBytes) { // (as mvx_upload_2d)
        for (size_t r = 0; r < rows; r++) {
            const char *d = (const char *)dev + r * (size_t)dp;
            char *h = (char *)host + (ptrdiff_t)r * hp;
            const size_t full = row_bytes / kStageBytes, tail = row_bytes % kStageBytes;
            int rc = mvx_download_2d(h, (ptrdiff_t)kStageBytes, d, (ptrdiff_t)kStageBytes, kStageBytes, full, stream);
            if (rc == MVX_OK && tail) rc = mvx_download_2d(h + full * kStageBytes, (ptrdiff_t)tail, d + full * kStageBytes, (ptrdiff_t)tail, tail, 1, stream);
            if (rc != MVX_OK) return rc;
        }
        return MVX_OK;
    }
    const size_t per = chunk_rows(dp, row_bytes);
    Stage *st[2];
    if (!stage_acquire_n(st, rows > per ? 2 : 1)) return MVX_E_DEVICE;
    auto bytes_of = [&](size_t n) { return (n - 1) * (size_t)dp + row_bytes; };
    hipError_t e = hipSuccess;
    int cur = 0;
    size_t r0 = 0, n = rows < per ? rows : per;
    e = hipMemcpyAsync(st[0]->p, dev, bytes_of(n), hipMemcpyDeviceToHost, (hipStream_t)stream);
    while (e == hipSuccess && r0 < rows) {
        e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) break;
        const size_t r1 = r0 + n, n1 = r1 < rows ? (rows - r1 < per ? rows - r1 : per) : 0;
        if (n1) e = hipMemcpyAsync(st[cur ^ 1]->p, (const char *)dev + r1 * (size_t)dp, bytes_of(n1), hipMemcpyDeviceToHost, (hipStream_t)stream);
        for (size_t r = 0; r < n; r++) memcpy((char *)host + (ptrdiff_t)(r0 + r) * hp, (const char *)st[cur]->p + r * (size_t)dp, row_bytes); // (under the DMA of the next chunk)
        r0 = r1; n = n1; cur ^= 1;
    }
    if (e != hipSuccess) (void)hipStreamSynchronize((hipStream_t)stream);
    stage_release(st[0]); stage_release(st[1]);
    if (e != hipSuccess) { mvx_set_error("mvx_download_2d: %s", hipGetErrorString(e)); return MVX_E_DEVICE; }
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) int mvx_dev_memset(void *dev, int value, size_t bytes, void *stream) {
    HIP_CHECK(hipMemsetAsync(dev, value, bytes, (hipStream_t)stream));
    return MVX_OK;
}

extern "C" __attribute__((visibility("default"))) int mvx_stream_sync(void *stream) { HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); return MVX_OK; }

/*
 * mvtools_vs.c -- VapourSynth API-4 filter shell over libmvtools_amd.so (the C ABI in include/mvtools_amd.h).
 *
 * Registers, under the reference's plugin id / namespace ("com.nodame.mvtools", "mv"), the filters of the hot path
 * with the reference's exact argument strings:
 *     Super       (src/MVSuper.c:279-291)        Analyse   (src/MVAnalyse.c:639-671)
 *     Degrain1..6 (src/MVDegrains.cpp:813-932)   Compensate (src/MVCompensate.c:579-592)   BlockFPS (src/MVBlockFPS.c:1017-1033)
 *     Recalculate (src/MVRecalculate.c:549-572)   Finest (src/MVFinest.c:213-218)   SCDetection (src/MVSCDetection.c:137-145)
 * and keeps the reference's inter-filter data layout: super-frame geometry + Super_* props on frame 0
 * (src/MVSuper.c:111-120), vector clips = copyFrame(super[n]) + binary props MVTools_MVAnalysisData / MVTools_vectors
 * (src/MVAnalyse.c:224-239).  This file is the only code that touches VSAPI; all arithmetic happens on the GPU behind
 * the C ABI, and every compute failure surfaces through setFilterError (there is no CPU path).
 *
 * Super frames produced here stay resident in a device-side cache (keyed by a per-frame id prop and checked against a
 * fingerprint of the host frame) so that Analyse / Degrain / Compensate do not round-trip 131 MB pyramids through host
 * memory; frames that did not come from this Super, or that another filter modified, are uploaded on demand.
 * mv.Analyse batches: the getFrame calls that VapourSynth's worker threads make concurrently (every filter here is
 * fmParallel, like the reference: MVAnalyse.c:634) are collected by a combining queue into ONE mvx_analyse_frames launch per
 * instance on the instance's own stream -- the search is a serial chain per frame whose throughput comes from the number
 * of chains in flight.  The other filters make one C-ABI call per frame; all handles are thread-safe.
 *
 * Not supported (fail loudly at creation, like the C ABI): dct 1..4.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vs4_api.h"
#include "vs4_api_check.h"
#include "../../include/mvtools_amd.h"

#define PROP_ADATA "MVTools_MVAnalysisData"
#define PROP_VECTORS "MVTools_vectors"
#define PROP_SUPER_ID "_MVX_SuperId"

/* ------------------------------------------------------------------------------------------------ device frame cache */

typedef struct DevFrame { int64_t id; void *arena; void *plane[3]; size_t bytes; int pins; uint64_t stamp; uint64_t print;
                         int trusted; /* built by this plugin's own mv.Super (superGetFrame / super_build_device), not uploaded from a host frame that merely carries the id */ } DevFrame;
#define CACHE_MAX 2048
static DevFrame g_cache[CACHE_MAX];
static int g_cache_cap = -1;       /* entries (MVX_VS_CACHE_FRAMES), at most CACHE_MAX */
static size_t g_cache_budget;      /* bytes the cached frames may hold together; 0 = not fixed yet */
static size_t g_cache_bytes;       /* bytes they hold now */
static uint64_t g_stamp = 1;
static int64_t g_next_instance = 1;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

/* ONE high-priority non-blocking stream carries every thread's uploads, per-frame kernels (Super, Degrain, ...) and downloads: that
 * work is short and PCIe is serial anyway; the searches run on low-priority streams of their own (one per Analyse instance) and a
 * different priority means a different hardware queue, so a frame's copies never wait behind a 0.3 s search launch.  (A stream per
 * worker thread was measured and is worse: 40 ms to create each, and they share hardware queues with the searches.) */
static void *g_frame_stream;
static int g_frame_stream_tried;
/* MVX_VS_STATS=1: thread-seconds by category, printed with the launch statistics when the plugin is unloaded */
enum { PF_STREAM, PF_UPLOAD, PF_DOWNLOAD, PF_SUPER, PF_SEARCH_WAIT, PF_DEGRAIN, PF_ALLOC, PF_LA_BLOCKED, PF_BUILD_LOCK, PF_GF_SUPER, PF_GF_ANALYSE, PF_GF_DEGRAIN, PF_LA_BUILD, PF_LA_ARENA, PF_LA_LAUNCH, PF_LA_SRCUP, PF_N };
static double g_prof[PF_N];
static long g_prof_n[PF_N];
static int g_prof_on = -1;
static double prof_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static void prof_add(int k, double t0) {
    if (g_prof_on < 0) g_prof_on = getenv("MVX_VS_STATS") != NULL;
    if (!g_prof_on) return;
    const double dt = prof_now() - t0;
    pthread_mutex_lock(&g_lock); g_prof[k] += dt; g_prof_n[k]++; pthread_mutex_unlock(&g_lock);
}
/* MVX_VS_TRACE=1: one line on stderr per look-ahead window event (seconds since the first event) */
static int g_trace_on = -1;
static double g_trace_t0;
static void la_trace(const void *inst, int w, const char *what, double dur) {
    if (g_trace_on < 0) { g_trace_on = getenv("MVX_VS_TRACE") != NULL; g_trace_t0 = prof_now(); }
    if (g_trace_on) fprintf(stderr, "mvtools_vs trace %8.3f  analyse %p window %d %s (%.3f s)\n", prof_now() - g_trace_t0, inst, w, what, dur);
}
static void prof_add_locked_ok(int k, double t0) { prof_add(k, t0); } /* (g_lock is a different mutex than the callers hold) */
/* The per-frame work of a worker thread (its uploads, its Super / Degrain / ... kernels) runs on ONE stream of a small pool
 * (MVX_VS_FRAME_STREAMS, default 4; r3 had ONE for all threads: a thread's kernel then waited behind every other thread's uploads), chosen per
 * thread the first time it asks.  Every getFrame waits for its own stream before it publishes or releases anything, so the streams need no
 * ordering among each other.  640 4K16 frames, lazy super frames: 405 fps with one stream, 481 with four, 508 with eight
 * (profiles/r4_vs_shell_frame_streams.txt); the default mode is bound by its super-frame downloads and does not change. */
#define FRAME_STREAMS_MAX 8
static void *g_frame_streams[FRAME_STREAMS_MAX];
static int g_frame_nstreams, g_frame_next;
static __thread int t_frame_stream = -1;
static void *thread_stream(void) {
    if (!__atomic_load_n(&g_frame_stream_tried, __ATOMIC_ACQUIRE)) {
        pthread_mutex_lock(&g_lock);
        if (!g_frame_stream_tried) {
            const char *e = getenv("MVX_VS_FRAME_STREAMS");
            int n = e ? atoi(e) : 4;
            n = n < 1 ? 1 : n > FRAME_STREAMS_MAX ? FRAME_STREAMS_MAX : n;
            for (int i = 0; i < n; i++) g_frame_streams[i] = mvx_stream_create_priority(1);
            g_frame_stream = g_frame_streams[0];
            g_frame_nstreams = n;
            __atomic_store_n(&g_frame_stream_tried, 1, __ATOMIC_RELEASE);
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (g_frame_nstreams <= 1) return g_frame_stream;
    if (t_frame_stream < 0) t_frame_stream = __atomic_fetch_add(&g_frame_next, 1, __ATOMIC_RELAXED) % g_frame_nstreams;
    return g_frame_streams[t_frame_stream] ? g_frame_streams[t_frame_stream] : g_frame_stream;
}
/* the 131 MB downloads of finished super frames (mv.Super's host frames) run on a stream of their own: they only depend on kernels that
 * were waited for already, and on their own stream they overlap the other threads' uploads (PCIe is full duplex) instead of queueing
 * in front of them */
/* r6: MVX_VS_DL_STREAMS of them (default 2), taken in turn: one stream is one DMA engine's worth of device-to-host bandwidth (~45 GB/s measured), and the 138 MB of super frame
 * per output frame that mv.Super's default mode sends to the host make that engine the bound of the whole graph */
#define DL_STREAMS_MAX 8
static void *g_dl_stream, *g_dl_streams[DL_STREAMS_MAX];
static int g_dl_stream_tried, g_dl_nstreams;
static unsigned g_dl_next;
static void *download_stream(void) {
    if (!__atomic_load_n(&g_dl_stream_tried, __ATOMIC_ACQUIRE)) {
        pthread_mutex_lock(&g_lock);
        if (!g_dl_stream_tried) {
            const char *e = getenv("MVX_VS_DL_STREAMS");
            int want = e ? atoi(e) : 2;
            if (want < 1) want = 1;
            if (want > DL_STREAMS_MAX) want = DL_STREAMS_MAX;
            for (int i = 0; i < want; i++) { void *st = mvx_stream_create_priority(1); if (st) g_dl_streams[g_dl_nstreams++] = st; }
            g_dl_stream = g_dl_nstreams ? g_dl_streams[0] : NULL;
            __atomic_store_n(&g_dl_stream_tried, 1, __ATOMIC_RELEASE);
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (!g_dl_nstreams) return thread_stream();
    return g_dl_streams[__atomic_fetch_add(&g_dl_next, 1u, __ATOMIC_RELAXED) % (unsigned)g_dl_nstreams];
}
/* Error paths: a getFrame that failed after it enqueued work skipped its stream waits, and mvx_dev_free hands a buffer straight to the next
 * caller (no stream ordering in the pool: ADVICE r2) -- so before such a path frees device memory it waits for whatever is still queued on the
 * streams it used.  A no-op when rc == 0. */
static void shell_quiesce(int rc) {
    if (!rc) return;
    (void)mvx_stream_sync(thread_stream());
    for (int i = 0; i < g_dl_nstreams; i++) (void)mvx_stream_sync(g_dl_streams[i]);
}
static int timed_download_on(void *stream, void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows) {
    const double t0 = prof_now();
    const int rc = mvx_download_2d(dst, dp, src, sp, rb, rows, stream);
    prof_add(PF_DOWNLOAD, t0);
    return rc;
}
static int timed_upload(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows) {
    const double t0 = prof_now();
    const int rc = mvx_upload_2d(dst, dp, src, sp, rb, rows, thread_stream());
    prof_add(PF_UPLOAD, t0);
    return rc;
}
static int timed_download(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows) {
    const double t0 = prof_now();
    const int rc = mvx_download_2d(dst, dp, src, sp, rb, rows, thread_stream());
    prof_add(PF_DOWNLOAD, t0);
    return rc;
}

/* The device cache is bounded in BYTES: MVX_VS_CACHE_BYTES, else 40 % of what the device has free when the first frame arrives (115 GB
 * of an idle MI355X; less on a smaller or shared GPU) -- and in entries (MVX_VS_CACHE_FRAMES, CACHE_MAX).  The budget limits what is
 * KEPT for later: frames somebody has pinned (a getFrame in progress, the windows of a look-ahead search) are held whatever it says,
 * and unpinned ones are evicted least recently used first to make room.  It should cover the frames a host has in flight (its thread
 * count + twice the temporal radius; with look-ahead two windows per vector clip): a frame that falls out is built or uploaded again.
 * Super clips of different sizes share the budget; when a device allocation fails, unpinned entries are evicted before the shell gives
 * up (shell_alloc). */
static int cache_cap(void) {
    if (g_cache_cap < 0) {
        const char *e = getenv("MVX_VS_CACHE_FRAMES");
        g_cache_cap = e ? atoi(e) : CACHE_MAX;
        if (g_cache_cap > CACHE_MAX) g_cache_cap = CACHE_MAX;
        if (g_cache_cap < 0) g_cache_cap = 0;
    }
    return g_cache_cap;
}
static size_t cache_budget(void) { /* g_lock held */
    if (!g_cache_budget) {
        const char *b = getenv("MVX_VS_CACHE_BYTES");
        double budget = b ? atof(b) : 48.0 * 1024 * 1024 * 1024;
        size_t fr = 0, tot = 0;
        if (!b && mvx_dev_mem_info(&fr, &tot) == 0) budget = 0.4 * (double)fr;
        g_cache_budget = budget < 1.0 ? 1 : (size_t)budget;
        /* evicted arenas must be REUSED, not freed: hipFree synchronises the whole device, i.e. waits for the search launches in flight (0.5 s
         * each) -- a window of 128 new 4K16 super frames evicts 34 GB, more than the pool's default 24 GiB holds, and the frees that followed
         * stalled every request thread for seconds.  The free list may therefore hold a quarter of the device memory. */
        if (tot || mvx_dev_mem_info(&fr, &tot) == 0) mvx_dev_pool_limit(tot / 4);
    }
    return g_cache_budget;
}
/* drops the least recently used unpinned entry; returns its arena (the caller frees it outside the lock) or NULL.  g_lock held. */
static void *cache_evict_lru_locked(void) {
    DevFrame *v = NULL;
    for (int i = 0; i < cache_cap(); i++) {
        DevFrame *e = &g_cache[i];
        if (e->arena && e->pins == 0 && (!v || e->stamp < v->stamp)) v = e;
    }
    if (!v) return NULL;
    void *a = v->arena;
    v->arena = NULL; g_cache_bytes -= v->bytes; v->bytes = 0;
    return a;
}
/* device memory for the shell: on failure the free list of the pool and then unpinned cache entries make room */
static void *shell_alloc(size_t bytes) {
    void *p = mvx_dev_alloc_uninit(bytes);
    while (!p) {
        pthread_mutex_lock(&g_lock);
        void *victim = cache_evict_lru_locked();
        pthread_mutex_unlock(&g_lock);
        if (!victim) break;
        mvx_dev_free(victim);
        mvx_dev_pool_trim(); /* the victim has another size: give it back to the driver */
        p = mvx_dev_alloc_uninit(bytes);
    }
    return p;
}

/* geometry of a super frame on the device: one arena per frame, planes at 256-byte pitches */
/* (every plane is followed by its shadow copies, mvx_super_shadow_frames: the search loads dword-aligned from them) */
typedef struct SuperGeo { const mvx_super *sup; mvx_super_info si; ptrdiff_t pitch[3], shadowStride[3]; size_t off[3]; size_t bytes; int bps, copies; } SuperGeo;

static void super_geo(SuperGeo *g, const mvx_super *s) {
    g->sup = s;
    mvx_super_get_info(s, &g->si);
    g->bps = (g->si.bits + 7) / 8;
    g->copies = 1 + mvx_super_shadow_copies(s);
    for (int p = 0; p < 3; p++) g->pitch[p] = p < g->si.num_planes ? ((ptrdiff_t)g->si.plane_width[p] * g->bps + 255) / 256 * 256 : 0;
    size_t extra[3];
    mvx_super_shadow_bytes(s, g->pitch, extra); /* room behind each plane for its shadow data */
    size_t o = 0;
    for (int p = 0; p < 3; p++) {
        g->off[p] = o; g->shadowStride[p] = 0;
        if (p < g->si.num_planes) {
            g->shadowStride[p] = ((ptrdiff_t)g->pitch[p] * g->si.plane_height[p] + 255) / 256 * 256;
            o += (size_t)g->shadowStride[p] + (extra[p] + 255) / 256 * 256;
        }
    }
    g->bytes = o;
}
/* the shifted copies behind freshly written planes (no-op for 8-bit clips) */
static int super_shadows(const SuperGeo *g, void *const plane[3], void *stream) {
    if (g->copies <= 1) return 0;
    return mvx_super_shadow_frames(g->sup, 1, plane, g->pitch, g->shadowStride, stream);
}
/* cheap fingerprint of a host super frame: a few rows of every plane.  Most filters copy frame props while changing pixels, so the
 * id prop alone does not prove that the cached device copy still matches the frame a consumer was handed. */
static uint64_t frame_print(const VSFrame *f, const SuperGeo *g, const VSAPI *vs) {
    uint64_t h = 1469598103934665603ULL;
    for (int p = 0; p < g->si.num_planes; p++) {
        const int H = g->si.plane_height[p];
        const size_t rb = (size_t)g->si.plane_width[p] * g->bps;
        const int rows[5] = { 0, H / 5, H / 2, (int)((int64_t)H * 4 / 5), H - 1 };
        for (int k = 0; k < 5; k++) {
            const uint8_t *r = vs->getReadPtr(f, p) + (ptrdiff_t)rows[k] * vs->getStride(f, p);
            for (size_t i = 0; i + 8 <= rb; i += 8) { uint64_t v; memcpy(&v, r + i, 8); h = (h ^ v) * 1099511628211ULL; }
        }
    }
    return h;
}

/* returns a pinned cache entry holding frame `id` with that fingerprint, or NULL.  own_bytes != 0: look-up by id alone, for entries this
 * plugin's own mv.Super built (`trusted`) with exactly that layout size -- a frame uploaded from the host under the same id (a filter
 * between mv.Super and its consumer may copy the properties and change the pixels) never answers it. */
static DevFrame *cache_find_ex(int64_t id, uint64_t print, size_t own_bytes) {
    DevFrame *r = NULL;
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < cache_cap(); i++) {
        DevFrame *e = &g_cache[i];
        if (e->arena && e->id == id && (own_bytes ? (e->trusted && e->bytes == own_bytes) : e->print == print)) { r = e; r->pins++; r->stamp = g_stamp++; break; }
    }
    pthread_mutex_unlock(&g_lock);
    return r;
}
static DevFrame *cache_find(int64_t id, uint64_t print) { return cache_find_ex(id, print, 0); }
/* drops the unpinned entries of one mv.Super instance (its free callback) */
static void cache_evict_instance(int64_t instance) {
    void *victims[CACHE_MAX];
    int nv = 0;
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < cache_cap(); i++)
        if (g_cache[i].arena && (g_cache[i].id >> 32) == instance && g_cache[i].pins == 0) { victims[nv++] = g_cache[i].arena; g_cache[i].arena = NULL; g_cache_bytes -= g_cache[i].bytes; g_cache[i].bytes = 0; }
    pthread_mutex_unlock(&g_lock);
    for (int i = 0; i < nv; i++) mvx_dev_free(victims[i]);
}
/* hands a freshly filled arena to the cache (pinned); returns NULL if the cache is full of pinned frames / disabled.  Entries are
 * evicted least recently used first until the new frame fits the byte budget. */
static DevFrame *cache_insert(int64_t id, uint64_t print, void *arena, const SuperGeo *g, int trusted) {
    DevFrame *slot = NULL;
    void *victims[CACHE_MAX];
    int nv = 0;
    pthread_mutex_lock(&g_lock);
    const size_t budget = cache_budget();
    for (int i = 0; i < cache_cap(); i++) { /* a stale unpinned copy of the same frame goes first */
        DevFrame *e = &g_cache[i];
        if (e->arena && e->id == id && e->pins == 0 && (trusted || !e->trusted)) { victims[nv++] = e->arena; e->arena = NULL; g_cache_bytes -= e->bytes; e->bytes = 0; } /* (an upload never displaces the genuine copy) */
    }
    while (g_cache_bytes + g->bytes > budget) {
        void *v = cache_evict_lru_locked();
        if (!v) break;
        victims[nv++] = v;
    }
    /* (still over the budget: everything left is pinned, i.e. needed right now -- the new frame is kept all the same) */
    for (int i = 0; i < cache_cap(); i++) if (!g_cache[i].arena) { slot = &g_cache[i]; break; }
    if (!slot && cache_cap() > 0) { /* every entry in use: recycle the least recently used unpinned one */
        void *v = cache_evict_lru_locked();
        if (v) { victims[nv++] = v; for (int i = 0; i < cache_cap(); i++) if (!g_cache[i].arena) { slot = &g_cache[i]; break; } }
    }
    if (slot) {
        slot->id = id; slot->print = print; slot->arena = arena; slot->bytes = g->bytes; slot->pins = 1; slot->stamp = g_stamp++; slot->trusted = trusted;
        g_cache_bytes += g->bytes;
        for (int p = 0; p < 3; p++) slot->plane[p] = p < g->si.num_planes ? (char *)arena + g->off[p] : NULL;
    }
    pthread_mutex_unlock(&g_lock);
    for (int i = 0; i < nv; i++) mvx_dev_free(victims[i]);
    return slot;
}
static void cache_unpin(DevFrame *e) {
    if (!e) return;
    pthread_mutex_lock(&g_lock);
    e->pins--;
    pthread_mutex_unlock(&g_lock);
}

/* a super frame on the device for the duration of one getFrame: from the cache or uploaded into a temporary arena */
typedef struct DevRef { DevFrame *cached; void *temp; void *plane[3]; } DevRef;

/* MVX_VS_SUPER_LAZY=1 (opt-in, r4): mv.Super hands out frames whose super pixels were never downloaded -- 138 MB per 4K16 frame, three quarters
 * of what the shell moves over PCIe per output frame.  Such a frame carries the SOURCE planes in the top-left corner of its planes (a host copy)
 * and the property MVX_super_lazy; the device copy lives in the cache, and a consumer that does not find it there rebuilds it on the device from
 * the embedded source with the mv.Super instance's own handle.  Only this plugin's filters understand such frames: any other consumer of the
 * super clip would see the source picture in a corner of an otherwise undefined frame -- hence opt-in. */
#define PROP_SUPER_LAZY "MVX_super_lazy"
static int super_lazy(void) { static int v = -1; if (v < 0) { const char *e = getenv("MVX_VS_SUPER_LAZY"); v = e && atoi(e) != 0; } return v; }
static int super_rebuild_lazy(DevRef *r, const VSFrame *f, int64_t id, uint64_t print, const SuperGeo *want, const VSAPI *vs);

static int super_to_device(DevRef *r, const VSFrame *f, const SuperGeo *g, const VSAPI *vs) {
    memset(r, 0, sizeof(*r));
    int err = 0;
    const int64_t id = vs->mapGetInt(vs->getFramePropertiesRO(f), PROP_SUPER_ID, 0, &err);
    const uint64_t print = frame_print(f, g, vs);
    if (!err && (r->cached = cache_find(id, print)) != NULL) {
        for (int p = 0; p < 3; p++) r->plane[p] = r->cached->plane[p];
        return 0;
    }
    {
        int lerr = 0;
        if (!err && vs->mapGetInt(vs->getFramePropertiesRO(f), PROP_SUPER_LAZY, 0, &lerr) == 1 && !lerr) return super_rebuild_lazy(r, f, id, print, g, vs);
    }
    void *arena = shell_alloc(g->bytes);
    if (!arena) return MVX_E_NOMEM;
    void *st = thread_stream();
    int rc = mvx_dev_memset(arena, 0, g->bytes, st); /* zero-filled like the frames mv.Super builds (pitch padding) */
    for (int p = 0; p < g->si.num_planes && !rc; p++) {
        void *d = (char *)arena + g->off[p];
        rc = timed_upload(d, g->pitch[p], vs->getReadPtr(f, p), vs->getStride(f, p), (size_t)g->si.plane_width[p] * g->bps, (size_t)g->si.plane_height[p]);
        r->plane[p] = d;
    }
    if (!rc) rc = super_shadows(g, r->plane, st);
    if (!rc) rc = mvx_stream_sync(st); /* consumers launch on their own streams */
    if (rc) { shell_quiesce(rc); mvx_dev_free(arena); memset(r, 0, sizeof(*r)); return rc; }
    if (!err && (r->cached = cache_insert(id, print, arena, g, 0)) != NULL) return 0; /* keep it for the next consumer (found again by id AND fingerprint only) */
    r->temp = arena;
    return 0;
}
static void dev_release(DevRef *r) {
    if (r->cached) cache_unpin(r->cached);
    if (r->temp) mvx_dev_free(r->temp);
    memset(r, 0, sizeof(*r));
}

/* ------------------------------------------------------------------------------------------------ small helpers */

static int32_t opt_int(const VSMap *in, const char *key, const VSAPI *vs) {
    int err = 0;
    const int v = vs->mapGetIntSaturated(in, key, 0, &err);
    return err ? MVX_UNSET : v;
}
static int64_t opt_int64(const VSMap *in, const char *key, const VSAPI *vs) {
    int err = 0;
    const int64_t v = vs->mapGetInt(in, key, 0, &err);
    return err ? (int64_t)MVX_UNSET : v;
}

/* fields / tff (src/MVAnalyse.c:367-370, MVRecalculate.c:332-335, MVCompensate.c:435,453-454) */
typedef struct FieldOpt { int fields, tff, tffExists; } FieldOpt;

static void field_opt(FieldOpt *o, const VSMap *in, const VSAPI *vs) {
    int e;
    o->fields = !!vs->mapGetInt(in, "fields", 0, &e);
    o->tff = !!vs->mapGetInt(in, "tff", 0, &e);
    o->tffExists = !e;
}

/* parity of frame n: its _Field prop unless tff was passed (src/MVAnalyse.c:135-145); *missing is set when neither exists */
static int frame_top_field(const FieldOpt *o, const VSFrame *f, int n, int *missing, const VSAPI *vs) {
    int e;
    int top = !!vs->mapGetInt(vs->getFramePropertiesRO(f), "_Field", 0, &e);
    if (e && !o->tffExists) *missing = 1;
    if (o->tffExists) top = o->tff ^ (n % 2);
    return top;
}

static int field_shift_of(int srcTop, int refTop, int pel) { /* src/MVAnalyse.c:177 */
    return (srcTop && !refTop) ? pel / 2 : ((refTop && !srcTop) ? -(pel / 2) : 0);
}

/* Super_* props of frame 0 of a super clip -> a geometry-only mvx_super handle (consumers: src/MVAnalyse.c:519-553,
 * src/MVDegrains.cpp:556-581, src/MVCompensate.c:470-500).  `filter` prefixes the reference's messages. */
static mvx_super *super_from_props(VSNode *super, const char *filter, char *error, size_t esz, const VSAPI *vs) {
    char msg[1024];
    const VSFrame *f0 = vs->getFrame(0, super, msg, sizeof(msg));
    if (!f0) { snprintf(error, esz, "%s: failed to retrieve first frame from super clip. Error message: %s", filter, msg); return NULL; }
    const VSMap *props = vs->getFramePropertiesRO(f0);
    int e[6];
    const int height = vs->mapGetIntSaturated(props, "Super_height", 0, &e[0]);
    const int hpad = vs->mapGetIntSaturated(props, "Super_hpad", 0, &e[1]);
    const int vpad = vs->mapGetIntSaturated(props, "Super_vpad", 0, &e[2]);
    const int pel = vs->mapGetIntSaturated(props, "Super_pel", 0, &e[3]);
    const int modeyuv = vs->mapGetIntSaturated(props, "Super_modeyuv", 0, &e[4]);
    const int levels = vs->mapGetIntSaturated(props, "Super_levels", 0, &e[5]);
    vs->freeFrame(f0);
    for (int i = 0; i < 6; i++)
        if (e[i]) {
            snprintf(error, esz, "%s: required properties not found in first frame of super clip. Maybe clip didn't come from mv.Super? Was the first frame trimmed away?", filter);
            return NULL;
        }
    const VSVideoInfo *vi = vs->getVideoInfo(super);
    if (height <= 0 || hpad < 0 || hpad >= vi->width / 2 || vpad < 0 || pel < 1 || pel > 4 || modeyuv < 0 || modeyuv > 7 || levels < 1) {
        snprintf(error, esz, "%s: parameters from super clip appear to be wrong.", filter);
        return NULL;
    }
    mvx_super_args a;
    a.width = vi->width - 2 * hpad; a.height = height; a.bits = vi->format.bitsPerSample;
    a.subsampling_w = vi->format.subSamplingW; a.subsampling_h = vi->format.subSamplingH; a.gray = vi->format.colorFamily == cfGray;
    a.hpad = hpad; a.vpad = vpad; a.pel = pel; a.levels = levels; a.chroma = modeyuv != 1; a.sharp = MVX_UNSET; a.rfilter = MVX_UNSET;
    mvx_super *s = NULL;
    char err[MVX_ERRLEN];
    if (mvx_super_create(&a, &s, err)) { snprintf(error, esz, "%s: parameters from super clip appear to be wrong.", filter); return NULL; }
    return s;
}

/* src/MVAnalysisData.c:34-64 adataFromVectorClip */
static int adata_from_clip(mvx_analysis_data *ad, VSNode *clip, const char *filter, const char *name, char *error, size_t esz, const VSAPI *vs) {
    char msg[1024];
    const VSFrame *f0 = vs->getFrame(0, clip, msg, sizeof(msg));
    if (!f0) { snprintf(error, esz, "%s: Failed to retrieve first frame from %s. Error message: %s", filter, name, msg); return -1; }
    const VSMap *props = vs->getFramePropertiesRO(f0);
    int err = 0;
    const char *data = vs->mapGetData(props, PROP_ADATA, 0, &err);
    int rc = 0;
    if (err) { snprintf(error, esz, "%s: Property '%s' not found in first frame of %s.", filter, PROP_ADATA, name); rc = -1; }
    else {
        const int size = vs->mapGetDataSize(props, PROP_ADATA, 0, NULL);
        if (size != (int)sizeof(*ad)) { snprintf(error, esz, "%s: Property '%s' in first frame of %s has wrong size (%d instead of %d).", filter, PROP_ADATA, name, size, (int)sizeof(*ad)); rc = -1; }
        else memcpy(ad, data, sizeof(*ad));
    }
    vs->freeFrame(f0);
    return rc;
}
/* src/MVAnalysisData.c:67-99 adataCheckSimilarity */
static int adata_similar(const mvx_analysis_data *a, const mvx_analysis_data *b, const char *filter, const char *n1, const char *n2, char *error, size_t esz) {
    const char *what = NULL;
    if (a->nWidth != b->nWidth) what = "widths";
    if (a->nHeight != b->nHeight) what = "heights";
    if (a->nBlkSizeX != b->nBlkSizeX || a->nBlkSizeY != b->nBlkSizeY) what = "block sizes";
    if (a->nPel != b->nPel) what = "pel precision";
    if (a->nOverlapX != b->nOverlapX || a->nOverlapY != b->nOverlapY) what = "overlap";
    if (a->xRatioUV != b->xRatioUV) what = "horizontal subsampling";
    if (a->yRatioUV != b->yRatioUV) what = "vertical subsampling";
    if (a->bitsPerSample != b->bitsPerSample) what = "bit depths";
    if (!what) return 0;
    snprintf(error, esz, "%s: %s and %s have different %s.", filter, n1, n2, what);
    return -1;
}

/* MVTools_vectors of a vector-clip frame -> device.  The array states its own size in its first int (gopGetArraySize,
 * GroupOfPlanes.c:167-174); clips made with divide carry an extra array the level formula does not describe. */
static int blob_to_device(void **dblob, int *size, const mvx_analysis_data *ad, const VSFrame *vf, const VSAPI *vs) {
    int e = 0;
    const VSMap *props = vs->getFramePropertiesRO(vf);
    const char *blob = vs->mapGetData(props, PROP_VECTORS, 0, &e);
    *dblob = NULL;
    if (e) return MVX_E_ARG;
    const int n = vs->mapGetDataSize(props, PROP_VECTORS, 0, NULL);
    int stated = 0;
    if (n >= 8) memcpy(&stated, blob, sizeof(stated));
    if (n < 8 || stated != n) return MVX_E_ARG;
    /* the planes' own size headers must tile the array exactly (fgopUpdate walks them, Fakery.c:112-123): a spliced or truncated
     * property must not send the device readers out of bounds */
    int planes = 0, lastOff = 0;
    for (int off = 8; off < n; planes++) {
        int psz = 0;
        if (off + 4 > n) return MVX_E_ARG;
        memcpy(&psz, blob + off, sizeof(psz));
        if (psz < 4 + 16 || (psz - 4) % 16 || psz > n - off) return MVX_E_ARG;
        lastOff = off;
        off += psz;
    }
    if (ad) { /* the readers skip nLvCount - 1 planes and then read nBlkX * nBlkY vectors */
        int off = 8;
        for (int i = ad->nLvCount - 1; i >= 1 && off < n; i--) { int psz; memcpy(&psz, blob + off, sizeof(psz)); off += psz; }
        if (planes < ad->nLvCount || off > lastOff || n - off < 4 + ad->nBlkX * ad->nBlkY * 16) return MVX_E_ARG;
    }
    *dblob = shell_alloc((size_t)n);
    if (!*dblob) return MVX_E_NOMEM;
    if (timed_upload(*dblob, n, blob, n, (size_t)n, 1)) return MVX_E_DEVICE; /* complete on return: the prop memory goes away with the frame */
    if (size) *size = n;
    return 0;
}

/* the planes of a host frame -> one device arena.  Returns 0 or an MVX_E_* code; the copies are complete on return (the host frame
 * may be released right away) */
static int upload_plane_set(void *dst[3], void **arena, const VSFrame *f, const ptrdiff_t pitch[3], int nplanes, int bps, const VSAPI *vs) {
    size_t off[3], total = 0;
    for (int p = 0; p < nplanes; p++) { off[p] = total; total += (size_t)pitch[p] * vs->getFrameHeight(f, p); }
    *arena = shell_alloc(total);
    for (int p = 0; p < 3; p++) dst[p] = NULL;
    if (!*arena) return MVX_E_NOMEM;
    int rc = 0;
    for (int p = 0; p < nplanes && !rc; p++) {
        dst[p] = (char *)*arena + off[p];
        rc = timed_upload(dst[p], pitch[p], vs->getReadPtr(f, p), vs->getStride(f, p), (size_t)vs->getFrameWidth(f, p) * bps, (size_t)vs->getFrameHeight(f, p));
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------ mv.Super */

typedef struct SuperData { VSNode *node, *pelclip; VSVideoInfo vi; mvx_super *sup; SuperGeo geo; ptrdiff_t srcPitch[3], pelPitch[3]; int32_t pelMode; int64_t instance; } SuperData;

/* The output nodes of this plugin's own mv.Super filters.  A consumer whose `super` argument IS one of them (pointer identity: no
 * filter in between that could have changed pixels while copying the props) may build the super frames it needs on the device itself
 * from the 5x smaller SOURCE frames -- what mv.Analyse's look-ahead does -- and trust cached device frames by their id alone. */
#define SUPER_REG_MAX 64
static struct { VSNode *out; struct SuperData *d; } g_supers[SUPER_REG_MAX];
static void super_register(VSNode *out, SuperData *d) {
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < SUPER_REG_MAX; i++) if (!g_supers[i].out) { g_supers[i].out = out; g_supers[i].d = d; break; }
    pthread_mutex_unlock(&g_lock);
}
static void super_unregister(SuperData *d) {
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < SUPER_REG_MAX; i++) if (g_supers[i].d == d) { g_supers[i].out = NULL; g_supers[i].d = NULL; }
    pthread_mutex_unlock(&g_lock);
}
static SuperData *super_lookup(VSNode *out) {
    SuperData *d = NULL;
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < SUPER_REG_MAX; i++) if (g_supers[i].out == out) d = g_supers[i].d;
    pthread_mutex_unlock(&g_lock);
    return d;
}
static SuperData *super_by_instance(int64_t instance) {
    SuperData *d = NULL;
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < SUPER_REG_MAX; i++) if (g_supers[i].d && g_supers[i].d->instance == instance) d = g_supers[i].d;
    pthread_mutex_unlock(&g_lock);
    return d;
}
/* lazy super frames (MVX_VS_SUPER_LAZY): the source picture in the top-left corner of every plane; the rows the fingerprint samples are defined */
static void lazy_fill(VSFrame *dst, const VSFrame *src, const SuperGeo *g, const VSAPI *vs) {
    for (int p = 0; p < g->si.num_planes; p++) {
        const int H = g->si.plane_height[p];
        const size_t rb = (size_t)g->si.plane_width[p] * g->bps;
        const int rows[5] = { 0, H / 5, H / 2, (int)((int64_t)H * 4 / 5), H - 1 }; /* (= frame_print's) */
        uint8_t *dp = vs->getWritePtr(dst, p);
        const ptrdiff_t ds = vs->getStride(dst, p), ss = vs->getStride(src, p);
        /* the whole plane is defined (zero outside the embedded source): newVideoFrame hands out uninitialised heap, and a consumer that is
         * not this plugin must neither see stale memory nor a frame that differs from run to run */
        for (int y = 0; y < H; y++) memset(dp + (ptrdiff_t)y * ds, 0, rb);
        (void)rows;
        const uint8_t *sp = vs->getReadPtr(src, p);
        const size_t wb = (size_t)vs->getFrameWidth(src, p) * g->bps;
        const int h = vs->getFrameHeight(src, p);
        for (int y = 0; y < h; y++) memcpy(dp + (ptrdiff_t)y * ds, sp + (ptrdiff_t)y * ss, wb);
    }
    vs->mapSetInt(vs->getFramePropertiesRW(dst), PROP_SUPER_LAZY, 1, maReplace);
}
/* a consumer found no device copy of a lazy frame: rebuild it from the embedded source with the handle of the mv.Super instance that made it */
static int super_rebuild_lazy(DevRef *r, const VSFrame *f, int64_t id, uint64_t print, const SuperGeo *want, const VSAPI *vs) {
    SuperData *sd = super_by_instance(id >> 32);
    if (!sd || sd->pelMode) return MVX_E_ARG; /* (its mv.Super instance is gone: cannot happen while a consumer holds the node) */
    const SuperGeo *g = &sd->geo;
    /* the id comes from frame properties: a stale or foreign one may name a live instance with ANOTHER layout -- the planes built below must be the
     * ones the consumer's kernels will address */
    if (g->bytes != want->bytes || g->copies != want->copies || g->bps != want->bps || g->si.num_planes != want->si.num_planes) return MVX_E_ARG;
    for (int p = 0; p < g->si.num_planes; p++)
        if (g->pitch[p] != want->pitch[p] || g->off[p] != want->off[p] || g->shadowStride[p] != want->shadowStride[p] ||
            g->si.plane_width[p] != want->si.plane_width[p] || g->si.plane_height[p] != want->si.plane_height[p]) return MVX_E_ARG;
    size_t off[3], total = 0;
    int w[3], h[3];
    for (int p = 0; p < g->si.num_planes; p++) {
        w[p] = p ? g->si.width / g->si.xRatioUV : g->si.width; h[p] = p ? g->si.height / g->si.yRatioUV : g->si.height;
        off[p] = total; total += (size_t)sd->srcPitch[p] * (size_t)h[p];
    }
    void *srcArena = shell_alloc(total), *arena = shell_alloc(g->bytes), *dsrc[3] = { NULL, NULL, NULL };
    int rc = (!srcArena || !arena) ? MVX_E_NOMEM : 0;
    void *st = thread_stream();
    if (!rc) rc = mvx_dev_memset(arena, 0, g->bytes, st);
    for (int p = 0; p < g->si.num_planes && !rc; p++) {
        dsrc[p] = (char *)srcArena + off[p];
        rc = timed_upload(dsrc[p], sd->srcPitch[p], vs->getReadPtr(f, p), vs->getStride(f, p), (size_t)w[p] * g->bps, (size_t)h[p]);
        r->plane[p] = (char *)arena + g->off[p];
    }
    if (!rc) {
        if (g->copies > 1) rc = mvx_super_frames_shadow(sd->sup, 1, (const void *const *)dsrc, sd->srcPitch, (void *const *)r->plane, g->pitch, g->shadowStride, st);
        else rc = mvx_super_frames(sd->sup, 1, (const void *const *)dsrc, sd->srcPitch, (void *const *)r->plane, g->pitch, st);
    }
    if (!rc) rc = mvx_stream_sync(st); else (void)mvx_stream_sync(st);
    if (srcArena) mvx_dev_free(srcArena);
    if (rc) { shell_quiesce(rc); if (arena) mvx_dev_free(arena); memset(r, 0, sizeof(*r)); return rc; }
    if ((r->cached = cache_insert(id, print, arena, g, 0)) != NULL) return 0;
    r->temp = arena;
    return 0;
}
/* builds the device super frames of `n` SOURCE frames in one batch (uploads, then one mvx_super_frames_shadow call) and hands them
 * to the cache, pinned; frames that are cached already are only pinned.  out[i] = the pinned entry of frame nums[i].  One builder at a
 * time: six vector clips ask for the same frames.  Returns 0 or an MVX_E_* code (entries pinned so far are released on failure). */
static pthread_mutex_t g_build_mu = PTHREAD_MUTEX_INITIALIZER;
#define BUILD_THREADS 8
static long env_long_early(const char *name, long def) { const char *e = getenv(name); return e ? atol(e) : def; }
typedef struct BuildJob { SuperData *sd; const VSAPI *vs; const int *miss; int nmiss; const VSFrame *const *srcs; void **srcArena, **arena; const void **sp; void **dp; int next, rc; pthread_mutex_t mu; void *st; } BuildJob;
static void *build_worker(void *arg) { /* uploads source frames and prepares their (zero-filled) super arenas, one frame at a time */
    BuildJob *j = (BuildJob *)arg;
    const SuperGeo *g = &j->sd->geo;
    void *st = j->st; /* the BUILDER's stream: the arenas' memsets must be ordered before the super kernels it launches there (a helper thread's own stream of the pool is not) */
    for (;;) {
        pthread_mutex_lock(&j->mu);
        const int k = (j->rc || j->next >= j->nmiss) ? -1 : j->next++;
        pthread_mutex_unlock(&j->mu);
        if (k < 0) return NULL;
        const int i = j->miss[k];
        void *d3[3];
        double tq = prof_now();
        int rc = upload_plane_set(d3, &j->srcArena[k], j->srcs[i], j->sd->srcPitch, g->si.num_planes, g->bps, j->vs);
        prof_add(PF_LA_SRCUP, tq);
        for (int p = 0; p < 3; p++) j->sp[k * 3 + p] = d3[p];
        tq = prof_now();
        if (!rc && !(j->arena[k] = shell_alloc(g->bytes))) rc = MVX_E_NOMEM;
        prof_add(PF_LA_ARENA, tq);
        if (!rc) rc = mvx_dev_memset(j->arena[k], 0, g->bytes, st);
        for (int p = 0; p < g->si.num_planes && !rc; p++) j->dp[k * 3 + p] = (char *)j->arena[k] + g->off[p];
        if (rc) { pthread_mutex_lock(&j->mu); if (!j->rc) j->rc = rc; pthread_mutex_unlock(&j->mu); }
    }
}
static int super_build_device(SuperData *sd, int n, const int *nums, const VSFrame *const *srcs, DevFrame **out, const VSAPI *vs) {
    const SuperGeo *g = &sd->geo;
    int rc = 0, nmiss = 0;
    const double tl = prof_now();
    pthread_mutex_lock(&g_build_mu);
    prof_add(PF_BUILD_LOCK, tl);
    int *miss = (int *)malloc(sizeof(int) * (size_t)n);
    void **srcArena = (void **)calloc((size_t)n, sizeof(void *)), **arena = (void **)calloc((size_t)n, sizeof(void *));
    const void **sp = (const void **)calloc((size_t)n * 3, sizeof(void *));
    void **dp = (void **)calloc((size_t)n * 3, sizeof(void *));
    if (!miss || !srcArena || !arena || !sp || !dp) rc = MVX_E_NOMEM;
    for (int i = 0; i < n; i++) out[i] = NULL;
    for (int i = 0; i < n && !rc; i++) {
        out[i] = cache_find_ex((sd->instance << 32) | (uint32_t)nums[i], 0, g->bytes);
        if (!out[i]) miss[nmiss++] = i;
    }
    void *st = thread_stream();
    const double tp = prof_now();
    if (!rc && nmiss) { /* the uploads (a host memcpy into pinned staging + a DMA each) are spread over a few helper threads: a window of 64 4K16
                         * frames is 1.6 GB, which one thread moves in ~0.4 s -- as long as it takes to consume the window */
        BuildJob job = { sd, vs, miss, nmiss, srcs, srcArena, arena, sp, dp, 0, 0, PTHREAD_MUTEX_INITIALIZER, st };
        pthread_t th[BUILD_THREADS];
        int want = (int)env_long_early("MVX_VS_BUILD_THREADS", 4);
        if (want < 1) want = 1;
        if (want > BUILD_THREADS) want = BUILD_THREADS;
        int nth = nmiss >= 2 * want ? want : (nmiss > 1 ? 2 : 1), started = 0;
        if (nth > want) nth = want;
        for (int t = 1; t < nth; t++) if (pthread_create(&th[started], NULL, build_worker, &job) == 0) started++;
        build_worker(&job);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
        rc = job.rc;
    }
    if (!rc && nmiss) {
        if (g->copies > 1) rc = mvx_super_frames_shadow(sd->sup, nmiss, sp, sd->srcPitch, dp, g->pitch, g->shadowStride, st);
        else rc = mvx_super_frames(sd->sup, nmiss, sp, sd->srcPitch, dp, g->pitch, st);
    }
    if (!rc && nmiss) rc = mvx_stream_sync(st);
    else if (rc && nmiss) (void)mvx_stream_sync(st); /* (part of the work may be queued: the arenas below go back to a pool that knows nothing of streams) */
    prof_add(PF_SUPER, tp);
    for (int k = 0; k < nmiss; k++) {
        if (srcArena && srcArena[k]) mvx_dev_free(srcArena[k]);
        if (!rc) {
            const int i = miss[k];
            out[i] = cache_insert((sd->instance << 32) | (uint32_t)nums[i], 0, arena[k], g, 1); /* (print 0: superGetFrame fills it in when it hands the host frame out) */
            if (!out[i]) { rc = MVX_E_NOMEM; mvx_dev_free(arena[k]); }
        } else if (arena && arena[k]) mvx_dev_free(arena[k]);
    }
    if (rc) for (int i = 0; i < n; i++) if (out[i]) { cache_unpin(out[i]); out[i] = NULL; }
    free(miss); free(srcArena); free(arena); free(sp); free(dp);
    pthread_mutex_unlock(&g_build_mu);
    return rc;
}

static void super_frame_props(VSFrame *dst, int n, int64_t id, const SuperGeo *g, const VSAPI *vs) {
    VSMap *props = vs->getFramePropertiesRW(dst);
    if (n == 0) { /* src/MVSuper.c:111-120 */
        vs->mapSetInt(props, "Super_height", g->si.height, maReplace);
        vs->mapSetInt(props, "Super_hpad", g->si.hpad, maReplace);
        vs->mapSetInt(props, "Super_vpad", g->si.vpad, maReplace);
        vs->mapSetInt(props, "Super_pel", g->si.pel, maReplace);
        vs->mapSetInt(props, "Super_modeyuv", g->si.modeYUV, maReplace);
        vs->mapSetInt(props, "Super_levels", g->si.levels, maReplace);
    }
    vs->mapSetInt(props, PROP_SUPER_ID, id, maReplace); /* (harmless extra prop; frames without it are uploaded by the consumers) */
}

static const VSFrame *VS_CC superGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    SuperData *d = (SuperData *)inst;
    if (reason == arInitial) { /* src/MVSuper.c:47-52 */
        vs->requestFrameFilter(n, d->node, ctx);
        if (d->pelMode) vs->requestFrameFilter(n, d->pelclip, ctx);
        return NULL;
    }
    if (reason != arAllFramesReady) return NULL;
    const double tgf = prof_now();
    const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
    const SuperGeo *g = &d->geo;
    const int64_t id = (d->instance << 32) | (uint32_t)n;
    DevFrame *have = d->pelMode ? NULL : cache_find_ex(id, 0, g->bytes); /* built on the device already (a vector clip's look-ahead): only the download is left */
    if (have) {
        int rc2 = 0;
        VSFrame *dst2 = vs->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, src, core);
        if (super_lazy()) lazy_fill(dst2, src, g, vs); /* (the pixels stay on the device) */
        else for (int p = 0; p < g->si.num_planes && !rc2; p++)
            rc2 = timed_download_on(download_stream(), vs->getWritePtr(dst2, p), vs->getStride(dst2, p), have->plane[p], g->pitch[p], (size_t)g->si.plane_width[p] * g->bps, (size_t)g->si.plane_height[p]);
        vs->freeFrame(src);
        if (rc2) { cache_unpin(have); vs->freeFrame(dst2); vs->setFilterError(mvx_last_error(), ctx); return NULL; }
        super_frame_props(dst2, n, id, g, vs);
        const uint64_t pr = frame_print(dst2, g, vs);
        pthread_mutex_lock(&g_lock); have->print = pr; pthread_mutex_unlock(&g_lock);
        cache_unpin(have);
        prof_add(PF_GF_SUPER, tgf);
        return dst2;
    }
    void *srcArena = NULL, *pelArena = NULL, *dsrc[3], *dpel[3] = { NULL, NULL, NULL }, *ddst[3] = { NULL, NULL, NULL };
    int rc = upload_plane_set(dsrc, &srcArena, src, d->srcPitch, g->si.num_planes, g->bps, vs);
    const VSFrame *pf = d->pelMode ? vs->getFrameFilter(n, d->pelclip, ctx) : NULL; /* src/MVSuper.c:62-64 */
    if (pf && !rc) rc = upload_plane_set(dpel, &pelArena, pf, d->pelPitch, g->si.num_planes, g->bps, vs);
    void *st = thread_stream();
    double tp = prof_now();
    void *arena = rc ? NULL : shell_alloc(g->bytes);
    prof_add(PF_ALLOC, tp);
    tp = prof_now();
    if (!rc && !arena) rc = MVX_E_NOMEM;
    if (!rc) rc = mvx_dev_memset(arena, 0, g->bytes, st); /* zero-filled: only the defined rectangles are written (MVSuper.c:73 memsets too) */
    if (!rc) {
        for (int p = 0; p < g->si.num_planes; p++) ddst[p] = (char *)arena + g->off[p];
        if (!d->pelMode && g->copies > 1) /* the level-0 kernels write their share of the shadow planes themselves */
            rc = mvx_super_frames_shadow(d->sup, 1, (const void *const *)dsrc, d->srcPitch, (void *const *)ddst, g->pitch, g->shadowStride, st);
        else {
            rc = mvx_super_frames_pelclip(d->sup, 1, (const void *const *)dsrc, d->srcPitch, (const void *const *)dpel, d->pelPitch, d->pelMode, (void *const *)ddst, g->pitch, st);
            if (!rc) rc = super_shadows(g, ddst, st);
        }
    }
    if (!rc) rc = mvx_stream_sync(st);
    prof_add(PF_SUPER, tp);
    VSFrame *dst = NULL;
    if (!rc) {
        dst = vs->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, src, core);
        if (super_lazy() && !d->pelMode) lazy_fill(dst, src, g, vs);
        else {
            for (int p = 0; p < g->si.num_planes && !rc; p++)
                rc = timed_download_on(download_stream(), vs->getWritePtr(dst, p), vs->getStride(dst, p), ddst[p], g->pitch[p], (size_t)g->si.plane_width[p] * g->bps, (size_t)g->si.plane_height[p]);
        }
    }
    shell_quiesce(rc);
    if (srcArena) mvx_dev_free(srcArena);
    if (pelArena) mvx_dev_free(pelArena);
    if (pf) vs->freeFrame(pf);
    vs->freeFrame(src);
    if (rc) {
        if (arena) mvx_dev_free(arena);
        if (dst) vs->freeFrame(dst);
        vs->setFilterError(mvx_last_error(), ctx);
        return NULL;
    }
    super_frame_props(dst, n, id, g, vs);
    /* the device copy stays resident for the consumers */
    DevFrame *e = cache_insert(id, frame_print(dst, g, vs), arena, g, 1);
    if (e) cache_unpin(e); else mvx_dev_free(arena);
    prof_add(PF_GF_SUPER, tgf);
    return dst;
}

static void VS_CC superFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    SuperData *d = (SuperData *)inst;
    vs->freeNode(d->node);
    if (d->pelclip) vs->freeNode(d->pelclip);
    super_unregister(d);
    cache_evict_instance(d->instance); /* its device-resident frames are of no use to anybody now */
    mvx_super_destroy(d->sup);
    free(d);
}

/* ------------------------------------------------------------------------------------------------ start-up on a background thread (r6)
 * The HIP runtime (0.25-0.3 s), this library's code objects (~0.1 s at the first launch) and the page-locked staging buffers (~20 ms each) come up on a thread started when the
 * plugin is loaded, while the host is still reading its script / opening its clips.  Two rules keep that safe: (1) every filter-creation function first waits until the RUNTIME
 * is up (what a synchronous start would have cost it anyway) -- so a host that exits after a creation error never tears the process down under a thread that is inside the
 * runtime's initialisation (the runtime's own exit handlers run before any handler registered earlier: measured, a segmentation fault at exit with the error message still in the
 * stdio buffer); (2) once the runtime is up the thread registers an exit handler -- which therefore runs BEFORE the runtime's -- that waits for the rest of the warm-up.
 * MVX_VS_WARMUP=0 turns the thread off, MVX_VS_WARM_STAGES = staging buffers to page-lock (default 48 = 768 MiB: what a 4K session allocates within its first second). */
static pthread_t g_warm_thread;
static int g_warm_started, g_warm_runtime_up, g_warm_done; /* (atomics: __atomic_*) */
static void warm_join(void) {
    if (__atomic_exchange_n(&g_warm_started, 0, __ATOMIC_ACQ_REL)) pthread_join(g_warm_thread, NULL);
}
static void *warmup_thread(void *arg) {
    (void)arg;
    (void)mvx_warmup(-1);         /* the runtime (no device: the filters report that when they are used) */
    __atomic_store_n(&g_warm_runtime_up, 1, __ATOMIC_RELEASE);
    atexit(warm_join);            /* (registered after the runtime's own handlers, so it runs before them) */
    const char *e = getenv("MVX_VS_WARM_STAGES");
    (void)mvx_warmup(e ? atoi(e) : 48); /* code objects, staging buffers */
    __atomic_store_n(&g_warm_done, 1, __ATOMIC_RELEASE);
    return NULL;
}
static void warm_barrier(void) { /* at the top of every *Create: the runtime is up (or no warm-up runs) */
    while (__atomic_load_n(&g_warm_started, __ATOMIC_ACQUIRE) && !__atomic_load_n(&g_warm_runtime_up, __ATOMIC_ACQUIRE)) { struct timespec t = { 0, 2000000 }; nanosleep(&t, NULL); }
}
__attribute__((destructor)) static void warm_unload(void) { warm_join(); } /* (dlclose without exit) */

static void VS_CC superCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    (void)user;
    VSNode *node = vs->mapGetNode(in, "clip", 0, 0);
    const VSVideoInfo *vi = vs->getVideoInfo(node);
    if (!mvx_vsh_is_constant_video_format(vi) || vi->format.bitsPerSample > 16 || vi->format.sampleType != stInteger || vi->format.subSamplingW > 1 ||
        vi->format.subSamplingH > 1 || (vi->format.colorFamily != cfYUV && vi->format.colorFamily != cfGray)) {
        /* argument-value errors come first in the reference (MVSuper.c:177-192): let the library report them on a dummy format */
        mvx_super_args t = { 64, 64, 8, 1, 1, 0, opt_int(in, "hpad", vs), opt_int(in, "vpad", vs), opt_int(in, "pel", vs), opt_int(in, "levels", vs), opt_int(in, "chroma", vs), opt_int(in, "sharp", vs), opt_int(in, "rfilter", vs) };
        mvx_super *ts = NULL; char terr[MVX_ERRLEN];
        if (mvx_super_create(&t, &ts, terr)) vs->mapSetError(out, terr);
        else { mvx_super_destroy(ts); vs->mapSetError(out, "Super: input clip must be GRAY, 420, 422, 440, or 444, up to 16 bits, with constant dimensions."); }
        vs->freeNode(node);
        return;
    }
    mvx_super_args a = { vi->width, vi->height, vi->format.bitsPerSample, vi->format.subSamplingW, vi->format.subSamplingH, vi->format.colorFamily == cfGray,
                         opt_int(in, "hpad", vs), opt_int(in, "vpad", vs), opt_int(in, "pel", vs), opt_int(in, "levels", vs), opt_int(in, "chroma", vs),
                         opt_int(in, "sharp", vs), opt_int(in, "rfilter", vs) };
    if (a.chroma != MVX_UNSET) a.chroma = !!a.chroma;
    mvx_super *sup = NULL;
    char err[MVX_ERRLEN];
    if (mvx_super_create(&a, &sup, err)) { vs->mapSetError(out, err); vs->freeNode(node); return; }
    /* src/MVSuper.c:229-256 */
    int perr = 0;
    int32_t pelMode = 0;
    VSNode *pelclip = vs->mapGetNode(in, "pelclip", 0, &perr);
    if (perr) pelclip = NULL;
    if (pelclip) {
        const VSVideoInfo *pvi = vs->getVideoInfo(pelclip);
        const int same = pvi->format.colorFamily == vi->format.colorFamily && pvi->format.sampleType == vi->format.sampleType &&
                         pvi->format.bitsPerSample == vi->format.bitsPerSample && pvi->format.subSamplingW == vi->format.subSamplingW &&
                         pvi->format.subSamplingH == vi->format.subSamplingH;
        err[0] = 0;
        if (!mvx_vsh_is_constant_video_format(pvi) || !same) snprintf(err, sizeof(err), "Super: pelclip must have the same format as the input clip, and it must have constant dimensions.");
        else mvx_super_pelclip_mode(sup, pvi->width, pvi->height, &pelMode, err);
        if (err[0]) { vs->mapSetError(out, err); mvx_super_destroy(sup); vs->freeNode(node); vs->freeNode(pelclip); return; }
    }
    SuperData *d = (SuperData *)calloc(1, sizeof(*d));
    d->node = node; d->sup = sup; d->vi = *vi;
    d->pelclip = pelclip; d->pelMode = pelMode;
    super_geo(&d->geo, sup);
    d->vi.width = d->geo.si.super_width; d->vi.height = d->geo.si.super_height;
    const int bps = d->geo.bps;
    for (int p = 0; p < 3; p++) {
        const int w = p ? vi->width >> vi->format.subSamplingW : vi->width;
        d->srcPitch[p] = ((ptrdiff_t)w * bps + 255) / 256 * 256;
        if (pelMode) {
            const VSVideoInfo *pvi = vs->getVideoInfo(pelclip);
            const int pw = p ? pvi->width >> vi->format.subSamplingW : pvi->width;
            d->pelPitch[p] = ((ptrdiff_t)pw * bps + 255) / 256 * 256;
        }
    }
    pthread_mutex_lock(&g_lock);
    d->instance = g_next_instance++;
    pthread_mutex_unlock(&g_lock);
    VSFilterDependency deps[2] = { { node, rpStrictSpatial }, { pelclip, rpStrictSpatial } };
    vs->createVideoFilter(out, "Super", &d->vi, superGetFrame, superFree, fmParallel, deps, pelMode ? 2 : 1, d, core);
    { /* remember the node the core made for this instance (identity only: no reference is kept) */
        int e = 0;
        VSNode *o = vs->mapGetNode(out, "clip", 0, &e);
        if (o && !e) { super_register(o, d); vs->freeNode(o); }
    }
}

/* ------------------------------------------------------------------------------------------------ mv.Analyse */

/* Combining queue: the getFrame calls of concurrent worker threads become ONE mvx_analyse_frames launch.  The first thread to
 * arrive leads: it waits for others -- until `maxBatch` requests are queued, or no new request has arrived for a quiet period, or a
 * total budget is used up -- takes the whole queue, launches it on the instance's stream, waits for it and wakes the others up.
 * Requests arriving meanwhile form the next batch.  Quiet period and budget follow the duration of the previous launch (1/8 and
 * 1/2 of it, at least MVX_VS_BATCH_WAIT_US, the quiet period at most MVX_VS_BATCH_QUIET_MAX_US): a 4K search takes ~0.3 s whether it
 * carries one chain or five hundred, so the frames a host delivers a few milliseconds apart are worth waiting for. */
typedef struct AnReq { mvx_analyse_job job; int rc, done; struct AnReq *next; } AnReq;
typedef struct Combiner {
    pthread_mutex_t mu; pthread_cond_t done, more;
    AnReq *head, *tail; int n, leader;
    int maxBatch; long waitUs, quietMaxUs; /* a few streams: the next batch may start while the previous one runs */
    long lastUs; /* duration of the previous launch: a search of few chains takes as long as one of hundreds, so waiting a fraction of it for more requests is cheap */
    long batches, jobs, largest; /* statistics (MVX_VS_STATS=1 prints them when the filter is freed) */
} Combiner;

/* Look-ahead (mv.Analyse whose `super` argument is this plugin's own mv.Super node).  A search is a serial chain per frame that
 * takes ~0.45 s at 4K however few chains a launch carries, so a vector clip is computed a WINDOW of B consecutive
 * frames at a time: the first request that touches window w also asks for the SOURCE frames (25 MB each at 4K16, not the 131 MB
 * super frames) of w and of the next two windows, builds their super frames on the device only, and launches one search per window on
 * the instance's low-priority streams; the windows ahead are gathered, built and searched while the frames of w are being consumed.  Every other request of the window just waits
 * for the launch that is already running (GPU work only: nothing it waits for needs a host worker thread) and copies its blob out
 * of the window's host array.  The request protocol stays the reference's (MVAnalyse.c:84-113: everything a frame needs is asked
 * for at arInitial and fetched at arAllFramesReady); the extra requests go to the source clip, declared as a second dependency. */
#define LA_SLOTS 6
enum { LW_EMPTY, LW_BUILDING, LW_LAUNCHED, LW_SYNCING, LW_READY, LW_FAILED };
typedef struct LaWindow { int w, state, first, count, users, rc; void *dblobs; size_t dstride; DevFrame **pins; int npins; void *stream; } LaWindow; /* dblobs: the window's vectors, on the device until the slot is recycled */
typedef struct LookAhead { int on, B, depth; volatile int started; /* a window of this instance has been launched (r6: the request that starts the FIRST one starts no window ahead) */ volatile int degraded; /* a window ran out of device memory: new requests take the per-frame path */ VSNode *srcNode; SuperData *sd; pthread_mutex_t mu; pthread_cond_t cv; LaWindow win[LA_SLOTS]; } LookAhead;
#define LA_DEPTH_MAX 3
typedef struct LaReq { int legacy, w, hold[1 + LA_DEPTH_MAX], want[1 + LA_DEPTH_MAX]; } LaReq;

typedef struct AnalyseData { VSNode *node; const VSVideoInfo *vi; mvx_super *sup; SuperGeo geo; mvx_analyse *an; mvx_analysis_data ad; int blobSize; FieldOpt fo; Combiner cb; LookAhead la; } AnalyseData;

static long env_long(const char *name, long def) { const char *e = getenv(name); return e ? atol(e) : def; }

/* totals over all instances, printed when the plugin is unloaded (MVX_VS_STATS=1): a host need not free its nodes before it exits */
static long g_stat_launches, g_stat_jobs, g_stat_largest, g_stat_instances;
__attribute__((destructor)) static void print_stats(void) {
    if (env_long("MVX_VS_STATS", 0) && g_stat_instances)
    {
        fprintf(stderr, "mvtools_vs: Analyse instances=%ld launches=%ld jobs=%ld largest_batch=%ld\n", g_stat_instances, g_stat_launches, g_stat_jobs, g_stat_largest);
        static const char *nm[PF_N] = { "stream_create", "upload", "download", "super_kernels", "search_wait", "degrain", "dev_alloc", "lookahead_blocked", "build_lock",
                                        "getframe_super", "getframe_analyse", "getframe_degrain", "window_build", "window_arena_alloc", "window_launch_call", "window_source_upload" };
        fprintf(stderr, "mvtools_vs: thread-seconds");
        for (int k = 0; k < PF_N; k++) fprintf(stderr, " %s=%.2f/%ld", nm[k], g_prof[k], g_prof_n[k]);
        fprintf(stderr, "\n");
    }
}

/* Streams of the search launches, shared by every mv.Analyse instance and handed out round robin.  The runtime maps streams onto a few
 * hardware queues per priority level, in creation order, and launches that share a queue run one after the other: with four streams per
 * instance the six instances of a Degrain3 graph put the six searches of a window on the SAME queue (traced: 128-chain launches enqueued
 * together finished 0.6 s apart, two at a time).  One pool makes consecutive launches land on different queues; the HOST should run with
 * GPU_MAX_HW_QUEUES >= SEARCH_STREAMS (the mini host sets 16 unless the user chose a value).  Priority: MVX_VS_SEARCH_PRIO (-1 lowest = default, 0, 1);
 * a NULL entry (the default stream) still works, it only serialises. */
#define SEARCH_STREAMS 12
static void *g_search_stream[SEARCH_STREAMS];
static unsigned g_search_next;
static int g_search_made;
static void *search_stream_next(void) {
    pthread_mutex_lock(&g_lock);
    if (!g_search_made) {
        g_search_made = 1;
        for (int i = 0; i < SEARCH_STREAMS; i++) g_search_stream[i] = mvx_stream_create_priority((int)env_long("MVX_VS_SEARCH_PRIO", -1));
    }
    void *st = g_search_stream[g_search_next++ % SEARCH_STREAMS];
    pthread_mutex_unlock(&g_lock);
    return st;
}
static void combiner_init(Combiner *c) {
    memset(c, 0, sizeof(*c));
    pthread_mutex_init(&c->mu, NULL); pthread_cond_init(&c->done, NULL); pthread_cond_init(&c->more, NULL);
    /* priority of the search streams: MVX_VS_SEARCH_PRIO (-1 lowest = default, 0, 1); NULL (the default stream) still works, it only serialises the launches */
    pthread_mutex_lock(&g_lock); g_stat_instances++; pthread_mutex_unlock(&g_lock);
    c->maxBatch = (int)env_long("MVX_VS_BATCH_MAX", 1024);
    c->waitUs = env_long("MVX_VS_BATCH_WAIT_US", 2000);
    c->quietMaxUs = env_long("MVX_VS_BATCH_QUIET_MAX_US", 30000);
    if (c->maxBatch < 1) c->maxBatch = 1;
}
static void combiner_free(Combiner *c, const char *what) {
    (void)what;
    pthread_mutex_destroy(&c->mu); pthread_cond_destroy(&c->done); pthread_cond_destroy(&c->more);
}
/* blocks until the request's blob is computed; returns its MVX_* code */
static int combiner_submit(Combiner *c, mvx_analyse *an, AnReq *r) {
    pthread_mutex_lock(&c->mu);
    r->next = NULL; r->done = 0;
    if (c->tail) c->tail->next = r; else c->head = r;
    c->tail = r; c->n++;
    if (c->leader) { /* follower */
        pthread_cond_signal(&c->more); /* the leader restarts its quiet period */
        while (!r->done) pthread_cond_wait(&c->done, &c->mu);
        pthread_mutex_unlock(&c->mu);
        return r->rc;
    }
    c->leader = 1;
    if (c->waitUs > 0 && c->n < c->maxBatch) {
        long quiet = c->lastUs / 8, budget = c->lastUs / 2;
        if (quiet > c->quietMaxUs) quiet = c->quietMaxUs;
        if (quiet < c->waitUs) quiet = c->waitUs;
        if (budget < c->waitUs) budget = c->waitUs;
        struct timespec t0, now, ts;
        clock_gettime(CLOCK_REALTIME, &t0);
        while (c->n < c->maxBatch) {
            clock_gettime(CLOCK_REALTIME, &now);
            const long used = (now.tv_sec - t0.tv_sec) * 1000000L + (now.tv_nsec - t0.tv_nsec) / 1000;
            if (used >= budget) break;
            const long w = quiet < budget - used ? quiet : budget - used;
            ts = now;
            ts.tv_nsec += (w % 1000000) * 1000; ts.tv_sec += w / 1000000 + ts.tv_nsec / 1000000000; ts.tv_nsec %= 1000000000;
            const int n0 = c->n;
            if (pthread_cond_timedwait(&c->more, &c->mu, &ts) == ETIMEDOUT && c->n == n0) break; /* quiet: nobody else is about to arrive */
        }
    }
    AnReq *list = c->head;
    const int n = c->n;
    c->head = c->tail = NULL; c->n = 0; c->leader = 0; /* the next arrival leads the next batch while this one runs */
    void *stream = search_stream_next();
    c->batches++; c->jobs += n; if (n > c->largest) c->largest = n;
    pthread_mutex_unlock(&c->mu);
    pthread_mutex_lock(&g_lock);
    g_stat_launches++; g_stat_jobs += n; if (n > g_stat_largest) g_stat_largest = n;
    pthread_mutex_unlock(&g_lock);
    int rc = 0;
    mvx_analyse_job *jobs = (mvx_analyse_job *)malloc(sizeof(mvx_analyse_job) * (size_t)n);
    if (!jobs) rc = MVX_E_NOMEM;
    else {
        int i = 0;
        for (AnReq *q = list; q; q = q->next) jobs[i++] = q->job;
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        rc = mvx_analyse_frames(an, n, jobs, stream);
        if (!rc) rc = mvx_stream_sync(stream);
        clock_gettime(CLOCK_MONOTONIC, &b);
        free(jobs);
        pthread_mutex_lock(&c->mu);
        c->lastUs = (b.tv_sec - a.tv_sec) * 1000000L + (b.tv_nsec - a.tv_nsec) / 1000;
        pthread_mutex_unlock(&c->mu);
    }
    pthread_mutex_lock(&c->mu);
    for (AnReq *q = list; q;) { AnReq *nx = q->next; q->rc = rc; q->done = 1; q = nx; } /* (q may be freed by its owner once done) */
    pthread_cond_broadcast(&c->done);
    pthread_mutex_unlock(&c->mu);
    return rc;
}

static int analyse_nref(const AnalyseData *d, int n) { /* src/MVAnalyse.c:84-104 */
    if (d->ad.nDeltaFrame > 0) return n + (d->ad.isBackward ? d->ad.nDeltaFrame : -d->ad.nDeltaFrame);
    return -d->ad.nDeltaFrame;
}

/* ---- look-ahead helpers (d->la.mu held where noted) */
static void la_window_release(LaWindow *s) { /* frees what a finished / failed window holds; the slot becomes EMPTY */
    for (int i = 0; i < s->npins; i++) cache_unpin(s->pins[i]);
    free(s->pins); s->pins = NULL; s->npins = 0;
    if (s->dblobs) { mvx_dev_free(s->dblobs); s->dblobs = NULL; }
    s->state = LW_EMPTY; s->count = 0; s->rc = 0;
}
/* the slot of window w, recycled from an older finished window if nobody uses it; NULL if it is taken.  mu held. */
static LaWindow *la_slot(AnalyseData *d, int w) {
    LaWindow *s = &d->la.win[w % LA_SLOTS];
    if (s->w == w) return s;
    if (s->users || s->state == LW_BUILDING || s->state == LW_LAUNCHED || s->state == LW_SYNCING) return NULL;
    la_window_release(s);
    s->w = w;
    return s;
}
/* source frames window v needs: [*lo, *hi] and, for a static reference (delta <= 0), the frame *fixed (else -1) */
static void la_inputs(const AnalyseData *d, int v, int *lo, int *hi, int *fixed) {
    const int first = v * d->la.B, last = (first + d->la.B < d->vi->numFrames ? first + d->la.B : d->vi->numFrames) - 1;
    *lo = first; *hi = last; *fixed = -1;
    if (d->ad.nDeltaFrame > 0) {
        if (d->ad.isBackward) { const int h = last + d->ad.nDeltaFrame; *hi = h < d->vi->numFrames ? h : d->vi->numFrames - 1; }
        else { const int l = first - d->ad.nDeltaFrame; *lo = l > 0 ? l : 0; }
    } else if (-d->ad.nDeltaFrame < d->vi->numFrames) *fixed = -d->ad.nDeltaFrame;
}
static void la_request_inputs(const AnalyseData *d, int v, VSFrameContext *ctx, const VSAPI *vs) {
    int lo, hi, fixed;
    la_inputs(d, v, &lo, &hi, &fixed);
    for (int k = lo; k <= hi; k++) vs->requestFrameFilter(k, d->la.srcNode, ctx);
    if (fixed >= 0 && (fixed < lo || fixed > hi)) vs->requestFrameFilter(fixed, d->la.srcNode, ctx);
}
static int analyse_nref(const AnalyseData *d, int n);
/* builds the super frames of window v on the device and enqueues its search (does not wait for it) */
static int la_launch(AnalyseData *d, LaWindow *s, int v, VSFrameContext *ctx, const VSAPI *vs) {
    int lo, hi, fixed;
    la_inputs(d, v, &lo, &hi, &fixed);
    const double tl0 = prof_now();
    const int first = v * d->la.B, count = (first + d->la.B < d->vi->numFrames ? first + d->la.B : d->vi->numFrames) - first;
    const int extra = fixed >= 0 && (fixed < lo || fixed > hi);
    const int nn = hi - lo + 1 + extra;
    int *nums = (int *)malloc(sizeof(int) * (size_t)nn), *top = (int *)calloc((size_t)nn, sizeof(int));
    const VSFrame **srcs = (const VSFrame **)calloc((size_t)nn, sizeof(VSFrame *));
    DevFrame **pins = (DevFrame **)calloc((size_t)nn, sizeof(DevFrame *));
    mvx_analyse_job *jobs = (mvx_analyse_job *)calloc((size_t)count, sizeof(mvx_analyse_job));
    int rc = (!nums || !top || !srcs || !pins || !jobs) ? MVX_E_NOMEM : 0, missing = 0;
    for (int i = 0; i < nn && !rc; i++) {
        nums[i] = i < hi - lo + 1 ? lo + i : fixed;
        srcs[i] = vs->getFrameFilter(nums[i], d->la.srcNode, ctx);
        if (!srcs[i]) { rc = MVX_E_ARG; break; }
        top[i] = frame_top_field(&d->fo, srcs[i], nums[i], &missing, vs); /* mv.Super copies the props of its source frame (MVSuper.c:104) */
    }
    if (!rc && missing && d->fo.fields) rc = -1000; /* reported by the caller with the reference's message */
    la_trace(d, v, "source frames fetched", prof_now() - tl0);
    { const double tq = prof_now(); if (!rc) rc = super_build_device(d->la.sd, nn, nums, srcs, pins, vs); prof_add(PF_LA_BUILD, tq); la_trace(d, v, "super frames on the device", prof_now() - tq); }
    for (int i = 0; i < nn; i++) if (srcs && srcs[i]) vs->freeFrame(srcs[i]);
    const size_t stride = ((size_t)d->blobSize + 255) / 256 * 256;
    void *dblobs = rc ? NULL : shell_alloc(stride * (size_t)count);
    /* the window's vectors stay on the device; every request downloads its own 2.7 MB when it is served.  (Bringing a whole window to the
     * host at once was tried both ways: a synchronous 350 MB download by the first waiter stalled all request threads six times per window,
     * and page-locked per-slot arrays for an asynchronous copy cost 0.4 s of hipHostMalloc each, during which every HIP call of the process
     * waits.) */
    if (!rc && !dblobs) rc = MVX_E_NOMEM;
    if (!rc) {
        for (int i = 0; i < count; i++) {
            const int k = first + i, nref = analyse_nref(d, k);
            const int haveRef = nref >= 0 && nref < d->vi->numFrames;
            const int is = k - lo, ir = !haveRef ? -1 : (nref >= lo && nref <= hi) ? nref - lo : nn - 1;
            for (int p = 0; p < 3; p++) { jobs[i].src[p] = pins[is]->plane[p]; jobs[i].ref[p] = haveRef ? pins[ir]->plane[p] : NULL; }
            jobs[i].blob = (char *)dblobs + stride * (size_t)i;
            jobs[i].field_shift = (haveRef && d->fo.fields && d->ad.nPel > 1 && (d->ad.nDeltaFrame % 2)) ? field_shift_of(top[is], top[ir], d->ad.nPel) : 0;
        }
        s->stream = search_stream_next();
        const double tq = prof_now();
        rc = mvx_analyse_frames(d->an, count, jobs, s->stream);
        prof_add(PF_LA_LAUNCH, tq);
        la_trace(d, v, "search enqueued", prof_now() - tl0);
        pthread_mutex_lock(&g_lock);
        g_stat_launches++; g_stat_jobs += count; if (count > g_stat_largest) g_stat_largest = count;
        pthread_mutex_unlock(&g_lock);
    }
    if (rc) {
        shell_quiesce(rc);
        if (s->stream) (void)mvx_stream_sync(s->stream);
        for (int i = 0; i < nn; i++) if (pins && pins[i]) cache_unpin(pins[i]);
        free(pins);
        if (dblobs) mvx_dev_free(dblobs);
    } else { s->pins = pins; s->npins = nn; s->dblobs = dblobs; s->dstride = stride; s->first = first; s->count = count; }
    free(nums); free(top); free(srcs); free(jobs);
    return rc;
}
/* waits until window s is READY (or FAILED): the first waiter synchronises the window's stream */
static int la_wait(AnalyseData *d, LaWindow *s) {
    pthread_mutex_lock(&d->la.mu);
    for (;;) {
        if (s->state == LW_READY || s->state == LW_FAILED || s->state == LW_EMPTY) break;
        if (s->state == LW_LAUNCHED) {
            s->state = LW_SYNCING;
            pthread_mutex_unlock(&d->la.mu);
            const double t0 = prof_now();
            const int rc = mvx_stream_sync(s->stream);
            prof_add(PF_SEARCH_WAIT, t0);
            la_trace(d, s->w, "first waiter saw the search finish", prof_now() - t0);
            pthread_mutex_lock(&d->la.mu);
            for (int i = 0; i < s->npins; i++) cache_unpin(s->pins[i]);
            free(s->pins); s->pins = NULL; s->npins = 0;
            s->rc = rc; s->state = rc ? LW_FAILED : LW_READY;
            pthread_cond_broadcast(&d->la.cv);
            continue;
        }
        { const double tb = prof_now(); pthread_cond_wait(&d->la.cv, &d->la.mu); prof_add_locked_ok(PF_LA_BLOCKED, tb); } /* BUILDING / SYNCING: somebody is on it */
    }
    const int rc = s->state == LW_READY ? 0 : (s->rc ? s->rc : MVX_E_DEVICE);
    pthread_mutex_unlock(&d->la.mu);
    return rc;
}
static void la_release_req(AnalyseData *d, LaReq *r) {
    if (!r) return;
    if (!r->legacy) {
        pthread_mutex_lock(&d->la.mu);
        for (int i = 0; i <= LA_DEPTH_MAX; i++) if (r->hold[i]) d->la.win[(r->w + i) % LA_SLOTS].users--;
        pthread_mutex_unlock(&d->la.mu);
    }
    free(r);
}

/* r6: the vectors of a look-ahead window stay on the device until its slot is recycled; a consumer of this plugin (mv.DegrainN) that finds the window of frame n still there
 * reads them where they are instead of uploading the 2.7 MB property the vector frame carries (six of them per 4K Degrain3 frame).  The vector FRAME is unchanged -- any
 * other consumer reads its properties as before -- and when the window is gone the consumer falls back on them too.  The registry maps the node the core made for an
 * mv.Analyse instance to the instance (identity only, like g_supers; a consumer holds a reference to the node for as long as it may ask). */
#define ANALYSE_REG_MAX 256
static struct { VSNode *out; AnalyseData *d; } g_analyses[ANALYSE_REG_MAX];
static void analyse_register(VSNode *out, AnalyseData *d) {
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < ANALYSE_REG_MAX; i++) if (!g_analyses[i].out) { g_analyses[i].out = out; g_analyses[i].d = d; break; }
    pthread_mutex_unlock(&g_lock);
}
static void analyse_unregister(AnalyseData *d) {
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < ANALYSE_REG_MAX; i++) if (g_analyses[i].d == d) { g_analyses[i].out = NULL; g_analyses[i].d = NULL; }
    pthread_mutex_unlock(&g_lock);
}
static AnalyseData *analyse_lookup(VSNode *out) {
    AnalyseData *d = NULL;
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < ANALYSE_REG_MAX; i++) if (g_analyses[i].out == out) d = g_analyses[i].d;
    pthread_mutex_unlock(&g_lock);
    return d;
}
/* the device copy of frame n's vectors, its window pinned (la_device_blob_release), or NULL */
static const void *la_device_blob(AnalyseData *d, int n, LaWindow **held) {
    *held = NULL;
    if (!d || !d->la.on) return NULL;
    const void *p = NULL;
    pthread_mutex_lock(&d->la.mu);
    LaWindow *s = &d->la.win[(n / d->la.B) % LA_SLOTS];
    if (s->w == n / d->la.B && s->state == LW_READY && s->dblobs && n >= s->first && n < s->first + s->count) {
        s->users++; *held = s;
        p = (const char *)s->dblobs + s->dstride * (size_t)(n - s->first);
    }
    pthread_mutex_unlock(&d->la.mu);
    return p;
}
static void la_device_blob_release(AnalyseData *d, LaWindow *held) {
    if (!held) return;
    pthread_mutex_lock(&d->la.mu);
    held->users--;
    pthread_mutex_unlock(&d->la.mu);
}

static const VSFrame *VS_CC analyseGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    AnalyseData *d = (AnalyseData *)inst;
    const int nref = analyse_nref(d, n);
    const int haveRef = nref >= 0 && nref < d->vi->numFrames;
    if (d->la.on) {
        if (reason == arInitial) {
            LaReq *r = (LaReq *)calloc(1, sizeof(LaReq));
            *fd = r;
            if (r) {
                r->w = n / d->la.B;
                pthread_mutex_lock(&d->la.mu);
                LaWindow *s0 = d->la.degraded ? NULL : la_slot(d, r->w);
                if (!s0) r->legacy = 1;
                else {
                    s0->users++; r->hold[0] = 1; r->want[0] = s0->state == LW_EMPTY;
                    /* (r6: not the request that starts an instance's FIRST window -- mv.DegrainN's creation reads frame 0 of every vector clip, and the windows ahead of
                     * the first instances' frame 0 kept the GPU from the later instances' first windows: 0.8 s of graph construction for six clips; the windows ahead are
                     * started by the next request, a moment later) */
                    for (int i = 1; d->la.started && i <= d->la.depth && (r->w + i) * d->la.B < d->vi->numFrames; i++) { /* the windows ahead: whoever sees them empty first starts them */
                        LaWindow *s1 = la_slot(d, r->w + i);
                        if (s1 && s1->state == LW_EMPTY) { s1->users++; r->hold[i] = 1; r->want[i] = 1; }
                    }
                }
                pthread_mutex_unlock(&d->la.mu);
                if (!r->legacy) {
                    vs->requestFrameFilter(n, d->node, ctx); /* the vector clip's frame is a copy of the super frame (MVAnalyse.c:224) */
                    /* one source frame of the window that will be started next: by the time some request sees that window empty and asks
                     * for all of its frames, the requests of this window have had the host produce them, in parallel */
                    { const int ahead = n + (d->la.depth + 1) * d->la.B; if (ahead < d->vi->numFrames) vs->requestFrameFilter(ahead, d->la.srcNode, ctx); }
                    for (int i = 0; i <= LA_DEPTH_MAX; i++) if (r->want[i]) { la_trace(d, r->w + i, "source frames requested", 0.0); la_request_inputs(d, r->w + i, ctx, vs); }
                    return NULL;
                }
            }
            /* (no slot / no memory: this request takes the per-frame path below) */
        } else if (*fd && !((LaReq *)*fd)->legacy) {
            LaReq *r = (LaReq *)*fd;
            *fd = NULL;
            if (reason != arAllFramesReady) { la_release_req(d, r); return NULL; } /* arError */
            const double tgf = prof_now();
            int rc = 0;
            for (int i = 0; i <= LA_DEPTH_MAX && !rc; i++) {
                if (!r->want[i]) continue;
                LaWindow *s = &d->la.win[(r->w + i) % LA_SLOTS];
                int mine = 0;
                pthread_mutex_lock(&d->la.mu);
                if (s->state == LW_EMPTY) { s->state = LW_BUILDING; mine = 1; }
                pthread_mutex_unlock(&d->la.mu);
                if (!mine) continue;
                const int lrc = la_launch(d, s, r->w + i, ctx, vs);
                pthread_mutex_lock(&d->la.mu);
                s->rc = lrc; s->state = lrc ? LW_FAILED : LW_LAUNCHED;
                d->la.started = 1;
                if (lrc == MVX_E_NOMEM) d->la.degraded = 1; /* (the requests already waiting for this window fail -- their reference frames were never asked for -- but nothing after them does) */
                pthread_cond_broadcast(&d->la.cv);
                pthread_mutex_unlock(&d->la.mu);
                if (lrc && i == 0) rc = lrc;
            }
            LaWindow *s = &d->la.win[r->w % LA_SLOTS];
            if (!rc) rc = la_wait(d, s);
            const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
            VSFrame *dst = NULL;
            char *blob = rc ? NULL : (char *)malloc((size_t)d->blobSize);
            if (!rc && !blob) rc = MVX_E_NOMEM;
            /* (r4: the window's search was waited for; on the download stream the copy does not queue behind other threads' uploads -- PCIe is full duplex) */
            if (!rc) rc = timed_download_on(download_stream(), blob, d->blobSize, (const char *)s->dblobs + s->dstride * (size_t)(n - s->first), d->blobSize, (size_t)d->blobSize, 1);
            if (!rc && src) { /* src/MVAnalyse.c:224-239 */
                dst = vs->copyFrame(src, core);
                VSMap *props = vs->getFramePropertiesRW(dst);
                vs->mapSetData(props, PROP_ADATA, (const char *)&d->ad, sizeof(d->ad), dtBinary, maReplace);
                vs->mapSetData(props, PROP_VECTORS, blob, d->blobSize, dtBinary, maReplace);
            } else
                vs->setFilterError(rc == -1000 ? "Analyse: _Field property not found in input frame. Therefore, you must pass tff argument." :
                                   rc == MVX_E_NOMEM ? "Analyse: out of memory." : mvx_last_error(), ctx);
            free(blob);
            if (src) vs->freeFrame(src);
            la_release_req(d, r);
            prof_add(PF_GF_ANALYSE, tgf);
            return dst;
        } else if (*fd) { free(*fd); *fd = NULL; }
    }
    if (reason == arInitial) {
        if (haveRef && nref < n) vs->requestFrameFilter(nref, d->node, ctx);
        vs->requestFrameFilter(n, d->node, ctx);
        if (haveRef && nref >= n && nref != n) vs->requestFrameFilter(nref, d->node, ctx);
        return NULL;
    }
    if (reason != arAllFramesReady) return NULL;
    const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
    const VSFrame *ref = haveRef ? vs->getFrameFilter(nref, d->node, ctx) : NULL;
    int fieldShift = 0;
    { /* src/MVAnalyse.c:135-179 */
        int missing = 0;
        const int srcTop = frame_top_field(&d->fo, src, n, &missing, vs);
        const int refTop = ref ? frame_top_field(&d->fo, ref, nref, &missing, vs) : 0;
        if (missing && d->fo.fields) {
            vs->setFilterError("Analyse: _Field property not found in input frame. Therefore, you must pass tff argument.", ctx);
            if (ref) vs->freeFrame(ref);
            vs->freeFrame(src);
            return NULL;
        }
        if (ref && d->fo.fields && d->ad.nPel > 1 && (d->ad.nDeltaFrame % 2)) fieldShift = field_shift_of(srcTop, refTop, d->ad.nPel);
    }
    DevRef ds, dr;
    memset(&dr, 0, sizeof(dr));
    int rc = super_to_device(&ds, src, &d->geo, vs);
    if (!rc && ref) rc = super_to_device(&dr, ref, &d->geo, vs);
    void *dblob = rc ? NULL : shell_alloc((size_t)d->blobSize);
    char *blob = (char *)malloc((size_t)d->blobSize);
    if (!rc && (!dblob || !blob)) rc = MVX_E_NOMEM;
    if (!rc) {
        AnReq req;
        memset(&req, 0, sizeof(req));
        for (int p = 0; p < 3; p++) { req.job.src[p] = ds.plane[p]; req.job.ref[p] = ref ? dr.plane[p] : NULL; }
        req.job.blob = dblob;
        req.job.field_shift = fieldShift;
        const double t0 = prof_now();
        rc = combiner_submit(&d->cb, d->an, &req); /* one launch for all the frames requested right now */
        prof_add(PF_SEARCH_WAIT, t0);
        if (!rc) rc = timed_download(blob, d->blobSize, dblob, d->blobSize, (size_t)d->blobSize, 1);
        if (!rc) rc = mvx_stream_sync(thread_stream());
    }
    shell_quiesce(rc);
    dev_release(&ds); dev_release(&dr);
    if (dblob) mvx_dev_free(dblob);
    if (ref) vs->freeFrame(ref);
    VSFrame *dst = NULL;
    if (!rc) { /* src/MVAnalyse.c:224-239 */
        dst = vs->copyFrame(src, core);
        VSMap *props = vs->getFramePropertiesRW(dst);
        vs->mapSetData(props, PROP_ADATA, (const char *)&d->ad, sizeof(d->ad), dtBinary, maReplace);
        vs->mapSetData(props, PROP_VECTORS, blob, d->blobSize, dtBinary, maReplace);
    } else
        vs->setFilterError(rc == MVX_E_NOMEM ? "Analyse: out of memory." : mvx_last_error(), ctx);
    free(blob);
    vs->freeFrame(src);
    return dst;
}

static void VS_CC analyseFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    AnalyseData *d = (AnalyseData *)inst;
    analyse_unregister(d);
    if (d->la.on) { /* the windows' pins go first: freeing the super node may run mv.Super's free callback, which drops its unpinned frames from the cache */
        for (int i = 0; i < LA_SLOTS; i++) {
            LaWindow *s = &d->la.win[i];
            if (s->state == LW_LAUNCHED) (void)mvx_stream_sync(s->stream); /* nobody came for it */
            la_window_release(s);
        }
    }
    vs->freeNode(d->node);
    if (d->la.on) {
        vs->freeNode(d->la.srcNode);
        pthread_mutex_destroy(&d->la.mu); pthread_cond_destroy(&d->la.cv);
    }
    combiner_free(&d->cb, "Analyse");
    mvx_analyse_destroy(d->an);
    mvx_super_destroy(d->sup);
    free(d);
}

static void VS_CC analyseCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    (void)user;
    static const char *keys[] = { "blksize", "blksizev", "levels", "search", "searchparam", "pelsearch", "isb", "lambda", "chroma", "delta", "truemotion",
                                  "lsad", "plevel", "global", "pnew", "pzero", "pglobal", "overlap", "overlapv", "divide", "badsad", "badrange", "opt",
                                  "meander", "trymany", "fields", "tff", "search_coarse", "dct" };
    mvx_analyse_args a;
    int32_t *av = (int32_t *)&a;
    for (size_t i = 0; i < sizeof(keys) / sizeof(keys[0]); i++) av[i] = opt_int(in, keys[i], vs);
    VSNode *node = vs->mapGetNode(in, "super", 0, 0);
    const VSVideoInfo *vi = vs->getVideoInfo(node);
    char err[1200];
    if (!mvx_vsh_is_constant_video_format(vi) || vi->format.bitsPerSample > 16 || vi->format.sampleType != stInteger || vi->format.subSamplingW > 1 ||
        vi->format.subSamplingH > 1 || (vi->format.colorFamily != cfYUV && vi->format.colorFamily != cfGray)) {
        vs->mapSetError(out, "Analyse: super clip must be GRAY, 420, 422, 440, or 444, up to 16 bits, with constant dimensions.");
        vs->freeNode(node);
        return;
    }
    mvx_super *sup = super_from_props(node, "Analyse", err, sizeof(err), vs);
    if (!sup) { vs->mapSetError(out, err); vs->freeNode(node); return; }
    AnalyseData *d = (AnalyseData *)calloc(1, sizeof(*d));
    d->node = node; d->vi = vi; d->sup = sup;
    field_opt(&d->fo, in, vs);
    super_geo(&d->geo, sup);
    char lerr[MVX_ERRLEN];
    if (mvx_analyse_create(&a, sup, vi->numFrames, d->geo.pitch, &d->an, lerr)) {
        vs->mapSetError(out, lerr);
        mvx_super_destroy(sup); vs->freeNode(node); free(d);
        return;
    }
    mvx_analyse_get_data(d->an, &d->ad);
    d->blobSize = mvx_analyse_blob_size(d->an);
    if (d->geo.copies > 1) mvx_analyse_set_ref_shadow(d->an, d->geo.shadowStride); /* every device super frame of this shell carries its copies */
    combiner_init(&d->cb);
    { /* MVX_VS_SEARCH_LDS=<bytes>: LDS floor of a search workgroup (four chains): 54000 caps a CU at eight chains = two per SIMD, so that the
       * per-frame kernels of the other request threads (Super, Degrain) find registers beside the long-running low-priority search waves */
        const long f = env_long("MVX_VS_SEARCH_LDS", 0);
        if (f > 0) mvx_debug_option("fast_lds_min", (int)f);
    }
    /* before the first launch has been timed: a chain walks every block of every level, a few microseconds each */
    d->cb.lastUs = (long)((double)d->ad.nBlkX * d->ad.nBlkY * 4.0 / 3.0 * 2.5);
    VSFilterDependency deps[2] = { { node, rpGeneral }, { NULL, rpGeneral } };
    int ndeps = 1;
    { /* look-ahead when `super` is this plugin's own mv.Super node (MVX_VS_LOOKAHEAD = window length in frames, 0 = off; default 128) */
        SuperData *sd = super_lookup(node);
        const long B = env_long("MVX_VS_LOOKAHEAD", 128);
        if (sd && !sd->pelMode && B > 0 && vi->numFrames > 1 && cache_cap() > 0) { /* (no device cache, no windows: they live in it) */
            d->la.on = 1; d->la.B = (int)(B > 512 ? 512 : B); d->la.sd = sd;
            /* windows started ahead of the one being consumed.  Two: gathering a window's source frames and building its super frames
             * takes a host about as long as consuming a window does, and the search itself another ~0.5 s */
            d->la.depth = (int)env_long("MVX_VS_LOOKAHEAD_DEPTH", 2);
            if (d->la.depth < 0) d->la.depth = 0;
            if (d->la.depth > LA_DEPTH_MAX) d->la.depth = LA_DEPTH_MAX;
            { /* what the windows in flight pin must fit the device: every window holds B + delta super frames (with their shadow planes)
               * and B blobs; half of the free memory is left to the other vector clips' share, Degrain's frames and the allocator's pool */
                size_t fr = 0, tot = 0;
                if (env_long("MVX_VS_LOOKAHEAD_AUTOSIZE", 1) && mvx_dev_mem_info(&fr, &tot) == 0 && fr) {
                    const int delta = d->ad.nDeltaFrame > 0 ? d->ad.nDeltaFrame : 1;
                    const double per = (double)sd->geo.bytes * 1.2 + (double)d->blobSize, avail = 0.5 * (double)fr;
                    while (d->la.depth > 0 && (double)(d->la.depth + 1) * (d->la.B + delta) * per > avail) d->la.depth--;
                    while (d->la.B > 8 && (double)(d->la.depth + 1) * (d->la.B + delta) * per > avail) d->la.B /= 2;
                    if ((double)(d->la.depth + 1) * (d->la.B + delta) * per > avail) d->la.on = 0; /* not even one small window: the per-frame path */
                }
            }
        }
        if (d->la.on) {
            d->la.srcNode = vs->addNodeRef(sd->node);
            pthread_mutex_init(&d->la.mu, NULL); pthread_cond_init(&d->la.cv, NULL);
            for (int i = 0; i < LA_SLOTS; i++) d->la.win[i].w = -1;
            deps[1].source = d->la.srcNode; ndeps = 2;
        }
    }
    vs->createVideoFilter(out, "Analyse", vi, analyseGetFrame, analyseFree, fmParallel, deps, ndeps, d, core);
    if (d->la.on) { /* remember the node the core made for this instance (identity only: no reference is kept) */
        int e = 0;
        VSNode *o = vs->mapGetNode(out, "clip", 0, &e);
        if (o && !e) { analyse_register(o, d); vs->freeNode(o); }
    }
}

/* ------------------------------------------------------------------------------------------------ mv.Finest */

typedef struct FinestData { VSNode *super; VSVideoInfo vi; mvx_super *sup; SuperGeo geo; ptrdiff_t pitch[3]; } FinestData;

static const VSFrame *VS_CC finestGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    FinestData *d = (FinestData *)inst;
    if (reason == arInitial) { vs->requestFrameFilter(n, d->super, ctx); return NULL; }
    if (reason != arAllFramesReady) return NULL;
    const VSFrame *ref = vs->getFrameFilter(n, d->super, ctx);
    VSFrame *dst = vs->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, ref, core);
    const int np = d->vi.format.numPlanes, bps = d->vi.format.bytesPerSample;
    DevRef ds;
    int rc = super_to_device(&ds, ref, &d->geo, vs);
    size_t off[3], total = 0;
    for (int p = 0; p < np; p++) { off[p] = total; total += (size_t)d->pitch[p] * vs->getFrameHeight(dst, p); }
    void *arena = rc ? NULL : mvx_dev_alloc(total);
    if (!rc && !arena) rc = MVX_E_NOMEM;
    if (!rc) {
        const void *src3[3]; void *dst3[3];
        for (int p = 0; p < 3; p++) { src3[p] = ds.plane[p]; dst3[p] = p < np ? (char *)arena + off[p] : NULL; }
        rc = mvx_finest_frames(d->sup, 1, src3, d->geo.pitch, dst3, d->pitch, thread_stream());
        for (int p = 0; p < np && !rc; p++)
            rc = timed_download(vs->getWritePtr(dst, p), vs->getStride(dst, p), dst3[p], d->pitch[p], (size_t)vs->getFrameWidth(dst, p) * bps, (size_t)vs->getFrameHeight(dst, p));
        if (!rc) rc = mvx_stream_sync(thread_stream());
    }
    dev_release(&ds);
    shell_quiesce(rc);
    if (arena) mvx_dev_free(arena);
    vs->freeFrame(ref);
    if (rc) { vs->freeFrame(dst); vs->setFilterError(rc == MVX_E_NOMEM ? "Finest: out of memory." : mvx_last_error(), ctx); return NULL; }
    return dst;
}
static void VS_CC finestFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    FinestData *d = (FinestData *)inst;
    vs->freeNode(d->super); mvx_super_destroy(d->sup); free(d);
}
static void VS_CC finestCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    (void)user;
    VSNode *super = vs->mapGetNode(in, "super", 0, 0);
    const VSVideoInfo *vi = vs->getVideoInfo(super);
    if (!mvx_vsh_is_constant_video_format(vi) || vi->format.bitsPerSample > 16 || vi->format.sampleType != stInteger || vi->format.subSamplingW > 1 ||
        vi->format.subSamplingH > 1 || (vi->format.colorFamily != cfYUV && vi->format.colorFamily != cfGray)) {
        vs->mapSetError(out, "Finest: input clip must be GRAY, 420, 422, 440, or 444, up to 16 bits, with constant dimensions.");
        vs->freeNode(super);
        return;
    }
    char err[1400] = "";
    mvx_super *sup = super_from_props(super, "Finest", err, sizeof(err), vs);
    if (!sup) { vs->mapSetError(out, err); vs->freeNode(super); return; }
    FinestData *d = (FinestData *)calloc(1, sizeof(*d));
    d->super = super; d->sup = sup; d->vi = *vi;
    super_geo(&d->geo, sup);
    int32_t w, h;
    mvx_finest_size(sup, &w, &h);
    d->vi.width = w; d->vi.height = h;
    for (int p = 0; p < 3; p++) {
        const int pw = p ? w >> vi->format.subSamplingW : w;
        d->pitch[p] = ((ptrdiff_t)pw * vi->format.bytesPerSample + 255) / 256 * 256;
    }
    VSFilterDependency deps[1] = { { super, rpStrictSpatial } };
    vs->createVideoFilter(out, "Finest", &d->vi, finestGetFrame, finestFree, fmParallel, deps, 1, d, core);
}

/* ------------------------------------------------------------------------------------------------ mv.SCDetection */

typedef struct ScdData { VSNode *node, *vectors; const VSVideoInfo *vi; mvx_analysis_data ad; int64_t thscd1; int32_t thscd2; } ScdData;

static const VSFrame *VS_CC scdGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    ScdData *d = (ScdData *)inst;
    if (reason == arInitial) { vs->requestFrameFilter(n, d->vectors, ctx); vs->requestFrameFilter(n, d->node, ctx); return NULL; }
    if (reason != arAllFramesReady) return NULL;
    const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
    VSFrame *dst = vs->copyFrame(src, core);
    vs->freeFrame(src);
    const VSFrame *mvn = vs->getFrameFilter(n, d->vectors, ctx);
    void *dblob = NULL;
    int rc = blob_to_device(&dblob, NULL, &d->ad, mvn, vs);
    vs->freeFrame(mvn);
    int32_t sc = 0;
    char lerr[MVX_ERRLEN];
    if (!rc) { const void *b1[1] = { dblob }; rc = mvx_scdetect(&d->ad, d->thscd1, d->thscd2, 1, b1, &sc, thread_stream(), lerr); }
    shell_quiesce(rc);
    if (dblob) mvx_dev_free(dblob);
    if (rc) { vs->freeFrame(dst); vs->setFilterError(rc == MVX_E_ARG ? "SCDetection: vector clip frame without a valid MVTools_vectors property." : mvx_last_error(), ctx); return NULL; }
    vs->mapSetInt(vs->getFramePropertiesRW(dst), d->ad.isBackward ? "_SceneChangeNext" : "_SceneChangePrev", sc, maReplace); /* :62-64 */
    return dst;
}
static void VS_CC scdFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    ScdData *d = (ScdData *)inst;
    vs->freeNode(d->node); vs->freeNode(d->vectors); free(d);
}
static void VS_CC scdCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    (void)user;
    ScdData *d = (ScdData *)calloc(1, sizeof(*d));
    char err[1400] = "";
    d->thscd1 = opt_int64(in, "thscd1", vs); d->thscd2 = opt_int(in, "thscd2", vs);
    d->vectors = vs->mapGetNode(in, "vectors", 0, NULL);
    adata_from_clip(&d->ad, d->vectors, "SCDetection", "vectors", err, sizeof(err), vs);
    if (!err[0] && (d->thscd1 == (int64_t)MVX_UNSET ? 400 : d->thscd1) > 8 * 8 * 255) snprintf(err, sizeof(err), "SCDetection: thscd1 can be at most %d.", 8 * 8 * 255);
    if (err[0]) { vs->mapSetError(out, err); vs->freeNode(d->vectors); free(d); return; }
    d->node = vs->mapGetNode(in, "clip", 0, NULL);
    d->vi = vs->getVideoInfo(d->node);
    VSFilterDependency deps[2] = { { d->node, rpStrictSpatial }, { d->vectors, rpStrictSpatial } };
    vs->createVideoFilter(out, "SCDetection", d->vi, scdGetFrame, scdFree, fmParallel, deps, 2, d, core);
}

/* ------------------------------------------------------------------------------------------------ mv.Recalculate */

typedef struct RecalcData { VSNode *node, *vectors; const VSVideoInfo *vi; mvx_super *sup; SuperGeo geo; mvx_recalculate *rc; mvx_analysis_data ad, old; int blobSize; FieldOpt fo; } RecalcData;

static int recalc_nref(const RecalcData *d, int n) { /* src/MVRecalculate.c:78-85 */
    const int off = d->ad.nDeltaFrame;
    if (off > 0) return d->ad.isBackward ? n + off : n - off;
    return -off;
}

static const VSFrame *VS_CC recalcGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    RecalcData *d = (RecalcData *)inst;
    const int nref = recalc_nref(d, n);
    const int haveRef = nref >= 0 && nref < d->vi->numFrames;
    if (reason == arInitial) {
        vs->requestFrameFilter(n, d->vectors, ctx);
        if (haveRef && nref < n) vs->requestFrameFilter(nref, d->node, ctx);
        vs->requestFrameFilter(n, d->node, ctx);
        if (haveRef && nref > n) vs->requestFrameFilter(nref, d->node, ctx);
        return NULL;
    }
    if (reason != arAllFramesReady) return NULL;
    const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
    const VSFrame *ref = haveRef ? vs->getFrameFilter(nref, d->node, ctx) : NULL;
    if (d->fo.fields) { /* src/MVRecalculate.c:121-163: only the error survives, the shift itself never reaches a result */
        int missing = 0;
        frame_top_field(&d->fo, src, n, &missing, vs);
        if (ref) frame_top_field(&d->fo, ref, nref, &missing, vs);
        if (missing) {
            vs->setFilterError("Recalculate: _Field property not found in input frame. Therefore, you must pass tff argument.", ctx);
            if (ref) vs->freeFrame(ref);
            vs->freeFrame(src);
            return NULL;
        }
    }
    const VSFrame *mvn = vs->getFrameFilter(n, d->vectors, ctx);
    DevRef ds, dr;
    memset(&dr, 0, sizeof(dr));
    void *oldBlob = NULL;
    int rc = blob_to_device(&oldBlob, NULL, &d->old, mvn, vs);
    vs->freeFrame(mvn);
    if (!rc) rc = super_to_device(&ds, src, &d->geo, vs); else memset(&ds, 0, sizeof(ds));
    if (!rc && ref) rc = super_to_device(&dr, ref, &d->geo, vs);
    void *dblob = rc ? NULL : mvx_dev_alloc((size_t)d->blobSize);
    char *blob = (char *)malloc((size_t)d->blobSize);
    if (!rc && (!dblob || !blob)) rc = MVX_E_NOMEM;
    if (!rc) {
        mvx_recalculate_job job;
        memset(&job, 0, sizeof(job));
        for (int p = 0; p < 3; p++) { job.src[p] = ds.plane[p]; job.ref[p] = ref ? dr.plane[p] : NULL; }
        job.old_blob = oldBlob; job.blob = dblob;
        rc = mvx_recalculate_frames(d->rc, 1, &job, thread_stream());
        if (!rc) rc = timed_download(blob, d->blobSize, dblob, d->blobSize, (size_t)d->blobSize, 1);
        if (!rc) rc = mvx_stream_sync(thread_stream());
    }
    shell_quiesce(rc);
    dev_release(&ds); dev_release(&dr);
    if (dblob) mvx_dev_free(dblob);
    if (oldBlob) mvx_dev_free(oldBlob);
    if (ref) vs->freeFrame(ref);
    VSFrame *dst = NULL;
    if (!rc) { /* src/MVRecalculate.c:218-235 */
        dst = vs->copyFrame(src, core);
        VSMap *props = vs->getFramePropertiesRW(dst);
        vs->mapSetData(props, PROP_ADATA, (const char *)&d->ad, sizeof(d->ad), dtBinary, maReplace);
        vs->mapSetData(props, PROP_VECTORS, blob, d->blobSize, dtBinary, maReplace);
    } else
        vs->setFilterError(rc == MVX_E_ARG ? "Recalculate: vector clip frame without a valid MVTools_vectors property." : rc == MVX_E_NOMEM ? "Recalculate: out of memory." : mvx_last_error(), ctx);
    free(blob);
    vs->freeFrame(src);
    return dst;
}

static void VS_CC recalcFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    RecalcData *d = (RecalcData *)inst;
    vs->freeNode(d->node); vs->freeNode(d->vectors);
    mvx_recalculate_destroy(d->rc);
    mvx_super_destroy(d->sup);
    free(d);
}

static void VS_CC recalcCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    (void)user;
    static const char *keys[] = { "thsad", "smooth", "blksize", "blksizev", "search", "searchparam", "lambda", "chroma", "truemotion", "pnew", "overlap", "overlapv",
                                  "divide", "meander", "fields", "dct" };
    mvx_recalculate_args a;
    int64_t *av = (int64_t *)&a;
    for (size_t i = 0; i < sizeof(keys) / sizeof(keys[0]); i++) av[i] = opt_int64(in, keys[i], vs);
    RecalcData *d = (RecalcData *)calloc(1, sizeof(*d));
    char err[1400] = "";
    field_opt(&d->fo, in, vs);
    d->node = vs->mapGetNode(in, "super", 0, 0);
    d->vi = vs->getVideoInfo(d->node);
    d->sup = super_from_props(d->node, "Recalculate", err, sizeof(err), vs);
    if (!err[0]) { d->vectors = vs->mapGetNode(in, "vectors", 0, NULL); adata_from_clip(&d->old, d->vectors, "Recalculate", "vectors", err, sizeof(err), vs); }
    if (!err[0]) {
        super_geo(&d->geo, d->sup);
        char lerr[MVX_ERRLEN];
        if (mvx_recalculate_create(&a, d->sup, &d->old, d->geo.pitch, &d->rc, lerr)) snprintf(err, sizeof(err), "%s", lerr);
    }
    if (err[0]) {
        vs->mapSetError(out, err);
        if (d->node) vs->freeNode(d->node);
        if (d->vectors) vs->freeNode(d->vectors);
        if (d->sup) mvx_super_destroy(d->sup);
        free(d);
        return;
    }
    mvx_recalculate_get_data(d->rc, &d->ad);
    d->blobSize = mvx_recalculate_blob_size(d->rc);
    VSFilterDependency deps[2] = { { d->node, rpGeneral }, { d->vectors, rpStrictSpatial } };
    vs->createVideoFilter(out, "Recalculate", d->vi, recalcGetFrame, recalcFree, fmParallel, deps, 2, d, core);
}

/* ------------------------------------------------------------------------------------------------ admission gate of the consuming filters (r6)
 * A host asks for as many output frames at once as it has worker threads -- a VapourSynth core starts one per logical CPU, 256 on the bench box.  Every output frame
 * in flight pins its 2 * radius + 1 super frames and pulls the look-ahead windows of the vector clips forward; past ~3/4 of a look-ahead window of frames in flight the
 * requests span three windows, the device cache turns over and 640 4K16 frames take 3.8-5.2 s (128 threads), 7-9.6 s (192) or 12-14 s (256) instead of 2.5 s (96):
 * profiles/r6_vs_shell_threads_96_to_256.txt.  So mv.DegrainN / mv.Compensate / mv.BlockFPS admit at most MVX_VS_MAX_INFLIGHT (default 96) of their own output frames
 * at a time, the lowest frame numbers first; a request beyond that waits at arInitial -- before it has asked for anything -- until an admitted frame is delivered or
 * fails.  The permit travels in the request's frame data (*fd), so arAllFramesReady / arError of the same request give it back.  One gate per filter instance: a chain of
 * such filters cannot deadlock on a shared pool. */
typedef struct GateWaiter { int n; pthread_cond_t cv; } GateWaiter; /* (one condition variable per waiter: a delivered frame wakes the ONE request it admits, not all of them) */
typedef struct Gate { pthread_mutex_t mu; int limit, inflight, nwait; long done; GateWaiter *waiting[1024]; } Gate;
static void gate_init(Gate *g) {
    pthread_mutex_init(&g->mu, NULL);
    g->limit = (int)env_long("MVX_VS_MAX_INFLIGHT", 96); /* (<= 0: no gate) */
    g->inflight = g->nwait = 0; g->done = 0;
}
static void gate_free(Gate *g) {
    if (g->inflight || g->nwait) fprintf(stderr, "mvtools_vs: admission gate freed with %d permits out, %d requests waiting\n", g->inflight, g->nwait); /* (a request that never ended: a bug here or in the host) */
    pthread_mutex_destroy(&g->mu);
}
static GateWaiter *gate_lowest(const Gate *g) { /* mu held */
    GateWaiter *w = NULL;
    for (int i = 0; i < g->nwait; i++) if (!w || g->waiting[i]->n < w->n) w = g->waiting[i];
    return w;
}
/* A waiter occupies one of the host's worker threads.  A host that has more requests outstanding than workers could end up with every worker waiting here and none
 * left for the upstream filters of the admitted frames; so the waiter that is next in line and sees NO admitted frame finish for a quarter of a second -- frames take
 * milliseconds -- goes on over the limit: slow, as without the gate, never stuck. */
static void gate_enter(Gate *g, int n, void **fd) {
    if (g->limit <= 0 || *fd) return;
    pthread_mutex_lock(&g->mu);
    if ((g->inflight >= g->limit || g->nwait) && g->nwait < 1024) {
        GateWaiter me;
        me.n = n;
        pthread_cond_init(&me.cv, NULL);
        g->waiting[g->nwait++] = &me;
        for (;;) {
            const int lowest = gate_lowest(g) == &me;
            if (g->inflight < g->limit && lowest) break;
            struct timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_nsec += 250000000L;
            if (ts.tv_nsec >= 1000000000L) { ts.tv_nsec -= 1000000000L; ts.tv_sec++; }
            const long seen = g->done;
            const int rc = pthread_cond_timedwait(&me.cv, &g->mu, &ts);
            if (rc == ETIMEDOUT && gate_lowest(g) == &me && g->done == seen) break;
        }
        for (int i = 0; i < g->nwait; i++) if (g->waiting[i] == &me) { g->waiting[i] = g->waiting[--g->nwait]; break; }
        pthread_cond_destroy(&me.cv);
        g->inflight++;
        if (g->inflight < g->limit && g->nwait) pthread_cond_signal(&gate_lowest(g)->cv); /* (room for the next one too) */
    } else g->inflight++;
    *fd = (void *)g;
    pthread_mutex_unlock(&g->mu);
}
static void gate_leave(Gate *g, void **fd) {
    if (!*fd) return;
    *fd = NULL;
    pthread_mutex_lock(&g->mu);
    g->inflight--; g->done++;
    if (g->nwait && g->inflight < g->limit) pthread_cond_signal(&gate_lowest(g)->cv);
    pthread_mutex_unlock(&g->mu);
}

/* ------------------------------------------------------------------------------------------------ mv.Degrain1..6 */

typedef struct DegrainData {
    VSNode *node, *super, *vectors[12];
    const VSVideoInfo *vi;
    int radius;
    mvx_super *sup; SuperGeo geo;
    mvx_degrain *dg;
    mvx_analysis_data ad[12];
    AnalyseData *an[12]; /* the mv.Analyse instances behind the vector clips, where they are this plugin's (r6: their vectors are read on the device) */
    ptrdiff_t pitch[3]; /* device pitch of clip / output planes */
    int blobSize;
    char name[16];
    Gate gate;
} DegrainData;

typedef struct FirstFrameReq { const VSAPI *vs; VSNode *node; } FirstFrameReq;
static void *first_frame_thread(void *arg) { /* (errors are reported by the caller's own read of the frame) */
    FirstFrameReq *q = (FirstFrameReq *)arg;
    char msg[256];
    const VSFrame *f = q->vs->getFrame(0, q->node, msg, sizeof(msg));
    if (f) q->vs->freeFrame(f);
    return NULL;
}

static const VSFrame *VS_CC degrainGetFrameUngated(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    DegrainData *d = (DegrainData *)inst;
    const int nr = 2 * d->radius;
    if (reason == arInitial) { /* src/MVDegrains.cpp:92-109 */
        for (int r = 0; r < nr; r += 2) {
            vs->requestFrameFilter(n, d->vectors[r], ctx);
            vs->requestFrameFilter(n, d->vectors[r + 1], ctx);
            const int offB = d->ad[r].nDeltaFrame, offF = -d->ad[r + 1].nDeltaFrame;
            if (n + offB < d->vi->numFrames) vs->requestFrameFilter(n + offB, d->super, ctx);
            if (n + offF >= 0) vs->requestFrameFilter(n + offF, d->super, ctx);
        }
        vs->requestFrameFilter(n, d->node, ctx);
        return NULL;
    }
    if (reason != arAllFramesReady) return NULL;
    const double tgf = prof_now();
    const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
    const int np = d->vi->format.numPlanes, bps = d->vi->format.bytesPerSample;
    mvx_degrain_job job;
    memset(&job, 0, sizeof(job));
    DevRef refs[12];
    void *blobArena[12];
    LaWindow *blobHeld[12];
    memset(refs, 0, sizeof(refs)); memset(blobArena, 0, sizeof(blobArena)); memset(blobHeld, 0, sizeof(blobHeld));
    int rc = 0;
    void *srcArena = NULL, *dsrc[3];
    rc = upload_plane_set(dsrc, &srcArena, src, d->pitch, np, bps, vs);
    size_t dstOff[3], dstBytes = 0;
    for (int p = 0; p < np; p++) { dstOff[p] = dstBytes; dstBytes += (size_t)d->pitch[p] * vs->getFrameHeight(src, p); }
    void *dstArena = shell_alloc(dstBytes);
    if (!rc && (!srcArena || !dstArena)) rc = MVX_E_NOMEM;
    for (int p = 0; p < np && !rc; p++) { job.src[p] = dsrc[p]; job.dst[p] = (char *)dstArena + dstOff[p]; }
    for (int r = 0; r < nr && !rc; r++) {
        const void *onDevice = la_device_blob(d->an[r], n, &blobHeld[r]);
        if (onDevice) job.blobs[r] = (void *)onDevice; /* (the window that made the frame's property: the same bytes) */
        else {
            const VSFrame *vf = vs->getFrameFilter(n, d->vectors[r], ctx);
            rc = blob_to_device(&blobArena[r], NULL, &d->ad[r], vf, vs);
            job.blobs[r] = blobArena[r];
            vs->freeFrame(vf);
        }
        const int nref = (r & 1) ? n - d->ad[r].nDeltaFrame : n + d->ad[r].nDeltaFrame;
        if (!rc && nref >= 0 && nref < d->vi->numFrames) {
            const VSFrame *sf = vs->getFrameFilter(nref, d->super, ctx);
            rc = super_to_device(&refs[r], sf, &d->geo, vs);
            vs->freeFrame(sf);
            for (int p = 0; p < 3; p++) job.refs[r][p] = refs[r].plane[p];
        }
    }
    const double tdg = prof_now();
    if (!rc) rc = mvx_degrain_frames(d->dg, 1, &job, thread_stream());
    if (!rc) rc = mvx_stream_sync(thread_stream());
    prof_add(PF_DEGRAIN, tdg);
    VSFrame *dst = NULL;
    if (!rc) {
        dst = vs->newVideoFrame(&d->vi->format, d->vi->width, d->vi->height, src, core);
        for (int p = 0; p < np && !rc; p++) /* (the kernels were waited for above: the download stream, beside the other threads' uploads) */
            rc = timed_download_on(download_stream(), vs->getWritePtr(dst, p), vs->getStride(dst, p), job.dst[p], d->pitch[p], (size_t)vs->getFrameWidth(dst, p) * bps, (size_t)vs->getFrameHeight(dst, p));
    }
    shell_quiesce(rc);
    for (int r = 0; r < nr; r++) { dev_release(&refs[r]); if (blobArena[r]) mvx_dev_free(blobArena[r]); la_device_blob_release(d->an[r], blobHeld[r]); }
    if (srcArena) mvx_dev_free(srcArena);
    if (dstArena) mvx_dev_free(dstArena);
    vs->freeFrame(src);
    if (rc) {
        char msg[MVX_ERRLEN + 64];
        snprintf(msg, sizeof(msg), "%s: %s", d->name, rc == MVX_E_ARG ? "vector clip frame without matching MVTools_vectors property." : rc == MVX_E_NOMEM ? "out of memory." : mvx_last_error());
        if (dst) vs->freeFrame(dst);
        vs->setFilterError(msg, ctx);
        return NULL;
    }
    prof_add(PF_GF_DEGRAIN, tgf);
    if ((n & 31) == 0) la_trace(d, n, "(Degrain output frame of this number done)", prof_now() - tgf);
    return dst;
}

/* (the admission gate around the filter proper: arInitial takes the permit, whatever ends the request -- the frame, a filter error, arError -- returns it) */
static const VSFrame *VS_CC degrainGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    DegrainData *d = (DegrainData *)inst;
    if (reason == arInitial) gate_enter(&d->gate, n, fd);
    const VSFrame *f = degrainGetFrameUngated(n, reason, inst, fd, ctx, core, vs);
    if (reason != arInitial || f) gate_leave(&d->gate, fd);
    return f;
}

static void VS_CC degrainFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    DegrainData *d = (DegrainData *)inst;
    vs->freeNode(d->node); vs->freeNode(d->super);
    for (int r = 0; r < 2 * d->radius; r++) vs->freeNode(d->vectors[r]);
    mvx_degrain_destroy(d->dg);
    mvx_super_destroy(d->sup);
    gate_free(&d->gate);
    free(d);
}

static void VS_CC degrainCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    const int radius = (int)(intptr_t)user;
    static const char *vnames[] = { "mvbw", "mvfw", "mvbw2", "mvfw2", "mvbw3", "mvfw3", "mvbw4", "mvfw4", "mvbw5", "mvfw5", "mvbw6", "mvfw6" };
    DegrainData *d = (DegrainData *)calloc(1, sizeof(*d));
    gate_init(&d->gate);
    d->radius = radius;
    snprintf(d->name, sizeof(d->name), "Degrain%d", radius);
    char err[1400] = "";
    mvx_degrain_args a;
    a.radius = radius; a.thsad = opt_int64(in, "thsad", vs); a.thsadc = opt_int64(in, "thsadc", vs); a.plane = opt_int(in, "plane", vs);
    a.limit = opt_int(in, "limit", vs); a.limitc = opt_int(in, "limitc", vs); a.thscd1 = opt_int64(in, "thscd1", vs); a.thscd2 = opt_int(in, "thscd2", vs);
    if (a.plane != MVX_UNSET && (a.plane < 0 || a.plane > 4)) snprintf(err, sizeof(err), "%s: plane must be between 0 and 4 (inclusive).", d->name);
    if (!err[0]) {
        d->super = vs->mapGetNode(in, "super", 0, NULL);
        d->sup = super_from_props(d->super, d->name, err, sizeof(err), vs);
    }
    const int nr = 2 * radius;
    for (int r = 0; r < nr && !err[0]; r++) d->vectors[r] = vs->mapGetNode(in, vnames[r], 0, NULL);
    if (!err[0] && env_long("MVX_VS_PARALLEL_FIRST", 1)) { /* r6: frame 0 of the vector clips is produced CONCURRENTLY (each is a whole look-ahead window of searches: six launches
                                                             * that share the GPU instead of queueing behind each other); the reads below then find the frames in the host's cache */
        pthread_t th[12];
        FirstFrameReq fr[12];
        int started[12];
        for (int r = 0; r < nr; r++) { fr[r].vs = vs; fr[r].node = d->vectors[r]; started[r] = pthread_create(&th[r], NULL, first_frame_thread, &fr[r]) == 0; }
        for (int r = 0; r < nr; r++) if (started[r]) pthread_join(th[r], NULL);
    }
    for (int r = 0; r < nr && !err[0]; r++) adata_from_clip(&d->ad[r], d->vectors[r], d->name, vnames[r], err, sizeof(err), vs);
    for (int r = 1; r < nr && !err[0]; r++) adata_similar(&d->ad[0], &d->ad[r], d->name, vnames[0], vnames[r], err, sizeof(err));
    if (!err[0]) { /* src/MVDegrains.cpp:606-640 */
        const char *m = NULL;
        for (int r = 0; r < nr; r++) if (d->ad[r].nDeltaFrame <= 0) m = "cannot use motion vectors with absolute frame references.";
        if (!d->ad[0].isBackward) m = "mvbw must be generated with isb=True.";
        if (d->ad[1].isBackward) m = "mvfw must be generated with isb=False.";
        for (int k = 1; k < radius; k++) {
            if (!d->ad[2 * k].isBackward) m = "mvbw must be generated with isb=True.";
            if (d->ad[2 * k + 1].isBackward) m = "mvfw must be generated with isb=False.";
            if (d->ad[2 * k].nDeltaFrame <= d->ad[2 * k - 2].nDeltaFrame) m = "mvbwN must have greater delta than mvbwP.";
            if (d->ad[2 * k + 1].nDeltaFrame <= d->ad[2 * k - 1].nDeltaFrame) m = "mvfwN must have greater delta than mvfwP.";
        }
        if (m) snprintf(err, sizeof(err), "%s: %s", d->name, m);
    }
    if (!err[0]) {
        d->node = vs->mapGetNode(in, "clip", 0, NULL);
        d->vi = vs->getVideoInfo(d->node);
        const VSVideoInfo *svi = vs->getVideoInfo(d->super);
        super_geo(&d->geo, d->sup);
        if (!mvx_vsh_is_constant_video_format(d->vi) || d->vi->format.bitsPerSample > 16 || d->vi->format.sampleType != stInteger || d->vi->format.subSamplingW > 1 ||
            d->vi->format.subSamplingH > 1 || (d->vi->format.colorFamily != cfYUV && d->vi->format.colorFamily != cfGray))
            snprintf(err, sizeof(err), "%s: input clip must be GRAY, 420, 422, 440, or 444, up to 16 bits, with constant dimensions.", d->name);
        else if (d->geo.si.height != d->vi->height || d->geo.si.super_width != svi->width || d->geo.si.super_height != svi->height || d->geo.si.width != d->vi->width ||
                 d->vi->format.bitsPerSample != svi->format.bitsPerSample || d->vi->format.subSamplingW != svi->format.subSamplingW || d->vi->format.subSamplingH != svi->format.subSamplingH)
            snprintf(err, sizeof(err), "%s: wrong source or super clip frame size.", d->name);
    }
    if (!err[0]) {
        const int bps = d->vi->format.bytesPerSample;
        for (int p = 0; p < 3; p++) {
            const int w = p ? d->vi->width >> d->vi->format.subSamplingW : d->vi->width;
            d->pitch[p] = ((ptrdiff_t)w * bps + 255) / 256 * 256;
        }
        char lerr[MVX_ERRLEN];
        if (mvx_degrain_create(&a, &d->ad[0], d->sup, d->pitch, d->geo.pitch, d->pitch, &d->dg, lerr)) snprintf(err, sizeof(err), "%s", lerr);
        else if (d->geo.copies > 1) mvx_degrain_set_ref_shadow(d->dg, d->geo.shadowStride); /* every device super frame of this shell carries its copies */
    }
    if (!err[0]) d->blobSize = mvx_vectors_size(&d->ad[0]);
    if (!err[0] && env_long("MVX_VS_DEVICE_VECTORS", 1)) for (int r = 0; r < nr; r++) d->an[r] = analyse_lookup(d->vectors[r]);
    if (err[0]) {
        vs->mapSetError(out, err);
        if (d->node) vs->freeNode(d->node);
        if (d->super) vs->freeNode(d->super);
        for (int r = 0; r < nr; r++) if (d->vectors[r]) vs->freeNode(d->vectors[r]);
        if (d->dg) mvx_degrain_destroy(d->dg);
        if (d->sup) mvx_super_destroy(d->sup);
        free(d);
        return;
    }
    VSFilterDependency deps[14];
    deps[0].source = d->node; deps[0].requestPattern = rpStrictSpatial;
    deps[1].source = d->super; deps[1].requestPattern = rpGeneral;
    for (int r = 0; r < nr; r++) { deps[2 + r].source = d->vectors[r]; deps[2 + r].requestPattern = rpStrictSpatial; }
    vs->createVideoFilter(out, d->name, d->vi, degrainGetFrame, degrainFree, fmParallel, deps, 2 + nr, d, core);
}

/* ------------------------------------------------------------------------------------------------ mv.Compensate */

typedef struct CompData { VSNode *node, *super, *vectors; const VSVideoInfo *vi; mvx_super *sup; SuperGeo geo; mvx_compensate *cp; mvx_analysis_data ad; ptrdiff_t pitch[3]; int blobSize; FieldOpt fo; Gate gate; } CompData;

static int comp_nref(const CompData *d, int n) { /* src/MVCompensate.c:84-92 */
    if (d->ad.nDeltaFrame > 0) return n + (d->ad.isBackward ? d->ad.nDeltaFrame : -d->ad.nDeltaFrame);
    return -d->ad.nDeltaFrame;
}

static const VSFrame *VS_CC compGetFrameUngated(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    CompData *d = (CompData *)inst;
    const int nref = comp_nref(d, n);
    const int haveRef = nref >= 0 && nref < d->vi->numFrames;
    if (reason == arInitial) {
        vs->requestFrameFilter(n, d->vectors, ctx);
        if (haveRef && nref < n) vs->requestFrameFilter(nref, d->super, ctx);
        vs->requestFrameFilter(n, d->super, ctx);
        if (haveRef && nref > n) vs->requestFrameFilter(nref, d->super, ctx);
        vs->requestFrameFilter(n, d->node, ctx);
        return NULL;
    }
    if (reason != arAllFramesReady) return NULL;
    const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
    const VSFrame *ssup = vs->getFrameFilter(n, d->super, ctx);
    const VSFrame *rsup = haveRef ? vs->getFrameFilter(nref, d->super, ctx) : NULL;
    const VSFrame *vf = vs->getFrameFilter(n, d->vectors, ctx);
    const int np = d->vi->format.numPlanes, bps = d->vi->format.bytesPerSample;
    int fieldShift = 0;
    if (rsup && d->fo.fields && d->ad.nPel > 1 && ((nref - n) % 2 != 0)) { /* src/MVCompensate.c:188-225 (the props of the two super frames).
        * The reference only looks at _Field when the vectors are usable (:162); usability is decided on the device here, so a missing _Field is
        * reported for scene-change frames too. */
        int missing = 0;
        const int srcTop = frame_top_field(&d->fo, ssup, n, &missing, vs);
        const int refTop = frame_top_field(&d->fo, rsup, nref, &missing, vs);
        if (missing) {
            vs->setFilterError("Compensate: _Field property not found in input frame. Therefore, you must pass tff argument.", ctx);
            vs->freeFrame(vf); vs->freeFrame(ssup); vs->freeFrame(rsup); vs->freeFrame(src);
            return NULL;
        }
        fieldShift = field_shift_of(srcTop, refTop, d->ad.nPel);
    }
    DevRef ds, dr;
    memset(&dr, 0, sizeof(dr));
    int rc = super_to_device(&ds, ssup, &d->geo, vs);
    if (!rc && rsup) rc = super_to_device(&dr, rsup, &d->geo, vs);
    void *dblob = NULL;
    if (!rc) rc = blob_to_device(&dblob, NULL, &d->ad, vf, vs);
    size_t dstOff[3], dstBytes = 0;
    for (int p = 0; p < np; p++) { dstOff[p] = dstBytes; dstBytes += (size_t)d->pitch[p] * vs->getFrameHeight(src, p); }
    void *dstArena = rc ? NULL : mvx_dev_alloc(dstBytes);
    if (!rc && !dstArena) rc = MVX_E_NOMEM;
    mvx_compensate_job job;
    memset(&job, 0, sizeof(job));
    if (!rc) {
        for (int p = 0; p < 3; p++) { job.src_super[p] = ds.plane[p]; job.ref_super[p] = rsup ? dr.plane[p] : NULL; }
        for (int p = 0; p < np; p++) job.dst[p] = (char *)dstArena + dstOff[p];
        job.blob = dblob;
        job.field_shift = fieldShift;
        if (!rc) rc = mvx_compensate_frames(d->cp, 1, &job, thread_stream());
    }
    VSFrame *dst = NULL;
    if (!rc) {
        dst = vs->newVideoFrame(&d->vi->format, d->vi->width, d->vi->height, src, core);
        for (int p = 0; p < np && !rc; p++)
            rc = timed_download(vs->getWritePtr(dst, p), vs->getStride(dst, p), job.dst[p], d->pitch[p], (size_t)vs->getFrameWidth(dst, p) * bps, (size_t)vs->getFrameHeight(dst, p));
        if (!rc) rc = mvx_stream_sync(thread_stream());
    }
    shell_quiesce(rc);
    dev_release(&ds); dev_release(&dr);
    if (dblob) mvx_dev_free(dblob);
    if (dstArena) mvx_dev_free(dstArena);
    vs->freeFrame(vf); vs->freeFrame(ssup); if (rsup) vs->freeFrame(rsup);
    vs->freeFrame(src);
    if (rc) {
        if (dst) vs->freeFrame(dst);
        vs->setFilterError(rc == MVX_E_ARG ? "Compensate: vector clip frame without matching MVTools_vectors property." : rc == MVX_E_NOMEM ? "Compensate: out of memory." : mvx_last_error(), ctx);
        return NULL;
    }
    return dst;
}

/* (the admission gate around the filter proper: arInitial takes the permit, whatever ends the request -- the frame, a filter error, arError -- returns it) */
static const VSFrame *VS_CC compGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    CompData *d = (CompData *)inst;
    if (reason == arInitial) gate_enter(&d->gate, n, fd);
    const VSFrame *f = compGetFrameUngated(n, reason, inst, fd, ctx, core, vs);
    if (reason != arInitial || f) gate_leave(&d->gate, fd);
    return f;
}

static void VS_CC compFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    CompData *d = (CompData *)inst;
    vs->freeNode(d->node); vs->freeNode(d->super); vs->freeNode(d->vectors);
    mvx_compensate_destroy(d->cp);
    mvx_super_destroy(d->sup);
    gate_free(&d->gate);
    free(d);
}

static void VS_CC compCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    (void)user;
    CompData *d = (CompData *)calloc(1, sizeof(*d));
    gate_init(&d->gate);
    char err[1400] = "";
    mvx_compensate_args a;
    a.scbehavior = opt_int(in, "scbehavior", vs); a.thsad = opt_int64(in, "thsad", vs); a.thscd1 = opt_int64(in, "thscd1", vs); a.thscd2 = opt_int(in, "thscd2", vs);
    int e = 0;
    a.time = vs->mapGetFloat(in, "time", 0, &e);
    if (e) a.time = 100.0;
    field_opt(&d->fo, in, vs);
    a.fields = d->fo.fields;
    if (!err[0]) {
        d->super = vs->mapGetNode(in, "super", 0, NULL);
        d->sup = super_from_props(d->super, "Compensate", err, sizeof(err), vs);
    }
    if (!err[0]) {
        d->vectors = vs->mapGetNode(in, "vectors", 0, NULL);
        adata_from_clip(&d->ad, d->vectors, "Compensate", "vectors", err, sizeof(err), vs);
    }
    if (!err[0]) {
        d->node = vs->mapGetNode(in, "clip", 0, NULL);
        d->vi = vs->getVideoInfo(d->node);
        super_geo(&d->geo, d->sup);
        const VSVideoInfo *svi = vs->getVideoInfo(d->super);
        if (d->geo.si.height != d->vi->height || d->geo.si.width != d->vi->width || d->geo.si.super_width != svi->width || d->geo.si.super_height != svi->height)
            snprintf(err, sizeof(err), "Compensate: wrong source or super clip frame size.");
    }
    if (!err[0]) {
        const int bps = d->vi->format.bytesPerSample;
        for (int p = 0; p < 3; p++) {
            const int w = p ? d->vi->width >> d->vi->format.subSamplingW : d->vi->width;
            d->pitch[p] = ((ptrdiff_t)w * bps + 255) / 256 * 256;
        }
        char lerr[MVX_ERRLEN];
        if (mvx_compensate_create(&a, &d->ad, d->sup, d->geo.pitch, d->pitch, &d->cp, lerr)) snprintf(err, sizeof(err), "%s", lerr);
    }
    if (!err[0]) d->blobSize = mvx_vectors_size(&d->ad);
    if (err[0]) {
        vs->mapSetError(out, err);
        if (d->node) vs->freeNode(d->node);
        if (d->super) vs->freeNode(d->super);
        if (d->vectors) vs->freeNode(d->vectors);
        if (d->cp) mvx_compensate_destroy(d->cp);
        if (d->sup) mvx_super_destroy(d->sup);
        free(d);
        return;
    }
    VSFilterDependency deps[3] = { { d->node, rpStrictSpatial }, { d->super, rpGeneral }, { d->vectors, rpStrictSpatial } };
    vs->createVideoFilter(out, "Compensate", d->vi, compGetFrame, compFree, fmParallel, deps, 3, d, core);
}

/* ------------------------------------------------------------------------------------------------ mv.BlockFPS */

typedef struct FpsData { VSNode *node, *super, *mvbw, *mvfw; const VSVideoInfo *oldvi; VSVideoInfo vi; mvx_super *sup; SuperGeo geo; mvx_blockfps *bf;
                         mvx_analysis_data bw, fw; ptrdiff_t pitch[3]; int blobSize; Gate gate; } FpsData;

static int fps_min(int a, int b) { return a < b ? a : b; }

static const VSFrame *VS_CC fpsGetFrameUngated(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    FpsData *d = (FpsData *)inst;
    int nleft, nright, time256;
    mvx_blockfps_map(d->bf, n, &nleft, &nright, &time256);
    const int last = d->oldvi->numFrames - 1;
    const int good = nleft < d->oldvi->numFrames && nright < d->oldvi->numFrames;
    if (reason == arInitial) { /* src/MVBlockFPS.c:236-276 */
        if (time256 == 0) { vs->requestFrameFilter(fps_min(nleft, last), d->node, ctx); return NULL; }
        if (time256 == 256) { vs->requestFrameFilter(fps_min(nright, last), d->node, ctx); return NULL; }
        if (good) {
            vs->requestFrameFilter(nright, d->mvfw, ctx);
            vs->requestFrameFilter(nleft, d->mvbw, ctx);
            vs->requestFrameFilter(nleft, d->super, ctx);
            vs->requestFrameFilter(nright, d->super, ctx);
        }
        vs->requestFrameFilter(fps_min(nleft, last), d->node, ctx);
        vs->requestFrameFilter(fps_min(nright, last), d->node, ctx); /* the reference only asks for it when blend=1; harmless */
        return NULL;
    }
    if (reason != arAllFramesReady) return NULL;
    if (time256 == 0) return vs->getFrameFilter(fps_min(nleft, last), d->node, ctx);   /* simply left  (:285-286) */
    if (time256 == 256) return vs->getFrameFilter(fps_min(nright, last), d->node, ctx); /* simply right (:287-288) */
    const VSFrame *cl = vs->getFrameFilter(fps_min(nleft, last), d->node, ctx);
    const VSFrame *cr = vs->getFrameFilter(fps_min(nright, last), d->node, ctx);
    const int np = d->vi.format.numPlanes, bps = d->vi.format.bytesPerSample;
    mvx_blockfps_job job;
    memset(&job, 0, sizeof(job));
    job.time256 = time256;
    void *arenaL = NULL, *arenaR = NULL, *dl[3], *dr[3], *blobF = NULL, *blobB = NULL;
    int rc = upload_plane_set(dl, &arenaL, cl, d->pitch, np, bps, vs);
    if (!rc) rc = upload_plane_set(dr, &arenaR, cr, d->pitch, np, bps, vs);
    size_t dstOff[3], dstBytes = 0;
    for (int p = 0; p < np; p++) { dstOff[p] = dstBytes; dstBytes += (size_t)d->pitch[p] * vs->getFrameHeight(cl, p); }
    void *dstArena = shell_alloc(dstBytes);
    if (!rc && (!arenaL || !arenaR || !dstArena)) rc = MVX_E_NOMEM;
    for (int p = 0; p < np && !rc; p++) { job.clip_left[p] = dl[p]; job.clip_right[p] = dr[p]; job.dst[p] = (char *)dstArena + dstOff[p]; }
    DevRef ds, dr2;
    memset(&ds, 0, sizeof(ds)); memset(&dr2, 0, sizeof(dr2));
    if (!rc && good) {
        const VSFrame *sl = vs->getFrameFilter(nleft, d->super, ctx), *sr = vs->getFrameFilter(nright, d->super, ctx);
        const VSFrame *vf = vs->getFrameFilter(nright, d->mvfw, ctx), *vb = vs->getFrameFilter(nleft, d->mvbw, ctx);
        rc = super_to_device(&ds, sl, &d->geo, vs);
        if (!rc) rc = super_to_device(&dr2, sr, &d->geo, vs);
        if (!rc) rc = blob_to_device(&blobF, NULL, &d->fw, vf, vs);
        if (!rc) rc = blob_to_device(&blobB, NULL, &d->bw, vb, vs);
        for (int p = 0; p < 3; p++) { job.src_super[p] = ds.plane[p]; job.ref_super[p] = dr2.plane[p]; }
        job.blob_fw = blobF; job.blob_bw = blobB;
        vs->freeFrame(sl); vs->freeFrame(sr); vs->freeFrame(vf); vs->freeFrame(vb);
    }
    if (!rc) rc = mvx_blockfps_frames(d->bf, 1, &job, thread_stream());
    VSFrame *dst = NULL;
    if (!rc) {
        dst = vs->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, cl, core);
        for (int p = 0; p < np && !rc; p++)
            rc = timed_download(vs->getWritePtr(dst, p), vs->getStride(dst, p), job.dst[p], d->pitch[p], (size_t)vs->getFrameWidth(dst, p) * bps, (size_t)vs->getFrameHeight(dst, p));
        if (!rc) rc = mvx_stream_sync(thread_stream());
    }
    dev_release(&ds); dev_release(&dr2);
    shell_quiesce(rc);
    if (blobF) mvx_dev_free(blobF);
    if (blobB) mvx_dev_free(blobB);
    if (arenaL) mvx_dev_free(arenaL);
    if (arenaR) mvx_dev_free(arenaR);
    if (dstArena) mvx_dev_free(dstArena);
    vs->freeFrame(cl); vs->freeFrame(cr);
    if (rc) {
        if (dst) vs->freeFrame(dst);
        vs->setFilterError(rc == MVX_E_ARG ? "BlockFPS: vector clip frame without matching MVTools_vectors property." : rc == MVX_E_NOMEM ? "BlockFPS: out of memory." : mvx_last_error(), ctx);
        return NULL;
    }
    return dst;
}

/* (the admission gate around the filter proper: arInitial takes the permit, whatever ends the request -- the frame, a filter error, arError -- returns it) */
static const VSFrame *VS_CC fpsGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    FpsData *d = (FpsData *)inst;
    if (reason == arInitial) gate_enter(&d->gate, n, fd);
    const VSFrame *f = fpsGetFrameUngated(n, reason, inst, fd, ctx, core, vs);
    if (reason != arInitial || f) gate_leave(&d->gate, fd);
    return f;
}

static void VS_CC fpsFree(void *inst, VSCore *core, const VSAPI *vs) {
    (void)core;
    FpsData *d = (FpsData *)inst;
    vs->freeNode(d->node); vs->freeNode(d->super); vs->freeNode(d->mvbw); vs->freeNode(d->mvfw);
    mvx_blockfps_destroy(d->bf);
    mvx_super_destroy(d->sup);
    gate_free(&d->gate);
    free(d);
}

static void VS_CC fpsCreate(const VSMap *in, VSMap *out, void *user, VSCore *core, const VSAPI *vs) {
    warm_barrier();
    (void)user;
    FpsData *d = (FpsData *)calloc(1, sizeof(*d));
    gate_init(&d->gate);
    char err[1400] = "";
    mvx_blockfps_args a;
    a.num = opt_int64(in, "num", vs); a.den = opt_int64(in, "den", vs); a.mode = opt_int(in, "mode", vs); a.blend = opt_int(in, "blend", vs);
    a.thscd1 = opt_int64(in, "thscd1", vs); a.thscd2 = opt_int(in, "thscd2", vs);
    int e = 0;
    a.ml = vs->mapGetFloat(in, "ml", 0, &e);
    if (e) a.ml = 100.0;
    if (a.mode != MVX_UNSET && (a.mode < 0 || a.mode > 8)) snprintf(err, sizeof(err), "BlockFPS: mode must be between 0 and 8 (inclusive).");
    if (!err[0]) { d->super = vs->mapGetNode(in, "super", 0, NULL); d->sup = super_from_props(d->super, "BlockFPS", err, sizeof(err), vs); }
    if (!err[0]) { d->mvbw = vs->mapGetNode(in, "mvbw", 0, NULL); adata_from_clip(&d->bw, d->mvbw, "BlockFPS", "mvbw", err, sizeof(err), vs); }
    if (!err[0]) { d->mvfw = vs->mapGetNode(in, "mvfw", 0, NULL); adata_from_clip(&d->fw, d->mvfw, "BlockFPS", "mvfw", err, sizeof(err), vs); }
    if (!err[0]) {
        d->node = vs->mapGetNode(in, "clip", 0, NULL);
        d->oldvi = vs->getVideoInfo(d->node);
        d->vi = *d->oldvi;
        super_geo(&d->geo, d->sup);
        const int bps = d->vi.format.bytesPerSample;
        for (int p = 0; p < 3; p++) {
            const int w = p ? d->vi.width >> d->vi.format.subSamplingW : d->vi.width;
            d->pitch[p] = ((ptrdiff_t)w * bps + 255) / 256 * 256;
        }
        char lerr[MVX_ERRLEN];
        if (mvx_blockfps_create(&a, &d->bw, &d->fw, d->sup, d->oldvi->numFrames, d->oldvi->fpsNum, d->oldvi->fpsDen, d->geo.pitch, d->pitch, d->pitch, &d->bf, lerr))
            snprintf(err, sizeof(err), "%s", lerr);
        else if (!mvx_vsh_is_constant_video_format(&d->vi) || d->vi.format.bitsPerSample > 16 || d->vi.format.sampleType != stInteger || d->vi.format.subSamplingW > 1 ||
                 d->vi.format.subSamplingH > 1 || (d->vi.format.colorFamily != cfYUV && d->vi.format.colorFamily != cfGray))
            snprintf(err, sizeof(err), "BlockFPS: input clip must be GRAY, 420, 422, 440, or 444, up to 16 bits, with constant dimensions.");
    }
    if (err[0]) {
        vs->mapSetError(out, err);
        if (d->node) vs->freeNode(d->node);
        if (d->super) vs->freeNode(d->super);
        if (d->mvbw) vs->freeNode(d->mvbw);
        if (d->mvfw) vs->freeNode(d->mvfw);
        if (d->bf) mvx_blockfps_destroy(d->bf);
        if (d->sup) mvx_super_destroy(d->sup);
        free(d);
        return;
    }
    mvx_blockfps_info info;
    mvx_blockfps_get_info(d->bf, &info);
    d->vi.numFrames = info.num_frames; d->vi.fpsNum = info.fps_num; d->vi.fpsDen = info.fps_den;
    d->blobSize = mvx_vectors_size(&d->bw);
    VSFilterDependency deps[4] = { { d->node, rpGeneral }, { d->super, rpGeneral }, { d->mvbw, rpGeneral }, { d->mvfw, rpGeneral } };
    vs->createVideoFilter(out, "BlockFPS", &d->vi, fpsGetFrame, fpsFree, fmParallel, deps, 4, d, core);
    /* AssumeFPS sets the _DurationNum / _DurationDen frame properties (src/MVBlockFPS.c:989-1014) */
    VSNode *node = vs->mapGetNode(out, "clip", 0, NULL);
    VSMap *args = vs->createMap();
    vs->mapSetNode(args, "clip", node, maReplace);
    vs->freeNode(node);
    vs->mapSetInt(args, "fpsnum", info.fps_num, maReplace);
    vs->mapSetInt(args, "fpsden", info.fps_den, maReplace);
    VSPlugin *std = vs->getPluginByID("com.vapoursynth.std", core);
    VSMap *ret = vs->invoke(std, "AssumeFPS", args);
    vs->freeMap(args);
    if (vs->mapGetError(ret)) {
        char msg[600];
        snprintf(msg, sizeof(msg), "BlockFPS: Failed to invoke AssumeFPS. Error message: %s", vs->mapGetError(ret));
        vs->mapSetError(out, msg);
        vs->freeMap(ret);
        return;
    }
    node = vs->mapGetNode(ret, "clip", 0, NULL);
    vs->freeMap(ret);
    vs->mapSetNode(out, "clip", node, maReplace);
    vs->freeNode(node);
}

/* ------------------------------------------------------------------------------------------------ entry point */

#define DEGRAIN_TAIL "thsad:int:opt;thsadc:int:opt;plane:int:opt;limit:int:opt;limitc:int:opt;thscd1:int:opt;thscd2:int:opt;opt:int:opt;"

/* replaces src/EntryPoint.c:28-52 for the filters of the hot path */
VS_EXTERNAL_API(void) VapourSynthPluginInit2(VSPlugin *plugin, const VSPLUGINAPI *vspapi) {
    /* (the search streams want GPU_MAX_HW_QUEUES >= 12 -- see search_stream_next.  That is a setting of the HOST PROCESS, read once when the HIP
     * runtime initialises: a plugin must not change the process environment behind its host's back, so it is documented in INTEGRATION.md and set
     * by the mini host / the launch scripts, not here.  r6: an unconfigured host is TOLD once, under MVX_VS_STATS, that its searches will share
     * the runtime's default four hardware queues) */
    if (env_long("MVX_VS_STATS", 0)) {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        if (!q || atoi(q) < SEARCH_STREAMS)
            fprintf(stderr, "mvtools (MI355X): GPU_MAX_HW_QUEUES is %s: the %d search streams of concurrent mv.Analyse instances share the HIP runtime's hardware queues and "
                            "serialise; export GPU_MAX_HW_QUEUES=16 before starting the host (INTEGRATION.md)\n", q ? q : "unset (runtime default: 4)", SEARCH_STREAMS);
    }
    /* r6: the HIP runtime, this library's code objects and the page-locked staging buffers come up on a background thread while the host is still reading its
     * script / opening its clips: 0.3 s + 0.1 s + ~20 ms per buffer that the first frames (mv.DegrainN's creation reads frame 0 of every vector clip) no longer wait for.
     * (see warmup_thread) */
    if (env_long("MVX_VS_WARMUP", 1)) {
        if (!__atomic_load_n(&g_warm_started, __ATOMIC_ACQUIRE)) {
            __atomic_store_n(&g_warm_started, 1, __ATOMIC_RELEASE);
            if (pthread_create(&g_warm_thread, NULL, warmup_thread, NULL) != 0) __atomic_store_n(&g_warm_started, 0, __ATOMIC_RELEASE);
        }
    }
    vspapi->configPlugin("com.nodame.mvtools", "mv", "MVTools v24", VS_MAKE_VERSION(24, 0), VS_MAKE_VERSION(VAPOURSYNTH_API_MAJOR, VAPOURSYNTH_API_MINOR), 0, plugin);
    vspapi->registerFunction("Super",
                             "clip:vnode;hpad:int:opt;vpad:int:opt;pel:int:opt;levels:int:opt;chroma:int:opt;sharp:int:opt;rfilter:int:opt;pelclip:vnode:opt;opt:int:opt;",
                             "clip:vnode;", superCreate, NULL, plugin);
    vspapi->registerFunction("Analyse",
                             "super:vnode;blksize:int:opt;blksizev:int:opt;levels:int:opt;search:int:opt;searchparam:int:opt;pelsearch:int:opt;isb:int:opt;lambda:int:opt;"
                             "chroma:int:opt;delta:int:opt;truemotion:int:opt;lsad:int:opt;plevel:int:opt;global:int:opt;pnew:int:opt;pzero:int:opt;pglobal:int:opt;"
                             "overlap:int:opt;overlapv:int:opt;divide:int:opt;badsad:int:opt;badrange:int:opt;opt:int:opt;meander:int:opt;trymany:int:opt;fields:int:opt;"
                             "tff:int:opt;search_coarse:int:opt;dct:int:opt;",
                             "clip:vnode;", analyseCreate, NULL, plugin);
    vspapi->registerFunction("Finest", "super:vnode;opt:int:opt;", "clip:vnode;", finestCreate, NULL, plugin);
    vspapi->registerFunction("SCDetection", "clip:vnode;vectors:vnode;thscd1:int:opt;thscd2:int:opt;", "clip:vnode;", scdCreate, NULL, plugin);
    vspapi->registerFunction("Recalculate",
                             "super:vnode;vectors:vnode;thsad:int:opt;smooth:int:opt;blksize:int:opt;blksizev:int:opt;search:int:opt;searchparam:int:opt;lambda:int:opt;"
                             "chroma:int:opt;truemotion:int:opt;pnew:int:opt;overlap:int:opt;overlapv:int:opt;divide:int:opt;opt:int:opt;meander:int:opt;fields:int:opt;"
                             "tff:int:opt;dct:int:opt;",
                             "clip:vnode;", recalcCreate, NULL, plugin);
    vspapi->registerFunction("Degrain1", "clip:vnode;super:vnode;mvbw:vnode;mvfw:vnode;" DEGRAIN_TAIL, "clip:vnode;", degrainCreate, (void *)(intptr_t)1, plugin);
    vspapi->registerFunction("Degrain2", "clip:vnode;super:vnode;mvbw:vnode;mvfw:vnode;mvbw2:vnode;mvfw2:vnode;" DEGRAIN_TAIL, "clip:vnode;", degrainCreate, (void *)(intptr_t)2, plugin);
    vspapi->registerFunction("Degrain3", "clip:vnode;super:vnode;mvbw:vnode;mvfw:vnode;mvbw2:vnode;mvfw2:vnode;mvbw3:vnode;mvfw3:vnode;" DEGRAIN_TAIL, "clip:vnode;", degrainCreate, (void *)(intptr_t)3, plugin);
    vspapi->registerFunction("Degrain4", "clip:vnode;super:vnode;mvbw:vnode;mvfw:vnode;mvbw2:vnode;mvfw2:vnode;mvbw3:vnode;mvfw3:vnode;mvbw4:vnode;mvfw4:vnode;" DEGRAIN_TAIL, "clip:vnode;", degrainCreate, (void *)(intptr_t)4, plugin);
    vspapi->registerFunction("Degrain5", "clip:vnode;super:vnode;mvbw:vnode;mvfw:vnode;mvbw2:vnode;mvfw2:vnode;mvbw3:vnode;mvfw3:vnode;mvbw4:vnode;mvfw4:vnode;mvbw5:vnode;mvfw5:vnode;" DEGRAIN_TAIL, "clip:vnode;", degrainCreate, (void *)(intptr_t)5, plugin);
    vspapi->registerFunction("Degrain6", "clip:vnode;super:vnode;mvbw:vnode;mvfw:vnode;mvbw2:vnode;mvfw2:vnode;mvbw3:vnode;mvfw3:vnode;mvbw4:vnode;mvfw4:vnode;mvbw5:vnode;mvfw5:vnode;mvbw6:vnode;mvfw6:vnode;" DEGRAIN_TAIL, "clip:vnode;", degrainCreate, (void *)(intptr_t)6, plugin);
    vspapi->registerFunction("BlockFPS",
                             "clip:vnode;super:vnode;mvbw:vnode;mvfw:vnode;num:int:opt;den:int:opt;mode:int:opt;ml:float:opt;blend:int:opt;thscd1:int:opt;thscd2:int:opt;opt:int:opt;",
                             "clip:vnode;", fpsCreate, NULL, plugin);
    vspapi->registerFunction("Compensate",
                             "clip:vnode;super:vnode;vectors:vnode;scbehavior:int:opt;thsad:int:opt;fields:int:opt;time:float:opt;thscd1:int:opt;thscd2:int:opt;opt:int:opt;tff:int:opt;",
                             "clip:vnode;", compCreate, NULL, plugin);
}

/*
 * minihost.c -- a minimal VapourSynth-API-4-shaped host for testing libmvtools_vs.so without a VapourSynth installation
 * (test infrastructure; shares vs4_api.h with the plugin, see the caveat there).  Single-threaded by default; with
 * x.threads=N the output frames are requested by N worker threads at once, so that the filters' getFrame callbacks run
 * concurrently like under VapourSynth's fmParallel (a frame being produced by one thread makes the others wait for it).
 *
 * It implements exactly the VSAPI / VSPLUGINAPI members the mvtools hot path uses: maps, frames, nodes, the two-phase
 * getFrame protocol (arInitial -> requestFrameFilter ... -> arAllFramesReady) evaluated synchronously and recursively,
 * and createVideoFilter.  Frames of every node are kept (no eviction): clips in the tests are a handful of frames.
 *
 *   mvx_vs_host <plugin.so> list
 *   mvx_vs_host <plugin.so> error  <Filter> <w> <h> <bits> [f.key=value ...]       -> prints the creation error (or OK)
 *   mvx_vs_host <plugin.so> run <pipeline> <in.raw> <w> <h> <bits> <nframes> <out.raw> [s.|a.|d.|c.key=value ...]
 *       pipeline: super | finest | analyse | scdetection (d.*) | recalculate (r.*) | degrainN | compensate | blockfps (b.*)
 *       in.raw  : nframes x (Y, U, V planes, 4:2:0, tightly packed, little endian); x.format=422 | 444 | gray (one plane): other chroma formats
 *       out.raw : super      -> every super frame (planes tightly packed) ; props of frame 0 on stdout
 *                 analyse    -> per frame: 84-byte MVTools_MVAnalysisData + MVTools_vectors, backward (isb=1) then forward
 *                 degrainN / compensate -> output frames, planes tightly packed
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

#include "vs4_api.h"

/* ---------------------------------------------------------------------------------------------------- maps */

typedef struct Item { int64_t i; double f; char *data; int size; int hint; VSNode *node; } Item;
typedef struct Entry { char *key; int type; int n; Item *v; struct Entry *next; } Entry;
struct VSMap { Entry *head; char *error; };

static VSNode *node_addref(VSNode *n);
static void node_free(VSNode *n);
static pthread_mutex_t g_host_mu = PTHREAD_MUTEX_INITIALIZER; /* reference counts and the per-node frame tables */
static pthread_cond_t g_host_cv = PTHREAD_COND_INITIALIZER;

static VSMap *VS_CC createMap(void) { return (VSMap *)calloc(1, sizeof(VSMap)); }
static void VS_CC clearMap(VSMap *m) {
    for (Entry *e = m->head; e;) {
        Entry *nx = e->next;
        for (int i = 0; i < e->n; i++) { free(e->v[i].data); if (e->v[i].node) node_free(e->v[i].node); }
        free(e->v); free(e->key); free(e);
        e = nx;
    }
    m->head = NULL;
    free(m->error); m->error = NULL;
}
static void VS_CC freeMap(VSMap *m) { if (m) { clearMap(m); free(m); } }
static Entry *find(const VSMap *m, const char *key) { for (Entry *e = m->head; e; e = e->next) if (!strcmp(e->key, key)) return e; return NULL; }
static Item *put(VSMap *m, const char *key, int type, int append) {
    Entry *e = find(m, key);
    if (e && (append == maReplace || e->type != type)) {
        for (int i = 0; i < e->n; i++) { free(e->v[i].data); if (e->v[i].node) node_free(e->v[i].node); }
        e->n = 0; e->type = type;
    }
    if (!e) { e = (Entry *)calloc(1, sizeof(Entry)); e->key = strdup(key); e->type = type; e->next = m->head; m->head = e; }
    e->v = (Item *)realloc(e->v, sizeof(Item) * (e->n + 1));
    memset(&e->v[e->n], 0, sizeof(Item));
    return &e->v[e->n++];
}
static const Item *get(const VSMap *m, const char *key, int index, int type, int *error) {
    Entry *e = find(m, key);
    int err = peSuccess;
    if (!e) err = peUnset; else if (e->type != type) err = peType; else if (index < 0 || index >= e->n) err = peIndex;
    if (error) *error = err;
    else if (err) { fprintf(stderr, "minihost: fatal property access %s\n", key); abort(); }
    return err ? NULL : &e->v[index];
}
static void VS_CC mapSetError(VSMap *m, const char *msg) { clearMap(m); m->error = strdup(msg ? msg : "Error: no error specified"); }
static const char *VS_CC mapGetError(const VSMap *m) { return m->error; }
static int VS_CC mapNumElements(const VSMap *m, const char *key) { Entry *e = find(m, key); return e ? e->n : -1; }
static int64_t VS_CC mapGetInt(const VSMap *m, const char *k, int i, int *err) { const Item *it = get(m, k, i, ptInt, err); return it ? it->i : 0; }
static int VS_CC mapGetIntSaturated(const VSMap *m, const char *k, int i, int *err) {
    int64_t v = mapGetInt(m, k, i, err);
    return v > 2147483647LL ? 2147483647 : v < -2147483647LL - 1 ? -2147483647 - 1 : (int)v;
}
static int VS_CC mapSetInt(VSMap *m, const char *k, int64_t v, int append) { put(m, k, ptInt, append)->i = v; return 0; }
static double VS_CC mapGetFloat(const VSMap *m, const char *k, int i, int *err) { const Item *it = get(m, k, i, ptFloat, err); return it ? it->f : 0; }
static int VS_CC mapSetFloat(VSMap *m, const char *k, double v, int append) { put(m, k, ptFloat, append)->f = v; return 0; }
static const char *VS_CC mapGetData(const VSMap *m, const char *k, int i, int *err) { const Item *it = get(m, k, i, ptData, err); return it ? it->data : NULL; }
static int VS_CC mapGetDataSize(const VSMap *m, const char *k, int i, int *err) { const Item *it = get(m, k, i, ptData, err); return it ? it->size : -1; }
static int VS_CC mapSetData(VSMap *m, const char *k, const char *d, int size, int type, int append) {
    if (size < 0) size = (int)strlen(d);
    Item *it = put(m, k, ptData, append);
    it->data = (char *)malloc((size_t)size + 1); memcpy(it->data, d, (size_t)size); it->data[size] = 0; it->size = size; it->hint = type;
    return 0;
}
static VSNode *VS_CC mapGetNode(const VSMap *m, const char *k, int i, int *err) { const Item *it = get(m, k, i, ptVideoNode, err); return it ? node_addref(it->node) : NULL; }
static int VS_CC mapSetNode(VSMap *m, const char *k, VSNode *n, int append) { put(m, k, ptVideoNode, append)->node = node_addref(n); return 0; }
static void copy_map(VSMap *dst, const VSMap *src) {
    for (Entry *e = src->head; e; e = e->next)
        for (int i = 0; i < e->n; i++) {
            if (e->type == ptInt) mapSetInt(dst, e->key, e->v[i].i, maAppend);
            else if (e->type == ptFloat) mapSetFloat(dst, e->key, e->v[i].f, maAppend);
            else if (e->type == ptData) mapSetData(dst, e->key, e->v[i].data, e->v[i].size, e->v[i].hint, maAppend);
            else if (e->type == ptVideoNode) mapSetNode(dst, e->key, e->v[i].node, maAppend);
        }
}

/* ---------------------------------------------------------------------------------------------------- frames */

/* planes are reference-counted buffers shared between a frame and its copies, copied when somebody asks for a write pointer:
 * VapourSynth's copy-on-write frames (mv.Analyse's copyFrame of a 131 MB super frame costs nothing there, and must not here) */
struct VSFrame { int refs; VSVideoFormat fmt; int w, h; uint8_t *data[3]; int *planeRefs[3]; ptrdiff_t stride[3]; VSMap *props; };

static int plane_w(const VSFrame *f, int p) { return p ? f->w >> f->fmt.subSamplingW : f->w; }
static int plane_h(const VSFrame *f, int p) { return p ? f->h >> f->fmt.subSamplingH : f->h; }

/* large planes on huge pages where the kernel offers them: a 131 MB super frame is 32 000 page faults otherwise, and the faults of
 * 64 worker threads serialise on the process's memory map (VapourSynth recycles frame buffers instead) */
/* freed plane buffers are kept for reuse, like the frame memory pool of a real core: a recycled buffer costs no page faults */
#define POOL_MAX 4096
static struct { void *p; size_t size; } g_pool[POOL_MAX];
static int g_pool_n;
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static void plane_free(void *p, size_t size) {
    pthread_mutex_lock(&g_pool_mu);
    if (g_pool_n < POOL_MAX) { g_pool[g_pool_n].p = p; g_pool[g_pool_n].size = size; g_pool_n++; p = NULL; }
    pthread_mutex_unlock(&g_pool_mu);
    free(p);
}
static uint8_t *plane_alloc(size_t size) {
    void *p = NULL;
    pthread_mutex_lock(&g_pool_mu);
    for (int i = g_pool_n - 1; i >= 0; i--) if (g_pool[i].size == size) { p = g_pool[i].p; g_pool[i] = g_pool[--g_pool_n]; break; }
    pthread_mutex_unlock(&g_pool_mu);
    if (p) return (uint8_t *)p;
    const size_t big = (size_t)2 << 20;
    if (size >= 2 * big) {
        if (posix_memalign(&p, big, (size + big - 1) / big * big)) abort();
        (void)madvise(p, (size + big - 1) / big * big, MADV_HUGEPAGE);
    } else if (posix_memalign(&p, 64, size)) abort();
    return (uint8_t *)p;
}
static VSFrame *VS_CC newVideoFrame(const VSVideoFormat *fmt, int w, int h, const VSFrame *propSrc, VSCore *core) {
    (void)core;
    VSFrame *f = (VSFrame *)calloc(1, sizeof(VSFrame));
    f->refs = 1; f->fmt = *fmt; f->w = w; f->h = h; f->props = createMap();
    for (int p = 0; p < fmt->numPlanes; p++) {
        f->stride[p] = ((ptrdiff_t)plane_w(f, p) * fmt->bytesPerSample + 63) / 64 * 64;
        f->data[p] = plane_alloc((size_t)f->stride[p] * plane_h(f, p));
        /* new frames are uninitialised in VapourSynth: poison them so that a filter that forgets to write shows up (small frames always; big
         * ones only with MVX_HOST_POISON=1 -- a pass over a 131 MB super frame per frame is what a throughput run should not measure) */
        if ((size_t)f->stride[p] * plane_h(f, p) < ((size_t)8 << 20) || getenv("MVX_HOST_POISON")) memset(f->data[p], 0xCD, (size_t)f->stride[p] * plane_h(f, p));
        f->planeRefs[p] = (int *)malloc(sizeof(int)); *f->planeRefs[p] = 1;
    }
    if (propSrc) copy_map(f->props, propSrc->props);
    return f;
}
static void VS_CC freeFrame(const VSFrame *cf) {
    VSFrame *f = (VSFrame *)cf;
    if (!f) return;
    pthread_mutex_lock(&g_host_mu);
    const int left = --f->refs;
    int drop[3] = { 0, 0, 0 };
    if (!left) for (int p = 0; p < 3; p++) if (f->planeRefs[p] && --*f->planeRefs[p] == 0) drop[p] = 1;
    pthread_mutex_unlock(&g_host_mu);
    if (left) return;
    for (int p = 0; p < 3; p++) if (drop[p]) { plane_free(f->data[p], (size_t)f->stride[p] * plane_h(f, p)); free(f->planeRefs[p]); }
    freeMap(f->props); free(f);
}
static const VSFrame *frame_addref(const VSFrame *f) { pthread_mutex_lock(&g_host_mu); ((VSFrame *)f)->refs++; pthread_mutex_unlock(&g_host_mu); return f; }
static VSFrame *VS_CC copyFrame(const VSFrame *s, VSCore *core) {
    (void)core;
    VSFrame *f = (VSFrame *)calloc(1, sizeof(VSFrame));
    f->refs = 1; f->fmt = s->fmt; f->w = s->w; f->h = s->h; f->props = createMap();
    pthread_mutex_lock(&g_host_mu);
    for (int p = 0; p < s->fmt.numPlanes; p++) { f->data[p] = s->data[p]; f->stride[p] = s->stride[p]; f->planeRefs[p] = s->planeRefs[p]; ++*f->planeRefs[p]; }
    pthread_mutex_unlock(&g_host_mu);
    copy_map(f->props, s->props);
    return f;
}
static const VSMap *VS_CC getFramePropertiesRO(const VSFrame *f) { return f->props; }
static VSMap *VS_CC getFramePropertiesRW(VSFrame *f) { return f->props; }
static ptrdiff_t VS_CC getStride(const VSFrame *f, int p) { return f->stride[p]; }
static const uint8_t *VS_CC getReadPtr(const VSFrame *f, int p) { return f->data[p]; }
static uint8_t *VS_CC getWritePtr(VSFrame *f, int p) {
    pthread_mutex_lock(&g_host_mu);
    const int shared = *f->planeRefs[p] > 1;
    pthread_mutex_unlock(&g_host_mu);
    if (shared) { /* copy on write */
        const size_t size = (size_t)f->stride[p] * plane_h(f, p);
        uint8_t *d;
        d = plane_alloc(size);
        memcpy(d, f->data[p], size);
        int *rc = (int *)malloc(sizeof(int)); *rc = 1;
        pthread_mutex_lock(&g_host_mu);
        --*f->planeRefs[p]; /* (cannot reach 0: it was > 1 and this frame held one of the references) */
        pthread_mutex_unlock(&g_host_mu);
        f->data[p] = d; f->planeRefs[p] = rc;
    }
    return f->data[p];
}
static const VSVideoFormat *VS_CC getVideoFrameFormat(const VSFrame *f) { return &f->fmt; }
static int VS_CC getFrameWidth(const VSFrame *f, int p) { return plane_w(f, p); }
static int VS_CC getFrameHeight(const VSFrame *f, int p) { return plane_h(f, p); }

/* ---------------------------------------------------------------------------------------------------- nodes */

struct VSNode { int refs; VSVideoInfo vi; VSFilterGetFrame getFrame; VSFilterFree freeFn; void *inst; const VSFrame **cache; unsigned char *busy; char name[32]; int ncached, lowest; };
static int g_cache_limit; /* x.cache=N: a node keeps at most N frames (the lowest-numbered ones go first), 0 = all, like the bounded caches of a real core */
static struct VSNode *g_uncapped; /* the graph's OUTPUT node keeps every frame whatever x.cache says: the result file is written from them after the timed interval (a client of a real core writes them as they arrive) */
struct VSFrameContext { int n; struct { int n; VSNode *node; const VSFrame *f; } req[1024]; int nreq; char error[1024]; };

static VSAPI g_api;
static VSNode *node_addref(VSNode *n) { pthread_mutex_lock(&g_host_mu); n->refs++; pthread_mutex_unlock(&g_host_mu); return n; }
static void node_free(VSNode *n) {
    if (!n) return;
    pthread_mutex_lock(&g_host_mu);
    const int left = --n->refs;
    pthread_mutex_unlock(&g_host_mu);
    if (left) return;
    for (int i = 0; i < n->vi.numFrames; i++) if (n->cache[i]) freeFrame(n->cache[i]);
    if (n->freeFn) n->freeFn(n->inst, NULL, &g_api);
    free(n->cache); free(n->busy); free(n);
}
static void VS_CC freeNode(VSNode *n) { node_free(n); }
static VSNode *VS_CC addNodeRef(VSNode *n) { return node_addref(n); }
static const VSVideoInfo *VS_CC getVideoInfo(VSNode *n) { return &n->vi; }

static double g_wait_s; /* thread-seconds spent waiting for a frame another thread is producing (MVX_HOST_TIMES) */
static double mono_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static const VSFrame *eval_frame(int n, VSNode *node, char *err, int errsz) {
    if (n < 0) n = 0;
    if (n >= node->vi.numFrames) n = node->vi.numFrames - 1;
    pthread_mutex_lock(&g_host_mu); /* one producer per frame; the others wait for it (a real core would park the request) */
    if (node->busy && node->busy[n] && !node->cache[n]) {
        const double t0 = mono_s();
        while (node->busy[n] && !node->cache[n]) pthread_cond_wait(&g_host_cv, &g_host_mu);
        g_wait_s += mono_s() - t0;
    }
    if (node->cache[n]) { ((VSFrame *)node->cache[n])->refs++; const VSFrame *hit = node->cache[n]; pthread_mutex_unlock(&g_host_mu); return hit; }
    if (node->busy) node->busy[n] = 1;
    pthread_mutex_unlock(&g_host_mu);
    if (!node->getFrame) { snprintf(err, (size_t)errsz, "source frame %d missing", n); return NULL; }
    VSFrameContext ctx;
    memset(&ctx, 0, sizeof(ctx));
    ctx.n = n;
    void *fd = NULL;
    const VSFrame *out = node->getFrame(n, arInitial, node->inst, &fd, &ctx, NULL, &g_api);
    if (!out && !ctx.error[0]) {
        for (int i = 0; i < ctx.nreq; i++) {
            ctx.req[i].f = eval_frame(ctx.req[i].n, ctx.req[i].node, ctx.error, sizeof(ctx.error));
            if (!ctx.req[i].f) break;
        }
        if (!ctx.error[0]) out = node->getFrame(n, arAllFramesReady, node->inst, &fd, &ctx, NULL, &g_api);
        else (void)node->getFrame(n, arError, node->inst, &fd, &ctx, NULL, &g_api); /* (a real core tells the filter that the request failed, so that it can drop its frame data) */
    }
    for (int i = 0; i < ctx.nreq; i++) if (ctx.req[i].f) freeFrame(ctx.req[i].f);
    const VSFrame *evict[8];
    int nev = 0;
    pthread_mutex_lock(&g_host_mu);
    if (out && !node->cache[n]) { node->cache[n] = out; ((VSFrame *)out)->refs++; node->ncached++; if (n < node->lowest) node->lowest = n; }
    if (node->busy) node->busy[n] = 0;
    while (g_cache_limit > 0 && node != g_uncapped && node->ncached > g_cache_limit && nev < 8) { /* drop the oldest frames (this one excepted) */
        int i = node->lowest;
        while (i < node->vi.numFrames && (!node->cache[i] || i == n)) i++;
        if (i >= node->vi.numFrames) break;
        evict[nev++] = node->cache[i]; node->cache[i] = NULL; node->ncached--; node->lowest = i + 1;
    }
    pthread_cond_broadcast(&g_host_cv);
    pthread_mutex_unlock(&g_host_mu);
    for (int i = 0; i < nev; i++) freeFrame(evict[i]);
    if (!out) { snprintf(err, (size_t)errsz, "%s", ctx.error[0] ? ctx.error : "filter returned no frame"); return NULL; }
    return out;
}
static const VSFrame *VS_CC getFrame(int n, VSNode *node, char *err, int sz) { return eval_frame(n, node, err, sz); }
static void VS_CC requestFrameFilter(int n, VSNode *node, VSFrameContext *ctx) {
    for (int i = 0; i < ctx->nreq; i++) if (ctx->req[i].n == n && ctx->req[i].node == node) return;
    if (ctx->nreq >= 1024) { fprintf(stderr, "minihost: too many frame requests\n"); abort(); }
    ctx->req[ctx->nreq].n = n; ctx->req[ctx->nreq].node = node; ctx->req[ctx->nreq].f = NULL; ctx->nreq++;
}
static const VSFrame *VS_CC getFrameFilter(int n, VSNode *node, VSFrameContext *ctx) {
    for (int i = 0; i < ctx->nreq; i++) if (ctx->req[i].n == n && ctx->req[i].node == node && ctx->req[i].f) return frame_addref(ctx->req[i].f);
    fprintf(stderr, "minihost: filter fetched frame %d it did not request\n", n);
    return NULL;
}
static void VS_CC setFilterError(const char *msg, VSFrameContext *ctx) { snprintf(ctx->error, sizeof(ctx->error), "%s", msg ? msg : "unknown error"); }

static void VS_CC createVideoFilter(VSMap *out, const char *name, const VSVideoInfo *vi, VSFilterGetFrame gf, VSFilterFree ff, int mode, const VSFilterDependency *deps, int ndeps, void *inst, VSCore *core) {
    (void)mode; (void)deps; (void)ndeps; (void)core;
    VSNode *n = (VSNode *)calloc(1, sizeof(VSNode));
    n->refs = 1; n->vi = *vi; n->getFrame = gf; n->freeFn = ff; n->inst = inst;
    n->cache = (const VSFrame **)calloc((size_t)vi->numFrames, sizeof(VSFrame *));
    n->busy = (unsigned char *)calloc((size_t)vi->numFrames, 1);
    snprintf(n->name, sizeof(n->name), "%s", name);
    mapSetNode(out, "clip", n, maAppend);
    node_free(n); /* the map holds the reference */
}
static void VS_CC logMessage(int t, const char *msg, VSCore *core) { (void)t; (void)core; fprintf(stderr, "vs: %s\n", msg); }

/* ---------------------------------------------------------------------------------------------------- plugin side */

typedef struct Func { char name[32]; char *args; VSPublicFunction fn; void *user; } Func;
static Func g_funcs[32];
static int g_nfuncs;
static char g_id[128], g_ns[32];

static int VS_CC papiVersion(void) { return VAPOURSYNTH_API_VERSION; }
static int VS_CC configPlugin(const char *id, const char *ns, const char *name, int pv, int av, int flags, VSPlugin *p) {
    (void)name; (void)pv; (void)av; (void)flags; (void)p;
    snprintf(g_id, sizeof(g_id), "%s", id); snprintf(g_ns, sizeof(g_ns), "%s", ns);
    return 1;
}
static int VS_CC registerFunction(const char *name, const char *args, const char *ret, VSPublicFunction fn, void *user, VSPlugin *p) {
    (void)ret; (void)p;
    Func *f = &g_funcs[g_nfuncs++];
    snprintf(f->name, sizeof(f->name), "%s", name); f->args = strdup(args); f->fn = fn; f->user = user;
    return 1;
}
/* invoke with argument checking against the registered signature (unknown keys are errors, like the real core) */
static VSNode *invoke(const char *name, VSMap *in, char *err, size_t errsz) {
    for (int i = 0; i < g_nfuncs; i++)
        if (!strcmp(g_funcs[i].name, name)) {
            for (Entry *e = in->head; e; e = e->next) {
                char pat[80];
                snprintf(pat, sizeof(pat), "%s:", e->key);
                const char *hit = strstr(g_funcs[i].args, pat);
                if (!hit || (hit != g_funcs[i].args && hit[-1] != ';')) { snprintf(err, errsz, "%s: Function does not take argument(s) named %s", name, e->key); return NULL; }
            }
            VSMap *out = createMap();
            g_funcs[i].fn(in, out, g_funcs[i].user, NULL, &g_api);
            VSNode *n = NULL;
            if (mapGetError(out)) snprintf(err, errsz, "%s", mapGetError(out));
            else { int e = 0; n = mapGetNode(out, "clip", 0, &e); if (e) snprintf(err, errsz, "%s: no clip returned", name); }
            freeMap(out);
            return n;
        }
    snprintf(err, errsz, "no function %s", name);
    return NULL;
}

/* the one foreign function the path needs: std.AssumeFPS (sets the clip's frame rate and the per-frame duration props) */
typedef struct AssumeData { VSNode *node; int64_t num, den; } AssumeData;
static const VSFrame *VS_CC assumeGetFrame(int n, int reason, void *inst, void **fd, VSFrameContext *ctx, VSCore *core, const VSAPI *vs) {
    (void)fd;
    AssumeData *d = (AssumeData *)inst;
    if (reason == arInitial) { vs->requestFrameFilter(n, d->node, ctx); return NULL; }
    if (reason != arAllFramesReady) return NULL;
    const VSFrame *src = vs->getFrameFilter(n, d->node, ctx);
    VSFrame *dst = vs->copyFrame(src, core);
    vs->freeFrame(src);
    vs->mapSetInt(vs->getFramePropertiesRW(dst), "_DurationNum", d->den, maReplace);
    vs->mapSetInt(vs->getFramePropertiesRW(dst), "_DurationDen", d->num, maReplace);
    return dst;
}
static void VS_CC assumeFree(void *inst, VSCore *core, const VSAPI *vs) { (void)core; AssumeData *d = (AssumeData *)inst; vs->freeNode(d->node); free(d); }
static struct VSPlugin { int dummy; } *g_std = (struct VSPlugin *)&g_nfuncs;
static VSPlugin *VS_CC getPluginByID(const char *id, VSCore *core) { (void)core; return strcmp(id, "com.vapoursynth.std") ? NULL : g_std; }
static VSMap *VS_CC apiInvoke(VSPlugin *plugin, const char *name, const VSMap *args) {
    VSMap *out = createMap();
    if (plugin != g_std || strcmp(name, "AssumeFPS")) { mapSetError(out, "minihost: only std.AssumeFPS exists"); return out; }
    int e = 0;
    AssumeData *d = (AssumeData *)calloc(1, sizeof(*d));
    d->node = mapGetNode(args, "clip", 0, &e);
    d->num = mapGetInt(args, "fpsnum", 0, &e); d->den = mapGetInt(args, "fpsden", 0, &e);
    VSVideoInfo vi = d->node->vi;
    vi.fpsNum = d->num; vi.fpsDen = d->den;
    createVideoFilter(out, "AssumeFPS", &vi, assumeGetFrame, assumeFree, fmParallel, NULL, 0, d, NULL);
    return out;
}

static void init_api(void) {
    memset(&g_api, 0, sizeof(g_api));
    g_api.createVideoFilter = createVideoFilter; g_api.freeNode = freeNode; g_api.addNodeRef = addNodeRef; g_api.getVideoInfo = getVideoInfo;
    g_api.newVideoFrame = newVideoFrame; g_api.freeFrame = freeFrame; g_api.copyFrame = copyFrame;
    g_api.getFramePropertiesRO = getFramePropertiesRO; g_api.getFramePropertiesRW = getFramePropertiesRW;
    g_api.getStride = getStride; g_api.getReadPtr = getReadPtr; g_api.getWritePtr = getWritePtr; g_api.getVideoFrameFormat = getVideoFrameFormat;
    g_api.getFrameWidth = getFrameWidth; g_api.getFrameHeight = getFrameHeight;
    g_api.getFrame = getFrame; g_api.getFrameFilter = getFrameFilter; g_api.requestFrameFilter = requestFrameFilter; g_api.setFilterError = setFilterError;
    g_api.createMap = createMap; g_api.freeMap = freeMap; g_api.clearMap = clearMap; g_api.mapSetError = mapSetError; g_api.mapGetError = mapGetError;
    g_api.mapNumElements = mapNumElements; g_api.mapGetInt = mapGetInt; g_api.mapGetIntSaturated = mapGetIntSaturated; g_api.mapSetInt = mapSetInt;
    g_api.mapGetFloat = mapGetFloat; g_api.mapSetFloat = mapSetFloat; g_api.mapGetData = mapGetData; g_api.mapGetDataSize = mapGetDataSize; g_api.mapSetData = mapSetData;
    g_api.mapGetNode = mapGetNode; g_api.mapSetNode = mapSetNode; g_api.logMessage = logMessage;
    g_api.getPluginByID = getPluginByID; g_api.invoke = apiInvoke;
}

/* ---------------------------------------------------------------------------------------------------- driver */

static int g_field_order = -1; /* x.fieldorder=0|1: source frame n carries _Field = order ^ (n % 2), like separated fields */

static int g_fmt_ssw = 1, g_fmt_ssh = 1, g_fmt_gray; /* x.format=420 (default) | 422 | 444 | gray: the source clip's chroma format */
static void format_from_args(int argc, char **argv) {
    for (int i = 0; i < argc; i++)
        if (!strncmp(argv[i], "x.format=", 9)) {
            const char *f = argv[i] + 9;
            g_fmt_gray = !strcmp(f, "gray");
            g_fmt_ssw = (!strcmp(f, "444") || g_fmt_gray) ? 0 : 1;
            g_fmt_ssh = !strcmp(f, "420") ? 1 : 0;
        }
}
static VSNode *source_clip(const char *path, int w, int h, int bits, int nframes) {
    VSNode *n = (VSNode *)calloc(1, sizeof(VSNode));
    n->refs = 1;
    n->vi.format.colorFamily = g_fmt_gray ? cfGray : cfYUV; n->vi.format.sampleType = stInteger; n->vi.format.bitsPerSample = bits; n->vi.format.bytesPerSample = bits > 8 ? 2 : 1;
    n->vi.format.subSamplingW = g_fmt_ssw; n->vi.format.subSamplingH = g_fmt_ssh; n->vi.format.numPlanes = g_fmt_gray ? 1 : 3;
    n->vi.fpsNum = 24; n->vi.fpsDen = 1; n->vi.width = w; n->vi.height = h; n->vi.numFrames = nframes;
    n->cache = (const VSFrame **)calloc((size_t)nframes, sizeof(VSFrame *));
    FILE *fp = path ? fopen(path, "rb") : NULL;
    if (path && !fp) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    for (int f = 0; f < nframes; f++) {
        VSFrame *fr = newVideoFrame(&n->vi.format, w, h, NULL, NULL);
        for (int p = 0; p < n->vi.format.numPlanes; p++)
            for (int y = 0; y < plane_h(fr, p); y++) {
                uint8_t *row = fr->data[p] + (size_t)y * fr->stride[p];
                const size_t rb = (size_t)plane_w(fr, p) * n->vi.format.bytesPerSample;
                if (fp) { if (fread(row, 1, rb, fp) != rb) { fprintf(stderr, "short read\n"); exit(2); } }
                else memset(row, 0, rb);
            }
        if (g_field_order >= 0) mapSetInt(fr->props, "_Field", g_field_order ^ (f % 2), maReplace);
        n->cache[f] = fr;
    }
    if (fp) fclose(fp);
    return n;
}

/* x.pelw=W x.pelh=H [x.pelclip=path] [x.pelbits=B]: a second source clip handed to mv.Super as pelclip */
static VSNode *pelclip_from_args(int argc, char **argv, int bits, int nframes) {
    const char *path = NULL;
    int pw = 0, ph = 0;
    for (int i = 0; i < argc; i++) {
        if (!strncmp(argv[i], "x.pelclip=", 10)) path = argv[i] + 10;
        else if (!strncmp(argv[i], "x.pelw=", 7)) pw = atoi(argv[i] + 7);
        else if (!strncmp(argv[i], "x.pelh=", 7)) ph = atoi(argv[i] + 7);
        else if (!strncmp(argv[i], "x.pelbits=", 10)) bits = atoi(argv[i] + 10);
    }
    return pw && ph ? source_clip(path, pw, ph, bits, nframes) : NULL;
}

/* key=value arguments with a one-letter filter prefix ("a.blksize=8") */
static void add_args(VSMap *m, char prefix, int argc, char **argv) {
    for (int i = 0; i < argc; i++) {
        if (argv[i][0] != prefix || argv[i][1] != '.') continue;
        char key[64];
        const char *eq = strchr(argv[i], '=');
        if (!eq) continue;
        snprintf(key, sizeof(key), "%.*s", (int)(eq - argv[i] - 2), argv[i] + 2);
        if (strchr(eq + 1, '.')) mapSetFloat(m, key, atof(eq + 1), maReplace);
        else mapSetInt(m, key, atoll(eq + 1), maReplace);
    }
}
static void dump_frame(FILE *fp, const VSFrame *f) {
    for (int p = 0; p < f->fmt.numPlanes; p++)
        for (int y = 0; y < plane_h(f, p); y++) fwrite(f->data[p] + (size_t)y * f->stride[p], 1, (size_t)plane_w(f, p) * f->fmt.bytesPerSample, fp);
}
static void die(const char *what, const char *err) { printf("ERROR %s: %s\n", what, err); exit(1); }

/* x.threads=N: request frames 0..count-1 of up to four nodes from N threads at once (results stay in the nodes' frame tables) */
typedef struct Work { VSNode *nodes[4]; int nnodes, count, next, done; char err[2048]; } Work;
static double g_phase_t0;
static void *worker(void *arg) {
    Work *w = (Work *)arg;
    for (;;) {
        pthread_mutex_lock(&g_host_mu);
        const int i = w->next < w->count * w->nnodes ? w->next++ : -1;
        pthread_mutex_unlock(&g_host_mu);
        if (i < 0) return NULL;
        char err[2048] = "";
        /* node-major order: every thread first asks the first node, so that one filter instance sees all requests at once */
        const VSFrame *f = eval_frame(i % w->count, w->nodes[i / w->count], err, sizeof(err));
        if (f) freeFrame(f);
        if (getenv("MVX_HOST_TIMES")) { /* progress: when every 64th request completed (the rate of the later part of a run = the steady state) */
            pthread_mutex_lock(&g_host_mu);
            const int done = ++w->done;
            pthread_mutex_unlock(&g_host_mu);
            if (done % 64 == 0) fprintf(stderr, "minihost: %d requests done at %.2f s\n", done, mono_s() - g_phase_t0);
        }
        else { pthread_mutex_lock(&g_host_mu); if (!w->err[0]) snprintf(w->err, sizeof(w->err), "%s", err); pthread_mutex_unlock(&g_host_mu); }
    }
}
static void prefetch_parallel(int threads, int count, VSNode **nodes, int nnodes) {
    if (threads <= 1) return;
    Work w;
    memset(&w, 0, sizeof(w));
    for (int i = 0; i < nnodes; i++) w.nodes[i] = nodes[i];
    w.nnodes = nnodes; w.count = count;
    pthread_t *t = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    g_phase_t0 = mono_s();
    for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, worker, &w);
    for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
    free(t);
    if (w.err[0]) die("parallel request", w.err);
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static double g_start;

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: see minihost.c\n"); return 2; }
    g_start = now_s();
    init_api();
    /* host-side setting: the plugin hands its search launches to a pool of 12 streams, and the HIP runtime maps streams onto
     * GPU_MAX_HW_QUEUES hardware queues (read once, when the runtime initialises).  The host owns its environment, so it is set here */
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    VSInitPlugin init = (VSInitPlugin)dlsym(h, "VapourSynthPluginInit2");
    if (!init) { fprintf(stderr, "VapourSynthPluginInit2 not exported\n"); return 2; }
    VSPLUGINAPI papi = { papiVersion, configPlugin, registerFunction };
    init(NULL, &papi);

    char err[2048] = "";
    if (!strcmp(argv[2], "list")) {
        printf("id=%s ns=%s\n", g_id, g_ns);
        for (int i = 0; i < g_nfuncs; i++) printf("%s %s\n", g_funcs[i].name, g_funcs[i].args);
        return 0;
    }
    if (!strcmp(argv[2], "error")) { /* error <Filter> w h bits [f.key=value] : creation-time behaviour on a blank clip */
        const char *filter = argv[3];
        const int w = atoi(argv[4]), hh = atoi(argv[5]), bits = atoi(argv[6]);
        VSNode *clip = source_clip(NULL, w, hh, bits, 4);
        VSMap *m = createMap();
        VSNode *out = NULL;
        if (!strcmp(filter, "Super")) {
            mapSetNode(m, "clip", clip, maReplace); add_args(m, 'f', argc - 7, argv + 7);
            VSNode *pc = pelclip_from_args(argc - 7, argv + 7, bits, 4);
            if (pc) mapSetNode(m, "pelclip", pc, maReplace);
            out = invoke("Super", m, err, sizeof(err));
        }
        else {
            VSMap *sm = createMap(); mapSetNode(sm, "clip", clip, maReplace); add_args(sm, 's', argc - 7, argv + 7);
            VSNode *sup = invoke("Super", sm, err, sizeof(err));
            if (!sup) die("Super", err);
            if (!strcmp(filter, "Analyse")) { mapSetNode(m, "super", sup, maReplace); add_args(m, 'f', argc - 7, argv + 7); out = invoke("Analyse", m, err, sizeof(err)); }
            else if (!strcmp(filter, "AnalyseOnClip")) { mapSetNode(m, "super", clip, maReplace); out = invoke("Analyse", m, err, sizeof(err)); }
            else { /* DegrainN / Compensate with deliberately swapped or plain vector clips: a.* args go to both analyses */
                VSMap *a1 = createMap(), *a2 = createMap();
                mapSetNode(a1, "super", sup, maReplace); mapSetNode(a2, "super", sup, maReplace);
                add_args(a1, 'a', argc - 7, argv + 7); add_args(a2, 'a', argc - 7, argv + 7);
                mapSetInt(a1, "isb", 1, maReplace); mapSetInt(a2, "isb", 0, maReplace);
                VSNode *bw = invoke("Analyse", a1, err, sizeof(err)); if (!bw) die("Analyse", err);
                VSNode *fw = invoke("Analyse", a2, err, sizeof(err)); if (!fw) die("Analyse", err);
                mapSetNode(m, "clip", clip, maReplace); mapSetNode(m, "super", sup, maReplace);
                add_args(m, 'f', argc - 7, argv + 7);
                if (!strcmp(filter, "Compensate")) { mapSetNode(m, "vectors", bw, maReplace); out = invoke("Compensate", m, err, sizeof(err)); }
                else if (!strcmp(filter, "Degrain1Swapped")) { mapSetNode(m, "mvbw", fw, maReplace); mapSetNode(m, "mvfw", bw, maReplace); out = invoke("Degrain1", m, err, sizeof(err)); }
                else { mapSetNode(m, "mvbw", bw, maReplace); mapSetNode(m, "mvfw", fw, maReplace); out = invoke(filter, m, err, sizeof(err)); }
            }
        }
        if (out) printf("OK %dx%d frames=%d\n", out->vi.width, out->vi.height, out->vi.numFrames);
        else printf("ERROR %s\n", err);
        return 0;
    }
    if (strcmp(argv[2], "run") || argc < 10) { fprintf(stderr, "bad command\n"); return 2; }
    const char *pipeline = argv[3], *inPath = argv[4], *outPath = argv[9];
    const int w = atoi(argv[5]), hh = atoi(argv[6]), bits = atoi(argv[7]), nframes = atoi(argv[8]);
    char **extra = argv + 10; const int nextra = argc - 10;
    int threads = 1;
    format_from_args(nextra, extra);
    for (int i = 0; i < nextra; i++) if (!strncmp(extra[i], "x.fieldorder=", 13)) g_field_order = atoi(extra[i] + 13);
    for (int i = 0; i < nextra; i++) if (!strncmp(extra[i], "x.threads=", 10)) threads = atoi(extra[i] + 10);
    for (int i = 0; i < nextra; i++) if (!strncmp(extra[i], "x.cache=", 8)) g_cache_limit = atoi(extra[i] + 8);
    VSNode *clip = source_clip(inPath, w, hh, bits, nframes);
    FILE *fo = fopen(outPath, "wb"); /* (before the time mark: truncating a previous run's multi-gigabyte result file takes a second) */
    if (!fo) { fprintf(stderr, "cannot write %s\n", outPath); return 2; }
    if (getenv("MVX_HOST_TIMES")) fprintf(stderr, "minihost: clip loaded at %.2f s after start\n", now_s() - g_start); /* (graph construction starts here) */

    VSMap *sm = createMap(); mapSetNode(sm, "clip", clip, maReplace); add_args(sm, 's', nextra, extra);
    VSNode *pelclip = pelclip_from_args(nextra, extra, bits, nframes);
    if (pelclip) mapSetNode(sm, "pelclip", pelclip, maReplace);
    VSNode *sup = invoke("Super", sm, err, sizeof(err));
    if (!sup) die("Super", err);
    if (getenv("MVX_HOST_TIMES")) fprintf(stderr, "minihost: mv.Super created at %.2f s after start\n", now_s() - g_start);
    if (!strcmp(pipeline, "super")) {
        for (int n = 0; n < nframes; n++) {
            const VSFrame *f = eval_frame(n, sup, err, sizeof(err));
            if (!f) die("Super frame", err);
            if (n == 0) {
                int e;
                printf("super %dx%d", f->w, f->h);
                const char *keys[] = { "Super_height", "Super_hpad", "Super_vpad", "Super_pel", "Super_modeyuv", "Super_levels" };
                for (int k = 0; k < 6; k++) printf(" %s=%lld", keys[k], (long long)mapGetInt(f->props, keys[k], 0, &e));
                printf("\n");
            }
            dump_frame(fo, f); freeFrame(f);
        }
        fclose(fo); printf("DONE\n"); return 0;
    }
    /* vector clips: delta 1..R, backward then forward */
    int R = 1;
    if (!strncmp(pipeline, "degrain", 7)) R = atoi(pipeline + 7);
    VSNode *vec[12];
    VSMap *amaps[12] = { NULL };
    for (int r = 0; r < R; r++)
        for (int isb = 1; isb >= 0; isb--) {
            VSMap *am = createMap(); mapSetNode(am, "super", sup, maReplace); add_args(am, 'a', nextra, extra);
            mapSetInt(am, "isb", isb, maReplace); mapSetInt(am, "delta", r + 1, maReplace);
            amaps[2 * r + (isb ? 0 : 1)] = am;
            vec[2 * r + (isb ? 0 : 1)] = invoke("Analyse", am, err, sizeof(err));
            if (getenv("MVX_HOST_TIMES")) fprintf(stderr, "minihost: mv.Analyse delta %d isb %d created at %.2f s after start\n", r + 1, isb, now_s() - g_start);
            if (!vec[2 * r + (isb ? 0 : 1)]) die("Analyse", err);
        }
    if (!strcmp(pipeline, "analyse")) {
        prefetch_parallel(threads, nframes, vec, 2);
        for (int n = 0; n < nframes; n++)
            for (int k = 0; k < 2; k++) {
                const VSFrame *f = eval_frame(n, vec[k], err, sizeof(err));
                if (!f) die("Analyse frame", err);
                int e;
                fwrite(mapGetData(f->props, "MVTools_MVAnalysisData", 0, &e), 1, (size_t)mapGetDataSize(f->props, "MVTools_MVAnalysisData", 0, &e), fo);
                fwrite(mapGetData(f->props, "MVTools_vectors", 0, &e), 1, (size_t)mapGetDataSize(f->props, "MVTools_vectors", 0, &e), fo);
                freeFrame(f);
            }
        fclose(fo); printf("DONE\n"); return 0;
    }
    VSMap *m = createMap();
    mapSetNode(m, "clip", clip, maReplace); mapSetNode(m, "super", sup, maReplace);
    VSNode *out;
    if (!strcmp(pipeline, "finest")) {
        VSMap *fm = createMap(); mapSetNode(fm, "super", sup, maReplace);
        VSNode *fin = invoke("Finest", fm, err, sizeof(err));
        if (!fin) die("Finest", err);
        printf("finest %dx%d\n", fin->vi.width, fin->vi.height);
        for (int n = 0; n < nframes; n++) { const VSFrame *f = eval_frame(n, fin, err, sizeof(err)); if (!f) die("Finest frame", err); dump_frame(fo, f); freeFrame(f); }
        fclose(fo); printf("DONE\n"); return 0;
    }
    if (!strcmp(pipeline, "scdetection")) { /* prints the scene-change props per frame for the backward and the forward vectors */
        for (int k = 0; k < 2; k++) {
            VSMap *sm2 = createMap(); mapSetNode(sm2, "clip", clip, maReplace); mapSetNode(sm2, "vectors", vec[k], maReplace); add_args(sm2, 'd', nextra, extra);
            VSNode *sc = invoke("SCDetection", sm2, err, sizeof(err));
            if (!sc) die("SCDetection", err);
            for (int n = 0; n < nframes; n++) {
                const VSFrame *f = eval_frame(n, sc, err, sizeof(err));
                if (!f) die("SCDetection frame", err);
                int e1, e2;
                const long long nx = (long long)mapGetInt(f->props, "_SceneChangeNext", 0, &e1), pv = (long long)mapGetInt(f->props, "_SceneChangePrev", 0, &e2);
                printf("%s frame %d next=%lld prev=%lld\n", k ? "fw" : "bw", n, e1 ? -1 : nx, e2 ? -1 : pv);
                if (k == 1) dump_frame(fo, f);
                freeFrame(f);
            }
        }
        fclose(fo); printf("DONE\n"); return 0;
    }
    if (!strcmp(pipeline, "recalculate")) { /* analyse (a.*) -> recalculate (r.*), backward then forward; dumps both props per frame */
        for (int k = 0; k < 2; k++) {
            VSMap *rm = createMap(); mapSetNode(rm, "super", sup, maReplace); mapSetNode(rm, "vectors", vec[k], maReplace); add_args(rm, 'r', nextra, extra);
            VSNode *rc = invoke("Recalculate", rm, err, sizeof(err));
            if (!rc) die("Recalculate", err);
            vec[2 + k] = rc;
        }
        for (int n = 0; n < nframes; n++)
            for (int k = 0; k < 2; k++) {
                const VSFrame *f = eval_frame(n, vec[2 + k], err, sizeof(err));
                if (!f) die("Recalculate frame", err);
                int e;
                fwrite(mapGetData(f->props, "MVTools_MVAnalysisData", 0, &e), 1, (size_t)mapGetDataSize(f->props, "MVTools_MVAnalysisData", 0, &e), fo);
                fwrite(mapGetData(f->props, "MVTools_vectors", 0, &e), 1, (size_t)mapGetDataSize(f->props, "MVTools_vectors", 0, &e), fo);
                freeFrame(f);
            }
        fclose(fo); printf("DONE\n"); return 0;
    }
    if (!strcmp(pipeline, "blockfps")) {
        mapSetNode(m, "mvbw", vec[0], maReplace); mapSetNode(m, "mvfw", vec[1], maReplace); add_args(m, 'b', nextra, extra);
        out = invoke("BlockFPS", m, err, sizeof(err));
        if (!out) die(pipeline, err);
        printf("blockfps frames=%d fps=%lld/%lld\n", out->vi.numFrames, (long long)out->vi.fpsNum, (long long)out->vi.fpsDen);
        for (int n = 0; n < out->vi.numFrames; n++) {
            const VSFrame *f = eval_frame(n, out, err, sizeof(err));
            if (!f) die("output frame", err);
            if (n == 1) { int e; printf("frame1 _DurationNum=%lld _DurationDen=%lld\n", (long long)mapGetInt(f->props, "_DurationNum", 0, &e), (long long)mapGetInt(f->props, "_DurationDen", 0, &e)); }
            dump_frame(fo, f); freeFrame(f);
        }
        fclose(fo); printf("DONE\n"); return 0;
    }
    if (!strcmp(pipeline, "compensate")) { mapSetNode(m, "vectors", vec[0], maReplace); add_args(m, 'c', nextra, extra); out = invoke("Compensate", m, err, sizeof(err)); }
    else {
        static const char *vn[] = { "mvbw", "mvfw", "mvbw2", "mvfw2", "mvbw3", "mvfw3", "mvbw4", "mvfw4", "mvbw5", "mvfw5", "mvbw6", "mvfw6" };
        for (int r = 0; r < 2 * R; r++) mapSetNode(m, vn[r], vec[r], maReplace);
        add_args(m, 'd', nextra, extra);
        char fn[16]; snprintf(fn, sizeof(fn), "Degrain%d", R);
        out = invoke(fn, m, err, sizeof(err));
    }
    if (!out) die(pipeline, err);
    const int times = getenv("MVX_HOST_TIMES") != NULL;
    double t0 = now_s();
    if (times) fprintf(stderr, "minihost: graph built at %.2f s after start\n", t0 - g_start);
    int frameOrder = 0; /* x.order=frame: the threads ask for OUTPUT frames only, as a client of a real core does (every upstream request is then made by the filters) */
    for (int i = 0; i < nextra; i++) if (!strcmp(extra[i], "x.order=frame")) frameOrder = 1;
    if (frameOrder) g_uncapped = out;
    if (threads > 1 && frameOrder) {
        prefetch_parallel(threads, nframes, &out, 1);
        if (times) { fprintf(stderr, "minihost: output clip (frame order) %.2f s\n", now_s() - t0); t0 = now_s(); }
    } else if (threads > 1) { /* the vector clips first (all threads inside one Analyse instance at a time), then the output */
        prefetch_parallel(threads, nframes, vec, 2 * R < 4 ? 2 * R : 4);
        if (2 * R > 4) prefetch_parallel(threads, nframes, vec + 4, 2 * R - 4 < 4 ? 2 * R - 4 : 4);
        if (2 * R > 8) prefetch_parallel(threads, nframes, vec + 8, 2 * R - 8);
        if (times) { fprintf(stderr, "minihost: vector clips %.2f s\n", now_s() - t0); t0 = now_s(); }
        prefetch_parallel(threads, nframes, &out, 1);
        if (times) { fprintf(stderr, "minihost: output clip %.2f s\n", now_s() - t0); t0 = now_s(); }
    }
    for (int n = 0; n < nframes; n++) {
        const VSFrame *f = eval_frame(n, out, err, sizeof(err));
        if (!f) die("output frame", err);
        dump_frame(fo, f); freeFrame(f);
    }
    fclose(fo);
    if (times) fprintf(stderr, "minihost: result file written in %.2f s; threads waited %.2f thread-seconds for frames other threads were producing\n", now_s() - t0, g_wait_s);
    for (int i = 0; i < nextra; i++)
        if (!strcmp(extra[i], "x.free=1")) { /* tear the graph down like a core that is being freed: every filter's free callback runs, consumers before producers */
            freeMap(m); freeMap(sm);
            for (int r = 0; r < 2 * R; r++) freeMap(amaps[r]);
            node_free(out);
            for (int r = 0; r < 2 * R; r++) node_free(vec[r]);
            node_free(sup);
            if (pelclip) node_free(pelclip);
            node_free(clip);
            printf("FREED\n");
        }
    printf("DONE\n");
    return 0;
}

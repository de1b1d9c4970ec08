/*
 * mvx_fakedev.c -- TEST DOUBLE of the device layer.  TEST INFRASTRUCTURE ONLY: built by tests/test_vs_shell_cpu.py, loaded with
 * LD_PRELOAD in front of libmvtools_amd.so for that test's mini-host process, never built by build(), never installed, never on any
 * product path (the library and the plugin fail loudly without a HIP device, tests/test_host.py::test_no_cpu_fallback).
 *
 * Why it exists: the VapourSynth filter shell (vsplugin/mvtools_vs.c) is host logic -- request protocol, look-ahead windows, the cache of
 * device frames, pinning, eviction, thread synchronisation -- and the only way to run it used to be a GPU.  With this file in front of the
 * library the shell's calls that would touch the device land in host memory ("device" buffers are malloc'ed, copies are memcpy, streams
 * are tokens) and the kernel calls (Super incl. pelclip, Finest, Analyse, Recalculate, Degrain, Compensate, BlockFPS, SCDetection) are the ORACLE's functions (oracle/mvoracle.h), so a graph
 * evaluated through the real plugin and the real mini host on a CPU-only machine must reproduce the oracle bit for bit, whatever the
 * thread count and the look-ahead configuration.  What it tests is the shell; it says nothing about the HIP kernels (the -m gpu suite does).
 *
 * Everything that does not touch the device (creation, argument resolution, geometry, blob validation, error strings) still runs in the
 * real library: the create functions here call the real ones (dlsym RTLD_NEXT) and only remember the arguments.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvtools_amd.h"
#include "mvoracle.h"

#define API __attribute__((visibility("default")))

/* ---- what the oracle needs to know about the handles the real library made */
typedef struct Rec {
    const void *handle;
    int kind; /* 1 super, 2 analyse, 3 degrain, 4 compensate, 5 recalculate, 6 blockfps */
    mvo_super s; /* (kinds > 1: a copy of the super clip's geometry) */
    mvo_analyse an;
    mvo_degrain dg;
    mvo_compensate cp;
    mvo_recalculate rc;
    mvo_blockfps bf;
    int superPitch[3], srcPitch[3], dstPitch[3];
    struct Rec *next;
} Rec;
static Rec *g_recs;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static long g_launches, g_jobs; /* search launches / jobs seen (printed at exit with MVX_FAKEDEV_STATS=1) */

static Rec *rec_find(const void *h, int kind) {
    Rec *r;
    pthread_mutex_lock(&g_mu);
    for (r = g_recs; r; r = r->next) if (r->handle == h && r->kind == kind) break;
    pthread_mutex_unlock(&g_mu);
    if (!r) { fprintf(stderr, "mvx_fakedev: unknown handle %p (kind %d)\n", h, kind); abort(); }
    return r;
}
static Rec *rec_new(const void *h, int kind) {
    Rec *r = (Rec *)calloc(1, sizeof(Rec));
    if (!r) abort();
    r->handle = h; r->kind = kind;
    pthread_mutex_lock(&g_mu);
    r->next = g_recs; g_recs = r;
    pthread_mutex_unlock(&g_mu);
    return r;
}
static void rec_drop(const void *h, int kind) { /* handles are reused by the allocator: forget destroyed ones */
    pthread_mutex_lock(&g_mu);
    for (Rec **pp = &g_recs; *pp; pp = &(*pp)->next)
        if ((*pp)->handle == h && (*pp)->kind == kind) { Rec *d = *pp; *pp = d->next; free(d); break; }
    pthread_mutex_unlock(&g_mu);
}
static void *real(const char *name) { /* the library's own function of that name (the plugin's dependency, loaded RTLD_GLOBAL by the mini host) */
    void *f = dlsym(RTLD_NEXT, name);
    if (!f) {
        void *h = dlopen("libmvtools_amd.so", RTLD_NOLOAD | RTLD_NOW);
        if (h) f = dlsym(h, name);
    }
    if (!f) { fprintf(stderr, "mvx_fakedev: %s not found behind the test double\n", name); abort(); }
    return f;
}
__attribute__((destructor)) static void fake_stats(void) {
    if (getenv("MVX_FAKEDEV_STATS")) fprintf(stderr, "mvx_fakedev: search launches=%ld jobs=%ld\n", g_launches, g_jobs);
}

/* ---- "device" memory, copies, streams */
API void *mvx_dev_alloc(size_t bytes) { return calloc(1, bytes ? bytes : 1); }
API void *mvx_dev_alloc_uninit(size_t bytes) {
    unsigned char *p = (unsigned char *)malloc(bytes ? bytes : 1);
    if (p) memset(p, 0xA5, bytes); /* "contents undefined": make a read of stale bytes visible */
    return p;
}
API void mvx_dev_free(void *p) { free(p); }
API void mvx_dev_pool_limit(size_t bytes) { (void)bytes; }
API int mvx_warmup(int staging_buffers) { (void)staging_buffers; return MVX_OK; } /* (the plugin's background warm-up at load: nothing to warm here) */
API void mvx_dev_pool_trim(void) {}
API int mvx_dev_mem_info(size_t *free_bytes, size_t *total_bytes) {
    /* small on purpose (MVX_FAKEDEV_MEM, default 64 MiB "free"): the shell sizes its frame cache from this, so eviction really happens */
    const char *e = getenv("MVX_FAKEDEV_MEM");
    const size_t f = e ? (size_t)atoll(e) : (size_t)64 << 20;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = 2 * f;
    return MVX_OK;
}
API void *mvx_stream_create(void) { return malloc(1); }
API void *mvx_stream_create_priority(int level) { (void)level; return malloc(1); }
API void mvx_stream_destroy(void *stream) { free(stream); }
API int mvx_stream_sync(void *stream) { (void)stream; return MVX_OK; }
static int copy2d(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows) {
    for (size_t y = 0; y < rows; y++) memcpy((char *)dst + (ptrdiff_t)y * dp, (const char *)src + (ptrdiff_t)y * sp, rb);
    return MVX_OK;
}
API int mvx_upload_2d(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows, void *st) { (void)st; return copy2d(dst, dp, src, sp, rb, rows); }
API int mvx_download_2d(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows, void *st) { (void)st; return copy2d(dst, dp, src, sp, rb, rows); }
API int mvx_copy_to_device(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows, void *st) { (void)st; return copy2d(dst, dp, src, sp, rb, rows); }
API int mvx_copy_to_host(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t rb, size_t rows, void *st) { (void)st; return copy2d(dst, dp, src, sp, rb, rows); }
API int mvx_dev_memset(void *dst, int value, size_t bytes, void *st) { (void)st; memset(dst, value, bytes); return MVX_OK; }

/* ---- mv.Super */
API int mvx_super_create(const mvx_super_args *a, mvx_super **out, char *err) {
    int (*f)(const mvx_super_args *, mvx_super **, char *) = (int (*)(const mvx_super_args *, mvx_super **, char *))real("mvx_super_create");
    const int rc = f(a, out, err);
    if (rc == MVX_OK) {
        Rec *r = rec_new(*out, 1);
        char e2[MVO_ERR];
        if (mvo_super_init(&r->s, a->width, a->height, a->bits, a->subsampling_w, a->subsampling_h, a->gray, a->hpad, a->vpad, a->pel, a->levels, a->chroma,
                           a->sharp, a->rfilter, e2)) { fprintf(stderr, "mvx_fakedev: the oracle refuses what the library accepted: %s\n", e2); abort(); }
    }
    return rc;
}
API void mvx_super_destroy(mvx_super *s) {
    void (*f)(mvx_super *) = (void (*)(mvx_super *))real("mvx_super_destroy");
    if (s) rec_drop(s, 1);
    f(s);
}
static int super_frames(mvx_super *s, int n, const void *const *src, const ptrdiff_t sp[3], void *const *dst, const ptrdiff_t dp[3]) {
    const Rec *r = rec_find(s, 1);
    const int spi[3] = { (int)sp[0], (int)sp[1], (int)sp[2] }, dpi[3] = { (int)dp[0], (int)dp[1], (int)dp[2] };
    for (int f = 0; f < n; f++) {
        const uint8_t *sv[3] = { (const uint8_t *)src[f * 3], (const uint8_t *)src[f * 3 + 1], (const uint8_t *)src[f * 3 + 2] };
        uint8_t *dv[3] = { (uint8_t *)dst[f * 3], (uint8_t *)dst[f * 3 + 1], (uint8_t *)dst[f * 3 + 2] };
        mvo_super_frame(&r->s, sv, spi, dv, dpi);
    }
    return MVX_OK;
}
API int mvx_super_frames(mvx_super *s, int n, const void *const *src, const ptrdiff_t sp[3], void *const *dst, const ptrdiff_t dp[3], void *st) {
    (void)st;
    return super_frames(s, n, src, sp, dst, dp);
}
API int mvx_super_frames_shadow(mvx_super *s, int n, const void *const *src, const ptrdiff_t sp[3], void *const *dst, const ptrdiff_t dp[3],
                                const ptrdiff_t shadow[3], void *st) {
    (void)st; (void)shadow; /* the shadow planes are a device-side layout extension: nothing here reads them */
    return super_frames(s, n, src, sp, dst, dp);
}
API int mvx_super_shadow_frames(const mvx_super *s, int n, void *const *planes, const ptrdiff_t pitch[3], const ptrdiff_t stride[3], void *st) {
    (void)s; (void)n; (void)planes; (void)pitch; (void)stride; (void)st;
    return MVX_OK;
}

/* ---- mv.Analyse */
API int mvx_analyse_create(const mvx_analyse_args *a, const mvx_super *sup, int num_frames, const ptrdiff_t super_pitch[3], mvx_analyse **out, char *err) {
    int (*f)(const mvx_analyse_args *, const mvx_super *, int, const ptrdiff_t *, mvx_analyse **, char *) =
        (int (*)(const mvx_analyse_args *, const mvx_super *, int, const ptrdiff_t *, mvx_analyse **, char *))real("mvx_analyse_create");
    const int rc = f(a, sup, num_frames, super_pitch, out, err);
    if (rc == MVX_OK) {
        const Rec *rs = rec_find(sup, 1);
        Rec *r = rec_new(*out, 2);
        mvo_analyse_args oa;
        _Static_assert(sizeof(mvo_analyse_args) == sizeof(mvx_analyse_args), "the two argument blocks list the same 29 integers in the same order");
        memcpy(&oa, a, sizeof(oa));
        char e2[MVO_ERR];
        if (mvo_analyse_init(&r->an, &oa, &rs->s, num_frames, e2)) { fprintf(stderr, "mvx_fakedev: the oracle refuses what the library accepted: %s\n", e2); abort(); }
        for (int p = 0; p < 3; p++) r->superPitch[p] = (int)super_pitch[p];
        if (mvo_analyse_blob_size(&r->an) != ((int (*)(const mvx_analyse *))real("mvx_analyse_blob_size"))(*out)) { fprintf(stderr, "mvx_fakedev: blob sizes differ\n"); abort(); }
    }
    return rc;
}
API void mvx_analyse_destroy(mvx_analyse *a) {
    void (*f)(mvx_analyse *) = (void (*)(mvx_analyse *))real("mvx_analyse_destroy");
    if (a) rec_drop(a, 2);
    f(a);
}
API int mvx_analyse_frames(mvx_analyse *a, int njobs, const mvx_analyse_job *jobs, void *st) {
    (void)st;
    const Rec *r = rec_find(a, 2);
    pthread_mutex_lock(&g_mu); g_launches++; g_jobs += njobs; pthread_mutex_unlock(&g_mu);
    for (int i = 0; i < njobs; i++) {
        const uint8_t *src[3] = { (const uint8_t *)jobs[i].src[0], (const uint8_t *)jobs[i].src[1], (const uint8_t *)jobs[i].src[2] };
        const uint8_t *ref[3] = { (const uint8_t *)jobs[i].ref[0], (const uint8_t *)jobs[i].ref[1], (const uint8_t *)jobs[i].ref[2] };
        mvo_analyse_frame(&r->an, src, r->superPitch, jobs[i].ref[0] ? ref : NULL, r->superPitch, jobs[i].field_shift, (uint8_t *)jobs[i].blob);
    }
    return MVX_OK;
}

/* ---- mv.DegrainN */
API int mvx_degrain_create(const mvx_degrain_args *a, const mvx_analysis_data *vd, const mvx_super *sup, const ptrdiff_t src_pitch[3],
                           const ptrdiff_t super_pitch[3], const ptrdiff_t dst_pitch[3], mvx_degrain **out, char *err) {
    int (*f)(const mvx_degrain_args *, const mvx_analysis_data *, const mvx_super *, const ptrdiff_t *, const ptrdiff_t *, const ptrdiff_t *, mvx_degrain **, char *) =
        (int (*)(const mvx_degrain_args *, const mvx_analysis_data *, const mvx_super *, const ptrdiff_t *, const ptrdiff_t *, const ptrdiff_t *, mvx_degrain **, char *))
            real("mvx_degrain_create");
    const int rc = f(a, vd, sup, src_pitch, super_pitch, dst_pitch, out, err);
    if (rc == MVX_OK) {
        const Rec *rs = rec_find(sup, 1);
        Rec *r = rec_new(*out, 3);
        mvo_analysis_data ad;
        _Static_assert(sizeof(mvo_analysis_data) == sizeof(mvx_analysis_data), "both mirror MVTools_MVAnalysisData");
        memcpy(&ad, vd, sizeof(ad));
        char e2[MVO_ERR];
        if (mvo_degrain_init(&r->dg, a->radius, &ad, &rs->s, a->thsad, a->thsadc, a->plane, a->limit, a->limitc, a->thscd1, a->thscd2, e2)) {
            fprintf(stderr, "mvx_fakedev: the oracle refuses what the library accepted: %s\n", e2); abort();
        }
        for (int p = 0; p < 3; p++) { r->superPitch[p] = (int)super_pitch[p]; r->srcPitch[p] = (int)src_pitch[p]; r->dstPitch[p] = (int)dst_pitch[p]; }
    }
    return rc;
}
API void mvx_degrain_destroy(mvx_degrain *d) {
    void (*f)(mvx_degrain *) = (void (*)(mvx_degrain *))real("mvx_degrain_destroy");
    if (d) rec_drop(d, 3);
    f(d);
}
API int mvx_degrain_frames(mvx_degrain *d, int nframes, const mvx_degrain_job *jobs, void *st) {
    (void)st;
    const Rec *r = rec_find(d, 3);
    const int nr = 2 * r->dg.radius;
    for (int i = 0; i < nframes; i++) {
        const uint8_t *src[3] = { (const uint8_t *)jobs[i].src[0], (const uint8_t *)jobs[i].src[1], (const uint8_t *)jobs[i].src[2] };
        uint8_t *dst[3] = { (uint8_t *)jobs[i].dst[0], (uint8_t *)jobs[i].dst[1], (uint8_t *)jobs[i].dst[2] };
        const uint8_t *refs[12][3];
        int pitches[12][3];
        const uint8_t *blobs[12];
        for (int k = 0; k < nr; k++) {
            for (int p = 0; p < 3; p++) { refs[k][p] = (const uint8_t *)jobs[i].refs[k][p]; pitches[k][p] = r->superPitch[p]; }
            blobs[k] = (const uint8_t *)jobs[i].blobs[k];
        }
        mvo_degrain_frame(&r->dg, src, r->srcPitch, (const uint8_t *const (*)[3])refs, (const int (*)[3])pitches, blobs, dst, r->dstPitch);
    }
    return MVX_OK;
}

/* ---- mv.Super(pelclip=...), mv.Finest */
API int mvx_super_frames_pelclip(mvx_super *s, int n, const void *const *src, const ptrdiff_t sp[3], const void *const *pel, const ptrdiff_t pp[3], int mode,
                                 void *const *dst, const ptrdiff_t dp[3], void *st) {
    (void)st;
    const Rec *r = rec_find(s, 1);
    const int spi[3] = { (int)sp[0], (int)sp[1], (int)sp[2] }, ppi[3] = { (int)pp[0], (int)pp[1], (int)pp[2] }, dpi[3] = { (int)dp[0], (int)dp[1], (int)dp[2] };
    for (int f = 0; f < n; f++) {
        const uint8_t *sv[3] = { (const uint8_t *)src[f * 3], (const uint8_t *)src[f * 3 + 1], (const uint8_t *)src[f * 3 + 2] };
        const uint8_t *pv[3] = { (const uint8_t *)pel[f * 3], (const uint8_t *)pel[f * 3 + 1], (const uint8_t *)pel[f * 3 + 2] };
        uint8_t *dv[3] = { (uint8_t *)dst[f * 3], (uint8_t *)dst[f * 3 + 1], (uint8_t *)dst[f * 3 + 2] };
        mvo_super_frame_pelclip(&r->s, sv, spi, pv, ppi, mode, dv, dpi);
    }
    return MVX_OK;
}
API int mvx_finest_frames(const mvx_super *s, int n, const void *const *sup, const ptrdiff_t sp[3], void *const *dst, const ptrdiff_t dp[3], void *st) {
    (void)st;
    const Rec *r = rec_find(s, 1);
    const int spi[3] = { (int)sp[0], (int)sp[1], (int)sp[2] }, dpi[3] = { (int)dp[0], (int)dp[1], (int)dp[2] };
    for (int f = 0; f < n; f++) {
        const uint8_t *sv[3] = { (const uint8_t *)sup[f * 3], (const uint8_t *)sup[f * 3 + 1], (const uint8_t *)sup[f * 3 + 2] };
        uint8_t *dv[3] = { (uint8_t *)dst[f * 3], (uint8_t *)dst[f * 3 + 1], (uint8_t *)dst[f * 3 + 2] };
        mvo_finest_frame(&r->s, sv, spi, dv, dpi);
    }
    return MVX_OK;
}

/* ---- mv.Compensate */
API int mvx_compensate_create(const mvx_compensate_args *a, const mvx_analysis_data *vd, const mvx_super *sup, const ptrdiff_t super_pitch[3],
                              const ptrdiff_t dst_pitch[3], mvx_compensate **out, char *err) {
    int (*f)(const mvx_compensate_args *, const mvx_analysis_data *, const mvx_super *, const ptrdiff_t *, const ptrdiff_t *, mvx_compensate **, char *) =
        (int (*)(const mvx_compensate_args *, const mvx_analysis_data *, const mvx_super *, const ptrdiff_t *, const ptrdiff_t *, mvx_compensate **, char *))real("mvx_compensate_create");
    const int rc = f(a, vd, sup, super_pitch, dst_pitch, out, err);
    if (rc == MVX_OK) {
        const Rec *rs = rec_find(sup, 1);
        Rec *r = rec_new(*out, 4);
        mvo_analysis_data ad;
        memcpy(&ad, vd, sizeof(ad));
        char e2[MVO_ERR];
        if (mvo_compensate_init(&r->cp, &ad, &rs->s, a->scbehavior, a->thsad, a->time, a->thscd1, a->thscd2, e2)) {
            fprintf(stderr, "mvx_fakedev: the oracle refuses what the library accepted: %s\n", e2); abort();
        }
        for (int p = 0; p < 3; p++) { r->superPitch[p] = (int)super_pitch[p]; r->dstPitch[p] = (int)dst_pitch[p]; }
    }
    return rc;
}
API void mvx_compensate_destroy(mvx_compensate *c) {
    void (*f)(mvx_compensate *) = (void (*)(mvx_compensate *))real("mvx_compensate_destroy");
    if (c) rec_drop(c, 4);
    f(c);
}
API int mvx_compensate_frames(mvx_compensate *c, int n, const mvx_compensate_job *jobs, void *st) {
    (void)st;
    const Rec *r = rec_find(c, 4);
    for (int i = 0; i < n; i++) {
        const uint8_t *src[3] = { (const uint8_t *)jobs[i].src_super[0], (const uint8_t *)jobs[i].src_super[1], (const uint8_t *)jobs[i].src_super[2] };
        const uint8_t *ref[3] = { (const uint8_t *)jobs[i].ref_super[0], (const uint8_t *)jobs[i].ref_super[1], (const uint8_t *)jobs[i].ref_super[2] };
        uint8_t *dst[3] = { (uint8_t *)jobs[i].dst[0], (uint8_t *)jobs[i].dst[1], (uint8_t *)jobs[i].dst[2] };
        mvo_compensate_frame(&r->cp, src, r->superPitch, jobs[i].ref_super[0] ? ref : NULL, r->superPitch, (const uint8_t *)jobs[i].blob, dst, r->dstPitch, jobs[i].field_shift);
    }
    return MVX_OK;
}

/* ---- mv.Recalculate */
API int mvx_recalculate_create(const mvx_recalculate_args *a, const mvx_super *sup, const mvx_analysis_data *vd, const ptrdiff_t super_pitch[3], mvx_recalculate **out, char *err) {
    int (*f)(const mvx_recalculate_args *, const mvx_super *, const mvx_analysis_data *, const ptrdiff_t *, mvx_recalculate **, char *) =
        (int (*)(const mvx_recalculate_args *, const mvx_super *, const mvx_analysis_data *, const ptrdiff_t *, mvx_recalculate **, char *))real("mvx_recalculate_create");
    const int rc = f(a, sup, vd, super_pitch, out, err);
    if (rc == MVX_OK) {
        const Rec *rs = rec_find(sup, 1);
        Rec *r = rec_new(*out, 5);
        mvo_recalculate_args oa = { a->thsad, a->smooth, a->blksize, a->blksizev, a->search, a->searchparam, a->lambda, a->chroma, a->truemotion, a->pnew, a->overlap,
                                    a->overlapv, a->divide, a->meander, a->dct }; /* (`fields` only shifts the job's reference rows; the shell computes that) */
        mvo_analysis_data ad;
        memcpy(&ad, vd, sizeof(ad));
        char e2[MVO_ERR];
        if (mvo_recalculate_init(&r->rc, &oa, &rs->s, &ad, e2)) { fprintf(stderr, "mvx_fakedev: the oracle refuses what the library accepted: %s\n", e2); abort(); }
        for (int p = 0; p < 3; p++) r->superPitch[p] = (int)super_pitch[p];
    }
    return rc;
}
API void mvx_recalculate_destroy(mvx_recalculate *x) {
    void (*f)(mvx_recalculate *) = (void (*)(mvx_recalculate *))real("mvx_recalculate_destroy");
    if (x) rec_drop(x, 5);
    f(x);
}
API int mvx_recalculate_frames(mvx_recalculate *x, int n, const mvx_recalculate_job *jobs, void *st) {
    (void)st;
    const Rec *r = rec_find(x, 5);
    for (int i = 0; i < n; i++) {
        const uint8_t *src[3] = { (const uint8_t *)jobs[i].src[0], (const uint8_t *)jobs[i].src[1], (const uint8_t *)jobs[i].src[2] };
        const uint8_t *ref[3] = { (const uint8_t *)jobs[i].ref[0], (const uint8_t *)jobs[i].ref[1], (const uint8_t *)jobs[i].ref[2] };
        mvo_recalculate_frame(&r->rc, src, r->superPitch, jobs[i].ref[0] ? ref : NULL, r->superPitch, (const uint8_t *)jobs[i].old_blob, (uint8_t *)jobs[i].blob);
    }
    return MVX_OK;
}

/* ---- mv.BlockFPS */
API int mvx_blockfps_create(const mvx_blockfps_args *a, const mvx_analysis_data *bw, const mvx_analysis_data *fw, const mvx_super *sup, int num_frames, int64_t fps_num,
                            int64_t fps_den, const ptrdiff_t super_pitch[3], const ptrdiff_t clip_pitch[3], const ptrdiff_t dst_pitch[3], mvx_blockfps **out, char *err) {
    int (*f)(const mvx_blockfps_args *, const mvx_analysis_data *, const mvx_analysis_data *, const mvx_super *, int, int64_t, int64_t, const ptrdiff_t *, const ptrdiff_t *,
             const ptrdiff_t *, mvx_blockfps **, char *) =
        (int (*)(const mvx_blockfps_args *, const mvx_analysis_data *, const mvx_analysis_data *, const mvx_super *, int, int64_t, int64_t, const ptrdiff_t *, const ptrdiff_t *,
                 const ptrdiff_t *, mvx_blockfps **, char *))real("mvx_blockfps_create");
    const int rc = f(a, bw, fw, sup, num_frames, fps_num, fps_den, super_pitch, clip_pitch, dst_pitch, out, err);
    if (rc == MVX_OK) {
        const Rec *rs = rec_find(sup, 1);
        Rec *r = rec_new(*out, 6);
        r->s = rs->s;
        mvo_analysis_data b, w;
        memcpy(&b, bw, sizeof(b)); memcpy(&w, fw, sizeof(w));
        char e2[MVO_ERR];
        if (mvo_blockfps_init(&r->bf, &b, &w, &rs->s, num_frames, fps_num, fps_den, a->num, a->den, a->mode, a->ml, a->blend, a->thscd1, a->thscd2, e2)) {
            fprintf(stderr, "mvx_fakedev: the oracle refuses what the library accepted: %s\n", e2); abort();
        }
        for (int p = 0; p < 3; p++) { r->superPitch[p] = (int)super_pitch[p]; r->srcPitch[p] = (int)clip_pitch[p]; r->dstPitch[p] = (int)dst_pitch[p]; }
    }
    return rc;
}
API void mvx_blockfps_destroy(mvx_blockfps *b) {
    void (*f)(mvx_blockfps *) = (void (*)(mvx_blockfps *))real("mvx_blockfps_destroy");
    if (b) rec_drop(b, 6);
    f(b);
}
API int mvx_blockfps_frames(mvx_blockfps *b, int n, const mvx_blockfps_job *jobs, void *st) {
    (void)st;
    const Rec *r = rec_find(b, 6);
    const int bps = (r->s.bits + 7) / 8, planes = (r->s.modeYUV & 6) ? 3 : 1;
    for (int i = 0; i < n; i++) {
        const mvx_blockfps_job *j = &jobs[i];
        const uint8_t *src[3] = { (const uint8_t *)j->src_super[0], (const uint8_t *)j->src_super[1], (const uint8_t *)j->src_super[2] };
        const uint8_t *ref[3] = { (const uint8_t *)j->ref_super[0], (const uint8_t *)j->ref_super[1], (const uint8_t *)j->ref_super[2] };
        const uint8_t *cl[3] = { (const uint8_t *)j->clip_left[0], (const uint8_t *)j->clip_left[1], (const uint8_t *)j->clip_left[2] };
        const uint8_t *cr[3] = { (const uint8_t *)j->clip_right[0], (const uint8_t *)j->clip_right[1], (const uint8_t *)j->clip_right[2] };
        uint8_t *dst[3] = { (uint8_t *)j->dst[0], (uint8_t *)j->dst[1], (uint8_t *)j->dst[2] };
        int copy = j->time256 == 0 ? 1 : j->time256 == 256 ? 2 : 0; /* MVBlockFPS.c:245-254 */
        if (!copy) {
            const int good = j->src_super[0] && j->ref_super[0] && j->blob_fw && j->blob_bw;
            if (mvo_blockfps_frame(&r->bf, j->time256, good ? src : NULL, r->superPitch, good ? ref : NULL, r->superPitch, good ? (const uint8_t *)j->blob_fw : NULL,
                                   good ? (const uint8_t *)j->blob_bw : NULL, cl, r->srcPitch, cr, r->srcPitch, dst, r->dstPitch) == 1) copy = 1;
        }
        if (copy)
            for (int p = 0; p < planes; p++) {
                const int wdt = p ? r->s.width / r->s.xRatioUV : r->s.width, hgt = p ? r->s.height / r->s.yRatioUV : r->s.height;
                copy2d(dst[p], r->dstPitch[p], copy == 1 ? cl[p] : cr[p], r->srcPitch[p], (size_t)wdt * bps, (size_t)hgt);
            }
    }
    return MVX_OK;
}

/* ---- mv.SCDetection: scene_change[i] = the i-th blob is not usable (MVSCDetection.c:43-73) */
API int mvx_scdetect(const mvx_analysis_data *vd, int64_t thscd1, int32_t thscd2, int n, const void *const *blobs, int32_t *scene_change, void *st, char *err) {
    (void)st; (void)err;
    mvo_analysis_data ad;
    memcpy(&ad, vd, sizeof(ad));
    int64_t t1 = thscd1 == MVX_UNSET ? 400 : thscd1; /* MVSCDetection.c:125-130 */
    int t2 = thscd2 == MVX_UNSET ? 130 : thscd2;
    mvo_scale_thscd(&t1, &t2, &ad);
    for (int i = 0; i < n; i++) scene_change[i] = !mvo_blob_is_usable(&ad, (const uint8_t *)blobs[i], t1, t2);
    return MVX_OK;
}

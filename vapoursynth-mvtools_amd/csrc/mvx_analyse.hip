// mvx_analyse.hip -- host side of mv.Analyse: argument resolution (MVAnalyse.c:267-635) and the batched launch.
// The search kernel itself lives in mvx_analyse_kernel.h and is instantiated by mvx_analyse_{any,u8,u16}.hip.
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include "mvx_analyse_kernel.h"
#include "mvx_analyse_fast.h"
#include "mvx_analyse_spec.h"
int mvx_analyse_launch_spec_u8(const AParams &P, const ASpecLaunch &S);
int mvx_analyse_launch_spec_u16(const AParams &P, const ASpecLaunch &S);

// ------------------------------------------------------------------------------------------------ host

static thread_local char g_err[512];
void mvx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" __attribute__((visibility("default"))) const char *mvx_last_error(void) { return g_err; }

// ---- developer / test options.  The library never reads the environment: kernel-variant choices that exist for A/B
// measurements and for running the parity suite through every kernel are set through ONE entry point, mvx_debug_option
// (the Python test binding forwards MVX_* environment variables to it; the VapourSynth shell never calls it).  None of them
// changes results, except "ablate", which only exists in MVX_LAB builds.
struct MvxDebug {
    int general = 0;   // 1: never use the lean kernel of the default search (mvx_analyse_fast.h)
    int fast_wpe = 0;  // > 0: chains per SIMD the lean kernel is launched at (when such a build exists)
    int cpw1 = 0;      // one chain per workgroup (general kernels)
    int no_wpe2 = 0, no_wpe3 = 0, wpe3_u16 = 0;
    int fast_cpw = 0;  // lean kernel: chains per workgroup (<= 4 * chains per SIMD)
    int fast_lds_min = 0; // lean kernel: LDS floor per workgroup in bytes (fewer workgroups per CU: a host that runs other kernels beside long search launches keeps registers free for them)
    int fast_k = 0;    // lean kernel: build for exactly that many chains per SIMD whatever the launch carries (several launches sharing the GPU)
    int fast_flags = -1; // lean kernel: MVX_FAST_* bits, -1 = default
    int pad_runs = -1;   // lean kernel: 1 = pad the job table so that the chains of one reference frame never straddle two workgroups
    int shadow_planes = 3; // 1 = luma only, 2 = chroma only uses the shadow copies
    int degrain_xcd = -1;  // Degrain cell kernels: XCD-contiguous tile order (1 / 0), -1 = default
    int degrain_shadow = 1; // Degrain: 0 = never read the shifted luma copies (takes effect at mvx_degrain_set_ref_shadow)
    int cpw_sync = -1; // barrier interval inside a workgroup (power of two, 0 = none)
    int lds_min = -1;  // LDS floor of the one-chain launches
    int spec = 1;      // default search: 1 = the speculative kernel (mvx_analyse_spec.h), 0 = the lean serial kernel (mvx_analyse_fast.h), 2 = the speculative kernel's code with speculation off (every block live), 3 = speculative without runs (every block's candidates loaded on their own), 5 = speculative for every shape it can run (by default only where its row passes apply)
    int team = -1;     // waves per chain of the speculative kernel's team form (mvx_analyse_spec.h: TEAM): 0 = never, 2..8 = always that many, -1 = the library's choice
    int super_rows_off = 0; // 1: mv.Super level 0 / first reduction through the LDS-tile / per-sample kernels only
    int shadow8 = 0;   // r6 experiment: 8-bit clips keep THREE byte-shifted copies of their luma planes too (takes effect at mvx_super_shadow_bytes / the Super calls / mvx_analyse_set_ref_shadow)
    int ablate = 0;
};
static MvxDebug g_dbg;
extern "C" __attribute__((visibility("default"))) int mvx_debug_option(const char *name, int value) {
    struct { const char *n; int *p; } tab[] = { { "general", &g_dbg.general }, { "fast_wpe", &g_dbg.fast_wpe }, { "cpw1", &g_dbg.cpw1 }, { "no_wpe2", &g_dbg.no_wpe2 }, { "no_wpe3", &g_dbg.no_wpe3 }, { "wpe3_u16", &g_dbg.wpe3_u16 }, { "fast_cpw", &g_dbg.fast_cpw }, { "fast_k", &g_dbg.fast_k }, { "fast_lds_min", &g_dbg.fast_lds_min }, { "fast_flags", &g_dbg.fast_flags }, { "pad_runs", &g_dbg.pad_runs }, { "shadow_planes", &g_dbg.shadow_planes }, { "degrain_xcd", &g_dbg.degrain_xcd }, { "degrain_shadow", &g_dbg.degrain_shadow }, { "cpw_sync", &g_dbg.cpw_sync }, { "lds_min", &g_dbg.lds_min }, { "super_rows_off", &g_dbg.super_rows_off }, { "spec", &g_dbg.spec }, { "team", &g_dbg.team }, { "shadow8", &g_dbg.shadow8 },
#ifdef MVX_LAB
        { "ablate", &g_dbg.ablate },
#endif
    };
    for (auto &t : tab) if (!strcmp(t.n, name)) { *t.p = value; return MVX_OK; }
    mvx_set_error("mvx_debug_option: unknown option %s", name);
    return MVX_E_ARG;
}

// what the last search launch of this process looked like (tests assert that a batch really took the build they mean to cover):
// out[0] = chains per SIMD of the lean kernel (0: the general kernel ran), out[1] = chains per workgroup, out[2] = barrier interval in
// blocks, out[3] = entries of the job table, out[4] = 1 when the LDS-window kernel ran
static std::atomic<int> g_lastLaunch[5];
extern "C" __attribute__((visibility("default"))) void mvx_debug_last_launch(int out[5]) {
    for (int i = 0; i < 5; i++) out[i] = g_lastLaunch[i].load();
}

int mvx_debug_value(const char *name, int def) {
    if (!strcmp(name, "degrain_xcd")) return g_dbg.degrain_xcd >= 0 ? g_dbg.degrain_xcd : def;
    if (!strcmp(name, "super_rows_off")) return g_dbg.super_rows_off;
    if (!strcmp(name, "shadow8")) return g_dbg.shadow8;
    if (!strcmp(name, "degrain_shadow")) return g_dbg.degrain_shadow;
    return def;
}

#define AFAIL(...) do { snprintf(err, MVX_ERRLEN, __VA_ARGS__); mvx_set_error("%s", err); return MVX_E_ARG; } while (0)
static int A(int v, int d) { return v == MVX_UNSET ? d : v; }

// MVAnalyse.c:267-635 mvanalyseCreate
// MVAnalyse.c:615-624 / MVRecalculate.c:533-543: the geometry readers see when divide > 0
void mvx_divided_data(const mvx_analysis_data *in, mvx_analysis_data *out) {
    *out = *in;
    out->nBlkX = in->nBlkX * 2; out->nBlkY = in->nBlkY * 2;
    out->nBlkSizeX = in->nBlkSizeX / 2; out->nBlkSizeY = in->nBlkSizeY / 2;
    out->nOverlapX = in->nOverlapX / 2; out->nOverlapY = in->nOverlapY / 2;
    out->nLvCount = in->nLvCount + 1;
}

// GroupOfPlanes.c:177-302 Median3 / GetMedian / gopExtraDivide.  One thread per block of the finest estimated plane writes its
// four sub-blocks; invalid blobs get the default vector (PlaneOfBlocks.cpp:1543-1553).  Also writes the divided array's size
// header, which the reference leaves uninitialised on the search path.
__device__ __forceinline__ int mvx_median3(int a, int b, int c) {
    if (((b <= a) && (a <= c)) || ((c <= a) && (a <= b))) return a;
    if (((a <= b) && (b <= c)) || ((c <= b) && (b <= a))) return b;
    return c;
}
__device__ __forceinline__ void mvx_get_median(GVec &o, const GVec &v1, const GVec &v2, const GVec &v3) {
    int vx = mvx_median3(v1.x, v2.x, v3.x), vy = mvx_median3(v1.y, v2.y, v3.y);
    if (!((vx == v1.x && vy == v1.y) || (vx == v2.x && vy == v2.y) || (vx == v3.x && vy == v3.y))) { vx = v1.x; vy = v1.y; }
    o.x = vx; o.y = vy;
}
__global__ __launch_bounds__(256) void analyse_divide_kernel(const AParams *Pp, const AJob *jobs) {
    const AParams &P = *Pp;
    unsigned char *blob = jobs[blockIdx.y].blob;
    if (!blob) return; // padding entry of the job table
    const int valid = ((const int *)blob)[1];
    const int nBlkX = P.lv[0].nBlkX, nBlkY = P.lv[0].nBlkY, nBlk = nBlkX * nBlkY;
    unsigned char *rec0 = blob + P.lv[0].blobOff;
    const GVec *in = (const GVec *)(rec0 + 4);
    unsigned char *hdr = rec0 + 4 + nBlk * 16;
    GVec *out = (GVec *)(hdr + 4);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *(int *)hdr = 4 + nBlk * 64;
    if (i >= nBlk) return;
    const int by = i / nBlkX, bx = i - by * nBlkX;
    GVec b[4];
    if (!valid) { for (int k = 0; k < 4; k++) { b[k].x = 0; b[k].y = 0; b[k].sad = P.verybigSAD; } }
    else {
        GVec c = in[i];
        GVec s = c; s.sad >>= 2;
        b[0] = b[1] = b[2] = b[3] = s;
        if (P.divide > 1 && by >= 1 && by < nBlkY - 1 && bx >= 1 && bx < nBlkX - 1) {
            const GVec l = in[i - 1], r = in[i + 1], u = in[i - nBlkX], d = in[i + nBlkX];
            mvx_get_median(b[0], c, l, u); mvx_get_median(b[1], c, r, u); mvx_get_median(b[2], c, l, d); mvx_get_median(b[3], c, r, d);
        }
    }
    GVec *o = out + (size_t)by * nBlkX * 4 + bx * 2;
    o[0] = b[0]; o[1] = b[1]; o[nBlkX * 2] = b[2]; o[nBlkX * 2 + 1] = b[3];
}

extern "C" __attribute__((visibility("default"))) int mvx_analyse_create(const mvx_analyse_args *a, const mvx_super *sup, int num_frames, const ptrdiff_t super_pitch[3],
                                  mvx_analyse **out, char *err) {
    char dummy[MVX_ERRLEN];
    if (!err) err = dummy;
    err[0] = 0;
    *out = nullptr;
    const mvx_super_info &si = sup->info;
    mvx_analysis_data ad;
    memset(&ad, 0, sizeof(ad));
    AParams P;
    memset(&P, 0, sizeof(P));
    ad.nBlkSizeX = A(a->blksize, 8);
    ad.nBlkSizeY = A(a->blksizev, ad.nBlkSizeX);
    const int levels = A(a->levels, 0);
    P.searchType = A(a->search, SearchHex2);
    P.searchTypeCoarse = A(a->search_coarse, SearchExhaustive);
    const int searchparam = A(a->searchparam, 2);
    P.nPelSearch = A(a->pelsearch, 0);
    ad.isBackward = !!A(a->isb, 0);
    int chroma = !!A(a->chroma, 1);
    ad.nDeltaFrame = A(a->delta, 1);
    const int truemotion = !!A(a->truemotion, 1);
    int nLambda = A(a->lambda, truemotion ? (1000 * ad.nBlkSizeX * ad.nBlkSizeY / 64) : 0);
    int lsad = A(a->lsad, truemotion ? 1200 : 400);
    P.plevel = A(a->plevel, truemotion ? 1 : 0);
    P.global = !!A(a->global, truemotion ? 1 : 0);
    P.pnew = A(a->pnew, truemotion ? 50 : 0);
    P.pzero = A(a->pzero, P.pnew);
    P.pglobal = A(a->pglobal, 0);
    ad.nOverlapX = A(a->overlap, 0);
    ad.nOverlapY = A(a->overlapv, ad.nOverlapX);
    P.dctmode = A(a->dct, 0);
    const int divide = A(a->divide, 0);
    long long badSAD = A(a->badsad, 10000);
    P.badrange = A(a->badrange, 24);
    const int opt = !!A(a->opt, 1);
    P.meander = !!A(a->meander, 1);
    P.tryMany = !!A(a->trymany, 0);

    if (P.searchType < 0 || P.searchType > 7) AFAIL("Analyse: search must be between 0 and 7 (inclusive).");
    if (P.searchTypeCoarse < 0 || P.searchTypeCoarse > 7) AFAIL("Analyse: search_coarse must be between 0 and 7 (inclusive).");
    if (P.dctmode < 0 || P.dctmode > 10) AFAIL("Analyse: dct must be between 0 and 10 (inclusive).");
    if (P.dctmode >= 5 && ad.nBlkSizeX == 16 && ad.nBlkSizeY == 2) AFAIL("Analyse: dct 5..10 cannot work with 16x2 blocks.");
    if (P.dctmode >= 1 && P.dctmode <= 4) AFAIL("Analyse: dct 1..4 (FFTW3 DCT cost) are not implemented on the GPU path.");
    if (divide < 0 || divide > 2) AFAIL("Analyse: divide must be between 0 and 2 (inclusive).");
    {
        static const int okb[12][2] = { { 4, 4 }, { 8, 4 }, { 8, 8 }, { 16, 2 }, { 16, 8 }, { 16, 16 }, { 32, 16 }, { 32, 32 }, { 64, 32 }, { 64, 64 }, { 128, 64 }, { 128, 128 } };
        bool found = false;
        for (auto &b : okb) found |= (ad.nBlkSizeX == b[0] && ad.nBlkSizeY == b[1]);
        if (!found) AFAIL("Analyse: the block size must be 4x4, 8x4, 8x8, 16x2, 16x8, 16x16, 32x16, 32x32, 64x32, 64x64, 128x64, or 128x128.");
    }
    if (P.plevel < 0 || P.plevel > 2) AFAIL("Analyse: plevel must be between 0 and 2 (inclusive).");
    if (P.pnew < 0 || P.pnew > 256) AFAIL("Analyse: pnew must be between 0 and 256 (inclusive).");
    if (P.pzero < 0 || P.pzero > 256) AFAIL("Analyse: pzero must be between 0 and 256 (inclusive).");
    if (P.pglobal < 0 || P.pglobal > 256) AFAIL("Analyse: pglobal must be between 0 and 256 (inclusive).");
    if (ad.nOverlapX < 0 || ad.nOverlapX > ad.nBlkSizeX / 2 || ad.nOverlapY < 0 || ad.nOverlapY > ad.nBlkSizeY / 2)
        AFAIL("Analyse: overlap must be at most half of blksize, overlapv must be at most half of blksizev, and they both need to be at least 0.");
    if (divide && (ad.nBlkSizeX < 8 || ad.nBlkSizeY < 8)) AFAIL("Analyse: blksize and blksizev must be at least 8 when divide=True."); // :447
    if (divide && (ad.nOverlapX % (2 * si.xRatioUV) || ad.nOverlapY % (2 * si.yRatioUV)))                                                    // :503-505
        AFAIL("Analyse: overlap and overlapv must be multiples of 2 or 4 when divide=True, depending on the super clip's subsampling.");
    if (P.searchType == SearchNstep) P.nSearchParam = searchparam < 0 ? 0 : searchparam;
    else P.nSearchParam = searchparam < 1 ? 1 : searchparam;

    if (si.gray) chroma = 0;
    const int nModeYUV = chroma ? 7 : 1;
    ad.bitsPerSample = si.bits;
    const int pixelMax = (1 << si.bits) - 1; // :477-483
    lsad = (int)((double)lsad * pixelMax / 255.0 + 0.5);
    badSAD = (int)((double)badSAD * pixelMax / 255.0 + 0.5);
    nLambda = (int)((double)nLambda * pixelMax / 255.0 + 0.5);
    lsad = (int)((int64_t)lsad * (ad.nBlkSizeX * ad.nBlkSizeY) / 64);
    badSAD = badSAD * (ad.nBlkSizeX * ad.nBlkSizeY) / 64;

    ad.nMotionFlags = (opt ? MOTION_USE_SIMD : 0) | (ad.isBackward ? MOTION_IS_BACKWARD : 0) | (chroma ? MOTION_USE_CHROMA_MOTION : 0);
    ad.nCPUFlags = 0; // host-dependent in the reference (g_cpuinfo, MVAnalyse.c:492-494); no CPU kernels here
    if (ad.nOverlapX % si.xRatioUV || ad.nOverlapY % si.yRatioUV)
        AFAIL("Analyse: The requested overlap is incompatible with the super clip's subsampling.");
    if (ad.nDeltaFrame <= 0 && (-ad.nDeltaFrame) >= num_frames) AFAIL("Analyse: delta points to frame past the input clip's end.");
    ad.yRatioUV = si.yRatioUV; ad.xRatioUV = si.xRatioUV;
    if ((nModeYUV & si.modeYUV) != nModeYUV) AFAIL("Analyse: super clip does not contain needed colour data.");
    ad.nWidth = si.super_width - si.hpad * 2;
    ad.nHeight = si.height;
    ad.nPel = si.pel;
    ad.nHPadding = si.hpad; ad.nVPadding = si.vpad;
    const int nBlkX = (ad.nWidth - ad.nOverlapX) / (ad.nBlkSizeX - ad.nOverlapX);
    const int nBlkY = (ad.nHeight - ad.nOverlapY) / (ad.nBlkSizeY - ad.nOverlapY);
    ad.nBlkX = nBlkX; ad.nBlkY = nBlkY;
    const int nWidth_B = (ad.nBlkSizeX - ad.nOverlapX) * nBlkX + ad.nOverlapX;
    const int nHeight_B = (ad.nBlkSizeY - ad.nOverlapY) * nBlkY + ad.nOverlapY;
    int nLevelsMax = 0;
    while (((nWidth_B >> nLevelsMax) - ad.nOverlapX) / (ad.nBlkSizeX - ad.nOverlapX) > 0 &&
           ((nHeight_B >> nLevelsMax) - ad.nOverlapY) / (ad.nBlkSizeY - ad.nOverlapY) > 0)
        nLevelsMax++;
    ad.nLvCount = levels > 0 ? levels : nLevelsMax + levels;
    if (ad.nLvCount < 1 || ad.nLvCount > nLevelsMax) AFAIL("Analyse: invalid number of levels.");
    if (ad.nLvCount > si.levels) AFAIL("Analyse: super clip has %d levels. Analyse needs %d levels.", si.levels, ad.nLvCount);
    if (ad.nLvCount > MVX_MAX_LEVELS) AFAIL("Analyse: too many levels.");
    if (P.nPelSearch <= 0) P.nPelSearch = ad.nPel;

    // ---- device parameter block
    P.nLevels = ad.nLvCount;
    P.blkX = ad.nBlkSizeX; P.blkY = ad.nBlkSizeY; P.ovX = ad.nOverlapX; P.ovY = ad.nOverlapY;
    P.xr = si.xRatioUV; P.yr = si.yRatioUV; P.logxr = mvx_ilog2(P.xr); P.logyr = mvx_ilog2(P.yr);
    P.bits = si.bits; P.bps = (si.bits + 7) / 8; P.chroma = chroma;
    P.lambda = nLambda; P.lsad = lsad; P.badSAD = badSAD;
    P.verybigSAD = (long long)P.blkX * P.blkY * (1 << si.bits);
    P.superHPad = si.hpad; P.superVPad = si.vpad;
    P.ablate = g_dbg.ablate;
    for (int p = 0; p < 3; p++) P.pitch[p] = p < si.num_planes ? super_pitch[p] : 0;
    if (si.num_planes > 1 && super_pitch[1] != super_pitch[2]) AFAIL("Analyse: the U and V planes of the super clip must share one pitch.");
    int blobOff = 8;
    for (int i = ad.nLvCount - 1; i >= 0; i--) { // GroupOfPlanes.c:25-56 (block grid per level), :167-174 (blob layout)
        ALevel &L = P.lv[i];
        L.nBlkX = ((nWidth_B >> i) - ad.nOverlapX) / (ad.nBlkSizeX - ad.nOverlapX);
        L.nBlkY = ((nHeight_B >> i) - ad.nOverlapY) / (ad.nBlkSizeY - ad.nOverlapY);
        L.pel = i == 0 ? ad.nPel : 1;
        L.logPel = mvx_ilog2(L.pel);
        LevelPlane y, c;
        mvx_level_plane(si, i, 0, P.pitch[0], &y);
        L.pw = y.pw; L.ph = y.ph; L.hpad = y.hpad; L.vpad = y.vpad;
        L.off[0] = y.off; L.pstride[0] = P.pitch[0] * y.ph;
        if (si.num_planes > 1) {
            mvx_level_plane(si, i, 1, P.pitch[1], &c);
            L.cpw = c.pw; L.cph = c.ph; L.chpad = c.hpad; L.cvpad = c.vpad;
            L.off[1] = c.off; L.pstride[1] = P.pitch[1] * c.ph;
            mvx_level_plane(si, i, 2, P.pitch[2], &c);
            L.off[2] = c.off; L.pstride[2] = P.pitch[2] * c.ph;
        }
        L.blobOff = blobOff;
        blobOff += 4 + L.nBlkX * L.nBlkY * 16;
    }
    P.divide = divide;
    if (divide) blobOff += 4 + P.lv[0].nBlkX * P.lv[0].nBlkY * 16 * 4; // PlaneOfBlocks.cpp:1517-1526
    P.blobSize = blobOff;

    mvx_analyse *h = new mvx_analyse();
    h->ad = ad;
    h->adOut = ad;
    if (divide) mvx_divided_data(&ad, &h->adOut);
    h->P = P;
    *out = h; // device state is created on first use so that argument validation works without a GPU
    return MVX_OK;
}

extern "C" __attribute__((visibility("default"))) void mvx_analyse_destroy(mvx_analyse *a) {
    if (!a) return;
    if (a->dP) (void)hipFree(a->dP);
    for (auto &sl : a->slot) if (sl.d) (void)hipFree(sl.d);
    delete a;
}
extern "C" __attribute__((visibility("default"))) int mvx_analyse_set_ref_shadow(mvx_analyse *a, const ptrdiff_t copy_stride[3]) {
    for (int p = 0; p < 3; p++) {
        const long long v = copy_stride ? (long long)copy_stride[p] : 0;
        if (v < 0 || v % 16) { mvx_set_error("mvx_analyse_set_ref_shadow: copy strides must be non-negative multiples of 16 bytes"); return MVX_E_ARG; }
        a->P.shadow[p] = ((p == 0 && !(g_dbg.shadow_planes & 1)) || (p > 0 && !(g_dbg.shadow_planes & 2))) ? 0 : v;
        if (p == 0 && a->P.bps == 1 && !g_dbg.shadow8) a->P.shadow[0] = 0; // (8-bit super frames carry the UV plane only)
    }
    std::lock_guard<std::mutex> lk(a->guard.mu);
    if (a->dP) HIP_CHECK(hipMemcpy(a->dP, &a->P, sizeof(AParams), hipMemcpyHostToDevice)); // (synchronous: no launch of this handle is reading it concurrently unless the caller races)
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) void mvx_analyse_get_data(const mvx_analyse *a, mvx_analysis_data *out) { *out = a->adOut; }
extern "C" __attribute__((visibility("default"))) int mvx_analyse_blob_size(const mvx_analyse *a) { return a->P.blobSize; }

// Waves per chain the speculative kernel is launched with when the caller does not say (mvx_debug_option "team"): 0 = one wave per chain.
// Measured r5 on cfg3 (profiles/r5_team_first_bench.txt, r5_team_batch_sweep.txt; ms per launch, one wave / team of 2 / 4 / 8):
//   132 chains 191 / - / 71 / 69,   258: 201 / - / 76 / 90,   516: 205 / - / 145 / 141,   768: 209 / 165 / 200 / -,   1020: 226 / 226 / - / -,
//   2046: 342 / 461 / 421 / 473.
// A launch that leaves wave slots empty with one wave per chain (two per SIMD fit) finishes sooner as teams; one that fills the GPU anyway does
// not: four chains of a workgroup that share a reference frame and walk in step fetch each line once for all of them, a team's chains do not
// (2 444 against 1 302 GB of line traffic per 2046-chain launch, L2 hit rate 43 against 62 %), and the token makes a team wave wait for its
// predecessors' verification (40 % of a wave's time on a chain with many live blocks: profiles/r5_team_phase_cycles.txt).
// 8-bit 8x8 (cfg2, profiles/r5_team_cfg2_sweep.txt): 128 chains 118 / - / 50 / 46,   512: 120 / 77 / 63 / -,   1024: 121 / 94 / - / -,   2048: 164 (one wave).
static int mvx_team_default(int njobs, int simds, bool strips, int bps) {
    const long long slots = 2LL * simds; // waves the 256-register build keeps resident
    if (njobs * 4LL <= slots * 11 / 10) return 4;
    if (!strips) return 0; // (shapes without row passes: teams of two do not beat the serial kernel)
    if (njobs * 2LL <= (bps == 1 ? slots : slots * 9 / 10)) return 2;
    return 0;
}

extern "C" __attribute__((visibility("default"))) int mvx_analyse_frames(mvx_analyse *a, int njobs, const mvx_analyse_job *jobs, void *stream) {
    if (njobs <= 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    const AParams &P = a->P;
    {
        std::lock_guard<std::mutex> lk(a->guard.mu);
        if (!a->dP) {
            HIP_CHECK(hipGetDevice(&a->device));
            HIP_CHECK(hipMalloc((void **)&a->dP, sizeof(AParams)));
            HIP_CHECK(hipMemcpy(a->dP, &P, sizeof(AParams), hipMemcpyHostToDevice));
        }
    }
    mvx_analyse::JobSlot &S = a->slot[a->nextSlot.fetch_add(1) % mvx_analyse::kSlots];
    CallGuard::Scope scope(S.guard, st);
    if ((size_t)njobs > S.cap) {
        if (S.d) (void)hipFree(S.d);
        S.cap = (size_t)njobs * 2;
        HIP_CHECK(hipMalloc((void **)&S.d, S.cap * sizeof(AJob)));
    }
    std::vector<AJob> hj(njobs);
    for (int i = 0; i < njobs; i++) {
        for (int p = 0; p < 3; p++) { hj[i].src[p] = (const unsigned char *)jobs[i].src[p]; hj[i].ref[p] = (const unsigned char *)jobs[i].ref[p]; }
        hj[i].blob = (unsigned char *)jobs[i].blob;
        hj[i].fieldShift = jobs[i].field_shift;
        hj[i].valid = jobs[i].ref[0] != nullptr;
        if (((uintptr_t)jobs[i].blob) & 15) { mvx_set_error("mvx_analyse_frames: blob must be 16-byte aligned"); return MVX_E_ARG; }
    }
    // Chain placement.  Chains that search the same reference frame read the same lines at about the same time (they start
    // together and advance at the same pace).  The specialised kernels run FOUR (or eight) chains per workgroup, one (two) per
    // SIMD of a CU, and the job table is sorted by reference frame first, so the waves of a workgroup share their reference
    // lines in the CU's L1 (+4.7 % at 4K16; measured r1: a barrier between them only costs -- per block -13 %, per 16 blocks 0 %,
    // per row +3.7 % -- and dealing the groups out so that each XCD gets a contiguous range of frames changes nothing).
    const bool spec = P.dctmode == 0 && P.xr == 2 && P.yr == 2 && P.blkX == P.blkY && (P.blkX == 16 || P.blkX == 8 || (P.blkX == 32 && P.bps == 2)) &&
                      !g_dbg.cpw1;
    int simds = 0; // of the device this call runs on
    {
        int dev = 0, cus = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        simds = 4 * cus;
    }
    // A launch with more chains than SIMDs runs TWO chains per SIMD where a 256-register build of the kernel exists without
    // spills (8-bit 8x8, 16x16; 16-bit 8x8, 16x16, 32x32): 8-bit 1080p +53 %; 16-bit 4K +13 %, and there only as workgroups of EIGHT chains
    // that share a reference frame -- eight unrelated chains per CU thrash its L1 / the XCD's L2 and lose.
    // ---- the default search runs in the lean kernel (mvx_analyse_fast.h): workgroups of 4 * k chains that share a reference
    // frame, k = 1..4 chains per SIMD depending on how many chains the launch carries and which builds exist / pay off.
    if (mvx_fast_eligible(P) && !g_dbg.general && !g_dbg.cpw1) {
        int srcB = P.blkX * P.blkY * P.bps + 2 * (P.blkX / 2) * (P.blkY / 2) * P.bps;
        const int fRow = (srcB + 15) & ~15;
        int fMaxBlkX = 0;
        for (int i = 0; i < P.nLevels; i++) if (P.lv[i].nBlkX > fMaxBlkX) fMaxBlkX = P.lv[i].nBlkX;
        // [source block | previous row's results, 16 B per block]; the histogram of the global-motion estimate (only used between
        // levels) lies over the row buffer
        const int fBins = 1024;
        int fNeed = fRow + fMaxBlkX * 16;
        if (fNeed < fRow + fBins * 4) fNeed = fRow + fBins * 4;
        // the speculative kernel (mvx_analyse_spec.h): [source block | previous row's results, 8 B per block | SAD table of a 32-block group];
        // the histogram lies over row buffer and table
        // The speculative kernel runs where its row passes apply (16-bit 16x16 blocks overlapping by half, with chroma: cfg3 612 against 555 fps);
        // one block at a time it loses to the serial lean kernel (cfg2 2 557 / 2 894, cfg4 17 180 / 18 764, cfg5 76 / 92 fps:
        // profiles/r4_configs_spec_vs_serial.txt), so everything else stays there unless "spec" asks for it (5: wherever it can run).
        const bool stripShape8 = P.bps == 1 && P.blkX == 8 && P.chroma && (P.ovX == 4 || P.ovX == 0) && P.shadow[1] != 0; // 8-bit 8x8 blocks overlapping by half or not at all, UV-interleaved plane present
        const bool uvOrLumaOnly = P.chroma ? P.shadow[1] != 0 : true; // the row passes read chroma from the UV-interleaved plane; a luma-only search (chroma = 0) needs none (r5)
        const bool stripShape16 = P.bps == 1 && P.blkX == 16 && uvOrLumaOnly && (P.ovX == 8 || P.ovX == 0); // r5: 8-bit 16x16 blocks overlapping by half (the 16-bit form with 8-byte columns)
        const bool stripShape = (P.bps == 2 && (P.blkX == 16 || (MVX_STRIP32 && P.blkX == 32)) && uvOrLumaOnly && (P.ovX == P.blkX / 2 || (P.blkX == 16 && P.ovX == 0))) || stripShape8 || stripShape16; // (16x16 side by side: r5) // (the row passes read the UV-interleaved plane: STRIP_OK)
        // ... except as TEAMS in a launch that leaves the GPU's wave slots empty (r5): there the speculative kernel wins for every shape it can run, row passes or
        // not (132 chains of cfg5: 974 ms serial, 352 ms as teams of four; 128 chains of 8-bit 16x16 blocks: 140 / 43 ms; at ~512 chains it is a tie:
        // profiles/r5_team_other_shapes.txt).  The library's own choice only: any forced "spec" / "team" value keeps its meaning.
        // (r6, ADVICE r5: whether a team FITS is decided here, before anything is sized for the speculative kernel -- a shape without row passes whose team
        // would not fit the CU's LDS (a very wide frame) stays on the serial kernel with the serial kernel's LDS sizing instead of running the one-wave
        // speculative kernel, which loses to it: cfg2, cfg4, cfg5 above)
        const int sStrip = (P.bps == 2 && P.blkX == 16 && MVX_STRIP_DMA) ? 2 * 24 * 128 /* r6: two strip buffers, filled by LDS-direct loads */ : (P.bps == 2 && (P.blkX == 16 || (MVX_STRIP32 && P.blkX == 32))) ? (P.blkX + P.blkX / 2) * 128 : (P.bps == 1 && P.blkX == 16) ? 24 * 64 : 0; // the source strip of a window of blocks: 24 (48) rows x 8 columns (mvx_analyse_spec.h: STRIP_OK)
        const int sSrc = stripShape8 ? 16 + 768 : fRow < sStrip ? sStrip : fRow; // (8-bit 8x8: 16 bytes of slack + 12 rows x 64 bytes)
        int sNeed = 0, sTabMax = 0; // per level: row buffer (8 B per block of THAT level) + the table of its search type; the largest SAD table of any level
        for (int i = 0; i < P.nLevels; i++) {
            const bool smallest = i == P.nLevels - 1;
            const int st = smallest ? (P.nLevels == 1 ? P.searchType : P.searchTypeCoarse) : (i == 0 ? P.searchType : P.searchTypeCoarse);
            const int tabB = (st == SearchHex2 ? SPEC_SLOTS_HEX : SPEC_SLOTS_EXH) * SPEC_STRIDE;
            const int need = sSrc + ((P.lv[i].nBlkX * 8 + 15) & ~15) + tabB;
            if (need > sNeed) sNeed = need;
            if (tabB > sTabMax) sTabMax = tabB;
        }
        if (sNeed < sSrc + fBins * 4) sNeed = sSrc + fBins * 4;
        // the team form's LDS: [64 B control words | row buffer] shared + per wave [source strip / block | SAD table]; the histogram of the global-motion estimate lies over the table
        const int teamShared = (64 + fMaxBlkX * 8 + 255) & ~255;
        const int teamPerWave = ((sSrc + sTabMax < sSrc + fBins * 4 ? sSrc + fBins * 4 : sSrc + sTabMax) + 255) & ~255;
        auto teamThatFits = [&](int team) { // waves per chain after the LDS limit (0: no team)
            if (team == 1 || team > 8) team = 0;
            while (team > 1 && teamShared + teamPerWave * team > 160 * 1024) team--;
            return team > 1 ? team : 0;
        };
        const bool teamAnyShape = g_dbg.spec == 1 && g_dbg.team < 0 && teamThatFits(mvx_team_default(njobs, simds, false, P.bps)) > 0;
        const bool useSpec = g_dbg.spec != 0 && (stripShape || g_dbg.spec >= 2 || teamAnyShape);
        const bool useSpecStrips = useSpec && g_dbg.spec != 3 && stripShape;
        int sTab = 0, sRow = fRow;
        if (useSpec) {
            sRow = sSrc;
            sTab = sSrc + ((fMaxBlkX * 8 + 15) & ~15);
            fNeed = sNeed;
        }
        const int perChain = (fNeed + 255) & ~255;
        // builds per (sample size, block size): chains per SIMD that exist (mvx_analyse_u8.hip / _u16.hip)
        auto have = [&](int k) {
            if (P.bps == 2 && P.blkX == 8) return k == 1 || k == 2 || k == 4;
            if (P.bps == 2 && P.blkX == 32) return k >= 1 && k <= 3;
            return k >= 1 && k <= 4;
        };
        int k = njobs > 3 * simds ? 4 : njobs > 2 * simds ? 3 : njobs > simds ? 2 : 1;
        // the row passes of the speculative kernel (16-bit 16x16) want the 256-register builds: 3072 chains take 878 ms at two per SIMD (in
        // two rounds) against 919 at three (profiles/r4_spec_loads_in_flight_and_batch.txt)
        if (useSpecStrips && k > 2 && g_dbg.fast_k != 3) k = 2;
        if (g_dbg.fast_wpe > 0 && g_dbg.fast_wpe < k) k = g_dbg.fast_wpe;
        if (g_dbg.fast_k > 0) k = g_dbg.fast_k;
        while (k > 1 && (!have(k) || (long long)perChain * 4 * k > 160 * 1024)) k--;
        if ((long long)perChain * 4 * k <= 160 * 1024) {
            std::stable_sort(hj.begin(), hj.end(), [](const AJob &x, const AJob &y) { return (uintptr_t)x.ref[0] > (uintptr_t)y.ref[0]; }); // (no reference: last)
            int cpw = 4 * k;
            // the speculative kernel's chains wait for each other less in workgroups of FOUR (two per CU) that meet every 128 blocks: 345 ms per
            // 2046-chain launch against 370 with eight and a barrier per group (profiles/r4_spec_twostage_workgroup_size.txt, ..._barrier_interval.txt)
            if (useSpecStrips) cpw = 4;
            if (g_dbg.fast_cpw > 0 && g_dbg.fast_cpw < 4 * k) cpw = g_dbg.fast_cpw;
            // keep the chains that share a reference frame in one workgroup where they fit: a run of the sorted table that would
            // straddle a workgroup boundary starts a new workgroup instead (the skipped slots become padding entries, blob == NULL)
            std::vector<AJob> padded;
            const std::vector<AJob> *table = &hj;
            if (g_dbg.pad_runs == 1) {
                AJob none;
                memset(&none, 0, sizeof(none));
                for (size_t i = 0; i < hj.size();) {
                    size_t j = i;
                    while (j < hj.size() && hj[j].ref[0] == hj[i].ref[0]) j++;
                    const size_t len = j - i, used = padded.size() % cpw;
                    if (used && len <= (size_t)cpw && used + len > (size_t)cpw) padded.resize(padded.size() + (cpw - used), none);
                    padded.insert(padded.end(), hj.begin() + i, hj.begin() + j);
                    i = j;
                }
                table = &padded;
            }
            const int ntab = (int)table->size();
            if ((size_t)ntab > S.cap) {
                if (S.d) (void)hipFree(S.d);
                S.cap = (size_t)ntab * 2;
                HIP_CHECK(hipMalloc((void **)&S.d, S.cap * sizeof(AJob)));
            }
            HIP_CHECK(hipMemcpyAsync(S.d, table->data(), sizeof(AJob) * ntab, hipMemcpyHostToDevice, st));
            // Barrier between the chains of a workgroup every that many blocks of a row.  16-bit clips (shadow layout: the kernel is
            // bound by what the XCD's L2 must re-fetch): the chains that share a reference frame have to stay within a few blocks of
            // each other to share its lines -- measured r2 at three chains per SIMD (4K16): 256 blocks 337 fps, 128: 378, 16-64: 389-390,
            // 8: 388, 4: 383; 8K16 at two per SIMD: 256: 59.5, 16: 64.6.  8-bit clips (plain layout): 256 stays best (1080p: 2005 against
            // 1953-1970 with 16-64).
            int syncEvery = k >= 2 ? (P.bps == 2 ? 32 : 256) : 0;
            if (useSpecStrips) syncEvery = (stripShape8 || stripShape16) ? 0 : 128; // (8-bit 16x16, 4096 chains: none 144 ms, every 128 blocks 147: profiles/r5_hd16_bench.txt) // (8-bit, 4096 chains: none 330 ms, every 512 blocks 337, 256: 344, 128: 366 -- profiles/r4_rows8_configs2.txt)
            if (g_dbg.cpw_sync >= 0 && (g_dbg.cpw_sync & (g_dbg.cpw_sync - 1)) == 0) syncEvery = g_dbg.cpw_sync;
            // XCD-contiguous workgroup order: neighbours in the (reference-sorted) job table share an L2 (+0.5 %, 4K16)
            const int flags = (g_dbg.fast_flags >= 0 ? g_dbg.fast_flags : MVX_FAST_XCD_REMAP) | ((P.shadow[1] != 0 && P.chroma) ? MVX_FAST_UV : 0) | (g_dbg.spec == 2 ? MVX_FAST_NOSPEC : 0) | (g_dbg.spec == 3 ? MVX_FAST_NOSTRIP : 0);
            ALaunch L = { ntab, fNeed, useSpec ? sRow : fRow, useSpec ? sRow : fRow, fBins, fNeed, simds, cpw, k, syncEvery, k, flags, st, a->dP, S.d };
            L.ldsBytes = g_dbg.fast_lds_min; // (floor of the workgroup's LDS request, 0 = none)
            int rc;
            // TEAM form (r5): the nw waves of a workgroup walk ONE chain.  A chain finishes ~nw times sooner and a launch keeps nw times fewer chains resident
            // per wave slot, so it is the form for launches that do not fill the GPU with one wave per chain (a frame server's look-ahead window); for
            // big batches the choice is measured.  LDS: [64 B control words | row buffer] shared + per wave [source strip | SAD table]
            const int specSide = (P.blkX == 16 && P.ovX == 0) ? 1 : 0; // 16x16 blocks side by side: the SIDE builds of the speculative kernel (their row passes step by a whole block)
            int team = 0;
            if (useSpec) {
                team = g_dbg.team >= 0 ? g_dbg.team : mvx_team_default(njobs, simds, useSpecStrips, P.bps);
                if (team == 1 || team > 8) team = 0;
            }
            team = teamThatFits(team);
            if (team) {
                const int shared = teamShared, perWave = teamPerWave;
                if (team > 1) {
                    ALaunch TL = L;
                    TL.ldsRow = shared; TL.ldsNeed = perWave; TL.ldsHist = sRow; TL.syncEvery = 0; TL.fast = 2; TL.cpw = 1;
                    // (the job table may carry padding entries, blob == NULL, from the run padding above: the kernel skips them, `!J.blob`)
                    const ASpecLaunch SL = { TL, sRow, team, specSide };
                    rc = P.bps == 1 ? mvx_analyse_launch_spec_u8(P, SL) : mvx_analyse_launch_spec_u16(P, SL);
                    if (rc == MVX_OK) {
                        g_lastLaunch[0] = 2; g_lastLaunch[1] = team; g_lastLaunch[2] = 0; g_lastLaunch[3] = ntab; g_lastLaunch[4] = 3;
                        if (P.divide) hipLaunchKernelGGL(analyse_divide_kernel, dim3((P.lv[0].nBlkX * P.lv[0].nBlkY + 255) / 256, ntab), dim3(256), 0, st, a->dP, S.d);
                        HIP_CHECK(hipGetLastError());
                        return MVX_OK;
                    }
                    if (rc != 1) return rc;
                }
            }
            if (useSpec) { const ASpecLaunch SL = { L, sTab, 0, specSide }; rc = P.bps == 1 ? mvx_analyse_launch_spec_u8(P, SL) : mvx_analyse_launch_spec_u16(P, SL); }
            else rc = P.bps == 1 ? mvx_analyse_launch_fast_u8(P, L) : mvx_analyse_launch_fast_u16(P, L);
            if (rc == MVX_OK) {
                g_lastLaunch[0] = k; g_lastLaunch[1] = cpw; g_lastLaunch[2] = syncEvery; g_lastLaunch[3] = ntab; g_lastLaunch[4] = useSpec ? 2 : 0;
                if (P.divide) hipLaunchKernelGGL(analyse_divide_kernel, dim3((P.lv[0].nBlkX * P.lv[0].nBlkY + 255) / 256, ntab), dim3(256), 0, st, a->dP, S.d);
                HIP_CHECK(hipGetLastError());
                return MVX_OK;
            }
            if (rc != 1) return rc;
        }
    }
    const bool oneChain = g_dbg.cpw1 != 0;
    int cpw = oneChain ? 1 : 4, wpe = 1; // (the generic kernels have four-chain builds too)
    if (spec && njobs > simds && !g_dbg.no_wpe2) {
        if (P.bps == 1 && (P.blkX == 8 || P.blkX == 16)) wpe = 2;
        // three per SIMD for the lightest kernel: its 168-register build spills 72 registers and still gains 14 % (1080p 1642 -> 1867 fps)
        if (P.bps == 1 && P.blkX == 8 && njobs > 2 * simds && !g_dbg.no_wpe3) wpe = 3;
        if (P.bps == 2 && (P.blkX == 16 || P.blkX == 8 || P.blkX == 32)) { wpe = 2; cpw = 8; }
        if (P.bps == 2 && P.blkX == 16 && njobs > 2 * simds && g_dbg.wpe3_u16) { wpe = 3; cpw = 12; }
    }
    if (cpw > 1) std::stable_sort(hj.begin(), hj.end(), [](const AJob &x, const AJob &y) { return (uintptr_t)x.ref[0] > (uintptr_t)y.ref[0]; }); // (no reference: last)
    HIP_CHECK(hipMemcpyAsync(S.d, hj.data(), sizeof(AJob) * njobs, hipMemcpyHostToDevice, st));
    // LDS: [source block (Y,U,V) | previous-row vectors | predictor rows (this, below) | histogram]
    int srcBytes = P.blkX * P.blkY * P.bps;
    if (P.chroma) srcBytes += 2 * (P.blkX / P.xr) * (P.blkY / P.yr) * P.bps;
    int ldsRow = (srcBytes + 15) & ~15;
    int maxBlkX = 0;
    for (int i = 0; i < P.nLevels; i++) if (P.lv[i].nBlkX > maxBlkX) maxBlkX = P.lv[i].nBlkX;
    const int histBins = 1024;
    // LDS of a chain: [source block | previous block row's results, 16 B per block | histogram of the global-motion estimate]
    const int ldsHist = ldsRow + maxBlkX * 16;
    int ldsBytes = ldsHist + histBins * 4;
    if (ldsBytes > 160 * 1024) { mvx_set_error("mvx_analyse_frames: frame too wide for the LDS row buffer"); return MVX_E_ARG; }
    // One-chain-per-workgroup launches (generic kernels, MVX_CPW=1, window / tile modes): asking for a little more than a fifth of
    // the CU's 160 KiB of LDS makes the dispatcher spread the chains four per CU instead of stacking some CUs (+5 % at 1008 chains).
    // The multi-chain workgroups are sized from ldsNeed instead.
    const int ldsNeed = ldsBytes;
    {
        int v = 33 * 1024;
        if (g_dbg.lds_min >= 0) v = g_dbg.lds_min; // developer override
        if (v > ldsBytes && v <= 160 * 1024) ldsBytes = v;
    }
    // a workgroup's chains share the CU's 160 KiB of LDS: very wide frames (long row buffers) get fewer chains per workgroup
    while (cpw > 1 && (long long)((ldsNeed + 255) & ~255) * cpw > 160 * 1024) {
        cpw = cpw == 12 ? 8 : cpw == 8 ? 4 : 1;
        if (wpe == 3) wpe = 2;
        if (P.bps == 2) wpe = 1; // (the 16-bit two-per-SIMD builds exist for eight chains per workgroup only)
    }
    // barrier between the chains of a workgroup every that many blocks (power of two) and at every row start: keeps the chains
    // of a two-per-SIMD workgroup on neighbouring blocks (+2 % at 4K16 for any interval from 64 blocks to a row, +2.8 % at 1080p
    // 8-bit); with one chain per SIMD it costs 1 %
    int syncEvery = wpe >= 2 ? 256 : 0;
    if (g_dbg.cpw_sync >= 0 && (g_dbg.cpw_sync & (g_dbg.cpw_sync - 1)) == 0) syncEvery = g_dbg.cpw_sync;
    ALaunch L = { njobs, ldsBytes, ldsRow, ldsHist, histBins, ldsNeed, simds, cpw, wpe, syncEvery, 0, 0, st, a->dP, S.d };
    // the specialised kernels address the reference as "64-bit base + 32-bit offset inside the level's plane set" (all sub-pel planes)
    bool off32 = true;
    for (int i = 0; i < P.nLevels; i++)
        if ((long long)P.lv[i].pel * P.lv[i].pel * P.lv[i].pstride[0] >= 0xffffffffLL || (long long)P.lv[i].pel * P.lv[i].pel * P.lv[i].pstride[1] >= 0xffffffffLL) off32 = false;
    int rc = (P.dctmode != 0 || !off32) ? 1 : P.bps == 1 ? mvx_analyse_launch_u8(P, L) : mvx_analyse_launch_u16(P, L); // specialised 4:2:0 geometries (SAD cost only)
    if (rc == 1) rc = mvx_analyse_launch_any(P, L);                                     // everything else
    if (rc) return rc;
    g_lastLaunch[0] = 0; g_lastLaunch[1] = cpw; g_lastLaunch[2] = syncEvery; g_lastLaunch[3] = njobs; g_lastLaunch[4] = 0;
    if (P.divide) hipLaunchKernelGGL(analyse_divide_kernel, dim3((P.lv[0].nBlkX * P.lv[0].nBlkY + 255) / 256, njobs), dim3(256), 0, st, a->dP, S.d);
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}

// ================================================================================================ mv.Recalculate
// MVRecalculate.c:263-545 (creation), :104-228 (frame); GroupOfPlanes.c:127-148; PlaneOfBlocks.cpp:1158-1424

struct mvx_recalculate {
    mvx_analysis_data ad, adOut, old;
    AParams P;
    RParams R;
    AParams *dP = nullptr;
    RParams *dR = nullptr;
    AJob *dJobs = nullptr;
    size_t jobsCap = 0;
    CallGuard guard;
};

#define RFAIL(...) do { snprintf(err, MVX_ERRLEN, __VA_ARGS__); mvx_set_error("%s", err); return MVX_E_ARG; } while (0)

extern "C" __attribute__((visibility("default"))) int mvx_recalculate_create(const mvx_recalculate_args *a, const mvx_super *sup, const mvx_analysis_data *vectors,
                                                                             const ptrdiff_t super_pitch[3], mvx_recalculate **out, char *err) {
    char dummy[MVX_ERRLEN];
    if (!err) err = dummy;
    err[0] = 0;
    *out = nullptr;
    const mvx_super_info &si = sup->info;
    auto A64 = [](int64_t v, int64_t d) { return v == MVX_UNSET ? d : v; };
    mvx_analysis_data ad;
    memset(&ad, 0, sizeof(ad));
    AParams P;
    memset(&P, 0, sizeof(P));
    long long thSAD = A64(a->thsad, 200);
    const int smooth = (int)A64(a->smooth, 1);
    ad.nBlkSizeX = (int)A64(a->blksize, 8);
    ad.nBlkSizeY = (int)A64(a->blksizev, ad.nBlkSizeX);
    P.searchType = (int)A64(a->search, SearchHex2);
    const int searchparam = (int)A64(a->searchparam, 2);
    int chroma = !!A64(a->chroma, 1);
    const int truemotion = !!A64(a->truemotion, 1);
    int nLambda = (int)A64(a->lambda, truemotion ? (1000 * ad.nBlkSizeX * ad.nBlkSizeY / 64) : 0);
    P.pnew = (int)A64(a->pnew, truemotion ? 50 : 0);
    ad.nOverlapX = (int)A64(a->overlap, 0);
    ad.nOverlapY = (int)A64(a->overlapv, ad.nOverlapX);
    P.dctmode = (int)A64(a->dct, 0);
    const int divide = (int)A64(a->divide, 0);
    P.meander = !!A64(a->meander, 1);
    // fields: the shift MVRecalculate.c:171-175 derives only feeds zeroMVfieldShifted / globalMVPredictor (PlaneOfBlocks.cpp:1167-1171),
    // which the recalculation never reads (ONLY_CHECK_NONDEFAULT_MV is not defined, PlaneOfBlocks.h:37): accepted, no effect on the blob.
    if (P.searchType < 0 || P.searchType > 7) RFAIL("Recalculate: search must be between 0 and 7 (inclusive).");
    if (P.dctmode < 0 || P.dctmode > 10) RFAIL("Recalculate: dct must be between 0 and 10 (inclusive).");
    if (P.dctmode >= 5 && ad.nBlkSizeX == 16 && ad.nBlkSizeY == 2) RFAIL("Recalculate: dct 5..10 cannot work with 16x2 blocks.");
    if (P.dctmode >= 1 && P.dctmode <= 4) RFAIL("Recalculate: dct 1..4 (FFTW3 DCT cost) are not implemented on the GPU path.");
    if (divide < 0 || divide > 2) RFAIL("Recalculate: divide must be between 0 and 2 (inclusive).");
    {
        static const int okb[12][2] = { { 4, 4 }, { 8, 4 }, { 8, 8 }, { 16, 2 }, { 16, 8 }, { 16, 16 }, { 32, 16 }, { 32, 32 }, { 64, 32 }, { 64, 64 }, { 128, 64 }, { 128, 128 } };
        bool found = false;
        for (auto &b : okb) found |= (ad.nBlkSizeX == b[0] && ad.nBlkSizeY == b[1]);
        if (!found) RFAIL("Recalculate: the block size must be 4x4, 8x4, 8x8, 16x2, 16x8, 16x16, 32x16, 32x32, 64x32, 64x64, 128x64, or 128x128.");
    }
    if (P.pnew < 0 || P.pnew > 256) RFAIL("Recalculate: pnew must be between 0 and 256 (inclusive).");
    if (ad.nOverlapX < 0 || ad.nOverlapX > ad.nBlkSizeX / 2 || ad.nOverlapY < 0 || ad.nOverlapY > ad.nBlkSizeY / 2)
        RFAIL("Recalculate: overlap must be at most half of blksize, overlapv must be at most half of blksizev, and they both need to be at least 0.");
    if (divide && (ad.nBlkSizeX < 8 || ad.nBlkSizeY < 8)) RFAIL("Recalculate: blksize and blksizev must be at least 8 when divide=True.");
    if (P.searchType == SearchNstep) P.nSearchParam = searchparam < 0 ? 0 : searchparam;
    else P.nSearchParam = searchparam < 1 ? 1 : searchparam;
    if (ad.nOverlapX % si.xRatioUV || ad.nOverlapY % si.yRatioUV) RFAIL("Recalculate: The requested overlap is incompatible with the super clip's subsampling.");
    if (divide && (ad.nOverlapX % (2 * si.xRatioUV) || ad.nOverlapY % (2 * si.yRatioUV)))
        RFAIL("Recalculate: overlap and overlapv must be multiples of 2 or 4 when divide=True, depending on the super clip's subsampling.");
    if (si.gray) chroma = 0;
    const int nModeYUV = chroma ? 7 : 1;
    if ((nModeYUV & si.modeYUV) != nModeYUV) RFAIL("Recalculate: super clip does not contain needed colour data.");
    ad.yRatioUV = vectors->yRatioUV; ad.xRatioUV = vectors->xRatioUV;
    ad.nWidth = vectors->nWidth; ad.nHeight = vectors->nHeight;
    ad.nDeltaFrame = vectors->nDeltaFrame; ad.isBackward = vectors->isBackward;
    ad.bitsPerSample = si.bits;
    const int pixelMax = (1 << si.bits) - 1;
    thSAD = (long long)((double)thSAD * pixelMax / 255.0 + 0.5);
    nLambda = (int)((double)nLambda * pixelMax / 255.0 + 0.5);
    thSAD = thSAD * (ad.nBlkSizeX * ad.nBlkSizeY) / 64;
    if (chroma) thSAD += thSAD / (ad.xRatioUV * ad.yRatioUV) * 2;
    ad.nMotionFlags = MOTION_USE_SIMD | (ad.isBackward ? MOTION_IS_BACKWARD : 0) | (chroma ? MOTION_USE_CHROMA_MOTION : 0);
    ad.nPel = si.pel;
    if (si.height != ad.nHeight || si.super_width - 2 * si.hpad != ad.nWidth) RFAIL("Recalculate: wrong frame size.");
    if (ad.xRatioUV != si.xRatioUV || ad.yRatioUV != si.yRatioUV) RFAIL("Recalculate: wrong frame size.");
    ad.nHPadding = si.hpad; ad.nVPadding = si.vpad;
    ad.nBlkX = (ad.nWidth - ad.nOverlapX) / (ad.nBlkSizeX - ad.nOverlapX);
    ad.nBlkY = (ad.nHeight - ad.nOverlapY) / (ad.nBlkSizeY - ad.nOverlapY);
    ad.nLvCount = 1;

    P.nLevels = 1;
    P.blkX = ad.nBlkSizeX; P.blkY = ad.nBlkSizeY; P.ovX = ad.nOverlapX; P.ovY = ad.nOverlapY;
    P.xr = si.xRatioUV; P.yr = si.yRatioUV; P.logxr = mvx_ilog2(P.xr); P.logyr = mvx_ilog2(P.yr);
    P.bits = si.bits; P.bps = (si.bits + 7) / 8; P.chroma = chroma;
    P.lambda = nLambda;
    P.verybigSAD = (long long)P.blkX * P.blkY * (1 << si.bits);
    P.superHPad = si.hpad; P.superVPad = si.vpad;
    for (int p = 0; p < 3; p++) P.pitch[p] = p < si.num_planes ? super_pitch[p] : 0;
    if (si.num_planes > 1 && super_pitch[1] != super_pitch[2]) RFAIL("Recalculate: the U and V planes of the super clip must share one pitch.");
    {
        ALevel &L = P.lv[0];
        L.nBlkX = ad.nBlkX; L.nBlkY = ad.nBlkY; L.pel = ad.nPel; L.logPel = mvx_ilog2(L.pel);
        LevelPlane y, c;
        mvx_level_plane(si, 0, 0, P.pitch[0], &y);
        L.pw = y.pw; L.ph = y.ph; L.hpad = y.hpad; L.vpad = y.vpad;
        L.off[0] = y.off; L.pstride[0] = P.pitch[0] * y.ph;
        if (si.num_planes > 1) {
            mvx_level_plane(si, 0, 1, P.pitch[1], &c);
            L.cpw = c.pw; L.cph = c.ph; L.chpad = c.hpad; L.cvpad = c.vpad;
            L.off[1] = c.off; L.pstride[1] = P.pitch[1] * c.ph;
            mvx_level_plane(si, 0, 2, P.pitch[2], &c);
            L.off[2] = c.off; L.pstride[2] = P.pitch[2] * c.ph;
        }
        L.blobOff = 8;
    }
    P.divide = divide;
    P.blobSize = 8 + 4 + ad.nBlkX * ad.nBlkY * 16 + (divide ? 4 + ad.nBlkX * ad.nBlkY * 64 : 0);
    mvx_recalculate *h = new mvx_recalculate();
    h->ad = ad; h->adOut = ad; h->old = *vectors; h->P = P;
    if (divide) mvx_divided_data(&ad, &h->adOut);
    RParams &R = h->R;
    R.nBlkX = vectors->nBlkX; R.nBlkY = vectors->nBlkY; R.blkX = vectors->nBlkSizeX; R.blkY = vectors->nBlkSizeY;
    R.stepX = vectors->nBlkSizeX - vectors->nOverlapX; R.stepY = vectors->nBlkSizeY - vectors->nOverlapY;
    R.logPel = mvx_ilog2(vectors->nPel); R.nLvCount = vectors->nLvCount;
    R.thSAD = thSAD; R.smooth = smooth;
    *out = h;
    return MVX_OK;
}

extern "C" __attribute__((visibility("default"))) void mvx_recalculate_destroy(mvx_recalculate *r) {
    if (!r) return;
    if (r->dP) (void)hipFree(r->dP);
    if (r->dR) (void)hipFree(r->dR);
    if (r->dJobs) (void)hipFree(r->dJobs);
    delete r;
}
extern "C" __attribute__((visibility("default"))) void mvx_recalculate_get_data(const mvx_recalculate *r, mvx_analysis_data *out) { *out = r->adOut; }
extern "C" __attribute__((visibility("default"))) int mvx_recalculate_blob_size(const mvx_recalculate *r) { return r->P.blobSize; }

extern "C" __attribute__((visibility("default"))) int mvx_recalculate_frames(mvx_recalculate *r, int njobs, const mvx_recalculate_job *jobs, void *stream) {
    if (njobs <= 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    CallGuard::Scope scope(r->guard, st);
    const AParams &P = r->P;
    if (!r->dP) {
        HIP_CHECK(hipMalloc((void **)&r->dP, sizeof(AParams)));
        HIP_CHECK(hipMemcpy(r->dP, &P, sizeof(AParams), hipMemcpyHostToDevice));
        HIP_CHECK(hipMalloc((void **)&r->dR, sizeof(RParams)));
        HIP_CHECK(hipMemcpy(r->dR, &r->R, sizeof(RParams), hipMemcpyHostToDevice));
    }
    if ((size_t)njobs > r->jobsCap) {
        if (r->dJobs) (void)hipFree(r->dJobs);
        r->jobsCap = (size_t)njobs * 2;
        HIP_CHECK(hipMalloc((void **)&r->dJobs, r->jobsCap * sizeof(AJob)));
    }
    std::vector<AJob> hj(njobs);
    for (int i = 0; i < njobs; i++) {
        for (int p = 0; p < 3; p++) { hj[i].src[p] = (const unsigned char *)jobs[i].src[p]; hj[i].ref[p] = (const unsigned char *)jobs[i].ref[p]; }
        hj[i].blob = (unsigned char *)jobs[i].blob;
        hj[i].oldBlob = (const unsigned char *)jobs[i].old_blob;
        hj[i].fieldShift = 0;
        hj[i].valid = jobs[i].ref[0] != nullptr;
        if (!hj[i].oldBlob) { mvx_set_error("mvx_recalculate_frames: old_blob is required"); return MVX_E_ARG; }
        if ((((uintptr_t)jobs[i].blob) & 15) || (((uintptr_t)jobs[i].old_blob) & 15)) { mvx_set_error("mvx_recalculate_frames: blobs must be 16-byte aligned"); return MVX_E_ARG; }
    }
    HIP_CHECK(hipMemcpyAsync(r->dJobs, hj.data(), sizeof(AJob) * njobs, hipMemcpyHostToDevice, st));
    int srcBytes = P.blkX * P.blkY * P.bps;
    if (P.chroma) srcBytes += 2 * (P.blkX / P.xr) * (P.blkY / P.yr) * P.bps;
    const int ldsRow = (srcBytes + 15) & ~15; // only the source block lives in LDS here
    RLaunch L = { njobs, P.lv[0].nBlkX * P.lv[0].nBlkY, ldsRow + 64, ldsRow, ldsRow, 0, st, r->dP, r->dR, r->dJobs };
    int rc = mvx_recalc_launch(P, L);
    if (rc) return rc;
    if (P.divide) hipLaunchKernelGGL(analyse_divide_kernel, dim3((L.nBlk + 255) / 256, njobs), dim3(256), 0, st, r->dP, r->dJobs);
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}

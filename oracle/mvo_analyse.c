/*
 * mvo_analyse.c -- oracle restatement of mv.Analyse (test infrastructure only, see mvoracle.h).
 * Follows /root/reference/src/MVAnalyse.c, GroupOfPlanes.c, PlaneOfBlocks.cpp; dct=0 and the SATD
 * modes dct=5..10 (no FFTW modes 1..4: third-party fftw3f, unpinned, out of scope).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvo_internal.h"

#define VMAX(a, b) ((a) > (b) ? (a) : (b))
#define VMIN(a, b) ((a) > (b) ? (b) : (a))

enum { SearchOnetime, SearchNstep, SearchLogarithmic, SearchExhaustive, SearchHex2, SearchUMH, SearchHorizontal, SearchVertical };

#define MOTION_USE_SIMD 1
#define MOTION_IS_BACKWARD 2
#define MOTION_SMALLEST_PLANE 4
#define MOTION_USE_CHROMA_MOTION 8

static const mvo_vector zeroMV = { 0, 0, -1 }; /* MVAnalysisData.h:79 */

/* -DMVO_STATS (tools/search_stats.py builds its own copy of the library with it; the test library has none of this): how the default
 * search behaves on a clip -- which predictor a block's predictor phase ends on, how many different vectors the seven predictors are,
 * how often a hexagon point wins -- the numbers a kernel design wants to know before it speculates.  Counted per level. */
#ifdef MVO_STATS
enum { ST_BLOCKS, ST_WIN0, ST_WIN6 = ST_WIN0 + 6, ST_BEST_IS_PRED, ST_BEST_IS_MEDIAN, ST_BEST_IS_LEFT, ST_DISTINCT1, ST_DISTINCT7 = ST_DISTINCT1 + 6, ST_HEX_WON, ST_HEX_TRIED, ST_RESCUE,
       /* r4: what a kernel that evaluates the candidates that do NOT depend on the left neighbour ahead of the serial walk would find */
       ST_PRED_HIT, ST_CENTRE_UP, ST_CENTRE_AHEAD, ST_CENTRE_HIER, ST_CENTRE_ANY5, ST_FULL_UP, ST_FULL_ANY5, ST_LEFT_IN, ST_MEDIAN_IN, ST_N };
static long long g_stat[16][ST_N];
void mvo_stats_get(long long *out, int levels) { memcpy(out, g_stat, sizeof(long long) * ST_N * (size_t)(levels < 16 ? levels : 16)); }
void mvo_stats_reset(void) { memset(g_stat, 0, sizeof(g_stat)); }
int mvo_stats_fields(void) { return ST_N; }
#define STAT(lvl, k, n) (g_stat[(lvl) & 15][k] += (n))
#else
#define STAT(lvl, k, n) ((void)0)
#endif

/* ------------------------------------------------------------------ SAD / SATD / luma */

#define DEFINE_SAD(T, SFX)                                                                                   \
    /* SADFunctions.cpp:353-367 sad_c */                                                                     \
    static unsigned sad_##SFX(int w, int h, const uint8_t *s8, intptr_t sp, const uint8_t *r8, intptr_t rp) { \
        unsigned sum = 0;                                                                                    \
        for (int y = 0; y < h; y++) {                                                                        \
            const T *s = (const T *)s8; const T *r = (const T *)r8;                                          \
            for (int x = 0; x < w; x++) { int d = (int)s[x] - (int)r[x]; sum += (unsigned)(d < 0 ? -d : d); } \
            s8 += sp; r8 += rp;                                                                              \
        }                                                                                                    \
        return sum;                                                                                          \
    }                                                                                                        \
    /* sum |H4 * D * H4^T| over one 4x4 block: SADFunctions.cpp:581-637 (the pseudo-SIMD packing there is    \
       an exact emulation of this plain form) */                                                             \
    static unsigned had4x4_##SFX(const uint8_t *s8, intptr_t sp, const uint8_t *r8, intptr_t rp) {           \
        int t[4][4];                                                                                         \
        for (int i = 0; i < 4; i++) {                                                                        \
            const T *s = (const T *)s8; const T *r = (const T *)r8;                                          \
            int a0 = s[0] - r[0], a1 = s[1] - r[1], a2 = s[2] - r[2], a3 = s[3] - r[3];                      \
            int t0 = a0 + a1, t1 = a0 - a1, t2 = a2 + a3, t3 = a2 - a3;                                      \
            t[i][0] = t0 + t2; t[i][2] = t0 - t2; t[i][1] = t1 + t3; t[i][3] = t1 - t3;                      \
            s8 += sp; r8 += rp;                                                                              \
        }                                                                                                    \
        unsigned sum = 0;                                                                                    \
        for (int i = 0; i < 4; i++) {                                                                        \
            int t0 = t[0][i] + t[1][i], t1 = t[0][i] - t[1][i], t2 = t[2][i] + t[3][i], t3 = t[2][i] - t[3][i]; \
            int d0 = t0 + t2, d2 = t0 - t2, d1 = t1 + t3, d3 = t1 - t3;                                      \
            sum += (unsigned)abs(d0) + (unsigned)abs(d1) + (unsigned)abs(d2) + (unsigned)abs(d3);            \
        }                                                                                                    \
        return sum;                                                                                          \
    }                                                                                                        \
    /* SADFunctions.cpp:686-710 Satd_C: one 4x4, else 8x4 partitions each >> 1 */                            \
    static unsigned satd_##SFX(int w, int h, const uint8_t *s8, intptr_t sp, const uint8_t *r8, intptr_t rp) { \
        if (w == 4 && h == 4) return had4x4_##SFX(s8, sp, r8, rp) >> 1;                                      \
        unsigned sum = 0;                                                                                    \
        for (int y = 0; y < h; y += 4) {                                                                     \
            for (int x = 0; x < w; x += 8)                                                                   \
                sum += (had4x4_##SFX(s8 + x * sizeof(T), sp, r8 + x * sizeof(T), rp) +                       \
                        had4x4_##SFX(s8 + (x + 4) * sizeof(T), sp, r8 + (x + 4) * sizeof(T), rp)) >> 1;      \
            s8 += sp * 4; r8 += rp * 4;                                                                      \
        }                                                                                                    \
        return sum;                                                                                          \
    }                                                                                                        \
    /* Luma.cpp:14-25 luma_c */                                                                              \
    static unsigned luma_##SFX(int w, int h, const uint8_t *s8, intptr_t sp) {                               \
        unsigned sum = 0;                                                                                    \
        for (int y = 0; y < h; y++) { const T *s = (const T *)s8; for (int x = 0; x < w; x++) sum += s[x]; s8 += sp; } \
        return sum;                                                                                          \
    }

DEFINE_SAD(uint8_t, u8)
DEFINE_SAD(uint16_t, u16)

unsigned mvo_sad(int w, int h, int bits, const uint8_t *s, intptr_t sp, const uint8_t *r, intptr_t rp) {
    return bits <= 8 ? sad_u8(w, h, s, sp, r, rp) : sad_u16(w, h, s, sp, r, rp);
}
unsigned mvo_satd(int w, int h, int bits, const uint8_t *s, intptr_t sp, const uint8_t *r, intptr_t rp) {
    return bits <= 8 ? satd_u8(w, h, s, sp, r, rp) : satd_u16(w, h, s, sp, r, rp);
}

/* ------------------------------------------------------------------ PlaneOfBlocks state (PlaneOfBlocks.h:39-135) */

typedef struct pob {
    int nBlkX, nBlkY, nBlkSizeX, nBlkSizeY, nBlkCount, nPel, nLogPel, nScale, nLogScale, nOverlapX, nOverlapY;
    int xRatioUV, yRatioUV, nLogxRatioUV, nLogyRatioUV, bytesPerSample, bits;
    mvo_vector *vectors;
    int smallestPlane, chroma;
    const mvo_frame *pSrcFrame, *pRefFrame;
    int nSrcPitch[3]; const uint8_t *pSrc[3]; int nRefPitch[3];
    mvo_vector bestMV; int64_t nMinCost; mvo_vector predictor; mvo_vector predictors[5];
    int nDxMin, nDyMin, nDxMax, nDyMax;
    int x[3], y[3], blkx, blky, blkIdx, blkScanDir;
    int searchType, nSearchParam; int64_t nLambda, LSAD; int penaltyNew, penaltyZero, pglobal;
    int64_t badSAD; int badrange, badcount, tryMany;
    mvo_vector globalMVPredictor, zeroMVfieldShifted;
    int dctmode, srcLuma, refLuma, sumLumaChange, dctweight16;
    int64_t verybigSAD;
    int nSrcPitch_temp[3]; uint8_t *pSrc_temp[3];
} pob;

/* PlaneOfBlocks.cpp:320-397 pobInit */
static void pob_init(pob *p, int nBlkX, int nBlkY, int bx, int by, int pel, int level, int flags, int ox, int oy, int xr, int yr, int bits) {
    memset(p, 0, sizeof(*p));
    p->nPel = pel; p->nLogPel = mvo_ilog2(pel); p->nLogScale = level; p->nScale = 1 << (level > 0 ? level : 0);
    p->nBlkSizeX = bx; p->nBlkSizeY = by; p->nOverlapX = ox; p->nOverlapY = oy;
    p->nBlkX = nBlkX; p->nBlkY = nBlkY; p->nBlkCount = nBlkX * nBlkY;
    p->xRatioUV = xr; p->yRatioUV = yr; p->nLogxRatioUV = mvo_ilog2(xr); p->nLogyRatioUV = mvo_ilog2(yr);
    p->bits = bits; p->bytesPerSample = (bits + 7) / 8;
    p->smallestPlane = !!(flags & MOTION_SMALLEST_PLANE);
    p->chroma = !!(flags & MOTION_USE_CHROMA_MOTION);
    p->globalMVPredictor = zeroMV;
    p->vectors = (mvo_vector *)calloc(p->nBlkCount, sizeof(mvo_vector)); /* memset 0: :355 */
    p->nSrcPitch_temp[0] = bx * p->bytesPerSample;
    p->nSrcPitch_temp[1] = p->nSrcPitch_temp[2] = bx / xr * p->bytesPerSample;
    p->pSrc_temp[0] = (uint8_t *)malloc(by * p->nSrcPitch_temp[0] + 4);
    p->pSrc_temp[1] = (uint8_t *)malloc(by / yr * p->nSrcPitch_temp[1] + 4);
    p->pSrc_temp[2] = (uint8_t *)malloc(by / yr * p->nSrcPitch_temp[2] + 4);
    p->verybigSAD = (int64_t)bx * by * (1 << bits);
}

static void pob_deinit(pob *p) {
    free(p->vectors); free(p->pSrc_temp[0]); free(p->pSrc_temp[1]); free(p->pSrc_temp[2]);
}

/* PlaneOfBlocks.cpp:35-101 */
static const uint8_t *ref_block(pob *p, int plane, int vx, int vy) {
    if (plane == 0)
        return mvo_plane_abs_pointer(&p->pRefFrame->pl[0], p->nLogPel, (p->x[0] << p->nLogPel) + vx, (p->y[0] << p->nLogPel) + vy);
    int xbias = (vx < 0) * ((1 << p->nLogxRatioUV) - 1);
    int ybias = (vy < 0) * ((1 << p->nLogyRatioUV) - 1);
    return mvo_plane_abs_pointer(&p->pRefFrame->pl[plane], p->nLogPel,
                                 (p->x[plane] << p->nLogPel) + ((vx + xbias) >> p->nLogxRatioUV),
                                 (p->y[plane] << p->nLogPel) + ((vy + ybias) >> p->nLogyRatioUV));
}

static unsigned blk_sad(pob *p, int plane, const uint8_t *ref) {
    int w = plane ? p->nBlkSizeX / p->xRatioUV : p->nBlkSizeX, h = plane ? p->nBlkSizeY / p->yRatioUV : p->nBlkSizeY;
    return p->bytesPerSample == 1 ? sad_u8(w, h, p->pSrc[plane], p->nSrcPitch[plane], ref, p->nRefPitch[plane])
                                  : sad_u16(w, h, p->pSrc[plane], p->nSrcPitch[plane], ref, p->nRefPitch[plane]);
}
static unsigned blk_satd(pob *p, const uint8_t *ref) {
    return p->bytesPerSample == 1 ? satd_u8(p->nBlkSizeX, p->nBlkSizeY, p->pSrc[0], p->nSrcPitch[0], ref, p->nRefPitch[0])
                                  : satd_u16(p->nBlkSizeX, p->nBlkSizeY, p->pSrc[0], p->nSrcPitch[0], ref, p->nRefPitch[0]);
}
static int blk_luma(pob *p, const uint8_t *ptr, int pitch) {
    return (int)(p->bytesPerSample == 1 ? luma_u8(p->nBlkSizeX, p->nBlkSizeY, ptr, pitch) : luma_u16(p->nBlkSizeX, p->nBlkSizeY, ptr, pitch));
}

/* PlaneOfBlocks.cpp:117-203 pobLumaSAD (dct modes 1-4 need FFTW: rejected at init) */
static int64_t luma_sad(pob *p, const uint8_t *ref0) {
    int64_t sad = 0;
    int m = p->dctmode;
    if (m == 0) return blk_sad(p, 0, ref0);
    if (m == 5) return blk_satd(p, ref0);
    if (m == 6) {
        sad = blk_sad(p, 0, ref0);
        if (p->dctweight16 > 0) { int64_t d = blk_satd(p, ref0); sad = (sad * (16 - p->dctweight16) + d * p->dctweight16) / 16; }
    } else if (m == 7 || m == 8 || m == 10) {
        p->refLuma = blk_luma(p, ref0, p->nRefPitch[0]);
        sad = blk_sad(p, 0, ref0);
        int sh = m == 10 ? 4 : 5;
        if (abs(p->srcLuma - p->refLuma) > ((p->srcLuma + p->refLuma) >> sh)) {
            int64_t d = blk_satd(p, ref0);
            if (m == 7) sad = sad / 2 + d / 2;
            else if (m == 8) sad = sad / 4 + d / 2 + d / 4;
            else sad = sad / 2 + d / 4 + sad / 4;
        }
    } else if (m == 9) {
        sad = blk_sad(p, 0, ref0);
        if (p->dctweight16 > 1) { int h = p->dctweight16 / 2; int64_t d = blk_satd(p, ref0); sad = (sad * (16 - h) + d * h) / 16; }
    }
    return sad;
}

/* PlaneOfBlocks.cpp:105-114: unsigned square norm, int dist, (int) of the int64 product >> 8 */
static int motion_distortion(pob *p, int vx, int vy) {
    unsigned u = (unsigned)((p->predictor.x - vx) * (p->predictor.x - vx) + (p->predictor.y - vy) * (p->predictor.y - vy));
    int dist = (int)u;
    return (int)((p->nLambda * dist) >> 8);
}

static int vector_ok(pob *p, int vx, int vy) { /* :207-212 */
    return vx >= p->nDxMin && vy >= p->nDyMin && vx < p->nDxMax && vy < p->nDyMax;
}

#define F_PNEW 2
#define F_DIR 4
#define F_BEST 8

/* PlaneOfBlocks.cpp:219-261 pobCheckMV_Template */
static void check_mv(pob *p, int flags, int vx, int vy, int *dir, int val) {
    if (!vector_ok(p, vx, vy)) return;
    int64_t cost = motion_distortion(p, vx, vy);
    if (cost >= p->nMinCost) return;
    int64_t sad = luma_sad(p, ref_block(p, 0, vx, vy));
    cost += sad + ((flags & F_PNEW) ? ((p->penaltyNew * sad) >> 8) : 0);
    if (cost >= p->nMinCost) return;
    int64_t saduv = 0;
    if (p->chroma) {
        saduv += blk_sad(p, 1, ref_block(p, 1, vx, vy));
        saduv += blk_sad(p, 2, ref_block(p, 2, vx, vy));
        cost += saduv + ((flags & F_PNEW) ? ((p->penaltyNew * saduv) >> 8) : 0);
        if (cost >= p->nMinCost) return;
    }
    if (flags & F_BEST) { p->bestMV.x = vx; p->bestMV.y = vy; }
    p->nMinCost = cost;
    p->bestMV.sad = sad + saduv;
    if (flags & F_DIR) *dir = val;
}
static void CheckMV0(pob *p, int vx, int vy) { check_mv(p, F_BEST, vx, vy, 0, 0); }                             /* :265-268 */
static void CheckMV(pob *p, int vx, int vy) { check_mv(p, F_PNEW | F_BEST, vx, vy, 0, 0); }                      /* :272-275 */
static void CheckMV2(pob *p, int vx, int vy, int *dir, int val) { check_mv(p, F_PNEW | F_DIR | F_BEST, vx, vy, dir, val); } /* :279-282 */
static void CheckMVdir(pob *p, int vx, int vy, int *dir, int val) { check_mv(p, F_PNEW | F_DIR, vx, vy, dir, val); }        /* :286-289 */

static mvo_vector clip_mv(pob *p, mvo_vector v) { /* :293-311 */
    mvo_vector r;
    r.x = VMIN(VMAX(v.x, p->nDxMin), p->nDxMax - 1);
    r.y = VMIN(VMAX(v.y, p->nDyMin), p->nDyMax - 1);
    r.sad = v.sad;
    return r;
}
static int median3(int a, int b, int c) { return VMAX(VMIN(a, b), VMIN(VMAX(a, b), c)); } /* :315-317 */

/* PlaneOfBlocks.cpp:419-463 pobFetchPredictors */
static void fetch_predictors(pob *p) {
    int d = p->blkScanDir;
    if ((d == 1 && p->blkx > 0) || (d == -1 && p->blkx < p->nBlkX - 1))
        p->predictors[1] = clip_mv(p, p->vectors[p->blkIdx - d]);
    else
        p->predictors[1] = clip_mv(p, p->zeroMVfieldShifted);
    if (p->blky > 0)
        p->predictors[2] = clip_mv(p, p->vectors[p->blkIdx - p->nBlkX]);
    else
        p->predictors[2] = clip_mv(p, p->zeroMVfieldShifted);
    int ahead = (d == 1 && p->blkx < p->nBlkX - 1) || (d == -1 && p->blkx > 0);
    if (p->blky < p->nBlkY - 1 && ahead)
        p->predictors[3] = clip_mv(p, p->vectors[p->blkIdx + p->nBlkX + d]);
    else if (p->blky > 0 && ahead)
        p->predictors[3] = clip_mv(p, p->vectors[p->blkIdx - p->nBlkX + d]);
    else
        p->predictors[3] = clip_mv(p, p->zeroMVfieldShifted);
    if (p->blky > 0) {
        p->predictors[0].x = median3(p->predictors[1].x, p->predictors[2].x, p->predictors[3].x);
        p->predictors[0].y = median3(p->predictors[1].y, p->predictors[2].y, p->predictors[3].y);
        p->predictors[0].sad = VMAX(p->predictors[1].sad, VMAX(p->predictors[2].sad, p->predictors[3].sad));
    } else
        p->predictors[0] = p->predictors[1];
    if (p->smallestPlane) p->predictor = p->predictors[0];
    double scale = p->LSAD / (double)(p->LSAD + (p->predictor.sad >> 1));
    p->nLambda = (int64_t)(p->nLambda * scale * scale);
}

/* search patterns: PlaneOfBlocks.cpp:466-816 */
static void NStepSearch(pob *p, int stp) { /* :467-485 */
    int length = stp;
    while (length > 0) {
        int dx = p->bestMV.x, dy = p->bestMV.y;
        CheckMV(p, dx + length, dy + length); CheckMV(p, dx + length, dy); CheckMV(p, dx + length, dy - length);
        CheckMV(p, dx, dy - length); CheckMV(p, dx, dy + length);
        CheckMV(p, dx - length, dy + length); CheckMV(p, dx - length, dy); CheckMV(p, dx - length, dy - length);
        length--;
    }
}

static void OneTimeSearch(pob *p, int length) { /* :489-527 */
    int direction = 0, dx = p->bestMV.x, dy = p->bestMV.y;
    CheckMV2(p, dx - length, dy, &direction, 2);
    CheckMV2(p, dx + length, dy, &direction, 1);
    if (direction == 1) {
        while (direction) { direction = 0; dx += length; CheckMV2(p, dx + length, dy, &direction, 1); }
    } else if (direction == 2) {
        while (direction) { direction = 0; dx -= length; CheckMV2(p, dx - length, dy, &direction, 1); }
    }
    CheckMV2(p, dx, dy - length, &direction, 2);
    CheckMV2(p, dx, dy + length, &direction, 1);
    if (direction == 1) {
        while (direction) { direction = 0; dy += length; CheckMV2(p, dx, dy + length, &direction, 1); }
    } else if (direction == 2) {
        while (direction) { direction = 0; dy -= length; CheckMV2(p, dx, dy - length, &direction, 1); }
    }
}

static void DiamondSearch(pob *p, int length) { /* :531-632 */
    enum { Right = 1, Left = 2, Down = 4, Up = 8 };
    int dx, dy, direction = 15, last;
    while (direction > 0) {
        dx = p->bestMV.x; dy = p->bestMV.y; last = direction; direction = 0;
        if (last & Right) CheckMV2(p, dx + length, dy, &direction, Right);
        if (last & Left) CheckMV2(p, dx - length, dy, &direction, Left);
        if (last & Down) CheckMV2(p, dx, dy + length, &direction, Down);
        if (last & Up) CheckMV2(p, dx, dy - length, &direction, Up);
        if (direction) {
            last = direction; dx = p->bestMV.x; dy = p->bestMV.y;
            if (last & (Right + Left)) {
                CheckMV2(p, dx, dy + length, &direction, Down); CheckMV2(p, dx, dy - length, &direction, Up);
            } else {
                CheckMV2(p, dx + length, dy, &direction, Right); CheckMV2(p, dx - length, dy, &direction, Left);
            }
        } else {
            switch (last) {
            case Right: CheckMV2(p, dx + length, dy + length, &direction, Right + Down); CheckMV2(p, dx + length, dy - length, &direction, Right + Up); break;
            case Left: CheckMV2(p, dx - length, dy + length, &direction, Left + Down); CheckMV2(p, dx - length, dy - length, &direction, Left + Up); break;
            case Down: CheckMV2(p, dx + length, dy + length, &direction, Right + Down); CheckMV2(p, dx - length, dy + length, &direction, Left + Down); break;
            case Up: CheckMV2(p, dx + length, dy - length, &direction, Right + Up); CheckMV2(p, dx - length, dy - length, &direction, Left + Up); break;
            case Right + Down:
                CheckMV2(p, dx + length, dy + length, &direction, Right + Down); CheckMV2(p, dx - length, dy + length, &direction, Left + Down);
                CheckMV2(p, dx + length, dy - length, &direction, Right + Up); break;
            case Left + Down:
                CheckMV2(p, dx + length, dy + length, &direction, Right + Down); CheckMV2(p, dx - length, dy + length, &direction, Left + Down);
                CheckMV2(p, dx - length, dy - length, &direction, Left + Up); break;
            case Right + Up:
                CheckMV2(p, dx + length, dy + length, &direction, Right + Down); CheckMV2(p, dx - length, dy - length, &direction, Left + Up);
                CheckMV2(p, dx + length, dy - length, &direction, Right + Up); break;
            case Left + Up:
                CheckMV2(p, dx - length, dy - length, &direction, Left + Up); CheckMV2(p, dx - length, dy + length, &direction, Left + Down);
                CheckMV2(p, dx + length, dy - length, &direction, Right + Up); break;
            default:
                CheckMV2(p, dx + length, dy + length, &direction, Right + Down); CheckMV2(p, dx - length, dy + length, &direction, Left + Down);
                CheckMV2(p, dx + length, dy - length, &direction, Right + Up); CheckMV2(p, dx - length, dy - length, &direction, Left + Up); break;
            }
        }
    }
}

static void ExpandingSearch(pob *p, int r, int s, int mvx, int mvy) { /* :636-658 */
    for (int i = -r + s; i < r; i += s) { CheckMV(p, mvx + i, mvy - r); CheckMV(p, mvx + i, mvy + r); }
    for (int j = -r + s; j < r; j += s) { CheckMV(p, mvx - r, mvy + j); CheckMV(p, mvx + r, mvy + j); }
    CheckMV(p, mvx - r, mvy - r); CheckMV(p, mvx - r, mvy + r); CheckMV(p, mvx + r, mvy - r); CheckMV(p, mvx + r, mvy + r);
}

static const int mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };                                                   /* :662 */
static const int hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } }; /* :664 */

static void Hex2Search(pob *p, int i_me_range) { /* :667-724 */
    int dir = -2, bmx = p->bestMV.x, bmy = p->bestMV.y;
    if (i_me_range > 1) {
        CheckMVdir(p, bmx - 2, bmy, &dir, 0); CheckMVdir(p, bmx - 1, bmy + 2, &dir, 1); CheckMVdir(p, bmx + 1, bmy + 2, &dir, 2);
        CheckMVdir(p, bmx + 2, bmy, &dir, 3); CheckMVdir(p, bmx + 1, bmy - 2, &dir, 4); CheckMVdir(p, bmx - 1, bmy - 2, &dir, 5);
        STAT(p->nLogScale, ST_HEX_TRIED, 1); STAT(p->nLogScale, ST_HEX_WON, dir != -2);
        if (dir != -2) {
            bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            for (int i = 1; i < i_me_range / 2 && vector_ok(p, bmx, bmy); i++) {
                const int odir = mod6m1[dir + 1];
                dir = -2;
                CheckMVdir(p, bmx + hex2[odir + 0][0], bmy + hex2[odir + 0][1], &dir, odir - 1);
                CheckMVdir(p, bmx + hex2[odir + 1][0], bmy + hex2[odir + 1][1], &dir, odir);
                CheckMVdir(p, bmx + hex2[odir + 2][0], bmy + hex2[odir + 2][1], &dir, odir + 1);
                if (dir == -2) break;
                bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            }
        }
        p->bestMV.x = bmx; p->bestMV.y = bmy;
    }
    ExpandingSearch(p, 1, 1, bmx, bmy);
}

static void CrossSearch(pob *p, int start, int x_max, int y_max, int mvx, int mvy) { /* :728-739 */
    for (int i = start; i < x_max; i += 2) { CheckMV(p, mvx - i, mvy); CheckMV(p, mvx + i, mvy); }
    for (int j = start; j < y_max; j += 2) { CheckMV(p, mvx, mvy - j); CheckMV(p, mvx, mvy + j); }
}

static void UMHSearch(pob *p, int i_me_range, int omx, int omy) { /* :743-769 */
    static const int hex4[16][2] = { { -4, 2 }, { -4, 1 }, { -4, 0 }, { -4, -1 }, { -4, -2 }, { 4, -2 }, { 4, -1 }, { 4, 0 },
                                     { 4, 1 }, { 4, 2 }, { 2, 3 }, { 0, 4 }, { -2, 3 }, { -2, -3 }, { 0, -4 }, { 2, -3 } };
    CrossSearch(p, 1, i_me_range, i_me_range, omx, omy);
    int i = 1;
    do {
        for (int j = 0; j < 16; j++) CheckMV(p, omx + hex4[j][0] * i, omy + hex4[j][1] * i);
    } while (++i <= i_me_range / 4);
    Hex2Search(p, i_me_range);
}

static void Refine(pob *p) { /* :773-816 */
    if (p->searchType == SearchOnetime) for (int i = p->nSearchParam; i > 0; i /= 2) OneTimeSearch(p, i);
    if (p->searchType == SearchNstep) NStepSearch(p, p->nSearchParam);
    if (p->searchType == SearchLogarithmic) for (int i = p->nSearchParam; i > 0; i /= 2) DiamondSearch(p, i);
    if (p->searchType == SearchExhaustive) {
        int mvx = p->bestMV.x, mvy = p->bestMV.y;
        for (int i = 1; i <= p->nSearchParam; i++) ExpandingSearch(p, i, 1, mvx, mvy);
    }
    if (p->searchType == SearchHex2) Hex2Search(p, p->nSearchParam);
    if (p->searchType == SearchUMH) UMHSearch(p, p->nSearchParam, p->bestMV.x, p->bestMV.y);
    if (p->searchType == SearchHorizontal) {
        int mvx = p->bestMV.x, mvy = p->bestMV.y;
        for (int i = 1; i <= p->nSearchParam; i++) { CheckMV(p, mvx - i, mvy); CheckMV(p, mvx + i, mvy); }
    }
    if (p->searchType == SearchVertical) {
        int mvx = p->bestMV.x, mvy = p->bestMV.y;
        for (int i = 1; i <= p->nSearchParam; i++) { CheckMV(p, mvx, mvy - i); CheckMV(p, mvx, mvy + i); }
    }
}

static int64_t full_sad(pob *p, int vx, int vy, int cvx, int cvy) {
    int64_t sad = luma_sad(p, ref_block(p, 0, vx, vy));
    if (p->chroma) { sad += blk_sad(p, 1, ref_block(p, 1, cvx, cvy)); sad += blk_sad(p, 2, ref_block(p, 2, cvx, cvy)); }
    return sad;
}

/* PlaneOfBlocks.cpp:819-968 pobPseudoEPZSearch */
static void PseudoEPZSearch(pob *p) {
    fetch_predictors(p);
    if (p->dctmode >= 3) p->srcLuma = blk_luma(p, p->pSrc[0], p->nSrcPitch[0]);

    p->bestMV.x = p->zeroMVfieldShifted.x; p->bestMV.y = p->zeroMVfieldShifted.y;
    int64_t sad = full_sad(p, 0, p->zeroMVfieldShifted.y, 0, 0); /* chroma uses (0,0): :836-839 */
    p->bestMV.sad = sad;
    p->nMinCost = sad + ((p->penaltyZero * sad) >> 8);

    mvo_vector bestMVMany[8]; int64_t nMinCostMany[8] = { 0 };
    if (p->tryMany) { Refine(p); bestMVMany[0] = p->bestMV; nMinCostMany[0] = p->nMinCost; }

    p->globalMVPredictor = clip_mv(p, p->globalMVPredictor); /* cumulative clip: :859 */
    sad = full_sad(p, p->globalMVPredictor.x, p->globalMVPredictor.y, p->globalMVPredictor.x, p->globalMVPredictor.y);
    int64_t cost = sad + ((p->pglobal * sad) >> 8);
    if (cost < p->nMinCost || p->tryMany) {
        p->bestMV.x = p->globalMVPredictor.x; p->bestMV.y = p->globalMVPredictor.y; p->bestMV.sad = sad; p->nMinCost = cost;
    }
    if (p->tryMany) { Refine(p); bestMVMany[1] = p->bestMV; nMinCostMany[1] = p->nMinCost; }

    sad = full_sad(p, p->predictor.x, p->predictor.y, p->predictor.x, p->predictor.y);
    cost = sad;
    if (cost < p->nMinCost || p->tryMany) {
        p->bestMV.x = p->predictor.x; p->bestMV.y = p->predictor.y; p->bestMV.sad = sad; p->nMinCost = cost;
    }
    if (p->tryMany) { Refine(p); bestMVMany[2] = p->bestMV; nMinCostMany[2] = p->nMinCost; }

    int npred = 4;
    for (int i = 0; i < npred; i++) {
        if (p->tryMany) p->nMinCost = p->verybigSAD + 1;
        CheckMV0(p, p->predictors[i].x, p->predictors[i].y);
        if (p->tryMany) { Refine(p); bestMVMany[i + 3] = p->bestMV; nMinCostMany[i + 3] = p->nMinCost; }
    }
    if (p->tryMany) {
        p->nMinCost = p->verybigSAD + 1;
        for (int i = 0; i < npred + 3; i++)
            if (nMinCostMany[i] < p->nMinCost) { p->bestMV = bestMVMany[i]; p->nMinCost = nMinCostMany[i]; }
    } else {
#ifdef MVO_STATS
        {
            const int lv = p->nLogScale;
            mvo_vector c[7] = { p->zeroMVfieldShifted, p->globalMVPredictor, p->predictor, p->predictors[0], p->predictors[1], p->predictors[2], p->predictors[3] };
            int win = 0, distinct = 0;
            for (int i = 6; i >= 0; i--) if (c[i].x == p->bestMV.x && c[i].y == p->bestMV.y) win = i; /* the first candidate with the winning vector */
            for (int i = 0; i < 7; i++) { int dup = 0; for (int j = 0; j < i; j++) dup |= c[j].x == c[i].x && c[j].y == c[i].y; distinct += !dup; }
            STAT(lv, ST_BLOCKS, 1); STAT(lv, ST_WIN0 + win, 1); STAT(lv, ST_DISTINCT1 + distinct - 1, 1);
            STAT(lv, ST_BEST_IS_PRED, p->bestMV.x == p->predictor.x && p->bestMV.y == p->predictor.y);
            STAT(lv, ST_BEST_IS_MEDIAN, p->bestMV.x == p->predictors[0].x && p->bestMV.y == p->predictors[0].y);
            STAT(lv, ST_BEST_IS_LEFT, p->bestMV.x == p->predictors[1].x && p->bestMV.y == p->predictors[1].y);
            {   /* the left-independent set: zero, global, hierarchical, up, ahead (c[0], c[1], c[2], c[5], c[6]) */
                static const int ind[5] = { 0, 1, 2, 5, 6 };
                int leftIn = 0, medIn = 0, bestIn = 0;
                for (int k = 0; k < 5; k++) {
                    const mvo_vector *q = &c[ind[k]];
                    if (k == 0 && p->zeroMVfieldShifted.y != 0) continue; /* (its chroma ignores the shift: not a plain vector) */
                    leftIn |= q->x == c[4].x && q->y == c[4].y;
                    medIn |= q->x == c[3].x && q->y == c[3].y;
                    bestIn |= q->x == p->bestMV.x && q->y == p->bestMV.y;
                }
                const int predHit = leftIn && medIn;
                const int up = p->bestMV.x == c[5].x && p->bestMV.y == c[5].y;
                STAT(lv, ST_LEFT_IN, leftIn); STAT(lv, ST_MEDIAN_IN, medIn); STAT(lv, ST_PRED_HIT, predHit);
                STAT(lv, ST_CENTRE_UP, up);
                STAT(lv, ST_CENTRE_AHEAD, p->bestMV.x == c[6].x && p->bestMV.y == c[6].y);
                STAT(lv, ST_CENTRE_HIER, p->bestMV.x == c[2].x && p->bestMV.y == c[2].y);
                STAT(lv, ST_CENTRE_ANY5, bestIn);
                STAT(lv, ST_FULL_UP, predHit && up); STAT(lv, ST_FULL_ANY5, predHit && bestIn);
            }
        }
#endif
        Refine(p);
    }

    int64_t foundSAD = p->bestMV.sad;
    if (p->blkIdx > 1 && foundSAD > (p->badSAD + p->badSAD * p->badcount / 16)) { /* :942 */
        STAT(p->nLogScale, ST_RESCUE, 1);
        p->badcount++;
        if (p->badrange > 0)
            UMHSearch(p, p->badrange * (1 << p->nLogPel), 0, 0);
        else if (p->badrange < 0) {
            for (int i = 1; i < -p->badrange * (1 << p->nLogPel); i += (1 << p->nLogPel)) {
                ExpandingSearch(p, i, 1 << p->nLogPel, 0, 0);
                if (p->bestMV.sad < foundSAD / 4) break;
            }
        }
        int mvx = p->bestMV.x, mvy = p->bestMV.y;
        for (int i = 1; i < (1 << p->nLogPel); i++) ExpandingSearch(p, i, 1, mvx, mvy);
    }
    p->vectors[p->blkIdx] = p->bestMV;
}

static void copy_block(uint8_t *d, int dp, const uint8_t *s, int sp, int wbytes, int h) { /* CopyCode.cpp:8-31 */
    for (int y = 0; y < h; y++) memcpy(d + (size_t)y * dp, s + (size_t)y * sp, wbytes);
}

/* PlaneOfBlocks.cpp:971-1131 doPobSearchMVs */
static void pob_search(pob *p, const mvo_frame *srcF, const mvo_frame *refF, int st, int stp, int lambda, int lsad, int pnew,
                       int plevel, uint8_t *out, mvo_vector *globalMVec, int fieldShift, int dctmode, int *pmeanLumaChange,
                       int pzero, int pglobal, int64_t badSAD, int badrange, int meander, int tryMany) {
    p->dctmode = dctmode;
    p->dctweight16 = VMIN(16, abs(*pmeanLumaChange) / (p->nBlkSizeX * p->nBlkSizeY));
    p->badSAD = badSAD; p->badrange = badrange;
    p->zeroMVfieldShifted.x = 0; p->zeroMVfieldShifted.y = fieldShift; p->zeroMVfieldShifted.sad = 0;
    p->globalMVPredictor.x = (1 << p->nLogPel) * globalMVec->x;
    p->globalMVPredictor.y = (1 << p->nLogPel) * globalMVec->y + fieldShift;
    p->globalMVPredictor.sad = globalMVec->sad;

    int size = (int)sizeof(int) + p->nBlkCount * (int)sizeof(mvo_vector); /* :413-416 */
    memcpy(out, &size, sizeof(size));
    mvo_vector *pBlkData = (mvo_vector *)(out + sizeof(int));

    p->pSrcFrame = srcF; p->pRefFrame = refF;
    const mvo_plane *s0 = &srcF->pl[0];
    p->y[0] = s0->vpad;
    if (srcF->mode & MVO_UPLANE) p->y[1] = srcF->pl[1].vpad;
    if (srcF->mode & MVO_VPLANE) p->y[2] = srcF->pl[2].vpad;
    p->nRefPitch[0] = refF->pl[0].pitch;
    if (p->chroma) { p->nRefPitch[1] = refF->pl[1].pitch; p->nRefPitch[2] = refF->pl[2].pitch; }
    p->searchType = st; p->nSearchParam = stp;

    int nLambdaLevel = lambda / ((1 << p->nLogPel) * (1 << p->nLogPel)); /* :1024-1028 */
    if (plevel == 1) nLambdaLevel = nLambdaLevel * p->nScale;
    else if (plevel == 2) nLambdaLevel = nLambdaLevel * p->nScale * p->nScale;

    p->penaltyZero = pzero; p->pglobal = pglobal; p->badcount = 0; p->tryMany = tryMany; p->sumLumaChange = 0;
    int bps = p->bytesPerSample;

    for (p->blky = 0; p->blky < p->nBlkY; p->blky++) {
        p->blkScanDir = (p->blky % 2 == 0 || meander == 0) ? 1 : -1;
        int blkxStart = (p->blky % 2 == 0 || meander == 0) ? 0 : p->nBlkX - 1;
        if (p->blkScanDir == 1) {
            p->x[0] = s0->hpad;
            if (p->chroma) { p->x[1] = srcF->pl[1].hpad; p->x[2] = srcF->pl[2].hpad; }
        } else {
            p->x[0] = s0->hpad + (p->nBlkSizeX - p->nOverlapX) * (p->nBlkX - 1);
            if (p->chroma) {
                p->x[1] = srcF->pl[1].hpad + ((p->nBlkSizeX - p->nOverlapX) / p->xRatioUV) * (p->nBlkX - 1);
                p->x[2] = srcF->pl[2].hpad + ((p->nBlkSizeX - p->nOverlapX) / p->xRatioUV) * (p->nBlkX - 1);
            }
        }
        for (int iblkx = 0; iblkx < p->nBlkX; iblkx++) {
            p->blkx = blkxStart + iblkx * p->blkScanDir;
            p->blkIdx = p->blky * p->nBlkX + p->blkx;

            /* source block staged into a contiguous temp: :1058-1079 */
            copy_block(p->pSrc_temp[0], p->nSrcPitch_temp[0], s0->p[0] + p->x[0] * bps + p->y[0] * s0->pitch, s0->pitch,
                       p->nBlkSizeX * bps, p->nBlkSizeY);
            p->pSrc[0] = p->pSrc_temp[0]; p->nSrcPitch[0] = p->nSrcPitch_temp[0];
            if (p->chroma) {
                for (int c = 1; c < 3; c++) {
                    const mvo_plane *sc = &srcF->pl[c];
                    copy_block(p->pSrc_temp[c], p->nSrcPitch_temp[c], sc->p[0] + p->x[c] * bps + p->y[c] * sc->pitch, sc->pitch,
                               p->nBlkSizeX / p->xRatioUV * bps, p->nBlkSizeY / p->yRatioUV);
                    p->pSrc[c] = p->pSrc_temp[c]; p->nSrcPitch[c] = p->nSrcPitch_temp[c];
                }
            }
            p->nLambda = p->blky == 0 ? 0 : nLambdaLevel; /* :1081-1084 */
            p->penaltyNew = pnew; p->LSAD = lsad;

            int hps = s0->hpad >> p->nLogScale, vps = s0->vpad >> p->nLogScale; /* :1091-1097 */
            p->nDxMax = (s0->pw - p->x[0] - p->nBlkSizeX - s0->hpad + hps) << p->nLogPel;
            p->nDyMax = (s0->ph - p->y[0] - p->nBlkSizeY - s0->vpad + vps) << p->nLogPel;
            p->nDxMin = -((p->x[0] - s0->hpad + hps) << p->nLogPel);
            p->nDyMin = -((p->y[0] - s0->vpad + vps) << p->nLogPel);

            p->predictor = clip_mv(p, p->vectors[p->blkIdx]);
            p->predictors[4] = clip_mv(p, zeroMV);

            PseudoEPZSearch(p);
            pBlkData[p->blkx] = p->bestMV;

            if (p->smallestPlane) /* :1109-1110 */
                p->sumLumaChange += blk_luma(p, ref_block(p, 0, 0, 0), p->nRefPitch[0]) - blk_luma(p, p->pSrc[0], p->nSrcPitch[0]);

            if (iblkx < p->nBlkX - 1) {
                p->x[0] += (p->nBlkSizeX - p->nOverlapX) * p->blkScanDir;
                if (srcF->mode & MVO_UPLANE) p->x[1] += ((p->nBlkSizeX - p->nOverlapX) >> p->nLogxRatioUV) * p->blkScanDir;
                if (srcF->mode & MVO_VPLANE) p->x[2] += ((p->nBlkSizeX - p->nOverlapX) >> p->nLogxRatioUV) * p->blkScanDir;
            }
        }
        pBlkData += p->nBlkX;
        p->y[0] += (p->nBlkSizeY - p->nOverlapY);
        if (srcF->mode & MVO_UPLANE) p->y[1] += ((p->nBlkSizeY - p->nOverlapY) >> p->nLogyRatioUV);
        if (srcF->mode & MVO_VPLANE) p->y[2] += ((p->nBlkSizeY - p->nOverlapY) >> p->nLogyRatioUV);
    }
    if (p->smallestPlane) *pmeanLumaChange = p->sumLumaChange / p->nBlkCount;
}

/* PlaneOfBlocks.cpp:1447-1514 pobInterpolatePrediction */
static void pob_interpolate(pob *p, const pob *p2) {
    int normFactor = 3 - p->nLogPel + p2->nLogPel;
    int mulFactor = (normFactor < 0) ? -normFactor : 0;
    normFactor = (normFactor < 0) ? 0 : normFactor;
    int normov = (p->nBlkSizeX - p->nOverlapX) * (p->nBlkSizeY - p->nOverlapY);
    int aoddx = (p->nBlkSizeX * 3 - p->nOverlapX * 2), aevenx = (p->nBlkSizeX * 3 - p->nOverlapX * 4);
    int aoddy = (p->nBlkSizeY * 3 - p->nOverlapY * 2), aeveny = (p->nBlkSizeY * 3 - p->nOverlapY * 4);
    double scaleov = 1.0 / normov;
    for (int l = 0, index = 0; l < p->nBlkY; l++) {
        for (int k = 0; k < p->nBlkX; k++, index++) {
            mvo_vector v1, v2, v3, v4;
            int i = k, j = l;
            if (i >= 2 * p2->nBlkX) i = 2 * p2->nBlkX - 1;
            if (j >= 2 * p2->nBlkY) j = 2 * p2->nBlkY - 1;
            int offy = -1 + 2 * (j % 2), offx = -1 + 2 * (i % 2);
            if ((i == 0) || (i >= 2 * p2->nBlkX - 1)) {
                if ((j == 0) || (j >= 2 * p2->nBlkY - 1)) {
                    v1 = v2 = v3 = v4 = p2->vectors[i / 2 + (j / 2) * p2->nBlkX];
                } else {
                    v1 = v2 = p2->vectors[i / 2 + (j / 2) * p2->nBlkX];
                    v3 = v4 = p2->vectors[i / 2 + (j / 2 + offy) * p2->nBlkX];
                }
            } else if ((j == 0) || (j >= 2 * p2->nBlkY - 1)) {
                v1 = v2 = p2->vectors[i / 2 + (j / 2) * p2->nBlkX];
                v3 = v4 = p2->vectors[i / 2 + offx + (j / 2) * p2->nBlkX];
            } else {
                v1 = p2->vectors[i / 2 + (j / 2) * p2->nBlkX];
                v2 = p2->vectors[i / 2 + offx + (j / 2) * p2->nBlkX];
                v3 = p2->vectors[i / 2 + (j / 2 + offy) * p2->nBlkX];
                v4 = p2->vectors[i / 2 + offx + (j / 2 + offy) * p2->nBlkX];
            }
            int64_t temp_sad;
            mvo_vector *o = &p->vectors[index];
            if (p->nOverlapX == 0 && p->nOverlapY == 0) {
                o->x = 9 * v1.x + 3 * v2.x + 3 * v3.x + v4.x;
                o->y = 9 * v1.y + 3 * v2.y + 3 * v3.y + v4.y;
                temp_sad = 9 * v1.sad + 3 * v2.sad + 3 * v3.sad + v4.sad + 8;
            } else if (p->nOverlapX <= (p->nBlkSizeX >> 1) && p->nOverlapY <= (p->nBlkSizeY >> 1)) {
                int ax1 = (offx > 0) ? aoddx : aevenx;
                int ax2 = (p->nBlkSizeX - p->nOverlapX) * 4 - ax1;
                int ay1 = (offy > 0) ? aoddy : aeveny;
                int ay2 = (p->nBlkSizeY - p->nOverlapY) * 4 - ay1;
                int64_t a11 = ax1 * ay1, a12 = ax1 * ay2, a21 = ax2 * ay1, a22 = ax2 * ay2;
                o->x = (int)((a11 * v1.x + a21 * v2.x + a12 * v3.x + a22 * v4.x) * scaleov);
                o->y = (int)((a11 * v1.y + a21 * v2.y + a12 * v3.y + a22 * v4.y) * scaleov);
                temp_sad = (int64_t)((a11 * v1.sad + a21 * v2.sad + a12 * v3.sad + a22 * v4.sad) * scaleov);
            } else {
                o->x = (v1.x + v2.x + v3.x + v4.x) << 2;
                o->y = (v1.y + v2.y + v3.y + v4.y) << 2;
                temp_sad = (v1.sad + v2.sad + v3.sad + v4.sad + 2) << 2;
            }
            o->x = (o->x >> normFactor) * (1 << mulFactor);
            o->y = (o->y >> normFactor) * (1 << mulFactor);
            o->sad = temp_sad >> 4;
        }
    }
}

/* PlaneOfBlocks.cpp:1559-1636 pobEstimateGlobalMVDoubled */
static void pob_estimate_global(pob *p, mvo_vector *g) {
    int freqSize = 8192 * p->nPel * 2;
    int *freq = (int *)malloc(freqSize * sizeof(int));
    int med[2];
    for (int c = 0; c < 2; c++) {
        memset(freq, 0, freqSize * sizeof(int));
        int indmin = freqSize - 1, indmax = 0;
        for (int i = 0; i < p->nBlkCount; i++) {
            int ind = (freqSize >> 1) + (c ? p->vectors[i].y : p->vectors[i].x);
            if (ind >= 0 && ind < freqSize) {
                freq[ind] += 1;
                if (ind > indmax) indmax = ind;
                if (ind < indmin) indmin = ind;
            }
        }
        int count = freq[indmin], index = indmin;
        for (int i = indmin + 1; i <= indmax; i++)
            if (freq[i] > count) { count = freq[i]; index = i; }
        med[c] = index - (freqSize >> 1);
    }
    free(freq);
    int meanvx = 0, meanvy = 0, num = 0;
    for (int i = 0; i < p->nBlkCount; i++)
        if (abs(p->vectors[i].x - med[0]) < 6 && abs(p->vectors[i].y - med[1]) < 6) { meanvx += p->vectors[i].x; meanvy += p->vectors[i].y; num += 1; }
    if (num > 0) { g->x = 2 * meanvx / num; g->y = 2 * meanvy / num; }
    else { g->x = 2 * med[0]; g->y = 2 * med[1]; }
}

static int pob_array_size(const pob *p, int divide) { /* :1517-1526 */
    int size = (int)sizeof(int) + p->nBlkCount * (int)sizeof(mvo_vector);
    if (p->nLogScale == 0 && divide) size += (int)sizeof(int) + p->nBlkCount * (int)sizeof(mvo_vector) * 4;
    return size;
}

static int pob_write_default(const pob *p, uint8_t *array, int divide) { /* :1529-1556 */
    int size = (int)sizeof(int) + p->nBlkCount * (int)sizeof(mvo_vector);
    memcpy(array, &size, sizeof(size));
    mvo_vector def = { 0, 0, p->verybigSAD };
    mvo_vector *blocks = (mvo_vector *)(array + sizeof(size));
    for (int i = 0; i < p->nBlkCount; i++) blocks[i] = def;
    if (p->nLogScale == 0 && divide) {
        array += size;
        size = (int)sizeof(int) + p->nBlkCount * (int)sizeof(mvo_vector) * 4;
        memcpy(array, &size, sizeof(size));
        blocks = (mvo_vector *)(array + sizeof(size));
        for (int i = 0; i < p->nBlkCount * 4; i++) blocks[i] = def;
    }
    return pob_array_size(p, divide);
}

/* PlaneOfBlocks.cpp:1158-1424 doPobRecalculateMVs: every block is independent -- its predictor is interpolated from the OLD
 * vector field, evaluated, and refined with the chosen pattern only when its SAD exceeds thSAD */
static void pob_recalculate(pob *p, const mvo_analysis_data *oldAd, const mvo_vector *oldVec, const mvo_frame *srcF, const mvo_frame *refF, int st, int stp,
                            int lambda, int pnew, uint8_t *out, int fieldShift, int64_t thSAD, int dctmode, int smooth, int meander) {
    p->dctmode = dctmode;
    p->dctweight16 = 8;
    p->zeroMVfieldShifted.x = 0; p->zeroMVfieldShifted.y = fieldShift; p->zeroMVfieldShifted.sad = 0;
    p->globalMVPredictor.x = 0; p->globalMVPredictor.y = fieldShift; p->globalMVPredictor.sad = 9999999;
    int size = (int)sizeof(int) + p->nBlkCount * (int)sizeof(mvo_vector);
    memcpy(out, &size, sizeof(size));
    mvo_vector *pBlkData = (mvo_vector *)(out + sizeof(int));
    p->pSrcFrame = srcF; p->pRefFrame = refF;
    const mvo_plane *s0 = &srcF->pl[0];
    const int bps = p->bytesPerSample;
    p->nRefPitch[0] = refF->pl[0].pitch;
    if (p->chroma) { p->nRefPitch[1] = refF->pl[1].pitch; p->nRefPitch[2] = refF->pl[2].pitch; }
    p->searchType = st; p->nSearchParam = stp;
    const int nLambdaLevel = lambda / ((1 << p->nLogPel) * (1 << p->nLogPel));
    const int nBlkXold = oldAd->nBlkX, nBlkYold = oldAd->nBlkY, bsxOld = oldAd->nBlkSizeX, bsyOld = oldAd->nBlkSizeY;
    const int stepXold = bsxOld - oldAd->nOverlapX, stepYold = bsyOld - oldAd->nOverlapY, logPelOld = mvo_ilog2(oldAd->nPel);
    (void)meander; /* the scan order cannot matter: blocks are independent */
    for (p->blky = 0; p->blky < p->nBlkY; p->blky++)
        for (p->blkx = 0; p->blkx < p->nBlkX; p->blkx++) {
            p->blkIdx = p->blky * p->nBlkX + p->blkx;
            p->blkScanDir = 1;
            p->x[0] = s0->hpad + (p->nBlkSizeX - p->nOverlapX) * p->blkx;
            p->y[0] = s0->vpad + (p->nBlkSizeY - p->nOverlapY) * p->blky;
            for (int c = 1; c < 3 && p->chroma; c++) {
                p->x[c] = srcF->pl[c].hpad + ((p->nBlkSizeX - p->nOverlapX) >> p->nLogxRatioUV) * p->blkx;
                p->y[c] = srcF->pl[c].vpad + ((p->nBlkSizeY - p->nOverlapY) >> p->nLogyRatioUV) * p->blky;
            }
            copy_block(p->pSrc_temp[0], p->nSrcPitch_temp[0], s0->p[0] + p->x[0] * bps + p->y[0] * s0->pitch, s0->pitch, p->nBlkSizeX * bps, p->nBlkSizeY);
            p->pSrc[0] = p->pSrc_temp[0]; p->nSrcPitch[0] = p->nSrcPitch_temp[0];
            for (int c = 1; c < 3 && p->chroma; c++) {
                const mvo_plane *sc = &srcF->pl[c];
                copy_block(p->pSrc_temp[c], p->nSrcPitch_temp[c], sc->p[0] + p->x[c] * bps + p->y[c] * sc->pitch, sc->pitch, p->nBlkSizeX / p->xRatioUV * bps, p->nBlkSizeY / p->yRatioUV);
                p->pSrc[c] = p->pSrc_temp[c]; p->nSrcPitch[c] = p->nSrcPitch_temp[c];
            }
            p->nLambda = p->blky == 0 ? 0 : nLambdaLevel;
            p->penaltyNew = pnew;
            p->nDxMax = (s0->pw - p->x[0] - p->nBlkSizeX) << p->nLogPel; /* :1262-1265: no level padding terms here */
            p->nDyMax = (s0->ph - p->y[0] - p->nBlkSizeY) << p->nLogPel;
            p->nDxMin = -(p->x[0] << p->nLogPel);
            p->nDyMin = -(p->y[0] << p->nLogPel);
            /* old vectors around the new block's centre (:1268-1321) */
            const int centerX = p->nBlkSizeX / 2 + (p->nBlkSizeX - p->nOverlapX) * p->blkx, blkxold = (centerX - bsxOld / 2) / stepXold;
            const int centerY = p->nBlkSizeY / 2 + (p->nBlkSizeY - p->nOverlapY) * p->blky, blkyold = (centerY - bsyOld / 2) / stepYold;
            const int deltaX = VMAX(0, centerX - (bsxOld / 2 + stepXold * blkxold)), deltaY = VMAX(0, centerY - (bsyOld / 2 + stepYold * blkyold));
            const int x1 = VMIN(nBlkXold - 1, VMAX(0, blkxold)), x2 = VMIN(nBlkXold - 1, VMAX(0, blkxold + 1));
            const int y1 = VMIN(nBlkYold - 1, VMAX(0, blkyold)), y2 = VMIN(nBlkYold - 1, VMAX(0, blkyold + 1));
            mvo_vector vo;
            if (smooth == 1) {
                const mvo_vector v1 = oldVec[x1 + y1 * nBlkXold], v2 = oldVec[x2 + y1 * nBlkXold], v3 = oldVec[x1 + y2 * nBlkXold], v4 = oldVec[x2 + y2 * nBlkXold];
                const int a_x = v1.x * stepXold + deltaX * (v2.x - v1.x), a_y = v1.y * stepXold + deltaX * (v2.y - v1.y);
                const int64_t a_s = v1.sad * stepXold + deltaX * (v2.sad - v1.sad);
                const int b_x = v3.x * stepXold + deltaX * (v4.x - v3.x), b_y = v3.y * stepXold + deltaX * (v4.y - v3.y);
                const int64_t b_s = v3.sad * stepXold + deltaX * (v4.sad - v3.sad);
                vo.x = (a_x + deltaY * (b_x - a_x) / stepYold) / stepXold;
                vo.y = (a_y + deltaY * (b_y - a_y) / stepYold) / stepXold;
                vo.sad = (a_s + deltaY * (b_s - a_s) / stepYold) / stepXold;
            } else {
                const int rx = deltaX * 2 >= stepXold, ry = deltaY * 2 >= stepYold;
                vo = oldVec[(rx ? x2 : x1) + (ry ? y2 : y1) * nBlkXold];
            }
            vo.x = (vo.x << p->nLogPel) >> logPelOld;
            vo.y = (vo.y << p->nLogPel) >> logPelOld;
            p->predictor = clip_mv(p, vo);
            p->predictor.sad = vo.sad * (p->nBlkSizeX * p->nBlkSizeY) / (bsxOld * bsyOld);
            p->bestMV = p->predictor;
            if (dctmode >= 3) p->srcLuma = blk_luma(p, p->pSrc[0], p->nSrcPitch[0]);
            const int64_t sad = full_sad(p, p->predictor.x, p->predictor.y, p->predictor.x, p->predictor.y);
            p->bestMV.sad = sad;
            p->nMinCost = sad;
            if (p->bestMV.sad > thSAD) Refine(p);
            p->vectors[p->blkIdx] = p->bestMV;
            pBlkData[p->blkIdx] = p->bestMV;
        }
}

/* GroupOfPlanes.c:177-302 Median3, GetMedian, gopExtraDivide (divide = 1: copies, 2: medians for the interior blocks) */
static int median3d(int a, int b, int c) {
    if (((b <= a) && (a <= c)) || ((c <= a) && (a <= b))) return a;
    else if (((a <= b) && (b <= c)) || ((c <= b) && (b <= a))) return b;
    return c;
}
static void get_median(int *vx, int *vy, int vx1, int vy1, int vx2, int vy2, int vx3, int vy3) {
    *vx = median3d(vx1, vx2, vx3); *vy = median3d(vy1, vy2, vy3);
    if ((*vx == vx1 && *vy == vy1) || (*vx == vx2 && *vy == vy2) || (*vx == vx3 && *vy == vy3)) return;
    *vx = vx1; *vy = vy1;
}
static void extra_divide(int divide, int nBlkX, int nBlkY, const mvo_vector *in, mvo_vector *outv) {
    for (int by = 0; by < nBlkY; by++)
        for (int bx = 0; bx < nBlkX; bx++) {
            mvo_vector b = in[by * nBlkX + bx];
            b.sad >>= 2;
            mvo_vector *o = outv + (size_t)by * nBlkX * 4 + bx * 2;
            o[0] = o[1] = o[nBlkX * 2] = o[nBlkX * 2 + 1] = b;
            if (divide > 1 && by >= 1 && by < nBlkY - 1 && bx >= 1 && bx < nBlkX - 1) {
                const mvo_vector *c = &in[by * nBlkX + bx];
                get_median(&o[0].x, &o[0].y, c->x, c->y, c[-1].x, c[-1].y, c[-nBlkX].x, c[-nBlkX].y);
                get_median(&o[1].x, &o[1].y, c->x, c->y, c[1].x, c[1].y, c[-nBlkX].x, c[-nBlkX].y);
                get_median(&o[nBlkX * 2].x, &o[nBlkX * 2].y, c->x, c->y, c[-1].x, c[-1].y, c[nBlkX].x, c[nBlkX].y);
                get_median(&o[nBlkX * 2 + 1].x, &o[nBlkX * 2 + 1].y, c->x, c->y, c[1].x, c[1].y, c[nBlkX].x, c[nBlkX].y);
            }
        }
}

/* ------------------------------------------------------------------ GroupOfPlanes.c */

typedef struct gop {
    int nLevelCount, divideExtra;
    pob planes[MVO_MAX_LEVELS];
} gop;

/* GroupOfPlanes.c:25-56 gopInit */
static void gop_init(gop *g, const mvo_analyse *d) {
    const mvo_analysis_data *a = &d->ad;
    g->nLevelCount = a->nLvCount; g->divideExtra = d->divideExtra;
    int pelCur = a->nPel, flags = a->nMotionFlags;
    int nWidth_B = (a->nBlkSizeX - a->nOverlapX) * a->nBlkX + a->nOverlapX;
    int nHeight_B = (a->nBlkSizeY - a->nOverlapY) * a->nBlkY + a->nOverlapY;
    for (int i = 0; i < g->nLevelCount; i++) {
        if (i == g->nLevelCount - 1) flags |= MOTION_SMALLEST_PLANE;
        int bx = ((nWidth_B >> i) - a->nOverlapX) / (a->nBlkSizeX - a->nOverlapX);
        int by = ((nHeight_B >> i) - a->nOverlapY) / (a->nBlkSizeY - a->nOverlapY);
        pob_init(&g->planes[i], bx, by, a->nBlkSizeX, a->nBlkSizeY, pelCur, i, flags, a->nOverlapX, a->nOverlapY, a->xRatioUV, a->yRatioUV, a->bitsPerSample);
        pelCur = 1;
    }
}
static void gop_deinit(gop *g) { for (int i = 0; i < g->nLevelCount; i++) pob_deinit(&g->planes[i]); }

static int gop_array_size(gop *g) { /* :167-174 */
    int size = 2 * (int)sizeof(int);
    for (int i = g->nLevelCount - 1; i >= 0; i--) size += pob_array_size(&g->planes[i], g->divideExtra);
    return size;
}

/* GroupOfPlanes.c:69-125 gopSearchMVs */
static void gop_search(gop *g, mvo_gof *srcG, mvo_gof *refG, const mvo_analyse *d, uint8_t *out, int fieldShift) {
    int size = gop_array_size(g);
    memcpy(out, &size, sizeof(size));
    int validity = 1;
    memcpy(out + sizeof(size), &validity, sizeof(validity));
    out += 2 * sizeof(int);
    int L = g->nLevelCount;
    int fieldShiftCur = (L - 1 == 0) ? fieldShift : 0;
    mvo_vector globalMV = zeroMV;
    int pglobal = d->pglobal;
    if (!d->global) pglobal = d->pzero;
    int meanLumaChange = 0;
    int st = d->searchType, cst = d->searchTypeCoarse;
    int stSmallest = (L == 1 || st == SearchHorizontal || st == SearchVertical) ? st : cst;
    int spSmallest = (L == 1) ? d->nPelSearch : d->nSearchParam;
    int tryManyLevel = d->tryMany && L > 1;
    pob_search(&g->planes[L - 1], &srcG->fr[L - 1], &refG->fr[L - 1], stSmallest, spSmallest, d->nLambda, d->lsad, d->pnew, d->plevel,
               out, &globalMV, fieldShiftCur, d->dctmode, &meanLumaChange, d->pzero, pglobal, d->badSAD, d->badrange, d->meander, tryManyLevel);
    out += pob_array_size(&g->planes[L - 1], g->divideExtra);
    for (int i = L - 2; i >= 0; i--) {
        int stLevel = (i == 0 || st == SearchHorizontal || st == SearchVertical) ? st : cst;
        int spLevel = (i == 0) ? d->nPelSearch : d->nSearchParam;
        if (d->global) pob_estimate_global(&g->planes[i + 1], &globalMV);
        pob_interpolate(&g->planes[i], &g->planes[i + 1]);
        fieldShiftCur = (i == 0) ? fieldShift : 0;
        tryManyLevel = d->tryMany && i > 0;
        pob_search(&g->planes[i], &srcG->fr[i], &refG->fr[i], stLevel, spLevel, d->nLambda, d->lsad, d->pnew, d->plevel,
                   out, &globalMV, fieldShiftCur, d->dctmode, &meanLumaChange, d->pzero, pglobal, d->badSAD, d->badrange, d->meander, tryManyLevel);
        out += pob_array_size(&g->planes[i], g->divideExtra);
    }
}

/* GroupOfPlanes.c:150-164 gopWriteDefaultToArray */
static void gop_write_default(gop *g, uint8_t *array) {
    int size = gop_array_size(g);
    memcpy(array, &size, sizeof(size));
    int validity = 0;
    memcpy(array + sizeof(size), &validity, sizeof(validity));
    array += 2 * sizeof(int);
    for (int i = g->nLevelCount - 1; i >= 0; i--) array += pob_write_default(&g->planes[i], array, g->divideExtra);
}

/* GroupOfPlanes.c:206-302: the divided array follows plane 0's; its size header is NOT written by the reference on this
 * path (uninitialised malloc bytes there) -- the oracle writes the value pobWriteDefaultToArray would (:1543-1547) */
static void gop_extra_divide(gop *g, uint8_t *out) {
    out += 2 * sizeof(int);
    for (int i = g->nLevelCount - 1; i >= 1; i--) out += pob_array_size(&g->planes[i], 0);
    int size;
    memcpy(&size, out, sizeof(size));
    const pob *p0 = &g->planes[0];
    int dsize = (int)sizeof(int) + p0->nBlkCount * (int)sizeof(mvo_vector) * 4;
    memcpy(out + size, &dsize, sizeof(dsize));
    extra_divide(g->divideExtra, p0->nBlkX, p0->nBlkY, (const mvo_vector *)(out + sizeof(int)), (mvo_vector *)(out + size + sizeof(int)));
}

/* ------------------------------------------------------------------ filter shell: MVAnalyse.c */

void mvo_analyse_args_default(mvo_analyse_args *a) {
    int *f = (int *)a;
    for (size_t i = 0; i < sizeof(*a) / sizeof(int); i++) f[i] = MVO_UNSET;
}

#define ARG(v, dflt) ((v) == MVO_UNSET ? (dflt) : (v))
#define FAIL(...) do { snprintf(err, MVO_ERR, __VA_ARGS__); return -1; } while (0)

/* MVAnalyse.c:267-635 mvanalyseCreate */
int mvo_analyse_init(mvo_analyse *d, const mvo_analyse_args *a, const mvo_super *s, int numFrames, char *err) {
    memset(d, 0, sizeof(*d));
    if (err) err[0] = 0;
    mvo_analysis_data *ad = &d->ad;
    ad->nBlkSizeX = ARG(a->blksize, 8);
    ad->nBlkSizeY = ARG(a->blksizev, ad->nBlkSizeX);
    int levels = ARG(a->levels, 0);
    d->searchType = ARG(a->search, SearchHex2);
    d->searchTypeCoarse = ARG(a->search_coarse, SearchExhaustive);
    int searchparam = ARG(a->searchparam, 2);
    d->nPelSearch = ARG(a->pelsearch, 0);
    ad->isBackward = !!ARG(a->isb, 0);
    d->chroma = !!ARG(a->chroma, 1);
    ad->nDeltaFrame = ARG(a->delta, 1);
    int truemotion = !!ARG(a->truemotion, 1);
    d->nLambda = ARG(a->lambda, truemotion ? (1000 * ad->nBlkSizeX * ad->nBlkSizeY / 64) : 0);
    d->lsad = ARG(a->lsad, truemotion ? 1200 : 400);
    d->plevel = ARG(a->plevel, truemotion ? 1 : 0);
    d->global = !!ARG(a->global, truemotion ? 1 : 0);
    d->pnew = ARG(a->pnew, truemotion ? 50 : 0);
    d->pzero = ARG(a->pzero, d->pnew);
    d->pglobal = ARG(a->pglobal, 0);
    ad->nOverlapX = ARG(a->overlap, 0);
    ad->nOverlapY = ARG(a->overlapv, ad->nOverlapX);
    d->dctmode = ARG(a->dct, 0);
    d->divideExtra = ARG(a->divide, 0);
    d->badSAD = ARG(a->badsad, 10000);
    d->badrange = ARG(a->badrange, 24);
    d->opt = !!ARG(a->opt, 1);
    d->meander = !!ARG(a->meander, 1);
    d->tryMany = !!ARG(a->trymany, 0);
    d->fields = !!ARG(a->fields, 0);
    d->tff = !!ARG(a->tff, 0);
    d->tff_exists = a->tff != MVO_UNSET;
    d->numFrames = numFrames;

    if (d->searchType < 0 || d->searchType > 7) FAIL("Analyse: search must be between 0 and 7 (inclusive).");
    if (d->searchTypeCoarse < 0 || d->searchTypeCoarse > 7) FAIL("Analyse: search_coarse must be between 0 and 7 (inclusive).");
    if (d->dctmode < 0 || d->dctmode > 10) FAIL("Analyse: dct must be between 0 and 10 (inclusive).");
    if (d->dctmode >= 1 && d->dctmode <= 4) FAIL("Analyse: dct 1..4 need FFTW3 (out of scope for the oracle).");
    if (d->dctmode >= 5 && ad->nBlkSizeX == 16 && ad->nBlkSizeY == 2) FAIL("Analyse: dct 5..10 cannot work with 16x2 blocks.");
    if (d->divideExtra < 0 || d->divideExtra > 2) FAIL("Analyse: divide must be between 0 and 2 (inclusive).");
    {
        static const int ok[12][2] = { { 4, 4 }, { 8, 4 }, { 8, 8 }, { 16, 2 }, { 16, 8 }, { 16, 16 }, { 32, 16 }, { 32, 32 }, { 64, 32 }, { 64, 64 }, { 128, 64 }, { 128, 128 } };
        int found = 0;
        for (int i = 0; i < 12; i++) if (ad->nBlkSizeX == ok[i][0] && ad->nBlkSizeY == ok[i][1]) found = 1;
        if (!found) FAIL("Analyse: the block size must be 4x4, 8x4, 8x8, 16x2, 16x8, 16x16, 32x16, 32x32, 64x32, 64x64, 128x64, or 128x128.");
    }
    if (d->plevel < 0 || d->plevel > 2) FAIL("Analyse: plevel must be between 0 and 2 (inclusive).");
    if (d->pnew < 0 || d->pnew > 256) FAIL("Analyse: pnew must be between 0 and 256 (inclusive).");
    if (d->pzero < 0 || d->pzero > 256) FAIL("Analyse: pzero must be between 0 and 256 (inclusive).");
    if (d->pglobal < 0 || d->pglobal > 256) FAIL("Analyse: pglobal must be between 0 and 256 (inclusive).");
    if (ad->nOverlapX < 0 || ad->nOverlapX > ad->nBlkSizeX / 2 || ad->nOverlapY < 0 || ad->nOverlapY > ad->nBlkSizeY / 2)
        FAIL("Analyse: overlap must be at most half of blksize, overlapv must be at most half of blksizev, and they both need to be at least 0.");
    if (d->divideExtra && (ad->nBlkSizeX < 8 || ad->nBlkSizeY < 8)) FAIL("Analyse: blksize and blksizev must be at least 8 when divide=True."); /* :447 */
    if (d->searchType == SearchNstep) d->nSearchParam = (searchparam < 0) ? 0 : searchparam;
    else d->nSearchParam = (searchparam < 1) ? 1 : searchparam;
    if (d->divideExtra && (ad->nOverlapX % (2 * s->xRatioUV) || ad->nOverlapY % (2 * s->yRatioUV))) /* :503-505 */
        FAIL("Analyse: overlap and overlapv must be multiples of 2 or 4 when divide=True, depending on the super clip's subsampling.");

    if (s->gray) d->chroma = 0;
    int nModeYUV = d->chroma ? MVO_YUVPLANES : MVO_YPLANE;
    ad->bitsPerSample = s->bits;
    int pixelMax = (1 << s->bits) - 1; /* :477-483 */
    d->lsad = (int)((double)d->lsad * pixelMax / 255.0 + 0.5);
    d->badSAD = (int)((double)d->badSAD * pixelMax / 255.0 + 0.5);
    d->nLambda = (int)((double)d->nLambda * pixelMax / 255.0 + 0.5);
    d->lsad = (int)((int64_t)d->lsad * (ad->nBlkSizeX * ad->nBlkSizeY) / 64);
    d->badSAD = d->badSAD * (ad->nBlkSizeX * ad->nBlkSizeY) / 64;

    ad->nMotionFlags = 0;
    ad->nMotionFlags |= d->opt ? MOTION_USE_SIMD : 0;
    ad->nMotionFlags |= ad->isBackward ? MOTION_IS_BACKWARD : 0;
    ad->nMotionFlags |= d->chroma ? MOTION_USE_CHROMA_MOTION : 0;

    if (ad->nOverlapX % s->xRatioUV || ad->nOverlapY % s->yRatioUV)
        FAIL("Analyse: The requested overlap is incompatible with the super clip's subsampling.");
    if (ad->nDeltaFrame <= 0 && (-ad->nDeltaFrame) >= numFrames) FAIL("Analyse: delta points to frame past the input clip's end.");
    ad->yRatioUV = s->yRatioUV; ad->xRatioUV = s->xRatioUV;

    d->nSuperHPad = s->hpad; d->nSuperVPad = s->vpad; d->nSuperPel = s->pel; d->nSuperModeYUV = s->modeYUV; d->nSuperLevels = s->levels;
    if ((nModeYUV & d->nSuperModeYUV) != nModeYUV) FAIL("Analyse: super clip does not contain needed colour data.");

    ad->nWidth = s->superWidth - d->nSuperHPad * 2;
    ad->nHeight = s->height;
    ad->nPel = d->nSuperPel;
    ad->nHPadding = d->nSuperHPad; ad->nVPadding = d->nSuperVPad;
    int nBlkX = (ad->nWidth - ad->nOverlapX) / (ad->nBlkSizeX - ad->nOverlapX);
    int nBlkY = (ad->nHeight - ad->nOverlapY) / (ad->nBlkSizeY - ad->nOverlapY);
    ad->nBlkX = nBlkX; ad->nBlkY = nBlkY;
    int nWidth_B = (ad->nBlkSizeX - ad->nOverlapX) * nBlkX + ad->nOverlapX;
    int nHeight_B = (ad->nBlkSizeY - ad->nOverlapY) * nBlkY + ad->nOverlapY;
    int nLevelsMax = 0;
    while (((nWidth_B >> nLevelsMax) - ad->nOverlapX) / (ad->nBlkSizeX - ad->nOverlapX) > 0 &&
           ((nHeight_B >> nLevelsMax) - ad->nOverlapY) / (ad->nBlkSizeY - ad->nOverlapY) > 0)
        nLevelsMax++;
    ad->nLvCount = levels > 0 ? levels : nLevelsMax + levels;
    if (ad->nLvCount < 1 || ad->nLvCount > nLevelsMax) FAIL("Analyse: invalid number of levels.");
    if (ad->nLvCount > d->nSuperLevels) FAIL("Analyse: super clip has %d levels. Analyse needs %d levels.", d->nSuperLevels, ad->nLvCount);
    if (d->nPelSearch <= 0) d->nPelSearch = ad->nPel;
    return 0;
}

int mvo_analyse_blob_size(const mvo_analyse *d) {
    gop g;
    gop_init(&g, d);
    int n = gop_array_size(&g);
    gop_deinit(&g);
    return n;
}

/* MVAnalyse.c:110-221 arAllFramesReady body */
void mvo_analyse_frame(const mvo_analyse *d, const uint8_t *const src[3], const int srcPitch[3],
                       const uint8_t *const ref[3], const int refPitch[3], int fieldShift, uint8_t *blob) {
    gop g;
    gop_init(&g, d);
    if (ref) {
        const mvo_analysis_data *a = &d->ad;
        mvo_gof sg, rg;
        mvo_gof_init(&sg, d->nSuperLevels, a->nWidth, a->nHeight, d->nSuperPel, d->nSuperHPad, d->nSuperVPad, d->nSuperModeYUV, a->xRatioUV, a->yRatioUV, a->bitsPerSample);
        mvo_gof_init(&rg, d->nSuperLevels, a->nWidth, a->nHeight, d->nSuperPel, d->nSuperHPad, d->nSuperVPad, d->nSuperModeYUV, a->xRatioUV, a->yRatioUV, a->bitsPerSample);
        mvo_gof_update(&sg, (uint8_t *const *)src, srcPitch, a->yRatioUV);
        mvo_gof_update(&rg, (uint8_t *const *)ref, refPitch, a->yRatioUV);
        gop_search(&g, &sg, &rg, d, blob, fieldShift);
        if (d->divideExtra) gop_extra_divide(&g, blob);
    } else
        gop_write_default(&g, blob);
    gop_deinit(&g);
}


/* the analysis data a reader sees: with divide the extra level of half-size blocks (MVAnalyse.c:615-624, MVRecalculate.c:533-543) */
void mvo_analysis_data_divided(const mvo_analysis_data *in, mvo_analysis_data *out) {
    *out = *in;
    out->nBlkX = in->nBlkX * 2; out->nBlkY = in->nBlkY * 2;
    out->nBlkSizeX = in->nBlkSizeX / 2; out->nBlkSizeY = in->nBlkSizeY / 2;
    out->nOverlapX = in->nOverlapX / 2; out->nOverlapY = in->nOverlapY / 2;
    out->nLvCount = in->nLvCount + 1;
}

/* ------------------------------------------------------------------ filter shell: MVRecalculate.c */

void mvo_recalculate_args_default(mvo_recalculate_args *a) {
    int64_t *f = (int64_t *)a;
    for (size_t i = 0; i < sizeof(*a) / sizeof(int64_t); i++) f[i] = MVO_UNSET;
}

/* MVRecalculate.c:263-545 mvrecalculateCreate (fields: its fieldShift never reaches a result, PlaneOfBlocks.cpp:1167-1171) */
int mvo_recalculate_init(mvo_recalculate *d, const mvo_recalculate_args *a, const mvo_super *s, const mvo_analysis_data *vectors, char *err) {
    memset(d, 0, sizeof(*d));
    if (err) err[0] = 0;
    mvo_analyse *an = &d->an;
    mvo_analysis_data *ad = &an->ad;
    d->thSAD = ARG(a->thsad, 200);
    d->smooth = (int)ARG(a->smooth, 1);
    ad->nBlkSizeX = (int)ARG(a->blksize, 8);
    ad->nBlkSizeY = (int)ARG(a->blksizev, ad->nBlkSizeX);
    an->searchType = (int)ARG(a->search, SearchHex2);
    const int searchparam = (int)ARG(a->searchparam, 2);
    an->chroma = !!ARG(a->chroma, 1);
    const int truemotion = !!ARG(a->truemotion, 1);
    an->nLambda = (int)ARG(a->lambda, truemotion ? (1000 * ad->nBlkSizeX * ad->nBlkSizeY / 64) : 0);
    an->pnew = (int)ARG(a->pnew, truemotion ? 50 : 0);
    ad->nOverlapX = (int)ARG(a->overlap, 0);
    ad->nOverlapY = (int)ARG(a->overlapv, ad->nOverlapX);
    an->dctmode = (int)ARG(a->dct, 0);
    an->divideExtra = (int)ARG(a->divide, 0);
    an->opt = 1;
    an->meander = !!ARG(a->meander, 1);
    if (an->searchType < 0 || an->searchType > 7) FAIL("Recalculate: search must be between 0 and 7 (inclusive).");
    if (an->dctmode < 0 || an->dctmode > 10) FAIL("Recalculate: dct must be between 0 and 10 (inclusive).");
    if (an->dctmode >= 1 && an->dctmode <= 4) FAIL("Recalculate: dct 1..4 need FFTW3 (out of scope for the oracle).");
    if (an->dctmode >= 5 && ad->nBlkSizeX == 16 && ad->nBlkSizeY == 2) FAIL("Recalculate: dct 5..10 cannot work with 16x2 blocks.");
    if (an->divideExtra < 0 || an->divideExtra > 2) FAIL("Recalculate: divide must be between 0 and 2 (inclusive).");
    {
        static const int ok[12][2] = { { 4, 4 }, { 8, 4 }, { 8, 8 }, { 16, 2 }, { 16, 8 }, { 16, 16 }, { 32, 16 }, { 32, 32 }, { 64, 32 }, { 64, 64 }, { 128, 64 }, { 128, 128 } };
        int found = 0;
        for (int i = 0; i < 12; i++) if (ad->nBlkSizeX == ok[i][0] && ad->nBlkSizeY == ok[i][1]) found = 1;
        if (!found) FAIL("Recalculate: the block size must be 4x4, 8x4, 8x8, 16x2, 16x8, 16x16, 32x16, 32x32, 64x32, 64x64, 128x64, or 128x128.");
    }
    if (an->pnew < 0 || an->pnew > 256) FAIL("Recalculate: pnew must be between 0 and 256 (inclusive).");
    if (ad->nOverlapX < 0 || ad->nOverlapX > ad->nBlkSizeX / 2 || ad->nOverlapY < 0 || ad->nOverlapY > ad->nBlkSizeY / 2)
        FAIL("Recalculate: overlap must be at most half of blksize, overlapv must be at most half of blksizev, and they both need to be at least 0.");
    if (an->divideExtra && (ad->nBlkSizeX < 8 || ad->nBlkSizeY < 8)) FAIL("Recalculate: blksize and blksizev must be at least 8 when divide=True.");
    if (an->searchType == SearchNstep) an->nSearchParam = (searchparam < 0) ? 0 : searchparam;
    else an->nSearchParam = (searchparam < 1) ? 1 : searchparam;
    if (ad->nOverlapX % s->xRatioUV || ad->nOverlapY % s->yRatioUV) FAIL("Recalculate: The requested overlap is incompatible with the super clip's subsampling.");
    if (an->divideExtra && (ad->nOverlapX % (2 * s->xRatioUV) || ad->nOverlapY % (2 * s->yRatioUV)))
        FAIL("Recalculate: overlap and overlapv must be multiples of 2 or 4 when divide=True, depending on the super clip's subsampling.");
    if (s->gray) an->chroma = 0;
    const int nModeYUV = an->chroma ? MVO_YUVPLANES : MVO_YPLANE;
    if ((nModeYUV & s->modeYUV) != nModeYUV) FAIL("Recalculate: super clip does not contain needed colour data.");
    d->old = *vectors;
    ad->yRatioUV = vectors->yRatioUV; ad->xRatioUV = vectors->xRatioUV;
    ad->nWidth = vectors->nWidth; ad->nHeight = vectors->nHeight;
    ad->nDeltaFrame = vectors->nDeltaFrame; ad->isBackward = vectors->isBackward;
    ad->bitsPerSample = s->bits;
    const int pixelMax = (1 << s->bits) - 1;
    d->thSAD = (int64_t)((double)d->thSAD * pixelMax / 255.0 + 0.5);
    an->nLambda = (int)((double)an->nLambda * pixelMax / 255.0 + 0.5);
    d->thSAD = d->thSAD * (ad->nBlkSizeX * ad->nBlkSizeY) / 64;
    if (an->chroma) d->thSAD += d->thSAD / (ad->xRatioUV * ad->yRatioUV) * 2;
    ad->nMotionFlags = MOTION_USE_SIMD | (ad->isBackward ? MOTION_IS_BACKWARD : 0) | (an->chroma ? MOTION_USE_CHROMA_MOTION : 0);
    ad->nPel = s->pel;
    if (s->height != ad->nHeight || s->superWidth - 2 * s->hpad != ad->nWidth) FAIL("Recalculate: wrong frame size.");
    ad->nHPadding = s->hpad; ad->nVPadding = s->vpad;
    ad->nBlkX = (ad->nWidth - ad->nOverlapX) / (ad->nBlkSizeX - ad->nOverlapX);
    ad->nBlkY = (ad->nHeight - ad->nOverlapY) / (ad->nBlkSizeY - ad->nOverlapY);
    ad->nLvCount = 1;
    an->nSuperLevels = s->levels; an->nSuperHPad = s->hpad; an->nSuperVPad = s->vpad; an->nSuperPel = s->pel; an->nSuperModeYUV = s->modeYUV;
    return 0;
}

int mvo_recalculate_blob_size(const mvo_recalculate *d) { return mvo_analyse_blob_size(&d->an); }

/* MVRecalculate.c:104-228: oldBlob = the old vector clip's MVTools_vectors at frame n; ref == NULL or an invalid old blob
 * give the default (invalid) array */
void mvo_recalculate_frame(const mvo_recalculate *d, const uint8_t *const src[3], const int srcPitch[3], const uint8_t *const ref[3], const int refPitch[3],
                           const uint8_t *oldBlob, uint8_t *blob) {
    gop g;
    gop_init(&g, &d->an);
    int valid;
    memcpy(&valid, oldBlob + sizeof(int), sizeof(valid));
    if (ref && valid) {
        const mvo_analysis_data *a = &d->an.ad;
        mvo_gof sg, rg;
        mvo_gof_init(&sg, d->an.nSuperLevels, a->nWidth, a->nHeight, d->an.nSuperPel, d->an.nSuperHPad, d->an.nSuperVPad, d->an.nSuperModeYUV, a->xRatioUV, a->yRatioUV, a->bitsPerSample);
        mvo_gof_init(&rg, d->an.nSuperLevels, a->nWidth, a->nHeight, d->an.nSuperPel, d->an.nSuperHPad, d->an.nSuperVPad, d->an.nSuperModeYUV, a->xRatioUV, a->yRatioUV, a->bitsPerSample);
        mvo_gof_update(&sg, (uint8_t *const *)src, srcPitch, a->yRatioUV);
        mvo_gof_update(&rg, (uint8_t *const *)ref, refPitch, a->yRatioUV);
        int size = gop_array_size(&g), one = 1; /* GroupOfPlanes.c:128-147 gopRecalculateMVs */
        memcpy(blob, &size, sizeof(size));
        memcpy(blob + sizeof(int), &one, sizeof(one));
        pob_recalculate(&g.planes[0], &d->old, mvo_blob_level0(&d->old, oldBlob), &sg.fr[0], &rg.fr[0], d->an.searchType, d->an.nSearchParam, d->an.nLambda, d->an.pnew,
                        blob + 2 * sizeof(int), 0, d->thSAD, d->an.dctmode, d->smooth, d->an.meander);
        if (d->an.divideExtra) gop_extra_divide(&g, blob);
    } else
        gop_write_default(&g, blob);
    gop_deinit(&g);
}

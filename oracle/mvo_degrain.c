/*
 * mvo_degrain.c -- oracle restatement of the vector-blob reader, overlap windows, mv.DegrainN and
 * mv.Compensate (test infrastructure only, see mvoracle.h).
 * Follows /root/reference/src/Fakery.c, MVAnalysisData.c, Overlap.cpp, MVDegrains.{h,cpp}, MVCompensate.c.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvo_internal.h"

#define VMAX(a, b) ((a) > (b) ? (a) : (b))
#define VMIN(a, b) ((a) > (b) ? (b) : (a))
#define MOTION_USE_CHROMA_MOTION 8

/* ------------------------------------------------------------------ blob reader */

/* MVAnalysisData.c:7-31 scaleThSCD (error path for thscd1 > 8*8*255 is checked by the caller) */
void mvo_scale_thscd(int64_t *thscd1, int *thscd2, const mvo_analysis_data *ad) {
    int referenceBlockSize = 8 * 8;
    *thscd1 = *thscd1 * (ad->nBlkSizeX * ad->nBlkSizeY) / referenceBlockSize;
    if (ad->nMotionFlags & MOTION_USE_CHROMA_MOTION)
        *thscd1 += *thscd1 / (ad->xRatioUV * ad->yRatioUV) * 2;
    int pixelMax = (1 << ad->bitsPerSample) - 1;
    *thscd1 = (int64_t)((double)*thscd1 * pixelMax / 255.0 + 0.5);
    *thscd2 = *thscd2 * ad->nBlkX * ad->nBlkY / 256;
}

/* Fakery.c:110-121 fgopUpdate: level 0 is the last plane in the blob */
const mvo_vector *mvo_blob_level0(const mvo_analysis_data *ad, const uint8_t *blob) {
    const uint8_t *pA = blob + 2 * sizeof(int);
    for (int i = ad->nLvCount - 1; i > 0; i--) {
        int size;
        memcpy(&size, pA, sizeof(size));
        pA += size;
    }
    return (const mvo_vector *)(pA + sizeof(int));
}

/* Fakery.c:52-58 fpobIsSceneChange, :103-107 validity, :144-146 fgopIsUsable */
int mvo_blob_is_usable(const mvo_analysis_data *ad, const uint8_t *blob, int64_t thscd1, int thscd2) {
    int validity;
    memcpy(&validity, blob + sizeof(int), sizeof(validity));
    const mvo_vector *v = mvo_blob_level0(ad, blob);
    int sum = 0;
    for (int i = 0; i < ad->nBlkX * ad->nBlkY; i++) sum += (v[i].sad > thscd1) ? 1 : 0;
    return !(sum > thscd2) && (validity == 1);
}

/* ------------------------------------------------------------------ overlap windows */

/* Overlap.cpp:40-125 overInit.  M_PI there is glibc's double constant (the float fallback at :26-28 is
 * not taken); the argument is formed in double, rounded to float for cosf. */
static void win1d(float *w, float *first, float *last, int n, int o) {
    for (int i = 0; i < o; i++) {
        w[i] = cosf((float)(M_PI * (i - o + 0.5f) / (o * 2)));
        w[i] = w[i] * w[i];
        first[i] = 1; last[i] = w[i];
    }
    for (int i = o; i < n - o; i++) { w[i] = 1; first[i] = 1; last[i] = 1; }
    for (int i = n - o; i < n; i++) {
        w[i] = cosf((float)(M_PI * (i - n + o + 0.5f) / (o * 2)));
        w[i] = w[i] * w[i];
        first[i] = w[i]; last[i] = 1;
    }
}

void mvo_over_init(int16_t *win9, int nx, int ny, int ox, int oy) {
    float *fx = (float *)malloc(sizeof(float) * nx * 3), *fy = (float *)malloc(sizeof(float) * ny * 3);
    float *x[3] = { fx + nx, fx, fx + 2 * nx }; /* [0]=first [1]=middle [2]=last */
    float *y[3] = { fy + ny, fy, fy + 2 * ny };
    win1d(fx, fx + nx, fx + 2 * nx, nx, ox);
    win1d(fy, fy + ny, fy + 2 * ny, ny, oy);
    int size = nx * ny;
    for (int wy = 0; wy < 3; wy++)
        for (int wx = 0; wx < 3; wx++) {
            int16_t *w = win9 + size * (wy * 3 + wx);
            for (int j = 0; j < ny; j++)
                for (int i = 0; i < nx; i++)
                    w[j * nx + i] = (int16_t)(int)(y[wy][j] * x[wx][i] * 2048 + 0.5f);
        }
    free(fx); free(fy);
}

/* Overlap.cpp:143-158 overlaps_c */
void mvo_overlaps(int w, int h, int bits, uint8_t *dst, intptr_t dstPitch, const uint8_t *src, intptr_t srcPitch,
                  const int16_t *win, intptr_t winPitch) {
    for (int j = 0; j < h; j++) {
        if (bits <= 8) {
            uint16_t *d = (uint16_t *)dst; const uint8_t *s = src;
            for (int i = 0; i < w; i++) d[i] += (uint16_t)((s[i] * win[i]) >> 6);
        } else {
            uint32_t *d = (uint32_t *)dst; const uint16_t *s = (const uint16_t *)src;
            for (int i = 0; i < w; i++) d[i] += (uint32_t)((s[i] * win[i]) >> 6);
        }
        dst += dstPitch; src += srcPitch; win += winPitch;
    }
}

/* Overlap.cpp:335-356 ToPixels */
void mvo_to_pixels(int bits, uint8_t *dst, int dstPitch, const uint8_t *src, int srcPitch, int w, int h) {
    int pixelMax = (1 << bits) - 1;
    for (int y = 0; y < h; y++) {
        if (bits <= 8) {
            const uint16_t *s = (const uint16_t *)src;
            for (int i = 0; i < w; i++) { int a = (s[i] + 16) >> 5; dst[i] = (uint8_t)(a | ((255 - a) >> 31)); }
        } else {
            const uint32_t *s = (const uint32_t *)src; uint16_t *d = (uint16_t *)dst;
            for (int i = 0; i < w; i++) { int a = (int)((s[i] + 16) >> 5); d[i] = (uint16_t)VMIN(pixelMax, a); }
        }
        dst += dstPitch; src += srcPitch;
    }
}

static void bitblt(uint8_t *d, int dp, const uint8_t *s, int sp, int rowbytes, int h) {
    for (int y = 0; y < h; y++) memcpy(d + (size_t)y * dp, s + (size_t)y * sp, rowbytes);
}

/* ------------------------------------------------------------------ mv.DegrainN */

/* MVDegrains.h:184-189 */
static int degrain_weight(int64_t thSAD, int64_t blockSAD) {
    if (blockSAD >= thSAD) return 0;
    return (int)((thSAD - blockSAD) * (thSAD + blockSAD) * 256 / (double)(thSAD * thSAD + blockSAD * blockSAD));
}

/* MVDegrains.h:208-223 */
static void normalise_weights(int n, int *WSrc, int *WRefs) {
    *WSrc = 256;
    int WSum = *WSrc + 1;
    for (int r = 0; r < n; r++) WSum += WRefs[r];
    double scale = 256.0 / WSum;
    for (int r = 0; r < n; r++) { WRefs[r] = (int)(WRefs[r] * scale); *WSrc -= WRefs[r]; }
}

/* MVDegrains.h:30-53 Degrain_C */
static void degrain_block(int n, int w, int h, int bps, uint8_t *dst, int dp, const uint8_t *src, int sp,
                          const uint8_t **refs, const int *rp, int WSrc, const int *WRefs) {
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int sum = 128 + (bps == 1 ? src[x] : ((const uint16_t *)src)[x]) * WSrc;
            for (int r = 0; r < n; r++) sum += (bps == 1 ? refs[r][x] : ((const uint16_t *)refs[r])[x]) * WRefs[r];
            if (bps == 1) dst[x] = (uint8_t)(sum >> 8); else ((uint16_t *)dst)[x] = (uint16_t)(sum >> 8);
        }
        dst += dp; src += sp;
        for (int r = 0; r < n; r++) refs[r] += rp[r];
    }
}

/* MVDegrains.h:163-181 LimitChanges_C */
static void limit_changes(int bps, uint8_t *dst, int dp, const uint8_t *src, int sp, int w, int h, int limit) {
    for (int y = 0; y < h; y++) {
        for (int i = 0; i < w; i++) {
            if (bps == 1) { int s = src[i], d = dst[i]; dst[i] = (uint8_t)VMIN(VMAX(d, s - limit), s + limit); }
            else { int s = ((const uint16_t *)src)[i], d = ((uint16_t *)dst)[i]; ((uint16_t *)dst)[i] = (uint16_t)VMIN(VMAX(d, s - limit), s + limit); }
        }
        dst += dp; src += sp;
    }
}

#define FAIL(...) do { snprintf(err, MVO_ERR, __VA_ARGS__); return -1; } while (0)

/* MVDegrains.cpp:511-809 mvdegrainCreate (vector clips assumed mutually consistent) */
int mvo_degrain_init(mvo_degrain *d, int radius, const mvo_analysis_data *ad, const mvo_super *s,
                     int64_t thsad, int64_t thsadc, int plane, int limit, int limitc, int64_t thscd1, int thscd2, char *err) {
    memset(d, 0, sizeof(*d));
    if (err) err[0] = 0;
    d->radius = radius; d->ad = *ad;
    d->thSAD[0] = thsad == MVO_UNSET ? 400 : thsad;
    d->thSAD[1] = d->thSAD[2] = thsadc == MVO_UNSET ? d->thSAD[0] : thsadc;
    if (plane == MVO_UNSET) plane = 4;
    d->nSCD1 = thscd1 == MVO_UNSET ? 400 : thscd1;
    d->nSCD2 = thscd2 == MVO_UNSET ? 130 : thscd2;
    if (plane < 0 || plane > 4) FAIL("Degrain%d: plane must be between 0 and 4 (inclusive).", radius);
    static const int planes[5] = { MVO_YPLANE, MVO_UPLANE, MVO_VPLANE, MVO_UPLANE | MVO_VPLANE, MVO_YUVPLANES };
    int YUVplanes = planes[plane];
    d->nSuperHPad = s->hpad; d->nSuperVPad = s->vpad; d->nSuperPel = s->pel; d->nSuperModeYUV = s->modeYUV; d->nSuperLevels = s->levels;
    if (d->nSCD1 > 8 * 8 * 255) FAIL("Degrain%d: thscd1 can be at most %d.", radius, 8 * 8 * 255);
    int64_t nSCD1_old = d->nSCD1;
    mvo_scale_thscd(&d->nSCD1, &d->nSCD2, ad);
    d->thSAD[0] = d->thSAD[0] * d->nSCD1 / nSCD1_old; /* :658-659 */
    d->thSAD[1] = d->thSAD[2] = d->thSAD[1] * d->nSCD1 / nSCD1_old;
    if (d->thSAD[0] >= 2147483647 || d->thSAD[1] >= 2147483647) FAIL("Degrain%d: thsad too large.", radius);
    if (ad->nHeight != s->height || ad->nWidth != s->superWidth - s->hpad * 2 || ad->nWidth != s->width || ad->nPel != s->pel)
        FAIL("Degrain%d: wrong source or super clip frame size.", radius);
    d->bits = s->bits; d->numPlanes = s->gray ? 1 : 3;
    int pixelMax = (1 << s->bits) - 1;
    d->nLimit[0] = limit == MVO_UNSET ? pixelMax : limit;
    d->nLimit[1] = d->nLimit[2] = limitc == MVO_UNSET ? d->nLimit[0] : limitc;
    if (d->nLimit[0] < 0 || d->nLimit[0] > pixelMax) FAIL("Degrain%d: limit must be between 0 and %d (inclusive).", radius, pixelMax);
    if (d->nLimit[1] < 0 || d->nLimit[1] > pixelMax) FAIL("Degrain%d: limitc must be between 0 and %d (inclusive).", radius, pixelMax);
    d->process[0] = !!(YUVplanes & MVO_YPLANE);
    d->process[1] = !!(YUVplanes & MVO_UPLANE & d->nSuperModeYUV);
    d->process[2] = !!(YUVplanes & MVO_VPLANE & d->nSuperModeYUV);
    d->xSubUV = mvo_ilog2(s->xRatioUV); d->ySubUV = mvo_ilog2(s->yRatioUV);
    d->nWidth[0] = ad->nWidth; d->nHeight[0] = ad->nHeight; d->nOverlapX[0] = ad->nOverlapX; d->nOverlapY[0] = ad->nOverlapY;
    d->nBlkSizeX[0] = ad->nBlkSizeX; d->nBlkSizeY[0] = ad->nBlkSizeY;
    d->nWidth_B[0] = ad->nBlkX * (ad->nBlkSizeX - ad->nOverlapX) + ad->nOverlapX;
    d->nHeight_B[0] = ad->nBlkY * (ad->nBlkSizeY - ad->nOverlapY) + ad->nOverlapY;
    for (int p = 1; p < 3; p++) {
        d->nWidth[p] = d->nWidth[0] >> d->xSubUV; d->nHeight[p] = d->nHeight[0] >> d->ySubUV;
        d->nOverlapX[p] = d->nOverlapX[0] >> d->xSubUV; d->nOverlapY[p] = d->nOverlapY[0] >> d->ySubUV;
        d->nBlkSizeX[p] = d->nBlkSizeX[0] >> d->xSubUV; d->nBlkSizeY[p] = d->nBlkSizeY[0] >> d->ySubUV;
        d->nWidth_B[p] = d->nWidth_B[0] >> d->xSubUV; d->nHeight_B[p] = d->nHeight_B[0] >> d->ySubUV;
    }
    return 0;
}

/* MVDegrains.cpp:85-330 mvdegrainGetFrame */
void mvo_degrain_frame(const mvo_degrain *d, const uint8_t *const src[3], const int srcPitch[3],
                       const uint8_t *const (*refs)[3], const int (*refPitch)[3], const uint8_t *const *blobs,
                       uint8_t *const dst[3], const int dstPitch[3]) {
    const int n = d->radius * 2;
    const mvo_analysis_data *ad = &d->ad;
    const int bps = (d->bits + 7) / 8;
    const int nLogPel = ad->nPel == 4 ? 2 : ad->nPel == 2 ? 1 : 0;
    const int nBlkX = ad->nBlkX, nBlkY = ad->nBlkY;
    int isUsable[12];
    const mvo_vector *vec[12];
    mvo_gof gofs[12];
    for (int r = 0; r < n; r++) {
        isUsable[r] = mvo_blob_is_usable(ad, blobs[r], d->nSCD1, d->nSCD2);
        vec[r] = mvo_blob_level0(ad, blobs[r]);
        if (isUsable[r]) {
            mvo_gof_init(&gofs[r], d->nSuperLevels, d->nWidth[0], d->nHeight[0], d->nSuperPel, d->nSuperHPad, d->nSuperVPad,
                         d->nSuperModeYUV, ad->xRatioUV, ad->yRatioUV, d->bits);
            mvo_gof_update(&gofs[r], (uint8_t *const *)refs[r], refPitch[r], ad->yRatioUV);
        }
    }
    const int dstTempPitch = ((ad->nWidth + 15) / 16) * 16 * bps * 2;
    uint8_t *DstTemp = NULL, *tmpBlock = NULL;
    const int tmpBlockPitch = d->nBlkSizeX[0] * bps;
    int16_t *wins[3] = { NULL, NULL, NULL };
    const int overlap = d->nOverlapX[0] > 0 || d->nOverlapY[0] > 0;
    if (overlap) {
        DstTemp = (uint8_t *)malloc((size_t)dstTempPitch * d->nHeight[0]);
        tmpBlock = (uint8_t *)malloc((size_t)tmpBlockPitch * d->nBlkSizeY[0]);
        wins[0] = (int16_t *)malloc(sizeof(int16_t) * 9 * d->nBlkSizeX[0] * d->nBlkSizeY[0]);
        mvo_over_init(wins[0], d->nBlkSizeX[0], d->nBlkSizeY[0], d->nOverlapX[0], d->nOverlapY[0]);
        if (d->numPlanes > 1) {
            wins[1] = (int16_t *)malloc(sizeof(int16_t) * 9 * d->nBlkSizeX[1] * d->nBlkSizeY[1]);
            mvo_over_init(wins[1], d->nBlkSizeX[1], d->nBlkSizeY[1], d->nOverlapX[1], d->nOverlapY[1]);
            wins[2] = wins[1];
        }
    }

    for (int plane = 0; plane < d->numPlanes; plane++) {
        uint8_t *pDstCur = dst[plane];
        const uint8_t *pSrcCur = src[plane];
        const int W = d->nWidth[plane], H = d->nHeight[plane], bsx = d->nBlkSizeX[plane], bsy = d->nBlkSizeY[plane];
        const int ovx = d->nOverlapX[plane], ovy = d->nOverlapY[plane], WB = d->nWidth_B[plane], HB = d->nHeight_B[plane];
        if (!d->process[plane]) { /* :211-214 */
            memcpy(pDstCur, pSrcCur, (size_t)srcPitch[plane] * H);
            continue;
        }
        uint8_t *pDstTemp = DstTemp;
        if (overlap) memset(DstTemp, 0, (size_t)dstTempPitch * d->nHeight_B[0]);
        for (int by = 0; by < nBlkY; by++) {
            int wby = overlap ? ((by + nBlkY - 3) / (nBlkY - 2)) * 3 : 0;
            int wbx = 0, xx = 0;
            for (int bx = 0; bx < nBlkX; bx++) {
                int i = by * nBlkX + bx;
                const uint8_t *pointers[12]; int strides[12]; int WSrc, WRefs[12];
                for (int r = 0; r < n; r++) { /* MVDegrains.h:192-206 useBlock */
                    if (isUsable[r]) {
                        int blkx0 = bx * (ad->nBlkSizeX - ad->nOverlapX), blky0 = by * (ad->nBlkSizeY - ad->nOverlapY); /* Fakery.c:31-32 */
                        int blx = (blkx0 << nLogPel) + vec[r][i].x, bly = (blky0 << nLogPel) + vec[r][i].y;
                        const mvo_plane *mp = &gofs[r].fr[0].pl[plane];
                        pointers[r] = mvo_plane_pointer(mp, plane ? blx >> d->xSubUV : blx, plane ? bly >> d->ySubUV : bly);
                        strides[r] = mp->pitch;
                        WRefs[r] = degrain_weight(d->thSAD[plane], vec[r][i].sad);
                    } else { pointers[r] = pSrcCur + xx; strides[r] = srcPitch[plane]; WRefs[r] = 0; }
                }
                normalise_weights(n, &WSrc, WRefs);
                if (!overlap) { /* :216-250 */
                    degrain_block(n, bsx, bsy, bps, pDstCur + xx, dstPitch[plane], pSrcCur + xx, srcPitch[plane], pointers, strides, WSrc, WRefs);
                    xx += bsx * bps;
                    if (bx == nBlkX - 1 && d->nWidth_B[0] < d->nWidth[0])
                        bitblt(pDstCur + WB * bps, dstPitch[plane], pSrcCur + WB * bps, srcPitch[plane], (W - WB) * bps, bsy);
                } else { /* :251-286 */
                    wbx = bx == nBlkX - 1 ? 2 : wbx;
                    const int16_t *winOver = wins[plane] + bsx * bsy * (wby + wbx);
                    degrain_block(n, bsx, bsy, bps, tmpBlock, tmpBlockPitch, pSrcCur + xx, srcPitch[plane], pointers, strides, WSrc, WRefs);
                    mvo_overlaps(bsx, bsy, d->bits, pDstTemp + xx * 2, dstTempPitch, tmpBlock, tmpBlockPitch, winOver, bsx);
                    xx += (bsx - ovx) * bps;
                    wbx = 1;
                }
            }
            if (!overlap) {
                pDstCur += bsy * dstPitch[plane];
                pSrcCur += bsy * srcPitch[plane];
                if (by == nBlkY - 1 && d->nHeight_B[0] < d->nHeight[0])
                    bitblt(pDstCur, dstPitch[plane], pSrcCur, srcPitch[plane], W * bps, H - HB);
            } else {
                pSrcCur += (bsy - ovy) * srcPitch[plane];
                pDstTemp += (bsy - ovy) * dstTempPitch;
            }
        }
        if (overlap) { /* :288-298 */
            mvo_to_pixels(d->bits, dst[plane], dstPitch[plane], DstTemp, dstTempPitch, WB, HB);
            if (d->nWidth_B[0] < d->nWidth[0])
                bitblt(dst[plane] + WB * bps, dstPitch[plane], src[plane] + WB * bps, srcPitch[plane], (W - WB) * bps, HB);
            if (d->nHeight_B[0] < d->nHeight[0])
                bitblt(dst[plane] + (size_t)dstPitch[plane] * HB, dstPitch[plane], src[plane] + (size_t)srcPitch[plane] * HB, srcPitch[plane], W * bps, H - HB);
        }
        int pixelMax = (1 << d->bits) - 1;
        if (d->nLimit[plane] < pixelMax) /* :301-305 */
            limit_changes(bps, dst[plane], dstPitch[plane], src[plane], srcPitch[plane], W, H, d->nLimit[plane]);
    }
    free(DstTemp); free(tmpBlock); free(wins[0]); free(wins[1]);
}

/* ------------------------------------------------------------------ mv.Compensate */

/* MVCompensate.c:419-575 mvcompensateCreate */
int mvo_compensate_init(mvo_compensate *d, const mvo_analysis_data *ad, const mvo_super *s, int scbehavior,
                        int64_t thsad, double time, int64_t thscd1, int thscd2, char *err) {
    memset(d, 0, sizeof(*d));
    if (err) err[0] = 0;
    d->ad = *ad;
    d->scBehavior = scbehavior == MVO_UNSET ? 1 : !!scbehavior;
    d->thSAD = thsad == MVO_UNSET ? 10000 : thsad;
    if (time < 0.0 || time > 100.0) FAIL("Compensate: time must be between 0.0 and 100.0 (inclusive).");
    d->nSCD1 = thscd1 == MVO_UNSET ? 400 : thscd1;
    d->nSCD2 = thscd2 == MVO_UNSET ? 130 : thscd2;
    if (d->nSCD1 > 8 * 8 * 255) FAIL("Compensate: thscd1 can be at most %d.", 8 * 8 * 255);
    d->nSuperHPad = s->hpad; d->nSuperVPad = s->vpad; d->nSuperPel = s->pel; d->nSuperModeYUV = s->modeYUV; d->nSuperLevels = s->levels;
    int64_t nSCD1_old = d->nSCD1;
    mvo_scale_thscd(&d->nSCD1, &d->nSCD2, ad);
    d->thSAD = d->thSAD * d->nSCD1 / nSCD1_old; /* :521 */
    if (ad->nHeight != s->height || ad->nWidth != s->superWidth - s->hpad * 2 || ad->nWidth != s->width || ad->nPel != s->pel)
        FAIL("Compensate: wrong source or super clip frame size.");
    d->time256 = (int)(time * 256 / 100); /* :560 */
    d->bits = s->bits; d->numPlanes = s->gray ? 1 : 3;
    return 0;
}

/* MVCompensate.c:73-374 mvcompensateGetFrame; fieldShift is what :188-225 derives from the _Field props / tff */
void mvo_compensate_frame(const mvo_compensate *d, const uint8_t *const srcSuper[3], const int srcPitch[3],
                          const uint8_t *const refSuper[3], const int refPitch[3], const uint8_t *blob,
                          uint8_t *const dst[3], const int dstPitch[3], int fieldShift) {
    const mvo_analysis_data *ad = &d->ad;
    const int bps = (d->bits + 7) / 8;
    const int xSubUV = ad->xRatioUV == 2, ySubUV = ad->yRatioUV == 2;
    int nWidth[3], nHeight[3], nOverlapX[3], nOverlapY[3], nBlkSizeX[3], nBlkSizeY[3], nHPadding[3], nVPadding[3], nWidth_B[3], nHeight_B[3];
    nWidth[0] = ad->nWidth; nHeight[0] = ad->nHeight; nOverlapX[0] = ad->nOverlapX; nOverlapY[0] = ad->nOverlapY;
    nBlkSizeX[0] = ad->nBlkSizeX; nBlkSizeY[0] = ad->nBlkSizeY; nHPadding[0] = ad->nHPadding; nVPadding[0] = ad->nVPadding;
    const int nBlkX = ad->nBlkX, nBlkY = ad->nBlkY, nPel = ad->nPel;
    nWidth_B[0] = nBlkX * (nBlkSizeX[0] - nOverlapX[0]) + nOverlapX[0];
    nHeight_B[0] = nBlkY * (nBlkSizeY[0] - nOverlapY[0]) + nOverlapY[0];
    for (int p = 1; p < 3; p++) {
        nWidth[p] = nWidth[0] >> xSubUV; nHeight[p] = nHeight[0] >> ySubUV; nOverlapX[p] = nOverlapX[0] >> xSubUV; nOverlapY[p] = nOverlapY[0] >> ySubUV;
        nBlkSizeX[p] = nBlkSizeX[0] >> xSubUV; nBlkSizeY[p] = nBlkSizeY[0] >> ySubUV; nHPadding[p] = nHPadding[0] >> xSubUV; nVPadding[p] = nVPadding[0] >> ySubUV;
        nWidth_B[p] = nWidth_B[0] >> xSubUV; nHeight_B[p] = nHeight_B[0] >> ySubUV;
    }
    const int dstTempPitch[3] = { ((ad->nWidth + 15) / 16) * 16 * bps * 2, (((ad->nWidth / ad->xRatioUV) + 15) / 16) * 16 * bps * 2,
                                  (((ad->nWidth / ad->xRatioUV) + 15) / 16) * 16 * bps * 2 };
    int num_planes = (d->nSuperModeYUV & (MVO_UPLANE | MVO_VPLANE)) ? 3 : 1;

    if (refSuper && mvo_blob_is_usable(ad, blob, d->nSCD1, d->nSCD2)) {
        const mvo_vector *vec = mvo_blob_level0(ad, blob);
        mvo_gof rg, sg;
        mvo_gof_init(&rg, d->nSuperLevels, nWidth[0], nHeight[0], d->nSuperPel, d->nSuperHPad, d->nSuperVPad, d->nSuperModeYUV, ad->xRatioUV, ad->yRatioUV, d->bits);
        mvo_gof_init(&sg, d->nSuperLevels, nWidth[0], nHeight[0], d->nSuperPel, d->nSuperHPad, d->nSuperVPad, d->nSuperModeYUV, ad->xRatioUV, ad->yRatioUV, d->bits);
        mvo_gof_update(&rg, (uint8_t *const *)refSuper, refPitch, ad->yRatioUV);
        mvo_gof_update(&sg, (uint8_t *const *)srcSuper, srcPitch, ad->yRatioUV);
        const int overlap = nOverlapX[0] != 0 || nOverlapY[0] != 0;
        uint8_t *DstTemp[3] = { NULL, NULL, NULL };
        int16_t *wins[3] = { NULL, NULL, NULL };
        if (overlap) {
            for (int p = 0; p < num_planes; p++) {
                DstTemp[p] = (uint8_t *)malloc((size_t)nHeight[p] * dstTempPitch[p]);
                memset(DstTemp[p], 0, (size_t)nHeight_B[p] * dstTempPitch[p]);
            }
            wins[0] = (int16_t *)malloc(sizeof(int16_t) * 9 * nBlkSizeX[0] * nBlkSizeY[0]);
            mvo_over_init(wins[0], nBlkSizeX[0], nBlkSizeY[0], nOverlapX[0], nOverlapY[0]);
            if (num_planes > 1) {
                wins[1] = (int16_t *)malloc(sizeof(int16_t) * 9 * nBlkSizeX[1] * nBlkSizeY[1]);
                mvo_over_init(wins[1], nBlkSizeX[1], nBlkSizeY[1], nOverlapX[1], nOverlapY[1]);
                wins[2] = wins[1];
            }
        }
        for (int by = 0; by < nBlkY; by++) {
            int wby = overlap ? ((by + nBlkY - 3) / (nBlkY - 2)) * 3 : 0, wbx = 0;
            for (int bx = 0; bx < nBlkX; bx++) {
                int i = by * nBlkX + bx;
                int blx[3], bly[3];
                const mvo_frame *fr;
                int stepx = overlap ? (nBlkSizeX[0] - nOverlapX[0]) : nBlkSizeX[0], stepy = overlap ? (nBlkSizeY[0] - nOverlapY[0]) : nBlkSizeY[0];
                if (vec[i].sad < d->thSAD) { /* :238-242, :286-290 */
                    int blockx = bx * (nBlkSizeX[0] - nOverlapX[0]), blocky = by * (nBlkSizeY[0] - nOverlapY[0]);
                    blx[0] = blockx * nPel + vec[i].x * d->time256 / 256;
                    bly[0] = blocky * nPel + vec[i].y * d->time256 / 256 + fieldShift;
                    fr = &rg.fr[0];
                } else {
                    blx[0] = bx * stepx * nPel;
                    bly[0] = by * stepy * nPel + fieldShift;
                    fr = &sg.fr[0];
                }
                blx[1] = blx[2] = blx[0] >> xSubUV;
                bly[1] = bly[2] = bly[0] >> ySubUV;
                wbx = bx == nBlkX - 1 ? 2 : wbx;
                for (int p = 0; p < num_planes; p++) {
                    const mvo_plane *mp = &fr->pl[p];
                    const uint8_t *ptr = mvo_plane_pointer(mp, blx[p], bly[p]);
                    if (!overlap) {
                        uint8_t *o = dst[p] + (size_t)by * nBlkSizeY[p] * dstPitch[p] + bx * nBlkSizeX[p] * bps;
                        bitblt(o, dstPitch[p], ptr, mp->pitch, nBlkSizeX[p] * bps, nBlkSizeY[p]);
                    } else {
                        uint8_t *o = DstTemp[p] + (size_t)by * (nBlkSizeY[p] - nOverlapY[p]) * dstTempPitch[p] + bx * (nBlkSizeX[p] - nOverlapX[p]) * bps * 2;
                        mvo_overlaps(nBlkSizeX[p], nBlkSizeY[p], d->bits, o, dstTempPitch[p], ptr, mp->pitch,
                                     wins[p] + nBlkSizeX[p] * nBlkSizeY[p] * (wby + wbx), nBlkSizeX[p]);
                    }
                }
                wbx = 1;
            }
        }
        if (overlap)
            for (int p = 0; p < num_planes; p++) {
                mvo_to_pixels(d->bits, dst[p], dstPitch[p], DstTemp[p], dstTempPitch[p], nWidth_B[p], nHeight_B[p]);
                free(DstTemp[p]);
            }
        free(wins[0]); free(wins[1]);
        const uint8_t *const *scSrc = d->scBehavior ? srcSuper : refSuper; /* :319-342 */
        const int *scPitches = d->scBehavior ? srcPitch : refPitch;
        for (int p = 0; p < num_planes; p++) {
            if (nWidth_B[0] < nWidth[0])
                bitblt(dst[p] + nWidth_B[p] * bps, dstPitch[p], scSrc[p] + (nWidth_B[p] + nHPadding[p]) * bps + (size_t)nVPadding[p] * scPitches[p],
                       scPitches[p], (nWidth[p] - nWidth_B[p]) * bps, nHeight_B[p]);
            if (nHeight_B[0] < nHeight[0])
                bitblt(dst[p] + (size_t)nHeight_B[p] * dstPitch[p], dstPitch[p], scSrc[p] + nHPadding[p] * bps + (size_t)(nHeight_B[p] + nVPadding[p]) * scPitches[p],
                       scPitches[p], nWidth[p] * bps, nHeight[p] - nHeight_B[p]);
        }
    } else { /* :348-364 */
        const uint8_t *const *s = srcSuper; const int *sp = srcPitch;
        if (!d->scBehavior && refSuper) { s = refSuper; sp = refPitch; }
        for (int p = 0; p < num_planes; p++)
            bitblt(dst[p], dstPitch[p], s[p] + nHPadding[p] * bps + (size_t)nVPadding[p] * sp[p], sp[p], nWidth[p] * bps, nHeight[p]);
    }
}

/* MVCompensate.c:188-225 (Analyse: MVAnalyse.c:135-176, same arithmetic with nDeltaFrame % 2 for the parity test) */
int mvo_field_shift(int fields, int pel, int n, int nref, int src_field, int ref_field, int tff, int *missing) {
    *missing = 0;
    if (!(fields && pel > 1 && ((nref - n) % 2 != 0))) return 0;
    int src_top = src_field > 0, ref_top = ref_field > 0;
    if ((src_field < 0 || ref_field < 0) && tff < 0) { *missing = 1; return 0; }
    if (tff >= 0) { src_top = tff ^ (n % 2); ref_top = tff ^ (nref % 2); }
    return (src_top && !ref_top) ? pel / 2 : ((ref_top && !src_top) ? -(pel / 2) : 0);
}

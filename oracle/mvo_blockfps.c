/* mvo_blockfps.c -- CPU restatement of mv.BlockFPS (test infrastructure only, see mvoracle.h).
 * Follows /root/reference/src/MVBlockFPS.c (filter :229-676, creation :741-1014), MaskFun.cpp:63-166,349-371
 * (occlusion / SAD masks at intermediate time, padding, Blend) and SimpleResize.cpp:27-121 (the 8-bit mask upsizer).
 * Parity of this file is UNPINNED: none of these reference translation units builds without VapourSynth headers. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "mvo_internal.h"

#define FAIL(...) do { if (err) snprintf(err, MVO_ERR, __VA_ARGS__); return -1; } while (0)
#define VMAX(a, b) ((a) > (b) ? (a) : (b))
#define VMIN(a, b) ((a) < (b) ? (a) : (b))

/* SimpleResize.cpp:27-57 InitTables (float arithmetic exactly as written there) */
void mvo_resize_tables(int *offsets, int *weights, int out, int in) {
    float leftmost = 0.5f;
    float rightmost = in - 0.5f;
    int leftmost_idx = VMAX((int)leftmost, 0);
    int rightmost_idx = VMIN((int)rightmost, in - 1);
    for (int i = 0; i < out; i++) {
        float position = (i + 0.5f) * (float)in / (float)out;
        float weight; int offset;
        if (position <= leftmost) { offset = leftmost_idx; weight = 0.0f; }
        else if (position >= rightmost) { offset = rightmost_idx - 1; weight = 1.0f; }
        else { offset = (int)(position - leftmost); weight = position - leftmost - offset; }
        offsets[i] = offset;
        weights[i] = (int)(weight * 16384);
    }
}

/* SimpleResize.cpp:62-121 simpleResize<uint8_t> (no vector limiting for 8-bit data) */
static void simple_resize_u8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int dw, int dh, int sw, int sh) {
    int *vo = (int *)malloc(sizeof(int) * dh), *vw = (int *)malloc(sizeof(int) * dh), *ho = (int *)malloc(sizeof(int) * dw), *hw = (int *)malloc(sizeof(int) * dw);
    mvo_resize_tables(ho, hw, dw, sw);
    mvo_resize_tables(vo, vw, dh, sh);
    int *work = (int *)malloc(sizeof(int) * sw);
    for (int y = 0; y < dh; y++) {
        const int wb = vw[y], wt = 16384 - wb;
        const uint8_t *s1 = src + vo[y] * srcStride, *s2 = s1 + srcStride;
        for (int x = 0; x < sw; x++) work[x] = (uint8_t)((s1[x] * wt + s2[x] * wb + 8192) >> 14);
        for (int x = 0; x < dw; x++) {
            const int wr = hw[x], wl = 16384 - wr, o = ho[x];
            dst[x] = (uint8_t)((work[o] * wl + work[o + 1] * wr + 8192) >> 14);
        }
        dst += dstStride;
    }
    free(vo); free(vw); free(ho); free(hw); free(work);
}

void mvo_simple_resize_u8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int dw, int dh, int sw, int sh) { simple_resize_u8(dst, dstStride, src, srcStride, dw, dh, sw, sh); }

/* MaskFun.cpp:83-89 */
static void byte_occ_mask(uint8_t *m, int occlusion, double occnorm, double gamma) {
    int v = gamma == 1.0 ? VMIN((int)(255 * occlusion * occnorm), 255) : VMIN((int)(255 * pow(occlusion * occnorm, gamma)), 255);
    *m = (uint8_t)VMAX((int)*m, v);
}
/* MaskFun.cpp:91-130 MakeVectorOcclusionMaskTime */
static void occlusion_mask_time(const mvo_vector *vec, int isBackward, int nBlkX, int nBlkY, double divider, double gamma, int nPel, uint8_t *occ, int pitch,
                                int time256, int stepX, int stepY) {
    memset(occ, 0, (size_t)pitch * nBlkY);
    const int tX = time256 * 16 / (stepX * nPel), tY = time256 * 16 / (stepY * nPel);
    const double nX = 80.0 / (divider * stepX * nPel), nY = 80.0 / (divider * stepY * nPel);
    for (int by = 0; by < nBlkY; by++)
        for (int bx = 0; bx < nBlkX; bx++) {
            const int i = bx + by * nBlkX, vx = vec[i].x, vy = vec[i].y;
            if (bx < nBlkX - 1) {
                const int vx1 = vec[i + 1].x;
                if (vx1 < vx) {
                    const int o = vx - vx1;
                    const int minb = isBackward ? VMAX(0, bx + 1 - o * tX / 4096) : bx;
                    const int maxb = isBackward ? bx + 1 : VMIN(bx + 1 - o * tX / 4096, nBlkX - 1);
                    for (int b = minb; b <= maxb; b++) byte_occ_mask(&occ[b + by * pitch], o, nX, gamma);
                }
            }
            if (by < nBlkY - 1) {
                const int vy1 = vec[i + nBlkX].y;
                if (vy1 < vy) {
                    const int o = vy - vy1;
                    const int minb = isBackward ? VMAX(0, by + 1 - o * tY / 4096) : by;
                    const int maxb = isBackward ? by + 1 : VMIN(by + 1 - o * tY / 4096, nBlkY - 1);
                    for (int b = minb; b <= maxb; b++) byte_occ_mask(&occ[bx + b * pitch], o, nY, gamma);
                }
            }
        }
}
/* MaskFun.cpp:133-166 ByteNorm + MakeSADMaskTime */
static void sad_mask_time(const mvo_vector *vec, int nBlkX, int nBlkY, double factor, double gamma, int nPel, uint8_t *mask, int pitch, int time256,
                          int stepX, int stepY, int bits) {
    memset(mask, 0, (size_t)pitch * nBlkY);
    const int tX = (256 - time256) * 16 / (stepX * nPel), tY = (256 - time256) * 16 / (stepY * nPel);
    for (int by = 0; by < nBlkY; by++)
        for (int bx = 0; bx < nBlkX; bx++) {
            const int i = bx + by * nBlkX;
            int bxi = bx - vec[i].x * tX / 4096, byi = by - vec[i].y * tY / 4096;
            if (bxi < 0 || bxi >= nBlkX || byi < 0 || byi >= nBlkY) { bxi = bx; byi = by; }
            const int64_t sad = vec[bxi + byi * nBlkX].sad >> (bits - 8);
            const double l = 255 * pow(sad * factor, gamma);
            mask[bx + by * pitch] = (unsigned char)((l > 255) ? 255 : l);
        }
}
/* MaskFun.cpp:63-80 CheckAndPadMaskSmall */
static void pad_mask_small(uint8_t *m, int xp, int yp, int nx, int ny) {
    if (xp > nx) for (int j = 0; j < ny; j++) { const uint8_t r = m[j * xp + nx - 1]; for (int dx = nx; dx < xp; dx++) m[j * xp + dx] = r; }
    if (yp > ny) for (int i = 0; i < xp; i++) { const uint8_t b = m[xp * (ny - 1) + i]; for (int dy = ny; dy < yp; dy++) m[xp * dy + i] = b; }
}
/* MaskFun.cpp:349-371 Blend */
static void blend(uint8_t *d, const uint8_t *s, const uint8_t *r, int h, int w, int dp, int sp, int rp, int time256, int bits) {
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            if (bits == 8) d[x] = (uint8_t)((s[x] * (256 - time256) + r[x] * time256) >> 8);
            else ((uint16_t *)d)[x] = (uint16_t)((((const uint16_t *)s)[x] * (256 - time256) + ((const uint16_t *)r)[x] * time256) >> 8);
        }
        d += dp; s += sp; r += rp;
    }
}
static int median3(int a, int b, int c) { int mn = VMIN(a, b), mx = VMAX(a, b), m = VMIN(mx, c); return VMAX(mn, m); }

/* MVBlockFPS.c:117-227 RealResultBlock */
static void result_block(uint8_t *pDst, int dp, const uint8_t *pMCB, int bp, const uint8_t *pMCF, int fp, const uint8_t *pRef, int rp, const uint8_t *pSrc, int sp,
                         const uint8_t *maskB, int mp, const uint8_t *maskF, const uint8_t *pOcc, int bw, int bh, int time256, int mode, int bits) {
#define PX(p, x) (bits == 8 ? (int)(p)[x] : (int)((const uint16_t *)(p))[x])
#define ST(x, v) do { if (bits == 8) pDst[x] = (uint8_t)(v); else ((uint16_t *)pDst)[x] = (uint16_t)(v); } while (0)
    for (int h = 0; h < bh; h++) {
        for (int w = 0; w < bw; w++) {
            const int b = PX(pMCB, w), f = PX(pMCF, w);
            if (mode == 0) ST(w, (b * time256 + f * (256 - time256)) >> 8);
            else if (mode == 1) { int mca = (b * time256 + f * (256 - time256)) >> 8; ST(w, median3(PX(pRef, w), PX(pSrc, w), bits == 8 ? (uint8_t)mca : (uint16_t)mca)); }
            else if (mode == 2) { int avg = (PX(pRef, w) * time256 + PX(pSrc, w) * (256 - time256)) >> 8; ST(w, median3(bits == 8 ? (uint8_t)avg : (uint16_t)avg, b, f)); }
            else if (mode == 3 || mode == 6)
                ST(w, (((maskB[w] * f + (255 - maskB[w]) * b + 255) >> 8) * time256 + ((maskF[w] * b + (255 - maskF[w]) * f + 255) >> 8) * (256 - time256)) >> 8);
            else if (mode == 4 || mode == 7) {
                int ff = (maskF[w] * b + (255 - maskF[w]) * f + 255) >> 8;
                int bb = (maskB[w] * f + (255 - maskB[w]) * b + 255) >> 8;
                int avg = (PX(pRef, w) * time256 + PX(pSrc, w) * (256 - time256) + 255) >> 8;
                int m = (bb * time256 + ff * (256 - time256)) >> 8;
                ST(w, (avg * pOcc[w] + m * (255 - pOcc[w]) + 255) >> 8);
            } else ST(w, pOcc[w] << (bits - 8));
        }
        pDst += dp; pMCB += bp; pMCF += fp; pRef += rp; pSrc += sp; maskB += mp; maskF += mp; pOcc += mp;
    }
#undef PX
#undef ST
}

static int64_t gcd64(int64_t x, int64_t y) { while (y) { int64_t t = x % y; x = y; y = t; } return x; }

/* MVBlockFPS.c:741-987 (argument handling; the clip's frame rate comes in as fpsNum / fpsDen) */
int mvo_blockfps_init(mvo_blockfps *d, const mvo_analysis_data *bw, const mvo_analysis_data *fw, const mvo_super *s, int numFrames, int64_t fpsNum, int64_t fpsDen,
                      int64_t num, int64_t den, int mode, double ml, int blendArg, int64_t thscd1, int thscd2, char *err) {
    memset(d, 0, sizeof(*d));
    if (err) err[0] = 0;
    if (num == MVO_UNSET) num = 25;
    if (den == MVO_UNSET) den = 1;
    d->mode = mode == MVO_UNSET ? 3 : mode;
    d->ml = ml;
    d->blend = blendArg == MVO_UNSET ? 1 : !!blendArg;
    d->thscd1 = thscd1 == MVO_UNSET ? 400 : thscd1;
    d->thscd2 = thscd2 == MVO_UNSET ? 130 : thscd2;
    if (d->mode < 0 || d->mode > 8) FAIL("BlockFPS: mode must be between 0 and 8 (inclusive).");
    d->bw = *bw; d->fw = *fw;
    if (d->thscd1 > 8 * 8 * 255) FAIL("BlockFPS: thscd1 can be at most %d.", 8 * 8 * 255);
    mvo_scale_thscd(&d->thscd1, &d->thscd2, bw);
    if (bw->nWidth != fw->nWidth) FAIL("BlockFPS: mvbw and mvfw have different widths.");
    if (bw->nHeight != fw->nHeight) FAIL("BlockFPS: mvbw and mvfw have different heights.");
    if (bw->nBlkSizeX != fw->nBlkSizeX || bw->nBlkSizeY != fw->nBlkSizeY) FAIL("BlockFPS: mvbw and mvfw have different block sizes.");
    if (bw->nPel != fw->nPel) FAIL("BlockFPS: mvbw and mvfw have different pel precision.");
    if (bw->nOverlapX != fw->nOverlapX || bw->nOverlapY != fw->nOverlapY) FAIL("BlockFPS: mvbw and mvfw have different overlap.");
    if (bw->nDeltaFrame <= 0 || fw->nDeltaFrame <= 0) FAIL("BlockFPS: cannot use motion vectors with absolute frame references.");
    if (bw->nDeltaFrame != fw->nDeltaFrame) FAIL("BlockFPS: mvbw and mvfw must be generated with the same delta.");
    if (!bw->isBackward) FAIL("BlockFPS: mvbw must be generated with isb=True.");
    if (fw->isBackward) FAIL("BlockFPS: mvfw must be generated with isb=False.");
    if (fpsNum == 0 || fpsDen == 0) FAIL("BlockFPS: The input clip must have a frame rate. Invoke AssumeFPS if necessary.");
    int64_t numerator, denominator;
    if (num != 0 && den != 0) { numerator = num; denominator = den; } else { numerator = fpsNum * 2; denominator = fpsDen; }
    d->fa = denominator * fpsNum; d->fb = numerator * fpsDen;
    const int64_t g = gcd64(d->fa, d->fb);
    d->fa /= g; d->fb /= g;
    if (numerator <= 0 || denominator <= 0) { d->outFpsNum = 0; d->outFpsDen = 1; }
    else { const int64_t x = gcd64(numerator, denominator); d->outFpsNum = numerator / x; d->outFpsDen = denominator / x; }
    d->inFrames = numFrames;
    d->outFrames = (int)(1 + (numFrames - 1) * d->fb / d->fa);
    if (bw->nHeight != s->height || bw->nWidth != s->superWidth - s->hpad * 2 || bw->nWidth != s->width || bw->nPel != s->pel)
        FAIL("BlockFPS: wrong source or super clip frame size.");
    d->nSuperHPad = s->hpad; d->nSuperVPad = s->vpad; d->nSuperPel = s->pel; d->nSuperModeYUV = s->modeYUV; d->nSuperLevels = s->levels;
    d->bits = s->bits;
    d->nBlkXP = bw->nBlkX; d->nBlkYP = bw->nBlkY;
    while (d->nBlkXP * (bw->nBlkSizeX - bw->nOverlapX) + bw->nOverlapX < bw->nWidth) d->nBlkXP++;
    while (d->nBlkYP * (bw->nBlkSizeY - bw->nOverlapY) + bw->nOverlapY < bw->nHeight) d->nBlkYP++;
    d->nWidthP = d->nBlkXP * (bw->nBlkSizeX - bw->nOverlapX) + bw->nOverlapX;
    d->nHeightP = d->nBlkYP * (bw->nBlkSizeY - bw->nOverlapY) + bw->nOverlapY;
    d->nWidthPUV = d->nWidthP / bw->xRatioUV; d->nHeightPUV = d->nHeightP / bw->yRatioUV;
    d->nPitchY = (d->nWidthP + 15) & ~15; d->nPitchUV = (d->nWidthPUV + 15) & ~15;
    return 0;
}

/* MVBlockFPS.c:245-254,278-292: output frame n -> (nleft, nright, time256) */
void mvo_blockfps_map(const mvo_blockfps *d, int n, int *nleft, int *nright, int *time256) {
    const int off = d->bw.nDeltaFrame;
    *nleft = (int)(n * d->fa / d->fb);
    int t = (int)(((double)n * d->fa / d->fb - *nleft) * 256 + 0.5);
    if (off > 1) t = t / off;
    *nright = *nleft + off;
    *time256 = t;
}

/* MVBlockFPS.c:278-673 for 0 < time256 < 256.  srcSuper = super[nleft], refSuper = super[nright], blobF = mvfw vectors at
 * nright, blobB = mvbw vectors at nleft (all NULL when nleft / nright fall outside the clip); clipL / clipR = the clip's
 * frames min(nleft, last) / min(nright, last) for the poor-estimation fallback.  Returns 0 when dst was written, 1 when the
 * result is simply clipL (blend=0 fallback). */
int mvo_blockfps_frame(const mvo_blockfps *d, int time256, const uint8_t *const srcSuper[3], const int srcPitch[3], const uint8_t *const refSuper[3],
                       const int refPitch[3], const uint8_t *blobF, const uint8_t *blobB, const uint8_t *const clipL[3], const int clipLPitch[3],
                       const uint8_t *const clipR[3], const int clipRPitch[3], uint8_t *const dst[3], const int dstPitch[3]) {
    const mvo_analysis_data *ad = &d->bw;
    const int bps = (d->bits + 7) / 8, bits = d->bits;
    const int xr = ad->xRatioUV, yr = ad->yRatioUV, nBlkX = ad->nBlkX, nBlkY = ad->nBlkY, nPel = ad->nPel, mode = d->mode;
    const int planes = (d->nSuperModeYUV & (MVO_UPLANE | MVO_VPLANE)) ? 3 : 1;
    const int nWidth[3] = { ad->nWidth, ad->nWidth / xr, ad->nWidth / xr }, nHeight[3] = { ad->nHeight, ad->nHeight / yr, ad->nHeight / yr };
    const int bsx[3] = { ad->nBlkSizeX, ad->nBlkSizeX / xr, ad->nBlkSizeX / xr }, bsy[3] = { ad->nBlkSizeY, ad->nBlkSizeY / yr, ad->nBlkSizeY / yr };
    const int ovx[3] = { ad->nOverlapX, ad->nOverlapX / xr, ad->nOverlapX / xr }, ovy[3] = { ad->nOverlapY, ad->nOverlapY / yr, ad->nOverlapY / yr };
    const int nPitch[3] = { d->nPitchY, d->nPitchUV, d->nPitchUV };
    const int nWidth_B0 = nBlkX * (bsx[0] - ovx[0]) + ovx[0], nHeight_B0 = nBlkY * (bsy[0] - ovy[0]) + ovy[0];
    const int nWidth_B[3] = { nWidth_B0, nWidth_B0 / xr, nWidth_B0 / xr }, nHeight_B[3] = { nHeight_B0, nHeight_B0 / yr, nHeight_B0 / yr };
    const int xRatio[3] = { 1, xr, xr }, yRatio[3] = { 1, yr, yr };

    int usableF = 0, usableB = 0;
    if (blobF && blobB && srcSuper && refSuper) {
        usableF = mvo_blob_is_usable(&d->fw, blobF, d->thscd1, d->thscd2);
        usableB = mvo_blob_is_usable(&d->bw, blobB, d->thscd1, d->thscd2);
    }
    if (!(usableB && usableF)) { /* poor estimation :640-673 */
        if (!d->blend) return 1;
        for (int p = 0; p < planes; p++) blend(dst[p], clipL[p], clipR[p], nHeight[p], nWidth[p], dstPitch[p], clipLPitch[p], clipRPitch[p], time256, bits);
        return 0;
    }
    const mvo_vector *vecF = mvo_blob_level0(&d->fw, blobF), *vecB = mvo_blob_level0(&d->bw, blobB);
    mvo_gof gB, gF;
    mvo_gof_init(&gB, d->nSuperLevels, nWidth[0], nHeight[0], d->nSuperPel, d->nSuperHPad, d->nSuperVPad, d->nSuperModeYUV, xr, yr, bits);
    mvo_gof_init(&gF, d->nSuperLevels, nWidth[0], nHeight[0], d->nSuperPel, d->nSuperHPad, d->nSuperVPad, d->nSuperModeYUV, xr, yr, bits);
    mvo_gof_update(&gB, (uint8_t *const *)refSuper, refPitch, yr);
    mvo_gof_update(&gF, (uint8_t *const *)srcSuper, srcPitch, yr);
    const mvo_plane *plB = gB.fr[0].pl, *plF = gF.fr[0].pl;

    uint8_t *MaskFullB[3] = { NULL, NULL, NULL }, *MaskFullF[3] = { NULL, NULL, NULL }, *MaskOcc[3] = { NULL, NULL, NULL };
    MaskFullB[0] = (uint8_t *)calloc((size_t)d->nHeightP * nPitch[0], 1); MaskFullF[0] = (uint8_t *)calloc((size_t)d->nHeightP * nPitch[0], 1);
    MaskOcc[0] = (uint8_t *)calloc((size_t)d->nHeightP * nPitch[0], 1); /* the reference leaves it uninitialised; only read in modes that fill it */
    if (planes > 1) {
        MaskFullB[1] = MaskFullB[2] = (uint8_t *)calloc((size_t)d->nHeightPUV * nPitch[1], 1);
        MaskFullF[1] = MaskFullF[2] = (uint8_t *)calloc((size_t)d->nHeightPUV * nPitch[1], 1);
        MaskOcc[1] = MaskOcc[2] = (uint8_t *)calloc((size_t)d->nHeightPUV * nPitch[1], 1);
    }
    const int XP = d->nBlkXP, YP = d->nBlkYP;
    if (mode >= 3) {
        uint8_t *sB = (uint8_t *)calloc((size_t)XP * YP, 1), *sF = (uint8_t *)calloc((size_t)XP * YP, 1), *sO = (uint8_t *)calloc((size_t)XP * YP, 1);
        if (mode <= 5) {
            occlusion_mask_time(vecF, 0, nBlkX, nBlkY, d->ml, 1.0, nPel, sF, XP, time256, bsx[0] - ovx[0], bsy[0] - ovy[0]);
            occlusion_mask_time(vecB, 1, nBlkX, nBlkY, d->ml, 1.0, nPel, sB, XP, 256 - time256, bsx[0] - ovx[0], bsy[0] - ovy[0]);
        } else {
            sad_mask_time(vecF, nBlkX, nBlkY, 4.0 / (d->ml * bsx[0] * bsy[0]), 1.0, nPel, sF, XP, time256, bsx[0] - ovx[0], bsy[0] - ovy[0], bits);
            sad_mask_time(vecB, nBlkX, nBlkY, 4.0 / (d->ml * bsx[0] * bsy[0]), 1.0, nPel, sB, XP, 256 - time256, bsx[0] - ovx[0], bsy[0] - ovy[0], bits);
        }
        pad_mask_small(sF, XP, YP, nBlkX, nBlkY);
        pad_mask_small(sB, XP, YP, nBlkX, nBlkY);
        simple_resize_u8(MaskFullF[0], nPitch[0], sF, XP, d->nWidthP, d->nHeightP, XP, YP);
        simple_resize_u8(MaskFullB[0], nPitch[0], sB, XP, d->nWidthP, d->nHeightP, XP, YP);
        if (planes > 1) {
            simple_resize_u8(MaskFullF[1], nPitch[1], sF, XP, d->nWidthPUV, d->nHeightPUV, XP, YP);
            simple_resize_u8(MaskFullB[1], nPitch[1], sB, XP, d->nWidthPUV, d->nHeightPUV, XP, YP);
        }
        if (mode == 4 || mode == 5 || mode == 7 || mode == 8) {
            for (int i = 0; i < XP * YP; i++) sO[i] = (uint8_t)((sF[i] * sB[i]) / 255);
            simple_resize_u8(MaskOcc[0], nPitch[0], sO, XP, d->nWidthP, d->nHeightP, XP, YP);
            if (planes > 1) simple_resize_u8(MaskOcc[1], nPitch[1], sO, XP, d->nWidthPUV, d->nHeightPUV, XP, YP);
        }
        free(sB); free(sF); free(sO);
    }
    /* un-padded level-0 plane 0 of the two super frames (:455-464; the chroma offset is the reference's >> 1) */
    const uint8_t *pSrc[3], *pRef[3];
    pSrc[0] = srcSuper[0] + d->nSuperHPad * bps + srcPitch[0] * d->nSuperVPad;
    pRef[0] = refSuper[0] + d->nSuperHPad * bps + refPitch[0] * d->nSuperVPad;
    for (int p = 1; p < planes; p++) {
        pSrc[p] = srcSuper[p] + (d->nSuperHPad >> 1) * bps + srcPitch[p] * (d->nSuperVPad >> 1);
        pRef[p] = refSuper[p] + (d->nSuperHPad >> 1) * bps + refPitch[p] * (d->nSuperVPad >> 1);
    }
    const int overlap = ovx[0] != 0 || ovy[0] != 0;
    if (!overlap) {
        for (int p = 0; p < planes; p++) {
            for (int by = 0; by < nBlkY; by++)
                for (int bx = 0; bx < nBlkX; bx++) {
                    const int i = by * nBlkX + bx;
                    const int x = bx * bsx[0], y = by * bsy[0]; /* FakeBlockData x, y (luma) */
                    const uint8_t *mcb = mvo_plane_pointer(&plB[p], (x * nPel + ((vecB[i].x * (256 - time256)) >> 8)) / xRatio[p], (y * nPel + ((vecB[i].y * (256 - time256)) >> 8)) / yRatio[p]);
                    const uint8_t *mcf = mvo_plane_pointer(&plF[p], (x * nPel + ((vecF[i].x * time256) >> 8)) / xRatio[p], (y * nPel + ((vecF[i].y * time256) >> 8)) / yRatio[p]);
                    const size_t po = (size_t)by * bsy[p], xo = (size_t)bx * bsx[p];
                    result_block(dst[p] + po * dstPitch[p] + xo * bps, dstPitch[p], mcb, plB[p].pitch, mcf, plF[p].pitch, pRef[p] + po * refPitch[p] + xo * bps, refPitch[p],
                                 pSrc[p] + po * srcPitch[p] + xo * bps, srcPitch[p], MaskFullB[p] + po * nPitch[p] + xo, nPitch[p], MaskFullF[p] + po * nPitch[p] + xo,
                                 MaskOcc[p] + po * nPitch[p] + xo, bsx[p], bsy[p], time256, mode, bits);
                }
            /* rest right (per block row) and rest bottom, blended with the time weight (:513-529) */
            const int wB = bsx[p] * nBlkX, hB = bsy[p] * nBlkY;
            if (nWidth[p] > wB)
                blend(dst[p] + (size_t)wB * bps, pSrc[p] + (size_t)wB * bps, pRef[p] + (size_t)wB * bps, hB, nWidth[p] - wB, dstPitch[p], srcPitch[p], refPitch[p], time256, bits);
            if (nHeight[p] > hB)
                blend(dst[p] + (size_t)hB * dstPitch[p], pSrc[p] + (size_t)hB * srcPitch[p], pRef[p] + (size_t)hB * refPitch[p], nHeight[p] - hB, nWidth[p], dstPitch[p], srcPitch[p], refPitch[p], time256, bits);
        }
    } else {
        for (int p = 0; p < planes; p++) {
            blend(dst[p] + (size_t)nWidth_B[p] * bps, pSrc[p] + (size_t)nWidth_B[p] * bps, pRef[p] + (size_t)nWidth_B[p] * bps, nHeight_B[p], nWidth[p] - nWidth_B[p],
                  dstPitch[p], srcPitch[p], refPitch[p], time256, bits);
            blend(dst[p] + (size_t)dstPitch[p] * nHeight_B[p], pSrc[p] + (size_t)srcPitch[p] * nHeight_B[p], pRef[p] + (size_t)refPitch[p] * nHeight_B[p], nHeight[p] - nHeight_B[p],
                  nWidth[p], dstPitch[p], srcPitch[p], refPitch[p], time256, bits);
            const int tmpPitch = (((ad->nWidth / xRatio[p]) + 15) / 16) * 16 * bps * 2;
            uint8_t *tmp = (uint8_t *)calloc((size_t)tmpPitch * nHeight[p], 1);
            const int blkPitch = ((ad->nBlkSizeX + 15) & ~15) * bps;
            uint8_t *blk = (uint8_t *)malloc((size_t)ad->nBlkSizeY * blkPitch);
            int16_t *win = (int16_t *)malloc(sizeof(int16_t) * 9 * bsx[p] * bsy[p]);
            mvo_over_init(win, bsx[p], bsy[p], ovx[p], ovy[p]);
            for (int by = 0; by < nBlkY; by++) {
                const int wby = ((by + nBlkY - 3) / (nBlkY - 2)) * 3;
                int wbx = 0;
                for (int bx = 0; bx < nBlkX; bx++) {
                    wbx = bx == nBlkX - 1 ? 2 : wbx;
                    const int i = by * nBlkX + bx;
                    const int x = bx * (bsx[0] - ovx[0]), y = by * (bsy[0] - ovy[0]);
                    const uint8_t *mcb = mvo_plane_pointer(&plB[p], (x * nPel + ((vecB[i].x * (256 - time256)) >> 8)) / xRatio[p], (y * nPel + ((vecB[i].y * (256 - time256)) >> 8)) / yRatio[p]);
                    const uint8_t *mcf = mvo_plane_pointer(&plF[p], (x * nPel + ((vecF[i].x * time256) >> 8)) / xRatio[p], (y * nPel + ((vecF[i].y * time256) >> 8)) / yRatio[p]);
                    const size_t po = (size_t)by * (bsy[p] - ovy[p]), xo = (size_t)bx * (bsx[p] - ovx[p]);
                    result_block(blk, blkPitch, mcb, plB[p].pitch, mcf, plF[p].pitch, pRef[p] + po * refPitch[p] + xo * bps, refPitch[p], pSrc[p] + po * srcPitch[p] + xo * bps, srcPitch[p],
                                 MaskFullB[p] + po * nPitch[p] + xo, nPitch[p], MaskFullF[p] + po * nPitch[p] + xo, MaskOcc[p] + po * nPitch[p] + xo, bsx[p], bsy[p], time256, mode, bits);
                    mvo_overlaps(bsx[p], bsy[p], bits, tmp + po * tmpPitch + xo * bps * 2, tmpPitch, blk, blkPitch, win + bsx[p] * bsy[p] * (wby + wbx), bsx[p]);
                    wbx = 1;
                }
            }
            mvo_to_pixels(bits, dst[p], dstPitch[p], tmp, tmpPitch, nWidth_B[p], nHeight_B[p]);
            free(tmp); free(blk); free(win);
        }
    }
    free(MaskFullB[0]); free(MaskFullF[0]); free(MaskOcc[0]);
    if (planes > 1) { free(MaskFullB[1]); free(MaskFullF[1]); free(MaskOcc[1]); }
    return 0;
}

/*
 * mvoracle.h -- CPU restatement (plain C99) of the mvtools hot path:
 *   mv.Super -> mv.Analyse -> mv.DegrainN / mv.Compensate, and mv.BlockFPS.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / reported CPU baseline.  The product (libmvtools_amd.so) never links or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference/src) it restates.
 *
 * PINNING STATUS (see oracle/README.md and DESIGN.md):
 *   - kernel level (SAD, SATD, overlap windows, overlaps accumulate, ToPixels, 8-bit Wiener/bilinear
 *     refine, block copy): checked against oracle/_ref, the reference's own translation units
 *     compiled unmodified from /root/reference (the ones that need no absent header).
 *   - Super+Analyse end to end: checked against the three vector-blob hashes SURVEY.md 8(c) records
 *     from the reference (6d02e770, b899c7de, 9ee5c88e).
 *   - MVFrame.cpp / PlaneOfBlocks.cpp / GroupOfPlanes.c / Fakery.c / MVDegrains.h cannot be compiled
 *     here without stand-ins for the absent VapourSynth headers, so beyond the two items above the
 *     pyramid, the search driver, Degrain weights and Compensate are "parity unpinned".
 *   - mv.BlockFPS (MVBlockFPS.c, MaskFun.cpp, SimpleResize.cpp: same absent headers) is "parity unpinned" entirely.
 */
#ifndef MVORACLE_H
#define MVORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVO_UNSET (-2147483647 - 1) /* "argument not passed" sentinel for optional filter args */
#define MVO_MAX_LEVELS 32
#define MVO_ERR 256

/* wire types: MVAnalysisData.h:40-44, :83-134 */
typedef struct mvo_vector { int x, y; int64_t sad; } mvo_vector;

typedef struct mvo_analysis_data {
    int nMagicKey, nVersion, nBlkSizeX, nBlkSizeY, nPel, nLvCount, nDeltaFrame, isBackward, nCPUFlags,
        nMotionFlags, nWidth, nHeight, nOverlapX, nOverlapY, nBlkX, nBlkY, bitsPerSample, yRatioUV,
        xRatioUV, nHPadding, nVPadding;
} mvo_analysis_data;

/* ---- geometry: MVFrame.cpp:1209-1247 ---- */
int mvo_plane_height_luma(int src_height, int level, int yRatioUV, int vpad);
int mvo_plane_width_luma(int src_width, int level, int xRatioUV, int hpad);
unsigned mvo_plane_super_offset(int chroma, int src_height, int level, int pel, int vpad, int plane_pitch, int yRatioUV);

/* ---- mv.Super: MVSuper.c ---- */
typedef struct mvo_super {
    /* clip */
    int width, height, bits, xRatioUV, yRatioUV, gray;
    /* resolved args */
    int hpad, vpad, pel, levels, chroma, sharp, rfilter;
    int modeYUV;                 /* 1 = Y only, 7 = YUV */
    int superWidth, superHeight; /* luma dims of the super frame */
} mvo_super;

/* args may be MVO_UNSET; returns 0 or -1 with message in err (same strings as MVSuper.c:179-206) */
int mvo_super_init(mvo_super *s, int width, int height, int bits, int subW, int subH, int gray,
                   int hpad, int vpad, int pel, int levels, int chroma, int sharp, int rfilter, char *err);
/* dst planes: pitch[p] * (superHeight >> (p?subH:0)) bytes each; zero-filled inside (MVSuper.c:75) */
void mvo_super_frame(const mvo_super *s, const uint8_t *const src[3], const int srcPitch[3],
                     uint8_t *const dst[3], const int dstPitch[3]);
/* pelclip (MVSuper.c:229-256, :91-102; MVFrame.cpp:1529-1631).  mode: 0 ignored, 1 plain, 2 padded, -1 error */
int mvo_super_pelclip_mode(const mvo_super *s, int pelWidth, int pelHeight, char *err);
void mvo_super_frame_pelclip(const mvo_super *s, const uint8_t *const src[3], const int srcPitch[3], const uint8_t *const pelclip[3],
                             const int pelPitch[3], int pelMode, uint8_t *const dst[3], const int dstPitch[3]);

/* ---- mv.Finest: MVFinest.c (parity unpinned) ---- */
void mvo_finest_size(const mvo_super *s, int *w, int *h);
void mvo_finest_frame(const mvo_super *s, const uint8_t *const sup[3], const int supPitch[3], uint8_t *const dst[3], const int dstPitch[3]);

/* ---- mv.Analyse: MVAnalyse.c, GroupOfPlanes.c, PlaneOfBlocks.cpp ---- */
typedef struct mvo_analyse_args { /* every field may be MVO_UNSET */
    int blksize, blksizev, levels, search, searchparam, pelsearch, isb, lambda, chroma, delta, truemotion,
        lsad, plevel, global, pnew, pzero, pglobal, overlap, overlapv, divide, badsad, badrange, opt,
        meander, trymany, fields, tff, search_coarse, dct;
} mvo_analyse_args;

typedef struct mvo_analyse {
    mvo_analysis_data ad;
    int searchType, searchTypeCoarse, nSearchParam, nPelSearch, nLambda, lsad, pnew, plevel, global, pglobal,
        pzero, divideExtra, badrange, meander, tryMany, dctmode, chroma, fields, tff, tff_exists, opt;
    int64_t badSAD;
    int nSuperLevels, nSuperHPad, nSuperVPad, nSuperPel, nSuperModeYUV;
    int numFrames;
} mvo_analyse;

void mvo_analyse_args_default(mvo_analyse_args *a); /* all MVO_UNSET */
int mvo_analyse_init(mvo_analyse *d, const mvo_analyse_args *a, const mvo_super *s, int numFrames, char *err);
int mvo_analyse_blob_size(const mvo_analyse *d);
/* ref == NULL -> "too close to the clip boundary": default (invalid) blob. fieldShift normally 0. */
void mvo_analyse_frame(const mvo_analyse *d, const uint8_t *const src[3], const int srcPitch[3],
                       const uint8_t *const ref[3], const int refPitch[3], int fieldShift, uint8_t *blob);

void mvo_analysis_data_divided(const mvo_analysis_data *in, mvo_analysis_data *out); /* what readers see with divide > 0 */

/* ---- mv.Recalculate: MVRecalculate.c, PlaneOfBlocks.cpp:1158-1424, GroupOfPlanes.c:127-148 ---- */
typedef struct mvo_recalculate_args { /* MVO_UNSET = not passed */
    int64_t thsad, smooth, blksize, blksizev, search, searchparam, lambda, chroma, truemotion, pnew, overlap, overlapv, divide, meander, dct;
} mvo_recalculate_args;
typedef struct mvo_recalculate { mvo_analyse an; mvo_analysis_data old; int64_t thSAD; int smooth; } mvo_recalculate;
void mvo_recalculate_args_default(mvo_recalculate_args *a);
int mvo_recalculate_init(mvo_recalculate *d, const mvo_recalculate_args *a, const mvo_super *s, const mvo_analysis_data *vectors, char *err);
int mvo_recalculate_blob_size(const mvo_recalculate *d);
void mvo_recalculate_frame(const mvo_recalculate *d, const uint8_t *const src[3], const int srcPitch[3], const uint8_t *const ref[3], const int refPitch[3],
                           const uint8_t *oldBlob, uint8_t *blob);

/* ---- vector blob reader: Fakery.c, MVAnalysisData.c:7-31 ---- */
void mvo_scale_thscd(int64_t *thscd1, int *thscd2, const mvo_analysis_data *ad);
int mvo_blob_is_usable(const mvo_analysis_data *ad, const uint8_t *blob, int64_t thscd1, int thscd2);
const mvo_vector *mvo_blob_level0(const mvo_analysis_data *ad, const uint8_t *blob);

/* ---- overlap windows: Overlap.cpp:40-125 ---- */
void mvo_over_init(int16_t *win9, int nx, int ny, int ox, int oy); /* win9: 9*nx*ny */

/* ---- mv.DegrainN: MVDegrains.cpp/.h ---- */
typedef struct mvo_degrain {
    int radius;
    mvo_analysis_data ad; /* vectors_data[0] */
    int64_t thSAD[3];
    int64_t nSCD1; int nSCD2;
    int nLimit[3];
    int process[3];
    int nSuperHPad, nSuperVPad, nSuperPel, nSuperModeYUV, nSuperLevels;
    int bits, numPlanes, xSubUV, ySubUV;
    int nWidth[3], nHeight[3], nOverlapX[3], nOverlapY[3], nBlkSizeX[3], nBlkSizeY[3], nWidth_B[3], nHeight_B[3];
} mvo_degrain;

int mvo_degrain_init(mvo_degrain *d, int radius, const mvo_analysis_data *ad, const mvo_super *s,
                     int64_t thsad, int64_t thsadc, int plane, int limit, int limitc, int64_t thscd1, int thscd2, char *err);
/* refs[r] = super frame planes of frame n+/-delta_r (may be NULL if out of clip); blobs[r] = MVTools_vectors
 * of vector clip r at frame n; order mvbw, mvfw, mvbw2, mvfw2 ... (MVDegrains.h:10-23) */
void mvo_degrain_frame(const mvo_degrain *d, const uint8_t *const src[3], const int srcPitch[3],
                       const uint8_t *const (*refs)[3], const int (*refPitch)[3], const uint8_t *const *blobs,
                       uint8_t *const dst[3], const int dstPitch[3]);

/* ---- mv.Compensate: MVCompensate.c ---- */
typedef struct mvo_compensate {
    mvo_analysis_data ad;
    int64_t thSAD, nSCD1; int nSCD2;
    int scBehavior, time256, fields;
    int nSuperHPad, nSuperVPad, nSuperPel, nSuperModeYUV, nSuperLevels;
    int bits, numPlanes;
} mvo_compensate;

int mvo_compensate_init(mvo_compensate *d, const mvo_analysis_data *ad, const mvo_super *s, int scbehavior,
                        int64_t thsad, double time, int64_t thscd1, int thscd2, char *err);
/* srcSuper = super frame n, refSuper = super frame nref (NULL if out of range), blob = vectors at n */
void mvo_compensate_frame(const mvo_compensate *d, const uint8_t *const srcSuper[3], const int srcPitch[3],
                          const uint8_t *const refSuper[3], const int refPitch[3], const uint8_t *blob,
                          uint8_t *const dst[3], const int dstPitch[3], int fieldShift);
/* MVAnalyse.c:135-176 / MVCompensate.c:188-225: the vertical shift between fields of opposite parity.
 * src_field / ref_field = the frames' _Field props (-1 = absent); tff = -1 when the argument was not passed.
 * Returns 0 and sets *missing when a needed _Field is absent and tff was not passed. */
int mvo_field_shift(int fields, int pel, int n, int nref, int src_field, int ref_field, int tff, int *missing);

/* ---- mv.BlockFPS: MVBlockFPS.c, MaskFun.cpp, SimpleResize.cpp (parity UNPINNED: no reference TU of it builds here) ---- */
typedef struct mvo_blockfps {
    mvo_analysis_data bw, fw;
    int mode, blend; double ml;
    int64_t thscd1; int thscd2;
    int64_t fa, fb, outFpsNum, outFpsDen;
    int inFrames, outFrames;
    int nSuperHPad, nSuperVPad, nSuperPel, nSuperModeYUV, nSuperLevels, bits;
    int nBlkXP, nBlkYP, nWidthP, nHeightP, nWidthPUV, nHeightPUV, nPitchY, nPitchUV;
} mvo_blockfps;

int mvo_blockfps_init(mvo_blockfps *d, const mvo_analysis_data *bw, const mvo_analysis_data *fw, const mvo_super *s, int numFrames, int64_t fpsNum, int64_t fpsDen,
                      int64_t num, int64_t den, int mode, double ml, int blend, int64_t thscd1, int thscd2, char *err);
void mvo_blockfps_map(const mvo_blockfps *d, int n, int *nleft, int *nright, int *time256);
int mvo_blockfps_frame(const mvo_blockfps *d, int time256, const uint8_t *const srcSuper[3], const int srcPitch[3], const uint8_t *const refSuper[3],
                       const int refPitch[3], const uint8_t *blobF, const uint8_t *blobB, const uint8_t *const clipL[3], const int clipLPitch[3],
                       const uint8_t *const clipR[3], const int clipRPitch[3], uint8_t *const dst[3], const int dstPitch[3]);
void mvo_resize_tables(int *offsets, int *weights, int out, int in);
void mvo_simple_resize_u8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int dw, int dh, int sw, int sh); /* SimpleResize.cpp:62-121 */

/* ---- kernel-level entry points (for pinning against oracle/_ref) ---- */
unsigned mvo_sad(int w, int h, int bits, const uint8_t *src, intptr_t srcPitch, const uint8_t *ref, intptr_t refPitch);
unsigned mvo_satd(int w, int h, int bits, const uint8_t *src, intptr_t srcPitch, const uint8_t *ref, intptr_t refPitch);
void mvo_overlaps(int w, int h, int bits, uint8_t *dst, intptr_t dstPitch, const uint8_t *src, intptr_t srcPitch,
                  const int16_t *win, intptr_t winPitch);
void mvo_to_pixels(int bits, uint8_t *dst, int dstPitch, const uint8_t *src, int srcPitch, int w, int h);
/* kind: 0 H-bilinear 1 V-bilinear 2 D-bilinear 3 H-bicubic 4 V-bicubic 5 H-wiener 6 V-wiener */
void mvo_refine_plane(int kind, int bits, uint8_t *dst, const uint8_t *src, intptr_t pitch, intptr_t w, intptr_t h);
void mvo_average2(int bits, uint8_t *dst, const uint8_t *a, const uint8_t *b, intptr_t pitch, intptr_t w, intptr_t h);

uint32_t mvo_fnv1a(const uint8_t *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif

/*
 * mvtools_amd.h -- C ABI of libmvtools_amd.so, the MI355X (gfx950) implementation of the mvtools hot path
 *   mv.Super -> mv.Analyse -> mv.Degrain1..6 / mv.Compensate.
 *
 * This is the drop-in boundary: a VapourSynth filter shell (or any other host) binds exactly these entry
 * points.  Plain pointers and sizes only.  Each entry point names the reference interface it replaces
 * (paths relative to dubhater/vapoursynth-mvtools src/).
 *
 * Conventions
 *   - every image pointer is a DEVICE pointer (HBM) unless the function name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued
 *     asynchronously on it, the caller synchronises;
 *   - optional filter arguments take MVX_UNSET to mean "not passed" (the reference's defaults apply);
 *   - functions return 0 on success, a negative code on failure; mvx_*_create additionally write the
 *     reference's user-visible error string (e.g. "Super: pel must be 1, 2, or 4.") into `err`.
 *   - there is NO CPU fallback: if no gfx950 device / kernel image is available the call fails.
 */
#ifndef MVTOOLS_AMD_H
#define MVTOOLS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVX_UNSET (-2147483647 - 1)
#define MVX_ERRLEN 256

#define MVX_OK 0
#define MVX_E_ARG (-1)     /* invalid filter argument (message in err) */
#define MVX_E_DEVICE (-2)  /* HIP error (message from mvx_last_error) */
#define MVX_E_NOMEM (-3)

const char *mvx_last_error(void);   /* thread-local text of the last failure */
int mvx_device_count(void);         /* number of visible gfx950 devices, <0 on error */
const char *mvx_version(void);

/* ---- wire formats (byte-identical to the reference) -------------------------------------------- */

/* VECTOR, MVAnalysisData.h:40-44 */
typedef struct mvx_vector { int32_t x, y; int64_t sad; } mvx_vector;

/* MVAnalysisData, MVAnalysisData.h:83-134 == the MVTools_MVAnalysisData frame property (84 bytes) */
typedef struct mvx_analysis_data {
    int32_t nMagicKey, nVersion, nBlkSizeX, nBlkSizeY, nPel, nLvCount, nDeltaFrame, isBackward, nCPUFlags,
        nMotionFlags, nWidth, nHeight, nOverlapX, nOverlapY, nBlkX, nBlkY, bitsPerSample, yRatioUV, xRatioUV,
        nHPadding, nVPadding;
} mvx_analysis_data;

/* ---- mv.Super ------------------------------------------------------------------------------------
 * replaces mvsuperCreate / mvsuperGetFrame, MVSuper.c:140-275 / :43-126 (argument string :279-291)   */

typedef struct mvx_super_args {
    /* the input clip's format (VSVideoInfo) */
    int32_t width, height, bits, subsampling_w, subsampling_h, gray;
    /* filter arguments, MVX_UNSET = default: hpad=16 vpad=16 pel=2 levels=0 chroma=1 sharp=2 rfilter=2 */
    int32_t hpad, vpad, pel, levels, chroma, sharp, rfilter;
} mvx_super_args;

typedef struct mvx_super_info {
    int32_t width, height, bits, xRatioUV, yRatioUV, gray;
    int32_t hpad, vpad, pel, levels, chroma, sharp, rfilter;
    int32_t modeYUV;                    /* Super_modeyuv */
    int32_t super_width, super_height;  /* luma dimensions of the super clip's frames */
    int32_t num_planes;
    int32_t plane_width[3], plane_height[3]; /* samples */
} mvx_super_info;

typedef struct mvx_super mvx_super;

int mvx_super_create(const mvx_super_args *args, mvx_super **out, char *err /* MVX_ERRLEN or NULL */);
void mvx_super_destroy(mvx_super *s);
void mvx_super_get_info(const mvx_super *s, mvx_super_info *info);

/* Builds `nframes` super frames.  src[f*3+p] / dst[f*3+p] are device pointers to plane p of frame f
 * (p >= num_planes ignored); pitches in bytes, shared by all frames.  dst planes must have been
 * zero-filled once by the caller when allocated: only the defined rectangles (every level's padded
 * plane, every sub-pel plane) are written, bytes outside them are never touched.
 * dst pitch must be a multiple of 16 bytes. */
int mvx_super_frames(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3],
                     void *const *dst, const ptrdiff_t dst_pitch[3], void *stream);

/* Optional device-side layout extension for search throughput ("shadow planes").  gfx950 serves vector loads at addresses that
 * are not multiples of four several times slower than aligned ones, and a motion search reads reference blocks at arbitrary
 * sample positions.  For clips of more than 8 bits (mvx_super_shadow_copies() == 1) a caller may therefore keep, behind the planes
 * of a super frame, two derived planes that mvx_super_shadow_frames fills from planes mvx_super_frames has written (same stream):
 *   - at luma plane + copy_stride[0]: the whole luma buffer shifted left by one sample (odd sample positions become aligned);
 *   - at U plane + copy_stride[1]: the whole U and V buffers interleaved sample by sample (2 x the chroma size; every chroma
 *     position is aligned, and a block row's U and V samples share one cache line).
 * mvx_super_shadow_bytes gives the bytes to reserve behind each plane.  The shadow planes never leave the device and are not part
 * of the super clip's frame format; a search uses them after mvx_analyse_set_ref_shadow.
 * (no reference counterpart: memory layout only, results are unchanged) */
int mvx_super_shadow_copies(const mvx_super *s);   /* 1: shadow planes pay off for this format (9..16 bits), 0: not (8 bits) */
void mvx_super_shadow_bytes(const mvx_super *s, const ptrdiff_t pitch[3], size_t extra[3]);
int mvx_super_shadow_frames(const mvx_super *s, int nframes, void *const *planes /* [f*3+p] */, const ptrdiff_t pitch[3],
                            const ptrdiff_t copy_stride[3] /* [2] unused */, void *stream);
/* mvx_super_frames followed by mvx_super_shadow_frames, as one call: same results in the planes and in the shadow planes, but for
 * pel 2 the level-0 kernels write their share of the shadow data while they have the samples in registers (about a third of a
 * Super pass's HBM traffic saved).  shadow_stride as copy_stride above. */
int mvx_super_frames_shadow(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3],
                            void *const *dst, const ptrdiff_t dst_pitch[3], const ptrdiff_t shadow_stride[3], void *stream);

/* mv.Super(pelclip=...): the sub-pel planes of level 0 are taken from the user's upsized clip instead of being interpolated.
 * replaces MVSuper.c:229-256 (mvx_super_pelclip_mode: 0 = ignored because pel is 1, 1 = pelclip is pel x the clip size,
 * 2 = pel x the padded size; other sizes -> MVX_E_ARG with the reference's message) and MVSuper.c:91-102 +
 * mvpRefineExt MVFrame.cpp:1529-1631 (mvx_super_frames_pelclip).  pelclip: [nframes*3] device planes of the clip's format,
 * rows aligned to pel samples.  mode 0 behaves like mvx_super_frames. */
int mvx_super_pelclip_mode(const mvx_super *s, int pelclip_width, int pelclip_height, int32_t *mode, char *err);
int mvx_super_frames_pelclip(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3],
                             const void *const *pelclip, const ptrdiff_t pelclip_pitch[3], int pelclip_mode,
                             void *const *dst, const ptrdiff_t dst_pitch[3], void *stream);

/* ---- mv.Finest -----------------------------------------------------------------------------------
 * replaces mvfinestGetFrame, MVFinest.c:48-140 (arg string :213-218): the pel^2 sub-pel planes of level 0 of a super frame
 * interleaved into one plane of (width + 2 hpad) * pel x (height + 2 vpad) * pel samples (chroma planes subsampled like
 * the clip).  Planes the super clip does not carry (chroma=0) are not written. */
void mvx_finest_size(const mvx_super *s, int32_t *width, int32_t *height);
int mvx_finest_frames(const mvx_super *s, int nframes, const void *const *super_frames /* [f*3+p] */, const ptrdiff_t super_pitch[3],
                      void *const *dst /* [f*3+p] */, const ptrdiff_t dst_pitch[3], void *stream);

/* ---- mv.Analyse ----------------------------------------------------------------------------------
 * replaces mvanalyseCreate / mvanalyseGetFrame, MVAnalyse.c:267-635 / :76-254 (argument string :639-671);
 * the search itself is GroupOfPlanes.c:69-125 + PlaneOfBlocks.cpp:419-1131,1447-1636.                */

typedef struct mvx_analyse_args { /* MVX_UNSET = not passed */
    int32_t blksize, blksizev, levels, search, searchparam, pelsearch, isb, lambda, chroma, delta, truemotion,
        lsad, plevel, global, pnew, pzero, pglobal, overlap, overlapv, divide, badsad, badrange, opt, meander,
        trymany, fields, tff, search_coarse, dct;
} mvx_analyse_args;

typedef struct mvx_analyse mvx_analyse;

int mvx_analyse_create(const mvx_analyse_args *args, const mvx_super *super_clip, int num_frames,
                       const ptrdiff_t super_pitch[3], mvx_analyse **out, char *err);
void mvx_analyse_destroy(mvx_analyse *a);
void mvx_analyse_get_data(const mvx_analyse *a, mvx_analysis_data *out); /* MVTools_MVAnalysisData */
int mvx_analyse_blob_size(const mvx_analyse *a);                          /* bytes of MVTools_vectors */

typedef struct mvx_analyse_job {
    const void *src[3]; /* super frame n (device) */
    const void *ref[3]; /* super frame n +/- delta (device); ref[0]==NULL -> frame too close to the clip
                           boundary: the invalid/default blob is written (GroupOfPlanes.c:150-164) */
    void *blob;         /* device, mvx_analyse_blob_size() bytes, 16-byte aligned */
    int32_t field_shift;/* MVAnalyse.c:172-176; 0 unless fields=1 */
    int32_t reserved;
} mvx_analyse_job;

/* The caller promises that the super frames of every job (src and ref) carry their shadow planes (mvx_super_shadow_frames) at
 * plane[0] + copy_stride[0] and plane[1] + copy_stride[1]; NULL or all zero: none (the default).  Only changes which addresses the
 * search loads from, never a result. */
int mvx_analyse_set_ref_shadow(mvx_analyse *a, const ptrdiff_t copy_stride[3]);

/* One chain (frame, direction) per job; all jobs run concurrently in one launch. `jobs` is a HOST array.
 * With divide > 0 the blob carries the extra array of half-size blocks and mvx_analyse_get_data reports the divided
 * geometry (MVAnalyse.c:229, :615-624), which is what readers of the vector clip must use. */
int mvx_analyse_frames(mvx_analyse *a, int njobs, const mvx_analyse_job *jobs, void *stream);

/* ---- mv.Recalculate ------------------------------------------------------------------------------
 * replaces mvrecalculateCreate / mvrecalculateGetFrame, MVRecalculate.c:263-545 / :68-254 (arg string :549-572);
 * the per-block refinement is PlaneOfBlocks.cpp:1158-1424, `divide` GroupOfPlanes.c:177-302.           */

typedef struct mvx_recalculate_args { /* MVX_UNSET = not passed */
    int64_t thsad, smooth, blksize, blksizev, search, searchparam, lambda, chroma, truemotion, pnew, overlap, overlapv,
        divide, meander, fields, dct;
} mvx_recalculate_args;

typedef struct mvx_recalculate mvx_recalculate;

/* `vectors_data`: MVTools_MVAnalysisData of the vector clip being refined */
int mvx_recalculate_create(const mvx_recalculate_args *args, const mvx_super *super_clip, const mvx_analysis_data *vectors_data,
                           const ptrdiff_t super_pitch[3], mvx_recalculate **out, char *err);
void mvx_recalculate_destroy(mvx_recalculate *r);
void mvx_recalculate_get_data(const mvx_recalculate *r, mvx_analysis_data *out); /* MVTools_MVAnalysisData of the result */
int mvx_recalculate_blob_size(const mvx_recalculate *r);

typedef struct mvx_recalculate_job {
    const void *src[3];   /* super frame n */
    const void *ref[3];   /* super frame n +/- delta; ref[0]==NULL -> the default (invalid) blob */
    const void *old_blob; /* MVTools_vectors of the old vector clip at frame n (device, 16-byte aligned) */
    void *blob;           /* out, mvx_recalculate_blob_size() bytes, 16-byte aligned */
} mvx_recalculate_job;

/* one workgroup per BLOCK: blocks of a Recalculate are independent (no spatial predictors) */
int mvx_recalculate_frames(mvx_recalculate *r, int njobs, const mvx_recalculate_job *jobs, void *stream);

/* ---- mv.Degrain1..6 ------------------------------------------------------------------------------
 * replaces mvdegrainCreate<r> / mvdegrainGetFrame<r>, MVDegrains.cpp:511-809 / :85-330 (arg strings :813-932) */

typedef struct mvx_degrain_args {
    int32_t radius;           /* 1..6 */
    int64_t thsad, thsadc;    /* MVX_UNSET -> 400 / thsad */
    int32_t plane, limit, limitc;
    int64_t thscd1; int32_t thscd2;
} mvx_degrain_args;

typedef struct mvx_degrain mvx_degrain;

int mvx_degrain_create(const mvx_degrain_args *args, const mvx_analysis_data *vectors_data /* of mvbw */,
                       const mvx_super *super_clip, const ptrdiff_t src_pitch[3], const ptrdiff_t super_pitch[3],
                       const ptrdiff_t dst_pitch[3], mvx_degrain **out, char *err);
void mvx_degrain_destroy(mvx_degrain *d);
/* The caller promises that every reference super frame of every job carries the shifted copy of its luma plane at plane[0] +
 * copy_stride[0] (mvx_super_shadow_frames; clips of more than 8 bits); NULL or zero: none (the default).  Blocks that start at an odd
 * sample are then read from the copy, at dword-aligned addresses.  Only changes which addresses are loaded, never a result.
 * (no reference counterpart: memory layout only) */
int mvx_degrain_set_ref_shadow(mvx_degrain *d, const ptrdiff_t copy_stride[3]);

typedef struct mvx_degrain_job {
    const void *src[3];          /* clip frame n */
    const void *refs[12][3];     /* super frame n+delta (mvbw), n-delta (mvfw), ... order mvbw,mvfw,mvbw2,mvfw2..;
                                    refs[r][0] may be NULL when that frame is outside the clip */
    const void *blobs[12];       /* MVTools_vectors of vector clip r at frame n (device) */
    void *dst[3];
} mvx_degrain_job;

int mvx_degrain_frames(mvx_degrain *d, int nframes, const mvx_degrain_job *jobs, void *stream);

/* ---- mv.Compensate -------------------------------------------------------------------------------
 * replaces mvcompensateCreate / mvcompensateGetFrame, MVCompensate.c:419-575 / :73-374 (arg string :579-592) */

typedef struct mvx_compensate_args {
    int32_t scbehavior;  /* MVX_UNSET -> 1 */
    int64_t thsad;       /* MVX_UNSET -> 10000 */
    double time;         /* 0..100, pass 100.0 for the default */
    int64_t thscd1; int32_t thscd2;
    int32_t fields;      /* MVX_UNSET/0 -> off; 1 needs pel > 1 (MVCompensate.c:514-517) */
} mvx_compensate_args;

typedef struct mvx_compensate mvx_compensate;

int mvx_compensate_create(const mvx_compensate_args *args, const mvx_analysis_data *vectors_data,
                          const mvx_super *super_clip, const ptrdiff_t super_pitch[3], const ptrdiff_t dst_pitch[3],
                          mvx_compensate **out, char *err);
void mvx_compensate_destroy(mvx_compensate *c);

typedef struct mvx_compensate_job {
    const void *src_super[3]; /* super frame n */
    const void *ref_super[3]; /* super frame nref; [0]==NULL if outside the clip */
    const void *blob;         /* MVTools_vectors at frame n */
    void *dst[3];
    int32_t field_shift;      /* MVCompensate.c:188-225: +-pel/2 when fields=1, pel>1, (nref-n) odd and the field parities differ; else 0 */
    int32_t reserved;
} mvx_compensate_job;

int mvx_compensate_frames(mvx_compensate *c, int nframes, const mvx_compensate_job *jobs, void *stream);

/* ---- mv.BlockFPS ---------------------------------------------------------------------------------
 * replaces mvblockfpsCreate / mvblockfpsGetFrame, MVBlockFPS.c:741-1014 / :229-676 (arg string :1017-1033);
 * masks MaskFun.cpp:63-166, mask upsizer SimpleResize.cpp:27-121.                                     */

typedef struct mvx_blockfps_args {
    int64_t num, den;        /* MVX_UNSET -> 25 / 1; 0 -> double the input rate */
    int32_t mode;            /* 0..8, MVX_UNSET -> 3 */
    double ml;               /* pass 100.0 for the default */
    int32_t blend;           /* MVX_UNSET -> 1 */
    int64_t thscd1; int32_t thscd2;
} mvx_blockfps_args;

typedef struct mvx_blockfps_info { int32_t num_frames; int64_t fps_num, fps_den; } mvx_blockfps_info; /* of the output clip */

typedef struct mvx_blockfps mvx_blockfps;

/* fps_num / fps_den: frame rate of the input clip (the reference refuses clips without one) */
int mvx_blockfps_create(const mvx_blockfps_args *args, const mvx_analysis_data *mvbw, const mvx_analysis_data *mvfw,
                        const mvx_super *super_clip, int num_frames, int64_t fps_num, int64_t fps_den,
                        const ptrdiff_t super_pitch[3], const ptrdiff_t clip_pitch[3], const ptrdiff_t dst_pitch[3],
                        mvx_blockfps **out, char *err);
void mvx_blockfps_destroy(mvx_blockfps *b);
void mvx_blockfps_get_info(const mvx_blockfps *b, mvx_blockfps_info *info);
/* output frame n -> the two input frames it lies between and its time position (MVBlockFPS.c:245-254,278-292) */
void mvx_blockfps_map(const mvx_blockfps *b, int n, int *nleft, int *nright, int *time256);

typedef struct mvx_blockfps_job {
    int32_t time256;             /* from mvx_blockfps_map; 0 / 256 copy clip_left / clip_right */
    int32_t reserved;
    const void *src_super[3];    /* super frame nleft  } all four NULL when nleft or nright lies outside the clip */
    const void *ref_super[3];    /* super frame nright }   (then, or when the vectors are unusable: blend / copy */
    const void *blob_fw;         /* mvfw vectors at nright }   of clip_left and clip_right, MVBlockFPS.c:640-673) */
    const void *blob_bw;         /* mvbw vectors at nleft  } */
    const void *clip_left[3];    /* clip frame min(nleft, last)  */
    const void *clip_right[3];   /* clip frame min(nright, last) */
    void *dst[3];
} mvx_blockfps_job;

int mvx_blockfps_frames(mvx_blockfps *b, int nframes, const mvx_blockfps_job *jobs, void *stream);

/* ---- mv.SCDetection -------------------------------------------------------------------------------
 * replaces the decision of mvscdetectionGetFrame, MVSCDetection.c:43-73 (arg string :137-145): scene_change[i] (HOST array) =
 * !usable(blobs[i]) for n device blobs of one vector clip, i.e. the value of _SceneChangePrev (forward vectors) or
 * _SceneChangeNext (backward vectors); thscd1 / thscd2 as passed by the user (MVX_UNSET -> 400 / 130). Synchronous. */
int mvx_scdetect(const mvx_analysis_data *vectors_data, int64_t thscd1, int32_t thscd2, int n, const void *const *blobs,
                 int32_t *scene_change, void *stream, char *err);

/* ---- vector blob helpers (reader side: Fakery.c, MVAnalysisData.c:7-31) -------------------------- */
void mvx_scale_thscd(int64_t *thscd1, int32_t *thscd2, const mvx_analysis_data *ad);
/* bytes of the MVTools_vectors property of a vector clip with this analysis data (Fakery.c:110-121 level geometry,
 * GroupOfPlanes.c:127-148 array layout) */
int mvx_vectors_size(const mvx_analysis_data *ad);

/* ---- test / measurement hook: selects among kernel variants that compute identical results (e.g. "general" = 1 keeps the
 * default search out of its specialised kernel so that the parity suite can run both).  The library never reads the
 * environment; a production host never needs this call. */
int mvx_debug_option(const char *name, int value);
/* the last search launch of this process: out[0] = chains per SIMD of the default-search kernel (0: the general kernel ran), out[1] = chains
 * per workgroup (team form: waves per chain), out[2] = barrier interval in blocks, out[3] = job-table entries, out[4] = which default-search kernel:
 * 0 the serial lean kernel (or the general one), 2 the speculative kernel with one wave per chain, 3 its team form (the waves of a workgroup walk one
 * chain; the library's choice for launches that leave wave slots empty).  Tests use it to assert which build a batch took. */
void mvx_debug_last_launch(int out[5]);

/* ---- small device-memory helpers so that a C host (e.g. the VapourSynth shell) needs no HIP headers */
void *mvx_dev_alloc(size_t bytes);            /* zero-filled */
void *mvx_dev_alloc_uninit(size_t bytes);     /* contents undefined (scratch, upload targets) */
void mvx_dev_free(void *p);                   /* goes to a size-keyed free list (no device synchronisation); wait for the work that uses p first */
void mvx_dev_pool_limit(size_t bytes);        /* bytes the free list may hold (default 24 GiB) */
void mvx_dev_pool_trim(void);                 /* returns the whole free list to the driver */
int mvx_dev_mem_info(size_t *free_bytes, size_t *total_bytes); /* of the current device; the free list counts as free */
void *mvx_stream_create(void);                /* a non-blocking stream for the `stream` arguments; NULL on failure */
void *mvx_stream_create_priority(int level);  /* < 0 lowest, 0 default, > 0 highest priority; different priorities never share a hardware queue */
void mvx_stream_destroy(void *stream);
int mvx_copy_to_device(void *dst, ptrdiff_t dst_pitch, const void *src_host, ptrdiff_t src_pitch, size_t row_bytes, size_t rows, void *stream);
int mvx_copy_to_host(void *dst_host, ptrdiff_t dst_pitch, const void *src, ptrdiff_t src_pitch, size_t row_bytes, size_t rows, void *stream);
int mvx_stream_sync(void *stream);
void *mvx_host_alloc_pinned(size_t bytes);    /* page-locked host memory for asynchronous mvx_copy_to_host targets; NULL on failure */
void mvx_host_free_pinned(void *p);
/* Synchronous 2-D transfers for hosts whose frames live in ordinary (pageable) memory: staged through a small set of pinned buffers
 * inside the library (a linear PCIe copy + a row-by-row memcpy on the calling thread) -- several times faster than the pageable 2-D
 * copies above for large planes, and safe to call from many threads.  Complete on return; work already enqueued on `stream` runs
 * before the copy. */
int mvx_upload_2d(void *dst, ptrdiff_t dst_pitch, const void *src_host, ptrdiff_t src_pitch, size_t row_bytes, size_t rows, void *stream);
int mvx_download_2d(void *dst_host, ptrdiff_t dst_pitch, const void *src, ptrdiff_t src_pitch, size_t row_bytes, size_t rows, void *stream);
int mvx_dev_memset(void *dst, int value, size_t bytes, void *stream); /* asynchronous on `stream` */
int mvx_set_device(int ordinal);
/* Start-up work a host can take off its first frame (r6; the VapourSynth shell calls it from a background thread when the plugin is loaded): brings the HIP runtime up on the
 * current device, loads this library's code objects (a first kernel launch pays ~0.1 s for that) and page-locks `staging_buffers` of the 16 MiB buffers mvx_upload_2d /
 * mvx_download_2d go through (`staging_buffers` < 0: the runtime only) (page-locking is slow -- ~20 ms per buffer -- and stalls every other HIP call of the process while it lasts).  Idempotent, thread-safe; MVX_OK,
 * or MVX_E_DEVICE when there is no usable device (nothing else fails because of it). */
int mvx_warmup(int staging_buffers);

#ifdef __cplusplus
}
#endif
#endif /* MVTOOLS_AMD_H */

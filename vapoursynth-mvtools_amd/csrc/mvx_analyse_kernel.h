// mvx_analyse.hip -- mv.Analyse on gfx950.
//
// The reference search (GroupOfPlanes.c:69-125, PlaneOfBlocks.cpp:419-1131) is a strictly serial chain per
// (frame, direction): with the default meander scan block i takes its predictors from block i-1 and from the row
// above, and a plane-wide running `badcount` feeds the rescue threshold.  Bit-exact vectors therefore need the
// reference scan order.  Parallelism comes from two places instead:
//   * across chains: one 64-lane wavefront (= one workgroup) owns one chain; a batch of frames x 2*tr directions
//     gives hundreds of chains resident at once (frames are independent);
//   * inside a block: every search round (predictor set / hexagon / square / exhaustive rings / UMH cross+grid) is
//     a list of candidates whose costs are independent given the running minimum, so the wave evaluates the whole
//     round at once -- G lanes per candidate split the block's rows, v_sad_u8/v_sad_u16 accumulate, a segmented
//     reduction sums them -- and an ordered arg-min (lowest candidate index wins ties) reproduces the reference's
//     strict `<` sequential update (PlaneOfBlocks.cpp:229,239,248).
// The source block is staged in LDS once per block, the previous block row's vectors live in LDS, the current
// level's hierarchical predictors are interpolated straight into the output blob (which doubles as vectors[]).
// Integer SAD reduction: no MFMA.  All double arithmetic of the reference (lambda scaling :461-462, predictor
// interpolation :1457,1500-1502) is done in IEEE fp64 with contraction off.
#include <stdarg.h>
#include <stdlib.h>

#pragma once
#include <atomic>
#include "mvx_common.h"

enum { SearchOnetime, SearchNstep, SearchLogarithmic, SearchExhaustive, SearchHex2, SearchUMH, SearchHorizontal, SearchVertical };
#define MOTION_USE_SIMD 1
#define MOTION_IS_BACKWARD 2
#define MOTION_USE_CHROMA_MOTION 8

struct ALevel {
    int nBlkX, nBlkY;
    int pel, logPel;
    // luma / chroma plane geometry at this level
    int pw, ph, hpad, vpad;
    int cpw, cph, chpad, cvpad;
    long long off[3];   // byte offset of sub-pel plane 0 inside super plane p
    long long pstride[3]; // byte distance between sub-pel planes (pitch * padded height)
    int blobOff;        // byte offset of this level's record (int size; VECTOR[]) in the blob
};

struct AParams {
    int nLevels;
    int blkX, blkY, ovX, ovY, xr, yr, logxr, logyr, bps, bits, chroma;
    int searchType, searchTypeCoarse, nSearchParam, nPelSearch, lambda, lsad, pnew, plevel, global, pglobal, pzero;
    int badrange, meander, tryMany, dctmode;
    int divide; // 0, or 1 / 2: an extra array of half-size blocks follows the finest plane (GroupOfPlanes.c:206-302)
    long long badSAD;
    long long verybigSAD;
    long long pitch[3];
    int blobSize;
    int superHPad, superVPad;
    long long shadow[3]; // byte distance between the shifted copies of a reference plane (mvx_analyse_set_ref_shadow), 0 = none
    int ablate; // developer-only (MVX_ABLATE env): 1 = skip the search, 2 = predictor round only; results are then WRONG
    ALevel lv[MVX_MAX_LEVELS];
};

struct AJob {
    const unsigned char *src[3];
    const unsigned char *ref[3];
    unsigned char *blob;
    int fieldShift, valid;
    const unsigned char *oldBlob; // mv.Recalculate only: MVTools_vectors of the clip being refined, at the same frame
};

// mv.Recalculate: geometry of the OLD vector field and the refinement threshold (MVRecalculate.c, PlaneOfBlocks.cpp:1158-1424)
struct RParams {
    int nBlkX, nBlkY, blkX, blkY, stepX, stepY, logPel, nLvCount;
    long long thSAD;
    int smooth;
};

struct mvx_analyse {
    mvx_analysis_data adOut; // what mvx_analyse_get_data reports: the divided geometry when divide > 0 (MVAnalyse.c:229, :615-624)
    mvx_analysis_data ad;
    AParams P;
    AParams *dP = nullptr;
    // The job table is the only device state a launch rewrites.  A small ring of tables (each with its own guard) lets consecutive
    // calls on one handle -- from different threads, on different streams -- overlap on the device: a launch of a few chains
    // leaves most of the chip idle and takes as long as a full one.
    struct JobSlot { AJob *d = nullptr; size_t cap = 0; CallGuard guard; };
    static constexpr int kSlots = 4;
    JobSlot slot[kSlots];
    std::atomic<unsigned> nextSlot{0};
    int ldsBytes = 0;
    int device = 0;
    CallGuard guard; // creation of dP, mvx_analyse_set_ref_shadow
};

// ------------------------------------------------------------------------------------------------ device

struct Vec { int x, y; long long sad; };

#define WAVE 64
#define BIG64 0x7fffffffffffffffLL
typedef __attribute__((address_space(3))) unsigned char lds_u8;
#define LDS_AS __attribute__((address_space(3)))
#define GL_AS __attribute__((address_space(1)))
typedef GL_AS const unsigned char gl_u8;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; } // (a no-op for the 64-thread kernels: launch bounds)

__device__ __forceinline__ int bcast_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ long long bcast_ll(long long v, int l) {
    int lo = __builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, l);
    int hi = __builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), l);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// The search state of a chain is wave-uniform by construction, but values that come back from vector memory / LDS / fp64
// VALU are "divergent" to the compiler, which then predicates every branch of the state machine on EXEC.  uni() moves such
// a value through v_readfirstlane so that it lives in SGPRs and control flow becomes scalar branches.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni(long long v) {
    int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
    int hi = __builtin_amdgcn_readfirstlane((int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ Vec uni(Vec v) { Vec r; r.x = uni(v.x); r.y = uni(v.y); r.sad = uni(v.sad); return r; }

#define DPP(v, ctrl, rmask) (unsigned)__builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), (rmask), 0xf, false)

// full-wave unsigned min via DPP (row_shr 1,2,4,8 ; row_bcast15 ; row_bcast31), result taken from lane 63
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, DPP(v, 0x111, 0xf));
    v = min(v, DPP(v, 0x112, 0xf));
    v = min(v, DPP(v, 0x114, 0xf));
    v = min(v, DPP(v, 0x118, 0xf));
    v = min(v, DPP(v, 0x142, 0xa));
    v = min(v, DPP(v, 0x143, 0xc));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// same reduction with the DPP modifier fused into v_min_u32 (the s_nop covers the VALU-write -> DPP-read hazard)
__device__ __forceinline__ unsigned wave_min_u32_fused(unsigned v) {
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// ordered arg-min of signed 32-bit costs (0x7fffffff = not a candidate): lowest lane holding the minimum, or -1
__device__ __forceinline__ int wave_argmin_i32(int cost, int *minOut) {
    const unsigned u = (unsigned)cost ^ 0x80000000u;
    const unsigned m = wave_min_u32_fused(u);
    if (m == 0xffffffffu) return -1;
    const unsigned long long mask = __ballot(u == m);
    *minOut = (int)(m ^ 0x80000000u);
    return __ffsll((long long)mask) - 1;
}
template <int LOGG> __device__ __forceinline__ unsigned group_sum_c(unsigned v) {
    if (LOGG >= 1) v += DPP(v, 0xB1, 0xf);
    if (LOGG >= 2) v += DPP(v, 0x4E, 0xf);
    if (LOGG >= 3) v += DPP(v, 0x141, 0xf);
    if (LOGG >= 4) v += DPP(v, 0x140, 0xf);
    if (LOGG >= 5) v += (unsigned)__shfl_xor((int)v, 16);
    if (LOGG >= 6) v += (unsigned)__shfl_xor((int)v, 32);
    return v;
}

__device__ __forceinline__ int wave_sum_i32(int v) {
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) { return (int)(wave_min_u32((unsigned)v ^ 0x80000000u) ^ 0x80000000u); }
__device__ __forceinline__ int wave_max_i32(int v) { return (int)((~wave_min_u32(~((unsigned)v ^ 0x80000000u))) ^ 0x80000000u); }

// sum over aligned groups of G = 1<<logG lanes; every lane of a group ends up with the group total.
// xor 1,2: quad_perm; xor 4: row_half_mirror; xor 8: row_mirror (valid once the smaller groups are uniform); 16/32: bpermute
__device__ __forceinline__ unsigned group_sum(unsigned v, int logG) {
    if (logG >= 1) v += DPP(v, 0xB1, 0xf);  // quad_perm [1,0,3,2]
    if (logG >= 2) v += DPP(v, 0x4E, 0xf);  // quad_perm [2,3,0,1]
    if (logG >= 3) v += DPP(v, 0x141, 0xf); // row_half_mirror
    if (logG >= 4) v += DPP(v, 0x140, 0xf); // row_mirror
    if (logG >= 5) v += (unsigned)__shfl_xor((int)v, 16);
    if (logG >= 6) v += (unsigned)__shfl_xor((int)v, 32);
    return v;
}

// ordered arg-min of signed 64-bit costs: lowest lane holding the minimum, or -1 if every lane has BIG64
__device__ __forceinline__ int wave_argmin_ll(long long cost, long long *minOut) {
    unsigned long long u = (unsigned long long)cost ^ 0x8000000000000000ULL; // order preserving
    unsigned hi = (unsigned)(u >> 32), lo = (unsigned)u;
    unsigned mh = wave_min_u32(hi);
    unsigned lo2 = hi == mh ? lo : 0xffffffffu;
    unsigned ml = wave_min_u32(lo2);
    unsigned long long mu = ((unsigned long long)mh << 32) | ml;
    long long mc = (long long)(mu ^ 0x8000000000000000ULL);
    if (mc == BIG64) return -1;
    unsigned long long mask = __ballot(hi == mh && lo == ml);
    *minOut = mc;
    return __ffsll((long long)mask) - 1;
}

// chunk types: global side unaligned, LDS side naturally aligned
struct __attribute__((packed, aligned(1))) U4x32 { unsigned v[4]; };
struct __attribute__((packed, aligned(1))) U2x32 { unsigned v[2]; };
struct __attribute__((packed, aligned(1))) U1x32 { unsigned v; };
struct __attribute__((packed, aligned(1))) U1x16 { unsigned short v; };
struct __attribute__((aligned(16))) A4x32 { unsigned v[4]; };
struct __attribute__((aligned(8))) A2x32 { unsigned v[2]; };
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// under-aligned vector / scalar types for the (arbitrarily aligned) reference samples in global memory
typedef unsigned uv4 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned uv2 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned uv1 __attribute__((aligned(1)));
typedef unsigned short uh1 __attribute__((aligned(1)));
// VECTOR as it sits in the blob: records start at byte 12 of a 16-byte aligned blob -> only 4-byte aligned
struct __attribute__((packed, aligned(4))) GVec { int x, y; long long sad; };

template <int BPS> __device__ __forceinline__ unsigned sad32(unsigned a, unsigned b, unsigned acc) {
    return BPS == 1 ? __builtin_amdgcn_sad_u8(a, b, acc) : __builtin_amdgcn_sad_u16(a, b, acc);
}

// load CB bytes (CB in {2,4,8,16}) from unaligned global memory into 4 dwords
__device__ __forceinline__ A4x32 ld_chunk_g(gl_u8 *g, int CB) {
    A4x32 a; a.v[0] = a.v[1] = a.v[2] = a.v[3] = 0;
    if (CB == 16) { uv4 t = *(GL_AS const uv4 *)g; a.v[0] = t[0]; a.v[1] = t[1]; a.v[2] = t[2]; a.v[3] = t[3]; }
    else if (CB == 8) { uv2 t = *(GL_AS const uv2 *)g; a.v[0] = t[0]; a.v[1] = t[1]; }
    else if (CB == 4) a.v[0] = *(GL_AS const uv1 *)g;
    else a.v[0] = *(GL_AS const uh1 *)g;
    return a;
}
__device__ __forceinline__ void st_chunk_l(lds_u8 *d, const A4x32 &a, int CB) {
    if (CB == 16) { v4u t = { a.v[0], a.v[1], a.v[2], a.v[3] }; *(LDS_AS v4u *)d = t; }
    else if (CB == 8) { v2u t = { a.v[0], a.v[1] }; *(LDS_AS v2u *)d = t; }
    else if (CB == 4) *(LDS_AS unsigned *)d = a.v[0];
    else *(LDS_AS unsigned short *)d = (unsigned short)a.v[0];
}
// SAD of one chunk: src from LDS, ref from global
template <int BPS> __device__ __forceinline__ unsigned sad_chunk(const lds_u8 *s, gl_u8 *r, int CB, unsigned acc) {
    if (CB == 16) {
        v4u a = *(const LDS_AS v4u *)s; uv4 b = *(GL_AS const uv4 *)r;
        acc = sad32<BPS>(a[0], b[0], acc); acc = sad32<BPS>(a[1], b[1], acc);
        acc = sad32<BPS>(a[2], b[2], acc); acc = sad32<BPS>(a[3], b[3], acc);
    } else if (CB == 8) {
        v2u a = *(const LDS_AS v2u *)s; uv2 b = *(GL_AS const uv2 *)r;
        acc = sad32<BPS>(a[0], b[0], acc); acc = sad32<BPS>(a[1], b[1], acc);
    } else if (CB == 4) {
        unsigned a = *(const LDS_AS unsigned *)s; unsigned b = *(GL_AS const uv1 *)r;
        acc = sad32<BPS>(a, b, acc);
    } else {
        unsigned short a = *(const LDS_AS unsigned short *)s; unsigned short b = *(GL_AS const uh1 *)r;
        acc = sad32<BPS>(a, b, acc);
    }
    return acc;
}

// small signed tables packed into 64-bit immediates (no memory traffic): entry i = (int8)(w >> 8i)
__device__ __forceinline__ int tab8(unsigned long long w, int i) { return (int)(signed char)(w >> (8 * i)); }
#define PACK8(a, b, c, d, e, f, g, h) ((unsigned long long)(unsigned char)(a) | ((unsigned long long)(unsigned char)(b) << 8) | ((unsigned long long)(unsigned char)(c) << 16) | \
    ((unsigned long long)(unsigned char)(d) << 24) | ((unsigned long long)(unsigned char)(e) << 32) | ((unsigned long long)(unsigned char)(f) << 40) | \
    ((unsigned long long)(unsigned char)(g) << 48) | ((unsigned long long)(unsigned char)(h) << 56))
// hex2[8][2], PlaneOfBlocks.cpp:664
#define HEX2X PACK8(-1, -2, -1, 1, 2, 1, -1, -2)
#define HEX2Y PACK8(-2, 0, 2, 2, 0, -2, -2, 0)
// hex4[16][2], PlaneOfBlocks.cpp:755-757
#define HEX4XA PACK8(-4, -4, -4, -4, -4, 4, 4, 4)
#define HEX4XB PACK8(4, 4, 2, 0, -2, -2, 0, 2)
#define HEX4YA PACK8(2, 1, 0, -1, -2, -2, -1, 0)
#define HEX4YB PACK8(1, 2, 3, 4, 3, -3, -4, -3)
// NStep order, PlaneOfBlocks.cpp:474-481
#define NSTEPX PACK8(1, 1, 1, 0, 0, -1, -1, -1)
#define NSTEPY PACK8(1, 0, -1, -1, 1, 1, 0, -1)

enum { G_SINGLE, G_RING, G_RINGS, G_HEX6, G_HEX3, G_NSTEP, G_CROSS, G_HEX4, G_LINEH, G_LINEV, G_LIST, G_ROUNDA };
// a, b: parameters of the pattern.  G_LIST: up to 4 unit offsets packed one per byte in lx / ly (scaled by a)
struct CandGen { int kind, cx, cy, a, b; unsigned lx, ly; };

// candidate c of a round, in the reference's evaluation order
__device__ __forceinline__ void gen_cand(const CandGen &G, int c, int &vx, int &vy) {
    int dx = 0, dy = 0;
    switch (G.kind) {
    case G_SINGLE: break;
    case G_RINGS: { // rings 1..a, step 1 (Exhaustive, PlaneOfBlocks.cpp:786-791)
        int r = 1;
        while (r < G.a && c >= 8 * r) { c -= 8 * r; r++; }
        const int n = 2 * r - 1;
        if (c < 2 * n) { dx = -r + 1 + (c >> 1); dy = (c & 1) ? r : -r; }
        else if ((c -= 2 * n) < 2 * n) { dy = -r + 1 + (c >> 1); dx = (c & 1) ? r : -r; }
        else { c -= 2 * n; dx = (c & 2) ? r : -r; dy = (c & 1) ? r : -r; }
        break;
    }
    case G_RING: { // one ring radius a, step b (ExpandingSearch :636-658)
        const int r = G.a, s = G.b;
        int n = 0;
        for (int i = -r + s; i < r; i += s) n++;
        if (c < 2 * n) { dx = -r + s + (c >> 1) * s; dy = (c & 1) ? r : -r; }
        else if ((c -= 2 * n) < 2 * n) { dy = -r + s + (c >> 1) * s; dx = (c & 1) ? r : -r; }
        else { c -= 2 * n; dx = (c & 2) ? r : -r; dy = (c & 1) ? r : -r; }
        break;
    }
    case G_HEX6: { int i = (c & 7) + 1; i = i > 7 ? 7 : i; dx = tab8(HEX2X, i); dy = tab8(HEX2Y, i); break; } // hex2[c+1], :682-687
    case G_HEX3: { int i = G.a + (c & 3); i = i > 7 ? 7 : i; dx = tab8(HEX2X, i); dy = tab8(HEX2Y, i); break; } // hex2[odir+c], :706-708
    case G_NSTEP: dx = tab8(NSTEPX, c & 7) * G.a; dy = tab8(NSTEPY, c & 7) * G.a; break;
    case G_CROSS: { // a = number of odd offsets (CrossSearch :728-739)
        if (c < 2 * G.a) { int i = 1 + 2 * (c >> 1); dx = (c & 1) ? i : -i; }
        else { int k = c - 2 * G.a; int j = 1 + 2 * (k >> 1); dy = (k & 1) ? j : -j; }
        break;
    }
    case G_HEX4: { int i = 1 + (c >> 4), j = c & 15; // :753-764
        dx = (j < 8 ? tab8(HEX4XA, j) : tab8(HEX4XB, j - 8)) * i; dy = (j < 8 ? tab8(HEX4YA, j) : tab8(HEX4YB, j - 8)) * i; break; }
    case G_LINEH: { int i = 1 + (c >> 1); dx = (c & 1) ? i : -i; break; } // :799-806
    case G_LINEV: { int i = 1 + (c >> 1); dy = (c & 1) ? i : -i; break; } // :808-815
    case G_LIST: dx = (int)(signed char)(G.lx >> (8 * (c & 3))) * G.a; dy = (int)(signed char)(G.ly >> (8 * (c & 3))) * G.a; break;
    default: break;
    }
    vx = G.cx + dx; vy = G.cy + dy;
}
#define PACK4(a, b, c, d) ((unsigned)(unsigned char)(a) | ((unsigned)(unsigned char)(b) << 8) | ((unsigned)(unsigned char)(c) << 16) | ((unsigned)(unsigned char)(d) << 24))


#ifdef MVX_PROFILE
__device__ unsigned long long g_prof[32];
#define PROF_T() ({ asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); (long long)__builtin_amdgcn_s_memtime(); })
#define PROF_ADD(i, v) (prof[i] += (v))
#else
#define PROF_T() 0LL
#define PROF_ADD(i, v) ((void)0)
#endif
#define PF_MAX 4 // source-block prefetch registers per lane (16 B each)
// Compile-time block geometry for the specialised kernels (BW == 0: geometry only known at run time -> generic loops).
constexpr int pow2c(int v, int r = 1) { return r >= v ? r : pow2c(v, r * 2); }
// DCT: the SATD cost modes (dct 5..10) are compiled into their own generic kernel only -- their code costs the default
// kernels 5-7 % through register allocation even when it never runs (measured)
template <int BW_, int BH_, int XR_, int YR_, bool DCT_ = false> struct Geo {
    static constexpr int BW = BW_, BH = BH_, XR = XR_, YR = YR_; static constexpr bool DCT = DCT_;
};
typedef Geo<0, 0, 0, 0> GeoAny;
typedef Geo<0, 0, 0, 0, true> GeoAnyDct;

// WPE: chains per SIMD the enclosing kernel is built for (register budget; picks register-saving variants below)
template <int BPS, typename GEO, int WPE = 1> struct Searcher {
    const AParams &P;
    const AJob &J;
    lds_u8 *lds;          // [srcblock | rowbuf | hist]
    int ldsRow, ldsHist, histBins;

    // level constants
    int level, nBlkX, nBlkY, pel, logPel;
    gl_u8 *srcY, *srcU, *srcV, *refY, *refU, *refV; // level bases (sub-pel plane 0); U and V share pitch and geometry
    long long pitchY, pitchC, pstrideY, pstrideC;
    int pw, ph, hpad, vpad, chpad, cvpad;
    int lumaRowB, chromaRowB, CBL, CBC, logCL, logCC, TL, TCp, TT, uoff, voff;
    int cBlkX, cBlkY;
    GL_AS GVec *vectors; // blob record of this level (after the int header) == the reference's vectors[]

    // hot filter parameters copied out of the parameter block once per level (kept in registers)
    int chroma, logxr, logyr, blkW, blkH, meander;
    long long verybig;

    // plane-scan state (uniform)
    int searchType, nSearchParam;
    int dctmode, dctweight16, srcLuma, sumLumaChange; // SATD cost modes (dct 5..10), PlaneOfBlocks.cpp:117-203
    long long nLambda, LSAD;
    int penaltyNew, penaltyZero, pglobal, badrange, badcount, tryMany;
    long long badSAD;
    Vec globalMVPredictor, zeroMVfieldShifted;
    int smallestPlane;

    // block state (uniform)
    int x0, y0, cx0, cy0, blkx, blky, blkIdx, blkScanDir;
    int nDxMin, nDyMin, nDxMax, nDyMax;
    Vec bestMV, predictor, predictors[4];
    long long nMinCost;

    __device__ Searcher(const AParams &p, const AJob &j) : P(p), J(j) {}

    __device__ __forceinline__ bool vector_ok(int vx, int vy) const { return vx >= nDxMin && vy >= nDyMin && vx < nDxMax && vy < nDyMax; }
    __device__ __forceinline__ Vec clip_mv(Vec v) const {
        Vec r;
        r.x = min(max(v.x, nDxMin), nDxMax - 1);
        r.y = min(max(v.y, nDyMin), nDyMax - 1);
        r.sad = v.sad;
        return r;
    }
    // PlaneOfBlocks.cpp:105-114
    __device__ __forceinline__ int motion_distortion(int vx, int vy) const {
        unsigned dx = (unsigned)(predictor.x - vx), dy = (unsigned)(predictor.y - vy);
        int dist = (int)(dx * dx + dy * dy);
        return (int)((nLambda * dist) >> 8);
    }

    // reference-block base pointers for a candidate (PlaneOfBlocks.cpp:35-101, MVFrame.cpp:1707-1729)
    __device__ __forceinline__ gl_u8 *ref_luma(int vx, int vy) const {
        int ax = (x0 << logPel) + vx, ay = (y0 << logPel) + vy;
        int m = pel - 1;
        int idx = (ax & m) | ((ay & m) << logPel);
        return refY + idx * pstrideY + (long long)(ay >> logPel) * pitchY + (long long)(ax >> logPel) * BPS;
    }
    // byte offset of the chroma reference block inside the U (== V) level plane
    __device__ __forceinline__ long long ref_chroma_off(int vx, int vy) const {
        int xbias = (vx < 0) ? ((1 << logxr) - 1) : 0;
        int ybias = (vy < 0) ? ((1 << logyr) - 1) : 0;
        int ax = (cx0 << logPel) + ((vx + xbias) >> logxr), ay = (cy0 << logPel) + ((vy + ybias) >> logyr);
        int m = pel - 1;
        int idx = (ax & m) | ((ay & m) << logPel);
        return idx * pstrideC + (long long)(ay >> logPel) * pitchC + (long long)(ax >> logPel) * BPS;
    }

    // 32-bit forms for the specialised kernels (the host checks that a level's plane set is smaller than 2 GiB): the loads then
    // address "uniform 64-bit base + per-lane 32-bit offset" directly, which saves ~40 vector instructions of 64-bit address
    // arithmetic per search round.  Offsets of inadmissible candidates may wrap; their lanes never load.
    __device__ __forceinline__ unsigned ref_luma_off32(int vx, int vy) const {
        const int ax = (x0 << logPel) + vx, ay = (y0 << logPel) + vy, m = pel - 1;
        const unsigned idx = (unsigned)((ax & m) | ((ay & m) << logPel));
        return idx * (unsigned)pstrideY + (unsigned)(ay >> logPel) * (unsigned)pitchY + (unsigned)(ax >> logPel) * BPS;
    }
    __device__ __forceinline__ unsigned ref_chroma_off32(int vx, int vy) const {
        const int lxr = GEO::BW ? (GEO::XR == 2 ? 1 : 0) : logxr, lyr = GEO::BW ? (GEO::YR == 2 ? 1 : 0) : logyr; // compile-time in the specialised kernels
        const int xbias = (vx < 0) ? ((1 << lxr) - 1) : 0, ybias = (vy < 0) ? ((1 << lyr) - 1) : 0;
        const int ax = (cx0 << logPel) + ((vx + xbias) >> lxr), ay = (cy0 << logPel) + ((vy + ybias) >> lyr), m = pel - 1;
        const unsigned idx = (unsigned)((ax & m) | ((ay & m) << logPel));
        return idx * (unsigned)pstrideC + (unsigned)(ay >> logPel) * (unsigned)pitchC + (unsigned)(ax >> logPel) * BPS;
    }

    // item t of the block -> LDS offset and (plane, row, byte offset in row)
    __device__ __forceinline__ void item(int t, int &pl, int &row, int &xb, int &loff, int &cb) const {
        if (t < TL) { pl = 0; row = t >> logCL; xb = (t & ((1 << logCL) - 1)) * CBL; loff = row * lumaRowB + xb; cb = CBL; }
        else {
            int tt = t - TL;
            pl = tt >= TCp ? 2 : 1;
            if (pl == 2) tt -= TCp;
            row = tt >> logCC; xb = (tt & ((1 << logCC) - 1)) * CBC;
            loff = (pl == 2 ? voff : uoff) + row * chromaRowB + xb; cb = CBC;
        }
    }

    // SAD of this lane's items of one plane region: rows of `rowB` bytes split in CB-byte chunks; items t = s, s+G, ...
    // All loads of a batch (<= EV_BATCH per lane) are issued before the first SAD so that they overlap: the wave pays
    // one memory latency per batch instead of one per chunk.
#define EV_BATCH 8
#ifdef MVX_NT_REF
#define LDREF(p) __builtin_nontemporal_load(p)
#else
#define LDREF(p) (*(p))
#endif
    template <int CB> __device__ __forceinline__ unsigned eval_region(int s, int logG, int T, int logC, int rowB, const lds_u8 *src,
                                                                       gl_u8 *ref, long long refPitch, unsigned acc) const {
        // T and G are powers of two: every lane owns exactly cnt = T/G items (or lanes s < T one item each when G > T),
        // so the trip count is uniform and the loop control is scalar.
        const int cnt = T >> logG;
        if (cnt == 0) {
            if (s < T) {
                const int row = s >> logC, xb = (s & ((1 << logC) - 1)) * CB;
                acc = sad_chunk<BPS>(src + row * rowB + xb, ref + (long long)row * refPitch + xb, CB, acc);
            }
            return acc;
        }
        for (int k0 = 0; k0 < cnt; k0 += EV_BATCH) {
            v4u r[EV_BATCH];
#pragma unroll
            for (int k = 0; k < EV_BATCH; k++) {
                r[k] = v4u{0, 0, 0, 0};
                if (k0 + k < cnt) {
                    const int t = s + ((k0 + k) << logG);
                    const int row = t >> logC, xb = (t & ((1 << logC) - 1)) * CB;
                    gl_u8 *p = ref + (long long)row * refPitch + xb;
                    if (CB == 16) { uv4 v = *(GL_AS const uv4 *)p; r[k] = v4u{v[0], v[1], v[2], v[3]}; }
                    else if (CB == 8) { uv2 v = *(GL_AS const uv2 *)p; r[k][0] = v[0]; r[k][1] = v[1]; }
                    else if (CB == 4) r[k][0] = *(GL_AS const uv1 *)p;
                    else r[k][0] = *(GL_AS const uh1 *)p;
                }
            }
#pragma unroll
            for (int k = 0; k < EV_BATCH; k++) {
                if (k0 + k < cnt) {
                    const int t = s + ((k0 + k) << logG);
                    const int row = t >> logC, xb = (t & ((1 << logC) - 1)) * CB;
                    const lds_u8 *sp = src + row * rowB + xb;
                    if (CB == 16) {
                        v4u a = *(const LDS_AS v4u *)sp;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                        acc = sad32<BPS>(a[2], r[k][2], acc); acc = sad32<BPS>(a[3], r[k][3], acc);
                    } else if (CB == 8) {
                        v2u a = *(const LDS_AS v2u *)sp;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                    } else if (CB == 4) acc = sad32<BPS>(*(const LDS_AS unsigned *)sp, r[k][0], acc);
                    else acc = sad32<BPS>(*(const LDS_AS unsigned short *)sp, r[k][0], acc);
                }
            }
        }
        return acc;
    }
    __device__ __forceinline__ unsigned eval_region_cb(int CB, int s, int logG, int T, int logC, int rowB, const lds_u8 *src, gl_u8 *ref,
                                                       long long refPitch, unsigned acc) const {
        switch (CB) {
        case 16: return eval_region<16>(s, logG, T, logC, rowB, src, ref, refPitch, acc);
        case 8: return eval_region<8>(s, logG, T, logC, rowB, src, ref, refPitch, acc);
        case 4: return eval_region<4>(s, logG, T, logC, rowB, src, ref, refPitch, acc);
        default: return eval_region<2>(s, logG, T, logC, rowB, src, ref, refPitch, acc);
        }
    }

    // ---- specialised evaluation: block geometry and lanes-per-candidate known at compile time -> straight-line code,
    // all loads of the lane issued back to back (one memory latency per pass), no per-item predicates.
    template <int LOGG, int T, int LOGC, int CB, int ROWB>
    __device__ __forceinline__ unsigned region_fixed(int s, const lds_u8 *src, gl_u8 *ref, long long refPitch, unsigned acc) const {
        constexpr int G = 1 << LOGG, C = 1 << LOGC;
        if (T < G) { // fewer items than lanes in the group: lanes s < T own one item each
            if (s < T) {
                const int row = s >> LOGC, xb = (s & (C - 1)) * CB;
                acc = sad_chunk<BPS>(src + row * ROWB + xb, ref + (long long)row * refPitch + xb, CB, acc);
            }
            return acc;
        }
        constexpr int N = T >= G ? T / G : 1;        // items per lane
#ifndef MVX_NB
#define MVX_NB 4 // chunks in flight per region: four measured +2 % over eight at 4K16 (register pressure), and keeps the 8x8 / 16x16 kernels spill-free at 256 registers (two would do that for the 16-bit 32x32 kernel too, but costs it 17 %: 8K 48.2 -> 40.1 fps)
#endif
        constexpr int NB = N < MVX_NB ? N : MVX_NB;  // loads in flight per batch
        if (G >= C) { // the chunk column is fixed per lane, rows advance by G / C per item
            const int row0 = s >> LOGC, xb = (s & (C - 1)) * CB;
            gl_u8 *p = ref + (long long)row0 * refPitch + xb;
            const lds_u8 *sp = src + row0 * ROWB + xb;
            const long long step = (long long)(G >> LOGC) * refPitch;
            constexpr int lstep = (G >> LOGC) * ROWB;
#pragma unroll 2
            for (int k0 = 0; k0 < N; k0 += NB) {
                v4u r[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    gl_u8 *q = p + (k0 + k) * step;
                    if (CB == 16) { uv4 v = LDREF((GL_AS const uv4 *)q); r[k] = v4u{v[0], v[1], v[2], v[3]}; }
                    else if (CB == 8) { uv2 v = LDREF((GL_AS const uv2 *)q); r[k] = v4u{v[0], v[1], 0, 0}; }
                    else if (CB == 4) r[k] = v4u{LDREF((GL_AS const uv1 *)q), 0, 0, 0};
                    else r[k] = v4u{LDREF((GL_AS const uh1 *)q), 0, 0, 0};
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const lds_u8 *l = sp + (k0 + k) * lstep;
                    if (CB == 16) {
                        v4u a = *(const LDS_AS v4u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                        acc = sad32<BPS>(a[2], r[k][2], acc); acc = sad32<BPS>(a[3], r[k][3], acc);
                    } else if (CB == 8) {
                        v2u a = *(const LDS_AS v2u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                    } else if (CB == 4) acc = sad32<BPS>(*(const LDS_AS unsigned *)l, r[k][0], acc);
                    else acc = sad32<BPS>(*(const LDS_AS unsigned short *)l, r[k][0], acc);
                }
            }
        } else { // several lanes' worth of chunks per row: general item -> (row, chunk) mapping, still compile-time counts
#pragma unroll 2
            for (int k0 = 0; k0 < N; k0 += NB) {
                v4u r[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int t = s + (k0 + k) * G, row = t >> LOGC, xb = (t & (C - 1)) * CB;
                    gl_u8 *q = ref + (long long)row * refPitch + xb;
                    if (CB == 16) { uv4 v = LDREF((GL_AS const uv4 *)q); r[k] = v4u{v[0], v[1], v[2], v[3]}; }
                    else if (CB == 8) { uv2 v = LDREF((GL_AS const uv2 *)q); r[k] = v4u{v[0], v[1], 0, 0}; }
                    else if (CB == 4) r[k] = v4u{LDREF((GL_AS const uv1 *)q), 0, 0, 0};
                    else r[k] = v4u{LDREF((GL_AS const uh1 *)q), 0, 0, 0};
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int t = s + (k0 + k) * G, row = t >> LOGC, xb = (t & (C - 1)) * CB;
                    const lds_u8 *l = src + row * ROWB + xb;
                    if (CB == 16) {
                        v4u a = *(const LDS_AS v4u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                        acc = sad32<BPS>(a[2], r[k][2], acc); acc = sad32<BPS>(a[3], r[k][3], acc);
                    } else if (CB == 8) {
                        v2u a = *(const LDS_AS v2u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                    } else if (CB == 4) acc = sad32<BPS>(*(const LDS_AS unsigned *)l, r[k][0], acc);
                    else acc = sad32<BPS>(*(const LDS_AS unsigned short *)l, r[k][0], acc);
                }
            }
        }
        return acc;
    }

    template <int LOGG, int T, int LOGC, int CB, int ROWB>
    __device__ __forceinline__ unsigned region_fixed32(int s, const lds_u8 *src, gl_u8 *base, unsigned off, unsigned refPitch, unsigned acc) const {
        constexpr int G = 1 << LOGG, C = 1 << LOGC;
        if (T < G) { // fewer items than lanes in the group: lanes s < T own one item each
            if (s < T) {
                const int row = s >> LOGC, xb = (s & (C - 1)) * CB;
                acc = sad_chunk<BPS>(src + row * ROWB + xb, base + (off + (unsigned)row * refPitch + (unsigned)xb), CB, acc);
            }
            return acc;
        }
        constexpr int N = T >= G ? T / G : 1;        // items per lane
#ifndef MVX_NB
#define MVX_NB 4 // chunks in flight per region: four measured +2 % over eight at 4K16 (register pressure), and keeps the 8x8 / 16x16 kernels spill-free at 256 registers (two would do that for the 16-bit 32x32 kernel too, but costs it 17 %: 8K 48.2 -> 40.1 fps)
#endif
        constexpr int NB = N < MVX_NB ? N : MVX_NB;  // loads in flight per batch
        if (G >= C) { // the chunk column is fixed per lane, rows advance by G / C per item
            const int row0 = s >> LOGC, xb = (s & (C - 1)) * CB;
            const unsigned p = off + (unsigned)row0 * refPitch + (unsigned)xb;
            const lds_u8 *sp = src + row0 * ROWB + xb;
            const unsigned step = (unsigned)(G >> LOGC) * refPitch;
            constexpr int lstep = (G >> LOGC) * ROWB;
            unsigned po = p; // running offset: one add per chunk (kept as a chain: as p + k * step the compiler multiplies per chunk)
#pragma unroll 2
            for (int k0 = 0; k0 < N; k0 += NB) {
                v4u r[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    gl_u8 *q = base + po;
                    po += step;
                    asm("" : "+v"(po)); // (not volatile: must not become a scheduling barrier)
                    if (CB == 16) { uv4 v = LDREF((GL_AS const uv4 *)q); r[k] = v4u{v[0], v[1], v[2], v[3]}; }
                    else if (CB == 8) { uv2 v = LDREF((GL_AS const uv2 *)q); r[k] = v4u{v[0], v[1], 0, 0}; }
                    else if (CB == 4) r[k] = v4u{LDREF((GL_AS const uv1 *)q), 0, 0, 0};
                    else r[k] = v4u{LDREF((GL_AS const uh1 *)q), 0, 0, 0};
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const lds_u8 *l = sp + (k0 + k) * lstep;
                    if (CB == 16) {
                        v4u a = *(const LDS_AS v4u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                        acc = sad32<BPS>(a[2], r[k][2], acc); acc = sad32<BPS>(a[3], r[k][3], acc);
                    } else if (CB == 8) {
                        v2u a = *(const LDS_AS v2u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                    } else if (CB == 4) acc = sad32<BPS>(*(const LDS_AS unsigned *)l, r[k][0], acc);
                    else acc = sad32<BPS>(*(const LDS_AS unsigned short *)l, r[k][0], acc);
                }
            }
        } else { // several lanes' worth of chunks per row: general item -> (row, chunk) mapping, still compile-time counts
#pragma unroll 2
            for (int k0 = 0; k0 < N; k0 += NB) {
                v4u r[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int t = s + (k0 + k) * G, row = t >> LOGC, xb = (t & (C - 1)) * CB;
                    gl_u8 *q = base + (off + (unsigned)row * refPitch + (unsigned)xb);
                    if (CB == 16) { uv4 v = LDREF((GL_AS const uv4 *)q); r[k] = v4u{v[0], v[1], v[2], v[3]}; }
                    else if (CB == 8) { uv2 v = LDREF((GL_AS const uv2 *)q); r[k] = v4u{v[0], v[1], 0, 0}; }
                    else if (CB == 4) r[k] = v4u{LDREF((GL_AS const uv1 *)q), 0, 0, 0};
                    else r[k] = v4u{LDREF((GL_AS const uh1 *)q), 0, 0, 0};
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int t = s + (k0 + k) * G, row = t >> LOGC, xb = (t & (C - 1)) * CB;
                    const lds_u8 *l = src + row * ROWB + xb;
                    if (CB == 16) {
                        v4u a = *(const LDS_AS v4u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                        acc = sad32<BPS>(a[2], r[k][2], acc); acc = sad32<BPS>(a[3], r[k][3], acc);
                    } else if (CB == 8) {
                        v2u a = *(const LDS_AS v2u *)l;
                        acc = sad32<BPS>(a[0], r[k][0], acc); acc = sad32<BPS>(a[1], r[k][1], acc);
                    } else if (CB == 4) acc = sad32<BPS>(*(const LDS_AS unsigned *)l, r[k][0], acc);
                    else acc = sad32<BPS>(*(const LDS_AS unsigned short *)l, r[k][0], acc);
                }
            }
        }
        return acc;
    }

    static constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v / 2); }

    template <int LOGG> __device__ __forceinline__ void eval_fixed(int s, int vx, int vy, int vyc, unsigned &aL, unsigned &aC) const {
        constexpr int BW = GEO::BW ? GEO::BW : 8, BH = GEO::BH ? GEO::BH : 8, XR = GEO::XR ? GEO::XR : 1, YR = GEO::YR ? GEO::YR : 1;
        constexpr int LROWB = BW * BPS, LCB = LROWB < 16 ? LROWB : 16, LLOGC = ilog2c(LROWB / LCB), LT = BH * (LROWB / LCB);
        constexpr int CROWB = (BW / XR) * BPS, CCB = CROWB < 16 ? CROWB : 16, CLOGC = ilog2c(CROWB / CCB), CT = (BH / YR) * (CROWB / CCB);
        constexpr int UOFF = BH * LROWB, VOFF = UOFF + (BH / YR) * CROWB;
        aL = region_fixed32<LOGG, LT, LLOGC, LCB, LROWB>(s, lds, refY, ref_luma_off32(vx, vy), (unsigned)pitchY, aL);
        if (chroma) {
            const unsigned co = ref_chroma_off32(vx, vyc);
            aC = region_fixed32<LOGG, CT, CLOGC, CCB, CROWB>(s, lds + UOFF, refU, co, (unsigned)pitchC, aC);
            aC = region_fixed32<LOGG, CT, CLOGC, CCB, CROWB>(s, lds + VOFF, refV, co, (unsigned)pitchC, aC);
        }
    }

    static constexpr int G_BW = GEO::BW ? GEO::BW : 8, G_BH = GEO::BH ? GEO::BH : 8, G_XR = GEO::XR ? GEO::XR : 1, G_YR = GEO::YR ? GEO::YR : 1;
    static constexpr int G_LROWB = G_BW * BPS, G_LCB = G_LROWB < 16 ? G_LROWB : 16, G_LC = G_LROWB / G_LCB, G_LT = G_BH * G_LC;
    static constexpr int G_CROWB = (G_BW / G_XR) * BPS, G_CCB = G_CROWB < 16 ? G_CROWB : 16, G_CC = G_CROWB / G_CCB, G_CT = (G_BH / G_YR) * G_CC;
    static constexpr int G_UOFF = G_BH * G_LROWB, G_VOFF = G_UOFF + (G_BH / G_YR) * G_CROWB;
    // ---- two-phase form of eval_fixed<3> (eight lanes per candidate): the reference chunks are requested into registers
    // early and compared against the source block later, so that the memory latency of the predictor round overlaps the
    // rest of the block prologue.
    static constexpr int P_LC = ilog2c(G_LC), P_CC = ilog2c(G_CC);
    static constexpr int P_NL = G_LT >= 8 ? G_LT / 8 : 1, P_NC = G_CT >= 8 ? G_CT / 8 : 1;
    struct PreA { int vx, vy, vyc; bool ok; v4u l[P_NL], u[P_NC], v[P_NC]; };
    template <int CB> __device__ __forceinline__ static v4u ld_ref(gl_u8 *q) {
        if (CB == 16) { uv4 t = LDREF((GL_AS const uv4 *)q); return v4u{t[0], t[1], t[2], t[3]}; }
        else if (CB == 8) { uv2 t = LDREF((GL_AS const uv2 *)q); return v4u{t[0], t[1], 0, 0}; }
        else if (CB == 4) return v4u{LDREF((GL_AS const uv1 *)q), 0, 0, 0};
        else return v4u{LDREF((GL_AS const uh1 *)q), 0, 0, 0};
    }
    template <int CB> __device__ __forceinline__ static unsigned sad_ref(const lds_u8 *l, const v4u &r, unsigned acc) {
        if (CB == 16) {
            v4u a = *(const LDS_AS v4u *)l;
            acc = sad32<BPS>(a[0], r[0], acc); acc = sad32<BPS>(a[1], r[1], acc);
            acc = sad32<BPS>(a[2], r[2], acc); acc = sad32<BPS>(a[3], r[3], acc);
        } else if (CB == 8) {
            v2u a = *(const LDS_AS v2u *)l;
            acc = sad32<BPS>(a[0], r[0], acc); acc = sad32<BPS>(a[1], r[1], acc);
        } else if (CB == 4) acc = sad32<BPS>(*(const LDS_AS unsigned *)l, r[0], acc);
        else acc = sad32<BPS>(*(const LDS_AS unsigned short *)l, r[0], acc);
        return acc;
    }
    template <int T, int LOGC, int CB, int N> __device__ __forceinline__ void region_load8(int s, gl_u8 *ref, long long refPitch, v4u *r) const {
        constexpr int C = 1 << LOGC;
        static_assert(C <= 8, "chunks per row");
        const int row0 = s >> LOGC, xb = (s & (C - 1)) * CB;
        if (T < 8) { if (s < T) r[0] = ld_ref<CB>(ref + (long long)row0 * refPitch + xb); return; }
        gl_u8 *p = ref + (long long)row0 * refPitch + xb;
        const long long step = (long long)(8 >> LOGC) * refPitch;
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = ld_ref<CB>(p + k * step);
    }
    template <int T, int LOGC, int CB, int ROWB, int N> __device__ __forceinline__ unsigned region_sad8(int s, const lds_u8 *src, const v4u *r, unsigned acc) const {
        constexpr int C = 1 << LOGC;
        const int row0 = s >> LOGC, xb = (s & (C - 1)) * CB;
        const lds_u8 *sp = src + row0 * ROWB + xb;
        if (T < 8) { if (s < T) acc = sad_ref<CB>(sp, r[0], acc); return acc; }
        constexpr int lstep = (8 >> LOGC) * ROWB;
#pragma unroll
        for (int k = 0; k < N; k++) acc = sad_ref<CB>(sp + k * lstep, r[k], acc);
        return acc;
    }
    __device__ __forceinline__ void pre_load(int s, PreA &A) const {
        region_load8<G_LT, P_LC, G_LCB, P_NL>(s, ref_luma(A.vx, A.vy), pitchY, A.l);
        if (chroma) {
            const long long co = ref_chroma_off(A.vx, A.vyc);
            region_load8<G_CT, P_CC, G_CCB, P_NC>(s, refU + co, pitchC, A.u);
            region_load8<G_CT, P_CC, G_CCB, P_NC>(s, refV + co, pitchC, A.v);
        }
    }
    __device__ __forceinline__ void pre_sad(int s, const PreA &A, unsigned &aL, unsigned &aC) const {
        aL = region_sad8<G_LT, P_LC, G_LCB, G_LROWB, P_NL>(s, lds, A.l, aL);
        if (chroma) {
            aC = region_sad8<G_CT, P_CC, G_CCB, G_CROWB, P_NC>(s, lds + G_UOFF, A.u, aC);
            aC = region_sad8<G_CT, P_CC, G_CCB, G_CROWB, P_NC>(s, lds + G_VOFF, A.v, aC);
        }
    }

    // ---- SATD / luma sums for the dct = 5..10 cost modes (rare configurations: plain per-lane loops, any geometry) ----
    // sum |H4 * D * H4^T| of one 4x4 block, SADFunctions.cpp:581-637 (the reference's packed form is an exact emulation
    // of this plain one)
    __device__ __forceinline__ unsigned had4x4(const lds_u8 *s, int sp, gl_u8 *r, long long rp) const {
        int t[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int a[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sv = BPS == 1 ? (int)*(const LDS_AS unsigned char *)(s + i * sp + k) : (int)*(const LDS_AS unsigned short *)(s + i * sp + 2 * k);
                const int rv = BPS == 1 ? (int)*(GL_AS const unsigned char *)(r + i * rp + k) : (int)*(GL_AS const uh1 *)(r + i * rp + 2 * k);
                a[k] = sv - rv;
            }
            const int t0 = a[0] + a[1], t1 = a[0] - a[1], t2 = a[2] + a[3], t3 = a[2] - a[3];
            t[i][0] = t0 + t2; t[i][2] = t0 - t2; t[i][1] = t1 + t3; t[i][3] = t1 - t3;
        }
        unsigned sum = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int t0 = t[0][i] + t[1][i], t1 = t[0][i] - t[1][i], t2 = t[2][i] + t[3][i], t3 = t[2][i] - t[3][i];
            sum += (unsigned)abs(t0 + t2) + (unsigned)abs(t1 + t3) + (unsigned)abs(t0 - t2) + (unsigned)abs(t1 - t3);
        }
        return sum;
    }
    // this lane's share of Satd_C (SADFunctions.cpp:686-710): one 4x4 block, otherwise 8x4 partitions, each (a + b) >> 1
    __device__ __forceinline__ unsigned eval_satd(int s, int logG, int vx, int vy) const {
        gl_u8 *ref = ref_luma(vx, vy);
        const int G = 1 << logG;
        if (blkW == 4 && blkH == 4) return s == 0 ? had4x4(lds, lumaRowB, ref, pitchY) >> 1 : 0u;
        const int ppr = blkW >> 3, np = ppr * (blkH >> 2);
        unsigned sum = 0;
        for (int t = s; t < np; t += G) {
            const int py = (t / ppr) * 4, px = (t % ppr) * 8;
            const lds_u8 *sp = lds + py * lumaRowB + px * BPS;
            gl_u8 *rp = ref + (long long)py * pitchY + px * BPS;
            sum += (had4x4(sp, lumaRowB, rp, pitchY) + had4x4(sp + 4 * BPS, lumaRowB, rp + 4 * BPS, pitchY)) >> 1;
        }
        return sum;
    }
    // this lane's share of luma_c (Luma.cpp:14-25) of a reference block
    __device__ __forceinline__ unsigned eval_luma_ref(int s, int logG, int vx, int vy) const {
        gl_u8 *ref = ref_luma(vx, vy);
        const int G = 1 << logG, n = blkW * blkH;
        unsigned sum = 0;
        for (int t = s; t < n; t += G) {
            const int y = t / blkW, x = t - y * blkW;
            sum += BPS == 1 ? (unsigned)*(GL_AS const unsigned char *)(ref + (long long)y * pitchY + x) : (unsigned)*(GL_AS const uh1 *)(ref + (long long)y * pitchY + 2 * x);
        }
        return sum;
    }
    // luma sum of the staged source block, whole wave (uniform result)
    __device__ __forceinline__ int src_luma() const {
        const int n = blkW * blkH;
        int sum = 0;
        for (int t = lane_id(); t < n; t += WAVE) {
            const int y = t / blkW, x = t - y * blkW;
            sum += BPS == 1 ? (int)*(const LDS_AS unsigned char *)(lds + y * lumaRowB + x) : (int)*(const LDS_AS unsigned short *)(lds + y * lumaRowB + 2 * x);
        }
        return uni(wave_sum_i32(sum));
    }
    // pobLumaSAD, PlaneOfBlocks.cpp:117-203 (dct 1-4 need FFTW: rejected at create time).  sad / satd / refLuma are the
    // candidate's group totals.
    __device__ __forceinline__ bool satd_wanted_always() const { return dctmode == 5 || (dctmode == 6 && dctweight16 > 0) || (dctmode == 9 && dctweight16 > 1); }
    __device__ __forceinline__ bool satd_by_luma(int refLuma) const {
        const int sh = dctmode == 10 ? 4 : 5;
        return abs(srcLuma - refLuma) > ((srcLuma + refLuma) >> sh);
    }
    __device__ __forceinline__ unsigned luma_cost(unsigned sadU, unsigned satdU, bool lumaHit) const {
        long long sad = sadU; const long long d = satdU;
        switch (dctmode) {
        case 5: return satdU;
        case 6: if (dctweight16 > 0) sad = (sad * (16 - dctweight16) + d * dctweight16) / 16; break;
        case 7: if (lumaHit) sad = sad / 2 + d / 2; break;
        case 8: if (lumaHit) sad = sad / 4 + d / 2 + d / 4; break;
        case 10: if (lumaHit) sad = sad / 2 + d / 4 + sad / 4; break;
        case 9: if (dctweight16 > 1) { const int h = dctweight16 / 2; sad = (sad * (16 - h) + d * h) / 16; } break;
        default: break;
        }
        return (unsigned)sad;
    }

    // SATD cost modes: the luma term of a candidate (group total aL) becomes a mix of SAD and SATD (:117-203)
    __device__ __forceinline__ unsigned apply_dct(bool ok, int s, int logG, int vx, int vy, unsigned aL) const {
        bool hit = false, want = satd_wanted_always();
        if (dctmode == 7 || dctmode == 8 || dctmode == 10) {
            unsigned rl = 0;
            if (ok) rl = eval_luma_ref(s, logG, vx, vy);
            rl = group_sum(rl, logG);
            hit = satd_by_luma((int)rl);
            want = hit;
        }
        unsigned sd = 0;
        if (ok && want) sd = eval_satd(s, logG, vx, vy);
        sd = group_sum(sd, logG);
        return luma_cost(aL, sd, hit);
    }

    // partial SADs of this lane's share (items s, s+G, ...) of one candidate
    __device__ __forceinline__ void eval_cand(int s, int logG, int vx, int vy, int vyc, unsigned &aL, unsigned &aC) const {
        if (GEO::BW != 0) {
            if (logG == 3) { eval_fixed<3>(s, vx, vy, vyc, aL, aC); return; } // predictor set, hexagon, square
            if (logG == 1) { eval_fixed<1>(s, vx, vy, vyc, aL, aC); return; } // 24-point exhaustive rings
        }
        aL = eval_region_cb(CBL, s, logG, TL, logCL, lumaRowB, lds, ref_luma(vx, vy), pitchY, aL);
        if (chroma) {
            const long long co = ref_chroma_off(vx, vyc);
            // U and V: every lane takes its strided share of each plane
            aC = eval_region_cb(CBC, s, logG, TCp, logCC, chromaRowB, lds + uoff, refU + co, pitchC, aC);
            aC = eval_region_cb(CBC, s, logG, TCp, logCC, chromaRowB, lds + voff, refV + co, pitchC, aC);
        }
    }

    __device__ __forceinline__ static int group_log(int total) { // lanes per candidate = 1 << return
        int logG = 6;
        while (logG > 0 && (64 >> logG) < total) logG--;
        return logG;
    }

    // lane-level results of the last round (all lanes of a candidate's group hold the same values)
    long long rCost, rTot; int rVx, rVy;
    long long prof[16];

    // One round of candidates (PlaneOfBlocks.cpp:219-261 pobCheckMV_Template).  Equivalent to calling the reference's check
    // sequentially in candidate order: every accepted candidate lowers nMinCost and later ones must beat it strictly,
    // so the winner is the FIRST candidate with the minimal cost.  G lanes share one candidate; all lanes of a group
    // carry the same cost, and groups are ordered by lane, so "lowest lane with the minimum" is that first candidate.
    // updateBest=false is pobCheckMVdir (:286-289).  G_ROUNDA is the fixed predictor set of pobPseudoEPZSearch
    // (:832-915) with its own per-candidate cost rules.  Returns the winning candidate index or -1.
    __device__ __forceinline__ int round(const CandGen &gen, int total, bool updateBest) {
        const int lane = lane_id();
        const int logG = group_log(total);
        const int G = 1 << logG, NG = 64 >> logG;
        const int g = lane >> logG, s = lane & (G - 1);
        int winner = -1;
        for (int base = 0; base < total; base += NG) {
            const int c = base + g;
            int vx = 0, vy = 0, vyc;
            bool ok;
            if (gen.kind == G_ROUNDA) {
                if (c == 1) { vx = globalMVPredictor.x; vy = globalMVPredictor.y; }
                else if (c == 2) { vx = predictor.x; vy = predictor.y; }
                else if (c == 3) { vx = predictors[0].x; vy = predictors[0].y; }
                else if (c == 4) { vx = predictors[1].x; vy = predictors[1].y; }
                else if (c == 5) { vx = predictors[2].x; vy = predictors[2].y; }
                else if (c == 6) { vx = predictors[3].x; vy = predictors[3].y; }
                vyc = vy;
                if (c == 0) { vy = zeroMVfieldShifted.y; vyc = 0; } // chroma of the zero candidate ignores fieldShift (:836-839)
                ok = c < total; // all pre-clipped
            } else {
                gen_cand(gen, c, vx, vy);
                vyc = vy;
                ok = c < total && vector_ok(vx, vy);
            }
            unsigned aL = 0, aC = 0;
            const long long pt0 = PROF_T();
            if (ok && ablate != 3) eval_cand(s, logG, vx, vy, vyc, aL, aC);
            const long long pt1 = PROF_T();
            aL = group_sum(aL, logG);
            aC = group_sum(aC, logG);
            if (GEO::DCT && dctmode != 0) aL = apply_dct(ok, s, logG, vx, vy, aL);
            const long long pt2 = PROF_T();
            PROF_ADD(4, pt1 - pt0); PROF_ADD(5, pt2 - pt1); PROF_ADD(8, 1);
            const long long tot = (long long)aL + (chroma ? (long long)aC : 0);
            long long cc = BIG64;
            if (ok) {
                if (gen.kind == G_ROUNDA) {
                    if (c == 0) cc = tot + ((penaltyZero * tot) >> 8);        // :846
                    else if (c == 1) cc = tot + ((pglobal * tot) >> 8);       // :870
                    else if (c == 2) cc = tot;                                // :894
                    else cc = (long long)motion_distortion(vx, vy) + tot;    // pobCheckMV0: no new-vector penalty
                } else {
                    cc = motion_distortion(vx, vy);
                    const long long sad = aL;
                    cc += sad + ((penaltyNew * sad) >> 8);
                    if (chroma) { const long long suv = aC; cc += suv + ((penaltyNew * suv) >> 8); }
                }
            }
            rCost = cc; rTot = tot; rVx = vx; rVy = vy;
            const long long cost = cc < nMinCost ? cc : BIG64;
            long long mc;
            const int w = wave_argmin_ll(cost, &mc);
            if (w >= 0) {
                nMinCost = mc;
                bestMV.sad = bcast_ll(tot, w);
                if (updateBest) { bestMV.x = bcast_i(vx, w); bestMV.y = bcast_i(vy, w); }
                winner = base + (w >> logG);
            }
            PROF_ADD(6, PROF_T() - pt2);
        }
        return winner;
    }

    // PlaneOfBlocks.cpp:419-463
    __device__ __forceinline__ void fetch_predictors(Vec prev, bool havePrev, Vec up, Vec ahead, bool haveAhead) {
        predictors[1] = clip_mv(havePrev ? prev : zeroMVfieldShifted);
        predictors[2] = clip_mv(blky > 0 ? up : zeroMVfieldShifted);
        predictors[3] = clip_mv(haveAhead ? ahead : zeroMVfieldShifted);
        if (blky > 0) {
            auto med = [](int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); };
            predictors[0].x = med(predictors[1].x, predictors[2].x, predictors[3].x);
            predictors[0].y = med(predictors[1].y, predictors[2].y, predictors[3].y);
            long long m = predictors[2].sad > predictors[3].sad ? predictors[2].sad : predictors[3].sad;
            predictors[0].sad = predictors[1].sad > m ? predictors[1].sad : m;
        } else
            predictors[0] = predictors[1];
        if (smallestPlane) predictor = predictors[0];
    }
    // :456-462: lambda shrinks with the predictor's SAD
    __device__ __forceinline__ void scale_lambda() {
        double scale = (double)LSAD / (double)(LSAD + (predictor.sad >> 1));
        nLambda = uni((long long)((double)nLambda * scale * scale));
    }

    int ablate;                                        // developer switch ("ablate" debug option, LAB builds), copied once: never re-read from memory inside the block loop
    int blockSync;                                     // several chains per workgroup: barrier interval in blocks (analyse_kernel, CPW)

    // ---- fast path -----------------------------------------------------------------------------------------------
    // The default search (predictor set, then Hex2 hexagon + square at level 0 or the 24-point exhaustive rings at the
    // coarse levels, no tryMany) as straight-line code with compile-time candidate tables; semantics identical to the
    // general state machine below, which still handles every other pattern and the bad-block rescue.
    // FR_HEXSQ: the hexagon round and, speculatively, the square refinement around the SAME centre in one pass (one memory
    // latency instead of two).  The square results are only used when no hexagon point improved the cost -- then the
    // reference's square refinement runs around the unchanged centre against the unchanged nMinCost, which is exactly
    // what was evaluated; otherwise they are discarded and the square is redone around the moved centre.
    enum { FR_A, FR_HEX6, FR_SQUARE, FR_EXH2, FR_HEXSQ };
    // block SADs bounded by 2^27 -> costs fit 32 bits once the (already int) motion distortion is added with saturation:
    // a saturated cost can never beat nMinCost, which is at most the zero candidate's cost.
    static constexpr bool COST32 = GEO::BW != 0 && GEO::BW * GEO::BH <= 1024;

    // the predictor set of pobPseudoEPZSearch (:832-915): zero, global, hierarchical predictor, median, left, up, ahead
    __device__ __forceinline__ void cand_A(int g, int &vx, int &vy, int &vyc) const {
        asm("" : "+v"(g)); // keeps the seven (g == k) lane masks from being hoisted out of the block loop as scalar pairs that are then spilled
        vx = 0; vy = zeroMVfieldShifted.y;
        vx = g == 1 ? globalMVPredictor.x : vx; vy = g == 1 ? globalMVPredictor.y : vy;
        vx = g == 2 ? predictor.x : vx; vy = g == 2 ? predictor.y : vy;
        vx = g == 3 ? predictors[0].x : vx; vy = g == 3 ? predictors[0].y : vy;
        vx = g == 4 ? predictors[1].x : vx; vy = g == 4 ? predictors[1].y : vy;
        vx = g == 5 ? predictors[2].x : vx; vy = g == 5 ? predictors[2].y : vy;
        vx = g == 6 ? predictors[3].x : vx; vy = g == 6 ? predictors[3].y : vy;
        vyc = g == 0 ? 0 : vy; // chroma of the zero candidate ignores fieldShift (:836-839)
    }
    // early request of the predictor round's reference samples (consumed by round_fast<FR_A, true>)
    __device__ __forceinline__ void pre_issue_A(PreA &A) const {
        const int lane = lane_id();
        const int g = lane >> 3, s = lane & 7;
        cand_A(g, A.vx, A.vy, A.vyc);
        A.ok = g < 7;
        if (A.ok) pre_load(s, A);
    }

    template <int KIND, bool PRE = false> __device__ __forceinline__ int round_fast(int cx, int cy, const PreA *pre = nullptr) {
        constexpr int LOGG = KIND == FR_EXH2 ? 1 : KIND == FR_HEXSQ ? 2 : 3;
        constexpr int TOTAL = KIND == FR_A ? 7 : KIND == FR_HEX6 ? 6 : KIND == FR_SQUARE ? 8 : KIND == FR_HEXSQ ? 14 : 24;
        const int lane = lane_id();
        const int g = lane >> LOGG, s = lane & ((1 << LOGG) - 1);
        int vx, vy, vyc;
        bool ok = g < TOTAL;
        if (KIND == FR_A) {
            if (PRE) { vx = pre->vx; vy = pre->vy; vyc = pre->vyc; }
            else cand_A(g, vx, vy, vyc);
        } else {
            int dx, dy;
            if (KIND == FR_HEX6) { dx = tab8(HEX2X >> 8, g & 7); dy = tab8(HEX2Y >> 8, g & 7); }                  // hex2[g+1], :682-687
            else if (KIND == FR_HEXSQ) {
                const int k = (g + 2) & 7; // square index of groups 6..13
                dx = g < 6 ? tab8(HEX2X >> 8, g & 7) : tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), k);
                dy = g < 6 ? tab8(HEX2Y >> 8, g & 7) : tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), k);
            }
            else if (KIND == FR_SQUARE) { dx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), g & 7); dy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), g & 7); } // :636-658 r=1
            else { // rings 1 and 2 (:786-791): 8 + 16 candidates in reference order
                const int k = g < 8 ? g : g - 8;
                if (g < 8) { dx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), k); dy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), k); }
                else if (k < 8) { dx = tab8(PACK8(-1, -1, 0, 0, 1, 1, -2, 2), k); dy = tab8(PACK8(-2, 2, -2, 2, -2, 2, -1, -1), k); }
                else { dx = tab8(PACK8(-2, 2, -2, 2, -2, -2, 2, 2), k - 8); dy = tab8(PACK8(0, 0, 1, 1, -2, 2, -2, 2), k - 8); }
            }
            vx = cx + dx; vy = cy + dy; vyc = vy;
            ok = ok && vector_ok(vx, vy);
        }
        unsigned aL = 0, aC = 0;
        const long long ft0 = PROF_T();
        if (PRE) {
            if (ok) pre_sad(s, *pre, aL, aC);
        } else if (ok) {
            if (GEO::BW != 0) eval_fixed<LOGG>(s, vx, vy, vyc, aL, aC);
            else eval_cand(s, LOGG, vx, vy, vyc, aL, aC);
        }
        const long long ft1 = PROF_T();
        PROF_ADD(4, ft1 - ft0); PROF_ADD(8, 1);
        aL = group_sum_c<LOGG>(aL);
        aC = group_sum_c<LOGG>(aC);
        int w;
        if (COST32) {
            const int tot = (int)aL + (chroma ? (int)aC : 0);
            int cc;
            if (KIND == FR_A) {
                const int pen = g == 0 ? penaltyZero : (g == 1 ? pglobal : 0);
                const int md = g >= 3 ? motion_distortion(vx, vy) : 0;
                cc = tot + (int)(((long long)pen * tot) >> 8);
                cc = sat_add(md, cc);
            } else {
                cc = (int)aL + ((penaltyNew * (int)aL) >> 8);
                if (chroma) cc += (int)aC + ((penaltyNew * (int)aC) >> 8);
                cc = sat_add(motion_distortion(vx, vy), cc);
            }
            const int lim = nMinCost > 0x7fffffffLL ? 0x7fffffff : (int)nMinCost;
            const bool first = KIND != FR_HEXSQ || g < 6; // FR_HEXSQ: hexagon points first
            const int cost = (ok && first && cc < lim) ? cc : 0x7fffffff;
            int mc;
            w = wave_argmin_i32(cost, &mc);
            if (w >= 0) {
                nMinCost = mc;
                bestMV.sad = (long long)bcast_i(tot, w);
                if (KIND != FR_HEX6 && KIND != FR_HEXSQ) { bestMV.x = bcast_i(vx, w); bestMV.y = bcast_i(vy, w); }
            } else if (KIND == FR_HEXSQ) { // no hexagon point improved: the speculative square results are the real ones
                const int cost2 = (ok && !first && cc < lim) ? cc : 0x7fffffff;
                const int w2 = wave_argmin_i32(cost2, &mc);
                if (w2 >= 0) { nMinCost = mc; bestMV.sad = (long long)bcast_i(tot, w2); bestMV.x = bcast_i(vx, w2); bestMV.y = bcast_i(vy, w2); }
            }
        } else {
            const long long tot = (long long)aL + (chroma ? (long long)aC : 0);
            long long cc;
            if (KIND == FR_A) {
                const long long pen = g == 0 ? penaltyZero : (g == 1 ? pglobal : 0);
                cc = tot + ((pen * tot) >> 8) + (g >= 3 ? (long long)motion_distortion(vx, vy) : 0);
            } else {
                cc = (long long)motion_distortion(vx, vy) + aL + ((penaltyNew * (long long)aL) >> 8);
                if (chroma) cc += (long long)aC + ((penaltyNew * (long long)aC) >> 8);
            }
            const bool first = KIND != FR_HEXSQ || g < 6;
            const long long cost = (ok && first && cc < nMinCost) ? cc : BIG64;
            long long mc;
            w = wave_argmin_ll(cost, &mc);
            if (w >= 0) {
                nMinCost = mc;
                bestMV.sad = bcast_ll(tot, w);
                if (KIND != FR_HEX6 && KIND != FR_HEXSQ) { bestMV.x = bcast_i(vx, w); bestMV.y = bcast_i(vy, w); }
            } else if (KIND == FR_HEXSQ) {
                const long long cost2 = (ok && !first && cc < nMinCost) ? cc : BIG64;
                const int w2 = wave_argmin_ll(cost2, &mc);
                if (w2 >= 0) { nMinCost = mc; bestMV.sad = bcast_ll(tot, w2); bestMV.x = bcast_i(vx, w2); bestMV.y = bcast_i(vy, w2); }
            }
        }
        PROF_ADD(6, PROF_T() - ft1);
        return w >= 0 ? (w >> LOGG) : -1;
    }
    __device__ __forceinline__ static int sat_add(int a, int b) { // b >= 0
        const long long r = (long long)a + b;
        return r > 0x7fffffffLL ? 0x7fffffff : (int)r;
    }

    // returns true when the block is finished; false -> continue in the general state machine at the bad-block check
    template <bool PRE> __device__ __forceinline__ bool search_block_fast(const PreA *pre) {
        if (!PRE) globalMVPredictor = clip_mv(globalMVPredictor); // cumulative clip (:859); done before the early request otherwise
        nMinCost = BIG64;
        round_fast<FR_A, PRE>(0, 0, pre);
        if (searchType == SearchHex2) { // pobHex2Search :667-724 with i_me_range <= 3: no half-hexagon iterations
            int bmx = bestMV.x, bmy = bestMV.y;
            if (nSearchParam > 1) {
                const int dir = round_fast<FR_HEXSQ>(bmx, bmy); // >= 0: a hexagon point won; < 0: the square around (bmx, bmy) is done too
                if (dir >= 0) {
                    bmx += tab8(HEX2X, dir + 1); bmy += tab8(HEX2Y, dir + 1);
                    bestMV.x = bmx; bestMV.y = bmy;
                    round_fast<FR_SQUARE>(bmx, bmy);
                }
            } else
                round_fast<FR_SQUARE>(bmx, bmy);
        } else
            round_fast<FR_EXH2>(bestMV.x, bestMV.y);
        return !(blkIdx > 1 && bestMV.sad > (badSAD + badSAD * badcount / 16)); // :942
    }

    // pobPseudoEPZSearch (PlaneOfBlocks.cpp:819-968) with pobRefine (:773-816) and every search pattern (:466-769)
    // flattened into one state machine around a SINGLE round() call site, so that the whole block state stays in
    // registers (a pattern-per-function structure would force it into scratch memory).
    // entry: 0 = whole block (false), 1 = resume at the bad-block check after the fast path (true), 2 = pobRefine only (Recalculate)
    __device__ __forceinline__ void search_block(int entry) {
        const bool fromBadCheck = entry != 0;
        enum { PC_ROUNDA, PC_TRY_NEXT, PC_REFINE, PC_EXH, PC_LINE, PC_NSTEP, PC_UMH, PC_UMH_HEX4, PC_HEX, PC_HEX3, PC_SQUARE,
               PC_OT_BEGIN, PC_OT_H0, PC_OT_HLOOP, PC_OT_V0, PC_OT_VLOOP, PC_DM_BEGIN, PC_DM_LOOP, PC_DM_SECOND, PC_DM_DIAG,
               PC_REFINE_END, PC_BADCHECK, PC_BADEXP, PC_FINAL, PC_FINAL_EXP, PC_DONE };
        enum { POST_NONE, POST_ROUNDA, POST_HEX6, POST_HEX3, POST_OT_H0, POST_OT_HLOOP, POST_OT_V0, POST_OT_VLOOP, POST_DM_FIRST, POST_DM_SECOND,
               POST_DM_DIAG, POST_BADEXP };
        enum { Right = 1, Left = 2, Down = 4, Up = 8 };

        // ---- round A: zero, global, predictor, predictors[0..3]
        CandGen gen = { G_ROUNDA, 0, 0, 0, 0, 0, 0 };
        if (!fromBadCheck) {
            globalMVPredictor = clip_mv(globalMVPredictor); // cumulative clip (:859)
            nMinCost = BIG64;
        }
        long long aCost = 0, aTot = 0; // per-candidate values of round A stay in lanes 8*i (tryMany)
        int aVx = 0, aVy = 0;

        // pattern state
        int pc, cont = PC_REFINE_END, post;
        int range = 0, omx = 0, omy = 0, bmx = 0, bmy = 0, dir = -2, it = 0;          // hex2 / umh
        int len = 0, dx = 0, dy = 0, direction = 0, last = 0, sgn = 0;                  // nstep / onetime / diamond
        unsigned listV = 0;                                                             // values of a G_LIST round
        int tryIdx = 0; Vec bestAll; bestAll.x = 0; bestAll.y = 0; bestAll.sad = 0; long long costAll = verybig + 1;
        long long foundSAD = 0; int expI = 0, mvx = 0, mvy = 0;

        pc = entry == 2 ? PC_REFINE : fromBadCheck ? PC_BADCHECK : PC_ROUNDA;
        while (pc != PC_DONE) {
            const long long st0 = PROF_T();
            int total = 0; bool upd = true;
            post = POST_NONE;
            switch (pc) {
            case PC_ROUNDA: total = 7; post = POST_ROUNDA; break; // group 0 (zero) always participates
            case PC_TRY_NEXT: { // :851-932: refine around every predictor, keep the first minimum
                if (tryIdx == 7) { bestMV = bestAll; nMinCost = costAll; pc = PC_BADCHECK; continue; }
                const long long ci = bcast_ll(aCost, tryIdx * 8), ti = bcast_ll(aTot, tryIdx * 8);
                const int cx = bcast_i(aVx, tryIdx * 8), cy = bcast_i(aVy, tryIdx * 8);
                if (tryIdx < 3) { bestMV.x = cx; bestMV.y = cy; bestMV.sad = ti; nMinCost = ci; } // unconditional (:834-846,:872,:896)
                else { nMinCost = verybig + 1; if (ci < nMinCost) { bestMV.x = cx; bestMV.y = cy; bestMV.sad = ti; nMinCost = ci; } } // :913-915
                pc = PC_REFINE; continue;
            }
            case PC_REFINE: // :773-816
                cont = PC_REFINE_END;
                switch (searchType) {
                case SearchOnetime: len = nSearchParam; pc = PC_OT_BEGIN; break;
                case SearchNstep: len = nSearchParam; pc = PC_NSTEP; break;
                case SearchLogarithmic: len = nSearchParam; pc = PC_DM_BEGIN; break;
                case SearchExhaustive: pc = PC_EXH; break;
                case SearchHex2: range = nSearchParam; pc = PC_HEX; break;
                case SearchUMH: range = nSearchParam; omx = bestMV.x; omy = bestMV.y; pc = PC_UMH; break;
                default: pc = PC_LINE; break;
                }
                continue;
            case PC_EXH: // rings 1..n around a fixed centre: all candidates known up front -> one round (:786-791)
                gen = { G_RINGS, bestMV.x, bestMV.y, nSearchParam, 0, 0, 0 }; total = 4 * nSearchParam * (nSearchParam + 1); pc = cont; break;
            case PC_LINE: // :799-815
                gen = { searchType == SearchHorizontal ? G_LINEH : G_LINEV, bestMV.x, bestMV.y, 0, 0, 0, 0 }; total = 2 * nSearchParam; pc = cont; break;
            case PC_NSTEP: // :467-485
                if (len <= 0) { pc = cont; continue; }
                gen = { G_NSTEP, bestMV.x, bestMV.y, len, 0, 0, 0 }; total = 8; len--; break;
            case PC_UMH: { // :743-769, cross first (:728-739)
                int nh = 0;
                for (int i = 1; i < range; i += 2) nh++;
                pc = PC_UMH_HEX4;
                if (nh == 0) continue;
                gen = { G_CROSS, omx, omy, nh, 0, 0, 0 }; total = 4 * nh; break;
            }
            case PC_UMH_HEX4: {
                int nrings = 0;
                { int i = 1; do { nrings++; } while (++i <= range / 4); }
                gen = { G_HEX4, omx, omy, 0, 0, 0, 0 }; total = 16 * nrings; pc = PC_HEX; break;
            }
            case PC_HEX: // pobHex2Search :667-724
                dir = -2; bmx = bestMV.x; bmy = bestMV.y;
                if (range <= 1) { pc = PC_SQUARE; continue; }
                gen = { G_HEX6, bmx, bmy, 0, 0, 0, 0 }; total = 6; upd = false; post = POST_HEX6; break;
            case PC_HEX3: { // half hexagon, not overlapping the previous iteration (:694-713)
                if (!(it < range / 2 && vector_ok(bmx, bmy))) { bestMV.x = bmx; bestMV.y = bmy; pc = PC_SQUARE; continue; }
                const int odir = (dir + 1 + 5) % 6; // mod6m1[dir + 1], :662
                gen = { G_HEX3, bmx, bmy, odir, 0, 0, 0 }; total = 3; upd = false; post = POST_HEX3; break;
            }
            case PC_SQUARE: // :723
                gen = { G_RING, bmx, bmy, 1, 1, 0, 0 }; total = 8; pc = cont; break;
            // ---- pobOneTimeSearch :489-527, for (i = param; i > 0; i /= 2)
            case PC_OT_BEGIN:
                if (len <= 0) { pc = cont; continue; }
                dx = bestMV.x; dy = bestMV.y; direction = 0; pc = PC_OT_H0; continue;
            case PC_OT_H0:
                gen = { G_LIST, dx, dy, len, 0, PACK4(-1, 1, 0, 0), PACK4(0, 0, 0, 0) }; total = 2; listV = PACK4(2, 1, 0, 0); post = POST_OT_H0; break;
            case PC_OT_HLOOP:
                if (!direction) { pc = PC_OT_V0; continue; }
                direction = 0; dx += sgn * len;
                gen = { G_SINGLE, dx + sgn * len, dy, 0, 0, 0, 0 }; total = 1; post = POST_OT_HLOOP; break;
            case PC_OT_V0:
                gen = { G_LIST, dx, dy, len, 0, PACK4(0, 0, 0, 0), PACK4(-1, 1, 0, 0) }; total = 2; listV = PACK4(2, 1, 0, 0); post = POST_OT_V0; break;
            case PC_OT_VLOOP:
                if (!direction) { len /= 2; pc = PC_OT_BEGIN; continue; }
                direction = 0; dy += sgn * len;
                gen = { G_SINGLE, dx, dy + sgn * len, 0, 0, 0, 0 }; total = 1; post = POST_OT_VLOOP; break;
            // ---- pobDiamondSearch :531-632, for (i = param; i > 0; i /= 2)
            case PC_DM_BEGIN:
                if (len <= 0) { pc = cont; continue; }
                direction = 15; pc = PC_DM_LOOP; continue;
            case PC_DM_LOOP: {
                if (direction <= 0) { len /= 2; pc = PC_DM_BEGIN; continue; }
                dx = bestMV.x; dy = bestMV.y; last = direction; direction = 0;
                unsigned lx = 0, ly = 0; listV = 0; int n = 0; // hinted directions first (:556-563)
                if (last & Right) { lx |= (unsigned)(unsigned char)1 << (8 * n); listV |= (unsigned)Right << (8 * n); n++; }
                if (last & Left) { lx |= (unsigned)(unsigned char)-1 << (8 * n); listV |= (unsigned)Left << (8 * n); n++; }
                if (last & Down) { ly |= (unsigned)(unsigned char)1 << (8 * n); listV |= (unsigned)Down << (8 * n); n++; }
                if (last & Up) { ly |= (unsigned)(unsigned char)-1 << (8 * n); listV |= (unsigned)Up << (8 * n); n++; }
                gen = { G_LIST, dx, dy, len, 0, lx, ly }; total = n; post = POST_DM_FIRST; break;
            }
            case PC_DM_SECOND: // one direction improved: test the two perpendicular ones from the new best (:567-579)
                if (last & (Right + Left)) { gen = { G_LIST, dx, dy, len, 0, PACK4(0, 0, 0, 0), PACK4(1, -1, 0, 0) }; listV = PACK4(Down, Up, 0, 0); }
                else { gen = { G_LIST, dx, dy, len, 0, PACK4(1, -1, 0, 0), PACK4(0, 0, 0, 0) }; listV = PACK4(Right, Left, 0, 0); }
                total = 2; post = POST_DM_SECOND; break;
            case PC_DM_DIAG: { // nothing improved: diagonals inferred from the last direction (:583-630)
                unsigned lx, ly;
                switch (last) {
                case Right: lx = PACK4(1, 1, 0, 0); ly = PACK4(1, -1, 0, 0); listV = PACK4(Right + Down, Right + Up, 0, 0); total = 2; break;
                case Left: lx = PACK4(-1, -1, 0, 0); ly = PACK4(1, -1, 0, 0); listV = PACK4(Left + Down, Left + Up, 0, 0); total = 2; break;
                case Down: lx = PACK4(1, -1, 0, 0); ly = PACK4(1, 1, 0, 0); listV = PACK4(Right + Down, Left + Down, 0, 0); total = 2; break;
                case Up: lx = PACK4(1, -1, 0, 0); ly = PACK4(-1, -1, 0, 0); listV = PACK4(Right + Up, Left + Up, 0, 0); total = 2; break;
                case Right + Down: lx = PACK4(1, -1, 1, 0); ly = PACK4(1, 1, -1, 0); listV = PACK4(Right + Down, Left + Down, Right + Up, 0); total = 3; break;
                case Left + Down: lx = PACK4(1, -1, -1, 0); ly = PACK4(1, 1, -1, 0); listV = PACK4(Right + Down, Left + Down, Left + Up, 0); total = 3; break;
                case Right + Up: lx = PACK4(1, -1, 1, 0); ly = PACK4(1, -1, -1, 0); listV = PACK4(Right + Down, Left + Up, Right + Up, 0); total = 3; break;
                case Left + Up: lx = PACK4(-1, -1, 1, 0); ly = PACK4(-1, 1, -1, 0); listV = PACK4(Left + Up, Left + Down, Right + Up, 0); total = 3; break;
                default: lx = PACK4(1, -1, 1, -1); ly = PACK4(1, 1, -1, -1); listV = PACK4(Right + Down, Left + Down, Right + Up, Left + Up); total = 4; break;
                }
                gen = { G_LIST, dx, dy, len, 0, lx, ly }; post = POST_DM_DIAG; break;
            }
            case PC_REFINE_END:
                if (entry == 2) pc = PC_DONE;
                else if (tryMany) { if (nMinCost < costAll) { bestAll = bestMV; costAll = nMinCost; } tryIdx++; pc = PC_TRY_NEXT; }
                else pc = PC_BADCHECK;
                continue;
            case PC_BADCHECK: // :938-963
                foundSAD = bestMV.sad;
                if (!(blkIdx > 1 && foundSAD > (badSAD + badSAD * badcount / 16))) { pc = PC_DONE; continue; }
                badcount++;
                if (badrange > 0) { range = badrange * pel; omx = 0; omy = 0; cont = PC_FINAL; pc = PC_UMH; }
                else if (badrange < 0) { expI = 1; pc = PC_BADEXP; }
                else pc = PC_FINAL;
                continue;
            case PC_BADEXP: { // :951-955
                if (!(expI < -badrange * pel)) { pc = PC_FINAL; continue; }
                int n = 0;
                for (int i = -expI + pel; i < expI; i += pel) n++;
                gen = { G_RING, 0, 0, expI, pel, 0, 0 }; total = 4 * n + 4; post = POST_BADEXP; break;
            }
            case PC_FINAL: mvx = bestMV.x; mvy = bestMV.y; expI = 1; pc = PC_FINAL_EXP; continue; // :958-962
            case PC_FINAL_EXP:
                if (!(expI < pel)) { pc = PC_DONE; continue; }
                gen = { G_RING, mvx, mvy, expI, 1, 0, 0 }; total = 8 * expI; expI++; break;
            default: pc = PC_DONE; continue;
            }

            const long long st1 = PROF_T();
            const int w = total > 0 ? round(gen, total, upd) : -1;
            const long long st2 = PROF_T();
            PROF_ADD(10, st1 - st0); PROF_ADD(11, st2 - st1);

            switch (post) {
            case POST_ROUNDA: aCost = rCost; aTot = rTot; aVx = rVx; aVy = rVy; pc = tryMany ? PC_TRY_NEXT : PC_REFINE; if (ablate == 2) pc = PC_DONE; break;
            case POST_HEX6:
                if (w >= 0) dir = w;
                if (dir != -2) { bmx += tab8(HEX2X, dir + 1); bmy += tab8(HEX2Y, dir + 1); it = 1; pc = PC_HEX3; }
                else { bestMV.x = bmx; bestMV.y = bmy; pc = PC_SQUARE; }
                break;
            case POST_HEX3: {
                const int odir = gen.a;
                dir = w >= 0 ? odir - 1 + w : -2;
                if (dir == -2) { bestMV.x = bmx; bestMV.y = bmy; pc = PC_SQUARE; }
                else { bmx += tab8(HEX2X, dir + 1); bmy += tab8(HEX2Y, dir + 1); it++; pc = PC_HEX3; }
                break;
            }
            case POST_OT_H0:
                if (w >= 0) direction = (listV >> (8 * w)) & 0xff;
                if (direction == 1) { sgn = 1; pc = PC_OT_HLOOP; } else if (direction == 2) { sgn = -1; pc = PC_OT_HLOOP; } else pc = PC_OT_V0;
                break;
            case POST_OT_HLOOP: if (w >= 0) direction = 1; pc = PC_OT_HLOOP; break;
            case POST_OT_V0:
                if (w >= 0) direction = (listV >> (8 * w)) & 0xff;
                if (direction == 1) { sgn = 1; pc = PC_OT_VLOOP; } else if (direction == 2) { sgn = -1; pc = PC_OT_VLOOP; } else { len /= 2; pc = PC_OT_BEGIN; }
                break;
            case POST_OT_VLOOP: if (w >= 0) direction = 1; pc = PC_OT_VLOOP; break;
            case POST_DM_FIRST:
                if (w >= 0) direction = (listV >> (8 * w)) & 0xff;
                if (direction) { last = direction; dx = bestMV.x; dy = bestMV.y; pc = PC_DM_SECOND; } else pc = PC_DM_DIAG;
                break;
            case POST_DM_SECOND: if (w >= 0) direction = (listV >> (8 * w)) & 0xff; pc = PC_DM_LOOP; break;
            case POST_DM_DIAG: if (w >= 0) direction = (listV >> (8 * w)) & 0xff; pc = PC_DM_LOOP; break;
            case POST_BADEXP:
                if (bestMV.sad < foundSAD / 4) pc = PC_FINAL; else { expI += pel; pc = PC_BADEXP; }
                break;
            default: break;
            }
            PROF_ADD(12, PROF_T() - st2);
        }
    }


    __device__ static Vec ld_vec(GL_AS const GVec *p) { Vec v; v.x = p->x; v.y = p->y; v.sad = p->sad; return v; }
    __device__ static void st_vec(GL_AS GVec *p, const Vec &v) { p->x = v.x; p->y = v.y; p->sad = v.sad; }
    __device__ static Vec ld_vec_lds16(const LDS_AS Vec *p) { // one ds_read_b128
        const v4u t = *(const LDS_AS v4u *)p;
        Vec v; v.x = (int)t[0]; v.y = (int)t[1]; v.sad = (long long)(((unsigned long long)t[3] << 32) | t[2]);
        return v;
    }
    __device__ static Vec ld_vec_lds(const LDS_AS Vec *p) { Vec v; v.x = p->x; v.y = p->y; v.sad = p->sad; return v; }
    __device__ static void st_vec_lds(LDS_AS Vec *p, const Vec &v) { p->x = v.x; p->y = v.y; p->sad = v.sad; }
    __device__ static int ilog2_dev(int i) { int r = 0; while (i > 1) { i >>= 1; r++; } return r; }

    // global address of item t of the source block at block column/row (bx, by): PlaneOfBlocks.cpp:1058-1079
    __device__ __forceinline__ gl_u8 *src_item_ptr(int t, int bx, int by, int stepX, int stepY, int &loff, int &cb) const {
        int pl, row, xb;
        item(t, pl, row, xb, loff, cb);
        if (pl == 0) return srcY + (long long)(vpad + stepY * by + row) * pitchY + (long long)(hpad + stepX * bx) * BPS + xb;
        return (pl == 1 ? srcU : srcV) + (long long)(cvpad + (stepY >> logyr) * by + row) * pitchC + (long long)(chpad + (stepX >> logxr) * bx) * BPS + xb;
    }

    // ---- specialised source-block staging (compile-time geometry): every lane owns the same NPF items of every block,
    // so their row/column offsets are computed once per level; per block only the block origin is added.
    static constexpr int G_NPF = (G_LT + 2 * G_CT + WAVE - 1) / WAVE;
    static constexpr bool G_PF = GEO::BW != 0 && G_NPF <= PF_MAX;
    int pfG[PF_MAX], pfL[PF_MAX], pfP[PF_MAX]; // per lane: global row/col offset, LDS offset, plane (0,1,2; -1 = no item)

    __device__ __forceinline__ void pf_setup() {
        const int l = lane_id();
#pragma unroll
        for (int k = 0; k < G_NPF; k++) {
            const int t = l + k * WAVE;
            const int TT_ = G_LT + (chroma ? 2 * G_CT : 0);
            if (t < G_LT) {
                const int row = t / G_LC, xb = (t % G_LC) * G_LCB;
                pfP[k] = 0; pfG[k] = (int)(row * pitchY) + xb; pfL[k] = row * G_LROWB + xb;
            } else if (t < TT_) {
                int tt = t - G_LT;
                const int pl = tt >= G_CT ? 2 : 1;
                if (pl == 2) tt -= G_CT;
                const int row = tt / G_CC, xb = (tt % G_CC) * G_CCB;
                pfP[k] = pl; pfG[k] = (int)(row * pitchC) + xb; pfL[k] = (pl == 2 ? G_VOFF : G_UOFF) + row * G_CROWB + xb;
            } else { pfP[k] = -1; pfG[k] = 0; pfL[k] = 0; }
        }
    }
    __device__ __forceinline__ void pf_issue(int bx, int by, int stepX, int stepY, A4x32 *pf) const {
        const long long offY = (long long)(vpad + stepY * by) * pitchY + (long long)(hpad + stepX * bx) * BPS;
        const long long offC = (long long)(cvpad + (stepY >> logyr) * by) * pitchC + (long long)(chpad + (stepX >> logxr) * bx) * BPS;
#pragma unroll
        for (int k = 0; k < G_NPF; k++) {
            const int pl = pfP[k];
            if (pl < 0) continue;
            gl_u8 *g = (pl == 0 ? srcY + offY : (pl == 1 ? srcU : srcV) + offC) + pfG[k];
            if (G_LCB == G_CCB) pf[k] = ld_chunk_g(g, G_LCB);
            else pf[k] = pl == 0 ? ld_chunk_g(g, G_LCB) : ld_chunk_g(g, G_CCB);
        }
    }
    __device__ __forceinline__ void pf_store(const A4x32 *pf) const {
#pragma unroll
        for (int k = 0; k < G_NPF; k++) {
            const int pl = pfP[k];
            if (pl < 0) continue;
            if (G_LCB == G_CCB) st_chunk_l(lds + pfL[k], pf[k], G_LCB);
            else if (pl == 0) st_chunk_l(lds + pfL[k], pf[k], G_LCB);
            else st_chunk_l(lds + pfL[k], pf[k], G_CCB);
        }
    }

    // plane geometry, pointers and block staging layout of one level
    __device__ __forceinline__ void setup_geometry(int lvl) {
        const ALevel &L = P.lv[lvl];
        level = lvl; nBlkX = L.nBlkX; nBlkY = L.nBlkY; pel = L.pel; logPel = L.logPel;
        chroma = P.chroma; logxr = P.logxr; logyr = P.logyr; blkW = P.blkX; blkH = P.blkY; meander = P.meander; verybig = P.verybigSAD;
        pw = L.pw; ph = L.ph; hpad = L.hpad; vpad = L.vpad; chpad = L.chpad; cvpad = L.cvpad;
        srcY = (gl_u8 *)(J.src[0] + L.off[0]); refY = (gl_u8 *)(J.ref[0] + L.off[0]);
        srcU = (gl_u8 *)(J.src[1] + L.off[1]); refU = (gl_u8 *)(J.ref[1] + L.off[1]);
        srcV = (gl_u8 *)(J.src[2] + L.off[2]); refV = (gl_u8 *)(J.ref[2] + L.off[2]);
        pitchY = P.pitch[0]; pitchC = P.pitch[1]; pstrideY = L.pstride[0]; pstrideC = L.pstride[1];
        cBlkX = P.blkX / P.xr; cBlkY = P.blkY / P.yr;
        lumaRowB = P.blkX * BPS; chromaRowB = cBlkX * BPS;
        CBL = min(16, lumaRowB); CBC = min(16, chromaRowB);
        logCL = ilog2_dev(lumaRowB / CBL); logCC = ilog2_dev(chromaRowB / CBC);
        TL = P.blkY << logCL; TCp = cBlkY << logCC;
        TT = TL + (chroma ? 2 * TCp : 0);
        uoff = P.blkY * lumaRowB; voff = uoff + cBlkY * chromaRowB;
        unsigned char *rec = J.blob + L.blobOff;
        vectors = (GL_AS GVec *)(rec + 4);
        if (lane_id() == 0) *(int *)rec = 4 + nBlkX * nBlkY * 16; // pobWriteHeaderToArray :413-416
    }

    // GroupOfPlanes.c:69-125 + PlaneOfBlocks.cpp:971-1131 for one level
    __device__ __forceinline__ void search_level(int lvl, Vec *globalMV, int *meanLumaChange, GL_AS const GVec *coarse, int coarseBlkX, int coarseBlkY, int coarseLogPel) {
        const int l = lane_id();
        setup_geometry(lvl);
        const int nBlk = nBlkX * nBlkY;
        smallestPlane = lvl == P.nLevels - 1;
        // ---- hierarchical predictors into vectors[] (pobInterpolatePrediction :1447-1514) or zero (pobInit :355)
        if (!coarse) {
            for (int i = l; i < nBlk; i += WAVE) { Vec z; z.x = 0; z.y = 0; z.sad = 0; st_vec(&vectors[i], z); }
        } else {
            int normFactor = 3 - logPel + coarseLogPel;
            const int mulFactor = normFactor < 0 ? -normFactor : 0;
            normFactor = normFactor < 0 ? 0 : normFactor;
            const int normov = (P.blkX - P.ovX) * (P.blkY - P.ovY);
            const int aoddx = P.blkX * 3 - P.ovX * 2, aevenx = P.blkX * 3 - P.ovX * 4;
            const int aoddy = P.blkY * 3 - P.ovY * 2, aeveny = P.blkY * 3 - P.ovY * 4;
            const double scaleov = 1.0 / normov;
            for (int index = l; index < nBlk; index += WAVE) {
                const int ly = index / nBlkX, k = index - ly * nBlkX;
                int i = k, j = ly;
                if (i >= 2 * coarseBlkX) i = 2 * coarseBlkX - 1;
                if (j >= 2 * coarseBlkY) j = 2 * coarseBlkY - 1;
                const int offy = -1 + 2 * (j % 2), offx = -1 + 2 * (i % 2);
                Vec v1, v2, v3, v4;
                const bool ex = (i == 0) || (i >= 2 * coarseBlkX - 1), ey = (j == 0) || (j >= 2 * coarseBlkY - 1);
                v1 = ld_vec(&coarse[i / 2 + (j / 2) * coarseBlkX]);
                if (ex && ey) { v2 = v3 = v4 = v1; }
                else if (ex) { v2 = v1; v3 = v4 = ld_vec(&coarse[i / 2 + (j / 2 + offy) * coarseBlkX]); }
                else if (ey) { v2 = v1; v3 = v4 = ld_vec(&coarse[i / 2 + offx + (j / 2) * coarseBlkX]); }
                else {
                    v2 = ld_vec(&coarse[i / 2 + offx + (j / 2) * coarseBlkX]);
                    v3 = ld_vec(&coarse[i / 2 + (j / 2 + offy) * coarseBlkX]);
                    v4 = ld_vec(&coarse[i / 2 + offx + (j / 2 + offy) * coarseBlkX]);
                }
                Vec o; long long temp_sad;
                if (P.ovX == 0 && P.ovY == 0) {
                    o.x = 9 * v1.x + 3 * v2.x + 3 * v3.x + v4.x;
                    o.y = 9 * v1.y + 3 * v2.y + 3 * v3.y + v4.y;
                    temp_sad = 9 * v1.sad + 3 * v2.sad + 3 * v3.sad + v4.sad + 8;
                } else if (P.ovX <= (P.blkX >> 1) && P.ovY <= (P.blkY >> 1)) {
                    const int ax1 = (offx > 0) ? aoddx : aevenx, ax2 = (P.blkX - P.ovX) * 4 - ax1;
                    const int ay1 = (offy > 0) ? aoddy : aeveny, ay2 = (P.blkY - P.ovY) * 4 - ay1;
                    const long long a11 = ax1 * ay1, a12 = ax1 * ay2, a21 = ax2 * ay1, a22 = ax2 * ay2;
                    o.x = (int)((double)(a11 * v1.x + a21 * v2.x + a12 * v3.x + a22 * v4.x) * scaleov);
                    o.y = (int)((double)(a11 * v1.y + a21 * v2.y + a12 * v3.y + a22 * v4.y) * scaleov);
                    temp_sad = (long long)((double)(a11 * v1.sad + a21 * v2.sad + a12 * v3.sad + a22 * v4.sad) * scaleov);
                } else {
                    o.x = (v1.x + v2.x + v3.x + v4.x) << 2;
                    o.y = (v1.y + v2.y + v3.y + v4.y) << 2;
                    temp_sad = (v1.sad + v2.sad + v3.sad + v4.sad + 2) << 2;
                }
                o.x = (o.x >> normFactor) * (1 << mulFactor);
                o.y = (o.y >> normFactor) * (1 << mulFactor);
                o.sad = temp_sad >> 4;
                st_vec(&vectors[index], o);
            }
        }
        // the interpolated field is re-read (by other lanes) during the scan: make it visible once per level
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_s_barrier();

        // ---- plane scan set-up (doPobSearchMVs :979-1034)
        const bool coarsest = smallestPlane;
        const int st = P.searchType, cst = P.searchTypeCoarse;
        const bool hv = st == SearchHorizontal || st == SearchVertical;
        if (coarsest) { searchType = (P.nLevels == 1 || hv) ? st : cst; nSearchParam = (P.nLevels == 1) ? P.nPelSearch : P.nSearchParam; }
        else { searchType = (lvl == 0 || hv) ? st : cst; nSearchParam = (lvl == 0) ? P.nPelSearch : P.nSearchParam; }
        tryMany = coarsest ? (P.tryMany && P.nLevels > 1) : (P.tryMany && lvl > 0);
        const int fieldShift = (lvl == 0) ? J.fieldShift : 0;
        badSAD = P.badSAD; badrange = P.badrange;
        zeroMVfieldShifted.x = 0; zeroMVfieldShifted.y = fieldShift; zeroMVfieldShifted.sad = 0;
        globalMVPredictor.x = pel * globalMV->x;
        globalMVPredictor.y = pel * globalMV->y + fieldShift;
        globalMVPredictor.sad = globalMV->sad;
        int nLambdaLevel = P.lambda / (pel * pel);
        const int nScale = 1 << lvl;
        if (P.plevel == 1) nLambdaLevel = nLambdaLevel * nScale;
        else if (P.plevel == 2) nLambdaLevel = nLambdaLevel * nScale * nScale;
        penaltyZero = P.pzero; pglobal = P.global ? P.pglobal : P.pzero; badcount = 0;
        penaltyNew = P.pnew; LSAD = P.lsad;
        dctmode = GEO::DCT ? P.dctmode : 0; sumLumaChange = 0; srcLuma = 0;
        dctweight16 = min(16, abs(*meanLumaChange) / (blkW * blkH)); // PlaneOfBlocks.cpp:981

        LDS_AS Vec *rowbuf = (LDS_AS Vec *)(lds + ldsRow);
        const int stepX = P.blkX - P.ovX, stepY = P.blkY - P.ovY;
        const int hps = hpad >> lvl, vps = vpad >> lvl; // :1091-1092
        Vec prev; prev.x = 0; prev.y = 0; prev.sad = 0;

        // software pipeline: the next block's hierarchical predictors and source samples are static data, so their
        // global loads are issued one block ahead and only consumed at the top of the next iteration.
        const bool usePF = TT <= PF_MAX * WAVE;
        // PF_LATE: the source block is fetched at the top of its own block instead of one block ahead -- one exposed latency per
        // block, but its registers are not live across the search (the 16-bit 32x32 kernel needs that to fit 256 registers)
        // (measured r1: 8K 32x32 at two chains per SIMD 48.2 -> 55.6 fps with it; 4K 16x16, which fits anyway, 260 -> 257 fps)
        constexpr bool PF_LATE = WPE == 2 && G_NPF > 1;
        A4x32 pf[PF_MAX];
        if (G_PF) pf_setup();
        // The hierarchical predictors (this level's interpolated vectors, static during the level) of the current block
        // row and of the row below sit in two LDS row buffers, refilled once per block row with coalesced loads: the
        // per-block predictor fetch is two LDS reads instead of two global loads whose registers the compiler had to
        // park (and wait for) before every search.
        // Measured (r1, A/B in one session): +5 % at full load / +9 % unloaded on 4K 16-bit, -7 % on 1080p 8-bit, whose
        // lighter kernels keep the prefetched vectors in registers for free -- hence the compile-time choice.
                // (Was on for the 16-bit kernels while one chain per workgroup was the rule: +5 % there.  With four / eight chains per
        // workgroup it no longer measures -- 223.2 against 224.6 fps -- and its 15 KiB per chain would keep eight chains out of a CU.)
        constexpr bool PRED_ROWS = false;
        const int predStride = (ldsHist - ldsRow) / 48; // host layout: [row buffer | 2 predictor rows], 16 bytes per block
        LDS_AS Vec *predRows = (LDS_AS Vec *)(lds + ldsRow + predStride * 16);
        auto load_pred_row = [&](int row) {
            LDS_AS Vec *dst = predRows + (row & 1) * predStride;
            GL_AS const GVec *srcv = vectors + row * nBlkX;
            for (int i0 = 0; i0 < nBlkX; i0 += 4 * WAVE) {
                Vec t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int i = i0 + k * WAVE + l; if (i < nBlkX) t[k] = ld_vec(&srcv[i]); }
#pragma unroll
                for (int k = 0; k < 4; k++) { const int i = i0 + k * WAVE + l; if (i < nBlkX) st_vec_lds(&dst[i], t[k]); }
            }
        };
        if (PRED_ROWS) load_pred_row(0);
        Vec nSelf, nBelow; // !PRED_ROWS: the next block's predictors, requested one block ahead into registers
        nSelf.x = nSelf.y = 0; nSelf.sad = 0; nBelow = nSelf;
        int nextIb = 0, nextBy = 0; // scan position of the block being prefetched (:1037-1056), advanced without divisions
        auto prefetch = [&]() {
            const int by = nextBy;
            const int bx = (by % 2 == 0 || meander == 0) ? nextIb : nBlkX - 1 - nextIb;
            if (++nextIb == nBlkX) { nextIb = 0; nextBy++; }
            if (!PRED_ROWS) {
                const int dir = (by % 2 == 0 || meander == 0) ? 1 : -1;
                const int idx = by * nBlkX + bx;
                nSelf = ld_vec(&vectors[idx]); // consumed (and made uniform) at the top of the next block: no wait here
                const bool aheadColN = (dir == 1 && bx < nBlkX - 1) || (dir == -1 && bx > 0);
                nBelow.x = 0; nBelow.y = 0; nBelow.sad = 0;
                if (by < nBlkY - 1 && aheadColN) nBelow = ld_vec(&vectors[idx + nBlkX + dir]);
            }
            if (G_PF) { if (!PF_LATE) pf_issue(bx, by, stepX, stepY, pf); }
            else if (usePF) {
#pragma unroll
                for (int k = 0; k < PF_MAX; k++) {
                    const int t = l + k * WAVE;
                    if (t < TT) { int loff, cb; gl_u8 *g = src_item_ptr(t, bx, by, stepX, stepY, loff, cb); pf[k] = ld_chunk_g(g, cb); }
                }
            }
        };
        prefetch();
        int curIb = 0, curBy = 0;
        const bool fast = !tryMany && dctmode == 0 && ((searchType == SearchHex2 && nSearchParam <= 3) || (searchType == SearchExhaustive && nSearchParam == 2));
        // early request of the predictor round (specialised kernels, reference samples from global memory): compile-time, one variant of
        // the fast path per kernel.  Measured (r1): at one chain per SIMD +1.7 % on 1080p 8-bit and -11 % on 4K 16-bit, at two chains per
        // SIMD -4 % / 0 %: the other chain hides that latency already.  Off.
        constexpr bool EARLY_K = false;
        const bool early = EARLY_K && fast;
        PreA preA;
        for (int n = 0; n < nBlk; n++) {
            const long long bt0 = PROF_T();
            if (blockSync && (curIb & (blockSync - 1)) == 0) __builtin_amdgcn_s_barrier(); // every blockSync blocks (a power of two) and at every row start
            blky = curBy;
            blkx = (blky % 2 == 0 || meander == 0) ? curIb : nBlkX - 1 - curIb;
            const bool rowStart = curIb == 0;
            if (++curIb == nBlkX) { curIb = 0; curBy++; }
            blkScanDir = (blky % 2 == 0 || meander == 0) ? 1 : -1;
            blkIdx = blky * nBlkX + blkx;
            x0 = hpad + stepX * blkx; y0 = vpad + stepY * blky;
            cx0 = chpad + (stepX >> logxr) * blkx; cy0 = cvpad + (stepY >> logyr) * blky; // :1048-1051,1116-1118,1123-1127

            if (PRED_ROWS && rowStart && blky + 1 < nBlkY) load_pred_row(blky + 1); // first block of a row: fetch the row below
            // hierarchical predictors from the LDS row buffers (:1100, :441-447)
            // (three 16-byte LDS reads issued back to back, one wait: this block, below-ahead, and the result above)
            const bool aheadCol = (blkScanDir == 1 && blkx < nBlkX - 1) || (blkScanDir == -1 && blkx > 0);
            const bool useBelow = (blky < nBlkY - 1) && aheadCol;
            const int colAhead = min(max(blkx + blkScanDir, 0), nBlkX - 1);
            const Vec vSelf = PRED_ROWS ? ld_vec_lds16(&predRows[(blky & 1) * predStride + blkx]) : nSelf;
            const Vec vBelow = PRED_ROWS ? ld_vec_lds16(&predRows[((blky + 1) & 1) * predStride + colAhead]) : nBelow;
            const Vec vUp = ld_vec_lds16(&rowbuf[blkx]);
            const Vec self = uni(vSelf);
            Vec below = uni(vBelow), up = uni(vUp);
            if (!useBelow) { below.x = 0; below.y = 0; below.sad = 0; }
            if (blky == 0) { up.x = 0; up.y = 0; up.sad = 0; }
            // consume the prefetched data: source block -> LDS (PlaneOfBlocks.cpp:1058-1079)
            if (G_PF) { if (PF_LATE) pf_issue(blkx, blky, stepX, stepY, pf); pf_store(pf); }
            else if (usePF) {
#pragma unroll
                for (int k = 0; k < PF_MAX; k++) {
                    const int t = l + k * WAVE;
                    if (t < TT) { int pl, row, xb, loff, cb; item(t, pl, row, xb, loff, cb); st_chunk_l(lds + loff, pf[k], cb); }
                }
            } else {
                for (int t = l; t < TT; t += WAVE) {
                    int loff, cb;
                    gl_u8 *g = src_item_ptr(t, blkx, blky, stepX, stepY, loff, cb);
                    A4x32 a = ld_chunk_g(g, cb);
                    st_chunk_l(lds + loff, a, cb);
                }
            }
            const long long btA = PROF_T();
            if (!early && n + 1 < nBlk) prefetch();
            const long long btW = PROF_T();
            const long long btB = btW;
            PROF_ADD(13, btA - bt0); PROF_ADD(14, btW - btA); PROF_ADD(15, btB - btW);

            nDxMax = (pw - x0 - blkW - hpad + hps) << logPel; // :1094-1097
            nDyMax = (ph - y0 - blkH - vpad + vps) << logPel;
            nDxMin = -((x0 - hpad + hps) << logPel);
            nDyMin = -((y0 - vpad + vps) << logPel);

            const bool useUpAhead = !useBelow && (blky > 0) && aheadCol; // last block row only
            Vec ahead = below;
            if (useUpAhead) ahead = uni(ld_vec_lds(&rowbuf[blkx + blkScanDir]));

            nLambda = blky == 0 ? 0 : nLambdaLevel; // :1081-1084
            predictor = clip_mv(self);              // :1100
            const bool havePrev = (blkScanDir == 1 && blkx > 0) || (blkScanDir == -1 && blkx < nBlkX - 1);
            fetch_predictors(prev, havePrev, up, ahead, useBelow || useUpAhead);
            if (early) { // the predictor round's reference samples are requested now; the rest of the prologue hides their latency
                globalMVPredictor = clip_mv(globalMVPredictor); // cumulative clip (:859)
                pre_issue_A(preA);
                if (n + 1 < nBlk) prefetch();
            }
            scale_lambda();

            __builtin_amdgcn_wave_barrier(); // single wave: DS ops are in order; keep the compiler from moving LDS reads above the staging writes
            if (GEO::DCT && (dctmode == 7 || dctmode == 8 || dctmode == 10)) srcLuma = src_luma(); // :829-830 (only these modes read it)
            const long long bt1 = PROF_T();
            if (ablate == 1) { bestMV = predictor; bestMV.sad = 0; }
            // (the hints matter: the general state machine is an inner loop, which the register allocator would otherwise favour
            // over the straight-line path that actually runs; measured +2 % at 4K16)
            else if (__builtin_expect(fast, 1)) { if (__builtin_expect(!search_block_fast<EARLY_K>(&preA), 0)) search_block(1); }
            else search_block(0);
            const long long bt2 = PROF_T();
            __builtin_amdgcn_wave_barrier();

            if (GEO::DCT && smallestPlane && (dctmode == 6 || dctmode == 9)) // :1109-1110 (feeds dctweight16 of the finer levels)
                sumLumaChange += uni(wave_sum_i32((int)eval_luma_ref(l, 6, 0, 0))) - src_luma();
            // results: vectors[blkIdx] (:967) == blob row (:1106); the row also stays in LDS for the next row's predictors
            if (l == 0) { st_vec(&vectors[blkIdx], bestMV); st_vec_lds(&rowbuf[blkx], bestMV); }
            prev = bestMV;
            const long long bt3 = PROF_T();
            PROF_ADD(0, bt1 - bt0); PROF_ADD(1, bt2 - bt1); PROF_ADD(2, bt3 - bt2); PROF_ADD(3, 1);
        }
        if (GEO::DCT && smallestPlane) *meanLumaChange = sumLumaChange / nBlk; // :1130
        // vectors[] of this level feed the next level's interpolation / global-MV estimate (other lanes read them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_s_barrier();
    }

    // pobEstimateGlobalMVDoubled, PlaneOfBlocks.cpp:1559-1636 (mode via LDS histogram windows; first maximum wins)
    __device__ __forceinline__ void estimate_global(GL_AS const GVec *v, int nBlk, int freqSizeHalf, Vec *g) {
        const int l = lane_id();
        LDS_AS int *hist = (LDS_AS int *)(lds + ldsHist);
        int med[2];
        for (int c = 0; c < 2; c++) {
            int lo = 0x7fffffff, hi = -0x7fffffff - 1;
            for (int i = l; i < nBlk; i += WAVE) {
                int val = c ? v[i].y : v[i].x;
                int ind = freqSizeHalf + val;
                if (ind >= 0 && ind < 2 * freqSizeHalf) { lo = min(lo, val); hi = max(hi, val); }
            }
            lo = wave_min_i32(lo); hi = wave_max_i32(hi);
            int bestCount = -1, bestVal = lo;
            for (int wbase = lo; wbase <= hi; wbase += histBins) {
                for (int i = l; i < histBins; i += WAVE) hist[i] = 0;
                __builtin_amdgcn_wave_barrier();
                for (int i = l; i < nBlk; i += WAVE) {
                    int val = c ? v[i].y : v[i].x;
                    int ind = freqSizeHalf + val;
                    if (ind >= 0 && ind < 2 * freqSizeHalf && val >= wbase && val < wbase + histBins)
                        __hip_atomic_fetch_add(&hist[val - wbase], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                int lc = -1, lv = 0; // first maximum in ascending order within this lane's stride
                for (int i = l; i < histBins && wbase + i <= hi; i += WAVE) {
                    int cnt = hist[i];
                    if (cnt > lc) { lc = cnt; lv = i; }
                }
                const int mcnt = wave_max_i32(lc);
                const int cand = (lc == mcnt) ? lv : 0x7fffffff;
                const int mv = wave_min_i32(cand);
                if (mcnt > bestCount) { bestCount = mcnt; bestVal = wbase + mv; }
                __builtin_amdgcn_wave_barrier();
            }
            med[c] = bestVal;
        }
        int sx = 0, sy = 0, n = 0;
        for (int i = l; i < nBlk; i += WAVE) {
            int vx = v[i].x, vy = v[i].y;
            if (abs(vx - med[0]) < 6 && abs(vy - med[1]) < 6) { sx += vx; sy += vy; n++; }
        }
        sx = wave_sum_i32(sx); sy = wave_sum_i32(sy); n = wave_sum_i32(n);
        if (n > 0) { g->x = uni(2 * sx / n); g->y = uni(2 * sy / n); }
        else { g->x = uni(2 * med[0]); g->y = uni(2 * med[1]); }
    }
};

// WPE = chains per SIMD the kernel is compiled for.  1: all 512 registers of a SIMD lane, for launches of at most one chain per
// SIMD.  2: at most 256 registers; only the 8-bit kernels fit that without spilling, and they gain 53 % from the second chain
// when a launch carries two chains per SIMD (1080p: 961 -> 1475 fps) -- their loads are light on the CU's texture path.  The
// 16-bit kernels do not (4K16: no gain even without spills; five chains per CU take as long as four, the shared L1/TA is
// the limit), so they stay at WPE = 1.
// CPW = chains per workgroup (1 or 4).  4: the waves of a workgroup are four chains the host has ordered so that they search
// the SAME reference frame (different current frames); a workgroup barrier per block keeps them on the same block, so the
// reference lines one of them pulls into the CU's L1 (and the XCD's L2) serve the others.
template <int BPS, typename GEO, int WPE = 1, int CPW = 1>
__global__ __launch_bounds__(64 * CPW, WPE) void analyse_kernel(const AParams *Pp, const AJob *jobs, int njobs, int ldsChain, int syncEvery, int ldsRow, int ldsHist, int histBins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // WPE == 1 is for launches with at most one chain per SIMD.  The host's LDS request limits a CU to four chains, but a
    // kernel that fits 256 VGPRs would let the dispatcher stack two of them on one SIMD while another SIMD idles (measured:
    // -12 % on the 8-bit 8x8 kernel when an unrelated refactoring moved it from 262 to 256 registers).  Touching the last
    // accumulator register pushes the wave's register allocation above 256, i.e. at most one wave per SIMD, for every variant.
    // (Four chains per 256-thread workgroup, one workgroup per CU, was also tried: deterministic too, but 5 % slower at 4K16 --
    // the four chains of a CU then run in lockstep and hit the texture path in the same phases.)
    if (WPE == 1) asm volatile("" ::: "a255");
    const AParams &P = *Pp;
    const int chain = CPW == 1 ? (int)blockIdx.x : uni((int)blockIdx.x * CPW + (int)(threadIdx.x >> 6));
    if (CPW > 1 && chain >= njobs) return; // (a finished wave no longer counts for the workgroup's barriers)
    const AJob &J = jobs[chain];
    const int l = lane_id();
    int *hdr = (int *)J.blob;
    if (!J.valid) { // gopWriteDefaultToArray GroupOfPlanes.c:150-164, pobWriteDefaultToArray PlaneOfBlocks.cpp:1529-1556
        if (l == 0) { hdr[0] = P.blobSize; hdr[1] = 0; }
        for (int lvl = P.nLevels - 1; lvl >= 0; lvl--) {
            const ALevel &L = P.lv[lvl];
            unsigned char *rec = J.blob + L.blobOff;
            const int nBlk = L.nBlkX * L.nBlkY;
            if (l == 0) *(int *)rec = 4 + nBlk * 16;
            GVec *v = (GVec *)(rec + 4);
            for (int i = l; i < nBlk; i += WAVE) { GVec d; d.x = 0; d.y = 0; d.sad = P.verybigSAD; v[i] = d; }
        }
        return;
    }
    if (l == 0) { hdr[0] = P.blobSize; hdr[1] = 1; } // GroupOfPlanes.c:77-85
    Searcher<BPS, GEO, WPE> S(P, J);
    S.lds = (lds_u8 *)smem + (CPW == 1 ? 0 : uni((int)(threadIdx.x >> 6)) * ldsChain);
    S.ldsRow = ldsRow; S.ldsHist = ldsHist; S.histBins = histBins;
    S.blockSync = CPW > 1 ? syncEvery : 0;
    S.ablate = uni(P.ablate) & 0xff;
#ifdef MVX_PROFILE
    for (int i = 0; i < 16; i++) S.prof[i] = 0;
    const long long kt0 = PROF_T();
#endif
    Vec globalMV; globalMV.x = 0; globalMV.y = 0; globalMV.sad = -1; // zeroMV, MVAnalysisData.h:79
    int meanLumaChange = 0; // GroupOfPlanes.c:94: set by the smallest plane, weights SATD in dct modes 6 / 9
    GL_AS const GVec *coarse = nullptr;
    int cbx = 0, cby = 0, clp = 0;
    for (int lvl = P.nLevels - 1; lvl >= 0; lvl--) {
        if (coarse && P.global) S.estimate_global(coarse, cbx * cby, 8192 * P.lv[lvl + 1].pel, &globalMV);
        S.search_level(lvl, &globalMV, &meanLumaChange, coarse, cbx, cby, clp);
        coarse = S.vectors; cbx = P.lv[lvl].nBlkX; cby = P.lv[lvl].nBlkY; clp = P.lv[lvl].logPel;
    }
#ifdef MVX_PROFILE
    S.prof[9] = PROF_T() - kt0;
    if (l == 0 && chain == (P.ablate >> 8)) for (int i = 0; i < 16; i++) g_prof[i] = (unsigned long long)S.prof[i]; // MVX_ABLATE = chain << 8
#endif
}

#if defined(MVX_PROFILE) && defined(MVX_PROF_EXPORT)
extern "C" __attribute__((visibility("default"))) int mvx_debug_prof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif



// ---- launch helper shared by the kernel translation units (each instantiates a few geometries so that hipcc can build
// them in parallel)
// ---- mv.Recalculate: PlaneOfBlocks.cpp:1158-1424.  Every block is independent (its predictor comes from the OLD vector
// field, there are no spatial predictors, no bad-block counter), so one workgroup = one wavefront = one BLOCK and the grid
// is blocks x frames; the candidate evaluation and pattern searches are the Searcher's (entry 2 = pobRefine only).
template <int BPS>
__global__ __launch_bounds__(64, 1) void recalc_kernel(const AParams *Pp, const RParams *Rp, const AJob *jobs, int ldsRow, int ldsHist, int histBins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AParams &P = *Pp;
    const RParams &R = *Rp;
    const AJob &J = jobs[blockIdx.y];
    typedef Searcher<BPS, GeoAnyDct> S_t;
    S_t S(P, J);
    S.lds = (lds_u8 *)smem; S.ldsRow = ldsRow; S.ldsHist = ldsHist; S.histBins = histBins;
    S.ablate = 0; S.blockSync = 0;
    for (int i = 0; i < 16; i++) S.prof[i] = 0;
    const int l = lane_id();
    S.setup_geometry(0);
    const int nBlk = S.nBlkX * S.nBlkY, b = blockIdx.x;
    const int valid = J.valid && ((const int *)J.oldBlob)[1] == 1; // MVRecalculate.c:152 fgopIsValid && reference frame inside the clip
    if (b == 0 && l == 0) { int *hdr = (int *)J.blob; hdr[0] = P.blobSize; hdr[1] = valid; }
    if (!valid) { // gopWriteDefaultToArray
        if (l == 0) { Vec d; d.x = 0; d.y = 0; d.sad = P.verybigSAD; S_t::st_vec(&S.vectors[b], d); }
        return;
    }
    S.smallestPlane = 0;
    S.blky = b / S.nBlkX; S.blkx = b - S.blky * S.nBlkX; S.blkIdx = b; S.blkScanDir = 1;
    const int stepX = P.blkX - P.ovX, stepY = P.blkY - P.ovY;
    S.x0 = S.hpad + stepX * S.blkx; S.y0 = S.vpad + stepY * S.blky;
    S.cx0 = S.chpad + (stepX >> S.logxr) * S.blkx; S.cy0 = S.cvpad + (stepY >> S.logyr) * S.blky;
    for (int t = l; t < S.TT; t += WAVE) { // source block -> LDS
        int loff, cb;
        gl_u8 *g = S.src_item_ptr(t, S.blkx, S.blky, stepX, stepY, loff, cb);
        A4x32 a = ld_chunk_g(g, cb);
        st_chunk_l(S.lds + loff, a, cb);
    }
    __builtin_amdgcn_wave_barrier();
    S.searchType = P.searchType; S.nSearchParam = P.nSearchParam; S.tryMany = 0;
    S.penaltyNew = P.pnew; S.penaltyZero = 0; S.pglobal = 0; S.badcount = 0; S.badrange = 0; S.badSAD = 0; S.LSAD = 0;
    S.dctmode = P.dctmode; S.dctweight16 = 8; S.sumLumaChange = 0; S.srcLuma = 0; // :1167
    S.zeroMVfieldShifted.x = 0; S.zeroMVfieldShifted.y = 0; S.zeroMVfieldShifted.sad = 0;
    S.globalMVPredictor.x = 0; S.globalMVPredictor.y = 0; S.globalMVPredictor.sad = 9999999;
    const int nLambdaLevel = P.lambda / (S.pel * S.pel);
    S.nLambda = S.blky == 0 ? 0 : nLambdaLevel;
    S.nDxMax = (S.pw - S.x0 - S.blkW) << S.logPel; // :1262-1265
    S.nDyMax = (S.ph - S.y0 - S.blkH) << S.logPel;
    S.nDxMin = -(S.x0 << S.logPel);
    S.nDyMin = -(S.y0 << S.logPel);
    // old vectors around the new block's centre (:1268-1321); plane headers walked like fgopUpdate
    const unsigned char *po = J.oldBlob + 8;
    for (int i = R.nLvCount - 1; i >= 1; i--) po += *(const int *)po;
    GL_AS const GVec *ov = (GL_AS const GVec *)(po + 4);
    const int centerX = P.blkX / 2 + stepX * S.blkx, blkxold = (centerX - R.blkX / 2) / R.stepX;
    const int centerY = P.blkY / 2 + stepY * S.blky, blkyold = (centerY - R.blkY / 2) / R.stepY;
    const int deltaX = max(0, centerX - (R.blkX / 2 + R.stepX * blkxold)), deltaY = max(0, centerY - (R.blkY / 2 + R.stepY * blkyold));
    const int x1 = min(R.nBlkX - 1, max(0, blkxold)), x2 = min(R.nBlkX - 1, max(0, blkxold + 1));
    const int y1 = min(R.nBlkY - 1, max(0, blkyold)), y2 = min(R.nBlkY - 1, max(0, blkyold + 1));
    Vec vo;
    if (R.smooth == 1) {
        const Vec v1 = S_t::ld_vec(&ov[x1 + y1 * R.nBlkX]), v2 = S_t::ld_vec(&ov[x2 + y1 * R.nBlkX]), v3 = S_t::ld_vec(&ov[x1 + y2 * R.nBlkX]), v4 = S_t::ld_vec(&ov[x2 + y2 * R.nBlkX]);
        const int ax = v1.x * R.stepX + deltaX * (v2.x - v1.x), ay = v1.y * R.stepX + deltaX * (v2.y - v1.y);
        const long long as = v1.sad * R.stepX + deltaX * (v2.sad - v1.sad);
        const int bx = v3.x * R.stepX + deltaX * (v4.x - v3.x), by = v3.y * R.stepX + deltaX * (v4.y - v3.y);
        const long long bs = v3.sad * R.stepX + deltaX * (v4.sad - v3.sad);
        vo.x = (ax + deltaY * (bx - ax) / R.stepY) / R.stepX;
        vo.y = (ay + deltaY * (by - ay) / R.stepY) / R.stepX;
        vo.sad = (as + deltaY * (bs - as) / R.stepY) / R.stepX;
    } else {
        const bool rx = deltaX * 2 >= R.stepX, ry = deltaY * 2 >= R.stepY;
        vo = S_t::ld_vec(&ov[(rx ? x2 : x1) + (ry ? y2 : y1) * R.nBlkX]);
    }
    vo = uni(vo);
    vo.x = (vo.x << S.logPel) >> R.logPel;
    vo.y = (vo.y << S.logPel) >> R.logPel;
    S.predictor = S.clip_mv(vo);
    S.predictor.sad = vo.sad * (P.blkX * P.blkY) / (R.blkX * R.blkY);
    S.bestMV = S.predictor;
    if (S.dctmode == 7 || S.dctmode == 8 || S.dctmode == 10) S.srcLuma = S.src_luma();
    unsigned aL = 0, aC = 0;
    S.eval_cand(l, 6, S.predictor.x, S.predictor.y, S.predictor.y, aL, aC);
    aL = group_sum(aL, 6); aC = group_sum(aC, 6);
    if (S.dctmode != 0) aL = S.apply_dct(true, l, 6, S.predictor.x, S.predictor.y, aL);
    const long long sad = uni((long long)aL + (S.chroma ? (long long)aC : 0));
    S.bestMV.sad = sad;
    S.nMinCost = sad;
    if (sad > R.thSAD) S.search_block(2);
    if (l == 0) S_t::st_vec(&S.vectors[b], S.bestMV);
}

struct RLaunch { int njobs, nBlk, ldsBytes, ldsRow, ldsHist, histBins; hipStream_t st; const AParams *dP; const RParams *dR; const AJob *dJobs; };
int mvx_recalc_launch(const AParams &P, const RLaunch &L);

struct ALaunch {
    int njobs, ldsBytes, ldsRow, ldsHist, histBins;
    int ldsNeed; // what the kernel really uses (ldsBytes may carry the one-chain-per-SIMD floor)
    int simds;   // SIMDs of the device (4 per CU)
    int cpw;     // chains per workgroup the host ordered the jobs for (1, 4 or 8)
    int wpe;     // chains per SIMD: 2 = the 256-register builds (launches with more chains than SIMDs, geometries that fit)
    int syncEvery; // cpw > 1: workgroup barrier every that many blocks of a row (power of two; a row start always syncs)
    int fast;      // > 0: the lean kernel of the default search (mvx_analyse_fast.h) at that many chains per SIMD
    int flags;     // lean kernel: MVX_FAST_* bits
    hipStream_t st;
    const AParams *dP;
    const AJob *dJobs;
};
template <int BPS_, typename GEO_, int WPE_ = 1, int CPW_ = 1> static int launch_analyse_kernel(const ALaunch &L) {
    const int perChain = CPW_ > 1 ? ((L.ldsNeed + 255) & ~255) : L.ldsBytes;
    const int lds = perChain * CPW_;
    if (lds > 64 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void *)analyse_kernel<BPS_, GEO_, WPE_, CPW_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((analyse_kernel<BPS_, GEO_, WPE_, CPW_>), dim3((L.njobs + CPW_ - 1) / CPW_), dim3(64 * CPW_), lds, L.st, L.dP, L.dJobs,
                       L.njobs, perChain, L.syncEvery, L.ldsRow, L.ldsHist, L.histBins);
    return MVX_OK;
}
// returns MVX_OK after launching, or 1 when this translation unit has no kernel for the geometry
int mvx_analyse_launch_any(const AParams &P, const ALaunch &L);
int mvx_analyse_launch_u8(const AParams &P, const ALaunch &L);
int mvx_analyse_launch_u16(const AParams &P, const ALaunch &L);
int mvx_analyse_launch_fast_u8(const AParams &P, const ALaunch &L);
int mvx_analyse_launch_fast_u16(const AParams &P, const ALaunch &L);

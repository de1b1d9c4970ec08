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
    long long badSAD;
    long long verybigSAD;
    long long pitch[3];
    int blobSize;
    int superHPad, superVPad;
    ALevel lv[MVX_MAX_LEVELS];
};

struct AJob {
    const unsigned char *src[3];
    const unsigned char *ref[3];
    unsigned char *blob;
    int fieldShift, valid;
};

struct mvx_analyse {
    mvx_analysis_data ad;
    AParams P;
    AParams *dP = nullptr;
    AJob *dJobs = nullptr;
    size_t jobsCap = 0;
    int ldsBytes = 0;
    int device = 0;
};

// ------------------------------------------------------------------------------------------------ device

struct Vec { int x, y; long long sad; };

#define WAVE 64
#define BIG64 0x7fffffffffffffffLL

__device__ __forceinline__ int lane_id() { return threadIdx.x; }

__device__ __forceinline__ int bcast_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ long long bcast_ll(long long v, int l) {
    int lo = __builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, l);
    int hi = __builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), l);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// full-wave unsigned min via DPP (row_shr 1,2,4,8 ; row_bcast15 ; row_bcast31), result broadcast from lane 63
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false); v = min(v, t);
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false); v = min(v, t);
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false); v = min(v, t);
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false); v = min(v, t);
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false); v = min(v, t);
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false); v = min(v, t);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int wave_sum_i32(int v) {
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) { return (int)(wave_min_u32((unsigned)v ^ 0x80000000u) ^ 0x80000000u); }
__device__ __forceinline__ int wave_max_i32(int v) { return (int)((~wave_min_u32(~((unsigned)v ^ 0x80000000u))) ^ 0x80000000u); }

// ordered arg-min of signed 64-bit costs: returns the lowest lane holding the minimum, or -1 if every lane has BIG64
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

// unaligned little chunks
struct __attribute__((packed, aligned(1))) U4x32 { unsigned v[4]; };
struct __attribute__((packed, aligned(1))) U2x32 { unsigned v[2]; };
struct __attribute__((packed, aligned(1))) U1x32 { unsigned v; };
struct __attribute__((packed, aligned(1))) U1x16 { unsigned short v; };

struct __attribute__((aligned(16))) A4x32 { unsigned v[4]; };
struct __attribute__((aligned(8))) A2x32 { unsigned v[2]; };
// VECTOR as it sits in the blob: records start at byte 12 of a 16-byte aligned blob -> only 4-byte aligned
struct __attribute__((packed, aligned(4))) GVec { int x, y; long long sad; };

template <int BPS> __device__ __forceinline__ unsigned sad32(unsigned a, unsigned b, unsigned acc) {
    return BPS == 1 ? __builtin_amdgcn_sad_u8(a, b, acc) : __builtin_amdgcn_sad_u16(a, b, acc);
}

// SAD of one chunk of CB bytes (CB in {2,4,8,16}); src from LDS (naturally aligned), ref from global (unaligned)
template <int BPS> __device__ __forceinline__ unsigned sad_chunk(const unsigned char *s, const unsigned char *r, int CB, unsigned acc) {
    if (CB == 16) {
        A4x32 a = *(const A4x32 *)s; U4x32 b = *(const U4x32 *)r;
        acc = sad32<BPS>(a.v[0], b.v[0], acc); acc = sad32<BPS>(a.v[1], b.v[1], acc);
        acc = sad32<BPS>(a.v[2], b.v[2], acc); acc = sad32<BPS>(a.v[3], b.v[3], acc);
    } else if (CB == 8) {
        A2x32 a = *(const A2x32 *)s; U2x32 b = *(const U2x32 *)r;
        acc = sad32<BPS>(a.v[0], b.v[0], acc); acc = sad32<BPS>(a.v[1], b.v[1], acc);
    } else if (CB == 4) {
        unsigned a = *(const unsigned *)s; U1x32 b = *(const U1x32 *)r;
        acc = sad32<BPS>(a, b.v, acc);
    } else {
        unsigned short a = *(const unsigned short *)s; U1x16 b = *(const U1x16 *)r;
        acc = sad32<BPS>(a, b.v, acc);
    }
    return acc;
}

template <int BPS> struct Searcher {
    const AParams &P;
    const AJob &J;
    unsigned char *lds;   // [srcblock | rowbuf | hist]
    int ldsRow;           // byte offset of rowbuf
    int ldsHist;          // byte offset of hist (ints)
    int histBins;

    // level constants
    int level, nBlkX, nBlkY, pel, logPel;
    const unsigned char *srcL[3], *refL[3];
    long long pitch[3], pstride[3];
    int pw, ph, hpad, vpad, cph;
    int lumaRowB, chromaRowB, CBL, CBC, logCL, logCC, TL, TCp; // bytes per row, chunk bytes, log2 chunks/row, item counts
    int cBlkX, cBlkY;
    GVec *vectors; // blob record of this level (after the int header)

    // plane-scan state (uniform)
    int searchType, nSearchParam;
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
    __device__ __forceinline__ const unsigned char *ref_luma(int vx, int vy) const {
        int ax = (x0 << logPel) + vx, ay = (y0 << logPel) + vy;
        int m = pel - 1;
        int idx = (ax & m) | ((ay & m) << logPel);
        return refL[0] + idx * pstride[0] + (long long)(ay >> logPel) * pitch[0] + (long long)(ax >> logPel) * BPS;
    }
    __device__ __forceinline__ const unsigned char *ref_chroma(int p, int vx, int vy) const {
        int xbias = (vx < 0) ? ((1 << P.logxr) - 1) : 0;
        int ybias = (vy < 0) ? ((1 << P.logyr) - 1) : 0;
        int ax = (cx0 << logPel) + ((vx + xbias) >> P.logxr), ay = (cy0 << logPel) + ((vy + ybias) >> P.logyr);
        int m = pel - 1;
        int idx = (ax & m) | ((ay & m) << logPel);
        return refL[p] + idx * pstride[p] + (long long)(ay >> logPel) * pitch[p] + (long long)(ax >> logPel) * BPS;
    }

    // Evaluate N candidates; lane i < N proposes luma vector (vx,vy), chroma is derived from (vx, vyc); ok = in-bounds.
    // Lane i receives its candidate's luma SAD and U+V SAD.  dct=0 only (SAD).
    __device__ void eval(int N, int vx, int vy, int vyc, bool ok, unsigned &outL, unsigned &outC) {
        const int lane = lane_id();
        int logG = 6;
        while (logG > 0 && (64 >> logG) < N) logG--; // largest group size with one pass, else G=1 and several passes
        const int G = 1 << logG, NG = 64 >> logG;
        const int g = lane >> logG, s = lane & (G - 1);
        outL = 0; outC = 0;
        for (int base = 0; base < N; base += NG) {
            const int c = base + g;
            const int cl = c & 63;
            const int cvx = __shfl(vx, cl), cvy = __shfl(vy, cl), cvyc = __shfl(vyc, cl);
            const bool cok = __shfl((int)ok, cl) && c < N;
            unsigned aL = 0, aC = 0;
            if (cok) {
                const unsigned char *rl = ref_luma(cvx, cvy);
#pragma unroll 4
                for (int t = s; t < TL; t += G) {
                    int row = t >> logCL, ch = t & ((1 << logCL) - 1);
                    aL = sad_chunk<BPS>(lds + row * lumaRowB + ch * CBL, rl + (long long)row * pitch[0] + ch * CBL, CBL, aL);
                }
                if (P.chroma) {
                    const unsigned char *ru = ref_chroma(1, cvx, cvyc), *rv = ref_chroma(2, cvx, cvyc);
                    const int uoff = P.blkY * lumaRowB, voff = uoff + cBlkY * chromaRowB;
#pragma unroll 2
                    for (int t = s; t < 2 * TCp; t += G) {
                        int pl = t >= TCp, tt = pl ? t - TCp : t;
                        int row = tt >> logCC, ch = tt & ((1 << logCC) - 1);
                        const unsigned char *r = (pl ? rv : ru) + (long long)row * pitch[1 + pl] + ch * CBC;
                        aC = sad_chunk<BPS>(lds + (pl ? voff : uoff) + row * chromaRowB + ch * CBC, r, CBC, aC);
                    }
                }
            }
            for (int m = 1; m < G; m <<= 1) { aL += __shfl_xor((int)aL, m); aC += __shfl_xor((int)aC, m); }
            const int srcLane = ((lane - base) & (NG - 1)) << logG;
            unsigned rL = (unsigned)__shfl((int)aL, srcLane), rC = (unsigned)__shfl((int)aC, srcLane);
            if (lane >= base && lane < base + NG) { outL = rL; outC = rC; }
        }
    }

    // One round of CheckMV-type candidates (PlaneOfBlocks.cpp:219-261): lane i<N proposes (vx,vy).
    // Equivalent to calling the reference's check sequentially in lane order, because every accepted candidate
    // lowers nMinCost and later ones must beat it strictly: the winner is the first lane with the minimal cost.
    // updateBest=false is pobCheckMVdir (:286-289).  Returns the winning lane or -1.
    __device__ int round(int N, int vx, int vy, bool pnew, bool updateBest) {
        const int lane = lane_id();
        bool ok = lane < N && vector_ok(vx, vy);
        unsigned sl, sc;
        eval(N, vx, vy, vy, ok, sl, sc);
        long long cost = BIG64;
        if (ok) {
            long long c = motion_distortion(vx, vy);
            long long sad = sl;
            c += sad + (pnew ? ((penaltyNew * sad) >> 8) : 0);
            if (P.chroma) { long long suv = sc; c += suv + (pnew ? ((penaltyNew * suv) >> 8) : 0); }
            if (c < nMinCost) cost = c;
        }
        long long mc;
        int w = wave_argmin_ll(cost, &mc);
        if (w < 0) return -1;
        nMinCost = mc;
        long long tot = (long long)sl + (P.chroma ? (long long)sc : 0);
        bestMV.sad = bcast_ll(tot, w);
        if (updateBest) { bestMV.x = bcast_i(vx, w); bestMV.y = bcast_i(vy, w); }
        return w;
    }

    // single candidate helpers for the strictly sequential patterns
    __device__ bool check1(int vx, int vy, bool pnew) { return round(1, vx, vy, pnew, true) >= 0; }

    // ExpandingSearch (PlaneOfBlocks.cpp:636-658)
    __device__ __forceinline__ static int ring_count(int r, int s) {
        int n = 0;
        for (int i = -r + s; i < r; i += s) n++;
        return 4 * n + 4;
    }
    __device__ void expanding(int r, int s, int mvx, int mvy) {
        const int total = ring_count(r, s);
        const int n = (total - 4) / 4;
        for (int base = 0; base < total; base += 64) {
            int k = base + lane_id();
            int dx = 0, dy = 0;
            if (k < total) {
                int kk = k;
                if (kk < 2 * n) { dx = -r + s + (kk >> 1) * s; dy = (kk & 1) ? r : -r; }
                else if ((kk -= 2 * n) < 2 * n) { dy = -r + s + (kk >> 1) * s; dx = (kk & 1) ? r : -r; }
                else { kk -= 2 * n; dx = (kk & 2) ? r : -r; dy = (kk & 1) ? r : -r; }
            }
            int cnt = min(64, total - base);
            round(cnt, mvx + dx, mvy + dy, true, true);
        }
    }

    __device__ void nstep(int stp) { // :467-485
        for (int length = stp; length > 0; length--) {
            const int dx = bestMV.x, dy = bestMV.y;
            const int l = lane_id();
            // order: (+,+) (+,0) (+,-) (0,-) (0,+) (-,+) (-,0) (-,-)
            const int ox[8] = { 1, 1, 1, 0, 0, -1, -1, -1 }, oy[8] = { 1, 0, -1, -1, 1, 1, 0, -1 };
            int i = l & 7;
            round(8, dx + ox[i] * length, dy + oy[i] * length, true, true);
        }
    }

    __device__ void onetime(int length) { // :489-527
        int dx = bestMV.x, dy = bestMV.y;
        int direction = 0;
        if (check1(dx - length, dy, true)) direction = 2;
        if (check1(dx + length, dy, true)) direction = 1;
        if (direction == 1) {
            while (direction) { direction = 0; dx += length; if (check1(dx + length, dy, true)) direction = 1; }
        } else if (direction == 2) {
            while (direction) { direction = 0; dx -= length; if (check1(dx - length, dy, true)) direction = 1; }
        }
        if (check1(dx, dy - length, true)) direction = 2;
        if (check1(dx, dy + length, true)) direction = 1;
        if (direction == 1) {
            while (direction) { direction = 0; dy += length; if (check1(dx, dy + length, true)) direction = 1; }
        } else if (direction == 2) {
            while (direction) { direction = 0; dy -= length; if (check1(dx, dy - length, true)) direction = 1; }
        }
    }

    __device__ void diamond(int length) { // :531-632
        enum { Right = 1, Left = 2, Down = 4, Up = 8 };
        int dx, dy, direction = 15, last;
#define CK2(X, Y, V) do { if (check1((X), (Y), true)) direction = (V); } while (0)
        while (direction > 0) {
            dx = bestMV.x; dy = bestMV.y; last = direction; direction = 0;
            if (last & Right) CK2(dx + length, dy, Right);
            if (last & Left) CK2(dx - length, dy, Left);
            if (last & Down) CK2(dx, dy + length, Down);
            if (last & Up) CK2(dx, dy - length, Up);
            if (direction) {
                last = direction; dx = bestMV.x; dy = bestMV.y;
                if (last & (Right + Left)) { CK2(dx, dy + length, Down); CK2(dx, dy - length, Up); }
                else { CK2(dx + length, dy, Right); CK2(dx - length, dy, Left); }
            } else {
                switch (last) {
                case Right: CK2(dx + length, dy + length, Right + Down); CK2(dx + length, dy - length, Right + Up); break;
                case Left: CK2(dx - length, dy + length, Left + Down); CK2(dx - length, dy - length, Left + Up); break;
                case Down: CK2(dx + length, dy + length, Right + Down); CK2(dx - length, dy + length, Left + Down); break;
                case Up: CK2(dx + length, dy - length, Right + Up); CK2(dx - length, dy - length, Left + Up); break;
                case Right + Down: CK2(dx + length, dy + length, Right + Down); CK2(dx - length, dy + length, Left + Down); CK2(dx + length, dy - length, Right + Up); break;
                case Left + Down: CK2(dx + length, dy + length, Right + Down); CK2(dx - length, dy + length, Left + Down); CK2(dx - length, dy - length, Left + Up); break;
                case Right + Up: CK2(dx + length, dy + length, Right + Down); CK2(dx - length, dy - length, Left + Up); CK2(dx + length, dy - length, Right + Up); break;
                case Left + Up: CK2(dx - length, dy - length, Left + Up); CK2(dx - length, dy + length, Left + Down); CK2(dx + length, dy - length, Right + Up); break;
                default:
                    CK2(dx + length, dy + length, Right + Down); CK2(dx - length, dy + length, Left + Down);
                    CK2(dx + length, dy - length, Right + Up); CK2(dx - length, dy - length, Left + Up); break;
                }
            }
        }
#undef CK2
    }

    __device__ void hex2search(int i_me_range) { // :667-724
        // hex2[dir+1]: { -1,-2 }, { -2,0 }, { -1,2 }, { 1,2 }, { 2,0 }, { 1,-2 }, { -1,-2 }, { -2,0 }
        const int hx[8] = { -1, -2, -1, 1, 2, 1, -1, -2 }, hy[8] = { -2, 0, 2, 2, 0, -2, -2, 0 };
        const int mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
        int dir = -2, bmx = bestMV.x, bmy = bestMV.y;
        const int l = lane_id();
        if (i_me_range > 1) {
            // candidates in order dir 0..5 = hex2[1..6]
            int i = l < 6 ? l : 0;
            int w = round(6, bmx + hx[i + 1], bmy + hy[i + 1], true, false);
            if (w >= 0) dir = w;
            if (dir != -2) {
                bmx += hx[dir + 1]; bmy += hy[dir + 1];
                for (int it = 1; it < i_me_range / 2 && vector_ok(bmx, bmy); it++) {
                    const int odir = mod6m1[dir + 1];
                    dir = -2;
                    int k = l < 3 ? l : 0;
                    int w2 = round(3, bmx + hx[odir + k], bmy + hy[odir + k], true, false);
                    if (w2 >= 0) dir = odir - 1 + w2;
                    if (dir == -2) break;
                    bmx += hx[dir + 1]; bmy += hy[dir + 1];
                }
            }
            bestMV.x = bmx; bestMV.y = bmy;
        }
        expanding(1, 1, bmx, bmy);
    }

    __device__ void umh(int i_me_range, int omx, int omy) { // :743-769 (+ CrossSearch :728-739)
        const int l = lane_id();
        // cross: for i=1,3,.. < range: (-i,0) (+i,0); then for j: (0,-j) (0,+j)
        int nh = 0;
        for (int i = 1; i < i_me_range; i += 2) nh++;
        const int crossN = 4 * nh;
        for (int base = 0; base < crossN; base += 64) {
            int k = base + l, dx = 0, dy = 0;
            if (k < 2 * nh) { int i = 1 + 2 * (k >> 1); dx = (k & 1) ? i : -i; }
            else if (k < crossN) { int kk = k - 2 * nh; int j = 1 + 2 * (kk >> 1); dy = (kk & 1) ? j : -j; }
            round(min(64, crossN - base), omx + dx, omy + dy, true, true);
        }
        const int h4x[16] = { -4, -4, -4, -4, -4, 4, 4, 4, 4, 4, 2, 0, -2, -2, 0, 2 };
        const int h4y[16] = { 2, 1, 0, -1, -2, -2, -1, 0, 1, 2, 3, 4, 3, -3, -4, -3 };
        int nrings = 0;
        { int i = 1; do { nrings++; } while (++i <= i_me_range / 4); }
        const int hexN = 16 * nrings;
        for (int base = 0; base < hexN; base += 64) {
            int k = base + l;
            int i = 1 + (k >> 4), j = k & 15;
            round(min(64, hexN - base), omx + h4x[j] * i, omy + h4y[j] * i, true, true);
        }
        hex2search(i_me_range);
    }

    __device__ void refine() { // :773-816
        const int st = searchType;
        if (st == SearchOnetime) for (int i = nSearchParam; i > 0; i /= 2) onetime(i);
        if (st == SearchNstep) nstep(nSearchParam);
        if (st == SearchLogarithmic) for (int i = nSearchParam; i > 0; i /= 2) diamond(i);
        if (st == SearchExhaustive) {
            const int mvx = bestMV.x, mvy = bestMV.y;
            // rings 1..nSearchParam around a fixed centre: all candidates are known up front -> one batch
            int total = 0;
            for (int i = 1; i <= nSearchParam; i++) total += 8 * i;
            if (total <= 64) {
                int k = lane_id(), r = 1, dx = 0, dy = 0;
                while (r < nSearchParam && k >= 8 * r) { k -= 8 * r; r++; }
                if (lane_id() < total) {
                    const int n = 2 * r - 1;
                    int kk = k;
                    if (kk < 2 * n) { dx = -r + 1 + (kk >> 1); dy = (kk & 1) ? r : -r; }
                    else if ((kk -= 2 * n) < 2 * n) { dy = -r + 1 + (kk >> 1); dx = (kk & 1) ? r : -r; }
                    else { kk -= 2 * n; dx = (kk & 2) ? r : -r; dy = (kk & 1) ? r : -r; }
                }
                round(total, mvx + dx, mvy + dy, true, true);
            } else
                for (int i = 1; i <= nSearchParam; i++) expanding(i, 1, mvx, mvy);
        }
        if (st == SearchHex2) hex2search(nSearchParam);
        if (st == SearchUMH) umh(nSearchParam, bestMV.x, bestMV.y);
        if (st == SearchHorizontal || st == SearchVertical) {
            const int mvx = bestMV.x, mvy = bestMV.y;
            const int total = 2 * nSearchParam;
            for (int base = 0; base < total; base += 64) {
                int k = base + lane_id();
                int i = 1 + (k >> 1), sg = (k & 1) ? 1 : -1;
                int dx = st == SearchHorizontal ? sg * i : 0, dy = st == SearchVertical ? sg * i : 0;
                round(min(64, total - base), mvx + dx, mvy + dy, true, true);
            }
        }
    }

    // PlaneOfBlocks.cpp:419-463
    __device__ void fetch_predictors(Vec prev, bool havePrev, Vec up, Vec ahead, bool haveAhead) {
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
        double scale = (double)LSAD / (double)(LSAD + (predictor.sad >> 1));
        nLambda = (long long)((double)nLambda * scale * scale);
    }

    // PlaneOfBlocks.cpp:819-968
    __device__ void pseudo_epz() {
        const int l = lane_id();
        // round A: zero, global, predictor, predictors[0..3] -- costs are independent of each other
        globalMVPredictor = clip_mv(globalMVPredictor); // cumulative clip (:859)
        int vx = 0, vy = 0, vyc = 0;
        if (l == 0) { vx = 0; vy = zeroMVfieldShifted.y; vyc = 0; }
        else if (l == 1) { vx = globalMVPredictor.x; vy = vyc = globalMVPredictor.y; }
        else if (l == 2) { vx = predictor.x; vy = vyc = predictor.y; }
        else if (l < 7) {
            int i = l - 3;
            Vec pv = i == 0 ? predictors[0] : i == 1 ? predictors[1] : i == 2 ? predictors[2] : predictors[3];
            vx = pv.x; vy = vyc = pv.y;
        }
        unsigned sl, sc;
        eval(7, vx, vy, vyc, l < 7, sl, sc);
        const long long tot = (long long)sl + (P.chroma ? (long long)sc : 0);
        long long cost;
        if (l == 0) cost = tot + ((penaltyZero * tot) >> 8);
        else if (l == 1) cost = tot + ((pglobal * tot) >> 8);
        else if (l == 2) cost = tot;
        else cost = (long long)motion_distortion(vx, vy) + tot; // CheckMV0: no new-vector penalty, predictors are pre-clipped
        if (l >= 7) cost = BIG64;

        if (!tryMany) {
            long long mc;
            int w = wave_argmin_ll(cost, &mc); // lane 0 always participates -> w >= 0
            nMinCost = mc;
            bestMV.x = bcast_i(vx, w); bestMV.y = bcast_i(vy, w); bestMV.sad = bcast_ll(tot, w);
            refine();
        } else {
            Vec bestMany[7]; long long costMany[7];
            for (int i = 0; i < 7; i++) {
                const long long ci = bcast_ll(cost, i);
                const long long ti = bcast_ll(tot, i);
                const int cx = bcast_i(vx, i), cy = bcast_i(vy, i);
                if (i == 0) { bestMV.x = cx; bestMV.y = cy; bestMV.sad = ti; nMinCost = ci; }
                else if (i < 3) { bestMV.x = cx; bestMV.y = cy; bestMV.sad = ti; nMinCost = ci; } // forced (:872,:896)
                else { // :913-915
                    nMinCost = P.verybigSAD + 1;
                    if (ci < nMinCost) { bestMV.x = cx; bestMV.y = cy; bestMV.sad = ti; nMinCost = ci; }
                }
                refine();
                bestMany[i] = bestMV; costMany[i] = nMinCost;
            }
            nMinCost = P.verybigSAD + 1;
            for (int i = 0; i < 7; i++)
                if (costMany[i] < nMinCost) { bestMV = bestMany[i]; nMinCost = costMany[i]; }
        }

        const long long foundSAD = bestMV.sad;
        if (blkIdx > 1 && foundSAD > (badSAD + badSAD * badcount / 16)) { // :942
            badcount++;
            if (badrange > 0)
                umh(badrange * pel, 0, 0);
            else if (badrange < 0) {
                for (int i = 1; i < -badrange * pel; i += pel) {
                    expanding(i, pel, 0, 0);
                    if (bestMV.sad < foundSAD / 4) break;
                }
            }
            const int mvx = bestMV.x, mvy = bestMV.y;
            for (int i = 1; i < pel; i++) expanding(i, 1, mvx, mvy);
        }
    }

    // stage the source block (luma + chroma) into LDS: PlaneOfBlocks.cpp:1058-1079
    __device__ void stage_src() {
        const int l = lane_id();
        for (int t = l; t < TL; t += WAVE) {
            int row = t >> logCL, ch = t & ((1 << logCL) - 1);
            const unsigned char *g = srcL[0] + (long long)(y0 + row) * pitch[0] + (long long)x0 * BPS + ch * CBL;
            unsigned char *d = lds + row * lumaRowB + ch * CBL;
            if (CBL == 16) { U4x32 t4 = *(const U4x32 *)g; A4x32 a4; a4.v[0] = t4.v[0]; a4.v[1] = t4.v[1]; a4.v[2] = t4.v[2]; a4.v[3] = t4.v[3]; *(A4x32 *)d = a4; }
            else if (CBL == 8) { U2x32 t2 = *(const U2x32 *)g; A2x32 a2; a2.v[0] = t2.v[0]; a2.v[1] = t2.v[1]; *(A2x32 *)d = a2; }
            else if (CBL == 4) *(unsigned *)d = ((const U1x32 *)g)->v; else *(unsigned short *)d = ((const U1x16 *)g)->v;
        }
        if (P.chroma) {
            const int uoff = P.blkY * lumaRowB, voff = uoff + cBlkY * chromaRowB;
            for (int t = l; t < 2 * TCp; t += WAVE) {
                int pl = t >= TCp, tt = pl ? t - TCp : t;
                int row = tt >> logCC, ch = tt & ((1 << logCC) - 1);
                const unsigned char *g = srcL[1 + pl] + (long long)(cy0 + row) * pitch[1 + pl] + (long long)cx0 * BPS + ch * CBC;
                unsigned char *d = lds + (pl ? voff : uoff) + row * chromaRowB + ch * CBC;
                if (CBC == 16) { U4x32 t4 = *(const U4x32 *)g; A4x32 a4; a4.v[0] = t4.v[0]; a4.v[1] = t4.v[1]; a4.v[2] = t4.v[2]; a4.v[3] = t4.v[3]; *(A4x32 *)d = a4; }
                else if (CBC == 8) { U2x32 t2 = *(const U2x32 *)g; A2x32 a2; a2.v[0] = t2.v[0]; a2.v[1] = t2.v[1]; *(A2x32 *)d = a2; }
                else if (CBC == 4) *(unsigned *)d = ((const U1x32 *)g)->v; else *(unsigned short *)d = ((const U1x16 *)g)->v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
    }

    __device__ static Vec ld_vec(const GVec *p) { Vec v; v.x = p->x; v.y = p->y; v.sad = p->sad; return v; }
    __device__ static void st_vec(GVec *p, const Vec &v) { GVec g; g.x = v.x; g.y = v.y; g.sad = v.sad; *p = g; }

    // GroupOfPlanes.c:69-125 + PlaneOfBlocks.cpp:971-1131 for one level
    __device__ void search_level(int lvl, Vec *globalMV, const GVec *coarse, int coarseBlkX, int coarseBlkY, int coarseLogPel) {
        const int l = lane_id();
        const ALevel &L = P.lv[lvl];
        level = lvl; nBlkX = L.nBlkX; nBlkY = L.nBlkY; pel = L.pel; logPel = L.logPel;
        pw = L.pw; ph = L.ph; hpad = L.hpad; vpad = L.vpad; cph = L.cph;
        for (int p = 0; p < 3; p++) {
            srcL[p] = J.src[p] ? J.src[p] + L.off[p] : nullptr;
            refL[p] = J.ref[p] ? J.ref[p] + L.off[p] : nullptr;
            pitch[p] = P.pitch[p]; pstride[p] = L.pstride[p];
        }
        cBlkX = P.blkX / P.xr; cBlkY = P.blkY / P.yr;
        lumaRowB = P.blkX * BPS; chromaRowB = cBlkX * BPS;
        CBL = min(16, lumaRowB); CBC = min(16, chromaRowB);
        logCL = mvx_ilog2_dev(lumaRowB / CBL); logCC = mvx_ilog2_dev(chromaRowB / CBC);
        TL = P.blkY << logCL; TCp = cBlkY << logCC;
        unsigned char *rec = J.blob + L.blobOff;
        vectors = (GVec *)(rec + 4);
        const int nBlk = nBlkX * nBlkY;
        if (l == 0) *(int *)rec = 4 + nBlk * 16; // pobWriteHeaderToArray :413-416

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

        Vec *rowbuf = (Vec *)(lds + ldsRow);
        const int stepX = P.blkX - P.ovX, stepY = P.blkY - P.ovY;
        const int hps = P.superHPad >> lvl, vps = P.superVPad >> lvl; // :1091-1092 (luma padding of the SOURCE frame's plane = super pad)
        Vec prev; prev.x = 0; prev.y = 0; prev.sad = 0;

        for (blky = 0; blky < nBlkY; blky++) {
            blkScanDir = (blky % 2 == 0 || P.meander == 0) ? 1 : -1;
            const int blkxStart = blkScanDir == 1 ? 0 : nBlkX - 1;
            y0 = vpad + stepY * blky;
            cy0 = L.cvpad + (stepY >> P.logyr) * blky;
            for (int iblkx = 0; iblkx < nBlkX; iblkx++) {
                blkx = blkxStart + iblkx * blkScanDir;
                blkIdx = blky * nBlkX + blkx;
                x0 = hpad + stepX * blkx;
                cx0 = L.chpad + (stepX >> P.logxr) * blkx; // :1050-1051,1116-1118

                nDxMax = (pw - x0 - P.blkX - hpad + hps) << logPel;
                nDyMax = (ph - y0 - P.blkY - vpad + vps) << logPel;
                nDxMin = -((x0 - hpad + hps) << logPel);
                nDyMin = -((y0 - vpad + vps) << logPel);

                // hierarchical predictor (vectors[blkIdx] before it is overwritten) and the not-yet-searched neighbour
                const bool aheadCol = (blkScanDir == 1 && blkx < nBlkX - 1) || (blkScanDir == -1 && blkx > 0);
                const bool useBelow = (blky < nBlkY - 1) && aheadCol;
                const bool useUpAhead = !useBelow && (blky > 0) && aheadCol;
                Vec self = ld_vec(&vectors[blkIdx]);
                Vec ahead; ahead.x = 0; ahead.y = 0; ahead.sad = 0;
                if (useBelow) ahead = ld_vec(&vectors[blkIdx + nBlkX + blkScanDir]);
                else if (useUpAhead) ahead = rowbuf[blkx + blkScanDir];
                Vec up; up.x = 0; up.y = 0; up.sad = 0;
                if (blky > 0) up = rowbuf[blkx];

                stage_src();

                nLambda = blky == 0 ? 0 : nLambdaLevel;
                predictor = clip_mv(self);
                const bool havePrev = (blkScanDir == 1 && blkx > 0) || (blkScanDir == -1 && blkx < nBlkX - 1);
                fetch_predictors(prev, havePrev, up, ahead, useBelow || useUpAhead);

                pseudo_epz();

                // results: vectors[blkIdx] (:967) == blob row (:1106); keep the row in LDS for the next row's predictors
                __builtin_amdgcn_s_barrier();
                if (l == 0) { st_vec(&vectors[blkIdx], bestMV); rowbuf[blkx] = bestMV; }
                prev = bestMV;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    __device__ static int mvx_ilog2_dev(int i) { int r = 0; while (i > 1) { i >>= 1; r++; } return r; }

    // pobEstimateGlobalMVDoubled, PlaneOfBlocks.cpp:1559-1636 (mode via LDS histogram windows; first maximum wins)
    __device__ void estimate_global(const GVec *v, int nBlk, int freqSizeHalf, Vec *g) {
        const int l = lane_id();
        int *hist = (int *)(lds + ldsHist);
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
                __builtin_amdgcn_s_barrier();
                for (int i = l; i < nBlk; i += WAVE) {
                    int val = c ? v[i].y : v[i].x;
                    int ind = freqSizeHalf + val;
                    if (ind >= 0 && ind < 2 * freqSizeHalf && val >= wbase && val < wbase + histBins) atomicAdd(&hist[val - wbase], 1);
                }
                __builtin_amdgcn_s_barrier();
                // first maximum in ascending order
                int lc = -1, lv = 0;
                for (int i = l; i < histBins && wbase + i <= hi; i += WAVE) {
                    int cnt = hist[i];
                    if (cnt > lc) { lc = cnt; lv = i; }
                }
                // wave: max count, then lowest index
                int mcnt = wave_max_i32(lc);
                int cand = (lc == mcnt) ? lv : 0x7fffffff;
                int mv = wave_min_i32(cand);
                if (mcnt > bestCount) { bestCount = mcnt; bestVal = wbase + mv; }
                __builtin_amdgcn_s_barrier();
            }
            med[c] = bestVal;
        }
        int sx = 0, sy = 0, n = 0;
        for (int i = l; i < nBlk; i += WAVE) {
            int vx = v[i].x, vy = v[i].y;
            if (abs(vx - med[0]) < 6 && abs(vy - med[1]) < 6) { sx += vx; sy += vy; n++; }
        }
        sx = wave_sum_i32(sx); sy = wave_sum_i32(sy); n = wave_sum_i32(n);
        if (n > 0) { g->x = 2 * sx / n; g->y = 2 * sy / n; }
        else { g->x = 2 * med[0]; g->y = 2 * med[1]; }
    }
};

template <int BPS>
__global__ __launch_bounds__(64) void analyse_kernel(const AParams *Pp, const AJob *jobs, int ldsRow, int ldsHist, int histBins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AParams &P = *Pp;
    const AJob &J = jobs[blockIdx.x];
    const int l = threadIdx.x;
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
    Searcher<BPS> S(P, J);
    S.lds = smem; S.ldsRow = ldsRow; S.ldsHist = ldsHist; S.histBins = histBins;
    Vec globalMV; globalMV.x = 0; globalMV.y = 0; globalMV.sad = -1; // zeroMV, MVAnalysisData.h:79
    const GVec *coarse = nullptr;
    int cbx = 0, cby = 0, clp = 0;
    for (int lvl = P.nLevels - 1; lvl >= 0; lvl--) {
        if (coarse && P.global) S.estimate_global(coarse, cbx * cby, 8192 * P.lv[lvl + 1].pel, &globalMV);
        S.search_level(lvl, &globalMV, coarse, cbx, cby, clp);
        coarse = S.vectors; cbx = P.lv[lvl].nBlkX; cby = P.lv[lvl].nBlkY; clp = P.lv[lvl].logPel;
    }
}

// ------------------------------------------------------------------------------------------------ host

static thread_local char g_err[512];
void mvx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" __attribute__((visibility("default"))) const char *mvx_last_error(void) { return g_err; }

#define AFAIL(...) do { snprintf(err, MVX_ERRLEN, __VA_ARGS__); mvx_set_error("%s", err); return MVX_E_ARG; } while (0)
static int A(int v, int d) { return v == MVX_UNSET ? d : v; }

// MVAnalyse.c:267-635 mvanalyseCreate
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
    if (P.dctmode != 0) AFAIL("Analyse: dct modes other than 0 are not implemented on the GPU path yet.");
    if (divide < 0 || divide > 2) AFAIL("Analyse: divide must be between 0 and 2 (inclusive).");
    if (divide) AFAIL("Analyse: divide is not implemented on the GPU path yet.");
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
    for (int p = 0; p < 3; p++) P.pitch[p] = p < si.num_planes ? super_pitch[p] : 0;
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
    P.blobSize = blobOff;

    mvx_analyse *h = new mvx_analyse();
    h->ad = ad;
    h->P = P;
    *out = h; // device state is created on first use so that argument validation works without a GPU
    return MVX_OK;
}

extern "C" __attribute__((visibility("default"))) void mvx_analyse_destroy(mvx_analyse *a) {
    if (!a) return;
    if (a->dP) (void)hipFree(a->dP);
    if (a->dJobs) (void)hipFree(a->dJobs);
    delete a;
}
extern "C" __attribute__((visibility("default"))) void mvx_analyse_get_data(const mvx_analyse *a, mvx_analysis_data *out) { *out = a->ad; }
extern "C" __attribute__((visibility("default"))) int mvx_analyse_blob_size(const mvx_analyse *a) { return a->P.blobSize; }

extern "C" __attribute__((visibility("default"))) int mvx_analyse_frames(mvx_analyse *a, int njobs, const mvx_analyse_job *jobs, void *stream) {
    if (njobs <= 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    const AParams &P = a->P;
    if (!a->dP) {
        HIP_CHECK(hipGetDevice(&a->device));
        HIP_CHECK(hipMalloc((void **)&a->dP, sizeof(AParams)));
        HIP_CHECK(hipMemcpy(a->dP, &P, sizeof(AParams), hipMemcpyHostToDevice));
    }
    if ((size_t)njobs > a->jobsCap) {
        if (a->dJobs) (void)hipFree(a->dJobs);
        a->jobsCap = (size_t)njobs * 2;
        HIP_CHECK(hipMalloc((void **)&a->dJobs, a->jobsCap * sizeof(AJob)));
    }
    std::vector<AJob> hj(njobs);
    for (int i = 0; i < njobs; i++) {
        for (int p = 0; p < 3; p++) { hj[i].src[p] = (const unsigned char *)jobs[i].src[p]; hj[i].ref[p] = (const unsigned char *)jobs[i].ref[p]; }
        hj[i].blob = (unsigned char *)jobs[i].blob;
        hj[i].fieldShift = jobs[i].field_shift;
        hj[i].valid = jobs[i].ref[0] != nullptr;
        if (((uintptr_t)jobs[i].blob) & 15) { mvx_set_error("mvx_analyse_frames: blob must be 16-byte aligned"); return MVX_E_ARG; }
    }
    HIP_CHECK(hipMemcpyAsync(a->dJobs, hj.data(), sizeof(AJob) * njobs, hipMemcpyHostToDevice, st));
    // LDS: [source block (Y,U,V) | previous-row vectors | histogram]
    int srcBytes = P.blkX * P.blkY * P.bps;
    if (P.chroma) srcBytes += 2 * (P.blkX / P.xr) * (P.blkY / P.yr) * P.bps;
    int ldsRow = (srcBytes + 15) & ~15;
    int maxBlkX = 0;
    for (int i = 0; i < P.nLevels; i++) if (P.lv[i].nBlkX > maxBlkX) maxBlkX = P.lv[i].nBlkX;
    int ldsHist = ldsRow + maxBlkX * 16;
    const int histBins = 2048;
    int ldsBytes = ldsHist + histBins * 4;
    if (ldsBytes > 160 * 1024) { mvx_set_error("mvx_analyse_frames: frame too wide for the LDS row buffer"); return MVX_E_ARG; }
    if (P.bps == 1) {
        if (ldsBytes > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)analyse_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
        hipLaunchKernelGGL(analyse_kernel<1>, dim3(njobs), dim3(64), ldsBytes, st, a->dP, a->dJobs, ldsRow, ldsHist, histBins);
    } else {
        if (ldsBytes > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)analyse_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
        hipLaunchKernelGGL(analyse_kernel<2>, dim3(njobs), dim3(64), ldsBytes, st, a->dP, a->dJobs, ldsRow, ldsHist, histBins);
    }
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}

// mvx_degrain.hip -- mv.Degrain1..6 and mv.Compensate on gfx950.
//
// The reference walks blocks, blends each block into a temp (Degrain_C, MVDegrains.h:30-53), scatters it times a
// raised-cosine window into a 16/32-bit accumulator (overlaps_c, Overlap.cpp:143-158) and finally normalises
// (ToPixels, Overlap.cpp:335-356).  The accumulator never saturates, so the sum is order-free and the GPU form is
// a GATHER: one thread per output sample visits the <=4 blocks covering it, recomputes that block's blended sample
// and accumulates it times the window tap.  No accumulator plane, no atomics, one pass over HBM:
//   plan kernel   : per (frame, block) -> per-reference weights (fp64 exactly as MVDegrains.h:184-223) and the byte
//                   offset of the motion-compensated block inside the reference super frame (MVDegrains.h:192-206)
//   gather kernel : per output sample  -> blend + window + normalise + uncovered strips + LimitChanges.
#include "mvx_common.h"

#define MOTION_USE_CHROMA_MOTION 8

struct __attribute__((packed, aligned(4))) GVecD { int x, y; long long sad; };

// level-0 vectors of a MVTools_vectors blob: skip size + validity, then every coarser plane by ITS OWN size header -- the
// reference's reader does exactly this, which is what makes clips produced with divide (an extra array of half-size
// blocks after the finest estimated plane, whose geometry the level formula does not describe) readable
// pointers that come out of job tables are generic ("flat") to the compiler: loads through them are slower and each is waited for on its own
#define DG_GL __attribute__((address_space(1)))
__device__ __forceinline__ DG_GL const unsigned char *dg_gl(const void *p) { return (DG_GL const unsigned char *)(unsigned long long)p; }
__device__ __forceinline__ const GVecD *mvx_level0(const unsigned char *blob, int nLvCount) {
    const unsigned char *p = blob + 8;
    for (int i = nLvCount - 1; i >= 1; i--) p += *(const int *)p;
    return (const GVecD *)(p + 4);
}

// ------------------------------------------------------------------------------------------------ host helpers

// Overlap.cpp:40-125 overInit.  M_PI is the double constant; the cosf argument is formed in double.
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

void mvx_over_windows(int16_t *win9, int nx, int ny, int ox, int oy) {
    std::vector<float> fx(nx * 3), fy(ny * 3);
    win1d(&fx[0], &fx[nx], &fx[2 * nx], nx, ox);
    win1d(&fy[0], &fy[ny], &fy[2 * ny], ny, oy);
    const float *x[3] = { &fx[nx], &fx[0], &fx[2 * nx] }; // first, middle, last
    const float *y[3] = { &fy[ny], &fy[0], &fy[2 * ny] };
    for (int wy = 0; wy < 3; wy++)
        for (int wx = 0; wx < 3; wx++) {
            int16_t *w = win9 + nx * ny * (wy * 3 + wx);
            for (int j = 0; j < ny; j++)
                for (int i = 0; i < nx; i++) w[j * nx + i] = (int16_t)(int)(y[wy][j] * x[wx][i] * 2048 + 0.5f);
        }
}

extern "C" __attribute__((visibility("default"))) int mvx_vectors_size(const mvx_analysis_data *ad) { // Fakery.c:110-121, GroupOfPlanes.c:127-148
    const int nWidth_B = (ad->nBlkSizeX - ad->nOverlapX) * ad->nBlkX + ad->nOverlapX;
    const int nHeight_B = (ad->nBlkSizeY - ad->nOverlapY) * ad->nBlkY + ad->nOverlapY;
    int size = 8;
    for (int i = ad->nLvCount - 1; i >= 0; i--) {
        const int bx = ((nWidth_B >> i) - ad->nOverlapX) / (ad->nBlkSizeX - ad->nOverlapX);
        const int by = ((nHeight_B >> i) - ad->nOverlapY) / (ad->nBlkSizeY - ad->nOverlapY);
        size += 4 + bx * by * 16;
    }
    return size;
}

extern "C" __attribute__((visibility("default"))) void mvx_scale_thscd(int64_t *thscd1, int32_t *thscd2, const mvx_analysis_data *ad) { // MVAnalysisData.c:7-31
    *thscd1 = *thscd1 * (ad->nBlkSizeX * ad->nBlkSizeY) / (8 * 8);
    if (ad->nMotionFlags & MOTION_USE_CHROMA_MOTION) *thscd1 += *thscd1 / (ad->xRatioUV * ad->yRatioUV) * 2;
    const int pixelMax = (1 << ad->bitsPerSample) - 1;
    *thscd1 = (int64_t)((double)*thscd1 * pixelMax / 255.0 + 0.5);
    *thscd2 = *thscd2 * ad->nBlkX * ad->nBlkY / 256;
}

// ------------------------------------------------------------------------------------------------ shared device structs

struct PlaneG { // one plane of the clip / of level 0 of the super frame
    int W, H, WB, HB;        // frame dims, block-covered dims
    int blkW, blkH, ovX, ovY, stepX, stepY;
    int hpadPel, vpadPel;    // super padding * pel, in sub-pel units
    int subX, subY;          // log2 subsampling of this plane relative to luma
    long long srcPitch, supPitch, dstPitch, supPlaneStride; // bytes
    int thIdx;               // 0 luma threshold, 1 chroma threshold
    int process;
    int limit;
    long long shadow;        // 16-bit luma: byte distance to the copy of the super plane shifted left by one sample (mvx_degrain_set_ref_shadow), 0 = none
};

struct DGParams {
    int nRefs, nBlkX, nBlkY, nBlk, pel, logPel, bits, bps, nplanes, overlap;
    int nLvCount;            // levels in a blob; the level-0 record is found by walking the per-plane size headers like fgopUpdate (Fakery.c:112-123)
    long long thSAD[2];
    long long thscd1; int thscd2;
    PlaneG pl[3];
    const int16_t *win[3];
    // compensate only
    long long cthSAD; int time256, scBehavior;
};

struct DGJob {
    const unsigned char *src[3];
    const unsigned char *refs[12][3];
    const unsigned char *blobs[12];
    unsigned char *dst[3];
    int fieldShift;          // compensate only (MVCompensate.c:188-225)
};

// plan record: one per (frame, plane class luma/chroma, block), sized for the filter's 2 * radius references (Degrain3: 40 bytes; a fixed
// 12-reference record was 76 -- at 4K16 the plan of a 512-frame batch was 10 GB written and read back, more than the vector blobs it digests)
template <int NR> struct PlanRecT {
    unsigned off[NR];  // byte offset of the compensated block's first sample inside the reference super plane
    short w[NR];
    short wsrc;
    short pad;
};
static size_t plan_rec_bytes(int nrefs) { return (size_t)6 * nrefs + 4; }
static_assert(sizeof(PlanRecT<6>) == 40 && sizeof(PlanRecT<12>) == 76 && sizeof(PlanRecT<2>) == 16, "plan record layout");

// ------------------------------------------------------------------------------------------------ kernels

// Fakery.c:52-58,103-107,144-146: usable = validity==1 && !(count(sad > thscd1) > thscd2)
__global__ __launch_bounds__(256) void usable_kernel(const DGParams *Pp, const DGJob *jobs, int *usable, int single) {
    const DGParams &P = *Pp;
    const int f = blockIdx.y, r = blockIdx.x;
    const unsigned char *blob = single ? jobs[f].blobs[0] : jobs[f].blobs[r];
    const bool haveRef = single ? true : jobs[f].refs[r][0] != nullptr;
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int c = 0;
    const GVecD *v = mvx_level0(blob, P.nLvCount);
    for (int i = threadIdx.x; i < P.nBlk; i += 256) c += v[i].sad > P.thscd1 ? 1 : 0;
    atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) {
        int validity = ((const int *)blob)[1];
        usable[f * 12 + r] = haveRef && validity == 1 && !(cnt > P.thscd2);
    }
}

// MVDegrains.h:184-189
__device__ __forceinline__ int degrain_weight(long long thSAD, long long blockSAD) {
    if (blockSAD >= thSAD) return 0;
    return (int)((double)((thSAD - blockSAD) * (thSAD + blockSAD) * 256) / (double)(thSAD * thSAD + blockSAD * blockSAD));
}

// MVFrame.cpp:1686-1704,1732-1734 mvpGetPointer as a byte offset inside the super plane (level 0)
__device__ __forceinline__ unsigned sup_offset(const PlaneG &g, int pel, int logPel, int bps, int nX, int nY) {
    nX += g.hpadPel; nY += g.vpadPel;
    const int m = pel - 1;
    const int idx = (nX & m) | ((nY & m) << logPel);
    return (unsigned)(idx * g.supPlaneStride + (long long)(nY >> logPel) * g.supPitch + (long long)(nX >> logPel) * bps);
}

// (r6: left alone the register allocator takes 162 registers for six references -- three waves per SIMD; asked for six it needs 71 and spills nothing:
// 4.41 -> 2.78 ms per 341 4K16 frames, profiles/r6_degrain_window_plan_ab.txt)
#ifndef MVX_DG_PLAN_WAVES
#define MVX_DG_PLAN_WAVES(NR) ((NR) <= 6 ? 6 : 4)
#endif
template <int NR>
__global__ __launch_bounds__(256, MVX_DG_PLAN_WAVES(NR)) void degrain_plan_kernel(const DGParams *Pp, const DGJob *jobs, const int *usable, PlanRecT<NR> *plan) {
    typedef PlanRecT<NR> PlanRec;
    const DGParams &P = *Pp;
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const DGJob &J = jobs[f];
    constexpr int n = NR; // (= P.nRefs)
    // where level 0 starts inside each blob: found once per workgroup (a walk over the per-level size headers: nLvCount - 1 dependent loads), not
    // once per thread and reference (r4); the vectors themselves through global-address-space pointers, all requested before the first is used
    __shared__ unsigned lv0[NR];
    if (threadIdx.x < NR) lv0[threadIdx.x] = usable[f * 12 + threadIdx.x] ? (unsigned)((const unsigned char *)mvx_level0(J.blobs[threadIdx.x], P.nLvCount) - J.blobs[threadIdx.x]) : 0u;
    __syncthreads();
    if (i >= P.nBlk) return;
    const int by = i / P.nBlkX, bx = i - by * P.nBlkX;
    int vx[NR], vy[NR]; long long sad[NR]; int us[NR];
    typedef unsigned pl_v4 __attribute__((ext_vector_type(4), aligned(4)));
    pl_v4 vv[NR];
#pragma unroll
    for (int r = 0; r < n; r++) {
        us[r] = usable[f * 12 + r];
        vv[r] = pl_v4{0, 0, 0, 0};
        if (us[r]) vv[r] = *(DG_GL const pl_v4 *)(dg_gl(J.blobs[r]) + lv0[r] + (size_t)i * sizeof(GVecD)); // (a reference that is not usable may have no blob at all)
    }
#pragma unroll
    for (int r = 0; r < n; r++) { vx[r] = (int)vv[r][0]; vy[r] = (int)vv[r][1]; sad[r] = (long long)(((unsigned long long)vv[r][3] << 32) | vv[r][2]); }
    const int ncls = P.nplanes > 1 ? 2 : 1;
    for (int c = 0; c < ncls; c++) {
        const PlaneG &g = P.pl[c];
        PlanRec rec;
        int W[NR], WSum = 256 + 1;
        for (int r = 0; r < n; r++) {
            W[r] = 0; rec.off[r] = 0;
            if (us[r]) { // MVDegrains.h:192-200 useBlock; block origin Fakery.c:31-32
                const int blx = ((bx * P.pl[0].stepX) << P.logPel) + vx[r], bly = ((by * P.pl[0].stepY) << P.logPel) + vy[r];
                rec.off[r] = sup_offset(g, P.pel, P.logPel, P.bps, c ? blx >> g.subX : blx, c ? bly >> g.subY : bly);
                // a block that starts at an odd sample is read from the shifted copy, where it starts at a dword-aligned address: same samples, but a
                // wave's 16-byte loads at 2-byte-aligned addresses cost the texture addresser 64 cycles instead of 16 (tools/micro/ta_pattern.hip, patterns
                // 21 / 22), and these loads keep it busy 78 % of the kernel's run time (profiles/r4_stream_kernel_counters.txt)
                if (g.shadow && (rec.off[r] & 2u)) rec.off[r] = (rec.off[r] & ~3u) + (unsigned)g.shadow;
                W[r] = degrain_weight(P.thSAD[g.thIdx], sad[r]);
            }
            WSum += W[r];
        }
        const double scale = 256.0 / WSum; // MVDegrains.h:208-223 normaliseWeights
        int WSrc = 256;
        for (int r = 0; r < n; r++) { W[r] = (int)(W[r] * scale); WSrc -= W[r]; rec.w[r] = (short)W[r]; }
        rec.wsrc = (short)WSrc; rec.pad = 0;
        plan[((size_t)f * 2 + c) * P.nBlk + i] = rec;
    }
}

template <typename T, int NR>
__global__ __launch_bounds__(256) void degrain_kernel(const DGParams *Pp, const DGJob *jobs, const PlanRecT<NR> *plan) {
    typedef PlanRecT<NR> PlanRec;
    const DGParams &P = *Pp;
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= P.nplanes) return;
    const PlaneG &g = P.pl[p];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= g.H) return;
    const DGJob &J = jobs[f];
    const T *srow = (const T *)(J.src[p] + (long long)y * g.srcPitch);
    T *drow = (T *)(J.dst[p] + (long long)y * g.dstPitch);
    const int s = srow[x];
    if (!g.process || x >= g.WB || y >= g.HB) { drow[x] = (T)s; return; } // MVDegrains.cpp:211-214,238-249,290-298
    const PlanRec *pl = plan + ((size_t)f * 2 + (p ? 1 : 0)) * P.nBlk;
    int out;
    if (!P.overlap) {
        const int bx = x / g.blkW, by = y / g.blkH;
        const int px = x - bx * g.blkW, py = y - by * g.blkH;
        const PlanRec &R = pl[by * P.nBlkX + bx];
        int sum = 128 + s * R.wsrc;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int w = R.w[r];
            if (w) sum += (int)((const T *)(J.refs[r][p] + R.off[r] + (long long)py * g.supPitch))[px] * w;
        }
        out = (T)(sum >> 8);
    } else {
        // blocks covering this sample: bx in [bx0, bx1], by in [by0, by1]
        int bx1 = x / g.stepX; if (bx1 > P.nBlkX - 1) bx1 = P.nBlkX - 1;
        int bx0 = x - g.blkW + 1 <= 0 ? 0 : (x - g.blkW + g.stepX) / g.stepX;
        int by1 = y / g.stepY; if (by1 > P.nBlkY - 1) by1 = P.nBlkY - 1;
        int by0 = y - g.blkH + 1 <= 0 ? 0 : (y - g.blkH + g.stepY) / g.stepY;
        unsigned acc = 0;
        const int16_t *win = P.win[p];
        for (int by = by0; by <= by1; by++) {
            const int py = y - by * g.stepY;
            const int wby = by == 0 ? 0 : (by == P.nBlkY - 1 ? 6 : 3); // ((by + nBlkY - 3) / (nBlkY - 2)) * 3, MVDegrains.cpp:256
            for (int bx = bx0; bx <= bx1; bx++) {
                const int px = x - bx * g.stepX;
                const int wbx = bx == P.nBlkX - 1 ? 2 : (bx == 0 ? 0 : 1); // :260-262,285
                const PlanRec &R = pl[by * P.nBlkX + bx];
                int sum = 128 + s * R.wsrc;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const int w = R.w[r];
                    if (w) sum += (int)((const T *)(J.refs[r][p] + R.off[r] + (long long)py * g.supPitch))[px] * w;
                }
                const int val = (T)(sum >> 8);
                acc += (unsigned)((val * (int)win[(wby + wbx) * g.blkW * g.blkH + py * g.blkW + px]) >> 6);
            }
        }
        if (sizeof(T) == 1) acc &= 0xffffu; // 16-bit accumulator of the 8-bit path (Overlap.cpp:254-256); never overflows in practice
        const int a = (int)((acc + 16) >> 5); // Overlap.cpp:335-356
        const int pm = (1 << P.bits) - 1;
        out = a > pm ? pm : a;
    }
    if (g.limit < (1 << P.bits) - 1) { // MVDegrains.h:163-181
        int lo = s - g.limit, hi = s + g.limit;
        out = out < lo ? lo : out;
        out = out > hi ? hi : out;
    }
    drow[x] = (T)out;
}

// ---- vectorised gather for overlapped blocks: one thread per row of an overlap cell.
// A cell is stepX consecutive samples starting at a multiple of stepX: all of them are covered by the same <= 2 x 2
// blocks, and inside one block they are contiguous, so every (block, reference) costs ONE unaligned vector load of
// W = stepX samples instead of W scalar loads, and the plan record / window row are fetched once per W outputs.
// Same arithmetic per sample as degrain_kernel (Degrain_C + overlaps_c + ToPixels + LimitChanges).
// Pointers that come out of the job tables are generic ("flat") to the compiler; flat loads are slower and every one of them is waited
// for with vmcnt(0) lgkmcnt(0).  The vector helpers therefore take global-address-space pointers (dg_gl casts).
__device__ __forceinline__ DG_GL unsigned char *dg_glw(void *p) { return (DG_GL unsigned char *)(unsigned long long)p; }
typedef unsigned dg_uv4 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned dg_uv2 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned dg_uv1 __attribute__((aligned(1)));
typedef unsigned short dg_uh1 __attribute__((aligned(1)));

// W samples of type T from an arbitrarily aligned address, widened to int
template <typename T, int W> __device__ __forceinline__ void dg_load(DG_GL const unsigned char *p, int *o) {
    constexpr int BYTES = W * (int)sizeof(T);
    unsigned d[(BYTES + 3) / 4];
    if (BYTES >= 16) {
#pragma unroll
        for (int k = 0; k < BYTES / 16; k++) { dg_uv4 t = *(DG_GL const dg_uv4 *)(p + 16 * k); d[4 * k] = t[0]; d[4 * k + 1] = t[1]; d[4 * k + 2] = t[2]; d[4 * k + 3] = t[3]; }
    } else if (BYTES == 8) { dg_uv2 t = *(DG_GL const dg_uv2 *)p; d[0] = t[0]; d[1] = t[1]; }
    else if (BYTES == 4) d[0] = *(DG_GL const dg_uv1 *)p;
    else d[0] = *(DG_GL const dg_uh1 *)p;
#pragma unroll
    for (int i = 0; i < W; i++) {
        if (sizeof(T) == 2) o[i] = (int)((d[i >> 1] >> (16 * (i & 1))) & 0xffffu);
        else o[i] = (int)((d[i >> 2] >> (8 * (i & 3))) & 0xffu);
    }
}
// the same in two steps, so that several loads can be in flight before the first one is unpacked
template <typename T, int W> struct DgRaw { unsigned d[(W * (int)sizeof(T) + 3) / 4]; };
template <typename T, int W> __device__ __forceinline__ DgRaw<T, W> dg_load_raw(DG_GL const unsigned char *p) {
    constexpr int BYTES = W * (int)sizeof(T);
    DgRaw<T, W> r;
    if (BYTES >= 16) {
#pragma unroll
        for (int k = 0; k < BYTES / 16; k++) { dg_uv4 t = *(DG_GL const dg_uv4 *)(p + 16 * k); r.d[4 * k] = t[0]; r.d[4 * k + 1] = t[1]; r.d[4 * k + 2] = t[2]; r.d[4 * k + 3] = t[3]; }
    } else if (BYTES == 8) { dg_uv2 t = *(DG_GL const dg_uv2 *)p; r.d[0] = t[0]; r.d[1] = t[1]; }
    else if (BYTES == 4) r.d[0] = *(DG_GL const dg_uv1 *)p;
    else r.d[0] = *(DG_GL const dg_uh1 *)p;
    return r;
}
template <typename T, int W> __device__ __forceinline__ int dg_sample(const DgRaw<T, W> &r, int i) {
    return sizeof(T) == 2 ? (int)((r.d[i >> 1] >> (16 * (i & 1))) & 0xffffu) : (int)((r.d[i >> 2] >> (8 * (i & 3))) & 0xffu);
}
// r5: the weighted sum of TWO references per instruction for 16-bit samples (MVDegrains.h:31-53: sum += ref * W, all terms non-negative and < 2^24):
// v_perm_b32 pairs sample i of reference a with sample i of reference b, v_dot2_u32_u16 multiplies the pair by the two weights and adds.  Per sample and pair of
// references 2 instructions instead of 4 (two shifts / masks + two multiply-adds); same integers.
typedef unsigned short dg_us2 __attribute__((ext_vector_type(2)));
template <int W> __device__ __forceinline__ void dg_mad_pair_u16(const DgRaw<unsigned short, W> &a, const DgRaw<unsigned short, W> &b, int wa, int wb, int *sum) {
    const dg_us2 w = { (unsigned short)wa, (unsigned short)wb };
#pragma unroll
    for (int j = 0; j < (W + 1) / 2; j++) {
        const unsigned lo = __builtin_amdgcn_perm(b.d[j], a.d[j], 0x05040100u); // (a[2j], b[2j])
        sum[2 * j] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(dg_us2, lo), w, (unsigned)sum[2 * j], false);
        if (2 * j + 1 < W) {
            const unsigned hi = __builtin_amdgcn_perm(b.d[j], a.d[j], 0x07060302u); // (a[2j + 1], b[2j + 1])
            sum[2 * j + 1] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(dg_us2, hi), w, (unsigned)sum[2 * j + 1], false);
        }
    }
}
// the weighted references of one block into sum[]: pairs of references for 16-bit samples (NR is even: 2 x radius), one at a time otherwise
template <typename T, int NR, int W, typename REC> __device__ __forceinline__ void dg_mad_refs(const DgRaw<T, W> *raw, const REC &R, int *sum) {
    if constexpr (sizeof(T) == 2 && NR % 2 == 0) {
#pragma unroll
        for (int r = 0; r < NR; r += 2) dg_mad_pair_u16<W>(raw[r], raw[r + 1], R.w[r], R.w[r + 1], sum);
    } else {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int w = R.w[r];
#pragma unroll
            for (int i = 0; i < W; i++) sum[i] += dg_sample<T, W>(raw[r], i) * w;
        }
    }
}
// N consecutive ints (dword-aligned address)
typedef int dg_iv4 __attribute__((ext_vector_type(4), aligned(4)));
typedef int dg_iv2 __attribute__((ext_vector_type(2), aligned(4)));
template <int N> __device__ __forceinline__ void dg_load_ints(DG_GL const unsigned char *p, int *o) {
    if (N >= 4) {
#pragma unroll
        for (int k = 0; k < N / 4; k++) { const dg_iv4 t = *(DG_GL const dg_iv4 *)(p + 16 * k); o[4 * k] = t[0]; o[4 * k + 1] = t[1]; o[4 * k + 2] = t[2]; o[4 * k + 3] = t[3]; }
    } else if (N == 2) { const dg_iv2 t = *(DG_GL const dg_iv2 *)p; o[0] = t[0]; o[1] = t[1]; }
    else o[0] = *(DG_GL const int *)p;
}
template <typename T, int W> __device__ __forceinline__ void dg_store(DG_GL unsigned char *p, const int *v) {
    constexpr int BYTES = W * (int)sizeof(T);
    unsigned d[(BYTES + 3) / 4];
#pragma unroll
    for (int k = 0; k < (BYTES + 3) / 4; k++) d[k] = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
        if (sizeof(T) == 2) d[i >> 1] |= (unsigned)v[i] << (16 * (i & 1));
        else d[i >> 2] |= (unsigned)v[i] << (8 * (i & 3));
    }
    if (BYTES >= 16) {
#pragma unroll
        for (int k = 0; k < BYTES / 16; k++) { dg_uv4 t = { d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3] }; *(DG_GL dg_uv4 *)(p + 16 * k) = t; }
    } else if (BYTES == 8) { dg_uv2 t = { d[0], d[1] }; *(DG_GL dg_uv2 *)p = t; }
    else if (BYTES == 4) *(DG_GL dg_uv1 *)p = d[0];
    else *(DG_GL dg_uh1 *)p = (unsigned short)d[0];
}

#ifndef MVX_DG_TILE
#define MVX_DG_TILE 1 // (developer A/B builds: 0 = every thread reads its plan records from global memory, the form of rounds 2-5)
#endif
#define DG_TILE_MAX (34 * 6) // plan records of a workgroup's tile (32 cells x 8 rows): 33 block columns x at most 6 block rows (2-sample cell rows: five)
// blocks of blkW x blkH stepping by stepX x stepY: the records that cover 32 cells x 8 rows fit the tile (always, for the geometries the cell kernel is launched on:
// a power-of-two step below the block size is half the block size)
constexpr bool dg_tiled(int nrefs) { return nrefs >= 6; }
static bool dg_tile_fits(int blkW, int blkH, int stepX, int stepY) { return (31 + (blkW + stepX - 1) / stepX) * (7 / stepY + (blkH + stepY - 1) / stepY + 1) <= DG_TILE_MAX; }
#ifndef MVX_DG_W4_NR6_WAVES
#define MVX_DG_W4_NR6_WAVES 7 // (eight needs six spilled registers with the tile: 12.1 against 9.8 ms per 341 4K16 chroma frame pairs, profiles/r6_degrain_tile_ab.txt)
#endif
template <int NR, int W> constexpr int dg_cell_waves() { return W <= 2 ? 7 : W == 4 ? (NR <= 4 ? 8 : NR <= 6 ? MVX_DG_W4_NR6_WAVES : 6) : W == 8 ? (NR <= 2 ? 6 : NR <= 6 ? 5 : 4) : (NR <= 8 ? 3 : 2); }
template <typename T, int NR, int W>
__global__ __launch_bounds__(256, (dg_cell_waves<NR, W>())) void degrain_cell_kernel(const DGParams *Pp, const DGJob *jobs, const PlanRecT<NR> *plan, int planeFirst, int planesPerFrame, int xcdOrder) {
    typedef PlanRecT<NR> PlanRec;
    const DGParams &P = *Pp;
    int bxi = blockIdx.x, byi = blockIdx.y, z = blockIdx.z;
    if (xcdOrder) { // workgroup w runs on XCD w % 8: give every XCD a contiguous range of tiles, so that vertically adjacent tiles (which
        // read the same reference rows) meet in one L2 instead of fetching those rows from HBM once per XCD
        const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
        const unsigned w = bxi + gx * (byi + gy * z), x8 = w & 7, slot = w >> 3;
        const unsigned nl = x8 * (n >> 3) + min(x8, n & 7) + slot;
        z = nl / (gx * gy);
        const unsigned rem = nl - z * (gx * gy);
        byi = rem / gx; bxi = rem - byi * gx;
    }
    const int f = z / planesPerFrame, p = planeFirst + z % planesPerFrame;
    const PlaneG &g = P.pl[p];
    const int c = bxi * 32 + (threadIdx.x & 31), y = byi * 8 + (threadIdx.x >> 5);
    const int x0 = c * W;
    // r6: the plan records of the blocks that cover this workgroup's tile (32 cells x 8 rows: with blocks overlapping by half 33 block columns x 2-5 block
    // rows) are fetched ONCE, cooperatively, into LDS.  Before, every thread read the records of its <= 4 covering blocks itself: three of the eleven memory
    // instructions of a block visit (12 of 45 per thread; a 16-byte load costs the texture addresser its 16 cycles whether or not the line is in L1), and each
    // visit was two dependent round trips (record -> reference rows)
    // Measured (profiles/r6_degrain_tile_ab.txt, ms per batch, every thread for itself -> tile): six 16-bit references 15.9 -> 14.3 (8-sample cells) and 12.8 -> 9.8
    // (4-sample cells), twelve 47.2 -> 43.0 / 29.2 -> 25.2; TWO 8-bit references (16-byte records, two loads per visit) lose: 15.5 -> 16.6, 12.7 -> 15.6 -- the tile is
    // for filters with six or more references (Degrain3-6)
    constexpr bool TILED = MVX_DG_TILE && dg_tiled(NR);
    constexpr int RD = (int)sizeof(PlanRec) / 4;
    __shared__ __attribute__((aligned(16))) unsigned tileD[TILED ? DG_TILE_MAX * RD : 1];
    int tBx = 0, tBy = 0, tNbx = 0; // (wave-uniform: scalar registers)
    // (the window taps next to the tile -- the nine windows of 16x16 blocks are 4.5 KB -- measured and removed: 14.7 against 14.5 ms, cfg5's 8-sample cells 26.2 against 25.1:
    // profiles/r6_degrain_window_plan_ab.txt)
    if (TILED && g.process && P.overlap) { // (uniform over the workgroup; the host launches this kernel only when the tile fits: dg_tile_fits)
        const int xt = bxi * 32 * W, yt = byi * 8;
        tBx = __builtin_amdgcn_readfirstlane(xt - g.blkW + 1 <= 0 ? 0 : (xt - g.blkW + g.stepX) / g.stepX);
        tBy = __builtin_amdgcn_readfirstlane(yt - g.blkH + 1 <= 0 ? 0 : (yt - g.blkH + g.stepY) / g.stepY);
        const int bxHi = min(bxi * 32 + 31, P.nBlkX - 1), byHi = min((yt + 7) / g.stepY, P.nBlkY - 1);
        tNbx = __builtin_amdgcn_readfirstlane(bxHi - tBx + 1);
        const int nby = __builtin_amdgcn_readfirstlane(byHi - tBy + 1);
        if (tNbx > 0) {
            const unsigned *src = (const unsigned *)(plan + ((size_t)f * 2 + (p ? 1 : 0)) * P.nBlk);
            const int rowD = tNbx * RD;
            for (int r = 0; r < nby; r++) {
                DG_GL const unsigned *q = (DG_GL const unsigned *)dg_gl((const unsigned char *)(src + ((size_t)(tBy + r) * P.nBlkX + tBx) * RD));
                for (int k = threadIdx.x; k < rowD; k += 256) tileD[r * rowD + k] = q[k];
            }
        }
        __syncthreads();
    }
    if (x0 >= g.W || y >= g.H) return;
    const DGJob &J = jobs[f];
    const unsigned char *srow = J.src[p] + (long long)y * g.srcPitch + (long long)x0 * sizeof(T);
    unsigned char *drow = J.dst[p] + (long long)y * g.dstPitch + (long long)x0 * sizeof(T);
    const bool fullW = x0 + W <= g.W;
    int s[W];
    if (fullW) dg_load<T, W>(dg_gl(srow), s);
    else {
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = x0 + i < g.W ? (int)((const T *)srow)[i] : 0;
    }
    int out[W];
#pragma unroll
    for (int i = 0; i < W; i++) out[i] = s[i];
    if (g.process && x0 < g.WB && y < g.HB && !P.overlap) {
        // r5: blocks side by side (MVDegrains.cpp:238-249: Degrain_C straight into the frame, no windows).  W divides the block width, so the cell lies in ONE
        // block: one plan record, one vector load per reference -- the per-sample gather this replaces took 42 of the 135 ms of a 2048-frame 1080p step
        const PlanRec &R = (plan + ((size_t)f * 2 + (p ? 1 : 0)) * P.nBlk)[(y / g.blkH) * P.nBlkX + x0 / g.blkW];
        const int px = x0 % g.blkW, py = y % g.blkH;
        const unsigned char *safe = nullptr;
#pragma unroll
        for (int r = 0; r < NR; r++) if (J.refs[r][p]) safe = J.refs[r][p];
        const long long rowOff = safe ? (long long)py * g.supPitch + (long long)px * sizeof(T) : 0;
        DgRaw<T, W> raw[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) raw[r] = dg_load_raw<T, W>(dg_gl((J.refs[r][p] ? J.refs[r][p] : (safe ? safe : J.src[p])) + R.off[r] + rowOff)); // (weight 0: any valid address)
        const int wsrc = R.wsrc, pm = (1 << P.bits) - 1;
        int sum[W];
#pragma unroll
        for (int i = 0; i < W; i++) sum[i] = 128 + s[i] * wsrc;
        dg_mad_refs<T, NR, W>(raw, R, sum);
#pragma unroll
        for (int i = 0; i < W; i++) {
            int o = (T)(sum[i] >> 8);
            if (g.limit < pm) { // MVDegrains.h:163-181
                const int lo = s[i] - g.limit, hi = s[i] + g.limit;
                o = o < lo ? lo : o;
                o = o > hi ? hi : o;
            }
            out[i] = o;
        }
    } else if (g.process && x0 < g.WB && y < g.HB) { // MVDegrains.cpp:211-214,238-249,290-298: uncovered strips keep the source
        const PlanRec *pl = plan + ((size_t)f * 2 + (p ? 1 : 0)) * P.nBlk;
        int bx1 = c; if (bx1 > P.nBlkX - 1) bx1 = P.nBlkX - 1;
        const int bx0 = x0 - g.blkW + 1 <= 0 ? 0 : (x0 - g.blkW + g.stepX) / g.stepX;
        int by1 = y / g.stepY; if (by1 > P.nBlkY - 1) by1 = P.nBlkY - 1;
        const int by0 = y - g.blkH + 1 <= 0 ? 0 : (y - g.blkH + g.stepY) / g.stepY;
        unsigned acc[W];
#pragma unroll
        for (int i = 0; i < W; i++) acc[i] = 0;
        const int16_t *win = P.win[p];
        // a frame outside the clip has no super frame: its loads (weight 0, plan offset 0) go to a super plane the job does have, so that
        // the row offset (super pitch) stays inside the buffer; a job without any reference reads the source plane at offset 0
        const unsigned char *refp[NR], *safe = nullptr;
#pragma unroll
        for (int r = 0; r < NR; r++) if (J.refs[r][p]) safe = J.refs[r][p];
#pragma unroll
        for (int r = 0; r < NR; r++) refp[r] = J.refs[r][p] ? J.refs[r][p] : (safe ? safe : J.src[p]);
        for (int by = by0; by <= by1; by++) {
            const int py = y - by * g.stepY;
            const int wby = by == 0 ? 0 : (by == P.nBlkY - 1 ? 6 : 3);
            for (int bx = bx0; bx <= bx1; bx++) {
                const int px = x0 - bx * g.stepX;   // >= 0; the block covers samples i < blkW - px of this cell
                const int nv = g.blkW - px;
                if (nv <= 0) continue;
                const int wbx = bx == P.nBlkX - 1 ? 2 : (bx == 0 ? 0 : 1);
                PlanRec R;
                if (TILED) __builtin_memcpy(&R, &tileD[((by - tBy) * tNbx + (bx - tBx)) * RD], sizeof(PlanRec));
                else R = pl[by * P.nBlkX + bx];
                const int wsrc = R.wsrc;
                int sum[W];
#pragma unroll
                for (int i = 0; i < W; i++) sum[i] = 128 + s[i] * wsrc;
                const long long rowOff = safe ? (long long)py * g.supPitch + (long long)px * sizeof(T) : 0;
                const int16_t *wrow = win + (wby + wbx) * g.blkW * g.blkH + py * g.blkW + px;
                DgRaw<unsigned short, W> wr; // the window taps of this row of the cell, packed (unpacked where they are used: four registers fewer across the weighted sums)
                if (nv >= W) { // whole cell inside the block: vector loads, ALL of the block's references requested before the first is used
                    // (a reference with weight 0 is loaded too -- from a valid address: refp -- which costs bandwidth the kernel has to
                    // spare; waiting for every load in turn, as a `if (w)` around each one makes the compiler do, cost 4x the round trips)
                    DgRaw<T, W> raw[NR];
#pragma unroll
                    for (int r = 0; r < NR; r++) raw[r] = dg_load_raw<T, W>(dg_gl(refp[r] + R.off[r] + rowOff));
                    wr = dg_load_raw<unsigned short, W>(dg_gl((const unsigned char *)wrow));
                    dg_mad_refs<T, NR, W>(raw, R, sum);
                } else { // partially covered (overlap != block/2): per-sample loads of the covered samples only
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        const int w = R.w[r];
                        if (w) {
                            const T *q = (const T *)(J.refs[r][p] + R.off[r] + rowOff);
#pragma unroll
                            for (int i = 0; i < W; i++) if (i < nv) sum[i] += (int)q[i] * w;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < (W + 1) / 2; k++) wr.d[k] = 0;
#pragma unroll
                    for (int i = 0; i < W; i++) if (i < nv) wr.d[i >> 1] |= (unsigned)(unsigned short)wrow[i] << (16 * (i & 1));
                }
#pragma unroll
                for (int i = 0; i < W; i++) {
                    const int val = (T)(sum[i] >> 8);
                    if (i < nv) acc[i] += (unsigned)((val * dg_sample<unsigned short, W>(wr, i)) >> 6);
                }
            }
        }
        const int pm = (1 << P.bits) - 1;
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (x0 + i < g.WB) {
                unsigned a0 = acc[i];
                if (sizeof(T) == 1) a0 &= 0xffffu; // 16-bit accumulator of the 8-bit path (Overlap.cpp:254-256)
                const int a = (int)((a0 + 16) >> 5); // Overlap.cpp:335-356
                int o = a > pm ? pm : a;
                if (g.limit < pm) { // MVDegrains.h:163-181
                    const int lo = s[i] - g.limit, hi = s[i] + g.limit;
                    o = o < lo ? lo : o;
                    o = o > hi ? hi : o;
                }
                out[i] = o;
            }
        }
    }
    if (fullW) dg_store<T, W>(dg_glw(drow), out);
    else {
#pragma unroll
        for (int i = 0; i < W; i++) if (x0 + i < g.W) ((T *)drow)[i] = (T)out[i];
    }
}

// ---- compensate

struct CPlanRec { unsigned off[2]; int fromRef; }; // off[0] luma, off[1] chroma

__global__ __launch_bounds__(256) void compensate_plan_kernel(const DGParams *Pp, const DGJob *jobs, const int *usable, CPlanRec *plan) {
    const DGParams &P = *Pp;
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.nBlk || !usable[f * 12]) return;
    const int by = i / P.nBlkX, bx = i - by * P.nBlkX;
    const GVecD *v = mvx_level0(jobs[f].blobs[0], P.nLvCount);
    const GVecD b = v[i];
    int blx, bly;
    CPlanRec rec;
    if (b.sad < P.cthSAD) { // MVCompensate.c:238-242,286-290
        blx = bx * P.pl[0].stepX * P.pel + b.x * P.time256 / 256;
        bly = by * P.pl[0].stepY * P.pel + b.y * P.time256 / 256 + jobs[f].fieldShift;
        rec.fromRef = 1;
    } else { // :243-247,291-295 (no-overlap: stepX == blkW)
        blx = bx * P.pl[0].stepX * P.pel;
        bly = by * P.pl[0].stepY * P.pel + jobs[f].fieldShift;
        rec.fromRef = 0;
    }
    rec.off[0] = sup_offset(P.pl[0], P.pel, P.logPel, P.bps, blx, bly);
    rec.off[1] = P.nplanes > 1 ? sup_offset(P.pl[1], P.pel, P.logPel, P.bps, blx >> P.pl[1].subX, bly >> P.pl[1].subY) : 0;
    plan[(size_t)f * P.nBlk + i] = rec;
}

// job layout for compensate: src[] = super frame n, refs[0][] = super frame nref (may be null), blobs[0], dst[]
template <typename T>
__global__ __launch_bounds__(256) void compensate_kernel(const DGParams *Pp, const DGJob *jobs, const int *usable, const CPlanRec *plan) {
    const DGParams &P = *Pp;
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= P.nplanes) return;
    const PlaneG &g = P.pl[p];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= g.H) return;
    const DGJob &J = jobs[f];
    T *drow = (T *)(J.dst[p] + (long long)y * g.dstPitch);
    const unsigned char *srcSup = J.src[p], *refSup = J.refs[0][p];
    // interior (pel plane 0) sample of a super frame
    const long long inner = (long long)(g.vpadPel / P.pel + y) * g.supPitch + (long long)(g.hpadPel / P.pel + x) * (long long)sizeof(T);
    if (!usable[f * 12]) { // MVCompensate.c:348-364
        const unsigned char *s = (!P.scBehavior && refSup) ? refSup : srcSup;
        drow[x] = *(const T *)(s + inner);
        return;
    }
    if (x >= g.WB || y >= g.HB) { // :319-342
        const unsigned char *s = P.scBehavior ? srcSup : refSup;
        drow[x] = *(const T *)(s + inner);
        return;
    }
    const CPlanRec *pl = plan + (size_t)f * P.nBlk;
    const int c = p ? 1 : 0;
    if (!P.overlap) {
        const int bx = x / g.blkW, by = y / g.blkH;
        const int px = x - bx * g.blkW, py = y - by * g.blkH;
        const CPlanRec R = pl[by * P.nBlkX + bx];
        drow[x] = ((const T *)((R.fromRef ? refSup : srcSup) + R.off[c] + (long long)py * g.supPitch))[px];
        return;
    }
    int bx1 = x / g.stepX; if (bx1 > P.nBlkX - 1) bx1 = P.nBlkX - 1;
    int bx0 = x - g.blkW + 1 <= 0 ? 0 : (x - g.blkW + g.stepX) / g.stepX;
    int by1 = y / g.stepY; if (by1 > P.nBlkY - 1) by1 = P.nBlkY - 1;
    int by0 = y - g.blkH + 1 <= 0 ? 0 : (y - g.blkH + g.stepY) / g.stepY;
    unsigned acc = 0;
    const int16_t *win = P.win[p];
    for (int by = by0; by <= by1; by++) {
        const int py = y - by * g.stepY;
        const int wby = by == 0 ? 0 : (by == P.nBlkY - 1 ? 6 : 3);
        for (int bx = bx0; bx <= bx1; bx++) {
            const int px = x - bx * g.stepX;
            const int wbx = bx == P.nBlkX - 1 ? 2 : (bx == 0 ? 0 : 1);
            const CPlanRec R = pl[by * P.nBlkX + bx];
            const int val = ((const T *)((R.fromRef ? refSup : srcSup) + R.off[c] + (long long)py * g.supPitch))[px];
            acc += (unsigned)((val * (int)win[(wby + wbx) * g.blkW * g.blkH + py * g.blkW + px]) >> 6);
        }
    }
    if (sizeof(T) == 1) acc &= 0xffffu;
    const int a = (int)((acc + 16) >> 5);
    const int pm = (1 << P.bits) - 1;
    drow[x] = (T)(a > pm ? pm : a);
}


// ---- blocks without overlap (MVCompensate.c:227-258,286-306: plain block copies): one thread per CW consecutive samples of one
// block row -- one vector load from the chosen super frame, one vector store.  CW divides the block width, the covered width and
// the frame width (the host checks), so a segment is never split between cases; block sizes are powers of two.
template <typename T, int CW>
__global__ __launch_bounds__(256) void compensate_rows_kernel(const DGParams *Pp, const DGJob *jobs, const int *usable, const CPlanRec *plan, int planeFirst, int planesPerFrame) {
    const DGParams &P = *Pp;
    const int z = blockIdx.z, f = z / planesPerFrame, p = planeFirst + z % planesPerFrame;
    const PlaneG &g = P.pl[p];
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * CW, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= g.H) return;
    const DGJob &J = jobs[f];
    const unsigned char *srcSup = J.src[p], *refSup = J.refs[0][p];
    const long long inner = (long long)(g.vpadPel / P.pel + y) * g.supPitch + (long long)(g.hpadPel / P.pel + x) * (long long)sizeof(T);
    const unsigned char *sp;
    if (!usable[f * 12]) sp = ((!P.scBehavior && refSup) ? refSup : srcSup) + inner;            // MVCompensate.c:348-364
    else if (x >= g.WB || y >= g.HB) sp = (P.scBehavior ? srcSup : refSup) + inner;             // :319-342
    else {
        const int lbw = __ffs(g.blkW) - 1, lbh = __ffs(g.blkH) - 1;
        const int bx = x >> lbw, by = y >> lbh, px = x & (g.blkW - 1), py = y & (g.blkH - 1);
        const CPlanRec R = plan[(size_t)f * P.nBlk + by * P.nBlkX + bx];
        sp = (R.fromRef ? refSup : srcSup) + R.off[p ? 1 : 0] + (long long)py * g.supPitch + (long long)px * (long long)sizeof(T);
    }
    int v[CW];
    dg_load<T, CW>(dg_gl(sp), v);
    dg_store<T, CW>(dg_glw(J.dst[p] + (long long)y * g.dstPitch + (long long)x * (long long)sizeof(T)), v);
}

// ------------------------------------------------------------------------------------------------ host objects

struct DGCommon {
    CallGuard guard;
    DGParams P;
    DGParams *dP = nullptr;
    DGJob *dJobs = nullptr;
    size_t jobsCap = 0;
    int *dUsable = nullptr;
    void *dPlan = nullptr;
    size_t planCap = 0;
    int16_t *dWin[2] = { nullptr, nullptr };
    int nWinClasses = 1;
    ~DGCommon() {
        if (dP) (void)hipFree(dP);
        if (dJobs) (void)hipFree(dJobs);
        if (dUsable) (void)hipFree(dUsable);
        if (dPlan) (void)hipFree(dPlan);
        if (dWin[0]) (void)hipFree(dWin[0]);
        if (dWin[1]) (void)hipFree(dWin[1]);
    }
};
struct mvx_degrain : DGCommon { int radius; };
struct mvx_compensate : DGCommon {};

#define DFAIL(...) do { snprintf(err, MVX_ERRLEN, __VA_ARGS__); mvx_set_error("%s", err); return MVX_E_ARG; } while (0)

static int fill_common(DGCommon *h, const mvx_analysis_data *ad, const mvx_super_info &si, const ptrdiff_t src_pitch[3],
                       const ptrdiff_t super_pitch[3], const ptrdiff_t dst_pitch[3], char *err) {
    DGParams &P = h->P;
    P.nBlkX = ad->nBlkX; P.nBlkY = ad->nBlkY; P.nBlk = ad->nBlkX * ad->nBlkY;
    P.pel = ad->nPel; P.logPel = ad->nPel == 4 ? 2 : ad->nPel == 2 ? 1 : 0;
    P.bits = si.bits; P.bps = (si.bits + 7) / 8; P.nplanes = si.num_planes;
    P.overlap = ad->nOverlapX > 0 || ad->nOverlapY > 0;
    if (P.overlap && (ad->nBlkX < 3 || ad->nBlkY < 3)) DFAIL("overlap needs at least 3x3 blocks (window selection divides by nBlk-2).");
    P.nLvCount = ad->nLvCount;
    const int xSub = mvx_ilog2(si.xRatioUV), ySub = mvx_ilog2(si.yRatioUV);
    for (int p = 0; p < 3; p++) {
        PlaneG &g = P.pl[p];
        const int sx = p ? xSub : 0, sy = p ? ySub : 0;
        g.subX = sx; g.subY = sy;
        g.W = ad->nWidth >> sx; g.H = ad->nHeight >> sy;
        g.blkW = ad->nBlkSizeX >> sx; g.blkH = ad->nBlkSizeY >> sy;
        g.ovX = ad->nOverlapX >> sx; g.ovY = ad->nOverlapY >> sy;
        g.stepX = g.blkW - g.ovX; g.stepY = g.blkH - g.ovY;
        g.WB = (ad->nBlkX * (ad->nBlkSizeX - ad->nOverlapX) + ad->nOverlapX) >> sx;
        g.HB = (ad->nBlkY * (ad->nBlkSizeY - ad->nOverlapY) + ad->nOverlapY) >> sy;
        g.hpadPel = (si.hpad >> sx) * si.pel; g.vpadPel = (si.vpad >> sy) * si.pel; // MVFrame.cpp:1334-1335,1775-1779
        g.srcPitch = src_pitch ? src_pitch[p < si.num_planes ? p : 0] : 0;
        g.supPitch = super_pitch[p < si.num_planes ? p : 0];
        g.dstPitch = dst_pitch[p < si.num_planes ? p : 0];
        g.supPlaneStride = g.supPitch * (long long)((si.height >> sy) + 2 * (si.vpad >> sy));
        g.thIdx = p ? 1 : 0;
        g.process = 1; g.limit = (1 << si.bits) - 1;
        g.shadow = 0;
    }
    if (si.num_planes > 1 && super_pitch[1] != super_pitch[2]) DFAIL("U and V super planes must share one pitch.");
    h->nWinClasses = si.num_planes > 1 ? 2 : 1;
    return MVX_OK;
}

// device state is created on first use so that argument validation works without a GPU
static int finish_common(DGCommon *h) {
    if (h->dP) return MVX_OK;
    DGParams &P = h->P;
    if (P.overlap) {
        for (int c = 0; c < h->nWinClasses; c++) {
            const PlaneG &g = P.pl[c];
            std::vector<int16_t> w(9 * g.blkW * g.blkH);
            mvx_over_windows(w.data(), g.blkW, g.blkH, g.ovX, g.ovY);
            HIP_CHECK(hipMalloc((void **)&h->dWin[c], w.size() * 2));
            HIP_CHECK(hipMemcpy(h->dWin[c], w.data(), w.size() * 2, hipMemcpyHostToDevice));
        }
        P.win[0] = h->dWin[0]; P.win[1] = P.win[2] = h->dWin[1];
    }
    HIP_CHECK(hipMalloc((void **)&h->dP, sizeof(DGParams)));
    HIP_CHECK(hipMemcpy(h->dP, &h->P, sizeof(DGParams), hipMemcpyHostToDevice));
    return MVX_OK;
}

static int ensure_jobs(DGCommon *h, int nframes, size_t planBytesPerFrame) {
    if ((size_t)nframes > h->jobsCap) {
        if (h->dJobs) (void)hipFree(h->dJobs);
        if (h->dUsable) (void)hipFree(h->dUsable);
        h->jobsCap = (size_t)nframes * 2;
        HIP_CHECK(hipMalloc((void **)&h->dJobs, h->jobsCap * sizeof(DGJob)));
        HIP_CHECK(hipMalloc((void **)&h->dUsable, h->jobsCap * 12 * sizeof(int)));
    }
    size_t need = planBytesPerFrame * nframes;
    if (need > h->planCap) {
        if (h->dPlan) (void)hipFree(h->dPlan);
        h->planCap = need + need / 2;
        HIP_CHECK(hipMalloc(&h->dPlan, h->planCap));
    }
    return MVX_OK;
}

// MVDegrains.cpp:511-809 mvdegrainCreate
extern "C" __attribute__((visibility("default"))) int mvx_degrain_create(const mvx_degrain_args *a, const mvx_analysis_data *ad, const mvx_super *sup, const ptrdiff_t src_pitch[3],
                                  const ptrdiff_t super_pitch[3], const ptrdiff_t dst_pitch[3], mvx_degrain **out, char *err) {
    char dummy[MVX_ERRLEN];
    if (!err) err = dummy;
    err[0] = 0;
    *out = nullptr;
    const mvx_super_info &si = sup->info;
    const int radius = a->radius;
    if (radius < 1 || radius > 6) DFAIL("Degrain: radius must be between 1 and 6.");
    long long thSAD0 = a->thsad == MVX_UNSET ? 400 : a->thsad;
    long long thSAD1 = a->thsadc == MVX_UNSET ? thSAD0 : a->thsadc;
    int plane = a->plane == MVX_UNSET ? 4 : a->plane;
    long long nSCD1 = a->thscd1 == MVX_UNSET ? 400 : a->thscd1; // MV_DEFAULT_SCD1
    int nSCD2 = a->thscd2 == MVX_UNSET ? 130 : a->thscd2;
    if (plane < 0 || plane > 4) DFAIL("Degrain%d: plane must be between 0 and 4 (inclusive).", radius);
    static const int planes[5] = { 1, 2, 4, 6, 7 };
    const int YUVplanes = planes[plane];
    if (nSCD1 > 8 * 8 * 255) DFAIL("Degrain%d: thscd1 can be at most %d.", radius, 8 * 8 * 255);
    const long long nSCD1_old = nSCD1;
    int64_t s1 = nSCD1; int32_t s2 = nSCD2;
    mvx_scale_thscd(&s1, &s2, ad);
    nSCD1 = s1; nSCD2 = s2;
    thSAD0 = thSAD0 * nSCD1 / nSCD1_old; // :658-659
    thSAD1 = thSAD1 * nSCD1 / nSCD1_old;
    if (thSAD0 >= 2147483647LL || thSAD1 >= 2147483647LL) {
        const bool c = thSAD0 < 2147483647LL;
        DFAIL("Degrain%d: with this block size and video format, thsad%s must not exceed %lld or some calculations would overflow.", radius,
              c ? "c" : "", (long long)(2147483647LL * nSCD1_old / nSCD1));
    }
    if (ad->nHeight != si.height || ad->nWidth != si.super_width - si.hpad * 2 || ad->nWidth != si.width || ad->nPel != si.pel)
        DFAIL("Degrain%d: wrong source or super clip frame size.", radius);
    const int pixelMax = (1 << si.bits) - 1;
    int limit = a->limit == MVX_UNSET ? pixelMax : a->limit;
    int limitc = a->limitc == MVX_UNSET ? limit : a->limitc;
    if (limit < 0 || limit > pixelMax) DFAIL("Degrain%d: limit must be between 0 and %d (inclusive).", radius, pixelMax);
    if (limitc < 0 || limitc > pixelMax) DFAIL("Degrain%d: limitc must be between 0 and %d (inclusive).", radius, pixelMax);

    mvx_degrain *h = new mvx_degrain();
    h->radius = radius;
    memset(&h->P, 0, sizeof(h->P));
    int rc = fill_common(h, ad, si, src_pitch, super_pitch, dst_pitch, err);
    if (rc) { delete h; return rc; }
    DGParams &P = h->P;
    P.nRefs = 2 * radius;
    P.thSAD[0] = thSAD0; P.thSAD[1] = thSAD1; P.thscd1 = nSCD1; P.thscd2 = nSCD2;
    P.pl[0].process = !!(YUVplanes & 1);
    P.pl[1].process = !!(YUVplanes & 2 & si.modeYUV);
    P.pl[2].process = !!(YUVplanes & 4 & si.modeYUV);
    P.pl[0].limit = limit; P.pl[1].limit = P.pl[2].limit = limitc;
    *out = h;
    return MVX_OK;
}

extern "C" __attribute__((visibility("default"))) void mvx_degrain_destroy(mvx_degrain *d) { delete d; }

// The caller promises that every reference super frame of every job carries, behind its luma plane at + copy_stride[0], the plane shifted left by
// one sample (mvx_super_shadow_frames).  Only changes which addresses the kernels load from.
extern "C" __attribute__((visibility("default"))) int mvx_degrain_set_ref_shadow(mvx_degrain *d, const ptrdiff_t copy_stride[3]) {
    const long long v = copy_stride ? (long long)copy_stride[0] : 0;
    const PlaneG &g0 = d->P.pl[0];
    if (v < 0 || v % 16) { mvx_set_error("mvx_degrain_set_ref_shadow: copy strides must be non-negative multiples of 16 bytes"); return MVX_E_ARG; }
    // (plan records hold 32-bit byte offsets into a super plane: the copy must lie inside that range)
    if (v && v + g0.supPlaneStride * d->P.pel * d->P.pel >= 0xffffffffLL) { mvx_set_error("mvx_degrain_set_ref_shadow: the shifted copy lies beyond 4 GiB of the plane"); return MVX_E_ARG; }
    std::lock_guard<std::mutex> lk(d->guard.mu);
    d->P.pl[0].shadow = (d->P.bps == 2 && mvx_debug_value("degrain_shadow", 1)) ? v : 0;
    if (d->dP) HIP_CHECK(hipMemcpy(d->dP, &d->P, sizeof(DGParams), hipMemcpyHostToDevice));
    return MVX_OK;
}

template <typename T> static void launch_degrain(int nr, dim3 grid, hipStream_t st, const DGParams *dP, const DGJob *dJ, const void *plan) {
#define DG(N) hipLaunchKernelGGL((degrain_kernel<T, N>), grid, dim3(256), 0, st, dP, dJ, (const PlanRecT<N> *)plan)
    switch (nr) { case 2: DG(2); break; case 4: DG(4); break; case 6: DG(6); break; case 8: DG(8); break; case 10: DG(10); break; default: DG(12); break; }
#undef DG
}

template <typename T, int W> static void launch_degrain_cells_w(int nr, dim3 grid, hipStream_t st, const DGParams *dP, const DGJob *dJ, const void *plan, int p0, int npl) {
    // XCD-contiguous tile order: measured r2 (4K16 Degrain3, 512 frames): HBM fetch 124 -> 88 GB (luma) and 75 -> 34 GB (chroma), but the
    // launch gets 8 ms SLOWER (189 against 181 ms for everything but the search) -- the kernel is not HBM-bound.  Off.
    const int xo = mvx_debug_value("degrain_xcd", 0);
#define DGC(N) hipLaunchKernelGGL((degrain_cell_kernel<T, N, W>), grid, dim3(256), 0, st, dP, dJ, (const PlanRecT<N> *)plan, p0, npl, xo)
    switch (nr) { case 2: DGC(2); break; case 4: DGC(4); break; case 6: DGC(6); break; case 8: DGC(8); break; case 10: DGC(10); break; default: DGC(12); break; }
#undef DGC
}
template <typename T> static void launch_degrain_cells(int nr, int W, dim3 grid, hipStream_t st, const DGParams *dP, const DGJob *dJ, const void *plan, int p0, int npl) {
    switch (W) {
    case 2: launch_degrain_cells_w<T, 2>(nr, grid, st, dP, dJ, plan, p0, npl); break;
    case 4: launch_degrain_cells_w<T, 4>(nr, grid, st, dP, dJ, plan, p0, npl); break;
    case 8: launch_degrain_cells_w<T, 8>(nr, grid, st, dP, dJ, plan, p0, npl); break;
    default: launch_degrain_cells_w<T, 16>(nr, grid, st, dP, dJ, plan, p0, npl); break;
    }
}

extern "C" __attribute__((visibility("default"))) int mvx_degrain_frames(mvx_degrain *d, int nframes, const mvx_degrain_job *jobs, void *stream) {
    if (nframes <= 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    CallGuard::Scope scope(d->guard, st);
    int rc = finish_common(d);
    if (rc) return rc;
    const DGParams &P = d->P;
    if ((rc = ensure_jobs(d, nframes, plan_rec_bytes(P.nRefs) * 2 * (size_t)P.nBlk))) return rc;
    std::vector<DGJob> hj(nframes);
    for (int f = 0; f < nframes; f++) {
        memset(&hj[f], 0, sizeof(DGJob));
        for (int p = 0; p < 3; p++) { hj[f].src[p] = (const unsigned char *)jobs[f].src[p]; hj[f].dst[p] = (unsigned char *)jobs[f].dst[p]; }
        for (int r = 0; r < P.nRefs; r++) {
            for (int p = 0; p < 3; p++) hj[f].refs[r][p] = (const unsigned char *)jobs[f].refs[r][p];
            hj[f].blobs[r] = (const unsigned char *)jobs[f].blobs[r];
        }
    }
    HIP_CHECK(hipMemcpyAsync(d->dJobs, hj.data(), sizeof(DGJob) * nframes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(usable_kernel, dim3(P.nRefs, nframes), dim3(256), 0, st, d->dP, d->dJobs, d->dUsable, 0);
    {
        const dim3 pg((P.nBlk + 255) / 256, nframes);
#define DGP(N) hipLaunchKernelGGL(degrain_plan_kernel<N>, pg, dim3(256), 0, st, d->dP, d->dJobs, d->dUsable, (PlanRecT<N> *)d->dPlan)
        switch (P.nRefs) { case 2: DGP(2); break; case 4: DGP(4); break; case 6: DGP(6); break; case 8: DGP(8); break; case 10: DGP(10); break; default: DGP(12); break; }
#undef DGP
    }
    // overlapped blocks with a power-of-two step: vectorised cell kernel, one launch per plane class; otherwise the
    // per-sample gather
    // (r5: blocks side by side take the cell kernel too -- a cell of 8 (4) samples inside one block)
    auto cellW = [&](int p) { const int w = P.pl[p].stepX; return P.overlap ? ((w == 2 || w == 4 || w == 8 || w == 16) ? w : 0) : (P.pl[p].blkW % 8 == 0 ? 8 : P.pl[p].blkW % 4 == 0 ? 4 : 0); };
    auto tileOk = [&](int p) { return !MVX_DG_TILE || !dg_tiled(P.nRefs) || !P.overlap || dg_tile_fits(P.pl[p].blkW, P.pl[p].blkH, P.pl[p].stepX, P.pl[p].stepY); };
    const bool cells = cellW(0) && tileOk(0) && (P.nplanes == 1 || (cellW(1) && P.pl[1].stepX == P.pl[2].stepX && tileOk(1) && tileOk(2)));
    if (cells) {
        for (int cls = 0; cls < (P.nplanes > 1 ? 2 : 1); cls++) {
            const int p0 = cls, npl = cls ? 2 : 1, W = cellW(p0);
            dim3 grid(((P.pl[p0].W + W - 1) / W + 31) / 32, (P.pl[p0].H + 7) / 8, nframes * npl);
            if (P.bps == 1) launch_degrain_cells<uint8_t>(P.nRefs, W, grid, st, d->dP, d->dJobs, d->dPlan, p0, npl);
            else launch_degrain_cells<uint16_t>(P.nRefs, W, grid, st, d->dP, d->dJobs, d->dPlan, p0, npl);
        }
    } else {
        dim3 grid((P.pl[0].W + 63) / 64, (P.pl[0].H + 3) / 4, nframes * 3);
        if (P.bps == 1) launch_degrain<uint8_t>(P.nRefs, grid, st, d->dP, d->dJobs, d->dPlan);
        else launch_degrain<uint16_t>(P.nRefs, grid, st, d->dP, d->dJobs, d->dPlan);
    }
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}

// MVCompensate.c:419-575 mvcompensateCreate
extern "C" __attribute__((visibility("default"))) int mvx_compensate_create(const mvx_compensate_args *a, const mvx_analysis_data *ad, const mvx_super *sup,
                                     const ptrdiff_t super_pitch[3], const ptrdiff_t dst_pitch[3], mvx_compensate **out, char *err) {
    char dummy[MVX_ERRLEN];
    if (!err) err = dummy;
    err[0] = 0;
    *out = nullptr;
    const mvx_super_info &si = sup->info;
    const int scBehavior = a->scbehavior == MVX_UNSET ? 1 : !!a->scbehavior;
    long long thSAD = a->thsad == MVX_UNSET ? 10000 : a->thsad;
    const double time = a->time;
    if (time < 0.0 || time > 100.0) DFAIL("Compensate: time must be between 0.0 and 100.0 (inclusive).");
    long long nSCD1 = a->thscd1 == MVX_UNSET ? 400 : a->thscd1;
    int nSCD2 = a->thscd2 == MVX_UNSET ? 130 : a->thscd2;
    if (nSCD1 > 8 * 8 * 255) DFAIL("Compensate: thscd1 can be at most %d.", 8 * 8 * 255);
    const long long nSCD1_old = nSCD1;
    int64_t s1 = nSCD1; int32_t s2 = nSCD2;
    mvx_scale_thscd(&s1, &s2, ad);
    nSCD1 = s1; nSCD2 = s2;
    thSAD = thSAD * nSCD1 / nSCD1_old; // :521
    if (ad->nHeight != si.height || ad->nWidth != si.super_width - si.hpad * 2 || ad->nWidth != si.width || ad->nPel != si.pel)
        DFAIL("Compensate: wrong source or super clip frame size.");
    if (a->fields != MVX_UNSET && a->fields && ad->nPel < 2) DFAIL("Compensate: fields option requires pel > 1."); // :514-517
    mvx_compensate *h = new mvx_compensate();
    memset(&h->P, 0, sizeof(h->P));
    int rc = fill_common(h, ad, si, nullptr, super_pitch, dst_pitch, err);
    if (rc) { delete h; return rc; }
    DGParams &P = h->P;
    P.nRefs = 1;
    P.thscd1 = nSCD1; P.thscd2 = nSCD2; P.cthSAD = thSAD;
    P.time256 = (int)(time * 256 / 100); // :560
    P.scBehavior = scBehavior;
    if (!(si.modeYUV & 6)) P.nplanes = 1; // num_planes :147-149
    *out = h;
    return MVX_OK;
}

extern "C" __attribute__((visibility("default"))) void mvx_compensate_destroy(mvx_compensate *c) { delete c; }

extern "C" __attribute__((visibility("default"))) int mvx_compensate_frames(mvx_compensate *c, int nframes, const mvx_compensate_job *jobs, void *stream) {
    if (nframes <= 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    CallGuard::Scope scope(c->guard, st);
    int rc = finish_common(c);
    if (rc) return rc;
    const DGParams &P = c->P;
    if ((rc = ensure_jobs(c, nframes, sizeof(CPlanRec) * (size_t)P.nBlk))) return rc;
    std::vector<DGJob> hj(nframes);
    for (int f = 0; f < nframes; f++) {
        memset(&hj[f], 0, sizeof(DGJob));
        for (int p = 0; p < 3; p++) {
            hj[f].src[p] = (const unsigned char *)jobs[f].src_super[p];
            hj[f].refs[0][p] = (const unsigned char *)jobs[f].ref_super[p];
            hj[f].dst[p] = (unsigned char *)jobs[f].dst[p];
        }
        hj[f].blobs[0] = (const unsigned char *)jobs[f].blob;
        hj[f].fieldShift = jobs[f].field_shift;
    }
    HIP_CHECK(hipMemcpyAsync(c->dJobs, hj.data(), sizeof(DGJob) * nframes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(usable_kernel, dim3(1, nframes), dim3(256), 0, st, c->dP, c->dJobs, c->dUsable, 0);
    hipLaunchKernelGGL(compensate_plan_kernel, dim3((P.nBlk + 255) / 256, nframes), dim3(256), 0, st, c->dP, c->dJobs, c->dUsable, (CPlanRec *)c->dPlan);
    // blocks without overlap: the vectorised row kernel (one launch per plane class); otherwise the per-sample gather
    auto rowsCW = [&](int p) { // samples per thread: up to 16 bytes, dividing block, covered and frame width
        if (P.overlap) return 0;
        const PlaneG &g = P.pl[p];
        int cw = 16 / P.bps;
        while (cw > 1 && (g.blkW % cw || g.W % cw || g.WB % cw)) cw >>= 1;
        return (cw >= 2 && (g.blkW & (g.blkW - 1)) == 0 && (g.blkH & (g.blkH - 1)) == 0) ? cw : 0;
    };
    const int cwY = rowsCW(0), cwC = P.nplanes > 1 ? rowsCW(1) : 1;
    if (cwY && cwC && (P.nplanes == 1 || P.pl[1].W == P.pl[2].W)) {
        for (int cls = 0; cls < (P.nplanes > 1 ? 2 : 1); cls++) {
            const int p0 = cls, npl = cls ? 2 : 1, cw = cls ? cwC : cwY;
            dim3 grid(((P.pl[p0].W / cw) + 63) / 64, (P.pl[p0].H + 3) / 4, nframes * npl);
#define CR(TT, W_) hipLaunchKernelGGL((compensate_rows_kernel<TT, W_>), grid, dim3(256), 0, st, c->dP, c->dJobs, c->dUsable, (const CPlanRec *)c->dPlan, p0, npl)
            if (P.bps == 1) { if (cw == 16) CR(uint8_t, 16); else if (cw == 8) CR(uint8_t, 8); else if (cw == 4) CR(uint8_t, 4); else CR(uint8_t, 2); }
            else { if (cw == 8) CR(uint16_t, 8); else if (cw == 4) CR(uint16_t, 4); else CR(uint16_t, 2); }
#undef CR
        }
    } else {
        dim3 grid((P.pl[0].W + 63) / 64, (P.pl[0].H + 3) / 4, nframes * 3);
        if (P.bps == 1) hipLaunchKernelGGL(compensate_kernel<uint8_t>, grid, dim3(256), 0, st, c->dP, c->dJobs, c->dUsable, (const CPlanRec *)c->dPlan);
        else hipLaunchKernelGGL(compensate_kernel<uint16_t>, grid, dim3(256), 0, st, c->dP, c->dJobs, c->dUsable, (const CPlanRec *)c->dPlan);
    }
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}

// ================================================================================================ mv.BlockFPS
// MVBlockFPS.c:229-676 (frame), :741-1014 (creation); MaskFun.cpp:63-166,349-371; SimpleResize.cpp:27-121.
// One gather pass per output sample like Degrain / Compensate: the two motion-compensated fetches (backward vectors at
// the left frame into the right super frame, forward vectors at the right frame into the left one, both scaled by the
// time position), the per-mode blend, the overlap window sum and ToPixels.  The occlusion / SAD masks live at block
// resolution (three small byte planes per frame pair) and are upsized bilinearly ON THE FLY per sample with the
// reference's integer tables, so no full-size mask planes exist.

struct BFParams {
    int mode, blend, XP, YP;            // padded small-mask grid
    int nWidthP[2], nHeightP[2];        // luma / chroma upsizer output sizes
    double ml;
    long long thscd1; int thscd2;
    int supInterior[3];                 // byte offset of the un-padded level-0 sample (0,0) inside each super plane (:455-464)
    const int *hOff[2], *hW[2], *vOff[2], *vW[2]; // SimpleResize tables, luma / chroma
    long long clipPitch[3];
};
struct BFJob {
    const unsigned char *srcSup[3], *refSup[3]; // super frame nleft / nright
    const unsigned char *blobF, *blobB;         // mvfw vectors at nright, mvbw vectors at nleft
    const unsigned char *clipL[3], *clipR[3];
    unsigned char *dst[3];
    int time256, good;
};
struct BFPlan { unsigned offB[2], offF[2]; };    // luma / chroma offsets of the two compensated blocks

// per job: usable = both vector fields valid and no scene change (Fakery.c:144-146); 0 -> fallback
__global__ __launch_bounds__(256) void bf_usable_kernel(const DGParams *Pp, const BFParams *Bp, const BFJob *jobs, int *usable) {
    const DGParams &P = *Pp; const BFParams &B = *Bp;
    const int f = blockIdx.x;
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const BFJob &J = jobs[f];
    if (J.good) {
        for (int d = 0; d < 2; d++) {
            const GVecD *v = mvx_level0((d ? J.blobB : J.blobF), P.nLvCount);
            int c = 0;
            for (int i = threadIdx.x; i < P.nBlk; i += 256) c += v[i].sad > B.thscd1 ? 1 : 0;
            atomicAdd(&cnt[d], c);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = J.good && J.time256 > 0 && J.time256 < 256;
        if (ok) ok = ((const int *)J.blobF)[1] == 1 && ((const int *)J.blobB)[1] == 1 && !(cnt[0] > B.thscd2) && !(cnt[1] > B.thscd2);
        usable[f] = ok;
    }
}

// small masks, pass 1: scatter-max (occlusion, MaskFun.cpp:91-130) or direct (SAD mask, :139-166) into int planes
// [job][F,B][YP*XP]; the scatter only ever takes maxima, so the order of the reference's loops does not matter.
__global__ __launch_bounds__(256) void bf_mask_kernel(const DGParams *Pp, const BFParams *Bp, const BFJob *jobs, const int *usable, int *small) {
    const DGParams &P = *Pp; const BFParams &B = *Bp;
    const int f = blockIdx.z, dir = blockIdx.y; // dir 0 = forward mask, 1 = backward mask
    if (!usable[f] || B.mode < 3) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.nBlk) return;
    const BFJob &J = jobs[f];
    const GVecD *vec = mvx_level0((dir ? J.blobB : J.blobF), P.nLvCount);
    const int nBlkX = P.nBlkX, nBlkY = P.nBlkY, by = i / nBlkX, bx = i - by * nBlkX;
    const int time256 = dir ? 256 - J.time256 : J.time256;
    const int stepX = P.pl[0].stepX, stepY = P.pl[0].stepY, nPel = P.pel;
    int *m = small + ((size_t)f * 2 + dir) * B.XP * B.YP;
    if (B.mode <= 5) {
        const int tX = time256 * 16 / (stepX * nPel), tY = time256 * 16 / (stepY * nPel);
        const double nX = 80.0 / (B.ml * stepX * nPel), nY = 80.0 / (B.ml * stepY * nPel);
        const int vx = vec[i].x, vy = vec[i].y;
        if (bx < nBlkX - 1) {
            const int vx1 = vec[i + 1].x;
            if (vx1 < vx) {
                const int o = vx - vx1;
                const int minb = dir ? max(0, bx + 1 - o * tX / 4096) : bx;
                const int maxb = dir ? bx + 1 : min(bx + 1 - o * tX / 4096, nBlkX - 1);
                const int val = min((int)(255 * o * nX), 255);
                for (int b = minb; b <= maxb; b++) atomicMax(&m[b + by * B.XP], val);
            }
        }
        if (by < nBlkY - 1) {
            const int vy1 = vec[i + nBlkX].y;
            if (vy1 < vy) {
                const int o = vy - vy1;
                const int minb = dir ? max(0, by + 1 - o * tY / 4096) : by;
                const int maxb = dir ? by + 1 : min(by + 1 - o * tY / 4096, nBlkY - 1);
                const int val = min((int)(255 * o * nY), 255);
                for (int b = minb; b <= maxb; b++) atomicMax(&m[bx + b * B.XP], val);
            }
        }
    } else {
        const int tX = (256 - time256) * 16 / (stepX * nPel), tY = (256 - time256) * 16 / (stepY * nPel);
        int bxi = bx - vec[i].x * tX / 4096, byi = by - vec[i].y * tY / 4096;
        if (bxi < 0 || bxi >= nBlkX || byi < 0 || byi >= nBlkY) { bxi = bx; byi = by; }
        const long long sad = vec[bxi + byi * nBlkX].sad >> (P.bits - 8);
        const double factor = 4.0 / (B.ml * P.pl[0].blkW * P.pl[0].blkH);
        const double l = 255 * ((double)sad * factor); // pow(x, 1.0) == x
        m[bx + by * B.XP] = (int)(unsigned char)((l > 255) ? 255 : l);
    }
}
// pass 2: byte planes [job][F,B,O][YP*XP] with the right / bottom padding clones (MaskFun.cpp:63-80) and O = F*B/255 (:93-101)
__global__ __launch_bounds__(256) void bf_mask_finish_kernel(const DGParams *Pp, const BFParams *Bp, const int *usable, const int *small, unsigned char *masks) {
    const DGParams &P = *Pp; const BFParams &B = *Bp;
    const int f = blockIdx.y;
    if (!usable[f] || B.mode < 3) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B.XP * B.YP) return;
    const int y = i / B.XP, x = i - y * B.XP;
    const int sx = min(x, P.nBlkX - 1), sy = min(y, P.nBlkY - 1); // right clone first, then bottom clone of the padded row
    const int *mF = small + ((size_t)f * 2 + 0) * B.XP * B.YP, *mB = small + ((size_t)f * 2 + 1) * B.XP * B.YP;
    const int vF = mF[sx + sy * B.XP], vB = mB[sx + sy * B.XP];
    unsigned char *o = masks + (size_t)f * 3 * B.XP * B.YP;
    o[i] = (unsigned char)vF;
    o[B.XP * B.YP + i] = (unsigned char)vB;
    o[2 * B.XP * B.YP + i] = (unsigned char)((vF * vB) / 255);
}

__global__ __launch_bounds__(256) void bf_plan_kernel(const DGParams *Pp, const BFJob *jobs, const int *usable, BFPlan *plan) {
    const DGParams &P = *Pp;
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.nBlk || !usable[f]) return;
    const BFJob &J = jobs[f];
    const int by = i / P.nBlkX, bx = i - by * P.nBlkX;
    const GVecD *vB = mvx_level0(J.blobB, P.nLvCount), *vF = mvx_level0(J.blobF, P.nLvCount);
    const int x = bx * P.pl[0].stepX, y = by * P.pl[0].stepY, t = J.time256; // FakeBlockData x / y, Fakery.c:31-32
    const int bX = x * P.pel + ((vB[i].x * (256 - t)) >> 8), bY = y * P.pel + ((vB[i].y * (256 - t)) >> 8);
    const int fX = x * P.pel + ((vF[i].x * t) >> 8), fY = y * P.pel + ((vF[i].y * t) >> 8);
    BFPlan r;
    for (int c = 0; c < 2; c++) { // the reference DIVIDES by the subsampling ratio here (:482-488), it does not shift
        const PlaneG &g = P.pl[c];
        const int xr = 1 << g.subX, yr = 1 << g.subY;
        r.offB[c] = sup_offset(g, P.pel, P.logPel, P.bps, bX / xr, bY / yr);
        r.offF[c] = sup_offset(g, P.pel, P.logPel, P.bps, fX / xr, fY / yr);
    }
    plan[(size_t)f * P.nBlk + i] = r;
}

// SimpleResize.cpp:62-121 at one output sample
__device__ __forceinline__ int bf_upsize(const unsigned char *m, int XP, const int *hOff, const int *hW, const int *vOff, const int *vW, int x, int y) {
    const int wb = vW[y], wt = 16384 - wb, o = hOff[x], wr = hW[x], wl = 16384 - wr;
    const unsigned char *s1 = m + vOff[y] * XP, *s2 = s1 + XP;
    const int a = (unsigned char)((s1[o] * wt + s2[o] * wb + 8192) >> 14), b = (unsigned char)((s1[o + 1] * wt + s2[o + 1] * wb + 8192) >> 14);
    return (unsigned char)((a * wl + b * wr + 8192) >> 14);
}
__device__ __forceinline__ int bf_median(int a, int b, int c) { const int mn = min(a, b), mx = max(a, b); return max(mn, min(mx, c)); }

template <typename T>
__global__ __launch_bounds__(256) void blockfps_kernel(const DGParams *Pp, const BFParams *Bp, const BFJob *jobs, const int *usable, const BFPlan *plan, const unsigned char *masks) {
    const DGParams &P = *Pp; const BFParams &B = *Bp;
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= P.nplanes) return;
    const PlaneG &g = P.pl[p];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= g.H) return;
    const BFJob &J = jobs[f];
    T *drow = (T *)(J.dst[p] + (long long)y * g.dstPitch);
    const int t = J.time256;
    if (!usable[f]) { // time256 0 / 256, vectors unusable or frames outside the clip (:285-288, :640-673)
        const int l = ((const T *)(J.clipL[p] + (long long)y * B.clipPitch[p]))[x];
        if (t <= 0 || (t < 256 && !B.blend)) { drow[x] = (T)l; return; }
        const int r = ((const T *)(J.clipR[p] + (long long)y * B.clipPitch[p]))[x];
        drow[x] = t >= 256 ? (T)r : (T)((l * (256 - t) + r * t) >> 8);
        return;
    }
    const int sVal = ((const T *)(J.srcSup[p] + B.supInterior[p] + (long long)y * g.supPitch))[x];
    const int rVal = ((const T *)(J.refSup[p] + B.supInterior[p] + (long long)y * g.supPitch))[x];
    const int covW = P.overlap ? g.WB : g.blkW * P.nBlkX, covH = P.overlap ? g.HB : g.blkH * P.nBlkY;
    if (x >= covW || y >= covH) { drow[x] = (T)((sVal * (256 - t) + rVal * t) >> 8); return; } // Blend of the uncovered strips
    const int mode = B.mode, c = p ? 1 : 0;
    int mF = 0, mB = 0, mO = 0;
    if (mode >= 3) {
        const unsigned char *m = masks + (size_t)f * 3 * B.XP * B.YP;
        if (mode != 5 && mode != 8) {
            mF = bf_upsize(m, B.XP, B.hOff[c], B.hW[c], B.vOff[c], B.vW[c], x, y);
            mB = bf_upsize(m + B.XP * B.YP, B.XP, B.hOff[c], B.hW[c], B.vOff[c], B.vW[c], x, y);
        }
        if (mode == 4 || mode == 5 || mode == 7 || mode == 8) mO = bf_upsize(m + 2 * B.XP * B.YP, B.XP, B.hOff[c], B.hW[c], B.vOff[c], B.vW[c], x, y);
    }
    const BFPlan *pl = plan + (size_t)f * P.nBlk;
    auto result = [&](const BFPlan &R, int px, int py) -> int { // RealResultBlock, MVBlockFPS.c:117-227
        const int b = ((const T *)(J.refSup[p] + R.offB[c] + (long long)py * g.supPitch))[px];
        const int fw = ((const T *)(J.srcSup[p] + R.offF[c] + (long long)py * g.supPitch))[px];
        switch (mode) {
        case 0: return (b * t + fw * (256 - t)) >> 8;
        case 1: return bf_median(rVal, sVal, (int)(T)((b * t + fw * (256 - t)) >> 8));
        case 2: return bf_median((int)(T)((rVal * t + sVal * (256 - t)) >> 8), b, fw);
        case 3: case 6: return (((mB * fw + (255 - mB) * b + 255) >> 8) * t + ((mF * b + (255 - mF) * fw + 255) >> 8) * (256 - t)) >> 8;
        case 4: case 7: {
            const int ff = (mF * b + (255 - mF) * fw + 255) >> 8, bb = (mB * fw + (255 - mB) * b + 255) >> 8;
            const int avg = (rVal * t + sVal * (256 - t) + 255) >> 8, m = (bb * t + ff * (256 - t)) >> 8;
            return (avg * mO + m * (255 - mO) + 255) >> 8;
        }
        default: return mO << (P.bits - 8);
        }
    };
    int out;
    if (!P.overlap) {
        const int bx = x / g.blkW, by = y / g.blkH;
        out = (T)result(pl[by * P.nBlkX + bx], x - bx * g.blkW, y - by * g.blkH);
    } else {
        int bx1 = x / g.stepX; if (bx1 > P.nBlkX - 1) bx1 = P.nBlkX - 1;
        const int bx0 = x - g.blkW + 1 <= 0 ? 0 : (x - g.blkW + g.stepX) / g.stepX;
        int by1 = y / g.stepY; if (by1 > P.nBlkY - 1) by1 = P.nBlkY - 1;
        const int by0 = y - g.blkH + 1 <= 0 ? 0 : (y - g.blkH + g.stepY) / g.stepY;
        unsigned acc = 0;
        const int16_t *win = P.win[p];
        for (int by = by0; by <= by1; by++) {
            const int py = y - by * g.stepY;
            const int wby = by == 0 ? 0 : (by == P.nBlkY - 1 ? 6 : 3);
            for (int bx = bx0; bx <= bx1; bx++) {
                const int px = x - bx * g.stepX;
                const int wbx = bx == P.nBlkX - 1 ? 2 : (bx == 0 ? 0 : 1);
                const int val = (T)result(pl[by * P.nBlkX + bx], px, py);
                acc += (unsigned)((val * (int)win[(wby + wbx) * g.blkW * g.blkH + py * g.blkW + px]) >> 6);
            }
        }
        if (sizeof(T) == 1) acc &= 0xffffu;
        const int a = (int)((acc + 16) >> 5);
        const int pm = (1 << P.bits) - 1;
        out = a > pm ? pm : a;
    }
    drow[x] = (T)out;
}


// ---- blocks without overlap: one thread per CW consecutive samples of one block row (same conditions as compensate_rows_kernel).
// The four vector loads (left / right frame at the sample, the two motion-compensated fetches) and the plan record are per segment;
// the mask upsizer (SimpleResize.cpp:62-121) interpolates vertically once per mask column the segment touches.
template <typename T, int CW>
__global__ __launch_bounds__(256) void blockfps_rows_kernel(const DGParams *Pp, const BFParams *Bp, const BFJob *jobs, const int *usable, const BFPlan *plan, const unsigned char *masks,
                                                            int planeFirst, int planesPerFrame) {
    const DGParams &P = *Pp; const BFParams &B = *Bp;
    const int z = blockIdx.z, f = z / planesPerFrame, p = planeFirst + z % planesPerFrame;
    const PlaneG &g = P.pl[p];
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * CW, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= g.H) return;
    const BFJob &J = jobs[f];
    unsigned char *dptr = J.dst[p] + (long long)y * g.dstPitch + (long long)x * (long long)sizeof(T);
    const int t = J.time256;
    int out[CW];
    if (!usable[f]) { // time256 0 / 256, vectors unusable or frames outside the clip (:285-288, :640-673)
        int l[CW];
        dg_load<T, CW>(dg_gl(J.clipL[p] + (long long)y * B.clipPitch[p] + (long long)x * (long long)sizeof(T)), l);
        if (t <= 0 || (t < 256 && !B.blend)) { dg_store<T, CW>(dg_glw(dptr), l); return; }
        int r[CW];
        dg_load<T, CW>(dg_gl(J.clipR[p] + (long long)y * B.clipPitch[p] + (long long)x * (long long)sizeof(T)), r);
#pragma unroll
        for (int i = 0; i < CW; i++) out[i] = t >= 256 ? r[i] : (int)(T)((l[i] * (256 - t) + r[i] * t) >> 8);
        dg_store<T, CW>(dg_glw(dptr), out);
        return;
    }
    int sVal[CW], rVal[CW];
    const long long inner = B.supInterior[p] + (long long)y * g.supPitch + (long long)x * (long long)sizeof(T);
    dg_load<T, CW>(dg_gl(J.srcSup[p] + inner), sVal);
    dg_load<T, CW>(dg_gl(J.refSup[p] + inner), rVal);
    const int covW = g.blkW * P.nBlkX, covH = g.blkH * P.nBlkY;
    if (x >= covW || y >= covH) { // Blend of the uncovered strips
#pragma unroll
        for (int i = 0; i < CW; i++) out[i] = (int)(T)((sVal[i] * (256 - t) + rVal[i] * t) >> 8);
        dg_store<T, CW>(dg_glw(dptr), out);
        return;
    }
    const int mode = B.mode, c = p ? 1 : 0;
    int mF[CW], mB[CW], mO[CW];
#pragma unroll
    for (int i = 0; i < CW; i++) { mF[i] = 0; mB[i] = 0; mO[i] = 0; }
    if (mode >= 3) {
        DG_GL const unsigned char *m = dg_gl(masks + (size_t)f * 3 * B.XP * B.YP);
        // (r4: the resize tables and the mask bytes through global-address-space pointers and as vectors -- the table pointers sit in a struct, so the
        // compiler emitted one FLAT load per table entry, each waited for on its own, and six byte loads per mask: 45 loads per thread, now 15)
        const int wb = *(DG_GL const int *)dg_gl(B.vW[c] + y), wt = 16384 - wb;
        const int rowOff = *(DG_GL const int *)dg_gl(B.vOff[c] + y) * B.XP;
        int hO[CW], hWr[CW];
        if (x + CW <= g.W) {
            dg_load_ints<CW>(dg_gl(B.hOff[c] + x), hO);
            dg_load_ints<CW>(dg_gl(B.hW[c] + x), hWr);
        } else {
#pragma unroll
            for (int i = 0; i < CW; i++) { const int xi = x + i < g.W ? x + i : g.W - 1; hO[i] = *(DG_GL const int *)dg_gl(B.hOff[c] + xi); hWr[i] = *(DG_GL const int *)dg_gl(B.hW[c] + xi); }
        }
        const int o0 = hO[0];
        // vertically interpolated values of up to three mask columns; columns further right (upsizing by less than CW) are done on demand
        auto vcol = [&](DG_GL const unsigned char *mm, int o) { return (int)(unsigned char)((mm[rowOff + o] * wt + mm[rowOff + B.XP + o] * wb + 8192) >> 14); };
        auto upsize = [&](DG_GL const unsigned char *mm, int *dst) {
            // (the third column is only used when a sample's left neighbour is o0 + 1; at the plane's right edge it does not exist: read o0 + 1 again)
            // the bytes o0 .. o0 + 3 of the two mask rows as ONE unaligned dword each (the mask buffer has four bytes of slack behind its last row)
            const unsigned r0 = *(DG_GL const dg_uv1 *)(mm + rowOff + o0), r1 = *(DG_GL const dg_uv1 *)(mm + rowOff + B.XP + o0);
            auto vc = [&](int k) { return (int)(unsigned char)((((r0 >> (8 * k)) & 0xffu) * wt + ((r1 >> (8 * k)) & 0xffu) * wb + 8192) >> 14); };
            const int a0 = vc(0), a1 = vc(1), a2 = o0 + 2 < B.XP ? vc(2) : a1;
#pragma unroll
            for (int i = 0; i < CW; i++) {
                const int o = hO[i], wr = hWr[i], wl = 16384 - wr, k = o - o0;
                int a, b;
                if (k == 0) { a = a0; b = a1; } else if (k == 1) { a = a1; b = a2; } else { a = vcol(mm, o); b = vcol(mm, o + 1); }
                dst[i] = (int)(unsigned char)((a * wl + b * wr + 8192) >> 14);
            }
        };
        if (mode != 5 && mode != 8) { upsize(m, mF); upsize(m + B.XP * B.YP, mB); }
        if (mode == 4 || mode == 5 || mode == 7 || mode == 8) upsize(m + 2 * B.XP * B.YP, mO);
    }
    const int lbw = __ffs(g.blkW) - 1, lbh = __ffs(g.blkH) - 1;
    const int bx = x >> lbw, by = y >> lbh, px = x & (g.blkW - 1), py = y & (g.blkH - 1);
    const BFPlan R = plan[(size_t)f * P.nBlk + by * P.nBlkX + bx];
    int bv[CW], fv[CW];
    const long long bo = (long long)py * g.supPitch + (long long)px * (long long)sizeof(T);
    dg_load<T, CW>(dg_gl(J.refSup[p] + R.offB[c] + bo), bv);
    dg_load<T, CW>(dg_gl(J.srcSup[p] + R.offF[c] + bo), fv);
#pragma unroll
    for (int i = 0; i < CW; i++) { // RealResultBlock, MVBlockFPS.c:117-227
        const int b = bv[i], fw = fv[i];
        int v;
        switch (mode) {
        case 0: v = (b * t + fw * (256 - t)) >> 8; break;
        case 1: v = bf_median(rVal[i], sVal[i], (int)(T)((b * t + fw * (256 - t)) >> 8)); break;
        case 2: v = bf_median((int)(T)((rVal[i] * t + sVal[i] * (256 - t)) >> 8), b, fw); break;
        case 3: case 6: v = (((mB[i] * fw + (255 - mB[i]) * b + 255) >> 8) * t + ((mF[i] * b + (255 - mF[i]) * fw + 255) >> 8) * (256 - t)) >> 8; break;
        case 4: case 7: {
            const int ff = (mF[i] * b + (255 - mF[i]) * fw + 255) >> 8, bb = (mB[i] * fw + (255 - mB[i]) * b + 255) >> 8;
            const int avg = (rVal[i] * t + sVal[i] * (256 - t) + 255) >> 8, mm = (bb * t + ff * (256 - t)) >> 8;
            v = (avg * mO[i] + mm * (255 - mO[i]) + 255) >> 8; break;
        }
        default: v = mO[i] << (P.bits - 8); break;
        }
        out[i] = (int)(T)v;
    }
    dg_store<T, CW>(dg_glw(dptr), out);
}

struct mvx_blockfps : DGCommon {
    BFParams B;
    BFParams *dB = nullptr;
    BFJob *dBJobs = nullptr;
    size_t bjobsCap = 0;
    int *dSmall = nullptr; unsigned char *dMasks = nullptr; size_t maskCap = 0;
    int *dTables = nullptr;
    mvx_analysis_data bw, fw;
    long long fa, fb, outNum, outDen;
    int inFrames, outFrames;
    ~mvx_blockfps() {
        if (dB) (void)hipFree(dB);
        if (dBJobs) (void)hipFree(dBJobs);
        if (dSmall) (void)hipFree(dSmall);
        if (dMasks) (void)hipFree(dMasks);
        if (dTables) (void)hipFree(dTables);
    }
};

// SimpleResize.cpp:27-57 InitTables (same float arithmetic)
static void bf_tables(int *offsets, int *weights, int out, int in) {
    const float leftmost = 0.5f, rightmost = in - 0.5f;
    const int leftmost_idx = std::max((int)leftmost, 0), rightmost_idx = std::min((int)rightmost, in - 1);
    for (int i = 0; i < out; i++) {
        const float position = (i + 0.5f) * (float)in / (float)out;
        float weight; int offset;
        if (position <= leftmost) { offset = leftmost_idx; weight = 0.0f; }
        else if (position >= rightmost) { offset = rightmost_idx - 1; weight = 1.0f; }
        else { offset = (int)(position - leftmost); weight = position - leftmost - offset; }
        offsets[i] = offset;
        weights[i] = (int)(weight * 16384);
    }
}
static long long bf_gcd(long long x, long long y) { while (y) { long long t = x % y; x = y; y = t; } return x; }

extern "C" __attribute__((visibility("default"))) int mvx_blockfps_create(const mvx_blockfps_args *a, const mvx_analysis_data *bw, const mvx_analysis_data *fw,
        const mvx_super *sup, int num_frames, int64_t fps_num, int64_t fps_den, const ptrdiff_t super_pitch[3], const ptrdiff_t clip_pitch[3],
        const ptrdiff_t dst_pitch[3], mvx_blockfps **out, char *err) {
    char dummy[MVX_ERRLEN];
    if (!err) err = dummy;
    err[0] = 0;
    *out = nullptr;
    const mvx_super_info &si = sup->info;
    const long long num = a->num == MVX_UNSET ? 25 : a->num, den = a->den == MVX_UNSET ? 1 : a->den;
    const int mode = a->mode == MVX_UNSET ? 3 : a->mode;
    const int blend = a->blend == MVX_UNSET ? 1 : !!a->blend;
    long long thscd1 = a->thscd1 == MVX_UNSET ? 400 : a->thscd1;
    int thscd2 = a->thscd2 == MVX_UNSET ? 130 : a->thscd2;
    if (mode < 0 || mode > 8) DFAIL("BlockFPS: mode must be between 0 and 8 (inclusive).");
    if (thscd1 > 8 * 8 * 255) DFAIL("BlockFPS: thscd1 can be at most %d.", 8 * 8 * 255);
    { int64_t s1 = thscd1; int32_t s2 = thscd2; mvx_scale_thscd(&s1, &s2, bw); thscd1 = s1; thscd2 = s2; }
    if (bw->nWidth != fw->nWidth) DFAIL("BlockFPS: mvbw and mvfw have different widths.");
    if (bw->nHeight != fw->nHeight) DFAIL("BlockFPS: mvbw and mvfw have different heights.");
    if (bw->nBlkSizeX != fw->nBlkSizeX || bw->nBlkSizeY != fw->nBlkSizeY) DFAIL("BlockFPS: mvbw and mvfw have different block sizes.");
    if (bw->nPel != fw->nPel) DFAIL("BlockFPS: mvbw and mvfw have different pel precision.");
    if (bw->nOverlapX != fw->nOverlapX || bw->nOverlapY != fw->nOverlapY) DFAIL("BlockFPS: mvbw and mvfw have different overlap.");
    if (bw->nDeltaFrame <= 0 || fw->nDeltaFrame <= 0) DFAIL("BlockFPS: cannot use motion vectors with absolute frame references.");
    if (bw->nDeltaFrame != fw->nDeltaFrame) DFAIL("BlockFPS: mvbw and mvfw must be generated with the same delta.");
    if (!bw->isBackward) DFAIL("BlockFPS: mvbw must be generated with isb=True.");
    if (fw->isBackward) DFAIL("BlockFPS: mvfw must be generated with isb=False.");
    if (fps_num == 0 || fps_den == 0) DFAIL("BlockFPS: The input clip must have a frame rate. Invoke AssumeFPS if necessary.");
    long long numerator, denominator;
    if (num != 0 && den != 0) { numerator = num; denominator = den; } else { numerator = fps_num * 2; denominator = fps_den; }
    if (bw->nHeight != si.height || bw->nWidth != si.super_width - si.hpad * 2 || bw->nWidth != si.width || bw->nPel != si.pel)
        DFAIL("BlockFPS: wrong source or super clip frame size.");
    mvx_blockfps *h = new mvx_blockfps();
    memset(&h->P, 0, sizeof(h->P));
    memset(&h->B, 0, sizeof(h->B));
    int rc = fill_common(h, bw, si, clip_pitch, super_pitch, dst_pitch, err);
    if (rc) { delete h; return rc; }
    DGParams &P = h->P;
    if (!(si.modeYUV & 6)) P.nplanes = 1;
    P.nRefs = 2; P.thscd1 = thscd1; P.thscd2 = thscd2;
    h->bw = *bw; h->fw = *fw;
    h->fa = denominator * fps_num; h->fb = numerator * fps_den;
    const long long g = bf_gcd(h->fa, h->fb);
    h->fa /= g; h->fb /= g;
    if (numerator <= 0 || denominator <= 0) { h->outNum = 0; h->outDen = 1; }
    else { const long long x = bf_gcd(numerator, denominator); h->outNum = numerator / x; h->outDen = denominator / x; }
    h->inFrames = num_frames;
    h->outFrames = (int)(1 + (num_frames - 1) * h->fb / h->fa);
    BFParams &B = h->B;
    B.mode = mode; B.blend = blend; B.ml = a->ml; B.thscd1 = thscd1; B.thscd2 = thscd2;
    B.XP = bw->nBlkX; B.YP = bw->nBlkY;
    while (B.XP * (bw->nBlkSizeX - bw->nOverlapX) + bw->nOverlapX < bw->nWidth) B.XP++;
    while (B.YP * (bw->nBlkSizeY - bw->nOverlapY) + bw->nOverlapY < bw->nHeight) B.YP++;
    B.nWidthP[0] = B.XP * (bw->nBlkSizeX - bw->nOverlapX) + bw->nOverlapX;
    B.nHeightP[0] = B.YP * (bw->nBlkSizeY - bw->nOverlapY) + bw->nOverlapY;
    B.nWidthP[1] = B.nWidthP[0] / bw->xRatioUV; B.nHeightP[1] = B.nHeightP[0] / bw->yRatioUV;
    const int bps = P.bps;
    B.supInterior[0] = si.hpad * bps + (int)super_pitch[0] * si.vpad;
    for (int p = 1; p < 3; p++) B.supInterior[p] = (si.hpad >> 1) * bps + (int)super_pitch[p < si.num_planes ? p : 0] * (si.vpad >> 1); // the reference's ">> 1" (:459-463)
    for (int p = 0; p < 3; p++) B.clipPitch[p] = clip_pitch[p < si.num_planes ? p : 0];
    *out = h;
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) void mvx_blockfps_destroy(mvx_blockfps *b) { delete b; }
extern "C" __attribute__((visibility("default"))) void mvx_blockfps_get_info(const mvx_blockfps *b, mvx_blockfps_info *info) {
    info->num_frames = b->outFrames; info->fps_num = b->outNum; info->fps_den = b->outDen;
}
// MVBlockFPS.c:245-254,278-292
extern "C" __attribute__((visibility("default"))) void mvx_blockfps_map(const mvx_blockfps *b, int n, int *nleft, int *nright, int *time256) {
    const int off = b->bw.nDeltaFrame;
    *nleft = (int)(n * b->fa / b->fb);
    int t = (int)(((double)n * b->fa / b->fb - *nleft) * 256 + 0.5);
    if (off > 1) t = t / off;
    *nright = *nleft + off;
    *time256 = t;
}

extern "C" __attribute__((visibility("default"))) int mvx_blockfps_frames(mvx_blockfps *b, int nframes, const mvx_blockfps_job *jobs, void *stream) {
    if (nframes <= 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    CallGuard::Scope scope(b->guard, st);
    int rc = finish_common(b);
    if (rc) return rc;
    const DGParams &P = b->P;
    BFParams &B = b->B;
    if (!b->dB) { // upsizer tables + parameter block
        const int n = B.nWidthP[0] + B.nWidthP[1] + B.nHeightP[0] + B.nHeightP[1];
        std::vector<int> t(2 * n);
        int *o = t.data(), *w = t.data() + n, pos = 0;
        HIP_CHECK(hipMalloc((void **)&b->dTables, sizeof(int) * 2 * n));
        for (int c = 0; c < 2; c++) {
            bf_tables(o + pos, w + pos, B.nWidthP[c], B.XP); B.hOff[c] = b->dTables + pos; B.hW[c] = b->dTables + n + pos; pos += B.nWidthP[c];
            bf_tables(o + pos, w + pos, B.nHeightP[c], B.YP); B.vOff[c] = b->dTables + pos; B.vW[c] = b->dTables + n + pos; pos += B.nHeightP[c];
        }
        HIP_CHECK(hipMemcpy(b->dTables, t.data(), sizeof(int) * 2 * n, hipMemcpyHostToDevice));
        HIP_CHECK(hipMalloc((void **)&b->dB, sizeof(BFParams)));
        HIP_CHECK(hipMemcpy(b->dB, &B, sizeof(BFParams), hipMemcpyHostToDevice));
    }
    if ((rc = ensure_jobs(b, nframes, sizeof(BFPlan) * (size_t)P.nBlk))) return rc;
    if ((size_t)nframes > b->bjobsCap) {
        if (b->dBJobs) (void)hipFree(b->dBJobs);
        b->bjobsCap = (size_t)nframes * 2;
        HIP_CHECK(hipMalloc((void **)&b->dBJobs, b->bjobsCap * sizeof(BFJob)));
    }
    const size_t cells = (size_t)B.XP * B.YP;
    if ((size_t)nframes > b->maskCap) {
        if (b->dSmall) (void)hipFree(b->dSmall);
        if (b->dMasks) (void)hipFree(b->dMasks);
        b->maskCap = (size_t)nframes * 2;
        HIP_CHECK(hipMalloc((void **)&b->dSmall, b->maskCap * 2 * cells * sizeof(int)));
        HIP_CHECK(hipMalloc((void **)&b->dMasks, b->maskCap * 3 * cells + 16)); // (+ slack: the row kernels read mask bytes four at a time)
    }
    std::vector<BFJob> hj(nframes);
    for (int f = 0; f < nframes; f++) {
        BFJob &j = hj[f];
        memset(&j, 0, sizeof(j));
        for (int p = 0; p < 3; p++) {
            j.srcSup[p] = (const unsigned char *)jobs[f].src_super[p]; j.refSup[p] = (const unsigned char *)jobs[f].ref_super[p];
            j.clipL[p] = (const unsigned char *)jobs[f].clip_left[p]; j.clipR[p] = (const unsigned char *)jobs[f].clip_right[p];
            j.dst[p] = (unsigned char *)jobs[f].dst[p];
        }
        j.blobF = (const unsigned char *)jobs[f].blob_fw; j.blobB = (const unsigned char *)jobs[f].blob_bw;
        j.time256 = jobs[f].time256;
        j.good = j.srcSup[0] && j.refSup[0] && j.blobF && j.blobB;
        if (!j.clipL[0] || (j.time256 > 0 && !j.clipR[0])) { mvx_set_error("mvx_blockfps_frames: clip_left / clip_right are required"); return MVX_E_ARG; }
    }
    HIP_CHECK(hipMemcpyAsync(b->dBJobs, hj.data(), sizeof(BFJob) * nframes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bf_usable_kernel, dim3(nframes), dim3(256), 0, st, b->dP, b->dB, b->dBJobs, b->dUsable);
    if (B.mode >= 3) {
        HIP_CHECK(hipMemsetAsync(b->dSmall, 0, (size_t)nframes * 2 * cells * sizeof(int), st));
        hipLaunchKernelGGL(bf_mask_kernel, dim3((P.nBlk + 255) / 256, 2, nframes), dim3(256), 0, st, b->dP, b->dB, b->dBJobs, b->dUsable, b->dSmall);
        hipLaunchKernelGGL(bf_mask_finish_kernel, dim3((unsigned)((cells + 255) / 256), nframes), dim3(256), 0, st, b->dP, b->dB, b->dUsable, b->dSmall, b->dMasks);
    }
    hipLaunchKernelGGL(bf_plan_kernel, dim3((P.nBlk + 255) / 256, nframes), dim3(256), 0, st, b->dP, b->dBJobs, b->dUsable, (BFPlan *)b->dPlan);
    auto rowsCW = [&](int p) { // as in mvx_compensate_frames; the uncovered strips start at nBlkX * blkW here
        if (P.overlap) return 0;
        const PlaneG &g = P.pl[p];
        int cw = 16 / P.bps;
        while (cw > 1 && (g.blkW % cw || g.W % cw)) cw >>= 1;
        return (cw >= 2 && (g.blkW & (g.blkW - 1)) == 0 && (g.blkH & (g.blkH - 1)) == 0) ? cw : 0;
    };
    const int cwY = rowsCW(0), cwC = P.nplanes > 1 ? rowsCW(1) : 1;
    if (cwY && cwC && (P.nplanes == 1 || P.pl[1].W == P.pl[2].W)) {
        for (int cls = 0; cls < (P.nplanes > 1 ? 2 : 1); cls++) {
            const int p0 = cls, npl = cls ? 2 : 1, cw = cls ? cwC : cwY;
            dim3 grid(((P.pl[p0].W / cw) + 63) / 64, (P.pl[p0].H + 3) / 4, nframes * npl);
#define BR(TT, W_) hipLaunchKernelGGL((blockfps_rows_kernel<TT, W_>), grid, dim3(256), 0, st, b->dP, b->dB, b->dBJobs, b->dUsable, (const BFPlan *)b->dPlan, b->dMasks, p0, npl)
            if (P.bps == 1) { if (cw == 16) BR(uint8_t, 16); else if (cw == 8) BR(uint8_t, 8); else if (cw == 4) BR(uint8_t, 4); else BR(uint8_t, 2); }
            else { if (cw == 8) BR(uint16_t, 8); else if (cw == 4) BR(uint16_t, 4); else BR(uint16_t, 2); }
#undef BR
        }
    } else {
        dim3 grid((P.pl[0].W + 63) / 64, (P.pl[0].H + 3) / 4, nframes * 3);
        if (P.bps == 1) hipLaunchKernelGGL(blockfps_kernel<uint8_t>, grid, dim3(256), 0, st, b->dP, b->dB, b->dBJobs, b->dUsable, (const BFPlan *)b->dPlan, b->dMasks);
        else hipLaunchKernelGGL(blockfps_kernel<uint16_t>, grid, dim3(256), 0, st, b->dP, b->dB, b->dBJobs, b->dUsable, (const BFPlan *)b->dPlan, b->dMasks);
    }
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}

// ================================================================================================ mv.SCDetection
// MVSCDetection.c:43-73: _SceneChangePrev / _SceneChangeNext = !fgopIsUsable(vectors at n) (Fakery.c:52-58,144-146)
__global__ __launch_bounds__(256) void scdetect_kernel(const unsigned char *const *blobs, int nLvCount, int nBlk, long long thscd1, int thscd2, int *out) {
    const unsigned char *blob = blobs[blockIdx.x];
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const GVecD *v = mvx_level0(blob, nLvCount);
    int c = 0;
    for (int i = threadIdx.x; i < nBlk; i += 256) c += v[i].sad > thscd1 ? 1 : 0;
    atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = !(((const int *)blob)[1] == 1 && !(cnt > thscd2));
}

extern "C" __attribute__((visibility("default"))) int mvx_scdetect(const mvx_analysis_data *ad, int64_t thscd1, int32_t thscd2, int n, const void *const *blobs,
                                                                   int32_t *scene_change, void *stream, char *err) {
    char dummy[MVX_ERRLEN];
    if (!err) err = dummy;
    err[0] = 0;
    if (n <= 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    int64_t s1 = thscd1 == MVX_UNSET ? 400 : thscd1;
    int32_t s2 = thscd2 == MVX_UNSET ? 130 : thscd2;
    if (s1 > 8 * 8 * 255) DFAIL("SCDetection: thscd1 can be at most %d.", 8 * 8 * 255); // MVAnalysisData.c:11-14
    mvx_scale_thscd(&s1, &s2, ad);
    const unsigned char **dB = nullptr; int *dOut = nullptr;
    HIP_CHECK(hipMalloc((void **)&dB, sizeof(void *) * n));
    if (hipMalloc((void **)&dOut, sizeof(int) * n) != hipSuccess) { (void)hipFree(dB); mvx_set_error("mvx_scdetect: out of device memory"); return MVX_E_NOMEM; }
    hipError_t e = hipMemcpyAsync(dB, blobs, sizeof(void *) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(scdetect_kernel, dim3(n), dim3(256), 0, st, dB, ad->nLvCount, ad->nBlkX * ad->nBlkY, (long long)s1, (int)s2, dOut);
        e = hipMemcpyAsync(scene_change, dOut, sizeof(int) * n, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dB); (void)hipFree(dOut);
    if (e != hipSuccess) { mvx_set_error("mvx_scdetect: %s", hipGetErrorString(e)); return MVX_E_DEVICE; }
    return MVX_OK;
}

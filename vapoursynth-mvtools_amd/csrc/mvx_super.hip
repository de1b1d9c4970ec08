// mvx_super.hip -- mv.Super on gfx950: padded hierarchical pyramid + sub-pel planes.
//
// What it computes is fixed by the reference (MVSuper.c:43-126, MVFrame.cpp:508-1197,1264-1318,1386-1527,1634-1683);
// how it is computed is not: instead of fill -> reduce chain -> pad -> three whole-plane refine passes, every output
// sample is produced directly from the source frame:
//   * level 0: one LDS-tiled kernel reads the source tile (+2/+3 halo, coordinates clamped = the reference's edge
//     replication) once and writes the padded plane and its H / V / HV half-pel planes (pel 2; pel 4 adds averages);
//   * level L+1: one kernel per level evaluates the separable 2x decimation filter at the clamped interior coordinate
//     of every sample of the padded rectangle (the padding is just the clamped coordinate, no separate pad pass).
// Both are HBM streaming kernels: algorithmic bytes = S_src read + S_super written.
#include "mvx_common.h"

// ------------------------------------------------------------------------------------------------ host: geometry

int mvx_plane_height_luma(int src_height, int level, int yRatioUV, int vpad) { // MVFrame.cpp:1209-1216
    int height = src_height;
    for (int i = 1; i <= level; i++)
        height = vpad >= yRatioUV ? ((height / yRatioUV + 1) / 2) * yRatioUV : ((height / yRatioUV) / 2) * yRatioUV;
    return height;
}

int mvx_plane_width_luma(int src_width, int level, int xRatioUV, int hpad) { // MVFrame.cpp:1219-1226
    int width = src_width;
    for (int i = 1; i <= level; i++)
        width = hpad >= xRatioUV ? ((width / xRatioUV + 1) / 2) * xRatioUV : ((width / xRatioUV) / 2) * xRatioUV;
    return width;
}

unsigned mvx_plane_super_offset(int chroma, int src_height, int level, int pel, int vpad, int plane_pitch, int yRatioUV) { // :1229-1247
    int height = src_height;
    unsigned offset = 0;
    if (level > 0) {
        offset = pel * pel * plane_pitch * (src_height + vpad * 2);
        for (int i = 1; i < level; i++) {
            height = chroma ? mvx_plane_height_luma(src_height * yRatioUV, i, yRatioUV, vpad * yRatioUV) / yRatioUV
                            : mvx_plane_height_luma(src_height, i, yRatioUV, vpad);
            offset += plane_pitch * (height + vpad * 2);
        }
    }
    return offset;
}

void mvx_level_plane(const mvx_super_info &si, int level, int plane, long long pitch, LevelPlane *o) {
    int xr = plane ? si.xRatioUV : 1, yr = plane ? si.yRatioUV : 1;
    o->w = mvx_plane_width_luma(si.width, level, si.xRatioUV, si.hpad) / xr;   // mvgofInit MVFrame.cpp:1871-1877 + mvfInit :1769-1779
    o->h = mvx_plane_height_luma(si.height, level, si.yRatioUV, si.vpad) / yr;
    o->hpad = si.hpad / xr;
    o->vpad = si.vpad / yr;
    o->pw = o->w + 2 * o->hpad;
    o->ph = o->h + 2 * o->vpad;
    // mvgofUpdate MVFrame.cpp:1898: the plane index is passed as the `chroma` flag, level-0 plane height/vpad
    o->off = (long long)mvx_plane_super_offset(plane, si.height / yr, level, si.pel, o->vpad, 1, si.yRatioUV) * pitch;
}

static int argdef(int v, int d) { return v == MVX_UNSET ? d : v; }

extern "C" __attribute__((visibility("default"))) int mvx_super_create(const mvx_super_args *a, mvx_super **out, char *err) { // MVSuper.c:140-264
    char dummy[MVX_ERRLEN];
    if (!err) err = dummy;
    err[0] = 0;
    *out = nullptr;
    mvx_super_info si;
    memset(&si, 0, sizeof(si));
    si.hpad = argdef(a->hpad, 16);
    si.vpad = argdef(a->vpad, 16);
    si.pel = argdef(a->pel, 2);
    si.levels = argdef(a->levels, 0);
    si.chroma = !!argdef(a->chroma, 1);
    si.sharp = argdef(a->sharp, 2);
    si.rfilter = argdef(a->rfilter, 2);
#define SFAIL(msg) do { snprintf(err, MVX_ERRLEN, "%s", msg); mvx_set_error("%s", msg); return MVX_E_ARG; } while (0)
    if (si.pel != 1 && si.pel != 2 && si.pel != 4) SFAIL("Super: pel must be 1, 2, or 4.");
    if (si.sharp < 0 || si.sharp > 2) SFAIL("Super: sharp must be between 0 and 2 (inclusive).");
    if (si.rfilter < 0 || si.rfilter > 4) SFAIL("Super: rfilter must be between 0 and 4 (inclusive).");
    if (a->bits < 8 || a->bits > 16 || a->subsampling_w < 0 || a->subsampling_w > 1 || a->subsampling_h < 0 || a->subsampling_h > 1 ||
        a->width <= 0 || a->height <= 0)
        SFAIL("Super: input clip must be GRAY, 420, 422, 440, or 444, up to 16 bits, with constant dimensions.");
    if (si.hpad < 0 || si.vpad < 0) SFAIL("Super: hpad and vpad must not be negative.");
    si.width = a->width; si.height = a->height; si.bits = a->bits; si.gray = !!a->gray;
    if (si.gray) si.chroma = 0;
    si.modeYUV = si.chroma ? 7 : 1;
    si.xRatioUV = 1 << a->subsampling_w;
    si.yRatioUV = 1 << a->subsampling_h;
    int nLevelsMax = 0; // :220-227
    while (mvx_plane_height_luma(si.height, nLevelsMax, si.yRatioUV, si.vpad) >= si.yRatioUV * 2 &&
           mvx_plane_width_luma(si.width, nLevelsMax, si.xRatioUV, si.hpad) >= si.xRatioUV * 2)
        nLevelsMax++;
    if (si.levels <= 0 || si.levels > nLevelsMax) si.levels = nLevelsMax;
    if (si.levels > MVX_MAX_LEVELS) SFAIL("Super: too many levels.");
    si.super_width = si.width + 2 * si.hpad; // :257-264
    si.super_height = mvx_plane_super_offset(0, si.height, si.levels, si.pel, si.vpad, si.super_width, si.yRatioUV) / si.super_width;
    if (si.yRatioUV == 2 && (si.super_height & 1)) si.super_height++;
    if (si.xRatioUV == 2 && (si.super_width & 1)) si.super_width++;
    si.num_planes = si.gray ? 1 : 3;
    for (int p = 0; p < 3; p++) {
        si.plane_width[p] = p ? si.super_width / si.xRatioUV : si.super_width;
        si.plane_height[p] = p ? si.super_height / si.yRatioUV : si.super_height;
    }
    mvx_super *s = new mvx_super();
    s->info = si;
    *out = s;
    return MVX_OK;
#undef SFAIL
}

extern "C" __attribute__((visibility("default"))) void mvx_super_destroy(mvx_super *s) { delete s; }
extern "C" __attribute__((visibility("default"))) void mvx_super_get_info(const mvx_super *s, mvx_super_info *info) { *info = s->info; }

// ------------------------------------------------------------------------------------------------ device kernels

struct SuperPlaneGeom {
    int w, h, hpad, vpad, pw, ph; // level-0 plane
    long long src_pitch, dst_pitch;
};

struct SuperL0Args {
    const void *const *src; // [nframes*3]
    void *const *dst;       // [nframes*3]
    SuperPlaneGeom g[3];
    int pel, bits, modeYUV, nplanes;
    int XA[3], XB[3]; // columns [XA, XB) are produced by super_rows_kernel (multiples of 16; XA == XB: none)
    int tcA[3], tcB[3]; // tile columns [tcA, tcB) lie entirely inside [XA, XB): no workgroups are launched for them
};

template <typename T> __device__ __forceinline__ int ldT(const void *p, long long i) { return ((const T *)p)[i]; }
__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// horizontal / vertical half-pel rule at absolute padded coordinate X of a line of length n; s(i) reads sample i.
// SHARP 0: MVFrame.cpp:530-548 / :508-527; 1: :1153-1176 / :1115-1150; 2: :1071-1111 / :1019-1068
template <int SHARP, typename F> __device__ __forceinline__ int half_rule(F s, int X, int n, int pm) {
    if (X == n - 1) return s(X);
    if (SHARP == 0) return (s(X) + s(X + 1) + 1) >> 1;
    if (SHARP == 1) {
        if (X < 1 || X >= n - 3) return (s(X) + s(X + 1) + 1) >> 1;
        int v = (-(s(X - 1) + s(X + 2)) + (s(X) + s(X + 1)) * 9 + 8) >> 4;
        return min(pm, max(0, v));
    }
    if (X < 2 || X >= n - 4) return (s(X) + s(X + 1) + 1) >> 1;
    int m0 = s(X - 2), m1 = s(X - 1), m2 = s(X), m3 = s(X + 1), m4 = s(X + 2), m5 = s(X + 3);
    m2 = (m2 + m3) * 4; m2 -= m1 + m4; m2 *= 5; m0 += m5 + m2 + 16; m0 >>= 5;
    return max(0, min(m0, pm));
}

#define L0_TW 128
#define L0_TH 16
#define L0_HL 2 // halo before
#define L0_HR 3 // halo after
#define L0_LW (L0_TW + L0_HL + L0_HR)
#define L0_LH (L0_TH + L0_HL + L0_HR)

template <typename T, int SHARP, int PEL>
__global__ __launch_bounds__(256) void super_level0_kernel(SuperL0Args A) {
    __shared__ unsigned short sp0[L0_LH][L0_LW + 1];
    __shared__ unsigned short sv[L0_TH][L0_LW + 1];
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= A.nplanes || !(A.modeYUV & (1 << p))) return;
    const SuperPlaneGeom g = A.g[p];
    // blockIdx.x counts the tile columns that have work left (empty workgroups are not free: ~0.5 ns of dispatch each, and a
    // 1080p batch would launch millions of them)
    const int tx = (int)blockIdx.x < A.tcA[p] ? (int)blockIdx.x : (int)blockIdx.x - A.tcA[p] + A.tcB[p];
    const int X0 = tx * L0_TW, Y0 = blockIdx.y * L0_TH;
    if (X0 >= g.pw || Y0 >= g.ph) return;
    const unsigned char *src = (const unsigned char *)A.src[f * 3 + p];
    unsigned char *dst = (unsigned char *)A.dst[f * 3 + p];
    const int pm = (1 << A.bits) - 1;
    const int tid = threadIdx.x;

    // phase 1: padded plane tile (+halo) from the source frame, coordinates clamped (== PadReferenceFrame, MVFrame.cpp:1264-1318)
    for (int i = tid; i < L0_LH * L0_LW; i += 256) {
        int ly = i / L0_LW, lx = i - ly * L0_LW;
        int X = iclamp(X0 - L0_HL + lx, 0, g.pw - 1), Y = iclamp(Y0 - L0_HL + ly, 0, g.ph - 1);
        int sx = iclamp(X - g.hpad, 0, g.w - 1), sy = iclamp(Y - g.vpad, 0, g.h - 1);
        sp0[ly][lx] = (unsigned short)((const T *)(src + (long long)sy * g.src_pitch))[sx];
    }
    __syncthreads();
    if (PEL > 1) { // phase 2: vertical half-pel plane for the tile rows, all tile columns incl. halo
        for (int i = tid; i < L0_TH * L0_LW; i += 256) {
            int ty = i / L0_LW, lx = i - ty * L0_LW;
            int Y = Y0 + ty;
            int v = 0;
            if (Y < g.ph) v = half_rule<SHARP>([&](int yy) { return (int)sp0[yy - Y0 + L0_HL][lx]; }, Y, g.ph, pm);
            sv[ty][lx] = (unsigned short)v;
        }
        __syncthreads();
    }
    // phase 3: 8 consecutive samples per thread
    const int ty = tid >> 4, tx8 = (tid & 15) * 8;
    const int Y = Y0 + ty;
    if (Y >= g.ph) return;
    if (X0 + tx8 >= A.XA[p] && X0 + tx8 < A.XB[p]) return;
    const long long planeStride = g.dst_pitch * g.ph;
    const int idxH = PEL == 2 ? 1 : 2, idxV = PEL == 2 ? 2 : 8, idxHV = PEL == 2 ? 3 : 10;
    T *r0 = (T *)(dst + (long long)Y * g.dst_pitch);
    T *rH = (T *)(dst + idxH * planeStride + (long long)Y * g.dst_pitch);
    T *rV = (T *)(dst + idxV * planeStride + (long long)Y * g.dst_pitch);
    T *rHV = (T *)(dst + idxHV * planeStride + (long long)Y * g.dst_pitch);
    T o0[8], oH[8], oV[8], oHV[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int X = X0 + tx8 + k;
        int lx = tx8 + k + L0_HL;
        o0[k] = (T)sp0[ty + L0_HL][lx];
        if (PEL > 1) {
            int Xc = min(X, g.pw - 1); // keeps LDS indices in range for the (unstored) overhang of the last tile
            oH[k] = (T)half_rule<SHARP>([&](int xx) { return (int)sp0[ty + L0_HL][xx - X0 + L0_HL]; }, Xc, g.pw, pm);
            oV[k] = (T)sv[ty][lx];
            if (SHARP == 0) { // DiagonalBilinear MVFrame.cpp:551-572
                int a = sp0[ty + L0_HL][lx], b = sp0[ty + L0_HL][lx + 1], c = sp0[ty + L0_HL + 1][lx], d = sp0[ty + L0_HL + 1][lx + 1];
                int v;
                if (Y < g.ph - 1) v = (Xc < g.pw - 1) ? (a + b + c + d + 2) >> 2 : (a + c + 1) >> 1;
                else v = (Xc < g.pw - 1) ? (a + b + 1) >> 1 : a;
                oHV[k] = (T)v;
            } else
                oHV[k] = (T)half_rule<SHARP>([&](int xx) { return (int)sv[ty][xx - X0 + L0_HL]; }, Xc, g.pw, pm);
        }
    }
    const int Xs = X0 + tx8;
    if (Xs + 8 <= g.pw) {
        typedef T vec8 __attribute__((ext_vector_type(8)));
        vec8 v0, vH, vV, vHV;
#pragma unroll
        for (int k = 0; k < 8; k++) { v0[k] = o0[k]; if (PEL > 1) { vH[k] = oH[k]; vV[k] = oV[k]; vHV[k] = oHV[k]; } }
        *(vec8 *)(r0 + Xs) = v0;
        if (PEL > 1) { *(vec8 *)(rH + Xs) = vH; *(vec8 *)(rV + Xs) = vV; *(vec8 *)(rHV + Xs) = vHV; }
    } else {
        for (int k = 0; k < 8 && Xs + k < g.pw; k++) {
            r0[Xs + k] = o0[k];
            if (PEL > 1) { rH[Xs + k] = oH[k]; rV[Xs + k] = oV[k]; rHV[Xs + k] = oHV[k]; }
        }
    }
}

#include "mvx_super_rows.h"

// pel 4: the twelve averaged planes, MVFrame.cpp:1489-1524.  dst = (a[shifted by ax,ay] + b + 1) >> 1 over (pw-ax) x (ph-ay);
// the untouched last column / row stays 0 as in the reference (frame memset, MVSuper.c:75).
struct AvgOp { int d, a, b, ax, ay; };
struct SuperAvgArgs {
    void *const *dst;
    SuperPlaneGeom g[3];
    int modeYUV, nplanes;
    AvgOp ops[8];
    int nops;
};

template <typename T> __global__ __launch_bounds__(256) void super_avg_kernel(SuperAvgArgs A) {
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= A.nplanes || !(A.modeYUV & (1 << p))) return;
    const SuperPlaneGeom g = A.g[p];
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= g.pw || Y >= g.ph) return;
    unsigned char *dst = (unsigned char *)A.dst[f * 3 + p];
    const long long ps = g.dst_pitch * g.ph;
    for (int k = 0; k < A.nops; k++) {
        const AvgOp o = A.ops[k];
        T *d = (T *)(dst + o.d * ps + (long long)Y * g.dst_pitch);
        int v = 0;
        if (X < g.pw - o.ax && Y < g.ph - o.ay) {
            const T *a = (const T *)(dst + o.a * ps + (long long)(Y + o.ay) * g.dst_pitch);
            const T *b = (const T *)(dst + o.b * ps + (long long)Y * g.dst_pitch);
            v = (a[X + o.ax] + b[X] + 1) >> 1;
        }
        d[X] = (T)v;
    }
}

// 2x reduction of level L into the padded rectangle of level L+1.  MVFrame.cpp:575-1014 + pad :1264-1318.
struct SuperReduceArgs {
    const void *const *src; // FROM_SRC: source frames [nframes*3]; else unused
    void *const *dst;       // super frames [nframes*3]
    long long src_pitch[3], dst_pitch[3];
    long long in_off[3], out_off[3]; // byte offsets of level L / L+1 (sub-pel plane 0) in the super plane
    int in_w[3], in_h[3], in_hpad[3], in_vpad[3];
    int out_w[3], out_h[3], out_hpad[3], out_vpad[3];
    int modeYUV, nplanes;
    int XA[3], XB[3], YA[3], YB[3]; // interior outputs super_reduce_rows_kernel produces (XA == XB: none)
    // blocks of 64 x 4 outputs: ncb x nrb of them; those with column block in [cbA, cbB) AND row block in [rbA, rbB) lie entirely
    // inside the rows kernel's region and are not launched.  blockIdx.x enumerates the others: nfull blocks of the full-width
    // row bands above and below, then the side blocks of the middle band.
    int ncb[3], nrb[3], cbA[3], cbB[3], rbA[3], rbB[3], nfull[3], ntotal[3];
};

template <typename T, int RF, bool FROM_SRC>
__global__ __launch_bounds__(256) void super_reduce_kernel(SuperReduceArgs A) {
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= A.nplanes || !(A.modeYUV & (1 << p))) return;
    const int ow = A.out_w[p], oh = A.out_h[p], ohp = A.out_hpad[p], ovp = A.out_vpad[p];
    int idx = blockIdx.x, cb, rb;
    if (idx >= A.ntotal[p]) return;
    if (idx < A.nfull[p]) { const int r = idx / A.ncb[p]; cb = idx - r * A.ncb[p]; rb = r < A.rbA[p] ? r : r - A.rbA[p] + A.rbB[p]; }
    else {
        idx -= A.nfull[p];
        const int ns = A.cbA[p] + A.ncb[p] - A.cbB[p], r = idx / ns, c = idx - r * ns;
        rb = A.rbA[p] + r; cb = c < A.cbA[p] ? c : c - A.cbA[p] + A.cbB[p];
    }
    const int X = cb * 64 + (threadIdx.x & 63), Y = rb * 4 + (threadIdx.x >> 6);
    if (X >= ow + 2 * ohp || Y >= oh + 2 * ovp) return;
    if (X - ohp >= A.XA[p] && X - ohp < A.XB[p] && Y - ovp >= A.YA[p] && Y - ovp < A.YB[p]) return;
    const int x = iclamp(X - ohp, 0, ow - 1), y = iclamp(Y - ovp, 0, oh - 1);
    unsigned char *dplane = (unsigned char *)A.dst[f * 3 + p];
    const unsigned char *sbase;
    long long sp;
    const int iw = A.in_w[p], ih = A.in_h[p];
    if (FROM_SRC) { sbase = (const unsigned char *)A.src[f * 3 + p]; sp = A.src_pitch[p]; }
    else { sp = A.dst_pitch[p]; sbase = dplane + A.in_off[p] + (long long)A.in_vpad[p] * sp + (long long)A.in_hpad[p] * (long long)sizeof(T); }
    // level 0 is reduced BEFORE it is padded (mvgofReduce then mvgofPad, MVSuper.c:88-89): outside the interior the
    // reference reads the zero-filled frame; deeper levels were padded right after they were produced (MVFrame.cpp:1931).
    auto S = [&](int xx, int yy) -> int {
        if (FROM_SRC && (xx >= iw || yy >= ih)) return 0;
        return (int)((const T *)(sbase + (long long)yy * sp))[xx];
    };
    int out;
    if (RF == 0) {
        out = (S(2 * x, 2 * y) + S(2 * x + 1, 2 * y) + S(2 * x + 1, 2 * y + 1) + S(2 * x, 2 * y + 1) + 2) / 4;
    } else {
        auto V = [&](int XX) -> int { // vertical pass at intermediate column XX
            if (RF == 1) {
                if (y == 0) return (S(XX, 0) + S(XX, 1) + 1) / 2;
                return (S(XX, 2 * y - 1) + S(XX, 2 * y) * 2 + S(XX, 2 * y + 1) + 2) / 4;
            }
            if (y == 0 || y == oh - 1) return (S(XX, 2 * y) + S(XX, 2 * y + 1) + 1) / 2;
            if (RF == 2) return (S(XX, 2 * y - 1) + (S(XX, 2 * y) + S(XX, 2 * y + 1)) * 3 + S(XX, 2 * y + 2) + 4) / 8;
            int m0 = S(XX, 2 * y - 2), m1 = S(XX, 2 * y - 1), m2 = S(XX, 2 * y), m3 = S(XX, 2 * y + 1), m4 = S(XX, 2 * y + 2), m5 = S(XX, 2 * y + 3);
            if (RF == 3) { m2 = (m2 + m3) * 22; m1 = (m1 + m4) * 9; m0 += m5 + m2 + m1 + 32; return m0 >> 6; }
            m2 = (m2 + m3) * 10; m1 = (m1 + m4) * 5; m0 += m5 + m2 + m1 + 16; return m0 >> 5;
        };
        if (x == 0) out = (V(0) + V(1) + 1) / 2;
        else if (RF == 1) out = (V(2 * x - 1) + V(2 * x) * 2 + V(2 * x + 1) + 2) / 4;
        else if (x == ow - 1) out = (V(2 * x) + V(2 * x + 1) + 1) / 2;
        else if (RF == 2) out = (V(2 * x - 1) + (V(2 * x) + V(2 * x + 1)) * 3 + V(2 * x + 2) + 4) / 8;
        else {
            int m0 = V(2 * x - 2), m1 = V(2 * x - 1), m2 = V(2 * x), m3 = V(2 * x + 1), m4 = V(2 * x + 2), m5 = V(2 * x + 3);
            if (RF == 3) { m2 = (m2 + m3) * 22; m1 = (m1 + m4) * 9; m0 += m5 + m2 + m1 + 32; out = m0 >> 6; }
            else { m2 = (m2 + m3) * 10; m1 = (m1 + m4) * 5; m0 += m5 + m2 + m1 + 16; out = m0 >> 5; }
        }
    }
    ((T *)(dplane + A.out_off[p] + (long long)Y * A.dst_pitch[p]))[X] = (T)out;
}

// ------------------------------------------------------------------------------------------------ host: launch

struct PtrScratch { // per-thread device scratch for the frame-pointer tables
    void *d = nullptr;
    size_t cap = 0;
    int dev = -1;
};
static thread_local PtrScratch g_scratch[3];

static thread_local CallGuard g_scratch_guard; // (per thread like the tables: a thread that alternates streams must not rewrite a table an earlier launch still reads)
static int upload_ptrs(int which, const void *const *host, size_t n, hipStream_t st, void **dev_out) {
    PtrScratch &s = g_scratch[which];
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    size_t bytes = n * sizeof(void *);
    if (s.cap < bytes || s.dev != dev) {
        if (s.d) (void)hipFree(s.d);
        s.cap = bytes < 4096 ? 4096 : bytes * 2;
        HIP_CHECK(hipMalloc(&s.d, s.cap));
        s.dev = dev;
    }
    HIP_CHECK(hipMemcpyAsync(s.d, host, bytes, hipMemcpyHostToDevice, st));
    *dev_out = s.d;
    return MVX_OK;
}

template <typename T> static void launch_level0(const SuperL0Args &A, int sharp, int pel, dim3 grid, hipStream_t st) {
#define L0(S, P) hipLaunchKernelGGL((super_level0_kernel<T, S, P>), grid, dim3(256), 0, st, A)
    if (pel == 1) L0(2, 1);
    else if (pel == 2) { if (sharp == 0) L0(0, 2); else if (sharp == 1) L0(1, 2); else L0(2, 2); }
    else { if (sharp == 0) L0(0, 4); else if (sharp == 1) L0(1, 4); else L0(2, 4); }
#undef L0
}

template <typename T, bool FS> static void launch_reduce(const SuperReduceArgs &A, int rf, dim3 grid, hipStream_t st) {
#define RD(R) hipLaunchKernelGGL((super_reduce_kernel<T, R, FS>), grid, dim3(256), 0, st, A)
    switch (rf) { case 0: RD(0); break; case 1: RD(1); break; case 2: RD(2); break; case 3: RD(3); break; default: RD(4); break; }
#undef RD
}

// mv.Super(pelclip=...): MVFrame.cpp:1529-1631 mvpRefineExt.  One thread per padded level-0 sample position; it fills the
// sub-pel planes 1..pel^2-1 there.  mode 1 (plain pelclip): plane i at padded (X, Y) = pelclip[(y*pel + i/pel)][(x*pel + i%pel)]
// with (x, y) the interior coordinate clamped into the picture (== copy + PadReferenceFrame :1548-1562).  mode 2 (padded
// pelclip): the reference reads the pelclip from its origin into the plane's origin over w x h samples only (:1543-1558);
// the rest of the plane keeps the frame's memset value, written here as 0 so the result does not depend on what the buffer held.
struct SuperExtArgs {
    const void *const *pel; // [nframes*3]
    void *const *dst;       // [nframes*3]
    SuperPlaneGeom g[3];
    long long pel_pitch[3];
    int pel_n, mode, modeYUV, nplanes;
};

template <typename T, int PEL>
__global__ __launch_bounds__(256) void super_ext_kernel(SuperExtArgs A) {
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= A.nplanes || !(A.modeYUV & (1 << p))) return;
    const SuperPlaneGeom g = A.g[p];
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= g.pw || Y >= g.ph) return;
    const unsigned char *pc = (const unsigned char *)A.pel[f * 3 + p];
    unsigned char *dst = (unsigned char *)A.dst[f * 3 + p];
    const long long planeStride = g.dst_pitch * g.ph;
    typedef T vecp __attribute__((ext_vector_type(PEL)));
    int x, y;
    bool zero = false;
    if (A.mode == 1) { x = iclamp(X - g.hpad, 0, g.w - 1); y = iclamp(Y - g.vpad, 0, g.h - 1); }
    else { x = X; y = Y; zero = X >= g.w || Y >= g.h; }
#pragma unroll
    for (int r = 0; r < PEL; r++) {
        vecp v = (vecp)0;
        if (!zero) v = *(const vecp *)(pc + (long long)(y * PEL + r) * A.pel_pitch[p] + (long long)x * PEL * sizeof(T));
#pragma unroll
        for (int c = 0; c < PEL; c++) {
            const int i = r * PEL + c;
            if (i == 0) continue;
            ((T *)(dst + i * planeStride + (long long)Y * g.dst_pitch))[X] = v[c];
        }
    }
}

static int super_frames_impl(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3], const void *const *pelclip,
                             const ptrdiff_t pelclip_pitch[3], int pelMode, void *const *dst, const ptrdiff_t dst_pitch[3], const ptrdiff_t *shadow_stride,
                             void *stream);

// ---- shifted copies of a super frame's planes ("shadows") ---------------------------------------------------------------
// The search reads reference blocks at arbitrary sample positions, i.e. at byte addresses that are mostly NOT multiples of
// four, and gfx950's texture-addresser path handles a wave-level 16-byte-per-lane load at such addresses ~3.7x slower than at
// dword-aligned ones (tools/micro/ta_pattern.hip: 85 against 23 CU-cycles for 32 rows x 32 bytes).  With 288 GB of HBM the fix
// is a layout one: next to every plane of a super frame the caller keeps 4 / bytes-per-sample - 1 copies of the WHOLE plane
// buffer shifted left by 1 .. n samples (copy k, byte i = plane byte i + k * bps).  A block at a sample position x with
// x % (4 / bps) == k is then read from copy k at x - k: same samples, dword-aligned address.  mvx_analyse_set_ref_shadow tells
// a search where the copies are.
struct ShadowArgs { void *const *planes; long long size[3], stride[3], begin[3]; int nplanes, bps, copies8; };
// blockIdx.y = frame * 2 + kind; kind 0: the luma plane shifted left by one sample; kind 1: U and V interleaved sample by sample
__global__ __launch_bounds__(256) void super_shadow_kernel(ShadowArgs A) {
    const int f = blockIdx.y >> 1, kind = blockIdx.y & 1;
    const long long i = A.begin[kind] + ((long long)blockIdx.x * 256 + threadIdx.x) * 16; // begin: bytes before it were written by the Super kernels themselves
    if (kind == 0) {
        if ((A.bps == 1 && !A.copies8) || i >= A.size[0]) return; // (8-bit clips keep no shifted luma copy, unless "shadow8" asks for the three)
        unsigned char *base = (unsigned char *)A.planes[f * 3];
        const uint4 a = *(const uint4 *)(base + i);
        const unsigned b = i + 16 < A.size[0] ? *(const unsigned *)(base + i + 16) : 0u;
        if (A.bps == 1) { // copy k = the plane shifted left by k bytes
            for (unsigned k = 1; k < 4; k++) {
                uint4 o;
                o.x = __builtin_amdgcn_alignbit(a.y, a.x, 8 * k); o.y = __builtin_amdgcn_alignbit(a.z, a.y, 8 * k);
                o.z = __builtin_amdgcn_alignbit(a.w, a.z, 8 * k); o.w = __builtin_amdgcn_alignbit(b, a.w, 8 * k);
                *(uint4 *)(base + (long long)k * A.stride[0] + i) = o;
            }
            return;
        }
        const unsigned sh = 8u * A.bps;
        uint4 o;
        o.x = __builtin_amdgcn_alignbit(a.y, a.x, sh); o.y = __builtin_amdgcn_alignbit(a.z, a.y, sh);
        o.z = __builtin_amdgcn_alignbit(a.w, a.z, sh); o.w = __builtin_amdgcn_alignbit(b, a.w, sh);
        *(uint4 *)(base + A.stride[0] + i) = o;
    } else {
        if (A.nplanes < 3 || i >= A.size[1]) return;
        const unsigned char *pu = (const unsigned char *)A.planes[f * 3 + 1], *pv = (const unsigned char *)A.planes[f * 3 + 2];
        const uint4 u = *(const uint4 *)(pu + i), v = *(const uint4 *)(pv + i);
        if (A.bps == 1) { // 8-bit samples: dword k of u holds samples 4k .. 4k+3 -> (U V U V) twice
            auto lo8 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05010400u); };
            auto hi8 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07030602u); };
            const uint4 o0 = { lo8(u.x, v.x), hi8(u.x, v.x), lo8(u.y, v.y), hi8(u.y, v.y) }, o1 = { lo8(u.z, v.z), hi8(u.z, v.z), lo8(u.w, v.w), hi8(u.w, v.w) };
            unsigned char *d8 = (unsigned char *)A.planes[f * 3 + 1] + A.stride[1] + 2 * i;
            *(uint4 *)d8 = o0; *(uint4 *)(d8 + 16) = o1;
            return;
        }
        // 16-bit samples: dword k of u holds samples 2k, 2k+1 -> (U 2k | V 2k), (U 2k+1 | V 2k+1)
        auto lo = [](unsigned a, unsigned b) { return (a & 0xffffu) | (b << 16); };
        auto hi = [](unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); };
        uint4 o0 = { lo(u.x, v.x), hi(u.x, v.x), lo(u.y, v.y), hi(u.y, v.y) }, o1 = { lo(u.z, v.z), hi(u.z, v.z), lo(u.w, v.w), hi(u.w, v.w) };
        unsigned char *d = (unsigned char *)A.planes[f * 3 + 1] + A.stride[1] + 2 * i;
        *(uint4 *)d = o0; *(uint4 *)(d + 16) = o1;
    }
}
// 16-bit clips: shadows exist (1).  8-bit clips: none (0) -- measured (r2, 1080p Degrain1): shifted copies of an 8-bit plane (three are
// needed) quadruple the cache footprint of every chain and cost more than the aligned loads save (1250-1340 fps with copies, 2005
// without); the search then simply loads from the planes themselves.
// r3: 8-bit 4:2:x clips get the UV-interleaved plane too (U and V of a chroma block in ONE row of twice the width: half the load
// instructions and cache lines per candidate), still no luma copy.
extern "C" __attribute__((visibility("default"))) int mvx_super_shadow_copies(const mvx_super *s) { return s->info.bits <= 8 ? (s->info.num_planes >= 3 ? 1 : 0) : 1; }
// bytes of shadow data a caller must provide behind plane p: the luma plane's shifted copy, and -- behind the U plane -- ONE plane
// of twice the chroma size holding U and V interleaved (every chroma position is dword-aligned there, no shifted copy needed)
extern "C" __attribute__((visibility("default"))) void mvx_super_shadow_bytes(const mvx_super *s, const ptrdiff_t pitch[3], size_t extra[3]) {
    extra[0] = extra[1] = extra[2] = 0;
    if (!mvx_super_shadow_copies(s)) return;
    if (s->info.bits > 8) extra[0] = (size_t)s->info.plane_height[0] * pitch[0];
    else if (mvx_debug_value("shadow8", 0)) extra[0] = 3 * (((size_t)s->info.plane_height[0] * pitch[0] + 255) & ~(size_t)255); // (three copies, each at a multiple of the 256-byte rounded plane size)
    if (s->info.num_planes >= 3) extra[1] = 2 * (size_t)s->info.plane_height[1] * pitch[1];
}
static int shadow_check(const mvx_super_info &si, const ptrdiff_t pitch[3], const ptrdiff_t copy_stride[3]) {
    for (int p = (si.bits > 8 || mvx_debug_value("shadow8", 0)) ? 0 : 1; p < si.num_planes && p < 2; p++) {
        const long long size = (long long)si.plane_height[p] * pitch[p];
        if (pitch[p] % 16 || copy_stride[p] % 16 || copy_stride[p] < size) { mvx_set_error("mvx_super_shadow_frames: pitch and shadow offset must be multiples of 16 bytes, the offset at least one plane"); return MVX_E_ARG; }
    }
    if (si.num_planes >= 3 && pitch[1] != pitch[2]) { mvx_set_error("mvx_super_shadow_frames: U and V must share one pitch"); return MVX_E_ARG; }
    return MVX_OK;
}
// dplanes: the device copy of the plane-pointer table; begin[p]: first byte of plane p the linear kernel has to handle
static int shadow_launch(const mvx_super_info &si, int nframes, void *const *dplanes, const ptrdiff_t pitch[3], const ptrdiff_t copy_stride[3],
                         const long long begin[2], hipStream_t st) {
    ShadowArgs A;
    memset(&A, 0, sizeof(A));
    A.nplanes = si.num_planes; A.bps = si.bits > 8 ? 2 : 1; A.planes = dplanes; A.copies8 = A.bps == 1 && mvx_debug_value("shadow8", 0);
    long long maxsize = 0;
    for (int p = (A.bps == 1 && !A.copies8) ? 1 : 0; p < si.num_planes && p < 2; p++) {
        A.size[p] = (long long)si.plane_height[p] * pitch[p]; A.stride[p] = copy_stride[p]; A.begin[p] = begin[p];
        if (A.size[p] - begin[p] > maxsize) maxsize = A.size[p] - begin[p];
    }
    if (maxsize <= 0) return MVX_OK;
    hipLaunchKernelGGL(super_shadow_kernel, dim3((unsigned)((maxsize / 16 + 255) / 256), nframes * 2), dim3(256), 0, st, A);
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) int mvx_super_shadow_frames(const mvx_super *s, int nframes, void *const *planes, const ptrdiff_t pitch[3],
                                                                              const ptrdiff_t copy_stride[3], void *stream) {
    if (nframes <= 0 || !mvx_super_shadow_copies(s)) return MVX_OK;
    const mvx_super_info &si = s->info;
    hipStream_t st = (hipStream_t)stream;
    CallGuard::Scope scope(g_scratch_guard, st);
    int rc;
    if ((rc = shadow_check(si, pitch, copy_stride))) return rc;
    void *dpl = nullptr;
    if ((rc = upload_ptrs(1, (const void *const *)planes, (size_t)nframes * 3, st, &dpl))) return rc;
    const long long begin[2] = { 0, 0 };
    return shadow_launch(si, nframes, (void *const *)dpl, pitch, copy_stride, begin, st);
}

// mvx_super_frames + mvx_super_shadow_frames in one call; for pel 2 the level-0 kernels write the shadow data themselves
extern "C" __attribute__((visibility("default"))) int mvx_super_frames_shadow(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3],
                                void *const *dst, const ptrdiff_t dst_pitch[3], const ptrdiff_t shadow_stride[3], void *stream) {
    return super_frames_impl(s, nframes, src, src_pitch, nullptr, nullptr, 0, dst, dst_pitch, shadow_stride, stream);
}

extern "C" __attribute__((visibility("default"))) int mvx_super_frames(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3],
                                void *const *dst, const ptrdiff_t dst_pitch[3], void *stream) {
    return super_frames_impl(s, nframes, src, src_pitch, nullptr, nullptr, 0, dst, dst_pitch, nullptr, stream);
}

// MVSuper.c:229-256
extern "C" __attribute__((visibility("default"))) int mvx_super_pelclip_mode(const mvx_super *s, int pelclip_width, int pelclip_height, int32_t *mode, char *err) {
    const mvx_super_info &si = s->info;
    if (err) err[0] = 0;
    *mode = 0;
    if (si.pel < 2) return MVX_OK;
    if (pelclip_width == si.width * si.pel && pelclip_height == si.height * si.pel) { *mode = 1; return MVX_OK; }
    if (pelclip_width == (si.width + si.hpad * 2) * si.pel && pelclip_height == (si.height + si.vpad * 2) * si.pel) { *mode = 2; return MVX_OK; }
    if (err) snprintf(err, MVX_ERRLEN, "Super: pelclip's dimensions must be multiples of the input clip's dimensions.");
    mvx_set_error("Super: pelclip's dimensions must be multiples of the input clip's dimensions.");
    return MVX_E_ARG;
}

extern "C" __attribute__((visibility("default"))) int mvx_super_frames_pelclip(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3],
                                const void *const *pelclip, const ptrdiff_t pelclip_pitch[3], int pelclip_mode, void *const *dst,
                                const ptrdiff_t dst_pitch[3], void *stream) {
    if (pelclip_mode < 0 || pelclip_mode > 2 || (pelclip_mode && s->info.pel < 2)) { mvx_set_error("mvx_super_frames_pelclip: bad pelclip mode"); return MVX_E_ARG; }
    if (pelclip_mode) {
        if (!pelclip || !pelclip_pitch) { mvx_set_error("mvx_super_frames_pelclip: no pelclip frames"); return MVX_E_ARG; }
        const int align = s->info.pel * (s->info.bits <= 8 ? 1 : 2); // the kernel loads pel samples at once
        for (int p = 0; p < s->info.num_planes; p++)
            if (pelclip_pitch[p] % align) { mvx_set_error("mvx_super_frames_pelclip: pelclip pitch must be a multiple of pel samples"); return MVX_E_ARG; }
        for (int f = 0; f < nframes; f++)
            for (int p = 0; p < s->info.num_planes; p++)
                if (((uintptr_t)pelclip[f * 3 + p]) % align) { mvx_set_error("mvx_super_frames_pelclip: pelclip planes must be aligned to pel samples"); return MVX_E_ARG; }
    }
    return super_frames_impl(s, nframes, src, src_pitch, pelclip, pelclip_pitch, pelclip_mode, dst, dst_pitch, nullptr, stream);
}

static int super_frames_impl(mvx_super *s, int nframes, const void *const *src, const ptrdiff_t src_pitch[3], const void *const *pelclip,
                             const ptrdiff_t pelclip_pitch[3], int pelMode, void *const *dst, const ptrdiff_t dst_pitch[3], const ptrdiff_t *shadow_stride,
                             void *stream) {
    if (nframes <= 0) return MVX_OK;
    if (shadow_stride && !mvx_super_shadow_copies(s)) shadow_stride = nullptr;
    if (shadow_stride) { int rc = shadow_check(s->info, dst_pitch, shadow_stride); if (rc) return rc; }
    const mvx_super_info &si = s->info;
    hipStream_t st = (hipStream_t)stream;
    CallGuard::Scope scope(g_scratch_guard, st);
    for (int p = 0; p < si.num_planes; p++)
        if (dst_pitch[p] % 16) { mvx_set_error("mvx_super_frames: dst pitch must be a multiple of 16 bytes"); return MVX_E_ARG; }
    void *dsrc = nullptr, *ddst = nullptr;
    int rc;
    if ((rc = upload_ptrs(0, src, (size_t)nframes * 3, st, &dsrc))) return rc;
    if ((rc = upload_ptrs(1, (const void *const *)dst, (size_t)nframes * 3, st, &ddst))) return rc;
    const bool u8 = si.bits <= 8;

    SuperL0Args A;
    memset(&A, 0, sizeof(A));
    A.src = (const void *const *)dsrc; A.dst = (void *const *)ddst;
    A.pel = si.pel; A.bits = si.bits; A.modeYUV = si.modeYUV; A.nplanes = si.num_planes;
    int maxpw = 0, maxph = 0;
    for (int p = 0; p < si.num_planes; p++) {
        LevelPlane lp;
        mvx_level_plane(si, 0, p, dst_pitch[p], &lp);
        A.g[p] = { lp.w, lp.h, lp.hpad, lp.vpad, lp.pw, lp.ph, (long long)src_pitch[p], (long long)dst_pitch[p] };
        if (lp.pw > maxpw) maxpw = lp.pw;
        if (lp.ph > maxph) maxph = lp.ph;
    }
    // with a pelclip only plane 0 comes from the source (PEL=1 instantiation); the sub-pel planes are taken from the pelclip
    const int l0pel = pelMode ? 1 : si.pel;
    // pel 2: the columns whose filter window lies inside the source row go to super_rows_kernel (mvx_super_rows.h), which also writes
    // the shadow data of level 0; it loads dword-aligned vectors, so the source rows and the left padding must keep that alignment
    bool rows = l0pel == 2 && !mvx_debug_value("super_rows_off", 0);
    const int bps = u8 ? 1 : 2, NS = 16 / bps;
    for (int p = 0; rows && p < si.num_planes; p++) {
        if ((A.g[p].hpad * bps) % 4 || src_pitch[p] % 4) rows = false;
        for (int f = 0; rows && f < nframes; f++)
            if (((uintptr_t)src[f * 3 + p] % 4) || ((uintptr_t)dst[f * 3 + p] % 16)) rows = false;
    }
    bool fusedShadow = false;
    if (rows) {
        SuperRowsArgs Q;
        memset(&Q, 0, sizeof(Q));
        Q.src = A.src; Q.dst = A.dst; Q.bits = si.bits; Q.modeYUV = si.modeYUV;
        for (int p = 0; p < si.num_planes; p++) {
            const SuperPlaneGeom &g = A.g[p];
            Q.g[p] = g;
            int xa = (g.hpad + 4 + NS - 1) / NS * NS, last = g.w + g.hpad - NS - 4; // last: the largest X whose window ends inside the row
            int xb = last >= xa ? last / NS * NS + NS : xa;
            Q.XA[p] = A.XA[p] = xa; Q.XB[p] = A.XB[p] = xb;
            if (shadow_stride && p < 2) Q.shadow[p] = shadow_stride[p];
        }
        const bool chromaPair = si.num_planes >= 3 && A.g[1].w == A.g[2].w && A.g[1].h == A.g[2].h && src_pitch[1] == src_pitch[2] && dst_pitch[1] == dst_pitch[2];
        fusedShadow = shadow_stride && !u8 && (si.num_planes < 3 || chromaPair);
        auto launch = [&](int kind, int first, int nz) {
            int mw = 0, mh = 0;
            bool any = false;
            for (int p = first; p < first + (kind == 2 ? 1 : nz); p++) { if (Q.XB[p] > Q.XA[p]) any = true; if (Q.XB[p] > mw) mw = Q.XB[p]; if (Q.g[p].ph > mh) mh = Q.g[p].ph; }
            if (!any) return;
            Q.firstPlane = first; Q.nz = nz;
            dim3 gr((mw / NS + 63) / 64, (mh + 7) / 8, nframes * nz); // mw: the largest XB (threads start at column 0)
#define RW(T, S, K) hipLaunchKernelGGL((super_rows_kernel<T, S, K, 2>), gr, dim3(256), 0, st, Q)
#define RWS(T, K) do { if (si.sharp == 0) RW(T, 0, K); else if (si.sharp == 1) RW(T, 1, K); else RW(T, 2, K); } while (0)
            if (u8) RWS(uint8_t, 0);
            else if (kind == 0) RWS(uint16_t, 0);
            else if (kind == 1) RWS(uint16_t, 1);
            else RWS(uint16_t, 2);
#undef RWS
#undef RW
        };
        if (fusedShadow) {
            launch(1, 0, 1);
            if (si.num_planes >= 3) launch(2, 1, 1);
        } else
            launch(0, 0, si.num_planes);
    }
    int gx = 0;
    for (int p = 0; p < si.num_planes; p++) {
        const int ntx = (A.g[p].pw + L0_TW - 1) / L0_TW;
        A.tcA[p] = (A.XA[p] + L0_TW - 1) / L0_TW; A.tcB[p] = A.XB[p] / L0_TW;
        if (A.tcB[p] < A.tcA[p]) A.tcB[p] = A.tcA[p];
        if (A.tcA[p] + ntx - A.tcB[p] > gx) gx = A.tcA[p] + ntx - A.tcB[p];
    }
    dim3 grid(gx, (maxph + L0_TH - 1) / L0_TH, nframes * 3);
    if (u8) launch_level0<uint8_t>(A, si.sharp, l0pel, grid, st); else launch_level0<uint16_t>(A, si.sharp, l0pel, grid, st);
    if (fusedShadow) { // the two strips super_level0_kernel produced
        ShadowStripArgs Z;
        memset(&Z, 0, sizeof(Z));
        Z.planes = A.dst; Z.nplanes = si.num_planes;
        int mw = 0, mh = 0;
        for (int p = 0; p < si.num_planes && p < 2; p++) {
            Z.g[p] = A.g[p]; Z.XA[p] = A.XA[p]; Z.XB[p] = A.XB[p]; Z.shadow[p] = shadow_stride[p];
            const int n = A.XA[p] + A.g[p].pw - A.XB[p];
            if (n > mw) mw = n;
            if (A.g[p].ph > mh) mh = A.g[p].ph;
        }
        hipLaunchKernelGGL(super_shadow_strip_kernel, dim3((mw + 63) / 64, mh, nframes * 2), dim3(256), 0, st, Z);
    }

    if (pelMode) {
        void *dpel = nullptr;
        if ((rc = upload_ptrs(2, pelclip, (size_t)nframes * 3, st, &dpel))) return rc;
        SuperExtArgs E;
        memset(&E, 0, sizeof(E));
        E.pel = (const void *const *)dpel; E.dst = (void *const *)ddst; E.mode = pelMode; E.modeYUV = si.modeYUV; E.nplanes = si.num_planes;
        for (int p = 0; p < si.num_planes; p++) { E.g[p] = A.g[p]; E.pel_pitch[p] = pelclip_pitch[p]; }
        dim3 ge((maxpw + 255) / 256, maxph, nframes * 3);
        if (si.pel == 2) { if (u8) hipLaunchKernelGGL((super_ext_kernel<uint8_t, 2>), ge, dim3(256), 0, st, E); else hipLaunchKernelGGL((super_ext_kernel<uint16_t, 2>), ge, dim3(256), 0, st, E); }
        else { if (u8) hipLaunchKernelGGL((super_ext_kernel<uint8_t, 4>), ge, dim3(256), 0, st, E); else hipLaunchKernelGGL((super_ext_kernel<uint16_t, 4>), ge, dim3(256), 0, st, E); }
    }

    if (si.pel == 4 && !pelMode) { // MVFrame.cpp:1511-1523, two dependent passes
        SuperAvgArgs B;
        memset(&B, 0, sizeof(B));
        B.dst = (void *const *)ddst; B.modeYUV = si.modeYUV; B.nplanes = si.num_planes;
        for (int p = 0; p < 3; p++) B.g[p] = A.g[p];
        dim3 g2((maxpw + 255) / 256, maxph, nframes * 3);
        const AvgOp pass1[8] = { { 1, 0, 2, 0, 0 }, { 9, 8, 10, 0, 0 }, { 4, 0, 8, 0, 0 }, { 6, 2, 10, 0, 0 },
                                 { 3, 0, 2, 1, 0 }, { 11, 8, 10, 1, 0 }, { 12, 0, 8, 0, 1 }, { 14, 2, 10, 0, 1 } };
        const AvgOp pass2[4] = { { 5, 4, 6, 0, 0 }, { 13, 12, 14, 0, 0 }, { 7, 4, 6, 1, 0 }, { 15, 12, 14, 1, 0 } };
        memcpy(B.ops, pass1, sizeof(pass1)); B.nops = 8;
        if (u8) hipLaunchKernelGGL(super_avg_kernel<uint8_t>, g2, dim3(256), 0, st, B); else hipLaunchKernelGGL(super_avg_kernel<uint16_t>, g2, dim3(256), 0, st, B);
        memcpy(B.ops, pass2, sizeof(pass2)); B.nops = 4;
        if (u8) hipLaunchKernelGGL(super_avg_kernel<uint8_t>, g2, dim3(256), 0, st, B); else hipLaunchKernelGGL(super_avg_kernel<uint16_t>, g2, dim3(256), 0, st, B);
    }

    for (int L = 0; L + 1 < si.levels; L++) {
        SuperReduceArgs R;
        memset(&R, 0, sizeof(R));
        R.src = (const void *const *)dsrc; R.dst = (void *const *)ddst; R.modeYUV = si.modeYUV; R.nplanes = si.num_planes;
        int mw = 0, mh = 0;
        for (int p = 0; p < si.num_planes; p++) {
            LevelPlane a, b;
            mvx_level_plane(si, L, p, dst_pitch[p], &a);
            mvx_level_plane(si, L + 1, p, dst_pitch[p], &b);
            R.src_pitch[p] = src_pitch[p]; R.dst_pitch[p] = dst_pitch[p];
            R.in_off[p] = a.off; R.out_off[p] = b.off;
            R.in_w[p] = a.w; R.in_h[p] = a.h; R.in_hpad[p] = a.hpad; R.in_vpad[p] = a.vpad;
            R.out_w[p] = b.w; R.out_h[p] = b.h; R.out_hpad[p] = b.hpad; R.out_vpad[p] = b.vpad;
            if (b.pw > mw) mw = b.pw;
            if (b.ph > mh) mh = b.ph;
        }
        { // interior outputs whose filter window needs no edge rule: rows-in-registers kernel (mvx_super_rows.h)
            SuperReduceRowsArgs Q;
            memset(&Q, 0, sizeof(Q));
            Q.src = R.src; Q.dst = R.dst; Q.modeYUV = R.modeYUV; Q.nplanes = R.nplanes;
            bool ok = !mvx_debug_value("super_rows_off", 0);
            int qw = 0, qh = 0;
            for (int p = 0; ok && p < si.num_planes; p++) {
                Q.src_pitch[p] = R.src_pitch[p]; Q.dst_pitch[p] = R.dst_pitch[p]; Q.in_off[p] = R.in_off[p]; Q.out_off[p] = R.out_off[p];
                Q.in_hpad[p] = R.in_hpad[p]; Q.in_vpad[p] = R.in_vpad[p]; Q.out_hpad[p] = R.out_hpad[p]; Q.out_vpad[p] = R.out_vpad[p];
                if ((R.in_hpad[p] * bps) % 4 || (R.out_hpad[p] * bps) % 4) ok = false;
                if (L == 0) {
                    if (src_pitch[p] % 4) ok = false;
                    for (int f = 0; ok && f < nframes; f++) if ((uintptr_t)src[f * 3 + p] % 4) ok = false;
                }
                for (int f = 0; ok && f < nframes; f++) if ((uintptr_t)dst[f * 3 + p] % 16) ok = false;
                const int iw = R.in_w[p], ih = R.in_h[p], ow = R.out_w[p], oh = R.out_h[p];
                int last = (iw - 2 * NS - 4) / 2; // the window 2*x0 - 4 .. 2*x0 + 2*NS + 3 ends inside the input row
                if (ow - NS - 1 < last) last = ow - NS - 1; // and the thread's last output is not the right edge
                // threads sit at padded columns that are multiples of NS: x0 = k * NS - hpad; the first with x0 >= 2 (window inside the row, not the left edge)
                const int hp = R.out_hpad[p], xa = (2 + hp + NS - 1) / NS * NS - hp;
                Q.XA[p] = xa; Q.XB[p] = (iw >= 2 * NS + 4 && last >= xa) ? (last + hp) / NS * NS - hp + NS : xa;
                int yb = oh - 1; // rows 1 .. oh - 2 whose taps 2y - 2 .. 2y + 3 exist
                if ((ih - 4) / 2 + 1 < yb) yb = (ih - 4) / 2 + 1;
                Q.YA[p] = 1; Q.YB[p] = (ih >= 4 && yb > 1) ? yb : 1;
                if (Q.XB[p] <= Q.XA[p] || Q.YB[p] <= Q.YA[p]) { Q.XB[p] = Q.XA[p]; Q.YB[p] = Q.YA[p]; }
                if (Q.XB[p] > Q.XA[p] && Q.XB[p] + hp > qw) qw = Q.XB[p] + hp;
                if (Q.YB[p] - Q.YA[p] > qh) qh = Q.YB[p] - Q.YA[p];
            }
            if (ok && qw > 0 && qh > 0) {
                for (int p = 0; p < si.num_planes; p++) { R.XA[p] = Q.XA[p]; R.XB[p] = Q.XB[p]; R.YA[p] = Q.YA[p]; R.YB[p] = Q.YB[p]; }
                dim3 gq((qw / NS + 63) / 64, (qh + 3) / 4, nframes * 3);
#define RR(T, F, FS) hipLaunchKernelGGL((super_reduce_rows_kernel<T, F, FS>), gq, dim3(256), 0, st, Q)
#define RRF(T, FS) do { switch (si.rfilter) { case 0: RR(T, 0, FS); break; case 1: RR(T, 1, FS); break; case 2: RR(T, 2, FS); break; case 3: RR(T, 3, FS); break; default: RR(T, 4, FS); break; } } while (0)
                if (L == 0) { if (u8) RRF(uint8_t, true); else RRF(uint16_t, true); }
                else { if (u8) RRF(uint8_t, false); else RRF(uint16_t, false); }
#undef RRF
#undef RR
            }
        }
        int gtot = 0;
        for (int p = 0; p < si.num_planes; p++) {
            const int pw = R.out_w[p] + 2 * R.out_hpad[p], ph = R.out_h[p] + 2 * R.out_vpad[p];
            const int ncb = (pw + 63) / 64, nrb = (ph + 3) / 4;
            int cbA = (R.out_hpad[p] + R.XA[p] + 63) / 64, cbB = (R.out_hpad[p] + R.XB[p]) / 64;
            int rbA = (R.out_vpad[p] + R.YA[p] + 3) / 4, rbB = (R.out_vpad[p] + R.YB[p]) / 4;
            if (R.XB[p] <= R.XA[p] || R.YB[p] <= R.YA[p] || cbB <= cbA || rbB <= rbA) { cbA = cbB = 0; rbA = rbB = nrb; } // no interior: every block is a band block
            R.ncb[p] = ncb; R.nrb[p] = nrb; R.cbA[p] = cbA; R.cbB[p] = cbB; R.rbA[p] = rbA; R.rbB[p] = rbB;
            R.nfull[p] = (rbA + nrb - rbB) * ncb;
            R.ntotal[p] = R.nfull[p] + (rbB - rbA) * (cbA + ncb - cbB);
            if (R.ntotal[p] > gtot) gtot = R.ntotal[p];
        }
        dim3 g3(gtot, 1, nframes * 3);
        if (L == 0) { if (u8) launch_reduce<uint8_t, true>(R, si.rfilter, g3, st); else launch_reduce<uint16_t, true>(R, si.rfilter, g3, st); }
        else { if (u8) launch_reduce<uint8_t, false>(R, si.rfilter, g3, st); else launch_reduce<uint16_t, false>(R, si.rfilter, g3, st); }
    }
    HIP_CHECK(hipGetLastError());
    if (shadow_stride) { // what the level-0 kernels did not write themselves: the coarser levels, or everything
        long long begin[2] = { 0, 0 };
        if (fusedShadow)
            for (int p = 0; p < si.num_planes && p < 2; p++) begin[p] = 4LL * A.g[p].ph * dst_pitch[p];
        return shadow_launch(si, nframes, (void *const *)ddst, dst_pitch, shadow_stride, begin, st);
    }
    return MVX_OK;
}

// ================================================================================================ mv.Finest
// MVFinest.c:48-140, Merge4PlanesToBig / Merge16PlanesToBig MaskFun.cpp:206-330: interleave the pel^2 sub-pel planes of level 0.
struct FinestArgs { const unsigned char *src; unsigned char *dst; long long srcPitch, dstPitch, planeStride; int pw, ph, pel, logPel; };

template <typename T> __global__ __launch_bounds__(256) void finest_kernel(FinestArgs A) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= A.pw * A.pel || y >= A.ph * A.pel) return;
    const int m = A.pel - 1, idx = (x & m) | ((y & m) << A.logPel);
    const T v = ((const T *)(A.src + idx * A.planeStride + (long long)(y >> A.logPel) * A.srcPitch))[x >> A.logPel];
    ((T *)(A.dst + (long long)y * A.dstPitch))[x] = v;
}

extern "C" __attribute__((visibility("default"))) void mvx_finest_size(const mvx_super *s, int32_t *width, int32_t *height) {
    *width = (s->info.width + 2 * s->info.hpad) * s->info.pel;
    *height = (s->info.height + 2 * s->info.vpad) * s->info.pel;
}

extern "C" __attribute__((visibility("default"))) int mvx_finest_frames(const mvx_super *s, int nframes, const void *const *super_frames, const ptrdiff_t super_pitch[3],
                                                                        void *const *dst, const ptrdiff_t dst_pitch[3], void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const mvx_super_info &si = s->info;
    const int logPel = si.pel == 4 ? 2 : si.pel == 2 ? 1 : 0;
    for (int f = 0; f < nframes; f++)
        for (int p = 0; p < si.num_planes; p++) {
            if (!(si.modeYUV & (1 << p))) continue; // planes the super clip does not carry stay untouched (MVFinest.c:87 pPlanes[i] == NULL)
            LevelPlane lp;
            mvx_level_plane(si, 0, p, super_pitch[p], &lp);
            FinestArgs A = { (const unsigned char *)super_frames[f * 3 + p] + lp.off, (unsigned char *)dst[f * 3 + p], (long long)super_pitch[p], (long long)dst_pitch[p],
                             (long long)super_pitch[p] * lp.ph, lp.pw, lp.ph, si.pel, logPel };
            dim3 grid((lp.pw * si.pel + 63) / 64, (lp.ph * si.pel + 3) / 4);
            if (si.bits <= 8) hipLaunchKernelGGL(finest_kernel<uint8_t>, grid, dim3(256), 0, st, A);
            else hipLaunchKernelGGL(finest_kernel<uint16_t>, grid, dim3(256), 0, st, A);
        }
    HIP_CHECK(hipGetLastError());
    return MVX_OK;
}

// mvx_super_rows.h -- level 0 of mv.Super (pel 2) with the source rows held in registers.
//
// super_level0_kernel stages a source tile in LDS sample by sample and reads every filter tap back from LDS; it runs at a
// quarter of the HBM roofline.  Here one thread owns NS consecutive samples of R output rows of the padded plane and of its
// H / V / HV half-pel planes: it loads the R + 5 source rows it needs as 16-byte vectors straight into registers (all loads
// requested before the first is used), filters out of registers (every tap index is a compile-time constant) and writes
// 16-byte vectors.  With the shadow planes of a 16-bit clip (mvx_super.hip) it also writes the shifted luma copy / the
// UV-interleaved chroma plane, so that those bytes are never read back.
//
// It covers the columns [XA, XB) of the padded plane in which a thread's window X-4 .. X+NS+3 lies inside the source row
// (no horizontal clamping, the horizontal edge rules of MVFrame.cpp:1071-1111 never apply); the host gives the two strips
// left and right of it (hpad + a few samples each) to super_level0_kernel.  Rows are clamped per row (scalar work), the
// vertical edge rules (MVFrame.cpp:1019-1068 / :1115-1150 / :508-527) are wave-uniform branches: a wave is 64 threads of one row group.
#pragma once

#define SR_GL __attribute__((address_space(1)))
typedef unsigned sr_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned sr_u2 __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned sr_u4a __attribute__((ext_vector_type(4)));

template <typename T> struct SrGeo;
template <> struct SrGeo<uint16_t> { static constexpr int NS = 8, WN = 16, RAW = 8; };
template <> struct SrGeo<uint8_t> { static constexpr int NS = 16, WN = 24, RAW = 6; };

template <typename T> __device__ __forceinline__ void sr_load(SR_GL const unsigned char *p, unsigned (&d)[SrGeo<T>::RAW]) {
    const sr_u4 a = *(SR_GL const sr_u4 *)p;
    d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3];
    if (sizeof(T) == 2) { const sr_u4 b = *(SR_GL const sr_u4 *)(p + 16); d[4] = b[0]; d[5] = b[1]; d[6] = b[2]; d[7] = b[3]; }
    else { const sr_u2 b = *(SR_GL const sr_u2 *)(p + 16); d[4] = b[0]; d[5] = b[1]; }
}
// sample j of a window (j is a compile-time constant at every call site)
template <typename T> __device__ __forceinline__ int sr_get(const unsigned (&d)[SrGeo<T>::RAW], int j) {
    if (sizeof(T) == 2) return (j & 1) ? (int)(d[j >> 1] >> 16) : (int)(d[j >> 1] & 0xffffu);
    return (int)((d[j >> 2] >> (8 * (j & 3))) & 0xffu);
}
// the interior form of the half-sample rule (same arithmetic as half_rule<SHARP>)
template <int SHARP> __device__ __forceinline__ int sr_taps(int m0, int m1, int m2, int m3, int m4, int m5, int pm) {
    if (SHARP == 0) return (m2 + m3 + 1) >> 1;
    if (SHARP == 1) { int v = (-(m1 + m4) + (m2 + m3) * 9 + 8) >> 4; return min(pm, max(0, v)); }
    m2 = (m2 + m3) * 4; m2 -= m1 + m4; m2 *= 5; m0 += m5 + m2 + 16; m0 >>= 5;
    return max(0, min(m0, pm));
}
// NS samples -> 16 bytes
template <typename T> __device__ __forceinline__ sr_u4a sr_pack(const int *v) {
    sr_u4a o;
    if (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (unsigned)v[2 * i] | ((unsigned)v[2 * i + 1] << 16);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (unsigned)v[4 * i] | ((unsigned)v[4 * i + 1] << 8) | ((unsigned)v[4 * i + 2] << 16) | ((unsigned)v[4 * i + 3] << 24);
    }
    return o;
}

struct SuperRowsArgs {
    const void *const *src; // [nframes*3]
    void *const *dst;       // [nframes*3]
    SuperPlaneGeom g[3];
    int XA[3], XB[3];       // columns [XA, XB) of the padded plane, multiples of NS
    long long shadow[3];    // byte distance from plane p to its shadow data (KIND 1: plane 0; KIND 2: plane 1)
    int bits, firstPlane, nz, modeYUV; // blockIdx.z = frame * nz + i, plane = firstPlane + i (KIND 2: nz 1, planes 1 and 2 together)
};

// KIND 0: one plane.  KIND 1: one plane + its copy shifted left by one sample.  KIND 2: U and V (same geometry) + the UV-interleaved plane.
template <typename T, int SHARP, int KIND, int R>
__global__ __launch_bounds__(256) void super_rows_kernel(SuperRowsArgs A) {
    typedef SrGeo<T> G;
    constexpr int NS = G::NS, NR = R + 5, NO = NS + (KIND == 1 ? 1 : 0);
    const int z = blockIdx.z, f = z / A.nz, p = A.firstPlane + z % A.nz;
    if (!(A.modeYUV & (1 << p))) return;
    const SuperPlaneGeom g = A.g[p];
    // a wave's 64 threads span 1 KB of a row starting at a multiple of 1 KB: its stores are whole 128-byte lines (measured,
    // tools/micro/write_bw.hip: 1 KB pieces that start 48 bytes into a line stream at 2.85 TB/s, line-aligned ones at 4.8)
    const int X = (blockIdx.x * 64 + (threadIdx.x & 63)) * NS;
    const int Y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * R;
    if (X < A.XA[p] || X >= A.XB[p] || Y0 >= g.ph) return;
    const int pm = (1 << A.bits) - 1;
    const long long planeStride = g.dst_pitch * g.ph;
    constexpr int E0 = SHARP == 1 ? 1 : 2, E1 = SHARP == 1 ? 3 : 4;
    sr_u4a keep[KIND == 2 ? R : 1][4]; // KIND 2: the U outputs until V is there

#pragma unroll
    for (int c = 0; c < (KIND == 2 ? 2 : 1); c++) {
        SR_GL const unsigned char *src = (SR_GL const unsigned char *)(unsigned long long)A.src[f * 3 + p + c] + (long long)(X - 4 - g.hpad) * (long long)sizeof(T);
        SR_GL unsigned char *dst = (SR_GL unsigned char *)(unsigned long long)A.dst[f * 3 + p + c];
        unsigned raw[NR][G::RAW];
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const int py = iclamp(Y0 - 2 + i, 0, g.ph - 1), sy = iclamp(py - g.vpad, 0, g.h - 1);
            sr_load<T>(src + (long long)sy * g.src_pitch, raw[i]);
        }
#define S_(i, j) sr_get<T>(raw[i], (j))
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int Y = Y0 + r;
            if (Y >= g.ph) break;
            int o0[NO], oH[NO], oV[NO], oHV[NO], vv[NO + 5]; // vv[j - 2]: the V plane at window column j = 2 .. NO + 6
            const bool copyV = Y == g.ph - 1, bilV = SHARP == 0 || Y < E0 || Y >= g.ph - E1;
            if (copyV) {
#pragma unroll
                for (int j = 2; j < NO + 7; j++) vv[j - 2] = S_(r + 2, j);
            } else if (bilV) {
#pragma unroll
                for (int j = 2; j < NO + 7; j++) vv[j - 2] = (S_(r + 2, j) + S_(r + 3, j) + 1) >> 1;
            } else {
#pragma unroll
                for (int j = 2; j < NO + 7; j++) vv[j - 2] = sr_taps<SHARP>(S_(r, j), S_(r + 1, j), S_(r + 2, j), S_(r + 3, j), S_(r + 4, j), S_(r + 5, j), pm);
            }
#pragma unroll
            for (int k = 0; k < NO; k++) {
                const int j = k + 4;
                o0[k] = S_(r + 2, j);
                oH[k] = sr_taps<SHARP>(S_(r + 2, j - 2), S_(r + 2, j - 1), S_(r + 2, j), S_(r + 2, j + 1), S_(r + 2, j + 2), S_(r + 2, j + 3), pm);
                oV[k] = vv[j - 2];
                if (SHARP == 0) { // DiagonalBilinear MVFrame.cpp:551-572 (X < pw - 1 throughout [XA, XB))
                    const int a = S_(r + 2, j), b = S_(r + 2, j + 1), cc = S_(r + 3, j), d = S_(r + 3, j + 1);
                    oHV[k] = copyV ? (a + b + 1) >> 1 : (a + b + cc + d + 2) >> 2;
                } else
                    oHV[k] = sr_taps<SHARP>(vv[j - 4], vv[j - 3], vv[j - 2], vv[j - 1], vv[j], vv[j + 1], pm);
            }
            const long long off = (long long)Y * g.dst_pitch + (long long)X * (long long)sizeof(T);
            const sr_u4a q0 = sr_pack<T>(o0), qH = sr_pack<T>(oH), qV = sr_pack<T>(oV), qHV = sr_pack<T>(oHV);
            *(SR_GL sr_u4a *)(dst + off) = q0;
            *(SR_GL sr_u4a *)(dst + planeStride + off) = qH;
            *(SR_GL sr_u4a *)(dst + 2 * planeStride + off) = qV;
            *(SR_GL sr_u4a *)(dst + 3 * planeStride + off) = qHV;
            if (KIND == 1) { // the copy shifted left by one sample: samples X+1 .. X+NS at the same byte offset
                SR_GL unsigned char *sh = dst + A.shadow[p];
                *(SR_GL sr_u4a *)(sh + off) = sr_pack<T>(o0 + (KIND == 1 ? 1 : 0));
                *(SR_GL sr_u4a *)(sh + planeStride + off) = sr_pack<T>(oH + (KIND == 1 ? 1 : 0));
                *(SR_GL sr_u4a *)(sh + 2 * planeStride + off) = sr_pack<T>(oV + (KIND == 1 ? 1 : 0));
                *(SR_GL sr_u4a *)(sh + 3 * planeStride + off) = sr_pack<T>(oHV + (KIND == 1 ? 1 : 0));
            }
            if (KIND == 2) {
                if (c == 0) { keep[r][0] = q0; keep[r][1] = qH; keep[r][2] = qV; keep[r][3] = qHV; }
                else { // 16-bit samples: dword k of a plane holds samples 2k, 2k+1 -> (U 2k | V 2k), (U 2k+1 | V 2k+1)
                    SR_GL unsigned char *uv = (SR_GL unsigned char *)(unsigned long long)A.dst[f * 3 + p] + A.shadow[p];
                    const sr_u4a qq[4] = { q0, qH, qV, qHV };
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const sr_u4a u = keep[r][s], v = qq[s];
                        sr_u4a a, b;
                        a[0] = (u[0] & 0xffffu) | (v[0] << 16); a[1] = (u[0] >> 16) | (v[0] & 0xffff0000u);
                        a[2] = (u[1] & 0xffffu) | (v[1] << 16); a[3] = (u[1] >> 16) | (v[1] & 0xffff0000u);
                        b[0] = (u[2] & 0xffffu) | (v[2] << 16); b[1] = (u[2] >> 16) | (v[2] & 0xffff0000u);
                        b[2] = (u[3] & 0xffffu) | (v[3] << 16); b[3] = (u[3] >> 16) | (v[3] & 0xffff0000u);
                        SR_GL unsigned char *d = uv + 2 * (s * planeStride + off);
                        *(SR_GL sr_u4a *)d = a; *(SR_GL sr_u4a *)(d + 16) = b;
                    }
                }
            }
        }
#undef S_
    }
}

// Shadow data of the two strips the rows kernel leaves to super_level0_kernel: columns [0, XA) and [XB, pw) of the four level-0
// planes.  One thread per sample; a block is 64 samples x 4 rows of the four stacked planes, blockIdx.z = frame * 2 + kind.
struct ShadowStripArgs { void *const *planes; SuperPlaneGeom g[3]; int XA[3], XB[3]; long long shadow[3]; int nplanes; };
__global__ __launch_bounds__(256) void super_shadow_strip_kernel(ShadowStripArgs A) {
    const int f = blockIdx.z >> 1, kind = blockIdx.z & 1, p = kind;
    if (p >= A.nplanes || (kind == 1 && A.nplanes < 3)) return;
    const SuperPlaneGeom g = A.g[p];
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6); // 0 .. 4 * ph - 1
    if (row >= 4 * g.ph) return;
    const int nl = A.XA[p], n = nl + (g.pw - A.XB[p]);
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    if (t >= n) return;
    const int x = t < nl ? t : A.XB[p] + (t - nl);
    const long long ro = (long long)row * g.dst_pitch;
    if (kind == 0) {
        unsigned char *base = (unsigned char *)A.planes[f * 3];
        const unsigned short *s = (const unsigned short *)(base + ro);
        ((unsigned short *)(base + A.shadow[0] + ro))[x] = x + 1 < g.pw ? s[x + 1] : (unsigned short)0;
    } else {
        unsigned char *bu = (unsigned char *)A.planes[f * 3 + 1];
        const unsigned short u = ((const unsigned short *)(bu + ro))[x], v = ((const unsigned short *)((const unsigned char *)A.planes[f * 3 + 2] + ro))[x];
        ((unsigned *)(bu + A.shadow[1] + 2 * ro))[x] = (unsigned)u | ((unsigned)v << 16);
    }
}

// ---- 2x reduction, interior outputs with the input rows in registers --------------------------------------------------------
// super_reduce_kernel evaluates the separable decimation filter per output sample with scalar loads (16 to 36 of them).  Here a
// thread owns 16 bytes of one output row: it loads the 2 to 6 input rows it needs as vectors (window 2*x0 - 4 .. 2*x0 + 2*NS + 3),
// filters vertically, then horizontally.  It covers the interior outputs x in [XA, XB), y in [YA, YB) for which no edge rule of
// MVFrame.cpp:575-1014 applies and the window lies inside the input rows; the host leaves the rest (edges, the padding) to
// super_reduce_kernel.
template <typename T> struct SdGeo;
template <> struct SdGeo<uint16_t> { static constexpr int NS = 8, WN = 24, RAW = 12; };
template <> struct SdGeo<uint8_t> { static constexpr int NS = 16, WN = 40, RAW = 10; };

template <typename T> __device__ __forceinline__ void sd_load(SR_GL const unsigned char *p, unsigned (&d)[SdGeo<T>::RAW]) {
    const sr_u4 a = *(SR_GL const sr_u4 *)p, b = *(SR_GL const sr_u4 *)(p + 16);
    d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3]; d[4] = b[0]; d[5] = b[1]; d[6] = b[2]; d[7] = b[3];
    if (sizeof(T) == 2) { const sr_u4 c = *(SR_GL const sr_u4 *)(p + 32); d[8] = c[0]; d[9] = c[1]; d[10] = c[2]; d[11] = c[3]; }
    else { const sr_u2 c = *(SR_GL const sr_u2 *)(p + 32); d[8] = c[0]; d[9] = c[1]; }
}
template <typename T> __device__ __forceinline__ int sd_get(const unsigned (&d)[SdGeo<T>::RAW], int j) {
    if (sizeof(T) == 2) return (j & 1) ? (int)(d[j >> 1] >> 16) : (int)(d[j >> 1] & 0xffffu);
    return (int)((d[j >> 2] >> (8 * (j & 3))) & 0xffu);
}
// the interior form of one pass of the decimation filter: taps at positions 2i-2 .. 2i+3
template <int RF> __device__ __forceinline__ int sd_taps(int m0, int m1, int m2, int m3, int m4, int m5) {
    if (RF == 1) return (m1 + m2 * 2 + m3 + 2) >> 2;
    if (RF == 2) return (m1 + (m2 + m3) * 3 + m4 + 4) >> 3;
    if (RF == 3) { m2 = (m2 + m3) * 22; m1 = (m1 + m4) * 9; m0 += m5 + m2 + m1 + 32; return m0 >> 6; }
    m2 = (m2 + m3) * 10; m1 = (m1 + m4) * 5; m0 += m5 + m2 + m1 + 16; return m0 >> 5;
}

struct SuperReduceRowsArgs {
    const void *const *src; // FROM_SRC: source frames [nframes*3]
    void *const *dst;
    long long src_pitch[3], dst_pitch[3], in_off[3], out_off[3];
    int in_hpad[3], in_vpad[3], out_hpad[3], out_vpad[3];
    int XA[3], XB[3], YA[3], YB[3]; // interior output coordinates; XA + out_hpad and XB + out_hpad are multiples of NS
    int modeYUV, nplanes;
};

template <typename T, int RF, bool FROM_SRC>
__global__ __launch_bounds__(256) void super_reduce_rows_kernel(SuperReduceRowsArgs A) {
    typedef SdGeo<T> G;
    constexpr int NS = G::NS;
    constexpr int R0 = RF == 0 ? 0 : (RF <= 2 ? -1 : -2), NRW = RF == 0 ? 2 : (RF == 1 ? 3 : (RF == 2 ? 4 : 6)); // input rows 2y + R0 .. 2y + R0 + NRW - 1
    const int z = blockIdx.z, f = z / 3, p = z % 3;
    if (p >= A.nplanes || !(A.modeYUV & (1 << p))) return;
    // threads are laid out over the PADDED output row, so that a wave stores whole 128-byte lines (see super_rows_kernel)
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * NS - A.out_hpad[p];
    const int y = A.YA[p] + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 < A.XA[p] || x0 >= A.XB[p] || y >= A.YB[p]) return;
    SR_GL unsigned char *dplane = (SR_GL unsigned char *)(unsigned long long)A.dst[f * 3 + p];
    SR_GL const unsigned char *sbase;
    long long sp;
    if (FROM_SRC) { sbase = (SR_GL const unsigned char *)(unsigned long long)A.src[f * 3 + p]; sp = A.src_pitch[p]; }
    else { sp = A.dst_pitch[p]; sbase = dplane + A.in_off[p] + (long long)A.in_vpad[p] * sp + (long long)A.in_hpad[p] * (long long)sizeof(T); }
    sbase += (long long)(2 * x0 - 4) * (long long)sizeof(T);
    unsigned raw[NRW][G::RAW];
#pragma unroll
    for (int i = 0; i < NRW; i++) sd_load<T>(sbase + (long long)(2 * y + R0 + i) * sp, raw[i]);
#define S_(i, j) sd_get<T>(raw[i], (j))
    int o[NS];
    if (RF == 0) {
#pragma unroll
        for (int k = 0; k < NS; k++) o[k] = (S_(0, 2 * k + 4) + S_(0, 2 * k + 5) + S_(1, 2 * k + 5) + S_(1, 2 * k + 4) + 2) >> 2;
    } else {
        int vv[2 * NS + 4]; // vertical pass at window columns 2 .. 2*NS + 5
#pragma unroll
        for (int j = 2; j < 2 * NS + 6; j++) {
            if (RF <= 2) vv[j - 2] = sd_taps<RF>(0, S_(0, j), S_(1, j), S_(2, j), RF == 2 ? S_(NRW - 1, j) : 0, 0);
            else vv[j - 2] = sd_taps<RF>(S_(0, j), S_(1, j), S_(2, j), S_(3, j), S_(NRW - 2, j), S_(NRW - 1, j));
        }
#pragma unroll
        for (int k = 0; k < NS; k++) o[k] = sd_taps<RF>(vv[2 * k], vv[2 * k + 1], vv[2 * k + 2], vv[2 * k + 3], vv[2 * k + 4], vv[2 * k + 5]);
    }
#undef S_
    *(SR_GL sr_u4a *)(dplane + A.out_off[p] + (long long)(y + A.out_vpad[p]) * A.dst_pitch[p] + (long long)(x0 + A.out_hpad[p]) * (long long)sizeof(T)) = sr_pack<T>(o);
}

// ref_shim.cpp -- thin C entry points into the reference's own object code (oracle/_ref build, test infrastructure).
// Nothing here restates reference logic: each function forwards to a symbol compiled from /root/reference/src.
#include <cstdint>
#include <cstring>
extern "C" {
#include "SADFunctions.h"
#include "Overlap.h"
#include "CopyCode.h"
#include "Luma.h"
#include "SimpleResize.h"
}
// SimpleResize_AVX2.cpp (C++ linkage there)
void simpleResize_uint8_t_avx2(const SimpleResize *simple, uint8_t *dstp, int dst_stride, const uint8_t *srcp, int src_stride, int horizontal_vectors);

// MVFrame_AVX2.cpp (C++ linkage there)
void Average2_avx2(uint8_t *pDst, const uint8_t *pSrc1, const uint8_t *pSrc2, intptr_t nPitch, intptr_t nWidth, intptr_t nHeight);
void VerticalBilinear_avx2(uint8_t *pDst, const uint8_t *pSrc, intptr_t nPitch, intptr_t nWidth, intptr_t nHeight, intptr_t bitsPerSample);
void HorizontalBilinear_avx2(uint8_t *pDst, const uint8_t *pSrc, intptr_t nPitch, intptr_t nWidth, intptr_t nHeight, intptr_t bitsPerSample);
void DiagonalBilinear_avx2(uint8_t *pDst, const uint8_t *pSrc, intptr_t nPitch, intptr_t nWidth, intptr_t nHeight, intptr_t bitsPerSample);
void VerticalWiener_avx2(uint8_t *pDst, const uint8_t *pSrc, intptr_t nPitch, intptr_t nWidth, intptr_t nHeight, intptr_t bitsPerSample);
void HorizontalWiener_avx2(uint8_t *pDst, const uint8_t *pSrc, intptr_t nPitch, intptr_t nWidth, intptr_t nHeight, intptr_t bitsPerSample);

extern "C" {

unsigned ref_sad(int w, int h, int bits, int avx2, const uint8_t *s, intptr_t sp, const uint8_t *r, intptr_t rp) {
    SADFunction f = nullptr;
    if (avx2) f = selectSADFunctionAVX2(w, h, bits);
    if (!f) f = selectSADFunction(w, h, bits, 0, 0);
    return f(s, sp, r, rp);
}
int ref_has_sad_avx2(int w, int h, int bits) { return selectSADFunctionAVX2(w, h, bits) != nullptr; }

unsigned ref_satd(int w, int h, int bits, const uint8_t *s, intptr_t sp, const uint8_t *r, intptr_t rp) {
    return selectSATDFunction(w, h, bits, 0, 0)(s, sp, r, rp);
}

void ref_over_init(int16_t *win9, int nx, int ny, int ox, int oy) {
    OverlapWindows ow;
    overInit(&ow, nx, ny, ox, oy);
    memcpy(win9, ow.Overlap9Windows, sizeof(int16_t) * 9 * nx * ny);
    overDeinit(&ow);
}

void ref_overlaps(int w, int h, int bits, int avx2, uint8_t *dst, intptr_t dp, const uint8_t *src, intptr_t sp, int16_t *win, intptr_t wp) {
    OverlapsFunction f = nullptr;
    if (avx2) f = selectOverlapsFunctionAVX2(w, h, bits);
    if (!f) f = selectOverlapsFunction(w, h, bits, 0);
    f(dst, dp, src, sp, win, wp);
}

void ref_to_pixels(int bits, uint8_t *dst, int dp, const uint8_t *src, int sp, int w, int h) {
    if (bits <= 8) ToPixels_uint16_t_uint8_t(dst, dp, src, sp, w, h, bits);
    else ToPixels_uint32_t_uint16_t(dst, dp, src, sp, w, h, bits);
}

void ref_copy(int w, int h, int bits, uint8_t *dst, intptr_t dp, const uint8_t *src, intptr_t sp) {
    selectCopyFunction(w, h, bits)(dst, dp, src, sp);
}

unsigned ref_luma(int w, int h, int bits, const uint8_t *src, intptr_t sp) { return selectLumaFunction(w, h, bits, 0)(src, sp); }

// kind as in mvo_refine_plane: 0 H-bilinear 1 V-bilinear 2 D-bilinear 5 H-wiener 6 V-wiener (8-bit AVX2 kernels only)
int ref_refine_avx2(int kind, uint8_t *dst, const uint8_t *src, intptr_t pitch, intptr_t w, intptr_t h) {
    switch (kind) {
    case 0: HorizontalBilinear_avx2(dst, src, pitch, w, h, 8); return 0;
    case 1: VerticalBilinear_avx2(dst, src, pitch, w, h, 8); return 0;
    case 2: DiagonalBilinear_avx2(dst, src, pitch, w, h, 8); return 0;
    case 5: HorizontalWiener_avx2(dst, src, pitch, w, h, 8); return 0;
    case 6: VerticalWiener_avx2(dst, src, pitch, w, h, 8); return 0;
    }
    return -1;
}
void ref_average2_avx2(uint8_t *dst, const uint8_t *a, const uint8_t *b, intptr_t pitch, intptr_t w, intptr_t h) { Average2_avx2(dst, a, b, pitch, w, h); }


// the reference's AVX2 mask upsizer on caller-supplied offset / weight tables (SimpleResize.cpp's InitTables needs <VSHelper.h>
// and is not built: the tables come from the oracle's restatement, the resampling arithmetic is the reference's object code)
void ref_simple_resize_u8_avx2(uint8_t *dst, int dst_stride, const uint8_t *src, int src_stride, int dw, int dh, int sw, int sh,
                               int *voff, int *vw, int *hoff, int *hw) {
    SimpleResize s;
    memset(&s, 0, sizeof(s));
    s.dst_width = dw; s.dst_height = dh; s.src_width = sw; s.src_height = sh;
    s.vertical_offsets = voff; s.vertical_weights = vw; s.horizontal_offsets = hoff; s.horizontal_weights = hw;
    simpleResize_uint8_t_avx2(&s, dst, dst_stride, src, src_stride, 0);
}

}

/*
 * mvo_super.c -- oracle restatement of mv.Super (test infrastructure only, see mvoracle.h).
 * Follows /root/reference/src/MVSuper.c and MVFrame.cpp; line numbers cited per function.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvo_internal.h"

/* ------------------------------------------------------------------ geometry */

/* MVFrame.cpp:1209-1216 */
int mvo_plane_height_luma(int src_height, int level, int yRatioUV, int vpad) {
    int height = src_height;
    for (int i = 1; i <= level; i++)
        height = vpad >= yRatioUV ? ((height / yRatioUV + 1) / 2) * yRatioUV : ((height / yRatioUV) / 2) * yRatioUV;
    return height;
}

/* MVFrame.cpp:1219-1226 */
int mvo_plane_width_luma(int src_width, int level, int xRatioUV, int hpad) {
    int width = src_width;
    for (int i = 1; i <= level; i++)
        width = hpad >= xRatioUV ? ((width / xRatioUV + 1) / 2) * xRatioUV : ((width / xRatioUV) / 2) * xRatioUV;
    return width;
}

/* MVFrame.cpp:1229-1247 */
unsigned mvo_plane_super_offset(int chroma, int src_height, int level, int pel, int vpad, int plane_pitch, int yRatioUV) {
    int height = src_height;
    unsigned offset;
    if (level == 0)
        offset = 0;
    else {
        offset = pel * pel * plane_pitch * (src_height + vpad * 2);
        for (int i = 1; i < level; i++) {
            height = chroma ? mvo_plane_height_luma(src_height * yRatioUV, i, yRatioUV, vpad * yRatioUV) / yRatioUV
                            : mvo_plane_height_luma(src_height, i, yRatioUV, vpad);
            offset += plane_pitch * (height + vpad * 2);
        }
    }
    return offset;
}

/* MVFrame.cpp:1327-1343 mvpInit, :1764-1787 mvfInit, :1852-1878 mvgofInit */
static void plane_init(mvo_plane *m, int w, int h, int pel, int hpad, int vpad, int bits) {
    memset(m, 0, sizeof(*m));
    m->w = w; m->h = h; m->pel = pel; m->hpad = hpad; m->vpad = vpad; m->bits = bits;
    m->bps = (bits + 7) / 8;
    m->pw = w + 2 * hpad;
    m->ph = h + 2 * vpad;
}

static void frame_init(mvo_frame *f, int w, int h, int pel, int hpad, int vpad, int mode, int xr, int yr, int bits) {
    f->mode = mode;
    plane_init(&f->pl[0], w, h, pel, hpad, vpad, bits);
    plane_init(&f->pl[1], w / xr, h / yr, pel, hpad / xr, vpad / yr, bits);
    plane_init(&f->pl[2], w / xr, h / yr, pel, hpad / xr, vpad / yr, bits);
}

void mvo_gof_init(mvo_gof *g, int levels, int w, int h, int pel, int hpad, int vpad, int mode, int xr, int yr, int bits) {
    g->nlevels = levels;
    frame_init(&g->fr[0], w, h, pel, hpad, vpad, mode, xr, yr, bits);
    for (int i = 1; i < levels; i++) {
        int wi = mvo_plane_width_luma(w, i, xr, hpad);
        int hi = mvo_plane_height_luma(h, i, yr, vpad);
        frame_init(&g->fr[i], wi, hi, 1, hpad, vpad, mode, xr, yr, bits); /* coarse levels keep full padding: :1876 */
    }
}

/* MVFrame.cpp:1356-1364 mvpUpdate, :1892-1903 mvgofUpdate (plane index doubles as the chroma flag :1898) */
static void gof_update_yr(mvo_gof *g, uint8_t *const planes[3], const int pitch[3], int yr) {
    const mvo_frame *f0 = &g->fr[0];
    for (int i = 0; i < g->nlevels; i++) {
        for (int p = 0; p < 3; p++) {
            mvo_plane *m = &g->fr[i].pl[p];
            if (!planes[p] || !(g->fr[i].mode & (1 << p))) { m->p[0] = NULL; continue; }
            m->pitch = pitch[p];
            m->offpad = m->pitch * m->vpad + m->hpad * m->bps;
            uint8_t *base = planes[p] + mvo_plane_super_offset(p, f0->pl[p].h, i, f0->pl[0].pel, f0->pl[p].vpad, pitch[p], yr);
            for (int k = 0; k < m->pel * m->pel; k++)
                m->p[k] = base + (size_t)k * m->pitch * m->ph;
        }
    }
}

/* exported for the other oracle TUs */
void mvo_gof_update(mvo_gof *g, uint8_t *const planes[3], const int pitch[3], int yr) { gof_update_yr(g, planes, pitch, yr); }

/* ------------------------------------------------------------------ pixel kernels (templated on sample type) */

#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))

#define DEFINE_KERNELS(T, SFX)                                                                                         \
    /* MVFrame.cpp:1264-1318 PadReferenceFrame */                                                                      \
    static void pad_##SFX(uint8_t *ref8, int pitch, int hp, int vp, int w, int h) {                                    \
        pitch /= (int)sizeof(T);                                                                                       \
        T *ref = (T *)ref8;                                                                                            \
        T *pf = ref + vp * pitch + hp;                                                                                 \
        for (int y = 0; y < h + 2 * vp; y++) {                                                                         \
            int sy = IMIN(IMAX(y - vp, 0), h - 1);                                                                     \
            T *row = ref + (size_t)y * pitch;                                                                          \
            const T *srow = pf + (size_t)sy * pitch;                                                                   \
            if (y < vp || y >= vp + h)                                                                                 \
                for (int x = 0; x < w; x++) row[hp + x] = srow[x];                                                     \
            T l = srow[0], r = srow[w - 1];                                                                            \
            for (int x = 0; x < hp; x++) { row[x] = l; row[hp + w + x] = r; }                                          \
        }                                                                                                              \
    }                                                                                                                  \
    /* MVFrame.cpp:508-527 */                                                                                          \
    static void vbilin_##SFX(uint8_t *d8, const uint8_t *s8, intptr_t pitch, intptr_t w, intptr_t h, int bits) {       \
        (void)bits; T *d = (T *)d8; const T *s = (const T *)s8; pitch /= sizeof(T);                                    \
        for (intptr_t j = 0; j < h - 1; j++) {                                                                         \
            for (intptr_t i = 0; i < w; i++) d[i] = (T)((s[i] + s[i + pitch] + 1) >> 1);                               \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t i = 0; i < w; i++) d[i] = s[i];                                                                  \
    }                                                                                                                  \
    /* MVFrame.cpp:530-548 */                                                                                          \
    static void hbilin_##SFX(uint8_t *d8, const uint8_t *s8, intptr_t pitch, intptr_t w, intptr_t h, int bits) {       \
        (void)bits; T *d = (T *)d8; const T *s = (const T *)s8; pitch /= sizeof(T);                                    \
        for (intptr_t j = 0; j < h; j++) {                                                                             \
            for (intptr_t i = 0; i < w - 1; i++) d[i] = (T)((s[i] + s[i + 1] + 1) >> 1);                               \
            d[w - 1] = s[w - 1];                                                                                       \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
    }                                                                                                                  \
    /* MVFrame.cpp:551-572 */                                                                                          \
    static void dbilin_##SFX(uint8_t *d8, const uint8_t *s8, intptr_t pitch, intptr_t w, intptr_t h, int bits) {       \
        (void)bits; T *d = (T *)d8; const T *s = (const T *)s8; pitch /= sizeof(T);                                    \
        for (intptr_t j = 0; j < h - 1; j++) {                                                                         \
            for (intptr_t i = 0; i < w - 1; i++) d[i] = (T)((s[i] + s[i + 1] + s[i + pitch] + s[i + pitch + 1] + 2) >> 2); \
            d[w - 1] = (T)((s[w - 1] + s[w + pitch - 1] + 1) >> 1);                                                    \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t i = 0; i < w - 1; i++) d[i] = (T)((s[i] + s[i + 1] + 1) >> 1);                                  \
        d[w - 1] = s[w - 1];                                                                                           \
    }                                                                                                                  \
    /* MVFrame.cpp:1019-1068 (1,-5,20,20,-5,1)/32 */                                                                   \
    static void vwiener_##SFX(uint8_t *d8, const uint8_t *s8, intptr_t pitch, intptr_t w, intptr_t h, int bits) {      \
        T *d = (T *)d8; const T *s = (const T *)s8; pitch /= sizeof(T);                                                \
        int pm = (1 << bits) - 1;                                                                                      \
        for (intptr_t j = 0; j < 2; j++) {                                                                             \
            for (intptr_t i = 0; i < w; i++) d[i] = (T)((s[i] + s[i + pitch] + 1) >> 1);                               \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t j = 2; j < h - 4; j++) {                                                                         \
            for (intptr_t i = 0; i < w; i++) {                                                                         \
                int m0 = s[i - pitch * 2], m1 = s[i - pitch], m2 = s[i], m3 = s[i + pitch], m4 = s[i + pitch * 2],     \
                    m5 = s[i + pitch * 3];                                                                             \
                m2 = (m2 + m3) * 4; m2 -= m1 + m4; m2 *= 5; m0 += m5 + m2 + 16; m0 >>= 5;                              \
                d[i] = (T)IMAX(0, IMIN(m0, pm));                                                                       \
            }                                                                                                          \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t j = h - 4; j < h - 1; j++) {                                                                     \
            for (intptr_t i = 0; i < w; i++) d[i] = (T)((s[i] + s[i + pitch] + 1) >> 1);                               \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t i = 0; i < w; i++) d[i] = s[i];                                                                  \
    }                                                                                                                  \
    /* MVFrame.cpp:1071-1111 */                                                                                        \
    static void hwiener_##SFX(uint8_t *d8, const uint8_t *s8, intptr_t pitch, intptr_t w, intptr_t h, int bits) {      \
        T *d = (T *)d8; const T *s = (const T *)s8; pitch /= sizeof(T);                                                \
        int pm = (1 << bits) - 1;                                                                                      \
        for (intptr_t j = 0; j < h; j++) {                                                                             \
            d[0] = (T)((s[0] + s[1] + 1) >> 1);                                                                        \
            d[1] = (T)((s[1] + s[2] + 1) >> 1);                                                                        \
            for (intptr_t i = 2; i < w - 4; i++) {                                                                     \
                int m0 = s[i - 2], m1 = s[i - 1], m2 = s[i], m3 = s[i + 1], m4 = s[i + 2], m5 = s[i + 3];              \
                m2 = (m2 + m3) * 4; m2 -= m1 + m4; m2 *= 5; m0 += m5 + m2 + 16; m0 >>= 5;                              \
                d[i] = (T)IMAX(0, IMIN(m0, pm));                                                                       \
            }                                                                                                          \
            for (intptr_t i = w - 4; i < w - 1; i++) d[i] = (T)((s[i] + s[i + 1] + 1) >> 1);                           \
            d[w - 1] = s[w - 1];                                                                                       \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
    }                                                                                                                  \
    /* MVFrame.cpp:1115-1150 Catmull-Rom */                                                                            \
    static void vbicubic_##SFX(uint8_t *d8, const uint8_t *s8, intptr_t pitch, intptr_t w, intptr_t h, int bits) {     \
        T *d = (T *)d8; const T *s = (const T *)s8; pitch /= sizeof(T);                                                \
        int pm = (1 << bits) - 1;                                                                                      \
        for (intptr_t j = 0; j < 1; j++) {                                                                             \
            for (intptr_t i = 0; i < w; i++) d[i] = (T)((s[i] + s[i + pitch] + 1) >> 1);                               \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t j = 1; j < h - 3; j++) {                                                                         \
            for (intptr_t i = 0; i < w; i++)                                                                           \
                d[i] = (T)IMIN(pm, IMAX(0, (-s[i - pitch] - s[i + pitch * 2] + (s[i] + s[i + pitch]) * 9 + 8) >> 4));  \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t j = h - 3; j < h - 1; j++) {                                                                     \
            for (intptr_t i = 0; i < w; i++) d[i] = (T)((s[i] + s[i + pitch] + 1) >> 1);                               \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
        for (intptr_t i = 0; i < w; i++) d[i] = s[i];                                                                  \
    }                                                                                                                  \
    /* MVFrame.cpp:1153-1176 */                                                                                        \
    static void hbicubic_##SFX(uint8_t *d8, const uint8_t *s8, intptr_t pitch, intptr_t w, intptr_t h, int bits) {     \
        T *d = (T *)d8; const T *s = (const T *)s8; pitch /= sizeof(T);                                                \
        int pm = (1 << bits) - 1;                                                                                      \
        for (intptr_t j = 0; j < h; j++) {                                                                             \
            d[0] = (T)((s[0] + s[1] + 1) >> 1);                                                                        \
            for (intptr_t i = 1; i < w - 3; i++)                                                                       \
                d[i] = (T)IMIN(pm, IMAX(0, (-(s[i - 1] + s[i + 2]) + (s[i] + s[i + 1]) * 9 + 8) >> 4));                \
            for (intptr_t i = w - 3; i < w - 1; i++) d[i] = (T)((s[i] + s[i + 1] + 1) >> 1);                           \
            d[w - 1] = s[w - 1];                                                                                       \
            d += pitch; s += pitch;                                                                                    \
        }                                                                                                              \
    }                                                                                                                  \
    /* MVFrame.cpp:1180-1197 */                                                                                        \
    static void avg2_##SFX(uint8_t *d8, const uint8_t *a8, const uint8_t *b8, intptr_t pitch, intptr_t w, intptr_t h) { \
        T *d = (T *)d8; const T *a = (const T *)a8; const T *b = (const T *)b8; pitch /= sizeof(T);                    \
        for (intptr_t j = 0; j < h; j++) {                                                                             \
            for (intptr_t i = 0; i < w; i++) d[i] = (T)((a[i] + b[i] + 1) >> 1);                                       \
            d += pitch; a += pitch; b += pitch;                                                                        \
        }                                                                                                              \
    }                                                                                                                  \
    /* MVFrame.cpp:575-594 RB2F_C */                                                                                   \
    static void rb2f_##SFX(uint8_t *d8, const uint8_t *s8, int dp, int sp, int w, int h) {                             \
        T *d = (T *)d8; const T *s = (const T *)s8; dp /= (int)sizeof(T); sp /= (int)sizeof(T);                        \
        for (int y = 0; y < h; y++) {                                                                                  \
            for (int x = 0; x < w; x++) d[x] = (T)((s[x * 2] + s[x * 2 + 1] + s[x * 2 + sp + 1] + s[x * 2 + sp] + 2) / 4); \
            d += dp; s += sp * 2;                                                                                      \
        }                                                                                                              \
    }                                                                                                                  \
    /* vertical passes of the separable reducers; kind 1 triangle :599-644, 2 bilinear :695-735,                        \
       3 quadratic :786-839, 4 cubic :903-955.  w here is the INTERMEDIATE width (2 * dst width). */                   \
    static void rb2_vert_##SFX(int kind, uint8_t *d8, const uint8_t *s8, int dp, int sp, int w, int h) {               \
        T *d = (T *)d8; const T *s = (const T *)s8; dp /= (int)sizeof(T); sp /= (int)sizeof(T);                        \
        if (kind == 1) {                                                                                               \
            for (int x = 0; x < w; x++) d[x] = (T)((s[x] + s[x + sp] + 1) / 2);                                        \
            d += dp; s += sp * 2;                                                                                      \
            for (int y = 1; y < h; y++) {                                                                              \
                for (int x = 0; x < w; x++) d[x] = (T)((s[x - sp] + s[x] * 2 + s[x + sp] + 2) / 4);                    \
                d += dp; s += sp * 2;                                                                                  \
            }                                                                                                          \
            return;                                                                                                    \
        }                                                                                                              \
        for (int y = 0; y < 1 && y < h; y++) {                                                                         \
            for (int x = 0; x < w; x++) d[x] = (T)((s[x] + s[x + sp] + 1) / 2);                                        \
            d += dp; s += sp * 2;                                                                                      \
        }                                                                                                              \
        for (int y = 1; y < h - 1; y++) {                                                                              \
            for (int x = 0; x < w; x++) {                                                                              \
                if (kind == 2)                                                                                         \
                    d[x] = (T)((s[x - sp] + (s[x] + s[x + sp]) * 3 + s[x + sp * 2] + 4) / 8);                          \
                else {                                                                                                 \
                    int m0 = s[x - sp * 2], m1 = s[x - sp], m2 = s[x], m3 = s[x + sp], m4 = s[x + sp * 2], m5 = s[x + sp * 3]; \
                    if (kind == 3) { m2 = (m2 + m3) * 22; m1 = (m1 + m4) * 9; m0 += m5 + m2 + m1 + 32; m0 >>= 6; }     \
                    else { m2 = (m2 + m3) * 10; m1 = (m1 + m4) * 5; m0 += m5 + m2 + m1 + 16; m0 >>= 5; }               \
                    d[x] = (T)m0;                                                                                      \
                }                                                                                                      \
            }                                                                                                          \
            d += dp; s += sp * 2;                                                                                      \
        }                                                                                                              \
        for (int y = IMAX(h - 1, 1); y < h; y++) {                                                                     \
            for (int x = 0; x < w; x++) d[x] = (T)((s[x] + s[x + sp] + 1) / 2);                                        \
            d += dp; s += sp * 2;                                                                                      \
        }                                                                                                              \
    }                                                                                                                  \
    /* in-place horizontal passes; kind 1 :649-680, 2 :740-771, 3 :844-888, 4 :961-1004 */                             \
    static void rb2_horz_##SFX(int kind, uint8_t *s8, int sp, int w, int h) {                                          \
        T *s = (T *)s8; sp /= (int)sizeof(T);                                                                          \
        for (int y = 0; y < h; y++) {                                                                                  \
            int s0 = (s[0] + s[1] + 1) / 2;                                                                            \
            if (kind == 1) {                                                                                           \
                for (int x = 1; x < w; x++) s[x] = (T)((s[x * 2 - 1] + s[x * 2] * 2 + s[x * 2 + 1] + 2) / 4);          \
                s[0] = (T)s0;                                                                                          \
            } else {                                                                                                   \
                for (int x = 1; x < w - 1; x++) {                                                                      \
                    if (kind == 2)                                                                                     \
                        s[x] = (T)((s[x * 2 - 1] + (s[x * 2] + s[x * 2 + 1]) * 3 + s[x * 2 + 2] + 4) / 8);             \
                    else {                                                                                             \
                        int m0 = s[x * 2 - 2], m1 = s[x * 2 - 1], m2 = s[x * 2], m3 = s[x * 2 + 1], m4 = s[x * 2 + 2], \
                            m5 = s[x * 2 + 3];                                                                         \
                        if (kind == 3) { m2 = (m2 + m3) * 22; m1 = (m1 + m4) * 9; m0 += m5 + m2 + m1 + 32; m0 >>= 6; } \
                        else { m2 = (m2 + m3) * 10; m1 = (m1 + m4) * 5; m0 += m5 + m2 + m1 + 16; m0 >>= 5; }           \
                        s[x] = (T)m0;                                                                                  \
                    }                                                                                                  \
                }                                                                                                      \
                s[0] = (T)s0;                                                                                          \
                for (int x = IMAX(w - 1, 1); x < w; x++) s[x] = (T)((s[x * 2] + s[x * 2 + 1] + 1) / 2);                \
            }                                                                                                          \
            s += sp;                                                                                                   \
        }                                                                                                              \
    }

DEFINE_KERNELS(uint8_t, u8)
DEFINE_KERNELS(uint16_t, u16)

typedef void (*refine_fn)(uint8_t *, const uint8_t *, intptr_t, intptr_t, intptr_t, int);

void mvo_refine_plane(int kind, int bits, uint8_t *dst, const uint8_t *src, intptr_t pitch, intptr_t w, intptr_t h) {
    static const refine_fn t8[7] = { hbilin_u8, vbilin_u8, dbilin_u8, hbicubic_u8, vbicubic_u8, hwiener_u8, vwiener_u8 };
    static const refine_fn t16[7] = { hbilin_u16, vbilin_u16, dbilin_u16, hbicubic_u16, vbicubic_u16, hwiener_u16, vwiener_u16 };
    (bits <= 8 ? t8 : t16)[kind](dst, src, pitch, w, h, bits);
}

void mvo_average2(int bits, uint8_t *dst, const uint8_t *a, const uint8_t *b, intptr_t pitch, intptr_t w, intptr_t h) {
    if (bits <= 8) avg2_u8(dst, a, b, pitch, w, h); else avg2_u16(dst, a, b, pitch, w, h);
}

/* MVFrame.cpp:1386-1527 mvpRefine */
static void plane_refine(mvo_plane *m, int sharp) {
    if (m->pel == 1) return;
    int k[3];
    if (sharp == 0) { k[0] = 0; k[1] = 1; k[2] = 2; }
    else if (sharp == 1) { k[0] = 3; k[1] = 4; k[2] = 3; }
    else { k[0] = 5; k[1] = 6; k[2] = 5; }
    const uint8_t *src[3]; uint8_t *dst[3];
    if (m->pel == 2) {
        dst[0] = m->p[1]; dst[1] = m->p[2]; dst[2] = m->p[3];
        src[0] = src[1] = m->p[0];
        src[2] = sharp == 0 ? m->p[0] : m->p[2];
    } else {
        dst[0] = m->p[2]; dst[1] = m->p[8]; dst[2] = m->p[10];
        src[0] = src[1] = m->p[0];
        src[2] = sharp == 0 ? m->p[0] : m->p[8];
    }
    for (int i = 0; i < 3; i++)
        mvo_refine_plane(k[i], m->bits, dst[i], src[i], m->pitch, m->pw, m->ph);
    if (m->pel == 4) { /* :1489-1524 */
        int b = m->bits, bps = m->bps; intptr_t P = m->pitch, W = m->pw, H = m->ph;
        mvo_average2(b, m->p[1], m->p[0], m->p[2], P, W, H);
        mvo_average2(b, m->p[9], m->p[8], m->p[10], P, W, H);
        mvo_average2(b, m->p[4], m->p[0], m->p[8], P, W, H);
        mvo_average2(b, m->p[6], m->p[2], m->p[10], P, W, H);
        mvo_average2(b, m->p[5], m->p[4], m->p[6], P, W, H);
        mvo_average2(b, m->p[3], m->p[0] + bps, m->p[2], P, W - 1, H);
        mvo_average2(b, m->p[11], m->p[8] + bps, m->p[10], P, W - 1, H);
        mvo_average2(b, m->p[12], m->p[0] + P, m->p[8], P, W, H - 1);
        mvo_average2(b, m->p[14], m->p[2] + P, m->p[10], P, W, H - 1);
        mvo_average2(b, m->p[13], m->p[12], m->p[14], P, W, H);
        mvo_average2(b, m->p[7], m->p[4] + bps, m->p[6], P, W - 1, H);
        mvo_average2(b, m->p[15], m->p[12] + bps, m->p[14], P, W - 1, H);
    }
}

/* MVFrame.cpp:1634-1683 mvpReduceTo */
static void plane_reduce(const mvo_plane *src, mvo_plane *dst, int rfilter) {
    uint8_t *d = dst->p[0] + dst->offpad;
    const uint8_t *s = src->p[0] + src->offpad;
    int u8 = src->bps == 1;
    if (rfilter == 0) {
        if (u8) rb2f_u8(d, s, dst->pitch, src->pitch, dst->w, dst->h); else rb2f_u16(d, s, dst->pitch, src->pitch, dst->w, dst->h);
        return;
    }
    if (u8) { rb2_vert_u8(rfilter, d, s, dst->pitch, src->pitch, dst->w * 2, dst->h); rb2_horz_u8(rfilter, d, dst->pitch, dst->w, dst->h); }
    else { rb2_vert_u16(rfilter, d, s, dst->pitch, src->pitch, dst->w * 2, dst->h); rb2_horz_u16(rfilter, d, dst->pitch, dst->w, dst->h); }
}

static void plane_pad(mvo_plane *m) { /* MVFrame.cpp:1374-1383 */
    if (m->bps == 1) pad_u8(m->p[0], m->pitch, m->hpad, m->vpad, m->w, m->h);
    else pad_u16(m->p[0], m->pitch, m->hpad, m->vpad, m->w, m->h);
}

/* ------------------------------------------------------------------ filter shell */

static int arg(int v, int dflt) { return v == MVO_UNSET ? dflt : v; }

/* MVSuper.c:140-264 mvsuperCreate */
int mvo_super_init(mvo_super *s, int width, int height, int bits, int subW, int subH, int gray,
                   int hpad, int vpad, int pel, int levels, int chroma, int sharp, int rfilter, char *err) {
    memset(s, 0, sizeof(*s));
    if (err) err[0] = 0;
    s->hpad = arg(hpad, 16);
    s->vpad = arg(vpad, 16);
    s->pel = arg(pel, 2);
    s->levels = arg(levels, 0);
    s->chroma = !!arg(chroma, 1);
    s->sharp = arg(sharp, 2);
    s->rfilter = arg(rfilter, 2);
    if (s->pel != 1 && s->pel != 2 && s->pel != 4) { snprintf(err, MVO_ERR, "Super: pel must be 1, 2, or 4."); return -1; }
    if (s->sharp < 0 || s->sharp > 2) { snprintf(err, MVO_ERR, "Super: sharp must be between 0 and 2 (inclusive)."); return -1; }
    if (s->rfilter < 0 || s->rfilter > 4) { snprintf(err, MVO_ERR, "Super: rfilter must be between 0 and 4 (inclusive)."); return -1; }
    if (bits > 16 || subW > 1 || subH > 1) {
        snprintf(err, MVO_ERR, "Super: input clip must be GRAY, 420, 422, 440, or 444, up to 16 bits, with constant dimensions.");
        return -1;
    }
    s->width = width; s->height = height; s->bits = bits; s->gray = gray;
    if (gray) s->chroma = 0;
    s->modeYUV = s->chroma ? MVO_YUVPLANES : MVO_YPLANE;
    s->xRatioUV = 1 << subW;
    s->yRatioUV = 1 << subH;
    int nLevelsMax = 0; /* :220-227 */
    while (mvo_plane_height_luma(height, nLevelsMax, s->yRatioUV, s->vpad) >= s->yRatioUV * 2 &&
           mvo_plane_width_luma(width, nLevelsMax, s->xRatioUV, s->hpad) >= s->xRatioUV * 2)
        nLevelsMax++;
    if (s->levels <= 0 || s->levels > nLevelsMax) s->levels = nLevelsMax;
    s->superWidth = width + 2 * s->hpad; /* :257-264 */
    s->superHeight = mvo_plane_super_offset(0, height, s->levels, s->pel, s->vpad, s->superWidth, s->yRatioUV) / s->superWidth;
    if (s->yRatioUV == 2 && (s->superHeight & 1)) s->superHeight++;
    if (s->xRatioUV == 2 && (s->superWidth & 1)) s->superWidth++;
    return 0;
}

/* MVSuper.c:229-256: how a pelclip of the given size is used.  0 = ignored (pel 1), 1 = plain, 2 = already padded, -1 = error */
int mvo_super_pelclip_mode(const mvo_super *s, int pelWidth, int pelHeight, char *err) {
    if (err) err[0] = 0;
    if (s->pel < 2) return 0;
    if (pelWidth == s->width * s->pel && pelHeight == s->height * s->pel) return 1;
    if (pelWidth == (s->width + s->hpad * 2) * s->pel && pelHeight == (s->height + s->vpad * 2) * s->pel) return 2;
    if (err) snprintf(err, MVO_ERR, "Super: pelclip's dimensions must be multiples of the input clip's dimensions.");
    return -1;
}

/* MVFrame.cpp:1529-1631 mvpRefineExt: sub-pel plane i takes every pel-th sample of the user's upsized clip at phase
 * (i / pel, i % pel).  A plain pelclip fills the interior and the planes are then edge-padded; a padded one is read
 * from ITS origin into the plane's origin over only w x h samples (the reference does not widen the loop to the padded
 * size), the rest of those planes stays as the frame memset left it. */
static void plane_refine_ext(mvo_plane *m, const uint8_t *pel8, int pelPitch, int padded) {
    const int n = m->pel * m->pel;
    for (int i = 1; i < n; i++) {
        uint8_t *d = m->p[i] + (padded ? 0 : m->offpad);
        const int ry = i / m->pel, rx = i % m->pel;
        for (int y = 0; y < m->h; y++)
            for (int x = 0; x < m->w; x++) {
                const uint8_t *sp = pel8 + (size_t)(y * m->pel + ry) * pelPitch + (size_t)(x * m->pel + rx) * m->bps;
                memcpy(d + (size_t)y * m->pitch + (size_t)x * m->bps, sp, m->bps);
            }
        if (!padded) {
            if (m->bps == 1) pad_u8(m->p[i], m->pitch, m->hpad, m->vpad, m->w, m->h);
            else pad_u16(m->p[i], m->pitch, m->hpad, m->vpad, m->w, m->h);
        }
    }
}

static void super_frame_impl(const mvo_super *s, const uint8_t *const src[3], const int srcPitch[3], const uint8_t *const pelclip[3],
                             const int pelPitch[3], int pelMode, uint8_t *const dst[3], const int dstPitch[3]);

/* MVSuper.c:43-126 mvsuperGetFrame (no pelclip) */
void mvo_super_frame(const mvo_super *s, const uint8_t *const src[3], const int srcPitch[3],
                     uint8_t *const dst[3], const int dstPitch[3]) {
    super_frame_impl(s, src, srcPitch, NULL, NULL, 0, dst, dstPitch);
}

/* MVSuper.c:43-126 with pelclip (:91-102); pelMode from mvo_super_pelclip_mode */
void mvo_super_frame_pelclip(const mvo_super *s, const uint8_t *const src[3], const int srcPitch[3], const uint8_t *const pelclip[3],
                             const int pelPitch[3], int pelMode, uint8_t *const dst[3], const int dstPitch[3]) {
    super_frame_impl(s, src, srcPitch, pelclip, pelPitch, pelMode, dst, dstPitch);
}

static void super_frame_impl(const mvo_super *s, const uint8_t *const src[3], const int srcPitch[3], const uint8_t *const pelclip[3],
                             const int pelPitch[3], int pelMode, uint8_t *const dst[3], const int dstPitch[3]) {
    int nplanes = s->gray ? 1 : 3;
    int bps = (s->bits + 7) / 8;
    for (int p = 0; p < nplanes; p++) { /* :75 */
        int ph = p ? s->superHeight / s->yRatioUV : s->superHeight;
        memset(dst[p], 0, (size_t)dstPitch[p] * ph);
    }
    mvo_gof g;
    mvo_gof_init(&g, s->levels, s->width, s->height, s->pel, s->hpad, s->vpad, s->modeYUV, s->xRatioUV, s->yRatioUV, s->bits);
    uint8_t *planes[3] = { dst[0], nplanes > 1 ? dst[1] : NULL, nplanes > 1 ? dst[2] : NULL };
    gof_update_yr(&g, planes, dstPitch, s->yRatioUV);
    for (int p = 0; p < nplanes; p++) { /* :85-86 mvfFillPlane -> MVFrame.cpp:1367-1371 */
        mvo_plane *m = &g.fr[0].pl[p];
        if (!m->p[0]) continue;
        for (int y = 0; y < m->h; y++)
            memcpy(m->p[0] + m->offpad + (size_t)y * m->pitch, src[p] + (size_t)y * srcPitch[p], (size_t)m->w * bps);
    }
    for (int i = 0; i < g.nlevels - 1; i++) { /* :88 mvgofReduce -> MVFrame.cpp:1928-1933 */
        for (int p = 0; p < 3; p++)
            if (g.fr[i].pl[p].p[0] && (s->modeYUV & (1 << p))) plane_reduce(&g.fr[i].pl[p], &g.fr[i + 1].pl[p], s->rfilter);
        for (int p = 0; p < 3; p++)
            if (g.fr[i + 1].pl[p].p[0]) plane_pad(&g.fr[i + 1].pl[p]);
    }
    for (int p = 0; p < 3; p++) /* :89 mvgofPad */
        if (g.fr[0].pl[p].p[0] && (s->modeYUV & (1 << p))) plane_pad(&g.fr[0].pl[p]);
    for (int p = 0; p < 3; p++) { /* :91-103 mvpRefineExt / mvgofRefine */
        if (!(g.fr[0].pl[p].p[0] && (s->modeYUV & (1 << p)))) continue;
        if (pelMode > 0) plane_refine_ext(&g.fr[0].pl[p], pelclip[p], pelPitch[p], pelMode == 2);
        else plane_refine(&g.fr[0].pl[p], s->sharp);
    }
}

uint32_t mvo_fnv1a(const uint8_t *p, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
    return h;
}

/* ---- mv.Finest: MVFinest.c:48-140 + Merge4PlanesToBig / Merge16PlanesToBig (MaskFun.cpp:206-330): the pel^2 sub-pel planes of
 * level 0 interleaved into one (padded width * pel) x (padded height * pel) plane; pel 1 copies the padded level-0 plane (the
 * reference copies luma-sized rectangles for every plane there, overrunning the chroma planes: restated per plane).
 * Planes the super clip does not carry (chroma=0) are left untouched.  Parity unpinned. */
void mvo_finest_size(const mvo_super *s, int *w, int *h) { *w = (s->width + 2 * s->hpad) * s->pel; *h = (s->height + 2 * s->vpad) * s->pel; }
void mvo_finest_frame(const mvo_super *s, const uint8_t *const sup[3], const int supPitch[3], uint8_t *const dst[3], const int dstPitch[3]) {
    mvo_gof g;
    mvo_gof_init(&g, s->levels, s->width, s->height, s->pel, s->hpad, s->vpad, s->modeYUV, s->xRatioUV, s->yRatioUV, s->bits);
    mvo_gof_update(&g, (uint8_t *const *)sup, supPitch, s->yRatioUV);
    const int bps = (s->bits + 7) / 8, pel = s->pel, np = s->gray ? 1 : 3;
    for (int p = 0; p < np; p++) {
        if (!(s->modeYUV & (1 << p))) continue;
        const mvo_plane *m = &g.fr[0].pl[p];
        for (int y = 0; y < m->ph * pel; y++)
            for (int x = 0; x < m->pw * pel; x++) {
                const uint8_t *q = m->p[(x % pel) | ((y % pel) * pel)] + (size_t)(y / pel) * m->pitch + (size_t)(x / pel) * bps;
                memcpy(dst[p] + (size_t)y * dstPitch[p] + (size_t)x * bps, q, bps);
            }
    }
}

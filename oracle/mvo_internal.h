/* mvo_internal.h -- pyramid view shared by the oracle's translation units (test infrastructure only). */
#ifndef MVO_INTERNAL_H
#define MVO_INTERNAL_H

#include "mvoracle.h"

#define MVO_YPLANE 1
#define MVO_UPLANE 2
#define MVO_VPLANE 4
#define MVO_YUVPLANES 7

/* one plane of one level: MVFrame.h MVPlane, MVFrame.cpp:1327-1364 */
typedef struct mvo_plane {
    uint8_t *p[16]; /* pel*pel sub-pel planes, p[i] = base + i*pitch*padded_h */
    int pitch, w, h, hpad, vpad, pel, bits, bps, pw, ph, offpad;
} mvo_plane;

typedef struct mvo_frame {
    mvo_plane pl[3];
    int mode;
} mvo_frame;

typedef struct mvo_gof {
    int nlevels;
    mvo_frame fr[MVO_MAX_LEVELS];
} mvo_gof;

void mvo_gof_init(mvo_gof *g, int levels, int w, int h, int pel, int hpad, int vpad, int mode, int xr, int yr, int bits);
void mvo_gof_update(mvo_gof *g, uint8_t *const planes[3], const int pitch[3], int yRatioUV);

/* MVFrame.cpp:1732-1734 mvpGetPointer: x,y in pel units relative to the interior origin */
static inline const uint8_t *mvo_plane_pointer(const mvo_plane *m, int nX, int nY) {
    nX += m->hpad * m->pel;
    nY += m->vpad * m->pel;
    if (m->pel == 1)
        return m->p[0] + nX * m->bps + nY * m->pitch;
    if (m->pel == 2) {
        int idx = (nX & 1) | ((nY & 1) << 1);
        return m->p[idx] + (nX >> 1) * m->bps + (nY >> 1) * m->pitch;
    }
    {
        int idx = (nX & 3) | ((nY & 3) << 2);
        return m->p[idx] + (nX >> 2) * m->bps + (nY >> 2) * m->pitch;
    }
}

/* absolute (padding-inclusive) coordinates in pel units: MVFrame.cpp:1707-1729 */
static inline const uint8_t *mvo_plane_abs_pointer(const mvo_plane *m, int logpel, int nX, int nY) {
    int mask = (1 << logpel) - 1;
    int idx = (nX & mask) | ((nY & mask) << logpel);
    return m->p[idx] + (nX >> logpel) * m->bps + (nY >> logpel) * m->pitch;
}

static inline int mvo_ilog2(int i) { int r = 0; while (i > 1) { i /= 2; r++; } return r; }

#endif

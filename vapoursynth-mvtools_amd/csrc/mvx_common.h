// mvx_common.h -- internal declarations shared by the host side of libmvtools_amd (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mvtools_amd.h"

#define MVX_MAX_LEVELS 24

void mvx_set_error(const char *fmt, ...);
int mvx_debug_value(const char *name, int def); // value of a developer option set through mvx_debug_option (mvx_analyse.hip), or def
void mvx_divided_data(const mvx_analysis_data *in, mvx_analysis_data *out);

#define HIP_CHECK(expr)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            mvx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MVX_E_DEVICE;                                                                  \
        }                                                                                         \
    } while (0)

// ---- thread / stream safety of the *_frames entry points.  A handle owns device scratch (job tables, plans, masks) that every
// call overwrites.  VapourSynth calls a filter's getFrame concurrently (fmParallel), possibly with different streams, so each
// call (a) holds the handle's mutex while it enqueues, and (b) makes its stream wait for the event the previous call on the same
// handle recorded after its last kernel -- so the scratch is never rewritten (or freed: hipFree synchronises the device) while
// an earlier call's kernels still read it.  Calls on one handle therefore run back to back on the GPU; calls on different
// handles overlap.  Declare one CallGuard per scratch owner and open a Scope at the top of the entry point.
struct CallGuard {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    ~CallGuard() { if (ev) (void)hipEventDestroy(ev); }
    struct Scope {
        CallGuard &g; hipStream_t st;
        Scope(CallGuard &g_, hipStream_t s) : g(g_), st(s) { g.mu.lock(); if (g.ev) (void)hipStreamWaitEvent(st, g.ev, 0); }
        ~Scope() {
            if (!g.ev) (void)hipEventCreateWithFlags(&g.ev, hipEventDisableTiming);
            if (g.ev) (void)hipEventRecord(g.ev, st);
            g.mu.unlock();
        }
    };
};

// ---- pyramid geometry (MVFrame.cpp:1209-1247) -- host only
int mvx_plane_height_luma(int src_height, int level, int yRatioUV, int vpad);
int mvx_plane_width_luma(int src_width, int level, int xRatioUV, int hpad);
unsigned mvx_plane_super_offset(int chroma, int src_height, int level, int pel, int vpad, int plane_pitch, int yRatioUV);

// Geometry of one level of one plane inside the super frame.
struct LevelPlane {
    int w, h;          // interior samples
    int hpad, vpad;    // samples
    int pw, ph;        // padded
    long long off;     // byte offset of sub-pel plane 0 of this level inside the super plane (pitch dependent)
};

struct mvx_super {
    mvx_super_info info;
    // per level / plane dims (pitch independent part)
    int lw[MVX_MAX_LEVELS][3], lh[MVX_MAX_LEVELS][3];
};

// fills LevelPlane for (level, plane) given the super plane pitch in bytes
void mvx_level_plane(const mvx_super_info &si, int level, int plane, long long pitch, LevelPlane *out);

struct mvx_analyse;
struct mvx_degrain;
struct mvx_compensate;

// overlap windows (Overlap.cpp:40-125), host generated
void mvx_over_windows(int16_t *win9, int nx, int ny, int ox, int oy);

static inline int mvx_ilog2(int i) { int r = 0; while (i > 1) { i /= 2; r++; } return r; }

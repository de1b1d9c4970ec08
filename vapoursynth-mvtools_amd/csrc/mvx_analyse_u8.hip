// 8-bit search kernels specialised for the common 4:2:0 block geometries (Geo<BW, BH, XR, YR>)
#include "mvx_analyse_kernel.h"
#include "mvx_analyse_fast.h"
// the lean kernel of the default search (mvx_analyse_fast.h): L.fast = chains per SIMD it is launched at, L.cpw chains per workgroup
int mvx_analyse_launch_fast_u8(const AParams &P, const ALaunch &L) {
    if (P.blkX == 8) {
        if (L.fast == 4) return launch_analyse_fast<1, 8, 4, 16>(L);
        if (L.fast == 3) return launch_analyse_fast<1, 8, 3, 12>(L);
        if (L.fast == 2) return launch_analyse_fast<1, 8, 2, 8>(L);
        if (L.fast == 1) return launch_analyse_fast<1, 8, 1, 4>(L);
    }
    if (P.blkX == 16) {
        if (L.fast == 4) return launch_analyse_fast<1, 16, 4, 16>(L);
        if (L.fast == 3) return launch_analyse_fast<1, 16, 3, 12>(L);
        if (L.fast == 2) return launch_analyse_fast<1, 16, 2, 8>(L);
        if (L.fast == 1) return launch_analyse_fast<1, 16, 1, 4>(L);
    }
    return 1;
}

int mvx_analyse_launch_u8(const AParams &P, const ALaunch &L) {
    if (P.xr != 2 || P.yr != 2) return 1;
    // More chains than SIMDs: the 8-bit kernels have a 256-register build so that two chains share a SIMD (+53 % at 1080p,
    // DESIGN.md 4.2).  It drops the LDS floor that spreads a small launch one chain per SIMD.
    if (L.cpw == 4) { // four chains per workgroup (mvx_analyse_frames sorted the job table by reference frame)
        if (P.blkX == 8 && P.blkY == 8 && L.wpe == 3) return launch_analyse_kernel<1, Geo<8, 8, 2, 2>, 3, 4>(L); // three chains per SIMD (launches with more than two chains per SIMD)
        if (P.blkX == 8 && P.blkY == 8) return L.wpe == 2 ? launch_analyse_kernel<1, Geo<8, 8, 2, 2>, 2, 4>(L) : launch_analyse_kernel<1, Geo<8, 8, 2, 2>, 1, 4>(L);
        if (P.blkX == 16 && P.blkY == 16) return L.wpe == 2 ? launch_analyse_kernel<1, Geo<16, 16, 2, 2>, 2, 4>(L) : launch_analyse_kernel<1, Geo<16, 16, 2, 2>, 1, 4>(L);
    }
    if (L.njobs > L.simds) {
        ALaunch L2 = L;
        L2.ldsBytes = L.ldsNeed;
        if (P.blkX == 8 && P.blkY == 8) return launch_analyse_kernel<1, Geo<8, 8, 2, 2>, 2>(L2); // (the 16x16 kernel would spill at 256 registers)
    }
    if (P.blkX == 8 && P.blkY == 8) return launch_analyse_kernel<1, Geo<8, 8, 2, 2>>(L);
    if (P.blkX == 16 && P.blkY == 16) return launch_analyse_kernel<1, Geo<16, 16, 2, 2>>(L);
    return 1;
}

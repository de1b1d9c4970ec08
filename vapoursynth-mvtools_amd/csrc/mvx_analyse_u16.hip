// 16-bit search kernels specialised for the common 4:2:0 block geometries (Geo<BW, BH, XR, YR, scan step>)
#define MVX_PROF_EXPORT 1
#include "mvx_analyse_kernel.h"
int mvx_analyse_launch_u16(const AParams &P, const ALaunch &L) {
    if (P.xr != 2 || P.yr != 2) return 1;
    // The LDS search-window kernels (Geo<..., scan step>) are bit-exact but measured SLOWER than the plain ones in round 1
    // (DESIGN.md 4.2): opt-in via MVX_WINDOW=1 until the window path is cheaper in instructions.
    const int S = L.mode == 1 ? P.blkX - P.ovX : 0;
    // refinement-tile kernel (opt-in, MVX_TILE=1): bit-exact, as fast as the plain kernel at one chain per SIMD.  A 256-register
    // build of it for two chains per SIMD was measured too (r1): every chain then takes 2.5x as long (2016 chains: 175 fps
    // against 214) -- see DESIGN.md 4.2.
    if (L.mode == 2 && P.blkX == 16 && P.blkY == 16) return launch_analyse_kernel<2, Geo<16, 16, 2, 2, 0, false, true>>(L);
    if (L.mode == 0 && L.cpw == 4) { // four chains per workgroup (mvx_analyse_frames sorted the job table by reference frame)
        if (P.blkX == 16 && P.blkY == 16) return launch_analyse_kernel<2, Geo<16, 16, 2, 2>, 1, 4>(L);
        if (P.blkX == 32 && P.blkY == 32) return launch_analyse_kernel<2, Geo<32, 32, 2, 2>, 1, 4>(L);
        if (P.blkX == 8 && P.blkY == 8) return launch_analyse_kernel<2, Geo<8, 8, 2, 2>, 1, 4>(L);
    }
    if (P.blkX == 16 && P.blkY == 16) return S == 8 ? launch_analyse_kernel<2, Geo<16, 16, 2, 2, 8>>(L) : launch_analyse_kernel<2, Geo<16, 16, 2, 2>>(L);
    if (P.blkX == 32 && P.blkY == 32) return launch_analyse_kernel<2, Geo<32, 32, 2, 2>>(L);
    if (P.blkX == 8 && P.blkY == 8) return S == 4 ? launch_analyse_kernel<2, Geo<8, 8, 2, 2, 4>>(L) : launch_analyse_kernel<2, Geo<8, 8, 2, 2>>(L);
    return 1;
}

// 16-bit search kernels specialised for the common 4:2:0 block geometries (Geo<BW, BH, XR, YR, SATD>)
#define MVX_PROF_EXPORT 1
#include "mvx_analyse_kernel.h"
#include "mvx_analyse_fast.h"
// the lean kernel of the default search (mvx_analyse_fast.h): L.fast = chains per SIMD it is launched at, L.cpw chains per workgroup
int mvx_analyse_launch_fast_u16(const AParams &P, const ALaunch &L) {
    if (P.blkX == 16) {
        if (L.fast == 4) return launch_analyse_fast<2, 16, 4, 16>(L);
        if (L.fast == 3) return launch_analyse_fast<2, 16, 3, 12>(L);
        if (L.fast == 2) return launch_analyse_fast<2, 16, 2, 8>(L);
        if (L.fast == 1) return launch_analyse_fast<2, 16, 1, 4>(L);
    }
    if (P.blkX == 8) {
        if (L.fast == 4) return launch_analyse_fast<2, 8, 4, 16>(L);
        if (L.fast == 2) return launch_analyse_fast<2, 8, 2, 8>(L);
        if (L.fast == 1) return launch_analyse_fast<2, 8, 1, 4>(L);
    }
    if (P.blkX == 32) {
        if (L.fast == 3) return launch_analyse_fast<2, 32, 3, 12>(L);
        if (L.fast == 2) return launch_analyse_fast<2, 32, 2, 8>(L);
        if (L.fast == 1) return launch_analyse_fast<2, 32, 1, 4>(L);
    }
    return 1;
}

int mvx_analyse_launch_u16(const AParams &P, const ALaunch &L) {
    if (P.xr != 2 || P.yr != 2) return 1;
    if (L.cpw >= 4) { // several chains per workgroup (mvx_analyse_frames sorted the job table by reference frame)
        if (L.wpe == 3 && L.cpw == 12) { // three per SIMD, twelve per CU
            if (P.blkX == 16 && P.blkY == 16) return launch_analyse_kernel<2, Geo<16, 16, 2, 2>, 3, 12>(L);
        }
        if (L.wpe == 2 && L.cpw == 8) { // two per SIMD, eight per CU
            if (P.blkX == 16 && P.blkY == 16) return launch_analyse_kernel<2, Geo<16, 16, 2, 2>, 2, 8>(L);
            if (P.blkX == 8 && P.blkY == 8) return launch_analyse_kernel<2, Geo<8, 8, 2, 2>, 2, 8>(L);
            if (P.blkX == 32 && P.blkY == 32) return launch_analyse_kernel<2, Geo<32, 32, 2, 2>, 2, 8>(L); // (fits 256 registers because it fetches the source block late, Searcher::PF_LATE)
        }
        if (P.blkX == 16 && P.blkY == 16) return launch_analyse_kernel<2, Geo<16, 16, 2, 2>, 1, 4>(L);
        if (P.blkX == 32 && P.blkY == 32) return launch_analyse_kernel<2, Geo<32, 32, 2, 2>, 1, 4>(L);
        if (P.blkX == 8 && P.blkY == 8) return launch_analyse_kernel<2, Geo<8, 8, 2, 2>, 1, 4>(L);
    }
    // one chain per workgroup (MVX_CPW=1)
    if (P.blkX == 16 && P.blkY == 16) return launch_analyse_kernel<2, Geo<16, 16, 2, 2>>(L);
    if (P.blkX == 32 && P.blkY == 32) return launch_analyse_kernel<2, Geo<32, 32, 2, 2>>(L);
    if (P.blkX == 8 && P.blkY == 8) return launch_analyse_kernel<2, Geo<8, 8, 2, 2>>(L);
    return 1;
}

// generic search kernels (block geometry only known at run time)
#include "mvx_analyse_kernel.h"
int mvx_analyse_launch_any(const AParams &P, const ALaunch &L) {
    if (L.cpw == 4) { // four chains per workgroup (mvx_analyse_frames sorted the job table by reference frame)
        if (P.dctmode != 0) return P.bps == 1 ? launch_analyse_kernel<1, GeoAnyDct, 1, 4>(L) : launch_analyse_kernel<2, GeoAnyDct, 1, 4>(L); // SATD cost modes
        return P.bps == 1 ? launch_analyse_kernel<1, GeoAny, 1, 4>(L) : launch_analyse_kernel<2, GeoAny, 1, 4>(L);
    }
    if (P.dctmode != 0) return P.bps == 1 ? launch_analyse_kernel<1, GeoAnyDct>(L) : launch_analyse_kernel<2, GeoAnyDct>(L); // SATD cost modes
    return P.bps == 1 ? launch_analyse_kernel<1, GeoAny>(L) : launch_analyse_kernel<2, GeoAny>(L);
}

int mvx_recalc_launch(const AParams &P, const RLaunch &L) {
    if (P.bps == 1) hipLaunchKernelGGL((recalc_kernel<1>), dim3(L.nBlk, L.njobs), dim3(64), L.ldsBytes, L.st, L.dP, L.dR, L.dJobs, L.ldsRow, L.ldsHist, L.histBins);
    else hipLaunchKernelGGL((recalc_kernel<2>), dim3(L.nBlk, L.njobs), dim3(64), L.ldsBytes, L.st, L.dP, L.dR, L.dJobs, L.ldsRow, L.ldsHist, L.histBins);
    return MVX_OK;
}

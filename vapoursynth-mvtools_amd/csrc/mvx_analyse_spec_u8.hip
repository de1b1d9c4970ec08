// 8-bit builds of the speculative default search (mvx_analyse_spec.h): S.L.fast = chains per SIMD it is launched at
#include "mvx_analyse_kernel.h"
#include "mvx_analyse_spec.h"
int mvx_analyse_launch_spec_u8(const AParams &P, const ASpecLaunch &S) {
    const int k = S.L.fast;
    if (S.side && P.blkX == 16) { // 16x16 blocks side by side: the SIDE builds (256 registers: two chains per SIMD or fewer; teams)
        if (S.team) return launch_analyse_spec_team<1, 16, 2, 8, true>(S);
        return launch_analyse_spec<1, 16, 2, 8, true>(S);
    }
    if (S.team) { // the team form: 256-register builds, up to eight waves per chain
        if (P.blkX == 8) return launch_analyse_spec_team<1, 8, 2, 8>(S);
        if (P.blkX == 16) return launch_analyse_spec_team<1, 16, 2, 8>(S);
        return 1;
    }
    if (P.blkX == 8) {
        if (k == 4) return launch_analyse_spec<1, 8, 4, 16>(S);
        if (k == 3) return launch_analyse_spec<1, 8, 3, 12>(S);
        if (k == 2) return launch_analyse_spec<1, 8, 2, 8>(S);
        if (k == 1) return launch_analyse_spec<1, 8, 1, 4>(S);
    }
    if (P.blkX == 16) {
        if (k == 4) return launch_analyse_spec<1, 16, 4, 16>(S);
        if (k == 3) return launch_analyse_spec<1, 16, 3, 12>(S);
        if (k == 2) return launch_analyse_spec<1, 16, 2, 8>(S);
        if (k == 1) return launch_analyse_spec<1, 16, 1, 4>(S);
    }
    return 1;
}

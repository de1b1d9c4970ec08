// the LDS-window kernel of the default search (mvx_analyse_win.h): 16-bit clips, 16x16 blocks; L.fast = chains per SIMD, L.cpw chains per workgroup
#include "mvx_analyse_kernel.h"
#include "mvx_analyse_win.h"
int mvx_analyse_launch_win(const AParams &P, const ALaunch &L) {
    if (L.fast == 3) return launch_analyse_win<3, 12>(L);
    if (L.fast == 2) return launch_analyse_win<2, 8>(L);
    if (L.fast == 1) return launch_analyse_win<1, 4>(L);
    return 1;
}

#ifdef MVX_WIN_PROF
extern "C" __attribute__((visibility("default"))) int mvx_debug_winprof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_winprof), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

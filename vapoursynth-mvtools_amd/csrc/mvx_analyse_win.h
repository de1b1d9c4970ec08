// mvx_analyse_win.h -- the default search of mv.Analyse for 16-bit clips with 16x16 blocks (BASELINE cfg3), reference samples
// served from LDS windows that LDS-DMA fills once per block.
//
// Why (profiles/r2_search_kernel_sq_tcp_counters.txt, r3): analyse_fast_kernel reads every candidate straight from global memory.
// A 16x16 block of 16-bit samples is 32 bytes per row, so each wave-level load touches ~32 different 128-byte lines and the CU's
// vector L1 looks up one line per clock: 694 line look-ups per block and chain, twelve chains per CU = 8 300 of the 12 000 cycles a
// block takes -- and the block's three search rounds (predictor set, hexagon + square, square) each wait for such a batch of loads.
// Here a block issues ONE batch of LDS-DMA loads (global_load_lds_dwordx4: no registers, 1 KiB per instruction) when it starts:
//   * the source block (as before, now by DMA),
//   * slot 0 / slot 1: the reference blocks of the zero vector and of the global predictor (exact positions, no margin),
//   * window W2: all pel^2 sub-pel planes around the block's predictor (PlaneOfBlocks.cpp:1100 / :449) -- 24 x 21 luma samples and
//     12 x 10 UV pairs per plane, i.e. vectors within about +-8 (x) / -4..+6 (y) half-pel units of the predictor,
// ~300 line look-ups instead of 694, one memory round trip instead of three.  Every candidate of the three rounds whose block lies
// inside a slot / the window is then evaluated from LDS (aligned dwords + v_alignbit, since a sample position is 2-byte aligned);
// any other candidate takes FastSearcher's global-memory path -- same samples, same arithmetic, so results are identical by
// construction.  The bad-block rescue (rare) always runs on the global path.
// Second change: the SOURCE block lives in registers (each lane of an 8-lane candidate group owns the same six 16-byte pieces in
// every round), so LDS bandwidth is spent on reference samples only; the hexagon + square round evaluates two candidates per group
// and the exhaustive radius-2 round three (it used 4 and 2 lanes per candidate).
// Third: the lambda scaling (PlaneOfBlocks.cpp:456-462, fp64) of 64 blocks is computed lane-parallel when their predictors are
// fetched (one lane per block) instead of per block in uniform code.
#pragma once
#include "mvx_analyse_fast.h"

struct WG16 { // LDS layout of one chain, bytes
    enum { RP = 48,                                   // row pitch of every reference region (3 DMA pieces of 16 bytes)
           SRC = 0, SRC_UV = 512,                     // source block: luma 16 rows x 32 B, UV 8 rows x 32 B (FastSearcher's layout)
           S0L = 768, S0C = S0L + 16 * RP,            // slot 0 (zero vector): luma 16 rows, UV 8 rows
           S1L = S0C + 8 * RP, S1C = S1L + 16 * RP,   // slot 1 (global predictor)
           W2L = S1C + 8 * RP,                        // window: luma, pel^2 planes of PHL bytes (21 rows x 48 B + 16)
           PHL = 1024, WW = 24, WH = 21, MX = 4, MY = 2,
           W2C = W2L + 4 * PHL,                       // window: UV, planes 2j / 2j + 1 in one KiB: p -> (p >> 1) * 1024 + (p & 1) * PHC
           PHC = 480, CWW = 12, CWH = 10, CMX = 2, CMY = 1,
           TOTAL = W2C + 2048,
           HIST = S0L };                              // the global-motion histogram (between levels only) lies over the windows
};
static_assert(WG16::W2L % 16 == 0 && WG16::TOTAL == 9216, "LDS layout");

struct WinSearcher : FastSearcher<2, 16, true> {
    typedef FastSearcher<2, 16, true> F;
    typedef WG16 W;
    __device__ WinSearcher(const AParams &p, const AJob &j) : F(p, j) {}

    // per-lane level constants: byte offsets of the DMA patterns (lane t <-> piece t of a region with 3 (2) pieces per row) and the
    // candidate offsets of the lane's group in every pass, packed (dy << 16) | (dx & 0xffff)
    unsigned o3Y, o2Y, o2C, o3C;
    int tHex, tSq, tR2a, tR2b;
    // window state of the current block (uniform)
    int wLoX, wHiX, wLoY, wHiY;   // vectors whose luma AND chroma blocks lie inside W2
    int wbx, wby, wcbx, wcby;     // (block origin - window origin) << logPel, luma / chroma
    int s0L, s0C, s1L, s1C;       // LDS byte offsets (chain-relative) of the slot candidates' first sample
    v4u srcR[6];                  // this lane's six pieces of the source block (4 luma, 2 UV)
#ifdef MVX_WIN_PROF
    long long prof[8];
#define WPROF(i, t0) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long t1_ = (long long)__builtin_amdgcn_s_memtime(); prof[i] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define WPROF(i, t0) ((void)0)
#endif

    // LDS-DMA: lanes 0 .. LANES-1 each fetch 16 bytes from base + voff and land them at ldsAddr + 16 * lane.  The block loop is
    // uniform control flow with all 64 lanes active, so EXEC is narrowed and restored inside the statement (no branch, no mask
    // registers); one wait state between the M0 write and its use.
    template <int LANES> __device__ __forceinline__ static void dma(gl_u8 *base, unsigned voff, unsigned ldsAddr) {
        const unsigned long long b = (unsigned long long)uni((long long)(unsigned long long)base); // (uniform by construction; readfirstlane makes the compiler see it)
        if (LANES == 64)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(uni((int)ldsAddr)), "v"(voff), "s"(b) : "memory", "m0");
        else
            asm volatile("s_lshr_b64 exec, -1, %3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                         : : "s"(uni((int)ldsAddr)), "v"(voff), "s"(b), "n"(64 - LANES) : "memory", "m0");
    }
    __device__ __forceinline__ static void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }

    __device__ __forceinline__ void win_setup_level() {
        const int l = lane_id(), g = l >> 3;
        const unsigned pY = pitchY, pUV = 2 * pitchC;
        o3Y = (unsigned)(l / 3) * pY + (unsigned)(l % 3) * 16;
        o2Y = (unsigned)(l >> 1) * pY + (unsigned)(l & 1) * 16;
        o2C = (unsigned)(l >> 1) * pUV + (unsigned)(l & 1) * 16;
        o3C = (unsigned)(l / 30) * (2 * pstrideC) + (unsigned)((l % 30) / 3) * pUV + (unsigned)(l % 3) * 16;
        auto pk = [](int dx, int dy) { return (int)(((unsigned)dy << 16) | ((unsigned)dx & 0xffffu)); };
        tHex = pk(tab8(HEX2X >> 8, g), tab8(HEX2Y >> 8, g));                                                   // hex2[g + 1], :682-687
        tSq = pk(tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), g), tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), g));        // pobExpandingSearch(1, 1), :636-658
        tR2a = pk(tab8(PACK8(-1, -1, 0, 0, 1, 1, -2, 2), g), tab8(PACK8(-2, 2, -2, 2, -2, 2, -1, -1), g));     // ring 2, candidates 0..7
        tR2b = pk(tab8(PACK8(-2, 2, -2, 2, -2, -2, 2, 2), g), tab8(PACK8(0, 0, 1, 1, -2, 2, -2, 2), g));       // ring 2, candidates 8..15
    }

    // ---- one batch of DMA loads per block: source block, the two slots, the window around (cvx, cvy) = the block's predictor
    __device__ __forceinline__ void win_issue(int cvx, int cvy) {
        const unsigned L = (unsigned)(unsigned long long)lds;
        const unsigned pY = pitchY, pUV = 2 * pitchC;
        const int m = pel - 1, npp = pel * pel;
        const int limX = (int)(pY >> 1) - W::WW, limCX = (int)(pUV >> 2) - W::CWW; // last origin whose 48-byte rows stay inside the pitch
        const int cx0 = x0 >> 1, cy0 = y0 >> 1;
        // source block
        dma<32>(srcY + ((unsigned)y0 * pY + (unsigned)x0 * 2), o2Y, L + W::SRC);
        if (chroma) dma<16>(srcUV + ((unsigned)cy0 * pUV + (unsigned)cx0 * 4), o2C, L + W::SRC_UV);
        { // slot 0: the zero vector (0, fieldShift); its chroma block ignores fieldShift (PlaneOfBlocks.cpp:836-839)
            const int ay = (y0 << logPel) + fieldShift;
            const int ox = min(x0, limX);
            const unsigned pp = (unsigned)((ay & m) << logPel);
            dma<48>(refY + (pp * pstrideY + (unsigned)(ay >> logPel) * pY + (unsigned)ox * 2), o3Y, L + W::S0L);
            s0L = W::S0L + (x0 - ox) * 2;
            if (chroma) {
                const int ocx = min(cx0, limCX);
                dma<24>(refUV + ((unsigned)cy0 * pUV + (unsigned)ocx * 4), o3C, L + W::S0C);
                s0C = W::S0C + (cx0 - ocx) * 4;
            }
        }
        { // slot 1: the global predictor (already clipped for this block, :859)
            const int ax = (x0 << logPel) + gmvx, ay = (y0 << logPel) + gmvy;
            const int fx = ax >> logPel, fy = ay >> logPel;
            const int ox = min(fx & ~1, limX);
            const unsigned pp = (unsigned)((ax & m) | ((ay & m) << logPel));
            dma<48>(refY + (pp * pstrideY + (unsigned)fy * pY + (unsigned)ox * 2), o3Y, L + W::S1L);
            s1L = W::S1L + (fx - ox) * 2;
            if (chroma) {
                const int cvx_ = (gmvx + (gmvx < 0 ? 1 : 0)) >> 1, cvy_ = (gmvy + (gmvy < 0 ? 1 : 0)) >> 1;
                const int cax = (cx0 << logPel) + cvx_, cay = (cy0 << logPel) + cvy_;
                const int fcx = cax >> logPel, fcy = cay >> logPel;
                const int ocx = min(fcx, limCX);
                const unsigned ppc = (unsigned)((cax & m) | ((cay & m) << logPel));
                dma<24>(refUV + (ppc * (2 * pstrideC) + (unsigned)fcy * pUV + (unsigned)ocx * 4), o3C, L + W::S1C);
                s1C = W::S1C + (fcx - ocx) * 4;
            }
        }
        { // W2: every sub-pel plane around the centre vector
            const int ax = (x0 << logPel) + cvx, ay = (y0 << logPel) + cvy;
            const int wX0 = min(max(((ax >> logPel) - W::MX) & ~1, 0), limX);
            const int wY0 = min(max((ay >> logPel) - W::MY, 0), ph - W::WH);
            const unsigned ob = (unsigned)wY0 * pY + (unsigned)wX0 * 2;
            dma<63>(refY + ob, o3Y, L + W::W2L);
            if (npp > 1) {
                dma<63>(refY + (ob + pstrideY), o3Y, L + W::W2L + W::PHL);
                dma<63>(refY + (ob + 2 * pstrideY), o3Y, L + W::W2L + 2 * W::PHL);
                dma<63>(refY + (ob + 3 * pstrideY), o3Y, L + W::W2L + 3 * W::PHL);
            }
            wbx = (x0 - wX0) << logPel; wby = (y0 - wY0) << logPel;
            // vectors inside: wX0 <= (x0 * pel + vx) >> logPel <= wX0 + WW - 16, same for y
            int loX = -wbx, hiX = ((wX0 + W::WW - 16 - x0) << logPel) + m;
            int loY = -wby, hiY = ((wY0 + W::WH - 16 - y0) << logPel) + m;
            if (chroma) {
                const int ccx = (cvx + (cvx < 0 ? 1 : 0)) >> 1, ccy = (cvy + (cvy < 0 ? 1 : 0)) >> 1;
                const int cax = (cx0 << logPel) + ccx, cay = (cy0 << logPel) + ccy;
                const int wCX0 = min(max((cax >> logPel) - W::CMX, 0), limCX);
                const int wCY0 = min(max((cay >> logPel) - W::CMY, 0), (ph >> 1) - W::CWH);
                const unsigned oc = (unsigned)wCY0 * pUV + (unsigned)wCX0 * 4;
                if (npp > 1) {
                    dma<60>(refUV + oc, o3C, L + W::W2C);
                    dma<60>(refUV + (oc + 4 * pstrideC), o3C, L + W::W2C + 1024);
                } else
                    dma<30>(refUV + oc, o3C, L + W::W2C);
                wcbx = (cx0 - wCX0) << logPel; wcby = (cy0 - wCY0) << logPel;
                // chroma vector c = trunc(v / 2) must satisfy a <= c <= b:  v >= (a > 0 ? 2a : 2a - 1),  v <= (b >= 0 ? 2b + 1 : 2b)
                const int ax_ = -wcbx, bx_ = ((wCX0 + W::CWW - 8 - cx0) << logPel) + m;
                const int ay_ = -wcby, by_ = ((wCY0 + W::CWH - 8 - cy0) << logPel) + m;
                loX = max(loX, ax_ > 0 ? 2 * ax_ : 2 * ax_ - 1); hiX = min(hiX, bx_ >= 0 ? 2 * bx_ + 1 : 2 * bx_);
                loY = max(loY, ay_ > 0 ? 2 * ay_ : 2 * ay_ - 1); hiY = min(hiY, by_ >= 0 ? 2 * by_ + 1 : 2 * by_);
            }
            wLoX = loX; wHiX = hiX; wLoY = loY; wHiY = hiY;
        }
    }
    // the source pieces of this lane: s = lane & 7 owns row s (pieces 0, 1 = its two halves) and row s + 8 (pieces 2, 3) of the luma
    // block and row s of the UV block (pieces 4, 5).  With one row per lane the eight lanes of a group read eight different LDS banks
    // (rows are 12 dwords apart: 12 s mod 32 is a permutation of 0, 4 .. 28).
    __device__ __forceinline__ void win_load_src() {
        const int s = lane_id() & 7;
        const lds_u8 *p = lds + W::SRC + s * 32;
        srcR[0] = *(const LDS_AS v4u *)p; srcR[1] = *(const LDS_AS v4u *)(p + 16);
        srcR[2] = *(const LDS_AS v4u *)(p + 256); srcR[3] = *(const LDS_AS v4u *)(p + 272);
        srcR[4] = *(const LDS_AS v4u *)(p + W::SRC_UV); srcR[5] = *(const LDS_AS v4u *)(p + W::SRC_UV + 16);
    }

    __device__ __forceinline__ bool in_w2(int vx, int vy) const { return vx >= wLoX && vx <= wHiX && vy >= wLoY && vy <= wHiY; }
    // LDS offsets of candidate (vx, vy) inside W2 (meaningful when in_w2(vx, vy)); vyc as in eval()
    __device__ __forceinline__ void w2_addr(int vx, int vy, int vyc, int &aL, int &aC) const {
        const int m = pel - 1;
        const int ax = vx + wbx, ay = vy + wby;
        const int pp = (ax & m) | ((ay & m) << logPel);
        aL = W::W2L + pp * W::PHL + (ay >> logPel) * W::RP + (ax >> logPel) * 2;
        const int cvx = (vx + (int)((unsigned)vx >> 31)) >> 1, cvy = (vyc + (int)((unsigned)vyc >> 31)) >> 1;
        const int cax = cvx + wcbx, cay = cvy + wcby;
        const int ppc = (cax & m) | ((cay & m) << logPel);
        aC = W::W2C + (ppc >> 1) * 1024 + (ppc & 1) * W::PHC + (cay >> logPel) * W::RP + (cax >> logPel) * 4;
    }

    typedef unsigned w2a4 __attribute__((ext_vector_type(2), aligned(4)));
    // partial SADs of ONE candidate from LDS.  A luma row of the candidate starts at a 2-byte aligned address: nine aligned dwords
    // + v_alignbit give its sixteen samples; UV rows are dword aligned.
    __device__ __forceinline__ void sad_win(int s, int aL, int aC, unsigned &accL, unsigned &accC) const {
        const unsigned a = (unsigned)(aL + s * W::RP);
        const lds_u8 *p = lds + (a & ~3u);
        const unsigned sh = (a & 2u) * 8;
#pragma unroll
        for (int h = 0; h < 2; h++) { // rows s and s + 8
            const lds_u8 *q = p + h * 8 * W::RP;
            const w2a4 d01 = *(const LDS_AS w2a4 *)q, d23 = *(const LDS_AS w2a4 *)(q + 8), d45 = *(const LDS_AS w2a4 *)(q + 16), d67 = *(const LDS_AS w2a4 *)(q + 24);
            const unsigned d8 = *(const LDS_AS unsigned *)(q + 32);
            accL = sad32<2>(srcR[2 * h][0], __builtin_amdgcn_alignbit(d01[1], d01[0], sh), accL);
            accL = sad32<2>(srcR[2 * h][1], __builtin_amdgcn_alignbit(d23[0], d01[1], sh), accL);
            accL = sad32<2>(srcR[2 * h][2], __builtin_amdgcn_alignbit(d23[1], d23[0], sh), accL);
            accL = sad32<2>(srcR[2 * h][3], __builtin_amdgcn_alignbit(d45[0], d23[1], sh), accL);
            accL = sad32<2>(srcR[2 * h + 1][0], __builtin_amdgcn_alignbit(d45[1], d45[0], sh), accL);
            accL = sad32<2>(srcR[2 * h + 1][1], __builtin_amdgcn_alignbit(d67[0], d45[1], sh), accL);
            accL = sad32<2>(srcR[2 * h + 1][2], __builtin_amdgcn_alignbit(d67[1], d67[0], sh), accL);
            accL = sad32<2>(srcR[2 * h + 1][3], __builtin_amdgcn_alignbit(d8, d67[1], sh), accL);
        }
        if (chroma) {
            const lds_u8 *q = lds + (aC + s * W::RP);
            const w2a4 d01 = *(const LDS_AS w2a4 *)q, d23 = *(const LDS_AS w2a4 *)(q + 8), d45 = *(const LDS_AS w2a4 *)(q + 16), d67 = *(const LDS_AS w2a4 *)(q + 24);
            accC = sad32<2>(srcR[4][0], d01[0], accC); accC = sad32<2>(srcR[4][1], d01[1], accC);
            accC = sad32<2>(srcR[4][2], d23[0], accC); accC = sad32<2>(srcR[4][3], d23[1], accC);
            accC = sad32<2>(srcR[5][0], d45[0], accC); accC = sad32<2>(srcR[5][1], d45[1], accC);
            accC = sad32<2>(srcR[5][2], d67[0], accC); accC = sad32<2>(srcR[5][3], d67[1], accC);
        }
    }

    // pobPseudoEPZSearch (:819-968) for the default parameters as a loop of PASSES.  A pass evaluates one candidate per 8-lane group
    // (from a slot / the window, else through FastSearcher::eval from global memory), forms the costs as pobCheckMV does and accepts
    // the first minimum (ordered arg-min = the reference's sequential strict `<` update).  Passes: the predictor set (:832-915), then
    // at a Hex2 level the hexagon (:682-687) and the square around the point it ends on (:636-658; i_me_range <= 3: no half-hexagon
    // iterations), at an Exhaustive level the rings 1 and 2 around the best predictor (:786-791; ring 2 takes two passes).  The code of
    // a pass exists once; what differs between passes is chosen by scalar branches.
    enum { K_PRED, K_HEX, K_SQ, K_R2A, K_R2B };
    __device__ __forceinline__ void search_block_win() {
        const int lane = lane_id();
        int g = lane >> 3;
        const int s = lane & 7;
        asm("" : "+v"(g)); // keeps the (g == k) masks out of scalar registers across the block loop
        const bool exh = searchType != SearchHex2;
        int kind = K_PRED, cx = 0, cy = 0;
        nMinCost = 0x7fffffff;
#pragma unroll 1
        for (;;) {
            int vx, vy, vyc, oL, oC;
            bool ok, inw;
            if (kind == K_PRED) { // zero, global, predictor, median, left, up, ahead
                vx = 0; vy = fieldShift;
                vx = g == 1 ? gmvx : vx; vy = g == 1 ? gmvy : vy;
                vx = g == 2 ? predX : vx; vy = g == 2 ? predY : vy;
                vx = g == 3 ? pX[0] : vx; vy = g == 3 ? pY[0] : vy;
                vx = g == 4 ? pX[1] : vx; vy = g == 4 ? pY[1] : vy;
                vx = g == 5 ? pX[2] : vx; vy = g == 5 ? pY[2] : vy;
                vx = g == 6 ? pX[3] : vx; vy = g == 6 ? pY[3] : vy;
                vyc = g == 0 ? 0 : vy; // the zero candidate's chroma ignores fieldShift (:836-839)
                ok = g < 7;            // (all of them are clipped vectors)
                w2_addr(vx, vy, vyc, oL, oC);
                inw = in_w2(vx, vy) || g < 2;
                oL = g == 0 ? s0L : oL; oC = g == 0 ? s0C : oC;
                oL = g == 1 ? s1L : oL; oC = g == 1 ? s1C : oC;
            } else {
                const int t = kind == K_HEX ? tHex : kind == K_SQ ? tSq : kind == K_R2A ? tR2a : tR2b;
                vx = cx + (int)(short)t; vy = cy + (t >> 16); vyc = vy;
                ok = vector_ok(vx, vy) && (kind != K_HEX || g < 6);
                w2_addr(vx, vy, vyc, oL, oC);
                inw = in_w2(vx, vy);
            }
            unsigned aL = 0, aC = 0;
            if (ok && inw) sad_win(s, oL, oC, aL, aC);
            if (ok && !inw) F::template eval<3>(s, vx, vy, vyc, aL, aC);
            aL = group_sum_c<3>(aL); aC = group_sum_c<3>(aC);
            const int tot = (int)aL + (chroma ? (int)aC : 0);
            int cc;
            if (kind == K_PRED) { // pobCheckMV0: no new-vector penalty (:846, :870, :894)
                const int pen = g == 0 ? penaltyZero : (g == 1 ? pglobal : 0);
                cc = tot + (int)(((long long)pen * tot) >> 8);
                cc = sat_add(g >= 3 ? motion_distortion(vx, vy) : 0, cc);
            } else
                cc = cost_new(vx, vy, aL, aC);
            const int w = accept<3>((ok && cc < nMinCost) ? cc : 0x7fffffff, tot);
            if (kind == K_HEX) { // pobCheckMVdir: a winner moves the centre; the square follows around wherever the centre is now
                if (w >= 0) { cx = bcast_i(vx, w); cy = bcast_i(vy, w); bestX = cx; bestY = cy; }
                kind = K_SQ;
                continue;
            }
            if (w >= 0) { bestX = bcast_i(vx, w); bestY = bcast_i(vy, w); }
            if (kind == K_PRED) {
                cx = bestX; cy = bestY;
                kind = (exh || nSearchParam <= 1) ? K_SQ : K_HEX; // (ring 1 of the exhaustive search = the square)
                continue;
            }
            if (!exh || kind == K_R2B) break;
            kind = kind == K_SQ ? K_R2A : K_R2B;
        }
        // ---- bad vector: wide search (:938-963), on the global path
        if (__builtin_expect(blkIdx > 1 && (long long)bestSad > badSAD + badSAD * badcount / 16, 0)) rescue();
    }

    // a VECTOR of the blob read past the L1 (the previous row's results were stored by this wave a row ago)
    __device__ __forceinline__ static v4u ld_batch_l2(GL_AS const GVec *p) {
        GL_AS const unsigned *q = (GL_AS const unsigned *)p;
        v4u r;
        r[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r[2] = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r[3] = 0;
        return r;
    }

    // GroupOfPlanes.c:69-125 + PlaneOfBlocks.cpp:971-1131 for one level (FastSearcher::search_level with the windows)
    __device__ __forceinline__ void search_level_win(int lvl, int globalX, int globalY, GL_AS const GVec *coarse, int coarseBlkX, int coarseBlkY, int coarseLogPel, int syncEvery) {
        const int l = lane_id();
        const ALevel &L = P.lv[lvl];
        nBlkX = uni(L.nBlkX); nBlkY = uni(L.nBlkY); pel = uni(L.pel); logPel = uni(L.logPel);
        chroma = uni(P.chroma);
        pw = uni(L.pw); ph = uni(L.ph); hpad = uni(L.hpad); vpad = uni(L.vpad);
        auto uptr = [](const unsigned char *p) { return (gl_u8 *)(unsigned long long)uni((long long)(unsigned long long)p); };
        srcY = uptr(J.src[0] + L.off[0]); refY = uptr(J.ref[0] + L.off[0]);
        pitchY = (unsigned)uni((int)P.pitch[0]); pitchC = (unsigned)uni((int)P.pitch[1]); pstrideY = (unsigned)uni((int)L.pstride[0]); pstrideC = (unsigned)uni((int)L.pstride[1]);
        shadowY = (unsigned)uni((int)P.shadow[0]);
        srcUV = uptr(J.src[1] + P.shadow[1] + 2 * L.off[1]); refUV = uptr((J.ref[1] ? J.ref[1] : J.src[1]) + P.shadow[1] + 2 * L.off[1]);
        srcU = srcV = refU = refV = srcUV; // (chroma is read from the UV plane only)
        unsigned char *rec = (unsigned char *)(unsigned long long)uni((long long)(unsigned long long)(J.blob + L.blobOff));
        vectors = (GL_AS GVec *)(rec + 4);
        if (l == 0) *(int *)rec = 4 + nBlkX * nBlkY * 16; // pobWriteHeaderToArray :413-416
        const int nBlk = nBlkX * nBlkY;
        const bool smallestPlane = lvl == P.nLevels - 1;
        interpolate(coarse, coarseBlkX, coarseBlkY, coarseLogPel);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_s_barrier();

        // ---- plane scan set-up (doPobSearchMVs :979-1034)
        if (smallestPlane) { searchType = P.nLevels == 1 ? P.searchType : P.searchTypeCoarse; nSearchParam = P.nLevels == 1 ? P.nPelSearch : P.nSearchParam; }
        else { searchType = lvl == 0 ? P.searchType : P.searchTypeCoarse; nSearchParam = lvl == 0 ? P.nPelSearch : P.nSearchParam; }
        searchType = uni(searchType); nSearchParam = uni(nSearchParam);
        fieldShift = uni(lvl == 0 ? J.fieldShift : 0);
        badSAD = uni(P.badSAD); badrange = uni(P.badrange); badcount = 0;
        gmvx = pel * globalX; gmvy = pel * globalY + fieldShift;
        int nLambdaLevel = P.lambda / (pel * pel);
        const int nScale = 1 << lvl;
        if (P.plevel == 1) nLambdaLevel = nLambdaLevel * nScale;
        else if (P.plevel == 2) nLambdaLevel = nLambdaLevel * nScale * nScale;
        nLambdaLevel = uni(nLambdaLevel);
        penaltyZero = uni(P.pzero); pglobal = uni(P.global ? P.pglobal : P.pzero); penaltyNew = uni(P.pnew); LSAD = uni((long long)P.lsad);
        const int stepX = uni(P.blkX - P.ovX), stepY = uni(P.blkY - P.ovY);
        const int hps = hpad >> lvl, vps = vpad >> lvl; // :1091-1092
        const bool meander = uni(P.meander) != 0;
        win_setup_level();

        // lambda of a block whose predictor SAD is predSad (:456-462), fp64 as the reference
        auto lambda_of = [&](int predSad) {
            const double scale = (double)LSAD / (double)(LSAD + (long long)(predSad >> 1));
            return (int)(long long)((double)(long long)nLambdaLevel * scale * scale);
        };
        v4u bSelf = {0, 0, 0, 0}, bBelow = {0, 0, 0, 0}, bUp = {0, 0, 0, 0}, bOut = {0, 0, 0, 0};
        int prevX = 0, prevY = 0, prevSad = 0;
        int curIb = 0, curBy = 0;
#ifdef MVX_WIN_PROF
        long long tp = (long long)__builtin_amdgcn_s_memtime();
#endif
        for (int n = 0; n < nBlk; n++) {
            if (syncEvery && (curIb & (syncEvery - 1)) == 0) __builtin_amdgcn_s_barrier();
            WPROF(0, tp);
            const int blky = curBy;
            const bool fwd = (blky & 1) == 0 || !meander;
            const int blkx = fwd ? curIb : nBlkX - 1 - curIb;
            const int dir = fwd ? 1 : -1;
            const bool rowStart = curIb == 0;
            if (++curIb == nBlkX) { curIb = 0; curBy++; }
            blkIdx = blky * nBlkX + blkx;
            x0 = hpad + stepX * blkx; y0 = vpad + stepY * blky;
            const int col = blkx & 63;
            if (rowStart || col == (fwd ? 0 : 63)) { // a new group of 64 columns: predictors of this row / the row below, results of the row above
                const int c0 = blkx & ~63, c = c0 + l;
                const bool in = c < nBlkX;
                if (in) {
                    bSelf = ld_batch(&vectors[blky * nBlkX + c]);
                    if (!smallestPlane) bSelf[3] = (unsigned)lambda_of((int)bSelf[2]); // one lane per block
                }
                const int cb = c + dir;
                bBelow = v4u{0, 0, 0, 0};
                if (in && blky < nBlkY - 1 && cb >= 0 && cb < nBlkX) bBelow = ld_batch(&vectors[(blky + 1) * nBlkX + cb]);
                if (in && blky > 0) bUp = ld_batch_l2(&vectors[(blky - 1) * nBlkX + c]);
            }
            // ---- motion-vector limits (:1094-1097)
            nDxMax = (pw - x0 - 16 - hpad + hps) << logPel;
            nDyMax = (ph - y0 - 16 - vpad + vps) << logPel;
            nDxMin = -((x0 - hpad + hps) << logPel);
            nDyMin = -((y0 - vpad + vps) << logPel);
            gmvx = clipx(gmvx); gmvy = clipy(gmvy); // cumulative clip (:859)
            // ---- predictors (:419-463, :1100)
            int sfx, sfy, sfs, blx, bly, bls, upx, upy, ups;
            batch_get(bSelf, col, sfx, sfy, sfs);
            batch_get(bBelow, col, blx, bly, bls);
            batch_get(bUp, col, upx, upy, ups);
            const bool aheadCol = fwd ? blkx < nBlkX - 1 : blkx > 0;
            const bool useBelow = blky < nBlkY - 1 && aheadCol;
            const bool useUpAhead = !useBelow && blky > 0 && aheadCol; // last block row only (:441-447)
            int ahx = blx, ahy = bly, ahs = bls;
            if (useUpAhead) {
                const v4u t = ld_batch_l2(&vectors[(blky - 1) * nBlkX + blkx + dir]);
                ahx = uni((int)t[0]); ahy = uni((int)t[1]); ahs = uni((int)t[2]);
            }
            const bool haveAhead = useBelow || useUpAhead;
            const bool havePrev = fwd ? blkx > 0 : blkx < nBlkX - 1;
            pX[1] = clipx(havePrev ? prevX : 0); pY[1] = clipy(havePrev ? prevY : fieldShift); const int s1 = havePrev ? prevSad : 0;
            pX[2] = clipx(blky > 0 ? upx : 0); pY[2] = clipy(blky > 0 ? upy : fieldShift); const int s2 = blky > 0 ? ups : 0;
            pX[3] = clipx(haveAhead ? ahx : 0); pY[3] = clipy(haveAhead ? ahy : fieldShift); const int s3 = haveAhead ? ahs : 0;
            int s0;
            if (blky > 0) {
                auto med = [](int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); };
                pX[0] = med(pX[1], pX[2], pX[3]); pY[0] = med(pY[1], pY[2], pY[3]);
                s0 = max(s1, max(s2, s3));
            } else { pX[0] = pX[1]; pY[0] = pY[1]; s0 = s1; }
            if (smallestPlane) { predX = pX[0]; predY = pY[0]; }
            else { predX = clipx(sfx); predY = clipy(sfy); }
            WPROF(1, tp);
            // ---- the block's memory traffic: one batch of DMA loads
            win_issue(predX, predY);
            WPROF(2, tp);
            // :456-462: lambda shrinks with the predictor's SAD (row 0 searches without the motion term, :1081-1084)
            nLambda = 0;
            if (blky > 0) nLambda = smallestPlane ? uni(lambda_of(s0)) : __builtin_amdgcn_readlane((int)bSelf[3], col);
            dma_wait();
            WPROF(3, tp);
            win_load_src();
            __builtin_amdgcn_wave_barrier();
            search_block_win();
            __builtin_amdgcn_wave_barrier();
            WPROF(4, tp);
            // ---- result (:967, :1106): collected per group, stored when the group (or the row) ends
            { const bool mine = l == col; bOut[0] = mine ? (unsigned)bestX : bOut[0]; bOut[1] = mine ? (unsigned)bestY : bOut[1]; bOut[2] = mine ? (unsigned)bestSad : bOut[2]; }
            prevX = bestX; prevY = bestY; prevSad = bestSad;
            const bool rowEnd = curIb == 0;
            if (rowEnd || col == (fwd ? 63 : 0)) {
                const int c = (blkx & ~63) + l;
                if (c < nBlkX) {
                    typedef unsigned a4v __attribute__((ext_vector_type(4), aligned(4)));
                    const a4v t = {bOut[0], bOut[1], bOut[2], 0u};
                    *(GL_AS a4v *)&vectors[blky * nBlkX + c] = t;
                }
            }
            WPROF(5, tp);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_s_barrier();
    }
};

#ifdef MVX_WIN_PROF
__device__ unsigned long long g_winprof[8];
#endif

template <int WPE, int MAXCPW>
__global__ __launch_bounds__(64 * MAXCPW, WPE) void analyse_win_kernel(const AParams *Pp, const AJob *jobs, int njobs, int ldsChain, int syncEvery, int histBins, int flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AParams &P = *Pp;
    const int cpw = (int)(blockDim.x >> 6);
    int wg = (int)blockIdx.x;
    if (flags & MVX_FAST_XCD_REMAP) {
        const int n = (int)gridDim.x, x = wg & 7, slot = wg >> 3;
        wg = x * (n >> 3) + min(x, n & 7) + slot;
    }
    const int chain = uni(wg * cpw + (int)(threadIdx.x >> 6));
    if (chain >= njobs) return;
    const AJob &J = jobs[chain];
    if (!J.blob) return;
    const int l = lane_id();
    int *hdr = (int *)J.blob;
    if (!J.valid) { // gopWriteDefaultToArray GroupOfPlanes.c:150-164, pobWriteDefaultToArray PlaneOfBlocks.cpp:1529-1556
        if (l == 0) { hdr[0] = P.blobSize; hdr[1] = 0; }
        for (int lvl = P.nLevels - 1; lvl >= 0; lvl--) {
            const ALevel &L = P.lv[lvl];
            unsigned char *rec = J.blob + L.blobOff;
            const int nBlk = L.nBlkX * L.nBlkY;
            if (l == 0) *(int *)rec = 4 + nBlk * 16;
            GVec *v = (GVec *)(rec + 4);
            for (int i = l; i < nBlk; i += WAVE) { GVec d; d.x = 0; d.y = 0; d.sad = P.verybigSAD; v[i] = d; }
        }
        return;
    }
    if (l == 0) { hdr[0] = P.blobSize; hdr[1] = 1; } // GroupOfPlanes.c:77-85
    WinSearcher S(P, J);
    S.lds = (lds_u8 *)smem + uni((int)(threadIdx.x >> 6)) * ldsChain;
    S.ldsRow = 0; S.ldsHist = WG16::HIST; S.histBins = histBins;
#ifdef MVX_WIN_PROF
    for (int i = 0; i < 8; i++) S.prof[i] = 0;
#endif
    int gx = 0, gy = 0; // zeroMV, MVAnalysisData.h:79
    GL_AS const GVec *coarse = nullptr;
    int cbx = 0, cby = 0, clp = 0;
    for (int lvl = P.nLevels - 1; lvl >= 0; lvl--) {
        if (coarse && P.global) S.estimate_global(coarse, cbx * cby, 8192 * P.lv[lvl + 1].pel, &gx, &gy);
        S.search_level_win(lvl, gx, gy, coarse, cbx, cby, clp, cpw > 1 ? syncEvery : 0);
        coarse = S.vectors; cbx = P.lv[lvl].nBlkX; cby = P.lv[lvl].nBlkY; clp = P.lv[lvl].logPel;
    }
#ifdef MVX_WIN_PROF
    if (l == 0 && chain == 5) for (int i = 0; i < 8; i++) g_winprof[i] = (unsigned long long)S.prof[i];
#endif
}

// can the window kernel run this parameter set?  (host)  16-bit 16x16 blocks, UV-interleaved chroma plane present (or luma only)
static inline bool mvx_win_eligible(const AParams &P) {
    if (!mvx_fast_eligible(P) || P.bps != 2 || P.blkX != 16) return false;
    if (P.chroma && !P.shadow[1]) return false; // chroma is read from the UV plane
    for (int i = 0; i < P.nLevels; i++) {
        if (P.lv[i].pel > 2) return false;
        // the windows must fit the planes of every level
        if (P.lv[i].pw < WG16::WW + 2 || P.lv[i].ph < WG16::WH || (P.chroma && ((P.lv[i].pw >> 1) < WG16::CWW || (P.lv[i].ph >> 1) < WG16::CWH))) return false;
        if ((P.lv[i].hpad & 1) || (P.lv[i].vpad & 1)) return false;
        if (4LL * P.lv[i].pstride[0] + 64LL * P.pitch[0] >= 0xffffffffLL || 8LL * P.lv[i].pstride[1] + 64LL * P.pitch[1] >= 0xffffffffLL) return false; // 32-bit DMA offsets
    }
    return P.pitch[0] >= 2 * WG16::WW && P.pitch[1] >= 2 * WG16::CWW && (P.blkX - P.ovX) % 2 == 0 && (P.blkY - P.ovY) % 2 == 0;
}
static constexpr int MVX_WIN_HIST_BINS = (WG16::TOTAL - WG16::HIST) / 4;

template <int WPE, int MAXCPW> static int launch_analyse_win(const ALaunch &L) {
    const int perChain = WG16::TOTAL;
    const int cpw = L.cpw < MAXCPW ? L.cpw : MAXCPW;
    const int lds = perChain * cpw;
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)analyse_win_kernel<WPE, MAXCPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((analyse_win_kernel<WPE, MAXCPW>), dim3((L.njobs + cpw - 1) / cpw), dim3(64 * cpw), lds, L.st, L.dP, L.dJobs, L.njobs, perChain, L.syncEvery,
                       MVX_WIN_HIST_BINS, L.flags);
    return MVX_OK;
}

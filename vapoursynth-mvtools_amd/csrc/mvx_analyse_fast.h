// mvx_analyse_fast.h -- the DEFAULT search of mv.Analyse as a lean kernel.
//
// analyse_kernel (mvx_analyse_kernel.h) carries every search pattern, tryMany, the SATD cost modes and run-time block
// geometry in one state machine; that generality costs the block loop of the default search ~240 registers, ~950 spilled
// scalars and about half of its ~1000 instructions per block (round-1 VERDICT).  This kernel implements ONLY what the
// default parameters of mv.Analyse execute (PlaneOfBlocks.cpp:819-968 with searchType Hex2 at the finest level,
// i_me_range <= 3 -- :667-724 -- and Exhaustive radius 2 above it -- :786-791 --, no tryMany, SAD cost, 4:2:0 or luma only)
// for the square block sizes 8 / 16 / 32, plus the bad-block rescue (:938-963: UMH / expanding rings around zero, then the
// small-radius refinement) as a plain loop.  mvx_analyse_frames picks it whenever a parameter set qualifies
// (fast_eligible); everything else still runs analyse_kernel.  Results are identical by construction and by test: the
// parity suite runs the same cases through both (MVX_LAB builds can force either).
//
// What is different from analyse_kernel's fast path:
//   * one chain per wavefront as before, but the per-block data that lives in memory is handled 64 blocks at a time: the
//     hierarchical predictors of the current row and of the row below, and the previous row's results, are fetched as ONE
//     vector load each per 64 blocks (lane i holds block 64*g + i) and read per block with v_readlane; the results of 64
//     blocks are collected in four registers (v_writelane) and stored with one coalesced store.  Per block that removes
//     three loads, two stores and ~25 address / exec-mask / readfirstlane instructions;
//   * all per-block control state is scalar and small (no candidate generators, no pattern program counter), costs are
//     32-bit (block SADs are < 2^27, the motion term saturates);
//   * the rescue is a loop over passes of eight candidates (eight lanes each) around one evaluation site.
#pragma once
// Tuning constants of the candidate evaluation.  Their alternatives (and the forms this file had before: group sums through the
// update_dpp builtin, LDS reads right before use, six loads in flight, a compare / select chain for the predictor candidates, no
// de-duplication, the hexagon pass without its speculative square) were measured one switch at a time: DESIGN.md 4.2.3,
// profiles/r3_lean_kernel_trims_ab.txt, r3_lean_kernel_dedup_nospec_ab.txt.
#define MVX_SRC_AHEAD 2 // the source block's LDS pieces are read this many pieces ahead of their use (a read right before its use costs the wave an LDS round trip per piece)
#define MVX_STREAM_MAX 12 // a candidate's pieces per lane up to which luma + chroma are ONE stream of loads (the serial kernel: longer streams measured slower).  The default of FastSearcher's STREAM_MAX parameter; the speculative kernel instantiates its base class with 48
#ifndef MVX_INFLIGHT
#define MVX_INFLIGHT 12 // reference loads a lane keeps in flight while it evaluates a candidate: all twelve of a hexagon-pass candidate
#endif
#include "mvx_analyse_kernel.h"

// -DMVX_FAST_PROF (tools/build_variant.py; read back by tools/fastprof.py): cycles of ONE chain per phase of the block loop, stamped with
// s_memtime.  A stamp waits for the scalar counter only, not for the vector loads in flight, so the phases overlap as they do in the
// normal build; what a phase shows is where the wave's time goes, including time it waits while the SIMD's other waves issue.
#ifdef MVX_FAST_PROF
#define FPROF_N 12
static __device__ unsigned long long g_fastprof[FPROF_N];
#define FPROF(i, t0) do { const long long t1_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(t1_) : "memory"); prof[i] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define FPROF(i, t0) ((void)0)
#endif

// sums of a and b over aligned groups of 1 << LOGG lanes (1 <= LOGG <= 4), every lane ends up with its group's totals.  v_add_u32_dpp
// reads the register it wrote one step earlier, which needs two wait states: the other chain's step and one s_nop provide them, so a
// step costs 1.5 instructions per value instead of the 4 (v_mov, s_nop, v_mov_dpp, v_add) the update_dpp builtin compiles to.
template <int LOGG> __device__ __forceinline__ void group_sum2(unsigned &a, unsigned &b) {
    static_assert(LOGG >= 1 && LOGG <= 4, "group_sum2: groups of 2, 4, 8 or 16 lanes");
    asm volatile("s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "+v"(a), "+v"(b));
    if (LOGG >= 2)
        asm volatile("s_nop 0\n\t"
                     "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                     : "+v"(a), "+v"(b));
    if (LOGG >= 3)
        asm volatile("s_nop 0\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_u32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf"
                     : "+v"(a), "+v"(b));
    if (LOGG >= 4) // (both halves of a row are uniform by now: lane i meets lane 15 - i of the other half)
        asm volatile("s_nop 0\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_u32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf"
                     : "+v"(a), "+v"(b));
    // (the sums are consumed by plain VALU instructions; the compiler's hazard recogniser treats an asm statement's outputs as just
    // written and adds what a DPP or lane read of them would need)
}

template <int BPS, int BW> struct FGeo {
    static constexpr int BH = BW;
    static constexpr int LROWB = BW * BPS, LCB = LROWB < 16 ? LROWB : 16, LC = LROWB / LCB, LT = BH * LC;
    static constexpr int CW = BW / 2, CH = BH / 2, CROWB = CW * BPS, CCB = CROWB < 16 ? CROWB : 16, CC = CROWB / CCB, CT = CH * CC;
    static constexpr int UOFF = BH * LROWB, VOFF = UOFF + CH * CROWB, SRCB = VOFF + CH * CROWB;
    // chroma from the UV-interleaved shadow plane (one row = the U and V samples of a block row, alternating): same bytes, half the rows
    static constexpr int UVROWB = 2 * CROWB, UVCB = UVROWB < 16 ? UVROWB : 16, UVC = UVROWB / UVCB, UVT = CH * UVC;
    static constexpr int NPF = (LT + (2 * CT > UVT ? 2 * CT : UVT) + WAVE - 1) / WAVE; // 16-byte (or smaller) pieces of the source block per lane
    static constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v / 2); }
    static constexpr int LLOGC = ilog2c(LC), CLOGC = ilog2c(CC), UVLOGC = ilog2c(UVC);
};

// can the lean kernel run this parameter set?  (host; mirrors the level-wise choice of doPobSearchMVs, PlaneOfBlocks.cpp:979-1034)
static inline bool mvx_fast_eligible(const AParams &P) {
    if (P.dctmode != 0 || P.tryMany || P.xr != 2 || P.yr != 2 || P.blkX != P.blkY) return false;
    if (!(P.blkX == 8 || P.blkX == 16 || (P.blkX == 32 && P.bps == 2))) return false;
    auto ok = [](int type, int param) { return (type == SearchHex2 && param >= 1 && param <= 3) || (type == SearchExhaustive && param == 2); };
    const int st = P.searchType, cst = P.searchTypeCoarse;
    if (st == SearchHorizontal || st == SearchVertical) return false;
    if (!ok(st, P.nPelSearch)) return false;                  // finest level (and the only level when nLevels == 1)
    if (P.nLevels > 1 && !ok(cst, P.nSearchParam)) return false; // every coarser level
    for (int i = 0; i < P.nLevels; i++) {
        const ALevel &L = P.lv[i];
        { // the level's lambda (doPobSearchMVs :1003-1009) must be a non-negative 32-bit value: the kernel multiplies it as one
            long long v = P.lambda / (L.pel * L.pel);
            const long long sc = 1LL << i;
            if (P.plevel == 1) v *= sc; else if (P.plevel == 2) v *= sc * sc;
            if (v < 0 || v > 0x7fffffffLL) return false;
        }
        if ((long long)L.pel * L.pel * L.pstride[0] >= 0xffffffffLL || (long long)L.pel * L.pel * L.pstride[1] >= 0xffffffffLL) return false; // 32-bit plane offsets
        if ((long long)L.pel * L.pel * L.pstride[0] + (P.bps == 1 ? 3 : 1) * P.shadow[0] >= 0xffffffffLL || 2 * (long long)L.pel * L.pel * L.pstride[1] + P.shadow[1] >= 0xffffffffLL) return false;
        if ((L.pw << L.logPel) >= 30000 || (L.ph << L.logPel) >= 30000) return false; // vectors and their squared distances stay well inside int
    }
    return true;
}

// UV: chroma is read from the UV-interleaved shadow plane (compile-time: the two chroma paths must not share a register allocation)
template <int BPS, int BW, bool UV, int STREAM_MAX = MVX_STREAM_MAX> struct FastSearcher {
    typedef FGeo<BPS, BW> G;
    const AParams &P;
    const AJob &J;
    lds_u8 *lds;     // [source block | previous-row results, 16 B per block | histogram]
    int ldsRow, ldsHist, histBins;

    // level constants (uniform)
    int nBlkX, nBlkY, pel, logPel, pw, ph, hpad, vpad;
    gl_u8 *srcY, *srcU, *srcV, *refY, *refU, *refV;
    gl_u8 *srcUV, *refUV;      // level bases inside the UV-interleaved shadow planes (uv != 0)
    unsigned pitchY, pitchC, pstrideY, pstrideC;
    unsigned shadowY;          // byte distance from a luma plane to its copy shifted by one sample, 0 = none (mvx_super_shadow_frames)
    static constexpr bool uv = UV;
    GL_AS GVec *vectors;
    int chroma, searchType, nSearchParam, penaltyNew, penaltyZero, pglobal, badrange, badcount, fieldShift;
    long long badSAD, LSAD;
    int gmvx, gmvy; // global motion predictor of this level, cumulatively clipped (PlaneOfBlocks.cpp:859)

    // block state (uniform)
    int x0, y0, blkIdx;
    int nDxMin, nDyMin, nDxMax, nDyMax;
    int predX, predY;          // predictor (:1100 / :449)
    int pX[4], pY[4];          // predictors[0..3]
    int nLambda;               // (0 <= nLambda <= the level's lambda < 2^31: mvx_fast_eligible)
    int bestX, bestY, bestSad; // bestMV
    int nMinCost;              // 0x7fffffff = nothing accepted yet (every real cost is smaller: see cost32)
#ifdef MVX_FAST_PROF
    long long prof[FPROF_N], tp; // 0 loop top + barrier, 1 group fetch + source block, 2 limits / predictors / lambda, 3 predictor pass: loads + SADs, 4 its sums + acceptance,
                                 // 5 hexagon pass: loads + SADs, 6 its sums + acceptance, 7 square pass, 8 other refinement, 9 rescue, 10 result, 11 blocks counted
#endif

    __device__ FastSearcher(const AParams &p, const AJob &j) : P(p), J(j) {}

    __device__ __forceinline__ bool vector_ok(int vx, int vy) const { return vx >= nDxMin && vy >= nDyMin && vx < nDxMax && vy < nDyMax; }
    __device__ __forceinline__ int clipx(int v) const { return min(max(v, nDxMin), nDxMax - 1); }
    __device__ __forceinline__ int clipy(int v) const { return min(max(v, nDyMin), nDyMax - 1); }
    // PlaneOfBlocks.cpp:105-114
    __device__ __forceinline__ int motion_distortion(int vx, int vy) const {
        const unsigned dx = (unsigned)(predX - vx), dy = (unsigned)(predY - vy);
        const int dist = (int)(dx * dx + dy * dy);
        return (int)(((long long)nLambda * dist) >> 8); // one signed 32 x 32 -> 64 multiply
    }
    __device__ __forceinline__ static int sat_add(int a, int b) { // b >= 0
        const long long r = (long long)a + b;
        return r > 0x7fffffffLL ? 0x7fffffff : (int)r;
    }

    // byte offsets of a candidate's reference block inside the level's plane set (PlaneOfBlocks.cpp:35-101, MVFrame.cpp:1707-1729)
    // With shadow copies the block at sample position x is read from copy k = x % (4 / BPS), where it starts at a dword-aligned
    // address (copy k = the plane shifted left by k samples): same samples, ~3.7x cheaper for the CU's texture path.
    __device__ __forceinline__ static unsigned shadow_off(unsigned off, unsigned shadow) {
        if (!shadow) return off;
        const unsigned k = BPS == 2 ? (off >> 1) & 1u : off & 3u;
        return (off & ~3u) + k * shadow;
    }
    __device__ __forceinline__ unsigned ref_luma_off(int vx, int vy) const {
        const int ax = (x0 << logPel) + vx, ay = (y0 << logPel) + vy, m = pel - 1;
        const unsigned idx = (unsigned)((ax & m) | ((ay & m) << logPel));
        return shadow_off(idx * pstrideY + (unsigned)(ay >> logPel) * pitchY + (unsigned)(ax >> logPel) * BPS, shadowY);
    }
    __device__ __forceinline__ unsigned ref_chroma_off(int vx, int vy) const { // 4:2:0; block origins are even (even hpad/vpad/steps: checked by the host)
        const int xb = vx < 0 ? 1 : 0, yb = vy < 0 ? 1 : 0;
        const int ax = ((x0 >> 1) << logPel) + ((vx + xb) >> 1), ay = ((y0 >> 1) << logPel) + ((vy + yb) >> 1), m = pel - 1;
        const unsigned idx = (unsigned)((ax & m) | ((ay & m) << logPel));
        return idx * pstrideC + (unsigned)(ay >> logPel) * pitchC + (unsigned)(ax >> logPel) * BPS; // (in the UV plane: twice this, always dword-aligned)
    }

    // ---- SAD of this lane's share of one plane region (T pieces of CB bytes, 1 << LOGC pieces per row) for a candidate that
    // 1 << LOGG lanes share: source from LDS, reference from global memory, NB loads in flight.
    template <int CB> __device__ __forceinline__ static v4u ld_ref(gl_u8 *q) {
        if (CB == 16) { uv4 t = *(GL_AS const uv4 *)q; return v4u{t[0], t[1], t[2], t[3]}; }
        else if (CB == 8) { uv2 t = *(GL_AS const uv2 *)q; return v4u{t[0], t[1], 0, 0}; }
        else if (CB == 4) return v4u{*(GL_AS const uv1 *)q, 0, 0, 0};
        else return v4u{*(GL_AS const uh1 *)q, 0, 0, 0};
    }
    // one CB-byte piece of the source block from LDS / its SAD against the reference piece r
    template <int CB> __device__ __forceinline__ static v4u lds_piece(const lds_u8 *l) {
        if (CB == 16) return *(const LDS_AS v4u *)l;
        else if (CB == 8) { const v2u a = *(const LDS_AS v2u *)l; return v4u{a[0], a[1], 0, 0}; }
        else if (CB == 4) return v4u{*(const LDS_AS unsigned *)l, 0, 0, 0};
        else return v4u{*(const LDS_AS unsigned short *)l, 0, 0, 0};
    }
    template <int CB> __device__ __forceinline__ static unsigned sad_regs(const v4u &a, const v4u &r, unsigned acc) {
        acc = sad32<BPS>(a[0], r[0], acc);
        if (CB >= 8) acc = sad32<BPS>(a[1], r[1], acc);
        if (CB == 16) { acc = sad32<BPS>(a[2], r[2], acc); acc = sad32<BPS>(a[3], r[3], acc); }
        return acc;
    }
    template <int CB> __device__ __forceinline__ static unsigned sad_piece(const lds_u8 *l, const v4u &r, unsigned acc) {
        return sad_regs<CB>(lds_piece<CB>(l), r, acc);
    }
    template <int LOGG, int T, int LOGC, int CB, int ROWB>
    __device__ __forceinline__ unsigned region(int s, const lds_u8 *src, gl_u8 *base, unsigned off, unsigned refPitch, unsigned acc) const {
        constexpr int GG = 1 << LOGG, C = 1 << LOGC;
        if (T < GG) { // fewer pieces than lanes in the group: lanes s < T own one piece each
            if (s < T) {
                const int row = s >> LOGC, xb = (s & (C - 1)) * CB;
                acc = sad_piece<CB>(src + row * ROWB + xb, ld_ref<CB>(base + (off + (unsigned)row * refPitch + (unsigned)xb)), acc);
            }
            return acc;
        }
        constexpr int N = T >= GG ? T / GG : 1, NB = N < 4 ? N : 4; // pieces per lane, loads in flight
        if (GG >= C) { // the piece column is fixed per lane, rows advance by GG / C per piece
            const int row0 = s >> LOGC, xb = (s & (C - 1)) * CB;
            unsigned po = off + (unsigned)row0 * refPitch + (unsigned)xb;
            const lds_u8 *sp = src + row0 * ROWB + xb;
            const unsigned step = (unsigned)(GG >> LOGC) * refPitch;
            constexpr int lstep = (GG >> LOGC) * ROWB;
            // rolling window: NB loads in flight, the register of a consumed piece is reloaded at once (issuing NB, consuming NB,
            // issuing the next NB costs one memory round trip per batch)
            constexpr bool ROLL = N > NB && N <= 8;
            if (ROLL) {
                v4u r[NB];
                auto issue = [&]() { const v4u v = ld_ref<CB>(base + po); po += step; asm("" : "+v"(po)); return v; };
#pragma unroll
                for (int k = 0; k < NB; k++) r[k] = issue();
                constexpr int D = MVX_SRC_AHEAD < N ? MVX_SRC_AHEAD : N; // the LDS reads run D pieces ahead
                v4u a[D];
#pragma unroll
                for (int k = 0; k < D; k++) a[k] = lds_piece<CB>(sp + k * lstep);
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const v4u cur = a[k % D];
                    if (k + D < N) a[k % D] = lds_piece<CB>(sp + (k + D) * lstep);
                    acc = sad_regs<CB>(cur, r[k % NB], acc);
                    if (k + NB < N) r[k % NB] = issue();
                }
                return acc;
            }
#pragma unroll 2
            for (int k0 = 0; k0 < N; k0 += NB) {
                v4u r[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    r[k] = ld_ref<CB>(base + po);
                    po += step;
                    asm("" : "+v"(po)); // a running offset: one add per piece (as po + k * step the compiler multiplies per piece)
                }
#pragma unroll
                for (int k = 0; k < NB; k++) acc = sad_piece<CB>(sp + (k0 + k) * lstep, r[k], acc);
            }
        } else { // several lanes' worth of pieces per row
#pragma unroll 2
            for (int k0 = 0; k0 < N; k0 += NB) {
                v4u r[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int t = s + (k0 + k) * GG, row = t >> LOGC, xb = (t & (C - 1)) * CB;
                    r[k] = ld_ref<CB>(base + (off + (unsigned)row * refPitch + (unsigned)xb));
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int t = s + (k0 + k) * GG, row = t >> LOGC, xb = (t & (C - 1)) * CB;
                    acc = sad_piece<CB>(src + row * ROWB + xb, r[k], acc);
                }
            }
        }
        return acc;
    }
    // two regions (luma, then the UV plane) as ONE stream of loads with a rolling window of W in flight: the chroma loads are
    // requested while the luma pieces are still being consumed.  Both regions must be of the "piece column fixed per lane" kind.
    template <int LOGG, int TA, int LOGCA, int CBA, int ROWBA, int TB, int LOGCB, int CBB, int ROWBB>
    __device__ __forceinline__ void region2(int s, const lds_u8 *srcA, gl_u8 *baseA, unsigned offA, unsigned pitchA, unsigned &accA,
                                            const lds_u8 *srcB, gl_u8 *baseB, unsigned offB, unsigned pitchB, unsigned &accB) const {
        constexpr int GG = 1 << LOGG, CA = 1 << LOGCA, CB_ = 1 << LOGCB;
        constexpr int NA = TA / GG, NBB = TB / GG, NT = NA + NBB, W = NT < MVX_INFLIGHT ? NT : MVX_INFLIGHT;
        const int rowA = s >> LOGCA, xbA = (s & (CA - 1)) * CBA, rowB = s >> LOGCB, xbB = (s & (CB_ - 1)) * CBB;
        unsigned poA = offA + (unsigned)rowA * pitchA + (unsigned)xbA, poB = offB + (unsigned)rowB * pitchB + (unsigned)xbB;
        const lds_u8 *spA = srcA + rowA * ROWBA + xbA, *spB = srcB + rowB * ROWBB + xbB;
        const unsigned stepA = (unsigned)(GG >> LOGCA) * pitchA, stepB = (unsigned)(GG >> LOGCB) * pitchB;
        constexpr int lstepA = (GG >> LOGCA) * ROWBA, lstepB = (GG >> LOGCB) * ROWBB;
        v4u r[W];
        // (the memory clobber keeps the loads in program order: the scheduler otherwise likes to issue the FIRST piece's load last,
        // and the first use then waits for all of them)
        auto issueA = [&]() { const v4u v = ld_ref<CBA>(baseA + poA); poA += stepA; asm volatile("" : "+v"(poA) : : "memory"); return v; };
        auto issueB = [&]() { const v4u v = ld_ref<CBB>(baseB + poB); poB += stepB; asm volatile("" : "+v"(poB) : : "memory"); return v; };
#pragma unroll
        for (int k = 0; k < W; k++) r[k] = k < NA ? issueA() : issueB();
        // source pieces D ahead of their use (see region); the compiler barrier inside issueA / issueB keeps the distance
        constexpr int D = MVX_SRC_AHEAD < NT ? MVX_SRC_AHEAD : NT;
        auto src_piece = [&](int k) { return k < NA ? lds_piece<CBA>(spA + k * lstepA) : lds_piece<CBB>(spB + (k - NA) * lstepB); };
        v4u a[D];
#pragma unroll
        for (int k = 0; k < D; k++) a[k] = src_piece(k);
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const v4u cur = a[k % D];
            if (k + D < NT) a[k % D] = src_piece(k + D);
            if (k < NA) accA = sad_regs<CBA>(cur, r[k % W], accA);
            else accB = sad_regs<CBB>(cur, r[k % W], accB);
            if (k + W < NT) r[k % W] = (k + W) < NA ? issueA() : issueB();
        }
    }
    // partial SADs (this lane's share) of candidate (vx, vy); vyc = the vertical component the chroma planes use (:836-839)
    template <int LOGG> __device__ __forceinline__ void eval(int s, int vx, int vy, int vyc, unsigned &aL, unsigned &aC) const {
        constexpr int GG = 1 << LOGG;
        constexpr bool STREAM = UV && G::LT >= GG && G::UVT >= GG && GG >= (1 << G::LLOGC) && GG >= (1 << G::UVLOGC) && (G::LT + G::UVT) / GG >= 2 && (G::LT + G::UVT) / GG <= STREAM_MAX;
        if constexpr (STREAM) { // (region2 does not exist for shapes with fewer pieces than lanes)
            if (chroma) {
                const unsigned co = ref_chroma_off(vx, vyc);
                region2<LOGG, G::LT, G::LLOGC, G::LCB, G::LROWB, G::UVT, G::UVLOGC, G::UVCB, G::UVROWB>(s, lds, refY, ref_luma_off(vx, vy), pitchY, aL, lds + G::UOFF, refUV, 2 * co, 2 * pitchC, aC);
                return;
            }
        }
        aL = region<LOGG, G::LT, G::LLOGC, G::LCB, G::LROWB>(s, lds, refY, ref_luma_off(vx, vy), pitchY, aL);
        if (chroma) {
            const unsigned co = ref_chroma_off(vx, vyc);
            if (uv) aC = region<LOGG, G::UVT, G::UVLOGC, G::UVCB, G::UVROWB>(s, lds + G::UOFF, refUV, 2 * co, 2 * pitchC, aC); // U and V in one pass
            else {
                aC = region<LOGG, G::CT, G::CLOGC, G::CCB, G::CROWB>(s, lds + G::UOFF, refU, co, pitchC, aC);
                aC = region<LOGG, G::CT, G::CLOGC, G::CCB, G::CROWB>(s, lds + G::VOFF, refV, co, pitchC, aC);
            }
        }
    }

    // Acceptance of a pass: every lane of a candidate's group holds the candidate's cost (0x7fffffff = not a candidate / not
    // better); groups are ordered by lane, so the lowest lane with the minimum is the FIRST minimal candidate -- the one the
    // reference's sequential strict `<` update ends on (PlaneOfBlocks.cpp:229,239,248).  Returns the winning lane or -1.
    // wave minimum of values that are uniform inside aligned groups of 1 << LOGG lanes: the row_shr steps below the group size
    // would compare a group with itself and are skipped (lane 15 of a row still meets one lane of every group of its row)
    template <int LOGG> __device__ __forceinline__ static unsigned group_min_u32(unsigned v) {
        if (LOGG <= 0) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
        if (LOGG <= 1) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(v));
        if (LOGG <= 2) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
        asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                     : "+v"(v)); // (the compiler puts the wait state a v_readlane of an asm output needs in front of it)
        return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    }
    template <int LOGG> __device__ __forceinline__ int accept(int cost, int tot) {
        int mc = 0, w = -1;
        {
            const unsigned u = (unsigned)cost ^ 0x80000000u;
            const unsigned m = group_min_u32<LOGG>(u);
            if (m != 0xffffffffu) { mc = (int)(m ^ 0x80000000u); w = __ffsll((long long)__ballot(u == m)) - 1; }
        }
        if (w >= 0) { nMinCost = mc; bestSad = bcast_i(tot, w); }
        return w;
    }
    // cost of a refinement candidate: pobCheckMV (:219-261, penalty for new vectors); saturating -- a saturated cost never wins
    __device__ __forceinline__ int cost_new(int vx, int vy, unsigned aL, unsigned aC) const {
        int cc = (int)aL + ((penaltyNew * (int)aL) >> 8);
        if (chroma) cc += (int)aC + ((penaltyNew * (int)aC) >> 8);
        return sat_add(motion_distortion(vx, vy), cc);
    }

    enum { K_SQUARE, K_HEXSQ, K_EXH2 };
    // one pass of the default refinement around (cx, cy).  K_SQUARE: the 8 points of pobExpandingSearch(1, 1) (:636-658).
    // K_HEXSQ: the hexagon (:682-687) and, speculatively, the square around the SAME centre; the square's results are used
    // only when no hexagon point improved the cost (then the reference runs exactly that square against the unchanged
    // nMinCost).  K_EXH2: rings 1 and 2 (:786-791).  Returns the index of the winning candidate, -1 if none.
    template <int KIND> __device__ __forceinline__ int refine_pass(int cx, int cy) {
        constexpr int LOGG = KIND == K_EXH2 ? 1 : KIND == K_HEXSQ ? 2 : 3;
        constexpr int TOTAL = KIND == K_SQUARE ? 8 : KIND == K_HEXSQ ? 14 : 24;
        const int lane = lane_id();
        const int g = lane >> LOGG, s = lane & ((1 << LOGG) - 1);
        int dx, dy;
        if (KIND == K_HEXSQ) {
            const int k = (g + 2) & 7; // square index of groups 6..13
            dx = g < 6 ? tab8(HEX2X >> 8, g & 7) : tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), k);
            dy = g < 6 ? tab8(HEX2Y >> 8, g & 7) : tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), k);
        } else if (KIND == K_SQUARE) { dx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), g & 7); dy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), g & 7); }
        else {
            const int k = g < 8 ? g : g - 8;
            if (g < 8) { dx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), k); dy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), k); }
            else if (k < 8) { dx = tab8(PACK8(-1, -1, 0, 0, 1, 1, -2, 2), k); dy = tab8(PACK8(-2, 2, -2, 2, -2, 2, -1, -1), k); }
            else { dx = tab8(PACK8(-2, 2, -2, 2, -2, -2, 2, 2), k - 8); dy = tab8(PACK8(0, 0, 1, 1, -2, 2, -2, 2), k - 8); }
        }
        const int vx = cx + dx, vy = cy + dy;
        const bool ok = g < TOTAL && vector_ok(vx, vy);
        unsigned aL = 0, aC = 0;
        if (ok) eval<LOGG>(s, vx, vy, vy, aL, aC);
        if (KIND == K_HEXSQ) FPROF(5, tp);
        group_sum2<LOGG>(aL, aC);
        const int tot = (int)aL + (chroma ? (int)aC : 0);
        const int cc = cost_new(vx, vy, aL, aC);
        const bool first = KIND != K_HEXSQ || g < 6;
        int w = accept<LOGG>((ok && first && cc < nMinCost) ? cc : 0x7fffffff, tot);
        if (w >= 0) {
            if (KIND != K_HEXSQ) { bestX = bcast_i(vx, w); bestY = bcast_i(vy, w); } // (the hexagon is pobCheckMVdir: bestMV.x/y untouched)
            if (KIND == K_HEXSQ) FPROF(6, tp);
            return w >> LOGG;
        }
        if (KIND == K_HEXSQ) {
            w = accept<LOGG>((ok && !first && cc < nMinCost) ? cc : 0x7fffffff, tot);
            if (w >= 0) { bestX = bcast_i(vx, w); bestY = bcast_i(vy, w); }
            FPROF(6, tp);
        }
        return -1;
    }

    // pobPseudoEPZSearch (:819-968) for the default parameters
    __device__ __forceinline__ void search_block() {
        // ---- the predictor set (:832-915): zero, global, hierarchical predictor, median, left, up, ahead; eight lanes each
        gmvx = clipx(gmvx); gmvy = clipy(gmvy); // cumulative clip (:859)
        {
            const int lane = lane_id();
            int g = lane >> 3;
            const int s = lane & 7;
            asm("" : "+v"(g)); // keeps the (g == k) masks out of scalar registers across the block loop
            // the seven vectors, packed (|x|, |y| < 30000: mvx_fast_eligible), go to the first lane of their groups; two DPP moves spread
            // them over the eight lanes (quad broadcast, then lanes 4..7 of every group copy lanes 0..3: row_shr:4 into banks 1 and 3)
            auto pk = [](int x, int y) { return (int)(((unsigned)x & 0xffffu) | ((unsigned)y << 16)); };
            const int pc[7] = {pk(0, fieldShift), uni(pk(gmvx, gmvy)), uni(pk(predX, predY)), uni(pk(pX[0], pY[0])), uni(pk(pX[1], pY[1])), uni(pk(pX[2], pY[2])), uni(pk(pX[3], pY[3]))};
            int c = pc[0];
            asm("" : "+v"(c));
#pragma unroll
            for (int k = 1; k < 7; k++) asm("v_writelane_b32 %0, %1, %2" : "+v"(c) : "s"(pc[k]), "n"(8 * k));
            c = __builtin_amdgcn_update_dpp(c, c, 0x00, 0xf, 0xf, false);  // quad_perm:[0,0,0,0]
            c = __builtin_amdgcn_update_dpp(c, c, 0x114, 0xf, 0xa, false); // row_shr:4, banks 1 and 3
            int vx = (int)(short)(c & 0xffff), vy = c >> 16;
            const int vyc = g == 0 ? 0 : vy; // the zero candidate's chroma ignores fieldShift (:836-839)
            const bool ok = g < 7;            // (all of them are clipped vectors)
            unsigned aL = 0, aC = 0;
            // Predictors repeat (a still or evenly moving area: most of the seven are one vector).  The SADs depend on the vector alone, so
            // a group whose vector an EARLIER group has does not load its block: it takes that group's sums (the costs are still computed
            // per group, with the group's own penalty; where two costs tie the earlier group wins anyway, as in the reference's sequential
            // strict `<`).  The zero candidate is a source only without a field shift (its chroma ignores the shift).
            int srcg = g;
            {
#pragma unroll
                for (int k = 5; k >= 0; k--) { // (min: a group at or before k keeps itself)
                    const bool same = c == pc[k] && (k > 0 || fieldShift == 0);
                    srcg = same ? min(srcg, k) : srcg;
                }
            }
            FPROF(2, tp);
            if (ok && srcg == g) eval<3>(s, vx, vy, vyc, aL, aC);
            FPROF(3, tp);
            group_sum2<3>(aL, aC);
            aL = (unsigned)__builtin_amdgcn_ds_bpermute(srcg << 5, (int)aL); // lane 8 * srcg: every lane of a group holds the group's sums
            aC = (unsigned)__builtin_amdgcn_ds_bpermute(srcg << 5, (int)aC);
            const int tot = (int)aL + (chroma ? (int)aC : 0);
            const int pen = g == 0 ? penaltyZero : (g == 1 ? pglobal : 0);                 // :846, :870, :894
            int cc = tot + (int)(((long long)pen * tot) >> 8);
            cc = sat_add(g >= 3 ? motion_distortion(vx, vy) : 0, cc);                       // pobCheckMV0: no new-vector penalty
            nMinCost = 0x7fffffff;
            const int w = accept<3>(ok ? cc : 0x7fffffff, tot); // (group 0 always has a finite cost)
            bestX = bcast_i(vx, w); bestY = bcast_i(vy, w);
            FPROF(4, tp);
        }
        // ---- pobRefine (:773-816)
        if (searchType == SearchHex2) { // pobHex2Search :667-724 with i_me_range <= 3: no half-hexagon iterations
            if (nSearchParam > 1) {
                const int bmx = bestX, bmy = bestY;
                // (without the speculative square -- 8 fewer candidate blocks whenever a hexagon point wins, one more pass whenever none
                // does -- the launch takes 8 % longer: DESIGN.md 4.2.3)
                const int dir = refine_pass<K_HEXSQ>(bmx, bmy); // >= 0: a hexagon point won; < 0: the square around (bmx, bmy) is done too
                if (dir >= 0) {
                    const int nx = bmx + tab8(HEX2X, dir + 1), ny = bmy + tab8(HEX2Y, dir + 1);
                    bestX = nx; bestY = ny;
                    refine_pass<K_SQUARE>(nx, ny);
                    FPROF(7, tp);
                }
            } else {
                refine_pass<K_SQUARE>(bestX, bestY);
                FPROF(7, tp);
            }
        } else {
            refine_pass<K_EXH2>(bestX, bestY);
            FPROF(8, tp);
        }
        // ---- bad vector: wide search (:938-963)
        if (__builtin_expect(blkIdx > 1 && (long long)bestSad > badSAD + badSAD * badcount / 16, 0)) { rescue(); FPROF(9, tp); }
    }

    // candidate i of the rescue patterns, in the reference's order
    enum { R_CROSS, R_HEX4, R_HEX6, R_HEX3, R_RING };
    __device__ __forceinline__ static void rescue_cand(int kind, int i, int a, int b, int &dx, int &dy) {
        dx = 0; dy = 0;
        switch (kind) {
        case R_CROSS: // pobCrossSearch :728-739, a = number of odd offsets per axis
            if (i < 2 * a) { const int o = 1 + 2 * (i >> 1); dx = (i & 1) ? o : -o; }
            else { const int k = i - 2 * a, o = 1 + 2 * (k >> 1); dy = (k & 1) ? o : -o; }
            break;
        case R_HEX4: { const int r = 1 + (i >> 4), j = i & 15; // :753-764
            dx = (j < 8 ? tab8(HEX4XA, j) : tab8(HEX4XB, j - 8)) * r; dy = (j < 8 ? tab8(HEX4YA, j) : tab8(HEX4YB, j - 8)) * r; break; }
        case R_HEX6: { const int j = min(i + 1, 7); dx = tab8(HEX2X, j); dy = tab8(HEX2Y, j); break; }     // hex2[i + 1], :682-687
        case R_HEX3: { const int j = min(a + i, 7); dx = tab8(HEX2X, j); dy = tab8(HEX2Y, j); break; }     // hex2[odir + i], :706-708
        default: { // R_RING: pobExpandingSearch(r = a, s = b) :636-658: sides without corners (x then y), then the corners
            const int r = a, st = b;
            int cnt = 0; // points per side: i = -r + s, -r + 2s, ... < r
            for (int t = -r + st; t < r; t += st) cnt++;
            if (i < 2 * cnt) { dx = -r + st + (i >> 1) * st; dy = (i & 1) ? r : -r; }
            else if (i < 4 * cnt) { const int k = i - 2 * cnt; dy = -r + st + (k >> 1) * st; dx = (k & 1) ? r : -r; }
            else { const int k = i - 4 * cnt; dx = (k & 2) ? r : -r; dy = (k & 1) ? r : -r; }
            break; }
        }
    }
    // (all of this is force-inlined: an out-of-line member function would pin the whole searcher state in scratch memory)
    // `total` candidates of one pattern around (cx, cy), eight per pass in reference order.  update: pobCheckMV (bestMV.x/y follow)
    // or pobCheckMVdir (only the cost / SAD; the caller reads the winner's index).  Returns the index of the last accepted
    // candidate (the overall first minimum), -1 if none improved.
    __device__ __forceinline__ int rescue_round(int kind, int total, int a, int b, int cx, int cy, bool update) {
        const int lane = lane_id();
        const int g = lane >> 3, s = lane & 7;
        int winner = -1;
        for (int base = 0; base < total; base += 8) {
            const int i = base + g;
            int dx, dy;
            rescue_cand(kind, i, a, b, dx, dy);
            const int vx = cx + dx, vy = cy + dy;
            const bool ok = i < total && vector_ok(vx, vy);
            unsigned aL = 0, aC = 0;
            if (ok) eval<3>(s, vx, vy, vy, aL, aC);
            group_sum2<3>(aL, aC);
            const int tot = (int)aL + (chroma ? (int)aC : 0);
            const int cc = cost_new(vx, vy, aL, aC);
            const int w = accept<3>((ok && cc < nMinCost) ? cc : 0x7fffffff, tot);
            if (w >= 0) {
                if (update) { bestX = bcast_i(vx, w); bestY = bcast_i(vy, w); }
                winner = base + (w >> 3);
            }
        }
        return winner;
    }
    __device__ __forceinline__ void rescue() {
        const int foundSAD = bestSad;
        badcount++;
        if (badrange > 0) { // pobUMHSearch(range, 0, 0) :743-769
            const int range = badrange * pel;
            int nh = 0;
            for (int i = 1; i < range; i += 2) nh++;
            if (nh > 0) rescue_round(R_CROSS, 4 * nh, nh, 0, 0, 0, true);
            int nrings = 0;
            { int i = 1; do { nrings++; } while (++i <= range / 4); }
            rescue_round(R_HEX4, 16 * nrings, 0, 0, 0, 0, true);
            // pobHex2Search(range) :667-724
            int bmx = bestX, bmy = bestY;
            if (range > 1) {
                int dir = rescue_round(R_HEX6, 6, 0, 0, bmx, bmy, false);
                if (dir >= 0) {
                    bmx += tab8(HEX2X, dir + 1); bmy += tab8(HEX2Y, dir + 1);
                    for (int it = 1; it < range / 2 && vector_ok(bmx, bmy); it++) { // half hexagons, not overlapping the previous iteration
                        const int odir = (dir + 1 + 5) % 6; // mod6m1[dir + 1]
                        const int w = rescue_round(R_HEX3, 3, odir, 0, bmx, bmy, false);
                        if (w < 0) break;
                        dir = odir - 1 + w;
                        bmx += tab8(HEX2X, dir + 1); bmy += tab8(HEX2Y, dir + 1);
                    }
                }
                bestX = bmx; bestY = bmy;
            }
            rescue_round(R_RING, 8, 1, 1, bmx, bmy, true);
        } else if (badrange < 0) { // expanding rings around zero (:951-955)
            for (int i = 1; i < -badrange * pel; i += pel) {
                int cnt = 0;
                for (int t = -i + pel; t < i; t += pel) cnt++;
                rescue_round(R_RING, 4 * cnt + 4, i, pel, 0, 0, true);
                if (bestSad < foundSAD / 4) break;
            }
        }
        const int mvx = bestX, mvy = bestY; // refine in a small area (:958-962)
        for (int i = 1; i < pel; i++) {
            int cnt = 0;
            for (int t = -i + 1; t < i; t++) cnt++;
            rescue_round(R_RING, 4 * cnt + 4, i, 1, mvx, mvy, true);
        }
    }

    // ---- per-level pieces shared with analyse_kernel's semantics ----------------------------------------------------
    __device__ static Vec ld_vec(GL_AS const GVec *p) { Vec v; v.x = p->x; v.y = p->y; v.sad = p->sad; return v; }
    __device__ static void st_vec(GL_AS GVec *p, const Vec &v) { p->x = v.x; p->y = v.y; p->sad = v.sad; }

    // pobInterpolatePrediction (:1447-1514) straight into vectors[], or zero (pobInit :355)
    // (first / stride: the share of this wave when several waves of a workgroup fill one plane -- the team form of the speculative kernel)
    __device__ __forceinline__ void interpolate(GL_AS const GVec *coarse, int coarseBlkX, int coarseBlkY, int coarseLogPel, int first = -1, int stride = WAVE) {
        const int l = first < 0 ? lane_id() : first;
        const int nBlk = nBlkX * nBlkY;
        if (!coarse) {
            for (int i = l; i < nBlk; i += stride) { Vec z; z.x = 0; z.y = 0; z.sad = 0; st_vec(&vectors[i], z); }
            return;
        }
        int normFactor = 3 - logPel + coarseLogPel;
        const int mulFactor = normFactor < 0 ? -normFactor : 0;
        normFactor = normFactor < 0 ? 0 : normFactor;
        const int normov = (P.blkX - P.ovX) * (P.blkY - P.ovY);
        const int aoddx = P.blkX * 3 - P.ovX * 2, aevenx = P.blkX * 3 - P.ovX * 4;
        const int aoddy = P.blkY * 3 - P.ovY * 2, aeveny = P.blkY * 3 - P.ovY * 4;
        const double scaleov = 1.0 / normov;
        for (int index = l; index < nBlk; index += stride) {
            const int ly = index / nBlkX, k = index - ly * nBlkX;
            int i = k, j = ly;
            if (i >= 2 * coarseBlkX) i = 2 * coarseBlkX - 1;
            if (j >= 2 * coarseBlkY) j = 2 * coarseBlkY - 1;
            const int offy = -1 + 2 * (j % 2), offx = -1 + 2 * (i % 2);
            Vec v1, v2, v3, v4;
            const bool ex = (i == 0) || (i >= 2 * coarseBlkX - 1), ey = (j == 0) || (j >= 2 * coarseBlkY - 1);
            v1 = ld_vec(&coarse[i / 2 + (j / 2) * coarseBlkX]);
            if (ex && ey) { v2 = v3 = v4 = v1; }
            else if (ex) { v2 = v1; v3 = v4 = ld_vec(&coarse[i / 2 + (j / 2 + offy) * coarseBlkX]); }
            else if (ey) { v2 = v1; v3 = v4 = ld_vec(&coarse[i / 2 + offx + (j / 2) * coarseBlkX]); }
            else {
                v2 = ld_vec(&coarse[i / 2 + offx + (j / 2) * coarseBlkX]);
                v3 = ld_vec(&coarse[i / 2 + (j / 2 + offy) * coarseBlkX]);
                v4 = ld_vec(&coarse[i / 2 + offx + (j / 2 + offy) * coarseBlkX]);
            }
            Vec o; long long temp_sad;
            if (P.ovX == 0 && P.ovY == 0) {
                o.x = 9 * v1.x + 3 * v2.x + 3 * v3.x + v4.x;
                o.y = 9 * v1.y + 3 * v2.y + 3 * v3.y + v4.y;
                temp_sad = 9 * v1.sad + 3 * v2.sad + 3 * v3.sad + v4.sad + 8;
            } else if (P.ovX <= (P.blkX >> 1) && P.ovY <= (P.blkY >> 1)) {
                const int ax1 = (offx > 0) ? aoddx : aevenx, ax2 = (P.blkX - P.ovX) * 4 - ax1;
                const int ay1 = (offy > 0) ? aoddy : aeveny, ay2 = (P.blkY - P.ovY) * 4 - ay1;
                const long long a11 = ax1 * ay1, a12 = ax1 * ay2, a21 = ax2 * ay1, a22 = ax2 * ay2;
                o.x = (int)((double)(a11 * v1.x + a21 * v2.x + a12 * v3.x + a22 * v4.x) * scaleov);
                o.y = (int)((double)(a11 * v1.y + a21 * v2.y + a12 * v3.y + a22 * v4.y) * scaleov);
                temp_sad = (long long)((double)(a11 * v1.sad + a21 * v2.sad + a12 * v3.sad + a22 * v4.sad) * scaleov);
            } else {
                o.x = (v1.x + v2.x + v3.x + v4.x) << 2;
                o.y = (v1.y + v2.y + v3.y + v4.y) << 2;
                temp_sad = (v1.sad + v2.sad + v3.sad + v4.sad + 2) << 2;
            }
            o.x = (o.x >> normFactor) * (1 << mulFactor);
            o.y = (o.y >> normFactor) * (1 << mulFactor);
            o.sad = temp_sad >> 4;
            st_vec(&vectors[index], o);
        }
    }

    // pobEstimateGlobalMVDoubled, PlaneOfBlocks.cpp:1559-1636 (mode via LDS histogram windows; first maximum wins)
    __device__ __forceinline__ void estimate_global(GL_AS const GVec *v, int nBlk, int freqSizeHalf, int *gx, int *gy) {
        const int l = lane_id();
        LDS_AS int *hist = (LDS_AS int *)(lds + ldsHist);
        int med[2];
        for (int c = 0; c < 2; c++) {
            int lo = 0x7fffffff, hi = -0x7fffffff - 1;
            for (int i = l; i < nBlk; i += WAVE) {
                const int val = c ? v[i].y : v[i].x;
                const int ind = freqSizeHalf + val;
                if (ind >= 0 && ind < 2 * freqSizeHalf) { lo = min(lo, val); hi = max(hi, val); }
            }
            lo = wave_min_i32(lo); hi = wave_max_i32(hi);
            int bestCount = -1, bestVal = lo;
            for (int wbase = lo; wbase <= hi; wbase += histBins) {
                for (int i = l; i < histBins; i += WAVE) hist[i] = 0;
                __builtin_amdgcn_wave_barrier();
                for (int i = l; i < nBlk; i += WAVE) {
                    const int val = c ? v[i].y : v[i].x;
                    const int ind = freqSizeHalf + val;
                    if (ind >= 0 && ind < 2 * freqSizeHalf && val >= wbase && val < wbase + histBins)
                        __hip_atomic_fetch_add(&hist[val - wbase], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                int lc = -1, lv = 0; // first maximum in ascending order within this lane's stride
                for (int i = l; i < histBins && wbase + i <= hi; i += WAVE) {
                    const int cnt = hist[i];
                    if (cnt > lc) { lc = cnt; lv = i; }
                }
                const int mcnt = wave_max_i32(lc);
                const int cand = (lc == mcnt) ? lv : 0x7fffffff;
                const int mv = wave_min_i32(cand);
                if (mcnt > bestCount) { bestCount = mcnt; bestVal = wbase + mv; }
                __builtin_amdgcn_wave_barrier();
            }
            med[c] = bestVal;
        }
        int sx = 0, sy = 0, n = 0;
        for (int i = l; i < nBlk; i += WAVE) {
            const int vx = v[i].x, vy = v[i].y;
            if (abs(vx - med[0]) < 6 && abs(vy - med[1]) < 6) { sx += vx; sy += vy; n++; }
        }
        sx = wave_sum_i32(sx); sy = wave_sum_i32(sy); n = wave_sum_i32(n);
        if (n > 0) { *gx = uni(2 * sx / n); *gy = uni(2 * sy / n); }
        else { *gx = uni(2 * med[0]); *gy = uni(2 * med[1]); }
    }

    // ---- source block staging: every lane owns the same NPF pieces of every block (offsets computed once per level)
    int pfG[G::NPF], pfL[G::NPF], pfP[G::NPF]; // per lane: global row/col offset, LDS offset, plane (0, 1, 2; -1: none)
    __device__ __forceinline__ void pf_setup() {
        const int l = lane_id();
#pragma unroll
        for (int k = 0; k < G::NPF; k++) {
            const int t = l + k * WAVE;
            const int TT = G::LT + (chroma ? (uv ? G::UVT : 2 * G::CT) : 0);
            if (t < G::LT) {
                const int row = t / G::LC, xb = (t % G::LC) * G::LCB;
                pfP[k] = 0; pfG[k] = (int)(row * pitchY) + xb; pfL[k] = row * G::LROWB + xb;
            } else if (t < TT && uv) { // one piece of a UV row
                const int tt = t - G::LT, row = tt / G::UVC, xb = (tt % G::UVC) * G::UVCB;
                pfP[k] = 1; pfG[k] = (int)(row * 2 * pitchC) + xb; pfL[k] = G::UOFF + row * G::UVROWB + xb;
            } else if (t < TT) {
                int tt = t - G::LT;
                const int pl = tt >= G::CT ? 2 : 1;
                if (pl == 2) tt -= G::CT;
                const int row = tt / G::CC, xb = (tt % G::CC) * G::CCB;
                pfP[k] = pl; pfG[k] = (int)(row * pitchC) + xb; pfL[k] = (pl == 2 ? G::VOFF : G::UOFF) + row * G::CROWB + xb;
            } else { pfP[k] = -1; pfG[k] = 0; pfL[k] = 0; }
        }
    }
    // source block whose origin inside the padded luma plane is (sx, sy) (PlaneOfBlocks.cpp:1058-1079)
    __device__ __forceinline__ void pf_issue(int sx, int sy, A4x32 *pf) const {
        const unsigned offY = (unsigned)sy * pitchY + (unsigned)sx * BPS;
        const unsigned offC = (unsigned)(sy >> 1) * pitchC + (unsigned)(sx >> 1) * BPS;
#pragma unroll
        for (int k = 0; k < G::NPF; k++) {
            const int pl = pfP[k];
            if (pl < 0) continue;
            if (pl == 0) pf[k] = ld_chunk_g(srcY + offY + pfG[k], G::LCB);
            else if (uv) pf[k] = ld_chunk_g(srcUV + 2 * offC + pfG[k], G::UVCB);
            else pf[k] = ld_chunk_g((pl == 1 ? srcU : srcV) + offC + pfG[k], G::CCB);
        }
    }
    __device__ __forceinline__ void pf_store(const A4x32 *pf) const {
#pragma unroll
        for (int k = 0; k < G::NPF; k++) {
            const int pl = pfP[k];
            if (pl < 0) continue;
            if (pl == 0) st_chunk_l(lds + pfL[k], pf[k], G::LCB);
            else if (uv) st_chunk_l(lds + pfL[k], pf[k], G::UVCB);
            else st_chunk_l(lds + pfL[k], pf[k], G::CCB);
        }
    }

    // lane `l` of a 16-byte-per-lane batch register set -> scalar vector (v_readlane with a scalar lane index)
    __device__ __forceinline__ static void batch_get(const v4u &b, int l, int &x, int &y, int &sad) {
        x = __builtin_amdgcn_readlane((int)b[0], l); y = __builtin_amdgcn_readlane((int)b[1], l); sad = __builtin_amdgcn_readlane((int)b[2], l);
    }
    __device__ __forceinline__ static v4u ld_batch(GL_AS const GVec *p) { // 4-byte aligned 16-byte record
        typedef unsigned a4v __attribute__((ext_vector_type(4), aligned(4)));
        const a4v t = *(GL_AS const a4v *)p;
        return v4u{t[0], t[1], t[2], t[3]};
    }

    // GroupOfPlanes.c:69-125 + PlaneOfBlocks.cpp:971-1131 for one level
    __device__ __forceinline__ void search_level(int lvl, int globalX, int globalY, GL_AS const GVec *coarse, int coarseBlkX, int coarseBlkY, int coarseLogPel, int syncEvery) {
        const int l = lane_id();
        const ALevel &L = P.lv[lvl];
        nBlkX = uni(L.nBlkX); nBlkY = uni(L.nBlkY); pel = uni(L.pel); logPel = uni(L.logPel);
        chroma = uni(P.chroma);
        pw = uni(L.pw); ph = uni(L.ph); hpad = uni(L.hpad); vpad = uni(L.vpad);
        // (uni: the job table and the parameter block are read with vector loads -- the kernel also stores to global memory, so the
        // compiler cannot prove them invariant -- and a value that arrives in a vector register drags every address computed from
        // it into vector registers too)
        auto uptr = [](const unsigned char *p) { return (gl_u8 *)(unsigned long long)uni((long long)(unsigned long long)p); };
        srcY = uptr(J.src[0] + L.off[0]); refY = uptr(J.ref[0] + L.off[0]);
        srcU = uptr(J.src[1] + L.off[1]); refU = uptr(J.ref[1] + L.off[1]);
        srcV = uptr(J.src[2] + L.off[2]); refV = uptr(J.ref[2] + L.off[2]);
        pitchY = (unsigned)uni((int)P.pitch[0]); pitchC = (unsigned)uni((int)P.pitch[1]); pstrideY = (unsigned)uni((int)L.pstride[0]); pstrideC = (unsigned)uni((int)L.pstride[1]);
        shadowY = (unsigned)uni((int)P.shadow[0]);
        // the UV plane mirrors the whole U / V buffers sample for sample: U byte offset o <-> UV byte offset 2 * o
        srcUV = uptr(J.src[1] + P.shadow[1] + 2 * L.off[1]); refUV = uptr((J.ref[1] ? J.ref[1] : J.src[1]) + P.shadow[1] + 2 * L.off[1]);
        unsigned char *rec = (unsigned char *)(unsigned long long)uni((long long)(unsigned long long)(J.blob + L.blobOff));
        vectors = (GL_AS GVec *)(rec + 4);
        if (l == 0) *(int *)rec = 4 + nBlkX * nBlkY * 16; // pobWriteHeaderToArray :413-416
        const int nBlk = nBlkX * nBlkY;
        const bool smallestPlane = lvl == P.nLevels - 1;
        interpolate(coarse, coarseBlkX, coarseBlkY, coarseLogPel);
        // the interpolated field is re-read (by other lanes) during the scan: make it visible once per level
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_s_barrier();

        // ---- plane scan set-up (doPobSearchMVs :979-1034); tryMany is off, the search types were checked by the host
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
        LDS_AS v4u *rowbuf = (LDS_AS v4u *)(lds + ldsRow);
        pf_setup();

        // 64 blocks at a time: bSelf / bBelow = the interpolated predictors of this row's group and of the blocks "below-ahead"
        // of it (lane i <-> column 64 * grp + i of the current row; bBelow holds column + dir of the next row), bUp = the previous
        // row's results for the group (LDS), bOut = this group's results.
        // lambda of a block whose predictor SAD is predSad (:456-462), fp64 as the reference
        auto lambda_of = [&](int predSad) {
            const double scale = (double)LSAD / (double)(LSAD + (long long)(predSad >> 1));
            return (int)(long long)((double)(long long)nLambdaLevel * scale * scale);
        };
        v4u bSelf = {0, 0, 0, 0}, bBelow = {0, 0, 0, 0}, bUp = {0, 0, 0, 0}, bOut = {0, 0, 0, 0};
        int prevX = 0, prevY = 0, prevSad = 0;
        A4x32 pf[G::NPF];
        pf_issue(hpad, vpad, pf); // block (0, 0)
        int curIb = 0, curBy = 0;
#ifdef MVX_FAST_PROF
        tp = (long long)__builtin_amdgcn_s_memtime();
#endif
        for (int n = 0; n < nBlk; n++) {
            if (syncEvery && (curIb & (syncEvery - 1)) == 0) __builtin_amdgcn_s_barrier(); // keeps the chains of a workgroup on neighbouring blocks (shared reference lines)
            FPROF(0, tp);
            const int blky = curBy;
            const bool fwd = (blky & 1) == 0 || !meander;
            const int blkx = fwd ? curIb : nBlkX - 1 - curIb;
            const int dir = fwd ? 1 : -1;
            const bool rowStart = curIb == 0;
            if (++curIb == nBlkX) { curIb = 0; curBy++; }
            blkIdx = blky * nBlkX + blkx;
            x0 = hpad + stepX * blkx; y0 = vpad + stepY * blky;
            const int col = blkx & 63;
            // ---- a new group of 64 columns (or a new row): flush the finished group, fetch the next one
            if (rowStart || col == (fwd ? 0 : 63)) {
                const int c0 = blkx & ~63, c = c0 + l;
                const bool in = c < nBlkX;
                if (in) {
                    bSelf = ld_batch(&vectors[blky * nBlkX + c]);
                    // :456-462 lane-parallel: every level but the coarsest scales lambda by its block's OWN interpolated SAD, which this lane
                    // just fetched -- 64 fp64 divisions at once instead of one per block in uniform code (the 4th dword of a batch is free)
                    if (!smallestPlane) bSelf[3] = (unsigned)lambda_of((int)bSelf[2]);
                }
                const int cb = c + dir; // "below-ahead" of column c
                bBelow = v4u{0, 0, 0, 0};
                if (in && blky < nBlkY - 1 && cb >= 0 && cb < nBlkX) bBelow = ld_batch(&vectors[(blky + 1) * nBlkX + cb]);
                if (in) bUp = rowbuf[c];
            }
            // ---- source block -> LDS; request the next block's (PlaneOfBlocks.cpp:1058-1079)
            pf_store(pf);
            if (n + 1 < nBlk) {
                const int nby = curBy;
                const bool nf = (nby & 1) == 0 || !meander;
                const int nbx = nf ? curIb : nBlkX - 1 - curIb;
                pf_issue(hpad + stepX * nbx, vpad + stepY * nby, pf);
            }
            FPROF(1, tp);
            // ---- motion-vector limits (:1094-1097)
            nDxMax = (pw - x0 - BW - hpad + hps) << logPel;
            nDyMax = (ph - y0 - BW - vpad + vps) << logPel;
            nDxMin = -((x0 - hpad + hps) << logPel);
            nDyMin = -((y0 - vpad + vps) << logPel);
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
                const v4u t = rowbuf[blkx + dir];
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
            int predSad;
            if (smallestPlane) { predX = pX[0]; predY = pY[0]; predSad = s0; }
            else { predX = clipx(sfx); predY = clipy(sfy); predSad = sfs; }
            // :456-462: lambda shrinks with the predictor's SAD (row 0 searches without the motion term, :1081-1084)
            nLambda = 0;
            if (blky > 0) nLambda = smallestPlane ? uni(lambda_of(predSad)) : __builtin_amdgcn_readlane((int)bSelf[3], col);
            __builtin_amdgcn_wave_barrier(); // single wave: DS ops are in order; keeps the compiler from moving LDS reads above the staging writes
            search_block();
            __builtin_amdgcn_wave_barrier();
#ifdef MVX_FAST_PROF
            prof[11] += 1;
#endif
            // ---- result (:967, :1106): collected per group, stored when the group (or the row) ends
            { const bool mine = l == col; bOut[0] = mine ? (unsigned)bestX : bOut[0]; bOut[1] = mine ? (unsigned)bestY : bOut[1]; bOut[2] = mine ? (unsigned)bestSad : bOut[2]; }
            prevX = bestX; prevY = bestY; prevSad = bestSad;
            const bool rowEnd = curIb == 0;
            if (rowEnd || col == (fwd ? 63 : 0)) {
                const int c = (blkx & ~63) + l;
                if (c < nBlkX) {
                    typedef unsigned a4v __attribute__((ext_vector_type(4), aligned(4)));
                    const a4v t = {bOut[0], bOut[1], bOut[2], 0u}; // (block SADs are non-negative and < 2^31)
                    *(GL_AS a4v *)&vectors[blky * nBlkX + c] = t;
                    rowbuf[c] = bOut;
                }
            }
            FPROF(10, tp);
        }
        // vectors[] of this level feed the next level's interpolation / global-MV estimate (other lanes read them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_s_barrier();
    }
};

// WPE = chains per SIMD the kernel is built for (register budget), MAXCPW = 4 * WPE the largest workgroup it may be launched with;
// the workgroup's chains (blockDim.x / 64 of them) are consecutive entries of the job table, which the host orders so that the
// chains of a workgroup search the same reference frame(s) (shared lines in the CU's L1 and the XCD's L2).
// flags: bit 0 = deal the workgroups out so that consecutive ones (which share reference frames) run on the same XCD.
#define MVX_FAST_XCD_REMAP 1
#define MVX_FAST_UV 2 // host -> launcher: the jobs' frames carry the UV-interleaved shadow plane
template <int BPS, int BW, int WPE, int MAXCPW, bool UV>
__global__ __launch_bounds__(64 * MAXCPW, WPE) void analyse_fast_kernel(const AParams *Pp, const AJob *jobs, int njobs, int ldsChain, int syncEvery, int ldsRow, int ldsHist, int histBins, int flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AParams &P = *Pp;
    const int cpw = (int)(blockDim.x >> 6);
    int wg = (int)blockIdx.x;
    if (flags & MVX_FAST_XCD_REMAP) { // workgroup b runs on XCD b % 8 (round-robin dispatch): give every XCD a contiguous range of the table
        const int n = (int)gridDim.x, x = wg & 7, slot = wg >> 3;
        wg = x * (n >> 3) + min(x, n & 7) + slot;
    }
    const int chain = uni(wg * cpw + (int)(threadIdx.x >> 6));
    if (chain >= njobs) return; // (a finished wave no longer counts for the workgroup's barriers)
    const AJob &J = jobs[chain];
    if (!J.blob) return;        // padding entry of the job table (the host keeps the chains of one reference frame in one workgroup)
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
    FastSearcher<BPS, BW, UV> S(P, J);
    S.lds = (lds_u8 *)smem + uni((int)(threadIdx.x >> 6)) * ldsChain;
    S.ldsRow = ldsRow; S.ldsHist = ldsHist; S.histBins = histBins;
#ifdef MVX_FAST_PROF
    for (int i = 0; i < FPROF_N; i++) S.prof[i] = 0;
#endif
    int gx = 0, gy = 0; // zeroMV, MVAnalysisData.h:79
    GL_AS const GVec *coarse = nullptr;
    int cbx = 0, cby = 0, clp = 0;
    for (int lvl = P.nLevels - 1; lvl >= 0; lvl--) {
        if (coarse && P.global) S.estimate_global(coarse, cbx * cby, 8192 * P.lv[lvl + 1].pel, &gx, &gy);
        S.search_level(lvl, gx, gy, coarse, cbx, cby, clp, cpw > 1 ? syncEvery : 0);
        coarse = S.vectors; cbx = P.lv[lvl].nBlkX; cby = P.lv[lvl].nBlkY; clp = P.lv[lvl].logPel;
    }
#ifdef MVX_FAST_PROF
    if (l == 0 && chain == 5) for (int i = 0; i < FPROF_N; i++) g_fastprof[i] = (unsigned long long)S.prof[i];
#endif
}
#if defined(MVX_FAST_PROF) && defined(MVX_PROF_EXPORT)
extern "C" __attribute__((visibility("default"))) int mvx_debug_fastprof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fastprof), sizeof(unsigned long long) * FPROF_N) == hipSuccess ? 0 : -1;
}
#endif

template <int BPS, int BW, int WPE, int MAXCPW, bool UV> static int launch_analyse_fast_uv(const ALaunch &L) {
    const int perChain = (L.ldsNeed + 255) & ~255;
    const int cpw = L.cpw < MAXCPW ? L.cpw : MAXCPW;
    int lds = perChain * cpw;
    if (L.ldsBytes > lds && L.ldsBytes <= 160 * 1024) lds = L.ldsBytes; // developer / host option: fewer workgroups per CU
    if (lds > 64 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void *)analyse_fast_kernel<BPS, BW, WPE, MAXCPW, UV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((analyse_fast_kernel<BPS, BW, WPE, MAXCPW, UV>), dim3((L.njobs + cpw - 1) / cpw), dim3(64 * cpw), lds, L.st, L.dP, L.dJobs,
                       L.njobs, perChain, L.syncEvery, L.ldsRow, L.ldsHist, L.histBins, L.flags);
    return MVX_OK;
}
template <int BPS, int BW, int WPE, int MAXCPW> static int launch_analyse_fast(const ALaunch &L) {
    if (L.flags & MVX_FAST_UV) return launch_analyse_fast_uv<BPS, BW, WPE, MAXCPW, true>(L);
    return launch_analyse_fast_uv<BPS, BW, WPE, MAXCPW, false>(L);
}

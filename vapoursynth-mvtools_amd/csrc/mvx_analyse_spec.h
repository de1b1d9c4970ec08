// mvx_analyse_spec.h -- the default search of mv.Analyse with the SAD work taken OFF the serial chain (round 4).
//
// The reference walks the blocks of a plane one after the other: block i takes its left predictor (and through it the median)
// from block i-1 (PlaneOfBlocks.cpp:421-440), so a chain of blocks is strictly serial, and the lean kernel (mvx_analyse_fast.h)
// pays two memory round trips plus ~700 dependent instructions per block (profiles/r4_fastprof_phase_cycles.txt: a block costs
// ~12 000 cycles, half of them in the scalar phases around the two candidate passes).  But almost everything a block evaluates does
// NOT depend on its left neighbour: the zero, global, hierarchical, up and ahead predictors (:834-915), every cost term (lambda and
// the cost centre come from the interpolated predictor, :449-462) and -- when the predictor phase ends on the up neighbour's vector,
// which it does whenever the motion field is locally smooth -- the whole refinement pattern around it (:667-724, :786-791).
// So a chain is processed in GROUPS of 32 blocks of a row:
//
//   A   for every block of the group, with all 64 lanes and nothing to wait for but memory: the SADs of the left-independent
//       predictors and of the refinement pattern around the block's UP predictor, written to a table in LDS;
//   A2  one LANE per block: costs, penalties, the predictor phase and the refinement in the reference's order (:219-261, strict <)
//       UNDER THE HYPOTHESIS "the left neighbour's result, clipped, equals my up predictor" (then left == median == up), giving
//       a speculative result per block and a flag "complete" (centre was up, no hexagon point won, no bad-block rescue);
//   B   verification, in walk order: a block whose hypothesis holds (its true left result is known by then) and whose flag is set
//       takes its speculative result -- whole runs of them at once, from a lane mask; any other block is searched live by
//       FastSearcher::search_block with its true predictors, exactly as the lean kernel does, and the walk goes on behind it.
//
// Results are identical by construction: a verified block evaluated the same candidates with the same costs in the same order as
// the serial walk; everything else IS the serial walk.  tools/search_stats.py (CPU, oracle counters) gives the share of blocks
// that verify: 99.3 % of the finest level on the bench clip, 88.9 % on the noisy test clip (profiles/r4_search_stats_*.txt).
// The first block row of a plane, the first block of every row and the coarsest level (whose cost centre is the median, :858)
// are always searched live.
#pragma once
#include <type_traits>
#include "mvx_analyse_fast.h"

#define SPEC_TB 32                        // blocks per group (one table column per block)
#define SPEC_STRIDE (SPEC_TB * 8 + 8)     // bytes between the slots of the table: 66 dwords, so that the 16 (32) group leaders of a pass write different banks
// slots: Hex2 levels: 0-5 hexagon, 6-13 square, 14 up, 15 ahead, 16 zero, 17 global, 18 hierarchical; exhaustive levels: 0-23 rings 1 and 2, 24 up, 25 ahead, 26 zero, 27 global, 28 hierarchical
#define SPEC_SLOTS_HEX 19
#define SPEC_SLOTS_EXH 29
#define MVX_FAST_NOSTRIP 8                // flags: no runs in pass A (every block's candidates loaded on their own)
#define MVX_FAST_NOSPEC 4                 // flags: verify nothing, search every block live (developer switch: the same kernel as a plain serial walk)

// -DMVX_SPEC_ABL=n (tools/build_variant.py): timing-only ablations of pass A, results are WRONG -- 1: no zero / global / hierarchical pass on Hex2
// levels, 2: every group of the pattern pass evaluates the centre (same lines for all lanes), 3: every speculative result is taken (no live blocks); row passes: 4: no LDS source reads, 5: no reference loads, 6: no SADs
#ifndef MVX_SPEC_ABL
#define MVX_SPEC_ABL 0
#endif
#ifndef MVX_STRIP_DMA
#define MVX_STRIP_DMA 1 // r6: the source strip of a window of 16-bit 16x16 blocks goes from global memory straight into LDS (global_load_lds_dwordx4), two buffers
#endif
#ifndef MVX_SPEC_SW3
#define MVX_SPEC_SW3 12 // row loads in flight per lane in the builds for three or more chains per SIMD (168 registers)
#endif
// -DMVX_SPEC_PROF (tools/specprof.py): cycles of ONE chain per phase of the group loop (s_memtime; a stamp waits for the scalar counter only)
#ifdef MVX_SPEC_PROF
#define SPROF_N 20 // (16: TEAM, waiting for the previous block row; 17: TEAM, waiting for the token)
static __device__ unsigned long long g_specprof[SPROF_N];
#define SPROF(i) do { const long long t1_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(t1_) : "memory"); sprof[i] += t1_ - sprofT; sprofT = t1_; } while (0)
#else
#define SPROF(i) ((void)0)
#endif
#ifdef MVX_SPEC_PROF
#define SPEC_PROF_DUMP_() do { if (l == 0 && chain == 5 && S.role == 1) for (int i = 0; i < SPROF_N; i++) g_specprof[i] = (unsigned long long)S.sprof[i]; } while (0)
#else
#define SPEC_PROF_DUMP_() ((void)0)
#endif
#if MVX_SPEC_ABL == 9 // debug build: what the kernel knew about the 64 columns at (level, block row, first column) = g_specdbgAt (mvx_debug_specdbg_at)
#define SPECDBG_N 24
static __device__ int g_specdbg[64 * SPECDBG_N];
static __device__ int g_specdbgAt[3];
#define SPECDBG_HERE() (lvl == g_specdbgAt[0] && blky == g_specdbgAt[1] && c0 == g_specdbgAt[2])
#endif
#ifdef MVX_SPEC_STATS
static __device__ unsigned long long g_specstat[MVX_MAX_LEVELS][8]; // per level: blocks in speculated rows, of them searched live, live because the flag was clear, rescues; 16x16 row passes: windows of stage 2 in strip form, in block form, blocks of block-form windows whose centre differs from the window's first block, block-form windows because of the limits alone
#endif

// SWIN: row loads a lane keeps in flight in the row passes (12 = half a pass; 24 = a whole pass: the builds with 256 registers)
// (pass A has nothing else to wait for: whole candidates as one stream of loads -- FastSearcher's STREAM_MAX = 48; a template argument, so the serial
// kernel's FastSearcher<.., 12> and this one are different types whatever translation unit instantiates them)
//
// TEAM (round 5): ONE chain is walked by the nw waves of a workgroup.  The groups of a level are dealt out round robin; a wave runs phases A1 / A / A2 of
// its group with nothing to wait for but the previous block row's results of the same columns, then waits for the TOKEN -- the count of groups whose
// results are written -- to reach its group, takes the true left neighbour (prevX / prevY / prevSad) and badcount from the workgroup's control words,
// runs phase B exactly as the single-wave form does, writes the group's results and passes the token on.  The serial part of the walk (B) stays serial
// and in reference order; the SAD work of nw groups overlaps.  A chain finishes ~nw times sooner, so a launch keeps nw times fewer chains resident for
// the same number of waves: their live reference rows share the L2 / the Infinity Cache among fewer chains, and small launches (a frame server's
// look-ahead window) fill the GPU.  Results are those of the single-wave form by construction: every decision in A2 / B is taken with the same inputs.
// SIDE (r6; a run-time flag of the level in r5, which cost the overlapped builds registers and 3 % of their launch time): the build for 16x16 blocks SIDE BY SIDE
// (overlap 0): a block is two columns and steps by two, a window holds four
template <int BPS, int BW, bool UV, int SWIN = 12, bool TEAM = false, bool SIDE = false> struct SpecSearcher : FastSearcher<BPS, BW, UV, 48> {
    typedef FastSearcher<BPS, BW, UV, 48> F;
    typedef FGeo<BPS, BW> G;
    using F::P; using F::J; using F::lds; using F::ldsRow; using F::ldsHist; using F::histBins;
    using F::nBlkX; using F::nBlkY; using F::pel; using F::logPel; using F::pw; using F::ph; using F::hpad; using F::vpad;
    using F::srcY; using F::srcU; using F::srcV; using F::refY; using F::refU; using F::refV; using F::srcUV; using F::refUV;
    using F::pitchY; using F::pitchC; using F::pstrideY; using F::pstrideC; using F::shadowY; using F::vectors;
    using F::chroma; using F::searchType; using F::nSearchParam; using F::penaltyNew; using F::penaltyZero; using F::pglobal; using F::badrange; using F::badcount; using F::fieldShift;
    using F::badSAD; using F::LSAD; using F::gmvx; using F::gmvy;
    using F::x0; using F::y0; using F::blkIdx; using F::nDxMin; using F::nDyMin; using F::nDxMax; using F::nDyMax;
    using F::predX; using F::predY; using F::pX; using F::pY; using F::nLambda; using F::bestX; using F::bestY; using F::bestSad;
    int ldsTab; // byte offset of the SAD table inside the chain's LDS
    // TEAM: this wave's number inside the workgroup and the workgroup's waves; the previous block row's results (shared); control words: [0] token = groups
    // of this level whose results are written, [1..3] the last block's result (x, y, sad), [4] badcount
    int role, nw;
    lds_u8 *shRow;
    LDS_AS int *ctl;
    // the running global predictor after the blocks lo..hiE-1 of the 64-column chunk at c0, in walk order (:859: every block clips the running value
    // with its own limits).  The limits are monotonic along a row: a value that neither the first nor the last block of the group clips passes them all
    __device__ __forceinline__ int team_advance(int g, int c0, int lo, int hiE, bool fwd, int stepX, int hps) const {
        auto clampAt = [&](int v, int li) { const int xb = stepX * (c0 + li); return min(max(v, -((xb + hps) << logPel)), ((pw - xb - hpad - BW - hpad + hps) << logPel) - 1); };
        if (clampAt(g, lo) == g && clampAt(g, hiE - 1) == g) return g;
        for (int i = 0; i < hiE - lo; i++) g = clampAt(g, fwd ? lo + i : hiE - 1 - i);
        return g;
    }
    // The token protocol rests on two properties of the launch (r6): every wave of the workgroup is RESIDENT while another one spins (a workgroup is
    // dispatched to one CU as a whole and is never pre-empted wave by wave: the spin cannot starve the wave it waits for), and the waves share the CU's L1
    // (the library is built without -mtgsplit: in threadgroup-split mode the waves of a workgroup may sit on different CUs, and global `vectors[]` written by one
    // wave would need an agent-scope release to reach another).  -DMVX_TEAM_WATCHDOG (developer builds): a wait that lasts ~2^26 sleeps traps instead of hanging,
    // so that an ordering bug shows up as a launch error.
    __device__ __forceinline__ void team_wait_ge(int need) const { // until the results of the first `need` groups of this level are written
#ifdef MVX_TEAM_WATCHDOG
        unsigned spins = 0;
#endif
        while (uni(__hip_atomic_load(ctl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
            __builtin_amdgcn_s_sleep(2);
#ifdef MVX_TEAM_WATCHDOG
            if (++spins > (1u << 26)) __builtin_trap();
#endif
        }
    }
    __device__ __forceinline__ void team_acquire(int myG, int &px, int &py, int &ps) { // the token reaches group myG: the walk's state behind group myG - 1
        team_wait_ge(myG);
        px = uni(__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        py = uni(__hip_atomic_load(ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        ps = uni(__hip_atomic_load(ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        badcount = uni(__hip_atomic_load(ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    __device__ __forceinline__ void team_release(int done, int px, int py, int ps) const { // this group's results (vectors[], the row buffer) are written
        if (lane_id() == 0) {
            __hip_atomic_store(ctl + 1, px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(ctl + 2, py, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(ctl + 3, ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(ctl + 4, badcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane_id() == 0) __hip_atomic_store(ctl, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef MVX_SPEC_PROF
    long long sprof[SPROF_N], sprofT; // 0 barrier, 1 fetch + A1, 2 row passes, 3 one-block passes, 4 A2, 5 verification, 6 live blocks, 7 results, 8 level prologue, 9 groups, 10 live blocks counted
#endif

    __device__ SpecSearcher(const AParams &p, const AJob &j) : F(p, j) {}

    __device__ __forceinline__ static int pk(int x, int y) { return (int)(((unsigned)x & 0xffffu) | ((unsigned)y << 16)); } // |x|, |y| < 30000: mvx_fast_eligible
    __device__ __forceinline__ static int upx(int c) { return (int)(short)(c & 0xffff); }
    __device__ __forceinline__ static int upy(int c) { return c >> 16; }

    // ---- pass A as STREAMS of loads (r4, second form).  One candidate = NA luma pieces + NB pieces of the UV plane per lane of its group
    // (1 << LOGG lanes).  A stream keeps a window of W loads in flight ACROSS blocks: when piece k of block b has been consumed its
    // register is refilled with piece k + W -- of block b while it has that many, of block b + 1 after that.  A wave therefore never
    // drains its loads between passes or blocks (pass A, first form: three exposed round trips per block, 895 ms per launch,
    // profiles/r4_spec_v1_*.txt).
    template <int LOGG, int WMAX> struct PG {
        static constexpr int GG = 1 << LOGG, CA = 1 << G::LLOGC, CB = 1 << G::UVLOGC;
        static constexpr bool OK = UV && G::LT % GG == 0 && G::UVT % GG == 0 && G::LT >= GG && G::UVT >= GG && GG >= CA && GG >= CB;
        static constexpr int NA = OK ? G::LT / GG : 1, NB = OK ? G::UVT / GG : 1, NT = NA + NB;
        static constexpr int pickW(int w) { return w <= 1 ? 1 : (NT % w == 0 ? w : pickW(w - 1)); }
        static constexpr int W = pickW(NT < WMAX ? NT : WMAX); // the largest divisor of NT that fits the budget
    };
    template <int LOGG, int WMAX> struct Pass {
        typedef PG<LOGG, WMAX> Q;
        v4u r[Q::W];
        unsigned curA, curB; // byte offsets (luma plane set / UV plane set) of the next piece to request
        unsigned aL, aC;
    };
    static constexpr int LOGGP = PG<4, 4>::OK ? 4 : PG<3, 4>::OK ? 3 : PG<2, 4>::OK ? 2 : 1; // lanes per candidate of the zero / global / hierarchical pass
    static constexpr bool STREAM_HEX = PG<2, 12>::OK && PG<LOGGP, 4>::OK, STREAM_EXH = PG<1, 12>::OK;

    __device__ __forceinline__ unsigned luma_off_at(int bx0, int vx, int vy) const { // ref_luma_off for a block at bx0 of this row
        const int ax = (bx0 << logPel) + vx, ay = (y0 << logPel) + vy, m = pel - 1;
        const unsigned idx = (unsigned)((ax & m) | ((ay & m) << logPel));
        return F::shadow_off(idx * pstrideY + (unsigned)(ay >> logPel) * pitchY + (unsigned)(ax >> logPel) * BPS, shadowY);
    }
    __device__ __forceinline__ unsigned chroma_off_at(int bx0, int vx, int vy) const {
        const int xb = vx < 0 ? 1 : 0, yb = vy < 0 ? 1 : 0;
        const int ax = ((bx0 >> 1) << logPel) + ((vx + xb) >> 1), ay = ((y0 >> 1) << logPel) + ((vy + yb) >> 1), m = pel - 1;
        const unsigned idx = (unsigned)((ax & m) | ((ay & m) << logPel));
        return idx * pstrideC + (unsigned)(ay >> logPel) * pitchC + (unsigned)(ax >> logPel) * BPS;
    }
    // this lane's first luma / UV piece of candidate (vx, vy; chroma rows from vyc) of the block at bx0
    template <int LOGG, int WMAX> __device__ __forceinline__ void pass_start(int s, int bx0, int vx, int vy, int vyc, unsigned &oA, unsigned &oB) const {
        typedef PG<LOGG, WMAX> Q;
        const int rowA = s >> G::LLOGC, xbA = (s & (Q::CA - 1)) * G::LCB, rowB = s >> G::UVLOGC, xbB = (s & (Q::CB - 1)) * G::UVCB;
        oA = luma_off_at(bx0, vx, vy) + (unsigned)rowA * pitchY + (unsigned)xbA;
        oB = 2 * chroma_off_at(bx0, vx, vyc) + (unsigned)rowB * 2 * pitchC + (unsigned)xbB;
    }
    template <int LOGG, int WMAX> __device__ __forceinline__ v4u pass_issue(Pass<LOGG, WMAX> &T, int piece) const {
        typedef PG<LOGG, WMAX> Q;
        v4u v;
        if (piece < Q::NA) { v = F::template ld_ref<G::LCB>(refY + T.curA); T.curA += (unsigned)(Q::GG >> G::LLOGC) * pitchY; asm volatile("" : "+v"(T.curA) : : "memory"); }
        else { v = F::template ld_ref<G::UVCB>(refUV + T.curB); T.curB += (unsigned)(Q::GG >> G::UVLOGC) * 2 * pitchC; asm volatile("" : "+v"(T.curB) : : "memory"); }
        return v;
    }
    template <int LOGG, int WMAX> __device__ __forceinline__ void pass_prime(Pass<LOGG, WMAX> &T, unsigned oA, unsigned oB) const {
        typedef PG<LOGG, WMAX> Q;
        T.curA = oA; T.curB = oB;
#pragma unroll
        for (int k = 0; k < Q::W; k++) T.r[k] = pass_issue(T, k);
    }
    // consume the block whose first W pieces are in flight (SADs into T.aL / T.aC), refilling the window; (nA, nB) = start of the NEXT block's candidate
    template <int LOGG, int WMAX, bool REFILL = true> __device__ __forceinline__ void pass_run(Pass<LOGG, WMAX> &T, int s, unsigned nA, unsigned nB) const {
        typedef PG<LOGG, WMAX> Q;
        const int rowA = s >> G::LLOGC, xbA = (s & (Q::CA - 1)) * G::LCB, rowB = s >> G::UVLOGC, xbB = (s & (Q::CB - 1)) * G::UVCB;
        const lds_u8 *spA = lds + rowA * G::LROWB + xbA, *spB = lds + G::UOFF + rowB * G::UVROWB + xbB;
        constexpr int lstepA = (Q::GG >> G::LLOGC) * G::LROWB, lstepB = (Q::GG >> G::UVLOGC) * G::UVROWB;
        auto src_piece = [&](int k) { return k < Q::NA ? F::template lds_piece<G::LCB>(spA + k * lstepA) : F::template lds_piece<G::UVCB>(spB + (k - Q::NA) * lstepB); };
        constexpr int D = MVX_SRC_AHEAD < Q::NT ? MVX_SRC_AHEAD : Q::NT;
        v4u a[D];
#pragma unroll
        for (int k = 0; k < D; k++) a[k] = src_piece(k);
        unsigned aL = 0, aC = 0;
#pragma unroll
        for (int k = 0; k < Q::NT; k++) {
            const v4u cur = a[k % D];
            if (k + D < Q::NT) a[k % D] = src_piece(k + D);
            if (k < Q::NA) aL = F::template sad_regs<G::LCB>(cur, T.r[k % Q::W], aL);
            else aC = F::template sad_regs<G::UVCB>(cur, T.r[k % Q::W], aC);
            // (the refill must FOLLOW the SADs of the register it overwrites: left alone the scheduler renames the register, requests the whole next
            // window and reads every source piece first -- 150 live registers -- instead of one load per consumed piece)
            asm volatile("" : "+v"(aL), "+v"(aC) : : "memory");
            const int kk = k + Q::W;
            if (kk == Q::NT) { T.curA = nA; T.curB = nB; } // the window moves on to the next block
            if (REFILL || kk < Q::NT) T.r[k % Q::W] = pass_issue(T, kk % Q::NT); // (REFILL = false: the last block of a list)
        }
        T.aL = aL; T.aC = aC;
    }

    // ---- pass A for RUNS of blocks with one displacement (r4, third form).  Blocks overlap by half, so a run of L blocks with the same
    // candidate vector reads ONE contiguous strip of L + 1 sixteen-byte columns per reference row, and the SAD of block m is the sum of the
    // column sums m and m + 1 -- every column sum is shared by the two blocks that cover it.  Eight lanes per candidate (lane p = column p,
    // up to seven blocks), eight candidates per pass, 24 rows (16 luma + 8 of the UV plane) per lane and pass.  Against one block at a time:
    // 128 contiguous bytes per candidate row instead of 7 x 32 scattered ones (~2.3x fewer L1 misses, 3.5x fewer look-ups), half the SADs.
    // The source blocks of the run are one strip in LDS (3 KB: 24 rows x 8 columns).
    // (the row-pass code below is written for 16x16 AND 32x32 blocks -- HC columns per half block -- but only 16x16 is enabled: at 32x32 it was no faster
    // than the serial kernel on cfg5 (90.1 against 92.3 fps: windows of three blocks share less, six staging pieces and 24 loads in flight do not fit the
    // registers) and one 8K bench clip disagreed with the oracle, which was not chased: profiles/r4_cfg5_rowpasses.txt)
#ifndef MVX_STRIP32
#define MVX_STRIP32 0 // developer builds (-DMVX_STRIP32=1, mvx_analyse.hip AND mvx_analyse_spec_u16.hip): row passes for 32x32 blocks too
#endif
    // (r5: 8-bit 16x16 blocks overlapping by 8 -- the common HD setting -- take the same code with 8-byte columns: a half block is 8 samples = COLB bytes, a window row
    // 8 * COLB = 64 bytes; loads are 8 bytes per lane, at any byte address -- 8-bit super frames have no shifted copies)
    // (r5: the builds WITHOUT the UV plane run the same passes for luma-only searches -- chroma = 0: 16 rows per pass, no UV rows; with chroma on they have no row passes)
    static constexpr bool STRIP_OK = (BPS == 2 && (BW == 16 || (MVX_STRIP32 && BW == 32))) || (BPS == 1 && BW == 16);
    static constexpr int COLB = BPS == 2 ? 16 : 8, ROWB = 8 * COLB; // bytes of a strip column / of a strip row (eight columns)
    // HC = 16-byte columns per half block (a 32x32 block row is four columns, blocks step by two); a window is eight columns: 7 (3) blocks;
    // block form: LPB lanes per block, LPC per candidate (four candidates: lanes 0..4 * LPC - 1; lanes 56-63 stay free for the zero vector's strip)
    static constexpr int HC = STRIP_OK ? BW / 16 : 1, SW_BLOCKS = 8 / HC - 1, LPB = 2 * HC, LPC = SW_BLOCKS * LPB;
    static_assert(!SIDE || (STRIP_OK && HC == 1), "blocks side by side: 16x16 row passes only");
    static constexpr int pickSW(int nt, int w) { return w <= 1 ? 1 : (w <= nt && nt % w == 0) ? w : pickSW(nt, w - 1); } // rows in flight: a divisor of the rows of a pass
    static constexpr int SNA = BW, SNB = UV ? BW / 2 : 0, SNT = SNA + SNB, SW = pickSW(SNT, (BW == 32 && SWIN > 12) ? 12 : SWIN), S_UV = SNA * ROWB, SSTG = SNT / 8; // (32x32: 12 in flight -- 24 plus the six staging pieces spill) // rows of a pass, loads in flight, LDS offset of the UV rows, staging pieces per lane
    struct StripPass { v4u r[SW]; unsigned curA, curB, aL, aC; };
    __device__ __forceinline__ v4u strip_issue(StripPass &T, int piece) const {
        v4u v;
        if (piece < SNA) { v = F::template ld_ref<COLB>(refY + T.curA); T.curA += pitchY; asm volatile("" : "+v"(T.curA) : : "memory"); }
        else { v = F::template ld_ref<COLB>(refUV + T.curB); T.curB += 2 * pitchC; asm volatile("" : "+v"(T.curB) : : "memory"); }
        return v;
    }
    __device__ __forceinline__ void strip_prime(StripPass &T, unsigned oA, unsigned oB) const {
        T.curA = oA; T.curB = oB;
#pragma unroll
        for (int k = 0; k < SW; k++) T.r[k] = strip_issue(T, k);
    }
    template <bool REFILL> __device__ __forceinline__ void strip_run(StripPass &T, int p, unsigned nA, unsigned nB, int bufOff = 0) const {
        const lds_u8 *sp = lds + bufOff + p * COLB;
        auto src_piece = [&](int k) { return F::template lds_piece<COLB>(sp + (k < SNA ? k * ROWB : S_UV + (k - SNA) * ROWB)); };
        constexpr int D = MVX_SRC_AHEAD;
        v4u a[D];
#pragma unroll
        for (int k = 0; k < D; k++) a[k] = src_piece(k);
        unsigned aL = 0, aC = 0;
#pragma unroll
        for (int k = 0; k < SNT; k++) {
            const v4u cur = a[k % D];
            if (MVX_SPEC_ABL != 4 && k + D < SNT) a[k % D] = src_piece(k + D);
            if (MVX_SPEC_ABL != 6) {
                if (k < SNA) aL = F::template sad_regs<COLB>(cur, T.r[k % SW], aL);
                else aC = F::template sad_regs<COLB>(cur, T.r[k % SW], aC);
            } else { aL += cur[0] + T.r[k % SW][0]; }
            asm volatile("" : "+v"(aL), "+v"(aC) : : "memory");
            const int kk = k + SW;
            if (kk == SNT) { T.curA = nA; T.curB = nB; }
            if (MVX_SPEC_ABL != 5 && (REFILL || kk < SNT)) T.r[k % SW] = strip_issue(T, kk % SNT);
        }
        T.aL = aL; T.aC = aC;
    }
    // ---- row passes for 8-bit clips, 8x8 blocks overlapping by half (r4).  A half block is ONE dword: a lane's 16-byte load covers four block steps,
    // and the SAD of block m is the sum of the dword sums m and m + 1.  A window is 15 blocks = 16 dwords = FOUR lanes per candidate in strip form
    // (16 candidates per pass: the whole Hex2 pattern in one); in block form a lane is one (candidate, block) and uses the first two dwords of its
    // load.  12 rows per pass (8 luma + 4 of the UV plane), all in flight; the window's source strip is 12 rows x 64 B in LDS.
    #ifndef MVX_NO_STRIP8
    static constexpr bool STRIP8_OK = UV && BPS == 1 && BW == 8;
#else
    static constexpr bool STRIP8_OK = false;
#endif
    static constexpr int W8_BLOCKS = 15, R8A = 8, R8B = 4, R8T = R8A + R8B;
    struct Strip8 { v4u r[R8T]; unsigned curA, curB; unsigned aL[4], aC[4]; };
    __device__ __forceinline__ v4u strip8_issue(Strip8 &T, int piece) const { // (byte-aligned 16-byte loads: 8-bit samples sit anywhere)
        v4u v;
        if (piece < R8A) { v = F::template ld_ref<16>(refY + T.curA); T.curA += pitchY; asm volatile("" : "+v"(T.curA) : : "memory"); }
        else { v = F::template ld_ref<16>(refUV + T.curB); T.curB += 2 * pitchC; asm volatile("" : "+v"(T.curB) : : "memory"); }
        return v;
    }
    __device__ __forceinline__ void strip8_prime(Strip8 &T, unsigned oA, unsigned oB) const {
        T.curA = oA; T.curB = oB;
#pragma unroll
        for (int k = 0; k < R8T; k++) T.r[k] = strip8_issue(T, k);
    }
    // A lane never reads past the sixteen dwords of its window / the two of its block when that would cross the end of the row (the last row of a plane
    // may end the caller's buffer): it starts its sixteen bytes sh dwords EARLIER instead (reference and source alike) and rotates its four sums back.
    // srcA / srcB: byte offsets of this lane's four source dwords inside a 64-byte strip row (luma rows / UV rows), shifts included
    static constexpr int S8_BASE = 16; // (a shifted first block reads up to 8 bytes in front of its strip row)
    template <bool REFILL> __device__ __forceinline__ void strip8_run(Strip8 &T, int srcA, int srcB, int shA, int shB, unsigned nA, unsigned nB) const {
        const lds_u8 *spA = lds + S8_BASE + srcA, *spB = lds + S8_BASE + srcB;
        auto src_piece = [&](int k) { // four dwords at a dword-aligned LDS address: two ds_read2_b32
            const LDS_AS unsigned *q = (const LDS_AS unsigned *)((k < R8A ? spA : spB) + k * 64);
            return v4u{q[0], q[1], q[2], q[3]};
        };
        unsigned aL0 = 0, aL1 = 0, aL2 = 0, aL3 = 0, aC0 = 0, aC1 = 0, aC2 = 0, aC3 = 0;
        v4u a = src_piece(0);
#pragma unroll
        for (int k = 0; k < R8T; k++) {
            const v4u cur = a;
            if (k + 1 < R8T) a = src_piece(k + 1);
            const v4u rr = T.r[k];
            if (k < R8A) { aL0 = __builtin_amdgcn_sad_u8(cur[0], rr[0], aL0); aL1 = __builtin_amdgcn_sad_u8(cur[1], rr[1], aL1); aL2 = __builtin_amdgcn_sad_u8(cur[2], rr[2], aL2); aL3 = __builtin_amdgcn_sad_u8(cur[3], rr[3], aL3); }
            else { aC0 = __builtin_amdgcn_sad_u8(cur[0], rr[0], aC0); aC1 = __builtin_amdgcn_sad_u8(cur[1], rr[1], aC1); aC2 = __builtin_amdgcn_sad_u8(cur[2], rr[2], aC2); aC3 = __builtin_amdgcn_sad_u8(cur[3], rr[3], aC3); }
            asm volatile("" : "+v"(aL0), "+v"(aC0) : : "memory");
            if (k == 0) { T.curA = nA; T.curB = nB; } // (a whole pass is in flight: every refill belongs to the next pass)
            if (REFILL) T.r[k] = strip8_issue(T, k);
        }
        auto rot = [](unsigned &a0, unsigned &a1, unsigned &a2, unsigned &a3, int sh) { // a'[j] = a[(j + sh) & 3]
            const bool o = sh & 1, t = sh & 2;
            const unsigned b0 = o ? a1 : a0, b1 = o ? a2 : a1, b2 = o ? a3 : a2, b3 = o ? a0 : a3;
            a0 = t ? b2 : b0; a1 = t ? b3 : b1; a2 = t ? b0 : b2; a3 = t ? b1 : b3;
        };
        rot(aL0, aL1, aL2, aL3, shA); rot(aC0, aC1, aC2, aC3, shB);
        T.aL[0] = aL0; T.aL[1] = aL1; T.aL[2] = aL2; T.aL[3] = aL3; T.aC[0] = aC0; T.aC[1] = aC1; T.aC[2] = aC2; T.aC[3] = aC3;
    }
    // offsets (dx, dy) of pattern point idx in the reference's order: Hex2 levels 0-5 hexagon (:682-687), 6-13 square (:636-658); exhaustive levels rings 1 and 2 (:786-791)
    __device__ __forceinline__ static void pat_delta(bool hex, int idx, int &dx, int &dy) {
        dx = 0; dy = 0;
        if (hex) {
            if (idx < 6) { dx = tab8(HEX2X >> 8, idx & 7); dy = tab8(HEX2Y >> 8, idx & 7); }
            else if (idx < 14) { dx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), idx - 6); dy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), idx - 6); }
        } else {
            if (idx < 8) { dx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), idx); dy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), idx); }
            else if (idx < 16) { dx = tab8(PACK8(-1, -1, 0, 0, 1, 1, -2, 2), idx - 8); dy = tab8(PACK8(-2, 2, -2, 2, -2, 2, -1, -1), idx - 8); }
            else if (idx < 24) { dx = tab8(PACK8(-2, 2, -2, 2, -2, -2, 2, 2), idx - 16); dy = tab8(PACK8(0, 0, 1, 1, -2, 2, -2, 2), idx - 16); }
        }
    }

    // GroupOfPlanes.c:69-125 + PlaneOfBlocks.cpp:971-1131 for one level, in groups of SPEC_TB blocks
    __device__ __forceinline__ void search_level_spec(int lvl, int globalX, int globalY, GL_AS const GVec *coarse, int coarseBlkX, int coarseBlkY, int coarseLogPel, int syncEvery, bool specEnabled, bool stripEnabled) {
        const int l = lane_id();
        const ALevel &L = P.lv[lvl];
        nBlkX = uni(L.nBlkX); nBlkY = uni(L.nBlkY); pel = uni(L.pel); logPel = uni(L.logPel);
        chroma = uni(P.chroma);
        pw = uni(L.pw); ph = uni(L.ph); hpad = uni(L.hpad); vpad = uni(L.vpad);
        auto uptr = [](const unsigned char *p) { return (gl_u8 *)(unsigned long long)uni((long long)(unsigned long long)p); };
        srcY = uptr(J.src[0] + L.off[0]); refY = uptr(J.ref[0] + L.off[0]);
        srcU = uptr(J.src[1] + L.off[1]); refU = uptr(J.ref[1] + L.off[1]);
        srcV = uptr(J.src[2] + L.off[2]); refV = uptr(J.ref[2] + L.off[2]);
        pitchY = (unsigned)uni((int)P.pitch[0]); pitchC = (unsigned)uni((int)P.pitch[1]); pstrideY = (unsigned)uni((int)L.pstride[0]); pstrideC = (unsigned)uni((int)L.pstride[1]);
        shadowY = (unsigned)uni((int)P.shadow[0]);
        srcUV = uptr(J.src[1] + P.shadow[1] + 2 * L.off[1]); refUV = uptr((J.ref[1] ? J.ref[1] : J.src[1]) + P.shadow[1] + 2 * L.off[1]);
        unsigned char *rec = (unsigned char *)(unsigned long long)uni((long long)(unsigned long long)(J.blob + L.blobOff));
        vectors = (GL_AS GVec *)(rec + 4);
        if (l == 0 && (!TEAM || role == 0)) *(int *)rec = 4 + nBlkX * nBlkY * 16; // pobWriteHeaderToArray :413-416
        const bool smallestPlane = lvl == P.nLevels - 1;
        if constexpr (TEAM) {
            if (role == 0 && l < 8) ctl[l] = 0; // (every wave passed the previous level's closing barrier: nobody reads the control words now)
            this->interpolate(coarse, coarseBlkX, coarseBlkY, coarseLogPel, l + 64 * role, 64 * nw);
        } else this->interpolate(coarse, coarseBlkX, coarseBlkY, coarseLogPel);
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
        LDS_AS v2u *rowbuf = (LDS_AS v2u *)(TEAM ? shRow : lds + ldsRow); // the previous block row's results, 8 bytes per block: (x | y << 16, sad)
        lds_u8 *tab = TEAM ? lds + ldsTab : lds + ldsRow + ((nBlkX * 8 + 15) & ~15); // the group's SAD table follows the row buffer of THIS level (the host sizes the chain's LDS for the worst level); TEAM: in the wave's own area
        this->pf_setup();
        SPROF(8);
        auto lambda_of = [&](int predSad) { // :456-462, fp64 as the reference
            const double scale = (double)LSAD / (double)(LSAD + (long long)(predSad >> 1));
            return (int)(long long)((double)(long long)nLambdaLevel * scale * scale);
        };
        const bool hexLevel = searchType == SearchHex2;
        const bool specLevel = specEnabled && !smallestPlane;
        // refinement offsets of this lane's candidate group in pass A (4 lanes per candidate on Hex2 levels, 2 on exhaustive ones)
        int rdx = 0, rdy = 0;
        if (hexLevel) {
            const int g = l >> 2;
            if (g < 6) { rdx = tab8(HEX2X >> 8, g & 7); rdy = tab8(HEX2Y >> 8, g & 7); }
            else if (g < 14) { rdx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), g - 6); rdy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), g - 6); }
        } else {
            const int g = l >> 1, k = g < 8 ? g : g - 8;
            if (g < 8) { rdx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), k); rdy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), k); }
            else if (g < 16) { rdx = tab8(PACK8(-1, -1, 0, 0, 1, 1, -2, 2), k); rdy = tab8(PACK8(-2, 2, -2, 2, -2, 2, -1, -1), k); }
            else if (g < 24) { rdx = tab8(PACK8(-2, 2, -2, 2, -2, -2, 2, 2), k - 8); rdy = tab8(PACK8(0, 0, 1, 1, -2, 2, -2, 2), k - 8); }
        }
        // the pattern offsets of this lane's candidates in the row passes, one byte per pass (dx in the low, dy in the high nibble): a table
        // look-up by lane inside the pass loop would keep a dozen 64-bit table constants in vector registers across the whole level
        unsigned sPat = 0, bPat0 = 0, bPat1 = 0;
        {
            // (8-bit row passes: sixteen strip candidates of four lanes per pass, block candidates of fifteen lanes)
            const bool side8 = STRIP8_OK && stepX == BW; // (blocks side by side: eight block candidates of eight lanes per pass)
            // (16-bit: LPC lanes per block-form candidate -- 14 for 16x16 blocks.  r4 had the 14 written out here, which made the block-form stage 2 of a 32x32 build
            // (LPC = 12) mix two pattern points in the blocks of lanes 12-13, 24-27 and 36-41: the "8K clip that disagreed with the oracle" of r4 -- one block in
            // 128 851, profiles/r5_strip32_mismatch_found.txt)
            constexpr bool side16 = SIDE; // (16x16 blocks side by side: four blocks per window, eight lanes per block-form candidate)
            const int gb = side8 ? l >> 3 : STRIP8_OK ? (l >= 45 ? 3 : l >= 30 ? 2 : l >= 15 ? 1 : 0) : side16 ? min(l >> 3, 3) : min(l / LPC, 3);
            for (int q = 0; q < 8; q++) {
                int dx, dy;
                if (q < 4) { pat_delta(hexLevel, STRIP8_OK ? q * 16 + (l >> 2) : q * 8 + (l >> 3), dx, dy); sPat |= (unsigned)((dx & 15) | ((dy & 15) << 4)) << (8 * q); }
                pat_delta(hexLevel, q * (side8 ? 8 : 4) + gb, dx, dy);
                const unsigned b = (unsigned)((dx & 15) | ((dy & 15) << 4)) << (8 * (q & 3));
                if (q < 4) bPat0 |= b; else bPat1 |= b;
            }
        }
#ifdef MVX_SPEC_STATS
        unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0, st7 = 0;
#endif

        int prevX = 0, prevY = 0, prevSad = 0;
        int syncCount = 0;
        if (syncEvery > 0 && syncEvery < 32) syncEvery = 32; // (a group is the unit)
        int Gc = 0;                                 // groups of this level so far, in walk order (TEAM: the token's unit)
        const int nGrpRow = (nBlkX + SPEC_TB - 1) / SPEC_TB;
        for (int blky = 0; blky < nBlkY; blky++) {
            const bool fwd = (blky & 1) == 0 || !meander;
            const int dir = fwd ? 1 : -1;
            y0 = vpad + stepY * blky;
            nDyMax = (ph - y0 - BW - vpad + vps) << logPel; // :1094-1097 (the vertical limits are the row's)
            nDyMin = -((y0 - vpad + vps) << logPel);
            if constexpr (TEAM) gmvy = this->clipy(gmvy); // (every wave follows the running global predictor through the groups it does not own; the row's clip is idempotent)
            const bool specRow = specLevel && blky > 0;
            const int ngrp = (nBlkX + 63) >> 6;
            for (int gi = 0; gi < ngrp; gi++) {
                const int grp = fwd ? gi : ngrp - 1 - gi;
                const int c0 = grp << 6, c = c0 + l;
                const bool in = c < nBlkX;
                const int ncol = min(64, nBlkX - c0);
                // ---- 64 columns at a time (lane i <-> column c0 + i): the interpolated predictors of this row and of the blocks "below-ahead",
                // the previous row's results; lambda lane-parallel (:456-462)
                // ---- 64 columns at a time (lane i <-> column c0 + i).  What a group keeps in registers across its passes is little: the results and,
                // in speculated rows, four packed predictors and lambda per block; the live search of a single block reads its predictors itself.
                v4u bOut = {0, 0, 0, 0};
                for (int hi = 0; hi < 2; hi++) {
                    const int h = fwd ? hi : 1 - hi;
                    const int lo = h * SPEC_TB, hiE = min(lo + SPEC_TB, ncol);
                    if (lo >= ncol) continue;
                    const int myG = Gc++;
                    bool teamHeld = false; // TEAM: this wave holds the token (prevX / prevY / prevSad / badcount are the walk's)
                    if constexpr (TEAM) {
                        if (uni(myG % nw) != role) { // another wave's group: only the running global predictor moves on (:859)
                            gmvx = team_advance(gmvx, c0, lo, hiE, fwd, stepX, hps);
                            continue;
                        }
                        if (specRow) {
                            // phase A1 reads the previous block row's results of this group's columns (in the last block row also of the column ahead, :441-447):
                            // they are written when the token has passed the group(s) of the previous row that cover them
                            const int j = (c0 >> 5) + h;
                            const bool fwdPrev = ((blky - 1) & 1) == 0 || !meander;
                            auto idxPrev = [&](int jj) { return fwdPrev ? jj : nGrpRow - 1 - jj; };
                            int need = (blky - 1) * nGrpRow + idxPrev(j) + 1;
                            if (blky == nBlkY - 1) { const int j2 = j + dir; if (j2 >= 0 && j2 < nGrpRow) need = max(need, (blky - 1) * nGrpRow + idxPrev(j2) + 1); }
                            SPROF(7);
                            team_wait_ge(need);
                            SPROF(16);
                        }
                    }
                    SPROF(7);
                    if (syncEvery && (syncCount++ & ((syncEvery >> 5) - 1)) == 0) __builtin_amdgcn_s_barrier(); // keeps the chains of a workgroup on neighbouring blocks (shared reference lines): every syncEvery / 32 groups
                    SPROF(0);
                    const bool act = l >= lo && l < hiE;
                    unsigned long long okmask = 0, flagmask = 0;
                    int rX = 0, rY = 0, rSad = 0;          // this lane's block: speculative result
                    int pkU = 0, pkAh = 0, pkH = 0, pkG = 0; // this lane's block: clipped up / ahead / hierarchical / global predictors
                    int lam = 0;                             // this lane's block: lambda
                    bool staged2G = false;                   // the two-stage row passes ran (the predictor phase's outcome below is valid)
                    int lamG = 0, pBestG = 0, pkWG = 0;      // this lane's block: lambda, the predictor phase's cost and vector under the hypothesis
                    int gEndX = gmvx;
                    if (specRow) {
                        // ======== A1: one lane per block -- limits (:1094-1097) and the predictors that do not depend on the left neighbour (:427-449)
                        const int xs = stepX * c;
                        const int dxMin = -((xs + hps) << logPel), dxMax1 = ((pw - xs - hpad - BW - hpad + hps) << logPel) - 1;
                        auto cx = [&](int v) { return min(max(v, dxMin), dxMax1); };
                        auto cy = [&](int v) { return min(max(v, nDyMin), nDyMax - 1); };
                        // the interpolated predictors of this row and of the blocks "below-ahead", the previous row's results (one load each per 64 columns)
                        v4u bSelf = {0, 0, 0, 0}, bBelow = {0, 0, 0, 0};
                        unsigned upPk = 0;
                        if (in) {
                            bSelf = this->ld_batch(&vectors[blky * nBlkX + c]);
                            const int cb = c + dir;
                            if (blky < nBlkY - 1 && cb >= 0 && cb < nBlkX) bBelow = this->ld_batch(&vectors[(blky + 1) * nBlkX + cb]);
                            upPk = rowbuf[c][0];
                        }
                        lam = in ? lambda_of((int)bSelf[2]) : 0; // :456-462 lane-parallel (fp64): every level but the coarsest scales lambda by the block's OWN interpolated SAD
                        const int ux = cx(upx((int)upPk)), uy = cy(upy((int)upPk));
                        pkU = pk(ux, uy);
                        const bool aheadCol = fwd ? c < nBlkX - 1 : c > 0;
                        const bool useBelow = blky < nBlkY - 1 && aheadCol;
                        int ahx = 0, ahy = fieldShift;
                        if (useBelow) { ahx = (int)bBelow[0]; ahy = (int)bBelow[1]; }
                        else if (aheadCol && in) { const v2u t = rowbuf[c + dir]; ahx = upx((int)t[0]); ahy = upy((int)t[0]); } // last block row only (:441-447)
                        const int ax = cx(ahx), ay = cy(ahy);
                        pkAh = pk(ax, ay);
                        const int hx = cx((int)bSelf[0]), hy = cy((int)bSelf[1]);
                        pkH = pk(hx, hy);
                        // the global predictor is clipped cumulatively (:859): every block clips the running value with its own limits
                        gmvy = this->clipy(gmvy);
                        if (__ballot(act && cx(gmvx) != gmvx) == 0) pkG = pk(gmvx, gmvy);
                        else {
                            for (int i = 0; i < hiE - lo; i++) {
                                const int li = fwd ? lo + i : hiE - 1 - i;
                                const int xb = stepX * (c0 + li);
                                gmvx = min(max(gmvx, -((xb + hps) << logPel)), ((pw - xb - hpad - BW - hpad + hps) << logPel) - 1);
                                pkG = l == li ? pk(gmvx, gmvy) : pkG;
                            }
                        }
                        gEndX = gmvx;

                        // ---- A2's pieces (one lane per block; the table is read per lane): the predictor phase (:832-915) under the hypothesis left == median == up,
                        // and the refinement (:773-816) around a centre, costs as pobCheckMV0 / pobCheckMV (:219-261), strict < in the reference's order
                        const int ti8L = (l & (SPEC_TB - 1)) * 8;
                        auto rd = [&](int slot) { return *(const LDS_AS v2u *)(tab + slot * SPEC_STRIDE + ti8L); };
                        auto tot = [&](const v2u &t) { return (int)t[0] + (chroma ? (int)t[1] : 0); };
                        auto md = [&](int vx, int vy) { // motion_distortion (:105-114) around the hierarchical predictor
                            const unsigned dx = (unsigned)(hx - vx), dy = (unsigned)(hy - vy);
                            const int dist = (int)(dx * dx + dy * dy);
                            return (int)(((long long)lam * dist) >> 8);
                        };
                        auto a2_pred = [&](int &best, int &bx, int &by, int &bs) {
                            const int sUp = hexLevel ? 14 : 24, sZ = hexLevel ? 16 : 26;
                            const v2u tU = rd(sUp), tA = rd(sUp + 1), tZ = rd(sZ), tG = rd(sZ + 1), tH = rd(sZ + 2);
                            bx = 0; by = fieldShift;
                            { const int t = tot(tZ); int cc = t + (int)(((long long)penaltyZero * t) >> 8); cc = F::sat_add(0, cc); best = cc; bs = t; }
                            { const int t = tot(tG); int cc = t + (int)(((long long)pglobal * t) >> 8); cc = F::sat_add(0, cc); if (cc < best) { best = cc; bx = upx(pkG); by = upy(pkG); bs = t; } }
                            { const int t = tot(tH); const int cc = F::sat_add(0, t); if (cc < best) { best = cc; bx = hx; by = hy; bs = t; } }
                            { const int t = tot(tU); const int cc = F::sat_add(md(ux, uy), t); if (cc < best) { best = cc; bx = ux; by = uy; bs = t; } } // median = left = up
                            { const int t = tot(tA); const int cc = F::sat_add(md(ax, ay), t); if (cc < best) { best = cc; bx = ax; by = ay; bs = t; } }
                        };
                        // refinement around (wx, wy) from the table; returns true when a hexagon point beats the predictor phase (that moves the centre, :682-724: live)
                        auto a2_refine = [&](int wx, int wy, int &best, int &bx, int &by, int &bs) {
                            auto vok = [&](int vx, int vy) { return vx >= dxMin && vy >= nDyMin && vx <= dxMax1 && vy < nDyMax; };
                            auto cnew = [&](int vx, int vy, const v2u &t) { // pobCheckMV: penalty for new vectors, saturating
                                int cc = (int)t[0] + ((penaltyNew * (int)t[0]) >> 8);
                                if (chroma) cc += (int)t[1] + ((penaltyNew * (int)t[1]) >> 8);
                                return F::sat_add(md(vx, vy), cc);
                            };
                            bool won = false;
                            // (the table entries of the whole pattern are read up front: one after the other inside the loops below, each look-up's LDS latency was
                            // exposed -- the loops stay rolled for the registers the row passes need, the reads do not depend on them)
                            if (hexLevel) {
                                v2u tt[14];
#pragma unroll
                                for (int k = 0; k < 14; k++) tt[k] = rd(k);
                                if (nSearchParam > 1) {
#pragma unroll
                                    for (int k = 0; k < 6; k++) {
                                        const int vx = wx + tab8(HEX2X >> 8, k), vy = wy + tab8(HEX2Y >> 8, k);
                                        won = won || (vok(vx, vy) && cnew(vx, vy, tt[k]) < best);
                                    }
                                }
#pragma unroll
                                for (int k = 0; k < 8; k++) { // pobExpandingSearch(1, 1) (:636-658)
                                    const int vx = wx + tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), k), vy = wy + tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), k);
                                    const v2u t = tt[6 + k];
                                    const int cc = cnew(vx, vy, t);
                                    if (vok(vx, vy) && cc < best) { best = cc; bx = vx; by = vy; bs = tot(t); }
                                }
                            } else {
#pragma unroll 1
                                for (int k0 = 0; k0 < 24; k0 += 8) { // rings 1 and 2 (:786-791), eight entries at a time
                                    v2u tt[8];
#pragma unroll
                                    for (int k = 0; k < 8; k++) tt[k] = rd(k0 + k);
#pragma unroll
                                    for (int k = 0; k < 8; k++) {
                                        int dx, dy;
                                        if (k0 == 0) { dx = tab8(PACK8(0, 0, -1, 1, -1, -1, 1, 1), k); dy = tab8(PACK8(-1, 1, 0, 0, -1, 1, -1, 1), k); }
                                        else if (k0 == 8) { dx = tab8(PACK8(-1, -1, 0, 0, 1, 1, -2, 2), k); dy = tab8(PACK8(-2, 2, -2, 2, -2, 2, -1, -1), k); }
                                        else { dx = tab8(PACK8(-2, 2, -2, 2, -2, -2, 2, 2), k); dy = tab8(PACK8(0, 0, 1, 1, -2, 2, -2, 2), k); }
                                        const int vx = wx + dx, vy = wy + dy;
                                        const v2u t = tt[k];
                                        const int cc = cnew(vx, vy, t);
                                        if (vok(vx, vy) && cc < best) { best = cc; bx = vx; by = vy; bs = tot(t); }
                                    }
                                }
                            }
                            return won;
                        };
                        bool staged2 = false;                   // the two-stage row passes ran: the pattern lies around the predictor phase's winner
                        int pBest = 0, pX_ = 0, pY_ = fieldShift, pSad = 0; // this lane's block: the predictor phase's result

                        // ======== A: the SADs of every block of the group, nothing serial in between
                        const int nb = hiE - lo;
                        const bool lumaStrips = STRIP_OK && !UV && !chroma && stripEnabled && stepX == (SIDE ? BW : BW / 2); // luma-only search: row passes without UV rows
                        const bool streamed = (chroma && (hexLevel ? STREAM_HEX : STREAM_EXH)) || lumaStrips;
                        if (streamed) {
                            const int pkZ = pk(0, fieldShift);
                            const int slotUp = hexLevel ? 14 : 24, slotZ = hexLevel ? 16 : 26; // (ahead = slotUp + 1, global / hierarchical = slotZ + 1 / + 2)
                            unsigned long long pbMask = __ballot(act), p3Mask = 0; // blocks evaluated one at a time: every candidate / only ahead, global, hierarchical

                            SPROF(1);
                            // ======== A, row passes: windows of seven columns, in TWO stages.  Every pass loads ONE row per lane and load instruction
                            // (24 rows: 16 luma, 8 of the UV plane), against the window's source strip in LDS.
                            //   stage 1, one pass per window: up, ahead, global, hierarchical in block form (lane = (candidate, block, half): any vectors --
                            //     neighbouring blocks with similar vectors still read neighbouring bytes of the same lines in one instruction) and the zero
                            //     vector in strip form (lanes 56-63 = the window's eight 16-byte columns);
                            //   A2a, one lane per block: the predictor phase -> the block's refinement centre W (under the hypothesis left == up);
                            //   stage 2: the pattern around W.  Strip form where the window's blocks share W and keep the pattern inside their limits
                            //     (lane = (candidate, column), eight candidates per pass: 2 passes on Hex2 levels, 3 on exhaustive ones); block form
                            //     elsewhere (four candidates per pass: 4 / 6 passes).
                            // (the first form evaluated the pattern around UP and gave every block whose predictor phase ended elsewhere to the live
                            // search: one block in eight one level up from the finest, most blocks of the chains with an odd frame distance there)
                            if constexpr (STRIP_OK) {
                                // (r5: 16x16 blocks SIDE BY SIDE -- overlap 0, the reference's default -- take the same passes: a block is two columns and steps by TC = two,
                                // a window holds four; no column sum is shared, the strips stay contiguous)
                                constexpr bool side = SIDE;
                                if (stripEnabled && stepX == (side ? BW : BW / 2) && (UV ? chroma != 0 : chroma == 0)) {
                                    constexpr int TC = side ? 2 : HC, SWB = side ? 4 : SW_BLOCKS, LPCr = SWB * LPB; // columns per block step, blocks per window, lanes per block-form candidate
                                    const int nw = (nb + SWB - 1) / SWB;
                                    const int npat = hexLevel ? 14 : 24;
                                    pbMask = 0;
                                    staged2 = true;
                                    unsigned stripW = 0; // stage 2: windows in strip form
                                    int pkW = 0;         // this lane's block: the refinement centre
                                    // lane roles: strip form (candidate l >> 3, column l & 7); block form (candidate l / 14, block (l % 14) >> 1, half l & 1);
                                    // recomputed from an opaque lane number in every pass (values derived from the lane number are loop invariants: the
                                    // compiler hoists all of them to the top of the level and then has to spill them)
                                    int lq = l;
                                    asm volatile("" : "+v"(lq));
                                    int gS, pS, gB, rB, mB, hB;
                                    bool tail, idleB; // lanes 56-63: stage 1: the zero vector's strip; block form: idle, like the lanes between 4 * LPC and 56
                                    auto roles = [&]() {
                                        lq = l;
                                        asm volatile("" : "+v"(lq));
                                        gS = lq >> 3; pS = lq & 7;
                                        gB = side ? min(lq >> 3, 3) : min(lq / LPC, 3);
                                        rB = side ? (lq & 7) : lq >= 4 * LPC ? (lq - 4 * LPC) % LPC : lq - LPC * gB; mB = rB / LPB; hB = rB % LPB;
                                        tail = lq >= 56; idleB = lq >= 4 * LPCr;
                                    };
                                    roles();
                                    // this lane's share of pass q of window w in stage st: table slot (-1: nothing to write), the column it writes, whether its
                                    // block sum is "column + next column" (strip) or "half + other half" (block), source column, first reference piece
                                    auto w_cand = [&](int st, int w, int q, int &slot, int &colW, bool &stripLane, int &srcCol, unsigned &oA, unsigned &oB) {
                                        const int f = lo + SWB * w, L = min(SWB, hiE - f);
                                        const int bxf = hpad + stepX * (c0 + f);
                                        const bool stripWin = st == 2 && ((stripW >> w) & 1);
                                        // the block lanes' vectors come from the lanes that own the blocks: fetched HERE, with every lane active (ds_bpermute returns 0
                                        // for a source lane that is masked off, and in stage 1 lanes 56-63 take the other branch below)
                                        const int meB = min(mB, L - 1), colB = f + meB; // (lanes beyond the window repeat its last block)
                                        int bU = 0, bAh = 0, bG = 0, bH = 0, bW = 0;
                                        if (st == 1) {
                                            bU = __builtin_amdgcn_ds_bpermute(colB << 2, pkU); bAh = __builtin_amdgcn_ds_bpermute(colB << 2, pkAh);
                                            bG = __builtin_amdgcn_ds_bpermute(colB << 2, pkG); bH = __builtin_amdgcn_ds_bpermute(colB << 2, pkH);
                                        } else if (!stripWin) bW = __builtin_amdgcn_ds_bpermute(colB << 2, pkW);
                                        if (stripWin || (st == 1 && tail)) { // strip lanes: one vector for the whole window
                                            const int p = st == 1 ? pS : pS, g = gS;
                                            int vx, vy, vyc;
                                            if (st == 1) { vx = 0; vy = fieldShift; vyc = 0; slot = slotZ; } // (the zero candidate's chroma ignores fieldShift, :836-839)
                                            else {
                                                const int sW = __builtin_amdgcn_readlane(pkW, f);
                                                const int idx = q * 8 + g;
                                                const int dd = (int)(sPat >> (8 * q)), dx = (dd << 28) >> 28, dy = (dd << 24) >> 28;
                                                vx = upx(sW) + dx; vy = upy(sW) + dy; vyc = vy;
                                                slot = idx < npat ? idx : -1;
                                            }
                                            const int pe = min(p, TC * (L - 1) + 2 * HC - 1) * COLB; // (columns beyond the run re-read its last one)
                                            if ((p % TC != 0) | (p / TC >= L)) slot = -1;  // (block m is written by the lane of its first column)
                                            colW = f + p / TC; stripLane = true; srcCol = p;
                                            oA = luma_off_at(bxf, vx, vy) + (unsigned)pe;
                                            oB = 2 * chroma_off_at(bxf, vx, vyc) + (unsigned)pe;
                                        } else { // block lanes: every block its own vector
                                            const int me = meB, col = colB;
                                            int base, dx = 0, dy = 0;
                                            if (st == 1) {
                                                base = gB == 0 ? bU : gB == 1 ? bAh : gB == 2 ? bG : bH;
                                                slot = gB == 0 ? slotUp : gB == 1 ? slotUp + 1 : slotZ + gB - 1;
                                            } else {
                                                base = bW;
                                                const int ci = q * 4 + gB;
                                                const int dd = (int)((q < 4 ? bPat0 : bPat1) >> (8 * (q & 3)));
                                                dx = (dd << 28) >> 28; dy = (dd << 24) >> 28;
                                                slot = ci < npat ? ci : -1;
                                            }
                                            const int bx0 = hpad + stepX * (c0 + col);
                                            const int xMax = (pw - bx0 - BW - hpad + hps) << logPel, xMin = -((bx0 - hpad + hps) << logPel);
                                            const int cxv = upx(base), cyv = upy(base), tx = cxv + dx, ty = cyv + dy;
                                            const bool ok = (tx >= xMin) & (ty >= nDyMin) & (tx < xMax) & (ty < nDyMax); // (outside the block's limits: the centre instead; A2 never reads the entry)
                                            const int vx = ok ? tx : cxv, vy = ok ? ty : cyv;
                                            if ((hB != 0) | (mB >= L) | idleB) slot = -1;
                                            colW = f + me; stripLane = false; srcCol = TC * me + hB;
                                            oA = luma_off_at(bx0, vx, vy) + (unsigned)(hB * COLB);
                                            oB = 2 * chroma_off_at(bx0, vx, vy) + (unsigned)(hB * COLB);
                                        }
                                    };
                                    // the source strip of window w: lane = (row l >> 3 of 8, column l & 7); luma rows r and r + 8, UV row r
                                    // r6, SDMA: sixteen-byte columns, 16x16 blocks -- lane l's piece of rows 8k .. 8k + 7 belongs at byte 16 l of that kilobyte of the strip, which is where
                                    // global_load_lds_dwordx4 puts it: the strip never passes through registers (twelve fewer live across the passes, no ds_write).  Two buffers: the next
                                    // window's strip arrives while this window's passes read theirs.  A strip is requested BEFORE the 24 row loads that follow it (the prime, or a pass's
                                    // refills), so "all but the 24 newest loads have returned" (vmcnt is in order) means it is there
                                    constexpr bool SDMA = MVX_STRIP_DMA && COLB == 16 && HC == 1 && SW == 24 && SNT == 24;
                                    constexpr int SBUF = SNT * ROWB; // bytes of one strip buffer (the host sizes the chain's LDS for two)
                                    int sbuf = 0;                    // the buffer of the window whose passes run
                                    A4x32 stg[SDMA ? 1 : SSTG]; // (rows gS, gS + 8, ...: the luma rows first, then the rows of the UV plane)
                                    auto stage_issue = [&](int w, int buf) {
                                        const int f = lo + SWB * w, L = min(SWB, hiE - f), bx0 = hpad + stepX * (c0 + f), pe = min(pS, TC * (L - 1) + 2 * HC - 1) * COLB;
#pragma unroll
                                        for (int k = 0; k < SSTG; k++) {
                                            const int row = gS + 8 * k;
                                            gl_u8 *g = 8 * k < SNA ? srcY + (unsigned)(y0 + row) * pitchY + (unsigned)bx0 * BPS + (unsigned)pe
                                                                   : srcUV + (unsigned)((y0 >> 1) + row - SNA) * 2 * pitchC + (unsigned)(bx0 >> 1) * 2 * BPS + (unsigned)pe;
                                            if constexpr (SDMA) {
                                                const unsigned m0v = (unsigned)(unsigned long long)(lds + buf * SBUF + 8 * k * ROWB); // (wave-uniform: the LDS address of lane 0's piece)
                                                asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(m0v) : "memory", "m0");
                                            } else stg[k] = ld_chunk_g(g, COLB);
                                        }
                                    };
                                    auto stage_store = [&]() {
                                        if constexpr (SDMA) asm volatile("s_waitcnt vmcnt(24)" : : : "memory");
                                        else {
#pragma unroll
                                            for (int k = 0; k < SSTG; k++) st_chunk_l(lds + (gS + 8 * k) * ROWB + pS * COLB, stg[k], COLB); // (S_UV = SNA * ROWB: the UV rows follow the luma rows)
                                        }
                                    };
                                    // (r6: the two stages WINDOW BY WINDOW -- stage 1, predictor phase, stage 2 of one window before the next window's stage 1, so that stage 2
                                    // would find stage 1's lines in the L2 -- fetches as much as this order: 1 357 against 1 302 GB per 2046-chain launch, profiles/r6_window_major_ab.txt.
                                    // On a clip whose vectors are whole pels stage 1 reads the integer-pel plane and stage 2 mostly the three others: little to share)
                                    // (a leading-edge prefetch -- one dword of every line a window two ahead will need, four scattered loads per window --
                                    // was measured and removed: 493 -> 544 ms per 2046-chain launch, profiles/r4_spec_prefetch.txt)
                                    auto widx = [&](int i) { return fwd ? i : nw - 1 - i; }; // windows in walk order
                                    auto npass = [&](int st, int w) { return st == 1 ? 1 : ((stripW >> w) & 1) ? (npat + 7) / 8 : (npat + 3) / 4; };
                                    // one stage: a stream of passes whose loads stay in flight across passes and windows
                                    auto run_stage = [&](int st, int wFrom, int wTo) { // the windows wFrom .. wTo - 1 in walk order
                                        int wi = wFrom, w = widx(wFrom), q = 0;
                                        StripPass T;
                                        int slot, colW, srcCol; bool stripLane; unsigned oA, oB;
                                        roles();
                                        w_cand(st, w, q, slot, colW, stripLane, srcCol, oA, oB);
                                        if constexpr (SDMA) { sbuf = 0; stage_issue(w, 0); strip_prime(T, oA, oB); } // (the strip first: see above)
                                        else { strip_prime(T, oA, oB); stage_issue(w, 0); }
                                        for (;;) {
                                            roles();
                                            int wn = w, qn = q + 1, win = wi;
                                            if (qn >= npass(st, w)) { qn = 0; win = wi + 1; wn = widx(win); }
                                            const bool more = win < wTo;
                                            int slotN = -1, colN = 0, srcN = 0; bool stripN = false; unsigned nA = 0, nB = 0;
                                            SPROF(2);
                                            if (more) w_cand(st, wn, qn, slotN, colN, stripN, srcN, nA, nB); // the pass after this one (its loads refill the window of loads while this one is consumed)
                                            SPROF(11);
                                            if (q == 0) { // a new window: its source strip (requested one window ahead)
                                                __builtin_amdgcn_wave_barrier();
                                                stage_store();
                                                if (wi + 1 < wTo) stage_issue(widx(wi + 1), sbuf ^ 1);
                                                __builtin_amdgcn_wave_barrier();
                                            }
                                            SPROF(12);
                                            if (more) strip_run<true>(T, srcCol, nA, nB, SDMA ? sbuf * SBUF : 0); else strip_run<false>(T, srcCol, 0, 0, SDMA ? sbuf * SBUF : 0);
                                            SPROF(13);
                                            // strip lanes: block m = columns m and m + 1; block lanes: the two halves of a block sit in neighbouring lanes
                                            unsigned sL = T.aL + (unsigned)__builtin_amdgcn_update_dpp(0, (int)T.aL, 0x101, 0xf, 0xf, true), sC = T.aC + (unsigned)__builtin_amdgcn_update_dpp(0, (int)T.aC, 0x101, 0xf, 0xf, true);
                                            unsigned hL = T.aL + (unsigned)__builtin_amdgcn_update_dpp(0, (int)T.aL, 0xB1, 0xf, 0xf, true), hC = T.aC + (unsigned)__builtin_amdgcn_update_dpp(0, (int)T.aC, 0xB1, 0xf, 0xf, true);
                                            if (HC == 2) { // a 32x32 block: four columns / four lanes
                                                sL += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sL, 0x102, 0xf, 0xf, true); sC += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sC, 0x102, 0xf, 0xf, true);
                                                hL += (unsigned)__builtin_amdgcn_update_dpp(0, (int)hL, 0x4E, 0xf, 0xf, true); hC += (unsigned)__builtin_amdgcn_update_dpp(0, (int)hC, 0x4E, 0xf, 0xf, true);
                                            }
                                            if (slot >= 0) *(LDS_AS v2u *)(tab + slot * SPEC_STRIDE + (colW & (SPEC_TB - 1)) * 8) = v2u{stripLane ? sL : hL, stripLane ? sC : hC};
                                            SPROF(14);
#ifdef MVX_SPEC_PROF
                                            sprof[15] += 1;
#endif
                                            if (!more) break;
                                            if (SDMA && win != wi) sbuf ^= 1;
                                            w = wn; wi = win; q = qn; slot = slotN; colW = colN; stripLane = stripN; srcCol = srcN;
                                        }
                                        __builtin_amdgcn_wave_barrier();
                                    };
                                    run_stage(1, 0, nw);
                                    a2_pred(pBest, pX_, pY_, pSad); // the predictor phase of every block of the group
                                    pkW = pk(pX_, pY_);
#if MVX_SPEC_ABL == 9
                                    if (SPECDBG_HERE() && act) { // the predictor phase of this lane's block
                                        int *o = g_specdbg + l * SPECDBG_N;
                                        const int sUp = hexLevel ? 14 : 24, sZ = hexLevel ? 16 : 26;
                                        o[0] = pkU; o[1] = pkAh; o[2] = pkG; o[3] = pkH; o[4] = pkW; o[5] = pBest;
                                        o[6] = tot(rd(sUp)); o[7] = tot(rd(sUp + 1)); o[8] = tot(rd(sZ)); o[9] = tot(rd(sZ + 1)); o[10] = tot(rd(sZ + 2)); o[11] = lam;
                                    }
#endif
                                    { // windows whose blocks share the centre and keep the whole pattern inside their limits
                                        const bool ok2 = (pX_ - 2 >= dxMin) & (pX_ + 2 <= dxMax1) & (pY_ - 2 >= nDyMin) & (pY_ + 2 < nDyMax);
                                        for (int w = 0; w < nw; w++) {
                                            const int f = lo + SWB * w, e = min(f + SWB, hiE);
                                            const bool inw = (l >= f) & (l < e);
                                            const int w0 = __builtin_amdgcn_readlane(pkW, f);
                                            if (MVX_SPEC_ABL != 7 && e - f >= 2 && __ballot(inw & ((pkW != w0) | !ok2)) == 0) stripW |= 1u << w; // (ABL 7: block form only)
#ifdef MVX_SPEC_STATS
                                            if ((stripW >> w) & 1) st4 += 1;
                                            else { st5 += 1; st6 += __builtin_popcountll(__ballot(inw & (pkW != w0))); if (__ballot(inw & (pkW != w0)) == 0) st7 += 1; }
#endif
                                        }
                                    }
                                    run_stage(2, 0, nw);
                                }
                            }

                            // ======== A, row passes of 8-bit clips (8x8 blocks overlapping by half or not at all): the same two stages, windows of up to 15 / 8 blocks
                            if constexpr (STRIP8_OK) {
                                // (a window has at least four dwords: groups of one or two blocks go one block at a time)
                                auto rows8 = [&](auto DS2) {
                                    constexpr bool ds2 = decltype(DS2)::value;    // a block step is two dwords: blocks do not overlap (a compile-time copy each: as a run-time flag it cost cfg2 4 %)
                                    const int wmax = ds2 ? 8 : W8_BLOCKS;        // sixteen dwords: 15 blocks of two dwords overlapping by one / 8 blocks side by side
                                    const int nw = (nb + wmax - 1) / wmax, WL = (nb + nw - 1) / nw; // windows of equal length (32 blocks overlapping: 11, 11, 10)
                                    const int npat = hexLevel ? 14 : 24;
                                    pbMask = 0;
                                    staged2 = true;
                                    unsigned stripW = 0;
                                    int pkW = 0;
                                    int lq, gS, qS, gB, mB;
                                    bool tail;
                                    // strip form: candidate l >> 2, lane-in-candidate l & 3; block form: candidate l / 15, block l % 15 (no overlap: l >> 3, l & 7);
                                    // lanes 60-63: stage 1: the zero vector's strip
                                    auto roles = [&]() {
                                        lq = l;
                                        asm volatile("" : "+v"(lq));
                                        gS = lq >> 2; qS = lq & 3;
                                        if (ds2) { gB = lq >> 3; mB = lq & 7; }
                                        else { gB = min(lq / W8_BLOCKS, 3); mB = lq >= 60 ? lq - 60 : lq - W8_BLOCKS * gB; }
                                        tail = lq >= 60;
                                    };
                                    roles();
                                    // this lane's share of pass q of window w in stage st: table slot (-1: nothing), first column it writes, how many consecutive
                                    // blocks it writes (strip lanes: up to four; block lanes: one), its source dwords, its first reference piece
                                    auto w_cand = [&](int st, int w, int q, int &slot, int &colW, int &nwr, int &srcA, int &srcB, int &shA, int &shB, unsigned &oA, unsigned &oB) {
                                        const int f = lo + WL * w, L = min(WL, hiE - f);
                                        const int bxf = hpad + stepX * (c0 + f);
                                        const bool stripWin = st == 2 && ((stripW >> w) & 1);
                                        const int meB = min(mB, L - 1), colB = f + meB;
                                        int bU = 0, bAh = 0, bG = 0, bH = 0, bW = 0; // (fetched with every lane active: ds_bpermute reads 0 from a masked-off lane)
                                        if (st == 1) {
                                            bU = __builtin_amdgcn_ds_bpermute(colB << 2, pkU); bAh = __builtin_amdgcn_ds_bpermute(colB << 2, pkAh);
                                            bG = __builtin_amdgcn_ds_bpermute(colB << 2, pkG); bH = __builtin_amdgcn_ds_bpermute(colB << 2, pkH);
                                        } else if (!stripWin) bW = __builtin_amdgcn_ds_bpermute(colB << 2, pkW);
                                        const int nd = ds2 ? 2 * L : L + 1; // the window's dwords
                                        if (stripWin || (st == 1 && tail)) { // strip lanes: lane qS of its candidate holds dwords 4 qS .. 4 qS + 3 of the window
                                            int vx, vy, vyc;
                                            if (st == 1) { vx = 0; vy = fieldShift; vyc = 0; slot = slotZ; }
                                            else {
                                                const int sW = __builtin_amdgcn_readlane(pkW, f);
                                                const int idx = q * 16 + gS;
                                                const int dd = (int)(sPat >> (8 * q)), dx = (dd << 28) >> 28, dy = (dd << 24) >> 28;
                                                vx = upx(sW) + dx; vy = upy(sW) + dy; vyc = vy;
                                                slot = idx < npat ? idx : -1;
                                            }
                                            // the lane that holds the window's last dwords starts early enough to end with them, lanes beyond repeat it
                                            const int qe = min(qS, (nd - 1) >> 2), d0 = min(4 * qe, nd - 4);
                                            nwr = ds2 ? max(0, min(2, L - 2 * qS)) : max(0, min(4, L - 4 * qS));
                                            if (nwr == 0) slot = -1;
                                            colW = f + (ds2 ? 2 : 4) * qS; srcA = srcB = d0 * 4; shA = shB = 4 * qe - d0;
                                            oA = luma_off_at(bxf, vx, vy) + (unsigned)(d0 * 4);
                                            oB = 2 * chroma_off_at(bxf, vx, vyc) + (unsigned)(d0 * 4);
                                        } else { // block lanes
                                            int base, dx = 0, dy = 0;
                                            if (st == 1) {
                                                base = gB == 0 ? bU : gB == 1 ? bAh : gB == 2 ? bG : bH;
                                                slot = gB == 0 ? slotUp : gB == 1 ? slotUp + 1 : slotZ + gB - 1;
                                            } else {
                                                base = bW;
                                                const int ci = q * (ds2 ? 8 : 4) + gB;
                                                const int dd = (int)((q < 4 ? bPat0 : bPat1) >> (8 * (q & 3)));
                                                dx = (dd << 28) >> 28; dy = (dd << 24) >> 28;
                                                slot = ci < npat ? ci : -1;
                                            }
                                            const int bx0 = hpad + stepX * (c0 + colB);
                                            const int xMax = (pw - bx0 - BW - hpad + hps) << logPel, xMin = -((bx0 - hpad + hps) << logPel);
                                            const int cxv = upx(base), cyv = upy(base), tx = cxv + dx, ty = cyv + dy;
                                            const bool ok = (tx >= xMin) & (ty >= nDyMin) & (tx < xMax) & (ty < nDyMax);
                                            const int vx = ok ? tx : cxv, vy = ok ? ty : cyv;
                                            if ((mB >= L) | (st == 1 ? gB >= 4 : !ds2 & tail)) slot = -1;
                                            nwr = 1; colW = colB;
                                            // sixteen bytes from the block's first sample that would cross the end of the row: the eight in front of the block instead
                                            const int xl = ((bx0 << logPel) + vx) >> logPel, xc = 2 * ((((bx0 >> 1) << logPel) + ((vx + (vx < 0 ? 1 : 0)) >> 1)) >> logPel);
                                            shA = xl + 16 > pw ? 2 : 0; shB = xc + 16 > pw ? 2 : 0;
                                            const int dB = ds2 ? 2 * meB : meB; // the block's first dword in the window
                                            srcA = (dB - shA) * 4; srcB = (dB - shB) * 4;
                                            oA = luma_off_at(bx0, vx, vy) - (unsigned)(shA * 4);
                                            oB = 2 * chroma_off_at(bx0, vx, vy) - (unsigned)(shB * 4);
                                        }
                                    };
                                    // the source strip of window w: 12 rows x 64 bytes, one 16-byte piece per lane (lanes 0-47)
                                    A4x32 stg;
                                    auto stage_issue = [&](int w) {
                                        const int f = lo + WL * w, L = min(WL, hiE - f), bx0 = hpad + stepX * (c0 + f);
                                        const int row = min(lq >> 2, R8T - 1), pe = min(lq & 3, ((ds2 ? 2 * L : L + 1) - 1) >> 2) * 16; // (a source block ends hpad samples before its row and vpad rows before its plane)
                                        gl_u8 *p8 = row < R8A ? srcY + (unsigned)(y0 + row) * pitchY + (unsigned)bx0 + (unsigned)pe
                                                              : srcUV + (unsigned)((y0 >> 1) + row - R8A) * 2 * pitchC + (unsigned)(bx0 >> 1) * 2 + (unsigned)pe;
                                        stg = ld_chunk_g(p8, 16);
                                    };
                                    auto stage_store = [&]() { if (lq < 4 * R8T) st_chunk_l(lds + S8_BASE + (lq >> 2) * 64 + (lq & 3) * 16, stg, 16); };
                                    auto widx = [&](int i) { return fwd ? i : nw - 1 - i; };
                                    auto npass = [&](int st, int w) { return st == 1 ? 1 : ((stripW >> w) & 1) ? (npat + 15) / 16 : ds2 ? (npat + 7) / 8 : (npat + 3) / 4; };
                                    auto run_stage = [&](int st) {
                                        int wi = 0, w = widx(0), q = 0;
                                        Strip8 T;
                                        int slot, colW, nwr, srcA, srcB, shA, shB; unsigned oA, oB;
                                        roles();
                                        w_cand(st, w, q, slot, colW, nwr, srcA, srcB, shA, shB, oA, oB);
                                        strip8_prime(T, oA, oB);
                                        stage_issue(w);
                                        for (;;) {
                                            roles();
                                            int wn = w, qn = q + 1, win = wi;
                                            if (qn >= npass(st, w)) { qn = 0; win = wi + 1; wn = widx(win); }
                                            const bool more = win < nw;
                                            int slotN = -1, colN = 0, nwrN = 0, srcAN = 0, srcBN = 0, shAN = 0, shBN = 0; unsigned nA = 0, nB = 0;
                                            if (more) w_cand(st, wn, qn, slotN, colN, nwrN, srcAN, srcBN, shAN, shBN, nA, nB);
                                            if (q == 0) {
                                                __builtin_amdgcn_wave_barrier();
                                                stage_store();
                                                if (wi + 1 < nw) stage_issue(widx(wi + 1));
                                                __builtin_amdgcn_wave_barrier();
                                            }
                                            if (more) strip8_run<true>(T, srcA, srcB, shA, shB, nA, nB); else strip8_run<false>(T, srcA, srcB, shA, shB, 0, 0);
                                            // overlapping blocks: block m = dwords m and m + 1, the fourth block of a strip lane needs the next lane's first dword;
                                            // blocks side by side: block m = dwords 2 m and 2 m + 1
                                            const unsigned xL = (unsigned)__builtin_amdgcn_update_dpp(0, (int)T.aL[0], 0x101, 0xf, 0xf, true), xC = (unsigned)__builtin_amdgcn_update_dpp(0, (int)T.aC[0], 0x101, 0xf, 0xf, true);
                                            if (slot >= 0) {
                                                lds_u8 *tp = tab + slot * SPEC_STRIDE;
                                                *(LDS_AS v2u *)(tp + (colW & (SPEC_TB - 1)) * 8) = v2u{T.aL[0] + T.aL[1], T.aC[0] + T.aC[1]};
                                                if (ds2) {
                                                    if (nwr > 1) *(LDS_AS v2u *)(tp + ((colW + 1) & (SPEC_TB - 1)) * 8) = v2u{T.aL[2] + T.aL[3], T.aC[2] + T.aC[3]};
                                                } else {
                                                    if (nwr > 1) *(LDS_AS v2u *)(tp + ((colW + 1) & (SPEC_TB - 1)) * 8) = v2u{T.aL[1] + T.aL[2], T.aC[1] + T.aC[2]};
                                                    if (nwr > 2) *(LDS_AS v2u *)(tp + ((colW + 2) & (SPEC_TB - 1)) * 8) = v2u{T.aL[2] + T.aL[3], T.aC[2] + T.aC[3]};
                                                    if (nwr > 3) *(LDS_AS v2u *)(tp + ((colW + 3) & (SPEC_TB - 1)) * 8) = v2u{T.aL[3] + xL, T.aC[3] + xC};
                                                }
                                            }
                                            if (!more) break;
                                            w = wn; wi = win; q = qn; slot = slotN; colW = colN; nwr = nwrN; srcA = srcAN; srcB = srcBN; shA = shAN; shB = shBN;
                                        }
                                        __builtin_amdgcn_wave_barrier();
                                    };
                                    run_stage(1);
                                    a2_pred(pBest, pX_, pY_, pSad);
                                    pkW = pk(pX_, pY_);
                                    {
                                        const bool ok2 = (pX_ - 2 >= dxMin) & (pX_ + 2 <= dxMax1) & (pY_ - 2 >= nDyMin) & (pY_ + 2 < nDyMax);
                                        for (int w = 0; w < nw; w++) {
                                            const int f = lo + WL * w, e = min(f + WL, hiE);
                                            const bool inw = (l >= f) & (l < e);
                                            const int w0 = __builtin_amdgcn_readlane(pkW, f);
                                            if (e - f >= 2 && __ballot(inw & ((pkW != w0) | !ok2)) == 0) stripW |= 1u << w;
                                        }
                                    }
                                    run_stage(2);
                                };
                                if (stripEnabled && nb >= 3) {
                                    if (stepX == BW / 2) rows8(std::false_type());
                                    else if (stepX == BW) rows8(std::true_type());
                                }
                            }

                            SPROF(2);
                            // ======== A, one block at a time: what no run covers
                            // the candidates of block li (lane roles as in the table's slots), clipped to what may be loaded: a candidate outside the
                            // block's limits is replaced by the centre (its table entry is never read: A2 checks the limits itself)
                            auto r_cand = [&](int li, int &bx0, int &vx, int &vy, int &vyc) { // pattern pass (Hex2 levels) / the only pass (exhaustive levels)
                                bx0 = hpad + stepX * (c0 + li);
                                const int xMax = (pw - bx0 - BW - hpad + hps) << logPel, xMin = -((bx0 - hpad + hps) << logPel);
                                const int sU = __builtin_amdgcn_readlane(pkU, li), sAh = __builtin_amdgcn_readlane(pkAh, li);
                                int base;
                                if (hexLevel) base = (l >> 2) == 15 ? sAh : sU;
                                else {
                                    const int g = l >> 1;
                                    const int sH = __builtin_amdgcn_readlane(pkH, li), sG = __builtin_amdgcn_readlane(pkG, li);
                                    base = g == 25 ? sAh : g == 26 ? pkZ : g == 27 ? sG : g == 28 ? sH : sU;
                                }
                                const int cxv = upx(base), cyv = upy(base);
                                const int tx = cxv + rdx, ty = cyv + rdy;
                                const bool ok = (tx >= xMin) & (ty >= nDyMin) & (tx < xMax) & (ty < nDyMax);
                                vx = ok ? tx : cxv; vy = ok ? ty : cyv;
                                vyc = (!hexLevel && (l >> 1) == 26) ? 0 : vy; // the zero candidate's chroma ignores fieldShift (:836-839)
                            };
                            // three candidates, 1 << LOGGP lanes each: zero (ahFirst: ahead), global, hierarchical
                            auto p_cand = [&](int li, bool ahFirst, int &bx0, int &vx, int &vy, int &vyc) {
                                bx0 = hpad + stepX * (c0 + li);
                                const int g = l >> LOGGP;
                                const int sH = __builtin_amdgcn_readlane(pkH, li), sG = __builtin_amdgcn_readlane(pkG, li), sAh = __builtin_amdgcn_readlane(pkAh, li);
                                const int base = g == 1 ? sG : g == 2 ? sH : ahFirst ? sAh : pkZ;
                                vx = upx(base); vy = upy(base); vyc = ((g == 1) | (g == 2) | ahFirst) ? vy : 0;
                            };
                            auto first_of = [](unsigned long long m) { return (int)__builtin_ctzll(m); };
                            if (pbMask) {
                                A4x32 pf[G::NPF];
                                unsigned long long m = pbMask;
                                int li = first_of(m);
                                m &= m - 1;
                                this->pf_issue(hpad + stepX * (c0 + li), y0, pf);
                                if (hexLevel) {
                                    if constexpr (STREAM_HEX) {
                                        Pass<2, 12> R; Pass<LOGGP, 4> Z;
                                        const int sR = l & 3, sZ = l & ((1 << LOGGP) - 1);
                                        {
                                            int bx0, vx, vy, vyc; unsigned oA, oB;
                                            r_cand(li, bx0, vx, vy, vyc); this->template pass_start<2, 12>(sR, bx0, vx, vy, vyc, oA, oB); this->template pass_prime<2, 12>(R, oA, oB);
                                            p_cand(li, false, bx0, vx, vy, vyc); this->template pass_start<LOGGP, 4>(sZ, bx0, vx, vy, vyc, oA, oB); this->template pass_prime<LOGGP, 4>(Z, oA, oB);
                                        }
                                        for (;;) {
                                            const bool more = m != 0;
                                            const int lin = more ? first_of(m) : li;
                                            m &= m - 1;
                                            __builtin_amdgcn_wave_barrier();
                                            this->pf_store(pf); // (PlaneOfBlocks.cpp:1058-1079)
                                            if (more) this->pf_issue(hpad + stepX * (c0 + lin), y0, pf);
                                            int bx0, vx, vy, vyc; unsigned nA = 0, nB = 0, mA = 0, mB = 0;
                                            if (more) {
                                                r_cand(lin, bx0, vx, vy, vyc); this->template pass_start<2, 12>(sR, bx0, vx, vy, vyc, nA, nB);
                                                p_cand(lin, false, bx0, vx, vy, vyc); this->template pass_start<LOGGP, 4>(sZ, bx0, vx, vy, vyc, mA, mB);
                                            }
                                            __builtin_amdgcn_wave_barrier();
                                            if (more) { this->template pass_run<2, 12, true>(R, sR, nA, nB); this->template pass_run<LOGGP, 4, true>(Z, sZ, mA, mB); }
                                            else { this->template pass_run<2, 12, false>(R, sR, 0, 0); this->template pass_run<LOGGP, 4, false>(Z, sZ, 0, 0); }
                                            const int ti8 = (li & (SPEC_TB - 1)) * 8;
                                            group_sum2<2>(R.aL, R.aC);
                                            if (sR == 0) *(LDS_AS v2u *)(tab + (l >> 2) * SPEC_STRIDE + ti8) = v2u{R.aL, R.aC};
                                            group_sum2<LOGGP>(Z.aL, Z.aC);
                                            if ((sZ == 0) & ((l >> LOGGP) < 3)) *(LDS_AS v2u *)(tab + (16 + (l >> LOGGP)) * SPEC_STRIDE + ti8) = v2u{Z.aL, Z.aC};
                                            if (!more) break;
                                            li = lin;
                                        }
                                    }
                                } else {
                                    if constexpr (STREAM_EXH) {
                                        Pass<1, 12> R;
                                        const int sR = l & 1;
                                        {
                                            int bx0, vx, vy, vyc; unsigned oA, oB;
                                            r_cand(li, bx0, vx, vy, vyc); this->template pass_start<1, 12>(sR, bx0, vx, vy, vyc, oA, oB); this->template pass_prime<1, 12>(R, oA, oB);
                                        }
                                        for (;;) {
                                            const bool more = m != 0;
                                            const int lin = more ? first_of(m) : li;
                                            m &= m - 1;
                                            __builtin_amdgcn_wave_barrier();
                                            this->pf_store(pf);
                                            if (more) this->pf_issue(hpad + stepX * (c0 + lin), y0, pf);
                                            int bx0, vx, vy, vyc; unsigned nA = 0, nB = 0;
                                            if (more) { r_cand(lin, bx0, vx, vy, vyc); this->template pass_start<1, 12>(sR, bx0, vx, vy, vyc, nA, nB); }
                                            __builtin_amdgcn_wave_barrier();
                                            if (more) this->template pass_run<1, 12, true>(R, sR, nA, nB); else this->template pass_run<1, 12, false>(R, sR, 0, 0);
                                            const int ti8 = (li & (SPEC_TB - 1)) * 8;
                                            group_sum2<1>(R.aL, R.aC);
                                            if ((sR == 0) & ((l >> 1) < SPEC_SLOTS_EXH)) *(LDS_AS v2u *)(tab + (l >> 1) * SPEC_STRIDE + ti8) = v2u{R.aL, R.aC};
                                            if (!more) break;
                                            li = lin;
                                        }
                                    }
                                }
                            }
                            // ahead / global / hierarchical of the runs that do not share them
                            if constexpr (STRIP_OK) {
                                if (p3Mask) {
                                    A4x32 pf[G::NPF];
                                    unsigned long long m = p3Mask;
                                    int li = first_of(m);
                                    m &= m - 1;
                                    this->pf_issue(hpad + stepX * (c0 + li), y0, pf);
                                    Pass<LOGGP, 4> Z;
                                    const int sZ = l & ((1 << LOGGP) - 1), g = l >> LOGGP;
                                    {
                                        int bx0, vx, vy, vyc; unsigned oA, oB;
                                        p_cand(li, true, bx0, vx, vy, vyc); this->template pass_start<LOGGP, 4>(sZ, bx0, vx, vy, vyc, oA, oB); this->template pass_prime<LOGGP, 4>(Z, oA, oB);
                                    }
                                    for (;;) {
                                        const bool more = m != 0;
                                        const int lin = more ? first_of(m) : li;
                                        m &= m - 1;
                                        __builtin_amdgcn_wave_barrier();
                                        this->pf_store(pf);
                                        if (more) this->pf_issue(hpad + stepX * (c0 + lin), y0, pf);
                                        int bx0, vx, vy, vyc; unsigned mA = 0, mB = 0;
                                        if (more) { p_cand(lin, true, bx0, vx, vy, vyc); this->template pass_start<LOGGP, 4>(sZ, bx0, vx, vy, vyc, mA, mB); }
                                        __builtin_amdgcn_wave_barrier();
                                        if (more) this->template pass_run<LOGGP, 4, true>(Z, sZ, mA, mB); else this->template pass_run<LOGGP, 4, false>(Z, sZ, 0, 0);
                                        group_sum2<LOGGP>(Z.aL, Z.aC);
                                        if ((sZ == 0) & (g < 3)) *(LDS_AS v2u *)(tab + (g == 0 ? slotUp + 1 : slotZ + g) * SPEC_STRIDE + (li & (SPEC_TB - 1)) * 8) = v2u{Z.aL, Z.aC};
                                        if (!more) break;
                                        li = lin;
                                    }
                                }
                            }
                        } else {
                            // (luma-only searches and block shapes whose pieces do not divide among the lanes: one block at a time, pass by pass)
                            A4x32 pf[G::NPF];
                            this->pf_issue(hpad + stepX * (c0 + (fwd ? lo : hiE - 1)), y0, pf);
                            for (int i = 0; i < hiE - lo; i++) {
                                const int li = fwd ? lo + i : hiE - 1 - i;
                                const int blkx = c0 + li;
                                __builtin_amdgcn_wave_barrier();
                                this->pf_store(pf); // (PlaneOfBlocks.cpp:1058-1079)
                                if (i + 1 < hiE - lo) this->pf_issue(hpad + stepX * (blkx + dir), y0, pf);
                                x0 = hpad + stepX * blkx;
                                nDxMax = (pw - x0 - BW - hpad + hps) << logPel;
                                nDxMin = -((x0 - hpad + hps) << logPel);
                                const int sU = __builtin_amdgcn_readlane(pkU, li), sAh = __builtin_amdgcn_readlane(pkAh, li);
                                const int sH = __builtin_amdgcn_readlane(pkH, li), sG = __builtin_amdgcn_readlane(pkG, li);
                                const int ti8 = (li & (SPEC_TB - 1)) * 8;
                                __builtin_amdgcn_wave_barrier();
                                // (every predicate below is bitwise: no branches inside a pass; up / ahead / global / hierarchical are clipped vectors)
                                auto vok = [&](int vx, int vy) { return (vx >= nDxMin) & (vy >= nDyMin) & (vx < nDxMax) & (vy < nDyMax); };
                                const int pkZ = pk(0, fieldShift);
                                if (hexLevel) {
                                    { // hexagon + square around up, up itself, ahead: 16 candidates, 4 lanes each
                                        const int g = l >> 2, s = l & 3;
                                        const int base = g == 15 ? sAh : sU;
                                        const int vx = upx(base) + (MVX_SPEC_ABL == 2 ? 0 : rdx), vy = upy(base) + (MVX_SPEC_ABL == 2 ? 0 : rdy);
                                        const bool ok = vok(vx, vy) & ((g >= 6) | (nSearchParam > 1));
                                        unsigned aL = 0, aC = 0;
                                        if (ok) this->template eval<2>(s, vx, vy, vy, aL, aC);
                                        group_sum2<2>(aL, aC);
                                        if (s == 0) *(LDS_AS v2u *)(tab + g * SPEC_STRIDE + ti8) = v2u{aL, aC};
                                    }
                                    { // zero, global, hierarchical: 16 lanes each
                                        const int g = l >> 4, s = l & 15;
                                        const int base = g == 1 ? sG : g == 2 ? sH : pkZ;
                                        const int vx = upx(base), vy = upy(base);
                                        const int vyc = g == 0 ? 0 : vy; // the zero candidate's chroma ignores fieldShift (:836-839)
                                        unsigned aL = 0, aC = 0;
                                        if (g < (MVX_SPEC_ABL == 1 ? 0 : 3)) this->template eval<4>(s, vx, vy, vyc, aL, aC);
                                        group_sum2<4>(aL, aC);
                                        if ((s == 0) & (g < 3)) *(LDS_AS v2u *)(tab + (16 + g) * SPEC_STRIDE + ti8) = v2u{aL, aC};
                                    }
                                } else { // rings 1 and 2 around up, up, ahead, zero, global, hierarchical: 29 candidates, 2 lanes each
                                    const int g = l >> 1, s = l & 1;
                                    const int base = g == 25 ? sAh : g == 26 ? pkZ : g == 27 ? sG : g == 28 ? sH : sU;
                                    const int vx = upx(base) + rdx, vy = upy(base) + rdy;
                                    const int vyc = g == 26 ? 0 : vy;
                                    const bool ok = (vok(vx, vy) | (g == 26)) & (g < SPEC_SLOTS_EXH);
                                    unsigned aL = 0, aC = 0;
                                    if (ok) this->template eval<1>(s, vx, vy, vyc, aL, aC);
                                    group_sum2<1>(aL, aC);
                                    if ((s == 0) & (g < SPEC_SLOTS_EXH)) *(LDS_AS v2u *)(tab + g * SPEC_STRIDE + ti8) = v2u{aL, aC};
                                }
                            }
                        }
                        __builtin_amdgcn_wave_barrier();

                        SPROF(3);
                        // ======== A2: one lane per block -- predictor phase (unless stage 1 already ran it) and refinement
                        {
                            bool live = false;
                            int best = pBest, bx = pX_, by = pY_, bs = pSad;
                            if (!staged2) {
                                a2_pred(best, bx, by, bs);
                                live = !(bx == ux && by == uy); // (one block at a time: the pattern was evaluated around up)
                                live = a2_refine(ux, uy, best, bx, by, bs) || live;
                            } else {
                                live = a2_refine(pX_, pY_, best, bx, by, bs);
                                if (MVX_SPEC_ABL == 8) live = live || !(pX_ == ux && pY_ == uy); // (debug: accept only blocks whose centre is up)
                            }
                            // TEAM: everything above was independent of the walk; from here on the group needs the walk's state behind its left neighbour
                            if constexpr (TEAM) { SPROF(4); team_acquire(myG, prevX, prevY, prevSad); teamHeld = true; SPROF(17); }
                            // the bad-block rescue (:938-963) is the live search's; a higher badcount later only raises the threshold
                            live = live || (blky * nBlkX + c > 1 && (long long)bs > badSAD + badSAD * badcount / 16);
                            rX = bx; rY = by; rSad = bs;
                            staged2G = staged2; lamG = lam; pBestG = pBest; pkWG = pk(pX_, pY_);
                            // the hypothesis: my left neighbour's (speculative) result, clipped to MY limits, is my up predictor
                            int nb = __builtin_amdgcn_ds_bpermute((l - dir) << 2, pk(bx, by));
                            if (l == (fwd ? lo : hiE - 1)) nb = pk(prevX, prevY);
                            const bool hyp = cx(upx(nb)) == ux && cy(upy(nb)) == uy;
                            const bool rowStart = c == (fwd ? 0 : nBlkX - 1); // (no left neighbour: predictors[1] is the zero vector, :421-426)
                            flagmask = __ballot(act && !live && !rowStart);
                            okmask = flagmask & __ballot(hyp);
                            if (MVX_SPEC_ABL == 3) okmask = flagmask = __ballot(act);
#if MVX_SPEC_ABL == 9
                            if (SPECDBG_HERE() && act) { int *o = g_specdbg + l * SPECDBG_N; o[12] = rX; o[13] = rY; o[14] = rSad; o[15] = (int)((flagmask >> l) & 1) | ((int)((okmask >> l) & 1) << 1) | (staged2 ? 4 : 0) | (live ? 8 : 0); o[16] = pk(prevX, prevY); o[17] = best; }
#endif
                        }
                    }

                    SPROF(4);
                    if constexpr (TEAM) { if (!teamHeld) { team_acquire(myG, prevX, prevY, prevSad); SPROF(17); } } // (rows that are not speculated: every block is searched live, in turn)
                    // ======== B: verification in walk order; whatever does not verify is searched live with its true predictors
                    int pos = fwd ? lo : hiE - 1;
                    const int end = fwd ? hiE : lo - 1;
                    while (pos != end) {
                        int run;
                        if (fwd) { const unsigned long long m = ~(okmask >> pos); run = min(m ? (int)__builtin_ctzll(m) : 64, hiE - pos); }
                        else { const unsigned long long m = ~(okmask << (63 - pos)); run = min(m ? (int)__builtin_clzll(m) : 64, pos - lo + 1); }
                        if (run > 0) { // lanes pos, pos + dir, ... take their speculative results
                            const int a = fwd ? pos : pos - run + 1;
                            const bool mine = l >= a && l < a + run;
                            bOut[0] = mine ? (unsigned)rX : bOut[0]; bOut[1] = mine ? (unsigned)rY : bOut[1]; bOut[2] = mine ? (unsigned)rSad : bOut[2];
                            const int last = fwd ? a + run - 1 : a;
                            prevX = __builtin_amdgcn_readlane(rX, last); prevY = __builtin_amdgcn_readlane(rY, last); prevSad = __builtin_amdgcn_readlane(rSad, last);
#if MVX_SPEC_ABL == 9
                            if (SPECDBG_HERE() && mine) g_specdbg[l * SPECDBG_N + 18] = 1;
#endif
                            pos += dir * run;
#ifdef MVX_SPEC_STATS
                            st0 += run;
#endif
                            if (pos == end) break;
                        }
                        SPROF(5);
                        // ---- a block whose hypothesis FAILED but whose flag is set: its true left / median vectors differ from up, everything else it needs is in
                        // the table.  If both vectors are among the five the table holds (they usually are: the left neighbour ended on one of MY predictors), the
                        // predictor phase is redone here with their costs in their places (:832-915 order: zero, global, hierarchical, MEDIAN, LEFT, up, ahead); when
                        // it ends on the same vector with the same cost as under the hypothesis, the refinement that was computed for that outcome stands.
                        if (staged2G && ((flagmask >> pos) & 1)) {
                            const int li = pos, blkx = c0 + li;
                            const bool havePrev = fwd ? blkx > 0 : blkx < nBlkX - 1;
                            const int xb = stepX * blkx;
                            const int xMin = -((xb + hps) << logPel), xMax1 = ((pw - xb - hpad - BW - hpad + hps) << logPel) - 1;
                            const int sU = __builtin_amdgcn_readlane(pkU, li), sA = __builtin_amdgcn_readlane(pkAh, li), sG = __builtin_amdgcn_readlane(pkG, li), sH = __builtin_amdgcn_readlane(pkH, li);
                            const int Lx = min(max(havePrev ? prevX : 0, xMin), xMax1), Ly = this->clipy(havePrev ? prevY : fieldShift);
                            auto med = [](int a, int b, int c2) { return max(min(a, b), min(max(a, b), c2)); };
                            const int Mx = med(Lx, upx(sU), upx(sA)), My = med(Ly, upy(sU), upy(sA));
                            const int slotUpB = hexLevel ? 14 : 24, slotZB = hexLevel ? 16 : 26;
                            auto find = [&](int vx, int vy) { // the table slot that holds this vector's SADs as a PLAIN vector (the zero candidate's chroma ignores a field shift)
                                const int v = pk(vx, vy);
                                return v == sU ? slotUpB : v == sA ? slotUpB + 1 : v == sG ? slotZB + 1 : v == sH ? slotZB + 2 : (v == 0 && fieldShift == 0) ? slotZB : -1;
                            };
                            const int fL = find(Lx, Ly), fM = find(Mx, My);
                            if (fL >= 0 && fM >= 0) {
                                const lds_u8 *tcol = tab + (li & (SPEC_TB - 1)) * 8;
                                auto totS = [&](int slot) { const v2u t = *(const LDS_AS v2u *)(tcol + slot * SPEC_STRIDE); return uni((int)t[0] + (chroma ? (int)t[1] : 0)); };
                                const int lamS = __builtin_amdgcn_readlane(lamG, li);
                                auto mdS = [&](int vx, int vy) { const unsigned dx = (unsigned)(upx(sH) - vx), dy = (unsigned)(upy(sH) - vy); return (int)(((long long)lamS * (int)(dx * dx + dy * dy)) >> 8); };
                                int best, wx = 0, wy = fieldShift, ws; (void)ws;
                                { const int t = totS(slotZB); best = F::sat_add(0, t + (int)(((long long)penaltyZero * t) >> 8)); ws = t; }
                                { const int t = totS(slotZB + 1); const int cc = F::sat_add(0, t + (int)(((long long)pglobal * t) >> 8)); if (cc < best) { best = cc; wx = upx(sG); wy = upy(sG); ws = t; } }
                                { const int t = totS(slotZB + 2); const int cc = F::sat_add(0, t); if (cc < best) { best = cc; wx = upx(sH); wy = upy(sH); ws = t; } }
                                { const int t = totS(fM); const int cc = F::sat_add(mdS(Mx, My), t); if (cc < best) { best = cc; wx = Mx; wy = My; ws = t; } }
                                { const int t = totS(fL); const int cc = F::sat_add(mdS(Lx, Ly), t); if (cc < best) { best = cc; wx = Lx; wy = Ly; ws = t; } }
                                { const int t = totS(slotUpB); const int cc = F::sat_add(mdS(upx(sU), upy(sU)), t); if (cc < best) { best = cc; wx = upx(sU); wy = upy(sU); ws = t; } }
                                { const int t = totS(slotUpB + 1); const int cc = F::sat_add(mdS(upx(sA), upy(sA)), t); if (cc < best) { best = cc; wx = upx(sA); wy = upy(sA); ws = t; } }
                                if (best == __builtin_amdgcn_readlane(pBestG, li) && pk(wx, wy) == __builtin_amdgcn_readlane(pkWG, li)) { // same centre, same cost to beat: same refinement, same result
                                    const bool mine = l == li;
                                    bOut[0] = mine ? (unsigned)rX : bOut[0]; bOut[1] = mine ? (unsigned)rY : bOut[1]; bOut[2] = mine ? (unsigned)rSad : bOut[2];
                                    prevX = __builtin_amdgcn_readlane(rX, li); prevY = __builtin_amdgcn_readlane(rY, li); prevSad = __builtin_amdgcn_readlane(rSad, li);
#if MVX_SPEC_ABL == 9
                                    if (SPECDBG_HERE() && l == 0) { int *o = g_specdbg + li * SPECDBG_N; o[18] = 2; o[19] = pk(Lx, Ly); o[20] = pk(Mx, My); o[21] = fL | (fM << 8); o[22] = best; o[23] = pk(wx, wy); }
#endif
#ifdef MVX_SPEC_STATS
                                    st0 += 1; st3 += 1;
#endif
                                    const int nx = pos + dir; // (the next block's hypothesis was checked against this block's speculative result: that IS its result)
                                    (void)nx;
                                    pos += dir;
                                    continue;
                                }
                            }
                        }
                        { // ---- block `pos` live (the lean kernel's block: predictors :419-463, pobPseudoEPZSearch :819-968)
                            const int li = pos, blkx = c0 + li;
                            A4x32 sb[G::NPF];
                            this->pf_issue(hpad + stepX * blkx, y0, sb); // its source block; the scalar set-up below runs under the loads
                            blkIdx = blky * nBlkX + blkx;
                            x0 = hpad + stepX * blkx;
                            nDxMax = (pw - x0 - BW - hpad + hps) << logPel;
                            nDxMin = -((x0 - hpad + hps) << logPel);
                            const bool aheadCol = fwd ? blkx < nBlkX - 1 : blkx > 0;
                            int sfx, sfy, sfs, blx = 0, bly = 0, bls = 0, upk = 0, ups = 0;
                            { const v4u t = this->ld_batch(&vectors[blkIdx]); sfx = uni((int)t[0]); sfy = uni((int)t[1]); sfs = uni((int)t[2]); } // (still the interpolated predictor: results are stored when the group ends)
                            if (blky < nBlkY - 1 && aheadCol) { const v4u t = this->ld_batch(&vectors[(blky + 1) * nBlkX + blkx + dir]); blx = uni((int)t[0]); bly = uni((int)t[1]); bls = uni((int)t[2]); }
                            if (blky > 0) { const v2u t = rowbuf[blkx]; upk = uni((int)t[0]); ups = uni((int)t[1]); }
                            const bool useBelow = blky < nBlkY - 1 && aheadCol;
                            const bool useUpAhead = !useBelow && blky > 0 && aheadCol; // last block row only (:441-447)
                            int ahx = blx, ahy = bly, ahs = bls;
                            if (useUpAhead) {
                                const v2u t = rowbuf[blkx + dir];
                                ahx = uni(upx((int)t[0])); ahy = uni(upy((int)t[0])); ahs = uni((int)t[1]);
                            }
                            const bool haveAhead = useBelow || useUpAhead;
                            const bool havePrev = fwd ? blkx > 0 : blkx < nBlkX - 1;
                            pX[1] = this->clipx(havePrev ? prevX : 0); pY[1] = this->clipy(havePrev ? prevY : fieldShift); const int s1 = havePrev ? prevSad : 0;
                            pX[2] = this->clipx(blky > 0 ? upx(upk) : 0); pY[2] = this->clipy(blky > 0 ? upy(upk) : fieldShift); const int s2 = blky > 0 ? ups : 0;
                            pX[3] = this->clipx(haveAhead ? ahx : 0); pY[3] = this->clipy(haveAhead ? ahy : fieldShift); const int s3 = haveAhead ? ahs : 0;
                            int s0;
                            if (blky > 0) {
                                auto med = [](int a, int b, int c2) { return max(min(a, b), min(max(a, b), c2)); };
                                pX[0] = med(pX[1], pX[2], pX[3]); pY[0] = med(pY[1], pY[2], pY[3]);
                                s0 = max(s1, max(s2, s3));
                            } else { pX[0] = pX[1]; pY[0] = pY[1]; s0 = s1; }
                            int predSad;
                            if (smallestPlane) { predX = pX[0]; predY = pY[0]; predSad = s0; }
                            else { predX = this->clipx(sfx); predY = this->clipy(sfy); predSad = sfs; }
                            nLambda = 0; // row 0 searches without the motion term (:1081-1084)
                            if (blky > 0) nLambda = uni(lambda_of(predSad)); // :456-462
                            if (specRow) { const int g = __builtin_amdgcn_readlane(pkG, li); gmvx = upx(g); gmvy = upy(g); } // the running global predictor at this block
                            __builtin_amdgcn_wave_barrier();
                            this->pf_store(sb);
                            __builtin_amdgcn_wave_barrier();
                            this->search_block();
                            __builtin_amdgcn_wave_barrier();
                            { const bool mine = l == li; bOut[0] = mine ? (unsigned)bestX : bOut[0]; bOut[1] = mine ? (unsigned)bestY : bOut[1]; bOut[2] = mine ? (unsigned)bestSad : bOut[2]; }
                            prevX = bestX; prevY = bestY; prevSad = bestSad;
#if MVX_SPEC_ABL == 9
                            if (SPECDBG_HERE() && l == 0) { int *o = g_specdbg + li * SPECDBG_N; o[18] = 3; o[19] = pk(pX[1], pY[1]); o[20] = pk(pX[0], pY[0]); o[22] = bestSad; o[23] = pk(bestX, bestY); }
#endif
#ifdef MVX_SPEC_STATS
                            if (specRow) { st0 += 1; st1 += 1; st2 += !((flagmask >> li) & 1); }
#endif
                            // the next block's hypothesis was checked against this block's SPECULATIVE result: check it against the real one
                            SPROF(6);
#ifdef MVX_SPEC_PROF
                            sprof[10] += 1;
#endif
                            const int nx = pos + dir;
                            if (specRow && nx != end) {
                                const int xb = stepX * (c0 + nx);
                                const int u = __builtin_amdgcn_readlane(pkU, nx);
                                const int lx = min(max(bestX, -((xb + hps) << logPel)), ((pw - xb - hpad - BW - hpad + hps) << logPel) - 1);
                                const bool hyp = lx == upx(u) && this->clipy(bestY) == upy(u);
                                const unsigned long long bit = 1ull << nx;
                                okmask = (hyp && (flagmask & bit)) ? (okmask | bit) : (okmask & ~bit);
                            }
                            pos += dir;
                        }
                    }
                    if (specRow) gmvx = gEndX;
                    if constexpr (TEAM) { // the group's results (:967, :1106), then the token moves on
                        if (act) {
                            typedef unsigned a4v __attribute__((ext_vector_type(4), aligned(4)));
                            const a4v t = {bOut[0], bOut[1], bOut[2], 0u};
                            *(GL_AS a4v *)&vectors[blky * nBlkX + c] = t;
                            rowbuf[c] = v2u{(unsigned)pk((int)bOut[0], (int)bOut[1]), bOut[2]};
                        }
                        team_release(myG + 1, prevX, prevY, prevSad);
                    }
                    SPROF(5);
#ifdef MVX_SPEC_PROF
                    sprof[9] += 1;
#endif
                }
                // ---- results of the 64 columns (:967, :1106)
                if (!TEAM && in) {
                    typedef unsigned a4v __attribute__((ext_vector_type(4), aligned(4)));
                    const a4v t = {bOut[0], bOut[1], bOut[2], 0u}; // (block SADs are non-negative and < 2^31)
                    *(GL_AS a4v *)&vectors[blky * nBlkX + c] = t;
                    rowbuf[c] = v2u{(unsigned)pk((int)bOut[0], (int)bOut[1]), bOut[2]};
                }
            }
        }
#ifdef MVX_SPEC_STATS
        if (l == 0) { atomicAdd(&g_specstat[lvl][0], st0); atomicAdd(&g_specstat[lvl][1], st1); atomicAdd(&g_specstat[lvl][2], st2); atomicAdd(&g_specstat[lvl][3], st3); atomicAdd(&g_specstat[lvl][4], st4); atomicAdd(&g_specstat[lvl][5], st5); atomicAdd(&g_specstat[lvl][6], st6); atomicAdd(&g_specstat[lvl][7], st7); }
#endif
        SPROF(7);
        // vectors[] of this level feed the next level's interpolation / global-MV estimate (other lanes read them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_s_barrier();
    }
};

// The launch shape is analyse_fast_kernel's: workgroups of 4 * WPE chains that are consecutive entries of the (reference-sorted) job table.
// TEAM: the workgroup's waves walk ONE chain (blockDim.x / 64 of them; workgroup b = entry b of the job table).  LDS: [control words 64 B | the previous block
// row's results | one area of ldsChain bytes per wave: source strip / block, SAD table]; ldsRow carries the size of the shared part
template <int BPS, int BW, int WPE, int MAXCPW, bool UV, bool TEAM = false, bool SIDE = false>
__global__ __launch_bounds__(64 * MAXCPW, WPE) void analyse_spec_kernel(const AParams *Pp, const AJob *jobs, int njobs, int ldsChain, int syncEvery, int ldsRow, int ldsHist, int histBins, int ldsTab, int flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AParams &P = *Pp;
    const int cpw = TEAM ? 1 : (int)(blockDim.x >> 6);
    int wg = (int)blockIdx.x;
    if (flags & MVX_FAST_XCD_REMAP) { // workgroup b runs on XCD b % 8 (round-robin dispatch): give every XCD a contiguous range of the table
        const int n = (int)gridDim.x, x = wg & 7, slot = wg >> 3;
        wg = x * (n >> 3) + min(x, n & 7) + slot;
    }
    const int wave = uni((int)(threadIdx.x >> 6));
    const int chain = uni(TEAM ? wg : wg * cpw + wave);
    if (chain >= njobs) return; // (a finished wave no longer counts for the workgroup's barriers)
    const AJob &J = jobs[chain];
    if (!J.blob) return;        // padding entry of the job table
    const int l = lane_id();
    int *hdr = (int *)J.blob;
    if (TEAM && wave != 0 && !J.valid) return;
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
    if (l == 0 && (!TEAM || wave == 0)) { hdr[0] = P.blobSize; hdr[1] = 1; } // GroupOfPlanes.c:77-85
    SpecSearcher<BPS, BW, UV, (WPE <= 2 ? 24 : MVX_SPEC_SW3), TEAM, SIDE> S(P, J);
    S.lds = (lds_u8 *)smem + (TEAM ? ldsRow : 0) + wave * ldsChain;
    S.ldsRow = ldsRow; S.ldsHist = ldsHist; S.histBins = histBins; S.ldsTab = ldsTab;
    S.role = wave; S.nw = uni((int)(blockDim.x >> 6)); S.ctl = (LDS_AS int *)((lds_u8 *)smem); S.shRow = (lds_u8 *)smem + 64;
#ifdef MVX_SPEC_PROF
    for (int i = 0; i < SPROF_N; i++) S.sprof[i] = 0;
    S.sprofT = (long long)__builtin_amdgcn_s_memtime();
#endif
    int gx = 0, gy = 0; // zeroMV, MVAnalysisData.h:79
    GL_AS const GVec *coarse = nullptr;
    int cbx = 0, cby = 0, clp = 0;
    for (int lvl = P.nLevels - 1; lvl >= 0; lvl--) {
        if (coarse && P.global) S.estimate_global(coarse, cbx * cby, 8192 * P.lv[lvl + 1].pel, &gx, &gy);
        S.search_level_spec(lvl, gx, gy, coarse, cbx, cby, clp, (cpw > 1 && !TEAM) ? syncEvery : 0, !(flags & MVX_FAST_NOSPEC), !(flags & MVX_FAST_NOSTRIP));
        coarse = S.vectors; cbx = P.lv[lvl].nBlkX; cby = P.lv[lvl].nBlkY; clp = P.lv[lvl].logPel;
    }
    SPEC_PROF_DUMP_();
}
#ifdef MVX_SPEC_PROF
#define SPEC_PROF_DUMP() do { if (l == 0 && chain == 5) for (int i = 0; i < SPROF_N; i++) g_specprof[i] = (unsigned long long)S.sprof[i]; } while (0)
#else
#define SPEC_PROF_DUMP() ((void)0)
#endif
#if defined(MVX_SPEC_PROF) && defined(MVX_PROF_EXPORT)
extern "C" __attribute__((visibility("default"))) int mvx_debug_specprof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_specprof), sizeof(unsigned long long) * SPROF_N) == hipSuccess ? 0 : -1;
}
#endif
#if MVX_SPEC_ABL == 9 && defined(MVX_PROF_EXPORT)
extern "C" __attribute__((visibility("default"))) int mvx_debug_specdbg(int *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_specdbg), sizeof(int) * 64 * SPECDBG_N) == hipSuccess ? 0 : -1; }
extern "C" __attribute__((visibility("default"))) int mvx_debug_specdbg_at(int lvl, int blky, int c0) { const int v[3] = { lvl, blky, c0 }; return hipMemcpyToSymbol(HIP_SYMBOL(g_specdbgAt), v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
#if defined(MVX_SPEC_STATS) && defined(MVX_PROF_EXPORT)
extern "C" __attribute__((visibility("default"))) int mvx_debug_specstats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_specstat), sizeof(unsigned long long) * MVX_MAX_LEVELS * 8) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[MVX_MAX_LEVELS * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_specstat), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// L.ldsRow = offset of the row buffer (8 bytes per block), L.ldsHist = offset of the histogram (lies over row buffer and table: it is
// only used between levels), L.ldsBytes = offset of the table when L.ldsNeed carries the chain's total
struct ASpecLaunch { ALaunch L; int ldsTab; int team; int side; }; // side: 16x16 blocks side by side (overlap 0): the SIDE builds // team: 0 = one wave per chain, n = the workgroup's n waves walk one chain
template <int BPS, int BW, int WPE, int MAXCPW, bool UV, bool SIDE = false> static int launch_analyse_spec_uv(const ASpecLaunch &S) {
    const ALaunch &L = S.L;
    const int perChain = (L.ldsNeed + 255) & ~255;
    const int cpw = L.cpw < MAXCPW ? L.cpw : MAXCPW;
    int lds = perChain * cpw;
    if (L.ldsBytes > lds && L.ldsBytes <= 160 * 1024) lds = L.ldsBytes; // developer / host option: fewer workgroups per CU
    if (lds > 64 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void *)analyse_spec_kernel<BPS, BW, WPE, MAXCPW, UV, false, SIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((analyse_spec_kernel<BPS, BW, WPE, MAXCPW, UV, false, SIDE>), dim3((L.njobs + cpw - 1) / cpw), dim3(64 * cpw), lds, L.st, L.dP, L.dJobs,
                       L.njobs, perChain, L.syncEvery, L.ldsRow, L.ldsHist, L.histBins, S.ldsTab, L.flags);
    return MVX_OK;
}
template <int BPS, int BW, int WPE, int MAXCPW, bool SIDE = false> static int launch_analyse_spec(const ASpecLaunch &S) {
    if (S.L.flags & MVX_FAST_UV) return launch_analyse_spec_uv<BPS, BW, WPE, MAXCPW, true, SIDE>(S);
    return launch_analyse_spec_uv<BPS, BW, WPE, MAXCPW, false, SIDE>(S);
}
// TEAM: S.team waves per chain, one chain per workgroup; S.L.ldsRow = the shared part (control words + row buffer), S.L.ldsNeed = one wave's own area,
// S.ldsTab / S.L.ldsHist = offsets of the SAD table / the histogram inside a wave's area
template <int BPS, int BW, int WPE, int MAXCPW, bool UV, bool SIDE = false> static int launch_analyse_spec_team_uv(const ASpecLaunch &S) {
    const ALaunch &L = S.L;
    const int nw = S.team < MAXCPW ? S.team : MAXCPW;
    const int perWave = (L.ldsNeed + 255) & ~255;
    int lds = L.ldsRow + perWave * nw;
    if (L.ldsBytes > lds && L.ldsBytes <= 160 * 1024) lds = L.ldsBytes;
    if (lds > 64 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void *)analyse_spec_kernel<BPS, BW, WPE, MAXCPW, UV, true, SIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((analyse_spec_kernel<BPS, BW, WPE, MAXCPW, UV, true, SIDE>), dim3(L.njobs), dim3(64 * nw), lds, L.st, L.dP, L.dJobs,
                       L.njobs, perWave, 0, L.ldsRow, L.ldsHist, L.histBins, S.ldsTab, L.flags);
    return MVX_OK;
}
template <int BPS, int BW, int WPE, int MAXCPW, bool SIDE = false> static int launch_analyse_spec_team(const ASpecLaunch &S) {
    if (S.L.flags & MVX_FAST_UV) return launch_analyse_spec_team_uv<BPS, BW, WPE, MAXCPW, true, SIDE>(S);
    return launch_analyse_spec_team_uv<BPS, BW, WPE, MAXCPW, false, SIDE>(S);
}

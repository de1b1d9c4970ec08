// developer microbenchmark: what one wave-level global_load_dwordx4 costs the CU's texture-addresser / L1 path on gfx950 as a
// function of how the 64 lanes' 16-byte pieces are spread over cache lines (all lines L1 / L2 resident).  It answers the design
// questions of the search kernel: is the cost per instruction, per distinct 128-byte line, or per lane?  do lanes that are not
// neighbours but touch the same line merge?  do masked-off lanes cost anything?
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/ta_pattern.hip -o /tmp/ta_pattern ; run: /tmp/ta_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned uv4 __attribute__((ext_vector_type(4), aligned(2)));

// pattern -> byte offset of this lane's piece; PITCH = one 4K 16-bit luma row (7744 B, not a multiple of 128)
#define PITCH 7744
__device__ __forceinline__ long long lane_off(int pat, int l, bool &active) {
    active = true;
    switch (pat) {
    case 0: return (long long)l * 16;                                                  // fully coalesced: 8 lines
    case 1: return (long long)(l >> 1) * PITCH + (l & 1) * 16;                         // 32 rows x 32 B, 16-B aligned
    case 2: return (long long)(l >> 1) * PITCH + (l & 1) * 16 + 6;                     // same, 2-byte aligned start
    case 3: return (long long)l * PITCH + 6;                                           // 64 rows x 16 B (chroma-like)
    case 4: { const int c = l >> 2, s = l & 3; return (long long)((s >> 1) + (c % 3)) * PITCH + (s & 1) * 16 + (c / 3) * 2; } // 16 candidates x 2 rows x 32 B on 4 rows: same lines from lanes that are not neighbours
    case 5: active = l < 32; return (long long)(l >> 1) * PITCH + (l & 1) * 16 + 6;   // pattern 2, upper half masked off
    case 6: active = (l & 2) == 0; return (long long)(l >> 1) * PITCH + (l & 1) * 16 + 6; // pattern 2, every other lane pair off
    case 7: return (long long)(l >> 2) * PITCH + (l & 3) * 16 + 6;                     // 16 rows x 64 B
    case 8: return (long long)(l >> 3) * PITCH + (l & 7) * 16;                         // 8 rows x 128 B (full lines)
    case 9: { const int c = l >> 3, s = l & 7; return (long long)(s >> 1) * PITCH + (s & 1) * 16 + c * 4; } // 8 candidates on the SAME 4 rows (x offsets 2 px apart)
    case 10: { const int c = l >> 3, s = l & 7; return (long long)((s >> 1) * 8 + c) * PITCH + (s & 1) * 16 + 6; } // 8 candidates x 4 rows, all different rows (same as 2, other order)
    // r3 (for the next round's layout decision: the lean search kernel runs at 0.77 L1 accesses per cycle and CU, i.e. at the tag rate --
    // DESIGN.md 4.2.3 -- so what matters is how many accesses one load instruction becomes): the same 32 x 32 B pieces with the rows of a
    // block packed closer than the plane's pitch, as a strip-wise copy of the reference planes would have them
    case 11: return (long long)(l >> 1) * 64 + (l & 1) * 16;                           // row pitch 64 B: two rows per 128-byte line, 32 B used of every 64
    case 12: return (long long)(l >> 1) * 64 + (l & 1) * 16 + 16;                      // same, the block sits at byte 16 of its 64-byte rows
    case 13: return (long long)(l >> 1) * 128 + (l & 1) * 16;                          // row pitch 128 B: one row per line, neighbouring lines
    case 14: return (long long)(l >> 2) * PITCH + (l & 3) * 16;                        // 16 rows x 64 B, 16-byte aligned (7 is its 2-byte-aligned form)
    case 15: return (long long)(l >> 1) * 32 + (l & 1) * 16;                           // row pitch 32 B = pattern 0 written as rows (control)
    // r4: which alignment does a 16-byte piece need?  (the search's row passes read 128-byte strips at dword-aligned addresses -- shadow planes --,
    // Degrain 16-byte rows at 2-byte-aligned ones)
    case 16: return (long long)(l >> 3) * PITCH + (l & 7) * 16 + 4;                    // 8 rows x 128 B, dword-aligned
    case 17: return (long long)(l >> 3) * PITCH + (l & 7) * 16 + 2;                    // 8 rows x 128 B, 2-byte-aligned
    case 18: return (long long)(l >> 3) * PITCH + (l & 7) * 16 + 8;                    // 8 rows x 128 B, 8-byte-aligned
    case 19: return (long long)(l >> 1) * PITCH + (l & 1) * 16 + 4;                    // 32 rows x 32 B, dword-aligned
    case 20: return (long long)(l >> 1) * PITCH + (l & 1) * 16 + 8;                    // 32 rows x 32 B, 8-byte-aligned
    case 21: return (long long)l * 16 + 4;                                             // contiguous, dword-aligned
    case 22: return (long long)l * 16 + 2;                                             // contiguous, 2-byte-aligned
    case 23: return (long long)(l >> 3) * PITCH + (l & 7) * 16 + 64;                   // 8 rows x 128 B starting in the middle of a line (aligned, two lines per row)
    default: return 0;
    }
}

template <int NINFLIGHT>
__global__ __launch_bounds__(256) void k(const unsigned char *buf, unsigned long long *out, int pat, int iters, long long waveStride) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool active;
    const unsigned char *p = buf + ((long long)blockIdx.x * 4 + w) * waveStride + lane_off(pat, l, active);
    v4u acc = {0, 0, 0, 0};
    // warm the lines
    if (active) { uv4 t = *(const uv4 *)p; acc += v4u{t[0], t[1], t[2], t[3]}; }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        v4u r[NINFLIGHT];
#pragma unroll
        for (int k2 = 0; k2 < NINFLIGHT; k2++) {
            const unsigned char *q = p + (k2 & 1) * 0; // same lines every time: L1 hits
            asm volatile("" : "+v"(q));
            if (active) { uv4 t = *(const uv4 *)q; r[k2] = v4u{t[0], t[1], t[2], t[3]}; } else r[k2] = v4u{0, 0, 0, 0};
        }
#pragma unroll
        for (int k2 = 0; k2 < NINFLIGHT; k2++) acc += r[k2];
    }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) out[blockIdx.x * 4 + w] = (unsigned long long)(t1 - t0);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345) out[0] = 0;
}

int main() {
    const long long waveStride = 64LL * PITCH + 4096; // every wave its own rows
    const int maxBlocks = 256 * 2;
    unsigned char *buf; hipMalloc(&buf, waveStride * 4 * maxBlocks + (1 << 20)); hipMemset(buf, 1, waveStride * 4 * maxBlocks + (1 << 20));
    unsigned long long *d; hipMalloc(&d, 8 * 4 * maxBlocks);
    const int iters = 2000, NIF = 4;
    const char *names[] = {"0 coalesced 64x16B contiguous (8 lines)", "1 32 rows x 32B aligned", "2 32 rows x 32B, 2B-aligned", "3 64 rows x 16B", "4 16 cand x2 rows on 4 rows (non-neighbour merge)",
                           "5 pattern 2, lanes 32-63 off", "6 pattern 2, alternate lane pairs off", "7 16 rows x 64B", "8 8 rows x 128B (full lines)", "9 8 cand on the same 4 rows", "10 pattern 2 rows, candidate-major order",
                           "11 32 rows x 32B, row pitch 64B (2 rows per line)", "12 same, block at byte 16 of its rows", "13 32 rows x 32B, row pitch 128B", "14 16 rows x 64B aligned", "15 32 rows x 32B, row pitch 32B (= 0)",
                           "16 8 rows x 128B, dword-aligned (+4)", "17 8 rows x 128B, 2-byte-aligned (+2)", "18 8 rows x 128B, 8-byte-aligned (+8)", "19 32 rows x 32B, dword-aligned (+4)",
                           "20 32 rows x 32B, 8-byte-aligned (+8)", "21 contiguous, dword-aligned (+4)", "22 contiguous, 2-byte-aligned (+2)", "23 8 rows x 128B from the middle of a line (+64)"};
    for (int blocks : {256, 512}) { // one / two workgroups of four waves per CU
        for (int pat = 0; pat <= 23; pat++) {
            for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<NIF>, dim3(blocks), dim3(256), 0, 0, buf, d, pat, iters, waveStride); hipDeviceSynchronize(); }
            std::vector<unsigned long long> h(blocks * 4);
            hipMemcpy(h.data(), d, 8 * blocks * 4, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            const double perWaveInstr = s / (blocks * 4) / (iters * NIF);
            const int wavesPerCU = blocks / 256 * 4;
            printf("waves/CU %d  %-52s %7.1f cycles per load per wave -> %6.1f CU-cycles per wave-instruction\n", wavesPerCU, names[pat], perWaveInstr, perWaveInstr / wavesPerCU);
        }
    }
    return 0;
}

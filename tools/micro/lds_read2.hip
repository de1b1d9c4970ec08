// developer microbenchmark: cost of reading a 32-byte reference row at 2-byte alignment from an LDS window, 8 lanes per candidate
// (rows s and s+8), seven candidates at different window offsets -- (a) two under-aligned ds_read_b128, (b) five ds_read2_b32 +
// v_alignbit, (c) three aligned ds_read_b128 (the pieces covering the row; realignment not included)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned uv4 __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned uv2 __attribute__((ext_vector_type(2), aligned(4)));
#define LDS_AS __attribute__((address_space(3)))

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int rs, int iters, int seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int l = threadIdx.x;
    for (int i = l; i < 36000 / 4; i += 64) ((LDS_AS unsigned *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int g = l >> 3, s = l & 7;
    // candidate g: sub-pel plane g & 3 (planes 32 rows apart), row offset 0..8, byte offset 2 * (0..23)
    const int pp = g & 3, ro = (g * 3 + seed) % 9, bo = ((g * 7 + seed * 5) % 24) * 2;
    const int base = (pp * 32 + ro + s) * rs + bo;
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        int o = base + (it & 1) * 2;
        asm volatile("" : "+v"(o) :: "memory");
#pragma unroll
        for (int rr = 0; rr < 2; rr++) { // rows s and s + 8
            const int a = o + rr * 8 * rs;
            if (MODE == 0) {
                uv4 b0 = *(const LDS_AS uv4 *)(lds + a), b1 = *(const LDS_AS uv4 *)(lds + a + 16);
                acc = __builtin_amdgcn_sad_u16(b0[0], b0[1], acc); acc = __builtin_amdgcn_sad_u16(b0[2], b0[3], acc);
                acc = __builtin_amdgcn_sad_u16(b1[0], b1[1], acc); acc = __builtin_amdgcn_sad_u16(b1[2], b1[3], acc);
            } else if (MODE == 1) {
                const int a4 = a & ~3, sh = (a & 2) * 8;
                unsigned d[10];
#pragma unroll
                for (int i = 0; i < 5; i++) { uv2 t = *(const LDS_AS uv2 *)(lds + a4 + 8 * i); d[2 * i] = t[0]; d[2 * i + 1] = t[1]; }
#pragma unroll
                for (int i = 0; i < 8; i += 2)
                    acc = __builtin_amdgcn_sad_u16(__builtin_amdgcn_alignbit(d[i + 1], d[i], sh), __builtin_amdgcn_alignbit(d[i + 2], d[i + 1], sh), acc);
            } else {
                const int a16 = a & ~15;
                v4u b0 = *(const LDS_AS v4u *)(lds + a16), b1 = *(const LDS_AS v4u *)(lds + a16 + 16), b2 = *(const LDS_AS v4u *)(lds + a16 + 32);
                acc = __builtin_amdgcn_sad_u16(b0[0], b0[1], acc); acc = __builtin_amdgcn_sad_u16(b0[2], b0[3], acc);
                acc = __builtin_amdgcn_sad_u16(b1[0], b1[1], acc); acc = __builtin_amdgcn_sad_u16(b1[2], b1[3], acc);
                acc = __builtin_amdgcn_sad_u16(b2[0], b2[1], acc); acc = __builtin_amdgcn_sad_u16(b2[2], b2[3], acc);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (acc == 0x12345) out[blockIdx.x] = 0;
}

int main() {
    unsigned long long *d; hipMalloc(&d, 8 * 4096);
    const int iters = 4096;
    auto run = [&](const char *name, int mode, int rs, int blocks, int seed) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 40960, 0, d, rs, iters, seed);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 40960, 0, d, rs, iters, seed);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 40960, 0, d, rs, iters, seed);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("%-40s rs %3d seed %d blocks %4d: %7.1f cycles per pair of rows (one wave-pass = 7 candidates x 2 rows x 32 B)\n", name, rs, seed, blocks, s / blocks / iters);
    };
    for (int blocks : {256, 1024})
        for (int rs : {144, 272})
            for (int seed : {0, 3}) {
                run("2x under-aligned ds_read_b128", 0, rs, blocks, seed);
                run("5x ds_read2_b32 + alignbit", 1, rs, blocks, seed);
                run("3x aligned ds_read_b128 (no realign)", 2, rs, blocks, seed);
            }
    return 0;
}

// developer microbenchmark (next round): does a buffer load whose offset lies beyond num_records cost the CU's L1 path
// anything?  If not, idle lanes of a candidate-evaluation pass can be switched off per lane without branches (the fused predictor + hexagon
// pass needed exactly that: predicated global loads split it into basic blocks and blew the register budget; unpredicated ones loaded with
// every lane).  Pattern: 32 rows x 32 B per wave instruction (the search kernel's), k of the 64 lanes in range.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/buffer_pred.hip -o /tmp/buffer_pred ; run: /tmp/buffer_pred
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ v4i raw_buffer_load_v4(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
#define PITCH 7744

template <int NIF>
__global__ __launch_bounds__(256) void k(const unsigned char *buf, unsigned long long *out, int lanesOn, int mode, int iters, long long waveStride) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned char *base = buf + ((long long)blockIdx.x * 4 + w) * waveStride;
    // raw buffer: base, stride 0, num_records = the wave's region, gfx9 dword 3
    const unsigned long long b = (unsigned long long)base;
    v4i rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)waveStride);
    rsrc[3] = 0x00020000;
    int off = (l >> 1) * PITCH + (l & 1) * 16;
    const bool on = l < lanesOn;
    if (mode == 1 && !on) off = 0x7ffffff0;              // out of range: returns zeros
    v4i acc = {0, 0, 0, 0};
    { const v4i t = raw_buffer_load_v4(rsrc, off, 0, 0); acc += t; } // warm
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        v4i r[NIF];
#pragma unroll
        for (int k2 = 0; k2 < NIF; k2++) {
            int o = off; asm volatile("" : "+v"(o));
            if (mode == 2) { if (on) r[k2] = raw_buffer_load_v4(rsrc, o, 0, 0); else r[k2] = v4i{0, 0, 0, 0}; } // exec-masked for comparison
            else r[k2] = raw_buffer_load_v4(rsrc, o, 0, 0);
        }
#pragma unroll
        for (int k2 = 0; k2 < NIF; k2++) acc += r[k2];
    }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) out[blockIdx.x * 4 + w] = (unsigned long long)(t1 - t0);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345) out[0] = 0;
}

int main() {
    const long long waveStride = 64LL * PITCH + 4096;
    const int blocks = 512; // two workgroups of four waves per CU
    unsigned char *buf; hipMalloc(&buf, waveStride * 4 * blocks + (1 << 20)); hipMemset(buf, 1, waveStride * 4 * blocks + (1 << 20));
    unsigned long long *d; hipMalloc(&d, 8 * 4 * blocks);
    const int iters = 2000, NIF = 4;
    const char *modes[] = {"all lanes in range (lanesOn ignored)", "lanes >= lanesOn out of range", "lanes >= lanesOn masked off (exec)"};
    for (int mode = 0; mode < 3; mode++)
        for (int lanesOn : {64, 48, 32, 16}) {
            if (mode == 0 && lanesOn != 64) continue;
            for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<NIF>, dim3(blocks), dim3(256), 0, 0, buf, d, lanesOn, mode, iters, waveStride); hipDeviceSynchronize(); }
            std::vector<unsigned long long> h(blocks * 4);
            hipMemcpy(h.data(), d, 8 * blocks * 4, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            const double per = s / (blocks * 4) / (iters * NIF);
            printf("%-40s lanes on %2d: %7.1f cycles per load per wave -> %6.1f CU-cycles per wave-instruction (8 waves per CU)\n", modes[mode], lanesOn, per, per / 8);
        }
    return 0;
}

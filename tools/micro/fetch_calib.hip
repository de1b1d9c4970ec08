// developer microbenchmark: what FETCH_SIZE counts for the search kernels' load patterns.  Each pattern reads a buffer far larger
// than the 256 MiB Infinity Cache exactly ONCE (every 128-byte line of the region is touched once), with 16 bytes per lane as the kernels do:
//   0  fully coalesced (64 lanes x 16 B = 8 whole lines per load): the guide's calibration case
//   1  row passes of the speculative kernel: 8 lanes read 128 contiguous bytes that start 4 bytes into a line (two lines per group), 8 groups per load
//   2  the serial lean kernel / one-block passes: a lane pair reads 32 bytes at byte 4 of its own line (32 lines per load, a quarter of each used)
// run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -o p -- /tmp/micro_fetch_calib
// and compare the kernel rows' FETCH_SIZE with the printed true bytes (region size = lines touched x 128).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned uv4 __attribute__((ext_vector_type(4), aligned(4)));
template <int PAT> __global__ __launch_bounds__(256) void k(const unsigned char *buf, unsigned *out, size_t bytes) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    const int l = threadIdx.x & 63;
    v4u acc = {0, 0, 0, 0};
    // a wave-load covers LINES lines; waves stride over the region so that every line is read once
    constexpr size_t LINES = PAT == 0 ? 8 : PAT == 1 ? 16 : 32;
    const size_t nloads = bytes / (LINES * 128);
    for (size_t i = wave; i < nloads; i += nwaves) {
        const unsigned char *p = buf + i * LINES * 128;
        size_t off;
        if (PAT == 0) off = (size_t)l * 16;
        else if (PAT == 1) off = (size_t)(l >> 3) * 256 + 4 + (size_t)(l & 7) * 16;   // group g: lines 2g, 2g + 1
        else off = (size_t)(l >> 1) * 128 + 4 + (size_t)(l & 1) * 16;                  // pair k: line k
        const uv4 t = *(const uv4 *)(p + off);
        acc += v4u{t[0], t[1], t[2], t[3]};
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) out[0] = 1; // (keeps the loads)
}
int main() {
    const size_t bytes = (size_t)12 << 30;
    unsigned char *buf; unsigned *out;
    if (hipMalloc(&buf, bytes + 4096) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes + 4096);
    (void)hipDeviceSynchronize();
    for (int pat = 0; pat < 3; pat++) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, buf, out, bytes);
        if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, buf, out, bytes);
        if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(4096), dim3(256), 0, 0, buf, out, bytes);
        (void)hipEventRecord(b); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("pattern %d: region %.3f GB (every 128-byte line once), bytes the lanes asked for %.3f GB, %.2f ms = %.2f TB/s of lines\n", pat, bytes / 1e9,
               (pat == 0 ? 1.0 : pat == 1 ? 0.5 : 0.25) * bytes / 1e9, ms, bytes / 1e9 / ms);
    }
    return 0;
}

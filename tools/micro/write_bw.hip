// developer microbenchmark: what HBM write bandwidth do streaming kernels reach on MI355X, as a function of the store pattern?
// mv.Super, the shadow planes and Degrain are write-heavy streaming kernels; this measures the ceilings they are judged against.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/write_bw.hip -o /tmp/write_bw ; run: /tmp/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 1: linear fill, 16 B per thread
__global__ __launch_bounds__(256) void fill_linear(v4u *d, long long n16, int nt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const v4u v = { (unsigned)i, 1u, 2u, 3u };
    if (nt) __builtin_nontemporal_store(v, d + i); else d[i] = v;
}
// 2: linear copy
__global__ __launch_bounds__(256) void copy_linear(const v4u *s, v4u *d, long long n16, int nt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const v4u v = s[i];
    if (nt) __builtin_nontemporal_store(v, d + i); else d[i] = v;
}
// 3: linear read (sum kept alive through a rare store)
__global__ __launch_bounds__(256) void read_linear(const v4u *s, v4u *d, long long n16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const v4u v = s[i];
    if (v[0] == 0x12345678u && v[1] == 0x9abcdef0u) d[0] = v;
}
// 4: "rows": a frame of NPL planes of ph rows x pitch bytes; block = 64 x 4 threads, thread = 16 B of R rows of every plane
// (the store pattern of super_rows_kernel: per wave 1 KB pieces into NPL x R different rows)
__global__ __launch_bounds__(256) void fill_rows(unsigned char *d, int npl, int R, int rowBytes, int ph, long long pitch, long long frameBytes, int x0, int nt) {
    const int xb = x0 + (blockIdx.x * 64 + (threadIdx.x & 63)) * 16;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * R;
    if (xb + 16 > rowBytes) return;
    unsigned char *f = d + blockIdx.z * frameBytes;
    const v4u v = { (unsigned)xb, (unsigned)y0, 2u, 3u };
    for (int r = 0; r < R; r++) {
        if (y0 + r >= ph) break;
        for (int p = 0; p < npl; p++) {
            v4u *q = (v4u *)(f + (long long)p * ph * pitch + (long long)(y0 + r) * pitch + xb);
            if (nt) __builtin_nontemporal_store(v, q); else *q = v;
        }
    }
}
// 5: "rows, one plane per block": blockIdx.z = frame * npl + plane, so that a wave writes 1 KB pieces of R rows of ONE plane
__global__ __launch_bounds__(256) void fill_rows_split(unsigned char *d, int npl, int R, int rowBytes, int ph, long long pitch, long long frameBytes, int x0) {
    const int xb = x0 + (blockIdx.x * 64 + (threadIdx.x & 63)) * 16;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * R;
    if (xb + 16 > rowBytes) return;
    unsigned char *f = d + (blockIdx.z / npl) * frameBytes + (long long)(blockIdx.z % npl) * ph * pitch;
    const v4u v = { (unsigned)xb, (unsigned)y0, 2u, 3u };
    for (int r = 0; r < R; r++) {
        if (y0 + r >= ph) break;
        *(v4u *)(f + (long long)(y0 + r) * pitch + xb) = v;
    }
}
// 6: thread = 16 B, a block covers 4 KB of ONE row (256 threads along x), rows in blockIdx.y, planes/frames in z
__global__ __launch_bounds__(256) void fill_rows_wide(unsigned char *d, int rowBytes, int ph, long long pitch, int x0) {
    const int xb = x0 + (blockIdx.x * 256 + threadIdx.x) * 16;
    if (xb + 16 > rowBytes) return;
    const v4u v = { (unsigned)xb, 1u, 2u, 3u };
    *(v4u *)(d + ((long long)blockIdx.z * ph + blockIdx.y) * pitch + xb) = v;
}

int main() {
    const long long bytes = 8LL << 30;
    unsigned char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long n16 = bytes / 16;
    auto timeit = [&](const char *name, double moved, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        const int reps = 5;
        for (int i = 0; i < reps; i++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %8.3f ms  %6.2f TB/s\n", name, ms / reps, moved * reps / (ms * 1e-3) / 1e12);
    };
    const unsigned gl = (unsigned)((n16 + 255) / 256);
    timeit("fill, linear 16 B/thread (8 GiB written)", (double)bytes, [&] { hipLaunchKernelGGL(fill_linear, dim3(gl), dim3(256), 0, 0, (v4u *)a, n16, 0); });
    timeit("fill, linear, nontemporal stores", (double)bytes, [&] { hipLaunchKernelGGL(fill_linear, dim3(gl), dim3(256), 0, 0, (v4u *)a, n16, 1); });
    timeit("read, linear (8 GiB read)", (double)bytes, [&] { hipLaunchKernelGGL(read_linear, dim3(gl), dim3(256), 0, 0, (const v4u *)a, (v4u *)b, n16); });
    timeit("copy, linear (8 GiB read + 8 GiB written)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_linear, dim3(gl), dim3(256), 0, 0, (const v4u *)a, (v4u *)b, n16, 0); });
    timeit("copy, linear, nontemporal stores", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_linear, dim3(gl), dim3(256), 0, 0, (const v4u *)a, (v4u *)b, n16, 1); });
    // 4K16 luma super-frame geometry: padded row 3872 samples = 7744 B, pitch 7936, 2192 rows, 4 planes (+4 shadow planes)
    const int rowBytes = 7744, ph = 2192; const long long pitch = 7936;
    for (int npl : { 1, 4, 8 }) {
        const long long frameBytes = (long long)npl * ph * pitch;
        const int nf = (int)(bytes / frameBytes);
        const double moved = (double)nf * npl * ph * rowBytes;
        for (int R : { 1, 2 }) for (int x0 : { 0, 48 }) for (int nt : { 0, 1 }) {
            char name[128]; snprintf(name, sizeof name, "rows: %d planes, %d row(s)/thread, first byte %d%s", npl, R, x0, nt ? ", nontemporal" : "");
            dim3 g((rowBytes / 16 + 63) / 64, (ph + 4 * R - 1) / (4 * R), nf);
            timeit(name, moved, [&] { hipLaunchKernelGGL(fill_rows, g, dim3(256), 0, 0, a, npl, R, rowBytes, ph, pitch, frameBytes, x0, nt); });
        }
        { char name[128]; snprintf(name, sizeof name, "rows: %d planes, one plane per block, 2 rows/thread", npl);
          dim3 g((rowBytes / 16 + 63) / 64, (ph + 7) / 8, nf * npl);
          timeit(name, moved, [&] { hipLaunchKernelGGL(fill_rows_split, g, dim3(256), 0, 0, a, npl, 2, rowBytes, ph, pitch, frameBytes, 0); }); }
    }
    { const int nrows = (int)(bytes / pitch) / ph; // "frames" of one plane
      dim3 g((rowBytes / 16 + 255) / 256, ph, nrows);
      timeit("rows: a block = 4 KB of one row", (double)nrows * ph * rowBytes, [&] { hipLaunchKernelGGL(fill_rows_wide, g, dim3(256), 0, 0, a, rowBytes, ph, pitch, 0); }); }
    return 0;
}

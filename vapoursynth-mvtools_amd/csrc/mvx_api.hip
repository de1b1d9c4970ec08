// mvx_api.hip -- device/runtime helpers of the C ABI (no pixel work here).
#include "mvx_common.h"

extern "C" __attribute__((visibility("default"))) const char *mvx_version(void) { return "mvtools_amd 0.1 (gfx950; mvtools v24 hot path)"; }

extern "C" __attribute__((visibility("default"))) int mvx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { mvx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return MVX_E_DEVICE; }
    return n;
}

extern "C" __attribute__((visibility("default"))) int mvx_set_device(int ordinal) { HIP_CHECK(hipSetDevice(ordinal)); return MVX_OK; }

extern "C" __attribute__((visibility("default"))) void *mvx_dev_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { mvx_set_error("hipMalloc(%zu) failed", bytes); return nullptr; }
    if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); mvx_set_error("hipMemset failed"); return nullptr; }
    return p;
}
extern "C" __attribute__((visibility("default"))) void mvx_dev_free(void *p) { if (p) (void)hipFree(p); }

extern "C" __attribute__((visibility("default"))) int mvx_copy_to_device(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t row_bytes, size_t rows, void *stream) {
    HIP_CHECK(hipMemcpy2DAsync(dst, dp, src, sp, row_bytes, rows, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) int mvx_copy_to_host(void *dst, ptrdiff_t dp, const void *src, ptrdiff_t sp, size_t row_bytes, size_t rows, void *stream) {
    HIP_CHECK(hipMemcpy2DAsync(dst, dp, src, sp, row_bytes, rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MVX_OK;
}
extern "C" __attribute__((visibility("default"))) int mvx_stream_sync(void *stream) { HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); return MVX_OK; }

// Diagnostic: which physical CUs does a stream created with hipExtStreamCreateWithCUMask use on this device?
//   hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/native/cumask_probe.hip -o tools/native/cumask_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_probe(unsigned *out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));    // XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
extern "C" int cumask_probe(const uint32_t *mask, int words, int nwg, int spin, unsigned *host_out)
{
    hipStream_t st;
    if (mask) { if (hipExtStreamCreateWithCUMask(&st, words, mask) != hipSuccess) return -1; }
    else if (hipStreamCreate(&st) != hipSuccess) return -1;
    unsigned *d;
    if (hipMalloc(&d, sizeof(unsigned) * 2 * nwg) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(64), 0, st, d, spin);
    if (hipStreamSynchronize(st) != hipSuccess) return -3;
    hipMemcpy(host_out, d, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost);
    hipFree(d); hipStreamDestroy(st);
    return 0;
}

// Is hipFuncSetAttribute(MaxDynamicSharedMemorySize) legal while a stream is being captured, and does the captured launch
// with > 64 KB of dynamic LDS replay correctly?  (K3's sweep kernels raise the limit lazily, at an instantiation's first
// launch, which may be a captured one.)   hipcc --offload-arch=gfx950 -O2 -o capture_attr_test capture_attr_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k(float* out) {
    extern __shared__ float s[];
    s[threadIdx.x + 20000] = (float)threadIdx.x;      // touches LDS beyond 64 KB
    __syncthreads();
    out[threadIdx.x] = s[(threadIdx.x ^ 1) + 20000] + 1.0f;
}
int main() {
    float* d; CK(hipMalloc(&d, 1024));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    printf("hipFuncSetAttribute during capture: %s\n", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 100 * 1024, st, d);
    printf("launch during capture: %s\n", hipGetErrorString(hipGetLastError()));
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float h[256]; CK(hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += h[i] != (float)(i ^ 1) + 1.0f;
    printf("replay: %d wrong of 256\n", bad);
    return bad != 0;
}

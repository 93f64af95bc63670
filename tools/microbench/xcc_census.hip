// Which XCD runs workgroup k?  Prints the XCC id of the first 32 workgroups and checks the "k % 8" rule.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void census(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x & 0xF;
    }
}
int main() {
    const int n = 65536;
    unsigned* d; hipMalloc(&d, n * 4);
    for (int bd : {64, 256}) {
        census<<<n, bd>>>(d);
        std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        printf("BD=%d XCC of WG 0..31:", bd); for (int i = 0; i < 32; ++i) printf(" %u", h[i]); printf("\n");
        int same = 0; for (int i = 8; i < n; ++i) same += (h[i] == h[i - 8]);
        int hist[16] = {0}; for (int i = 0; i < n; ++i) hist[h[i]]++;
        printf("  WG k and k-8 on same XCC: %d / %d; per-XCC counts:", same, n - 8); for (int i = 0; i < 8; ++i) printf(" %d", hist[i]); printf("\n");
    }
    return 0;
}

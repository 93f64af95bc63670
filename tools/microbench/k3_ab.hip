// Same-process, interleaved A/B of the product's K3 sweep instantiations (vector stores / 64-apart dword layout, two / four
// columns per lane) on one output buffer: rounds of back-to-back launches, variants alternating, mean and spread per variant.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -o tools/microbench/k3_ab tools/microbench/k3_ab.hip
#include "../../protstruc_amd/csrc/pairwise_angles.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6, per = argc > 2 ? atoi(argv[2]) : 20;
    const int A = 15;
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    for (int N : {512, 384, 256}) {
        const int B = (1 << 25) / (N * N);
        std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> h((size_t)B * N * A * 3); for (auto& x : h) x = nd(rng);
        float *xyz, *out; const size_t ob = (size_t)B * N * N * 4;
        CK(hipMalloc(&xyz, h.size() * 4)); CK(hipMalloc(&out, ob));
        CK(hipMemcpy(xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        struct V { std::string name; std::function<int()> fn; std::vector<float> us; };
        auto bench = [&](const char* title, std::vector<V> vs) {
            for (auto& v : vs) { if (v.fn()) { printf("launch error %s\n", v.name.c_str()); return; } }
            CK(hipDeviceSynchronize());
            for (int r = 0; r < rounds; ++r)
                for (auto& v : vs) {
                    v.fn();
                    CK(hipEventRecord(ea)); for (int k = 0; k < per; ++k) v.fn(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
                    float ms; CK(hipEventElapsedTime(&ms, ea, eb)); v.us.push_back(ms * 1e3f / per);
                }
            printf("N=%d B=%d %s:", N, B, title);
            for (auto& v : vs) {
                std::sort(v.us.begin(), v.us.end());
                float m = 0; for (float x : v.us) m += x; m /= v.us.size();
                printf("  %s %.1f [%.1f..%.1f]", v.name.c_str(), m, v.us.front(), v.us.back());
            }
            printf("\n");
        };
        AtomSel s22{{1, 4, 1, 4}}, s31{{0, 1, 4, 4}}, sp{{1, 4, 4, 0}};
#define L(NPv, SRCv, NCv, VECv, sel) [&] { return launch_sweep<NPv, SRCv, NCv, VECv>(xyz, out, B, N, A, sel, 0, N, N, 0, nullptr); }
        bench("dihedral (2,2)", {{"nc4 vec", L(4, 12, 4, true, s22)}, {"nc4 dword", L(4, 12, 4, false, s22)}, {"nc2 vec", L(4, 12, 2, true, s22)}, {"nc2 dword", L(4, 12, 2, false, s22)}});
        bench("dihedral (3,1)", {{"nc4 vec", L(4, 8, 4, true, s31)}, {"nc4 dword", L(4, 8, 4, false, s31)}, {"nc2 vec", L(4, 8, 2, true, s31)}, {"nc2 dword", L(4, 8, 2, false, s31)}});
        bench("planar (2,1)", {{"nc4 vec", L(3, 4, 4, true, sp)}, {"nc4 dword", L(3, 4, 4, false, sp)}, {"nc2 vec", L(3, 4, 2, true, sp)}, {"nc2 dword", L(3, 4, 2, false, sp)}});
        CK(hipFree(xyz)); CK(hipFree(out));
    }
    return 0;
}

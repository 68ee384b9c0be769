// Probe the lane layout of v_mfma_f64_4x4x4_4b (A, B, D) and the cbsz/abid broadcast on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID> __global__ void probe(double *D) {
    // block (la, lb): A one-hot at lane la, B one-hot at lane lb
    const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, CBSZ, ABID, 0);
    D[(size_t)blockIdx.x * 64 + lane] = c;
}
template <int CBSZ, int ABID> void run(const char *name) {
    double *dD;
    hipMalloc(&dD, sizeof(double) * 4096 * 64);
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(4096), dim3(64), 0, 0, dD);
    std::vector<double> h(4096 * 64);
    hipMemcpy(h.data(), dD, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    printf("== %s: for each A lane la: list of (lb -> D lanes)\n", name);
    for (int la = 0; la < 64; ++la) {
        printf("la=%2d:", la);
        int shown = 0;
        for (int lb = 0; lb < 64; ++lb) {
            bool any = false;
            for (int l = 0; l < 64; ++l) any = any || h[((size_t)la * 64 + lb) * 64 + l] != 0.0;
            if (!any) continue;
            printf(" lb%d->[", lb);
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf("%d,", l);
            printf("]");
            if (++shown >= 16) break;
        }
        printf("\n");
        if (CBSZ == 0 && la >= 20 && la < 60) { if (la == 20) printf(" ...\n"); continue; }
    }
    hipFree(dD);
}
int main() {
    run<0, 0>("cbsz=0 abid=0");
    run<2, 0>("cbsz=2 abid=0");
    run<2, 1>("cbsz=2 abid=1");
    run<1, 0>("cbsz=1 abid=0");
    return 0;
}

// Issue cost of v_rsq_f64 against an f32 seed (v_cvt_f32_f64 + v_rsq_f32 + v_cvt_f64_f32) and against v_fma_f64 on gfx950:
// cycles per wave instruction, 8 independent chains per lane, one wave per SIMD.   hipcc --offload-arch=gfx950 -O3 -o rsq_rate rsq_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(double *out, int iters, unsigned long long *cyc) {
    double v[8];
    for (int q = 0; q < 8; ++q) v[q] = 1.0 + threadIdx.x * 1e-3 + q;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) v[q] = __builtin_amdgcn_rsq(v[q]) + 1.0;
            else if (MODE == 1) v[q] = (double)__builtin_amdgcn_rsqf((float)v[q]) + 1.0;
            else v[q] = fma(v[q], 0.999999, 1.0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int q = 0; q < 8; ++q) s += v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char *names[3] = {"v_rsq_f64 (+ v_add_f64)", "cvt + v_rsq_f32 + cvt (+ v_add_f64)", "v_fma_f64"};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on this part: report relative numbers
        std::printf("%-40s %.2f memtime ticks per 1000 loop bodies (8 ops each)\n", names[m], (double)h / iters * 1000.0);
    }
    return 0;
}

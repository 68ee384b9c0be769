// FP64 throughput microbenchmarks on gfx950: v_mfma_f64_16x16x4 vs v_mfma_f64_4x4x4 vs v_fma_f64,
// with in-kernel clock measurement (s_memtime / s_memrealtime).  Development aid for DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC> __global__ __launch_bounds__(256) void mfma16(double *sink, int iters, unsigned long long *st) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    v4f64 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = (v4f64){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        st[2 * w] = t1 - t0;
        st[2 * w + 1] = r1 - r0;
    }
}

template <int NACC> __global__ __launch_bounds__(256) void mfma4(double *sink, int iters, unsigned long long *st) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        st[2 * w] = t1 - t0;
        st[2 * w + 1] = r1 - r0;
    }
}

template <int NACC> __global__ __launch_bounds__(256) void vfma(double *sink, int iters, unsigned long long *st) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
    double c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_fma(c[i], a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        st[2 * w] = t1 - t0;
        st[2 * w + 1] = r1 - r0;
    }
}

// sqrt / rsq cost probes
template <int MODE> __global__ __launch_bounds__(256) void vsqrt(double *sink, int iters, unsigned long long *st) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.5 + i + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = sqrt(x[i]) + 1.0;
            if (MODE == 1) x[i] = __builtin_amdgcn_rsq(x[i]) + 1.0;
            if (MODE == 2) x[i] = log(x[i]) + 2.0;
            if (MODE == 3) x[i] = 1.0 / x[i] + 1.0;
            if (MODE == 4) x[i] = __builtin_amdgcn_sqrt(x[i]) + 1.0;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        st[2 * w] = t1 - t0;
        st[2 * w + 1] = r1 - r0;
    }
}

template <class K> void run(const char *name, K kernel, int blocks, int threads, int iters, double ops_per_iter_per_wave,
                            double flops_per_op) {
    double *sink;
    unsigned long long *st;
    hipMalloc(&sink, 8);
    const int waves = blocks * threads / 64;
    hipMalloc(&st, 16 * waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, sink, 8, st);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, sink, iters, st);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), st, 16 * waves, hipMemcpyDeviceToHost);
    double sc = 0, sr = 0;
    for (int w = 0; w < waves; ++w) { sc += h[2 * w]; sr += h[2 * w + 1]; }
    const double cyc_per_op = sc / waves / (iters * ops_per_iter_per_wave);
    const double mhz = sc / sr * 100.0;
    const double tf = waves * (double)iters * ops_per_iter_per_wave * flops_per_op / (ms * 1e-3) / 1e12;
    printf("%-28s blocks=%5d thr=%3d  cyc/op/wave=%8.2f  clock=%7.1f MHz  %8.2f TFLOP/s  (%.3f ms)\n", name, blocks,
           threads, cyc_per_op, mhz, tf, ms);
    hipFree(sink);
    hipFree(st);
}

int main() {
    const int it = 4096;
    printf("== v_mfma_f64_16x16x4 (2048 flop/op)\n");
    run("mfma16 acc1 1wave", mfma16<1>, 1, 64, it, 1, 2048);
    run("mfma16 acc2 1wave", mfma16<2>, 1, 64, it, 2, 2048);
    run("mfma16 acc4 1wave", mfma16<4>, 1, 64, it, 4, 2048);
    run("mfma16 acc8 1wave", mfma16<8>, 1, 64, it, 8, 2048);
    run("mfma16 acc16 1wave", mfma16<16>, 1, 64, it, 16, 2048);
    for (int b : {256, 512, 1024, 2048})
        run("mfma16 acc4 chip", mfma16<4>, b, 256, it, 4, 2048);
    run("mfma16 acc8 chip 512thr", mfma16<8>, 256, 512, it, 8, 2048);
    printf("== v_mfma_f64_4x4x4 (4 blocks: 512 flop/op)\n");
    run("mfma4 acc1 1wave", mfma4<1>, 1, 64, it, 1, 512);
    run("mfma4 acc4 1wave", mfma4<4>, 1, 64, it, 4, 512);
    run("mfma4 acc8 1wave", mfma4<8>, 1, 64, it, 8, 512);
    for (int b : {256, 512, 2048}) run("mfma4 acc8 chip", mfma4<8>, b, 256, it, 8, 512);
    printf("== v_fma_f64 (128 flop/op)\n");
    run("vfma acc1 1wave", vfma<1>, 1, 64, it * 4, 1, 128);
    run("vfma acc8 1wave", vfma<8>, 1, 64, it * 4, 8, 128);
    run("vfma acc16 1wave", vfma<16>, 1, 64, it * 4, 16, 128);
    for (int b : {256, 512, 1024, 2048}) run("vfma acc16 chip", vfma<16>, b, 256, it * 4, 16, 128);
    printf("== transcendental cost (cycles per op incl. one add)\n");
    run("sqrt(double)", vsqrt<0>, 1, 64, 1024, 8, 1);
    run("rsq_f64 builtin", vsqrt<1>, 1, 64, 1024, 8, 1);
    run("log(double)", vsqrt<2>, 1, 64, 1024, 8, 1);
    run("1.0/x", vsqrt<3>, 1, 64, 1024, 8, 1);
    run("v_sqrt_f64 builtin", vsqrt<4>, 1, 64, 1024, 8, 1);
    return 0;
}

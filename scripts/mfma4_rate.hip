// What limits v_mfma_f64_4x4x4 issue in the real M2L loops?  Lone-wave and 2-waves/SIMD cycle
// counts for: constant operands, many accumulators, distinct B registers, B from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STAMP0 const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define STAMP1(nops) const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;

template <int NA> __global__ __launch_bounds__(512) void k_const(double *sink, int iters, unsigned long long *st) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double c[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) c[i] = 0;
    STAMP0
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
    }
    STAMP1()
    double s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += c[i];
    if (s == 12345.678) sink[0] = s;
}

template <int NA, int NB> __global__ __launch_bounds__(512) void k_breg(double *sink, int iters, unsigned long long *st) {
    double a[4], b[NB], c[NA];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = 1.0 + i + threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = 1.0 - i * 1e-3 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < NA; ++i) c[i] = 0;
    STAMP0
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], b[i % NB], c[i], 0, 0, 0);
    }
    STAMP1()
    double s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += c[i];
    if (s == 12345.678) sink[0] = s;
}

template <int NA> __global__ __launch_bounds__(512) void k_lds(double *sink, int iters, unsigned long long *st) {
    __shared__ double lds[4 * (NA / 2) * 32];
    for (int i = threadIdx.x; i < 4 * (NA / 2) * 32; i += blockDim.x) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63, ak = lane >> 4, ai = lane & 3;
    const double *bfrag = lds + (ak * 4 + ai) * 2;
    double a[4], c[NA];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = 1.0 + i + threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < NA; ++i) c[i] = 0;
    STAMP0
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int G = 0; G < NA / 2; ++G) {
                const double2 b2 = *reinterpret_cast<const double2 *>(bfrag + ((e * (NA / 2) + G) * 16) * 2);
                c[2 * G] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[e], b2.x, c[2 * G], 0, 0, 0);
                c[2 * G + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[e], b2.y, c[2 * G + 1], 0, 0, 0);
            }
        }
    }
    STAMP1()
    double s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += c[i];
    if (s == 12345.678) sink[0] = s;
}

template <class K> void run(const char *name, K kernel, int blocks, int threads, int iters, double mfma_per_iter) {
    double *sink;
    unsigned long long *st;
    hipMalloc(&sink, 8);
    const int waves = blocks * threads / 64;
    hipMalloc(&st, 8 * waves);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, sink, 2, st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, sink, iters, st);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), st, 8 * waves, hipMemcpyDeviceToHost);
    double sc = 0;
    for (auto v : h) sc += v;
    printf("%-34s blocks=%4d thr=%3d cyc/mfma/wave=%7.2f  chip %7.2f TFLOP/s\n", name, blocks, threads,
           sc / waves / (iters * mfma_per_iter), waves * (double)iters * mfma_per_iter * 512 / (ms * 1e-3) / 1e12);
    hipFree(sink);
    hipFree(st);
}

int main() {
    const int it = 512;
    run("const ops, 8 acc, 1 wave", k_const<8>, 1, 64, it, 8);
    run("const ops, 88 acc, 1 wave", k_const<88>, 1, 64, it, 88);
    run("const ops, 88 acc, 512thr x256", k_const<88>, 256, 512, it, 88);
    run("breg 16 acc 16 b, 1 wave", k_breg<16, 16>, 1, 64, it, 16);
    run("breg 88 acc 8 b, 1 wave", k_breg<88, 8>, 1, 64, it, 88);
    run("breg 88 acc 8 b, 512thr x256", k_breg<88, 8>, 256, 512, it, 88);
    run("lds  88 acc, 1 wave", k_lds<88>, 1, 64, it / 4, 352);
    run("lds  88 acc, 256thr x1", k_lds<88>, 1, 256, it / 4, 352);
    run("lds  88 acc, 512thr x1", k_lds<88>, 1, 512, it / 4, 352);
    run("lds  88 acc, 512thr x256", k_lds<88>, 256, 512, it / 4, 352);
    run("lds  16 acc, 512thr x256", k_lds<16>, 256, 512, it, 64);
    return 0;
}

// One kernel evaluation of the pair loops (P2P / M2P / P2L: device.hip direct_tile, p2p_sym*_kernel, wx_sym_kernel),
// isolated so that scripts/pair_instruction_counts.py can count its FP64 instructions in the ISA: the distance from
// wave-uniform target coordinates (SGPR operands, as in the unordered-pair kernels), kernel_value_r2<KID>, and the
// two accumulations (row sum and column sum).  Compiled with the flags of the library; never linked into it.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

using namespace bbfmm;

template <int KID>
__global__ void pair_probe(KernelSpec ks, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                           const double *__restrict__ w, const double *__restrict__ t, int n, double *__restrict__ out) {
    const double tx = t[0], ty = t[1], tz = t[2], tw = t[3]; // uniform addresses: scalar loads
    double racc = 0.0, csum = 0.0;
#pragma unroll 1
    for (int j = threadIdx.x; j < n; j += 64) {
        const double dx = tx - x[j], dy = ty - y[j], dz = tz - z[j];
        const double v = kernel_value_r2<KID>(ks, dx * dx + dy * dy + dz * dz);
        racc += v * w[j];
        csum += v * tw;
    }
    out[threadIdx.x] = racc;
    out[64 + threadIdx.x] = csum;
}

#define INST(K) template __global__ void pair_probe<K>(KernelSpec, const double *, const double *, const double *, const double *, const double *, int, double *);
INST(0) INST(1) INST(2) INST(3) INST(4) INST(5) INST(6) INST(7) INST(8) INST(9) INST(100) INST(101)

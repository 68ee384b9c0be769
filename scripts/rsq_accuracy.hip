// Accuracy of v_rsq_f64 on gfx950 and of the square roots refined from it (kernels.hpp bb_sqrt): max relative
// error against the correctly rounded library sqrt over 2^24 arguments.   hipcc --offload-arch=gfx950 -O3 -o rsq_accuracy rsq_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *seed, double *one, double *two, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double y = __builtin_amdgcn_rsq(v);
    seed[i] = y;
    {   // one Newton step on g = x*y
        double g = v * y, h = 0.5 * y;
        const double d = fma(-g, g, v);
        one[i] = fma(d, h, g);
    }
    {   // the two-stage refinement of kernels.hpp
        double g = v * y, h = 0.5 * y;
        const double r = fma(-h, g, 0.5);
        g = fma(g, r, g);
        h = fma(h, r, h);
        const double d = fma(-g, g, v);
        two[i] = fma(d, h, g);
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::ldexp(0.5 + u, (int)(s % 40) - 30); // 2^-30 .. 2^10
    }
    double *dx, *ds, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, d1, d2, n);
    std::vector<double> sd(n), o(n), t(n);
    hipMemcpy(sd.data(), ds, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double es = 0, e1 = 0, e2 = 0;
    long w1 = 0, w2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double r = sqrtl((long double)x[i]);
        es = std::fmax(es, std::fabs((double)((long double)sd[i] * r - 1.0L)));
        const double ex = std::sqrt(x[i]);
        e1 = std::fmax(e1, std::fabs((double)(((long double)o[i] - r) / r)));
        e2 = std::fmax(e2, std::fabs((double)(((long double)t[i] - r) / r)));
        w1 += o[i] != ex; w2 += t[i] != ex;
    }
    std::printf("{\"v_rsq_f64_max_rel_err\": %.3e, \"one_step_max_rel_err\": %.3e, \"one_step_not_correctly_rounded\": %.4f, "
                "\"two_stage_max_rel_err\": %.3e, \"two_stage_not_correctly_rounded\": %.4f}\n", es, e1, (double)w1 / n, e2, (double)w2 / n);
    return 0;
}

// Accuracy of v_rsq_f64 on gfx950 and of the square roots refined from it (kernels.hpp bb_sqrt): max relative
// error against the correctly rounded library sqrt over 2^24 arguments.   hipcc --offload-arch=gfx950 -O3 -o rsq_accuracy rsq_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *seed, double *one, double *two, double *cub, double *cubr, int n) {
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
    {   // round 6: ONE cubic step (kernels.hpp bb_sqrt / bb_sqrt_rsqrt)
        const double t = v * y;
        const double e = fma(-t, y, 1.0);
        const double p = fma(0.375, e, 0.5);
        cub[i] = fma(t * e, p, t);          // sqrt
        cubr[i] = fma(y * e, p, y);         // 1 / sqrt
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
    double *dx, *ds, *d1, *d2, *d3, *d4;
    hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8); hipMalloc(&d4, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, d1, d2, d3, d4, n);
    std::vector<double> sd(n), o(n), t(n), c(n), cr(n);
    hipMemcpy(c.data(), d3, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cr.data(), d4, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(sd.data(), ds, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double es = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, u3 = 0;
    long w1 = 0, w2 = 0, w3 = 0;
    for (int i = 0; i < n; ++i) {
        const long double r = sqrtl((long double)x[i]);
        es = std::fmax(es, std::fabs((double)((long double)sd[i] * r - 1.0L)));
        const double ex = std::sqrt(x[i]);
        e1 = std::fmax(e1, std::fabs((double)(((long double)o[i] - r) / r)));
        e2 = std::fmax(e2, std::fabs((double)(((long double)t[i] - r) / r)));
        w1 += o[i] != ex; w2 += t[i] != ex; w3 += c[i] != ex;
        e3 = std::fmax(e3, std::fabs((double)(((long double)c[i] - r) / r)));
        e4 = std::fmax(e4, std::fabs((double)((long double)cr[i] * r - 1.0L)));
        e5 = std::fmax(e5, std::fabs((double)(((long double)x[i] * (long double)cr[i] - r) / r)));   // sqrt as x * rsqrt (bb_sqrt_rsqrt)
        u3 = std::fmax(u3, std::fabs(c[i] - ex) / (std::nextafter(ex, 2 * ex) - ex));
    }
    std::printf("{\"v_rsq_f64_max_rel_err\": %.3e, \"one_step_max_rel_err\": %.3e, \"one_step_not_correctly_rounded\": %.4f, "
                "\"two_stage_max_rel_err\": %.3e, \"two_stage_not_correctly_rounded\": %.4f, \"cubic_sqrt_max_rel_err\": %.3e, \"cubic_sqrt_max_ulp\": %.2f, "
                "\"cubic_sqrt_not_correctly_rounded\": %.4f, \"cubic_rsqrt_max_rel_err\": %.3e, \"sqrt_as_x_times_cubic_rsqrt_max_rel_err\": %.3e}\n",
                es, e1, (double)w1 / n, e2, (double)w2 / n, e3, u3, (double)w3 / n, e4, e5);
    return 0;
}

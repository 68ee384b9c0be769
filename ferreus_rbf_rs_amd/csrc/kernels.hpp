// The closed kernel set of ferreus_rbf_utils (KernelType registry,
// ferreus_rbf_utils/src/utils.rs:558-571) as host+device functions, plus the two
// extension kernels (Gaussian, multiquadric) BASELINE.json names but the reference
// does not ship.  Arithmetic follows rbf_kernels.rs / non_rbf_kernels.rs.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define BBFMM_HD __host__ __device__
#else
#define BBFMM_HD
#endif

namespace bbfmm {

enum KernelId : int {
    kLinear = 0,
    kThinPlateSpline = 1,
    kCubic = 2,
    kSpheroidal3 = 3,
    kSpheroidal5 = 4,
    kSpheroidal7 = 5,
    kSpheroidal9 = 6,
    kLaplacian = 7,
    kOneOverR2 = 8,
    kOneOverR4 = 9,
    kGaussianExt = 100,
    kMultiquadricExt = 101,
};

inline bool kernel_id_valid(int id) { return (id >= 0 && id <= 9) || id == 100 || id == 101; }

// KernelParams (kernel_helpers.rs:17-23) + the values SpheroidalRbfKernel::new derives
// once (rbf_kernels.rs:232-243).
struct KernelSpec {
    int id;
    int sph_pow;
    double base_range, total_sill;
    double s2, ip2, near_slope, far_coef;
    double inv_br2;
};

inline KernelSpec make_kernel_spec(int id, double base_range, double total_sill) {
    // constants.rs:21-50: inflexion_point, linear_slope, range_scaling, inv_y_intercept
    static const double sph[4][4] = {
        {0.5000000000, 0.7500000000, 2.6798340586, 0.8734640537},
        {0.4082482905, 1.0206207262, 1.5822795750, 0.8575980168},
        {0.3535533906, 1.2374368671, 1.2008676644, 0.8494862533},
        {0.3162277660, 1.4230249471, 1.0000000000, 0.8445585690},
    };
    KernelSpec k{};
    k.id = id;
    k.base_range = base_range;
    k.total_sill = total_sill;
    if (id >= kSpheroidal3 && id <= kSpheroidal9) {
        const double *c = sph[id - kSpheroidal3];
        const double s = c[2] / base_range;
        k.s2 = s * s;
        k.ip2 = c[0] * c[0];
        k.near_slope = total_sill * c[1] * s;
        k.far_coef = total_sill * c[3];
        k.sph_pow = id - kSpheroidal3 + 1; // POW, rbf_kernels.rs:178-205
    }
    k.inv_br2 = 1.0 / (base_range * base_range);
    return k;
}

// sqrt / 1/sqrt / 1/x for the pair loops.  Host: libm and IEEE division.  Device: v_rsq_f64 / v_rcp_f64 seeds refined by
// FMAs.  x >= 0 always (sums of squares).
// Round 6: ONE cubic step instead of the two-stage Goldschmidt refinement of rounds 1-5.  The seed y = v_rsq_f64(x) is good to
// 5.2e-8 (scripts/rsq_accuracy.hip), e = 1 - x y^2 is twice that, and 1 / sqrt(1 - e) = 1 + e/2 + 3 e^2/8 + O(e^3) leaves a
// truncation of 3e-22: the result is within 2 ulp of the correctly rounded root (measured: rsq_accuracy.hip), for five
// instructions behind the seed where the two-stage form took seven (sqrt) and ten (sqrt and 1/sqrt) -- 17 -> 15 FP64
// instructions per LinearRbf pair, 28 -> 24 per Spheroidal3 pair, in kernels that run at the FP64 issue roof.  The reference's
// own sqrt is the correctly rounded one; two ulp per kernel value sit five orders below the 1e-11 the parity tests hold.
// x = 0 (a point against itself): only the argument of rsq is clamped (to 1e-300), so the seed stays finite and x * y = 0
// carries an exact zero through (sqrt(0) = 0 exactly); the select that would otherwise guard 0 * inf costs three
// instructions per pair.
BBFMM_HD inline void bb_sqrt_rsqrt(double x, double *s, double *rs) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(fmax(x, 1e-300));
    const double t = x * y;
    const double e = fma(-t, y, 1.0);      // 1 - x y^2
    const double p = fma(0.375, e, 0.5);
    const double r = fma(y * e, p, y);     // 1 / sqrt(x)
    *rs = r;
    *s = x * r;                            // sqrt(x)
#else
    *s = sqrt(x);
    *rs = 1.0 / *s;
#endif
}
BBFMM_HD inline double bb_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(fmax(x, 1e-300));
    const double t = x * y;                // sqrt(x) to the seed's accuracy
    const double e = fma(-t, y, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(t * e, p, t);
#else
    return sqrt(x);
#endif
}
BBFMM_HD inline double bb_rcp(double x) { // x > 0, finite
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
#else
    return 1.0 / x;
#endif
}

// Natural logarithm for the pair loops (x > 0, normal).  Host: libm.  Device: exponent / mantissa
// split, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m - 1) / (m + 1)) as an odd series in
// s = (m - 1) / (m + 1), |s| <= 0.1716, through s^23 (truncation < 1e-18): about 35 FMA-class
// instructions against ~420 cycles per wave measured for the library log on MI355X.
BBFMM_HD inline double bb_log(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double s = (m - 1.0) * bb_rcp(m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 23.0;
    p = fma(p, s2, 1.0 / 21.0);
    p = fma(p, s2, 1.0 / 19.0);
    p = fma(p, s2, 1.0 / 17.0);
    p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0);
    p = fma(p, s2, 1.0 / 11.0);
    p = fma(p, s2, 1.0 / 9.0);
    p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0);
    p = fma(p, s2, 1.0 / 3.0);
    const double ed = (double)e;
    const double t = s + s;
    // e ln2 in two parts; ln2_hi has 11 trailing zero bits so e * ln2_hi is exact
    const double hi = ed * 6.93147180369123816490e-01;
    const double lo2 = fma(ed, 1.90821492927058770002e-10, t * (s2 * p));
    return hi + (t + lo2);
#else
    return log(x);
#endif
}

// Value from r^2 = distance_sq (utils.rs:230-237).
template <int ID> BBFMM_HD inline double kernel_value_r2(const KernelSpec &k, double r2) {
    if constexpr (ID == kLinear) { // rbf_kernels.rs:25-36
        return -bb_sqrt(r2);
    } else if constexpr (ID == kThinPlateSpline) { // rbf_kernels.rs:69-84
#if defined(__HIP_DEVICE_COMPILE__)
        // r^2 ln r = r2 ln(r2) / 2: no square root; |r| < eps  <=>  r2 < eps^2
        return (r2 < DBL_EPSILON * DBL_EPSILON) ? 0.0 : 0.5 * r2 * bb_log(r2);
#else
        const double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : (r * r) * log(r);
#endif
    } else if constexpr (ID == kCubic) { // rbf_kernels.rs:118-130
        const double r = bb_sqrt(r2);
        return r * r * r;
    } else if constexpr (ID >= kSpheroidal3 && ID <= kSpheroidal9) { // rbf_kernels.rs:245-256
        constexpr int POW = ID - kSpheroidal3 + 1;
        const double sr2 = k.s2 * r2;
#if defined(__HIP_DEVICE_COMPILE__)
        // near: sill - slope * sqrt(r2); far: far_coef / ((1 + s2 r2)^POW * sqrt(1 + s2 r2)).  Both
        // come from one square root of the selected argument (lanes of a wave diverge here).
        const bool near = sr2 <= k.ip2;
        double sq, rs;
        bb_sqrt_rsqrt(near ? r2 : 1.0 + sr2, &sq, &rs);
        const double rs2 = rs * rs;
        double rp = rs; // rs^(2 POW + 1) = 1 / (t^POW sqrt(t))
#pragma unroll
        for (int i = 0; i < POW; ++i) rp *= rs2;
        return near ? k.total_sill - k.near_slope * sq : k.far_coef * rp;
#else
        if (sr2 <= k.ip2) return k.total_sill - k.near_slope * sqrt(r2);
        const double t = 1.0 + sr2;
        double tp = t;
        for (int i = 1; i < POW; ++i) tp *= t;
        return k.far_coef / (tp * sqrt(t));
#endif
    } else if constexpr (ID == kLaplacian) { // non_rbf_kernels.rs:20-37: 0 if |r| < eps, else 1 / r
#if defined(__HIP_DEVICE_COMPILE__)
        double sq, rs;
        bb_sqrt_rsqrt(r2, &sq, &rs);
        return (sq < DBL_EPSILON) ? 0.0 : rs;
#else
        const double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : 1.0 / r;
#endif
    } else if constexpr (ID == kOneOverR2) { // non_rbf_kernels.rs:70-86
#if defined(__HIP_DEVICE_COMPILE__)
        return (r2 < DBL_EPSILON * DBL_EPSILON) ? 0.0 : bb_rcp(r2);
#else
        const double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : 1.0 / (r * r);
#endif
    } else if constexpr (ID == kOneOverR4) { // non_rbf_kernels.rs:121-137
#if defined(__HIP_DEVICE_COMPILE__)
        const double q = (r2 < DBL_EPSILON * DBL_EPSILON) ? 0.0 : bb_rcp(r2);
        return q * q;
#else
        const double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : 1.0 / ((r * r) * (r * r));
#endif
    } else if constexpr (ID == kGaussianExt) {
        return exp(-r2 * k.inv_br2);
    } else { // kMultiquadricExt
        return bb_sqrt(1.0 + r2 * k.inv_br2);
    }
}

// Value and the scalar `factor` such that grad = factor * (target - source)
// (evaluate_value_gradient of each kernel: rbf_kernels.rs:38-57,86-106,132-152,266-300;
// non_rbf_kernels.rs:39-58,88-109,139-157).  factor = 0 where the reference zero-fills.
template <int ID>
BBFMM_HD inline double kernel_value_grad_r2(const KernelSpec &k, double r2, double *factor) {
    const bool zero = (r2 <= DBL_EPSILON);
    if constexpr (ID == kLinear) {
        const double r = bb_sqrt(r2);
        *factor = zero ? 0.0 : -1.0 / r;
        return -r;
    } else if constexpr (ID == kThinPlateSpline) {
        if (zero) { *factor = 0.0; return 0.0; }
        const double lr = 0.5 * bb_log(r2);
        *factor = 2.0 * lr + 1.0;
        return r2 * lr;
    } else if constexpr (ID == kCubic) {
        if (zero) { *factor = 0.0; return 0.0; }
        const double r = bb_sqrt(r2);
        *factor = 3.0 * r;
        return r2 * r;
    } else if constexpr (ID >= kSpheroidal3 && ID <= kSpheroidal9) {
        const double value = kernel_value_r2<ID>(k, r2);
        if (zero) { *factor = 0.0; return value; }
        const double sr2 = k.s2 * r2;
        if (sr2 <= k.ip2) {
            *factor = -k.near_slope * (1.0 / bb_sqrt(r2));
        } else {
            const double t = 1.0 + sr2;
            const double p = static_cast<double>(ID - kSpheroidal3 + 1) + 0.5;
            *factor = -2.0 * p * k.s2 * k.far_coef / pow(t, p + 1.0);
        }
        return value;
    } else if constexpr (ID == kLaplacian) {
        if (zero) { *factor = 0.0; return 0.0; }
        const double ir = 1.0 / bb_sqrt(r2);
        *factor = -(ir * ir * ir);
        return ir;
    } else if constexpr (ID == kOneOverR2) {
        if (zero) { *factor = 0.0; return 0.0; }
        *factor = -2.0 * (1.0 / (r2 * r2));
        return 1.0 / r2;
    } else if constexpr (ID == kOneOverR4) {
        if (zero) { *factor = 0.0; return 0.0; }
        *factor = -4.0 * (1.0 / (r2 * r2 * r2));
        return 1.0 / (r2 * r2);
    } else if constexpr (ID == kGaussianExt) {
        const double v = exp(-r2 * k.inv_br2);
        *factor = -2.0 * k.inv_br2 * v;
        return v;
    } else {
        const double v = bb_sqrt(1.0 + r2 * k.inv_br2);
        *factor = k.inv_br2 / v;
        return v;
    }
}

// Host-side runtime dispatch (operator assembly).
inline double kernel_value_r2_rt(const KernelSpec &k, double r2) {
    switch (k.id) {
    case kLinear: return kernel_value_r2<kLinear>(k, r2);
    case kThinPlateSpline: return kernel_value_r2<kThinPlateSpline>(k, r2);
    case kCubic: return kernel_value_r2<kCubic>(k, r2);
    case kSpheroidal3: return kernel_value_r2<kSpheroidal3>(k, r2);
    case kSpheroidal5: return kernel_value_r2<kSpheroidal5>(k, r2);
    case kSpheroidal7: return kernel_value_r2<kSpheroidal7>(k, r2);
    case kSpheroidal9: return kernel_value_r2<kSpheroidal9>(k, r2);
    case kLaplacian: return kernel_value_r2<kLaplacian>(k, r2);
    case kOneOverR2: return kernel_value_r2<kOneOverR2>(k, r2);
    case kOneOverR4: return kernel_value_r2<kOneOverR4>(k, r2);
    case kGaussianExt: return kernel_value_r2<kGaussianExt>(k, r2);
    case kMultiquadricExt: return kernel_value_r2<kMultiquadricExt>(k, r2);
    default: return NAN;
    }
}

// Every kernel of the closed set implements evaluate_value_gradient (the trait's
// default returns None, ferreus_bbfmm/src/traits.rs:26-33).
inline bool kernel_supports_gradients(int id) { return kernel_id_valid(id); }

} // namespace bbfmm

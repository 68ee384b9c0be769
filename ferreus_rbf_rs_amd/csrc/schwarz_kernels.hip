// See schwarz_kernels.hpp.
#include "schwarz_kernels.hpp"

namespace bbfmm {

namespace {
constexpr int kMaxBasis = 10; // quadratic drift in 3-D

__global__ __launch_bounds__(256) void residual_kernel(const double *__restrict__ rg, const double *__restrict__ y,
                                                       const double *__restrict__ sl, double nugget,
                                                       const int32_t *__restrict__ rows, int64_t m,
                                                       double *__restrict__ res) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t r = rows ? rows[j] : j;
    res[r] = rg[r] - y[j] - nugget * sl[r];
}

__global__ __launch_bounds__(256) void add_rows_kernel(const double *__restrict__ corr, const int32_t *__restrict__ rows,
                                                       int64_t m, double *__restrict__ sl) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t r = rows ? rows[j] : j;
    sl[r] += corr[r];
}

// stage 1: block b sums a fixed contiguous share of the rows (fixed order inside a thread, fixed tree across
// the threads): the result does not depend on scheduling
__global__ __launch_bounds__(256) void project_partial_kernel(const double *__restrict__ ortho, int64_t n, int basis,
                                                              const double *__restrict__ corr,
                                                              const int32_t *__restrict__ rows, int64_t m,
                                                              double *__restrict__ part) {
    __shared__ double red[256];
    const int64_t per = (m + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m ? lo + per : m;
    double acc[kMaxBasis];
#pragma unroll
    for (int b = 0; b < kMaxBasis; ++b) acc[b] = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
        const int64_t r = rows ? rows[j] : j;
        const double v = corr[r];
#pragma unroll
        for (int b = 0; b < kMaxBasis; ++b)
            if (b < basis) acc[b] += ortho[(int64_t)b * n + r] * v;
    }
    for (int b = 0; b < basis; ++b) {
        __syncthreads();
        red[threadIdx.x] = acc[b];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * basis + b] = red[0];
    }
}

// stage 2: one block, basis values, partial sums added in block order
__global__ void project_final_kernel(const double *__restrict__ part, int n_blocks, int basis, double *__restrict__ proj) {
    const int b = threadIdx.x;
    if (b >= basis) return;
    double s = 0.0;
    for (int i = 0; i < n_blocks; ++i) s += part[(int64_t)i * basis + b];
    proj[b] = s;
}

__global__ __launch_bounds__(256) void subtract_projection_kernel(const double *__restrict__ ortho, int64_t n, int basis,
                                                                  const double *__restrict__ proj,
                                                                  double *__restrict__ sl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < basis; ++b) s += ortho[(int64_t)b * n + i] * proj[b];
    sl[i] -= s;
}

__global__ __launch_bounds__(256) void gather64_kernel(const double *__restrict__ src, const int64_t *__restrict__ idx,
                                                       int64_t m, double *__restrict__ dst) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m) dst[j] = src[idx[j]];
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const double *__restrict__ src, const int32_t *__restrict__ rows,
                                                          int64_t m, double *__restrict__ dst) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m) dst[j] = src[rows ? rows[j] : j];
}

// dst[row_j] = src ? src[j] : 0
__global__ __launch_bounds__(256) void scatter_rows_kernel(const double *__restrict__ src, const int32_t *__restrict__ rows,
                                                           int64_t m, double *__restrict__ dst) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m) dst[rows ? rows[j] : j] = src ? src[j] : 0.0;
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }
} // namespace

void launch_schwarz_gather_rows(const double *src, const int32_t *rows, int64_t m, double *dst, hipStream_t s) {
    if (m == 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(m)), dim3(256), 0, s, src, rows, m, dst);
}
void launch_schwarz_scatter_rows(const double *src, const int32_t *rows, int64_t m, double *dst, hipStream_t s) {
    if (m == 0) return;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(m)), dim3(256), 0, s, src, rows, m, dst);
}

void launch_schwarz_residual(const double *rg, const double *y, const double *sl, double nugget, const int32_t *rows,
                             int64_t m, double *res, hipStream_t s) {
    if (m == 0) return;
    hipLaunchKernelGGL(residual_kernel, dim3(grid_for(m)), dim3(256), 0, s, rg, y, sl, nugget, rows, m, res);
}
void launch_schwarz_add_rows(const double *corr, const int32_t *rows, int64_t m, double *sl, hipStream_t s) {
    if (m == 0) return;
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(m)), dim3(256), 0, s, corr, rows, m, sl);
}
void launch_schwarz_project(const double *ortho, int64_t n, int basis, const double *corr, const int32_t *rows, int64_t m,
                            double *part, int n_blocks, double *proj, hipStream_t s) {
    hipLaunchKernelGGL(project_partial_kernel, dim3(n_blocks), dim3(256), 0, s, ortho, n, basis, corr, rows, m, part);
    hipLaunchKernelGGL(project_final_kernel, dim3(1), dim3(64), 0, s, part, n_blocks, basis, proj);
}
void launch_schwarz_subtract_projection(const double *ortho, int64_t n, int basis, const double *proj, double *sl,
                                        hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(subtract_projection_kernel, dim3(grid_for(n)), dim3(256), 0, s, ortho, n, basis, proj, sl);
}
void launch_gather_rows64(const double *src, const int64_t *idx, int64_t m, double *dst, hipStream_t s) {
    if (m == 0) return;
    hipLaunchKernelGGL(gather64_kernel, dim3(grid_for(m)), dim3(256), 0, s, src, idx, m, dst);
}

} // namespace bbfmm

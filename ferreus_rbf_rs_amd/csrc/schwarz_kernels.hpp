// Vector kernels of the device-resident Schwarz sweep (schwarz.cpp; schwarz.rs:32-155): residual on a
// level's rows, RAS write-back into the running correction, orthogonalisation against the global
// polynomial basis.  All HBM-streaming, f64.  rows == nullptr means "all rows 0..m-1 in order".
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bbfmm {

// res[row_j] = rg[row_j] - y[j] - nugget * sl[row_j]        (rg - matvec_partial(sl), rbf.rs:1366-1376 with a zero tail)
void launch_schwarz_residual(const double *rg, const double *y, const double *sl, double nugget, const int32_t *rows,
                             int64_t m, double *res, hipStream_t s);
// sl[row_j] += corr[row_j]
void launch_schwarz_add_rows(const double *corr, const int32_t *rows, int64_t m, double *sl, hipStream_t s);
// proj[b] = sum_j ortho[b][row_j] * corr[row_j], b < basis (two deterministic stages; part: n_blocks * basis scratch)
void launch_schwarz_project(const double *ortho, int64_t n, int basis, const double *corr, const int32_t *rows, int64_t m,
                            double *part, int n_blocks, double *proj, hipStream_t s);
// sl[i] -= sum_b ortho[b][i] * proj[b], i < n      (orthogonalise, schwarz.rs:128-132)
void launch_schwarz_subtract_projection(const double *ortho, int64_t n, int basis, const double *proj, double *sl,
                                        hipStream_t s);
// dst[j] = src[idx[j]] with 64-bit indices (the coarse domain's entry list)
void launch_gather_rows64(const double *src, const int64_t *idx, int64_t m, double *dst, hipStream_t s);

// dst[j] = src[row_j]; dst[row_j] = src[j] (src == nullptr: zero) -- the packed exchange buffer of a sharded level
void launch_schwarz_gather_rows(const double *src, const int32_t *rows, int64_t m, double *dst, hipStream_t s);
void launch_schwarz_scatter_rows(const double *src, const int32_t *rows, int64_t m, double *dst, hipStream_t s);

constexpr int kSchwarzProjectBlocks = 1024;

} // namespace bbfmm

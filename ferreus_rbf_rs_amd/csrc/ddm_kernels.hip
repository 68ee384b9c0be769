// Device kernels of the Schwarz preconditioner's local solvers (see ddm_solver.hpp): one workgroup
// per leaf domain.  A domain's points are stored special points first (k of them), then the m
// others; its reduced matrix  lhs = Q^T A11 Q + Q^T A12 + A21 Q + A22  (domain.rs:312-346) is m x m,
// column-major, lower triangle used.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "ddm_solver.hpp"
#include "ferreus_bbfmm_hip.h"

namespace bbfmm {
namespace {

struct View {
    const double *x, *y, *z;
    const int64_t *gidx, *dom_off, *q_off, *fac_off;
    const int32_t *k;
    const uint8_t *internal;
    const double *q;
    double *t, *g, *fac, *work;
    const uint8_t *mode; // per domain: 0 = Cholesky factor, 1 = packed symmetric inverse (host fallback); may be null
    double *tmp;         // n_entries scratch of the fallback path
};

View make_view(const DdmLevelSolver &lv) {
    return View{lv.d_xyz[0], lv.d_xyz[1], lv.d_xyz[2], lv.d_gidx, lv.d_dom_off, lv.d_q_off, lv.d_fac_off,
                lv.d_k,      lv.d_internal, lv.d_q, lv.d_t, lv.d_g, lv.d_fac, lv.d_work, lv.d_mode, lv.d_tmp};
}

template <int KID> __device__ inline double phi(const KernelSpec &ks, const View &v, int64_t a, int64_t b) {
    const double dx = v.x[a] - v.x[b], dy = v.y[a] - v.y[b], dz = v.z[a] - v.z[b];
    return kernel_value_r2<KID>(ks, dx * dx + dy * dy + dz * dz);
}

// T[a][j] = phi(s_a, x_j);  G[a][j] = T[a][j] + sum_b A11[a][b] Q[b][j],  A11 = phi(s, s) + nugget I
template <int KID> __global__ __launch_bounds__(256) void ddm_prep_kernel(KernelSpec ks, double nugget, View v) {
    const int dom = blockIdx.x;
    const int64_t o = v.dom_off[dom];
    const int k = v.k[dom], m = static_cast<int>(v.dom_off[dom + 1] - o) - k;
    if (k == 0) return;
    __shared__ double a11[16 * 16];
    const int tid = threadIdx.x;
    for (int e = tid; e < k * k; e += 256) {
        const int a = e / k, b = e % k;
        a11[e] = phi<KID>(ks, v, o + a, o + b) + (a == b ? nugget : 0.0);
    }
    __syncthreads();
    const double *Q = v.q + v.q_off[dom];
    double *T = v.t + v.q_off[dom], *G = v.g + v.q_off[dom];
    for (int e = tid; e < k * m; e += 256) {
        const int a = e / m, j = e % m;
        const double tv = phi<KID>(ks, v, o + a, o + k + j);
        double w = 0.0;
        for (int b = 0; b < k; ++b) w += a11[a * k + b] * Q[b * m + j];
        T[e] = tv;
        G[e] = tv + w;
    }
}

// lhs[i][j] (i >= j) = phi(x_i, x_j) + nugget [i == j] + sum_a (Q[a][i] G[a][j] + T[a][i] Q[a][j])
// Factors are stored packed: the lower triangle column by column, column c = rows c..m-1
// (m(m+1)/2 doubles; the reference packs too, LltRfp linalg.rs:37-72).  Element (r, c), r >= c:
__device__ __forceinline__ int64_t pk(int r, int c, int m) {
    return static_cast<int64_t>(c) * m - (static_cast<int64_t>(c) * (c - 1)) / 2 + (r - c);
}

template <int KID> __global__ __launch_bounds__(256) void ddm_assemble_kernel(KernelSpec ks, double nugget, View v) {
    const int dom = blockIdx.x;
    const int64_t o = v.dom_off[dom];
    const int k = v.k[dom], m = static_cast<int>(v.dom_off[dom + 1] - o) - k;
    const double *Q = v.q + v.q_off[dom], *T = v.t + v.q_off[dom], *G = v.g + v.q_off[dom];
    double *A = v.fac + v.fac_off[dom];
    // blockIdx.y takes every gridDim.y-th group of four columns; a thread keeps its row's coordinates and its
    // Q / T entries in registers across the four columns of a group
    constexpr int KMAX = 10; // monomials up to degree 2 in 3-D
    for (int jg = 4 * blockIdx.y; jg < m; jg += 4 * gridDim.y)
        for (int i = jg + threadIdx.x; i < m; i += 256) {
            const int64_t gi = o + k + i;
            const double xi = v.x[gi], yi = v.y[gi], zi = v.z[gi];
            double qi[KMAX], ti[KMAX];
#pragma unroll
            for (int a = 0; a < KMAX; ++a) {
                qi[a] = a < k ? Q[a * m + i] : 0.0;
                ti[a] = a < k ? T[a * m + i] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = jg + c;
                if (j >= m || i < j) continue;
                const int64_t gj = o + k + j;
                const double dx = xi - v.x[gj], dy = yi - v.y[gj], dz = zi - v.z[gj];
                double s = kernel_value_r2<KID>(ks, dx * dx + dy * dy + dz * dz) + (i == j ? nugget : 0.0);
#pragma unroll
                for (int a = 0; a < KMAX; ++a)
                    if (a < k) s += qi[a] * G[a * m + j] + ti[a] * Q[a * m + j];
                A[pk(i, j, m)] = s;
            }
        }
}

constexpr int NB = 32; // panel width of the one-large-matrix path and of the substitution kernels
constexpr int CB = 64; // block-column width of the per-domain factorisation

// In-place lower Cholesky, one workgroup per matrix, LEFT-looking over 64-column block columns: a
// block column is first brought up to date with everything left of it (64 x 64 tiles on
// v_mfma_f64_16x16x4, one tile = 4 x 4 MFMA tiles per wave; the accumulators stay in registers over the
// whole sweep, so a tile is read and written once), then its diagonal block is factorised in LDS,
// inverted there in place, and the rows below are a product with that inverse on the same tiles.
// (The right-looking order with 32-column panels re-read and re-wrote the whole trailing matrix per
// panel: 0.45 GB per 1,220-point domain, HBM bound; this order reads 0.08 GB.)
//
// MFMA operand map (cdna_hip_programming.md): lane l supplies A[l&15][l>>4] and B[l>>4][l&15], and
// holds D[(l>>4) + 4v][l&15], v = 0..3.  The tiles are computed transposed, D' = Pc Pr^T, so that
// the sixteen lanes of a result register walk down a column of the packed factor (contiguous).
using d4 = __attribute__((ext_vector_type(4))) double;

// A 64 x 64 tile per WAVE, operands straight from HBM into the MFMA registers: lane l of an operand fragment holds
// element [l & 15][k + (l >> 4)] of a 16-row group, and the sixteen lanes of a k column are sixteen consecutive rows of
// a packed column -- one 128-byte segment -- so the loads need no staging, the waves of a workgroup never wait for
// each other inside a sweep, and the k step after the one being multiplied is already on its way.
//   acc[ci][ri] += C[16 ci + .][k] R[16 ri + .][k]^T,  k = k0 .. k0 + K (K a multiple of 4)
// colsrc(ci, k) / rowsrc(ri, k) return the lane's element of the column-side / row-side fragment.
// The lane's packed column (column c0 + k + (l >> 4), as a pointer such that p[r] is row r of it) is carried along.
// The loads are unconditional (the callers clamp rows and select zeros afterwards) and the two fragment sets take
// turns, so that no wait stands between a set's loads and the other set's products (K a multiple of 8).
template <class CF, class RF>
__device__ __forceinline__ void wave_tile_sweep(const double *A, int m, int c0, int K, int lk, CF &&colsrc, RF &&rowsrc,
                                                d4 (&acc)[4][4]) {
    int c = c0 + lk;
    const double *p = A + (static_cast<int64_t>(c) * m - static_cast<int64_t>(c) * (c - 1) / 2 - c); // pk(r, c) = p[r]
    double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a0[i] = colsrc(i, 0, p);
        b0[i] = rowsrc(i, 0, p);
    }
#pragma unroll 1
    for (int k = 0; k < K; k += 8) {
        p += 4 * m - 4 * c - 10; // pk(r, c + 4) - pk(r, c)
        c += 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a1[i] = colsrc(i, k + 4, p);
            b1[i] = rowsrc(i, k + 4, p);
        }
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) acc[ci][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ci], b0[ri], acc[ci][ri], 0, 0, 0);
        if (k + 8 < K) { // (wave-uniform; the last turn re-reads its own columns instead of running past the sweep)
            p += 4 * m - 4 * c - 10;
            c += 4;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] = colsrc(i, k + 8 < K ? k + 8 : k + 4, p);
            b0[i] = rowsrc(i, k + 8 < K ? k + 8 : k + 4, p);
        }
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) acc[ci][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ci], b1[ri], acc[ci][ri], 0, 0, 0);
    }
}

__global__ __launch_bounds__(256, 2) void ddm_cholesky_kernel(View v, int *fail) {
    const int dom = blockIdx.x;
    const int64_t o = v.dom_off[dom];
    const int m = static_cast<int>(v.dom_off[dom + 1] - o) - v.k[dom];
    double *A = v.fac + v.fac_off[dom];
    // LDS: the diagonal block (steps 2, 3) only -- 34 KB per workgroup
    __shared__ double Ldbuf[CB * (CB + 1)];
    __shared__ double colj[CB];
    __shared__ int bad;
    double(*Ld)[CB + 1] = reinterpret_cast<double(*)[CB + 1]>(Ldbuf);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (tid == 0) bad = 0;
    for (int jb = 0; jb < m; jb += CB) {
        const int nb = min(CB, m - jb);
        // 1. A[jb.., jb..jb+nb) -= L[jb.., 0..jb) L[jb..jb+nb, 0..jb)^T, a 64-row tile per wave and turn: the four waves
        //    walk adjacent tiles through the same columns at about the same time (the block row's fragments come from
        //    the cache for three of them)
        if (jb > 0)
            for (int tr = jb + 64 * wave; tr < m; tr += 256) {
                d4 acc[4][4] = {};
                auto colsrc = [&](int ci, int, const double *p) {
                    const int r = 16 * ci + li;
                    const double x = p[jb + min(r, nb - 1)];
                    return r < nb ? x : 0.0;
                };
                auto rowsrc = [&](int ri, int, const double *p) {
                    const int r = tr + 16 * ri + li;
                    const double x = p[min(r, m - 1)];
                    return r < m ? x : 0.0;
                };
                wave_tile_sweep(A, m, 0, jb, lk, colsrc, rowsrc, acc); // jb is a multiple of 64
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = tr + 16 * ri + li, c = jb + 16 * ci + lk + 4 * q;
                            if (r < m && c < jb + nb && r >= c) A[pk(r, c, m)] -= acc[ci][ri][q];
                        }
            }
        __threadfence_block();
        __syncthreads();
        // 2. diagonal block, always handled as 64 x 64 (a short last block is padded with the identity)
        for (int e = tid; e < CB * CB; e += 256) {
            const int r = e & (CB - 1), c = e >> 6;
            Ld[r][c] = (r < nb && c < nb) ? (r >= c ? A[pk(jb + r, jb + c, m)] : 0.0) : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        for (int c = 0; c < nb; ++c) {
            if (tid == 0) {
                const double dd = Ld[c][c];
                if (!(dd > 0.0)) bad = 1;
                Ld[c][c] = sqrt(dd > 0.0 ? dd : 1.0);
            }
            __syncthreads();
            if (tid > c && tid < CB) Ld[tid][c] /= Ld[c][c];
            __syncthreads();
            // rank-1 update of the columns right of c: thread = (row, one of four column phases)
            {
                const int r = tid & (CB - 1);
                const double lrc = Ld[r][c];
                for (int c2 = c + 1 + (tid >> 6); c2 <= r; c2 += 4) Ld[r][c2] -= lrc * Ld[c2][c];
            }
            __syncthreads();
        }
        for (int e = tid; e < CB * CB; e += 256) {
            const int r = e & (CB - 1), c = e >> 6;
            if (r >= c && r < nb) A[pk(jb + r, jb + c, m)] = Ld[r][c];
        }
        if (jb + nb >= m) break;
        // 3. rows below the block: X = A21 L11^{-T}.  L11 is inverted in place in LDS (column by column from
        // the right, as LAPACK's trti2), then X is a product on the same MFMA tiles, a row tile per wave.
        __syncthreads();
        for (int j = CB - 1; j >= 0; --j) {
            if (tid < CB) colj[tid] = Ld[tid][j];
            __syncthreads();
            const double dj = 1.0 / colj[j];
            if (tid == j) Ld[j][j] = dj;
            if (tid > j && tid < CB) { // new[i] = -dj * sum_{k=j+1..i} inv[i][k] * old[k][j]
                double sacc = 0.0;
                for (int k = j + 1; k <= tid; ++k) sacc += Ld[tid][k] * colj[k];
                Ld[tid][j] = -dj * sacc;
            }
            __syncthreads();
        }
        for (int tr = jb + nb + 64 * wave; tr < m; tr += 256) {
            d4 acc[4][4] = {};
            auto colsrc = [&](int ci, int k, const double *) { return Ld[16 * ci + li][k + lk]; }; // inv(L11)[column of X][k]
            auto rowsrc = [&](int ri, int, const double *p) { // (nb = 64 here: the last block column has no rows below)
                const int r = tr + 16 * ri + li;
                const double x = p[min(r, m - 1)];
                return r < m ? x : 0.0;
            };
            wave_tile_sweep(A, m, jb, CB, lk, colsrc, rowsrc, acc);
            // (every element of this wave's row tile has been read: the sweep's last loads fed its last products)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = tr + 16 * ri + li, c = 16 * ci + lk + 4 * q;
                        if (r < m && c < nb) A[pk(r, jb + c, m)] = acc[ci][ri][q];
                    }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0 && bad) fail[dom] = 1; // per-domain flag: the host refactorises that domain (ddm_solver.cpp)
}

// The same factorisation for ONE large matrix (the coarse domain, m up to tens of thousands) spread over
// the chip, right-looking over 128-column panels so that the trailing matrix is read and written once per
// 128 columns (m/128 passes over m^2/2 doubles).  A panel is two 64-column halves; per half one workgroup
// factorises the 64 x 64 diagonal block and stores its inverse, a launch over the rows below multiplies
// them by that inverse; the second half is first updated with the first (K = 64); the trailing update
// uses both (K = 128).  All products run on the MFMA tiles of the per-domain kernel.
__global__ __launch_bounds__(256) void big_diag_kernel(double *A, int m, int jb, double *Linv, double *keep, int *fail) {
    __shared__ double Ld[CB][CB + 1];
    __shared__ double colj[CB];
    const int tid = threadIdx.x, nb = min(CB, m - jb);
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e & (CB - 1), c = e >> 6;
        Ld[r][c] = (r < nb && c < nb) ? (r >= c ? A[pk(jb + r, jb + c, m)] : 0.0) : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int c = 0; c < nb; ++c) {
        if (tid == 0) {
            const double dd = Ld[c][c];
            if (!(dd > 0.0)) fail[0] = 1;
            Ld[c][c] = sqrt(dd > 0.0 ? dd : 1.0);
        }
        __syncthreads();
        if (tid > c && tid < CB) Ld[tid][c] /= Ld[c][c];
        __syncthreads();
        {
            const int r = tid & (CB - 1);
            const double lrc = Ld[r][c];
            for (int c2 = c + 1 + (tid >> 6); c2 <= r; c2 += 4) Ld[r][c2] -= lrc * Ld[c2][c];
        }
        __syncthreads();
    }
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e & (CB - 1), c = e >> 6;
        if (r >= c && r < nb) A[pk(jb + r, jb + c, m)] = Ld[r][c];
    }
    __syncthreads();
    for (int j = CB - 1; j >= 0; --j) { // in-place inverse, column by column from the right
        if (tid < CB) colj[tid] = Ld[tid][j];
        __syncthreads();
        const double dj = 1.0 / colj[j];
        if (tid == j) Ld[j][j] = dj;
        if (tid > j && tid < CB) {
            double sacc = 0.0;
            for (int k = j + 1; k <= tid; ++k) sacc += Ld[tid][k] * colj[k];
            Ld[tid][j] = -dj * sacc;
        }
        __syncthreads();
    }
    for (int e = tid; e < CB * CB; e += 256) { // row-major [column of X][k]; kept per block for the substitutions
        const double x = Ld[e >> 6][e & (CB - 1)];
        Linv[e] = x;
        keep[static_cast<int64_t>(jb / CB) * (CB * CB) + e] = x;
    }
}

// rows r0 + 64 * (4 blockIdx.x + wave) ..: X = A[rows, jb..jb+64) inv(L11)^T, a row tile per wave (the block has 64
// columns: the launches below only come when rows lie under it)
__global__ __launch_bounds__(256, 2) void big_panel_kernel(double *A, int m, int jb, int r0, const double *__restrict__ Linv) {
    __shared__ double Ld[CB][CB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int tr = r0 + 64 * (4 * blockIdx.x + wave);
    for (int e = tid; e < CB * CB; e += 256) Ld[e >> 6][e & (CB - 1)] = Linv[e];
    __syncthreads();
    if (tr >= m) return; // (whole wave; no barrier below)
    d4 acc[4][4] = {};
    auto colsrc = [&](int ci, int k, const double *) { return Ld[16 * ci + li][k + lk]; };
    auto rowsrc = [&](int ri, int, const double *p) {
        const int r = tr + 16 * ri + li;
        const double x = p[min(r, m - 1)];
        return r < m ? x : 0.0;
    };
    wave_tile_sweep(A, m, jb, CB, lk, colsrc, rowsrc, acc);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = tr + 16 * ri + li, c = 16 * ci + lk + 4 * q;
                if (r < m) A[pk(r, jb + c, m)] = acc[ci][ri][q];
            }
}

// A[tile] -= L[rows, k0..k0+K) L[cols, k0..k0+K)^T for the 64 x 64 tiles (tr, tc), tr = r0 + 64 (4 bx + wave),
// tc = c0 + 64 by, tc <= tr, columns below c1 (K a multiple of 8); a tile per wave, the four waves of a workgroup
// share the column side through the cache
__global__ __launch_bounds__(256, 2) void big_syrk_kernel(double *A, int m, int k0, int K, int r0, int c0, int c1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int tr = r0 + 64 * (4 * blockIdx.x + wave), tc = c0 + 64 * blockIdx.y;
    if (tc > tr || tr >= m || tc >= c1) return;
    d4 acc[4][4] = {};
    auto colsrc = [&](int ci, int, const double *p) {
        const int r = tc + 16 * ci + li;
        const double x = p[min(r, c1 - 1)];
        return r < c1 ? x : 0.0;
    };
    auto rowsrc = [&](int ri, int, const double *p) {
        const int r = tr + 16 * ri + li;
        const double x = p[min(r, m - 1)];
        return r < m ? x : 0.0;
    };
    wave_tile_sweep(A, m, k0, K, lk, colsrc, rowsrc, acc);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = tr + 16 * ri + li, c = tc + 16 * ci + lk + 4 * q;
                if (r < m && c < c1 && r >= c) A[pk(r, c, m)] -= acc[ci][ri][q];
            }
}

// Domain::solve (domain.rs:393-475) for one right-hand side: gather, rhs = Q^T d_s + d_ns,
// L L^T gamma = rhs, lambda = [Q gamma; gamma], scatter.
__global__ __launch_bounds__(256) void ddm_solve_kernel(View v, const double *__restrict__ values,
                                                        double *__restrict__ out, int all_points) {
    const int dom = blockIdx.x;
    const int64_t o = v.dom_off[dom];
    const int n = static_cast<int>(v.dom_off[dom + 1] - o), k = v.k[dom], m = n - k;
    const double *Q = v.q + v.q_off[dom];
    const double *L = v.fac + v.fac_off[dom];
    double *w = v.work + o; // [d_s (k) | y (m)]
    double *y = w + k;
    __shared__ double yb[NB];
    __shared__ double Ld[NB][NB + 1];
    __shared__ double red[256];
    const int tid = threadIdx.x;
    for (int e = tid; e < n; e += 256) w[e] = values[v.gidx[o + e]];
    __threadfence_block();
    __syncthreads();
    for (int j = tid; j < m; j += 256) {
        double s = y[j];
        for (int a = 0; a < k; ++a) s += Q[a * m + j] * w[a];
        y[j] = s;
    }
    __threadfence_block();
    __syncthreads();
    if (v.mode && v.mode[dom]) {
        // the domain's Cholesky failed and the host stored the symmetric inverse instead (the reference
        // falls back to an LBL^T solve, domain.rs:60-75): gamma = inv * rhs on the packed lower triangle
        double *tp = v.tmp + o;
        for (int i = tid; i < m; i += 256) {
            double s = 0.0;
            for (int j = 0; j < i; ++j) s += L[pk(i, j, m)] * y[j];
            for (int j = i; j < m; ++j) s += L[pk(j, i, m)] * y[j];
            tp[i] = s;
        }
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < m; i += 256) y[i] = tp[i];
        __threadfence_block();
        __syncthreads();
    } else {
    // forward substitution L z = rhs, 32 columns at a time: the diagonal block goes to LDS and is solved
    // by one wave (column sweep, lane r owns row r), the rows below are updated by all threads
    for (int jb = 0; jb < m; jb += NB) {
        const int nb = min(NB, m - jb);
        for (int e = tid; e < nb * nb; e += 256) {
            const int r = e % nb, c = e / nb;
            Ld[r][c] = r >= c ? L[pk(jb + r, jb + c, m)] : 0.0;
        }
        if (tid < nb) yb[tid] = y[jb + tid];
        __syncthreads();
        if (tid < 64) {
            for (int c = 0; c < nb; ++c) {
                if (tid == c) yb[c] = yb[c] / Ld[c][c];
                __builtin_amdgcn_wave_barrier();
                if (tid > c && tid < nb) yb[tid] -= Ld[tid][c] * yb[c];
                __builtin_amdgcn_wave_barrier();
            }
            if (tid < nb) y[jb + tid] = yb[tid];
        }
        __syncthreads();
        for (int r = jb + nb + tid; r < m; r += 256) {
            double s = y[r];
            for (int c = 0; c < nb; ++c) s -= L[pk(r, jb + c, m)] * yb[c];
            y[r] = s;
        }
        __threadfence_block();
        __syncthreads();
    }
    // back substitution L^T gamma = z: a wave reduces the part below the block for its columns, then
    // one wave solves the transposed diagonal block (row sweep from the bottom)
    const int wave = tid >> 6, lane = tid & 63;
    for (int jb = ((m - 1) / NB) * NB; jb >= 0; jb -= NB) {
        const int nb = min(NB, m - jb);
        for (int e = tid; e < nb * nb; e += 256) {
            const int r = e % nb, c = e / nb;
            Ld[r][c] = r >= c ? L[pk(jb + r, jb + c, m)] : 0.0;
        }
        for (int c = wave; c < nb; c += 4) {
            double s = 0.0;
            for (int r = jb + nb + lane; r < m; r += 64) s += L[pk(r, jb + c, m)] * y[r];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) yb[c] = y[jb + c] - s;
        }
        __syncthreads();
        if (tid < 64) {
            for (int c = nb - 1; c >= 0; --c) {
                if (tid == c) yb[c] = yb[c] / Ld[c][c];
                __builtin_amdgcn_wave_barrier();
                if (tid < c) yb[tid] -= Ld[c][tid] * yb[c];
                __builtin_amdgcn_wave_barrier();
            }
            if (tid < nb) y[jb + tid] = yb[tid];
        }
        __threadfence_block();
        __syncthreads();
    }
    } // (Cholesky path)
    // lambda of the special points = Q gamma
    for (int a = 0; a < k; ++a) {
        double s = 0.0;
        for (int j = tid; j < m; j += 256) s += Q[a * m + j] * y[j];
        red[tid] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) w[a] = red[0];
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    for (int e = tid; e < n; e += 256)
        if (all_points || v.internal[o + e]) out[v.gidx[o + e]] = w[e];
}

// ---- Domain::solve for ONE large domain (the coarse level): the substitutions run as a sequence of
// launches over column blocks so that the whole chip streams the factor instead of one workgroup.
// Both sweeps are right-looking (a solved block updates the entries still to be solved), so every
// output has one writer and the result is deterministic.

__global__ __launch_bounds__(256) void big_gather_kernel(View v, const double *__restrict__ values, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) v.work[e] = values[v.gidx[e]];
}

__global__ __launch_bounds__(256) void big_rhs_kernel(View v, int k, int m) { // y += Q^T d_s
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double s = v.work[k + j];
    for (int a = 0; a < k; ++a) s += v.q[static_cast<int64_t>(a) * m + j] * v.work[a];
    v.work[k + j] = s;
}

// ---- substitutions of the one large domain in blocks of BB = 1024 (kBigSolveBlock): the 64-wide steps above are a chain
// of 2 m / 64 dependent launches of ~22 us each (13.7 ms at m = 19.7k, whatever launches them); with the inverses of the
// 1024 x 1024 diagonal blocks a block step is two launches -- a triangular product with the inverse, then the update of
// everything below (above) -- and the chain is 4 m / 1024 launches over the same 2 x 1.5 GB of factor.
constexpr int BB = kBigSolveBlock;

// X = T^{-1} for the diagonal block T = L[j0.., j0..] (nb x nb, nb <= BB), one thread per column: row i of the
// forward substitution on the unit vector; X is stored row-major (X[i][c] at i * BB + c), zero above the diagonal.
// Every thread walks the same (i, k) -- rows above its own diagonal element come out as the zeros they are -- so that
// T(i, k) is one (scalar) address per step and X[k][.] one coalesced row.
__global__ __launch_bounds__(256) void big_block_inverse_kernel(const double *__restrict__ L, int m, double *__restrict__ binv) {
    const int blk = blockIdx.x, j0 = blk * BB, nb = min(BB, m - j0);
    const int c = blockIdx.y * 256 + threadIdx.x;
    double *X = binv + static_cast<int64_t>(blk) * BB * BB;
    const int cfirst = blockIdx.y * 256; // rows above the workgroup's first column are zero for all of its columns
    for (int i = 0; i < cfirst; ++i) X[static_cast<int64_t>(i) * BB + c] = 0.0;
    for (int i = cfirst; i < nb; ++i) {
        const double *t = L + pk(j0 + i, j0 + cfirst, m); // T(i, cfirst), then along the row: + (m - column - 1)
        double sacc[8] = {i == c ? 1.0 : 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int k = cfirst, col = j0 + cfirst;
        for (; k + 7 < i; k += 8) { // eight independent loads and sums in flight
            double tv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                tv[u] = t[0];
                t += m - col - 1;
                ++col;
                xv[u] = X[static_cast<int64_t>(k + u) * BB + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sacc[u] -= tv[u] * xv[u];
        }
        for (; k < i; ++k) {
            sacc[0] -= t[0] * X[static_cast<int64_t>(k) * BB + c];
            t += m - col - 1;
            ++col;
        }
        const double s0 = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]), s1 = (sacc[4] + sacc[5]) + (sacc[6] + sacc[7]);
        // (t now points at T(i, i))
        X[static_cast<int64_t>(i) * BB + c] = (c < nb && i >= c) ? (s0 + s1) / t[0] : 0.0;
    }
    for (int i = nb; i < BB; ++i) X[static_cast<int64_t>(i) * BB + c] = 0.0;
}

// z[j0 + r] = sum_{k <= r} X[r][k] y[j0 + k]: 64 rows per workgroup, 16 per wave, the lanes run along a row
__global__ __launch_bounds__(256) void big_blk_fwd_solve_kernel(const double *__restrict__ binv, int m, int j0,
                                                                const double *__restrict__ y, double *__restrict__ z) {
    __shared__ double ys[BB];
    const int nb = min(BB, m - j0), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *X = binv + static_cast<int64_t>(j0 / BB) * BB * BB;
    for (int e = tid; e < BB; e += 256) ys[e] = e < nb ? y[j0 + e] : 0.0;
    __syncthreads();
    for (int q = 0; q < 16; ++q) {
        const int r = blockIdx.x * 64 + wave * 16 + q;
        if (r >= nb) break;
        double sacc = 0.0;
        for (int k = lane; k <= r; k += 64) sacc += X[static_cast<int64_t>(r) * BB + k] * ys[k];
        for (int off = 32; off > 0; off >>= 1) sacc += __shfl_down(sacc, off, 64);
        if (lane == 0) z[j0 + r] = sacc;
    }
}

// y[r] -= sum_c L[r, j0 + c] z[j0 + c] for the rows r >= j0 + nb: 64 rows per workgroup (the lanes), the block's columns
// split over sixteen waves (64 each); partial sums meet in LDS in a fixed order
__global__ __launch_bounds__(1024) void big_blk_fwd_update_kernel(const double *__restrict__ L, int m, int j0, double *y,
                                                                  const double *__restrict__ z) {
    __shared__ double zs[BB];
    __shared__ double part[16][64];
    const int nb = min(BB, m - j0), tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    for (int e = tid; e < BB; e += 1024) zs[e] = e < nb ? z[j0 + e] : 0.0;
    __syncthreads();
    const int r = j0 + nb + blockIdx.x * 64 + lane;
    double sacc[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < m) {
        const int c0 = q * 64, c1 = min(nb, c0 + 64);
        const double *p = L + pk(r, j0 + c0, m); // along the row: + (m - column - 1)
        int col = j0 + c0;
        for (int c = c0; c < c1; ++c) {
            sacc[c & 3] += p[0] * zs[c];
            p += m - col - 1;
            ++col;
        }
    }
    part[q][lane] = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
    __syncthreads();
    if (q == 0 && r < m) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][lane];
        y[r] -= t;
    }
}

// g[j0 + c] = sum_{i >= c} X[i][c] z[j0 + i]: 64 columns per workgroup (the lanes), the rows split over sixteen waves
__global__ __launch_bounds__(1024) void big_blk_bwd_solve_kernel(const double *__restrict__ binv, int m, int j0,
                                                                 const double *__restrict__ z, double *__restrict__ g) {
    __shared__ double zs[BB];
    __shared__ double part[16][64];
    const int nb = min(BB, m - j0), tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const double *X = binv + static_cast<int64_t>(j0 / BB) * BB * BB;
    for (int e = tid; e < BB; e += 1024) zs[e] = e < nb ? z[j0 + e] : 0.0;
    __syncthreads();
    const int c = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0;
    {
        // rows blockIdx.x * 64 .. nb (above them X is zero for these columns), dealt to the waves in runs of 64
        const int i_lo = blockIdx.x * 64 + q * 64;
        for (int i0 = i_lo; i0 < nb; i0 += 1024) {
            const int i1 = min(nb, i0 + 64);
            int i = i0;
            for (; i + 1 < i1; i += 2) {
                s0 += X[static_cast<int64_t>(i) * BB + c] * zs[i];
                s1 += X[static_cast<int64_t>(i + 1) * BB + c] * zs[i + 1];
            }
            if (i < i1) s0 += X[static_cast<int64_t>(i) * BB + c] * zs[i];
        }
    }
    part[q][lane] = s0 + s1;
    __syncthreads();
    if (q == 0 && c < nb) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][lane];
        g[j0 + c] = t;
    }
}

// z[c] -= sum_i L[j0 + i, c] g[j0 + i] for the columns c < j0: 64 columns per workgroup, 4 per wave (sixteen waves); the
// lanes run down the block's rows (contiguous in a packed column)
__global__ __launch_bounds__(1024) void big_blk_bwd_update_kernel(const double *__restrict__ L, int m, int j0, double *z,
                                                                  const double *__restrict__ g) {
    __shared__ double gs[BB];
    const int nb = min(BB, m - j0), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < BB; e += 1024) gs[e] = e < nb ? g[j0 + e] : 0.0;
    __syncthreads();
    const int cbeg = blockIdx.x * 64 + wave * 4, cend = min(j0, cbeg + 4);
    for (int c = cbeg; c < cend; ++c) {
        const double *col = L + pk(j0, c, m); // rows j0 .. of column c are contiguous
        double s0 = 0.0, s1 = 0.0;
        int i = lane;
        for (; i + 64 < nb; i += 128) {
            s0 += col[i] * gs[i];
            s1 += col[i + 64] * gs[i + 64];
        }
        if (i < nb) s0 += col[i] * gs[i];
        double sacc = s0 + s1;
        for (int off = 32; off > 0; off >>= 1) sacc += __shfl_down(sacc, off, 64);
        if (lane == 0) z[c] -= sacc;
    }
}

__global__ __launch_bounds__(256) void big_special_kernel(View v, int m, const double *__restrict__ g) { // Q gamma
    __shared__ double red[256];
    const int a = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int j = tid; j < m; j += 256) s += v.q[static_cast<int64_t>(a) * m + j] * g[j];
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) v.work[a] = red[0];
}

__global__ __launch_bounds__(256) void big_scatter_kernel(View v, int n, int k, const double *__restrict__ g,
                                                          double *__restrict__ out, int all_points) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    if (all_points || v.internal[e]) out[v.gidx[e]] = e < k ? v.work[e] : g[e - k];
}

template <class F> void dispatch_rbf_kernel(int id, F &&f) { // the kernels the solver supports (interpolant_config.rs:30-36)
    switch (id) {
    case kLinear: f(std::integral_constant<int, kLinear>{}); break;
    case kThinPlateSpline: f(std::integral_constant<int, kThinPlateSpline>{}); break;
    case kCubic: f(std::integral_constant<int, kCubic>{}); break;
    case kSpheroidal3: f(std::integral_constant<int, kSpheroidal3>{}); break;
    case kSpheroidal5: f(std::integral_constant<int, kSpheroidal5>{}); break;
    case kSpheroidal7: f(std::integral_constant<int, kSpheroidal7>{}); break;
    case kSpheroidal9: f(std::integral_constant<int, kSpheroidal9>{}); break;
    default: break;
    }
}

} // namespace

void launch_ddm_prep(const KernelSpec &ks, double nugget, int d, const DdmLevelSolver &lv, hipStream_t s) {
    (void)d;
    if (lv.n_dom == 0 || !lv.d_q) return;
    const View v = make_view(lv);
    dispatch_rbf_kernel(ks.id, [&](auto idc) {
        constexpr int ID = decltype(idc)::value;
        hipLaunchKernelGGL((ddm_prep_kernel<ID>), dim3(static_cast<unsigned>(lv.n_dom)), dim3(256), 0, s, ks, nugget, v);
    });
}

void launch_ddm_assemble(const KernelSpec &ks, double nugget, int d, const DdmLevelSolver &lv, hipStream_t s) {
    (void)d;
    if (lv.n_dom == 0) return;
    const View v = make_view(lv);
    const unsigned gy = lv.n_dom >= 512 ? 4 : 64; // few domains (the coarse one): split the columns wider
    dispatch_rbf_kernel(ks.id, [&](auto idc) {
        constexpr int ID = decltype(idc)::value;
        hipLaunchKernelGGL((ddm_assemble_kernel<ID>), dim3(static_cast<unsigned>(lv.n_dom), gy), dim3(256), 0, s, ks,
                           nugget, v);
    });
}

void launch_ddm_cholesky(const DdmLevelSolver &lv, int *d_fail, hipStream_t s) {
    if (lv.n_dom == 0) return;
    if (ddm_level_is_big(lv)) { // one large matrix: the whole chip per step (d_work is free here: the inverse goes there)
        const int m = lv.max_m;
        double *A = lv.d_fac, *Linv = lv.d_work;
        auto tiles = [](int rows) { return static_cast<unsigned>((rows + 63) / 64); };
        auto tiles4 = [](int rows) { return static_cast<unsigned>((rows + 255) / 256); }; // a 64-row tile per wave
        for (int jb = 0; jb < m; jb += 2 * CB) {
            const int h1 = jb + CB, end = std::min(m, jb + 2 * CB); // second half [h1, end), trailing part from end
            hipLaunchKernelGGL(big_diag_kernel, dim3(1), dim3(256), 0, s, A, m, jb, Linv, lv.d_linv, d_fail);
            if (h1 >= m) break;
            hipLaunchKernelGGL(big_panel_kernel, dim3(tiles4(m - h1)), dim3(256), 0, s, A, m, jb, h1, Linv);
            // second half of the panel brought up to date with the first (K = 64)
            hipLaunchKernelGGL(big_syrk_kernel, dim3(tiles4(m - h1), 1), dim3(256), 0, s, A, m, jb, CB, h1, h1, end);
            hipLaunchKernelGGL(big_diag_kernel, dim3(1), dim3(256), 0, s, A, m, h1, Linv, lv.d_linv, d_fail);
            if (end >= m) break;
            hipLaunchKernelGGL(big_panel_kernel, dim3(tiles4(m - end)), dim3(256), 0, s, A, m, h1, end, Linv);
            const unsigned t = tiles(m - end);
            hipLaunchKernelGGL(big_syrk_kernel, dim3(tiles4(m - end), t), dim3(256), 0, s, A, m, jb, 2 * CB, end, end, m);
        }
        return;
    }
    hipLaunchKernelGGL(ddm_cholesky_kernel, dim3(static_cast<unsigned>(lv.n_dom)), dim3(256), 0, s, make_view(lv), d_fail);
}

void launch_ddm_big_block_inverses(const DdmLevelSolver &lv, hipStream_t s) {
    if (!ddm_level_is_big(lv) || !lv.d_binv) return;
    const int m = lv.max_m;
    hipLaunchKernelGGL(big_block_inverse_kernel, dim3((m + BB - 1) / BB, BB / 256), dim3(256), 0, s, lv.d_fac, m, lv.d_binv);
}

__global__ __launch_bounds__(256) void unpack_symmetric_kernel(const double *__restrict__ packed, int m, double *__restrict__ full) {
    const int c = blockIdx.x;
    for (int r = c + threadIdx.x; r < m; r += 256) {
        const double v = packed[pk(r, c, m)];
        full[static_cast<int64_t>(c) * m + r] = v;
        full[static_cast<int64_t>(r) * m + c] = v;
    }
}
void launch_ddm_unpack_symmetric(const double *packed, int m, double *full, hipStream_t s) {
    if (m > 0) hipLaunchKernelGGL(unpack_symmetric_kernel, dim3(m), dim3(256), 0, s, packed, m, full);
}

// Returns BBFMM_OK, or the error of the pivoted-LU fallback (rocSOLVER) -- the scatter is then skipped, so that an
// unsolved right-hand side never reaches the correction.
int launch_ddm_solve(const DdmLevelSolver &lv, const double *d_values, double *d_out, bool all_points, hipStream_t s) {
    if (lv.n_dom == 0) return BBFMM_OK;
    if (ddm_level_is_big(lv)) { // one large domain: work = [d_s | y] (n), z (m), gamma (m)
        const View v = make_view(lv);
        const int n = static_cast<int>(lv.n_entries), k = lv.k[0], m = n - k;
        double *y = lv.d_work + k, *z = lv.d_work + n, *g = z + m;
        const double *L = lv.d_fac;
        hipLaunchKernelGGL(big_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, v, d_values, n);
        if (k) hipLaunchKernelGGL(big_rhs_kernel, dim3((m + 255) / 256), dim3(256), 0, s, v, k, m);
        if (lv.d_lu) { // not positive definite: pivoted LU of the full matrix (the reference's LBL^T role)
            if (hipMemcpyAsync(g, y, static_cast<size_t>(m) * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
                return BBFMM_DEVICE_ERROR;
            const int lrc = big_lu_solve(lv, g, s);
            if (lrc != BBFMM_OK) return lrc;
        } else {
            for (int j0 = 0; j0 < m; j0 += BB) { // L z = y
                const int nb = std::min(BB, m - j0), rest = m - j0 - nb;
                hipLaunchKernelGGL(big_blk_fwd_solve_kernel, dim3((nb + 63) / 64), dim3(256), 0, s, lv.d_binv, m, j0, y, z);
                if (rest > 0)
                    hipLaunchKernelGGL(big_blk_fwd_update_kernel, dim3((rest + 63) / 64), dim3(1024), 0, s, L, m, j0, y, z);
            }
            for (int j0 = ((m - 1) / BB) * BB; j0 >= 0; j0 -= BB) { // L^T g = z
                const int nb = std::min(BB, m - j0);
                hipLaunchKernelGGL(big_blk_bwd_solve_kernel, dim3((nb + 63) / 64), dim3(1024), 0, s, lv.d_binv, m, j0, z, g);
                if (j0 > 0)
                    hipLaunchKernelGGL(big_blk_bwd_update_kernel, dim3((j0 + 63) / 64), dim3(1024), 0, s, L, m, j0, z, g);
            }
        }
        if (k) hipLaunchKernelGGL(big_special_kernel, dim3(k), dim3(256), 0, s, v, m, g);
        hipLaunchKernelGGL(big_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, v, n, k, g, d_out, all_points ? 1 : 0);
        return BBFMM_OK;
    }
    hipLaunchKernelGGL(ddm_solve_kernel, dim3(static_cast<unsigned>(lv.n_dom)), dim3(256), 0, s, make_view(lv), d_values,
                       d_out, all_points ? 1 : 0);
    return BBFMM_OK;
}

} // namespace bbfmm

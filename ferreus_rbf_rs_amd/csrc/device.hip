// Hand-written CDNA4 (gfx950) kernels for the BBFMM matvec.  See device.hpp for the
// HBM layout.  Wavefront = 64 everywhere; FP64 throughout (the reference is f64 end
// to end, ferreus_bbfmm/src/traits.rs:20).
//
//   gather/scatter  HBM streaming
//   P2M / L2P       Chebyshev anterpolation / interpolation (chebyshev.rs:831-927),
//                   tensor factors staged in LDS
//   M2M / L2L       sum-factorised 1-D transfers (the reference multiplies by the
//                   dense Kronecker matrix, bbfmm.rs:742-772,1051-1086; same operator)
//   M2L             two batched small-GEMM stages on v_mfma_f64_16x16x4_f64
//   P2P/M2P/P2L     direct kernel evaluation, LDS-tiled sources, lanes = target x slice
#include "device.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace bbfmm {

typedef double v4f64 __attribute__((ext_vector_type(4)));

struct Xyz {
    const double *x, *y, *z;
};

// ------------------------------------------------------------------ gather/scatter
__global__ void gather_weights_kernel(const double *__restrict__ w, int64_t ldw, const int32_t *__restrict__ order,
                                      int64_t N, double *__restrict__ ws) {
    const int k = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        ws[k * N + i] = w[k * ldw + order[i]];
}

__global__ void scatter_output_kernel(const double *__restrict__ os, int64_t n, const int32_t *__restrict__ perm,
                                      double *__restrict__ out, int64_t ldo, int accumulate) {
    const int k = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t dst = k * ldo + perm[i];
        if (accumulate)
            out[dst] += os[k * n + i];
        else
            out[dst] = os[k * n + i];
    }
}

__global__ void gather_rows_kernel(const double *__restrict__ src, int64_t ld_src, const int32_t *__restrict__ idx,
                                   int64_t n, double *__restrict__ dst, int64_t ld_dst) {
    const int c = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[c * ld_dst + i] = src[c * ld_src + idx[i]];
}

static inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g > 2048) g = 2048; // 256 CUs x 8 blocks, grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

// the same for a subset of the sorted positions (a partition reads the weights of its subtree and halo only)
__global__ void gather_weights_subset_kernel(const double *__restrict__ w, int64_t ldw, const int32_t *__restrict__ order,
                                             const int32_t *__restrict__ pos, int64_t n_pos, int64_t N, double *__restrict__ ws) {
    const int k = blockIdx.y;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_pos; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t i = pos[t];
        ws[k * N + i] = w[k * ldw + order[i]];
    }
}
void launch_gather_weights_subset(const double *w, int64_t ldw, int K, const int32_t *order, const int32_t *pos, int64_t n_pos,
                                  int64_t N, double *w_sorted, hipStream_t s) {
    if (n_pos == 0) return;
    hipLaunchKernelGGL(gather_weights_subset_kernel, dim3(grid_for(n_pos, 256), K), dim3(256), 0, s, w, ldw, order, pos, n_pos, N,
                       w_sorted);
}

void launch_gather_weights(const double *w, int64_t ldw, int K, const int32_t *order, int64_t N, double *w_sorted,
                           hipStream_t s) {
    if (N == 0) return;
    hipLaunchKernelGGL(gather_weights_kernel, dim3(grid_for(N, 256), K), dim3(256), 0, s, w, ldw, order, N, w_sorted);
}
__global__ void scatter_parts_kernel(const double *__restrict__ all, ScatterParts parts, int64_t m_max, int K,
                                     const int32_t *__restrict__ order, double *__restrict__ out, int64_t ldo) {
    const int k = blockIdx.y;
    const int64_t n = parts.bound[parts.n] - parts.bound[0];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = parts.bound[0] + i; // sorted position
        int r = 0;
        while (r + 1 < parts.n && g >= parts.bound[r + 1]) ++r;
        out[k * ldo + order[g]] = all[((int64_t)r * K + k) * m_max + (g - parts.bound[r])];
    }
}
void launch_scatter_parts(const double *all, const ScatterParts &parts, int64_t m_max, int K, const int32_t *order, double *out,
                          int64_t ldo, hipStream_t s) {
    const int64_t n = parts.bound[parts.n] - parts.bound[0];
    if (n <= 0) return;
    hipLaunchKernelGGL(scatter_parts_kernel, dim3(grid_for(n, 256), K), dim3(256), 0, s, all, parts, m_max, K, order, out, ldo);
}

// out[i] = slots[0][i] + slots[1][i] + ... in that order (the coarse multipoles of a device group: every device adds the
// parts' partial sums in the same fixed order, so all of them hold the same bits)
__global__ void sum_slots_kernel(const double *__restrict__ slots, int n_slots, int64_t len, double *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
        double acc = slots[i];
        for (int g = 1; g < n_slots; ++g) acc += slots[(int64_t)g * len + i];
        out[i] = acc;
    }
}
void launch_sum_slots(const double *slots, int n_slots, int64_t len, double *out, hipStream_t s) {
    if (len <= 0 || n_slots < 1) return;
    hipLaunchKernelGGL(sum_slots_kernel, dim3(grid_for(len, 256)), dim3(256), 0, s, slots, n_slots, len, out);
}

void launch_scatter_output(const double *out_sorted, int64_t n, int K, const int32_t *perm, double *out, int64_t ldo,
                           int accumulate, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(scatter_output_kernel, dim3(grid_for(n, 256), K), dim3(256), 0, s, out_sorted, n, perm, out,
                       ldo, accumulate);
}
void launch_gather_rows(const double *src, int64_t ld_src, int ncols, const int32_t *idx, int64_t n, double *dst,
                        int64_t ld_dst, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n, 256), ncols), dim3(256), 0, s, src, ld_src, idx, n, dst,
                       ld_dst);
}

// ------------------------------------------------------------------ Chebyshev helpers
// Per-axis node counts: axes >= d have a single node with S = 1, so 1-D/2-D trees run
// through the same 3-D index arithmetic (node index = (i0*P1 + i1)*P2 + i2).
__device__ inline void axis_sizes(int p, int d, int &P0, int &P1, int &P2) {
    P0 = p;
    P1 = d > 1 ? p : 1;
    P2 = d > 2 ? p : 1;
}

// S_j(x) = (2 sum_k T_k(x) T_k(node_j) - 1)/p ; dS_j = (2/p) sum_k T'_k(x) T_k(node_j)
// (chebyshev.rs:47-142).  Results go to S[j*stride] (thread-private LDS columns).
template <bool GRAD>
__device__ inline void cheb_S_to(int p, double x, const double *__restrict__ polyn, double *S, double *dS, int stride) {
    double T[kMaxOrder], dT[kMaxOrder];
    T[0] = 1.0;
    dT[0] = 0.0;
    T[1] = x;
    dT[1] = 1.0;
#pragma unroll
    for (int j = 2; j < kMaxOrder; ++j) {
        if (j < p) {
            T[j] = 2.0 * x * T[j - 1] - T[j - 2];
            if (GRAD) dT[j] = 2.0 * T[j - 1] + 2.0 * x * dT[j - 1] - dT[j - 2];
        }
    }
    for (int j = 0; j < p; ++j) {
        double s = 0.0, ds = 0.0;
#pragma unroll
        for (int k = 0; k < kMaxOrder; ++k) {
            if (k < p) {
                const double pk = polyn[j * p + k];
                s += T[k] * pk;
                if (GRAD) ds += dT[k] * pk;
            }
        }
        S[j * stride] = (s * 2.0 - 1.0) / (double)p;
        if (GRAD) dS[j * stride] = ds * (2.0 / (double)p);
    }
}

// ------------------------------------------------------------------ Chebyshev factors in registers
template <int P, bool GRAD>
__device__ inline void cheb_S_reg(double x, const double *__restrict__ polyn, double (&S)[P], double (&dS)[P]) {
    double T[P], dT[P];
    T[0] = 1.0;
    dT[0] = 0.0;
    if (P > 1) {
        T[1] = x;
        dT[1] = 1.0;
    }
#pragma unroll
    for (int j = 2; j < P; ++j) {
        T[j] = 2.0 * x * T[j - 1] - T[j - 2];
        if (GRAD) dT[j] = 2.0 * T[j - 1] + 2.0 * x * dT[j - 1] - dT[j - 2];
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double s = 0.0, ds = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const double pk = polyn[j * P + k];
            s += T[k] * pk;
            if (GRAD) ds += dT[k] * pk;
        }
        S[j] = (s * 2.0 - 1.0) / (double)P;
        dS[j] = GRAD ? ds * (2.0 / (double)P) : 0.0;
    }
}

// ------------------------------------------------------------------ P2M
// particle_to_multipole (bbfmm.rs:691-739): M_c[:, k] += S(x_leaf)^T w_leaf[:, k].
// One wave per leaf.  Points are taken 32 at a time: lanes compute the three 1-D factor rows
// of their point (registers, order P is a template parameter) and park them in a wave-private
// LDS slice; then lane q owns the node pairs (i1, i2) = q and keeps the P sums over i0 in
// registers, so a point costs two private LDS reads plus P broadcast reads per lane.
constexpr int P2M_WAVES = 4;
// 3-D orders above 12 (they work, slowly: bbfmm.rs:77-104 takes any order): two waves per workgroup
template <int P, int D> constexpr int p2m_waves() { return D == 3 && P > 12 ? 2 : P2M_WAVES; }
constexpr int P2M_PTS = 64; // measured at 10M points, K = 1: 1.12 ms against 1.38 ms with 32
                            // (one batch covers a 38-point leaf) and 1.55 ms with two rhs slots

template <int P, int D, int P2M_KB>
__global__ __launch_bounds__((64 * p2m_waves<P, D>())) void p2m_kernel(const DevCheb *__restrict__ chp, int n_leaves, Xyz src,
                                                             const double *__restrict__ ws, int64_t N, int K,
                                                             int64_t C, const int32_t *__restrict__ leaf_cells,
                                                             const int32_t *__restrict__ pt_begin,
                                                             const int32_t *__restrict__ pt_end,
                                                             const double *__restrict__ centers,
                                                             const double *__restrict__ lengths,
                                                             double *__restrict__ M) {
    constexpr int P1 = D > 1 ? P : 1, P2 = D > 2 ? P : 1, NPAIR = P1 * P2;
    constexpr int NPASS = (NPAIR + 63) / 64;
    constexpr int SROW = 3 * P + P2M_KB; // per point: S0[P], S1[P], S2[P], w[KB]
    __shared__ double s_polyn[P * P];
    constexpr int WAVES = p2m_waves<P, D>();
    __shared__ double s_pts[WAVES][P2M_PTS][SROW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < P * P; i += 64 * WAVES) s_polyn[i] = chp->polyn[i];
    __syncthreads();
    const int job = blockIdx.x * WAVES + wave;
    if (job >= n_leaves) return; // whole wave; no block barrier below
    const int n_pad = chp->n_pad;
    const int cell = leaf_cells[job];
    const int b = pt_begin[cell], e = pt_end[cell];
    const double len = lengths[cell];
    const double cc[3] = {centers[cell * 3 + 0], centers[cell * 3 + 1], centers[cell * 3 + 2]};
    double(*sp)[SROW] = s_pts[wave];
    for (int k0 = 0; k0 < K; k0 += P2M_KB) {
        double acc[NPASS][P2M_KB][P];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
            for (int kk = 0; kk < P2M_KB; ++kk)
#pragma unroll
                for (int i0 = 0; i0 < P; ++i0) acc[ps][kk][i0] = 0.0;
        for (int base = b; base < e; base += P2M_PTS) {
            const int npts = min(P2M_PTS, e - base);
            if (lane < npts) {
                const int pt = base + lane;
                double S[P], dS[P];
                cheb_S_reg<P, false>((src.x[pt] - cc[0]) / (len * 0.5), s_polyn, S, dS); // chebyshev.rs:841-845
#pragma unroll
                for (int jx = 0; jx < P; ++jx) sp[lane][jx] = S[jx];
                if (D > 1) {
                    cheb_S_reg<P, false>((src.y[pt] - cc[1]) / (len * 0.5), s_polyn, S, dS);
#pragma unroll
                    for (int jx = 0; jx < P; ++jx) sp[lane][P + jx] = S[jx];
                } else {
                    sp[lane][P] = 1.0;
                }
                if (D > 2) {
                    cheb_S_reg<P, false>((src.z[pt] - cc[2]) / (len * 0.5), s_polyn, S, dS);
#pragma unroll
                    for (int jx = 0; jx < P; ++jx) sp[lane][2 * P + jx] = S[jx];
                } else {
                    sp[lane][2 * P] = 1.0;
                }
#pragma unroll
                for (int kk = 0; kk < P2M_KB; ++kk)
                    sp[lane][3 * P + kk] = (k0 + kk < K) ? ws[(int64_t)(k0 + kk) * N + pt] : 0.0;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): wave-private slice, in-order LDS
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int q = lane + 64 * ps;
                if (q < NPAIR) {
                    const int i1 = q / P2, i2 = q - i1 * P2;
                    for (int pt = 0; pt < npts; ++pt) {
                        const double *row = sp[pt];
                        const double s12 = row[P + i1] * row[2 * P + i2];
#pragma unroll
                        for (int kk = 0; kk < P2M_KB; ++kk) {
                            const double v = s12 * row[3 * P + kk];
#pragma unroll
                            for (int i0 = 0; i0 < P; ++i0) acc[ps][kk][i0] += v * row[i0];
                        }
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // reads done before the slice is rewritten
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int q = lane + 64 * ps;
            if (q < NPAIR) {
#pragma unroll
                for (int kk = 0; kk < P2M_KB; ++kk) {
                    if (k0 + kk < K) {
                        double *Mc = M + ((int64_t)(k0 + kk) * C + cell) * n_pad + q;
#pragma unroll
                        for (int i0 = 0; i0 < P; ++i0) Mc[i0 * NPAIR] = acc[ps][kk][i0]; // a leaf is written once
                    }
                }
            }
        }
    }
}

// P2M as a small matrix product on the FP64 matrix pipe (round 4; 3-D, orders up to 10).  Per leaf
//   M[i0][q] = sum_p (S0[p][i0] w_p) * (S1[p][i1(q)] S2[p][i2(q)]),   q = i1 * P + i2
// is (P x npts) x (npts x P^2).  The kernel above evaluates it with lane = q and one point at a time: two private and
// four broadcast LDS reads for nine multiply-adds per point and lane, which keeps the CU's one LDS pipe busy for four
// waves' worth of arithmetic (VALUBusy 46 %).  Here four points form the contraction of a v_mfma_f64_4x4x4 (four
// independent 4 x 4 x 4 products per instruction, lanes A: 16 k + 4 b + i, B: 16 k + 4 b + j, D: 16 i + 4 b + j):
// row block rb holds i0 = 4 rb + i, column group g holds q = 16 g + 4 b + j; per step of four points a lane reads one
// S0 value per row block and an S1 / S2 pair per column group (private LDS reads, no broadcasts) and issues RB x G
// matrix instructions.  Points beyond the leaf are zero rows.  Same sums as the kernel above in another order.
template <int P>
__global__ __launch_bounds__(64 * P2M_WAVES) void p2m_mfma_kernel(const DevCheb *__restrict__ chp, int n_leaves, Xyz src,
                                                                 const double *__restrict__ ws, int64_t N, int K, int64_t C,
                                                                 const int32_t *__restrict__ leaf_cells,
                                                                 const int32_t *__restrict__ pt_begin,
                                                                 const int32_t *__restrict__ pt_end,
                                                                 const double *__restrict__ centers,
                                                                 const double *__restrict__ lengths, double *__restrict__ M) {
    constexpr int NPAIR = P * P, RB = (P + 3) / 4, G = (NPAIR + 15) / 16;
    constexpr int SROW = 3 * P + 2; // per point: S0[P], S1[P], S2[P], w, one spare (rows start 16-byte aligned for even P)
    __shared__ double s_pts[P2M_WAVES][64][SROW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (the node values T_k(node_j) come through the scalar cache as SGPR operands: the same 8 P^2 bytes for every wave;
    // staged in LDS the compiler kept all P^2 of them in vector registers across the three axes)
    const double *__restrict__ s_polyn = chp->polyn;
    const int job = blockIdx.x * P2M_WAVES + wave;
    if (job >= n_leaves) return; // whole wave; no block barrier below
    const int n_pad = chp->n_pad;
    const int cell = leaf_cells[job];
    const int b = pt_begin[cell], e = pt_end[cell];
    const double len = lengths[cell];
    const double cc[3] = {centers[cell * 3 + 0], centers[cell * 3 + 1], centers[cell * 3 + 2]};
    double(*sp)[SROW] = s_pts[wave];
    const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
    // this lane's column of every group (clamped: columns past P^2 read valid memory and are never stored)
    int o1[G], o2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int q = min(16 * g + 4 * blk + lo, NPAIR - 1);
        o1[g] = P + q / P;
        o2[g] = 2 * P + q % P;
    }
    for (int k = 0; k < K; ++k) {
        double acc[RB][G];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[rb][g] = 0.0;
        for (int base = b; base < e; base += 64) {
            const int npts = min(64, e - base);
            if (k == 0 || e - b > 64) { // the factor rows of the batch (kept across right-hand sides when the leaf is one batch)
                double S[P], dS[P];
                const int pt = base + min(lane, npts - 1);
                const bool live = lane < npts;
                cheb_S_reg<P, false>((src.x[pt] - cc[0]) / (len * 0.5), s_polyn, S, dS); // chebyshev.rs:841-845
#pragma unroll
                for (int jx = 0; jx < P; ++jx) sp[lane][jx] = live ? S[jx] : 0.0;
                cheb_S_reg<P, false>((src.y[pt] - cc[1]) / (len * 0.5), s_polyn, S, dS);
#pragma unroll
                for (int jx = 0; jx < P; ++jx) sp[lane][P + jx] = live ? S[jx] : 0.0;
                cheb_S_reg<P, false>((src.z[pt] - cc[2]) / (len * 0.5), s_polyn, S, dS);
#pragma unroll
                for (int jx = 0; jx < P; ++jx) sp[lane][2 * P + jx] = live ? S[jx] : 0.0;
            }
            sp[lane][3 * P] = lane < npts ? ws[(int64_t)k * N + base + lane] : 0.0;
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): wave-private slice, in-order LDS
            const int nsteps = (npts + 3) >> 2;
            for (int st = 0; st < nsteps; ++st) {
                const double *row = sp[4 * st + hi];
                const double wv = row[3 * P];
                double av[RB], bv[G];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) av[rb] = 4 * rb + lo < P ? row[4 * rb + lo] * wv : 0.0;
#pragma unroll
                for (int g = 0; g < G; ++g) bv[g] = row[o1[g]] * row[o2[g]];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[rb][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[rb], bv[g], acc[rb][g], 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // reads done before the slice is rewritten
        }
        double *Mc = M + ((int64_t)k * C + cell) * n_pad;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i0 = 4 * rb + hi; // D: lane = 16 i + 4 b + j
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int q = 16 * g + 4 * blk + lo;
                if (i0 < P && q < NPAIR) Mc[i0 * NPAIR + q] = acc[rb][g]; // a leaf is written once
            }
        }
    }
}

// ------------------------------------------------------------------ M2M / L2L
// One pass of a sum-factorised transfer along `axis`.  in/out are n-vectors in LDS with
// index (i0*P1 + i1)*P2 + i2.  FORWARD (M2M): out[.., i, ..] = sum_a xf[a][i] in[.., a, ..];
// transposed (L2L): out[.., a, ..] = sum_i xf[a][i] in[.., i, ..].  xf = xfer[side] (p x p).
template <bool FORWARD>
__device__ inline void transfer_pass(const double *in, double *out, const double *xf, int axis, int p, int P0, int P1,
                                     int P2, int n, int tid, int nthreads) {
    const int stride = axis == 0 ? P1 * P2 : (axis == 1 ? P2 : 1);
    for (int I = tid; I < n; I += nthreads) {
        const int ia = (I / stride) % p;
        const int base = I - ia * stride;
        double s = 0.0;
        for (int a = 0; a < p; ++a) {
            const double f = FORWARD ? xf[a * p + ia] : xf[ia * p + a];
            s += f * in[base + a * stride];
        }
        out[I] = s;
    }
}

// multipole_to_multipole (bbfmm.rs:742-772): one workgroup per parent.
__global__ __launch_bounds__(256) void m2m_kernel(const DevCheb *__restrict__ chp, int K, int64_t C,
                                                  const int32_t *__restrict__ parents,
                                                  const int64_t *__restrict__ child_ptr,
                                                  const int32_t *__restrict__ child_idx,
                                                  const int32_t *__restrict__ octant, double *__restrict__ M) {
    extern __shared__ double lds[];
    const int p = chp->p, d = chp->d, n = chp->n, n_pad = chp->n_pad;
    int P0, P1, P2;
    axis_sizes(p, d, P0, P1, P2);
    double *bufA = lds, *bufB = lds + n, *acc = lds + 2 * n, *xf = lds + 3 * n;
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * p * p; i += 256) xf[i] = chp->xfer[i];
    const int P = parents[blockIdx.x];
    const int64_t c0 = child_ptr[P], c1 = child_ptr[P + 1];
    for (int k = 0; k < K; ++k) {
        for (int I = tid; I < n; I += 256) acc[I] = 0.0;
        for (int64_t q = c0; q < c1; ++q) {
            const int ch = child_idx[q];
            const int oct = octant[ch];
            const double *Mc = M + ((int64_t)k * C + ch) * n_pad;
            __syncthreads();
            for (int I = tid; I < n; I += 256) bufA[I] = Mc[I];
            __syncthreads();
            double *in = bufA, *out = bufB;
            for (int axis = d - 1; axis >= 0; --axis) {
                const double *x1 = xf + ((oct >> axis) & 1) * p * p; // chebyshev.rs:183-192: bit a <-> axis a
                transfer_pass<true>(in, out, x1, axis, p, P0, P1, P2, n, tid, 256);
                __syncthreads();
                double *t = in;
                in = out;
                out = t;
            }
            for (int I = tid; I < n; I += 256) acc[I] += in[I];
        }
        __syncthreads();
        double *Mp = M + ((int64_t)k * C + P) * n_pad;
        for (int I = tid; I < n; I += 256) Mp[I] = acc[I]; // a parent is written once
        __syncthreads();
    }
}

// local_to_local (bbfmm.rs:1051-1086): one workgroup per child cell.
__global__ __launch_bounds__(256) void l2l_kernel(const DevCheb *__restrict__ chp, int K, int64_t C,
                                                  const int32_t *__restrict__ cells,
                                                  const int32_t *__restrict__ parent,
                                                  const int32_t *__restrict__ octant,
                                                  const uint8_t *__restrict__ active, double *__restrict__ L) {
    extern __shared__ double lds[];
    const int p = chp->p, d = chp->d, n = chp->n, n_pad = chp->n_pad;
    int P0, P1, P2;
    axis_sizes(p, d, P0, P1, P2);
    double *bufA = lds, *bufB = lds + n, *xf = lds + 2 * n;
    const int tid = threadIdx.x;
    const int c = cells[blockIdx.x];
    if (active && !active[c]) return;
    const int P = parent[c];
    if (P < 0) return;
    for (int i = tid; i < 2 * p * p; i += 256) xf[i] = chp->xfer[i];
    const int oct = octant[c];
    for (int k = 0; k < K; ++k) {
        const double *Lp = L + ((int64_t)k * C + P) * n_pad;
        __syncthreads();
        for (int I = tid; I < n; I += 256) bufA[I] = Lp[I];
        __syncthreads();
        double *in = bufA, *out = bufB;
        for (int axis = 0; axis < d; ++axis) {
            const double *x1 = xf + ((oct >> axis) & 1) * p * p;
            transfer_pass<false>(in, out, x1, axis, p, P0, P1, P2, n, tid, 256);
            __syncthreads();
            double *t = in;
            in = out;
            out = t;
        }
        double *Lc = L + ((int64_t)k * C + c) * n_pad;
        for (int I = tid; I < n; I += 256) Lc[I] += in[I];
    }
}

// ------------------------------------------------------------------ M2M / L2L, 3-D fast path
// One wave per cell, order P a template parameter.  A lane owns "pencils" of P entries along one
// axis in registers, so a 1-D transfer is P*P register FMAs against a wave-uniform P x P block
// (scalar loads); between the axes the vector is transposed through a wave-private LDS slice (no
// workgroup barrier).  Layout of a vector: index (i0*P + i1)*P + i2.
template <int P, bool FORWARD>
__device__ inline void pencil_apply(double (&v)[P], const double *__restrict__ xf) {
    double o[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) s += (FORWARD ? xf[k * P + j] : xf[j * P + k]) * v[k];
        o[j] = s;
    }
#pragma unroll
    for (int j = 0; j < P; ++j) v[j] = o[j];
}

// in: global vector of the source cell; buf: wave-private LDS (P^3 doubles); on return buf holds
// the transferred vector.  oct: octant of the child (bit a <-> axis a, chebyshev.rs:183-192).
template <int P, bool FORWARD>
__device__ inline void transfer3_wave(const double *__restrict__ in, double *buf, const double *__restrict__ xfer,
                                      int oct, int lane) {
    constexpr int PP = P * P, NPEN = (PP + 63) / 64;
    const double *x0 = xfer + ((oct >> 0) & 1) * PP, *x1 = xfer + ((oct >> 1) & 1) * PP,
                 *x2 = xfer + ((oct >> 2) & 1) * PP;
    // axis 0: pencil q = (i1, i2), entries at i0 * PP + q (coalesced global reads)
#pragma unroll
    for (int ps = 0; ps < NPEN; ++ps) {
        const int q = lane + 64 * ps;
        if (q < PP) {
            double v[P];
#pragma unroll
            for (int i = 0; i < P; ++i) v[i] = in[i * PP + q];
            pencil_apply<P, FORWARD>(v, x0);
#pragma unroll
            for (int i = 0; i < P; ++i) buf[i * PP + q] = v[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // axis 1: pencil r = (i0, i2), entries at i0 * PP + i1 * P + i2
#pragma unroll
    for (int ps = 0; ps < NPEN; ++ps) {
        const int r = lane + 64 * ps;
        if (r < PP) {
            const int i0 = r / P, i2 = r - i0 * P;
            double v[P];
#pragma unroll
            for (int i = 0; i < P; ++i) v[i] = buf[i0 * PP + i * P + i2];
            pencil_apply<P, FORWARD>(v, x1);
#pragma unroll
            for (int i = 0; i < P; ++i) buf[i0 * PP + i * P + i2] = v[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // axis 2: pencil s = (i0, i1), entries at s * P + i2
#pragma unroll
    for (int ps = 0; ps < NPEN; ++ps) {
        const int sidx = lane + 64 * ps;
        if (sidx < PP) {
            double v[P];
#pragma unroll
            for (int i = 0; i < P; ++i) v[i] = buf[sidx * P + i];
            pencil_apply<P, FORWARD>(v, x2);
#pragma unroll
            for (int i = 0; i < P; ++i) buf[sidx * P + i] = v[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

constexpr int XFER_WAVES = 4;

// local_to_local, one wave per child cell: L_child += T_child^T L_parent (bbfmm.rs:1051-1086)
template <int P>
__global__ __launch_bounds__(64 * XFER_WAVES) void l2l3_kernel(const DevCheb *__restrict__ chp, int K, int64_t C,
                                                               const int32_t *__restrict__ cells, int n_cells,
                                                               const int32_t *__restrict__ parent,
                                                               const int32_t *__restrict__ octant,
                                                               const uint8_t *__restrict__ active,
                                                               double *__restrict__ L) {
    constexpr int N = P * P * P;
    __shared__ double s_buf[XFER_WAVES][N];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int job = blockIdx.x * XFER_WAVES + wave;
    if (job >= n_cells) return;
    const int c = __builtin_amdgcn_readfirstlane(cells[job]);
    if (active && !active[c]) return;
    const int Pc = __builtin_amdgcn_readfirstlane(parent[c]);
    if (Pc < 0) return;
    const int oct = __builtin_amdgcn_readfirstlane(octant[c]);
    const int n_pad = chp->n_pad;
    for (int k = 0; k < K; ++k) {
        transfer3_wave<P, false>(L + ((int64_t)k * C + Pc) * n_pad, s_buf[wave], chp->xfer, oct, lane);
        double *Lc = L + ((int64_t)k * C + c) * n_pad;
        for (int I = lane; I < N; I += 64) Lc[I] += s_buf[wave][I];
        __builtin_amdgcn_wave_barrier();
    }
}

// multipole_to_multipole, one workgroup per parent, one wave per child:
// M_parent += sum_children T_child M_child (bbfmm.rs:742-772)
template <int P>
__global__ __launch_bounds__(512) void m2m3_kernel(const DevCheb *__restrict__ chp, int K, int64_t C,
                                                   const int32_t *__restrict__ parents,
                                                   const int64_t *__restrict__ child_ptr,
                                                   const int32_t *__restrict__ child_idx,
                                                   const int32_t *__restrict__ octant, double *__restrict__ M) {
    constexpr int N = P * P * P;
    __shared__ double s_buf[8][N];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Pc = parents[blockIdx.x];
    const int64_t c0 = child_ptr[Pc];
    const int n_ch = (int)(child_ptr[Pc + 1] - c0); // <= 8 in 3-D
    const int n_pad = chp->n_pad;
    for (int k = 0; k < K; ++k) {
        if (wave < n_ch) {
            const int ch = __builtin_amdgcn_readfirstlane(child_idx[c0 + wave]);
            const int oct = __builtin_amdgcn_readfirstlane(octant[ch]);
            transfer3_wave<P, true>(M + ((int64_t)k * C + ch) * n_pad, s_buf[wave], chp->xfer, oct, lane);
        }
        __syncthreads();
        double *Mp = M + ((int64_t)k * C + Pc) * n_pad;
        for (int I = threadIdx.x; I < N; I += 512) {
            double s = 0.0;
            for (int w = 0; w < n_ch; ++w) s += s_buf[w][I];
            Mp[I] = s; // a parent is written once
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ L2P
// local_to_particle (bbfmm.rs:1358-1440): y[t] += S(x_t) . L_leaf, optionally gradients
// (dS scaled by 2/length, chebyshev.rs:862-869).  One wave per leaf, one lane per target; the
// 1-D factors live in registers (order P is a template parameter), L_leaf is broadcast from LDS.
constexpr int L2P_WAVES = 4;
template <int P, int D> constexpr int l2p_waves() { return D == 3 && P > 12 ? (P > 14 ? 1 : 2) : L2P_WAVES; } // P^3 doubles of LDS per wave

template <int P, int D, bool GRAD>
__global__ __launch_bounds__((64 * l2p_waves<P, D>())) void l2p_kernel(const DevCheb *__restrict__ chp, int n_jobs,
                                                             const int32_t *__restrict__ leaf_cells,
                                                             const int32_t *__restrict__ tgt_begin,
                                                             const int32_t *__restrict__ tgt_end,
                                                             const double *__restrict__ centers,
                                                             const double *__restrict__ lengths, Xyz tgt,
                                                             int64_t n_tgt, int K, int64_t C,
                                                             const double *__restrict__ L,
                                                             double *__restrict__ out, double *__restrict__ grad) {
    constexpr int P1 = D > 1 ? P : 1, P2 = D > 2 ? P : 1, N = P * P1 * P2;
    constexpr int WAVES = l2p_waves<P, D>();
    __shared__ double s_L[WAVES][N];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // Values only: the node values T_k(node_j) come through the scalar cache as SGPR operands -- staged in LDS the
    // compiler kept all P^2 of them in vector registers across the three axes (166 -> 102 VGPRs at order 7, L2P 0.98 ->
    // 0.81 ms at 10M points).  With gradients the kernel is out of registers either way and keeps the LDS copy.
    // (A matrix-pipe version like p2m_mfma_kernel -- sixteen points per v_mfma_f64_4x4x4, the coefficients as B operands
    // in registers -- was built and measured: 0.92 ms at order 7, 1.84 ms against 1.42 at order 9, ahead only with eight
    // right-hand sides (4.3 against 5.1 ms).  Not kept.)
    __shared__ double s_polyn_lds[GRAD ? P * P : 1];
    if constexpr (GRAD) {
        for (int i = tid; i < P * P; i += 64 * WAVES) s_polyn_lds[i] = chp->polyn[i];
        __syncthreads();
    }
    const double *__restrict__ s_polyn = GRAD ? s_polyn_lds : chp->polyn;
    const int job = blockIdx.x * WAVES + wave;
    if (job >= n_jobs) return; // whole wave; no block barrier below
    const int n_pad = chp->n_pad;
    const int cell = leaf_cells[job];
    const int b = tgt_begin[job], e = tgt_end[job];
    const double len = lengths[cell];
    const double cc[3] = {centers[cell * 3 + 0], centers[cell * 3 + 1], centers[cell * 3 + 2]};
    double *Lw = s_L[wave];
    for (int base = b; base < e; base += 64) {
        const int t = base + lane;
        const bool valid = t < e;
        double S0[P], S1[P1], S2[P2], D0[P], D1[P1], D2[P2];
        {
            const double x0 = valid ? (tgt.x[t] - cc[0]) / (len * 0.5) : 0.0; // chebyshev.rs:841-845
            cheb_S_reg<P, GRAD>(x0, s_polyn, S0, D0);
            if (D > 1) {
                const double x1 = valid ? (tgt.y[t] - cc[1]) / (len * 0.5) : 0.0;
                double s[P], d[P];
                cheb_S_reg<P, GRAD>(x1, s_polyn, s, d);
#pragma unroll
                for (int j = 0; j < P1; ++j) { S1[j] = s[j]; D1[j] = d[j]; }
            } else {
                S1[0] = 1.0;
                D1[0] = 0.0;
            }
            if (D > 2) {
                const double x2 = valid ? (tgt.z[t] - cc[2]) / (len * 0.5) : 0.0;
                double s[P], d[P];
                cheb_S_reg<P, GRAD>(x2, s_polyn, s, d);
#pragma unroll
                for (int j = 0; j < P2; ++j) { S2[j] = s[j]; D2[j] = d[j]; }
            } else {
                S2[0] = 1.0;
                D2[0] = 0.0;
            }
        }
        const double gs = 2.0 / len;
        for (int k = 0; k < K; ++k) {
            const double *Lc = L + ((int64_t)k * C + cell) * n_pad;
            // wave-private LDS slice: in-order LDS ops of one wave need no barrier, only the wait
            for (int I = lane; I < N; I += 64) Lw[I] = Lc[I];
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            double y = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
#pragma unroll
            for (int a = 0; a < P; ++a) {
                double ua = 0.0, uay = 0.0, uaz = 0.0;
#pragma unroll
                for (int bb = 0; bb < P1; ++bb) {
                    double t0 = 0.0, t0z = 0.0;
#pragma unroll
                    for (int c = 0; c < P2; ++c) {
                        const double lv = Lw[(a * P1 + bb) * P2 + c];
                        t0 += S2[c] * lv;
                        if (GRAD) t0z += D2[c] * lv;
                    }
                    ua += S1[bb] * t0;
                    if (GRAD) {
                        uay += D1[bb] * t0;
                        uaz += S1[bb] * t0z;
                    }
                }
                y += S0[a] * ua;
                if (GRAD) {
                    gx += D0[a] * ua;
                    gy += S0[a] * uay;
                    gz += S0[a] * uaz;
                }
            }
            if (valid) {
                out[(int64_t)k * n_tgt + t] += y;
                if (GRAD) {
                    grad[((int64_t)k * D + 0) * n_tgt + t] += gx * gs;
                    if (D > 1) grad[((int64_t)k * D + 1) * n_tgt + t] += gy * gs;
                    if (D > 2) grad[((int64_t)k * D + 2) * n_tgt + t] += gz * gs;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ direct interactions
// Shared inner loop of P2P / M2P / P2L: every thread owns two targets and one "slice" of the
// staged source tile (stride S); sources are read from LDS as {x, y}, z, w and serve both targets
// (half the LDS traffic per kernel evaluation of a one-target loop, which ran at ~70 % of the LDS
// bandwidth).  acc[t][kk] += K(t, s) * w_kk(s); gradient accumulators optional.
constexpr int DIRECT_TILE = 512;
constexpr int DIRECT_KB = 4;
constexpr int DIRECT_KB_WIDE = 8; // P2P values without gradients: eight rhs per kernel evaluation (config 4)

template <int KB> struct SrcTile { // LDS per workgroup: 12 KB + 4 KB per right-hand side of the pass
    double2 xy[DIRECT_TILE];
    double zs[DIRECT_TILE];
    double w[KB][DIRECT_TILE];
};

template <int KID, bool GRAD, int KB>
__device__ inline void direct_tile(const KernelSpec &ks, const SrcTile<KB> &tile, int count, int first, int stride,
                                   const double (&t)[2][3], double (&acc)[2][KB], double (&gacc)[2][KB][3]) {
    for (int j = first; j < count; j += stride) {
        const double2 xy = tile.xy[j];
        const double z = tile.zs[j];
        double wk[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) wk[kk] = tile.w[kk][j];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double dx = t[h][0] - xy.x, dy = t[h][1] - xy.y, dz = t[h][2] - z;
            const double r2 = dx * dx + dy * dy + dz * dz; // distance_sq, utils.rs:230-237
            if (GRAD) {
                double f;
                const double v = kernel_value_grad_r2<KID>(ks, r2, &f);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    acc[h][kk] += v * wk[kk];
                    gacc[h][kk][0] += (f * dx) * wk[kk];
                    gacc[h][kk][1] += (f * dy) * wk[kk];
                    gacc[h][kk][2] += (f * dz) * wk[kk];
                }
            } else {
                const double v = kernel_value_r2<KID>(ks, r2);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) acc[h][kk] += v * wk[kk];
            }
        }
    }
}

// Split nt targets into m passes of n_c targets; a pass runs ceil(n_c / 2) target pairs in
// S = 256 / pairs source slices, so the source tiles are walked m / S times in total.
struct TargetPlan {
    int n_c, pairs, S;
};
__device__ inline TargetPlan plan_targets(int nt) {
    const int m0 = (nt + 511) / 512;
    TargetPlan best{nt, 256, 1};
    float best_cost = 1e30f;
    for (int m = m0; m < m0 + 4; ++m) {
        const int nc = (nt + m - 1) / m, pairs = (nc + 1) / 2, S = 256 / pairs;
        const float cost = (float)m / (float)S + 0.02f * (float)m; // + restaging the tiles per pass
        if (cost < best_cost) {
            best_cost = cost;
            best = TargetPlan{nc, pairs, S};
        }
    }
    return best;
}

// Cross-slice reduction through LDS; returns the total in the slice-0 thread.
__device__ inline double slice_reduce(double v, double *red, int ti, int sl, int S, int nt, bool participates) {
    __syncthreads();
    if (participates) red[sl * nt + ti] = v;
    __syncthreads();
    double s = 0.0;
    if (participates && sl == 0)
        for (int q = 0; q < S; ++q) s += red[q * nt + ti];
    return s;
}

// particle_to_particle (bbfmm.rs:1162-1251).  One workgroup per target leaf.
template <int KID, bool GRAD, int KB>
__global__ __launch_bounds__(256) void p2p_kernel(KernelSpec ks, int d, DirectJobs jobs, Xyz tgt, int64_t n_tgt,
                                                  Xyz src, const double *__restrict__ ws, int64_t N, int k0, int kb,
                                                  double *__restrict__ out, double *__restrict__ grad) {
    __shared__ SrcTile<KB> tile;
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int job = blockIdx.x;
    const int t0 = jobs.tgt_begin[job], t1 = jobs.tgt_end[job];
    const int jcell = jobs.job_cell[job];
    const int64_t r0 = jobs.run_ptr[jcell], r1 = jobs.run_ptr[jcell + 1];
    const TargetPlan tp = plan_targets(t1 - t0);
    for (int tc = t0; tc < t1; tc += tp.n_c) {
        const int nt = min(tp.n_c, t1 - tc);
        const int ti = tid % tp.pairs, sl = tid / tp.pairs;
        const bool part = sl < tp.S && ti < nt;
        const bool two = ti + tp.pairs < nt;
        double t[2][3] = {{0, 0, 0}, {0, 0, 0}};
        if (part) {
            const int ia = tc + ti, ib = two ? ia + tp.pairs : ia;
            t[0][0] = tgt.x[ia], t[0][1] = tgt.y[ia], t[0][2] = tgt.z[ia];
            t[1][0] = tgt.x[ib], t[1][1] = tgt.y[ib], t[1][2] = tgt.z[ib];
        }
        double acc[2][KB], gacc[2][KB][3];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) acc[h][kk] = gacc[h][kk][0] = gacc[h][kk][1] = gacc[h][kk][2] = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const int sb = jobs.runs[2 * r], se = jobs.runs[2 * r + 1];
            for (int base = sb; base < se; base += DIRECT_TILE) {
                const int cnt = min(DIRECT_TILE, se - base);
                __syncthreads();
                for (int j = tid; j < cnt; j += 256) {
                    tile.xy[j] = make_double2(src.x[base + j], src.y[base + j]);
                    tile.zs[j] = src.z[base + j];
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk)
                        tile.w[kk][j] = kk < kb ? ws[(int64_t)(k0 + kk) * N + base + j] : 0.0;
                }
                __syncthreads();
                if (part) direct_tile<KID, GRAD, KB>(ks, tile, cnt, sl, tp.S, t, acc, gacc);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool wr = part && sl == 0 && (h == 0 || two);
            const int64_t it = tc + ti + h * tp.pairs;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                if (kk >= kb) break;
                const double v = slice_reduce(acc[h][kk], red, ti, sl, tp.S, tp.pairs, part);
                if (wr) out[(int64_t)(k0 + kk) * n_tgt + it] += v;
                if (GRAD) {
                    for (int a = 0; a < d; ++a) {
                        const double g = slice_reduce(gacc[h][kk][a], red, ti, sl, tp.S, tp.pairs, part);
                        if (wr) grad[((int64_t)(k0 + kk) * d + a) * n_tgt + it] += g;
                    }
                }
            }
        }
    }
}

// particle_to_particle with targets = sources (the matvec): every unordered pair of points once.
// The reference evaluates phi(x_i, x_j) for both (i, j) and (j, i) (bbfmm.rs:1162-1251 runs once per target
// leaf over its whole U list); U lists are symmetric and so is every kernel of the closed set, so a leaf A
// here takes only the part of its U list that lies AFTER it in the sorted order ("two-sided" runs): each
// value phi(a_i, b_j) is added to the row sum of a_i (times w_b[j]) and to the column sum of b_j (times
// w_a[i]).  "One-sided" runs (the leaf itself; for a partition the U points other ranks own) only feed the
// rows.  A job = up to SYM_WAVES * SYM_TR consecutive targets of one leaf, one workgroup (8 waves) each: a wave
// owns up to SYM_TR targets whose coordinates are wave-uniform (scalar loads), its lanes walk the staged source
// tile -- the leaf's runs packed back to back, lane-linear LDS reads; row sums stay in registers over all
// tiles and are reduced across the wave once, column sums go to an LDS accumulator by ds_add_f64 (distinct
// addresses inside a wave) and from there to HBM with one f64 atomic per source and tile.  One right-hand
// side per launch (more rhs take the ordered-pair kernel above).
// KB right-hand sides in one pass (round 4): one kernel evaluation feeds KB row and KB column sums; the rows of a wave
// go by in passes of sym_rows_pass<KB>() (their weights are SGPR operands), the tile shrinks with KB so that its
// weights and column accumulators stay under 64 KB of LDS.  rhs k reads ws + k * ldw and adds to out + k * ldo.
constexpr int SYM_TILE = 768;
constexpr int SYM_TR = 6;
constexpr int SYM_WAVES = 8;
constexpr int SYM_SEG = 64; // runs packed into one tile at most
template <int KB> constexpr int sym_tile() { return KB <= 2 ? SYM_TILE : SYM_TILE / 2; }
template <int KB> constexpr int sym_rows_pass() { return KB == 1 ? SYM_TR : 3; }

struct SymJobs {
    int n_jobs;
    const int32_t *tgt_begin, *tgt_end; // targets of job i: positions in the target set (sorted order)
    const int64_t *run_range;           // 2 per job: first and one-past-last run of the job's leaf
    const int32_t *runs;                // 3 ints per run: begin, end (sorted source indices), 1 = two-sided
    int32_t tgt_off;                    // sorted source index of target position 0
};

template <int KB> struct SymTile {
    static constexpr int T = sym_tile<KB>();
    double x[T], y[T], z[T], w[KB][T], col[KB][T];
    int32_t cidx[T]; // target position of a two-sided column, -1 for a one-sided one
    int32_t seg_src[SYM_SEG], seg_off[SYM_SEG], seg_two[SYM_SEG]; // runs packed into the tile
    int32_t fill, nseg, next_pos;
    int64_t next_q;
};

// The value is loaded whatever the flag says (a load inside `flag ? load : 0` sits behind a branch of its own, and a
// chunk's dozen scalar loads then wait for one another instead of going out together).
__device__ inline double keep_if(bool keep, double loaded) { return keep ? loaded : 0.0; }

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int KID, int KB>
__global__ __launch_bounds__(64 * SYM_WAVES) void p2p_sym_kernel(KernelSpec ks, SymJobs jobs, Xyz src,
                                                                const double *__restrict__ ws, int64_t ldw, int kb,
                                                                double *__restrict__ out, int64_t ldo) {
    constexpr int T = sym_tile<KB>();
    constexpr int TRP = sym_rows_pass<KB>();
    __shared__ SymTile<KB> tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x;
    const int t0 = jobs.tgt_begin[job], t1 = jobs.tgt_end[job];
    const int rpw = (t1 - t0 + SYM_WAVES - 1) / SYM_WAVES; // rows per wave, <= SYM_TR
    const int r_lo = min(t0 + wave * rpw, t1);
    const int nr = min(rpw, t1 - r_lo);
    double racc[SYM_TR][KB];
#pragma unroll
    for (int r = 0; r < SYM_TR; ++r)
#pragma unroll
        for (int k = 0; k < KB; ++k) racc[r][k] = 0.0;
    int64_t q = jobs.run_range[2 * job];
    const int64_t q1 = jobs.run_range[2 * job + 1];
    int pos = 0; // points of run q already staged
    while (q < q1) {
        __syncthreads(); // the previous tile has been read and its columns flushed
        // The tile's segment table: up to SYM_SEG runs packed back to back until T columns are full.  One
        // wave reads the run triples in one go and scans their lengths; then every thread finds the run of its
        // column by bisection in LDS, so that all source loads of the tile leave in one batch (one memory round
        // trip for the table, one for the columns, whatever the number of runs).
        if (wave == 0) {
            const int64_t r = q + lane;
            const bool valid = r < q1;
            int b = valid ? jobs.runs[3 * r] : 0;
            const int e = valid ? jobs.runs[3 * r + 1] : 0;
            const int two = valid ? jobs.runs[3 * r + 2] : 0;
            if (lane == 0) b += pos;
            const int len = e - b;
            int incl = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            const int excl = incl - len;
            const int take = min(len, max(T - excl, 0));
            tile.seg_src[lane] = b;
            tile.seg_off[lane] = excl;
            tile.seg_two[lane] = two;
            const unsigned long long used = __ballot(take > 0);
            const int nseg = __popcll(used);
            if (lane == nseg - 1) {
                const bool full = take == len;
                tile.fill = excl + take;
                tile.nseg = nseg;
                tile.next_q = q + lane + (full ? 1 : 0);
                tile.next_pos = full ? 0 : ((lane == 0 ? pos : 0) + take);
            }
        }
        __syncthreads();
        const int fill = tile.fill, nseg = tile.nseg;
        q = tile.next_q;
        pos = tile.next_pos;
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) {
            int lo = 0, hi = nseg;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (tile.seg_off[mid] <= j) lo = mid;
                else hi = mid;
            }
            const int g = tile.seg_src[lo] + (j - tile.seg_off[lo]);
            tile.x[j] = src.x[g];
            tile.y[j] = src.y[g];
            tile.z[j] = src.z[g];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                tile.w[k][j] = ws[min(k, kb - 1) * ldw + g]; // (idle slots repeat the last rhs; their sums are never stored)
                tile.col[k][j] = 0.0;
            }
            tile.cidx[j] = tile.seg_two[lo] ? g - jobs.tgt_off : -1;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < SYM_TR; p += TRP) {
            if (p < nr) { // wave-uniform
                double tx[TRP], ty[TRP], tz[TRP], tw[TRP][KB];
#pragma unroll
                for (int r = 0; r < TRP; ++r) { // sorted source index of the target (wave-uniform: scalar loads)
                    const int g = jobs.tgt_off + min(r_lo + min(p + r, nr - 1), t1 - 1);
                    tx[r] = src.x[g], ty[r] = src.y[g], tz[r] = src.z[g];
#pragma unroll
                    for (int k = 0; k < KB; ++k) tw[r][k] = keep_if(p + r < nr, ws[min(k, kb - 1) * ldw + g]);
                }
                for (int j = lane; j < fill; j += 64) {
                    const double xs = tile.x[j], ys = tile.y[j], zs = tile.z[j];
                    double wj[KB], csum[KB];
#pragma unroll
                    for (int k = 0; k < KB; ++k) wj[k] = tile.w[k][j], csum[k] = 0.0;
#pragma unroll
                    for (int r = 0; r < TRP; ++r) {
                        if (p + r < SYM_TR) { // (rows past nr are clamped copies with weight 0; their row sums are dropped)
                            const double dx = tx[r] - xs, dy = ty[r] - ys, dz = tz[r] - zs;
                            const double v = kernel_value_r2<KID>(ks, dx * dx + dy * dy + dz * dz);
#pragma unroll
                            for (int k = 0; k < KB; ++k) {
                                racc[p + r < SYM_TR ? p + r : 0][k] += v * wj[k];
                                csum[k] += v * tw[r][k];
                            }
                        }
                    }
                    if (tile.cidx[j] >= 0) {
#pragma unroll
                        for (int k = 0; k < KB; ++k)
                            if (k < kb) unsafeAtomicAdd(&tile.col[k][j], csum[k]);
                    }
                }
            }
        }
        __syncthreads();
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) {
            const int c = tile.cidx[j];
            if (c >= 0) {
#pragma unroll
                for (int k = 0; k < KB; ++k)
                    if (k < kb) unsafeAtomicAdd(&out[k * ldo + c], tile.col[k][j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < SYM_TR; ++r) {
        if (r < nr) {
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (k < kb) {
                    const double s = wave_sum(racc[r][k]);
                    if (lane == 0) unsafeAtomicAdd(&out[k * ldo + r_lo + r], s);
                }
            }
        }
    }
}

// ---- the workgroup kernel for WHOLE big leaves, one right-hand side (round 6).  The kernel above takes a leaf of 153
// rows (40M points) as four jobs of 38-39: every job stages the same tiles and flushes the same column sums again, and its
// eight waves get five rows each, computed as six (rows per wave are padded to the pass) -- 0.80 of the pair arithmetic
// it executes is real, and the potentials are written 54 times over (profiles/r05_final_config5_size_*_counters.txt).
// Here a job is a leaf (bigger ones than p2p_sym3_rows_per_job() in equal parts): its rows are dealt to the waves
// evenly (counts differ by one at most), a wave runs them against a tile as full passes of SYM_TR rows plus ONE pass of
// exactly the remaining rows (six instances of the pass, selected by a wave-uniform switch): no padded row is ever
// evaluated.  A pass reduces its row sums across the wave and adds them to the potentials at once -- accumulators that
// lived over all tiles cost the kernel its occupancy (111 VGPRs for three passes' worth, 164 for five: 11.1 and 19.9 ms
// at 5M Spheroidal3 points against 12.4 for the chunk kernel); column sums collect in the tile's LDS accumulator over
// all of the leaf's rows and leave with one atomic per source and (leaf, tile).
// One pass: NR rows (sorted sources g0 .. g0 + NR, wave-uniform) against the tile.  ALLCOLS: every column takes its column
// sum (the nodes of W cells); otherwise only the two-sided ones (cidx >= 0) are flushed by the caller.  The rows' sums over this tile are reduced
// across the wave and added to out[row0 + r] at once (rows outside [win_lo, win_hi) are dropped: a partition's window) --
// the accumulators do not outlive the pass, so the kernel holds NR of them whatever the size of the leaf.
template <int KID, int NR, bool ALLCOLS>
__device__ inline void sym3_pass(const KernelSpec &ks, SymTile<1> &tile, int fill, int lane, int g0, int row0, const Xyz &src,
                                 const double *__restrict__ ws, double *__restrict__ out, int win_lo, int win_hi) {
    double tx[NR], ty[NR], tz[NR], tw[NR], racc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { // wave-uniform: scalar loads, SGPR operands
        tx[r] = src.x[g0 + r], ty[r] = src.y[g0 + r], tz[r] = src.z[g0 + r];
        tw[r] = ws[g0 + r];
        racc[r] = 0.0;
    }
    for (int j = lane; j < fill; j += 64) {
        const double xs = tile.x[j], ys = tile.y[j], zs = tile.z[j], wj = tile.w[0][j];
        double csum = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double dx = tx[r] - xs, dy = ty[r] - ys, dz = tz[r] - zs;
            const double v = kernel_value_r2<KID>(ks, dx * dx + dy * dy + dz * dz);
            racc[r] += v * wj;
            csum += v * tw[r];
        }
        // (one-sided columns -- cidx < 0: the leaf itself, another part's points -- collect their sums too and are skipped
        // when the tile is flushed: cheaper than reading cidx and masking the add in every iteration)
        unsafeAtomicAdd(&tile.col[0][j], csum);
    }
    if constexpr (NR >= 4) {
        // eight (padded) row sums across the wave by halving: lanes l and l ^ 32 split the rows between them and exchange
        // the halves they give up (four exchanges), then l ^ 16 (two), then l ^ 8 (one) -- every lane is left with ONE row,
        // (l >> 3) & 7, summed over eight lanes -- and three plain steps finish it: 10 exchanges where NR separate
        // reductions took 6 NR (a pass of eight rows: 48), 8 % of a LinearRbf pass.
        const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
        double u[4], t2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double a = racc[i], b = i + 4 < NR ? racc[i + 4 < NR ? i + 4 : 0] : 0.0;
            u[i] = (b5 ? b : a) + __shfl_xor(b5 ? a : b, 32, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) t2[i] = (b4 ? u[i + 2] : u[i]) + __shfl_xor(b4 ? u[i] : u[i + 2], 16, 64);
        double tot = (b3 ? t2[1] : t2[0]) + __shfl_xor(b3 ? t2[0] : t2[1], 8, 64);
        tot += __shfl_xor(tot, 4, 64);
        tot += __shfl_xor(tot, 2, 64);
        tot += __shfl_xor(tot, 1, 64);
        const int r = (lane >> 3) & 7, o = row0 + r;
        if ((lane & 7) == 0 && r < NR && o >= win_lo && o < win_hi) unsafeAtomicAdd(&out[o], tot);
    } else {
        double mine = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double sres = wave_sum(racc[r]);
            if (lane == r) mine = sres;
        }
        const int o = row0 + lane;
        if (lane < NR && o >= win_lo && o < win_hi) unsafeAtomicAdd(&out[o], mine);
    }
}

// A wave's nr rows in passes of at most MAXR rows, all of sz or sz + 1 rows (equal passes: a pass of one or two rows costs
// a third of a full one -- the tile's LDS reads and the column bookkeeping do not shrink with it: 153-row leaves as single
// jobs with passes 6 + 6 + 6 + 1 ran 15.2 ms against 10.9 for two half leaves)
template <int KID, int MAXR, bool ALLCOLS>
__device__ inline void sym3_rows(const KernelSpec &ks, SymTile<1> &tile, int fill, int lane, int g_lo, int row_lo, int nr,
                                 const Xyz &src, const double *__restrict__ ws, double *__restrict__ out, int win_lo, int win_hi) {
    const int npass = (nr + MAXR - 1) / MAXR;
    const int sz = npass > 0 ? nr / npass : 0, big = nr - sz * npass;
    int row = 0;
    for (int p = 0; p < npass; ++p) {
        const int cnt = sz + (p < big ? 1 : 0);
        const int g0 = g_lo + row, row0 = row_lo + row;
        switch (cnt) { // wave-uniform
        case 1: sym3_pass<KID, 1, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        case 2: sym3_pass<KID, 2, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        case 3: sym3_pass<KID, 3, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        case 4: sym3_pass<KID, 4, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        case 5: sym3_pass<KID, 5, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        case 6: sym3_pass<KID, 6, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        case 7: if constexpr (MAXR >= 7) sym3_pass<KID, 7, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        case 8: if constexpr (MAXR >= 8) sym3_pass<KID, 8, ALLCOLS>(ks, tile, fill, lane, g0, row0, src, ws, out, win_lo, win_hi); break;
        default: break;
        }
        row += cnt;
    }
}

template <int KID, int MAXR>
__global__ __launch_bounds__(64 * SYM_WAVES) void p2p_sym3_kernel(KernelSpec ks, SymJobs jobs, Xyz src,
                                                                 const double *__restrict__ ws, double *__restrict__ out) {
    constexpr int T = sym_tile<1>();
    __shared__ SymTile<1> tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x;
    const int t0 = jobs.tgt_begin[job], t1 = jobs.tgt_end[job];
    const int rows = t1 - t0, base = rows / SYM_WAVES, extra = rows - base * SYM_WAVES;
    const int nr = base + (wave < extra ? 1 : 0);
    const int r_lo = t0 + wave * base + min(wave, extra);
    const int g_lo = jobs.tgt_off + r_lo;                       // sorted source index of the wave's first row
    int64_t q = jobs.run_range[2 * job];
    const int64_t q1 = jobs.run_range[2 * job + 1];
    int pos = 0;
    while (q < q1) {
        __syncthreads(); // the previous tile has been read and its columns flushed
        if (wave == 0) { // the tile's segment table, as in p2p_sym_kernel
            const int64_t r = q + lane;
            const bool valid = r < q1;
            int bq = valid ? jobs.runs[3 * r] : 0;
            const int e = valid ? jobs.runs[3 * r + 1] : 0;
            const int two = valid ? jobs.runs[3 * r + 2] : 0;
            if (lane == 0) bq += pos;
            const int len = e - bq;
            int incl = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            const int excl = incl - len;
            const int take = min(len, max(T - excl, 0));
            tile.seg_src[lane] = bq;
            tile.seg_off[lane] = excl;
            tile.seg_two[lane] = two;
            const unsigned long long used = __ballot(take > 0);
            const int nseg = __popcll(used);
            if (lane == nseg - 1) {
                const bool full = take == len;
                tile.fill = excl + take;
                tile.nseg = nseg;
                tile.next_q = q + lane + (full ? 1 : 0);
                tile.next_pos = full ? 0 : ((lane == 0 ? pos : 0) + take);
            }
        }
        __syncthreads();
        const int fill = tile.fill, nseg = tile.nseg;
        q = tile.next_q;
        pos = tile.next_pos;
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) {
            int lo = 0, hi = nseg;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (tile.seg_off[mid] <= j) lo = mid;
                else hi = mid;
            }
            const int g = tile.seg_src[lo] + (j - tile.seg_off[lo]);
            tile.x[j] = src.x[g];
            tile.y[j] = src.y[g];
            tile.z[j] = src.z[g];
            tile.w[0][j] = ws[g];
            tile.col[0][j] = 0.0;
            tile.cidx[j] = tile.seg_two[lo] ? g - jobs.tgt_off : -1;
        }
        __syncthreads();
        sym3_rows<KID, MAXR, false>(ks, tile, fill, lane, g_lo, r_lo, nr, src, ws, out, 0, 0x7fffffff);
        __syncthreads();
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) {
            const int c = tile.cidx[j];
            if (c >= 0) unsafeAtomicAdd(&out[c], tile.col[0][j]);
        }
    }
}

// ---- the same unordered-pair sums, one WAVE per job (round 3).  The workgroup version above spends a third of its
// busy cycles around the pair arithmetic (three workgroup barriers per tile, one wave building the segment table while
// seven wait, column sums through LDS atomics, six-row reductions per wave) and leaves the vector ALU idle a quarter of
// the time.  Here a wave owns a whole leaf (up to SYM2_MAX_ROWS rows):
//   * the columns of a tile (SYM2_CG x 64 sources: coordinates, weight, output index) live in REGISTERS -- lane l
//     owns columns l, l + 64, ... -- loaded once per tile through the wave's own segment table (wave-private LDS
//     slice, in-order LDS: no barrier anywhere in the kernel);
//   * the rows go by in chunks of SYM2_R whose coordinates and weights are wave-uniform (scalar loads, SGPR
//     operands); the SYM2_R pair chains of a column are independent and branch-free (rows past the leaf are clamped
//     copies with weight 0), so the compiler interleaves them;
//   * column sums stay in registers over all rows of the leaf and leave with one atomic per source and tile; row sums
//     are reduced per (tile, chunk) through a 4 KB wave-private transpose (8 ds_write_b64, 4 ds_read_b128, 3 DPP
//     steps) and one atomic per row.
//
// KB <= 4 right-hand sides in one pass (round 4; config 4's near field): one kernel evaluation feeds the KB row sums and
// the KB column sums (15 + 2 KB FP64 instructions per unordered pair for LinearRbf against 2 x (15 + KB) of the
// ordered-pair kernel) -- the reference evaluates the kernel once per rhs (bbfmm.rs:1162-1251: loop order rhs, target,
// source); the values are the same, the sums differ in order only.  A column's KB weights and KB sums live in registers
// like its coordinates; the rows' weights are wave-uniform (SGPR operands), which bounds rows-per-chunk x KB: 8 rows for
// one rhs, 4 for 2-4.  rhs k reads ws + k * ldw and adds to out + k * ldo; kb <= KB of them are live.
constexpr int SYM2_CG = 4; // column groups of a tile (5 and 6 measured at one rhs: 4.23 against 4.16 ms, one wave less per SIMD)
constexpr int SYM2_WAVES = 4;
constexpr int SYM2_MAX_ROWS = 256;
template <int KB> constexpr int sym2_rows() { return KB == 1 ? 8 : 4; }

// The transpose of the row sums: NV rows of 64 partial sums, every NV-column segment followed by two pad columns so that
// the 16-byte reads of the reduction (lane l reads segment l % LPV of row l / LPV) fall on sixteen different bank groups
// per pass (round 5: unpadded, 65 % of the kernel's LDS cycles at four rhs and 36 % at one were bank conflicts --
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05_nearfield_counters_before.json; time unchanged, the kernel
// does not wait for its LDS: DESIGN.md section 10).
template <int NV> struct Sym2Wave { // wave-private
    static constexpr int kSeg = NV + 2, kRow = (64 / NV) * kSeg;
    double red[NV][kRow];
    int32_t seg_src[64], seg_off[64], seg_two[64];
};

template <int KID, int KB>
__global__ __launch_bounds__(64 * SYM2_WAVES) void p2p_sym2_kernel(KernelSpec ks, SymJobs jobs, Xyz src,
                                                                  const double *__restrict__ ws, int64_t ldw, int kb,
                                                                  double *__restrict__ out, int64_t ldo) {
    constexpr int R = sym2_rows<KB>();
    constexpr int NV = R * KB;        // row sums per (tile, chunk): 8 or 16
    constexpr int LPV = 64 / NV;      // lanes that share the final sum of one of them
    __shared__ Sym2Wave<NV> lds[SYM2_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job = blockIdx.x * SYM2_WAVES + wave;
    if (job >= jobs.n_jobs) return; // whole wave; no workgroup barrier below
    Sym2Wave<NV> &W = lds[wave];
    const int t0 = jobs.tgt_begin[job], t1 = jobs.tgt_end[job];
    int64_t q = jobs.run_range[2 * job];
    const int64_t q1 = jobs.run_range[2 * job + 1];
    int pos = 0; // points of run q already taken
    constexpr int TILE = 64 * SYM2_CG;
    while (q < q1) {
        // segment table of the tile: up to 64 runs packed back to back until TILE columns are full
        int fill, nseg;
        {
            const int64_t r = q + lane;
            const bool valid = r < q1;
            int b = valid ? jobs.runs[3 * r] : 0;
            const int e = valid ? jobs.runs[3 * r + 1] : 0;
            const int two = valid ? jobs.runs[3 * r + 2] : 0;
            if (lane == 0) b += pos;
            const int len = e - b;
            int incl = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            const int excl = incl - len;
            const int take = min(len, max(TILE - excl, 0));
            W.seg_src[lane] = b;
            W.seg_off[lane] = excl;
            W.seg_two[lane] = two;
            nseg = __popcll(__ballot(take > 0));
            const int last = max(nseg - 1, 0);
            fill = __builtin_amdgcn_readlane(excl + take, last);
            const int take_l = __builtin_amdgcn_readlane(take, last), len_l = __builtin_amdgcn_readlane(len, last);
            const bool full = take_l == len_l;
            const int pos_now = pos;
            q = q + last + (full ? 1 : 0);
            pos = full ? 0 : ((last == 0 ? pos_now : 0) + take_l);
            if (nseg == 0) break; // (empty runs only: cannot happen with the host's lists; never spin)
        }
        // the tile's columns into registers
        double cx[SYM2_CG], cy[SYM2_CG], cz[SYM2_CG], cw[SYM2_CG][KB], csum[SYM2_CG][KB];
        int cidx[SYM2_CG];
#pragma unroll
        for (int cg = 0; cg < SYM2_CG; ++cg) {
            const int j = cg * 64 + lane;
            cx[cg] = cy[cg] = cz[cg] = 0.0; // a padding column: weight 0, its sums are dropped
            cidx[cg] = -1;
#pragma unroll
            for (int k = 0; k < KB; ++k) cw[cg][k] = csum[cg][k] = 0.0;
            if (j < fill) {
                int lo = 0, hi = nseg;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (W.seg_off[mid] <= j) lo = mid;
                    else hi = mid;
                }
                const int g = W.seg_src[lo] + (j - W.seg_off[lo]);
                cx[cg] = src.x[g];
                cy[cg] = src.y[g];
                cz[cg] = src.z[g];
#pragma unroll
                for (int k = 0; k < KB; ++k) // (idle slots repeat the last rhs; their sums are never stored)
                    cw[cg][k] = ws[min(k, kb - 1) * ldw + g];
                cidx[cg] = W.seg_two[lo] ? g - jobs.tgt_off : -1;
            }
        }
        const int ncg = (fill + 63) >> 6; // wave-uniform
        for (int rb = t0; rb < t1; rb += R) {
            // the rows of a chunk: coordinates and weights are wave-uniform (scalar loads, SGPR operands)
            double tx[R], ty[R], tz[R], tw[R][KB], racc[R][KB];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int g = jobs.tgt_off + min(rb + r, t1 - 1);
                tx[r] = src.x[g], ty[r] = src.y[g], tz[r] = src.z[g];
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    tw[r][k] = keep_if(rb + r < t1, ws[min(k, kb - 1) * ldw + g]);
                    racc[r][k] = 0.0;
                }
            }
#pragma unroll
            for (int cg = 0; cg < SYM2_CG; ++cg) {
                if (cg < ncg) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double dx = tx[r] - cx[cg], dy = ty[r] - cy[cg], dz = tz[r] - cz[cg];
                        const double v = kernel_value_r2<KID>(ks, dx * dx + dy * dy + dz * dz);
#pragma unroll
                        for (int k = 0; k < KB; ++k) {
                            racc[r][k] += v * cw[cg][k];
                            csum[cg][k] += v * tw[r][k];
                        }
                    }
                }
            }
            // row sums: transpose through the wave's slice; LPV lanes share one of the NV sums, each adds NV of its 64
            // partial sums (NV / 2 16-byte reads)
            const int wcol = lane + 2 * (lane / NV); // (pad columns behind every NV-column segment)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < KB; ++k) W.red[r * KB + k][wcol] = racc[r][k];
            const double2 *pr = reinterpret_cast<const double2 *>(&W.red[lane / LPV][(lane % LPV) * Sym2Wave<NV>::kSeg]);
            double sum = 0.0;
#pragma unroll
            for (int i = 0; i < NV / 2; i += 2) {
                const double2 a0 = pr[i], a1 = pr[i + 1];
                sum += (a0.x + a0.y) + (a1.x + a1.y);
            }
#pragma unroll
            for (int off = 1; off < LPV; off <<= 1) sum += __shfl_xor(sum, off, 64);
            const int vi = lane / LPV, row = rb + vi / KB, k = vi % KB;
            if (lane % LPV == 0 && row < t1 && k < kb) unsafeAtomicAdd(&out[k * ldo + row], sum);
        }
#pragma unroll
        for (int cg = 0; cg < SYM2_CG; ++cg)
            if (cg < ncg && cidx[cg] >= 0) {
#pragma unroll
                for (int k = 0; k < KB; ++k)
                    if (k < kb) unsafeAtomicAdd(&out[k * ldo + cidx[cg]], csum[cg][k]);
            }
    }
}

// Stage the Chebyshev nodes of `cell` (scale_cheb_nodes_to_cell, chebyshev.rs:951-968) and
// its coefficients as a source tile (n <= DIRECT_TILE assumed per chunk).
template <int KB>
__device__ inline void stage_nodes(SrcTile<KB> &tile, const DevCheb *chp, int P1, int P2, int j0, int cnt, double cx,
                                   double cy, double cz, double half, int d, const double *coef, int64_t coef_stride,
                                   int kb, int tid, int nthreads) {
    for (int j = tid; j < cnt; j += nthreads) {
        const int I = j0 + j;
        const int i2 = I % P2, i1 = (I / P2) % P1, i0 = I / (P2 * P1);
        const double x = cx + half * chp->nodes[i0];
        const double y = d > 1 ? cy + half * chp->nodes[i1] : 0.0;
        const double z = d > 2 ? cz + half * chp->nodes[i2] : 0.0;
        tile.xy[j] = make_double2(x, y);
        tile.zs[j] = z;
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) tile.w[kk][j] = (kk < kb && coef) ? coef[kk * coef_stride + I] : 0.0;
    }
}

// multipole_to_particle (bbfmm.rs:1254-1355).  One workgroup per (target leaf, chunk of its W
// list); chunks of one leaf add into the same targets with hardware f64 atomics.
template <int KID, bool GRAD, int KB>
__global__ __launch_bounds__(256) void m2p_kernel(KernelSpec ks, const DevCheb *__restrict__ chp,
                                                  const int32_t *__restrict__ tgt_begin,
                                                  const int32_t *__restrict__ tgt_end,
                                                  const int64_t *__restrict__ w_begin,
                                                  const int64_t *__restrict__ w_end,
                                                  const int32_t *__restrict__ w_cells,
                                                  const double *__restrict__ centers,
                                                  const double *__restrict__ lengths, Xyz tgt, int64_t n_tgt, int k0,
                                                  int kb, int64_t C, const double *__restrict__ M,
                                                  double *__restrict__ out, double *__restrict__ grad) {
    __shared__ SrcTile<KB> tile;
    __shared__ double red[256];
    const int p = chp->p, d = chp->d, n = chp->n, n_pad = chp->n_pad;
    int P0, P1, P2;
    axis_sizes(p, d, P0, P1, P2);
    const int tid = threadIdx.x;
    const int job = blockIdx.x;
    const int t0 = tgt_begin[job], t1 = tgt_end[job];
    const int64_t r0 = w_begin[job], r1 = w_end[job]; // a chunk of the leaf's W list
    const TargetPlan tp = plan_targets(t1 - t0);
    for (int tc = t0; tc < t1; tc += tp.n_c) {
        const int nt = min(tp.n_c, t1 - tc);
        const int ti = tid % tp.pairs, sl = tid / tp.pairs;
        const bool part = sl < tp.S && ti < nt;
        const bool two = ti + tp.pairs < nt;
        double t[2][3] = {{0, 0, 0}, {0, 0, 0}};
        if (part) {
            const int ia = tc + ti, ib = two ? ia + tp.pairs : ia;
            t[0][0] = tgt.x[ia], t[0][1] = tgt.y[ia], t[0][2] = tgt.z[ia];
            t[1][0] = tgt.x[ib], t[1][1] = tgt.y[ib], t[1][2] = tgt.z[ib];
        }
        double acc[2][KB], gacc[2][KB][3];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) acc[h][kk] = gacc[h][kk][0] = gacc[h][kk][1] = gacc[h][kk][2] = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const int wc = w_cells[r];
            const double half = lengths[wc] * 0.5;
            const double cx = centers[wc * 3], cy = centers[wc * 3 + 1], cz = centers[wc * 3 + 2];
            const double *coef = M + ((int64_t)k0 * C + wc) * n_pad;
            for (int j0 = 0; j0 < n; j0 += DIRECT_TILE) {
                const int cnt = min(DIRECT_TILE, n - j0);
                __syncthreads();
                stage_nodes(tile, chp, P1, P2, j0, cnt, cx, cy, cz, half, d, coef, C * n_pad, kb, tid, 256);
                __syncthreads();
                if (part) direct_tile<KID, GRAD, KB>(ks, tile, cnt, sl, tp.S, t, acc, gacc);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool wr = part && sl == 0 && (h == 0 || two);
            const int64_t it = tc + ti + h * tp.pairs;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                if (kk >= kb) break;
                const double v = slice_reduce(acc[h][kk], red, ti, sl, tp.S, tp.pairs, part);
                if (wr) unsafeAtomicAdd(&out[(int64_t)(k0 + kk) * n_tgt + it], v);
                if (GRAD) {
                    for (int a = 0; a < d; ++a) {
                        const double g = slice_reduce(gacc[h][kk][a], red, ti, sl, tp.S, tp.pairs, part);
                        if (wr) unsafeAtomicAdd(&grad[((int64_t)(k0 + kk) * d + a) * n_tgt + it], g);
                    }
                }
            }
        }
    }
}

// multipole_to_particle and particle_to_local in one pass when targets = sources (one rhs): X = W^T
// (linear_tree.rs:388-392), so the values phi(x_t, node) of a leaf's points against the Chebyshev nodes of its
// W cells are exactly the values P2L needs for those cells' X lists.  Same scheme as p2p_sym_kernel with the
// nodes of the W cells as columns: row sums (x M_W[node]) are M2P (bbfmm.rs:1254-1355), column sums (x w_t)
// are P2L (bbfmm.rs:1001-1048), added to L with f64 atomics after M2L stage 2 has assigned it.
struct WxJobs {
    int n_jobs;
    const int32_t *tgt_begin, *tgt_end; // rows of job i (sorted source positions, at most SYM_WAVES * SYM_TR, one leaf)
    const int64_t *w_range;             // 2 per job: the leaf's range in w_cells
    const int32_t *w_cells;
};

// KB right-hand sides per pass like p2p_sym_kernel: rhs k reads ws + k * ldw and M + k * C * n_pad, adds to
// out + k * ldo and L + k * C * n_pad.
template <int KID, int KB>
__global__ __launch_bounds__(64 * SYM_WAVES) void wx_sym_kernel(KernelSpec ks, WxJobs jobs, const DevCheb *__restrict__ chp,
                                                               const double *__restrict__ centers,
                                                               const double *__restrict__ lengths, Xyz src,
                                                               const double *__restrict__ ws, int64_t ldw, int kb,
                                                               const double *__restrict__ M, double *__restrict__ L,
                                                               int64_t ld_ml, double *__restrict__ out, int64_t ldo,
                                                               int out_off, int out_n) {
    constexpr int T = sym_tile<KB>();
    constexpr int TRP = sym_rows_pass<KB>();
    __shared__ SymTile<KB> tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x;
    const int p = chp->p, d = chp->d, n = chp->n, n_pad = chp->n_pad;
    int P0, P1, P2;
    axis_sizes(p, d, P0, P1, P2);
    const int t0 = jobs.tgt_begin[job], t1 = jobs.tgt_end[job];
    const int rpw = (t1 - t0 + SYM_WAVES - 1) / SYM_WAVES;
    const int r_lo = min(t0 + wave * rpw, t1);
    const int nr = min(rpw, t1 - r_lo);
    double racc[SYM_TR][KB];
#pragma unroll
    for (int r = 0; r < SYM_TR; ++r)
#pragma unroll
        for (int k = 0; k < KB; ++k) racc[r][k] = 0.0;
    const int64_t q0 = jobs.w_range[2 * job], q1 = jobs.w_range[2 * job + 1];
    const int64_t total = (q1 - q0) * n;
    for (int64_t base = 0; base < total; base += T) {
        const int fill = static_cast<int>(min<int64_t>(T, total - base));
        __syncthreads();
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) {
            const int64_t P = base + j;
            const int ci = static_cast<int>(P / n), I = static_cast<int>(P - static_cast<int64_t>(ci) * n);
            const int cell = jobs.w_cells[q0 + ci];
            const double half = lengths[cell] * 0.5;
            const int i2 = I % P2, i1 = (I / P2) % P1, i0 = I / (P2 * P1); // scale_cheb_nodes_to_cell, chebyshev.rs:951-968
            tile.x[j] = centers[cell * 3] + half * chp->nodes[i0];
            tile.y[j] = d > 1 ? centers[cell * 3 + 1] + half * chp->nodes[i1] : 0.0;
            tile.z[j] = d > 2 ? centers[cell * 3 + 2] + half * chp->nodes[i2] : 0.0;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                tile.w[k][j] = M[min(k, kb - 1) * ld_ml + static_cast<int64_t>(cell) * n_pad + I];
                tile.col[k][j] = 0.0;
            }
            tile.cidx[j] = cell * n_pad + I;
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < SYM_TR; pp += TRP) {
            if (pp < nr) { // wave-uniform
                double tx[TRP], ty[TRP], tz[TRP], tw[TRP][KB];
#pragma unroll
                for (int r = 0; r < TRP; ++r) {
                    const int g = min(r_lo + min(pp + r, nr - 1), t1 - 1);
                    tx[r] = src.x[g], ty[r] = src.y[g], tz[r] = src.z[g];
#pragma unroll
                    for (int k = 0; k < KB; ++k) tw[r][k] = keep_if(pp + r < nr, ws[min(k, kb - 1) * ldw + g]);
                }
                for (int j = lane; j < fill; j += 64) {
                    const double xs = tile.x[j], ys = tile.y[j], zs = tile.z[j];
                    double wj[KB], csum[KB];
#pragma unroll
                    for (int k = 0; k < KB; ++k) wj[k] = tile.w[k][j], csum[k] = 0.0;
#pragma unroll
                    for (int r = 0; r < TRP; ++r) {
                        if (pp + r < SYM_TR) { // (rows past nr: clamped copies with weight 0, their row sums are dropped)
                            const double dx = tx[r] - xs, dy = ty[r] - ys, dz = tz[r] - zs;
                            const double v = kernel_value_r2<KID>(ks, dx * dx + dy * dy + dz * dz);
#pragma unroll
                            for (int k = 0; k < KB; ++k) {
                                racc[pp + r < SYM_TR ? pp + r : 0][k] += v * wj[k];
                                csum[k] += v * tw[r][k];
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KB; ++k)
                        if (k < kb) unsafeAtomicAdd(&tile.col[k][j], csum[k]);
                }
            }
        }
        __syncthreads();
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) {
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k < kb) unsafeAtomicAdd(&L[k * ld_ml + tile.cidx[j]], tile.col[k][j]);
        }
    }
#pragma unroll
    for (int r = 0; r < SYM_TR; ++r) {
        if (r < nr) {
            // (a partition's output holds its own rows only: out_off = first owned row; the rows of a leaf outside are
            // here for their column sums -- P2L into the partition's cells -- alone)
            const int o = r_lo + r - out_off;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (k < kb) {
                    const double s = wave_sum(racc[r][k]);
                    if (lane == 0 && o >= 0 && o < out_n) unsafeAtomicAdd(&out[k * ldo + o], s);
                }
            }
        }
    }
}

// The fused M2P + P2L pass for ONE right-hand side on whole leaves (round 6): like p2p_sym3_kernel with the Chebyshev nodes
// of a chunk of the leaf's W cells as columns.  A job = (all rows of a leaf -- up to wx_sym3_rows_per_job(), bigger ones in
// equal parts) x (a chunk of its W list): the nodes of a cell are staged and their column sums flushed to L once per leaf
// instead of once per 48 rows, and no padded row is evaluated.
template <int KID, int MAXR>
__global__ __launch_bounds__(64 * SYM_WAVES) void wx_sym3_kernel(KernelSpec ks, WxJobs jobs, const DevCheb *__restrict__ chp,
                                                                const double *__restrict__ centers,
                                                                const double *__restrict__ lengths, Xyz src,
                                                                const double *__restrict__ ws, const double *__restrict__ M,
                                                                double *__restrict__ L, double *__restrict__ out, int out_off,
                                                                int out_n) {
    constexpr int T = sym_tile<1>();
    __shared__ SymTile<1> tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x;
    const int p = chp->p, d = chp->d, n = chp->n, n_pad = chp->n_pad;
    int P0, P1, P2;
    axis_sizes(p, d, P0, P1, P2);
    const int t0 = jobs.tgt_begin[job], t1 = jobs.tgt_end[job];
    const int rows = t1 - t0, base_r = rows / SYM_WAVES, extra = rows - base_r * SYM_WAVES;
    const int nr = base_r + (wave < extra ? 1 : 0);
    const int r_lo = t0 + wave * base_r + min(wave, extra); // sorted source position of the wave's first row
    const int64_t q0 = jobs.w_range[2 * job], q1 = jobs.w_range[2 * job + 1];
    const int64_t total = (q1 - q0) * n;
    for (int64_t base = 0; base < total; base += T) {
        const int fill = static_cast<int>(min<int64_t>(T, total - base));
        __syncthreads();
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) {
            const int64_t P = base + j;
            const int ci = static_cast<int>(P / n), I = static_cast<int>(P - static_cast<int64_t>(ci) * n);
            const int cell = jobs.w_cells[q0 + ci];
            const double half = lengths[cell] * 0.5;
            const int i2 = I % P2, i1 = (I / P2) % P1, i0 = I / (P2 * P1); // scale_cheb_nodes_to_cell, chebyshev.rs:951-968
            tile.x[j] = centers[cell * 3] + half * chp->nodes[i0];
            tile.y[j] = d > 1 ? centers[cell * 3 + 1] + half * chp->nodes[i1] : 0.0;
            tile.z[j] = d > 2 ? centers[cell * 3 + 2] + half * chp->nodes[i2] : 0.0;
            tile.w[0][j] = M[static_cast<int64_t>(cell) * n_pad + I];
            tile.col[0][j] = 0.0;
            tile.cidx[j] = cell * n_pad + I;
        }
        __syncthreads();
        // (a partition's output holds its own rows only: out_off = first owned row; the rows of a leaf outside are here for
        // their column sums -- P2L into the partition's cells -- alone)
        sym3_rows<KID, MAXR, true>(ks, tile, fill, lane, r_lo, r_lo - out_off, nr, src, ws, out, 0, out_n);
        __syncthreads();
        for (int j = tid; j < fill; j += 64 * SYM_WAVES) unsafeAtomicAdd(&L[tile.cidx[j]], tile.col[0][j]);
    }
}

// particle_to_local (bbfmm.rs:1001-1048).  One workgroup per cell with an X list; the
// targets are the cell's Chebyshev nodes.
template <int KID, int KB>
__global__ __launch_bounds__(256) void p2l_kernel(KernelSpec ks, const DevCheb *__restrict__ chp,
                                                  const int32_t *__restrict__ cells,
                                                  const int64_t *__restrict__ run_ptr,
                                                  const int32_t *__restrict__ runs,
                                                  const double *__restrict__ centers,
                                                  const double *__restrict__ lengths, Xyz src,
                                                  const double *__restrict__ ws, int64_t N, int k0, int kb, int64_t C,
                                                  double *__restrict__ L) {
    __shared__ SrcTile<KB> tile;
    __shared__ double red[256];
    const int p = chp->p, d = chp->d, n = chp->n, n_pad = chp->n_pad;
    int P0, P1, P2;
    axis_sizes(p, d, P0, P1, P2);
    const int tid = threadIdx.x;
    const int job = blockIdx.x;
    const int cell = cells[job];
    const double half = lengths[cell] * 0.5;
    const double cx = centers[cell * 3], cy = centers[cell * 3 + 1], cz = centers[cell * 3 + 2];
    const int64_t r0 = run_ptr[job], r1 = run_ptr[job + 1];
    const TargetPlan tp = plan_targets(n);
    for (int tc = 0; tc < n; tc += tp.n_c) {
        const int nt = min(tp.n_c, n - tc);
        const int ti = tid % tp.pairs, sl = tid / tp.pairs;
        const bool part = sl < tp.S && ti < nt;
        const bool two = ti + tp.pairs < nt;
        double t[2][3] = {{0, 0, 0}, {0, 0, 0}};
        if (part) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int I = tc + ti + ((h && two) ? tp.pairs : 0);
                const int i2 = I % P2, i1 = (I / P2) % P1, i0 = I / (P2 * P1);
                t[h][0] = cx + half * chp->nodes[i0];
                t[h][1] = d > 1 ? cy + half * chp->nodes[i1] : 0.0;
                t[h][2] = d > 2 ? cz + half * chp->nodes[i2] : 0.0;
            }
        }
        double acc[2][KB], gacc[2][KB][3];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) acc[h][kk] = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const int sb = runs[2 * r], se = runs[2 * r + 1];
            for (int base = sb; base < se; base += DIRECT_TILE) {
                const int cnt = min(DIRECT_TILE, se - base);
                __syncthreads();
                for (int j = tid; j < cnt; j += 256) {
                    tile.xy[j] = make_double2(src.x[base + j], src.y[base + j]);
                    tile.zs[j] = src.z[base + j];
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk)
                        tile.w[kk][j] = kk < kb ? ws[(int64_t)(k0 + kk) * N + base + j] : 0.0;
                }
                __syncthreads();
                if (part) direct_tile<KID, false, KB>(ks, tile, cnt, sl, tp.S, t, acc, gacc);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool wr = part && sl == 0 && (h == 0 || two);
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                if (kk >= kb) break;
                const double v = slice_reduce(acc[h][kk], red, ti, sl, tp.S, tp.pairs, part);
                if (wr) L[((int64_t)(k0 + kk) * C + cell) * n_pad + tc + ti + h * tp.pairs] += v;
            }
        }
    }
}

// ------------------------------------------------------------------ M2L (MFMA FP64)
// multipole_to_local (bbfmm.rs:864-986) regrouped for the matrix cores.  The reference
// permutes each V cell's multipoles onto one of 16 reference operators, multiplies by
// Vt then U, and permutes back.  Here the permutations are folded into per-transfer-
// vector operators stacked per octant class (host side, fmm_m2l_tables.cpp), which turns the
// whole level into two dense streamed-operator GEMMs with no data permutation:
//   stage 1:  Cbuf[target(V,t)][(t,kk)] = sum_m VtAll[(t,kk)][m] * M_V[m]   (X-stationary)
//   stage 2:  L_B[i]                    = sum_k UAll[i][k] * Cbuf[B][k]     (accumulator-stationary)
// ------------------------------------------------------------------ M2L on v_mfma_f64_4x4x4_4b
// Measured on MI355X (scripts/fp64_microbench.hip): v_mfma_f64_16x16x4 sustains ~46 TFLOP/s
// chip-wide (~100-144 cycles per instruction), v_mfma_f64_4x4x4 (4 blocks) ~70 TFLOP/s (17
// cycles per 512-flop instruction).  The kernels below are the same two streamed-operator
// GEMMs as above on the faster instruction.  Lane layout (probed, scripts/mfma4_probe.hip):
//   A[b][i][k]: lane = 16k + 4b + i    B[b][k][j]: lane = 16k + 4b + j    D[b][i][j]: lane = 16i + 4b + j
// with D_b = A_b * B_b for the four independent blocks b.
//
// Stage 1 uses the blocks as four slices of the contraction index (t = lane>>2 = 4k + b selects
// the m values a lane owns), so one operator fragment feeds four cell groups and each D register
// holds four partial sums that are added across lanes (xor 4, xor 8) once per tile.
// Operator tiles are staged with the asynchronous global->LDS DMA (global_load_lds_dwordx4,
// no VGPR round trip) into two LDS buffers: the tile for step i+1 streams in while step i is
// multiplied.  The LDS image is in MFMA-fragment order, so every fragment read is a lane-linear,
// conflict-free ds_read, and the per-lane DMA source address performs the permutation from the
// operator's row-major HBM layout.  The DMA is issued from inline asm: through the builtin the
// compiler must assume the in-flight LDS write may alias every later ds_read and drains it
// (s_waitcnt vmcnt(0)) before each fragment read, which serialises the whole pipeline.
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ inline unsigned lds_offset(const double *p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((lptr_t)p));
}
// Wave-uniform values the compiler cannot prove uniform (derived from threadIdx.x >> 6).
__device__ inline unsigned uniform_u32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline const double *uniform_ptr(const double *p) {
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return reinterpret_cast<const double *>((static_cast<uintptr_t>(hi) << 32) | lo);
}
// 64 lanes x 16 B land at LDS byte offset m0 + lane * 16 (destination = wave-uniform base + lane*16).
__device__ inline void dma16(const double *g, unsigned lds_wave_byte_offset) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g),
                 "s"(uniform_u32(lds_wave_byte_offset))
                 : "memory");
}
// 64 lanes x 4 B (a gather of table entries) land at LDS byte offset m0 + lane * 4.
__device__ inline void dma4(const int32_t *g, unsigned lds_wave_byte_offset) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g),
                 "s"(uniform_u32(lds_wave_byte_offset))
                 : "memory");
}
__device__ inline void wait_dma_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// Same, but lets the N most recent vector-memory operations of the wave stay in flight.  gfx9
// retires loads and stores in issue order on one counter, so when N stores were issued after the
// DMA the DMA has landed once at most N operations are outstanding (the field holds 0..63).
template <int N> __device__ inline void wait_dma_keep_stores_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
    __syncthreads();
}
// v + (v rotated right by N lanes inside each row of 16 lanes), via DPP row_ror
template <int N> __device__ inline double add_row_ror(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int rlo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
    const int rhi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
    return v + __hiloint2double(rhi, rlo);
}

// Same with a uniform (SGPR) base and a 32-bit per-lane byte offset: saves address VGPRs.
__device__ inline void dma16s(const double *sbase, unsigned voff_bytes, unsigned lds_wave_byte_offset) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes),
                 "s"(uniform_ptr(sbase)), "s"(uniform_u32(lds_wave_byte_offset))
                 : "memory");
}

// One kernel serves both stages: OUT[cell][col] = sum_k IN[cell][k] * OP[k][col] for the 128
// cells of a workgroup and a block of 16*NG16 output columns.
//   stage 1: IN = multipoles M (k = Chebyshev node m, n_pad of them), OP = VtAll (n_pad x r_pad),
//            col = stacked operator row (t, kk); the result is scattered into the target slots.
//   stage 2: IN = slot contents (k over the slot, k_pad), OP = UAll (k_pad x n_pad), col = node;
//            the result is the local expansion L.
//   stage 3: a plain product per cell, IN = rows of in_len values per cell (like M), OP = cls.u_all
//            (in_len x n_pad), OUT = rows of n_pad values per cell (like L): the change of basis of the
//            shared-basis extension (multipoles -> coordinates in the level's basis, and back for the locals).
// The four MFMA blocks are four groups of four output columns: A_b = operator fragment
// OP[k][col 4b + i] (one lane-linear ds_read_b64 per 16 columns, feeding four MFMAs), B_b = the
// IN values of four cells (the same for every block), D_b[i][j] = OUT[cell j][col 4b + i].  A wave
// owns 16 cells (four groups tg) and keeps 4 x NG16 accumulators; the operator tile and the
// cells' IN values of step q+1 stream into LDS by DMA while step q is multiplied.
template <int NG16, int STAGE, int MINW>
__global__ __launch_bounds__(512, MINW) void m2l_gemm_k4(const M2lClass *__restrict__ classes,
                                                  const M2lTileDesc *__restrict__ tiles, int n_pad, int g16_0,
                                                  int64_t C, const double *__restrict__ in, int64_t in_len,
                                                  double *__restrict__ out, int64_t out_len,
                                                  const uint16_t *__restrict__ qlist, int slot_t,
                                                  const int32_t *__restrict__ tile_idx) {
    // 2 x { operator [e][ng][k*16 + col], IN tile [wave][tg][eh][k][j] x 2 }; stage 1 adds the slot
    // lookups of the current column block, [wave][cell 0..15][slot_t] int32
    extern __shared__ double lds[];
    const M2lTileDesc tile = tiles[blockIdx.x];
    const M2lClass cls = classes[tile.level_class];
    const int kr = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // stage 1: the workgroup walks the column blocks zb0 .. zb1 of kM2lS1Block stacked rows (g16_0
    // selects a chunk inside a block); stage 2: one block, g16_0 selects the nodes
    const int n_zb = STAGE == 1 ? cls.r_pad16 / kM2lS1Block : 1;
    // (a tile of a sparse plan, pad == 2, names its own column blocks: q_first, q_count)
    const bool own_blocks = STAGE == 1 && tile.pad == 2;
    const int zb0 = STAGE == 1 ? (own_blocks ? tile.q_first : (int)((int64_t)n_zb * blockIdx.z / gridDim.z)) : 0;
    const int zb1 = STAGE == 1 ? (own_blocks ? tile.q_first + tile.q_count : (int)((int64_t)n_zb * (blockIdx.z + 1) / gridDim.z)) : 1;
    const int ld = STAGE == 1 ? cls.r_pad16 : n_pad;            // operator leading dimension
    // contraction steps of 16: all of them in stage 1; in stage 2 only those for which some cell
    // of the tile has a V-list entry (tile.q_first/q_count index the compact list qlist)
    // Stage 2 of a launch with few tiles is split over the contraction as well (slot_t = number of parts, the
    // parameter is stage 1's otherwise): a tile's chain of ~290 steps is what a small tree's matvec waits for.
    // blockIdx.z = part * column chunks + column chunk; the parts add their results to the zeroed L with atomics.
    const int ksplit = STAGE == 2 && slot_t > 1 ? slot_t : 1;
    const int zcols = STAGE == 2 ? (int)gridDim.z / ksplit : 1;
    const int zk = STAGE == 2 ? (int)blockIdx.z / zcols : 0, zc = STAGE == 2 ? (int)blockIdx.z - zk * zcols : (int)blockIdx.z;
    const int q_lo = STAGE == 2 ? tile.q_count * zk / ksplit : 0;
    const int nq = STAGE == 1 ? n_pad / 16 : STAGE == 2 ? tile.q_count * (zk + 1) / ksplit - q_lo : (int)(in_len / 16);
    const uint16_t *ql = qlist + tile.q_first + q_lo;
    if (zb0 >= zb1) return;
    // stage 2 with gridDim.z > 1: the z workgroups of a tile take adjacent chunks of NG16 column groups
    const int g16 = g16_0 + (STAGE >= 2 ? zc * NG16 : 0);
    const double *opbase = (STAGE == 1 ? cls.vt_all : cls.u_all) + 16 * g16;

    constexpr int OP_CHUNKS = 2 * NG16;
    constexpr int NCH = (OP_CHUNKS + 7) / 8;
    constexpr int OP_DOUBLES = OP_CHUNKS * 128;
    constexpr int BUF = OP_DOUBLES + 2048;

    // Operator image in LDS.  The NG16 column groups are taken as NP pairs (+ one single group when
    // NG16 is odd).  A pair chunk (e, pr) is [k][p = 4b + i][2]: lane 16k + p holds columns
    // 32 pr + 2p and + 1 of row 4k + e, fetched as one 16-byte DMA granule and read back with one
    // ds_read_b128 feeding the MFMAs of groups 2 pr and 2 pr + 1.  An accumulator pair of a lane is
    // thus two ADJACENT output columns, which the epilogues store as 16 bytes (the stage-1 scatter
    // is bound by the number of store instructions).  The single group keeps [e][k][16 columns].
    constexpr int NP = NG16 / 2, NS = NG16 & 1;
    unsigned voff[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = wave + 8 * i;
        if (c < 4 * NP) {
            const int e = c / NP, pr = c - e * NP, k = lane >> 4, p = lane & 15;
            voff[i] = (unsigned)(((4 * k + e) * ld + 32 * pr + 2 * p) * 8);
        } else {
            const int e = 2 * (c - 4 * NP) + (lane >> 5), r = lane & 31, k = r >> 3, pair = r & 7;
            voff[i] = (unsigned)(((4 * k + e) * ld + 32 * NP + 2 * pair) * 8);
        }
    }
    // position of the tile's cell `pos` in its class list (a partition's source tiles are compact
    // lists of class positions, tile.pad != 0)
    auto cell_p = [&](int pos) {
        const int q = pos < tile.count ? pos : 0;
        return tile.pad ? tile_idx[tile.first + q] : tile.first + q;
    };
    // IN tile: chunk h of this wave covers tg = 2h + (lane>>5), eh = (lane>>4)&1, k = (lane>>2)&3, j = lane&3
    const double *cptr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tg = 2 * h + (lane >> 5), eh = (lane >> 4) & 1, k = (lane >> 2) & 3, j = lane & 3;
        const int pos = wave * 16 + 4 * tg + j;
        const int p = cell_p(pos);
        const int64_t base = STAGE == 2 ? (int64_t)kr * in_len + cls.cbase[p]
                                        : ((int64_t)kr * C + cls.cells[p]) * (STAGE == 1 ? (int64_t)n_pad : in_len);
        cptr[h] = in + base + 4 * k + 2 * eh;
    }
    const unsigned lds0 = lds_offset(lds);
    const int64_t qstride = (int64_t)16 * ld;
    // step s of the flattened (column block, contraction step) loop
    auto stage = [&](int sidx, int buf) {
        const int zb = zb0 + sidx / nq, qi = sidx - (sidx / nq) * nq;
        const int q = STAGE == 2 ? (int)ql[qi] : qi;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (wave + 8 * i < OP_CHUNKS)
                dma16s(opbase + (int64_t)zb * kM2lS1Block + q * qstride, voff[i],
                       lds0 + (unsigned)(buf * BUF + (wave + 8 * i) * 128) * 8u);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            dma16(cptr[h] + 16 * q, lds0 + (unsigned)(buf * BUF + OP_DOUBLES + (2 * wave + h) * 128) * 8u);
    };

    double acc[4][NG16];
#pragma unroll
    for (int tg = 0; tg < 4; ++tg)
#pragma unroll
        for (int g = 0; g < NG16; ++g) acc[tg][g] = 0.0;

    const int bk = lane >> 4, bj = lane & 3; // B layout (k, j); the block index is broadcast
    const bool wave_live = wave * 16 < tile.count;
    const int n_steps = (zb1 - zb0) * nq;
    // epilogue coordinates: D[b][i][j] at lane 16 i + 4 b + j = OUT[cell 4 tg + j][col0 + 16 g + 4 b + i]
    const int di = lane >> 4, db = (lane >> 2) & 3, dj = lane & 3;
    // Stage-1 scatter tables.  gfx9 retires vector-memory operations in issue order on one counter,
    // and the compiler cannot see the DMA in its wait counting: a wait for any ordinary load in this
    // loop would be vmcnt(0) and drain the DMA in flight and the scatter stores with it.  The loop
    // therefore issues no ordinary loads at all.  All lookups come in by DMA and are read from LDS:
    //   aux[parity][0 .. 16 NG16)  packed row entries of a column block, aux[parity][16 NG16] its first
    //                              transfer-vector position (requested during the previous block)
    //   slots[wave][cell][slot_t]  slot bases of the wave's 16 cells for the block's transfer vectors
    //                              (requested two steps before the block ends)
    // and the stores of a block leave together and drain under the next block.
    constexpr int AUX = 192;
    const int32_t *aux = reinterpret_cast<const int32_t *>(lds + 2 * BUF);
    int32_t *ptab = reinterpret_cast<int32_t *>(lds + 2 * BUF) + 2 * AUX; // class positions of the 128 cells
    const int32_t *slds = aux + 2 * AUX + 128 + wave * 16 * slot_t;
    if (STAGE == 1 && lane < 16) ptab[wave * 16 + lane] = cell_p(wave * 16 + lane); // read by this wave only
    const unsigned aux0 = lds0 + (unsigned)(2 * BUF) * 8u;
    auto stage_cols = [&](int zb_, int par) {
        if (STAGE == 1 && 64 * wave <= 16 * NG16) {
            const int e = 64 * wave + lane;
            const int32_t *src = e < 16 * NG16 ? cls.row_dst + zb_ * kM2lS1Block + 16 * g16 + e : cls.blk_t0 + zb_;
            dma4(src, aux0 + (unsigned)(par * AUX + 64 * wave) * 4u);
        }
    };
    const int slot_sh = 31 - __builtin_clz(slot_t | 1); // slot_t is a power of two >= 16 in stage 1
    auto stage_slots = [&](int par) {
        const int t0 = __builtin_amdgcn_readfirstlane(aux[par * AUX + 16 * NG16]);
        const unsigned dst0 = aux0 + (unsigned)(2 * AUX + 128 + wave * 16 * slot_t) * 4u;
        for (int i = 0; i < slot_t / 4; ++i) {
            const int e = i * 64 + lane, cl = e >> slot_sh, tl = e & (slot_t - 1);
            const int32_t *src = cls.cslot + (int64_t)ptab[wave * 16 + cl] * cls.n_t + min(t0 + tl, cls.n_t - 1);
            dma4(src, dst0 + (unsigned)i * 256u);
        }
    };
    if (n_steps > 0) {
        stage(0, 0);
        stage_cols(zb0, 0);
    }
    wait_dma_and_barrier();
    int qcnt = 0, zb = zb0;
    for (int sidx = 0; sidx < n_steps; ++sidx) {
        const double *op = lds + (sidx & 1) * BUF + lane;
        const double2 *op2 = reinterpret_cast<const double2 *>(lds + (sidx & 1) * BUF) + lane;
        const double *ct = lds + (sidx & 1) * BUF + OP_DOUBLES + wave * 256 + (bk * 4 + bj) * 2;
        if (sidx + 1 < n_steps) stage(sidx + 1, (sidx + 1) & 1); // streams in under the MFMAs below
        if (STAGE == 1) {
            if (qcnt == 0 && zb + 1 < zb1) stage_cols(zb + 1, (zb + 1 - zb0) & 1);
            if (qcnt == nq - 2) stage_slots((zb - zb0) & 1);
        }
        if (wave_live) { // a wave without cells (short tile) only helps with the DMA and the barriers
            double bq[4][4];
#pragma unroll
            for (int tg = 0; tg < 4; ++tg)
#pragma unroll
                for (int eh = 0; eh < 2; ++eh) {
                    const double2 v = *reinterpret_cast<const double2 *>(ct + ((tg * 2 + eh) * 16) * 2);
                    bq[tg][2 * eh] = v.x;
                    bq[tg][2 * eh + 1] = v.y;
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    const double2 a = op2[(e * NP + pr) * 64];
#pragma unroll
                    for (int tg = 0; tg < 4; ++tg) {
                        acc[tg][2 * pr] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, bq[tg][e], acc[tg][2 * pr], 0, 0, 0);
                        acc[tg][2 * pr + 1] =
                            __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, bq[tg][e], acc[tg][2 * pr + 1], 0, 0, 0);
                    }
                }
                if (NS) {
                    const double a = op[4 * NP * 128 + e * 64];
#pragma unroll
                    for (int tg = 0; tg < 4; ++tg)
                        acc[tg][NG16 - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bq[tg][e], acc[tg][NG16 - 1], 0, 0, 0);
                }
            }
        }
        if (++qcnt == nq) { // a column block is complete: write it out, start the next one
            qcnt = 0;
            const int col0 = zb * kM2lS1Block + 16 * g16;
            ++zb;
            if (STAGE == 1 && wave_live) {
                // Scatter into the target slots, branch-free: entries without a destination (padding
                // rows, absent targets, cells beyond the tile) go to a dump area behind the slot buffer.
                // Measured: the epilogue is store-issue bound (about 75 cycles per store instruction
                // and CU whatever its width, the address pattern or the wait after it): adjacent
                // column pairs leave as 16-byte stores, half the instructions of 8-byte ones.
                double *cb = out + (int64_t)kr * out_len;
                double *dump = cb + (out_len - 128) + 2 * lane; // 16-byte aligned (out_len is even)
                const int32_t *auxb = aux + ((zb - 1 - zb0) & 1) * AUX;
                int pk[NP + NS];
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) pk[pr] = auxb[32 * pr + 2 * (4 * db + di)]; // the even column
                if (NS) pk[NP] = auxb[32 * NP + 4 * db + di];
#pragma unroll
                for (int tg = 0; tg < 4; ++tg) {
                    const bool spv = wave * 16 + 4 * tg + dj < tile.count;
                    const int32_t *srow = slds + (4 * tg + dj) * slot_t;
#pragma unroll
                    for (int pr = 0; pr < NP; ++pr) {
                        const int sl = srow[max(pk[pr] >> 24, 0)];
                        const int okm = (spv ? -1 : 0) & ~(pk[pr] | sl); // sign bit set: valid cell, row, slot
                        double *dst = okm < 0 ? cb + (int64_t)sl * 2 + (pk[pr] & 0xffffff) : dump;
                        *reinterpret_cast<double2 *>(dst) = make_double2(acc[tg][2 * pr], acc[tg][2 * pr + 1]);
                    }
                    if (NS) {
                        const int sl = srow[max(pk[NP] >> 24, 0)];
                        const int okm = (spv ? -1 : 0) & ~(pk[NP] | sl);
                        double *dst = okm < 0 ? cb + (int64_t)sl * 2 + (pk[NP] & 0xffffff) : dump;
                        *dst = acc[tg][NG16 - 1];
                    }
                }
            } else if (STAGE >= 2) {
#pragma unroll
                for (int tg = 0; tg < 4; ++tg) {
                    const int tp = wave * 16 + 4 * tg + dj;
                    if (tp < tile.count) {
                        const int cell = cls.cells[cell_p(tp)];
                        double *Lc = out + ((int64_t)kr * C + cell) * n_pad + col0;
                        if (ksplit > 1) { // one of several parts of the contraction
#pragma unroll
                            for (int pr = 0; pr < NP; ++pr) {
                                unsafeAtomicAdd(Lc + 32 * pr + 2 * (4 * db + di), acc[tg][2 * pr]);
                                unsafeAtomicAdd(Lc + 32 * pr + 2 * (4 * db + di) + 1, acc[tg][2 * pr + 1]);
                            }
                            if (NS) unsafeAtomicAdd(Lc + 32 * NP + 4 * db + di, acc[tg][NG16 - 1]);
                        } else {
#pragma unroll
                            for (int pr = 0; pr < NP; ++pr)
                                *reinterpret_cast<double2 *>(Lc + 32 * pr + 2 * (4 * db + di)) =
                                    make_double2(acc[tg][2 * pr], acc[tg][2 * pr + 1]);
                            if (NS) Lc[32 * NP + 4 * db + di] = acc[tg][NG16 - 1];
                        }
                    }
                }
            }
#pragma unroll
            for (int tg = 0; tg < 4; ++tg)
#pragma unroll
                for (int g = 0; g < NG16; ++g) acc[tg][g] = 0.0;
            // the 4 * (NP + NS) scatter stores issued after this step's DMA drain under the next block
            // (a wave without cells stored nothing: it waits for its DMA as usual)
            if (STAGE == 1) {
                if (wave_live) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NP + NS) < 63 ? 4 * (NP + NS) : 63) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                continue;
            }
        }
        wait_dma_and_barrier();
    }
}

// ------------------------------------------------------------------ launch helpers
static Xyz make_xyz(const double *const *p) { return Xyz{p[0], p[1], p[2]}; }

// Dynamic LDS above the 64 KB default needs the function attribute, per device (3-D orders 14-16 of the general
// M2M / L2L kernels): set once per (kernel, device) -- `done` is that kernel's flag word, as in m2l_gemm_launch --
// and a failure is returned to the caller instead of surfacing later as a generic launch error.
static hipError_t allow_large_dynamic_lds(const void *fn, size_t bytes, std::atomic<uint64_t> *done) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool cached = dev >= 0 && dev < 256;
    if (cached && ((done[dev >> 6].load(std::memory_order_acquire) >> (dev & 63)) & 1)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && cached) done[dev >> 6].fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    return e;
}

template <int P, int D>
static void p2m_launch_pd(const ChebRef &ch, Xyz src, const double *w_sorted, int64_t N, int K, int64_t C,
                          const int32_t *leaf_cells, int n_leaves, const int32_t *pt_begin, const int32_t *pt_end,
                          const double *centers, const double *lengths, double *M, hipStream_t s) {
    constexpr int WAVES = p2m_waves<P, D>();
    const int blocks = (n_leaves + WAVES - 1) / WAVES;
    if constexpr (D == 3 && P >= 4 && P <= 10) { // the matrix-pipe version
        hipLaunchKernelGGL((p2m_mfma_kernel<P>), dim3((n_leaves + P2M_WAVES - 1) / P2M_WAVES), dim3(64 * P2M_WAVES), 0, s, ch.dev,
                           n_leaves, src, w_sorted, N, K, C, leaf_cells, pt_begin, pt_end, centers, lengths, M);
        return;
    }
    if (K == 1 || (D == 3 && P > 12)) // (two rhs slots at orders above 12 would spill: one rhs per pass there)
        hipLaunchKernelGGL((p2m_kernel<P, D, 1>), dim3(blocks), dim3(64 * WAVES), 0, s, ch.dev, n_leaves, src,
                           w_sorted, N, K, C, leaf_cells, pt_begin, pt_end, centers, lengths, M);
    else
        hipLaunchKernelGGL((p2m_kernel<P, D, 2>), dim3(blocks), dim3(64 * WAVES), 0, s, ch.dev, n_leaves, src,
                           w_sorted, N, K, C, leaf_cells, pt_begin, pt_end, centers, lengths, M);
}

void launch_p2m(const ChebRef &ch, const double *const *src_xyz, const double *w_sorted, int64_t N, int K, int64_t C,
                const int32_t *leaf_cells, int n_leaves, const int32_t *pt_begin, const int32_t *pt_end,
                const double *centers, const double *lengths, double *M, hipStream_t s) {
    if (n_leaves == 0) return;
    const Xyz src = make_xyz(src_xyz);
#define P2M_CASE(PP)                                                                                               \
    case PP:                                                                                                       \
        if (ch.d == 3) {                                                                                           \
            p2m_launch_pd<PP, 3>(ch, src, w_sorted, N, K, C, leaf_cells, n_leaves, pt_begin, pt_end, centers,      \
                                 lengths, M, s);                                                                   \
        } else if (ch.d == 2) {                                                                                    \
            p2m_launch_pd<PP, 2>(ch, src, w_sorted, N, K, C, leaf_cells, n_leaves, pt_begin, pt_end, centers,      \
                                 lengths, M, s);                                                                   \
        } else {                                                                                                   \
            p2m_launch_pd<PP, 1>(ch, src, w_sorted, N, K, C, leaf_cells, n_leaves, pt_begin, pt_end, centers,      \
                                 lengths, M, s);                                                                   \
        }                                                                                                          \
        break;
    switch (ch.p) {
        P2M_CASE(2) P2M_CASE(3) P2M_CASE(4) P2M_CASE(5) P2M_CASE(6) P2M_CASE(7) P2M_CASE(8) P2M_CASE(9)
        P2M_CASE(10) P2M_CASE(11) P2M_CASE(12) P2M_CASE(13) P2M_CASE(14) P2M_CASE(15) P2M_CASE(16)
    default: break;
    }
#undef P2M_CASE
}

int launch_m2m(const ChebRef &ch, int K, int64_t C, const int32_t *parents, int n_parents, const int64_t *child_ptr,
               const int32_t *child_idx, const int32_t *octant, double *M, hipStream_t s) {
    if (n_parents == 0) return 0;
#define M2M3_CASE(PP)                                                                                              \
    case PP:                                                                                                       \
        hipLaunchKernelGGL((m2m3_kernel<PP>), dim3(n_parents), dim3(512), 0, s, ch.dev, K, C, parents, child_ptr,  \
                           child_idx, octant, M);                                                                  \
        return 0;
    if (ch.d == 3) switch (ch.p) { // wave-per-child register kernels (8 x P^3 doubles of LDS)
            M2M3_CASE(2) M2M3_CASE(3) M2M3_CASE(4) M2M3_CASE(5) M2M3_CASE(6) M2M3_CASE(7) M2M3_CASE(8) M2M3_CASE(9)
            M2M3_CASE(10)
        default: break;
        }
#undef M2M3_CASE
    const size_t lds = sizeof(double) * (3 * (size_t)ch.n + 2 * ch.p * ch.p);
    static std::atomic<uint64_t> attr_set[4] = {{0}, {0}, {0}, {0}};
    if (const hipError_t e = allow_large_dynamic_lds(reinterpret_cast<const void *>(&m2m_kernel), lds, attr_set)) return static_cast<int>(e);
    hipLaunchKernelGGL(m2m_kernel, dim3(n_parents), dim3(256), lds, s, ch.dev, K, C, parents, child_ptr, child_idx,
                       octant, M);
    return 0;
}

int launch_l2l(const ChebRef &ch, int K, int64_t C, const int32_t *cells, int n_cells, const int32_t *parent,
               const int32_t *octant, const uint8_t *active, double *L, hipStream_t s) {
    if (n_cells == 0) return 0;
#define L2L3_CASE(PP)                                                                                              \
    case PP:                                                                                                       \
        hipLaunchKernelGGL((l2l3_kernel<PP>), dim3((n_cells + XFER_WAVES - 1) / XFER_WAVES), dim3(64 * XFER_WAVES),  \
                           0, s, ch.dev, K, C, cells, n_cells, parent, octant, active, L);                         \
        return 0;
    if (ch.d == 3) switch (ch.p) {
            L2L3_CASE(2) L2L3_CASE(3) L2L3_CASE(4) L2L3_CASE(5) L2L3_CASE(6) L2L3_CASE(7) L2L3_CASE(8) L2L3_CASE(9)
            L2L3_CASE(10) L2L3_CASE(11) L2L3_CASE(12)
        default: break;
        }
#undef L2L3_CASE
    const size_t lds = sizeof(double) * (2 * (size_t)ch.n + 2 * ch.p * ch.p);
    static std::atomic<uint64_t> attr_set[4] = {{0}, {0}, {0}, {0}};
    if (const hipError_t e = allow_large_dynamic_lds(reinterpret_cast<const void *>(&l2l_kernel), lds, attr_set)) return static_cast<int>(e);
    hipLaunchKernelGGL(l2l_kernel, dim3(n_cells), dim3(256), lds, s, ch.dev, K, C, cells, parent, octant, active, L);
    return 0;
}

template <int P, int D>
static void l2p_launch_pd(const ChebRef &ch, int n_jobs, const int32_t *leaf_cells, const int32_t *tgt_begin,
                          const int32_t *tgt_end, const double *centers, const double *lengths, Xyz tgt,
                          int64_t n_tgt, int K, int64_t C, const double *L, double *out_sorted, double *grad_sorted,
                          hipStream_t s) {
    constexpr int WAVES = l2p_waves<P, D>();
    const int blocks = (n_jobs + WAVES - 1) / WAVES;
    if (grad_sorted)
        hipLaunchKernelGGL((l2p_kernel<P, D, true>), dim3(blocks), dim3(64 * WAVES), 0, s, ch.dev, n_jobs,
                           leaf_cells, tgt_begin, tgt_end, centers, lengths, tgt, n_tgt, K, C, L, out_sorted,
                           grad_sorted);
    else
        hipLaunchKernelGGL((l2p_kernel<P, D, false>), dim3(blocks), dim3(64 * WAVES), 0, s, ch.dev, n_jobs,
                           leaf_cells, tgt_begin, tgt_end, centers, lengths, tgt, n_tgt, K, C, L, out_sorted,
                           grad_sorted);
}

template <int P>
static void l2p_launch_p(const ChebRef &ch, int n_jobs, const int32_t *leaf_cells, const int32_t *tgt_begin,
                         const int32_t *tgt_end, const double *centers, const double *lengths, Xyz tgt, int64_t n_tgt,
                         int K, int64_t C, const double *L, double *out_sorted, double *grad_sorted, hipStream_t s) {
    if (ch.d == 3) {
        l2p_launch_pd<P, 3>(ch, n_jobs, leaf_cells, tgt_begin, tgt_end, centers, lengths, tgt, n_tgt, K, C, L, out_sorted, grad_sorted, s);
    } else if (ch.d == 2) {
        l2p_launch_pd<P, 2>(ch, n_jobs, leaf_cells, tgt_begin, tgt_end, centers, lengths, tgt, n_tgt, K, C, L, out_sorted, grad_sorted, s);
    } else {
        l2p_launch_pd<P, 1>(ch, n_jobs, leaf_cells, tgt_begin, tgt_end, centers, lengths, tgt, n_tgt, K, C, L, out_sorted, grad_sorted, s);
    }
}

bool l2p_order_supported(int p, int d) { return p >= 2 && p <= kMaxOrder && d >= 1 && d <= 3; }

void launch_l2p(const ChebRef &ch, int n_jobs, const int32_t *leaf_cells, const int32_t *tgt_begin,
                const int32_t *tgt_end, const double *centers, const double *lengths, const double *const *tgt_xyz,
                int64_t n_tgt, int K, int64_t C, const double *L, double *out_sorted, double *grad_sorted,
                hipStream_t s) {
    if (n_jobs == 0) return;
    const Xyz tgt = make_xyz(tgt_xyz);
#define L2P_CASE(PP)                                                                                            \
    case PP:                                                                                                    \
        l2p_launch_p<PP>(ch, n_jobs, leaf_cells, tgt_begin, tgt_end, centers, lengths, tgt, n_tgt, K, C, L,    \
                         out_sorted, grad_sorted, s);                                                           \
        break;
    switch (ch.p) {
        L2P_CASE(2) L2P_CASE(3) L2P_CASE(4) L2P_CASE(5) L2P_CASE(6) L2P_CASE(7) L2P_CASE(8) L2P_CASE(9)
        L2P_CASE(10) L2P_CASE(11) L2P_CASE(12) L2P_CASE(13) L2P_CASE(14) L2P_CASE(15) L2P_CASE(16)
    default: break;
    }
#undef L2P_CASE
}

// Kernel-id dispatch: F is a generic lambda taking std::integral_constant<int, ID>.
template <class F> static void dispatch_kernel_id(int id, F &&f) {
    switch (id) {
    case kLinear: f(std::integral_constant<int, kLinear>{}); break;
    case kThinPlateSpline: f(std::integral_constant<int, kThinPlateSpline>{}); break;
    case kCubic: f(std::integral_constant<int, kCubic>{}); break;
    case kSpheroidal3: f(std::integral_constant<int, kSpheroidal3>{}); break;
    case kSpheroidal5: f(std::integral_constant<int, kSpheroidal5>{}); break;
    case kSpheroidal7: f(std::integral_constant<int, kSpheroidal7>{}); break;
    case kSpheroidal9: f(std::integral_constant<int, kSpheroidal9>{}); break;
    case kLaplacian: f(std::integral_constant<int, kLaplacian>{}); break;
    case kOneOverR2: f(std::integral_constant<int, kOneOverR2>{}); break;
    case kOneOverR4: f(std::integral_constant<int, kOneOverR4>{}); break;
    case kGaussianExt: f(std::integral_constant<int, kGaussianExt>{}); break;
    case kMultiquadricExt: f(std::integral_constant<int, kMultiquadricExt>{}); break;
    default: break;
    }
}

void launch_p2p(const KernelSpec &ks, int d, const DirectJobs &jobs, const double *const *tgt_xyz, int64_t n_tgt,
                const double *const *src_xyz, const double *w_sorted, int64_t N, int K, double *out_sorted,
                double *grad_sorted, hipStream_t s) {
    if (jobs.n_jobs == 0) return;
    dispatch_kernel_id(ks.id, [&](auto idc) {
        constexpr int ID = decltype(idc)::value;
        for (int k0 = 0; k0 < K;) {
            const bool wide = !grad_sorted && K - k0 > DIRECT_KB; // more than four rhs left: eight per pass
            const int kb = std::min(wide ? DIRECT_KB_WIDE : DIRECT_KB, K - k0);
            if (grad_sorted && kb == 1) { // single right-hand side: a quarter of the accumulators
                hipLaunchKernelGGL((p2p_kernel<ID, true, 1>), dim3(jobs.n_jobs), dim3(256), 0, s, ks, d, jobs,
                                   make_xyz(tgt_xyz), n_tgt, make_xyz(src_xyz), w_sorted, N, k0, kb, out_sorted,
                                   grad_sorted);
            } else if (grad_sorted) {
                hipLaunchKernelGGL((p2p_kernel<ID, true, DIRECT_KB>), dim3(jobs.n_jobs), dim3(256), 0, s, ks, d, jobs,
                                   make_xyz(tgt_xyz), n_tgt, make_xyz(src_xyz), w_sorted, N, k0, kb, out_sorted,
                                   grad_sorted);
            } else if (kb == 1) {
                hipLaunchKernelGGL((p2p_kernel<ID, false, 1>), dim3(jobs.n_jobs), dim3(256), 0, s, ks, d, jobs,
                                   make_xyz(tgt_xyz), n_tgt, make_xyz(src_xyz), w_sorted, N, k0, kb, out_sorted,
                                   grad_sorted);
            } else if (wide) {
                hipLaunchKernelGGL((p2p_kernel<ID, false, DIRECT_KB_WIDE>), dim3(jobs.n_jobs), dim3(256), 0, s, ks, d,
                                   jobs, make_xyz(tgt_xyz), n_tgt, make_xyz(src_xyz), w_sorted, N, k0, kb,
                                   out_sorted, grad_sorted);
            } else {
                hipLaunchKernelGGL((p2p_kernel<ID, false, DIRECT_KB>), dim3(jobs.n_jobs), dim3(256), 0, s, ks, d,
                                   jobs, make_xyz(tgt_xyz), n_tgt, make_xyz(src_xyz), w_sorted, N, k0, kb,
                                   out_sorted, grad_sorted);
            }
            k0 += kb;
        }
    });
}

// K right-hand sides in passes of at most kSymMaxRhs = 4: kernel instances for 1, 2 and 4 rhs (three run the 4-slot
// instance).  Measured at 10M points, LinearRbf (scripts/p2p_rhs_sweep.py): 4.4 / 5.1 / 7.3 ms for 1 / 2 / 4 rhs; an
// 8-slot instance (two rows per chunk, 222 VGPRs, two waves per SIMD; also with point-major weights and software-pipelined
// row loads) took 16.6 ms per pass against 2 x 7.3: not kept.
int p2p_sym3_passes();
static int p2p_sym3_max_rows_per_pass() {
    static const int v = [] {
        const char *e = std::getenv("BBFMM_P2P_SYM_LEAF_PASS"); // rows per pass: 8 (default) or 6
        return e && std::atoi(e) == 6 ? 6 : 8;
    }();
    return v;
}
void launch_p2p_sym(const KernelSpec &ks, int n_jobs, const int32_t *tgt_begin, const int32_t *tgt_end,
                    const int64_t *run_range, int n_leaf_jobs, const int32_t *l_tgt_begin, const int32_t *l_tgt_end,
                    const int64_t *l_run_range, int n_wave_jobs, const int32_t *w_tgt_begin, const int32_t *w_tgt_end,
                    const int64_t *w_run_range, const int32_t *runs3, int32_t tgt_off, const double *const *src_xyz,
                    const double *w_sorted, int64_t ldw, int K, double *out_sorted, int64_t ldo, hipStream_t s) {
    dispatch_kernel_id(ks.id, [&](auto idc) {
        constexpr int ID = decltype(idc)::value;
        const SymJobs wj{n_wave_jobs, w_tgt_begin, w_tgt_end, w_run_range, runs3, tgt_off};
        const SymJobs gj{n_jobs, tgt_begin, tgt_end, run_range, runs3, tgt_off};
        const SymJobs lj{n_leaf_jobs, l_tgt_begin, l_tgt_end, l_run_range, runs3, tgt_off};
        const Xyz src = make_xyz(src_xyz);
        for (int k0 = 0; k0 < K; k0 += kSymMaxRhs) {
            const int kb = std::min(kSymMaxRhs, K - k0);
            const double *w = w_sorted + static_cast<int64_t>(k0) * ldw;
            double *o = out_sorted + static_cast<int64_t>(k0) * ldo;
#define SYM_GO(KBV)                                                                                                   \
    do {                                                                                                              \
        if (n_wave_jobs > 0)                                                                                          \
            hipLaunchKernelGGL((p2p_sym2_kernel<ID, KBV>), dim3((n_wave_jobs + SYM2_WAVES - 1) / SYM2_WAVES),          \
                               dim3(64 * SYM2_WAVES), 0, s, ks, wj, src, w, ldw, kb, o, ldo);                         \
        if (KBV == 1 && n_leaf_jobs > 0) { /* one rhs: whole big leaves (the same rows as the chunk jobs below) */     \
            if (p2p_sym3_max_rows_per_pass() == 8)                                                                    \
                hipLaunchKernelGGL((p2p_sym3_kernel<ID, 8>), dim3(n_leaf_jobs), dim3(64 * SYM_WAVES), 0, s, ks, lj, src, w, o); \
            else                                                                                                      \
                hipLaunchKernelGGL((p2p_sym3_kernel<ID, 6>), dim3(n_leaf_jobs), dim3(64 * SYM_WAVES), 0, s, ks, lj, src, w, o); \
        } else if (n_jobs > 0)                                                                                        \
            hipLaunchKernelGGL((p2p_sym_kernel<ID, KBV>), dim3(n_jobs), dim3(64 * SYM_WAVES), 0, s, ks, gj, src, w,    \
                               ldw, kb, o, ldo);                                                                      \
    } while (0)
            if (kb == 1) SYM_GO(1);
            else if (kb == 2) SYM_GO(2);
            else SYM_GO(4);
#undef SYM_GO
        }
    });
}

// Leaves of at most this many rows are one job of the wave-per-job kernel (BBFMM_P2P_SYM_WAVE=<rows>, 0: none).
// Measured on MI355X: 38-point leaves (10M uniform points) 5.55 -> 4.46 ms with the wave kernel; 153- and 238-point
// leaves (40M, 1M points) are faster in the workgroup kernel (89.7 against 103 ms, 2.5 against 4.7 ms).
int p2p_sym_wave_rows() {
    static const int rows = [] {
        const char *e = std::getenv("BBFMM_P2P_SYM_WAVE");
        const int v = e ? std::atoi(e) : 64;
        return v < 0 ? 0 : (v > SYM2_MAX_ROWS ? SYM2_MAX_ROWS : v);
    }();
    return rows;
}

// A launch of the wave kernel with fewer jobs than this leaves the chip to a handful of waves per SIMD, each walking its
// whole leaf alone: such trees give every leaf to a workgroup instead (eight waves share its rows).  Measured on MI355X,
// uniform points, near field per matvec, wave kernel -> workgroup kernels: 512 leaves (20k / 36k points) 0.055 -> 0.018 /
// 0.130 -> 0.035 ms, 4,096 leaves (150k / 200k / 300k) 0.134 -> 0.083 / 0.131 -> 0.112 / 0.288 -> 0.203 ms; 32,768 leaves
// (1.6M / 2M) 0.71 -> 0.89 / 1.11 -> 1.23 ms the other way.  Default 48 jobs per CU; BBFMM_P2P_SYM_WAVE_MIN=<jobs> overrides.
static int device_cu_count();
int64_t p2p_sym_wave_min_jobs() {
    const char *e = std::getenv("BBFMM_P2P_SYM_WAVE_MIN"); // read per plan (a handle's job lists are built once): the tests
    if (e && *e) return std::max<int64_t>(0, std::atoll(e)); // run small trees through either kind of job in one process
    return int64_t(48) * device_cu_count();
}

int p2p_sym_rows_per_job() { return SYM_WAVES * SYM_TR; }
// Whole-leaf jobs of the one-rhs workgroup kernel: rows per job (BBFMM_P2P_SYM_LEAF=<rows>; 0: no such jobs, the chunk
// jobs serve one rhs too).  A job of R rows gives each of the eight waves R / 8 of them.
int p2p_sym3_passes() {
    static const int v = [] {
        const char *e = std::getenv("BBFMM_P2P_SYM_LEAF");
        const int x = e ? std::atoi(e) : 512; // (305-point leaves -- max_points_per_cell 512 at 10M points -- as ONE job: 24.3 -> 23.0 ms with 8 rows per pass)
        return x <= 0 ? 0 : std::max(x, SYM_WAVES * SYM_TR);
    }();
    return v;
}
int p2p_sym3_rows_per_job() { return p2p_sym3_passes(); }
int wx_sym_rows_per_job() { return SYM_WAVES * SYM_TR; }
// rows per whole-leaf job of the one-rhs fused pass (BBFMM_WX_SYM_LEAF=<rows>; 0: none, the chunk jobs serve one rhs too)
int wx_sym3_rows_per_job() {
    static const int v = [] {
        const char *e = std::getenv("BBFMM_WX_SYM_LEAF");
        const int x = e ? std::atoi(e) : 256;
        return x <= 0 ? 0 : std::max(x, SYM_WAVES * SYM_TR);
    }();
    return v;
}

void launch_wx_sym(const KernelSpec &ks, const ChebRef &ch, int n_jobs, const int32_t *tgt_begin, const int32_t *tgt_end,
                   const int64_t *w_range, int n_leaf_jobs, const int32_t *l_tgt_begin, const int32_t *l_tgt_end,
                   const int64_t *l_w_range, const int32_t *w_cells, const double *centers, const double *lengths,
                   const double *const *src_xyz, const double *w_sorted, int64_t ldw, int K, const double *M, double *L,
                   int64_t ld_ml, double *out_sorted, int64_t ldo, int out_off, int out_n, hipStream_t s) {
    if (n_jobs == 0) return;
    const WxJobs jobs{n_jobs, tgt_begin, tgt_end, w_range, w_cells};
    const WxJobs ljobs{n_leaf_jobs, l_tgt_begin, l_tgt_end, l_w_range, w_cells};
    dispatch_kernel_id(ks.id, [&](auto idc) {
        constexpr int ID = decltype(idc)::value;
        for (int k0 = 0; k0 < K; k0 += kSymMaxRhs) { // (an 8-slot instance would be bound by its LDS traffic)
            const int kb = std::min(kSymMaxRhs, K - k0);
#define WX_GO(KBV)                                                                                                    \
    hipLaunchKernelGGL((wx_sym_kernel<ID, KBV>), dim3(n_jobs), dim3(64 * SYM_WAVES), 0, s, ks, jobs, ch.dev, centers,    \
                       lengths, make_xyz(src_xyz), w_sorted + static_cast<int64_t>(k0) * ldw, ldw, kb,                 \
                       M + static_cast<int64_t>(k0) * ld_ml, L + static_cast<int64_t>(k0) * ld_ml, ld_ml,               \
                       out_sorted + static_cast<int64_t>(k0) * ldo, ldo, out_off, out_n)
            if (kb == 1 && n_leaf_jobs > 0 && p2p_sym3_max_rows_per_pass() == 8) // one rhs: whole leaves (the same rows and W cells as the chunk jobs)
                hipLaunchKernelGGL((wx_sym3_kernel<ID, 8>), dim3(n_leaf_jobs), dim3(64 * SYM_WAVES), 0, s, ks, ljobs, ch.dev, centers, lengths,
                                   make_xyz(src_xyz), w_sorted + static_cast<int64_t>(k0) * ldw, M + static_cast<int64_t>(k0) * ld_ml,
                                   L + static_cast<int64_t>(k0) * ld_ml, out_sorted + static_cast<int64_t>(k0) * ldo, out_off, out_n);
            else if (kb == 1 && n_leaf_jobs > 0)
                hipLaunchKernelGGL((wx_sym3_kernel<ID, 6>), dim3(n_leaf_jobs), dim3(64 * SYM_WAVES), 0, s, ks, ljobs, ch.dev, centers, lengths,
                                   make_xyz(src_xyz), w_sorted + static_cast<int64_t>(k0) * ldw, M + static_cast<int64_t>(k0) * ld_ml,
                                   L + static_cast<int64_t>(k0) * ld_ml, out_sorted + static_cast<int64_t>(k0) * ldo, out_off, out_n);
            else if (kb == 1) WX_GO(1);
            else if (kb == 2) WX_GO(2);
            else WX_GO(4);
#undef WX_GO
        }
    });
}

void launch_m2p(const KernelSpec &ks, const ChebRef &ch, int n_jobs, const int32_t *tgt_begin,
                const int32_t *tgt_end, const int64_t *w_begin, const int64_t *w_end,
                const int32_t *w_cells, const double *centers,
                const double *lengths, const double *const *tgt_xyz, int64_t n_tgt, int K, int64_t C,
                const double *M, double *out_sorted, double *grad_sorted, hipStream_t s) {
    if (n_jobs == 0) return;
    dispatch_kernel_id(ks.id, [&](auto idc) {
        constexpr int ID = decltype(idc)::value;
        for (int k0 = 0; k0 < K; k0 += DIRECT_KB) {
            const int kb = std::min(DIRECT_KB, K - k0);
#define M2P_GO(GR, KBV)                                                                                           \
    hipLaunchKernelGGL((m2p_kernel<ID, GR, KBV>), dim3(n_jobs), dim3(256), 0, s, ks, ch.dev, tgt_begin, tgt_end,   \
                       w_begin, w_end, w_cells, centers, lengths, make_xyz(tgt_xyz), n_tgt, k0, kb, C, M,          \
                       out_sorted, grad_sorted)
            if (grad_sorted && kb == 1) M2P_GO(true, 1);
            else if (grad_sorted) M2P_GO(true, DIRECT_KB);
            else if (kb == 1) M2P_GO(false, 1);
            else M2P_GO(false, DIRECT_KB);
#undef M2P_GO
        }
    });
}

void launch_p2l(const KernelSpec &ks, const ChebRef &ch, int n_jobs, const int32_t *cells, const int64_t *run_ptr,
                const int32_t *runs, const double *centers, const double *lengths, const double *const *src_xyz,
                const double *w_sorted, int64_t N, int K, int64_t C, double *L, hipStream_t s) {
    if (n_jobs == 0) return;
    dispatch_kernel_id(ks.id, [&](auto idc) {
        constexpr int ID = decltype(idc)::value;
        for (int k0 = 0; k0 < K; k0 += DIRECT_KB) {
            const int kb = std::min(DIRECT_KB, K - k0);
            if (kb == 1)
                hipLaunchKernelGGL((p2l_kernel<ID, 1>), dim3(n_jobs), dim3(256), 0, s, ks, ch.dev, cells, run_ptr, runs,
                                   centers, lengths, make_xyz(src_xyz), w_sorted, N, k0, kb, C, L);
            else
                hipLaunchKernelGGL((p2l_kernel<ID, DIRECT_KB>), dim3(n_jobs), dim3(256), 0, s, ks, ch.dev, cells,
                                   run_ptr, runs, centers, lengths, make_xyz(src_xyz), w_sorted, N, k0, kb, C, L);
        }
    });
}

template <int NG16, int STAGE, int MINW>
static void m2l_gemm_launch(const M2lClass *classes, const M2lTileDesc *tiles, int n_tiles, int n_pad, int g16_0,
                            int n_colblocks, int K, int64_t C, const double *in, int64_t in_len, double *out,
                            int64_t out_len, const uint16_t *qlist, int slot_t, const int32_t *tile_idx, hipStream_t s) {
    const size_t lds = 2 * sizeof(double) * (size_t)(2 * NG16 * 128 + 2048) + (STAGE == 1 ? (size_t)(2 * 192 + 128 + 8 * 16 * (slot_t & 0xffff)) * 4 : 0); // + aux, cell and slot tables
    // function attributes are per device: one flag per (template instance, device), set from whichever thread
    // launches there first (handles are bound to their device and may be used from any host thread)
    static std::atomic<uint64_t> attr_set[4] = {{0}, {0}, {0}, {0}};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 256 || !((attr_set[dev >> 6].load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&m2l_gemm_k4<NG16, STAGE, MINW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 256) attr_set[dev >> 6].fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    const int zdim = STAGE == 2 && slot_t > 1 ? n_colblocks * slot_t : n_colblocks; // stage 2: slot_t = parts of the contraction
    hipLaunchKernelGGL((m2l_gemm_k4<NG16, STAGE, MINW>), dim3(n_tiles, K, zdim), dim3(512), lds, s, classes,
                       tiles, n_pad, g16_0, C, in, in_len, out, out_len, qlist, slot_t, tile_idx);
}

// Column-chunk plan: 16-column groups per workgroup.  Stage 1 walks column blocks of kM2lS1Block =
// 11 groups (44 accumulators per lane; 22 spills in the persistent walk).  Stage 2 covers the n_pad
// output nodes with 22-group chunks (88 accumulators, one workgroup per CU), measured faster there
// than 11 or 8.

template <int STAGE> constexpr int m2l_chunk_pref() { return STAGE == 1 ? kM2lS1Block / 16 : 22; }
constexpr int kM2lS2KsplitFill = 4; // workgroups per CU up to which stage 2 keeps splitting the contraction

template <int STAGE>
static void m2l_dispatch_chunks(int total_groups, const M2lClass *classes, const M2lTileDesc *tiles, int n_tiles,
                                int n_pad, int n_colblocks, int K, int64_t C, const double *in, int64_t in_len,
                                double *out, int64_t out_len, const uint16_t *qlist, int slot_t,
                                const int32_t *tile_idx, hipStream_t s) {
    int done = 0;
    const int pref = m2l_chunk_pref<STAGE>();
    while (done < total_groups) {
        const int left = total_groups - done;
        int take;
#define M2L_GO(NG, MW)                                                                                              \
    {                                                                                                               \
        take = NG;                                                                                                  \
        m2l_gemm_launch<NG, STAGE, MW>(classes, tiles, n_tiles, n_pad, done,                                        \
                                       STAGE >= 2 ? 1 : n_colblocks, K, C, in, in_len,                            \
                                       out, out_len, qlist, slot_t, tile_idx, s);                                   \
    }
        if (STAGE >= 2 && !(n_colblocks == 1 && left >= 16)) { // (one workgroup per tile and a wide chunk: the plan below)
            // stage 2 / 3 with several workgroups per tile (gridDim.z = zc): the largest NG x zc <= left from the
            // kernels without spills, at most 2 x 11 or 3 x (8, 7, 6, 4, 2) -- 22 = 2 x 11, 46 = 3 x 8 + 2 x 11,
            // 18 = 3 x 6, 16 = 2 x 8, 7 = 1 x 7
            int best_ng = 2, best_z = 1;
            for (int ng : {11, 8, 7, 6, 4, 2})
                for (int zc = 1; zc <= (n_colblocks == 1 ? 1 : ng == 11 ? 2 : 3); ++zc)
                    if (ng * zc <= left && ((left - ng * zc) % 2 == 0 || left - ng * zc == 7) && // what remains must be coverable
                        (ng * zc > best_ng * best_z || (ng * zc == best_ng * best_z && ng > best_ng))) {
                        best_ng = ng;
                        best_z = zc;
                    }
            take = best_ng * best_z;
#define M2L_ZGO(NG, MW)                                                                                             \
    m2l_gemm_launch<NG, STAGE, MW>(classes, tiles, n_tiles, n_pad, done, best_z, K, C, in, in_len, out, out_len,    \
                                   qlist, slot_t, tile_idx, s);
            if constexpr (STAGE >= 2) {
                switch (best_ng) {
                case 11: M2L_ZGO(11, 1) break;
                case 8: M2L_ZGO(8, 4) break;
                case 7: M2L_ZGO(7, 2) break;
                case 6: M2L_ZGO(6, 4) break;
                case 4: M2L_ZGO(4, 4) break;
                default: M2L_ZGO(2, 4) break;
                }
            }
#undef M2L_ZGO
        } else if (pref == 22 && left >= 22) M2L_GO(22, 1)
        else if (pref == 22 && left >= 16) M2L_GO(16, 1)
        else if (pref == 11 && left >= 11) M2L_GO(11, 1)
        else if (left >= 8) M2L_GO(8, 4)
        else if (left >= 6) M2L_GO(6, 4)
        else if (left >= 4) M2L_GO(4, 4)
        else M2L_GO(2, 4)
#undef M2L_GO
        done += take;
    }
}

static int device_cu_count() { // of the current device (every entry point binds its thread to the handle's device)
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int n_cu = 256;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    if (dev >= 0 && dev < 64) cache[dev].store(n_cu, std::memory_order_relaxed);
    return n_cu;
}

// Stage 1: every class's stacked operator is padded to a whole number of kM2lS1Block columns;
// blockIdx.z walks the column blocks, the chunk plan splits a block.
void launch_m2l_stage1(const M2lClass *classes, const M2lTileDesc *tiles, const int32_t *tile_idx, int n_tiles,
                       int n_pad, int max_slot_t, int K, int64_t C, const double *M, double *cbuf, int64_t cbuf_len,
                       hipStream_t s, bool own_blocks, int max_blocks) {
    if (n_tiles == 0) return;
    int slot_t = 16; // LDS slot-table width: power of two covering the transfer vectors of any block
    while (slot_t < max_slot_t) slot_t *= 2;
    // every workgroup walks its share of the column blocks; splitting the walk over gridDim.z
    // workgroups shortens the last, partially filled round of the launch
    const int n_cu = device_cu_count();
    // One workgroup per CU at a time: n_tiles * z workgroups take ceil(n_tiles * z / CUs) rounds of 1/z of the
    // column-block walk each; z is chosen so that the last, partially filled round is short (a per-workgroup
    // overhead of about half a percent of a walk keeps z small).  Measured at 10M points (2,336 tiles): z = 2 18.2 ms,
    // 3 18.0, 4 17.75, 7 18.0, 13 18.3.
    // A launch that does not fill the chip even at z = 8 (a small tree: 16 tiles at 36k points, where a workgroup's walk of
    // four column blocks WAS the stage: 0.155 ms of a 0.46 ms matvec) may split the walk down to one block per workgroup.
    const int zmax = n_tiles * 8 < n_cu ? std::max(8, std::min(max_blocks, 32)) : 8;
    int zsplit = 1;
    double best = 1e300;
    for (int z = 1; z <= zmax; ++z) {
        const double rounds = std::ceil(static_cast<double>(n_tiles) * z / n_cu);
        const double cost = rounds / z * (1.0 + 0.005 * z);
        if (cost < best - 1e-12) {
            best = cost;
            zsplit = z;
        }
    }
    const int n_colblocks = own_blocks ? 1 : zsplit; // tiles that name their own blocks are not split further
    m2l_dispatch_chunks<1>(kM2lS1Block / 16, classes, tiles, n_tiles, n_pad, n_colblocks, K, C, M, 0, cbuf, cbuf_len, nullptr,
                           slot_t, tile_idx, s);
}

// Stage 2: the output nodes (n_pad, in groups of 16; n_pad is a multiple of 32) in column chunks.
void launch_m2l_stage2(const M2lClass *classes, const M2lTileDesc *tiles, const int32_t *tile_idx, int n_tiles,
                       int n_pad, int K, int64_t C, const double *cbuf, int64_t cbuf_len, const uint16_t *qlist,
                       double *L, hipStream_t s, bool allow_ksplit) {
    if (n_tiles == 0) return;
    // Two workgroups per tile, each with half of a 22-group chunk of the output nodes (gridDim.z = 2, 11 groups,
    // 78 KB of LDS: two fit a CU).  The halves of a tile read the same slot contents at about the same time (L2
    // hits) and twice as many, shorter workgroups balance tiles of unequal length (per-tile active steps) and fill
    // the CUs of launches with few tiles.  Measured, stage 2 per matvec: 10M points 16.2 -> 15.8 ms, p = 9 55.1 ->
    // 53.6 (52.5 with the remainder as 3 x 8 instead of 22 + 2), 1M points (about 200 tiles) 2.10 -> 1.54 and 1.12 -> 0.71,
    // one rank of an 8-way partition 2.81 -> 2.28.  (Two separate launches of 11 groups were slower than one of 22.)
    // BBFMM_M2L_S2_ZSPLIT=1 turns it off.
    static const int z = [] {
        const char *e = std::getenv("BBFMM_M2L_S2_ZSPLIT");
        return e && std::atoi(e) == 1 ? 1 : 2;
    }();
    // Few tiles (a small tree, a thin slice of a partition): also split the contraction, so that no workgroup walks a
    // tile's chain of ~290 dependent steps alone.  The parts add to L (zeroed by the caller before every downward pass) with
    // f64 atomics; launches that fill the chip keep plain stores.
    // Stage 2 per matvec: 50k points 0.50 -> 0.11 ms (the whole matvec 0.79 -> 0.39 ms), 200k 0.52 -> 0.43, 1M 0.95 -> 0.78;
    // from 3M points on the launch fills the chip and nothing changes.  BBFMM_M2L_S2_KSPLIT=<n> overrides (1: off).
    // Round 6: the number of parts from what was measured instead of a power of two -- a CU works its workgroups off one after
    // the other (two resident ones share its matrix pipe), a part costs a fixed 4-5 % of a whole chain plus its share of it,
    // so the launch takes ceil(workgroups * parts / CUs) * (0.05 + 1 / parts) chains: 585 cells (32 workgroups) 16 -> 8 parts,
    // 0.105 -> 0.085 ms; 4,681 cells (158 workgroups) 4 -> 3 parts, 0.419 -> 0.368 ms (5 parts: 0.462, 8: 0.411, 2: 0.511).
    static const int ks_env = [] {
        const char *e = std::getenv("BBFMM_M2L_S2_KSPLIT");
        const int v = e ? std::atoi(e) : 0;
        return v >= 1 && v <= 32 ? v : 0;
    }();
    int ksplit = allow_ksplit ? ks_env : 1; // (the parts add atomically: not for BBFMM_FLAG_DETERMINISTIC handles)
    if (ksplit == 0) {
        const int n_cu = device_cu_count();
        const int64_t wgs = static_cast<int64_t>(n_tiles) * z * K;
        ksplit = 1;
        if (wgs * 2 <= static_cast<int64_t>(kM2lS2KsplitFill) * n_cu) { // (launches of at most two workgroups per CU, as before)
            double best = 1e300;
            for (int ks = 1; ks <= 16; ++ks) {
                const double cost = std::ceil(static_cast<double>(wgs * ks) / n_cu) * (0.05 + 1.0 / ks);
                if (cost < best - 1e-12) {
                    best = cost;
                    ksplit = ks;
                }
            }
        }
    }
    m2l_dispatch_chunks<2>(n_pad / 16, classes, tiles, n_tiles, n_pad, z, K, C, cbuf, cbuf_len, L, 0, qlist, ksplit > 1 ? ksplit : 0, tile_idx, s);
}

// Slot segments of absent pairs: 16 lanes per segment, 16 bytes per lane and round.
__global__ __launch_bounds__(256) void m2l_zero_segments_kernel(const int32_t *__restrict__ segs, int64_t n_segs,
                                                                double *__restrict__ cbuf, int64_t cbuf_len) {
    const int64_t sg = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (sg >= n_segs) return;
    const int2 e = reinterpret_cast<const int2 *>(segs)[sg];
    double2 *dst = reinterpret_cast<double2 *>(cbuf + static_cast<int64_t>(blockIdx.y) * cbuf_len) + e.x;
    for (int i = l; i < e.y; i += 16) dst[i] = make_double2(0.0, 0.0);
}

void launch_m2l_zero_segments(const int32_t *segs, int64_t n_segs, int K, double *cbuf, int64_t cbuf_len, hipStream_t s) {
    if (n_segs <= 0) return;
    hipLaunchKernelGGL(m2l_zero_segments_kernel, dim3(static_cast<unsigned>((n_segs * 16 + 255) / 256), K), dim3(256), 0, s, segs,
                       n_segs, cbuf, cbuf_len);
}

// Shared-basis extension: change of basis of every cell of a level (stage 3 of the GEMM kernel), OUT[cell][0..out_ld) =
// sum_k IN[cell][k] * OP_level[k][0..out_ld).  classes[].cells / u_all: the level's cells and its in_ld x out_ld operator.
void launch_m2l_basis(const M2lClass *classes, const M2lTileDesc *tiles, int n_tiles, int in_ld, int out_ld, int K,
                      int64_t C, const double *in, double *out, hipStream_t s) {
    if (n_tiles == 0) return;
    m2l_dispatch_chunks<3>(out_ld / 16, classes, tiles, n_tiles, out_ld, 2, K, C, in, in_ld, out, 0, nullptr, 0, nullptr, s);
}

// Setup-time product for the projected operators: C[i][j] = sum_k A(i, k) * B[k][j], A(i, k) = A[k * lda + i] (TA) or
// A[i * lda + k]; 64 x 64 tiles, 4 x 4 per thread, 16 contraction indices per LDS round.
template <bool TA>
__global__ __launch_bounds__(256) void small_gemm_kernel(int M, int N, int Kd, const double *__restrict__ A, int64_t lda,
                                                         const double *__restrict__ B, int64_t ldb, double *__restrict__ Cm,
                                                         int64_t ldc) {
    __shared__ double sa[16][64 + 1], sb[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < Kd; k0 += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int kk = TA ? e >> 6 : e & 15, ii = TA ? e & 63 : e >> 4;
            const int k = k0 + kk, i = i0 + ii;
            sa[kk][ii] = (k < Kd && i < M) ? (TA ? A[(int64_t)k * lda + i] : A[(int64_t)i * lda + k]) : 0.0;
        }
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int kk = e >> 6, jj = e & 63;
            const int k = k0 + kk, j = j0 + jj;
            sb[kk][jj] = (k < Kd && j < N) ? B[(int64_t)k * ldb + j] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = sa[kk][ty + 16 * r];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = sb[kk][tx + 16 * c];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += a[r] * b[c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + ty + 16 * r, j = j0 + tx + 16 * c;
            if (i < M && j < N) Cm[(int64_t)i * ldc + j] = acc[r][c];
        }
}

void launch_small_gemm(bool trans_a, int M, int N, int Kd, const double *A, int64_t lda, const double *B, int64_t ldb,
                       double *Cm, int64_t ldc, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
    if (trans_a) hipLaunchKernelGGL(small_gemm_kernel<true>, grid, dim3(256), 0, s, M, N, Kd, A, lda, B, ldb, Cm, ldc);
    else hipLaunchKernelGGL(small_gemm_kernel<false>, grid, dim3(256), 0, s, M, N, Kd, A, lda, B, ldb, Cm, ldc);
}

// ------------------------------------------------------------------ stacked M2L operators, assembled in HBM
// fmm_m2l_tables.cpp fill_m2l_operator_arrays on the device: the reference operators of a level (16 in 3-D) and the
// symmetry tables go up once (MBs), the stacked per-class operators (GBs) are gathered from them here instead
// of being filled on the host and sent over PCIe.  VtAll[m][first_row(t) + kk] = Vt_ref(t)[kk][invperm_t[m]]
// (identity rows when uncompressed); UAll[tgt_off(t) + kk][i] = U_ref(t)[invperm_t[i]][kk].
__global__ __launch_bounds__(256) void assemble_vt_kernel(M2lAssembleClass c, int n, int n_pad, int compressed,
                                                          const double *__restrict__ ops, const int32_t *__restrict__ invperm,
                                                          double *__restrict__ vt_all) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= static_cast<int64_t>(c.n_src) * n) return;
    const int pos = static_cast<int>(t / n), m = static_cast<int>(t % n);
    const M2lAssembleTv tv = c.src[pos];
    const int im = invperm[static_cast<int64_t>(tv.perm) * n + m];
    double *dst = vt_all + static_cast<int64_t>(m) * c.r_pad16 + tv.row;
    if (compressed) {
        const double *src = ops + tv.vt_off + static_cast<int64_t>(im) * tv.rank;
        for (int kk = 0; kk < tv.rank; ++kk) dst[kk] = src[kk];
    } else {
        dst[im] = 1.0;
    }
}

__global__ __launch_bounds__(256) void assemble_u_kernel(M2lAssembleClass c, int n, int n_pad, const double *__restrict__ ops,
                                                         const int32_t *__restrict__ invperm, double *__restrict__ u_all) {
    const int pos = blockIdx.y;
    const M2lAssembleTv tv = c.tgt[pos];
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= static_cast<int64_t>(tv.rank) * n) return;
    const int kk = static_cast<int>(t / n), i = static_cast<int>(t % n);
    u_all[static_cast<int64_t>(tv.row + kk) * n_pad + i] = ops[tv.u_off + static_cast<int64_t>(kk) * n + invperm[static_cast<int64_t>(tv.perm) * n + i]];
}

void launch_m2l_assemble(const M2lAssembleClass &c, int n, int n_pad, bool compressed, const double *ops,
                         const int32_t *invperm, double *vt_all, double *u_all, hipStream_t s) {
    (void)hipMemsetAsync(vt_all, 0, static_cast<size_t>(n_pad) * c.r_pad16 * sizeof(double), s);
    (void)hipMemsetAsync(u_all, 0, static_cast<size_t>(c.k_pad) * n_pad * sizeof(double), s);
    if (c.n_src > 0)
        // (one thread per entry, no grid-stride loop: the exact block count, not grid_for's capped one)
        hipLaunchKernelGGL(assemble_vt_kernel, dim3(static_cast<unsigned>((static_cast<int64_t>(c.n_src) * n + 255) / 256)), dim3(256), 0, s, c, n, n_pad,
                           compressed ? 1 : 0, ops, invperm, vt_all);
    if (c.n_tgt > 0 && c.max_rank > 0)
        hipLaunchKernelGGL(assemble_u_kernel, dim3(static_cast<unsigned>((static_cast<int64_t>(c.max_rank) * n + 255) / 256), c.n_tgt), dim3(256), 0, s, c,
                           n, n_pad, ops, invperm, u_all);
}

// ------------------------------------------------------------------ MFMA self test
// Four independent 4x4x4 products: A[b][i][k], B[b][k][j], D[b][i][j] (row-major per block).
__global__ void mfma_layout_kernel(const double *A, const double *B, double *D) {
    const int lane = threadIdx.x & 63;
    const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
    const double a = A[(b * 4 + lo) * 4 + hi];  // A: lane = 16k + 4b + i
    const double bb = B[(b * 4 + hi) * 4 + lo]; // B: lane = 16k + 4b + j
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, c, 0, 0, 0);
    D[(b * 4 + hi) * 4 + lo] = c;               // D: lane = 16i + 4b + j
}

__global__ __launch_bounds__(256) void mfma_peak_kernel(double *sink, int iters, unsigned long long *stamps) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[q], 0, 0, 0);
    }
    double sum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += c[q];
    if (sum == 12345.678) sink[0] = sum;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (stamps && (threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;     // shader cycles
        stamps[2 * w + 1] = r1 - r0; // 100 MHz ticks
    }
}

int mfma_f64_selftest(double *tflops, int *layout_errors, double *info) {
    double hA[64], hB[64], hD[256], ref[256];
    // exact small integers, asymmetric B (catches a transposed or block-swapped map)
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 4; ++k) {
                hA[(b * 4 + i) * 4 + k] = (double)(1 + b * 17 + i * 5 + k * 3);
                hB[(b * 4 + i) * 4 + k] = (double)(2 + b * 11 + i * 7 + k * k); // B[b][k=i][j=k]
            }
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += hA[(b * 4 + i) * 4 + k] * hB[(b * 4 + k) * 4 + j];
                ref[(b * 4 + i) * 4 + j] = s;
            }
    double *dA = nullptr, *dB = nullptr, *dD = nullptr;
    if (hipMalloc(&dA, sizeof hA) != hipSuccess) return 1;
    if (hipMalloc(&dB, sizeof hB) != hipSuccess) return 1;
    if (hipMalloc(&dD, sizeof hD) != hipSuccess) return 1;
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int errs = 0;
    for (int i = 0; i < 64; ++i)
        if (hD[i] != ref[i]) ++errs;
    *layout_errors = errs;
    // peak of v_mfma_f64_4x4x4 (512 flop): `blocks` x 4 waves, 8 independent accumulators per wave.  info[0..5]:
    //   [0] cycles per MFMA, one wave alone on a CU        [1] its shader clock (MHz)
    //   [2] cycles per MFMA per SIMD, 1 wave/SIMD, all CUs [3] clock under that load (MHz)
    //   [4] TFLOP/s at 1 wave/SIMD                          [5] TFLOP/s at 2 waves/SIMD
    const int iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    unsigned long long *dS = nullptr;
    const int max_blocks = 2048;
    (void)hipMalloc(&dS, sizeof(unsigned long long) * 2 * 4 * max_blocks);
    std::vector<unsigned long long> hS(2 * 4 * max_blocks);
    auto run = [&](int blocks, int threads, double *cyc, double *mhz) {
        hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(threads), 0, 0, dD, 16, nullptr); // warm-up
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(threads), 0, 0, dD, iters, dS);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const int waves = blocks * (threads / 64);
        (void)hipMemcpy(hS.data(), dS, sizeof(unsigned long long) * 2 * 4 * blocks, hipMemcpyDeviceToHost);
        double sc = 0, sr = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < threads / 64; ++w) {
                sc += (double)hS[2 * (b * 4 + w)];
                sr += (double)hS[2 * (b * 4 + w) + 1];
            }
        if (cyc) *cyc = sc / waves / (8.0 * iters);
        if (mhz) *mhz = sr > 0 ? sc / sr * 100.0 : 0.0;
        return (double)waves * iters * 8.0 * 512.0 / (ms * 1e-3) / 1e12;
    };
    double i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    run(1, 64, &i0, &i1);
    const double tf1 = run(256, 256, &i2, &i3);
    const double tf2 = run(512, 256, nullptr, nullptr);
    const double tf8 = run(2048, 256, nullptr, nullptr);
    *tflops = tf8 > tf2 ? (tf8 > tf1 ? tf8 : tf1) : (tf2 > tf1 ? tf2 : tf1);
    if (info) {
        info[0] = i0; info[1] = i1; info[2] = i2; info[3] = i3; info[4] = tf1; info[5] = tf2;
    }
    (void)hipFree(dS);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(dA);
    (void)hipFree(dB);
    (void)hipFree(dD);
    return 0;
}

// ------------------------------------------------------------------ FP64 vector-ALU peak (the pair kernels' roofline)
// Eight independent v_fma_f64 chains per lane: what the vector pipe sustains chip-wide, and at which clock (the
// FP64 load pulls the shader clock well under the 2.4 GHz of AMD's 78.6 TFLOP/s).
__global__ __launch_bounds__(256) void valu_peak_kernel(double *sink, int iters, unsigned long long *stamps) {
    double a = 1.0 + threadIdx.x * 1e-9;
    const double b = 1.0 - 1e-9;
    double c[8] = {0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = fma(c[q], b, a);
    }
    double sum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += c[q];
    if (sum == 12345.678) sink[0] = sum;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (stamps && (threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

// tflops: FMA flops per second chip-wide (2 per lane and instruction); mhz: the shader clock during the run.
int valu_f64_selftest(double *tflops, double *mhz) {
    const int iters = 8192, blocks = 4096;
    double *sink = nullptr;
    unsigned long long *dS = nullptr;
    if (hipMalloc(&sink, 64) != hipSuccess) return 1;
    if (hipMalloc(&dS, sizeof(unsigned long long) * 2 * 4 * blocks) != hipSuccess) {
        (void)hipFree(sink);
        return 1;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        (void)hipFree(sink);
        (void)hipFree(dS);
        return 1;
    }
    hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, 0, sink, 64, nullptr); // warm-up
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, 0, sink, iters, dS);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hS(2 * 4 * blocks);
    (void)hipMemcpy(hS.data(), dS, sizeof(unsigned long long) * hS.size(), hipMemcpyDeviceToHost);
    double sc = 0, sr = 0;
    for (size_t w = 0; w < (size_t)4 * blocks; ++w) {
        sc += (double)hS[2 * w];
        sr += (double)hS[2 * w + 1];
    }
    *tflops = (double)blocks * 256 * iters * 8.0 * 2.0 / (ms * 1e-3) / 1e12;
    *mhz = sr > 0 ? sc / sr * 100.0 : 0.0;
    (void)hipFree(sink);
    (void)hipFree(dS);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

} // namespace bbfmm

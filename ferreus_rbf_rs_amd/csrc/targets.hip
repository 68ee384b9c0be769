// See targets.hpp.
#include "targets.hpp"

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace bbfmm {
namespace {

constexpr uint64_t kLevelDisp = 15; // morton_constants.rs:15

__device__ inline uint64_t f64_to_u64_sat(double v) { // Rust `f64 as u64` (morton.rs:46)
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return ~uint64_t(0);
    return static_cast<uint64_t>(v);
}

__device__ inline uint64_t spread(uint64_t v, int d) { // bit i -> bit i*d (morton.rs:58-119)
    v &= 0xFFFF;
    if (d == 1) return v;
    if (d == 2) {
        v = (v | (v << 8)) & 0x00FF00FFull;
        v = (v | (v << 4)) & 0x0F0F0F0Full;
        v = (v | (v << 2)) & 0x33333333ull;
        v = (v | (v << 1)) & 0x55555555ull;
        return v;
    }
    v = (v | (v << 16)) & 0x0000FF0000FFull;
    v = (v | (v << 8)) & 0x00F00F00F00Full;
    v = (v | (v << 4)) & 0x0C30C30C30C3ull;
    v = (v | (v << 2)) & 0x249249249249ull;
    return v;
}

__device__ inline uint64_t mix(uint64_t x) { // KeyTable::hash
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

__device__ inline int32_t table_find(const DevLeafLookup &lk, uint64_t key) {
    uint64_t h = mix(key) & lk.mask;
    while (true) {
        const int32_t v = lk.vals[h];
        if (v < 0) return -1;
        if (lk.keys[h] == key) return v;
        h = (h + 1) & lk.mask;
    }
}

__global__ __launch_bounds__(256) void points_to_leaves_kernel(DevLeafLookup lk, const double *__restrict__ x0,
                                                               const double *__restrict__ x1,
                                                               const double *__restrict__ x2, int64_t m,
                                                               int32_t *__restrict__ cell, unsigned long long *bad_row) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= m) return;
    const double *xs[3] = {x0, x1, x2};
    uint64_t code = 0;
    for (int a = 0; a < lk.d; ++a) // morton.rs:35-51
        code |= spread(f64_to_u64_sat(floor((xs[a][i] - lk.disp[a]) / lk.side)), lk.d) << a;
    uint64_t cur = (code << kLevelDisp) | static_cast<uint64_t>(lk.depth);
    int32_t found = -1;
    while (true) { // linear_tree.rs:505-508
        const int32_t j = lk.mask ? table_find(lk, cur) : -1;
        if (j >= 0 && lk.is_leaf[j]) {
            found = j;
            break;
        }
        const uint64_t level = cur & 0x7FFF;
        if (level == 0) break;
        cur = (((cur >> kLevelDisp) >> lk.d) << kLevelDisp) | (level - 1);
    }
    cell[i] = found;
    if (found < 0) atomicMin(bad_row, static_cast<unsigned long long>(i)); // smallest failing row (linear_tree.rs:514-517)
}

__global__ __launch_bounds__(256) void heads_kernel(const int32_t *__restrict__ cs, int64_t m, uint8_t *__restrict__ heads) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < m) heads[i] = (i == 0 || cs[i] != cs[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void runs_kernel(const int32_t *__restrict__ cs, int64_t m, const int32_t *__restrict__ n_runs,
                                                   const int32_t *__restrict__ tgt_begin, int32_t *__restrict__ job_cell,
                                                   int32_t *__restrict__ tgt_end) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n = *n_runs;
    if (r >= n) return;
    job_cell[r] = cs[tgt_begin[r]];
    tgt_end[r] = r + 1 < n ? tgt_begin[r + 1] : static_cast<int32_t>(m);
}

__global__ __launch_bounds__(256) void gather_targets_kernel(const double *__restrict__ in0, const double *__restrict__ in1,
                                                             const double *__restrict__ in2, const int32_t *__restrict__ perm,
                                                             int64_t m, double *__restrict__ out0, double *__restrict__ out1,
                                                             double *__restrict__ out2) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= m) return;
    const int32_t p = perm[i];
    if (out0) out0[i] = in0[p];
    if (out1) out1[i] = in1[p];
    if (out2) out2[i] = in2[p];
}

inline unsigned blocks(int64_t n) { return static_cast<unsigned>((n + 255) / 256); }

size_t sort_bytes(int64_t m, int end_bit) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, static_cast<const int32_t *>(nullptr), static_cast<int32_t *>(nullptr),
                                    rocprim::counting_iterator<int32_t>(0), static_cast<int32_t *>(nullptr),
                                    static_cast<size_t>(m), 0u, static_cast<unsigned>(end_bit));
    return b;
}
size_t select_bytes(int64_t m) {
    size_t b = 0;
    (void)rocprim::select(nullptr, b, rocprim::counting_iterator<int32_t>(0), static_cast<const uint8_t *>(nullptr),
                          static_cast<int32_t *>(nullptr), static_cast<int32_t *>(nullptr), static_cast<size_t>(m));
    return b;
}

} // namespace

void launch_points_to_leaves(const DevLeafLookup &lk, const double *x0, const double *x1, const double *x2, int64_t m,
                             int32_t *cell, unsigned long long *bad_row, hipStream_t s) {
    if (m <= 0) return;
    hipLaunchKernelGGL(points_to_leaves_kernel, dim3(blocks(m)), dim3(256), 0, s, lk, x0, x1, x2, m, cell, bad_row);
}

size_t group_targets_temp_bytes(int64_t m, int end_bit) {
    if (m <= 0) return 0;
    return std::max(sort_bytes(m, end_bit), select_bytes(m));
}

int group_targets(const int32_t *cell, int64_t m, int end_bit, int32_t *cell_sorted, int32_t *perm, uint8_t *heads,
                  int32_t *job_cell, int32_t *tgt_begin, int32_t *tgt_end, int32_t *n_runs, void *temp, size_t temp_bytes,
                  hipStream_t s) {
    if (m <= 0) return static_cast<int>(hipMemsetAsync(n_runs, 0, sizeof(int32_t), s));
    size_t b = temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(temp, b, cell, cell_sorted, rocprim::counting_iterator<int32_t>(0), perm,
                                             static_cast<size_t>(m), 0u, static_cast<unsigned>(end_bit), s);
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL(heads_kernel, dim3(blocks(m)), dim3(256), 0, s, cell_sorted, m, heads);
    b = temp_bytes;
    e = rocprim::select(temp, b, rocprim::counting_iterator<int32_t>(0), heads, tgt_begin, n_runs, static_cast<size_t>(m), s);
    if (e != hipSuccess) return static_cast<int>(e);
    // at most one run per row; the launch covers the capacity and reads the count on the device
    hipLaunchKernelGGL(runs_kernel, dim3(blocks(m)), dim3(256), 0, s, cell_sorted, m, n_runs, tgt_begin, job_cell, tgt_end);
    return static_cast<int>(hipGetLastError());
}

void launch_gather_targets(const double *in0, const double *in1, const double *in2, const int32_t *perm, int64_t m,
                           double *out0, double *out1, double *out2, hipStream_t s) {
    if (m <= 0) return;
    hipLaunchKernelGGL(gather_targets_kernel, dim3(blocks(m)), dim3(256), 0, s, in0, in1, in2, perm, m, out0, out1, out2);
}

} // namespace bbfmm

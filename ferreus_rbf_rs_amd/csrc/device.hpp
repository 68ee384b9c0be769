// Device-side data structures and kernel launchers (implemented in device.hip).
//
// HBM layout (all f64 unless noted; "sorted" = hierarchical Morton order in which
// every tree cell owns a contiguous range of points):
//   src_xyz      d arrays of N sorted source coordinates (SoA -> coalesced loads)
//   w_sorted     K x N   sorted weights, rhs-major
//   M, L         K x C x n_pad   multipole / local coefficients, cell-major,
//                n_pad = n rounded up to 32 (rows >= n stay zero) so that one
//                cell's column is a whole number of 128-B lines and of MFMA tiles
//   cbuf         K x cbuf_len   compressed M2L intermediates c = Vt * M (per target,
//                per transfer vector; see M2lLevel)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace bbfmm {

constexpr int kMaxOrder = 16;      // Chebyshev nodes per axis supported on device
constexpr int kM2lTile = 128;      // cells per M2L workgroup: 8 waves x 16
constexpr int kM2lS1Block = 176;   // stacked-operator rows per stage-1 column block (11 groups of 16)

struct DevCheb { // lives in device memory; kernels take a pointer
    int p, d, n, n_pad;
    double polyn[kMaxOrder * kMaxOrder]; // T_k(node_j), row j
    double nodes[kMaxOrder];
    double xfer[2 * kMaxOrder * kMaxOrder]; // [side][child node a][parent node i]
};

struct ChebRef { // device pointer + the host copies the launchers size their grids with
    const DevCheb *dev;
    int p, d, n, n_pad;
};

// One tree level of M2L work (levels 2..depth).  Cells of the level are grouped by
// octant class o (position inside the parent); all cells of a class share the set
// of admissible transfer vectors t, hence one stacked ("tall") operator per class.
struct M2lClass {
    // stage 1 (source side): c[(t,kk)] = sum_m VtAllT[m][row] * M_V[m]
    const double *vt_all;   // n_pad x r_pad16 (row m contiguous over tall rows)
    const int32_t *row_dst;  // r_pad16: -1 (padding row) or ((tpos - blk_t0[block]) << 24) | off, with tpos the
                             // position of the row's transfer vector in the class list and off the offset of
                             // the row inside the target's slot (= off_target_class[t] + kk)
    const int32_t *blk_t0;   // r_pad16 / kM2lS1Block: first tpos of each column block
    int32_t n_rows;          // exact number of tall rows
    int32_t r_pad16;         // n_rows rounded up to a multiple of kM2lS1Block
    int32_t n_t;             // number of transfer vectors of this class (189 in 3-D)
    // stage 2 (target side): L_B[i] = sum_k UAllT[k][i] * ccat_B[k]
    const double *u_all;     // k_pad x n_pad
    int32_t k_pad;           // slot length of a target of this class (multiple of 16)
    // cells of this class at this level
    const int32_t *cells;    // cell indices
    int32_t n_cells;
    const int32_t *cslot;    // n_cells x n_t: slot base (units of 2 doubles) of target V+t, -1 if absent
    const int64_t *cbase;    // n_cells: slot base of the cell itself as a target (units of doubles)
};

struct M2lTileDesc { // one workgroup of stage 1 or stage 2
    int32_t level_class; // index into the M2lClass table
    int32_t first;       // first cell (position inside the class list)
    int32_t count;       // <= kM2lTile
    int32_t q_first;     // stage 2: first entry / number of entries of the tile's active-step list
    int32_t q_count;
    int32_t pad;         // 0: cells first..first+count of the class; 1: first indexes tile_idx (class positions);
                         // 2: as 1, and a stage-1 tile covers only the column blocks q_first..q_first+q_count
};

// ---- launchers (all asynchronous on `s`) ----
void launch_gather_weights(const double *w, int64_t ldw, int K, const int32_t *order, int64_t N,
                           double *w_sorted, hipStream_t s);
void launch_gather_weights_subset(const double *w, int64_t ldw, int K, const int32_t *order, const int32_t *pos, int64_t n_pos,
                                  int64_t N, double *w_sorted, hipStream_t s);
// Gathered potentials of a partitioned matvec (all[part][k][m_max], each part's rows in the tree's sorted order) to their
// rows of out: part r holds the sorted points bound[r] .. bound[r + 1).
constexpr int kMaxScatterParts = 64;
struct ScatterParts {
    int n;
    int64_t bound[kMaxScatterParts + 1];
};
void launch_scatter_parts(const double *all, const ScatterParts &parts, int64_t m_max, int K, const int32_t *order, double *out,
                          int64_t ldo, hipStream_t s);
// out[i] = sum over the slots, in slot order (slots[g * len + i]): the in-process exchange of a device group
void launch_sum_slots(const double *slots, int n_slots, int64_t len, double *out, hipStream_t s);
void launch_scatter_output(const double *out_sorted, int64_t n, int K, const int32_t *perm,
                           double *out, int64_t ldo, int accumulate, hipStream_t s);
void launch_gather_rows(const double *src, int64_t ld_src, int ncols, const int32_t *idx, int64_t n,
                        double *dst, int64_t ld_dst, hipStream_t s);

void launch_p2m(const ChebRef &ch, const double *const *src_xyz, const double *w_sorted, int64_t N,
                int K, int64_t C, const int32_t *leaf_cells, int n_leaves, const int32_t *pt_begin,
                const int32_t *pt_end, const double *centers, const double *lengths, double *M,
                hipStream_t s);
// (both return 0 or the hipError_t of a refused function attribute: dynamic LDS above 64 KB at 3-D orders 14-16)
int launch_m2m(const ChebRef &ch, int K, int64_t C, const int32_t *parents, int n_parents,
               const int64_t *child_ptr, const int32_t *child_idx, const int32_t *octant, double *M,
               hipStream_t s);
int launch_l2l(const ChebRef &ch, int K, int64_t C, const int32_t *cells, int n_cells,
               const int32_t *parent, const int32_t *octant, const uint8_t *active, double *L,
               hipStream_t s);

// tile_idx: NULL, or for tiles with pad != 0 the class positions of the tile's cells
// (tile.first indexes tile_idx; a partition's compact source tiles)
void launch_m2l_stage1(const M2lClass *classes, const M2lTileDesc *tiles, const int32_t *tile_idx, int n_tiles,
                       int n_pad, int max_slot_t, int K, int64_t C, const double *M, double *cbuf,
                       int64_t cbuf_len, hipStream_t s, bool own_blocks = false, int max_blocks = 8);
void launch_m2l_stage2(const M2lClass *classes, const M2lTileDesc *tiles, const int32_t *tile_idx, int n_tiles,
                       int n_pad, int K, int64_t C, const double *cbuf, int64_t cbuf_len,
                       const uint16_t *qlist, double *L, hipStream_t s, bool allow_ksplit = true);
// Zeroes the slot segments of absent V pairs of a batch: segs = (slot / 2, length / 2) pairs, K buffers of cbuf_len.
void launch_m2l_zero_segments(const int32_t *segs, int64_t n_segs, int K, double *cbuf, int64_t cbuf_len, hipStream_t s);
// shared-basis extension: per-cell change of basis (OUT[cell] = IN[cell] * OP_level), setup-time dense product
void launch_m2l_basis(const M2lClass *classes, const M2lTileDesc *tiles, int n_tiles, int in_ld, int out_ld, int K,
                      int64_t C, const double *in, double *out, hipStream_t s);
void launch_small_gemm(bool trans_a, int M, int N, int Kd, const double *A, int64_t lda, const double *B, int64_t ldb,
                       double *Cm, int64_t ldc, hipStream_t s);

// Assembly of one class's stacked operators on the device (device.hip "stacked M2L operators").
struct M2lAssembleTv { // one transfer vector of a class list
    int32_t perm;    // symmetry permutation (row of invperm)
    int32_t rank;
    int32_t row;     // source side: first stacked row; target side: offset inside the slot
    int64_t vt_off;  // offsets of the reference operator's factors inside the level's operator buffer
    int64_t u_off;
};
struct M2lAssembleClass {
    const M2lAssembleTv *src, *tgt; // device arrays
    int32_t n_src, n_tgt, r_pad16, k_pad, max_rank;
};
void launch_m2l_assemble(const M2lAssembleClass &c, int n, int n_pad, bool compressed, const double *ops,
                         const int32_t *invperm, double *vt_all, double *u_all, hipStream_t s);

// Direct (kernel-evaluating) interactions.  Targets are sorted by leaf; job i handles
// the targets [tgt_begin[i], tgt_end[i]) of leaf job_cell[i] against the source runs
// runs[run_ptr[cell] .. run_ptr[cell+1]) (pairs of [begin, end) into the sorted sources;
// adjacent U-list leaves are merged into one run on the host).
struct DirectJobs {
    int n_jobs;
    const int32_t *job_cell;
    const int32_t *tgt_begin, *tgt_end;
    const int64_t *run_ptr; // per cell
    const int32_t *runs;    // 2 ints per run
};
void launch_p2p(const KernelSpec &ks, int d, const DirectJobs &jobs, const double *const *tgt_xyz,
                int64_t n_tgt, const double *const *src_xyz, const double *w_sorted, int64_t N, int K,
                double *out_sorted, double *grad_sorted, hipStream_t s);
// P2P for targets that are (a contiguous sorted range of) the sources: every unordered pair once, one rhs.
// Job i holds the target positions [tgt_begin[i], tgt_end[i]) -- at most p2p_sym_rows_per_job() of them, all of
// one leaf; position p is sorted source tgt_off + p; its leaf's runs are runs3[3 * run_range[2i] .. 3 * run_range[2i+1])
// = (begin, end, two_sided) triples of sorted source ranges: two-sided runs also receive the column sums (they
// must lie inside the target range and after the leaf), one-sided runs only feed the job's rows.  out_sorted
// (n_tgt values, zeroed by the caller) is accumulated with f64 atomics.
// Two job lists over the same runs: (n_jobs, tgt_begin, tgt_end, run_range) row chunks of big leaves for the
// workgroup-per-job kernel, (n_wave_jobs, w_*) whole small leaves (at most p2p_sym_wave_rows() rows) for the
// wave-per-job kernel.
// K right-hand sides (rhs k: weights w_sorted + k * ldw, sums out_sorted + k * ldo) go in passes of at most
// kSymMaxRhs, one kernel evaluation per unordered pair and pass.
constexpr int kSymMaxRhs = 4;
// (n_leaf_jobs, l_*): the big leaves again as WHOLE-leaf jobs (at most p2p_sym3_rows_per_job() rows: bigger leaves in
// equal parts) for the one-rhs workgroup kernel (round 6); a pass of one rhs takes these instead of the chunk jobs
// (0 of them: the chunk jobs serve one rhs too).
void launch_p2p_sym(const KernelSpec &ks, int n_jobs, const int32_t *tgt_begin, const int32_t *tgt_end,
                    const int64_t *run_range, int n_leaf_jobs, const int32_t *l_tgt_begin, const int32_t *l_tgt_end,
                    const int64_t *l_run_range, int n_wave_jobs, const int32_t *w_tgt_begin, const int32_t *w_tgt_end,
                    const int64_t *w_run_range, const int32_t *runs3, int32_t tgt_off, const double *const *src_xyz,
                    const double *w_sorted, int64_t ldw, int K, double *out_sorted, int64_t ldo, hipStream_t s);
int p2p_sym_rows_per_job();
int p2p_sym3_rows_per_job(); // 0: no whole-leaf jobs (BBFMM_P2P_SYM_LEAF=0)
int p2p_sym_wave_rows(); // 0: no wave jobs (BBFMM_P2P_SYM_WAVE=0)
int64_t p2p_sym_wave_min_jobs(); // fewer wave-sized leaves than this in a job list: none of them goes to the wave kernel
int wx_sym_rows_per_job();
int wx_sym3_rows_per_job(); // 0: no whole-leaf jobs (BBFMM_WX_SYM_LEAF=0)
// M2P + P2L in one pass for targets = all sources (X = W^T; K rhs in passes of kSymMaxRhs: rhs k at w_sorted + k * ldw,
// M / L + k * ld_ml, out_sorted + k * ldo): jobs are row chunks of the leaves that have a
// W list, w_range their ranges in w_cells; row sums go to out_sorted (M2P), column sums to L (P2L), both atomically.
// (n_leaf_jobs, l_*): the same rows and W cells as whole-leaf jobs (at most wx_sym3_rows_per_job() rows) for a pass of one
// rhs (round 6; 0 of them: the chunk jobs serve one rhs too)
void launch_wx_sym(const KernelSpec &ks, const ChebRef &ch, int n_jobs, const int32_t *tgt_begin, const int32_t *tgt_end,
                   const int64_t *w_range, int n_leaf_jobs, const int32_t *l_tgt_begin, const int32_t *l_tgt_end,
                   const int64_t *l_w_range, const int32_t *w_cells, const double *centers, const double *lengths,
                   const double *const *src_xyz, const double *w_sorted, int64_t ldw, int K, const double *M, double *L,
                   int64_t ld_ml, double *out_sorted, int64_t ldo, int out_off, int out_n, hipStream_t s);
// M2P: sources are the Chebyshev nodes of the W-list cells, weights their multipoles.  Job i
// covers targets [tgt_begin[i], tgt_end[i]) and the W cells w_cells[w_begin[i] .. w_end[i]).
void launch_m2p(const KernelSpec &ks, const ChebRef &ch, int n_jobs, const int32_t *tgt_begin,
                const int32_t *tgt_end, const int64_t *w_begin, const int64_t *w_end,
                const int32_t *w_cells,
                const double *centers, const double *lengths, const double *const *tgt_xyz,
                int64_t n_tgt, int K, int64_t C, const double *M, double *out_sorted,
                double *grad_sorted, hipStream_t s);
// P2L: targets are the Chebyshev nodes of the cell, sources the points of the X-list leaves.
void launch_p2l(const KernelSpec &ks, const ChebRef &ch, int n_jobs, const int32_t *cells,
                const int64_t *run_ptr, const int32_t *runs, const double *centers,
                const double *lengths, const double *const *src_xyz, const double *w_sorted,
                int64_t N, int K, int64_t C, double *L, hipStream_t s);
void launch_l2p(const ChebRef &ch, int n_jobs, const int32_t *leaf_cells, const int32_t *tgt_begin,
                const int32_t *tgt_end, const double *centers, const double *lengths,
                const double *const *tgt_xyz, int64_t n_tgt, int K, int64_t C, const double *L,
                double *out_sorted, double *grad_sorted, hipStream_t s);

// Orders the device Chebyshev kernels are instantiated for (3-D: p <= 12).
bool l2p_order_supported(int p, int d);

// FP64 MFMA lane-layout check + peak microbenchmark.
int mfma_f64_selftest(double *tflops, int *layout_errors, double *info);
// FP64 vector-ALU peak: chip-wide v_fma_f64 rate (TFLOP/s) and the shader clock it runs at (MHz).
int valu_f64_selftest(double *tflops, double *mhz);

} // namespace bbfmm

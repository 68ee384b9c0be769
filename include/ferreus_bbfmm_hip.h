/*
 * ferreus_bbfmm_hip.h -- C ABI of the MI355X-native BBFMM evaluator.
 *
 * Drop-in boundary for the hot path of graphic-goose/ferreus_rbf_rs: the
 * `ferreus_bbfmm` matrix-vector product behind the type-erased evaluator
 * `ferreus_rbf_utils::FmmTree` (ferreus_rbf_utils/src/utils.rs:383-494).  The
 * reference has no FFI for this path; every entry point below cites the Rust
 * method it replaces (paths relative to the reference repository root).  A
 * Rust `extern "C"` shim / ctypes stub binding these symbols is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - all matrices are f64, column-major with an explicit leading dimension
 *     (faer::Mat layout); `pts` are N x d (column a at pts + a*ld).
 *   - one in-flight call per handle (the reference wraps the tree in a Mutex,
 *     ferreus_rbf/src/rbf.rs:87,106,124); a handle may be used from any thread.
 *   - nothing throws across the ABI; every function returns a bbfmm_status.
 *   - the library fails loudly (BBFMM_DEVICE_ERROR) when no HIP device is
 *     usable: there is no CPU fallback for the compute entry points.
 */
#ifndef FERREUS_BBFMM_HIP_H
#define FERREUS_BBFMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Return codes.  1 and 2 mirror FmmError (ferreus_bbfmm/src/bbfmm.rs:20-27). */
typedef enum {
    BBFMM_OK = 0,
    BBFMM_POINT_OUTSIDE_TREE = 1,         /* FmmError::PointOutsideTree{point_index} */
    BBFMM_KERNEL_NO_GRADIENTS = 2,        /* FmmError::KernelDoesNotSupportGradients */
    BBFMM_BAD_ARGUMENT = 3,               /* the reference panics (e.g. bbfmm.rs:293-298) */
    BBFMM_DEVICE_ERROR = 4,               /* HIP failure or no device */
    BBFMM_UNSUPPORTED = 5                 /* valid in the reference, not built here yet */
} bbfmm_status;

/* KernelType, in registry order (ferreus_rbf_utils/src/utils.rs:558-571).
 * Ids >= 100 are extension kernels of this repository that the reference does
 * not ship (BASELINE.json configs name them); their only oracle is the dense sum. */
typedef enum {
    BBFMM_KERNEL_LINEAR_RBF = 0,
    BBFMM_KERNEL_THIN_PLATE_SPLINE_RBF = 1,
    BBFMM_KERNEL_CUBIC_RBF = 2,
    BBFMM_KERNEL_SPHEROIDAL3_RBF = 3,
    BBFMM_KERNEL_SPHEROIDAL5_RBF = 4,
    BBFMM_KERNEL_SPHEROIDAL7_RBF = 5,
    BBFMM_KERNEL_SPHEROIDAL9_RBF = 6,
    BBFMM_KERNEL_LAPLACIAN = 7,
    BBFMM_KERNEL_ONE_OVER_R2 = 8,
    BBFMM_KERNEL_ONE_OVER_R4 = 9,
    BBFMM_KERNEL_GAUSSIAN_EXT = 100,      /* exp(-(r/base_range)^2)      -- extension */
    BBFMM_KERNEL_MULTIQUADRIC_EXT = 101   /* sqrt(1+(r/base_range)^2)    -- extension */
} bbfmm_kernel_type;

/* M2LCompressionType (ferreus_bbfmm/src/bbfmm.rs:62-73). */
typedef enum {
    BBFMM_COMPRESSION_NONE = 0,
    BBFMM_COMPRESSION_SVD = 1,
    BBFMM_COMPRESSION_ACA = 2
} bbfmm_compression_type;

/* FmmParams (ferreus_bbfmm/src/bbfmm.rs:77-104). */
typedef struct {
    int64_t max_points_per_cell;   /* default 256 */
    int32_t compression_type;      /* bbfmm_compression_type, default ACA */
    double epsilon;                /* default 10^-interpolation_order */
    int64_t eval_chunk_size;       /* default 1024 (host chunking knob of the reference; kept for API parity) */
} bbfmm_params;

/* Fills FmmParams::new_defaults(interpolation_order) (bbfmm.rs:96-103). */
void bbfmm_params_defaults(int32_t interpolation_order, bbfmm_params *out);

typedef struct bbfmm_handle bbfmm_handle;

/* Creation flags. */
#define BBFMM_FLAG_HOST_ONLY 1u /* build tree/lists/operators on the host only; no device is
                                   touched and every compute call returns BBFMM_DEVICE_ERROR.
                                   Used by the CPU-side structure tests. */
#define BBFMM_FLAG_M2L_SHARED_BASIS 2u /* EXTENSION beyond the reference (off by default): the M2L stages run on
                                        * coordinates in one orthonormal basis per level (the dominant subspace of
                                        * all of the level's compressed operators, cut at params.epsilon by the
                                        * operators' own rule: rank 107 of 343 for LinearRbf at order 7, 183-259 of 729
                                        * for the thin-plate spline at order 9), with the reference's factors
                                        * projected onto it.  About a third of the M2L flops; results
                                        * differ from the default path by the projection error (a few epsilon of the
                                        * far field; tests/test_gpu_shared_basis.py).  Needs a compressed operator
                                        * type (ACA or SVD) and a device; when the basis would keep more than 60 % of
                                        * the nodes (short-range spheroidal kernels) the default stages run and
                                        * bbfmm_tree_stats.m2l_basis_len stays 0. */
#define BBFMM_FLAG_DIRECT_SMALL_W_LEAVES 4u /* EXTENSION beyond the reference (off by default): a W-list cell that is a leaf
                                        * with no more points than the expansion has nodes is treated as near field --
                                        * its points are summed directly (both ways: the transposed X-list entry goes
                                        * too) instead of through M2P / P2L, which cost `nodes` kernel evaluations per
                                        * target where the direct sum costs `points`.  Exact where the reference
                                        * approximates, so results move by the reference's own M2P / P2L error (about
                                        * epsilon); on mixed-level trees of moderate size these two passes dominate the
                                        * matvec (1M uniform points, Spheroidal3: 9 of 14 ms). */
#define BBFMM_FLAG_DETERMINISTIC 8u /* bitwise reproducible results from run to run, as the reference's fixed-order
                                    * per-target sums are.  The default one-rhs matvec accumulates through hardware f64
                                    * atomics in three places (column sums of the unordered-pair near field, L and the
                                    * potentials of the fused M2P + P2L pass, the contraction split of M2L stage 2 on
                                    * small trees) and M2P adds the chunks of a leaf's W list atomically: the last bits
                                    * of a result depend on the scheduling (differences <= 1e-12 relative; an FGMRES run
                                    * near its tolerance may take one iteration more or less).  With this flag the handle
                                    * uses the ordered-pair kernels, one M2P job per leaf and plain stores in stage 2:
                                    * every sum has a fixed order (about 1.1x slower at 10M points, 1.5x on mixed-level
                                    * trees).  Same arithmetic as the reference either way. */

/*
 * FmmTree::new (ferreus_rbf_utils/src/utils.rs:392-421 -> ferreus_bbfmm/src/bbfmm.rs:272-353).
 * Copies the points (the caller keeps its buffer), builds the Morton tree, the
 * U/V/W/X lists and the M2L operators on the host and uploads them.
 *   pts       N x d column-major, leading dimension ld (>= n)
 *   extents   NULL (computed from the data, bbfmm.rs:281-284) or 2*d values
 *             [mins..., maxs...]
 *   params    NULL -> FmmParams::new_defaults(order)
 * d must be 1, 2 or 3 (bbfmm.rs:293-298).
 */
int bbfmm_create(const double *pts, int64_t n, int32_t d, int64_t ld, int32_t interpolation_order,
                 int32_t kernel_type, double base_range, double total_sill, int32_t adaptive_tree,
                 int32_t sparse, const double *extents, const bbfmm_params *params,
                 uint32_t flags, bbfmm_handle **out);

/*
 * One handle, several devices, ONE process (round 6).  The reference keeps a single FmmTree behind a Mutex
 * (ferreus_rbf/src/rbf.rs:85-133) and its FGMRES is not an SPMD program, so the GPUs of a node are reached behind the
 * unchanged method set of ferreus_rbf_utils::FmmTree (utils.rs:392-449): a handle created on a device list owns one
 * part per list entry -- the tree on that device and one subtree partition of the matvec -- and
 * bbfmm_set_weights + bbfmm_evaluate at the sources, bbfmm_fast_matrix_vector_product (all rows) and
 * bbfmm_matvec_device run partitioned over the parts: weights to every device over its own link, own-subtree upward
 * pass, the partial coarse multipoles copied to a slot on every device (peer copies beside the near field) and added in
 * part order, restricted downward + leaf pass, the owned blocks of the potentials straight back to the host (or, for
 * device callers, to the first device).  No caller-supplied collective, no second process.  Arbitrary targets (values,
 * gradients, Leaves mode after bbfmm_set_local_coefficients) with the weights of bbfmm_set_weights are SHARDED by target rows
 * when there are at least 16384 per part (BBFMM_GROUP_SHARD_MIN): every part completes its own multipoles from its
 * device's copy of the weights and evaluates a contiguous share of the rows (bbfmm_last_evaluate_at_sources: 3).  Row subsets
 * (bbfmm_fast_matrix_vector_product with target_indices; the unchanged caller's evaluate at rows of the sources from the second
 * sighting of a set on) are dealt to the parts that own the rows.  Everything else (few targets, other weights than
 * set_weights') is served by the first device alone, with unchanged results.
 *   devices    n_devices HIP device ids; the first one holds the handle's own tree (introspection, bbfmm_stream,
 *              device-resident vectors).  An id may repeat: logical parts on one device -- the one-GPU rehearsal of the
 *              N-device path (peer copies become device copies).  One entry: a plain handle on that device.
 * bbfmm_create itself reads FERREUS_BBFMM_DEVICES ("0,1,2,3", "all", "0,0,0"; unset or empty: the current device, no
 * group) -- the reference's constructor has no argument for a device list, so the unchanged caller is switched there.
 * The calling thread's current device is restored before returning.
 */
int bbfmm_create_on_devices(const double *pts, int64_t n, int32_t d, int64_t ld, int32_t interpolation_order,
                            int32_t kernel_type, double base_range, double total_sill, int32_t adaptive_tree,
                            int32_t sparse, const double *extents, const bbfmm_params *params, uint32_t flags,
                            const int32_t *devices, int32_t n_devices, bbfmm_handle **out);
int32_t bbfmm_device_count(const bbfmm_handle *h);              /* parts of the handle (1: no group) */
int32_t bbfmm_part_device(const bbfmm_handle *h, int32_t part); /* HIP device of a part; part 0 = the handle's own device */
int bbfmm_group_bounds(const bbfmm_handle *h, int64_t *bounds_out); /* parts + 1 offsets into the sorted points (group handles) */
int bbfmm_get_part_phase_ms(bbfmm_handle *h, int32_t part, double *ms_out, int64_t *count_out); /* bbfmm_get_phase_ms of one part */

void bbfmm_destroy(bbfmm_handle *h);

/* Message of the last failure on this handle ("" if none). Owned by the handle. */
const char *bbfmm_last_error(const bbfmm_handle *h);

/* FmmTree::set_weights (utils.rs:425-429 -> bbfmm.rs:383-401): upward pass.
 * w is rows x k (rows >= N; only rows < N are read), leading dimension ldw. */
int bbfmm_set_weights(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw);

/* FmmTree::set_local_coefficients (utils.rs:433-437 -> bbfmm.rs:518-524). */
int bbfmm_set_local_coefficients(bbfmm_handle *h, const double *w, int64_t rows, int32_t k,
                                 int64_t ldw);

/* FmmTree::evaluate (utils.rs:441-449 -> bbfmm.rs:411-418): downward pass restricted
 * to the ancestors of the target leaves + leaf pass.  x is m x d (ldx), out is
 * m x k (ldo), caller allocated.  On BBFMM_POINT_OUTSIDE_TREE *bad_point_index
 * receives the smallest offending row (linear_tree.rs:505-517). */
int bbfmm_evaluate(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw,
                   const double *x, int64_t m, int64_t ldx, double *out, int64_t ldo,
                   int64_t *bad_point_index);
/* The unchanged caller of the FGMRES matvec (rbf.rs:1357-1364) calls set_weights(w) and then
 * evaluate(w, select_mat_rows(source_points, all rows)).  bbfmm_evaluate recognises that sequence without being
 * told: m == N targets that equal the handle's source points bit for bit and row for row (threaded comparison on
 * the host, run beside the M2L) are served by the resident sorted target set -- every unordered near-field pair
 * once, M2P fused with P2L, no target upload or grouping -- and weights equal to those of the preceding
 * bbfmm_set_weights are not transferred a second time.  One differing bit (a perturbed coordinate, two rows
 * swapped, -0.0 for 0.0) takes the general path.  Results of the two paths agree to summation order (1e-12).
 * The same caller's matvec_partial (rbf.rs:119-133: target_indices = Some(idx)) evaluates at
 * select_mat_rows(source_points, idx): on a handle created the way the solver creates its tree (sparse, extents NULL:
 * rbf.rs:456-467), targets that are ROWS of the sources (one rhs, N / 2048 <= m <= N / 2 rows, each
 * found bit for bit in a table over the source points that the first such call builds) are served by the cached plan
 * bbfmm_fast_matrix_vector_product(target_indices) uses -- sorted targets and restricted downward pass once per index
 * set, not once per call.  One target that is no source point takes the general path.
 * Returns which path the last bbfmm_evaluate on this handle took: 1 the resident sources, 2 the cached plan of a row
 * subset, 0 the general path.  BBFMM_EVAL_SOURCES_FAST=0 in the environment disables both detections (checker). */
int bbfmm_last_evaluate_at_sources(const bbfmm_handle *h);
/* The comparison itself (host only, also on BBFMM_FLAG_HOST_ONLY handles): 1 when x (m x d, ldx) equals the handle's
 * source points bit for bit and row for row, else 0. */
int bbfmm_debug_targets_are_sources(const bbfmm_handle *h, const double *x, int64_t m, int64_t ldx);
/* ... and the lookup of targets that are rows of the sources: 1 and rows_out[j] = a source row with the coordinates of
 * target j (rows with equal coordinates are interchangeable) when every target is a source point, else 0. */
int bbfmm_debug_rows_of_sources(bbfmm_handle *h, const double *x, int64_t m, int64_t ldx, int64_t *rows_out);

/* FmmTree::evaluate_with_gradients (utils.rs:453-461 -> bbfmm.rs:434-441).
 * grad is m x (k*d), columns [rhs0_dx, rhs0_dy, rhs0_dz, rhs1_dx, ...]. */
int bbfmm_evaluate_with_gradients(bbfmm_handle *h, const double *w, int64_t rows, int32_t k,
                                  int64_t ldw, const double *x, int64_t m, int64_t ldx,
                                  double *out, int64_t ldo, double *grad, int64_t ldg,
                                  int64_t *bad_point_index);

/* FmmTree::evaluate_leaves (utils.rs:465-473 -> bbfmm.rs:537-544): leaf pass only,
 * after bbfmm_set_local_coefficients.
 * Extension for many small batches (isosurfacing, ferreus_rmt/src/isosurface.rs:574,693): w may be
 * NULL in the two leaves-only entry points; the weights already resident on the device (those of
 * the last call that took weights, normally bbfmm_set_local_coefficients) are used and the N x k
 * host-to-device copy per call is skipped.  k must still equal the set_weights column count. */
int bbfmm_evaluate_leaves(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw,
                          const double *x, int64_t m, int64_t ldx, double *out, int64_t ldo,
                          int64_t *bad_point_index);

/* FmmTree::evaluate_leaves_with_gradients (utils.rs:477-485 -> bbfmm.rs:560-567). */
int bbfmm_evaluate_leaves_with_gradients(bbfmm_handle *h, const double *w, int64_t rows, int32_t k,
                                         int64_t ldw, const double *x, int64_t m, int64_t ldx,
                                         double *out, int64_t ldo, double *grad, int64_t ldg,
                                         int64_t *bad_point_index);

/* FmmTree::source_points (utils.rs:489-493): copies the N x d points to out (ld). */
int bbfmm_source_points(const bbfmm_handle *h, double *out, int64_t ld);

/*
 * The FGMRES matvec, ferreus_rbf/src/rbf.rs:1338-1379 (fast_matrix_vector_product):
 *   set_weights(w); y = evaluate(w, source_points[target_indices])
 *   result[i] = y + nugget*w[i] + P[i,:] * w[N..N+basis_size]   for i in target_indices,
 *   every other row of result (incl. the last basis_size rows) = 0.
 * w and result have N + basis_size rows, one column.  target_indices NULL -> all
 * N sources.  poly is N x basis_size column-major (ldp) or NULL.
 * Host buffers; the targets never leave the device (they are the sources).
 */
int bbfmm_fast_matrix_vector_product(bbfmm_handle *h, const double *w, int64_t rows,
                                     int64_t basis_size, const int64_t *target_indices,
                                     int64_t n_target_indices, const double *poly, int64_t ldp,
                                     double nugget, double *result);

/* Builds (and caches) what a later bbfmm_fast_matrix_vector_product with these target_indices needs -- the
 * sorted targets and the restricted downward pass -- without running a product; also allocates the
 * pinned staging buffer.  Optional: the first product with an index set does the same.  The Schwarz
 * preconditioner calls it for its levels at creation, so that setup cost does not land in the solve. */
int bbfmm_prepare_target_subset(bbfmm_handle *h, const int64_t *target_indices, int64_t n_target_indices);

/*
 * Device-resident form of the same product for the case target_indices = all,
 * basis_size = 0, nugget = 0 generalised to k right-hand sides:
 *   d_out[:, j] = K(X, X) d_w[:, j]      (set_weights + evaluate at the sources)
 * d_w / d_out are DEVICE pointers (N x k, ldw / ldo).  Asynchronous on the
 * handle's stream unless sync != 0.  This is what bench.py times.
 * Reproducibility: for ANY k (since round 4 also k > 1, and on partitioned handles) the near field runs the
 * unordered-pair kernels and M2P is fused with P2L; both accumulate with f64 atomics, so two runs agree to summation
 * order (<= 1e-12 relative), not bit for bit.  Handles created with BBFMM_FLAG_DETERMINISTIC (or a process with
 * BBFMM_P2P_SYM=0 in its environment: ordered pairs, separate P2L and M2P) keep a fixed order.  The same holds for
 * bbfmm_fast_matrix_vector_product, for bbfmm_evaluate on targets that are the sources, and for the
 * bbfmm_matvec_partition_* calls.
 */
int bbfmm_matvec_device(bbfmm_handle *h, const double *d_w, int64_t ldw, int32_t k, double *d_out,
                        int64_t ldo, int32_t sync);

/* IterativeSolver::matvec_partial (rbf.rs:119-133) on device-resident vectors, for callers that keep
 * their vectors in HBM between products (the Schwarz sweep, bbfmm_schwarz_apply).  An index set is
 * registered once -- sorted targets and restricted downward pass are built then and kept for the life
 * of the handle -- and named by the returned id afterwards (-1 is returned for "all rows in order").
 *   d_y[j] = sum_i phi(x_target_indices[j], x_i) * d_w[i],   j < n_target_indices
 * d_w: N values in the caller's point numbering, d_y: n_target_indices values (N for id -1), both DEVICE
 * pointers.  The nugget and polynomial terms of rbf.rs:1366-1376 are the caller's (they are O(n)).
 * Asynchronous on the handle's stream unless sync != 0. */
int bbfmm_target_subset_create(bbfmm_handle *h, const int64_t *target_indices, int64_t n_target_indices,
                               int32_t *subset_id);
int bbfmm_matvec_subset_device(bbfmm_handle *h, int32_t subset_id, const double *d_w, double *d_y, int32_t sync);

/* HIP stream the handle launches on (hipStream_t as void*), for event timing and for device work that has to
 * stay in order with the handle's.  A handle belongs to the HIP device that was current in bbfmm_create; every
 * entry point that takes the handle (this one included) makes that device current in the calling thread, so the
 * handle can be used from threads that never selected a device. */
void *bbfmm_stream(bbfmm_handle *h);

/* Multi-GPU target partition: restrict the downward + leaf pass of
 * bbfmm_matvec_device to the contiguous Morton range of target leaves owned by
 * `rank` of `world`; rows of d_out that this rank does not own are left
 * untouched.  The owned rows, in original point numbering, are reported by
 * bbfmm_partition_rows.  rank=0, world=1 restores the full evaluation. */
int bbfmm_set_partition(bbfmm_handle *h, int32_t rank, int32_t world);
int64_t bbfmm_partition_row_count(const bbfmm_handle *h);
int bbfmm_partition_rows(const bbfmm_handle *h, int64_t *rows_out);

/* The partitioned matvec in two calls, around the one exchange the upward pass needs (the passes
 * being split are ferreus_bbfmm/src/bbfmm.rs:666-772 and 444-507).  Every rank holds all weights but anterpolates
 * only its share: above a coarse level (4 for a uniform 10M-point tree) the cells it owns and the three-cell halo
 * its V / W lists read, complete; at and below it the partial sums over the sources it owns.  M2M is linear, so
 * one all-reduce (sum) of those partial coarse multipoles -- a contiguous prefix of M, 13 MB per right-hand side at
 * order 7 -- completes them on every rank.
 *   bbfmm_partition_coarse_count   doubles per right-hand side of that prefix (0: nothing to exchange, e.g. depth < 2)
 *   bbfmm_matvec_partition_upward  gather + this rank's P2M / M2M on the handle's stream; packs k x count partial
 *                                  multipoles rhs-major into d_coarse (device memory of the caller)
 *   -- caller: all-reduce (sum) d_coarse over the ranks, ordered after the handle's stream (RCCL on that stream, or
 *      any stream that waits for it) --
 *   bbfmm_matvec_partition_finish  takes the summed d_coarse, runs the downward and leaf passes of the owned
 *                                  targets, writes the owned rows of d_out (ld ldo) like bbfmm_matvec_device.
 * comm_stream (hipStream_t as void*, may be NULL): the stream the caller issues the all-reduce on.  _upward makes it wait
 * for the packed multipoles only and queues the near field (P2P) of the owned targets behind the pack on the handle's own
 * stream, so that the collective runs beside it; _finish makes the handle's stream wait for comm_stream before it reads
 * d_coarse.  With NULL the caller orders the collective after the handle's stream itself (everything serial).
 * bbfmm_matvec_device on a partitioned handle still works on its own (it then runs the whole upward pass).  After
 * _upward the handle's multipoles are this rank's share: bbfmm_evaluate* and bbfmm_set_local_coefficients return
 * BBFMM_BAD_ARGUMENT ("one partition's share") until bbfmm_set_weights or bbfmm_matvec_device has run a whole upward pass
 * (a device-group handle completes its first part's multipoles by itself). */
int64_t bbfmm_partition_coarse_count(const bbfmm_handle *h);
int bbfmm_matvec_partition_upward(bbfmm_handle *h, const double *d_w, int64_t ldw, int32_t k, double *d_coarse, void *comm_stream);
int bbfmm_matvec_partition_finish(bbfmm_handle *h, const double *d_coarse, double *d_out, int64_t ldo, int32_t sync, void *comm_stream);
/* The exchange of the owned potentials in the tree's sorted order (what distributed.py's PartitionedMatvec uses): a part
 * owns ONE range of the sorted points, so its potentials travel as one contiguous block and every rank writes the gathered
 * blocks to their rows in a single pass over the tree's permutation -- no index arrays on the caller's side.
 *   bbfmm_partition_world / _rank         parts of the handle's partition (1: none set) and the part it owns
 *   bbfmm_partition_bounds                world + 1 offsets: part r owns the sorted points bounds[r] .. bounds[r + 1)
 *                                         (the same on every rank; its own count is bounds[rank + 1] - bounds[rank]);
 *                                         `world` must be bbfmm_partition_world
 *   bbfmm_matvec_partition_finish_sorted  like _finish, but leaves the k x count owned potentials in d_seg (row stride
 *                                         ld >= count) instead of scattering them -- the block to all-gather
 *   bbfmm_partition_scatter               d_all[n_parts][k][m_max] = the gathered blocks of parts first_part ..
 *                                         first_part + n_parts (normally 0, world) -> their rows of d_out (k x N, row
 *                                         stride ldo); asynchronous on the handle's stream like the other calls */
int32_t bbfmm_partition_world(const bbfmm_handle *h);
int32_t bbfmm_partition_rank(const bbfmm_handle *h);
int bbfmm_partition_bounds(const bbfmm_handle *h, int32_t world, int64_t *bounds_out);
int bbfmm_matvec_partition_finish_sorted(bbfmm_handle *h, const double *d_coarse, double *d_seg, int64_t ld, void *comm_stream);
int bbfmm_partition_scatter(bbfmm_handle *h, const double *d_all, int32_t first_part, int32_t n_parts, int64_t m_max, int32_t k,
                            double *d_out, int64_t ldo);
/* Test hook (host side, also on BBFMM_FLAG_HOST_ONLY handles): walks this rank's upward plan with point COUNTS in
 * place of multipoles (P2M -> points of the leaf, M2M -> sum over the plan's children).  counts_out[c] (n_cells):
 * what the plan leaves in cell c before the exchange (-1: never written); reads_out[c] = 1 where the rank's downward
 * or leaf pass reads M_c.  info_out[0..3] = coarse level, coarse cells, leaves anterpolated, parents translated.
 * Summed over the ranks the coarse prefix must equal the true point counts, and every cell a rank reads above the
 * coarse level must hold its true count already. */
int bbfmm_debug_partition_upward_counts(const bbfmm_handle *h, int64_t *counts_out, uint8_t *reads_out, int64_t *info_out);

/* Host-side legs of the host-buffer entry points by themselves (scripts/host_buffer_legs.py; no device needed): n doubles
 * copied by the library's thread pool in 2 MB pieces, gathered through a random permutation, scattered through it.
 * out4 = {copy ms, gather ms, scatter ms, pool threads}. */
int bbfmm_debug_host_copy_rates(int64_t n, double *out4);

/* ---- introspection (tests, bench statistics; host side, no device needed) ---- */
typedef struct {
    int32_t d, order, n_nodes, depth;
    int64_t n_points, n_cells, n_leaves;
    int64_t n_u, n_v, n_w, n_x;              /* total list entries */
    int64_t p2p_pairs;                       /* sum_leaf n_t * sum_{U} n_s, targets = sources */
    int64_t p2p_tile_bytes_k1;               /* SURVEY.md 8(d) tile-traffic bytes at K=1 */
    double m2l_flops_k1;                     /* sum_pairs 4*n*r (2*n*n if uncompressed); shared basis: see m2l_basis_rank */
    double center[3];
    double radius;
    int64_t wx_pairs;                        /* sum over (leaf, W cell) of n_t * n: kernel evaluations of M2P (= of P2L) */
    int64_t wx_tile_bytes_k1;                /* tile traffic of M2P + P2L at K=1: per (leaf, W cell) n_t*(8d+8) + 2*n*8 */
    int32_t m2l_basis_rank;                  /* BBFMM_FLAG_M2L_SHARED_BASIS: largest rank of a level's basis (0: flag not set); */
    int32_t m2l_basis_len;                   /* coordinates kept per cell (rank padded to the kernel's column groups);      */
                                             /* m2l_flops_k1 then counts the stages in the basis + both changes of basis      */
    /* The intermediate of the two M2L stages (one slot of sum_t rank_t doubles per target cell; the reference holds
     * none, bbfmm.rs:864-986 multiplies pair by pair) is bounded: the levels -- or, for a level that alone exceeds the
     * budget, 2 / 4 / 8 groups of its target classes -- go through one buffer in `m2l_batches` passes, a few
     * right-hand sides at a time.  Budget: BBFMM_M2L_CBUF_MB (default: a sixteenth of the device's memory, 18 GiB on
     * MI355X, at least 4096). */
    int32_t m2l_batches;                     /* passes through the buffer per right-hand-side chunk                        */
    int32_t m2l_rhs_per_pass;                /* right-hand sides per pass (0 until weights were set)                        */
    int64_t m2l_slots_bytes_per_rhs;         /* all slots of one right-hand side (what an unbounded buffer would hold)      */
    int64_t m2l_intermediate_bytes;          /* bytes of the buffer actually allocated (0 until weights were set)           */
} bbfmm_tree_stats;

int bbfmm_get_tree_stats(const bbfmm_handle *h, bbfmm_tree_stats *out);
/* 1 when the source tree (linear_tree.rs:20-175: Morton codes, sort, subdivision, per-leaf point lists) was built
 * on the device, 0 when the host build ran (BBFMM_FLAG_HOST_ONLY, BBFMM_TREE_DEVICE=0, or a source point outside
 * the root box).  Both produce the same tree; tests compare them. */
int bbfmm_tree_built_on_device(const bbfmm_handle *h);

/* Cells in (level, key) order: Morton key (morton.rs:58-119), leaf flag. */
int bbfmm_get_cells(const bbfmm_handle *h, uint64_t *keys, uint8_t *is_leaf);
/* Source rows of every leaf: ptr has n_cells+1 entries, idx has n_points entries
 * (ascending inside a leaf, linear_tree.rs:55-66). */
int bbfmm_get_leaf_sources(const bbfmm_handle *h, int64_t *ptr, int64_t *idx);
/* which: 'U','V','W','X'.  ptr may be NULL to query the size; returns entries via *n_entries.
 * idx holds cell indices (positions in bbfmm_get_cells), sorted by key. */
int bbfmm_get_list(const bbfmm_handle *h, char which, int64_t *ptr, int32_t *idx,
                   int64_t *n_entries);
/* M2L operator ranks per (level, reference vector): ranks[level*n_ref + ref], levels
 * 0..depth (0 where no operator exists).  n_ref_out: 2/7/16 for d = 1/2/3. */
int bbfmm_get_m2l_ranks(const bbfmm_handle *h, int32_t *ranks, int32_t *n_ref_out);
/* Dense reconstruction U*Vt (n x n column-major) of one reference operator. */
int bbfmm_get_m2l_operator(const bbfmm_handle *h, int32_t level, int32_t ref, double *out);
/* The factors themselves: u is n x rank, vt is rank x n (both column-major; vt may be NULL,
 * and is not written for uncompressed operators where U is the n x n kernel block). */
int bbfmm_get_m2l_factors(const bbfmm_handle *h, int32_t level, int32_t ref, double *u, double *vt);
/* Symmetry tables (chebyshev.rs:486-585): perm / invperm are n_perm x n (row-major),
 * perm_lookup / ref_lookup have 7^d entries.  Any pointer may be NULL. */
int bbfmm_get_permutation_tables(const bbfmm_handle *h, int32_t *n_perm, int32_t *perm,
                                 int32_t *invperm, int32_t *perm_lookup, int32_t *ref_lookup);
/* Target -> leaf assignment (linear_tree.rs:487-520) for arbitrary points; cell_out
 * receives cell indices. */
int bbfmm_points_to_leaves(const bbfmm_handle *h, const double *x, int64_t m, int64_t ldx,
                           int32_t *cell_out, int64_t *bad_point_index);

/* Per-phase device time in milliseconds, accumulated since the last reset while profiling is
 * enabled.  hipEvent pairs are recorded on the handle's stream around each phase WITHOUT
 * synchronising (the timed kernels still run back to back); this call synchronises once and
 * resolves them.  count_out (optional) receives the number of recorded intervals per phase.
 * Order: gather, P2M, M2M, M2L_stage1, M2L_stage2, P2L, L2L, P2P, M2P, L2P, scatter. */
#define BBFMM_N_PHASES 11
int bbfmm_set_profiling(bbfmm_handle *h, int32_t enable);
int bbfmm_get_phase_ms(bbfmm_handle *h, double *ms_out, int64_t *count_out);
int bbfmm_reset_phase_ms(bbfmm_handle *h);

/* FP64 MFMA self-test + peak microbenchmark (device): returns measured TFLOP/s of
 * back-to-back v_mfma_f64_16x16x4 in *tflops and 0 mismatches in *layout_errors when
 * the lane layout assumed by the M2L kernels holds on this device.  info6 (optional, 6 doubles):
 * cycles/MFMA of a lone wave, its clock (MHz), cycles/MFMA/SIMD and clock with every CU busy at
 * 1 wave/SIMD, TFLOP/s at 1 and at 2 waves/SIMD. */
int bbfmm_mfma_f64_selftest(double *tflops, int32_t *layout_errors, double *info6);
/* FP64 vector-ALU microbenchmark (device): chip-wide v_fma_f64 rate in *tflops and the shader clock it ran at in
 * *clock_mhz -- the issue roofline of the pair kernels (P2P, M2P, P2L), which the FP64 load pulls under the nominal
 * 2.4 GHz. */
int bbfmm_fp64_valu_selftest(double *tflops, double *clock_mhz);

/* ---- test hooks (host loops, no device; never reached from a compute entry point) ----
 * Dense n x n (column-major) M2M matrix of child `child_index` exactly as the reference
 * stores it (chebyshev.rs:216-240); the device applies the same operator sum-factorised. */
int bbfmm_debug_dense_m2m(const bbfmm_handle *h, int32_t child_index, double *out);
/* Applies the stacked per-class M2L tables (what the MFMA kernels consume) with plain host
 * loops: L[c][:] += M2L(M) for one rhs, M and L being n_cells x n cell-major.  Only valid on
 * BBFMM_FLAG_HOST_ONLY handles (the tables are released after upload otherwise). */
int bbfmm_debug_apply_m2l_tables_host(const bbfmm_handle *h, const double *M, double *L);
/* Number of stage-1 boundary variants (stacked operators with the transfer vectors towards missing targets left
 * out, fmm_m2l_tables.cpp build_m2l_tables) and of source cells that use one. */
int bbfmm_debug_m2l_variants(const bbfmm_handle *h, int64_t *n_variants, int64_t *n_cells);

/* The Morton primitives of csrc/morton.hpp as the host tree build uses them (morton.rs:58-263; the
 * reference's byte lookup tables, morton_constants.rs:77-346, are replaced by bit arithmetic): checked
 * bit-exactly against the reference's own tables in tests/test_reference_tables.py.
 *   encode: anchor[d] (16 bits per axis are read) + level -> key;   decode: key -> anchor[d], level
 *   neighbours: same-level neighbour keys inside the root box, in the reference's direction order
 *               (morton_constants.rs:32-74); keys_out holds up to 26, the count is returned
 *   direction_vectors: out is n x d (row-major), n = 2 / 8 / 26 returned
 *   reference_vectors: the handle's M2L reference vectors (chebyshev.rs:272-294), n_ref x d row-major */
uint64_t bbfmm_debug_morton_encode(int32_t d, const uint64_t *anchor, uint64_t level);
void bbfmm_debug_morton_decode(int32_t d, uint64_t key, uint64_t *anchor_out, uint64_t *level_out);
int32_t bbfmm_debug_morton_neighbours(int32_t d, uint64_t key, uint64_t *keys_out);
int32_t bbfmm_debug_direction_vectors(int32_t d, int32_t *out);
int bbfmm_debug_reference_vectors(const bbfmm_handle *h, int32_t *out, int32_t *n_ref_out);

/* Copies the device-resident multipole ('M') or local ('L') coefficients of the last pass to
 * the host as k x n_cells x n (rhs-major, cell-major), i.e. column c + j*n_cells of the
 * reference's n x (C*K) matrices (bbfmm.rs:234-242).  Per-phase parity checks. */
int bbfmm_debug_get_coefficients(bbfmm_handle *h, char which, int32_t k, double *out);

/* ------------------------------------------------------------------ iterative solvers
 * ferreus_rbf/src/iterative_solvers.rs: host FGMRES / stationary Schwarz
 * drivers around operator callbacks, so the reference's closures (`matvec`, `precon`,
 * rbf.rs:523-524) plug in unchanged.  Vectors are host arrays of n doubles. */

/* y = Op(x); return BBFMM_OK or an error code (which the solver returns as is). */
typedef int (*bbfmm_apply_fn)(void *user, const double *x, double *y, int64_t n);
/* ProgressMsg::SolverIteration {iter, residual, progress} (iterative_solvers.rs:142-148) */
typedef void (*bbfmm_iteration_fn)(void *user, int64_t iter, double residual, double progress);

/* FittingAccuracyType (interpolant_config.rs:54-63) */
enum { BBFMM_ACCURACY_ABSOLUTE = 0, BBFMM_ACCURACY_RELATIVE = 1 };

/* givens_rotation (iterative_solvers.rs:185-227), a port of LAPACK dlartg:
 * [c s; -s c] [f; g] = [r; 0]. */
void bbfmm_givens_rotation(double f, double g, double *c, double *s, double *r);

/*
 * fgmres (iterative_solvers.rs:38-172): restarted flexible GMRES, right preconditioner m
 * (NULL: none), initial guess x0 (NULL: zero), max_outer_iterations restarts of
 * max_inner_iterations Krylov vectors (the solver uses 20 x 5, rbf.rs:545-554), modified
 * Gram-Schmidt, Givens rotations.  Stopping: Absolute -> |g[j+1]| (max-norm of the true residual
 * at restarts) < tolerance; Relative -> the same over the initial 2-norm.  callback (may be NULL)
 * receives every inner iteration.  Outputs: x (n), the number of inner iterations done and the
 * last residual measure (both optional).
 * Deviation: an exactly zero residual returns x instead of dividing by zero.
 */
int bbfmm_fgmres(int64_t n, bbfmm_apply_fn a, void *a_user, const double *b, bbfmm_apply_fn m, void *m_user,
                 const double *x0, int32_t max_outer_iterations, int32_t max_inner_iterations,
                 int32_t tolerance_type, double tolerance, bbfmm_iteration_fn callback, void *cb_user,
                 double *x, int64_t *iterations, double *final_residual);

/* schwarz_ddm_solver (iterative_solvers.rs:229-281): s += M(r); r = rhs - A s, at most
 * max_iterations times (the solver uses 100, rbf.rs:555-562).  m NULL -> zero vector, as the
 * reference. */
int bbfmm_schwarz_ddm_solver(int64_t n, bbfmm_apply_fn matvec, void *a_user, const double *rhs,
                             bbfmm_apply_fn m, void *m_user, int32_t max_iterations, int32_t tolerance_type,
                             double tolerance, bbfmm_iteration_fn callback, void *cb_user, double *x,
                             int64_t *iterations, double *final_residual);

/* IterativeSolver::matvec (rbf.rs:105-117) as an operator callback: pass a bbfmm_rbf_system as
 * `user`; n must be N + basis_size.  Runs bbfmm_fast_matrix_vector_product on all sources. */
typedef struct bbfmm_rbf_system {
    bbfmm_handle *tree;
    int64_t basis_size;            /* InterpolantSettings::basis_size */
    const double *monomial_matrix; /* N x basis_size column-major, or NULL */
    int64_t ld_monomial;
    double nugget;
} bbfmm_rbf_system;
int bbfmm_rbf_system_apply(void *user, const double *x, double *y, int64_t n);

/* ------------------------------------------------------------------ domain decomposition (host part)
 * DDMTree::new (ferreus_rbf/src/preconditioning/domain_decomposition.rs:67-347), SURVEY.md 8(f)-1:
 * the multi-level overlapping decomposition of the Schwarz preconditioner -- which points form which
 * leaf domain on which level.  The local factorisations (domain.rs) and the apply (schwarz.rs) on these
 * index sets are the bbfmm_schwarz_* entry points further down. */
typedef struct bbfmm_ddm bbfmm_ddm;
typedef struct bbfmm_ddm_params { /* DDMParams, config.rs:42-69 */
    int64_t leaf_threshold;   /* 1024 */
    double overlap_quota;     /* 0.5 */
    double coarse_ratio;      /* 0.125 */
    int64_t coarse_threshold; /* 4096 */
} bbfmm_ddm_params;
void bbfmm_ddm_params_defaults(bbfmm_ddm_params *out);
/* Extension (not in the reference): the defaults with coarse_threshold raised so that the hierarchy over
 * n points keeps at most three fine levels, coarse_threshold = max(4096, n/470 + 1): a level keeps
 * ceil(ceil(N/8)/leaves) points per leaf, at most N (1/8 + 1/341), so three levels leave at most n/478.  With the plain
 * defaults a fourth fine level appears above about 2.1M points and leaves one coarse point per ~7 level-0
 * domains, where the sweep of schwarz.rs stalls for the thin-plate spline. */
void bbfmm_ddm_params_for_points(int64_t n, bbfmm_ddm_params *out);
/* points: n x d column-major (ld); params NULL -> defaults */
int bbfmm_ddm_build(const double *points, int64_t n, int32_t d, int64_t ld, const bbfmm_ddm_params *params,
                    bbfmm_ddm **out);
void bbfmm_ddm_destroy(bbfmm_ddm *t);
int32_t bbfmm_ddm_num_levels(const bbfmm_ddm *t);                 /* finest first, coarse domain last */
int64_t bbfmm_ddm_level_size(const bbfmm_ddm *t, int32_t level);  /* Level::point_indices.len() */
int bbfmm_ddm_level_points(const bbfmm_ddm *t, int32_t level, int64_t *out);
int64_t bbfmm_ddm_num_domains(const bbfmm_ddm *t, int32_t level); /* Level::leaf_domains.len() */
int64_t bbfmm_ddm_domain_size(const bbfmm_ddm *t, int32_t level, int64_t domain);
/* Domain::{overlapping_point_indices, internal_points_mask, extents [mins..., maxs...]} */
int bbfmm_ddm_domain(const bbfmm_ddm *t, int32_t level, int64_t domain, int64_t *indices, uint8_t *internal,
                     double *extents);

/* ------------------------------------------------------------------ Schwarz preconditioner
 * schwarz_preconditioner (ferreus_rbf/src/preconditioning/schwarz.rs:32-155) with Domain::factorise /
 * Domain::solve (domain.rs:153-475) for all leaf domains of a level batched on the device: Beatson's Q
 * formulation, Q^T A Q assembled from the points, blocked Cholesky and substitutions one workgroup
 * per domain.  The two partial matvecs per fine level go through `tree`
 * (IterativeSolver::precon, rbf.rs:140-155).  Global trend transforms are not supported.
 * A domain whose Cholesky factorisation fails is solved through a host-computed inverse instead (the
 * role of the reference's LBL^T fallback, domain.rs:60-68); the one large coarse domain (more than 2048
 * points) is then factorised by pivoted LU on the device (rocSOLVER, loaded on demand).  BBFMM_UNSUPPORTED
 * is returned only when a local system is singular or that library is missing. */
typedef struct bbfmm_schwarz bbfmm_schwarz;
typedef struct bbfmm_interpolant { /* InterpolantSettings, interpolant_config.rs:118-147, as the solver reads it */
    int32_t kernel_type;       /* bbfmm_kernel_type 0..6 (Linear, ThinPlateSpline, Cubic, Spheroidal3/5/7/9) */
    int32_t polynomial_degree; /* -1 none, 0 constant, 1 linear, 2 quadratic (Drift) */
    double nugget, base_range, total_sill;
    uint32_t flags;            /* 0, or BBFMM_FLAG_GLOBAL_SCALING */
} bbfmm_interpolant;
/* Default (0): the coarse domain scales its monomials by the extents of its own points, as the
 * reference does (domain.rs:171-172).  With this flag it uses the extents of all points, the basis
 * of the system's monomial matrix, which makes its polynomial tail exact (see csrc/schwarz.cpp). */
#define BBFMM_FLAG_GLOBAL_SCALING 1u
/* tree: a handle over the same points, kernel and ranges (it serves matvec_partial); it must outlive
 * the preconditioner.  params NULL -> DDMParams defaults. */
int bbfmm_schwarz_create(bbfmm_handle *tree, const double *points, int64_t n, int32_t d, int64_t ld,
                         const bbfmm_interpolant *settings, const bbfmm_ddm_params *params, bbfmm_schwarz **out);
/* The same preconditioner with its FACTORS SHARDED over the ranks of a job (level 0 of 10M points holds
 * 95 GB of Cholesky factors, one GPU's 288 GB end near 28M points).  No counterpart in the reference (one address space).
 * Every rank decomposes the same points the same way, keeps the contiguous share [nd * rank / world, nd * (rank + 1) /
 * world) of every fine level's domains -- factorises and solves only those -- and after a level's local solves the
 * corrections of all ranks are summed: the internal points of the domains partition the level's rows (restricted
 * additive Schwarz, schwarz.rs:96-113), so each row is written by exactly one rank and the sum adds zeros (the result
 * equals the unsharded preconditioner's bit for bit).  The coarse domain (one domain) and the partial products through
 * `tree` are replicated.
 *   d_exchange: DEVICE buffer of the caller, exchange_capacity >= the largest fine level (N) doubles;
 *   allreduce(user, count): sum d_exchange[0 .. count) over the ranks in place (ncclAllReduce / torch.distributed);
 *     called from bbfmm_schwarz_apply with the library's stream idle, must return with the result visible to the device.
 * The ranks run the replicated parts (the partial products through `tree`, the coarse solve, the solver around the sweep)
 * in LOCK STEP: `tree` must be a BBFMM_FLAG_DETERMINISTIC handle on every rank (the default path's f64 atomics leave the
 * ranks' replicated results different in their last bits, and a solver near its tolerance could stop on one rank while
 * the others wait in the next all-reduce), or the caller broadcasts the solver's decisions.  A rank that fails inside a
 * sharded level still takes part in the level's all-reduce with a poisoned (NaN) contribution, so every rank returns an
 * error from that bbfmm_schwarz_apply instead of one returning and the others waiting; after a failed CREATE the caller
 * must agree on the status itself before the first apply (ddm.py does: MIN over the group).
 * rank 0 / world 1 (no exchange) is bbfmm_schwarz_create. */
typedef int (*bbfmm_allreduce_fn)(void *user, int64_t count);
int bbfmm_schwarz_create_sharded(bbfmm_handle *tree, const double *points, int64_t n, int32_t d, int64_t ld,
                                 const bbfmm_interpolant *settings, const bbfmm_ddm_params *params, int32_t rank,
                                 int32_t world, double *d_exchange, int64_t exchange_capacity,
                                 bbfmm_allreduce_fn allreduce, void *allreduce_user, bbfmm_schwarz **out);
int64_t bbfmm_schwarz_factor_bytes(const bbfmm_schwarz *h); /* device bytes of the packed factors this handle holds */
/* domains of `level` this handle factorised (returned), the first of them and the level's total in the decomposition */
int64_t bbfmm_schwarz_domains_owned(const bbfmm_schwarz *h, int32_t level, int64_t *first, int64_t *total);
void bbfmm_schwarz_destroy(bbfmm_schwarz *h);
int64_t bbfmm_schwarz_basis_size(const bbfmm_schwarz *h);           /* InterpolantSettings::basis_size */
int32_t bbfmm_schwarz_num_levels(const bbfmm_schwarz *h);
const double *bbfmm_schwarz_monomial_matrix(const bbfmm_schwarz *h); /* N x basis column-major (rbf.rs:485-491) or NULL */
/* evaluate_monomials (ferreus_rbf/src/polynomials.rs:30-74) as the solver's matrix and the domains' matrices compute it:
 * out (n x basis column-major, basis = 1 / d + 1 / (d + 1)(d + 2) / 2 for degree 0 / 1 / 2) on (x - translation) / scale
 * (NULL: 0 and 1).  Host only; the reference's own known answers (polynomials.rs:163-242) are checked through it. */
int bbfmm_debug_evaluate_monomials(const double *points, int64_t n, int32_t d, int64_t ld, int32_t degree,
                                   const double *translation, const double *scale, double *out);
int64_t bbfmm_schwarz_level_size(const bbfmm_schwarz *h, int32_t level);      /* Level::point_indices */
int bbfmm_schwarz_level_points(const bbfmm_schwarz *h, int32_t level, int64_t *out);
/* solve_fine_level / solve_coarse_level (schwarz.rs:84-155) of one level for a given residual
 * (n = N + basis_size values in, n out); parity checks. */
int bbfmm_schwarz_debug_level_solve(bbfmm_schwarz *h, int32_t level, const double *residual, double *out,
                                    int64_t n, int32_t add_poly);
/* a bbfmm_apply_fn: user = bbfmm_schwarz*, vectors of N + basis_size doubles */
int bbfmm_schwarz_apply(void *user, const double *residual, double *correction, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* FERREUS_BBFMM_HIP_H */

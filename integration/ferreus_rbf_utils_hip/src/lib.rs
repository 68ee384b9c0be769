//! `FmmTree` with the method set of `ferreus_rbf_utils::FmmTree`
//! (ferreus_rbf_utils/src/utils.rs:383-494), every pass running on an MI355X through
//! `libferreus_bbfmm_hip.so` (C ABI: include/ferreus_bbfmm_hip.h).
//!
//! Conventions carried over from the reference:
//! * matrices are faer column-major; leading dimensions are passed explicitly, so `MatRef` views with an
//!   arbitrary column stride work unchanged (utils.rs:425-429);
//! * `evaluate*` return `Result<_, ferreus_bbfmm::FmmError>` with the same two variants (bbfmm.rs:20-27);
//! * constructor failures panic, as `FmmTree::new` does (bbfmm.rs:293-298, kernel_helpers.rs:69-70);
//! * all mutating methods take `&mut self`; the solver keeps the tree in a `Mutex` (rbf.rs:87).
//!
//! This file cannot be compiled in the repository that ships it (no Rust toolchain in the build image).
//! The `extern "C"` block below is checked against the header by tests/test_abi_symbols.py.

use faer::{Mat, MatRef};
use ferreus_bbfmm::{FmmError, FmmParams, M2LCompressionType};
use ferreus_rbf_utils::{KernelParams, KernelType};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct BbfmmHandle {
    _private: [u8; 0],
}

/// `bbfmm_params` <-> `FmmParams` (bbfmm.rs:77-104)
#[repr(C)]
pub struct BbfmmParams {
    max_points_per_cell: i64,
    compression_type: i32,
    epsilon: f64,
    eval_chunk_size: i64,
}

const BBFMM_OK: c_int = 0;
const BBFMM_POINT_OUTSIDE_TREE: c_int = 1;
const BBFMM_KERNEL_NO_GRADIENTS: c_int = 2;

unsafe extern "C" {
    fn bbfmm_create(pts: *const f64, n: i64, d: i32, ld: i64, interpolation_order: i32, kernel_type: i32,
                    base_range: f64, total_sill: f64, adaptive_tree: i32, sparse: i32, extents: *const f64,
                    params: *const BbfmmParams, flags: u32, out: *mut *mut BbfmmHandle) -> c_int;
    fn bbfmm_destroy(h: *mut BbfmmHandle);
    fn bbfmm_last_error(h: *const BbfmmHandle) -> *const c_char;
    fn bbfmm_set_weights(h: *mut BbfmmHandle, w: *const f64, rows: i64, k: i32, ldw: i64) -> c_int;
    fn bbfmm_set_local_coefficients(h: *mut BbfmmHandle, w: *const f64, rows: i64, k: i32, ldw: i64) -> c_int;
    fn bbfmm_evaluate(h: *mut BbfmmHandle, w: *const f64, rows: i64, k: i32, ldw: i64, x: *const f64, m: i64,
                      ldx: i64, out: *mut f64, ldo: i64, bad_point_index: *mut i64) -> c_int;
    fn bbfmm_evaluate_with_gradients(h: *mut BbfmmHandle, w: *const f64, rows: i64, k: i32, ldw: i64,
                                     x: *const f64, m: i64, ldx: i64, out: *mut f64, ldo: i64, grad: *mut f64,
                                     ldg: i64, bad_point_index: *mut i64) -> c_int;
    fn bbfmm_evaluate_leaves(h: *mut BbfmmHandle, w: *const f64, rows: i64, k: i32, ldw: i64, x: *const f64,
                             m: i64, ldx: i64, out: *mut f64, ldo: i64, bad_point_index: *mut i64) -> c_int;
    fn bbfmm_evaluate_leaves_with_gradients(h: *mut BbfmmHandle, w: *const f64, rows: i64, k: i32, ldw: i64,
                                            x: *const f64, m: i64, ldx: i64, out: *mut f64, ldo: i64,
                                            grad: *mut f64, ldg: i64, bad_point_index: *mut i64) -> c_int;
    fn bbfmm_source_points(h: *const BbfmmHandle, out: *mut f64, ld: i64) -> c_int;
    fn bbfmm_fast_matrix_vector_product(h: *mut BbfmmHandle, w: *const f64, rows: i64, basis_size: i64,
                                        target_indices: *const i64, n_target_indices: i64, poly: *const f64,
                                        ldp: i64, nugget: f64, result: *mut f64) -> c_int;
}

/// Type-erased evaluator; the kernel is selected at run time by `KernelParams::kernel_type`
/// (the closed registry of utils.rs:558-571, ids in registry order).
#[derive(Debug)]
pub struct FmmTree {
    h: *mut BbfmmHandle,
    source_points: Mat<f64>,
    dimensions: usize,
}

// One in-flight call per handle; the handle owns its HIP stream and may move between host threads.
unsafe impl Send for FmmTree {}

fn kernel_id(k: KernelType) -> i32 {
    match k {
        KernelType::LinearRbf => 0,
        KernelType::ThinPlateSplineRbf => 1,
        KernelType::CubicRbf => 2,
        KernelType::Spheroidal3Rbf => 3,
        KernelType::Spheroidal5Rbf => 4,
        KernelType::Spheroidal7Rbf => 5,
        KernelType::Spheroidal9Rbf => 6,
        KernelType::Laplacian => 7,
        KernelType::OneOverR2 => 8,
        KernelType::OneOverR4 => 9,
    }
}

/// `BBFMM_FLAG_M2L_SHARED_BASIS` (include/ferreus_bbfmm_hip.h): an extension beyond the reference, off by default.
/// The reference's constructor has no argument for it, so the shim reads `FERREUS_BBFMM_M2L_SHARED_BASIS=1` from the
/// environment; without it the handle computes exactly what `ferreus_bbfmm` does.
pub const BBFMM_FLAG_M2L_SHARED_BASIS: u32 = 2;
/// `BBFMM_FLAG_DIRECT_SMALL_W_LEAVES`: the other extension (`FERREUS_BBFMM_DIRECT_SMALL_W_LEAVES=1`).
pub const BBFMM_FLAG_DIRECT_SMALL_W_LEAVES: u32 = 4;
/// `BBFMM_FLAG_DETERMINISTIC` (`FERREUS_BBFMM_DETERMINISTIC=1`): fixed summation order everywhere (no f64 atomics), so
/// that two runs give bitwise equal results, as the reference's per-target sums do.
pub const BBFMM_FLAG_DETERMINISTIC: u32 = 8;
/// Multi-GPU needs nothing here: `bbfmm_create` itself reads `FERREUS_BBFMM_DEVICES` ("0,1,2,3", "all", or "0,0" for
/// logical parts on one device) and then returns ONE handle that spans those devices of this process -- `set_weights` +
/// `evaluate` at the sources, `fast_matrix_vector_product` run partitioned over them with the exchange inside the
/// library (include/ferreus_bbfmm_hip.h: bbfmm_create_on_devices).  `FmmTree::new` has no argument for a device list
/// (utils.rs:392-421), so the environment is where an unchanged caller says it.
fn creation_flags() -> u32 {
    let on = |name: &str| matches!(std::env::var(name), Ok(v) if v == "1");
    (if on("FERREUS_BBFMM_M2L_SHARED_BASIS") { BBFMM_FLAG_M2L_SHARED_BASIS } else { 0 })
        | (if on("FERREUS_BBFMM_DIRECT_SMALL_W_LEAVES") { BBFMM_FLAG_DIRECT_SMALL_W_LEAVES } else { 0 })
        | (if on("FERREUS_BBFMM_DETERMINISTIC") { BBFMM_FLAG_DETERMINISTIC } else { 0 })
}

impl FmmTree {
    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(bbfmm_last_error(self.h)) }.to_string_lossy().into_owned()
    }

    /// `FmmTree::new` (utils.rs:392-421).  Takes ownership of the points like the reference; the library
    /// keeps its own sorted copy on the device.
    #[allow(clippy::too_many_arguments)]
    pub fn new(
        source_points: Mat<f64>,
        interpolation_order: usize,
        kernel_params: KernelParams,
        adaptive_tree: bool,
        sparse: bool,
        extents: Option<Vec<f64>>,
        params: Option<FmmParams>,
    ) -> Self {
        let dimensions = source_points.ncols();
        if let Some(e) = &extents {
            assert_eq!(e.len(), 2 * dimensions, "extents must be [mins..., maxs...]");
        }
        let p = params.map(|p| BbfmmParams {
            max_points_per_cell: p.max_points_per_cell as i64,
            compression_type: match p.compression_type {
                M2LCompressionType::None => 0,
                M2LCompressionType::SVD => 1,
                M2LCompressionType::ACA => 2,
            },
            epsilon: p.epsilon,
            eval_chunk_size: p.eval_chunk_size as i64,
        });
        let mut h: *mut BbfmmHandle = std::ptr::null_mut();
        let rc = unsafe {
            bbfmm_create(
                source_points.as_ptr(),
                source_points.nrows() as i64,
                dimensions as i32,
                source_points.col_stride() as i64,
                interpolation_order as i32,
                kernel_id(kernel_params.kernel_type),
                kernel_params.base_range,
                kernel_params.total_sill,
                adaptive_tree as i32,
                sparse as i32,
                extents.as_ref().map_or(std::ptr::null(), |e| e.as_ptr()),
                p.as_ref().map_or(std::ptr::null(), |p| p as *const BbfmmParams),
                creation_flags(),
                &mut h,
            )
        };
        let tree = Self { h, source_points, dimensions };
        if rc != BBFMM_OK {
            let msg = if h.is_null() { "bbfmm_create failed".to_string() } else { tree.last_error() };
            panic!("{msg}"); // the reference panics on bad constructor arguments (bbfmm.rs:293-298)
        }
        tree
    }

    fn check(&self, rc: c_int, bad: i64) -> Result<(), FmmError> {
        match rc {
            BBFMM_OK => Ok(()),
            BBFMM_POINT_OUTSIDE_TREE => Err(FmmError::PointOutsideTree { point_index: bad as usize }),
            BBFMM_KERNEL_NO_GRADIENTS => Err(FmmError::KernelDoesNotSupportGradients),
            _ => panic!("{}", self.last_error()),
        }
    }

    /// `set_weights` (utils.rs:425-429): upward pass.
    pub fn set_weights(&mut self, w: &MatRef<'_, f64>) {
        let rc = unsafe {
            bbfmm_set_weights(self.h, w.as_ptr(), w.nrows() as i64, w.ncols() as i32, w.col_stride() as i64)
        };
        if rc != BBFMM_OK {
            panic!("{}", self.last_error());
        }
    }

    /// `set_local_coefficients` (utils.rs:433-437): whole-tree downward pass, kept for `evaluate_leaves`.
    pub fn set_local_coefficients(&mut self, w: &MatRef<'_, f64>) {
        let rc = unsafe {
            bbfmm_set_local_coefficients(self.h, w.as_ptr(), w.nrows() as i64, w.ncols() as i32, w.col_stride() as i64)
        };
        if rc != BBFMM_OK {
            panic!("{}", self.last_error());
        }
    }

    /// `evaluate` (utils.rs:441-449): M_t x K potentials.
    ///
    /// The unchanged `ferreus_rbf::fast_matrix_vector_product` (rbf.rs:1357-1364) calls `set_weights(w)` and then
    /// `evaluate(w, select_mat_rows(source_points, idx))`.  Nothing has to change in that caller: the library
    /// recognises targets that are the tree's source points (all rows in order: the FGMRES matvec) or rows of them
    /// (`matvec_partial`, rbf.rs:119-133) bit for bit and serves them from its resident / cached target sets, and
    /// weights equal to those of the preceding `set_weights` are not transferred twice (include/ferreus_bbfmm_hip.h,
    /// `bbfmm_evaluate`; INTEGRATION.md "What each costs": 47.2 ms against 47.0 for the patched entry point at 10M points).
    pub fn evaluate(&mut self, w: &MatRef<'_, f64>, x: &Mat<f64>) -> Result<Mat<f64>, FmmError> {
        let mut out = Mat::<f64>::zeros(x.nrows(), w.ncols());
        let mut bad: i64 = -1;
        let rc = unsafe {
            bbfmm_evaluate(self.h, w.as_ptr(), w.nrows() as i64, w.ncols() as i32, w.col_stride() as i64, x.as_ptr(),
                           x.nrows() as i64, x.col_stride() as i64, out.as_ptr_mut(), out.col_stride() as i64, &mut bad)
        };
        self.check(rc, bad).map(|_| out)
    }

    /// `evaluate_with_gradients` (utils.rs:453-461): potentials and M_t x (K * d) gradients, columns
    /// `[rhs0_dx, rhs0_dy, rhs0_dz, rhs1_dx, ...]` (bbfmm.rs:434-441).
    pub fn evaluate_with_gradients(&mut self, w: &MatRef<'_, f64>, x: &Mat<f64>) -> Result<(Mat<f64>, Mat<f64>), FmmError> {
        let mut out = Mat::<f64>::zeros(x.nrows(), w.ncols());
        let mut grad = Mat::<f64>::zeros(x.nrows(), w.ncols() * self.dimensions);
        let mut bad: i64 = -1;
        let rc = unsafe {
            bbfmm_evaluate_with_gradients(self.h, w.as_ptr(), w.nrows() as i64, w.ncols() as i32, w.col_stride() as i64,
                                          x.as_ptr(), x.nrows() as i64, x.col_stride() as i64, out.as_ptr_mut(),
                                          out.col_stride() as i64, grad.as_ptr_mut(), grad.col_stride() as i64, &mut bad)
        };
        self.check(rc, bad).map(|_| (out, grad))
    }

    /// `evaluate_leaves` (utils.rs:465-473): leaf pass only, after `set_local_coefficients`.
    pub fn evaluate_leaves(&mut self, w: &MatRef<'_, f64>, x: &Mat<f64>) -> Result<Mat<f64>, FmmError> {
        let mut out = Mat::<f64>::zeros(x.nrows(), w.ncols());
        let mut bad: i64 = -1;
        let rc = unsafe {
            bbfmm_evaluate_leaves(self.h, w.as_ptr(), w.nrows() as i64, w.ncols() as i32, w.col_stride() as i64, x.as_ptr(),
                                  x.nrows() as i64, x.col_stride() as i64, out.as_ptr_mut(), out.col_stride() as i64,
                                  &mut bad)
        };
        self.check(rc, bad).map(|_| out)
    }

    /// `evaluate_leaves_with_gradients` (utils.rs:477-485).
    pub fn evaluate_leaves_with_gradients(&mut self, w: &MatRef<'_, f64>, x: &Mat<f64>) -> Result<(Mat<f64>, Mat<f64>), FmmError> {
        let mut out = Mat::<f64>::zeros(x.nrows(), w.ncols());
        let mut grad = Mat::<f64>::zeros(x.nrows(), w.ncols() * self.dimensions);
        let mut bad: i64 = -1;
        let rc = unsafe {
            bbfmm_evaluate_leaves_with_gradients(self.h, w.as_ptr(), w.nrows() as i64, w.ncols() as i32,
                                                 w.col_stride() as i64, x.as_ptr(), x.nrows() as i64, x.col_stride() as i64,
                                                 out.as_ptr_mut(), out.col_stride() as i64, grad.as_ptr_mut(),
                                                 grad.col_stride() as i64, &mut bad)
        };
        self.check(rc, bad).map(|_| (out, grad))
    }

    /// `source_points` (utils.rs:489-493): the Rust-side copy the tree was built from.
    pub fn source_points(&self) -> &Mat<f64> {
        &self.source_points
    }

    /// The library's own copy of the points (`bbfmm_source_points`); equals `source_points()`.
    pub fn source_points_from_device_handle(&self) -> Mat<f64> {
        let mut out = Mat::<f64>::zeros(self.source_points.nrows(), self.dimensions);
        let rc = unsafe { bbfmm_source_points(self.h, out.as_ptr_mut(), out.col_stride() as i64) };
        assert_eq!(rc, BBFMM_OK, "{}", self.last_error());
        out
    }

    /// The whole FGMRES matvec (`fast_matrix_vector_product`, ferreus_rbf/src/rbf.rs:1338-1379) in one call:
    /// the targets are the sources, which never leave the device.  `weights` has N + basis_size rows; the
    /// result has the same shape, rows outside `target_indices` (and the last basis_size rows) are zero.
    pub fn fast_matrix_vector_product(
        &mut self,
        weights: &MatRef<'_, f64>,
        basis_size: usize,
        target_indices: Option<&Vec<usize>>,
        monomial_matrix: Option<&Mat<f64>>,
        nugget: f64,
    ) -> Mat<f64> {
        assert_eq!(weights.ncols(), 1);
        assert_eq!(weights.row_stride(), 1);
        let idx: Option<Vec<i64>> = target_indices.map(|v| v.iter().map(|&i| i as i64).collect());
        let mut result = Mat::<f64>::zeros(weights.nrows(), 1);
        let rc = unsafe {
            bbfmm_fast_matrix_vector_product(
                self.h,
                weights.as_ptr(),
                weights.nrows() as i64,
                basis_size as i64,
                idx.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()),
                idx.as_ref().map_or(0, |v| v.len() as i64),
                monomial_matrix.map_or(std::ptr::null(), |p| p.as_ptr()),
                monomial_matrix.map_or(0, |p| p.col_stride() as i64),
                nugget,
                result.as_ptr_mut(),
            )
        };
        if rc != BBFMM_OK {
            panic!("{}", self.last_error());
        }
        result
    }
}

impl Drop for FmmTree {
    fn drop(&mut self) {
        if !self.h.is_null() {
            unsafe { bbfmm_destroy(self.h) }
        }
    }
}

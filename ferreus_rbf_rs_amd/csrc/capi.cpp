// extern "C" boundary: include/ferreus_bbfmm_hip.h over bbfmm::FmmTree.
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <memory>

#include "device_group.hpp"
#include "ferreus_bbfmm_hip.h"
#include "fmm_tree.hpp"
#include "morton.hpp"

struct bbfmm_handle {
    bbfmm::FmmTree tree;                       // the handle's tree; part 0 of its device group when there is one
    std::unique_ptr<bbfmm::DeviceGroup> group; // several devices (or logical parts) behind this one handle
    std::string err;
};

using bbfmm::FmmTree;

// A failure of the group: its message becomes the handle's.
static int group_rc(bbfmm_handle *h, int rc) {
    if (rc != BBFMM_OK) h->err = h->group->last_error();
    return rc;
}

#define GUARD(h)                              \
    if (!(h)) return BBFMM_BAD_ARGUMENT;      \
    try {                                     \
        (h)->err.clear();                     \
        (h)->tree.bind_device();
#define END_GUARD(h)                                   \
    }                                                  \
    catch (const std::bad_alloc &) {                   \
        (h)->err = "out of host memory";               \
        return BBFMM_BAD_ARGUMENT;                     \
    }                                                  \
    catch (const std::exception &e) {                  \
        (h)->err = std::string("exception: ") + e.what(); \
        return BBFMM_BAD_ARGUMENT;                     \
    }                                                  \
    catch (...) {                                      \
        (h)->err = "unknown exception";                \
        return BBFMM_BAD_ARGUMENT;                     \
    }

extern "C" {

void bbfmm_params_defaults(int32_t interpolation_order, bbfmm_params *out) {
    if (!out) return;
    out->max_points_per_cell = 256;
    out->compression_type = BBFMM_COMPRESSION_ACA;
    double eps = 1.0; // 10f64.powi(-order), bbfmm.rs:100
    for (int i = 0; i < interpolation_order; ++i) eps *= 10.0;
    out->epsilon = 1.0 / eps;
    out->eval_chunk_size = 1024;
}

// bbfmm_create on an explicit device list; devices == NULL: the current device (and no group).
static int create_impl(const double *pts, int64_t n, int32_t d, int64_t ld, int32_t interpolation_order, int32_t kernel_type,
                       double base_range, double total_sill, int32_t adaptive_tree, int32_t sparse, const double *extents,
                       const bbfmm_params *params, uint32_t flags, const std::vector<int> &devices, bbfmm_handle **out) {
    if (!out) return BBFMM_BAD_ARGUMENT;
    *out = nullptr;
    bbfmm_handle *h = nullptr;
    int cur = -1;
    try {
        h = new bbfmm_handle();
        *out = h; // returned even on failure so that bbfmm_last_error can be read; the caller destroys it either way
        if (!devices.empty()) {
            if (flags & BBFMM_FLAG_HOST_ONLY) {
                h->err = "a device list needs devices (BBFMM_FLAG_HOST_ONLY is set)";
                return BBFMM_BAD_ARGUMENT;
            }
            int ndev = 0;
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
                h->err = "no HIP device available (the BBFMM passes have no CPU fallback)";
                return BBFMM_DEVICE_ERROR;
            }
            for (int dev : devices)
                if (dev < 0 || dev >= ndev) {
                    h->err = "device " + std::to_string(dev) + " of the device list does not exist (" + std::to_string(ndev) + " visible)";
                    return BBFMM_BAD_ARGUMENT;
                }
            if (devices.size() > static_cast<size_t>(bbfmm::kMaxScatterParts)) {
                h->err = "more parts in the device list than a handle takes";
                return BBFMM_BAD_ARGUMENT;
            }
            (void)hipGetDevice(&cur);
            if (hipSetDevice(devices[0]) != hipSuccess) {
                h->err = "hipSetDevice failed for the first device of the list";
                return BBFMM_DEVICE_ERROR;
            }
        }
        int rc = h->tree.create(pts, n, d, ld, interpolation_order, kernel_type, base_range, total_sill, adaptive_tree != 0, sparse != 0,
                                extents, params, flags);
        if (rc == BBFMM_OK && devices.size() > 1) {
            h->group.reset(new bbfmm::DeviceGroup());
            const bbfmm::GroupCreateArgs a{pts, n, d, ld, interpolation_order, kernel_type, base_range, total_sill, adaptive_tree != 0,
                                           sparse != 0, extents, params, flags};
            rc = h->group->init(&h->tree, a, devices);
            if (rc != BBFMM_OK) {
                h->err = h->group->last_error();
                h->group.reset();
            }
        }
        if (cur >= 0) (void)hipSetDevice(cur); // the caller's current device is not the library's to change
        return rc;
    } catch (const std::exception &e) {
        if (cur >= 0) (void)hipSetDevice(cur);
        if (h) h->err = std::string("exception: ") + e.what();
        return BBFMM_BAD_ARGUMENT;
    } catch (...) {
        if (cur >= 0) (void)hipSetDevice(cur);
        if (h) h->err = "unknown exception";
        return BBFMM_BAD_ARGUMENT;
    }
}

int bbfmm_create(const double *pts, int64_t n, int32_t d, int64_t ld, int32_t interpolation_order,
                 int32_t kernel_type, double base_range, double total_sill, int32_t adaptive_tree, int32_t sparse,
                 const double *extents, const bbfmm_params *params, uint32_t flags, bbfmm_handle **out) {
    // FERREUS_BBFMM_DEVICES: the device list of every handle this process creates through the reference's constructor,
    // which has no argument for it ("0,1,2,3", "all", or "0,0" for logical parts on one device).  Host-only handles
    // (structure tests) ignore it.
    std::vector<int> devices;
    const char *env = std::getenv("FERREUS_BBFMM_DEVICES");
    if (env && *env && !(flags & BBFMM_FLAG_HOST_ONLY)) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
        std::string perr;
        if (!bbfmm::parse_device_list(env, ndev, &devices, &perr)) {
            if (!out) return BBFMM_BAD_ARGUMENT;
            bbfmm_handle *h = new (std::nothrow) bbfmm_handle();
            if (h) h->err = "FERREUS_BBFMM_DEVICES: " + perr;
            *out = h;
            return BBFMM_BAD_ARGUMENT;
        }
    }
    return create_impl(pts, n, d, ld, interpolation_order, kernel_type, base_range, total_sill, adaptive_tree, sparse, extents, params, flags,
                       devices, out);
}

int bbfmm_create_on_devices(const double *pts, int64_t n, int32_t d, int64_t ld, int32_t interpolation_order, int32_t kernel_type,
                            double base_range, double total_sill, int32_t adaptive_tree, int32_t sparse, const double *extents,
                            const bbfmm_params *params, uint32_t flags, const int32_t *devices, int32_t n_devices, bbfmm_handle **out) {
    if (!out) return BBFMM_BAD_ARGUMENT;
    *out = nullptr;
    if (!devices || n_devices < 1) return BBFMM_BAD_ARGUMENT;
    return create_impl(pts, n, d, ld, interpolation_order, kernel_type, base_range, total_sill, adaptive_tree, sparse, extents, params, flags,
                       std::vector<int>(devices, devices + n_devices), out);
}

int32_t bbfmm_device_count(const bbfmm_handle *h) { return !h ? -1 : (h->group ? h->group->n_parts() : 1); }
int32_t bbfmm_part_device(const bbfmm_handle *h, int32_t part) {
    if (!h || part < 0) return -1;
    if (h->group) return part < h->group->n_parts() ? h->group->part_device(part) : -1;
    return part == 0 ? h->tree.device() : -1;
}
int bbfmm_group_bounds(const bbfmm_handle *h, int64_t *bounds_out) {
    if (!h || !h->group || !bounds_out) return BBFMM_BAD_ARGUMENT;
    const std::vector<int64_t> &b = h->group->bounds();
    std::copy(b.begin(), b.end(), bounds_out);
    return BBFMM_OK;
}
int bbfmm_get_part_phase_ms(bbfmm_handle *h, int32_t part, double *ms_out, int64_t *count_out) {
    if (!h || !ms_out) return BBFMM_BAD_ARGUMENT;
    if (h->group) return h->group->part_phase_ms(part, ms_out, count_out);
    return part == 0 ? bbfmm_get_phase_ms(h, ms_out, count_out) : BBFMM_BAD_ARGUMENT;
}

void bbfmm_destroy(bbfmm_handle *h) {
    if (!h) return;
    h->group.reset(); // (its parts end before the primary they lean on)
    delete h;
}

const char *bbfmm_last_error(const bbfmm_handle *h) {
    if (!h) return "null handle";
    if (!h->err.empty()) return h->err.c_str();
    return h->tree.last_error();
}

int bbfmm_set_weights(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw) {
    GUARD(h)
    if (h->group) return group_rc(h, h->group->set_weights(w, rows, k, ldw));
    return h->tree.set_weights(w, rows, k, ldw);
    END_GUARD(h)
}

int bbfmm_set_local_coefficients(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw) {
    GUARD(h)
    if (h->group) { // the weights of set_weights: every part stores the whole-tree expansions (Leaves mode over the group)
        bool handled = false;
        int rc = group_rc(h, h->group->set_local_coefficients_all(w, rows, k, ldw, &handled));
        if (rc != BBFMM_OK || handled) return rc;
        rc = group_rc(h, h->group->prepare_primary(false)); // other weights: the primary alone
        if (rc != BBFMM_OK) return rc;
    }
    return h->tree.set_local_coefficients(w, rows, k, ldw);
    END_GUARD(h)
}

int bbfmm_evaluate(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw, const double *x, int64_t m,
                   int64_t ldx, double *out, int64_t ldo, int64_t *bad_point_index) {
    GUARD(h)
    if (h->group) { // targets = sources with the weights of set_weights: partitioned over the group; anything else: the primary
        bool handled = false;
        int rc = group_rc(h, h->group->evaluate_at_sources(w, rows, k, ldw, x, m, ldx, out, ldo, &handled));
        if (rc != BBFMM_OK || handled) return rc;
        rc = h->group->evaluate_rows_of_sources(w, rows, k, ldw, x, m, ldx, out, &handled); // the unchanged caller's matvec_partial
        if (rc != BBFMM_OK || handled) return group_rc(h, rc);
        rc = h->group->evaluate_sharded(w, rows, k, ldw, x, m, ldx, out, ldo, nullptr, 0, false, false, bad_point_index, &handled);
        if (rc != BBFMM_OK || handled) return group_rc(h, rc);
        rc = group_rc(h, h->group->prepare_primary(h->group->weights_match_staged(w, rows, k, ldw)));
        if (rc != BBFMM_OK) return rc;
    }
    return h->tree.evaluate(w, rows, k, ldw, x, m, ldx, out, ldo, nullptr, 0, false, false, bad_point_index);
    END_GUARD(h)
}

int bbfmm_evaluate_with_gradients(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw,
                                  const double *x, int64_t m, int64_t ldx, double *out, int64_t ldo, double *grad,
                                  int64_t ldg, int64_t *bad_point_index) {
    GUARD(h)
    if (h->group) {
        bool handled = false;
        int rc = h->group->evaluate_sharded(w, rows, k, ldw, x, m, ldx, out, ldo, grad, ldg, true, false, bad_point_index, &handled);
        if (rc != BBFMM_OK || handled) return group_rc(h, rc);
        rc = group_rc(h, h->group->prepare_primary(!w || h->group->weights_match_staged(w, rows, k, ldw)));
        if (rc != BBFMM_OK) return rc;
    }
    return h->tree.evaluate(w, rows, k, ldw, x, m, ldx, out, ldo, grad, ldg, true, false, bad_point_index);
    END_GUARD(h)
}

int bbfmm_evaluate_leaves(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw, const double *x,
                          int64_t m, int64_t ldx, double *out, int64_t ldo, int64_t *bad_point_index) {
    GUARD(h)
    if (h->group) {
        bool handled = false;
        int rc = h->group->evaluate_sharded(w, rows, k, ldw, x, m, ldx, out, ldo, nullptr, 0, false, true, bad_point_index, &handled);
        if (rc != BBFMM_OK || handled) return group_rc(h, rc);
        rc = group_rc(h, h->group->prepare_primary(!w || h->group->weights_match_staged(w, rows, k, ldw)));
        if (rc != BBFMM_OK) return rc;
    }
    return h->tree.evaluate(w, rows, k, ldw, x, m, ldx, out, ldo, nullptr, 0, false, true, bad_point_index);
    END_GUARD(h)
}

int bbfmm_evaluate_leaves_with_gradients(bbfmm_handle *h, const double *w, int64_t rows, int32_t k, int64_t ldw,
                                         const double *x, int64_t m, int64_t ldx, double *out, int64_t ldo,
                                         double *grad, int64_t ldg, int64_t *bad_point_index) {
    GUARD(h)
    if (h->group) {
        bool handled = false;
        int rc = h->group->evaluate_sharded(w, rows, k, ldw, x, m, ldx, out, ldo, grad, ldg, true, true, bad_point_index, &handled);
        if (rc != BBFMM_OK || handled) return group_rc(h, rc);
        rc = group_rc(h, h->group->prepare_primary(!w || h->group->weights_match_staged(w, rows, k, ldw)));
        if (rc != BBFMM_OK) return rc;
    }
    return h->tree.evaluate(w, rows, k, ldw, x, m, ldx, out, ldo, grad, ldg, true, true, bad_point_index);
    END_GUARD(h)
}

int bbfmm_source_points(const bbfmm_handle *h, double *out, int64_t ld) {
    if (!h || !out) return BBFMM_BAD_ARGUMENT;
    const auto &p = h->tree.source_points();
    const int64_t n = h->tree.tree().n_points;
    const int d = h->tree.tree().d;
    if (ld < n) return BBFMM_BAD_ARGUMENT;
    for (int a = 0; a < d; ++a) std::memcpy(out + a * ld, p.data() + static_cast<size_t>(a) * n, n * sizeof(double));
    return BBFMM_OK;
}

int bbfmm_fast_matrix_vector_product(bbfmm_handle *h, const double *w, int64_t rows, int64_t basis_size,
                                     const int64_t *target_indices, int64_t n_target_indices, const double *poly,
                                     int64_t ldp, double nugget, double *result) {
    GUARD(h)
    if (h->group) {
        if (!target_indices || h->tree.is_identity_subset(target_indices, n_target_indices))
            return group_rc(h, h->group->fast_matrix_vector_product(w, rows, basis_size, poly, ldp, nugget, result));
        // a row subset (matvec_partial, rbf.rs:119-133): every part takes the rows of the set it owns
        return group_rc(h, h->group->fast_matvec_subset(w, rows, basis_size, target_indices, n_target_indices, poly, ldp, nugget, result));
    }
    return h->tree.fast_matrix_vector_product(w, rows, basis_size, target_indices, n_target_indices, poly, ldp, nugget,
                                              result);
    END_GUARD(h)
}

int bbfmm_prepare_target_subset(bbfmm_handle *h, const int64_t *target_indices, int64_t n_target_indices) {
    GUARD(h) return h->tree.prepare_target_subset(target_indices, n_target_indices);
    END_GUARD(h)
}

int bbfmm_matvec_device(bbfmm_handle *h, const double *d_w, int64_t ldw, int32_t k, double *d_out, int64_t ldo,
                        int32_t sync) {
    GUARD(h)
    if (h->group) return group_rc(h, h->group->matvec_device(d_w, ldw, k, d_out, ldo, sync != 0));
    return h->tree.matvec_device(d_w, ldw, k, d_out, ldo, sync != 0);
    END_GUARD(h)
}

void *bbfmm_stream(bbfmm_handle *h) {
    if (!h) return nullptr;
    h->tree.bind_device();
    return static_cast<void *>(h->tree.stream());
}

int bbfmm_target_subset_create(bbfmm_handle *h, const int64_t *target_indices, int64_t n_target_indices,
                               int32_t *subset_id) {
    GUARD(h)
    int id = 0;
    const int rc = h->tree.register_subset(target_indices, n_target_indices, &id);
    if (rc == BBFMM_OK && subset_id) *subset_id = id;
    return rc;
    END_GUARD(h)
}

int bbfmm_matvec_subset_device(bbfmm_handle *h, int32_t subset_id, const double *d_w, double *d_y, int32_t sync) {
    GUARD(h)
    if (h->group) {
        if (subset_id == -1) return group_rc(h, h->group->matvec_device(d_w, h->tree.tree().n_points, 1, d_y, h->tree.tree().n_points, sync != 0));
        h->group->primary_state_changed(); // a registered row subset: the primary alone
    }
    return h->tree.matvec_subset_device(subset_id, d_w, d_y, sync != 0);
    END_GUARD(h)
}

int bbfmm_set_partition(bbfmm_handle *h, int32_t rank, int32_t world) {
    GUARD(h)
    if (h->group) {
        h->err = "the handle spans a device group: its partition is the group's";
        return BBFMM_BAD_ARGUMENT;
    }
    return h->tree.set_partition(rank, world);
    END_GUARD(h)
}

int64_t bbfmm_partition_row_count(const bbfmm_handle *h) {
    if (!h) return -1;
    const auto &r = h->tree.partition_rows();
    return h->tree.partitioned() ? static_cast<int64_t>(r.size()) : h->tree.tree().n_points;
}

int bbfmm_partition_rows(const bbfmm_handle *h, int64_t *rows_out) {
    if (!h || !rows_out) return BBFMM_BAD_ARGUMENT;
    const auto &r = h->tree.partition_rows();
    if (!h->tree.partitioned()) {
        for (int64_t i = 0; i < h->tree.tree().n_points; ++i) rows_out[i] = i;
    } else if (!r.empty()) {
        std::memcpy(rows_out, r.data(), r.size() * sizeof(int64_t));
    }
    return BBFMM_OK;
}

int64_t bbfmm_partition_coarse_count(const bbfmm_handle *h) { return h ? h->tree.partition_coarse_count() : -1; }

int bbfmm_matvec_partition_upward(bbfmm_handle *h, const double *d_w, int64_t ldw, int32_t k, double *d_coarse, void *comm_stream) {
    GUARD(h)
    if (h->group) {
        h->err = "the handle spans a device group: the exchange is the library's";
        return BBFMM_BAD_ARGUMENT;
    }
    return h->tree.matvec_partition_upward(d_w, ldw, k, d_coarse, static_cast<hipStream_t>(comm_stream));
    END_GUARD(h)
}

int32_t bbfmm_partition_world(const bbfmm_handle *h) {
    if (!h) return -1;
    const std::vector<int64_t> &b = h->tree.partition_bounds();
    return b.empty() ? 1 : static_cast<int32_t>(b.size()) - 1;
}
int32_t bbfmm_partition_rank(const bbfmm_handle *h) { return h ? h->tree.partition_rank() : -1; }
int bbfmm_partition_bounds(const bbfmm_handle *h, int32_t world, int64_t *bounds_out) {
    if (!h || !bounds_out) return BBFMM_BAD_ARGUMENT;
    const std::vector<int64_t> &b = h->tree.partition_bounds();
    if (b.empty() || static_cast<size_t>(world) + 1 != b.size()) return BBFMM_BAD_ARGUMENT;
    std::copy(b.begin(), b.end(), bounds_out);
    return BBFMM_OK;
}
int bbfmm_matvec_partition_finish_sorted(bbfmm_handle *h, const double *d_coarse, double *d_seg, int64_t ld, void *comm_stream) {
    GUARD(h)
    if (h->group) {
        h->err = "the handle spans a device group: the exchange is the library's";
        return BBFMM_BAD_ARGUMENT;
    }
    return h->tree.matvec_partition_finish_sorted(d_coarse, d_seg, ld, static_cast<hipStream_t>(comm_stream));
    END_GUARD(h)
}
int bbfmm_partition_scatter(bbfmm_handle *h, const double *d_all, int32_t first_part, int32_t n_parts, int64_t m_max, int32_t k,
                            double *d_out, int64_t ldo) {
    GUARD(h)
    if (h->group) {
        h->err = "the handle spans a device group: the exchange is the library's";
        return BBFMM_BAD_ARGUMENT;
    }
    return h->tree.partition_scatter(d_all, first_part, n_parts, m_max, k, d_out, ldo);
    END_GUARD(h)
}
int bbfmm_matvec_partition_finish(bbfmm_handle *h, const double *d_coarse, double *d_out, int64_t ldo, int32_t sync, void *comm_stream) {
    GUARD(h)
    if (h->group) {
        h->err = "the handle spans a device group: the exchange is the library's";
        return BBFMM_BAD_ARGUMENT;
    }
    return h->tree.matvec_partition_finish(d_coarse, d_out, ldo, sync != 0, static_cast<hipStream_t>(comm_stream));
    END_GUARD(h)
}

int bbfmm_debug_partition_upward_counts(const bbfmm_handle *h, int64_t *counts_out, uint8_t *reads_out, int64_t *info_out) {
    if (!h || !counts_out || !reads_out || !info_out) return BBFMM_BAD_ARGUMENT;
    return h->tree.debug_partition_upward_counts(counts_out, reads_out, info_out);
}

int bbfmm_get_tree_stats(const bbfmm_handle *h, bbfmm_tree_stats *out) {
    if (!h || !out) return BBFMM_BAD_ARGUMENT;
    h->tree.stats(out);
    return BBFMM_OK;
}

int bbfmm_debug_targets_are_sources(const bbfmm_handle *h, const double *x, int64_t m, int64_t ldx) {
    return (h && h->tree.targets_are_sources(x, m, ldx)) ? 1 : 0;
}
int bbfmm_debug_rows_of_sources(bbfmm_handle *h, const double *x, int64_t m, int64_t ldx, int64_t *rows_out) {
    if (!h || !rows_out) return 0;
    try {
        std::vector<int64_t> rows;
        if (!h->tree.targets_are_rows_of_sources(x, m, ldx, &rows)) return 0;
        std::copy(rows.begin(), rows.end(), rows_out);
        return 1;
    } catch (...) {
        return 0;
    }
}
int bbfmm_last_evaluate_at_sources(const bbfmm_handle *h) {
    if (!h) return 0;
    if (h->group && h->group->last_path()) return h->group->last_path(); // 1: partitioned at the sources, 3: targets sharded over the parts
    return h->tree.last_evaluate_path();
}
int bbfmm_tree_built_on_device(const bbfmm_handle *h) { return (h && h->tree.tree_built_on_device()) ? 1 : 0; }

int bbfmm_get_cells(const bbfmm_handle *h, uint64_t *keys, uint8_t *is_leaf) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    const auto &t = h->tree.tree();
    if (keys) std::memcpy(keys, t.key.data(), t.key.size() * sizeof(uint64_t));
    if (is_leaf) std::memcpy(is_leaf, t.is_leaf.data(), t.is_leaf.size());
    return BBFMM_OK;
}

int bbfmm_get_leaf_sources(const bbfmm_handle *h, int64_t *ptr, int64_t *idx) {
    if (!h || !ptr || !idx) return BBFMM_BAD_ARGUMENT;
    const auto &t = h->tree.tree();
    const int64_t C = t.n_cells();
    int64_t cur = 0;
    for (int64_t c = 0; c < C; ++c) {
        ptr[c] = cur;
        if (t.is_leaf[c])
            for (int64_t q = t.pt_begin[c]; q < t.pt_end[c]; ++q) idx[cur++] = t.order[q];
    }
    ptr[C] = cur;
    return BBFMM_OK;
}

int bbfmm_get_list(const bbfmm_handle *h, char which, int64_t *ptr, int32_t *idx, int64_t *n_entries) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    const auto &t = h->tree.tree();
    const bbfmm::Csr *l = nullptr;
    switch (which) {
    case 'U': case 'u': l = &t.u; break;
    case 'V': case 'v': l = &t.v; break;
    case 'W': case 'w': l = &t.w; break;
    case 'X': case 'x': l = &t.x; break;
    default: return BBFMM_BAD_ARGUMENT;
    }
    if (n_entries) *n_entries = static_cast<int64_t>(l->idx.size());
    if (ptr) std::memcpy(ptr, l->ptr.data(), l->ptr.size() * sizeof(int64_t));
    if (idx && !l->idx.empty()) std::memcpy(idx, l->idx.data(), l->idx.size() * sizeof(int32_t));
    return BBFMM_OK;
}

int bbfmm_get_m2l_ranks(const bbfmm_handle *h, int32_t *ranks, int32_t *n_ref_out) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    const auto &o = h->tree.ops();
    if (n_ref_out) *n_ref_out = o.n_ref;
    if (ranks)
        for (size_t level = 0; level < o.m2l.size(); ++level)
            for (int r = 0; r < o.n_ref; ++r)
                ranks[level * o.n_ref + r] = o.m2l[level].empty() ? 0 : o.m2l[level][r].rank;
    return BBFMM_OK;
}

int bbfmm_get_m2l_operator(const bbfmm_handle *h, int32_t level, int32_t ref, double *out) {
    if (!h || !out) return BBFMM_BAD_ARGUMENT;
    const auto &o = h->tree.ops();
    if (level < 2 || level >= static_cast<int>(o.m2l.size()) || ref < 0 || ref >= o.n_ref) return BBFMM_BAD_ARGUMENT;
    const auto &op = o.m2l[level][ref];
    const int n = o.n;
    if (op.vt.empty()) {
        std::memcpy(out, op.u.data(), sizeof(double) * n * n);
        return BBFMM_OK;
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = 0; k < op.rank; ++k) s += op.u[static_cast<size_t>(k) * n + i] * op.vt[static_cast<size_t>(j) * op.rank + k];
            out[static_cast<size_t>(j) * n + i] = s;
        }
    return BBFMM_OK;
}

int bbfmm_get_m2l_factors(const bbfmm_handle *h, int32_t level, int32_t ref, double *u, double *vt) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    const auto &o = h->tree.ops();
    if (level < 2 || level >= static_cast<int>(o.m2l.size()) || ref < 0 || ref >= o.n_ref) return BBFMM_BAD_ARGUMENT;
    const auto &op = o.m2l[level][ref];
    if (u) std::memcpy(u, op.u.data(), op.u.size() * sizeof(double));
    if (vt && !op.vt.empty()) std::memcpy(vt, op.vt.data(), op.vt.size() * sizeof(double));
    return BBFMM_OK;
}

int bbfmm_get_permutation_tables(const bbfmm_handle *h, int32_t *n_perm, int32_t *perm, int32_t *invperm,
                                 int32_t *perm_lookup, int32_t *ref_lookup) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    const auto &o = h->tree.ops();
    if (n_perm) *n_perm = o.n_perm;
    if (perm) std::memcpy(perm, o.perm.data(), o.perm.size() * sizeof(int32_t));
    if (invperm) std::memcpy(invperm, o.invperm.data(), o.invperm.size() * sizeof(int32_t));
    if (perm_lookup) std::memcpy(perm_lookup, o.perm_lookup.data(), o.perm_lookup.size() * sizeof(int32_t));
    if (ref_lookup) std::memcpy(ref_lookup, o.ref_lookup.data(), o.ref_lookup.size() * sizeof(int32_t));
    return BBFMM_OK;
}

int bbfmm_points_to_leaves(const bbfmm_handle *h, const double *x, int64_t m, int64_t ldx, int32_t *cell_out,
                           int64_t *bad_point_index) {
    if (!h || (m > 0 && (!x || !cell_out)) || ldx < m) return BBFMM_BAD_ARGUMENT;
    const int64_t bad = bbfmm::points_to_leaves(h->tree.tree(), x, m, ldx, cell_out);
    if (bad >= 0) {
        if (bad_point_index) *bad_point_index = bad;
        return BBFMM_POINT_OUTSIDE_TREE;
    }
    return BBFMM_OK;
}

int bbfmm_set_profiling(bbfmm_handle *h, int32_t enable) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    if (h->group) h->group->set_profiling(enable != 0);
    else h->tree.set_profiling(enable != 0);
    return BBFMM_OK;
}

int bbfmm_get_phase_ms(bbfmm_handle *h, double *ms_out, int64_t *count_out) {
    if (!h || !ms_out) return BBFMM_BAD_ARGUMENT;
    h->tree.bind_device();
    std::memcpy(ms_out, h->tree.phase_ms(), sizeof(double) * BBFMM_N_PHASES);
    if (count_out) std::memcpy(count_out, h->tree.phase_count(), sizeof(int64_t) * BBFMM_N_PHASES);
    return BBFMM_OK;
}

int bbfmm_reset_phase_ms(bbfmm_handle *h) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    if (h->group) h->group->reset_phase_ms();
    else h->tree.reset_phase_ms();
    return BBFMM_OK;
}

int bbfmm_fp64_valu_selftest(double *tflops, double *clock_mhz) {
    int ndev = 0;
    if (!tflops || !clock_mhz) return BBFMM_BAD_ARGUMENT;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return BBFMM_DEVICE_ERROR;
    return bbfmm::valu_f64_selftest(tflops, clock_mhz) == 0 ? BBFMM_OK : BBFMM_DEVICE_ERROR;
}

int bbfmm_mfma_f64_selftest(double *tflops, int32_t *layout_errors, double *info6) {
    double tf = 0;
    int errs = -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return BBFMM_DEVICE_ERROR;
    const int rc = bbfmm::mfma_f64_selftest(&tf, &errs, info6);
    if (tflops) *tflops = tf;
    if (layout_errors) *layout_errors = errs;
    return rc == 0 ? BBFMM_OK : BBFMM_DEVICE_ERROR;
}

// Host-side legs of the host-buffer entry points, by themselves (scripts/host_buffer_legs.py): n doubles copied by the
// pool in 2 MB pieces (the staging copy), gathered through a random permutation (the row writes of a device group),
// scattered through it.  out4 = {memcpy ms, gather ms, scatter ms, threads}; medians of five.
int bbfmm_debug_host_copy_rates(int64_t n, double *out4) {
    if (n < 1 || !out4) return BBFMM_BAD_ARGUMENT;
    try {
        std::vector<double, bbfmm::DefaultInitAllocator<double>> a(static_cast<size_t>(n)), b(static_cast<size_t>(n));
        std::vector<int32_t> perm(static_cast<size_t>(n));
        bbfmm::parallel_for_chunks(n, int64_t(1) << 18, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) a[static_cast<size_t>(i)] = static_cast<double>(i), b[static_cast<size_t>(i)] = 0.0, perm[static_cast<size_t>(i)] = static_cast<int32_t>(i);
        });
        uint64_t st = 88172645463325252ull;
        for (int64_t i = n; i > 1; --i) {
            st ^= st << 13, st ^= st >> 7, st ^= st << 17;
            std::swap(perm[static_cast<size_t>(i - 1)], perm[static_cast<size_t>(st % static_cast<uint64_t>(i))]);
        }
        auto med5 = [&](auto &&fn) {
            double t[5];
            for (double &v : t) {
                const auto t0 = std::chrono::steady_clock::now();
                fn();
                v = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            std::sort(t, t + 5);
            return t[2];
        };
        out4[0] = med5([&] {
            bbfmm::parallel_for_chunks(n, int64_t(1) << 18, [&](int64_t lo, int64_t hi) { std::memcpy(&b[static_cast<size_t>(lo)], &a[static_cast<size_t>(lo)], static_cast<size_t>(hi - lo) * 8); });
        });
        out4[1] = med5([&] {
            bbfmm::parallel_for_chunks(n, int64_t(1) << 16, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) b[static_cast<size_t>(i)] = a[static_cast<size_t>(perm[static_cast<size_t>(i)])]; });
        });
        out4[2] = med5([&] {
            bbfmm::parallel_for_chunks(n, int64_t(1) << 16, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) b[static_cast<size_t>(perm[static_cast<size_t>(i)])] = a[static_cast<size_t>(i)]; });
        });
        out4[3] = static_cast<double>(bbfmm::host_threads());
        return BBFMM_OK;
    } catch (...) {
        return BBFMM_BAD_ARGUMENT;
    }
}

// Test hooks (host only): dense M2M matrix of the reference and the stacked-table M2L.
int bbfmm_debug_dense_m2m(const bbfmm_handle *h, int32_t child_index, double *out) {
    if (!h || !out) return BBFMM_BAD_ARGUMENT;
    std::vector<double> m;
    bbfmm::dense_m2m_matrix(h->tree.ops(), child_index, &m);
    std::memcpy(out, m.data(), m.size() * sizeof(double));
    return BBFMM_OK;
}

int bbfmm_debug_apply_m2l_tables_host(const bbfmm_handle *h, const double *M, double *L) {
    if (!h || !M || !L) return BBFMM_BAD_ARGUMENT;
    return h->tree.debug_apply_m2l_tables_host(M, L);
}

uint64_t bbfmm_debug_morton_encode(int32_t d, const uint64_t *anchor, uint64_t level) {
    if (d < 1 || d > 3 || !anchor) return 0;
    return bbfmm::encode_morton_point(anchor, level, d);
}
void bbfmm_debug_morton_decode(int32_t d, uint64_t key, uint64_t *anchor_out, uint64_t *level_out) {
    if (d < 1 || d > 3 || !anchor_out || !level_out) return;
    bbfmm::decode_key(key, d, anchor_out, level_out);
}
int32_t bbfmm_debug_morton_neighbours(int32_t d, uint64_t key, uint64_t *keys_out) {
    if (d < 1 || d > 3 || !keys_out) return -1;
    return bbfmm::get_neighbours(key, d, keys_out);
}
int32_t bbfmm_debug_direction_vectors(int32_t d, int32_t *out) {
    if (d < 1 || d > 3 || !out) return -1;
    const int(*dirs)[3];
    const int n = bbfmm::direction_vectors(d, &dirs);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < d; ++a) out[i * d + a] = dirs[i][a];
    return n;
}
int bbfmm_debug_reference_vectors(const bbfmm_handle *h, int32_t *out, int32_t *n_ref_out) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    const bbfmm::Operators &ops = h->tree.ops();
    if (n_ref_out) *n_ref_out = ops.n_ref;
    if (out) std::copy(ops.ref_vecs.begin(), ops.ref_vecs.end(), out);
    return BBFMM_OK;
}

int bbfmm_debug_m2l_variants(const bbfmm_handle *h, int64_t *n_variants, int64_t *n_cells) {
    if (!h) return BBFMM_BAD_ARGUMENT;
    int64_t nv = 0, nc = 0;
    h->tree.m2l_variant_stats(&nv, &nc);
    if (n_variants) *n_variants = nv;
    if (n_cells) *n_cells = nc;
    return BBFMM_OK;
}

int bbfmm_debug_get_coefficients(bbfmm_handle *h, char which, int32_t k, double *out) {
    GUARD(h) return h->tree.debug_get_coefficients(which, k, out);
    END_GUARD(h)
}

} // extern "C"

// ------------------------------------------------------------------ domain decomposition (host part)
#include "ddm.hpp"
struct bbfmm_ddm {
    bbfmm::DdmTree tree;
};
extern "C" {
void bbfmm_ddm_params_defaults(bbfmm_ddm_params *out) {
    if (!out) return;
    const bbfmm::DdmParams p;
    out->leaf_threshold = p.leaf_threshold;
    out->overlap_quota = p.overlap_quota;
    out->coarse_ratio = p.coarse_ratio;
    out->coarse_threshold = p.coarse_threshold;
}

void bbfmm_ddm_params_for_points(int64_t n, bbfmm_ddm_params *out) {
    if (!out) return;
    bbfmm_ddm_params_defaults(out);
    // per level at most N (1/8 + 1/341) points survive (rounding up per leaf, leaves of more than 341 points)
    out->coarse_threshold = std::max<int64_t>(out->coarse_threshold, n / 470 + 1);
}
int bbfmm_ddm_build(const double *points, int64_t n, int32_t d, int64_t ld, const bbfmm_ddm_params *params,
                    bbfmm_ddm **out) {
    if (!out) return BBFMM_BAD_ARGUMENT;
    *out = nullptr;
    bbfmm::DdmParams p;
    if (params) {
        p.leaf_threshold = params->leaf_threshold;
        p.overlap_quota = params->overlap_quota;
        p.coarse_ratio = params->coarse_ratio;
        p.coarse_threshold = params->coarse_threshold;
    }
    bbfmm_ddm *t = new (std::nothrow) bbfmm_ddm();
    if (!t) return BBFMM_DEVICE_ERROR;
    int rc = BBFMM_BAD_ARGUMENT;
    try {
        rc = bbfmm::build_ddm_tree(points, n, d, ld, p, &t->tree);
    } catch (...) {
        rc = BBFMM_DEVICE_ERROR;
    }
    if (rc != BBFMM_OK) {
        delete t;
        return rc;
    }
    *out = t;
    return BBFMM_OK;
}
void bbfmm_ddm_destroy(bbfmm_ddm *t) { delete t; }
int32_t bbfmm_ddm_num_levels(const bbfmm_ddm *t) { return t ? static_cast<int32_t>(t->tree.levels.size()) : 0; }
static const bbfmm::DdmLevel *ddm_level(const bbfmm_ddm *t, int32_t level) {
    if (!t || level < 0 || level >= static_cast<int32_t>(t->tree.levels.size())) return nullptr;
    return &t->tree.levels[static_cast<size_t>(level)];
}
int64_t bbfmm_ddm_level_size(const bbfmm_ddm *t, int32_t level) {
    const bbfmm::DdmLevel *l = ddm_level(t, level);
    return l ? static_cast<int64_t>(l->point_indices.size()) : -1;
}
int bbfmm_ddm_level_points(const bbfmm_ddm *t, int32_t level, int64_t *out) {
    const bbfmm::DdmLevel *l = ddm_level(t, level);
    if (!l || !out) return BBFMM_BAD_ARGUMENT;
    std::copy(l->point_indices.begin(), l->point_indices.end(), out);
    return BBFMM_OK;
}
int64_t bbfmm_ddm_num_domains(const bbfmm_ddm *t, int32_t level) {
    const bbfmm::DdmLevel *l = ddm_level(t, level);
    return l ? static_cast<int64_t>(l->leaves.size()) : -1;
}
int64_t bbfmm_ddm_domain_size(const bbfmm_ddm *t, int32_t level, int64_t domain) {
    const bbfmm::DdmLevel *l = ddm_level(t, level);
    if (!l || domain < 0 || domain >= static_cast<int64_t>(l->leaves.size())) return -1;
    return static_cast<int64_t>(l->leaves[static_cast<size_t>(domain)].idx.size());
}
int bbfmm_ddm_domain(const bbfmm_ddm *t, int32_t level, int64_t domain, int64_t *indices, uint8_t *internal,
                     double *extents) {
    const bbfmm::DdmLevel *l = ddm_level(t, level);
    if (!l || domain < 0 || domain >= static_cast<int64_t>(l->leaves.size())) return BBFMM_BAD_ARGUMENT;
    const bbfmm::DdmDomain &dm = l->leaves[static_cast<size_t>(domain)];
    if (indices) std::copy(dm.idx.begin(), dm.idx.end(), indices);
    if (internal) std::copy(dm.internal.begin(), dm.internal.end(), internal);
    if (extents) std::copy(dm.extents.begin(), dm.extents.end(), extents);
    return BBFMM_OK;
}
} // extern "C"


// DeviceGroup: one handle over several devices of one process.  See device_group.hpp.
#include "device_group.hpp"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "fmm_tree_impl.hpp"

namespace bbfmm {

#define GHIP(expr)                                              \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);     \
    } while (0)
// inside a per-part job: the job's return code
#define PHIP(expr)                                              \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return -static_cast<int>(e__);   \
    } while (0)

bool parse_device_list(const char *text, int n_devices, std::vector<int> *out, std::string *err) {
    out->clear();
    if (!text) return true;
    std::string s(text);
    size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
    if (a == std::string::npos) return true;
    s = s.substr(a, b - a + 1);
    if (s == "all" || s == "ALL") {
        for (int i = 0; i < n_devices; ++i) out->push_back(i);
        return true;
    }
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t comma = s.find(',', pos);
        if (comma == std::string::npos) comma = s.size();
        const std::string tok = s.substr(pos, comma - pos);
        char *end = nullptr;
        const long v = std::strtol(tok.c_str(), &end, 10);
        if (tok.empty() || !end || *end != '\0' || v < 0) {
            *err = "malformed device list '" + s + "' (expected ids separated by commas, or 'all')";
            out->clear();
            return false;
        }
        out->push_back(static_cast<int>(v));
        pos = comma + 1;
    }
    for (int v : *out) // (the syntax of the whole list first: a typo is reported as one even where no device is visible)
        if (n_devices >= 0 && v >= n_devices) {
            *err = "device " + std::to_string(v) + " of the device list '" + s + "' does not exist (" + std::to_string(n_devices) + " visible)";
            out->clear();
            return false;
        }
    if (out->size() > static_cast<size_t>(kMaxScatterParts)) {
        *err = "more parts in the device list than a handle takes (" + std::to_string(kMaxScatterParts) + ")";
        out->clear();
        return false;
    }
    return true;
}

int DeviceGroup::fail(int code, const std::string &msg) {
    err_ = msg;
    return code;
}
int DeviceGroup::hip_fail(hipError_t e, const char *what) {
    err_ = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " (device group)";
    return BBFMM_DEVICE_ERROR;
}
int DeviceGroup::part_fail(const Part &p, int rc) {
    const int g = static_cast<int>(&p - parts_.data());
    if (rc < 0) { // a HIP error of the group's own calls inside a part's job
        err_ = "part " + std::to_string(g) + " (device " + std::to_string(p.device) + "): HIP error: " +
               hipGetErrorString(static_cast<hipError_t>(-rc));
        return BBFMM_DEVICE_ERROR;
    }
    err_ = "part " + std::to_string(g) + " (device " + std::to_string(p.device) + "): " + p.t->last_error();
    return rc;
}

namespace {
// FmmTree::group_weights_resident_ for the length of one call (cleared again whatever way the call ends)
struct ResidentWeights {
    bool &flag;
    explicit ResidentWeights(bool &f) : flag(f) { flag = true; }
    ~ResidentWeights() { flag = false; }
};
} // namespace

template <class F> int DeviceGroup::for_parts(F &&fn) {
    const int G = n_parts();
    std::vector<int> rcs(static_cast<size_t>(G), BBFMM_OK);
    if (!threads_ || G == 1) {
        for (int g = 0; g < G; ++g) {
            parts_[static_cast<size_t>(g)].t->bind_device();
            rcs[static_cast<size_t>(g)] = fn(g);
        }
    } else { // one host thread per part queues its kernels: the devices start together
        parallel_for(G, 1, [&](int64_t g) {
            parts_[static_cast<size_t>(g)].t->bind_device();
            rcs[static_cast<size_t>(g)] = fn(static_cast<int>(g));
        });
    }
    parts_[0].t->bind_device();
    for (int g = 0; g < G; ++g)
        if (rcs[static_cast<size_t>(g)] != BBFMM_OK) return part_fail(parts_[static_cast<size_t>(g)], rcs[static_cast<size_t>(g)]);
    return BBFMM_OK;
}

void DeviceGroup::free_buffers() {
    for (Part &p : parts_) {
        (void)hipSetDevice(p.device);
        for (double **b : {&p.d_send, &p.d_slots, &p.d_sum, &p.d_seg}) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
        }
    }
    if (!parts_.empty()) {
        (void)hipSetDevice(parts_[0].device);
        if (d_all_) (void)hipFree(d_all_);
        d_all_ = nullptr;
    }
    k_cap_ = 0;
    have_blocks_ = false;
}

DeviceGroup::~DeviceGroup() {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (Part &p : parts_) {
        (void)hipSetDevice(p.device);
        if (p.t && p.t->stream_) (void)hipStreamSynchronize(p.t->stream_);
        if (p.comm) (void)hipStreamSynchronize(p.comm);
    }
    free_buffers();
    for (Part &p : parts_) {
        (void)hipSetDevice(p.device);
        for (hipEvent_t *e : {&p.ev_w, &p.ev_up, &p.ev_sent, &p.ev_sum, &p.ev_done})
            if (*e) (void)hipEventDestroy(*e);
        if (p.comm) (void)hipStreamDestroy(p.comm);
        p.own.reset(); // (the primary belongs to the handle)
    }
    if (!parts_.empty()) {
        (void)hipSetDevice(parts_[0].device);
        if (ev_in_) (void)hipEventDestroy(ev_in_);
        if (parts_[0].t) {
            parts_[0].t->mirrors_.clear();
            parts_[0].t->group_primary_ = false;
        }
    }
    if (have_cur) (void)hipSetDevice(cur);
}

int DeviceGroup::init(FmmTree *primary, const GroupCreateArgs &a, const std::vector<int> &devices) {
    const int G = static_cast<int>(devices.size());
    if (G < 2) return fail(BBFMM_BAD_ARGUMENT, "a device group needs at least two parts");
    if (G > kMaxScatterParts) return fail(BBFMM_BAD_ARGUMENT, "more parts than a handle takes");
    if (primary->host_only()) return fail(BBFMM_BAD_ARGUMENT, "a device group needs devices (BBFMM_FLAG_HOST_ONLY is set)");
    {
        const char *e = std::getenv("BBFMM_GROUP_THREADS"); // 0: the calling thread queues every part's work (checker)
        threads_ = !e || std::atoi(e) != 0;
    }
    if (const char *e = std::getenv("BBFMM_GROUP_SHARD_MIN")) shard_min_rows_ = std::max<int64_t>(1, std::atoll(e));
    parts_.resize(static_cast<size_t>(G));
    parts_[0].t = primary;
    bool distinct = true;
    for (int g = 0; g < G; ++g) {
        Part &p = parts_[static_cast<size_t>(g)];
        p.device = devices[static_cast<size_t>(g)];
        p.owner = g;
        for (int h = 0; h < g; ++h)
            if (parts_[static_cast<size_t>(h)].device == p.device) {
                p.owner = parts_[static_cast<size_t>(h)].owner;
                distinct = false;
                break;
            }
    }
    if (primary->device() != parts_[0].device) return fail(BBFMM_BAD_ARGUMENT, "the primary was not created on the first device of the list");
    // the other parts: the same tree from the same points on their devices (side by side when the devices differ)
    std::vector<int> rcs(static_cast<size_t>(G), BBFMM_OK);
    auto create_part = [&](int g) {
        Part &p = parts_[static_cast<size_t>(g)];
        if (hipSetDevice(p.device) != hipSuccess) {
            rcs[static_cast<size_t>(g)] = BBFMM_DEVICE_ERROR;
            return;
        }
        p.own.reset(new FmmTree());
        p.t = p.own.get();
        try {
            rcs[static_cast<size_t>(g)] = p.t->create(a.pts, a.n, a.d, a.ld, a.order, a.kernel_type, a.base_range, a.total_sill, a.adaptive,
                                                      a.sparse, a.extents, a.params, a.flags);
        } catch (...) {
            rcs[static_cast<size_t>(g)] = BBFMM_BAD_ARGUMENT;
        }
    };
    if (distinct && threads_) {
        std::vector<std::thread> th;
        for (int g = 1; g < G; ++g) th.emplace_back(create_part, g);
        for (auto &t : th) t.join();
    } else {
        for (int g = 1; g < G; ++g) create_part(g);
    }
    for (int g = 1; g < G; ++g)
        if (rcs[static_cast<size_t>(g)] != BBFMM_OK) {
            const Part &p = parts_[static_cast<size_t>(g)];
            err_ = "part " + std::to_string(g) + " (device " + std::to_string(p.device) + "): " + (p.t ? p.t->last_error() : "not created");
            return rcs[static_cast<size_t>(g)];
        }
    n_ = primary->tree().n_points;
    for (int g = 0; g < G; ++g) {
        Part &p = parts_[static_cast<size_t>(g)];
        p.t->bind_device();
        if (p.t->tree().n_cells() != primary->tree().n_cells() || p.t->tree().n_points != n_)
            return fail(BBFMM_DEVICE_ERROR, "the parts of a device group built different trees");
        const int rc = p.t->set_partition(g, G);
        if (rc != BBFMM_OK) return part_fail(p, rc);
        if (g > 0 && p.t->partition_bounds() != primary->partition_bounds())
            return fail(BBFMM_DEVICE_ERROR, "the parts of a device group cut the sorted points differently");
        GHIP(hipStreamCreateWithFlags(&p.comm, hipStreamNonBlocking));
        for (hipEvent_t *e : {&p.ev_w, &p.ev_up, &p.ev_sent, &p.ev_sum, &p.ev_done}) GHIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    bounds_ = primary->partition_bounds();
    cnt_ = primary->partition_coarse_count();
    m_max_ = 1;
    for (int g = 0; g < G; ++g) {
        Part &p = parts_[static_cast<size_t>(g)];
        p.pb = bounds_[static_cast<size_t>(g)];
        p.m = bounds_[static_cast<size_t>(g) + 1] - p.pb;
        m_max_ = std::max(m_max_, p.m);
        if (p.t->partition_coarse_count() != cnt_) return fail(BBFMM_DEVICE_ERROR, "the parts of a device group exchange different prefixes");
    }
    // peer access between the distinct devices (the copies fall back to staged ones where it is refused)
    for (int g = 0; g < G; ++g)
        for (int h = 0; h < G; ++h) {
            const int da = parts_[static_cast<size_t>(g)].device, db = parts_[static_cast<size_t>(h)].device;
            if (da == db || parts_[static_cast<size_t>(g)].owner != g || parts_[static_cast<size_t>(h)].owner != h) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                (void)hipSetDevice(da);
                const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e != hipSuccess) (void)hipGetLastError(); // already enabled, or refused
            }
        }
    primary->bind_device();
    GHIP(hipEventCreateWithFlags(&ev_in_, hipEventDisableTiming));
    inv_order_.resize(static_cast<size_t>(n_));
    {
        const auto &order = primary->tree().order;
        int32_t *inv = inv_order_.data();
        parallel_for_chunks(n_, int64_t(1) << 16, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) inv[order[static_cast<size_t>(i)]] = static_cast<int32_t>(i);
        });
    }
    primary->group_primary_ = true;
    if (std::getenv("BBFMM_VERBOSE")) {
        std::fprintf(stderr, "[bbfmm] device group: %d parts on devices", G);
        for (const Part &p : parts_) std::fprintf(stderr, " %d", p.device);
        std::fprintf(stderr, "; %lld coarse doubles exchanged per rhs; rows per part", static_cast<long long>(cnt_));
        for (const Part &p : parts_) std::fprintf(stderr, " %lld", static_cast<long long>(p.m));
        std::fprintf(stderr, "\n");
    }
    return BBFMM_OK;
}

// Exchange buffers for k right-hand sides (and, for device callers, the blocks of the potentials).
int DeviceGroup::ensure_capacity(int k, bool device_blocks) {
    if (k <= k_cap_ && (!device_blocks || have_blocks_)) return BBFMM_OK;
    const int kk = std::max(k, k_cap_);
    const int G = n_parts();
    for (Part &p : parts_) { // nothing of an earlier product may still be reading the old buffers
        (void)hipSetDevice(p.device);
        GHIP(hipStreamSynchronize(p.t->stream_));
        GHIP(hipStreamSynchronize(p.comm));
    }
    const bool blocks = device_blocks || have_blocks_;
    free_buffers();
    const size_t len = static_cast<size_t>(kk) * static_cast<size_t>(std::max<int64_t>(cnt_, 1));
    for (Part &p : parts_) {
        GHIP(hipSetDevice(p.device));
        GHIP(hipMalloc(reinterpret_cast<void **>(&p.d_send), len * sizeof(double)));
        GHIP(hipMalloc(reinterpret_cast<void **>(&p.d_slots), len * static_cast<size_t>(G) * sizeof(double)));
        GHIP(hipMalloc(reinterpret_cast<void **>(&p.d_sum), len * sizeof(double)));
        if (blocks) GHIP(hipMalloc(reinterpret_cast<void **>(&p.d_seg), static_cast<size_t>(kk) * m_max_ * sizeof(double)));
    }
    GHIP(hipSetDevice(parts_[0].device));
    if (blocks) GHIP(hipMalloc(reinterpret_cast<void **>(&d_all_), static_cast<size_t>(G) * kk * m_max_ * sizeof(double)));
    k_cap_ = kk;
    have_blocks_ = blocks;
    return BBFMM_OK;
}

bool DeviceGroup::weights_match_staged(const double *w, int64_t rows, int k, int64_t ldw) const {
    const FmmTree &P = *parts_[0].t;
    if (!w || staged_k_ < 1 || staged_k_ != k || rows < n_ || ldw < rows || !P.h_pin_) return false;
    std::atomic<bool> same{true};
    for (int j = 0; j < k && same.load(std::memory_order_relaxed); ++j) {
        const double *a = w + static_cast<size_t>(j) * ldw, *b = P.h_pin_ + static_cast<size_t>(j) * n_;
        parallel_for_chunks(n_, FmmTree::kHostPiece, [&](int64_t lo, int64_t hi) {
            if (same.load(std::memory_order_relaxed) && std::memcmp(a + lo, b + lo, static_cast<size_t>(hi - lo) * sizeof(double)) != 0)
                same.store(false, std::memory_order_relaxed);
        });
    }
    return same.load();
}

// The caller's weights -> the primary's pinned buffer -> every device of the group (piece by piece, each over its own link).
int DeviceGroup::stage(const double *w, int64_t rows, int k, int64_t ldw) {
    (void)rows;
    FmmTree &P = *parts_[0].t;
    staged_k_ = dev_staged_k_ = 0;
    all_complete_ = all_locals_ = false;
    // parts that share a device read its owner's copy: the new weights wait until the last product's parts have read the old
    for (size_t g = 0; g < parts_.size(); ++g) {
        Part &p = parts_[g];
        if (static_cast<size_t>(p.owner) == g) continue;
        Part &o = parts_[static_cast<size_t>(p.owner)];
        o.t->bind_device();
        GHIP(hipStreamWaitEvent(o.t->stream_, p.ev_up, 0));
    }
    P.mirrors_.clear();
    for (size_t g = 1; g < parts_.size(); ++g) {
        Part &p = parts_[g];
        if (static_cast<size_t>(p.owner) != g) continue;
        p.t->bind_device();
        const int rc = p.t->ensure_w_in(k);
        if (rc != BBFMM_OK) return part_fail(p, rc);
        P.mirrors_.push_back(FmmTree::WeightMirror{p.device, p.t->stream_, p.t->d_w_in_.p});
    }
    P.bind_device();
    {
        const int rc = P.stage_weights_to_device(w, n_, k, ldw);
        if (rc != BBFMM_OK) return part_fail(parts_[0], rc);
    }
    for (size_t g = 0; g < parts_.size(); ++g) {
        Part &p = parts_[g];
        if (static_cast<size_t>(p.owner) != g) continue;
        p.t->bind_device();
        GHIP(hipEventRecord(p.ev_w, p.t->stream_));
    }
    P.bind_device();
    staged_k_ = dev_staged_k_ = k;
    return BBFMM_OK;
}

int DeviceGroup::run_upward(int k, const double *d_w_primary, int64_t ld_primary, bool near_field) {
    const int G = n_parts();
    const size_t len = static_cast<size_t>(k) * static_cast<size_t>(cnt_);
    primary_complete_ = all_complete_ = all_locals_ = false; // (the parts' coarse multipoles become partial sums, L the restricted pass's)
    pending_k_ = 0;
    int rc = for_parts([&](int g) -> int {
        Part &p = parts_[static_cast<size_t>(g)];
        FmmTree &t = *p.t;
        const Part &o = parts_[static_cast<size_t>(p.owner)];
        PHIP(hipStreamWaitEvent(t.stream_, p.ev_sent, 0)); // the last round's copies out of d_send
        const double *dw = o.t->d_w_in_.p;
        int64_t ld = n_;
        if (d_w_primary && p.device == parts_[0].device) { // the caller's device buffer serves the parts on its device
            dw = d_w_primary;
            ld = ld_primary;
            if (g != 0) PHIP(hipStreamWaitEvent(t.stream_, ev_in_, 0));
        } else if (p.owner != g) {
            PHIP(hipStreamWaitEvent(t.stream_, o.ev_w, 0));
        }
        const int prc = t.matvec_partition_upward(dw, ld, k, p.d_send, p.comm, near_field);
        if (prc != BBFMM_OK) return prc;
        PHIP(hipEventRecord(p.ev_up, t.stream_));
        if (len > 0) {
            for (int h = 0; h < G; ++h) PHIP(hipStreamWaitEvent(p.comm, parts_[static_cast<size_t>(h)].ev_sum, 0)); // slots free again
            for (int h = 0; h < G; ++h) {
                const Part &q = parts_[static_cast<size_t>(h)];
                double *dst = q.d_slots + static_cast<size_t>(g) * len;
                if (q.device == p.device)
                    PHIP(hipMemcpyAsync(dst, p.d_send, len * sizeof(double), hipMemcpyDeviceToDevice, p.comm));
                else
                    PHIP(hipMemcpyPeerAsync(dst, q.device, p.d_send, p.device, len * sizeof(double), p.comm));
            }
        }
        PHIP(hipEventRecord(p.ev_sent, p.comm));
        return BBFMM_OK;
    });
    if (rc != BBFMM_OK) return rc;
    if (len > 0) {
        rc = for_parts([&](int h) -> int {
            Part &p = parts_[static_cast<size_t>(h)];
            for (int g = 0; g < G; ++g) PHIP(hipStreamWaitEvent(p.comm, parts_[static_cast<size_t>(g)].ev_sent, 0));
            launch_sum_slots(p.d_slots, G, static_cast<int64_t>(len), p.d_sum, p.comm);
            PHIP(hipGetLastError());
            PHIP(hipEventRecord(p.ev_sum, p.comm));
            return BBFMM_OK;
        });
        if (rc != BBFMM_OK) return rc;
    }
    pending_k_ = near_field ? k : -k; // (negative: the owned targets' near field was not queued -- a partial product's upward pass)
    return BBFMM_OK;
}

// An index set dealt to the parts: row r belongs to the part that owns its sorted position.  Cached (8 sets, least
// recently used evicted): the Schwarz sweep asks for the same five sets in every apply.
int DeviceGroup::subset_split(const int64_t *idx, int64_t n_idx, SubsetSplit **out) {
    FmmTree &P = *parts_[0].t;
    const uint64_t key = P.subset_key(idx, n_idx);
    ++split_clock_;
    for (auto &sp : splits_)
        if (sp->key == key && static_cast<int64_t>(sp->idx.size()) == n_idx &&
            (n_idx == 0 || std::memcmp(sp->idx.data(), idx, static_cast<size_t>(n_idx) * sizeof(int64_t)) == 0)) {
            sp->last_use = split_clock_;
            *out = sp.get();
            return BBFMM_OK;
        }
    for (int64_t j = 0; j < n_idx; ++j)
        if (idx[j] < 0 || idx[j] >= n_) return fail(BBFMM_BAD_ARGUMENT, "target index out of range");
    if (splits_.size() >= 8) {
        size_t victim = 0;
        for (size_t i = 1; i < splits_.size(); ++i)
            if (splits_[i]->last_use < splits_[victim]->last_use) victim = i;
        splits_.erase(splits_.begin() + static_cast<std::ptrdiff_t>(victim));
    }
    const int G = n_parts();
    std::unique_ptr<SubsetSplit> sp(new SubsetSplit());
    sp->key = key;
    sp->last_use = split_clock_;
    sp->idx.assign(idx, idx + n_idx);
    sp->rows.resize(static_cast<size_t>(G));
    sp->where.resize(static_cast<size_t>(G));
    for (int64_t j = 0; j < n_idx; ++j) {
        const int64_t pos = inv_order_[static_cast<size_t>(idx[j])];
        const int g = static_cast<int>(std::upper_bound(bounds_.begin() + 1, bounds_.end() - 1, pos) - (bounds_.begin() + 1));
        sp->rows[static_cast<size_t>(g)].push_back(idx[j]);
        sp->where[static_cast<size_t>(g)].push_back(j);
    }
    sp->offset.assign(static_cast<size_t>(G) + 1, 0);
    for (int g = 0; g < G; ++g) sp->offset[static_cast<size_t>(g) + 1] = sp->offset[static_cast<size_t>(g)] + static_cast<int64_t>(sp->rows[static_cast<size_t>(g)].size());
    *out = sp.get();
    splits_.push_back(std::move(sp));
    return BBFMM_OK;
}

// The parts' shares of a partial product behind the staged weights: upward + exchange unless one is pending, then every
// part's restricted passes on its rows of the set; consume(j, row, value) for every row once all blocks are home.
template <class F> int DeviceGroup::subset_product(SubsetSplit *sp, bool upward_pending, F &&consume) {
    FmmTree &P = *parts_[0].t;
    if (!upward_pending) CHK(run_upward(1, nullptr, 0, false));
    double *h_out = P.h_pin_ + n_; // (behind the staged weights; sized 2 N by the staging)
    const int rc = for_parts([&](int g) -> int {
        Part &p = parts_[static_cast<size_t>(g)];
        const std::vector<int64_t> &r = sp->rows[static_cast<size_t>(g)];
        const int prc = p.t->partition_subset_finish(p.d_sum, r.data(), static_cast<int64_t>(r.size()), h_out + sp->offset[static_cast<size_t>(g)], p.comm);
        if (prc != BBFMM_OK) return prc;
        PHIP(hipEventRecord(p.ev_done, p.t->stream_));
        return BBFMM_OK;
    });
    pending_k_ = 0;
    if (rc != BBFMM_OK) return rc;
    for (Part &p : parts_) {
        p.t->bind_device();
        GHIP(hipEventSynchronize(p.ev_done));
    }
    P.bind_device();
    const int G = n_parts();
    for (int g = 0; g < G; ++g) {
        const std::vector<int64_t> &r = sp->rows[static_cast<size_t>(g)], &wh = sp->where[static_cast<size_t>(g)];
        const double *h = h_out + sp->offset[static_cast<size_t>(g)];
        parallel_for_chunks(static_cast<int64_t>(r.size()), int64_t(1) << 15, [&](int64_t b, int64_t e) {
            for (int64_t j = b; j < e; ++j) consume(wh[static_cast<size_t>(j)], r[static_cast<size_t>(j)], h[j]);
        });
    }
    return BBFMM_OK;
}

int DeviceGroup::fast_matvec_subset(const double *w, int64_t rows, int64_t basis_size, const int64_t *idx, int64_t n_idx, const double *poly,
                                    int64_t ldp, double nugget, double *result) {
    const int64_t N = n_;
    if (!w || !result || basis_size < 0 || rows != N + basis_size) return fail(BBFMM_BAD_ARGUMENT, "weights must have N + basis_size rows");
    if (poly && ldp < N) return fail(BBFMM_BAD_ARGUMENT, "polynomial matrix needs N rows");
    if (!idx || n_idx < 0) return fail(BBFMM_BAD_ARGUMENT, "bad target index array");
    last_path_ = 0;
    if (static_cast<size_t>(2) * N > FmmTree::kMaxPinnedDoubles) return fail(BBFMM_UNSUPPORTED, "too many points for the pinned mirror of a device group");
    SubsetSplit *sp = nullptr;
    CHK(subset_split(idx, n_idx, &sp));
    std::fill(result, result + rows, 0.0); // rbf.rs:1346
    if (n_idx == 0) return BBFMM_OK;
    CHK(ensure_capacity(1, false));
    CHK(stage(w, N, 1, N));
    CHK(subset_product(sp, false, [&](int64_t, int64_t i, double v) { // rbf.rs:1366-1376
        double r = v + w[i] * nugget;
        if (poly) {
            double sacc = 0.0;
            for (int64_t q = 0; q < basis_size; ++q) sacc += poly[q * ldp + i] * w[N + q];
            r += sacc;
        }
        result[i] = r;
    }));
    last_path_ = 2;
    return BBFMM_OK;
}

int DeviceGroup::evaluate_rows_of_sources(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx, double *out,
                                          bool *handled) {
    *handled = false;
    FmmTree &P = *parts_[0].t;
    last_path_ = 0;
    // (the conditions under which a single handle looks its targets up: a solver's tree, one rhs, N / 2048 .. N / 2 rows)
    if (!P.solver_tree_ || k != 1 || staged_k_ != 1 || !w || !x || !out || ldx < m || m >= n_ || m < std::max<int64_t>(1024, n_ / 2048) ||
        m > n_ / 2 || n_ > (int64_t(1) << 26) || all_locals_ || P.locals_requested_)
        return BBFMM_OK;
    if (!weights_match_staged(w, rows, 1, ldw)) return BBFMM_OK;
    std::vector<int64_t> rows_of;
    if (!P.targets_are_rows_of_sources(x, m, ldx, &rows_of)) return BBFMM_OK;
    const uint64_t key = P.subset_key(rows_of.data(), m);
    bool known = key == last_rows_miss_;
    for (const auto &sp : splits_) known = known || (sp->key == key && static_cast<int64_t>(sp->idx.size()) == m);
    if (!known) { // first sighting of the set: no plans for a caller that may never come back (as on one device)
        last_rows_miss_ = key;
        return BBFMM_OK;
    }
    SubsetSplit *sp = nullptr;
    CHK(subset_split(rows_of.data(), m, &sp));
    CHK(subset_product(sp, pending_k_ == 1 || pending_k_ == -1, [&](int64_t j, int64_t, double v) { out[j] = v; }));
    *handled = true;
    last_path_ = 2;
    return BBFMM_OK;
}

// Second half on every part; the owned blocks land in the primary's pinned buffer (sorted order, k rows of N) and
// consume(begin, end, h_sorted) is called on row chunks by the host threads once all of them are there.
template <class F> int DeviceGroup::finish_to_host(int k, F &&consume) {
    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    FmmTree &P = *parts_[0].t;
    double *h_sorted = P.h_pin_ + static_cast<size_t>(k) * n_; // behind the staged weights (sized 2 k N by the staging)
    int rc = for_parts([&](int g) -> int {
        Part &p = parts_[static_cast<size_t>(g)];
        const int prc = p.t->matvec_partition_finish_host(p.d_sum, h_sorted + p.pb, n_, p.comm);
        if (prc != BBFMM_OK) return prc;
        PHIP(hipEventRecord(p.ev_done, p.t->stream_));
        return BBFMM_OK;
    });
    pending_k_ = 0;
    if (rc != BBFMM_OK) return rc;
    for (Part &p : parts_) {
        p.t->bind_device();
        GHIP(hipEventSynchronize(p.ev_done));
    }
    P.bind_device();
    const auto t_1 = std::chrono::steady_clock::now();
    parallel_for_chunks(n_, int64_t(1) << 16, [&](int64_t b, int64_t e) { consume(b, e, h_sorted); });
    if (verbose) {
        const auto t_2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm] device group: downward + leaf passes and the blocks' way back %.3f ms, rows written by the host threads %.3f ms\n",
                     std::chrono::duration<double, std::milli>(t_1 - t_0).count(), std::chrono::duration<double, std::milli>(t_2 - t_1).count());
    }
    return BBFMM_OK;
}

int DeviceGroup::set_weights(const double *w, int64_t rows, int k, int64_t ldw) {
    FmmTree &P = *parts_[0].t;
    if (!w || rows < n_ || ldw < rows || k < 1) return fail(BBFMM_BAD_ARGUMENT, "weights must be rows x k with rows >= N");
    last_path_ = 0;
    if (static_cast<size_t>(2) * k * n_ > FmmTree::kMaxPinnedDoubles) { // too large for the pinned mirror: the primary alone
        P.bind_device();
        const int rc = P.set_weights(w, rows, k, ldw);
        primary_state_changed();
        return rc == BBFMM_OK ? rc : part_fail(parts_[0], rc);
    }
    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    CHK(ensure_capacity(k, false));
    CHK(stage(w, rows, k, ldw));
    const auto t_1 = std::chrono::steady_clock::now();
    const int rc = run_upward(k, nullptr, 0);
    if (verbose) {
        const auto t_2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm] device group: set_weights staged the weights in %.3f ms, queued the upward passes and the exchange in %.3f ms\n",
                     std::chrono::duration<double, std::milli>(t_1 - t_0).count(), std::chrono::duration<double, std::milli>(t_2 - t_1).count());
    }
    return rc;
}

int DeviceGroup::evaluate_at_sources(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx,
                                     double *out, int64_t ldo, bool *handled) {
    *handled = false;
    static const bool sources_fast = [] {
        const char *e = std::getenv("BBFMM_EVAL_SOURCES_FAST"); // 0: always the general path on the primary (checker)
        return !e || std::atoi(e) != 0;
    }();
    last_path_ = 0;
    FmmTree &P = *parts_[0].t;
    if (!sources_fast || staged_k_ < 1 || k != staged_k_ || m != n_ || !x || !out || ldx < m || ldo < m) return BBFMM_OK;
    if (all_locals_ || P.locals_requested_) return BBFMM_OK; // Leaves mode: the partitioned downward pass would overwrite the stored expansions
    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    if (!P.targets_are_sources(x, m, ldx)) return BBFMM_OK;
    if (!weights_match_staged(w, rows, k, ldw)) return BBFMM_OK; // other weights than set_weights': the primary's mixture
    if (verbose)
        std::fprintf(stderr, "[bbfmm] device group: targets and weights compared with the sources / the staged weights in %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count());
    if (pending_k_ != k) CHK(run_upward(k, nullptr, 0));         // (a second evaluate behind one set_weights)
    const int32_t *inv = inv_order_.data();
    const int64_t N = n_;
    CHK(finish_to_host(k, [&](int64_t b, int64_t e, const double *h_sorted) {
        for (int j = 0; j < k; ++j) {
            const double *src = h_sorted + static_cast<size_t>(j) * N;
            double *dst = out + static_cast<size_t>(j) * ldo;
            for (int64_t r = b; r < e; ++r) dst[r] = src[inv[r]];
        }
    }));
    *handled = true;
    last_path_ = 1;
    return BBFMM_OK;
}

// All rows (the FGMRES matvec, rbf.rs:105-117, 1338-1379): set_weights + evaluate at the sources + the nugget and
// polynomial terms, the latter by the host threads while they write the rows.
int DeviceGroup::fast_matrix_vector_product(const double *w, int64_t rows, int64_t basis_size, const double *poly, int64_t ldp, double nugget,
                                            double *result) {
    const int64_t N = n_;
    if (!w || !result || basis_size < 0 || rows != N + basis_size) return fail(BBFMM_BAD_ARGUMENT, "weights must have N + basis_size rows");
    if (poly && ldp < N) return fail(BBFMM_BAD_ARGUMENT, "polynomial matrix needs N rows");
    last_path_ = 0;
    if (static_cast<size_t>(2) * N > FmmTree::kMaxPinnedDoubles) return fail(BBFMM_UNSUPPORTED, "too many points for the pinned mirror of a device group");
    CHK(ensure_capacity(1, false));
    CHK(stage(w, N, 1, N));
    CHK(run_upward(1, nullptr, 0));
    const int32_t *inv = inv_order_.data();
    CHK(finish_to_host(1, [&](int64_t b, int64_t e, const double *h_sorted) {
        for (int64_t i = b; i < e; ++i) {
            double v = h_sorted[inv[i]] + w[i] * nugget;
            if (poly) {
                double sacc = 0.0;
                for (int64_t q = 0; q < basis_size; ++q) sacc += poly[q * ldp + i] * w[N + q];
                v += sacc;
            }
            result[i] = v;
        }
    }));
    std::fill(result + N, result + rows, 0.0); // rbf.rs:1346
    last_path_ = 1;
    return BBFMM_OK;
}

// Device-resident vectors on the primary's device: the weights reach the other devices by peer copies, the blocks of
// the potentials come back the same way and one pass over the permutation writes the rows (asynchronous on the
// primary's stream unless sync).
int DeviceGroup::matvec_device(const double *d_w, int64_t ldw, int k, double *d_out, int64_t ldo, bool sync) {
    FmmTree &P = *parts_[0].t;
    const int64_t N = n_;
    const int G = n_parts();
    if (!d_w || !d_out || k < 1 || ldw < N || ldo < N) return fail(BBFMM_BAD_ARGUMENT, "bad device matvec arguments");
    last_path_ = 0;
    CHK(ensure_capacity(k, true));
    staged_k_ = dev_staged_k_ = 0; // the owners' staging buffers are overwritten with the caller's device weights
    all_complete_ = all_locals_ = false;
    P.bind_device();
    GHIP(hipEventRecord(ev_in_, P.stream_));
    // The primary keeps a copy as well (its parts read the caller's buffer itself): the partitioned upward pass leaves it with
    // the partial sums of its own subtree, and a later call that it serves alone -- evaluate at other targets without another
    // set_weights, as after bbfmm_matvec_device on a plain handle -- completes its multipoles from this copy.
    {
        const int rc = P.ensure_w_in(k);
        if (rc != BBFMM_OK) return part_fail(parts_[0], rc);
    }
    for (size_t h = 1; h < parts_.size(); ++h) // (parts sharing the primary's device may still be reading the old copy)
        if (parts_[h].owner == 0) GHIP(hipStreamWaitEvent(P.stream_, parts_[h].ev_up, 0));
    for (int j = 0; j < k; ++j)
        GHIP(hipMemcpyAsync(P.d_w_in_.p + static_cast<size_t>(j) * N, d_w + static_cast<size_t>(j) * ldw, static_cast<size_t>(N) * sizeof(double),
                            hipMemcpyDeviceToDevice, P.stream_));
    GHIP(hipEventRecord(parts_[0].ev_w, P.stream_));
    for (int g = 1; g < G; ++g) {
        Part &p = parts_[static_cast<size_t>(g)];
        if (p.owner != g) continue;
        p.t->bind_device();
        {
            const int rc = p.t->ensure_w_in(k);
            if (rc != BBFMM_OK) return part_fail(p, rc);
        }
        GHIP(hipStreamWaitEvent(p.t->stream_, ev_in_, 0));
        for (size_t h = 0; h < parts_.size(); ++h) // (parts sharing this device may still be reading the old copy)
            if (parts_[h].owner == g && static_cast<int>(h) != g) GHIP(hipStreamWaitEvent(p.t->stream_, parts_[h].ev_up, 0));
        for (int j = 0; j < k; ++j)
            GHIP(hipMemcpyPeerAsync(p.t->d_w_in_.p + static_cast<size_t>(j) * N, p.device, d_w + static_cast<size_t>(j) * ldw, parts_[0].device,
                                    static_cast<size_t>(N) * sizeof(double), p.t->stream_));
        GHIP(hipEventRecord(p.ev_w, p.t->stream_));
    }
    P.bind_device();
    CHK(run_upward(k, d_w, ldw));
    dev_staged_k_ = k;
    const int64_t m_max = m_max_;
    int rc = for_parts([&](int g) -> int {
        Part &p = parts_[static_cast<size_t>(g)];
        const int prc = p.t->matvec_partition_finish_sorted(p.d_sum, p.d_seg, m_max, p.comm);
        if (prc != BBFMM_OK) return prc;
        double *dst = d_all_ + static_cast<size_t>(g) * k * m_max;
        const size_t bytes = static_cast<size_t>(k) * m_max * sizeof(double);
        if (p.device == parts_[0].device)
            PHIP(hipMemcpyAsync(dst, p.d_seg, bytes, hipMemcpyDeviceToDevice, p.t->stream_));
        else
            PHIP(hipMemcpyPeerAsync(dst, parts_[0].device, p.d_seg, p.device, bytes, p.t->stream_));
        PHIP(hipEventRecord(p.ev_done, p.t->stream_));
        return BBFMM_OK;
    });
    pending_k_ = 0;
    if (rc != BBFMM_OK) return rc;
    P.bind_device();
    for (int g = 1; g < G; ++g) GHIP(hipStreamWaitEvent(P.stream_, parts_[static_cast<size_t>(g)].ev_done, 0));
    {
        const int prc = P.partition_scatter(d_all_, 0, G, m_max, k, d_out, ldo);
        if (prc != BBFMM_OK) return part_fail(parts_[0], prc);
    }
    last_path_ = 1;
    if (sync) GHIP(hipStreamSynchronize(P.stream_));
    return BBFMM_OK;
}

// The whole upward pass on every part, each from its device's copy of the staged weights.
int DeviceGroup::complete_all(int k) {
    if (all_complete_) return BBFMM_OK;
    const int rc = for_parts([&](int g) -> int {
        Part &p = parts_[static_cast<size_t>(g)];
        const Part &o = parts_[static_cast<size_t>(p.owner)];
        if (p.owner != g) PHIP(hipStreamWaitEvent(p.t->stream_, o.ev_w, 0));
        const int prc = p.t->complete_upward_from(o.t->d_w_in_.p, k);
        if (prc != BBFMM_OK) return prc;
        PHIP(hipEventRecord(p.ev_up, p.t->stream_)); // (the part has read the staged copy)
        return BBFMM_OK;
    });
    if (rc != BBFMM_OK) return rc;
    parts_[0].t->pin_w_k_ = staged_k_;
    all_complete_ = primary_complete_ = true;
    pending_k_ = 0;
    return BBFMM_OK;
}

int DeviceGroup::evaluate_sharded(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx, double *out,
                                  int64_t ldo, double *grad, int64_t ldg, bool with_grads, bool leaves_only, int64_t *bad_point_index,
                                  bool *handled) {
    *handled = false;
    const int G = n_parts();
    last_path_ = 0;
    if (staged_k_ < 1 || k != staged_k_ || m < shard_min_rows_ * G || !x || !out || ldx < m || ldo < m) return BBFMM_OK;
    if (with_grads && (!grad || ldg < m)) return BBFMM_OK;
    if (w ? !weights_match_staged(w, rows, k, ldw) : !leaves_only) return BBFMM_OK; // other weights: the primary's mixture
    if (leaves_only && !all_locals_) return BBFMM_OK;
    if (!leaves_only) CHK(complete_all(k));
    std::vector<int64_t> bad(static_cast<size_t>(G), -1);
    std::vector<int> prc(static_cast<size_t>(G), BBFMM_OK);
    const int rc = for_parts([&](int g) -> int {
        FmmTree &t = *parts_[static_cast<size_t>(g)].t;
        const int64_t r0 = m * g / G, r1 = m * (g + 1) / G;
        const ResidentWeights resident(t.group_weights_resident_);
        prc[static_cast<size_t>(g)] = t.evaluate(w, rows, k, ldw, x + r0, r1 - r0, ldx, out + r0, ldo, grad ? grad + r0 : nullptr, ldg, with_grads,
                                                 leaves_only, &bad[static_cast<size_t>(g)]);
        return BBFMM_OK; // (the parts' verdicts are combined below: the first offending row of the whole call wins)
    });
    if (rc != BBFMM_OK) return rc;
    for (int g = 0; g < G; ++g) { // parts in row order: the first failure is the one the reference would report
        const int e = prc[static_cast<size_t>(g)];
        if (e == BBFMM_OK) continue;
        if (e == BBFMM_POINT_OUTSIDE_TREE && bad_point_index) *bad_point_index = m * g / G + bad[static_cast<size_t>(g)];
        *handled = true;
        return part_fail(parts_[static_cast<size_t>(g)], e);
    }
    *handled = true;
    last_path_ = 3;
    return BBFMM_OK;
}

int DeviceGroup::set_local_coefficients_all(const double *w, int64_t rows, int k, int64_t ldw, bool *handled) {
    *handled = false;
    all_locals_ = false;
    if (staged_k_ < 1 || k != staged_k_ || !weights_match_staged(w, rows, k, ldw)) return BBFMM_OK;
    CHK(complete_all(k));
    const int rc = for_parts([&](int g) -> int {
        FmmTree &t = *parts_[static_cast<size_t>(g)].t;
        const ResidentWeights resident(t.group_weights_resident_);
        return t.set_local_coefficients(w, rows, k, ldw);
    });
    if (rc != BBFMM_OK) return rc;
    all_locals_ = true;
    *handled = true;
    return BBFMM_OK;
}

int DeviceGroup::prepare_primary(bool same_weights) {
    FmmTree &P = *parts_[0].t;
    last_path_ = 0;
    if (!primary_complete_ && dev_staged_k_ > 0) { // (after set_weights: the staged weights; after matvec_device: its device copy)
        P.bind_device();
        const int rc = P.complete_upward_from_staged(dev_staged_k_);
        if (rc != BBFMM_OK) return part_fail(parts_[0], rc);
        P.pin_w_k_ = staged_k_; // the pinned buffer holds exactly the weights the sorted copy was gathered from (0: it does not)
        primary_complete_ = true;
        pending_k_ = 0;
    }
    if (!same_weights) { // the primary is about to stage others (the reference's mixture: old multipoles, new near field)
        staged_k_ = dev_staged_k_ = 0;
        all_complete_ = all_locals_ = false;
    }
    return BBFMM_OK;
}

void DeviceGroup::set_profiling(bool on) {
    for (Part &p : parts_) p.t->set_profiling(on);
}
int DeviceGroup::part_phase_ms(int g, double *ms_out, int64_t *count_out) {
    if (g < 0 || g >= n_parts() || !ms_out) return BBFMM_BAD_ARGUMENT;
    Part &p = parts_[static_cast<size_t>(g)];
    p.t->bind_device();
    std::memcpy(ms_out, p.t->phase_ms(), sizeof(double) * BBFMM_N_PHASES);
    if (count_out) std::memcpy(count_out, p.t->phase_count(), sizeof(int64_t) * BBFMM_N_PHASES);
    parts_[0].t->bind_device();
    return BBFMM_OK;
}
void DeviceGroup::reset_phase_ms() {
    for (Part &p : parts_) {
        p.t->bind_device();
        p.t->reset_phase_ms();
    }
    parts_[0].t->bind_device();
}

} // namespace bbfmm

// Morton (Z-order) keys for the linear BBFMM tree.
//
// Integer layout follows ferreus_bbfmm/src/morton.rs and morton_constants.rs of the
// reference exactly (bit-exact keys are part of the parity contract):
//   key = (interleave(x, y, z) << 15) | level,  x in bit 0, 16 bits per axis,
//   level <= 16 (morton_constants.rs:12-18).
// The reference interleaves with byte lookup tables; the tables are plain bit
// spreads (verified entry by entry), so the same keys are produced with
// arithmetic here.
#pragma once
#include <cmath>
#include <cstdint>

namespace bbfmm {

constexpr uint64_t kMaximumLevel = 16;      // morton_constants.rs:12
constexpr uint64_t kLevelDisplacement = 15; // morton_constants.rs:15
constexpr uint64_t kLevelMask = 0x7FFF;     // morton_constants.rs:18

// morton.rs:29-32
inline double get_side_length(double radius, uint64_t level) {
    return 2.0 * radius / static_cast<double>(uint64_t(1) << level);
}

// Rust `f64 as u64` (morton.rs:46): saturating, NaN -> 0.
inline uint64_t f64_to_u64_saturating(double v) {
    if (!(v > 0.0)) return 0; // negatives, -0, NaN
    if (v >= 18446744073709551616.0) return ~uint64_t(0);
    return static_cast<uint64_t>(v);
}

// morton.rs:35-51 (one axis)
inline uint64_t point_to_anchor_axis(double x, double displacement, double side_length) {
    return f64_to_u64_saturating(std::floor((x - displacement) / side_length));
}

// Spread the low 16 bits of v so that bit i lands on bit i*d.
inline uint64_t spread_bits(uint64_t v, int d) {
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

// Inverse of spread_bits restricted to the bits the reference's decode loops read:
// 21 bits per axis in 3-D (7 loops x 3 bits), 28 in 2-D (7 x 4), 16 in 1-D
// (morton.rs:127-167).
inline uint64_t compact_bits(uint64_t k, int d, int axis) {
    if (d == 1) return k & 0xFFFF;
    uint64_t v = 0;
    const int nbits = (d == 3) ? 21 : 28;
    for (int i = 0; i < nbits; ++i) v |= ((k >> (d * i + axis)) & 1ull) << i;
    return v;
}

// morton.rs:58-119
inline uint64_t encode_morton_point(const uint64_t *anchor, uint64_t level, int d) {
    uint64_t code = 0;
    for (int a = 0; a < d; ++a) code |= spread_bits(anchor[a], d) << a;
    return (code << kLevelDisplacement) | level;
}

inline uint64_t get_level(uint64_t key) { return key & kLevelMask; }

// morton.rs:127-167
inline void decode_key(uint64_t key, int d, uint64_t *anchor, uint64_t *level) {
    *level = key & kLevelMask;
    const uint64_t k = key >> kLevelDisplacement;
    for (int a = 0; a < d; ++a) anchor[a] = compact_bits(k, d, a);
}

// morton.rs:170-190; returns false at level 0.
inline bool get_parent(uint64_t key, int d, uint64_t *parent) {
    const uint64_t level = key & kLevelMask;
    if (level == 0) return false;
    *parent = (((key >> kLevelDisplacement) >> d) << kLevelDisplacement) | (level - 1);
    return true;
}

// morton.rs:266-297: child `suffix` of key.
inline uint64_t get_child(uint64_t key, int d, uint64_t suffix) {
    const uint64_t level = key & kLevelMask;
    return ((((key >> kLevelDisplacement) << d) | suffix) << kLevelDisplacement) | (level + 1);
}

// morton.rs:300-305
inline int get_child_index(uint64_t key, int d) {
    return static_cast<int>((key >> kLevelDisplacement) & ((uint64_t(1) << d) - 1));
}

// morton.rs:328-346
inline void get_center_length(uint64_t key, const double *tree_center, double tree_radius, int d,
                              double *center, double *length) {
    uint64_t anchor[3], level;
    decode_key(key, d, anchor, &level);
    const double side = get_side_length(tree_radius, level);
    for (int a = 0; a < d; ++a)
        center[a] = (static_cast<double>(anchor[a]) + 0.5) * side + (tree_center[a] - tree_radius);
    *length = side;
}

// morton.rs:308-325 on precomputed centres/lengths.
inline bool are_adjacent_cl(const double *ca, double la, const double *cb, double lb, int d) {
    const double tolerance = 1e-6;
    const double length = 0.5 * (la + lb);
    for (int a = 0; a < d; ++a)
        if (!(std::fabs(cb[a] - ca[a]) <= tolerance + length)) return false;
    return true;
}

// Direction vectors in the reference's order (morton_constants.rs:32-74).
inline int direction_vectors(int d, const int (**out)[3]) {
    static const int d1[2][3] = {{-1, 0, 0}, {1, 0, 0}};
    static const int d2[8][3] = {{-1, -1, 0}, {-1, 0, 0}, {-1, 1, 0}, {0, -1, 0},
                                 {0, 1, 0},   {1, -1, 0}, {1, 0, 0},  {1, 1, 0}};
    static const int d3[26][3] = {
        {-1, -1, -1}, {-1, -1, 0}, {-1, -1, 1}, {-1, 0, -1}, {-1, 1, -1}, {-1, 0, 0}, {-1, 0, 1},
        {-1, 1, 0},   {-1, 1, 1},  {0, -1, -1}, {1, -1, -1}, {0, -1, 0},  {0, -1, 1}, {1, -1, 0},
        {1, -1, 1},   {0, 0, -1},  {0, 1, -1},  {1, 0, -1},  {1, 1, -1},  {0, 0, 1},  {0, 1, 0},
        {0, 1, 1},    {1, 0, 0},   {1, 0, 1},   {1, 1, 0},   {1, 1, 1}};
    if (d == 1) { *out = d1; return 2; }
    if (d == 2) { *out = d2; return 8; }
    *out = d3;
    return 26;
}

// morton.rs:214-263: same-level neighbours inside the root box; returns count (<= 26).
inline int get_neighbours(uint64_t key, int d, uint64_t *out) {
    uint64_t anchor[3] = {0, 0, 0}, level;
    decode_key(key, d, anchor, &level);
    const int64_t max_num_boxes = int64_t(1) << level;
    const int(*dirs)[3];
    const int nd = direction_vectors(d, &dirs);
    int cnt = 0;
    for (int i = 0; i < nd; ++i) {
        uint64_t na[3];
        bool ok = true;
        for (int a = 0; a < d; ++a) {
            const int64_t v = static_cast<int64_t>(anchor[a]) + dirs[i][a];
            if (v < 0 || v >= max_num_boxes) { ok = false; break; }
            na[a] = static_cast<uint64_t>(v);
        }
        if (ok) out[cnt++] = encode_morton_point(na, level, d);
    }
    return cnt;
}

// morton.rs:349-373; extents = [mins..., maxs...].
inline void calculate_tree_center_and_radius(const double *extents, int d, double *center,
                                             double *radius) {
    const double eps = 1e-3;
    double r = -INFINITY;
    for (int a = 0; a < d; ++a) {
        const double lo = std::floor(extents[a]);
        const double hi = std::ceil(extents[d + a]);
        center[a] = (lo + hi) / 2.0;
        r = std::fmax(r, (hi - lo) / 2.0 + eps);
    }
    *radius = r;
}

} // namespace bbfmm

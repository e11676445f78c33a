// occupancy_restatement.cpp — CPU ORACLE of the bounds filter + occupancy binning.  TEST INFRASTRUCTURE ONLY.
//
// Restates, function by function, what the reference node computes in its own source right before the hot
// path (od.cpp = /root/reference/minibot_cr18/src/obstacle_detection.cpp):
//   get_occupancy_grid_x_y           od.cpp:134-150   (a linear walk per coordinate, float arithmetic)
//   get_occupancy_grid_location      od.cpp:153-157
//   build_initial_occupancy_grid_dataset  od.cpp:175-269   (predicate, counts, row averages, "hole" marks,
//                                                          the cloud handed to downsample_cloud)
// Unlike the PCL path, all of this arithmetic is in the reference's own translation unit; nothing is taken
// from a third-party library.  The reference still holds no tests or recorded outputs for it, so this file
// is pinned by reading the source, by the hand-computable cases of tests/test_oracle_occupancy.py and by an
// independent NumPy formulation there - not by vectors produced by the reference.
//
// Written from the behaviour of the code, not from its text: the walk is kept as a walk (its result is what
// the float comparisons say, including for NaN arguments), the types are the reference's (float limits,
// long long counts, int indices).  Built with -ffp-contract=off: the node ran as plain x86-64 mul-then-add.
//
// Only tests/, __graft_entry__.smoke() and bench.py may load this library.

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace {

// od.cpp:134-150.  `a` walks up from lo in steps of `block` until lo + (k+1)*block >= a;
// `b` walks down from hi until hi - (k+1)*block <= b.  A NaN argument fails the first test: count 0.
void walk_counts(float a, float b, float lo, float hi, float block, int* ka, int* kb) {
    int ca = 0, cb = 0;
    while (lo + static_cast<float>(ca + 1) * block < a) ++ca;
    while (hi - static_cast<float>(cb + 1) * block > b) ++cb;
    *ka = ca; *kb = cb;
}

}  // namespace

extern "C" {

struct pcop_oracle_occ_params {
    float x_min, x_max, y_min, y_max, z_min, z_max;   // pt_lower_lim_* / pt_upper_lim_* (od.cpp:97-102)
    float block_size, dev_percent;                    // od.cpp:87-88
    int32_t grid_width, grid_height;                  // od.cpp:958-959
};

// od.cpp:958-960: grid size from the limits
void pcop_oracle_occ_grid_dims(float x_min, float x_max, float y_min, float y_max, float block, int32_t* width, int32_t* height) {
    *width = static_cast<int32_t>(std::ceil((std::fabs(y_min) + std::fabs(y_max)) / block));
    *height = static_cast<int32_t>(std::ceil((std::fabs(x_min) + std::fabs(x_max)) / block));
}

// Grid cell of one point: od.cpp:203-204 calls get_occupancy_grid_location(p.y, p.x, y_min, x_max, block, width):
// the column walks up from y_min along the point's y, the row walks down from x_max along the point's x.
int32_t pcop_oracle_occ_cell(float px, float py, const pcop_oracle_occ_params* p) {
    int col, row;
    walk_counts(py, px, p->y_min, p->x_max, p->block_size, &col, &row);
    return row * p->grid_width + col;
}

// build_initial_occupancy_grid_dataset (od.cpp:175-269).  xyz: n points, 3 floats each (x,y,z).
// out_xyz (3 floats per kept point, input order), counts[w*h], row_avg[h], grid[w*h] (100 = "hole", else 0);
// *index_errors = kept points whose cell index is >= w*h (logged, not counted, still kept: od.cpp:205-213).
int pcop_oracle_occupancy(const float* xyz, size_t n, const pcop_oracle_occ_params* p, float* out_xyz, size_t* n_out,
                          long long* counts, long long* row_avg, signed char* grid, int32_t* index_errors) {
    const int w = p->grid_width, h = p->grid_height, size = w * h;
    if (w <= 0 || h <= 0) return -1;
    for (int i = 0; i < size; ++i) counts[i] = 0;
    for (int i = 0; i < h; ++i) row_avg[i] = 0;
    size_t kept = 0;
    int32_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        // only x is tested for NaN; a NaN y or z fails every comparison below and therefore passes (od.cpp:196-199)
        if (std::isnan(x) || x < p->x_min || x > p->x_max || z < p->z_min || z > p->z_max || y < p->y_min || y > p->y_max)
            continue;
        const int32_t index = pcop_oracle_occ_cell(x, y, p);
        if (index >= size) ++bad;
        else ++counts[index];
        out_xyz[3 * kept] = x; out_xyz[3 * kept + 1] = y; out_xyz[3 * kept + 2] = z;
        ++kept;
    }
    for (int r = 0; r < h; ++r) {   // od.cpp:226-234: integer mean of a row
        long long s = 0;
        for (int c = 0; c < w; ++c) s += counts[r * w + c];
        row_avg[r] = s / w;
    }
    const float keep = 1 - p->dev_percent;   // int 1 - float, as written at od.cpp:258
    for (int i = 0; i < size; ++i) {         // od.cpp:241-266: long long < long long * float, i.e. a float comparison
        const long long avg = row_avg[i / w];
        grid[i] = (static_cast<float>(counts[i]) < static_cast<float>(avg) * keep) ? 100 : 0;
    }
    *n_out = kept;
    *index_errors = bad;
    return 0;
}

}  // extern "C"

// sor_restatement.cpp — CPU ORACLE of statistical outlier removal.  TEST INFRASTRUCTURE ONLY.
//
// Restates what the reference node's remove_statistical_outliers (od.cpp:316-340, called at od.cpp:751-754 with
// meanK 15 and the stddev multiplier of params.yaml:20-21) computes through
// pcl::StatisticalOutlierRemoval<PointXYZ>::applyFilterIndices (filters/impl/statistical_outlier_removal.hpp):
//   1. for every finite point, the k = mean_k + 1 nearest neighbours (KdTreeFLANN::nearestKSearch, squared
//      float32 distances by flann::L2_Simple, ascending); entry 0 is dropped, the square roots of the other
//      mean_k are added up in double and divided by mean_k; the mean distance is stored as float.
//      A non-finite point gets 0 and does not count as valid.
//   2. sum and sum of squares of ALL stored distances, sequentially in index order, in double (the square is a
//      float product); mean = sum / valid; variance = (sq_sum - sum*sum/valid) / (valid - 1);
//      threshold = mean + stddev_mul * sqrt(variance).
//   3. a point is removed iff its float distance > threshold (so a non-finite point, distance 0, stays).
//
// PARITY UNPINNED, like the rest of the PCL path (pcl_restatement.cpp): PCL/FLANN are not vendored or installed.
// Choices where the upstream text is implementation-dependent:
//   * `sqrt(nn_dists[k])` is an unqualified call on a float inside namespace pcl; with the <cmath> of the GCC 5
//     toolchain the node was built with (ROS Kinetic) only ::sqrt(double) is visible there, so the square root is
//     taken in double.
//   * a cloud with fewer than mean_k + 1 finite points makes the reference read past the search result (the
//     kd-tree clamps k, the loop does not): rejected here (-2).
//   * ties among equal distances may pick different neighbours but contribute equal values: no effect.
// The k-NN here is an exact uniform-grid search written for this file (ring by ring until the k-th distance
// is covered); tests/test_oracle_sor.py checks it against brute force.
//
// Only tests/, __graft_entry__.smoke() and bench.py may load this library.

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace {

struct P { float x, y, z; };
inline bool finite3(const P& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); }
// flann::L2_Simple, float32, plain mul-then-add
inline float l2(const P& a, const P& b) {
    float r = 0.0f;
    float d = a.x - b.x; r += d * d;
    d = a.y - b.y; r += d * d;
    d = a.z - b.z; r += d * d;
    return r;
}

}  // namespace

extern "C" {

// distances[n] (float), *threshold (double), keep[n] (1 = stays), returns 0 or a negative code.
int pcop_oracle_sor(const float* xyz, size_t n, int mean_k, float stddev_mul, float* distances, double* threshold,
                    uint8_t* keep, size_t* n_keep) {
    if (mean_k < 1) return -1;
    const P* pts = reinterpret_cast<const P*>(xyz);
    std::vector<uint32_t> valid;
    valid.reserve(n);
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        if (!finite3(pts[i])) continue;
        const float c[3] = {pts[i].x, pts[i].y, pts[i].z};
        if (valid.empty()) for (int a = 0; a < 3; ++a) mn[a] = mx[a] = c[a];
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], c[a]); mx[a] = std::max(mx[a], c[a]); }
        valid.push_back((uint32_t)i);
    }
    const size_t nv = valid.size();
    const size_t k = (size_t)mean_k + 1;
    if (nv < k) return -2;
    // uniform grid with about two points per cell (bounded cell count)
    double ext[3];
    for (int a = 0; a < 3; ++a) ext[a] = std::max((double)mx[a] - (double)mn[a], 1e-6);
    double edge = std::cbrt(ext[0] * ext[1] * ext[2] * 2.0 / (double)nv);
    const double longest = std::max(ext[0], std::max(ext[1], ext[2]));
    edge = std::max(edge, longest / 1024.0);
    int dim[3];
    for (int a = 0; a < 3; ++a) dim[a] = (int)(ext[a] / edge) + 1;
    auto cell_of = [&](const P& p, int* c) {
        const float v[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; ++a) { c[a] = (int)(((double)v[a] - (double)mn[a]) / edge); c[a] = std::min(std::max(c[a], 0), dim[a] - 1); }
    };
    const size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
    std::vector<uint32_t> start(ncell + 1, 0), order(nv);
    for (size_t j = 0; j < nv; ++j) { int c[3]; cell_of(pts[valid[j]], c); ++start[((size_t)c[2] * dim[1] + c[1]) * dim[0] + c[0] + 1]; }
    for (size_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
    {
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (size_t j = 0; j < nv; ++j) { int c[3]; cell_of(pts[valid[j]], c); order[fill[((size_t)c[2] * dim[1] + c[1]) * dim[0] + c[0]]++] = valid[j]; }
    }
    std::vector<float> best(k);
    for (size_t i = 0; i < n; ++i) distances[i] = 0.0f;
    const int rmax = std::max(dim[0], std::max(dim[1], dim[2]));
    for (size_t j = 0; j < nv; ++j) {
        const uint32_t i = valid[j];
        const P q = pts[i];
        int qc[3];
        cell_of(q, qc);
        size_t have = 0;
        for (int r = 0; r <= rmax; ++r) {
            // the shell of cells at Chebyshev distance r, clipped to the grid: whole rows on its faces, two ends elsewhere
            for (int cz = std::max(qc[2] - r, 0); cz <= std::min(qc[2] + r, dim[2] - 1); ++cz)
            for (int cy = std::max(qc[1] - r, 0); cy <= std::min(qc[1] + r, dim[1] - 1); ++cy) {
                const bool face = std::abs(cz - qc[2]) == r || std::abs(cy - qc[1]) == r;
                const int step = face ? 1 : std::max(2 * r, 1);
                for (int cx = qc[0] - r; cx <= qc[0] + r; cx += step) {
                    if (cx < 0 || cx >= dim[0]) continue;
                    const size_t c = ((size_t)cz * dim[1] + cy) * dim[0] + cx;
                    for (uint32_t t = start[c]; t < start[c + 1]; ++t) {
                        const float d = l2(q, pts[order[t]]);
                        if (have < k) { best[have++] = d; std::push_heap(best.begin(), best.begin() + have); }
                        else if (d < best[0]) { std::pop_heap(best.begin(), best.end()); best[k - 1] = d; std::push_heap(best.begin(), best.end()); }
                    }
                }
            }
            // every point not yet seen lies outside the cube of cells within r of the query's cell, hence more
            // than r * edge away along some axis (the clamping of cell_of only moves boundary points inwards)
            if (have == k) {
                const double reach = (double)r * edge;
                if ((double)best[0] <= reach * reach) break;
            }
        }
        std::vector<float> s(best.begin(), best.begin() + have);
        std::sort(s.begin(), s.end());
        double dist_sum = 0.0;
        for (size_t t = 1; t < k; ++t) dist_sum += std::sqrt((double)s[t]);   // entry 0 is "the query point"
        distances[i] = (float)(dist_sum / mean_k);
    }
    double sum = 0.0, sq_sum = 0.0;
    for (size_t i = 0; i < n; ++i) { sum += distances[i]; sq_sum += distances[i] * distances[i]; }
    const double mean = sum / (double)nv;
    const double variance = (sq_sum - sum * sum / (double)nv) / ((double)nv - 1.0);
    const double stddev = std::sqrt(variance);
    const double thr = mean + (double)stddev_mul * stddev;
    size_t kept = 0;
    for (size_t i = 0; i < n; ++i) { keep[i] = !((double)distances[i] > thr) ? 1 : 0; kept += keep[i]; }
    *threshold = thr;
    *n_keep = kept;
    return 0;
}

}  // extern "C"

// pcl_restatement.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A dependency-free restatement of what the reference node's hot path computes
// through PCL: pcl::VoxelGrid<PointXYZ>::filter (called at od.cpp:282-285),
// pcl::search::KdTree / KdTreeFLANN radius search (built at od.cpp:791-792) and
// pcl::EuclideanClusterExtraction<PointXYZ>::extract (called at od.cpp:446-454),
// where od.cpp = /root/reference/minibot_cr18/src/obstacle_detection.cpp.
//
// PARITY UNPINNED: the arithmetic of this path lives in PCL (+FLANN, +Eigen),
// which the reference links (CMakeLists.txt:13, package.xml:49, no version
// pinned; ROS Kinetic era => PCL 1.7.2 / 1.8.x, FLANN 1.8.4) but does not
// vendor, and which is not installed in the build image.  The reference holds
// no tests, fixtures or golden outputs for it.  So this
// file restates the published algorithm of those libraries (each function
// names the upstream routine it follows) and is anchored only by the
// hand-computable known-answer tests of tests/test_oracle_kat.py and an
// independent SciPy cross-check (tests/test_oracle_crosscheck.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  The product (libpcop.so) never does.
//
// Build: g++ -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  FP
// contraction must stay off: FLANN's L2_Simple and PCL's key arithmetic are
// plain mul-then-add on the x86-64 builds the node ran on.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

constexpr int LAYOUT_AOS16 = 0;
constexpr int LAYOUT_SOA = 1;
constexpr int LAYOUT_AOS12 = 2;

struct P3 {
    float x, y, z;
};

inline P3 load_pt(const float* xyz, int layout, size_t n, size_t i) {
    switch (layout) {
        case LAYOUT_AOS16: return {xyz[4 * i], xyz[4 * i + 1], xyz[4 * i + 2]};
        case LAYOUT_SOA: return {xyz[i], xyz[n + i], xyz[2 * n + i]};
        default: return {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    }
}

inline void store_pt(float* xyz, int layout, size_t stride, size_t i, const P3& p) {
    switch (layout) {
        case LAYOUT_AOS16:
            xyz[4 * i] = p.x; xyz[4 * i + 1] = p.y; xyz[4 * i + 2] = p.z; xyz[4 * i + 3] = 1.0f;
            break;
        case LAYOUT_SOA:
            xyz[i] = p.x; xyz[stride + i] = p.y; xyz[2 * stride + i] = p.z;
            break;
        default:
            xyz[3 * i] = p.x; xyz[3 * i + 1] = p.y; xyz[3 * i + 2] = p.z;
    }
}

inline bool finite3(const P3& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); }

// ---------------------------------------------------------------------------
// VoxelGrid grid set-up: pcl/filters/impl/voxel_grid.hpp, VoxelGrid<PointT>::
// applyFilter, from "getMinMax3D" to "divb_mul_ = ..." (SURVEY.md §8a-1 steps 1-4).
// ---------------------------------------------------------------------------
struct VoxelSetup {
    float inv;            // inverse_leaf_size_ = 1.0f / leaf (float32)
    float min_p[3], max_p[3];
    int32_t min_b[3], max_b[3], div_b[3];
    uint32_t mul[3];      // divb_mul_ = (1, div_b.x, div_b.x*div_b.y), int32 arithmetic (wraps)
    bool passthrough;     // "Leaf size is too small for the input dataset" rule
    size_t n_valid;
};

// The reference's static_cast<int64_t>(float) is undefined for |v| >= 2^63 or
// NaN.  We define those cases as "overflow" (the rule's intent); DESIGN.md
// records this as the one place the restatement picks a meaning for UB.
inline bool extent_cells(float ext_times_inv, int64_t* out) {
    if (!(ext_times_inv < 2147483648.0f)) return false;  // also catches NaN / +inf
    if (ext_times_inv < 0.0f) return false;
    *out = static_cast<int64_t>(ext_times_inv) + 1;
    return true;
}

VoxelSetup voxel_setup(const float* xyz, int layout, size_t n, float leaf, bool is_dense) {
    VoxelSetup s{};
    s.inv = 1.0f / leaf;  // Eigen::Array4f::Ones() / leaf_size_.array()
    const float fmax = std::numeric_limits<float>::max();
    for (int a = 0; a < 3; ++a) { s.min_p[a] = fmax; s.max_p[a] = -fmax; }
    s.n_valid = 0;
    // pcl::getMinMax3D(cloud, indices, min, max): non-finite points are skipped
    // only when the cloud is not dense.
    for (size_t i = 0; i < n; ++i) {
        P3 p = load_pt(xyz, layout, n, i);
        if (!is_dense && !finite3(p)) continue;
        ++s.n_valid;
        const float v[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; ++a) {
            if (v[a] < s.min_p[a]) s.min_p[a] = v[a];
            if (v[a] > s.max_p[a]) s.max_p[a] = v[a];
        }
    }
    s.passthrough = false;
    if (s.n_valid == 0) {  // empty cloud: nothing to bin, output empty
        for (int a = 0; a < 3; ++a) { s.min_b[a] = s.max_b[a] = 0; s.div_b[a] = 1; }
        s.mul[0] = s.mul[1] = s.mul[2] = 1;
        return s;
    }
    // dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inverse_leaf_size_[0]) + 1; ...
    int64_t d[3];
    bool ok = true;
    for (int a = 0; a < 3; ++a) ok = extent_cells((s.max_p[a] - s.min_p[a]) * s.inv, &d[a]) && ok;
    if (ok) {
        // (dx*dy*dz) > INT32_MAX, evaluated without overflowing int64 (each d < 2^31+1)
        const int64_t lim = std::numeric_limits<int32_t>::max();
        int64_t dxy = d[0] * d[1];
        if (dxy > lim || dxy * d[2] > lim) ok = false;
    }
    if (!ok) { s.passthrough = true; return s; }
    for (int a = 0; a < 3; ++a) {
        s.min_b[a] = static_cast<int32_t>(std::floor(s.min_p[a] * s.inv));
        s.max_b[a] = static_cast<int32_t>(std::floor(s.max_p[a] * s.inv));
        s.div_b[a] = s.max_b[a] - s.min_b[a] + 1;
    }
    s.mul[0] = 1u;
    s.mul[1] = static_cast<uint32_t>(s.div_b[0]);
    s.mul[2] = static_cast<uint32_t>(s.div_b[0]) * static_cast<uint32_t>(s.div_b[1]);
    return s;
}

// ijk = (int)(floor(p * inv) - (float)min_b); idx = ijk0*mul0 + ijk1*mul1 + ijk2*mul2
// (voxel_grid.hpp "First pass").  int32 arithmetic that
// may wrap is done in uint32, which is what the two's-complement build does.
inline uint32_t voxel_key(const VoxelSetup& s, const P3& p) {
    const int32_t i0 = static_cast<int32_t>(std::floor(p.x * s.inv) - static_cast<float>(s.min_b[0]));
    const int32_t i1 = static_cast<int32_t>(std::floor(p.y * s.inv) - static_cast<float>(s.min_b[1]));
    const int32_t i2 = static_cast<int32_t>(std::floor(p.z * s.inv) - static_cast<float>(s.min_b[2]));
    return static_cast<uint32_t>(i0) * s.mul[0] + static_cast<uint32_t>(i1) * s.mul[1] +
           static_cast<uint32_t>(i2) * s.mul[2];
}

struct KeyIdx {
    uint32_t key;
    uint32_t idx;
};

// ---------------------------------------------------------------------------
// Exact fixed-radius search structure standing in for
// pcl::KdTreeFLANN -> flann::KDTreeSingleIndex (leaf_max_size 15, reorder=true,
// eps 0 => exact).  The tree shape does not influence results: the neighbour
// set is exactly { j : L2_Simple(q, p_j) < r2 }.
// ---------------------------------------------------------------------------
struct KdNode {
    int32_t left, right;      // children (internal) or [begin,end) range in `order` (leaf)
    float lo[3], hi[3];       // bounding box of the node's points
    bool leaf;
};

struct KdTree {
    std::vector<P3> pts;          // reordered copy (FLANN reorder=true)
    std::vector<int32_t> order;   // order[k] = original index of pts[k]
    std::vector<KdNode> nodes;
    static constexpr int kLeafMax = 15;   // KDTreeSingleIndexParams(15)

    int32_t build_rec(int32_t b, int32_t e) {
        KdNode nd{};
        for (int a = 0; a < 3; ++a) { nd.lo[a] = std::numeric_limits<float>::max(); nd.hi[a] = -nd.lo[a]; }
        for (int32_t k = b; k < e; ++k) {
            const float v[3] = {pts[k].x, pts[k].y, pts[k].z};
            for (int a = 0; a < 3; ++a) { nd.lo[a] = std::min(nd.lo[a], v[a]); nd.hi[a] = std::max(nd.hi[a], v[a]); }
        }
        const int32_t me = static_cast<int32_t>(nodes.size());
        nodes.push_back(nd);
        if (e - b <= kLeafMax) {
            nodes[me].leaf = true; nodes[me].left = b; nodes[me].right = e;
            return me;
        }
        int dim = 0; float span = -1.0f;
        for (int a = 0; a < 3; ++a) { float sp = nd.hi[a] - nd.lo[a]; if (sp > span) { span = sp; dim = a; } }
        const int32_t mid = b + (e - b) / 2;
        // co-sort pts and order on the split dimension
        std::vector<int32_t> perm(e - b);
        for (int32_t k = 0; k < e - b; ++k) perm[k] = b + k;
        auto coord = [&](int32_t k) { return dim == 0 ? pts[k].x : (dim == 1 ? pts[k].y : pts[k].z); };
        std::nth_element(perm.begin(), perm.begin() + (mid - b), perm.end(),
                         [&](int32_t u, int32_t v) { return coord(u) < coord(v); });
        std::vector<P3> tp(e - b); std::vector<int32_t> to(e - b);
        for (int32_t k = 0; k < e - b; ++k) { tp[k] = pts[perm[k]]; to[k] = order[perm[k]]; }
        std::copy(tp.begin(), tp.end(), pts.begin() + b);
        std::copy(to.begin(), to.end(), order.begin() + b);
        nodes[me].leaf = false;
        const int32_t l = build_rec(b, mid);
        const int32_t r = build_rec(mid, e);
        nodes[me].left = l; nodes[me].right = r;
        return me;
    }

    void build(const std::vector<P3>& in) {
        pts = in;
        order.resize(in.size());
        for (size_t i = 0; i < in.size(); ++i) order[i] = static_cast<int32_t>(i);
        nodes.clear();
        nodes.reserve(in.size() / 4 + 16);
        if (!in.empty()) build_rec(0, static_cast<int32_t>(in.size()));
    }
};

// flann::L2_Simple<float>::operator(): result = 0; diff = a - b; result += diff*diff, x then y then z.
inline float l2_simple(const P3& a, const P3& b) {
    float result = 0.0f;
    float diff = a.x - b.x; result += diff * diff;
    diff = a.y - b.y; result += diff * diff;
    diff = a.z - b.z; result += diff * diff;
    return result;
}

struct DistIdx {
    float d;
    int32_t i;
    // flann::DistanceIndex::operator<: by distance, ties by index
    bool operator<(const DistIdx& o) const { return d < o.d || (d == o.d && i < o.i); }
};

// RadiusResultSet: addPoint keeps dist < radius (strict).  Pruning uses a
// box distance in double with slack, so it never drops a true neighbour.
void radius_search(const KdTree& t, const P3& q, float r2, bool sorted, std::vector<DistIdx>& out) {
    out.clear();
    if (t.nodes.empty()) return;
    const double r2d = static_cast<double>(r2) * (1.0 + 1e-5) + 1e-300;
    int32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const KdNode& nd = t.nodes[stack[--sp]];
        double md = 0.0;
        const float qv[3] = {q.x, q.y, q.z};
        for (int a = 0; a < 3; ++a) {
            double dd = 0.0;
            if (qv[a] < nd.lo[a]) dd = static_cast<double>(nd.lo[a]) - qv[a];
            else if (qv[a] > nd.hi[a]) dd = static_cast<double>(qv[a]) - nd.hi[a];
            md += dd * dd;
        }
        if (md > r2d) continue;
        if (nd.leaf) {
            for (int32_t k = nd.left; k < nd.right; ++k) {
                const float d = l2_simple(q, t.pts[k]);
                if (d < r2) out.push_back({d, t.order[k]});
            }
        } else {
            stack[sp++] = nd.left; stack[sp++] = nd.right;
        }
    }
    if (sorted) std::sort(out.begin(), out.end());  // search::KdTree default sorted_results_=true
}

// pcl::comparePointClusters(a, b) = a.indices.size() < b.indices.size()
struct ClusterRef {
    int32_t size;
    int32_t id;   // discovery order
};
inline bool compare_clusters(const ClusterRef& a, const ClusterRef& b) { return a.size < b.size; }

}  // namespace

extern "C" {

// r2 as pcl::KdTreeFLANN::radiusSearch forms it: static_cast<float>(radius*radius)
// with radius the double widened from the float tolerance.
float pcop_oracle_r2(float tol) {
    const double radius = static_cast<double>(tol);
    return static_cast<float>(radius * radius);
}

// Per-point voxel keys and grid set-up, for stage-wise parity tests.
// keys[i] = 0xFFFFFFFF for dropped (non-finite) points.  grid[0..8] =
// min_b xyz, div_b xyz, mul xyz.  Returns 1 if the overflow rule fires, else 0.
int pcop_oracle_voxel_keys(const float* xyz, int layout, size_t n, float leaf, int is_dense,
                           uint32_t* keys, int32_t* grid) {
    VoxelSetup s = voxel_setup(xyz, layout, n, leaf, is_dense != 0);
    if (grid) {
        for (int a = 0; a < 3; ++a) { grid[a] = s.min_b[a]; grid[3 + a] = s.div_b[a]; grid[6 + a] = static_cast<int32_t>(s.mul[a]); }
    }
    if (s.passthrough) return 1;
    for (size_t i = 0; i < n; ++i) {
        P3 p = load_pt(xyz, layout, n, i);
        keys[i] = (!is_dense && !finite3(p)) ? 0xFFFFFFFFu : voxel_key(s, p);
    }
    return 0;
}

// VoxelGrid<PointXYZ>::applyFilter, defaults of od.cpp:282-285 (cubic leaf,
// downsample_all_data_ = true, min_points_per_voxel_ = 0, no filter field).
// Canonical instance of the implementation-defined parts: equal keys keep
// ascending point index (PCL's std::sort is unstable), centroid = sum / float(n)
// (PCL >= 1.8 AccumulatorXYZ).  Output layout = input layout; SOA stride = n.
// variant: the other legal instances of what PCL leaves open, for the test that shows the
// cluster indices do not depend on them:
//   bit 0: centroid = sum * (1.0f / n)  (PCL 1.7 + Eigen 3.2: the Vector4f accumulator's "/= n" multiplies by the reciprocal)
//   bit 1: equal keys keep DESCENDING point index (one of the orders an unstable std::sort may leave)
//   bit 2: equal keys in the order std::sort leaves them when it compares the key alone, as PCL's call does (this image's libstdc++)
int pcop_oracle_voxel_downsample_ex(const float* xyz, int layout, size_t n, float leaf, int is_dense, int variant,
                                    float* out_xyz, size_t* out_m, int* passthrough);

int pcop_oracle_voxel_downsample(const float* xyz, int layout, size_t n, float leaf, int is_dense,
                                 float* out_xyz, size_t* out_m, int* passthrough) {
    return pcop_oracle_voxel_downsample_ex(xyz, layout, n, leaf, is_dense, 0, out_xyz, out_m, passthrough);
}

int pcop_oracle_voxel_downsample_ex(const float* xyz, int layout, size_t n, float leaf, int is_dense, int variant,
                                    float* out_xyz, size_t* out_m, int* passthrough) {
    if (!xyz && n) return -1;
    if (!(leaf > 0.0f) || !std::isfinite(leaf)) return -2;
    VoxelSetup s = voxel_setup(xyz, layout, n, leaf, is_dense != 0);
    *passthrough = s.passthrough ? 1 : 0;
    if (s.passthrough) {  // output = *input_ (every point, unchanged)
        for (size_t i = 0; i < n; ++i) store_pt(out_xyz, layout, n, i, load_pt(xyz, layout, n, i));
        *out_m = n;
        return 0;
    }
    std::vector<KeyIdx> iv;
    iv.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        P3 p = load_pt(xyz, layout, n, i);
        if (!is_dense && !finite3(p)) continue;
        iv.push_back({voxel_key(s, p), static_cast<uint32_t>(i)});
    }
    // "Second pass: sort the index_vector" — by key; ties by point index (canonical).
    const bool descending = (variant & 2) != 0;
    if (variant & 4) {
        // the call as PCL makes it: std::sort with cloud_point_index_idx::operator< - the key ALONE - so equal keys end up in whatever
        // order this libstdc++'s introsort leaves them (the order a PCL built with the same library sums a voxel's points in)
        std::sort(iv.begin(), iv.end(), [](const KeyIdx& a, const KeyIdx& b) { return a.key < b.key; });
    } else
    std::sort(iv.begin(), iv.end(), [descending](const KeyIdx& a, const KeyIdx& b) {
        return a.key < b.key || (a.key == b.key && (descending ? a.idx > b.idx : a.idx < b.idx));
    });
    // "Third/Fourth pass": one output point per run of equal keys, in key order.
    size_t m = 0, i = 0;
    while (i < iv.size()) {
        size_t j = i;
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;   // Eigen::Vector3f::Zero()
        while (j < iv.size() && iv[j].key == iv[i].key) {
            P3 p = load_pt(xyz, layout, n, iv[j].idx);
            sx += p.x; sy += p.y; sz += p.z;      // xyz += t.getVector3fMap()
            ++j;
        }
        const float cnt = static_cast<float>(j - i);
        if (variant & 1) { const float inv = 1.0f / cnt; store_pt(out_xyz, layout, n, m, P3{sx * inv, sy * inv, sz * inv}); }
        else store_pt(out_xyz, layout, n, m, P3{sx / cnt, sy / cnt, sz / cnt});
        ++m;
        i = j;
    }
    *out_m = m;
    return 0;
}

// pcl::extractEuclideanClusters (segmentation/impl/extract_clusters.hpp) as
// EuclideanClusterExtraction::extract drives it, followed by extract()'s
// std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters).
// use_tree = 1: kd-tree radius search (the CPU-baseline configuration, tree
// built twice like od.cpp:791-792 + extract()); 0: brute force O(m^2).
int pcop_oracle_euclidean_clusters_ex(const float* xyz, int layout, size_t m, float tol,
                                      int32_t min_cluster, int32_t max_cluster, int use_tree,
                                      int32_t* offsets, int32_t* indices, int32_t* k) {
    if (!xyz && m) return -1;
    *k = 0;
    offsets[0] = 0;
    if (m == 0) return 0;                       // clusters.clear()
    std::vector<P3> pts(m);
    for (size_t i = 0; i < m; ++i) {
        pts[i] = load_pt(xyz, layout, m, i);
        if (!finite3(pts[i])) return -6;        // radiusSearch asserts on invalid points
    }
    const float r2 = pcop_oracle_r2(tol);
    KdTree tree;
    if (use_tree) {
        tree.build(pts);                        // od.cpp:791-792
        tree.build(pts);                        // tree_->setInputCloud(input_, indices_) in extract()
    }
    std::vector<bool> processed(m, false);
    std::vector<DistIdx> nn;
    std::vector<std::vector<int32_t>> clusters;
    std::vector<int32_t> seed_queue;
    for (size_t i = 0; i < m; ++i) {
        if (processed[i]) continue;
        seed_queue.clear();
        size_t sq_idx = 0;
        seed_queue.push_back(static_cast<int32_t>(i));
        processed[i] = true;
        while (sq_idx < seed_queue.size()) {
            const P3& q = pts[seed_queue[sq_idx]];
            if (use_tree) {
                radius_search(tree, q, r2, true, nn);
            } else {
                nn.clear();
                for (size_t j = 0; j < m; ++j) {
                    const float d = l2_simple(q, pts[j]);
                    if (d < r2) nn.push_back({d, static_cast<int32_t>(j)});
                }
                std::sort(nn.begin(), nn.end());
            }
            if (nn.empty()) { ++sq_idx; continue; }          // if (!ret) { sq_idx++; continue; }
            for (size_t j = 1; j < nn.size(); ++j) {          // nn_start_idx = 1 (sorted results)
                if (processed[nn[j].i]) continue;
                seed_queue.push_back(nn[j].i);
                processed[nn[j].i] = true;
            }
            ++sq_idx;
        }
        // extractEuclideanClusters takes the limits as `unsigned int` (a negative int wraps)
        if (seed_queue.size() >= static_cast<uint32_t>(min_cluster) &&
            seed_queue.size() <= static_cast<uint32_t>(max_cluster)) {
            std::vector<int32_t> r(seed_queue);
            std::sort(r.begin(), r.end());
            r.erase(std::unique(r.begin(), r.end()), r.end());
            clusters.push_back(std::move(r));
        }
    }
    // Largest first; the permutation among equal sizes is libstdc++'s std::sort on
    // the reverse range, a function of the size sequence only.
    std::vector<ClusterRef> refs(clusters.size());
    for (size_t c = 0; c < clusters.size(); ++c) refs[c] = {static_cast<int32_t>(clusters[c].size()), static_cast<int32_t>(c)};
    std::sort(refs.rbegin(), refs.rend(), compare_clusters);
    int32_t pos = 0;
    for (size_t c = 0; c < refs.size(); ++c) {
        const std::vector<int32_t>& r = clusters[refs[c].id];
        std::copy(r.begin(), r.end(), indices + pos);
        pos += static_cast<int32_t>(r.size());
        offsets[c + 1] = pos;
    }
    *k = static_cast<int32_t>(refs.size());
    return 0;
}

int pcop_oracle_euclidean_clusters(const float* xyz, int layout, size_t m, float tol,
                                   int32_t min_cluster, int32_t max_cluster,
                                   int32_t* offsets, int32_t* indices, int32_t* k) {
    return pcop_oracle_euclidean_clusters_ex(xyz, layout, m, tol, min_cluster, max_cluster, 1, offsets, indices, k);
}

// The permutation extract()'s final std::sort applies to a discovery-ordered
// size sequence; exposed so tests can pin KAT 7 directly.
void pcop_oracle_cluster_order(const int32_t* sizes, int32_t k, int32_t* order) {
    std::vector<ClusterRef> refs(k);
    for (int32_t c = 0; c < k; ++c) refs[c] = {sizes[c], c};
    std::sort(refs.rbegin(), refs.rend(), compare_clusters);
    for (int32_t c = 0; c < k; ++c) order[c] = refs[c].id;
}

// PointWithRad per cluster (msg/PointWithRad.msg).  The reference never defines
// the producer (create_cluster_cloud is only a commented call, od.cpp:813), so
// this is the build's definition: centre = mean of the members in
// double, rounded to float; r = max over members of calculate_distance()
// (od.cpp:457-464: float diffs, pow(.,2) in double cast to float, float sum,
// sqrt in double cast to float).
int pcop_oracle_cluster_summaries(const float* xyz, int layout, size_t m, const int32_t* offsets,
                                  const int32_t* indices, int32_t k, float* xyzr) {
    for (int32_t c = 0; c < k; ++c) {
        double sx = 0, sy = 0, sz = 0;
        const int32_t b = offsets[c], e = offsets[c + 1];
        for (int32_t j = b; j < e; ++j) { P3 p = load_pt(xyz, layout, m, indices[j]); sx += p.x; sy += p.y; sz += p.z; }
        const double cnt = static_cast<double>(e - b);
        const float cx = static_cast<float>(sx / cnt), cy = static_cast<float>(sy / cnt), cz = static_cast<float>(sz / cnt);
        float r = 0.0f;
        for (int32_t j = b; j < e; ++j) {
            P3 p = load_pt(xyz, layout, m, indices[j]);
            const float xd = static_cast<float>(std::pow(static_cast<double>(p.x - cx), 2));
            const float yd = static_cast<float>(std::pow(static_cast<double>(p.y - cy), 2));
            const float zd = static_cast<float>(std::pow(static_cast<double>(p.z - cz), 2));
            const float d = static_cast<float>(std::sqrt(static_cast<double>(xd + yd + zd)));
            if (d > r) r = d;
        }
        xyzr[4 * c] = cx; xyzr[4 * c + 1] = cy; xyzr[4 * c + 2] = cz; xyzr[4 * c + 3] = r;
    }
    return 0;
}

// Margin diagnostic: number of unordered pairs whose d2 lies
// within `ulps` units-in-the-last-place of r2.  Brute force; small m only.
int64_t pcop_oracle_knife_edge_pairs(const float* xyz, int layout, size_t m, float tol, int ulps) {
    const float r2 = pcop_oracle_r2(tol);
    int32_t r2bits; std::memcpy(&r2bits, &r2, 4);
    int64_t cnt = 0;
    for (size_t i = 0; i < m; ++i) {
        P3 a = load_pt(xyz, layout, m, i);
        for (size_t j = i + 1; j < m; ++j) {
            const float d = l2_simple(a, load_pt(xyz, layout, m, j));
            int32_t db; std::memcpy(&db, &d, 4);
            if (std::llabs(static_cast<long long>(db) - r2bits) <= ulps) ++cnt;
        }
    }
    return cnt;
}

// The margin diagnostic for clouds of any size (a uniform grid of cell edge tol finds the candidate pairs):
//   out[0] = unordered pairs with d2 within `ulps` units in the last place of r2 (on either side),
//   out[1] = how many of them decide the components: a pair just outside the radius whose points lie in
//            different components, plus the components that would fall apart if every pair just inside
//            the radius were dropped.  0 means: no last-bit change of one squared distance moves a cluster border.
int pcop_oracle_knife_edge_report(const float* xyz, int layout, size_t m, float tol, int ulps, int64_t* out) {
    out[0] = out[1] = 0;
    if (m == 0) return 0;
    const float r2 = pcop_oracle_r2(tol);
    if (!(r2 > 0.0f)) return 0;
    int32_t r2bits; std::memcpy(&r2bits, &r2, 4);
    const double inv = 1.0 / std::fabs(static_cast<double>(tol));
    std::vector<P3> pts(m);
    std::vector<int64_t> cx(m), cy(m), cz(m);
    for (size_t i = 0; i < m; ++i) {
        pts[i] = load_pt(xyz, layout, m, i);
        if (!finite3(pts[i])) return -3;
        cx[i] = static_cast<int64_t>(std::floor(pts[i].x * inv)); cy[i] = static_cast<int64_t>(std::floor(pts[i].y * inv));
        cz[i] = static_cast<int64_t>(std::floor(pts[i].z * inv));
    }
    std::vector<uint32_t> order(m);
    for (size_t i = 0; i < m; ++i) order[i] = static_cast<uint32_t>(i);
    auto less = [&](uint32_t a, uint32_t b) { return cz[a] != cz[b] ? cz[a] < cz[b] : cy[a] != cy[b] ? cy[a] < cy[b] : cx[a] < cx[b]; };
    std::sort(order.begin(), order.end(), less);
    struct Cell { int64_t x, y, z; size_t b, e; };
    std::vector<Cell> cells;
    for (size_t i = 0; i < m;) {
        size_t j = i;
        while (j < m && cx[order[j]] == cx[order[i]] && cy[order[j]] == cy[order[i]] && cz[order[j]] == cz[order[i]]) ++j;
        cells.push_back({cx[order[i]], cy[order[i]], cz[order[i]], i, j});
        i = j;
    }
    auto find_cell = [&](int64_t x, int64_t y, int64_t z) -> const Cell* {
        size_t lo = 0, hi = cells.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const Cell& c = cells[mid];
            const bool lt = c.z != z ? c.z < z : c.y != y ? c.y < y : c.x < x;
            if (lt) lo = mid + 1; else hi = mid;
        }
        return (lo < cells.size() && cells[lo].x == x && cells[lo].y == y && cells[lo].z == z) ? &cells[lo] : nullptr;
    };
    std::vector<uint32_t> ua(m), ub(m);   // components with every pair inside the radius / without the knife-edge ones
    for (size_t i = 0; i < m; ++i) ua[i] = ub[i] = static_cast<uint32_t>(i);
    auto find = [](std::vector<uint32_t>& u, uint32_t v) { while (u[v] != v) { u[v] = u[u[v]]; v = u[v]; } return v; };
    auto unite = [&](std::vector<uint32_t>& u, uint32_t a, uint32_t b) { a = find(u, a); b = find(u, b); if (a != b) u[std::max(a, b)] = std::min(a, b); };
    std::vector<std::pair<uint32_t, uint32_t>> near_out;
    for (const Cell& c : cells) {
        for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
            const Cell* o = find_cell(c.x + dx, c.y + dy, c.z + dz);
            if (!o) continue;
            for (size_t a = c.b; a < c.e; ++a) for (size_t b = o->b; b < o->e; ++b) {
                const uint32_t i = order[a], j = order[b];
                if (i >= j) continue;
                const float d = l2_simple(pts[i], pts[j]);
                int32_t db; std::memcpy(&db, &d, 4);
                const bool knife = std::llabs(static_cast<long long>(db) - r2bits) <= ulps;
                if (knife) ++out[0];
                if (d < r2) { unite(ua, i, j); if (!knife) unite(ub, i, j); }
                else if (knife) near_out.push_back({i, j});
            }
        }
    }
    size_t ca = 0, cb = 0;
    for (size_t i = 0; i < m; ++i) { if (find(ua, static_cast<uint32_t>(i)) == i) ++ca; if (find(ub, static_cast<uint32_t>(i)) == i) ++cb; }
    out[1] = static_cast<int64_t>(cb - ca);
    for (const auto& pr : near_out) if (find(ua, pr.first) != find(ua, pr.second)) ++out[1];
    return 0;
}

// Whole chain of SURVEY.md §3.3 on one cloud: voxel grid, then clustering of
// the downsampled cloud.  voxel_xyz: AOS12, capacity n points.
int pcop_oracle_process_xyz(const float* xyz, int layout, size_t n, int is_dense, float leaf, float tol,
                            int32_t min_cluster, int32_t max_cluster, float* voxel_xyz, size_t* out_m,
                            int* passthrough, int32_t* offsets, int32_t* indices, int32_t* k) {
    // run the voxel grid in the input layout, then repack to AOS12
    std::vector<float> tmp(n * 4 + 4);
    int rc = pcop_oracle_voxel_downsample(xyz, layout, n, leaf, is_dense, tmp.data(), out_m, passthrough);
    if (rc) return rc;
    for (size_t i = 0; i < *out_m; ++i) {
        P3 p = load_pt(tmp.data(), layout, n, i);
        voxel_xyz[3 * i] = p.x; voxel_xyz[3 * i + 1] = p.y; voxel_xyz[3 * i + 2] = p.z;
    }
    return pcop_oracle_euclidean_clusters(voxel_xyz, LAYOUT_AOS12, *out_m, tol, min_cluster, max_cluster,
                                          offsets, indices, k);
}

}  // extern "C"

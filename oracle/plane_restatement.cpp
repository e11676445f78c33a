// plane_restatement.cpp — CPU ORACLE of the RANSAC plane removal.  TEST INFRASTRUCTURE ONLY.
//
// Restates what the reference node's segment_plane_and_extract_indices (od.cpp:342-428, called at od.cpp:778-781 with
// axis (0,0,1), plane_segment_dist_thres and plane_segment_angle of params.yaml) computes through PCL:
//   pcl::SACSegmentation<PointXYZ> (segmentation/impl/sac_segmentation.hpp) with setOptimizeCoefficients(true),
//   SACMODEL_PERPENDICULAR_PLANE, SAC_RANSAC, setAxis, setEpsAngle, setDistanceThreshold, its defaults
//   max_iterations_ = 50, probability_ = 0.99, random_ = false, and pcl::ExtractIndices, inside the loop
//   "while more than 30 % of the cloud is left: segment, stop if no inliers, keep the non-inliers".
//
// One call of seg.segment() [SACSegmentation::segment]:
//   * a fresh SampleConsensusModelPerpendicularPlane (random_ = false => boost::mt19937 seeded with 12345u;
//     rnd() = boost::variate_generator<mt19937&, uniform_int<>(0, INT_MAX)> = the generator's 32-bit word / 2;
//     shuffled_indices_ = 0..n-1) and a fresh RandomSampleConsensus (threshold, probability 0.99, max_iterations 50);
//   * RandomSampleConsensus::computeModel [sample_consensus/impl/ransac.hpp]: while (iterations_ < k && skipped < 10*max_it):
//     getSamples (drawIndexSample: for i in 0..2 swap(shuffled[i], shuffled[i + rnd() % (n - i)]); retried up to 1000
//     times until isSampleGood); computeModelCoefficients (false => ++skipped, continue); countWithinDistance; a
//     strictly better count becomes the best model and k = log(1 - 0.99) / log(1 - w^3) with w = count / n clamped by
//     DBL_EPSILON; ++iterations_; break once iterations_ > max_iterations_;  then selectWithinDistance of the best model;
//   * optimizeModelCoefficients [sac_model_plane.hpp]: skipped unless more than 3 inliers; computeMeanAndCovarianceMatrix
//     (nine sequential float sums in inlier order, divided by n; covariance = E[ab] - E[a]E[b]), pcl::eigen33 (smallest
//     eigenvalue by the closed form of common/impl/eigen.hpp computeRoots, eigenvector = largest of the three row cross
//     products of (A - lambda I), normalised), d = -n . centroid; an invalid result keeps the RANSAC coefficients;
//   * "Refine inliers": selectWithinDistance again with the refined coefficients.
// The plane predicate (SampleConsensusModelPlane::countWithinDistance / selectWithinDistance):
//   fabs(a*x + b*y + c*z + d) < threshold, the sum in float32, the comparison against the double threshold.
// SampleConsensusModelPerpendicularPlane::isModelValid: with eps_angle > 0 the angle between the normal and the axis
// (getAngle3D, double acos of the clamped normalised dot), folded to [0, pi/2], must not exceed eps_angle.  The node
// passes plane_segment_angle = 20 as is (od.cpp:111, 371: an int in what PCL reads as radians), so every plane passes.
//
// PARITY UNPINNED, like the rest of the PCL path: PCL / Eigen / Boost are neither vendored nor installed.  Choices
// where the upstream result depends on the build (all flagged `open choice` below):
//   * the 4-float dot products are summed left to right, ((a*x + b*y) + c*z) + d (Eigen's SSE reduction pairs them
//     differently; without vectorisation it is this order);
//   * VectorXf::normalize() divides by the norm (Eigen >= 3.3; Eigen 3.2 multiplies by the reciprocal);
//   * std::sort / unstable steps do not occur here; the only randomness is the fixed-seed generator above.
// Only tests/ may load this library.

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct P { float x, y, z; };

// boost::mt19937 (the 32-bit Mersenne twister of Matsumoto and Nishimura, seeded by the usual linear recurrence)
struct Mt19937 {
    uint32_t s[624];
    int idx;
    explicit Mt19937(uint32_t seed) {
        s[0] = seed;
        for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7FFFFFFFu);
                s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
            }
            idx = 0;
        }
        uint32_t y = s[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
        return y;
    }
};

// boost::uniform_int<>(0, INT_MAX) on a 32-bit engine: two engine values per bucket
inline int rnd(Mt19937& g) { return (int)(g.next() >> 1); }

// open choice: left-to-right float sum of the four products
inline float plane_dot(const float c[4], const P& p) { return ((c[0] * p.x + c[1] * p.y) + c[2] * p.z) + c[3]; }

inline bool within(const float c[4], const P& p, double threshold) { return std::fabs((double)plane_dot(c, p)) < threshold; }

// SampleConsensusModelPlane::isSampleGood / the collinearity test of computeModelCoefficients
inline void ratios(const P& p0, const P& p1, const P& p2, float r[3]) {
    r[0] = (p1.x - p0.x) / (p2.x - p0.x); r[1] = (p1.y - p0.y) / (p2.y - p0.y); r[2] = (p1.z - p0.z) / (p2.z - p0.z);
}

bool model_from_sample(const P& p0, const P& p1, const P& p2, float c[4]) {
    float r[3];
    ratios(p0, p1, p2, r);
    if (r[0] == r[1] && r[2] == r[1]) return false;
    const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z, bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
    c[0] = ay * bz - az * by;
    c[1] = az * bx - ax * bz;
    c[2] = ax * by - ay * bx;
    c[3] = 0.0f;
    const float nrm = std::sqrt(((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) + c[3] * c[3]);
    c[0] /= nrm; c[1] /= nrm; c[2] /= nrm; c[3] /= nrm;   // open choice: true division
    c[3] = -1.0f * (((c[0] * p0.x + c[1] * p0.y) + c[2] * p0.z) + c[3] * 1.0f);
    return true;
}

// SampleConsensusModelPerpendicularPlane::isModelValid
bool model_valid(const float c[4], const float axis[3], double eps_angle) {
    if (!(eps_angle > 0.0)) return true;
    const double d = (double)((c[0] * axis[0] + c[1] * axis[1]) + c[2] * axis[2]);
    const double n1 = std::sqrt((double)((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]));
    const double n2 = std::sqrt((double)((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]));
    double rad = d / (n1 * n2);
    if (rad < -1.0) rad = -1.0; else if (rad > 1.0) rad = 1.0;
    double ang = std::fabs(std::acos(rad));
    ang = std::min(ang, M_PI - ang);
    return !(ang > eps_angle);
}

// pcl::computeRoots2 / computeRoots (common/impl/eigen.hpp), float
void roots2(float b, float c, float r[3]) {
    r[0] = 0.0f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    const float sd = std::sqrt(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

void roots3(const float m[3][3], float r[3]) {
    const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] -
                     m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
    const float c2 = m[0][0] + m[1][1] + m[2][2];
    if (std::fabs(c0) < FLT_EPSILON) { roots2(c2, c1, r); return; }
    const float inv3 = 1.0f / 3.0f, sqrt3 = std::sqrt(3.0f);
    const float c2_3 = c2 * inv3;
    float a_3 = (c1 - c2 * c2_3) * inv3;
    if (a_3 > 0.0f) a_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_3 * (2.0f * c2_3 * c2_3 - c1));
    float q = half_b * half_b + a_3 * a_3 * a_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = std::sqrt(-a_3);
    const float theta = std::atan2(std::sqrt(-q), half_b) * inv3;
    const float ct = std::cos(theta), st = std::sin(theta);
    r[0] = c2_3 + 2.0f * rho * ct;
    r[1] = c2_3 - rho * (ct + sqrt3 * st);
    r[2] = c2_3 - rho * (ct - sqrt3 * st);
    if (r[0] >= r[1]) std::swap(r[0], r[1]);
    if (r[1] >= r[2]) { std::swap(r[1], r[2]); if (r[0] >= r[1]) std::swap(r[0], r[1]); }
    if (r[0] <= 0.0f) roots2(c2, c1, r);
}

// SampleConsensusModelPlane::optimizeModelCoefficients on the inliers (in index order)
void optimise(const std::vector<P>& pts, const std::vector<int>& inl, const float c_in[4], const float axis[3], double eps_angle, float c_out[4]) {
    std::memcpy(c_out, c_in, 16);
    if (!model_valid(c_in, axis, eps_angle) || inl.size() <= 3) return;
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : inl) {
        const P& p = pts[(size_t)i];
        acc[0] += p.x * p.x; acc[1] += p.x * p.y; acc[2] += p.x * p.z; acc[3] += p.y * p.y; acc[4] += p.y * p.z; acc[5] += p.z * p.z;
        acc[6] += p.x; acc[7] += p.y; acc[8] += p.z;
    }
    const float cnt = (float)inl.size();
    for (float& a : acc) a /= cnt;
    float m[3][3];
    m[0][0] = acc[0] - acc[6] * acc[6]; m[0][1] = acc[1] - acc[6] * acc[7]; m[0][2] = acc[2] - acc[6] * acc[8];
    m[1][1] = acc[3] - acc[7] * acc[7]; m[1][2] = acc[4] - acc[7] * acc[8]; m[2][2] = acc[5] - acc[8] * acc[8];
    m[1][0] = m[0][1]; m[2][0] = m[0][2]; m[2][1] = m[1][2];
    // pcl::eigen33: smallest eigenvalue and its vector
    float scale = 0.0f;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) scale = std::max(scale, std::fabs(m[a][b]));
    if (scale <= FLT_MIN) scale = 1.0f;
    float s[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) s[a][b] = m[a][b] / scale;
    float ev[3];
    roots3(s, ev);
    for (int a = 0; a < 3; ++a) s[a][a] -= ev[0];
    auto cross = [](const float u[3], const float v[3], float o[3]) { o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0]; };
    float v1[3], v2[3], v3[3];
    cross(s[0], s[1], v1); cross(s[0], s[2], v2); cross(s[1], s[2], v3);
    auto sq = [](const float v[3]) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; };
    const float l1 = sq(v1), l2 = sq(v2), l3 = sq(v3);
    const float* best; float len;
    if (l1 >= l2 && l1 >= l3) { best = v1; len = l1; } else if (l2 >= l1 && l2 >= l3) { best = v2; len = l2; } else { best = v3; len = l3; }
    const float rl = std::sqrt(len);
    float c[4] = {best[0] / rl, best[1] / rl, best[2] / rl, 0.0f};
    c[3] = -1.0f * (((c[0] * acc[6] + c[1] * acc[7]) + c[2] * acc[8]) + 0.0f * 1.0f);   // dot with the centroid (x, y, z, 1), d still 0
    if (model_valid(c, axis, eps_angle)) std::memcpy(c_out, c, 16);
}

// One seg.segment(): false = no model (inliers empty)
bool segment(const std::vector<P>& pts, double threshold, const float axis[3], double eps_angle, int max_iterations, double probability,
             std::vector<int>& inliers, float coeff[4], int* iterations_out) {
    const size_t n = pts.size();
    inliers.clear();
    *iterations_out = 0;
    if (n < 3) return false;
    Mt19937 gen(12345u);
    std::vector<int> shuffled(n);
    for (size_t i = 0; i < n; ++i) shuffled[i] = (int)i;
    int iterations = 0, best_count = -INT_MAX;
    double k = 1.0;
    const double log_probability = std::log(1.0 - probability), one_over = 1.0 / (double)n;
    unsigned skipped = 0;
    const unsigned max_skip = (unsigned)max_iterations * 10u;
    bool have = false;
    float best[4] = {0, 0, 0, 0};
    while (iterations < k && skipped < max_skip) {
        int smp[3];
        bool good = false;
        for (unsigned it = 0; it < 1000 && !good; ++it) {   // getSamples
            for (size_t i = 0; i < 3; ++i) std::swap(shuffled[i], shuffled[i + (size_t)(rnd(gen) % (int)(n - i))]);
            smp[0] = shuffled[0]; smp[1] = shuffled[1]; smp[2] = shuffled[2];
            float r[3];
            ratios(pts[(size_t)smp[0]], pts[(size_t)smp[1]], pts[(size_t)smp[2]], r);
            good = (r[0] != r[1]) || (r[2] != r[1]);
        }
        if (!good) break;   // "No samples could be selected!"
        float c[4];
        if (!model_from_sample(pts[(size_t)smp[0]], pts[(size_t)smp[1]], pts[(size_t)smp[2]], c)) { ++skipped; continue; }
        int cnt = 0;
        if (model_valid(c, axis, eps_angle)) for (const P& p : pts) cnt += within(c, p, threshold) ? 1 : 0;
        if (cnt > best_count) {
            best_count = cnt; have = true; std::memcpy(best, c, 16);
            const double w = (double)cnt * one_over;
            double p_no = 1.0 - std::pow(w, 3.0);
            p_no = std::max(DBL_EPSILON, p_no); p_no = std::min(1.0 - DBL_EPSILON, p_no);
            k = log_probability / std::log(p_no);
        }
        ++iterations;
        if (iterations > max_iterations) break;
    }
    *iterations_out = iterations;
    if (!have) return false;
    std::vector<int> inl;
    if (model_valid(best, axis, eps_angle)) for (size_t i = 0; i < n; ++i) if (within(best, pts[i], threshold)) inl.push_back((int)i);
    float refined[4];
    optimise(pts, inl, best, axis, eps_angle, refined);
    std::memcpy(coeff, refined, 16);
    if (model_valid(refined, axis, eps_angle)) for (size_t i = 0; i < n; ++i) if (within(refined, pts[i], threshold)) inliers.push_back((int)i);
    return true;
}

}  // namespace

extern "C" {

// segment_plane_and_extract_indices (od.cpp:342-428) on an AOS12 cloud.  out_xyz (room for n points) receives what is
// left ("planar_cloud" at the end), last_plane_xyz (may be NULL, room for n) the inliers of the last plane removed
// ("indices_cloud"), coeffs (may be NULL, room for 4 * max_planes floats) the refined coefficients of every plane removed.
// *n_planes = planes removed.  Returns 0, or 1 when a round found no plane ("Couldn't estimate a planar model").
int pcop_oracle_segment_planes(const float* xyz, size_t n, float dist_thresh, const float* axis, double eps_angle,
                               float* out_xyz, size_t* n_out, float* last_plane_xyz, size_t* n_last_plane,
                               float* coeffs, int max_planes, int* n_planes, int* iterations) {
    std::vector<P> cloud(n);
    for (size_t i = 0; i < n; ++i) cloud[i] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const double threshold = (double)dist_thresh;   // setDistanceThreshold(double) from the float parameter
    const int nr_points = (int)n;
    int planes = 0, rc = 0;
    std::vector<P> last;
    while ((double)cloud.size() > 0.3 * (double)nr_points) {
        std::vector<int> inl;
        float c[4];
        int its = 0;
        const bool ok = segment(cloud, threshold, axis, eps_angle, 50, 0.99, inl, c, &its);
        if (iterations && planes < max_planes) iterations[planes] = its;
        if (!ok || inl.empty()) { rc = 1; break; }
        std::vector<P> rest;
        last.clear();
        size_t q = 0;
        for (size_t i = 0; i < cloud.size(); ++i) {
            if (q < inl.size() && (size_t)inl[q] == i) { last.push_back(cloud[i]); ++q; } else rest.push_back(cloud[i]);
        }
        if (coeffs && planes < max_planes) std::memcpy(coeffs + 4 * planes, c, 16);
        cloud.swap(rest);
        ++planes;
    }
    for (size_t i = 0; i < cloud.size(); ++i) { out_xyz[3 * i] = cloud[i].x; out_xyz[3 * i + 1] = cloud[i].y; out_xyz[3 * i + 2] = cloud[i].z; }
    *n_out = cloud.size();
    if (last_plane_xyz) for (size_t i = 0; i < last.size(); ++i) { last_plane_xyz[3 * i] = last[i].x; last_plane_xyz[3 * i + 1] = last[i].y; last_plane_xyz[3 * i + 2] = last[i].z; }
    if (n_last_plane) *n_last_plane = last.size();
    *n_planes = planes;
    return rc;
}

// The generator alone, for the known-answer test: the first `count` values of rnd() after the fixed seed.
void pcop_oracle_plane_rnd(int* out, int count) {
    Mt19937 g(12345u);
    for (int i = 0; i < count; ++i) out[i] = rnd(g);
}

}  // extern "C"

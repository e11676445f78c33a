// pcop_device.hpp — gfx950 device code of the obstacle hot path.
//
// Three groups of kernels, all HBM/LDS index-and-scatter work (no MFMA):
//   1. voxel grid     : min/max -> stable LSD radix partition of {x,y,z,idx} records by
//                       PCL's linear voxel key -> run heads -> in-order centroid sums
//                       (replaces pcl::VoxelGrid::applyFilter called at od.cpp:282-285)
//   2. cell grid      : the same partition by a coarse cell key (cell edge = tol/1.99)
//                       (replaces the FLANN kd-tree built at od.cpp:791-792)
//   3. clusters       : wave-per-cell neighbour test + lock-free union-find, component
//                       sizes / discovery order, stable partition of point indices by
//                       cluster rank (replaces EuclideanClusterExtraction::extract, od.cpp:446-454)
//
// Every count that depends on the data (valid points, voxels M, cells, clusters) lives in
// DevState in device memory; kernels read it there, so a frame is one uninterrupted stream
// of launches with no host round trip.
//
// Built with -ffp-contract=off and correctly rounded f32 divide: the float arithmetic below
// must match PCL/FLANN's x86-64 mul-then-add sequences bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcop {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxBits = 11;                     // widest radix digit
constexpr int kMaxBins = (1 << kMaxBits) + 1;    // + 1 trash bin (non-finite points)
constexpr int kItems = 8;                        // records per thread in the partition kernels
constexpr int kTile = kBlock * kItems;           // records per partition block
constexpr int kMaxPasses = 3;                    // 3 x 11 bits >= 32-bit keys
constexpr int kSegs = 32;                        // row segments of the tile-count table scan
constexpr int kScanItems = 4;                    // elements per thread in flag-scan kernels
constexpr int kScanTile = kBlock * kScanItems;
constexpr int kReduceBlocks = 1024;              // grid of the min/max kernel

enum ErrBits : uint32_t {
    kErrNonFinite = 1u,    // NaN/Inf where the reference is undefined
    kErrReplanVoxel = 2u,  // voxel key needs more radix passes than were launched
    kErrReplanCell = 4u,   // cell key needs more radix passes than were launched
    kErrReplanCsr = 8u,    // cluster-rank key needs more radix passes than were launched
    kErrRange = 16u,       // coarse cell grid >= 2^32 cells
};

struct RadixPlan {
    uint32_t key_bits;
    uint32_t npass;
    uint32_t bits[kMaxPasses];
    uint32_t shift[kMaxPasses];
};

// Device-resident frame state.  Copied to pinned host memory at the end of a frame.
struct DevState {
    uint32_t err;
    uint32_t n_in;          // points handed in
    uint32_t is_dense;
    uint32_t n_valid;       // finite points (== n_in for dense clouds)
    uint32_t passthrough;   // VoxelGrid overflow rule fired
    uint32_t m;             // voxels (downsampled points)
    // voxel grid (pcl::VoxelGrid members)
    float    inv_leaf;
    float    vmin[3], vmax[3];
    int32_t  min_b[3], div_b[3];
    uint32_t mul[3];
    RadixPlan plan_voxel;
    // cluster stage
    uint32_t c_n;           // points entering clustering
    uint32_t c_from_input;  // 1: cluster the input cloud (passthrough / stage API), 0: the centroids
    float    r2;            // (float)((double)tol*(double)tol)
    uint32_t min_cluster, max_cluster;
    double   inv_cell;      // 1.99 / tol
    int64_t  cmin[3];
    uint32_t cdim[3];
    RadixPlan plan_cell;
    uint32_t n_cells;
    uint32_t hash_mask;     // cell hash table has hash_mask+1 slots (power of two >= 2*n_cells)
    uint32_t n_kept;        // clusters with min <= size <= max
    uint32_t n_kept_pts;    // sum of their sizes
    RadixPlan plan_csr;
    uint32_t pad[3];
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ bool finite3(float x, float y, float z) {
    return isfinite(x) && isfinite(y) && isfinite(z);
}
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// inclusive scan of one value per lane across the wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t t = __shfl_up(v, d, kWave);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread across a 256-thread block; *total = block sum.
// s_w: kWavesPerBlock + 1 words of LDS.  Contains two barriers.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const uint32_t incl = wave_incl_scan(v);
    const uint32_t w = threadIdx.x / kWave;
    __syncthreads();  // s_w may still be read from a previous call
    if (lane_id() == kWave - 1) s_w[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) {
        const uint32_t t = s_w[i];
        if ((uint32_t)i < w) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

// ------------------------------------------------------------------------------------------
// record sources and key functors
// ------------------------------------------------------------------------------------------
// A source that is one of two SoA clouds, chosen on the device (passthrough decides whether
// clustering reads the centroids or the input cloud).
struct SoaPair {
    const float* ax; const float* ay; const float* az;   // used when !sel
    const float* bx; const float* by; const float* bz;   // used when sel
};

struct SoaSrc {   // pass 0: planes + implicit idx
    const float* x; const float* y; const float* z;
    __device__ __forceinline__ float4 load(uint32_t i) const { return make_float4(x[i], y[i], z[i], u2f(i)); }
};
struct RecSrc {   // later passes: 16-byte records
    const float4* r;
    __device__ __forceinline__ float4 load(uint32_t i) const { return r[i]; }
};

// PCL's linear voxel index (voxel_grid.hpp, "First pass"): ijk = (int)(floor(p*inv) - (float)min_b).
struct VoxelKey {
    float inv, fb0, fb1, fb2;
    uint32_t mul1, mul2;
    __device__ __forceinline__ void init(const DevState* st) {
        inv = st->inv_leaf;
        fb0 = (float)st->min_b[0]; fb1 = (float)st->min_b[1]; fb2 = (float)st->min_b[2];
        mul1 = st->mul[1]; mul2 = st->mul[2];
    }
    __device__ __forceinline__ uint32_t operator()(const float4& p) const {
        const int i0 = (int)(floorf(p.x * inv) - fb0);
        const int i1 = (int)(floorf(p.y * inv) - fb1);
        const int i2 = (int)(floorf(p.z * inv) - fb2);
        return (uint32_t)i0 + (uint32_t)i1 * mul1 + (uint32_t)i2 * mul2;
    }
};

// Coarse cell of edge g = 1/inv_cell.  (double)p * inv_cell is exact enough that the two
// geometric guarantees hold with a 0.5 % margin: same cell => d2 < r2, and d2 < r2 => cell
// offsets <= 2 per axis ("cell grid").
struct CellKey {
    double ig;
    long long c0, c1, c2;
    uint32_t dx, dy;
    __device__ __forceinline__ void init(const DevState* st) {
        ig = st->inv_cell; c0 = st->cmin[0]; c1 = st->cmin[1]; c2 = st->cmin[2];
        dx = st->cdim[0]; dy = st->cdim[1];
    }
    __device__ __forceinline__ uint32_t operator()(const float4& p) const {
        const uint32_t rx = (uint32_t)((long long)floor((double)p.x * ig) - c0);
        const uint32_t ry = (uint32_t)((long long)floor((double)p.y * ig) - c1);
        const uint32_t rz = (uint32_t)((long long)floor((double)p.z * ig) - c2);
        return rx + dx * (ry + dy * rz);
    }
};

// ------------------------------------------------------------------------------------------
// frame init + layout conversion
// ------------------------------------------------------------------------------------------
__global__ void k_frame_init(DevState* st, uint32_t n, uint32_t is_dense, float inv_leaf, float r2,
                             double inv_cell, uint32_t min_cluster, uint32_t max_cluster) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        DevState s;
        memset(&s, 0, sizeof(s));
        s.n_in = n; s.is_dense = is_dense; s.inv_leaf = inv_leaf; s.r2 = r2; s.inv_cell = inv_cell;
        s.min_cluster = min_cluster; s.max_cluster = max_cluster;
        *st = s;
    }
}

// Any point_step / field offsets / endianness -> SoA planes (the PointCloud2 -> PointXYZ
// conversion of od.cpp:688-689, on the device).
__global__ __launch_bounds__(kBlock) void k_unpack_points(const uint8_t* __restrict__ raw, uint32_t n,
                                                          uint32_t point_step, uint32_t off_x, uint32_t off_y,
                                                          uint32_t off_z, int bigendian, float* __restrict__ x,
                                                          float* __restrict__ y, float* __restrict__ z) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint8_t* p = raw + (size_t)i * point_step;
        uint32_t v[3];
        const uint32_t off[3] = {off_x, off_y, off_z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint8_t* q = p + off[a];
            uint32_t w = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            if (bigendian) w = __builtin_bswap32(w);
            v[a] = w;
        }
        x[i] = u2f(v[0]); y[i] = u2f(v[1]); z[i] = u2f(v[2]);
    }
}

// SoA planes -> packed output in the caller's layout (count read on the device).
__global__ __launch_bounds__(kBlock) void k_pack_points(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ z, const uint32_t* count,
                                                        int layout, uint32_t soa_stride, float* __restrict__ out) {
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (layout == 0) {
            reinterpret_cast<float4*>(out)[i] = make_float4(x[i], y[i], z[i], 1.0f);
        } else if (layout == 1) {
            out[i] = x[i]; out[soa_stride + i] = y[i]; out[2 * (size_t)soa_stride + i] = z[i];
        } else {
            out[3 * (size_t)i] = x[i]; out[3 * (size_t)i + 1] = y[i]; out[3 * (size_t)i + 2] = z[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// min / max (pcl::getMinMax3D) — block partials, finished by the set-up kernels
// ------------------------------------------------------------------------------------------
struct MinMaxPartial {
    float mn[3], mx[3];
    uint32_t valid, nonfinite;
};

// mode 0: voxel stage (source = input planes, count = n_in, honours is_dense)
// mode 1: cluster stage (source selected by st->c_from_input, count = st->c_n)
__global__ __launch_bounds__(kBlock) void k_minmax(SoaPair src, const DevState* st, int mode,
                                                   MinMaxPartial* __restrict__ partials) {
    __shared__ MinMaxPartial s_p[kWavesPerBlock];
    if (st->err) return;
    const bool sel = mode == 0 ? true : (st->c_from_input != 0);
    const float* x = sel ? src.bx : src.ax;
    const float* y = sel ? src.by : src.ay;
    const float* z = sel ? src.bz : src.az;
    const uint32_t n = mode == 0 ? st->n_in : st->c_n;
    const float fmax = 3.402823466e+38f;
    float mn0 = fmax, mn1 = fmax, mn2 = fmax, mx0 = -fmax, mx1 = -fmax, mx2 = -fmax;
    uint32_t valid = 0, bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float a = x[i], b = y[i], c = z[i];
        if (finite3(a, b, c)) {
            ++valid;
            mn0 = a < mn0 ? a : mn0; mx0 = a > mx0 ? a : mx0;
            mn1 = b < mn1 ? b : mn1; mx1 = b > mx1 ? b : mx1;
            mn2 = c < mn2 ? c : mn2; mx2 = c > mx2 ? c : mx2;
        } else {
            ++bad;
        }
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, d, kWave)); mx0 = fmaxf(mx0, __shfl_xor(mx0, d, kWave));
        mn1 = fminf(mn1, __shfl_xor(mn1, d, kWave)); mx1 = fmaxf(mx1, __shfl_xor(mx1, d, kWave));
        mn2 = fminf(mn2, __shfl_xor(mn2, d, kWave)); mx2 = fmaxf(mx2, __shfl_xor(mx2, d, kWave));
        valid += __shfl_xor(valid, d, kWave); bad += __shfl_xor(bad, d, kWave);
    }
    const uint32_t w = threadIdx.x / kWave;
    if (lane_id() == 0) {
        MinMaxPartial p;
        p.mn[0] = mn0; p.mn[1] = mn1; p.mn[2] = mn2; p.mx[0] = mx0; p.mx[1] = mx1; p.mx[2] = mx2;
        p.valid = valid; p.nonfinite = bad;
        s_p[w] = p;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPartial p = s_p[0];
        for (int i = 1; i < kWavesPerBlock; ++i) {
            for (int a = 0; a < 3; ++a) { p.mn[a] = fminf(p.mn[a], s_p[i].mn[a]); p.mx[a] = fmaxf(p.mx[a], s_p[i].mx[a]); }
            p.valid += s_p[i].valid; p.nonfinite += s_p[i].nonfinite;
        }
        partials[blockIdx.x] = p;
    }
}

__device__ __forceinline__ MinMaxPartial reduce_partials(const MinMaxPartial* partials, uint32_t nblocks,
                                                         MinMaxPartial* s_p) {
    // one block of kBlock threads
    const float fmax = 3.402823466e+38f;
    MinMaxPartial p;
    for (int a = 0; a < 3; ++a) { p.mn[a] = fmax; p.mx[a] = -fmax; }
    p.valid = 0; p.nonfinite = 0;
    for (uint32_t i = threadIdx.x; i < nblocks; i += blockDim.x) {
        const MinMaxPartial q = partials[i];
        for (int a = 0; a < 3; ++a) { p.mn[a] = fminf(p.mn[a], q.mn[a]); p.mx[a] = fmaxf(p.mx[a], q.mx[a]); }
        p.valid += q.valid; p.nonfinite += q.nonfinite;
    }
    s_p[threadIdx.x] = p;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            MinMaxPartial a = s_p[threadIdx.x];
            const MinMaxPartial b = s_p[threadIdx.x + d];
            for (int k = 0; k < 3; ++k) { a.mn[k] = fminf(a.mn[k], b.mn[k]); a.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
            a.valid += b.valid; a.nonfinite += b.nonfinite;
            s_p[threadIdx.x] = a;
        }
        __syncthreads();
    }
    return s_p[0];
}

__device__ __forceinline__ uint32_t bits_for(uint64_t count) {   // bits to index `count` values (>= 1)
    uint32_t b = 1;
    while (b < 64 && (1ull << b) < count) ++b;
    return b;
}

// Spread key_bits over the `npass` radix passes the host launched.  Returns false when a
// digit would be wider than kMaxBits (the host then re-runs the frame with more passes).
__device__ __forceinline__ bool make_plan(RadixPlan* pl, uint32_t key_bits, uint32_t npass) {
    pl->key_bits = key_bits;
    pl->npass = npass;
    const uint32_t each = (key_bits + npass - 1) / npass;
    uint32_t sh = 0;
    for (uint32_t i = 0; i < (uint32_t)kMaxPasses; ++i) {
        uint32_t b = 0;
        if (i < npass && sh < key_bits) b = (key_bits - sh < each) ? key_bits - sh : each;
        pl->bits[i] = b; pl->shift[i] = sh;
        sh += b;
    }
    return each <= (uint32_t)kMaxBits;
}

// VoxelGrid::applyFilter up to "divb_mul_ = ..." (SURVEY.md §8a-1 steps 1-4), one block.
__global__ __launch_bounds__(kBlock) void k_voxel_setup(DevState* st, const MinMaxPartial* partials,
                                                        uint32_t nblocks, uint32_t npass) {
    __shared__ MinMaxPartial s_p[kBlock];
    if (st->err) return;
    const MinMaxPartial p = reduce_partials(partials, nblocks, s_p);
    if (threadIdx.x != 0) return;
    if (st->is_dense && p.nonfinite) { st->err = kErrNonFinite; return; }
    st->n_valid = p.valid;
    for (int a = 0; a < 3; ++a) { st->vmin[a] = p.mn[a]; st->vmax[a] = p.mx[a]; }
    if (p.valid == 0) {   // empty cloud -> empty output
        for (int a = 0; a < 3; ++a) { st->min_b[a] = 0; st->div_b[a] = 1; st->mul[a] = 1; }
        make_plan(&st->plan_voxel, 1, npass);
        return;
    }
    const float inv = st->inv_leaf;
    // dx = (int64)((max - min) * inv) + 1; overflow rule: dx*dy*dz > INT32_MAX -> output = input
    long long d[3];
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        const float e = (p.mx[a] - p.mn[a]) * inv;
        if (!(e < 2147483648.0f) || e < 0.0f) { ok = false; d[a] = 1; } else d[a] = (long long)e + 1;
    }
    if (ok) {
        const long long lim = 2147483647ll;
        const long long dxy = d[0] * d[1];
        if (dxy > lim || dxy * d[2] > lim) ok = false;
    }
    if (!ok) {
        st->passthrough = 1;
        for (int a = 0; a < 3; ++a) { st->min_b[a] = 0; st->div_b[a] = 1; st->mul[a] = 1; }
        make_plan(&st->plan_voxel, 1, npass);
        return;
    }
    int32_t div[3];
    for (int a = 0; a < 3; ++a) {
        const int32_t lo = (int32_t)floorf(p.mn[a] * inv);
        const int32_t hi = (int32_t)floorf(p.mx[a] * inv);
        st->min_b[a] = lo;
        div[a] = hi - lo + 1;
        st->div_b[a] = div[a];
    }
    st->mul[0] = 1u; st->mul[1] = (uint32_t)div[0]; st->mul[2] = (uint32_t)div[0] * (uint32_t)div[1];
    const uint64_t cells = (uint64_t)(uint32_t)div[0] * (uint32_t)div[1] * (uint32_t)div[2];
    const uint32_t kb = cells > 0xFFFFFFFFull ? 32u : bits_for(cells);
    if (!make_plan(&st->plan_voxel, kb, npass)) st->err = kErrReplanVoxel;
}

// Cell grid for the neighbour search (stands in for KdTreeFLANN::setInputCloud).
__global__ __launch_bounds__(kBlock) void k_cell_setup(DevState* st, const MinMaxPartial* partials,
                                                       uint32_t nblocks, uint32_t npass) {
    __shared__ MinMaxPartial s_p[kBlock];
    if (st->err) return;
    const MinMaxPartial p = reduce_partials(partials, nblocks, s_p);
    if (threadIdx.x != 0) return;
    if (p.nonfinite) { st->err = kErrNonFinite; return; }   // radiusSearch on NaN is undefined in PCL
    if (st->c_n == 0) {
        for (int a = 0; a < 3; ++a) { st->cmin[a] = 0; st->cdim[a] = 1; }
        make_plan(&st->plan_cell, 1, npass);
        return;
    }
    const double ig = st->inv_cell;
    uint64_t cells = 1;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        const double lo = floor((double)p.mn[a] * ig), hi = floor((double)p.mx[a] * ig);
        const double ext = hi - lo + 1.0;
        if (!(ext < 4294967296.0) || !(fabs(lo) < 9.0e18) || !(fabs(hi) < 9.0e18)) { ok = false; break; }
        st->cmin[a] = (long long)lo;
        st->cdim[a] = (uint32_t)ext;
        if (cells > 0xFFFFFFFFull / (uint64_t)ext) { ok = false; break; }
        cells *= (uint64_t)ext;
    }
    if (!ok) { st->err = kErrRange; return; }
    if (!make_plan(&st->plan_cell, bits_for(cells), npass)) st->err = kErrReplanCell;
}

// After the voxel stage: decide what the cluster stage reads.
__global__ void k_cluster_source(DevState* st, int force_input) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (st->err) return;
        if (force_input) { st->c_from_input = 1; st->c_n = st->n_in; }
        else if (st->passthrough) { st->c_from_input = 1; st->c_n = st->n_in; }
        else { st->c_from_input = 0; st->c_n = st->m; }
    }
}

// ------------------------------------------------------------------------------------------
// stable LSD radix partition of records by a digit of their key
//   k_radix_hist    : per-tile digit counts            -> table[tile][bin]
//   k_table_scan    : exclusive scan down each column  -> table_x[tile][bin], bin_total[bin]
//   k_radix_scatter : stable ranks (wave match-any) + scatter
// which selects the key/count: 0 = voxel keys over n_in/n_valid, 1 = cell keys over c_n.
// ------------------------------------------------------------------------------------------
struct PassInfo {
    uint32_t count, bits, shift, nb;   // nb = (1<<bits) + 1 bins (last = trash)
    bool drop_nonfinite;
};

__device__ __forceinline__ PassInfo pass_info(const DevState* st, int which, int pass) {
    PassInfo pi;
    const RadixPlan& pl = which == 0 ? st->plan_voxel : st->plan_cell;
    pi.bits = pl.bits[pass]; pi.shift = pl.shift[pass];
    pi.nb = (1u << pi.bits) + 1u;
    if (which == 0) {
        pi.count = pass == 0 ? st->n_in : st->n_valid;
        pi.drop_nonfinite = pass == 0 && st->is_dense == 0;
    } else {
        pi.count = st->c_n;
        pi.drop_nonfinite = false;
    }
    return pi;
}

template <class Src, class KeyFn>
__global__ __launch_bounds__(kBlock) void k_radix_hist(Src src_a, Src src_b, const DevState* st, int which, int pass,
                                                       uint32_t* __restrict__ table) {
    __shared__ uint32_t s_h[kMaxBins];
    if (st->err || (which == 0 && st->passthrough)) return;
    const PassInfo pi = pass_info(st, which, pass);
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= pi.count) return;
    const bool sel_b = which == 1 && st->c_from_input != 0;
    const Src src = sel_b ? src_b : src_a;
    KeyFn key; key.init(st);
    for (uint32_t b = threadIdx.x; b < pi.nb; b += kBlock) s_h[b] = 0;
    __syncthreads();
    const uint32_t mask = (1u << pi.bits) - 1u;
    const uint32_t wbase = tile_base + (threadIdx.x / kWave) * (kItems * kWave) + lane_id();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < pi.count) {
            const float4 r = src.load(i);
            uint32_t d;
            if (pi.drop_nonfinite && !finite3(r.x, r.y, r.z)) d = pi.nb - 1;
            else d = (key(r) >> pi.shift) & mask;
            atomicAdd(&s_h[d], 1u);
        }
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * pi.nb;
    for (uint32_t b = threadIdx.x; b < pi.nb; b += kBlock) row[b] = s_h[b];
}

// which = 2: CSR partition (count = c_n, bits from plan_csr)
__device__ __forceinline__ bool table_dims(const DevState* st, int which, int pass, uint32_t* ntiles, uint32_t* nb) {
    if (which == 2) {
        *nb = (1u << st->plan_csr.bits[pass]) + 1u;
        *ntiles = (st->c_n + kTile - 1) / kTile;
        return st->n_kept != 0;
    }
    const PassInfo pi = pass_info(st, which, pass);
    *nb = pi.nb;
    *ntiles = (pi.count + kTile - 1) / kTile;
    return !(which == 0 && st->passthrough);
}

// Exclusive scan down each column of the tile-count table: block b owns bin b.
// grid = kMaxBins blocks (blocks past the pass's bin count exit at once).
__global__ __launch_bounds__(kBlock) void k_table_scan(const DevState* st, int which, int pass,
                                                       const uint32_t* __restrict__ table,
                                                       uint32_t* __restrict__ table_x,
                                                       uint32_t* __restrict__ bin_total) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err) return;
    uint32_t ntiles, nb;
    if (!table_dims(st, which, pass, &ntiles, &nb)) return;
    const uint32_t b = blockIdx.x;
    if (b >= nb) return;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += kBlock) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < ntiles ? table[(size_t)t * nb + b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, s_w, &tot);
        if (t < ntiles) table_x[(size_t)t * nb + b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) bin_total[b] = carry;
}

// Exclusive scan of bin_total[0..nb) into LDS s_base[0..nb); every thread of the block calls.
__device__ __forceinline__ void scan_bins_to_lds(const uint32_t* __restrict__ bin_total, uint32_t nb,
                                                 uint32_t* s_base, uint32_t* s_w) {
    constexpr int kPer = (kMaxBins + kBlock - 1) / kBlock;   // 9
    uint32_t v[kPer];
    uint32_t sum = 0;
    const uint32_t b0 = threadIdx.x * kPer;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint32_t b = b0 + k;
        v[k] = b < nb ? bin_total[b] : 0u;
        sum += v[k];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan(sum, s_w, &tot);
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint32_t b = b0 + k;
        if (b < nb) s_base[b] = run;
        run += v[k];
    }
    __syncthreads();
}

// Peers of this lane's digit among the active lanes of the wave (match-any over nbits bits).
__device__ __forceinline__ uint64_t match_digit(uint32_t d, uint32_t nbits, bool active) {
    uint64_t m = __ballot(active);
    for (uint32_t b = 0; b < nbits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

template <class Src, class KeyFn>
__global__ __launch_bounds__(kBlock) void k_radix_scatter(Src src_a, Src src_b, const DevState* st, int which,
                                                          int pass, const uint32_t* __restrict__ table_x,
                                                          const uint32_t* __restrict__ bin_total,
                                                          float4* __restrict__ out) {
    __shared__ uint32_t s_hist[kWavesPerBlock * kMaxBins];
    __shared__ uint32_t s_base[kMaxBins];
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err || (which == 0 && st->passthrough)) return;
    const PassInfo pi = pass_info(st, which, pass);
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= pi.count) return;
    const bool sel_b = which == 1 && st->c_from_input != 0;
    const Src src = sel_b ? src_b : src_a;
    KeyFn key; key.init(st);
    const uint32_t nb = pi.nb;
    for (uint32_t b = threadIdx.x; b < kWavesPerBlock * nb; b += kBlock) s_hist[b] = 0;
    scan_bins_to_lds(bin_total, nb, s_base, s_w);   // ends with a barrier

    // phase 1: each wave walks its contiguous sub-tile in order; ranks are stable by construction
    const uint32_t w = threadIdx.x / kWave;
    uint32_t* hist_w = s_hist + w * nb;
    const uint32_t mask = (1u << pi.bits) - 1u;
    const uint32_t wbase = tile_base + w * (kItems * kWave) + lane_id();
    float4 rec[kItems];
    uint32_t dig[kItems], lrank[kItems];
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        const bool active = i < pi.count;
        uint32_t d = 0;
        if (active) {
            rec[j] = src.load(i);
            if (pi.drop_nonfinite && !finite3(rec[j].x, rec[j].y, rec[j].z)) d = nb - 1;
            else d = (key(rec[j]) >> pi.shift) & mask;
        }
        const uint64_t peers = match_digit(d, pi.bits + 1, active);
        const uint32_t rank = __popcll(peers & lanemask_lt());
        uint32_t old = 0;
        if (active) old = hist_w[d];
        if (active && rank == 0) hist_w[d] = old + (uint32_t)__popcll(peers);
        dig[j] = d; lrank[j] = old + rank;
    }
    __syncthreads();
    // phase 2: per bin, turn the per-wave counts into per-wave output bases
    const uint32_t* row = table_x + (size_t)blockIdx.x * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) {
        uint32_t run = s_base[b] + row[b];
#pragma unroll
        for (int q = 0; q < kWavesPerBlock; ++q) {
            const uint32_t t = s_hist[q * nb + b];
            s_hist[q * nb + b] = run;
            run += t;
        }
    }
    __syncthreads();
    // phase 3: scatter
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < pi.count) out[hist_w[dig[j]] + lrank[j]] = rec[j];
    }
}

// ------------------------------------------------------------------------------------------
// run heads over sorted records: count per tile, scan of tile counts, then consumers
// ------------------------------------------------------------------------------------------
// count of the sorted range: which 0 -> n_valid (voxel), 1 -> c_n (cells)
__device__ __forceinline__ uint32_t sorted_count(const DevState* st, int which) {
    return which == 0 ? st->n_valid : st->c_n;
}

template <class KeyFn>
__global__ __launch_bounds__(kBlock) void k_heads_count(const float4* __restrict__ rec, const DevState* st, int which,
                                                        uint32_t* __restrict__ tile_heads) {
    __shared__ uint32_t s_w[kWavesPerBlock];
    if (st->err || (which == 0 && st->passthrough)) return;
    const uint32_t n = sorted_count(st, which);
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    KeyFn key; key.init(st);
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = base + threadIdx.x * kScanItems + k;
        if (i < n) c += (i == 0 || key(rec[i]) != key(rec[i - 1])) ? 1u : 0u;
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) c += __shfl_xor(c, d, kWave);
    if (lane_id() == 0) s_w[threadIdx.x / kWave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < kWavesPerBlock; ++i) t += s_w[i];
        tile_heads[blockIdx.x] = t;
    }
}

// One block: exclusive scan of per-tile values (in place); total -> *total_out (device word).
// ntiles_of: 0 -> ceil(n_valid/kScanTile), 1 -> ceil(c_n/kScanTile)
__global__ __launch_bounds__(kBlock) void k_scan_tiles(uint32_t* __restrict__ vals, DevState* st, int which,
                                                       int dst /*0: m, 1: n_cells*/) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err || (which == 0 && st->passthrough)) return;
    const uint32_t n = sorted_count(st, which);
    const uint32_t ntiles = (n + kScanTile - 1) / kScanTile;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntiles ? vals[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, s_w, &tot);
        if (i < ntiles) vals[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        if (dst == 0) {
            st->m = carry;
        } else {
            st->n_cells = carry;
            uint32_t cap = 1024;
            while (cap < 2u * carry && cap < 0x80000000u) cap <<= 1;
            st->hash_mask = cap - 1u;
        }
    }
}

// Cell hash table: slot = (key+1) | cell_id << 32, 0 = empty.  Open addressing, linear probing.
__device__ __forceinline__ uint32_t cell_hash(uint32_t key) {
    uint32_t h = key * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}

__global__ __launch_bounds__(kBlock) void k_hash_clear(const DevState* st, unsigned long long* __restrict__ tab) {
    if (st->err) return;
    const uint32_t cap = st->hash_mask + 1u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) tab[i] = 0ull;
}

__device__ __forceinline__ void hash_insert(unsigned long long* tab, uint32_t mask, uint32_t key, uint32_t id) {
    const unsigned long long v = (unsigned long long)(key + 1u) | ((unsigned long long)id << 32);
    uint32_t slot = cell_hash(key) & mask;
    while (atomicCAS(tab + slot, 0ull, v) != 0ull) slot = (slot + 1u) & mask;
}

// returns the cell id or 0xFFFFFFFF
__device__ __forceinline__ uint32_t hash_lookup(const unsigned long long* __restrict__ tab, uint32_t mask, uint32_t key) {
    uint32_t slot = cell_hash(key) & mask;
    while (true) {
        const unsigned long long v = tab[slot];
        if (v == 0ull) return 0xFFFFFFFFu;
        if ((uint32_t)v == key + 1u) return (uint32_t)(v >> 32);
        slot = (slot + 1u) & mask;
    }
}

// Voxel centroids (voxel_grid.hpp "Fourth pass", PCL >= 1.8 CentroidPoint / AccumulatorXYZ):
// the thread that owns a run head sums the run in order (= ascending point index, the
// partition is stable) starting from 0.0f and divides by float(count).
__global__ __launch_bounds__(kBlock) void k_voxel_reduce(const float4* __restrict__ rec, const DevState* st,
                                                         const uint32_t* __restrict__ tile_heads_x,
                                                         float* __restrict__ cx, float* __restrict__ cy,
                                                         float* __restrict__ cz) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err || st->passthrough) return;
    const uint32_t n = st->n_valid;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    VoxelKey key; key.init(st);
    uint32_t kcur[kScanItems];
    bool head[kScanItems];
    uint32_t c = 0;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
    uint32_t kprev = (i0 > 0 && i0 <= n) ? key(rec[i0 - 1]) : 0u;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        head[k] = false;
        if (i < n) {
            kcur[k] = key(rec[i]);
            head[k] = (i == 0) || (kcur[k] != kprev);
            kprev = kcur[k];
            c += head[k] ? 1u : 0u;
        }
    }
    uint32_t tot;
    uint32_t rank = tile_heads_x[blockIdx.x] + block_excl_scan(c, s_w, &tot);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        if (i < n && head[k]) {
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
            uint32_t j = i;
            float4 r = rec[j];
            while (true) {
                sx += r.x; sy += r.y; sz += r.z;
                ++j;
                if (j >= n) break;
                r = rec[j];
                if (key(r) != kcur[k]) break;
            }
            const float cnt = (float)(j - i);
            cx[rank] = sx / cnt; cy[rank] = sy / cnt; cz[rank] = sz / cnt;
            ++rank;
        }
    }
}

// Passthrough: "output = *input_" — the voxel cloud is the input cloud, every point kept.
__global__ __launch_bounds__(kBlock) void k_voxel_passthrough(DevState* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (st->err) return;
        if (st->passthrough) st->m = st->n_in;
    }
}

// Cells: for every sorted record its cell id; for every cell its key, first record and the
// smallest point index it holds (= the first record's idx: the partition is stable).
__global__ __launch_bounds__(kBlock) void k_cells_build(const float4* __restrict__ rec, const DevState* st,
                                                        const uint32_t* __restrict__ tile_heads_x,
                                                        uint32_t* __restrict__ rec_cell, uint32_t* __restrict__ cell_key,
                                                        uint32_t* __restrict__ cell_start, uint32_t* __restrict__ cell_minidx,
                                                        uint32_t* __restrict__ parent, uint32_t* __restrict__ comp_size,
                                                        uint32_t* __restrict__ comp_min, uint32_t* __restrict__ kept_val,
                                                        unsigned long long* __restrict__ hash_tab) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err) return;
    const uint32_t n = st->c_n;
    const uint32_t hmask = st->hash_mask;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    CellKey key; key.init(st);
    uint32_t kcur[kScanItems];
    bool head[kScanItems];
    uint32_t c = 0;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
    uint32_t kprev = (i0 > 0 && i0 <= n) ? key(rec[i0 - 1]) : 0u;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        head[k] = false;
        if (i < n) {
            kcur[k] = key(rec[i]);
            head[k] = (i == 0) || (kcur[k] != kprev);
            kprev = kcur[k];
            c += head[k] ? 1u : 0u;
        }
    }
    uint32_t tot;
    uint32_t rank = tile_heads_x[blockIdx.x] + block_excl_scan(c, s_w, &tot);   // heads before this thread
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        if (i < n) {
            if (head[k]) {
                cell_key[rank] = kcur[k];
                cell_start[rank] = i;
                cell_minidx[rank] = f2u(rec[i].w);
                parent[rank] = rank;
                comp_size[rank] = 0u;
                comp_min[rank] = 0xFFFFFFFFu;
                hash_insert(hash_tab, hmask, kcur[k], rank);
                ++rank;
            }
            rec_cell[i] = rank - 1;
            kept_val[i] = 0u;
            if (i == n - 1) cell_start[st->n_cells] = n;   // sentinel (n_cells set by k_scan_tiles)
        }
    }
}

// ------------------------------------------------------------------------------------------
// union-find over cells (lock-free, roots are the smallest cell id of their component)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t x) {
    while (true) {
        const uint32_t p = ld_agent(parent + x);
        if (p == x) return x;
        const uint32_t gp = ld_agent(parent + p);
        if (gp == p) return p;
        st_agent(parent + x, gp);   // path halving: gp is always an ancestor of x
        x = gp;
    }
}

__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        // hang the larger root b under the smaller root a
        const uint32_t old = atomicCAS(parent + b, b, a);
        if (old == b) return;
    }
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// FLANN's L2_Simple: ((dx*dx + dy*dy) + dz*dz) in float32, no contraction.
__device__ __forceinline__ float l2_simple(const float4& a, const float4& b) {
    float result = 0.0f;
    float d = a.x - b.x; result += d * d;
    d = a.y - b.y; result += d * d;
    d = a.z - b.z; result += d * d;
    return result;
}

// ------------------------------------------------------------------------------------------
// cell graph -> components.  Points of one cell are mutually within the radius (clique), and a
// neighbour of a point of A lies in a cell at most 2 away per axis, so the components of the
// point graph {d2 < r2} are the components of the cell graph whose edges are "some cross
// pair satisfies d2 < r2".  One wave per cell; lane t owns stencil offset t of one half of the
// 5x5x5 stencil (62 offsets, ascending in cell key), found through the cell hash table.
//   k_cell_hook    : backward half, no atomics: parent[c] = smallest adjacent cell id (or c)
//   k_cell_compress: pointer jumping, parent[c] = root
//   k_cell_union   : forward half: pairs whose roots already agree are skipped without a
//                    distance test; the rest are tested and united by CAS (few survive the hook)
// ------------------------------------------------------------------------------------------
constexpr int kCellPtsCap = 64;
constexpr int kStencil = 62;

// forward offset number t (0..61), ascending in key: (0,0,+1..2) ; (0,+1..2,*) ; (+1..2,*,*)
__device__ __forceinline__ void stencil_offset(uint32_t t, int* dx, int* dy, int* dz) {
    if (t < 2) { *dz = 0; *dy = 0; *dx = (int)t + 1; return; }
    const uint32_t u = t - 2;
    if (u < 10) { *dz = 0; *dy = 1 + (int)(u / 5); *dx = (int)(u % 5) - 2; return; }
    const uint32_t v = u - 10;
    *dz = 1 + (int)(v / 25); *dy = (int)((v % 25) / 5) - 2; *dx = (int)(v % 5) - 2;
}

struct CellGrid {
    uint32_t dimx, dimy, dimz, hmask;
    const unsigned long long* tab;
    // cell id at (rx,ry,rz) + lane's offset of the forward (sign=+1) or backward (-1) half-stencil
    __device__ __forceinline__ uint32_t neighbour(uint32_t rx, uint32_t ry, uint32_t rz, uint32_t lane, int sign) const {
        if (lane >= (uint32_t)kStencil) return 0xFFFFFFFFu;
        int dx, dy, dz;
        // backward candidates in ascending key order = negated forward offsets in reverse order
        stencil_offset(sign > 0 ? lane : (uint32_t)(kStencil - 1) - lane, &dx, &dy, &dz);
        const long long nx = (long long)rx + sign * dx, ny = (long long)ry + sign * dy, nz = (long long)rz + sign * dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= (long long)dimx || ny >= (long long)dimy || nz >= (long long)dimz)
            return 0xFFFFFFFFu;
        const uint32_t key = (uint32_t)nx + dimx * ((uint32_t)ny + dimy * (uint32_t)nz);
        return hash_lookup(tab, hmask, key);
    }
};

// does some point of B (this lane's cell) lie within the radius of some point of A (in LDS)?
__device__ __forceinline__ bool lane_pair_hit(const float4* s_pa, uint32_t an, const float4* __restrict__ rec,
                                              uint32_t b0, uint32_t bn, float r2) {
    for (uint32_t ib = 0; ib < bn; ++ib) {
        const float4 pb = rec[b0 + ib];
        for (uint32_t ia = 0; ia < an; ++ia)
            if (l2_simple(s_pa[ia], pb) < r2) return true;
    }
    return false;
}

// whole wave on one cell pair: lanes spread over B's points
__device__ __forceinline__ bool wave_pair_hit(const float4* __restrict__ rec, uint32_t a0, uint32_t an, uint32_t b0,
                                              uint32_t bn, float r2) {
    bool hit = false;
    const uint32_t lane = lane_id();
    for (uint32_t ia = 0; ia < an && !hit; ++ia) {
        const float4 pa = rec[a0 + ia];
        for (uint32_t jb = 0; jb < bn && !hit; jb += kWave) {
            const uint32_t ib = jb + lane;
            bool h = false;
            if (ib < bn) h = l2_simple(pa, rec[b0 + ib]) < r2;
            hit = __any(h);
        }
    }
    return hit;
}

__global__ __launch_bounds__(kBlock) void k_cell_hook(const float4* __restrict__ rec, const DevState* st,
                                                      const uint32_t* __restrict__ cell_key,
                                                      const uint32_t* __restrict__ cell_start,
                                                      const unsigned long long* __restrict__ hash_tab,
                                                      uint32_t* __restrict__ parent) {
    __shared__ float4 s_pa[kWavesPerBlock][kCellPtsCap];
    if (st->err) return;
    const uint32_t ncell = st->n_cells;
    CellGrid g{st->cdim[0], st->cdim[1], st->cdim[2], st->hash_mask, hash_tab};
    const float r2 = st->r2;
    const uint32_t w = threadIdx.x / kWave;
    const uint32_t lane = lane_id();
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    for (uint32_t a = blockIdx.x * kWavesPerBlock + w; a < ncell; a += nwaves) {
        const uint32_t ka = cell_key[a];
        const uint32_t rx = ka % g.dimx, ry = (ka / g.dimx) % g.dimy, rz = ka / (g.dimx * g.dimy);
        const uint32_t b = g.neighbour(rx, ry, rz, lane, -1);
        const bool found = b != 0xFFFFFFFFu;
        uint64_t found_mask = __ballot(found);
        uint32_t target = a;
        if (found_mask != 0) {
            const uint32_t a0 = cell_start[a], an = cell_start[a + 1] - a0;
            uint32_t b0 = 0, bn = 0;
            if (found) { b0 = cell_start[b]; bn = cell_start[b + 1] - b0; }
            uint32_t bn_max = bn;
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) bn_max = max(bn_max, (uint32_t)__shfl_xor(bn_max, d, kWave));
            if (an <= (uint32_t)kCellPtsCap && bn_max <= (uint32_t)kCellPtsCap) {
                if (lane < an) s_pa[w][lane] = rec[a0 + lane];
                __builtin_amdgcn_wave_barrier();
                const bool hit = found && lane_pair_hit(s_pa[w], an, rec, b0, bn, r2);
                const uint64_t hits = __ballot(hit);
                if (hits) target = __shfl(b, __ffsll((unsigned long long)hits) - 1, kWave);   // lanes ascend in key
                __builtin_amdgcn_wave_barrier();
            } else {
                while (found_mask) {   // ascending key: the first hit is the smallest adjacent id
                    const int t = __ffsll((unsigned long long)found_mask) - 1;
                    found_mask &= found_mask - 1;
                    const uint32_t cb = __shfl(b, t, kWave), cb0 = __shfl(b0, t, kWave), cbn = __shfl(bn, t, kWave);
                    if (wave_pair_hit(rec, a0, an, cb0, cbn, r2)) { target = cb; break; }
                }
            }
        }
        if (lane == 0) parent[a] = target;   // target <= a: a forest, no cycles
    }
}

__global__ __launch_bounds__(kBlock) void k_cell_compress(const DevState* st, uint32_t* __restrict__ parent) {
    if (st->err) return;
    const uint32_t ncell = st->n_cells;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) {
        const uint32_t r = uf_find(parent, c);
        st_agent(parent + c, r);
    }
}

__global__ __launch_bounds__(kBlock) void k_cell_union(const float4* __restrict__ rec, const DevState* st,
                                                       const uint32_t* __restrict__ cell_key,
                                                       const uint32_t* __restrict__ cell_start,
                                                       const unsigned long long* __restrict__ hash_tab,
                                                       uint32_t* __restrict__ parent) {
    __shared__ float4 s_pa[kWavesPerBlock][kCellPtsCap];
    if (st->err) return;
    const uint32_t ncell = st->n_cells;
    CellGrid g{st->cdim[0], st->cdim[1], st->cdim[2], st->hash_mask, hash_tab};
    const float r2 = st->r2;
    const uint32_t w = threadIdx.x / kWave;
    const uint32_t lane = lane_id();
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    for (uint32_t a = blockIdx.x * kWavesPerBlock + w; a < ncell; a += nwaves) {
        const uint32_t ka = cell_key[a];
        const uint32_t rx = ka % g.dimx, ry = (ka / g.dimx) % g.dimy, rz = ka / (g.dimx * g.dimy);
        const uint32_t b = g.neighbour(rx, ry, rz, lane, +1);
        bool need = false;
        if (b != 0xFFFFFFFFu) need = uf_find(parent, a) != uf_find(parent, b);
        uint64_t need_mask = __ballot(need);
        if (need_mask == 0) continue;
        const uint32_t a0 = cell_start[a], an = cell_start[a + 1] - a0;
        uint32_t b0 = 0, bn = 0;
        if (need) { b0 = cell_start[b]; bn = cell_start[b + 1] - b0; }
        uint32_t bn_max = bn;
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) bn_max = max(bn_max, (uint32_t)__shfl_xor(bn_max, d, kWave));
        if (an <= (uint32_t)kCellPtsCap && bn_max <= (uint32_t)kCellPtsCap) {
            if (lane < an) s_pa[w][lane] = rec[a0 + lane];
            __builtin_amdgcn_wave_barrier();
            if (need && lane_pair_hit(s_pa[w], an, rec, b0, bn, r2)) uf_union(parent, a, b);
            __builtin_amdgcn_wave_barrier();
        } else {
            while (need_mask) {
                const int t = __ffsll((unsigned long long)need_mask) - 1;
                need_mask &= need_mask - 1;
                const uint32_t cb = __shfl(b, t, kWave), cb0 = __shfl(b0, t, kWave), cbn = __shfl(bn, t, kWave);
                uint32_t ra = 0, rb = 1;
                if (lane == 0) { ra = uf_find(parent, a); rb = uf_find(parent, cb); }
                ra = __shfl(ra, 0, kWave); rb = __shfl(rb, 0, kWave);
                if (ra == rb) continue;
                if (wave_pair_hit(rec, a0, an, cb0, cbn, r2) && lane == 0) uf_union(parent, a, cb);
            }
        }
    }
}

// Component of every cell; component size and smallest point index (its discovery key).
// Cells are handled 64 at a time; lanes whose cells share a root combine their contribution
// in registers first, so a component costs one atomic pair per wave, not per cell.
__global__ __launch_bounds__(kBlock) void k_cell_flatten(const DevState* st, const uint32_t* __restrict__ cell_start,
                                                         const uint32_t* __restrict__ cell_minidx,
                                                         uint32_t* __restrict__ parent, uint32_t* __restrict__ cell_root,
                                                         uint32_t* __restrict__ comp_size, uint32_t* __restrict__ comp_min) {
    if (st->err) return;
    const uint32_t ncell = st->n_cells;
    const uint32_t lane = lane_id();
    const uint32_t nthreads = gridDim.x * blockDim.x;
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) - lane; base < ncell; base += nthreads) {
        const uint32_t c = base + lane;
        const bool valid = c < ncell;
        uint32_t r = 0xFFFFFFFFu, cnt = 0, mn = 0xFFFFFFFFu;
        if (valid) {
            r = uf_find(parent, c);
            cell_root[c] = r;
            cnt = cell_start[c + 1] - cell_start[c];
            mn = cell_minidx[c];
        }
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t rl = __shfl(r, leader, kWave);
            const bool mine = valid && r == rl;
            uint32_t c2 = mine ? cnt : 0u, m2 = mine ? mn : 0xFFFFFFFFu;
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) {
                c2 += __shfl_xor(c2, d, kWave);
                m2 = min(m2, (uint32_t)__shfl_xor(m2, d, kWave));
            }
            if ((int)lane == leader) { atomicAdd(comp_size + rl, c2); atomicMin(comp_min + rl, m2); }
            todo &= ~__ballot(mine);
        }
    }
}

// extractEuclideanClusters' size filter: kept_val[first point of component] = size if kept.
__global__ __launch_bounds__(kBlock) void k_mark_kept(const DevState* st, const uint32_t* __restrict__ cell_root,
                                                      const uint32_t* __restrict__ comp_size,
                                                      const uint32_t* __restrict__ comp_min,
                                                      uint32_t* __restrict__ kept_val) {
    if (st->err) return;
    const uint32_t ncell = st->n_cells;
    const uint32_t lo = st->min_cluster, hi = st->max_cluster;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) {
        if (cell_root[c] != c) continue;
        const uint32_t s = comp_size[c];
        if (s >= lo && s <= hi) kept_val[comp_min[c]] = s;
    }
}

// Two-value scan over point-index space: (number of kept clusters, sum of their sizes).
__global__ __launch_bounds__(kBlock) void k_kept_count(const DevState* st, const uint32_t* __restrict__ kept_val,
                                                       uint2* __restrict__ tile_sums) {
    __shared__ uint32_t s_a[kWavesPerBlock], s_b[kWavesPerBlock];
    if (st->err) return;
    const uint32_t n = st->c_n;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    uint32_t c = 0, s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = base + threadIdx.x * kScanItems + k;
        if (i < n) { const uint32_t v = kept_val[i]; c += v ? 1u : 0u; s += v; }
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) { c += __shfl_xor(c, d, kWave); s += __shfl_xor(s, d, kWave); }
    if (lane_id() == 0) { s_a[threadIdx.x / kWave] = c; s_b[threadIdx.x / kWave] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ta = 0, tb = 0;
        for (int i = 0; i < kWavesPerBlock; ++i) { ta += s_a[i]; tb += s_b[i]; }
        tile_sums[blockIdx.x] = make_uint2(ta, tb);
    }
}

__global__ __launch_bounds__(kBlock) void k_kept_scan_tiles(uint2* __restrict__ tile_sums, DevState* st,
                                                            uint32_t* __restrict__ offsets_out, uint32_t npass_csr) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err) return;
    const uint32_t n = st->c_n;
    const uint32_t ntiles = (n + kScanTile - 1) / kScanTile;
    uint32_t ca = 0, cb = 0;
    for (uint32_t base = 0; base < ntiles; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const uint2 v = i < ntiles ? tile_sums[i] : make_uint2(0u, 0u);
        uint32_t ta, tb;
        const uint32_t ea = block_excl_scan(v.x, s_w, &ta);
        const uint32_t eb = block_excl_scan(v.y, s_w, &tb);
        if (i < ntiles) tile_sums[i] = make_uint2(ca + ea, cb + eb);
        ca += ta; cb += tb;
    }
    if (threadIdx.x == 0) {
        st->n_kept = ca; st->n_kept_pts = cb;
        offsets_out[ca] = cb;
        // keys of the CSR partition are 0..n_kept (n_kept = "not in a kept cluster")
        if (!make_plan(&st->plan_csr, bits_for((uint64_t)ca + 1), npass_csr)) st->err = kErrReplanCsr;
    }
}

// rank_at[first point] = discovery rank of the kept cluster; offsets_out[rank] = CSR offset.
__global__ __launch_bounds__(kBlock) void k_kept_apply(const DevState* st, const uint32_t* __restrict__ kept_val,
                                                       const uint2* __restrict__ tile_sums_x,
                                                       uint32_t* __restrict__ rank_at, uint32_t* __restrict__ offsets_out) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err) return;
    const uint32_t n = st->c_n;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    uint32_t v[kScanItems];
    uint32_t c = 0, s = 0;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        v[k] = i < n ? kept_val[i] : 0u;
        c += v[k] ? 1u : 0u; s += v[k];
    }
    uint32_t tot;
    const uint2 tb = tile_sums_x[blockIdx.x];
    uint32_t rc = tb.x + block_excl_scan(c, s_w, &tot);
    uint32_t rs = tb.y + block_excl_scan(s, s_w, &tot);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (v[k]) { rank_at[i0 + k] = rc; offsets_out[rc] = rs; ++rc; rs += v[k]; }
    }
}

// label[point] = rank of its cluster, or n_kept when its component was filtered out.
__global__ __launch_bounds__(kBlock) void k_label_points(const float4* __restrict__ rec, const DevState* st,
                                                         const uint32_t* __restrict__ rec_cell,
                                                         const uint32_t* __restrict__ cell_root,
                                                         const uint32_t* __restrict__ comp_size,
                                                         const uint32_t* __restrict__ comp_min,
                                                         const uint32_t* __restrict__ rank_at,
                                                         uint32_t* __restrict__ label) {
    if (st->err) return;
    const uint32_t n = st->c_n;
    const uint32_t lo = st->min_cluster, hi = st->max_cluster, trash = st->n_kept;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t r = cell_root[rec_cell[i]];
        const uint32_t s = comp_size[r];
        const uint32_t idx = f2u(rec[i].w);
        label[idx] = (s >= lo && s <= hi) ? rank_at[comp_min[r]] : trash;
    }
}

// ------------------------------------------------------------------------------------------
// CSR: stable partition of point indices 0..c_n-1 by label (LSD over plan_csr digits).
// Pass 0 reads the identity permutation, later passes read the previous output.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool csr_pass_info(const DevState* st, int pass, uint32_t* count, uint32_t* bits,
                                              uint32_t* shift) {
    *count = st->c_n; *bits = st->plan_csr.bits[pass]; *shift = st->plan_csr.shift[pass];
    return st->n_kept != 0;
}

__global__ __launch_bounds__(kBlock) void k_csr_hist(const uint32_t* __restrict__ idx_in, const uint32_t* __restrict__ label,
                                                     const DevState* st, int pass, uint32_t* __restrict__ table) {
    __shared__ uint32_t s_h[kMaxBins];
    if (st->err) return;
    uint32_t count, bits, shift;
    if (!csr_pass_info(st, pass, &count, &bits, &shift)) return;
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= count) return;
    const uint32_t nb = (1u << bits) + 1u;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) s_h[b] = 0;
    __syncthreads();
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t wbase = tile_base + (threadIdx.x / kWave) * (kItems * kWave) + lane_id();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < count) {
            const uint32_t p = pass == 0 ? i : idx_in[i];
            atomicAdd(&s_h[(label[p] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) row[b] = s_h[b];
}

__global__ __launch_bounds__(kBlock) void k_csr_scatter(const uint32_t* __restrict__ idx_in, const uint32_t* __restrict__ label,
                                                        const DevState* st, int pass, const uint32_t* __restrict__ table_x,
                                                        const uint32_t* __restrict__ bin_total, uint32_t* __restrict__ idx_out) {
    __shared__ uint32_t s_hist[kWavesPerBlock * kMaxBins];
    __shared__ uint32_t s_base[kMaxBins];
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    if (st->err) return;
    uint32_t count, bits, shift;
    if (!csr_pass_info(st, pass, &count, &bits, &shift)) return;
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= count) return;
    const uint32_t nb = (1u << bits) + 1u;
    for (uint32_t b = threadIdx.x; b < kWavesPerBlock * nb; b += kBlock) s_hist[b] = 0;
    scan_bins_to_lds(bin_total, nb, s_base, s_w);
    const uint32_t w = threadIdx.x / kWave;
    uint32_t* hist_w = s_hist + w * nb;
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t wbase = tile_base + w * (kItems * kWave) + lane_id();
    uint32_t pt[kItems], dig[kItems], lrank[kItems];
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        const bool active = i < count;
        uint32_t d = 0;
        pt[j] = 0;
        if (active) {
            pt[j] = pass == 0 ? i : idx_in[i];
            d = (label[pt[j]] >> shift) & mask;
        }
        const uint64_t peers = match_digit(d, bits, active);
        const uint32_t rank = __popcll(peers & lanemask_lt());
        uint32_t old = 0;
        if (active) old = hist_w[d];
        if (active && rank == 0) hist_w[d] = old + (uint32_t)__popcll(peers);
        dig[j] = d; lrank[j] = old + rank;
    }
    __syncthreads();
    const uint32_t* row = table_x + (size_t)blockIdx.x * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) {
        uint32_t run = s_base[b] + row[b];
#pragma unroll
        for (int q = 0; q < kWavesPerBlock; ++q) {
            const uint32_t t = s_hist[q * nb + b];
            s_hist[q * nb + b] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < count) idx_out[hist_w[dig[j]] + lrank[j]] = pt[j];
    }
}

// ------------------------------------------------------------------------------------------
// PointWithRad per cluster (msg/PointWithRad.msg): one wave per cluster.
// centre = double mean rounded to float; r = max calculate_distance() (od.cpp:457-464).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cluster_summaries(SoaPair src, const DevState* st,
                                                              const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ indices,
                                                              float4* __restrict__ xyzr) {
    if (st->err) return;
    const bool sel = st->c_from_input != 0;
    const float* x = sel ? src.bx : src.ax;
    const float* y = sel ? src.by : src.ay;
    const float* z = sel ? src.bz : src.az;
    const uint32_t k = st->n_kept;
    const uint32_t lane = lane_id();
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    for (uint32_t c = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; c < k; c += nwaves) {
        const uint32_t b = offsets[c], e = offsets[c + 1];
        double sx = 0, sy = 0, sz = 0;
        for (uint32_t j = b + lane; j < e; j += kWave) { const uint32_t p = indices[j]; sx += x[p]; sy += y[p]; sz += z[p]; }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            sx += __shfl_xor(sx, d, kWave); sy += __shfl_xor(sy, d, kWave); sz += __shfl_xor(sz, d, kWave);
        }
        const double cnt = (double)(e - b);
        const float mx = (float)(sx / cnt), my = (float)(sy / cnt), mz = (float)(sz / cnt);
        float r = 0.0f;
        for (uint32_t j = b + lane; j < e; j += kWave) {
            const uint32_t p = indices[j];
            const float dx = x[p] - mx, dy = y[p] - my, dz = z[p] - mz;
            const float xd = dx * dx, yd = dy * dy, zd = dz * dz;     // (float)pow(d, 2): exact square, rounded once
            const float d = sqrtf((xd + yd) + zd);
            r = fmaxf(r, d);
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) r = fmaxf(r, __shfl_xor(r, d, kWave));
        if (lane == 0) xyzr[c] = make_float4(mx, my, mz, r);
    }
}

}  // namespace pcop

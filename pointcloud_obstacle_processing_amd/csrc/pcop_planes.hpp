// pcop_planes.hpp - RANSAC plane removal (replaces segment_plane_and_extract_indices, od.cpp:342-428): the two passes over the
// cloud that cost something run here, the sequential parts (the fixed-seed sampling, the plane through three points, the
// least-squares refit of pcl::SACSegmentation's optimizeModelCoefficients) stay on the host (pcop_capi.hip).
// Part of pcop_device.hpp (included from there, in this order; not a stand-alone header).
#pragma once

namespace pcop {

constexpr uint32_t kPlaneModelsMax = 64;   // candidate planes counted in one pass (RANSAC runs at most 51 iterations: max_iterations_ 50)

struct PlaneModels {
    float c[kPlaneModelsMax][4];
    uint32_t valid[kPlaneModelsMax];   // SampleConsensusModelPerpendicularPlane::isModelValid (decided on the host): an invalid model counts 0
    uint32_t count;
};

// SampleConsensusModelPlane::countWithinDistance / selectWithinDistance: |a x + b y + c z + d| < threshold, the sum in
// float32, left to right, compared in double (the threshold is PCL's double).
__device__ __forceinline__ bool plane_within(const float* c, float x, float y, float z, double threshold) {
    const float d = ((c[0] * x + c[1] * y) + c[2] * z) + c[3];
    return fabs((double)d) < threshold;
}

// Inlier counts of up to kPlaneModelsMax candidate planes in one pass over the cloud (RandomSampleConsensus evaluates them
// one after the other; the counts do not depend on each other, only the loop's end does, and the host replays that).
__global__ __launch_bounds__(kBlock) void k_plane_count(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                        uint32_t n, PlaneModels models, double threshold, uint32_t* __restrict__ counts) {
    __shared__ float s_c[kPlaneModelsMax][4];
    __shared__ uint32_t s_cnt[kWavesPerBlock][kPlaneModelsMax];
    for (uint32_t i = threadIdx.x; i < kPlaneModelsMax * 4u; i += kBlock) s_c[i / 4u][i % 4u] = models.c[i / 4u][i % 4u];
    for (uint32_t i = threadIdx.x; i < kWavesPerBlock * kPlaneModelsMax; i += kBlock) s_cnt[i / kPlaneModelsMax][i % kPlaneModelsMax] = 0u;
    __syncthreads();
    const uint32_t w = threadIdx.x / kWave, nm = models.count;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x + w * kWave; base < n; base += nthreads) {   // whole waves: the ballots need every lane
        const uint32_t i = base + lane_id();
        const bool in = i < n;
        const float px = in ? x[i] : 0.0f, py = in ? y[i] : 0.0f, pz = in ? z[i] : 0.0f;
        for (uint32_t m = 0; m < nm; ++m) {
            const uint64_t hit = __ballot(in && plane_within(s_c[m], px, py, pz, threshold));
            if (lane_id() == 0) s_cnt[w][m] += (uint32_t)__popcll(hit);
        }
    }
    __syncthreads();
    for (uint32_t m = threadIdx.x; m < nm; m += kBlock) {
        uint32_t t = 0;
#pragma unroll
        for (int q = 0; q < kWavesPerBlock; ++q) t += s_cnt[q][m];
        if (t && models.valid[m]) atomicAdd(counts + m, t);
    }
}

// pcl::ExtractIndices, negative: the points outside the plane, in order, into (ox, oy, oz); *n_out their number.
// Tiles publish their counts and add up the lower tiles (tile_pub: zeroed by the host, one word per kScanTile points).
__global__ __launch_bounds__(kBlock) void k_plane_split(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                        uint32_t n, float c0, float c1, float c2, float c3, double threshold,
                                                        float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                                        uint32_t* __restrict__ tile_pub, uint32_t* __restrict__ n_out, uint32_t* __restrict__ fail) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    const float c[4] = {c0, c1, c2, c3};
    float px[kScanItems], py[kScanItems], pz[kScanItems];
    bool keep[kScanItems];
    uint32_t cnt = 0;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = min(i0 + k, n - 1u);
        px[k] = x[i]; py[k] = y[i]; pz[k] = z[i];
        keep[k] = i0 + k < n && !plane_within(c, px[k], py[k], pz[k], threshold);
        cnt += keep[k] ? 1u : 0u;
    }
    uint32_t tot;
    uint32_t rank = block_excl_scan(cnt, s_w, &tot);
    if (threadIdx.x == 0) st_agent(tile_pub + blockIdx.x, tot + 1u);
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t v = ld_agent(tile_pub + t), spins = 0;
        while (v == 0u) {
            if (++spins > kSpinLimit) { atomicOr(fail, 1u); v = 1u; break; }
            __builtin_amdgcn_s_sleep(2);
            v = ld_agent(tile_pub + t);
        }
        mine += v - 1u;
    }
    __syncthreads();
    const uint32_t before = block_sum(mine, s_w);
    rank += before;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (keep[k]) { ox[rank] = px[k]; oy[rank] = py[k]; oz[rank] = pz[k]; ++rank; }
    if (threadIdx.x == 0 && base + kScanTile >= n) *n_out = before + tot;
}

// The cloud the cluster stage will see, set from outside a frame's own kernels (after the plane removal of the node chain).
// The cloud the cell stage would see right now (the voxel centroids, or the planes a filter step left in DevState) as 16-byte
// {x, y, z, 1.0f} records (pcl::PointXYZ, the layout of pcl::toROSMsg) into pinned host memory: the topics between the steps of
// the node's chain (od.cpp:288-295, 331-338).
__global__ __launch_bounds__(kBlock) void k_cloud16(const Frame* frames, float4* __restrict__ out) {
    const Frame& F = frames[0];
    const DevState* st = &F.st;
    if (st->err) return;
    const bool sel = st->c_from_input != 0;
    const float* __restrict__ x = sel ? st->src_x.get() : F.cen_x.get();
    const float* __restrict__ y = sel ? st->src_y.get() : F.cen_y.get();
    const float* __restrict__ z = sel ? st->src_z.get() : F.cen_z.get();
    const uint32_t n = st->c_n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = make_float4(x[i], y[i], z[i], 1.0f);
}

__global__ void k_set_cloud(Frame* frames, const float* x, const float* y, const float* z, uint32_t n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevState* st = &frames[0].st;
    st->c_n = n; st->c_from_input = 1u;
    st->src_x = x; st->src_y = y; st->src_z = z;
    uint32_t cap = 1024;
    while (cap < 2u * n && cap < 0x80000000u) cap <<= 1;
    st->hash_mask = cap - 1u;
}

}  // namespace pcop

// pcop_occupancy.hpp - bounds filter + occupancy binning (the step before the path).
// Part of pcop_device.hpp (included from there, in this order; not a stand-alone header).
#pragma once

namespace pcop {

// ------------------------------------------------------------------------------------------
// Bounds filter + occupancy binning: build_initial_occupancy_grid_dataset (od.cpp:175-269).
//   k_occ_filter   predicate, grid cell, counts (LDS histogram per block, 64-bit global adds), stable
//                  compaction of the kept points (tiles publish their counts / add up the lower tiles)
//   k_occ_finish   integer row means, hole marks
// 12 N bytes in, 12 N_kept out; the counts are integers, so the order of the adds does not matter.
// ------------------------------------------------------------------------------------------
struct OccParams {
    float x_min, x_max, y_min, y_max, z_min, z_max, block, dev;
    int32_t w, h;
};
constexpr int kOccLdsCells = 4096;   // grids up to this many cells are counted in LDS first ...
constexpr int kOccPartials = 64;     // ... and flushed to one of this many partial grids (977 blocks adding into the same
                                     // few hundred 64-bit words would serialise in L2); k_occ_finish adds the partials up

// Smallest c >= 0 for which `lo + (c+1)*block < a` is false: the walk of od.cpp:139-142, whose float
// expression is monotone in c, so an estimate plus a local walk in both directions lands on the same c.
__device__ __forceinline__ int occ_walk_up(float lo, float block, float a) {
    if (!(lo + block < a)) return 0;   // also a NaN argument
    const float t = (a - lo) / block;
    int c = t < 2.0e9f ? (int)t : 2000000000;
    c = c > 2 ? c - 2 : 0;
    while (c > 0 && !(lo + (float)c * block < a)) --c;          // the test at c-1 must still hold
    while (lo + (float)(c + 1) * block < a) ++c;
    return c;
}
// Smallest c >= 0 for which `hi - (c+1)*block > b` is false (od.cpp:144-147).
__device__ __forceinline__ int occ_walk_down(float hi, float block, float b) {
    if (!(hi - block > b)) return 0;
    const float t = (hi - b) / block;
    int c = t < 2.0e9f ? (int)t : 2000000000;
    c = c > 2 ? c - 2 : 0;
    while (c > 0 && !(hi - (float)c * block > b)) --c;
    while (hi - (float)(c + 1) * block > b) ++c;
    return c;
}

__global__ __launch_bounds__(kBlock) void k_occ_filter(OccParams p, const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ z, uint32_t n, float* __restrict__ ox,
                                                       float* __restrict__ oy, float* __restrict__ oz,
                                                       unsigned long long* __restrict__ counts, uint32_t* __restrict__ partials,
                                                       uint32_t* __restrict__ tile_pub,
                                                       uint32_t* __restrict__ totals) {   // totals: {kept, index errors, internal error}
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    __shared__ uint32_t s_cnt[kOccLdsCells];
    const int size = p.w * p.h;
    const bool lds = size <= kOccLdsCells;
    if (lds) for (int i = threadIdx.x; i < size; i += kBlock) s_cnt[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kScanTile;
    float px[kScanItems], py[kScanItems], pz[kScanItems];
    bool keep[kScanItems];
    uint32_t c = 0, bad = 0;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {   // loads first (clamped, not tested)
        const uint32_t i = min(i0 + k, n - 1u);
        px[k] = x[i]; py[k] = y[i]; pz[k] = z[i];
    }
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        keep[k] = false;
        if (i < n) {
            // od.cpp:196-199: only x is tested for NaN; a NaN y or z fails every comparison and passes
            keep[k] = !(isnan(px[k]) || px[k] < p.x_min || px[k] > p.x_max || pz[k] < p.z_min || pz[k] > p.z_max ||
                        py[k] < p.y_min || py[k] > p.y_max);
            if (keep[k]) {
                // od.cpp:203-204: the column walks up from y_min along y, the row walks down from x_max along x
                const int col = occ_walk_up(p.y_min, p.block, py[k]);
                const int row = occ_walk_down(p.x_max, p.block, px[k]);
                const int index = row * p.w + col;
                if (index >= size) ++bad;
                else if (lds) atomicAdd(&s_cnt[index], 1u);
                else atomicAdd(counts + index, 1ull);
                ++c;
            }
        }
    }
    uint32_t tot;
    uint32_t rank = block_excl_scan(c, s_w, &tot);
    if (threadIdx.x == 0) st_agent(tile_pub + blockIdx.x, tot + 1u);   // publish
    if (lds) {
        __syncthreads();
        uint32_t* mine_part = partials + (size_t)(blockIdx.x % kOccPartials) * size;
        for (int i = threadIdx.x; i < size; i += kBlock) { const uint32_t v = s_cnt[i]; if (v) atomicAdd(mine_part + i, v); }
    }
    const uint32_t nbad = block_sum(bad, s_w);
    if (threadIdx.x == 0 && nbad) atomicAdd(totals + 1, nbad);
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t v = ld_agent(tile_pub + t), spins = 0;
        while (v == 0u) {
            if (++spins > kSpinLimit) { atomicOr(totals + 2, 1u); v = 1u; break; }
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
    if (threadIdx.x == 0 && base + kScanTile >= n) totals[0] = before + tot;   // the last tile
}

// one block per grid row: integer mean of the row (od.cpp:226-234), then the hole marks (od.cpp:241-266)
__global__ __launch_bounds__(kBlock) void k_occ_finish(OccParams p, unsigned long long* __restrict__ counts,
                                                       const uint32_t* __restrict__ partials,
                                                       long long* __restrict__ row_avg, signed char* __restrict__ grid) {
    __shared__ unsigned long long s_sum[kBlock];
    __shared__ unsigned int s_cell[kOccLdsCells];
    const int r = blockIdx.x;
    const int size = p.w * p.h;
    if (size <= kOccLdsCells) {   // add up the partial grids of k_occ_filter: all threads, (partial, cell) pairs
        for (int cidx = threadIdx.x; cidx < p.w; cidx += kBlock) s_cell[cidx] = 0u;
        __syncthreads();
        for (int e = threadIdx.x; e < p.w * kOccPartials; e += kBlock) {
            const int q = e / p.w, cidx = e - q * p.w;
            const uint32_t v = partials[(size_t)q * size + (size_t)r * p.w + cidx];
            if (v) atomicAdd(&s_cell[cidx], v);
        }
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < p.w; cidx += kBlock) counts[(size_t)r * p.w + cidx] += s_cell[cidx];
        __syncthreads();
    }
    unsigned long long s = 0;
    for (int cidx = threadIdx.x; cidx < p.w; cidx += kBlock) s += counts[(size_t)r * p.w + cidx];
    s_sum[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) s_sum[threadIdx.x] += s_sum[threadIdx.x + d];
        __syncthreads();
    }
    const long long avg = (long long)s_sum[0] / (long long)p.w;
    if (threadIdx.x == 0) row_avg[r] = avg;
    const float keep = 1 - p.dev;   // int 1 - float, as written at od.cpp:258
    const float limit = (float)avg * keep;
    for (int cidx = threadIdx.x; cidx < p.w; cidx += kBlock) {
        const long long cnt = (long long)counts[(size_t)r * p.w + cidx];
        grid[(size_t)r * p.w + cidx] = ((float)cnt < limit) ? 100 : 0;
    }
}

}  // namespace pcop

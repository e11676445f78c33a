// pcop_neighbours.hpp - mean k-NN distance per point (statistical outlier removal).
// Part of pcop_device.hpp (included from there, in this order; not a stand-alone header).
#pragma once

namespace pcop {

// ------------------------------------------------------------------------------------------
// Mean distance to the k nearest neighbours: the per-point part of pcl::StatisticalOutlierRemoval
// (remove_statistical_outliers, od.cpp:316-340).  Runs on the dense cell grid of a cluster-only
// frame whose cell edge the host picked from the density (k_cgrid_setup .. k_cell_place): one lane
// per point walks the shells of cells around its own cell until the k-th best squared distance
// (flann::L2_Simple, float32) is covered, keeps the k smallest in registers, drops the smallest
// (the query itself in the reference's k+1 search) and adds up the square roots of the others in
// ascending order, in double, as the reference does.  mean[idx] = (float)(sum / mean_k).
// The global mean / sigma of these values is a sequential double sum in the reference; the host forms it.
// ------------------------------------------------------------------------------------------
constexpr int kKnnMax = 32;   // mean_k + 1 <= 32

__global__ __launch_bounds__(kBlock) void k_knn_mean(const Frame* frames, uint32_t mean_k, double edge) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    const uint32_t* __restrict__ pt_slot = F.pt_slot;
    float* __restrict__ mean_out = F.cen_x;   // free in a cluster-only frame
    if (cgrid_off(st)) return;
    const uint32_t n = st->c_n;
    const uint32_t K = mean_k + 1u;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    const int d0 = (int)st->c_dim[0], d1 = (int)st->c_dim[1], d2 = (int)st->c_dim[2];
    const int rmax = max(d0, max(d1, d2));
    const float inf = __uint_as_float(0x7F800000u);
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float4 q = rec[j];   // records are sorted by cell: neighbouring lanes search neighbouring cells
        const uint32_t idx = f2u(q.w);
        const uint32_t pos = slot_meta[pt_slot[idx]].w;
        const int gz = (int)(pos / plane), gy = (int)((pos % plane) / rowbits), gx = (int)(pos % rowbits);
        float best[kKnnMax];
#pragma unroll
        for (int t = 0; t < kKnnMax; ++t) best[t] = inf;
        float worst = inf;          // largest of the K kept values (inf until K are kept)
        uint32_t have = 0;
        for (int r = 0; r <= rmax; ++r) {
            const int z0 = max(gz - r, 0), z1 = min(gz + r, d2 - 1);
            const int y0 = max(gy - r, 0), y1 = min(gy + r, d1 - 1);
            for (int cz = z0; cz <= z1; ++cz) {
                for (int cy = y0; cy <= y1; ++cy) {
                    const bool face = abs(cz - gz) == r || abs(cy - gy) == r;   // whole row on the shell, else its two ends
                    const int step = face ? 1 : max(2 * r, 1);
                    for (int cx = gx - r; cx <= gx + r; cx += step) {
                        if (cx < 0 || cx >= d0) continue;
                        const uint32_t at = (uint32_t)cz * plane + (uint32_t)cy * rowbits + (uint32_t)cx;
                        const uint2 e = cmap[at >> 5];
                        if (!((e.x >> (at & 31u)) & 1u)) continue;
                        const uint4 mb = slot_meta[e.y + __popc(e.x & ((1u << (at & 31u)) - 1u))];
                        for (uint32_t t = 0; t < mb.y; ++t) {
                            const float d = l2_simple(q, rec[mb.x + t]);
                            if (!(d < worst)) continue;
                            // replace the largest kept value (an empty place holds inf), then find the new largest
                            bool done = false;
                            float nw = -1.0f;
#pragma unroll
                            for (int u = 0; u < kKnnMax; ++u) {
                                if ((uint32_t)u < K) {
                                    if (!done && best[u] == worst) { best[u] = d; done = true; }
                                    nw = fmaxf(nw, best[u]);
                                }
                            }
                            worst = nw;
                            have = min(have + 1u, K);
                        }
                    }
                }
            }
            // every point not yet seen sits in a cell more than r cells away along some axis, i.e. farther than
            // r * edge (less a hair for the rounding of the cell assignment)
            if (have == K) {
                const double reach = fmax((double)r * edge - 1.0e-6 * edge, 0.0);
                if ((double)worst <= reach * reach) break;
            }
        }
        // ascending order: pick the smallest remaining value K times; the first (the query point) is dropped
        double dist_sum = 0.0;
        for (uint32_t t = 0; t < K; ++t) {
            float m = inf;
#pragma unroll
            for (int u = 0; u < kKnnMax; ++u) if ((uint32_t)u < K) m = fminf(m, best[u]);
            bool gone = false;
#pragma unroll
            for (int u = 0; u < kKnnMax; ++u) if ((uint32_t)u < K && !gone && best[u] == m) { best[u] = inf; gone = true; }
            if (t) dist_sum += sqrt((double)m);
        }
        mean_out[idx] = (float)(dist_sum / (double)mean_k);
    }
}

}  // namespace pcop

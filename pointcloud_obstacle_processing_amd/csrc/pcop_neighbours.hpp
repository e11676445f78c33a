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

// edge_arg > 0: the cell edge the host chose; otherwise the one k_sor_setup put into DevState (1 / inv_cell).
// out_sel = 0: the values go to cen_x (free in a cluster-only frame); 1: inside the node chain, where cen_x holds the
// voxel cloud: to pt_root (free between k_cell_count and k_cell_flatten) and, for the host's sums, to pinned memory.
__global__ __launch_bounds__(kBlock) void k_knn_mean(const Frame* frames, uint32_t mean_k, double edge_arg, uint32_t out_sel) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    const uint32_t* __restrict__ pt_slot = F.pt_slot;
    float* __restrict__ mean_out = out_sel ? reinterpret_cast<float*>(F.pt_root.get()) : F.cen_x.get();
    float* __restrict__ mean_host = out_sel ? F.h_aux.get() : nullptr;
    if (cgrid_off(st)) return;
    const double edge = edge_arg > 0.0 ? edge_arg : 1.0 / st->inv_cell;
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
        const float mean = (float)(dist_sum / (double)mean_k);
        mean_out[idx] = mean;
        if (mean_host) mean_host[idx] = mean;
    }
}

// ------------------------------------------------------------------------------------------
// Statistical outlier removal inside the node chain (od.cpp:751-754 between od.cpp:740 and od.cpp:791-797): the cloud
// is the voxel cloud in HBM.  k_minmax(from_cen) -> k_sor_setup -> the cell grid kernels -> k_knn_mean; the host forms the
// two sequential double sums of the n mean distances (a parallel sum rounds differently, a single lane is slower than the
// host); k_sor_select applies the threshold and compacts the kept points, in order, into planes carved from rec_b,
// which become the cloud the cluster stage sees.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sor_setup(Frame* frames, uint32_t nblocks, uint32_t mean_k) {
    __shared__ MinMaxPartial s_p[kBlock];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (st->err) return;
    const MinMaxPartial* __restrict__ partials = F.partials;
    const float fbig = 3.402823466e+38f;
    MinMaxPartial q;
    for (int a = 0; a < 3; ++a) { q.mn[a] = fbig; q.mx[a] = -fbig; }
    q.valid = 0; q.nonfinite = 0;
    for (uint32_t i = threadIdx.x; i < nblocks; i += kBlock) {
        const MinMaxPartial p = partials[i];
        for (int a = 0; a < 3; ++a) { q.mn[a] = fminf(q.mn[a], p.mn[a]); q.mx[a] = fmaxf(q.mx[a], p.mx[a]); }
        q.valid += p.valid; q.nonfinite += p.nonfinite;
    }
    s_p[threadIdx.x] = q;
    reduce_partials_lds(s_p);
    uint32_t* __restrict__ tile_pub = F.tile_a;   // k_sor_select publishes here
    const uint32_t n = st->c_n;
    for (uint32_t t = threadIdx.x; t < (n + kScanTile - 1) / kScanTile; t += kBlock) tile_pub[t] = 0u;
    if (threadIdx.x == 0) {
        const MinMaxPartial p = s_p[0];
        if (p.nonfinite) atomicOr(&st->err, (uint32_t)kErrNonFinite);
        else if (n != 0u && n < mean_k + 1u) atomicOr(&st->err, (uint32_t)kErrSorFew);
        else if (n != 0u) {
            for (int a = 0; a < 3; ++a) { st->vmin[a] = p.mn[a]; st->vmax[a] = p.mx[a]; }
            // about four points per cell if the cloud filled its box; any edge gives the same neighbours
            double ext[3], longest = 0.0;
            for (int a = 0; a < 3; ++a) { ext[a] = fmax((double)p.mx[a] - (double)p.mn[a], 1e-6); longest = fmax(longest, ext[a]); }
            double edge = cbrt(ext[0] * ext[1] * ext[2] * 4.0 / (double)n);
            edge = fmax(edge, longest / 2000.0);
            for (int it = 0; it < 64; ++it) {
                double cells = 1.0;
                for (int a = 0; a < 3; ++a) cells *= floor(ext[a] / edge) + 1.0 + 2.0 * kCellMargin + (a == 0 ? 32.0 : 0.0);
                if (cells <= 0.5 * (double)kCmapCells) break;
                edge *= 1.26;
            }
            st->inv_cell = 1.0 / edge;
        }
        st->cgrid_req = 1u;   // the k-NN pass runs on the dense cell grid whatever the cluster stage uses
    }
    __syncthreads();
    cgrid_setup_block(F, n, n == 0u);
}

__global__ __launch_bounds__(kBlock) void k_sor_select(Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    __shared__ double s_thr;
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (st->err) return;
    const uint32_t n = st->c_n;
    float* __restrict__ ax = reinterpret_cast<float*>(F.rec_b.get());   // the kept cloud: three planes of `cap` floats (rec_b: 16 bytes per point)
    float* __restrict__ ay = ax + F.cap; float* __restrict__ az = ay + F.cap;
    if (n == 0u) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->inv_cell = (*F.args).inv_cell; st->cgrid_req = (*F.args).cell_grid; st->cpath = 0u;
            st->c_from_input = 1u; st->src_x = (const float*)ax; st->src_y = (const float*)ay; st->src_z = (const float*)az;
        }
        return;
    }
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    if (threadIdx.x == 0) s_thr = (*F.args).sor_thr;
    __syncthreads();
    const double thr = s_thr;
    const bool from_in = st->c_from_input != 0;
    const float* __restrict__ x = from_in ? st->src_x.get() : F.cen_x.get();
    const float* __restrict__ y = from_in ? st->src_y.get() : F.cen_y.get();
    const float* __restrict__ z = from_in ? st->src_z.get() : F.cen_z.get();
    const float* __restrict__ dist = reinterpret_cast<const float*>(F.pt_root.get());
    uint32_t* __restrict__ tile_pub = F.tile_a;
    float px[kScanItems], py[kScanItems], pz[kScanItems];
    bool keep[kScanItems];
    uint32_t cnt = 0;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = min(i0 + k, n - 1u);
        px[k] = x[i]; py[k] = y[i]; pz[k] = z[i];
        keep[k] = i0 + k < n && !((double)dist[i] > thr);   // the outliers are the points above the threshold
        cnt += keep[k] ? 1u : 0u;
    }
    uint32_t tot;
    uint32_t rank = block_excl_scan(cnt, s_w, &tot);
    if (threadIdx.x == 0) st_agent(tile_pub + blockIdx.x, tot + 1u);
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t v = ld_agent(tile_pub + t), spins = 0;
        while (v == 0u) {
            if (++spins > kSpinLimit) { atomicOr(&st->err, (uint32_t)kErrInternal); v = 1u; break; }
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
        if (keep[k]) { ax[rank] = px[k]; ay[rank] = py[k]; az[rank] = pz[k]; ++rank; }
    if (threadIdx.x == 0 && base + kScanTile >= n) {   // the last tile: the kept cloud is what the cluster stage sees
        const uint32_t kept = before + tot;
        st->c_n = kept; st->c_from_input = 1u;
        st->src_x = (const float*)ax; st->src_y = (const float*)ay; st->src_z = (const float*)az;
        st->inv_cell = (*F.args).inv_cell; st->cgrid_req = (*F.args).cell_grid; st->cpath = 0u;   // back to the cluster stage's cells
        uint32_t cap = 1024;
        while (cap < 2u * kept && cap < 0x80000000u) cap <<= 1;
        st->hash_mask = cap - 1u;
    }
}

// The per-cell counters the dense grid borrows from the hash table go back clean (k_emit does this at the end of a frame;
// here another cell grid is about to be built in the same frame).
__global__ __launch_bounds__(kBlock) void k_cells_reset(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    uint32_t* __restrict__ hcount = F.hcount; uint32_t* __restrict__ hmin = F.hmin;
    if (!st->cpath) return;
    const uint32_t nc = st->err ? st->c_n : st->n_cells;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) { hcount[i] = 0u; hmin[i] = 0xFFFFFFFFu; }
}

// DevState -> pinned host memory in the middle of a call (k_emit does it at the end of a frame).
__global__ void k_state_out(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    if (threadIdx.x == 0) *F.h_state = F.st;
}

}  // namespace pcop

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
constexpr uint32_t kKnnBatch = 8;   // candidate records requested together
constexpr int kKnnFineMax = 15;     // k_knn_far: shells below this radius are walked cell row by cell row, the space beyond by blocks

// best[] ascending; keeps the KCAP smallest of what it has seen (a compare-exchange chain, no indexing).  A search for
// K < KCAP values starts with KCAP - K places at -inf, which nothing displaces: the K smallest then sit in the last K
// places and the K-th is always best[KCAP - 1] - a place known at compile time (an index computed at run time would
// send the whole array to memory).
template <int KCAP>
__device__ __forceinline__ void knn_push(float (&best)[KCAP], float d) {
#pragma unroll
    for (int u = 0; u < KCAP; ++u) { const float lo = fminf(best[u], d); d = fmaxf(best[u], d); best[u] = lo; }
}
// The records rec[a0 .. a0 + len) against the query: loads in groups of kKnnBatch with clamped indices, then the tests.
template <int KCAP>
__device__ __forceinline__ void knn_scan_range(float (&best)[KCAP], const float4& q, const float4* __restrict__ rec, uint32_t a0, uint32_t len) {
    for (uint32_t t = 0; t < len; t += kKnnBatch) {
        float4 c[kKnnBatch];
#pragma unroll
        for (uint32_t u = 0; u < kKnnBatch; ++u) c[u] = rec[a0 + min(t + u, len - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < kKnnBatch; ++u) {
            const float d = l2_simple(q, c[u]);
            if (t + u < len && d < best[KCAP - 1]) knn_push(best, d);
        }
    }
}

// One bit per block of 4x4x4 cells of the dense grid: is any of them occupied?  (x blocks never straddle an occupancy
// word: rows are padded to 32 cells.)  Block index = (bz * ceil(d1 / 4) + by) * (rowbits / 4) + bx.
__global__ __launch_bounds__(kBlock) void k_cmap_coarse(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint2* __restrict__ cmap = F.cmap;
    uint32_t* __restrict__ cc = F.ccoarse;
    if (cgrid_off(st) || st->n_cells == 0u) return;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane, d1 = st->c_dim[1], d2 = st->c_dim[2];
    const uint32_t cdx = rowbits >> 2, cdy = (d1 + 3u) >> 2, cdz = (d2 + 3u) >> 2;
    const uint32_t total = min(cdx * cdy * cdz, kCoarseBits);   // (the largest grid has about half as many)
    const uint32_t lane = lane_id(), nthreads = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x - lane; base < total; base += nthreads) {
        const uint32_t t = base + lane;
        bool occ = false;
        if (t < total) {
            const uint32_t bx = t % cdx, by = (t / cdx) % cdy, bz = t / (cdx * cdy);
            for (uint32_t dz = 0; dz < 4u; ++dz)
                for (uint32_t dy = 0; dy < 4u; ++dy) {
                    const uint32_t z = min(bz * 4u + dz, d2 - 1u), y = min(by * 4u + dy, d1 - 1u);   // (clamped: a cell read twice changes nothing)
                    const uint32_t at = z * plane + y * rowbits + bx * 4u;
                    occ |= ((cmap[at >> 5].x >> (at & 31u)) & 0xFu) != 0u;
                }
        }
        const uint64_t m = __ballot(occ);
        if (lane == 0) cc[base >> 5] = (uint32_t)m;
        if (lane == 32 && (base >> 5) + 1u < kCoarseBits / 32u) cc[(base >> 5) + 1u] = (uint32_t)(m >> 32);
    }
}

// edge_arg > 0: the cell edge the host chose; otherwise the one k_sor_setup put into DevState (1 / inv_cell).
// out_sel = 0: the values go to cen_x (free in a cluster-only frame); 1: inside the node chain, where cen_x holds the
// voxel cloud: to pt_root (free between k_cell_count and k_cell_flatten) and, for the host's sums, to pinned memory.
// KCAP = 16 or 32 >= mean_k + 1: the values kept in registers.
//
// Shells 0 and 1 (the 27 cells around the query's) are nine rows of three cells.  The occupied cells of a row window have
// consecutive ids and the records are sorted by cell, so a window is ONE contiguous range of records: two rounds of
// independent loads (9 windows, then the metadata of each window's first and last cell) give nine ranges, scanned with
// batched loads.  The grid's edge is sized so that these shells settle nearly every point.
template <int KCAP>
__global__ __launch_bounds__(kBlock) void k_knn_mean(const Frame* frames, uint32_t mean_k, double edge_arg, uint32_t out_sel) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    const uint32_t* __restrict__ pt_slot = F.pt_slot;
    float* __restrict__ mean_out = out_sel ? reinterpret_cast<float*>(F.pt_root.get()) : F.cen_x.get();
    float* __restrict__ mean_host = out_sel ? F.h_aux.get() : nullptr;
    uint32_t* __restrict__ far_list = F.idx_a;        // (idle until the clusters' CSR)
    uint32_t* __restrict__ far_count = F.bin_total;   // zeroed by k_sor_setup
    if (cgrid_off(st)) return;
    const double edge = edge_arg > 0.0 ? edge_arg : 1.0 / st->inv_cell;
    const uint32_t n = st->c_n;
    const uint32_t K = mean_k + 1u;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    const float inf = __uint_as_float(0x7F800000u);
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float4 q = rec[j];   // records are sorted by cell: neighbouring lanes search neighbouring cells
        const uint32_t idx = f2u(q.w);
        const uint32_t own = pt_slot[idx];
        const uint32_t pos = slot_meta[own].w;
        float best[KCAP];
#pragma unroll
        for (int t = 0; t < KCAP; ++t) best[t] = (uint32_t)t < (uint32_t)KCAP - K ? -inf : inf;
        // round 1: the nine row windows (every point has two free cells on each side, so they lie inside the array)
        uint2 e0[9];
        uint32_t e1[9], sh[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const uint32_t rp = pos + (uint32_t)(r / 3 - 1) * plane + (uint32_t)(r % 3 - 1) * rowbits - 1u;
            sh[r] = rp & 31u;
            e0[r] = cmap[rp >> 5];
            e1[r] = cmap[(rp >> 5) + 1u].x;
        }
        // round 2: metadata of the first and the last occupied cell of each window (an empty window re-reads the own cell's)
        uint4 mf[9], ml[9];
        uint32_t cnt[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            uint32_t bits = e0[r].x >> sh[r];
            if (sh[r] + 3u > 32u) bits |= (e1[r] << 1) << (31u - sh[r]);
            cnt[r] = __popc(bits & 7u);
            const uint32_t id0 = e0[r].y + __popc(e0[r].x & ((1u << sh[r]) - 1u));
            mf[r] = slot_meta[cnt[r] ? id0 : own];
            ml[r] = slot_meta[cnt[r] ? id0 + cnt[r] - 1u : own];
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) knn_scan_range<KCAP>(best, q, rec, mf[r].x, cnt[r] ? ml[r].x + ml[r].y - mf[r].x : 0u);
        // every point not yet seen sits in a cell more than r cells away along some axis, i.e. farther than
        // r * edge (less a hair for the rounding of the cell assignment)
        bool settled;
        {
            const double reach = fmax(edge - 1.0e-6 * edge, 0.0);
            settled = (double)best[KCAP - 1] <= reach * reach;   // inf (fewer than K seen) never is
        }
        if (!settled) {   // for k_knn_far: one wave per such point
            const uint64_t m = __ballot(true);
            const uint32_t first = (uint32_t)__ffsll((long long)m) - 1u;
            uint32_t base = 0;
            if (lane_id() == first) base = atomicAdd(far_count, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
            far_list[base + (uint32_t)__popcll(m & lanemask_lt())] = j;
            continue;
        }
        // ascending order; the first (the query point itself in the reference's k + 1 search) is dropped
        double dist_sum = 0.0;
#pragma unroll
        for (int t = 1; t < KCAP; ++t) if ((uint32_t)t > (uint32_t)KCAP - K) dist_sum += sqrt((double)best[t]);
        const float mean = (float)(dist_sum / (double)mean_k);
        mean_out[idx] = mean;
        if (mean_host) mean_host[idx] = mean;
    }
}

// ------------------------------------------------------------------------------------------
// The points the 27 cells do not settle - outliers, mostly: the points this filter is there for - one WAVE each
// (k_knn_mean listed them).  A lane that walks outwards alone pays a memory round trip per probe, and its 63
// neighbours wait for it; here the lanes share the walk.  Shell r of cells around the point's cell is (2r+1)^2 rows
// along x - whole rows on four of its faces, the two end cells elsewhere - and a row is one or more occupancy words:
// the lanes take a (row, word) item each, the occupied cells found are put on a list in LDS, and the lanes take a cell
// each from the list and scan its records into their own sorted list of the smallest distances (knn_push, as in
// k_knn_mean).  After shell r every point not seen is farther than r edges, so the point is settled once mean_k + 1 of
// the values the lanes hold are within that reach: one compare-and-count per shell.  From kKnnFineMax on the walk goes by
// blocks of 4x4x4 cells with one bit each (k_cmap_coarse): a point alone in a large empty space probes a 64th.
// At the end the K-th smallest value of the union of the lanes' lists is found by bisection on its bit pattern
// (non-negative floats order like unsigned integers), the smaller ones are sorted by counting, and lane 0 adds the
// square roots in ascending order in double, dropping the first - the reference's arithmetic.
// ------------------------------------------------------------------------------------------
constexpr int kFarRounds = 1;          // occupancy words a lane requests together (more only add index arithmetic: the shells are small)
constexpr uint32_t kFarCells = 2048;   // cells a wave can list per round: 64 lanes x the 32 bits of an occupancy word

template <int KCAP>
__device__ __forceinline__ uint32_t knn_count_le(const float (&best)[KCAP], uint32_t bound_bits) {
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < KCAP; ++u) c += f2u(best[u]) <= bound_bits ? 1u : 0u;
    return wave_sum(c);
}

// the lanes' cells of one round -> the wave's list, behind the `listed` entries it holds (the caller has made room)
__device__ __forceinline__ void far_list_cells(uint32_t* s_cell, uint32_t listed, uint32_t incl, uint32_t word, uint32_t before, uint32_t bits) {
    uint32_t at = listed + incl - (uint32_t)__popc(bits);
    while (bits) {
        const uint32_t k = (uint32_t)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        s_cell[at++] = before + (uint32_t)__popc(word & ((1u << k) - 1u));
    }
}

template <int KCAP>
__device__ __forceinline__ void far_scan_cells(float (&best)[KCAP], const float4& q, const float4* __restrict__ rec,
                                               const uint4* __restrict__ slot_meta, const uint32_t* s_cell, uint32_t total) {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t ci = lane_id(); ci < total; ci += kWave) {
        const uint4 mc = slot_meta[s_cell[ci]];
        knn_scan_range<KCAP>(best, q, rec, mc.x, mc.y);
    }
    __builtin_amdgcn_wave_barrier();
}

template <int KCAP>
__global__ __launch_bounds__(kBlock) void k_knn_far(const Frame* frames, uint32_t mean_k, double edge_arg, uint32_t out_sel) {
    __shared__ uint32_t s_cells[kWavesPerBlock][kFarCells];
    __shared__ float s_vals[kWavesPerBlock][kKnnMax];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint32_t* __restrict__ ccoarse = F.ccoarse;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    const uint32_t* __restrict__ pt_slot = F.pt_slot;
    float* __restrict__ mean_out = out_sel ? reinterpret_cast<float*>(F.pt_root.get()) : F.cen_x.get();
    float* __restrict__ mean_host = out_sel ? F.h_aux.get() : nullptr;
    const uint32_t* __restrict__ far_list = F.idx_a;
    if (cgrid_off(st)) return;
    const uint32_t n_far = min(F.bin_total[0], st->c_n);
    const double edge = edge_arg > 0.0 ? edge_arg : 1.0 / st->inv_cell;
    const double hair = 1.0e-6 * edge;
    const uint32_t K = mean_k + 1u;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    const int d1 = (int)st->c_dim[1], d2 = (int)st->c_dim[2];
    const int cdx = (int)(rowbits >> 2), cdy = (d1 + 3) >> 2, cdz = (d2 + 3) >> 2;
    const int rlast = max((int)rowbits, max(d1, d2));
    const float inf = __uint_as_float(0x7F800000u);
    const uint32_t lane = lane_id(), wv = threadIdx.x / kWave;
    uint32_t* s_cell = s_cells[wv];
    for (uint32_t w = blockIdx.x * kWavesPerBlock + wv; w < n_far; w += gridDim.x * kWavesPerBlock) {
        const float4 q = rec[far_list[w]];
        const uint32_t idx = f2u(q.w);
        const uint32_t pos = slot_meta[pt_slot[idx]].w;
        const int gz = (int)(pos / plane), gy = (int)((pos % plane) / rowbits), gx = (int)(pos % rowbits);
        float best[KCAP];
#pragma unroll
        for (int t = 0; t < KCAP; ++t) best[t] = inf;
        bool settled = false;
        int r = 0;
        // cell row by cell row
        for (; r < kKnnFineMax && r <= rlast && !settled; ++r) {
            const int side = 2 * r + 1, nrows = side * side;
            const int xlo = max(gx - r, 0), xhi = min(gx + r, (int)rowbits - 1);
            const int wa = xlo >> 5, nw = (xhi >> 5) - wa + 1;
            const int nitems = nrows * nw;
            uint32_t listed = 0;   // the cells of a shell are listed round by round and scanned together (when they fit)
            for (int it0 = 0; it0 < nitems; it0 += kWave * kFarRounds) {   // kFarRounds words per lane requested together
                uint2 e[kFarRounds];
                uint32_t mask[kFarRounds];
#pragma unroll
                for (int u = 0; u < kFarRounds; ++u) {
                    const int it = it0 + u * kWave + (int)lane;
                    const int rho = it / nw, wi = wa + it % nw;
                    const int cz = gz - r + rho / side, cy = gy - r + rho % side;
                    const bool in = it < nitems && cz >= 0 && cz < d2 && cy >= 0 && cy < d1;
                    const bool face = abs(cz - gz) == r || abs(cy - gy) == r;
                    // the window's bits inside word wi: a whole row, or its two ends
                    const int lo = max(xlo, wi * 32), hi = min(xhi, wi * 32 + 31);
                    uint32_t m = 0u;
                    if (face) m = (0xFFFFFFFFu >> (31 - (hi & 31))) & (0xFFFFFFFFu << (lo & 31));
                    else {
                        if (gx - r >= 0 && ((gx - r) >> 5) == wi) m |= 1u << ((gx - r) & 31);
                        if (gx + r <= (int)rowbits - 1 && ((gx + r) >> 5) == wi) m |= 1u << ((gx + r) & 31);
                    }
                    mask[u] = in ? m : 0u;
                    e[u] = cmap[in ? (((uint32_t)cz * plane + (uint32_t)cy * rowbits) >> 5) + (uint32_t)wi : 0u];
                }
#pragma unroll
                for (int u = 0; u < kFarRounds; ++u) {
                    const uint32_t bits = e[u].x & mask[u];
                    const uint32_t incl = wave_incl_scan((uint32_t)__popc(bits));
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
                    if (listed + total > kFarCells) { far_scan_cells<KCAP>(best, q, rec, slot_meta, s_cell, listed); listed = 0; }
                    far_list_cells(s_cell, listed, incl, e[u].x, e[u].y, bits);
                    listed += total;
                }
            }
            far_scan_cells<KCAP>(best, q, rec, slot_meta, s_cell, listed);
            const double reach = fmax((double)r * edge - hair, 0.0);
            settled = knn_count_le<KCAP>(best, f2u(__double2float_rd(reach * reach))) >= K;   // (float v <= double x  <=>  v <= x rounded down)
        }
        // block by block: after the blocks within R blocks of the point's, every cell within 4R cells has been seen; the
        // cells nearer than kKnnFineMax were taken above.  (Rare, so every lane simply scans what its own blocks hold.)
        for (int R = kKnnFineMax / 4; !settled && R <= max(cdx, max(cdy, cdz)); ++R) {
            const int side = 2 * R + 1, nblk = side * side * side;
            const int bxq = gx >> 2, byq = gy >> 2, bzq = gz >> 2;
            for (int it0 = 0; it0 < nblk; it0 += kWave) {
                const int it = it0 + (int)lane;
                const int ox = it % side - R, oy = (it / side) % side - R, oz = it / (side * side) - R;
                const int bx = bxq + ox, by = byq + oy, bz = bzq + oz;
                if (!(it < nblk && max(abs(ox), max(abs(oy), abs(oz))) == R && bx >= 0 && bx < cdx && by >= 0 && by < cdy && bz >= 0 && bz < cdz)) continue;
                const uint32_t lin = ((uint32_t)bz * (uint32_t)cdy + (uint32_t)by) * (uint32_t)cdx + (uint32_t)bx;
                if (!((ccoarse[lin >> 5] >> (lin & 31u)) & 1u)) continue;
                for (int rr = 0; rr < 16; ++rr) {
                    const int z = bz * 4 + rr / 4, y = by * 4 + rr % 4;
                    if (z >= d2 || y >= d1) continue;
                    const uint32_t at = (uint32_t)z * plane + (uint32_t)y * rowbits + (uint32_t)bx * 4u;   // 4 cells, never across words
                    const uint2 e = cmap[at >> 5];
                    const uint32_t four = (e.x >> (at & 31u)) & 0xFu;
                    if (!four) continue;
                    const int far_yz = max(abs(y - gy), abs(z - gz));
                    uint32_t id = e.y + (uint32_t)__popc(e.x & ((1u << (at & 31u)) - 1u));
                    for (int k = 0; k < 4; ++k) {
                        if (!((four >> k) & 1u)) continue;
                        if (max(abs(bx * 4 + k - gx), far_yz) >= kKnnFineMax) {   // (not seen yet)
                            const uint4 mc = slot_meta[id];
                            knn_scan_range<KCAP>(best, q, rec, mc.x, mc.y);
                        }
                        ++id;
                    }
                }
            }
            const double reach = fmax(4.0 * (double)R * edge - hair, 0.0);
            settled = knn_count_le<KCAP>(best, f2u(__double2float_rd(reach * reach))) >= K;
        }
        // the K-th smallest of all the lanes' values: the smallest bit pattern T with at least K values <= T
        uint32_t lo = 0u, hi = 0x7F800000u;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            if (knn_count_le<KCAP>(best, mid) >= K) hi = mid; else lo = mid + 1u;
        }
        const uint32_t T = lo;
        // the values below T (fewer than K), sorted by counting
        uint32_t below = 0;
#pragma unroll
        for (int u = 0; u < KCAP; ++u) below += f2u(best[u]) < T ? 1u : 0u;
        const uint32_t incl = wave_incl_scan(below);
        const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);   // < K <= kKnnMax
        uint32_t at = incl - below;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < KCAP; ++u) if (f2u(best[u]) < T) s_vals[wv][at++] = best[u];
        __builtin_amdgcn_wave_barrier();
        float mine = inf; uint32_t rank = 0;
        if (lane < m) {
            mine = s_vals[wv][lane];
            for (uint32_t o = 0; o < m; ++o) { const float v = s_vals[wv][o]; rank += (v < mine || (v == mine && o < lane)) ? 1u : 0u; }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < m) s_vals[wv][rank] = mine;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double dist_sum = 0.0;
            for (uint32_t t = 1; t < K; ++t) dist_sum += sqrt((double)(t < m ? s_vals[wv][t] : u2f(T)));
            const float mean = (float)(dist_sum / (double)mean_k);
            mean_out[idx] = mean;
            if (mean_host) mean_host[idx] = mean;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// Statistical outlier removal inside the node chain (od.cpp:751-754 between od.cpp:740 and od.cpp:791-797): the cloud
// is the voxel cloud in HBM.  k_minmax(from_cen) -> k_sor_setup -> the cell grid kernels -> k_knn_mean; the host forms the
// two sequential double sums of the n mean distances (a parallel sum rounds differently, a single lane is slower than the
// host); k_sor_select applies the threshold and compacts the kept points, in order, into planes carved from rec_b,
// which become the cloud the cluster stage sees.
// ------------------------------------------------------------------------------------------
// the edge, enlarged until the grid fits the occupancy map
__device__ __forceinline__ double sor_fit_edge(const double* ext, double edge) {
    for (int it = 0; it < 64; ++it) {
        double cells = 1.0;
        for (int a = 0; a < 3; ++a) cells *= floor(ext[a] / edge) + 1.0 + 2.0 * kCellMargin + (a == 0 ? 32.0 : 0.0);
        if (cells <= 0.5 * (double)kCmapCells) break;
        edge *= 1.26;
    }
    return edge;
}

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
            // about four points per cell if the cloud filled its box; any edge gives the same neighbours.  (Measured and
            // dropped: sizing the edge from the occupancy the average point sees on a trial grid - fewer candidates per
            // point on surfaces, but more shells for the outliers, which are what the kernel's time hangs on.)
            double ext[3], longest = 0.0;
            for (int a = 0; a < 3; ++a) { ext[a] = fmax((double)p.mx[a] - (double)p.mn[a], 1e-6); longest = fmax(longest, ext[a]); }
            const double edge = sor_fit_edge(ext, fmax(cbrt(ext[0] * ext[1] * ext[2] * 4.0 / (double)n), longest / 2000.0));
            st->inv_cell = 1.0 / edge;
        }
        st->cgrid_req = 1u;   // the k-NN pass runs on the dense cell grid whatever the cluster stage uses
        F.bin_total[0] = 0u;  // k_knn_mean counts the points it leaves to k_knn_far here (the radix tables' totals are idle)
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
            if (++spins > st->spin_limit) { atomicOr(&st->err, (uint32_t)kErrInternal); v = 1u; break; }
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

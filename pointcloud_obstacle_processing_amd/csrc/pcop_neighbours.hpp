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
constexpr int kKnnFineMax = 15;     // widest shell walked row by row (31 cells: at most two occupancy words per row)
constexpr int kKnnRows = 8;         // rows of a shell whose occupancy words are requested together
constexpr int kKnnQueue = 32;       // records a lane can hold back for the wave's common scan

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
    __shared__ uint32_t s_bits[2 * kKnnRows][kBlock], s_word[2 * kKnnRows][kBlock], s_before[2 * kKnnRows][kBlock];   // 48 KB: a batch of rows
    __shared__ uint32_t s_queue[kKnnQueue][kBlock];   // 32 KB: per lane, the records of the occupied cells the batch turned up
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
    const int d1 = (int)st->c_dim[1], d2 = (int)st->c_dim[2];
    const float inf = __uint_as_float(0x7F800000u);
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
#ifdef PCOP_STAMPS
        const unsigned long long t_0 = wall_clock64();
#endif
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
#ifdef PCOP_STAMPS
        const unsigned long long t_1 = wall_clock64();
        const uint32_t n_unsettled = (uint32_t)__popcll(__ballot(!settled));
#endif
        if (!settled) {
            // The others (outliers, mostly - the points this filter is there for) look further out, shell by shell.  A
            // shell of radius r is (2r+1)^2 rows of cells along x: whole rows on four of its faces, the two end cells
            // elsewhere; up to kKnnFineMax a row's window lies in at most two occupancy words.  The words of kKnnRows rows
            // are requested together: a probe per row instead of one per cell, and eight in flight instead of one.
            const int gz = (int)(pos / plane), gy = (int)((pos % plane) / rowbits), gx = (int)(pos % rowbits);
            const double hair = 1.0e-6 * edge;
            for (int r = 2; r <= kKnnFineMax && !settled; ++r) {
                const int side = 2 * r + 1, nrows = side * side;
                const int xlo = max(gx - r, 0), xhi = min(gx + r, (int)rowbits - 1);
                const uint32_t wa = (uint32_t)xlo >> 5, wb = (uint32_t)xhi >> 5;
                // window bits inside the two words: a whole row / its two ends
                const uint32_t hi_a = wa == wb ? (uint32_t)xhi & 31u : 31u;
                const uint32_t face_a = (0xFFFFFFFFu >> (31u - hi_a)) & (0xFFFFFFFFu << ((uint32_t)xlo & 31u));
                const uint32_t face_b = wa == wb ? 0u : 0xFFFFFFFFu >> (31u - ((uint32_t)xhi & 31u));
                const uint32_t bl = gx - r >= 0 ? 1u << ((uint32_t)(gx - r) & 31u) : 0u;
                const uint32_t bh = gx + r <= (int)rowbits - 1 ? 1u << ((uint32_t)(gx + r) & 31u) : 0u;
                const uint32_t ends_a = bl | (wa == wb ? bh : 0u), ends_b = wa == wb ? 0u : bh;
                for (int r0 = 0; r0 < nrows; r0 += kKnnRows) {
                    uint2 ea[kKnnRows], eb[kKnnRows];
                    uint32_t ma[kKnnRows], mb[kKnnRows];
#pragma unroll
                    for (int u = 0; u < kKnnRows; ++u) {
                        const int rho = r0 + u;
                        const int cz = gz - r + rho / side, cy = gy - r + rho % side;
                        const bool in = rho < nrows && cz >= 0 && cz < d2 && cy >= 0 && cy < d1;
                        const bool face = abs(cz - gz) == r || abs(cy - gy) == r;
                        const uint32_t rowword = in ? ((uint32_t)cz * plane + (uint32_t)cy * rowbits) >> 5 : 0u;
                        ma[u] = in ? (face ? face_a : ends_a) : 0u;
                        mb[u] = in ? (face ? face_b : ends_b) : 0u;
                        ea[u] = cmap[rowword + wa];
                        eb[u] = cmap[rowword + wb];
                    }
                    // the words with a cell in them are parked in LDS (a loop over them that is not unrolled needs an index) ...
#pragma unroll
                    for (int u = 0; u < kKnnRows; ++u) {
                        s_bits[2 * u][threadIdx.x] = ea[u].x & ma[u]; s_word[2 * u][threadIdx.x] = ea[u].x; s_before[2 * u][threadIdx.x] = ea[u].y;
                        s_bits[2 * u + 1][threadIdx.x] = eb[u].x & mb[u]; s_word[2 * u + 1][threadIdx.x] = eb[u].x; s_before[2 * u + 1][threadIdx.x] = eb[u].y;
                    }
                    // ... the records of the occupied cells found go on the lane's list (out there a cell holds a point or two) ...
                    uint32_t queued = 0;
                    for (int w = 0; w < 2 * kKnnRows; ++w) {
                        uint32_t bits = s_bits[w][threadIdx.x];
                        if (!bits) continue;
                        const uint32_t word = s_word[w][threadIdx.x], before = s_before[w][threadIdx.x];
                        while (bits) {
                            const uint32_t k = (uint32_t)__ffs((int)bits) - 1u;
                            bits &= bits - 1u;
                            const uint4 mc = slot_meta[before + __popc(word & ((1u << k) - 1u))];
                            if (queued + mc.y <= (uint32_t)kKnnQueue) {
                                for (uint32_t i = 0; i < mc.y; ++i) s_queue[queued + i][threadIdx.x] = mc.x + i;
                                queued += mc.y;
                            } else knn_scan_range<KCAP>(best, q, rec, mc.x, mc.y);   // (no room on the list: straight away)
                        }
                    }
                    // ... and the lanes of the wave go through their lists side by side: the lanes find their cells at
                    // different moments, and a scan started by one lane alone would cost the wave as much as one started by all
                    // (a lane reads back only what it wrote itself: no barrier)
                    for (uint32_t t = 0; __any(t < queued); t += kKnnBatch) {
                        if (t < queued) {
                            float4 c[kKnnBatch];
#pragma unroll
                            for (uint32_t u = 0; u < kKnnBatch; ++u) c[u] = rec[s_queue[min(t + u, queued - 1u)][threadIdx.x]];
#pragma unroll
                            for (uint32_t u = 0; u < kKnnBatch; ++u) {
                                const float d = l2_simple(q, c[u]);
                                if (t + u < queued && d < best[KCAP - 1]) knn_push(best, d);
                            }
                        }
                    }
                }
                const double reach = fmax((double)r * edge - hair, 0.0);
                settled = (double)best[KCAP - 1] <= reach * reach;
            }
        }
#ifdef PCOP_STAMPS
        const unsigned long long t_2 = wall_clock64();
#endif
        if (!settled) {
            // Beyond kKnnFineMax cells (a point alone in a large empty space): block by block.  A block is 4x4x4 cells with
            // one bit in F.ccoarse, so empty space costs a 64th of the probes.  After the blocks within R blocks of the
            // query's, every cell within 4R cells of the query's has been seen; the blocks within 3 hold nothing new.
            const uint32_t* __restrict__ ccoarse = F.ccoarse;
            const int gz = (int)(pos / plane), gy = (int)((pos % plane) / rowbits), gx = (int)(pos % rowbits);
            const int bxq = gx >> 2, byq = gy >> 2, bzq = gz >> 2;
            const int cdx = (int)(rowbits >> 2), cdy = (d1 + 3) >> 2, cdz = (d2 + 3) >> 2;
            const int Rmax = max(cdx, max(cdy, cdz));
            const double hair = 1.0e-6 * edge;
            for (int R = (kKnnFineMax + 1) / 4; R <= Rmax; ++R) {
                const int z0 = max(bzq - R, 0), z1 = min(bzq + R, cdz - 1);
                const int y0 = max(byq - R, 0), y1 = min(byq + R, cdy - 1);
                for (int bz = z0; bz <= z1; ++bz) {
                    for (int by = y0; by <= y1; ++by) {
                        const bool face = abs(bz - bzq) == R || abs(by - byq) == R;   // whole row on the shell, else its two ends
                        const int step = face ? 1 : 2 * R;
                        for (int bx = bxq - R; bx <= bxq + R; bx += step) {
                            if (bx < 0 || bx >= cdx) continue;
                            const uint32_t lin = ((uint32_t)bz * (uint32_t)cdy + (uint32_t)by) * (uint32_t)cdx + (uint32_t)bx;
                            if (!((ccoarse[lin >> 5] >> (lin & 31u)) & 1u)) continue;
                            for (int dz = 0; dz < 4; ++dz) {
                                const int z = bz * 4 + dz;
                                if (z >= d2) break;
                                for (int dy = 0; dy < 4; ++dy) {
                                    const int y = by * 4 + dy;
                                    if (y >= d1) break;
                                    const int far_yz = max(abs(y - gy), abs(z - gz));
                                    const uint32_t at = (uint32_t)z * plane + (uint32_t)y * rowbits + (uint32_t)bx * 4u;   // 4 cells, never across words
                                    const uint2 e = cmap[at >> 5];
                                    const uint32_t bits = (e.x >> (at & 31u)) & 0xFu;
                                    if (!bits) continue;
                                    uint32_t id = e.y + __popc(e.x & ((1u << (at & 31u)) - 1u));
                                    for (int k = 0; k < 4; ++k) {
                                        if (!((bits >> k) & 1u)) continue;
                                        const int far = max(abs(bx * 4 + k - gx), far_yz);   // cells between this one and the query's, the larger count over the axes
                                        // far <= kKnnFineMax: seen already.  Otherwise nothing in the cell is nearer than
                                        // (far - 1) edges, which rules it out once the K-th best is no larger.
                                        const double lb = fmax((double)(far - 1) * edge - hair, 0.0);
                                        if (far > kKnnFineMax && !((double)best[KCAP - 1] <= lb * lb)) {
                                            const uint4 mc = slot_meta[id];
                                            knn_scan_range<KCAP>(best, q, rec, mc.x, mc.y);
                                        }
                                        ++id;
                                    }
                                }
                            }
                        }
                    }
                }
                const double reach = fmax(4.0 * (double)R * edge - hair, 0.0);
                if ((double)best[KCAP - 1] <= reach * reach) break;
            }
        }
#ifdef PCOP_STAMPS
        if (lane_id() == 0) {   // diagnostic build: per wave {phase 1, shells, blocks} in 10 ns ticks, lanes that went on
            const unsigned long long t_3 = wall_clock64();
            uint32_t* dbg = F.idx_b.get() + 4u * (j >> 6);
            dbg[0] = (uint32_t)(t_1 - t_0); dbg[1] = (uint32_t)(t_2 - t_1); dbg[2] = (uint32_t)(t_3 - t_2); dbg[3] = n_unsettled;
        }
#endif
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

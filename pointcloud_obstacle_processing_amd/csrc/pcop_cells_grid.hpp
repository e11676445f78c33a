// pcop_cells_grid.hpp - cells on a dense grid (default): occupancy bits -> cell ids, hook / union with lane = cell.
// Part of pcop_device.hpp (included from there, in this order; not a stand-alone header).
#pragma once

namespace pcop {

// ------------------------------------------------------------------------------------------
// Dense cell grid (default; the hash table of pcop_cells_hash.hpp is the fallback for grids beyond kCmapCells).
// Same cells (edge tol/1.99), same predicate, same union-find; only the way a neighbour cell is
// found differs.  The bounding box of the cloud is known, so the cells form a dense 3-D array
// with an occupancy bit each (x fastest, rows padded to 32 bits, kCellMargin empty cells all
// round) and cell id = rank of the bit = occupied cells before it:
//   * the neighbour at (dx,dy,dz) is the bit at a constant offset: no probing, no bounds checks;
//   * the 5 (or 3) x-neighbours of a row sit in one or two 32-bit words, and their ids are
//     consecutive, so one 8-byte load resolves a whole row of the stencil;
//   * ids follow the grid, so cells that are neighbours in space are neighbours in every
//     per-cell array and the lanes of a wave (64 consecutive cells) touch the same cache lines.
//   k_cgrid_setup   grid origin / size from the bounding box (margins added), or "replan"
//   k_cell_mark     grid position of every point; one byte per occupied cell (a chain frame's centroids mark theirs as
//                   k_voxel_gather / k_voxel_reduce write them: no launch of this kernel)
//   k_occ_scan      bytes -> occupancy bits + running popcount => cell ids, n_cells
//   k_cell_count    points per cell, rank inside the cell
//   k_cell_starts   exclusive scan of the counts; per-cell metadata; union-find init
//   k_cell_place    (shared with the hash path) records sorted by cell
//   k_cgrid_hook / k_cell_compress / [k_cgrid_hook2 / k_cell_compress2] / k_cgrid_union   components, lane = cell
// ------------------------------------------------------------------------------------------

// The whole block: grid origin and size from the bounding box in st->vmin / vmax (a superset of what the cell stage will
// see), the scan tiles' publish words cleared.  npts_max bounds the points that will enter the cell stage; empty: none will.
__device__ __forceinline__ void cgrid_setup_block(Frame& F, uint32_t npts_max, bool empty) {
    DevState* st = &F.st;
    if (cluster_stage_off(st) || !st->cgrid_req) return;
    uint32_t* __restrict__ ctile = F.ctile; uint32_t* __restrict__ tile_pub = F.tile_b;   // (tile_a carries the voxel stage's publish words until k_voxel_reduce is through)
    for (uint32_t t = threadIdx.x; t < kCmapCells / 32u / kVmapTile; t += kBlock) ctile[t] = 0u;
    const uint32_t ntiles = (npts_max + kScanTile - 1) / kScanTile;   // k_cell_starts publishes here
    for (uint32_t t = threadIdx.x; t < ntiles; t += kBlock) tile_pub[t] = 0u;
    if (threadIdx.x != 0) return;
    st->n_cells = 0;
    if (empty) { st->cpath = 1u; st->cmap_words = 0u; return; }
    const double ig = st->inv_cell;
    long long dim[3];
    for (int a = 0; a < 3; ++a) {
        const double lo = floor((double)st->vmin[a] * ig), hi = floor((double)st->vmax[a] * ig);
        if (!(fabs(lo) < 1e9 && fabs(hi) < 1e9)) { atomicOr(&st->err, (uint32_t)kErrReplanCpath); return; }
        st->c0[a] = (int32_t)lo - kCellMargin;
        dim[a] = (long long)hi - (long long)lo + 1 + 2 * kCellMargin;
    }
    const long long rowbits = (dim[0] + 31) / 32 * 32;
    const long long plane = rowbits * dim[1], total = plane * dim[2];
    if (dim[0] > (1 << 24) || dim[1] > (1 << 24) || dim[2] > (1 << 24) || plane > (long long)kCmapCells || total > (long long)kCmapCells) {
        atomicOr(&st->err, (uint32_t)kErrReplanCpath);
        return;
    }
    for (int a = 0; a < 3; ++a) st->c_dim[a] = (uint32_t)dim[a];
    st->c_rowbits = (uint32_t)rowbits; st->c_plane = (uint32_t)plane;
    st->cmap_words = (uint32_t)(total / 32);
    st->cpath = 1u;
}

__global__ __launch_bounds__(kBlock) void k_cgrid_setup(Frame* frames) {
    Frame& F = frames[blockIdx.y];
    cgrid_setup_block(F, F.st.c_n, F.st.c_n == 0u);
}

__global__ __launch_bounds__(kBlock) void k_cell_mark(Frame* frames) {
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const float* __restrict__ cen_x = F.cen_x; const float* __restrict__ cen_y = F.cen_y; const float* __restrict__ cen_z = F.cen_z;
    uint8_t* __restrict__ cflag = F.cflag; uint32_t* __restrict__ pt_pos = F.pt_root;   // pt_root is free until k_cell_flatten
    if (cgrid_off(st)) return;
    const bool from_in = st->c_from_input != 0;
    const float* __restrict__ x = from_in ? st->src_x.get() : cen_x;
    const float* __restrict__ y = from_in ? st->src_y.get() : cen_y;
    const float* __restrict__ z = from_in ? st->src_z.get() : cen_z;
    const uint32_t n = st->c_n;
    CellMarker mark; mark.init(st);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) pt_pos[i] = mark(st, cflag, x[i], y[i], z[i]);
}

// End of the streaming voxel path and start of the cell stage in one pass over the voxels: the centroids k_bucket_reduce left
// at their buckets' places go to their ranks (output order = voxel key order), and each marks its grid cell on the way,
// so the chain has no k_cell_mark launch.  One block per bucket.  A frame the voxel grid passed through (or gave no voxels)
// has no buckets: its cell stage sees the input cloud, marked here by all blocks together.
__global__ __launch_bounds__(kBlock) void k_voxel_gather(Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    uint8_t* __restrict__ cflag = F.cflag; uint32_t* __restrict__ pt_pos = F.pt_root;
    float* __restrict__ cx = F.cen_x; float* __restrict__ cy = F.cen_y; float* __restrict__ cz = F.cen_z;
    if (st->err || st->kind == kFrameClusterOnly) return;
    const bool cells = !cgrid_off(st);
    CellMarker mark;
    if (cells) mark.init(st);
    if (st->vpath && st->n_valid && !st->passthrough) {
        const uint32_t b = blockIdx.x, nb = st->nbuckets;
        if (b >= nb) return;
        const uint32_t* __restrict__ bucket_pub = F.bucket_pub;
        // output rank of the bucket's first voxel = voxels of the buckets before it (a few hundred words: every block adds
        // them up for itself rather than wait for a scan kernel); the last bucket's block knows M
        uint32_t before = 0;
        for (uint32_t i = threadIdx.x; i < b; i += kBlock) before += bucket_pub[i];
        const uint32_t base = block_sum(before, s_w);
        const float4* __restrict__ scratch = F.rec_b;
        const uint32_t first = F.binfo[b].x, nvox = bucket_pub[b];
        if (b == nb - 1u && threadIdx.x == 0) finish_voxel(st, base + nvox);
        for (uint32_t v = threadIdx.x; v < nvox; v += kBlock) {
            const float4 p = scratch[first + v];
            const uint32_t i = base + v;
            cx[i] = p.x; cy[i] = p.y; cz[i] = p.z;
            if (cells) pt_pos[i] = mark(st, cflag, p.x, p.y, p.z);
        }
    } else if (cells) {
        const bool from_in = st->c_from_input != 0;
        const float* __restrict__ x = from_in ? st->src_x.get() : cx;
        const float* __restrict__ y = from_in ? st->src_y.get() : cy;
        const float* __restrict__ z = from_in ? st->src_z.get() : cz;
        const uint32_t n = st->c_n;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) pt_pos[i] = mark(st, cflag, x[i], y[i], z[i]);
    }
}

// id of the occupied cell at grid position pos
__device__ __forceinline__ uint32_t cell_id_at(const uint2* __restrict__ cmap, uint32_t pos) {
    const uint2 e = cmap[pos >> 5];
    return e.y + __popc(e.x & ((1u << (pos & 31u)) - 1u));
}

__global__ __launch_bounds__(kBlock) void k_cell_count(Frame* frames) {
    Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint2* __restrict__ cmap = F.cmap; const uint32_t* __restrict__ pt_pos = F.pt_root;
    uint32_t* __restrict__ ccount = F.hcount;
    uint32_t* __restrict__ pt_slot = F.pt_slot; uint32_t* __restrict__ pt_rank = F.pt_rank;
    uint32_t* __restrict__ cell_pos = F.cell_list;
    if (cgrid_off(st)) return;
    const uint32_t n = st->c_n;
    const uint32_t lane = lane_id();
    const uint32_t nthreads = gridDim.x * blockDim.x;
    // ccount is all zero on entry (and hmin, which k_cell_flatten uses, all ones): k_emit leaves the borrowed arrays that way
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n; base += nthreads) {
        const uint32_t i = base + lane;
        const bool valid = i < n;
        uint32_t cid = 0xFFFFFFFFu, pos = 0;
        if (valid) { pos = pt_pos[i]; cid = cell_id_at(cmap, pos); }
        // runs of consecutive points in one cell (frequent: the points arrive in voxel order) share one set of atomics
        const uint32_t cprev = __shfl_up(cid, 1, kWave);
        const bool leader = valid && (lane == 0 || cid != cprev);
        const uint64_t leaders = __ballot(leader);
        const uint64_t breaks = __ballot(leader || !valid);
        uint32_t rank0 = 0;
        if (leader) {
            const uint64_t later = lane == 63 ? 0ull : (breaks >> (lane + 1));
            const uint32_t runlen = later ? (uint32_t)__ffsll((unsigned long long)later) : (uint32_t)(kWave - lane);
            rank0 = atomicAdd(ccount + cid, runlen);
            cell_pos[cid] = pos;   // every writer of a cell stores the same value
        }
        const uint64_t le = leaders & (lanemask_lt() | (1ull << lane));
        const int ll = le ? 63 - __clzll((unsigned long long)le) : 0;
        const uint32_t r = __shfl(rank0, ll, kWave);
        if (valid) { pt_slot[i] = cid; pt_rank[i] = r + (lane - (uint32_t)ll); }
    }
}

// cell_start = exclusive scan of the counts (tiles publish / add up the lower tiles); per-cell
// metadata {first record, points, priority, grid position}; union-find init.  The priority of a cell (who becomes whose
// parent) is its id: ids follow the grid, so the 13 backward cells of a cell all have smaller ones.  The first point of a
// component - extract()'s discovery order - is taken where the components are known (k_cell_flatten), so no per-cell minimum
// of point indices is kept (rounds 1-3 kept one: an atomicMin per run of points in k_cell_count, a priority load per root
// in the hook and union kernels).
__global__ __launch_bounds__(kBlock) void k_cell_starts(Frame* frames, uint32_t hook2) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    unsigned long long* __restrict__ cand = reinterpret_cast<unsigned long long*>(F.cell_info.get());
    const uint32_t* __restrict__ ccount = F.hcount;
    const uint32_t* __restrict__ cell_pos = F.cell_list;
    uint32_t* __restrict__ tile_pub = F.tile_b;
    uint32_t* __restrict__ cell_start = F.cell_start; uint32_t* __restrict__ parent = F.parent;
    uint32_t* __restrict__ comp_size = F.comp_size; uint32_t* __restrict__ comp_rank = F.comp_rank;
    uint4* __restrict__ slot_meta = F.slot_meta;
    if (cgrid_off(st)) return;
    const uint32_t nc = st->n_cells;
    const uint32_t base = blockIdx.x * kScanTile;
    if (blockIdx.x == 0 && threadIdx.x == 0) F.counters[0] = 0u;   // k_cgrid_union's list
    if (base >= nc) return;
    uint32_t v[kScanItems];
    uint32_t c = 0;
    const uint32_t s0 = base + threadIdx.x * kScanItems;
    bool stale = false;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = s0 + k < nc ? ccount[s0 + k] : 0u; c += v[k]; stale |= s0 + k < nc && v[k] == 0u; }
    // Every flagged cell was flagged by a point of this frame, so every cell has points.  One without: a flag byte that was set before
    // the frame began (the "all zero between frames" scratch was not).  The frame is given up here - the kernels after this one never
    // follow such a cell (and clamp what they read), the host scrubs every flag and runs the batch again.
    if (stale) atomicOr(&st->err, (uint32_t)kErrStale);
    uint32_t tot;
    uint32_t run = block_excl_scan(c, s_w, &tot);
    if (threadIdx.x == 0) st_agent(tile_pub + blockIdx.x, tot + 1u);
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t q = ld_agent(tile_pub + t), spins = 0;
        while (q == 0u) {
            if (++spins > st->spin_limit) { atomicOr(&st->err, (uint32_t)kErrInternal); q = 1u; break; }
            __builtin_amdgcn_s_sleep(2);
            q = ld_agent(tile_pub + t);
        }
        mine += q - 1u;
    }
    __syncthreads();
    run += block_sum(mine, s_w);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t id = s0 + k;
        if (id < nc) {
            cell_start[id] = run;
            parent[id] = id; comp_size[id] = 0u; comp_rank[id] = 0xFFFFFFFFu;
            slot_meta[id] = make_uint4(run, v[k], id, cell_pos[id]);   // {first record, points, priority = id, grid position}
            if (hook2) cand[id] = ~0ull;   // k_cgrid_hook2's candidate parent of a root: none yet
        }
        run += v[k];
    }
}

// One lane walks the cross pairs of two cells: is some pair within the radius?
__device__ __forceinline__ bool lane_cells_hit(const float4* __restrict__ rec, uint32_t a0, uint32_t an, uint32_t b0,
                                               uint32_t bn, float r2) {
    for (uint32_t ia = 0; ia < an; ++ia) {
        const float4 pa = rec[a0 + ia];
        for (uint32_t ib = 0; ib < bn; ++ib)
            if (l2_simple(pa, rec[b0 + ib]) < r2) return true;
    }
    return false;
}

// Occupied cells among `width` (<= 5) consecutive grid positions starting at `pos`: bit k of the result
// is position pos + k; *id0 = id of the first occupied one, the others follow consecutively.
__device__ __forceinline__ uint32_t row_window(const uint2* __restrict__ cmap, uint32_t pos, uint32_t width, uint32_t* id0) {
    const uint32_t w = pos >> 5, sh = pos & 31u;
    const uint2 e0 = cmap[w];
    uint32_t bits = e0.x >> sh;
    uint32_t before = e0.y + __popc(e0.x & ((1u << sh) - 1u));   // occupied cells before pos
    if (sh + width > 32u) bits |= cmap[w + 1].x << (32u - sh);   // sh > 0 here
    *id0 = before;
    return bits & ((1u << width) - 1u);
}

// parent[c] = the adjacent cell of smallest id among the 13 backward cells of the 3x3x3
// neighbourhood (5 rows; all of them have smaller ids than c).  No atomics: a forest.
// Two rounds of independent loads (row windows, then the metadata of the 13 neighbour slots; a slot
// without a neighbour re-reads the cell's own), then pair tests only against cells of smaller id than the best so far.
__global__ __launch_bounds__(kBlock) void k_cgrid_hook(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    uint32_t* __restrict__ parent = F.parent;
    if (cgrid_off(st)) return;
    const uint32_t nc = st->n_cells, c_n = st->c_n, grid_cells = st->cmap_words * 32u;
    const float r2 = st->r2;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
        const uint4 ma = slot_meta[c];   // {first record, points, priority, grid position}
        if (ma.y == 0u || ma.x + ma.y > c_n || ma.w >= grid_cells) continue;   // (a flagged cell no point of this frame lies in has no position either: never followed; a record range or a position that leaves the frame: never read)
        // rows: (dy,dz) = (-1,-1) (0,-1) (1,-1) (-1,0) with dx = -1..1, and the cell's own row with dx = -1
        uint2 e0[5];
        uint32_t e1[5], sh[5], wd[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            uint32_t pos;
            if (r < 3) { pos = ma.w - plane + (uint32_t)(r - 1) * rowbits - 1u; wd[r] = 3u; }
            else if (r == 3) { pos = ma.w - rowbits - 1u; wd[r] = 3u; }
            else { pos = ma.w - 1u; wd[r] = 1u; }
            sh[r] = pos & 31u;
            e0[r] = cmap[pos >> 5];
            e1[r] = cmap[(pos >> 5) + 1u].x;
        }
        uint4 mb[13];
        uint32_t nid[13];
        int slot = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            uint32_t bits = e0[r].x >> sh[r];
            if (sh[r] + wd[r] > 32u) bits |= (e1[r] << 1) << (31u - sh[r]);
            const uint32_t cnt = __popc(bits & ((1u << wd[r]) - 1u));
            const uint32_t id0 = e0[r].y + __popc(e0[r].x & ((1u << sh[r]) - 1u));
            const int w = r == 4 ? 1 : 3;
#pragma unroll
            for (int k = 0; k < w; ++k) {
                nid[slot] = (uint32_t)k < cnt ? id0 + (uint32_t)k : c;   // the ids of a row's occupied cells are consecutive
                mb[slot] = slot_meta[nid[slot]];
                ++slot;
            }
        }
        uint32_t best = c;   // (a slot without a neighbour holds c itself)
#pragma unroll
        for (int q = 0; q < 13; ++q)
            if (nid[q] < best && lane_cells_hit(rec, ma.x, ma.y, mb[q].x, mb[q].y, r2)) best = nid[q];
        if (best != c) parent[c] = best;
    }
}

// A second hook round, one level up (for frames with so many cells that the first round leaves hundreds of thousands of trees:
// C5's three million cells hook into 251 k trees of a dozen cells each, and the union kernel then merges them pair by pair,
// by CAS, mostly into ONE component).  After the compress every cell knows its tree's root.  Here every cell looks at the same
// 13 backward cells once more: a neighbour in ANOTHER tree that a pair test connects is an edge between two trees, and the root
// of larger priority value takes the other root as its candidate parent if it is the smallest it has seen - atomicMin on one
// 64-bit word {priority, root} per root, fire and forget, no finds, no CAS loops.  Links go to strictly smaller priorities:
// still a forest, and a root is still the cell of smallest priority of its tree.  k_cell_compress2 then follows a cell's root
// through the candidate links to the end.  Roots that are not local minima among their neighbours disappear: the trees left
// for k_cgrid_union shrink by the factor the first round achieved once more.
__global__ __launch_bounds__(kBlock) void k_cgrid_hook2(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    const uint32_t* __restrict__ parent = F.parent;
    unsigned long long* __restrict__ cand = reinterpret_cast<unsigned long long*>(F.cell_info.get());   // (the hash path's per-cell records: idle on the grid)
    if (cgrid_off(st) || st->debug_skip_union == 1u) return;
    const uint32_t nc = st->n_cells, c_n = st->c_n, grid_cells = st->cmap_words * 32u;
    const float r2 = st->r2;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
        const uint4 ma = slot_meta[c];   // {first record, points, priority, grid position}
        if (ma.y == 0u || ma.x + ma.y > c_n || ma.w >= grid_cells) continue;
        const uint32_t ra = parent[c];   // its tree's root (k_cell_compress ran)
        uint2 e0[5];
        uint32_t e1[5], sh[5], wd[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {   // the rows of k_cgrid_hook
            uint32_t pos;
            if (r < 3) { pos = ma.w - plane + (uint32_t)(r - 1) * rowbits - 1u; wd[r] = 3u; }
            else if (r == 3) { pos = ma.w - rowbits - 1u; wd[r] = 3u; }
            else { pos = ma.w - 1u; wd[r] = 1u; }
            sh[r] = pos & 31u;
            e0[r] = cmap[pos >> 5];
            e1[r] = cmap[(pos >> 5) + 1u].x;
        }
        uint32_t nid[13], rb[13];
        int slot = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            uint32_t bits = e0[r].x >> sh[r];
            if (sh[r] + wd[r] > 32u) bits |= (e1[r] << 1) << (31u - sh[r]);
            const uint32_t cnt = __popc(bits & ((1u << wd[r]) - 1u));
            const uint32_t id0 = e0[r].y + __popc(e0[r].x & ((1u << sh[r]) - 1u));
            const int w = r == 4 ? 1 : 3;
#pragma unroll
            for (int k = 0; k < w; ++k) {
                nid[slot] = (uint32_t)k < cnt ? min(id0 + (uint32_t)k, nc - 1u) : c;
                rb[slot] = parent[nid[slot]];
                ++slot;
            }
        }
        bool any = false;
#pragma unroll
        for (int q = 0; q < 13; ++q) any |= rb[q] != ra;
        if (!any) continue;
        const uint32_t pa = ra;   // (priority = id)
#pragma unroll 1
        for (int q = 0; q < 13; ++q) {
            if (rb[q] == ra) continue;
            const uint32_t pb = rb[q];
            const uint32_t hi = pa > pb ? ra : rb[q], lo = pa > pb ? rb[q] : ra;
            const unsigned long long key = ((unsigned long long)min(pa, pb) << 32) | lo;
            if (key >= cand[hi]) continue;   // (a stale read only costs a pair test: the atomicMin decides)
            const uint4 mb = slot_meta[nid[q]];
            if (mb.y == 0u || mb.x + mb.y > c_n) continue;
            if (lane_cells_hit(rec, ma.x, ma.y, mb.x, mb.y, r2)) atomicMin(cand + hi, key);
        }
    }
}

// parent[c] = the end of the candidate links that start at c's root (read-only links: no races; strictly falling priorities: no cycles)
__global__ __launch_bounds__(kBlock) void k_cell_compress2(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    uint32_t* __restrict__ parent = F.parent;
    const unsigned long long* __restrict__ cand = reinterpret_cast<const unsigned long long*>(F.cell_info.get());
    if (cgrid_off(st) || st->debug_skip_union == 1u) return;
    const uint32_t nc = st->n_cells;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
        uint32_t r = min(parent[c], nc - 1u);
        bool moved = false;
        for (uint32_t hops = 0; hops < nc; ++hops) {
            const unsigned long long k = cand[r];
            if (k == ~0ull) break;
            r = min((uint32_t)k, nc - 1u); moved = true;
        }
        if (moved) parent[c] = r;
    }
}

// Forward half of the 5x5x5 stencil (13 rows, 62 cells).  Neighbours whose parent word equals the
// cell's are in its set already, and after hook + compress that is nearly all of them, so the common
// case is made of two rounds of independent loads: the 13 row windows, then the parent words of
// every occupied neighbour.  Only a cell that sees a different parent word walks its rows again and
// looks properly (roots, pair test, union by CAS).
__device__ __forceinline__ void union_row(int r, uint32_t at, uint32_t rowbits, uint32_t plane, uint32_t* pos, uint32_t* width) {
    // rows: own row dx = 1..2; (dy = 1..2, dz = 0) and (dy = -2..2, dz = 1..2) with dx = -2..2
    if (r == 0) { *pos = at + 1u; *width = 2u; }
    else if (r < 3) { *pos = at + (uint32_t)r * rowbits - 2u; *width = 5u; }
    else { const int q = r - 3; *pos = at + (uint32_t)(1 + q / 5) * plane + (uint32_t)(q % 5 - 2) * rowbits - 2u; *width = 5u; }
}

constexpr uint32_t kFixWaveMax = 1u << 16;   // cells of a frame up to which k_cgrid_fix may be asked for

// Cells that see a neighbour of another tree: either the lane looks at those neighbours itself, here, or the cell goes on
// the list of k_cgrid_fix, which gives it a wave.  The wave pays when a cell has many occupied neighbours to look at (a
// surface sampled several points to the cell: 20 of the 62), not when the cells are sparse (a lane's walk is then short and
// 62 lanes mostly idle), and only for a single frame, where latency counts: in a batch the waves that wait for a walking
// lane cost nothing the other contexts' kernels could not use, while a wave per cell takes twenty times the wave-slot time.
__device__ __forceinline__ bool fix_by_wave(const DevState* st, uint32_t single_frame) {
    return single_frame && st->n_cells <= kFixWaveMax && st->c_n >= 3u * st->n_cells;
}

__global__ __launch_bounds__(kBlock) void k_cgrid_union(const Frame* frames, uint32_t single_frame) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    uint32_t* __restrict__ parent = F.parent;
    uint32_t* __restrict__ fix_list = F.idx_a;   // (idle until the clusters' CSR)
    uint32_t* __restrict__ fix_count = F.counters;
    if (cgrid_off(st) || st->debug_skip_union == 1u) return;
    const bool scan_only = st->debug_skip_union == 2u;
    const bool list = fix_by_wave(st, single_frame);
    const uint32_t nc = st->n_cells, c_n = st->c_n, grid_cells = st->cmap_words * 32u;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
        const uint4 ma = slot_meta[c];
        if (ma.y == 0u || ma.x + ma.y > c_n || ma.w >= grid_cells) continue;   // (as in k_cgrid_hook)
        const uint32_t pa = ld_agent(parent + c);
        // round 1: the windows of all rows.  Every load is issued before any is used and none sits under a
        // branch (the second word of a window is fetched whether it is needed or not).
        uint2 e0[13];
        uint32_t e1[13], sh[13], wd[13];
#pragma unroll
        for (int r = 0; r < 13; ++r) {
            uint32_t pos;
            union_row(r, ma.w, rowbits, plane, &pos, &wd[r]);
            sh[r] = pos & 31u;
            e0[r] = cmap[pos >> 5];
            e1[r] = cmap[(pos >> 5) + 1u].x;
        }
        uint32_t win[13], id0[13];
#pragma unroll
        for (int r = 0; r < 13; ++r) {
            uint32_t bits = e0[r].x >> sh[r];
            if (sh[r] + wd[r] > 32u) bits |= (e1[r] << 1) << (31u - sh[r]);
            win[r] = bits & ((1u << wd[r]) - 1u);
            id0[r] = e0[r].y + __popc(e0[r].x & ((1u << sh[r]) - 1u));
        }
        // round 2: parent words of the occupied neighbours (the ids of a row's occupied cells are consecutive).
        // Plain loads are enough here: a parent word, however stale, names a member of the neighbour's set, and
        // sets only ever merge, so equal words prove "same set".  Slots without a neighbour re-read the cell's own.
        bool differs = false;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t got[7 * 5];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int r = half * 7 + q;
                if (r < 13) {
                    const uint32_t cnt = __popc(win[r]);
#pragma unroll
                    for (uint32_t k = 0; k < 5; ++k) got[q * 5 + k] = parent[k < cnt ? id0[r] + k : c];
                }
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int r = half * 7 + q;
                if (r < 13) {
                    const uint32_t cnt = __popc(win[r]);
#pragma unroll
                    for (uint32_t k = 0; k < 5; ++k) differs |= (k < cnt) && (got[q * 5 + k] != pa);
                }
            }
        }
        if (!differs || scan_only) continue;
        if (!list) {
            // some neighbour may be in another set: walk the rows again and look properly at those whose parent word
            // differs from the cell's (roots, pair test, union by CAS)
            const float4* __restrict__ rec = F.rec_a;
            const uint32_t* __restrict__ prio = nullptr;   // (priority = id)
            const float r2 = st->r2;
            uint32_t ra = uf_find(parent, c);
#pragma unroll 1
            for (int r = 0; r < 13; ++r) {
                uint32_t id = id0[r];
                for (uint32_t left = __popc(win[r]); left; --left, ++id) {
                    const uint32_t pb = ld_agent(parent + id);
                    if (pb == pa || pb == ra) continue;
                    const uint32_t rb = uf_find(parent, id);
                    if (rb == ra) continue;
                    ra = uf_find(parent, ra);   // the cell's root may have been hung under another one meanwhile
                    if (rb == ra) continue;
                    const uint4 mb = slot_meta[id];
                    if (lane_cells_hit(rec, ma.x, ma.y, mb.x, mb.y, r2)) { uf_union(parent, prio, ra, rb); ra = uf_find(parent, ra); }
                }
            }
            continue;
        }
        // the cell goes on the list of k_cgrid_fix
        const uint64_t m = __ballot(true);
        const uint32_t first = (uint32_t)__ffsll((long long)m) - 1u;
        uint32_t at = 0;
        if (lane_id() == first) at = atomicAdd(fix_count, (uint32_t)__popcll(m));
        at = (uint32_t)__builtin_amdgcn_readlane((int)at, (int)first);
        fix_list[at + (uint32_t)__popcll(m & lanemask_lt())] = c;
    }
}

// The cells k_cgrid_union listed (fix_by_wave): one WAVE per cell, lane = one of the 62 forward neighbours; roots, pair tests
// side by side, unions by CAS one at a time (union_fixup, shared with the hash-table kernels).
__global__ __launch_bounds__(kBlock) void k_cgrid_fix(const Frame* frames) {
    __shared__ float4 s_pa[kWavesPerBlock][kCellPtsCap];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    const uint32_t* __restrict__ prio = nullptr;   // (priority = id)
    uint32_t* __restrict__ parent = F.parent;
    const uint32_t* __restrict__ fix_list = F.idx_a;
    if (cgrid_off(st) || st->debug_skip_union || !fix_by_wave(st, 1u)) return;
    const uint32_t count = min(F.counters[0], st->n_cells);
    const float r2 = st->r2;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    const uint32_t lane = lane_id(), wv = threadIdx.x / kWave;
    int dx = 0, dy = 0, dz = 0;
    const bool in = lane < (uint32_t)kStencil;
    if (in) stencil_offset(lane, &dx, &dy, &dz);
    const uint32_t off = (uint32_t)dz * plane + (uint32_t)dy * rowbits + (uint32_t)dx;   // (wraps like the signed sum)
    for (uint32_t w = blockIdx.x * kWavesPerBlock + wv; w < count; w += gridDim.x * kWavesPerBlock) {
        const uint32_t a = fix_list[w];
        const uint4 ma = slot_meta[a];
        const uint32_t pa = ld_agent(parent + a);
        const uint32_t npos = ma.w + off;           // the grid's margins keep the whole stencil inside the array
        const uint2 e = cmap[npos >> 5];
        const bool occ = in && ((e.x >> (npos & 31u)) & 1u);
        const uint32_t b = occ ? e.y + (uint32_t)__popc(e.x & ((1u << (npos & 31u)) - 1u)) : a;
        bool need = occ && ld_agent(parent + b) != pa;
        if (!__any(need)) continue;
        const uint32_t ra = uf_find(parent, a);
        need = need && uf_find(parent, b) != ra;
        if (__any(need)) union_fixup(rec, slot_meta, prio, parent, s_pa[wv], a, ma.x, ma.y, b, need, r2);
    }
}

}  // namespace pcop

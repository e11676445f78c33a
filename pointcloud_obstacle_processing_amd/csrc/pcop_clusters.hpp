// pcop_clusters.hpp - components -> clusters: root per point, size filter and discovery order, CSR partition, results to the host, PointWithRad.
// Part of pcop_device.hpp (included from there, in this order; not a stand-alone header).
#pragma once

namespace pcop {

// Root cell of every point; component sizes; on the dense grid also the first point of every component (hmin[root]: on the hash
// path the root is the cell that holds it and has it already).  Each wave walks a contiguous chunk of points 64
// at a time; lanes whose points share a root add up in registers, and the count of the root a
// wave is currently running through is carried across steps, so a component that spans a long
// stretch of points (a ground plane) costs one atomic per wave, not one per step.
__global__ __launch_bounds__(kBlock) void k_cell_flatten(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ pt_slot = F.pt_slot;
    uint32_t* __restrict__ parent = F.parent; uint32_t* __restrict__ pt_root = F.pt_root;
    uint32_t* __restrict__ comp_size = F.comp_size; uint32_t* __restrict__ first_pt = F.hmin;
    if (cluster_stage_off(st)) return;
    const bool firsts = st->cpath != 0;
    const uint32_t n = st->c_n;
    {   // k_kept_apply publishes one 64-bit word per tile of points in `table` (free between the voxel and the CSR passes)
        unsigned long long* __restrict__ tile_pub = reinterpret_cast<unsigned long long*>(F.table.get());
        const uint32_t ntiles = (n + kScanTile - 1) / kScanTile;
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) tile_pub[t] = 0ull;
    }
    const uint32_t lane = lane_id();
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
    const uint32_t steps = (n + kWave - 1) / kWave;
    const uint32_t per = (steps + nwaves - 1) / nwaves;
    const uint32_t s0 = wave * per, s1 = min(s0 + per, steps);
    uint32_t carry_root = 0xFFFFFFFFu, carry_cnt = 0;   // wave-uniform
    for (uint32_t sidx = s0; sidx < s1; ++sidx) {
        const uint32_t i = sidx * kWave + lane;
        const bool valid = i < n;
        uint32_t r = 0xFFFFFFFFu;
        if (valid) { r = uf_find(parent, pt_slot[i]); pt_root[i] = r; }
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t rl = __shfl(r, leader, kWave);
            const uint64_t same = __ballot(valid && r == rl);
            const uint32_t cnt = (uint32_t)__popcll(same);
            if (rl == carry_root) {
                carry_cnt += cnt;
            } else {
                if (carry_cnt && lane == 0) atomicAdd(comp_size + carry_root, carry_cnt);
                carry_root = rl; carry_cnt = cnt;
                // the wave walks its points in index order: the first one of a run is the smallest the wave will see of that root
                if (firsts && lane == 0) atomicMin(first_pt + rl, sidx * kWave + (uint32_t)leader);
            }
            todo &= ~same;
        }
    }
    if (carry_cnt && lane == 0) atomicAdd(comp_size + carry_root, carry_cnt);
}

// extractEuclideanClusters' size filter and discovery order.  Point i opens a kept cluster iff it
// is the first point of its component (hmin[root] == i) and min <= size <= max.
__device__ __forceinline__ uint32_t kept_size(uint32_t i, const uint32_t* __restrict__ pt_root,
                                              const uint32_t* __restrict__ hmin, const uint32_t* __restrict__ comp_size,
                                              uint32_t lo, uint32_t hi, uint32_t* root) {
    const uint32_t r = pt_root[i];
    *root = r;
    if (hmin[r] != i) return 0u;
    const uint32_t s = comp_size[r];
    return (s >= lo && s <= hi) ? s : 0u;
}

// comp_rank[root] = discovery rank of the kept cluster; offsets_out[rank] = CSR offset.  One kernel: a
// tile publishes {clusters it opens + 1, their points} as one 64-bit word and adds up the words of the
// lower tiles (bounded wait, as in k_voxel_reduce; k_cell_flatten zeroes the words).
__global__ __launch_bounds__(kBlock) void k_kept_apply(Frame* frames, uint32_t npass_csr, uint32_t maxbits) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const uint32_t* __restrict__ pt_root = F.pt_root; const uint32_t* __restrict__ hmin = F.hmin;
    const uint32_t* __restrict__ comp_size = F.comp_size;
    unsigned long long* __restrict__ tile_pub = reinterpret_cast<unsigned long long*>(F.table.get());
    uint32_t* __restrict__ comp_rank = F.comp_rank; uint32_t* __restrict__ offsets_out = F.d_offsets;
    if (cluster_stage_off(st)) return;
    if (threadIdx.x == 0) F.tile_a[blockIdx.x] = 0u;   // the flags of k_csr_onepass (one per 2048 points; a block here has 1024)
    const uint32_t n = st->c_n;
    if (n == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->n_kept = 0; st->n_kept_pts = 0; offsets_out[0] = 0; make_plan(&st->plan_csr, 1, npass_csr, maxbits); }
        return;
    }
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    const uint32_t lo = st->min_cluster, hi = st->max_cluster;
    uint32_t v[kScanItems], root[kScanItems];
    uint32_t c = 0, s = 0;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
    // root, first point and size of the component of each of the thread's points: three rounds of independent loads
    uint32_t first[kScanItems], size[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) root[k] = pt_root[min(i0 + k, n - 1u)];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { first[k] = hmin[root[k]]; size[k] = comp_size[root[k]]; }
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        // point i opens a kept cluster iff it is the first point of its component and min <= size <= max
        v[k] = (i < n && first[k] == i && size[k] >= lo && size[k] <= hi) ? size[k] : 0u;
        c += v[k] ? 1u : 0u; s += v[k];
    }
    uint32_t totc, tots;
    uint32_t rc = block_excl_scan(c, s_w, &totc);
    uint32_t rs = block_excl_scan(s, s_w, &tots);
    if (threadIdx.x == 0)
        __hip_atomic_store(tile_pub + blockIdx.x, ((unsigned long long)(totc + 1u) << 32) | tots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t mine_c = 0, mine_s = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        unsigned long long q = __hip_atomic_load(tile_pub + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while ((q >> 32) == 0ull) {
            if (++spins > st->spin_limit) { atomicOr(&st->err, (uint32_t)kErrInternal); q = 1ull << 32; break; }
            __builtin_amdgcn_s_sleep(2);
            q = __hip_atomic_load(tile_pub + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        mine_c += (uint32_t)(q >> 32) - 1u; mine_s += (uint32_t)q;
    }
    __syncthreads();
    const uint32_t before_c = block_sum(mine_c, s_w);
    const uint32_t before_s = block_sum(mine_s, s_w);
    rc += before_c; rs += before_s;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (v[k]) { comp_rank[root[k]] = rc; offsets_out[rc] = rs; ++rc; rs += v[k]; }
    }
    if (threadIdx.x == 0 && base + kScanTile >= n) {   // the last tile knows the totals
        const uint32_t nk = before_c + totc, np = before_s + tots;
        st->n_kept = nk; st->n_kept_pts = np;
        offsets_out[nk] = np;
        // keys of the CSR partition are 0..n_kept (n_kept = "not in a kept cluster")
        if (!make_plan(&st->plan_csr, bits_for((uint64_t)nk + 1), npass_csr, maxbits)) atomicOr(&st->err, (uint32_t)kErrReplanCsr);
    }
}

// ------------------------------------------------------------------------------------------
// CSR: stable partition of point indices 0..c_n-1 by cluster rank (LSD over plan_csr digits).
// Pass 0 reads the identity permutation, later passes read the previous output.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool csr_pass_info(const DevState* st, int pass, uint32_t* count, uint32_t* bits,
                                              uint32_t* shift) {
    if (cluster_stage_off(st)) return false;
    *count = st->c_n; *bits = st->plan_csr.bits[pass]; *shift = st->plan_csr.shift[pass];
    return st->n_kept != 0;
}

__device__ __forceinline__ uint32_t point_label(uint32_t p, const uint32_t* __restrict__ pt_root,
                                                const uint32_t* __restrict__ comp_rank, uint32_t trash) {
    return min(comp_rank[pt_root[p]], trash);
}

// Cluster rank of each of a thread's kItems records (count = "not there").  Three rounds of independent loads
// (point index, root, rank): a load under a per-record branch would be waited for at the branch's join, which
// would put the records' dependent chains one after the other.
__device__ __forceinline__ void point_labels(uint32_t wbase, uint32_t count, int pass, const uint32_t* __restrict__ idx_in,
                                             const uint32_t* __restrict__ pt_root, const uint32_t* __restrict__ comp_rank,
                                             uint32_t trash, uint32_t* pt, uint32_t* label) {
    const uint32_t last = count - 1u;   // count >= 1 here
#pragma unroll
    for (int j = 0; j < kItems; ++j) { const uint32_t i = min(wbase + j * kWave, last); pt[j] = pass == 0 ? i : idx_in[i]; }
    uint32_t root[kItems];
#pragma unroll
    for (int j = 0; j < kItems; ++j) root[j] = pt_root[pt[j]];
#pragma unroll
    for (int j = 0; j < kItems; ++j) label[j] = min(comp_rank[root[j]], trash);
}

template <int MAXB>
__global__ __launch_bounds__(kBlock) void k_csr_hist(const Frame* frames, int pass) {
    __shared__ uint32_t s_h[(1 << MAXB) + 1];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ idx_in = pass == 0 ? nullptr : (((pass - 1) & 1) ? F.idx_b : F.idx_a);
    const uint32_t* __restrict__ pt_root = F.pt_root; const uint32_t* __restrict__ comp_rank = F.comp_rank;
    uint32_t* __restrict__ table = F.table;
    uint32_t count, bits, shift;
    if (!csr_pass_info(st, pass, &count, &bits, &shift)) return;
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= count) return;
    const uint32_t nb = (1u << bits) + 1u, trash = st->n_kept;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) s_h[b] = 0;
    __syncthreads();
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t wbase = tile_base + (threadIdx.x / kWave) * (kItems * kWave) + lane_id();
    uint32_t pt[kItems], label[kItems];
    point_labels(wbase, count, pass, idx_in, pt_root, comp_rank, trash, pt, label);
#pragma unroll
    for (int j = 0; j < kItems; ++j)
        if (wbase + j * kWave < count) atomicAdd(&s_h[(label[j] >> shift) & mask], 1u);
    __syncthreads();
    const uint32_t ntiles = (count + kTile - 1) / kTile;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) table[(size_t)b * ntiles + blockIdx.x] = s_h[b];
}

template <int MAXB>
__global__ __launch_bounds__(kBlock) void k_csr_scatter(const Frame* frames, int pass) {
    __shared__ uint32_t s_hist[kWavesPerBlock * ((1 << MAXB) + 1)];
    __shared__ uint32_t s_base[(1 << MAXB) + 1];
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ idx_in = pass == 0 ? nullptr : (((pass - 1) & 1) ? F.idx_b : F.idx_a);
    uint32_t* __restrict__ idx_out = (pass & 1) ? F.idx_b : F.idx_a;
    const uint32_t* __restrict__ pt_root = F.pt_root; const uint32_t* __restrict__ comp_rank = F.comp_rank;
    const uint32_t* __restrict__ table_x = F.table_x; const uint32_t* __restrict__ bin_total = F.bin_total;
    uint32_t count, bits, shift;
    if (!csr_pass_info(st, pass, &count, &bits, &shift)) return;
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= count) return;
    const uint32_t nb = (1u << bits) + 1u, trash = st->n_kept;
    for (uint32_t b = threadIdx.x; b < kWavesPerBlock * nb; b += kBlock) s_hist[b] = 0;
    scan_bins_to_lds<MAXB>(bin_total, nb, s_base, s_w);
    const uint32_t w = threadIdx.x / kWave;
    uint32_t* hist_w = s_hist + w * nb;
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t wbase = tile_base + w * (kItems * kWave) + lane_id();
    uint32_t pt[kItems], label[kItems], dig[kItems], lrank[kItems];
    point_labels(wbase, count, pass, idx_in, pt_root, comp_rank, trash, pt, label);
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        const bool active = i < count;
        const uint32_t d = active ? (label[j] >> shift) & mask : 0u;
        // few tiles, latency-bound: the ballots (pure ALU) beat the LDS round trips of match_digit_lds here
        const uint64_t peers = match_digit(d, bits, active);
        const uint32_t rank = __popcll(peers & lanemask_lt());
        uint32_t old = 0;
        if (active) old = hist_w[d];
        if (active && rank == 0) hist_w[d] = old + (uint32_t)__popcll(peers);
        dig[j] = d; lrank[j] = old + rank;
    }
    __syncthreads();
    const uint32_t ntiles = (count + kTile - 1) / kTile;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) {
        uint32_t run = s_base[b] + table_x[(size_t)b * ntiles + blockIdx.x];
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

// The CSR partition in ONE kernel, for the usual case of a single 8-bit pass (at most 255 kept clusters).  What the
// general form needs three launches for - per-tile counts, their scan over the tiles, the scatter - collapses because the
// bins are the kept clusters: where a cluster's indices start is its CSR offset (k_kept_apply has it), the points outside
// kept clusters need not be written at all, and the only thing a tile must wait for is how many points of each cluster the
// tiles before it hold.  A tile publishes its counts and adds up the lower tiles' (bounded wait; lower tiles never wait
// on higher ones and blocks are dispatched in index order; k_kept_apply zeroes the flags).  Used for single frames: it
// saves two launches (10 us of a 0.21 ms frame); in a batch with other contexts in flight the waiting tiles hold wave
// slots and the three-launch form is 1.7 x faster.
constexpr uint32_t kCsrOnepassMax = 1u << 18;   // points (128 tiles): beyond, the tiles' look-back (quadratic in the tiles) costs more than two launches

__global__ __launch_bounds__(kBlock) void k_csr_onepass(const Frame* frames) {
    constexpr uint32_t NB = (1u << kNarrowBits) + 1u;
    __shared__ uint32_t s_hist[kWavesPerBlock * NB];
    __shared__ uint32_t s_tile[NB];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    uint32_t* __restrict__ idx_out = F.idx_a;
    const uint32_t* __restrict__ pt_root = F.pt_root; const uint32_t* __restrict__ comp_rank = F.comp_rank;
    const uint32_t* __restrict__ d_offsets = F.d_offsets;
    uint32_t* hist_pub = F.table_x;   // [tile][NB]  (the scanned tables are idle: no k_table_scan in this form)
    uint32_t* flags = F.tile_a;
    uint32_t count, bits, shift;
    if (!csr_pass_info(st, 0, &count, &bits, &shift)) return;
    if (count > kCsrOnepassMax) {   // too many tiles for the look-back: the host re-runs the frame with the three-launch form
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(const_cast<uint32_t*>(&st->err), (uint32_t)kErrReplanCsr);
        return;
    }
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= count) return;
    const uint32_t nb = (1u << bits) + 1u, trash = st->n_kept;
    for (uint32_t b = threadIdx.x; b < kWavesPerBlock * nb; b += kBlock) s_hist[b] = 0;
    __syncthreads();
    const uint32_t w = threadIdx.x / kWave;
    uint32_t* hist_w = s_hist + w * nb;
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t wbase = tile_base + w * (kItems * kWave) + lane_id();
    uint32_t pt[kItems], label[kItems], dig[kItems], lrank[kItems];
    point_labels(wbase, count, 0, nullptr, pt_root, comp_rank, trash, pt, label);
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        const bool active = i < count;
        const uint32_t d = active ? label[j] & mask : 0u;
        const uint64_t peers = match_digit(d, bits, active);
        const uint32_t rank = __popcll(peers & lanemask_lt());
        uint32_t old = 0;
        if (active) old = hist_w[d];
        if (active && rank == 0) hist_w[d] = old + (uint32_t)__popcll(peers);
        dig[j] = d; lrank[j] = old + rank;
    }
    __syncthreads();
    // per bin: the waves' counts -> where each wave's points start inside the tile; the tile's count goes out
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) {
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < kWavesPerBlock; ++q) { const uint32_t t = s_hist[q * nb + b]; s_hist[q * nb + b] = run; run += t; }
        st_agent(hist_pub + (size_t)blockIdx.x * NB + b, run);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) st_agent(flags + blockIdx.x, 1u);
    // wait for the tiles before this one (a thread per tile), then add up their counts: loads without a wait between them
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t spins = 0;
        while (ld_agent(flags + t) == 0u) {
            if (++spins > st->spin_limit) { atomicOr(const_cast<uint32_t*>(&st->err), (uint32_t)kErrInternal); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __threadfence();
    __syncthreads();
    const uint32_t* __restrict__ pub = hist_pub;   // (final once its tile's flag is up; this CU has not read it before)
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) {
        uint32_t sum = b < trash ? d_offsets[b] : 0u;   // the cluster's CSR offset + its points in the lower tiles
        uint32_t t = 0;
        for (; t + 8u <= blockIdx.x; t += 8u) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) v[u] = pub[(size_t)(t + u) * NB + b];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) sum += v[u];
        }
        for (; t < blockIdx.x; ++t) sum += pub[(size_t)t * NB + b];
        s_tile[b] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < count && dig[j] < trash) idx_out[s_tile[dig[j]] + hist_w[dig[j]] + lrank[j]] = pt[j];
    }
}

// ------------------------------------------------------------------------------------------
// end of frame: results -> pinned host memory (zero-copy stores), hash table reset
// ------------------------------------------------------------------------------------------
#ifndef PCOP_DIAG_NO_INDEX_COPY
#define PCOP_DIAG_NO_INDEX_COPY 0
#endif
#ifndef PCOP_EMIT_VEC
#define PCOP_EMIT_VEC 0   // 1 (tuning build, tools/ab_libs.sh): k_emit copies the cluster indices to the host 16 bytes a lane - built at the end of round 4, not measured yet
#endif
struct HostOut {
    DevState* state;
    uint32_t* offsets;     // n_kept + 1
    uint32_t* indices;     // n_kept_pts, discovery order
    float* cloud;          // voxel cloud in st->out_layout
};

__device__ __forceinline__ void box_true_grid(Frame& F, MinMaxPartial* s_p);

__global__ __launch_bounds__(kBlock) void k_emit(const Frame* frames, uint32_t npass_csr) {
    __shared__ MinMaxPartial s_p[kBlock];   // (keys on the host's box: the first block reduces the tiles' bounds)
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const HostOut out{F.h_state, F.h_offsets, F.h_indices, F.h_cloud};
    const uint32_t* __restrict__ d_offsets = F.d_offsets;
    const uint32_t* __restrict__ d_indices = ((npass_csr - 1) & 1) ? F.idx_b : F.idx_a;
    const float* __restrict__ cen_x = F.cen_x; const float* __restrict__ cen_y = F.cen_y; const float* __restrict__ cen_z = F.cen_z;
    const CellTable tab{F.hkey, F.hcount, F.hmin};
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    // reset the cell table for the next frame: the occupied slots are listed in cell_list; after an
    // error the list may be incomplete, so everything below hash_mask + 1 is cleared instead
    if (st->kind != kFrameVoxelOnly && !st->trivial && st->cpath) {
        // dense cell grid: the per-cell counters borrowed the hash table's arrays; give them back clean
        const uint32_t nc = st->err ? st->c_n : st->n_cells;
        for (uint32_t i = tid; i < nc; i += nth) { tab.hcount[i] = 0u; tab.hmin[i] = 0xFFFFFFFFu; }
    } else if (st->kind != kFrameVoxelOnly && !st->trivial && st->hash_mask) {
        if (st->err) {
            const uint32_t cap = st->hash_mask + 1u;
            for (uint32_t i = tid; i < cap; i += nth) { tab.hkey[i] = 0ull; tab.hcount[i] = 0u; tab.hmin[i] = 0xFFFFFFFFu; }
        } else {
            const uint32_t nc = st->n_cells;
            const uint32_t* __restrict__ cell_list = F.cell_list;
            for (uint32_t i = tid; i < nc; i += nth) {
                const uint32_t s = cell_list[i];
                tab.hkey[s] = 0ull; tab.hcount[s] = 0u; tab.hmin[s] = 0xFFFFFFFFu;
            }
        }
    }
    if (blockIdx.x == 0) {
        // a frame keyed against the host's box: every kernel that forms voxel keys is through, PCL's own min_b / div_b go into DevState
        if (st->sbox && !st->err && !st->passthrough) { box_true_grid(const_cast<Frame&>(F), s_p); __syncthreads(); }
        if (threadIdx.x == 0) *out.state = *st;
    }
    if (st->err) return;
    if (st->kind != kFrameVoxelOnly && !st->trivial) {
        const uint32_t k = st->n_kept, np = st->n_kept_pts;
        for (uint32_t i = tid; i <= k; i += nth) out.offsets[i] = d_offsets[i];
#if PCOP_DIAG_NO_INDEX_COPY
        // diagnostic build (never the product; results are WRONG on the host: run without PARITY): the cluster indices stay on the device - what the zero-copy
        // stores over the bus cost the bench line, i.e. the most any other way of getting them to the host could win
#elif PCOP_EMIT_VEC
        {   // tuning build: four indices a lane (1 KB a wave instruction on the bus instead of 256 bytes); both arrays are allocations of their own, 16-byte aligned
            const uint32_t np4 = np / 4u;
            const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(d_indices);
            uint4* __restrict__ d4 = reinterpret_cast<uint4*>(out.indices);
            for (uint32_t i = tid; i < np4; i += nth) d4[i] = s4[i];
            for (uint32_t i = np4 * 4u + tid; i < np; i += nth) out.indices[i] = d_indices[i];
        }
#else
        for (uint32_t i = tid; i < np; i += nth) out.indices[i] = d_indices[i];
#endif
    }
    if (st->emit_cloud) {   // the cloud the cluster indices refer to: the voxel centroids, or the input planes (pass-through; the cloud a filter kept)
        const bool pt = st->c_from_input != 0 && st->kind != kFrameClusterOnly;
        const float* __restrict__ px = pt ? st->src_x : cen_x;
        const float* __restrict__ py = pt ? st->src_y : cen_y;
        const float* __restrict__ pz = pt ? st->src_z : cen_z;
        const uint32_t m = pt ? st->c_n : st->m, layout = st->out_layout, stride = st->out_soa_stride;
        for (uint32_t i = tid; i < m; i += nth) {
            const float x = px[i], y = py[i], z = pz[i];
            if (layout == 0) {
                reinterpret_cast<float4*>(out.cloud)[i] = make_float4(x, y, z, 1.0f);
            } else if (layout == 1) {
                out.cloud[i] = x; out.cloud[stride + i] = y; out.cloud[2 * (size_t)stride + i] = z;
            } else {
                out.cloud[3 * (size_t)i] = x; out.cloud[3 * (size_t)i + 1] = y; out.cloud[3 * (size_t)i + 2] = z;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The `euc_clusters` cloud: the points of the kept clusters in the host's (PCL) order, as pcl::PointXYZ records.
// order_idx: the ordered point indices (pinned host memory, read over the bus); out: pinned host memory.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cluster_cloud(const Frame* frames, const int32_t* __restrict__ order_idx, uint32_t count,
                                                          float4* __restrict__ out) {
    const Frame& F = frames[0];
    const DevState* st = &F.st;
    const float* __restrict__ cen_x = F.cen_x; const float* __restrict__ cen_y = F.cen_y; const float* __restrict__ cen_z = F.cen_z;
    const bool sel = st->c_from_input != 0;
    const float* __restrict__ x = sel ? st->src_x : cen_x;
    const float* __restrict__ y = sel ? st->src_y : cen_y;
    const float* __restrict__ z = sel ? st->src_z : cen_z;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < count; j += gridDim.x * blockDim.x) {
        const uint32_t p = (uint32_t)order_idx[j];
        out[j] = make_float4(x[p], y[p], z[p], 1.0f);
    }
}

// ------------------------------------------------------------------------------------------
// PointWithRad per cluster (msg/PointWithRad.msg): one wave per cluster.
// centre = double mean rounded to float; r = max calculate_distance() (od.cpp:457-464).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cluster_summaries(const Frame* frames, uint32_t npass_csr) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float* __restrict__ cen_x = F.cen_x; const float* __restrict__ cen_y = F.cen_y; const float* __restrict__ cen_z = F.cen_z;
    const uint32_t* __restrict__ offsets = F.d_offsets;
    const uint32_t* __restrict__ indices = ((npass_csr - 1) & 1) ? F.idx_b : F.idx_a;
    float4* __restrict__ xyzr = F.d_xyzr;
    if (st->err) return;
    const bool sel = st->c_from_input != 0;
    const float* __restrict__ x = sel ? st->src_x : cen_x;
    const float* __restrict__ y = sel ? st->src_y : cen_y;
    const float* __restrict__ z = sel ? st->src_z : cen_z;
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

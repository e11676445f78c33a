// pcop_cells_hash.hpp - cells in a hash table (fallback), union-find, the float32 L2_Simple predicate, hash-path hook / compress / union.
// Part of pcop_device.hpp (included from there, in this order; not a stand-alone header).
#pragma once

namespace pcop {

// ------------------------------------------------------------------------------------------
// cell grid: hash table of occupied cells.  Cell edge g = 1/inv_cell = |tol|/1.99 and the cell
// of a point is floor((double)p * inv_cell) per axis (the product of two floats widened to
// double is exact up to 2^-53), which gives two guarantees with margin:
//   same cell            =>  d2 < r2         (3 g^2 = 0.757 tol^2)
//   d2 < r2              =>  cell offsets <= 2 per axis   (tol * inv_cell = 1.99 < 2)
// Coordinates are biased by 2^20 and packed 21 bits per axis into a 64-bit key.
// slot arrays: hkey (key+1, 0 = empty), hcount (points), hmin (smallest point index = priority)
// ------------------------------------------------------------------------------------------
struct CellTable {
    unsigned long long* hkey;
    uint32_t* hcount;
    uint32_t* hmin;
};

// The table is 64 interleaved sub-tables: a cell may only sit in slots whose low 6 bits equal its
// index inside its 4x4x4 block of cells, and the bucket (slot >> 6) is hashed from the block's
// key.  Cells of one block therefore share a 64-slot bucket, so the slots, parents, metadata and
// point records of spatial neighbours share cache lines.  Probing steps from bucket to bucket.
__device__ __forceinline__ uint32_t cell_home(unsigned long long key, uint32_t mask) {
    const unsigned long long bx = (key & 0x1FFFFFull) >> 2, by = ((key >> 21) & 0x1FFFFFull) >> 2, bz = (key >> 42) >> 2;
    unsigned long long h = (bx | (by << 19) | (bz << 38)) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    const uint32_t idx = (uint32_t)(key & 3ull) | ((uint32_t)((key >> 21) & 3ull) << 2) | ((uint32_t)((key >> 42) & 3ull) << 4);
    return (((uint32_t)h << 6) | idx) & mask;
}

__device__ __forceinline__ unsigned long long pack_cell(long long cx, long long cy, long long cz) {
    return (unsigned long long)(cx + kCellBias) | ((unsigned long long)(cy + kCellBias) << 21) |
           ((unsigned long long)(cz + kCellBias) << 42);
}

// Probe number p of the cell whose home slot is `home`: the first table-size / 64 probes step from bucket to
// bucket inside the cell's own sub-table (low 6 slot bits = its index in the 4x4x4 block); should that sub-table be
// full - a cloud whose cells all share a few in-block indices, e.g. a sparse axis-aligned plane - the walk goes on
// through the next sub-table, and so on, so that it visits every slot of the table exactly once.  The table has
// at least twice as many slots as cells, so an insertion always finds a free slot and a lookup always ends.
__device__ __forceinline__ uint32_t probe_slot(uint32_t home, uint32_t p, uint32_t mask) {
    const uint32_t nb = (mask + 1u) >> 6;                 // buckets (a power of two, >= 16)
    const uint32_t b = ((home >> 6) + p) & (nb - 1u);     // bucket: p mod nb steps from the home bucket
    const uint32_t j = p >> (31 - __clz((int)nb));        // p / nb: sub-tables gone through so far
    return (b << 6) | (((home & 63u) + j) & 63u);
}

__device__ __forceinline__ uint32_t hash_find_or_insert(unsigned long long* hkey, uint32_t mask, unsigned long long key) {
    const unsigned long long want = key + 1ull;
    const uint32_t home = cell_home(key, mask);
    uint32_t slot = home, p = 0;
    while (true) {
        unsigned long long k = hkey[slot];
        if (k == 0ull) k = atomicCAS(hkey + slot, 0ull, want);
        if (k == 0ull || k == want) return slot;
        slot = probe_slot(home, ++p, mask);
    }
}

__device__ __forceinline__ uint32_t hash_lookup(const unsigned long long* __restrict__ hkey, uint32_t mask,
                                                unsigned long long key) {
    const unsigned long long want = key + 1ull;
    const uint32_t home = cell_home(key, mask);
    uint32_t slot = home, p = 0;
    while (true) {
        const unsigned long long k = hkey[slot];
        if (k == want) return slot;
        if (k == 0ull) return kNoSlot;
        slot = probe_slot(home, ++p, mask);
    }
}


// Every point: its cell's slot and its rank inside the cell.  Runs of consecutive points that
// fall in one cell (frequent: the points arrive in voxel order) share one set of atomics.
__global__ __launch_bounds__(kBlock) void k_cell_insert(Frame* frames) {
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const float* __restrict__ cen_x = F.cen_x; const float* __restrict__ cen_y = F.cen_y; const float* __restrict__ cen_z = F.cen_z;
    const CellTable tab{F.hkey, F.hcount, F.hmin};
    uint32_t* __restrict__ pt_slot = F.pt_slot; uint32_t* __restrict__ pt_rank = F.pt_rank;
    if (cluster_stage_off(st)) return;
    const bool from_in = st->c_from_input != 0;
    const float* __restrict__ x = from_in ? st->src_x : cen_x;
    const float* __restrict__ y = from_in ? st->src_y : cen_y;
    const float* __restrict__ z = from_in ? st->src_z : cen_z;
    const uint32_t n = st->c_n;
    const uint32_t mask = st->hash_mask;
    const double ig = st->inv_cell;
    const uint32_t lane = lane_id();
    const uint32_t nthreads = gridDim.x * blockDim.x;
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n; base += nthreads) {
        const uint32_t i = base + lane;
        bool valid = i < n;
        unsigned long long key = 0;
        if (valid) {
            const float px = x[i], py = y[i], pz = z[i];
            if (!finite3(px, py, pz)) {   // radiusSearch on a non-finite point is undefined in PCL
                atomicOr(&st->err, (uint32_t)kErrNonFinite);
                valid = false;
            } else {
                const double fx = floor((double)px * ig), fy = floor((double)py * ig), fz = floor((double)pz * ig);
                const double lim = (double)(kCellBias - 4);
                if (!(fabs(fx) < lim && fabs(fy) < lim && fabs(fz) < lim)) {
                    atomicOr(&st->err, (uint32_t)kErrRange);
                    valid = false;
                } else {
                    key = pack_cell((long long)fx, (long long)fy, (long long)fz);
                }
            }
        }
        const unsigned long long kprev = __shfl_up(key, 1, kWave);
        const uint32_t vprev = __shfl_up((uint32_t)valid, 1, kWave);
        const bool leader = valid && (lane == 0 || !vprev || key != kprev);
        const uint64_t leaders = __ballot(leader);
        const uint64_t breaks = __ballot(leader || !valid);
        uint32_t slot = 0, rank0 = 0;
        if (leader) {
            const uint64_t later = lane == 63 ? 0ull : (breaks >> (lane + 1));
            const uint32_t runlen = later ? (uint32_t)__ffsll((unsigned long long)later) : (uint32_t)(kWave - lane);
            slot = hash_find_or_insert(tab.hkey, mask, key);
            rank0 = atomicAdd(tab.hcount + slot, runlen);
            atomicMin(tab.hmin + slot, i);
        }
        // the run's leader is the highest leader at or below this lane
        const uint64_t le = leaders & (lanemask_lt() | (1ull << lane));
        const int ll = le ? 63 - __clzll((unsigned long long)le) : 0;
        const uint32_t s = __shfl(slot, ll, kWave);
        const uint32_t r = __shfl(rank0, ll, kWave);
        if (valid) { pt_slot[i] = s; pt_rank[i] = r + (lane - (uint32_t)ll); }
    }
}

// points and occupied slots per tile of kScanTile slots
__global__ __launch_bounds__(kBlock) void k_slot_count(const Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ hcount = F.hcount;
    uint32_t* __restrict__ tile_pts = F.tile_a; uint32_t* __restrict__ tile_occ = F.tile_b;
    if (cluster_stage_off(st)) return;
    const uint32_t cap = st->hash_mask + 1u;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= cap) return;
    uint32_t c = 0, o = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {   // cap is a multiple of 1024
        const uint32_t v = hcount[base + threadIdx.x * kScanItems + k];
        c += v; o += v ? 1u : 0u;
    }
    const uint32_t t = block_sum(c, s_w);
    const uint32_t u = block_sum(o, s_w);
    if (threadIdx.x == 0) { tile_pts[blockIdx.x] = t; tile_occ[blockIdx.x] = u; }
}

// cell_start[slot] = number of points in slots before it; dense list of occupied slots;
// union-find init
__global__ __launch_bounds__(kBlock) void k_slot_apply(Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const uint32_t* __restrict__ hcount = F.hcount;
    const uint32_t* __restrict__ tile_pts = F.tile_a; const uint32_t* __restrict__ tile_occ = F.tile_b;
    uint32_t* __restrict__ cell_start = F.cell_start; uint32_t* __restrict__ cell_list = F.cell_list;
    uint32_t* __restrict__ parent = F.parent; uint32_t* __restrict__ comp_size = F.comp_size;
    uint32_t* __restrict__ comp_rank = F.comp_rank;
    uint4* __restrict__ slot_meta = F.slot_meta; uint4* __restrict__ cell_info = F.cell_info;
    const unsigned long long* __restrict__ hkey = F.hkey; const uint32_t* __restrict__ hmin = F.hmin;
    if (cluster_stage_off(st)) return;
    const uint32_t cap = st->hash_mask + 1u;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= cap) return;
    const uint32_t before = tiles_before(tile_pts, blockIdx.x, s_w);
    const uint32_t before_occ = tiles_before(tile_occ, blockIdx.x, s_w);
    uint32_t v[kScanItems];
    uint32_t c = 0, occ = 0;
    const uint32_t s0 = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = hcount[s0 + k]; c += v[k]; occ += v[k] ? 1u : 0u; }
    uint32_t tot, tot_occ;
    uint32_t run = before + block_excl_scan(c, s_w, &tot);
    uint32_t id = before_occ + block_excl_scan(occ, s_w, &tot_occ);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (v[k]) {   // only occupied slots are ever read
            cell_start[s0 + k] = run;
            parent[s0 + k] = s0 + k;
            comp_size[s0 + k] = 0u;
            comp_rank[s0 + k] = 0xFFFFFFFFu;
            const unsigned long long k1 = hkey[s0 + k];
            const uint32_t mi = hmin[s0 + k];
            slot_meta[s0 + k] = make_uint4(run, v[k], mi, 0u);
            cell_info[2 * (size_t)id] = make_uint4((uint32_t)k1, (uint32_t)(k1 >> 32), run, v[k]);
            cell_info[2 * (size_t)id + 1] = make_uint4(mi, s0 + k, 0u, 0u);
            cell_list[id++] = s0 + k;
        }
        run += v[k];
    }
    if (threadIdx.x == 0 && base + kScanTile >= cap) st->n_cells = before_occ + tot_occ;   // the last tile
}

// counting sort: record of point i goes to cell_start[slot] + rank
__global__ __launch_bounds__(kBlock) void k_cell_place(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float* __restrict__ cen_x = F.cen_x; const float* __restrict__ cen_y = F.cen_y; const float* __restrict__ cen_z = F.cen_z;
    const uint32_t* __restrict__ pt_slot = F.pt_slot; const uint32_t* __restrict__ pt_rank = F.pt_rank;
    const uint32_t* __restrict__ cell_start = F.cell_start;
    float4* __restrict__ rec = F.rec_a;
    if (cluster_stage_off(st)) return;
    const bool from_in = st->c_from_input != 0;
    const float* __restrict__ x = from_in ? st->src_x : cen_x;
    const float* __restrict__ y = from_in ? st->src_y : cen_y;
    const float* __restrict__ z = from_in ? st->src_z : cen_z;
    const uint32_t n = st->c_n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        rec[cell_start[pt_slot[i]] + pt_rank[i]] = make_float4(x[i], y[i], z[i], u2f(i));
}

// ------------------------------------------------------------------------------------------
// union-find over cells (slots).  The root of a set is its cell of smallest priority, and the
// priority of a cell is the smallest point index it holds, so the root of a finished component
// carries the component's discovery key (its first point in index order).
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

__device__ __forceinline__ void uf_union(uint32_t* parent, const uint32_t* __restrict__ prio, uint32_t a, uint32_t b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (prio ? prio[a] > prio[b] : a > b) { const uint32_t t = a; a = b; b = t; }   // (no priority array: the dense grid, where a cell's id is its priority)
        // hang the root of larger priority value b under a
        const uint32_t old = atomicCAS(parent + b, b, a);
        if (old == b) return;
    }
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
// pair satisfies d2 < r2".  One wave per cell; lane t owns offset t of one half of the 5x5x5
// stencil (62 offsets), looked up through the hash table.
//   k_cell_hook    : backward half, no atomics: parent[c] = adjacent cell of smallest priority
//   k_cell_compress: pointer jumping, parent[c] = root
//   k_cell_union   : forward half: pairs whose roots already agree are skipped without a
//                    distance test; the rest are tested and united by CAS (few survive the hook)
// ------------------------------------------------------------------------------------------
constexpr int kCellPtsCap = 64;
constexpr int kStencil = 62;

// forward offset number t (0..61): (0,0,+1..2) ; (0,+1..2,*) ; (+1..2,*,*)
__device__ __forceinline__ void stencil_offset(uint32_t t, int* dx, int* dy, int* dz) {
    if (t < 2) { *dz = 0; *dy = 0; *dx = (int)t + 1; return; }
    const uint32_t u = t - 2;
    if (u < 10) { *dz = 0; *dy = 1 + (int)(u / 5); *dx = (int)(u % 5) - 2; return; }
    const uint32_t v = u - 10;
    *dz = 1 + (int)(v / 25); *dy = (int)((v % 25) / 5) - 2; *dx = (int)(v % 5) - 2;
}

// slot of the cell at this lane's stencil offset from the cell with packed key `ka`
__device__ __forceinline__ uint32_t stencil_neighbour(const unsigned long long* __restrict__ hkey, uint32_t mask,
                                                      unsigned long long ka, uint32_t lane, int sign) {
    if (lane >= (uint32_t)kStencil) return kNoSlot;
    int dx, dy, dz;
    stencil_offset(lane, &dx, &dy, &dz);
    // biased coordinates stay inside 21 bits: k_cell_insert keeps |c| < 2^20 - 4
    const long long off = (long long)(sign * dx) + ((long long)(sign * dy) << 21) + ((long long)(sign * dz) << 42);
    return hash_lookup(hkey, mask, (unsigned long long)((long long)ka + off));
}

// does some point of B (this lane's cell) lie within the radius of some point of A (in LDS)?
__device__ __forceinline__ bool lane_pair_hit(const float4* s_pa, uint32_t an, const float4* __restrict__ rec,
                                              uint32_t b0, uint32_t bn, float r2) {
    for (uint32_t ib = 0; ib < bn; ib += 4) {   // four independent loads in flight per step
        float4 pb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pb[q] = rec[b0 + min(ib + q, bn - 1)];
        bool hit = false;
        for (uint32_t ia = 0; ia < an; ++ia) {
            const float4 pa = s_pa[ia];
#pragma unroll
            for (int q = 0; q < 4; ++q) hit |= l2_simple(pa, pb[q]) < r2;   // a repeated last point changes nothing
        }
        if (hit) return true;
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

// Hooking only needs SOME adjacent cell of smaller priority, so it looks at the 13 backward cells
// of the 3x3x3 neighbourhood only: four cells per wave, 16 lanes each (13 used).
constexpr int kHookGroup = 16;
constexpr int kHookStencil = 13;
constexpr int kHookPtsCap = 32;

// backward offset t (0..12) of the 3x3x3 stencil: (0,0,-1) ; (0,-1,*) ; (-1,*,*)
__device__ __forceinline__ long long hook_offset(uint32_t t) {
    int dx, dy, dz;
    if (t == 0) { dx = -1; dy = 0; dz = 0; }
    else if (t < 4) { dx = (int)t - 2; dy = -1; dz = 0; }
    else { const uint32_t v = t - 4; dx = (int)(v % 3) - 1; dy = (int)(v / 3) - 1; dz = -1; }
    return (long long)dx + ((long long)dy << 21) + ((long long)dz << 42);
}

__global__ __launch_bounds__(kBlock, 8) void k_cell_hook(const Frame* frames) {
    __shared__ float4 s_pa[kBlock / kHookGroup][kHookPtsCap];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const unsigned long long* __restrict__ hkey = F.hkey;
    const uint4* __restrict__ slot_meta = F.slot_meta; const uint4* __restrict__ cell_info = F.cell_info;
    uint32_t* __restrict__ parent = F.parent;
    if (cluster_stage_off(st)) return;
    const uint32_t mask = st->hash_mask, ncell = st->n_cells;
    const float r2 = st->r2;
    const uint32_t g = threadIdx.x / kHookGroup;          // group in block
    const uint32_t gl = threadIdx.x % kHookGroup;         // lane in group
    const uint32_t ngroups = gridDim.x * (kBlock / kHookGroup);
    // trip count is uniform per wave: the four groups of a wave stride together
    const uint32_t wave_first = (blockIdx.x * (kBlock / kHookGroup) + (threadIdx.x / kWave) * (kWave / kHookGroup));
    for (uint32_t base = wave_first; base < ncell; base += ngroups) {
        const uint32_t ai = base + (g % (kWave / kHookGroup));
        const bool live = ai < ncell;
        uint4 i0 = make_uint4(1u, 0u, 0u, 0u), i1 = make_uint4(0u, 0u, 0u, 0u);
        if (live) { i0 = cell_info[2 * (size_t)ai]; i1 = cell_info[2 * (size_t)ai + 1]; }
        const unsigned long long ka = (((unsigned long long)i0.y << 32) | i0.x) - 1ull;
        const uint32_t a0 = i0.z, an = i0.w, pa = i1.x, a = i1.y;
        uint32_t b = kNoSlot;
        if (live && gl < (uint32_t)kHookStencil) b = hash_lookup(hkey, mask, (unsigned long long)((long long)ka + hook_offset(gl)));
        uint4 mb = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
        if (b != kNoSlot) mb = slot_meta[b];
        const bool cand = b != kNoSlot && mb.z < pa;     // only cells of smaller priority can become the parent
        const uint32_t b0 = mb.x, bn = cand ? mb.y : 0u, pb = mb.z;
        // stage A's points (per group) when they fit
        const bool small = an <= (uint32_t)kHookPtsCap;
        if (live && small) for (uint32_t t = gl; t < an; t += kHookGroup) s_pa[g][t] = rec[a0 + t];
        __builtin_amdgcn_wave_barrier();
        bool hit = false;
        if (cand) {
            if (small) {
                hit = lane_pair_hit(s_pa[g], an, rec, b0, bn, r2);
            } else {   // a big cell: this lane walks A from memory
                for (uint32_t ia = 0; ia < an && !hit; ++ia) {
                    const float4 pa4 = rec[a0 + ia];
                    for (uint32_t ib = 0; ib < bn; ++ib)
                        if (l2_simple(pa4, rec[b0 + ib]) < r2) { hit = true; break; }
                }
            }
        }
        // smallest priority among the group's hits
        unsigned long long best = hit ? (((unsigned long long)pb << 32) | b) : ~0ull;
#pragma unroll
        for (int d = kHookGroup / 2; d > 0; d >>= 1) {
            const unsigned long long o = __shfl_xor(best, d, kWave);
            best = o < best ? o : best;
        }
        if (live && gl == 0 && best != ~0ull) parent[a] = (uint32_t)best;   // strictly decreasing priority: a forest
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(kBlock) void k_cell_compress(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ cell_list = F.cell_list;
    uint32_t* __restrict__ parent = F.parent;
    if (cluster_stage_off(st)) return;
    const uint32_t ncell = st->n_cells;
    const bool dense = st->cpath != 0;   // dense cell grid: cell ids are the slots
    for (uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x; ci < ncell; ci += gridDim.x * blockDim.x) {
        const uint32_t c = dense ? ci : cell_list[ci];
        // pointer jumping on the cell's own word: every cell does this at the same time, so the distance a
        // pointer spans doubles per round and a chain of depth d is flat after about log2(d) rounds
        uint32_t p = ld_agent(parent + c);
        while (true) {
            const uint32_t gp = ld_agent(parent + p);
            if (gp == p) break;
            st_agent(parent + c, gp);
            p = gp;
        }
    }
}

// The slow part of the union for one cell: lanes whose forward neighbour is in another set run the
// pair test; the wave then unites ONE hit at a time and re-reads the roots, because in practice all
// hits of a wave lead to the same one or two foreign trees and a CAS per lane would only hammer
// the same root word.
__device__ __forceinline__ void union_fixup(const float4* __restrict__ rec, const uint4* __restrict__ slot_meta,
                                            const uint32_t* __restrict__ hmin, uint32_t* parent, float4* s_pa,
                                            uint32_t a, uint32_t a0, uint32_t an, uint32_t b, bool need, float r2) {
    const uint32_t lane = lane_id();
    uint64_t need_mask = __ballot(need);
    uint32_t b0 = 0, bn = 0;
    if (need) { const uint4 mb = slot_meta[b]; b0 = mb.x; bn = mb.y; }
    const uint32_t bn_max = wave_max(bn);
    if (an <= (uint32_t)kCellPtsCap && bn_max <= (uint32_t)kCellPtsCap) {
        if (lane < an) s_pa[lane] = rec[a0 + lane];
        __builtin_amdgcn_wave_barrier();
        const bool hit = need && lane_pair_hit(s_pa, an, rec, b0, bn, r2);
        uint64_t pending = __ballot(hit);
        uint32_t ra = 0;
        while (pending) {
            const int t = __ffsll((unsigned long long)pending) - 1;
            const uint32_t cb = __shfl(b, t, kWave);
            if (lane == 0) { uf_union(parent, hmin, a, cb); ra = uf_find(parent, a); }
            ra = __shfl(ra, 0, kWave);
            const bool still = ((pending >> lane) & 1ull) && (int)lane != t && uf_find(parent, b) != ra;
            pending = __ballot(still);
        }
        __builtin_amdgcn_wave_barrier();
    } else {
        while (need_mask) {
            const int t = __ffsll((unsigned long long)need_mask) - 1;
            need_mask &= need_mask - 1;
            const uint32_t cb = __shfl(b, t, kWave), cb0 = __shfl(b0, t, kWave), cbn = __shfl(bn, t, kWave);
            uint32_t rra = 0, rrb = 1;
            if (lane == 0) { rra = uf_find(parent, a); rrb = uf_find(parent, cb); }
            rra = __shfl(rra, 0, kWave); rrb = __shfl(rrb, 0, kWave);
            if (rra == rrb) continue;
            if (wave_pair_hit(rec, a0, an, cb0, cbn, r2) && lane == 0) uf_union(parent, hmin, a, cb);
        }
    }
}

// One wave per cell, lane = one of the 62 forward offsets of the 5x5x5 stencil.  After hook +
// compress almost every neighbour already points at the cell's root, so the common case is a chain
// of dependent loads (cell info -> hash probe -> parent word) and no arithmetic.  The loop is
// software-pipelined over that chain: two cells per step, and while the parent words of step k are
// in flight the hash probes of step k+1 and the cell info of step k+2 are already issued.
// parent[b] == parent[a] proves "same set" without knowing the root (a parent is always a member of
// the same set), so the fast path never walks the tree.
struct UnionCells {          // the two cells of one step
    uint4 i0, i1, j0, j1;    // cell_info of A and of C
};
__device__ __forceinline__ UnionCells union_load_cells(const uint4* __restrict__ cell_info, uint32_t ai, uint32_t ncell) {
    UnionCells u;
    const uint32_t a = min(ai, ncell - 1u), c = min(ai + 1u, ncell - 1u);   // clamped: never read past the list
    u.i0 = cell_info[2 * (size_t)a]; u.i1 = cell_info[2 * (size_t)a + 1];
    u.j0 = cell_info[2 * (size_t)c]; u.j1 = cell_info[2 * (size_t)c + 1];
    return u;
}

__global__ __launch_bounds__(kBlock) void k_cell_union(const Frame* frames) {
    __shared__ float4 s_pa[kWavesPerBlock][kCellPtsCap];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const unsigned long long* __restrict__ hkey = F.hkey;
    const uint32_t* __restrict__ hmin = F.hmin;
    const uint4* __restrict__ slot_meta = F.slot_meta; const uint4* __restrict__ cell_info = F.cell_info;
    uint32_t* __restrict__ parent = F.parent;
    if (cluster_stage_off(st) || st->debug_skip_union == 1u) return;
    const bool scan_only = st->debug_skip_union == 2u;   // diagnostics: time the scan without the fix-ups
    const uint32_t mask = st->hash_mask, ncell = st->n_cells;
    const float r2 = st->r2;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));   // wave-uniform: cell info goes through scalar loads
    const uint32_t lane = lane_id();
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const bool in = lane < (uint32_t)kStencil;
    long long off = 0;
    if (in) {
        int dx, dy, dz;
        stencil_offset(lane, &dx, &dy, &dz);
        off = (long long)dx + ((long long)dy << 21) + ((long long)dz << 42);
    }
    uint32_t ai = 2 * (blockIdx.x * kWavesPerBlock + w);
    if (ai >= ncell) return;
    const uint32_t step = 2 * nwaves;
    // prologue: info(0) -> probes(0) issued, info(1) issued
    UnionCells cur = union_load_cells(cell_info, ai, ncell);
    UnionCells nxt = union_load_cells(cell_info, ai + step, ncell);
    unsigned long long wa = (unsigned long long)((long long)((((unsigned long long)cur.i0.y << 32) | cur.i0.x) - 1ull) + off) + 1ull;
    unsigned long long wc = (unsigned long long)((long long)((((unsigned long long)cur.j0.y << 32) | cur.j0.x) - 1ull) + off) + 1ull;
    uint32_t sa = in ? cell_home(wa - 1ull, mask) : 0u, sc = in ? cell_home(wc - 1ull, mask) : 0u;
    unsigned long long va = in ? hkey[sa] : 0ull, vc = in ? hkey[sc] : 0ull;
    uint32_t pa = ld_agent(parent + cur.i1.y), pc = ld_agent(parent + cur.j1.y);
    for (; ai < ncell; ai += step) {
        const bool two = ai + 1 < ncell;
        const uint32_t a = cur.i1.y, c = cur.j1.y;
        // finish the probes of this step, then ask for the neighbours' parent words
        { const uint32_t home = sa; uint32_t p = 0; while (va != wa && va != 0ull) { sa = probe_slot(home, ++p, mask); va = hkey[sa]; } }
        { const uint32_t home = sc; uint32_t p = 0; while (vc != wc && vc != 0ull) { sc = probe_slot(home, ++p, mask); vc = hkey[sc]; } }
        const uint32_t ba = va == wa ? sa : kNoSlot, bc = (two && vc == wc) ? sc : kNoSlot;
        uint32_t pba = pa, pbc = pc;
        if (ba != kNoSlot) pba = ld_agent(parent + ba);
        if (bc != kNoSlot) pbc = ld_agent(parent + bc);
        // meanwhile: probes of the next step, cell info of the one after
        const UnionCells here = cur;
        const uint32_t pa_here = pa, pc_here = pc;
        const bool more = ai + step < ncell;
        cur = nxt;
        if (more) {
            wa = (unsigned long long)((long long)((((unsigned long long)cur.i0.y << 32) | cur.i0.x) - 1ull) + off) + 1ull;
            wc = (unsigned long long)((long long)((((unsigned long long)cur.j0.y << 32) | cur.j0.x) - 1ull) + off) + 1ull;
            sa = in ? cell_home(wa - 1ull, mask) : 0u; sc = in ? cell_home(wc - 1ull, mask) : 0u;
            va = in ? hkey[sa] : 0ull; vc = in ? hkey[sc] : 0ull;
            pa = ld_agent(parent + cur.i1.y); pc = ld_agent(parent + cur.j1.y);
            nxt = union_load_cells(cell_info, ai + 2 * step, ncell);
        }
        // neighbours whose parent word differs from the cell's may be in another set: look properly
        bool need_a = pba != pa_here && !scan_only, need_c = pbc != pc_here && !scan_only;
        if (__any(need_a)) {
            const uint32_t ra = uf_find(parent, a);
            need_a = need_a && uf_find(parent, ba) != ra;
            if (__any(need_a)) union_fixup(rec, slot_meta, hmin, parent, s_pa[w], a, here.i0.z, here.i0.w, ba, need_a, r2);
        }
        if (__any(need_c)) {
            const uint32_t rc = uf_find(parent, c);
            need_c = need_c && uf_find(parent, bc) != rc;
            if (__any(need_c)) union_fixup(rec, slot_meta, hmin, parent, s_pa[w], c, here.j0.z, here.j0.w, bc, need_c, r2);
        }
    }
}

}  // namespace pcop

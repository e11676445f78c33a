// pcop_device.hpp — gfx950 device code of the obstacle hot path.
//
// All HBM/LDS index-and-scatter work (no MFMA).  One frame is a fixed sequence of launches:
//
//   voxel grid (replaces pcl::VoxelGrid::applyFilter, called at od.cpp:282-285)
//     k_frame_init            mailbox -> DevState
//     k_minmax, k_voxel_setup pcl::getMinMax3D; min_b/div_b/divb_mul, the overflow rule, the plan
//    streaming path (default):
//     k_vox_mark, k_occ_scan<0> occupied grid cells -> voxel id = rank of the key, M
//     k_vid_hist / k_table_scan / k_vid_scatter   one stable 8-bit partition of {x,y,z,id} by bucket
//     k_voxel_stream          per bucket: in-order float32 sums in LDS accumulators -> centroids
//    sort path (fallback for grids or voxel counts the streaming path does not cover):
//     k_radix_hist / k_table_scan / k_radix_scatter   (x passes) stable LSD radix partition of
//                             packed xyz records by PCL's linear voxel index
//     k_voxel_reduce          run heads -> rank; in-order float32 sums -> centroids
//   cell grid (replaces the FLANN kd-tree built at od.cpp:791-792): cells of edge tol/1.99
//    dense grid (default): k_cgrid_setup, k_cell_mark, k_occ_scan<1>, k_cell_count, k_cell_starts,
//                          k_cell_place   occupancy bits -> cell ids; points sorted by cell
//    hash table (fallback): k_cell_insert, k_slot_count, k_slot_apply, k_cell_place
//   clusters (replaces EuclideanClusterExtraction::extract, od.cpp:446-454)
//     k_cgrid_hook | k_cell_hook, k_cell_compress, k_cgrid_union | k_cell_union
//                             connected components of the cell graph
//     k_cell_flatten          root per point, component sizes
//     k_kept_apply            size filter, discovery order, CSR offsets
//     k_csr_hist / k_table_scan / k_csr_scatter   stable partition of point indices by cluster
//     k_emit                  results -> pinned host memory; hash table reset for the next frame
//
// Every count that depends on the data (valid points, voxels M, cells, clusters) lives in
// DevState in device memory and every per-frame input comes through a pinned mailbox, so the
// kernel arguments of a frame are constant and the whole frame replays as one hipGraph.
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
constexpr int kMaxBits = 11;                     // widest radix digit any kernel variant supports
constexpr int kMaxBins = (1 << kMaxBits) + 1;    // + 1 trash bin (non-finite points)
constexpr int kNarrowBits = 8;                   // digit limit of the small-LDS kernel variants
#ifndef PCOP_ITEMS
#define PCOP_ITEMS 8
#endif
constexpr int kItems = PCOP_ITEMS;               // records per thread in the partition kernels
constexpr int kTile = kBlock * kItems;           // records per partition block
constexpr int kMaxPasses = 4;                    // 4 x 8 bits = 3 x 11 bits >= 32-bit keys
constexpr int kScanItems = 4;                    // elements per thread in flag-scan kernels
constexpr int kScanTile = kBlock * kScanItems;
constexpr int kReduceBlocks = 256;               // grid of the min/max kernel: one block per CU
constexpr int kCellBias = 1 << 20;               // cell coordinates are stored biased, 21 bits per axis
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kVmapCells = 1u << 26;       // streaming voxel path: largest voxel grid (cells) with an occupancy map
constexpr uint32_t kVmapTile = 1024;            // occupancy words per block of k_occ_scan
constexpr uint32_t kStreamBuckets = 256;        // buckets of the streaming voxel path (one 8-bit partition pass)
constexpr uint32_t kVcapMin = 512, kVcapMax = 2048;   // voxels per bucket (LDS accumulators: 24 bytes per voxel)
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr uint32_t kCmapCells = 1u << 24;       // dense cell grid: largest grid (cells, margins included) with an occupancy map
constexpr int kCellMargin = 3;                  // empty cells around the grid: 2 for the 5x5x5 stencil + 1 of slack (a centroid
                                                // may round a hair outside the bounding box of its points)        // voxel key / id of a dropped (non-finite) point

// Pointers that live in device-visible structs (Frame, DevState, the mailbox).  A pointer loaded from
// memory is a generic ("flat") pointer to the compiler, and every access through it becomes a
// flat_load/flat_store, which counts against both the vector-memory and the LDS counter and forces
// full s_waitcnt drains.  DevPtr stores the same 8 bytes but, in the device pass, types them as
// address space 1, so the kernels get global_load/global_store.  The host pass sees a plain pointer.
template <class T>
struct DevPtr {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(1))) T* gptr;
    gptr p;
    __host__ __device__ DevPtr& operator=(T* q) { p = (gptr)q; return *this; }
    __host__ __device__ operator gptr() const { return p; }
    __host__ __device__ T* get() const { return (T*)p; }
#else
    T* p;
    DevPtr& operator=(T* q) { p = q; return *this; }
    operator T*() const { return p; }
    T* get() const { return p; }
#endif
};
static_assert(sizeof(DevPtr<float>) == sizeof(float*), "DevPtr is layout-compatible with a pointer");

enum ErrBits : uint32_t {
    kErrNonFinite = 1u,    // NaN/Inf where the reference is undefined
    kErrReplanVoxel = 2u,  // voxel key needs more radix passes than were launched
    kErrReplanCsr = 8u,    // cluster-rank key needs more radix passes than were launched
    kErrRange = 16u,       // a cell coordinate leaves +-2^20 cells (|p| / tol too large)
    kErrInternal = 32u,    // a bounded wait inside a kernel ran out (never expected)
    kErrReplanPath = 64u,  // the streaming voxel path does not apply to this frame (grid or voxel count too large)
    kErrReplanVcap = 128u, // the streaming voxel path needs more voxels per bucket than the launch provided LDS for
    kErrReplanCpath = 256u,  // the dense cell grid does not apply to this frame (grid too large)
    kErrReplanCtiles = 512u, // the occupancy scan of the dense cell grid was launched with too few tiles
};

enum FrameKind : uint32_t { kFrameChain = 0, kFrameVoxelOnly = 1, kFrameClusterOnly = 2 };

struct RadixPlan {
    uint32_t key_bits;
    uint32_t npass;
    uint32_t bits[kMaxPasses];
    uint32_t shift[kMaxPasses];
};

// Per-frame inputs, written by the host into pinned memory before the graph is launched.
struct FrameArgs {
    uint32_t n, is_dense, kind, trivial;      // trivial: r2 == 0, clustering skipped (singletons)
    float inv_leaf, r2;
    double inv_cell;                          // 1.99 / |tol|
    uint32_t min_cluster, max_cluster;
    uint32_t emit_cloud, out_layout, out_soa_stride, debug_skip_union;
    uint32_t voxel_stream, vcap;              // streaming voxel path requested; voxels per bucket the launch has LDS for
    uint32_t cell_grid, pad0;                 // dense cell grid requested (else: the hash table of cells)
    DevPtr<const float> src_x, src_y, src_z;
};

// Device-resident frame state; k_emit copies it to pinned host memory at the end of a frame.
struct DevState {
    uint32_t err;
    uint32_t n_in;          // points handed in
    uint32_t is_dense;
    uint32_t n_valid;       // finite points (== n_in for dense clouds)
    uint32_t passthrough;   // VoxelGrid overflow rule fired
    uint32_t m;             // voxels (downsampled points)
    uint32_t kind, trivial, emit_cloud, out_layout, out_soa_stride;
    uint32_t reserved0;
    // voxel grid (pcl::VoxelGrid members)
    float    inv_leaf;
    float    vmin[3], vmax[3];
    int32_t  min_b[3], div_b[3];
    uint32_t mul[3];
    RadixPlan plan_voxel;
    DevPtr<const float> src_x, src_y, src_z;
    // cluster stage
    double   inv_cell;
    uint32_t c_n;           // points entering clustering
    uint32_t c_from_input;  // 1: cluster the input cloud (passthrough / stage API), 0: the centroids
    float    r2;            // (float)((double)tol*(double)tol)
    uint32_t min_cluster, max_cluster;
    uint32_t hash_mask;     // cell hash table has hash_mask+1 slots (power of two >= 2*c_n, >= 1024)
    uint32_t n_cells;
    uint32_t n_kept;        // clusters with min <= size <= max
    uint32_t n_kept_pts;    // sum of their sizes
    RadixPlan plan_csr;
    uint32_t debug_skip_union;   // diagnostics: leave the hook forest untouched
    // streaming voxel path
    uint32_t vpath;         // 1: this frame's voxel stage runs the streaming path
    uint32_t vmap_words;    // occupancy words in use = ceil(cells / 32)
    uint32_t vshift;        // bucket of a voxel = id >> vshift
    uint32_t nbuckets;
    uint32_t vcap;          // voxels per bucket the launch has LDS for
    // dense cell grid
    uint32_t cgrid_req;     // requested by the host
    uint32_t cpath;         // 1: this frame's cell stage runs on the dense grid
    uint32_t cmap_words;    // occupancy words in use
    uint32_t c_rowbits;     // bits per x-row (multiple of 32), margins included
    uint32_t c_plane;       // bits per z-plane
    int32_t  c0[3];         // cell coordinate of grid position 0 (margin included)
    uint32_t c_dim[3];      // grid size per axis, margins included
    uint32_t pad1;
};
static_assert(sizeof(DevState) % 8 == 0, "DevState layout");

struct MinMaxPartial {
    float mn[3], mx[3];
    uint32_t valid, nonfinite;
};

// Everything one frame of a batch owns.  A launch covers a whole batch: blockIdx.y selects the
// frame, and every kernel takes its buffers from frames[blockIdx.y], so the kernel arguments
// are the same for every launch and the batch replays as one hipGraph.
struct Frame {
    DevState st;
    DevPtr<const FrameArgs> args;     // pinned host mailbox
    DevPtr<float> cen_x, cen_y, cen_z;   // voxel centroids (SoA)
    DevPtr<float4> rec_a, rec_b;      // partition ping-pong / cell-sorted records
    DevPtr<uint32_t> table, table_x, bin_total, tile_a, tile_b;
    DevPtr<MinMaxPartial> partials;
    DevPtr<unsigned long long> hkey;  // cell hash table: key+1 (0 = empty)
    DevPtr<uint32_t> hcount, hmin;    //                  points, smallest point index (= priority)
    DevPtr<uint32_t> cell_start, cell_list, parent, comp_size, comp_rank;
    DevPtr<uint4> slot_meta;          // per occupied slot: {first record, points, smallest point index, 0}
    DevPtr<uint4> cell_info;          // per cell (dense), 2 x uint4: {key lo, key hi, first record, points}, {min idx, slot, 0, 0}
    DevPtr<uint32_t> pt_slot, pt_rank, pt_root;
    DevPtr<uint32_t> idx_a, idx_b, d_offsets;
    DevPtr<float4> d_xyzr;
    // streaming voxel path (null when the context was created without it)
    DevPtr<uint8_t> vflag;            // one byte per voxel-grid cell: occupied?  all zero between frames
    DevPtr<uint2> vmap;               // per 32 cells: {occupancy bits, occupied cells before this word}
    DevPtr<uint32_t> vkey;            // per input point: voxel key, then voxel id (rank among occupied cells)
    DevPtr<uint32_t> vtile;           // per occupancy tile: its popcount + 1 once known (0 = not yet)
    // dense cell grid (null when the context was created without it)
    DevPtr<uint8_t> cflag;            // one byte per grid cell: occupied?  all zero between frames
    DevPtr<uint2> cmap;               // per 32 cells: {occupancy bits, occupied cells before this word}
    DevPtr<uint32_t> ctile;           // scan tiles: popcount + 1 once known
    // pinned host outputs
    DevPtr<DevState> h_state;
    DevPtr<uint32_t> h_offsets;       // n_kept + 1, discovery order
    DevPtr<uint32_t> h_indices;       // n_kept_pts
    DevPtr<float> h_cloud;            // voxel cloud in st.out_layout
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
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, kWave);
        v = o < v ? o : v;
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

// Sum over the whole block, returned to every thread (two barriers).
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* s_w) {
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) s_w[threadIdx.x / kWave] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) t += s_w[i];
    return t;
}

// Sum of per-tile values of all tiles before tile b: each block re-derives its own prefix
// (b <= a few thousand L2-resident words) instead of paying a kernel boundary for a scan.
__device__ __forceinline__ uint32_t tiles_before(const uint32_t* __restrict__ tile_vals, uint32_t b, uint32_t* s_w) {
    uint32_t s = 0;
    for (uint32_t t = threadIdx.x; t < b; t += kBlock) s += tile_vals[t];
    return block_sum(s, s_w);
}

__device__ __forceinline__ uint32_t bits_for(uint64_t count) {   // bits to index `count` values (>= 1)
    uint32_t b = 1;
    while (b < 64 && (1ull << b) < count) ++b;
    return b;
}

// Spread key_bits over the `npass` radix passes the host launched.  Returns false when a
// digit would be wider than kMaxBits (the host then re-runs the frame with more passes).
__device__ __forceinline__ bool make_plan(RadixPlan* pl, uint32_t key_bits, uint32_t npass, uint32_t maxbits) {
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
    return each <= maxbits;
}

// ------------------------------------------------------------------------------------------
// record sources and the voxel key
// ------------------------------------------------------------------------------------------
// Voxel-stage records are packed 12-byte xyz triples: the partition is stable, so the point index
// never has to travel.  Pass 0 reads the frame's input planes, later passes the records the previous
// pass wrote; pass p writes rec_a when p is even, rec_b when odd.
struct Rec3 {
    float x, y, z;
};
struct RecSrc {
    const float* x; const float* y; const float* z;
    const Rec3* r;   // null on pass 0
    __device__ __forceinline__ Rec3 load(uint32_t i) const {
        if (r) return r[i];
        Rec3 v; v.x = x[i]; v.y = y[i]; v.z = z[i];
        return v;
    }
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
    template <class P>
    __device__ __forceinline__ uint32_t operator()(const P& p) const {
        const int i0 = (int)(floorf(p.x * inv) - fb0);
        const int i1 = (int)(floorf(p.y * inv) - fb1);
        const int i2 = (int)(floorf(p.z * inv) - fb2);
        return (uint32_t)i0 + (uint32_t)i1 * mul1 + (uint32_t)i2 * mul2;
    }
};

// ------------------------------------------------------------------------------------------
// frame init + layout conversion
// ------------------------------------------------------------------------------------------
__global__ void k_frame_init(Frame* frames) {
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const FrameArgs a = *F.args;
        DevState s;
        memset(&s, 0, sizeof(s));
        s.n_in = a.n; s.is_dense = a.is_dense; s.kind = a.kind; s.trivial = a.trivial;
        s.emit_cloud = a.emit_cloud; s.out_layout = a.out_layout; s.out_soa_stride = a.out_soa_stride;
        s.inv_leaf = a.inv_leaf; s.r2 = a.r2; s.inv_cell = a.inv_cell;
        s.min_cluster = a.min_cluster; s.max_cluster = a.max_cluster;
        s.src_x = a.src_x; s.src_y = a.src_y; s.src_z = a.src_z;
        s.debug_skip_union = a.debug_skip_union;
        s.vcap = a.vcap; s.cgrid_req = a.cell_grid;
        if (a.kind == kFrameClusterOnly) {   // no voxel stage: cluster the input cloud as it is
            s.c_from_input = 1; s.c_n = a.n; s.n_valid = a.n;
            uint32_t cap = 1024;
            while (cap < 2u * a.n && cap < 0x80000000u) cap <<= 1;
            s.hash_mask = cap - 1u;
        }
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

// ------------------------------------------------------------------------------------------
// min / max (pcl::getMinMax3D) + grid set-up by the last block to arrive
// ------------------------------------------------------------------------------------------
// VoxelGrid::applyFilter up to "divb_mul_ = ..." (SURVEY.md §8a-1 steps 1-4).
__device__ __forceinline__ void finish_voxel(DevState* st, uint32_t m);

__device__ __forceinline__ void voxel_setup(DevState* st, const MinMaxPartial& p, uint32_t npass, uint32_t maxbits, bool stream) {
    if (st->is_dense && p.nonfinite) { atomicOr(&st->err, (uint32_t)kErrNonFinite); return; }
    st->n_valid = p.valid;
    for (int a = 0; a < 3; ++a) { st->vmin[a] = p.mn[a]; st->vmax[a] = p.mx[a]; }
    for (int a = 0; a < 3; ++a) { st->min_b[a] = 0; st->div_b[a] = 1; st->mul[a] = 1; }
    make_plan(&st->plan_voxel, 1, npass, maxbits);
    if (p.valid == 0) {   // empty cloud -> empty output
        if (stream) finish_voxel(st, 0);
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
        if (stream) finish_voxel(st, st->n_in);   // "output = *input_": the sort path does this in k_voxel_reduce
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
    const bool fits = make_plan(&st->plan_voxel, kb, npass, maxbits);
    if (stream) {
        if (cells > (uint64_t)kVmapCells) { atomicOr(&st->err, (uint32_t)kErrReplanPath); return; }
        st->vpath = 1u;
        st->vmap_words = (uint32_t)((cells + 31u) / 32u);
    } else if (!fits) {
        atomicOr(&st->err, (uint32_t)kErrReplanVoxel);
    }
}

__device__ __forceinline__ void reduce_partials_lds(MinMaxPartial* s_p) {   // kBlock entries -> s_p[0]
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
}

__global__ __launch_bounds__(kBlock) void k_minmax(Frame* frames) {
    __shared__ MinMaxPartial s_p[kBlock];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    MinMaxPartial* partials = F.partials;
    if (st->kind == kFrameClusterOnly && !st->cgrid_req) return;   // bounds of a cluster-only frame: for the dense cell grid
    const float* __restrict__ x = st->src_x;
    const float* __restrict__ y = st->src_y;
    const float* __restrict__ z = st->src_z;
    const uint32_t n = st->n_in;
    const float fmax = 3.402823466e+38f;
    float mn0 = fmax, mn1 = fmax, mn2 = fmax, mx0 = -fmax, mx1 = -fmax, mx2 = -fmax;
    uint32_t valid = 0, bad = 0;
    auto take = [&](float a, float b, float c) {
        if (finite3(a, b, c)) {
            ++valid;
            mn0 = a < mn0 ? a : mn0; mx0 = a > mx0 ? a : mx0;
            mn1 = b < mn1 ? b : mn1; mx1 = b > mx1 ? b : mx1;
            mn2 = c < mn2 ? c : mn2; mx2 = c > mx2 ? c : mx2;
        } else {
            ++bad;
        }
    };
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    // 16-byte loads when the three planes are 16-byte aligned (they are when n % 4 == 0 or the
    // planes come from the context's own staging buffers)
    const bool vec = ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)z)) & 15u) == 0;
    const uint32_t n4 = vec ? n / 4 : 0;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ y4 = reinterpret_cast<const float4*>(y);
    const float4* __restrict__ z4 = reinterpret_cast<const float4*>(z);
    for (uint32_t i = gtid; i < n4; i += gsz) {
        const float4 a = x4[i], b = y4[i], c = z4[i];
        take(a.x, b.x, c.x); take(a.y, b.y, c.y); take(a.z, b.z, c.z); take(a.w, b.w, c.w);
    }
    for (uint32_t i = n4 * 4 + gtid; i < n; i += gsz) take(x[i], y[i], z[i]);
    MinMaxPartial p;
    p.mn[0] = mn0; p.mn[1] = mn1; p.mn[2] = mn2; p.mx[0] = mx0; p.mx[1] = mx1; p.mx[2] = mx2;
    p.valid = valid; p.nonfinite = bad;
    s_p[threadIdx.x] = p;
    reduce_partials_lds(s_p);
    if (threadIdx.x == 0) partials[blockIdx.x] = s_p[0];
}

// one block per frame: reduce the block partials and derive the voxel grid
__global__ __launch_bounds__(kBlock) void k_voxel_setup(Frame* frames, uint32_t nblocks, uint32_t npass, uint32_t maxbits, uint32_t stream) {
    __shared__ MinMaxPartial s_p[kBlock];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (st->kind == kFrameClusterOnly && !st->cgrid_req) return;
    const MinMaxPartial* __restrict__ partials = F.partials;
    const float fmax = 3.402823466e+38f;
    MinMaxPartial q;
    for (int a = 0; a < 3; ++a) { q.mn[a] = fmax; q.mx[a] = -fmax; }
    q.valid = 0; q.nonfinite = 0;
    for (uint32_t i = threadIdx.x; i < nblocks; i += kBlock) {
        const MinMaxPartial p = partials[i];
        for (int a = 0; a < 3; ++a) { q.mn[a] = fminf(q.mn[a], p.mn[a]); q.mx[a] = fmaxf(q.mx[a], p.mx[a]); }
        q.valid += p.valid; q.nonfinite += p.nonfinite;
    }
    s_p[threadIdx.x] = q;
    reduce_partials_lds(s_p);
    if (st->kind == kFrameClusterOnly) {   // only the bounds are wanted (radiusSearch on a non-finite point is undefined in PCL)
        if (threadIdx.x == 0) {
            const MinMaxPartial p = s_p[0];
            if (p.nonfinite) atomicOr(&st->err, (uint32_t)kErrNonFinite);
            for (int a = 0; a < 3; ++a) { st->vmin[a] = p.mn[a]; st->vmax[a] = p.mx[a]; }
        }
        return;
    }
    if (threadIdx.x == 0) voxel_setup(st, s_p[0], npass, maxbits, stream != 0u);
    if (stream) {   // k_occ_scan<0> publishes its tile counts in vtile
        uint32_t* __restrict__ vtile = F.vtile;
        for (uint32_t t = threadIdx.x; t < kVmapCells / 32u / kVmapTile; t += kBlock) vtile[t] = 0u;
        return;
    }
    // k_voxel_reduce publishes "heads of tile t, plus one" here; zero = not yet published
    uint32_t* __restrict__ tile_heads = F.tile_a;
    const uint32_t ntiles = (st->n_in + kScanTile - 1) / kScanTile;
    for (uint32_t t = threadIdx.x; t < ntiles; t += kBlock) tile_heads[t] = 0u;
}

// ------------------------------------------------------------------------------------------
// stable LSD radix partition of records by a digit of their voxel key
//   k_radix_hist    : per-tile digit counts            -> table[tile][bin]
//   k_table_scan    : exclusive scan down each column  -> table_x[tile][bin], bin_total[bin]
//   k_radix_scatter : stable ranks (wave match-any) + scatter
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ RecSrc rec_source(const Frame& F, int pass) {
    RecSrc s;
    s.x = F.st.src_x; s.y = F.st.src_y; s.z = F.st.src_z;
    s.r = pass == 0 ? nullptr : reinterpret_cast<const Rec3*>(((pass - 1) & 1) ? F.rec_b.get() : F.rec_a.get());
    return s;
}
// output of the last of npass partition passes
__device__ __forceinline__ const Rec3* sorted_records(const Frame& F, uint32_t npass) {
    return reinterpret_cast<const Rec3*>(((npass - 1) & 1) ? F.rec_b.get() : F.rec_a.get());
}

struct PassInfo {
    uint32_t count, bits, shift, nb;   // nb = (1<<bits) + 1 bins (last = trash)
    bool drop_nonfinite;
};

__device__ __forceinline__ bool voxel_stage_off(const DevState* st) {
    return st->err || st->passthrough || st->kind == kFrameClusterOnly;
}

__device__ __forceinline__ PassInfo pass_info(const DevState* st, int pass) {
    PassInfo pi;
    pi.bits = st->plan_voxel.bits[pass]; pi.shift = st->plan_voxel.shift[pass];
    pi.nb = (1u << pi.bits) + 1u;
    pi.count = pass == 0 ? st->n_in : st->n_valid;
    pi.drop_nonfinite = pass == 0 && st->is_dense == 0;
    return pi;
}

template <int MAXB>
__global__ __launch_bounds__(kBlock) void k_radix_hist(const Frame* frames, int pass) {
    __shared__ uint32_t s_h[(1 << MAXB) + 1];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    uint32_t* __restrict__ table = F.table;
    if (voxel_stage_off(st)) return;
    const PassInfo pi = pass_info(st, pass);
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= pi.count) return;
    const RecSrc src = rec_source(F, pass);
    VoxelKey key; key.init(st);
    for (uint32_t b = threadIdx.x; b < pi.nb; b += kBlock) s_h[b] = 0;
    __syncthreads();
    const uint32_t mask = (1u << pi.bits) - 1u;
    const uint32_t wbase = tile_base + (threadIdx.x / kWave) * (kItems * kWave) + lane_id();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < pi.count) {
            const Rec3 r = src.load(i);
            uint32_t d;
            if (pi.drop_nonfinite && !finite3(r.x, r.y, r.z)) d = pi.nb - 1;
            else d = (key(r) >> pi.shift) & mask;
            atomicAdd(&s_h[d], 1u);
        }
    }
    __syncthreads();
    // bin-major table (table[bin * ntiles + tile]): the scan kernel then reads and writes whole rows
    const uint32_t ntiles = (pi.count + kTile - 1) / kTile;
    for (uint32_t b = threadIdx.x; b < pi.nb; b += kBlock) table[(size_t)b * ntiles + blockIdx.x] = s_h[b];
}

// which = 0: voxel partition pass, 1: bucket partition of the streaming voxel path, 2: CSR partition pass
__device__ __forceinline__ bool table_dims(const DevState* st, int which, int pass, uint32_t* ntiles, uint32_t* nb) {
    if (st->err) return false;
    if (which == 2) {
        if (st->trivial || st->kind == kFrameVoxelOnly) return false;
        *nb = (1u << st->plan_csr.bits[pass]) + 1u;
        *ntiles = (st->c_n + kTile - 1) / kTile;
        return st->n_kept != 0;
    }
    if (voxel_stage_off(st)) return false;
    if (which == 1) {   // the bucket partition of the streaming voxel path
        if (!st->vpath || st->n_valid == 0) return false;
        *nb = st->nbuckets + 1u;
        *ntiles = (st->n_in + kTile - 1) / kTile;
        return true;
    }
    const PassInfo pi = pass_info(st, pass);
    *nb = pi.nb;
    *ntiles = (pi.count + kTile - 1) / kTile;
    return true;
}

// Exclusive scan along each row of the bin-major tile-count table: block b owns bin b.
// grid = kMaxBins blocks (blocks past the pass's bin count exit at once).
__global__ __launch_bounds__(kBlock) void k_table_scan(const Frame* frames, int which, int pass) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ table = F.table;
    uint32_t* __restrict__ table_x = F.table_x;
    uint32_t* __restrict__ bin_total = F.bin_total;
    uint32_t ntiles, nb;
    if (!table_dims(st, which, pass, &ntiles, &nb)) return;
    const uint32_t b = blockIdx.x;
    if (b >= nb) return;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += kBlock) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < ntiles ? table[(size_t)b * ntiles + t] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, s_w, &tot);
        if (t < ntiles) table_x[(size_t)b * ntiles + t] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) bin_total[b] = carry;
}

// Exclusive scan of bin_total[0..nb) into LDS s_base[0..nb); every thread of the block calls.
template <int MAXB>
__device__ __forceinline__ void scan_bins_to_lds(const uint32_t* __restrict__ bin_total, uint32_t nb,
                                                 uint32_t* s_base, uint32_t* s_w) {
    constexpr int kPer = ((1 << MAXB) + 1 + kBlock - 1) / kBlock;   // 9 for 11 bits, 2 for 8
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

// The same peer mask through LDS instead of one ballot per digit bit: every lane ORs its lane bit into
// the 64-bit word of its digit (ds_or_b64), reads the word back, and the first peer clears it for the
// next step.  A wave's LDS operations execute in program order, so the read sees all 64 ORs.  About ten
// VALU instructions per record instead of about sixty; `match_w` must be all zero on entry and is left so.
__device__ __forceinline__ uint64_t match_digit_lds(unsigned long long* match_w, uint32_t d, bool active) {
    unsigned long long peers = 0ull;
    if (active) atomicOr(match_w + d, 1ull << lane_id());
    __builtin_amdgcn_wave_barrier();
    if (active) peers = match_w[d];   // not forwardable: the atomic OR above changed the word
    __builtin_amdgcn_wave_barrier();
    if (active && (peers & lanemask_lt()) == 0ull) match_w[d] = 0ull;
    __builtin_amdgcn_wave_barrier();
    return peers;
}
constexpr int kMatchBins(int maxb) { return maxb <= kNarrowBits ? (1 << maxb) + 1 : 1; }   // wide digits keep the ballots

template <int MAXB>
__global__ __launch_bounds__(kBlock) void k_radix_scatter(const Frame* frames, int pass) {
    constexpr int kBins = (1 << MAXB) + 1;
    __shared__ uint32_t s_hist[kWavesPerBlock * kBins];   // per wave and bin: count, then offset inside the bin
    __shared__ uint32_t s_base[kBins];                    // global start of the bin
    __shared__ uint32_t s_lbase[kBins];                   // start of the bin inside the staged tile
    __shared__ uint32_t s_delta[kBins];                   // global position minus staged position, per bin
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    // the tile in output order (phases 3-4): {x, y, z, global position - staged position}, one 16-byte LDS
    // access per record on either side
    __shared__ float4 s_stage[kTile];
    // the peer-mask words of phase 1 live in the staging area, which nothing touches before phase 3
    unsigned long long* s_match = reinterpret_cast<unsigned long long*>(s_stage);
    static_assert(sizeof(unsigned long long) * kWavesPerBlock * kMatchBins(MAXB) <= sizeof(float4) * kTile, "peer masks fit the staging area");
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ table_x = F.table_x;
    const uint32_t* __restrict__ bin_total = F.bin_total;
    Rec3* __restrict__ out = reinterpret_cast<Rec3*>((pass & 1) ? F.rec_b.get() : F.rec_a.get());
    if (voxel_stage_off(st)) return;
    const PassInfo pi = pass_info(st, pass);
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= pi.count) return;
    const RecSrc src = rec_source(F, pass);
    VoxelKey key; key.init(st);
    const uint32_t nb = pi.nb;
    for (uint32_t b = threadIdx.x; b < kWavesPerBlock * nb; b += kBlock) s_hist[b] = 0;
    for (uint32_t b = threadIdx.x; b < (uint32_t)(kWavesPerBlock * kMatchBins(MAXB)); b += kBlock) s_match[b] = 0ull;
    scan_bins_to_lds<MAXB>(bin_total, nb, s_base, s_w);   // ends with a barrier

    // phase 1: each wave walks its contiguous sub-tile in order; ranks are stable by construction
    const uint32_t w = threadIdx.x / kWave;
    uint32_t* hist_w = s_hist + w * nb;
    const uint32_t mask = (1u << pi.bits) - 1u;
    const uint32_t wbase = tile_base + w * (kItems * kWave) + lane_id();
    Rec3 rec[kItems];
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
        uint64_t peers;
        if constexpr (MAXB <= kNarrowBits) peers = match_digit_lds(s_match + w * kBins, d, active);
        else peers = match_digit(d, pi.bits + 1, active);
        const uint32_t rank = __popcll(peers & lanemask_lt());
        uint32_t old = 0;
        if (active) old = hist_w[d];
        if (active && rank == 0) hist_w[d] = old + (uint32_t)__popcll(peers);
        dig[j] = d; lrank[j] = old + rank;
    }
    __syncthreads();
    // phase 2: per bin, per-wave counts -> per-wave offsets inside the bin; tile count of the bin
    const uint32_t ntiles = (pi.count + kTile - 1) / kTile;
    constexpr int kPer = (kBins + kBlock - 1) / kBlock;
    uint32_t tcnt[kPer], gstart[kPer];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint32_t b = threadIdx.x * kPer + k;
        tcnt[k] = 0; gstart[k] = 0;
        if (b < nb) {
            uint32_t run = 0;
#pragma unroll
            for (int q = 0; q < kWavesPerBlock; ++q) {
                const uint32_t t = s_hist[q * nb + b];
                s_hist[q * nb + b] = run;
                run += t;
            }
            tcnt[k] = run;
            gstart[k] = s_base[b] + table_x[(size_t)b * ntiles + blockIdx.x];
        }
        tsum += tcnt[k];
    }
    uint32_t ttot;
    uint32_t lrun = block_excl_scan(tsum, s_w, &ttot);   // bins of one thread are consecutive
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint32_t b = threadIdx.x * kPer + k;
        if (b < nb) { s_lbase[b] = lrun; s_delta[b] = gstart[k] - lrun; }
        lrun += tcnt[k];
    }
    __syncthreads();
    // phase 3: stage the tile in output order (bin-major, stable inside a bin)
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < pi.count) {
            const uint32_t lp = s_lbase[dig[j]] + hist_w[dig[j]] + lrank[j];
            s_stage[lp] = make_float4(rec[j].x, rec[j].y, rec[j].z, u2f(s_delta[dig[j]]));
        }
    }
    __syncthreads();
    // phase 4: consecutive staged records of a bin go to consecutive global positions
    for (uint32_t e = threadIdx.x; e < ttot; e += kBlock) {
        const float4 v = s_stage[e];
        Rec3 r; r.x = v.x; r.y = v.y; r.z = v.z;
        out[e + f2u(v.w)] = r;
    }
}

// ------------------------------------------------------------------------------------------
// run heads over the sorted records -> voxel rank; centroids
// ------------------------------------------------------------------------------------------
// End of the voxel stage: M, and what the cluster stage reads.
__device__ __forceinline__ void finish_voxel(DevState* st, uint32_t m) {
    st->m = m;
    st->c_from_input = st->passthrough ? 1u : 0u;
    st->c_n = m;
    uint32_t cap = 1024;
    while (cap < 2u * m && cap < 0x80000000u) cap <<= 1;
    st->hash_mask = cap - 1u;
}

// Voxel centroids (voxel_grid.hpp "Fourth pass", PCL >= 1.8 CentroidPoint / AccumulatorXYZ): one
// thread per run head sums the run in order (= ascending point index, the partition is stable)
// starting from 0.0f and divides by float(count).
// The rank of a tile's first voxel is the number of run heads in the tiles before it.  Each block
// publishes its own count (plus one) as soon as it knows it and reads the counts of the lower tiles
// after its sums are done; a lower tile never waits for a higher one and blocks are dispatched in
// index order, so the lower tiles are always running or finished (the wait is bounded anyway).
constexpr uint32_t kSpinLimit = 1u << 24;

__global__ __launch_bounds__(kBlock) void k_voxel_reduce(Frame* frames, uint32_t npass) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    __shared__ Rec3 s_rec[kScanTile];
    __shared__ uint32_t s_key[kScanTile + 1];
    __shared__ uint16_t s_hpos[kScanTile + 2];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const Rec3* __restrict__ rec = sorted_records(F, npass);
    uint32_t* __restrict__ tile_heads = F.tile_a;
    float* __restrict__ cx = F.cen_x; float* __restrict__ cy = F.cen_y; float* __restrict__ cz = F.cen_z;
    if (st->err || st->kind == kFrameClusterOnly) return;
    if (st->passthrough) {   // "output = *input_": the voxel cloud is the input cloud, every point kept
        if (blockIdx.x == 0 && threadIdx.x == 0) finish_voxel(st, st->n_in);
        return;
    }
    const uint32_t n = st->n_valid;
    if (n == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) finish_voxel(st, 0);
        return;
    }
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    VoxelKey key; key.init(st);
    // stage the tile's records and keys in LDS (coalesced), so that the in-order run sums below
    // chase LDS, not L2
    const uint32_t cnt = min((uint32_t)kScanTile, n - base);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t j = k * kBlock + threadIdx.x;
        if (j < cnt) { const Rec3 r = rec[base + j]; s_rec[j] = r; s_key[j + 1] = key(r); }
    }
    if (threadIdx.x == 0) s_key[0] = base > 0 ? key(rec[base - 1]) : 0u;
    __syncthreads();
    // run heads of the tile, compacted: s_hpos[h] = tile position of the h-th head, s_hpos[tot] = cnt
    bool head[kScanItems];
    uint32_t c = 0;
    const uint32_t j0 = threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t j = j0 + k;
        head[k] = j < cnt && ((base + j == 0) || (s_key[j + 1] != s_key[j]));
        c += head[k] ? 1u : 0u;
    }
    uint32_t tot;
    uint32_t h0 = block_excl_scan(c, s_w, &tot);
    if (threadIdx.x == 0) st_agent(tile_heads + blockIdx.x, tot + 1u);   // publish
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (head[k]) s_hpos[h0++] = (uint16_t)(j0 + k);
    if (threadIdx.x == 0) s_hpos[tot] = (uint16_t)cnt;
    __syncthreads();
    // one thread per head: every lane has a run to sum (a thread per record would leave most lanes idle)
    float ox[kScanItems], oy[kScanItems], oz[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t h = k * kBlock + threadIdx.x;
        ox[k] = oy[k] = oz[k] = 0.0f;
        if (h < tot) {
            const uint32_t j = s_hpos[h], e = s_hpos[h + 1];
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
            for (uint32_t jj = j; jj < e; ++jj) {   // the part of the run inside this tile
                const Rec3 r = s_rec[jj];
                sx += r.x; sy += r.y; sz += r.z;
            }
            uint32_t len = e - j;
            if (e == cnt) {   // the tile's last run may continue in the next tile(s)
                const uint32_t kc = s_key[j + 1];
                for (uint32_t g = base + cnt; g < n; ++g) {
                    const Rec3 r = rec[g];
                    if (key(r) != kc) break;
                    sx += r.x; sy += r.y; sz += r.z;
                    ++len;
                }
            }
            const float fc = (float)len;
            ox[k] = sx / fc; oy[k] = sy / fc; oz[k] = sz / fc;
        }
    }
    // heads in the tiles before this one
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t v = ld_agent(tile_heads + t), spins = 0;
        while (v == 0u) {
            if (++spins > kSpinLimit) { atomicOr(&st->err, (uint32_t)kErrInternal); v = 1u; break; }
            __builtin_amdgcn_s_sleep(2);
            v = ld_agent(tile_heads + t);
        }
        mine += v - 1u;
    }
    __syncthreads();   // s_w is reused
    const uint32_t before = block_sum(mine, s_w);
    if (threadIdx.x == 0 && base + kScanTile >= n) finish_voxel(st, before + tot);   // the last tile
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t h = k * kBlock + threadIdx.x;
        if (h < tot) { cx[before + h] = ox[k]; cy[before + h] = oy[k]; cz[before + h] = oz[k]; }
    }
}

// ------------------------------------------------------------------------------------------
// Streaming voxel path (default when the voxel grid has at most kVmapCells cells and the frame at
// most kStreamBuckets * kVcapMax voxels; otherwise the host re-runs the frame on the sort path).
//
// PCL sorts the points by voxel key and sums every run in order.  What that needs is (a) the
// rank of every occupied voxel in key order and (b) for every voxel its points in index order.
//   k_vox_mark     : key of every point; one byte per grid cell set for the occupied ones
//   k_occ_scan<0>  : bytes -> occupancy bits + running popcount  =>  voxel id = rank in key order, M
//   k_vid_hist / k_table_scan / k_vid_scatter : ONE stable 8-bit partition of {x,y,z,id} records
//                    by bucket = id >> vshift (up to 256 buckets of up to vcap voxels)
//   k_voxel_stream : one wave per bucket walks the bucket's records in order (= index order) and
//                    adds each to its voxel's accumulator in LDS, starting from 0.0f; lanes that
//                    hit the same voxel in one step go in lane order.  cen[id] = sum / float(count).
// Same float operations in the same order as the sort path, at about half of its HBM traffic.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool stream_off(const DevState* st) {
    return voxel_stage_off(st) || !st->vpath || st->n_valid == 0;
}

__global__ __launch_bounds__(kBlock) void k_vox_mark(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    uint8_t* __restrict__ vflag = F.vflag; uint32_t* __restrict__ vkey = F.vkey;
    if (stream_off(st)) return;
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    const uint32_t n = st->n_in;
    const bool drop = st->is_dense == 0;
    VoxelKey key; key.init(st);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Rec3 r; r.x = x[i]; r.y = y[i]; r.z = z[i];
        uint32_t k = kNoKey;
        if (!drop || finite3(r.x, r.y, r.z)) {
            k = key(r);
            vflag[k] = 1;   // every writer stores the same value: no atomic needed
        }
        vkey[i] = k;
    }
}

// 16 flag bytes (each 0 or 1) -> 16 bits
__device__ __forceinline__ uint32_t flags_to_bits(const uint4& v) {
    const uint32_t a = ((v.x & 0x01010101u) * 0x01020408u) >> 24, b = ((v.y & 0x01010101u) * 0x01020408u) >> 24;
    const uint32_t c = ((v.z & 0x01010101u) * 0x01020408u) >> 24, d = ((v.w & 0x01010101u) * 0x01020408u) >> 24;
    return (a & 15u) | ((b & 15u) << 4) | ((c & 15u) << 8) | ((d & 15u) << 12);
}

// One block per kVmapTile occupancy words: bytes -> bits (and the bytes are zeroed again for the
// next frame), popcount scan; the tile publishes its count and adds up the counts of the lower
// tiles (same publish / bounded-wait scheme as k_voxel_reduce).  The last tile knows the total.
//   WHICH = 0: the voxel grid (k_vox_mark)   -> voxel ids, M, the bucket plan
//   WHICH = 1: the dense cell grid (k_cell_mark) -> cell ids, n_cells
__device__ __forceinline__ bool cluster_stage_off(const DevState* st);

template <int WHICH>
__global__ __launch_bounds__(kBlock) void k_occ_scan(Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    __shared__ uint16_t s_half[2 * kVmapTile];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    uint4* __restrict__ flags = reinterpret_cast<uint4*>(WHICH == 0 ? F.vflag.get() : F.cflag.get());
    uint2* __restrict__ omap = WHICH == 0 ? F.vmap.get() : F.cmap.get();
    uint32_t* __restrict__ otile = WHICH == 0 ? F.vtile.get() : F.ctile.get();
    // Runs whenever the marking kernel ran (the replan bits this kernel raises itself must not stop its
    // other tiles: every tile has flag bytes to turn back to zero).
    uint32_t words;
    if (WHICH == 0) {
        const uint32_t own = (uint32_t)(kErrReplanPath | kErrReplanVcap | kErrInternal);
        if ((st->err & ~own) || st->passthrough || st->kind == kFrameClusterOnly || !st->vpath || st->n_valid == 0) return;
        words = st->vmap_words;
    } else {
        const uint32_t own = (uint32_t)(kErrReplanCpath | kErrReplanCtiles | kErrInternal | kErrNonFinite);
        if ((st->err & ~own) || st->trivial || st->kind == kFrameVoxelOnly || !st->cpath) return;
        words = st->cmap_words;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && words > gridDim.x * kVmapTile)   // the host sized this grid from the previous frame
        atomicOr(&st->err, (uint32_t)(WHICH == 0 ? kErrReplanVcap : kErrReplanCtiles));
    const uint32_t base = blockIdx.x * kVmapTile;
    if (base >= words) return;
    const uint32_t cnt = min(kVmapTile, words - base);
    // a word = 32 cells = two 16-byte loads; the flag array is padded to whole tiles
#pragma unroll
    for (uint32_t k = 0; k < 2 * kVmapTile / kBlock; ++k) {
        const uint32_t h = k * kBlock + threadIdx.x;   // half-word of the tile
        uint32_t bits = 0;
        if (h < 2 * cnt) {
            const size_t at = (size_t)base * 2 + h;
            bits = flags_to_bits(flags[at]);
            if (bits) flags[at] = make_uint4(0u, 0u, 0u, 0u);
        }
        s_half[h] = (uint16_t)bits;
    }
    __syncthreads();
    constexpr uint32_t kPer = kVmapTile / kBlock;   // consecutive words per thread
    uint32_t w[kPer];
    uint32_t c = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
        const uint32_t j = threadIdx.x * kPer + k;
        w[k] = (uint32_t)s_half[2 * j] | ((uint32_t)s_half[2 * j + 1] << 16);
        c += __popc(w[k]);
    }
    uint32_t tot;
    uint32_t run = block_excl_scan(c, s_w, &tot);
    if (threadIdx.x == 0) st_agent(otile + blockIdx.x, tot + 1u);   // publish
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t v = ld_agent(otile + t), spins = 0;
        while (v == 0u) {
            if (++spins > kSpinLimit) { atomicOr(&st->err, (uint32_t)kErrInternal); v = 1u; break; }
            __builtin_amdgcn_s_sleep(2);
            v = ld_agent(otile + t);
        }
        mine += v - 1u;
    }
    __syncthreads();   // s_w is reused
    const uint32_t before = block_sum(mine, s_w);
    run += before;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
        const uint32_t j = threadIdx.x * kPer + k;
        if (j < cnt) omap[base + j] = make_uint2(w[k], run);
        run += __popc(w[k]);
    }
    if (threadIdx.x == 0 && base + kVmapTile >= words) {   // the last tile
        const uint32_t m = before + tot;
        if (WHICH == 0) {   // M and the bucket plan
            finish_voxel(st, m);
            uint32_t sh = 0;
            while (((uint64_t)kStreamBuckets << sh) < (uint64_t)m) ++sh;   // 256 buckets of 2^sh voxels cover M
            st->vshift = sh;
            st->nbuckets = m ? ((m - 1u) >> sh) + 1u : 0u;
            if ((1u << sh) > kVcapMax) atomicOr(&st->err, (uint32_t)kErrReplanPath);
            else if ((1u << sh) > st->vcap) atomicOr(&st->err, (uint32_t)kErrReplanVcap);
        } else {
            st->n_cells = m;
        }
    }
}

// voxel id of a key: occupied cells before it in key order
__device__ __forceinline__ uint32_t voxel_id(const uint2* __restrict__ vmap, uint32_t key) {
    const uint2 e = vmap[key >> 5];
    return e.y + __popc(e.x & ((1u << (key & 31u)) - 1u));
}

template <int MAXB>
__global__ __launch_bounds__(kBlock) void k_vid_hist(const Frame* frames) {
    __shared__ uint32_t s_h[(1 << MAXB) + 1];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    uint32_t* __restrict__ table = F.table; uint32_t* __restrict__ vkey = F.vkey;
    const uint2* __restrict__ vmap = F.vmap;
    if (stream_off(st)) return;
    const uint32_t n = st->n_in;
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= n) return;
    const uint32_t nb = st->nbuckets + 1u, sh = st->vshift;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) s_h[b] = 0;
    __syncthreads();
    const uint32_t wbase = tile_base + (threadIdx.x / kWave) * (kItems * kWave) + lane_id();
    uint32_t k[kItems];
#pragma unroll
    for (int j = 0; j < kItems; ++j) { const uint32_t i = wbase + j * kWave; k[j] = i < n ? vkey[i] : kNoKey; }
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < n) {
            uint32_t d = nb - 1u;
            if (k[j] != kNoKey) {
                const uint32_t id = voxel_id(vmap, k[j]);
                vkey[i] = id;   // key -> id, in place
                d = id >> sh;
            }
            atomicAdd(&s_h[d], 1u);
        }
    }
    __syncthreads();
    const uint32_t ntiles = (n + kTile - 1) / kTile;
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) table[(size_t)b * ntiles + blockIdx.x] = s_h[b];
}

// The partition of the streaming path: records {x, y, z, id} go to rec_b, stable, bucket-major.
template <int MAXB>
__global__ __launch_bounds__(kBlock) void k_vid_scatter(const Frame* frames) {
    constexpr int kBins = (1 << MAXB) + 1;
    __shared__ uint32_t s_hist[kWavesPerBlock * kBins];
    __shared__ uint32_t s_base[kBins];
    __shared__ uint32_t s_lbase[kBins];
    __shared__ uint32_t s_delta[kBins];
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    __shared__ float4 s_stage[kTile];
    unsigned long long* s_match = reinterpret_cast<unsigned long long*>(s_stage);
    static_assert(MAXB <= kNarrowBits, "the streaming path partitions on an 8-bit digit");
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ table_x = F.table_x; const uint32_t* __restrict__ bin_total = F.bin_total;
    const uint32_t* __restrict__ vkey = F.vkey;
    float4* __restrict__ out = F.rec_b;
    if (stream_off(st)) return;
    const uint32_t n = st->n_in;
    const uint32_t tile_base = blockIdx.x * kTile;
    if (tile_base >= n) return;
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    const uint32_t nb = st->nbuckets + 1u, sh = st->vshift;
    for (uint32_t b = threadIdx.x; b < kWavesPerBlock * nb; b += kBlock) s_hist[b] = 0;
    for (uint32_t b = threadIdx.x; b < (uint32_t)(kWavesPerBlock * kBins); b += kBlock) s_match[b] = 0ull;
    scan_bins_to_lds<MAXB>(bin_total, nb, s_base, s_w);   // ends with a barrier
    const uint32_t w = threadIdx.x / kWave;
    uint32_t* hist_w = s_hist + w * nb;
    const uint32_t wbase = tile_base + w * (kItems * kWave) + lane_id();
    float4 rec[kItems];
    uint32_t dig[kItems], lrank[kItems];
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < n) rec[j] = make_float4(x[i], y[i], z[i], u2f(vkey[i]));
    }
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        const bool active = i < n;
        uint32_t d = 0;
        if (active) { const uint32_t id = f2u(rec[j].w); d = id == kNoKey ? nb - 1u : id >> sh; }
        const uint64_t peers = match_digit_lds(s_match + w * kBins, d, active);
        const uint32_t rank = __popcll(peers & lanemask_lt());
        uint32_t old = 0;
        if (active) old = hist_w[d];
        if (active && rank == 0) hist_w[d] = old + (uint32_t)__popcll(peers);
        dig[j] = d; lrank[j] = old + rank;
    }
    __syncthreads();
    const uint32_t ntiles = (n + kTile - 1) / kTile;
    constexpr int kPer = (kBins + kBlock - 1) / kBlock;
    uint32_t tcnt[kPer], gstart[kPer];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint32_t b = threadIdx.x * kPer + k;
        tcnt[k] = 0; gstart[k] = 0;
        if (b < nb) {
            uint32_t run = 0;
#pragma unroll
            for (int q = 0; q < kWavesPerBlock; ++q) {
                const uint32_t t = s_hist[q * nb + b];
                s_hist[q * nb + b] = run;
                run += t;
            }
            tcnt[k] = run;
            gstart[k] = s_base[b] + table_x[(size_t)b * ntiles + blockIdx.x];
        }
        tsum += tcnt[k];
    }
    uint32_t ttot;
    uint32_t lrun = block_excl_scan(tsum, s_w, &ttot);
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint32_t b = threadIdx.x * kPer + k;
        if (b < nb) { s_lbase[b] = lrun; s_delta[b] = gstart[k] - lrun; }
        lrun += tcnt[k];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < n) s_stage[s_lbase[dig[j]] + hist_w[dig[j]] + lrank[j]] = rec[j];
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < ttot; e += kBlock) {
        const float4 v = s_stage[e];
        const uint32_t id = f2u(v.w);
        const uint32_t d = id == kNoKey ? nb - 1u : id >> sh;
        out[e + s_delta[d]] = v;
    }
}

// One block per bucket.  With kStreamWaves > 1 the bucket's voxels are dealt out to the block's waves
// (voxel & (waves-1)) and every wave walks all of the bucket's records, accumulating the ones it owns;
// measured on C3: 1 wave 66 us, 2 waves 70 us, 4 waves 77 us per batch - the kernel is bound by the chain of
// LDS round trips of the longest bucket (about 8 per step), not by waves in flight, so one wave it is.
// Dynamic LDS: float4 acc[vcap] {sum x, sum y, sum z, count} and unsigned long long peers[vcap].
constexpr int kStreamWaves = 1;
__global__ __launch_bounds__(kStreamWaves * kWave) void k_voxel_stream(const Frame* frames) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_b;
    const uint32_t* __restrict__ bin_total = F.bin_total;
    float* __restrict__ cx = F.cen_x; float* __restrict__ cy = F.cen_y; float* __restrict__ cz = F.cen_z;
    if (stream_off(st)) return;
    const uint32_t b = blockIdx.x;
    if (b >= st->nbuckets) return;
    const uint32_t vcap = st->vcap, sh = st->vshift, m = st->m;
    float4* acc = reinterpret_cast<float4*>(s_dyn);
    unsigned long long* peers_w = reinterpret_cast<unsigned long long*>(s_dyn + (size_t)vcap * sizeof(float4));
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x / kWave;
    // records of the bucket: [start, start + count)
    uint32_t before = 0;
    for (uint32_t t = lane; t < b; t += kWave) before += bin_total[t];
    const uint32_t start = wave_sum(before), count = bin_total[b];
    const uint32_t id0 = b << sh;
    const uint32_t nv = min(1u << sh, m - id0);   // voxels of this bucket
    for (uint32_t v = threadIdx.x; v < nv; v += blockDim.x) { acc[v] = make_float4(0.0f, 0.0f, 0.0f, u2f(0u)); peers_w[v] = 0ull; }
    __syncthreads();
    const uint32_t steps = (count + kWave - 1) / kWave;
    // records two steps ahead are in flight while a step runs (clamped, unconditional loads: a load
    // under a branch would be waited for at the branch's join)
    const uint32_t last = start + count - 1u;
    auto step = [&](const float4& r, uint32_t i) {
        const uint32_t lv0 = f2u(r.w) - id0;
        const bool active = i < count && (lv0 & (uint32_t)(kStreamWaves - 1)) == wv;   // this wave's voxels only
        const uint32_t lv = active ? lv0 : 0u;
        const uint64_t peers = match_digit_lds(peers_w, lv, active);
        const uint32_t rank = active ? (uint32_t)__popcll(peers & lanemask_lt()) : 0xFFFFFFFFu;
        for (uint32_t q = 0; __any(active && rank >= q); ++q) {   // same voxel: lane order = index order
            if (rank == q) {
                float4 a = acc[lv];
                a.x += r.x; a.y += r.y; a.z += r.z; a.w = u2f(f2u(a.w) + 1u);
                acc[lv] = a;
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    // kStreamDepth register sets, each refilled right after it is consumed: a record is requested
    // kStreamDepth - 1 steps (a few thousand cycles) before it is needed, and no set is ever copied
    constexpr uint32_t kStreamDepth = 8;
    float4 rs[kStreamDepth];
#pragma unroll
    for (uint32_t k = 0; k < kStreamDepth; ++k) {
        rs[k] = rec[min(start + k * kWave + lane, last)];
        __builtin_amdgcn_sched_barrier(0);   // keep the issue order = the order of use (the waits count loads in order)
    }
    for (uint32_t sidx = 0; sidx < steps; sidx += kStreamDepth) {
#pragma unroll
        for (uint32_t k = 0; k < kStreamDepth; ++k) {
            const uint32_t i = (sidx + k) * kWave + lane;
            if (sidx + k < steps) step(rs[k], i);   // wave-uniform
            // not under the branch: a load issued inside it would be waited for at its join
            rs[k] = rec[min(start + i + kStreamDepth * kWave, last)];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < nv; v += blockDim.x) {
        const float4 a = acc[v];
        const float fc = (float)f2u(a.w);
        cx[id0 + v] = a.x / fc; cy[id0 + v] = a.y / fc; cz[id0 + v] = a.z / fc;
    }
}

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

__device__ __forceinline__ uint32_t hash_find_or_insert(unsigned long long* hkey, uint32_t mask, unsigned long long key) {
    const unsigned long long want = key + 1ull;
    uint32_t slot = cell_home(key, mask);
    while (true) {
        unsigned long long k = hkey[slot];
        if (k == 0ull) k = atomicCAS(hkey + slot, 0ull, want);
        if (k == 0ull || k == want) return slot;
        slot = (slot + 64u) & mask;
    }
}

__device__ __forceinline__ uint32_t hash_lookup(const unsigned long long* __restrict__ hkey, uint32_t mask,
                                                unsigned long long key) {
    const unsigned long long want = key + 1ull;
    uint32_t slot = cell_home(key, mask);
    while (true) {
        const unsigned long long k = hkey[slot];
        if (k == want) return slot;
        if (k == 0ull) return kNoSlot;
        slot = (slot + 64u) & mask;
    }
}

__device__ __forceinline__ bool cluster_stage_off(const DevState* st) {
    return st->err || st->trivial || st->kind == kFrameVoxelOnly;
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
        if (prio[a] > prio[b]) { const uint32_t t = a; a = b; b = t; }
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
        while (va != wa && va != 0ull) { sa = (sa + 64u) & mask; va = hkey[sa]; }
        while (vc != wc && vc != 0ull) { sc = (sc + 64u) & mask; vc = hkey[sc]; }
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

// ------------------------------------------------------------------------------------------
// Dense cell grid (default; the hash table above is the fallback for grids beyond kCmapCells).
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
//   k_cell_mark     grid position of every point; one byte per occupied cell
//   k_occ_scan<1>   bytes -> occupancy bits + running popcount => cell ids, n_cells
//   k_cell_count    points per cell, rank inside the cell, smallest point index (= priority)
//   k_cell_starts   exclusive scan of the counts; per-cell metadata; union-find init
//   k_cell_place    (shared with the hash path) records sorted by cell
//   k_cgrid_hook / k_cell_compress / k_cgrid_union   components, lane = cell
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cgrid_off(const DevState* st) { return cluster_stage_off(st) || !st->cpath; }

__global__ __launch_bounds__(kBlock) void k_cgrid_setup(Frame* frames) {
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (cluster_stage_off(st) || !st->cgrid_req) return;
    uint32_t* __restrict__ ctile = F.ctile; uint32_t* __restrict__ tile_pub = F.tile_a;
    for (uint32_t t = threadIdx.x; t < kCmapCells / 32u / kVmapTile; t += kBlock) ctile[t] = 0u;
    const uint32_t ntiles = (st->c_n + kScanTile - 1) / kScanTile;   // k_cell_starts publishes here
    for (uint32_t t = threadIdx.x; t < ntiles; t += kBlock) tile_pub[t] = 0u;
    if (threadIdx.x != 0) return;
    st->n_cells = 0;
    if (st->c_n == 0) { st->cpath = 1u; st->cmap_words = 0u; return; }
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
    const double ig = st->inv_cell;
    const long long c0x = st->c0[0], c0y = st->c0[1], c0z = st->c0[2];
    const long long d0 = st->c_dim[0], d1 = st->c_dim[1], d2 = st->c_dim[2];
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float px = x[i], py = y[i], pz = z[i];
        uint32_t pos = 0;
        if (!finite3(px, py, pz)) {   // radiusSearch on a non-finite point is undefined in PCL
            atomicOr(&st->err, (uint32_t)kErrNonFinite);
        } else {
            const long long gx = (long long)floor((double)px * ig) - c0x, gy = (long long)floor((double)py * ig) - c0y,
                            gz = (long long)floor((double)pz * ig) - c0z;
            // two free cells on every side keep the whole 5x5x5 stencil inside the array
            if (gx < 2 || gy < 2 || gz < 2 || gx > d0 - 3 || gy > d1 - 3 || gz > d2 - 3) {
                atomicOr(&st->err, (uint32_t)kErrReplanCpath);
            } else {
                pos = (uint32_t)gz * plane + (uint32_t)gy * rowbits + (uint32_t)gx;
                cflag[pos] = 1;
            }
        }
        pt_pos[i] = pos;
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
    uint32_t* __restrict__ ccount = F.hcount; uint32_t* __restrict__ cmin = F.hmin;
    uint32_t* __restrict__ pt_slot = F.pt_slot; uint32_t* __restrict__ pt_rank = F.pt_rank;
    uint32_t* __restrict__ cell_pos = F.cell_list;
    if (cgrid_off(st)) return;
    const uint32_t n = st->c_n;
    const uint32_t lane = lane_id();
    const uint32_t nthreads = gridDim.x * blockDim.x;
    // ccount is all zero and cmin all ones on entry: k_emit leaves the borrowed arrays that way
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
            atomicMin(cmin + cid, i);
            cell_pos[cid] = pos;   // every writer of a cell stores the same value
        }
        const uint64_t le = leaders & (lanemask_lt() | (1ull << lane));
        const int ll = le ? 63 - __clzll((unsigned long long)le) : 0;
        const uint32_t r = __shfl(rank0, ll, kWave);
        if (valid) { pt_slot[i] = cid; pt_rank[i] = r + (lane - (uint32_t)ll); }
    }
}

// cell_start = exclusive scan of the counts (tiles publish / add up the lower tiles); per-cell
// metadata {first record, points, smallest point index, grid position}; union-find init
__global__ __launch_bounds__(kBlock) void k_cell_starts(Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const uint32_t* __restrict__ ccount = F.hcount; const uint32_t* __restrict__ cmin = F.hmin;
    const uint32_t* __restrict__ cell_pos = F.cell_list;
    uint32_t* __restrict__ tile_pub = F.tile_a;
    uint32_t* __restrict__ cell_start = F.cell_start; uint32_t* __restrict__ parent = F.parent;
    uint32_t* __restrict__ comp_size = F.comp_size; uint32_t* __restrict__ comp_rank = F.comp_rank;
    uint4* __restrict__ slot_meta = F.slot_meta;
    if (cgrid_off(st)) return;
    const uint32_t nc = st->n_cells;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= nc) return;
    uint32_t v[kScanItems];
    uint32_t c = 0;
    const uint32_t s0 = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = s0 + k < nc ? ccount[s0 + k] : 0u; c += v[k]; }
    uint32_t tot;
    uint32_t run = block_excl_scan(c, s_w, &tot);
    if (threadIdx.x == 0) st_agent(tile_pub + blockIdx.x, tot + 1u);
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) {
        uint32_t q = ld_agent(tile_pub + t), spins = 0;
        while (q == 0u) {
            if (++spins > kSpinLimit) { atomicOr(&st->err, (uint32_t)kErrInternal); q = 1u; break; }
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
            slot_meta[id] = make_uint4(run, v[k], cmin[id], cell_pos[id]);
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

// parent[c] = the adjacent cell of smallest priority among the 13 backward cells of the 3x3x3
// neighbourhood (5 rows), if it has a smaller priority than c.  No atomics: a forest.
// Two rounds of independent loads (row windows, then the metadata of the 13 neighbour slots; a slot
// without a neighbour re-reads the cell's own), then pair tests only against cells of smaller priority.
__global__ __launch_bounds__(kBlock) void k_cgrid_hook(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    uint32_t* __restrict__ parent = F.parent;
    if (cgrid_off(st)) return;
    const uint32_t nc = st->n_cells;
    const float r2 = st->r2;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
        const uint4 ma = slot_meta[c];   // {first record, points, priority, grid position}
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
        unsigned long long best = ~0ull;
#pragma unroll
        for (int q = 0; q < 13; ++q) {
            const unsigned long long key = ((unsigned long long)mb[q].z << 32) | nid[q];
            if (mb[q].z < ma.z && key < best && lane_cells_hit(rec, ma.x, ma.y, mb[q].x, mb[q].y, r2)) best = key;
        }
        if (best != ~0ull) parent[c] = (uint32_t)best;
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

__global__ __launch_bounds__(kBlock) void k_cgrid_union(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const float4* __restrict__ rec = F.rec_a;
    const uint2* __restrict__ cmap = F.cmap;
    const uint4* __restrict__ slot_meta = F.slot_meta;
    const uint32_t* __restrict__ prio = F.hmin;
    uint32_t* __restrict__ parent = F.parent;
    if (cgrid_off(st) || st->debug_skip_union == 1u) return;
    const bool scan_only = st->debug_skip_union == 2u;
    const uint32_t nc = st->n_cells;
    const float r2 = st->r2;
    const uint32_t rowbits = st->c_rowbits, plane = st->c_plane;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
        const uint4 ma = slot_meta[c];
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
        // some neighbour may be in another set: walk the rows again and look properly at those whose parent
        // word differs from the cell's (roots, pair test, union by CAS)
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
    }
}

// Root cell of every point; component sizes.  Each wave walks a contiguous chunk of points 64
// at a time; lanes whose points share a root add up in registers, and the count of the root a
// wave is currently running through is carried across steps, so a component that spans a long
// stretch of points (a ground plane) costs one atomic per wave, not one per step.
__global__ __launch_bounds__(kBlock) void k_cell_flatten(const Frame* frames) {
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint32_t* __restrict__ pt_slot = F.pt_slot;
    uint32_t* __restrict__ parent = F.parent; uint32_t* __restrict__ pt_root = F.pt_root;
    uint32_t* __restrict__ comp_size = F.comp_size;
    if (cluster_stage_off(st)) return;
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
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        v[k] = 0; root[k] = 0;
        if (i < n) v[k] = kept_size(i, pt_root, hmin, comp_size, lo, hi, &root[k]);
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
            if (++spins > kSpinLimit) { atomicOr(&st->err, (uint32_t)kErrInternal); q = 1ull << 32; break; }
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
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < count) {
            const uint32_t p = pass == 0 ? i : idx_in[i];
            atomicAdd(&s_h[(point_label(p, pt_root, comp_rank, trash) >> shift) & mask], 1u);
        }
    }
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
    uint32_t pt[kItems], dig[kItems], lrank[kItems];
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        const bool active = i < count;
        uint32_t d = 0;
        pt[j] = 0;
        if (active) {
            pt[j] = pass == 0 ? i : idx_in[i];
            d = (point_label(pt[j], pt_root, comp_rank, trash) >> shift) & mask;
        }
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

// ------------------------------------------------------------------------------------------
// end of frame: results -> pinned host memory (zero-copy stores), hash table reset
// ------------------------------------------------------------------------------------------
struct HostOut {
    DevState* state;
    uint32_t* offsets;     // n_kept + 1
    uint32_t* indices;     // n_kept_pts, discovery order
    float* cloud;          // voxel cloud in st->out_layout
};

__global__ __launch_bounds__(kBlock) void k_emit(const Frame* frames, uint32_t npass_csr) {
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
    if (tid == 0) *out.state = *st;
    if (st->err) return;
    if (st->kind != kFrameVoxelOnly && !st->trivial) {
        const uint32_t k = st->n_kept, np = st->n_kept_pts;
        for (uint32_t i = tid; i <= k; i += nth) out.offsets[i] = d_offsets[i];
        for (uint32_t i = tid; i < np; i += nth) out.indices[i] = d_indices[i];
    }
    if (st->emit_cloud) {
        const bool pt = st->passthrough != 0;
        const float* __restrict__ px = pt ? st->src_x : cen_x;
        const float* __restrict__ py = pt ? st->src_y : cen_y;
        const float* __restrict__ pz = pt ? st->src_z : cen_z;
        const uint32_t m = st->m, layout = st->out_layout, stride = st->out_soa_stride;
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
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t i = i0 + k;
        keep[k] = false;
        if (i < n) {
            px[k] = x[i]; py[k] = y[i]; pz[k] = z[i];
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

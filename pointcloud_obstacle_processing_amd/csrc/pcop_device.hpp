// pcop_device.hpp — gfx950 device code of the obstacle hot path.
//
// All HBM/LDS index-and-scatter work (no MFMA).  One frame is a fixed sequence of launches:
//
//   voxel grid (replaces pcl::VoxelGrid::applyFilter, called at od.cpp:282-285)
//     k_frame_init            mailbox -> DevState
//     k_minmax, k_voxel_setup pcl::getMinMax3D; min_b/div_b/divb_mul, the overflow rule, the plan
//    streaming path (batches of frames; no scattered global access anywhere):
//     k_bin_hist              per 16384-point tile: counts of the coarse key bins (key >> cshift)
//     k_bin_colsum, k_bin_plan  bins -> buckets of consecutive bins, at most kBucketCap points each
//     k_bin_tilecnt           per (tile, bucket): the tile's points in the bucket (k_bin_scatter adds up the lower tiles itself)
//     k_bin_scatter           ONE stable partition of packed 12-byte xyz records by bucket
//     k_heavy_split           (only once a frame needed it) a coarse bin above a bucket, cut into buckets along the keys
//     k_bucket_reduce         per bucket: occupancy bits in LDS -> local voxel ids, stable LDS counting
//                             sort, one lane per voxel sums its run in order -> the bucket's centroids
//     k_voxel_gather          buckets' voxel counts -> output ranks, M; centroids to their ranks (and, in a
//                             chain frame on the dense cell grid, each marks its grid cell on the way)
//    sort path (single frames - the shorter chain of launches - and clouds the streaming path does not cover):
//     k_radix_hist / k_table_scan / k_radix_scatter   (x passes) stable LSD radix partition of
//                             packed xyz records by PCL's linear voxel index
//     k_voxel_reduce          run heads -> rank; in-order float32 sums -> centroids (which mark their grid cells)
//   cell grid (replaces the FLANN kd-tree built at od.cpp:791-792): cells of edge tol/1.99
//    dense grid (default): k_cgrid_setup, k_cell_mark, k_occ_scan, k_cell_count, k_cell_starts,
//                          k_cell_place   occupancy bits -> cell ids; points sorted by cell
//    hash table (fallback): k_cell_insert, k_slot_count, k_slot_apply, k_cell_place
//   clusters (replaces EuclideanClusterExtraction::extract, od.cpp:446-454)
//     k_cgrid_hook | k_cell_hook, k_cell_compress, k_cgrid_union (+ k_cgrid_fix: single frames) | k_cell_union
//                             connected components of the cell graph
//     k_cell_flatten          root per point, component sizes
//     k_kept_apply            size filter, discovery order, CSR offsets
//     k_csr_hist / k_table_scan / k_csr_scatter | k_csr_onepass (single frames)   stable partition of point indices by cluster
//     k_emit                  results -> pinned host memory; hash table reset for the next frame
//   next to the path (pcop_occupancy.hpp, pcop_neighbours.hpp, pcop_planes.hpp):
//     k_occ_filter, k_occ_finish            bounds filter + occupancy binning (od.cpp:175-269)
//     k_sor_setup, k_cmap_coarse, k_knn_mean, k_knn_far, k_sor_select   statistical outlier removal (od.cpp:316-340)
//     k_plane_count, k_plane_split          RANSAC plane removal (od.cpp:342-428)
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
#ifndef PCOP_DIAG_REUSE_PLAN
#define PCOP_DIAG_REUSE_PLAN 0   // 1: diagnostic build, see pcop_ctx::diag_reuse in pcop_capi.hip (never the product: it reuses a slot's bucket plan when the same resident cloud comes back)
#endif
constexpr int kItems = PCOP_ITEMS;               // records per thread in the partition kernels
constexpr int kTile = kBlock * kItems;           // records per partition block
constexpr int kMaxPasses = 4;                    // 4 x 8 bits = 3 x 11 bits >= 32-bit keys
constexpr int kScanItems = 4;                    // elements per thread in flag-scan kernels
constexpr int kScanTile = kBlock * kScanItems;
constexpr int kReduceBlocks = 256;               // grid of the min/max kernel: one block per CU
constexpr int kCellBias = 1 << 20;               // cell coordinates are stored biased, 21 bits per axis
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kVmapTile = 1024;            // occupancy words per block of k_occ_scan
// streaming voxel path
constexpr uint32_t kBinTile = 16384;            // points per block of k_bin_hist / k_bin_scatter: kBinTileSmall when that fills the chip better
constexpr uint32_t kBinTileSmall = 4096;
constexpr uint32_t kBinTileLarge = 32768;       // (the tile tables count in 16 bits: no tile above 2^15 points)
constexpr uint32_t kBinThreads = 1024;          // threads of k_bin_plan
#ifndef PCOP_HIST_THREADS
#define PCOP_HIST_THREADS 512
#endif
// Threads of k_bin_hist.  Two blocks fit a CU either way (64 KB of LDS each); with 1024 threads they take every wave slot of the CU and
// nothing of another context's batch can run beside the pass, with 512 half the slots stay free - and the pass itself is as fast
// (tools/probes/stream_read.hip: 59 us either way).
constexpr uint32_t kHistThreads = PCOP_HIST_THREADS;
constexpr uint32_t kBinBitsMin = 11, kBinBitsMax = 14;   // coarse key bins per frame: 2^11 .. 2^14 (a host hint; any value gives the same result)
constexpr uint32_t kBucketCap = 2048;           // points per bucket (k_bucket_reduce: 256 threads x 8 records, all in LDS)
constexpr uint32_t kBucketCells = 1u << 17;     // grid cells a bucket may span (bits of its LDS occupancy map)
constexpr uint32_t kHeavyMax = 4u * 2048u;       // points of a coarse bin that holds more than a bucket and is split by k_heavy_split (beyond: sort path)
constexpr uint32_t kHeavyGroups = 16;           // ... into at most this many buckets
constexpr uint32_t kHeavySub = 2048;            // ... along this many key sub-ranges
constexpr uint32_t kNbMin = 1024, kNbMax = 4096;   // buckets per frame a launch provides LDS for (grows on demand)
#ifndef PCOP_SCAT_ITEMS
#define PCOP_SCAT_ITEMS 8
#endif
constexpr uint32_t kScatThreads = 512, kScatItems = PCOP_SCAT_ITEMS, kScatChunk = kScatThreads * kScatItems;   // k_bin_scatter: a tile in chunks of 4096 (a tuning build may take 4 or 16 records a thread: tools/ab_build.sh)
constexpr uint32_t kCmapCells = 1u << 24;       // dense cell grid: largest grid (cells, margins included) with an occupancy map
constexpr uint32_t kCoarseBits = 1u << 20;      // room for the blocks of 4x4x4 cells of the largest grid (partial blocks at the ends of y and z included)
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
    __host__ __device__ DevPtr& operator=(T* q) { p = q; return *this; }   // (the host pass checks kernel bodies too)
    __host__ __device__ operator T*() const { return p; }
    __host__ __device__ T* get() const { return p; }
#endif
};
static_assert(sizeof(DevPtr<float>) == sizeof(float*), "DevPtr is layout-compatible with a pointer");

enum ErrBits : uint32_t {
    kErrNonFinite = 1u,    // NaN/Inf where the reference is undefined
    kErrReplanVoxel = 2u,  // voxel key needs more radix passes than were launched
    kErrReplanCsr = 8u,    // cluster-rank key needs more radix passes than were launched
    kErrRange = 16u,       // a cell coordinate leaves +-2^20 cells (|p| / tol too large)
    kErrInternal = 32u,    // a bounded wait inside a kernel ran out (never expected)
    kErrReplanPath = 64u,  // the streaming voxel path does not apply to this frame (too many buckets)
    kErrReplanVcap = 128u, // the streaming voxel path needs more buckets than the launch provided LDS for
    kErrReplanBins = 1024u,  // a coarse key bin holds more points than a bucket (or spans more cells): finer bins needed
    kErrSorFew = 2048u,      // statistical outlier removal on fewer than mean_k + 1 points (the reference reads past its search result)
    kErrReplanCpath = 256u,  // the dense cell grid does not apply to this frame (grid too large)
    kErrReplanCtiles = 512u, // the occupancy scan of the dense cell grid was launched with too few tiles
    kErrReplanBox = 4096u,   // keys on the context's box: a point lies outside the box the keys were formed with (the box of earlier frames)
    kErrStale = 8192u,       // dense cell grid: an occupancy flag no point of this frame set (scratch a given-up frame left behind): the host scrubs the flags and runs the batch again
    kErrOrder = 16384u,      // the streamed partition's self-check: the two ways of ranking disagree (LDS atomics not lane-ordered after all)
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
    uint32_t voxel_stream, nb_cap;            // streaming voxel path requested; buckets the launch has LDS for
    uint32_t cell_grid, gbits;                // dense cell grid requested (else: the hash table of cells); log2 of the coarse bins wanted
    uint32_t tile, heavy_ok;                  // streaming voxel path: points per tile (kBinTile or kBinTileSmall); k_heavy_split is in the launch sequence
    DevPtr<const float> src_x, src_y, src_z;
    DevPtr<const uint32_t> n_dev;             // when set: the point count lives on the device (at most n): the cloud a previous kernel kept
    double sor_thr;                           // statistical outlier removal: mean + stddev_mul * sigma, formed by the host between the two halves of the node chain
    // keys on the context's box: the voxel box the keys are formed with (a superset of every earlier frame's)
    uint32_t box_on, pad0;                    // box_on: 2 the keys of this batch are formed against the host's box (no k_minmax)
    int32_t box_lo[3];
    uint32_t box_dim[3];
    uint32_t spin_limit, pad1;                // bounded waits of the publish / look-back kernels: polls before a block gives up
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
    uint32_t tile;          // streaming voxel path: points per tile of k_bin_hist / k_bin_scatter
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
    uint32_t gbits;         // coarse bins = 2^gbits (<= the host's hint, <= the key bits)
    uint32_t cshift;        // coarse bin of a voxel key = key >> cshift
    uint32_t nbuckets;
    uint32_t nb_cap;        // buckets the launch has LDS for
    // dense cell grid
    uint32_t cgrid_req;     // requested by the host
    uint32_t cpath;         // 1: this frame's cell stage runs on the dense grid
    uint32_t cmap_words;    // occupancy words in use
    uint32_t c_rowbits;     // bits per x-row (multiple of 32), margins included
    uint32_t c_plane;       // bits per z-plane
    int32_t  c0[3];         // cell coordinate of grid position 0 (margin included)
    uint32_t c_dim[3];      // grid size per axis, margins included
    uint32_t maxbin;        // streaming voxel path: points in the fullest coarse bin (the host tunes gbits with it)
    // keys on the context's box
    uint32_t box_nonfinite; // non-finite points the pass that took the bounds dropped
    uint32_t pad1;
    uint32_t spin_limit;    // polls a block may make on a lower block's publish word before it gives the frame up (kErrInternal)
    uint32_t box_tile;      // keys on the host's box: points per tile of the pass that took the bounds (one MinMaxPartial each)
    uint32_t sbox;          // 1: the voxel keys of this frame are formed against the host's box (no bounds pass); k_emit puts PCL's own
                            // grid here when the keys are done with
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
    DevPtr<uint16_t> ctab;            // [tile][coarse bin]: points of the tile in the bin
    DevPtr<uint16_t> bin2bucket;      // [coarse bin]: bucket of the bin
    DevPtr<uint4> binfo;              // [bucket]: {first record, records, first voxel key, voxel keys spanned}
    DevPtr<uint32_t> binsum;          // [coarse bin]: points of the frame in the bin
    DevPtr<uint16_t> tcnt;            // [bucket][tstride]: points of the tile in the bucket
    uint32_t tstride, cap;            // tiles a context provides for, rounded up to a multiple of 8; points a slot provides for
    DevPtr<float> h_aux;              // pinned host: per-point values a call hands to the host (mean neighbour distances)
    DevPtr<uint32_t> bucket_pub;      // [bucket]: its voxel count
    DevPtr<uint4> hsub;               // [bucket][kHeavyGroups]: the buckets a heavy bucket was split into (binfo's fields)
    DevPtr<uint32_t> hparts;          // [bucket]: how many
    // dense cell grid (null when the context was created without it)
    DevPtr<uint8_t> cflag;            // one byte per grid cell: occupied?  all zero between frames
    DevPtr<uint2> cmap;               // per 32 cells: {occupancy bits, occupied cells before this word}
    DevPtr<uint32_t> ctile;           // scan tiles: popcount + 1 once known
    DevPtr<uint32_t> ccoarse;         // one bit per block of 4x4x4 cells: any of them occupied?  (k-NN pass: far neighbours)
    DevPtr<uint32_t> counters;        // [0]: cells k_cgrid_union listed for k_cgrid_fix (k_cell_starts zeroes it)
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

// Inclusive scan of one value per lane across the wave, by data-parallel-primitive moves (no LDS traffic, unlike
// __shfl_up): shifts by 1, 2, 4, 8 inside each row of 16 lanes, then the last lane of rows 0 and 2 is added to rows 1
// and 3, then the last lane of row 1 to rows 2 and 3.  A lane without a source adds 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v = dpp_add<0x111, 0xF>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), kWave - 1);
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

// The same for a block of NW waves (s_w: NW words of LDS).
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan_n(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const uint32_t incl = wave_incl_scan(v);
    const uint32_t w = threadIdx.x / kWave;
    __syncthreads();
    if (lane_id() == kWave - 1) s_w[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
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

// ---- predicates and the cell marker shared by the voxel and the cell stage -------------------------------------------
__device__ __forceinline__ bool cluster_stage_off(const DevState* st) {
    return st->err || st->trivial || st->kind == kFrameVoxelOnly;
}
__device__ __forceinline__ bool cgrid_off(const DevState* st) { return cluster_stage_off(st) || !st->cpath; }

// Grid position of a point of the cell stage; its cell's flag byte set.  0 (and an error bit) for a point the grid cannot take.
struct CellMarker {
    double ig;
    long long c0x, c0y, c0z, d0, d1, d2;
    uint32_t rowbits, plane;
    __device__ __forceinline__ void init(const DevState* st) {
        ig = st->inv_cell;
        c0x = st->c0[0]; c0y = st->c0[1]; c0z = st->c0[2];
        d0 = st->c_dim[0]; d1 = st->c_dim[1]; d2 = st->c_dim[2];
        rowbits = st->c_rowbits; plane = st->c_plane;
    }
    __device__ __forceinline__ uint32_t operator()(DevState* st, uint8_t* __restrict__ cflag, float px, float py, float pz) const {
        if (!finite3(px, py, pz)) {   // radiusSearch on a non-finite point is undefined in PCL
            atomicOr(&st->err, (uint32_t)kErrNonFinite);
            return 0u;
        }
        const long long gx = (long long)floor((double)px * ig) - c0x, gy = (long long)floor((double)py * ig) - c0y,
                        gz = (long long)floor((double)pz * ig) - c0z;
        // two free cells on every side keep the whole 5x5x5 stencil inside the array
        if (gx < 2 || gy < 2 || gz < 2 || gx > d0 - 3 || gy > d1 - 3 || gz > d2 - 3) {
            atomicOr(&st->err, (uint32_t)kErrReplanCpath);
            return 0u;
        }
        const uint32_t pos = (uint32_t)gz * plane + (uint32_t)gy * rowbits + (uint32_t)gx;
        cflag[pos] = 1;
        return pos;
    }
};

}  // namespace pcop

// the kernels, by stage (each part opens namespace pcop itself)
#include "pcop_voxel.hpp"
#include "pcop_voxel_box.hpp"
#include "pcop_cells_hash.hpp"
#include "pcop_cells_grid.hpp"
#include "pcop_clusters.hpp"
#include "pcop_neighbours.hpp"
#include "pcop_occupancy.hpp"
#include "pcop_planes.hpp"

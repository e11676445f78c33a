// pcop_capi.hip — the C ABI of include/pcop.h on top of the gfx950 kernels in pcop_device.hpp.
//
// One context = one GPU + one HIP stream + the buffers of up to `max_frames` frames.  A batch of
// frames is one fixed sequence of kernel launches (blockIdx.y = frame) whose data-dependent sizes
// live in device memory (DevState) and whose per-frame inputs come through pinned mailboxes
// (FrameArgs), so the launch arguments are constant: the sequence is captured once into a
// hipGraph and replayed per batch.  The last kernel writes the results straight into pinned host
// memory; the host syncs once.  There is no CPU fallback: without a HIP device pcop_create fails.
#include "../../include/pcop.h"
#include "pcop_device.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sched.h>
#include <new>
#include <string>
#include <vector>

using namespace pcop;

namespace {

struct ClusterRef {
    int32_t size;
    int32_t id;   // discovery order
};
// pcl::comparePointClusters: a.indices.size() < b.indices.size()
inline bool compare_clusters(const ClusterRef& a, const ClusterRef& b) { return a.size < b.size; }

struct GraphKey {
    uint32_t kind, bucket, npass_voxel, npass_csr, nframes;
    uint32_t tile;                   // streaming voxel path: points per tile
    uint32_t stream, nb_cap, gbits;  // streaming voxel path: on?, buckets provided for (LDS, grids), log2 of the coarse bins (LDS)
    // (lds_ordered is fixed for the life of a context: not part of the key)
    uint32_t cgrid, ctiles;          // dense cell grid: on? (2: with the second hook round), occupancy tiles (grid)
    uint32_t box;                    // 3: keys against the host's box (no bounds pass); + 16: streaming path with its order check
    bool operator<(const GraphKey& o) const {
        if (box != o.box) return box < o.box;
        if (cgrid != o.cgrid) return cgrid < o.cgrid;
        if (ctiles != o.ctiles) return ctiles < o.ctiles;
        if (stream != o.stream) return stream < o.stream;
        if (tile != o.tile) return tile < o.tile;
        if (nb_cap != o.nb_cap) return nb_cap < o.nb_cap;
        if (gbits != o.gbits) return gbits < o.gbits;
        if (kind != o.kind) return kind < o.kind;
        if (bucket != o.bucket) return bucket < o.bucket;
        if (npass_voxel != o.npass_voxel) return npass_voxel < o.npass_voxel;
        if (npass_csr != o.npass_csr) return npass_csr < o.npass_csr;
        return nframes < o.nframes;
    }
};

// host-side view of one frame slot
struct Slot {
    Frame f{};                    // device pointers of the slot (DevState part unused on the host)
    float *in_x = nullptr, *in_y = nullptr, *in_z = nullptr;   // staged input cloud (SoA)
    uint8_t* d_raw = nullptr;     // raw input bytes (AoS / PointCloud2), grown on demand
    size_t d_raw_bytes = 0;
    FrameArgs* h_args = nullptr;  // pinned mailbox
    DevState* h_state = nullptr;
    int32_t *h_offsets_disc = nullptr, *h_indices_disc = nullptr;   // discovery order, as the GPU wrote them
    int32_t *h_offsets = nullptr, *h_indices = nullptr;             // PCL order
    float* h_vox = nullptr;
    float* h_xyzr = nullptr;
    float* h_clu = nullptr;       // the euc_clusters cloud (16-byte records), filled by pcop_cluster_cloud
    int32_t last_indices = 0;     // indices of the last ordered result
    std::vector<int32_t> order;   // order[c] = discovery id of output cluster c
    const float *src_x = nullptr, *src_y = nullptr, *src_z = nullptr;
    uint32_t n = 0;
    bool result_trivial = false;
    bool have_result = false;
};

}  // namespace

struct pcop_ctx {
    int device = -1;
    size_t cap = 0;               // max_points per frame
    size_t hash_cap = 0;          // slots of a frame's cell hash table
    uint32_t max_frames = 1;
    hipStream_t stream = nullptr;
    pcop_params params{};
    bool params_set = false;
    std::string err;

    std::vector<Slot> slots;
    Frame* d_frames = nullptr;    // device array [max_frames]
    std::vector<void*> dev_allocs, host_allocs, user_allocs;

    // current batch
    uint32_t nframes = 0;
    int is_dense = 1;
    uint32_t kind = kFrameChain;
    int out_layout = PCOP_LAYOUT_AOS12;
    int cloud_layout = PCOP_LAYOUT_AOS12;   // layout of the frame API's voxel cloud (pcop_set_cloud_layout)
    bool emit_cloud = false;
    bool in_flight = false;
    uint32_t npass_voxel = kMaxPasses, npass_csr = kMaxPasses;
    uint32_t frame_npass_csr = kMaxPasses;   // npass_csr of the batch in flight / last batch
    int radix_bits_target = 8;    // preferred digit width when choosing pass counts
    int maxbits = kNarrowBits;    // digit limit of the kernel variant in use (8: small LDS, 11: wide)
    bool use_graph = true;
    uint32_t cell_blocks = 2560;      // total blocks per launch of the wave-per-cell kernels (whole batch)
    uint32_t stride_blocks = 4096;    // total blocks per launch of the grid-strided point kernels
    uint32_t flatten_blocks = 128;    // blocks per frame of k_cell_flatten (each wave walks one contiguous chunk of points)
    uint32_t debug_skip_union = 0;    // diagnostics: 1 = no union kernel, 2 = union scans but never fixes up
    // streaming voxel path (pcop_voxel.hpp): on unless PCOP_VOXEL_PATH=sort; a frame it does not fit (more than
    // kNbMax buckets, or a coarse bin above a bucket's capacity even with the finest bins) is re-run on the sort
    // path and the context stays there until the leaf size changes
    bool voxel_stream = true, stream_ok = true;
    bool stream_single = false;      // PCOP_VOXEL_PATH=stream: the streaming voxel path also for a single frame (default: batches only)
    bool csr_onepass = true;         // PCOP_CSR_ONEPASS=0: the three-kernel form of the CSR partition also for a single pass
    bool csr_small = true;           // the last batch's clouds had few enough points for the one-kernel form (its look-back grows with the square of the tiles)
    uint32_t nb_cap = kNbMin;                               // buckets the launch provides LDS and blocks for (grows on demand)
    uint32_t gbits = 13;                                    // log2 of the coarse key bins (tuned from the fullest bin seen; any value gives the same result)
    bool gbits_fixed = false;                               // PCOP_BIN_BITS given: no tuning
    bool heavy_ok = false;           // streaming voxel path: k_heavy_split is in the launch sequence (a frame had a coarse bin above a bucket at the finest bins)
    // LDS atomic adds of one wave instruction return their old values in lane order on this device (checked by
    // k_lds_order_probe when the context is made; PCOP_LDS_ORDER=0 skips the check and keeps the peer-mask ranks)
    bool lds_ordered = false;
    // The context's voxel box: the union of the voxel grids of the frames it has seen since the leaf size was set, with a margin.
    // With the keys formed against it there is no bounds pass (k_minmax) - k_bin_hist (k_radix_hist of pass 0 on the sort path) takes
    // the true bounds on its way and checks every point against the box; a point outside: the batch is re-run with the bounds pass.
    // PCOP_VOXEL_BOX=0 keeps the bounds pass.
    bool sbox_enabled = true;
    bool sbox_batch = false;                 // the batch in flight runs the streaming path on the box
    bool box_skip = false;                   // the batch in flight left the box: its re-run takes the bounds pass
    bool stream_checked = false, stream_chk = false;   // the streaming path's partition was verified once against the peer-mask ranks / this batch does it
    uint32_t bin_tile_forced = 0;            // PCOP_BIN_TILE=4096|8192|16384|32768 (a multiple of the partition's chunk; the tile tables count in 16 bits)
    bool debug_sync = false;                 // PCOP_DEBUG_SYNC=1: see mark()
    bool debug_voxel_only = false;           // PCOP_DEBUG_VOXEL_ONLY=1: pcop_submit_batch runs the voxel stage only (measurements)
    bool debug_replan = false;               // PCOP_DEBUG_REPLAN=1: every re-plan says on stderr what it was about (read once, when the context is made)
    bool debug_order_mismatch = false;       // PCOP_DEBUG_ORDER_MISMATCH=1: that check fails once
    // A bounded device-side wait that runs out (kErrInternal: a block polled a lower block's publish word too long - only
    // possible when the GPU is shared so heavily that blocks of one launch are kept apart) costs a re-run, not the frame: the batch is
    // queued once more, by eager launches.  PCOP_DEBUG_SPIN0=1 makes the first batch of a context give up at its first poll (tests).
    bool internal_retry = false;             // the batch in flight is such a re-run
#if PCOP_DIAG_REUSE_PLAN
    // Diagnostic build only (tools/ab_libs.sh build reuse "-DPCOP_DIAG_REUSE_PLAN=1"): the UPPER BOUND of what a bucket plan carried from batch to batch can
    // give.  A batch that hands every slot the same resident cloud as the three clean batches before it - what bench.py does - skips the histogram
    // pass and the three kernels that turn it into a plan, and partitions by the tables those batches left in the slot (exact for the same cloud; a real stream needs slots).
    std::vector<const float*> diag_last_src;
    std::vector<uint32_t> diag_last_n;
    uint32_t diag_cfg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t diag_same = 0;
    bool diag_reuse = false;
#endif
    bool debug_spin0 = false;                // the next batch runs with a spin limit of zero
    bool debug_plane_mismatch = false;       // PCOP_DEBUG_PLANE_MISMATCH=1: the next plane split "disagrees" with the host's selection (tests of the error path)
    uint32_t internal_retries = 0;           // re-runs so far (diagnostics)
    uint64_t replans = 0;                    // batches re-run with other launch parameters / on another path (diagnostics)
    uint64_t stale_reruns = 0;               // batches re-run after a kernel found occupancy flags no point of the batch had set (kErrStale)
    uint32_t stale_reruns_batch = 0;         // ... of the batch at hand (a third time: an error, not a loop)
    bool box_valid = false;
    int32_t box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};   // voxel coordinates, inclusive
    std::vector<float> sor_sq;               // outlier removal: the squares of the mean distances
    const uint32_t* cur_n_dev = nullptr;     // the batch in flight takes frame 0's point count from the device (the cloud a filter kept)
    bool retry_to_caller = false;            // finish_batch hands a re-plan back to its caller instead of re-running the plain chain
    // dense cell grid: on unless PCOP_CELL_PATH=hash; a frame whose grid exceeds kCmapCells is re-run on the hash
    // table and the context stays there until the tolerance changes
    bool cell_grid = true, cgrid_ok = true;
    // The grid kernels give a lane to a cell and let it walk its cross pairs alone, which suits the usual few points
    // per cell; when cells are crowded (large tolerance / tiny leaf) the hash-table kernels, which put a whole wave
    // on a cell, are faster.  Chosen from the previous batch's points per cell; either choice gives the same result.
    bool crowded_cells = false;
    // A second hook round at tree level (k_cgrid_hook2) before the union: the first round leaves a tree per dozen cells, and the union
    // kernel merges trees pair by pair, by CAS (C3, 16 frames: union 49 -> 14 us for 23 us of hook2 + compress2, and shallower trees for
    // k_cell_flatten; C5: 251 k trees -> 20 k).  Two more launches: not for frames of a few thousand cells.  Chosen from the previous
    // batch's cells (PCOP_HOOK2=0|1 pins it); either choice gives the same result.
    bool hook2 = false, hook2_fixed = false;
    bool scat_big = true;            // PCOP_SCAT_BIG=0: the partition always in chunks of 4096 points (default: 8192 where the tile and the LDS allow)
    uint32_t ctiles = kCmapCells / 32u / kVmapTile;
    std::map<GraphKey, hipGraphExec_t> graphs;

    // bounds filter + occupancy binning (pcop_occupancy_filter): buffers made on first use
    float* occ_xyz = nullptr;                 // three planes of `cap` floats: the kept points
    unsigned long long* occ_counts = nullptr; // grid cells (64-bit counts)
    long long* occ_rowavg = nullptr;
    signed char* occ_grid = nullptr;
    uint32_t* occ_totals = nullptr;           // {kept, index errors, internal error}
    uint32_t* occ_partials = nullptr;         // kOccPartials partial grids of up to kOccLdsCells cells
    size_t occ_cells_cap = 0, occ_rows_cap = 0;
    uint32_t occ_n = 0;
    bool occ_valid = false;

    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> events;
    std::vector<const char*> ev_names;
    size_t ev_used = 0;
    pcop_stage_times times{};
};

namespace {

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[512];                                                                      \
            snprintf(buf_, sizeof(buf_), "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            (ctx)->err = buf_;                                                                   \
            (void)hipGetLastError();   /* reported here: not once more after somebody's next launch */ \
            return PCOP_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

template <class T>
int dev_alloc(pcop_ctx* c, T** p, size_t count) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
    c->dev_allocs.push_back(*p);
    return PCOP_OK;
}
template <class T>
int dev_alloc(pcop_ctx* c, DevPtr<T>* p, size_t count) {
    T* q = nullptr;
    const int rc = dev_alloc(c, &q, count);
    *p = q;
    return rc;
}
template <class T>
int host_alloc(pcop_ctx* c, T** p, size_t count) {
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault));
    c->host_allocs.push_back(*p);
    return PCOP_OK;
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// profiling: an event after every launch (eager mode only)
void mark(pcop_ctx* c, const char* name) {
    if (c->debug_sync) {   // PCOP_DEBUG_SYNC=1 (eager launches): drain the stream after every marked step and say which one it was
        const hipError_t e = hipStreamSynchronize(c->stream);
        fprintf(stderr, "pcop: [sync] %s: %s\n", name, hipGetErrorName(e));
    }
    if (!c->profiling) return;
    if (c->ev_used >= c->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->events.push_back(e);
        c->ev_names.push_back(name);
    }
    c->ev_names[c->ev_used] = name;
    hipEventRecord(c->events[c->ev_used], c->stream);
    ++c->ev_used;
}

uint32_t pow2_at_least(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

constexpr uint32_t kFarBlocks = 1024;   // k_knn_far: 4096 waves share the points the first k-NN pass left over
uint32_t stride_grid(uint32_t n) { return std::min(2048u, std::max(1u, div_up(n, kBlock))); }

uint32_t bucket_of(const pcop_ctx* c, uint32_t n) {   // launch-size bucket: next power of two, at most cap
    uint32_t b = 1024;
    while (b < n) b <<= 1;
    return (uint32_t)std::min<size_t>(b, std::max<size_t>(c->cap, 1024));
}

uint32_t hash_cap_for(uint32_t n) {
    uint32_t cap = 1024;
    while (cap < 2u * n && cap < 0x80000000u) cap <<= 1;
    return cap;
}

// Points per tile of the streaming voxel path: large tiles keep the per-tile tables small, small ones give a single
// frame (or a small batch) enough blocks to fill the chip.  Any value gives the same result.
uint32_t tile_for(const pcop_ctx* c, uint32_t nb, uint32_t nf) {
    if (c->bin_tile_forced) return c->bin_tile_forced;   // (PCOP_BIN_TILE, a tuning knob: points per block of the histogram / partition passes)
    // the largest tile that still gives every CU a block: fewer tiles are fewer rows of the tile tables and fewer bucket
    // cursors to set up (measured on C3, 16 frames: 2^15 against 2^14 points a tile: k_bin_scatter 108-112 against 116-117 us,
    // k_bin_hist 56 against 60-62 us, the bench line + 1-2 %)
    if ((uint64_t)nf * div_up(nb, kBinTileLarge) >= 256u) return kBinTileLarge;
    return (uint64_t)nf * div_up(nb, kBinTile) >= 256u ? kBinTile : kBinTileSmall;
}

bool grid_on(const pcop_ctx* c) { return c->cell_grid && c->cgrid_ok && !c->crowded_cells && c->kind != kFrameVoxelOnly; }
// The streaming voxel path wins by its traffic when frames share launches; a single frame is bound by launch count and
// latency, where the radix path's three passes are shorter (measured on every BASELINE config: 0.208 against 0.219 ms on
// C3).  So, unless PCOP_VOXEL_PATH says otherwise: batches stream, single frames sort.  Both give the same bytes.
bool stream_on(const pcop_ctx* c) {
    return c->voxel_stream && c->stream_ok && c->kind != kFrameClusterOnly && (c->nframes > 1 || c->stream_single);
}

// The streaming path without its bounds pass: any batch the streaming path serves, once there is a box.
bool sbox_on(const pcop_ctx* c) {
    if (!(stream_on(c) && c->sbox_enabled && c->box_valid && !c->box_skip)) return false;
    if (c->retry_to_caller || c->cur_n_dev) return false;
    uint64_t cells = 1;
    for (int a = 0; a < 3; ++a) cells *= (uint64_t)((int64_t)c->box_hi[a] - c->box_lo[a] + 1);
    return cells <= (1ull << 30);
}

// The sort path without its bounds pass (single frames, and whatever the streaming path does not take), once there is a box.
bool sbox_sort_on(const pcop_ctx* c) {
    if (!(c->sbox_enabled && c->box_valid && !c->box_skip && !stream_on(c) && c->kind != kFrameClusterOnly)) return false;
    if (c->retry_to_caller || c->cur_n_dev) return false;
    uint64_t cells = 1;
    for (int a = 0; a < 3; ++a) cells *= (uint64_t)((int64_t)c->box_hi[a] - c->box_lo[a] + 1);
    return cells <= (1ull << 30);
}

// The launch sequence of one batch.  nb = launch-size bucket (>= every frame's point count), nf =
// frames; all real counts are read on the device.  Everything here is capturable into a hipGraph.
enum LaunchParts : uint32_t { kPartVoxel = 1u, kPartCluster = 2u, kPartEmit = 4u, kPartAll = 7u,
                              kPartOwnGridSetup = 8u };   // the cluster part sets up its cell grid itself (something ran between the stages)

template <int MAXB>
void enqueue_launches_t(pcop_ctx* c, uint32_t kind, uint32_t nb, uint32_t nf, uint32_t parts) {
    hipStream_t s = c->stream;
    Frame* fr = c->d_frames;
    constexpr uint32_t kBinsGrid = (1u << MAXB) + 1u;
    const dim3 pg(std::max(1u, div_up(nb, kTile)), nf);        // partition tiles
    const dim3 sg(std::max(1u, div_up(nb, kScanTile)), nf);    // scan tiles over points
    // Grid-strided kernels: the counts they loop over (voxels, cells) are only known on the device and are
    // usually far below the bucket, and a block that finds nothing to do still costs its launch and its
    // prologue (kernarg -> Frame -> DevState loads).  So their grids are sized to fill the chip a few times
    // over for the whole batch, not to cover the bucket.
    const dim3 tg(std::min(stride_grid(nb), std::max(64u, c->stride_blocks / nf)), nf);
    const uint32_t cell_blocks = std::max(64u, c->cell_blocks / nf);
    // k_voxel_setup also writes DevState from the mailbox and, when the dense cell grid is on, sets that up as well
    // (both need nothing but the bounds), so a frame has no separate init and grid set-up launches.  Only a cluster-only
    // frame on the hash table of cells has no bounds pass: it keeps its k_frame_init.
    const uint32_t fold_grid = (kind != kFrameVoxelOnly && grid_on(c) && !(parts & kPartOwnGridSetup)) ? 4u : 0u;
    if (kind == kFrameClusterOnly && !grid_on(c) && (parts & kPartVoxel)) {
        k_frame_init<<<dim3(1, nf), 64, 0, s>>>(fr);
        mark(c, "k_frame_init");
    }
    if (kind != kFrameClusterOnly && (parts & kPartVoxel)) {
        const bool stream = stream_on(c);
        if (c->sbox_batch) {   // keys against the host's box: no bounds pass, k_bin_hist / k_radix_hist of pass 0 takes the bounds on its way
            k_box_setup<<<dim3(1, nf), kBlock, 0, s>>>(fr, c->npass_voxel, (uint32_t)MAXB, fold_grid, stream ? 2u : 3u);
            mark(c, "k_box_setup");
        } else {
            k_minmax<<<dim3(kReduceBlocks, nf), kBlock, 0, s>>>(fr, 0u);
            mark(c, "k_minmax");
            k_voxel_setup<<<dim3(1, nf), kBlock, 0, s>>>(fr, kReduceBlocks, c->npass_voxel, (uint32_t)MAXB, (stream ? 1u : 0u) | 2u | fold_grid);
            mark(c, "k_voxel_setup");
        }
        if (stream) {
            const uint32_t tile = tile_for(c, nb, nf);
            const dim3 bt(std::max(1u, div_up(nb, tile)), nf);       // tiles of the coarse histogram and of the partition
            const size_t bins = (size_t)1 << c->gbits;
#if PCOP_DIAG_REUSE_PLAN
            if (!(c->diag_reuse && c->sbox_batch)) {   // (the slot's tables of the last batch are this cloud's: see pcop_ctx::diag_reuse)
#endif
            k_bin_hist<<<bt, kHistThreads, bins * 4u, s>>>(fr);
            mark(c, "k_bin_hist");
            k_bin_colsum<<<dim3((uint32_t)div_up((uint32_t)bins, kWave), nf), kBlock, 0, s>>>(fr);
            mark(c, "k_bin_colsum");
            k_bin_plan<<<dim3(1, nf), kBinThreads, plan_lds_bytes((uint32_t)bins), s>>>(fr);
            mark(c, "k_bin_plan");
            k_bin_tilecnt<<<bt, kBlock, (bins + 8u) * 2u, s>>>(fr);
            mark(c, "k_bin_tilecnt");
#if PCOP_DIAG_REUSE_PLAN
            }
#endif
            // chunks of 8192 points where the tile and the LDS allow (see k_bin_scatter), else of 4096
            const bool big = c->scat_big && tile % (16u * kScatThreads) == 0u && c->nb_cap <= 4u * kScatThreads;
            auto scatter = [&](uint32_t alt, bool ordered) {
                constexpr uint32_t C16 = 16u * kScatThreads;
                if (ordered) {
                    if (c->nb_cap <= 2u * kScatThreads) { if (big) k_bin_scatter<2, true, 16><<<bt, kScatThreads, scatter_lds_bytes(2u * kScatThreads, C16), s>>>(fr, alt); else k_bin_scatter<2, true><<<bt, kScatThreads, scatter_lds_bytes(2u * kScatThreads), s>>>(fr, alt); }
                    else if (c->nb_cap <= 4u * kScatThreads) { if (big) k_bin_scatter<4, true, 16><<<bt, kScatThreads, scatter_lds_bytes(4u * kScatThreads, C16), s>>>(fr, alt); else k_bin_scatter<4, true><<<bt, kScatThreads, scatter_lds_bytes(4u * kScatThreads), s>>>(fr, alt); }
                    else k_bin_scatter<8, true><<<bt, kScatThreads, scatter_lds_bytes(8u * kScatThreads), s>>>(fr, alt);
                } else {
                    if (c->nb_cap <= 2u * kScatThreads) { if (big) k_bin_scatter<2, false, 16><<<bt, kScatThreads, scatter_lds_bytes(2u * kScatThreads, C16), s>>>(fr, alt); else k_bin_scatter<2, false><<<bt, kScatThreads, scatter_lds_bytes(2u * kScatThreads), s>>>(fr, alt); }
                    else if (c->nb_cap <= 4u * kScatThreads) { if (big) k_bin_scatter<4, false, 16><<<bt, kScatThreads, scatter_lds_bytes(4u * kScatThreads, C16), s>>>(fr, alt); else k_bin_scatter<4, false><<<bt, kScatThreads, scatter_lds_bytes(4u * kScatThreads), s>>>(fr, alt); }
                    else k_bin_scatter<8, false><<<bt, kScatThreads, scatter_lds_bytes(8u * kScatThreads), s>>>(fr, alt);
                }
            };
            scatter(0u, c->lds_ordered);
            mark(c, "k_bin_scatter");
            if (c->stream_chk) {   // first streamed batch: the same partition once more with ranks from peer masks, byte for byte
                scatter(1u, false);
                k_rec_compare<<<dim3(kReduceBlocks, nf), kBlock, 0, s>>>(fr, c->debug_order_mismatch ? 1u : 0u);
                mark(c, "k_rec_compare");
            }
            if (c->heavy_ok) {   // (only once a frame needed it: an ordinary cloud has no bin above a bucket)
                k_heavy_split<<<dim3(c->nb_cap, nf), kBlock, 0, s>>>(fr);
                mark(c, "k_heavy_split");
            }
            if (c->lds_ordered) k_bucket_reduce<true><<<dim3(div_up(c->nb_cap, kReduceBucketsPerBlock), nf), kBlock, 0, s>>>(fr);
            else k_bucket_reduce<false><<<dim3(div_up(c->nb_cap, kReduceBucketsPerBlock), nf), kBlock, 0, s>>>(fr);
            mark(c, "k_bucket_reduce");
            k_voxel_gather<<<dim3(c->nb_cap, nf), kBlock, 0, s>>>(fr);   // (also marks the grid cells of a chain frame: see marks_cells below)
            mark(c, "k_voxel_gather");
        } else {
            for (uint32_t p = 0; p < c->npass_voxel; ++p) {
                k_radix_hist<MAXB><<<pg, kBlock, 0, s>>>(fr, (int)p);
                mark(c, "k_radix_hist");
                k_table_scan<<<dim3(kBinsGrid, nf), kBlock, 0, s>>>(fr, 0, (int)p);
                mark(c, "k_table_scan");
                k_radix_scatter<MAXB><<<pg, kBlock, 0, s>>>(fr, (int)p);
                mark(c, "k_radix_scatter");
            }
            k_voxel_reduce<<<sg, kBlock, 0, s>>>(fr, c->npass_voxel, fold_grid ? 1u : 0u);   // (fold_grid: the cell grid is set up, the centroids mark their cells)
            mark(c, "k_voxel_reduce");
        }
    }
    if (kind != kFrameVoxelOnly && (parts & kPartCluster)) {
        const dim3 cg(std::max(1u, std::min(cell_blocks, div_up(nb, kBlock))), nf);   // lane = cell, grid-strided
        if (grid_on(c)) {
            if (parts & kPartOwnGridSetup) {
                k_cgrid_setup<<<dim3(1, nf), kBlock, 0, s>>>(fr);
                mark(c, "k_cgrid_setup");
            }
            if (kind == kFrameClusterOnly) {   // the bounding box of the cloud (a chain frame has it from the voxel stage)
                k_minmax<<<dim3(kReduceBlocks, nf), kBlock, 0, s>>>(fr, 0u);
                mark(c, "k_minmax");
                k_voxel_setup<<<dim3(1, nf), kBlock, 0, s>>>(fr, kReduceBlocks, c->npass_voxel, (uint32_t)MAXB, 2u | 4u);
                mark(c, "k_voxel_setup");
            }
            // a chain frame has its cells marked by the voxel stage's last kernel (same launch sequence, grid set up before
            // the voxel stage)
            const bool marks_cells = kind == kFrameChain && (parts & kPartVoxel) && !(parts & kPartOwnGridSetup);   // by k_voxel_gather / k_voxel_reduce
            if (!marks_cells) {
                k_cell_mark<<<tg, kBlock, 0, s>>>(fr);
                mark(c, "k_cell_mark");
            }
            k_occ_scan<<<dim3(c->ctiles, nf), kBlock, 0, s>>>(fr);
            mark(c, "k_cmap_scan");
            k_cell_count<<<tg, kBlock, 0, s>>>(fr);
            mark(c, "k_cell_count");
            const uint32_t hook2 = c->hook2 ? 1u : 0u;   // a second, tree-level hook round: frames with very many cells (see k_cgrid_hook2)
            k_cell_starts<<<sg, kBlock, 0, s>>>(fr, hook2);
            mark(c, "k_cell_starts");
            k_cell_place<<<tg, kBlock, 0, s>>>(fr);
            mark(c, "k_cell_place");
            k_cgrid_hook<<<cg, kBlock, 0, s>>>(fr);
            mark(c, "k_cgrid_hook");
            k_cell_compress<<<tg, kBlock, 0, s>>>(fr);
            mark(c, "k_cell_compress");
            if (hook2) {
                k_cgrid_hook2<<<cg, kBlock, 0, s>>>(fr);
                mark(c, "k_cgrid_hook2");
                k_cell_compress2<<<tg, kBlock, 0, s>>>(fr);
                mark(c, "k_cell_compress2");
            }
            k_cgrid_union<<<cg, kBlock, 0, s>>>(fr, nf == 1 ? 1u : 0u);
            mark(c, "k_cgrid_union");
            if (nf == 1) {   // one frame: where it pays, the cells at tree borders get a wave each (fix_by_wave)
                k_cgrid_fix<<<cg, kBlock, 0, s>>>(fr);
                mark(c, "k_cgrid_fix");
            }
        } else {
            const uint32_t hcap = hash_cap_for(nb);
            const dim3 hg(std::max(1u, hcap / kScanTile), nf);       // scan tiles over hash slots
            k_cell_insert<<<tg, kBlock, 0, s>>>(fr);
            mark(c, "k_cell_insert");
            k_slot_count<<<hg, kBlock, 0, s>>>(fr);
            mark(c, "k_slot_count");
            k_slot_apply<<<hg, kBlock, 0, s>>>(fr);
            mark(c, "k_slot_apply");
            k_cell_place<<<tg, kBlock, 0, s>>>(fr);
            mark(c, "k_cell_place");
            k_cell_hook<<<dim3(std::min(cell_blocks, std::max(1u, div_up(nb, kBlock / kHookGroup))), nf), kBlock, 0, s>>>(fr);
            mark(c, "k_cell_hook");
            k_cell_compress<<<tg, kBlock, 0, s>>>(fr);
            mark(c, "k_cell_compress");
            k_cell_union<<<dim3(std::min(cell_blocks, std::max(1u, div_up(nb, 2 * kWavesPerBlock))), nf), kBlock, 0, s>>>(fr);
            mark(c, "k_cell_union");
        }
        k_cell_flatten<<<dim3(c->flatten_blocks, nf), kBlock, 0, s>>>(fr);
        mark(c, "k_cell_flatten");
        k_kept_apply<<<sg, kBlock, 0, s>>>(fr, c->npass_csr, (uint32_t)MAXB);
        mark(c, "k_kept_apply");
        if (nf == 1 && c->npass_csr == 1 && MAXB == kNarrowBits && c->csr_onepass && c->csr_small) {   // one frame, at most 255 kept clusters: one kernel
            // (a batch keeps the three launches: tiles that wait for lower tiles hold wave slots the other contexts' kernels want)
            k_csr_onepass<<<pg, kBlock, 0, s>>>(fr);
            mark(c, "k_csr_onepass");
        } else for (uint32_t p = 0; p < c->npass_csr; ++p) {
            k_csr_hist<MAXB><<<pg, kBlock, 0, s>>>(fr, (int)p);
            mark(c, "k_csr_hist");
            k_table_scan<<<dim3(kBinsGrid, nf), kBlock, 0, s>>>(fr, 2, (int)p);
            mark(c, "k_table_scan");
            k_csr_scatter<MAXB><<<pg, kBlock, 0, s>>>(fr, (int)p);
            mark(c, "k_csr_scatter");
        }
    }
    if (parts & kPartEmit) {
        k_emit<<<tg, kBlock, 0, s>>>(fr, c->npass_csr);
        mark(c, "k_emit");
    }
}

void enqueue_launches(pcop_ctx* c, uint32_t kind, uint32_t nb, uint32_t nf, uint32_t parts = kPartAll) {
    if (c->maxbits <= kNarrowBits) enqueue_launches_t<kNarrowBits>(c, kind, nb, nf, parts);
    else enqueue_launches_t<kMaxBits>(c, kind, nb, nf, parts);
}

// Per-frame inputs into the pinned mailboxes; returns the launch bucket.
uint32_t fill_frame_args(pcop_ctx* c) {
    const pcop_params& p = c->params;
    const float inv_leaf = 1.0f / p.leaf;                         // Array4f::Ones() / leaf_size_
    const double radius = (double)p.tol;
    const float r2 = (float)(radius * radius);                    // KdTreeFLANN::radiusSearch
    const bool trivial = c->kind != kFrameVoxelOnly && !(r2 > 0.0f);
    uint32_t nmax = 0;
    for (uint32_t f = 0; f < c->nframes; ++f) {
        Slot& sl = c->slots[f];
        sl.result_trivial = trivial;
        sl.have_result = false;
        FrameArgs& a = *sl.h_args;
        a.n = sl.n; a.is_dense = c->is_dense ? 1u : 0u; a.kind = c->kind; a.trivial = trivial ? 1u : 0u;
        a.inv_leaf = inv_leaf; a.r2 = r2;
        a.inv_cell = trivial ? 1.0 : 1.99 / std::fabs((double)p.tol);
        a.min_cluster = (uint32_t)p.min_cluster; a.max_cluster = (uint32_t)p.max_cluster;
        a.emit_cloud = c->emit_cloud ? 1u : 0u; a.out_layout = (uint32_t)c->out_layout; a.out_soa_stride = sl.n; a.debug_skip_union = c->debug_skip_union;
        a.voxel_stream = stream_on(c) ? 1u : 0u; a.nb_cap = c->nb_cap; a.cell_grid = grid_on(c) ? 1u : 0u; a.gbits = c->gbits;
        a.tile = 0u; a.heavy_ok = c->heavy_ok ? 1u : 0u;   // (tile: set below, once the launch bucket is known)
        a.src_x = sl.src_x; a.src_y = sl.src_y; a.src_z = sl.src_z;
        a.n_dev = f == 0 ? c->cur_n_dev : nullptr; a.sor_thr = 0.0;
        a.spin_limit = (c->debug_spin0 && !c->internal_retry) ? 1u : kSpinLimit; a.pad1 = 0u;   // (PCOP_DEBUG_SPIN0: give up at the second poll)
        a.box_on = c->sbox_batch ? 2u : 0u; a.pad0 = 0u;
        for (int k = 0; k < 3; ++k) { a.box_lo[k] = c->box_lo[k]; a.box_dim[k] = (uint32_t)(c->box_hi[k] - c->box_lo[k] + 1); }
        nmax = std::max(nmax, sl.n);
    }
    c->frame_npass_csr = c->npass_csr;
    const uint32_t nb = bucket_of(c, nmax);
    for (uint32_t f = 0; f < c->nframes; ++f) c->slots[f].h_args->tile = tile_for(c, nb, c->nframes);
    return nb;
}

int enqueue_batch(pcop_ctx* c) {
    c->sbox_batch = sbox_on(c) || sbox_sort_on(c);
    c->stream_chk = c->kind != kFrameClusterOnly && stream_on(c) && c->lds_ordered && !c->stream_checked;
    const uint32_t nb = fill_frame_args(c);
#if PCOP_DIAG_REUSE_PLAN
    {
        const uint32_t cfg[8] = {c->gbits, c->nb_cap, tile_for(c, nb, c->nframes), c->heavy_ok ? 1u : 0u, c->lds_ordered ? 1u : 0u,
                                 (uint32_t)c->box_lo[0] ^ ((uint32_t)c->box_lo[1] << 10) ^ ((uint32_t)c->box_lo[2] << 20),
                                 (uint32_t)c->box_hi[0] ^ ((uint32_t)c->box_hi[1] << 10) ^ ((uint32_t)c->box_hi[2] << 20), nb};
        bool same = c->diag_last_src.size() == c->nframes && std::memcmp(cfg, c->diag_cfg, sizeof cfg) == 0;
        for (uint32_t f = 0; same && f < c->nframes; ++f) same = c->diag_last_src[f] == c->slots[f].src_x && c->diag_last_n[f] == c->slots[f].n;
        const bool plain = c->kind == kFrameChain && c->sbox_batch && stream_on(c) && !c->stream_chk && !c->internal_retry && !c->heavy_ok && c->cur_n_dev == nullptr;
        c->diag_same = (same && plain) ? c->diag_same + 1u : 0u;
        c->diag_last_src.resize(c->nframes); c->diag_last_n.resize(c->nframes);
        for (uint32_t f = 0; f < c->nframes; ++f) { c->diag_last_src[f] = c->slots[f].src_x; c->diag_last_n[f] = c->slots[f].n; }
        std::memcpy(c->diag_cfg, cfg, sizeof cfg);
        c->diag_reuse = c->diag_same >= 3u;
        for (uint32_t f = 0; f < c->nframes; ++f) c->slots[f].h_args->pad1 = c->diag_reuse ? 1u : 0u;
    }
#endif
    c->ev_used = 0;
    if (c->profiling || !c->use_graph || c->internal_retry) {
        mark(c, "start");
        enqueue_launches(c, c->kind, nb, c->nframes);
        HIPCHK(c, hipGetLastError());
    } else {
        const bool st_on = stream_on(c);
        const GraphKey key{c->kind, nb, st_on ? 0u : c->npass_voxel, c->npass_csr | ((c->csr_onepass && c->csr_small) ? 0x100u : 0u), c->nframes, st_on ? tile_for(c, nb, c->nframes) : 0u, st_on ? (c->heavy_ok ? 2u : 1u) : 0u, st_on ? c->nb_cap : 0u, st_on ? c->gbits : 0u,
                           grid_on(c) ? (c->hook2 ? 2u : 1u) : 0u, grid_on(c) ? c->ctiles : 0u, (c->sbox_batch ? 3u : 0u) | (c->stream_chk ? 16u : 0u)
#if PCOP_DIAG_REUSE_PLAN
                           | (c->diag_reuse ? 32u : 0u)
#endif
                           };
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            enqueue_launches(c, c->kind, nb, c->nframes);
            HIPCHK(c, hipStreamEndCapture(c->stream, &g));
            HIPCHK(c, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipGraphDestroy(g);
            it = c->graphs.emplace(key, ge).first;
        }
        HIPCHK(c, hipGraphLaunch(it->second, c->stream));
    }
    c->in_flight = true;
    return PCOP_OK;
}

uint32_t passes_for(const pcop_ctx* c, uint32_t key_bits) {
    const int target_bits = c->radix_bits_target;
    uint32_t p = (key_bits + (uint32_t)target_bits - 1) / (uint32_t)target_bits;
    const uint32_t pmin = (key_bits + (uint32_t)c->maxbits - 1) / (uint32_t)c->maxbits;
    p = std::max(p, pmin);
    return std::min<uint32_t>(std::max<uint32_t>(p, 1u), kMaxPasses);
}

void collect_times(pcop_ctx* c) {
    c->times.count = 0;
    if (!c->profiling || c->ev_used < 2) return;
    for (size_t i = 1; i < c->ev_used && c->times.count < PCOP_MAX_STAGES; ++i) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->events[i - 1], c->events[i]) != hipSuccess) ms = -1.0f;
        c->times.name[c->times.count] = c->ev_names[i];
        c->times.ms[c->times.count] = ms;
        ++c->times.count;
    }
}

// Cluster order of one finished frame: extract()'s std::sort(clusters.rbegin(), clusters.rend(),
// comparePointClusters) applied to the discovery-ordered clusters the GPU wrote.
void order_frame(pcop_ctx* c, Slot& sl, pcop_result* out) {
    const DevState* hs = sl.h_state;
    const uint32_t m = hs->m;
    const bool clustered = c->kind != kFrameVoxelOnly;
    uint32_t k = 0;
    if (clustered && !sl.result_trivial) k = hs->n_kept;
    if (clustered && sl.result_trivial) {
        // r2 == 0: radiusSearch returns nothing, every point is a cluster of one
        const pcop_params& p = c->params;
        const bool keep = (uint32_t)p.min_cluster <= 1u && 1u <= (uint32_t)p.max_cluster;
        const uint32_t c_n = c->kind == kFrameClusterOnly ? sl.n : m;
        k = keep ? c_n : 0;
        for (uint32_t i = 0; i < k; ++i) { sl.h_offsets_disc[i] = (int32_t)i; sl.h_indices_disc[i] = (int32_t)i; }
        sl.h_offsets_disc[k] = (int32_t)k;
    }
    std::vector<ClusterRef> refs(k);
    for (uint32_t i = 0; i < k; ++i) refs[i] = {sl.h_offsets_disc[i + 1] - sl.h_offsets_disc[i], (int32_t)i};
    std::sort(refs.rbegin(), refs.rend(), compare_clusters);
    sl.order.resize(k);
    int32_t pos = 0;
    sl.h_offsets[0] = 0;
    for (uint32_t i = 0; i < k; ++i) {
        const int32_t id = refs[i].id;
        sl.order[i] = id;
        const int32_t b = sl.h_offsets_disc[id], sz = refs[i].size;
        std::memcpy(sl.h_indices + pos, sl.h_indices_disc + b, (size_t)sz * 4);
        pos += sz;
        sl.h_offsets[i + 1] = pos;
    }
    sl.have_result = clustered;
    sl.last_indices = pos;
    if (out) {
        out->passthrough = (int32_t)hs->passthrough;
        out->num_voxels = (int32_t)m;
        out->num_clusters = (int32_t)k;
        out->num_indices = pos;
        out->offsets = sl.h_offsets;
        out->indices = sl.h_indices;
        out->voxel_xyz = (c->emit_cloud && m) ? sl.h_vox : nullptr;
    }
}

// A batch given up half-way may leave occupancy flags set: the streaming voxel path needs them all zero.
void scrub_flags(pcop_ctx* c, uint32_t nslots = 0xFFFFFFFFu) {   // (nslots: the frames of the batch at hand; default: every slot)
    uint32_t k = 0;
    for (Slot& sl : c->slots) {
        if (k++ >= nslots) break;
        if (sl.f.cflag.get()) hipMemsetAsync(sl.f.cflag.get(), 0, (size_t)kCmapCells, c->stream);
    }
    hipStreamSynchronize(c->stream);
}

// Waits for the batch (re-running it when a radix plan was too short) and puts every frame's
// clusters in PCL's order.  The kernels have already written everything into pinned host memory.
int finish_batch(pcop_ctx* c, pcop_result* out, uint32_t nout) {
    if (!c->in_flight) return PCOP_ERR_STATE;
    for (int attempt = 0;; ++attempt) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        uint32_t e = 0, kb_voxel = 0, kb_csr = 0;
        for (uint32_t f = 0; f < c->nframes; ++f) {
            const DevState* hs = c->slots[f].h_state;
            e |= hs->err;
            if (hs->err & kErrReplanVoxel) kb_voxel = std::max(kb_voxel, hs->plan_voxel.key_bits);
            if (hs->err & kErrReplanCsr) kb_csr = std::max(kb_csr, hs->plan_csr.key_bits);
        }
        // every error return leaves the occupancy flags all zero again (a frame given up half-way may have marked cells
        // beyond the tiles its scan was launched with)
        if (e & kErrNonFinite) { c->in_flight = false; scrub_flags(c); c->err = "non-finite coordinates where the reference is undefined"; return PCOP_ERR_NONFINITE; }
        if (e & kErrRange) { c->in_flight = false; scrub_flags(c); c->err = "a neighbour-grid cell coordinate exceeds +-2^20 (|coordinate| / tolerance too large)"; return PCOP_ERR_RANGE; }
        if (e & kErrInternal) {
            if (!c->internal_retry && !c->retry_to_caller) {   // once more, by eager launches (the flags a frame given up half-way left behind are cleared first)
                c->internal_retry = true; ++c->internal_retries;
                scrub_flags(c);
                int rc = enqueue_batch(c);
                if (rc) { c->internal_retry = false; return rc; }
                continue;
            }
            c->in_flight = false; c->internal_retry = false; scrub_flags(c); c->err = "internal: a bounded device-side wait ran out"; return PCOP_ERR_STATE;
        }
        if (e & kErrSorFew) { c->in_flight = false; scrub_flags(c); c->err = "fewer points than mean_k + 1"; return PCOP_ERR_STATE; }
        if (e & (kErrReplanVoxel | kErrReplanCsr | kErrReplanPath | kErrReplanVcap | kErrReplanBins | kErrReplanCpath | kErrReplanCtiles |
                 kErrReplanBox | kErrOrder | kErrStale)) {
            // (every adaptive piece may have to move once for a cloud unlike the last: bins, buckets, heavy bins, voxel path,
            //  pass counts, occupancy tiles, cell path, CSR form - each moves towards "larger" only, so the loop ends)
            ++c->replans;
#if PCOP_DIAG_REUSE_PLAN
            c->diag_same = 0; c->diag_last_src.clear();   // (no carried tables after a batch that had to be run again)
#endif
            // A frame given up half-way may have marked cells beyond the tiles its occupancy scan was launched with (kErrReplanCtiles),
            // and the re-run need not mark the same cells again: a batch that leaves the host's box for the bounds pass sets its cell
            // grid up from other bounds.  Flags nobody consumes would sit there until some later frame's grid covers them - cells
            // without points.  So no re-run starts on anything but clean flags.
            if (e & kErrStale) {   // a flag byte that was set before the batch began (a slot's "all zero between frames" scratch was not): every slot is scrubbed
                ++c->stale_reruns;
                scrub_flags(c);
                if (c->stale_reruns_batch++ >= 2u) { c->in_flight = false; c->stale_reruns_batch = 0; c->err = "internal: occupancy flags keep turning up set"; return PCOP_ERR_STATE; }
            } else scrub_flags(c, c->nframes);
            if (c->debug_replan) {
                fprintf(stderr, "pcop: replan, attempt %d, bits 0x%x, box %d;", attempt, e, c->sbox_batch ? 1 : 0);
                for (uint32_t f = 0; f < c->nframes; ++f) {
                    const DevState* hs = c->slots[f].h_state;
                    fprintf(stderr, " [f%u err 0x%x nb %u maxbin %u n_valid %u]", f, hs->err, hs->nbuckets, hs->maxbin, hs->n_valid);
                }
                fprintf(stderr, "\n");
            }
            if (c->sbox_batch && (e & kErrReplanBox)) {   // a point outside the box: this batch is re-run with the bounds pass
                c->box_skip = true;   // (whatever else a frame keyed against the wrong box reported says nothing about the cloud)
                e &= ~(uint32_t)(kErrReplanPath | kErrReplanBins | kErrReplanVcap);
            } else if (c->stream_chk && (e & kErrOrder)) {
                c->lds_ordered = false;          // the streamed partition differs between the two ways of ranking: peer masks from here on
                c->debug_order_mismatch = false;
            }
            if (attempt >= 16) { c->in_flight = false; scrub_flags(c); c->err = "replanning did not converge"; return PCOP_ERR_STATE; }
            if (kb_voxel) c->npass_voxel = std::max(c->npass_voxel, passes_for(c, kb_voxel));
            if (kb_csr) c->npass_csr = std::max(c->npass_csr, passes_for(c, kb_csr));
            if (e & kErrReplanCsr)
                for (uint32_t f = 0; f < c->nframes; ++f)
                    if (c->slots[f].h_state->c_n > kCsrOnepassMax) c->csr_small = false;   // (k_csr_onepass refused the frame)
            if (e & kErrReplanCpath) {
                c->cgrid_ok = false;             // the cell grid of this cloud is too large for an occupancy map: hash table from here on
            } else if (e & kErrReplanCtiles) {
                for (uint32_t f = 0; f < c->nframes; ++f) {
                    const DevState* hs = c->slots[f].h_state;
                    if (hs->err & kErrReplanCtiles) c->ctiles = std::max(c->ctiles, std::min(kCmapCells / 32u / kVmapTile, pow2_at_least(div_up(hs->cmap_words, kVmapTile)) * 2u));
                }
            }
            if ((e & kErrReplanBins) && (c->gbits >= kBinBitsMax || c->gbits_fixed)) {   // the finest bins are still too coarse:
                if (!c->heavy_ok && !(e & kErrReplanPath)) { c->heavy_ok = true; e &= ~(uint32_t)kErrReplanBins; }   // bins above a bucket get split (k_heavy_split)
                else e |= kErrReplanPath;
            }
            if (e & kErrReplanPath) {
                c->stream_ok = false;            // this cloud does not fit the streaming voxel path: sort path from here on
                c->npass_voxel = kMaxPasses;     // (its own plan check then trims the pass count)
            } else if (e & kErrReplanBins) {
                uint32_t worst = 0;              // finer coarse bins: a bit halves a bin, so go straight to where the fullest one fits
                for (uint32_t f = 0; f < c->nframes; ++f)
                    if (c->slots[f].h_state->err & kErrReplanBins) worst = std::max(worst, c->slots[f].h_state->maxbin);
                uint32_t more = 1;
                while (more < kBinBitsMax - kBinBitsMin && (worst >> more) > kBucketCap) ++more;
                c->gbits = std::min(kBinBitsMax, c->gbits + more);
            } else if (e & kErrReplanVcap) {
                for (uint32_t f = 0; f < c->nframes; ++f) {
                    const DevState* hs = c->slots[f].h_state;
                    if (hs->err & kErrReplanVcap) c->nb_cap = std::max(c->nb_cap, std::min(kNbMax, pow2_at_least(hs->nbuckets)));   // 1024, 2048, 4096
                }
            }
            if (c->retry_to_caller) { c->in_flight = false; return 1; }   // (not a pcop_status: the caller runs its own sequence again)
            int rc = enqueue_batch(c);
            if (rc) return rc;
            continue;
        }
        break;
    }
    collect_times(c);
    c->in_flight = false;
    c->internal_retry = false; c->debug_spin0 = false; c->stale_reruns_batch = 0;
    if (c->stream_chk) c->stream_checked = true;
    c->box_skip = false;
    // the voxel box of the next batches: the union of the frames' own grids (every path leaves PCL's min_b / div_b in DevState),
    // widened to multiples of eight voxels with a margin, so that it settles after a few batches of a steady source
    if (c->kind != kFrameClusterOnly) {
        bool grow = false;
        int64_t lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = c->box_valid ? c->box_lo[a] : INT64_MAX; hi[a] = c->box_valid ? c->box_hi[a] : INT64_MIN; }
        for (uint32_t f = 0; f < c->nframes; ++f) {
            const DevState* hs = c->slots[f].h_state;
            if (hs->passthrough || !hs->n_valid) continue;
            for (int a = 0; a < 3; ++a) {
                const int64_t l = hs->min_b[a], h = (int64_t)hs->min_b[a] + hs->div_b[a] - 1;
                if (l < lo[a]) { lo[a] = l; grow = true; }
                if (h > hi[a]) { hi[a] = h; grow = true; }
            }
        }
        if (grow) {
            // The coarse bins are the top bits of the key, so a box that crosses a power of two of cells doubles every bin:
            // the margin is the widest of 8 / 4 / 2 / 1 voxels (multiples of as many) that keeps the key as short as a margin of one does.
            auto widened = [&](int sh, int64_t* l, int64_t* h, double* cells) {
                const int64_t q = 1ll << sh;
                *cells = 1.0;
                for (int a = 0; a < 3; ++a) {
                    l[a] = ((lo[a] - (sh ? q / 4 : 1)) >> sh) << sh;
                    h[a] = ((((hi[a] + (sh ? q / 4 : 1)) >> sh) + 1) << sh) - 1;
                    *cells *= (double)(h[a] - l[a] + 1);
                }
            };
            auto bits_of = [](double cells) { int b = 1; while (b < 62 && std::ldexp(1.0, b) < cells) ++b; return b; };
            int64_t l[3], h[3], lt[3], ht[3];
            double cells, cells_tight;
            widened(0, lt, ht, &cells_tight);
            for (int sh = 3; sh >= 0; --sh) {
                widened(sh, l, h, &cells);
                if (bits_of(cells) == bits_of(cells_tight)) break;
            }
            // The kernels form keys as floor(p * inv) - (float)box_lo: exact while the box's voxel coordinates are floats' integers, i.e. within
            // +-2^24 (beyond, (float)box_lo rounds and min_b = box_lo + i0 could be off by one); its spans then fit 32 bits with room to
            // spare.  A cloud that far from the origin (in voxels) keeps its bounds pass: no box.
            bool fits = true;
            for (int a = 0; a < 3; ++a) if (l[a] < -(1ll << 24) || h[a] > (1ll << 24)) fits = false;
            c->box_valid = fits;
            if (fits) for (int a = 0; a < 3; ++a) { c->box_lo[a] = (int32_t)l[a]; c->box_hi[a] = (int32_t)h[a]; }
        }
    }
    // adapt the pass counts of the next batch to what this one needed
    uint32_t kb_voxel = 0, kb_csr = 0;
    for (uint32_t f = 0; f < c->nframes; ++f) {
        const DevState* hs = c->slots[f].h_state;
        if (c->kind != kFrameClusterOnly && !hs->passthrough && hs->n_valid) kb_voxel = std::max(kb_voxel, hs->plan_voxel.key_bits);
        if (c->kind != kFrameVoxelOnly && !c->slots[f].result_trivial) kb_csr = std::max(kb_csr, hs->plan_csr.key_bits);
    }
    // (a frame keyed on the context's box plans its passes on the BOX's key bits - k_box_setup sees nothing else - which are at least the
    //  true grid's and often one more: the pass count never goes below what the box needs, or every single frame after a batch
    //  would raise kErrReplanVoxel and run twice)
    if (kb_voxel && c->sbox_enabled && c->box_valid) {
        uint64_t cells = 1;
        for (int a = 0; a < 3; ++a) cells *= (uint64_t)((int64_t)c->box_hi[a] - c->box_lo[a] + 1);
        uint32_t bb = 1;
        while (bb < 64u && (1ull << bb) < cells) ++bb;
        if (cells <= (1ull << 30)) kb_voxel = std::max(kb_voxel, bb);
    }
    if (kb_voxel) c->npass_voxel = passes_for(c, kb_voxel);
    if (kb_csr) c->npass_csr = passes_for(c, kb_csr);
    if (c->kind != kFrameVoxelOnly) {
        uint32_t cn = 0;
        for (uint32_t f = 0; f < c->nframes; ++f) cn = std::max(cn, c->slots[f].h_state->c_n);
        c->csr_small = cn <= kCsrOnepassMax;
    }
    if (c->kind != kFrameVoxelOnly) {
        uint64_t pts = 0, cells = 0;
        for (uint32_t f = 0; f < c->nframes; ++f) {
            const DevState* hs = c->slots[f].h_state;
            if (!c->slots[f].result_trivial) { pts += hs->c_n; cells += hs->n_cells; }
        }
        if (cells) {
            if (pts > 24 * cells) c->crowded_cells = true;
            else if (pts < 16 * cells) c->crowded_cells = false;
        }
        if (!c->hook2_fixed) {   // (with hysteresis, like the choice above)
            uint32_t most = 0;
            for (uint32_t f = 0; f < c->nframes; ++f) if (!c->slots[f].result_trivial) most = std::max(most, c->slots[f].h_state->n_cells);
            // (a frame on its own gains as well - tests/tools/latency_env.py PCOP_HOOK2 "0 1", one run: C3 0.209 -> 0.201 ms, C3 with L = 10 m 0.283 -> 0.242 ms -
            //  although the kernel sums say otherwise: the union's walk is a chain of dependent finds, and shallower trees shorten it)
            // Sparse cells (a point or two to a cell) get it on large grids only: C5's three million cells 1.49 -> 1.38 ms a frame, 2.44 -> 2.32 ms a batch
            // of two (tests/tools/stage_times.py, since the hook kernels load no priorities); C1's nine thousand 0.240 -> 0.243 ms.
            const bool filled = pts >= 3 * cells, filled_off = 2 * pts < 5 * cells;
            if ((most >= (1u << 12) && filled) || most >= (1u << 18)) c->hook2 = true;
            else if (most < (1u << 11) || (filled_off && most < (1u << 17))) c->hook2 = false;
        }
    }
    if (grid_on(c)) {
        uint32_t tiles = 0;
        for (uint32_t f = 0; f < c->nframes; ++f)
            if (c->slots[f].h_state->cpath) tiles = std::max(tiles, div_up(c->slots[f].h_state->cmap_words, kVmapTile));
        if (tiles) {
            const uint32_t want = std::min(kCmapCells / 32u / kVmapTile, pow2_at_least(tiles) * 2u);
            if (c->ctiles > 2u * want) c->ctiles = want;
        }
    }
    if (stream_on(c) && !c->gbits_fixed) {   // coarse bins of the next batch: the fullest bin between 1/8 and 1/2 of a bucket
        uint32_t maxbin = 0, seen = 0;
        for (uint32_t f = 0; f < c->nframes; ++f)
            if (c->slots[f].h_state->vpath) { maxbin = std::max(maxbin, c->slots[f].h_state->maxbin); ++seen; }
        if (seen) {
            if (maxbin > kBucketCap / 2u && c->gbits < kBinBitsMax) ++c->gbits;
            else if (maxbin < kBucketCap / 8u && c->gbits > kBinBitsMin) --c->gbits;
        }
    }
    for (uint32_t f = 0; f < c->nframes; ++f) order_frame(c, c->slots[f], (out && f < nout) ? out + f : nullptr);
    return PCOP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// RANSAC plane removal, host side (the GPU passes are in pcop_planes.hpp).  What pcl::SACSegmentation does around its two
// passes over the cloud is sequential and tiny: a fixed-seed generator picks three points, a plane goes through them, the
// inliers of the best plane are refitted by least squares.  Same float operations, in the same order, as the reference's
// libraries perform them (the open choices are listed with the CPU restatement, oracle/plane_restatement.cpp).
// ---------------------------------------------------------------------------------------------------------------------
struct PlaneRng {   // boost::mt19937 seeded with 12345u (SampleConsensusModel with random = false); rnd() = uniform_int<>(0, INT_MAX) = word / 2
    uint32_t s[624];
    int idx;
    PlaneRng() {
        s[0] = 12345u;
        for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    int rnd() {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7FFFFFFFu);
                s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
            }
            idx = 0;
        }
        uint32_t y = s[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
        return (int)(y >> 1);
    }
};

struct HostCloud {   // the cloud of the current round on the host: three planes of `n` floats
    const float *x, *y, *z;
    size_t n;
};

inline bool host_plane_within(const float c[4], float x, float y, float z, double thr) {
    const float d = ((c[0] * x + c[1] * y) + c[2] * z) + c[3];
    return std::fabs((double)d) < thr;
}

// SampleConsensusModelPlane::computeModelCoefficients (false: the three points are collinear by its test)
bool plane_through(const HostCloud& h, const int smp[3], float c[4]) {
    const float p0[3] = {h.x[smp[0]], h.y[smp[0]], h.z[smp[0]]}, p1[3] = {h.x[smp[1]], h.y[smp[1]], h.z[smp[1]]}, p2[3] = {h.x[smp[2]], h.y[smp[2]], h.z[smp[2]]};
    const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2], bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    const float r0 = ax / bx, r1 = ay / by, r2 = az / bz;
    if (r0 == r1 && r2 == r1) return false;
    c[0] = ay * bz - az * by;
    c[1] = az * bx - ax * bz;
    c[2] = ax * by - ay * bx;
    c[3] = 0.0f;
    const float nrm = std::sqrt(((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) + c[3] * c[3]);
    c[0] /= nrm; c[1] /= nrm; c[2] /= nrm; c[3] /= nrm;
    c[3] = -1.0f * (((c[0] * p0[0] + c[1] * p0[1]) + c[2] * p0[2]) + c[3] * 1.0f);
    return true;
}

// SampleConsensusModelPerpendicularPlane::isModelValid
bool plane_valid(const float c[4], const pcop_plane_params& p) {
    if (!(p.eps_angle > 0.0)) return true;
    const double d = (double)((c[0] * p.axis[0] + c[1] * p.axis[1]) + c[2] * p.axis[2]);
    const double n1 = std::sqrt((double)((p.axis[0] * p.axis[0] + p.axis[1] * p.axis[1]) + p.axis[2] * p.axis[2]));
    const double n2 = std::sqrt((double)((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]));
    double rad = d / (n1 * n2);
    if (rad < -1.0) rad = -1.0; else if (rad > 1.0) rad = 1.0;
    double ang = std::fabs(std::acos(rad));
    ang = std::min(ang, M_PI - ang);
    return !(ang > p.eps_angle);
}

void plane_roots2(float b, float c, float r[3]) {   // pcl::computeRoots2
    r[0] = 0.0f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    const float sd = std::sqrt(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

void plane_roots3(const float m[3][3], float r[3]) {   // pcl::computeRoots
    const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] -
                     m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
    const float c2 = m[0][0] + m[1][1] + m[2][2];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) { plane_roots2(c2, c1, r); return; }
    const float inv3 = 1.0f / 3.0f, sqrt3 = std::sqrt(3.0f);
    const float c2_3 = c2 * inv3;
    float a_3 = (c1 - c2 * c2_3) * inv3;
    if (a_3 > 0.0f) a_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_3 * (2.0f * c2_3 * c2_3 - c1));
    float q = half_b * half_b + a_3 * a_3 * a_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = std::sqrt(-a_3);
    const float theta = std::atan2(std::sqrt(-q), half_b) * inv3;
    const float ct = std::cos(theta), st = std::sin(theta);
    r[0] = c2_3 + 2.0f * rho * ct;
    r[1] = c2_3 - rho * (ct + sqrt3 * st);
    r[2] = c2_3 - rho * (ct - sqrt3 * st);
    if (r[0] >= r[1]) std::swap(r[0], r[1]);
    if (r[1] >= r[2]) { std::swap(r[1], r[2]); if (r[0] >= r[1]) std::swap(r[0], r[1]); }
    if (r[0] <= 0.0f) plane_roots2(c2, c1, r);
}

// SampleConsensusModelPlane::optimizeModelCoefficients over the inliers of `c_in` (in index order): nine sequential float
// sums (computeMeanAndCovarianceMatrix), the eigenvector of the smallest eigenvalue (pcl::eigen33), d through the centroid.
void plane_refit(const HostCloud& h, const float c_in[4], const pcop_plane_params& p, double thr, float c_out[4]) {
    std::memcpy(c_out, c_in, 16);
    if (!plane_valid(c_in, p)) return;
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t cnt = 0;
    for (size_t i = 0; i < h.n; ++i) {
        const float x = h.x[i], y = h.y[i], z = h.z[i];
        if (!host_plane_within(c_in, x, y, z, thr)) continue;
        acc[0] += x * x; acc[1] += x * y; acc[2] += x * z; acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
        acc[6] += x; acc[7] += y; acc[8] += z;
        ++cnt;
    }
    if (cnt <= 3) return;
    const float fc = (float)cnt;
    for (float& a : acc) a /= fc;
    float m[3][3];
    m[0][0] = acc[0] - acc[6] * acc[6]; m[0][1] = acc[1] - acc[6] * acc[7]; m[0][2] = acc[2] - acc[6] * acc[8];
    m[1][1] = acc[3] - acc[7] * acc[7]; m[1][2] = acc[4] - acc[7] * acc[8]; m[2][2] = acc[5] - acc[8] * acc[8];
    m[1][0] = m[0][1]; m[2][0] = m[0][2]; m[2][1] = m[1][2];
    float scale = 0.0f;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) scale = std::max(scale, std::fabs(m[a][b]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
    float sm[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) sm[a][b] = m[a][b] / scale;
    float ev[3];
    plane_roots3(sm, ev);
    for (int a = 0; a < 3; ++a) sm[a][a] -= ev[0];
    auto cross = [](const float u[3], const float v[3], float o[3]) { o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0]; };
    float v1[3], v2[3], v3[3];
    cross(sm[0], sm[1], v1); cross(sm[0], sm[2], v2); cross(sm[1], sm[2], v3);
    auto sq = [](const float v[3]) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; };
    const float l1 = sq(v1), l2 = sq(v2), l3 = sq(v3);
    const float* best; float len;
    if (l1 >= l2 && l1 >= l3) { best = v1; len = l1; } else if (l2 >= l1 && l2 >= l3) { best = v2; len = l2; } else { best = v3; len = l3; }
    const float rl = std::sqrt(len);
    float c[4] = {best[0] / rl, best[1] / rl, best[2] / rl, 0.0f};
    c[3] = -1.0f * (((c[0] * acc[6] + c[1] * acc[7]) + c[2] * acc[8]) + 0.0f * 1.0f);
    if (plane_valid(c, p)) std::memcpy(c_out, c, 16);
}

// The loop of segment_plane_and_extract_indices on a cloud that lives on the device (planes d[0..2], n points) with a copy on
// the host (h: three planes in `hbuf`, compacted in place round by round).  Output planes alternate between the two scratch
// plane sets `sa` and `sb` (each three device pointers; neither may be the input).  On return *dx.. point to the planes
// that hold what is left and *n_left is its size.
struct PlaneScratch { float* p[3]; };
int plane_removal_loop(pcop_ctx* c, Slot& sl, const float* in_x, const float* in_y, const float* in_z, uint32_t n, float* hbuf,
                       const pcop_plane_params& prm, PlaneScratch sa, PlaneScratch sb, const float** ox, const float** oy, const float** oz,
                       uint32_t* n_left, std::vector<float>* coeffs, int32_t* found_all, std::vector<float>* last_plane = nullptr,
                       std::vector<float>* before_last = nullptr, const float** left_hx = nullptr) {
    const double thr = (double)prm.dist_thresh;   // setDistanceThreshold(double) from the node's float
    const uint32_t nr_points = n;
    const float *dx = in_x, *dy = in_y, *dz = in_z;
    float *hx = hbuf, *hy = hbuf + n, *hz = hbuf + 2 * (size_t)n;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(hx, dx, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hy, dy, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hz, dz, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *found_all = 1;
    uint32_t* d_counts = sl.f.bin_total.get();           // free outside a frame's partition passes: kPlaneModelsMax + 2 words used
    uint32_t* d_tiles = sl.f.tile_a.get();
    int round = 0;
    while ((double)n > 0.3 * (double)nr_points) {
        const HostCloud h{hx, hy, hz, n};
        // the candidate planes of RandomSampleConsensus::computeModel, in the order the generator yields them
        struct Event { bool model; float c[4]; };
        std::vector<Event> events;
        PlaneModels pm;
        std::memset(&pm, 0, sizeof(pm));
        if (n >= 3) {
            PlaneRng gen;
            std::vector<int> shuffled(n);
            for (uint32_t i = 0; i < n; ++i) shuffled[i] = (int)i;
            unsigned skipped = 0;
            while (pm.count < 51u && skipped < 500u) {
                int smp[3];
                bool good = false;
                for (unsigned it = 0; it < 1000 && !good; ++it) {   // getSamples: drawIndexSample until isSampleGood
                    for (uint32_t i = 0; i < 3; ++i) std::swap(shuffled[i], shuffled[i + (uint32_t)(gen.rnd() % (int)(n - i))]);
                    smp[0] = shuffled[0]; smp[1] = shuffled[1]; smp[2] = shuffled[2];
                    const float r0 = (hx[smp[1]] - hx[smp[0]]) / (hx[smp[2]] - hx[smp[0]]), r1 = (hy[smp[1]] - hy[smp[0]]) / (hy[smp[2]] - hy[smp[0]]),
                                r2 = (hz[smp[1]] - hz[smp[0]]) / (hz[smp[2]] - hz[smp[0]]);
                    good = (r0 != r1) || (r2 != r1);
                }
                if (!good) break;
                Event e;
                e.model = plane_through(h, smp, e.c);
                if (e.model) {
                    std::memcpy(pm.c[pm.count], e.c, 16);
                    pm.valid[pm.count] = plane_valid(e.c, prm) ? 1u : 0u;
                    ++pm.count;
                } else {
                    ++skipped;
                }
                events.push_back(e);
            }
        }
        // inlier counts of all of them in one pass over the cloud
        std::vector<uint32_t> counts(kPlaneModelsMax, 0u);
        if (pm.count) {
            HIPCHK(c, hipMemsetAsync(d_counts, 0, kPlaneModelsMax * 4, c->stream));
            k_plane_count<<<std::min(2048u, std::max(1u, div_up(n, kBlock))), kBlock, 0, c->stream>>>(dx, dy, dz, n, pm, thr, d_counts);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(counts.data(), d_counts, kPlaneModelsMax * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        // the loop itself, replayed: while (iterations_ < k && skipped_count < max_skip)
        int iterations = 0, best_count = -INT_MAX;
        double k = 1.0;
        const double log_probability = std::log(1.0 - 0.99), one_over = 1.0 / (double)n;
        unsigned skipped = 0;
        bool have = false;
        float best[4] = {0, 0, 0, 0};
        uint32_t mi = 0;
        for (size_t ei = 0; ei < events.size() && iterations < k && skipped < 500u; ++ei) {
            if (!events[ei].model) { ++skipped; continue; }
            const int cnt = (int)counts[mi++];
            if (cnt > best_count) {
                best_count = cnt; have = true; std::memcpy(best, events[ei].c, 16);
                const double w = (double)cnt * one_over;
                double p_no = 1.0 - std::pow(w, 3.0);
                p_no = std::max(std::numeric_limits<double>::epsilon(), p_no); p_no = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no);
                k = log_probability / std::log(p_no);
            }
            ++iterations;
            if (iterations > 50) break;
        }
        if (!have) { *found_all = 0; break; }   // "Couldn't estimate a planar model for the given dataset."
        float refined[4];
        plane_refit(h, best, prm, thr, refined);
        if (!plane_valid(refined, prm)) { *found_all = 0; break; }   // selectWithinDistance of an invalid model: no inliers
        // ExtractIndices (negative) on the device; the same selection on the host copy
        PlaneScratch& dst = (round & 1) ? sb : sa;
        HIPCHK(c, hipMemsetAsync(d_tiles, 0, (size_t)div_up(n, kScanTile) * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d_counts, 0, 8, c->stream));
        k_plane_split<<<div_up(n, kScanTile), kBlock, 0, c->stream>>>(dx, dy, dz, n, refined[0], refined[1], refined[2], refined[3], thr,
                                                                      dst.p[0], dst.p[1], dst.p[2], d_tiles, d_counts, d_counts + 1);
        HIPCHK(c, hipGetLastError());
        uint32_t kept = 0;
        std::vector<float> cur_plane;   // (a round that removes nothing leaves the last plane of the round before standing, as the reference's break does: od.cpp:379-383)
        std::vector<float> snapshot;   // the cloud this round starts from: the reference's cloud_f after planar_cloud.swap(cloud_f) (od.cpp:389-392)
        if (before_last) { snapshot.reserve(3 * (size_t)n); for (uint32_t i = 0; i < n; ++i) { snapshot.push_back(hx[i]); snapshot.push_back(hy[i]); snapshot.push_back(hz[i]); } }
        for (uint32_t i = 0; i < n; ++i) {
            if (host_plane_within(refined, hx[i], hy[i], hz[i], thr)) {
                if (last_plane) { cur_plane.push_back(hx[i]); cur_plane.push_back(hy[i]); cur_plane.push_back(hz[i]); }
                continue;
            }
            hx[kept] = hx[i]; hy[kept] = hy[i]; hz[kept] = hz[i];   // (in place: kept <= i)
            ++kept;
        }
        uint32_t dev[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(dev, d_counts, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->debug_plane_mismatch) { c->debug_plane_mismatch = false; dev[0] = kept + 1u; }   // (PCOP_DEBUG_PLANE_MISMATCH: the branch below, once)
        if (dev[1] || dev[0] != kept) { c->err = "internal: the plane split disagrees with the host's selection"; return PCOP_ERR_STATE; }
        if (kept == n) { *found_all = 0; break; }   // inliers->indices.size() == 0
        if (coeffs) coeffs->insert(coeffs->end(), refined, refined + 4);
        if (last_plane) last_plane->swap(cur_plane);
        if (before_last) before_last->swap(snapshot);
        dx = dst.p[0]; dy = dst.p[1]; dz = dst.p[2];
        n = kept;
        ++round;
    }
    *ox = dx; *oy = dy; *oz = dz; *n_left = n;
    if (left_hx) *left_hx = hx;   // (the host's copy of what is left: planes hx, hx + n0, hx + 2 n0 of the buffer handed in)
    return PCOP_OK;
}

int check_params(pcop_ctx* c, bool need_leaf, bool need_tol) {
    if (!c->params_set) { c->err = "pcop_set_params was not called"; return PCOP_ERR_STATE; }
    const pcop_params& p = c->params;
    if (need_leaf && !(p.leaf > 0.0f && std::isfinite(p.leaf))) { c->err = "downsample_size must be finite and > 0"; return PCOP_ERR_PARAM; }
    if (need_tol && !(std::isfinite(p.tol) && std::fabs(p.tol) < 1e18f)) { c->err = "euc_cluster_tolerance must be finite"; return PCOP_ERR_PARAM; }
    return PCOP_OK;
}

int ensure_raw(pcop_ctx* c, Slot& sl, size_t bytes) {
    if (bytes <= sl.d_raw_bytes) return PCOP_OK;
    if (sl.d_raw) { HIPCHK(c, hipFree(sl.d_raw)); sl.d_raw = nullptr; sl.d_raw_bytes = 0; }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&sl.d_raw), bytes));
    sl.d_raw_bytes = bytes;
    return PCOP_OK;
}

// Put an n-point cloud (host or device, any layout) where the kernels read it: SoA planes in HBM.
int stage_input(pcop_ctx* c, Slot& sl, const float* xyz, int layout, size_t n, int mem) {
    if (n > c->cap) { c->err = "n exceeds max_points of the context"; return PCOP_ERR_CAPACITY; }
    if (!xyz && n) return PCOP_ERR_ARG;
    if (layout != PCOP_LAYOUT_AOS16 && layout != PCOP_LAYOUT_SOA && layout != PCOP_LAYOUT_AOS12) return PCOP_ERR_ARG;
    sl.n = (uint32_t)n;
    if (mem == PCOP_MEM_DEVICE && layout == PCOP_LAYOUT_SOA) {   // read the caller's planes in place
        sl.src_x = xyz; sl.src_y = xyz + n; sl.src_z = xyz + 2 * n;
        return PCOP_OK;
    }
    sl.src_x = sl.in_x; sl.src_y = sl.in_y; sl.src_z = sl.in_z;
    if (n == 0) return PCOP_OK;
    if (layout == PCOP_LAYOUT_SOA) {   // host planes
        HIPCHK(c, hipMemcpyAsync(sl.in_x, xyz, n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(sl.in_y, xyz + n, n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(sl.in_z, xyz + 2 * n, n * 4, hipMemcpyHostToDevice, c->stream));
        return PCOP_OK;
    }
    const uint32_t step = layout == PCOP_LAYOUT_AOS16 ? 16u : 12u;
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(xyz);
    if (mem == PCOP_MEM_HOST) {
        int rc = ensure_raw(c, sl, n * step);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(sl.d_raw, xyz, n * step, hipMemcpyHostToDevice, c->stream));
        raw = sl.d_raw;
    }
    k_unpack_points<<<stride_grid((uint32_t)n), kBlock, 0, c->stream>>>(raw, (uint32_t)n, step, 0, 4, 8, 0, sl.in_x, sl.in_y, sl.in_z);
    HIPCHK(c, hipGetLastError());
    return PCOP_OK;
}

// The same for a sensor_msgs/PointCloud2 blob (the conversion of od.cpp:688-689 on the device): any point_step, row padding, field
// offsets, endianness.  The message is checked before a byte of it is read: offsets inside a point, rows that hold their points.
int stage_pointcloud2(pcop_ctx* c, Slot& sl, const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                      uint32_t off_x, uint32_t off_y, uint32_t off_z, int is_bigendian) {
    const size_t n = (size_t)width * height;
    if (n > c->cap) { c->err = "n exceeds max_points of the context"; return PCOP_ERR_CAPACITY; }
    if (n && !data) return PCOP_ERR_ARG;
    if (point_step < 4 || off_x > point_step - 4 || off_y > point_step - 4 || off_z > point_step - 4) { c->err = "PointCloud2: a field offset leaves the point"; return PCOP_ERR_ARG; }   // (written so that an offset near 2^32 cannot wrap past the test)
    if (n && row_step < (size_t)width * point_step) { c->err = "PointCloud2: row_step below width * point_step"; return PCOP_ERR_ARG; }
    sl.n = (uint32_t)n;
    sl.src_x = sl.in_x; sl.src_y = sl.in_y; sl.src_z = sl.in_z;
    if (n) {
        const size_t row_bytes = (size_t)width * point_step;
        int rc = ensure_raw(c, sl, n * point_step);
        if (rc) return rc;
        if (row_step == row_bytes) {
            HIPCHK(c, hipMemcpyAsync(sl.d_raw, data, n * point_step, hipMemcpyHostToDevice, c->stream));
        } else {
            HIPCHK(c, hipMemcpy2DAsync(sl.d_raw, row_bytes, data, row_step, row_bytes, height, hipMemcpyHostToDevice, c->stream));
        }
        k_unpack_points<<<stride_grid((uint32_t)n), kBlock, 0, c->stream>>>(sl.d_raw, (uint32_t)n, point_step, off_x, off_y, off_z,
                                                                           is_bigendian, sl.in_x, sl.in_y, sl.in_z);
        HIPCHK(c, hipGetLastError());
    }
    return PCOP_OK;
}

int set_device(pcop_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    return PCOP_OK;
}

int busy(pcop_ctx* c) {
    if (c->in_flight) { c->err = "a batch is already in flight on this context"; return PCOP_ERR_STATE; }
    return PCOP_OK;
}

int alloc_slot(pcop_ctx* c, Slot& sl, size_t n, size_t ntiles, size_t hcap, size_t nscan) {
    int rc;
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
    Frame& f = sl.f;
    TRY(dev_alloc(c, &sl.in_x, n)); TRY(dev_alloc(c, &sl.in_y, n)); TRY(dev_alloc(c, &sl.in_z, n));
    TRY(dev_alloc(c, &f.cen_x, n)); TRY(dev_alloc(c, &f.cen_y, n)); TRY(dev_alloc(c, &f.cen_z, n));
    TRY(dev_alloc(c, &f.rec_a, n)); TRY(dev_alloc(c, &f.rec_b, n));
    TRY(dev_alloc(c, &f.table, ntiles * kMaxBins)); TRY(dev_alloc(c, &f.table_x, ntiles * kMaxBins));
    TRY(dev_alloc(c, &f.bin_total, (size_t)kMaxBins));
    TRY(dev_alloc(c, &f.tile_a, nscan)); TRY(dev_alloc(c, &f.tile_b, nscan));
    TRY(dev_alloc(c, &f.partials, std::max<size_t>(kReduceBlocks, n / kTile + 2)));   // (keys on the host's box: one per tile of k_bin_hist / k_radix_hist)
    TRY(dev_alloc(c, &f.hkey, hcap)); TRY(dev_alloc(c, &f.hcount, hcap)); TRY(dev_alloc(c, &f.hmin, hcap));
    TRY(dev_alloc(c, &f.cell_start, hcap)); TRY(dev_alloc(c, &f.parent, hcap));
    TRY(dev_alloc(c, &f.comp_size, hcap)); TRY(dev_alloc(c, &f.comp_rank, hcap)); TRY(dev_alloc(c, &f.cell_list, n + 1));
    TRY(dev_alloc(c, &f.slot_meta, hcap)); TRY(dev_alloc(c, &f.cell_info, 2 * (n + 1)));
    TRY(dev_alloc(c, &f.pt_slot, n + 1)); TRY(dev_alloc(c, &f.pt_rank, n + 1)); TRY(dev_alloc(c, &f.pt_root, n + 1));
    TRY(dev_alloc(c, &f.idx_a, n + 1)); TRY(dev_alloc(c, &f.idx_b, n + 1)); TRY(dev_alloc(c, &f.d_offsets, n + 2));
    TRY(dev_alloc(c, &f.d_xyzr, n + 1));
    if (c->cell_grid) {
        TRY(dev_alloc(c, &f.cflag, (size_t)kCmapCells));
        HIPCHK(c, hipMemset(f.cflag.get(), 0, (size_t)kCmapCells));   // the kernels keep it all zero between frames
        TRY(dev_alloc(c, &f.cmap, (size_t)kCmapCells / 32u + 1u));
        TRY(dev_alloc(c, &f.ctile, (size_t)kCmapCells / 32u / kVmapTile));
        TRY(dev_alloc(c, &f.ccoarse, (size_t)kCoarseBits / 32u));
        TRY(dev_alloc(c, &f.counters, (size_t)16));
    }
    if (c->voxel_stream) {
        const size_t tiles = (n + kBinTileSmall - 1) / kBinTileSmall + 1;
        TRY(dev_alloc(c, &f.ctab, tiles << kBinBitsMax));
        TRY(dev_alloc(c, &f.bin2bucket, ((size_t)1 << kBinBitsMax) + 8));
        TRY(dev_alloc(c, &f.binfo, (size_t)kNbMax + 1));
        TRY(dev_alloc(c, &f.binsum, ((size_t)1 << kBinBitsMax) + 8));
        f.tstride = (uint32_t)((tiles + 7) / 8 * 8);
        TRY(dev_alloc(c, &f.tcnt, (size_t)f.tstride * kNbMax + 8));
        TRY(dev_alloc(c, &f.bucket_pub, (size_t)kNbMax));
        TRY(dev_alloc(c, &f.hsub, ((size_t)kNbMax + 1) * kHeavyGroups));
        TRY(dev_alloc(c, &f.hparts, (size_t)kNbMax + 1));
    }
    TRY(host_alloc(c, &sl.h_args, 1));
    TRY(host_alloc(c, &sl.h_state, 1));
    TRY(host_alloc(c, &sl.h_offsets_disc, n + 2)); TRY(host_alloc(c, &sl.h_indices_disc, n + 1));
    TRY(host_alloc(c, &sl.h_offsets, n + 2)); TRY(host_alloc(c, &sl.h_indices, n + 1));
    TRY(host_alloc(c, &sl.h_vox, 4 * n)); TRY(host_alloc(c, &sl.h_xyzr, 4 * (n + 1))); TRY(host_alloc(c, &sl.h_clu, 4 * (n + 1)));
#undef TRY
    f.args = sl.h_args;
    f.cap = (uint32_t)n;
    f.h_aux = sl.h_xyzr;
    f.h_state = sl.h_state;
    f.h_offsets = reinterpret_cast<uint32_t*>(sl.h_offsets_disc);
    f.h_indices = reinterpret_cast<uint32_t*>(sl.h_indices_disc);
    f.h_cloud = sl.h_vox;
    // the cell table starts empty; afterwards every batch's k_emit leaves it empty again
    HIPCHK(c, hipMemset(f.hkey, 0, hcap * 8));
    HIPCHK(c, hipMemset(f.hcount, 0, hcap * 4));
    HIPCHK(c, hipMemset(f.hmin, 0xFF, hcap * 4));
    std::memset(sl.h_state, 0, sizeof(DevState));
    std::memset(sl.h_args, 0, sizeof(FrameArgs));
    return PCOP_OK;
}

int submit_common(pcop_ctx* c, uint32_t kind, bool emit_cloud, int out_layout, int is_dense) {
    c->cur_n_dev = nullptr; c->retry_to_caller = false; c->box_skip = false;
    c->is_dense = is_dense; c->kind = kind; c->emit_cloud = emit_cloud; c->out_layout = out_layout;
    return enqueue_batch(c);
}

}  // namespace

extern "C" {

int pcop_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// "0-47,96-143" -> CPU numbers (the format of /sys/bus/pci/devices/*/local_cpulist); false on anything else
static bool parse_cpulist(const char* txt, std::vector<int>* cpus) {
    cpus->clear();
    const char* p = txt;
    while (*p && *p != '\n') {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) return false;
        long b = a;
        p = end;
        if (*p == '-') {
            ++p;
            b = strtol(p, &end, 10);
            if (end == p || b < a) return false;
            p = end;
        }
        if (b - a > 1 << 16) return false;
        for (long v = a; v <= b; ++v) cpus->push_back((int)v);
        if (*p == ',') ++p;
        else if (*p && *p != '\n') return false;
    }
    return true;
}

int pcop_debug_parse_cpulist(const char* txt, int32_t* cpus, int32_t cap) {
    std::vector<int> v;
    if (!txt || !parse_cpulist(txt, &v)) return PCOP_ERR_ARG;
    for (size_t i = 0; i < v.size() && (int32_t)i < cap && cpus; ++i) cpus[i] = v[i];
    return (int)v.size();
}

int pcop_bind_thread_near_device(int device_id) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device_id) != hipSuccess) {
        (void)hipGetLastError();   // (the runtime keeps the failure as the thread's last error: it must not surface after somebody's next launch)
        return PCOP_ERR_NO_DEVICE;
    }
    for (char* q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return 0;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    std::vector<int> local;
    if (!got || !parse_cpulist(line, &local) || local.empty()) return 0;
    cpu_set_t now, want;
    CPU_ZERO(&now); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof now, &now) != 0) return 0;
    int n = 0;
    for (int cpu : local)
        if (cpu < CPU_SETSIZE && CPU_ISSET(cpu, &now)) { CPU_SET(cpu, &want); ++n; }
    if (n < 4) return 0;                        // (almost) none of the GPU's CPUs is ours to run on (a cpuset): stay where we are rather than squeeze a feeder's threads onto one core
    if (sched_setaffinity(0, sizeof want, &want) != 0) return 0;
    return n;
}

const char* pcop_status_string(int s) {
    switch (s) {
        case PCOP_OK: return "ok";
        case PCOP_ERR_ARG: return "bad argument";
        case PCOP_ERR_PARAM: return "bad parameter";
        case PCOP_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case PCOP_ERR_HIP: return "HIP runtime error";
        case PCOP_ERR_CAPACITY: return "cloud larger than the context's max_points";
        case PCOP_ERR_NONFINITE: return "non-finite coordinates";
        case PCOP_ERR_RANGE: return "neighbour grid too large";
        case PCOP_ERR_STATE: return "bad call order";
        default: return "unknown status";
    }
}

int pcop_create_batch(pcop_ctx** out, int device_id, size_t max_points, int max_frames) {
    if (!out) return PCOP_ERR_ARG;
    *out = nullptr;
    if (max_points == 0 || max_points > 0x3FFFFFF0ull || max_frames < 1 || max_frames > 1024) return PCOP_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return PCOP_ERR_NO_DEVICE; }
    if (device_id < 0 || device_id >= ndev) return PCOP_ERR_NO_DEVICE;
    pcop_ctx* c = new (std::nothrow) pcop_ctx();
    if (!c) return PCOP_ERR_ARG;
    c->device = device_id;
    c->cap = max_points;
    c->max_frames = (uint32_t)max_frames;
    if (const char* e = getenv("PCOP_RADIX_BITS")) {
        const int b = atoi(e);
        if (b >= 4 && b <= kMaxBits) c->radix_bits_target = b;
    }
    c->maxbits = c->radix_bits_target <= kNarrowBits ? kNarrowBits : kMaxBits;
    if (const char* e = getenv("PCOP_NO_GRAPH")) c->use_graph = atoi(e) == 0;
    if (const char* e = getenv("PCOP_VOXEL_PATH")) { c->voxel_stream = std::string(e) != "sort"; c->stream_single = std::string(e) == "stream"; }
    if (const char* e = getenv("PCOP_CSR_ONEPASS")) c->csr_onepass = atoi(e) != 0;
    if (const char* e = getenv("PCOP_VOXEL_BOX")) c->sbox_enabled = atoi(e) != 0;
    if (const char* e = getenv("PCOP_DEBUG_SPIN0")) c->debug_spin0 = atoi(e) != 0;
    if (const char* e = getenv("PCOP_DEBUG_PLANE_MISMATCH")) c->debug_plane_mismatch = atoi(e) != 0;
    if (const char* e = getenv("PCOP_BIN_TILE")) { const uint32_t v = (uint32_t)atoi(e); if (v == 4096u || v == 8192u || v == 16384u || v == 32768u) c->bin_tile_forced = v; }
    if (const char* e = getenv("PCOP_DEBUG_SYNC")) { c->debug_sync = atoi(e) != 0; if (c->debug_sync) c->use_graph = false; }
    if (const char* e = getenv("PCOP_DEBUG_ORDER_MISMATCH")) c->debug_order_mismatch = atoi(e) != 0;
    if (const char* e = getenv("PCOP_DEBUG_REPLAN")) c->debug_replan = atoi(e) != 0;
    if (const char* e = getenv("PCOP_DEBUG_VOXEL_ONLY")) c->debug_voxel_only = atoi(e) != 0;
    if (const char* e = getenv("PCOP_CELL_PATH")) c->cell_grid = std::string(e) != "hash";
    if (const char* e = getenv("PCOP_BIN_BITS")) { c->gbits = (uint32_t)std::min<int>(kBinBitsMax, std::max<int>(kBinBitsMin, atoi(e))); c->gbits_fixed = true; }
    if (const char* e = getenv("PCOP_CELL_BLOCKS")) c->cell_blocks = (uint32_t)std::max(64, atoi(e));
    if (const char* e = getenv("PCOP_STRIDE_BLOCKS")) c->stride_blocks = (uint32_t)std::max(64, atoi(e));
    if (const char* e = getenv("PCOP_FLATTEN_BLOCKS")) c->flatten_blocks = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("PCOP_DEBUG_SKIP_UNION")) c->debug_skip_union = (uint32_t)atoi(e);
    if (const char* e = getenv("PCOP_HOOK2")) { c->hook2 = atoi(e) != 0; c->hook2_fixed = true; }
    if (const char* e = getenv("PCOP_SCAT_BIG")) c->scat_big = atoi(e) != 0;
    const size_t nbk = bucket_of(c, (uint32_t)max_points);           // launch sizes round up to this
    const size_t n = std::max<size_t>(max_points, nbk);
    const size_t ntiles = (n + kTile - 1) / kTile + 1;
    const size_t hcap = hash_cap_for((uint32_t)n);
    c->hash_cap = hcap;
    const size_t nscan = std::max((n + kScanTile - 1) / kScanTile, hcap / kScanTile) + 1;
    if (hipSetDevice(device_id) != hipSuccess) { (void)hipGetLastError(); delete c; return PCOP_ERR_NO_DEVICE; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PCOP_ERR_HIP; }
    int rc = PCOP_OK;
    c->slots.resize(c->max_frames);
    for (uint32_t f = 0; f < c->max_frames && !rc; ++f) rc = alloc_slot(c, c->slots[f], n, ntiles, hcap, nscan);
    if (!rc) rc = dev_alloc(c, &c->d_frames, c->max_frames);
    if (!rc) {
        std::vector<Frame> hf(c->max_frames);
        for (uint32_t f = 0; f < c->max_frames; ++f) hf[f] = c->slots[f].f;
        if (hipMemcpy(c->d_frames, hf.data(), sizeof(Frame) * c->max_frames, hipMemcpyHostToDevice) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) { c->err = "frame table upload failed"; rc = PCOP_ERR_HIP; }
    }
    if (!rc && c->voxel_stream) {   // may the ranks of the streaming path come straight from LDS atomics on this device?
        const char* e = getenv("PCOP_LDS_ORDER");
        if (!(e && atoi(e) == 0)) {
            uint32_t* d_bad = nullptr;
            uint32_t bad = 1;
            if (hipMalloc(reinterpret_cast<void**>(&d_bad), 4) == hipSuccess) {
                if (hipMemsetAsync(d_bad, 0, 4, c->stream) == hipSuccess) {
                    k_lds_order_probe<<<64, kBlock, 0, c->stream>>>(d_bad);
                    if (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) bad = 1;
                }
                hipFree(d_bad);
            }
            c->lds_ordered = bad == 0;
        }
    }
    if (rc) { fprintf(stderr, "pcop_create: %s\n", c->err.c_str()); pcop_destroy(c); return rc; }
    if (c->debug_replan) fprintf(stderr, "pcop: context: lds_ordered %d\n", c->lds_ordered ? 1 : 0);
    *out = c;
    return PCOP_OK;
}

int pcop_create(pcop_ctx** out, int device_id, size_t max_points) { return pcop_create_batch(out, device_id, max_points, 1); }

void pcop_destroy(pcop_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    hipDeviceSynchronize();   // nothing of this context may still be queued anywhere when its graphs and buffers go
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    for (void* p : c->dev_allocs) if (p) hipFree(p);
    for (Slot& sl : c->slots) if (sl.d_raw) hipFree(sl.d_raw);
    for (void* p : c->user_allocs) if (p) hipFree(p);
    for (void* p : c->host_allocs) if (p) hipHostFree(p);
    for (hipEvent_t e : c->events) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int pcop_max_frames(const pcop_ctx* c) { return c ? (int)c->max_frames : 0; }

int pcop_set_params(pcop_ctx* c, const pcop_params* p) {
    if (!c || !p) return PCOP_ERR_ARG;
    if (c->in_flight) { c->err = "parameters cannot change while a batch is in flight (its re-planning and ordering read them)"; return PCOP_ERR_STATE; }
    if (!c->params_set || p->leaf != c->params.leaf) { c->stream_ok = true; c->heavy_ok = false; c->box_valid = false; }   // a new leaf size is a new voxel grid
    if (!c->params_set || p->tol != c->params.tol) c->cgrid_ok = true;      // a new tolerance is a new cell grid
    c->params = *p;
    c->params_set = true;
    return PCOP_OK;
}

int pcop_get_params(const pcop_ctx* c, pcop_params* p) {
    if (!c || !p) return PCOP_ERR_ARG;
    *p = c->params;
    return PCOP_OK;
}

const char* pcop_last_error(const pcop_ctx* c) { return c ? c->err.c_str() : "null context"; }

int pcop_set_profiling(pcop_ctx* c, int enable) {
    if (!c) return PCOP_ERR_ARG;
    c->profiling = enable != 0;
    return PCOP_OK;
}

int pcop_get_stage_times(const pcop_ctx* c, pcop_stage_times* out) {
    if (!c || !out) return PCOP_ERR_ARG;
    *out = c->times;
    return PCOP_OK;
}

void* pcop_stream(const pcop_ctx* c) { return c ? (void*)c->stream : nullptr; }

int pcop_voxel_downsample(pcop_ctx* c, const float* xyz, int layout, size_t n, int is_dense, float* out_xyz,
                          size_t* out_m, int* passthrough) {
    if (!c || !out_m || !passthrough || (!out_xyz && n)) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    if ((rc = check_params(c, true, false))) return rc;
    if ((rc = set_device(c))) return rc;
    Slot& sl = c->slots[0];
    if ((rc = stage_input(c, sl, xyz, layout, n, PCOP_MEM_HOST))) return rc;
    c->nframes = 1;
    if ((rc = submit_common(c, kFrameVoxelOnly, true, layout, is_dense))) return rc;
    pcop_result r{};
    if ((rc = finish_batch(c, &r, 1))) return rc;
    *out_m = (size_t)r.num_voxels;
    *passthrough = r.passthrough;
    if (r.num_voxels == 0) return PCOP_OK;
    const size_t fl = layout == PCOP_LAYOUT_AOS16 ? 4 : 3;
    if (r.passthrough) {   // "output = *input_"
        std::memcpy(out_xyz, xyz, n * fl * sizeof(float));
        return PCOP_OK;
    }
    const size_t m = (size_t)r.num_voxels;
    if (layout == PCOP_LAYOUT_SOA) {
        for (int a = 0; a < 3; ++a) std::memcpy(out_xyz + (size_t)a * n, sl.h_vox + (size_t)a * n, m * 4);
    } else {
        std::memcpy(out_xyz, sl.h_vox, m * fl * 4);
    }
    return PCOP_OK;
}

int pcop_euclidean_clusters(pcop_ctx* c, const float* xyz, int layout, size_t m, int32_t* offsets, int32_t* indices,
                            int32_t* k) {
    if (!c || !offsets || !k || (!indices && m)) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    if ((rc = check_params(c, false, true))) return rc;
    if ((rc = set_device(c))) return rc;
    if ((rc = stage_input(c, c->slots[0], xyz, layout, m, PCOP_MEM_HOST))) return rc;
    c->nframes = 1;
    if ((rc = submit_common(c, kFrameClusterOnly, false, PCOP_LAYOUT_AOS12, 1))) return rc;
    pcop_result r{};
    if ((rc = finish_batch(c, &r, 1))) return rc;
    *k = r.num_clusters;
    std::memcpy(offsets, r.offsets, (size_t)(r.num_clusters + 1) * 4);
    if (r.num_indices) std::memcpy(indices, r.indices, (size_t)r.num_indices * 4);
    return PCOP_OK;
}

int pcop_submit_batch(pcop_ctx* c, const float* const* xyz, int layout, const size_t* n, int nframes, int is_dense, int mem) {
    if (!c || !xyz || !n) return PCOP_ERR_ARG;
    if (nframes < 1 || (uint32_t)nframes > c->max_frames) { c->err = "nframes exceeds max_frames of the context"; return PCOP_ERR_CAPACITY; }
    int rc = busy(c);
    if (rc) return rc;
    if ((rc = check_params(c, true, true))) return rc;
    if (mem != PCOP_MEM_HOST && mem != PCOP_MEM_DEVICE) return PCOP_ERR_ARG;
    if ((rc = set_device(c))) return rc;
    for (int f = 0; f < nframes; ++f)
        if ((rc = stage_input(c, c->slots[f], xyz[f], layout, n[f], mem))) return rc;
    c->nframes = (uint32_t)nframes;
    // (PCOP_DEBUG_VOXEL_ONLY=1, a measuring knob: batches stop after the voxel stage - what the cluster stage costs a pipelined run)
    return submit_common(c, c->debug_voxel_only ? kFrameVoxelOnly : kFrameChain, c->params.publish_clouds != 0, c->cloud_layout, is_dense);
}

int pcop_wait_batch(pcop_ctx* c, pcop_result* out, int nframes) {
    if (!c || nframes < 0) return PCOP_ERR_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    return finish_batch(c, out, (uint32_t)nframes);
}

int pcop_submit_xyz(pcop_ctx* c, const float* xyz, int layout, size_t n, int is_dense, int mem) {
    const float* ptrs[1] = {xyz};
    const size_t ns[1] = {n};
    return pcop_submit_batch(c, ptrs, layout, ns, 1, is_dense, mem);
}

int pcop_wait(pcop_ctx* c, pcop_result* out) { return pcop_wait_batch(c, out, out ? 1 : 0); }

int pcop_process_xyz(pcop_ctx* c, const float* xyz, int layout, size_t n, int is_dense, int mem, pcop_result* out) {
    int rc = pcop_submit_xyz(c, xyz, layout, n, is_dense, mem);
    if (rc) return rc;
    return pcop_wait(c, out);
}

int pcop_process_pointcloud2(pcop_ctx* c, const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step,
                             uint32_t row_step, uint32_t off_x, uint32_t off_y, uint32_t off_z, int is_bigendian,
                             int is_dense, pcop_result* out) {
    if (!c) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    if ((rc = check_params(c, true, true))) return rc;
    if ((rc = set_device(c))) return rc;
    if ((rc = stage_pointcloud2(c, c->slots[0], data, width, height, point_step, row_step, off_x, off_y, off_z, is_bigendian))) return rc;
    c->nframes = 1;
    if ((rc = submit_common(c, kFrameChain, c->params.publish_clouds != 0, c->cloud_layout, is_dense))) return rc;
    return finish_batch(c, out, out ? 1 : 0);
}

void pcop_occupancy_grid_dims(float x_min, float x_max, float y_min, float y_max, float block_size,
                              int32_t* grid_width, int32_t* grid_height) {
    // od.cpp:958-959, float arithmetic as written there (<math.h> in C++: the float overloads)
    if (grid_width) *grid_width = (int32_t)std::ceil((std::fabs(y_min) + std::fabs(y_max)) / block_size);
    if (grid_height) *grid_height = (int32_t)std::ceil((std::fabs(x_min) + std::fabs(x_max)) / block_size);
}

// The occupancy step, part one: argument checks.
static int occ_check(pcop_ctx* c, const pcop_occ_params* p) {
    const float lim[6] = {p->x_min, p->x_max, p->y_min, p->y_max, p->z_min, p->z_max};
    for (float v : lim) if (!std::isfinite(v)) { c->err = "occupancy limits must be finite"; return PCOP_ERR_PARAM; }
    if (!(p->block_size > 0.0f) || !std::isfinite(p->block_size) || !std::isfinite(p->dev_percent)) { c->err = "block_size must be finite and > 0"; return PCOP_ERR_PARAM; }
    if (p->grid_width <= 0 || p->grid_height <= 0 || (int64_t)p->grid_width * p->grid_height > (1 << 26)) { c->err = "occupancy grid must have 1 .. 2^26 cells"; return PCOP_ERR_PARAM; }
    // the walks of od.cpp:139-147 end only if the limits are reachable in a sane number of steps
    if ((std::fabs(p->x_min) + std::fabs(p->x_max) + std::fabs(p->y_min) + std::fabs(p->y_max)) / p->block_size > 1.0e8f) { c->err = "limits / block_size too large"; return PCOP_ERR_PARAM; }
    return PCOP_OK;
}

// ... part two: buffers and launches on the slot's staged cloud (no synchronisation; the kept cloud and its point count
// stay on the device: c->occ_xyz, c->occ_totals[0]).
static int occ_launch(pcop_ctx* c, Slot& sl, size_t n, const pcop_occ_params* p) {
    const size_t cells = (size_t)p->grid_width * p->grid_height, rows = (size_t)p->grid_height;
    if (!c->occ_xyz) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->occ_xyz), std::max<size_t>(c->cap, 1) * 3 * sizeof(float)));
        c->dev_allocs.push_back(c->occ_xyz);
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->occ_totals), 4 * sizeof(uint32_t)));
        c->dev_allocs.push_back(c->occ_totals);
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->occ_partials), (size_t)kOccPartials * kOccLdsCells * sizeof(uint32_t)));
        c->dev_allocs.push_back(c->occ_partials);
    }
    if (cells > c->occ_cells_cap) {   // the old, smaller buffers stay on the context's free list
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->occ_counts), cells * sizeof(unsigned long long)));
        c->dev_allocs.push_back(c->occ_counts);
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->occ_grid), cells));
        c->dev_allocs.push_back(c->occ_grid);
        c->occ_cells_cap = cells;
    }
    if (rows > c->occ_rows_cap) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->occ_rowavg), rows * sizeof(long long)));
        c->dev_allocs.push_back(c->occ_rowavg);
        c->occ_rows_cap = rows;
    }
    c->occ_valid = false;
    float* ox = c->occ_xyz; float* oy = ox + c->cap; float* oz = oy + c->cap;
    const uint32_t ntiles = div_up((uint32_t)n, kScanTile);
    HIPCHK(c, hipMemsetAsync(c->occ_counts, 0, cells * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->occ_totals, 0, 4 * sizeof(uint32_t), c->stream));
    if (cells <= (size_t)kOccLdsCells) HIPCHK(c, hipMemsetAsync(c->occ_partials, 0, (size_t)kOccPartials * cells * sizeof(uint32_t), c->stream));
    const OccParams op{p->x_min, p->x_max, p->y_min, p->y_max, p->z_min, p->z_max, p->block_size, p->dev_percent, p->grid_width, p->grid_height};
    c->ev_used = 0;
    if (n) {
        HIPCHK(c, hipMemsetAsync(sl.f.tile_a.get(), 0, (size_t)ntiles * sizeof(uint32_t), c->stream));
        mark(c, "start");
        k_occ_filter<<<ntiles, kBlock, 0, c->stream>>>(op, sl.src_x, sl.src_y, sl.src_z, (uint32_t)n, ox, oy, oz, c->occ_counts,
                                                        c->occ_partials, sl.f.tile_a.get(), c->occ_totals);
        mark(c, "k_occ_filter");
    } else {
        mark(c, "start");
    }
    k_occ_finish<<<(uint32_t)rows, kBlock, 0, c->stream>>>(op, c->occ_counts, c->occ_partials, c->occ_rowavg, c->occ_grid);
    mark(c, "k_occ_finish");
    HIPCHK(c, hipGetLastError());
    return PCOP_OK;
}

int pcop_occupancy_filter(pcop_ctx* c, const float* xyz, int layout, size_t n, int mem, const pcop_occ_params* p,
                          float* out_xyz, size_t* n_out, int64_t* counts, int64_t* row_avg, int8_t* grid,
                          int32_t* index_errors) {
    if (!c || !p) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    if ((rc = occ_check(c, p))) return rc;
    if ((rc = set_device(c))) return rc;
    Slot& sl = c->slots[0];
    sl.have_result = false;   // slot 0's staging buffers are about to be overwritten
    if ((rc = stage_input(c, sl, xyz, layout, n, mem))) return rc;
    if ((rc = occ_launch(c, sl, n, p))) return rc;
    const size_t cells = (size_t)p->grid_width * p->grid_height, rows = (size_t)p->grid_height;
    float* ox = c->occ_xyz; float* oy = ox + c->cap; float* oz = oy + c->cap;
    uint32_t totals[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(totals, c->occ_totals, sizeof(totals), hipMemcpyDeviceToHost, c->stream));
    if (counts) HIPCHK(c, hipMemcpyAsync(counts, c->occ_counts, cells * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    if (row_avg) HIPCHK(c, hipMemcpyAsync(row_avg, c->occ_rowavg, rows * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    if (grid) HIPCHK(c, hipMemcpyAsync(grid, c->occ_grid, cells, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_times(c);
    if (totals[2]) { c->err = "internal: a bounded device-side wait ran out"; return PCOP_ERR_STATE; }
    const uint32_t kept = totals[0];
    if (out_xyz && kept) {
        std::vector<float> planes((size_t)kept * 3);
        HIPCHK(c, hipMemcpy(planes.data(), ox, (size_t)kept * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(planes.data() + kept, oy, (size_t)kept * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(planes.data() + 2 * (size_t)kept, oz, (size_t)kept * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < kept; ++i) {
            out_xyz[3 * (size_t)i] = planes[i]; out_xyz[3 * (size_t)i + 1] = planes[(size_t)kept + i]; out_xyz[3 * (size_t)i + 2] = planes[2 * (size_t)kept + i];
        }
    }
    if (n_out) *n_out = kept;
    if (index_errors) *index_errors = (int32_t)totals[1];
    c->occ_n = kept;
    c->occ_valid = true;
    return PCOP_OK;
}

int pcop_process_filtered(pcop_ctx* c, pcop_result* out) {
    if (!c) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    if (!c->occ_valid) { c->err = "pcop_occupancy_filter has not produced a cloud on this context"; return PCOP_ERR_STATE; }
    if ((rc = check_params(c, true, true))) return rc;
    if ((rc = set_device(c))) return rc;
    Slot& sl = c->slots[0];
    sl.n = c->occ_n;
    sl.src_x = c->occ_xyz; sl.src_y = c->occ_xyz + c->cap; sl.src_z = c->occ_xyz + 2 * c->cap;
    c->nframes = 1;
    // the cloud build_initial_occupancy_grid_dataset fills by push_back keeps PointCloud's default is_dense = true (od.cpp:214)
    if ((rc = submit_common(c, kFrameChain, c->params.publish_clouds != 0, c->cloud_layout, 1))) return rc;
    return finish_batch(c, out, out ? 1 : 0);
}

// StatisticalOutlierRemoval::applyFilterIndices, second half: sequential double sums in index order, then the threshold.
static double sor_threshold(const float* dist, size_t n, float stddev_mul, std::vector<float>& sq);

int pcop_statistical_outlier_removal(pcop_ctx* c, const float* xyz, int layout, size_t n, int mean_k, float stddev_mul,
                                     float* out_xyz, size_t* n_out, float* distances, double* threshold) {
    if (!c || !n_out || (!out_xyz && n) || (!xyz && n)) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    c->slots[0].have_result = false;
    if (mean_k < 1 || mean_k + 1 > kKnnMax) { c->err = "mean_k must be 1..31"; return PCOP_ERR_PARAM; }
    if (!std::isfinite(stddev_mul)) { c->err = "stddev multiplier must be finite"; return PCOP_ERR_PARAM; }
    if (n == 0) {   // sor2.filter on an empty cloud gives an empty cloud and the callback carries on (the robot sees nothing in range)
        *n_out = 0;
        if (threshold) *threshold = 0.0;
        return PCOP_OK;
    }
    if (n < (size_t)mean_k + 1) { c->err = "fewer points than mean_k + 1: the reference reads past its k-NN result"; return PCOP_ERR_PARAM; }
    if (!c->cell_grid) { c->err = "statistical outlier removal needs the dense cell grid (context created with PCOP_CELL_PATH=hash)"; return PCOP_ERR_STATE; }
    if ((rc = set_device(c))) return rc;
    Slot& sl = c->slots[0];
    if ((rc = stage_input(c, sl, xyz, layout, n, PCOP_MEM_HOST))) return rc;
    // One pass of a cluster-only frame: bounds, a cell grid sized from them on the device (k_sor_setup), the k-NN kernel
    // writing the mean distances to pinned host memory, k_emit handing the borrowed per-cell counters back clean.
    FrameArgs& a = *sl.h_args;
    std::memset(&a, 0, sizeof(a));
    a.n = sl.n; a.is_dense = 1u; a.kind = kFrameClusterOnly; a.trivial = 0u;
    a.inv_leaf = 1.0f; a.r2 = 1.0f; a.inv_cell = 1.0;
    a.min_cluster = 1u; a.max_cluster = 0xFFFFFFFFu;
    a.out_layout = (uint32_t)PCOP_LAYOUT_AOS12; a.out_soa_stride = sl.n;
    a.cell_grid = 1u; a.nb_cap = c->nb_cap; a.gbits = c->gbits; a.tile = kBinTile;
    a.src_x = sl.src_x; a.src_y = sl.src_y; a.src_z = sl.src_z;
    Frame* fr = c->d_frames;
    hipStream_t s = c->stream;
    const dim3 tg(stride_grid(sl.n), 1), sg(std::max(1u, div_up(sl.n, kScanTile)), 1);
    c->ev_used = 0;
    mark(c, "start");
    k_frame_init<<<dim3(1, 1), 64, 0, s>>>(fr);
    k_minmax<<<dim3(kReduceBlocks, 1), kBlock, 0, s>>>(fr, 1u);
    k_sor_setup<<<dim3(1, 1), kBlock, 0, s>>>(fr, kReduceBlocks, (uint32_t)mean_k);
    mark(c, "bounds");
    k_cell_mark<<<tg, kBlock, 0, s>>>(fr);
    k_occ_scan<<<dim3(kCmapCells / 32u / kVmapTile, 1), kBlock, 0, s>>>(fr);
    k_cmap_coarse<<<dim3(256, 1), kBlock, 0, s>>>(fr);
    k_cell_count<<<tg, kBlock, 0, s>>>(fr);
    k_cell_starts<<<sg, kBlock, 0, s>>>(fr, 0u);
    k_cell_place<<<tg, kBlock, 0, s>>>(fr);
    mark(c, "cell_grid");
    if (mean_k + 1 <= 16) { k_knn_mean<16><<<tg, kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); mark(c, "k_knn_mean"); k_knn_far<16><<<dim3(kFarBlocks, 1), kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); }
    else { k_knn_mean<32><<<tg, kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); mark(c, "k_knn_mean"); k_knn_far<32><<<dim3(kFarBlocks, 1), kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); }
    mark(c, "k_knn_far");
    k_emit<<<tg, kBlock, 0, s>>>(fr, 1u);
    mark(c, "k_emit");
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    collect_times(c);
    const uint32_t e = sl.h_state->err;
    if (e & kErrNonFinite) { scrub_flags(c); c->err = "non-finite coordinates"; return PCOP_ERR_NONFINITE; }
    if (e) { scrub_flags(c); c->err = (e & kErrReplanCpath) ? "no cell grid fits this cloud" : "internal error in the k-NN pass"; return PCOP_ERR_STATE; }
    const float* dist = sl.h_xyzr;   // pinned: the kernel wrote the n mean distances here
    const double thr = sor_threshold(dist, n, stddev_mul, c->sor_sq);
    size_t kept = 0;
    const size_t fl = layout == PCOP_LAYOUT_AOS16 ? 4 : 3;
    // every point is written where the next kept one goes and the place advances only past an inlier: no branch on the data
    if (layout == PCOP_LAYOUT_SOA) {
        for (size_t i = 0; i < n; ++i) {
            for (int ax = 0; ax < 3; ++ax) out_xyz[(size_t)ax * n + kept] = xyz[(size_t)ax * n + i];
            kept += !((double)dist[i] > thr);
        }
    } else if (fl == 4) {
        for (size_t i = 0; i < n; ++i) { std::memmove(out_xyz + 4 * kept, xyz + 4 * i, 16); kept += !((double)dist[i] > thr); }
    } else {
        for (size_t i = 0; i < n; ++i) { std::memmove(out_xyz + 3 * kept, xyz + 3 * i, 12); kept += !((double)dist[i] > thr); }
    }
    *n_out = kept;
    if (distances) std::memcpy(distances, dist, n * sizeof(float));
    if (threshold) *threshold = thr;
    return PCOP_OK;
}

// sum of v[0..n) in double, bit for bit what `for (i) s += v[i]` gives.  The values are floats >= 0: each is a whole
// multiple of g = 2^(smallest biased exponent - 150), so every partial sum is one too; when n * largest / g stays
// below 2^53 every partial sum is representable, no addition rounds, and the order of the additions cannot matter -
// the sum is then taken with eight accumulators instead of one chain of dependent additions.  Otherwise: in order.
static double seq_sum_floats(const float* v, size_t n) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (size_t i = 0; i < n; ++i) {
        uint32_t b; std::memcpy(&b, v + i, 4);
        const uint32_t m = b - 1u;            // a zero adds nothing and must not drag the smallest value down
        lo = m < lo ? m : lo;
        hi = b > hi ? b : hi;
    }
    bool exact = false;
    if (hi == 0u) return 0.0;
    if (!(hi >> 31) && hi < 0x7F800000u) {     // all finite and non-negative (as non-negative floats order like their bit patterns)
        const uint32_t elo = std::max(1u, (lo + 1u) >> 23), ehi = std::max(1u, hi >> 23);
        uint32_t L = 0; while ((1ull << L) < n) ++L;
        exact = L + 24u + (ehi - elo) <= 53u;
    }
    if (!exact) {
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += v[i];
        return s;
    }
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; ++k) a[k] += v[i + k];
    for (; i < n; ++i) a[0] += v[i];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

static double sor_threshold(const float* dist, size_t n, float stddev_mul, std::vector<float>& sq) {
    sq.resize(n);
    for (size_t i = 0; i < n; ++i) sq[i] = dist[i] * dist[i];   // a float product, as the reference forms it
    const double sum = seq_sum_floats(dist, n), sq_sum = seq_sum_floats(sq.data(), n);
    const double valid = (double)n;
    const double mean = sum / valid;
    const double variance = (sq_sum - sum * sum / valid) / (valid - 1.0);
    return mean + (double)stddev_mul * std::sqrt(variance);
}

int pcop_segment_planes(pcop_ctx* c, const float* xyz, int layout, size_t n, const pcop_plane_params* p, float* out_xyz, size_t* n_out,
                        float* last_plane_xyz, size_t* n_last_plane, float* coeffs, int32_t cap_planes, int32_t* n_planes, int32_t* found_all) {
    return pcop_segment_planes_clouds(c, xyz, layout, n, p, out_xyz, n_out, last_plane_xyz, n_last_plane, nullptr, nullptr, coeffs, cap_planes, n_planes, found_all, nullptr);
}

int pcop_segment_planes_clouds(pcop_ctx* c, const float* xyz, int layout, size_t n, const pcop_plane_params* p, float* out_xyz, size_t* n_out,
                               float* last_plane_xyz, size_t* n_last_plane, float* before_last_xyz, size_t* n_before_last,
                               float* coeffs, int32_t cap_planes, int32_t* n_planes, int32_t* found_all, float* last_coeffs) {
    if (!c || !p || !n_out || !n_planes || !found_all || (!out_xyz && n) || (!xyz && n) || cap_planes < 0) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    if (!(p->dist_thresh >= 0.0f) || !std::isfinite(p->dist_thresh) || !std::isfinite(p->eps_angle)) { c->err = "plane removal: the distance threshold must be finite and >= 0"; return PCOP_ERR_PARAM; }
    if ((rc = set_device(c))) return rc;
    Slot& sl = c->slots[0];
    sl.have_result = false;
    if ((rc = stage_input(c, sl, xyz, layout, n, PCOP_MEM_HOST))) return rc;   // the cloud as SoA planes in HBM
    float* rb = reinterpret_cast<float*>(sl.f.rec_b.get());
    float* ra = reinterpret_cast<float*>(sl.f.rec_a.get());
    const PlaneScratch sa{{rb, rb + c->cap, rb + 2 * c->cap}}, sb{{ra, ra + c->cap, ra + 2 * c->cap}};   // rec_a / rec_b: 16 bytes per point each
    const float *ox, *oy, *oz;
    uint32_t left = 0;
    std::vector<float> cf, lastp, beforep;
    if ((rc = plane_removal_loop(c, sl, sl.src_x, sl.src_y, sl.src_z, (uint32_t)n, sl.h_vox, *p, sa, sb, &ox, &oy, &oz, &left, &cf, found_all,
                                 last_plane_xyz ? &lastp : nullptr, before_last_xyz ? &beforep : nullptr))) return rc;
    // what is left is on the host already (the loop keeps its copy in step): three planes in h_vox
    const float *hx = sl.h_vox, *hy = hx + n, *hz = hy + n;
    const size_t fl = layout == PCOP_LAYOUT_AOS16 ? 4 : 3;
    for (size_t i = 0; i < left; ++i) {
        if (layout == PCOP_LAYOUT_SOA) { out_xyz[i] = hx[i]; out_xyz[n + i] = hy[i]; out_xyz[2 * n + i] = hz[i]; }
        else { out_xyz[fl * i] = hx[i]; out_xyz[fl * i + 1] = hy[i]; out_xyz[fl * i + 2] = hz[i]; if (fl == 4) out_xyz[fl * i + 3] = 1.0f; }
    }
    *n_out = left;
    if (last_plane_xyz) {
        const size_t nl = lastp.size() / 3;
        for (size_t i = 0; i < nl; ++i) {
            if (layout == PCOP_LAYOUT_SOA) { last_plane_xyz[i] = lastp[3 * i]; last_plane_xyz[n + i] = lastp[3 * i + 1]; last_plane_xyz[2 * n + i] = lastp[3 * i + 2]; }
            else { last_plane_xyz[fl * i] = lastp[3 * i]; last_plane_xyz[fl * i + 1] = lastp[3 * i + 1]; last_plane_xyz[fl * i + 2] = lastp[3 * i + 2]; if (fl == 4) last_plane_xyz[fl * i + 3] = 1.0f; }
        }
        if (n_last_plane) *n_last_plane = nl;
    } else if (n_last_plane) {
        *n_last_plane = 0;
    }
    if (before_last_xyz) {
        const size_t nbl = beforep.size() / 3;
        for (size_t i = 0; i < nbl; ++i) {
            if (layout == PCOP_LAYOUT_SOA) { before_last_xyz[i] = beforep[3 * i]; before_last_xyz[n + i] = beforep[3 * i + 1]; before_last_xyz[2 * n + i] = beforep[3 * i + 2]; }
            else { before_last_xyz[fl * i] = beforep[3 * i]; before_last_xyz[fl * i + 1] = beforep[3 * i + 1]; before_last_xyz[fl * i + 2] = beforep[3 * i + 2]; if (fl == 4) before_last_xyz[fl * i + 3] = 1.0f; }
        }
        if (n_before_last) *n_before_last = nbl;
    } else if (n_before_last) {
        *n_before_last = 0;
    }
    *n_planes = (int32_t)(cf.size() / 4);
    if (coeffs) std::memcpy(coeffs, cf.data(), std::min<size_t>(cf.size(), (size_t)cap_planes * 4) * sizeof(float));
    if (last_coeffs && cf.size() >= 4) std::memcpy(last_coeffs, cf.data() + cf.size() - 4, 4 * sizeof(float));
    return PCOP_OK;
}

int pcop_process_node(pcop_ctx* c, const float* xyz, int layout, size_t n, int is_dense, int mem, const pcop_occ_params* occ,
                      int64_t* counts, int64_t* row_avg, int8_t* grid, int mean_k, float stddev_mul, const pcop_plane_params* plane,
                      pcop_result* out, pcop_node_stats* stats) {
    return pcop_process_node_clouds(c, xyz, layout, n, is_dense, mem, occ, counts, row_avg, grid, mean_k, stddev_mul, plane, out, stats, nullptr);
}

// The node chain on a cloud that is staged in slot 0 (any entry point): checks, then node_chain_run.
static int node_chain_checks(pcop_ctx* c, pcop_result* out, const pcop_occ_params* occ, int mean_k, float stddev_mul, const pcop_plane_params* plane,
                             pcop_node_clouds* clouds);
static int node_chain_run(pcop_ctx* c, size_t n, int is_dense, const pcop_occ_params* occ, int64_t* counts, int64_t* row_avg, int8_t* grid,
                          int mean_k, float stddev_mul, const pcop_plane_params* plane, pcop_result* out, pcop_node_stats* stats, pcop_node_clouds* clouds);

int pcop_process_node_clouds(pcop_ctx* c, const float* xyz, int layout, size_t n, int is_dense, int mem, const pcop_occ_params* occ,
                             int64_t* counts, int64_t* row_avg, int8_t* grid, int mean_k, float stddev_mul, const pcop_plane_params* plane,
                             pcop_result* out, pcop_node_stats* stats, pcop_node_clouds* clouds) {
    if (!c || !out) return PCOP_ERR_ARG;
    if (mem != PCOP_MEM_HOST && mem != PCOP_MEM_DEVICE) return PCOP_ERR_ARG;
    int rc = node_chain_checks(c, out, occ, mean_k, stddev_mul, plane, clouds);
    if (rc) return rc;
    if ((rc = set_device(c))) return rc;
    c->slots[0].have_result = false;
    if ((rc = stage_input(c, c->slots[0], xyz, layout, n, mem))) return rc;
    return node_chain_run(c, n, is_dense, occ, counts, row_avg, grid, mean_k, stddev_mul, plane, out, stats, clouds);
}

int pcop_process_node_pointcloud2(pcop_ctx* c, const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                                  uint32_t off_x, uint32_t off_y, uint32_t off_z, int is_bigendian, int is_dense, const pcop_occ_params* occ,
                                  int64_t* counts, int64_t* row_avg, int8_t* grid, int mean_k, float stddev_mul, const pcop_plane_params* plane,
                                  pcop_result* out, pcop_node_stats* stats, pcop_node_clouds* clouds) {
    if (!c || !out) return PCOP_ERR_ARG;
    int rc = node_chain_checks(c, out, occ, mean_k, stddev_mul, plane, clouds);
    if (rc) return rc;
    if ((rc = set_device(c))) return rc;
    c->slots[0].have_result = false;
    if ((rc = stage_pointcloud2(c, c->slots[0], data, width, height, point_step, row_step, off_x, off_y, off_z, is_bigendian))) return rc;
    return node_chain_run(c, (size_t)width * height, is_dense, occ, counts, row_avg, grid, mean_k, stddev_mul, plane, out, stats, clouds);
}

static int node_chain_checks(pcop_ctx* c, pcop_result* out, const pcop_occ_params* occ, int mean_k, float stddev_mul, const pcop_plane_params* plane,
                             pcop_node_clouds* clouds) {
    (void)out;
    if (clouds) clouds->n_voxel_grid = clouds->n_statistical_outliers = clouds->n_planar_cloud = clouds->n_indices_cloud = clouds->n_cloud_f = -1;
    int rc = busy(c);
    if (rc) return rc;
    if ((rc = check_params(c, true, true))) return rc;
    if (occ && (rc = occ_check(c, occ))) return rc;
    const bool sor = mean_k != 0;
    if (sor) {
        if (mean_k < 1 || mean_k + 1 > kKnnMax) { c->err = "mean_k must be 1..31"; return PCOP_ERR_PARAM; }
        if (!std::isfinite(stddev_mul)) { c->err = "stddev multiplier must be finite"; return PCOP_ERR_PARAM; }
        if (!c->cell_grid) { c->err = "statistical outlier removal needs the dense cell grid (context created with PCOP_CELL_PATH=hash)"; return PCOP_ERR_STATE; }
    }
    if (plane && (!(plane->dist_thresh >= 0.0f) || !std::isfinite(plane->dist_thresh) || !std::isfinite(plane->eps_angle))) {
        c->err = "plane removal: the distance threshold must be finite and >= 0"; return PCOP_ERR_PARAM;
    }
    if (sor || plane) {
        const double radius = (double)c->params.tol;
        if (!((float)(radius * radius) > 0.0f)) { c->err = "the node chain with outlier or plane removal needs a tolerance > 0"; return PCOP_ERR_PARAM; }
    }
    return PCOP_OK;
}

static int node_chain_run(pcop_ctx* c, size_t n, int is_dense, const pcop_occ_params* occ, int64_t* counts, int64_t* row_avg, int8_t* grid,
                          int mean_k, float stddev_mul, const pcop_plane_params* plane, pcop_result* out, pcop_node_stats* stats, pcop_node_clouds* clouds) {
    int rc = PCOP_OK;
    const bool sor = mean_k != 0;
    Slot& sl = c->slots[0];
    c->occ_valid = false;
    const uint32_t* n_dev = nullptr;
    if (occ) {   // the kept cloud and its point count stay on the device
        if ((rc = occ_launch(c, sl, n, occ))) return rc;
        const size_t cells = (size_t)occ->grid_width * occ->grid_height, rows = (size_t)occ->grid_height;
        if (counts) HIPCHK(c, hipMemcpyAsync(counts, c->occ_counts, cells * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        if (row_avg) HIPCHK(c, hipMemcpyAsync(row_avg, c->occ_rowavg, rows * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        if (grid) HIPCHK(c, hipMemcpyAsync(grid, c->occ_grid, cells, hipMemcpyDeviceToHost, c->stream));
        sl.src_x = c->occ_xyz; sl.src_y = c->occ_xyz + c->cap; sl.src_z = c->occ_xyz + 2 * c->cap;
        n_dev = c->occ_totals;
        is_dense = 1;   // the cloud build_initial_occupancy_grid_dataset fills keeps PointCloud's default is_dense = true (od.cpp:214)
    }
    c->nframes = 1;
    c->is_dense = is_dense; c->kind = kFrameChain; c->emit_cloud = c->params.publish_clouds != 0; c->out_layout = c->cloud_layout;
    c->cur_n_dev = n_dev;
    double thr = 0.0;
    uint32_t n_vox = 0, n_inl = 0, n_clu = 0;
    int32_t found_all = 1;
    std::vector<float> plane_coeffs, last_plane, before_last;
    const float* left_host = nullptr;
    uint32_t left_n0 = 0, left_n = 0;
    // a cloud between two steps -> the caller's 16-byte records: written by the GPU into pinned memory, copied after the next sync
    float4* pin16 = reinterpret_cast<float4*>(sl.h_clu);
    auto queue_cloud16 = [&](Frame* fr) { k_cloud16<<<stride_grid((uint32_t)std::max<size_t>(n, 1)), kBlock, 0, c->stream>>>(fr, pin16); };
    auto take_cloud16 = [&](float* dst, int32_t* cnt, uint32_t count) { if (count) std::memcpy(dst, pin16, (size_t)count * 16); *cnt = (int32_t)count; };
    auto mid_state = [&](Frame* fr) -> int {   // DevState to the host in the middle of the call
        k_state_out<<<dim3(1, 1), 64, 0, c->stream>>>(fr);
        HIPCHK(c, hipGetLastError());
        c->in_flight = true;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return PCOP_OK;
    };
    for (int attempt = 0;; ++attempt) {
        if (attempt >= 16) { c->cur_n_dev = nullptr; c->retry_to_caller = false; scrub_flags(c); c->err = "replanning did not converge"; return PCOP_ERR_STATE; }
        if (!sor && !plane) {   // the plain chain on the kept cloud; finish_batch re-plans by itself (the mailbox keeps the device count)
            c->retry_to_caller = false;
            if ((rc = enqueue_batch(c))) break;
            rc = finish_batch(c, out, 1);
            n_vox = n_inl = n_clu = sl.h_state->m;
            break;
        }
        c->retry_to_caller = true;
        c->sbox_batch = false; c->stream_chk = false;   // (the node chain launches by parts: the plain voxel stage with its bounds pass)
        const uint32_t nb = fill_frame_args(c);
        Frame* fr = c->d_frames;
        hipStream_t s = c->stream;
        c->ev_used = 0;
        mark(c, "start");
        enqueue_launches(c, kFrameChain, nb, 1, kPartVoxel | kPartOwnGridSetup);
        const dim3 tg(stride_grid(nb), 1), sg(std::max(1u, div_up(nb, kScanTile)), 1);
        const bool want_vg = clouds && clouds->voxel_grid, want_so = clouds && clouds->statistical_outliers && sor;
        if (want_vg) queue_cloud16(fr);   // the voxel cloud as downsample_cloud publishes it (od.cpp:288-295): before any filter
        if (sor) {   // outlier removal, first half: bounds of the voxel cloud, a cell grid sized for it, mean distance to the mean_k nearest neighbours
            k_minmax<<<dim3(kReduceBlocks, 1), kBlock, 0, s>>>(fr, 1u);
            k_sor_setup<<<dim3(1, 1), kBlock, 0, s>>>(fr, kReduceBlocks, (uint32_t)mean_k);
            k_cell_mark<<<tg, kBlock, 0, s>>>(fr);
            k_occ_scan<<<dim3(kCmapCells / 32u / kVmapTile, 1), kBlock, 0, s>>>(fr);
            k_cmap_coarse<<<dim3(256, 1), kBlock, 0, s>>>(fr);
            k_cell_count<<<tg, kBlock, 0, s>>>(fr);
            k_cell_starts<<<sg, kBlock, 0, s>>>(fr, 0u);
            k_cell_place<<<tg, kBlock, 0, s>>>(fr);
            mark(c, "sor_grid");
            if (mean_k + 1 <= 16) { k_knn_mean<16><<<tg, kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); mark(c, "k_knn_mean"); k_knn_far<16><<<dim3(kFarBlocks, 1), kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); }
            else { k_knn_mean<32><<<tg, kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); mark(c, "k_knn_mean"); k_knn_far<32><<<dim3(kFarBlocks, 1), kBlock, 0, s>>>(fr, (uint32_t)mean_k, 0.0, 1u); }
            mark(c, "k_knn_far");
            k_cells_reset<<<tg, kBlock, 0, s>>>(fr);
        }
        if ((rc = mid_state(fr))) break;
        const DevState* hs = sl.h_state;
        if (hs->err) {   // let finish_batch sort the error out: a re-plan comes back as 1, anything else as its status
            rc = finish_batch(c, nullptr, 0);
            if (rc == 1) continue;
            if (rc == PCOP_OK) { c->err = "internal: error bits without a status"; rc = PCOP_ERR_STATE; }
            break;
        }
        n_vox = hs->m;
        uint32_t npts = hs->c_n;   // the voxel cloud (or the input after a pass-through)
        n_inl = npts;
        if (want_vg) take_cloud16(clouds->voxel_grid, &clouds->n_voxel_grid, npts);
        if (sor) {   // second half: the points within the threshold
            thr = npts ? sor_threshold(sl.h_xyzr, npts, stddev_mul, c->sor_sq) : 0.0;
            sl.h_args->sor_thr = thr;
            k_sor_select<<<sg, kBlock, 0, s>>>(fr);
            mark(c, "k_sor_select");
            if (want_so) queue_cloud16(fr);   // what remove_statistical_outliers publishes (od.cpp:331-338)
            if (plane || want_so) {
                if ((rc = mid_state(fr))) break;
                c->in_flight = false;
                if (hs->err) { c->in_flight = true; rc = finish_batch(c, nullptr, 0); if (rc == 1) continue; if (rc == PCOP_OK) { c->err = "internal: error bits without a status"; rc = PCOP_ERR_STATE; } break; }
                npts = hs->c_n;
                if (want_so) take_cloud16(clouds->statistical_outliers, &clouds->n_statistical_outliers, npts);
            }
            n_inl = plane ? npts : 0u;   // (without plane removal: read from the final state below)
        }
        if (plane) {   // the cloud lives where the previous step left it; what is left alternates between two scratch plane sets
            const bool from_in = hs->c_from_input != 0;
            const float* px = from_in ? hs->src_x.get() : sl.f.cen_x.get();
            const float* py = from_in ? hs->src_y.get() : sl.f.cen_y.get();
            const float* pz = from_in ? hs->src_z.get() : sl.f.cen_z.get();
            float* rb = reinterpret_cast<float*>(sl.f.rec_b.get());
            float* ra = reinterpret_cast<float*>(sl.f.rec_a.get());
            // what is left alternates between the planes of rec_a and rec_b (16 bytes per point each; both idle between the
            // voxel / outlier steps and the cell stage).  After the outlier removal the cloud sits in rec_b's planes, so the
            // first round must not write there.  The staged input is never written: a re-plan runs the frame again from it.
            PlaneScratch s0{{ra, ra + c->cap, ra + 2 * c->cap}}, s1{{rb, rb + c->cap, rb + 2 * c->cap}};
            PlaneScratch first = px == rb ? s0 : s1, second = px == rb ? s1 : s0;
            const float *ox, *oy, *oz;
            uint32_t left = 0;
            plane_coeffs.clear();
            c->in_flight = false;
            last_plane.clear(); before_last.clear();
            rc = plane_removal_loop(c, sl, px, py, pz, npts, sl.h_vox, *plane, first, second, &ox, &oy, &oz, &left, &plane_coeffs, &found_all,
                                    clouds ? &last_plane : nullptr, clouds ? &before_last : nullptr, &left_host);
            if (rc) break;
            left_n0 = npts; left_n = left;
            if (clouds && clouds->planar_cloud) {   // (before the final stage overwrites the pinned buffer the host copy lives in)
                for (uint32_t i = 0; i < left; ++i) { float* q = clouds->planar_cloud + 4 * (size_t)i; q[0] = left_host[i]; q[1] = left_host[left_n0 + i]; q[2] = left_host[2 * (size_t)left_n0 + i]; q[3] = 1.0f; }
                clouds->n_planar_cloud = (int32_t)left;
            }
            if (clouds && plane_coeffs.empty()) {
                // no plane went: the reference's indices_cloud_y / filtered_cloud_y are still the copies of the outlier removal's cloud
                // cloud_cb made of it (od.cpp:774-775), and both are published as they are (od.cpp:410-426).  Nothing was removed, so
                // that cloud is what is left.
                for (float* dst : {clouds->indices_cloud, clouds->cloud_f}) {
                    if (!dst) continue;
                    for (uint32_t i = 0; i < left; ++i) { float* q = dst + 4 * (size_t)i; q[0] = left_host[i]; q[1] = left_host[left_n0 + i]; q[2] = left_host[2 * (size_t)left_n0 + i]; q[3] = 1.0f; }
                }
                if (clouds->indices_cloud) clouds->n_indices_cloud = (int32_t)left;
                if (clouds->cloud_f) clouds->n_cloud_f = (int32_t)left;
            }
            if (ox == ra && left) {   // the cell stage sorts its records into rec_a: what is left moves to rec_b's planes (free by now)
                HIPCHK(c, hipMemcpyAsync(rb, ox, (size_t)left * 4, hipMemcpyDeviceToDevice, s));
                HIPCHK(c, hipMemcpyAsync(rb + c->cap, oy, (size_t)left * 4, hipMemcpyDeviceToDevice, s));
                HIPCHK(c, hipMemcpyAsync(rb + 2 * c->cap, oz, (size_t)left * 4, hipMemcpyDeviceToDevice, s));
                ox = rb; oy = rb + c->cap; oz = rb + 2 * c->cap;
            }
            k_set_cloud<<<1, 64, 0, s>>>(fr, ox, oy, oz, left);
            mark(c, "plane_removal");
        }
        enqueue_launches(c, kFrameChain, nb, 1, kPartCluster | kPartEmit | kPartOwnGridSetup);
        HIPCHK(c, hipGetLastError());
        c->in_flight = true;
        rc = finish_batch(c, out, 1);
        if (rc == 1) continue;
        n_clu = sl.h_state->c_n;
        if (sor && !plane) n_inl = n_clu;
        break;
    }
    c->cur_n_dev = nullptr; c->retry_to_caller = false;
    if (rc) {
        // one way out for every error of the call, wherever in the frame it happened: nothing in flight, the occupancy flags all
        // zero again (the streaming voxel path and the cell grid rely on that), the stream drained
        c->in_flight = false;
        scrub_flags(c);
        if (rc == PCOP_ERR_STATE && (sl.h_state->err & kErrSorFew)) { c->err = "fewer voxels than mean_k + 1: the reference reads past its k-NN result"; return PCOP_ERR_PARAM; }
        return rc;
    }
    if (clouds && plane && !plane_coeffs.empty()) {   // at least one plane went: its points, and the cloud it was cut from
        auto put = [](const std::vector<float>& v, float* dst, int32_t* cnt) {
            const size_t m = v.size() / 3;
            if (dst) for (size_t i = 0; i < m; ++i) { dst[4 * i] = v[3 * i]; dst[4 * i + 1] = v[3 * i + 1]; dst[4 * i + 2] = v[3 * i + 2]; dst[4 * i + 3] = 1.0f; }
            if (dst) *cnt = (int32_t)m;
        };
        put(last_plane, clouds->indices_cloud, &clouds->n_indices_cloud);
        put(before_last, clouds->cloud_f, &clouds->n_cloud_f);
    }
    (void)left_n;
    const DevState* hs = sl.h_state;
    // the cloud the cluster indices refer to: what the last filter step kept (also of a pass-through frame's input), else the voxel cloud
    if (c->emit_cloud) out->num_voxels = (int32_t)((sor || plane) ? hs->c_n : (hs->c_from_input && !hs->passthrough ? hs->c_n : hs->m));
    out->voxel_xyz = (c->emit_cloud && out->num_voxels) ? sl.h_vox : nullptr;
    if (stats) {
        uint32_t totals[4] = {0, 0, 0, 0};
        if (occ) HIPCHK(c, hipMemcpy(totals, c->occ_totals, sizeof(totals), hipMemcpyDeviceToHost));
        stats->n_in = (int32_t)n; stats->n_filtered = (int32_t)hs->n_in; stats->index_errors = (int32_t)totals[1];
        stats->n_voxels = (int32_t)n_vox; stats->n_inliers = (int32_t)n_inl; stats->n_planes = (int32_t)(plane_coeffs.size() / 4);
        stats->n_clustered = (int32_t)n_clu; stats->planes_found_all = found_all; stats->threshold = thr;
    }
    if (occ) { c->occ_n = hs->n_in; c->occ_valid = true; }
    return PCOP_OK;
}

int pcop_cluster_summaries_frame(pcop_ctx* c, int frame, float* xyzr, int32_t cap_clusters) {
    if (!c || (!xyzr && cap_clusters) || frame < 0 || (uint32_t)frame >= c->max_frames) return PCOP_ERR_ARG;
    Slot& sl = c->slots[frame];
    if (!sl.have_result || c->in_flight) { c->err = "no finished clustered frame in this slot"; return PCOP_ERR_STATE; }
    int rc = set_device(c);
    if (rc) return rc;
    const uint32_t k = (uint32_t)sl.order.size();
    if (k == 0) return PCOP_OK;
    if ((uint32_t)cap_clusters < k) return PCOP_ERR_ARG;
    if (sl.result_trivial) {
        // singletons: centre = the point itself, r = 0
        const bool from_input = c->kind == kFrameClusterOnly || sl.h_state->passthrough;
        const float *px = from_input ? sl.src_x : sl.f.cen_x, *py = from_input ? sl.src_y : sl.f.cen_y, *pz = from_input ? sl.src_z : sl.f.cen_z;
        float* tmp = sl.h_xyzr;   // 3 planes of k floats
        HIPCHK(c, hipMemcpyAsync(tmp, px, (size_t)k * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(tmp + k, py, (size_t)k * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(tmp + 2 * (size_t)k, pz, (size_t)k * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<float> o(4 * (size_t)k);
        for (uint32_t i = 0; i < k; ++i) {
            const uint32_t s = (uint32_t)sl.order[i];
            o[4 * (size_t)i] = tmp[s]; o[4 * (size_t)i + 1] = tmp[k + s]; o[4 * (size_t)i + 2] = tmp[2 * (size_t)k + s]; o[4 * (size_t)i + 3] = 0.0f;
        }
        std::memcpy(xyzr, o.data(), o.size() * 4);
        return PCOP_OK;
    }
    // one launch over this slot only: a one-frame view of the frame table
    k_cluster_summaries<<<dim3(std::min(2048u, std::max(1u, div_up(k, kWavesPerBlock))), 1), kBlock, 0, c->stream>>>(
        c->d_frames + frame, c->frame_npass_csr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(sl.h_xyzr, sl.f.d_xyzr, (size_t)k * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < k; ++i) std::memcpy(xyzr + 4 * (size_t)i, sl.h_xyzr + 4 * (size_t)sl.order[i], 16);
    return PCOP_OK;
}

int pcop_set_cloud_layout(pcop_ctx* c, int layout) {
    if (!c || (layout != PCOP_LAYOUT_AOS16 && layout != PCOP_LAYOUT_AOS12)) return PCOP_ERR_ARG;
    int rc = busy(c);
    if (rc) return rc;
    c->cloud_layout = layout;
    return PCOP_OK;
}

int pcop_cluster_cloud(pcop_ctx* c, int frame, const float** xyz16, int32_t* n_points) {
    if (!c || !xyz16 || !n_points || frame < 0 || (uint32_t)frame >= c->max_frames) return PCOP_ERR_ARG;
    Slot& sl = c->slots[frame];
    if (!sl.have_result || c->in_flight) { c->err = "no finished clustered frame in this slot"; return PCOP_ERR_STATE; }
    int rc = set_device(c);
    if (rc) return rc;
    *xyz16 = sl.h_clu;
    *n_points = sl.last_indices;
    if (sl.last_indices == 0) return PCOP_OK;
    // h_indices (PCL order) and h_clu are pinned host memory: the kernel reads the one and writes the other over the bus
    k_cluster_cloud<<<std::min(1024u, std::max(1u, div_up((uint32_t)sl.last_indices, kBlock))), kBlock, 0, c->stream>>>(
        c->d_frames + frame, sl.h_indices, (uint32_t)sl.last_indices, reinterpret_cast<float4*>(sl.h_clu));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PCOP_OK;
}

int pcop_cluster_summaries(pcop_ctx* c, float* xyzr, int32_t cap_clusters) {
    return pcop_cluster_summaries_frame(c, 0, xyzr, cap_clusters);
}

int pcop_debug_buffer(pcop_ctx* c, const char* name, void** dev_ptr, size_t* bytes) {
    if (!c || !name || !dev_ptr || !bytes) return PCOP_ERR_ARG;
    const size_t n = c->cap, h = c->hash_cap;
    // "name" is slot 0's buffer, "name@k" slot k's
    std::string base(name);
    size_t slot = 0;
    const size_t at = base.find('@');
    if (at != std::string::npos) { slot = (size_t)std::strtoul(base.c_str() + at + 1, nullptr, 10); base.resize(at); }
    if (slot >= c->slots.size()) return PCOP_ERR_ARG;
    name = base.c_str();
    const Frame& f = c->slots[slot].f;
    struct Ent { const char* nm; void* p; size_t b; };
    const Ent tab[] = {
        {"state", c->d_frames + slot, sizeof(DevState)},   // DevState is the first member of Frame
        {"cen_x", f.cen_x, n * 4}, {"cen_y", f.cen_y, n * 4}, {"cen_z", f.cen_z, n * 4},
        {"rec_a", f.rec_a, n * 16}, {"rec_b", f.rec_b, n * 16},
        {"hkey", f.hkey, h * 8}, {"hcount", f.hcount, h * 4}, {"hmin", f.hmin, h * 4},
        {"cell_start", f.cell_start, h * 4}, {"parent", f.parent, h * 4}, {"comp_size", f.comp_size, h * 4},
        {"comp_rank", f.comp_rank, h * 4}, {"pt_slot", f.pt_slot, n * 4}, {"pt_rank", f.pt_rank, n * 4},
        {"pt_root", f.pt_root, n * 4}, {"idx_a", f.idx_a, n * 4}, {"idx_b", f.idx_b, n * 4},
        {"d_offsets", f.d_offsets, (n + 1) * 4}, {"bin_total", f.bin_total, (size_t)kMaxBins * 4},
        {"cell_list", f.cell_list, n * 4}, {"partials", f.partials, (size_t)kReduceBlocks * sizeof(MinMaxPartial)},
        {"binfo", f.binfo, f.binfo.get() ? (size_t)(kNbMax + 1) * 16 : 0}, {"bin2bucket", f.bin2bucket, f.bin2bucket.get() ? (size_t)2 << kBinBitsMax : 0},
        {"cflag", f.cflag, f.cflag.get() ? (size_t)kCmapCells : 0}, {"slot_meta", f.slot_meta, h * 16},
        {"hsub", f.hsub, f.hsub.get() ? ((size_t)kNbMax + 1) * kHeavyGroups * 16 : 0},   // (diagnostic builds: the tiled path's stamps)
    };
    for (const Ent& e : tab) {
        if (std::strcmp(e.nm, name) == 0) { *dev_ptr = e.p; *bytes = e.b; return PCOP_OK; }
    }
    return PCOP_ERR_ARG;
}

int pcop_debug_counter(pcop_ctx* c, const char* name, uint64_t* value) {
    if (!c || !name || !value) return PCOP_ERR_ARG;
    if (std::strcmp(name, "internal_retries") == 0) { *value = c->internal_retries; return PCOP_OK; }
    if (std::strcmp(name, "replans") == 0) { *value = c->replans; return PCOP_OK; }
    if (std::strcmp(name, "stale_reruns") == 0) { *value = c->stale_reruns; return PCOP_OK; }
    if (std::strcmp(name, "lds_ordered") == 0) { *value = c->lds_ordered ? 1u : 0u; return PCOP_OK; }
    if (std::strcmp(name, "stream_checked") == 0) { *value = c->stream_checked ? 1u : 0u; return PCOP_OK; }
    return PCOP_ERR_ARG;
}

int pcop_dev_alloc(pcop_ctx* c, size_t bytes, void** dev_ptr) {
    if (!c || !dev_ptr) return PCOP_ERR_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, std::max<size_t>(bytes, 1)));
    c->user_allocs.push_back(p);
    *dev_ptr = p;
    return PCOP_OK;
}

int pcop_dev_free(pcop_ctx* c, void* p) {
    if (!c) return PCOP_ERR_ARG;
    for (auto& q : c->user_allocs) {
        if (q == p && p) { hipSetDevice(c->device); hipFree(p); q = nullptr; return PCOP_OK; }
    }
    return PCOP_ERR_ARG;
}

int pcop_dev_upload(pcop_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return PCOP_ERR_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PCOP_OK;
}

int pcop_dev_download(pcop_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return PCOP_ERR_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PCOP_OK;
}

}  // extern "C"

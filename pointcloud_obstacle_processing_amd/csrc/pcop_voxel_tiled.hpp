// pcop_voxel_tiled.hpp - voxel grid, tiled path: the cloud is read once, written once as records and read once more.
// Part of pcop_device.hpp (included from there, after pcop_voxel.hpp; not a stand-alone header).
//
// The streaming path of pcop_voxel.hpp reads the cloud four times before a centroid exists (bounds, histogram, partition,
// reduce).  Two of those reads only serve bookkeeping: the bounds fix the origin of the voxel keys, the histogram sizes the
// partition.  Here
//   * the keys are formed against a box the HOST supplies (the union of the voxel boxes of the frames this context has seen,
//     with a margin): PCL's linear index orders voxels lexicographically by (iz, iy, ix), and floor(p * inv) does not depend
//     on the origin (the float subtraction of integer-valued floats is exact), so any box that holds the cloud gives the same
//     voxels in the same order.  k_vt_sort checks every point against the box and produces the true bounds on the way;
//     a point outside raises kErrReplanBox and the host re-runs the batch on the streaming path;
//   * the bucket plan comes from a SAMPLE (one row of 64 points in four, picked by a hash of the row group): k_vt_sample ->
//     k_bin_colsum -> k_bin_plan as before, with the capacity scaled.  The plan only divides the work - any plan gives the
//     same centroids - so an estimate is enough; k_vt_reduce sees the exact count of its bucket and refuses
//     (kErrReplanVt) if it does not fit its LDS;
//   * the partition is TILE-LOCAL: k_vt_sort sorts a tile of 8192 points by bucket (stably) inside the tile's own place in
//     rec_a and writes, per bucket, where the bucket starts in the tile.  No count has to be known before the tile is read,
//     so there is no histogram pass, no per-(tile, bucket) offset table to build, and the stores are whole contiguous
//     runs of 48 KB;
//   * k_vt_reduce gathers the pieces of its bucket from the tiles, in tile order (= index order), and from there on is
//     k_bucket_reduce: local voxel ids, stable counting sort in LDS, one lane per voxel sums its run in index order.
//   k_vt_setup     mailbox -> DevState with the host's box as the grid; dense cell grid set-up folded in
//   k_vt_sample    coarse-bin counts of the sampled rows -> ctab (k_bin_colsum / k_bin_plan follow)
//   k_vt_sort      tile-local stable sort by bucket + true bounds + box check + valid count
//   k_vt_reduce    pieces -> LDS -> centroids at the bucket's place in the scratch records, voxel count published
//   (k_voxel_gather  as on the streaming path; its last block also puts PCL's own min_b / div_b into DevState)
// Same float operations in the same order as the other two paths and the oracle.  Needs lane-ordered LDS atomics
// (k_lds_order_probe / k_lds_order_probe8); a context without them keeps the streaming path.
#pragma once

namespace pcop {

__device__ __forceinline__ bool vt_off(const DevState* st) { return voxel_stage_off(st) || !st->vt; }

// Diagnostic build only (-DPCOP_STAMPS): the shader clock at phase borders of one block, into the heavy-bucket tables (idle on
// this path); tools/stamps_vt.py prints the differences.
#ifdef PCOP_STAMPS
#define PCOP_VSTAMP(F, kern, idx, blk) do { if (threadIdx.x == 0 && blockIdx.x == (blk) && blockIdx.y == 0) \
    reinterpret_cast<unsigned long long*>((F).hsub.get())[(kern) * 32 + (idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define PCOP_VSTAMP(F, kern, idx, blk) do { } while (0)
#endif

// Voxel coordinates of a point relative to the box, and whether it lies inside.
struct BoxKey {
    float inv, fb0, fb1, fb2;
    uint32_t d0, d1, d2, mul1, mul2;
    __device__ __forceinline__ void init(const DevState* st) {
        inv = st->inv_leaf;
        fb0 = (float)st->min_b[0]; fb1 = (float)st->min_b[1]; fb2 = (float)st->min_b[2];
        d0 = (uint32_t)st->div_b[0]; d1 = (uint32_t)st->div_b[1]; d2 = (uint32_t)st->div_b[2];
        mul1 = st->mul[1]; mul2 = st->mul[2];
    }
    template <class P>
    __device__ __forceinline__ bool operator()(const P& p, uint32_t* key) const {   // p finite
        // clamped to [-1, 2^31) before the conversion (one v_med3_f32 each): a value far outside stays outside as an integer
        const float big = 2147483520.0f;
        const uint32_t i0 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(p.x * inv) - fb0, -1.0f, big), i1 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(p.y * inv) - fb1, -1.0f, big),
                       i2 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(p.z * inv) - fb2, -1.0f, big);
        *key = i0 + i1 * mul1 + i2 * mul2;
        return i0 < d0 && i1 < d1 && i2 < d2;
    }
    // The same, for a pass that also wants the bounds: the voxel coordinates themselves (floor(p * inv) is monotone in p, so their
    // minima and maxima are PCL's min_b / max_b relative to the box), and - as unsigned numbers - a coordinate outside the box on
    // either side is one at or above the box's size, so the maxima alone say whether every point was inside.
    template <class P>
    __device__ __forceinline__ uint32_t coords(const P& p, uint32_t* i0, uint32_t* i1, uint32_t* i2) const {   // p finite
        const float big = 2147483520.0f;
        *i0 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(p.x * inv) - fb0, -1.0f, big);
        *i1 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(p.y * inv) - fb1, -1.0f, big);
        *i2 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(p.z * inv) - fb2, -1.0f, big);
        return *i0 + *i1 * mul1 + *i2 * mul2;
    }
};

// Bounds of a tile as voxel coordinates relative to the box, kept in a MinMaxPartial's words (mn / mx bit for bit).
__device__ __forceinline__ void box_partial(MinMaxPartial* p, const uint32_t* mn, const uint32_t* mx, uint32_t valid, uint32_t nonfinite) {
    for (int a = 0; a < 3; ++a) { p->mn[a] = u2f(mn[a]); p->mx[a] = u2f(mx[a]); }
    p->valid = valid; p->nonfinite = nonfinite;
}

// One block per frame: DevState from the mailbox, the host's box as the voxel grid, the coarse bins, the cell grid.
// mode 1: the tiled path (k_vt_sort takes the bounds on its way); 2: the streaming path (k_bin_hist does); 3: the sort path (k_radix_hist of
// pass 0 does).
__global__ __launch_bounds__(kBlock) void k_vt_setup(Frame* frames, uint32_t npass, uint32_t maxbits, uint32_t fold_grid, uint32_t mode) {
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (threadIdx.x == 0) {
        frame_init(F);
        const FrameArgs a = *F.args;
        st->sbox = 1u; st->vpath = mode == 3u ? 0u : 1u;
        st->box_tile = mode == 1u ? kVtTile : (mode == 2u ? st->tile : (uint32_t)kTile);
        if (mode == 1u) { st->vt = 1u; st->plan_cap = a.plan_cap; st->tile = kBinTile; }
        uint64_t cells = 1;
        for (int k = 0; k < 3; ++k) {
            st->min_b[k] = a.box_lo[k]; st->div_b[k] = (int32_t)a.box_dim[k];
            cells *= a.box_dim[k];
            // bounds for the cell grid: one voxel of slack on either side swallows the rounding of p * inv
            st->vmin[k] = ((float)a.box_lo[k] - 1.0f) / a.inv_leaf;
            st->vmax[k] = ((float)a.box_lo[k] + (float)a.box_dim[k] + 1.0f) / a.inv_leaf;
        }
        st->mul[0] = 1u; st->mul[1] = a.box_dim[0]; st->mul[2] = a.box_dim[0] * a.box_dim[1];
        const uint32_t kb = bits_for(cells);   // (the host keeps the box below 2^30 cells)
        const bool fits = make_plan(&st->plan_voxel, kb, npass, maxbits);
        if (mode == 3u) {
            if (!fits) atomicOr(&st->err, (uint32_t)kErrReplanVoxel);
        } else {
            st->gbits = min(kb, st->gbits);
            st->cshift = kb - st->gbits;
            if (((mode == 1u ? kVtSpanMax : kBucketCells) >> st->cshift) == 0u) atomicOr(&st->err, (uint32_t)kErrReplanBins);   // a coarse bin spans more voxels than a bucket may
        }
        if (mode == 1u && (st->n_in == 0u || st->n_in > (kVtMaxTiles - 1u) * kVtTile)) atomicOr(&st->err, (uint32_t)kErrReplanVt);
    }
    if (fold_grid) { __syncthreads(); cgrid_setup_block(F, st->n_in, st->n_in == 0u); }
    if (mode == 3u) {   // k_voxel_reduce publishes "heads of tile t, plus one" here; zero = not yet published
        __syncthreads();
        uint32_t* __restrict__ tile_heads = F.tile_a;
        const uint32_t ntiles = (st->n_in + kScanTile - 1) / kScanTile;
        for (uint32_t t = threadIdx.x; t < ntiles; t += kBlock) tile_heads[t] = 0u;
    }
}

// Pass 0 of the sort path with the keys on the host's box: the digit counts as k_radix_hist makes them, and on the way the tile's
// bounds (voxel coordinates: box_partial), its valid points and the box check.  The block's 256 threads call; s_h is zeroed.
template <int MAXB>
__device__ __forceinline__ void radix_hist_on_box(const Frame& F, uint32_t* s_h, uint32_t tile_base, const PassInfo& pi, const RecSrc& src) {
    __shared__ uint32_t s_red[kWavesPerBlock * 8u];
    DevState* st = const_cast<DevState*>(&F.st);
    BoxKey key; key.init(st);
    const uint32_t mask = (1u << pi.bits) - 1u;
    const uint32_t wbase = tile_base + (threadIdx.x / kWave) * (kItems * kWave) + lane_id();
    Rec3 rec[kItems];
    load_records(src, wbase, pi.count, rec);
    uint32_t mn0 = 0xFFFFFFFFu, mn1 = 0xFFFFFFFFu, mn2 = 0xFFFFFFFFu, mx0 = 0u, mx1 = 0u, mx2 = 0u, nbad = 0u, nok = 0u;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < pi.count) {
            const Rec3 r = rec[j];
            uint32_t d;
            if (!finite3(r.x, r.y, r.z)) { d = pi.nb - 1; ++nbad; }   // (a cloud declared dense: the frame is given up below)
            else {
                uint32_t i0, i1, i2;
                d = (key.coords(r, &i0, &i1, &i2) >> pi.shift) & mask;
                mn0 = min(mn0, i0); mx0 = max(mx0, i0); mn1 = min(mn1, i1); mx1 = max(mx1, i1); mn2 = min(mn2, i2); mx2 = max(mx2, i2);
                ++nok;
            }
            atomicAdd(&s_h[d], 1u);
        }
    }
    uint32_t v[6] = {mn0, mn1, mn2, mx0, mx1, mx2};
#pragma unroll
    for (int dd = kWave / 2; dd > 0; dd >>= 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { v[q] = min(v[q], (uint32_t)__shfl_xor(v[q], dd, kWave)); v[3 + q] = max(v[3 + q], (uint32_t)__shfl_xor(v[3 + q], dd, kWave)); }
    }
    nbad = wave_sum(nbad); nok = wave_sum(nok);
    const uint32_t w = threadIdx.x / kWave;
    if (lane_id() == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) s_red[w * 8u + q] = v[q];
        s_red[w * 8u + 6u] = nbad; s_red[w * 8u + 7u] = nok;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u}, bad = 0u, ok = 0u;
        for (uint32_t q = 0; q < (uint32_t)kWavesPerBlock; ++q) {
            for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], s_red[q * 8u + a]); mx[a] = max(mx[a], s_red[q * 8u + 3 + a]); }
            bad += s_red[q * 8u + 6u]; ok += s_red[q * 8u + 7u];
        }
        MinMaxPartial p;
        box_partial(&p, mn, mx, ok, bad);
        F.partials.get()[blockIdx.x] = p;
        if (ok) atomicAdd(&st->n_valid, ok);
        if (bad) { if (st->is_dense == 0) atomicAdd(&st->vt_nonfinite, bad); else atomicOr(&st->err, (uint32_t)kErrNonFinite); }
        if (ok && (mx[0] >= key.d0 || mx[1] >= key.d1 || mx[2] >= key.d2)) atomicOr(&st->err, (uint32_t)kErrReplanBox);   // a point outside the box
    }
}

__device__ __forceinline__ void bin_hist_on_box(Frame& F, uint32_t* s_hist, uint32_t tile_base, uint32_t tile, uint32_t n, uint32_t G, uint32_t cs) {
    constexpr uint32_t NW = kBinThreads / kWave;
    __shared__ uint32_t s_red[NW * 8u];
    DevState* st = &F.st;
    uint16_t* __restrict__ ctab = F.ctab;
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    BoxKey key; key.init(st);
    uint32_t mn0 = 0xFFFFFFFFu, mn1 = 0xFFFFFFFFu, mn2 = 0xFFFFFFFFu, mx0 = 0u, mx1 = 0u, mx2 = 0u;
    uint32_t nbad = 0u, nok = 0u;
    constexpr int kRound = kBinTileSmall / kBinThreads;   // records in flight per thread: 4
    for (uint32_t r0 = 0; r0 < tile / kBinThreads; r0 += kRound) {
        Rec3 r[kRound];
#pragma unroll
        for (int k = 0; k < kRound; ++k) {   // clamped, unconditional loads
            const uint32_t i = min(tile_base + (r0 + k) * kBinThreads + threadIdx.x, n - 1u);
            r[k].x = x[i]; r[k].y = y[i]; r[k].z = z[i];
        }
#pragma unroll
        for (int k = 0; k < kRound; ++k) {
            const uint32_t i = tile_base + (r0 + k) * kBinThreads + threadIdx.x;
            if (i < n) {
                if (finite3(r[k].x, r[k].y, r[k].z)) {
                    uint32_t i0, i1, i2;
                    const uint32_t kk = key.coords(r[k], &i0, &i1, &i2);
                    mn0 = min(mn0, i0); mx0 = max(mx0, i0); mn1 = min(mn1, i1); mx1 = max(mx1, i1); mn2 = min(mn2, i2); mx2 = max(mx2, i2);
                    ++nok;
                    atomicAdd(&s_hist[min(kk >> cs, G - 1u)], 1u);   // (a point outside the box lands in some bin: the frame is given up below)
                } else ++nbad;
            }
        }
    }
    {
        uint32_t v[6] = {mn0, mn1, mn2, mx0, mx1, mx2};
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { v[q] = min(v[q], (uint32_t)__shfl_xor(v[q], d, kWave)); v[3 + q] = max(v[3 + q], (uint32_t)__shfl_xor(v[3 + q], d, kWave)); }
        }
        nbad = wave_sum(nbad); nok = wave_sum(nok);
        const uint32_t w = threadIdx.x / kWave;
        if (lane_id() == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) s_red[w * 8u + q] = v[q];
            s_red[w * 8u + 6u] = nbad; s_red[w * 8u + 7u] = nok;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u}, bad = 0u, ok = 0u;
        for (uint32_t q = 0; q < NW; ++q) {
            for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], s_red[q * 8u + a]); mx[a] = max(mx[a], s_red[q * 8u + 3 + a]); }
            bad += s_red[q * 8u + 6u]; ok += s_red[q * 8u + 7u];
        }
        MinMaxPartial p;
        box_partial(&p, mn, mx, ok, bad);
        F.partials[blockIdx.x] = p;
        if (ok) atomicAdd(&st->n_valid, ok);
        if (bad) { if (st->is_dense == 0) atomicAdd(&st->vt_nonfinite, bad); else atomicOr(&st->err, (uint32_t)kErrNonFinite); }
        if (ok && (mx[0] >= key.d0 || mx[1] >= key.d1 || mx[2] >= key.d2)) atomicOr(&st->err, (uint32_t)kErrReplanBox);   // a point outside the box
    }
    uint16_t* __restrict__ row = ctab + (size_t)blockIdx.x * G;
    for (uint32_t b = threadIdx.x; b < G; b += kBinThreads) row[b] = (uint16_t)s_hist[b];   // a tile has at most 2^15 points (kBinTileLarge): fits
}

// Which of the kVtSampleRows rows of group g is the sampled one (a hash, so that no period of the input lines up with it).
__device__ __forceinline__ uint32_t vt_sample_row(uint32_t g) {
    uint32_t h = g * 0x9E3779B9u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
    return h & (kVtSampleRows - 1u);
}

// Coarse-bin counts of the sampled rows of a tile of kBinTile points -> ctab[tile][bin], the layout k_bin_colsum reads.
__global__ __launch_bounds__(kBinThreads) void k_vt_sample(Frame* frames) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_dyn);
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    uint16_t* __restrict__ ctab = F.ctab;
    if (vt_off(st)) return;
    const uint32_t n = st->n_in;
    const uint32_t tile_base = blockIdx.x * kBinTile;
    if (tile_base >= n) return;
    const uint32_t G = 1u << st->gbits, cs = st->cshift;
    for (uint32_t b = threadIdx.x; b < G; b += kBinThreads) s_hist[b] = 0u;
    __syncthreads();
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    BoxKey key; key.init(st);
    constexpr uint32_t NW = kBinThreads / kWave;                              // 16 waves
    constexpr uint32_t kGroups = kBinTile / (kWave * kVtSampleRows);          // 64 row groups per tile
    constexpr uint32_t kPer = kGroups / NW;                                   // 4 sampled rows per wave
    const uint32_t w = threadIdx.x / kWave, lane = lane_id();
    Rec3 r[kPer];
    uint32_t idx[kPer];
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t g = w + NW * q;
        const uint32_t row = g * kVtSampleRows + vt_sample_row(blockIdx.x * kGroups + g);
        idx[q] = tile_base + row * kWave + lane;
        const uint32_t i = min(idx[q], n - 1u);   // clamped, unconditional
        r[q].x = x[i]; r[q].y = y[i]; r[q].z = z[i];
    }
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        uint32_t k;
        if (idx[q] < n && finite3(r[q].x, r[q].y, r[q].z) && key(r[q], &k)) atomicAdd(&s_hist[min(k >> cs, G - 1u)], 1u);
    }
    __syncthreads();
    uint16_t* __restrict__ row = ctab + (size_t)blockIdx.x * G;
    for (uint32_t b = threadIdx.x; b < G; b += kBinThreads) row[b] = (uint16_t)s_hist[b];
}

// The 8-wave form of k_lds_order_probe: what k_vt_sort relies on.  Eight waves add to 16-bit fields of shared words (waves 2k
// and 2k + 1 share a word), sixteen steps each, pseudo-random words, some lanes idle, while the other waves hammer the same
// array.  Every returned field must equal (earlier steps of this wave on the word) + (lower lanes of this step on it).
__global__ __launch_bounds__(kVtThreads) void k_lds_order_probe8(uint32_t* __restrict__ violations) {
    constexpr uint32_t NB = 1024u, NW = kVtThreads / kWave;
    __shared__ uint32_t cnt[(NW / 2u) * NB];
    __shared__ uint16_t seen[NW * NB];   // per wave: what it has added to every word so far (plain stores, program order)
    const uint32_t lane = lane_id(), w = threadIdx.x / kWave;
    const uint32_t wword = w >> 1, wshift = 16u * (w & 1u);
    uint32_t bad = 0u;
    for (uint32_t trial = 0; trial < 4u; ++trial) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (NW / 2u) * NB; i += kVtThreads) cnt[i] = 0u;
        for (uint32_t i = threadIdx.x; i < NW * NB; i += kVtThreads) seen[i] = 0u;
        __syncthreads();
        const uint32_t ranges[4] = {NB, 64u, 7u, 1u};
        const uint32_t range = ranges[trial];
        for (uint32_t step = 0; step < kVtItems; ++step) {
            uint32_t h = ((blockIdx.x * 4u + trial) * kVtItems + step) * 0x9E3779B9u + (threadIdx.x + 1u) * 0x85EBCA6Bu;
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
            const uint32_t a = h % range;
            const bool active = ((h >> 20) & 15u) != 0u;
            uint32_t old = 0u;
            if (active) old = atomicAdd(&cnt[wword * NB + a], 1u << wshift);
            const uint32_t prev = seen[w * NB + a];
            uint32_t lower = 0u, total = 0u;
            for (uint32_t l = 0; l < kWave; ++l) {
                const uint32_t al = __shfl(a, l, kWave);
                const bool actl = __shfl(active ? 1u : 0u, l, kWave) != 0u;
                if (actl && al == a) { ++total; if (l < lane) ++lower; }
            }
            if (active && ((old >> wshift) & 0xFFFFu) != prev + lower) ++bad;
            __builtin_amdgcn_wave_barrier();
            if (active && lower == 0u) seen[w * NB + a] = (uint16_t)(prev + total);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (bad) atomicAdd(violations, bad);
}

// LDS of k_vt_sort for nbc buckets and a staging area of stg records.
__host__ __device__ constexpr size_t vt_sort_lds_bytes(uint32_t nbc, uint32_t stg) {
    return (size_t)stg * 12u + (size_t)nbc * 16u + 64u * 8u;
}

// One tile of kVtTile points: sorted by bucket, stably, into the tile's own place of rec_a (record i of the tile's output is
// rec_a[tile * kVtTile + i]); pre[b][tile] = where bucket b starts in the tile's output, pre[nb][tile] = the tile's valid points.
// A wave owns 1024 consecutive points and takes them 64 at a time, so the value an LDS atomic add returns - lanes in
// order, steps in order - is the point's rank among the wave's points of its bucket; the waves' counts sit side by side in
// 16-bit fields and a scan over (bucket, wave) turns them into places.  CHECK: the finished order is verified (one more
// pass through LDS with {bucket, index} words that must rise strictly); the host runs this build for a context's first
// tiled batch.
template <uint32_t PB, bool CHECK>   // PB: buckets per thread in the per-bucket step; the launch provides for PB * 512 buckets
__global__ __launch_bounds__(kVtThreads, PB <= 3u ? 4 : 2) void k_vt_sort(Frame* frames) {   // (two blocks per CU in the usual instances)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    constexpr uint32_t NW = kVtThreads / kWave;                       // 8
    constexpr uint32_t NBC = PB * kVtThreads;
    constexpr uint32_t STG = PB <= 3u ? 4096u : 2048u;                // records staged per round
    static_assert(NW == 8u, "four words of two 16-bit fields per bucket");
    static_assert(kVtTile * 4u <= STG * 12u + NBC * 16u, "the checking pass fits the staging area and the counters behind it (both idle by then)");
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (vt_off(st)) return;
    const uint32_t n = st->n_in;
    const uint32_t tile_base = blockIdx.x * kVtTile;
    if (tile_base >= n) return;
    const uint32_t nb = st->nbuckets, G = 1u << st->gbits, cs = st->cshift, ts = F.tstride;
    if (st->nb_cap > NBC) return;   // (the host launches an instance that fits)
    const uint16_t* __restrict__ bin2bucket = F.bin2bucket;
    uint16_t* __restrict__ pre = F.tcnt;
    unsigned char* stage = s_dyn;
    uint32_t* wcnt32 = reinterpret_cast<uint32_t*>(s_dyn + (size_t)STG * 12u);   // [4 words][NBC]: counts of waves (2k, 2k+1) in the halves of word k, then their places
    uint32_t* s_red = reinterpret_cast<uint32_t*>(s_dyn + (size_t)STG * 12u + (size_t)NBC * 16u);   // 8 waves x 8 words
    __shared__ uint32_t s_w[NW];
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    const bool drop = st->is_dense == 0;
    BoxKey key; key.init(st);
    const uint32_t w = threadIdx.x / kWave, lane = lane_id();
    const uint32_t wword = w >> 1, wshift = 16u * (w & 1u);
    const uint32_t wbase = tile_base + w * (kVtItems * kWave) + lane;
    Rec3 rec[kVtItems];
#pragma unroll
    for (uint32_t j = 0; j < kVtItems; ++j) {
        const uint32_t i = min(wbase + j * kWave, n - 1u);   // clamped, unconditional
        rec[j].x = x[i]; rec[j].y = y[i]; rec[j].z = z[i];
    }
    for (uint32_t i = threadIdx.x; i < NBC; i += kVtThreads) reinterpret_cast<uint4*>(wcnt32)[i] = make_uint4(0u, 0u, 0u, 0u);
    PCOP_VSTAMP(F, 0, 0, 40);
    // keys, the true bounds (as voxel coordinates relative to the box: box_partial), the box check
    uint32_t mn0 = 0xFFFFFFFFu, mn1 = 0xFFFFFFFFu, mn2 = 0xFFFFFFFFu, mx0 = 0u, mx1 = 0u, mx2 = 0u;
    uint32_t nbad = 0u, nout = 0u;
    uint32_t bin[kVtItems];
    uint32_t actm = 0u;
#pragma unroll
    for (uint32_t j = 0; j < kVtItems; ++j) {
        const bool there = wbase + j * kWave < n;
        const bool fin = finite3(rec[j].x, rec[j].y, rec[j].z);
        uint32_t i0, i1, i2;
        const uint32_t k = key.coords(rec[j], &i0, &i1, &i2);
        const bool in = i0 < key.d0 && i1 < key.d1 && i2 < key.d2;
        if (there && fin) { mn0 = min(mn0, i0); mx0 = max(mx0, i0); mn1 = min(mn1, i1); mx1 = max(mx1, i1); mn2 = min(mn2, i2); mx2 = max(mx2, i2); }
        nbad += (there && !fin) ? 1u : 0u;
        nout += (there && fin && !in) ? 1u : 0u;
        const bool act = there && fin && in;
        actm |= act ? (1u << j) : 0u;
        bin[j] = act ? min(k >> cs, G - 1u) : 0u;
    }
    uint32_t pos[kVtItems];   // low half: rank inside (wave, bucket), then the place in the tile's output; high half: the bucket
#pragma unroll
    for (uint32_t j = 0; j < kVtItems; ++j) pos[j] = min((uint32_t)bin2bucket[bin[j]], NBC - 1u) << 16;   // all requested before any is used
    __syncthreads();   // the counters are zero
    PCOP_VSTAMP(F, 0, 1, 40);
#pragma unroll
    for (uint32_t j = 0; j < kVtItems; ++j) {
        uint32_t old = 0u;
        if ((actm >> j) & 1u) old = atomicAdd(&wcnt32[wword * NBC + (pos[j] >> 16)], 1u << wshift);
        pos[j] |= (old >> wshift) & 0xFFFFu;
    }
    // bounds of the tile (min / max are exact in any order)
    {
        uint32_t v[6] = {mn0, mn1, mn2, mx0, mx1, mx2};
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { v[q] = min(v[q], (uint32_t)__shfl_xor(v[q], d, kWave)); v[3 + q] = max(v[3 + q], (uint32_t)__shfl_xor(v[3 + q], d, kWave)); }
        }
        nbad = wave_sum(nbad); nout = wave_sum(nout);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) s_red[w * 8u + q] = v[q];
            s_red[w * 8u + 6u] = nbad; s_red[w * 8u + 7u] = nout;
        }
    }
    __syncthreads();
    PCOP_VSTAMP(F, 0, 2, 40);
    // per bucket: the tile's points, the waves' places (absolute inside the tile: below 2^16)
    uint32_t tot[PB];
    uint4 cw[PB];
    uint32_t tsum = 0u;
#pragma unroll
    for (uint32_t k = 0; k < PB; ++k) {
        const uint32_t b = threadIdx.x * PB + k;
        cw[k] = make_uint4(wcnt32[b], wcnt32[NBC + b], wcnt32[2u * NBC + b], wcnt32[3u * NBC + b]);
        tot[k] = (cw[k].x & 0xFFFFu) + (cw[k].x >> 16) + (cw[k].y & 0xFFFFu) + (cw[k].y >> 16) +
                 (cw[k].z & 0xFFFFu) + (cw[k].z >> 16) + (cw[k].w & 0xFFFFu) + (cw[k].w >> 16);
        tsum += tot[k];
    }
    uint32_t ttot;
    uint32_t lrun = block_excl_scan_n<NW>(tsum, s_w, &ttot);
#pragma unroll
    for (uint32_t k = 0; k < PB; ++k) {
        const uint32_t b = threadIdx.x * PB + k;
        const uint32_t wv[4] = {cw[k].x, cw[k].y, cw[k].z, cw[k].w};
        uint32_t o[4], r = lrun;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t lo = r; r += wv[q] & 0xFFFFu; const uint32_t hi = r; r += wv[q] >> 16; o[q] = (lo & 0xFFFFu) | (hi << 16); }
        wcnt32[b] = o[0]; wcnt32[NBC + b] = o[1]; wcnt32[2u * NBC + b] = o[2]; wcnt32[3u * NBC + b] = o[3];
        if (b <= nb) pre[(size_t)b * ts + blockIdx.x] = (uint16_t)lrun;   // (b == nb: every bucket is below it, so this is the tile's total; 8192 fits)
        lrun += tot[k];
    }
    if (threadIdx.x == 0) {
        uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u}, bad = 0u, out = 0u;
        for (uint32_t q = 0; q < NW; ++q) {
            for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], s_red[q * 8u + a]); mx[a] = max(mx[a], s_red[q * 8u + 3 + a]); }
            bad += s_red[q * 8u + 6u]; out += s_red[q * 8u + 7u];
        }
        MinMaxPartial p;
        box_partial(&p, mn, mx, ttot + out, bad);
        F.partials[blockIdx.x] = p;
        if (ttot) atomicAdd(&st->n_valid, ttot);
        if (bad) { if (drop) atomicAdd(&st->vt_nonfinite, bad); else atomicOr(&st->err, (uint32_t)kErrNonFinite); }
        if (out) atomicOr(&st->err, (uint32_t)kErrReplanBox);
    }
    __syncthreads();
    PCOP_VSTAMP(F, 0, 3, 40);
#pragma unroll
    for (uint32_t j = 0; j < kVtItems; ++j)
        pos[j] += (wcnt32[wword * NBC + (pos[j] >> 16)] >> wshift) & 0xFFFFu;   // place inside the tile's output (below 2^13: no carry into the bucket)
    // the tile's output, STG records at a time: staged in LDS in output order, stored as one contiguous run
    uint4* __restrict__ out4 = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(F.rec_a.get()) + (size_t)tile_base * 12u);
    Rec3* stage_r = reinterpret_cast<Rec3*>(stage);
    for (uint32_t r0 = 0; r0 < ttot; r0 += STG) {
#pragma unroll
        for (uint32_t j = 0; j < kVtItems; ++j) {
            const uint32_t p = (pos[j] & 0xFFFFu) - r0;
            if (((actm >> j) & 1u) && p < STG) stage_r[p] = rec[j];
        }
        __syncthreads();
        const uint32_t cnt = min(STG, ttot - r0);
        const uint32_t nvec = (cnt * 12u + 15u) / 16u;   // (the last store may carry up to 12 bytes past the records: still inside the tile's place)
        const uint4* stage4 = reinterpret_cast<const uint4*>(stage);
        uint4* __restrict__ dst = out4 + (size_t)r0 * 12u / 16u;
        PCOP_VSTAMP(F, 0, 4 + 2 * (r0 / STG), 40);
        for (uint32_t v = threadIdx.x; v < nvec; v += kVtThreads) dst[v] = stage4[v];
        __syncthreads();
        PCOP_VSTAMP(F, 0, 5 + 2 * (r0 / STG), 40);
    }
    if constexpr (CHECK) {
        // {bucket, index inside the tile} of every record at its place: the words must rise strictly
        uint32_t* chk = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
        for (uint32_t j = 0; j < kVtItems; ++j)
            if ((actm >> j) & 1u) chk[pos[j] & 0xFFFFu] = (pos[j] & 0xFFFF0000u) | (w * (kVtItems * kWave) + j * kWave + lane);
        __syncthreads();
        bool wrong = false;
        for (uint32_t e = threadIdx.x + 1u; e < ttot; e += kVtThreads) wrong |= chk[e - 1u] >= chk[e];
        if (wrong) atomicOr(&st->err, (uint32_t)kErrOrder);
    }
}

// The same check for what k_vt_accum relies on: FLOAT adds to LDS.  A wave's ds_add_f32 instructions execute in program order,
// and the lanes of one instruction that hit the same word are added in lane order, each add rounded like v_add_f32 (denormal
// range included): the word then holds exactly the sequential sum in (instruction, lane) order.  Values of mixed sign over
// 24 binades (every second trial in the denormal range) make any other order or rounding show in the last bits.
__global__ __launch_bounds__(kBlock) void k_lds_float_order_probe(uint32_t* __restrict__ violations) {
    __shared__ float acc[kWavesPerBlock][512];
    const uint32_t lane = lane_id(), w = threadIdx.x / kWave;
    uint32_t bad = 0u;
    for (uint32_t trial = 0; trial < 16u; ++trial) {
        for (uint32_t i = lane; i < 512u; i += kWave) acc[w][i] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        const uint32_t ranges[4] = {512u, 64u, 9u, 1u};
        const uint32_t range = ranges[trial & 3u];
        float v[16]; uint32_t id[16];
#pragma unroll
        for (uint32_t q = 0; q < 16u; ++q) {
            uint32_t h = ((blockIdx.x * 16u + trial) * 16u + q) * 0x9E3779B9u + (threadIdx.x + 1u) * 0x85EBCA6Bu;
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
            id[q] = h % range;
            const int e = (trial & 4u) ? -140 + (int)((h >> 8) & 15u) : -12 + (int)((h >> 8) % 25u);
            const float m = 1.0f + (float)((h >> 3) & 0xFFFFu) * (1.0f / 65536.0f);
            v[q] = ldexpf(((h >> 30) & 1u) ? -m : m, e);
            if (((h >> 20) & 15u) == 0u) v[q] = 0.0f;
        }
#pragma unroll
        for (uint32_t q = 0; q < 16u; ++q) atomicAdd(&acc[w][id[q]], v[q]);
        __builtin_amdgcn_wave_barrier();
        const uint32_t myid = id[0];
        float ref = 0.0f;
        for (uint32_t q = 0; q < 16u; ++q)
            for (uint32_t l = 0; l < kWave; ++l) {
                const uint32_t il = __shfl(id[q], l, kWave);
                const float vl = __shfl(v[q], l, kWave);
                if (il == myid) ref += vl;
            }
        if (f2u(acc[w][myid]) != f2u(ref)) ++bad;
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(violations, bad);
}

// ONE WAVE PER BUCKET, no sort: the bucket's records are taken in index order - its pieces in tile order, 64 consecutive
// positions to an instruction - and every record ADDS ITSELF to its voxel's three float words in LDS (ds_add_f32) and counts
// itself in a fourth.  A wave's LDS instructions execute in order and the lanes of one instruction are added in lane order
// (k_lds_float_order_probe), so each word ends up holding the sum of the voxel's points in ascending point index, starting
// from 0.0f, every addition rounded as v_add_f32 rounds it: the arithmetic of voxel_grid.hpp's "Fourth pass", with no
// counting sort, no staging and no barrier (the four waves of a block never meet).
// A wave has kVtSpan voxel ids.  A bucket that spans no more keys uses id = key - first key (the dense parts of a cloud: a
// bucket there is one coarse bin).  A wider one (up to kVtSpanMax keys: the sparse parts) first marks its keys in an
// occupancy map, one pass over its records, and uses id = rank of the key among the occupied ones; should there be more
// than kVtSpan voxels it takes them kVtSpan at a time, one pass each.  The occupied ids are compacted in order and divided by
// float(count); the centroid of the bucket's v-th voxel goes to scratch[first + v], first = the records of the lower
// buckets; binfo[b] = {first, records, ...} and bucket_pub[b] = voxels tell k_voxel_gather where they are.
// A bucket may hold any number of points: the sample's estimate only balances the work.
// Persistent waves: wave g takes buckets g, g + waves, ...; the rows of the next bucket are requested while this one is
// summed, the records come eight instructions at a time with the next eight in flight.
__global__ __launch_bounds__(kBlock, 7) void k_vt_accum(Frame* frames) {
    constexpr uint32_t S = kVtSpan;                       // voxel ids per pass
    constexpr uint32_t T = kVtMaxTiles;                   // pieces per bucket
    constexpr uint32_t kRows = T / kWave;                 // rows of `pre` per lane
    constexpr uint32_t kRound = 8u;                      // instructions (of 64 records) whose record addresses are found in one go
    constexpr uint32_t kChunk = 4u;                       // instructions per round of loads
    constexpr uint32_t kMapW = kVtSpanMax / 32u;          // occupancy words of a wide bucket
    constexpr uint32_t kMapPer = kMapW / kWave;           // per lane
    static_assert(S % kWave == 0u && kMapW % kWave == 0u && kRound % kChunk == 0u, "whole rows of ids and of occupancy words");
    __shared__ __attribute__((aligned(16))) float s_acc[kWavesPerBlock][4u * S];   // per wave: x, y, z sums and the counts, by voxel id
    __shared__ uint32_t s_src[kWavesPerBlock][T];                                   // per piece: its first record in rec_a
    __shared__ uint16_t s_dst[kWavesPerBlock][T + 2u];                              // per piece: its first position in the bucket (0xFFFF beyond the tiles)
    __shared__ uint32_t s_map[kWavesPerBlock][kMapW];                               // wide buckets: one bit per key of the span
    __shared__ uint16_t s_mpre[kWavesPerBlock][kMapW];                              //               occupied keys before the word
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    uint4* __restrict__ binfo = F.binfo;
    const Rec3* __restrict__ recs = reinterpret_cast<const Rec3*>(F.rec_a.get());
    const uint16_t* __restrict__ pre = F.tcnt;
    uint32_t* __restrict__ bucket_pub = F.bucket_pub;
    float4* __restrict__ scratch = F.rec_b;
    if (vt_off(st)) return;
    const uint32_t nb = st->nbuckets, ts = F.tstride;
    const uint32_t ntiles = (st->n_in + kVtTile - 1u) / kVtTile;
    if (blockIdx.x == 0 && threadIdx.x == 0 && st->n_valid == 0u) atomicOr(&st->err, (uint32_t)kErrReplanVt);   // (an empty cloud: the streaming path knows what to do)
    const uint32_t w = threadIdx.x / kWave, lane = lane_id();
    float* ax = s_acc[w]; float* ay = ax + S; float* az = ay + S;
    uint32_t* an = reinterpret_cast<uint32_t*>(az + S);
    uint32_t* src = s_src[w]; uint16_t* dst = s_dst[w];
    uint32_t* omap = s_map[w]; uint16_t* opre = s_mpre[w];
    uint32_t tpad = 2u;
    while (tpad < ntiles + 1u) tpad <<= 1;                // search range: a power of two above the tiles (<= T: the host keeps ntiles < T)
    VoxelKey key; key.init(st);                           // (k_vt_sort has checked every record against the box)
    const uint32_t step = gridDim.x * kWavesPerBlock;
    const uint32_t b_end = nb;
    uint32_t b = blockIdx.x * kWavesPerBlock + w;
    if (b >= b_end) return;
    // rows of bucket b (clamped, unconditional): where it starts and ends in every tile's output
    uint32_t ra[kRows], re[kRows];
    uint4 rinfo;
    auto rows = [&](uint32_t bb) {
        const uint32_t bc = min(bb, nb - 1u);
        rinfo = binfo[bc];
#pragma unroll
        for (uint32_t q = 0; q < kRows; ++q) {
            const uint32_t tc = min(lane + q * kWave, ntiles - 1u);
            ra[q] = pre[(size_t)bc * ts + tc]; re[q] = pre[(size_t)(bc + 1u) * ts + tc];
        }
    };
    rows(b);
#ifdef PCOP_STAMPS
    unsigned long long tt[6] = {0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#define ACC_T(i) do { const unsigned long long t_now = __builtin_readcyclecounter(); tt[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define ACC_T(i) do { } while (0)
#endif
    for (; b < b_end; b += step) {
        ACC_T(5);
        const uint32_t key_lo = rinfo.z, span = rinfo.w;
        // the piece table: per tile (in order) the first record and the first position; the bucket's points and its place
        uint32_t cnt = 0u, first = 0u;
#pragma unroll
        for (uint32_t q = 0; q < kRows; ++q) {
            const uint32_t t = lane + q * kWave;
            const bool there = t < ntiles;
            const uint32_t len = there ? re[q] - ra[q] : 0u;
            const uint32_t incl = wave_incl_scan(len);
            if (t < tpad) { src[t] = t * kVtTile + ra[q]; dst[t] = there ? (uint16_t)min(cnt + incl - len, 0xFFFEu) : (uint16_t)0xFFFFu; }
            cnt += (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
            first += wave_sum(there ? ra[q] : 0u);
        }
        if (b + step < b_end) rows(b + step);   // in flight while this bucket is summed
        ACC_T(0);
        const bool bad = span > kVtSpanMax || span == 0u || cnt > 0xFFFEu;   // never: k_bin_plan keeps the span, a tile holds 8192 points
        if (bad || cnt == 0u) {
            if (lane == 0) {
                if (bad) atomicOr(&st->err, (uint32_t)kErrReplanVt);
                binfo[b] = make_uint4(first, 0u, key_lo, span); bucket_pub[b] = 0u;
            }
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        // One pass over the bucket's records in index order: instruction after instruction, lanes in order.  A float add
        // to LDS holds a wave's LDS queue for hundreds of cycles (the lanes go one by one), and an LDS read issued behind it
        // waits: so the record addresses of sixteen instructions are found first - each lane looks up the piece of each of
        // its sixteen positions, sixteen searches side by side, every round one LDS read each - and only then come the
        // loads (four instructions at a time, the next four in flight) and the adds, with no LDS read in between.
        bool wrong = false;
        auto pass = [&](auto&& each) {
            for (uint32_t r0 = 0; r0 < cnt; r0 += kRound * kWave) {
                uint32_t at[kRound];
                {
                    uint32_t lo[kRound];
#pragma unroll
                    for (uint32_t j = 0; j < kRound; ++j) { at[j] = min(r0 + j * kWave + lane, cnt - 1u); lo[j] = 0u; }   // (clamped, unconditional)
                    for (uint32_t h = tpad >> 1; h; h >>= 1) {
#pragma unroll
                        for (uint32_t j = 0; j < kRound; ++j) lo[j] += ((uint32_t)dst[lo[j] + h] <= at[j]) ? h : 0u;
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kRound; ++j) at[j] = src[lo[j]] + (at[j] - (uint32_t)dst[lo[j]]);
                }
                Rec3 rec[kChunk], recn[kChunk];
#pragma unroll
                for (uint32_t j = 0; j < kChunk; ++j) recn[j] = recs[at[j]];
#pragma unroll
                for (uint32_t c = 0; c < kRound; c += kChunk) {
                    if (r0 + c * kWave >= cnt) break;
#pragma unroll
                    for (uint32_t j = 0; j < kChunk; ++j) rec[j] = recn[j];
                    if (c + kChunk < kRound) {
#pragma unroll
                        for (uint32_t j = 0; j < kChunk; ++j) recn[j] = recs[at[c + kChunk + j]];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kChunk; ++j) {
                        const uint32_t lk = key(rec[j]) - key_lo;
                        const bool act = r0 + (c + j) * kWave + lane < cnt;
                        if (act && lk >= span) wrong = true;   // never: the tile sort put the record here by this key
                        else if (act) each(rec[j], lk);
                    }
                }
            }
        };
        auto clear = [&](uint32_t ids) {
            for (uint32_t i = lane * 4u; i < ids; i += kWave * 4u) {   // (S is a multiple of 4: the stores stay inside the planes)
                *reinterpret_cast<float4*>(ax + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *reinterpret_cast<float4*>(ay + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *reinterpret_cast<float4*>(az + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *reinterpret_cast<uint4*>(an + i) = make_uint4(0u, 0u, 0u, 0u);
            }
            __builtin_amdgcn_wave_barrier();
        };
        // the occupied ids below `ids` in order -> centroids at scratch[at ...]; returns how many.  Row r = ids 64 r .. 64 r + 63,
        // one to a lane: a ballot numbers the occupied ones.
        auto emit = [&](uint32_t ids, uint32_t at) -> uint32_t {
            __builtin_amdgcn_wave_barrier();
            uint32_t v0 = at;
            for (uint32_t r0 = 0; r0 < ids; r0 += kWave) {
                const uint32_t i = r0 + lane;
                const uint32_t n = i < ids ? an[i] : 0u;
                const uint64_t occ = __ballot(n != 0u);
                if (n) {
                    const float fc = (float)n;
                    scratch[v0 + (uint32_t)__popcll(occ & lanemask_lt())] = make_float4(ax[i] / fc, ay[i] / fc, az[i] / fc, 0.0f);
                }
                v0 += (uint32_t)__popcll(occ);
            }
            __builtin_amdgcn_wave_barrier();
            return v0 - at;
        };
        uint32_t nvox = 0u;
        if (span <= S) {   // id = key - first key
            clear(span);
            ACC_T(1);
            pass([&](const Rec3& r, uint32_t lk) {
                atomicAdd(ax + lk, r.x); atomicAdd(ay + lk, r.y); atomicAdd(az + lk, r.z);
                atomicAdd(an + lk, 1u);
            });
            ACC_T(2);
            nvox = emit(span, first);
            ACC_T(3);
#ifdef PCOP_STAMPS
            tt[5] += 0; 
#endif
        } else {           // id = rank of the key among the bucket's occupied keys
            const uint32_t words = (span + 31u) / 32u;
            for (uint32_t i = lane; i < words; i += kWave) omap[i] = 0u;
            __builtin_amdgcn_wave_barrier();
            pass([&](const Rec3&, uint32_t lk) { atomicOr(omap + (lk >> 5), 1u << (lk & 31u)); });
            __builtin_amdgcn_wave_barrier();
            {   // lane l owns words l * kMapPer .. : occupied keys before each
                uint32_t wv[kMapPer], c = 0u;
#pragma unroll
                for (uint32_t k = 0; k < kMapPer; ++k) { const uint32_t wi = lane * kMapPer + k; wv[k] = wi < words ? omap[wi] : 0u; c += __popc(wv[k]); }
                const uint32_t incl = wave_incl_scan(c);
                nvox = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
                uint32_t run = incl - c;
#pragma unroll
                for (uint32_t k = 0; k < kMapPer; ++k) { const uint32_t wi = lane * kMapPer + k; if (wi < words) opre[wi] = (uint16_t)run; run += __popc(wv[k]); }
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t q0 = 0; q0 < nvox; q0 += S) {   // (nearly always once: a wide bucket holds few points)
                const uint32_t ids = min(S, nvox - q0);
                clear(ids);
                pass([&](const Rec3& r, uint32_t lk) {
                    const uint32_t wi = lk >> 5;
                    const uint32_t id = (uint32_t)opre[wi] + __popc(omap[wi] & ((1u << (lk & 31u)) - 1u)) - q0;
                    if (id < ids) {
                        atomicAdd(ax + id, r.x); atomicAdd(ay + id, r.y); atomicAdd(az + id, r.z);
                        atomicAdd(an + id, 1u);
                    }
                });
                emit(ids, first + q0);
            }
            ACC_T(4);
        }
        if (__any(wrong)) {
            if (lane == 0) { atomicOr(&st->err, (uint32_t)kErrInternal); binfo[b] = make_uint4(first, 0u, key_lo, span); bucket_pub[b] = 0u; }
            continue;
        }
        if (lane == 0) { binfo[b] = make_uint4(first, cnt, key_lo, span); bucket_pub[b] = nvox; }
        __builtin_amdgcn_wave_barrier();   // the next bucket clears and fills the same LDS
    }
#ifdef PCOP_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 7 && blockIdx.y == 0) { unsigned long long* q = reinterpret_cast<unsigned long long*>(F.hsub.get()); for (int i = 0; i < 6; ++i) q[32 + 24 + i] = tt[i]; }
#endif
}

// Output rank of every bucket's first voxel (its voxels follow): one block per frame scans the buckets' voxel counts; M.
// k_voxel_gather then moves the centroids.
__global__ __launch_bounds__(kBlock) void k_vt_scan(Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (vt_off(st)) return;
    const uint32_t* __restrict__ bucket_pub = F.bucket_pub;
    uint32_t* __restrict__ base = F.tbase;   // (the streaming path's tile offsets: idle here)
    const uint32_t nb = st->nbuckets;
    uint32_t carry = 0u;
    for (uint32_t b0 = 0; b0 < nb; b0 += kBlock * 4u) {
        uint32_t v[4], sum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const uint32_t b = b0 + threadIdx.x * 4u + k; v[k] = b < nb ? bucket_pub[b] : 0u; sum += v[k]; }
        uint32_t tot;
        uint32_t run = carry + block_excl_scan(sum, s_w, &tot);
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const uint32_t b = b0 + threadIdx.x * 4u + k; if (b < nb) base[b] = run; run += v[k]; }
        carry += tot;
    }
    if (threadIdx.x == 0) finish_voxel(st, carry);
}

// PCL's own grid into DevState, once nothing forms keys any more (k_emit's first block, before it hands DevState to the host):
// the true bounds from the partials of the tiles of DevState::box_tile points, min_b / div_b / divb_mul as
// VoxelGrid::applyFilter derives them (voxel_setup does the same where a bounds pass comes first).  The whole block calls.
__device__ __forceinline__ void vt_true_grid(Frame& F, MinMaxPartial* s_p) {
    DevState* st = &F.st;
    const MinMaxPartial* __restrict__ partials = F.partials;
    const uint32_t tile = max(st->box_tile, 1u);
    const uint32_t ntiles = (st->n_in + tile - 1u) / tile;
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    for (uint32_t i = threadIdx.x; i < ntiles; i += kBlock) {
        const MinMaxPartial p = partials[i];   // (voxel coordinates relative to the box: box_partial)
        if (p.valid) for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], f2u(p.mn[a])); mx[a] = max(mx[a], f2u(p.mx[a])); }
    }
    for (int a = 0; a < 3; ++a) { s_p[threadIdx.x].mn[a] = u2f(mn[a]); s_p[threadIdx.x].mx[a] = u2f(mx[a]); }
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            for (int a = 0; a < 3; ++a) {
                s_p[threadIdx.x].mn[a] = u2f(min(f2u(s_p[threadIdx.x].mn[a]), f2u(s_p[threadIdx.x + d].mn[a])));
                s_p[threadIdx.x].mx[a] = u2f(max(f2u(s_p[threadIdx.x].mx[a]), f2u(s_p[threadIdx.x + d].mx[a])));
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && st->n_valid) {
        // min_b = floor(min_p * inv) = the box's origin + the smallest coordinate; vmin / vmax keep the box's bounds (a superset)
        int32_t div[3];
        for (int a = 0; a < 3; ++a) {
            const int32_t lo = st->min_b[a] + (int32_t)f2u(s_p[0].mn[a]), hi = st->min_b[a] + (int32_t)f2u(s_p[0].mx[a]);
            st->min_b[a] = lo; div[a] = hi - lo + 1; st->div_b[a] = div[a];
        }
        st->mul[0] = 1u; st->mul[1] = (uint32_t)div[0]; st->mul[2] = (uint32_t)div[0] * (uint32_t)div[1];
        const uint64_t cells = (uint64_t)(uint32_t)div[0] * (uint32_t)div[1] * (uint32_t)div[2];
        st->plan_voxel.key_bits = cells > 0xFFFFFFFFull ? 32u : bits_for(cells);
    }
}

}  // namespace pcop

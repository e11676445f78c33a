// pcop_voxel.hpp - voxel grid: frame init, bounds, the streaming path and the sort path (replaces pcl::VoxelGrid::applyFilter).
// Part of pcop_device.hpp (included from there, in this order; not a stand-alone header).
#pragma once

namespace pcop {

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

// The kItems records of a thread (record j at wbase + j * 64), all requested before any is used: the indices are
// clamped instead of tested, and the choice of source is made once, outside the loop (a load under a per-record
// branch is waited for at the branch's join, which would cost one HBM round trip per record).
__device__ __forceinline__ void load_records(const RecSrc& src, uint32_t wbase, uint32_t count, Rec3* rec) {
    const uint32_t last = count - 1u;   // count >= 1: the caller's tile starts below count
    if (src.r) {
#pragma unroll
        for (int j = 0; j < kItems; ++j) rec[j] = src.r[min(wbase + j * kWave, last)];
    } else {
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
            const uint32_t i = min(wbase + j * kWave, last);
            rec[j].x = src.x[i]; rec[j].y = src.y[i]; rec[j].z = src.z[i];
        }
    }
}

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
constexpr uint32_t kSpinLimit = 1u << 24;   // polls of a bounded wait (the default of FrameArgs::spin_limit; the kernels next to the path keep it as a constant)

__device__ __forceinline__ void frame_init(Frame& F) {   // one thread: the mailbox -> DevState
    DevState* st = &F.st;
    {
        const FrameArgs a = *F.args;
        DevState s;
        memset(&s, 0, sizeof(s));
        s.n_in = a.n_dev.get() ? min(*a.n_dev, a.n) : a.n; s.is_dense = a.is_dense; s.kind = a.kind; s.trivial = a.trivial;
        s.emit_cloud = a.emit_cloud; s.out_layout = a.out_layout; s.out_soa_stride = a.out_soa_stride;
        s.inv_leaf = a.inv_leaf; s.r2 = a.r2; s.inv_cell = a.inv_cell;
        s.min_cluster = a.min_cluster; s.max_cluster = a.max_cluster;
        s.src_x = a.src_x; s.src_y = a.src_y; s.src_z = a.src_z;
        s.debug_skip_union = a.debug_skip_union; s.spin_limit = a.spin_limit ? a.spin_limit : kSpinLimit;   // (0: a caller that leaves the field alone gets the default)
        s.nb_cap = a.nb_cap; s.gbits = a.gbits; s.tile = a.tile ? a.tile : kBinTile; s.cgrid_req = a.cell_grid;
        if (a.kind == kFrameClusterOnly) {   // no voxel stage: cluster the input cloud as it is
            s.c_from_input = 1; s.c_n = s.n_in; s.n_valid = s.n_in;
            uint32_t cap = 1024;
            while (cap < 2u * s.n_in && cap < 0x80000000u) cap <<= 1;
            s.hash_mask = cap - 1u;
        }
        *st = s;
    }
}

__global__ void k_frame_init(Frame* frames) {
    if (threadIdx.x == 0 && blockIdx.x == 0) frame_init(frames[blockIdx.y]);
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
    if (stream) {   // coarse bins of the streaming path: the top gbits bits of the key
        st->vpath = 1u;
        st->gbits = min(kb, st->gbits);   // on entry: the host's hint
        st->cshift = kb - st->gbits;
        if ((kBucketCells >> st->cshift) == 0u) atomicOr(&st->err, (uint32_t)kErrReplanBins);   // one bin spans more cells than a bucket may
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

// from_cen = 0: the input cloud, straight from the mailbox; 1: the cloud the cell stage will see (voxel centroids or, after a
// pass-through, the input), from DevState - the bounds pass of the statistical outlier removal inside the node chain.
__global__ __launch_bounds__(kBlock) void k_minmax(Frame* frames, uint32_t from_cen) {
    __shared__ MinMaxPartial s_p[kBlock];
    Frame& F = frames[blockIdx.y];
    MinMaxPartial* partials = F.partials;
    // Straight from the mailbox (pinned host memory): DevState is written by k_voxel_setup, after this kernel, so that
    // a frame needs no separate init launch.
    __shared__ FrameArgs s_args;
    if (threadIdx.x == 0) s_args = *F.args;
    __syncthreads();
    if (!from_cen && s_args.kind == kFrameClusterOnly && !s_args.cell_grid) return;   // bounds of a cluster-only frame: for the dense cell grid
    const DevState* st = &F.st;
    const bool cen = from_cen && !st->c_from_input;
    const float* __restrict__ x = from_cen ? (cen ? F.cen_x.get() : st->src_x.get()) : s_args.src_x.get();
    const float* __restrict__ y = from_cen ? (cen ? F.cen_y.get() : st->src_y.get()) : s_args.src_y.get();
    const float* __restrict__ z = from_cen ? (cen ? F.cen_z.get() : st->src_z.get()) : s_args.src_z.get();
    const uint32_t n = from_cen ? (st->err ? 0u : st->c_n) : (s_args.n_dev.get() ? min(*s_args.n_dev, s_args.n) : s_args.n);
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
    uint32_t i = gtid;
    for (; i + gsz < n4; i += 2u * gsz) {   // two trips' worth of loads in flight
        const float4 a = x4[i], b = y4[i], c = z4[i];
        const float4 d = x4[i + gsz], e = y4[i + gsz], f = z4[i + gsz];
        take(a.x, b.x, c.x); take(a.y, b.y, c.y); take(a.z, b.z, c.z); take(a.w, b.w, c.w);
        take(d.x, e.x, f.x); take(d.y, e.y, f.y); take(d.z, e.z, f.z); take(d.w, e.w, f.w);
    }
    for (; i < n4; i += gsz) {
        const float4 a = x4[i], b = y4[i], c = z4[i];
        take(a.x, b.x, c.x); take(a.y, b.y, c.y); take(a.z, b.z, c.z); take(a.w, b.w, c.w);
    }
    for (uint32_t j = n4 * 4 + gtid; j < n; j += gsz) take(x[j], y[j], z[j]);
    MinMaxPartial p;
    p.mn[0] = mn0; p.mn[1] = mn1; p.mn[2] = mn2; p.mx[0] = mx0; p.mx[1] = mx1; p.mx[2] = mx2;
    p.valid = valid; p.nonfinite = bad;
    s_p[threadIdx.x] = p;
    reduce_partials_lds(s_p);
    if (threadIdx.x == 0) partials[blockIdx.x] = s_p[0];
}

// one block per frame: reduce the block partials and derive the voxel grid
// flags: bit 0 = the streaming voxel path is requested, bit 1 = write DevState from the mailbox first (no k_frame_init
// launch before), bit 2 = set up the dense cell grid as well (no k_cgrid_setup launch after: it needs the bounds only).
__device__ __forceinline__ void cgrid_setup_block(Frame& F, uint32_t npts_max, bool empty);

__global__ __launch_bounds__(kBlock) void k_voxel_setup(Frame* frames, uint32_t nblocks, uint32_t npass, uint32_t maxbits, uint32_t flags) {
    __shared__ MinMaxPartial s_p[kBlock];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const uint32_t stream = flags & 1u;
    if (flags & 2u) {
        if (threadIdx.x == 0) frame_init(F);
        __syncthreads();
    }
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
        if (flags & 4u) { __syncthreads(); cgrid_setup_block(F, st->n_in, st->n_in == 0u); }
        return;
    }
    if (threadIdx.x == 0) voxel_setup(st, s_p[0], npass, maxbits, stream != 0u);
    if (flags & 4u) { __syncthreads(); cgrid_setup_block(F, st->n_in, s_p[0].valid == 0u); }
    if (stream) return;
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
__device__ __forceinline__ void radix_hist_on_box(const Frame& F, uint32_t* s_h, uint32_t tile_base, const PassInfo& pi, const RecSrc& src);

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
    if (pass == 0 && st->sbox) {   // keys on the host's box: this pass takes the bounds too
        radix_hist_on_box<MAXB>(F, s_h, tile_base, pi, src);
    } else {
    Rec3 rec[kItems];
    load_records(src, wbase, pi.count, rec);
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < pi.count) {
            const Rec3 r = rec[j];
            uint32_t d;
            if (pi.drop_nonfinite && !finite3(r.x, r.y, r.z)) d = pi.nb - 1;
            else d = (key(r) >> pi.shift) & mask;
            atomicAdd(&s_h[d], 1u);
        }
    }
    }
    __syncthreads();
    // bin-major table (table[bin * ntiles + tile]): the scan kernel then reads and writes whole rows
    const uint32_t ntiles = (pi.count + kTile - 1) / kTile;
    for (uint32_t b = threadIdx.x; b < pi.nb; b += kBlock) table[(size_t)b * ntiles + blockIdx.x] = s_h[b];
}

// which = 0: voxel partition pass, 2: CSR partition pass
__device__ __forceinline__ bool table_dims(const DevState* st, int which, int pass, uint32_t* ntiles, uint32_t* nb) {
    if (st->err) return false;
    if (which == 2) {
        if (st->trivial || st->kind == kFrameVoxelOnly) return false;
        *nb = (1u << st->plan_csr.bits[pass]) + 1u;
        *ntiles = (st->c_n + kTile - 1) / kTile;
        return st->n_kept != 0;
    }
    if (voxel_stage_off(st)) return false;
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
    load_records(src, wbase, pi.count, rec);
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        const bool active = i < pi.count;
        uint32_t d = 0;
        if (active) {
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
#ifndef PCOP_REDUCE_BPB
#define PCOP_REDUCE_BPB 2
#endif
#ifndef PCOP_REDUCE_AOS
#define PCOP_REDUCE_AOS 0   // 1 (tuning build, tools/ab_libs.sh): k_bucket_reduce stages a bucket as 12-byte records instead of three coordinate planes - built at the end of round 4, not measured yet
#endif
constexpr uint32_t kReduceBucketsPerBlock = PCOP_REDUCE_BPB;   // k_bucket_reduce: consecutive buckets per block (2: measured against 1, 3 and 4, tools/ab_build.sh)

// mark_cells: a chain frame on the dense cell grid (set up by k_voxel_setup): every centroid marks its grid cell as it is
// written, so the cell stage has no k_cell_mark launch (k_voxel_gather does the same on the streaming path).
__global__ __launch_bounds__(kBlock) void k_voxel_reduce(Frame* frames, uint32_t npass, uint32_t mark_cells) {
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
    const bool cells = mark_cells && !cgrid_off(st);
    uint8_t* __restrict__ cflag = F.cflag; uint32_t* __restrict__ pt_pos = F.pt_root;
    CellMarker mark;
    if (cells) mark.init(st);
    if (st->passthrough) {   // "output = *input_": the voxel cloud is the input cloud, every point kept
        if (blockIdx.x == 0 && threadIdx.x == 0) finish_voxel(st, st->n_in);
        if (cells) {   // the cell stage sees the input cloud: each tile marks its points
            const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
            const uint32_t nin = st->n_in;
            for (uint32_t i = blockIdx.x * kScanTile + threadIdx.x; i < min(nin, (blockIdx.x + 1u) * (uint32_t)kScanTile); i += kBlock)
                pt_pos[i] = mark(st, cflag, x[i], y[i], z[i]);
        }
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
    Rec3 staged[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) staged[k] = rec[base + min((uint32_t)(k * kBlock) + threadIdx.x, cnt - 1u)];   // loads first
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t j = k * kBlock + threadIdx.x;
        if (j < cnt) { s_rec[j] = staged[k]; s_key[j + 1] = key(staged[k]); }
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
            if (++spins > st->spin_limit) { atomicOr(&st->err, (uint32_t)kErrInternal); v = 1u; break; }
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
        if (h < tot) {
            cx[before + h] = ox[k]; cy[before + h] = oy[k]; cz[before + h] = oz[k];
            if (cells) pt_pos[before + h] = mark(st, cflag, ox[k], oy[k], oz[k]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Streaming voxel path (default; a frame it does not fit is re-run on the sort path).
//
// PCL sorts the points by voxel key and sums every run in order.  What that needs is every voxel's points
// together, in index order, and the voxels in key order - not a full sort of the cloud in HBM:
//   k_bin_hist      : per tile of kBinTile points, the counts of the coarse key bins (key >> cshift, at most
//                     2^kBinBitsMax of them) in an LDS histogram -> ctab[tile][bin]
//   k_bin_colsum    : bin totals over the tiles
//   k_bin_plan      : one block per frame: buckets = runs of consecutive bins with at most kBucketCap points and
//                     kBucketCells cells each, packed greedily -> bin2bucket, binfo
//   k_bin_tilecnt   : per (tile, bucket) the tile's points in the bucket -> tcnt (bucket-major)
//   k_bin_scatter   : ONE stable partition of packed 12-byte xyz records by bucket (ranks from lane-ordered LDS atomics or peer
//                     masks, per-wave counts, chunks of kScatChunk points staged in LDS so that stores are contiguous); a tile adds up
//                     the counts of the lower tiles for itself
//   k_bucket_reduce : a bucket lives in LDS: occupancy bits of its key range -> local voxel ids (rank of the key),
//                     stable counting sort by id, one lane per voxel sums its run in order from 0.0f and divides
//                     by float(count); the centroids stay at the bucket's place, the bucket leaves its voxel count
//   k_voxel_gather  : (pcop_cells_grid.hpp) the voxels of the lower buckets added up -> output rank of the bucket's first
//                     voxel, M; centroids to their ranks = voxel key order, grid cells marked on the way
// Same float operations in the same order as the sort path.  No global atomics, no scattered global access: the
// cloud is read three times (bounds, histogram, partition), written once and read once as 12-byte records.
// Any bucket plan that respects the two limits gives the same result, so the host may tune gbits freely.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool stream_off(const DevState* st) {
    return voxel_stage_off(st) || !st->vpath || (st->n_valid == 0 && !st->sbox);   // (on the host's box the valid points are counted as the cloud is read)
}

// Diagnostic build only (-DPCOP_STAMPS): the shader clock at phase borders of one block of a kernel, into a buffer that
// no kernel reads (the min/max partials, dead after k_voxel_setup).  tools/stamps.py prints the differences.
#ifdef PCOP_STAMPS
#define PCOP_STAMP(F, kern, idx, blk) do { if (threadIdx.x == 0 && blockIdx.x == (blk) && blockIdx.y == 0) \
    reinterpret_cast<unsigned long long*>((F).partials.get())[(kern) * 32 + (idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define PCOP_STAMP(F, kern, idx, blk) do { } while (0)
#endif

// The same pass when the keys are formed against the host's box (DevState::sbox): there is no bounds pass before it, so it
// also takes the true bounds of its tile (min / max are exact in any order), counts the valid points and checks every point
// against the box (one outside: kErrReplanBox, the host re-runs the batch with the bounds pass).
struct BoxKey;
__device__ __forceinline__ void bin_hist_on_box(Frame& F, uint32_t* s_hist, uint32_t tile_base, uint32_t tile, uint32_t n, uint32_t G, uint32_t cs);

__global__ __launch_bounds__(kHistThreads) void k_bin_hist(Frame* frames) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_dyn);
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    uint16_t* __restrict__ ctab = F.ctab;
    if (stream_off(st)) return;
    const uint32_t n = st->n_in, tile = st->tile;
    const uint32_t tile_base = blockIdx.x * tile;
    if (tile_base >= n) return;
    const uint32_t G = 1u << st->gbits, cs = st->cshift;
    for (uint32_t b = threadIdx.x; b < G; b += kHistThreads) s_hist[b] = 0u;
    __syncthreads();
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    const bool drop = st->is_dense == 0;
    VoxelKey key; key.init(st);
    bool bad = false;
    constexpr int kRound = kBinTileSmall / kHistThreads;   // records in flight per thread: 8
    if (st->sbox) { bin_hist_on_box(F, s_hist, tile_base, tile, n, G, cs); return; }
    for (uint32_t r0 = 0; r0 < tile / kHistThreads; r0 += kRound) {
        Rec3 r[kRound];
#pragma unroll
        for (int k = 0; k < kRound; ++k) {   // clamped, unconditional loads
            const uint32_t i = min(tile_base + (r0 + k) * kHistThreads + threadIdx.x, n - 1u);
            r[k].x = x[i]; r[k].y = y[i]; r[k].z = z[i];
        }
#pragma unroll
        for (int k = 0; k < kRound; ++k) {
            const uint32_t i = tile_base + (r0 + k) * kHistThreads + threadIdx.x;
            if (i < n && (!drop || finite3(r[k].x, r[k].y, r[k].z))) {
                const uint32_t bin = key(r[k]) >> cs;
                if (bin < G) atomicAdd(&s_hist[bin], 1u); else bad = true;
            }
        }
    }
    if (bad) atomicOr(&st->err, (uint32_t)kErrInternal);   // a key beyond the grid: never (the keys are those PCL computes)
    __syncthreads();
    uint16_t* __restrict__ row = ctab + (size_t)blockIdx.x * G;
    for (uint32_t b = threadIdx.x; b < G; b += kHistThreads) row[b] = (uint16_t)s_hist[b];   // a tile has at most 2^15 points (kBinTileLarge): fits
}

// Bin totals over the tiles: a block takes 64 bins (a wave reads 128 contiguous bytes of a row), its four waves
// every fourth tile each.
__global__ __launch_bounds__(kBlock) void k_bin_colsum(const Frame* frames) {
    __shared__ uint32_t s_part[kBlock];
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint16_t* __restrict__ ctab = F.ctab;
    uint32_t* __restrict__ binsum = F.binsum;
    if (stream_off(st)) return;
    const uint32_t G = 1u << st->gbits;
    const uint32_t bin = blockIdx.x * kWave + lane_id();
    if (blockIdx.x * kWave >= G) return;
    const uint32_t ntiles = (st->n_in + st->tile - 1u) / st->tile;
    const uint32_t w = threadIdx.x / kWave;
    const uint32_t bc = min(bin, G - 1u);
    uint32_t c = 0u;
    for (uint32_t t0 = w; t0 < ntiles; t0 += 4u * 16u) {
        uint32_t v[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) v[k] = ctab[(size_t)min(t0 + 4u * k, ntiles - 1u) * G + bc];   // clamped, unconditional
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) c += (t0 + 4u * k < ntiles) ? v[k] : 0u;
    }
    s_part[threadIdx.x] = c;
    __syncthreads();
    if (w == 0u && bin < G) binsum[bin] = s_part[lane_id()] + s_part[kWave + lane_id()] + s_part[2 * kWave + lane_id()] + s_part[3 * kWave + lane_id()];
}

// Dynamic LDS of k_bin_plan for G bins.
__host__ __device__ constexpr size_t plan_lds_bytes(uint32_t G) { return ((size_t)G + (G >> 4) + 24u) * 4u + ((size_t)G + 8u) * 2u; }

// One block per frame: the bucket plan.  Thread t owns the `per` consecutive bins starting at t * per.
// The bins are cut into groups of bmax (a bucket may span kBucketCells cells), every group into windows of kBucketCap
// points by the running sum at the start of a bin; the bins that start in one window hold less than one capacity
// plus the window's last bin, so a window is one bucket, or two when the last bin has to stand alone.  Constant work
// per bin, no searches; windows and buckets are numbered by prefix sums.  Segments without points are dropped.
// s_P is padded by one word per 16 so that a thread's 16 consecutive words fall on different LDS banks than its neighbours'.
__global__ __launch_bounds__(kBinThreads) void k_bin_plan(Frame* frames) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    constexpr int NW = kBinThreads / kWave;
    constexpr int kPerMax = (1 << kBinBitsMax) / kBinThreads;   // 16
    __shared__ uint32_t s_w[NW];
    __shared__ uint32_t s_max[NW];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const uint32_t* __restrict__ binsum = F.binsum;
    uint16_t* __restrict__ bin2bucket = F.bin2bucket;
    uint4* __restrict__ binfo = F.binfo;
    uint32_t* __restrict__ bucket_pub = F.bucket_pub;
    if (stream_off(st)) return;
    if (st->sbox && st->n_valid == 0u) {   // (k_bin_hist found nothing valid in the cloud: an empty output, as voxel_setup makes it)
        if (threadIdx.x == 0) { st->nbuckets = 0u; finish_voxel(st, 0); }
        return;
    }
    const uint32_t G = 1u << st->gbits, cs = st->cshift;
    const uint32_t per = G >= kBinThreads ? G / kBinThreads : 1u;
    uint32_t* s_Pp = reinterpret_cast<uint32_t*>(s_dyn);                                   // points before bin i (i = 0 .. G), padded
    uint16_t* s_first = reinterpret_cast<uint16_t*>(s_dyn + ((size_t)G + (G >> 4) + 24u) * 4u);   // first bin of every window, then of every segment
    auto P = [&](uint32_t i) -> uint32_t& { return s_Pp[i + (i >> 4)]; };
    const uint32_t b0 = threadIdx.x * per;
    const bool mine = b0 < G;
    PCOP_STAMP(F, 0, 0, 0);
    for (uint32_t q = threadIdx.x; q < kNbMax; q += kBinThreads) bucket_pub[q] = 0u;   // k_bucket_reduce publishes here
    uint32_t c[kPerMax];
    if (per == 16u) {   // the usual sizes: 16-byte loads
        const uint4* q = reinterpret_cast<const uint4*>(binsum + b0);
        const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        c[0] = v0.x; c[1] = v0.y; c[2] = v0.z; c[3] = v0.w; c[4] = v1.x; c[5] = v1.y; c[6] = v1.z; c[7] = v1.w;
        c[8] = v2.x; c[9] = v2.y; c[10] = v2.z; c[11] = v2.w; c[12] = v3.x; c[13] = v3.y; c[14] = v3.z; c[15] = v3.w;
    } else {
#pragma unroll
        for (int k = 0; k < kPerMax; ++k) c[k] = binsum[min(b0 + (uint32_t)k, G - 1u)];   // clamped, unconditional: one round trip for all
#pragma unroll
        for (int k = 0; k < kPerMax; ++k) c[k] = (mine && (uint32_t)k < per) ? c[k] : 0u;
    }
    uint32_t csum = 0u, cmax = 0u;
#pragma unroll
    for (int k = 0; k < kPerMax; ++k) { csum += c[k]; cmax = max(cmax, c[k]); }
    uint32_t total;
    const uint32_t run0 = block_excl_scan_n<NW>(csum, s_w, &total);
    cmax = wave_max(cmax);
    if (lane_id() == 0) s_max[threadIdx.x / kWave] = cmax;
    if (mine) {
        uint32_t r = run0;
#pragma unroll
        for (int k = 0; k < kPerMax; ++k)
            if ((uint32_t)k < per) { P(b0 + k) = r; r += c[k]; }
    }
    if (threadIdx.x == 0) { P(G) = total; P(G + 1u) = total; }
    __syncthreads();
    PCOP_STAMP(F, 0, 1, 0);
    uint32_t maxbin = 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) maxbin = max(maxbin, s_max[i]);
    constexpr uint32_t kCap = kBucketCap;
    if (threadIdx.x == 0) st->maxbin = maxbin;
    const uint32_t bmax = min(G, kBucketCells >> cs);   // bins a bucket may span
    // A bin with more points than a bucket: finer bins are asked for; when there are none finer (a dense sheet of points
    // lying in a few rows of the key space) the bin becomes a bucket of its own - the window rule below does that by
    // itself - which k_heavy_split cuts into buckets along the keys before k_bucket_reduce sees it.
    const bool finest = st->gbits >= kBinBitsMax || cs == 0u;
    if (bmax == 0u || (maxbin > kCap && !(finest && (*F.args).heavy_ok && maxbin <= kHeavyMax))) {
        if (threadIdx.x == 0) atomicOr(&st->err, (uint32_t)kErrReplanBins);
        return;
    }
    // window of a bin: (points of its group before it) / capacity.  bmax is a power of two (both G and the cell limit are).
    auto window = [&](uint32_t b, uint32_t pb) -> uint32_t { return (pb - P(b & ~(bmax - 1u))) / kCap; };
    auto wstart = [&](uint32_t b, uint32_t pb, uint32_t bprev_p) -> bool {   // a window starts at bin b (b < G)
        return (b & (bmax - 1u)) == 0u || window(b, pb) != window(b - 1u, bprev_p);
    };
    uint32_t wbits = 0u, nws = 0u;
    if (mine) {
        uint32_t r = run0, rprev = b0 ? P(b0 - 1u) : 0u;
#pragma unroll
        for (int k = 0; k < kPerMax; ++k) {
            if ((uint32_t)k < per) {
                if (wstart(b0 + k, r, rprev)) { wbits |= 1u << k; ++nws; }
                rprev = r; r += c[k];
            }
        }
    }
    uint32_t nwin;
    const uint32_t win0 = block_excl_scan_n<NW>(nws, s_w, &nwin);   // windows that start before bin b0
    if (mine) {
        uint32_t wi = win0;
#pragma unroll
        for (int k = 0; k < kPerMax; ++k)
            if ((uint32_t)k < per && ((wbits >> k) & 1u)) s_first[wi++] = (uint16_t)(b0 + k);
    }
    __syncthreads();
    PCOP_STAMP(F, 0, 2, 0);
    // bucket starts: the window starts, and a window's last bin when the window holds more than one capacity
    uint32_t sbits = 0u, nstart = 0u;
    if (mine) {
        uint32_t r = run0, wi = win0;
#pragma unroll
        for (int k = 0; k < kPerMax; ++k) {
            if ((uint32_t)k < per) {
                const uint32_t b = b0 + k;
                const bool ws = (wbits >> k) & 1u;
                if (ws) ++wi;                                   // bin b is in window wi - 1
                const uint32_t pend = r + c[k];                 // = P(b + 1)
                bool lastw;                                      // the next bin starts a window (or there is none)
                if ((uint32_t)k + 1u < per) lastw = (wbits >> (k + 1)) & 1u;
                else lastw = b + 1u >= G || wstart(b + 1u, pend, r);
                const uint32_t wf = s_first[wi - 1u];
                const bool alone = lastw && wf != b && pend - P(wf) > kCap;
                if (ws || alone) { sbits |= 1u << k; ++nstart; }
                r = pend;
            }
        }
    }
    uint32_t nseg;
    const uint32_t seg0 = block_excl_scan_n<NW>(nstart, s_w, &nseg);   // segments that start before bin b0 (the scan's barriers end the reads of the window starts)
    if (mine) {
        uint32_t si = seg0;
#pragma unroll
        for (int k = 0; k < kPerMax; ++k)
            if ((uint32_t)k < per && ((sbits >> k) & 1u)) s_first[si++] = (uint16_t)(b0 + k);
    }
    if (threadIdx.x == 0) s_first[nseg] = (uint16_t)G;   // G <= 2^14 fits
    __syncthreads();
    PCOP_STAMP(F, 0, 3, 0);
    // buckets = the segments that hold points, numbered in order
    uint32_t fbits = 0u, nfull = 0u;
    if (mine) {
        uint32_t si = seg0, r = run0;
#pragma unroll
        for (int k = 0; k < kPerMax; ++k) {
            if ((uint32_t)k < per) {
                if ((sbits >> k) & 1u) {
                    if (P(s_first[si + 1u]) != r) { fbits |= 1u << k; ++nfull; }
                    ++si;
                }
                r += c[k];
            }
        }
    }
    uint32_t nb;
    const uint32_t bk0 = block_excl_scan_n<NW>(nfull, s_w, &nb);
    if (threadIdx.x == 0) {
        st->nbuckets = nb;
        if (nb > kNbMax) atomicOr(&st->err, (uint32_t)kErrReplanPath);
        else if (nb > st->nb_cap) atomicOr(&st->err, (uint32_t)kErrReplanVcap);
    }
    if (nb > kNbMax) return;
    PCOP_STAMP(F, 0, 4, 0);
    if (mine) {
        uint32_t si = seg0, bk = bk0, r = run0;
#pragma unroll
        for (int k = 0; k < kPerMax; ++k) {
            if ((uint32_t)k < per) {
                const uint32_t b = b0 + k;
                if ((sbits >> k) & 1u) {
                    if ((fbits >> k) & 1u) {   // first record, records, first key, keys spanned
                        const uint32_t nx = s_first[si + 1u];
                        binfo[bk] = make_uint4(r, P(nx) - r, b << cs, (nx - b) << cs);
                        ++bk;
                    }
                    ++si;
                }
                bin2bucket[b] = (uint16_t)(bk ? bk - 1u : 0u);   // the last bucket started up to here (a bin outside any bucket holds no points)
                r += c[k];
            }
        }
    }
    PCOP_STAMP(F, 0, 15, 0);
}

// Points of every (tile, bucket): the tile's row of bin counts as a running sum in LDS, read at the bucket borders.
// A thread sums G / 256 consecutive bins in registers (16-byte loads), one block scan joins the threads.
// tcnt is bucket-major: [bucket][tile].
__global__ __launch_bounds__(kBlock) void k_bin_tilecnt(const Frame* frames) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    uint16_t* s_pre = reinterpret_cast<uint16_t*>(s_dyn);   // points of the row before bin i, i = 0 .. G
    const Frame& F = frames[blockIdx.y];
    const DevState* st = &F.st;
    const uint16_t* __restrict__ ctab = F.ctab;
    const uint4* __restrict__ binfo = F.binfo;
    uint16_t* __restrict__ tcnt = F.tcnt;
    if (stream_off(st)) return;
    const uint32_t tile = blockIdx.x;
    if (tile * st->tile >= st->n_in) return;
    const uint32_t G = 1u << st->gbits, cs = st->cshift, nb = st->nbuckets, ts = F.tstride;
    const uint16_t* __restrict__ row = ctab + (size_t)tile * G;
    constexpr uint32_t kVecMax = (1u << kBinBitsMax) / kBlock / 8u;   // 16-byte loads per thread: at most 8
    if (G >= 8u * kBlock) {
        const uint32_t nvec = G / kBlock / 8u;   // 1, 2, 4 or 8
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(row) + (size_t)threadIdx.x * nvec;
        uint4 v[kVecMax];
#pragma unroll
        for (uint32_t k = 0; k < kVecMax; ++k) v[k] = src[min(k, nvec - 1u)];
        uint32_t sum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < kVecMax; ++k) {
            if (k < nvec) {
                const uint32_t wv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += (wv[j] & 0xFFFFu) + (wv[j] >> 16);
            }
        }
        uint32_t tot;
        uint32_t run = block_excl_scan(sum, s_w, &tot);
        uint4* dst = reinterpret_cast<uint4*>(s_pre) + (size_t)threadIdx.x * nvec;
#pragma unroll
        for (uint32_t k = 0; k < kVecMax; ++k) {
            if (k < nvec) {
                const uint32_t wv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // a tile has 16384 points: the running sums fit 16 bits (mod 2^16 for the last one)
                    const uint32_t lo = run & 0xFFFFu; run += wv[j] & 0xFFFFu;
                    const uint32_t hi = run & 0xFFFFu; run += wv[j] >> 16;
                    o[j] = lo | (hi << 16);
                }
                dst[k] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        if (threadIdx.x == 0) s_pre[G] = (uint16_t)tot;
    } else {   // small grids: one wave, 64 bins at a time
        if (threadIdx.x < kWave) {
            uint32_t carry = 0u;
            for (uint32_t i0 = 0; i0 < G; i0 += kWave) {
                const uint32_t i = i0 + lane_id();
                const uint32_t v = i < G ? row[i] : 0u;
                const uint32_t incl = wave_incl_scan(v);
                if (i < G) s_pre[i] = (uint16_t)(carry + incl - v);
                carry += __shfl(incl, kWave - 1, kWave);
            }
            if (lane_id() == 0) s_pre[G] = (uint16_t)carry;
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += kBlock) {
        const uint4 info = binfo[b];
        const uint32_t first = info.z >> cs, end = first + ((info.w + (1u << cs) - 1u) >> cs);
        tcnt[(size_t)b * ts + tile] = (uint16_t)(s_pre[end] - s_pre[first]);
    }
}

// Peers of this lane's value among the active lanes of the wave, through two peer-mask words in LDS: word `a` of
// m_lo and word `b` of m_hi, where (a, b) determines the value (a = its low 6 bits, b = its high bits shifted by a, so
// that values with equal high bits still spread over the words of m_hi).  Lanes that share both words hold the same
// value.  All words are zero on entry and are left zero (every lane clears its own two, unconditionally, so nothing
// waits for the reads).  A wave's LDS operations execute in program order.
__device__ __forceinline__ uint64_t match_split_lds(unsigned long long* m_lo, unsigned long long* m_hi, uint32_t a, uint32_t b, bool active) {
    unsigned long long plo = 0ull, phi = 0ull;
    const unsigned long long bit = 1ull << lane_id();
    if (active) { atomicOr(m_lo + a, bit); atomicOr(m_hi + b, bit); }
    __builtin_amdgcn_wave_barrier();
    if (active) { plo = m_lo[a]; phi = m_hi[b]; }
    __builtin_amdgcn_wave_barrier();
    if (active) { m_lo[a] = 0ull; m_hi[b] = 0ull; }
    __builtin_amdgcn_wave_barrier();
    return plo & phi;
}

// Checks on the device, once per context, what the ORDERED variants rely on: lanes of one wave instruction that add to
// the same LDS word get their old values in lane order, and a following instruction continues from there.
// ok[0] counts violations; patterns: every lane picks one of `range` words, pseudo-randomly, some lanes sit out.
__global__ __launch_bounds__(kBlock) void k_lds_order_probe(uint32_t* __restrict__ violations) {
    __shared__ uint32_t cnt[kWavesPerBlock][2048];
    const uint32_t lane = lane_id(), w = threadIdx.x / kWave;
    uint32_t bad = 0u;
    for (uint32_t trial = 0; trial < 64u; ++trial) {
        for (uint32_t i = lane; i < 2048u; i += kWave) cnt[w][i] = 0u;
        __builtin_amdgcn_wave_barrier();
        const uint32_t ranges[8] = {1u, 2u, 4u, 16u, 64u, 256u, 2048u, 33u};
        const uint32_t range = ranges[trial & 7u];
        uint32_t h = (blockIdx.x * 64u + trial) * 0x9E3779B9u + (threadIdx.x + 1u) * 0x85EBCA6Bu;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        uint32_t a = h % range;
        if ((trial & 7u) == 7u) a = (a * 32u) & 2047u;   // one bank, different words
        const bool active = ((h >> 20) & 15u) != 0u;
        uint32_t old = 0u, old2 = 0u;
        if (active) old = atomicAdd(&cnt[w][a], 1u);
        if (active) old2 = atomicAdd(&cnt[w][a], 1u << 16);
        uint32_t expect = 0u, total = 0u;
        for (uint32_t l = 0; l < kWave; ++l) {
            const uint32_t al = __shfl(a, l, kWave);
            const bool actl = __shfl(active ? 1u : 0u, l, kWave) != 0u;
            if (actl && al == a) { ++total; if (l < lane) ++expect; }
        }
        if (active && (old != expect || old2 != (total | (expect << 16)))) ++bad;
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(violations, bad);
}

// Dynamic LDS of k_bin_scatter for nbc buckets and chunks of `chunk` points.
__host__ __device__ constexpr size_t scatter_lds_bytes(uint32_t nbc, uint32_t chunk = kScatChunk) {
    return (size_t)chunk * 12u + (size_t)chunk * 2u + (size_t)nbc * (4u + 16u) + 8u + 4u * (kScatThreads / kWave + 1u);
}

// ORDERED: the device hands out the old values of an LDS atomic add in lane order among the lanes of a wave
// instruction that hit the same word (k_lds_order_probe checked it when the context was made): the rank of a point
// inside (wave, bucket) is then the value one atomic returns.  Otherwise the ranks come from peer masks.
// IT: records a thread holds = a chunk of IT * 512 points.  8 (chunks of 4096, two blocks to a CU) or 16 (chunks of 8192: one block to a CU -
// half the barriers per point, runs twice as long in the stores, half the wave slots left to other contexts' kernels: measured 114 against
// 117 us for the pass and + 5 % on the bench line; needs a tile of a multiple of 8192 points and at most 2048 buckets: 155 KB of LDS).
template <uint32_t PB, bool ORDERED, uint32_t IT = kScatItems>   // PB: buckets per thread in the per-bucket steps: the launch provides for PB * 512 buckets
__global__ __launch_bounds__(kScatThreads) void k_bin_scatter(Frame* frames, uint32_t alt) {   // alt: write to rec_b (the order self-check's second opinion)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    constexpr uint32_t kScatItems = IT, kScatChunk = kScatThreads * IT;   // (shadow the defaults of pcop_device.hpp)
    constexpr uint32_t NW = kScatThreads / kWave;   // 8 waves; a wave ranks IT * 64 consecutive points of a chunk
    constexpr uint32_t kMatchWords = 128u;          // per wave: 64 words for the low 6 bits of the bucket, 64 for the high 6
    static_assert(NW == 8u && kScatChunk <= 65535u, "per bucket: eight 16-bit fields, one per wave");
    static_assert(NW * kMatchWords * 8u <= kScatChunk * 12u, "the peer masks fit the staging area");
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const uint16_t* __restrict__ bin2bucket = F.bin2bucket;
    const uint4* __restrict__ binfo = F.binfo;
    const uint16_t* __restrict__ tcnt = F.tcnt;
    Rec3* __restrict__ out = reinterpret_cast<Rec3*>(alt ? F.rec_b.get() : F.rec_a.get());
    if (stream_off(st)) return;
    const uint32_t n = st->n_in, nchunks = st->tile / kScatChunk;
    const uint32_t tile_base = blockIdx.x * st->tile;
    if (tile_base >= n) return;
    const uint32_t nbc = PB * kScatThreads, nb = st->nbuckets, G = 1u << st->gbits, cs = st->cshift;
    if (st->nb_cap > nbc) return;   // (the host launches an instance that fits)
    Rec3* stage = reinterpret_cast<Rec3*>(s_dyn);
    unsigned long long* s_match = reinterpret_cast<unsigned long long*>(s_dyn);   // shares the staging area (used in turns)
    uint16_t* sbkt = reinterpret_cast<uint16_t*>(s_dyn + (size_t)kScatChunk * 12u);
    uint4* wcnt = reinterpret_cast<uint4*>(s_dyn + (size_t)kScatChunk * 14u);     // [word][bucket]: the waves' counts (two 16-bit fields per word), then their starts in the staged chunk
    uint32_t* wcnt32 = reinterpret_cast<uint32_t*>(wcnt);
    uint32_t* cur = wcnt32 + (size_t)4u * nbc;   // per bucket: where the tile's next point goes (between chunks), minus its start in the staged chunk (inside one)
    uint32_t* s_w = cur + nbc;
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    const bool drop = st->is_dense == 0;
    VoxelKey key; key.init(st);
    const uint32_t w = threadIdx.x / kWave, lane = lane_id();
    const uint32_t wword = w >> 1, wshift = 16u * (w & 1u);
    // where this tile's points start in every bucket: the bucket's first record + its points in the lower tiles (a row of tcnt is
    // bucket-major: the lower tiles' 16-bit counts are one short contiguous run, eight to a 16-byte load - every block adds them up
    // for itself rather than pay a kernel for the scan)
    {
        const uint32_t ts = F.tstride, ngroups = (blockIdx.x + 7u) / 8u;   // ts: a multiple of 8
        for (uint32_t b = threadIdx.x; b < nbc; b += kScatThreads) {
            uint32_t sum = 0u;
            if (b < nb) {
                const uint4* __restrict__ row = reinterpret_cast<const uint4*>(tcnt + (size_t)b * ts);
                for (uint32_t g = 0; g < ngroups; ++g) {
                    const uint4 v = row[g];
                    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {   // tiles 8 g + 2 j and 8 g + 2 j + 1
                        sum += (8u * g + 2u * j < blockIdx.x) ? (wv[j] & 0xFFFFu) : 0u;
                        sum += (8u * g + 2u * j + 1u < blockIdx.x) ? (wv[j] >> 16) : 0u;
                    }
                }
                sum += binfo[b].x;
            }
            cur[b] = sum;
        }
    }
    constexpr uint32_t pb = PB;
    unsigned long long* m_lo = s_match + w * kMatchWords;
    unsigned long long* m_hi = m_lo + 64u;
    // The records of the next chunk are requested while the current one is ranked and staged.
    Rec3 rec[kScatItems], recn[kScatItems];
    auto request = [&](Rec3* r, uint32_t cb) {
#pragma unroll
        for (uint32_t j = 0; j < kScatItems; ++j) {
            const uint32_t i = min(cb + w * (kScatItems * kWave) + lane + j * kWave, n - 1u);   // clamped, unconditional
            r[j].x = x[i]; r[j].y = y[i]; r[j].z = z[i];
        }
    };
    request(recn, tile_base);
    PCOP_STAMP(F, 1, 0, 30);
    for (uint32_t ch = 0; ch < nchunks; ++ch) {
        const uint32_t cbase = tile_base + ch * kScatChunk;
        if (cbase >= n) break;
#pragma unroll
        for (uint32_t j = 0; j < kScatItems; ++j) rec[j] = recn[j];
        if (!ORDERED) for (uint32_t i = threadIdx.x; i < NW * kMatchWords; i += kScatThreads) s_match[i] = 0ull;
        for (uint32_t i = threadIdx.x; i < nbc; i += kScatThreads) wcnt[i] = make_uint4(0u, 0u, 0u, 0u);   // (as 16-byte stores: the layout is [4 words][bucket])
        const uint32_t wbase = cbase + w * (kScatItems * kWave) + lane;
        uint32_t bin[kScatItems], bkt[kScatItems], rank[kScatItems];
        bool act[kScatItems];
#pragma unroll
        for (uint32_t j = 0; j < kScatItems; ++j) {
            act[j] = wbase + j * kWave < n && (!drop || finite3(rec[j].x, rec[j].y, rec[j].z));
            const uint32_t bn = act[j] ? key(rec[j]) >> cs : 0u;
            bin[j] = min(bn, G - 1u);
        }
#pragma unroll
        for (uint32_t j = 0; j < kScatItems; ++j) bkt[j] = bin2bucket[bin[j]];   // all requested before any is used
        __builtin_amdgcn_sched_barrier(0);   // the next chunk's records are requested after the bucket lookups: loads return in order
        if (ch + 1u < nchunks) request(recn, cbase + kScatChunk);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (ch == 1u) PCOP_STAMP(F, 1, 1, 30);
        // ranks inside (wave, bucket): stable by construction (steps in order, lanes in order)
        if constexpr (ORDERED) {
#pragma unroll
            for (uint32_t j = 0; j < kScatItems; ++j) {
                uint32_t old = 0u;
                if (act[j]) old = atomicAdd(&wcnt32[wword * nbc + bkt[j]], 1u << wshift);
                rank[j] = (old >> wshift) & 0xFFFFu;
            }
        } else
#pragma unroll
        for (uint32_t j = 0; j < kScatItems; ++j) {
            const uint32_t d = bkt[j];
            const uint64_t peers = match_split_lds(m_lo, m_hi, d & 63u, ((d >> 6) + d) & 63u, act[j]);   // d < 4096
            const uint32_t lr = (uint32_t)__popcll(peers & lanemask_lt());
            uint32_t old = 0u;
            if (act[j] && lr == 0u) old = atomicAdd(&wcnt32[wword * nbc + d], (uint32_t)__popcll(peers) << wshift);
            const int leader = peers ? __ffsll((unsigned long long)peers) - 1 : 0;
            old = __shfl(old, leader, kWave);
            rank[j] = ((old >> wshift) & 0xFFFFu) + lr;
        }
        __syncthreads();
        if (ch == 1u) PCOP_STAMP(F, 1, 2, 30);
        // per bucket: points of the chunk; then every wave's start inside the staged chunk (bucket-major, waves in order)
        uint32_t tot[PB], lstart[PB];
        uint4 cw[PB];
        uint32_t tsum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < PB; ++k) {
            tot[k] = 0u; lstart[k] = 0u; cw[k] = make_uint4(0u, 0u, 0u, 0u);
            if (k < pb) {
                const uint32_t b = threadIdx.x * pb + k;
                cw[k] = make_uint4(wcnt32[b], wcnt32[nbc + b], wcnt32[2u * nbc + b], wcnt32[3u * nbc + b]);
                tot[k] = (cw[k].x & 0xFFFFu) + (cw[k].x >> 16) + (cw[k].y & 0xFFFFu) + (cw[k].y >> 16) +
                         (cw[k].z & 0xFFFFu) + (cw[k].z >> 16) + (cw[k].w & 0xFFFFu) + (cw[k].w >> 16);
            }
            tsum += tot[k];
        }
        uint32_t ctot;
        uint32_t lrun = block_excl_scan_n<NW>(tsum, s_w, &ctot);
#pragma unroll
        for (uint32_t k = 0; k < PB; ++k) {
            if (k < pb) {
                const uint32_t b = threadIdx.x * pb + k;
                const uint32_t wv[4] = {cw[k].x, cw[k].y, cw[k].z, cw[k].w};
                uint32_t o[4], r = lrun;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const uint32_t lo = r; r += wv[q] & 0xFFFFu; const uint32_t hi = r; r += wv[q] >> 16; o[q] = lo | (hi << 16); }
                wcnt32[b] = o[0]; wcnt32[nbc + b] = o[1]; wcnt32[2u * nbc + b] = o[2]; wcnt32[3u * nbc + b] = o[3];
                lstart[k] = lrun;
                cur[b] -= lrun;        // global position of a staged record = cur[bucket] + its place in the staged chunk
                lrun += tot[k];
            }
        }
        __syncthreads();
        if (ch == 1u) PCOP_STAMP(F, 1, 3, 30);
        // stage the chunk in output order (bucket-major, stable inside a bucket)
#pragma unroll
        for (uint32_t j = 0; j < kScatItems; ++j) {
            if (act[j]) {
                const uint32_t d = bkt[j];
                const uint32_t slot = ((wcnt32[wword * nbc + d] >> wshift) & 0xFFFFu) + rank[j];
                stage[slot] = rec[j];
                sbkt[slot] = (uint16_t)d;
            }
        }
        __syncthreads();
        if (ch == 1u) PCOP_STAMP(F, 1, 4, 30);
        for (uint32_t e = threadIdx.x; e < ctot; e += kScatThreads) {
            const Rec3 r = stage[e];
            out[cur[sbkt[e]] + e] = r;
        }
        __syncthreads();
        if (ch == 1u) PCOP_STAMP(F, 1, 5, 30);
#pragma unroll
        for (uint32_t k = 0; k < PB; ++k)   // back to "where the tile's next point goes"
            if (k < pb) cur[threadIdx.x * pb + k] += lstart[k] + tot[k];
    }
}

// Order self-check of the streaming path (first streamed batch of a context that ranks with lane-ordered LDS atomics): the
// same partition was written twice, to rec_a with ranks from the atomics and to rec_b with ranks from peer masks.  Both are
// stable, so the bytes agree; if they do not, the atomics did not hand out lane order on this part and the host falls back
// to the peer masks for good (kErrOrder).  force: the debug knob that makes the check fail once.
__global__ __launch_bounds__(kBlock) void k_rec_compare(Frame* frames, uint32_t force) {
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (stream_off(st)) return;
    const uint32_t* __restrict__ a = reinterpret_cast<const uint32_t*>(F.rec_a.get());
    const uint32_t* __restrict__ b = reinterpret_cast<const uint32_t*>(F.rec_b.get());
    const size_t words = (size_t)st->n_valid * 3u;
    bool bad = force != 0u && blockIdx.x == 0u && threadIdx.x == 0u;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock) bad |= a[i] != b[i];
    if (__ballot(bad) && lane_id() == 0u) atomicOr(&st->err, kErrOrder);
}

// A bucket with more than kBucketCap points (a single coarse bin, k_bin_plan) is cut into buckets that fit: counts of up
// to kHeavySub equal key sub-ranges, sub-ranges packed greedily into groups of at most kBucketCap points, the records
// partitioned by group - stably, so that index order survives - through the centroid planes (idle until k_voxel_gather)
// and written back over the bucket's records.  hsub[b][g] then describes group g the way binfo describes a bucket, and
// k_bucket_reduce runs them one after the other.  One block per bucket; blocks of ordinary buckets return at once.
__global__ __launch_bounds__(kBlock) void k_heavy_split(Frame* frames) {
    __shared__ uint32_t s_cnt[kHeavySub];
    __shared__ uint8_t s_grp[kHeavySub];
    __shared__ uint32_t s_goff[kHeavyGroups + 1], s_gsub[kHeavyGroups + 1], s_cur[kHeavyGroups];
    __shared__ uint32_t s_hist[kWavesPerBlock * kHeavyGroups];
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    __shared__ uint32_t s_ng;
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (stream_off(st)) return;
    const uint32_t b = blockIdx.x;
    if (b >= st->nbuckets) return;
    const uint4 info = F.binfo[b];
    if (info.y <= kBucketCap) return;
    Rec3* __restrict__ recs = reinterpret_cast<Rec3*>(F.rec_a.get());
    float* __restrict__ tx = F.cen_x; float* __restrict__ ty = F.cen_y; float* __restrict__ tz = F.cen_z;
    const uint32_t first = info.x, C = info.y, key_lo = info.z, span = info.w;
    uint32_t hs = 0u;
    while (((span + (1u << hs) - 1u) >> hs) > kHeavySub) ++hs;
    const uint32_t nsub = (span + (1u << hs) - 1u) >> hs;
    VoxelKey key; key.init(st);
    for (uint32_t i = threadIdx.x; i < kHeavySub; i += kBlock) s_cnt[i] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < C; i += kBlock) {
        const uint32_t lk = key(recs[first + i]) - key_lo;
        atomicAdd(&s_cnt[min(lk >> hs, nsub - 1u)], 1u);
    }
    __syncthreads();
    // Groups of consecutive sub-ranges with at most kBucketCap points each, by a window rule (no serial walk): a sub-range
    // belongs to window (points before it) / (kBucketCap / 2); the sub-ranges that start in one window hold less than
    // half a capacity plus the last one, and no sub-range may hold more than half a capacity itself (else: sort path), so
    // a window is a group.  Thread t owns the 8 consecutive sub-ranges from 8 t.
    {
        constexpr uint32_t kPer = kHeavySub / kBlock;   // 8
        constexpr uint32_t kHalf = kBucketCap / 2u;
        uint32_t c[kPer], csum = 0u, cmax = 0u;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { c[k] = s_cnt[threadIdx.x * kPer + k]; csum += c[k]; cmax = max(cmax, c[k]); }
        uint32_t total;
        const uint32_t p0 = block_excl_scan(csum, s_w, &total);
        __syncthreads();   // (s_cnt is read; it now takes the points before each sub-range)
        uint32_t r = p0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { s_cnt[threadIdx.x * kPer + k] = r; r += c[k]; }
        __syncthreads();
        // a group starts at sub-range i (with points, or i = 0) when its window differs from that of the sub-range before it
        uint32_t starts = 0u, nst = 0u;
        r = p0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t i = threadIdx.x * kPer + k;
            const bool st_here = i < nsub && (i == 0u || r / kHalf != s_cnt[i - 1u] / kHalf);
            if (st_here) { starts |= 1u << k; ++nst; }
            r += c[k];
        }
        uint32_t ngroups;
        uint32_t g = block_excl_scan(nst, s_w, &ngroups);
        const bool bad = __syncthreads_or((cmax > kHalf || ngroups > kHeavyGroups) ? 1 : 0) != 0;
        r = p0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t i = threadIdx.x * kPer + k;
            if ((starts >> k) & 1u) { if (!bad) { s_gsub[g] = i; s_goff[g] = r; } ++g; }
            if (i < nsub) s_grp[i] = (uint8_t)(g - 1u);
            r += c[k];
        }
        if (threadIdx.x == 0) {
            if (bad) atomicOr(&st->err, (uint32_t)kErrReplanPath);   // the sort path takes any cloud
            else { s_gsub[ngroups] = nsub; s_goff[ngroups] = C; }
            s_ng = bad ? 0u : ngroups;
        }
    }
    if (threadIdx.x < kHeavyGroups) s_cur[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t ng = s_ng;
    if (ng == 0u) { if (threadIdx.x == 0) F.hparts[b] = 0u; return; }
    const uint32_t w = threadIdx.x / kWave, lane = lane_id();
    uint32_t* hist_w = s_hist + w * kHeavyGroups;
    constexpr uint32_t kItemsH = 8u, kChunkH = kBlock * kItemsH;
    for (uint32_t cb = 0; cb < C; cb += kChunkH) {
        for (uint32_t i = threadIdx.x; i < kWavesPerBlock * kHeavyGroups; i += kBlock) s_hist[i] = 0u;
        __syncthreads();
        Rec3 rec[kItemsH];
        uint32_t grp[kItemsH], lrank[kItemsH];
        const uint32_t pbase = cb + w * (kItemsH * kWave) + lane;   // position in the bucket = index order
#pragma unroll
        for (uint32_t j = 0; j < kItemsH; ++j) rec[j] = recs[first + min(pbase + j * kWave, C - 1u)];
#pragma unroll
        for (uint32_t j = 0; j < kItemsH; ++j) {
            const bool active = pbase + j * kWave < C;
            const uint32_t lk = key(rec[j]) - key_lo;
            const uint32_t g = active ? (uint32_t)s_grp[min(lk >> hs, nsub - 1u)] : 0u;
            const uint64_t peers = match_digit(g, 7u, active);   // kHeavyGroups <= 128
            const uint32_t rank = (uint32_t)__popcll(peers & lanemask_lt());
            uint32_t old = 0u;
            if (active) old = hist_w[g];
            if (active && rank == 0u) hist_w[g] = old + (uint32_t)__popcll(peers);
            grp[j] = g; lrank[j] = old + rank;
        }
        __syncthreads();
        if (threadIdx.x < kHeavyGroups) {   // per group: where each wave's records go, the group's cursor moved past the chunk
            uint32_t run = s_cur[threadIdx.x];
#pragma unroll
            for (uint32_t q = 0; q < (uint32_t)kWavesPerBlock; ++q) { const uint32_t t = s_hist[q * kHeavyGroups + threadIdx.x]; s_hist[q * kHeavyGroups + threadIdx.x] = run; run += t; }
            s_cur[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < kItemsH; ++j) {
            if (pbase + j * kWave < C) {
                const uint32_t dest = first + s_goff[grp[j]] + hist_w[grp[j]] + lrank[j];
                tx[dest] = rec[j].x; ty[dest] = rec[j].y; tz[dest] = rec[j].z;
            }
        }
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < C; i += kBlock) {
        Rec3 r; r.x = tx[first + i]; r.y = ty[first + i]; r.z = tz[first + i];
        recs[first + i] = r;
    }
    if (threadIdx.x < ng) {
        const uint32_t g = threadIdx.x;
        const uint32_t k0 = s_gsub[g] << hs, k1 = min(span, s_gsub[g + 1u] << hs);
        F.hsub[(size_t)b * kHeavyGroups + g] = make_uint4(first + s_goff[g], s_goff[g + 1u] - s_goff[g], key_lo + k0, k1 - k0);
    }
    if (threadIdx.x == 0) F.hparts[b] = ng;
}

// A bucket lives in LDS, a block of four waves works on it (two consecutive buckets per block, the second one's records requested
// while the first is worked on).  The centroids stay at the bucket's place in the scratch records, the bucket leaves its voxel count in
// bucket_pub; k_voxel_gather (the next launch) adds the counts up and moves the centroids to their ranks.
// Round 4 built two other forms of this kernel, bit-exact both, and measured them slower: ONE WAVE per bucket (no
// barriers, 28 KB of LDS: 176 us a batch against 105 + 35 for this kernel and the gather), and this kernel FUSED with the gather
// (the bucket publishes its voxel count, looks back over the lower buckets' counts and writes its centroids straight to their ranks:
// 144-153 us, and - 6 % on the bench line in the same run): a device-scope load of another block's word is served past the XCD's
// L2, a fabric round trip of a microsecond or two that sits in every block's critical path (a block of this kernel takes 8.4 k cycles
// by its stamps, the fused one 19.9 k), while the gather's plain loads after a kernel boundary hit the L2.
template <bool ORDERED>   // ORDERED: as in k_bin_scatter
__global__ __launch_bounds__(kBlock) void k_bucket_reduce(Frame* frames) {
    constexpr uint32_t NW = kBlock / kWave;                    // 4 waves; a wave ranks 512 consecutive points of the bucket
    constexpr uint32_t kItemsB = kBucketCap / kBlock;          // 8
    constexpr uint32_t kBucketsPerBlock = kReduceBucketsPerBlock;
    constexpr uint32_t kMapWords = kBucketCells / 32u;         // 4096
    constexpr uint32_t kPerW = kMapWords / kBlock;             // at most 16 occupancy words per thread
    constexpr uint32_t kPlane = kBucketCap + kBucketCap / 32u + 8u;   // a staged coordinate plane: slot s lives at s + (s >> 5)
    static_assert(kHeavyGroups <= 128u, "k_heavy_split matches groups on 7 bits");
    static_assert(kPlane * 12u >= kMapWords * 4u + kMapWords * 2u, "occupancy words + their running counts fit the staging area");
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[kPlane * 12u];   // occupancy bits + running counts, then the staged records (three planes)
    __shared__ __attribute__((aligned(16))) uint32_t s_wc[2u * kBucketCap];   // per voxel id: counts of waves 0,1 (low / high half) and of waves 2,3; then the waves' places
    __shared__ __attribute__((aligned(16))) uint16_t s_vs[kBucketCap + 8u];   // per occupied voxel id, in order: where its run starts in the staged bucket; then the end of the last run
    __shared__ unsigned long long s_match[ORDERED ? 1u : NW * 96u];   // per wave: 64 + 32 peer-mask words (voxel ids below 2048)
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    const uint4* __restrict__ binfo = F.binfo;
    const Rec3* __restrict__ recs = reinterpret_cast<const Rec3*>(F.rec_a.get());
    uint32_t* __restrict__ bucket_pub = F.bucket_pub;
    float4* __restrict__ scratch = F.rec_b;
    if (stream_off(st)) return;
    const uint32_t nb = st->nbuckets;
    const uint32_t blk = blockIdx.x;
    if (blk * kBucketsPerBlock >= nb) return;
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(s_mem);
    uint16_t* wpre = reinterpret_cast<uint16_t*>(s_mem + kMapWords * 4u);
#if PCOP_REDUCE_AOS
    Rec3* stage_r = reinterpret_cast<Rec3*>(s_mem);   // tuning build: the staged records as 12-byte records (a stride of three words walks every bank) instead of three planes
#else
    float* stage_x = reinterpret_cast<float*>(s_mem); float* stage_y = stage_x + kPlane; float* stage_z = stage_y + kPlane;
#endif
    const uint32_t w = threadIdx.x / kWave, lane = lane_id();
    VoxelKey key; key.init(st);
    const uint32_t pbase = w * (kItemsB * kWave) + lane;   // position in the bucket = index order
    // A block takes kBucketsPerBlock consecutive buckets, one after the other; the records of the next one are requested
    // before the current one is worked on, so only the first bucket's load latency is exposed.
    Rec3 rec[kItemsB], recn[kItemsB];
    uint4 info_n = binfo[blk * kBucketsPerBlock];
    auto request = [&](Rec3* r, const uint4& inf) {
        const uint32_t lastr = inf.y ? min(inf.y, kBucketCap) - 1u : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kItemsB; ++j) r[j] = recs[inf.x + min(pbase + j * kWave, lastr)];   // clamped, unconditional
    };
    request(recn, info_n);
    for (uint32_t it = 0; it < kBucketsPerBlock; ++it) {
    const uint32_t b = blk * kBucketsPerBlock + it;
    if (b >= nb) break;
    const uint4 info_b = info_n;   // {first record, records, first key, keys spanned}
#pragma unroll
    for (uint32_t j = 0; j < kItemsB; ++j) rec[j] = recn[j];
    if (it + 1u < kBucketsPerBlock && b + 1u < nb) {
        info_n = binfo[b + 1u];
        request(recn, info_n);
    }
    // a heavy bucket comes as the buckets k_heavy_split cut it into, taken one after the other; their voxels follow each
    // other in the bucket's part of the scratch records
    const bool heavy = info_b.y > kBucketCap;
    const uint32_t nparts = heavy ? min(F.hparts[b], kHeavyGroups) : 1u;
    uint32_t nv_acc = 0u;
    bool failed = heavy && nparts == 0u;
    for (uint32_t part = 0; part < nparts && !failed; ++part) {
    uint4 info = info_b;
    if (heavy) {
        info = F.hsub[(size_t)b * kHeavyGroups + part];
        if (info.y == 0u) continue;   // (a group of empty key sub-ranges at the end)
        request(rec, info);
    }
    const uint32_t cnt = info.y, key_lo = info.z, span = info.w;
    // A bucket that spans at most kBucketCap cells sorts by the cell itself (ids of empty cells get empty runs);
    // a wider one first ranks its occupied cells through an occupancy map, so that ids stay below the point count.
    const bool direct = span <= kBucketCap;
    const uint32_t words = (span + 31u) / 32u;
    bool bad = cnt == 0u || cnt > kBucketCap || span > kBucketCells;   // never: k_bin_plan keeps both limits
    uint32_t lk[kItemsB], id[kItemsB], rank[kItemsB];
    bool act[kItemsB];
    PCOP_STAMP(F, 2, 0, 150);
    // (the LDS is cleared while the records are on their way)
    if (!direct) for (uint32_t i = threadIdx.x; i < min(words, kMapWords); i += kBlock) bitmap[i] = 0u;
    // ids stay below nidr (the cells spanned, or the points, rounded up to a multiple of the block): the per-id steps stop there
    const uint32_t nidr = min(kBucketCap, ((direct ? span : cnt) + kBlock - 1u) / kBlock * kBlock);
    for (uint32_t v = threadIdx.x; v < nidr; v += kBlock) { s_wc[v] = 0u; s_wc[kBucketCap + v] = 0u; }
    if (!ORDERED) for (uint32_t i = threadIdx.x; i < NW * 96u; i += kBlock) s_match[i] = 0ull;
#pragma unroll
    for (uint32_t j = 0; j < kItemsB; ++j) {
        lk[j] = key(rec[j]) - key_lo;
        act[j] = !bad && pbase + j * kWave < cnt;
        if (act[j] && lk[j] >= span) { bad = true; act[j] = false; }
        id[j] = act[j] ? lk[j] : 0u;
    }
    bad = __syncthreads_or(bad ? 1 : 0) != 0;
    PCOP_STAMP(F, 2, 1, 150);
    if (bad) { failed = true; break; }
    if (!direct) {
#pragma unroll
        for (uint32_t j = 0; j < kItemsB; ++j)
            if (act[j]) atomicOr(&bitmap[lk[j] >> 5], 1u << (lk[j] & 31u));
        __syncthreads();
    PCOP_STAMP(F, 2, 2, 150);
        // running count of the occupied cells: thread t owns the perw consecutive words from t * perw
        const uint32_t perw = (words + kBlock - 1u) / kBlock;   // 1 .. 16
        uint32_t wv[kPerW];
        uint32_t c = 0u;
#pragma unroll
        for (uint32_t k = 0; k < kPerW; ++k) {
            const uint32_t wi = threadIdx.x * perw + k;
            wv[k] = (k < perw && wi < words) ? bitmap[wi] : 0u;
            c += __popc(wv[k]);
        }
        uint32_t nocc;
        uint32_t run = block_excl_scan(c, s_w, &nocc);
#pragma unroll
        for (uint32_t k = 0; k < kPerW; ++k) {
            const uint32_t wi = threadIdx.x * perw + k;
            if (k < perw && wi < words) wpre[wi] = (uint16_t)run;
            run += __popc(wv[k]);
        }
        __syncthreads();
    PCOP_STAMP(F, 2, 3, 150);
#pragma unroll
        for (uint32_t j = 0; j < kItemsB; ++j) {   // id = occupied cells of the bucket before this one
            const uint32_t wi = act[j] ? lk[j] >> 5 : 0u;
            id[j] = (uint32_t)wpre[wi] + __popc(bitmap[wi] & ((1u << (lk[j] & 31u)) - 1u));
        }
        __syncthreads();   // the occupancy words are done with: the staging area may be written from here on
    PCOP_STAMP(F, 2, 4, 150);
    }
    // ranks inside (wave, id)
    const uint32_t wword = w >> 1, wshift = 16u * (w & 1u);
    if constexpr (ORDERED) {
#pragma unroll
        for (uint32_t j = 0; j < kItemsB; ++j) {
            uint32_t old = 0u;
            if (act[j]) old = atomicAdd(&s_wc[wword * kBucketCap + id[j]], 1u << wshift);
            rank[j] = (old >> wshift) & 0xFFFFu;
        }
    } else {
    unsigned long long* m_lo = s_match + w * 96u;
    unsigned long long* m_hi = m_lo + 64u;
#pragma unroll
    for (uint32_t j = 0; j < kItemsB; ++j) {
        const uint32_t d = id[j];
        const uint64_t peers = match_split_lds(m_lo, m_hi, d & 63u, ((d >> 6) + d) & 31u, act[j]);   // d < 2048
        const uint32_t lr = (uint32_t)__popcll(peers & lanemask_lt());
        uint32_t old = 0u;
        if (act[j] && lr == 0u) old = atomicAdd(&s_wc[wword * kBucketCap + d], (uint32_t)__popcll(peers) << wshift);
        const int leader = peers ? __ffsll((unsigned long long)peers) - 1 : 0;
        old = __shfl(old, leader, kWave);
        rank[j] = ((old >> wshift) & 0xFFFFu) + lr;
    }
    }
    __syncthreads();
    PCOP_STAMP(F, 2, 5, 150);
    // Thread t owns ids 8 t .. 8 t + 7 (two 16-byte reads per count plane): points per id, then - after one scan over the
    // block that carries the points in its low half and the occupied ids in its high half - every wave's place for every
    // id (written back over the counts: the staging step then needs one word per record), and the run starts of the
    // OCCUPIED ids in order (s_vs), so that the sums below give every lane a voxel.
    uint32_t nvox = 0u;
    {
        const bool mine = threadIdx.x * kItemsB < nidr;
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
        const uint4* wc4 = reinterpret_cast<const uint4*>(s_wc);
        const uint4 a0 = mine ? wc4[threadIdx.x * 2u] : z4, a1 = mine ? wc4[threadIdx.x * 2u + 1u] : z4;
        const uint4 q0 = mine ? wc4[kBucketCap / 4u + threadIdx.x * 2u] : z4, q1 = mine ? wc4[kBucketCap / 4u + threadIdx.x * 2u + 1u] : z4;
        // cw[2 q] = the word of waves 0, 1 for id 8 t + q, cw[2 q + 1] = that of waves 2, 3
        uint32_t cw[16] = {a0.x, q0.x, a0.y, q0.y, a0.z, q0.z, a0.w, q0.w, a1.x, q1.x, a1.y, q1.y, a1.z, q1.z, a1.w, q1.w};
        uint32_t tsum = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t c = (cw[2 * q] & 0xFFFFu) + (cw[2 * q] >> 16) + (cw[2 * q + 1] & 0xFFFFu) + (cw[2 * q + 1] >> 16);
            tsum += c + (c ? 0x10000u : 0u);
        }
        uint32_t ttot;
        uint32_t srun = block_excl_scan(tsum, s_w, &ttot);
        nvox = ttot >> 16;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t c0 = cw[2 * q] & 0xFFFFu, c1 = cw[2 * q] >> 16, c2 = cw[2 * q + 1] & 0xFFFFu, c3 = cw[2 * q + 1] >> 16;
            const uint32_t c = c0 + c1 + c2 + c3;
            const uint32_t r = srun & 0xFFFFu;
            if (c) s_vs[srun >> 16] = (uint16_t)r;
            cw[2 * q] = r | ((r + c0) << 16);
            cw[2 * q + 1] = (r + c0 + c1) | ((r + c0 + c1 + c2) << 16);
            srun += c + (c ? 0x10000u : 0u);
        }
        if (mine) {
            uint4* wo4 = reinterpret_cast<uint4*>(s_wc);
            wo4[threadIdx.x * 2u] = make_uint4(cw[0], cw[2], cw[4], cw[6]); wo4[threadIdx.x * 2u + 1u] = make_uint4(cw[8], cw[10], cw[12], cw[14]);
            wo4[kBucketCap / 4u + threadIdx.x * 2u] = make_uint4(cw[1], cw[3], cw[5], cw[7]); wo4[kBucketCap / 4u + threadIdx.x * 2u + 1u] = make_uint4(cw[9], cw[11], cw[13], cw[15]);
        }
        if (threadIdx.x == 0) s_vs[nvox] = (uint16_t)(ttot & 0xFFFFu);
    }
    __syncthreads();
    PCOP_STAMP(F, 2, 7, 150);
#pragma unroll
    for (uint32_t j = 0; j < kItemsB; ++j) {
        if (act[j]) {
            const uint32_t slot = ((s_wc[wword * kBucketCap + id[j]] >> wshift) & 0xFFFFu) + rank[j];
#if PCOP_REDUCE_AOS
            stage_r[slot] = rec[j];
#else
            const uint32_t ps = slot + (slot >> 5);
            stage_x[ps] = rec[j].x; stage_y[ps] = rec[j].y; stage_z[ps] = rec[j].z;
#endif
        }
    }
    __syncthreads();
    PCOP_STAMP(F, 2, 8, 150);
    // one lane per VOXEL: the run in order, from 0.0f (voxel_grid.hpp "Fourth pass"), divided by float(count).  The centroid
    // of the bucket's v-th voxel goes to place (first record of the bucket + v) of the scratch records: a bucket has at
    // least as many records as voxels, so the places of different buckets do not meet; k_voxel_gather moves them to their ranks.
    for (uint32_t v = threadIdx.x; v < nvox; v += kBlock) {
        const uint32_t s0 = s_vs[v], e0 = s_vs[v + 1u];
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (uint32_t q = s0; q < e0; q += 4u) {   // four records requested at a time (clamped), added in order
            float vx[4], vy[4], vz[4];
#pragma unroll
#if PCOP_REDUCE_AOS
            for (uint32_t u = 0; u < 4u; ++u) { const Rec3 t = stage_r[min(q + u, e0 - 1u)]; vx[u] = t.x; vy[u] = t.y; vz[u] = t.z; }
#else
            for (uint32_t u = 0; u < 4u; ++u) { const uint32_t qq = min(q + u, e0 - 1u), ps = qq + (qq >> 5); vx[u] = stage_x[ps]; vy[u] = stage_y[ps]; vz[u] = stage_z[ps]; }
#endif
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) if (q + u < e0) { sx += vx[u]; sy += vy[u]; sz += vz[u]; }
        }
        const float fc = (float)(e0 - s0);
        scratch[info_b.x + nv_acc + v] = make_float4(sx / fc, sy / fc, sz / fc, 0.0f);
    }
    nv_acc += nvox;
    PCOP_STAMP(F, 2, 20, 150);
    __syncthreads();   // the next bucket clears and fills the same LDS
    }
    if (threadIdx.x == 0) {   // k_voxel_gather adds the counts up to output ranks
        if (failed) { if (!heavy || !st->err) atomicOr(&st->err, (uint32_t)kErrInternal); bucket_pub[b] = 0u; }
        else bucket_pub[b] = nv_acc;
    }
    }
}


// ------------------------------------------------------------------------------------------
// Occupancy scan of the dense cell grid (pcop_cells_grid.hpp): one byte per cell -> bits + running popcount.
// ------------------------------------------------------------------------------------------
// 16 flag bytes (each 0 or 1) -> 16 bits
__device__ __forceinline__ uint32_t flags_to_bits(const uint4& v) {
    const uint32_t a = ((v.x & 0x01010101u) * 0x01020408u) >> 24, b = ((v.y & 0x01010101u) * 0x01020408u) >> 24;
    const uint32_t c = ((v.z & 0x01010101u) * 0x01020408u) >> 24, d = ((v.w & 0x01010101u) * 0x01020408u) >> 24;
    return (a & 15u) | ((b & 15u) << 4) | ((c & 15u) << 8) | ((d & 15u) << 12);
}

// One block per kVmapTile occupancy words: bytes -> bits (and the bytes are zeroed again for the
// next frame), popcount scan; the tile publishes its count and adds up the counts of the lower
// tiles (same publish / bounded-wait scheme as k_voxel_reduce).  The last tile knows the total: n_cells.
__global__ __launch_bounds__(kBlock) void k_occ_scan(Frame* frames) {
    __shared__ uint32_t s_w[kWavesPerBlock + 1];
    __shared__ uint16_t s_half[2 * kVmapTile];
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    uint4* __restrict__ flags = reinterpret_cast<uint4*>(F.cflag.get());
    uint2* __restrict__ omap = F.cmap.get();
    uint32_t* __restrict__ otile = F.ctile.get();
    // Runs whenever the marking kernel ran (the replan bits this kernel raises itself must not stop its
    // other tiles: every tile has flag bytes to turn back to zero).
    const uint32_t own = (uint32_t)(kErrReplanCpath | kErrReplanCtiles | kErrInternal | kErrNonFinite | kErrStale);
    if ((st->err & ~own) || st->trivial || st->kind == kFrameVoxelOnly || !st->cpath) return;
    const uint32_t words = st->cmap_words;
    if (blockIdx.x == 0 && threadIdx.x == 0 && words > gridDim.x * kVmapTile)   // the host sized this grid from the previous frame
        atomicOr(&st->err, (uint32_t)kErrReplanCtiles);
    const uint32_t base = blockIdx.x * kVmapTile;
    if (base >= words) return;
    const uint32_t cnt = min(kVmapTile, words - base);
    // a word = 32 cells = two 16-byte loads; the flag array is padded to whole tiles
    constexpr uint32_t kHalves = 2 * kVmapTile / kBlock;
    uint4 fl[kHalves];
#pragma unroll
    for (uint32_t k = 0; k < kHalves; ++k)   // all loads first (clamped, not tested: no load sits under a branch)
        fl[k] = flags[(size_t)base * 2 + min(k * kBlock + threadIdx.x, 2 * cnt - 1u)];
#pragma unroll
    for (uint32_t k = 0; k < kHalves; ++k) {
        const uint32_t h = k * kBlock + threadIdx.x;   // half-word of the tile
        uint32_t bits = 0;
        if (h < 2 * cnt) {
            bits = flags_to_bits(fl[k]);
            if (bits) flags[(size_t)base * 2 + h] = make_uint4(0u, 0u, 0u, 0u);
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
            if (++spins > st->spin_limit) { atomicOr(&st->err, (uint32_t)kErrInternal); v = 1u; break; }
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
        st->n_cells = before + tot;
        // every point flags one cell: more cells than points means flag bytes that were set before the frame began (cell ids would
        // leave the per-cell arrays): the frame is given up before anything is indexed by them (kErrStale: scrub and run again)
        if (before + tot > st->c_n) atomicOr(&st->err, (uint32_t)kErrStale);
    }
}

}  // namespace pcop

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
    // all occupancy words of the thread's records are requested before any is used (a dropped point reads word 0;
    // a load under a branch would be waited for at the branch's join, one L2 round trip per record)
    uint2 e[kItems];
#pragma unroll
    for (int j = 0; j < kItems; ++j) e[j] = vmap[k[j] != kNoKey ? k[j] >> 5 : 0u];
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t i = wbase + j * kWave;
        if (i < n) {
            uint32_t d = nb - 1u;
            if (k[j] != kNoKey) {
                const uint32_t id = e[j].y + __popc(e[j].x & ((1u << (k[j] & 31u)) - 1u));
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

}  // namespace pcop

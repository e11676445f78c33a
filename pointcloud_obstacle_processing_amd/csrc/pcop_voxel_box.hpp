// pcop_voxel_box.hpp - voxel keys formed against the context's box: no bounds pass over the cloud.
// Part of pcop_device.hpp (included from there, after pcop_voxel.hpp; not a stand-alone header).
//
// VoxelGrid needs the cloud's bounds before it can form a key, which costs a pass over the cloud that computes six numbers.  But
// PCL's linear index orders voxels lexicographically by (iz, iy, ix), and floor(p * inv) does not depend on the origin (the float
// subtraction of integer-valued floats is exact): any box that holds the cloud gives the same voxels in the same order.  So the
// HOST keeps a box per context (the union of the voxel grids of the frames it has seen since the leaf size was set, with a margin)
// and the pass that reads the cloud first anyway takes the true bounds on its way - as minima / maxima of the voxel coordinates -
// and checks every point against the box; a point outside raises kErrReplanBox and the host re-runs the batch with the bounds pass.
//   k_box_setup        mailbox -> DevState with the host's box as the grid; dense cell grid set-up folded in
//   bin_hist_on_box    k_bin_hist's body on the box (streaming path)
//   radix_hist_on_box  k_radix_hist's pass 0 on the box (sort path)
//   box_true_grid      PCL's own min_b / div_b into DevState once nothing forms keys any more (k_emit's first block)
// (Round 3 also had a third voxel path here that sorted tile-locally and planned from a sample: built, bit-exact, measured 22 %
//  slower than the streaming path on C3 and removed in round 4; DESIGN.md keeps what it taught.)
#pragma once

namespace pcop {

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
// mode 2: the streaming path (k_bin_hist takes the bounds on its way); 3: the sort path (k_radix_hist of pass 0 does).
__global__ __launch_bounds__(kBlock) void k_box_setup(Frame* frames, uint32_t npass, uint32_t maxbits, uint32_t fold_grid, uint32_t mode) {
    Frame& F = frames[blockIdx.y];
    DevState* st = &F.st;
    if (threadIdx.x == 0) {
#if PCOP_DIAG_REUSE_PLAN
        // diagnostic build: what the skipped histogram pass and k_bin_plan would have left in DevState (the last batch's values: same cloud, same box, same bins)
        const uint32_t keep_nb = st->nbuckets, keep_valid = st->n_valid, keep_maxbin = st->maxbin, keep_nonfinite = st->box_nonfinite;
#endif
        frame_init(F);
        const FrameArgs a = *F.args;
#if PCOP_DIAG_REUSE_PLAN
        if (a.pad1 == 1u && mode == 2u) { st->nbuckets = keep_nb; st->n_valid = keep_valid; st->maxbin = keep_maxbin; st->box_nonfinite = keep_nonfinite; }
#endif
        st->sbox = 1u; st->vpath = mode == 3u ? 0u : 1u;
        st->box_tile = mode == 2u ? st->tile : (uint32_t)kTile;
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
            if ((kBucketCells >> st->cshift) == 0u) atomicOr(&st->err, (uint32_t)kErrReplanBins);   // a coarse bin spans more voxels than a bucket may
        }
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
        if (bad) { if (st->is_dense == 0) atomicAdd(&st->box_nonfinite, bad); else atomicOr(&st->err, (uint32_t)kErrNonFinite); }
        if (ok && (mx[0] >= key.d0 || mx[1] >= key.d1 || mx[2] >= key.d2)) atomicOr(&st->err, (uint32_t)kErrReplanBox);   // a point outside the box
    }
}

__device__ __forceinline__ void bin_hist_on_box(Frame& F, uint32_t* s_hist, uint32_t tile_base, uint32_t tile, uint32_t n, uint32_t G, uint32_t cs) {
    constexpr uint32_t NW = kHistThreads / kWave;
    __shared__ uint32_t s_red[NW * 8u];
    DevState* st = &F.st;
    uint16_t* __restrict__ ctab = F.ctab;
    const float* __restrict__ x = st->src_x; const float* __restrict__ y = st->src_y; const float* __restrict__ z = st->src_z;
    BoxKey key; key.init(st);
    uint32_t mn0 = 0xFFFFFFFFu, mn1 = 0xFFFFFFFFu, mn2 = 0xFFFFFFFFu, mx0 = 0u, mx1 = 0u, mx2 = 0u;
    uint32_t nbad = 0u, nok = 0u;
    constexpr int kRound = kBinTileSmall / kHistThreads;   // records in flight per thread: 8
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
        if (bad) { if (st->is_dense == 0) atomicAdd(&st->box_nonfinite, bad); else atomicOr(&st->err, (uint32_t)kErrNonFinite); }
        if (ok && (mx[0] >= key.d0 || mx[1] >= key.d1 || mx[2] >= key.d2)) atomicOr(&st->err, (uint32_t)kErrReplanBox);   // a point outside the box
    }
    uint16_t* __restrict__ row = ctab + (size_t)blockIdx.x * G;
    for (uint32_t b = threadIdx.x; b < G; b += kHistThreads) row[b] = (uint16_t)s_hist[b];   // a tile has at most 2^15 points (kBinTileLarge): fits
}

// PCL's own grid into DevState, once nothing forms keys any more (k_emit's first block, before it hands DevState to the host):
// the true bounds from the partials of the tiles of DevState::box_tile points, min_b / div_b / divb_mul as
// VoxelGrid::applyFilter derives them (voxel_setup does the same where a bounds pass comes first).  The whole block calls.
__device__ __forceinline__ void box_true_grid(Frame& F, MinMaxPartial* s_p) {
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

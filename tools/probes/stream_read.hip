// What a pass over a batch of SoA clouds can reach on gfx950: 16 frames x 3 planes x 1 M floats (192 MB), read once, by kernels
// shaped like k_bin_hist (tile of 32768 points per block of 1024 threads) - plain sum vs voxel keys + an LDS histogram + bounds,
// 4-byte vs 16-byte loads, 512 vs 1024 threads.  Stand-alone probe: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o stream_read stream_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <math.h>

__device__ uint32_t mul1_rt = 240u, mul2_rt = 57600u;   // (run-time values, as in the library: the compiler must not turn them into shifts)
constexpr uint32_t kN = 1000000, kFrames = 16, kTile = 32768, kBins = 16384;
struct Planes { const float* x[kFrames]; const float* y[kFrames]; const float* z[kFrames]; uint16_t* tab[kFrames]; uint32_t* part[kFrames]; };

template <bool M24>
__device__ __forceinline__ uint32_t vkey(float x, float y, float z, uint32_t* i0, uint32_t* i1, uint32_t* i2) {
    const float inv = 50.0f, big = 2147483520.0f;
    *i0 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(x * inv) - (-8.0f), -1.0f, big);
    *i1 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(y * inv) - (-8.0f), -1.0f, big);
    *i2 = (uint32_t)(int)__builtin_amdgcn_fmed3f(floorf(z * inv) - (-8.0f), -1.0f, big);
    if (M24) return *i0 + __umul24(*i1, mul1_rt) + __umul24(*i2, mul2_rt);
    return *i0 + *i1 * mul1_rt + *i2 * mul2_rt;
}

// MODE 0: sum only; 1: keys + LDS histogram + bounds + table write (what k_bin_hist does); 2: keys + bounds, no LDS atomics;
// 3: LDS atomics on a cheap key (the bits of x), no bounds; 4: as 1 with 24-bit multiplies in the key; 5: as 1 without the bounds
template <int THREADS, int VEC, int MODE, int ROUND>
__global__ __launch_bounds__(THREADS) void k_read(Planes P, float* sink) {
    extern __shared__ uint32_t s_hist[];
    const uint32_t f = blockIdx.y, tile_base = blockIdx.x * kTile;
    if (tile_base >= kN) return;
    const float* __restrict__ x = P.x[f]; const float* __restrict__ y = P.y[f]; const float* __restrict__ z = P.z[f];
    if (MODE != 0) { for (uint32_t b = threadIdx.x; b < kBins; b += THREADS) s_hist[b] = 0u; __syncthreads(); }
    float acc = 0.0f;
    uint32_t mn0 = ~0u, mn1 = ~0u, mn2 = ~0u, mx0 = 0, mx1 = 0, mx2 = 0;
    auto take = [&](float a, float b, float c, bool ok) {
        if (MODE == 0) { acc += a + b + c; }
        else if (MODE == 3) { if (ok) atomicAdd(&s_hist[(__float_as_uint(a) >> 3) & (kBins - 1u)], 1u); acc += b + c; }
        else if (ok) {
            uint32_t i0, i1, i2;
            const uint32_t k = vkey<MODE == 4>(a, b, c, &i0, &i1, &i2);
            if (MODE != 5) { mn0 = min(mn0, i0); mx0 = max(mx0, i0); mn1 = min(mn1, i1); mx1 = max(mx1, i1); mn2 = min(mn2, i2); mx2 = max(mx2, i2); }
            if (MODE == 2) mx0 ^= k; else atomicAdd(&s_hist[min(k >> 7, kBins - 1u)], 1u);
        }
    };
    constexpr uint32_t per = kTile / THREADS / VEC;   // load steps per thread
    for (uint32_t r0 = 0; r0 < per; r0 += ROUND) {
        if (VEC == 1) {
            float a[ROUND], b[ROUND], c[ROUND];
#pragma unroll
            for (int k = 0; k < ROUND; ++k) { const uint32_t i = min(tile_base + (r0 + k) * THREADS + threadIdx.x, kN - 1u); a[k] = x[i]; b[k] = y[i]; c[k] = z[i]; }
#pragma unroll
            for (int k = 0; k < ROUND; ++k) take(a[k], b[k], c[k], tile_base + (r0 + k) * THREADS + threadIdx.x < kN);
        } else {
            float4 a[ROUND], b[ROUND], c[ROUND];
#pragma unroll
            for (int k = 0; k < ROUND; ++k) {
                const uint32_t i = min(tile_base / 4u + (r0 + k) * THREADS + threadIdx.x, kN / 4u - 1u);
                a[k] = reinterpret_cast<const float4*>(x)[i]; b[k] = reinterpret_cast<const float4*>(y)[i]; c[k] = reinterpret_cast<const float4*>(z)[i];
            }
#pragma unroll
            for (int k = 0; k < ROUND; ++k) {
                const bool ok = tile_base / 4u + (r0 + k) * THREADS + threadIdx.x < kN / 4u;
                take(a[k].x, b[k].x, c[k].x, ok); take(a[k].y, b[k].y, c[k].y, ok); take(a[k].z, b[k].z, c[k].z, ok); take(a[k].w, b[k].w, c[k].w, ok);
            }
        }
    }
    if (MODE == 0) { if (acc == 12345.678f) sink[0] = acc; return; }
    if (MODE == 3 && acc == 12345.678f) sink[0] = acc;
    // bounds: wave reduce, one word per wave out (stands in for the partials)
    uint32_t v[6] = {mn0, mn1, mn2, mx0, mx1, mx2};
    for (int d = 32; d > 0; d >>= 1)
        for (int q = 0; q < 3; ++q) { v[q] = min(v[q], (uint32_t)__shfl_xor(v[q], d, 64)); v[3 + q] = max(v[3 + q], (uint32_t)__shfl_xor(v[3 + q], d, 64)); }
    if ((threadIdx.x & 63u) == 0) P.part[f][blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5];
    __syncthreads();
    uint16_t* row = P.tab[f] + (size_t)blockIdx.x * kBins;
    for (uint32_t b = threadIdx.x; b < kBins; b += THREADS) row[b] = (uint16_t)s_hist[b];
}

// COLD: a 1.5 GB buffer is streamed through before every timed launch, so that neither the L2s nor the Infinity Cache (256 MB) still
// hold the clouds - the state a batch of the library finds them in.
__global__ void k_flush(const float4* __restrict__ p, size_t n4, float* sink) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 12345.678f) sink[0] = acc;
}
static const float4* g_flush = nullptr;
static size_t g_flush_n4 = 0;
static bool g_cold = false;

template <int THREADS, int VEC, int MODE, int ROUND>
void run(const char* name, const Planes& P, float* sink) {
    const dim3 grid((kN + kTile - 1) / kTile, kFrames);
    const size_t lds = MODE != 0 ? kBins * 4 : 0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        if (g_cold) k_flush<<<2048, 256>>>(g_flush, g_flush_n4, sink);
        hipEventRecord(a);
        k_read<THREADS, VEC, MODE, ROUND><<<grid, THREADS, lds>>>(P, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 1 && ms < best) best = ms;
    }
    printf("%-5s %-54s %7.1f us  %5.2f TB/s\n", g_cold ? "cold" : "warm", name, best * 1e3, 12.0 * kN * kFrames / (best * 1e-3) / 1e12);
}

int main() {
    Planes P;
    // a C3-like cloud: x, y uniform on a 4.5 m field, 80 % of the points on a ground plane (z ~ N(0, 3 mm)), the rest up to 0.6 m above it
    std::vector<float> h(3 * (size_t)kN);
    uint64_t sd = 88172645463325252ull;
    auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (double)(sd >> 11) / 9007199254740992.0; };
    for (uint32_t i = 0; i < kN; ++i) {
        h[i] = (float)(rnd() * 4.5); h[kN + i] = (float)(rnd() * 4.5);
        if (rnd() < 0.8) { const double u1 = rnd() + 1e-12, u2 = rnd(); h[2 * (size_t)kN + i] = (float)(0.003 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2)); }
        else h[2 * (size_t)kN + i] = (float)(0.2 + 0.4 * rnd());
    }
    for (uint32_t f = 0; f < kFrames; ++f) {
        float* d; hipMalloc(&d, 3 * kN * 4);
        hipMemcpy(d, h.data(), 3 * (size_t)kN * 4, hipMemcpyHostToDevice);
        P.x[f] = d; P.y[f] = d + kN; P.z[f] = d + 2 * kN;
        hipMalloc(&P.tab[f], 32 * kBins * 2); hipMalloc(&P.part[f], 32 * 16 * 4);
    }
    float* sink; hipMalloc(&sink, 4);
    { float4* fb; g_flush_n4 = (size_t)96 << 20; hipMalloc(&fb, g_flush_n4 * 16); hipMemset(fb, 0, g_flush_n4 * 16); g_flush = fb; }
    for (int cold = 0; cold < 2; ++cold) {
    g_cold = cold != 0;
    run<1024, 1, 0, 4>("sum, 4-byte loads, 1024 thr, 4 in flight", P, sink);
    run<1024, 1, 0, 8>("sum, 4-byte loads, 1024 thr, 8 in flight", P, sink);
    run<512, 1, 0, 8>("sum, 4-byte loads, 512 thr, 8 in flight", P, sink);
    run<1024, 4, 0, 1>("sum, 16-byte loads, 1024 thr, 1 in flight", P, sink);
    run<1024, 4, 0, 2>("sum, 16-byte loads, 1024 thr, 2 in flight", P, sink);
    run<512, 4, 0, 2>("sum, 16-byte loads, 512 thr, 2 in flight", P, sink);
    run<512, 4, 0, 4>("sum, 16-byte loads, 512 thr, 4 in flight", P, sink);
    run<256, 4, 0, 4>("sum, 16-byte loads, 256 thr, 4 in flight", P, sink);
    run<1024, 1, 1, 4>("hist, 4-byte loads, 1024 thr, 4 in flight (as built)", P, sink);
    run<1024, 1, 1, 8>("hist, 4-byte loads, 1024 thr, 8 in flight", P, sink);
    run<1024, 4, 1, 1>("hist, 16-byte loads, 1024 thr, 1 in flight", P, sink);
    run<1024, 4, 1, 2>("hist, 16-byte loads, 1024 thr, 2 in flight", P, sink);
    run<512, 4, 1, 2>("hist, 16-byte loads, 512 thr, 2 in flight", P, sink);
    run<512, 4, 1, 4>("hist, 16-byte loads, 512 thr, 4 in flight", P, sink);
    run<1024, 4, 2, 2>("keys + bounds, NO LDS atomics, 16-byte loads", P, sink);
    run<1024, 4, 3, 2>("LDS atomics on a cheap key, no bounds, 16-byte loads", P, sink);
    run<1024, 4, 4, 2>("hist with 24-bit multiplies, 16-byte loads", P, sink);
    run<1024, 4, 5, 2>("hist without the bounds, 16-byte loads", P, sink);
    run<1024, 1, 4, 4>("hist with 24-bit multiplies, 4-byte loads", P, sink);
    }
    return 0;
}

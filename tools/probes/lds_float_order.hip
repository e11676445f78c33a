#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((address_space(3))) float lds_float;
__global__ __launch_bounds__(256) void k_probe(uint32_t* __restrict__ viol, uint32_t* __restrict__ denorm_viol, int mode) {
    __shared__ float acc[4][512];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t bad = 0, dbad = 0;
    for (uint32_t trial = 0; trial < 32u; ++trial) {
        for (uint32_t i = lane; i < 512u; i += 64u) acc[w][i] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        const uint32_t ranges[4] = {512u, 64u, 9u, 1u};
        const uint32_t range = ranges[trial & 3u];
        float v[16]; uint32_t id[16];
        for (uint32_t s = 0; s < 16u; ++s) {
            uint32_t h = ((blockIdx.x * 32u + trial) * 16u + s) * 0x9E3779B9u + (threadIdx.x + 1u) * 0x85EBCA6Bu;
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
            id[s] = h % range;
            // magnitude 2^-12 .. 2^12 (or denormal range in mode 1), random mantissa, random sign
            const int e = mode == 1 ? -140 + (int)((h >> 8) & 15u) : -12 + (int)((h >> 8) & 24u ? ((h >> 8) % 25u) : 3);
            const float m = 1.0f + (float)((h >> 3) & 0xFFFFu) * (1.0f / 65536.0f);
            v[s] = ldexpf(((h >> 30) & 1u) ? -m : m, e);
            if (((h >> 20) & 15u) == 0u) v[s] = 0.0f;
        }
        for (uint32_t s = 0; s < 16u; ++s) {
            if (mode == 2) atomicAdd(&acc[w][id[s]], v[s]);
            else __builtin_amdgcn_ds_faddf((lds_float*)&acc[w][id[s]], v[s], __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
        }
        __builtin_amdgcn_wave_barrier();
        // reference: every lane sums, for the word of its step-0 id, all contributions in (step, lane) order
        const uint32_t myid = id[0];
        float ref = 0.0f;
        for (uint32_t s = 0; s < 16u; ++s)
            for (uint32_t l = 0; l < 64u; ++l) {
                const uint32_t il = __shfl(id[s], l, 64);
                const float vl = __shfl(v[s], l, 64);
                if (il == myid) ref += vl;
            }
        const float got = acc[w][myid];
        if (__float_as_uint(got) != __float_as_uint(ref)) { if (mode == 1) ++dbad; else ++bad; }
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(viol, bad);
    if (dbad) atomicAdd(denorm_viol, dbad);
}
int main() {
    uint32_t *d; hipMalloc(&d, 8);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(d, 0, 8);
        k_probe<<<256, 256>>>(d, d + 1, mode);
        uint32_t h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("mode %d: violations %u, denormal-range violations %u (of %d checks)\n", mode, h[0], h[1], 256 * 256 * 32);
    }
    return 0;
}

// Throughput of LDS atomics on gfx950: cycles per wave-instruction, by address pattern.  Stand-alone probe.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KIND>   // 0: ds_add_f32, 1: ds_add_u32 (no return), 2: ds_add_rtn_u32, 3: ds_write_b32
__global__ __launch_bounds__(256) void k_rate(unsigned long long* out, int pattern, int range) {
    __shared__ float acc[4][2048];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t i = lane; i < 2048u; i += 64u) acc[w][i] = 0.0f;
    __syncthreads();
    uint32_t h = (threadIdx.x + 1u) * 0x85EBCA6Bu + blockIdx.x * 0x9E3779B9u;
    uint32_t sink = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 512; ++it) {
        h = h * 1664525u + 1013904223u;
        uint32_t id;
        if (pattern == 0) id = lane;                       // distinct, conflict-free
        else if (pattern == 1) id = (h >> 8) % (uint32_t)range;   // random over range
        else id = (lane & 7u);                             // 8 lanes per word
        if (KIND == 0) atomicAdd(&acc[w][id], 1.0f);
        else if (KIND == 1) atomicAdd(reinterpret_cast<uint32_t*>(&acc[w][id]), 1u);
        else if (KIND == 2) sink += atomicAdd(reinterpret_cast<uint32_t*>(&acc[w][id]), 1u);
        else acc[w][id] = (float)it;
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (sink == 0x12345678u) out[1] = sink;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    const char* kn[4] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_u32", "ds_write_b32"};
    for (int blocks : {256, 2048}) for (int kind = 0; kind < 4; ++kind) for (int pat = 0; pat < 4; ++pat) {
        const int pattern = pat == 3 ? 1 : pat, range = pat == 3 ? 2048 : 128;
        hipMemset(d, 0, 16);
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) k_rate<0><<<blocks, 256>>>(d, pattern, range);
            else if (kind == 1) k_rate<1><<<blocks, 256>>>(d, pattern, range);
            else if (kind == 2) k_rate<2><<<blocks, 256>>>(d, pattern, range);
            else k_rate<3><<<blocks, 256>>>(d, pattern, range);
        }
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const char* pn[4] = {"distinct", "random/128", "8 lanes per word", "random/2048"};
        printf("blocks %4d  %-15s %-18s %7.1f cycles per wave-instruction (4 waves per block issuing)\n", blocks, kn[kind], pn[pat], (double)h[0] / 512.0);
    }
    return 0;
}

// Per-CU THROUGHPUT of the LDS primitives a counting sort can be built from, on gfx950: wall-clock over a whole grid (HIP
// events), every CU busy, at 1 / 2 / 4 / 8 blocks of 256 threads per CU.  Prints ns and shader cycles (at 2.4 GHz) per
// wave-instruction per CU.  Stand-alone probe: hipcc --offload-arch=gfx950 -O3 -o lds_rank_rate lds_rank_rate.hip
//   kind 0  ds_add_u32 (no return)        kind 1  ds_add_rtn_u32
//   kind 2  ds_write_b32                  kind 3  ds_read_b32
//   kind 4  read + write (a plain RMW: what a wave-private counter costs)
//   kind 5  10-bit match by ballots (pure VALU / SALU, no LDS)
//   kind 6  ds_add_rtn_u32 with only every 4th lane active (does the cost follow the active lanes?)
//   kind 7  ds_or_b32 (no return)         kind 8  ds_max_rtn_u32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

constexpr int kIters = 2048;

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint32_t range, uint32_t lds_words) {
    extern __shared__ uint32_t s[];
    for (uint32_t i = threadIdx.x; i < lds_words; i += 256) s[i] = 0u;
    __syncthreads();
    uint32_t h = (threadIdx.x + 1u) * 0x85EBCA6Bu + blockIdx.x * 0x9E3779B9u;
    uint32_t sink = 0;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll 8
    for (int it = 0; it < kIters; ++it) {
        h = h * 1664525u + 1013904223u;
        const uint32_t id = (h >> 8) % range;
        if (KIND == 0) atomicAdd(&s[id], 1u);
        else if (KIND == 1) sink += atomicAdd(&s[id], 1u);
        else if (KIND == 2) s[id] = (uint32_t)it;
        else if (KIND == 3) sink += s[id];
        else if (KIND == 4) { const uint32_t v = s[id]; s[id] = v + 1u; sink += v; }
        else if (KIND == 5) {
            uint64_t peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 10; ++b) {
                const uint64_t m = __ballot((id >> b) & 1u);
                peers &= ((id >> b) & 1u) ? m : ~m;
            }
            sink += __popcll(peers & ((1ull << lane) - 1ull)) + __popcll(peers);
        } else if (KIND == 6) { if ((lane & 3u) == 0u) sink += atomicAdd(&s[id], 1u); }
        else if (KIND == 7) atomicOr(&s[id], 1u << (h & 31u));
        else if (KIND == 8) sink += atomicMax(&s[id], h);
    }
    if (sink == 0x12345678u) out[0] = sink;
}

template <int KIND>
float run(uint32_t* d, int blocks, uint32_t range, uint32_t lds_words) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_rate<KIND><<<blocks, 256, lds_words * 4u>>>(d, range, lds_words);
    hipEventRecord(a);
    k_rate<KIND><<<blocks, 256, lds_words * 4u>>>(d, range, lds_words);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main() {
    uint32_t* d; hipMalloc(&d, 16);
    int cus = 256;
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess) cus = p.multiProcessorCount;
    const char* kn[9] = {"ds_add_u32", "ds_add_rtn_u32", "ds_write_b32", "ds_read_b32", "read+write", "ballot match 10b", "ds_add_rtn 1/4 lanes", "ds_or_b32", "ds_max_rtn_u32"};
    printf("CUs %d, %d iterations per thread, 256 threads per block\n", cus, kIters);
    for (int bpc : {1, 2, 4, 8}) {
        for (uint32_t range : {192u, 2048u, 16384u}) {
            const uint32_t words = range < 2048u ? 2048u : range;
            if (words * 4u * bpc > 160u * 1024u) continue;
            for (int kind = 0; kind < 9; ++kind) {
                float ms = 0;
                const int blocks = cus * bpc;
                switch (kind) {
                    case 0: ms = run<0>(d, blocks, range, words); break;
                    case 1: ms = run<1>(d, blocks, range, words); break;
                    case 2: ms = run<2>(d, blocks, range, words); break;
                    case 3: ms = run<3>(d, blocks, range, words); break;
                    case 4: ms = run<4>(d, blocks, range, words); break;
                    case 5: ms = run<5>(d, blocks, range, words); break;
                    case 6: ms = run<6>(d, blocks, range, words); break;
                    case 7: ms = run<7>(d, blocks, range, words); break;
                    default: ms = run<8>(d, blocks, range, words); break;
                }
                const double winstr_per_cu = (double)bpc * 4.0 * kIters;   // wave-instructions per CU
                const double ns = ms * 1e6 / winstr_per_cu;
                printf("blocks/CU %d  range %5u  %-22s %8.3f ms  %7.2f ns = %6.1f cycles per wave-instruction per CU\n", bpc, range, kn[kind], ms, ns, ns * 2.4);
            }
        }
    }
    return 0;
}

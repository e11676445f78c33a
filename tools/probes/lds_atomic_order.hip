// Probe: does an LDS atomic add with return, executed by the 64 lanes of one wave instruction, hand out its
// old values in lane order among the lanes that hit the same address?  (Not an architectural promise; the
// library checks it on the device it runs on before it relies on it.)
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_order lds_atomic_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned* __restrict__ addr, unsigned* __restrict__ old_out, unsigned* __restrict__ bad, int trials, int nwords) {
    __shared__ unsigned cnt[4][4096];
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        for (int i = lane; i < nwords; i += 64) cnt[w][i] = 0u;
        __builtin_amdgcn_wave_barrier();
        const unsigned a = addr[(size_t)t * 64 + lane];
        const bool active = a != 0xFFFFFFFFu;
        unsigned old = 0u;
        if (active) old = atomicAdd(&cnt[w][a], 1u);
        // a second instruction on the same words: must continue where the first stopped
        unsigned old2 = 0u;
        if (active) old2 = atomicAdd(&cnt[w][a], 1u << 16);
        // expected: number of lower active lanes with the same address
        unsigned expect = 0u;
        for (unsigned l = 0; l < 64; ++l) {
            const unsigned al = __shfl(a, l, 64);
            if (l < lane && al == a) ++expect;
        }
        unsigned total = 0u;
        for (unsigned l = 0; l < 64; ++l) { const unsigned al = __shfl(a, l, 64); if (al == a) ++total; }
        if (active) {
            if (old != expect) atomicAdd(bad, 1u);
            if (old2 != (total | (expect << 16))) atomicAdd(bad + 1, 1u);
            if (w == 0) old_out[(size_t)t * 64 + lane] = old;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int main() {
    const int trials = 1 << 16;
    std::vector<unsigned> h((size_t)trials * 64);
    srand(12345);
    for (int t = 0; t < trials; ++t) {
        const int mode = t % 8;
        const int range = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 4 : mode == 3 ? 16 : mode == 4 ? 64 : mode == 5 ? 256 : mode == 6 ? 2048 : 33;
        for (int l = 0; l < 64; ++l) {
            unsigned a = (unsigned)(rand() % range);
            if (mode == 7) a = a * 32u;                      // same bank, different words
            if (rand() % 16 == 0) a = 0xFFFFFFFFu;           // inactive lane
            h[(size_t)t * 64 + l] = a;
        }
    }
    unsigned *d_addr, *d_old, *d_bad;
    hipMalloc(&d_addr, h.size() * 4); hipMalloc(&d_old, h.size() * 4); hipMalloc(&d_bad, 8);
    hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_bad, 0, 8);
    for (int rep = 0; rep < 20; ++rep) probe<<<2048, 256>>>(d_addr, d_old, d_bad, trials, 4096);
    hipDeviceSynchronize();
    unsigned bad[2];
    hipMemcpy(bad, d_bad, 8, hipMemcpyDeviceToHost);
    printf("lane-ordered LDS atomic add: %s (violations %u, second-instruction violations %u of %d x 64 x 20 x 4 lanes)\n",
           (bad[0] == 0 && bad[1] == 0) ? "HOLDS" : "DOES NOT HOLD", bad[0], bad[1], trials);
    return bad[0] || bad[1];
}

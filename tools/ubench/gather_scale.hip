// Scattered 4-byte loads from an L2-resident table: is the ~0.45 lanes per clock and CU of DESIGN.md 3.1 a limit of the
// L2 (then the chip-wide rate stays when fewer CUs ask) or of each CU's own path (then it scales with the CUs that ask)?
// Sweeps workgroups (CUs in use), waves per workgroup, loads in flight per lane, and dependent fp64 work per load (the
// iterate kernel issues ~130 VALU instructions per visit: what does a gather cost NEXT to that?).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gather_scale gather_scale.hip ; gpurun -- tools/ubench/gather_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

template <int INFLIGHT, int FP>
__global__ void k_gather(const uint32_t* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t* out, double seed) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0000001;
    uint32_t v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) v[k] = table[(lcg(s) >> 4) & mask];
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            // FP unfused fp64 operations in three independent chains, like the map
#pragma unroll
            for (int f = 0; f < FP / 3; ++f) { a = a * c + b; b = b * c; c = c + 1e-9; }
            acc += v[k];                                   // consumes the load issued INFLIGHT steps ago
            v[k] = table[(lcg(s) >> 4) & mask];
        }
    }
    if (acc == 0x12345678u || a + b + c == 1.2345) out[0] = acc;
}

template <typename K>
float time_ms(K k, int blocks, int threads, const uint32_t* table, uint32_t mask, uint32_t iters, uint32_t* out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, table, mask, iters, out, 0.25); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, table, mask, iters, out, 0.25);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    uint32_t* table; CK(hipMalloc(&table, 64u << 20)); CK(hipMemset(table, 1, 64u << 20));
    const uint32_t iters = 600;
    for (uint32_t elems : {1u << 19, 1u << 20}) {  // 2 MiB / 4 MiB of hints
        printf("== table of %u dwords\n", elems);
        for (int blocks : {32, 64, 128, 256, 512, 1024}) {
            for (int threads : {256, 512}) {
                const uint32_t mask = elems - 1;
#define RUN(I, F) ((double)blocks * threads * iters * I / time_ms(k_gather<I, F>, blocks, threads, table, mask, iters, d_out) / 1e6)
                printf("  %4d wg x %3d thr | G loads/s, no fp64: inflight 1 %.0f  2 %.0f  4 %.0f  8 %.0f | 60 fp64 ops per load: 2 %.0f 4 %.0f | 132 per load: 2 %.0f 4 %.0f\n",
                       blocks, threads, RUN(1, 0), RUN(2, 0), RUN(4, 0), RUN(8, 0), RUN(2, 60), RUN(4, 60), RUN(2, 132), RUN(4, 132));
                fflush(stdout);
            }
        }
    }
    return 0;
}

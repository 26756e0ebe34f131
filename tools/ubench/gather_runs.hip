// Micro-benchmark: read throughput of pseudo-random contiguous runs of 64/128/256/512/1024 B out of a 2.3 GB
// arena — the access pattern of the record-accumulate kernel (one chunk per lane group, next address known only
// after the load returns is NOT modelled here: addresses are hashed from a counter, so this is the bandwidth side).
// build: hipcc --offload-arch=gfx950 -O3 -o gather_runs gather_runs.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// LANES lanes cooperate on one run of LANES*16 bytes; DEP = 1 makes the next run's address depend on the data
// just loaded (pointer chase), DEP = 0 hashes it from a counter.
template <int LANES, int DEP>
__global__ void gather(const uint4* __restrict__ arena, uint32_t runs_mask, uint32_t steps, uint32_t* out) {
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint32_t l = threadIdx.x % LANES;
    uint32_t acc = 0, run = mix(gid) & runs_mask;
    for (uint32_t s = 0; s < steps; ++s) {
        const uint4 v = arena[(size_t)run * LANES + l];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        const uint32_t dep = DEP ? __shfl(v.x, 0, LANES) : 0u;
        run = mix(gid * 0x9E3779B9u + s + dep) & runs_mask;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int LANES, int DEP>
float run(const uint4* arena, size_t bytes, uint32_t* out, int blocks, int threads) {
    const size_t runs = bytes / (LANES * 16);
    uint32_t mask = 1; while ((size_t)mask * 2 <= runs) mask *= 2; mask -= 1;
    const size_t groups = (size_t)blocks * threads / LANES;
    const uint32_t steps = (uint32_t)((bytes / (LANES * 16)) / groups) + 1;  // ~ one pass over the arena
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<LANES, DEP>), dim3(blocks), dim3(threads), 0, 0, arena, mask, steps, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<LANES, DEP>), dim3(blocks), dim3(threads), 0, 0, arena, mask, steps, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double moved = (double)groups * steps * LANES * 16;
    return (float)(moved / ms / 1e9);  // TB/s
}

int main() {
    const size_t bytes = (size_t)2304 << 20;
    uint4* arena; uint32_t* out;
    CK(hipMalloc(&arena, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(arena, 1, bytes));
    printf("gather of contiguous runs out of %zu MiB: TB/s (counter-hashed / data-dependent addresses)\n", bytes >> 20);
    for (int wpc : {8, 16, 32}) {  // waves per CU
        const int blocks = 256 * wpc / 4, threads = 256;
        printf("  %2d waves/CU:", wpc);
        printf("  64B %.2f/%.2f", run<4, 0>(arena, bytes, out, blocks, threads), run<4, 1>(arena, bytes, out, blocks, threads));
        printf("  128B %.2f/%.2f", run<8, 0>(arena, bytes, out, blocks, threads), run<8, 1>(arena, bytes, out, blocks, threads));
        printf("  256B %.2f/%.2f", run<16, 0>(arena, bytes, out, blocks, threads), run<16, 1>(arena, bytes, out, blocks, threads));
        printf("  512B %.2f/%.2f", run<32, 0>(arena, bytes, out, blocks, threads), run<32, 1>(arena, bytes, out, blocks, threads));
        printf("  1024B %.2f/%.2f\n", run<64, 0>(arena, bytes, out, blocks, threads), run<64, 1>(arena, bytes, out, blocks, threads));
    }
    return 0;
}

// Where do L2 misses of a scattered-line reader land — Infinity Cache (256 MiB) or HBM? rocprofv3 on this box has no counter that
// separates them (TCC_EA0_RDREQ_DRAM counts the DRAM address space as opposed to GMI / IO and equals TCC_EA0_RDREQ). So: time.
// Every lane reads ONE 2-byte value from a random 128-byte line of a table of S bytes (what a depth-hint miss of the 4096^2 share
// is), S from inside one L2 (2 MiB) over Infinity-Cache sizes (32 .. 192 MiB) to far beyond (1 .. 8 GiB); a first pass warms the
// caches. A knee between 192 MiB and 1 GiB says the Infinity Cache serves the smaller tables; its height says what that is worth.
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip ; gpurun -- tools/ubench/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int INFLIGHT>
__global__ void k_lines(const unsigned short* __restrict__ table, uint64_t n_lines, uint32_t iters, uint32_t* out) {
    uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x2545F4914F6CDD1Dull + 99u;
    uint32_t acc = 0;
    unsigned short v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) v[k] = table[(mix(s) % n_lines) * 64u];
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            acc += v[k];
            const uint64_t r = mix(s);
            v[k] = table[(r % n_lines) * 64u + ((r >> 40) & 63u)];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    const size_t cap = 8ull << 30;
    unsigned short* table; CK(hipMalloc(&table, cap)); CK(hipMemset(table, 1, cap));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 2048, threads = 256;
    printf("scattered 2-byte reads, one per random 128-byte line, %d x %d lanes, 4 in flight per lane\n", blocks, threads);
    for (size_t mb : {2, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 4096, 8192}) {
        const uint64_t n_lines = (mb << 20) / 128u;
        const uint32_t iters = 200;
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {  // (the first pass warms whatever cache holds the table)
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_lines<4>, dim3(blocks), dim3(threads), 0, 0, table, n_lines, iters, d_out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r > 0 && ms < best) best = ms;
        }
        const double lines = (double)blocks * threads * iters * 4;
        printf("  table %5zu MiB: %7.3f ms  %6.1f G lines/s  = %5.2f TB/s of 128-byte lines (%5.2f TB/s at 64 B)\n", mb, best, lines / best / 1e6,
               lines * 128 / best / 1e9, lines * 64 / best / 1e9);
        fflush(stdout);
    }
    return 0;
}

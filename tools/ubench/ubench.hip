// Micro-benchmarks that bound the iterate/accumulate path on MI355X (not part of the product).
//   scatter_load : random 4-byte loads over a region (the depth-hint access pattern)
//   lds_atomic   : random ds_add_rtn_u32 / ds_add_u32 over a table in LDS
//   fp64_chain   : unfused v_mul_f64 + v_add_f64 chains (3 independent chains per lane, like next_point)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/ubench.hip -o ubench && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

template <int ILP>
__global__ void scatter_load(const uint32_t* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t* out) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        uint32_t v[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) v[u] = table[(lcg(s) >> 4) & mask];
#pragma unroll
        for (int u = 0; u < ILP; ++u) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <bool RTN>
__global__ void lds_atomic(uint32_t entries_mask, uint32_t iters, uint32_t* out) {
    extern __shared__ uint32_t tab[];
    for (uint32_t k = threadIdx.x; k <= entries_mask; k += blockDim.x) tab[k] = 0;
    __syncthreads();
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 777u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t a = (lcg(s) >> 4) & entries_mask;
        if (RTN) acc += atomicAdd(&tab[a], 1u);
        else __hip_atomic_fetch_add(&tab[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (acc == 0x12345678u || tab[threadIdx.x & entries_mask] == 0xFFFFFFFFu) out[0] = acc;
}

__global__ void fp64_chain(double c0, double c1, uint32_t iters, double* out) {
    double x = 0.1 + threadIdx.x * 1e-6, y = 0.2, z = 0.3;
    for (uint32_t i = 0; i < iters; ++i) {
        // 3 independent chains of 10 mul + 10 add, separate multiply and add
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            x = x * c0 + c1;
            y = y * c0 + c1;
            z = z * c0 + c1;
        }
    }
    if (x + y + z == 12345.0) out[0] = x;
}

template <typename F>
float time_ms(F f, int reps = 3) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    double* d_outd; CK(hipMalloc(&d_outd, 64));
    // ---- scattered loads
    const size_t maxb = 512u << 20;
    uint32_t* table; CK(hipMalloc(&table, maxb)); CK(hipMemset(table, 1, maxb));
    printf("scatter_load: region, waves/SIMD, ILP -> G loads/s\n");
    for (size_t region : {size_t(3) << 20, size_t(16) << 20, size_t(128) << 20}) {
        size_t pow2 = 1; while (pow2 * 2 <= region) pow2 *= 2;
        const uint32_t mask = (uint32_t)(pow2 / 4 - 1);
        for (int wps : {2, 8}) {
            const int blocks = 256 * wps;  // 256-thread blocks: 4 waves each -> wps blocks per CU
            const uint32_t iters = 2000;
            float ms1 = time_ms([&] { hipLaunchKernelGGL(scatter_load<1>, dim3(blocks), dim3(256), 0, 0, table, mask, iters, d_out); });
            float ms4 = time_ms([&] { hipLaunchKernelGGL(scatter_load<4>, dim3(blocks), dim3(256), 0, 0, table, mask, iters / 4, d_out); });
            const double n = (double)blocks * 256 * iters;
            printf("  %4zu MiB  %d w/SIMD  ILP1 %.1f  ILP4 %.1f\n", pow2 >> 20, wps, n / ms1 / 1e6, n / ms4 / 1e6);
        }
    }
    // ---- LDS atomics
    printf("lds_atomic: table entries, waves/CU -> G atomics/s (rtn / no-rtn)\n");
    CK(hipFuncSetAttribute((const void*)lds_atomic<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)lds_atomic<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (uint32_t entries : {256u, 16384u}) {
        for (int blocks_per_cu : {1, 2}) {
            const int blocks = 256 * blocks_per_cu;
            const uint32_t iters = 4000;
            const size_t lds = (size_t)entries * 4;
            float msr = time_ms([&] { hipLaunchKernelGGL(lds_atomic<true>, dim3(blocks), dim3(256), lds, 0, entries - 1, iters, d_out); });
            float msn = time_ms([&] { hipLaunchKernelGGL(lds_atomic<false>, dim3(blocks), dim3(256), lds, 0, entries - 1, iters, d_out); });
            const double n = (double)blocks * 256 * iters;
            printf("  %6u entries  %d waves/CU  rtn %.1f  no-rtn %.1f\n", entries, 4 * blocks_per_cu, n / msr / 1e6, n / msn / 1e6);
        }
    }
    // ---- fp64 unfused
    printf("fp64_chain: waves/SIMD -> T unfused fp64 op/s\n");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        const uint32_t iters = 20000;
        float ms = time_ms([&] { hipLaunchKernelGGL(fp64_chain, dim3(blocks), dim3(256), 0, 0, 1.0000001, 1e-9, iters, d_outd); });
        const double ops = (double)blocks * 256 * iters * 60.0;
        printf("  %d w/SIMD  %.2f T op/s\n", wps, ops / ms / 1e9);
    }
    return 0;
}

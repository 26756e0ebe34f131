// Scattered 4-byte loads (the depth-hint access of k_iterate_lean) under every cache policy of gfx950's global_load:
// does a policy exist under which a scattered dword costs less than a whole 128-byte line of L2 -> L1 traffic?
//   hipcc --offload-arch=gfx950 -O3 -o gather_policy gather_policy.hip ; gpurun -- tools/ubench/gather_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

#define LOAD_KERNEL(NAME, POLICY, OP, STRIDE_SHIFT)                                                                      \
    __global__ void NAME(const char* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t* out) {                 \
        uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;                                     \
        uint32_t acc = 0;                                                                                                \
        for (uint32_t i = 0; i < iters; ++i) {                                                                           \
            uint32_t v0, v1, v2, v3;                                                                                     \
            const char* p0 = table + ((size_t)((lcg(s) >> 4) & mask) << STRIDE_SHIFT);                                   \
            const char* p1 = table + ((size_t)((lcg(s) >> 4) & mask) << STRIDE_SHIFT);                                   \
            const char* p2 = table + ((size_t)((lcg(s) >> 4) & mask) << STRIDE_SHIFT);                                   \
            const char* p3 = table + ((size_t)((lcg(s) >> 4) & mask) << STRIDE_SHIFT);                                   \
            asm volatile(OP " %0, %4, off " POLICY "\n\t" OP " %1, %5, off " POLICY "\n\t" OP " %2, %6, off " POLICY "\n\t" \
                         OP " %3, %7, off " POLICY "\n\ts_waitcnt vmcnt(0)"                                             \
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory"); \
            acc += v0 + v1 + v2 + v3;                                                                                    \
        }                                                                                                                \
        if (acc == 0x12345678u) out[0] = acc;                                                                            \
    }

LOAD_KERNEL(k_default, "", "global_load_dword", 2)
LOAD_KERNEL(k_nt, "nt", "global_load_dword", 2)
LOAD_KERNEL(k_sc0, "sc0", "global_load_dword", 2)
LOAD_KERNEL(k_sc1, "sc1", "global_load_dword", 2)
LOAD_KERNEL(k_sc0sc1, "sc0 sc1", "global_load_dword", 2)
LOAD_KERNEL(k_sc0nt, "sc0 nt", "global_load_dword", 2)
LOAD_KERNEL(k_all, "sc0 sc1 nt", "global_load_dword", 2)
LOAD_KERNEL(k_u16, "", "global_load_ushort", 1)
LOAD_KERNEL(k_u16_nt, "nt", "global_load_ushort", 1)
LOAD_KERNEL(k_u8, "", "global_load_ubyte", 0)

template <typename K>
float time_ms(K k, int blocks, const char* table, uint32_t mask, uint32_t iters, uint32_t* out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, mask, iters, out); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, mask, iters, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    const size_t maxb = 256u << 20;
    char* table; CK(hipMalloc(&table, maxb)); CK(hipMemset(table, 1, maxb));
    printf("scattered loads, 4 in flight per lane: ELEMENTS (so 4-byte hints cover 4x the bytes of 1-byte ones) -> G loads/s\n");
    for (uint32_t elems : {1u << 19, 1u << 20, 1u << 22, 1u << 24}) {   // 2048^2 has 2^22 pixels; ~2^20 of them are touched
        for (int wps : {2, 3}) {
            const int blocks = 256 * wps;
            const uint32_t iters = 1500;
            const uint32_t mask = elems - 1;
            const double n = (double)blocks * 256 * iters * 4;
#define RUN(K) (n / time_ms(K, blocks, table, mask, iters, d_out) / 1e6)
            printf("  %8u elems %d w/SIMD | u32: default %.0f nt %.0f sc0 %.0f sc1 %.0f sc0sc1 %.0f sc0nt %.0f all %.0f | u16: default %.0f nt %.0f | u8 %.0f\n",
                   elems, wps, RUN(k_default), RUN(k_nt), RUN(k_sc0), RUN(k_sc1), RUN(k_sc0sc1), RUN(k_sc0nt), RUN(k_all), RUN(k_u16), RUN(k_u16_nt), RUN(k_u8));
        }
    }
    return 0;
}

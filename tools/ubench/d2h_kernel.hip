// Device -> page-locked host copy of one image (21.6 MB) by a kernel of ours against the runtime's blit (hipMemcpyAsync), alone and next
// to a chip kept busy by (a) an fp64 arithmetic kernel, (b) a device-memory streaming kernel — what the read-back of the `sequence`
// sweep meets (profiles/r05_experiments.md §5).   hipcc --offload-arch=gfx950 -O3 -o d2h_kernel d2h_kernel.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int PRIO, int NT, int UNROLL>
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16) {
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            if (NT) __builtin_nontemporal_store(u32x4{v[u].x, v[u].y, v[u].z, v[u].w}, (u32x4*)&dst[i + u * stride]);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_busy_fp64(double* out, uint32_t iters, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(1);
    double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5, d = blockIdx.x * 1e-6;
    for (uint32_t i = 0; i < iters; ++i) { a = a * b + c; d = d * b + a; c = c * b + d; b = b * 0.9999999 + 1e-7; }
    if (a + d + c + b == 12345.678) out[0] = a;
}

__global__ void __launch_bounds__(256) k_busy_stream(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n16, uint32_t passes) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = 0; p < passes; ++p)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}

int main() {
    const size_t bytes = 1800ull * 2000 * 6;
    const uint32_t n16 = bytes / 16;
    uint4 *src, *host, *big_in, *big_out;
    double* dout;
    CK(hipMalloc(&src, bytes));
    CK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
    const size_t big = 1ull << 30;
    CK(hipMalloc(&big_in, big)); CK(hipMalloc(&big_out, big)); CK(hipMalloc(&dout, 64));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(big_in, 2, big));
    hipStream_t sc, sb;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    for (int company = 0; company < 4; ++company) {
        const char* cname[] = {"alone", "next to fp64 arithmetic (8 waves per SIMD)", "next to fp64 arithmetic at s_setprio 1", "next to a device-memory stream"};
        auto start_company = [&]() {
            if (company == 1 || company == 2) hipLaunchKernelGGL(k_busy_fp64, dim3(256 * 8), dim3(256), 0, sb, dout, 6000000u, company == 2);
            if (company == 3) hipLaunchKernelGGL(k_busy_stream, dim3(256 * 8), dim3(256), 0, sb, big_in, big_out, (uint32_t)(big / 16), 200u);
        };
        auto timed = [&](const char* name, auto&& launch) {
            CK(hipDeviceSynchronize());
            start_company();
            for (int w = 0; w < 2; ++w) launch();
            CK(hipEventRecord(e0, sc));
            const int reps = 20;
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1, sc));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const bool still = company == 0 || hipStreamQuery(sb) == hipErrorNotReady;
            printf("  %-44s %7.3f ms per image  %5.1f GB/s%s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e9, still ? "" : "  (the company had finished!)");
            fflush(stdout);
            CK(hipDeviceSynchronize());
        };
        printf("%s:\n", cname[company]);
        timed("hipMemcpyAsync (the runtime's blit)", [&]() { CK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, sc)); });
#define RUN(P, NT, U, WG) timed("kernel prio " #P " nt " #NT " unroll " #U " workgroups " #WG, [&]() { hipLaunchKernelGGL((k_copy<P, NT, U>), dim3(WG), dim3(256), 0, sc, src, host, n16); });
        RUN(0, 0, 4, 64) RUN(0, 0, 4, 256) RUN(0, 0, 4, 1024)
        RUN(0, 1, 4, 256) RUN(3, 0, 4, 64) RUN(3, 0, 4, 256) RUN(3, 0, 4, 1024) RUN(3, 1, 4, 256)
        RUN(3, 0, 8, 256) RUN(3, 0, 1, 1024) RUN(3, 0, 8, 1024)
    }
    return 0;
}

// Device-to-host read-back of a frame's image (21.6 MB RGB16 at 1800x2000): hipMemcpyAsync into page-locked memory is a blit KERNEL
// in ROCm 7.2 on these boxes (`__amd_rocclr_copyBuffer` in every kernel trace of the sweep: 0.41 ms per image next to the render).
// What does the SDMA engine reach through the HSA runtime itself (hsa_amd_memory_async_copy), alone and next to a kernel that keeps
// every CU busy — and what does each kind of copy cost that kernel?
//   hipcc --offload-arch=gfx950 -O2 -o d2h_path d2h_path.hip -lhsa-runtime64 ; gpurun -- ./d2h_path
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define HK(x) do { hsa_status_t e = (x); if (e != HSA_STATUS_SUCCESS) { const char* s = ""; hsa_status_string(e, &s); printf("HSA error %s at %d\n", s, __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_busy(double* x, uint32_t* scatter, uint32_t mask, int n) {
    // fp64 chains + one scattered 4-byte store per 16 steps: the render's stand-in (arithmetic and a stream of records)
    double a = x[threadIdx.x & 7] + threadIdx.x, b = 1.0000001, c = 0.5;
    uint32_t h = blockIdx.x * 256u + threadIdx.x;
    for (int i = 0; i < n; ++i) {
        a = a * b + 1e-9; c = c * b + a;
        if ((i & 15) == 0) { h = h * 1664525u + 1013904223u; scatter[h & mask] = (uint32_t)i; }
    }
    if (a + c == 12345.678) x[0] = a;
}

static hsa_agent_t g_gpu, g_cpu;
static bool have_gpu = false, have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = 1800ull * 2000 * 6;
    const int reps = 24;
    CK(hipSetDevice(0));
    HK(hsa_init());
    HK(hsa_iterate_agents(on_agent, nullptr));
    uint32_t free_mask = 0, pref_mask = 0;
    hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &free_mask);
    hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &pref_mask);
    printf("SDMA engines for GPU -> CPU: free mask 0x%x, preferred 0x%x\n", free_mask, pref_mask);
    unsigned char* src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 0x5A, bytes));
    std::vector<unsigned char*> dst(reps);
    for (auto& p : dst) CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    double* x; CK(hipMalloc(&x, 64)); CK(hipMemset(x, 0, 64));
    uint32_t* scatter; const uint32_t words = 1u << 28; CK(hipMalloc(&scatter, (size_t)words * 4));
    hipStream_t sk, sc; CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t k0, k1; CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    std::vector<hsa_signal_t> sig(reps);
    for (auto& s : sig) HK(hsa_signal_create(1, 0, nullptr, &s));

    auto hip_copies = [&](int n) {
        const double t = now_ms();
        for (int i = 0; i < n; ++i) CK(hipMemcpyAsync(dst[i], src, bytes, hipMemcpyDeviceToHost, sc));
        CK(hipStreamSynchronize(sc));
        return now_ms() - t;
    };
    auto hsa_copies = [&](int n, int in_flight) {
        const double t = now_ms();
        for (int i = 0; i < n; ++i) {
            if (i >= in_flight) hsa_signal_wait_scacquire(sig[i - in_flight], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
            hsa_signal_store_relaxed(sig[i], 1);
            HK(hsa_amd_memory_async_copy(dst[i], g_cpu, src, g_gpu, bytes, 0, nullptr, sig[i]));
        }
        for (int i = n > in_flight ? n - in_flight : 0; i < n; ++i) hsa_signal_wait_scacquire(sig[i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        return now_ms() - t;
    };
    auto kernel = [&](int n) {
        CK(hipEventRecord(k0, sk));
        hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, sk, x, scatter, words - 1u, n);
        CK(hipEventRecord(k1, sk));
    };
    auto kernel_ms = [&]() { CK(hipEventSynchronize(k1)); float ms; CK(hipEventElapsedTime(&ms, k0, k1)); return (double)ms; };
    auto check = [&](const char* what) {
        for (int i = 0; i < reps; ++i) if (dst[i][0] != 0x5A || dst[i][bytes - 1] != 0x5A) { printf("%s: WRONG DATA in image %d\n", what, i); exit(1); }
        for (int i = 0; i < reps; ++i) { dst[i][0] = 0; dst[i][bytes - 1] = 0; }
    };

    hip_copies(reps); check("warm-up hip"); hsa_copies(reps, 2); check("warm-up hsa");
    for (int round = 0; round < 2; ++round) {
        double t = hip_copies(reps); check("hip");
        printf("alone   hipMemcpyAsync            %d x %.1f MB: %7.2f ms  %5.1f GB/s\n", reps, bytes / 1e6, t, reps * bytes / t / 1e6);
        for (int fl : {1, 2, 4}) {
            t = hsa_copies(reps, fl); check("hsa");
            printf("alone   hsa_amd_memory_async_copy (%d in flight)   : %7.2f ms  %5.1f GB/s\n", fl, t, reps * bytes / t / 1e6);
        }
    }
    hipStream_t sc2; CK(hipStreamCreateWithFlags(&sc2, hipStreamNonBlocking));
    hipStream_t sc3; CK(hipStreamCreateWithFlags(&sc3, hipStreamNonBlocking));
    // the same images over two / three streams at once (copies pending at the same time take different engines)
    auto hip_copies_split = [&](int n, int ways) {
        hipStream_t ss[3] = {sc, sc2, sc3};
        const double t = now_ms();
        for (int i = 0; i < n; ++i) CK(hipMemcpyAsync(dst[i], src, bytes, hipMemcpyDeviceToHost, ss[i % ways]));
        for (int w = 0; w < ways; ++w) CK(hipStreamSynchronize(ss[w]));
        return now_ms() - t;
    };
    // every image in two halves on two streams
    auto hip_copies_halves = [&](int n) {
        const double t = now_ms();
        for (int i = 0; i < n; ++i) {
            CK(hipMemcpyAsync(dst[i], src, bytes / 2, hipMemcpyDeviceToHost, sc));
            CK(hipMemcpyAsync(dst[i] + bytes / 2, src + bytes / 2, bytes - bytes / 2, hipMemcpyDeviceToHost, sc2));
        }
        CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(sc2));
        return now_ms() - t;
    };
    const int n = 60000;
    kernel(n); double alone = kernel_ms();
    kernel(n); alone = kernel_ms();
    printf("k_busy alone: %.2f ms\n", alone);
    for (int round = 0; round < 2; ++round) {
        kernel(n); double t = hip_copies(reps); double k = kernel_ms(); check("hip busy");
        printf("busy    hipMemcpyAsync            : copies %7.2f ms %5.1f GB/s, kernel %.2f ms (+%.1f %%)\n", t, reps * bytes / t / 1e6, k, (k / alone - 1) * 100);
        for (int ways : {2, 3}) {
            kernel(n); t = hip_copies_split(reps, ways); k = kernel_ms(); check("hip split busy");
            printf("busy    hipMemcpyAsync over %d streams: copies %7.2f ms %5.1f GB/s, kernel %.2f ms (+%.1f %%)\n", ways, t, reps * bytes / t / 1e6, k, (k / alone - 1) * 100);
        }
        kernel(n); t = hip_copies_halves(reps); k = kernel_ms(); check("hip halves busy");
        printf("busy    hipMemcpyAsync, halves on 2 streams: copies %7.2f ms %5.1f GB/s, kernel %.2f ms (+%.1f %%)\n", t, reps * bytes / t / 1e6, k, (k / alone - 1) * 100);
        for (int fl : {1, 2}) {
            kernel(n); t = hsa_copies(reps, fl); k = kernel_ms(); check("hsa busy");
            printf("busy    hsa_amd_memory_async_copy (%d): copies %7.2f ms %5.1f GB/s, kernel %.2f ms (+%.1f %%)\n", fl, t, reps * bytes / t / 1e6, k, (k / alone - 1) * 100);
        }
    }
    return 0;
}

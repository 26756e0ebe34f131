// CU masks (hipExtStreamCreateWithCUMask): which CUs does bit i of the mask name, and what does a device-to-host copy (under a profiler a blit
// kernel in ROCm 7.2 on these boxes) reach on a stream of its own FEW CUs while another stream — masked to the other CUs — keeps
// the chip busy? (The read-back of a `sequence` frame runs at 55 GB/s alone and at 30 GB/s next to the render: DESIGN.md 3.5.)
//   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip ; gpurun -- tools/ubench/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_where(uint32_t* out) {
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    long long t0 = clock64();
    while (clock64() - t0 < 200000) { }
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xFFu) | (((hw >> 13) & 7u) << 8 << 4);
}
__global__ void __launch_bounds__(256) k_busy(double* x, int n) {   // fp64 chains on every lane: the render's stand-in
    double a = x[threadIdx.x & 7] + threadIdx.x, b = 1.0000001, c = 0.5;
    for (int i = 0; i < n; ++i) { a = a * b + 1e-9; c = c * b + a; }
    if (a + c == 12345.678) x[0] = a;
}

static hipStream_t masked(const std::vector<uint32_t>& m) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
    return s;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, words = (cus + 31) / 32;
    printf("%d CUs, %d mask words\n", cus, words);
    uint32_t* d; CK(hipMalloc(&d, 8192 * 4));
    // 1. which CUs are bits [0, 32), [32, 64), every 8th bit, ...
    struct Case { const char* name; std::vector<uint32_t> m; };
    std::vector<Case> cases;
    { std::vector<uint32_t> m(words, 0); m[0] = 0xFFFFFFFFu; cases.push_back({"bits 0..31", m}); }
    { std::vector<uint32_t> m(words, 0); m[1] = 0xFFFFFFFFu; cases.push_back({"bits 32..63", m}); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < cus; b += 8) m[b / 32] |= 1u << (b % 32); cases.push_back({"every 8th bit", m}); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < 8; ++b) m[0] |= 1u << b; cases.push_back({"bits 0..7", m}); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < cus; b += 32) m[b / 32] |= 1u << (b % 32); cases.push_back({"every 32nd bit", m}); }
    for (auto& c : cases) {
        hipStream_t s = masked(c.m);
        CK(hipMemsetAsync(d, 0xFF, 8192 * 4, s));
        hipLaunchKernelGGL(k_where, dim3(2048), dim3(64), 0, s, d);
        CK(hipStreamSynchronize(s));
        std::vector<uint32_t> h(2048); CK(hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost));
        std::map<uint32_t, int> per_xcc; std::map<uint32_t, int> per_cu;
        for (uint32_t v : h) { per_xcc[v >> 16]++; per_cu[v]++; }
        printf("%-16s -> %zu distinct (xcc, se, cu); workgroups per XCD:", c.name, per_cu.size());
        for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    // 2. the copy's rate: 21.6 MB device -> page-locked host, alone / next to a busy chip, on a plain stream / on K CUs of its own
    const size_t bytes = 1800u * 2000u * 6u;
    void *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipHostMalloc(&dst, bytes, 0));
    double* x; CK(hipMalloc(&x, 64)); CK(hipMemset(x, 0, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int k_copy : {0, 8, 16, 32}) {
        std::vector<uint32_t> mc(words, 0), mr(words, 0xFFFFFFFFu);
        if (cus % 32) mr[words - 1] = (1u << (cus % 32)) - 1u;
        // the copy's CUs: every (cus / k)-th bit, so that they spread over the XCDs whatever the bit order turns out to be
        if (k_copy) for (int i = 0; i < k_copy; ++i) { int bit = i * (cus / k_copy); mc[bit / 32] |= 1u << (bit % 32); mr[bit / 32] &= ~(1u << (bit % 32)); }
        hipStream_t sc, sr;
        if (k_copy) { sc = masked(mc); sr = masked(mr); } else { CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking)); }
        for (int busy = 0; busy < 2; ++busy) {
            float best = 1e30f, busy_ms = 0;
            for (int r = 0; r < 4; ++r) {
                hipEvent_t ra, rb; CK(hipEventCreate(&ra)); CK(hipEventCreate(&rb));
                if (busy) { CK(hipEventRecord(ra, sr)); for (int q = 0; q < 6; ++q) hipLaunchKernelGGL(k_busy, dim3(cus * 8), dim3(256), 0, sr, x, 40000); CK(hipEventRecord(rb, sr)); }
                CK(hipEventRecord(a, sc));
                for (int q = 0; q < 8; ++q) CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, sc));
                CK(hipEventRecord(b, sc));
                CK(hipEventSynchronize(b));
                CK(hipStreamSynchronize(sr));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
                if (busy) { CK(hipEventElapsedTime(&busy_ms, ra, rb)); }
                CK(hipEventDestroy(ra)); CK(hipEventDestroy(rb));
            }
            printf("copy on %s, chip %s: %.3f ms per 21.6 MB image = %.1f GB/s%s", k_copy ? "its own CUs" : "a plain stream", busy ? "busy" : "idle",
                   best / 8, bytes * 8 / best / 1e6, busy ? "" : "\n");
            if (busy) printf("; the busy stream's 6 kernels: %.2f ms (%d CUs masked off for the copy)\n", busy_ms, k_copy);
        }
        CK(hipStreamDestroy(sc)); CK(hipStreamDestroy(sr));
    }
    return 0;
}

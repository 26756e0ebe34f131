// Where do the waves of 128-thread workgroups with ~20 KiB of LDS land? (the launch shape of k_iterate_split)
// prints, per (XCD, SE, CU), which (workgroup, wave) sit on which SIMD. hipcc --offload-arch=gfx950 -O2 placement.hip -o placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
__global__ void __launch_bounds__(128) k(uint32_t* out, int spin) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID, all 32 bits
    uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    // keep the workgroup resident for a while so that the whole grid is placed together
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) {
        uint32_t w = blockIdx.x * 2 + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc | (lds[threadIdx.x] << 8);
    }
}
int main(int argc, char** argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 2048;
    int ldsb = argc > 2 ? atoi(argv[2]) : 20352;
    uint32_t* d;
    hipMalloc(&d, grid * 2 * 2 * sizeof(uint32_t));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(grid), dim3(128), ldsb, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(grid * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx94x: se_id [14:13])
    std::map<uint32_t, std::vector<std::pair<uint32_t, uint32_t>>> by_cu;  // key (xcc, se, sh, cu) -> (simd<<8|slot, wave index)
    for (int w = 0; w < grid * 2; ++w) {
        uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 7u;
        uint32_t slot = hw & 15u, simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back({(simd << 8) | slot, (uint32_t)w});
    }
    printf("CUs seen: %zu\n", by_cu.size());
    int shown = 0;
    long prod_hist[4][9] = {};
    for (auto& kv : by_cu) {
        auto v = kv.second;
        std::sort(v.begin(), v.end());
        int per_simd_w0[4] = {0, 0, 0, 0}, per_simd[4] = {0, 0, 0, 0};
        for (auto& p : v) { per_simd[p.first >> 8]++; if ((p.second & 1) == 0) per_simd_w0[p.first >> 8]++; }
        for (int s = 0; s < 4; ++s) prod_hist[s][std::min(per_simd_w0[s], 8)]++;
        if (shown++ < 3) {
            printf("xcc %u se %u sh %u cu %u: %zu waves\n", kv.first >> 16, (kv.first >> 8) & 255, (kv.first >> 4) & 15, kv.first & 15, v.size());
            for (auto& p : v) printf("   simd %u slot %u : wg %u wave %u\n", p.first >> 8, p.first & 255, p.second >> 1, p.second & 1);
        }
    }
    for (int s = 0; s < 4; ++s) {
        printf("simd %d: CUs by number of wave-0s (producers) on it:", s);
        for (int c = 0; c <= 8; ++c) printf(" %d:%ld", c, prod_hist[s][c]);
        printf("\n");
    }
    return 0;
}

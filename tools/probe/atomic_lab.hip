// Lab (not part of the product): fp32 atomic adds at agent scope vs workgroup scope (= executed in the issuing XCD's
// L2) when every address is only touched from ONE XCD, and the XCC_ID hardware register against blockIdx % 8.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/probe/atomic_lab.hip -o tools/probe/atomic_lab_bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ int xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15;
}

__global__ void probe_kernel(int* ids) { if (threadIdx.x == 0) ids[blockIdx.x] = xcc_id(); }

// region r = [r * region_px, (r + 1) * region_px) pixels of 256 floats; workgroup takes region (own XCC) and adds 1.0
// to `rounds` pseudo-random windows of `win` consecutive pixels (clustered like RoI footprints)
template <int SCOPE>
__global__ __launch_bounds__(256) void atom_kernel(float* buf, long region_px, int rounds, int win, int nx, int* counters, int per_region) {
    const int x = xcc_id() % nx;
    __shared__ int job;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (;;) {
        if (threadIdx.x == 0) job = atomicAdd(&counters[x], 1);
        __syncthreads();
        const int j = job;
        __syncthreads();
        if (j >= per_region) return;
        unsigned h = (unsigned)(j * 2654435761u) ^ (x * 40503u);
        const long p0 = (long)(h % (unsigned)(region_px - win));
        float* base = buf + ((long)x * region_px + p0) * 256;
        for (int r = 0; r < rounds; ++r)
            for (int pi = wave; pi < win; pi += 4) {
                float* d = base + (long)pi * 256;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (SCOPE == 0) __hip_atomic_fetch_add(d + lane + 64 * q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else __hip_atomic_fetch_add(d + lane + 64 * q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
    }
}

int main() {
    int* ids; hipMalloc(&ids, 4096 * 4);
    hipLaunchKernelGGL(probe_kernel, dim3(4096), dim3(64), 0, 0, ids);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), ids, 4096 * 4, hipMemcpyDeviceToHost);
    int mism = 0, mx = 0;
    for (int i = 0; i < 4096; ++i) { mism += (h[i] != i % 8); mx = h[i] > mx ? h[i] : mx; }
    printf("XCC_ID: max %d, blocks with id != blockIdx%%8: %d of 4096; first 16:", mx, mism);
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    printf("\n");
    const int nx = mx + 1;
    const long region_px = 131072;                 // 128 MB of fp32 per region
    float* buf; hipMalloc(&buf, (size_t)nx * region_px * 1024);
    int* counters; hipMalloc(&counters, 64);
    const int per_region = 512, rounds = 1, win = 200;
    for (int scope = 0; scope < 2; ++scope) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(buf, 0, (size_t)nx * region_px * 1024);
            hipMemset(counters, 0, 64);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (scope == 0) hipLaunchKernelGGL(atom_kernel<0>, dim3(2048), dim3(256), 0, 0, buf, region_px, rounds, win, nx, counters, per_region);
            else hipLaunchKernelGGL(atom_kernel<1>, dim3(2048), dim3(256), 0, 0, buf, region_px, rounds, win, nx, counters, per_region);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // check: total sum must equal nx * per_region * rounds * win * 256
            std::vector<float> hb((size_t)region_px * 256);
            double tot = 0;
            for (int x = 0; x < nx; ++x) {
                hipMemcpy(hb.data(), buf + (size_t)x * region_px * 256, (size_t)region_px * 1024, hipMemcpyDeviceToHost);
                for (float v : hb) tot += v;
            }
            const double want = (double)nx * per_region * rounds * win * 256;
            printf("scope %s: %.3f ms, %.1f M atomics, %.2f TB/s of 4-byte adds, sum %.0f (want %.0f) %s\n", scope ? "workgroup (L2-local)" : "agent", ms,
                   want / 1e6, want * 4 / ms / 1e9, tot, want, tot == want ? "OK" : "MISMATCH");
        }
    }
    return 0;
}

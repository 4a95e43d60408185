// Calibration of rocprofv3's FETCH_SIZE for the access pattern of k_walk_block (MI355X_MICROARCH.md, HBM section:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   calib_ushort   : every wave reads consecutive 128-byte chunks with 2-byte-per-lane bounds-checked buffer loads,
//                    eight loads in flight (the walk kernel's event loads);
//   calib_dwordx4  : 16 bytes per lane streaming read (the pattern the guide's 1/2 factor was measured on).
// Both read BYTES bytes exactly once (4 GiB: far past the 256 MiB Infinity Cache).
//   hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void calib_ushort(const uint16_t* p, uint64_t n_chunks, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t c = wave * 8; c < n_chunks; c += n_waves * 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t cc = c + u < n_chunks ? c + u : n_chunks - 1;
            const uint64_t a64 = (uint64_t)(uintptr_t)(p + cc * 64);
            const uint64_t addr = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a64) |
                                  ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a64 >> 32)) << 32);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)addr, 0, c + u < n_chunks ? 128 : 0, 0x00020000);
            v[u] = __builtin_amdgcn_raw_buffer_load_b16(rs, 2 * lane, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}

__global__ void calib_dwordx4(const uint4* p, uint64_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}

int main() {
    const uint64_t BYTES = 4ull << 30;
    void* buf; uint32_t* out;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, BYTES);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_ushort, dim3(256 * 6), dim3(256), 0, 0, (const uint16_t*)buf, BYTES / 128, out);
        hipLaunchKernelGGL(calib_dwordx4, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)buf, BYTES / 16, out);
    }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    printf("bytes_read_per_launch %llu\n", (unsigned long long)BYTES);
    return 0;
}

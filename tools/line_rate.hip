// What can the walk kernel's access pattern reach?  Every wave reads whole 128-byte lines, one 2-byte-per-lane buffer load per
// line (k_walk_block's event load), lines chosen at random over a buffer far larger than every cache, U loads in flight
// per wave, W waves per SIMD.  Prints GB/s for U in {8, 16, 32} x {random, sequential} (hipEvent timing, best of 3).
//   hipcc -O3 --offload-arch=gfx950 tools/line_rate.hip -o tools/line_rate.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int U, bool RANDOM>
__global__ __launch_bounds__(256) void k_lines(const uint16_t* p, uint64_t n_lines, uint64_t per_wave, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    uint64_t x = wave * 0x9E3779B97F4A7C15ull + 12345;
    for (uint64_t i = 0; i < per_wave; i += U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t line;
            if (RANDOM) { x = x * 6364136223846793005ull + 1442695040888963407ull; line = (x >> 20) % n_lines; }
            else line = (wave * per_wave + i + u) % n_lines;
            const uint64_t a64 = (uint64_t)(uintptr_t)p + line * 128;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a64), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a64 >> 32));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, 128, 0x00020000);
            v[u] = __builtin_amdgcn_raw_buffer_load_b16(rs, 2 * lane, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}

// the same with B bytes per lane (4, 8, 16): one instruction covers 2, 4 or 8 consecutive lines
template <int U, typename T>
__global__ __launch_bounds__(256) void k_wide(const T* p, uint64_t n_elems, uint64_t per_wave, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < per_wave; i += U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(wave * per_wave + i + u) * 64 + lane];      // stays inside the buffer: checked on the host
#pragma unroll
        for (int u = 0; u < U; ++u) acc += ((const uint32_t*)&v[u])[0];
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}
// Two DIFFERENT 128-byte lines per load instruction: lanes 0-31 read line i of one stream as dwords, lanes 32-63 line i of a second
// stream `gap` lines away (what pairing two pileup entries in one instruction would look like: 256 bytes per instruction, but the
// two lines are not adjacent in memory).  Tells instruction shape apart from DRAM adjacency.
template <int U>
__global__ __launch_bounds__(256) void k_pairs(const uint32_t* p, uint64_t n_lines, uint64_t per_wave, uint64_t gap, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < per_wave; i += U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t a = (wave * 2 * per_wave + i + u) % n_lines, b = (a + per_wave + gap) % n_lines;
            v[u] = p[((lane < 32) ? a : b) * 32 + (lane & 31)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}
template <int U>
static void run_pairs(const void* buf, uint64_t n_lines, uint32_t* out, int waves_per_simd, uint64_t gap) {
    const unsigned blocks = 256u * (unsigned)waves_per_simd;
    const uint64_t per_wave = 2048;                                   // instructions per wave = 2 x 2048 lines
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_pairs<U>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)buf, n_lines, per_wave, gap, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)blocks * 4 * per_wave * 256;
    printf("two lines per instruction, second stream %llu lines away, U=%2d waves/SIMD=%d  %.2f ms  %.0f GB/s\n", (unsigned long long)gap, U, waves_per_simd, best, bytes / best / 1e6);
}

template <int U, typename T>
static void run_wide(const void* buf, uint64_t bytes, uint32_t* out, int waves_per_simd) {
    const unsigned blocks = 256u * (unsigned)waves_per_simd;
    const uint64_t per_wave = 4096 * 2 / sizeof(T) * 2;              // same bytes per wave as the line test x 2
    const double total = (double)blocks * 4 * per_wave * 64 * sizeof(T);
    if (total > (double)bytes) { printf("buffer too small for the %zu-byte test\n", sizeof(T)); return; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_wide<U, T>), dim3(blocks), dim3(256), 0, 0, (const T*)buf, bytes / sizeof(T), per_wave, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("sequential %2zu B/lane U=%2d waves/SIMD=%d  %.2f ms  %.0f GB/s\n", sizeof(T), U, waves_per_simd, best, total / best / 1e6);
}

template <int U, bool RANDOM>
static void run(const uint16_t* buf, uint64_t n_lines, uint32_t* out, int waves_per_simd) {
    const unsigned blocks = 256u * (unsigned)waves_per_simd;          // 256 CUs x 4 SIMDs x W waves / 4 waves per block
    const uint64_t per_wave = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_lines<U, RANDOM>), dim3(blocks), dim3(256), 0, 0, buf, n_lines, per_wave, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)blocks * 4 * per_wave * 128;
    printf("%-10s U=%2d waves/SIMD=%d  %.2f ms  %.0f GB/s\n", RANDOM ? "random" : "sequential", U, waves_per_simd, best, bytes / best / 1e6);
}

int main(int argc, char** argv) {
    const uint64_t BYTES = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 24ull) << 30;
    void* buf; uint32_t* out;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, BYTES);
    (void)hipDeviceSynchronize();
    const uint64_t n_lines = BYTES / 128;
    for (int w : {4, 8}) {
        run<8, true>((const uint16_t*)buf, n_lines, out, w);
        run<16, true>((const uint16_t*)buf, n_lines, out, w);
        run<32, true>((const uint16_t*)buf, n_lines, out, w);
        run<16, false>((const uint16_t*)buf, n_lines, out, w);
        run_wide<16, uint32_t>(buf, BYTES, out, w);
        run_wide<16, uint2>(buf, BYTES, out, w);
        run_wide<8, uint4>(buf, BYTES, out, w);
        run_pairs<16>(buf, n_lines, out, w, 0);
        run_pairs<16>(buf, n_lines, out, w, 19);
        run_pairs<16>(buf, n_lines, out, w, 1000003);
    }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    return 0;
}

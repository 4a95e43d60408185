// Scattered reads the way the count makes them: a wave works through "tiles"; a tile's lines lie scattered over a few MB of a buffer far
// larger than the caches (the reads of one gene); U loads in flight per wave.  One aligned BLOCK per load instruction: 128 bytes as 2 bytes per
// lane (k_tm_count_direct over tile-phased events), or 256 bytes as 4 bytes per lane (what a 128-position window would fetch).
//   hipcc -O3 --offload-arch=gfx950 tools/block_rate.hip -o tools/block_rate.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define AS1 __attribute__((address_space(1)))

template <int U, int BYTES>      // BYTES per lane: 2 -> 128-byte blocks, 4 -> 256-byte blocks, 8 -> 512
__global__ __launch_bounds__(128) void k_blocks(const uint8_t* p, uint64_t n_regions, uint32_t region_blocks, uint32_t per_tile, uint32_t tiles, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    uint64_t x = wave * 0x9E3779B97F4A7C15ull + 12345;
    const uint32_t off = (uint32_t)lane * BYTES;
    for (uint32_t t = 0; t < tiles; ++t) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t region = ((x >> 24) % n_regions) * (uint64_t)region_blocks * (64u * BYTES);
        for (uint32_t i = 0; i < per_tile; i += U) {
            uint32_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                const uint64_t a = (uint64_t)(uintptr_t)p + region + (uint64_t)((uint32_t)(x >> 33) % region_blocks) * (64u * BYTES);
                const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
                uint32_t o = off;
                asm volatile("" : "+v"(o));
                if (BYTES == 2) v[u] = (uint32_t)(int32_t)*(const AS1 int16_t*)((const AS1 char*)(uintptr_t)sb + (uint64_t)o);
                else if (BYTES == 4) v[u] = *(const AS1 uint32_t*)((const AS1 char*)(uintptr_t)sb + (uint64_t)o);
                else v[u] = (uint32_t)*(const AS1 uint64_t*)((const AS1 char*)(uintptr_t)sb + (uint64_t)o);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}

// the two waves of a workgroup fetch the two 128-byte halves of the SAME scattered 256-byte blocks, 2 bytes per lane each (no barrier: they drift as they will)
template <int U>
__global__ __launch_bounds__(128) void k_halves(const uint8_t* p, uint64_t n_regions, uint32_t region_blocks, uint32_t per_tile, uint32_t tiles, uint32_t* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t acc = 0;
    uint64_t x = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull + 12345;
    const uint32_t off = (uint32_t)lane * 2u + (uint32_t)wv * 128u;
    for (uint32_t t = 0; t < tiles; ++t) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t region = ((x >> 24) % n_regions) * (uint64_t)region_blocks * 256u;
        for (uint32_t i = 0; i < per_tile; i += U) {
            uint32_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                const uint64_t a = (uint64_t)(uintptr_t)p + region + (uint64_t)((uint32_t)(x >> 33) % region_blocks) * 256u;
                const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
                uint32_t o = off;
                asm volatile("" : "+v"(o));
                v[u] = (uint32_t)(int32_t)*(const AS1 int16_t*)((const AS1 char*)(uintptr_t)sb + (uint64_t)o);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}
template <int U>
static void run_halves(const uint8_t* buf, uint64_t bytes, uint32_t* out, int waves_per_simd, uint32_t region_mb) {
    const unsigned blocks = 256u * 4u * (unsigned)waves_per_simd / 2u;
    const uint32_t region_blocks = region_mb * (1u << 20) / 256u;
    const uint64_t n_regions = bytes / ((uint64_t)region_mb << 20);
    const uint32_t per_tile = 1024, tiles = 24;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_halves<U>), dim3(blocks), dim3(128), 0, 0, buf, n_regions, region_blocks, per_tile, tiles, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double n_blk = (double)blocks * tiles * per_tile;
    printf(" 256-byte blocks as two 128-byte halves by the two waves of a workgroup, %3u MB, U=%2d waves/SIMD=%d: %.3f ms  %.2f G blocks/s  %.0f GB/s\n", region_mb, U, waves_per_simd, best,
           n_blk / best / 1e6, n_blk * 256 / best / 1e6);
    fflush(stdout);
}

template <int U, int BYTES>
static void run(const uint8_t* buf, uint64_t bytes, uint32_t* out, int waves_per_simd, uint32_t region_mb) {
    const unsigned blocks = 256u * 4u * (unsigned)waves_per_simd / 2u;       // two waves per workgroup
    const uint32_t region_blocks = region_mb * (1u << 20) / (64u * BYTES);
    const uint64_t n_regions = bytes / ((uint64_t)region_mb << 20);
    const uint32_t per_tile = 1024, tiles = 24;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_blocks<U, BYTES>), dim3(blocks), dim3(128), 0, 0, buf, n_regions, region_blocks, per_tile, tiles, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double n_blk = (double)blocks * 2 * tiles * per_tile;
    printf("%4d-byte blocks scattered over %3u MB, U=%2d waves/SIMD=%d: %.3f ms  %.2f G blocks/s  %.0f GB/s\n", 64 * BYTES, region_mb, U, waves_per_simd, best, n_blk / best / 1e6,
           n_blk * 64 * BYTES / best / 1e6);
    fflush(stdout);
}

int main() {
    const uint64_t bytes = 16ull << 30;
    uint8_t* buf = nullptr; uint32_t* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    for (int w : {6, 8}) { run_halves<16>(buf, bytes, out, w, 4); run_halves<32>(buf, bytes, out, w, 4); }
    for (uint32_t mb : {4u}) {
        for (int w : {6}) {
            run<16, 2>(buf, bytes, out, w, mb);
            run<32, 2>(buf, bytes, out, w, mb);
            run<16, 4>(buf, bytes, out, w, mb);
            run<32, 4>(buf, bytes, out, w, mb);
            run<16, 8>(buf, bytes, out, w, mb);
        }
    }
    (void)hipFree(buf); (void)hipFree(out);
    return 0;
}

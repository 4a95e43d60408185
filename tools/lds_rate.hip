// What does an LDS atomic cost a CU?  Every wave issues ds_add_u32 / ds_add_u64 / ds_write_b32 / ds_or_b32 at its own conflict-free words
// (lane-consecutive), U back to back, W waves per SIMD; prints cycles per instruction and CU at 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_rate.hip -o tools/lds_rate.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k_lds(int iters, uint32_t* out) {
    __shared__ __attribute__((aligned(4096))) uint32_t buf[4][8 * 128];      // per wave: 8 rows of 128 words
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 8 * 128; i += 256) (&buf[0][0])[i] = 0;
    __syncthreads();
    const uint32_t a32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&buf[wv][lane];          // 64 consecutive words
    const uint32_t a64 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&buf[wv][2 * lane];      // 64 consecutive 8-byte words
    // (two half-waves in the SAME banks: lanes 0..31 at 8-byte words 0..31, lanes 32..63 at the same words 2 KB further on)
    const uint32_t a64s = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&buf[wv][0] + (uint32_t)(lane & 31) * 8u + (uint32_t)(lane >> 5) * 2048u;
    uint32_t one = 1; uint64_t one64 = 0x100000001ull;
    asm volatile("" : "+v"(one), "+v"(one64));
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) asm volatile("ds_add_u32 %0, %1\n\tds_add_u32 %0, %1 offset:512\n\tds_add_u32 %0, %1 offset:1024\n\tds_add_u32 %0, %1 offset:1536\n\t"
                                  "ds_add_u32 %0, %1 offset:2048\n\tds_add_u32 %0, %1 offset:2560\n\tds_add_u32 %0, %1 offset:3072\n\tds_add_u32 %0, %1 offset:3584" :: "v"(a32), "v"(one) : "memory");
        if (OP == 1) asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %1 offset:512\n\tds_add_u64 %0, %1 offset:1024\n\tds_add_u64 %0, %1 offset:1536\n\t"
                                  "ds_add_u64 %0, %1 offset:2048\n\tds_add_u64 %0, %1 offset:2560\n\tds_add_u64 %0, %1 offset:3072\n\tds_add_u64 %0, %1 offset:3584" :: "v"(a64), "v"(one64) : "memory");
        if (OP == 2) asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:512\n\tds_write_b32 %0, %1 offset:1024\n\tds_write_b32 %0, %1 offset:1536\n\t"
                                  "ds_write_b32 %0, %1 offset:2048\n\tds_write_b32 %0, %1 offset:2560\n\tds_write_b32 %0, %1 offset:3072\n\tds_write_b32 %0, %1 offset:3584" :: "v"(a32), "v"(one) : "memory");
        if (OP == 3) asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %1 offset:512\n\tds_or_b32 %0, %1 offset:1024\n\tds_or_b32 %0, %1 offset:1536\n\t"
                                  "ds_or_b32 %0, %1 offset:2048\n\tds_or_b32 %0, %1 offset:2560\n\tds_or_b32 %0, %1 offset:3072\n\tds_or_b32 %0, %1 offset:3584" :: "v"(a32), "v"(one) : "memory");
        if (OP == 4) asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:512\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:1536\n\t"
                                  "ds_write_b64 %0, %1 offset:2048\n\tds_write_b64 %0, %1 offset:2560\n\tds_write_b64 %0, %1 offset:3072\n\tds_write_b64 %0, %1 offset:3584" :: "v"(a64), "v"(one64) : "memory");
        if (OP == 6) asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %1 offset:256\n\tds_add_u64 %0, %1 offset:512\n\tds_add_u64 %0, %1 offset:768\n\t"
                                  "ds_add_u64 %0, %1 offset:1024\n\tds_add_u64 %0, %1 offset:1280\n\tds_add_u64 %0, %1 offset:1536\n\tds_add_u64 %0, %1 offset:1792" :: "v"(a64s), "v"(one64) : "memory");
        if (OP == 5) {        // half the lanes masked off (what an entry of 32 events does)
            asm volatile("s_mov_b64 exec, 0xffffffff\n\tds_add_u32 %0, %1\n\tds_add_u32 %0, %1 offset:512\n\tds_add_u32 %0, %1 offset:1024\n\tds_add_u32 %0, %1 offset:1536\n\t"
                         "ds_add_u32 %0, %1 offset:2048\n\tds_add_u32 %0, %1 offset:2560\n\tds_add_u32 %0, %1 offset:3072\n\tds_add_u32 %0, %1 offset:3584\n\ts_mov_b64 exec, -1" :: "v"(a32), "v"(one) : "memory");
        }
    }
    __syncthreads();
    if (buf[wv][lane] == 0xdeadbeefu) out[0] = 1;
}

template <int OP>
static void run(uint32_t* out, int waves_per_simd, const char* what) {
    const unsigned blocks = 256u * (unsigned)waves_per_simd;
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_lds<OP>), dim3(blocks), dim3(256), 0, 0, iters, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
    const double ns = best * 1e6 / instr_per_cu;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns per instruction and CU = %.1f cycles at 2.4 GHz\n", what, waves_per_simd, best, ns, ns * 2.4);
    fflush(stdout);
}

int main() {
    uint32_t* out = nullptr;
    if (hipMalloc(&out, 64) != hipSuccess) return 1;
    for (int w : {2, 4}) {
        run<0>(out, w, "ds_add_u32");
        run<1>(out, w, "ds_add_u64");
        run<2>(out, w, "ds_write_b32");
        run<3>(out, w, "ds_or_b32");
        run<4>(out, w, "ds_write_b64");
        run<5>(out, w, "ds_add_u32, 32 lanes");
        run<6>(out, w, "ds_add_u64, halves 2 KB apart");
    }
    (void)hipFree(out);
    return 0;
}

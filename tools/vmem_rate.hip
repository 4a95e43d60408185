// What does ONE vector memory instruction cost a CU's address unit, whatever it fetches?  Every wave issues global_load_sshort (2 bytes per
// lane, one 128-byte line per instruction: k_tm_count_direct's event load) at lines of a 16 KB region per CU (L1 / L2 hits: no fabric
// traffic), U in flight, under an EXEC mask of `active` lanes.  Prints ns per instruction and CU, and the same in cycles at 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 tools/vmem_rate.hip -o tools/vmem_rate.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define AS1 __attribute__((address_space(1)))

template <int BYTES>
__global__ __launch_bounds__(256) void k_rate(const uint8_t* p, unsigned long long mask, int iters, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    uint32_t off = (uint32_t)lane * BYTES;
    uint64_t base0 = (uint64_t)(uintptr_t)p + (uint64_t)(wave & 1023u) * 16384u;
    // (the loads in ONE asm statement: nothing the compiler computes runs under the mask; 16 lines of the wave's 16 KB, 512 bytes apart)
    const uint64_t b0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base0);
    const uint64_t b1 = b0 + 4096u;
    for (int i = 0; i < iters; ++i) {
        uint32_t v[16];
#define LD(OP)                                                                                                   \
        asm volatile("s_mov_b64 exec, %[m]\n\t"                                                                   \
                     OP " %0, %[o], %[b0]\n\t" OP " %1, %[o], %[b0] offset:512\n\t" OP " %2, %[o], %[b0] offset:1024\n\t" OP " %3, %[o], %[b0] offset:1536\n\t"        \
                     OP " %4, %[o], %[b0] offset:2048\n\t" OP " %5, %[o], %[b0] offset:2560\n\t" OP " %6, %[o], %[b0] offset:3072\n\t" OP " %7, %[o], %[b0] offset:3584\n\t" \
                     OP " %8, %[o], %[b1]\n\t" OP " %9, %[o], %[b1] offset:512\n\t" OP " %10, %[o], %[b1] offset:1024\n\t" OP " %11, %[o], %[b1] offset:1536\n\t"        \
                     OP " %12, %[o], %[b1] offset:2048\n\t" OP " %13, %[o], %[b1] offset:2560\n\t" OP " %14, %[o], %[b1] offset:3072\n\t" OP " %15, %[o], %[b1] offset:3584\n\t" \
                     "s_mov_b64 exec, -1\n\t"                                                                     \
                     "s_waitcnt vmcnt(0)"                                                                          \
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),           \
                       "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])    \
                     : [o] "v"(off), [b0] "s"(b0), [b1] "s"(b1), [m] "s"(mask) : "memory")
        if (BYTES == 2) LD("global_load_sshort"); else LD("global_load_dword");
#undef LD
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}

template <int BYTES>
static void run(const uint8_t* buf, uint32_t* out, unsigned long long mask, int waves_per_simd, const char* what) {
    const unsigned blocks = 256u * (unsigned)waves_per_simd;          // 256 CUs x 4 SIMDs x waves / 4 waves per block
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_rate<BYTES>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double instr_per_cu = (double)blocks * 4 * iters * 16 / 256.0;
    const double ns = best * 1e6 / instr_per_cu;
    printf("%-34s %d B/lane waves/SIMD=%d  %.3f ms  %.2f ns per instruction and CU = %.1f cycles at 2.4 GHz\n", what, BYTES, waves_per_simd, best, ns, ns * 2.4);
    fflush(stdout);
}

int main() {
    uint8_t* buf = nullptr; uint32_t* out = nullptr;
    if (hipMalloc(&buf, 1024u * 16384u + 4096) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, 1024u * 16384u + 4096);
    for (int w : {4, 8}) {
        run<2>(buf, out, ~0ull, w, "all 64 lanes");
        run<2>(buf, out, 0xffffffffull, w, "lanes 0..31");
        run<2>(buf, out, 0xffffull, w, "lanes 0..15");
        run<2>(buf, out, 0xfull, w, "lanes 0..3");
        run<2>(buf, out, 0x1ull, w, "lane 0");
        run<2>(buf, out, 0ull, w, "no lane (EXEC = 0)");
        run<2>(buf, out, 0x0000ffffffff0000ull, w, "lanes 16..47");
        run<4>(buf, out, ~0ull, w, "all 64 lanes");
        run<4>(buf, out, 0xffffffffull, w, "lanes 0..31");
    }
    (void)hipFree(buf); (void)hipFree(out);
    return 0;
}

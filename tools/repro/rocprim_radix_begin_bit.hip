// Reproduction: hipcub::DeviceRadixSort::SortPairsDescending with begin_bit > 0 (ROCm 7.2.0, gfx950) returns wrong pairs when the
// input is small enough for rocprim's merge path (5 000 and 227 397 items: wrong; 2.27 M: right; begin_bit 0: right at every size).
// store.hip sorts shifted keys from bit 0 for that reason.  hipcc --offload-arch=gfx950 -O2 -o rs rocprim_radix_begin_bit.hip && ./rs
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <vector>
#include <algorithm>
#include <cstdio>
int run(int n, int b0, int b1, bool nb) {
    std::vector<uint32_t> k(n), v(n);
    uint32_t s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; k[i] = (s >> 8) % 20 + ((s >> 28) == 0 ? (s >> 12) % 20000 : 0); v[i] = i * 3 + 7; }
    uint32_t *dk, *dko, *dv, *dvo; void* tmp;
    hipMalloc(&dk, n * 4); hipMalloc(&dko, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&dvo, n * 4);
    hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t st; if (nb) hipStreamCreateWithFlags(&st, hipStreamNonBlocking); else st = 0;
    size_t tb = 0;
    hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, dk, dko, dv, dvo, n, b0, b1, st);
    hipMalloc(&tmp, tb + 256);
    hipcub::DeviceRadixSort::SortPairsDescending(tmp, tb, dk, dko, dv, dvo, n, b0, b1, st);
    hipStreamSynchronize(st);
    std::vector<uint32_t> ko(n), vo(n);
    hipMemcpy(ko.data(), dko, n * 4, hipMemcpyDeviceToHost); hipMemcpy(vo.data(), dvo, n * 4, hipMemcpyDeviceToHost);
    std::vector<int> idx(n); for (int i = 0; i < n; ++i) idx[i] = i;
    uint32_t mask = (b1 >= 32 ? 0xffffffffu : ((1u << b1) - 1)) & ~((1u << b0) - 1);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (k[a] & mask) > (k[b] & mask); });
    int bad = 0; for (int i = 0; i < n; ++i) bad += ko[i] != k[idx[i]] || vo[i] != v[idx[i]];
    printf("n %d bits %d..%d nonblocking %d tmp %zu: %d wrong\n", n, b0, b1, (int)nb, tb, bad);
    return bad;
}
int main() { run(227397, 11, 32, false); run(227397, 11, 32, true); run(227397, 0, 32, true); run(227397, 0, 21, true); run(2273970, 11, 32, true); run(5000, 11, 32, true); return 0; }

// Host build of longsom_amd/csrc/inflate_core.h (the GPU's per-lane DEFLATE decoder) against zlib: every stream zlib's deflate writes
// (levels 0-9, default / fixed / huffman-only / RLE strategies, random and compressible data, BGZF-sized) must inflate to the same
// bytes; truncated and corrupted streams must fail cleanly (the binary is built with -fsanitize=address,undefined by the test).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <zlib.h>
#include "../../longsom_amd/csrc/inflate_core.h"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& src, int level, int strategy) {
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) { fprintf(stderr, "deflateInit2 failed\n"); exit(2); }
    std::vector<uint8_t> out(src.size() + src.size() / 8 + 1024);
    zs.next_in = (Bytef*)src.data(); zs.avail_in = (uInt)src.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "deflate failed\n"); exit(2); }
    out.resize(zs.total_out); deflateEnd(&zs);
    return out;
}

static std::vector<uint8_t> make_data(int kind, size_t n) {
    std::vector<uint8_t> d(n);
    for (size_t i = 0; i < n; ++i) {
        switch (kind) {
            case 0: d[i] = (uint8_t)rnd(); break;                                  // incompressible
            case 1: d[i] = "ACGT"[rnd() & 3]; break;                               // 2 bits of entropy per byte
            case 2: d[i] = (uint8_t)(i < 64 ? rnd() : d[i - 1 - (rnd() % 64)]); break;   // long matches at short distances
            case 3: d[i] = (uint8_t)((i / 97) & 0xff); break;                      // runs
            default: d[i] = (uint8_t)((rnd() % 100) < 90 ? 'A' + (rnd() % 4) : rnd()); break;   // BAM-like: mostly a few symbols
        }
    }
    return d;
}

int main() {
    std::vector<uint8_t> tab(lsi::T_SYM); std::vector<uint8_t> lens(lsi::T_LENS);
    lsi::Tab t{tab.data(), lens.data(), 1};
    long n_ok = 0, n_bad = 0;
    const size_t sizes[] = {0, 1, 2, 17, 255, 4096, 65280, 65536};
    const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
    for (int kind = 0; kind < 5; ++kind)
        for (size_t n : sizes)
            for (int level : {0, 1, 6, 9})
                for (int st : strategies) {
                    const std::vector<uint8_t> src = make_data(kind, n), z = deflate_raw(src, level, st);
                    std::vector<uint8_t> out(n + 1, 0xEE);
                    const int rc = lsi::inflate_raw(z.data(), z.size(), out.data(), n, t);
                    if (rc != 0 || (n && memcmp(out.data(), src.data(), n) != 0) || out[n] != 0xEE) {
                        fprintf(stderr, "MISMATCH kind %d n %zu level %d strategy %d rc %d\n", kind, n, level, st, rc); return 1;
                    }
                    ++n_ok;
                    // wrong expected size, truncation, bit flips: must return an error or (bit flips) any result, never touch memory outside
                    if (n > 0 && lsi::inflate_raw(z.data(), z.size(), out.data(), n - 1, t) == 0) { fprintf(stderr, "short output accepted\n"); return 1; }
                    std::vector<uint8_t> out2(n + 2, 0xEE);
                    if (lsi::inflate_raw(z.data(), z.size(), out2.data(), n + 1, t) == 0) { fprintf(stderr, "long output accepted\n"); return 1; }
                    if (z.size() > 2) {
                        std::vector<uint8_t> cut(z.begin(), z.begin() + (long)(z.size() / 2));
                        if (lsi::inflate_raw(cut.data(), cut.size(), out.data(), n, t) == 0 && n > 8) { fprintf(stderr, "truncated stream accepted\n"); return 1; }
                        for (int k = 0; k < 8; ++k) {
                            std::vector<uint8_t> bad = z;
                            bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() & 7));
                            std::vector<uint8_t> o(n + 1, 0xEE);
                            (void)lsi::inflate_raw(bad.data(), bad.size(), o.data(), n, t);
                            if (o[n] != 0xEE) { fprintf(stderr, "wrote past the output\n"); return 1; }
                            ++n_bad;
                        }
                    }
                }
    // the strided table layout the device uses ([index][lane], stride 64): same result through lane 37 of a 64-lane image
    {
        std::vector<uint8_t> img((size_t)lsi::T_SYM * 64, 0xCD); std::vector<uint8_t> limg((size_t)lsi::T_LENS * 64, 0xAB);
        lsi::Tab ts{img.data() + 37, limg.data() + 37, 64};
        const std::vector<uint8_t> src = make_data(4, 60000), z = deflate_raw(src, 6, Z_DEFAULT_STRATEGY);
        std::vector<uint8_t> out(src.size());
        if (lsi::inflate_raw(z.data(), z.size(), out.data(), out.size(), ts) != 0 || out != src) { fprintf(stderr, "strided tables failed\n"); return 1; }
        for (size_t i = 0; i < img.size(); ++i) if ((i & 63) != 37 && img[i] != 0xCD) { fprintf(stderr, "strided tables wrote another lane's word\n"); return 1; }
        for (size_t i = 0; i < limg.size(); ++i) if ((i & 63) != 37 && limg[i] != 0xAB) { fprintf(stderr, "strided tables wrote another lane's byte\n"); return 1; }
    }
    printf("inflate ok: %ld streams equal zlib, %ld corrupted streams handled\n", n_ok, n_bad);
    return 0;
}

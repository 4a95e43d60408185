// CPU fuzz driver of the BAM reader (longsom_amd/csrc/hostio/bamio.cpp), built with -fsanitize=address,undefined by
// tests/test_bamio_fuzz_cpu.py.  Takes a valid BAM, and for N seeded iterations damages it — truncation, bit flips and
// overwritten length fields in the UNCOMPRESSED stream (re-framed as BGZF so the damage reaches the record parser), and flips in
// the BGZF framing itself — then runs lsio_decode_bam / lsio_stream_next / lsio_split_bam / lsio_build_bai on the result.  Every call must return
// (0 or an error code); the sanitizers turn any read past a buffer into a failure of this program.
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct lsio_decoded; struct lsio_stream;
extern "C" {
int lsio_decode_bam(const char*, const char*, int32_t, const int32_t*, int32_t, int32_t, lsio_decoded**);
void lsio_free_decoded(lsio_decoded*);
int lsio_stream_open(const char*, const char*, int32_t, const int32_t*, int32_t, int32_t, lsio_stream**);
int lsio_stream_next(lsio_stream*, int64_t, lsio_decoded**);
void lsio_stream_close(lsio_stream*);
int lsio_split_bam(const char*, const char*, int32_t, const uint8_t*, int32_t, const char*, int32_t, int64_t*);
int lsio_build_bai(const char*, const char*);
}

static uint64_t rng_state = 1;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static std::vector<uint8_t> read_file(const char* p) {
    FILE* f = fopen(p, "rb"); std::vector<uint8_t> v;
    if (!f) return v;
    fseek(f, 0, SEEK_END); v.resize((size_t)ftell(f)); fseek(f, 0, SEEK_SET);
    if (fread(v.data(), 1, v.size(), f) != v.size()) v.clear();
    fclose(f); return v;
}
static std::vector<uint8_t> inflate_bgzf(const std::vector<uint8_t>& raw) {
    std::vector<uint8_t> out; size_t off = 0;
    while (off + 18 <= raw.size()) {
        const uint8_t* h = raw.data() + off;
        const uint32_t xlen = h[10] | (h[11] << 8), bsize = (h[16] | (h[17] << 8)) + 1u;
        const uint32_t usize = h[bsize - 4] | (h[bsize - 3] << 8) | (h[bsize - 2] << 16) | ((uint32_t)h[bsize - 1] << 24);
        const size_t at = out.size(); out.resize(at + usize);
        if (usize) { z_stream zs; memset(&zs, 0, sizeof(zs)); inflateInit2(&zs, -15); zs.next_in = (Bytef*)(h + 12 + xlen); zs.avail_in = bsize - xlen - 20;
                     zs.next_out = out.data() + at; zs.avail_out = usize; inflate(&zs, Z_FINISH); inflateEnd(&zs); }
        off += bsize;
    }
    return out;
}
static void write_bgzf(const char* path, const std::vector<uint8_t>& data, size_t block) {
    FILE* f = fopen(path, "wb");
    for (size_t at = 0; at <= data.size(); at += block) {
        const size_t n = at < data.size() ? (data.size() - at < block ? data.size() - at : block) : 0;
        uint8_t out[70000]; z_stream zs; memset(&zs, 0, sizeof(zs)); deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)(data.data() + (n ? at : 0)); zs.avail_in = (uInt)n; zs.next_out = out + 18; zs.avail_out = sizeof(out) - 26;
        deflate(&zs, Z_FINISH); const uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
        const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0}; memcpy(out, hdr, 16);
        const uint32_t bsize = clen + 25; out[16] = (uint8_t)(bsize & 0xff); out[17] = (uint8_t)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data.data() + (n ? at : 0), (uInt)n);
        uint8_t* t = out + 18 + clen;
        for (int i = 0; i < 4; ++i) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
        fwrite(out, 1, 18 + clen + 8, f);
        if (n == 0) break;
    }
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: fuzz_bamio in.bam workdir iterations seed\n"); return 2; }
    const std::string work = argv[2]; const int iters = atoi(argv[3]); rng_state = strtoull(argv[4], nullptr, 10) * 2654435761ull + 88172645463325252ull;
    const std::vector<uint8_t> raw = read_file(argv[1]);
    const std::vector<uint8_t> plain = inflate_bgzf(raw);
    if (plain.size() < 64) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const std::string tmp = work + "/fuzz.bam", o0 = work + "/o0.bam", o1 = work + "/o1.bam";
    const std::string outs = o0 + "\n" + o1;
    const uint8_t ct[2] = {0, 1};
    int ok = 0, failed = 0;
    for (int it = 0; it < iters; ++it) {
        const int kind = (int)(rnd() % 5);
        if (kind == 4) {                                                   // damage the BGZF framing / compressed bytes
            std::vector<uint8_t> r = raw;
            const int n = 1 + (int)(rnd() % 6);
            for (int k = 0; k < n; ++k) { const size_t at = rnd() % r.size(); r[at] ^= (uint8_t)(1u << (rnd() % 8)); }
            if (rnd() % 3 == 0) r.resize(rnd() % r.size());
            FILE* f = fopen(tmp.c_str(), "wb"); fwrite(r.data(), 1, r.size(), f); fclose(f);
        } else {
            std::vector<uint8_t> p = plain;
            if (kind == 0) p.resize(rnd() % p.size());                                                  // truncation anywhere
            else if (kind == 1) { const int n = 1 + (int)(rnd() % 16); for (int k = 0; k < n; ++k) p[rnd() % p.size()] ^= (uint8_t)(1u << (rnd() % 8)); }
            else if (kind == 2) { const size_t at = rnd() % (p.size() - 4); const uint32_t v = (rnd() % 2) ? 0xffffffffu : (uint32_t)rnd(); memcpy(&p[at], &v, 4); }
            else { const size_t at = rnd() % p.size(); const size_t n = 1 + rnd() % 64; for (size_t k = 0; k < n && at + k < p.size(); ++k) p[at + k] = (uint8_t)rnd(); }
            write_bgzf(tmp.c_str(), p, 1 + rnd() % 0xff00);
        }
        lsio_decoded* d = nullptr;
        const int rc = lsio_decode_bam(tmp.c_str(), nullptr, (it & 1) ? -1 : 0, nullptr, 60, 2, &d);
        if (rc == 0) { ++ok; lsio_free_decoded(d); } else ++failed;
        lsio_stream* st = nullptr;
        if (lsio_stream_open(tmp.c_str(), "AAAC0001GG\nTTTG0002CC", 2, nullptr, 60, 2, &st) == 0) {
            for (int b = 0; b < 1000; ++b) { d = nullptr; const int r = lsio_stream_next(st, 1 + (int64_t)(rnd() % 20000), &d); if (r == 1) lsio_free_decoded(d); else break; }
            lsio_stream_close(st);
        }
        int64_t cnt[5];
        (void)lsio_split_bam(tmp.c_str(), "AAAC0001GG\nTTTG0002CC", 2, ct, 2, outs.c_str(), 60, cnt);
        (void)lsio_build_bai(tmp.c_str(), (work + "/fuzz.bam.bai").c_str());
    }
    printf("fuzz_bamio: %d inputs decoded, %d rejected\n", ok, failed);
    return 0;
}

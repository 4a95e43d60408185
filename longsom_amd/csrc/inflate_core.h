// Raw DEFLATE (RFC 1951) decoder of ONE BGZF block by ONE thread, written for a GPU lane: no recursion, no allocation, every loop
// bounded by the input or the output size, tables in caller-provided storage addressed with a stride (device: LDS laid out
// [index][lane], stride 64; host tests: stride 1).  Canonical-code decoding (count per code length + symbols in code order), the
// lit/len and distance code-length counts kept in registers, one table read per decoded symbol.
//
// This is the device half of reading a BAM (the reference reads BAMs through pysam / htslib's bgzf + zlib,
// BaseCellCounter.py:190-191, SplitBamCellTypes.py:51-65).  The same source is compiled for the host by tests/native/test_inflate.cpp,
// which checks it byte for byte against zlib's inflate on random, compressible, stored, fixed-code and corrupted streams.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __HIPCC__
#define LSI_FN __host__ __device__ __forceinline__
#else
#define LSI_FN inline
#endif

namespace lsi {

constexpr int MAXBITS = 15, MAXLCODES = 286, MAXDCODES = 30, FIXLCODES = 288, MAXCODES = MAXLCODES + MAXDCODES;
// per-thread table storage, every element at [index * stride]: T_WORDS uint16_t
//   [0, 288)    lit/len symbols in code order      [288, 318)  distance symbols in code order
//   [318, 334)  count per length (construction)    [334, 350)  next offset per length (construction)
// and T_LENS bytes: the code lengths being read (dynamic header) / scratch.  1020 bytes per thread: two waves of 64 threads per CU.
constexpr int T_LSYM = 0, T_DSYM = 288, T_CNT = 318, T_OFFS = 334, T_WORDS = 350, T_LENS = 320;

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

struct Tab {
    uint16_t* base; uint8_t* lens; int stride;
    LSI_FN uint16_t get(int i) const { return base[(size_t)i * stride]; }
    LSI_FN void set(int i, uint16_t v) const { base[(size_t)i * stride] = v; }
    LSI_FN uint32_t len(int i) const { return lens[(size_t)i * stride]; }
    LSI_FN void set_len(int i, uint32_t v) const { lens[(size_t)i * stride] = (uint8_t)v; }
};

// Where the compressed bytes come from.  PlainIn: memory the thread reads directly.  csrc/ingest.hip has a second one (a wave's LDS
// window over the stream, refilled a kilobyte at a time by all lanes).
struct PlainIn {
    const uint8_t* in;
    LSI_FN uint8_t byte(size_t pos) { return in[pos]; }
    LSI_FN uint32_t word(size_t pos) {         // four bytes, little endian (the caller knows they are there)
#ifdef __HIP_DEVICE_COMPILE__
        return *reinterpret_cast<const u32_unaligned*>(in + pos);
#else
        return (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
#endif
    }
};

// bits are consumed LSB first; reading past the end of the input sets `bad` and yields zeros.  A refill is ONE four-byte load, and the
// word it adds was requested a refill earlier (`nxt`): with a lane per stream every lane of a wave waits for each load any of them
// issues, so the loads are few, sit at the same instruction for all lanes (inflate_to calls fill() where the lanes are converged: a
// symbol and its extra bits need at most 28 bits, a refill leaves more than 32) and have their latency behind them when they are used.
template <class In>
struct Bits {
    In src; size_t n, pos; uint64_t buf; int cnt; int bad; uint32_t nxt; int have;       // nxt = bytes [pos, pos + 4) when have
    LSI_FN void fill() {
        if (cnt > 32) return;
        if (have) { buf |= (uint64_t)nxt << cnt; cnt += 32; pos += 4; }
        have = pos + 4 <= n;
        if (have) nxt = src.word(pos);
        else while (cnt <= 56 && pos < n) { buf |= (uint64_t)src.byte(pos++) << cnt; cnt += 8; }      // the stream's last bytes
    }
    LSI_FN uint32_t take(int need) {          // need <= 16
        if (cnt < need) { fill(); if (cnt < need) fill(); if (cnt < need) { bad = 1; cnt = 0; buf = 0; return 0; } }
        const uint32_t v = (uint32_t)(buf & ((1ull << need) - 1ull));
        buf >>= need; cnt -= need;
        return v;
    }
};

struct Counts { uint16_t c[MAXBITS + 1]; };          // codes per length (kept in registers: the decode loop is fully unrolled)

// builds the canonical decoding tables of n symbols whose code lengths are lens[first .. first + n): symbols in code order at
// tab[sym_at ..], counts per length in *cnt.  Returns 0 for a complete code, > 0 incomplete, < 0 over-subscribed (as zlib's puff).
LSI_FN int construct(const Tab& t, int first, int n, int sym_at, Counts* cnt) {
    for (int len = 0; len <= MAXBITS; ++len) t.set(T_CNT + len, 0);
    for (int s = 0; s < n; ++s) { const int l = (int)(t.len(first + s) & 15u); t.set(T_CNT + l, (uint16_t)(t.get(T_CNT + l) + 1)); }
    int left = 1;
    for (int len = 1; len <= MAXBITS; ++len) { left <<= 1; left -= (int)t.get(T_CNT + len); if (left < 0) break; }
    uint16_t off = 0;
    t.set(T_OFFS + 1, 0);
    for (int len = 1; len < MAXBITS; ++len) { off = (uint16_t)(off + t.get(T_CNT + len)); t.set(T_OFFS + len + 1, off); }
    for (int s = 0; s < n; ++s) {
        const int l = (int)(t.len(first + s) & 15u);
        if (l) { const uint16_t o = t.get(T_OFFS + l); t.set(sym_at + o, (uint16_t)s); t.set(T_OFFS + l, (uint16_t)(o + 1)); }
    }
    for (int len = 0; len <= MAXBITS; ++len) cnt->c[len] = t.get(T_CNT + len);
    return left;
}

// one symbol of the code described by (cnt, symbols at sym_at); -1 when the bits run out or no code matches
template <class In>
LSI_FN int decode(Bits<In>& b, const Tab& t, const Counts& cnt, int sym_at) {
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= MAXBITS; ++len) {
        code |= (int)b.take(1);
        const int count = cnt.c[len];
        if (code - count < first) return b.bad ? -1 : (int)t.get(sym_at + index + (code - first));
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}

// Where the decoded bytes go.  PlainOut: a buffer of exactly n_out bytes (one thread per stream; host tests).  csrc/ingest.hip has a
// second one (a wave's LDS ring, flushed to global memory in chunks, with the match copies spread over the wave's lanes).
struct PlainOut {
    uint8_t* out; size_t n_out, pos;
    LSI_FN bool room(uint32_t len) const { return pos + len <= n_out; }
    LSI_FN void lit(uint8_t v) { out[pos++] = v; }
    LSI_FN bool copy(uint32_t dist, uint32_t len) {
        if (dist > pos) return false;
#if defined(__HIP_DEVICE_COMPILE__) || defined(LSI_WIDE_COPY)          // (LSI_WIDE_COPY: the host test runs the device's copy under ASan)
        // A lane per stream: the lanes of a wave wait for the longest copy among them at every turn, and a byte-by-byte copy pays a
        // memory round trip per byte.  Eight bytes a turn when the source lies at least eight behind (a chunk then never overlaps its
        // own destination), four such chunks in flight when the whole match lies behind the write position.
        uint8_t* d = out + pos; const uint8_t* s = d - dist;
        pos += len;
        if (dist >= len)
            for (; len >= 32; len -= 32, s += 32, d += 32) {
                const uint64_t a = *reinterpret_cast<const u64_unaligned*>(s), b = *reinterpret_cast<const u64_unaligned*>(s + 8),
                               c = *reinterpret_cast<const u64_unaligned*>(s + 16), e = *reinterpret_cast<const u64_unaligned*>(s + 24);
                *reinterpret_cast<u64_unaligned*>(d) = a; *reinterpret_cast<u64_unaligned*>(d + 8) = b;
                *reinterpret_cast<u64_unaligned*>(d + 16) = c; *reinterpret_cast<u64_unaligned*>(d + 24) = e;
            }
        if (dist >= 8)
            for (; len >= 8; len -= 8, s += 8, d += 8) *reinterpret_cast<u64_unaligned*>(d) = *reinterpret_cast<const u64_unaligned*>(s);
        for (; len; --len) *d++ = *s++;
#else
        for (uint32_t i = 0; i < len; ++i) { out[pos] = out[pos - dist]; ++pos; }
#endif
        return true;
    }
    LSI_FN bool done() const { return pos == n_out; }
};

// Inflates the raw DEFLATE stream in[0, n_in) into `o`: the stream must produce EXACTLY the bytes o has room for (BGZF's ISIZE).
// Returns 0 on success, a negative code otherwise (never reads outside the input, never writes where o.room() said no).
template <class In, class Out>
LSI_FN int inflate_to(In src, size_t n_in, Out& o, const Tab& t) {
    const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Bits<In> b{src, n_in, 0, 0, 0, 0, 0, 0};
    int last = 0;
    while (!last) {
        last = (int)b.take(1);
        const int type = (int)b.take(2);
        if (b.bad) return -1;
        if (type == 0) {                                   // stored
            b.buf >>= (b.cnt & 7); b.cnt -= (b.cnt & 7);   // to the byte boundary
            const uint32_t len = b.take(16), nlen = b.take(16);
            if (b.bad || (len ^ 0xffffu) != nlen) return -2;
            if (!o.room(len)) return -3;
            for (uint32_t i = 0; i < len; ++i) o.lit((uint8_t)b.take(8));
            if (b.bad) return -1;
            continue;
        }
        if (type == 3) return -4;
        Counts lc, dc;
        if (type == 1) {                                   // fixed codes
            for (int s = 0; s < 144; ++s) t.set_len(s, 8);
            for (int s = 144; s < 256; ++s) t.set_len(s, 9);
            for (int s = 256; s < 280; ++s) t.set_len(s, 7);
            for (int s = 280; s < FIXLCODES; ++s) t.set_len(s, 8);
            construct(t, 0, FIXLCODES, T_LSYM, &lc);
            for (int s = 0; s < MAXDCODES; ++s) t.set_len(s, 5);
            construct(t, 0, MAXDCODES, T_DSYM, &dc);
        } else {                                           // dynamic codes
            const int nlen = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
            if (b.bad || nlen > MAXLCODES || ndist > MAXDCODES) return -5;
            for (int i = 0; i < 19; ++i) t.set_len(i, 0);
            for (int i = 0; i < ncode; ++i) t.set_len(order[i], b.take(3));
            if (b.bad) return -1;
            Counts cc;
            if (construct(t, 0, 19, T_DSYM, &cc) != 0) return -6;                  // the code-length code must be complete (its symbols borrow the distance table's place)
            int idx = 0;
            while (idx < nlen + ndist) {
                b.fill();
                int sym = decode(b, t, cc, T_DSYM);
                if (sym < 0) return -7;
                if (sym < 16) { t.set_len(idx, (uint32_t)sym); ++idx; }
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (idx == 0) return -8; val = (int)t.len(idx - 1); rep = 3 + (int)b.take(2); }
                    else if (sym == 17) rep = 3 + (int)b.take(3);
                    else rep = 11 + (int)b.take(7);
                    if (b.bad || idx + rep > nlen + ndist) return -9;
                    while (rep--) { t.set_len(idx, (uint32_t)val); ++idx; }
                }
            }
            if (t.len(256) == 0) return -10;                             // no end-of-block code
            int err = construct(t, 0, nlen, T_LSYM, &lc);
            if (err < 0 || (err > 0 && nlen - (int)lc.c[0] != 1)) return -11;        // incomplete only for a single code
            err = construct(t, nlen, ndist, T_DSYM, &dc);
            if (err < 0 || (err > 0 && ndist - (int)dc.c[0] != 1)) return -12;
        }
        for (;;) {                                         // every turn writes at least one byte or ends the block
            b.fill();
            int sym = decode(b, t, lc, T_LSYM);
            if (sym < 0) return -13;
            if (sym < 256) { if (!o.room(1)) return -3; o.lit((uint8_t)sym); continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return -14;
            const uint32_t len = lbase[sym] + b.take(lext[sym]);
            b.fill();
            const int ds = decode(b, t, dc, T_DSYM);
            if (ds < 0 || ds >= 30) return -15;
            const uint32_t dist = dbase[ds] + b.take(dext[ds]);
            if (b.bad) return -1;
            if (!o.room(len)) return -3;
            if (!o.copy(dist, len)) return -16;
        }
    }
    return o.done() ? 0 : -17;
}

// one thread, a plain output buffer
LSI_FN int inflate_raw(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out, const Tab& t) {
    PlainOut o{out, n_out, 0};
    return inflate_to(PlainIn{in}, n_in, o, t);
}

} // namespace lsi

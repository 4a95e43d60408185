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
// per-thread table storage, every element at [index * stride].
//   sym (T_SYM bytes; device: LDS, [index][lane]): the decoding tables proper, all a symbol's decode reads
//     [0, 288)    low bytes of the lit/len symbols in code order      [288, 324)  their ninth bits (symbols 256..285), one bit a symbol
//     [324, 356)  distance symbols in code order (and, while a dynamic header is read, the code-length code's)
//     [356, 388)  lit/len: per code length, (index of the length's first symbol) - (its first code), 16-bit little endian
//     [388, 420)  the same for the distance code (the code-length code's while a header is read)
//   lens (T_LENS bytes; device: global memory): what only the head of a block touches
//     [0, 320)    the code lengths being read      [320, 384)  count per length and next offset per length, 16-bit little endian
// 420 bytes of LDS per thread: five waves of 64 threads per CU (a 16-bit word per symbol was 700: three waves, a SIMD without a wave).
constexpr int T_LLO = 0, T_LHI = 288, T_DSYM = 324, T_LBASE = 356, T_DBASE = 388, T_SYM = 420, T_W = 320, T_CNT = 0, T_OFFS = 16, T_LENS = 384;

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

LSI_FN uint32_t brev32(uint32_t v) {
#if defined(__clang__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

struct Tab {
    uint8_t* sym; uint8_t* lens; int stride;
    LSI_FN uint32_t lsym(int i) const { return (uint32_t)sym[(size_t)(T_LLO + i) * stride] | ((((uint32_t)sym[(size_t)(T_LHI + (i >> 3)) * stride] >> (i & 7)) & 1u) << 8); }
    LSI_FN void set_lsym(int i, uint32_t v) const {
        sym[(size_t)(T_LLO + i) * stride] = (uint8_t)v;
        if (v & 256u) { uint8_t* h = sym + (size_t)(T_LHI + (i >> 3)) * stride; *h = (uint8_t)(*h | (1u << (i & 7))); }
    }
    LSI_FN void clear_lhi() const { for (int i = T_LHI; i < T_DSYM; ++i) sym[(size_t)i * stride] = 0; }
    LSI_FN uint32_t dsym(int i) const { return sym[(size_t)(T_DSYM + i) * stride]; }
    LSI_FN void set_dsym(int i, uint32_t v) const { sym[(size_t)(T_DSYM + i) * stride] = (uint8_t)v; }
    LSI_FN int base(int at, int l) const { return (int)(int16_t)(uint16_t)(sym[(size_t)(at + 2 * l) * stride] | (sym[(size_t)(at + 2 * l + 1) * stride] << 8)); }
    LSI_FN void set_base(int at, int l, int v) const { sym[(size_t)(at + 2 * l) * stride] = (uint8_t)v; sym[(size_t)(at + 2 * l + 1) * stride] = (uint8_t)((uint32_t)v >> 8); }
    LSI_FN uint32_t len(int i) const { return lens[(size_t)i * stride]; }
    LSI_FN void set_len(int i, uint32_t v) const { lens[(size_t)i * stride] = (uint8_t)v; }
    LSI_FN uint16_t get(int i) const { return (uint16_t)(lens[(size_t)(T_W + 2 * i) * stride] | (lens[(size_t)(T_W + 2 * i + 1) * stride] << 8)); }      // the construction's 16-bit words
    LSI_FN void set(int i, uint16_t v) const { lens[(size_t)(T_W + 2 * i) * stride] = (uint8_t)v; lens[(size_t)(T_W + 2 * i + 1) * stride] = (uint8_t)(v >> 8); }
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

// What a symbol's decode compares against, in registers (the comparisons are unrolled): lim[l] = the end of the length-l codes' range when
// every code is written left-justified in fifteen bits (codes of a canonical code grow with their length: the ranges follow each other,
// an empty length's end is its predecessor's); zeros = the symbols without a code.
struct Counts { uint16_t lim[MAXBITS + 1]; uint16_t zeros; };

// builds the canonical decoding tables of n symbols whose code lengths are lens[first .. first + n): symbols in code order in the
// lit/len table (LIT) or the distance table, counts per length in *cnt.  Returns 0 for a complete code, > 0 incomplete, < 0 over-subscribed (as zlib's puff).
template <bool LIT>
LSI_FN int construct(const Tab& t, int first, int n, Counts* cnt) {
    if (LIT) t.clear_lhi();
    for (int len = 0; len <= MAXBITS; ++len) t.set(T_CNT + len, 0);
    for (int s = 0; s < n; ++s) { const int l = (int)(t.len(first + s) & 15u); t.set(T_CNT + l, (uint16_t)(t.get(T_CNT + l) + 1)); }
    int left = 1;
    for (int len = 1; len <= MAXBITS; ++len) { left <<= 1; left -= (int)t.get(T_CNT + len); if (left < 0) break; }
    uint16_t off = 0;
    t.set(T_OFFS + 1, 0);
    for (int len = 1; len < MAXBITS; ++len) { off = (uint16_t)(off + t.get(T_CNT + len)); t.set(T_OFFS + len + 1, off); }
    for (int s = 0; s < n; ++s) {
        const int l = (int)(t.len(first + s) & 15u);
        if (l) { const uint16_t o = t.get(T_OFFS + l); if (LIT) t.set_lsym(o, (uint32_t)s); else t.set_dsym(o, (uint32_t)s); t.set(T_OFFS + l, (uint16_t)(o + 1)); }
    }
    {
        int fst = 0, index = 0;
        for (int len = 1; len <= MAXBITS; ++len) {
            const int count = (int)t.get(T_CNT + len);
            const int lim = (fst + count) << (MAXBITS - len);
            cnt->lim[len] = (uint16_t)(lim > 0xffff ? 0xffff : lim);           // (an over-subscribed code - refused by the caller - could pass 2^15)
            t.set_base(LIT ? T_LBASE : T_DBASE, len, index - fst);
            index += count; fst = (fst + count) << 1;
        }
        cnt->lim[0] = 0; cnt->zeros = t.get(T_CNT + 0);
    }
    return left;
}

// one symbol of the code described by (cnt, the lit/len or the distance table); -1 when the bits run out or no code matches.
// Fifteen PEEKED bits, reversed into a left-justified code, are compared with the lengths' range ends: the code's length is one more
// than the number of ends it has reached - no loop, no branch; its symbol sits at base[length] + the code.  (A lane per stream executes
// what ANY lane of its wave needs: the bit-by-bit walk left early for a short code in one lane, but the wave ran to the longest code
// among 64, paid a taken branch a length and a refill test a bit.)
template <bool LIT, class In>
LSI_FN int decode(Bits<In>& b, const Tab& t, const Counts& cnt) {
    if (b.cnt < MAXBITS) { b.fill(); if (b.cnt < MAXBITS) b.fill(); }          // (the stream's end may leave fewer: the bits beyond are zeros)
    const uint32_t c15 = brev32((uint32_t)b.buf) >> 17;
    int len = 1;
#pragma unroll
    for (int l = 1; l < MAXBITS; ++l) len += c15 >= cnt.lim[l] ? 1 : 0;
    if (c15 >= cnt.lim[MAXBITS] || len > b.cnt) { if (c15 < cnt.lim[MAXBITS]) b.bad = 1; return -1; }
    b.buf >>= len; b.cnt -= len;
    const int at = t.base(LIT ? T_LBASE : T_DBASE, len) + (int)(c15 >> (MAXBITS - len));
    return (int)(LIT ? t.lsym(at) : t.dsym(at));
}

// Where the decoded bytes go.  PlainOut: a buffer of exactly n_out bytes (one thread per stream).
struct PlainOut {
    uint8_t* out; size_t n_out, pos;
    uint64_t stage; uint32_t ns;               // (wide build) the last ns < 8 literals, bytes [pos - ns, pos), not yet in memory
    LSI_FN bool room(uint32_t len) const { return pos + len <= n_out; }
#if defined(__HIP_DEVICE_COMPILE__) || defined(LSI_WIDE_COPY)          // (LSI_WIDE_COPY: the host test runs the device's code under ASan)
    // A lane per stream: every store and every load of a lane is a memory request of its own (64 lanes, 64 different lines), and the
    // memory system takes ~30 G scattered requests a second whatever their size - the decoder was bound by their NUMBER (more waves per
    // CU did not make it faster).  So literals are gathered eight to a store, and a match moves eight bytes a request even when it is
    // shorter (what lands behind its end is written again by whatever the stream produces next: a block fills its n_out bytes exactly).
    static LSI_FN void st8(uint8_t* d, uint64_t v) { *reinterpret_cast<u64_unaligned*>(d) = v; }
    static LSI_FN uint64_t ld8(const uint8_t* s) { return *reinterpret_cast<const u64_unaligned*>(s); }
    LSI_FN void lit(uint8_t v) {
        stage |= (uint64_t)v << (8 * ns); ++ns; ++pos;
        if (ns == 8) { st8(out + pos - 8, stage); stage = 0; ns = 0; }
    }
    LSI_FN void flush() {
        if (!ns) return;
        uint8_t* d = out + pos - ns;
        if (pos - ns + 8 <= n_out) st8(d, stage);
        else for (uint32_t i = 0; i < ns; ++i) d[i] = (uint8_t)(stage >> (8 * i));
        stage = 0; ns = 0;
    }
    LSI_FN bool copy(uint32_t dist, uint32_t len) {
        if (dist > pos) return false;
        flush();
        uint8_t* d = out + pos; const uint8_t* s = d - dist;
        uint8_t* const end8 = out + n_out - (n_out < 8 ? n_out : 8);      // the last place an 8-byte store may start (n_out >= 8)
        pos += len;
        if (dist >= 8) {
            if (dist >= len)                                              // the whole match lies behind the write position: four chunks in flight
                for (; len >= 32; len -= 32, s += 32, d += 32) {
                    const uint64_t a = ld8(s), b = ld8(s + 8), c = ld8(s + 16), e = ld8(s + 24);
                    st8(d, a); st8(d + 8, b); st8(d + 16, c); st8(d + 24, e);
                }
            for (; len >= 8; len -= 8, s += 8, d += 8) st8(d, ld8(s));
            if (len && n_out >= 8 && d <= end8) { st8(d, ld8(s)); len = 0; }
        } else if (n_out >= 8 && d <= end8) {
            // a source less than eight bytes behind: the match repeats a pattern of `dist` bytes.  One word holds the pattern from its
            // first byte; it is stored every `step` = the largest multiple of dist <= 8 bytes
            uint64_t v = ld8(s);
            if (dist < 8) v &= (1ull << (8 * dist)) - 1ull;
            for (uint32_t have = dist; have < 8; have *= 2) v |= v << (8 * have);
            const uint32_t step = (8u / dist) * dist;
            for (; len >= step && d <= end8; len -= step, d += step) st8(d, v);
            if (len && d <= end8) { st8(d, v); len = 0; }
            s = d - dist;
        }
        for (; len; --len) *d++ = *s++;
        return true;
    }
    LSI_FN bool done() { flush(); return pos == n_out; }
#else
    LSI_FN void lit(uint8_t v) { out[pos++] = v; }
    LSI_FN bool copy(uint32_t dist, uint32_t len) {
        if (dist > pos) return false;
        for (uint32_t i = 0; i < len; ++i) { out[pos] = out[pos - dist]; ++pos; }
        return true;
    }
    LSI_FN bool done() const { return pos == n_out; }
#endif
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
            construct<true>(t, 0, FIXLCODES, &lc);
            for (int s = 0; s < MAXDCODES; ++s) t.set_len(s, 5);
            construct<false>(t, 0, MAXDCODES, &dc);
        } else {                                           // dynamic codes
            const int nlen = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
            if (b.bad || nlen > MAXLCODES || ndist > MAXDCODES) return -5;
            for (int i = 0; i < 19; ++i) t.set_len(i, 0);
            for (int i = 0; i < ncode; ++i) t.set_len(order[i], b.take(3));
            if (b.bad) return -1;
            Counts cc;
            if (construct<false>(t, 0, 19, &cc) != 0) return -6;                  // the code-length code must be complete (its symbols borrow the distance table's place)
            int idx = 0;
            while (idx < nlen + ndist) {
                b.fill();
                int sym = decode<false>(b, t, cc);
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
            int err = construct<true>(t, 0, nlen, &lc);
            if (err < 0 || (err > 0 && nlen - (int)lc.zeros != 1)) return -11;        // incomplete only for a single code
            err = construct<false>(t, nlen, ndist, &dc);
            if (err < 0 || (err > 0 && ndist - (int)dc.zeros != 1)) return -12;
        }
        for (;;) {                                         // every turn writes at least one byte or ends the block
            b.fill();
            int sym = decode<true>(b, t, lc);
            if (sym < 0) return -13;
            if (sym < 256) { if (!o.room(1)) return -3; o.lit((uint8_t)sym); continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return -14;
            const uint32_t len = lbase[sym] + b.take(lext[sym]);
            b.fill();
            const int ds = decode<false>(b, t, dc);
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
    PlainOut o{out, n_out, 0, 0, 0};
    return inflate_to(PlainIn{in}, n_in, o, t);
}

} // namespace lsi

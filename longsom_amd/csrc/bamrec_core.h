// One BAM record -> read-record arrays, written for the GPU (csrc/ingest.hip) and compiled for the host by tests/native/test_bamrec.cpp,
// which checks it record by record against the host decoder (hostio/bamio.cpp decode_record: the CB tag, the SplitBam counters, the
// htslib / pysam column semantics of SURVEY.md §8a rows a4-a6).  No allocation, every loop bounded by the record's own length fields,
// which validate() checks against the record's block_size before anything else walks them.
//
// Replaces, per record: read.opt("CB") + the barcode lookup + the MAPQ counters of split_bam (SplitBamCellTypes.py:65-124) and the
// CIGAR -> column step of bam.pileup (BaseCellCounter.py:191-216; htslib resolve_cigar2).
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __HIPCC__
#define LSR_FN __host__ __device__ __forceinline__
#else
#define LSR_FN inline
#endif

namespace lsr {

LSR_FN uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
LSR_FN uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
LSR_FN bool is_ref_op(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }

// symbol class of a 4-bit BAM base code: A C G T N -> 0 1 3 2 6, everything else ('=', IUPAC) -> NA (15)
LSR_FN uint32_t nt16_sym(uint32_t c) { return c == 1 ? 0u : c == 2 ? 1u : c == 4 ? 3u : c == 8 ? 2u : c == 15 ? 6u : 15u; }

enum { REC_OK = 0, REC_SHORT = 1, REC_FIELDS = 2, REC_TID = 3, REC_CIGAR_OP = 4, REC_QLEN = 5, REC_SPAN = 6 };

// rec points at refID (after block_size), len = block_size.  The checks of bamio.cpp's next_records, in its order.
LSR_FN int validate(const uint8_t* rec, uint32_t len, int32_t n_ref, const int64_t* ref_len) {
    if (len < 32) return REC_SHORT;
    const uint64_t l_name = rec[8], n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16);
    if (32 + l_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq > (uint64_t)len) return REC_FIELDS;
    const int32_t tid = (int32_t)rd32(rec), pos = (int32_t)rd32(rec + 4);
    if (tid >= n_ref) return REC_TID;
    const uint32_t flag = rd16(rec + 14);
    const uint8_t* cg = rec + 32 + l_name;
    uint64_t qlen = 0, rlen = 0;
    for (uint32_t k = 0; k < n_cigar; ++k) {
        const uint32_t c = rd32(cg + 4ull * k), op = c & 0xf, l = c >> 4;
        if (op > 8) return REC_CIGAR_OP;
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += l;
        if (is_ref_op(op)) rlen += l;
    }
    if (tid >= 0 && n_cigar && !(flag & 0x4)) {
        if (l_seq && qlen != l_seq) return REC_QLEN;
        if (pos < 0 || (uint64_t)pos + rlen > (uint64_t)ref_len[tid]) return REC_SPAN;
    }
    return REC_OK;
}

// CB:Z value of a (validated) record: *cb = offset of the value inside rec, *raw = its length, *clean = length up to the first '-'
// (barcode.split("-")[0], SplitBamCellTypes.py:83).  The LAST CB:Z tag wins, as in bamio.cpp's decode_record.
LSR_FN bool find_cb(const uint8_t* rec, uint32_t len, uint32_t* cb, uint32_t* raw, uint32_t* clean) {
    const uint32_t l_name = rec[8], n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16);
    uint64_t a = 32ull + l_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq;
    const uint64_t end = len;
    bool found = false;
    while (a + 3 <= end) {
        const uint8_t t0 = rec[a], t1 = rec[a + 1], ty = rec[a + 2];
        a += 3;
        uint64_t sz = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
        else if (ty == 's' || ty == 'S') sz = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
        else if (ty == 'Z' || ty == 'H') {
            uint64_t z = a;
            while (z < end && rec[z]) ++z;
            sz = (z - a) + 1;
            if (t0 == 'C' && t1 == 'B' && ty == 'Z' && z < end) { *cb = (uint32_t)a; *raw = (uint32_t)(z - a); found = true; }
        } else if (ty == 'B') {
            if (a + 5 > end) break;
            const uint8_t st = rec[a]; const uint64_t cnt = rd32(rec + a + 1);
            sz = 5 + cnt * ((st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : 4u);
        } else break;
        if (sz > end - a) break;                      // a field that claims more bytes than the record has
        a += sz;
    }
    if (found) { uint32_t c = 0; while (c < *raw && rec[*cb + c] != '-') ++c; *clean = c; }
    return found;
}

LSR_FN uint64_t fnv64(const uint8_t* p, uint32_t n) {
    uint64_t h = 1469598103934665603ull;
    for (uint32_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

// open-addressing table of the listed barcodes (built on the host, ingest.hip): slot -> {hash, dense id, string}
struct CbTable {
    const uint64_t* hash; const int32_t* id; const uint32_t* str_off; const uint32_t* str_len; const uint8_t* strs;
    uint32_t mask;                                    // slots - 1 (a power of two); id < 0 = empty slot
};
LSR_FN int32_t cb_lookup(const CbTable& t, const uint8_t* p, uint32_t n) {
    const uint64_t h = fnv64(p, n);
    uint32_t s = (uint32_t)h & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        const int32_t id = t.id[s];
        if (id < 0) return -1;
        if (t.hash[s] == h && t.str_len[s] == n) {
            const uint8_t* q = t.strs + t.str_off[s];
            bool eq = true;
            for (uint32_t i = 0; i < n; ++i) if (q[i] != p[i]) { eq = false; break; }
            if (eq) return id;
        }
        s = (s + 1) & t.mask;
    }
    return -1;
}

// indel flag of the LAST reference position of CIGAR operation k (peek at the next operations): 15 = none, 4 = I, 5 = D.
// legacy_del_merge: htslib <= 1.10 flags a D that follows a D too.
LSR_FN uint32_t indel_after(const uint8_t* cigar, uint32_t n_cigar, uint32_t k, uint32_t op, int legacy_del_merge) {
    if (k + 1 >= n_cigar) return 15;
    const uint32_t op2 = rd32(cigar + 4ull * (k + 1)) & 0xf;
    if (op2 == 2 && (op != 2 || legacy_del_merge)) return 5;
    if (op2 == 1) return 4;
    if (op2 == 6 && k + 2 < n_cigar) {
        uint32_t l3 = 0;
        for (uint32_t j = k + 2; j < n_cigar; ++j) {
            const uint32_t cj = rd32(cigar + 4ull * j), oj = cj & 0xf;
            if (oj == 1) l3 += cj >> 4;
            else if (oj == 2 || oj == 0 || oj == 3 || oj == 7 || oj == 8) break;
        }
        if (l3 > 0) return 4;
    }
    return 15;
}

struct Shape { uint32_t n_segs; uint64_t n_events; };

// The CIGAR walk of a kept record (htslib resolve_cigar2 semantics), operation by operation.  An operation's emitted
// positions are consecutive; a new segment starts where a position does not follow the last emitted one.
//   EMIT = false: only the shape (segments, events).
//   EMIT = true:  lane `lane` of `nlanes` writes the events i = lane, lane + nlanes, ... of every operation (coalesced across a wave);
//                 lane 0 writes the segments.  seg_* / events are the record's own places in the output arrays.
//   phased:       the record's events in the tile-phased layout (LSG_LAYOUT_PHASED, include/longsom_hip.h): ev_base is a multiple of 128, every
//                 segment lies at an offset congruent to its reference start modulo 128 (the gaps are the caller's to zero), the shape's
//                 n_events is the record's region: its last segment's end rounded up to 128.
template <bool EMIT>
LSR_FN Shape walk(const uint8_t* rec, int legacy_del_merge, uint32_t lane, uint32_t nlanes, uint32_t read_index,
                  uint32_t* seg_read, int32_t* seg_start, int32_t* seg_len, int64_t* seg_ev_off, int64_t ev_base, uint16_t* events, bool phased = false) {
    const int32_t pos = (int32_t)rd32(rec + 4);
    const uint32_t l_name = rec[8], n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16);
    const uint8_t* cigar = rec + 32 + l_name;
    const uint8_t* seq = cigar + 4ull * n_cigar;
    const uint8_t* qual = seq + (l_seq + 1) / 2;
    int64_t x = pos, last_pos = -2;
    uint32_t y = 0, n_segs = 0;
    uint64_t ne = 0;
    auto open = [&](int64_t p0, uint64_t c) {            // c consecutive positions from p0 are emitted next
        if (p0 != last_pos + 1 || n_segs == 0) {
            if (phased) ne += (uint64_t)((p0 - (int64_t)ne) & 127);
            if (EMIT && lane == 0) { seg_read[n_segs] = read_index; seg_start[n_segs] = (int32_t)p0; seg_len[n_segs] = 0; seg_ev_off[n_segs] = ev_base + (int64_t)ne; }
            ++n_segs;
        }
        if (EMIT && lane == 0) seg_len[n_segs - 1] += (int32_t)c;
        last_pos = p0 + (int64_t)c - 1;
    };
    for (uint32_t k = 0; k < n_cigar; ++k) {
        const uint32_t c = rd32(cigar + 4ull * k), op = c & 0xf, L = c >> 4;
        if (op == 1 || op == 4) { y += L; continue; }                       // I, S consume the query only
        if (!is_ref_op(op)) continue;                                       // H, P
        const uint32_t over = indel_after(cigar, n_cigar, k, op, legacy_del_merge);
        if (op == 0 || op == 7 || op == 8) {
            if (L) {
                open(x, L);
                if (EMIT)
                    for (uint32_t i = lane; i < L; i += nlanes) {
                        const uint32_t q = y + i;
                        uint32_t sym = q < l_seq ? nt16_sym((seq[q >> 1] >> ((~q & 1u) << 2)) & 0xfu) : 6u;      // beyond l_qseq pysam prints 'N'
                        if (i + 1 == L && over != 15) sym = over;
                        const uint32_t qv = q < l_seq ? qual[q] : 0u;
                        events[ev_base + (int64_t)ne + i] = (uint16_t)(sym < 8 ? (0x0800u | (sym << 8) | (qv & 0xffu)) : 0u);
                    }
                ne += L;
            }
            x += L; y += L;
        } else if (op == 2) {                                               // deletion: '*' -> O, quality of the next query base
            if (L) {
                open(x, L);
                if (EMIT) {
                    const uint32_t qv = y < l_seq ? qual[y] : 0u;
                    for (uint32_t i = lane; i < L; i += nlanes) {
                        const uint32_t sym = (i + 1 == L && over != 15) ? over : 7u;
                        events[ev_base + (int64_t)ne + i] = (uint16_t)(0x0800u | (sym << 8) | (qv & 0xffu));
                    }
                }
                ne += L;
            }
            x += L;
        } else {                                                            // N: '>' '<' are NA, except an indel flag on its last column
            if (L > 0 && over != 15) {
                open(x + L - 1, 1);
                if (EMIT && lane == 0) { const uint32_t qv = y < l_seq ? qual[y] : 0u; events[ev_base + (int64_t)ne] = (uint16_t)(0x0800u | (over << 8) | (qv & 0xffu)); }
                ne += 1;
            }
            x += L;
        }
    }
    return Shape{n_segs, phased ? (ne + 127u) & ~(uint64_t)127 : ne};
}

} // namespace lsr

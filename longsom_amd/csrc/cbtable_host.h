// Host side of bamrec_core.h's barcode table: the listed barcodes ('\n'-joined cleaned strings, ids[i] = dense id of barcode i or i)
// in an open-addressing table keyed by FNV-1a; duplicated strings: the last one wins (pandas to_dict, SplitBamCellTypes.py:31).
#pragma once
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "bamrec_core.h"

namespace lsr {
struct CbTableHost {
    std::vector<uint64_t> hash; std::vector<int32_t> id; std::vector<uint32_t> str_off, str_len; std::vector<uint8_t> strs;
    uint32_t mask = 0; int64_t n_tally = 0;
    void build(const char* barcodes, int32_t n_barcodes, const int32_t* ids) {
        std::unordered_map<std::string, int32_t> m;
        m.reserve((size_t)(n_barcodes > 0 ? n_barcodes : 0) * 2 + 16);
        const char* s = barcodes ? barcodes : "";
        for (int32_t i = 0; i < n_barcodes; ++i) {
            const char* e = strchr(s, '\n'); const size_t l = e ? (size_t)(e - s) : strlen(s);
            m[std::string(s, l)] = ids ? ids[i] : i;
            s += l + (e ? 1 : 0);
        }
        uint32_t slots = 16;
        while (slots < 2 * m.size() + 2) slots <<= 1;
        mask = slots - 1;
        hash.assign(slots, 0); id.assign(slots, -1); str_off.assign(slots, 0); str_len.assign(slots, 0); strs.clear();
        n_tally = 0;
        for (auto& kv : m) {
            const uint64_t h = fnv64((const uint8_t*)kv.first.data(), (uint32_t)kv.first.size());
            uint32_t q = (uint32_t)h & mask;
            while (id[q] >= 0) q = (q + 1) & mask;
            hash[q] = h; id[q] = kv.second; str_off[q] = (uint32_t)strs.size(); str_len[q] = (uint32_t)kv.first.size();
            strs.insert(strs.end(), kv.first.begin(), kv.first.end());
            if (kv.second + 1 > n_tally) n_tally = kv.second + 1;
        }
        strs.push_back(0);
    }
    CbTable view() const { return CbTable{hash.data(), id.data(), str_off.data(), str_len.data(), strs.data(), mask}; }
};
} // namespace lsr

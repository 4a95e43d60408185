// Compact call records of the step-1 call as the device holds them (call.hip writes them, tables.hip prints them; the C-ABI's lsg_call is
// their expansion).
#pragma once
#include <cstdint>
#include "../../include/longsom_hip.h"

namespace lsg {

struct SiteRec {                      // every merged site
    int64_t key;
    uint8_t ref, present, considered, has_cand;
    uint32_t site_filter;
    int32_t sum_alts_bc, sum_dp, sum_alts_cc, sum_nc;
    int16_t noise_p_bc, noise_p_cc;
    uint8_t cell_types_min, pad[3];
    uint32_t cand;                    // index of the candidate detail block
    uint32_t pad2;
};
static_assert(sizeof(SiteRec) == 48, "SiteRec layout");
struct CandCt {                       // one per (candidate site, cell type)
    uint8_t n_alt, ct_filter, alt[LSG_CALL_MAX_ALT], pad[2];
    uint32_t alt_bc[LSG_CALL_MAX_ALT], alt_cc[LSG_CALL_MAX_ALT];
    int16_t p_bc[LSG_CALL_MAX_ALT], p_cc[LSG_CALL_MAX_ALT];
};
static_assert(sizeof(CandCt) == 56, "CandCt layout");

} // namespace lsg

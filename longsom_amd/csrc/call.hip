// placeholder TU replaced below in this round (step-1 call kernel + position-set probe)
#include "lsg_ctx.h"
namespace lsg {
int run_call(lsg_ctx*, const lsg_call_params*) { set_error("lsg_call_step1: not built yet"); return -9; }
int run_fetch_calls(lsg_ctx*, lsg_call*, int64_t, int, int64_t*) { set_error("lsg_fetch_calls: not built yet"); return -9; }
int run_probe(lsg_ctx*, int, const int64_t*, int64_t, uint8_t*, int) { set_error("lsg_probe_posset: not built yet"); return -9; }
}

// C ABI, part 2: database pages, query batches and the search stages (being built).
#include "../../include/priblast_hip.h"
#include "context.hpp"

extern "C" {
int prb_search_const_upload(prb_ctx *) { return PRB_OK; }
void prb_search_const_free(prb_ctx *) {}
}

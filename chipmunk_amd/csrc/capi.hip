// Error plumbing and version of the C ABI (include/chipmunk_hip.h).
#include "common.h"

static thread_local char g_last_error[512] = "";

void chipmunk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *chipmunk_last_error(void) { return g_last_error; }
extern "C" int chipmunk_abi_version(void) { return 1; }

// ---- tuning knobs (kernel variant selection for A/B measurement; defaults are the shipped choices) ----
#include <string.h>
namespace {
struct Option { const char *name; int value; };
Option g_options[] = {{"mm1_variant", 0}, {"mm2_variant", 0}, {"attn_variant", 0}, {"m2i_variant", 0}, {"topk_variant", 0}, {"mm1_nr", 0}, {"mm2_nr", 0}, {"mm1_probe", 0}, {"colsum_fused", 0}, {"attn_xcd_chunks", 0}, {"attn_no_order", 0}, {"mm1_no_split", 0}, {"mm2_no_split", 0}};
}
int chipmunk_get_option(const char *name) {
    for (auto &o : g_options) if (strcmp(o.name, name) == 0) return o.value;
    return 0;
}
extern "C" int chipmunk_set_option(const char *name, int value) {
    for (auto &o : g_options) if (strcmp(o.name, name) == 0) { o.value = value; return CHIPMUNK_OK; }
    chipmunk_set_error("unknown option '%s'", name);
    return CHIPMUNK_ERR_INVALID;
}

// Runtime support of the spn C-ABI: thread-local error string (no C++ exceptions cross the boundary), ABI version, tuning knobs.
#include <string.h>
#include <atomic>
#include "tuning.h"

static thread_local char g_err[512] = "";
extern "C" void spn_set_error(const char* msg) { strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* spn_last_error(void) { return g_err; }
// 2: caller-owned workspaces (spn_gemm_workspace_bytes, band buffers), spn_set_tuning instead of environment reads, trainable mask in spn_adamw_step
// 3: spn_dec_head / spn_dec_head_sample take the next step's position scalar (pos_next); spn_dec_step_begin, spn_dec_cat_gemv
extern "C" int spn_abi_version(void) { return 10; }

namespace {
struct Knob { const char* name; double def; };
const Knob kKnobs[SPN_TUNE_COUNT] = {
    {"attn_band", 30.0},         {"attn_order", 1.0},          {"gemm_variant", 0.0},         {"gemm_ngroup", 8.0},
    {"gemm_slice_xcd", 1.0},     {"gemm_split_blocks", 0.0},   {"gemm_persist", 2.0},         {"glu_persist", 2.0},
    {"embed_stats_blocks", 2048.0}, {"embed_scatter_mfma", 1.0}, {"embed_scatter_blocks", 256.0}, {"ln_bwd_blocks", 2048.0},
    {"gemm_duo", 1.0},           {"gemm_duo_ngroup", 8.0},      {"gemm_stagger", 0.0},        {"glu_bwd_duo", 2.0},
    {"gemm_persist_bwd", 0.0},   {"gemm_f32_mfma", 1.0},
};
std::atomic<double> g_val[SPN_TUNE_COUNT];
std::atomic<bool> g_set[SPN_TUNE_COUNT];
int find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < SPN_TUNE_COUNT; ++i)
        if (strcmp(name, kKnobs[i].name) == 0) return i;
    return -1;
}
}  // namespace

double spn_tune(SpnTune k) {
    return g_set[k].load(std::memory_order_acquire) ? g_val[k].load(std::memory_order_relaxed) : kKnobs[k].def;
}

extern "C" int spn_set_tuning(const char* name, double value) {
    const int i = find(name);
    if (i < 0) { spn_set_error("spn_set_tuning: unknown knob"); return -1; }
    g_val[i].store(value, std::memory_order_relaxed);
    g_set[i].store(true, std::memory_order_release);
    return 0;
}

extern "C" int spn_get_tuning(const char* name, double* value) {
    const int i = find(name);
    if (i < 0 || !value) { spn_set_error("spn_get_tuning: unknown knob"); return -1; }
    *value = spn_tune((SpnTune)i);
    return 0;
}

extern "C" int spn_tuning_count(void) { return SPN_TUNE_COUNT; }
extern "C" const char* spn_tuning_name(int i) { return (i >= 0 && i < SPN_TUNE_COUNT) ? kKnobs[i].name : nullptr; }

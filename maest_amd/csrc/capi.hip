// Error plumbing and version of the C ABI (include/maest_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace maest {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return MAEST_ERR_LAUNCH;
    }
    return MAEST_OK;
}

// ---- process-wide switches: environment read ONCE (first use), then lock-free atomics
static constexpr int kNumOptions = 11;
static const int kOptionDefault[kNumOptions] = {8192, 0, -1, 0, 1024, 1, 0, 0, 0, 0, 0};
static const char* const kOptionEnv[kNumOptions] = {"MAEST_GEMM_MIN_M", "MAEST_GEMM_VARIANT", "MAEST_GEMM_EPILOGUE",
                                                  "MAEST_ATTN_BWD", "MAEST_LN_BWD_BLOCKS", "MAEST_GEMM_TAIL",
                                                  "MAEST_ATTN_FWD", "MAEST_ATTN_FWD_WAVES",
                                                  "MAEST_TN_REDUCE", "MAEST_GEMM_WGS", "MAEST_GEMM_PANEL"};
static std::atomic<int> g_option[kNumOptions];
static int g_option_env[kNumOptions];
static std::once_flag g_option_once;

static void options_init() {
    std::call_once(g_option_once, [] {
        for (int i = 0; i < kNumOptions; ++i) {
            const char* e = getenv(kOptionEnv[i]);
            g_option_env[i] = e ? atoi(e) : kOptionDefault[i];
            g_option[i].store(g_option_env[i], std::memory_order_relaxed);
        }
    });
}

// per-thread overrides (maest_set_option_thread): a pass that wants its own launch form sets them on the thread that launches its
// kernels -- two engines on two threads of one process then never see each other's choice
static thread_local int t_option[kNumOptions];
static thread_local unsigned t_option_mask = 0;

int option(int opt) {
    if ((t_option_mask >> opt) & 1u) return t_option[opt];
    options_init();
    return g_option[opt].load(std::memory_order_relaxed);
}

}  // namespace maest

extern "C" int maest_set_option(int opt, int value, int restore_default) {
    MAEST_REQUIRE(opt >= 0 && opt < maest::kNumOptions, "maest_set_option: unknown option %d", opt);
    maest::options_init();
    maest::g_option[opt].store(restore_default ? maest::g_option_env[opt] : value, std::memory_order_relaxed);
    return MAEST_OK;
}
extern "C" int maest_set_option_thread(int opt, int value, int clear) {
    MAEST_REQUIRE(opt >= 0 && opt < maest::kNumOptions, "maest_set_option_thread: unknown option %d", opt);
    if (clear) maest::t_option_mask &= ~(1u << opt);
    else {
        maest::t_option[opt] = value;
        maest::t_option_mask |= 1u << opt;
    }
    return MAEST_OK;
}
extern "C" int maest_get_option(int opt, int* value) {
    MAEST_REQUIRE(opt >= 0 && opt < maest::kNumOptions && value, "maest_get_option: unknown option %d", opt);
    *value = maest::option(opt);
    return MAEST_OK;
}

namespace maest {
bool gemm_nt256o_available();   // gemm_nt_ow.hip
bool gemm_tn256o_available();   // gemm_tn_ow.hip
bool attn_fwd_pw_available();   // attn_fwd_pw.hip
}  // namespace maest
extern "C" int maest_kernel_forms(int* mask) {
    MAEST_REQUIRE(mask, "maest_kernel_forms: null pointer");
    *mask = (maest::gemm_nt256o_available() ? MAEST_FORM_GEMM_NT_OW : 0) | (maest::gemm_tn256o_available() ? MAEST_FORM_GEMM_TN_OW : 0) |
            (maest::attn_fwd_pw_available() ? MAEST_FORM_ATTN_FWD_PW : 0);
    return MAEST_OK;
}

extern "C" int maest_version(void) { return MAEST_ABI_VERSION; }
extern "C" const char* maest_last_error(void) { return maest::g_error; }

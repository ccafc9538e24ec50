// core.hip - error reporting + version for libdpot_hip.so
#include "common.h"

namespace dpot {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(err));
    return DPOT_EHIP;
  }
  return DPOT_OK;
}

// DPOT_TUNE="key=val,key=val" (integers; unknown keys are ignored here - dpot_amd/ops.py rejects them): parsed on first use
int tune(const char* key, int dflt) {
  static const char* env = getenv("DPOT_TUNE");
  if (!env || !*env) return dflt;
  const size_t kl = strlen(key);
  for (const char* p = env; *p;) {
    const char* end = strchr(p, ',');
    const size_t len = end ? (size_t)(end - p) : strlen(p);
    if (len > kl + 1 && strncmp(p, key, kl) == 0 && p[kl] == '=') return atoi(p + kl + 1);
    if (!end) break;
    p = end + 1;
  }
  return dflt;
}

}  // namespace dpot

/* 0.2.6: round 6 ABI - dpot_adam_step_packs (Adam that writes the bf16 weight packs); dpot_gemm_bf16p_pair back to its 0.2.0
 * argument list (the row-form operand / transposed-output arguments of 0.2.5 are gone with the kernels they selected);
 * dpot_afno_fused_bwd removed; every fallback selector behind DPOT_TUNE */
extern "C" int dpot_version(void) { return 261; }
extern "C" int dpot_tune(const char* key, int dflt) { return dpot::tune(key, dflt); }
extern "C" const char* dpot_last_error(void) { return dpot::g_err; }

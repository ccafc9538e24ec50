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

}  // namespace dpot

extern "C" int dpot_version(void) { return 250; /* 0.2.5: round 5 ABI (one-launch AFNO layer fwd / bwd, row-form pair launch, gradient packs from the GroupNorm backward, kernel-kind query) */ }
extern "C" const char* dpot_last_error(void) { return dpot::g_err; }

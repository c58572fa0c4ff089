// Error reporting / version entry points of libwslhip.so.
#include <stdarg.h>
#include <stdio.h>

#include "wsl_rt.h"

namespace wsl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return WSL_EHIP;
  }
  return WSL_OK;
}

}  // namespace wsl

extern "C" int wsl_version(void) { return 100; }
extern "C" const char* wsl_last_error(void) { return wsl::g_err; }
extern "C" const char* wsl_build_info(void) {
#ifdef WSL_HOST_EMUL
  return "HOST-EMULATION (tests only; not a product build)";
#else
  return "gfx950 hipcc";
#endif
}

// api.hip -- version / error reporting of libmgs.so (the stage entry points live next to
// their kernels: projection.hip, binning.hip, raster_fwd.hip, raster_bwd.hip, backward.hip).
#include "mgs_common.h"

namespace mgs {

char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace mgs

extern "C" int mgs_version(void) { return MGS_VERSION; }
extern "C" const char* mgs_last_error_string(void) { return mgs::error_buffer(); }

// placeholder until the raster backward lands (returns MGS_ERR_UNSUPPORTED)
#include "mgs_common.h"
using namespace mgs;
extern "C" int mgs_rasterize_bwd(int, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, const int32_t*, const int32_t*, const float*, const int32_t*, const float*, const float*, float*, float*, float*, float*, float*, mgs_stream_t) { return set_error(MGS_ERR_UNSUPPORTED, "rasterize_bwd: not built yet"); }

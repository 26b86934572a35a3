// placeholder until the projection / colour backward lands (returns MGS_ERR_UNSUPPORTED)
#include "mgs_common.h"
using namespace mgs;
extern "C" int mgs_projection_bwd(int, const float*, const float*, const float*, const float*, const float*, int, int, float, const int32_t*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, mgs_stream_t) { return set_error(MGS_ERR_UNSUPPORTED, "projection_bwd: not built yet"); }
extern "C" int mgs_sh_bwd(int, int, int, const float*, const float*, const uint8_t*, const float*, float*, float*, mgs_stream_t) { return set_error(MGS_ERR_UNSUPPORTED, "sh_bwd: not built yet"); }
extern "C" int mgs_project_color_bwd(int, const float*, const float*, const float*, const float*, int, int, const float*, const float*, const float*, int, int, float, const int32_t*, const float*, int, int, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*, mgs_stream_t) { return set_error(MGS_ERR_UNSUPPORTED, "project_color_bwd: not built yet"); }

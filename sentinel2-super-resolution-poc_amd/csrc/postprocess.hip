// placeholder until the CLAHE / unsharp / vegetation kernels land (next milestone)
#include "s2sr_internal.h"
namespace s2sr {
size_t postprocess_work_bytes(int, int, int, const s2sr_pp_params&) { return 256; }
hipError_t launch_postprocess(const uint8_t*, int, int, int, const s2sr_pp_params&, uint8_t*, void*, size_t, hipStream_t) {
    return hipErrorNotSupported;
}
}  // namespace s2sr

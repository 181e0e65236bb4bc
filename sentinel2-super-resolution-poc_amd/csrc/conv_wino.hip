// placeholder, filled in below
#include "s2sr_internal.h"
namespace s2sr {
hipError_t launch_conv_trunk_wino(const ConvParams&, hipStream_t) { return hipErrorNotSupported; }
size_t conv_wpack_bytes_wino(int cin, int cout) { return (size_t)((cin + 15) / 16) * 12 * ((cout + 31) / 32) * 1024; }
hipError_t launch_pack_trunk_wino(const float*, int, int, void*, hipStream_t) { return hipErrorNotSupported; }
}  // namespace s2sr

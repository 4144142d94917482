// va_gauss_fused.hip -- placeholder until the LDS-staged kernel lands
#include "va_common.h"
namespace va {
bool gauss_fused_supported(int, int, const TapsQ8 &) { return false; }
int launch_gauss_fused_u8(const uint8_t *, uint8_t *, uint32_t *, int, int, int, int,
                          const TapsQ8 &, hipStream_t)
{
    set_error("fused Gaussian not built");
    return VA_ERR_INVALID;
}
}  // namespace va

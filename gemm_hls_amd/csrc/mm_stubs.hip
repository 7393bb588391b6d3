// Kernel families not built yet report "not supported"; the dispatcher then serves the request
// with another GPU family (never the CPU).  Entries disappear from here as families land.
#include "mm_common.h"
namespace mm {
bool mfma_f16_serves(const Problem &) { return false; }
int launch_mfma_f16(hipStream_t, const Problem &) { return kErrNotSupported; }
}  // namespace mm

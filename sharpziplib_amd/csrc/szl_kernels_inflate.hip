// szl_kernels_inflate.hip — placeholder translation unit (device inflate kernels land here).
#include <hip/hip_runtime.h>
#include "szl_internal.h"
namespace szl { }

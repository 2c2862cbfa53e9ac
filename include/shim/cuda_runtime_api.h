/* cuda_runtime_api.h -- see cuda_runtime.h in this directory */
#include "cuda_runtime.h"

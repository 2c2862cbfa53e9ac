/* cuda.h -- see cuda_runtime.h in this directory */
#include "cuda_runtime.h"

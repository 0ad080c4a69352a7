// include-path shim used ONLY by oracle/build_ref.py: lets hipcc compile the reference's .cu
// sources in place (no hipify, no copy) by mapping the three CUDA header names they include onto HIP.
#pragma once
#include <hip/hip_runtime.h>

// shim for oracle/build_ref.py: the ROCm build of torch ships the HIP context under ATen/hip/.
#pragma once
#include <ATen/hip/HIPContext.h>

#pragma once
#include <hip/hip_runtime_api.h>

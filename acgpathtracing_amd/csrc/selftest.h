// selftest.h — see selftest.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {
hipError_t launch_selftest(int op, const uint32_t* d_in, uint32_t n, uint32_t* d_out, hipStream_t stream);
}

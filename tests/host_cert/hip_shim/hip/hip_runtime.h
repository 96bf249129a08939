// Host shim of <hip/hip_runtime.h> for tests/host_cert/host_cert.cpp (TEST INFRASTRUCTURE): just enough for the LANE-LEVEL
// certificate code of kernel family 3 (toppra_amd/csrc/tpr_cert_lane.hip.inc + tpr_device.hpp) to compile with g++ as plain
// C++ -- qualifiers become nothing, the handful of device intrinsics that code uses get host definitions.  -ffp-contract=off
// keeps the arithmetic as separately rounded as the device build's.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(x)
using std::fabs; using std::fmax; using std::fmin; using std::isnan; using std::sqrt;
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
static inline int __double2hiint(double d) { return (int)(__double_as_longlong(d) >> 32); }
static inline int __double2loint(double d) { return (int)(__double_as_longlong(d) & 0xffffffffll); }
static inline double __hiloint2double(int hi, int lo) {
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

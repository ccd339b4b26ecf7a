// Device-side helpers shared by the gfx950 kernels of libwfl.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "common.h"

#include <map>
#include <mutex>
#include <utility>

#define WFL_NEG_INF (-__builtin_inff())

#define WFL_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      wfl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return WFL_ERR_RUNTIME;                                                            \
    }                                                                                    \
  } while (0)

namespace wfl {
// Flags of the events that only order one stream of this device behind another (fork / join of the side streams).
// Nothing on the host or on another device ever inspects them, so they need neither a timestamp nor the SYSTEM-scope
// fence a default event performs when it is recorded (a write-back of every XCD's L2 to memory: measured, same box,
// alternating: Transducer benchmark step 0.358 -> 0.350 ms without it).  What they do need is that the side stream's
// results are visible to the kernels the other stream launches behind the join: a DEVICE-scope release, which is what
// hipEventReleaseToDevice asks for (hip_runtime_api.h: "Use a device-scope release when recording this event").
// Round 5 used hipEventDisableSystemFence instead -- documented for timing-only events, with visibility resting on each
// kernel packet's own release; WFL_ORDER_EVENTS=nofence selects that again for A/B.
inline unsigned order_event_flags_from_env() {
  const char* v = getenv("WFL_ORDER_EVENTS");
  if (v && !strcmp(v, "nofence")) return hipEventDisableTiming | hipEventDisableSystemFence;
  if (v && !strcmp(v, "default")) return hipEventDisableTiming;
  return hipEventDisableTiming | hipEventReleaseToDevice;
}
inline unsigned order_event_flags() {
  static const unsigned flags = order_event_flags_from_env();
  return flags;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) instead of once per launch: the
// call costs the host a microsecond or two, and the CTC operator is host-bound at B = 128
inline hipError_t set_max_dynamic_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  int& have = done[{dev, kern}];
  if (have >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}
}  // namespace wfl

#define WFL_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      wfl::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return WFL_ERR_RUNTIME;                                                      \
    }                                                                              \
  } while (0)

namespace wfl {

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kLdsBytes = 160 * 1024;  // per CU (and per workgroup) on MI355X

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
// every call site passes a sum of exponentials relative to their maximum (>= 1): no denormal
// inputs, so the bare v_log_f32 (log2) is enough
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718f; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL
// stores (s_waitcnt vmcnt(0)): in a per-frame loop that streams its results to HBM that is one store round trip
// (~1 us) per frame on the dependent path.  Use where the waves only exchange data through LDS.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NaN policy: a NaN score is an impossible arc.
__device__ __forceinline__ float nan_to_neg(float v) { return (v != v) ? WFL_NEG_INF : v; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- DPP wave reductions (gfx9 row_shr / row_bcast): 6 VALU instructions, no LDS, result valid in
// lane 63 only; wave_all_* broadcast it through an SGPR (v_readlane).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float identity, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// bare v_max_f32: fmaxf() also emits a canonicalising v_max x,x,x per operand, which matters on
// latency-bound chains.  NaN operands: returns the other operand (IEEE mode), like fmaxf.
__device__ __forceinline__ float vmax(float a, float b) {
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ float wave_shr1(float v, float fill) {
  // lane i receives lane i-1's value; lane 0 receives `fill` (DPP wave_shr:1, bound_ctrl off)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v, float fill) {
  // lane i receives lane i+1's value; lane 63 receives `fill` (DPP wave_shl:1)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_reduce_max_lane63(float v) {
  v = vmax(v, dpp_f32<0x111, 0xf>(WFL_NEG_INF, v));  // row_shr:1
  v = vmax(v, dpp_f32<0x112, 0xf>(WFL_NEG_INF, v));  // row_shr:2
  v = vmax(v, dpp_f32<0x114, 0xf>(WFL_NEG_INF, v));  // row_shr:4
  v = vmax(v, dpp_f32<0x118, 0xf>(WFL_NEG_INF, v));  // row_shr:8
  v = vmax(v, dpp_f32<0x142, 0xa>(WFL_NEG_INF, v));  // row_bcast:15 -> rows 1,3
  v = vmax(v, dpp_f32<0x143, 0xc>(WFL_NEG_INF, v));  // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ float wave_reduce_sum_lane63(float v) {
  v += dpp_f32<0x111, 0xf>(0.f, v);
  v += dpp_f32<0x112, 0xf>(0.f, v);
  v += dpp_f32<0x114, 0xf>(0.f, v);
  v += dpp_f32<0x118, 0xf>(0.f, v);
  v += dpp_f32<0x142, 0xa>(0.f, v);
  v += dpp_f32<0x143, 0xc>(0.f, v);
  return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
// the same reduction for ints (exponents of the probability-domain renormalisations): every lane gets the maximum
__device__ __forceinline__ int wave_all_max_int(int v) {
  constexpr int kMin = -2147483647 - 1;
  v = max(v, dpp_i32<0x111, 0xf>(kMin, v));
  v = max(v, dpp_i32<0x112, 0xf>(kMin, v));
  v = max(v, dpp_i32<0x114, 0xf>(kMin, v));
  v = max(v, dpp_i32<0x118, 0xf>(kMin, v));
  v = max(v, dpp_i32<0x142, 0xa>(kMin, v));
  v = max(v, dpp_i32<0x143, 0xc>(kMin, v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_all_max(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_reduce_max_lane63(v)), 63));
}
__device__ __forceinline__ float wave_all_sum(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_reduce_sum_lane63(v)), 63));
}

// log of the log-add's sum: s >= 1 (a sum that contains its largest term, 1), finite.  The library's double-precision
// log was 27 % of the general sweep (100 of 374 us on the bigram Transducer's epsilon numerator, measured with a float
// log in its place -- which costs the digits the double state vectors exist for).  Here: s = m 2^e with m in
// [sqrt(1/2), sqrt(2)), log m = 2 atanh z with z = (m - 1) / (m + 1), |z| <= 0.1716, nine terms of the series: within
// 1.8e-15 of the library's value for 1 <= s <= 1e6 (checked on the CPU: tests/test_host_library.py restates it).
__device__ __forceinline__ double lse_log(double s) {
  double m = __builtin_amdgcn_frexp_mant(s);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(s);
  if (m < 0.70710678118654752) m *= 2.0, e -= 1;
  const double z = (m - 1.0) / (m + 1.0), z2 = z * z;
  double p = 1.0 / 17.0;
  p = fma(p, z2, 1.0 / 15.0);
  p = fma(p, z2, 1.0 / 13.0);
  p = fma(p, z2, 1.0 / 11.0);
  p = fma(p, z2, 1.0 / 9.0);
  p = fma(p, z2, 1.0 / 7.0);
  p = fma(p, z2, 1.0 / 5.0);
  p = fma(p, z2, 1.0 / 3.0);
  p = fma(p, z2, 1.0);
  return fma((double)e, 0.69314718055994531, 2.0 * z * p);
}

// log(exp(a) + exp(b)) with -inf handled
__device__ __forceinline__ float log_add(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == WFL_NEG_INF) return WFL_NEG_INF;
  return m + fast_log(fast_exp(a - m) + fast_exp(b - m));
}

}  // namespace wfl

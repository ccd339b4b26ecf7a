// Second latency micro-benchmark (round 2): unrolled dependent chains (no loop overhead) of the operations the
// probability-domain sweeps are made of, and replicas of their frame loops with pieces knocked out.
//   hipcc --offload-arch=gfx950 -O3 scratch/lat_ubench2.hip -o gpurun_out/lat_ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
#define REP16(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP64(...) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__)

template <int KIND>
__global__ void k_chain(long long* out, double* sink) {
  float a = threadIdx.x * 1e-3f + 0.5f;
  double d = threadIdx.x * 1e-3 + 0.5, e = 1.0000001;
  const long long c0 = clock64();
  for (int it = 0; it < 100; ++it) {
    if (KIND == 0) { REP64(a = fmaf(a, 1.0000001f, 0.25f);) }
    if (KIND == 1) { REP64(d = fma(d, e, 0.25);) }
    if (KIND == 2) { REP64(d = d * e;) }
    if (KIND == 3) { REP64(d = (double)a * e; a = (float)d;) }
    if (KIND == 4) { REP64(a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xf, 0xf, false)) + 1.f;) }
    if (KIND == 5) {  // the banded frame's dependent chain: two DPP moves of the halves, mul, fma
      REP64({ const int lo = __double2loint(d), hi = __double2hiint(d);
              const double pn = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false),
                                                 __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false));
              d = fma(d, e, pn * 0.5); })
    }
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0) out[0] = c1 - c0;
  sink[threadIdx.x] = d + a;
}

// replica of the banded frame loop: STORE: 0 none, 1 the state vector (8 B/lane) every frame, 2 + lane 0's offset
template <int STORE>
__global__ void k_banded(long long* out, double* vec, double* offs, int T) {
  double p = threadIdx.x == 0 ? 1.0 : 0.0, cum = 0.0;
  const double cs = 0.7, ca = 0.6;
  const long long c0 = clock64();
  for (int t = 0; t < T; t += 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lo = __double2loint(p), hi = __double2hiint(p);
      const double pn = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false),
                                         __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false));
      p = fma(p, cs, pn * ca);
      cum += 0.123 * 1.4426950408889634;
      if (STORE >= 1 && threadIdx.x < 45) vec[(size_t)blockIdx.x * (T + 1) * 45 + (size_t)(t + i + 1) * 45 + threadIdx.x] = p;
      if (STORE >= 2 && threadIdx.x == 0) offs[(size_t)blockIdx.x * (T + 1) + t + i + 1] = cum;
    }
    // renormalise like the chunk boundary does
    const int ex = p > 0.0 ? ilogb(p) : -(1 << 30);
    int m = ex;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if (m > -(1 << 30)) p = scalbn(p, -m), cum += m;
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
  if (p == 123.0) vec[0] = p + cum;
}

int main() {
  long long* out; double *sink, *vec, *offs;
  const int T = 1024, NB = 256;
  CK(hipMalloc(&out, 8 * NB)); CK(hipMalloc(&sink, 8 * 1024));
  CK(hipMalloc(&vec, (size_t)8 * NB * (T + 1) * 45)); CK(hipMalloc(&offs, (size_t)8 * NB * (T + 1)));
  long long h[NB];
  const char* names[] = {"v_fma_f32 (dependent)", "v_fma_f64 (dependent)", "v_mul_f64 (dependent)", "cvt f32->f64, mul, cvt f64->f32",
                         "v_mov_dpp wave_shr + add", "banded frame chain: 2 DPP + mul_f64 + fma_f64"};
#define RUN(K) k_chain<K><<<1, 64>>>(out, sink); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost)); \
  printf("  %-48s %.1f cycles per op group\n", names[K], (double)h[0] / 6400);
  printf("lone wave, unrolled dependent chains:\n");
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  for (int nb : {1, 256}) {
    printf("banded frame-loop replica, %d workgroup(s) of one wave, T = %d:\n", nb, T);
    k_banded<0><<<nb, 64>>>(out, vec, offs, T); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    printf("  no stores                      %.1f cycles per frame\n", (double)h[0] / T);
    k_banded<1><<<nb, 64>>>(out, vec, offs, T); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    printf("  + state vector store per frame %.1f cycles per frame\n", (double)h[0] / T);
    k_banded<2><<<nb, 64>>>(out, vec, offs, T); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    printf("  + lane-0 offset store          %.1f cycles per frame\n", (double)h[0] / T);
  }
  return 0;
}

// Replica of the dense sweep's per-frame loop (dense_kernels.hip, chain_step) with its pieces switched on one at a
// time: where do ~950 cycles per frame go?   hipcc --offload-arch=gfx950 -O3 scratch/lat_ubench3.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// bits: 1 LDS read of the chunk registers, 2 the 64 DPP multiply-adds, 4 cross-lane add, 8 LDS write of the result,
//       16 global store, 32 scale (frexp / ldexp of a second LDS read), 64 helper wave does work (wave max + 2 exp2 + LDS writes)
template <int M>
__global__ void __launch_bounds__(320) k_dense(long long* out, float* ob, int T) {
  __shared__ __attribute__((aligned(16))) float vecT[2][2][16][4];
  __shared__ float eh[4][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qq = (lane >> 4) & 1, il = lane & 15, q = 32 * wave + 16 * (lane >> 5) + il;
  const bool owner = qq == 0;
  float P[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) P[k] = 1.0f / (float)(k + 1 + lane);
  for (int i = tid; i < 256; i += 320) (&vecT[0][0][0][0])[i] = 1.f;
  for (int i = tid; i < 512; i += 320) (&eh[0][0])[i] = 0.9f;
  __syncthreads();
  float last = 0.f;
  const long long c0 = clock64();
  if (wave < 4) {
    for (int n = 1; n < T; ++n) {
      const int cur = (n - 1) & 1;
      float4 vc4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (M & 1) vc4 = *reinterpret_cast<const float4*>(vecT[cur][qq][il]);
      float inv = 1.f;
      if (M & 32) {
        const float4 s4 = *reinterpret_cast<const float4*>(vecT[cur][0][0]);
        inv = __builtin_amdgcn_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(fmaxf(fmaxf(s4.x, s4.y), fmaxf(s4.z, s4.w))));
      }
      const float vc[4] = {vc4.x, vc4.y, vc4.z, vc4.w};
#ifndef NACC
#define NACC 4
#endif
      float acc[NACC] = {};
      if (M & 2) {
        asm volatile("s_nop 1" ::: "memory");
#define BC(c, k) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc[(k) & (NACC - 1)]) : "v"(vc[c]), "v"(P[16 * (c) + (k)]));
#define BC16(c) BC(c,0) BC(c,1) BC(c,2) BC(c,3) BC(c,4) BC(c,5) BC(c,6) BC(c,7) BC(c,8) BC(c,9) BC(c,10) BC(c,11) BC(c,12) BC(c,13) BC(c,14) BC(c,15)
#ifdef NFM52
        BC16(0) BC16(1) BC16(2) BC(3,0) BC(3,1) BC(3,2) BC(3,3)
#else
        BC16(0) BC16(1) BC16(2) BC16(3)
#endif
      } else {
        acc[0] = vc[0] * P[0];
      }
      float part = 0.f;
#pragma unroll
      for (int z = 0; z < NACC; ++z) part += acc[z];
      if (M & 4) part += __shfl_xor(part, 16, 64);
      const float val = eh[n & 3][q & 127] * inv * part + 1e-3f;
      if ((M & 8) && owner) vecT[cur ^ 1][q >> 6][q & 15][(q >> 4) & 3] = val;
      last = val;
      if ((M & 16) && owner && q < 100) ob[(size_t)blockIdx.x * T * 100 + (size_t)n * 100 + q] = val;
      lds_barrier();
    }
  } else {
    for (int n = 1; n < T; ++n) {
      if (M & 64) {
        float s0 = last + lane * 1e-3f, s1 = s0 * 0.5f;
        float m = fmaxf(s0, s1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        eh[(n + 2) & 3][lane] = __builtin_amdgcn_exp2f(s0 - m) * 0.9f + 0.05f;
        eh[(n + 2) & 3][lane + 64] = __builtin_amdgcn_exp2f(s1 - m) * 0.9f + 0.05f;
        last = m * 1e-6f;
      }
      lds_barrier();
    }
  }
  const long long c1 = clock64();
  if (tid == 0) out[blockIdx.x] = c1 - c0;
  if (last == 123.f) ob[0] = last;
}

int main() {
  long long* out; float* ob;
  const int T = 1000, NB = 256;
  CK(hipMalloc(&out, 8 * NB)); CK(hipMalloc(&ob, (size_t)4 * NB * T * 100));
  long long h[NB];
#define RUN(M, what) k_dense<M><<<NB, 320>>>(out, ob, T); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost)); \
  printf("  mask %3d  %-58s %.1f cycles per frame\n", M, what, (double)h[0] / (T - 1));
  printf("dense frame-loop replica, 256 workgroups x 5 waves, T = %d:\n", T);
  RUN(0, "barrier only")
  RUN(1, "+ chunk LDS read")
  RUN(2, "DPP multiply-adds only")
  RUN(3, "read + DPP multiply-adds")
  RUN(7, "+ cross-lane add (ds_bpermute)")
  RUN(15, "+ LDS write of the new vector")
  RUN(47, "+ scale (second LDS read, frexp, ldexp)")
  RUN(63, "+ global store per frame")
  RUN(127, "+ helper wave working")
  return 0;
}

// Latency / clock micro-benchmark for the cost model of DESIGN.md (round 2): shader clock vs the 100 MHz wall clock,
// dependent-issue cost of the instruction kinds the sweeps are made of, for a lone wave and for a 5-wave workgroup
// with one barrier per iteration.   hipcc --offload-arch=gfx950 -O3 scratch/lat_ubench.hip -o gpurun_out/lat_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_clock(long long* out) {
  const long long w0 = wall_clock64(), c0 = clock64();
  float v = threadIdx.x;
  for (int i = 0; i < 200000; ++i) v = fmaf(v, 1.000001f, 0.5f);
  const long long w1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0) out[0] = w1 - w0, out[1] = c1 - c0;
  if (v == 12345.f) out[2] = 1;
}
template <int KIND>
__global__ void k_dep(long long* out, float* sink, int iters) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = threadIdx.x * 1e-3f, b = 1.0000001f;
  double d = threadIdx.x * 1e-3, e = 1.0000001;
  int idx = threadIdx.x;
  const long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { a = fmaf(a, b, 0.25f); }                                   // dependent v_fma_f32
    if (KIND == 1) { d = fma(d, e, 0.25); }                                     // dependent v_fma_f64
    if (KIND == 2) { asm volatile("v_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); }
    if (KIND == 3) { a = lds[(int)a & 1023] + 1.f; }                            // dependent ds_read_b32
    if (KIND == 4) { a = __shfl_xor(a, 16, 64) + 1.f; }                         // dependent ds_bpermute
    if (KIND == 5) { lds[threadIdx.x] = a; __syncthreads(); a = lds[(threadIdx.x + 64) & 255] + 1.f; }  // write, barrier, read
    if (KIND == 6) { d = (double)a * e; a = (float)d + 1.f; }                   // cvt f32->f64, mul, cvt back
    if (KIND == 7) { a = __builtin_amdgcn_exp2f(a) * 0.5f; }                    // v_exp_f32 + mul
    if (KIND == 8) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); a += 1.f; }  // bare barrier
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0) out[0] = c1 - c0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a + (float)d + idx;
}
int main() {
  long long* out; float* sink;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 1 << 20));
  long long h[4];
  k_clock<<<1, 64>>>(out); CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
  printf("lone wave: %lld wall ticks (100 MHz) = %.1f us, %lld shader cycles -> clock64 runs at %.0f MHz; 200000 dependent fma = %.2f cycles each\n",
         h[0], h[0] / 100.0, h[1], h[1] / (h[0] / 100.0), (double)h[1] / 200000);
  k_clock<<<256 * 8, 256>>>(out); CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
  printf("busy chip: %.1f us, clock64 at %.0f MHz\n", h[0] / 100.0, h[1] / (h[0] / 100.0));
  const char* names[] = {"v_fma_f32", "v_fma_f64", "v_fmac_f32_dpp", "ds_read_b32 (dependent)", "ds_bpermute (shfl_xor)",
                         "ds_write + __syncthreads + ds_read", "cvt f64 mul cvt", "v_exp_f32 + mul", "s_barrier alone"};
  const int iters = 20000;
  for (int threads : {64, 320}) {
    printf("-- %d threads per workgroup, 1 workgroup\n", threads);
#define RUN(K) k_dep<K><<<1, threads>>>(out, sink, iters); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost)); \
    printf("  %-40s %.1f cycles per iteration\n", names[K], (double)h[0] / iters);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
  }
  // wall-clock check of one of them
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k_dep<1><<<256, 64>>>(out, sink, 200000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
  printf("200000 dependent v_fma_f64 on 256 CUs: %.3f ms wall, %lld cycles -> %.0f MHz effective\n", ms, h[0], h[0] / (ms * 1e3));
  return 0;
}

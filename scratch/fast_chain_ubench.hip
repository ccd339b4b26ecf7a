// scratch microbenchmark (not product): cost breakdown of the lane-exponent CTC chain wave.
// One wave per workgroup, factors pre-staged in LDS (static ring), T frames.
// build: hipcc -O3 --offload-arch=gfx950 scratch/fast_chain_ubench.hip -o /tmp/fcu && /tmp/fcu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int kBlk = 16, kGap = 5, kEmptyE = -(1 << 28);
__device__ __forceinline__ float wave_shr1(float v,float fill){return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill),__float_as_int(v),0x138,0xf,0xf,false));}
__device__ __forceinline__ int wave_shr1_i(int v,int fill){return __builtin_amdgcn_update_dpp(fill,v,0x138,0xf,0xf,false);}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_i32(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_prefix_max_i(int v) {
  constexpr int ID = -(1 << 30);
  v = max(v, dpp_i32<0x111, 0xf>(ID, v)); v = max(v, dpp_i32<0x112, 0xf>(ID, v)); v = max(v, dpp_i32<0x114, 0xf>(ID, v));
  v = max(v, dpp_i32<0x118, 0xf>(ID, v)); v = max(v, dpp_i32<0x142, 0xa>(ID, v)); v = max(v, dpp_i32<0x143, 0xc>(ID, v));
  return v;
}
__device__ __forceinline__ float vmax(float a,float b){float m; asm("v_max_f32 %0, %1, %2":"=v"(m):"v"(a),"v"(b)); return m;}

// MODE bit0: renorm each block; bit1: checkpoint (logs + prefix max + LDS write); bit2: prefetch ring from LDS each block
// bit3: lagged renorm (exponent scan on the state two frames before the boundary, applied at the boundary)
template <int MODE>
__global__ void __launch_bounds__(64) k(int T, int L, float2* out, long long* cyc) {
  __shared__ float2 ring[4][kBlk][64];
  __shared__ float2 ckbuf[2][64];
  const int lane = threadIdx.x, b = blockIdx.x;
  const bool skip = lane >= 1 && lane < L && (lane % 3);
  for (int i = lane; i < 4 * kBlk * 64; i += 64) {
    const int l = i & 63;
    const float f = 0.3f + 0.6f * (((i * 2654435761u) >> 20) & 255) / 256.f;
    (&ring[0][0][0])[i] = make_float2(l <= L ? f : 0.f, l < L ? f * 0.9f : 0.f);
  }
  __syncthreads();
  float pb = lane == 0 ? 1.f : 0.f, pl = 0.f;
  int e = 0;
  float g = 0.f, gs = 0.f;
  bool had = false;
  auto lane_renorm = [&]() {
    const float mx = vmax(pb, pl);
    const int k2 = __builtin_amdgcn_frexp_expf(mx);
    pb = ldexpf(pb, -k2); pl = ldexpf(pl, -k2);
    const int own = mx > 0.f ? e + k2 : kEmptyE;
    const int pre = wave_prefix_max_i(own + kGap * lane) - kGap * lane;
    const int sh = mx > 0.f ? pre - own : 0;
    pb = ldexpf(pb, -min(sh, 200)); pl = ldexpf(pl, -min(sh, 200));
    e = pre;
    const int d = wave_shr1_i(e, e) - e;
    g = lane == 0 ? 0.f : ldexpf(1.f, max(d, -200));
    gs = skip ? g : 0.f;
    had = vmax(pb, pl) > 0.f;
  };
  auto frame = [&](const float2 f) {
    const float q = wave_shr1(pl, 0.f);
    const float tb = fmaf(q, g, pb);
    const float tl = fmaf(q, gs, pl + pb);
    pb = tb * f.x; pl = tl * f.y;
  };
  float2 fcur[kBlk], fnxt[kBlk];
#pragma unroll
  for (int j = 0; j < kBlk; ++j) fcur[j] = ring[0][j][lane];
  g = lane == 0 ? 0.f : 1.f; gs = skip ? g : 0.f;
  const int NB = T / kBlk;
  long long t0 = clock64();
  // lagged-renorm state: per-lane scale (power of two) decided from the state at frame kBlk-3
  for (int kk = 0; kk < NB; ++kk) {
    if (MODE & 4) {
#pragma unroll
      for (int j = 0; j < kBlk; ++j) fnxt[j] = ring[(kk + 1) & 3][j][lane];
    }
    if ((MODE & 1) && !(MODE & 8)) lane_renorm();
    if (MODE & 2) {
      const int emax = __builtin_amdgcn_readlane(wave_prefix_max_i(had ? e : kEmptyE), 63);
      const float de = (float)(e - emax);
      const float lb = pb > 0.f ? __builtin_amdgcn_logf(pb) + de : -1e30f;
      const float ll = pl > 0.f ? __builtin_amdgcn_logf(pl) + de : -1e30f;
      ckbuf[kk & 1][lane] = make_float2(lb, ll);
    }
    if (MODE & 8) {
      // lagged: the scan runs on the state as of frame kBlk-3 and is applied (folded into the factors
      // of the last frame) two frames later -- off the dependent path
      int pre = 0, own = 0; float mx = 0.f; int k2 = 0;
#pragma unroll
      for (int j = 0; j < kBlk; ++j) {
        if (j == kBlk - 3) {
          mx = vmax(pb, pl);
          k2 = __builtin_amdgcn_frexp_expf(mx);
          own = mx > 0.f ? e + k2 : kEmptyE;
          pre = wave_prefix_max_i(own + kGap * lane) - kGap * lane;
        }
        if (j == kBlk - 1) {
          const int sh = mx > 0.f ? pre - own : 0;
          const float sc = ldexpf(1.f, -(k2 + min(sh, 200)));  // per-lane power of two
          const int d = wave_shr1_i(pre, pre) - pre;
          const float gn = lane == 0 ? 0.f : ldexpf(1.f, max(d, -200));
          // last frame of the block with the OLD coupling, then rescale via the factors
          const float q = wave_shr1(pl, 0.f);
          const float tb = fmaf(q, g, pb);
          const float tl = fmaf(q, gs, pl + pb);
          pb = tb * (fcur[j].x * sc); pl = tl * (fcur[j].y * sc);
          e = pre; g = gn; gs = skip ? gn : 0.f;
        } else {
          frame(fcur[j]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < kBlk; ++j) frame(fcur[j]);
    }
    if (MODE & 4) {
#pragma unroll
      for (int j = 0; j < kBlk; ++j) fcur[j] = fnxt[j];
    }
  }
  long long t1 = clock64();
  if (lane == 0) cyc[b] = t1 - t0;
  out[b * 64 + lane] = make_float2(pb + e, pl + ckbuf[0][lane].x);
}
template <int MODE> void run(const char* name, int B, int T, int L, float2* out, long long* cyc) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(B), dim3(64), 0, 0, T, L, out, cyc);
  CK(hipEventRecord(e0)); const int R = 20;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k<MODE>, dim3(B), dim3(64), 0, 0, T, L, out, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(B); CK(hipMemcpy(h.data(), cyc, B * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (auto v : h) avg += v; avg /= B;
  printf("%-44s %.1f us/launch, clock64 ticks/frame %.1f (ticks are 100 MHz: x24 for 2.4 GHz cycles)\n", name, ms * 1e3 / R, avg / T);
}
int main() {
  const int B = 256, T = 1008, L = 44;
  float2* out; long long* cyc; CK(hipMalloc(&out, B * 64 * 8)); CK(hipMalloc(&cyc, B * 8));
  run<0>("frames only (regs)", B, T, L, out, cyc);
  run<4>("frames + LDS prefetch", B, T, L, out, cyc);
  run<5>("frames + prefetch + renorm", B, T, L, out, cyc);
  run<7>("frames + prefetch + renorm + checkpoint", B, T, L, out, cyc);
  run<12>("frames + prefetch + lagged renorm", B, T, L, out, cyc);
  run<14>("frames + prefetch + lagged renorm + ckpt", B, T, L, out, cyc);
  return 0;
}

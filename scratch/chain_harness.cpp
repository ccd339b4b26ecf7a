// scratch (not product): time wfl_ctc_forward(WFL_CTC_FAST_CHAIN) of a given libwfl build on synthetic cfg2 data.
// build: hipcc -O2 scratch/chain_harness.cpp -o scratch/chain_harness.bin -ldl ; run: chain_harness.bin <lib.so> [flags]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef int (*ws_fn)(int, int, int, int, int64_t*);
typedef int (*fwd_fn)(const float*, int, int, int, const int32_t*, const int64_t*, int, int, int, float*, float*, void*);
int main(int argc, char** argv) {
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  const int flags = argc > 2 ? atoi(argv[2]) : 2;
  ws_fn wsf = (ws_fn)dlsym(h, "wfl_ctc_workspace");
  fwd_fn fwd = (fwd_fn)dlsym(h, "wfl_ctc_forward");
  const int B = 128, T = argc > 3 ? atoi(argv[3]) : 1000, C = 100, L = 44;
  std::vector<float> x((size_t)B * T * C);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.f); };
  for (auto& v : x) { float a = 0; for (int i = 0; i < 12; ++i) a += rnd(); v = a - 6.f; }
  std::vector<int32_t> tg(B * L); for (auto& v : tg) v = (int)(rnd() * (C - 2));
  std::vector<int64_t> off(B + 1); for (int i = 0; i <= B; ++i) off[i] = (int64_t)i * L;
  int64_t nws = 0; wsf(B, T, C, L, &nws);
  float *dx, *dws, *dnll; int32_t* dtg; int64_t* doff;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dws, nws * 4)); CK(hipMalloc(&dnll, B * 4));
  CK(hipMalloc(&dtg, tg.size() * 4)); CK(hipMalloc(&doff, off.size() * 8));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtg, tg.data(), tg.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(doff, off.data(), off.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) if (fwd(dx, B, T, C, dtg, doff, L, C - 1, flags, dws, dnll, nullptr)) { printf("fwd failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  const int R = 50;
  CK(hipEventRecord(e0));
  for (int i = 0; i < R; ++i) fwd(dx, B, T, C, dtg, doff, L, C - 1, flags, dws, dnll, nullptr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> nll(B); CK(hipMemcpy(nll.data(), dnll, B * 4, hipMemcpyDeviceToHost));
  printf("%s flags=%d T=%d: %.1f us per forward; nll[0]=%.3f nll[1]=%.3f\n", argv[1], flags, T, ms * 1e3 / R, nll[0], nll[1]);
  return 0;
}

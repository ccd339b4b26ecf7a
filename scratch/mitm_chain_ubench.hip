// scratch microbenchmark (not product): where the ~990 cycles per 16-frame block of the meet-in-the-middle CTC chain
// wave go.  A replica of the steady block of ctc_mitm.h's chain wave (same frame asm, same renormalisation, same LDS
// traffic), alone on its CU or next to 15 polling waves, with its pieces switched off one at a time.
// build: hipcc -O3 --offload-arch=gfx950 scratch/mitm_chain_ubench.hip -o scratch/mitm_chain_ubench && scratch/mitm_chain_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
constexpr int kBlk = 16, kGap = 5, kEmptyE = -(1 << 28), kSlots = 9;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ int lds_peek(const int* p) { return *(const volatile lds_int_t*)(const lds_int_t*)p; }
__device__ __forceinline__ void lds_post(int* p, int v) { asm volatile("" ::: "memory"); *(volatile lds_int_t*)(lds_int_t*)p = v; }
__device__ __forceinline__ int wave_shr1_i(int v,int fill){return __builtin_amdgcn_update_dpp(fill,v,0x138,0xf,0xf,false);}
__device__ __forceinline__ float vmax(float a,float b){float m; asm("v_max_f32 %0, %1, %2":"=v"(m):"v"(a),"v"(b)); return m;}
__device__ __forceinline__ int wave_prefix_max_i(int v) {
  asm volatile(
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v));
  return v;
}
#define WFL_FRAME(P, PH, TT, TL, TH, F, FY)                                   \
  "v_pk_mul_f32 v[6:7], " F ", %[G]\n\t"                                      \
  "v_pk_mul_f32 " TT ", " F ", " P " op_sel_hi:[1,0]\n\t"                     \
  "v_fmac_f32_dpp " TL ", " PH ", v6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp " TH ", " PH ", v7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32 " TH ", " FY ", " PH "\n\t"

struct Lds {
  float2 ring[kSlots][kBlk][64];
  float4 ck[kSlots][64];
  int clampe[kSlots][64];
  int staged[kSlots];
  int chainpos;
  int done;
};

// MODE bits: 1 renorm, 2 ring reads, 4 checkpoint + post, 8 frames, 16 flag peek + test, 32 renorm WITHOUT the prefix scan
// (clamp exponents read from LDS), 128 renorm every second block only
template <int MODE>
__global__ void __launch_bounds__(1024, 1) k(int NB, int L, float* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Lds& S = *reinterpret_cast<Lds*>(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool skip = lane >= 1 && lane < L && (lane % 3);
  for (int i = threadIdx.x; i < kSlots * kBlk * 64; i += blockDim.x) {
    const int l = i & 63;
    const float f = 0.55f + 0.6f * (((i * 2654435761u) >> 20) & 255) / 256.f;
    (&S.ring[0][0][0])[i] = make_float2(l <= L ? f : 0.f, l < L ? f * 0.9f : 0.f);
  }
  for (int i = threadIdx.x; i < kSlots * 64; i += blockDim.x) (&S.clampe[0][0])[i] = kEmptyE;
  if (threadIdx.x < kSlots) S.staged[threadIdx.x] = 0x7fffffff;
  if (threadIdx.x == 0) S.chainpos = 0, S.done = 0;
  __syncthreads();
  if (wave != 0) {
    // co-resident waves: poll like the flusher / fetcher / idle emitters of the first half do
    if (wave & 1) { while (lds_peek(&S.done) == 0) __builtin_amdgcn_s_sleep(1); } else { while (lds_peek(&S.done) == 0) __builtin_amdgcn_s_sleep(4); }
    return;
  }
  __builtin_amdgcn_s_setprio(3);
  float pb = lane == 0 ? 1.f : 0.f, pl = 0.f;
  int e = 0;
  float g = lane == 0 ? 0.f : 1.f, gs = skip ? g : 0.f;
  auto lane_renorm = [&]() {
    const float mx = vmax(pb, pl);
    const int k2 = __builtin_amdgcn_frexp_expf(mx);
    const int own = mx > 0.f ? e + k2 : kEmptyE;
    const int pre = wave_prefix_max_i(own + kGap * lane) - kGap * lane;
    const int sh = mx > 0.f ? pre - own : 0;
    pb = ldexpf(pb, -(k2 + min(sh, 200)));
    pl = ldexpf(pl, -(k2 + min(sh, 200)));
    e = pre;
    const int d = wave_shr1_i(e, e) - e;
    g = lane == 0 ? 0.f : ldexpf(1.f, max(d, -200));
    gs = skip ? g : 0.f;
  };
  auto lane_renorm_cheap = [&](int clampv) {
    const float mx = vmax(pb, pl);
    const int k2 = __builtin_amdgcn_frexp_expf(mx);
    const int own = mx > 0.f ? e + k2 : kEmptyE;
    const int pre = max(own, clampv);
    pb = ldexpf(pb, e - pre);  // (-(k2 + sh) = e - pre when mx > 0; a zero stays zero)
    pl = ldexpf(pl, e - pre);
    e = pre;
    const int d = wave_shr1_i(e, e) - e;
    g = lane == 0 ? 0.f : ldexpf(1.f, min(max(d, -200), 60));
    gs = skip ? g : 0.f;
  };
  auto frames4 = [&](const float2& f0, const float2& f1, const float2& f2, const float2& f3) {
    v2f Pq = {pb, pl};
    const v2f G = {g, gs};
    const v2f F0 = {f0.x, f0.y}, F1 = {f1.x, f1.y}, F2 = {f2.x, f2.y}, F3 = {f3.x, f3.y};
    asm volatile(WFL_FRAME("v[2:3]", "v3", "v[4:5]", "v4", "v5", "%[F0]", "%[Y0]")
                 WFL_FRAME("v[4:5]", "v5", "v[2:3]", "v2", "v3", "%[F1]", "%[Y1]")
                 WFL_FRAME("v[2:3]", "v3", "v[4:5]", "v4", "v5", "%[F2]", "%[Y2]")
                 WFL_FRAME("v[4:5]", "v5", "v[2:3]", "v2", "v3", "%[F3]", "%[Y3]")
                 : "+{v[2:3]}"(Pq)
                 : [G] "v"(G), [F0] "v"(F0), [F1] "v"(F1), [F2] "v"(F2), [F3] "v"(F3), [Y0] "v"(f0.y), [Y1] "v"(f1.y),
                   [Y2] "v"(f2.y), [Y3] "v"(f3.y)
                 : "v4", "v5", "v6", "v7");
    pb = Pq.x;
    pl = Pq.y;
  };
  // four instructions per frame: (pb, pl + pb) by one packed fma with the constant pair (0, 1), the two DPP
  // multiply-adds with the block's g / gs, ONE packed multiply by the frame's factors -- no F * G product
#define WFL_FRAME4(F, NOP)                                                              \
  "v_pk_fma_f32 v[4:5], v[2:3], %[C01], v[2:3] op_sel_hi:[0,1,1]\n\t" NOP              \
  "v_fmac_f32_dpp v4, v3, %[Gx] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"             \
  "v_fmac_f32_dpp v5, v3, %[Gy] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"             \
  "v_pk_mul_f32 v[2:3], v[4:5], " F "\n\t"
  auto frames4b = [&](const float2& f0, const float2& f1, const float2& f2, const float2& f3) {
    v2f Pq = {pb, pl};
    const v2f C01 = {0.f, 1.f};
    const v2f F0 = {f0.x, f0.y}, F1 = {f1.x, f1.y}, F2 = {f2.x, f2.y}, F3 = {f3.x, f3.y};
    if (MODE & 512)
      asm volatile(WFL_FRAME4("%[F0]", "") WFL_FRAME4("%[F1]", "") WFL_FRAME4("%[F2]", "") WFL_FRAME4("%[F3]", "")
                   : "+{v[2:3]}"(Pq)
                   : [C01] "v"(C01), [Gx] "v"(g), [Gy] "v"(gs), [F0] "v"(F0), [F1] "v"(F1), [F2] "v"(F2), [F3] "v"(F3)
                   : "v4", "v5");
    else
      asm volatile(WFL_FRAME4("%[F0]", "s_nop 0\n\t") WFL_FRAME4("%[F1]", "s_nop 0\n\t") WFL_FRAME4("%[F2]", "s_nop 0\n\t")
                   WFL_FRAME4("%[F3]", "s_nop 0\n\t")
                   : "+{v[2:3]}"(Pq)
                   : [C01] "v"(C01), [Gx] "v"(g), [Gy] "v"(gs), [F0] "v"(F0), [F1] "v"(F1), [F2] "v"(F2), [F3] "v"(F3)
                   : "v4", "v5");
    pb = Pq.x;
    pl = Pq.y;
  };
  float2 fa[kBlk], fz[kBlk];
#pragma unroll
  for (int j = 0; j < kBlk; ++j) fa[j] = S.ring[0][j][lane], fz[j] = S.ring[1][j][lane];
  int nflag = lds_peek(&S.staged[1]);
  int cl_next = S.clampe[1][lane];
  int s0 = 0, s1 = 1, s2 = 2;
  auto block = [&](int kk, const float2 (&fcur)[kBlk], float2 (&fnxt)[kBlk]) {
    if (MODE & 16) {
      if (nflag < kk + 2) {
        while (lds_peek(&S.staged[s1]) < kk + 2) {}
      }
      asm volatile("" ::: "memory");
    }
    if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < kBlk; ++j) fnxt[j] = S.ring[s1][j][lane];
    }
    const int cl = cl_next;
    if (MODE & 32) cl_next = S.clampe[s2][lane];
    if (MODE & 16) nflag = lds_peek(&S.staged[s2]);
    if (MODE & 32) lane_renorm_cheap(cl);
    else if ((MODE & 1) && (!(MODE & 128) || (kk & 1))) lane_renorm();
    if (MODE & 4) {
      S.ck[s0][lane] = make_float4(pb, pl, __int_as_float(e), 0.f);
      lds_post(&S.chainpos, kk + 2);
    }
    s0 = s1, s1 = s2, s2 = s2 + 1 == kSlots ? 0 : s2 + 1;
    if ((MODE & 8) && (MODE & 256)) {
#pragma unroll
      for (int j = 0; j < kBlk; j += 4) frames4b(fcur[j], fcur[j + 1], fcur[j + 2], fcur[j + 3]);
    } else if (MODE & 8) {
#pragma unroll
      for (int j = 0; j < kBlk; j += 4) frames4(fcur[j], fcur[j + 1], fcur[j + 2], fcur[j + 3]);
    } else {
      pb += fcur[0].x + fcur[kBlk - 1].y;
      pl += fcur[3].x;
    }
  };
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int kk = 1; kk + 1 < NB; kk += 2) {
    block(kk, fz, fa);
    block(kk + 1, fa, fz);
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  lds_post(&S.done, 1);
  out[blockIdx.x * 64 + lane] = pb + pl + (float)e + g;
  if (lane == 0) cyc[blockIdx.x * 2] = t1 - t0, cyc[blockIdx.x * 2 + 1] = w1 - w0;
}

template <int MODE>
void run(const char* what, int waves) {
  const int NB = 4096, WG = 256;
  float* out; long long* cyc;
  CK(hipMalloc(&out, WG * 64 * 4)); CK(hipMalloc(&cyc, WG * 16));
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<MODE><<<WG, waves * 64, sizeof(Lds)>>>(NB, 44, out, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k<MODE><<<WG, waves * 64, sizeof(Lds)>>>(NB, 44, out, cyc);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<long long> h(WG * 2); CK(hipMemcpy(h.data(), cyc, WG * 16, hipMemcpyDeviceToHost));
  std::vector<float> o(WG * 64); CK(hipMemcpy(o.data(), out, WG * 64 * 4, hipMemcpyDeviceToHost));
  double c = 0, w = 0; for (int i = 0; i < WG; ++i) c += h[2 * i], w += h[2 * i + 1];
  c /= WG; w /= WG;
  printf("%-58s waves %2d | kernel %8.1f us | clock64/block %7.1f | wall ns/block %7.1f (=> %5.2f GHz) | out %g\n", what, waves, ms * 1e3,
         c / (NB - 2), w * 10.0 / (NB - 2), c / (w * 10.0), o[45]);
  CK(hipFree(out)); CK(hipFree(cyc));
}


// ---------------------------------------------------------------------------------------------------------------------
// Pipeline replica: the chain wave fed by REAL stagers (gathers from HBM with the CTC access pattern, exp2 factors, ring
// writes, slot handshake) and followed by a flusher (checkpoint pick-up, optional device-coherent global stores), the
// other waves polling like idle emitters.  PIPE bits: 1 stagers gather from x (else constants), 2 stagers compute the
// factors (else copy), 4 flusher stores three words per block to global memory, 8 chain skips its frames,
// 16 flusher sleeps longer between polls, 32 stagers poll with s_sleep 8 instead of 2
struct PLds {
  float2 ring[kSlots][kBlk][64];
  float4 ck[kSlots][64];
  float fref[kSlots][kBlk];
  int staged[kSlots];
  int chainpos, ckdone, done;
};
template <int PIPE>
__global__ void __launch_bounds__(1024, 1) kp(int NB, int L, int C, const float* __restrict__ x, const int* __restrict__ cols,
                                               float* out, long long* cyc, unsigned long long* pub) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PLds& S = *reinterpret_cast<PLds*>(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool skip = lane >= 1 && lane < L && (lane % 3);
  if (threadIdx.x < kSlots) S.staged[threadIdx.x] = 0;
  if (threadIdx.x == 0) S.chainpos = 0, S.ckdone = 0, S.done = 0;
  __syncthreads();
  const float* xrow = x + (size_t)blockIdx.x * NB * kBlk * C;
  const int col = cols[blockIdx.x * 64 + lane];
  const bool has_label = lane < L, has_blank = lane <= L;
  if (wave == 1 || wave == 2 || wave == 3 || wave == 5) {
    __builtin_amdgcn_s_setprio(2);
    const int h = wave == 5 ? 3 : wave - 1;
    auto issue = [&](int n, float (&raw)[kBlk]) {
      if (!(PIPE & 1)) {
#pragma unroll
        for (int j = 0; j < kBlk; ++j) raw[j] = 0.01f * (float)((lane + j + n) & 31);
        return;
      }
#pragma unroll
      for (int j = 0; j < kBlk; ++j) raw[j] = xrow[(size_t)(n * kBlk + j) * C + col];
    };
    auto stage = [&](int n, const float (&raw)[kBlk]) {
      const int slot = n % kSlots;
      if (!(PIPE & 2)) {
#pragma unroll
        for (int j = 0; j < kBlk; ++j) S.ring[slot][j][lane] = make_float2(has_blank ? 0.7f : 0.f, has_label ? 0.7f + 0.001f * raw[j] : 0.f);
        if (lane < kBlk) S.fref[slot][lane] = 0.f;
        return;
      }
      float m = raw[0];
#pragma unroll
      for (int j = 1; j < kBlk; ++j) m = vmax(m, raw[j]);
      // (a stand-in for the 16-way fold: ~50 dependent DPP instructions)
      float r = m;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        r = vmax(r, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x111, 0xf, 0xf, false)));
        r = vmax(r, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x114, 0xf, 0xf, false)));
        r = vmax(r, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x142, 0xa, 0xf, false)));
      }
      const float rr = rintf(r * 1.4426950408889634f);
      const float hb = has_blank ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < kBlk; ++j) {
        const float rj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rr), j));
        const float f = __builtin_amdgcn_exp2f(vmax(fmaf(raw[j], 1.4426950408889634f, -rj), -__builtin_inff()));
        const float fb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), L));
        S.ring[slot][j][lane] = make_float2(fb * hb, has_label ? f : 0.f);
      }
      if (lane < kBlk) S.fref[slot][lane] = rr;
    };
    auto wait_slot = [&](int n) {
      if (n < kSlots) return;
      const int m = n - kSlots;
      while (lds_peek(&S.chainpos) < m + 1 || lds_peek(&S.ckdone) < m + 1) {
        if (PIPE & 32) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
      }
    };
    float ra[kBlk], rb[kBlk];
    if (h < NB) issue(h, ra);
    if (h + 4 < NB) issue(h + 4, rb);
    for (int n = h; n < NB; n += 8) {
      wait_slot(n);
      stage(n, ra);
      lds_post(&S.staged[n % kSlots], n + 1);
      if (n + 8 < NB) issue(n + 8, ra);
      const int n2 = n + 4;
      if (n2 < NB) {
        wait_slot(n2);
        stage(n2, rb);
        lds_post(&S.staged[n2 % kSlots], n2 + 1);
        if (n2 + 8 < NB) issue(n2 + 8, rb);
      }
    }
    return;
  }
  if (wave == 4) {
    unsigned long long* dst = pub + (size_t)blockIdx.x * NB * 128;
    float acc = 0.f;
    for (int kk = 0; kk < NB; ++kk) {
      while (lds_peek(&S.chainpos) < kk + 2 && lds_peek(&S.done) == 0) {
        if (PIPE & 16) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      const int slot = kk % kSlots;
      const float rj = lane < kBlk ? S.fref[slot][lane] : 0.f;
      if (PIPE & 4) {
        const float4 m = S.ck[slot][lane];
        const unsigned long long vb = (unsigned long long)__float_as_uint(m.x) | ((unsigned long long)__float_as_uint(m.z) << 32);
        const unsigned long long vl = (unsigned long long)__float_as_uint(m.y) | ((unsigned long long)__float_as_uint(m.z) << 32);
        unsigned long long* d0 = dst + (size_t)kk * 128 + lane;
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(d0), "v"(vb) : "memory");
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(d0 + 64), "v"(vl) : "memory");
        if (lane == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(d0 + 127), "v"(vb) : "memory");
      }
      lds_post(&S.ckdone, kk + 1);
      float r = rj;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x111, 0xf, 0xf, false));
        r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x114, 0xf, 0xf, false));
        r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x142, 0xa, 0xf, false));
      }
      acc += r;
    }
    out[blockIdx.x * 64 + lane] += acc * 1e-30f;
    return;
  }
  if (wave != 0) {
    while (lds_peek(&S.done) == 0) __builtin_amdgcn_s_sleep(4);
    return;
  }
  __builtin_amdgcn_s_setprio(3);
  float pb = lane == 0 ? 1.f : 0.f, pl = 0.f;
  int e = 0;
  float g = lane == 0 ? 0.f : 1.f, gs = skip ? g : 0.f;
  auto lane_renorm = [&]() {
    const float mx = vmax(pb, pl);
    const int k2 = __builtin_amdgcn_frexp_expf(mx);
    const int own = mx > 0.f ? e + k2 : kEmptyE;
    const int pre = wave_prefix_max_i(own + kGap * lane) - kGap * lane;
    const int sh = mx > 0.f ? pre - own : 0;
    pb = ldexpf(pb, -(k2 + min(sh, 200)));
    pl = ldexpf(pl, -(k2 + min(sh, 200)));
    e = pre;
    const int d = wave_shr1_i(e, e) - e;
    g = lane == 0 ? 0.f : ldexpf(1.f, max(d, -200));
    gs = skip ? g : 0.f;
  };
  auto frames4 = [&](const float2& f0, const float2& f1, const float2& f2, const float2& f3) {
    v2f Pq = {pb, pl};
    const v2f G = {g, gs};
    const v2f F0 = {f0.x, f0.y}, F1 = {f1.x, f1.y}, F2 = {f2.x, f2.y}, F3 = {f3.x, f3.y};
    asm volatile(WFL_FRAME("v[2:3]", "v3", "v[4:5]", "v4", "v5", "%[F0]", "%[Y0]")
                 WFL_FRAME("v[4:5]", "v5", "v[2:3]", "v2", "v3", "%[F1]", "%[Y1]")
                 WFL_FRAME("v[2:3]", "v3", "v[4:5]", "v4", "v5", "%[F2]", "%[Y2]")
                 WFL_FRAME("v[4:5]", "v5", "v[2:3]", "v2", "v3", "%[F3]", "%[Y3]")
                 : "+{v[2:3]}"(Pq)
                 : [G] "v"(G), [F0] "v"(F0), [F1] "v"(F1), [F2] "v"(F2), [F3] "v"(F3), [Y0] "v"(f0.y), [Y1] "v"(f1.y),
                   [Y2] "v"(f2.y), [Y3] "v"(f3.y)
                 : "v4", "v5", "v6", "v7");
    pb = Pq.x;
    pl = Pq.y;
  };
  float2 fa[kBlk], fz[kBlk];
  while (lds_peek(&S.staged[0]) != 1) {}
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < kBlk; ++j) fa[j] = S.ring[0][j][lane];
  lds_post(&S.chainpos, 1);
  int nflag = lds_peek(&S.staged[1]);
  int s0 = 0, s1 = 1, s2 = 2;
  long long waited = 0;
  auto block = [&](int kk, const float2 (&fcur)[kBlk], float2 (&fnxt)[kBlk]) {
    if (nflag != kk + 2) {
      const long long w0 = clock64();
      while (lds_peek(&S.staged[s1]) != kk + 2) {}
      waited += clock64() - w0;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < kBlk; ++j) fnxt[j] = S.ring[s1][j][lane];
    nflag = lds_peek(&S.staged[s2]);
    lane_renorm();
    S.ck[s0][lane] = make_float4(pb, pl, __int_as_float(e), 0.f);
    lds_post(&S.chainpos, kk + 2);
    s0 = s1, s1 = s2, s2 = s2 + 1 == kSlots ? 0 : s2 + 1;
    if (!(PIPE & 8)) {
#pragma unroll
      for (int j = 0; j < kBlk; j += 4) frames4(fcur[j], fcur[j + 1], fcur[j + 2], fcur[j + 3]);
    } else {
      pb += fcur[0].x + fcur[kBlk - 1].y;
      pl += fcur[3].x;
    }
  };
  // (the first 16 blocks are start-up: cold gathers; the pace is taken over the rest)
  long long t0 = 0, w0 = 0;
  block(0, fa, fz);
  for (int kk = 1; kk + 3 < NB; kk += 2) {
    if (kk == 17) t0 = clock64(), w0 = wall_clock64(), waited = 0;
    block(kk, fz, fa);
    block(kk + 1, fa, fz);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  lds_post(&S.chainpos, NB + 8);
  lds_post(&S.done, 1);
  out[blockIdx.x * 64 + lane] = pb + pl + (float)e + g;
  if (lane == 0) cyc[blockIdx.x * 4] = t1 - t0, cyc[blockIdx.x * 4 + 1] = w1 - w0, cyc[blockIdx.x * 4 + 2] = waited;
}

template <int PIPE>
void runp(const char* what) {
  const int NB = 256, WG = 256, C = 100, L = 44;
  float* out; long long* cyc; float* x; int* cols; unsigned long long* pub;
  const size_t xn = (size_t)WG * NB * kBlk * C;
  CK(hipMalloc(&out, WG * 64 * 4)); CK(hipMalloc(&cyc, WG * 32)); CK(hipMalloc(&x, xn * 4)); CK(hipMalloc(&cols, WG * 64 * 4));
  CK(hipMalloc(&pub, (size_t)WG * NB * 128 * 8));
  CK(hipMemset(out, 0, WG * 64 * 4));
  CK(hipMemset(x, 0, xn * 4));
  std::vector<int> hc(WG * 64);
  for (size_t i = 0; i < hc.size(); ++i) hc[i] = (int)((i * 2654435761u >> 8) % (unsigned)(C - 2));
  for (int wgi = 0; wgi < WG; ++wgi) for (int l = L; l < 64; ++l) hc[wgi * 64 + l] = C - 1;
  CK(hipMemcpy(cols, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)kp<PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PLds)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  kp<PIPE><<<WG, 1024, sizeof(PLds)>>>(NB, L, C, x, cols, out, cyc, pub);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  kp<PIPE><<<WG, 1024, sizeof(PLds)>>>(NB, L, C, x, cols, out, cyc, pub);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<long long> h(WG * 4); CK(hipMemcpy(h.data(), cyc, WG * 32, hipMemcpyDeviceToHost));
  double c = 0, w = 0, wt = 0; for (int i = 0; i < WG; ++i) c += h[4 * i], w += h[4 * i + 1], wt += h[4 * i + 2];
  c /= WG; w /= WG; wt /= WG;
  const int nb = NB - 3 - 17;
  printf("%-66s | kernel %8.1f us | clock64/block %7.1f (waiting for staged %6.1f) | wall ns/block %7.1f\n", what, ms * 1e3, c / nb, wt / nb, w * 10.0 / nb);
  CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(x)); CK(hipFree(cols)); CK(hipFree(pub));
}

int main() {
  if (getenv("FRAME_ONLY")) {
    for (int waves : {1, 16}) {
      run<8>("frames only, five instructions", waves);
      run<8 + 256>("frames only, four instructions + s_nop 0", waves);
      run<8 + 256 + 512>("frames only, four instructions, NO nop (hazard: timing only)", waves);
      run<31>("full block, five-instruction frames", waves);
      run<31 + 256>("full block, four-instruction frames + s_nop 0", waves);
    }
    return 0;
  }
  if (getenv("PIPE_ONLY") == nullptr) {
  for (int waves : {1, 16}) {
    run<31>("full block (renorm, reads, ck, frames, flag)", waves);
    run<30>("no renorm", waves);
    run<29>("no ring reads", waves);
    run<27>("no checkpoint/post", waves);
    run<23>("no frames", waves);
    run<15>("no flag peek", waves);
    run<8>("frames only", waves);
    run<10>("frames + reads", waves);
    run<9>("frames + renorm", waves);
    run<1>("renorm only", waves);
    run<2>("reads only", waves);
    run<31 + 32>("cheap renorm (clamp from LDS, no scan), everything else", waves);
    run<31 + 128>("full, renorm every second block", waves);
  }
  }
  runp<0>("pipeline: constants, no compute, no stores");
  runp<1>("pipeline: + gathers");
  runp<2>("pipeline: + factor compute");
  runp<3>("pipeline: + gathers + compute");
  runp<4>("pipeline: + flusher stores");
  runp<7>("pipeline: gathers + compute + stores (the first half of the launch)");
  runp<7 + 8>("pipeline: everything, chain without frames");
  runp<7 + 16>("pipeline: everything, flusher polls with s_sleep 8");
  runp<7 + 32>("pipeline: everything, stagers poll with s_sleep 8");
  runp<7 + 16 + 32>("pipeline: everything, both poll slowly");
  return 0;
}

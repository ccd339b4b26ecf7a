// scratch microbenchmark (not product): the chain wave of the meet-in-the-middle CTC launch WITHOUT its renormalisation --
// the lane exponents of block n are predicted by a helper wave from the checkpoint of block n - LAG, the chain wave only
// multiplies its state by the per-lane power of two it is handed -- and with its LDS reads issued BETWEEN the frames of
// the block before (one per two frames) instead of back to back.  Same pipeline replica as mitm_chain_ubench.hip's kp<>:
// real stagers (gathers with the CTC access pattern, exp2 factors, ring writes), a flusher, the other waves polling.
// build: hipcc -O3 --offload-arch=gfx950 scratch/lean_chain_ubench.hip -o scripts/_build/lean_chain_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
constexpr int kBlk = 16, kGap = 5, kEmptyE = -(1 << 28), kSlots = 9, kScl = 16;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ int lds_peek(const int* p) { return *(const volatile lds_int_t*)(const lds_int_t*)p; }
__device__ __forceinline__ void lds_post(int* p, int v) { asm volatile("" ::: "memory"); *(volatile lds_int_t*)(lds_int_t*)p = v; }
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
// one frame: state pair P -> T (the two swap roles every frame).  F: the frame's (fb, fl) register pair, FY its high half
#define WFL_FRAME(P, PH, TT, TL, TH, F, FY)                                   \
  "v_pk_mul_f32 v[6:7], " F ", %[G]\n\t"                                      \
  "v_pk_mul_f32 " TT ", " F ", " P " op_sel_hi:[1,0]\n\t"                     \
  "v_fmac_f32_dpp " TL ", " PH ", v6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp " TH ", " PH ", v7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32 " TH ", " FY ", " PH "\n\t"
#define FR_A(F, FY) WFL_FRAME("v[2:3]", "v3", "v[4:5]", "v4", "v5", F, FY)
#define FR_B(F, FY) WFL_FRAME("v[4:5]", "v5", "v[2:3]", "v2", "v3", F, FY)

struct PLds {
  v4f ring[kSlots][kBlk / 2][64];  // (fb, fl) of two frames per entry
  float2 ck[kSlots][64];           // chain: state at the block's start (after its scaling)
  v4f scl[kScl][64];               // helper: (g, gs, 2^(e[n-1] - e[n]), tag = n + 1) of block n
  int cke[kScl][64];               // helper: lane exponents of block n
  float fref[kSlots][kBlk];
  int staged[kSlots];
  int chainpos, ckpos, ckdone, done;
};

// PIPE bits: 1 stagers gather from x (else constants), 2 stagers compute the factors (else copy), 4 flusher stores,
// 64 helper on the chain's SIMD (wave 12) instead of wave 6
template <int PIPE, int LAG, int CH = 7>
__global__ void __launch_bounds__(1024, 1) kp(int NB, int L, int C, const float* __restrict__ x, const int* __restrict__ cols,
                                               float* out, long long* cyc, unsigned long long* pub) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PLds& S = *reinterpret_cast<PLds*>(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool skip = lane >= 1 && lane < L && (lane % 3);
  if (threadIdx.x < kSlots) S.staged[threadIdx.x] = 0;
  if (threadIdx.x == 0) S.chainpos = 0, S.ckpos = 0, S.ckdone = 0, S.done = 0;
  for (int i = threadIdx.x; i < kScl * 64; i += blockDim.x) (&S.scl[0][0])[i] = v4f{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const float* xrow = x + (size_t)blockIdx.x * NB * kBlk * C;
  const int col = cols[blockIdx.x * 64 + lane];
  const bool has_label = lane < L, has_blank = lane <= L;
  constexpr int kHelper = (PIPE & 64) ? 12 : 6;
  if ((CH & 8) && wave != 0) return;
  if ((CH & 16) && wave != 0 && wave != 1 && wave != 2 && wave != 3 && wave != 5 && wave != 4 && wave != kHelper) return;
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
        for (int j = 0; j < kBlk; j += 2)
          S.ring[slot][j >> 1][lane] = v4f{has_blank ? 0.45f : 0.f, has_label ? 0.45f + 0.001f * raw[j] : 0.f, has_blank ? 0.45f : 0.f,
                                           has_label ? 0.45f + 0.001f * raw[j + 1] : 0.f};
        if (lane < kBlk) S.fref[slot][lane] = 0.f;
        return;
      }
      float m = raw[0];
#pragma unroll
      for (int j = 1; j < kBlk; ++j) m = vmax(m, raw[j]);
      float r = m;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        r = vmax(r, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x111, 0xf, 0xf, false)));
        r = vmax(r, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x114, 0xf, 0xf, false)));
        r = vmax(r, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x142, 0xa, 0xf, false)));
      }
      const float rr = rintf(r * 1.4426950408889634f);
      const float hb = has_blank ? 1.f : 0.f;
      float fbv[kBlk], flv[kBlk];
#pragma unroll
      for (int j = 0; j < kBlk; ++j) {
        const float rj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rr), j));
        const float f = __builtin_amdgcn_exp2f(vmax(fmaf(raw[j], 1.4426950408889634f, -rj), -__builtin_inff()));
        fbv[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), L)) * hb;
        flv[j] = has_label ? f : 0.f;
      }
#pragma unroll
      for (int j = 0; j < kBlk; j += 2) S.ring[slot][j >> 1][lane] = v4f{fbv[j], flv[j], fbv[j + 1], flv[j + 1]};
      if (lane < kBlk) S.fref[slot][lane] = rr;
    };
    auto wait_slot = [&](int n) {
      if (n < kSlots) return;
      const int m = n - kSlots;
      while (lds_peek(&S.chainpos) < m + 1 || lds_peek(&S.ckdone) < m + 1) __builtin_amdgcn_s_sleep(2);
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
    // flusher: as in the launch (reference sums, checkpoints to global memory in the first half)
    unsigned long long* dst = pub + (size_t)blockIdx.x * NB * 128;
    float acc = 0.f;
    for (int kk = 0; kk < NB; ++kk) {
      while (lds_peek(&S.ckpos) < kk + 1 && lds_peek(&S.done) == 0) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      const int slot = kk % kSlots;
      const float rj = lane < kBlk ? S.fref[slot][lane] : 0.f;
      if (PIPE & 4) {
        const float2 m = S.ck[slot][lane];
        const int e = S.cke[kk % kScl][lane];
        const unsigned long long vb = (unsigned long long)__float_as_uint(m.x) | ((unsigned long long)(unsigned)e << 32);
        const unsigned long long vl = (unsigned long long)__float_as_uint(m.y) | ((unsigned long long)(unsigned)e << 32);
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
  if (wave == kHelper) {
    // ---------------------------------------------------------------- the helper: lane exponents, a few blocks ahead.
    // One look at a checkpoint gives the exponents of TWO blocks (m, m + 1), predicted LAG and LAG + 1 blocks ahead.
    __builtin_amdgcn_s_setprio(2);
    int own_prev = kEmptyE;
    {
      const float g0 = lane == 0 ? 0.f : 1.f;
      S.scl[0][lane] = v4f{g0, skip ? g0 : 0.f, __int_as_float(0), __int_as_float(1)};
      S.cke[0][lane] = 0;
    }
    auto emit = [&](int m, int pred) {
      const int e_m = wave_prefix_max_i(pred + kGap * lane) - kGap * lane;
      int d = -300;
      asm("s_nop 1\n\tv_sub_u32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(e_m));
      const float g = ldexpf(1.f, max(d, -300));
      S.cke[m % kScl][lane] = e_m;
      S.scl[m % kScl][lane] = v4f{g, skip ? g : 0.f, __int_as_float(e_m), __int_as_float(m + 1)};
    };
    for (int m = 1; m < NB; m += 2) {
      const int src = m <= LAG ? 0 : m - LAG;
      while (lds_peek(&S.ckpos) < src + 1) {
        if (lds_peek(&S.done)) return;
      }
      asm volatile("" ::: "memory");
      const float2 c = S.ck[src % kSlots][lane];
      const int e_src = S.cke[src % kScl][lane];
      const float mx = vmax(c.x, c.y);
      const int k2 = __builtin_amdgcn_frexp_expf(mx);
      const int own = mx > 0.f ? e_src + k2 : kEmptyE;
      // trend of this lane's exponent over the last two blocks, extrapolated over the lag (bounded: a prediction that
      // is too LOW only leaves a large mantissa, one that is too high flushes the state)
      int d2 = (own > kEmptyE && own_prev > kEmptyE && m > LAG) ? own - own_prev : 0;
      const int lag0 = m - src;
      const int p0 = own > kEmptyE ? own + min(max((d2 * lag0) >> 1, -80), 40) : kEmptyE;
      const int p1 = own > kEmptyE ? own + min(max((d2 * (lag0 + 1)) >> 1, -100), 50) : kEmptyE;
      emit(m, p0);
      if (m + 1 < NB) emit(m + 1, p1);
      own_prev = own;
    }
    return;
  }
  if (wave != 0) {
    while (lds_peek(&S.done) == 0) __builtin_amdgcn_s_sleep(4);
    return;
  }
  // ------------------------------------------------------------------ the chain
  __builtin_amdgcn_s_setprio(3);
  v2f P = {lane == 0 ? 1.f : 0.f, 0.f};
  while (lds_peek(&S.staged[0]) != 1) {}
  asm volatile("" ::: "memory");
  v4f fa[kBlk / 2], fz[kBlk / 2];
#pragma unroll
  for (int j = 0; j < kBlk / 2; ++j) fa[j] = S.ring[0][j][lane];
  v4f sc = S.scl[0][lane];
  while (__float_as_int(sc.w) != 1) sc = S.scl[0][lane];
  int sflag = lds_peek(&S.staged[1]);
  long long waited = 0;
  int nslow = 0, nslow_scl = 0;
  float eprev = __int_as_float(0);
  // LDS byte addresses of this lane's entries
  const unsigned ring0 = (unsigned)(uintptr_t)(lds_int_t*)(int*)&S.ring[0][0][lane];
  const unsigned scl0 = (unsigned)(uintptr_t)(lds_int_t*)(int*)&S.scl[0][lane];
  const unsigned ck0 = (unsigned)(uintptr_t)(lds_int_t*)(int*)&S.ck[0][lane];
  const unsigned stg0 = (unsigned)(uintptr_t)(lds_int_t*)(int*)&S.staged[0];
  const unsigned pos0 = (unsigned)(uintptr_t)(lds_int_t*)(int*)&S.chainpos;
  const unsigned ckp0 = (unsigned)(uintptr_t)(lds_int_t*)(int*)&S.ckpos;
  int s0 = 0, s1 = 1, s2 = 2;
  // One block: scale, checkpoint, 16 frames; the reads of block n + 1's factors (ring slot s1), of its scale entry and of
  // the staged flag of block n + 2 are issued between the frames.  Outputs land in `fn`, `scn`, `sfn` -- NOT tracked by the
  // compiler's wait counts: the next block starts with s_waitcnt lgkmcnt(0).
  auto block = [&](int n, const v4f (&fc)[kBlk / 2], v4f (&fn)[kBlk / 2], v4f& sccur, int& sf) {
    // (everything read during the previous block has arrived)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sccur), "+v"(sf) ::"memory");
    if ((CH & 2) && !(CH & 8) && !(CH & 32)) {
      const int t1 = __builtin_amdgcn_readfirstlane(__float_as_int(sccur.w)), t2 = __builtin_amdgcn_readfirstlane(sf);
      if (__builtin_expect((t1 != n + 1) | (t2 != n + 2), 0)) {
        const long long w0 = clock64();
        ++nslow;
        while (__float_as_int(sccur.w) != n + 1) { asm volatile("" ::: "memory"); sccur = *(volatile v4f*)&S.scl[n % kScl][lane]; ++nslow_scl; }
        while (lds_peek(&S.staged[s1]) != n + 2) {}
        waited += clock64() - w0;
      }
    }
    const v2f G = {sccur.x, sccur.y};
    const unsigned ra = ring0 + (unsigned)s1 * (unsigned)sizeof(S.ring[0]);
    const unsigned sa = scl0 + (unsigned)((n + 1) & (kScl - 1)) * (unsigned)sizeof(S.scl[0]);
    const unsigned ca = ck0 + (unsigned)s0 * (unsigned)sizeof(S.ck[0]);
    const unsigned fa2 = stg0 + (unsigned)s2 * 4u;
    v4f scn;
    int sfn;
    const int posv = n + 2, ckv = n + 1;
#define ON(x) x
#define OFF(x) ""
#define LEAN_ASM(RF, RS, WR)                                                                                                    \
    asm volatile(                                                                                                               \
        "v_sub_u32 v6, %[ep], %[em]\n\t"                                                                                       \
        "v_ldexp_f32 v2, v2, v6\n\t"                                                                                            \
        "v_ldexp_f32 v3, v3, v6\n\t"                                                                                            \
        WR("ds_write_b64 %[ca], v[2:3]\n\t")                                                                                    \
        WR("ds_write_b32 %[ckp], %[ckv]\n\t")                                                                                   \
        FR_A("%[A0]", "%[A0y]") FR_B("%[B0]", "%[B0y]")                                                                         \
        RF("ds_read_b128 %[N0], %[ra]\n\t")                                                                                     \
        FR_A("%[A1]", "%[A1y]") FR_B("%[B1]", "%[B1y]")                                                                         \
        RF("ds_read_b128 %[N1], %[ra] offset:1024\n\t")                                                                         \
        FR_A("%[A2]", "%[A2y]") FR_B("%[B2]", "%[B2y]")                                                                         \
        RF("ds_read_b128 %[N2], %[ra] offset:2048\n\t")                                                                         \
        FR_A("%[A3]", "%[A3y]") FR_B("%[B3]", "%[B3y]")                                                                         \
        RF("ds_read_b128 %[N3], %[ra] offset:3072\n\t")                                                                         \
        FR_A("%[A4]", "%[A4y]") FR_B("%[B4]", "%[B4y]")                                                                         \
        RF("ds_read_b128 %[N4], %[ra] offset:4096\n\t")                                                                         \
        FR_A("%[A5]", "%[A5y]") FR_B("%[B5]", "%[B5y]")                                                                         \
        RF("ds_read_b128 %[N5], %[ra] offset:5120\n\t")                                                                         \
        RS("ds_read_b32 %[sfn], %[fa2]\n\t")                                                                                    \
        FR_A("%[A6]", "%[A6y]") FR_B("%[B6]", "%[B6y]")                                                                         \
        RF("ds_read_b128 %[N6], %[ra] offset:6144\n\t")                                                                         \
        FR_A("%[A7]", "%[A7y]")                                                                                                 \
        RF("ds_read_b128 %[N7], %[ra] offset:7168\n\t")                                                                         \
        RS("ds_read_b128 %[scn], %[sa]\n\t")                                                                                    \
        FR_B("%[B7]", "%[B7y]")                                                                                                 \
        WR("ds_write_b32 %[pos], %[posv]\n\t")                                                                                  \
        : "+{v[2:3]}"(P), [N0] "=&v"(fn[0]), [N1] "=&v"(fn[1]), [N2] "=&v"(fn[2]), [N3] "=&v"(fn[3]), [N4] "=&v"(fn[4]),       \
          [N5] "=&v"(fn[5]), [N6] "=&v"(fn[6]), [N7] "=&v"(fn[7]), [scn] "=&v"(scn), [sfn] "=&v"(sfn)                          \
        : [G] "v"(G), [ep] "v"(eprev), [em] "v"(sccur.z), [ca] "v"(ca), [ra] "v"(ra), [sa] "v"(sa), [fa2] "v"(fa2), [pos] "v"(pos0), [posv] "v"(posv), \
          [ckp] "v"(ckp0), [ckv] "v"(ckv),                                                                                      \
          FIN(0), FIN(1), FIN(2), FIN(3), FIN(4), FIN(5), FIN(6), FIN(7)                                                        \
        : "v4", "v5", "v6", "v7", "memory")
#define FIN(k)                                                                                                                  \
  [A##k] "v"(v2f{fc[k].x, fc[k].y}), [A##k##y] "v"(fc[k].y), [B##k] "v"(v2f{fc[k].z, fc[k].w}), [B##k##y] "v"(fc[k].w)
    if constexpr ((CH & 7) == 7) LEAN_ASM(ON, ON, ON);
    else if constexpr ((CH & 7) == 6) LEAN_ASM(OFF, ON, ON);
    else if constexpr ((CH & 7) == 5) LEAN_ASM(ON, OFF, ON);
    else if constexpr ((CH & 7) == 3) LEAN_ASM(ON, ON, OFF);
    else if constexpr ((CH & 7) == 4) LEAN_ASM(OFF, OFF, ON);
    else if constexpr ((CH & 7) == 1) LEAN_ASM(ON, OFF, OFF);
    else LEAN_ASM(OFF, OFF, OFF);
    if (!(CH & 2)) scn = sccur, sfn = sf;
    eprev = sccur.z;
    sccur = scn;
    sf = sfn;
    s0 = s1, s1 = s2, s2 = s2 + 1 == kSlots ? 0 : s2 + 1;
  };
  long long t0 = 0, w0 = 0;
  block(0, fa, fz, sc, sflag);
  for (int kk = 1; kk + 3 < NB; kk += 2) {
    if (kk == 17) t0 = clock64(), w0 = wall_clock64(), waited = 0;
    block(kk, fz, fa, sc, sflag);
    block(kk + 1, fa, fz, sc, sflag);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  lds_post(&S.chainpos, NB + 8);
  lds_post(&S.done, 1);
  out[blockIdx.x * 64 + lane] = P.x + P.y;
  if (lane == 0) cyc[blockIdx.x * 4] = t1 - t0, cyc[blockIdx.x * 4 + 1] = w1 - w0, cyc[blockIdx.x * 4 + 2] = waited;
  if (lane == 0) cyc[blockIdx.x * 4 + 3] = ((long long)nslow << 32) | nslow_scl;
}

template <int PIPE, int LAG, int CH = 7>
void runp(const char* what) {
  const int NB = 256, WG = 256, C = 100, L = 44;
  float* out; long long* cyc; float* x; int* cols; unsigned long long* pub;
  const size_t xn = (size_t)WG * NB * kBlk * C;
  CK(hipMalloc(&out, WG * 64 * 4)); CK(hipMalloc(&cyc, WG * 32)); CK(hipMalloc(&x, xn * 4)); CK(hipMalloc(&cols, WG * 64 * 4));
  CK(hipMalloc(&pub, (size_t)WG * NB * 128 * 8));
  CK(hipMemset(out, 0, WG * 64 * 4));
  CK(hipMemset(cyc, 0, WG * 32));
  {
    std::vector<float> hx(xn);
    unsigned s = 12345u;
    for (size_t i = 0; i < xn; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) & 0xffff) / 65536.f * 4.f - 2.f; }
    CK(hipMemcpy(x, hx.data(), xn * 4, hipMemcpyHostToDevice));
  }
  std::vector<int> hc(WG * 64);
  for (size_t i = 0; i < hc.size(); ++i) hc[i] = (int)((i * 2654435761u >> 8) % (unsigned)(C - 2));
  for (int wgi = 0; wgi < WG; ++wgi) for (int l = L; l < 64; ++l) hc[wgi * 64 + l] = C - 1;
  CK(hipMemcpy(cols, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)kp<PIPE, LAG, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PLds)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  kp<PIPE, LAG, CH><<<WG, 1024, sizeof(PLds)>>>(NB, L, C, x, cols, out, cyc, pub);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  kp<PIPE, LAG, CH><<<WG, 1024, sizeof(PLds)>>>(NB, L, C, x, cols, out, cyc, pub);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<long long> h(WG * 4); CK(hipMemcpy(h.data(), cyc, WG * 32, hipMemcpyDeviceToHost));
  std::vector<float> o(WG * 64); CK(hipMemcpy(o.data(), out, WG * 64 * 4, hipMemcpyDeviceToHost));
  double c = 0, w = 0, wt = 0, lt = 0, ls = 0; for (int i = 0; i < WG; ++i) c += h[4 * i], w += h[4 * i + 1], wt += h[4 * i + 2], lt += h[4 * i + 3] >> 32, ls += h[4 * i + 3] & 0xffffffff;
  c /= WG; w /= WG; wt /= WG; lt /= WG;
  const int nb = NB - 3 - 17;
  printf("%-60s | kernel %8.1f us | clock64/block %7.1f (waiting %6.1f) | wall ns/block %7.1f | slow paths %5.1f%% of blocks, scale re-reads %5.1f / 100 blocks | out %g\n",
         what, ms * 1e3, c / nb, wt / nb, w * 10.0 / nb, 100.0 * lt / WG / (NB - 1), 100.0 * ls / WG / (NB - 1), o[45]);
  fflush(stdout);
  CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(x)); CK(hipFree(cols)); CK(hipFree(pub));
}

int main() {
  runp<6, 2, 7>("lean2: no gathers, compute + stores, lags 2,3");
  runp<6, 3, 7>("lean2: no gathers, compute + stores, lags 3,4");
  runp<6, 3, 32 + 7>("lean2: same, chain does not check its flags");
  runp<6 + 64, 3, 7>("lean2: lags 3,4, helper on the chain's SIMD");
  runp<7, 3, 7>("lean2: everything, lags 3,4");
  runp<7, 2, 7>("lean2: everything, lags 2,3");
  return 0;
}

// Dense (fully connected) transitions for ANY class count (included by dense_kernels.hip, inside namespace wfl).
//
// criterions/asg.py:198-199 sizes `transitions` to any N; the kernels above keep the (N+1) x N matrix on chip (registers:
// N <= 192 -- the log semiring's sweeps, N <= 256 -- Viterbi; LDS: N <= 195 -- the log-domain launches).  Beyond that the matrix is 4 N^2 bytes -- 4 MB at N = 1000 -- and cannot be private
// to a workgroup; but then the per-frame update of ALL utterances,
//     alpha_t[b][i] = E_t[b][i] * sum_j P[i][j] alpha_{t-1}[b][j]        (beta: the transpose)
// is a [N x N] x [N x B] matrix product that is worth sharing: the frame is ONE launch for the whole batch (both
// sweeps: alpha's frame s and beta's frame T-1-s), tiled over (states, utterances) with P streamed from L2 / MALL
// once per 64 utterances instead of once per utterance.  fp32 FMAs on the vector pipe (fp32 MFMA has the same peak
// on gfx950; the tiles are small and the launch is latency-bound at N ~ 1000, B ~ 128).
//
//   P[i][j]   = exp(W[1+i][j] - rm_i),  rm_i = max_j W[1+i][j]                               in (0, 1]
//   e_t[b][i] = exp(x[b,t,i] + rm_i - mxp_t[b]),  mxp_t[b] = max_i (x[b,t,i] + rm_i)          in (0, 1]
//   araw_t    = e_t (.) (P (araw_{t-1} / maxa_{t-1})),  maxa_t[b] = max_i araw_t[b][i]        (atomic max: >= 0)
//   alpha_t   = araw_t / maxa_t * exp(cuma_t),  cuma_t = cuma_{t-1} + mxp_t + log maxa_t,  cuma_0 = max_i(x_0 + W_0)
//   braw_t    = P^T (e_{t+1} (.) braw_{t+1} / maxb_{t+1}),  beta_t = braw_t / maxb_t * exp(cb_t),
//               cb_t = cb_{t+1} + mxp_{t+1} + log maxb_t,  braw_{T-1} = 1, cb_{T-1} = 0
// The normalisation lags the product by one frame (a frame's maximum is only complete when its launch ends), so a
// stored value is at most N times / at least 2^-126 of the frame's largest: fp32 holds it.  The cumulative offsets
// are doubles (they reach 10^4 at T = 1000).  `alpha` / `beta` hold araw / braw; the workspace the rest.
//
// Gradient: emission posteriors elementwise; the transition gradient is the second matrix product of the path,
//     dW[1+i][j] = P[i][j] * sum_{b, t >= 1} U[(b,t)][i] V[(b,t)][j],   V = araw_{t-1} / maxa_{t-1},
//     U = e_t (.) braw_t / maxb_t * coef_w[b] * exp(cuma_{t-1} + mxp_t + cb_t - log Z_b)
// over K = B (T-1) rows, split into wide_split(C) slabs reduced in a fixed order (deterministic, no atomics).
// Tropical semiring (ASG.viterbi, asg.py:217-226): the same tiling with (max, +) and the arg max (lowest previous
// label on ties, as the LDS-resident kernel).
#pragma once

constexpr int kWideTM = 64, kWideTN = 64, kWideTK = 16;  // output tile (states x utterances) and K chunk
// K slabs of the transition-gradient product: as many as bring the launch to ~2048 workgroups (between 8 and 128).  A
// workgroup walks its slab in chunks of 16 rows behind three barriers each -- ~4 us a chunk, measured -- so the launch
// takes (rows per slab / 16) chunks whatever the class count: with 8 slabs, C = 200 ... 320 at B = 128, T = 1000 were 128 ... 200
// workgroups of 1000 chunks (4.1 ms, four times the sweeps); the slabs cost 4 C^2 bytes each (C = 1000: 8, as before).
__host__ __device__ inline int wide_split(int C) {
  const int tiles = ((C + kWideTM - 1) / kWideTM) * ((C + kWideTN - 1) / kWideTN);
  const int s = 2048 / (tiles > 0 ? tiles : 1);
  return s < 8 ? 8 : s > 128 ? 128 : s;
}

struct WideWs {
  float* P;      // [C][C]
  float* PT;     // [C][C]
  float* rm;     // [C]
  float* m0;     // [B]
  float* mxp;    // [B][T]
  float* maxa;   // [B][T]
  float* maxb;   // [B][T]
  double* cuma;  // [B][T]
  double* cb;    // [B][T]
};
__host__ __device__ inline size_t wide_align(size_t n) { return (n + 15) & ~(size_t)15; }
__host__ __device__ inline WideWs wide_carve(void* ws, int B, int T, int C) {
  WideWs w;
  char* p = (char*)ws;
  w.P = (float*)p, p += wide_align((size_t)4 * C * C);
  w.PT = (float*)p, p += wide_align((size_t)4 * C * C);
  w.rm = (float*)p, p += wide_align((size_t)4 * C);
  w.m0 = (float*)p, p += wide_align((size_t)4 * B);
  w.mxp = (float*)p, p += wide_align((size_t)4 * B * T);
  w.maxa = (float*)p, p += wide_align((size_t)4 * B * T);
  w.maxb = (float*)p, p += wide_align((size_t)4 * B * T);
  w.cuma = (double*)p, p += wide_align((size_t)8 * B * T);
  w.cb = (double*)p, p += wide_align((size_t)8 * B * T);
  return w;
}
static size_t wide_ws_bytes(int B, int T, int C) {
  return 2 * wide_align((size_t)4 * C * C) + wide_align((size_t)4 * C) + wide_align((size_t)4 * B) +
         3 * wide_align((size_t)4 * B * T) + 2 * wide_align((size_t)8 * B * T) + 64;
}

__device__ __forceinline__ float wide_clean(float v) { return (v == v) ? v : WFL_NEG_INF; }  // NaN policy: impossible

// one workgroup per row i of W[1:, :]: rm_i, P[i][:], PT[:][i]
__global__ void __launch_bounds__(256) wide_prep_kernel(const float* __restrict__ W, int C, WideWs w) {
  __shared__ float red[64];
  const int i = blockIdx.x;
  const float* row = W + (int64_t)(1 + i) * C;
  float m = WFL_NEG_INF;
  for (int j = threadIdx.x; j < C; j += 256) m = fmaxf(m, wide_clean(row[j]));
  m = blk_max(m, red);
  const float ref = (m > -3.0e38f && m < 3.0e38f) ? m : 0.f;  // (a row of -inf: P = 0, any finite reference does)
  if (threadIdx.x == 0) w.rm[i] = ref;
  for (int j = threadIdx.x; j < C; j += 256) {
    const float p = __expf(wide_clean(row[j]) - ref);
    w.P[(int64_t)i * C + j] = p;
    w.PT[(int64_t)j * C + i] = p;
  }
}

// one wave per (b, t): mxp; t == 0: m0 and alpha's first frame; t == T-1: beta's last frame.  The maxima of the
// frames in between are zeroed for the frame launches' atomic max.
__global__ void __launch_bounds__(256) wide_rows_kernel(const float* __restrict__ x, const float* __restrict__ W, int B, int T,
                                                        int C, WideWs w, float* __restrict__ alpha, float* __restrict__ beta) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)B * T) return;
  const int b = (int)(r / T), t = (int)(r % T);
  const float* xr = x + r * C;
  float m = WFL_NEG_INF;
  for (int i = lane; i < C; i += 64) m = fmaxf(m, wide_clean(xr[i]) + w.rm[i]);
  m = wave_max(m);
  const float ref = (m > -3.0e38f && m < 3.0e38f) ? m : 0.f;
  if (lane == 0) w.mxp[r] = ref;
  if (t == 0) {
    float m0 = WFL_NEG_INF;
    for (int i = lane; i < C; i += 64) m0 = fmaxf(m0, wide_clean(xr[i]) + wide_clean(W[i]));
    m0 = wave_max(m0);
    const float r0 = (m0 > -3.0e38f && m0 < 3.0e38f) ? m0 : 0.f;
    float top = 0.f;
    for (int i = lane; i < C; i += 64) {
      const float a = __expf(wide_clean(xr[i]) + wide_clean(W[i]) - r0);
      alpha[r * C + i] = a;
      top = fmaxf(top, a);
    }
    top = wave_max(top);
    if (lane == 0) w.m0[b] = r0, w.maxa[r] = top;
  } else if (lane == 0) {
    w.maxa[r] = 0.f;
  }
  if (beta) {
    if (t == T - 1) {
      for (int i = lane; i < C; i += 64) beta[r * C + i] = 1.f;
      if (lane == 0) w.maxb[r] = 1.f;
    } else if (lane == 0) {
      w.maxb[r] = 0.f;
    }
  }
}

__device__ __forceinline__ void atomic_max_pos(float* p, float v) {  // v >= 0: the bit patterns order like the values
  atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
}

// One frame of both sweeps (blockIdx.z = 0 alpha's frame s (s >= 1), 1 beta's frame T-1-s) on the matrix cores, tiled so
// that the launch fills the chip.  (Round 3's vector-pipe kernel -- deleted in round 5 -- made 64 x 64 output tiles: at
// N = 1000, B = 32 that was 16 x 1 x 2 = 32 workgroups on 256 CUs, each walking K = 1000 by itself: 156 us per frame,
// 39 ms per step at T = 250.)  Here a workgroup owns a 16-state x 16-utterance tile
// (v_mfma_f32_16x16x4_f32), its four waves each take a quarter of K and the partial tiles meet in LDS: 63 x 2 x 2 = 252
// workgroups at that shape, ~250 K steps of dependent products per wave instead of 1000 k-chunks behind barriers.
// Operands straight from L2 into the products' registers (P, 4 MB, stays in L2 / MALL across the frame's workgroups): a
// lane loads four consecutive k of its row -- lanes l and l + 16 j of a product then hold k = kb + 4 j + c for the c-th of
// four products, the same four k on both operands, which is all the instruction asks for.
typedef float wide_v4f __attribute__((ext_vector_type(4)));
constexpr int kWideMfmaWaves = 8;
__global__ void __launch_bounds__(64 * kWideMfmaWaves) wide_frame_mfma_kernel(const float* __restrict__ x, int B, int T, int C, int s,
                                                                             WideWs w, float* __restrict__ alpha,
                                                                             float* __restrict__ beta) {
  const int dir = blockIdx.z;
  if (dir == 1 && !beta) return;
  __shared__ float part[kWideMfmaWaves - 1][64][4];
  const int i0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const int t = dir == 0 ? s : T - 1 - s;   // the frame being produced
  const int tp = dir == 0 ? t - 1 : t + 1;  // the frame it is produced from
  const float* A = dir == 0 ? w.P : w.PT;
  const float* vec = dir == 0 ? alpha : beta;
  const float* vmax = dir == 0 ? w.maxa : w.maxb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;  // operand row (state / utterance) and k group of this lane
  const int ai = min(i0 + lr, C - 1), bn = min(n0 + lr, B - 1);
  const bool a_ok = i0 + lr < C, b_ok = n0 + lr < B;
  // A workgroup is one dependent chain -- loads, products, reduction, store -- with nothing else on its CU to hide a
  // round trip behind (252 workgroups at N = 1000, B = 32): EVERY load of the chain is issued up front, the operands of
  // all of a wave's K steps and what the epilogue multiplies with, so that the chain pays one round trip, not five.
  const float mx = vmax[(int64_t)bn * T + tp];   // the utterance's scale (lane's column of the OUTPUT tile too)
  const float bref = w.mxp[(int64_t)bn * T + tp];
  const int64_t orow = (int64_t)bn * T + t;
  const float oref = dir == 0 ? w.mxp[orow] : 0.f;
  float ox[4], orm[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int i = min(i0 + 4 * lg + v, C - 1);
    ox[v] = dir == 0 ? x[orow * C + i] : 0.f;
    orm[v] = dir == 0 ? w.rm[i] : 0.f;
  }
  const float* arow = A + (int64_t)ai * C;
  const float* brow = vec + ((int64_t)bn * T + tp) * C;
  const float* xrow = x + ((int64_t)bn * T + tp) * C;
  const int nsteps = (C + 15) / 16;  // K in steps of 16 (four products); the waves take interleaved steps
  const bool vec4 = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(vec) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(A)) & 15) == 0;
  auto load4 = [&](const float* row, int k) -> wide_v4f {  // row[k .. k+3], zeros beyond C
    if (vec4) {
      if (k + 3 < C) {
        const float4 q = *reinterpret_cast<const float4*>(row + k);
        return wide_v4f{q.x, q.y, q.z, q.w};
      }
      return wide_v4f{0.f, 0.f, 0.f, 0.f};  // (C % 4 == 0: a group of four is inside or outside)
    }
    wide_v4f r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = k + c < C ? row[k + c] : 0.f;
    return r;
  };
  wide_v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // (two chains of dependent products)
  constexpr int kU = 8;  // steps per trip: one trip up to N = 1024
  for (int st0 = wave; st0 < nsteps; st0 += kWideMfmaWaves * kU) {
    wide_v4f av[kU], bv[kU], xv[kU], rv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int k = (st0 + kWideMfmaWaves * u) * 16 + 4 * lg;  // (steps past the end: k >= C, zeros)
      av[u] = load4(arow, k), bv[u] = load4(brow, k);
      if (dir == 1) xv[u] = load4(xrow, k), rv[u] = load4(w.rm, k);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int k = (st0 + kWideMfmaWaves * u) * 16 + 4 * lg;
      if (!a_ok) av[u] = wide_v4f{0.f, 0.f, 0.f, 0.f};
      if (dir == 1) {  // e_{t+1} (.) beta_{t+1}
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[u][c] *= k + c < C ? __expf(wide_clean(xv[u][c]) + rv[u][c] - bref) : 0.f;
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bv[u][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], bv[u][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][2], bv[u][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][3], bv[u][3], acc1, 0, 0, 0);
    }
  }
  wide_v4f acc = acc0 + acc1;
  // acc[v]: state i0 + 4 lg + v, utterance n0 + lr.  The other waves hand their partial tiles to wave 0.
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) part[wave - 1][lane][v] = acc[v];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int q = 0; q < kWideMfmaWaves - 1; ++q)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] += part[q][lane][v];
  const float binv = (b_ok && mx > 0.f) ? 1.f / mx : 0.f;  // (the operand's scale, applied to the lane's output column)
  float top = 0.f;
  if (b_ok) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + 4 * lg + v;
      if (i >= C) continue;
      float y = acc[v] * binv;
      if (dir == 0) y *= __expf(wide_clean(ox[v]) + orm[v] - oref);
      (dir == 0 ? alpha : beta)[orow * C + i] = y;
      top = fmaxf(top, y);
    }
  }
  // (lanes lr, lr + 16, lr + 32, lr + 48 share the utterance)
  top = fmaxf(top, __shfl_xor(top, 16, 64));
  top = fmaxf(top, __shfl_xor(top, 32, 64));
  if (lg == 0 && b_ok && top > 0.f) atomic_max_pos((dir == 0 ? w.maxa : w.maxb) + orow, top);
}

// ------------------------------------------------------------------------------------------------------------------
// The frames of ONE utterance's sweep in ONE workgroup, the matrix in its registers: the step for class counts just
// beyond what dense_fast_chain_kernel keeps on chip (193 .. 320).  There the per-frame launch above is a poor fit: the
// product of a frame is only ~10 MFLOP for the whole batch, a launch is ~7 us of stream time whatever it computes, and a
// sweep is T of them (C = 200, T = 1000, B = 128: 14.9 ms fwd + bwd against 1.1 ms at C = 190).  But 4 C^2 bytes
// (160 KB at C = 200, 410 KB at C = 320) still fit ONE compute unit's register file (512 KB): grid (B, 2), 1024 threads,
// a group of 16 lanes owns NR consecutive rows of P (P^T for beta), a lane NR float4 column chunks of each
// (P[row0 + r][64 k + 4 l .. + 3]: NR * NR * 4 <= 100 registers), the frame vector ping-pongs in LDS.  A frame:
//     NR 16-byte LDS reads of the vector (the wave's four groups read the same addresses: broadcast), 4 NR^2
//     multiply-adds, NR all-reductions over the group's 16 lanes (four DPP adds each), the emission factor, one
//     LDS-only barrier.
// Same recurrence, same stored format as the per-frame launches (araw / braw and their per-frame divisors in
// maxa / maxb: the header of this file), so wide_rows_kernel in front and wide_scan_kernel / wide_grad behind are
// unchanged.  The divisor of a frame is the largest entry of the frame before it, exactly as there: the waves' lanes
// fold theirs into one LDS word per frame with ds_max (four words in rotation: written in frame n, read in n + 1,
// cleared in n + 2).
// Nothing in the frame is conditional: every lane takes part in loads and stores -- lane l of a group works on row
// l % NR (l / 4 for NR = 4), so up to four lanes compute, and store, the same value -- because emission scores are requested kWideAhead frames
// ahead and loads and stores share ONE in-order counter (vmcnt): behind a test the compiler's static wait-count pass
// must assume the path that skipped the later loads, and waits for the youngest of them instead of the oldest
// (csrc/lattice_kernels.hip, the comment above the chunk loop of run_chain_prob, has the long version).
// ------------------------------------------------------------------------------------------------------------------
template <int NR>
constexpr int wide_ahead() { return NR <= 4 ? 8 : 4; }  // emission scores in flight per lane (frames)
constexpr int kWideResidentMaxT = 8192;  // the utterance's row references are staged in LDS (4 T bytes)
__host__ __device__ inline int wide_resident_rows(int C) { return C <= 256 ? 4 : C <= 320 ? 5 : 0; }  // NR (0: does not fit)
static size_t wide_resident_lds(int NR, int T) { return (size_t)2 * 64 * NR * 4 + 16 + (size_t)8 * T; }

__device__ __forceinline__ float group16_sum(float v) {  // every lane of a row of 16 receives the row's sum
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

__device__ __forceinline__ float group16_max(float v) {  // v >= 0; every lane of a row of 16 receives the row's maximum
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true)));
  return v;
}
typedef float wide_v2f __attribute__((ext_vector_type(2)));
// Four rows' sums over a row of 16 lanes, scattered: lane i ends with the total of row i / 4 (its quad's row).  A
// reduce-scatter over the cross-lane network -- every step halves what a lane still carries -- in 11 instructions where
// four all-reductions take 29: mirror (i <-> 15 - i: lanes 0-7 keep rows 0, 1, lanes 8-15 rows 2, 3), half mirror
// (i <-> 7 - i in each half: the lower quad of a half keeps its first row, the upper quad the second), then the quad's
// four lanes add up.  (Lane 0 of the row collects lanes {0, 15, 7, 8}, lane 1 {1, 14, 6, 9}, ...: every lane once.)
__device__ __forceinline__ float group16_scatter4(const float (&t)[4], int l16) {
  auto dpp = [](float v, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  const bool hi = (l16 & 8) != 0, mid = (l16 & 4) != 0;
  const float ka = hi ? t[2] : t[0], kb = hi ? t[3] : t[1], sa = hi ? t[0] : t[2], sb = hi ? t[1] : t[3];
  const float ra = ka + dpp(sa, std::integral_constant<int, 0x140>{});
  const float rb = kb + dpp(sb, std::integral_constant<int, 0x140>{});
  float r = (mid ? rb : ra) + dpp(mid ? ra : rb, std::integral_constant<int, 0x141>{});
  r += dpp(r, std::integral_constant<int, 0xB1>{});
  r += dpp(r, std::integral_constant<int, 0x4E>{});
  return r;
}

template <int NR>
__global__ void __launch_bounds__(1024) wide_resident_sweep_kernel(const float* __restrict__ x, int B, int T, int C, WideWs w,
                                                                   float* __restrict__ alpha, float* __restrict__ beta) {
  extern __shared__ __attribute__((aligned(16))) char wide_smem[];
  constexpr int CP = 64 * NR;
  constexpr bool PK = NR <= 4;  // packed multiply-adds
  float* vbuf = reinterpret_cast<float*>(wide_smem);           // [2][CP] frame vector, ping-pong
  int* dmax = reinterpret_cast<int*>(vbuf + 2 * CP);           // [4] largest entry of a frame, as bits
  float* mxp_l = reinterpret_cast<float*>(dmax + 4);           // [T]
  float* dlog = mxp_l + T;                                     // [T] the frames' divisors (stored to maxa / maxb at the end)
  const int b = blockIdx.x, dir = blockIdx.y;
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  const int row0 = NR * grp;
  const float* A = dir == 0 ? w.P : w.PT;
  float* out = dir == 0 ? alpha : beta;
  float* dvec = dir == 0 ? w.maxa : w.maxb;
  const float* xb = x + (int64_t)b * T * C;
  const int64_t bt = (int64_t)b * T;
  // ---- the lane's share of the matrix (every load first, from clamped addresses; masked in a second pass: with the test
  // next to the load the compiler sinks each load under its test -- 4 NR^2 dependent L2 round trips in a row)
  float Pl[NR][NR][4];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        Pl[r][k][c] = A[(int64_t)min(row0 + r, C - 1) * C + min(64 * k + 4 * l16 + c, C - 1)];
  asm volatile("" ::: "memory");
  // (pairs of columns where the frame's multiply-adds are packed -- v_pk_fma_f32, two per instruction at the same issue
  // cost; plain floats for NR = 5, which has no room for register pairs)
  wide_v2f Pr[PK ? NR : 1][PK ? NR : 1][2];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = (row0 + r < C && 64 * k + 4 * l16 + c < C) ? Pl[r][k][c] : 0.f;
        if constexpr (PK)
          Pr[r][k][c >> 1][c & 1] = v;
        else
          Pl[r][k][c] = v;
      }
  // ---- the row this lane finishes: l16 % NR of the group's (lanes NR .. 15 duplicate lanes 0 .. NR-1: same value to the
  // same address).  A lane whose row does not exist (rows C .. 64 NR - 1) stores into the sweep's LAST frame instead, at a
  // column of its own -- that frame is written for good by the last step, behind a wait for every earlier store
  const int rsel = NR == 4 ? l16 >> 2 : l16 % NR;  // (NR = 4: the quad's row, where group16_scatter4 leaves the totals)
  const int myrow = row0 + rsel;
  const bool live = myrow < C;
  const int rowc = min(myrow, C - 1);
  const float rm = w.rm[rowc];
  for (int t = tid; t < T; t += 1024) mxp_l[t] = w.mxp[bt + t];
  const int t0 = dir == 0 ? 0 : T - 1;
  for (int i = tid; i < 2 * CP; i += 1024) {
    float v = 0.f;
    if (i < C) {
      v = out[(bt + t0) * C + i];
      if (dir == 1) v *= __expf(wide_clean(xb[(int64_t)t0 * C + i]) + w.rm[i] - w.mxp[bt + t0]);
    }
    vbuf[i] = v;
  }
  if (tid == 0) dmax[0] = __float_as_int(dvec[bt + t0]), dmax[1] = 0, dmax[2] = 0, dmax[3] = 0;
  float* dump = out + (bt + (dir == 0 ? T - 1 : 0)) * C + max(myrow - C, 0);  // (see `live` above)
  // where the lane stores (one row further every step; the parked lanes stay where they are) and where its scores come from
  const int64_t fstep = dir == 0 ? (int64_t)C : -(int64_t)C;
  float* po = live ? out + (bt + (dir == 0 ? 1 : T - 2)) * C + myrow : dump;
  const int64_t pstep = live ? fstep : 0;
  // step n (1 .. T-1) produces frame tf(n) from frame tf(n - 1)
  auto tf = [&](int n) { return dir == 0 ? n : T - 1 - n; };
  constexpr int kWideAhead = wide_ahead<NR>();
  float xr[kWideAhead];
#pragma unroll
  for (int u = 0; u < kWideAhead; ++u) xr[u] = xb[(int64_t)tf(min(1 + u, T - 1)) * C + rowc];
  __syncthreads();
  auto step = [&](int n, float xv, auto last_) {
    constexpr bool LAST = decltype(last_)::value;  // the sweep's last step: only lanes with a row store (see `dump`)
    const int t = tf(n), tp = tf(n - 1);
    const float* vc = vbuf + ((n - 1) & 1) * CP;
    float* vn = vbuf + (n & 1) * CP;
    const float dprev = __int_as_float(dmax[(n - 1) & 3]);
    const float mx = mxp_l[t];
    // (one accumulator per row: NR independent chains already; a second one per row costs the NR = 5 instantiation
    // registers it does not have -- 100 of its 128 hold the matrix)
    float tot[NR];
    if constexpr (PK) {
      wide_v2f acc[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] = wide_v2f{0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const float4 v4 = *reinterpret_cast<const float4*>(vc + 64 * k + 4 * l16);
        const wide_v2f lo = {v4.x, v4.y}, hi = {v4.z, v4.w};
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          acc[r] = __builtin_elementwise_fma(Pr[r][k][0], lo, acc[r]);
          acc[r] = __builtin_elementwise_fma(Pr[r][k][1], hi, acc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) tot[r] = acc[r][0] + acc[r][1];
    } else {
      // (one scalar accumulator per row: NR independent chains; the NR = 5 instantiation has no registers for more -- 100
      // of its 128 hold the matrix, and register PAIRS for the packed form cost it 60 spills)
#pragma unroll
      for (int r = 0; r < NR; ++r) tot[r] = 0.f;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const float4 v4 = *reinterpret_cast<const float4*>(vc + 64 * k + 4 * l16);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          tot[r] = fmaf(Pl[r][k][0], v4.x, tot[r]);
          tot[r] = fmaf(Pl[r][k][1], v4.y, tot[r]);
          tot[r] = fmaf(Pl[r][k][2], v4.z, tot[r]);
          tot[r] = fmaf(Pl[r][k][3], v4.w, tot[r]);
        }
      }
    }
    float mine = 0.f;
    if constexpr (NR == 4) {
      mine = group16_scatter4(tot, l16);
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float s = group16_sum(tot[r]);
        mine = rsel == r ? s : mine;
      }
    }
    const float inv = dprev > 0.f ? 1.f / dprev : 0.f;  // (a dead utterance stays dead: as the per-frame launch)
    const float e = __expf(wide_clean(xv) + rm - mx);
    const float ys = mine * inv;
    const float glob = live ? (dir == 0 ? ys * e : ys) : 0.f;
    vn[myrow] = live ? ys * e : 0.f;                     // (myrow < CP; rows beyond C stay 0)
    if (!LAST || live) *po = glob;
    po += pstep;
    // the frame's largest entry: one LDS atomic per wave (64 lanes on one LDS word are served one after the other; the
    // wave's own maximum over the cross-lane network and four scalar reads, not six LDS permutes in a row).  Four words in
    // rotation: written in step n, read in n + 1, cleared in n + 2.
    const int gm = __float_as_int(group16_max(glob));    // (glob >= 0: the bit patterns order like the values)
    const int wm = max(max(__builtin_amdgcn_readlane(gm, 0), __builtin_amdgcn_readlane(gm, 16)),
                       max(__builtin_amdgcn_readlane(gm, 32), __builtin_amdgcn_readlane(gm, 48)));
    if ((tid & 63) == 0) atomicMax(&dmax[n & 3], wm);
    if (tid == 0) dmax[(n + 1) & 3] = 0, dlog[tp] = dprev;
    lds_barrier();
  };
  int n = 1;
  // (whole groups whose requests stay inside the utterance: straight-line code, no test between the steps, no clamp)
  const float* px = xb + (int64_t)tf(min(1 + kWideAhead, T - 1)) * C + rowc;  // the scores of step n + kWideAhead
  for (; n + 2 * kWideAhead <= T; n += kWideAhead) {
#pragma unroll
    for (int u = 0; u < kWideAhead; ++u) {
      step(n + u, xr[u], std::false_type{});
      xr[u] = px[u * fstep];
    }
    px += kWideAhead * fstep;
  }
  for (; n + 1 < T; ++n) step(n, xb[(int64_t)tf(n) * C + rowc], std::false_type{});  // (the last steps: their scores requested where they are used)
  if (n < T) {
    // the last frame holds what the lanes without a row parked there: every store so far must have landed before it is written
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    step(n, xb[(int64_t)tf(n) * C + rowc], std::true_type{});
  }
  if (tid == 0) dlog[tf(T - 1)] = __int_as_float(dmax[(T - 1) & 3]);
  __syncthreads();
  for (int t = tid; t < T; t += 1024) dvec[bt + t] = dlog[t];
}

// wide_forward's frames for class counts whose matrix fits one CU's registers (see above); false: not this case
static bool wide_resident_frames(const float* x, int B, int T, int C, const WideWs& w, float* alpha, float* beta, hipStream_t st) {
  const int NR = wide_resident_rows(C);
  static const bool off = [] {
    const char* e = getenv("WFL_DENSE_WIDE_RESIDENT");
    return e && atoi(e) == 0;
  }();
  if (NR == 0 || off || T < 2 || T > kWideResidentMaxT) return false;
  const size_t lds = wide_resident_lds(NR, T);
  const dim3 grid((unsigned)B, beta ? 2u : 1u);
  if (NR == 4)
    hipLaunchKernelGGL(wide_resident_sweep_kernel<4>, grid, dim3(1024), lds, st, x, B, T, C, w, alpha, beta);
  else
    hipLaunchKernelGGL(wide_resident_sweep_kernel<5>, grid, dim3(1024), lds, st, x, B, T, C, w, alpha, beta);
  return true;
}

// per utterance: the cumulative offsets (a serial scan over T by one thread) and log Z
__global__ void __launch_bounds__(256) wide_scan_kernel(int B, int T, int C, WideWs w, const float* __restrict__ alpha,
                                                        bool with_beta, float* __restrict__ logz) {
  __shared__ float red[64];
  const int b = blockIdx.x;
  // (256 threads, a contiguous run of frames each: the run's sum, a serial pass over the 256 sums, the run again -- one
  // thread walking all T frames through dependent loads took 0.26 ms at T = 1000, a fifth of the resident sweeps)
  __shared__ double runs[256];
  const int tid = threadIdx.x;
  const int per = (T - 1 + 255) / 256;  // terms 1 .. T-1 (alpha: frame t; beta: frame T-1-t)
  const int64_t bt = (int64_t)b * T;
  auto term_a = [&](int t) {
    const float mx = w.maxa[bt + t];
    return (double)w.mxp[bt + t] + (mx > 0.f ? (double)__logf(mx) : -1.0e300);
  };
  auto term_b = [&](int t) {  // what frame t adds on the way down from T-1
    const float mx = w.maxb[bt + t];
    return (double)w.mxp[bt + t + 1] + (mx > 0.f ? (double)__logf(mx) : -1.0e300);
  };
  const int lo = 1 + tid * per, hi = min(T, lo + per);
  {
    double sum = 0.0;
    for (int t = lo; t < hi; ++t) sum += term_a(t);
    runs[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      double c = (double)w.m0[b];
      w.cuma[bt] = c;
      for (int i = 0; i < 256; ++i) {
        const double r = runs[i];
        runs[i] = c;
        c += r;
      }
    }
    __syncthreads();
    double c = runs[tid];
    for (int t = lo; t < hi; ++t) c += term_a(t), w.cuma[bt + t] = c;
    __syncthreads();
  }
  if (with_beta) {
    // frame t = T-1-k for k = 1 .. T-1, in the same runs
    double sum = 0.0;
    for (int k = lo; k < hi; ++k) sum += term_b(T - 1 - k);
    runs[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      double d = 0.0;
      w.cb[bt + T - 1] = 0.0;
      for (int i = 0; i < 256; ++i) {
        const double r = runs[i];
        runs[i] = d;
        d += r;
      }
    }
    __syncthreads();
    double d = runs[tid];
    for (int k = lo; k < hi; ++k) d += term_b(T - 1 - k), w.cb[bt + T - 1 - k] = d;
  }
  __syncthreads();
  const float* last = alpha + ((int64_t)b * T + T - 1) * C;
  float sum = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) sum += last[i];
  sum = blk_sum(sum, red);  // (the scan's last barrier orders cuma before the read below)
  if (threadIdx.x == 0) {
    const float mx = w.maxa[(int64_t)b * T + T - 1];
    const double z = (sum > 0.f && mx > 0.f) ? (double)logf(sum / mx) + w.cuma[(int64_t)b * T + T - 1] : -1.0e300;
    logz[b] = z > -1.0e299 ? (float)z : WFL_NEG_INF;
  }
}

// dx[b,t,i] = (accumulate ? dx : 0) + gout * addend + coef[b] * gout * posterior_t(i); one wave per row
__global__ void __launch_bounds__(256) wide_grad_x_kernel(int B, int T, int C, WideWs w, const float* __restrict__ alpha,
                                                          const float* __restrict__ beta, const float* __restrict__ logz,
                                                          const float* __restrict__ coef, const float* __restrict__ gout,
                                                          int accumulate, const float* __restrict__ addend,
                                                          float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)B * T) return;
  const int b = (int)(r / T);
  const float g = gout ? gout[0] : 1.f;
  const float ma = w.maxa[r], mb = w.maxb[r], lz = logz[b];
  float k = 0.f;
  if (ma > 0.f && mb > 0.f && lz > -3.0e38f)
    k = (coef ? coef[b] : 1.f) * g * (float)exp(w.cuma[r] + w.cb[r] - (double)lz) / (ma * mb);
  for (int i = lane; i < C; i += 64) {
    float v = k != 0.f ? k * alpha[r * C + i] * beta[r * C + i] : 0.f;
    if (addend) v += g * addend[r * C + i];
    if (accumulate) v += dx[r * C + i];
    dx[r * C + i] = v;
  }
}

// partial[z][i][j] = sum over the z-th slab of rows k = (b, t >= 1) of U[k][i] V[k][j]   (see the header)
__global__ void __launch_bounds__(256) wide_grad_w_kernel(const float* __restrict__ x, int B, int T, int C, WideWs w,
                                                          const float* __restrict__ alpha, const float* __restrict__ beta,
                                                          const float* __restrict__ logz, const float* __restrict__ coef_w,
                                                          float* __restrict__ partial) {
  __shared__ float Us[kWideTK][kWideTM + 4];
  __shared__ float Vs[kWideTK][kWideTN + 4];
  __shared__ float ku[2][kWideTK], kv[2][kWideTK];  // the rows' scalars, double-buffered: chunk n + 1's are formed while chunk
  __shared__ int64_t krow[2][kWideTK];              // n's operands are on their way
  const int i0 = blockIdx.x * kWideTM, j0 = blockIdx.y * kWideTN;
  const int64_t K = (int64_t)B * (T - 1);
  const int nsplit = (int)gridDim.z;
  const int64_t per = (K + nsplit - 1) / nsplit;
  const int64_t kbeg = (int64_t)blockIdx.z * per, kend = min(K, kbeg + per);
  const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
  float acc[4][4] = {};
  // the scalars of a chunk's rows (threads 0 .. 15): a double-precision exponential each.  In front of the chunk's loads
  // they were a serial stage of every chunk -- scalars, barrier, loads, barrier, products, barrier: ~4 us a chunk
  // whatever the class count; now the NEXT chunk's are formed behind this chunk's loads, in their shadow.
  auto scalars = [&](int64_t k0, int buf) {
    const int64_t k = k0 + tid;
    float su = 0.f, sv = 0.f;
    int64_t row = 1;  // (a valid row for the clamped loads of rows past the slab: scaled by zero)
    if (k < kend) {
      const int b = (int)(k / (T - 1)), t = 1 + (int)(k % (T - 1));
      row = (int64_t)b * T + t;
      const float ma = w.maxa[row - 1], mb = w.maxb[row], lz = logz[b];
      if (ma > 0.f && mb > 0.f && lz > -3.0e38f) {
        sv = 1.f / ma;
        su = (coef_w ? coef_w[b] : 1.f) * (float)exp(w.cuma[row - 1] + (double)w.mxp[row] + w.cb[row] - (double)lz) / mb;
      }
    }
    ku[buf][tid] = su, kv[buf][tid] = sv, krow[buf][tid] = row;
  };
  if (tid < kWideTK) scalars(kbeg, 0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += kWideTK, cur ^= 1) {
    // 16 rows x 64 columns per operand: thread -> (row = tid / 16, columns (tid % 16) + 16 q); every load first, from
    // clamped addresses
    const int kr = tid >> 4, c = tid & 15;
    const int64_t row = krow[cur][kr];
    float rb[4], rx[4], ra[4], rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = min(i0 + c + 16 * q, C - 1), j = min(j0 + c + 16 * q, C - 1);
      rb[q] = beta[row * C + i], rx[q] = x[row * C + i], rr[q] = w.rm[i], ra[q] = alpha[(row - 1) * C + j];
    }
    const float ref = w.mxp[row];
    if (tid < kWideTK && k0 + kWideTK < kend) scalars(k0 + kWideTK, cur ^ 1);
    const float su = ku[cur][kr], sv = kv[cur][kr];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + c + 16 * q, j = j0 + c + 16 * q;
      float u = 0.f, v = 0.f;
      if (su != 0.f && i < C) u = su * rb[q] * __expf(wide_clean(rx[q]) + rr[q] - ref);
      if (sv != 0.f && j < C) v = sv * ra[q];
      Us[kr][c + 16 * q] = u, Vs[kr][c + 16 * q] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWideTK; ++k) {
      float a[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = Us[k][ti + 16 * u], bv[u] = Vs[k][tj + 16 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(a[u], bv[v], acc[u][v]);
    }
    __syncthreads();
  }
  float* out = partial + (int64_t)blockIdx.z * C * C;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + ti + 16 * u, j = j0 + tj + 16 * v;
      if (i < C && j < C) out[(int64_t)i * C + j] = acc[u][v];
    }
}

// dW = (accumulate ? dW : 0) + gout * dW_addend + gout * [start row: sum_b coef_w[b] posterior_0 ; P (.) sum of slabs]
__global__ void __launch_bounds__(256) wide_reduce_w_kernel(int B, int T, int C, WideWs w, const float* __restrict__ alpha,
                                                            const float* __restrict__ beta, const float* __restrict__ logz,
                                                            const float* __restrict__ coef_w, const float* __restrict__ gout,
                                                            int accumulate, const float* __restrict__ dW_addend,
                                                            const float* __restrict__ partial, float* __restrict__ dW) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)(C + 1) * C) return;
  const float g = gout ? gout[0] : 1.f;
  float v = 0.f;
  if (e < C) {  // start row
    const int i = (int)e;
    for (int b = 0; b < B; ++b) {
      const int64_t r = (int64_t)b * T;
      const float ma = w.maxa[r], mb = w.maxb[r], lz = logz[b];
      if (ma > 0.f && mb > 0.f && lz > -3.0e38f)
        v += (coef_w ? coef_w[b] : 1.f) * (float)exp(w.cuma[r] + w.cb[r] - (double)lz) / (ma * mb) * alpha[r * C + i] * beta[r * C + i];
    }
  } else {
    const int64_t p = e - C;
    float s = 0.f;
    const int nsplit = wide_split(C);
    for (int z = 0; z < nsplit; ++z) s += partial[(int64_t)z * C * C + p];
    v = w.P[p] * s;
  }
  v *= g;
  if (dW_addend) v += g * dW_addend[e];
  if (accumulate) v += dW[e];
  dW[e] = v;
}

// Tropical frame: v_t[b][i] = x[b,t,i] + max_j (v_{t-1}[b][j] + W[1+i][j]), bptr = the lowest arg max
__global__ void __launch_bounds__(256) wide_viterbi_frame_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                 int B, int T, int C, int t, float* __restrict__ alpha,
                                                                 int32_t* __restrict__ bptr) {
  __shared__ float As[kWideTK][kWideTM + 4];
  __shared__ float Bs[kWideTK][kWideTN + 4];
  const int i0 = blockIdx.x * kWideTM, n0 = blockIdx.y * kWideTN;
  const int tid = threadIdx.x, ti = tid & 15, tn = tid >> 4;
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  float best[4][4];
  int arg[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) best[u][v] = WFL_NEG_INF, arg[u][v] = -1;
  for (int k0 = 0; k0 < C; k0 += kWideTK) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + lk + q;
      const int ai = i0 + lr, bn = n0 + lr;
      As[lk + q][lr] = (ai < C && k < C) ? wide_clean(W[(int64_t)(1 + ai) * C + k]) : WFL_NEG_INF;
      Bs[lk + q][lr] = (bn < B && k < C) ? alpha[((int64_t)bn * T + t - 1) * C + k] : WFL_NEG_INF;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWideTK; ++k) {
      float a[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = As[k][ti + 16 * u], bv[u] = Bs[k][tn + 16 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float cand = a[u] + bv[v];
          if (cand > best[u][v]) best[u][v] = cand, arg[u][v] = k0 + k;  // strict: the lowest previous label wins ties
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int n = n0 + tn + 16 * v;
    if (n >= B) continue;
    const int64_t row = (int64_t)n * T + t;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + ti + 16 * u;
      if (i >= C) continue;
      alpha[row * C + i] = wide_clean(x[row * C + i]) + best[u][v];
      bptr[row * C + i] = arg[u][v];
    }
  }
}
// The same frame WITHOUT back-pointers, for wfl_dense_viterbi (whose back-trace re-derives the one back-pointer per frame
// it follows: dense_viterbi_walk_kernel): v_t[b][i] = (max_j (v_{t-1}[b][j] + W[1+i][j])) + x[b,t,i], the same additions in
// the same order as the kernel above -- the stored vectors are identical.  That kernel's 64 x 64 tiles are 16 workgroups
// at N = 1000, B = 32, each walking all 1000 previous labels behind barriers: 236 us per frame, 59 ms per call.  Here a
// workgroup owns 16 states x 16 utterances, its sixteen waves each take every sixteenth group of four previous labels, two
// groups' loads in flight (operands straight from L2: the matrix and the previous vectors stay there across the frame's
// workgroups; with four waves and one group at a time a wave waited out 62 L2 round trips: 27 us per frame) and meet in
// LDS: 63 x 2 workgroups at that shape.
// VEC4: rows of W and of the vectors are 16-byte aligned (C % 4 == 0, aligned bases): 16-byte loads
constexpr int kVitWideWaves = 16;
template <bool VEC4>
__global__ void __launch_bounds__(64 * kVitWideWaves)
    wide_viterbi_max_kernel(const float* __restrict__ x, const float* __restrict__ W, int B, int T, int C, int t,
                            float* __restrict__ alpha) {
  __shared__ float part[kVitWideWaves][16][17];
  const int i0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, il = lane & 15, ng = lane >> 4;
  const float* wrow = W + (int64_t)(1 + min(i0 + il, C - 1)) * C;
  const float* arow[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) arow[v] = alpha + ((int64_t)min(n0 + ng + 4 * v, B - 1) * T + (t - 1)) * C;
  float best[4] = {WFL_NEG_INF, WFL_NEG_INF, WFL_NEG_INF, WFL_NEG_INF};
  const int ngroups = (C + 3) >> 2;
  auto load = [&](int g, float (&wv)[4], float (&av)[4][4]) {  // (g clamped by the caller: a valid group, maybe a repeated one)
    const int k = 4 * g;
    if (VEC4) {
      const float4 q = *reinterpret_cast<const float4*>(wrow + k);
      wv[0] = q.x, wv[1] = q.y, wv[2] = q.z, wv[3] = q.w;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 a = *reinterpret_cast<const float4*>(arow[v] + k);
        av[v][0] = a.x, av[v][1] = a.y, av[v][2] = a.z, av[v][3] = a.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int kc = min(k + c, C - 1);
        wv[c] = k + c < C ? wrow[kc] : WFL_NEG_INF;
#pragma unroll
        for (int v = 0; v < 4; ++v) av[v][c] = arow[v][kc];
      }
    }
  };
  auto fold = [&](float (&wv)[4], const float (&av)[4][4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) wv[c] = wide_clean(wv[c]);
#pragma unroll
    for (int v = 0; v < 4; ++v)
      best[v] = fmaxf(fmaxf(best[v], fmaxf(wv[0] + av[v][0], wv[1] + av[v][1])), fmaxf(wv[2] + av[v][2], wv[3] + av[v][3]));
  };
  // (two groups per trip: the second one clamped to the wave's last -- folding a group twice changes no maximum)
  for (int g = wave; g < ngroups; g += 2 * kVitWideWaves) {
    float w0[4], a0[4][4], w1[4], a1[4][4];
    load(g, w0, a0);
    load(min(g + kVitWideWaves, ngroups - 1), w1, a1);
    fold(w0, a0);
    fold(w1, a1);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) part[wave][ng + 4 * v][il] = best[v];
  __syncthreads();
  if (tid < 256) {  // thread (n, i) of the tile: the waves' maxima, + the emission
    const int i = i0 + (tid & 15), n = n0 + (tid >> 4);
    if (i < C && n < B) {
      float m = part[0][tid >> 4][tid & 15];
#pragma unroll
      for (int w = 1; w < kVitWideWaves; ++w) m = fmaxf(m, part[w][tid >> 4][tid & 15]);
      const int64_t at = ((int64_t)n * T + t) * C + i;
      alpha[at] = wide_clean(x[at]) + m;
    }
  }
}
// The max-plus frames of ONE utterance in ONE workgroup, the matrix in its registers: wide_resident_sweep_kernel's layout
// (a group of 16 lanes owns NR consecutive rows of W, a lane NR float4 column chunks of each; the previous vector in LDS)
// for wfl_dense_viterbi between 257 and 320 classes, where the per-frame launches above cost ~8 us of stream time per
// frame (C = 320, T = 1000, B = 128: 8.6 ms).  Every sum is ONE fp32 addition of the same two operands as in
// wide_viterbi_max_kernel and a maximum does not depend on the order it is taken in: the stored vectors are identical,
// bit for bit, and so is the path dense_viterbi_walk_kernel reads off them.
template <int NR>
__global__ void __launch_bounds__(1024) wide_resident_viterbi_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                     int B, int T, int C, float* __restrict__ alpha) {
  constexpr int CP = 64 * NR;
  __shared__ __attribute__((aligned(16))) float vbuf[2][CP];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  const int row0 = NR * grp;
  const float* xb = x + (int64_t)b * T * C;
  float* ob = alpha + (int64_t)b * T * C;
  float Wl[NR][NR][4];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        Wl[r][k][c] = W[(int64_t)(1 + min(row0 + r, C - 1)) * C + min(64 * k + 4 * l16 + c, C - 1)];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        Wl[r][k][c] = (row0 + r < C && 64 * k + 4 * l16 + c < C) ? wide_clean(Wl[r][k][c]) : WFL_NEG_INF;
  const int rsel = l16 % NR;
  const int myrow = row0 + rsel;
  const bool live = myrow < C;
  const int rowc = min(myrow, C - 1);
  for (int i = tid; i < 2 * CP; i += 1024) (&vbuf[0][0])[i] = i < C ? ob[i] : WFL_NEG_INF;  // frame 0 (wide_viterbi_first_max_kernel)
  // (lanes without a row park their stores in the LAST frame, at a column of their own: wide_resident_sweep_kernel)
  float* po = live ? ob + (int64_t)C + myrow : ob + (int64_t)(T - 1) * C + max(myrow - C, 0);
  const int64_t pstep = live ? (int64_t)C : 0;
  constexpr int kAhead = 4;
  float xr[kAhead];
#pragma unroll
  for (int u = 0; u < kAhead; ++u) xr[u] = xb[(int64_t)min(1 + u, T - 1) * C + rowc];
  __syncthreads();
  auto step = [&](int n, float xv, auto last_) {
    constexpr bool LAST = decltype(last_)::value;
    const float* vc = vbuf[(n - 1) & 1];
    float* vn = vbuf[n & 1];
    float best[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) best[r] = WFL_NEG_INF;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const float4 v4 = *reinterpret_cast<const float4*>(vc + 64 * k + 4 * l16);
#pragma unroll
      for (int r = 0; r < NR; ++r)
        best[r] = fmaxf(fmaxf(best[r], fmaxf(Wl[r][k][0] + v4.x, Wl[r][k][1] + v4.y)), fmaxf(Wl[r][k][2] + v4.z, Wl[r][k][3] + v4.w));
    }
    float mine = WFL_NEG_INF;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float m = best[r];  // the row's maximum over the group's 16 lanes (any order: exact)
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0xB1, 0xf, 0xf, false)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0x4E, 0xf, 0xf, false)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0x141, 0xf, 0xf, false)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0x140, 0xf, 0xf, false)));
      mine = rsel == r ? m : mine;
    }
    const float y = wide_clean(xv) + mine;
    vn[myrow] = live ? y : WFL_NEG_INF;
    if (!LAST || live) *po = y;
    po += pstep;
    lds_barrier();
  };
  int n = 1;
  const float* px = xb + (int64_t)min(1 + kAhead, T - 1) * C + rowc;
  for (; n + 2 * kAhead <= T; n += kAhead) {
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      step(n + u, xr[u], std::false_type{});
      xr[u] = px[(int64_t)u * C];
    }
    px += (int64_t)kAhead * C;
  }
  for (; n + 1 < T; ++n) step(n, xb[(int64_t)n * C + rowc], std::false_type{});
  if (n < T) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): what was parked in the last frame has landed
    __syncthreads();
    step(n, xb[(int64_t)n * C + rowc], std::true_type{});
  }
}

__global__ void __launch_bounds__(256) wide_viterbi_first_max_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                     int B, int T, int C, float* __restrict__ alpha) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * C) return;
  const int b = (int)(e / C), i = (int)(e % C);
  const int64_t at = (int64_t)b * T * C + i;
  alpha[at] = wide_clean(x[at]) + wide_clean(W[i]);
}
// The path from the stored vectors for ANY class count (dense_viterbi_backtrace_kernel keeps the vectors' chunks and
// the matrix in LDS and serves up to 256 classes): one wave per utterance, a step reads the previous frame's vector and
// the current state's row of W from L2, C / 64 values per lane, keeps the lane's first maximum, then the wave's
// maximum and the lowest label that attains it.
__global__ void __launch_bounds__(64) dense_viterbi_walk_kernel(const float* __restrict__ alpha, const float* __restrict__ W, int B,
                                                                int T, int C, int32_t* __restrict__ path) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B || T <= 0) return;
  const float* ab = alpha + (int64_t)b * T * C;
  int32_t* out = path + (int64_t)b * T;
  int cur;
  {
    float best = WFL_NEG_INF;
    int arg = 0x3fffffff;
    for (int i = lane; i < C; i += 64) {
      const float v = ab[(int64_t)(T - 1) * C + i];
      if (v > best || arg == 0x3fffffff) best = v, arg = i;
    }
    const float m = wave_all_max(best);
    arg = -wave_all_max_int(-(best == m ? arg : 0x3fffffff));
    cur = arg == 0x3fffffff ? 0 : arg;
  }
  for (int t = T - 1; t >= 0; --t) {
    if (lane == 0) out[t] = cur;
    if (t == 0) break;
    const float* prev = ab + (int64_t)(t - 1) * C;
    const float* wrow = W + (int64_t)(1 + cur) * C;
    float best = WFL_NEG_INF;
    int arg = 0x3fffffff;
    for (int j0 = lane; j0 < C; j0 += 64 * 8) {  // (eight loads of each operand in flight: clamped addresses, no test around them)
      float pv[8], wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = min(j0 + 64 * u, C - 1);
        pv[u] = prev[j], wv[u] = wrow[j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + 64 * u;
        const float v = j < C ? pv[u] + wide_clean(wv[u]) : WFL_NEG_INF;
        if (v > best) best = v, arg = j;  // (strict, ascending j: the lane's lowest)
      }
    }
    const float top = wave_all_max(best);
    const int lowest = -wave_all_max_int(-((best == top && top > WFL_NEG_INF) ? arg : 0x3fffffff));
    cur = lowest == 0x3fffffff ? 0 : lowest;  // unreachable state (all -inf): keep the path well-formed
  }
}
__global__ void __launch_bounds__(256) wide_viterbi_first_kernel(const float* __restrict__ x, const float* __restrict__ W, int B,
                                                                 int T, int C, float* __restrict__ alpha,
                                                                 int32_t* __restrict__ bptr) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * C) return;
  const int b = (int)(e / C), i = (int)(e % C);
  const int64_t at = (int64_t)b * T * C + i;
  alpha[at] = wide_clean(x[at]) + wide_clean(W[i]);
  bptr[at] = -1;
}

// ------------------------------------------------------------------------------------------------ host side
static int wide_forward(const float* x, const float* W, int B, int T, int C, int semiring, float* alpha, float* beta,
                        int32_t* bptr, float* logz, void* ws, hipStream_t st) {
  const dim3 tiles((unsigned)((C + kWideTM - 1) / kWideTM), (unsigned)((B + kWideTN - 1) / kWideTN));
  if (semiring == WFL_SEMIRING_TROPICAL) {
    hipLaunchKernelGGL(wide_viterbi_first_kernel, dim3((unsigned)(((int64_t)B * C + 255) / 256)), dim3(256), 0, st, x, W, B, T,
                       C, alpha, bptr);
    for (int t = 1; t < T; ++t)
      hipLaunchKernelGGL(wide_viterbi_frame_kernel, tiles, dim3(256), 0, st, x, W, B, T, C, t, alpha, bptr);
    return WFL_OK;
  }
  const WideWs w = wide_carve(ws, B, T, C);
  hipLaunchKernelGGL(wide_prep_kernel, dim3((unsigned)C), dim3(256), 0, st, W, C, w);
  hipLaunchKernelGGL(wide_rows_kernel, dim3((unsigned)(((int64_t)B * T + 3) / 4)), dim3(256), 0, st, x, W, B, T, C, w, alpha,
                     beta);
  if (!wide_resident_frames(x, B, T, C, w, alpha, beta, st)) {
    const dim3 grid((unsigned)((C + 15) / 16), (unsigned)((B + 15) / 16), beta ? 2u : 1u);
    for (int s = 1; s < T; ++s)
      hipLaunchKernelGGL(wide_frame_mfma_kernel, grid, dim3(64 * kWideMfmaWaves), 0, st, x, B, T, C, s, w, alpha, beta);
  }
  hipLaunchKernelGGL(wide_scan_kernel, dim3((unsigned)B), dim3(256), 0, st, B, T, C, w, alpha, beta != nullptr, logz);
  return WFL_OK;
}

static int wide_grad(const float* x, int B, int T, int C, const float* alpha, const float* beta, const float* logz,
                     const float* coef, const float* coef_w, const float* gout, int accumulate, const float* addend,
                     const float* dW_addend, float* dx, float* dW, float* dW_partial, const void* ws, hipStream_t st) {
  const WideWs w = wide_carve(const_cast<void*>(ws), B, T, C);
  if (dx)
    hipLaunchKernelGGL(wide_grad_x_kernel, dim3((unsigned)(((int64_t)B * T + 3) / 4)), dim3(256), 0, st, B, T, C, w, alpha, beta,
                       logz, coef, gout, accumulate, addend, dx);
  if (dW) {
    if (T > 1) {
      const dim3 grid((unsigned)((C + kWideTM - 1) / kWideTM), (unsigned)((C + kWideTN - 1) / kWideTN), (unsigned)wide_split(C));
      hipLaunchKernelGGL(wide_grad_w_kernel, grid, dim3(256), 0, st, x, B, T, C, w, alpha, beta, logz, coef_w, dW_partial);
    } else {
      (void)hipMemsetAsync(dW_partial, 0, (size_t)4 * wide_split(C) * C * C, st);
    }
    hipLaunchKernelGGL(wide_reduce_w_kernel, dim3((unsigned)(((int64_t)(C + 1) * C + 255) / 256)), dim3(256), 0, st, B, T, C, w,
                       alpha, beta, logz, coef_w, gout, accumulate, dW_addend, dW_partial, dW);
  }
  return WFL_OK;
}

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
// over K = B (T-1) rows, split into kWideSplit slabs reduced in a fixed order (deterministic, no atomics).
// Tropical semiring (ASG.viterbi, asg.py:217-226): the same tiling with (max, +) and the arg max (lowest previous
// label on ties, as the LDS-resident kernel).
#pragma once

constexpr int kWideTM = 64, kWideTN = 64, kWideTK = 16;  // output tile (states x utterances) and K chunk
constexpr int kWideSplit = 8;                             // K slabs of the transition-gradient product

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

// per utterance: the cumulative offsets (a serial scan over T by one thread) and log Z
__global__ void __launch_bounds__(256) wide_scan_kernel(int B, int T, int C, WideWs w, const float* __restrict__ alpha,
                                                        bool with_beta, float* __restrict__ logz) {
  __shared__ float red[64];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    double c = (double)w.m0[b];
    w.cuma[(int64_t)b * T] = c;
    for (int t = 1; t < T; ++t) {
      const float mx = w.maxa[(int64_t)b * T + t];
      c += (double)w.mxp[(int64_t)b * T + t] + (mx > 0.f ? (double)__logf(mx) : -1.0e300);
      w.cuma[(int64_t)b * T + t] = c;
    }
    if (with_beta) {
      double d = 0.0;
      w.cb[(int64_t)b * T + T - 1] = 0.0;
      for (int t = T - 2; t >= 0; --t) {
        const float mx = w.maxb[(int64_t)b * T + t];
        d += (double)w.mxp[(int64_t)b * T + t + 1] + (mx > 0.f ? (double)__logf(mx) : -1.0e300);
        w.cb[(int64_t)b * T + t] = d;
      }
    }
  }
  const float* last = alpha + ((int64_t)b * T + T - 1) * C;
  float sum = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) sum += last[i];
  sum = blk_sum(sum, red);  // (its barriers also order thread 0's scan before the read below)
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
  __shared__ float ku[kWideTK], kv[kWideTK];
  __shared__ int64_t krow[kWideTK];
  const int i0 = blockIdx.x * kWideTM, j0 = blockIdx.y * kWideTN;
  const int64_t K = (int64_t)B * (T - 1);
  const int64_t per = (K + kWideSplit - 1) / kWideSplit;
  const int64_t kbeg = (int64_t)blockIdx.z * per, kend = min(K, kbeg + per);
  const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
  float acc[4][4] = {};
  for (int64_t k0 = kbeg; k0 < kend; k0 += kWideTK) {
    if (tid < kWideTK) {
      const int64_t k = k0 + tid;
      float su = 0.f, sv = 0.f;
      int64_t row = 0;
      if (k < kend) {
        const int b = (int)(k / (T - 1)), t = 1 + (int)(k % (T - 1));
        row = (int64_t)b * T + t;
        const float ma = w.maxa[row - 1], mb = w.maxb[row], lz = logz[b];
        if (ma > 0.f && mb > 0.f && lz > -3.0e38f) {
          sv = 1.f / ma;
          su = (coef_w ? coef_w[b] : 1.f) * (float)exp(w.cuma[row - 1] + (double)w.mxp[row] + w.cb[row] - (double)lz) / mb;
        }
      }
      ku[tid] = su, kv[tid] = sv, krow[tid] = row;
    }
    __syncthreads();
    // 16 rows x 64 columns per operand: thread -> (row = tid / 16, columns (tid % 16) + 16 q)
    {
      const int kr = tid >> 4, c = tid & 15;
      const int64_t row = krow[kr];
      const float su = ku[kr], sv = kv[kr], ref = w.mxp[row];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + c + 16 * q, j = j0 + c + 16 * q;
        float u = 0.f, v = 0.f;
        if (su != 0.f && i < C) u = su * beta[row * C + i] * __expf(wide_clean(x[row * C + i]) + w.rm[i] - ref);
        if (sv != 0.f && j < C) v = sv * alpha[(row - 1) * C + j];
        Us[kr][c + 16 * q] = u, Vs[kr][c + 16 * q] = v;
      }
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
    for (int z = 0; z < kWideSplit; ++z) s += partial[(int64_t)z * C * C + p];
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
  {
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
      const dim3 grid((unsigned)((C + kWideTM - 1) / kWideTM), (unsigned)((C + kWideTN - 1) / kWideTN), (unsigned)kWideSplit);
      hipLaunchKernelGGL(wide_grad_w_kernel, grid, dim3(256), 0, st, x, B, T, C, w, alpha, beta, logz, coef_w, dW_partial);
    } else {
      (void)hipMemsetAsync(dW_partial, 0, (size_t)4 * kWideSplit * C * C, st);
    }
    hipLaunchKernelGGL(wide_reduce_w_kernel, dim3((unsigned)(((int64_t)(C + 1) * C + 255) / 256)), dim3(256), 0, st, B, T, C, w,
                       alpha, beta, logz, coef_w, gout, accumulate, dW_addend, dW_partial, dW);
  }
  return WFL_OK;
}

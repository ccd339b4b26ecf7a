// CTC fast path for gfx950: create_ctc_graph + intersect + forward_score + backward of
// criterions/ctc.py:15-94 without any graph.
//
// Lane mapping (one 64-lane wavefront per utterance and direction): lane i owns target position i,
// i.e. the blank state 2i ("ab") and the label state 2i+1 ("al") of the CTC label graph
// (ctc.py:18-27); lane L owns the trailing blank.  One frame of the recursion needs exactly one
// cross-lane value (al of lane i-1), fetched with a DPP wave shift -- no LDS on the dependent
// chain:
//     ab' = xb   + LSE(ab, al[i-1])
//     al' = xl_i + LSE(al, ab, skip_i ? al[i-1] : -inf)         skip_i = (y_i != y_{i-1})
// The beta sweep is the same recursion on the reversed target and reversed time (the CTC graph is
// mirror-symmetric), so one routine serves both directions; it stores the value BEFORE the
// emission is added, so that posterior(t, s) = alpha_t(s) * beta~_t(s) / Z needs no emission.
// Scores are kept in base-2 log units (v_exp_f32 / v_log_f32 are base 2): no multiplies on the
// dependent chain.  -inf is represented by a large finite sentinel so the chain is branch-free.
// Emissions are gathered straight from the [B,T,C] tensor (one dword per lane per frame, all
// addresses of a wave inside one C-float row) with an 8-frame register prefetch ring.
//
//   stage A  ctc_chain_kernel   grid (B, 2): alpha and beta chains run concurrently
//   stage B  ctc_grad_kernel    all CUs: one wave per (b, t) row: posteriors, label reduction in
//                               LDS, dense row store (zeros included, as ctc.py:75 returns)
#include "device_common.h"

namespace wfl {

constexpr float kNegBig = -1.0e30f;          // stands in for -inf on the chain
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kCtcPrefetch = 16;

__device__ __forceinline__ float wave_shr1(float v, float fill) {
  // lane i receives lane i-1's value; lane 0 receives `fill` (DPP wave_shr:1, bound_ctrl off)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

__device__ __forceinline__ float lse2_b2(float a, float b) {
  const float m = fmaxf(a, b);
  return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m));
}
__device__ __forceinline__ float lse3_b2(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) +
                                   __builtin_amdgcn_exp2f(c - m));
}

// Workspace layout (float units):
//   [0, B*2*T*P*2)            float2 ws[b][dir][t][pos], pos < P (= max_len + 1): scores RELATIVE to
//                             the block offset in force when the frame was produced
//   then doubles              off[b][dir][blk], blk = step / kCtcRenorm: cumulative offset (base-2 log)
//   then doubles              z2[b]: log2 Z
// Every kCtcRenorm frames the running maximum over the wave is subtracted from the state vector and
// added to a double-precision offset.  Stored scores therefore stay O(10) instead of drifting to
// O(T): fp32 log-domain rounding is ~1e-6 instead of ~3e-4 at T = 1000 (where plain fp32 log-domain,
// which is what gtn.forward_score uses, already loses the 4th digit of the posteriors).
constexpr int kCtcRenorm = 16;

__host__ __device__ inline int64_t ctc_main_floats(int B, int T, int P) { return (int64_t)B * 2 * T * P * 2; }
__host__ __device__ inline int ctc_blocks(int T) { return (T + kCtcRenorm - 1) / kCtcRenorm; }

__global__ void __launch_bounds__(64)
    ctc_chain_kernel(const float* __restrict__ x, int B, int T, int C, const int32_t* __restrict__ targets,
                     const int64_t* __restrict__ offsets, int P, int blank, float* __restrict__ ws_raw,
                     float* __restrict__ nll) {
  const int b = blockIdx.x, dir = blockIdx.y, lane = threadIdx.x;
  const int64_t o0 = offsets[b];
  const int L = (int)(offsets[b + 1] - o0);
  // this lane's label (reversed target for the beta sweep) and skip flag
  int y = -1, yprev = -1;
  if (lane < L) y = targets[o0 + (dir == 0 ? lane : L - 1 - lane)];
  if (lane >= 1 && lane - 1 < L) yprev = targets[o0 + (dir == 0 ? lane - 1 : L - lane)];
  const bool has_label = lane < L, has_blank = lane <= L;
  const bool skip = has_label && lane >= 1 && y != yprev;
  const float* xb_ptr = x + (int64_t)b * T * C;  // row base; + t*C + column
  const int col = has_label ? y : blank;
  float2* out = (float2*)ws_raw + ((int64_t)(b * 2 + dir) * T) * P;
  const int NB = ctc_blocks(T);
  double* offs = (double*)(ws_raw + ctc_main_floats(B, T, P)) + (int64_t)(b * 2 + dir) * NB;
  double* z2out = (double*)(ws_raw + ctc_main_floats(B, T, P)) + (int64_t)B * 2 * NB;

  float ab = (lane == 0) ? 0.f : kNegBig;  // virtual slot "before the first frame"
  float al = kNegBig;
  double off = 0.0;
  // Prefetch ring of RAW emissions (scaling/clamping happens at use, kCtcPrefetch frames later, so
  // that no load is consumed right after it is issued).  Lanes >= L fetch the blank column; the
  // trailing-blank lane L therefore holds x[t, blank], broadcast with v_readlane (no SMEM load, no
  // second VMEM load per frame).
  float ring[kCtcPrefetch];
#pragma unroll
  for (int j = 0; j < kCtcPrefetch; ++j) {
    const int step = min(j, T - 1);
    const int t = dir == 0 ? step : T - 1 - step;
    ring[j] = xb_ptr[(int64_t)t * C + col];
  }
  auto renorm = [&](int blk) {
    if (blk > 0) {
      const float m = wave_max(fmaxf(ab, al));
      if (m > 0.5f * kNegBig) {
        ab = fmaxf(ab - m, kNegBig);
        al = fmaxf(al - m, kNegBig);
        off += (double)m;
      }
    }
    if (lane == 0) offs[blk] = off;
  };
  auto frame = [&](float raw, int t) {
    float xs = raw * kLog2e;
    xs = (xs > kNegBig) ? xs : kNegBig;  // NaN and -inf both become the sentinel (NaN policy)
    const float xblank = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), L));
    const float xl = has_label ? xs : kNegBig;
    const float xb = has_blank ? xblank : kNegBig;
    const float pal = wave_shr1(al, kNegBig);
    const float nb = lse2_b2(ab, pal);
    const float nl = lse3_b2(al, ab, skip ? pal : kNegBig);
    ab = fmaxf(nb + xb, kNegBig);
    al = fmaxf(nl + xl, kNegBig);
    if (lane < P) out[(int64_t)t * P + lane] = dir == 0 ? make_float2(ab, al) : make_float2(nb, nl);
  };
  const int nfull = T / kCtcPrefetch;
  for (int c = 0; c < nfull; ++c) {
    if ((c * kCtcPrefetch) % kCtcRenorm == 0) renorm(c * kCtcPrefetch / kCtcRenorm);
#pragma unroll
    for (int j = 0; j < kCtcPrefetch; ++j) {
      const int step = c * kCtcPrefetch + j;
      const float raw = ring[j];
      const int sn = min(step + kCtcPrefetch, T - 1);  // clamped: a valid address, unused past the end
      ring[j] = xb_ptr[(int64_t)(dir == 0 ? sn : T - 1 - sn) * C + col];
      frame(raw, dir == 0 ? step : T - 1 - step);
    }
  }
  {
    const int s0 = nfull * kCtcPrefetch, rem = T - s0;
    if (rem > 0 && s0 % kCtcRenorm == 0) renorm(s0 / kCtcRenorm);
#pragma unroll
    for (int j = 0; j < kCtcPrefetch; ++j)
      if (j < rem) frame(ring[j], dir == 0 ? s0 + j : T - 1 - (s0 + j));
  }
  if (dir == 0) {
    // logZ = LSE(alpha_{T-1}[2L], alpha_{T-1}[2L-1]) = LSE(ab[L], al[L-1])   (ctc.py:21 accept states)
    const float a_last = __shfl(ab, L, 64);
    const float l_last = L > 0 ? __shfl(al, L - 1, 64) : kNegBig;
    if (lane == 0) {
      const float zr = lse2_b2(a_last, l_last);
      const bool alive = zr > 0.5f * kNegBig;
      const double z2 = alive ? (double)zr + off : -1.0e300;
      z2out[b] = z2;
      nll[b] = alive ? (float)(-z2 * 0.6931471805599453) : __builtin_inff();
    }
  }
}

__global__ void __launch_bounds__(256)
    ctc_grad_kernel(int B, int T, int C, const int32_t* __restrict__ targets, const int64_t* __restrict__ offsets,
                    int P, int blank, const float* __restrict__ ws_raw, const float* __restrict__ nll,
                    const float* __restrict__ coef, const float* __restrict__ gout, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* row = (float*)smem + (size_t)wave * C;
  const int64_t o0 = offsets[b];
  const int L = (int)(offsets[b + 1] - o0);
  const int y = lane < L ? targets[o0 + lane] : blank;
  const bool dead = !(nll[b] < __builtin_inff());
  const float cf = (coef ? coef[b] : 1.f) * (gout ? gout[0] : 1.f);
  const int NB = ctc_blocks(T);
  const float2* al = (const float2*)ws_raw + ((int64_t)(b * 2 + 0) * T) * P;
  const float2* be = (const float2*)ws_raw + ((int64_t)(b * 2 + 1) * T) * P;
  const double* offa = (const double*)(ws_raw + ctc_main_floats(B, T, P)) + (int64_t)(b * 2 + 0) * NB;
  const double* offb = offa + NB;
  const double z2 = ((const double*)(ws_raw + ctc_main_floats(B, T, P)))[(int64_t)B * 2 * NB + b];
  for (int c = lane; c < C; c += 64) row[c] = 0.f;
  __syncthreads();
  for (int tb = blockIdx.x * 4; tb < T; tb += gridDim.x * 4) {  // uniform trip count: barriers inside
    const int t = tb + wave;
    const bool live = t < T && !dead;
    float gb = 0.f, gl = 0.f;
    if (live && lane <= L) {
      // block offsets of the two sweeps at this frame, combined with log2 Z in double
      const float delta = (float)(offa[t / kCtcRenorm] + offb[(T - 1 - t) / kCtcRenorm] - z2);
      const float2 a = al[(int64_t)t * P + lane];
      // mirrored beta: blank state 2i <-> reversed position L-i; label of position i <-> L-1-i
      const float bb = be[(int64_t)t * P + (L - lane)].x;
      gb = __builtin_amdgcn_exp2f(a.x + bb + delta);
      if (lane < L) {
        const float bl = be[(int64_t)t * P + (L - 1 - lane)].y;
        gl = __builtin_amdgcn_exp2f(a.y + bl + delta);
      }
    }
    gb = wave_sum(gb);
    if (lane == 0 && gb != 0.f) atomicAdd(&row[blank], gb * cf);
    if (lane < L && gl != 0.f) atomicAdd(&row[y], gl * cf);
    __syncthreads();
    if (t < T) {
      float* dst = dx + ((int64_t)b * T + t) * C;
      for (int c = lane; c < C; c += 64) {
        dst[c] = row[c];
        row[c] = 0.f;
      }
    }
    __syncthreads();
  }
}

}  // namespace wfl

using namespace wfl;

extern "C" {

static int ctc_check(int B, int T, int C, int max_len, int blank, const char* who) {
  if (B <= 0 || T <= 0 || C <= 0 || blank < 0 || blank >= C || max_len < 0) {
    set_error("%s: bad arguments (B=%d T=%d C=%d blank=%d max_len=%d)", who, B, T, C, blank, max_len);
    return WFL_ERR_INVALID;
  }
  if (max_len + 1 > 64) {
    set_error("%s: target length %d needs more than one 64-lane wavefront (use the lattice engine)", who, max_len);
    return WFL_ERR_UNSUPPORTED;
  }
  return WFL_OK;
}

int wfl_ctc_workspace(int B, int T, int C, int max_len, int64_t* ws_elems) {
  if (!ws_elems || B <= 0 || T <= 0 || max_len < 0) {
    set_error("ctc_workspace: bad arguments");
    return WFL_ERR_INVALID;
  }
  (void)C;
  *ws_elems = ctc_main_floats(B, T, max_len + 1) + 2 * ((int64_t)B * 2 * ctc_blocks(T) + B) + 2;
  return WFL_OK;
}

int wfl_ctc_forward(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets, int max_len,
                    int blank, float* ws, float* nll, void* stream) {
  if (int rc = ctc_check(B, T, C, max_len, blank, "ctc_forward")) return rc;
  if (!x || !targets || !offsets || !ws || !nll) {
    set_error("ctc_forward: null buffer");
    return WFL_ERR_INVALID;
  }
  hipLaunchKernelGGL(ctc_chain_kernel, dim3((unsigned)B, 2u), dim3(64), 0, (hipStream_t)stream, x, B, T, C, targets,
                     offsets, max_len + 1, blank, ws, nll);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_ctc_grad(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets, int max_len,
                 int blank, const float* ws, const float* nll, const float* coef, const float* gout, float* dx,
                 void* stream) {
  if (int rc = ctc_check(B, T, C, max_len, blank, "ctc_grad")) return rc;
  if (!targets || !offsets || !ws || !nll || !dx) {
    set_error("ctc_grad: null buffer");
    return WFL_ERR_INVALID;
  }
  (void)x;
  const int blocks_t = std::max(1, std::min((T + 3) / 4, (4096 + B - 1) / B));
  hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)blocks_t, (unsigned)B), dim3(256), (size_t)4 * C * 4,
                     (hipStream_t)stream, B, T, C, targets, offsets, max_len + 1, blank, ws, nll, coef, gout, dx);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

}  // extern "C"

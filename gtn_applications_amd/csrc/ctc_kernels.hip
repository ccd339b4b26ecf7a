// CTC fast path for gfx950: create_ctc_graph + intersect + forward_score + backward of
// criterions/ctc.py:15-94 without any graph.
//
// Lane mapping (one 64-lane wavefront per utterance and direction): lane i owns target position i,
// i.e. the blank state 2i ("b") and the label state 2i+1 ("l") of the CTC label graph
// (ctc.py:18-27); lane L owns the trailing blank.  One frame of the recursion
//     b' = xb   + LSE(b, l[i-1])
//     l' = xl_i + LSE(l, b, skip_i ? l[i-1] : -inf)            skip_i = (y_i != y_{i-1})
// needs exactly one cross-lane value (the label state of lane i-1), fetched with a DPP wave shift:
// no LDS on the dependent chain.  The beta sweep is the same recursion on the reversed target and
// reversed time (the CTC graph is mirror-symmetric), so one routine serves both directions.
//
// Measured on MI355X (scratch/chain_ubench.hip): a lone wavefront issues ~1 instruction per 4
// cycles and every VMEM instruction costs it ~50 cycles -- the chain is bound by its instruction
// count per frame, not by bandwidth.  Therefore:
//   * the chain stores NOTHING per frame: only the state vector at every 16-frame boundary
//     (a "checkpoint": 1/16 of the alpha/beta volume) together with its scale offset;
//   * the gradient kernel recomputes alpha forward and beta backward INSIDE each 16-frame block from
//     the two checkpoints that bracket it -- 8064 independent wave-sized chains at cfg2, i.e. a
//     throughput problem spread over all 256 CUs -- forms the posteriors and assembles the dense
//     gradient rows of the block in LDS, one coalesced 16-B/lane copy per block;
//   * emissions are gathered straight from the [B,T,C] tensor (one dword per lane per frame, all
//     addresses of a wave inside one C-float row) through a 16-frame register ring of RAW values (no
//     load is consumed right after issue); the trailing-blank lane's value is broadcast with
//     v_readlane, so there is exactly one VMEM load per frame and no SMEM load.
//
// Arithmetic: base-2 log domain (v_exp_f32 / v_log_f32 are base 2: no multiplies on the dependent
// chain), -inf represented by a large finite sentinel so the chain is branch-free.  Every 16 frames
// the wave maximum is moved into a double-precision offset: stored scores stay O(10) instead of
// drifting to O(T) (plain fp32 log-domain, which is what gtn.forward_score does, already loses the
// 4th digit of the posteriors at T = 1000).
//
// Kernels.  The training step (wfl_ctc_forward_backward) is ctc_mitm_kernel (ctc_mitm.h: the sweeps emit the gradient)
// for targets of up to 63 labels -- plain or with the fused log_softmax, rows of any width the entry point admits; it
// runs in lane-exponent (probability-domain) arithmetic, every block certifies what it computed, and ctc_repair_kernel
// re-runs rejected utterances with the log-domain bodies.  ctc_pipelined_kernel is the single launch with the log-domain bodies
// throughout (WFL_CTC_PIPELINE=log, a step whose predecessor was mostly repaired), ctc_long_pipelined_kernel the one
// for targets of 64-255 labels.  ctc_log_chain_kernel + ctc_grad_kernel (+ wfl_reduce_loss) are the forward-only /
// three-launch path behind wfl_ctc_forward / wfl_ctc_grad.  (Retired in round 4: the standalone lane-exponent chain
// launch with its certify kernel -- WFL_CTC_FAST_CHAIN -- and the dbg_* kernels.  Retired in round 5: the round-2
// lane-exponent launch whose gradient waves recomputed the blocks, ctc_fast_pipelined_kernel, with its compact
// pre-pass and its switches WFL_CTC_MITM / _WIDE / _LSM.)
#include <atomic>
#include <string>
#include <type_traits>

#include "device_common.h"

#ifndef WFL_MITM_STATS
#define WFL_MITM_STATS 0
#endif

namespace wfl {

constexpr float kNegBig = -1.0e30f;  // stands in for -inf on the chain
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kBlk = 16;  // frames per block: renormalisation, checkpoint and prefetch period

// log2(2^a + 2^b) with ONE exp and one log: max + log2(1 + 2^-|a-b|).  With the finite -inf sentinel
// the difference of two sentinels is 0 (-> sentinel + 1, still a sentinel) and sentinel vs finite
// gives 2^-huge = 0: branch-free.  Cost model on gfx950 (measured): 4 cycles per VALU, 16 per
// transcendental: 4*4 + 2*16 = 48 cycles.  The bare v_max_f32 avoids the two canonicalising
// v_max x,x,x that fmaxf() costs under IEEE mode (no NaN can reach this point: NaN policy at load).
__device__ __forceinline__ float lse2_b2(float a, float b) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(a - b));
  return vmax(a, b) + __builtin_amdgcn_logf(1.f + e);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float to_score(float raw) {
  const float v = raw * kLog2e;
  return (v > kNegBig) ? v : kNegBig;  // NaN and -inf both become the sentinel (NaN policy)
}

// ------------------------------------------------------------------------------------------------
// workspace layout (float units).  NB = ceil(T / 16), P = max_len + 1.
//   float2 ck[b][dir][kk][P]   kk = 0..NB-1 in PROCESSING order: state before the kk-th block the
//                              sweep processed (alpha: block kk; beta: block NB-1-kk), base-2 log
//                              scores relative to off[b][dir][kk], emissions of all frames consumed
//                              so far included
//   double off[b][dir][kk]     the offset
//   double z2[b]               log2 Z
//   int32  flag[b], pbad[b][dir]   bookkeeping of the fast chain's certificate (see below)
// ------------------------------------------------------------------------------------------------
struct CtcWs {
  int64_t ck, off, z2, flag, pbad, ready, done, perr, dup, own, zloc, suspect, zcnt, clk, total, dbg;
};
__host__ __device__ inline int ctc_blocks(int T) { return (T + kBlk - 1) / kBlk; }
__host__ __device__ inline CtcWs ctc_ws_layout(int B, int T, int P) {
  const int64_t NB = ctc_blocks(T);
  CtcWs w;
  int64_t o = 0;
  w.ck = o, o += (int64_t)B * 2 * (NB + 1) * P * 2;  // (+1 block: the raw checkpoints of ctc_mitm.h, [NB/2 + 1][2][P] x 8 B per sweep)
  o = (o + 1) & ~1ll;
  w.off = o, o += 2 * (int64_t)B * 2 * NB;
  w.z2 = o, o += 2 * (int64_t)B;
  w.flag = o, o += B;      // int32 flag[b]: 1 = the fast chain's result was rejected, log-domain chain re-ran
  w.pbad = o, o += 2 * B;  // int32 pbad[b][dir]: the fast chain could not vouch for an interval
  o = (o + 1) & ~1ll;
  w.ready = o, o += 2 * (int64_t)B * 2 * NB;  // uint64 ready[b][dir][block]: == the launch token once published
  w.done = o, o += 2 * (int64_t)B;            // uint64 done[b]: == the launch token once nll[b] is published
  w.perr = o, o += 2;                         // int32: a gradient wave of the pipelined step gave up waiting
  w.dup = o, o += 2 * (int64_t)B;             // uint64 dup[b]: bit i = target label i also occurs elsewhere in the target (or is the blank)
  w.own = o, o += 64 * (int64_t)B;            // int32 own[b][64]: lane of the first occurrence of the lane's label (63: the blank's slot)
  w.zloc = o, o += 4 * (int64_t)B;            // int64 zloc[b][2]: min / max over the blocks of log2 Z (x 2^16) as their gradient waves reproduced it
  w.suspect = o, o += 2;                      // uint64: == the meet-in-the-middle launch's token once ANY of its certificates has a doubt (ctc_mitm.h)
  w.zcnt = o, o += B;                         // int32 zcnt[b]: sweeps of b whose first emitted block has folded its log2 Z into zloc
  o = (o + 1) & ~1ll;
  w.clk = o, o += 8 * (int64_t)B;             // int64 clk[b][dir][2]: the constant 100 MHz clock when the sweep's workgroup entered the
                                              // meet-in-the-middle launch and when its last wave left it (bench.py: the launch alone)
#if WFL_MITM_STATS
  o = (o + 1) & ~1ll;
  w.dbg = o, o += 2 * (8 * 16 + 256) * 2 * (int64_t)B;  // int64 [b][dir][wave][8], then [b][dir][256] block clocks (ctc_mitm.h)
#endif
  w.total = o + 2;
  return w;
}

struct CtcArgs {
  const float* x;
  int B, T, C, P, blank;
  const int32_t* targets;
  const int64_t* offsets;
  float* ws;
  float* nll;
  unsigned long long token;  // pipelined step: value a ready flag takes when its checkpoint is published
  const float* loss_scale;   // pipelined step, optional: loss_out[0] = mean_b(loss_scale[b] * nll[b])  (ctc.py:68-69)
  float* loss_out;
  // pipelined step, optional: x holds raw scores and row_lse[b*T + t] their log-sum-exp -- the fused
  // torch.nn.functional.log_softmax of ctc.py:107 (forward: subtracted at the gather; backward: the rows
  // start at -cf * softmax(x) because the posteriors of a frame sum to one)
  const float* row_lse;
  // meet-in-the-middle launch, wide rows beyond the cache, optional: xc[b][t][kXcStride] = the emissions a sweep's first
  // half gathered -- slot i = x[b][t][y_i] (target position i), slot L = x[b][t][blank] -- left there for the partner's
  // second half (ctc_mitm.h, stagers)
  const float* xc;
  // lane-exponent steps, optional: a word of pinned HOST memory where the repair launch leaves the number of utterances
  // it recomputed -- read by the NEXT call on this workspace, without a synchronisation (wfl_ctc_forward_backward)
  int32_t* host_repaired;
  // repair launch behind the meet-in-the-middle launch: that launch's token -- it raised ws.suspect to it if any
  // certificate had a doubt, otherwise the repair launch has nothing to look at (0: the word is not maintained)
  unsigned long long suspect_token;
  // meet-in-the-middle launch: number of labels the caller vouches for behind `targets` (0: unknown) -- what lets the
  // workgroups ask for their labels before the offsets have arrived (ctc_mitm.h)
  long long n_labels;
  // meet-in-the-middle launch: polls a wave waits for a hand-off before it gives up (ctc_mitm.h, mitm_give_up)
  int spin;
};
constexpr int kXcStride = 64;

// ---- pieces shared by the chains of the pipelined launches -------------------------------------------
__device__ __forceinline__ void coherent_store64(void* p, unsigned long long v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long coherent_load64(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coherent_load_f64(const double* p) {
  const unsigned long long bits = coherent_load64(p);
  double v;
  __builtin_memcpy(&v, &bits, 8);
  return v;
}

// one lane: publish nll[b] (and, when a workgroup of this launch reduces the loss, the done flag)
template <bool SIGNAL, bool REPAIR>
__device__ __forceinline__ void publish_nll(const CtcArgs& a, const CtcWs& w, int b, bool alive, double z2) {
  const float nllb = alive ? (float)(-z2 * 0.6931471805599453) : __builtin_inff();
  if (SIGNAL && (a.loss_out || REPAIR)) {  // device-coherently for the workgroup that reduces the loss
    // (ONLY this store: a plain store to the same word first would leave a dirty non-coherent line in
    // this XCD's L2 that the coherent store merges into instead of writing through -- measured: the
    // reducer then read stale values)
    __hip_atomic_store(reinterpret_cast<unsigned int*>(a.nll + b), __float_as_uint(nllb), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store((unsigned long long*)(a.ws + w.done) + b, a.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    a.nll[b] = nllb;
  }
}

// The certificate of the fast pipelined launch, evaluated by the launch that follows it: every block's
// gradient wave reproduced log2 Z from the two sweeps (sum_s alpha beta at its last frame) and folded it
// into a per-utterance minimum / maximum (16.16 fixed point, device-scope atomics; initialised by the
// utterance's alpha chain before it publishes its first checkpoint).  Pruning or overflow in the
// lane-exponent chains breaks the agreement with the chain's own Z.
constexpr long long kZDead = -(1ll << 62);
__device__ __forceinline__ long long z_fixed(double z2) { return z2 > -1.0e299 ? (long long)llrint(z2 * 65536.0) : kZDead; }
__device__ __forceinline__ bool utterance_rejected(const CtcArgs& a, const CtcWs& w, int u) {
  const int32_t* pbad = (const int32_t*)(a.ws + w.pbad) + u * 2;
  const double z2 = ((const double*)(a.ws + w.z2))[u];
  const long long* zmm = (const long long*)(a.ws + w.zloc) + (int64_t)u * 2;
  const long long zq = z_fixed(z2);
  // a wave of the launch gave up waiting (ctc_mitm.h, mitm_give_up): nothing it left behind is vouched for
  if (__hip_atomic_load((const int32_t*)(a.ws + w.perr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
  if (zq == kZDead && (((const unsigned long long*)(a.ws + w.dup))[u] >> 63)) return false;  // cannot be aligned: exact
  // 1.5e-4 in log2 units: a lost mass fraction of 1e-4 (the parity bar).  The lane-exponent blocks reproduce the
  // chain's log2 Z to 1-3e-5 on well-represented data (measured; fixed-point resolution 1.5e-5)
  constexpr long long tol = 10;
  return pbad[0] != 0 || pbad[1] != 0 || zq == kZDead || zmm[0] < zq - tol || zmm[1] > zq + tol;
}
__device__ __forceinline__ bool utterance_rejected_wave(const CtcArgs& a, const CtcWs& w, int u, int lane) {
  (void)lane;
  return utterance_rejected(a, w, u);  // (same words for every lane: wave-uniform)
}

// one wave: wait for the alpha chains that publish in this launch (all of them, or -- in the repair
// launch -- the rejected utterances only; nothing to do if there are none), then reduce the loss in a
// fixed order: mean_b(scale_b * nll_b), no extra launch, deterministic
__device__ __forceinline__ void reduce_loss_when_done(const CtcArgs& a, const CtcWs& w, int lane, bool rejected_only) {
  unsigned long long* done = (unsigned long long*)(a.ws + w.done);
  bool any = false;
  for (int u = lane; u < a.B; u += 64) any = any || !rejected_only || utterance_rejected(a, w, u);
  if (__builtin_amdgcn_ballot_w64(any) == 0) return;
  float part = 0.f;
  bool ok = true;
  for (int u = lane; u < a.B; u += 64) {
    if (!rejected_only || utterance_rejected(a, w, u)) {
      bool seen = false;
      for (int spin = 0; spin < (1 << 20) && !seen; ++spin) {
        seen = __hip_atomic_load(done + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.token;
        if (!seen) __builtin_amdgcn_s_sleep(16);
        // (a wave of the launch gave up: its utterance may never be published -- the repair launch reduces the loss)
        if (!seen && !rejected_only && (spin & 63) == 63 &&
            __hip_atomic_load((const int32_t*)(a.ws + w.perr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
          break;
      }
      ok = ok && seen;
      __hip_atomic_store(done + u, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sole consumer: clear
    }
    const float v = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(a.nll + u), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT));
    part += (a.loss_scale ? a.loss_scale[u] : 1.f) * v;
  }
  const float total = wave_all_sum(part);  // fixed lane order
  if (lane == 0 && a.loss_out) a.loss_out[0] = total / (float)a.B;
  if (!ok && lane == 0) {
    // not every utterance was published in time: say so (status word; in the first launch also the doubt word, so that
    // the repair launch recomputes the batch and reduces the loss again) -- no trap
    __hip_atomic_fetch_or((int32_t*)(a.ws + w.perr), rejected_only ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!rejected_only) coherent_store64(a.ws + w.suspect, a.token);
  }
}

// ------------------------------------------------------------------------------------------------
// log-domain chains: grid (B, 2) x 128.  Wave 0 runs the dependent chain and nothing else; wave 1 (on another
// SIMD of the same CU) feeds it: it gathers the emissions of block kk+2 from HBM, applies the scale /
// NaN policy / blank broadcast / lane masks and leaves ready (xb, xl) pairs in an LDS ring, two
// blocks ahead.  One s_barrier per 16-frame block.  This takes the VMEM instruction (~50 cycles for
// a lone wave), ~6 VALU and ~10 SALU of address arithmetic per frame off the critical wave.
// ------------------------------------------------------------------------------------------------
constexpr int kRing = 3;  // LDS ring depth in blocks: consumer at kk, producer at kk+2

struct ChainLdsT {
  float2 ring[kRing][kBlk][64];  // 24 KiB
  float2 ckbuf[2][64];           // checkpoint hand-off chain wave -> helper wave
  double offbuf[2];
  float refsum[kRing];           // sum of the per-frame references of the block in the ring slot (integer valued)
};
template <bool MAX>
__device__ __forceinline__ float fold16(const float (&v)[16], int lane);

// only_flagged != 0: repair pass -- run only for utterances whose fast-chain result was rejected.
// SIGNAL: publish ready[b][dir][block] (agent-scope release) after each checkpoint reached HBM, for the
// gradient waves of the pipelined step that are waiting for it.
template <bool SIGNAL, bool LSM = false, bool REPAIR = false>
__device__ __forceinline__ void ctc_log_chain_body(const CtcArgs& a, int b, int dir, int only_flagged, ChainLdsT& S) {
  auto& ring = S.ring;
  auto& ckbuf = S.ckbuf;
  auto& offbuf = S.offbuf;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = a.T, C = a.C, P = a.P;
  const int64_t o0 = a.offsets[b];
  const int L = (int)(a.offsets[b + 1] - o0);
  // this lane's label (reversed target for the beta sweep) and skip flag
  int y = -1, yprev = -1;
  if (lane < L) y = a.targets[o0 + (dir == 0 ? lane : L - 1 - lane)];
  if (lane >= 1 && lane - 1 < L) yprev = a.targets[o0 + (dir == 0 ? lane - 1 : L - lane)];
  const bool has_label = lane < L, has_blank = lane <= L;
  const bool skip = has_label && lane >= 1 && y != yprev;
  const float* xrow = a.x + (int64_t)b * T * C;
  const int col = has_label ? y : a.blank;
  const CtcWs w = ctc_ws_layout(a.B, T, P);
  const int NB = ctc_blocks(T);
  if (only_flagged && ((const int32_t*)(a.ws + w.flag))[b] == 0) return;  // uniform per workgroup

  // alpha walks t = 0..T-1; beta walks block by block from the last block to the first, frames
  // descending inside each block, so that its checkpoints fall on the same absolute 16-frame
  // boundaries as alpha's
  // helper wave: `issue` starts the 16 gathers of the kk-th processed block, `stage` (one loop
  // iteration = one block of chain work later) turns the landed values into ring entries, so the
  // HBM / L2 latency never sits in front of a barrier
  // Two helper waves alternate blocks (helper h owns the processed blocks kk with kk % 2 == h): in a
  // helper's own instruction stream a block's gathers are issued two chain blocks before they are
  // consumed and nothing newer is in flight at that point, so the compiler's s_waitcnt vmcnt(0)
  // in front of the first use costs nothing even when the rows come from HBM (cfg5: x is 524 MB).
  const int h = wave - 1;
  // pipelined step: a fourth wave does nothing but publish checkpoints -- its wait for the stores'
  // acknowledgement must not also wait for a helper's emission gathers
  constexpr int kFlusher = SIGNAL ? 3 : 1;
  float raw[kBlk];
  float lse_raw = 0.f;  // lane j < 16: log-sum-exp of the block's j-th processed frame (fused log_softmax)
  auto issue = [&](int kk) {
    const int k = dir == 0 ? kk : NB - 1 - kk;
    const int t0 = k * kBlk, n = min(kBlk, T - t0);
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      const int t = dir == 0 ? t0 + j : t0 + n - 1 - j;
      raw[j] = xrow[(int64_t)min(max(t, 0), T - 1) * C + col];  // clamped: valid address, unused past the block
    }
    if (LSM) {
      const int t = dir == 0 ? t0 + (lane & 15) : t0 + n - 1 - (lane & 15);
      lse_raw = a.row_lse[(int64_t)b * T + min(max(t, 0), T - 1)];
    }
  };
  // Per-frame references r_t = rint(largest score of the frame over the target's labels and the blank): the helper
  // hands the chain FACTORS 2^(x_t - r_t), the block's sum of references goes to the double offset.
  auto stage = [&](int kk) {
    const int k = dir == 0 ? kk : NB - 1 - kk;
    const int n = min(kBlk, T - k * kBlk);
    float xs[kBlk];
#pragma unroll
    for (int j = 0; j < kBlk; ++j) xs[j] = to_score(LSM ? raw[j] - readlane_f(lse_raw, j) : raw[j]);
    const float m = fold16<true>(xs, lane);  // every lane: the largest score of frame lane % 16 (lanes > L hold the blank's)
    const float rr = m > 0.5f * kNegBig ? rintf(m) : 0.f;
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      // the frame's FACTOR 2^(x_t - r_t) <= ~1.4 (0 for an impossible label): the chain multiplies, see below
      const float f = xs[j] > 0.5f * kNegBig ? __builtin_amdgcn_exp2f(xs[j] - readlane_f(rr, j)) : 0.f;
      const float fblank = readlane_f(f, L);
      ring[kk % kRing][j][lane] = make_float2(has_blank ? fblank : 0.f, has_label ? f : 0.f);
    }
    const float rs = wave_all_sum(lane < n ? rr : 0.f);
    if (lane == 0) S.refsum[kk % kRing] = rs;
  };
  if (wave == 1 || wave == 2) {
    if (h < NB) {
      issue(h);
      stage(h);
    }
    if (h + 2 < NB) issue(h + 2);
  }
  __syncthreads();

  // The chain itself runs in the PROBABILITY domain on doubles: alpha' = (alpha + shifted alpha) * factor, renormalised
  // by an exact power of two per block (offset in `off`, log2 units).  The fp32 log-add chain it replaces
  // (max + log2(1 + 2^-d) per arc, two transcendentals each) reproduced log Z to ~1e-4 nats over 1000 frames and the
  // posteriors to 6e-5 .. 1.3e-4 of the coefficient against the float64 oracle -- its per-frame errors do not average
  // out; a product of independently rounded factors does (2e-6).  Checkpoints keep their format (log2 of the state as
  // float2 + the double offset): the gradient blocks recompute 16 frames from them as before.
  double ab = (lane == 0) ? 1.0 : 0.0;  // virtual slot "before the first frame"
  double al = 0.0;
  double off = 0.0;
  const double skd = skip ? 1.0 : 0.0;
  auto shr1_d = [&](double v) {  // lane i receives lane i-1's value, lane 0 receives 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  };
  auto log2_ck = [&](double v) {  // log2 of a checkpointed state as a float (sentinel for 0): exponent + log2(mantissa)
    if (!(v > 0.0)) return kNegBig;
    const int ex = ((__double2hiint(v) >> 20) & 0x7ff) - 1023;
    const float mant = (float)__hiloint2double((__double2hiint(v) & 0x800fffff) | 0x3ff00000, __double2loint(v));
    return (float)ex + __builtin_amdgcn_logf(mant);
  };
  float2* ck = (float2*)(a.ws + w.ck) + ((int64_t)(b * 2 + dir) * NB) * P;
  double* offs = (double*)(a.ws + w.off) + (int64_t)(b * 2 + dir) * NB;
  unsigned long long* ready = (unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + dir) * NB;
  auto flush_checkpoint = [&](int kk) {  // helper wave: LDS -> HBM, one block behind the chain
    if (!SIGNAL) {
      if (lane < P) ck[(int64_t)kk * P + lane] = ckbuf[kk & 1][lane];
      if (lane == 0) offs[kk] = offbuf[kk & 1];
    } else {
      // The consumers run on other CUs / XCDs of the same launch.  A release fence at agent scope
      // would write back this XCD's whole L2 -- including the gradient rows streaming through it --
      // once per block (measured: 10x slower).  Instead the few checkpoint words themselves are
      // stored device-coherently (agent-scope relaxed atomics bypass the non-coherent L2 state),
      // the wave waits for their acknowledgement, and only then raises the flag the same way.
      if (lane < P) {
        const float2 v = ckbuf[kk & 1][lane];
        unsigned long long bits;
        __builtin_memcpy(&bits, &v, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&ck[(int64_t)kk * P + lane]), bits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        unsigned long long bits;
        const double o = offbuf[kk & 1];
        __builtin_memcpy(&bits, &o, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&offs[kk]), bits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): the stores are acknowledged
      if (lane == 0) __hip_atomic_store(&ready[kk], a.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  float2 e[kBlk], en[kBlk];
  float rs_cur = 0.f, rs_next = 0.f;  // reference sums of the block in `e` / `en`
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < kBlk; ++j) e[j] = ring[0][j][lane];
    rs_cur = S.refsum[0];
  }
  for (int kk = 0; kk < NB; ++kk) {
    if (wave == 0) {
      const int k = dir == 0 ? kk : NB - 1 - kk;
      const int n = min(kBlk, T - k * kBlk);
      if (kk + 1 < NB) {  // block kk+1 is already staged (the helper runs two blocks ahead)
#pragma unroll
        for (int j = 0; j < kBlk; ++j) en[j] = ring[(kk + 1) % kRing][j][lane];
        rs_next = S.refsum[(kk + 1) % kRing];
      }
      if (kk > 0) {  // renormalise: the largest state's binary exponent -> double offset (exact)
        const double mx = fmax(ab, al);
        const int ex = mx > 0.0 ? ((__double2hiint(mx) >> 20) & 0x7ff) - 1023 : -(1 << 20);
        const int em = wave_all_max_int(ex);
        if (em > -(1 << 20)) {
          ab = __builtin_amdgcn_ldexp(ab, -em);
          al = __builtin_amdgcn_ldexp(al, -em);
          off += (double)em;
        }
      }
      ckbuf[kk & 1][lane] = make_float2(log2_ck(ab), log2_ck(al));  // checkpoint: state BEFORE this block
      if (lane == 0) offbuf[kk & 1] = off;
      auto frame = [&](const float2 f) {
        const double pal = shr1_d(al);
        const double nb = ab + pal;
        const double nl = fma(skd, pal, al + ab);  // al + ab (+ pal when the skip arc exists): no select on a double
        ab = nb * (double)f.x;
        al = nl * (double)f.y;
      };
      if (n == kBlk) {  // straight-line: a per-frame branch costs the lone wave more than the frame's math
#pragma unroll
        for (int j = 0; j < kBlk; ++j) frame(e[j]);
      } else {
#pragma unroll
        for (int j = 0; j < kBlk; ++j)
          if (j < n) frame(e[j]);
      }
#pragma unroll
      for (int j = 0; j < kBlk; ++j) e[j] = en[j];
      off += (double)rs_cur;  // (the block's references: part of every state's score from here on)
      rs_cur = rs_next;
    } else {
      if (kk > 0 && wave == kFlusher) flush_checkpoint(kk - 1);
      if ((kk & 1) == h && wave <= 2) {
        if (kk + 2 < NB) stage(kk + 2);  // issued two iterations ago
        if (kk + 4 < NB) issue(kk + 4);
      }
    }
    __syncthreads();
  }
  if (wave == kFlusher) flush_checkpoint(NB - 1);
  if (dir == 0 && wave == 0) {
    // logZ = LSE(alpha_{T-1}[2L], alpha_{T-1}[2L-1])   (ctc.py:21 accept states)
    auto readlane_d = [&](double v, int l) {
      return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    const double a_last = readlane_d(ab, L);
    const double l_last = L > 0 ? readlane_d(al, L - 1) : 0.0;
    if (lane == 0) {
      const double tot = a_last + l_last;
      const bool alive = tot > 0.0;
      const double z2 = alive ? log2(tot) + off : -1.0e300;
      ((double*)(a.ws + w.z2))[b] = z2;
      publish_nll<SIGNAL, REPAIR>(a, w, b, alive, z2);
    }
    if (SIGNAL && !REPAIR && a.loss_out && b == 0) reduce_loss_when_done(a, w, lane, false);
  }
}

__global__ void __launch_bounds__(192) ctc_log_chain_kernel(CtcArgs a, int only_flagged) {
  __shared__ ChainLdsT S;
  ctc_log_chain_body<false>(a, blockIdx.x, blockIdx.y, only_flagged, S);
}

// ------------------------------------------------------------------------------------------------
// "Lane-exponent" arithmetic: the sweeps of the meet-in-the-middle launch (ctc_mitm.h).
//
// The log-domain frame is a chain of ~15 DEPENDENT instructions with 4 transcendentals (~155 cycles
// for a lone wave).  Here a state is a float mantissa with an integer exponent PER LANE (shared by
// the lane's blank / label pair): value = m * 2^(e_lane + off).  A frame is
//     pb' = fb (pb + g q),   pl' = fl (pl + pb + gs q),   q = pl of lane i-1
// -- five instructions, no transcendental (see frames4) -- with g = 2^(e[i-1] - e[i]) fixed for a
// 16-frame block and emission factors f = 2^(x*log2e - r_t) <= 2^0.5 relative to a per-FRAME reference
// r_t (the rounded largest target-label score of the frame).  Once per block each lane
// renormalises ITS OWN exponent, and a prefix-max scan enforces e[i] >= e[i-1] - kGap so that mass
// flowing up the lanes cannot overflow inside a block (growth <= 2^(16*(kGap+1.6)) < 2^127); a lane
// pulled up by the clamp only loses mass that is 2^-126 below what its predecessor is about to hand
// it.  A wave-uniform power-of-two scale instead of the per-lane exponents does NOT work on the
// benchmark's data: with unnormalised scores and T >> L the alpha mass piles up at the last states
// and the beta mass at the first ones, 2^328 apart at cfg2, and the cells that carry the posterior
// are flushed (measured); fp64 with a wave-uniform scale works but its dependent latency leaves the
// chain at 47 us (measured; scratch/ctc_kernels_fp64_chain.hip.txt).
//
// A lone wave is ISSUE-bound, not latency-bound (measured: ~4.5 cycles per VALU instruction whether
// dependent or not), so the design minimises the chain wave's instruction count: one wave runs the
// chain and nothing else; stager waves prepare whole blocks of factors (gather, NaN policy, the 16
// per-frame maxima in one fold, exp2, blank broadcast) in an LDS ring; a flusher wave publishes the raw
// (mantissa, exponent) checkpoints.  No barrier inside the sweep: the waves meet in LDS mailboxes (ds_write /
// ds_read of a wave execute in order, so "data, then flag" needs no wait).
//
// What cannot be represented is flushed to zero or overflows; the results are certified: every gradient block
// reproduces log2 Z and checks that the posteriors of its frames sum to one (ctc_mitm_emit_block);
// ctc_repair_kernel evaluates.
// ------------------------------------------------------------------------------------------------
constexpr int kGap = 5;       // max exponent drop from lane i-1 to lane i
constexpr int kEmptyE = -(1 << 28);

__device__ __forceinline__ int wave_shr1_i(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
// (dpp_i32: device_common.h)
// inclusive prefix maximum over the 64 lanes (row_shr scan + row broadcasts).  v_max_i32 with a DPP source:
// a lane whose source lane does not exist (or whose row is masked) is simply not written, which is the
// identity of a running maximum -- one instruction per step instead of mov / mov_dpp / max.  The s_nops
// are the two wait states a DPP read needs after a VALU write of the same register.
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
// acc0 += lane[i-1].src * c0, acc1 += lane[i-1].src * c1 (lane 0: unchanged)
__device__ __forceinline__ void fmac2_shr1(float& acc0, float& acc1, float src, float c0, float c1) {
  asm volatile(
      "s_nop 1\n\tv_fmac_f32_dpp %0, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf"
      : "+&v"(acc0), "+&v"(acc1)
      : "v"(src), "v"(c0), "v"(c1));
}

// Reduce 16 per-lane values over the 64 lanes at once: instead of 16 wave reductions (16 x 18 instructions)
// the lanes fold the 16 values pairwise -- after the exchange with lane^1 a lane only keeps the 8
// values whose index has its bit 0, after lane^2 four, ... -- so that lane l (EVERY lane: l % 16)
// ends up with the 64-lane total (MAX: maximum) of value l % 16.  ~50 instructions:
//   levels 1, 2 (partner lane^1, lane^2: quad_perm): two selects (what to keep, what to send) and one DPP
//     operation per pair, the selects of the next pair issued in front of the operation (VALU write -> DPP read
//     of the same register needs two wait states);
//   levels 3, 4 (within the row of 16): a lane receives from lane+4 / lane+8 (row_ror:12 / row_ror:8) instead of
//     lane^4 / lane^8 -- just as good: the sender has the opposite bit, so what it does not keep is what the
//     receiver keeps, and the four lanes i, i+4, i+8, i+12 cover the row's four quads -- and which value a lane
//     keeps is decided by the DPP bank mask (a bank is a quad of lanes: bits 2, 3 of the lane), two masked
//     operations per pair and no select;
//   rows: v_permlane16_swap / v_permlane32_swap (one VALU instruction each where __shfl_xor is a ds_bpermute
//     round trip).
// The leading s_nop covers inputs written by the instruction in front of the block.
#define WFL_FOLD16_BODY(OP)  \
  "s_nop 1\n\t"  \
  "v_cndmask_b32_e64 %[t0], %[v1], %[v0], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t1], %[v0], %[v1], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t2], %[v3], %[v2], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t3], %[v2], %[v3], %[m0]\n\t"  \
  OP " %[a0], %[t0], %[t1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t0], %[v5], %[v4], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t1], %[v4], %[v5], %[m0]\n\t"  \
  OP " %[a1], %[t2], %[t3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t2], %[v7], %[v6], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t3], %[v6], %[v7], %[m0]\n\t"  \
  OP " %[a2], %[t0], %[t1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t0], %[v9], %[v8], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t1], %[v8], %[v9], %[m0]\n\t"  \
  OP " %[a3], %[t2], %[t3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t2], %[v11], %[v10], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t3], %[v10], %[v11], %[m0]\n\t"  \
  OP " %[a4], %[t0], %[t1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t0], %[v13], %[v12], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t1], %[v12], %[v13], %[m0]\n\t"  \
  OP " %[a5], %[t2], %[t3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t2], %[v15], %[v14], %[m0]\n\t"  \
  "v_cndmask_b32_e64 %[t3], %[v14], %[v15], %[m0]\n\t"  \
  OP " %[a6], %[t0], %[t1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "s_nop 0\n\t"  \
  OP " %[a7], %[t2], %[t3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t0], %[a1], %[a0], %[m1]\n\t"  \
  "v_cndmask_b32_e64 %[t1], %[a0], %[a1], %[m1]\n\t"  \
  "v_cndmask_b32_e64 %[t2], %[a3], %[a2], %[m1]\n\t"  \
  "v_cndmask_b32_e64 %[t3], %[a2], %[a3], %[m1]\n\t"  \
  OP " %[a0], %[t0], %[t1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t0], %[a5], %[a4], %[m1]\n\t"  \
  "v_cndmask_b32_e64 %[t1], %[a4], %[a5], %[m1]\n\t"  \
  OP " %[a2], %[t2], %[t3] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"  \
  "v_cndmask_b32_e64 %[t2], %[a7], %[a6], %[m1]\n\t"  \
  "v_cndmask_b32_e64 %[t3], %[a6], %[a7], %[m1]\n\t"  \
  OP " %[a4], %[t0], %[t1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"  \
  "s_nop 0\n\t"  \
  OP " %[a6], %[t2], %[t3] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"  \
  OP " %[a1], %[a0], %[a0] row_ror:12 row_mask:0xf bank_mask:0x5\n\t"  \
  OP " %[a1], %[a2], %[a2] row_ror:12 row_mask:0xf bank_mask:0xa\n\t"  \
  OP " %[a3], %[a4], %[a4] row_ror:12 row_mask:0xf bank_mask:0x5\n\t"  \
  OP " %[a3], %[a6], %[a6] row_ror:12 row_mask:0xf bank_mask:0xa\n\t"  \
  "s_nop 1\n\t"  \
  OP " %[a0], %[a1], %[a1] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"  \
  OP " %[a0], %[a3], %[a3] row_ror:8 row_mask:0xf bank_mask:0xc"
template <bool MAX>
__device__ __forceinline__ float fold16(const float (&v)[16], int lane) {
  float a0, a1, a2, a3, a4, a5, a6, a7, t0, t1, t2, t3;
  const unsigned long long m0 = 0xaaaaaaaaaaaaaaaaull, m1 = 0xccccccccccccccccull;  // lanes with bit 0 / bit 1 set
#define WFL_FOLD16_OPERANDS                                                                                             \
  : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6),    \
    [a7] "=&v"(a7), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)                                      \
  : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]),    \
    [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]), [v11] "v"(v[11]), [v12] "v"(v[12]),              \
    [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15]), [m0] "s"(m0), [m1] "s"(m1)
  if (MAX)
    asm(WFL_FOLD16_BODY("v_max_f32_dpp") WFL_FOLD16_OPERANDS);
  else
    asm(WFL_FOLD16_BODY("v_add_f32_dpp") WFL_FOLD16_OPERANDS);
#undef WFL_FOLD16_OPERANDS
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  v2u sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0), __float_as_uint(a0), false, false);
  float t = MAX ? vmax(__uint_as_float(sw.x), __uint_as_float(sw.y)) : __uint_as_float(sw.x) + __uint_as_float(sw.y);
  sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  t = MAX ? vmax(__uint_as_float(sw.x), __uint_as_float(sw.y)) : __uint_as_float(sw.x) + __uint_as_float(sw.y);
  return t;
}
__device__ __forceinline__ float fold16_sum(const float (&v)[16], int lane) { return fold16<false>(v, lane); }

// LDS mailboxes between the waves of a workgroup.  DS instructions of a wave execute in issue order and
// the LDS serves one instruction at a time, so "data, then flag" from the producer and "flag, then data"
// from the consumer need no wait in between -- only the compiler has to keep the order.
// (The casts to the LDS address space matter: a volatile access through a generic pointer is compiled to
// a system-coherent FLAT instruction with an immediate s_waitcnt -- measured: 4 of them per block cost the
// chain wave more than its 16 frames.)
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ int lds_peek(const int* p) { return *(const volatile lds_int_t*)(const lds_int_t*)p; }
__device__ __forceinline__ void lds_post(int* p, int v) {
  asm volatile("" ::: "memory");
  *(volatile lds_int_t*)(lds_int_t*)p = v;
}

// The five-instruction frame of the lane-exponent arithmetic (ctc_mitm.h): packed multiplies for the two coefficient
// products and the two pb products, the two DPP multiply-adds, and the fma that adds fl * pl.  State pair P = (pb, pl)
// and the running pair TT swap roles every frame; the two packed multiplies between the write of pl and its DPP read
// are the required wait states.  v[6:7] are scratch.
#define WFL_FRAME(P, PH, TT, TL, TH, F, FY)                                   \
  "v_pk_mul_f32 v[6:7], " F ", %[G]\n\t"                                      \
  "v_pk_mul_f32 " TT ", " F ", " P " op_sel_hi:[1,0]\n\t"                     \
  "v_fmac_f32_dpp " TL ", " PH ", v6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp " TH ", " PH ", v7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32 " TH ", " FY ", " PH "\n\t"

// ------------------------------------------------------------------------------------------------
// gradient: one wave per (utterance, 16-frame block), 4 waves per workgroup
// ------------------------------------------------------------------------------------------------
// Fused log_softmax backward, first half: rows[j*C + c] = -cf * softmax(x)[t0 + j, c] for the block's n
// frames (contiguous in x).  Flat and vectorised, eight loads in flight per lane: the wave has
// nothing else to hide the latency behind.  lse_blk: lane j < 16 holds the log-sum-exp of frame t0 + j.
__device__ __forceinline__ void lsm_seed_rows(float* rows, const float* __restrict__ xsrc, int n, int C, float lse_blk,
                                              float cf_row, int lane) {
  const int total = n * C;
  const unsigned magic = (unsigned)((0x100000000ull + (unsigned)C - 1) / (unsigned)C);  // i / C for i * C < 2^32
  auto seed = [&](float xv, float l) {
    const float e = __expf((xv == xv ? xv : WFL_NEG_INF) - l);
    return l > WFL_NEG_INF ? -cf_row * e : 0.f;  // a frame without finite scores: no softmax term
  };
  constexpr int U = 8;
  if ((C & 3) == 0 && (((uintptr_t)xsrc) & 15) == 0) {
    const int n4 = total >> 2;
    for (int i0 = lane; i0 < n4; i0 += 64 * U) {
      float4 v[U];
      float l[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + 64 * u, n4 - 1);
        v[u] = ((const float4*)xsrc)[i];
        l[u] = __shfl(lse_blk, (int)__umulhi((unsigned)i * 4u, magic), 64);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 64 * u;
        if (i < n4) ((float4*)rows)[i] = float4{seed(v[u].x, l[u]), seed(v[u].y, l[u]), seed(v[u].z, l[u]), seed(v[u].w, l[u])};
      }
    }
  } else {
    for (int i0 = lane; i0 < total; i0 += 64 * U) {
      float v[U], l[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + 64 * u, total - 1);
        v[u] = xsrc[i];
        l[u] = __shfl(lse_blk, (int)__umulhi((unsigned)i, magic), 64);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 64 * u;
        if (i < total) rows[i] = seed(v[u], l[u]);
      }
    }
  }
}

// COMPACT gradient tile (wide rows): the wave accumulates a block's posteriors in [16][65] floats -- one slot per
// target position (the slot of a repeated label is the lane of its first occurrence), slot 63 the blank, slot 64
// zero -- next to a column -> slot byte map, and the dense rows are expanded while they are written:
//   dx[t0 + j, c] = tile[j][slot(c)]  - cf * softmax(x)[c] (fused log_softmax).
// The dense LDS tile [16][C] of the narrow case would leave a single gradient workgroup per CU from C = 160 on and
// does not fit at all beyond C = 602.
// Tile rows are kCS = 65 floats: slots 0..62 target positions, 63 the blank, 64 a constant zero that every column
// without a slot maps to -- the expansion reads the tile unconditionally (four ds_read_b32 + one global store per
// float4; with a "no slot" test per element it was five times as many instructions).
constexpr int kCS = 65;
constexpr int kCTileBytes = kBlk * kCS * 4;  // 4160
__host__ __device__ __forceinline__ size_t compact_wave_bytes(int C) { return (size_t)kCTileBytes + ((C + 15) & ~15); }
__device__ __forceinline__ void compact_init(float* tile, unsigned char* cmap, int C, int lane) {
  for (int i = lane; i < kBlk * kCS; i += 64) tile[i] = 0.f;
  for (int i = lane; i < ((C + 15) & ~15) / 4; i += 64) ((unsigned int*)cmap)[i] = 0x40404040u;  // 64: the zero slot
}
__device__ __forceinline__ void compact_expand(const float* tile, const unsigned char* cmap, float* __restrict__ dst,
                                               const float* __restrict__ xsrc, float lse_blk, int n, int C, float cf,
                                               bool alive, bool soft, int lane) {
  auto softterm = [&](float xv, float l) { return l > WFL_NEG_INF ? cf * __expf((xv == xv ? xv : WFL_NEG_INF) - l) : 0.f; };
  // (rows of dx start at any 4-byte boundary: global dwordx4 accesses only need dword alignment, the type says so)
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int c4n = C >> 2;
#pragma unroll 2
  for (int j = 0; j < n; ++j) {  // row after row: whole rows leave in address order
    const float l = soft ? readlane_f(lse_blk, j) : 0.f;
    const float* tr = tile + j * kCS;
    float* drow = dst + (int64_t)j * C;
    const float* xr = xsrc + (int64_t)j * C;
    for (int c4 = lane; c4 < c4n; c4 += 64) {
      const unsigned m4 = alive ? ((const unsigned int*)cmap)[c4] : 0x40404040u;
      f4u o = {tr[m4 & 255u], tr[(m4 >> 8) & 255u], tr[(m4 >> 16) & 255u], tr[m4 >> 24]};
      if (soft) {
        const f4u xv = *reinterpret_cast<const f4u*>(xr + 4 * c4);
        o.x -= softterm(xv.x, l), o.y -= softterm(xv.y, l), o.z -= softterm(xv.z, l), o.w -= softterm(xv.w, l);
      }
      *reinterpret_cast<f4u*>(drow + 4 * c4) = o;
    }
    const int c = 4 * c4n + lane;  // the last C % 4 columns
    if (c < C) {
      float v = tr[alive ? (int)cmap[c] : 64];
      if (soft) v -= softterm(xr[c], l);
      drow[c] = v;
    }
  }
}

// PIPE: the pipelined step -- wait for the two checkpoints of block k to be published by the chain
// workgroups of the same launch, and normalise the posteriors by the Z the block itself reproduces
// (sum_s alpha(s) beta(s) at its last frame; the certificate's identity) instead of the log Z that the
// alpha chain only knows when it has finished.
template <bool PIPE, bool LSM = false, bool COMPACT = false>
__device__ __forceinline__ void ctc_grad_body(const CtcArgs& a, bool valid, int b, int k, const float* __restrict__ coef,
                                              const float* __restrict__ gout, float* __restrict__ dx, char* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = a.T, C = a.C, P = a.P;
  const int NB = ctc_blocks(T);
  const CtcWs w = ctc_ws_layout(a.B, T, P);
  // [16][C] gradient rows + [C] label counts per wave, or (COMPACT, wide rows) the compact tile + column map
  char* wbase = smem + (size_t)wave * (COMPACT ? compact_wave_bytes(C) : (size_t)(kBlk + 1) * C * 4);
  float* rows = (float*)wbase;
  int* cnt = (int*)(rows + (size_t)kBlk * C);                             // (dense tile only)
  unsigned char* cmap = (unsigned char*)(wbase + (size_t)kCTileBytes);  // (COMPACT only)
  const int t0 = k * kBlk, n = min(kBlk, T - t0);
  bool live = valid && (PIPE || a.nll[b] < __builtin_inff());  // no accepting path: zero gradient
  constexpr bool lsm = PIPE && LSM;  // fused log_softmax (raw scores in x)
  const float cf_row = (coef ? coef[valid ? b : 0] : 1.f) * (gout ? gout[0] : 1.f);
  float lse_blk = 0.f;  // lane j < 16: log-sum-exp of frame t0 + j
  bool alive_blk = true;  // (COMPACT) the block carries posterior mass
  if (valid) {
    if (COMPACT)
      compact_init(rows, cmap, C, lane);
    else
      for (int i = lane; i < (kBlk + 1) * C; i += 64) rows[i] = 0.f;  // (int 0 == float 0 bit pattern)
    if (lsm) {
      // d loss / d raw score = g - softmax(x) * sum_c g, and the posteriors of a frame sum to one:
      // the rows start at -cf * softmax(x) (done before waiting: it does not depend on the chains)
      lse_blk = a.row_lse[(int64_t)b * T + min(t0 + (lane & 15), T - 1)];
      if (!COMPACT) lsm_seed_rows(rows, a.x + ((int64_t)b * T + t0) * C, n, C, lse_blk, cf_row, lane);
    }
  }
  if (PIPE && valid) {
    // alpha checkpoint k and beta checkpoint NB-1-k: published by wave 1 of the two chain workgroups
    // (flags are compared with a 64-bit token that is unique to this launch: the workspace needs no clearing)
    const unsigned long long* ra = (const unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + 0) * NB + k;
    const unsigned long long* rb = (const unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + 1) * NB + (NB - 1 - k);
    // poll with relaxed device-scope loads (an acquire per poll would invalidate this XCD's L2 every
    // time: measured 35x slower); the checkpoints are then read device-coherently as well
    int ok = 0;
    for (int spin = 0; spin < (1 << 20); ++spin) {  // (bounded: a lost signal must not hang the GPU)
      ok = __hip_atomic_load(ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.token &&
           __hip_atomic_load(rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.token;
      if (ok) break;
      __builtin_amdgcn_s_sleep(64);
    }
    if (!ok) {  // ~2 s without the signal: never expected (in-order dispatch puts every chain ahead of the
                // waiting waves).  Fail loudly rather than return a silently wrong gradient.
      if (lane == 0) atomicOr((int32_t*)(a.ws + w.perr), 1);
      __builtin_trap();
    }
  }
  if (live) {
    const int64_t o0 = a.offsets[b];
    const int L = (int)(a.offsets[b + 1] - o0);
    const int y = lane < L ? a.targets[o0 + lane] : -1;
    const int yprev = (lane >= 1 && lane - 1 < L) ? a.targets[o0 + lane - 1] : -1;
    const int ynext = lane + 1 < L ? a.targets[o0 + lane + 1] : -1;
    const bool has_label = lane < L, has_blank = lane <= L;
    const bool skip = has_label && lane >= 1 && y != yprev;  // label i-1 -> label i
    const bool skipn = lane + 1 < L && ynext != y;           // label i -> label i+1
    const int col = has_label ? y : a.blank;
    const float* xrow = a.x + (int64_t)b * T * C;
    // A label that occurs once in the target (and is not the blank column) owns its gradient column:
    // plain ds_write instead of ds_add_f32, which costs ~1 LDS cycle per active lane (PMC: 42 cycles
    // per wave-instruction with 45 lanes).  One counting atomic per block finds the duplicates.
    bool dup;
    int slot = lane;  // COMPACT: lane of the label's first occurrence
    if (COMPACT) {
      bool dupl = false;
      for (int j = 0; j < L; ++j) {
        const bool same = __builtin_amdgcn_readlane(y, j) == y;
        dupl = dupl || (same && j != lane);
        if (same && j < slot) slot = j;
      }
      dup = has_label && (dupl || y == a.blank);
      if (y == a.blank) slot = 63;
      if (has_label && slot == lane) cmap[y] = (unsigned char)lane;
      if (lane == 0) cmap[a.blank] = 63;
    } else {
      if (has_label) atomicAdd(&cnt[y], 1);
      dup = has_label && (cnt[y] > 1 || y == a.blank);
    }
    const bool uniq = has_label && !dup;
    float xl[kBlk], xb[kBlk];
#pragma unroll
    for (int j = 0; j < kBlk; ++j) xl[j] = xrow[(int64_t)min(t0 + j, T - 1) * C + col];  // all 16 gathers in flight
#pragma unroll
    for (int j = 0; j < kBlk; ++j) xl[j] = to_score(lsm ? xl[j] - readlane_f(lse_blk, j) : xl[j]);
    // The block in the probability domain on doubles, with per-frame references as in the chain (ctc_log_chain_body):
    // alpha'_j beta~'_j = alpha_j beta~_j 2^-R for every frame j < n, R = the sum of the block's references.  The fp32
    // log-add recursions this replaces carried the states that HOLD the posterior mass at ~2^-100 of the wave's largest
    // state (random scores: alpha peaks at the late states, beta at the early ones, their product in between), i.e.
    // rounded every operation at ulp(100) = 7.6e-6: posteriors off by 2..4e-5 mid-utterance, 1.3e-4 at worst
    // (measured against the float64 oracle).  A double does not care where the mass sits.
    float ref_sum;
    {
      const float m = fold16<true>(xl, lane);  // every lane: the largest score of frame lane % 16 (lanes > L: the blank's)
      const float rr = m > 0.5f * kNegBig ? rintf(m) : 0.f;
#pragma unroll
      for (int j = 0; j < kBlk; ++j) {  // (xb, xl hold FACTORS 2^(x - r) from here on: 0 for impossible / absent states)
        const float f = xl[j] > 0.5f * kNegBig ? __builtin_amdgcn_exp2f(xl[j] - readlane_f(rr, j)) : 0.f;
        xb[j] = has_blank ? readlane_f(f, L) : 0.f;
        xl[j] = has_label ? f : 0.f;
      }
      ref_sum = wave_all_sum(lane < n ? rr : 0.f);
    }
    const float2* cka = (const float2*)(a.ws + w.ck) + ((int64_t)(b * 2 + 0) * NB) * P;
    const float2* ckb = (const float2*)(a.ws + w.ck) + ((int64_t)(b * 2 + 1) * NB) * P;
    const double* offa = (const double*)(a.ws + w.off) + (int64_t)(b * 2 + 0) * NB;
    const double* offb = (const double*)(a.ws + w.off) + (int64_t)(b * 2 + 1) * NB;
    const double z2 = PIPE ? 0.0 : ((const double*)(a.ws + w.z2))[b];
    // alpha checkpoint k: state before frame t0.  beta processed blocks NB-1..0, so its checkpoint
    // before block k has processing index NB-1-k: the full beta of frame t0+n (mirrored lanes:
    // blank state 2i <-> reversed position L-i; label of position i <-> L-1-i).
    auto load_ck = [&](const float2* p) {  // PIPE: written by another CU during this launch
      if (!PIPE) return *p;
      const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float2 v;
      __builtin_memcpy(&v, &bits, 8);
      return v;
    };
    const float2 ca = lane < P ? load_ck(&cka[(int64_t)k * P + lane]) : make_float2(kNegBig, kNegBig);
    float bb = lane <= L ? load_ck(&ckb[(int64_t)(NB - 1 - k) * P + (L - lane)]).x : kNegBig;
    float bl = lane < L ? load_ck(&ckb[(int64_t)(NB - 1 - k) * P + (L - 1 - lane)]).y : kNegBig;
    if (PIPE && lane == 0) {  // this wave was the only consumer of the two flags: leave them cleared
      unsigned long long* rdy = (unsigned long long*)(a.ws + w.ready);
      __hip_atomic_store(rdy + (int64_t)(b * 2 + 0) * NB + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rdy + (int64_t)(b * 2 + 1) * NB + (NB - 1 - k), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // posterior_t(s) = alpha_t(s) beta~_t(s) scale,  scale = 2^(off_alpha(k) + off_beta(k) + R - log2 Z)
    const float cf = (coef ? coef[b] : 1.f) * (gout ? gout[0] : 1.f);
    auto from_log2 = [](float lg) -> double {  // a checkpointed state: 2^lg, 0 for the sentinel
      if (!(lg > 0.5f * kNegBig)) return 0.0;
      const float fl = floorf(lg);
      return __builtin_amdgcn_ldexp((double)__builtin_amdgcn_exp2f(lg - fl), (int)fl);
    };
    auto shr1_d = [](double v) {  // lane i receives lane i-1's value, lane 0 receives 0
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
      return __hiloint2double(hi, lo);
    };
    auto shl1_d = [](double v) {  // lane i receives lane i+1's value, lane 63 receives 0
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
      return __hiloint2double(hi, lo);
    };
    double pa_b[kBlk], pa_l[kBlk];
    double ab = from_log2(ca.x), al = from_log2(ca.y);
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {  // alpha forward through the block, kept in registers
      const double pal = shr1_d(al);
      const double nb = ab + pal;
      const double nl = al + (skip ? nb : ab);
      ab = nb * (double)xb[j];
      al = nl * (double)xl[j];
      pa_b[j] = ab, pa_l[j] = al;
    }
    double bbd = from_log2(bb), bld = from_log2(bl);
    double scale;
    if (PIPE) {
      // the block's own Z at its last frame: sum_s alpha_{n-1}(s) [transition-propagated beta](s)
      const double tb0 = bbd + bld;
      const double tbn0 = shl1_d(tb0), bbn0 = shl1_d(bbd);  // (both shifts outside any divergent select: see below)
      const double tl0 = bld + (skipn ? tbn0 : bbn0);
      double zs = 0.0;
#pragma unroll
      for (int j = 0; j < kBlk; ++j)
        if (j == n - 1) zs = pa_b[j] * tb0 + pa_l[j] * tl0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) zs += __shfl_xor(zs, o, 64);
      scale = zs > 0.0 && zs < 1.0e300 ? 1.0 / zs : 0.0;  // (0: no accepting path through this block)
      alive_blk = scale > 0.0;
    } else {
      const double e = offa[k] + offb[NB - 1 - k] - z2 + (double)ref_sum;
      scale = (z2 > -1.0e299 && e < 1000.0) ? exp2(e) : 0.0;
    }
    float gbv[kBlk];
#pragma unroll
    for (int j = 0; j < kBlk; ++j) gbv[j] = 0.f;
#pragma unroll
    for (int j = kBlk - 1; j >= 0; --j) {  // beta backwards, in the forward lane mapping
      if (j < n) {
        // blank i -> blank i, label i.  label i -> label i, blank i+1 (and label i+1 if allowed):
        // bl + bb[i+1] + bl[i+1] = bl + tb[i+1] -- the neighbour's freshly computed value
        const double tb = bbd + bld;
        // (both shifts are taken unconditionally: a DPP inside a divergent branch would read
        // neighbours that are masked off)
        const double tbn = shl1_d(tb), bbn = shl1_d(bbd);
        const double tl = bld + (skipn ? tbn : bbn);
        const float gb = (float)(pa_b[j] * tb * scale);
        const float gl = (float)(pa_l[j] * tl * scale);
        // blank column: the 16 frames' lane values are summed over the wave together after the loop
        // (per-lane ds_add_f32 instead was measured at 49 us for the kernel vs 28 us: LDS float atomics
        // serialise per active lane; one wave reduction per frame costs 18 instructions x 16)
        gbv[j] = gb;
        if (COMPACT) {
          if (uniq) rows[j * kCS + lane] = gl * cf;
          if (dup && gl != 0.f) atomicAdd(&rows[j * kCS + slot], gl * cf);
        } else {
          if (uniq) rows[j * C + y] = (lsm ? rows[j * C + y] : 0.f) + gl * cf;  // sole writer of this column
          if (dup && gl != 0.f) atomicAdd(&rows[j * C + y], gl * cf);
        }
        bbd = tb * (double)xb[j];
        bld = tl * (double)xl[j];
      }
    }
    const float gtot = fold16_sum(gbv, lane);  // lane l < 16: blank posterior of frame l
    if (lane < n && gtot != 0.f) atomicAdd(&rows[COMPACT ? lane * kCS + 63 : lane * C + a.blank], gtot * cf);
    if (lsm && !alive_blk && !COMPACT)  // no accepting path: zero gradient, softmax term included
      for (int i = lane; i < kBlk * C; i += 64) rows[i] = 0.f;
  }
  // (rows are private to the wave: LDS operations of one wave complete in order, no barrier needed)
  if (valid && COMPACT) {
    const bool alive = live && alive_blk;
    compact_expand(rows, cmap, dx + ((int64_t)b * T + t0) * C, a.x + ((int64_t)b * T + t0) * C, lse_blk, n, C, cf_row, alive,
                   lsm && alive, lane);
  } else if (valid) {  // the dense rows of a block are contiguous in dx: one coalesced copy (zeros included)
    float* dst = dx + ((int64_t)b * T + t0) * C;
    const int total = n * C;
    if ((((uintptr_t)dst) & 15) == 0) {
      const int n4 = total >> 2;
      for (int i = lane; i < n4; i += 64) ((float4*)dst)[i] = ((const float4*)rows)[i];
      for (int i = (n4 << 2) + lane; i < total; i += 64) dst[i] = rows[i];
    } else {
      for (int i = lane; i < total; i += 64) dst[i] = rows[i];
    }
  }
}

// acc += lane[i+1].s0 * c0 + lane[i+1].s1 * c1 (lane 63: unchanged)
__device__ __forceinline__ void fmac2_shl1(float& acc, float s0, float s1, float c0, float c1) {
  asm volatile(
      "s_nop 1\n\tv_fmac_f32_dpp %0, %1, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %4 wave_shl:1 row_mask:0xf bank_mask:0xf"
      : "+&v"(acc)  // (early clobber: acc starts as a copy of s1 and must not share its register)
      : "v"(s0), "v"(s1), "v"(c0), "v"(c1));
}

template <bool COMPACT>
__global__ void __launch_bounds__(256)
    ctc_grad_kernel(CtcArgs a, const float* __restrict__ coef, const float* __restrict__ gout, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NB = ctc_blocks(a.T);
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, k) pairs
  const bool valid = item < (int64_t)a.B * NB;
  ctc_grad_body<false, false, COMPACT>(a, valid, valid ? (int)(item / NB) : 0, valid ? (int)(item % NB) : 0, coef, gout, dx, smem);
}

// ------------------------------------------------------------------------------------------------
// Pipelined forward + backward: ONE launch.  Workgroups 0 .. 2B-1 are the alpha / beta chains (they
// are dispatched first and there is one per CU at B = 128); the remaining workgroups are gradient
// waves that wait for "their" two checkpoints and then recompute / emit their 16 frames while the
// chains are still running.  Block k needs alpha checkpoint k (ready after k blocks of the alpha
// sweep) and beta checkpoint NB-1-k (ready after NB-k blocks of the beta sweep): the middle of the
// utterance is ready after half the chain time, the ends when the chains finish -- so gradient
// items are numbered from the middle outwards and almost all of the gradient kernel's work
// disappears behind the latency-bound chains, which leave most of every CU idle.
// ------------------------------------------------------------------------------------------------
template <bool LSM, bool COMPACT>
__global__ void __launch_bounds__(256)
    ctc_pipelined_kernel(CtcArgs a, const float* __restrict__ coef, const float* __restrict__ gout,
                         float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nchain = 2 * a.B;
  if ((int)blockIdx.x < nchain) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // (a gradient wave gives up only after ~1 s of polling)
      int32_t* perr = (int32_t*)(a.ws + ctc_ws_layout(a.B, a.T, a.P).perr);
      perr[0] = 0, perr[1] = 0;
    }
    if (threadIdx.x < 64)  // the dependent chain (and its feeders) go first on their SIMDs
      __builtin_amdgcn_s_setprio(3);
    else
      __builtin_amdgcn_s_setprio(2);
    ctc_log_chain_body<true, LSM>(a, (int)blockIdx.x >> 1, (int)blockIdx.x & 1, 0, *reinterpret_cast<ChainLdsT*>(smem));
    return;
  }
  const int NB = ctc_blocks(a.T);
  const int64_t item = (int64_t)(blockIdx.x - nchain) * 4 + (threadIdx.x >> 6);
  const bool valid = item < (int64_t)a.B * NB;
  const int r = valid ? (int)(item / a.B) : 0, b = valid ? (int)(item % a.B) : 0;  // r: rank in readiness order
  const int mid = (NB - 1) / 2;
  const int k = (r & 1) ? mid + (r + 1) / 2 : mid - r / 2;
  ctc_grad_body<true, LSM, COMPACT>(a, valid, b, k, coef, gout, dx, smem);
}

template <bool LSM, bool COMPACT>
__global__ void __launch_bounds__(256)
    ctc_repair_kernel(CtcArgs a, const float* __restrict__ coef, const float* __restrict__ gout, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const CtcWs w = ctc_ws_layout(a.B, a.T, a.P);
  const int nchain = 2 * a.B;
  const bool no_doubt = a.suspect_token != 0 && *(const unsigned long long*)(a.ws + w.suspect) != a.suspect_token;
  if (no_doubt) {
    // The launch before found nothing to doubt (it says so in ONE word: see ctc_mitm.h): nothing to evaluate, the
    // loss it reduced stands.  (The usual case: what the launch then costs is its dispatch and this load.)
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.host_repaired)
      __hip_atomic_store(a.host_repaired, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if ((int)blockIdx.x < nchain) {
    const int b = (int)blockIdx.x >> 1, dir = (int)blockIdx.x & 1;
    if (utterance_rejected_wave(a, w, b, lane)) {  // (uniform over the workgroup: every wave evaluates it)
      if (dir == 0 && threadIdx.x == 0) atomicAdd((int32_t*)(a.ws + w.perr) + 1, 1);
      if (threadIdx.x < 64)
        __builtin_amdgcn_s_setprio(3);
      else
        __builtin_amdgcn_s_setprio(2);
      ctc_log_chain_body<true, LSM, true>(a, b, dir, 0, *reinterpret_cast<ChainLdsT*>(smem));
    }
    // the loss of the fast launch stands unless an utterance was repaired
    if (blockIdx.x == 0 && threadIdx.x < 64 && a.loss_out) reduce_loss_when_done(a, w, lane, true);
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128 && a.host_repaired) {
      // how many utterances this launch recomputes (the certificates were written by the launch before): for the next
      // call's choice of path.  A plain word in pinned host memory, system scope; nobody waits for it.
      int n = 0;
      for (int u = lane; u < a.B; u += 64) n += utterance_rejected(a, w, u) ? 1 : 0;
      n = (int)wave_all_sum((float)n);  // (B <= 2^24: exact)
      if (lane == 0) __hip_atomic_store(a.host_repaired, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  // gradient workgroups: a small persistent grid (the launch is empty on data the fast chains can represent,
  // and what it costs then is its dispatch).  Nothing to repair anywhere: leave after one look at the certificates.
  bool any = false;
  for (int u = threadIdx.x; u < a.B; u += blockDim.x) any = any || utterance_rejected(a, w, u);
  if (!__syncthreads_or(any)) return;
  const int NB = ctc_blocks(a.T);
  const int64_t stride = (int64_t)(gridDim.x - nchain) * 4;
  const int mid = (NB - 1) / 2;
#pragma unroll 1
  for (int64_t item = (int64_t)(blockIdx.x - nchain) * 4 + (threadIdx.x >> 6); item < (int64_t)a.B * NB; item += stride) {
    const int r = (int)(item / a.B), b = (int)(item % a.B);  // r: rank in readiness order
    const int k = (r & 1) ? mid + (r + 1) / 2 : mid - r / 2;
    ctc_grad_body<true, LSM, COMPACT>(a, utterance_rejected_wave(a, w, b, lane), b, k, coef, gout, dx, smem);
  }
}

#include "ctc_mitm.h"

// ------------------------------------------------------------------------------------------------
// Long targets (64 <= L + 1 <= 256): PPL = 2, 3 or 4 target positions per lane, lane i owning the
// contiguous positions PPL*i .. PPL*i + PPL-1.  Within a frame all positions read the PREVIOUS
// frame's values, so a lane's PPL updates are independent (instruction-level parallelism that
// the single-position chain does not have) and still only ONE value crosses lanes per frame (the
// label state of the lane's first position needs the last position of lane i-1: a DPP shift).
// Same checkpoint format (indexed by position), same pipelining, same numerics as above; the
// emission ring holds the label emissions per position and the frame's blank emission once
// (states that do not exist stay at the sentinel by themselves: everything that feeds them is one).
// ------------------------------------------------------------------------------------------------
template <int PPL>
struct LongLds {
  float ring_xl[kRing][kBlk][64][PPL];
  float ring_xb[kRing][kBlk];
  float2 ckbuf[2][64 * PPL];
  double offbuf[2];
  float refsum[kRing];  // sum of the per-frame references of the block in the ring slot (integer valued)
};

template <int PPL>
struct LongLane {  // what a lane knows about its PPL positions (alpha: forward target, beta: reversed)
  int col[PPL];
  bool has_label[PPL], skip[PPL];
};

template <int PPL>
__device__ __forceinline__ LongLane<PPL> long_lane(const CtcArgs& a, int64_t o0, int L, int lane, int dir) {
  LongLane<PPL> c;
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    const int pos = PPL * lane + p;
    int y = -1, yprev = -1;
    if (pos < L) y = a.targets[o0 + (dir == 0 ? pos : L - 1 - pos)];
    if (pos >= 1 && pos - 1 < L) yprev = a.targets[o0 + (dir == 0 ? pos - 1 : L - pos)];
    c.has_label[p] = pos < L;
    c.skip[p] = c.has_label[p] && pos >= 1 && y != yprev;
    c.col[p] = c.has_label[p] ? y : a.blank;
  }
  return c;
}

// value held for position `pos` (lane pos / PPL, slot pos % PPL) broadcast to the wave
template <int PPL>
__device__ __forceinline__ float long_read(const float (&v)[PPL], int pos) {
  const int slot = pos % PPL;
  float s = v[0];
#pragma unroll
  for (int p = 1; p < PPL; ++p) s = slot == p ? v[p] : s;  // wave-uniform select
  return readlane_f(s, pos / PPL);
}

template <int PPL>
__device__ __forceinline__ double long_read_d(const double (&v)[PPL], int pos) {
  const int slot = pos % PPL;
  double s = v[0];
#pragma unroll
  for (int p = 1; p < PPL; ++p) s = slot == p ? v[p] : s;  // wave-uniform select
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), pos / PPL),
                          __builtin_amdgcn_readlane(__double2loint(s), pos / PPL));
}
// the probability-domain arithmetic of the log-domain bodies (see ctc_log_chain_body): shifts of doubles, log2 of a
// state for its float checkpoint, 2^checkpoint as a double, the binary exponent of a positive double
__device__ __forceinline__ double dshr1(double v) {  // lane i receives lane i-1's value, lane 0 receives 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dshl1(double v) {  // lane i receives lane i+1's value, lane 63 receives 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int dexponent(double v) {  // (v > 0)
  return ((__double2hiint(v) >> 20) & 0x7ff) - 1023;
}
__device__ __forceinline__ float dlog2_ck(double v) {
  if (!(v > 0.0)) return kNegBig;
  const float mant = (float)__hiloint2double((__double2hiint(v) & 0x800fffff) | 0x3ff00000, __double2loint(v));
  return (float)dexponent(v) + __builtin_amdgcn_logf(mant);
}
__device__ __forceinline__ double dfrom_log2(float lg) {
  if (!(lg > 0.5f * kNegBig)) return 0.0;
  const float fl = floorf(lg);
  return __builtin_amdgcn_ldexp((double)__builtin_amdgcn_exp2f(lg - fl), (int)fl);
}

template <int PPL, bool SIGNAL, bool LSM = false>
__device__ __forceinline__ void ctc_long_chain_body(const CtcArgs& a, int b, int dir, LongLds<PPL>& S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = a.T, C = a.C, P = a.P;
  const int64_t o0 = a.offsets[b];
  const int L = (int)(a.offsets[b + 1] - o0);
  const LongLane<PPL> c = long_lane<PPL>(a, o0, L, lane, dir);
  const float* xrow = a.x + (int64_t)b * T * C;
  const CtcWs w = ctc_ws_layout(a.B, T, P);
  const int NB = ctc_blocks(T);
  const int h = wave - 1;
  constexpr int kFlusher = SIGNAL ? 3 : 1;
  float raw[kBlk][PPL];
  float lse_raw = 0.f;
  auto issue = [&](int kk) {
    const int k = dir == 0 ? kk : NB - 1 - kk;
    const int t0 = k * kBlk, n = min(kBlk, T - t0);
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      const int t = dir == 0 ? t0 + j : t0 + n - 1 - j;
      const float* row = xrow + (int64_t)min(max(t, 0), T - 1) * C;
#pragma unroll
      for (int p = 0; p < PPL; ++p) raw[j][p] = row[c.col[p]];
    }
    if (LSM) {
      const int t = dir == 0 ? t0 + (lane & 15) : t0 + n - 1 - (lane & 15);
      lse_raw = a.row_lse[(int64_t)b * T + min(max(t, 0), T - 1)];
    }
  };
  // (probability domain on doubles with per-frame references, as ctc_log_chain_body: the ring holds FACTORS 2^(x - r_t))
  auto stage = [&](int kk) {
    const int kq = dir == 0 ? kk : NB - 1 - kk;
    const int nq = min(kBlk, T - kq * kBlk);
    float mxs[kBlk];
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      const float lj = LSM ? readlane_f(lse_raw, j) : 0.f;
      float m = kNegBig;
#pragma unroll
      for (int p = 0; p < PPL; ++p) raw[j][p] = to_score(raw[j][p] - lj), m = vmax(m, raw[j][p]);
      mxs[j] = m;
    }
    const float mm = fold16<true>(mxs, lane);  // every lane: the largest score of frame lane % 16 (positions past L: the blank's)
    const float rr = mm > 0.5f * kNegBig ? rintf(mm) : 0.f;
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      float fs[PPL];
      const float rj = readlane_f(rr, j);
#pragma unroll
      for (int p = 0; p < PPL; ++p) fs[p] = raw[j][p] > 0.5f * kNegBig ? __builtin_amdgcn_exp2f(raw[j][p] - rj) : 0.f;
      const float fblank = long_read<PPL>(fs, L);  // position L has no label: its column is the blank
      if (lane == 0) S.ring_xb[kk % kRing][j] = fblank;
#pragma unroll
      for (int p = 0; p < PPL; ++p) S.ring_xl[kk % kRing][j][lane][p] = c.has_label[p] ? fs[p] : 0.f;
    }
    const float rs = wave_all_sum(lane < nq ? rr : 0.f);
    if (lane == 0) S.refsum[kk % kRing] = rs;
  };
  if (wave == 1 || wave == 2) {
    if (h < NB) {
      issue(h);
      stage(h);
    }
    if (h + 2 < NB) issue(h + 2);
  }
  __syncthreads();

  double ab[PPL], al[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) ab[p] = 0.0, al[p] = 0.0;
  if (lane == 0) ab[0] = 1.0;  // virtual slot "before the first frame"
  double off = 0.0;
  float2* ck = (float2*)(a.ws + w.ck) + ((int64_t)(b * 2 + dir) * NB) * P;
  double* offs = (double*)(a.ws + w.off) + (int64_t)(b * 2 + dir) * NB;
  unsigned long long* ready = (unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + dir) * NB;
  auto flush_checkpoint = [&](int kk) {
    for (int i = lane; i < P; i += 64) {
      const float2 v = S.ckbuf[kk & 1][i];
      if (!SIGNAL) {
        ck[(int64_t)kk * P + i] = v;
      } else {
        unsigned long long bits;
        __builtin_memcpy(&bits, &v, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&ck[(int64_t)kk * P + i]), bits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) {
      const double o = S.offbuf[kk & 1];
      if (!SIGNAL) {
        offs[kk] = o;
      } else {
        unsigned long long bits;
        __builtin_memcpy(&bits, &o, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&offs[kk]), bits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (SIGNAL) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the stores are acknowledged
      if (lane == 0) __hip_atomic_store(&ready[kk], a.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  float e[kBlk][PPL], en[kBlk][PPL], eb[kBlk], ebn[kBlk];
  float rs_cur = 0.f, rs_next = 0.f;  // reference sums of the block in `e` / `en`
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      eb[j] = S.ring_xb[0][j];
#pragma unroll
      for (int p = 0; p < PPL; ++p) e[j][p] = S.ring_xl[0][j][lane][p];
    }
    rs_cur = S.refsum[0];
  }
  for (int kk = 0; kk < NB; ++kk) {
    if (wave == 0) {
      const int k = dir == 0 ? kk : NB - 1 - kk;
      const int n = min(kBlk, T - k * kBlk);
      if (kk + 1 < NB) {
#pragma unroll
        for (int j = 0; j < kBlk; ++j) {
          ebn[j] = S.ring_xb[(kk + 1) % kRing][j];
#pragma unroll
          for (int p = 0; p < PPL; ++p) en[j][p] = S.ring_xl[(kk + 1) % kRing][j][lane][p];
        }
        rs_next = S.refsum[(kk + 1) % kRing];
      }
      if (kk > 0) {  // renormalise: the largest state's binary exponent -> double offset (exact)
        double mx = 0.0;
#pragma unroll
        for (int p = 0; p < PPL; ++p) mx = fmax(mx, fmax(ab[p], al[p]));
        const int em = wave_all_max_int(mx > 0.0 ? dexponent(mx) : -(1 << 20));
        if (em > -(1 << 20)) {
#pragma unroll
          for (int p = 0; p < PPL; ++p) ab[p] = __builtin_amdgcn_ldexp(ab[p], -em), al[p] = __builtin_amdgcn_ldexp(al[p], -em);
          off += (double)em;
        }
      }
#pragma unroll
      for (int p = 0; p < PPL; ++p) S.ckbuf[kk & 1][PPL * lane + p] = make_float2(dlog2_ck(ab[p]), dlog2_ck(al[p]));
      if (lane == 0) S.offbuf[kk & 1] = off;
#pragma unroll
      for (int j = 0; j < kBlk; ++j) {
        if (j < n) {  // (n is uniform; only the last block is short)
          double pal[PPL], nb[PPL], nl[PPL];
          pal[0] = dshr1(al[PPL - 1]);
#pragma unroll
          for (int p = 1; p < PPL; ++p) pal[p] = al[p - 1];
#pragma unroll
          for (int p = 0; p < PPL; ++p) {
            nb[p] = ab[p] + pal[p];
            nl[p] = al[p] + (c.skip[p] ? nb[p] : ab[p]);
          }
#pragma unroll
          for (int p = 0; p < PPL; ++p) ab[p] = nb[p] * (double)eb[j], al[p] = nl[p] * (double)e[j][p];
        }
      }
#pragma unroll
      for (int j = 0; j < kBlk; ++j) {
        eb[j] = ebn[j];
#pragma unroll
        for (int p = 0; p < PPL; ++p) e[j][p] = en[j][p];
      }
      off += (double)rs_cur;  // (the block's references: part of every state's score from here on)
      rs_cur = rs_next;
    } else {
      if (kk > 0 && wave == kFlusher) flush_checkpoint(kk - 1);
      if ((kk & 1) == h && wave <= 2) {
        if (kk + 2 < NB) stage(kk + 2);
        if (kk + 4 < NB) issue(kk + 4);
      }
    }
    __syncthreads();
  }
  if (wave == kFlusher) flush_checkpoint(NB - 1);
  if (dir == 0 && wave == 0) {
    const double a_last = long_read_d<PPL>(ab, L);
    const double l_last = L > 0 ? long_read_d<PPL>(al, L - 1) : 0.0;
    if (lane == 0) {
      const double tot = a_last + l_last;
      const bool alive = tot > 0.0;
      const double z2 = alive ? log2(tot) + off : -1.0e300;
      ((double*)(a.ws + w.z2))[b] = z2;
      a.nll[b] = alive ? (float)(-z2 * 0.6931471805599453) : __builtin_inff();
    }
  }
}

template <int PPL, bool PIPE, bool LSM = false>
__device__ __forceinline__ void ctc_long_grad_body(const CtcArgs& a, bool valid, int b, int k,
                                                   const float* __restrict__ coef, const float* __restrict__ gout,
                                                   float* __restrict__ dx, char* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = a.T, C = a.C, P = a.P;
  const int NB = ctc_blocks(T);
  const CtcWs w = ctc_ws_layout(a.B, T, P);
  float* rows = (float*)smem + (size_t)wave * (kBlk + 1) * C;
  int* cnt = (int*)(rows + (size_t)kBlk * C);
  const int t0 = k * kBlk, n = min(kBlk, T - t0);
  bool live = valid && (PIPE || a.nll[b] < __builtin_inff());
  constexpr bool lsm = PIPE && LSM;
  const float cf_row = (coef ? coef[valid ? b : 0] : 1.f) * (gout ? gout[0] : 1.f);
  float lse_blk = 0.f;
  if (valid) {
    for (int i = lane; i < (kBlk + 1) * C; i += 64) rows[i] = 0.f;
    if (lsm) {
      lse_blk = a.row_lse[(int64_t)b * T + min(t0 + (lane & 15), T - 1)];
      lsm_seed_rows(rows, a.x + ((int64_t)b * T + t0) * C, n, C, lse_blk, cf_row, lane);
    }
  }
  if (PIPE && valid) {
    const unsigned long long* ra = (const unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + 0) * NB + k;
    const unsigned long long* rb = (const unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + 1) * NB + (NB - 1 - k);
    int ok = 0;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      ok = __hip_atomic_load(ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.token &&
           __hip_atomic_load(rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.token;
      if (ok) break;
      __builtin_amdgcn_s_sleep(64);
    }
    if (!ok) {
      if (lane == 0) atomicOr((int32_t*)(a.ws + w.perr), 1);
      __builtin_trap();
    }
  }
  if (!live) {
    if (valid) {  // zero rows for a dead utterance
      float* dst = dx + ((int64_t)b * T + t0) * C;
      for (int i = lane; i < n * C; i += 64) dst[i] = 0.f;
    }
    return;
  }
  const int64_t o0 = a.offsets[b];
  const int L = (int)(a.offsets[b + 1] - o0);
  int y[PPL];
  bool has_label[PPL], skip[PPL], skipn[PPL], uniq[PPL], dup[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    const int pos = PPL * lane + p;
    has_label[p] = pos < L;
    y[p] = has_label[p] ? a.targets[o0 + pos] : a.blank;
    const int yprev = (pos >= 1 && pos - 1 < L) ? a.targets[o0 + pos - 1] : -1;
    const int ynext = pos + 1 < L ? a.targets[o0 + pos + 1] : -1;
    skip[p] = has_label[p] && pos >= 1 && y[p] != yprev;
    skipn[p] = pos + 1 < L && ynext != y[p];
    if (has_label[p]) atomicAdd(&cnt[y[p]], 1);
  }
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    dup[p] = has_label[p] && (cnt[y[p]] > 1 || y[p] == a.blank);
    uniq[p] = has_label[p] && !dup[p];
  }
  const float* xrow = a.x + (int64_t)b * T * C;
  float xl[kBlk][PPL], xb[kBlk];
#pragma unroll
  for (int j = 0; j < kBlk; ++j) {
    const float* row = xrow + (int64_t)min(t0 + j, T - 1) * C;
#pragma unroll
    for (int p = 0; p < PPL; ++p) xl[j][p] = row[y[p]];
  }
  // The block in the probability domain on doubles with per-frame references (see ctc_grad_body): xl / xb hold FACTORS.
  float ref_sum;
  {
    float mxs[kBlk];
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      const float lj = lsm ? readlane_f(lse_blk, j) : 0.f;
      float m = kNegBig;
#pragma unroll
      for (int p = 0; p < PPL; ++p) xl[j][p] = to_score(xl[j][p] - lj), m = vmax(m, xl[j][p]);
      mxs[j] = m;
    }
    const float mm = fold16<true>(mxs, lane);  // every lane: the largest score of frame lane % 16
    const float rr = mm > 0.5f * kNegBig ? rintf(mm) : 0.f;
#pragma unroll
    for (int j = 0; j < kBlk; ++j) {
      float fs[PPL];
      const float rj = readlane_f(rr, j);
#pragma unroll
      for (int p = 0; p < PPL; ++p) fs[p] = xl[j][p] > 0.5f * kNegBig ? __builtin_amdgcn_exp2f(xl[j][p] - rj) : 0.f;
      xb[j] = long_read<PPL>(fs, L);
#pragma unroll
      for (int p = 0; p < PPL; ++p) xl[j][p] = has_label[p] ? fs[p] : 0.f;
    }
    ref_sum = wave_all_sum(lane < n ? rr : 0.f);
  }
  const float2* cka = (const float2*)(a.ws + w.ck) + ((int64_t)(b * 2 + 0) * NB) * P;
  const float2* ckb = (const float2*)(a.ws + w.ck) + ((int64_t)(b * 2 + 1) * NB) * P;
  const double* offa = (const double*)(a.ws + w.off) + (int64_t)(b * 2 + 0) * NB;
  const double* offb = (const double*)(a.ws + w.off) + (int64_t)(b * 2 + 1) * NB;
  auto load_ck = [&](const float2* p) {
    if (!PIPE) return *p;
    const unsigned long long bits =
        __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float2 v;
    __builtin_memcpy(&v, &bits, 8);
    return v;
  };
  double ab[PPL], al[PPL], bb[PPL], bl[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    const int pos = PPL * lane + p;
    const float2 ca = pos < P ? load_ck(&cka[(int64_t)k * P + pos]) : make_float2(kNegBig, kNegBig);
    ab[p] = dfrom_log2(ca.x), al[p] = dfrom_log2(ca.y);
    bb[p] = dfrom_log2(pos <= L ? load_ck(&ckb[(int64_t)(NB - 1 - k) * P + (L - pos)]).x : kNegBig);
    bl[p] = dfrom_log2(pos < L ? load_ck(&ckb[(int64_t)(NB - 1 - k) * P + (L - 1 - pos)]).y : kNegBig);
  }
  if (PIPE && lane == 0) {
    unsigned long long* rdy = (unsigned long long*)(a.ws + w.ready);
    __hip_atomic_store(rdy + (int64_t)(b * 2 + 0) * NB + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rdy + (int64_t)(b * 2 + 1) * NB + (NB - 1 - k), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const float cf = (coef ? coef[b] : 1.f) * (gout ? gout[0] : 1.f);
  double pa_b[kBlk][PPL], pa_l[kBlk][PPL];
#pragma unroll
  for (int j = 0; j < kBlk; ++j) {  // alpha forward through the block, kept in registers
    double pal[PPL], nb[PPL], nl[PPL];
    pal[0] = dshr1(al[PPL - 1]);
#pragma unroll
    for (int p = 1; p < PPL; ++p) pal[p] = al[p - 1];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      nb[p] = ab[p] + pal[p];
      nl[p] = al[p] + (skip[p] ? nb[p] : ab[p]);
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      ab[p] = nb[p] * (double)xb[j], al[p] = nl[p] * (double)xl[j][p];
      pa_b[j][p] = ab[p], pa_l[j][p] = al[p];
    }
  }
  // transition-propagated beta of one frame: tb (blank states), tl (label states)
  auto propagate = [&](double (&tb)[PPL], double (&tl)[PPL]) {
#pragma unroll
    for (int p = 0; p < PPL; ++p) tb[p] = bb[p] + bl[p];
    const double tb_next_lane = dshl1(tb[0]), bb_next_lane = dshl1(bb[0]);  // (both shifts outside any divergent select)
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const double tbn = p + 1 < PPL ? tb[p + 1 < PPL ? p + 1 : 0] : tb_next_lane;
      const double bbn = p + 1 < PPL ? bb[p + 1 < PPL ? p + 1 : 0] : bb_next_lane;
      tl[p] = bl[p] + (skipn[p] ? tbn : bbn);
    }
  };
  double scale;
  if (PIPE) {  // the block's own Z at its last frame
    double tb[PPL], tl[PPL];
    propagate(tb, tl);
    double zs = 0.0;
#pragma unroll
    for (int j = 0; j < kBlk; ++j)
      if (j == n - 1) {
#pragma unroll
        for (int p = 0; p < PPL; ++p) zs += pa_b[j][p] * tb[p] + pa_l[j][p] * tl[p];
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zs += __shfl_xor(zs, o, 64);
    scale = zs > 0.0 && zs < 1.0e300 ? 1.0 / zs : 0.0;
  } else {
    const double z2 = ((const double*)(a.ws + w.z2))[b];
    const double e = offa[k] + offb[NB - 1 - k] - z2 + (double)ref_sum;
    scale = (z2 > -1.0e299 && e < 1000.0) ? exp2(e) : 0.0;
  }
#pragma unroll
  for (int j = kBlk - 1; j >= 0; --j) {
    if (j < n) {
      double tb[PPL], tl[PPL];
      propagate(tb, tl);
      float gbs = 0.f;
#pragma unroll
      for (int p = 0; p < PPL; ++p) {
        gbs += (float)(pa_b[j][p] * tb[p] * scale);
        const float gl = (float)(pa_l[j][p] * tl[p] * scale);
        if (uniq[p]) rows[j * C + y[p]] = (lsm ? rows[j * C + y[p]] : 0.f) + gl * cf;
        if (dup[p] && gl != 0.f) atomicAdd(&rows[j * C + y[p]], gl * cf);
      }
      const float gsum = wave_reduce_sum_lane63(gbs);
      if (lane == 63 && gsum != 0.f) atomicAdd(&rows[j * C + a.blank], gsum * cf);
#pragma unroll
      for (int p = 0; p < PPL; ++p) bb[p] = tb[p] * (double)xb[j], bl[p] = tl[p] * (double)xl[j][p];
    }
  }
  if (lsm && !(scale > 0.0))  // no accepting path: zero gradient, softmax term included
    for (int i = lane; i < kBlk * C; i += 64) rows[i] = 0.f;
  {
    float* dst = dx + ((int64_t)b * T + t0) * C;
    const int total = n * C;
    if ((((uintptr_t)dst) & 15) == 0) {
      const int n4 = total >> 2;
      for (int i = lane; i < n4; i += 64) ((float4*)dst)[i] = ((const float4*)rows)[i];
      for (int i = (n4 << 2) + lane; i < total; i += 64) dst[i] = rows[i];
    } else {
      for (int i = lane; i < total; i += 64) dst[i] = rows[i];
    }
  }
}

template <int PPL, bool LSM>
__global__ void __launch_bounds__(256)
    ctc_long_pipelined_kernel(CtcArgs a, const float* __restrict__ coef, const float* __restrict__ gout,
                              float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nchain = 2 * a.B;
  if ((int)blockIdx.x < nchain) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      int32_t* perr = (int32_t*)(a.ws + ctc_ws_layout(a.B, a.T, a.P).perr);
      perr[0] = 0, perr[1] = 0;
    }
    if (threadIdx.x < 64)
      __builtin_amdgcn_s_setprio(3);
    else
      __builtin_amdgcn_s_setprio(2);
    ctc_long_chain_body<PPL, true, LSM>(a, (int)blockIdx.x >> 1, (int)blockIdx.x & 1, *reinterpret_cast<LongLds<PPL>*>(smem));
    return;
  }
  const int NB = ctc_blocks(a.T);
  const int64_t item = (int64_t)(blockIdx.x - nchain) * 4 + (threadIdx.x >> 6);
  const bool valid = item < (int64_t)a.B * NB;
  const int r = valid ? (int)(item / a.B) : 0, b = valid ? (int)(item % a.B) : 0;
  const int mid = (NB - 1) / 2;
  const int k = (r & 1) ? mid + (r + 1) / 2 : mid - r / 2;
  ctc_long_grad_body<PPL, true, LSM>(a, valid, b, k, coef, gout, dx, smem);
}

template <int PPL>
__global__ void __launch_bounds__(192) ctc_long_chain_kernel(CtcArgs a) {
  __shared__ LongLds<PPL> S;
  ctc_long_chain_body<PPL, false>(a, blockIdx.x, blockIdx.y, S);
}

template <int PPL>
__global__ void __launch_bounds__(256)
    ctc_long_grad_kernel(CtcArgs a, const float* __restrict__ coef, const float* __restrict__ gout, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NB = ctc_blocks(a.T);
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool valid = item < (int64_t)a.B * NB;
  ctc_long_grad_body<PPL, false>(a, valid, valid ? (int)(item / NB) : 0, valid ? (int)(item % NB) : 0, coef, gout, dx,
                                 smem);
}

}  // namespace wfl

using namespace wfl;

extern "C" {

static int ctc_check(int B, int T, int C, int max_len, int blank, const char* who) {
  if (B <= 0 || T <= 0 || C <= 0 || blank < 0 || blank >= C || max_len < 0) {
    set_error("%s: bad arguments (B=%d T=%d C=%d blank=%d max_len=%d)", who, B, T, C, blank, max_len);
    return WFL_ERR_INVALID;
  }
  if (max_len + 1 > 256) {
    set_error("%s: target length %d exceeds four positions per lane (use the lattice engine)", who, max_len);
    return WFL_ERR_UNSUPPORTED;
  }
  // gradient tiles in LDS: dense [17][C] per wave for long targets, compact [16][64] + C bytes otherwise
  const bool wide = max_len + 1 > 64 ? (size_t)4 * (kBlk + 1) * C * 4 > (size_t)kLdsBytes
                                     : (size_t)8 * compact_wave_bytes(C) > (size_t)kLdsBytes;
  if (wide) {
    set_error("%s: C=%d too large for the LDS tiles of the gradient kernel (use the lattice engine)", who, C);
    return WFL_ERR_UNSUPPORTED;
  }
  return WFL_OK;
}

// The exchange of gathered emissions between an utterance's two sweeps pays when x does not stay in the Infinity Cache
// between them (256 MiB) and its rows are wide enough for a 45-label gather to waste most of what it touches.
static bool ctc_use_xc(int B, int T, int C, int max_len) {
  if (max_len + 1 > 64) return false;  // (the lane-exponent step: one target position per lane)
  return (int64_t)B * T * C * 4 >= (192ll << 20) && C >= 192;
}

int wfl_ctc_workspace(int B, int T, int C, int max_len, int64_t* ws_elems) {
  if (!ws_elems || B <= 0 || T <= 0 || max_len < 0) {
    set_error("ctc_workspace: bad arguments");
    return WFL_ERR_INVALID;
  }
  *ws_elems = ctc_ws_layout(B, T, max_len + 1).total + 4 + (ctc_use_xc(B, T, C, max_len) ? (int64_t)B * T * kXcStride : 0);
  return WFL_OK;
}

int wfl_ctc_workspace_field(int B, int T, int max_len, int field, int64_t* offset_elems, int64_t* length_elems) {
  if (!offset_elems || !length_elems || B <= 0 || T <= 0 || max_len < 0) {
    set_error("ctc_workspace_field: bad arguments");
    return WFL_ERR_INVALID;
  }
  const CtcWs w = ctc_ws_layout(B, T, max_len + 1);
  switch (field) {
    case WFL_CTC_WS_REJECTED: *offset_elems = w.flag, *length_elems = B; break;
    case WFL_CTC_WS_STATUS: *offset_elems = w.perr, *length_elems = 2; break;
    case WFL_CTC_WS_LOG2Z: *offset_elems = w.z2, *length_elems = 2 * (int64_t)B; break;
    case WFL_CTC_WS_ZRANGE: *offset_elems = w.zloc, *length_elems = 4 * (int64_t)B; break;
    case WFL_CTC_WS_CLOCK: *offset_elems = w.clk, *length_elems = 8 * (int64_t)B; break;
    case WFL_CTC_WS_DEBUG:
      *offset_elems = WFL_MITM_STATS ? w.dbg : 0, *length_elems = WFL_MITM_STATS ? 2 * (8 * 16 + 256) * 2 * (int64_t)B : 0;
      break;
    default: set_error("ctc_workspace_field: unknown field %d", field); return WFL_ERR_INVALID;
  }
  return WFL_OK;
}

int wfl_ctc_forward(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets, int max_len,
                    int blank, int flags, float* ws, float* nll, void* stream) {
  if (int rc = ctc_check(B, T, C, max_len, blank, "ctc_forward")) return rc;
  if (!x || !targets || !offsets || !ws || !nll) {
    set_error("ctc_forward: null buffer");
    return WFL_ERR_INVALID;
  }
  if (flags != 0) {
    set_error("ctc_forward: flags must be 0 (the three-launch lane-exponent step of rounds 1-3 is retired: "
              "wfl_ctc_forward_backward is the training step)");
    return WFL_ERR_UNSUPPORTED;
  }
  CtcArgs a{x, B, T, C, max_len + 1, blank, targets, offsets, ws, nll};
  const int ppl = (max_len + 1 + 63) / 64;  // target positions per lane
  if (ppl > 1) {  // long targets: multi-position lanes
    const dim3 grid((unsigned)B, 2u);
    if (ppl == 2)
      hipLaunchKernelGGL(ctc_long_chain_kernel<2>, grid, dim3(192), 0, (hipStream_t)stream, a);
    else if (ppl == 3)
      hipLaunchKernelGGL(ctc_long_chain_kernel<3>, grid, dim3(192), 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL(ctc_long_chain_kernel<4>, grid, dim3(192), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(ctc_log_chain_kernel, dim3((unsigned)B, 2u), dim3(192), 0, (hipStream_t)stream, a, 0);
  }
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

static int ctc_forward_backward_impl(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets,
                                     int max_len, int blank, float* ws, float* nll, const float* coef, const float* gout,
                                     float* dx, const float* loss_scale, float* loss_out, const float* row_lse,
                                     const wfl_ctc_call* call, void* stream);

int wfl_ctc_forward_backward(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets,
                             int max_len, int blank, float* ws, float* nll, const float* coef, const float* gout,
                             float* dx, const float* loss_scale, float* loss_out, const float* row_lse, void* stream) {
  return ctc_forward_backward_impl(x, B, T, C, targets, offsets, max_len, blank, ws, nll, coef, gout, dx, loss_scale, loss_out,
                                   row_lse, nullptr, stream);
}

int wfl_ctc_forward_backward_call(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets,
                                  int max_len, int blank, float* ws, float* nll, const float* coef, const float* gout,
                                  float* dx, const float* loss_scale, float* loss_out, const float* row_lse,
                                  const wfl_ctc_call* call, void* stream) {
  return ctc_forward_backward_impl(x, B, T, C, targets, offsets, max_len, blank, ws, nll, coef, gout, dx, loss_scale, loss_out,
                                   row_lse, call, stream);
}

static int ctc_forward_backward_impl(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets,
                                     int max_len, int blank, float* ws, float* nll, const float* coef, const float* gout,
                                     float* dx, const float* loss_scale, float* loss_out, const float* row_lse,
                                     const wfl_ctc_call* call, void* stream) {
  if (int rc = ctc_check(B, T, C, max_len, blank, "ctc_forward_backward")) return rc;
  if (!x || !targets || !offsets || !ws || !nll || !dx) {
    set_error("ctc_forward_backward: null buffer");
    return WFL_ERR_INVALID;
  }
  CtcArgs a{x, B, T, C, max_len + 1, blank, targets, offsets, ws, nll, 0ull, loss_scale, loss_out, row_lse};
  a.n_labels = call && call->n_labels > 0 ? call->n_labels : 0;
  // how long a wave of the meet-in-the-middle launch waits for a hand-off before it gives up and the repair launch
  // recomputes the batch (WFL_CTC_MITM_SPIN: polls; tests force the time-out with 1)
  static const int mitm_spin = [] {
    const char* e = getenv("WFL_CTC_MITM_SPIN");
    return e && atoi(e) > 0 ? atoi(e) : kMSpinDefault;
  }();
  a.spin = mitm_spin;
  // launch token: process-wide counter mixed with the workspace address -- uninitialised memory or flags
  // left by a launch that used the block earlier cannot equal it; consumers clear the flags they used,
  // so replaying the SAME launch from a hipGraph (same token, same workspace) starts from cleared flags
  static std::atomic<unsigned long long> counter{0x9e3779b97f4a7c15ull};
  a.token = counter.fetch_add(0x9e3779b97f4a7c15ull) ^ (unsigned long long)(uintptr_t)ws;
  if (a.token == 0) a.token = 1;
  const int64_t items = (int64_t)B * ctc_blocks(T);
  const int ppl = (max_len + 1 + 63) / 64;  // target positions per lane
  const dim3 grid((unsigned)(2 * B + (items + 3) / 4));
  // Which step?  The lane-exponent step + repair launch costs fast + log-domain time when the certificate rejects
  // (scores without structure and a spread of 2 nats or more: DESIGN.md section 4); the log-domain pipelined step alone
  // is cheaper than that.  The repair launch leaves its count in a pinned host word per workspace; when the LAST
  // lane-exponent step on this workspace recomputed more than an eighth of its utterances, this call goes straight to
  // the log-domain step, and every 16th such call tries the lane-exponent step again (the data may have changed).
  // Results are within the parity bar on either path (bit-identical only on the same path).  WFL_CTC_ADAPTIVE=0: off.
  static const bool adaptive_on = [] {
    const char* e = getenv("WFL_CTC_ADAPTIVE");
    return !(e && atoi(e) == 0);
  }();
  bool prefer_log = false;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing((hipStream_t)stream, &capturing);  // (a captured step must not allocate, and replays one choice)
  // (the memory of the choice is the CALLER's: two int32 of pinned host memory per criterion / workspace -- [0] written
  // by the repair launch, system scope, read here without a synchronisation; [1] this function's counter.  Without
  // it every call starts with the lane-exponent step.)
  int32_t* state = call ? call->host_state : nullptr;
  if (adaptive_on && ppl == 1 && capturing == hipStreamCaptureStatusNone && state) {
    a.host_repaired = state;
    if ((int64_t) * (volatile int32_t*)state * 8 > B) {
      prefer_log = (++state[1] & 15) != 0;
    } else {
      state[1] = 0;
    }
  }
  // log-domain kernels (4-wave workgroups): dense row tiles while five workgroups share a CU with them, compact beyond
  const bool lcompact = ppl == 1 && C > 120;
  const size_t rows_lds = lcompact ? (size_t)4 * compact_wave_bytes(C) : (size_t)4 * (kBlk + 1) * C * 4;
  auto launch = [&](auto kern, size_t chain_lds) -> int {
    const size_t lds = std::max(rows_lds, chain_lds);
    if (lds > (size_t)kLdsBytes) {
      set_error("ctc_forward_backward: needs %zu B of LDS (limit %d)", lds, kLdsBytes);
      return WFL_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)kern, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, a, coef, gout, dx);
    return WFL_OK;
  };
  int rc = WFL_OK;
  // targets of at most 63 labels: the meet-in-the-middle launch (lane-exponent sweeps that emit the gradient themselves,
  // ctc_mitm.h) + certificate + repair launch; WFL_CTC_PIPELINE=log selects the log-domain pipelined launch
  static const bool env_log = [] {
    const char* e = getenv("WFL_CTC_PIPELINE");
    return e && std::string(e) == "log";
  }();
  const bool force_log = env_log || prefer_log;
  // two workgroup shapes (ctc_mitm.h): 16 waves, one workgroup per CU, while the batch has no more sweeps than the chip
  // has CUs; 8 waves, two per CU, beyond
  const int cus = [] {  // (of the CURRENT device: a process may drive several)
    static std::mutex mu;
    static std::map<int, int> seen;
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find(dev);
    if (it != seen.end()) return it->second;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    seen[dev] = n;
    return n;
  }();
  const bool small_wg = 2 * (int64_t)B > cus;
  auto mitm_lds_of = [&](auto k, bool wide) {
    using K = decltype(k);
    const size_t tiles = wide ? (size_t)K::kEmitters * kBlk * kCS * 4 + (((size_t)C + 15) & ~(size_t)15)  // compact tiles + column map
                              : (size_t)K::kEmitters * kBlk * kMTile * 4;
    return ((sizeof(MitmLds<K>) + 15) & ~(size_t)15) + tiles;
  };
  static_assert(((sizeof(MitmLds<MitmK<16>>) + 15) & ~(size_t)15) + (size_t)MitmK<16>::kEmitters * kBlk * kMTile * 4 <= (size_t)kLdsBytes,
                "ctc_mitm.h: LDS of the 16-wave shape");
  static_assert(2 * (((sizeof(MitmLds<MitmK<8>>) + 15) & ~(size_t)15) + (size_t)MitmK<8>::kEmitters * kBlk * kMTile * 4) <= (size_t)kLdsBytes,
                "ctc_mitm.h: two 8-wave workgroups per CU");
  // rows wider than the dense tile: the emitters accumulate in the compact tile and expand through a column map
  // (WIDE); its LDS: one byte per class behind the tiles
  const bool wide = C > kMTile;
  // (the LDS of the wide shape is bounded by the class count the entry point admits: both shapes always fit)
  const bool wide_fits = small_wg ? 2 * mitm_lds_of(MitmK<8>{}, true) <= (size_t)kLdsBytes : mitm_lds_of(MitmK<16>{}, true) <= (size_t)kLdsBytes;
  if (ppl == 1 && !force_log && (!wide || wide_fits)) {
    auto launch_mitm = [&](auto kern, auto k) -> int {
      using K = decltype(k);
      const size_t lds = mitm_lds_of(k, wide);
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)kern, (int)lds));
      hipLaunchKernelGGL(kern, dim3((unsigned)(2 * B)), dim3(K::kWaves * 64), lds, (hipStream_t)stream, a, coef, gout, dx);
      return WFL_OK;
    };
    // wide rows beyond the cache: the sweeps exchange what they gathered as compact rows (ctc_mitm.h, stagers) in the
    // region the round-2 compact pre-pass used
    a.xc = (wide && ctc_use_xc(B, T, C, max_len)) ? ws + ((ctc_ws_layout(B, T, max_len + 1).total + 3) & ~(int64_t)3)
                                                              : nullptr;
    if (row_lse) {  // the fused log_softmax criterion (raw scores in, gradient w.r.t. raw scores out)
      if (small_wg)
        rc = wide ? launch_mitm(ctc_mitm_kernel<MitmK<8>, true, true>, MitmK<8>{}) : launch_mitm(ctc_mitm_kernel<MitmK<8>, true, false>, MitmK<8>{});
      else
        rc = wide ? launch_mitm(ctc_mitm_kernel<MitmK<16>, true, true>, MitmK<16>{}) : launch_mitm(ctc_mitm_kernel<MitmK<16>, true, false>, MitmK<16>{});
    } else if (small_wg)
      rc = wide ? launch_mitm(ctc_mitm_kernel<MitmK<8>, false, true>, MitmK<8>{}) : launch_mitm(ctc_mitm_kernel<MitmK<8>, false, false>, MitmK<8>{});
    else
      rc = wide ? launch_mitm(ctc_mitm_kernel<MitmK<16>, false, true>, MitmK<16>{}) : launch_mitm(ctc_mitm_kernel<MitmK<16>, false, false>, MitmK<16>{});
    if (rc) return rc;
    WFL_LAUNCH_CHECK();
    a.xc = nullptr;  // (only the first halves' frames are in it)
    a.suspect_token = a.token;
    a.token = counter.fetch_add(0x9e3779b97f4a7c15ull) ^ (unsigned long long)(uintptr_t)ws;
    if (a.token == 0) a.token = 1;
    auto launch_repair = [&](auto kern) -> int {
      const size_t lds = std::max(rows_lds, sizeof(ChainLdsT));
      if (lds > 48 * 1024)
        WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)kern, (int)lds));
      const dim3 rgrid((unsigned)(2 * B + std::min<int64_t>((items + 3) / 4, 512)));
      hipLaunchKernelGGL(kern, rgrid, dim3(256), lds, (hipStream_t)stream, a, coef, gout, dx);
      return WFL_OK;
    };
    rc = lcompact ? (row_lse ? launch_repair(ctc_repair_kernel<true, true>) : launch_repair(ctc_repair_kernel<false, true>))
                  : (row_lse ? launch_repair(ctc_repair_kernel<true, false>) : launch_repair(ctc_repair_kernel<false, false>));
  } else if (ppl == 1) {
    rc = lcompact ? (row_lse ? launch(ctc_pipelined_kernel<true, true>, sizeof(ChainLdsT))
                             : launch(ctc_pipelined_kernel<false, true>, sizeof(ChainLdsT)))
                  : (row_lse ? launch(ctc_pipelined_kernel<true, false>, sizeof(ChainLdsT))
                             : launch(ctc_pipelined_kernel<false, false>, sizeof(ChainLdsT)));
  } else {
    a.loss_out = nullptr;  // the long-target chains do not reduce the loss in-kernel
    if (row_lse)
      rc = ppl == 2   ? launch(ctc_long_pipelined_kernel<2, true>, sizeof(LongLds<2>))
           : ppl == 3 ? launch(ctc_long_pipelined_kernel<3, true>, sizeof(LongLds<3>))
                      : launch(ctc_long_pipelined_kernel<4, true>, sizeof(LongLds<4>));
    else
      rc = ppl == 2   ? launch(ctc_long_pipelined_kernel<2, false>, sizeof(LongLds<2>))
           : ppl == 3 ? launch(ctc_long_pipelined_kernel<3, false>, sizeof(LongLds<3>))
                      : launch(ctc_long_pipelined_kernel<4, false>, sizeof(LongLds<4>));
  }
  if (rc) return rc;
  WFL_LAUNCH_CHECK();
  if (ppl > 1 && loss_out) return wfl_reduce_loss(nll, nullptr, loss_scale, B, 1.f, 0, loss_out, stream);
  return WFL_OK;
}

int wfl_ctc_grad(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets, int max_len,
                 int blank, const float* ws, const float* nll, const float* coef, const float* gout, float* dx,
                 void* stream) {
  if (int rc = ctc_check(B, T, C, max_len, blank, "ctc_grad")) return rc;
  if (!x || !targets || !offsets || !ws || !nll || !dx) {
    set_error("ctc_grad: null buffer");
    return WFL_ERR_INVALID;
  }
  CtcArgs a{x, B, T, C, max_len + 1, blank, targets, offsets, (float*)ws, (float*)nll};
  const int64_t items = (int64_t)B * ctc_blocks(T);
  const int ppl = (max_len + 1 + 63) / 64;
  const bool lcompact = ppl == 1 && C > 120;  // (see wfl_ctc_forward_backward)
  const size_t lds = lcompact ? (size_t)4 * compact_wave_bytes(C) : (size_t)4 * (kBlk + 1) * C * 4;
  auto launch = [&](auto kern) -> int {
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)kern, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((items + 3) / 4)), dim3(256), lds, (hipStream_t)stream, a, coef, gout, dx);
    return WFL_OK;
  };
  if (int rc = ppl == 1   ? (lcompact ? launch(ctc_grad_kernel<true>) : launch(ctc_grad_kernel<false>))
               : ppl == 2 ? launch(ctc_long_grad_kernel<2>)
               : ppl == 3 ? launch(ctc_long_grad_kernel<3>)
                          : launch(ctc_long_grad_kernel<4>))
    return rc;
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

}  // extern "C"

// Meet-in-the-middle CTC training step (included by ctc_kernels.hip, inside namespace wfl).
//
// criterions/ctc.py:38-94 (forward_score of the composed lattice + its backward) as ONE launch in which the
// gradient is emitted BY the sweeps instead of being recomputed by a third pass:
//
//   * one workgroup per (utterance, direction) -- 2B workgroups, one per CU at B = 128.  The alpha sweep runs
//     0 -> T, the beta sweep T -> 0 (the same recursion on the reversed target and reversed time), both in the
//     lane-exponent arithmetic of ctc_kernels.hip (float mantissa + one integer exponent per lane, five
//     instructions per frame, renormalised every 16 frames);
//   * FIRST half of a sweep (blocks n < H0): its state before every 16-frame block -- raw mantissas and lane
//     exponents, ~0.5 KB -- is published for the partner workgroup (device-coherent stores + a flag);
//   * SECOND half (n >= H0): the frames the sweep now crosses were crossed by the partner in ITS first half.
//     "Emitter" waves of the same workgroup take the block's emission factors from the LDS ring the chain wave was
//     fed from (no second gather, no second exp2, no second reference fold), the sweep's own state before the
//     block from the LDS mailbox, the partner's state after the block from the published checkpoints, redo the 16
//     frames of both recursions in registers, normalise by the locally reproduced Z (sum_s alpha beta at the
//     block's last frame: the certificate's identity -- log Z itself is only known when the alpha sweep ends) and
//     write the block's dense gradient rows.  Every emission factor is computed once per sweep and x is read
//     twice (once per direction) instead of 3.5 times; nothing remains to be done when the sweeps end but the
//     last block of each.
//
// Waves of a workgroup (roles are fixed per wave index so that the chain wave's SIMD carries only light waves --
// waves are placed round-robin on the four SIMDs):
//     0 chain | 4 flusher | 8 fetcher | 1,2,3,5 stagers | 6,7,9,10,11,13,14,15,12 emitters  (SIMD of wave w: w % 4 --
//     the chain wave shares its SIMD with the two light waves and one emitter)
//   chain    the dependent recursion and nothing else; factors travel ring -> registers a whole block ahead
//   stagers  gather x (two blocks in flight each), references, exp2 -> LDS ring
//   flusher  sums the references (double offsets) and, in the first half, publishes the raw checkpoints
//   fetcher  second half: waits for the partner's flags, loads its checkpoints, mirrors / rescales them into the
//            sweep's own lane order and leaves them in LDS -- emitters never wait on another CU and never issue a
//            global load (their global stores are fire-and-forget: vmcnt of a storing wave is never waited on)
//   emitters one block each, round robin
// No barrier after the initial one: all hand-offs are LDS mailboxes (DS instructions of a wave execute in order).
//
// The certificate (per-block log2 Z against the chain's, posterior mass per frame) and the repair launch are those
// of the fast pipelined step (ctc_repair_kernel re-runs rejected utterances in the log domain).
#pragma once

#ifndef WFL_MITM_ABL
#define WFL_MITM_ABL 0  // scratch: switch off parts of the chain wave's steady block (1 renorm, 2 ring reads, 4 checkpoint, 8 frames)
#endif
#ifndef WFL_MITM_STORE
#define WFL_MITM_STORE 2  // gradient row stores: 0 plain, 1 non-temporal, 2 sc0 sc1, 3 sc1, 4 sc0 sc1 nt (scratch A/B)
#endif
#ifndef WFL_MITM_CLAMP_EVERY
#define WFL_MITM_CLAMP_EVERY 1  // the neighbour-gap clamp in every n-th block of the chain (1 or 4; 4: -0.5 us at cfg2, but fronts that cross four lanes inside one block overflow: 10 of 128 peaked utterances repaired)
#endif
#ifndef WFL_MITM_LEAN
#define WFL_MITM_LEAN 1  // the chain wave's complete blocks: 1 the factor reads between the frames, 0 plain blocks only (A/B)
#endif
#ifndef WFL_MITM_LSM_ROWS
#define WFL_MITM_LSM_ROWS 8  // fused log_softmax: rows of a block written out per pass (4: quarters -- 1 % slower at the module's cfg2 shape)
#endif
constexpr int kLsmRows = WFL_MITM_LSM_ROWS;
#ifndef WFL_MITM_STATS
#define WFL_MITM_STATS 0  // 1: per-wave wait / busy cycle counts in the workspace (scratch/mitm_stats.py)
#endif

constexpr int kMSpinDefault = 1 << 24;  // polls a wave waits for a hand-off before it gives up (CtcArgs::spin: WFL_CTC_MITM_SPIN)
constexpr int kMaxCouple = 100;  // bound on the exponent of a coupling factor (chain wave and emitters alike)
constexpr int kMTile = 128;   // floats per row of an emitter's gradient tile (a compile-time stride: the 16 rows of a label's
                              // column are one LDS address + immediate offsets); the step takes C <= kMTile

// Workgroup shapes.  role of a wave: 0 chain, 1 stager, 2 flusher, 3 fetcher, 4 emitter; index within the role.  Waves
// are placed round-robin on the four SIMDs (wave w: SIMD w % 4).
//   MitmK<16>  one workgroup per CU (2 B <= CUs: the BASELINE shape).  4 stagers + 9 emitters: an emitter is latency-bound
//              (one block is ~7000 cycles of dependent DPP / LDS work on a shared SIMD), so the second half goes at the pace
//              of emitters x blocks per emitter -- cfg2 42.0 us with 6 + 7, 40.1 with 4 + 9, 41.4 with 3 + 10 and an 8-slot
//              ring (the stagers no longer cover the HBM latency).  The chain wave shares its SIMD with the two light
//              waves and one emitter (a stager there slows the chain, and falls behind itself).
//   MitmK<8>   two workgroups per CU (half the waves, half the LDS) for batches with more sweeps than CUs: every wave of
//              the step is latency-bound, so two chains per CU use the vector pipes the single chain leaves idle.
template <int WAVES>
struct MitmK;
template <>
struct MitmK<16> {
  static constexpr int kSlots = 9;    // LDS ring depth in blocks: factors, references, own checkpoints
#if WFL_MITM_STATS
  static constexpr int kPSlots = 5;   // (the statistics build keeps 64 block clocks in LDS)
#else
  static constexpr int kPSlots = 6;   // partner checkpoints handed from the fetcher to the emitters
#endif
  static constexpr int kStagers = 4, kEmitters = 9, kWaves = 16;
  __device__ static __forceinline__ void role(int wave, int& role, int& idx) {
    switch (wave) {  // (a switch on a scalar: compiled to scalar compares)
      case 0: role = 0, idx = 0; break;
      case 4: role = 2, idx = 0; break;
      case 8: role = 3, idx = 0; break;
      case 12: role = 4, idx = 8; break;
      case 1: role = 1, idx = 0; break;
      case 2: role = 1, idx = 1; break;
      case 3: role = 1, idx = 2; break;
      case 5: role = 1, idx = 3; break;
      case 6: role = 4, idx = 0; break;
      case 7: role = 4, idx = 1; break;
      case 9: role = 4, idx = 2; break;
      case 10: role = 4, idx = 3; break;
      case 11: role = 4, idx = 4; break;
      case 13: role = 4, idx = 5; break;
      case 14: role = 4, idx = 6; break;
      default: role = 4, idx = 7; break;
    }
  }
};
template <>
struct MitmK<8> {
  static constexpr int kSlots = 5, kPSlots = 3;
  static constexpr int kStagers = 2, kEmitters = 3, kWaves = 8;
  __device__ static __forceinline__ void role(int wave, int& role, int& idx) {
    switch (wave) {
      case 0: role = 0, idx = 0; break;
      case 4: role = 2, idx = 0; break;
      case 1: role = 3, idx = 0; break;
      case 2: role = 1, idx = 0; break;
      case 3: role = 1, idx = 1; break;
      case 5: role = 4, idx = 0; break;
      case 6: role = 4, idx = 1; break;
      default: role = 4, idx = 2; break;
    }
  }
};

typedef float mv4f __attribute__((ext_vector_type(4)));
template <class K>
struct MitmLds {
  mv4f ring[K::kSlots][kBlk / 2][64];  // (fb, fl) of two frames per entry and lane: 72 KiB
  float4 pck[K::kPSlots][64];        // partner state after the block, own lane order: (bb, bl, eb bits, -)
  float4 ck[K::kSlots][64];          // own state before block n: (blank mantissa, label mantissa, lane exponent bits, -)
  float fref[K::kSlots][kBlk];       // per-frame references r_t (integer valued; 0 past the block's frames)
  double offc[K::kSlots];            // sum of the references of all blocks before n
  double poff[K::kPSlots];           // the partner's sum before its checkpoint
  double offtot;
  int staged[K::kSlots];             // == n + 1 once block n sits in slot n % K::kSlots
  int pready[K::kPSlots];            // == n + 1 once the partner checkpoint for block n sits in slot n % K::kPSlots
  int egrab[K::kSlots];              // == n + 1 once an emitter holds block n's factors and own checkpoint in registers
  int pgrab[K::kPSlots];             // == n + 1 once an emitter holds the partner checkpoint for block n
  double zref;                     // log2 Z as the sweep's first emitted block reproduced it (every later block normalises by it)
  int zready;
  int enext;                       // next block to emit: the emitters take blocks as they become free (a static round robin
                                   // lets the slowest emitter -- the one next to the chain wave -- hold up the whole ring)
  int chainpos;                    // p: the chain wave holds the factors of all blocks < p in registers
  int ckpos;                       // q: the chain wave has written its checkpoints < q (block n's: when it enters the block)
  int ckdone;                      // ... picked up by the flusher (offc valid)
  int offdone;                     // offtot valid
  int cmap_ready;                  // (wide rows) the column -> slot map behind the emitters' tiles is written
  int abort;                       // a wave of the workgroup gave up waiting: the others stop waiting too
#if WFL_MITM_STATS
  long long blk_t[64];             // chain wave: clock at the start of every block (the first 64)
#endif
};


// A wave whose hand-off never arrives gives up -- cleanly: it raises the launch's status word (every utterance then
// counts as rejected: utterance_rejected) and its doubt word (the repair launch behind this one recomputes the batch in
// the log domain and the caller still receives correct results), tells the other waves of its workgroup (which stop
// waiting as well) and ENDS.  No trap: the HIP context survives, the step reports through WFL_CTC_WS_STATUS and the
// caller's host_state words (the next calls take the log-domain launch).
template <class K>
__device__ __forceinline__ void mitm_give_up(const CtcArgs& a, const CtcWs& w, MitmLds<K>& S) {
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_fetch_or((int32_t*)(a.ws + w.perr), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    coherent_store64(a.ws + w.suspect, a.token);
  }
  lds_post(&S.abort, 1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_endpgm();
}
// (looked at every 256th poll: a wave of the workgroup has given up)
template <class K>
__device__ __forceinline__ bool mitm_waited_out(const CtcArgs& a, MitmLds<K>& S, int spin) {
  return spin > a.spin || ((spin & 255) == 255 && lds_peek(&S.abort) != 0);
}

__host__ __device__ inline int mitm_first_emitted(int NB, int dir) { return dir == 0 ? NB / 2 : NB - NB / 2; }
// published checkpoints of sweep (b, dir): [H0][2][P] 8-byte entries (mantissa bits | exponent << 32), blank states
// then label states, inside the (enlarged) ck region; stride per sweep (NB + 1) * P float2
__device__ __forceinline__ unsigned long long* mitm_pub(const CtcArgs& a, const CtcWs& w, int b, int dir, int NB) {
  return (unsigned long long*)(a.ws + w.ck) + (int64_t)(b * 2 + dir) * (NB + 1) * a.P;
}

#if WFL_MITM_STATS
#define MITM_T0() const long long _t0 = clock64()
#define MITM_ACC(var) var += clock64() - _t0
#else
#define MITM_T0()
#define MITM_ACC(var)
#endif

typedef float mv2f __attribute__((ext_vector_type(2)));

// One 16-frame block of a sweep's second half: gradient rows from the sweep's own state before the block (sa, ea),
// the partner's state after it (pk) and the block's emission factors (F), all in the sweep's own lane order.
//   own_j(s)  = ma_j(s) 2^ea(s)    forward through the block (the chain's five-instruction frame)
//   part_j(s) = mb_j(s) 2^eb(s)    backwards, with K(s) = cf 2^(ea + eb - E) / Zm folded into its start: the coupling
//                                  factor of the scaled recursion is 2^(ea[i] - ea[i+1]) (bounded by the chain's clamp)
//   posterior_j(s) = own_j(s) [A part_{j+1}](s)   -- one packed multiply per frame and lane pair
// DIR: tile row of frame j (the reversed sweep only emits complete blocks, FULL).
// LSM (fused log_softmax: the gradient w.r.t. RAW scores): the rows leave as  cf (gamma - softmax(x)),  the softmax
// term formed from the raw rows (`xsrc`, laid out like `dst`) and the rows' log-sum-exps (`lse_rows`: lane r < 16 holds
// that of tile row r) while the tile is written out -- the sweeps never see more than the gathered target columns.
template <class K, int DIR, bool FULL, bool WIDE, bool LSM = false>
__device__ __forceinline__ void ctc_mitm_emit_block(const mv2f (&F)[kBlk], mv2f sa, int ea, float4 pk, float rsum, double off_sum,
                                                    MitmLds<K>& S, bool first,
                                                    int cnt, float cf, float g, float gs, bool skipn, bool owner, bool adder,
                                                    int lane, float* rows, const unsigned char* cmap, int ycol, int blank, int C, long long* zmm,
                                                    int32_t* zcnt, unsigned long long* suspect, unsigned long long suspect_token,
                                                    int32_t* perr, int spin_bound, float* __restrict__ dst, long long* st_part,
                                                    const float* __restrict__ xsrc = nullptr, float lse_rows = 0.f) {
#if WFL_MITM_STATS
  const long long st_e0 = clock64();
#endif
  const int eb = __float_as_int(pk.z);
  // ---- own sweep forward through the block, kept in registers
  mv2f pa[kBlk];
  const mv2f G = {g, gs};
#pragma unroll
  for (int j = 0; j < kBlk; ++j) {
    if (FULL || j < cnt) {
      const mv2f t = F[j] * G;
      const mv2f o = F[j] * mv2f{sa.x, sa.x};
      float ox = o.x, oy = o.y;
      fmac2_shr1(ox, oy, sa.y, t.x, t.y);
      oy = fmaf(F[j].y, sa.y, oy);
      sa = mv2f{ox, oy};
    }
    pa[j] = sa;
  }
#if WFL_MITM_STATS
  {
    // (the clock is read after the forward recursion has produced its last value)
    const long long now = __builtin_amdgcn_readfirstlane(__float_as_int(sa.x)) * 0 + clock64();
    st_part[0] += now - st_e0;
  }
#endif
  // ---- Kn(s) = cf 2^(ea + eb) / Z in the block's units.  The emitter's FIRST block reproduces Z itself -- sum_s own(s)
  // [A partner](s) at its last frame, the certificate's identity -- and folds its log2 Z into the utterance's min / max
  // for the comparison with the chain's; its later blocks reuse that log2 Z (Z is one number; the blocks only differ in
  // their offsets, integer-valued doubles): no prefix maximum, no wave sum, no reciprocal.  They are still certified:
  // their posteriors must sum to one per frame, which compares their own sum_s alpha beta with the reference to 2e-4.
  float bb = pk.x, bl = pk.y;
  float Kn = 0.f;
  bool alive;
  double zk = -1.0e300;  // (lane 0) log2 Z as this block reproduces it
  const int sx = ea + eb;
  if (first) {
    const int eb_next = __builtin_amdgcn_update_dpp(eb, eb, 0x130, 0xf, 0xf, false);  // wave_shl:1 (lane 63: own)
    const float h = lane == 63 ? 0.f : ldexpf(1.f, min(max(eb_next - eb, -200), 100));
    const float hs = skipn ? h : 0.f;
    const float tb0 = bb + bl;
    float tl0 = bl;
    fmac2_shl1(tl0, bb, bl, h, hs);
    const float v = sa.x * tb0 + sa.y * tl0;
    const int own = v > 0.f ? sx + __builtin_amdgcn_frexp_expf(v) : kEmptyE;
    const int E = __builtin_amdgcn_readlane(wave_prefix_max_i(own), 63);
    const float term = v > 0.f ? ldexpf(v, max(sx - E, -200)) : 0.f;
    const float Zm = wave_all_sum(term);
    alive = Zm > 0.f && Zm < 3.0e38f && E > kEmptyE;
    if (alive) Kn = cf * ldexpf(__builtin_amdgcn_rcpf(Zm), min(max(sx - E, -200), 100));  // (v_rcp_f32: 1 ulp)
    zk = alive ? off_sum + (double)E + (double)__builtin_amdgcn_logf(Zm) + (double)rsum : -1.0e300;
    if (lane == 0) S.zref = zk;
    lds_post(&S.zready, 1);
    if (lane == 0) {
      const long long zq = z_fixed(zk);
      // the minimum and the maximum FIRST, acknowledged (their return values are waited for), then the count: whoever
      // sees the count has both -- see the alpha chain's end.  (Ordered by hand: a RELEASE on the count and an ACQUIRE at
      // its reader are an L2 write-back and an L2 invalidate at agent scope on this chip -- with 256 workgroups
      // streaming their gradient rows at that moment they cost the launch 6 of its 41 us, measured.)
      const long long was_lo = __hip_atomic_fetch_min(zmm, zq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long was_hi = __hip_atomic_fetch_max(zmm + 1, zq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(was_lo), "v"(was_hi) : "memory");
      __hip_atomic_fetch_add(zcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    // (ONE reference per sweep, from its first emitted block whichever emitter took it: results do not depend on
    // which emitter served which block)
    for (int spin = 0; lds_peek(&S.zready) != 1; ++spin) {
      __builtin_amdgcn_s_sleep(2);
      if (spin > spin_bound || ((spin & 255) == 255 && lds_peek(&S.abort) != 0)) {
        // (see mitm_give_up: this function has the doubt word and the status word's address)
        if (lane == 0) {
          __hip_atomic_fetch_or(perr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          coherent_store64(suspect, suspect_token);
        }
        lds_post(&S.abort, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_endpgm();
      }
    }
    asm volatile("" ::: "memory");
    const double zref = S.zref;
    alive = zref > -1.0e299;
    if (alive) {
      const double D = off_sum + (double)rsum - zref;  // = -(E + log2 Zm) of this block
      const double Di = floor(D);
      const float frac = __builtin_amdgcn_exp2f((float)(D - Di));
      Kn = cf * frac * ldexpf(1.f, min(max(sx + (int)Di, -200), 100));
    }
  }
#if WFL_MITM_STATS
  st_part[2] += __builtin_amdgcn_readfirstlane(__float_as_int(Kn)) * 0 + clock64() - st_e0;
#endif
  // ---- partner sweep backwards through the block (scaled by Kn), posteriors
  bb *= Kn, bl *= Kn;
  const int ea_next = __builtin_amdgcn_update_dpp(ea, ea, 0x130, 0xf, 0xf, false);
  const float hk = lane == 63 ? 0.f : ldexpf(1.f, min(max(ea - ea_next, -200), 100));
  const float hks = skipn ? hk : 0.f;
  float gbv[kBlk], glv[kBlk];
  mv2f wsum = {0.f, 0.f};  // sum_j w_j (gb, gl)[j], w_j = 1 + j / 32: the frames' posterior mass, weighted so that errors cannot cancel
#pragma unroll
  for (int j = 0; j < kBlk; ++j) gbv[j] = 0.f, glv[j] = 0.f;
#pragma unroll
  for (int j = kBlk - 1; j >= 0; --j) {
    if (FULL || j < cnt) {
      // blank i -> blank i, label i.  label i -> label i, blank i+1 (and label i+1 if allowed)
      const float tb = bb + bl;
      float tl = bl;
      fmac2_shl1(tl, bb, bl, hk, hks);
      const mv2f t = {tb, tl};
      const mv2f gam = pa[j] * t;
      gbv[j] = gam.x, glv[j] = gam.y;
      const float wj = 1.f + (float)j * (1.f / 32.f);
      wsum = gam * mv2f{wj, wj} + wsum;
      const mv2f nb = t * F[j];
      bb = nb.x, bl = nb.y;
    }
  }
#if WFL_MITM_STATS
  st_part[3] += __builtin_amdgcn_readfirstlane(__float_as_int(bb + wsum.x)) * 0 + clock64() - st_e0;
#endif
  // ---- gradient rows
  constexpr int R0 = DIR == 0 ? 0 : kBlk - 1, RS = DIR == 0 ? 1 : -1;  // tile row of frame j: R0 + RS * j
  const float gtot = fold16_sum(gbv, lane);  // lane l < 16: blank posterior of frame l
  // (WIDE: the compact tile of ctc_kernels.hip -- [16][kCS] floats, one slot per target position, 63 the blank, 64 zero --
  // `ycol` is then the slot, `blank` 63, and the dense rows are expanded through the column map while they are written)
  constexpr int TS = WIDE ? kCS : kMTile;
  if (lane < (FULL ? kBlk : cnt)) rows[(R0 + RS * lane) * TS + blank] = gtot;
  float* cell = rows + ycol;
  if (owner) {
#pragma unroll
    for (int j = 0; j < kBlk; ++j)
      if (FULL || j < cnt) cell[(R0 + RS * j) * TS] = glv[j];
  }
  if (adder) {
#pragma unroll
    for (int j = 0; j < kBlk; ++j)
      if (FULL || j < cnt) atomicAdd(&cell[(R0 + RS * j) * TS], glv[j]);
  }
#if WFL_MITM_STATS
  st_part[4] += __builtin_amdgcn_readfirstlane(__float_as_int(gtot)) * 0 + clock64() - st_e0;
#endif
  {
    // certificate: the posterior mass of every frame
    const float stot = wave_all_sum(wsum.x + wsum.y);
    const int n = FULL ? kBlk : cnt;
    const float want = cf * ((float)n + (float)(n * (n - 1)) * (1.f / 64.f));
    const bool bad_block = alive && !(fabsf(stot - want) <= 2e-4f * fabsf(cf));
    if (bad_block && lane == 0) {
      __hip_atomic_fetch_min(zmm, kZDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      coherent_store64(suspect, suspect_token);
    }
  }
  // (rows are private to the wave: LDS operations of one wave complete in order, no barrier needed)
#if WFL_MITM_STATS
  const long long st_e1 = clock64();
  st_part[5] += st_e1 - st_e0;
  struct StAcc {
    long long* p;
    long long t;
    __device__ ~StAcc() { p[1] += clock64() - t; }
  } st_acc{st_part, st_e1};
#endif
  const int nrows = FULL ? kBlk : cnt;
  if (WFL_MITM_ABL & 64) return;  // (scratch: a launch that computes the rows and does not store them)
  const bool soft = LSM && alive;  // (no accepting path: zero gradient, softmax term included)
  auto softterm = [&](float xv, float l) { return l > WFL_NEG_INF ? cf * __expf((xv == xv ? xv : WFL_NEG_INF) - l) : 0.f; };
  if (WIDE) {
    asm volatile("" ::: "memory");  // (the tile was written through float pointers)
    if (LSM)
      compact_expand(rows, cmap, dst, xsrc, lse_rows, nrows, C, cf, true, soft, lane);
    else
      compact_expand(rows, cmap, dst, nullptr, 0.f, nrows, C, 1.f, true, false, lane);
    return;
  }
  if ((C & 3) == 0 && (((uintptr_t)dst) & 15) == 0 && (!LSM || (((uintptr_t)xsrc) & 15) == 0)) {
    // two rows per instruction: lanes 0..31 one row, lanes 32..63 the next (C / 4 <= 32 float4 per row)
    const int half = lane >> 5, c4 = lane & 31;
    const float4* src = (const float4*)rows + half * (kMTile / 4) + c4;
    float4* out = (float4*)dst + half * (C >> 2) + c4;
    const bool act = c4 < (C >> 2);
    typedef float nf4 __attribute__((ext_vector_type(4)));
    auto store = [&](int r, const nf4& q) {
      if (act && (FULL || r + half < nrows)) {
        // write-through (sc0 sc1): the rows leave the L2 as they are produced instead of at the end of the kernel, when
        // 30 MB of dirty lines would be written back at once (measured: 46.5 -> 44.7 us; non-temporal stores: 61 us)
#if WFL_MITM_STORE == 2
        // (s_nop: a VALU write of the data registers right behind a store of more than 64 bits is a hazard the compiler
        // cannot see through the asm)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(&out[(r >> 1) * (C >> 1)]), "v"(q) : "memory");
#else
        *(nf4*)&out[(r >> 1) * (C >> 1)] = q;
#endif
      }
    };
    if (LSM) {
      // in WFL_MITM_LSM_ROWS rows at a time (8: two halves; the raw rows of all sixteen in registers next to the tile spill 64 of them):
      // the half's raw rows requested first (clamped to the block's rows), consumed behind its tile reads
      // (the raw rows' stride through an opaque asm: their per-lane addresses are then formed HERE -- hoisted above the
      // block's recursion they were four 64-bit values per lane held across it, spilled to scratch)
      int Cx = C;
      asm volatile("" : "+s"(Cx));
#pragma unroll
      for (int h8 = 0; h8 < kBlk; h8 += kLsmRows) {
        nf4 xv[kLsmRows / 2], v[kLsmRows / 2];
#pragma unroll
        for (int r = 0; r < kLsmRows; r += 2)
          xv[r >> 1] = *((const nf4*)(xsrc + (int64_t)min(h8 + r + half, nrows - 1) * Cx) + (act ? c4 : 0));
        asm volatile("" ::: "memory");  // (the tile was written through float pointers)
#pragma unroll
        for (int r = 0; r < kLsmRows; r += 2) {
          const float4 q = src[((h8 + r) >> 1) * (kMTile / 2)];
          v[r >> 1] = nf4{q.x, q.y, q.z, q.w};
        }
        if (soft) {
#pragma unroll
          for (int r = 0; r < kLsmRows; r += 2) {
            const float l0 = readlane_f(lse_rows, h8 + r), l1 = readlane_f(lse_rows, h8 + r + 1);
            const float l = half ? l1 : l0;
            nf4& o = v[r >> 1];
            const nf4 xq = xv[r >> 1];
            o.x -= softterm(xq.x, l), o.y -= softterm(xq.y, l), o.z -= softterm(xq.z, l), o.w -= softterm(xq.w, l);
          }
        }
#pragma unroll
        for (int r = 0; r < kLsmRows; r += 2) store(h8 + r, v[r >> 1]);
      }
    } else {
      // all the tile reads first, then the stores: a store waits for its own read only (the stores are volatile asm: the
      // compiler does not move a read across one, and read / wait / store eight times over is eight exposed LDS round trips)
      nf4 v[kBlk / 2];
      asm volatile("" ::: "memory");  // (the tile was written through float pointers)
#pragma unroll
      for (int r = 0; r < kBlk; r += 2) {
        const float4 q = src[(r >> 1) * (kMTile / 2)];
        v[r >> 1] = nf4{q.x, q.y, q.z, q.w};
      }
#pragma unroll
      for (int r = 0; r < kBlk; r += 2) store(r, v[r >> 1]);
    }
  } else {
#ifndef WFL_MITM_NO_UNALIGNED  // (scratch: instruction counts of the aligned path alone)
    // (rows that are not 16-byte aligned: C % 4 != 0 or an offset view.  The row stride passes through an opaque asm so
    // that nothing of this loop is computed ahead of the emitter's block loop: unrolled and hoisted, its row offsets were
    // ~580 SGPRs spilled to VGPR lanes in front of that loop -- in every launch, also those that never come here)
    int Cs = C;
    asm volatile("" : "+s"(Cs));
#pragma unroll 1
    for (int r = 0; r < nrows; ++r) {
      const float l = LSM ? readlane_f(lse_rows, r) : 0.f;
#pragma unroll 1
      for (int c = lane; c < Cs; c += 64)
        dst[r * Cs + c] = rows[r * kMTile + c] - ((LSM && soft) ? softterm(xsrc[r * Cs + c], l) : 0.f);
    }
#endif
  }
}

template <class K, bool LSM, int DIR, bool WIDE>
__device__ __forceinline__ void ctc_mitm_emitter(const CtcArgs& a, MitmLds<K>& S, int b, int em, int H0, int NB, int L, int y,
                                                 bool skip, bool skipn, bool has_label, const float* __restrict__ coef,
                                                 const float* __restrict__ gout, float* __restrict__ dx, char* smem,
                                                 const CtcWs& w) {
  const int lane = threadIdx.x & 63;
  const int T = a.T, C = a.C;
  constexpr int kTileFloats = WIDE ? kBlk * kCS : kBlk * kMTile;
  float* const tiles = (float*)(smem + ((sizeof(MitmLds<K>) + 15) & ~(size_t)15));
  float* rows = tiles + (size_t)em * kTileFloats;
  unsigned char* cmap = (unsigned char*)(tiles + (size_t)K::kEmitters * kTileFloats);  // (WIDE) [C] bytes, one per workgroup
  for (int i = lane; i < kTileFloats; i += 64) rows[i] = 0.f;
  // Labels that occur once in the target own their gradient column: plain ds_write.  A repeated label's first
  // occurrence owns the column, the others add to it afterwards; a target label equal to the blank index adds to
  // the blank column.  (Columns are the same for every block of the sweep: the tile is zeroed once.)
  bool owner = has_label && y != a.blank;
  int first = lane;  // lane of the label's first occurrence: its slot in the compact tile
  for (int j = 0; j < L; ++j) {
    const bool same = __builtin_amdgcn_readlane(y, j) == y;
    if (same && j < lane) owner = false, first = min(first, j);
  }
  const bool adder = has_label && !owner;
  const int ycol = WIDE ? (has_label ? (y == a.blank ? 63 : first) : 64) : (has_label ? y : 0);
  if (WIDE) {
    // the column -> slot map of the utterance, written by emitter 0 (a wave's DS operations execute in order: no
    // barrier between the fill and the labels' entries), the other emitters look at the flag before their first block
    if (em == 0) {
      for (int i = lane; i < ((C + 15) & ~15) / 4; i += 64) ((unsigned int*)cmap)[i] = 0x40404040u;  // 64: the zero slot
      if (owner) cmap[y] = (unsigned char)lane;
      if (lane == 0) cmap[a.blank] = 63;
      lds_post(&S.cmap_ready, 1);
    } else {
      for (int spin = 0; lds_peek(&S.cmap_ready) != 1; ++spin) {
        __builtin_amdgcn_s_sleep(8);
        if (mitm_waited_out(a, S, spin)) mitm_give_up(a, w, S);
      }
    }
  }
  const float cf = (coef ? coef[b] : 1.f) * (gout ? gout[0] : 1.f);
  long long* zmm = (long long*)(a.ws + w.zloc) + (int64_t)b * 2;
  int32_t* zcnt = (int32_t*)(a.ws + w.zcnt) + b;
  unsigned long long* suspect = (unsigned long long*)(a.ws + w.suspect);
#if WFL_MITM_STATS
  long long st_wait0 = 0, st_wait1 = 0, st_wait2 = 0;
  long long st_part[6] = {0, 0, 0, 0, 0, 0};
  const long long st_begin = clock64();
#else
  long long* st_part = nullptr;
#endif
  auto give_up = [&]() { mitm_give_up(a, w, S); };
  for (;;) {
    int n = 0;
    if (lane == 0) n = atomicAdd(&S.enext, 1);
    n = __builtin_amdgcn_readfirstlane(n);
    if (n >= NB) break;
    const int k = DIR == 0 ? n : NB - 1 - n;
    const int t0 = k * kBlk, cnt = min(kBlk, T - t0);
    const int slot = n % K::kSlots, ps = n % K::kPSlots;
    {
      int spin = 0;
      MITM_T0();
      while (lds_peek(&S.ckdone) < n + 1) {
        __builtin_amdgcn_s_sleep(4);
        if (mitm_waited_out(a, S, ++spin)) give_up();
      }
      MITM_ACC(st_wait0);
    }
    asm volatile("" ::: "memory");
    mv2f F[kBlk];
#pragma unroll
    for (int j = 0; j < kBlk; j += 2) {
      const mv4f f = S.ring[slot][j >> 1][lane];
      F[j] = mv2f{f.x, f.y}, F[j + 1] = mv2f{f.z, f.w};
    }
    const float4 cko = S.ck[slot][lane];
    const int ea = __float_as_int(cko.z);
    const float rr = lane < kBlk ? S.fref[slot][lane] : 0.f;
    const double off_own = S.offc[slot];
    lds_post(&S.egrab[slot], n + 1);  // (after the reads were issued)
    {
      int spin = 0;
      MITM_T0();
      while (lds_peek(&S.pready[ps]) != n + 1) {
        __builtin_amdgcn_s_sleep(4);
        if (mitm_waited_out(a, S, ++spin)) give_up();
      }
      MITM_ACC(st_wait1);
    }
    asm volatile("" ::: "memory");
    const float4 pk = S.pck[ps][lane];
    const double off_sum = off_own + S.poff[ps];
    lds_post(&S.pgrab[ps], n + 1);
    MITM_T0();
    const float rsum = wave_all_sum(rr);
    const int ea_prev = wave_shr1_i(ea, ea);
    const float g = lane == 0 ? 0.f : ldexpf(1.f, min(max(ea_prev - ea, -200), kMaxCouple));
    const float gs = skip ? g : 0.f;
    float* dst = dx + ((int64_t)b * T + ((WFL_MITM_ABL & 256) ? (t0 & 63) : t0)) * C;  // (256, scratch: the same stores, no HBM traffic)
    if (WFL_MITM_ABL & 128) {  // (scratch: the hand-offs without the block's arithmetic and stores)
      if (n == H0) lds_post(&S.zready, 1);
      continue;
    }
    const float* xsrc = LSM ? a.x + ((int64_t)b * T + t0) * C : nullptr;
    const float lse_rows = LSM ? a.row_lse[(int64_t)b * T + min(t0 + (lane & 15), T - 1)] : 0.f;  // (tile row r = frame t0 + r)
    if (cnt == kBlk)
      ctc_mitm_emit_block<K, DIR, true, WIDE, LSM>(F, mv2f{cko.x, cko.y}, ea, pk, rsum, off_sum, S, n == H0, cnt, cf, g, gs, skipn, owner, adder, lane, rows,
                                     cmap, ycol, WIDE ? 63 : a.blank, C, zmm, zcnt, suspect, a.token, (int32_t*)(a.ws + w.perr), a.spin, dst, st_part, xsrc, lse_rows);
    else if (DIR == 0)
      ctc_mitm_emit_block<K, 0, false, WIDE, LSM>(F, mv2f{cko.x, cko.y}, ea, pk, rsum, off_sum, S, n == H0, cnt, cf, g, gs, skipn, owner, adder, lane, rows,
                                    cmap, ycol, WIDE ? 63 : a.blank, C, zmm, zcnt, suspect, a.token, (int32_t*)(a.ws + w.perr), a.spin, dst, st_part, xsrc, lse_rows);
    MITM_ACC(st_wait2);
  }
#if WFL_MITM_STATS
  if (lane == 0) {
    const int wave = threadIdx.x >> 6;
    long long* d = (long long*)(a.ws + w.dbg) + ((int64_t)(b * 2 + DIR) * K::kWaves + wave) * 8;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    d[0] = clock64() - st_begin, d[1] = st_wait0, d[2] = st_wait1, d[3] = st_wait2, d[4] = st_part[0], d[5] = hw;
    d[6] = wall_clock64(), d[7] = st_part[1];
    long long* x = (long long*)(a.ws + w.dbg) + 2 * 8 * K::kWaves * (int64_t)a.B + (int64_t)(b * 2 + DIR) * 256 + 128 + em * 8;
    for (int q = 0; q < 6; ++q) x[q] = st_part[q];
  }
#endif
}

template <class K, bool LSM, bool WIDE>
__device__ __forceinline__ void ctc_mitm_body(const CtcArgs& a, int b, int dir, const float* __restrict__ coef,
                                              const float* __restrict__ gout, float* __restrict__ dx, char* smem) {
  MitmLds<K>& S = *reinterpret_cast<MitmLds<K>*>(smem);
#if WFL_MITM_STATS
  const long long st_wall0 = wall_clock64();
#endif
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int T = a.T, C = a.C, P = a.P;
  // The labels sit behind the offsets: two dependent round trips before the first gather can even be addressed (~1.5 us
  // each at kernel entry, when every workgroup of the chip asks at once).  Batches whose targets all have the batch's
  // maximal length -- the benchmark's, and any bucketed batch -- have o0 = b (P - 1): the labels are requested under
  // that guess TOGETHER with the offsets and kept if the offsets confirm it, requested again otherwise.
  const int Lg = P - 1;
  const int64_t og = (int64_t)b * Lg;
  int yg = -1, ygp = -1, ygn = -1;
  // (the guessed addresses lie inside the flat label array whenever the guess can be right: sum of lengths = B (P - 1);
  // otherwise they may lie beyond it -- bounded by the array's end, which the offsets' last entry gives only later, so
  // the speculative loads are clamped to the first B (P - 1) labels' worth that a.n_labels vouches for)
  const int64_t nlab = a.n_labels;
  if (lane < Lg && og + Lg <= nlab) yg = a.targets[og + (dir == 0 ? lane : Lg - 1 - lane)];
  if (lane >= 1 && lane - 1 < Lg && og + Lg <= nlab) ygp = a.targets[og + (dir == 0 ? lane - 1 : Lg - lane)];
  if (lane + 1 < Lg && og + Lg <= nlab) ygn = a.targets[og + (dir == 0 ? lane + 1 : Lg - 2 - lane)];
  const int64_t o0 = a.offsets[b];
  const int L = (int)(a.offsets[b + 1] - o0);
  int y = -1, yprev = -1, ynext = -1;  // the sweep's own orientation: position `lane` of the (reversed) target
  if (o0 == og && L == Lg && og + Lg <= nlab) {  // (uniform over the workgroup)
    y = yg, yprev = ygp, ynext = ygn;
  } else {
    if (lane < L) y = a.targets[o0 + (dir == 0 ? lane : L - 1 - lane)];
    if (lane >= 1 && lane - 1 < L) yprev = a.targets[o0 + (dir == 0 ? lane - 1 : L - lane)];
    if (lane + 1 < L) ynext = a.targets[o0 + (dir == 0 ? lane + 1 : L - 2 - lane)];
  }
  const bool has_label = lane < L, has_blank = lane <= L;
  const bool skip = has_label && lane >= 1 && y != yprev;
  const bool skipn = lane + 1 < L && ynext != y;
  const float* xrow = a.x + (int64_t)b * T * C;
  const int col = has_label ? y : a.blank;
  const CtcWs w = ctc_ws_layout(a.B, T, P);
  const int NB = ctc_blocks(T);
  const int H0 = mitm_first_emitted(NB, dir);  // blocks n < H0 are published, blocks n >= H0 emitted
  if (threadIdx.x < K::kSlots) S.staged[threadIdx.x] = 0;
  if (threadIdx.x < K::kPSlots) S.pready[threadIdx.x] = 0;
  if (threadIdx.x < K::kSlots) S.egrab[threadIdx.x] = 0;
  if (threadIdx.x < K::kPSlots) S.pgrab[threadIdx.x] = 0;
  if (threadIdx.x == 0) S.enext = mitm_first_emitted(ctc_blocks(a.T), dir), S.zready = 0;
  if (threadIdx.x == 0) S.chainpos = 0, S.ckpos = 0, S.ckdone = 0, S.offdone = 0, S.cmap_ready = 0, S.abort = 0;
#if WFL_MITM_STATS
  if (threadIdx.x < 64) S.blk_t[threadIdx.x] = 0;
#endif
  __syncthreads();
  int role, ridx;
  K::role(wave, role, ridx);
  if (role == 5) return;  // (a wave that has ended no longer counts for anything: no barrier after the first)
  if ((WFL_MITM_ABL & 4096) && role == 4) return;   // (scratch: no emitter waves at all)
  if ((WFL_MITM_ABL & (16384 | 4096)) && role == 3) return;  // (scratch: no fetcher; nobody would take its blocks)
#if WFL_MITM_STATS
  long long st_wait0 = 0, st_wait1 = 0, st_wait2 = 0, st_polls = 0;
  const long long st_begin = clock64();
  auto stats_out = [&]() {
    if (lane == 0) {
      long long* d = (long long*)(a.ws + w.dbg) + ((int64_t)(b * 2 + dir) * K::kWaves + wave) * 8;
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      d[0] = clock64() - st_begin, d[1] = st_wait0, d[2] = st_wait1, d[3] = st_wait2, d[4] = st_polls, d[5] = hw | ((long long)(xcc & 15) << 32);
      d[6] = wall_clock64(), d[7] = st_wall0;
    }
  };
#else
  auto stats_out = [&]() {};
#endif
  // a wave gives up instead of hanging the GPU when a hand-off never arrives (mitm_give_up: it ends, nothing traps)
  auto give_up = [&]() { mitm_give_up(a, w, S); };
  // has emitter-owned block m (m >= H0) been taken into registers?
  auto grabbed = [&](int m) { return lds_peek(&S.egrab[m % K::kSlots]) == m + 1; };
  auto pgrabbed = [&](int m) { return lds_peek(&S.pgrab[m % K::kPSlots]) == m + 1; };

  if (role == 1) {
    // ================================================================ stagers
    __builtin_amdgcn_s_setprio(2);  // the chain waits for them; the emitters next to them are throughput work
    const int h = ridx;
    // Wide rows from HBM (a.xc != NULL: the caller reserved [B][T][64] floats): the frames a sweep crosses in its SECOND
    // half were gathered by the partner in ITS first half -- 45 scattered columns of a 2 KB row, most of the row's
    // cache lines for 4 bytes each.  The first-half stagers therefore leave what they gathered as compact rows (slot
    // = target position in forward orientation, slot L the blank: 256 contiguous bytes per frame, write-through) with a
    // flag per block, and a second-half stager reads those instead of gathering again -- from kXcLag blocks behind the
    // middle on (the stagers run ahead of their chain, so right behind the middle the partner's rows are not there yet;
    // with the lag the flag is always up when it is looked at).  x is then gathered ~1.1 times instead of twice.
    const bool xchg = WIDE && a.xc != nullptr;
    constexpr int kXcLag = K::kSlots + 3;  // (4 .. 20 measured at cfg5: the same to 1 %)
    const int H0p = mitm_first_emitted(NB, 1 - dir);  // the partner's first half: its blocks n < H0p
    float* const xcb = xchg ? const_cast<float*>(a.xc) + (int64_t)b * T * kXcStride : nullptr;
    const int pslot = dir == 0 ? lane : (lane < L ? L - 1 - lane : lane);  // forward-orientation slot of this lane's column
    unsigned long long* const xflag = (unsigned long long*)(a.ws + w.ready);  // [b][dir][NB]: entry 1 + n (0: the half flag)
    auto issue = [&](int n, float (&raw)[kBlk]) {
      const int k = dir == 0 ? n : NB - 1 - n;
      const int t0 = k * kBlk, cnt = min(kBlk, T - t0);
      if (WFL_MITM_ABL & 2048) {  // (scratch: no gathers)
#pragma unroll
        for (int j = 0; j < kBlk; ++j) raw[j] = 0.01f * (float)(lane + j + n);
        return;
      }
      if (xchg && n >= H0 + kXcLag) {
        // the partner crossed this block as ITS block NB - 1 - n and published it
        const unsigned long long* f = xflag + (int64_t)(b * 2 + (1 - dir)) * NB + 1 + (NB - 1 - n);
        int spin = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.token) {
          __builtin_amdgcn_s_sleep(8);
          if (mitm_waited_out(a, S, ++spin)) give_up();
        }
#pragma unroll
        for (int j = 0; j < kBlk; ++j) {
          const int t = dir == 0 ? t0 + j : t0 + cnt - 1 - j;
          const unsigned bits = __hip_atomic_load(
              reinterpret_cast<const unsigned*>(xcb + (int64_t)min(max(t, 0), T - 1) * kXcStride + min(pslot, kXcStride - 1)),
              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          raw[j] = __uint_as_float(bits);
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < kBlk; ++j) {
        const int t = dir == 0 ? t0 + j : t0 + cnt - 1 - j;
        raw[j] = xrow[(int64_t)min(max(t, 0), T - 1) * C + col];  // clamped: valid address, unused past the block
      }
    };
    // first half: leave block n's gathered values for the partner (only the blocks it will read: see issue)
    auto publish_xc = [&](int n, const float (&raw)[kBlk]) {
      if (!xchg || n > NB - 1 - H0p - kXcLag) return;
      const int k = dir == 0 ? n : NB - 1 - n;
      const int t0 = k * kBlk, cnt = min(kBlk, T - t0);
      if (lane <= L) {
#pragma unroll
        for (int j = 0; j < kBlk; ++j) {
          const int t = dir == 0 ? t0 + j : t0 + cnt - 1 - j;
          if (j < cnt)
            __hip_atomic_store(reinterpret_cast<unsigned*>(xcb + (int64_t)t * kXcStride + pslot), __float_as_uint(raw[j]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows are acknowledged (and so is whatever else this wave had in flight)
      if (lane == 0)
        __hip_atomic_store(xflag + (int64_t)(b * 2 + dir) * NB + 1 + n, a.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    float lse_raw = 0.f;
    auto issue_lse = [&](int n) {
      if (LSM) {
        const int k = dir == 0 ? n : NB - 1 - n;
        const int t0 = k * kBlk, cnt = min(kBlk, T - t0);
        const int t = dir == 0 ? t0 + (lane & 15) : t0 + cnt - 1 - (lane & 15);
        lse_raw = a.row_lse[(int64_t)b * T + min(max(t, 0), T - 1)];
      }
    };
    auto stage = [&](int n, const float (&raw)[kBlk], float lse_blk) {
      const int k = dir == 0 ? n : NB - 1 - n;
      const int cnt = min(kBlk, T - k * kBlk);
      const int slot = n % K::kSlots;
      if (((WFL_MITM_ABL & 16) && n >= H0) || (WFL_MITM_ABL & 512)) {  // (scratch: what a second half fed with ready-made factors would cost)
#pragma unroll
        for (int j = 0; j < kBlk; j += 2)
          S.ring[slot][j >> 1][lane] = mv4f{has_blank ? 0.7f : 0.f, has_label ? 0.7f + 0.001f * raw[j] : 0.f, has_blank ? 0.7f : 0.f,
                                            has_label ? 0.7f + 0.001f * raw[j + 1] : 0.f};
        if (lane < kBlk) S.fref[slot][lane] = 0.f;
        return;
      }
      // The frame's reference r_t = rint(log2(e) * its largest target-label score): the maximum is taken over the RAW
      // scores (scaling by a positive constant commutes with it; v_max ignores a NaN operand -- the NaN policy's
      // "impossible" -- and an all-NaN / all -inf frame gets the reference 0), the factor is one fused multiply-add, a
      // NaN filter and the exp2:  f = 2^(max(x log2 e - r, -inf))  (max(NaN, -inf) = -inf).  Seven instructions per
      // frame and lane with the two broadcasts (the reference from lane j, the blank's factor from lane L).
      float xr[kBlk];
#pragma unroll
      for (int j = 0; j < kBlk; ++j) xr[j] = LSM ? raw[j] - readlane_f(lse_blk, j) : raw[j];
      const float m = fold16<true>(xr, lane) * kLog2e;  // every lane: the largest target-label score of frame lane % 16
      const float rr = (m > -3.0e38f && m < 3.0e38f) ? rintf(m) : 0.f;
      const float hb = has_blank ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < kBlk; j += 2) {
        const float f0 = __builtin_amdgcn_exp2f(vmax(fmaf(xr[j], kLog2e, -readlane_f(rr, j)), WFL_NEG_INF));
        const float f1 = __builtin_amdgcn_exp2f(vmax(fmaf(xr[j + 1], kLog2e, -readlane_f(rr, j + 1)), WFL_NEG_INF));
        S.ring[slot][j >> 1][lane] = mv4f{readlane_f(f0, L) * hb, has_label ? f0 : 0.f, readlane_f(f1, L) * hb, has_label ? f1 : 0.f};
      }
      if (lane < kBlk) S.fref[slot][lane] = lane < cnt ? rr : 0.f;
    };
    auto wait_slot = [&](int n) {
      if (n < K::kSlots) return;
      const int m = n - K::kSlots;  // the block that held the slot
      int spin = 0;
      MITM_T0();
      while (lds_peek(&S.chainpos) < m + 1 || lds_peek(&S.ckdone) < m + 1 || (!(WFL_MITM_ABL & 4096) && m >= H0 && !grabbed(m))) {
        __builtin_amdgcn_s_sleep(2);
        if (mitm_waited_out(a, S, ++spin)) give_up();
      }
      MITM_ACC(st_wait0);
#if WFL_MITM_STATS
      st_polls += spin;
#endif
    };
    // two blocks of gathers in flight per stager: a round trip to HBM is longer than four chain blocks
    float ra[kBlk], rb[kBlk];
    float la = 0.f, lb = 0.f;
    if (h < NB) issue(h, ra), issue_lse(h), la = lse_raw;
    if (h + K::kStagers < NB) issue(h + K::kStagers, rb), issue_lse(h + K::kStagers), lb = lse_raw;
    for (int n = h; n < NB; n += 2 * K::kStagers) {
      wait_slot(n);
      {
        MITM_T0();
        stage(n, ra, la);
        MITM_ACC(st_wait1);
      }
      lds_post(&S.staged[n % K::kSlots], n + 1);
      publish_xc(n, ra);
      if (n + 2 * K::kStagers < NB) issue(n + 2 * K::kStagers, ra), issue_lse(n + 2 * K::kStagers), la = lse_raw;
      const int n2 = n + K::kStagers;
      if (n2 < NB) {
        wait_slot(n2);
        {
          MITM_T0();
          stage(n2, rb, lb);
          MITM_ACC(st_wait1);
        }
        lds_post(&S.staged[n2 % K::kSlots], n2 + 1);
        publish_xc(n2, rb);
        if (n2 + 2 * K::kStagers < NB) issue(n2 + 2 * K::kStagers, rb), issue_lse(n2 + 2 * K::kStagers), lb = lse_raw;
      }
    }
    stats_out();
    return;
  }

  if (role == 2) {
    // ================================================================ flusher
    unsigned long long* pub = mitm_pub(a, w, b, dir, NB);
    double* offs = (double*)(a.ws + w.off) + (int64_t)(b * 2 + dir) * NB;
    unsigned long long* half = (unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + dir) * NB;  // "first half published"
    double offcum = 0.0;  // sum of r_t over the blocks before the checkpoint
    if (dir == 0 && lane == 0) {
      // certificate accumulators of the utterance.  By THIS wave: its stores are acknowledged (vmcnt(0) below) before it
      // raises the flag that lets the partner's emitters run, and before it hands the first emitted block of this
      // sweep to the local emitters (ckdone) -- no other wave has to wait for them (measured: the same two stores and a
      // release fence by the chain wave before the workgroup's barrier cost every alpha sweep 1-2 us of prologue)
      long long* zmm = (long long*)(a.ws + w.zloc) + (int64_t)b * 2;
      coherent_store64(zmm, (unsigned long long)(1ll << 62));
      coherent_store64(zmm + 1, (unsigned long long)(-(1ll << 62) - 1));
      __hip_atomic_store((int32_t*)(a.ws + w.zcnt) + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dir == 0) {
      // bit 63 of dup[b]: the target cannot be aligned at all -- T < L + adjacent repeats.  Such an utterance has
      // Z = 0 exactly, in any arithmetic: loss inf, zero gradient, nothing for the certificate to doubt
      const int repeats = __builtin_popcountll(__builtin_amdgcn_ballot_w64(has_label && lane >= 1 && y == yprev));
      if (lane == 0) coherent_store64((unsigned long long*)(a.ws + w.dup) + b, T < L + repeats ? 1ull << 63 : 0ull);
    }
    for (int kk = 0; kk < NB; ++kk) {
      {
        int spin = 0;
        MITM_T0();
        while (lds_peek(&S.ckpos) < kk + 1) {
          __builtin_amdgcn_s_sleep((WFL_MITM_ABL & 1024) ? 8 : 1);
          if (mitm_waited_out(a, S, ++spin)) give_up();
        }
        MITM_ACC(st_wait0);
      }
      asm volatile("" ::: "memory");
      const int slot = kk % K::kSlots;
      const float rj = lane < kBlk ? S.fref[slot][lane] : 0.f;
      if (kk < H0 && !(WFL_MITM_ABL & 8192)) {
        // Device-coherent stores (see ctc_log_chain_body), three per checkpoint and never waited for one by one: the
        // partner only needs them when this sweep has finished its first half (it emits from the middle outwards, so
        // the LAST checkpoint published is the first one it uses) -- ONE flag per sweep, raised after the stores of
        // the whole half have been acknowledged.
        const float4 m = S.ck[slot][lane];
        const int e = __float_as_int(m.z);
        const unsigned long long vb = (unsigned long long)__float_as_uint(m.x) | ((unsigned long long)(unsigned)e << 32);
        const unsigned long long vl = (unsigned long long)__float_as_uint(m.y) | ((unsigned long long)(unsigned)e << 32);
        unsigned long long obits;
        __builtin_memcpy(&obits, &offcum, 8);
        unsigned long long* dstb = pub + (int64_t)kk * 2 * P + min(lane, P - 1);
        unsigned long long* dstl = dstb + P;
        if (lane < P) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dstb), "v"(vb) : "memory");
        if (lane < P) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dstl), "v"(vl) : "memory");
        if (lane == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(&offs[kk]), "v"(obits) : "memory");
        if (kk == H0 - 1) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) coherent_store64(half, a.token);
        }
      }
      if (H0 == 0 && kk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing published: the accumulators' stores)
      if (lane == 0) S.offc[slot] = offcum;
      lds_post(&S.ckdone, kk + 1);
      offcum += (double)wave_all_sum(rj);
    }
    if (lane == 0) S.offtot = offcum;
    lds_post(&S.offdone, 1);
    stats_out();
    return;
  }

  if (role == 3) {
    // ================================================================ fetcher
    const unsigned long long* ppub = mitm_pub(a, w, b, 1 - dir, NB);
    const double* poffs = (const double*)(a.ws + w.off) + (int64_t)(b * 2 + 1 - dir) * NB;
    unsigned long long* phalf = (unsigned long long*)(a.ws + w.ready) + (int64_t)(b * 2 + 1 - dir) * NB;
    if (H0 < NB) {
      int spin = 0;
      MITM_T0();
      while (__hip_atomic_load(phalf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.token) {
        __builtin_amdgcn_s_sleep(8);
        if (mitm_waited_out(a, S, (++spin) << 3)) give_up();  // (polls of s_sleep 8 on another CU's word)
#if WFL_MITM_STATS
        ++st_polls;
#endif
      }
      MITM_ACC(st_wait1);
    }
    // everything the partner published is there now: its checkpoints stream in, kMFetch blocks of loads in flight
    // mirrored lanes: our blank state 2i is the partner's blank of its position L - i, our label i its label
    // L - 1 - i; the pair (bb, bl) of a lane gets one exponent (the larger).  The partner's exponents obey the
    // neighbour clamp in ITS lane order, which is exactly eb[i] >= eb[i+1] - kGap here.
    constexpr int kMFetch = 4;
    unsigned long long vb[kMFetch], vl[kMFetch];
    double po[kMFetch];
    auto issue = [&](int n, unsigned long long& rb, unsigned long long& rl, double& ro) {
      const int q = NB - 1 - n;  // the partner's checkpoint: its state before ITS block q = after our block n
      rb = 0, rl = 0, ro = 0.0;
      if (n < NB) {
        if (lane <= L) rb = coherent_load64(&ppub[(int64_t)q * 2 * P + (L - lane)]);
        if (lane < L) rl = coherent_load64(&ppub[(int64_t)q * 2 * P + P + (L - 1 - lane)]);
        if (lane == 0) ro = coherent_load_f64(&poffs[q]);
      }
    };
#pragma unroll
    for (int u = 0; u < kMFetch; ++u) issue(H0 + u, vb[u], vl[u], po[u]);
    for (int n0 = H0; n0 < NB; n0 += kMFetch) {
#pragma unroll
      for (int u = 0; u < kMFetch; ++u) {
        const int n = n0 + u;
        if (n < NB) {
          if (n - H0 >= K::kPSlots) {  // slot n % K::kPSlots held block n - K::kPSlots
            int spin = 0;
            MITM_T0();
            while (!pgrabbed(n - K::kPSlots)) {
              __builtin_amdgcn_s_sleep(4);
              if (mitm_waited_out(a, S, ++spin)) give_up();
            }
            MITM_ACC(st_wait0);
          }
          float bb = 0.f, bl = 0.f;
          int eb = kEmptyE;
          if (lane <= L) {
            const int e_b = (int)(unsigned)(vb[u] >> 32), e_l = lane < L ? (int)(unsigned)(vl[u] >> 32) : e_b;
            eb = max(e_b, e_l);
            bb = ldexpf(__uint_as_float((unsigned)vb[u]), max(e_b - eb, -200));
            bl = lane < L ? ldexpf(__uint_as_float((unsigned)vl[u]), max(e_l - eb, -200)) : 0.f;
          }
          const int ps = n % K::kPSlots;
          S.pck[ps][lane] = make_float4(bb, bl, __int_as_float(eb), 0.f);
          if (lane == 0) S.poff[ps] = po[u];
          lds_post(&S.pready[ps], n + 1);
        }
        issue(n + kMFetch, vb[u], vl[u], po[u]);
      }
    }
    if (H0 < NB && lane == 0) coherent_store64(phalf, 0ull);  // sole consumer of the flag: leave it cleared (graph replays)
    stats_out();
    return;
  }

  if (role == 0) {
    // ================================================================ the chain
    __builtin_amdgcn_s_setprio(3);
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f P = {(lane == 0) ? 1.f : 0.f, 0.f};  // virtual slot "before the first frame": only state 0 alive
    int e = 0;
    v2f G = {0.f, 0.f};
    bool bad = false;
    const unsigned long long skipmask = __builtin_amdgcn_ballot_w64(skip);
    // coupling factors of the interval: g = 2^(e[i-1] - e[i]) (lane 0 has no source lane: the DPP subtraction leaves its
    // -300 alone and 2^-300 is zero), bounded by 2^kMaxCouple -- after a full renormalisation by 2^kGap by construction
    auto coupling = [&]() {
      int d = -300;
      asm("s_nop 1\n\tv_sub_u32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(e));
      const float g = ldexpf(1.f, min(d, kMaxCouple));
      G = v2f{g, skip ? g : 0.f};
    };
    // per-lane power-of-two renormalisation WITH the neighbour-gap clamp (a wave-wide prefix maximum: ~146 cycles)
    auto renorm_full = [&]() {
      // (no overflow test here: an inf or NaN mantissa stays one -- ldexp, the frames' multiply-adds -- and is
      // seen after the last frame; the blocks' certificates see it as well)
      const float mx = vmax(P.x, P.y);
      const int k = __builtin_amdgcn_frexp_expf(mx);  // 0 for mx == 0
      const int own = mx > 0.f ? e + k : kEmptyE;
      const int pre = wave_prefix_max_i(own + kGap * lane) - kGap * lane;  // >= e[i-1] - kGap (transitively)
      // mantissa to [0.5, 1), then down by what the clamp pulled the exponent up: ONE scaling by 2^(e - pre)
      const int shf = e - pre;
      P.x = ldexpf(P.x, shf), P.y = ldexpf(P.y, shf);
      e = pre;
      coupling();
    };
    // ... and WITHOUT it: every lane by its own exponent (an empty lane keeps the exponent it has).  What the clamp is
    // for -- a lane far below its neighbour must not be flooded by the neighbour's inflow, 2^(e[i-1] - e[i]) times a
    // mantissa -- only needs the gap bounded by float's range, not by kGap: the clamp is renewed every fourth block
    // (and in every block of the start, while lanes are still empty) and may slip in between.  ~55 cycles.
    auto renorm_own = [&]() {
      const float mx = vmax(P.x, P.y);
      const int nk = -__builtin_amdgcn_frexp_expf(mx);  // 0 for mx == 0 (and for inf / NaN)
      P.x = ldexpf(P.x, nk), P.y = ldexpf(P.y, nk);
      e -= nk;
      coupling();
    };
    auto frame = [&](const v2f f) {
      const float c0 = f.x * G.x, c1 = f.y * G.y;
      float t0 = f.x * P.x, t1 = f.y * P.x;
      fmac2_shr1(t0, t1, P.y, c0, c1);
      P.y = fmaf(f.y, P.y, t1);
      P.x = t0;
    };
    int s0 = 0, s1 = 1 % K::kSlots, s2 = 2 % K::kSlots;  // ring slots of blocks kk, kk + 1, kk + 2
    // wait until the factors of block kk + 1 are staged; `sf` is what the flag read a block ago
    auto staged_next = [&](int kk, int sf) {
      if (kk + 1 >= NB) return;
      if (__builtin_expect(__builtin_amdgcn_readfirstlane(sf) != kk + 2, 0)) {
        int spin = 0;
        MITM_T0();
        while (lds_peek(&S.staged[s1]) != kk + 2)
          if (mitm_waited_out(a, S, ++spin)) give_up();
        MITM_ACC(st_wait1);
#if WFL_MITM_STATS
        st_polls += spin;
#endif
      }
      asm volatile("" ::: "memory");
    };
    auto checkpoint = [&](int kk) {
      // (two stores of what is in registers as it is: a 16-byte store would want the three words moved together first)
      float4& c = S.ck[s0][lane];
      *reinterpret_cast<v2f*>(&c) = P;
      c.z = __int_as_float(e);
      lds_post(&S.ckpos, kk + 1);
    };
    // The factors of ONE block in registers (eight entries of two frames): an entry is refilled with the next block's
    // right behind the two frames that consumed it -- the reads go out BETWEEN the frames, one per two frames (issued
    // back to back, as they were, they cost the wave ~16 cycles each and the sixteen frames waited behind them), and
    // each has seven entries' worth of frames to arrive.  The frames are volatile asm statements that clobber memory:
    // the compiler keeps the reads where they are written.
    mv4f f[kBlk / 2];
    auto frames2 = [&](const mv4f& q) {
      const v2f F0 = {q.x, q.y}, F1 = {q.z, q.w};
      asm volatile(WFL_FRAME("v[2:3]", "v3", "v[4:5]", "v4", "v5", "%[F0]", "%[Y0]")
                   WFL_FRAME("v[4:5]", "v5", "v[2:3]", "v2", "v3", "%[F1]", "%[Y1]")
                   : "+{v[2:3]}"(P)
                   : [G] "v"(G), [F0] "v"(F0), [F1] "v"(F1), [Y0] "v"(q.y), [Y1] "v"(q.w)
                   : "v4", "v5", "v6", "v7", "memory");
    };
    // One block.  FULL: renormalise with the clamp; LEAN: a complete block with a block behind it.
    auto block = [&](int kk, int& sf, auto full, auto lean) {
      constexpr bool FULL = decltype(full)::value, LEAN = decltype(lean)::value;
      staged_next(kk, sf);
#if WFL_MITM_STATS
      // (three stamps per sweep only: reading the clock waits for every LDS read in flight)
      if (lane == 0 && (kk == 4 || kk == NB / 2 || kk == NB - 2)) S.blk_t[kk == 4 ? 0 : kk == NB / 2 ? 1 : 2] = clock64() - st_begin;
#endif
      if (FULL) renorm_full(); else renorm_own();
      checkpoint(kk);
      if (LEAN) {
        sf = lds_peek(&S.staged[s2]);  // (looked at a block ahead of its use: consumed behind this block's frames)
#pragma unroll
        for (int j = 0; j < kBlk / 2; ++j) {
          frames2(f[j]);
          f[j] = S.ring[s1][j][lane];
        }
        lds_post(&S.chainpos, kk + 2);  // (behind the reads: a wave's LDS operations execute in order)
        // (its compare would otherwise be scheduled in front of the frames, with a wait for the flag's round trip there)
        asm volatile("" : "+v"(sf));
      } else {
        const int k = dir == 0 ? kk : NB - 1 - kk;
        const int n = min(kBlk, T - k * kBlk);
        mv4f c[kBlk / 2];
#pragma unroll
        for (int j = 0; j < kBlk / 2; ++j) c[j] = f[j];
        if (kk + 1 < NB) {
#pragma unroll
          for (int j = 0; j < kBlk / 2; ++j) f[j] = S.ring[s1][j][lane];
        }
        lds_post(&S.chainpos, kk + 2);
        if (kk + 2 < NB) sf = lds_peek(&S.staged[s2]);
#pragma unroll
        for (int j = 0; j < kBlk; ++j)
          if (j < n) frame((j & 1) ? v2f{c[j >> 1].z, c[j >> 1].w} : v2f{c[j >> 1].x, c[j >> 1].y});
      }
      s0 = s1, s1 = s2, s2 = s2 + 1 == K::kSlots ? 0 : s2 + 1;
    };
    {
      int spin = 0;
      MITM_T0();
      while (lds_peek(&S.staged[0]) != 1)
        if (mitm_waited_out(a, S, ++spin)) give_up();
      MITM_ACC(st_wait0);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < kBlk / 2; ++j) f[j] = S.ring[0][j][lane];
    lds_post(&S.chainpos, 1);
    int sf = NB > 1 ? lds_peek(&S.staged[1 % K::kSlots]) : 0;
    {
      const std::true_type yes{};
      const std::false_type no{};
      int kk = 0;
      block(0, sf, yes, no);
      kk = 1;
#if WFL_MITM_LEAN
      // blocks 1 .. NB - 2 are complete in both directions and have a block behind them.  The clamp in every block of the
      // start (lanes fill up one per frame at most), then in every fourth.
      for (; kk <= 4 && kk + 1 < NB; ++kk) block(kk, sf, yes, yes);
      if (kk == 5)
        for (; kk + 4 < NB; kk += 4) {
#if WFL_MITM_CLAMP_EVERY == 1
          block(kk, sf, yes, yes);
          block(kk + 1, sf, yes, yes);
          block(kk + 2, sf, yes, yes);
#else
          block(kk, sf, no, yes);
          block(kk + 1, sf, no, yes);
          block(kk + 2, sf, no, yes);
#endif
          block(kk + 3, sf, yes, yes);
        }
#endif
      for (; kk < NB; ++kk) block(kk, sf, yes, no);
    }
    // the state behind the last frame
    float pb = P.x, pl = P.y;
    auto lane_renorm = [&]() {
      const float mx = vmax(pb, pl);
      const int k = __builtin_amdgcn_frexp_expf(mx);  // 0 for mx == 0
      const int own = mx > 0.f ? e + k : kEmptyE;
      const int pre = wave_prefix_max_i(own + kGap * lane) - kGap * lane;
      const int shf = e - pre;
      pb = ldexpf(pb, shf), pl = ldexpf(pl, shf);
      e = pre;
    };
    bad = !(fabsf(pb) < 3.0e38f) || !(fabsf(pl) < 3.0e38f);
    lane_renorm();
    {
      int spin = 0;
      MITM_T0();
      while (lds_peek(&S.offdone) != 1) {
        __builtin_amdgcn_s_sleep(1);
        if (mitm_waited_out(a, S, ++spin)) give_up();
      }
      MITM_ACC(st_wait2);
    }
    asm volatile("" ::: "memory");
    const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;  // (any lane)
    if (lane == 0) ((int32_t*)(a.ws + w.pbad))[b * 2 + dir] = any_bad ? 1 : 0;
    // ONE word for the launch behind this one: raised (to this launch's token: nothing to zero) by whoever finds a
    // reason to doubt an utterance -- a chain that overflowed (here), a block whose posteriors do not sum to one (the
    // emitters), the alpha chain comparing its log2 Z with what the first emitted blocks of both sweeps reproduced
    // (below).  The repair launch evaluates the certificates in full only if it is up (ctc_repair_kernel).
    unsigned long long* suspect = (unsigned long long*)(a.ws + w.suspect);
    if (any_bad && lane == 0) coherent_store64(suspect, a.token);
    const bool cannot_align = T < L + __builtin_popcountll(__builtin_amdgcn_ballot_w64(has_label && lane >= 1 && y == yprev));
    if (dir == 0) {
      // Z = alpha_{T-1}[2L] + alpha_{T-1}[2L-1]   (ctc.py:21 accept states), lanes L and L-1
      const float zb = readlane_f(pb, L);
      const float zl = L > 0 ? readlane_f(pl, L - 1) : 0.f;
      const int eb = __builtin_amdgcn_readlane(e, L);
      const int el = L > 0 ? __builtin_amdgcn_readlane(e, L - 1) : kEmptyE;
      if (lane == 0) {
        const int em = max(zb > 0.f ? eb : kEmptyE, zl > 0.f ? el : kEmptyE);
        const float s = (zb > 0.f ? ldexpf(zb, max(eb - em, -200)) : 0.f) + (zl > 0.f ? ldexpf(zl, max(el - em, -200)) : 0.f);
        const bool ok = s > 0.f && s < 3.0e38f;
        const double z2 = ok ? (double)__builtin_amdgcn_logf(s) + (double)em + S.offtot : -1.0e300;
        ((double*)(a.ws + w.z2))[b] = z2;
        publish_nll<true, false>(a, w, b, ok, z2);
        // utterance_rejected()'s comparison, here and now.  The two first emitted blocks folded their log2 Z into zloc
        // half a sweep ago; should one of them not have arrived yet (its workgroup started late: nothing orders
        // workgroups) the count says so and the doubt is raised for the repair launch to settle.
        const long long* zmm = (const long long*)(a.ws + w.zloc) + (int64_t)b * 2;
        // (the count FIRST, and waited for: a count of nfirst then vouches for the two words read behind it -- read in the
        // other order, a partner's three updates could land between the loads and leave the initial range unseen)
        const int cnt = __hip_atomic_load((const int32_t*)(a.ws + w.zcnt) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(cnt) : "memory");  // (it has ARRIVED before the two loads below are issued)
        const long long lo = (long long)coherent_load64(zmm), hi = (long long)coherent_load64(zmm + 1);
        const long long zq = z_fixed(z2);
        constexpr long long tol = 10;  // (utterance_rejected)
        const int nfirst = (H0 < NB ? 1 : 0) + (mitm_first_emitted(NB, 1) < NB ? 1 : 0);  // sweeps that emit at all
        const bool doubt = zq == kZDead ? !cannot_align : (lo < zq - tol || hi > zq + tol || cnt != nfirst);
        if (doubt) coherent_store64(suspect, a.token);
      }
      stats_out();
#if WFL_MITM_STATS
      for (int q = lane; q < min(NB, 64); q += 64) ((long long*)(a.ws + w.dbg) + 2 * 8 * K::kWaves * (int64_t)a.B)[(int64_t)(b * 2 + dir) * 256 + q] = S.blk_t[q];
#endif
      if (a.loss_out && b == 0) reduce_loss_when_done(a, w, lane, false);
      return;
    }
    stats_out();
#if WFL_MITM_STATS
    for (int q = lane; q < min(NB, 64); q += 64) ((long long*)(a.ws + w.dbg) + 2 * 8 * K::kWaves * (int64_t)a.B)[(int64_t)(b * 2 + dir) * 256 + q] = S.blk_t[q];
#endif
    return;
  }

  // ================================================================ emitters
  if (dir == 0)
    ctc_mitm_emitter<K, LSM, 0, WIDE>(a, S, b, ridx, H0, NB, L, y, skip, skipn, has_label, coef, gout, dx, smem, w);
  else
    ctc_mitm_emitter<K, LSM, 1, WIDE>(a, S, b, ridx, H0, NB, L, y, skip, skipn, has_label, coef, gout, dx, smem, w);
}

template <class K, bool LSM, bool WIDE = false>
__global__ void __launch_bounds__(K::kWaves * 64, 4)  // (second argument: waves per SIMD -- at most 128 VGPRs, so that two
                                                     // 8-wave workgroups share a CU)
    ctc_mitm_kernel(CtcArgs a, const float* __restrict__ coef, const float* __restrict__ gout, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int32_t* perr = (int32_t*)(a.ws + ctc_ws_layout(a.B, a.T, a.P).perr);
    // (write-through, like the doubt word below: a give-up is an atomic from another XCD)
    __hip_atomic_store(perr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // a wave gave up waiting
    __hip_atomic_store(perr + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // utterances the repair launch recomputed
    // the doubt word: down at every launch (a write-through store by the first workgroup, half a sweep before anybody can
    // raise it) -- a launch replayed from a graph carries the SAME token, and a doubt of an earlier replay would otherwise
    // stand for ever
    coherent_store64(a.ws + ctc_ws_layout(a.B, a.T, a.P).suspect, 0ull);
  }
  // the launch's own clock: entry of the workgroup, exit of its last wave (two words per sweep; nobody waits for them)
  unsigned long long* clk = (unsigned long long*)(a.ws + ctc_ws_layout(a.B, a.T, a.P).clk) + (int64_t)blockIdx.x * 2;
  if (threadIdx.x == 0) clk[0] = wall_clock64(), clk[1] = 0ull;  // (in front of the body's barrier: no wave leaves before it)
  ctc_mitm_body<K, LSM, WIDE>(a, (int)blockIdx.x >> 1, (int)blockIdx.x & 1, coef, gout, dx, smem);
  // (measured: these sixteen atomics move ~1.2 us from the launch behind this one into this one -- the two together take
  // what they took without them)
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_max(clk + 1, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

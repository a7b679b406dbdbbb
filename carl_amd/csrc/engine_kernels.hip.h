// engine_kernels.hip.h -- the generic batched reset / step / rollout kernels.
//
// One lane (thread) = one env instance.  All traffic is coalesced: SoA state
// columns, feature-major context rows, lane-major observation records written with
// one wide store per lane.  The arithmetic intensity is a few flop/byte, so these
// kernels are HBM- (large N) or latency- (N ~ 65 536: one wavefront per SIMD) bound,
// never MFMA material; the levers are bytes moved per step, the length of a single
// wave's instruction stream, and never making that wave wait on its own stores.
#pragma once

#include <type_traits>

#include "carl_device.hip.h"

// (Rounds 1-3 carried ~30 CARL_EXP_* / CARL_TRY_* profiling switches in these loops -- kernels that skip stores, loads
// or the done path, for cost attribution.  Their measurements are recorded under profiles/ and in DESIGN.md's
// appendix; the switches themselves were removed in round 4 so that the product loops read straight.  The tree that
// builds them is commit 3a3c7b2 (tools/build_ablations.sh there).)

namespace carl {

template <bool LDS>
using ctx_t = std::conditional_t<LDS, LdsCtx, GlobalCtx>;

// Families whose physics reads a constant table from LDS (Acrobot's fp64 sin/cos grid, classic_control.hip.h)
// stage it at the top of every kernel, before the first `prepare` / `step`.
template <class Fam, class = void>
struct has_tables : std::false_type {};
template <class Fam>
struct has_tables<Fam, std::void_t<decltype(Fam::kUsesSinCosTab)>> : std::bool_constant<Fam::kUsesSinCosTab> {};

template <class Fam>
__device__ __forceinline__ void stage_family_tables() {
  if constexpr (has_tables<Fam>::value) {
    Fam::stage_tables();
    __syncthreads();
  }
}

template <bool LDS, int F>
__device__ __forceinline__ ctx_t<LDS> make_ctx(const carl_batch_t& b, float* lds) {
  if constexpr (LDS) {
    stage_ctx_table<F>(lds, b);
    return LdsCtx{lds, b.n_contexts};
  } else {
    return GlobalCtx{b.ctx_table, b.ctx_stride};
  }
}

// Reset of one lane: selector advance -> init-state draw -> context observation.
// carl/envs/carl_env.py:245-274 + the family's reset override.
// `p` holds the parameters of context `cidx` on entry; they are re-gathered (and the lane's
// context observation rewritten) only when the selector moves the lane to another context
// (`force`: first reset).  With a static selector a reset therefore reads no memory at all.
// Returns true when memory was read.
// `pre`: the episode's init-state words drawn ahead of time (staged rollout, families with kPredraw);
// nullptr: drawn here.
// CTX_OBS = false: the caller rewrites the lane's context observation itself (the step / rollout kernels
// do it once per launch, in store_lane, instead of on every reset).
template <class Fam, class Ctx, bool CTX_OBS = true>
__device__ __forceinline__ bool reset_lane(const carl_batch_t& b, const Ctx& ctx, int lane, uint64_t glane,
                                           int& cidx, uint32_t& episode, typename Fam::Params& p,
                                           float (&s)[Fam::S], bool force, bool valid = true,
                                           const u32x4* pre = nullptr) {
  const int old = cidx;
  cidx = select_context(b, cidx, glane, episode);
  const bool changed = force || (cidx != old);
  if (changed) {
    p = Fam::load(ctx, cidx, b.flags);
    // the gathered parameters must have ARRIVED before this branch rejoins: otherwise the
    // compiler parks the matching `s_waitcnt vmcnt(0)` at the join, where EVERY reset pays it --
    // and with loads and stores in one in-order vmcnt queue that wait also drains the episode
    // statistics just stored (a full store round trip per reset; CartPole under a random policy
    // resets some lane of nearly every wave on nearly every step: 983 -> 852 ns/step, A/B on one box)
    settle(p);
    if constexpr (CTX_OBS) {
      if (b.ctx_obs != nullptr && valid) {
        for (int k = 0; k < b.n_ctx_obs; ++k)
          b.ctx_obs[(size_t)k * b.n_lanes + lane] = ctx.get(b.ctx_obs_feat[k], cidx);
      }
    }
  }
  const u32x4 w = (pre != nullptr) ? *pre : lane_words(b.seed, glane, episode, kSubInit);
  Fam::reset(p, w, s);
  episode += 1u;
  return changed;
}

// -------------------------------- reset -------------------------------------------
// mask == nullptr && idx == nullptr : every lane
// mask != nullptr                   : lanes with mask[lane] != 0
// idx  != nullptr                   : lanes idx[0 .. *count)  (compacted done list)
template <class Fam, bool LDS>
__global__ void __launch_bounds__(256) reset_kernel(const carl_batch_t b, const uint8_t* __restrict__ mask,
                                                    const int32_t* __restrict__ idx,
                                                    const int32_t* __restrict__ count, float* __restrict__ obs) {
  extern __shared__ float lds_ctx[];
  stage_family_tables<Fam>();
  const ctx_t<LDS> ctx = make_ctx<LDS, Fam::F>(b, lds_ctx);
  const int n_work = (idx != nullptr) ? *count : b.n_lanes;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_work; k += gridDim.x * blockDim.x) {
    const int lane = (idx != nullptr) ? idx[k] : k;
    if (mask != nullptr && mask[lane] == 0) continue;
    const uint64_t glane = (uint64_t)(b.lane_offset + lane);
    int cidx = b.ctx_idx[lane];
    uint32_t episode = b.episode[lane];
    float s[Fam::S];
    typename Fam::Params p;
    reset_lane<Fam>(b, ctx, lane, glane, cidx, episode, p, s, true);
#pragma unroll
    for (int j = 0; j < Fam::S; ++j) b.state[(size_t)j * b.n_lanes + lane] = s[j];
    b.elapsed[lane] = 0;
    b.ep_return[lane] = 0.0f;
    b.ctx_idx[lane] = cidx;
    b.episode[lane] = episode;
    b.n_calls[lane] += 1;
    if (obs != nullptr) {
      float o[Fam::D];
      typename Fam::Aux aux;
      Fam::prepare(s, aux);
      Fam::observe(s, aux, o);
      store_obs<Fam::D>(obs, (size_t)lane, o);
    }
  }
}

// ---------------------------- per-lane step body ------------------------------------
// Everything a lane carries between steps lives in registers.
template <class Fam>
struct LaneRegs {
  float s[Fam::S];
  float ep_return;
  int elapsed;
  int cidx;
  uint32_t episode;     // lazily loaded: only the reset path (and step noise) reads it
  bool episode_valid;
  int n_new_calls;      // resets performed in this launch
  int n_new_episodes;   // episodes finished in this launch
  float fin_return;     // return / length of the last of them (-> last_return / last_length at the end of
  int fin_length;       // the launch: two global stores less on the done path of every finished episode)
  bool valid;           // false: a padding lane of a ragged last workgroup (a register-only clone of
                        // the batch's last lane: it computes, but nothing it does reaches global memory)
  u32x4 next_w;         // kPredraw families in the staged rollout: the init-state words of the lane's NEXT
  bool next_ok;         // episode, drawn once per chunk for the whole wave (see predraw)
  typename Fam::Params p;
  typename Fam::Aux aux;  // derived from s: shared by this step's obs and the next step
};

// Per-lane output cursors, advanced by one step's stride after every step.  Keeping them
// as lane-private addresses (VGPRs) keeps kernarg reloads (s_load + lgkmcnt waits) out
// of the step loop.
template <class Fam>
struct Cursors {
  float* obs;        // += n * D
  float* reward;     // += n
  uint8_t* term;     // += n
  uint8_t* trunc;    // += n
  float* final_obs;  // += n * D (nullable)
  uint8_t* done;     // per-call step only (nullable): terminated | truncated
  __device__ __forceinline__ void advance(size_t n) {
    obs += n * Fam::D;
    reward += n;
    term += n;
    trunc += n;
    if (final_obs != nullptr) final_obs += n * Fam::D;
  }
  // output-sink interface of step_lane
  static constexpr bool kLazyFlags = false;  // flags are written on every step
  __device__ __forceinline__ void put_reward(float r) const { *reward = r; }
  __device__ __forceinline__ void put_flags(bool te, bool tr) const {
    *term = (uint8_t)te;
    *trunc = (uint8_t)tr;
    if (done != nullptr) *done = (uint8_t)(te | tr);
  }
  __device__ __forceinline__ void put_obs(const float (&o)[Fam::D]) const { store_obs<Fam::D>(obs, 0, o); }
  __device__ __forceinline__ float* final_obs_ptr() const { return final_obs; }
};

// LDS output sink of the staged rollout: a step's records of one workgroup, laid out exactly
// like the workgroup's slice of the HBM arrays ([256][D] obs | [256] reward | [256] term |
// [256] trunc), so the storer wave can drain them with 16-byte stores.
template <class Fam>
struct LdsSink {
  static constexpr int kObsBytes = 256 * Fam::D * 4;
  static constexpr int kFlagOff = kObsBytes + 1024;  // [256] term | [256] trunc
  static constexpr int kStepBytes = kObsBytes + 256 * 4 + 256 + 256;
  // The flag rows are all-zero except on steps where some lane of the wave finishes an episode:
  // the storer re-zeroes them after draining, and step_lane writes them only on the (wave-
  // uniform) done path -- two LDS writes and their operands less in every ordinary step.
  static constexpr bool kLazyFlags = true;
  char* step_base;    // this step's record block in LDS
  float* final_base;  // this lane's terminal-observation slot of step 0 in HBM, nullable
  size_t n_obs;       // n_lanes * D: elements per step of the observation arrays
  int t;              // global step index of this record
  int tid;
  __device__ __forceinline__ void put_reward(float r) const {
    reinterpret_cast<float*>(step_base + kObsBytes)[tid] = r;
  }
  __device__ __forceinline__ void put_flags(bool te, bool tr) const {
    reinterpret_cast<uint8_t*>(step_base + kFlagOff)[tid] = (uint8_t)te;
    reinterpret_cast<uint8_t*>(step_base + kFlagOff + 256)[tid] = (uint8_t)tr;
  }
  __device__ __forceinline__ void put_obs(const float (&o)[Fam::D]) const {
    store_obs<Fam::D>(reinterpret_cast<float*>(step_base), (size_t)tid, o);
  }
  // only evaluated on the done path (the terminal observation is stored to HBM directly)
  __device__ __forceinline__ float* final_obs_ptr() const {
    return final_base != nullptr ? final_base + (size_t)t * n_obs : nullptr;
  }
};

// Families whose episodes are short under any policy (CartPole: ~22 steps under a random one) finish an
// episode in some lane of nearly every wave on nearly every step, so the wave-uniform done path runs
// almost every step and its largest piece -- the Philox4x32-10 block of the init-state draw, ~110
// instructions -- is issued for 64 lanes to serve one or two.  The words only depend on (seed, lane,
// episode number), not on the trajectory: the staged rollout draws them for every lane of the wave ONCE
// PER CHUNK, ahead of time, and a finishing lane just consumes its pending words.  A lane that finishes
// twice inside one chunk draws inline (rare).  Same words, same order: results are bit-identical.
template <class Fam, class = void>
struct predraw_of : std::false_type {};
template <class Fam>
struct predraw_of<Fam, std::void_t<decltype(Fam::kPredraw)>> : std::bool_constant<Fam::kPredraw> {};

template <class Fam>
__device__ __forceinline__ void predraw(const carl_batch_t& b, uint64_t glane, LaneRegs<Fam>& r) {
  if (ballot(!r.next_ok) != 0ull) {
    const u32x4 w = lane_words(b.seed, glane, r.episode, kSubInit);
    if (!r.next_ok) r.next_w = w;
    r.next_ok = true;
  }
}

// The rarely-taken part of a step, entered only by wavefronts in which some lane just
// finished an episode (wave-uniform branch on a ballot): episode statistics, the compact
// finished-episode log (ballot + one atomic per wave), and the in-kernel auto-reset
// (Philox draws, selector advance, context re-gather).
template <class Fam, class Ctx, bool PRE = false>
__device__ __forceinline__ void finish_episodes(const carl_batch_t& b, const Ctx& ctx, bool done, int lane,
                                                uint64_t glane, float* final_obs, float (&o)[Fam::D],
                                                LaneRegs<Fam>& r) {
  const float fin_ret = r.ep_return;
  const int fin_len = r.elapsed;
  if (done && r.valid) {  // the lane's last finished episode: kept in registers, written once by store_lane
    r.fin_return = fin_ret;
    r.fin_length = fin_len;
    r.n_new_episodes += 1;
  }
  log_finished(b, done && r.valid, glane, fin_ret, fin_len);
  if ((b.flags & CARL_FLAG_AUTORESET) && done) {
    if (final_obs != nullptr) store_obs<Fam::D>(final_obs, 0, o);
    if (!r.episode_valid) {  // (never on a padding lane: the rollout kernels preload the counter)
      r.episode = b.episode[lane];
      r.episode_valid = true;
      settle(r.episode);  // wait here, inside the branch (see reset_lane)
    }
    if constexpr (PRE) {
      u32x4 w = r.next_w;
      if (!r.next_ok) w = lane_words(b.seed, glane, r.episode, kSubInit);  // second finish inside one chunk
      r.next_ok = false;
      reset_lane<Fam, Ctx, false>(b, ctx, lane, glane, r.cidx, r.episode, r.p, r.s, false, r.valid, &w);
    } else {
      reset_lane<Fam, Ctx, false>(b, ctx, lane, glane, r.cidx, r.episode, r.p, r.s, false, r.valid);
    }
    r.elapsed = 0;
    r.ep_return = 0.0f;
    r.n_new_calls += 1;
    Fam::prepare(r.s, r.aux);
    Fam::observe(r.s, r.aux, o);
  }
  // ... and every scalar (kernarg) load too: LDS operations retire in order but scalar loads do
  // not, so a possibly-outstanding s_load anywhere on this path would turn every LDS wait of the
  // step loop into lgkmcnt(0) (= also wait for the record writes just issued).  The compiler
  // models an explicit s_waitcnt: vmcnt/expcnt untouched, lgkmcnt(0).
  __builtin_amdgcn_s_waitcnt(0xC07F);
}

// word-wise `c ? a : b` of any trivially copyable register-resident value (v_cndmask per 32-bit word)
template <class T>
__device__ __forceinline__ T select_words(bool c, const T& a, const T& b) {
  if constexpr (sizeof(T) < 4) {
    return c ? a : b;
  } else {
    static_assert(sizeof(T) % 4 == 0, "select_words() works on 32-bit words");
    uint32_t wa[sizeof(T) / 4], wb[sizeof(T) / 4];
    __builtin_memcpy(wa, &a, sizeof(T));
    __builtin_memcpy(wb, &b, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; ++k) wa[k] = c ? wa[k] : wb[k];
    T out;
    __builtin_memcpy(&out, wa, sizeof(T));
    return out;
  }
}

// PLAIN (staged rollout of a kPredraw family, chosen by the host when lanes keep their contexts -- static /
// host selector -- and neither the finished-episode log nor terminal observations are asked for): the done
// path of the common fused-rollout configuration without the code of the optional features.  Those cost
// even when a wave-uniform branch skips them -- scalar registers spilled to VGPR lanes and restored,
// exec-mask bookkeeping, waits at the joins: CartPole, 65 536 lanes, 768 -> 553 ns/step.
// It is written as straight-line selects: entered under the wave-uniform
// ballot branch, it has no divergent control flow of its own (the one inner branch -- a lane finishing
// twice inside a chunk -- is wave-uniform as well), so the compiler has one join to reconcile instead of
// five and the exec mask is never rewritten.  With `if (done) { ... }` blocks the same work cost ~30
// register copies at the joins on top of its ~35 instructions.
template <class Fam>
__device__ __forceinline__ void finish_plain(const carl_batch_t& b, uint64_t glane, bool done, float (&o)[Fam::D],
                                             LaneRegs<Fam>& r) {
  u32x4 w = r.next_w;
  if (ballot(done && !r.next_ok) != 0ull) {
    const u32x4 wi = lane_words(b.seed, glane, r.episode, kSubInit);
    w = select_words(r.next_ok, w, wi);
  }
  float ns[Fam::S], no[Fam::D];
  typename Fam::Aux na;
  Fam::reset(r.p, w, ns);
  Fam::prepare(ns, na);
  Fam::observe(ns, na, no);
  const bool rs = done && (b.flags & CARL_FLAG_AUTORESET) != 0;
  r.fin_return = done ? r.ep_return : r.fin_return;
  r.fin_length = done ? r.elapsed : r.fin_length;
  r.n_new_episodes += done ? 1 : 0;
#pragma unroll
  for (int j = 0; j < Fam::S; ++j) r.s[j] = rs ? ns[j] : r.s[j];
#pragma unroll
  for (int d = 0; d < Fam::D; ++d) o[d] = rs ? no[d] : o[d];
  r.aux = select_words(rs, na, r.aux);
  r.elapsed = rs ? 0 : r.elapsed;
  r.ep_return = rs ? 0.0f : r.ep_return;
  r.episode += rs ? 1u : 0u;
  r.n_new_calls += rs ? 1 : 0;
  r.next_ok = r.next_ok && !rs;
  __builtin_amdgcn_s_waitcnt(0xC07F);  // as at the end of finish_episodes
}

// DENSE done handling (families with kDenseDone, in the PLAIN staged rollout): CartPole under a random
// policy ends an episode every ~22 steps, so SOME lane of nearly every wave finishes on nearly every step
// (P(no lane of 64 finishes) ~ 5 %): the "rarely taken" wave-uniform done branch of step_lane is the common
// path there, and what it costs is not its ~45 instructions but what surrounds them with one compute wave
// per SIMD and nothing to overlap: five taken branches per step, the VALU -> SALU -> branch hand-overs, an
// lgkmcnt(0) at its end (r01j: 104 VALU + 23 SALU in ~1000 cycles per wave-step, against a 298 ns memory
// floor).  Here the done handling is STRAIGHT-LINE for every step:
//   * once per 8-step chunk the wave draws the init-state words of every lane whose pending draw was consumed
//     (same Philox block as predraw) and turns them into the lane's next reset STATE (and its Aux) -- the
//     uniform conversion and the fma leave the per-step path;
//   * a step then ends with S + |Aux| + 6 selects on `done` (v_cndmask on the compare's own lane mask): no
//     ballot branch, no exec-mask change, no wait;
//   * "pending draw valid" is a wave-uniform 64-bit mask in SGPRs, maintained with scalar ops from the done
//     ballot; the only branch left is the never-taken one for a lane that finishes a second time before the
//     next chunk's draw (drawn inline: same words, same order -> bit-identical to the per-call kernel).
template <class Fam, class = void>
struct deep_action_prefetch_of : std::false_type {};
template <class Fam>
struct deep_action_prefetch_of<Fam, std::void_t<decltype(Fam::kDeepActionPrefetch)>> : std::bool_constant<Fam::kDeepActionPrefetch> {};

// lean int32 / float32 launches of fewer lanes than this keep two chunks of actions in flight (rollout_staged_body)
template <class Fam, class = void>
struct deep_below_lanes_of : std::integral_constant<int, 0> {};
template <class Fam>
struct deep_below_lanes_of<Fam, std::void_t<decltype(Fam::kDeepBelowLanes)>> : std::integral_constant<int, Fam::kDeepBelowLanes> {};

template <class Fam, class = void>
struct dense_done_of : std::false_type {};
template <class Fam>
struct dense_done_of<Fam, std::void_t<decltype(Fam::kDenseDone)>> : std::bool_constant<Fam::kDenseDone> {};

template <class Fam>
struct DenseNext {
  unsigned long long ok_mask;  // bit l: lane l's fields below hold its NEXT episode's init state
  float s[Fam::S];
  typename Fam::Aux aux;
  // MOVES (round-robin / random selector: a reset moves the lane to another context -- the reference's DEFAULT
  // selector is round robin): the next episode's context id and its gathered parameters, prepared with the draw
  int cidx;
  typename Fam::Params p;
};

// MOVES = false: lanes keep their contexts (static / host selector).  MOVES = true: the selector rule is applied
// here, once per chunk and for the whole wave, and the next context's parameters are gathered ahead of time (from
// the LDS copy of the table when it fits, else from HBM -- once per chunk, off the step path); a step then ends
// with |Params| + 1 more selects.  Round 1 ran this configuration through the branchy done path: CartPole under a
// round-robin selector 951 ns/step against 505 under a static one.
template <class Fam, class Ctx, bool MOVES>
__device__ __forceinline__ void dense_draw(const carl_batch_t& b, const Ctx& ctx, uint64_t glane,
                                           const LaneRegs<Fam>& r, unsigned long long lanes, DenseNext<Fam>& nx) {
  int cn = r.cidx;
  typename Fam::Params pn = r.p;
  if constexpr (MOVES) {
    cn = select_context(b, r.cidx, glane, r.episode);  // carl/context/selection.py rules, as reset_lane applies them
    if (ballot(cn != r.cidx) != 0ull) {
      typename Fam::Params q = Fam::load(ctx, cn, b.flags);
      settle(q);
      pn = select_words(cn != r.cidx, q, r.p);
    }
  }
  const u32x4 w = lane_words(b.seed, glane, r.episode, kSubInit);
  float fresh[Fam::S];
  typename Fam::Aux fa;
  Fam::reset(pn, w, fresh);
  Fam::prepare(fresh, fa);
  const bool mine = ((lanes >> lane_id()) & 1ull) != 0ull;
#pragma unroll
  for (int j = 0; j < Fam::S; ++j) nx.s[j] = mine ? fresh[j] : nx.s[j];
  nx.aux = select_words(mine, fa, nx.aux);
  if constexpr (MOVES) {
    nx.cidx = mine ? cn : nx.cidx;
    nx.p = select_words(mine, pn, nx.p);
  }
}

// FINAL: terminal observations requested (io.final_obs): a finishing lane stores its pre-reset observation straight
// to HBM (one exec-masked store; compute waves issue no loads, so nothing queues behind it).
// AR: the launch is known to run with CARL_FLAG_AUTORESET (the host picks the specialisation): `done` IS the reset
// mask, and -- `ENTRY` false: any step but the launch's first -- the step function is told elapsed = 0.  Of the step
// functions only CartPole's reads `elapsed`, to recognise a lane that is stepped AGAIN after terminating (reward 0,
// gymnasium's steps_beyond_terminated): `elapsed > 0 && out_of_bounds(pre-step state)`.  Under auto-reset a lane
// that went out of bounds was reset in that same step, so past the launch's first step (whose loaded state may
// date from a launch without auto-reset) the pre-step state is either fresh (elapsed 0) or in bounds: the rule
// never fires, and with the constant the compiler drops its seven instructions and the reward select.
// `max_steps_eff`: max_episode_steps, or INT_MAX for "no TimeLimit" (one compare instead of compare + mask AND).
template <class Fam, class Ctx, bool MOVES, bool FINAL, bool AR, bool ENTRY, class Sink>
__device__ __forceinline__ void step_dense(const carl_batch_t& b, const Ctx& ctx, const Sink& cur, int max_steps_eff,
                                           bool autoreset_in, uint64_t glane, typename Fam::Action action,
                                           LaneRegs<Fam>& r, DenseNext<Fam>& nx) {
  static_assert(!Fam::kNeedsStepNoise, "dense done handling: families without per-step noise");
  const bool autoreset = AR || autoreset_in;
  float reward;
  const bool terminated = Fam::step(r.p, r.s, r.aux, action, 0.0f, (AR && !ENTRY) ? 0 : r.elapsed, reward);
  r.elapsed += 1;
  const bool truncated = r.elapsed >= max_steps_eff;  // gymnasium TimeLimit.step
  r.ep_return += reward;
  cur.put_reward(reward);
  cur.put_flags(terminated, truncated);  // every step (the lazy flag rows only save work when done is rare)
  const bool done = terminated | truncated;
  const unsigned long long dm = ballot(done);
  const unsigned long long again = dm & ~nx.ok_mask;
  if (__builtin_expect(again != 0ull, 0)) dense_draw<Fam, Ctx, MOVES>(b, ctx, glane, r, again, nx);
  const bool rs = done && autoreset;
  if constexpr (FINAL) {
    if (rs && r.valid) {  // (as finish_episodes: with auto-reset, the done lanes of the batch)
      float fo[Fam::D];
      Fam::observe(r.s, r.aux, fo);
      store_obs<Fam::D>(cur.final_obs_ptr(), 0, fo);
    }
  }
  r.fin_return = done ? r.ep_return : r.fin_return;
  r.fin_length = done ? r.elapsed : r.fin_length;
  r.n_new_episodes += done ? 1 : 0;
#pragma unroll
  for (int j = 0; j < Fam::S; ++j) r.s[j] = rs ? nx.s[j] : r.s[j];
  r.aux = select_words(rs, nx.aux, r.aux);
  if constexpr (MOVES) {
    r.cidx = rs ? nx.cidx : r.cidx;
    r.p = select_words(rs, nx.p, r.p);
  }
  r.elapsed = rs ? 0 : r.elapsed;
  r.ep_return = rs ? 0.0f : r.ep_return;
  r.episode += rs ? 1u : 0u;
  if constexpr (!AR) r.n_new_calls += rs ? 1 : 0;  // (AR: = n_new_episodes, taken from it after the last step)
  nx.ok_mask &= autoreset ? ~dm : ~0ull;
  float o[Fam::D];
  Fam::observe(r.s, r.aux, o);
  cur.put_obs(o);
}

// One step of one lane.  `cur` points at this step's output records for this lane.
// ALL_ACTIVE: the whole wave is inside the batch (every full workgroup), so the per-step
// `if (active)` exec-mask dance disappears from the loop.
template <class Fam, class Ctx, bool ALL_ACTIVE = false, class Sink = Cursors<Fam>, bool PLAIN = false>
__device__ __forceinline__ void step_lane(const carl_batch_t& b, const Ctx& ctx, const Sink& cur,
                                          int max_steps, bool active_in, int lane, uint64_t glane,
                                          typename Fam::Action action, LaneRegs<Fam>& r) {
  const bool active = ALL_ACTIVE || active_in;
  bool done = false;
  bool te = false, tr = false;
  float o[Fam::D];
  if (active) {
    float noise = 0.0f;
    if constexpr (Fam::kNeedsStepNoise) noise = Fam::step_noise(r.p, b, glane, r.episode - 1u, r.elapsed);
    float reward;
    const bool terminated = Fam::step(r.p, r.s, r.aux, action, noise, r.elapsed, reward);
    r.elapsed += 1;
    // gymnasium TimeLimit.step: truncated = elapsed >= max_episode_steps
    const bool truncated = (max_steps > 0) && (r.elapsed >= max_steps);
    r.ep_return += reward;
    Fam::observe(r.s, r.aux, o);
    cur.put_reward(reward);
    if constexpr (!Sink::kLazyFlags) cur.put_flags(terminated, truncated);
    te = terminated;
    tr = truncated;
    done = terminated | truncated;
  }
  if (__builtin_expect(ballot(done) != 0ull, 0)) {
    if constexpr (Sink::kLazyFlags) {  // the flag rows are pre-zeroed: only waves with a finished lane write
      if (active) cur.put_flags(te, tr);
    }
    if constexpr (PLAIN)
      finish_plain<Fam>(b, glane, done, o, r);
    else
      finish_episodes<Fam, Ctx, Sink::kLazyFlags && predraw_of<Fam>::value>(b, ctx, done, lane, glane,
                                                                            cur.final_obs_ptr(), o, r);
  }
  if (active) cur.put_obs(o);
}

template <class Fam, class Ctx>
__device__ __forceinline__ void load_lane(const carl_batch_t& b, const Ctx& ctx, int lane, LaneRegs<Fam>& r) {
#pragma unroll
  for (int j = 0; j < Fam::S; ++j) r.s[j] = b.state[(size_t)j * b.n_lanes + lane];
  r.elapsed = b.elapsed[lane];
  r.cidx = b.ctx_idx[lane];
  r.ep_return = b.ep_return[lane];
  r.episode_valid = Fam::kNeedsStepNoise;
  if constexpr (Fam::kNeedsStepNoise) r.episode = b.episode[lane];
  r.n_new_calls = 0;
  r.n_new_episodes = 0;
  r.valid = true;
  r.p = Fam::load(ctx, r.cidx, b.flags);
  Fam::prepare(r.s, r.aux);
  // have all of it in registers NOW (see finish_episodes): the first use must not leave
  // a vmcnt(0) wait inside the step loop
  settle(r.p);
  settle(r.s);
  settle(r.aux);
  settle(r.elapsed);
  settle(r.ep_return);
  settle(r.cidx);
  if constexpr (Fam::kNeedsStepNoise) settle(r.episode);
}

template <class Fam, class Ctx>
__device__ __forceinline__ void store_lane(const carl_batch_t& b, const Ctx& ctx, int lane, const LaneRegs<Fam>& r) {
#pragma unroll
  for (int j = 0; j < Fam::S; ++j) b.state[(size_t)j * b.n_lanes + lane] = r.s[j];
  b.elapsed[lane] = r.elapsed;
  b.ep_return[lane] = r.ep_return;
  if (r.n_new_episodes != 0) {
    if (b.episodes_done != nullptr) b.episodes_done[lane] += r.n_new_episodes;
    if (b.last_return != nullptr) b.last_return[lane] = r.fin_return;
    if (b.last_length != nullptr) b.last_length[lane] = r.fin_length;
  }
  if (r.n_new_calls != 0) {  // only lanes that were reset in this launch
    const bool moved = b.ctx_idx[lane] != r.cidx;  // (re-read here: no register is held for it across the steps)
    b.ctx_idx[lane] = r.cidx;
    b.episode[lane] = r.episode;
    b.n_calls[lane] += r.n_new_calls;
    // the lane's context observation: only its value at the end of the launch is visible, so it is
    // written here instead of on every context change.  (On the done path this loop -- per feature an
    // s_load of the feature id, a wait, the table read, a wait, the store -- cost 750 ns of a 1670 ns
    // CartPole step under a round-robin / random selector with the default eight observed features.)
    if (moved && b.ctx_obs != nullptr) {
      for (int k = 0; k < b.n_ctx_obs; ++k)
        b.ctx_obs[(size_t)k * b.n_lanes + lane] = ctx.get(b.ctx_obs_feat[k], r.cidx);
    }
  }
}

template <class Fam>
__device__ __forceinline__ Cursors<Fam> make_cursors(const carl_step_io_t& io, int lane, bool per_call = false) {
  Cursors<Fam> c;
  c.done = (per_call && io.done != nullptr) ? io.done + lane : nullptr;
  c.obs = io.obs + (size_t)lane * Fam::D;
  c.reward = io.reward + lane;
  c.term = io.terminated + lane;
  c.trunc = io.truncated + lane;
  c.final_obs = io.final_obs ? io.final_obs + (size_t)lane * Fam::D : nullptr;
  return c;
}

// action element type as stored by the caller: discrete families accept int32 (AK 0) or int64 (AK 1 -- the `A64 = true`
// of the kernels' template lists converts to it); the lean staged rollout also uint8 (AK 2 = kActU8: one byte per
// lane-step instead of four -- the action stream is the fused rollout's only per-step READ)
// ... and the Box families float16 / bfloat16 (kActF16 / kActBF16: two bytes per lane-step, widened exactly).
constexpr int kActU8 = 2, kActF16 = 3, kActBF16 = 4;
struct f16_bits {
  unsigned short v;
};
struct bf16_bits {
  unsigned short v;
};
template <class Fam, int AK>
using action_store_t =
    std::conditional_t<std::is_same_v<typename Fam::Action, float>,
                       std::conditional_t<AK == kActF16, f16_bits, std::conditional_t<AK == kActBF16, bf16_bits, float>>,
                       std::conditional_t<AK == 1, long long, std::conditional_t<AK == kActU8, unsigned char, int>>>;

// -------------------------------- step (per call) -----------------------------------
template <class Fam, bool LDS, bool A64>
__global__ void __launch_bounds__(256) step_kernel(const carl_batch_t b, const carl_step_io_t io) {
  extern __shared__ float lds_ctx[];
  stage_family_tables<Fam>();
  const ctx_t<LDS> ctx = make_ctx<LDS, Fam::F>(b, lds_ctx);
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = lane < b.n_lanes;
  const uint64_t glane = (uint64_t)(b.lane_offset + lane);
  LaneRegs<Fam> r{};
  typename Fam::Action action{};
  if (active) {
    load_lane<Fam>(b, ctx, lane, r);
    action = (typename Fam::Action) static_cast<const action_store_t<Fam, A64>*>(io.action)[lane];
  }
  const Cursors<Fam> cur = make_cursors<Fam>(io, active ? lane : 0, true);
  step_lane<Fam>(b, ctx, cur, b.max_episode_steps, active, lane, glane, action, r);
  if (active) store_lane<Fam>(b, ctx, lane, r);
}

// -------------------------------- rollout (T steps fused) ---------------------------
// State, context parameters and counters stay in registers for T steps; per step the
// lane reads one action and writes one full transition.
//
// Where the actions come from matters more than the arithmetic: on gfx9-family hardware
// a wave's loads and stores retire through ONE in-order counter (vmcnt), so an action
// load issued in the step loop drags a wait for every older store of that wave; with
// one wave per SIMD (65 536 lanes) that serialised each step behind the previous
// step's store round trip (~1500 cycles/step measured, profiles/r01a).  So the step
// loop of a compute wave issues NO global loads: each workgroup has one extra LOADER
// wave that streams the next chunk of actions HBM -> LDS (its vmcnt only ever covers
// its own loads) while the compute waves consume the current chunk from LDS (lgkmcnt)
// and only ever issue stores.  Double-buffered, one __syncthreads() per chunk.
constexpr int kRolloutLanes = 256;                 // compute lanes per workgroup
constexpr int kRolloutThreads = kRolloutLanes + kWave;  // + the loader wave
constexpr int kActChunk = 16;                      // steps per LDS buffer (2 x 16 KiB per workgroup)

__host__ __device__ constexpr size_t rollout_action_lds_bytes() {
  return (size_t)2 * kActChunk * kRolloutLanes * sizeof(float);
}

// loader wave: actions of steps [t0, t0 + kActChunk) for this workgroup's 256 lanes.
// A step's 256 actions are 1 KiB contiguous in HBM: one 16-byte load per loader lane.
template <class AStore, class Action, int CHUNK = kActChunk>
__device__ __forceinline__ void stage_actions(Action* buf, const AStore* __restrict__ act, size_t n, int lane_base,
                                              int t0, int n_steps) {
  const int l = threadIdx.x - kRolloutLanes;  // 0..63
  const int first = lane_base + 4 * l;
  if constexpr (std::is_same_v<AStore, Action>) {
    // wave-uniform fast path: whole workgroup in range, rows 16-byte aligned
    // and a full chunk
    if ((n % 4 == 0) && (lane_base + kRolloutLanes <= (int)n) && (t0 + CHUNK <= n_steps)) {
      using V = std::conditional_t<std::is_same_v<Action, float>, float4, int4>;
      V tmp[CHUNK];
      const V* src = reinterpret_cast<const V*>(act + (size_t)t0 * n + first);
      const size_t row_v = n / 4;
      // all loads of the chunk in flight together, then all LDS writes
#pragma unroll
      for (int u = 0; u < CHUNK; ++u) tmp[u] = src[(size_t)u * row_v];
#pragma unroll
      for (int u = 0; u < CHUNK; ++u) *reinterpret_cast<V*>(buf + u * kRolloutLanes + 4 * l) = tmp[u];
      return;
    }
  }
  // ragged workgroup / unaligned rows / ragged tail / int64 actions: element loads, predicated instead
  // of loop-bounded so that the loop unrolls completely -- every load of the chunk is in flight before
  // the first LDS write (a load-then-store loop paid one memory round trip per step and made the
  // ragged last workgroup 3x slower than the others, which is what the whole launch then waits for)
  Action tmp[CHUNK][4];
#pragma unroll
  for (int u = 0; u < CHUNK; ++u) {
    const AStore* row = act + (size_t)(t0 + u) * n;
    const bool row_ok = t0 + u < n_steps;
#pragma unroll
    for (int k = 0; k < 4; ++k) tmp[u][k] = (row_ok && first + k < (int)n) ? (Action)row[first + k] : Action{};
  }
#pragma unroll
  for (int u = 0; u < CHUNK; ++u) {
    Action* dst = buf + u * kRolloutLanes + 4 * l;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = tmp[u][k];
  }
}

template <class Fam, bool LDS, bool A64>
__global__ void __launch_bounds__(kRolloutThreads) rollout_kernel(const carl_batch_t b, const carl_step_io_t io,
                                                                  const int n_steps) {
  extern __shared__ float lds_dyn[];
  stage_family_tables<Fam>();
  using AStore = action_store_t<Fam, A64>;
  using Action = typename Fam::Action;
  Action* act_buf = reinterpret_cast<Action*>(lds_dyn);                    // [2][kActChunk][256]
  float* lds_ctx = lds_dyn + rollout_action_lds_bytes() / sizeof(float);  // [F][C] when LDS
  const ctx_t<LDS> ctx = make_ctx<LDS, Fam::F>(b, lds_ctx);
  const bool loader = threadIdx.x >= kRolloutLanes;
  const int lane_base = blockIdx.x * kRolloutLanes;
  const int lane = lane_base + (loader ? 0 : (int)threadIdx.x);
  const bool active = !loader && lane < b.n_lanes;
  const uint64_t glane = (uint64_t)(b.lane_offset + lane);
  const size_t n = (size_t)io.row_pitch;  // lanes per row of the action / output arrays (>= n_lanes; resolved by the host)
  const int max_steps = b.max_episode_steps;
  const AStore* act = static_cast<const AStore*>(io.action);
  LaneRegs<Fam> r{};
  Cursors<Fam> cur = make_cursors<Fam>(io, active ? lane : 0);
  if (loader)
    stage_actions<AStore, Action>(act_buf, act, n, lane_base, 0, n_steps);
  else if (active) {
    load_lane<Fam>(b, ctx, lane, r);
    if (!r.episode_valid) {  // a T-step rollout will reset: fetch the RNG counter up front so
      r.episode = b.episode[lane];  // that no reset inside the loop has to wait on memory
      r.episode_valid = true;
      settle(r.episode);
    }
  }
  __syncthreads();
  int buf = 0;
  for (int t0 = 0; t0 < n_steps; t0 += kActChunk, buf ^= 1) {
    if (loader) {
      if (t0 + kActChunk < n_steps)
        stage_actions<AStore, Action>(act_buf + (buf ^ 1) * kActChunk * kRolloutLanes, act, n, lane_base,
                                      t0 + kActChunk, n_steps);
    } else {
      const Action* my = act_buf + buf * kActChunk * kRolloutLanes + threadIdx.x;
      const int steps = min(kActChunk, n_steps - t0);
      Action a_next = my[0];
      settle(a_next);  // see rollout_staged_kernel
      if (lane_base + kRolloutLanes <= b.n_lanes) {  // full workgroup: no per-step predicate
        for (int u = 0; u < steps; ++u) {
          const Action a = a_next;
          a_next = my[min(u + 1, kActChunk - 1) * kRolloutLanes];  // LDS read one step ahead
          step_lane<Fam, ctx_t<LDS>, true>(b, ctx, cur, max_steps, true, lane, glane, a, r);
          cur.advance(n);
        }
      } else {
        for (int u = 0; u < steps; ++u) {
          const Action a = a_next;
          a_next = my[min(u + 1, kActChunk - 1) * kRolloutLanes];
          step_lane<Fam>(b, ctx, cur, max_steps, active, lane, glane, a, r);
          cur.advance(n);
        }
      }
    }
    __syncthreads();
  }
  if (active) store_lane<Fam>(b, ctx, lane, r);
}

// -------------------------------- rollout, LDS-staged outputs -----------------------
// Measured on the direct-store kernel (profiles/r01b + ablations): the four per-step store
// instructions of a compute wave (dwordx3 / dword / byte / byte, 64 lane addresses each) cost
// ~27 % of the step at 65 536 lanes and cap the kernel at 3.3 TB/s at 1 M lanes, against
// 6.9 TB/s for a plain fill: the per-CU address path is paid per lane address, not per byte.
// Here the compute waves write their records into LDS (ds_write, no address VGPR arithmetic)
// and a STORER wave streams them out: a step's records of 256 lanes are contiguous in HBM, so
// it drains them with 16-byte per-lane stores (1 KiB per instruction): 6 wide stores per
// workgroup-step instead of 16 narrow ones.  Double-buffered in chunks of kStageChunk steps,
// one barrier per chunk.
//
// Loading and storing are SEPARATE helper waves (wave 4 loads, waves 5-7 store).  With one
// wave doing both, every chunk paid the full HBM load latency on the barrier path: the loads
// of the next chunk sat behind the previous drain's stores in the wave's single in-order vmcnt
// queue, and had to land before the barrier (~350 ns/step floor for every family, r01d).  The
// loader now holds a chunk of actions IN REGISTERS across the barrier: at iteration c it
// commits chunk c+1 (loaded during iteration c-1) to LDS and issues the loads of chunk c+2, so
// the HBM latency overlaps a whole chunk of compute.  The storer never waits on vmcnt.
constexpr int kStageChunk = 8;
constexpr int kStorers = 3;                                              // storer waves per workgroup (2, 4, 5, 8: +-1 %)
constexpr int kStagedThreads = kRolloutLanes + (1 + kStorers) * kWave;  // + loader wave + storer waves

template <class Fam, int CHUNK = kStageChunk>
__host__ __device__ constexpr size_t rollout_staged_lds_bytes() {
  return (size_t)2 * CHUNK * (LdsSink<Fam>::kStepBytes + kRolloutLanes * sizeof(float));
}

// A chunk of actions in flight: issue() starts the HBM loads into registers, commit() writes
// them to the LDS buffer the compute waves will read.  int64 actions (torch's default integer type)
// travel as two 16-byte vectors per lane-row and are narrowed at commit time.
// DEEP (narrow formats of the families with the shortest step): issue() always loads all CHUNK rows and commit() always
// writes them, so that a second chunk can stay in flight across a commit (see rollout_staged_body).
template <class AStore, class Action, int CHUNK, bool DEEP = false>
struct ActionPipe {
  static_assert(CHUNK == 8 || CHUNK == 4, "the in-flight chunk is held in eight (four) named register groups");
  static constexpr bool kSame = std::is_same_v<AStore, Action>;
  // native 16-byte vectors in named members: first-class register values.  (A float4 array
  // member that is live across the chunk loop stayed a private-memory object -> scratch.)
  typedef float vf4 __attribute__((ext_vector_type(4)));
  typedef int vi4 __attribute__((ext_vector_type(4)));
  using V = std::conditional_t<std::is_same_v<Action, float>, vf4, vi4>;
  struct Wide {  // four int64 actions: little-endian, values fit 32 bits
    vi4 lo, hi;
  };
  static constexpr bool kU8 = std::is_same_v<AStore, unsigned char>;  // four uint8 actions: one dword per lane-row
  static constexpr bool kF16 = std::is_same_v<AStore, f16_bits>, kBF16 = std::is_same_v<AStore, bf16_bits>;
  typedef unsigned int vu2 __attribute__((ext_vector_type(2)));  // four 16-bit floats
  using R = std::conditional_t<kSame, V, std::conditional_t<kU8, unsigned int, std::conditional_t<kF16 || kBF16, vu2, Wide>>>;
  // The narrow formats stay narrow in LDS: the loader wave copies each row's dword(s) as loaded and the COMPUTE waves
  // widen their own element when they read it (ds_read_u8 / ds_read_u16 + at most one conversion): the loader wave's
  // per-chunk work is then eight loads and eight 4- / 8-byte LDS writes whatever the format.  (Pendulum with float16
  // torques at 250-step launches: 66.7 -> 64.2 us against 66.4 with float32; it is NOT what makes MountainCar slower
  // with uint8 actions than with int32 -- 214 -> 225 us per 1 000 steps either way)
  static constexpr bool kNarrow = kU8 || kF16 || kBF16;
  using LdsElem = std::conditional_t<kU8, unsigned char, std::conditional_t<kF16 || kBF16, unsigned short, Action>>;
  using LdsRow = std::conditional_t<kNarrow, R, V>;  // four lanes' worth of one step
  __device__ static __forceinline__ Action widen(LdsElem e) {
    if constexpr (kU8) {
      return (Action)e;
    } else if constexpr (kF16) {
      return (Action)__builtin_bit_cast(_Float16, e);  // every float16 is a float32: exact
    } else if constexpr (kBF16) {
      return __uint_as_float((unsigned int)e << 16);  // bfloat16 = the high half of the float32
    } else {
      return e;
    }
  }
  R a0, a1, a2, a3, a4, a5, a6, a7;
  int t0;

  __device__ static __forceinline__ R load_row(const AStore* __restrict__ p) {
#define CARL_LD(q) __builtin_nontemporal_load(q)  // read once
    if constexpr (kSame) {
      return CARL_LD(reinterpret_cast<const V*>(p));
    } else if constexpr (kU8) {
      return CARL_LD(reinterpret_cast<const unsigned int*>(p));  // (4-byte aligned: n % 16 == 0, base checked by the host)
    } else if constexpr (kF16 || kBF16) {
      return CARL_LD(reinterpret_cast<const vu2*>(p));  // (8-byte aligned likewise)
    } else {
      const vi4* q = reinterpret_cast<const vi4*>(p);
      return Wide{CARL_LD(q), CARL_LD(q + 1)};
    }
#undef CARL_LD
  }
  __device__ static __forceinline__ LdsRow narrow(const R& r) {
    if constexpr (kSame || kNarrow) {
      return r;
    } else {
      return V{r.lo.x, r.lo.z, r.hi.x, r.hi.z};
    }
  }

  bool fast;  // full chunk: its loads are in flight in a0..a7; else (the rollout's last, ragged chunk)
              // commit() loads it itself

  // l: lane of the loader wave (0..63).  ONE unconditional basic block of eight loads: per-row
  // branches (a predicated tail in the same function) made the register allocator spill the in-flight
  // rows and wait for each load before issuing the next.  Lanes past the end of a ragged last
  // workgroup load the batch's last four actions instead (valid memory, valid actions).
  __device__ __forceinline__ void issue(const AStore* __restrict__ act, size_t n, int cols, int lane_base, int l, int t0_,
                                        int n_steps) {
    t0 = t0_;
    fast = t0 + CHUNK <= n_steps;
    if constexpr (DEEP) {
      // two chunks of narrow rows are in flight at a time (rollout_staged_body), and the wait-count pass can only
      // let the YOUNGER set stay in flight across the older set's commit if the number of loads issued here is a
      // compile-time fact: always eight, rows past the end of the rollout re-read its last row (valid memory; `fast`
      // still tells commit() whether the registers hold the chunk)
      const int lane4 = min(lane_base + 4 * l, cols - 4);
      const int last = n_steps - 1;
      a0 = load_row(act + (size_t)min(t0, last) * n + lane4);
      a1 = load_row(act + (size_t)min(t0 + 1, last) * n + lane4);
      a2 = load_row(act + (size_t)min(t0 + 2, last) * n + lane4);
      a3 = load_row(act + (size_t)min(t0 + 3, last) * n + lane4);
      if constexpr (CHUNK == 8) {
        a4 = load_row(act + (size_t)min(t0 + 4, last) * n + lane4);
        a5 = load_row(act + (size_t)min(t0 + 5, last) * n + lane4);
        a6 = load_row(act + (size_t)min(t0 + 6, last) * n + lane4);
        a7 = load_row(act + (size_t)min(t0 + 7, last) * n + lane4);
      }
      return;
    }
    if (!fast) return;
    // uniform row base + 32-bit lane offset: the loads take the scalar-base addressing form
    const int lane4 = min(lane_base + 4 * l, cols - 4);
    const AStore* row = act + (size_t)t0 * n;
    a0 = load_row(row + lane4);
    row += n;
    a1 = load_row(row + lane4);
    row += n;
    a2 = load_row(row + lane4);
    row += n;
    a3 = load_row(row + lane4);
    if constexpr (CHUNK == 8) {
      row += n;
      a4 = load_row(row + lane4);
      row += n;
      a5 = load_row(row + lane4);
      row += n;
      a6 = load_row(row + lane4);
      row += n;
      a7 = load_row(row + lane4);
    }
  }
  __device__ __forceinline__ void commit(LdsElem* buf, const AStore* __restrict__ act, size_t n, int cols, int lane_base, int l,
                                         int n_steps) const {
    // padding lanes of a ragged last workgroup must see VALID actions too (they run as clones of the
    // last lane; a garbage torque sent the Acrobot's angle to 1e7 rad and its wrap loop with it):
    // they get the batch's last four actions (fast path: what issue() loaded) or zeros (tail chunk)
    LdsRow* dst = reinterpret_cast<LdsRow*>(buf + 4 * l);
    constexpr int row = kRolloutLanes / 4;
    if (fast || DEEP) {  // (DEEP: issue() always loads all rows -- past the end, the rollout's last row)
      dst[0] = narrow(a0);
      dst[row] = narrow(a1);
      dst[2 * row] = narrow(a2);
      dst[3 * row] = narrow(a3);
      if constexpr (CHUNK == 8) {
        dst[4 * row] = narrow(a4);
        dst[5 * row] = narrow(a5);
        dst[6 * row] = narrow(a6);
        dst[7 * row] = narrow(a7);
      }
      return;
    }
    // last, ragged chunk of the rollout (once per launch): a plain row loop
    const AStore* src = act + (size_t)t0 * n + min(lane_base + 4 * l, cols - 4);
#pragma unroll 1
    for (int u = 0; u < CHUNK && t0 + u < n_steps; ++u) dst[u * row] = narrow(load_row(src + (size_t)u * n));
  }
};

// storer wave `which` of kStorers: its share (every kStorers-th step) of the records of steps
// [t0, t0 + steps) from LDS to HBM; l = lane of the wave (0..63)
template <class Fam>
__device__ __forceinline__ void drain_records(char* buf, const carl_step_io_t& io, size_t n, int cols, int lane_base, int l,
                                              int which, int t0, int steps) {
  using SK = LdsSink<Fam>;
  // lanes of this workgroup inside the row's columns (cols = n_lanes rounded up to a multiple of 16, <= the pitch n): 256,
  // or a multiple of 16 in the ragged last workgroup, so every 16-byte piece below is entirely inside or entirely
  // outside; columns [n_lanes, cols) of a padded row receive the records of the padding lanes (clones of the batch's
  // last lane) -- nothing beyond cols is ever written
  const int valid = min(kRolloutLanes, cols - lane_base);
  typedef float vf4 __attribute__((ext_vector_type(4)));
  // streamed once, never re-read by this kernel: non-temporal stores (temporal ones: +6 %, measured)
  auto put = [](char* dst, const char* src) {
    const vf4 v = *reinterpret_cast<const vf4*>(src);
    __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(dst));
  };
  for (int u = which; u < steps; u += kStorers) {
    char* rec = buf + (size_t)u * SK::kStepBytes;
    const size_t row = (size_t)(t0 + u) * n + lane_base;
    char* g_obs = reinterpret_cast<char*>(io.obs + row * Fam::D);
    char* fl = rec + SK::kFlagOff + 16 * l;  // [256] term | [256] trunc are contiguous in the record
    const int fl_lane = 16 * (l & 15);
    uint8_t* fl_dst = (l < 16) ? io.terminated + row + fl_lane : io.truncated + row + fl_lane;
    if (valid == kRolloutLanes) {  // full workgroup (wave-uniform): straight-line, the LDS reads of a
                                   // step issue together and the stores follow as they arrive
#pragma unroll
      for (int off = 0; off < SK::kObsBytes; off += 1024) put(g_obs + off + 16 * l, rec + off + 16 * l);
      put(reinterpret_cast<char*>(io.reward + row) + 16 * l, rec + SK::kObsBytes + 16 * l);
      if (l < 32) put(reinterpret_cast<char*>(fl_dst), fl);
    } else {  // ragged last workgroup: per-piece guards (each guard is its own block with its own LDS wait:
              // slower, but only this one workgroup pays it)
#pragma unroll
      for (int off = 0; off < SK::kObsBytes; off += 1024)
        if (off + 16 * l < valid * Fam::D * 4) put(g_obs + off + 16 * l, rec + off + 16 * l);
      if (16 * l < valid * 4) put(reinterpret_cast<char*>(io.reward + row) + 16 * l, rec + SK::kObsBytes + 16 * l);
      if (l < 32 && fl_lane < valid) put(reinterpret_cast<char*>(fl_dst), fl);
    }
    if (l < 32) *reinterpret_cast<vf4*>(fl) = vf4{0.0f, 0.0f, 0.0f, 0.0f};  // LdsSink::kLazyFlags
  }
}

// before the first step: all flag rows of both record buffers to zero (LdsSink::kLazyFlags)
template <class Fam, int CHUNK>
__device__ __forceinline__ void zero_flag_rows(char* out_buf, int l, int which) {
  using SK = LdsSink<Fam>;
  typedef float vf4 __attribute__((ext_vector_type(4)));
  if (l < 32)
    for (int u = which; u < 2 * CHUNK; u += kStorers)
      *reinterpret_cast<vf4*>(out_buf + (size_t)u * SK::kStepBytes + SK::kFlagOff + 16 * l) = vf4{0.0f, 0.0f, 0.0f, 0.0f};
}

// Preconditions (checked by the host): row pitch % 16 == 0 and >= n_lanes (the last workgroup may be ragged; lanes
// [n_lanes, pitch) are padding), global context table.
// LDSCTX (short-episode families under a round-robin / random selector, small tables; chosen by the host):
// the [F][C] context table is staged in LDS behind the record buffers, so the parameter re-gather of a
// lane that moves to another context -- on the done path of nearly every step for CartPole -- is an LDS
// read instead of an HBM / L2 round trip the whole wave waits for.
// MOVES (with PLAIN, kDenseDone families): the dense done handling with context changes on reset (see dense_draw).
// FINAL (with PLAIN, kDenseDone families): ... and with terminal observations written (see step_dense).
// AR (with PLAIN, kDenseDone families): compiled for launches with CARL_FLAG_AUTORESET set (see step_dense).
// CHUNK: steps per LDS buffer (8; the heterogeneous pair launch below runs its families at 4 so that two workgroups
// fit on a compute unit).  `wg`: the workgroup's index among the batch's workgroups (blockIdx.x, or the index inside
// this family's share of a pair launch).
template <class Fam, int A64, bool PLAIN, bool LDSCTX, bool MOVES, bool FINAL, bool AR, int CHUNK, bool DEEPALL = false>
__device__ __forceinline__ void rollout_staged_body(const carl_batch_t& b, const carl_step_io_t& io, const int n_steps,
                                                    const int wg, float* lds_dyn) {
  constexpr int kStageChunk = CHUNK;  // (shadows the namespace-scope default inside this body)
  stage_family_tables<Fam>();
  using AStore = action_store_t<Fam, A64>;
  using Action = typename Fam::Action;
  using SK = LdsSink<Fam>;
  // A second chunk of actions in flight: for the narrow formats (a row is one or two dwords per lane: 8 - 16 registers of
  // the loader wave) of the families whose step is so short that a chunk (1.6 us for MountainCar) does not cover the
  // loader's HBM latency.  MountainCar x 65 536 with uint8 actions: 225 -> 183 us per 1 000 steps (int32: 216).  Not for
  // the others: Pendulum with float16 torques 248 -> 260 us with it (A/B, one box).
  // DEEPALL: the same for int32 / float32 rows (32 more registers in the loader wave), chosen by the host for batches
  // that leave compute units empty (carl_amd.hip: deep_below_lanes): there the launch lasts as long as one workgroup
  // needs, and a loader that waits on HBM is on that path -- CartPole x 8 192 / 16 384 / 32 768: 225 -> 214, 228 -> 217,
  // 237 -> 227 us per 1 000 steps; Pendulum x 8 192 / 16 384: 181 -> 177, 184 -> 180; at 65 536 lanes (store-bound) it
  // costs 0 - 3 % and is not used (A/B on two boxes, tools/ab_probe.sh).
  constexpr bool kDeep = !std::is_same_v<AStore, long long> &&
                         (DEEPALL || (deep_action_prefetch_of<Fam>::value && !std::is_same_v<AStore, Action>));
  using Pipe = ActionPipe<AStore, Action, kStageChunk, kDeep>;
  using LdsAct = typename Pipe::LdsElem;  // = Action, or the narrow storage type (widened by the reader)
  LdsAct* act_buf = reinterpret_cast<LdsAct*>(lds_dyn);  // [2][kStageChunk][256]
  char* out_buf = reinterpret_cast<char*>(lds_dyn) + (size_t)2 * kStageChunk * kRolloutLanes * sizeof(float);
  float* ctx_lds = reinterpret_cast<float*>(out_buf + (size_t)2 * kStageChunk * SK::kStepBytes);
  const ctx_t<LDSCTX> ctx = make_ctx<LDSCTX, Fam::F>(b, ctx_lds);  // LDSCTX: stages the table, then a barrier
  // 0..3 compute, 4 loader, 5.. storers (wave-uniform).  Eight waves = two per SIMD: every
  // compute wave shares its SIMD with exactly one light helper wave, so no compute wave is
  // slowed more than the others before the chunk barrier.
  // (readfirstlane: the compiler then KNOWS the role tests below are wavefront-uniform -- scalar branches and scalar
  // loop control / address arithmetic in the helper waves instead of exec-mask loops over per-lane counters)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int storer = wave - (kRolloutLanes / kWave + 1);  // 0..kStorers-1 on storer waves
  const bool compute = wave < kRolloutLanes / kWave;
  const bool loader = wave == kRolloutLanes / kWave;
  const int hl = threadIdx.x % kWave;    // lane within a helper wave
  const int lane_base = wg * kRolloutLanes;
  const int lane = lane_base + (compute ? (int)threadIdx.x : 0);
  const bool active = compute && lane < b.n_lanes;
  const uint64_t glane = (uint64_t)(b.lane_offset + lane);
  const size_t n = (size_t)io.row_pitch;  // lanes per row of the action / output arrays (>= n_lanes; resolved by the host)
  // columns of a row this launch may touch: the lanes and the padding up to the next multiple of 16 (<= the pitch: host);
  // a pitch beyond that is a view into a wider array whose other columns belong to someone else
  const int n_cols = (b.n_lanes + 15) & ~15;
  const int max_steps = b.max_episode_steps;
  const AStore* act = static_cast<const AStore*>(io.action);
  constexpr int kBufActs = kStageChunk * kRolloutLanes;
  LaneRegs<Fam> r{};
  Pipe pipe;
  [[maybe_unused]] Pipe pipe_b;  // kDeep: the second chunk in flight
  float* const final_base = (io.final_obs != nullptr && active) ? io.final_obs + (size_t)lane * Fam::D : nullptr;
  if (!compute && !loader) zero_flag_rows<Fam, CHUNK>(out_buf, hl, storer);
  if (loader) {
    pipe.issue(act, n, n_cols, lane_base, hl, 0, n_steps);
    pipe.commit(act_buf, act, n, n_cols, lane_base, hl, n_steps);
    pipe.issue(act, n, n_cols, lane_base, hl, kStageChunk, n_steps);  // in flight across the barrier
    if constexpr (kDeep) pipe_b.issue(act, n, n_cols, lane_base, hl, 2 * kStageChunk, n_steps);
  } else if (compute) {
    // padding lanes of a ragged last workgroup run as register-only clones of the batch's last lane
    // (valid numbers, so the step loop needs no per-lane predicate and takes no slow math path)
    const int src = min(lane, b.n_lanes - 1);
    load_lane<Fam>(b, ctx, src, r);
    if (!r.episode_valid) {
      r.episode = b.episode[src];
      r.episode_valid = true;
      settle(r.episode);
    }
    r.valid = active;
  }
  __syncthreads();
  // One chunk loop PER ROLE (the roles are wavefront-uniform: `wave` comes from readfirstlane), each with its own
  // barrier per chunk -- every wavefront of the workgroup executes the same number of s_barriers.  With one shared
  // loop body the compiler's wait-count analysis merged the roles' paths at the loop header: the storer block
  // then waited on vmcnt for loads only the LOADER path has in flight (`s_waitcnt vmcnt(3)` inside the drain loop:
  // a storer stalled until all but three of ITS stores had completed), and all roles shared one register
  // allocation.
  auto run_role = [&](auto role_tag) -> int {
  constexpr int ROLE = decltype(role_tag)::value;  // 0 compute, 1 loader, 2 storer
  if constexpr (ROLE == 1 && kDeep) {
    // The loader's loop with two chunks in flight, two chunks per trip (one barrier each, as every role executes): chunk
    // c+1 (loads issued TWO chunks ago) -> LDS, then chunk c+3 starts in the registers it leaves; likewise c+2 / c+4 in
    // the other set.  Written out per set: with one loop body choosing the set by chunk parity the compiler's
    // wait-count pass cannot tell which set's loads are the older ones and waits for the loads it has just issued.
    // (the only way back to the loop's head is through BOTH halves: on every path into a commit the other set's loads
    // are the younger ones, so its wait is vmcnt(8 + ...), not vmcnt(0))
    int lbuf = 0;
    for (int t0 = 0;;) {  // n_steps >= 1
      pipe.commit(act_buf + (lbuf ^ 1) * kBufActs, act, n, n_cols, lane_base, hl, n_steps);
      pipe.issue(act, n, n_cols, lane_base, hl, t0 + 3 * kStageChunk, n_steps);
      __syncthreads();
      lbuf ^= 1;
      if (t0 + kStageChunk >= n_steps) break;
      pipe_b.commit(act_buf + (lbuf ^ 1) * kBufActs, act, n, n_cols, lane_base, hl, n_steps);
      pipe_b.issue(act, n, n_cols, lane_base, hl, t0 + 4 * kStageChunk, n_steps);
      __syncthreads();
      lbuf ^= 1;
      t0 += 2 * kStageChunk;
      if (t0 >= n_steps) break;
    }
    return lbuf;
  }
  int buf = 0;
  [[maybe_unused]] DenseNext<Fam> nx{};  // dense done handling (PLAIN, kDenseDone families): nothing drawn yet
  [[maybe_unused]] const bool autoreset = (b.flags & CARL_FLAG_AUTORESET) != 0;
  [[maybe_unused]] const int max_steps_eff = max_steps > 0 ? max_steps : 0x7fffffff;
  for (int t0 = 0; t0 < n_steps; t0 += kStageChunk, buf ^= 1) {
    const int steps = min(kStageChunk, n_steps - t0);
    if constexpr (ROLE == 0) {
      const LdsAct* my = act_buf + buf * kBufActs + threadIdx.x;
      char* rec = out_buf + (size_t)buf * kStageChunk * SK::kStepBytes;
      if constexpr (PLAIN && dense_done_of<Fam>::value) {
        // every lane's next init state in registers before the chunk's first step
        if (nx.ok_mask != ~0ull) {
          dense_draw<Fam, ctx_t<LDSCTX>, MOVES>(b, ctx, glane, r, ~nx.ok_mask, nx);
          nx.ok_mask = ~0ull;
        }
        // the chunk's actions in registers: one LDS wait per chunk instead of one per step
        Action acts[kStageChunk];
#pragma unroll
        for (int u = 0; u < kStageChunk; ++u) acts[u] = Pipe::widen(my[u * kRolloutLanes]);
        if (steps == kStageChunk) {  // fully unrolled: record addresses are immediates, no loop control
          // (AR: the chunk's first step is the launch's first step when t0 == 0; two copies of the unrolled chunk
          // would double the kernel for one step's worth of instructions, so that step always takes ENTRY)
#pragma unroll
          for (int u = 0; u < kStageChunk; ++u) {
            const SK sink{rec + (size_t)u * SK::kStepBytes, FINAL ? final_base : nullptr, n * Fam::D, t0 + u,
                          (int)threadIdx.x};
            if (u == 0)
              step_dense<Fam, ctx_t<LDSCTX>, MOVES, FINAL, AR, true, SK>(b, ctx, sink, max_steps_eff, autoreset, glane,
                                                                         acts[u], r, nx);
            else
              step_dense<Fam, ctx_t<LDSCTX>, MOVES, FINAL, AR, false, SK>(b, ctx, sink, max_steps_eff, autoreset, glane,
                                                                          acts[u], r, nx);
          }
        } else {  // the rollout's last, ragged chunk
#pragma unroll 1
          for (int u = 0; u < steps; ++u) {
            const SK sink{rec + (size_t)u * SK::kStepBytes, FINAL ? final_base : nullptr, n * Fam::D, t0 + u,
                          (int)threadIdx.x};
            Action a = acts[0];
#pragma unroll
            for (int k = 1; k < kStageChunk; ++k) a = (u == k) ? acts[k] : a;
            step_dense<Fam, ctx_t<LDSCTX>, MOVES, FINAL, AR, true, SK>(b, ctx, sink, max_steps_eff, autoreset, glane, a, r,
                                                                       nx);
          }
        }
        if constexpr (AR) r.n_new_calls = r.n_new_episodes;
      }
      if constexpr (!(PLAIN && dense_done_of<Fam>::value)) {
      if constexpr (predraw_of<Fam>::value) predraw<Fam>(b, glane, r);
      if (PLAIN && steps == kStageChunk) {
        // full chunk, lean configuration: the chunk's actions in registers (one LDS wait per chunk) and the eight
        // steps unrolled (record addresses become immediates, no loop control, no per-step action read): A/B on
        // one box Pendulum 271 -> 264, MountainCar 254 -> 230, Acrobot 941 -> 910 ns/step, same bits
        Action acts[kStageChunk];
#pragma unroll
        for (int u = 0; u < kStageChunk; ++u) acts[u] = Pipe::widen(my[u * kRolloutLanes]);
#pragma unroll
        for (int u = 0; u < kStageChunk; ++u) {
          const SK sink{rec + (size_t)u * SK::kStepBytes, final_base, n * Fam::D, t0 + u, (int)threadIdx.x};
          step_lane<Fam, ctx_t<LDSCTX>, true, SK, PLAIN>(b, ctx, sink, max_steps, true, lane, glane, acts[u], r);
        }
      } else {
      Action a_next = Pipe::widen(my[0]);
      settle(a_next);  // arrive before the loop: its head then only waits for the read issued one
                       // step earlier (lgkmcnt(#record writes)), not for the record writes
      for (int u = 0; u < steps; ++u) {
        const Action a = a_next;
        a_next = Pipe::widen(my[min(u + 1, kStageChunk - 1) * kRolloutLanes]);
        const SK sink{rec + (size_t)u * SK::kStepBytes, final_base, n * Fam::D, t0 + u, (int)threadIdx.x};
        step_lane<Fam, ctx_t<LDSCTX>, true, SK, PLAIN>(b, ctx, sink, max_steps, true, lane, glane, a, r);
      }
      }
      }
    } else if constexpr (ROLE == 1) {
      // chunk c+1 (loads issued one iteration ago) -> LDS; then start chunk c+2
      pipe.commit(act_buf + (buf ^ 1) * kBufActs, act, n, n_cols, lane_base, hl, n_steps);
      pipe.issue(act, n, n_cols, lane_base, hl, t0 + 2 * kStageChunk, n_steps);
    } else if (t0 > 0) {  // storer: the previous chunk's records (always a full chunk)
      drain_records<Fam>(out_buf + (size_t)(buf ^ 1) * kStageChunk * SK::kStepBytes, io, n, n_cols, lane_base, hl, storer,
                         t0 - kStageChunk, kStageChunk);
    }
    __syncthreads();
  }
  return buf;
  };
  const int buf = compute ? run_role(std::integral_constant<int, 0>{})
                          : loader ? run_role(std::integral_constant<int, 1>{}) : run_role(std::integral_constant<int, 2>{});
  if (compute) {
    if (active) store_lane<Fam>(b, ctx, lane, r);
  } else if (!loader) {  // records of the last chunk
    const int last_t0 = ((n_steps - 1) / kStageChunk) * kStageChunk;
    drain_records<Fam>(out_buf + (size_t)(buf ^ 1) * kStageChunk * SK::kStepBytes, io, n, n_cols, lane_base, hl, storer,
                       last_t0, n_steps - last_t0);
  }
}

template <class Fam, int A64, bool PLAIN = false, bool LDSCTX = false, bool MOVES = false, bool FINAL = false,
          bool AR = false, bool DEEPALL = false>  // A64: the action storage kind (0 int32 / float32, 1 int64, kActU8 ...)
__global__ void __launch_bounds__(kStagedThreads) rollout_staged_kernel(const carl_batch_t b, const carl_step_io_t io,
                                                                        const int n_steps) {
  extern __shared__ float lds_dyn[];
  rollout_staged_body<Fam, A64, PLAIN, LDSCTX, MOVES, FINAL, AR, kStageChunk, DEEPALL>(b, io, n_steps, (int)blockIdx.x,
                                                                                        lds_dyn);
}

// -------------------------------- two families in ONE launch ------------------------
// BASELINE config 3 is a mixed batch (Acrobot + MountainCar).  As two launches the families run back to back: an
// Acrobot workgroup's 139 KB of LDS (8-step chunks) leaves no room for a MountainCar workgroup on the same compute
// unit, and at 65 536 lanes each launch is ONE compute wavefront per SIMD -- Acrobot's float64 RK4 then issues an
// instruction every ~5.3 cycles and the SIMD idles in between, while MountainCar waits for the next launch (350 us
// per 250-step mixed launch: 246 + 59 + two launch boundaries).  Stream-level co-residence was tried in round 3 (a
// fork / join per call costs more than the overlap returns).  Here ONE launch carries both families: workgroups
// [0, grid_a) run family A's staged rollout, [grid_a, grid_a + grid_b) family B's, both at 4-step chunks so that one
// workgroup of each fits a compute unit (2 x (69.6 KB + the 8 KB sin/cos table) < 160 KB) -- B's wavefronts issue
// beside A's.  The blocked order matters: consecutive workgroups go to consecutive XCDs and, inside an XCD, to the
// compute unit with the most free resources, so the first grid_a workgroups take one slot on every compute unit and
// B's take the second (measured with tools/wg_placement: of 512 such workgroups, i and i + 256 share a compute unit,
// always; an interleaved order would put all of A on four XCDs).  What it buys is bounded by the vector ALU: Acrobot's
// float64 stream keeps the SIMD's pipe ~85 % busy by itself, so B's instructions still cost their own pipe time
// (MountainCar: ~25 us of its 58 us) -- 65 536 + 65 536 lanes: 314 us against 277 + 58 + a launch boundary = 340 us
// on the same box; 8 192 + 8 192: 211 against 252 us.  Raising Acrobot's issue priority (s_setprio 3) changes
// nothing (measured).  Same bodies, same arithmetic,
// same per-lane order of operations as the single-family kernel: results are bit-identical to separate launches
// (tests/test_gpu_mixed_and_multiproc.py).  Lean configuration only (PLAIN; AR for a dense-done family), int32 /
// float32 actions: with int64 actions the Acrobot body needs 134 VGPRs and two workgroups no longer share a SIMD.
constexpr int kPairChunk = 4;

template <class FamA, class FamB>
__host__ __device__ constexpr size_t rollout_pair_lds_bytes() {
  constexpr size_t a = rollout_staged_lds_bytes<FamA, kPairChunk>(), b = rollout_staged_lds_bytes<FamB, kPairChunk>();
  return a > b ? a : b;
}

template <class FamA, class FamB, bool ARA, bool ARB>
__global__ void __launch_bounds__(kStagedThreads) __attribute__((amdgpu_waves_per_eu(4)))
rollout_staged_pair_kernel(const carl_batch_t ba, const carl_step_io_t ioa, const carl_batch_t bb, const carl_step_io_t iob,
                           const int n_steps, const int grid_a) {
  extern __shared__ float lds_dyn[];
  if ((int)blockIdx.x < grid_a)  // (workgroup-uniform)
    rollout_staged_body<FamA, false, true, false, false, false, ARA && dense_done_of<FamA>::value, kPairChunk>(
        ba, ioa, n_steps, (int)blockIdx.x, lds_dyn);
  else
    rollout_staged_body<FamB, false, true, false, false, false, ARB && dense_done_of<FamB>::value, kPairChunk>(
        bb, iob, n_steps, (int)blockIdx.x - grid_a, lds_dyn);
}

// -------------------------------- done-mask compaction ------------------------------
// Ordered (ascending lane id) compaction of terminated|truncated in two launches:
//   count: per-block popcount of wave ballots            -> block_counts[nb]
//   write: block offset = sum of lower blocks' counts; within the block each wave's
//          offset = sum of lower waves' popcounts, each lane's rank = mbcnt(ballot).
constexpr int kCompactBlock = 1024;  // 16 waves

__global__ void __launch_bounds__(kCompactBlock) done_count_kernel(const uint8_t* __restrict__ term,
                                                                    const uint8_t* __restrict__ trunc, int n,
                                                                    int32_t* __restrict__ block_counts) {
  __shared__ int wave_counts[kCompactBlock / kWave];
  const int i = blockIdx.x * kCompactBlock + threadIdx.x;
  const bool done = (i < n) && ((term[i] | trunc[i]) != 0);
  const unsigned long long m = ballot(done);
  if (lane_id() == 0) wave_counts[threadIdx.x / kWave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
#pragma unroll
    for (int w = 0; w < kCompactBlock / kWave; ++w) c += wave_counts[w];
    block_counts[blockIdx.x] = c;
  }
}

__global__ void __launch_bounds__(kCompactBlock) done_write_kernel(const uint8_t* __restrict__ term,
                                                                    const uint8_t* __restrict__ trunc, int n,
                                                                    const int32_t* __restrict__ block_counts,
                                                                    int32_t* __restrict__ idx_out,
                                                                    int32_t* __restrict__ count_out) {
  __shared__ int wave_counts[kCompactBlock / kWave];
  __shared__ int partial[kCompactBlock / kWave];
  __shared__ int block_base;
  // offset of this block = sum of the counts of all lower blocks
  int acc = 0;
  for (int k = threadIdx.x; k < (int)blockIdx.x; k += kCompactBlock) acc += block_counts[k];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane_id() == 0) partial[threadIdx.x / kWave] = acc;
  const int i = blockIdx.x * kCompactBlock + threadIdx.x;
  const bool done = (i < n) && ((term[i] | trunc[i]) != 0);
  const unsigned long long m = ballot(done);
  if (lane_id() == 0) wave_counts[threadIdx.x / kWave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int base = 0;
#pragma unroll
    for (int w = 0; w < kCompactBlock / kWave; ++w) base += partial[w];
    block_base = base;
  }
  __syncthreads();
  int wave_off = 0;
  const int wave = threadIdx.x / kWave;
  for (int w = 0; w < wave; ++w) wave_off += wave_counts[w];
  if (done) idx_out[block_base + wave_off + prefix_popc(m)] = i;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    int total = block_base;
#pragma unroll
    for (int w = 0; w < kCompactBlock / kWave; ++w) total += wave_counts[w];
    *count_out = total;
  }
}

}  // namespace carl

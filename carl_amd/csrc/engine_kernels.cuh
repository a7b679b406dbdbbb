// engine_kernels.cuh -- the generic batched reset / step / rollout kernels.
//
// One lane (thread) = one env instance.  All traffic is coalesced: SoA state
// columns, feature-major context rows, lane-major observation records written with
// one wide store per lane.  The arithmetic intensity is a few flop/byte, so these
// kernels are HBM- (large N) or launch-latency- (N ~ 65 536) bound, never MFMA
// material; the levers are bytes moved per step, wave count and launch count.
#pragma once

#include <type_traits>

#include "carl_device.cuh"

namespace carl {

template <bool LDS>
using ctx_t = std::conditional_t<LDS, LdsCtx, GlobalCtx>;

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
template <class Fam, class Ctx>
__device__ __forceinline__ void reset_lane(const carl_batch_t& b, const Ctx& ctx, int lane, uint64_t glane,
                                           int& cidx, uint32_t& episode, float (&s)[Fam::S]) {
  cidx = select_context(b, cidx, glane, episode);
  const u32x4 w = lane_words(b.seed, glane, episode, kSubInit);
  Fam::reset(ctx, cidx, w, s);
  episode += 1u;
  if (b.ctx_obs != nullptr) {
    for (int k = 0; k < b.n_ctx_obs; ++k)
      b.ctx_obs[(size_t)k * b.n_lanes + lane] = ctx.get(b.ctx_obs_feat[k], cidx);
  }
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
  const ctx_t<LDS> ctx = make_ctx<LDS, Fam::F>(b, lds_ctx);
  const int n_work = (idx != nullptr) ? *count : b.n_lanes;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_work; k += gridDim.x * blockDim.x) {
    const int lane = (idx != nullptr) ? idx[k] : k;
    if (mask != nullptr && mask[lane] == 0) continue;
    const uint64_t glane = (uint64_t)(b.lane_offset + lane);
    int cidx = b.ctx_idx[lane];
    uint32_t episode = b.episode[lane];
    float s[Fam::S];
    reset_lane<Fam>(b, ctx, lane, glane, cidx, episode, s);
#pragma unroll
    for (int j = 0; j < Fam::S; ++j) b.state[(size_t)j * b.n_lanes + lane] = s[j];
    b.elapsed[lane] = 0;
    b.ep_return[lane] = 0.0f;
    b.ctx_idx[lane] = cidx;
    b.episode[lane] = episode;
    b.n_calls[lane] += 1;
    if (obs != nullptr) {
      float o[Fam::D];
      Fam::observe(s, o);
      store_obs<Fam::D>(obs, (size_t)lane, o);
    }
  }
}

// ---------------------------- per-lane step body ------------------------------------
// Shared by the per-call kernel (T = 1) and the fused rollout kernel.  Everything a
// lane carries between steps lives in this struct (registers).
template <class Fam>
struct LaneRegs {
  float s[Fam::S];
  float ep_return;
  int elapsed;
  int cidx;
  uint32_t episode;    // lazily loaded: only the reset path (and step noise) reads it
  bool episode_valid;
  int n_new_calls;     // resets performed in this launch
  typename Fam::Params p;
};

template <class Fam, class Ctx>
__device__ __forceinline__ void step_lane(const carl_batch_t& b, const Ctx& ctx, const carl_step_io_t& io,
                                          bool active, int lane, uint64_t glane, size_t out, /* t*n + lane */
                                          typename Fam::Action action, LaneRegs<Fam>& r) {
  bool done = false;
  float o[Fam::D];
  float fin_ret = 0.0f;
  int fin_len = 0;
  if (active) {
    float noise = 0.0f;
    if constexpr (Fam::kNeedsStepNoise) noise = Fam::step_noise(r.p, b, glane, r.episode - 1u, r.elapsed);
    float reward;
    const bool terminated = Fam::step(r.p, r.s, action, noise, r.elapsed, reward);
    r.elapsed += 1;
    // gymnasium TimeLimit.step: truncated = elapsed >= max_episode_steps
    const bool truncated = (b.max_episode_steps > 0) && (r.elapsed >= b.max_episode_steps);
    r.ep_return += reward;
    Fam::observe(r.s, o);
    io.reward[out] = reward;
    io.terminated[out] = (uint8_t)terminated;
    io.truncated[out] = (uint8_t)truncated;
    done = terminated | truncated;
    fin_ret = r.ep_return;
    fin_len = r.elapsed;
    if (done) {
      if (b.last_return) b.last_return[lane] = fin_ret;
      if (b.last_length) b.last_length[lane] = fin_len;
      if (b.episodes_done) b.episodes_done[lane] += 1;
    }
  }
  // finished-episode log: wave ballot + one atomic per wavefront
  log_finished(b, done, glane, fin_ret, fin_len);
  // auto-reset: the branch is wave-uniform (skipped unless some lane of the wave is
  // done), so the Philox rounds cost nothing on the common path
  if ((b.flags & CARL_FLAG_AUTORESET) && __ballot(done) != 0ull) {
    if (done) {
      if (io.final_obs != nullptr) store_obs<Fam::D>(io.final_obs, out, o);
      if (!r.episode_valid) {
        r.episode = b.episode[lane];
        r.episode_valid = true;
      }
      reset_lane<Fam>(b, ctx, lane, glane, r.cidx, r.episode, r.s);
      r.p = Fam::load(ctx, r.cidx, b.flags);
      r.elapsed = 0;
      r.ep_return = 0.0f;
      r.n_new_calls += 1;
      Fam::observe(r.s, o);
    }
  }
  if (active) store_obs<Fam::D>(io.obs, out, o);
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
  r.p = Fam::load(ctx, r.cidx, b.flags);
}

template <class Fam>
__device__ __forceinline__ void store_lane(const carl_batch_t& b, int lane, const LaneRegs<Fam>& r) {
#pragma unroll
  for (int j = 0; j < Fam::S; ++j) b.state[(size_t)j * b.n_lanes + lane] = r.s[j];
  b.elapsed[lane] = r.elapsed;
  b.ep_return[lane] = r.ep_return;
  if (r.n_new_calls != 0) {  // rare: only lanes that were reset in this launch
    b.ctx_idx[lane] = r.cidx;
    b.episode[lane] = r.episode;
    b.n_calls[lane] += r.n_new_calls;
  }
}

// -------------------------------- step (per call) -----------------------------------
template <class Fam, bool LDS>
__global__ void __launch_bounds__(256) step_kernel(const carl_batch_t b, const carl_step_io_t io) {
  extern __shared__ float lds_ctx[];
  const ctx_t<LDS> ctx = make_ctx<LDS, Fam::F>(b, lds_ctx);
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = lane < b.n_lanes;
  const uint64_t glane = (uint64_t)(b.lane_offset + lane);
  LaneRegs<Fam> r{};
  typename Fam::Action action{};
  if (active) {
    load_lane<Fam>(b, ctx, lane, r);
    action = load_action<typename Fam::Action>(io.action, io.action_dtype, (size_t)lane);
  }
  step_lane<Fam>(b, ctx, io, active, lane, glane, (size_t)lane, action, r);
  if (active) store_lane<Fam>(b, lane, r);
}

// -------------------------------- rollout (T steps fused) ---------------------------
// State, context parameters and counters stay in registers for T steps; per step the
// lane reads one action and writes one full transition.  The next action is fetched
// before the current step's dependent arithmetic so its latency is hidden.
template <class Fam, bool LDS>
__global__ void __launch_bounds__(256) rollout_kernel(const carl_batch_t b, const carl_step_io_t io,
                                                      const int n_steps) {
  extern __shared__ float lds_ctx[];
  const ctx_t<LDS> ctx = make_ctx<LDS, Fam::F>(b, lds_ctx);
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = lane < b.n_lanes;
  const uint64_t glane = (uint64_t)(b.lane_offset + lane);
  const size_t n = (size_t)b.n_lanes;
  LaneRegs<Fam> r{};
  typename Fam::Action next{};
  if (active) {
    load_lane<Fam>(b, ctx, lane, r);
    next = load_action<typename Fam::Action>(io.action, io.action_dtype, (size_t)lane);
  }
  for (int t = 0; t < n_steps; ++t) {
    const typename Fam::Action action = next;
    if (active && t + 1 < n_steps)
      next = load_action<typename Fam::Action>(io.action, io.action_dtype, (size_t)(t + 1) * n + lane);
    step_lane<Fam>(b, ctx, io, active, lane, glane, (size_t)t * n + lane, action, r);
  }
  if (active) store_lane<Fam>(b, lane, r);
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
  const unsigned long long m = __ballot(done);
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
  const unsigned long long m = __ballot(done);
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

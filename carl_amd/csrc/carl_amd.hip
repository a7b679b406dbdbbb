// carl_amd.hip -- C-ABI entry points (include/carl_amd.h) and kernel dispatch: classic control,
// context sets, done compaction.  The Brax entry points live in carl_brax.hip (own compile flags).
// gfx950 only.  No persistent device allocations.  The one piece of process-wide state is ensure_dynamic_lds's
// record of which kernels were already granted > 48 KiB of dynamic LDS (an idempotent driver attribute per
// (device, kernel), guarded by a mutex: it only saves a ~2 us driver call per launch).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "../../include/carl_amd.h"
#include "classic_control.hip.h"
#include "context_kernels.hip.h"
#include "engine_kernels.hip.h"
#include "host_common.hpp"

namespace carl_host {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size).  The map below is the library's
// only mutable process-wide state (include/carl_amd.h "Conventions"): a cache of an idempotent attribute, never
// read by the kernels, safe under concurrent callers.
int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* who) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> granted;
  int dev = 0;
  if (const hipError_t e = hipGetDevice(&dev); e != hipSuccess)
    return fail((int)e, "%s: hipGetDevice: %s", who, hipGetErrorString(e));
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[{dev, kernel}];
  if (have >= bytes) return 0;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%zu B of LDS): %s", who, bytes, hipGetErrorString(e));
  have = bytes;
  return 0;
}

}  // namespace carl_host

namespace {

using carl_host::check_launch;
using carl_host::fail;

const carl_family_info_t kInfo[CARL_N_FAMILIES] = {
    /* state obs feat adim disc nact max_steps rsv lo hi */
    {4, 4, 8, 1, 1, 2, 500, 0, 0.0f, 1.0f},     // CartPole-v1
    {2, 3, 7, 1, 0, 0, 200, 0, -2.0f, 2.0f},    // Pendulum-v1
    {4, 6, 14, 1, 1, 3, 500, 0, 0.0f, 2.0f},    // Acrobot-v1
    {2, 2, 11, 1, 1, 3, 200, 0, 0.0f, 2.0f},    // MountainCar-v0
    {2, 2, 10, 1, 0, 0, 999, 0, -1.0f, 1.0f},   // MountainCarContinuous-v0
};

int validate_batch(const carl_batch_t* b, const char* who) {
  if (b == nullptr) return fail(CARL_ERR_INVALID_ARGUMENT, "%s: batch is NULL", who);
  if (b->family < 0 || b->family >= CARL_N_FAMILIES)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: unknown family %d", who, b->family);
  if (b->n_lanes < 0) return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_lanes %d < 0", who, b->n_lanes);
  if (b->n_contexts <= 0 || b->ctx_stride < b->n_contexts)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_contexts %d / ctx_stride %d invalid", who, b->n_contexts,
                b->ctx_stride);
  if (b->selector < CARL_SEL_STATIC || b->selector > CARL_SEL_HOST)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: unknown selector %d", who, b->selector);
  if (!b->state || !b->elapsed || !b->ctx_idx || !b->episode || !b->n_calls || !b->ep_return || !b->ctx_table)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: a required batch pointer is NULL", who);
  if (b->n_ctx_obs < 0 || b->n_ctx_obs > CARL_MAX_CTX_OBS)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_ctx_obs %d out of range", who, b->n_ctx_obs);
  for (int k = 0; k < b->n_ctx_obs; ++k)
    if (b->ctx_obs_feat[k] < 0 || b->ctx_obs_feat[k] >= kInfo[b->family].n_features)
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: ctx_obs_feat[%d] = %d out of range", who, k, b->ctx_obs_feat[k]);
  if (b->n_ctx_obs > 0 && b->ctx_obs == nullptr)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_ctx_obs > 0 but ctx_obs is NULL", who);
  if (b->fin_count != nullptr && (b->fin_capacity <= 0 || !b->fin_lane || !b->fin_return || !b->fin_length))
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: finished-episode log is incomplete", who);
  return 0;
}

int validate_io(const carl_batch_t* b, const carl_step_io_t* io, const char* who) {
  if (io == nullptr) return fail(CARL_ERR_INVALID_ARGUMENT, "%s: io is NULL", who);
  if (!io->action || !io->obs || !io->reward || !io->terminated || !io->truncated)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: a required io pointer is NULL", who);
  const bool discrete = kInfo[b->family].action_is_discrete != 0;
  if (discrete && io->action_dtype != CARL_ACTION_I32 && io->action_dtype != CARL_ACTION_I64 &&
      io->action_dtype != CARL_ACTION_U8)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: discrete family needs int32/int64 (carl_rollout: or uint8) actions", who);
  if (io->action_dtype == CARL_ACTION_U8 && (reinterpret_cast<uintptr_t>(io->action) & 3) != 0)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: uint8 actions must be 4-byte aligned", who);
  const bool half = io->action_dtype == CARL_ACTION_F16 || io->action_dtype == CARL_ACTION_BF16;
  if (!discrete && io->action_dtype != CARL_ACTION_F32 && !half)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: continuous family needs float32 (carl_rollout: or float16 / bfloat16) actions", who);
  if (half && (reinterpret_cast<uintptr_t>(io->action) & 7) != 0)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: float16 / bfloat16 actions must be 8-byte aligned", who);
  if (io->row_pitch != 0 && io->row_pitch < b->n_lanes)  // (carl_step ignores the pitch; a wrong one is refused anyway)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: io.row_pitch %d < n_lanes %d (0 = dense rows)", who, io->row_pitch, b->n_lanes);
  return 0;
}

// Stage the whole [F][C] table in LDS when the context set is small relative to the
// lanes (many lanes share a context, ids are not lane-ordered).
template <class Fam>
bool use_lds_ctx(const carl_batch_t* b) {
  const size_t bytes = (size_t)Fam::F * b->n_contexts * sizeof(float);
  return bytes <= 32 * 1024 && (int64_t)b->n_contexts * 8 <= (int64_t)b->n_lanes;
}

// lanes per row of the action / output arrays of a rollout (carl_step_io_t::row_pitch; 0 = dense rows)
int row_pitch_of(const carl_batch_t* b, const carl_step_io_t* io) {
  return (io != nullptr && io->row_pitch > 0) ? io->row_pitch : b->n_lanes;
}

// The staged kernel writes 16-byte pieces of every row: rows must start on 16-byte boundaries (pitch % 16 == 0) and the
// columns [n_lanes, n_lanes rounded up to 16) must be the caller's to lose -- they are when the lane count is a multiple
// of 16 (there are none) or when the pitch IS that rounded-up count (the padded layout of carl_rollout_pitch).  A wider
// pitch with an odd lane count is a view into an array whose neighbouring columns belong to someone else: direct stores.
int rollout_variant(const carl_batch_t* b, const carl_step_io_t* io = nullptr) {
  if (b->flags & CARL_FLAG_ROLLOUT_DIRECT) return CARL_ROLLOUT_DIRECT_FLAG;
  const int pitch = row_pitch_of(b, io), n16 = (b->n_lanes + 15) / 16 * 16;
  // ... and every array must START on a 16-byte boundary (fresh allocations do; a column view `array[:, k:]` need not)
  bool aligned = true;
  if (io != nullptr) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(io->obs) | reinterpret_cast<uintptr_t>(io->reward) |
                           reinterpret_cast<uintptr_t>(io->terminated) | reinterpret_cast<uintptr_t>(io->truncated) |
                           reinterpret_cast<uintptr_t>(io->final_obs);
    // the loader wave reads a lane-row of four actions per load: 16 bytes (int32 / float32; int64: two of them), 4 (uint8), 8
    const uintptr_t amask = io->action_dtype == CARL_ACTION_U8 ? 3u
                            : (io->action_dtype == CARL_ACTION_F16 || io->action_dtype == CARL_ACTION_BF16) ? 7u : 15u;
    aligned = (bits & 15) == 0 && (reinterpret_cast<uintptr_t>(io->action) & amask) == 0;
  }
  return (aligned && pitch % 16 == 0 && (b->n_lanes % 16 == 0 || pitch == n16)) ? CARL_ROLLOUT_STAGED
                                                                                : CARL_ROLLOUT_DIRECT_SHAPE;
}

// 64-thread workgroups spread a small batch over all 256 CUs x 4 SIMDs (65 536
// lanes = 1024 waves = one per SIMD); large batches use 256.
int pick_block(int n, bool lds) { return (lds || n > 256 * 1024) ? 256 : 64; }

template <class Fam>
int launch_reset(const carl_batch_t* b, const uint8_t* mask, const int32_t* idx, const int32_t* count, float* obs,
                 hipStream_t s) {
  if (b->n_lanes == 0) return 0;
  const bool lds = use_lds_ctx<Fam>(b);
  const int block = 256;
  int grid = (b->n_lanes + block - 1) / block;
  if (grid > 4096) grid = 4096;
  if (lds) {
    const size_t sh = (size_t)Fam::F * b->n_contexts * sizeof(float);
    hipLaunchKernelGGL((carl::reset_kernel<Fam, true>), dim3(grid), dim3(block), sh, s, *b, mask, idx, count, obs);
  } else {
    hipLaunchKernelGGL((carl::reset_kernel<Fam, false>), dim3(grid), dim3(block), 0, s, *b, mask, idx, count, obs);
  }
  return check_launch("carl_reset");
}

template <class Fam>
int launch_step(const carl_batch_t* b, const carl_step_io_t* io, int n_steps, hipStream_t s) {
  if (b->n_lanes == 0 || n_steps == 0) return 0;
  const bool lds = use_lds_ctx<Fam>(b);
  const bool rollout = n_steps >= 0;
  // rollout: 256 compute lanes + one loader wave per workgroup, actions double-buffered
  // in LDS; per-call step: plain lane-per-thread workgroups
  const int block = rollout ? carl::kRolloutThreads : pick_block(b->n_lanes, lds);
  const int lanes_per_block = rollout ? carl::kRolloutLanes : block;
  const int grid = (b->n_lanes + lanes_per_block - 1) / lanes_per_block;
  const size_t sh = (lds ? (size_t)Fam::F * b->n_contexts * sizeof(float) : 0) +
                    (rollout ? carl::rollout_action_lds_bytes() : 0);
  const bool a64 = io->action_dtype == CARL_ACTION_I64;
  // the narrow formats (uint8 / float16 / bfloat16): the lean staged rollout only (include/carl_amd.h: CARL_ACTION_U8)
  const bool au8 = io->action_dtype == CARL_ACTION_U8;
  const bool af16 = io->action_dtype == CARL_ACTION_F16, abf16 = io->action_dtype == CARL_ACTION_BF16;
  if (au8 || af16 || abf16) {
    const bool keeps_context = b->selector == CARL_SEL_STATIC || b->selector == CARL_SEL_HOST;
    const bool lean = b->fin_count == nullptr && io->final_obs == nullptr;
    if (!rollout || rollout_variant(b, io) != CARL_ROLLOUT_STAGED || !keeps_context || !lean || !carl::predraw_of<Fam>::value)
      return fail(CARL_ERR_UNSUPPORTED,
                  "uint8 / float16 / bfloat16 actions: carl_rollout in its lean staged configuration only (row pitch %% 16 == 0, "
                  "static / host selector, no finished-episode log, no final_obs); pass int32 / int64 / float32 actions");
  }
  const dim3 g(grid), t(block);
#define CARL_LAUNCH(KERNEL, ...)                                                                    \
  do {                                                                                              \
    if (lds && a64) hipLaunchKernelGGL((carl::KERNEL<Fam, true, true>), g, t, sh, s, __VA_ARGS__);   \
    else if (lds) hipLaunchKernelGGL((carl::KERNEL<Fam, true, false>), g, t, sh, s, __VA_ARGS__);    \
    else if (a64) hipLaunchKernelGGL((carl::KERNEL<Fam, false, true>), g, t, sh, s, __VA_ARGS__);    \
    else hipLaunchKernelGGL((carl::KERNEL<Fam, false, false>), g, t, sh, s, __VA_ARGS__);            \
  } while (0)
  if (n_steps < 0) {  // per-call step
    CARL_LAUNCH(step_kernel, *b, *io);
    return check_launch("carl_step");
  }
  carl_step_io_t io_resolved = *io;  // the kernels read the pitch as given: never 0
  io_resolved.row_pitch = row_pitch_of(b, io);
  io = &io_resolved;
  // row pitch % 16 == 0 (dense rows: n_lanes % 16 == 0): records are staged in LDS and written out by the workgroup's
  // storer waves with 16-byte stores (rollout_staged_kernel).  Other shapes -- and CARL_FLAG_ROLLOUT_DIRECT, the A/B switch -- take
  // rollout_kernel (per-lane stores, ~50 % slower); carl_rollout_variant() tells a caller which one it gets.
  // (also for tables small enough for LDS: a fused rollout gathers parameters once per launch and on
  // resets, so the global table costs nothing there; the LDS copy pays off in the per-call kernel)
  if (rollout_variant(b, io) == CARL_ROLLOUT_STAGED) {  // 16-byte pieces of every output row stay inside the row
    size_t sh_staged = carl::rollout_staged_lds_bytes<Fam>();
    using kern_t = void (*)(carl_batch_t, carl_step_io_t, int);
    kern_t kern = a64 ? static_cast<kern_t>(carl::rollout_staged_kernel<Fam, true>)
                      : static_cast<kern_t>(carl::rollout_staged_kernel<Fam, false>);
    if constexpr (carl::predraw_of<Fam>::value) {
      // two specialisations of the done path (made for CartPole, whose done path runs on nearly every step;
      // every family opts in: the leaner code also helps the step loop's register allocation)
      const bool keeps_context = b->selector == CARL_SEL_STATIC || b->selector == CARL_SEL_HOST;
      const size_t table_bytes = (size_t)Fam::F * b->n_contexts * sizeof(float);
      const bool lean = b->fin_count == nullptr && io->final_obs == nullptr;
      // static LDS of the kernel (Acrobot's fp64 kernels carry the 8 KiB sin/cos table) counts against the 160 KiB too
      constexpr size_t static_lds = carl::has_tables<Fam>::value ? sizeof(double) * 2 * CARL_SINCOS_TAB_N : 0;
      const bool table_fits = lds && sh_staged + table_bytes + static_lds <= 160 * 1024;
      bool picked = false;
      if (keeps_context && lean) {
        // none of the optional features is on: the done path compiled without them
        kern = a64 ? static_cast<kern_t>(carl::rollout_staged_kernel<Fam, true, true>)
                   : static_cast<kern_t>(carl::rollout_staged_kernel<Fam, false, true>);
        if constexpr (carl::dense_done_of<Fam>::value) {
          // ... and, for the short-episode family, with auto-reset a compile-time fact (step_dense: AR)
          if (b->flags & CARL_FLAG_AUTORESET)
            kern = a64 ? static_cast<kern_t>(carl::rollout_staged_kernel<Fam, true, true, false, false, false, true>)
                       : static_cast<kern_t>(carl::rollout_staged_kernel<Fam, false, true, false, false, false, true>);
        }
        if constexpr (carl::deep_below_lanes_of<Fam>::value > 0) {
          // a batch that leaves compute units empty: two chunks of actions in flight (same results)
          if (!a64 && !au8 && !af16 && !abf16 && b->n_lanes < carl::deep_below_lanes_of<Fam>::value) {
            kern = static_cast<kern_t>(carl::rollout_staged_kernel<Fam, 0, true, false, false, false, false, true>);
            if constexpr (carl::dense_done_of<Fam>::value) {
              if (b->flags & CARL_FLAG_AUTORESET)
                kern = static_cast<kern_t>(carl::rollout_staged_kernel<Fam, 0, true, false, false, false, true, true>);
            }
          }
        }
        if constexpr (std::is_same_v<typename Fam::Action, float>) {
          if (af16) kern = static_cast<kern_t>(carl::rollout_staged_kernel<Fam, carl::kActF16, true>);
          if (abf16) kern = static_cast<kern_t>(carl::rollout_staged_kernel<Fam, carl::kActBF16, true>);
        }
        if constexpr (std::is_same_v<typename Fam::Action, int>) {
          if (au8) {  // the same two kernels reading one byte per action
            kern = static_cast<kern_t>(carl::rollout_staged_kernel<Fam, carl::kActU8, true>);
            if constexpr (carl::dense_done_of<Fam>::value) {
              if (b->flags & CARL_FLAG_AUTORESET)
                kern = static_cast<kern_t>(carl::rollout_staged_kernel<Fam, carl::kActU8, true, false, false, false, true>);
            }
          }
        }
        picked = true;
      }
      if constexpr (carl::dense_done_of<Fam>::value) {
        // short-episode family: the dense done handling also covers lanes that change contexts on reset (round
        // robin -- the reference's default selector -- or random; the next context's parameters are gathered per
        // chunk) and terminal observations; only the finished-episode log still takes the generic path
        if (!picked && b->fin_count == nullptr) {
          const bool moves = !keeps_context, fin = io->final_obs != nullptr, tl = moves && table_fits;
          using carl::rollout_staged_kernel;
#define CARL_DENSE(A, L, M, F) static_cast<kern_t>(rollout_staged_kernel<Fam, A, true, L, M, F>)
          if (a64) {
            kern = tl ? (fin ? CARL_DENSE(true, true, true, true) : CARL_DENSE(true, true, true, false))
                      : moves ? (fin ? CARL_DENSE(true, false, true, true) : CARL_DENSE(true, false, true, false))
                              : CARL_DENSE(true, false, false, true);
          } else {
            kern = tl ? (fin ? CARL_DENSE(false, true, true, true) : CARL_DENSE(false, true, true, false))
                      : moves ? (fin ? CARL_DENSE(false, false, true, true) : CARL_DENSE(false, false, true, false))
                              : CARL_DENSE(false, false, false, true);
          }
#undef CARL_DENSE
          if (tl) sh_staged += table_bytes;
          picked = true;
        }
      }
      if (!picked && !keeps_context && table_fits) {
        // lanes change contexts on reset and the table is small: re-gather from LDS, not from HBM
        kern = a64 ? static_cast<kern_t>(carl::rollout_staged_kernel<Fam, true, false, true>)
                   : static_cast<kern_t>(carl::rollout_staged_kernel<Fam, false, false, true>);
        sh_staged += table_bytes;
      }
    }
    if (int e = carl_host::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), sh_staged, "carl_rollout")) return e;
    const dim3 ts(carl::kStagedThreads);  // 4 compute waves + loader wave + storer wave
    hipLaunchKernelGGL(kern, g, ts, sh_staged, s, *b, *io, n_steps);
    return check_launch("carl_rollout");
  }
  CARL_LAUNCH(rollout_kernel, *b, *io, n_steps);
#undef CARL_LAUNCH
  return check_launch("carl_rollout");
}

// One launch for a mixed batch of two classic families (engine_kernels.hip.h: rollout_staged_pair_kernel).  A is the
// float64 Acrobot -- the family whose single wavefront per SIMD leaves issue slots for another family's wavefronts;
// pairing two memory-bound float32 families would gain nothing.
template <class FamB>
int launch_pair(const carl_batch_t* a, const carl_step_io_t* ioa, const carl_batch_t* b, const carl_step_io_t* iob,
                int n_steps, hipStream_t s) {
  using FamA = carl::Acrobot;
  using kern_t = void (*)(carl_batch_t, carl_step_io_t, carl_batch_t, carl_step_io_t, int, int);
  kern_t kern = static_cast<kern_t>(carl::rollout_staged_pair_kernel<FamA, FamB, false, false>);
  if constexpr (carl::dense_done_of<FamB>::value) {
    if (b->flags & CARL_FLAG_AUTORESET) kern = static_cast<kern_t>(carl::rollout_staged_pair_kernel<FamA, FamB, false, true>);
  }
  const size_t sh = carl::rollout_pair_lds_bytes<FamA, FamB>();
  if (int e = carl_host::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), sh, "carl_rollout_pair")) return e;
  const int grid_a = (a->n_lanes + carl::kRolloutLanes - 1) / carl::kRolloutLanes;
  const int grid_b = (b->n_lanes + carl::kRolloutLanes - 1) / carl::kRolloutLanes;
  carl_step_io_t ra = *ioa, rb = *iob;  // the kernels read the pitch as given: never 0
  ra.row_pitch = row_pitch_of(a, ioa);
  rb.row_pitch = row_pitch_of(b, iob);
  hipLaunchKernelGGL(kern, dim3(grid_a + grid_b), dim3(carl::kStagedThreads), sh, s, *a, ra, *b, rb, n_steps, grid_a);
  return check_launch("carl_rollout_pair");
}

// the lean staged configuration of one part of a pair launch (what launch_step would run as
// rollout_staged_kernel<Fam, false, PLAIN = true>)
bool pair_part_ok(const carl_batch_t* b, const carl_step_io_t* io) {
  const bool keeps_context = b->selector == CARL_SEL_STATIC || b->selector == CARL_SEL_HOST;
  return b->n_lanes > 0 && rollout_variant(b, io) == CARL_ROLLOUT_STAGED && keeps_context && b->fin_count == nullptr &&
         io->final_obs == nullptr && (io->action_dtype == CARL_ACTION_I32 || io->action_dtype == CARL_ACTION_F32);
}

#define CARL_DISPATCH(family, CALL)                                   \
  switch (family) {                                                   \
    case CARL_CARTPOLE: return CALL(carl::CartPole);                  \
    case CARL_PENDULUM: return CALL(carl::Pendulum);                  \
    case CARL_ACROBOT:                                                \
      return (batch->flags & CARL_FLAG_ACROBOT_FP32) ? CALL(carl::AcrobotFast) : CALL(carl::Acrobot); \
    case CARL_MOUNTAINCAR: return CALL(carl::MountainCar);            \
    case CARL_MOUNTAINCAR_CONT: return CALL(carl::MountainCarCont);   \
    default: return fail(CARL_ERR_INVALID_ARGUMENT, "unknown family %d", family); \
  }

// ---------------------------- Brax-locomotion families -----------------------------------
int validate_specs(const carl_feature_spec_t* sd, const carl_feature_spec_t* sh, int n_features, int n_contexts,
                   int ctx_stride, const void* table, const char* who) {
  if (sd == nullptr || sh == nullptr || table == nullptr)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: specs / table pointer is NULL", who);
  if (n_features < 1 || n_features > 256)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_features %d out of range [1, 256]", who, n_features);
  if (n_contexts < 0 || ctx_stride < n_contexts)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_contexts %d / ctx_stride %d invalid", who, n_contexts, ctx_stride);
  for (int f = 0; f < n_features; ++f) {
    const carl_feature_spec_t& s = sh[f];
    if (s.kind < CARL_FEAT_CONSTANT || s.kind > CARL_FEAT_CATEGORICAL)
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: feature %d: unknown kind %d", who, f, s.kind);
    if (s.kind == CARL_FEAT_CATEGORICAL && (s.n_choices < 1 || s.n_choices > CARL_MAX_CHOICES))
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: feature %d: n_choices %d out of range", who, f, s.n_choices);
    if (s.kind != CARL_FEAT_CONSTANT && s.kind != CARL_FEAT_CATEGORICAL && !(s.lower <= s.upper))
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: feature %d: lower %g > upper %g", who, f, s.lower, s.upper);
    if ((s.kind == CARL_FEAT_UNIFORM_FLOAT || s.kind == CARL_FEAT_UNIFORM_INT) &&
        !(s.lower > -3.0e38f && s.upper < 3.0e38f))
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: feature %d: a uniform distribution needs finite bounds", who, f);
    if (s.kind == CARL_FEAT_UNIFORM_FLOAT && s.log_scale && !(s.lower > 0.0f))
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: feature %d: log-uniform needs lower > 0", who, f);
    if (s.kind == CARL_FEAT_NORMAL_FLOAT && !(s.sigma >= 0.0f))
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: feature %d: sigma %g < 0", who, f, s.sigma);
  }
  return 0;
}

}  // namespace

extern "C" {

int carl_abi_version(void) { return CARL_ABI_VERSION; }

const char* carl_last_error(void) { return carl_host::g_err; }

int carl_family_info(int family, carl_family_info_t* out) {
  if (out == nullptr) return fail(CARL_ERR_INVALID_ARGUMENT, "carl_family_info: out is NULL");
  if (family < 0 || family >= CARL_N_FAMILIES)
    return fail(CARL_ERR_INVALID_ARGUMENT, "carl_family_info: unknown family %d", family);
  *out = kInfo[family];
  return 0;
}

int carl_reset(const carl_batch_t* batch, const uint8_t* mask, float* obs, void* stream) {
  if (int e = validate_batch(batch, "carl_reset")) return e;
#define CALL(F) launch_reset<F>(batch, mask, nullptr, nullptr, obs, (hipStream_t)stream)
  CARL_DISPATCH(batch->family, CALL)
#undef CALL
}

int carl_reset_indexed(const carl_batch_t* batch, const int32_t* idx, const int32_t* count, float* obs,
                       void* stream) {
  if (int e = validate_batch(batch, "carl_reset_indexed")) return e;
  if (idx == nullptr || count == nullptr)
    return fail(CARL_ERR_INVALID_ARGUMENT, "carl_reset_indexed: idx/count is NULL");
#define CALL(F) launch_reset<F>(batch, nullptr, idx, count, obs, (hipStream_t)stream)
  CARL_DISPATCH(batch->family, CALL)
#undef CALL
}

int carl_step(const carl_batch_t* batch, const carl_step_io_t* io, void* stream) {
  if (int e = validate_batch(batch, "carl_step")) return e;
  if (int e = validate_io(batch, io, "carl_step")) return e;
#define CALL(F) launch_step<F>(batch, io, -1, (hipStream_t)stream)
  CARL_DISPATCH(batch->family, CALL)
#undef CALL
}

int carl_rollout(const carl_batch_t* batch, const carl_step_io_t* io, int32_t n_steps, void* stream) {
  if (int e = validate_batch(batch, "carl_rollout")) return e;
  if (int e = validate_io(batch, io, "carl_rollout")) return e;
  if (n_steps < 0) return fail(CARL_ERR_INVALID_ARGUMENT, "carl_rollout: n_steps %d < 0", n_steps);
#define CALL(F) launch_step<F>(batch, io, n_steps, (hipStream_t)stream)
  CARL_DISPATCH(batch->family, CALL)
#undef CALL
}

int carl_rollout_pair(const carl_batch_t* batch_a, const carl_step_io_t* io_a, const carl_batch_t* batch_b,
                      const carl_step_io_t* io_b, int32_t n_steps, void* stream) {
  if (int e = validate_batch(batch_a, "carl_rollout_pair")) return e;
  if (int e = validate_batch(batch_b, "carl_rollout_pair")) return e;
  if (int e = validate_io(batch_a, io_a, "carl_rollout_pair")) return e;
  if (int e = validate_io(batch_b, io_b, "carl_rollout_pair")) return e;
  if (n_steps < 0) return fail(CARL_ERR_INVALID_ARGUMENT, "carl_rollout_pair: n_steps %d < 0", n_steps);
  if (n_steps == 0) return 0;
  // exactly one part is the float64 Acrobot (A); the other is any float32 family
  const bool a_acro = batch_a->family == CARL_ACROBOT && !(batch_a->flags & CARL_FLAG_ACROBOT_FP32);
  const bool b_acro = batch_b->family == CARL_ACROBOT && !(batch_b->flags & CARL_FLAG_ACROBOT_FP32);
  if (a_acro == b_acro || batch_a->family == batch_b->family)
    return fail(CARL_ERR_UNSUPPORTED, "carl_rollout_pair: a pair launch is Acrobot (float64) + one other family");
  if (!a_acro) {
    const carl_batch_t* tb = batch_a; batch_a = batch_b; batch_b = tb;
    const carl_step_io_t* ti = io_a; io_a = io_b; io_b = ti;
  }
  if (batch_b->family == CARL_ACROBOT)
    return fail(CARL_ERR_UNSUPPORTED, "carl_rollout_pair: the second family cannot be Acrobot");
  if (!pair_part_ok(batch_a, io_a) || !pair_part_ok(batch_b, io_b))
    return fail(CARL_ERR_UNSUPPORTED, "carl_rollout_pair: both parts must be lean staged rollouts (row pitch %% 16 == 0, static / host "
                "selector, no finished-episode log, no terminal observations, int32 / float32 actions)");
  hipStream_t s = (hipStream_t)stream;
  switch (batch_b->family) {
    case CARL_CARTPOLE: return launch_pair<carl::CartPole>(batch_a, io_a, batch_b, io_b, n_steps, s);
    case CARL_PENDULUM: return launch_pair<carl::Pendulum>(batch_a, io_a, batch_b, io_b, n_steps, s);
    case CARL_MOUNTAINCAR: return launch_pair<carl::MountainCar>(batch_a, io_a, batch_b, io_b, n_steps, s);
    case CARL_MOUNTAINCAR_CONT: return launch_pair<carl::MountainCarCont>(batch_a, io_a, batch_b, io_b, n_steps, s);
    default: return fail(CARL_ERR_UNSUPPORTED, "carl_rollout_pair: family %d", batch_b->family);
  }
}

int carl_rollout_variant(const carl_batch_t* batch) {
  if (batch == nullptr) {
    fail(CARL_ERR_INVALID_ARGUMENT, "carl_rollout_variant: batch is NULL");
    return CARL_ERR_INVALID_ARGUMENT;
  }
  if (batch->family < 0 || batch->family >= CARL_N_FAMILIES) {  // the Brax families have one rollout kernel: not a question
    fail(CARL_ERR_INVALID_ARGUMENT, "carl_rollout_variant: family %d is not a classic-control family", batch->family);
    return CARL_ERR_INVALID_ARGUMENT;
  }
  return rollout_variant(batch);
}

int carl_rollout_variant_io(const carl_batch_t* batch, const carl_step_io_t* io) {
  const int dense = carl_rollout_variant(batch);
  if (dense == CARL_ERR_INVALID_ARGUMENT || io == nullptr) return dense;
  if (io->row_pitch != 0 && io->row_pitch < batch->n_lanes) {
    fail(CARL_ERR_INVALID_ARGUMENT, "carl_rollout_variant_io: io.row_pitch %d < n_lanes %d", io->row_pitch, batch->n_lanes);
    return CARL_ERR_INVALID_ARGUMENT;
  }
  return rollout_variant(batch, io);
}

int32_t carl_rollout_pitch(int32_t n_lanes) { return n_lanes <= 0 ? 0 : (n_lanes + 15) / 16 * 16; }

int32_t carl_done_compact_scratch_elems(int32_t n) {
  return n <= 0 ? 1 : (n + carl::kCompactBlock - 1) / carl::kCompactBlock;
}

int carl_done_compact(const uint8_t* terminated, const uint8_t* truncated, int32_t n, int32_t* idx_out,
                      int32_t* count_out, int32_t* scratch, void* stream) {
  if (n < 0) return fail(CARL_ERR_INVALID_ARGUMENT, "carl_done_compact: n %d < 0", n);
  if (!count_out) return fail(CARL_ERR_INVALID_ARGUMENT, "carl_done_compact: count_out is NULL");
  if (n > 0 && (!terminated || !truncated || !idx_out || !scratch))
    return fail(CARL_ERR_INVALID_ARGUMENT, "carl_done_compact: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {  // empty batch: nothing to read, the count is 0
    const hipError_t e = hipMemsetAsync(count_out, 0, sizeof(int32_t), s);
    return e == hipSuccess ? 0 : fail((int)e, "carl_done_compact: %s", hipGetErrorString(e));
  }
  const int nb = carl_done_compact_scratch_elems(n);
  hipLaunchKernelGGL(carl::done_count_kernel, dim3(nb), dim3(carl::kCompactBlock), 0, s, terminated, truncated, n,
                     scratch);
  hipLaunchKernelGGL(carl::done_write_kernel, dim3(nb), dim3(carl::kCompactBlock), 0, s, terminated, truncated, n,
                     scratch, idx_out, count_out);
  return check_launch("carl_done_compact");
}

int carl_sample_contexts(const carl_feature_spec_t* specs_dev, const carl_feature_spec_t* specs_host,
                         int32_t n_features, int32_t n_contexts, int32_t ctx_stride, int64_t context_offset,
                         uint64_t seed, float* ctx_table, void* stream) {
  if (int e = validate_specs(specs_dev, specs_host, n_features, n_contexts, ctx_stride, ctx_table,
                             "carl_sample_contexts"))
    return e;
  if (n_contexts == 0) return 0;
  const size_t sh = (size_t)n_features * sizeof(carl_feature_spec_t);
  hipLaunchKernelGGL(carl::sample_contexts_kernel, dim3((n_contexts + 255) / 256), dim3(256), sh, (hipStream_t)stream,
                     specs_dev, n_features, n_contexts, ctx_stride, (long long)context_offset, seed, ctx_table);
  return check_launch("carl_sample_contexts");
}

int carl_verify_contexts(const carl_feature_spec_t* specs_dev, const carl_feature_spec_t* specs_host,
                         int32_t n_features, int32_t n_contexts, int32_t ctx_stride, const float* ctx_table,
                         int32_t* n_bad_out, void* stream) {
  if (int e = validate_specs(specs_dev, specs_host, n_features, n_contexts, ctx_stride, ctx_table,
                             "carl_verify_contexts"))
    return e;
  if (n_bad_out == nullptr) return fail(CARL_ERR_INVALID_ARGUMENT, "carl_verify_contexts: n_bad_out is NULL");
  const hipError_t z = hipMemsetAsync(n_bad_out, 0, sizeof(int32_t), (hipStream_t)stream);
  if (z != hipSuccess) return fail((int)z, "carl_verify_contexts: hipMemsetAsync: %s", hipGetErrorString(z));
  if (n_contexts == 0) return 0;
  const size_t sh = (size_t)n_features * sizeof(carl_feature_spec_t);
  hipLaunchKernelGGL(carl::verify_contexts_kernel, dim3((n_contexts + 255) / 256), dim3(256), sh, (hipStream_t)stream,
                     specs_dev, n_features, n_contexts, ctx_stride, ctx_table, n_bad_out);
  return check_launch("carl_verify_contexts");
}

}  // extern "C"

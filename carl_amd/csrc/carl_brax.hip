// carl_brax.hip -- C-ABI entry points of the Brax-locomotion families (include/carl_amd.h) and
// their kernel dispatch.  A translation unit of its own because it is compiled with
// -fno-slp-vectorize: the SLP vectoriser packs the per-lane 3-vector algebra into v_pk_*_f32
// pairs and then spends as many v_mov as it saved to build the register pairs; the kernel is
// bound by instruction issue, and without packing it is 9-22 % faster (profiles/r01g).  (Round 4: the classic-control
// unit is built with the same flag -- carl_amd/build.py; its one useful packing is written by hand.)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/carl_amd.h"
#include "brax_kernels.hip.h"
#include "host_common.hpp"

namespace {

using carl_host::check_launch;
using carl_host::fail;

int validate_brax(const carl_batch_t* b, const carl_brax_sys_t* sd, const carl_brax_sys_t* sh, const char* who) {
  if (b == nullptr || sd == nullptr || sh == nullptr)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: batch / sys pointer is NULL", who);
  if (b->n_lanes < 0) return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_lanes %d < 0", who, b->n_lanes);
  if (b->n_contexts <= 0 || b->ctx_stride < b->n_contexts)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: n_contexts %d / ctx_stride %d invalid", who, b->n_contexts,
                b->ctx_stride);
  if (!b->state || !b->elapsed || !b->ctx_idx || !b->episode || !b->n_calls || !b->ep_return || !b->ctx_table)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: a required batch pointer is NULL", who);
  if (sh->n_links < 1 || sh->n_links > CARL_BRAX_MAX_LINKS || sh->n_dof > CARL_BRAX_MAX_DOF ||
      sh->n_q > CARL_BRAX_MAX_Q || sh->n_act > CARL_BRAX_MAX_ACT || sh->n_coll > CARL_BRAX_MAX_COLL ||
      sh->n_frames < 1 || sh->obs_dim < 1)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: model table out of range", who);
  for (int i = 0; i < sh->n_links; ++i) {
    const bool free_root = sh->parent[i] < 0 && sh->n_link_dof[i] == 6;
    const int n_rot = sh->n_link_dof[i] - sh->n_slide[i];
    const bool hinge = sh->n_slide[i] >= 0 && sh->n_slide[i] <= 2 && n_rot >= 0 && n_rot <= 3 && sh->n_link_dof[i] >= 1;
    if (sh->parent[i] >= i || !(free_root || hinge))
      return fail(CARL_ERR_UNSUPPORTED,
                  "%s: link %d: supported joints are a free root, or 0-2 prismatic dofs + 0-3 stacked hinges", who, i);
    if (!free_root && n_rot == 3 && sh->dof_sign3[i] != 1.0f && sh->dof_sign3[i] != -1.0f)
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: link %d: dof_sign3 must be +1 or -1", who, i);
  }
  if (sh->obs_trig_from < 0 || sh->obs_trig_from >= sh->n_q ||
      (sh->obs_trig_from > 0 && sh->obs_trig_from < sh->exclude_current_positions))
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: obs_trig_from %d out of range", who, sh->obs_trig_from);
  if (sh->tip_link < 0 || sh->tip_link >= sh->n_links)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: tip_link %d out of range", who, sh->tip_link);
  if (sh->tip_link > 0 && sh->target_link == 0 && sh->push_link == 0)
    for (int k = 0; k < 2; ++k)
      if (sh->tip_vel_dof[k] < 0 || sh->tip_vel_dof[k] >= sh->n_dof)
        return fail(CARL_ERR_INVALID_ARGUMENT, "%s: tip_vel_dof[%d] = %d out of range", who, k, sh->tip_vel_dof[k]);
  for (int k = 0; k < sh->n_act; ++k) {
    if (sh->act_dof[k] < 0 || sh->act_dof[k] >= sh->n_dof)
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: actuator %d drives dof %d (of %d)", who, k, sh->act_dof[k], sh->n_dof);
    for (int j = 0; j < k; ++j)
      if (sh->act_dof[j] == sh->act_dof[k])  // the kernel adds actuator torques to their dofs in parallel
        return fail(CARL_ERR_UNSUPPORTED, "%s: actuators %d and %d drive the same dof", who, j, k);
  }
  if (sh->healthy_q_index >= 0 &&
      (sh->healthy_q_index < sh->exclude_current_positions || sh->healthy_q_index >= sh->n_q))
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: healthy_q_index %d must be an observed coordinate", who,
                sh->healthy_q_index);
  if (sh->push_link != 0) {  // push task (see carl_brax_sys_t::push_link)
    const int t = sh->push_link;
    if (t != sh->n_links - 1 || t < 1 || sh->parent[t] != -1 || sh->n_slide[t] != 2 || sh->n_link_dof[t] != 2)
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: push_link must be the last link, on two slides against the world", who);
    if (sh->target_link != 0 || sh->tip_link < 1 || sh->tip_link >= t || sh->exclude_current_positions != 0 ||
        sh->obs_trig_from != 0 || sh->obs_extended || sh->goal_mode || !sh->reset_vel_uniform)
      return fail(CARL_ERR_INVALID_ARGUMENT,
                  "%s: the push task needs an end-effector link, the plain q ++ qd observation and uniform reset rates", who);
    if (sh->q_start[t] != sh->dof_start[t])
      return fail(CARL_ERR_UNSUPPORTED, "%s: push task: the arm must be hinges only", who);
    if (sh->n_pair < 0 || sh->n_pair > CARL_BRAX_MAX_PAIR || (sh->n_pair > 0 && (sh->pair_link < 0 || sh->pair_link >= t)))
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: pair contact table out of range", who);
  } else if (sh->n_pair != 0) {
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: pair contacts belong to the push task", who);
  }
  if (sh->target_link != 0) {  // reach task (see carl_brax_sys_t::target_link)
    const int t = sh->target_link;
    if (t != sh->n_links - 1 || t < 1 || sh->parent[t] != -1 || sh->n_slide[t] != 2 || sh->n_link_dof[t] != 2)
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: target_link must be the last link, on two slides against the world", who);
    if (sh->tip_link < 1 || sh->tip_link >= t || sh->exclude_current_positions != 0 || sh->obs_trig_from != 0 ||
        sh->obs_extended || sh->goal_mode || !sh->reset_vel_uniform)
      return fail(CARL_ERR_INVALID_ARGUMENT,
                  "%s: the reach task needs a tip link, the plain q ++ qd observation and uniform reset rates", who);
    if (sh->n_q + sh->dof_start[t] > 12 * sh->n_links)
      return fail(CARL_ERR_UNSUPPORTED, "%s: reach task: %d coordinates do not fit the staging rows", who, sh->n_q);
  }
  {
    const int base = sh->n_q - sh->exclude_current_positions + sh->n_dof +
                     (sh->obs_trig_from > 0 ? sh->n_q - sh->obs_trig_from : 0);
    int want = sh->obs_extended ? base + 16 * sh->n_links + sh->n_dof : base;
    if (sh->target_link > 0) want = sh->q_start[sh->target_link] + sh->n_q + sh->dof_start[sh->target_link] + 3;
    if (sh->push_link > 0) want = 2 * sh->q_start[sh->push_link] + 9;
    if (sh->obs_dim != want)
      return fail(CARL_ERR_INVALID_ARGUMENT, "%s: obs_dim %d does not match the model (%d)", who, sh->obs_dim, want);
  }
  if (b->fin_count != nullptr && (b->fin_capacity <= 0 || !b->fin_lane || !b->fin_return || !b->fin_length))
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: finished-episode log is incomplete", who);
  if ((b->flags & CARL_FLAG_AUTORESET_FIRST_STATE) && b->first_state == nullptr)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: CARL_FLAG_AUTORESET_FIRST_STATE needs carl_batch_t::first_state", who);
  return 0;
}

// Lanes per env (the kernels' kSub).  The Brax kernel is bound by the instruction stream a
// wavefront issues (~4.3 cycles per wavefront instruction with two resident waves per SIMD,
// profiles/r01g), so a joint or body round with one busy lane per env costs as much as a full one:
// the width is the number of links (every phase = one round), rounded up to an instantiated
// (width, MULTI) pair, and widened for small batches so that the launch has at least two
// wavefronts per SIMD.  carl_brax_sys_t::lanes_per_env pins it (autotune, tests).
constexpr int kBraxWidths[] = {2, 4, 7, 8, 9, 11, 16};
constexpr bool brax_instantiated(int k, bool multi, bool task) {
  if (task) return k == 4 || k == 8 || k == 16;  // reacher: 3 links, pusher: 8
  return multi ? (k == 2 || k == 11 || k == 16) : (k == 4 || k == 7 || k == 8 || k == 9 || k == 16);
}
// The GENERAL kernels (template parameter TASK of brax_kernel): the reach / push task models -- and every model with a link
// whose principal moments of inertia differ.  Every shipped locomotion model has isotropic effective inertia
// (spring_inertia_scale = 1), so the rotated-inertia code R diag(1 / I) R^T lives in the general kernels only: the lean and
// multi-hinge kernels compile isotropy in (brax_kernels.hip.h: substep, all_iso) and carry neither its instructions nor its
// registers through the substep loop.  A general kernel runs a non-task model unchanged (its task epilogues are guarded by
// the model's own target_link / push_link / n_pair).
bool brax_is_task(const carl_brax_sys_t* sh) {
  if (sh->target_link > 0 || sh->push_link > 0) return true;
  for (int i = 0; i < sh->n_links; ++i)
    if (!(sh->inv_inertia[i][0] == sh->inv_inertia[i][1] && sh->inv_inertia[i][1] == sh->inv_inertia[i][2])) return true;
  return false;
}
bool brax_is_planar(const carl_brax_sys_t* sh);
// The general kernels (MULTI): any link with 0, 2 or 3 hinges (Euler-angle path), a link frame that is rotated against
// its parent's (link_rot != identity: the relative rotation of the joint frames then needs the full 4 x 4 map,
// brax_kernels.hip.h: LinkRec), or prismatic dofs outside the planar kernels (whose root slides are their own code) --
// of the shipped models Humanoid / HumanoidStandup (stacked hinges, rotated frames), the inverted pendulums (the cart).
// The lean kernels (MULTI = false) are "a free root and single hinges": Ant.
bool brax_is_multi(const carl_brax_sys_t* sh) {
  bool multi = false, slides = false;
  for (int i = 0; i < sh->n_links; ++i) {
    const bool free_root = sh->parent[i] < 0 && sh->n_link_dof[i] == 6;
    multi |= !free_root && sh->n_link_dof[i] - sh->n_slide[i] != 1;
    multi |= !free_root && !(sh->link_rot[i][0] == 1.0f && sh->link_rot[i][1] == 0.0f && sh->link_rot[i][2] == 0.0f &&
                             sh->link_rot[i][3] == 0.0f);
    slides |= !free_root && sh->n_slide[i] > 0;
  }
  return multi || (slides && !brax_is_planar(sh));
}
// Planar single-hinge model (Halfcheetah, Hopper, Walker2d as this package builds them): the root hangs on the world by
// two slides along x and z and a hinge about y, every other link by one hinge about +-y; link frames are unrotated, the
// joint frame is "x -> +-y", and anchors, centres of mass and spheres all lie in the y = 0 plane.  Then a state produced
// by reset never leaves that plane and brax_kernels.hip.h's substep_planar computes the same substep without the
// zero components.  Anything else -- and any batch with CARL_FLAG_BRAX_GENERIC -- takes the general substep.
bool brax_is_planar(const carl_brax_sys_t* sh) {
  if (brax_is_task(sh) || sh->n_links < 1 || sh->n_pair > 0) return false;  // (the per-link checks below cover the rest)
  const float r = 0.70710678f;
  auto near = [](float a, float b) { return a - b < 1e-6f && b - a < 1e-6f; };
  for (int i = 0; i < sh->n_links; ++i) {
    const bool root = i == 0;
    if (root ? sh->parent[i] >= 0 : (sh->parent[i] < 0 || sh->parent[i] >= i)) return false;
    if (sh->n_slide[i] != (root ? 2 : 0) || sh->n_link_dof[i] != (root ? 3 : 1)) return false;
    if (!(sh->link_rot[i][0] == 1.0f && sh->link_rot[i][1] == 0.0f && sh->link_rot[i][2] == 0.0f && sh->link_rot[i][3] == 0.0f))
      return false;
    if (!(near(sh->joint_rot[i][0], r) && sh->joint_rot[i][1] == 0.0f && sh->joint_rot[i][2] == 0.0f &&
          (near(sh->joint_rot[i][3], r) || near(sh->joint_rot[i][3], -r))))  // x -> +y or x -> -y
      return false;
    if (sh->link_pos[i][1] != 0.0f || sh->joint_pos[i][1] != 0.0f || sh->com[i][1] != 0.0f) return false;
    if (!(sh->inv_inertia[i][0] == sh->inv_inertia[i][1] && sh->inv_inertia[i][1] == sh->inv_inertia[i][2])) return false;
    if (root && !(sh->slide_axis[i][0][0] == 1.0f && sh->slide_axis[i][0][1] == 0.0f && sh->slide_axis[i][0][2] == 0.0f &&
                  sh->slide_axis[i][1][0] == 0.0f && sh->slide_axis[i][1][1] == 0.0f && sh->slide_axis[i][1][2] == 1.0f))
      return false;
  }
  for (int k = 0; k < sh->n_coll; ++k)
    if (sh->coll_pos[k][1] != 0.0f) return false;
  return true;
}

int brax_lanes_per_env(int n_links, bool multi, bool task, int n_lanes, int hint) {
  int want = n_links;
  bool pinned = false;
  if (hint > 0) {  // sys.lanes_per_env (autotuned by the caller); never narrower than one lane per link (the kernels'
                   // mapping since round 5: a lane keeps its link's body in registers through the substeps)
    want = hint > n_links ? hint : n_links;
    pinned = true;
  }
  int k = 16;
  for (int w : kBraxWidths)
    if (w >= want && brax_instantiated(w, multi, task)) {
      k = w;
      break;
    }
  if (!pinned)
    while (k < 16 && ((long long)n_lanes + 64 / k - 1) / (64 / k) < 2048) {
      int next = 16;
      for (int w : kBraxWidths)
        if (w > k && brax_instantiated(w, multi, task)) {
          next = w;
          break;
        }
      k = next;
    }
  return k;
}

template <int MODE>
int launch_brax(const carl_batch_t* b, const carl_brax_sys_t* sd, const carl_brax_sys_t* sh,
                       const carl_step_io_t* io, const uint8_t* mask, float* reset_obs, int n_steps, hipStream_t st,
                       const char* who) {
  if (b->n_lanes == 0 || (MODE == 1 && n_steps == 0)) return 0;
  const bool task = brax_is_task(sh);  // task models have a hinge-less last link: multi
  const bool planar_model = brax_is_planar(sh);
  const bool planar = MODE == 1 && !(b->flags & CARL_FLAG_BRAX_GENERIC) && planar_model;
  // a planar model stepped by the general substep (CARL_FLAG_BRAX_GENERIC): its root's slides need the general kernels
  const bool multi = brax_is_multi(sh) || (MODE == 1 && planar_model && !planar);
  const int K = brax_lanes_per_env(sh->n_links, multi, task, b->n_lanes, sh->lanes_per_env);
  const int envs = carl::brax::kLanes / K;  // one wavefront = envs x K lanes; LDS rows are `envs` floats wide
  const carl::brax::Layout lay = carl::brax::layout_of(*sh);
  // independent wavefronts per workgroup, sharing the LDS copy of the static tables: as many (<= kMaxWavesPerWg) as fit
  // (the kernel's static LDS: model table, prepared topology / records, the fragment hand-over flags)
  const size_t static_lds = sizeof(carl_brax_sys_t) + sizeof(carl::brax::Prepared) +
                            (size_t)carl::brax::kLinkRecBytes * CARL_BRAX_MAX_LINKS + 16 * CARL_BRAX_MAX_DOF +
                            (size_t)carl::brax::kSphRecBytes * (CARL_BRAX_MAX_COLL + 1) +
                            sizeof(int) * carl::brax::kMaxWavesPerWg3;
  const size_t wave_bytes = lay.bytes(envs);
  if (wave_bytes + static_lds > 160 * 1024)
    return fail(CARL_ERR_UNSUPPORTED, "%s: model needs %zu B of LDS per wavefront", who, wave_bytes);
  // Registers allow 2 (multi-hinge / task models) or 3 wavefronts per SIMD = 8 / 12 per CU (brax_kernels.hip.h:
  // CARL_BRAX_WAVES_PER_EU): take the SMALLEST
  // workgroup that gets there LDS-wise (or as close as LDS allows) -- small workgroups retire independently, larger
  // ones cost the launch's tail (Ant, 4 wavefronts per workgroup: 2.97e8 -> 2.57e8 env-steps/s)
  // MODE 1 (step / rollout) with more groups (a group = one wavefront's `envs` envs) than the chip holds at once:
  // launch exactly the resident number of workgroups; each deals its share of the groups to its wavefronts in
  // equal (group, step-range) pieces (brax_kernels.hip.h: run(), "fragments").  Then fewer, larger workgroups
  // lose less to the rounding of groups per workgroup: take the size with the smallest ceil(groups per
  // workgroup) / wavefronts.
  int n_cu = 256;  // (asked of the current device at every launch: no process-wide cache to go stale in a multi-device process)
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n_cu = v;
  }
  const int max_w = carl::brax::max_waves_per_wg(task);
  const int n_groups = (b->n_lanes + envs - 1) / envs;
  int per_cu_of[16] = {0};
  int W = 1, best = 0;
  for (int w = 1; w <= max_w; ++w) {
    const size_t wg = (size_t)w * wave_bytes + static_lds;
    if (wg > 160 * 1024) break;
    if (w > 1 && (long long)(w - 1) * envs >= b->n_lanes) break;  // a small batch: no empty wavefronts
    int per_cu = (int)((160 * 1024) / wg) * w;                     // resident wavefronts per CU, LDS-wise
    const int waves_per_eu = (MODE == 1 && (b->flags & CARL_FLAG_BRAX_FP32)) ? CARL_BRAX_WAVES_PER_EU_F32(task)
                                                                              : CARL_BRAX_WAVES_PER_EU(task);
    if (per_cu > 4 * waves_per_eu) per_cu = 4 * waves_per_eu;
    per_cu -= per_cu % w;  // whole workgroups
    per_cu_of[w] = per_cu;
    if (per_cu > best) {
      best = per_cu;
      W = w;
    }
  }
  int grid = (n_groups + W - 1) / W;
  // (launches of fewer than 4 env steps keep one wavefront per group, dispatched as slots free up: a fragment
  // boundary costs about one env step, which such a launch cannot amortise -- measured, tools/brax_per_call.py:
  // Ant x 32 768, 2-step launches 205 us as groups vs 226 us as fragments; 5-step launches 452 vs 411 us)
  if (MODE == 1 && n_steps >= 4 && (long long)n_groups > (long long)n_cu * best) {  // more than one "round": the balanced schedule
    double best_cost = 1e30;
    for (int w = 1; w <= max_w; ++w) {
      if (per_cu_of[w] != best) continue;
      const int n_wg = n_cu * (best / w);
      const double cost = (double)((n_groups + n_wg - 1) / n_wg) / (double)w;
      if (cost < best_cost - 1e-9) {
        best_cost = cost;
        W = w;
        grid = n_wg;
      }
    }
  }
  const size_t sh_bytes = (size_t)W * wave_bytes;
  using kern_t = void (*)(carl_batch_t, const carl_brax_sys_t*, carl::brax::Prepared, carl_step_io_t, const uint8_t*,
                          float*, int);
  kern_t kern = nullptr;
#define CARL_PICK(KK, MM) \
  if (!task && K == KK && multi == MM) kern = static_cast<kern_t>(carl::brax::brax_kernel<MODE, MM, KK>)
#define CARL_PICK_TASK(KK) \
  if (task && K == KK) kern = static_cast<kern_t>(carl::brax::brax_kernel<MODE, true, KK, true>)
  CARL_PICK_TASK(4);
  CARL_PICK_TASK(8);
  CARL_PICK_TASK(16);
  CARL_PICK(2, true);
  CARL_PICK(11, true);
  CARL_PICK(16, true);
  CARL_PICK(4, false);
  CARL_PICK(7, false);
  CARL_PICK(8, false);
  CARL_PICK(9, false);
  CARL_PICK(16, false);
  if constexpr (MODE == 1) {
#define CARL_PICK_PLANAR(KK) \
  if (planar && K == KK) kern = static_cast<kern_t>(carl::brax::brax_kernel<1, false, KK, false, true>)
    CARL_PICK_PLANAR(4);
    CARL_PICK_PLANAR(7);
    CARL_PICK_PLANAR(8);
    CARL_PICK_PLANAR(9);
    CARL_PICK_PLANAR(16);
#undef CARL_PICK_PLANAR
    // CARL_FLAG_BRAX_FP32 (opt-in): the same kernels with the substeps' pose algebra in float32
    if (b->flags & CARL_FLAG_BRAX_FP32) {
      if (task) return fail(CARL_ERR_UNSUPPORTED, "%s: CARL_FLAG_BRAX_FP32 is not built for the reach / push task models or anisotropic inertia", who);
      kern = nullptr;
#define CARL_PICK_F32(KK, MM) \
  if (!planar && K == KK && multi == MM) kern = static_cast<kern_t>(carl::brax::brax_kernel<1, MM, KK, false, false, true>)
#define CARL_PICK_F32_PLANAR(KK) \
  if (planar && K == KK) kern = static_cast<kern_t>(carl::brax::brax_kernel<1, false, KK, false, true, true>)
      CARL_PICK_F32(2, true);
      CARL_PICK_F32(11, true);
      CARL_PICK_F32(16, true);
      CARL_PICK_F32(4, false);
      CARL_PICK_F32(7, false);
      CARL_PICK_F32(8, false);
      CARL_PICK_F32(9, false);
      CARL_PICK_F32(16, false);
      CARL_PICK_F32_PLANAR(4);
      CARL_PICK_F32_PLANAR(7);
      CARL_PICK_F32_PLANAR(8);
      CARL_PICK_F32_PLANAR(9);
      CARL_PICK_F32_PLANAR(16);
#undef CARL_PICK_F32
#undef CARL_PICK_F32_PLANAR
    }
  }
#undef CARL_PICK
#undef CARL_PICK_TASK
  if (kern == nullptr) return fail(CARL_ERR_UNSUPPORTED, "%s: no kernel for %d lanes per env", who, K);
  if (sh_bytes > 48 * 1024) {
    if (int e = carl_host::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), sh_bytes, who)) return e;
  }
  carl_step_io_t io_v{};
  if (io != nullptr) io_v = *io;
  carl::brax::Prepared prep{};  // topology + derived per-link constants, host-side (microseconds)
  carl::brax::build_topo_host(*sh, prep.topo);
  carl::brax::build_packed_host(*sh, prep.topo, prep.packed);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(carl::brax::kLanes * W), sh_bytes, st, *b, sd, prep, io_v, mask, reset_obs,
                     n_steps);
  return check_launch(who);
}

int validate_brax_io(const carl_step_io_t* io, const char* who) {
  if (io == nullptr) return fail(CARL_ERR_INVALID_ARGUMENT, "%s: io is NULL", who);
  if (!io->action || !io->obs || !io->reward || !io->terminated || !io->truncated)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: a required io pointer is NULL", who);
  if (io->action_dtype != CARL_ACTION_F32)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: Brax families take float32 actions", who);
  if (io->row_pitch != 0)
    return fail(CARL_ERR_INVALID_ARGUMENT, "%s: Brax families take dense rows (io.row_pitch = 0)", who);
  return 0;
}


}  // namespace

extern "C" {

int carl_brax_reset(const carl_batch_t* batch, const carl_brax_sys_t* sys_dev, const carl_brax_sys_t* sys_host,
                    const uint8_t* mask, float* obs, void* stream) {
  if (int e = validate_brax(batch, sys_dev, sys_host, "carl_brax_reset")) return e;
  return launch_brax<0>(batch, sys_dev, sys_host, nullptr, mask, obs, 0, (hipStream_t)stream, "carl_brax_reset");
}

int carl_brax_step(const carl_batch_t* batch, const carl_brax_sys_t* sys_dev, const carl_brax_sys_t* sys_host,
                   const carl_step_io_t* io, void* stream) {
  if (int e = validate_brax(batch, sys_dev, sys_host, "carl_brax_step")) return e;
  if (int e = validate_brax_io(io, "carl_brax_step")) return e;
  return launch_brax<1>(batch, sys_dev, sys_host, io, nullptr, nullptr, 1, (hipStream_t)stream, "carl_brax_step");
}

int carl_brax_rollout(const carl_batch_t* batch, const carl_brax_sys_t* sys_dev, const carl_brax_sys_t* sys_host,
                      const carl_step_io_t* io, int32_t n_steps, void* stream) {
  if (int e = validate_brax(batch, sys_dev, sys_host, "carl_brax_rollout")) return e;
  if (int e = validate_brax_io(io, "carl_brax_rollout")) return e;
  if (n_steps < 0) return fail(CARL_ERR_INVALID_ARGUMENT, "carl_brax_rollout: n_steps %d < 0", n_steps);
  return launch_brax<1>(batch, sys_dev, sys_host, io, nullptr, nullptr, n_steps, (hipStream_t)stream,
                        "carl_brax_rollout");
}

int carl_brax_fragment_plan(int32_t n_groups, int32_t n_workgroups, int32_t waves_per_workgroup, int32_t n_steps,
                            int32_t workgroup, int32_t wave, int32_t* out, int32_t cap) {
  if (n_groups < 1 || n_workgroups < 1 || waves_per_workgroup < 1 || n_steps < 1 || workgroup < 0 ||
      workgroup >= n_workgroups || wave < 0 || wave >= waves_per_workgroup || (cap > 0 && out == nullptr)) {
    fail(CARL_ERR_INVALID_ARGUMENT, "carl_brax_fragment_plan: argument out of range");
    return -1;
  }
  const carl::brax::WgShare sh = carl::brax::wg_share(n_groups, n_workgroups, workgroup);
  const carl::brax::Piece p = carl::brax::make_piece(sh.G, n_steps, waves_per_workgroup, wave);
  for (int fi = 0; fi < p.n_frag && fi < cap; ++fi) {
    const carl::brax::Fragment f = carl::brax::fragment_of(p, n_steps, fi);
    int32_t* o = out + 5 * fi;
    o[0] = sh.g_lo + f.grp; o[1] = f.t_lo; o[2] = f.t_hi; o[3] = f.wait_head ? 1 : 0; o[4] = f.signal_head ? 1 : 0;
  }
  return p.n_frag;
}

#ifdef CARL_BRAX_PROFILE
// measurement build only (not declared in include/carl_amd.h): the region clocks of brax_kernels.hip.h, summed over every
// wavefront since the last reset; out[kProfRegions] = wavefronts
int carl_brax_profile_read(unsigned long long* out, int reset) {
  unsigned long long zero[carl::brax::kProfRegions + 1] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(carl::brax::g_brax_prof), sizeof(zero)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(carl::brax::g_brax_prof), zero, sizeof(zero)) != hipSuccess) return -1;
  return carl::brax::kProfRegions;
}
#endif

int carl_brax_model_is_planar(const carl_brax_sys_t* sys_host) {
  if (sys_host == nullptr) {
    fail(CARL_ERR_INVALID_ARGUMENT, "carl_brax_model_is_planar: NULL argument");
    return 0;
  }
  return brax_is_planar(sys_host) ? 1 : 0;
}

int carl_brax_lane_widths(const carl_brax_sys_t* sys_host, uint32_t batch_flags, int32_t* widths_out, int32_t cap) {
  if (sys_host == nullptr || widths_out == nullptr || cap < 1) {
    fail(CARL_ERR_INVALID_ARGUMENT, "carl_brax_lane_widths: NULL argument");
    return 0;
  }
  // the kernels a STEP / ROLLOUT launch of such a batch takes (launch_brax<1>): a planar model stepped by the general
  // substep (CARL_FLAG_BRAX_GENERIC) runs the multi-hinge kernels, whose widths differ from the planar ones
  const bool multi = brax_is_multi(sys_host) || (brax_is_planar(sys_host) && (batch_flags & CARL_FLAG_BRAX_GENERIC));
  int n = 0;
  for (int w : kBraxWidths)  // one lane per link or wider
    if (w >= sys_host->n_links && brax_instantiated(w, multi, brax_is_task(sys_host)) && n < cap) widths_out[n++] = w;
  return n;
}

}  // extern "C"

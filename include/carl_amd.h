/* carl_amd.h -- C ABI of the MI355X batched step/reset engine for CARL's
 * contextual classic-control (and Brax-locomotion) families.
 *
 * The reference (automl/CARL v1.1.1) has NO FFI on this path: its boundary is a
 * Python protocol, "the object CARLEnv wraps" (SURVEY.md section 8b).  This header
 * is what a ctypes binding on the reference side would load (INTEGRATION.md shows
 * the stub).  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch/HIP types in signatures:
 *    `stream` is a hipStream_t passed as void* (0 = the null stream).
 *  - All array pointers are DEVICE pointers owned by the caller (the Python shim
 *    allocates them as PyTorch-ROCm tensors and passes tensor.data_ptr()).  The
 *    library never allocates persistent device memory and never frees caller
 *    memory; scratch is caller-provided.
 *  - Calls enqueue on `stream` and return without synchronising; one carl_batch_t per
 *    device, caller serialises calls per batch.  No global mutable state that a result
 *    depends on: the only process-wide data is a mutex-guarded record of which kernels
 *    were already granted > 48 KiB of dynamic LDS (an idempotent driver attribute).
 *  - Return value: 0 on success, otherwise a hipError_t value or CARL_ERR_*;
 *    carl_last_error() returns a thread-local message.  Nothing throws or exits.
 *  - Layouts: per-lane state is struct-of-arrays  state[s * n_lanes + lane];
 *    the context table is feature-major  ctx_table[f * ctx_stride + c]  with
 *    features in the order of the reference class's get_context_features();
 *    observations are lane-major  obs[lane * obs_dim + d]  (what a VectorEnv
 *    returns: carl/envs/brax/wrappers.py:111-118).
 */
#ifndef CARL_AMD_H_
#define CARL_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARL_ABI_VERSION 9
#define CARL_MAX_CTX_OBS 32

#define CARL_ERR_INVALID_ARGUMENT (-1)
#define CARL_ERR_UNSUPPORTED (-2)

/* env families.  Context-feature order per family = get_context_features():
 *  CARTPOLE          carl/envs/gymnasium/classic_control/carl_cartpole.py:15-42
 *  PENDULUM          carl/envs/gymnasium/classic_control/carl_pendulum.py:15-39
 *  ACROBOT           carl/envs/gymnasium/classic_control/carl_acrobot.py:15-69
 *  MOUNTAINCAR       carl/envs/gymnasium/classic_control/carl_mountaincar.py:15-51
 *  MOUNTAINCAR_CONT  carl/envs/gymnasium/classic_control/carl_mountaincarcontinuous.py:15-48 */
typedef enum carl_family {
  CARL_CARTPOLE = 0,
  CARL_PENDULUM = 1,
  CARL_ACROBOT = 2,
  CARL_MOUNTAINCAR = 3,
  CARL_MOUNTAINCAR_CONT = 4,
  CARL_N_FAMILIES = 5
} carl_family_t;

/* per-lane context selection rule applied on every reset of a lane
 * (carl/context/selection.py: StaticSelector :125-136, RoundRobinSelector
 * :110-122, RandomSelector :98-107).  HOST = ids are written by the caller
 * (CustomSelector :139-180); the device leaves ctx_idx alone. */
typedef enum carl_selector {
  CARL_SEL_STATIC = 0,
  CARL_SEL_ROUND_ROBIN = 1,
  CARL_SEL_RANDOM = 2,
  CARL_SEL_HOST = 3
} carl_selector_t;

enum {
  CARL_FLAG_AUTORESET = 1,          /* reset done lanes inside step */
  CARL_FLAG_CARTPOLE_RECOMPUTE = 2, /* recompute total_mass / polemass_length from
                                       the context (NOT reference behaviour; the
                                       default replicates Quirk C1) */
  CARL_FLAG_ACROBOT_FP32 = 4,       /* evaluate Acrobot's _dsdt/rk4 in fp32 instead of
                                       fp64 (faster; up to ~1e-3 relative error on
                                       states near the velocity bounds).  Without the flag the step is float64
                                       ARITHMETIC, not a float64 reference: sin / cos come from a 512-entry table with a
                                       short correction (7e-14), 1 / det from v_rcp_f64 + one Newton step (4e-15), the
                                       angle wrap is x - rint(x / 2 pi) 2 pi (the reference's strict > pi / < -pi
                                       compares differ for the doubles within one ulp of +-pi), the state is stored as
                                       float32 and the observation / _terminal trig is taken at the STORED angles
                                       (2e-7 from the reference's unrounded ones).  Each transition is within 1e-5 of
                                       gymnasium's arithmetic; Acrobot is chaotic, so free-running trajectories
                                       separate from a true float64 run at the float32 state rounding's rate */
  CARL_FLAG_ROLLOUT_DIRECT = 16,    /* carl_rollout: launch the direct-store kernel even where the staged one
                                       applies (A/B measurements and tests; ~50 % slower, same results) */
  CARL_FLAG_AUTORESET_FIRST_STATE = 8 /* Brax families, with AUTORESET: a done env is put back to the state
                                       its last explicit reset produced -- no new draw, no selector advance,
                                       no context change: brax.envs.wrappers.training.AutoResetWrapper as the
                                       reference reaches it through carl/envs/brax/wrappers.py:54-78,121-145
                                       (needs carl_batch_t::first_state).  Default: re-draw. */
  ,
  CARL_FLAG_BRAX_GENERIC = 32         /* Brax families: step planar models (Halfcheetah, Hopper, Walker2d) with the general
                                       3-D substep too -- the A/B and test switch for the planar substep, which the
                                       library otherwise picks for models whose joints and geometry lie in the y = 0
                                       plane (same records, same results to rounding, a third of the instructions).  A
                                       planar model's state must BE planar (what reset produces); a caller that sets
                                       an out-of-plane state itself passes this flag. */
  ,
  CARL_FLAG_BRAX_FP32 = 64            /* Brax families, OPT-IN, never a default (ABI 9): carl_brax_step / carl_brax_rollout run the
                                       n_frames substeps of an env step with their pose algebra (anchor separation, relative
                                       joint rotation, joint angles, contact depth, integration) in float32 -- the precision
                                       brax itself runs at under JAX's default (carl/envs/brax/carl_brax_env.py:163-176 creates
                                       the env without enabling x64).  The product path forms these differences of poses in
                                       float64 to stay within north_star's 1e-5 of the float64 restatement; this flag tells a
                                       user what that bar costs: faster, and off the restatement by the amounts reported in
                                       profiles/r06_brax_fp32_deviation.txt (like CARL_FLAG_ACROBOT_FP32 for the classic
                                       path).  Between env steps the state record is the same 48-bit pose; reset, observe
                                       and reward are unchanged.  Not built for the reach / push task models
                                       (CARL_ERR_UNSUPPORTED). */
};

/* Storage type of carl_step_io.action.  The reference's discrete action is whatever integer the caller's policy
 * produced (gymnasium Discrete: np.int64; carl/envs/carl_env.py:321 passes it through): I32 / I64 everywhere.
 * U8 (ABI 7) is a ROLLOUT-ONLY input format for discrete families: one byte per lane-step -- the action stream is the
 * fused rollout's only per-step read, and at 4 bytes it costs a fifth of a CartPole launch.  Accepted by carl_rollout
 * in its lean staged configuration (row pitch % 16 == 0 -- dense rows: n_lanes % 16 == 0 --, STATIC / HOST selector, no
 * finished-episode log, no final_obs;
 * action pointer 4-byte aligned); anything else returns CARL_ERR_UNSUPPORTED and the caller widens the actions.
 * Same values in, same transitions out: bit-identical to the I32 launch (tests/test_gpu_parity.py). */
enum { CARL_ACTION_I32 = 0, CARL_ACTION_I64 = 1, CARL_ACTION_F32 = 2, CARL_ACTION_U8 = 3,
       /* Box families, carl_rollout's lean staged configuration only (pointer 8-byte aligned): IEEE float16 / bfloat16 as
        * a policy network under autocast emits them; widened exactly, so the transitions are those of the float32
        * launch fed the widened values */
       CARL_ACTION_F16 = 4, CARL_ACTION_BF16 = 5 };

typedef struct carl_family_info {
  int32_t state_dim;          /* S: columns of `state` */
  int32_t obs_dim;            /* D */
  int32_t n_features;         /* F: rows of ctx_table */
  int32_t action_dim;         /* 1 */
  int32_t action_is_discrete; /* 1: Discrete(n_actions); 0: Box */
  int32_t n_actions;          /* discrete: 2 or 3 */
  int32_t max_episode_steps;  /* gymnasium registry TimeLimit default */
  int32_t reserved;
  float action_low, action_high; /* Box bounds when continuous */
} carl_family_info_t;

/* One batch of lanes (env instances) resident on one device.  Replaces the
 * attribute state of N x {CARLEnv + TimeLimit + gymnasium env} objects:
 *   state      <- env.unwrapped.state              (carl_cartpole.py:51 ...)
 *   elapsed    <- TimeLimit._elapsed_steps         (gymnasium.make, carl_gymnasium_env.py:64)
 *   ctx_idx    <- context_selector.context_id      (carl_env.py:118-120)
 *   n_calls    <- context_selector.n_calls         (selection.py:45,75)
 *   ctx_table  <- env.contexts after insert_defaults (carl_env.py:122-137) and the
 *                 setattr loop of CARLGymnasiumEnv._update_context
 *                 (carl_gymnasium_env.py:75-77)
 *   ctx_obs    <- the "context" half of the observation dict (carl_env.py:276-305)
 * RNG: Philox4x32-10, key = seed, counter = (global lane id lo, hi, episode, sub);
 * global lane id = lane_offset + lane, so results do not depend on how lanes are
 * split across devices. */
typedef struct carl_batch {
  int32_t family;            /* carl_family_t */
  int32_t n_lanes;
  int32_t n_contexts;        /* C: valid columns of ctx_table */
  int32_t ctx_stride;        /* elements between feature rows, >= n_contexts */
  int32_t max_episode_steps; /* TimeLimit; <= 0 -> never truncate */
  int32_t selector;          /* carl_selector_t */
  int32_t selector_stride;   /* round robin: id = (id + stride) mod C */
  int32_t flags;             /* CARL_FLAG_* */
  int64_t lane_offset;
  uint64_t seed;
  /* persistent per-lane state */
  float* state;              /* [S][n_lanes] */
  int32_t* elapsed;          /* [n_lanes] */
  int32_t* ctx_idx;          /* [n_lanes] in [0, C) */
  uint32_t* episode;         /* [n_lanes] resets so far under this seed (RNG counter) */
  int32_t* n_calls;          /* [n_lanes] selector calls so far */
  float* ep_return;          /* [n_lanes] running return of the current episode */
  /* contexts */
  const float* ctx_table;    /* [F][ctx_stride] */
  float* ctx_obs;            /* [n_ctx_obs][n_lanes] or NULL */
  int32_t n_ctx_obs;
  int32_t ctx_obs_feat[CARL_MAX_CTX_OBS]; /* table row of each observed feature */
  int32_t fin_capacity;      /* entries in the finished-episode log */
  /* episode statistics, all nullable */
  float* last_return;        /* [n_lanes] return of the lane's last finished episode */
  int32_t* last_length;      /* [n_lanes] */
  int32_t* episodes_done;    /* [n_lanes] finished-episode count */
  /* finished-episode log (wave-ballot compaction, one atomic per wavefront):
   * entry k < min(*fin_count, fin_capacity) = one finished episode.  Entry order
   * is unspecified; the multiset of entries is deterministic. */
  int32_t* fin_count;        /* [1] or NULL */
  int64_t* fin_lane;         /* [fin_capacity] global lane id */
  float* fin_return;         /* [fin_capacity] */
  int32_t* fin_length;       /* [fin_capacity] */
  /* goal-directed Brax mode (BraxWalkerGoalWrapper, brax_walker_goal_wrapper.py:113-140);
   * NULL otherwise */
  float* goal_pos;           /* [2][n_lanes] position integrated from the observed x/y velocities */
  uint8_t* success;          /* [n_lanes] (or [T][n_lanes] in a rollout): goal reached on this step */
  /* Brax families, CARL_FLAG_AUTORESET_FIRST_STATE: [n_lanes][20 L] (records like `state`), written by carl_brax_reset, read by the
   * in-kernel auto-reset; NULL otherwise */
  float* first_state;
} carl_batch_t;

/* Inputs/outputs of one step (or, for carl_rollout, of T steps: every array gains
 * a leading [T] dimension).  Replaces the return tuple of CARLEnv.step
 * (carl/envs/carl_env.py:321-342) minus the context half (see carl_batch.ctx_obs). */
typedef struct carl_step_io {
  const void* action;   /* [n_lanes * action_dim], dtype per action_dtype */
  int32_t action_dtype; /* CARL_ACTION_* ; discrete families take I32/I64 (carl_rollout also U8), Box F32 (carl_rollout
                           also F16 / BF16) */
  int32_t row_pitch;    /* carl_rollout / carl_rollout_pair (classic-control families; ABI 9): lanes per ROW of `action` and of
                           every [T][...] output -- row t of an array starts row_pitch lanes after row t - 1; 0 = n_lanes (dense
                           rows).  Must be 0 or >= n_lanes.  The staged kernel (16-byte stores) runs when the pitch is a multiple
                           of 16 AND either n_lanes is one too or the pitch is exactly carl_rollout_pitch(n_lanes): it then also
                           WRITES columns [n_lanes, carl_rollout_pitch(n_lanes)) of every output row (the records of the padding
                           lanes) and reads the same columns of the action rows, which must hold valid actions (the engine
                           repeats the last lane's) -- never anything beyond.  Any other pitch (a view into a wider array) takes
                           the direct-store kernel, which touches columns [0, n_lanes) only.  Ignored by carl_step; the Brax
                           entry points take 0 only.  (this field was `reserved`, always 0, in ABI <= 8) */
  float* obs;           /* [n_lanes][D]; with AUTORESET the post-reset observation
                           for done lanes (gymnasium vector-env convention) */
  float* reward;        /* [n_lanes] */
  uint8_t* terminated;  /* [n_lanes] */
  uint8_t* truncated;   /* [n_lanes] TimeLimit */
  float* final_obs;     /* [n_lanes][D] or NULL; written for done lanes only */
  uint8_t* done;        /* [n_lanes] or NULL: terminated | truncated, the done mask of this step (what
                           info["_final_observation"] of a vector env is); per-call carl_step / carl_brax_step
                           only, ignored by the rollout entry points */
  uint32_t* branch_sig; /* Brax families: [n_lanes][2] or NULL.  Hash of the DISCRETE decisions the physics took in
                           this env step: [0] which collision spheres delivered an impulse in which substep
                           (discontinuous: an impulse appears when a point starts to approach; push task: also which
                           gripper / object contacts pushed -- round 6), [1] which joint
                           range limits were active (continuous: the limit spring starts at zero).  Two
                           implementations of the same arithmetic can only be compared to rounding on lanes
                           whose word [0] agrees; tests/test_gpu_brax.py uses it for exactly that.  Ignored by
                           the classic-control entry points. */
} carl_step_io_t;

int carl_abi_version(void);
const char* carl_last_error(void);

/* static facts about a family: replaces reading env.observation_space /
 * env.action_space / spec.max_episode_steps of the gymnasium env the reference
 * builds with gymnasium.make (carl_gymnasium_env.py:63-64). */
int carl_family_info(int family, carl_family_info_t* out);

/* CARLEnv.reset for the lanes selected by `mask` (device u8 [n_lanes]; NULL = all):
 * selector advance -> (context switch) -> CARL init-state draw -> elapsed = 0 -> obs.
 * Replaces carl/envs/carl_env.py:245-274 + the per-family reset overrides
 * (carl_cartpole.py:44-66, carl_pendulum.py:41-65, carl_acrobot.py:71-115,
 * carl_mountaincar.py:53-85, carl_mountaincarcontinuous.py:50-82).
 * `obs` [n_lanes][D] receives the initial observation of reset lanes only. */
int carl_reset(const carl_batch_t* batch, const uint8_t* mask, float* obs, void* stream);

/* Same, for the compact list idx[0 .. *count) produced by carl_done_compact
 * ("reset_masked over the compacted done list"). */
int carl_reset_indexed(const carl_batch_t* batch, const int32_t* idx, const int32_t* count,
                       float* obs, void* stream);

/* CARLEnv.step over all lanes: env physics + TimeLimit + episodic-return
 * bookkeeping + (AUTORESET) in-kernel reset of done lanes.  Replaces
 * carl/envs/carl_env.py:321-342 -> gymnasium TimeLimit.step -> <Env>.step. */
int carl_step(const carl_batch_t* batch, const carl_step_io_t* io, void* stream);

/* T consecutive steps in ONE launch, state held in registers between steps; every
 * step still emits its full transition (obs/reward/terminated/truncated at
 * [t][lane]).  action is [T][n_lanes].  No reference counterpart (the reference
 * steps one env object per Python call); exists because a single step of 65 536
 * lanes is shorter than a kernel launch. */
int carl_rollout(const carl_batch_t* batch, const carl_step_io_t* io, int32_t n_steps, void* stream);

/* T consecutive steps of TWO batches of different classic-control families in ONE launch: BASELINE config 3's mixed
 * batch ("CARLAcrobot + CARLMountainCar", carl/envs/gymnasium/classic_control/carl_acrobot.py:11-115 +
 * carl_mountaincar.py:11-85 -- in the reference two unrelated env objects, carl/envs/carl_env.py:245-342 holds no
 * cross-env state).  One part must be the float64 Acrobot, the other any other family; both in the lean staged
 * configuration (row pitch % 16 == 0, static / host selector, no finished-episode log, io.final_obs NULL, int32 /
 * float32 actions).  Each family's workgroups run the same staged rollout as carl_rollout at 4-step chunks, so one
 * workgroup of each fits a compute unit and the second family's wavefronts issue in the gaps of Acrobot's RK4:
 * results are bit-identical to two carl_rollout calls, the launch takes ~max instead of the sum.
 * Returns CARL_ERR_UNSUPPORTED (nothing enqueued) for any other combination: the caller then makes the two
 * carl_rollout calls itself (carl_amd.mixed.MixedVecEngine.rollout does). */
int carl_rollout_pair(const carl_batch_t* batch_a, const carl_step_io_t* io_a, const carl_batch_t* batch_b,
                      const carl_step_io_t* io_b, int32_t n_steps, void* stream);

/* Which kernel carl_rollout launches for this batch (classic-control families).  The staged kernel (transition
 * records assembled in LDS, written with 16-byte stores by dedicated waves) needs rows of a pitch that is a multiple of
 * 16 lanes; this entry point answers for DENSE rows (io.row_pitch = 0: n_lanes % 16 == 0).  Any other lane count
 * silently took the ~50 % slower direct-store kernel in ABI <= 4; a caller could ask since ABI 5; since ABI 9 it lays
 * its rows out at carl_rollout_pitch(n_lanes) and gets the staged kernel (carl_rollout_variant_io). */
enum { CARL_ROLLOUT_STAGED = 0, CARL_ROLLOUT_DIRECT_SHAPE = 1, CARL_ROLLOUT_DIRECT_FLAG = 2 };
int carl_rollout_variant(const carl_batch_t* batch); /* CARL_ERR_INVALID_ARGUMENT for a non-classic family */
/* ... for this batch WITH these buffers (ABI 9): the staged kernel needs (io->row_pitch ? io->row_pitch : n_lanes) % 16
 * == 0 -- a caller whose lane count is not a multiple of 16 lays its rows out at carl_rollout_pitch(n_lanes) and gets
 * the staged kernel (carl_amd.engine.VecEngine.alloc_rollout does; uneven lane shards of a multi-GPU run end up here:
 * carl_amd/distributed.py::lane_shard).  No reference counterpart: the reference has no batched layout at all
 * (carl/envs/carl_env.py:321-342 returns one env's tuple). */
int carl_rollout_variant_io(const carl_batch_t* batch, const carl_step_io_t* io);
int32_t carl_rollout_pitch(int32_t n_lanes); /* n_lanes rounded up to the next multiple of 16 (0 for n_lanes <= 0) */

/* done-mask compaction: ascending lane ids with terminated|truncated set.
 * idx_out [n], count_out [1], scratch >= carl_done_compact_scratch_elems(n) int32.
 * No reference counterpart (the gymnasium path has no auto-reset; the user calls
 * reset() per env, carl_env.py:245). */
int carl_done_compact(const uint8_t* terminated, const uint8_t* truncated, int32_t n,
                      int32_t* idx_out, int32_t* count_out, int32_t* scratch, void* stream);
int32_t carl_done_compact_scratch_elems(int32_t n);

/* ======================= Brax-locomotion families (spring backend) =======================
 * Replaces CARLBraxEnv + BraxGymWrapper/VectorGymWrapper + brax.spring.pipeline.step x
 * n_frames + brax.envs.<env>.step/reset (carl/envs/brax/carl_brax_env.py:115-336,
 * carl/envs/brax/wrappers.py:32-158; brax==0.12.1 itself is NOT in the reference tree:
 * the pipeline is restated from upstream memory, SURVEY.md section 8a "Brax restatement";
 * PARITY UNPINNED).  One maximal-coordinate rigid-body system per lane; the model (links,
 * joints, colliders, actuators, env reward constants) is a carl_brax_sys_t table shared by
 * all lanes, the per-lane context overrides gravity / friction / elasticity / ang_damping /
 * link masses / joint-stiffness scale. */
#define CARL_BRAX_MAX_LINKS 16
#define CARL_BRAX_MAX_DOF 24
#define CARL_BRAX_MAX_Q 32
#define CARL_BRAX_MAX_ACT 24
#define CARL_BRAX_MAX_COLL 32
#define CARL_BRAX_MAX_CTX_MASS 16
#define CARL_BRAX_MAX_PAIR 8

enum { CARL_BRAX_ANT = 0, CARL_BRAX_HALFCHEETAH = 1, CARL_BRAX_HUMANOID = 2, CARL_BRAX_HOPPER = 3, CARL_BRAX_WALKER2D = 4,
       CARL_BRAX_INVERTED_PENDULUM = 5, CARL_BRAX_HUMANOIDSTANDUP = 6, CARL_BRAX_INVERTED_DOUBLE_PENDULUM = 7,
       CARL_BRAX_REACHER = 8, CARL_BRAX_PUSHER = 9 };

/* context-table rows the physics reads (row index in the family's feature table, -1 =
 * feature absent -> the model default is used) */
typedef struct carl_brax_ctx_map {
  int32_t gravity, friction, elasticity, ang_damping, joint_stiffness_scale;
  int32_t target_distance, target_direction, target_radius; /* goal mode rows (carl_ant.py:40-48) */
  int32_t n_mass;                               /* mass_<link> features */
  int32_t mass_row[CARL_BRAX_MAX_CTX_MASS];     /* table row */
  int32_t mass_link[CARL_BRAX_MAX_CTX_MASS];    /* link it scales */
  float mass_nominal[CARL_BRAX_MAX_CTX_MASS];   /* CARL default: value / nominal scales the link's effective mass
                                                 * (an EXTENSION of this build, like joint_stiffness: with brax's
                                                 * spring_mass_scale = 1 the spring backend runs every link at
                                                 * m**(1 - 1) = 1, so upstream a mass context would not move it) */
  float mass_ratio_floor[CARL_BRAX_MAX_CTX_MASS]; /* the ratio is clamped from below at this value per env (0 = no
                                                 * clamp): lighter links push the explicit spring integration past
                                                 * its stability bound (NaNs within a few steps); the context
                                                 * OBSERVATION keeps the unclamped value.  This floor applies to an
                                                 * env in which ONE mass feature is lighter than nominal ... */
  float mass_ratio_floor_multi[CARL_BRAX_MAX_CTX_MASS]; /* ... and this (higher) one when two or more are (ratio <
                                                 * 0.999): several light links at once are less stable than each
                                                 * alone (measured: tools/mass_combo_sweep.py) */
  int32_t goal_position[3];                     /* push task: goal_position_x / _y / _z rows (carl_pusher.py:80-103) */
} carl_brax_ctx_map_t;

typedef struct carl_brax_sys {
  int32_t env_kind;      /* CARL_BRAX_* : selects reward / obs / done rule */
  int32_t n_links, n_q, n_dof, n_act, n_coll, n_frames, obs_dim;
  int32_t max_episode_steps;   /* brax EpisodeWrapper episode_length (1000) */
  int32_t terminate_when_unhealthy;
  int32_t exclude_current_positions;  /* leading q entries left out of the observation (Ant 2, Halfcheetah 1) */
  int32_t reserved;
  float dt;              /* substep */
  float gravity_z, vel_damping, ang_damping, baumgarte_erp, elasticity, friction;
  float healthy_z_lo, healthy_z_hi, healthy_reward, ctrl_cost_weight, forward_reward_weight;
  float reset_noise_scale, reset_vel_scale;
  /* links, topological order, parent < child */
  int32_t parent[CARL_BRAX_MAX_LINKS];      /* -1: free root */
  int32_t n_link_dof[CARL_BRAX_MAX_LINKS];  /* 6 = free root; else n_slide prismatic dofs followed by 0..3
                                               revolute dofs turning, in order, about the joint frame's x, y,
                                               +-z axes (each carried by the preceding ones: MuJoCo's stacking of
                                               several hinges in one body).  parent -1 with n_link_dof < 6 =
                                               jointed to the static world (planar roots) */
  int32_t q_start[CARL_BRAX_MAX_LINKS], dof_start[CARL_BRAX_MAX_LINKS];
  float link_pos[CARL_BRAX_MAX_LINKS][3], link_rot[CARL_BRAX_MAX_LINKS][4];   /* child frame in parent frame at q = 0 */
  float joint_pos[CARL_BRAX_MAX_LINKS][3], joint_rot[CARL_BRAX_MAX_LINKS][4]; /* anchor / joint frame in the child frame */
  float com[CARL_BRAX_MAX_LINKS][3];        /* centre of mass in the link frame */
  float mass[CARL_BRAX_MAX_LINKS];          /* effective (spring_mass_scale applied) */
  float inv_inertia[CARL_BRAX_MAX_LINKS][3];/* effective inverse principal moments, link frame.  Every shipped model is
                                             * isotropic ([0] == [1] == [2]: brax's spring_inertia_scale = 1); a model with a
                                             * link whose three moments differ is stepped by the general kernels (the ones
                                             * the reach / push task models take: 4, 8 or 16 lanes per env), the only ones
                                             * that carry R diag(inv_inertia) R^T -- same results, lower throughput */
  float k_pos[CARL_BRAX_MAX_LINKS], k_vel[CARL_BRAX_MAX_LINKS];        /* constraint_stiffness / _vel_damping */
  float k_limit[CARL_BRAX_MAX_LINKS], k_ang_damp[CARL_BRAX_MAX_LINKS]; /* constraint_limit_stiffness / _ang_damping */
  float dof_lo[CARL_BRAX_MAX_DOF], dof_hi[CARL_BRAX_MAX_DOF];
  float dof_damping[CARL_BRAX_MAX_DOF], dof_stiffness[CARL_BRAX_MAX_DOF];
  int32_t act_dof[CARL_BRAX_MAX_ACT];
  float act_gear[CARL_BRAX_MAX_ACT], act_lo[CARL_BRAX_MAX_ACT], act_hi[CARL_BRAX_MAX_ACT];
  int32_t coll_link[CARL_BRAX_MAX_COLL];    /* collision spheres vs the plane z = plane_z (0: the ground) */
  float coll_pos[CARL_BRAX_MAX_COLL][3], coll_radius[CARL_BRAX_MAX_COLL];
  float init_q[CARL_BRAX_MAX_Q];
  /* goal-directed reward epilogue (carl/envs/brax/brax_walker_goal_wrapper.py:113-140): position +=
   * obs[goal_obs_idx] * goal_dt (the RAW MJCF timestep, Quirk B3); reward = max(0, d_prev - d_cur);
   * terminate with success when d_cur <= target_radius */
  int32_t goal_mode;
  int32_t goal_obs_idx[2];
  float goal_dt;
  int32_t n_slide[CARL_BRAX_MAX_LINKS];        /* 0..2 prismatic dofs (q order: slides, then the hinges) */
  float dof_sign3[CARL_BRAX_MAX_LINKS];        /* +1 / -1: third hinge axis = sign * (x cross y) of the joint frame */
  int32_t reset_vel_uniform;                   /* qd noise: 1 = U(-scale, scale) (humanoid), 0 = scale * N(0,1) */
  int32_t reward_on_com;                       /* forward velocity of the whole-body centre of mass (humanoid) */
  int32_t obs_extended;                        /* append com inertia (L x 10), com velocity (L x 6), qfrc_actuator */
  int32_t healthy_q_index;                     /* >= 0: the env is healthy only while q[index] is inside
                                                * [healthy_q_lo, healthy_q_hi] (hopper / walker2d torso pitch,
                                                * inverted-pendulum pole angle); -1: no such check */
  float healthy_q_lo, healthy_q_hi;
  float obs_qd_clip;                           /* > 0: velocities in the observation are clipped to +-this */
  int32_t lanes_per_env;                       /* HOST-side launch hint: lanes that share one env (rounded up
                                                * to an instantiated width, carl_brax_lane_widths); 0 = chosen
                                                * by the library from the model and the batch size */
  int32_t reward_height;                       /* 1: the "forward" term is weight * (root z) / dt_env
                                                * (humanoidstandup's uph_cost) instead of weight * dx / dt_env */
  int32_t obs_trig_from;                       /* > 0: coordinates q[obs_trig_from:] appear in the observation as
                                                * sin(q[from:]) ++ cos(q[from:]) instead of q[from:]
                                                * (inverted double pendulum); 0: plain q */
  /* tip reward / termination (brax.envs.inverted_double_pendulum): with tip = frame origin of
   * tip_link + R * tip_offset, reward = healthy_reward - (tip_x_weight * tip.x^2 + (tip.z -
   * tip_height)^2) - (tip_vel_weight[k] * qd[tip_vel_dof[k]]^2, k = 0, 1); done when tip.z <=
   * tip_min_height.  tip_link = 0: not used (the root is never the tip) */
  int32_t tip_link;
  float tip_offset[3];
  float tip_x_weight, tip_height, tip_min_height;
  float tip_vel_weight[2];
  int32_t tip_vel_dof[2];
  /* reach task (brax.envs.reacher; reference class carl/envs/brax/carl_reacher.py:9-39): target_link > 0
   * names the LAST link, jointed to the world by two slides whose coordinates q[tq : tq + 2]
   * (tq = q_start[target_link]) are the goal.  reset: q[tq], q[tq + 1] = d cos(a), d sin(a) with
   * d = target_max_dist * U, a = 2 pi U (draws n_q + n_dof and + 1 of the reset stream), goal rates 0;
   * observation = cos(q[:tq]) ++ sin(q[:tq]) ++ q[tq:] ++ qd[:dof_start[target_link]] ++ (tip - goal
   * position), tip = frame origin of tip_link + R * tip_offset; reward = -|tip - goal| -
   * ctrl_cost_weight * |a|^2; never terminates.  0: not used */
  int32_t target_link;
  float target_max_dist;
  /* push task (brax.envs.pusher; reference class carl/envs/brax/carl_pusher.py:9-103): push_link > 0 names
   * the LAST link (the object), on two slides against the world; every link before it is the arm (hinges
   * only), tip_link its end effector.  With goal = the context's goal_position_* (ctx.goal_position; rows
   * absent: push_goal):
   *   reset: arm q = init_q + reset_noise_scale U(-1, 1), arm rates U(-reset_vel_scale, reset_vel_scale);
   *          object offset c = push_lo + (push_hi - push_lo) U (draws n_q + n_dof and + 1 of the reset
   *          stream); d = link_pos[push_link].xy + c - goal.xy is stretched to length push_min_dist when
   *          shorter; object slides = goal.xy + d - link_pos.xy, at rest;
   *   observation = q[:na] ++ qd[:na] ++ COM(tip_link) ++ COM(push_link) ++ goal (na = arm dofs);
   *   reward = -|COM(object) - goal| - ctrl_cost_weight |a|^2 - push_near_weight |COM(object) - COM(tip)|;
   *   never terminates.  0: not used */
  int32_t push_link;
  float push_goal[3];
  float push_near_weight, push_min_dist;
  float push_lo[2], push_hi[2];
  /* link-pair contact of the push task: n_pair spheres fixed to pair_link (centre pair_pos[k] in its
   * frame, radius pair_radius[k]) against the object, an upright cylinder (radius pair_obj_radius, half
   * height pair_obj_half) centred at the frame origin of push_link.  A sphere whose centre is within
   * pair_obj_half + r_k of the object's mid-height and closer than r_k + R in the horizontal plane pushes
   * the object along the horizontal normal n with fm = max(0, pair_k depth + pair_c closing speed); the
   * opposite force acts on pair_link at the sphere's centre (the object's own rotation is locked by its
   * joint).  ABI 8 -- contact friction and the table:
   *   pair_ct > 0: Coulomb friction in the pair contact, regularised -- with vt the HORIZONTAL part of the sphere's
   *     velocity relative to the object minus its normal part, the object is dragged along vt by min(pair_ct |vt|, friction fm)
   *     (`friction`: the env's context value, as in the plane contacts) and the sphere held back by the same;
   *   obj_support = 1: the object rests on the plane z = plane_z (its joint has no vertical freedom, so the
   *     plane carries its weight): every substep, after the force update, its horizontal velocity shrinks by
   *     min(friction |gravity_z| dt, |v_h|) -- Coulomb friction under the normal load m |g| as an impulse, the
   *     form the sphere / plane contacts use (a sliding puck stops dead after v0^2 / (2 mu g));
   *   plane_z: the height of THE collision plane (0 for the locomotion models; the push task's table): every
   *     sphere of coll_* collides with z = plane_z. */
  int32_t n_pair, pair_link;
  float pair_pos[CARL_BRAX_MAX_PAIR][3], pair_radius[CARL_BRAX_MAX_PAIR];
  float pair_obj_radius, pair_obj_half, pair_k, pair_c;
  float pair_ct, plane_z;
  int32_t obj_support, reserved2;
  float slide_axis[CARL_BRAX_MAX_LINKS][2][3]; /* unit axes in the PARENT frame, mutually orthogonal */
  carl_brax_ctx_map_t ctx;
} carl_brax_sys_t;

/* persistent state per link: COM position 3, rotation 4 (w,x,y,z), linear velocity 3, angular velocity 3
 * (world frame) = 13 quantities.  The pose (the first 7) is carried to 48 significant bits as a float32 head
 * plus a float32 tail -- the spring pipeline multiplies pose DIFFERENCES by k dt / m = 20 .. 50 per substep, so a
 * pose rounded to float32 between steps costs 1e-5 of velocity per env step -- which makes
 * CARL_BRAX_LINK_RECORD = 20 floats per link in HBM.  One env's record = its L links' 80-byte records one after the
 * other (ABI 8; ABI 5 - 7 kept three blocks per env: heads, tails, velocities), a link's record being
 *   [0, 7)    pose head (p.x p.y p.z r.w r.x r.y r.z): the pose to float32
 *   [7, 14)   pose tail: pose = (double)head + (double)tail
 *   [14, 20)  velocities (v.x v.y v.z w.x w.y w.z)
 * -- the lane that owns a link moves it as five 16-byte pieces, a wavefront's envs x links one contiguous run. */
#define CARL_BRAX_LINK_STATE 13
#define CARL_BRAX_LINK_RECORD 20

/* carl_batch_t is reused: family = CARL_N_FAMILIES + env_kind is ignored here (sys decides),
 * state is [n_lanes][n_links][CARL_BRAX_LINK_RECORD] (env-major, then link-major: the lanes that share an env move
 * its 20 L floats as one contiguous run; the classic-control families keep [S][n_lanes]), ctx_table rows follow the CARL
 * class's feature table.
 * action is float32 [n_lanes][n_act] (lane-major, like obs).  `sys` is a DEVICE pointer to
 * one carl_brax_sys_t.  */
int carl_brax_reset(const carl_batch_t* batch, const carl_brax_sys_t* sys_dev, const carl_brax_sys_t* sys_host,
                    const uint8_t* mask, float* obs, void* stream);
int carl_brax_step(const carl_batch_t* batch, const carl_brax_sys_t* sys_dev, const carl_brax_sys_t* sys_host,
                   const carl_step_io_t* io, void* stream);
int carl_brax_rollout(const carl_batch_t* batch, const carl_brax_sys_t* sys_dev, const carl_brax_sys_t* sys_host,
                      const carl_step_io_t* io, int32_t n_steps, void* stream);
/* the lane-group widths (values for sys.lanes_per_env) this library can launch for STEP / ROLLOUT launches of the model
 * in a batch with these carl_batch_t::flags (ABI 9: CARL_FLAG_BRAX_GENERIC moves a planar model to the general kernels,
 * whose instantiated widths differ), ascending; returns their number (<= cap).  Results do not depend on the width:
 * it is a pure scheduling choice (carl_amd.brax_engine.BraxVecEngine.autotune times them on the real batch).  No
 * reference counterpart (brax vmaps one env per array row, carl/envs/brax/carl_brax_env.py:163-167). */
int carl_brax_lane_widths(const carl_brax_sys_t* sys_host, uint32_t batch_flags, int32_t* widths_out, int32_t cap);
/* 1 when step / rollout launches of this model take the planar substep (root on two world slides x, z and a hinge
 * about y, every other link on a hinge about +-y, all geometry in the y = 0 plane: Halfcheetah, Hopper, Walker2d as
 * carl_amd.envs.brax.models builds them) unless the batch carries CARL_FLAG_BRAX_GENERIC; 0 otherwise.  The planar
 * substep neither reads nor updates out-of-plane state components, so a caller that writes states itself asks here
 * (carl_amd.brax_engine.BraxVecEngine.set_state64 does, and falls back to the general substep).  No reference
 * counterpart: brax's spring pipeline has one code path (carl/envs/brax/carl_brax_env.py:163-176). */
int carl_brax_model_is_planar(const carl_brax_sys_t* sys_host);
/* How a step / rollout launch of a batch larger than the chip holds at once is divided (no reference counterpart: the
 * reference steps its envs in vmapped lock-step, carl/envs/brax/carl_brax_env.py:163-190).  A "group" is one
 * wavefront's worth of envs; workgroup w of n_workgroups owns a contiguous share of the n_groups groups and cuts its
 * groups x n_steps group-steps into one contiguous piece per wavefront.  Writes, for (workgroup, wave), the fragments
 * in the order the wavefront runs them -- five int32 each: global group, first step, one-past-last step, waits for the
 * previous wavefront's hand-over (0/1), hands over to the next wavefront (0/1) -- and returns their number (-1: bad
 * argument).  Pure integer arithmetic, no device access: it is the same code the kernel runs (tests/test_abi.py checks
 * that every group-step is run exactly once, in step order, and that no hand-over can wait on a later one). */
int carl_brax_fragment_plan(int32_t n_groups, int32_t n_workgroups, int32_t waves_per_workgroup, int32_t n_steps,
                            int32_t workgroup, int32_t wave, int32_t* fragments_out, int32_t cap);

/* ======================= context sets on the device =======================
 * Replaces ContextSampler.sample_contexts (carl/context/sampler.py:45-61: per-feature draws from
 * the distributions, defaults filled for every other feature) and ContextSpace.verify_context
 * (carl/context/context_space.py:54-59) for dense context sets: the [F][C] table the step kernels
 * read is produced (and bounds-checked) in HBM instead of as C Python dicts.
 * Draws are Philox4x32-10 keyed (seed; global context id, feature index, attempt): a pure function
 * of (seed, context id, feature) -- independent of sharding -- and NOT the reference's NumPy
 * MT19937 stream (carl_amd.context.sampler reproduces that one on the host, pinned by the
 * reference's notebook outputs; this entry point is the one that scales).
 *   CONSTANT       value                                   (features without a distribution: default)
 *   UNIFORM_FLOAT  fma(upper - lower, u, lower); log_scale: exp of the same between the logs
 *   NORMAL_FLOAT   mu + sigma * z (Box-Muller), redrawn while outside [lower, upper] (<= 32 times,
 *                  then clipped) -- ConfigSpace NormalFloat with bounds
 *   UNIFORM_INT    lower + floor(u * (upper - lower + 1))
 *   CATEGORICAL    choices[floor(u * n_choices)] (numeric choices, e.g. Brax target_direction) */
#define CARL_MAX_CHOICES 32
typedef enum {
  CARL_FEAT_CONSTANT = 0,
  CARL_FEAT_UNIFORM_FLOAT = 1,
  CARL_FEAT_NORMAL_FLOAT = 2,
  CARL_FEAT_UNIFORM_INT = 3,
  CARL_FEAT_CATEGORICAL = 4
} carl_feature_kind_t;

typedef struct {
  int32_t kind;       /* carl_feature_kind_t */
  int32_t n_choices;  /* CATEGORICAL */
  int32_t log_scale;  /* UNIFORM_FLOAT */
  int32_t reserved;
  float lower, upper; /* bounds: sampling range and carl_verify_contexts (+-inf allowed) */
  float mu, sigma;    /* NORMAL_FLOAT */
  float value;        /* CONSTANT */
  float reserved_f;
  float choices[CARL_MAX_CHOICES];
} carl_feature_spec_t;

/* ctx_table[f * ctx_stride + c] for c in [0, n_contexts), f in [0, n_features); context c is the
 * global context `context_offset + c`.  specs_dev / specs_host: the same n_features specs on the
 * device and on the host (validation). */
int carl_sample_contexts(const carl_feature_spec_t* specs_dev, const carl_feature_spec_t* specs_host,
                         int32_t n_features, int32_t n_contexts, int32_t ctx_stride, int64_t context_offset,
                         uint64_t seed, float* ctx_table, void* stream);
/* n_bad_out[0] (device int32) = number of table entries outside their feature's bounds (or not one
 * of a categorical feature's choices, or NaN) */
int carl_verify_contexts(const carl_feature_spec_t* specs_dev, const carl_feature_spec_t* specs_host,
                         int32_t n_features, int32_t n_contexts, int32_t ctx_stride, const float* ctx_table,
                         int32_t* n_bad_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARL_AMD_H_ */

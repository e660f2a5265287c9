/*
 * madrl_hip.h -- C ABI of libmadrl_hip.so, the MI355X (gfx950) batched rollout
 * engine for the sisl/MADRL environments.
 *
 * This is the drop-in boundary.  The reference has no FFI of its
 * own -- it is pure Python -- so every entry point below replaces a *Python method*
 * of the reference; the binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md (madrl_amd/_lib.py is that stub).  All buffers are plain device
 * pointers (PyTorch tensors are only the transport), all sizes are explicit, no
 * C++/torch types cross the boundary, nothing throws.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MADRL_E* code;
 *     madrl_last_error() returns a thread-local message for the last failure;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work
 *     is enqueued asynchronously, no host synchronisation inside reset/step;
 *   - the caller owns every I/O and state buffer; the library owns only a small
 *     host handle and a few KB of read-only device tables created in *_create;
 *     no allocation happens in reset/step (graph-capture safe);
 *   - a handle is not thread-safe; one handle per (process, device).
 */
#ifndef MADRL_HIP_H
#define MADRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MADRL_ABI_VERSION 7

#define MADRL_OK 0
#define MADRL_EINVAL (-1)   /* bad argument / unsupported configuration */
#define MADRL_EHIP (-2)     /* a HIP runtime call failed */
#define MADRL_ENOMEM (-3)

int madrl_abi_version(void);
const char *madrl_last_error(void);

/* ------------------------------------------------------------------------------------------
 * PursuitEvade  (reference: madrl_environments/pursuit/pursuit_evade.py)
 * ---------------------------------------------------------------------------------------- */

/* Constructor kwargs of PursuitEvade.__init__ (pursuit_evade.py:49-148) plus the batching
 * parameters.  Field names follow the reference kwargs. */
typedef struct madrl_pursuit_config {
    int32_t struct_size;   /* = sizeof(madrl_pursuit_config), ABI check */
    int32_t xs, ys;        /* map_matrix.shape (:54) */
    int32_t n_pursuers;    /* :61 */
    int32_t n_evaders;     /* :60 */
    int32_t obs_range;     /* :63 */
    int32_t n_catch;       /* :79 */
    int32_t surround;      /* :142 */
    int32_t flatten;       /* :67  1: (3R^2[+1]) per agent, 0: (R,R,4) per agent */
    int32_t include_id;    /* :100 */
    int32_t reward_global; /* reward_mech == 'global' (:58, :260-261) */
    int32_t sample_maps;   /* :49, :182-183 */
    int32_t n_maps;        /* len(map_pool) */
    int32_t max_steps;     /* 0 = none; else done bit1 is raised when the episode reaches it
                              (the sampler's max_path_length, runners/__init__.py:88) */
    int32_t auto_reset;    /* 1: an env whose step ends with done != 0 is reset inside the
                              same launch and its obs row holds the new episode's first obs */
    int32_t max_opponents; /* 0: fixed n_evaders; > 0: random_opponents with train_pursuit (:177-181): every reset creates
                              randint(1, max_opponents) evaders (at most n_evaders), the other slots count as gone */
    int32_t control_evaders; /* = not train_pursuit (:105).  0 (a zero-initialised struct keeps the reference's default,
                              train_pursuit=True): the actions drive the pursuers.  1: evader control (:204-207,
                              :215-224) -- action k moves the k-th REMAINING evader (layer order; n_pursuers actions per env, as
                              many as env.agents has entries), every pursuer moves by one pursuer_controller.act() draw
                              (in-kernel Philox, or entry j of the injected array, then [n_envs][n_pursuers]); observation row
                              k shows the window of the k-th remaining evader among slots 0..n_pursuers-1 with id k/n_pursuers
                              (collect_obs walks range(n_agents()) = range(n_pursuers), :418-428), rows past the last such
                              evader are left untouched; rewards stay the pursuers' (:213, :254-256).  Needs
                              n_evaders >= n_pursuers, no random_opponents; shapes of up to 64 agents listed in
                              pursuit_specializations.def run on the one-wavefront kernel, everything else on the generic one. */
    int32_t reserved0;
    double catchr;            /* :92 */
    double term_pursuit;      /* :95 */
    double urgency_reward;    /* :98 */
    double layer_norm;        /* :77 */
    double constraint_window; /* :144 */
    uint64_t seed;            /* Philox key (DESIGN.md "RNG contract") */
    int64_t env_id_base;      /* global index of env 0 of this shard (multi-GPU: rank*n_envs) */
} madrl_pursuit_config;

typedef struct madrl_pursuit madrl_pursuit; /* opaque */

/* Observation length per agent: 3R^2 (+1 with id) if flatten, else 4R^2
 * (DiscreteAgent._obs_shape, utils/DiscreteAgent.py:50-53). */
int madrl_pursuit_obs_dim(const madrl_pursuit_config *cfg, int32_t *out_dim);

/* Bytes of per-env state the caller must allocate (and zero: an all-zero state is the
 * reference's constructor state -- every agent at (0,0) on map 0, pursuit_evade.py:69-75).
 * Layout: n_envs packed records of madrl_pursuit_record_bytes() each, padding to 256 bytes, then -- for shapes with a
 * compiled fast path -- the stale-zero masks of that path (256 bytes per env, wavefront and mask word, see obs_dev below), then the
 * flag plane (one dword per env, madrl_pursuit_flags_offset).
 * Everything the library knows about an env lives in this buffer and in the observation buffer: copying both clones it. */
int madrl_pursuit_state_bytes(const madrl_pursuit_config *cfg, int64_t n_envs, uint64_t *out_bytes);
/* Byte offset, inside the state buffer, of the FLAG PLANE: uint32 [n_envs], written by every step launch next to done_dev.
 * Word n = the done byte of env n split into one 0 / 1 byte per meaning (little endian): byte 0 = bit0 (is_terminal, the `done`
 * PursuitEvade.step returns, pursuit_evade.py:259), byte 1 = bit1 (max_steps reached), byte 2 = bit7 (count overflow), byte 3 = the
 * done byte itself.  A host binding hands these bytes out as strided bool arrays (madrl_amd/pursuit.py: the `done` and `info` tensors
 * of step() are views of this plane) instead of launching kernels that mask bits of done_dev after every step. */
int madrl_pursuit_flags_offset(const madrl_pursuit_config *cfg, int64_t n_envs, uint64_t *out_offset);
/* bytes of one packed record: [u32 tick][u32 t][u32 map_id][u32 0][u8 x,y per agent, pursuers first][u32 gone bits][u32 terminal bits], 16-B aligned */
int madrl_pursuit_record_bytes(const madrl_pursuit_config *cfg, int32_t *out_bytes);

/* Replaces PursuitEvade.__init__ for n_envs instances.  map_pool_host: n_maps*xs*ys int8 on
 * the HOST, row-major [map][x][y], 0 = free, -1 = building (utils/TwoDMaps.py:8-22).
 * state_dev: device buffer of madrl_pursuit_state_bytes() bytes, zero-filled by the caller. */
int madrl_pursuit_create(const madrl_pursuit_config *cfg, const int8_t *map_pool_host,
                         int64_t n_envs, int32_t device, void *state_dev, madrl_pursuit **out);
void madrl_pursuit_destroy(madrl_pursuit *h);

/* Kernel selection.  Two implementations share the packed state and produce identical results:
 * GENERIC (any configuration, one workgroup of `threads` lanes per env) and WAVE (one wavefront
 * per env, compile-time specialised; only for the shapes listed in
 * madrl_amd/csrc/pursuit_specializations.def).  AUTO = WAVE when available. */
#define MADRL_KERNEL_AUTO 0
#define MADRL_KERNEL_GENERIC 1
#define MADRL_KERNEL_WAVE 2
int madrl_pursuit_set_kernel(madrl_pursuit *h, int32_t kind);
int madrl_pursuit_kernel_kind(const madrl_pursuit *h, int32_t *out_kind);

/* Launch shape: threads per workgroup (GENERIC kernel only; multiple of 64, 0 = heuristic)
 * and the maximum number of workgroups (0 = default: one per env for GENERIC, 5120 persistent
 * workgroups for WAVE); workgroups stride over envs. */
int madrl_pursuit_set_launch(madrl_pursuit *h, int32_t threads, int64_t max_blocks);
/* Order in which successive step launches of the fast path walk the env range: 0 = automatic (forward; alternating directions
 * once a launch moves more than ~375 MB, so that the rows written last are the first ones touched again while they are still in
 * the memory-side cache), 1 = always alternate, 2 = always forward.  Results do not depend on it. */
int madrl_pursuit_set_walk(madrl_pursuit *h, int32_t mode);

/* Replaces PursuitEvade.reset (pursuit_evade.py:173-207) for every env with mask[n] != 0
 * (mask_dev NULL = all).
 *   inj_pos_dev  int32 [n_envs][P+E][2] or NULL: positions used instead of the rejection
 *                sampler (agent_utils.py:31-47) -- parity harness hook, pursuers first; an evader
 *                entry with x < 0 is not created (random_opponents);
 *   inj_map_dev  int32 [n_envs] or NULL: map index used instead of the sample_maps draw;
 *   obs_dev      float32 [n_envs][P][obs_dim], IN/OUT: this buffer is the reference's
 *                persistent `local_obs` (pursuit_evade.py:119-120).  Cells of channels 1-2
 *                that fall outside the map are left untouched, exactly as :438-439 leaves
 *                them; pass the same zero-initialised buffer to every
 *                reset/step call of a handle to get the reference's values.
 *                CONTRACT of the fast path: it remembers (stale-zero masks, in the state buffer) which untouched cells of
 *                THIS buffer hold 0.0 and stores whole float4s over them.  The memory is forgotten automatically when
 *                another pointer is passed, after madrl_pursuit_set_state and after a generic-kernel launch; a caller
 *                that WRITES into the buffer itself (or frees and re-allocates it at the same address) must call
 *                madrl_pursuit_invalidate_obs() before the next reset/step. */
int madrl_pursuit_reset(madrl_pursuit *h, const uint8_t *mask_dev, const int32_t *inj_pos_dev,
                        const int32_t *inj_map_dev, float *obs_dev, void *stream);

/* Replaces PursuitEvade.step (pursuit_evade.py:209-262) for all envs.
 *   actions_dev  int32 [n_envs][P], values 0..4 (utils/DiscreteAgent.py:28-38); anything
 *                else is treated as 4 (stay) -- the reference raises IndexError instead;
 *   inj_evader_actions_dev  int32 [n_envs][E] or NULL: entry k is the action of the k-th
 *                REMAINING evader in layer order (one RandomPolicy.act per remaining evader,
 *                pursuit_evade.py:238-241); NULL = in-kernel Philox draws.  (control_evaders = 1: the
 *                opponents are the pursuers, the array is [n_envs][P], entry j = pursuer j.)
 *   obs_dev      as in reset (IN/OUT);
 *   rew_dev      float32 [n_envs][P]  (computed in float64 like the reference, then rounded);
 *   done_dev     uint8 [n_envs]: bit0 = is_terminal (:383-389), bit1 = max_steps reached; bit7 = count overflow: the env has had more
 *                than 253 agents of one kind on ONE cell since its last reset (the kernel counts per cell in bytes; possible only with
 *                more than 253 pursuers or evaders, up to 1 023 of each are accepted) -- its results are void until it is reset;
 *   removed_dev  int32 [n_envs]: info['removed'] (:261-262). */
int madrl_pursuit_step(madrl_pursuit *h, const int32_t *actions_dev,
                       const int32_t *inj_evader_actions_dev, float *obs_dev, float *rew_dev,
                       uint8_t *done_dev, int32_t *removed_dev, void *stream);

/* One batch stepped as n_shards independent sub-batches, each a handle of its own (created with env_id_base advanced by its offset)
 * on its own HIP stream: ONE host call issues every launch.  Env instances never interact (the reference's own parallelism is N
 * pickled env copies in sampler workers, runners/rurllab.py:259), so nothing orders sub-batch A's step t + 1 against sub-batch
 * B's step t: one launch's drain overlaps the other's ramp-up.  fork != 0: every sub-batch stream first waits for what
 * caller_stream holds so far (the actions); join != 0: caller_stream then waits for every sub-batch (before it reads their
 * observations / rewards).  Both as events recorded and waited for inside this call -- the five event calls per step a Python
 * caller would make (madrl_amd/sharded.py).  A sampler that drives each sub-batch from its own stream passes 0 / 0.
 * io[j]: the arguments of madrl_pursuit_step for sub-batch j and the stream it runs on. */
typedef struct madrl_pursuit_shard_io {
    const int32_t *actions;
    const int32_t *inj_evader_actions;   /* or NULL */
    float *obs;
    float *rew;
    uint8_t *done;
    int32_t *removed;
    void *stream;
} madrl_pursuit_shard_io;
int madrl_pursuit_step_sharded(madrl_pursuit *const *hs, const madrl_pursuit_shard_io *io, int32_t n_shards, void *caller_stream,
                               int32_t fork, int32_t join);

/* Unpacked view of the env state (device pointers; any may be NULL to skip).  This is the
 * checkpoint / parity-injection hook (the reference pokes AgentLayer.set_position,
 * pursuit/test_pursuit.py:22-51).  Evaders are in SLOT order; a removed evader has
 * gone = 1 and position (-1,-1) on get.
 *   pos_p int32 [N][P][2], pos_e int32 [N][E][2], gone uint8 [N][E],
 *   term_p uint8 [N][P], term_e uint8 [N][E]  (DiscreteAgent.terminal, DiscreteAgent.py:45),
 *   map_id int32 [N], tick uint32 [N] (RNG draw counter), t int32 [N] (episode step). */
int madrl_pursuit_get_state(madrl_pursuit *h, int32_t *pos_p, int32_t *pos_e, uint8_t *gone,
                            uint8_t *term_p, uint8_t *term_e, int32_t *map_id, uint32_t *tick,
                            int32_t *t, void *stream);
int madrl_pursuit_set_state(madrl_pursuit *h, const int32_t *pos_p, const int32_t *pos_e,
                            const uint8_t *gone, const uint8_t *term_p, const uint8_t *term_e,
                            const int32_t *map_id, const uint32_t *tick, const int32_t *t,
                            void *stream);
/* The caller modified the observation buffer behind the library's back (see obs_dev above): forget what is known about it. */
int madrl_pursuit_invalidate_obs(madrl_pursuit *h);
/* The opposite promise: every element of obs_dev holds +0.0f right now (a freshly zeroed buffer, which is what the reference's
 * local_obs starts as, pursuit_evade.py:119-120).  The fast path then knows that no cell outside the map can hold a stale value that
 * needs protecting and stores whole 16-byte words from the first step on; without the promise a cell counts as "unknown" until it has
 * been inside the map once.  Results are the same either way -- provided the promise is true. */
int madrl_pursuit_declare_obs_zero(madrl_pursuit *h, const float *obs_dev, void *stream);

/* Curriculum (PursuitEvade.update_curriculum, pursuit_evade.py:264-272; set_param_values, madrl_environments/__init__.py:64-67)
 * without re-creating the handle.  set_params replaces the batch-wide catchr / constraint_window of the config;
 * set_curriculum binds PER-ENV values: float64 [n_envs] device arrays (caller-owned, valid while bound, may be rewritten
 * between launches), NULL = the batch-wide scalar.  constraint_window is read by resets, catchr by every step. */
int madrl_pursuit_set_params(madrl_pursuit *h, double catchr, double constraint_window);
int madrl_pursuit_set_curriculum(madrl_pursuit *h, const double *constraint_window_dev, const double *catchr_dev);

/* ------------------------------------------------------------------------------------------
 * MAWaterWorld  (reference: madrl_environments/pursuit/waterworld.py), float32 arithmetic
 * ---------------------------------------------------------------------------------------- */

/* Constructor arguments of MAWaterWorld.__init__ (waterworld.py:77-118). */
typedef struct madrl_waterworld_config {
    int32_t struct_size;     /* = sizeof(madrl_waterworld_config) */
    int32_t n_pursuers, n_evaders, n_coop, n_poison, n_sensors;   /* :77-81 */
    int32_t addid, speed_features;                                 /* :81 */
    int32_t reward_global;   /* reward_mech == 'global' */
    int32_t obstacle_fixed;  /* 1: obstacle at obstacle_loc; 0: obstacle_loc=None, random per reset (:147-151) */
    int32_t max_steps;       /* 0 = the reference's timestep_limit of 1000 (:124-126) */
    int32_t auto_reset;      /* 1: an env whose step ends with done is reset in the same launch */
    int32_t reserved0;
    double radius, obstacle_radius, ev_speed, poison_speed, sensor_range, action_scale;
    double poison_reward, food_reward, encounter_reward, control_penalty;
    double obstacle_loc[2];
    uint64_t seed;
    int64_t env_id_base;
} madrl_waterworld_config;

typedef struct madrl_waterworld madrl_waterworld; /* opaque */

/* n_sensors * (7 or 4) + 2 + (1 if addid)  (Archea.__init__, waterworld.py:18-24) */
int madrl_waterworld_obs_dim(const madrl_waterworld_config *cfg, int32_t *out_dim);
/* packed state per env: float32 pos[NP][2] vel[NP][2] obstacle[2], int32 t, uint32 tick
 * (NP = pursuers + evaders + poisons, in that order) */
int madrl_waterworld_state_bytes(const madrl_waterworld_config *cfg, int64_t n_envs, uint64_t *out_bytes);
/* sensors_host: float64 [n_sensors][2] unit vectors, np.c_[cos, sin] of linspace(0, 2pi, K+1)[:-1]
 * (Archea.__init__ :29-31), rounded to float32 once by the library. */
int madrl_waterworld_create(const madrl_waterworld_config *cfg, const double *sensors_host, int64_t n_envs,
                            int32_t device, void *state_dev, madrl_waterworld **out);
void madrl_waterworld_destroy(madrl_waterworld *h);
int madrl_waterworld_set_launch(madrl_waterworld *h, int64_t max_blocks);

/* Fused StandardizedEnv (madrl_environments/__init__.py:204-311): bind the wrapper's state to the env handle and the step /
 * reset kernels normalise the observation row as it leaves LDS (and the rewards as they are produced) instead of storing it
 * raw for a second launch (madrl_wrap_obsnorm / _rewnorm) to read back: 36 instead of 44 bytes of HBM traffic per observation
 * element, one launch instead of three.  Same arithmetic (float64 exponential running mean / variance per env, agent and
 * element, :242-271).  All pointers are device memory that stays valid while bound; running statistics start at mean 0 /
 * var 1 (:229-232).  While bound, obs_dev of reset / step may be NULL (the raw row is then not stored).  args NULL unbinds. */
typedef struct madrl_standardize_args {
    int32_t struct_size, enable_obsnorm, enable_rewnorm, reserved0;
    double obs_alpha, rew_alpha, eps, scale_reward;
    double *obs_mean, *obs_var;   /* [N][A][D] */
    float *obs_out;               /* [N][A][D] */
    double *rew_mean, *rew_var;   /* [N][A] */
    float *rew_out;               /* [N][A] or NULL: rewards are not touched */
} madrl_standardize_args;
int madrl_waterworld_set_standardize(madrl_waterworld *h, const madrl_standardize_args *args);

/* MAWaterWorld.reset (:144-172) incl. its trailing zero-action step; obs float32 [N][Np][obs_dim]. */
int madrl_waterworld_reset(madrl_waterworld *h, const uint8_t *mask_dev, float *obs_dev, void *stream);
/* MAWaterWorld.step (:220-436).
 *   actions_dev       float32 [N][Np][2]  (any layout that reshapes to it, :221-222)
 *   inj_respawn_dev   float32 [N][NP][4] or NULL: for a particle caught in this step, (x, y) is the
 *                     accepted respawn position and (u0, u1) the two velocity uniforms -- parity hook
 *                     replacing the np_random draws of :355-374; NULL = in-kernel Philox
 *   rew_dev float32 [N][Np]; done_dev uint8 [N] (:174-178); info_dev int32 [N][2] = evcatches, pocatches */
int madrl_waterworld_step(madrl_waterworld *h, const float *actions_dev, const float *inj_respawn_dev,
                          float *obs_dev, float *rew_dev, uint8_t *done_dev, int32_t *info_dev, void *stream);
/* teacher-forcing / checkpoint hook: float32 pos [N][NP][2], vel [N][NP][2], obst [N][2], int32 t [N],
 * uint32 tick [N]; any pointer may be NULL */
int madrl_waterworld_get_state(madrl_waterworld *h, float *pos, float *vel, float *obst, int32_t *t,
                               uint32_t *tick, void *stream);
int madrl_waterworld_set_state(madrl_waterworld *h, const float *pos, const float *vel, const float *obst,
                               const int32_t *t, const uint32_t *tick, void *stream);

/* ------------------------------------------------------------------------------------------
 * ContinuousHostageWorld (reference: madrl_environments/hostage.py).  One wavefront per env, float32.
 * ---------------------------------------------------------------------------------------- */
/* Constructor arguments of ContinuousHostageWorld.__init__ (hostage.py:76-81). */
typedef struct madrl_hostage_config {
    int32_t struct_size;     /* = sizeof(madrl_hostage_config) */
    int32_t n_good, n_hostages, n_bad, n_coop_save, n_coop_avoid, n_sensors;
    int32_t addid;
    int32_t reward_global;   /* reward_mech == 'global' (the reference's default here) */
    int32_t key_fixed;       /* 1: key at key_loc; 0: key_loc=None, sampled by the first reset of an env's life (:143-146) */
    int32_t max_steps;       /* 0 = the reference's timestep_limit of 1000 (:118-120) */
    int32_t auto_reset;      /* 1: an env whose step ends with done is reset in the same launch */
    double radius, bad_speed, sensor_range, action_scale;
    double save_reward, hit_reward, encounter_reward, not_saved_reward, bomb_reward, bomb_radius, key_radius, control_penalty;
    double key_loc[2];
    uint64_t seed;
    int64_t env_id_base;
} madrl_hostage_config;

typedef struct madrl_hostage madrl_hostage; /* opaque */

/* n_sensors * 5 + 5 + (1 if addid)  (CircAgent.__init__, hostage.py:17-23) */
int madrl_hostage_obs_dim(const madrl_hostage_config *cfg, int32_t *out_dim);
/* packed state per env: float32 pos[NP][2] vel[NP][2] key[2] bomb[2], uint32 saved_lo, saved_hi, flags (bit0 gate open,
 * bit1 bombed, bit2 key sampled), int32 t, uint32 tick  (NP = rescuers + hostages + criminals, in that order) */
int madrl_hostage_state_bytes(const madrl_hostage_config *cfg, int64_t n_envs, uint64_t *out_bytes);
/* sensors_host: float64 [n_sensors][2] unit vectors (CircAgent.__init__ :27-29), rounded to float32 once by the library */
int madrl_hostage_create(const madrl_hostage_config *cfg, const double *sensors_host, int64_t n_envs, int32_t device,
                         void *state_dev, madrl_hostage **out);
void madrl_hostage_destroy(madrl_hostage *h);
int madrl_hostage_set_launch(madrl_hostage *h, int64_t max_blocks);
/* ContinuousHostageWorld.reset (:137-177) incl. its trailing zero-action step; obs float32 [N][n_good][obs_dim] */
int madrl_hostage_reset(madrl_hostage *h, const uint8_t *mask_dev, float *obs_dev, void *stream);
/* ContinuousHostageWorld.step (:228-430).  actions float32 [N][n_good][2]; inj_respawn_dev float32 [N][n_bad][4] or NULL:
 * the four uniforms (x, y, u_vx, u_vy) of a criminal respawned in this step (parity hook replacing np_random, :371-374);
 * rew float32 [N][n_good]; done uint8 [N] (:179-182); info int32 [N][2] = ho_saved, cr_encs */
int madrl_hostage_step(madrl_hostage *h, const float *actions_dev, const float *inj_respawn_dev, float *obs_dev,
                       float *rew_dev, uint8_t *done_dev, int32_t *info_dev, void *stream);
/* teacher-forcing / checkpoint hook; any pointer may be NULL */
int madrl_hostage_get_state(madrl_hostage *h, float *pos, float *vel, float *key, float *bomb, uint64_t *saved,
                            uint8_t *flags, int32_t *t, uint32_t *tick, void *stream);
int madrl_hostage_set_state(madrl_hostage *h, const float *pos, const float *vel, const float *key, const float *bomb,
                            const uint64_t *saved, const uint8_t *flags, const int32_t *t, const uint32_t *tick,
                            void *stream);

/* ------------------------------------------------------------------------------------------
 * MultiWalkerEnv  (reference: madrl_environments/walker/multi_walker.py), float32.
 * The rigid-body dynamics the reference delegates to Box2D (`world.Step(1/50, 180, 60)`,
 * multi_walker.py:365) are restated from scratch in Box2D 2.3.0's own order (islands by depth-first search, contact list
 * order, sleeping, fat-AABB broad phase, continuous pass); parity with Box2D itself is UNPINNED (DESIGN.md "MultiWalker"):
 * the checker is an independent plain-C restatement (oracle/multiwalker_ref.c), not the library.  The env layer around the
 * dynamics -- the world reset constructs, apply_action (:194-203), get_observation (:205-237), ContactDetector (:50-84), the
 * step tail (:369-428) -- IS pinned: to recordings of the unmodified module (tests/test_multiwalker_envlayer.py).
 * ---------------------------------------------------------------------------------------- */

/* Constructor arguments of MultiWalkerEnv.__init__ (multi_walker.py:256-270). */
typedef struct madrl_multiwalker_config {
    int32_t struct_size;      /* = sizeof(madrl_multiwalker_config) */
    int32_t n_walkers;        /* 1..10: the reference's curriculum runs 2 .. 10 (lessons/multiwalker/env.yaml:1-27); the package and the
                                 terrain grow with it (multi_walker.py:293-301) */
    int32_t reward_global;    /* reward_mech != 'local' (:426-428) */
    int32_t terminate_on_fall;
    int32_t one_hot;          /* 1: the id is np.eye(MAX_AGENTS = 40)[i] instead of i / n_walkers (:397-400): obs_dim 71 */
    int32_t max_steps;        /* 0 = none; else done bit1 when the episode reaches it */
    int32_t auto_reset;
    int32_t discrete_only;    /* 0 (default): Box2D's continuous pass (b2World::SolveTOI, continuousPhysics = true, the b2World default
                                 the reference runs with) follows every discrete solve; 1: b2World.continuousPhysics = False */
    int32_t polygon_revision; /* which b2CollidePolygons the hull / package pairs go through.  0 (default): Box2D 2.3.0 -- hill-climbing
                                 b2FindMaxSeparation from the edge that faces the other centroid, reference face by the 0.98 / 0.001
                                 hysteresis.  1: later 2.3.x revisions -- every edge normal against the deepest vertex, poly1 = B only beyond
                                 0.1 * b2_linearSlop.  The revision the authors' pybox2d wrapped is not recorded in the reference tree; the two
                                 differ on about 1 - 4 % of env-steps.  Both restatements implement both. */
    int32_t reserved0;        /* 0 */
    double position_noise, angle_noise, forward_reward, fall_reward, drop_reward;
    uint64_t seed;
    int64_t env_id_base;
} madrl_multiwalker_config;

typedef struct madrl_multiwalker madrl_multiwalker; /* opaque */

int madrl_multiwalker_obs_dim(const madrl_multiwalker_config *cfg, int32_t *out_dim);        /* 32 (:243) */
/* two records per env (the live world + the next episode prepared ahead of time, each followed by the step's scratch: see
 * madrl_multiwalker_record_bytes), the prepared episode's first observation, and a few bytes of bookkeeping per env; caller
 * allocates + zeroes, create() initialises the bookkeeping */
int madrl_multiwalker_state_bytes(const madrl_multiwalker_config *cfg, int64_t n_envs, uint64_t *out_bytes);
int madrl_multiwalker_create(const madrl_multiwalker_config *cfg, int64_t n_envs, int32_t device, void *state_dev,
                             madrl_multiwalker **out);
void madrl_multiwalker_destroy(madrl_multiwalker *h);
/* How a step is issued.  fused: 0 (default) = three launches per b2World::Step (collide | solve | continuous pass + observe), 1 = one
 * launch (9 and 10 walkers always take three: the one-launch kernel is not built for the sixteen-lane class, multiwalker_impl.hpp).  use_spares: 1 (default) = an env whose episode ends takes the next episode prepared ahead of time, 0 = every auto-reset
 * runs the reset + trailing step in a second pass.  Results do not depend on either (tests/test_multiwalker_gpu.py). */
int madrl_multiwalker_set_mode(madrl_multiwalker *h, int32_t fused, int32_t use_spares);
int madrl_multiwalker_dims(const madrl_multiwalker *h, int32_t *n_bodies, int32_t *n_terrain);
/* The kernels exist in three capacity classes -- room for 4 / 8 / 10 walkers, an env spread over 4 / 8 / 16 lanes of a wavefront (16 / 8 / 4
 * envs per wavefront) -- and a handle belongs to the smallest one that holds its n_walkers.  Record sizes (madrl_multiwalker_state_bytes,
 * madrl_multiwalker_record_bytes) depend on the class: a change of n_walkers (the reference's update_curriculum builds a new env,
 * runners/curriculum.py:56-91) is a new handle over a new state buffer.  Either pointer may be NULL. */
int madrl_multiwalker_lanes(const madrl_multiwalker *h, int32_t *cap_walkers, int32_t *lanes_per_env);
/* layout of the state buffer: first n_envs blocks of stride_bytes each -- the world record (world_bytes: bodies, flags, joints,
 * contacts, broad phase, terrain) followed by the step's scratch (solver schedule and manifolds, handed from launch to launch: one
 * step is a sequence of kernel launches); behind them the library's own part (pending flags, the spare records of the auto-reset,
 * their observations and bookkeeping) */
int madrl_multiwalker_record_bytes(const madrl_multiwalker *h, int32_t *stride_bytes, int32_t *world_bytes);
/* MultiWalkerEnv.reset (:330-357) incl. its trailing zero-action step; obs float32 [N][W][obs_dim] (32, or 71 with one_hot) */
int madrl_multiwalker_reset(madrl_multiwalker *h, const uint8_t *mask_dev, float *obs_dev, void *stream);
/* MultiWalkerEnv.step (:359-428): actions float32 [N][W][4]; rew float32 [N][W]; done uint8 [N]
 * (bit0 = the reference's done, bit1 = max_steps reached, bit7 = capacity overflow, sticky until the env is reset: a contact did
 * not fit its cache or the step's manifold pool -- sized for walking and falling walkers, not for all of them lying in a heap with
 * terminate_on_fall off -- and was ignored, or the episode has outlasted the 16-bit creation stamps of its contacts (65 280 FindNewContacts
 * calls: some 45 000 steps of walking, 9 000 with ten fallen walkers), so this episode is no longer Box2D's; the same flag as get_state's).  With auto_reset an env whose episode ended continues with its next
 * episode (reset + trailing step, :330-357) and obs holds that episode's first observation; the result is the same as
 * reset(mask = done) after the call, whichever way the library gets there (a prepared spare record or a second pass) */
int madrl_multiwalker_step(madrl_multiwalker *h, const float *actions_dev, float *obs_dev, float *rew_dev,
                           uint8_t *done_dev, void *stream);
/* inspection: bodies float32 [N][NB][6] = centre x, y, angle, vx, vy, w (body 0 = package, then per walker hull,
 * upper/lower left leg, upper/lower right leg); flags uint8 [N][1+3W] = game_over, fallen[W], ground_contact[W][2];
 * terrain float32 [N][NT] heights.  Any pointer may be NULL. */
int madrl_multiwalker_get_bodies(madrl_multiwalker *h, float *bodies_dev, uint8_t *flags_dev, float *terrain_dev,
                                 void *stream);
/* Unpacked view of the world state (checkpoint / teacher-forcing hook; device pointers, any may be NULL):
 *   bodies float32 [N][NB][6] as above (b2Body: m_sweep.c, m_sweep.a, m_linearVelocity, m_angularVelocity);
 *   joints float32 [N][4W][6] = accumulated impulse x, y, z, motor impulse, limit state (0 inactive, 1 at lower, 2 at upper,
 *          3 equal), motor speed (b2RevoluteJoint: m_impulse, m_motorImpulse, m_limitState, m_motorSpeed), per walker
 *          hip / knee of the left leg, hip / knee of the right leg;
 *   aux    float32 [N][NB][6] = the broad phase's fat AABB of the body's proxy (lower x, y, upper x, y), b2Body::m_sleepTime,
 *          awake flag;
 *   flags  uint8 [N][2+3W] = game_over, fallen[W], ground_contact[W][2], overflow (sticky: a contact did not fit its cache or
 *          the step's manifold pool and was ignored);
 *   terrain float32 [N][NT].
 * The contacts (b2Contact list: pairs, creation order, feature ids, warm-start impulses) stay in the raw state buffer; its
 * layout is mw::World in madrl_amd/csrc/multiwalker_core.hpp.
 * set_state overwrites body poses / velocities and (optionally) the joints' accumulated impulses; contacts, fat AABBs and sleep
 * times stay as they are. */
int madrl_multiwalker_get_state(madrl_multiwalker *h, float *bodies_dev, float *joints_dev, float *aux_dev, uint8_t *flags_dev,
                                float *terrain_dev, void *stream);
int madrl_multiwalker_set_state(madrl_multiwalker *h, const float *bodies_dev, const float *joints_dev, void *stream);
/* reset with the random draws given (parity hook): terrain float64 [N][NT] heights instead of the _generate_terrain walk
 * (:516-612), push float64 [N][W] instead of uniform(-INITIAL_RANDOM, INITIAL_RANDOM) (:130-131); either may be NULL */
int madrl_multiwalker_reset_with(madrl_multiwalker *h, const uint8_t *mask_dev, const double *terrain_dev, const double *push_dev,
                                 float *obs_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * Env wrappers as epilogue kernels (reference: madrl_environments/__init__.py:143-389).
 * Every env instance carries its own wrapper state (caller-owned device buffers, zero / one
 * initialised as noted); `mask` / `reset_mask` are uint8 [n_envs] or NULL.  Where a kernel reads DONE BYTES of a step launch as episode
 * boundaries (obsbuffer's reset_mask, diagnostics' done, madrl_rollout_gae's done) bit 7 -- the sticky overflow report of the Pursuit and
 * MultiWalker kernels, which resets nothing -- is ignored.
 * ---------------------------------------------------------------------------------------- */
/* StandardizedEnv.standardize_obs (:242-263): mean/var float64 [n_elems] (init 0 / 1, :229-230) */
int madrl_wrap_obsnorm(const float *obs_in, double *mean, double *var, float *obs_out, int64_t n_elems,
                       int64_t elems_per_env, const uint8_t *mask, double alpha, double eps, void *stream);
/* StandardizedEnv.standardize_rew + scale_reward (:251-271, :290): mean/var float64 [n] (init 0 / 1) */
int madrl_wrap_rewnorm(const float *rew_in, double *mean, double *var, float *rew_out, int64_t n, int64_t per_env,
                       const uint8_t *mask, double alpha, double eps, double scale, int32_t enable_norm, void *stream);
/* ObservationBuffer (:176-195): buf float32 [n_elems][k]; envs flagged in reset_mask fill all k slots (reset,
 * :190-192), the others shift their history and push (step, :179-181).  active_mask (uint8 [N] or NULL = all):
 * envs with a 0 are left untouched -- a partial reset(mask) passes the same mask twice. */
int madrl_wrap_obsbuffer(const float *obs, float *buf, int64_t n_elems, int64_t elems_per_env, int32_t k,
                         const uint8_t *reset_mask, const uint8_t *active_mask, void *stream);
/* DiagnosticsWrapper.step (:335-369): per-env accumulators ep_reward float64 [N][A], ep_len int32 [N],
 * disc_ret / disc_pow float64 [N] (all init 0); on episode end (done bit 0 or 1, or max_traj_len) the out_* rows
 * receive episode_reward_agent*, episode_disc_return, episode_length and out_finished = 1 */
int madrl_wrap_diagnostics(const float *rew, const uint8_t *done, double *ep_reward, int32_t *ep_len,
                           double *disc_ret, double *disc_pow, int64_t n_envs, int32_t n_agents, double discount,
                           int32_t max_traj_len, double *out_ep_reward, double *out_disc, int32_t *out_len,
                           uint8_t *out_finished, void *stream);

/* ------------------------------------------------------------------------------------------
 * Rollout post-processing: what the external samplers the runners hand the env to do with a batch of
 * paths (runners/rurllab.py:298-305 discount / gae_lambda; runners/rurltools.py:196-209), as one reverse
 * scan in time over the time-major trajectory tensors of the collector.
 *   rew float32 [T][N][A], done uint8 [T][N] (bit 0 or 1 = episode boundary after step t; bit 7, the overflow report, is none),
 *   values float32 [T+1][N][A] or NULL (baseline predictions; row T bootstraps the unfinished tail),
 *   returns[t] = rew[t] + gamma * (done[t] ? 0 : returns[t+1]),      returns[T] = values[T] or 0
 *   delta[t]   = rew[t] + gamma * (done[t] ? 0 : values[t+1]) - values[t]
 *   adv[t]     = delta[t] + gamma * lambda * (done[t] ? 0 : adv[t+1])       (adv may be NULL; needs values)
 * Accumulation in float64, results float32 [T][N][A].
 * ---------------------------------------------------------------------------------------- */
int madrl_rollout_gae(const float *rew, const uint8_t *done, const float *values, int64_t T, int64_t n_envs,
                      int32_t n_agents, double gamma, double lambda, float *returns, float *adv, void *stream);

/* ------------------------------------------------------------------------------------------
 * Hand-written policies (reference: heuristics/pursuit.py:13-56, heuristics/waterworld.py:6-62,
 * heuristics/multi_walker.py:10-86): one action per observation row, n_rows = n_envs * n_agents.
 * ---------------------------------------------------------------------------------------- */
/* PursuitHeuristicPolicy.sample_actions: element (i, j) of the evader channel of row r is
 * obs[r * row_stride + ch_offset + (i * obs_range + j) * cell_stride]  (flatten rows: ch_offset = 2*R*R,
 * cell_stride = 1; (R,R,4) windows: ch_offset = 2, cell_stride = 4).  table_dev uint8 [R*R]: the action for
 * "nearest evader at window cell k" (255 = sample); empty window / 255 -> Philox(seed; row_id_base + r,
 * tick + tick_dev[0]).  tick_dev: NULL, or MADRL_POLICY_COUNTER_WORDS zero-initialised uint32 words on the device -- [0]
 * is the draw counter of a captured hipGraph and is advanced by one BY THE LAUNCH ITSELF when its last workgroup retires
 * (the other words count retired workgroups and are zero again between launches), so launches that share a counter must
 * be ordered on one stream.
 * actions int32 [n_rows] */
#define MADRL_POLICY_COUNTER_WORDS (32 * 65)
int madrl_heuristic_pursuit(const float *obs, int64_t n_rows, int32_t obs_range, int64_t row_stride, int32_t cell_stride,
                            int32_t ch_offset, const uint8_t *table_dev, uint64_t seed, int64_t row_id_base,
                            uint32_t tick, uint32_t *tick_dev, int32_t *actions, void *stream);
/* WaterworldHeuristicPolicy.sample_actions, every row normalised on its own; cos_sin_dev float64 [K][2] with
 * K = obs_dim / 7 (np.linspace(0, 2 pi, K + 1)[:-1]); actions float32 [n_rows][2] */
int madrl_heuristic_waterworld(const float *obs, int64_t n_rows, int32_t obs_dim, const double *cos_sin_dev,
                               float *actions, void *stream);
/* MultiWalkerHeuristicPolicy.sample_actions; actions float32 [n_rows][4] */
int madrl_heuristic_multiwalker(const float *obs, int64_t n_rows, int32_t obs_dim, float *actions, void *stream);

/* Philox4x32-10 on the host, exported so tests can pin the generator the kernels use
 * against the published known-answer vectors. */
void madrl_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* MADRL_HIP_H */

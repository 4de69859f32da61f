/*
 * msnake.h -- C-ABI of the MI355X-native batched multi-snake environment step.
 *
 * One handle owns the state of `num_envs` independent dim x dim multi-snake games in HBM and
 * steps all of them with ONE HIP kernel launch.  No torch types, no C++ types: plain pointers and
 * sizes, so it can be bound from ctypes / cffi / cgo / JNI alike.  All `*_dev` pointers are
 * DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr()); the library never copies
 * observations through the host.  Every function returns 0 on success or a negative MSNAKE_E_*
 * code; msnake_last_error() returns a thread-local description of the last failure.
 *
 * What each entry point replaces in the reference (paths under /root/reference/src/):
 *   msnake_create   <- gym.make + env.__init__ + env.seed(seed+rank) for every SubprocVecEnv worker
 *                      (utils.py:34-49 make_basic_env, baselines/common/vec_env/subproc_vec_env.py:32-50)
 *   msnake_reset    <- SubprocVecEnv.reset (subproc_vec_env.py:63-66) -> SnakeEnv.reset
 *                      (gym-snake/gym_snake/envs/snake_multiple_test.py:219-232) /
 *                      NewMultipleSnakes.reset (envs/snake_multiple_env_new.py:27-33)
 *   msnake_step     <- SubprocVecEnv.step_async/step_wait (subproc_vec_env.py:52-61), the worker's
 *                      auto-reset (:13-16), Monitor.step episode stats (baselines/bench/monitor.py:57-78)
 *                      and SnakeEnv.step (snake_multiple_test.py:166-197) /
 *                      World.move_snakes + NewMultipleSnakes.step (core/new_world.py:88-109,
 *                      envs/snake_multiple_env_new.py:35-50) / SnakeAdversarial.step
 *                      (envs/snake_adversarial_env.py:166-201), including the observation render
 *                      get_multi_snake_ob (snake_multiple_test.py:35-58,93-95)
 *   msnake_destroy  <- SubprocVecEnv.close (subproc_vec_env.py:73-83)
 *   msnake_get_state / msnake_set_state / msnake_get_state_all / msnake_set_state_all: no reference
 *                      counterpart (env state is never checkpointed there); used by
 *                      the parity tests to install hand-built states and by callers to checkpoint.
 *   msnake_state_blob_info: no reference counterpart either; validates such a checkpoint on the host.
 *   msnake_get_stats <- the epinfobuf aggregation in ppo_multi_agent.py:288,331,366-390
 *
 * RNG contract (shared with oracle/ and tests/golden): draw i of global env g is word (i & 3) of
 * Philox4x32-10(counter = {i>>2 lo, i>>2 hi, g lo, g hi}, key = {seed lo, seed hi}) -- rocRAND's
 * (seed, subsequence, offset) convention -- and randint(n) = (u32 * n) >> 32.  g = env_id_base +
 * local index, so trajectories do not depend on how envs are sharded over GPUs.
 */
#ifndef MSNAKE_H
#define MSNAKE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSNAKE_ABI_VERSION 3

/* rule sets = the reference's gym ids (gym-snake/gym_snake/__init__.py:11-26) */
#define MSNAKE_RULES_SNAKE_ENV 0   /* snake-multiple-test-v0  : SnakeEnv            */
#define MSNAKE_RULES_NEW_WORLD 1   /* snake-new-multiple-v0   : NewMultipleSnakes   */
#define MSNAKE_RULES_ADVERSARIAL 2 /* snake-adversarial-v0    : SnakeAdversarial    */

#define MSNAKE_MAX_SNAKES 4
#define MSNAKE_MAX_FRUITS 32 /* inline fruits (snake_env / new_world) */
#define MSNAKE_MAX_DIM 62

#define MSNAKE_OK 0
#define MSNAKE_E_ARG (-1)     /* bad argument / configuration                     */
#define MSNAKE_E_HIP (-2)     /* a HIP runtime call failed (message has the text) */
#define MSNAKE_E_HANDLE (-3)  /* NULL or destroyed handle                         */
#define MSNAKE_E_ALIGN (-4)   /* device pointer not aligned as required           */
#define MSNAKE_E_STATE (-5)   /* malformed state buffer in set_state              */
#define MSNAKE_E_NOGPU (-6)   /* no usable HIP device                             */

typedef struct msnake_config {
    uint32_t struct_size;  /* = sizeof(msnake_config), for ABI evolution                   */
    int32_t device;        /* HIP device ordinal                                            */
    int32_t num_envs;      /* envs owned by this handle (this GPU's shard)                  */
    int32_t dim;           /* grid is dim x dim; observation is (dim+2) x (dim+2) x 3*views */
    int32_t n_snakes;      /* 1..3 for snake_env/adversarial (views fixed at 3), 1..4 new_world */
    int32_t n_fruits;      /* must equal n_snakes for snake_env/adversarial; 0..32 new_world */
    int32_t rules;         /* MSNAKE_RULES_*                                                */
    int32_t max_steps;     /* episode cap, 2000 in the reference                            */
    int32_t auto_reset;    /* 1 = vec-env semantics (reset on done, return reset obs)       */
    int32_t obs_scale;     /* integer pixel replication of the observation (1 = native)     */
    uint64_t seed;         /* Philox key                                                    */
    uint64_t env_id_base;  /* global id of local env 0 (Philox subsequence = base + index)  */
    /* ---- ABI 3: launch tuning, all optional (0 = the library decides from the batch size).
     *      They change how the work is laid out on the GPU, never a result: every value is parity-tested.
     *      A caller built against ABI 2 passes struct_size = MSNAKE_CONFIG_SIZE_V2 (the fields above);
     *      the tail is then taken as zeros. */
    int32_t envs_per_block;    /* envs (= wavefronts) per workgroup: 0 auto | 1..8                      */
    int32_t record_policy;     /* MSNAKE_AUTO | MSNAKE_RECORD_FULL | MSNAKE_RECORD_SHORT (snake_env / adversarial) */
    int32_t obs_store_policy;  /* MSNAKE_AUTO | MSNAKE_STORE_PLAIN | MSNAKE_STORE_STREAM: msnake_step / reset / render */
    int32_t tape_store_policy; /* the same for msnake_rollout_tape                                      */
} msnake_config;

#define MSNAKE_CONFIG_SIZE_V2 56u
#define MSNAKE_AUTO 0
#define MSNAKE_RECORD_FULL 1  /* 256-byte env record: the upper half parks Philox draws between launches */
#define MSNAKE_RECORD_SHORT 2 /* only the first 128 bytes of the record move (bandwidth-bound batches)   */
#define MSNAKE_STORE_PLAIN 1  /* observation stores stay in L2 / Infinity Cache                          */
#define MSNAKE_STORE_STREAM 2 /* observation stores carry the nt (streaming) hint                        */

/* per-env info written by msnake_step, 16 bytes, same meaning as the reference's info dict */
typedef struct msnake_info {
    float ep_return;    /* info['episode']['r'] when done, else 0 */
    int32_t ep_len;     /* info['episode']['l'] when done, else 0 */
    int32_t num_snakes; /* info['num_snakes']                     */
    int32_t flags;      /* bit 0: done                            */
} msnake_info;

/* aggregate episode statistics since create (or since the last msnake_get_stats(reset=1)).  An
 * episode is counted on the step it ends; with auto_reset = 0 a finished env keeps returning
 * done = 1 until msnake_reset, but is counted once.  Per env the totals are kept as 32-bit episode
 * count / 64-bit length sum / 32-bit signed return sum between two msnake_get_stats(reset=1) calls. */
typedef struct msnake_stats {
    int64_t episodes;      /* number of finished episodes                 */
    int64_t ep_len_sum;    /* sum of their lengths                        */
    int64_t ep_return_sum; /* sum of their returns (rewards are integral) */
    int64_t env_steps;     /* env-steps executed                          */
    int64_t errors;        /* internal capacity guards tripped (must be 0) */
    int64_t reserved[3];
} msnake_stats;

typedef struct msnake_env* msnake_handle;

int msnake_abi_version(void);
const char* msnake_last_error(void);

int msnake_create(const msnake_config* cfg, msnake_handle* out);
int msnake_destroy(msnake_handle h);

/* observation layout: uint8 [num_envs][H][W][C], C fastest (HWC like the reference) */
int msnake_obs_shape(msnake_handle h, int32_t* H, int32_t* W, int32_t* C);

/* Reset every env; writes observations if obs_dev != NULL.  Asynchronous on `stream`
 * (a hipStream_t passed as void*, NULL = the default stream). */
int msnake_reset(msnake_handle h, uint8_t* obs_dev, void* stream);

/* One lockstep step of every env.  actions_dev: int32 [num_envs][action_stride], entry s of a
 * row is snake s's action in {0..4}; action_stride >= n_snakes, surplus entries are ignored
 * (ppo_multi_agent.py:41-44 always sends tuples of 2 or 3).  rew_dev float32[num_envs],
 * done_dev uint8[num_envs], info_dev msnake_info[num_envs] (may be NULL).  obs_dev may be NULL (no
 * render); it needs no alignment at obs_scale 1 and 4-byte alignment at obs_scale 4 / 7
 * (MSNAKE_E_ALIGN otherwise).  Asynchronous. */
int msnake_step(msnake_handle h, const int32_t* actions_dev, int32_t action_stride, uint8_t* obs_dev,
                float* rew_dev, uint8_t* done_dev, msnake_info* info_dev, void* stream);

/* Same, for `n_steps` consecutive steps from an action tape int32 [n_steps][num_envs][stride];
 * outputs of step k go to obs_dev + k*obs_step_bytes etc. when the *_step_stride arguments are
 * non-zero, or are overwritten in place when they are zero.  One launch per step, issued from C. */
int msnake_step_tape(msnake_handle h, const int32_t* actions_dev, int32_t action_stride, int32_t n_steps,
                     uint8_t* obs_dev, size_t obs_step_stride, float* rew_dev, uint8_t* done_dev,
                     msnake_info* info_dev, size_t scalar_step_stride, void* stream);

/* Same contract and same results as msnake_step_tape, but ONE persistent launch: every wave keeps
 * its env in registers across the n_steps steps, so there is no per-step launch boundary and no
 * per-step state round trip.  For callers that have the actions of several steps up front
 * (scripted / random opponents, evaluation replays, benchmarks); a policy in the loop needs
 * msnake_step. */
int msnake_rollout_tape(msnake_handle h, const int32_t* actions_dev, int32_t action_stride, int32_t n_steps,
                        uint8_t* obs_dev, size_t obs_step_stride, float* rew_dev, uint8_t* done_dev,
                        msnake_info* info_dev, size_t scalar_step_stride, void* stream);

/* Canonical per-env state as int32 words (blocking; test / checkpoint path):
 *  [0] t  [1] ctr_lo  [2] ctr_hi  [3] spare_fruits  [4] ep_len  [5] ep_return (f32 bits)
 *  [6] n_fruits_cur  [7] n_snakes | finished << 8, then n_fruits_cur x (c0,c1), then per snake:
 *  len, v0, v1, grow_to, alive, in_dead, len x (c0,c1) head first.
 * `finished` (bit 8 of word 7): the episode has ended and the env has not been reset since (only ever set
 * with auto_reset = 0); it keeps a restored env from counting that episode into msnake_get_stats again.
 * Heads and the entries of the adversarial fruit list lie in [-1, dim] (one step outside the grid is
 * where the reference can put them); body pieces behind the head and the fruits of snake_env /
 * new_world lie inside the grid [0, dim).  Anything else is MSNAKE_E_STATE.
 * msnake_get_state returns the number of words needed/written (>0) or a negative error. */
int msnake_get_state(msnake_handle h, int32_t env, int32_t* words, int32_t cap);
int msnake_set_state(msnake_handle h, int32_t env, const int32_t* words, int32_t n);

/* The same canonical words for EVERY env of the handle in one host buffer (blocking; checkpoints):
 * one packing kernel on the device and one copy, instead of num_envs round trips.  Layout of `buf`:
 *   { uint32 magic "MSST", uint32 version (2; version-1 blobs, which lack the finished bit, are accepted),
 *     int32 num_envs, dim, n_snakes, n_fruits, rules, reserved,
 *     uint64 total_words }  (40 bytes)
 *   uint64 offsets[num_envs + 1]   word offset of env e's state inside `words`
 *   int32  words[total_words]      env e's words = words[offsets[e] .. offsets[e+1]), layout as above
 * msnake_get_state_all returns the number of bytes needed; it writes them only if buf != NULL and
 * cap_bytes suffices (call once with NULL to size the buffer).  msnake_set_state_all checks the
 * blob against the handle's configuration and every env's words like msnake_set_state does; envs
 * with malformed words are left untouched and the call fails with MSNAKE_E_STATE.  The per-env
 * logging totals (msnake_get_stats) are not part of the canonical state and are kept. */
int64_t msnake_get_state_all(msnake_handle h, void* buf, size_t cap_bytes);
int msnake_set_state_all(msnake_handle h, const void* buf, size_t bytes);

/* What a blob holds, checked on the host without a handle or a device: magic / version, the offset table
 * (starts at 0, never decreases, ends at total_words) and that `bytes` covers all of it.  MSNAKE_E_STATE
 * otherwise.  Lets a caller size the handle a checkpoint needs before creating it; msnake_set_state_all
 * runs the same check first.  `out` may be NULL. */
typedef struct msnake_blob_info {
    int32_t version, num_envs, dim, n_snakes, n_fruits, rules;
    int64_t total_words;
} msnake_blob_info;
int msnake_state_blob_info(const void* buf, size_t bytes, msnake_blob_info* out);

/* Render the current state of every env without stepping (asynchronous). */
int msnake_render(msnake_handle h, uint8_t* obs_dev, void* stream);

/* Copy the aggregate statistics to the host.  Blocking: waits for the device (every step issued so
 * far, on any stream) before it sums the per-env totals.  episodes / ep_len_sum / ep_return_sum /
 * errors are accumulated on the device; env_steps is counted on the host per API call, so replays
 * of a captured HIP graph are not included in it. */
int msnake_get_stats(msnake_handle h, msnake_stats* out, int32_t reset);

/* Name of the step kernel (for profilers; the string is owned by the handle and lives until
 * msnake_destroy) and algorithmic HBM bytes per env-step. */
const char* msnake_kernel_name(msnake_handle h);
int64_t msnake_algorithmic_bytes_per_env_step(msnake_handle h);

#ifdef __cplusplus
}
#endif
#endif /* MSNAKE_H */

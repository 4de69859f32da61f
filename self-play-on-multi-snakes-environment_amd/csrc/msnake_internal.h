// msnake_internal.h -- shared between the kernels and the C-ABI glue (not installed).
#ifndef MSNAKE_INTERNAL_H
#define MSNAKE_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msnake.h"

// One wavefront per env; up to 8 envs (512 threads) per workgroup, no workgroup barrier anywhere.
#define MSNAKE_BLOCK_THREADS 512
#define MSNAKE_MAX_ENVS_PER_BLOCK 8

// Per-env record in HBM: a 256-byte slot of 64 dwords (lane l <-> word l), loaded/stored by ONE
// coalesced wave instruction; snake_env / adversarial keep everything a step needs in the FIRST 128
// bytes, so large batches move only that half (StepParams::short_rec).  Next to it, per snake, a
// 128-byte RING of its 64 most recent body cells (slot l <-> lane l; piece i < min(len, 64) sits in
// slot (hp0 + i) & 63), and, only for bodies longer than 64 cells, an overflow ring with the older
// pieces (piece i >= 64 at ovf[(ohp + i - 64) % cap]).  Record and body rings sit at addresses that
// depend only on the env index: one memory round trip per step.  A step dirties the record, ONE
// 32-byte sector per moving snake (the sector that holds the new head slot) and the outputs.
#define MSNAKE_HDR_WORDS 64
#define MSNAKE_HDR_SHORT_WORDS 32
// words 0..11: three 4-lane "columns" so that lane s of a DPP row shift holds snake s's field
#define SN_A(s) (0 + (s))   // overflow ring head pos ohp | len << 16
#define SN_B(s) (4 + (s))   // grow_to ([S] grow_to_lengths / [N] Snake.snake_length)
#define SN_C(s) (8 + (s))   // head cell | velocity code << 16 | hp0 (body ring slot of the head) << 24
#define SN_C_HP0_SHIFT 24
#define HDR_T 12         // steps since reset ([S] state[4] / [NE] current_step)
#define HDR_CTR_LO 13    // Philox draws consumed (64 bit)
#define HDR_CTR_HI 14
#define HDR_FLAGS 15     // [N] bit s: Snake.alive, bit 4+s: snake in World.dead_snakes; all rules bit 8:
                         // the episode has ended and the env has not been reset since (auto_reset off)
#define HDR_FLAG_FINISHED 0x100u
#define HDR_EP_RETURN 16 // Monitor: running episode return (f32 bits)
#define HDR_EP_LEN 17    // Monitor: running episode length
#define HDR_SPARE 18     // [A] spare_fruits
#define HDR_NLIST 19     // [A] length of the fruit list
#define HDR_FRUIT0_S 20  // snake_env: words 20..22 = fruit f's cell (one fruit per snake, at most 3)
#define HDR_ACC_EPISODES 24  // per-env totals since the last msnake_get_stats(reset=1): no hot-path atomics
#define HDR_ACC_LEN 25       // low word; the carry goes to HDR_ACC_LEN_HI
#define HDR_ACC_RETURN 26    // int32, rewards are integral
#define HDR_ACC_ERRORS 27
#define HDR_ACC_LEN_HI 28
#define HDR_FRUIT0_N 32  // new_world: words 32+f = fruit f's cell (up to 32 fruits; always the full record)
// snake_env / adversarial, full record only: the upper half carries the next Philox draws, so the
// waves that respawn a fruit or reset -- the last ones to finish in a launch -- rarely have to
// evaluate Philox themselves.  Worth its 256 bytes per env-step only while launches are latency
// bound (small batches); with short_rec the words are neither loaded nor stored.
#define HDR_PC_VALID 35  // 1: words 37..63 hold the u32 of draws [PC_BASE, PC_BASE + 27)
#define HDR_PC_BASE 36   // low 32 bits of the first cached draw's index
#define HDR_PC_FIRST 37
#define HDR_PC_N 27

// native-size handles keep 16 pre-shifted copies of the background image (copy s: the image starts s bytes into its
// buffer), for the aligned copy-out of the step kernel; fused x4 / x7 handles keep one
#define MSNAKE_TMPL_COPIES 16

#define MSNAKE_NO_CELL 0xFFFFu  // never equals a real cell (rows/cols <= 63)

namespace msnake {

// kernel arguments that did not fit the 13 preloaded dwords, passed by value behind them
struct StepRest {
    msnake_info* info;           // per-call output (may be NULL)
    float* rew;                  // (host copy; the kernel gets these two preloaded)
    uint8_t* done;
    uint64_t env_id_base;
    uint32_t seed_lo, seed_hi;
    int32_t cap;                 // overflow ring capacity in cells (multiple of 64)
    int32_t max_steps;
    uint32_t dbg_stage;          // timing-only early exits (MSNAKE_DBG_STAGES builds)
    unsigned long long* dbg_buf; // MSNAKE_DBG_STAGES builds: [nenv][8] s_memrealtime stamps (MSNAKE_DBG_BUF)
    unsigned long long* dbg_span; // MSNAKE_DBG_STAGES builds: this launch's [nenv][8] wave stamps (see SPAN in the kernel;
                                  // MSNAKE_DBG_SPAN = base of [slots][nenv][8], one slot per launch)
    int32_t n_steps;             // MODE 3 (msnake_rollout_tape): steps per launch
    uint64_t obs_step_stride;    // MODE 3: bytes between consecutive steps' observations (0 = overwrite)
    uint64_t scalar_step_stride; // MODE 3: elements between consecutive steps' rew/done/info
    uint32_t stream_obs;         // (host copy; the kernel gets it in pk2)
};

// pk2: flag bits preloaded with the other kernel arguments
#define PK2_SHORT_REC 1u   // move only the first 128 bytes of each record
#define PK2_STREAM_OBS 2u  // observation stores carry the nt (streaming) hint (fused frames: the flat 16-byte copy-out)
#define PK2_STREAM_TAPE 4u // same for the persistent tape kernel (msnake_rollout_tape)
#define PK2_TAPE_ALIGNED 8u // msnake_rollout_tape: the step stride is a multiple of 16 bytes -> aligned copy-out for every step
#define PK2_EPB_SHIFT 8    // bits 8..15: envs (waves) per workgroup
#define PK2_DIVM_SHIFT 16  // bits 16..31: the multiplier of the division-free x / dim (slow paths)

struct StepParams {
    // configuration
    int32_t nenv, dim, n_snakes, n_fruits, views, C, S, auto_reset;
    int32_t lds_per_wave;  // bytes of LDS per env: padded image + occupancy bytes
    int32_t img_bytes;     // S rounded up to 1 KiB
    int32_t action_stride;
    int32_t obs_scale;     // fused WarpFrame replication factor (1, 4 or 7)
    int32_t short_rec;     // 1: the step moves only the first 128 bytes of each record
    int32_t stream_tape;   // 1: msnake_rollout_tape stores observations with the nt hint (set per call)
    // state (HBM, owned by the handle): ONE allocation
    //   [hdr: nenv x 256 B][body0: nenv x n_snakes x 128 B][tmpl: img_bytes][ovf: nenv x n_snakes x cap x 2 B]
    //   adversarial only: [fl0: nenv x 128 B][flist: nenv x fcap x 2 B] (fruit list chunk 0 / complete)
    uint8_t* state;
    uint32_t* hdr;               // views into `state` for the host-side paths
    uint16_t* body0;             // per snake: ring of the 64 most recent cells
    const uint8_t* tmpl;
    uint16_t* ring;              // per snake: overflow ring, pieces >= 64
    uint16_t* fl0;               // adversarial: first 64 fruit-list entries per env
    uint16_t* flist;             // adversarial: complete fruit list per env, fcap entries
    int32_t fcap;
    unsigned long long* stats;   // [8], separate small allocation (msnake_get_stats)
    // per-call i/o (device pointers owned by the caller)
    const int32_t* actions;
    uint8_t* obs;
    StepRest rest;
};

hipError_t launch_stats(uint32_t* hdr, int nenv, unsigned long long* stats, int clear, hipStream_t stream);
hipError_t launch_step(const StepParams& p, int rules, int mode, int envs_per_block, hipStream_t stream);
// canonical state words (include/msnake.h) of envs [env0, env0 + count) <-> the device layout
hipError_t launch_state_sizes(const StepParams& p, int rules, int env0, int count, uint32_t* need_words, hipStream_t stream);
hipError_t launch_state_pack(const StepParams& p, int rules, int env0, int count, const uint64_t* offsets, int32_t* words,
                             hipStream_t stream);
hipError_t launch_state_unpack(const StepParams& p, int rules, int env0, int count, const uint64_t* offsets,
                               const int32_t* words, uint32_t* status, hipStream_t stream);
void step_kernel_name(int rules, int n_snakes, int obs_scale, char* out, size_t n);

}  // namespace msnake

#endif

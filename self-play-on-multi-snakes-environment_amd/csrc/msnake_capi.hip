// msnake_capi.hip -- the C-ABI of include/msnake.h over the HIP kernels.  Host-side glue only:
// argument checks, HBM allocation of the per-env state, the background image, launches, and the
// (blocking, test/checkpoint-path) canonical state import/export, whose packing itself runs on the
// device.  No CPU fallback: without a HIP device every entry point fails with MSNAKE_E_NOGPU /
// MSNAKE_E_HIP.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "msnake_internal.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(MSNAKE_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                        __LINE__);                                                                      \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

constexpr uint32_t kMagic = 0x4D534E4Bu;  // "MSNK"

}  // namespace

struct msnake_env {
    uint32_t magic;
    msnake_config cfg;
    msnake::StepParams p;  // configuration + state pointers; i/o pointers filled per call
    int epb;               // envs per workgroup
    int64_t env_steps;
    void* d_state;
    void* d_stats;
    void* d_scratch;       // per-env state import/export staging (grown on demand)
    size_t scratch_bytes;
    char kname[64];
    // MSNAKE_DBG_STAGES builds only (tools/span_gap.py): per-launch slots of wave stamps
    unsigned long long* dbg_span;
    uint32_t dbg_span_slots, dbg_launch;
};

namespace {

int check(msnake_handle h) {
    if (!h || h->magic != kMagic) return fail(MSNAKE_E_HANDLE, "invalid or destroyed msnake handle");
    return MSNAKE_OK;
}

int ensure_scratch(msnake_env* h, size_t bytes) {
    if (bytes <= h->scratch_bytes) return MSNAKE_OK;
    if (h->d_scratch) (void)hipFree(h->d_scratch);
    h->d_scratch = nullptr; h->scratch_bytes = 0;
    hipError_t e = hipMalloc(&h->d_scratch, bytes);
    if (e != hipSuccess) return fail(MSNAKE_E_HIP, "allocating %zu bytes of state staging failed: %s", bytes, hipGetErrorString(e));
    h->scratch_bytes = bytes;
    return MSNAKE_OK;
}

const char* state_reason(uint32_t code) {
    switch (code) {
        case 1: return "state buffer too short / truncated";
        case 2: return "snake count differs from the handle's";
        case 3: return "fruit count differs from the handle's (or exceeds the fruit-list capacity)";
        case 4: return "a cell lies outside [-1, dim], or a body piece behind the head outside the grid";
        case 5: return "a body length is outside [0, capacity]";
        default: return "malformed state";
    }
}

}  // namespace

extern "C" {

int msnake_abi_version(void) { return MSNAKE_ABI_VERSION; }
const char* msnake_last_error(void) { return g_err; }

int msnake_create(const msnake_config* cfg_in, msnake_handle* out) {
    if (!cfg_in || !out) return fail(MSNAKE_E_ARG, "msnake_create: NULL argument");
    *out = nullptr;
    // ABI 3 appended the launch-tuning fields; an ABI-2 caller passes the 56-byte prefix and gets "auto"
    if (cfg_in->struct_size != sizeof(msnake_config) && cfg_in->struct_size != MSNAKE_CONFIG_SIZE_V2)
        return fail(MSNAKE_E_ARG, "msnake_config.struct_size %u is neither %zu (ABI 3) nor %u (ABI 2)", cfg_in->struct_size,
                    sizeof(msnake_config), MSNAKE_CONFIG_SIZE_V2);
    msnake_config cfg_full;
    memset(&cfg_full, 0, sizeof(cfg_full));
    memcpy(&cfg_full, cfg_in, cfg_in->struct_size);
    cfg_full.struct_size = (uint32_t)sizeof(msnake_config);
    const msnake_config* cfg = &cfg_full;
    if (cfg->num_envs < 1) return fail(MSNAKE_E_ARG, "num_envs must be >= 1 (got %d)", cfg->num_envs);
    if (cfg->dim < 2 || cfg->dim > MSNAKE_MAX_DIM)
        return fail(MSNAKE_E_ARG, "dim must be in [2, %d] (got %d)", MSNAKE_MAX_DIM, cfg->dim);
    if (cfg->rules != MSNAKE_RULES_SNAKE_ENV && cfg->rules != MSNAKE_RULES_NEW_WORLD &&
        cfg->rules != MSNAKE_RULES_ADVERSARIAL)
        return fail(MSNAKE_E_ARG, "rules %d is not one of MSNAKE_RULES_*", cfg->rules);
    if (cfg->rules == MSNAKE_RULES_NEW_WORLD) {
        if (cfg->n_snakes < 1 || cfg->n_snakes > MSNAKE_MAX_SNAKES)
            return fail(MSNAKE_E_ARG, "new_world: n_snakes must be in [1, %d]", MSNAKE_MAX_SNAKES);
        if (cfg->n_fruits < 0 || cfg->n_fruits > MSNAKE_MAX_FRUITS)
            return fail(MSNAKE_E_ARG, "new_world: n_fruits must be in [0, %d]", MSNAKE_MAX_FRUITS);
    } else {
        // SnakeEnv.reset hard-codes 3 velocity / grow_to slots and one fruit per snake
        // (snake_multiple_test.py:223-228)
        if (cfg->n_snakes < 1 || cfg->n_snakes > 3)
            return fail(MSNAKE_E_ARG, "snake_env: n_snakes must be in [1, 3] (got %d)", cfg->n_snakes);
        if (cfg->n_fruits != cfg->n_snakes)
            return fail(MSNAKE_E_ARG, "snake_env: n_fruits must equal n_snakes");
    }
    if (cfg->max_steps < 1 || cfg->max_steps > 60000)
        return fail(MSNAKE_E_ARG, "max_steps must be in [1, 60000]");
    if ((uint64_t)cfg->num_envs * (uint64_t)(MSNAKE_HDR_WORDS * 4 + MSNAKE_MAX_SNAKES * 128) >= (1ull << 32))
        return fail(MSNAKE_E_ARG, "num_envs %d: records and body rings of one handle must stay below 4 GB (the kernel "
                                  "addresses them with 32-bit offsets); use several handles", cfg->num_envs);
    if (cfg->obs_scale != 1 && cfg->obs_scale != 4 && cfg->obs_scale != 7)
        return fail(MSNAKE_E_ARG, "obs_scale must be 1, 4 (21->84) or 7 (12->84), got %d", cfg->obs_scale);
    if (cfg->envs_per_block < 0 || cfg->envs_per_block > MSNAKE_MAX_ENVS_PER_BLOCK)
        return fail(MSNAKE_E_ARG, "envs_per_block must be 0 (auto) or in [1, %d], got %d", MSNAKE_MAX_ENVS_PER_BLOCK, cfg->envs_per_block);
    if (cfg->record_policy < 0 || cfg->record_policy > MSNAKE_RECORD_SHORT)
        return fail(MSNAKE_E_ARG, "record_policy must be MSNAKE_AUTO, MSNAKE_RECORD_FULL or MSNAKE_RECORD_SHORT, got %d", cfg->record_policy);
    if (cfg->obs_store_policy < 0 || cfg->obs_store_policy > MSNAKE_STORE_STREAM || cfg->tape_store_policy < 0 ||
        cfg->tape_store_policy > MSNAKE_STORE_STREAM)
        return fail(MSNAKE_E_ARG, "obs_store_policy / tape_store_policy must be MSNAKE_AUTO, MSNAKE_STORE_PLAIN or MSNAKE_STORE_STREAM");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MSNAKE_E_NOGPU, "no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(MSNAKE_E_ARG, "device %d out of range (have %d)", cfg->device, ndev);
    DeviceGuard guard(cfg->device);

    msnake_env* h = static_cast<msnake_env*>(calloc(1, sizeof(msnake_env)));
    if (!h) return fail(MSNAKE_E_ARG, "out of host memory");
    h->cfg = *cfg;
    msnake::StepParams& p = h->p;
    const int dim = cfg->dim, W = dim + 2, n2 = dim * dim;
    p.nenv = cfg->num_envs;
    p.dim = dim;
    p.n_snakes = cfg->n_snakes;
    p.n_fruits = cfg->n_fruits;
    p.views = cfg->rules == MSNAKE_RULES_NEW_WORLD ? cfg->n_snakes : 3;
    p.C = 3 * p.views;
    p.S = W * W * p.C;
    p.obs_scale = cfg->obs_scale;
    p.rest.max_steps = cfg->max_steps;
    p.auto_reset = cfg->auto_reset ? 1 : 0;
    // ring capacity: snake_env bodies hold distinct in-grid cells plus one transient head;
    // new_world bodies can stack duplicates and are only bounded by the episode length
    int need = n2 + 2;
    if (cfg->rules == MSNAKE_RULES_NEW_WORLD && cfg->max_steps + 2 > need) need = cfg->max_steps + 2;
    p.rest.cap = (need + 63) / 64 * 64;
    const int K = cfg->obs_scale;
    if (K > 1 && (W * K * p.C) % 4 != 0) {
        free(h);
        return fail(MSNAKE_E_ARG, "obs_scale %d: output rows must be whole dwords", K);
    }
    p.img_bytes = (p.S * K + (K == 1 ? 15 : 0) + 1023) / 1024 * 1024;  // K-fold wide image, whole 1 KiB wave-instructions
    p.rest.seed_lo = (uint32_t)cfg->seed;
    p.rest.seed_hi = (uint32_t)(cfg->seed >> 32);
    p.rest.env_id_base = cfg->env_id_base;
    p.lds_per_wave = p.img_bytes + (n2 + 15) / 16 * 16;  // composed image + respawn occupancy
    // envs (= waves) per workgroup: 8 up to 8 192 envs (512 workgroups at 4 096 envs start 3 % sooner
    // than 1 024: 7.63 -> 7.40 us per launch), 4 above (an 8-wave workgroup needs 8 free wave slots on
    // one CU at once: 29.5 vs 30.9 us at 32 768 envs)
    h->epb = p.nenv <= 8192 ? MSNAKE_MAX_ENVS_PER_BLOCK : 4;
    while (h->epb > 1 && (size_t)h->epb * p.lds_per_wave > 64 * 1024) h->epb >>= 1;
    if ((size_t)p.lds_per_wave > 64 * 1024 - 16 || p.S > 0xFFFF) {
        free(h);
        return fail(MSNAKE_E_ARG, "observation image of %d bytes does not fit in LDS", p.S);
    }

    const size_t hdr_bytes = (size_t)p.nenv * MSNAKE_HDR_WORDS * 4;
    const size_t body0_bytes = (size_t)p.nenv * p.n_snakes * 64 * 2;
    const int tmpl_copies = K == 1 ? MSNAKE_TMPL_COPIES : 1;
    const size_t tmpl_bytes = (size_t)p.img_bytes * tmpl_copies;
    const size_t ring_bytes = (size_t)p.nenv * p.n_snakes * p.rest.cap * 2;
    const bool adv = cfg->rules == MSNAKE_RULES_ADVERSARIAL;
    p.fcap = (p.n_snakes + p.n_snakes * (n2 + 2) + 63) / 64 * 64;  // n fruits + every body of one episode
    const size_t fl0_bytes = adv ? (size_t)p.nenv * 64 * 2 : 0;
    const size_t flist_bytes = adv ? (size_t)p.nenv * p.fcap * 2 : 0;
    const size_t state_bytes = hdr_bytes + body0_bytes + tmpl_bytes + ring_bytes + fl0_bytes + flist_bytes;
    hipError_t e;
    if ((e = hipMalloc(&h->d_state, state_bytes)) != hipSuccess || (e = hipMalloc(&h->d_stats, 64)) != hipSuccess ||
        (e = hipMemset(h->d_state, 0, state_bytes)) != hipSuccess || (e = hipMemset(h->d_stats, 0, 64)) != hipSuccess) {
        (void)hipFree(h->d_state); (void)hipFree(h->d_stats);
        free(h);
        return fail(MSNAKE_E_HIP, "allocating %zu bytes of env state failed: %s", state_bytes, hipGetErrorString(e));
    }
    p.state = static_cast<uint8_t*>(h->d_state);
    p.hdr = reinterpret_cast<uint32_t*>(p.state);
    p.body0 = reinterpret_cast<uint16_t*>(p.state + hdr_bytes);
    p.tmpl = p.state + hdr_bytes + body0_bytes;
    p.ring = reinterpret_cast<uint16_t*>(p.state + hdr_bytes + body0_bytes + tmpl_bytes);
    p.fl0 = reinterpret_cast<uint16_t*>(p.state + hdr_bytes + body0_bytes + tmpl_bytes + ring_bytes);
    p.flist = reinterpret_cast<uint16_t*>(p.state + hdr_bytes + body0_bytes + tmpl_bytes + ring_bytes + fl0_bytes);
    p.stats = static_cast<unsigned long long*>(h->d_stats);
    // background image: black interior, white 1-px wall ring (snake_multiple_test.py:38,52-56)
    std::vector<uint8_t> tmpl(tmpl_bytes, 0);
    for (int copy = 0; copy < tmpl_copies; ++copy)  // (copy s: the image starts s bytes into its buffer)
        for (int r = 0; r < W; ++r)
            for (int c = 0; c < W; ++c)
                if (r == 0 || r == W - 1 || c == 0 || c == W - 1)
                    memset(&tmpl[(size_t)copy * p.img_bytes + copy + ((size_t)r * W + c) * p.C * K], 255, (size_t)p.C * K);
    if ((e = hipMemcpy(const_cast<uint8_t*>(p.tmpl), tmpl.data(), tmpl_bytes, hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipFree(h->d_state); (void)hipFree(h->d_stats);
        free(h);
        return fail(MSNAKE_E_HIP, "uploading the background image failed: %s", hipGetErrorString(e));
    }
    if (cfg->envs_per_block > 0 && cfg->envs_per_block < h->epb) h->epb = cfg->envs_per_block;  // (never above what LDS allows)
    // record_policy: snake_env / adversarial steps can run on the first 128 bytes of the 256-byte
    // record; the upper half only parks Philox draws for the waves that respawn or reset, which pays
    // while a launch is latency bound (<= 8 192 envs) and is pure traffic above
    p.short_rec = (cfg->rules != MSNAKE_RULES_NEW_WORLD && p.nenv > 8192) ? 1 : 0;
    if (cfg->record_policy != MSNAKE_AUTO)  // (new_world keeps its fruits in the upper half: always the full record)
        p.short_rec = (cfg->rules != MSNAKE_RULES_NEW_WORLD && cfg->record_policy == MSNAKE_RECORD_SHORT) ? 1 : 0;
    // obs_store_policy: should the observation stores carry the nt (streaming) hint?  Measured on
    // MI355X with the native 16-byte-per-lane copy-out (tools/kbench.py --store-policy plain/stream), per launch:
    //   <= 32 MiB of observations (<= 8 192 envs at 19x19x3; fits the 8 L2s): 2-3 % faster with nt
    //      (nothing is left to write back when the kernel ends);
    //   64-130 MiB (fits the 256 MB Infinity Cache): 0-4 % slower (a re-used buffer no longer hits);
    //   >= 260 MiB: 33-42 % faster (388 -> 243 us at 262 144 envs: plain stores thrash the cache).
    // The fused x4 / x7 copy-out stores dwords (256 B per wave instruction) and is 2.5x SLOWER with
    // nt, so it never streams; neither does the persistent tape kernel (187 -> 208 us).
    {
        const double obs_mib = (double)p.nenv * p.S * p.obs_scale * p.obs_scale / (1024.0 * 1024.0);
        p.rest.stream_obs = (p.obs_scale == 1 ? (obs_mib <= 32.0 || obs_mib >= 192.0) : obs_mib >= 400.0) ? 1u : 0u;
    }
    if (cfg->obs_store_policy != MSNAKE_AUTO) p.rest.stream_obs = cfg->obs_store_policy == MSNAKE_STORE_STREAM ? 1u : 0u;
#ifdef MSNAKE_DBG_STAGES  // diagnostic builds only (tools/stamp_profile.py): never in the shipped library
    if (const char* dbg = getenv("MSNAKE_DBG_STAGE")) p.rest.dbg_stage = (uint32_t)atoi(dbg);
    if (const char* dbg = getenv("MSNAKE_DBG_BUF")) p.rest.dbg_buf = reinterpret_cast<unsigned long long*>(strtoull(dbg, nullptr, 0));
#endif
#if defined(MSNAKE_DBG_STAGES) || defined(MSNAKE_SPAN_LIGHT)  // (measurement builds only, like the block above)
    if (const char* dbg = getenv("MSNAKE_DBG_SPAN")) {  // base of a caller-owned [MSNAKE_DBG_SPAN_SLOTS][num_envs][8] u64 device buffer
        h->dbg_span = reinterpret_cast<unsigned long long*>(strtoull(dbg, nullptr, 0));
        const char* sl = getenv("MSNAKE_DBG_SPAN_SLOTS");
        h->dbg_span_slots = sl ? (uint32_t)atoi(sl) : 1u;
        if (h->dbg_span_slots < 1) h->dbg_span_slots = 1;
    }
#endif
    msnake::step_kernel_name(cfg->rules, cfg->n_snakes, cfg->obs_scale, h->kname, sizeof(h->kname));
    h->magic = kMagic;
    *out = h;
    return MSNAKE_OK;
}

int msnake_destroy(msnake_handle h) {
    if (int rc = check(h)) return rc;
    DeviceGuard guard(h->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->d_state); (void)hipFree(h->d_stats);
    if (h->d_scratch) (void)hipFree(h->d_scratch);
    h->magic = 0;
    free(h);
    return MSNAKE_OK;
}

int msnake_obs_shape(msnake_handle h, int32_t* H, int32_t* W, int32_t* C) {
    if (int rc = check(h)) return rc;
    if (H) *H = (h->p.dim + 2) * h->p.obs_scale;
    if (W) *W = (h->p.dim + 2) * h->p.obs_scale;
    if (C) *C = h->p.C;
    return MSNAKE_OK;
}

static int launch(msnake_handle h, int mode, const int32_t* actions, int32_t action_stride, uint8_t* obs, float* rew,
                  uint8_t* done, msnake_info* info, void* stream) {
    msnake::StepParams p = h->p;
    p.actions = actions; p.action_stride = action_stride;
    p.obs = obs; p.rest.rew = rew; p.rest.done = done; p.rest.info = info;
    DeviceGuard guard(h->cfg.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (obs && p.obs_scale > 1 && ((uintptr_t)obs & 3))  // the fused x4 / x7 copy-out stores dwords
        return fail(MSNAKE_E_ALIGN, "obs_dev must be 4-byte aligned when obs_scale > 1");
#if defined(MSNAKE_DBG_STAGES) || defined(MSNAKE_SPAN_LIGHT)
    if (h->dbg_span && mode == 0)  // every msnake_step launch stamps into its own slot (they wrap)
        p.rest.dbg_span = h->dbg_span + (size_t)(h->dbg_launch++ % h->dbg_span_slots) * (size_t)p.nenv * 8;
#endif
    hipError_t e = msnake::launch_step(p, h->cfg.rules, mode, h->epb, s);
    if (e != hipSuccess) return fail(MSNAKE_E_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return MSNAKE_OK;
}

int msnake_reset(msnake_handle h, uint8_t* obs_dev, void* stream) {
    if (int rc = check(h)) return rc;
    return launch(h, 1, nullptr, 0, obs_dev, nullptr, nullptr, nullptr, stream);
}

int msnake_render(msnake_handle h, uint8_t* obs_dev, void* stream) {
    if (int rc = check(h)) return rc;
    if (!obs_dev) return fail(MSNAKE_E_ARG, "msnake_render: obs_dev is NULL");
    return launch(h, 2, nullptr, 0, obs_dev, nullptr, nullptr, nullptr, stream);
}

int msnake_step(msnake_handle h, const int32_t* actions_dev, int32_t action_stride, uint8_t* obs_dev, float* rew_dev,
                uint8_t* done_dev, msnake_info* info_dev, void* stream) {
    if (int rc = check(h)) return rc;
    if (!actions_dev || !rew_dev || !done_dev) return fail(MSNAKE_E_ARG, "msnake_step: actions/rew/done must not be NULL");
    if (action_stride < h->p.n_snakes || action_stride > 7)
        return fail(MSNAKE_E_ARG, "action_stride %d must be in [n_snakes=%d, 7]", action_stride, h->p.n_snakes);
    if (((uintptr_t)actions_dev & 3) || ((uintptr_t)rew_dev & 3) || ((uintptr_t)info_dev & 15))
        return fail(MSNAKE_E_ALIGN, "actions/rew must be 4-byte and info 16-byte aligned");
    int rc = launch(h, 0, actions_dev, action_stride, obs_dev, rew_dev, done_dev, info_dev, stream);
    if (rc == MSNAKE_OK) h->env_steps += h->p.nenv;
    return rc;
}

int msnake_step_tape(msnake_handle h, const int32_t* actions_dev, int32_t action_stride, int32_t n_steps, uint8_t* obs_dev,
                     size_t obs_step_stride, float* rew_dev, uint8_t* done_dev, msnake_info* info_dev,
                     size_t scalar_step_stride, void* stream) {
    if (int rc = check(h)) return rc;
    if (n_steps < 0) return fail(MSNAKE_E_ARG, "n_steps < 0");
    const size_t astep = (size_t)h->p.nenv * (size_t)action_stride;
    for (int32_t k = 0; k < n_steps; ++k) {
        int rc = msnake_step(h, actions_dev + (size_t)k * astep, action_stride,
                             obs_dev ? obs_dev + (size_t)k * obs_step_stride : nullptr,
                             rew_dev + (size_t)k * scalar_step_stride, done_dev + (size_t)k * scalar_step_stride,
                             info_dev ? info_dev + (size_t)k * scalar_step_stride : nullptr, stream);
        if (rc != MSNAKE_OK) return rc;
    }
    return MSNAKE_OK;
}

int msnake_rollout_tape(msnake_handle h, const int32_t* actions_dev, int32_t action_stride, int32_t n_steps, uint8_t* obs_dev,
                        size_t obs_step_stride, float* rew_dev, uint8_t* done_dev, msnake_info* info_dev,
                        size_t scalar_step_stride, void* stream) {
    if (int rc = check(h)) return rc;
    if (n_steps < 1) return fail(MSNAKE_E_ARG, "n_steps must be >= 1");
    if (!actions_dev || !rew_dev || !done_dev) return fail(MSNAKE_E_ARG, "msnake_rollout_tape: actions/rew/done must not be NULL");
    if (action_stride < h->p.n_snakes || action_stride > 7)
        return fail(MSNAKE_E_ARG, "action_stride %d must be in [n_snakes=%d, 7]", action_stride, h->p.n_snakes);
    if (((uintptr_t)actions_dev & 3) || ((uintptr_t)rew_dev & 3) || ((uintptr_t)info_dev & 15))
        return fail(MSNAKE_E_ALIGN, "actions/rew must be 4-byte and info 16-byte aligned");
    msnake::StepParams p = h->p;
    p.actions = actions_dev; p.action_stride = action_stride;
    p.obs = obs_dev; p.rest.rew = rew_dev; p.rest.done = done_dev; p.rest.info = info_dev;
    p.rest.n_steps = n_steps;
    // tape_store_policy: a tape whose steps keep one alignment (step stride a multiple of 16 bytes: the aligned copy-out)
    // and whose observations go to per-step slices of >= 192 MiB in all streams (4 096 envs x 256 steps = 4.2 GB: 3.8 us per
    // step with nt, 4.05 plain); a re-used buffer (stride 0) merges in the L2s and stays plain, and so does the byte-aligned
    // shape of an odd stride (nt there: 4.45 us)
    {
        const double mib = (double)n_steps * p.nenv * p.S * p.obs_scale * p.obs_scale / (1024.0 * 1024.0);
        const bool aligned = p.obs_scale == 1 && obs_dev && obs_step_stride != 0 && obs_step_stride % 16 == 0;
        p.stream_tape = h->cfg.tape_store_policy == MSNAKE_STORE_STREAM || (h->cfg.tape_store_policy == MSNAKE_AUTO && aligned && mib >= 192.0) ? 1 : 0;
    }
    p.rest.obs_step_stride = obs_step_stride;
    p.rest.scalar_step_stride = scalar_step_stride;
    if (obs_dev && p.obs_scale > 1 && (((uintptr_t)obs_dev | obs_step_stride) & 3))
        return fail(MSNAKE_E_ALIGN, "obs_dev and obs_step_stride must be multiples of 4 when obs_scale > 1");
    DeviceGuard guard(h->cfg.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = msnake::launch_step(p, h->cfg.rules, 3, h->epb, s);
    if (e != hipSuccess) return fail(MSNAKE_E_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    h->env_steps += (int64_t)h->p.nenv * n_steps;
    return MSNAKE_OK;
}

// ---- canonical state import / export.  The device packs / unpacks (msnake_state_*_kernel); the host
//      only sizes buffers and copies.  Blocking: every call starts with a device synchronise. ----
namespace {

struct BlobHeader {  // msnake_get_state_all / msnake_set_state_all
    uint32_t magic, version;
    int32_t num_envs, dim, n_snakes, n_fruits, rules, reserved;
    uint64_t total_words;
};
constexpr uint32_t kBlobMagic = 0x5453534Du;  // "MSST"
constexpr uint32_t kBlobVersion = 2u;         // 2: word 7 of an env's words carries the "episode finished" bit

// need[i] (words) of envs [env0, env0 + count) -> host; offsets[count + 1] prefix sums
int state_offsets(msnake_env* h, int env0, int count, std::vector<uint64_t>& offsets) {
    if (int rc = ensure_scratch(h, (size_t)count * 4)) return rc;
    HIP_TRY(msnake::launch_state_sizes(h->p, h->cfg.rules, env0, count, static_cast<uint32_t*>(h->d_scratch), nullptr));
    std::vector<uint32_t> need((size_t)count);
    HIP_TRY(hipMemcpy(need.data(), h->d_scratch, (size_t)count * 4, hipMemcpyDeviceToHost));
    offsets.assign((size_t)count + 1, 0);
    for (int i = 0; i < count; ++i) offsets[(size_t)i + 1] = offsets[(size_t)i] + need[(size_t)i];
    return MSNAKE_OK;
}

// words of envs [env0, env0 + count) at `offsets` -> host buffer `out`
int state_export(msnake_env* h, int env0, int count, const std::vector<uint64_t>& offsets, int32_t* out) {
    const size_t off_bytes = ((size_t)count + 1) * 8, word_bytes = (size_t)offsets.back() * 4;
    if (int rc = ensure_scratch(h, off_bytes + word_bytes)) return rc;
    uint8_t* d = static_cast<uint8_t*>(h->d_scratch);
    HIP_TRY(hipMemcpy(d, offsets.data(), off_bytes, hipMemcpyHostToDevice));
    HIP_TRY(msnake::launch_state_pack(h->p, h->cfg.rules, env0, count, reinterpret_cast<const uint64_t*>(d),
                                      reinterpret_cast<int32_t*>(d + off_bytes), nullptr));
    HIP_TRY(hipMemcpy(out, d + off_bytes, word_bytes, hipMemcpyDeviceToHost));
    return MSNAKE_OK;
}

int state_import(msnake_env* h, int env0, int count, const uint64_t* offsets, const int32_t* words) {
    const size_t off_bytes = ((size_t)count + 1) * 8, word_bytes = (size_t)offsets[count] * 4;
    if (int rc = ensure_scratch(h, 16 + off_bytes + word_bytes)) return rc;
    uint8_t* d = static_cast<uint8_t*>(h->d_scratch);
    const uint32_t st0[4] = {0u, 0xFFFFFFFFu, 0u, 0u};  // [0] rejected envs, [1] min over them of (local index + 1) << 8 | reason
    HIP_TRY(hipMemcpy(d, st0, 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d + 16, offsets, off_bytes, hipMemcpyHostToDevice));
    if (word_bytes) HIP_TRY(hipMemcpy(d + 16 + off_bytes, words, word_bytes, hipMemcpyHostToDevice));
    HIP_TRY(msnake::launch_state_unpack(h->p, h->cfg.rules, env0, count, reinterpret_cast<const uint64_t*>(d + 16),
                                        reinterpret_cast<const int32_t*>(d + 16 + off_bytes),
                                        reinterpret_cast<uint32_t*>(d), nullptr));
    uint32_t st[4];
    HIP_TRY(hipMemcpy(st, d, 16, hipMemcpyDeviceToHost));
    if (st[0] != 0)
        return fail(MSNAKE_E_STATE, "%u env state(s) rejected; first: env %d: %s (the rejected envs were left untouched)", st[0],
                    env0 + (int)(st[1] >> 8) - 1, state_reason(st[1] & 0xFFu));
    return MSNAKE_OK;
}

}  // namespace

int msnake_get_state(msnake_handle h, int32_t env, int32_t* words, int32_t cap_words) {
    if (int rc = check(h)) return rc;
    if (env < 0 || env >= h->p.nenv) return fail(MSNAKE_E_ARG, "env %d out of range", env);
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint64_t> offsets;
    if (int rc = state_offsets(h, env, 1, offsets)) return rc;
    const int32_t need = (int32_t)offsets[1];
    if (!words || cap_words < need) return need;
    if (int rc = state_export(h, env, 1, offsets, words)) return rc;
    return need;
}

int msnake_set_state(msnake_handle h, int32_t env, const int32_t* words, int32_t n) {
    if (int rc = check(h)) return rc;
    if (env < 0 || env >= h->p.nenv) return fail(MSNAKE_E_ARG, "env %d out of range", env);
    if (!words || n < 8) return fail(MSNAKE_E_STATE, "state buffer too short");
    if ((words[7] & 0xFF) != h->p.n_snakes) return fail(MSNAKE_E_STATE, "state has %d snakes, handle has %d", words[7] & 0xFF, h->p.n_snakes);
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const uint64_t offsets[2] = {0, (uint64_t)n};
    return state_import(h, env, 1, offsets, words);
}

int64_t msnake_get_state_all(msnake_handle h, void* buf, size_t cap_bytes) {
    if (int rc = check(h)) return rc;
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const int n = h->p.nenv;
    std::vector<uint64_t> offsets;
    if (int rc = state_offsets(h, 0, n, offsets)) return rc;
    const size_t head = sizeof(BlobHeader) + ((size_t)n + 1) * 8;
    const size_t need = head + (size_t)offsets.back() * 4;
    if (!buf || cap_bytes < need) return (int64_t)need;
    BlobHeader bh = {kBlobMagic, kBlobVersion, n, h->p.dim, h->p.n_snakes, h->p.n_fruits, h->cfg.rules, 0, offsets.back()};
    uint8_t* out = static_cast<uint8_t*>(buf);
    memcpy(out, &bh, sizeof(bh));
    memcpy(out + sizeof(bh), offsets.data(), ((size_t)n + 1) * 8);
    if (int rc = state_export(h, 0, n, offsets, reinterpret_cast<int32_t*>(out + head))) return rc;
    return (int64_t)need;
}

int msnake_state_blob_info(const void* buf, size_t bytes, msnake_blob_info* out) {
    // host-only: header, offset table and length of a msnake_get_state_all blob, before any handle exists
    if (!buf || bytes < sizeof(BlobHeader)) return fail(MSNAKE_E_STATE, "state blob too short");
    BlobHeader bh;
    memcpy(&bh, buf, sizeof(bh));
    if (bh.magic != kBlobMagic || (bh.version != 1u && bh.version != kBlobVersion))
        return fail(MSNAKE_E_STATE, "not an msnake state blob (magic/version)");
    if (bh.num_envs < 1) return fail(MSNAKE_E_STATE, "state blob: num_envs %d", bh.num_envs);
    const size_t n = (size_t)bh.num_envs;
    if ((bytes - sizeof(BlobHeader)) / 8 < n + 1) return fail(MSNAKE_E_STATE, "state blob truncated in its offset table");
    const size_t head = sizeof(BlobHeader) + (n + 1) * 8;
    const uint8_t* offp = static_cast<const uint8_t*>(buf) + sizeof(bh);
    uint64_t prev = 0, last = 0;
    memcpy(&prev, offp, 8);
    memcpy(&last, offp + n * 8, 8);
    // (the word count is bounded BEFORE it is multiplied: 2^62 words * 4 would wrap to 0 and pass a byte comparison)
    if (prev != 0 || last != bh.total_words || bh.total_words > (uint64_t)(bytes - head) / 4)
        return fail(MSNAKE_E_STATE, "state blob truncated or its offset table is inconsistent");
    for (size_t i = 1; i <= n; ++i) {
        uint64_t cur;
        memcpy(&cur, offp + i * 8, 8);
        if (cur < prev) return fail(MSNAKE_E_STATE, "state blob: offsets of env %zu decrease", i - 1);
        prev = cur;
    }
    if (out) {
        out->version = (int32_t)bh.version; out->num_envs = bh.num_envs; out->dim = bh.dim; out->n_snakes = bh.n_snakes;
        out->n_fruits = bh.n_fruits; out->rules = bh.rules; out->total_words = (int64_t)bh.total_words;
    }
    return MSNAKE_OK;
}

int msnake_set_state_all(msnake_handle h, const void* buf, size_t bytes) {
    if (int rc = check(h)) return rc;
    msnake_blob_info bi;
    if (int rc = msnake_state_blob_info(buf, bytes, &bi)) return rc;
    const msnake::StepParams& p = h->p;
    if (bi.num_envs != p.nenv || bi.dim != p.dim || bi.n_snakes != p.n_snakes || bi.rules != h->cfg.rules ||
        (h->cfg.rules != MSNAKE_RULES_ADVERSARIAL && bi.n_fruits != p.n_fruits))
        return fail(MSNAKE_E_STATE, "state blob is for %d envs, dim %d, %d snakes, %d fruits, rules %d; the handle differs",
                    bi.num_envs, bi.dim, bi.n_snakes, bi.n_fruits, bi.rules);
    const size_t n = (size_t)p.nenv, head = sizeof(BlobHeader) + (n + 1) * 8;
    std::vector<uint64_t> offsets(n + 1);
    memcpy(offsets.data(), static_cast<const uint8_t*>(buf) + sizeof(BlobHeader), (n + 1) * 8);
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    return state_import(h, 0, p.nenv, offsets.data(),
                        reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(buf) + head));
}

int msnake_get_stats(msnake_handle h, msnake_stats* out, int32_t reset) {
    if (int rc = check(h)) return rc;
    if (!out) return fail(MSNAKE_E_ARG, "msnake_get_stats: out is NULL");
    DeviceGuard guard(h->cfg.device);
    // the totals live in the env records; sum (and optionally clear) them once every step issued so
    // far, on whatever stream, has finished (no stream handle is remembered between calls)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemsetAsync(h->d_stats, 0, 64, nullptr));
    HIP_TRY(msnake::launch_stats(h->p.hdr, h->p.nenv, h->p.stats, reset ? 1 : 0, nullptr));
    unsigned long long raw[8];
    HIP_TRY(hipMemcpy(raw, h->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->episodes = (int64_t)raw[0];
    out->ep_len_sum = (int64_t)raw[1];
    out->ep_return_sum = (int64_t)raw[2];
    out->env_steps = h->env_steps;
    out->errors = (int64_t)raw[4];
    if (reset) h->env_steps = 0;
    return MSNAKE_OK;
}

const char* msnake_kernel_name(msnake_handle h) {
    if (check(h)) return "";
    return h->kname;  // owned by the handle: valid until msnake_destroy
}

int64_t msnake_algorithmic_bytes_per_env_step(msnake_handle h) {
    if (check(h)) return -1;
    // obs write + one read of per-cell occupancy + actions + reward/done +
    // per-snake and per-env scalar read-modify-write
    const msnake::StepParams& p = h->p;
    return (int64_t)p.S * p.obs_scale * p.obs_scale + (int64_t)p.dim * p.dim + 4 * p.n_snakes + 5 + 16 * p.n_snakes + 16;
}

}  // extern "C"

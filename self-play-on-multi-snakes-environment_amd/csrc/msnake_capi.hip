// msnake_capi.hip -- the C-ABI of include/msnake.h over the HIP kernels.  Host-side glue only:
// argument checks, HBM allocation of the per-env state, the background image, launches, and the
// (blocking, test/checkpoint-path) canonical state import/export.  No CPU fallback: without a HIP
// device every entry point fails with MSNAKE_E_NOGPU / MSNAKE_E_HIP.
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
    hipStream_t last_stream;
    int64_t env_steps;
    void* d_state;
    void* d_stats;
};

namespace {

int check(msnake_handle h) {
    if (!h || h->magic != kMagic) return fail(MSNAKE_E_HANDLE, "invalid or destroyed msnake handle");
    return MSNAKE_OK;
}

int vel_code(int v0, int v1) {
    if (v0 == 1 && v1 == 0) return 1;
    if (v0 == 0 && v1 == 1) return 2;
    if (v0 == -1 && v1 == 0) return 3;
    if (v0 == 0 && v1 == -1) return 4;
    return 0;
}
const int kVel0[5] = {0, 1, 0, -1, 0}, kVel1[5] = {0, 0, 1, 0, -1};

}  // namespace

extern "C" {

int msnake_abi_version(void) { return MSNAKE_ABI_VERSION; }
const char* msnake_last_error(void) { return g_err; }

int msnake_create(const msnake_config* cfg, msnake_handle* out) {
    if (!cfg || !out) return fail(MSNAKE_E_ARG, "msnake_create: NULL argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(msnake_config))
        return fail(MSNAKE_E_ARG, "msnake_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                    sizeof(msnake_config));
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
    if (cfg->obs_scale != 1 && cfg->obs_scale != 4 && cfg->obs_scale != 7)
        return fail(MSNAKE_E_ARG, "obs_scale must be 1, 4 (21->84) or 7 (12->84), got %d", cfg->obs_scale);

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
    p.img_bytes = (p.S * K + 1023) / 1024 * 1024;  // K-fold wide image, whole 1 KiB wave-instructions
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
    const size_t tmpl_bytes = (size_t)p.img_bytes;
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
    for (int r = 0; r < W; ++r)
        for (int c = 0; c < W; ++c)
            if (r == 0 || r == W - 1 || c == 0 || c == W - 1)
                memset(&tmpl[((size_t)r * W + c) * p.C * K], 255, (size_t)p.C * K);
    if ((e = hipMemcpy(const_cast<uint8_t*>(p.tmpl), tmpl.data(), tmpl_bytes, hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipFree(h->d_state); (void)hipFree(h->d_stats);
        free(h);
        return fail(MSNAKE_E_HIP, "uploading the background image failed: %s", hipGetErrorString(e));
    }
    if (const char* epb = getenv("MSNAKE_EPB")) {  // tuning knob: envs (waves) per workgroup, 1..8
        const int v = atoi(epb);
        if (v >= 1 && v <= h->epb) h->epb = v;
    }
    // obs_store_policy: should the observation stores carry the nt (streaming) hint?  Measured on
    // MI355X with the native 16-byte-per-lane copy-out (tools/kbench.py, MSNAKE_NT=0/1), per launch:
    //   <= 32 MiB of observations (<= 8 192 envs at 19x19x3; fits the 8 L2s): 2-3 % faster with nt
    //      (nothing is left to write back when the kernel ends);
    //   64-130 MiB (fits the 256 MB Infinity Cache): 0-4 % slower (a re-used buffer no longer hits);
    //   >= 260 MiB: 33-42 % faster (388 -> 243 us at 262 144 envs: plain stores thrash the cache).
    // The fused x4 / x7 copy-out stores dwords (256 B per wave instruction) and is 2.5x SLOWER with
    // nt, so it never streams; neither does the persistent tape kernel (187 -> 208 us).
    {
        const double obs_mib = (double)p.nenv * p.S * p.obs_scale * p.obs_scale / (1024.0 * 1024.0);
        p.rest.stream_obs = (p.obs_scale == 1 && (obs_mib <= 32.0 || obs_mib >= 192.0)) ? 1u : 0u;
    }
    if (const char* nt = getenv("MSNAKE_NT")) p.rest.stream_obs = atoi(nt) ? 1u : 0u;  // experiment knob
    if (const char* dbg = getenv("MSNAKE_DBG_STAGE")) p.rest.dbg_stage = (uint32_t)atoi(dbg);
    if (const char* dbg = getenv("MSNAKE_DBG_BUF")) p.rest.dbg_buf = reinterpret_cast<unsigned long long*>(strtoull(dbg, nullptr, 0));
    h->magic = kMagic;
    *out = h;
    return MSNAKE_OK;
}

int msnake_destroy(msnake_handle h) {
    if (int rc = check(h)) return rc;
    DeviceGuard guard(h->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->d_state); (void)hipFree(h->d_stats);
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
    hipError_t e = msnake::launch_step(p, h->cfg.rules, mode, h->epb, s);
    if (e != hipSuccess) return fail(MSNAKE_E_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    h->last_stream = s;
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
    p.rest.obs_step_stride = obs_step_stride;
    p.rest.scalar_step_stride = scalar_step_stride;
    DeviceGuard guard(h->cfg.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = msnake::launch_step(p, h->cfg.rules, 3, h->epb, s);
    if (e != hipSuccess) return fail(MSNAKE_E_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    h->last_stream = s;
    h->env_steps += (int64_t)h->p.nenv * n_steps;
    return MSNAKE_OK;
}

int msnake_get_state(msnake_handle h, int32_t env, int32_t* words, int32_t cap_words) {
    if (int rc = check(h)) return rc;
    if (env < 0 || env >= h->p.nenv) return fail(MSNAKE_E_ARG, "env %d out of range", env);
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const msnake::StepParams& p = h->p;
    uint32_t hdr[MSNAKE_HDR_WORDS];
    HIP_TRY(hipMemcpy(hdr, p.hdr + (size_t)env * MSNAKE_HDR_WORDS, sizeof(hdr), hipMemcpyDeviceToHost));
    std::vector<uint16_t> ring((size_t)p.n_snakes * p.rest.cap);
    HIP_TRY(hipMemcpy(ring.data(), p.ring + (size_t)env * p.n_snakes * p.rest.cap, ring.size() * 2, hipMemcpyDeviceToHost));
    const bool adv = h->cfg.rules == MSNAKE_RULES_ADVERSARIAL;
    const int nfr = adv ? (int)hdr[HDR_NLIST] : p.n_fruits;
    std::vector<uint16_t> flist;
    if (adv) {
        flist.resize((size_t)p.fcap);
        HIP_TRY(hipMemcpy(flist.data(), p.flist + (size_t)env * p.fcap, flist.size() * 2, hipMemcpyDeviceToHost));
    }
    int32_t need = 8 + 2 * nfr;
    for (int s = 0; s < p.n_snakes; ++s) need += 6 + 2 * (int32_t)(hdr[SN_A(s)] >> 16);
    if (!words || cap_words < need) return need;
    int32_t k = 0;
    words[k++] = (int32_t)hdr[HDR_T];
    words[k++] = (int32_t)hdr[HDR_CTR_LO];
    words[k++] = (int32_t)hdr[HDR_CTR_HI];
    words[k++] = (int32_t)hdr[HDR_SPARE];
    words[k++] = (int32_t)hdr[HDR_EP_LEN];
    words[k++] = (int32_t)hdr[HDR_EP_RETURN];
    words[k++] = nfr;
    words[k++] = p.n_snakes;
    for (int f = 0; f < nfr; ++f) {
        const uint32_t c = adv ? (uint32_t)flist[(size_t)f] : (hdr[HDR_FRUIT0 + f] & 0xFFFFu);
        words[k++] = (int32_t)(c >> 8) - 1;
        words[k++] = (int32_t)(c & 255u) - 1;
    }
    for (int s = 0; s < p.n_snakes; ++s) {
        const uint32_t w0 = hdr[SN_A(s)], w2 = hdr[SN_C(s)];
        const int hp = (int)(w0 & 0xFFFFu), len = (int)(w0 >> 16), vel = (int)((w2 >> 16) & 7u);
        words[k++] = len;
        words[k++] = kVel0[vel];
        words[k++] = kVel1[vel];
        words[k++] = (int32_t)hdr[SN_B(s)];
        const bool nw = h->cfg.rules == MSNAKE_RULES_NEW_WORLD;
        words[k++] = nw ? (int32_t)((hdr[HDR_FLAGS] >> s) & 1u) : 1;
        words[k++] = nw ? (int32_t)((hdr[HDR_FLAGS] >> (4 + s)) & 1u) : 0;
        for (int i = 0; i < len; ++i) {
            const uint32_t c = ring[(size_t)s * p.rest.cap + (size_t)((hp + i) % p.rest.cap)];
            words[k++] = (int32_t)(c >> 8) - 1;
            words[k++] = (int32_t)(c & 255u) - 1;
        }
    }
    return k;
}

int msnake_set_state(msnake_handle h, int32_t env, const int32_t* words, int32_t n) {
    if (int rc = check(h)) return rc;
    if (env < 0 || env >= h->p.nenv) return fail(MSNAKE_E_ARG, "env %d out of range", env);
    const msnake::StepParams& p = h->p;
    if (!words || n < 8) return fail(MSNAKE_E_STATE, "state buffer too short");
    if (words[7] != p.n_snakes) return fail(MSNAKE_E_STATE, "state has %d snakes, handle has %d", words[7], p.n_snakes);
    const bool adv = h->cfg.rules == MSNAKE_RULES_ADVERSARIAL;
    const int nfr = words[6];
    if (!adv && nfr != p.n_fruits) return fail(MSNAKE_E_STATE, "state has %d fruits, handle has %d", nfr, p.n_fruits);
    if (adv && (nfr < 0 || nfr > p.fcap)) return fail(MSNAKE_E_STATE, "fruit list of %d entries exceeds %d", nfr, p.fcap);
    std::vector<uint16_t> flist(adv ? (size_t)p.fcap : 0, 0), fl0(adv ? 64 : 0, 0);
    uint32_t hdr[MSNAKE_HDR_WORDS] = {0};
    std::vector<uint16_t> ring((size_t)p.n_snakes * p.rest.cap, 0);
    std::vector<uint16_t> body0((size_t)p.n_snakes * 64, 0);
    auto cell = [&](int32_t c0, int32_t c1, uint32_t* out) -> bool {
        if (c0 < -1 || c0 > p.dim || c1 < -1 || c1 > p.dim) return false;
        *out = ((uint32_t)(c0 + 1) << 8) | (uint32_t)(c1 + 1);
        return true;
    };
    int32_t k = 0;
    hdr[HDR_T] = (uint32_t)words[k++];
    hdr[HDR_CTR_LO] = (uint32_t)words[k++];
    hdr[HDR_CTR_HI] = (uint32_t)words[k++];
    hdr[HDR_SPARE] = (uint32_t)words[k++];
    hdr[HDR_EP_LEN] = (uint32_t)words[k++];
    hdr[HDR_EP_RETURN] = (uint32_t)words[k++];
    k += 2;
    if (n < k + 2 * nfr) return fail(MSNAKE_E_STATE, "state buffer truncated in fruits");
    for (int f = 0; f < nfr; ++f, k += 2) {
        uint32_t c;
        if (!cell(words[k], words[k + 1], &c)) return fail(MSNAKE_E_STATE, "fruit %d outside [-1, dim]", f);
        if (adv) {
            flist[(size_t)f] = (uint16_t)c;
            if (f < 64) fl0[(size_t)f] = (uint16_t)c;
        } else {
            hdr[HDR_FRUIT0 + f] = c;
        }
    }
    if (adv) hdr[HDR_NLIST] = (uint32_t)nfr;
    uint32_t flags = 0;
    for (int s = 0; s < p.n_snakes; ++s) {
        if (n < k + 6) return fail(MSNAKE_E_STATE, "state buffer truncated in snake %d", s);
        const int len = words[k], v0 = words[k + 1], v1 = words[k + 2], grow = words[k + 3];
        const int alive = words[k + 4], in_dead = words[k + 5];
        k += 6;
        if (len < 0 || len > p.rest.cap - 2 || n < k + 2 * len) return fail(MSNAKE_E_STATE, "snake %d: bad length %d", s, len);
        uint32_t headc = 0;
        for (int i = 0; i < len; ++i, k += 2) {
            uint32_t c;
            if (!cell(words[k], words[k + 1], &c)) return fail(MSNAKE_E_STATE, "snake %d piece %d outside [-1, dim]", s, i);
            ring[(size_t)s * p.rest.cap + i] = (uint16_t)c;
            if (i < 64) body0[(size_t)s * 64 + i] = (uint16_t)c;
            if (i == 0) headc = c;
        }
        hdr[SN_A(s)] = 0u | ((uint32_t)len << 16);
        hdr[SN_B(s)] = (uint32_t)grow;
        hdr[SN_C(s)] = headc | ((uint32_t)vel_code(v0, v1) << 16);
        if (alive) flags |= 1u << s;
        if (in_dead) flags |= 16u << s;
    }
    hdr[HDR_FLAGS] = flags;
    // every other word stays 0: in particular HDR_PC_VALID, so draws parked in the record for the old
    // counter value are dropped.  The env's logging totals are not part of the canonical state: keep them.
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&hdr[HDR_ACC_EPISODES], p.hdr + (size_t)env * MSNAKE_HDR_WORDS + HDR_ACC_EPISODES,
                      4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(p.hdr + (size_t)env * MSNAKE_HDR_WORDS, hdr, sizeof(hdr), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.ring + (size_t)env * p.n_snakes * p.rest.cap, ring.data(), ring.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.body0 + (size_t)env * p.n_snakes * 64, body0.data(), body0.size() * 2, hipMemcpyHostToDevice));
    if (adv) {
        HIP_TRY(hipMemcpy(p.flist + (size_t)env * p.fcap, flist.data(), flist.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p.fl0 + (size_t)env * 64, fl0.data(), fl0.size() * 2, hipMemcpyHostToDevice));
    }
    return MSNAKE_OK;
}

int msnake_get_stats(msnake_handle h, msnake_stats* out, int32_t reset) {
    if (int rc = check(h)) return rc;
    if (!out) return fail(MSNAKE_E_ARG, "msnake_get_stats: out is NULL");
    DeviceGuard guard(h->cfg.device);
    // the totals live in the env records; sum (and optionally clear) them behind the last step
    HIP_TRY(hipMemsetAsync(h->d_stats, 0, 64, h->last_stream));
    HIP_TRY(msnake::launch_stats(h->p.hdr, h->p.nenv, h->p.stats, reset ? 1 : 0, h->last_stream));
    unsigned long long raw[8];
    HIP_TRY(hipMemcpyAsync(raw, h->d_stats, sizeof(raw), hipMemcpyDeviceToHost, h->last_stream));
    HIP_TRY(hipStreamSynchronize(h->last_stream));
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
    return msnake::step_kernel_name(h->cfg.rules, h->cfg.n_snakes, h->cfg.obs_scale);
}

int64_t msnake_algorithmic_bytes_per_env_step(msnake_handle h) {
    if (check(h)) return -1;
    // obs write + one read of per-cell occupancy + actions + reward/done +
    // per-snake and per-env scalar read-modify-write
    const msnake::StepParams& p = h->p;
    return (int64_t)p.S * p.obs_scale * p.obs_scale + (int64_t)p.dim * p.dim + 4 * p.n_snakes + 5 + 16 * p.n_snakes + 16;
}

}  // extern "C"

// Host-side sanitizer run of the C-ABI glue (msnake_capi.hip) WITHOUT a GPU: everything that runs
// before the first device call -- argument checks, error strings, handle checks, blob header
// validation entry -- under AddressSanitizer + UBSan (host code only: -fno-gpu-sanitize).
// Built and run by tests/test_capi_sanitizers.py.  With a GPU present it stops after the
// no-device checks (host ASan and the HIP runtime are not meant to share a process).
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/msnake.h"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { fprintf(stderr, "FAILED: %s (line %d): %s\n", #cond, __LINE__, msnake_last_error()); return 1; } \
    } while (0)

int main() {
    CHECK(msnake_abi_version() == MSNAKE_ABI_VERSION);
    msnake_handle h = nullptr;
    CHECK(msnake_create(nullptr, &h) == MSNAKE_E_ARG);
    msnake_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = 12;  // ABI mismatch
    CHECK(msnake_create(&cfg, &h) == MSNAKE_E_ARG && strstr(msnake_last_error(), "struct_size"));
    const msnake_config good = {sizeof(msnake_config), 0, 4096, 19, 3, 3, MSNAKE_RULES_SNAKE_ENV, 2000, 1, 1, 7, 0};
    struct { const char* what; msnake_config c; } bad[] = {
        {"num_envs", good}, {"dim", good}, {"dim", good}, {"n_snakes", good}, {"n_fruits", good}, {"rules", good},
        {"max_steps", good}, {"obs_scale", good}, {"num_envs", good},
    };
    bad[0].c.num_envs = 0; bad[1].c.dim = 1; bad[2].c.dim = MSNAKE_MAX_DIM + 1; bad[3].c.n_snakes = 4; bad[4].c.n_fruits = 2;
    bad[5].c.rules = 9; bad[6].c.max_steps = 0; bad[7].c.obs_scale = 3; bad[8].c.num_envs = 0x7FFFFFFF;  // > 4 GB of records
    for (auto& b : bad) {
        CHECK(msnake_create(&b.c, &h) == MSNAKE_E_ARG && h == nullptr);
        CHECK(strstr(msnake_last_error(), b.what) != nullptr);
    }
    // every entry point refuses a NULL handle before it touches anything
    int32_t H, W, C;
    std::vector<int32_t> words(64);
    std::vector<unsigned char> blob(256, 0);
    msnake_stats st;
    CHECK(msnake_obs_shape(nullptr, &H, &W, &C) == MSNAKE_E_HANDLE);
    CHECK(msnake_reset(nullptr, nullptr, nullptr) == MSNAKE_E_HANDLE);
    CHECK(msnake_render(nullptr, nullptr, nullptr) == MSNAKE_E_HANDLE);
    CHECK(msnake_step(nullptr, nullptr, 3, nullptr, nullptr, nullptr, nullptr, nullptr) == MSNAKE_E_HANDLE);
    CHECK(msnake_step_tape(nullptr, nullptr, 3, 1, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr) == MSNAKE_E_HANDLE);
    CHECK(msnake_rollout_tape(nullptr, nullptr, 3, 1, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr) == MSNAKE_E_HANDLE);
    CHECK(msnake_get_state(nullptr, 0, words.data(), 64) == MSNAKE_E_HANDLE);
    CHECK(msnake_set_state(nullptr, 0, words.data(), 64) == MSNAKE_E_HANDLE);
    CHECK(msnake_get_state_all(nullptr, blob.data(), blob.size()) == MSNAKE_E_HANDLE);
    CHECK(msnake_set_state_all(nullptr, blob.data(), blob.size()) == MSNAKE_E_HANDLE);
    CHECK(msnake_get_stats(nullptr, &st, 0) == MSNAKE_E_HANDLE);
    CHECK(msnake_destroy(nullptr) == MSNAKE_E_HANDLE);
    CHECK(msnake_kernel_name(nullptr)[0] == 0 && msnake_algorithmic_bytes_per_env_step(nullptr) == -1);
    CHECK(strstr(msnake_last_error(), "handle") != nullptr);
    // a well-formed configuration gets as far as the device query; without a GPU that is a clean refusal
    const int rc = msnake_create(&good, &h);
    if (rc == MSNAKE_OK) {
        printf("GPU present: stopping after the no-device checks\n");
        // (no msnake_destroy under host ASan on purpose: see the header comment)
    } else {
        CHECK(rc == MSNAKE_E_NOGPU || rc == MSNAKE_E_HIP);
        CHECK(h == nullptr && strlen(msnake_last_error()) > 0);
    }
    printf("CAPI ASAN/UBSAN run clean\n");
    return 0;
}

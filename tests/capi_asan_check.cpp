// Host-side sanitizer run of the C-ABI glue (msnake_capi.hip) WITHOUT a GPU: everything that runs
// before the first device call -- argument checks, error strings, handle checks, blob header
// validation entry -- under AddressSanitizer + UBSan (host code only: -fno-gpu-sanitize).
// Built and run by tests/test_capi_sanitizers.py.  With a GPU present it stops after the
// no-device checks (host ASan and the HIP runtime are not meant to share a process).
#include <algorithm>
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
    CHECK(msnake_state_blob_info(blob.data(), blob.size(), nullptr) == MSNAKE_E_STATE);  // (zeros: no magic)
    CHECK(msnake_get_stats(nullptr, &st, 0) == MSNAKE_E_HANDLE);
    CHECK(msnake_destroy(nullptr) == MSNAKE_E_HANDLE);
    CHECK(msnake_kernel_name(nullptr)[0] == 0 && msnake_algorithmic_bytes_per_env_step(nullptr) == -1);
    CHECK(strstr(msnake_last_error(), "handle") != nullptr);
    // ---- msnake_state_blob_info: the host-side blob checks msnake_set_state_all runs before any device call
    {
        const int n = 5;
        const size_t head = 40 + (size_t)(n + 1) * 8;
        std::vector<unsigned char> b(head + 6 * 4, 0);
        auto put32 = [&](size_t off, uint32_t v) { memcpy(&b[off], &v, 4); };
        auto put64 = [&](size_t off, uint64_t v) { memcpy(&b[off], &v, 8); };
        auto well_formed = [&]() {
            std::fill(b.begin(), b.end(), 0);
            put32(0, 0x5453534Du); put32(4, 2u); put32(8, (uint32_t)n); put32(12, 19u); put32(16, 3u); put32(20, 3u); put32(24, 0u);
            put64(32, 6u);                                             // total_words
            const uint64_t offs[n + 1] = {0, 1, 2, 4, 5, 6};
            for (int i = 0; i <= n; ++i) put64(40 + 8 * (size_t)i, offs[i]);
        };
        msnake_blob_info bi;
        well_formed();
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_OK && bi.num_envs == n && bi.dim == 19 && bi.total_words == 6 && bi.version == 2);
        CHECK(msnake_state_blob_info(b.data(), b.size(), nullptr) == MSNAKE_OK);
        put32(4, 1u);                                                  // version-1 blobs are still read
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_OK && bi.version == 1);
        CHECK(msnake_state_blob_info(b.data(), b.size() - 1, &bi) == MSNAKE_E_STATE && strstr(msnake_last_error(), "truncated"));
        CHECK(msnake_state_blob_info(b.data(), 39, &bi) == MSNAKE_E_STATE);
        CHECK(msnake_state_blob_info(b.data(), 40 + 8 * n, &bi) == MSNAKE_E_STATE && strstr(msnake_last_error(), "offset table"));
        CHECK(msnake_state_blob_info(nullptr, 100, &bi) == MSNAKE_E_STATE);
        put32(4, 7u);
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_E_STATE && strstr(msnake_last_error(), "magic/version"));
        well_formed(); put32(8, 0u);
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_E_STATE);
        well_formed(); put32(8, 0x7FFFFFFFu);                         // more envs than the buffer could hold offsets for
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_E_STATE && strstr(msnake_last_error(), "offset table"));
        well_formed(); put64(40 + 8 * 2, 9u);                         // offsets decrease after env 2
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_E_STATE && strstr(msnake_last_error(), "decrease"));
        well_formed(); put64(40, 1u);                                  // does not start at 0
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_E_STATE);
        // the crafted blob of the round-2 review: every offset but the first = total_words = 2^62, monotone and
        // consistent; 2^62 * 4 wraps to 0 bytes, which a byte-size comparison would have accepted
        well_formed(); put64(32, 1ull << 62);
        for (int i = 1; i <= n; ++i) put64(40 + 8 * (size_t)i, 1ull << 62);
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_E_STATE && strstr(msnake_last_error(), "inconsistent"));
        well_formed(); put64(32, (1ull << 62) + 6); put64(40 + 8 * n, (1ull << 62) + 6);   // (2^62 + 6) * 4 wraps to 24 = the real payload
        CHECK(msnake_state_blob_info(b.data(), b.size(), &bi) == MSNAKE_E_STATE);
    }
    // ABI 2 callers pass the 56-byte prefix of the configuration
    {
        msnake_config old = good;
        old.struct_size = MSNAKE_CONFIG_SIZE_V2; old.num_envs = 0;
        old.envs_per_block = 1234; old.record_policy = -7;            // behind the prefix: must not be read
        CHECK(msnake_create(&old, &h) == MSNAKE_E_ARG && strstr(msnake_last_error(), "num_envs"));
        msnake_config tune = good;
        tune.envs_per_block = 9;
        CHECK(msnake_create(&tune, &h) == MSNAKE_E_ARG && strstr(msnake_last_error(), "envs_per_block"));
        tune = good; tune.record_policy = 3;
        CHECK(msnake_create(&tune, &h) == MSNAKE_E_ARG && strstr(msnake_last_error(), "record_policy"));
        tune = good; tune.tape_store_policy = 3;
        CHECK(msnake_create(&tune, &h) == MSNAKE_E_ARG && strstr(msnake_last_error(), "store_policy"));
    }
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

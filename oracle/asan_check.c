/* Sanitizer run of the CPU oracle (test infrastructure): gcc -fsanitize=address,undefined.
 * Steps every rule set with random actions, with and without auto-reset, export/import round trips. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "snake_oracle.c"

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void) {
    uint32_t seed = 12345;
    const int cfgs[][6] = {/* rules dim ns nf auto envs */ {0, 19, 3, 3, 1, 64}, {0, 10, 1, 1, 1, 64}, {0, 3, 1, 1, 1, 16},
                           {1, 19, 3, 5, 1, 64}, {1, 10, 4, 4, 0, 32}, {1, 10, 2, 0, 0, 16},
                           {2, 10, 3, 3, 1, 64}, {2, 10, 2, 2, 0, 32}, {2, 19, 3, 3, 1, 32}};
    for (unsigned c = 0; c < sizeof(cfgs) / sizeof(cfgs[0]); ++c) {
        orc_config cfg = {cfgs[c][5], cfgs[c][1], cfgs[c][2], cfgs[c][3], cfgs[c][0], 60, cfgs[c][4], 0, 7 + c, 1000 * c};
        void* h = orc_create(&cfg);
        if (!h) { printf("create failed for cfg %u\n", c); return 1; }
        int32_t H, W, C;
        orc_obs_shape(h, &H, &W, &C);
        const int n = cfg.num_envs;
        uint8_t* obs = malloc((size_t)n * H * W * C);
        float* rew = malloc(sizeof(float) * n); uint8_t* done = malloc(n);
        int32_t* ns = malloc(4 * n); float* er = malloc(4 * n); int32_t* el = malloc(4 * n);
        int32_t* act = malloc(4 * n * 4);
        orc_reset(h, obs);
        long episodes = 0;
        for (int t = 0; t < 400; ++t) {
            for (int i = 0; i < n * 4; ++i) act[i] = (int32_t)(lcg(&seed) % 6) - (lcg(&seed) % 50 == 0);
            orc_step(h, act, 4, obs, rew, done, ns, er, el);
            for (int i = 0; i < n; ++i) episodes += done[i];
            if (t % 37 == 0) {  /* export / import round trip */
                int32_t need = orc_export_state(h, t % n, NULL, 0);
                int32_t* buf = malloc(4 * need);
                orc_export_state(h, t % n, buf, need);
                if (orc_import_state(h, (t + 1) % n, buf, need) != 0) { printf("import failed\n"); return 1; }
                free(buf);
            }
        }
        orc_render(h, obs);
        printf("cfg %u (rules %d dim %d ns %d nf %d auto %d): %ld episode ends, ok\n", c, cfgs[c][0], cfgs[c][1], cfgs[c][2],
               cfgs[c][3], cfgs[c][4], episodes);
        free(obs); free(rew); free(done); free(ns); free(er); free(el); free(act);
        orc_destroy(h);
    }
    printf("ASAN/UBSAN run clean\n");
    return 0;
}

// msnake_kernels.hip -- the batched multi-snake environment step for gfx950 (MI355X, CDNA4).
//
// One 64-lane wavefront owns one environment for the whole step; a 256-thread workgroup carries
// four independent envs and never executes a workgroup barrier.  Game logic runs on wave-uniform
// (scalar) values; the lanes cooperate on the O(body) and O(dim^2) jobs:
//   * body pieces are lane-distributed (lane l = piece l): collisions are 64 compares + a ballot
//     (head-vs-piece matrix), fruit eating is a ballot over lane-resident fruits;
//   * fruit respawn = occupancy bytes in LDS -> per-chunk ballots parked in lanes -> k-th free cell
//     by popcount / mbcnt (no loop over cells);
//   * the observation is composed in LDS: a precomputed wall/background image (L2-resident,
//     16 byte-shifted copies so that every LDS chunk matches a 16-byte ALIGNED global chunk) is
//     copied with 16-byte LDS writes, fruit and body pixels are painted over it in reference order,
//     and the image leaves as coalesced 16-byte global stores (1 KiB per wave instruction);
//   * RNG is Philox4x32-10, one block per 4 draws, evaluated on the scalar unit.
// Integer / byte work bounded by HBM writes of the observation tensor; no MFMA on purpose.
//
// Rules restated (reference paths under /root/reference/src/gym-snake/gym_snake/):
//   snake_env  : envs/snake_multiple_test.py:97-232         new_world : core/new_world.py:25-158 +
//   adversarial: envs/snake_adversarial_env.py:95-201                   envs/snake_multiple_env_new.py:27-50
//   vec layer  : baselines/common/vec_env/subproc_vec_env.py:13-16, baselines/bench/monitor.py:57-78
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msnake_internal.h"

namespace msnake {

// ------------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
// hipcc 7.2 exposes no writelane builtin; a lane-id select does the same job
#define wrlane(old, sval, l) (lane == (l) ? (uint32_t)(sval) : (uint32_t)(old))
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// lanes of one wave exchange data through LDS: DS operations of a wave execute in program order,
// this only stops the compiler from moving/forwarding accesses across the exchange point
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 on wave-uniform operands (rocRAND's seed / subsequence / offset convention)
// ------------------------------------------------------------------------------------------------
struct Rng {
    uint32_t k0, k1, e_lo, e_hi;  // key = seed, counter.zw = global env id
    uint32_t ctr_lo, ctr_hi;      // draws consumed so far
    uint32_t b0, b1, b2, b3, blk; // cached block
    bool valid;
};

__device__ __forceinline__ void philox_block(Rng& r, uint32_t c0, uint32_t c1) {
    uint32_t c2 = r.e_lo, c3 = r.e_hi, k0 = r.k0, k1 = r.k1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r.b0 = c0; r.b1 = c1; r.b2 = c2; r.b3 = c3;
}

// randint(n) = (u32 * n) >> 32 on draw number ctr, then ctr += 1
__device__ __forceinline__ uint32_t randint(Rng& r, uint32_t n) {
    const uint32_t blk_lo = (r.ctr_lo >> 2) | (r.ctr_hi << 30), blk_hi = r.ctr_hi >> 2;
    if (!r.valid || blk_lo != r.blk) {
        philox_block(r, blk_lo, blk_hi);
        r.blk = blk_lo;
        r.valid = true;
    }
    const uint32_t sel = r.ctr_lo & 3u;
    const uint32_t u = sel == 0 ? r.b0 : sel == 1 ? r.b1 : sel == 2 ? r.b2 : r.b3;
    r.ctr_lo += 1;
    r.ctr_hi += (r.ctr_lo == 0);
    return (uint32_t)(((uint64_t)u * n) >> 32);
}

// ------------------------------------------------------------------------------------------------
// cells: 16 bits, (row << 8) | col in PADDED observation coordinates: row = c0 + 1, col = c1 + 1,
// so the grid is 1..dim and one step outside lands on the wall ring 0 / dim+1.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_step(int v) {  // velocity code -> cell delta
    return v == 1 ? 256 : v == 2 ? 1 : v == 3 ? -256 : v == 4 ? -1 : 0;
}
__device__ __forceinline__ bool in_grid(uint32_t cell, int dim) {
    const uint32_t r = cell >> 8, c = cell & 255u;
    return r >= 1 && r <= (uint32_t)dim && c >= 1 && c <= (uint32_t)dim;
}

// The step kernel. RESET_ONLY = msnake_reset / msnake_render (no game logic).
//   mode 0: step, 1: reset every env, 2: render only
template <int RULES, int MODE>
__global__ __launch_bounds__(MSNAKE_BLOCK_THREADS) void msnake_step_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)uni(threadIdx.x >> 6);
    const int e = (int)(blockIdx.x * (blockDim.x >> 6)) + wave;
    if (e >= p.nenv) return;

    const int ns = p.n_snakes, nf = p.n_fruits, dim = p.dim, W = dim + 2, C = p.C, cap = p.cap;
    const int n2 = dim * dim;
    uint8_t* img = smem + (size_t)wave * p.lds_per_wave;
    uint8_t* occ = img + p.occ_off;

    // ---- 0. independent loads: header, actions, background image ----------------------------
    uint32_t* hdr_g = p.hdr + (size_t)e * MSNAKE_HDR_WORDS;
    uint32_t hv = hdr_g[lane];  // lane l holds header word l; lanes 32+f hold fruit f
    int actv = 0;
    if (MODE == 0 && lane < ns) actv = p.actions[(size_t)e * p.action_stride + lane];

    uint8_t* obs_env = p.obs ? p.obs + (size_t)e * p.S : nullptr;
    const uint32_t a = (uint32_t)((uintptr_t)obs_env & 15u);  // misalignment of this env's image
    uint8_t* obs_al = obs_env - a;
    const int nchunks = (int)((a + (uint32_t)p.S + 15u) >> 4);
    if (p.obs) {
        const uint4* tsrc = reinterpret_cast<const uint4*>(p.tmpl + (size_t)a * p.tmpl_stride);
        for (int k0 = 0; k0 < nchunks; k0 += 256) {
            uint4 t0, t1, t2, t3;
            const int ka = k0 + lane, kb = ka + 64, kc = ka + 128, kd = ka + 192;
            if (ka < nchunks) t0 = tsrc[ka];
            if (kb < nchunks) t1 = tsrc[kb];
            if (kc < nchunks) t2 = tsrc[kc];
            if (kd < nchunks) t3 = tsrc[kd];
            uint4* dst = reinterpret_cast<uint4*>(img);
            if (ka < nchunks) dst[ka] = t0;
            if (kb < nchunks) dst[kb] = t1;
            if (kc < nchunks) dst[kc] = t2;
            if (kd < nchunks) dst[kd] = t3;
        }
    }

    // ---- 1. header -> wave-uniform registers -------------------------------------------------
    uint32_t t = rdlane(hv, HDR_T);
    Rng rng;
    rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
    {
        const uint64_t gid = p.env_id_base + (uint64_t)e;
        rng.e_lo = (uint32_t)gid; rng.e_hi = (uint32_t)(gid >> 32);
    }
    rng.ctr_lo = rdlane(hv, HDR_CTR_LO); rng.ctr_hi = rdlane(hv, HDR_CTR_HI);
    rng.valid = false; rng.blk = 0; rng.b0 = rng.b1 = rng.b2 = rng.b3 = 0;
    float ep_ret = __uint_as_float(rdlane(hv, HDR_EP_RETURN));
    uint32_t ep_len = rdlane(hv, HDR_EP_LEN);
    uint32_t flags = rdlane(hv, HDR_FLAGS);
    uint32_t err = 0;

    int hp[MSNAKE_MAX_SNAKES], len[MSNAKE_MAX_SNAKES], grow[MSNAKE_MAX_SNAKES];
    int head[MSNAKE_MAX_SNAKES], vel[MSNAKE_MAX_SNAKES], nvel[MSNAKE_MAX_SNAKES];
    int shift[MSNAKE_MAX_SNAKES];     // 1 while a snake that will move has not moved yet
    bool stored[MSNAKE_MAX_SNAKES];   // new head must be written to the ring
    uint32_t cellreg[MSNAKE_MAX_SNAKES];  // lane l: piece (l - shift) of snake s (chunk 0)
    uint16_t* ring_g = p.ring + (size_t)e * ns * cap;

#pragma unroll
    for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s) {
        hp[s] = len[s] = grow[s] = head[s] = vel[s] = nvel[s] = shift[s] = 0;
        stored[s] = false;
        cellreg[s] = MSNAKE_NO_CELL;
        if (s < ns) {
            const uint32_t w0 = rdlane(hv, HDR_SNAKE0 + 4 * s), w2 = rdlane(hv, HDR_SNAKE0 + 4 * s + 2);
            hp[s] = (int)(w0 & 0xFFFFu); len[s] = (int)(w0 >> 16);
            grow[s] = (int)rdlane(hv, HDR_SNAKE0 + 4 * s + 1);
            head[s] = (int)(w2 & 0xFFFFu); vel[s] = (int)((w2 >> 16) & 7u);
            nvel[s] = vel[s];
            if (MODE == 0 && len[s] > 0) {
                // turn now (depends only on this snake): [S]:108-115 == [N]:34-41 == [A]:106-113
                const int act = (int)rdlane((uint32_t)actv, s);
                if (act >= 1 && act <= 4 && vel[s] != ((act + 1) & 3) + 1) nvel[s] = act;
                // snake_env moves only with a non-zero velocity ([S]:119); new_world always
                // inserts a head, even a duplicate of itself ([N]:43-48,153)
                shift[s] = (RULES == MSNAKE_RULES_NEW_WORLD) ? 1 : (nvel[s] != 0);
            }
            // body chunk 0: lane l <- piece (l - shift) of the PRE-move body
            const int q = lane - shift[s];
            if (q >= 0 && q < len[s]) {
                int idx = hp[s] + q;
                idx = idx >= cap ? idx - cap : idx;
                cellreg[s] = ring_g[(size_t)s * cap + idx];
            }
        }
    }

    // piece iterator: f(i, cell) for every piece i of snake s in its CURRENT logical state
    auto for_each_piece = [&](int s_hp, int s_len, int s_shift, uint32_t s_reg, const uint16_t* s_ring, auto&& f) {
        const int i0 = lane - s_shift;
        if (i0 >= 0 && i0 < s_len) f(i0, s_reg);
        for (int base = 64 - s_shift; base < s_len; base += 64) {
            const int i = base + lane;
            if (i < s_len) {
                int idx = s_hp + i;
                idx = idx >= cap ? idx - cap : idx;
                f(i, (uint32_t)s_ring[idx]);
            }
        }
    };

    // ---- fruit respawn: [S]:202-217 safe_choose_cell == [N]:235-247 get_safe_cell ------------
    uint64_t freemask = 0;  // lane c: ballot of the free cells of index chunk c
    int nfree = 0;
    auto build_free = [&]() {
        for (int i = lane * 4; i < p.occ_bytes; i += 256) *reinterpret_cast<uint32_t*>(occ + i) = 0u;
        wave_sync();
#pragma unroll
        for (int j = 0; j < MSNAKE_MAX_SNAKES; ++j)
            if (j < ns && len[j] > 0)
                for_each_piece(hp[j], len[j], shift[j], cellreg[j], ring_g + (size_t)j * cap,
                               [&](int, uint32_t cell) {
                                   // used = c1*dim + c0; out-of-grid heads alias or fall outside
                                   const int used = ((int)(cell & 255u) - 1) * dim + ((int)(cell >> 8) - 1);
                                   if ((uint32_t)used < (uint32_t)n2) occ[used] = 1;
                               });
        wave_sync();
        nfree = 0; freemask = 0;
        for (int c = 0; c * 64 < n2; ++c) {
            const int idx = c * 64 + lane;
            const uint64_t m = ballot(idx < n2 && occ[idx] == 0);
            nfree += __builtin_popcountll(m);
            if (lane == c) freemask = m;
        }
        wave_sync();
    };
    auto safe_cell = [&]() -> uint32_t {
        int x = 0;
        if (nfree > 0) {
            int k = (int)randint(rng, (uint32_t)nfree);
            for (int c = 0; c * 64 < n2; ++c) {
                const uint64_t m = ((uint64_t)rdlane((uint32_t)(freemask >> 32), c) << 32) |
                                   rdlane((uint32_t)freemask, c);
                const int cnt = __builtin_popcountll(m);
                if (k < cnt) {
                    const uint64_t sel = ballot(((m >> lane) & 1ull) && (int)mbcnt(m) == k);
                    x = c * 64 + (__builtin_ffsll((long long)sel) - 1);
                    break;
                }
                k -= cnt;
            }
        }
        return (uint32_t)(((x % dim + 1) << 8) | (x / dim + 1));
    };

    float reward = 0.0f;
    bool done = false;
    int num_alive = 0;

    // ---- reset: [S]:219-232 / [NE]:27-33 -> [N]:55-73 ------------------------------------------
    auto do_reset = [&]() {
#pragma unroll
        for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
            if (s < ns) { len[s] = 0; shift[s] = 0; }
#pragma unroll
        for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
            if (s < ns) {
                const uint32_t c0 = randint(rng, (uint32_t)dim), c1 = randint(rng, (uint32_t)dim);
                head[s] = (int)(((c0 + 1) << 8) | (c1 + 1));
                hp[s] = 0; len[s] = 1; vel[s] = 0; grow[s] = 3; stored[s] = true;
                cellreg[s] = lane == 0 ? (uint32_t)head[s] : MSNAKE_NO_CELL;
                if (RULES != MSNAKE_RULES_NEW_WORLD) {  // snake cell, then its fruit, unconstrained
                    const uint32_t f0 = randint(rng, (uint32_t)dim), f1 = randint(rng, (uint32_t)dim);
                    if (lane == 32 + s) hv = ((f0 + 1) << 8) | (f1 + 1);
                }
            }
        if (RULES == MSNAKE_RULES_NEW_WORLD) {  // all snakes first, then n_fruits safe cells
            build_free();
            for (int f = 0; f < nf; ++f) {
                const uint32_t c = safe_cell();
                if (lane == 32 + f) hv = c;
            }
        }
        t = 0;
        flags = (1u << ns) - 1u;  // alive bits set, dead_snakes empty
    };

    if (MODE == 1) {
        do_reset();
        ep_ret = 0.0f; ep_len = 0;
    }

    if (MODE == 0) {
        // ---- 2. sequential snake updates (order matters: respawn sees earlier snakes moved) ---
#pragma unroll
        for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s) {
            if (s < ns && len[s] > 0 && shift[s] != 0) {
                const int nh = head[s] + cell_step(nvel[s]);
                // fruits equal to the new head, as a bit mask over fruit indices
                const uint64_t em = ballot(lane >= 32 && lane < 32 + nf && (hv & 0xFFFFu) == (uint32_t)nh) >> 32;
                int g = grow[s], l = len[s];
                if (RULES == MSNAKE_RULES_NEW_WORLD) {
                    // [N]:143-150: the pop test sits inside the per-fruit loop
                    for (int f = 0; f < nf; ++f) {
                        if ((em >> f) & 1ull) g += 2;
                        if (l >= g) l -= 1;
                    }
                } else {
                    g += 2 * __builtin_popcountll(em);  // [S]:126-132
                    if (l >= g) l -= 1;                 // [S]:134-135
                }
                // insert(0, head)
                l += 1;
                if (l > cap - 1) { l = cap - 1; err = 1; }
                hp[s] = hp[s] == 0 ? cap - 1 : hp[s] - 1;
                len[s] = l; head[s] = nh; grow[s] = g; vel[s] = nvel[s];
                shift[s] = 0; stored[s] = true;
                if (lane == 0) cellreg[s] = (uint32_t)nh;
                if (s == 0) reward = (float)__builtin_popcountll(em);
                if (em != 0) {  // slow path: respawn each eaten fruit, in index order
                    build_free();
                    uint64_t m = em;
                    while (m) {
                        const int f = __builtin_ffsll((long long)m) - 1;
                        m &= m - 1;
                        const uint32_t c = safe_cell();
                        if (lane == 32 + f) hv = c;
                    }
                }
            }
        }

        // ---- 3. head-vs-piece matrix: hit[s][j] = some piece of j (other than s's own head)
        //         lies on s's head ------------------------------------------------------------
        bool hitl[MSNAKE_MAX_SNAKES][MSNAKE_MAX_SNAKES];
#pragma unroll
        for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
#pragma unroll
            for (int j = 0; j < MSNAKE_MAX_SNAKES; ++j) hitl[s][j] = false;
#pragma unroll
        for (int j = 0; j < MSNAKE_MAX_SNAKES; ++j)
            if (j < ns && len[j] > 0)
                for_each_piece(hp[j], len[j], shift[j], cellreg[j], ring_g + (size_t)j * cap,
                               [&](int i, uint32_t cell) {
#pragma unroll
                                   for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
                                       if (s < ns)
                                           hitl[s][j] |= (cell == (uint32_t)head[s]) && !(j == s && i == 0);
                               });
        bool hit[MSNAKE_MAX_SNAKES][MSNAKE_MAX_SNAKES];
#pragma unroll
        for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
#pragma unroll
            for (int j = 0; j < MSNAKE_MAX_SNAKES; ++j)
                hit[s][j] = (s < ns && j < ns && len[s] > 0) ? (ballot(hitl[s][j]) != 0) : false;

        // ---- 4. aliveness, reward, done -----------------------------------------------------
        if (RULES == MSNAKE_RULES_NEW_WORLD) {
            // [N]:101-107 + :111-132: in snake order, clearing bodies as it goes
            bool done0 = false;
#pragma unroll
            for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
                if (s < ns) {
                    bool alive;
                    if (len[s] == 0) alive = false;
                    else if (!in_grid((uint32_t)head[s], dim)) { len[s] = 0; alive = false; }
                    else {
                        bool other = false;
#pragma unroll
                        for (int j = 0; j < MSNAKE_MAX_SNAKES; ++j)
                            if (j < ns && j != s && len[j] > 0 && hit[s][j]) other = true;
                        if (other) { len[s] = 0; alive = false; }
                        else alive = !hit[s][s];  // self hit: not alive, body kept
                    }
                    flags = alive ? (flags | (1u << s)) : (flags & ~(1u << s));
                    if (!alive) flags |= (16u << s);  // dead_snakes, append-once
                    if (s == 0) done0 = alive;        // [N]:107 (sic)
                }
            if (done0) reward = -1.0f;  // [NE]:39-40
            t += 1;
            done = (t >= (uint32_t)p.max_steps) || done0;
            num_alive = ns - __builtin_popcount((flags >> 4) & 15u);
        } else {
            // [S]:178-197: simultaneous
            bool dead[MSNAKE_MAX_SNAKES];
            int ndead = 0;
#pragma unroll
            for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s) {
                dead[s] = false;
                if (s < ns) {
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < MSNAKE_MAX_SNAKES; ++j) any |= hit[s][j];
                    dead[s] = len[s] == 0 || !in_grid((uint32_t)head[s], dim) || any;
                    ndead += dead[s];
                }
            }
#pragma unroll
            for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
                if (s < ns && dead[s]) len[s] = 0;
            const bool main_dead = len[0] == 0;
            if (main_dead) reward = -1.0f;
            t += 1;
            done = (t >= (uint32_t)p.max_steps) || main_dead;
            num_alive = ns - ndead;
        }

        // ---- 5. vec layer: episode statistics and auto reset ---------------------------------
        ep_ret += reward;
        ep_len += 1;
        float out_ret = 0.0f;
        uint32_t out_len = 0;
        if (done) {
            out_ret = ep_ret; out_len = ep_len;
            if (lane == 0) {
                atomicAdd(&p.stats[0], 1ull);
                atomicAdd(&p.stats[1], (unsigned long long)ep_len);
                atomicAdd(&p.stats[2], (unsigned long long)(long long)ep_ret);
            }
            if (p.auto_reset) {
                ep_ret = 0.0f; ep_len = 0;
                do_reset();
            }
        }
        if (lane == 0) {
            p.rew[e] = reward;
            p.done[e] = done ? 1 : 0;
            if (p.info) {
                int4 iv;
                iv.x = (int)__float_as_uint(out_ret); iv.y = (int)out_len; iv.z = num_alive; iv.w = done ? 1 : 0;
                reinterpret_cast<int4*>(p.info)[e] = iv;
            }
            if (err) atomicAdd(&p.stats[4], 1ull);
        }
    }

    // ---- 6. paint the observation over the background, in reference order ----------------------
    if (p.obs) {
        wave_sync();
        uint8_t* px = img + a;
        // fruits first ([S]:43-44): red in every view; the background is already black
        if (lane >= 32 && lane < 32 + nf) {
            const uint32_t cell = hv & 0xFFFFu;
            if (in_grid(cell, dim)) {
                const int off = ((int)(cell >> 8) * W + (int)(cell & 255u)) * C;
                for (int v = 0; v < p.views; ++v) px[off + 3 * v] = 255;
            }
        }
        // snakes in index order, head over body ([S]:46-50, draw_snake :24-33)
#pragma unroll
        for (int j = 0; j < MSNAKE_MAX_SNAKES; ++j) {
            if (j < ns && len[j] > 0 && (RULES != MSNAKE_RULES_NEW_WORLD || ((flags >> j) & 1u))) {
                for_each_piece(hp[j], len[j], shift[j], cellreg[j], ring_g + (size_t)j * cap,
                               [&](int i, uint32_t cell) {
                                   if (!in_grid(cell, dim)) return;
                                   const int off = ((int)(cell >> 8) * W + (int)(cell & 255u)) * C;
                                   const bool hd = i == 0;
                                   for (int v = 0; v < p.views; ++v) {
                                       const bool self = v == j;
                                       // body/head: self green (0,204,0)/(191,242,191), other blue (0,51,204)/(128,154,230)
                                       px[off + 3 * v + 0] = hd ? (self ? 191 : 128) : 0;
                                       px[off + 3 * v + 1] = hd ? (self ? 242 : 154) : (self ? 204 : 51);
                                       px[off + 3 * v + 2] = hd ? (self ? 191 : 230) : (self ? 0 : 204);
                                   }
                               });
            }
        }
        wave_sync();
        // ---- 7. LDS image -> HBM: 16-byte aligned chunks, 1 KiB per wave instruction ----------
        const uint32_t lo_b = a, hi_b = a + (uint32_t)p.S;
        for (int k = lane; k < nchunks; k += 64) {
            const uint32_t b0 = (uint32_t)k << 4;
            if (b0 >= lo_b && b0 + 16u <= hi_b)
                reinterpret_cast<uint4*>(obs_al)[k] = reinterpret_cast<const uint4*>(img)[k];
        }
        if (lane < 32) {  // the (at most two) chunks shared with the neighbouring envs: byte stores
            const int kk = lane < 16 ? 0 : nchunks - 1;
            const uint32_t b0 = (uint32_t)kk << 4, b = b0 + (uint32_t)(lane & 15);
            const bool full = b0 >= lo_b && b0 + 16u <= hi_b;
            if (!full && b >= lo_b && b < hi_b) obs_al[b] = img[b];
        }
    }

    // ---- 8. state write-back --------------------------------------------------------------------
    if (MODE != 2) {
#pragma unroll
        for (int s = 0; s < MSNAKE_MAX_SNAKES; ++s)
            if (s < ns) {
                if (stored[s] && lane == 0) ring_g[(size_t)s * cap + hp[s]] = (uint16_t)head[s];
                hv = wrlane(hv, (uint32_t)hp[s] | ((uint32_t)len[s] << 16), HDR_SNAKE0 + 4 * s);
                hv = wrlane(hv, (uint32_t)grow[s], HDR_SNAKE0 + 4 * s + 1);
                hv = wrlane(hv, (uint32_t)head[s] | ((uint32_t)vel[s] << 16), HDR_SNAKE0 + 4 * s + 2);
            }
        hv = wrlane(hv, t, HDR_T);
        hv = wrlane(hv, rng.ctr_lo, HDR_CTR_LO);
        hv = wrlane(hv, rng.ctr_hi, HDR_CTR_HI);
        hv = wrlane(hv, __float_as_uint(ep_ret), HDR_EP_RETURN);
        hv = wrlane(hv, ep_len, HDR_EP_LEN);
        hv = wrlane(hv, flags, HDR_FLAGS);
        hdr_g[lane] = hv;
    }
}

// ------------------------------------------------------------------------------------------------
// launch glue (called from the C-ABI in msnake_capi.hip)
// ------------------------------------------------------------------------------------------------
template <int RULES>
static hipError_t launch_rules(const StepParams& p, int mode, int epb, hipStream_t stream) {
    const dim3 grid((unsigned)((p.nenv + epb - 1) / epb));
    const dim3 block(64u * (unsigned)epb);
    const size_t lds = (size_t)p.lds_per_wave * (size_t)epb;
    switch (mode) {
        case 0: hipLaunchKernelGGL((msnake_step_kernel<RULES, 0>), grid, block, lds, stream, p); break;
        case 1: hipLaunchKernelGGL((msnake_step_kernel<RULES, 1>), grid, block, lds, stream, p); break;
        default: hipLaunchKernelGGL((msnake_step_kernel<RULES, 2>), grid, block, lds, stream, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_step(const StepParams& p, int rules, int mode, int epb, hipStream_t stream) {
    if (epb < 1 || epb > MSNAKE_MAX_ENVS_PER_BLOCK) return hipErrorInvalidValue;
    switch (rules) {
        case MSNAKE_RULES_SNAKE_ENV: return launch_rules<MSNAKE_RULES_SNAKE_ENV>(p, mode, epb, stream);
        case MSNAKE_RULES_NEW_WORLD: return launch_rules<MSNAKE_RULES_NEW_WORLD>(p, mode, epb, stream);
        default: return hipErrorInvalidValue;
    }
}

const char* step_kernel_name(int rules) {
    return rules == MSNAKE_RULES_NEW_WORLD ? "msnake_step_kernel<1, 0>" : "msnake_step_kernel<0, 0>";
}

}  // namespace msnake

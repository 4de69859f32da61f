// msnake_kernels.hip -- the batched multi-snake environment step for gfx950 (MI355X, CDNA4).
//
// One 64-lane wavefront owns one environment for the whole step; a workgroup carries eight or four
// independent envs and never executes a workgroup barrier.  Design points:
//   * ONE memory round trip before the wave can decide: the env record (lane l <-> word l; 128 or 256
//     bytes), the 64-slot body ring of every snake (lane l <-> slot l) and the actions are all at
//     addresses that depend only on the env index, so they are issued together at kernel entry.
//     Only bodies longer than 64 cells touch their overflow ring (dependent loads, rare).
//   * the env record stays in ONE VGPR for the whole kernel: scalar game logic reads fields with
//     v_readlane and writes them back with v_writelane, so almost nothing is live in SGPRs; the three
//     per-snake columns of the record (lane s = snake s after a DPP row shift) let all snakes turn,
//     move, eat and grow at once on the VALU, and go back by one row/bank-masked DPP move each;
//   * per-lane decisions are lane masks: compares write SGPR pairs directly (llvm.amdgcn.icmp), the
//     mask arithmetic runs on the scalar unit, a mask becomes a predicate again for free (inverse ballot);
//   * body pieces are lane-distributed: a move overwrites ONE ring slot (the new head, v_writelane),
//     collisions are compares into lane masks (head-vs-piece matrix), fruit eating is one compare per
//     snake of the fruit lanes against its new head;
//   * fruit respawn (slow path): one occupancy BIT per cell in LDS (ds_or) -> lane c owns the 64
//     cells of chunk c as a register mask -> DPP prefix sum of the per-lane free counts -> k-th free
//     cell; Philox4x32-10 on the VALU, lane l computing draw ctr + l (one evaluation = 64 draws);
//   * the observation is composed in LDS: the wall/background image (L1/L2-resident) goes memory ->
//     LDS directly (global_load_lds_dwordx4, no data VGPRs), fruit and body pixels are painted over
//     it in reference order, and the image leaves as 16-byte-per-lane global stores, 1 KiB of whole
//     128-byte lines per wave instruction: the image is composed at ITS OWN alignment in LDS (16
//     pre-shifted backgrounds), so both sides of every chunk are 16-byte aligned;
//   * write-back = what changed: the record, one 32-byte sector per moving snake, the outputs.
// Integer / byte work bounded by the HBM writes of the observation tensor; no MFMA on purpose.
//
// Rules restated (reference paths under /root/reference/src/gym-snake/gym_snake/):
//   snake_env  : envs/snake_multiple_test.py:97-232         new_world : core/new_world.py:25-158 +
//   adversarial: envs/snake_adversarial_env.py:95-201                   envs/snake_multiple_env_new.py:27-50
//   vec layer  : baselines/common/vec_env/subproc_vec_env.py:13-16, baselines/bench/monitor.py:57-78
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "msnake_internal.h"

namespace msnake {

// ------------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
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
// within each row of 16 lanes: lane l <- lane l+N (row_shl) / lane l-N (row_shr); lanes without a
// source keep `old`
template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x100 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x110 + N, 0xF, 0xF, false);
}
// v_mov_b32 dpp restricted to some rows (ROWS) and some groups of four lanes within each row (BANKS);
// every other lane keeps `old`: a column of the env record is replaced by ONE instruction
template <int CTRL, int ROWS, int BANKS>
__device__ __forceinline__ uint32_t dpp_into(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWS, BANKS, false);
}
constexpr int DPP_IDENTITY = 0xE4;  // quad_perm:[0,1,2,3]
// lane compares straight to a 64-bit lane mask (v_cmp into an SGPR pair), and a lane mask back to a
// per-lane predicate without an instruction: `ballot(b)` of a bool that also has other uses costs a
// v_cndmask + v_cmp pair, `(mask >> lane) & 1` a shift, an and and a compare
constexpr int CMP_EQ = 32, CMP_NE = 33, CMP_UGT = 34, CMP_UGE = 35, CMP_ULT = 36;
template <int PRED>
__device__ __forceinline__ uint64_t lanes_where(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, PRED); }
__device__ __forceinline__ bool in_mask(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// 1 << BIT if the (wave-uniform) lane mask is not empty, else 0: two scalar instructions.  (Written in C, a
// uniform `m != 0` that is used as a number takes a detour through a VGPR.)
template <int BIT>
__device__ __forceinline__ uint32_t bit_if_any(uint64_t m) {
    uint32_t b;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, %2, 0" : "=s"(b) : "s"(m), "n"(1u << BIT) : "scc");
    return b;
}
// lane l <- lane l-1 (lane 0 keeps its value): v_mov_b32 dpp wave_shr:1
__device__ __forceinline__ uint32_t shift_up1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false);
}

// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1,2,4,8),
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 (lanes without a source add 0)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (rocRAND's seed / subsequence / offset convention): draw i of global env g is word
// (i & 3) of philox(counter = {i>>2, g}, key = seed).  Evaluated on the VALU with lane l computing
// draw (base + l): one evaluation yields the next 64 draws of the env (a reset needs 12).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t philox_draws(uint32_t base_lo, uint32_t base_hi, int lane, uint32_t g_lo,
                                                 uint32_t g_hi, uint32_t k0, uint32_t k1) {
    const uint32_t d_lo = base_lo + (uint32_t)lane;              // draw index of this lane, 64 bit
    const uint32_t d_hi = base_hi + (d_lo < base_lo ? 1u : 0u);
    uint32_t c0 = (d_lo >> 2) | (d_hi << 30), c1 = d_hi >> 2, c2 = g_lo, c3 = g_hi;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // full 64-bit products: one v_mad_u64_u32 each instead of a v_mul_hi + v_mul_lo pair
        const uint64_t p0 = (uint64_t)0xCD9E8D57u * c2, p1 = (uint64_t)0xD2511F53u * c0;
        const uint32_t n0 = (uint32_t)(p0 >> 32) ^ c1 ^ k0;
        c1 = (uint32_t)p0;
        const uint32_t n2 = (uint32_t)(p1 >> 32) ^ c3 ^ k1;
        c3 = (uint32_t)p1;
        c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t sel = d_lo & 3u;
    return sel == 0 ? c0 : sel == 1 ? c1 : sel == 2 ? c2 : c3;
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

// record word idx <- a wave-uniform value by v_writelane_b32, no lane mask (a `lane == idx` select
// costs a compare plus an SGPR pair per index, and the compiler keeps all those masks live for the
// whole kernel: 60+ SGPRs, the source of the SGPR spills of the persistent kernel).  clang has no
// builtin for it.  val must be wave-uniform (it goes through v_readfirstlane unless it is known to
// be).  gfx9 VALU instructions read ONE SGPR, so the lane select is an inline constant (HV_SET_C,
// literal index) or M0 (HV_SET, computed index; the s_nop covers an index that a VALU instruction
// produced: v_readlane -> SGPR -> lane select needs 4 wait states).
template <int IDX>
__device__ __forceinline__ uint32_t hv_writelane_c(uint32_t old, uint32_t val) {
    static_assert(IDX >= 0 && IDX < 64, "lane index");
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(uni(val)), "n"(IDX));
    return old;
}
__device__ __forceinline__ uint32_t hv_writelane(uint32_t old, uint32_t val, uint32_t idx) {
    asm("s_nop 3\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(uni(val)), "s"(uni(idx)) : "m0");
    return old;
}
#define HV_SET_C(idx, val) hv = hv_writelane_c<(idx)>(hv, (uint32_t)(val))
#define HV_SET(idx, val) hv = hv_writelane(hv, (uint32_t)(val), (uint32_t)(idx))
// index that is a constant once the enclosing loop is unrolled (snake columns, inline fruits): the switch
// folds to the immediate form -- one instruction instead of M0 write + hazard nop + v_writelane.
// Never for an index that is only known at run time (that would be a 32-way jump table).
__device__ __forceinline__ uint32_t hv_writelane_unrolled(uint32_t old, uint32_t val, uint32_t idx) {
    switch (idx) {
        case 0: return hv_writelane_c<0>(old, val); case 1: return hv_writelane_c<1>(old, val); case 2: return hv_writelane_c<2>(old, val); case 3: return hv_writelane_c<3>(old, val); case 4: return hv_writelane_c<4>(old, val); case 5: return hv_writelane_c<5>(old, val); case 6: return hv_writelane_c<6>(old, val); case 7: return hv_writelane_c<7>(old, val); case 8: return hv_writelane_c<8>(old, val); case 9: return hv_writelane_c<9>(old, val); case 10: return hv_writelane_c<10>(old, val); case 11: return hv_writelane_c<11>(old, val); case 12: return hv_writelane_c<12>(old, val); case 13: return hv_writelane_c<13>(old, val); case 14: return hv_writelane_c<14>(old, val); case 15: return hv_writelane_c<15>(old, val); case 16: return hv_writelane_c<16>(old, val); case 17: return hv_writelane_c<17>(old, val); case 18: return hv_writelane_c<18>(old, val); case 19: return hv_writelane_c<19>(old, val); case 20: return hv_writelane_c<20>(old, val); case 21: return hv_writelane_c<21>(old, val); case 22: return hv_writelane_c<22>(old, val); case 23: return hv_writelane_c<23>(old, val); case 24: return hv_writelane_c<24>(old, val); case 25: return hv_writelane_c<25>(old, val); case 26: return hv_writelane_c<26>(old, val); case 27: return hv_writelane_c<27>(old, val); case 28: return hv_writelane_c<28>(old, val); case 29: return hv_writelane_c<29>(old, val); case 30: return hv_writelane_c<30>(old, val); case 31: return hv_writelane_c<31>(old, val);
        default: return hv_writelane(old, val, idx);
    }
}
#define HV_SET_U(idx, val) hv = hv_writelane_unrolled(hv, (uint32_t)(val), (uint32_t)(idx))
#define HV_SET_DYN(idx, val) HV_SET(idx, val)

typedef uint4 __attribute__((aligned(1))) uint4_unaligned;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
typedef u32x4 __attribute__((aligned(4))) u32x4_dw;  // 16 bytes at a dword-aligned address

// One snake pixel in the C = 3*VIEWS interleaved channels: view v shows snake J as "self" when v == J.
// body: self (0,204,0) / other (0,51,204); head: self (191,242,191) / other (128,154,230)
// ([S]:24-33,46-50, [N] Color :256-274).  The C bytes are compile-time constants per (J, head?): they
// are stored as whole words selected by ONE per-lane condition instead of C byte-wise selects.
template <int J, int VIEWS>
__host__ __device__ __forceinline__ constexpr uint8_t pix_byte(bool head, int b) {
    const bool self = (b / 3) == J;
    const int ch = b % 3;
    return head ? (self ? (ch == 1 ? 242 : 191) : (ch == 0 ? 128 : ch == 1 ? 154 : 230))
                : (self ? (ch == 1 ? 204 : 0) : (ch == 0 ? 0 : ch == 1 ? 51 : 204));
}
template <int J, int VIEWS, int NB>
__host__ __device__ __forceinline__ constexpr uint64_t pix_word(bool head, int first) {  // bytes first .. first+NB-1, little endian
    uint64_t w = 0;
    for (int b = 0; b < NB; ++b) w |= (uint64_t)pix_byte<J, VIEWS>(head, first + b) << (8 * b);
    return w;
}
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
template <int J, int VIEWS>
__device__ __forceinline__ void paint_pixel(uint8_t* p, bool head) {
    constexpr int C = 3 * VIEWS;
    if constexpr (C == 3) {
        constexpr uint16_t H = (uint16_t)pix_word<J, VIEWS, 2>(true, 0), B = (uint16_t)pix_word<J, VIEWS, 2>(false, 0);
        constexpr uint8_t H2 = pix_byte<J, VIEWS>(true, 2), B2 = pix_byte<J, VIEWS>(false, 2);
        *reinterpret_cast<u16_unaligned*>(p) = head ? H : B;
        p[2] = head ? H2 : B2;
    } else if constexpr (C == 6) {
        constexpr uint32_t H = (uint32_t)pix_word<J, VIEWS, 4>(true, 0), B = (uint32_t)pix_word<J, VIEWS, 4>(false, 0);
        constexpr uint16_t H4 = (uint16_t)pix_word<J, VIEWS, 2>(true, 4), B4 = (uint16_t)pix_word<J, VIEWS, 2>(false, 4);
        *reinterpret_cast<u32_unaligned*>(p) = head ? H : B;
        *reinterpret_cast<u16_unaligned*>(p + 4) = head ? H4 : B4;
    } else if constexpr (C == 9) {
        constexpr uint64_t H = pix_word<J, VIEWS, 8>(true, 0), B = pix_word<J, VIEWS, 8>(false, 0);
        constexpr uint8_t H8 = pix_byte<J, VIEWS>(true, 8), B8 = pix_byte<J, VIEWS>(false, 8);
        *reinterpret_cast<u64_unaligned*>(p) = head ? H : B;
        p[8] = head ? H8 : B8;
    } else {
        static_assert(C == 12, "views");
        constexpr uint64_t H = pix_word<J, VIEWS, 8>(true, 0), B = pix_word<J, VIEWS, 8>(false, 0);
        constexpr uint32_t H8 = (uint32_t)pix_word<J, VIEWS, 4>(true, 8), B8 = (uint32_t)pix_word<J, VIEWS, 4>(false, 8);
        *reinterpret_cast<u64_unaligned*>(p) = head ? H : B;
        *reinterpret_cast<u32_unaligned*>(p + 8) = head ? H8 : B8;
    }
}

// MODE 0: step, 1: reset every env (msnake_reset), 2: render only (msnake_render), 3: n_steps steps of
//      an action tape in ONE launch with the env kept in registers (msnake_rollout_tape)
// K: integer pixel replication of the observation fused into the copy-out (the reference's WarpFrame,
//    src/utils.py:15-31: cv2.resize to 84x84 with INTER_AREA, which for the exact integer up-scales
//    used there -- 21->84 = x4, 12->84 = x7 -- is plain pixel replication)
// The first 14 kernel-argument dwords (pointers + packed configuration) are preloaded into SGPRs
// by the dispatcher (-mllvm -amdgpu-kernarg-preload-count), so a wave can issue its state loads
// without waiting for a scalar-memory round trip; the rarely used rest comes by value behind them.
template <int RULES, int NS, int MODE, int K>
__global__ __launch_bounds__(MSNAKE_BLOCK_THREADS) void msnake_step_kernel(
    uint8_t* __restrict__ state, uint8_t* __restrict__ obs, const int32_t* __restrict__ actions,
    float* __restrict__ rew_out, uint8_t* __restrict__ done_out, const int32_t nenv, const uint32_t pk0,
    const uint32_t pk1, const uint32_t pk2, const StepRest p) {
    constexpr int VIEWS = RULES == MSNAKE_RULES_NEW_WORLD ? NS : 3;
    constexpr int C = 3 * VIEWS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int lane = (int)(threadIdx.x & 63u);  // (not const: MODE 3 hides it from loop-invariant hoisting, see the step loop)
    const int wave = (int)uni(threadIdx.x >> 6);
    // XCD-aware env mapping: consecutive workgroup ids go round-robin to the 8 XCDs, so within
    // every aligned group of 64 workgroups XCD x takes 8 CONSECUTIVE env blocks (bits 0-2 and 3-5 of
    // the id swapped).  Each XCD's L2 then writes runs of 8 workgroups' envs of the observation tensor
    // (127-254 KB contiguous) instead of single-workgroup pieces whose edge cache lines it shares with two other XCDs.
    // (Measured: 0.4 % less WRITE_SIZE, launch time unchanged -- the envs share nothing else across workgroups.)
    const uint32_t epb = (pk2 >> PK2_EPB_SHIFT) & 0xFFu;  // = blockDim.x / 64, without the scalar-memory round trip
    // (the launch glue rounds the grid up to whole groups of 64 workgroups, so the swap is a permutation of
    //  every group; workgroups it maps beyond the batch leave)
    uint32_t b = blockIdx.x;
    b = (b & ~63u) | ((b & 7u) << 3) | ((b >> 3) & 7u);
    const int e = (int)(b * epb) + wave;
    if (e >= nenv) return;

    // pk0 = dim | n_fruits<<6 | action_stride<<12 | auto_reset<<15 | max_steps<<16 ; pk1 = S | cap<<16
    // pk2 = PK2_* flags
    // Everything below is one or two scalar instructions away from the three preloaded words.  The
    // persistent kernel (MODE 3) re-derives it at the top of every step from copies the compiler
    // cannot see through, so that none of it has to stay in SGPRs (or be spilled) across the loop.
    uint32_t pk0v = pk0, pk1v = pk1, pk2v = pk2;
    int dim, nf, action_stride, S, cap, W, n2, img_bytes, occ_bytes;
    bool auto_reset, short_rec;
    uint32_t max_steps;
    uint8_t *img, *bg;
    // inline fruits: record words FR0 .. FR0 + nf - 1 (lanes of hv)
    constexpr int FR0 = RULES == MSNAKE_RULES_NEW_WORLD ? HDR_FRUIT0_N : HDR_FRUIT0_S;
    // the persistent tape kernel keeps a pristine copy of the background in LDS (native size only)
    constexpr bool LDSBG = MODE == 3 && K == 1;
    // Aligned copy-out (native size, per-step launches): the image is composed `shift` = (address of the env's observation)
    // & 15 bytes into its LDS buffer, over background number `shift` of the 16 pre-shifted ones, so that both sides of every
    // 16-byte chunk of the copy-out are 16-byte aligned and every wave instruction stores whole 128-byte lines (step 7a).
    // The persistent tape kernel keeps the image at offset 0 and byte-aligned stores (its background lives in LDS).
    constexpr int TMPL_COPIES = K == 1 ? MSNAKE_TMPL_COPIES : 1;
#ifdef MSNAKE_NO_ALIGN  // (A/B builds: every launch takes the byte-aligned copy-out of rounds 1-2)
    constexpr bool CAN_ALIGN = false;
#else
    constexpr bool CAN_ALIGN = K == 1;
#endif
    // (the persistent tape kernel: only when every step's observations keep the alignment of the first step's, i.e. the
    //  step stride is a multiple of 16 bytes -- the launch glue checks -- so that its LDS-resident background fits all steps)
    const bool align_now = CAN_ALIGN && (MODE != 3 || (pk2 & PK2_TAPE_ALIGNED));
    auto unpack = [&]() {
        dim = (int)(pk0v & 63u); nf = (int)((pk0v >> 6) & 63u); action_stride = (int)((pk0v >> 12) & 7u);
        auto_reset = (pk0v >> 15) & 1u;
        max_steps = pk0v >> 16;
        S = (int)(pk1v & 0xFFFFu); cap = (int)(pk1v >> 16);
        // snake_env / adversarial can run on the first 128 bytes of the record (no parked Philox draws)
        short_rec = RULES != MSNAKE_RULES_NEW_WORLD && (pk2v & PK2_SHORT_REC);
        W = dim + 2; n2 = dim * dim;
    };
    auto lds_layout = [&]() {
        // LDS image: W rows of W*K pixels (already replicated horizontally when K > 1), padded to
        // whole 1 KiB wave-instructions
        img_bytes = (S * K + (K == 1 ? 15 : 0) + 1023) & ~1023;  // (native size: + room for the image's shift, see OBS_SHIFT)
        occ_bytes = (n2 + 15) & ~15;
        img = smem + (size_t)wave * (size_t)(img_bytes + occ_bytes + (LDSBG ? img_bytes : 0));  // observation being composed
        bg = img + img_bytes + occ_bytes;                             // LDSBG: background, copied to img every step (the respawn occupancy sits in between)
    };
    unpack();
#ifdef MSNAKE_DBG_STAGES
    const uint32_t dbg = p.dbg_stage;  // timing-only early exits
// (s_endpgm rather than `return`: an early return from the middle of this kernel trips the backend --
//  "illegal VGPR to SGPR copy" -- once a wave-uniform bool is live across it)
#define DBG_EXIT(n) if (dbg == (n)) { if (hv == 0xDEADBEEFu) hdr_g[lane] = hv + cr[0] + (uint32_t)actv; asm volatile("s_endpgm"); }
    // diagnostic build only: 100 MHz wall-clock stamps per wave, never read by the kernel itself
#define STAMP(k) if (p.dbg_buf && lane == 0) p.dbg_buf[(size_t)e * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
#define STAMP_FLAG(v) if (p.dbg_buf && lane == 0) p.dbg_buf[(size_t)e * 8 + 7] = (v)
// second row of stamps per env (rows nenv .. 2*nenv-1 of the buffer): inside the respawn path
#define STAMP2(k) if (p.dbg_buf && lane == 0) p.dbg_buf[((size_t)nenv + (size_t)e) * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
#define DBG_FEWER_STORES(i) && !((dbg & 0x200) && (i) > 0)  /* stage bit 9: timing with a quarter of the observation stores */
#define DBG_NO_OBS_STORES(S) ((dbg & 0x400) ? 0 : (S))     /* bit 10: ... with none of them */
#else
#define DBG_FEWER_STORES(i)
#define DBG_NO_OBS_STORES(S) (S)
#define DBG_EXIT(n)
#define STAMP2(k)
#define STAMP(k)
#define STAMP_FLAG(v)
#endif
// span stamps (tools/span_gap.py; -DMSNAKE_DBG_STAGES, or -DMSNAKE_SPAN_LIGHT = the production kernel plus ONLY these):
// one row of 8 per env and LAUNCH (the glue hands every launch its own slot), so that back-to-back launches can be told
// apart: [0] the wave has started, [1] its loads have landed, [2] logic done (reward / done / info stored), [3] image
// painted, [4] last store issued, [5] every store of the wave acknowledged (it waits for them: only when the span buffer
// is set), [6] flags: 1 a snake ate, 2 episode ended, 4 Philox evaluated, 8 ... behind the stores, XCC id << 8.
// MSNAKE_SPAN_LIGHT keeps [0], [5] and [6] only.
#if defined(MSNAKE_DBG_STAGES) || defined(MSNAKE_SPAN_LIGHT)
#define MSNAKE_HAVE_SPAN 1
#ifdef MSNAKE_SPAN_LIGHT  // (the start stamp stays in registers until the end: a store ahead of the state loads would sit in their vmcnt wait)
    unsigned long long span_t0 = 0;
#define SPAN(k) do { if ((k) == 0) span_t0 = __builtin_amdgcn_s_memrealtime(); \
                     if ((k) == 5 && p.dbg_span && lane == 0) { p.dbg_span[(size_t)e * 8] = span_t0; p.dbg_span[(size_t)e * 8 + 5] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define SPAN(k) if (p.dbg_span && lane == 0) p.dbg_span[(size_t)e * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
#endif
#define SPAN_FLAG(v) dbgf |= (v)
    uint32_t dbgf = 0;
#else
#define SPAN(k)
#define SPAN_FLAG(v)
#endif
    // Section boundary for the register allocator: lane masks (`lane == k`, `lane < k`: an SGPR pair
    // each) computed before it are not kept alive past it.  Only in the instantiations whose peak
    // SGPR demand would otherwise spill (adversarial rules, four snakes); a no-op elsewhere.
    constexpr bool FENCED = RULES == MSNAKE_RULES_ADVERSARIAL || NS == 4 || MODE == 3;
#define LANE_FENCE() do { if (FENCED) asm volatile("" : "+v"(lane)); } while (0)
    // The same for everything unpacked from the three configuration words (a dozen SGPRs): re-derived behind the fence
    // from copies the compiler cannot see through, so the values of the section before die there.  (The persistent
    // kernel does this once per step already.)
#define CONFIG_FENCE() do { if (FENCED && MODE != 3) { asm volatile("" : "+s"(pk0v), "+s"(pk1v), "+s"(pk2v)); unpack(); lds_layout(); } } while (0)

    // ---- 0. every load whose address depends only on the env index.  One allocation holds
    //         [records | chunk-0 bodies | background image | rings]: one preloaded pointer ---------
    STAMP(0);
    SPAN(0);
    // (byte offsets in 32 bits: records + body rings of a handle stay below 4 GB, msnake_create checks)
    uint32_t* hdr_g = reinterpret_cast<uint32_t*>(state + (uint32_t)e * (uint32_t)(MSNAKE_HDR_WORDS * 4));
    uint16_t* body0_g = reinterpret_cast<uint16_t*>(state + ((uint32_t)nenv * (uint32_t)(MSNAKE_HDR_WORDS * 4) +
                                                             (uint32_t)e * (uint32_t)(NS * 128)));
    // background image and, behind it, the overflow rings (bodies longer than 64 cells only): the
    // addresses are derived where they are needed -- the asm keeps the compiler from computing them
    // at kernel entry and holding them in SGPRs for the whole launch
    auto tmpl_of = [&]() -> const uint8_t* {
        uint32_t n = (uint32_t)nenv;
        if (MODE == 3) asm volatile("" : "+s"(n));
        return state + n * (uint32_t)(MSNAKE_HDR_WORDS * 4 + NS * 128);
    };
    auto ring_of = [&](int s) -> uint16_t* {
        uint32_t ee = (uint32_t)e;
        asm volatile("" : "+s"(ee));
        return reinterpret_cast<uint16_t*>(const_cast<uint8_t*>(tmpl_of()) + img_bytes * TMPL_COPIES) + ((size_t)ee * NS + (size_t)s) * (size_t)cap;
    };
    // THE env record: lane l holds word l; lanes FR0+f hold fruit f.  Short record: lanes 32..63 re-read
    // words 0..31 (the same cache line: no extra traffic, no exec-mask juggling) and are zeroed
    uint32_t hv = hdr_g[short_rec ? (lane & (MSNAKE_HDR_SHORT_WORDS - 1)) : lane];
    hv = (short_rec && lane >= MSNAKE_HDR_SHORT_WORDS) ? 0u : hv;
    uint32_t cr[NS];            // cr[s], lane l: slot l of snake s's body ring; piece i sits in slot (hp0 + i) & 63
    int pidx[NS];               // pidx[s], lane l: (l - hp0) & 63 = the piece in slot l (set by the collision test and by reset, for the painters)
#pragma unroll
    for (int s = 0; s < NS; ++s) cr[s] = body0_g[s * 64 + lane];
    int actv = 0;
    constexpr bool STEPS = MODE == 0 || MODE == 3;  // MODE 3: n_steps steps of an action tape in one launch
    // (every lane loads -- lanes >= NS re-read the last snake's action, which nothing looks at -- instead of
    //  an exec-mask region around three lanes)
    if (MODE == 0) actv = actions[(uint32_t)e * (uint32_t)action_stride + (uint32_t)(lane < NS ? lane : NS - 1)];
    // Early Philox (per-step launches with parked draws): the draw counter and the state of the parked draws also come
    // by two SCALAR loads, which land before the background has (issued AHEAD of the state loads they cost every wave 0.03 us), so that a wave whose parked draws no longer cover a
    // reset (4*NS) evaluates Philox while it waits for its loads instead of ahead of its logic when the reset comes.
    // (snake_env only: the adversarial per-step kernels have no SGPRs to spare for it -- 5 spilled instead of 2)
    constexpr bool EARLY_PHILOX = MODE == 0 && RULES == MSNAKE_RULES_SNAKE_ENV;
    unsigned long long early_ctr = 0, early_pc = 0;
    if (EARLY_PHILOX && !short_rec)
        asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4"
                     : "=&s"(early_ctr), "=&s"(early_pc) : "s"(hdr_g), "n"(HDR_CTR_LO * 4), "n"(HDR_PC_VALID * 4) : "memory");
    // (the state loads above are in flight before anything else of the entry block is computed)
    __builtin_amdgcn_sched_barrier(0);
    lds_layout();
    // adversarial rules keep a growing fruit LIST ([A]:183-185 appends dead bodies to it): entries
    // 0..63 in a VGPR like a body chunk (lane l = entry l), the complete list in HBM behind the rings
    const int fcap = (NS + NS * (n2 + 2) + 63) & ~63;
    // (addresses derived at the point of use, like the overflow rings: nothing kept in SGPRs)
    auto fl0_of = [&]() -> uint16_t* {
        uint32_t ee = (uint32_t)e;
        asm volatile("" : "+s"(ee));
        return reinterpret_cast<uint16_t*>(const_cast<uint8_t*>(tmpl_of()) + img_bytes * TMPL_COPIES) + (size_t)nenv * NS * cap + (size_t)ee * 64;
    };
    auto flist_of = [&]() -> uint16_t* {
        uint32_t ee = (uint32_t)e;
        asm volatile("" : "+s"(ee));
        return reinterpret_cast<uint16_t*>(const_cast<uint8_t*>(tmpl_of()) + img_bytes * TMPL_COPIES) + (size_t)nenv * NS * cap + (size_t)nenv * 64 +
               (size_t)ee * fcap;
    };
    uint32_t fr = 0;
    bool fr_dirty = false;      // fr changed in this launch (wave-uniform): chunk 0 of the list is written back
    if (RULES == MSNAKE_RULES_ADVERSARIAL) {
        fr = fl0_of()[lane];
    }

    // ---- RNG: randint(n) = (u32 * n) >> 32 on draw number ctr (kept in the record).  Slow paths
    //      only; `draws` caches the u32 of draws [draw_base, draw_base + 64) --------------------
    //      `draws_n` of them are valid: 64 after a Philox evaluation, 27 when they were taken from
    //      the record's cache words (HDR_PC_*), 0 = none.
    constexpr bool PCACHE = RULES != MSNAKE_RULES_NEW_WORLD;  // new_world's fruits may fill the record
    uint32_t draws = 0, draw_base = 0, draws_n = 0;
    // 0: no Philox evaluation is cached in `draws`; 1: one is, and what it leaves is parked at the end of the step; 2: one is,
    // short record (nowhere to park).  (Wave-uniform flags are kept as NUMBERS: as bools they become 64-bit lane masks, and
    // every test of a combination of them is a handful of scalar instructions on this kernel's busiest port.)
    uint32_t refilled = 0;
    // (the tape kernels and the adversarial ones keep the short-record test at the END of the step instead: folding it into the
    //  flags keeps the test alive across their slow paths, which costs them SGPRs they do not have)
    constexpr bool NUMFLAGS = MODE != 3 && RULES != MSNAKE_RULES_ADVERSARIAL;
#ifdef MSNAKE_LATE_REFILL
    uint32_t slow_step = 0;  // != 0: this wave respawned a fruit or ended an episode in this launch (MODE 0 / 1)
#endif
    // words 32..63 of a snake_env record only hold the Philox cache: they go back to memory in the launches
    // that change it (a Philox evaluation, the 2^32-draw wrap), not in every one.  (new_world keeps its fruits
    // there; the adversarial kernels have no SGPR to spare for the flag.)
    constexpr bool UPPER_TRACKED = RULES == MSNAKE_RULES_SNAKE_ENV;
    // words of the record beyond the first 32 that go back to memory at the end of the step (0 or 32; a number, see `refilled`)
    uint32_t rec_extra = (UPPER_TRACKED || (NUMFLAGS && short_rec)) ? 0u : (uint32_t)(MSNAKE_HDR_WORDS - MSNAKE_HDR_SHORT_WORDS);
    auto upper_is_dirty = [&]() { if (UPPER_TRACKED) rec_extra = (NUMFLAGS && short_rec) ? 0u : (uint32_t)(MSNAKE_HDR_WORDS - MSNAKE_HDR_SHORT_WORDS); };
    auto refill_draws = [&](uint32_t ctr_lo, uint32_t ctr_hi) {
        uint32_t ee = (uint32_t)e;
        asm volatile("" : "+s"(ee));  // (slow path only, like the key schedule below)
        const uint64_t gid = p.env_id_base + (uint64_t)ee;
        uint32_t k0 = p.seed_lo, k1 = p.seed_hi;
        // slow path only: keep the ten-round key schedule from being hoisted to kernel entry,
        // where it would pin 20 SGPRs for every wave
        asm volatile("" : "+s"(k0), "+s"(k1));
#ifdef MSNAKE_DBG_STAGES
        if (dbg & 0x4000) draws = ((ctr_lo + (uint32_t)lane) * 2654435761u) ^ ((uint32_t)gid * 40503u);  // stage bit 14: timing with a free Philox (wrong draws)
        else
#endif
        draws = philox_draws(ctr_lo, ctr_hi, lane, (uint32_t)gid, (uint32_t)(gid >> 32), k0, k1);
        draw_base = ctr_lo; draws_n = 64; refilled = !NUMFLAGS ? 1u : (short_rec ? 2u : 1u);
        SPAN_FLAG(4u);
    };
    // the next `need` (<= 64) draws are cached afterwards.  Callers run this BEFORE they build
    // their wide temporaries (free-cell masks), so Philox does not set the kernel's VGPR peak.
    auto ensure_draws = [&](uint32_t need) {
        const uint32_t ctr_lo = rdlane(hv, HDR_CTR_LO), ctr_hi = rdlane(hv, HDR_CTR_HI);
        if (need <= draws_n && ctr_lo - draw_base <= draws_n - need) return;
        if (PCACHE && refilled == 0 && need <= HDR_PC_N && rdlane(hv, HDR_PC_VALID) == 1u) {
            const uint32_t pb = rdlane(hv, HDR_PC_BASE);
            if (ctr_lo - pb <= HDR_PC_N - need) {
                // the record register itself serves as the cache: draw PC_BASE + i sits in lane HDR_PC_FIRST + i, i.e.
                // "64 draws from PC_BASE - HDR_PC_FIRST on" of which only the parked ones are ever asked for (no shuffle)
                draws = hv;
                draw_base = pb - (uint32_t)HDR_PC_FIRST; draws_n = (uint32_t)(HDR_PC_FIRST + HDR_PC_N);
                return;
            }
        }
        refill_draws(ctr_lo, ctr_hi);
    };
    auto ctr_wrapped = [&](uint32_t ctr_hi) {  // once per 2^32 draws: drop both caches
        HV_SET_C(HDR_CTR_HI, ctr_hi + 1);
        draws_n = 0;
        if (PCACHE) { HV_SET_C(HDR_PC_VALID, 0u); upper_is_dirty(); }
    };
    auto randint = [&](uint32_t n) -> uint32_t {  // ensure_draws() has covered this draw
        const uint32_t ctr_lo = rdlane(hv, HDR_CTR_LO), ctr_hi = rdlane(hv, HDR_CTR_HI);
        const uint32_t u = rdlane(draws, (int)((ctr_lo - draw_base) & 63u));
        const uint32_t nlo = ctr_lo + 1;
        HV_SET_C(HDR_CTR_LO, nlo);
        if (nlo == 0) ctr_wrapped(ctr_hi);
        return (uint32_t)(((uint64_t)u * n) >> 32);
    };

    // ---- per-piece visitor: f(i, cell) for every piece i of snake s: the 64 most recent pieces from
    //      the body ring in registers (slot (hp0 + i) & 63), pieces >= 64 from the overflow ring ----
    //      wA / wC = the snake's SN_A / SN_C record words
    //      (for_each_piece_at: the caller already holds i0 = the piece index of this lane's ring slot)
    auto for_each_piece_at = [&](int s, uint32_t reg, int i0, uint32_t wA, int s_len, auto&& f) {
        if (i0 < s_len) f(i0, reg);
        for (int base = 64; base < s_len; base += 64) {  // long bodies only
            const int i = base + lane;
            if (i < s_len) {
                int idx = (int)(wA & 0xFFFFu) + i - 64;
                idx = idx >= cap ? idx - cap : idx;
                f(i, (uint32_t)__hip_atomic_load(&ring_of(s)[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
    };
    auto for_each_piece = [&](int s, uint32_t reg, uint32_t wA, uint32_t wC, int s_len, auto&& f) {
        for_each_piece_at(s, reg, (lane - (int)(wC >> SN_C_HP0_SHIFT)) & 63, wA, s_len, f);
    };
    // the 32-byte sector of snake s's body ring that holds slot `slot` goes back to memory (whole
    // sectors: nothing for the memory side to merge)
    auto store_ring_sector = [&](int s, uint32_t reg, int slot) {
        if ((lane >> 4) == (slot >> 4)) body0_g[s * 64 + lane] = (uint16_t)reg;
    };

    // [A] fruit-list visitor: f(i, cell) for every list entry i < nlist
    auto for_each_fruit = [&](int nlist, auto&& f) {
        if (lane < nlist) f(lane, fr);
        for (int base = 64; base < nlist; base += 64) {
            const int i = base + lane;
            if (i < nlist) f(i, (uint32_t)__hip_atomic_load(&flist_of()[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    };
    // fruits (of any rule set) lying on `cell`: how many, wave-uniform
    auto fruits_on = [&](uint32_t cell) -> int {
        if (RULES == MSNAKE_RULES_ADVERSARIAL) {
            const int nlist = (int)rdlane(hv, HDR_NLIST);
            int n = __builtin_popcountll(ballot(lane < nlist && fr == cell));
            for (int base = 64; base < nlist; base += 64) {
                const int i = base + lane;
                n += __builtin_popcountll(ballot(i < nlist && (uint32_t)__hip_atomic_load(&flist_of()[i < nlist ? i : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cell));
            }
            return n;
        }
        return __builtin_popcountll(ballot(lane >= FR0 && lane < FR0 + nf && (hv & 0xFFFFu) == cell));
    };

    // ---- fruit respawn: [S]:202-217 safe_choose_cell == [N]:235-247 get_safe_cell --------------
    // occupancy is ONE BIT per cell index x = c1*dim + c0 in LDS (set with ds_or); lane c then owns
    // the 64 cells of chunk c as a register bit mask, a DPP prefix sum over the per-lane free
    // counts gives len(available) and locates the k-th free cell without any loop over cells.
    uint64_t freemask = 0;   // lane c: free cells of chunk c
    uint32_t freescan = 0;   // lane c: number of free cells in chunks 0..c
    int nfree = 0;
    // [S] respawns run AHEAD of the vector move (section 1a): snakes in `bf_moved` count as already
    // moved, i.e. without their popped tail (lane j of bf_pop) and with their new head (bf_nh).
    uint32_t bf_moved = 0, bf_pop = 0, bf_nh = 0;
    // the same for the straight-line form of the map (every body within its 64-slot ring): lane j = the number of
    // pieces of snake j that count at this moment (its length, less the popped tail if it has moved)
    uint32_t bf_len = 0;
    bool bf_short = false;
    auto build_free = [&]() {
#ifdef MSNAKE_DBG_STAGES
        if (dbg & 0x2000) { nfree = n2; return; }  // stage bit 13: timing with a free free-cell map and pick (wrong cells)
#endif
        uint8_t* occ = img + img_bytes;  // respawn occupancy (slow path: derived here, not held across the kernel)
        uint32_t* occw = reinterpret_cast<uint32_t*>(occ);
        // one bit per cell, 64 of them per lane of the readback below
        if (n2 <= 2048) { if (lane < ((n2 + 63) >> 6) * 2) occw[lane] = 0u; }
        else for (int i = lane; i < ((n2 + 63) >> 6) * 2; i += 64) occw[i] = 0u;
        wave_sync();
        auto mark = [&](int, uint32_t cell) {
            // used = c1*dim + c0; out-of-grid heads alias onto other cells or fall outside
            const int used = ((int)(cell & 255u) - 1) * dim + ((int)(cell >> 8) - 1);
            if ((uint32_t)used < (uint32_t)n2) atomicOr(&occw[used >> 5], 1u << (used & 31));
        };
        if (RULES == MSNAKE_RULES_SNAKE_ENV && bf_short) {
            // straight-line form (section 1a, no body beyond its ring): the new heads of the snakes already moved go in
            // by ONE masked store (lane j = snake j), each body by one more
            if (in_mask((uint64_t)bf_moved)) mark(0, bf_nh);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const uint32_t pi = (uint32_t)(lane - (int)(rdlane(hv, SN_C(j)) >> SN_C_HP0_SHIFT)) & 63u;
                if (pi < rdlane(bf_len, j)) mark(0, cr[j]);
            }
        } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const uint32_t w0 = rdlane(hv, SN_A(j));
            int len = (int)(w0 >> 16);
            if (RULES == MSNAKE_RULES_SNAKE_ENV && ((bf_moved >> j) & 1u)) {
                len -= (int)rdlane(bf_pop, j);
                if (lane == 0) mark(0, rdlane(bf_nh, j));
            }
            for_each_piece(j, cr[j], w0, rdlane(hv, SN_C(j)), len, mark);
        }
        }
        wave_sync();
        const int base = lane * 64;
        uint64_t occbits = ~0ull;
        if (base < n2) occbits = *reinterpret_cast<const uint64_t*>(occ + lane * 8);
        const int rem = n2 - base;  // cells of this chunk that exist
        const uint64_t valid = rem >= 64 ? ~0ull : rem > 0 ? ((1ull << rem) - 1ull) : 0ull;
        freemask = ~occbits & valid;
        freescan = wave_scan_incl((uint32_t)__builtin_popcountll(freemask));
        nfree = (int)rdlane(freescan, 63);
        wave_sync();
    };
    // the draw that is parked in the record for the current counter, straight out of the record register (the caller
    // has checked: parked draws valid, counter within them, no Philox evaluation cached in `draws`)
    auto randint_parked = [&](uint32_t n) -> uint32_t {
        const uint32_t ctr_lo = rdlane(hv, HDR_CTR_LO), ctr_hi = rdlane(hv, HDR_CTR_HI);
        const uint32_t u = rdlane(hv, (int)((uint32_t)HDR_PC_FIRST + ctr_lo - rdlane(hv, HDR_PC_BASE)));
        const uint32_t nlo = ctr_lo + 1;
        HV_SET_C(HDR_CTR_LO, nlo);
        if (nlo == 0) ctr_wrapped(ctr_hi);
        return (uint32_t)(((uint64_t)u * n) >> 32);
    };
    auto pick_cell = [&](auto&& rnd) -> uint32_t {
        int x = 0;
        if (nfree > 0) {
            const uint32_t k = rnd((uint32_t)nfree);
#ifdef MSNAKE_DBG_STAGES
            if (dbg & 0x2000) x = (int)k; else {
#endif
            const int L = __builtin_ffsll((long long)ballot(freescan > k)) - 1;  // chunk holding the k-th free cell
            const uint64_t m = ((uint64_t)rdlane((uint32_t)(freemask >> 32), L) << 32) | rdlane((uint32_t)freemask, L);
            const uint32_t kk = k - (rdlane(freescan, L) - (uint32_t)__builtin_popcountll(m));
            const uint64_t sel = ballot(((m >> lane) & 1ull) && mbcnt(m) == kk);
            x = L * 64 + (__builtin_ffsll((long long)sel) - 1);
#ifdef MSNAKE_DBG_STAGES
            }
#endif
        }
        // x / dim and x % dim without a division: pk2 carries M = floor(2^k / dim) + 1 with k = 14 + bits(dim),
        // exact for every x < 2^14 (cell indices stay below 63^2); the launch glue computes M
        uint32_t p2 = pk2v;
        asm volatile("" : "+s"(p2));  // (slow path only: nothing of this belongs in the entry block)
        const uint32_t sh = 14u + 32u - (uint32_t)__builtin_clz((uint32_t)dim);
        const uint32_t q = ((uint32_t)x * (p2 >> PK2_DIVM_SHIFT)) >> sh;
        const uint32_t r = (uint32_t)x - q * (uint32_t)dim;
        return ((r + 1u) << 8) | (q + 1u);
    };
    auto safe_cell = [&]() -> uint32_t {
        if (RULES == MSNAKE_RULES_ADVERSARIAL && nfree > 0) ensure_draws(1);  // its respawn count is not known up front
        return pick_cell(randint);
    };

    // ---- reset: [S]:219-232 / [NE]:27-33 -> [N]:55-73 ------------------------------------------
    auto do_reset = [&]() {
        if (RULES == MSNAKE_RULES_NEW_WORLD) {  // bodies must be empty while the first ones are placed
#pragma unroll
            for (int s = 0; s < NS; ++s) HV_SET_U(SN_A(s), 0u);
        }
        if (RULES != MSNAKE_RULES_NEW_WORLD) {
            // [S]:223-225 draws snake s's cell then fruit s's cell, each (randint(dim), randint(dim)):
            // 4*NS consecutive draws.  All of them at once: lane l takes draw ctr + l.
            const uint32_t ctr_lo = rdlane(hv, HDR_CTR_LO), ctr_hi = rdlane(hv, HDR_CTR_HI);
            const uint32_t pc_at = ctr_lo - rdlane(hv, HDR_PC_BASE);
            uint32_t u;
            if (PCACHE && refilled == 0 && rdlane(hv, HDR_PC_VALID) == 1u && pc_at <= (uint32_t)(HDR_PC_N - 4 * NS)) {
                u = (uint32_t)__shfl((int)hv, (int)((uint32_t)HDR_PC_FIRST + pc_at) + lane);  // the usual case: parked draws, in place
            } else {
                ensure_draws(4u * NS);
                u = (uint32_t)__shfl((int)draws, (int)(ctr_lo - draw_base) + lane);
            }
            const uint32_t v = (uint32_t)(((uint64_t)u * (uint32_t)dim) >> 32) + 1u;  // padded coordinate
            const uint32_t cellv = (v << 8) | row_shl<1>(0u, v);                     // even lanes: (c0+1, c1+1)
            hv = lane < NS ? (1u << 16) : hv;                    // SN_A: overflow empty, len 1
            hv = (lane >= 4 && lane < 4 + NS) ? 3u : hv;         // SN_B: grow_to 3
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t hd = rdlane(cellv, 4 * s), fc = rdlane(cellv, 4 * s + 2);
                HV_SET_U(SN_C(s), hd);  // head cell, velocity (0,0), head in ring slot 0
                cr[s] = hd; pidx[s] = lane;
                store_ring_sector(s, hd, 0);
                if (RULES == MSNAKE_RULES_ADVERSARIAL) {  // fruits = [] then append ([A]:224-229)
                    if (lane == s) { fr = fc; flist_of()[s] = (uint16_t)fc; }
                    fr_dirty = true;
                } else {
                    HV_SET_U(FR0 + s, fc);
                }
            }
            const uint32_t nlo = ctr_lo + 4u * NS;
            HV_SET_C(HDR_CTR_LO, nlo);
            if (nlo < ctr_lo) ctr_wrapped(ctr_hi);
        } else {
            ensure_draws(2u * NS + (uint32_t)nf);  // <= 40
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t c0 = randint((uint32_t)dim), c1 = randint((uint32_t)dim);
                const uint32_t hd = ((c0 + 1) << 8) | (c1 + 1);
                HV_SET_U(SN_A(s), 1u << 16);   // overflow empty, len 1
                HV_SET_U(SN_B(s), 3u);         // grow_to 3
                HV_SET_U(SN_C(s), hd);         // head cell, velocity (0,0), head in ring slot 0
                cr[s] = hd; pidx[s] = lane;
                store_ring_sector(s, hd, 0);
            }
        }
        if (RULES == MSNAKE_RULES_NEW_WORLD) {  // all snakes first, then n_fruits safe cells
            build_free();
            for (int f = 0; f < nf; ++f) {
                const uint32_t c = safe_cell();
                HV_SET_DYN(FR0 + f, c);
            }
        }
        if (RULES == MSNAKE_RULES_ADVERSARIAL) HV_SET_C(HDR_NLIST, (uint32_t)NS);  // spare_fruits survives ([A]:14)
        HV_SET_C(HDR_T, 0u);
        HV_SET_C(HDR_FLAGS, (1u << NS) - 1u);  // alive bits set, dead_snakes empty, episode running
    };

    if (MODE == 1) {
        do_reset();
        HV_SET_C(HDR_EP_RETURN, 0u);
        HV_SET_C(HDR_EP_LEN, 0u);
    }

    DBG_EXIT(1)
    const int n_steps = MODE == 3 ? p.n_steps : 1;
    float* rew_t = rew_out;
    uint8_t* done_t = done_out;
    msnake_info* info_t = p.info;
    uint8_t* obs_t = obs;
    uint32_t abatch = 0;  // MODE 3: the actions of 16 tape rows
#pragma nounroll
    for (int step_i = 0; step_i < n_steps; ++step_i) {
    // every `lane == k` / `lane < k` mask is an SGPR pair; seen as loop invariants they would all be
    // computed ahead of the loop and stay live across it (or be spilled).  Redefining `lane` per
    // iteration keeps them local to the step.
    if (MODE == 3) {
        asm volatile("" : "+v"(lane), "+s"(pk0v), "+s"(pk1v), "+s"(pk2v));
        unpack();
        lds_layout();
        __builtin_amdgcn_s_setprio(0);  // (a slow path of the previous step raised it)
    }
    // MODE 3 issues NO vector-memory load in a normal step: gfx9 has one counter for loads and
    // stores, so waiting for a load would also wait for every observation store still in flight
    // (those of the previous step).  Actions come 16 tape rows at a time (lane 4*j + s <- snake s at
    // step step_i + j; one wait, and thus one store drain, per 16 steps), the background comes from
    // its LDS copy.
    if (MODE == 3) {
        if (step_i > 0) {  // outputs advance by their strides
            rew_t += p.scalar_step_stride; done_t += p.scalar_step_stride;
            if (info_t) info_t += p.scalar_step_stride;
            if (obs_t) obs_t += p.obs_step_stride;
        }
        if ((step_i & 15) == 0) {
            const int t = step_i + (lane >> 2);
            abatch = 0;
            if ((lane & 3) < NS && t < n_steps)
                abatch = (uint32_t)actions[((size_t)t * (size_t)nenv + (size_t)e) * action_stride + (lane & 3)];
        }
    }
    // aligned copy-out: where in its LDS buffer the image starts (see CAN_ALIGN)
    // (derived where it is used -- three scalar instructions -- instead of being held in an SGPR for the whole kernel)
    auto obs_shift_of = [&]() -> uint32_t {
        uint32_t ee = (uint32_t)e;
        if (FENCED) asm volatile("" : "+s"(ee));
        return (align_now && obs_t) ? (((uint32_t)(uintptr_t)obs_t + ee * (uint32_t)S) & 15u) : 0u;
    };
    // background image (black interior, white wall ring): one L1/L2-resident copy shared by every
    // wave, a whole number of 1 KiB wave-instructions so that no lane needs a predicate.
    // (Storing the background to HBM right here, ahead of the logic, was measured and is SLOWER:
    //  the 16 MB of early stores clog each CU's memory pipe in front of every later access.)
    // It goes memory -> LDS directly (global_load_lds_dwordx4: lane l's 16 bytes land at
    // M0 base + 16*l), so the copy holds no data VGPRs and needs no ds_write.
    // (Issuing it only after the state has arrived, so that no wave's state load queues behind
    //  another wave's 4 KB of background, was measured too: 6.88 instead of 6.79 us, same box.)
#ifdef MSNAKE_DBG_STAGES
    if (obs_t && (!LDSBG || step_i == 0) && !(dbg & 0x100)) {  // (stage bit 8: timing without the background DMA -- wrong images)
#else
    if (obs_t && (!LDSBG || step_i == 0)) {
#endif
        // (aligned copy-out: background number `shift`, the image starts `shift` bytes into it; the offset joins the scalar base)
        const uint4* tsrc = reinterpret_cast<const uint4*>(tmpl_of() + (CAN_ALIGN ? obs_shift_of() * (uint32_t)img_bytes : 0u)) + lane;
        uint8_t* dst = LDSBG ? bg : img;
        const int nk = img_bytes >> 10;
#define MSNAKE_GP(k) ((const __attribute__((address_space(1))) void*)(tsrc + (k) * 64))
#define MSNAKE_LP(k) ((__attribute__((address_space(3))) void*)(dst + (k) * 1024))
        // The first four KiB (the native images: 4 for 21x21x9, 2 for 12x12x9) go by ONE asm block: one M0
        // write, one address pair (the instruction offset moves both sides), plain tests in between.  (The
        // builtin re-writes M0 per load, and an if / else-if over nk becomes a flag-driven state machine.)
        const uint32_t lds0 = (uint32_t)(uintptr_t)MSNAKE_LP(0);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\t"
                     "s_cmp_lt_u32 %2, 2\n\ts_cbranch_scc1 1f\n\t"
                     "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "s_cmp_lt_u32 %2, 3\n\ts_cbranch_scc1 1f\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                     "s_cmp_lt_u32 %2, 4\n\ts_cbranch_scc1 1f\n\t"
                     "global_load_lds_dwordx4 %0, off offset:3072\n"
                     "1:"
                     :: "v"(tsrc), "s"(uni(lds0)), "s"(uni((uint32_t)nk)) : "m0", "scc", "memory");
        if (nk > 4)
            for (int k = 4; k < nk; ++k) __builtin_amdgcn_global_load_lds(MSNAKE_GP(k), MSNAKE_LP(k), 16, 0, 0);
#undef MSNAKE_GP
#undef MSNAKE_LP
    }
    // early Philox: lane l >= HDR_PC_FIRST computes draw ctr + (l - HDR_PC_FIRST), i.e. the word it will park
    uint32_t early = 0;
    // (the test every wave makes is kept to integer arithmetic and ONE compare: written with bools it compiles to a dozen
    //  scalar select / and instructions and four branches, and the scalar unit is this kernel's busiest port -- an extra
    //  scalar instruction per wave costs a launch 7.5 ns, MSNAKE_PAD_SALU below)
    uint32_t early_ok = 0;
    if (EARLY_PHILOX && !short_rec) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(early_ctr), "+s"(early_pc));
        const uint32_t c_lo = (uint32_t)early_ctr, c_hi = (uint32_t)(early_ctr >> 32);
        const uint32_t pv = (uint32_t)early_pc, pb = (uint32_t)(early_pc >> 32);
        // fewer than 4*NS parked draws left, or none parked (HDR_PC_VALID is 0 or 1) ...
        uint32_t used = (c_lo - pb) | ((pv ^ 1u) << 5);
        asm volatile("" : "+s"(used));
        // (parked draws may straddle the 2^32 wrap of the counter's low word: lane arithmetic is 64 bit, and crossing the
        //  wrap drops the parked draws -- ctr_wrapped -- before any of them could be taken for the wrong counter)
        if (used > (uint32_t)(HDR_PC_N - 4 * NS)) {
            early_ok = 1u;
            uint32_t ee = (uint32_t)e;
            asm volatile("" : "+s"(ee));
            const uint64_t gid = p.env_id_base + (uint64_t)ee;
            uint32_t k0 = p.seed_lo, k1 = p.seed_hi;
            asm volatile("" : "+s"(k0), "+s"(k1));
            const uint32_t b_lo = c_lo - (uint32_t)HDR_PC_FIRST, b_hi = c_hi - (c_lo < (uint32_t)HDR_PC_FIRST ? 1u : 0u);
            early = philox_draws(b_lo, b_hi, lane, (uint32_t)gid, (uint32_t)(gid >> 32), k0, k1);
            SPAN_FLAG(8u);
        }
    }
    // every load issued so far (state, actions, background) has landed past this point: the env
    // logic needs the state right away, and the painters must find the background in LDS.  In MODES
    // 0-2 no store has been issued yet, so this waits for loads only.
    if (!LDSBG || (step_i & 15) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(MSNAKE_PAD_SALU) || defined(MSNAKE_PAD_VALU) || defined(MSNAKE_PAD_BRANCH)
    // experiment (tools: make variant NAME=pads EXTRA=-DMSNAKE_PAD_SALU=32): what one more scalar / vector / branch instruction per
    // wave costs a launch -- 7.5 ns per scalar instruction (16 env-waves per CU share ONE scalar port), 1.6 ns per vector one
    {
#ifdef MSNAKE_PAD_SALU
        uint32_t pad = pk0v;
#pragma unroll
        for (int i = 0; i < MSNAKE_PAD_SALU; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(pad) :: "scc");
        if (pad == 0xDEADBEEFu) hv = pad;
#endif
#ifdef MSNAKE_PAD_VALU
        uint32_t padv = (uint32_t)lane;
#pragma unroll
        for (int i = 0; i < MSNAKE_PAD_VALU; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(padv));
        if (padv == 0xDEADBEEFu) hv = padv;
#endif
#ifdef MSNAKE_PAD_BRANCH
#pragma unroll
        for (int i = 0; i < MSNAKE_PAD_BRANCH; ++i) asm volatile("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" ::: "scc");
#endif
    }
#endif
    asm volatile("" : "+s"(early_ok));
    if (EARLY_PHILOX && early_ok != 0) {  // the fresh draws are parked at once: whatever needs draws in this step finds them there
        hv = lane >= HDR_PC_FIRST ? early : hv;
        HV_SET_C(HDR_PC_BASE, (uint32_t)early_ctr);
        HV_SET_C(HDR_PC_VALID, 1u);
        upper_is_dirty();
    }
    // (the ring slots arrive zero-extended; hiding that they were 16-bit loads spares a v_and per snake)
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(cr[s]));
    if (MODE == 3) {
        const int a = __shfl((int)abatch, 4 * (step_i & 15) + lane);
        actv = lane < NS ? a : 0;
    }
    if (LDSBG && obs_t) {  // img <- pristine background (LDS operations of one wave execute in order)
        wave_sync();
        const int nk = img_bytes >> 10;
        for (int k = 0; k < nk; ++k)
            reinterpret_cast<uint4*>(img)[k * 64 + lane] = reinterpret_cast<const uint4*>(bg)[k * 64 + lane];
    }

    DBG_EXIT(7)
    if (STEPS) {
        LANE_FENCE();
        float reward = 0.0f;
        // ---- 1. snake updates.  Fast path: all snakes at once on the VALU, lane s = snake s.
        //         Valid whenever no moving snake eats (then no fruit respawns, so the updates do
        //         not depend on each other); otherwise the sequential loop below runs instead.
        constexpr int NF_STATIC = RULES == MSNAKE_RULES_NEW_WORLD ? -1 : NS;  // [S]: one fruit per snake
        constexpr uint64_t NSMASK = (1ull << NS) - 1ull;                      // = ballot(lane < NS)
        const uint32_t sA = hv, sB = row_shl<4>(0u, hv), sC = row_shl<8>(0u, hv);
        const int v_len = (int)(sA >> 16), v_grow = (int)sB;
        const int v_head = (int)(sC & 0xFFFFu), v_vel = (int)((sC >> 16) & 7u);
        // turn unless it is a 180-degree reversal: [S]:108-115 == [N]:34-41 == [A]:106-113.  Opposite
        // directions are 1<->3 and 2<->4, i.e. (a - 1) ^ (vel - 1) == 2; no velocity yet (vel 0) never matches
        const uint32_t am = (uint32_t)actv - 1u;
        const bool turn = (am < 4u) & ((am ^ ((uint32_t)v_vel - 1u)) != 2u);
        const int v_nvel = turn ? actv : v_vel;
        // new head = head + step(vel): velocity 1..4 = (+1,0) (0,+1) (-1,0) (0,-1) on (row, column), 0 = none;
        // both deltas come out of 2-bit signed fields of two constants (bit 2*vel)
        const uint32_t vsh = (uint32_t)v_nvel << 1;
        const int v_nh = v_head + __builtin_amdgcn_sbfe(0x310, vsh, 2u) + (__builtin_amdgcn_sbfe(0xC4, vsh, 2u) << 8);
        // snake_env moves only with a velocity ([S]:119); new_world always inserts a head, even a
        // duplicate of itself ([N]:43-48,153)
        const uint32_t mvmask = (uint32_t)NSMASK & (uint32_t)lanes_where<CMP_NE>((uint32_t)v_len, 0u) &
                                (RULES == MSNAKE_RULES_NEW_WORLD ? ~0u : (uint32_t)lanes_where<CMP_NE>((uint32_t)v_nvel, 0u));
        const bool v_moves = in_mask((uint64_t)mvmask);
        uint32_t v_em = 0;  // bit f: fruit f lies on this snake's new head
        bool any_eat = false;
        STAMP(1);
        SPAN(1);
        if (RULES == MSNAKE_RULES_ADVERSARIAL) {  // list based
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (((mvmask >> s) & 1u) && fruits_on(rdlane((uint32_t)v_nh, s)) != 0) any_eat = true;
        } else if (RULES == MSNAKE_RULES_SNAKE_ENV) {
            // fruit f is record word FR0 + f = lane FR0 + f: ONE compare per snake of every lane's low
            // half against that snake's new head; the per-lane eat masks are only built when some bit is set
            const uint32_t cellv = hv & 0xFFFFu;
            const uint32_t nh_mv = v_moves ? (uint32_t)v_nh : 0xFFFFu;  // (no cell: a snake that stays eats nothing)
            uint32_t on = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) on |= (uint32_t)lanes_where<CMP_EQ>(cellv, rdlane(nh_mv, s));
            any_eat = (on & ((uint32_t)NSMASK << FR0)) != 0;
            if (any_eat) {
#pragma unroll
                for (int f = 0; f < NS; ++f)
                    v_em |= ((uint32_t)v_nh == (rdlane(hv, FR0 + f) & 0xFFFFu)) ? (1u << f) : 0u;
                v_em = v_moves ? v_em : 0u;
            }
        } else {
            for (int f = 0; f < nf; ++f)
                v_em |= ((uint32_t)v_nh == (rdlane(hv, FR0 + f) & 0xFFFFu)) ? (1u << f) : 0u;
            v_em = v_moves ? v_em : 0u;
            any_eat = ballot(v_em != 0) != 0;
        }
#ifdef MSNAKE_DBG_STAGES
        if (dbg & 0x1000) any_eat = false;  // stage bit 12: timing with every wave on the fast path (wrong results)
#endif
        if (RULES == MSNAKE_RULES_SNAKE_ENV && any_eat) {
            // A wave on a slow path (respawn, reset) is the one the launch will wait for: it gets the issue
            // slots of its SIMD first from here on.  (Same box, median of six: 5.74 vs 5.87 us per launch.)
            __builtin_amdgcn_s_setprio(3);
            // ---- 1a. [S] respawns, in snake order, ahead of the vector move.  update_snake
            //      ([S]:119-141) pops, inserts the head and only then re-places the fruits it ate, so
            //      snake s's respawns see snakes <= s moved and snakes > s unmoved; a fruit that lands
            //      on a later snake's new head is eaten by it in this very step ([S]:126-132 runs on
            //      the current fruit positions), which is what re-deriving bit f of the later lanes'
            //      eat masks reproduces.
            bf_nh = (uint32_t)v_nh;
            bf_short = ((uint32_t)lanes_where<CMP_UGT>((uint32_t)v_len, 64u) & (uint32_t)NSMASK) == 0;
            // The usual case in straight-line code: ONE snake eats ONE fruit, every body within its ring, the draw parked
            // in the record.  Anything else -- and whatever a later snake eats off the respawned fruit -- takes the loop.
            int s_begin = 0;
            {
                const uint32_t eaters = (uint32_t)lanes_where<CMP_NE>(v_em, 0u) & (uint32_t)NSMASK;
                const int s1 = __builtin_ctz(eaters | (1u << NS));
                const uint32_t m1 = rdlane(v_em, s1);
                const uint32_t pc_at = rdlane(hv, HDR_CTR_LO) - rdlane(hv, HDR_PC_BASE);
                if (eaters != 0 && (eaters & (eaters - 1u)) == 0 && (m1 & (m1 - 1u)) == 0 && bf_short && PCACHE && refilled == 0 &&
                    rdlane(hv, HDR_PC_VALID) == 1u && pc_at < (uint32_t)HDR_PC_N) {
                    bf_pop = (v_moves && v_len >= v_grow + 2 * __builtin_popcount(v_em)) ? 1u : 0u;
                    bf_moved = (uint32_t)mvmask & ((2u << s1) - 1u);
                    bf_len = (uint32_t)v_len - (in_mask((uint64_t)bf_moved) ? bf_pop : 0u);
                    build_free();
                    const int f = __builtin_ctz(m1 | (1u << NS));
                    const uint32_t c = pick_cell(randint_parked);
                    HV_SET_DYN(FR0 + f, c);
                    const uint32_t bit = (uint32_t)v_nh == c ? (1u << f) : 0u;
                    v_em = (v_moves && lane > s1) ? ((v_em & ~(1u << f)) | bit) : v_em;
                    s_begin = s1 + 1;
                    // (a later snake whose new head is where the fruit went eats it in this step: the loop below finds it)
                    if (((uint32_t)lanes_where<CMP_NE>(v_em, 0u) & (uint32_t)NSMASK & ~((2u << s1) - 1u)) == 0) s_begin = NS;
                }
            }
#pragma nounroll
            for (int s = s_begin; s < NS; ++s) {
                uint32_t m = rdlane(v_em, s);
                if (m == 0) continue;
                bf_pop = (v_moves && v_len >= v_grow + 2 * __builtin_popcount(v_em)) ? 1u : 0u;  // lanes <= s are final
                bf_moved = (uint32_t)mvmask & ((2u << s) - 1u);
                bf_len = (uint32_t)v_len - (in_mask((uint64_t)bf_moved) ? bf_pop : 0u);
                STAMP2(0);
                ensure_draws((uint32_t)__builtin_popcount(m));
                STAMP2(1);
                build_free();
                STAMP2(2);
                while (m) {
                    const int f = __builtin_ffs((int)m) - 1;
                    m &= m - 1;
                    const uint32_t c = safe_cell();
                    HV_SET_DYN(FR0 + f, c);
                    const uint32_t bit = (uint32_t)v_nh == c ? (1u << f) : 0u;
                    v_em = (v_moves && lane > s) ? ((v_em & ~(1u << f)) | bit) : v_em;
                }
                STAMP2(3);
            }
            bf_moved = 0;
        }
        // the vector update: valid whenever no moving snake eats -- and, for snake_env, after section 1a
        const bool vec = RULES == MSNAKE_RULES_SNAKE_ENV || !any_eat;
        if (vec) {
            uint32_t nB = (uint32_t)v_grow;
            uint32_t nA;  // SN_A: overflow head position | length << 16
            if (RULES == MSNAKE_RULES_NEW_WORLD) {  // [N]:143-150 runs the pop test once per fruit
                int pops = v_len - v_grow + 1;
                pops = pops < 0 ? 0 : (pops > nf ? nf : pops);
                nA = sA + (v_moves ? ((uint32_t)(1 - pops) << 16) : 0u);  // insert(0, head) after the pops
            } else {
                if (RULES == MSNAKE_RULES_SNAKE_ENV && any_eat) {
                    const int neat = __builtin_popcount(v_em);
                    nB = (uint32_t)(v_grow + 2 * neat);      // [S]:126-132
                    reward = (float)rdlane((uint32_t)neat, 0);
                }
                // [S]:134-135 / [A]:134-135 (eating steps of [A] take the loop below): the tail is popped iff
                // len >= grow_to, then insert(0, head): the body grows by one exactly when len < grow_to
                nA = sA + (in_mask((uint64_t)(mvmask & (uint32_t)lanes_where<CMP_ULT>((uint32_t)v_len, nB))) ? (1u << 16) : 0u);
            }
            // bodies of 64+ cells (never under random play): capacity guard and eviction, behind ONE test
            uint32_t evmask = 0;
            if ((mvmask & (uint32_t)lanes_where<CMP_UGE>(nA, 64u << 16)) != 0) {
                int nlen = (int)(nA >> 16);
                if (ballot(v_moves && nlen > cap - 1) != 0) {  // unreachable under the documented caps
                    HV_SET_C(HDR_ACC_ERRORS, rdlane(hv, HDR_ACC_ERRORS) + 1u);
                    nlen = nlen > cap - 1 ? cap - 1 : nlen;
                }
                // the new head takes ring slot hp0 - 1, which held piece 63; if that piece stays part of the
                // body (it becomes piece 64) it moves to the front of the overflow ring first
                const bool v_evict = v_moves && v_len >= 64 && nlen >= 65;
                evmask = (uint32_t)ballot(v_evict);
                const int v_hp = (int)(sA & 0xFFFFu);
                const int nohp = v_evict ? (v_hp == 0 ? cap - 1 : v_hp - 1) : v_hp;
                nA = v_moves ? ((uint32_t)nohp | ((uint32_t)nlen << 16)) : sA;
            }
            // SN_C: head | vel << 16 | ring slot of the head << 24, the slot steps down by one (mod 64)
            const uint32_t nC = v_moves ? ((((sC - (1u << SN_C_HP0_SHIFT)) & (63u << SN_C_HP0_SHIFT)) | (uint32_t)v_nh) | ((uint32_t)v_nvel << 16)) : sC;
            // the three columns go back into the record by one masked DPP move each (lanes s, 4+s, 8+s <- lane s;
            // a lane without a moving snake carries its old words)
            hv = dpp_into<DPP_IDENTITY, 0x1, 0x1>(hv, nA);
            if (RULES == MSNAKE_RULES_SNAKE_ENV) hv = dpp_into<0x114, 0x1, 0x2>(hv, nB);
            hv = dpp_into<0x118, 0x1, 0x4>(hv, nC);
            if (evmask != 0) {  // bodies of 64+ cells only
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (((evmask >> s) & 1u) && lane == (int)(rdlane(hv, SN_C(s)) >> SN_C_HP0_SHIFT))
                        ring_of(s)[rdlane(hv, SN_A(s)) & 0xFFFFu] = (uint16_t)cr[s];
            }
            // (the new heads enter the body rings below, off the record words the collision test reads anyway)
        } else {
        // ---- 1b. sequential snake updates (order matters: a respawn sees earlier snakes moved,
        //          a later snake can eat a fruit respawned this very step) ----------------------
        __builtin_amdgcn_s_setprio(3);
#pragma nounroll
        for (int s = 0; s < NS; ++s) {  // a real loop: the slow path below exists once in the code
            LANE_FENCE();
            const uint32_t w0 = rdlane(hv, SN_A(s)), w2 = rdlane(hv, SN_C(s));
            int len = (int)(w0 >> 16);
            if (len == 0) continue;
            const int vel = (int)((w2 >> 16) & 7u), head = (int)(w2 & 0xFFFFu);
            const int act = (int)rdlane((uint32_t)actv, s);
            const int nvel = (act >= 1 && act <= 4 && vel != ((act + 1) & 3) + 1) ? act : vel;
            if (RULES != MSNAKE_RULES_NEW_WORLD && nvel == 0) continue;
            const int nh = head + cell_step(nvel);
            // fruits equal to the new head, as a bit mask over fruit indices (list rules: a count)
            const uint64_t em = RULES == MSNAKE_RULES_ADVERSARIAL ? 0ull :
                ballot(lane >= FR0 && lane < FR0 + nf && (hv & 0xFFFFu) == (uint32_t)nh) >> FR0;
            const int neat = RULES == MSNAKE_RULES_ADVERSARIAL ? fruits_on((uint32_t)nh) : __builtin_popcountll(em);
            const int len0 = len;
            int g = (int)rdlane(hv, SN_B(s));
            if (RULES == MSNAKE_RULES_NEW_WORLD) {
                for (int f = 0; f < nf; ++f) {  // [N]:143-150: the pop test sits inside the fruit loop
                    if ((em >> f) & 1ull) g += 2;
                    if (len >= g) len -= 1;
                }
            } else {
                g += 2 * neat;           // [S]:126-132
                if (len >= g) len -= 1;  // [S]:134-135
            }
            len += 1;  // insert(0, head)
            if (len > cap - 1) { len = cap - 1; HV_SET_C(HDR_ACC_ERRORS, rdlane(hv, HDR_ACC_ERRORS) + 1u); }
            int ohp = (int)(w0 & 0xFFFFu);
            const int slot = ((int)(w2 >> SN_C_HP0_SHIFT) - 1) & 63;  // ring slot of the new head (held piece 63)
            const bool evict = len0 >= 64 && len >= 65;              // piece 63 becomes piece 64: to the overflow ring
            if (evict) ohp = ohp == 0 ? cap - 1 : ohp - 1;
            HV_SET(SN_A(s), (uint32_t)ohp | ((uint32_t)len << 16));
            HV_SET(SN_B(s), (uint32_t)g);
            HV_SET(SN_C(s), (uint32_t)nh | ((uint32_t)nvel << 16) | ((uint32_t)slot << SN_C_HP0_SHIFT));
#pragma unroll
            for (int j = 0; j < NS; ++j)
                if (j == s) {
                    if (evict && lane == slot) ring_of(s)[ohp] = (uint16_t)cr[j];
                    cr[j] = lane == slot ? (uint32_t)nh : cr[j];
                    store_ring_sector(j, cr[j], slot);
                }
            if (s == 0) reward = (float)neat;
            if (neat != 0) {  // respawn each eaten fruit, in index order
                if (RULES == MSNAKE_RULES_ADVERSARIAL) {
                    // [A]:137-141: while spare_fruits > 0 an eaten fruit stays where it is
                    int spare = (int)rdlane(hv, HDR_SPARE);
                    const int nlist = (int)rdlane(hv, HDR_NLIST);
                    bool built = false;
                    for (int base = 0; base < nlist; base += 64) {
                        const int i = base + lane;
                        const uint32_t c = base == 0 ? fr : (uint32_t)__hip_atomic_load(&flist_of()[i < nlist ? i : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        uint64_t m = ballot(i < nlist && c == (uint32_t)nh);
                        while (m) {
                            const int bit = __builtin_ffsll((long long)m) - 1;
                            m &= m - 1;
                            if (spare > 0) { spare -= 1; continue; }
                            if (!built) { build_free(); built = true; }
                            const uint32_t nc = safe_cell();
                            if (base == 0 && lane == bit) fr = nc;
                            fr_dirty = fr_dirty || base == 0;
                            if (lane == 0) flist_of()[base + bit] = (uint16_t)nc;
                        }
                    }
                    HV_SET_C(HDR_SPARE, (uint32_t)spare);
                } else {
                    ensure_draws((uint32_t)neat);
                    build_free();
                    uint64_t m = em;
                    while (m) {
                        const int f = __builtin_ffsll((long long)m) - 1;
                        m &= m - 1;
                        const uint32_t c = safe_cell();
                        HV_SET_DYN(FR0 + f, c);
                    }
                }
            }
        }
        }

        LANE_FENCE();
        CONFIG_FENCE();
        uint32_t hd[NS], ln[NS], hp2[NS], wc[NS];  // head cell, length, overflow head pos, SN_C word
        uint32_t lenor = 0;                        // OR of the SN_A words: some body holds 64+ cells iff >= 64 << 16
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t w0 = rdlane(hv, SN_A(s));
            wc[s] = rdlane(hv, SN_C(s));
            hd[s] = wc[s] & 0xFFFFu;
            ln[s] = w0 >> 16; hp2[s] = w0 & 0xFFFFu;
            lenor |= w0;
        }
        const bool longbody = lenor >= (64u << 16);  // (gates the overflow-ring passes; a body of exactly 64 finds nothing there)
        if (vec) {
            // vector update, last part: each moving snake's new head enters its body ring at the slot SN_C
            // names (one v_writelane), and the 32-byte sector around it goes back to memory
            uint32_t mv32 = mvmask;
            asm volatile("" : "+s"(mv32));  // (plain bit tests: known to be < 2^NS, bit NS-1 becomes a 64-bit VALU compare)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if ((mv32 >> s) & 1u) {  // straight-line: a snake that does not move rewrites nothing
                    const uint32_t slot = wc[s] >> SN_C_HP0_SHIFT;
                    cr[s] = hv_writelane(cr[s], hd[s], slot);
                    if (((uint32_t)lane ^ slot) < 16u) body0_g[s * 64 + lane] = (uint16_t)cr[s];
                }
        }
        STAMP(2);
        DBG_EXIT(2)
        // ---- 2. head-vs-piece matrix: some piece of snake j other than s's own head lies on
        //         s's head.  [S] only needs "any j" per s; [N] needs the full matrix -------------
        uint32_t hitl = 0;      // [N]: per-lane accumulation, bit (4*s + j)
        uint64_t hit_s[NS];     // [S]/[A]: lanes whose cell lies on snake s's head (own head slot excluded)
        uint32_t crvv[NS], crvo[NS];  // [S]/[A]: this lane's cell of snake j, ~0 where the slot holds no piece (crvo: nor in the head's slot)
#pragma unroll
        for (int s = 0; s < NS; ++s) hit_s[s] = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const uint32_t pi = (uint32_t)(lane - (int)(wc[j] >> SN_C_HP0_SHIFT)) & 63u;  // piece index of this slot
            pidx[j] = (int)pi;  // (kept for the painters)
            const bool valid = pi < ln[j];
            if (RULES == MSNAKE_RULES_NEW_WORLD) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    hitl |= (valid && cr[j] == hd[s] && !(j == s && pi == 0)) ? (1u << (4 * s + j)) : 0u;
            } else {
                // slots that hold no piece compare as "no cell", and so does a snake's own head slot against its own head
                crvv[j] = valid ? cr[j] : 0xFFFFFFFFu;
                crvo[j] = pi == 0 ? 0xFFFFFFFFu : crvv[j];
                if (RULES == MSNAKE_RULES_ADVERSARIAL) {
                    // (the adversarial kernels keep the lane-mask form: the vector form below costs them 3 more spilled SGPRs)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        uint64_t m = lanes_where<CMP_EQ>(crvv[j], hd[s]);
                        if (j == s) m &= ~(1ull << (wc[j] >> SN_C_HP0_SHIFT));
                        hit_s[s] |= m;
                    }
                }
            }
        }
        if (RULES == MSNAKE_RULES_SNAKE_ENV) {
            // "some piece lies on snake s's head" per lane as ONE number: the minimum over the snakes of cell ^ head is 0 iff
            // one of them matches -- vector xor / min and one compare per head, instead of a lane mask per (snake, head) pair
            // and-ed and or-ed on the scalar unit (this kernel's busiest port: a scalar instruction costs a launch 7.5 ns here)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint32_t d = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const uint32_t x = (j == s ? crvo[j] : crvv[j]) ^ hd[s];
                    d = x < d ? x : d;
                }
                hit_s[s] = lanes_where<CMP_EQ>(d, 0u);
            }
        }
        if (longbody) {  // bodies longer than one chunk: the rest comes from the ring
#pragma unroll
            for (int j = 0; j < NS; ++j)
                for (int base = 64; base < (int)ln[j]; base += 64) {
                    const int i = base + lane;
                    if (i < (int)ln[j]) {
                        int idx = (int)hp2[j] + i - 64;
                        idx = idx >= cap ? idx - cap : idx;
                        const uint32_t cell = __hip_atomic_load(&ring_of(j)[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            hitl |= (cell == hd[s]) ? (1u << (4 * s + j)) : 0u;  // (all rule sets: folded in below)
                    }
                }
        }
        DBG_EXIT(3)
        // ---- 3. aliveness, reward, done --------------------------------------------------------
        uint32_t done;  // 0 / 1, wave-uniform (an integer on the scalar unit, not a lane mask)
        int num_alive;
        const uint32_t t = rdlane(hv, HDR_T) + 1;
        if (RULES == MSNAKE_RULES_NEW_WORLD) {
            uint32_t hits = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < NS; ++j)
                    if (ballot((hitl >> (4 * s + j)) & 1u) != 0) hits |= 1u << (4 * s + j);
            // [N]:101-107 + :111-132: in snake order, clearing bodies as it goes
            uint32_t flags = rdlane(hv, HDR_FLAGS);
            bool done0 = false;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                bool alive;
                if (ln[s] == 0) alive = false;
                else if (!in_grid(hd[s], dim)) { ln[s] = 0; alive = false; }
                else {
                    bool other = false;
#pragma unroll
                    for (int j = 0; j < NS; ++j)
                        if (j != s && ln[j] > 0 && ((hits >> (4 * s + j)) & 1u)) other = true;
                    if (other) { ln[s] = 0; alive = false; }
                    else alive = !((hits >> (4 * s + s)) & 1u);  // self hit: not alive, body kept
                }
                if (ln[s] == 0) HV_SET_U(SN_A(s), hp2[s]);
                flags = alive ? (flags | (1u << s)) : (flags & ~(1u << s));
                if (!alive) flags |= (16u << s);  // dead_snakes, append-once
                if (s == 0) done0 = alive;        // [N]:107 (sic)
            }
            HV_SET_C(HDR_FLAGS, flags);
            if (done0) reward = -1.0f;  // [NE]:39-40
            done = ((t >= max_steps) || done0) ? 1u : 0u;
            num_alive = NS - __builtin_popcount((flags >> 4) & 15u);
        } else {
            // [S]:147-164,178-197: simultaneous; lane s decides for snake s
            uint32_t hitmask = 0;  // bit s = lane s: something lies on snake s's head
#pragma unroll
            for (int s = 0; s < NS; ++s) {  // (j is a constant after unrolling)
                if (s == 0) hitmask |= bit_if_any<0>(hit_s[s]);
                else if (s == 1) hitmask |= bit_if_any<1>(hit_s[s]);
                else if (s == 2) hitmask |= bit_if_any<2>(hit_s[s]);
                else hitmask |= bit_if_any<3>(hit_s[s]);
            }
            if (longbody) {  // overflow pieces were collected per lane
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (ballot(((hitl >> (4 * s)) & 15u) != 0) != 0) hitmask |= 1u << s;
            }
            const uint32_t myhead = row_shl<8>(0u, hv) & 0xFFFFu;
            // (lane masks on purpose: `&&` / `||` here compile to nested exec-mask regions with branches)
            const uint32_t hr = (myhead >> 8) - 1u, hc = (myhead & 255u) - 1u;  // head row / column in the grid, or >= dim
            const uint32_t deadmask = (uint32_t)NSMASK &
                                      ((uint32_t)lanes_where<CMP_ULT>(hv, 1u << 16) | (uint32_t)lanes_where<CMP_UGE>(hr, (uint32_t)dim) |
                                       (uint32_t)lanes_where<CMP_UGE>(hc, (uint32_t)dim) | hitmask);
            const bool dead = in_mask((uint64_t)deadmask);
            if (RULES == MSNAKE_RULES_ADVERSARIAL && deadmask != 0) {
                // [A]:183-186: every piece of a snake that dies this step (its out-of-grid head
                // included) joins the fruit list, and spare_fruits grows by len per piece
                int nlist = (int)rdlane(hv, HDR_NLIST), spare = (int)rdlane(hv, HDR_SPARE);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int len_s = (int)ln[s];
                    if (((deadmask >> s) & 1u) && len_s > 0) {
                        // list entry nlist + i <- piece i: lane L of the first chunk takes ring slot hp0 + L - nlist
                        const uint32_t moved = (uint32_t)__shfl((int)cr[s], ((int)(wc[s] >> SN_C_HP0_SHIFT) + lane - nlist) & 63);
                        if (lane >= nlist && lane < nlist + len_s) fr = moved;
                        fr_dirty = true;
                        for_each_piece(s, cr[s], hp2[s], wc[s], len_s, [&](int i, uint32_t cell) {
                            flist_of()[nlist + i] = (uint16_t)cell;
                        });
                        nlist += len_s;
                        spare += len_s * len_s;
                    }
                }
                HV_SET_C(HDR_NLIST, (uint32_t)nlist);
                HV_SET_C(HDR_SPARE, (uint32_t)spare);
            }
            hv = dead ? (hv & 0xFFFFu) : hv;  // snakes[idx] = []
            const uint32_t main_dead = deadmask & 1u;
            if (main_dead) reward = -1.0f;
            done = main_dead | ((max_steps - 1u - t) >> 31);  // t >= max_steps (both below 2^16), without a predicate
            num_alive = NS - __builtin_popcount(deadmask);
        }
        HV_SET_C(HDR_T, t);

        STAMP(3);
        LANE_FENCE();
        DBG_EXIT(4)
        // ---- 4. vec layer: episode statistics and auto reset -----------------------------------
        LANE_FENCE();
        CONFIG_FENCE();
        float ep_ret = __uint_as_float(rdlane(hv, HDR_EP_RETURN)) + reward;
        uint32_t ep_len = rdlane(hv, HDR_EP_LEN) + 1;
        float out_ret = 0.0f;
        uint32_t out_len = 0;
        done = uni(done);
#ifdef MSNAKE_DBG_STAGES
        if (dbg & 0x1000) done = 0;
#endif
        asm volatile("" : "+s"(done));  // (keeps it one SGPR: as a predicate it becomes a lane mask, a select and a compare)
#ifdef MSNAKE_LATE_REFILL
        slow_step = done | (any_eat ? 1u : 0u);
#endif
        if (done) {
            out_ret = ep_ret; out_len = ep_len;
            // logging totals stay in the env record (summed by msnake_get_stats): same-address
            // atomics from every finishing env would serialise at ~12 ns each.  An episode counts once:
            // without auto-reset a finished env keeps reporting done until the caller resets it.
            const uint32_t fl = rdlane(hv, HDR_FLAGS);
            if (!(fl & HDR_FLAG_FINISHED)) {
                if (lane == HDR_ACC_EPISODES) hv += 1u;
                if (lane == HDR_ACC_LEN) hv += ep_len;
                const bool carry = ballot(lane == HDR_ACC_LEN && hv < ep_len) != 0;  // 64-bit length total
                if (carry && lane == HDR_ACC_LEN_HI) hv += 1u;
                if (lane == HDR_ACC_RETURN) hv += (uint32_t)(int)ep_ret;
                HV_SET_C(HDR_FLAGS, fl | HDR_FLAG_FINISHED);
            }
            if (auto_reset) {
                ep_ret = 0.0f; ep_len = 0;
                __builtin_amdgcn_s_setprio(3);
                do_reset();
            }
        }
        HV_SET_C(HDR_EP_RETURN, __float_as_uint(ep_ret));
        HV_SET_C(HDR_EP_LEN, ep_len);
        STAMP(4);
        DBG_EXIT(5)
        STAMP_FLAG((unsigned long long)(any_eat ? 1 : 0) | (done ? 2ull : 0ull));
        SPAN_FLAG((any_eat ? 1u : 0u) | (done ? 2u : 0u));
        SPAN(2);
        if (lane == 0) {
            rew_t[e] = reward;
            done_t[e] = (uint8_t)done;
            if (info_t) {
                int4 iv;
                iv.x = (int)__float_as_uint(out_ret); iv.y = (int)out_len; iv.z = num_alive; iv.w = (int)done;
                reinterpret_cast<int4*>(info_t)[e] = iv;
            }
        }
    }

    // ---- 6. paint the observation over the background, in reference order ----------------------
    LANE_FENCE();
    CONFIG_FENCE();
    if (obs_t) {
#ifdef MSNAKE_PRIO_PAINT
        __builtin_amdgcn_s_setprio(MSNAKE_PRIO_PAINT);
#endif
        wave_sync();
        uint8_t* px = img + (CAN_ALIGN ? obs_shift_of() : 0u);
        // fruits first ([S]:43-44): red in every view; the background is already black
        if (RULES == MSNAKE_RULES_ADVERSARIAL) {
            // list entries outside the grid (dead out-of-grid heads) would land on the wall ring,
            // which the reference paints last: skip them
            for_each_fruit((int)rdlane(hv, HDR_NLIST), [&](int, uint32_t cell) {
                if (!in_grid(cell, dim)) return;
                const int off = ((int)(cell >> 8) * W + (int)(cell & 255u)) * (C * K);
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int v = 0; v < VIEWS; ++v) px[off + k * C + 3 * v] = 255;
            });
        } else if (lane >= FR0 && lane < FR0 + nf) {
            const uint32_t cell = hv & 0xFFFFu;
            // (inline fruits are always inside the grid: resets and respawns place them there and
            //  msnake_set_state refuses others; MODE 2 renders freshly installed states and keeps the clip)
            if (MODE != 2 || in_grid(cell, dim)) {
                const int off = ((int)(cell >> 8) * W + (int)(cell & 255u)) * (C * K);
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int v = 0; v < VIEWS; ++v) px[off + k * C + 3 * v] = 255;
            }
        }
        // snakes in index order, head over body ([S]:46-50, draw_snake :24-33); LDS writes of one
        // wave execute in program order, which is what makes later painters win
        const uint32_t flags = RULES == MSNAKE_RULES_NEW_WORLD ? rdlane(hv, HDR_FLAGS) : 0u;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const uint32_t w0 = rdlane(hv, SN_A(j));
            if (RULES == MSNAKE_RULES_NEW_WORLD && !((flags >> j) & 1u)) continue;  // [N]:219
            if (MODE == 2) pidx[j] = (lane - (int)(rdlane(hv, SN_C(j)) >> SN_C_HP0_SHIFT)) & 63;
            for_each_piece_at(j, cr[j], pidx[j], w0, (int)(w0 >> 16), [&](int i, uint32_t cell) {
                // Only a HEAD can lie outside the grid (msnake_set_state refuses anything else), and after
                // a step or a reset no live snake's head does ([S]:147-164 / [N]:117-119 clear it): the
                // clip against the wall ring is needed only when a freshly installed state is rendered.
                if (MODE == 2 && !in_grid(cell, dim)) return;
                const int off = ((int)(cell >> 8) * W + (int)(cell & 255u)) * (C * K);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    uint8_t* q = px + off + k * C;
                    // (j is a constant after unrolling: one of the four calls remains)
                    if (j == 0) paint_pixel<0, VIEWS>(q, i == 0);
                    else if (j == 1) paint_pixel<1, VIEWS>(q, i == 0);
                    else if (j == 2) paint_pixel<2, VIEWS>(q, i == 0);
                    else paint_pixel<3, VIEWS>(q, i == 0);
                }
            });
        }
        wave_sync();
        STAMP(5);
        SPAN(3);
#ifdef MSNAKE_DBG_STAGES
        if (dbg == 6) asm volatile("s_endpgm");
#endif
#if defined(MSNAKE_PAD_SALU_COPYOUT)
        {   // (the same experiment at the start of the copy-out)
            uint32_t pad = pk0v;
#pragma unroll
            for (int i = 0; i < MSNAKE_PAD_SALU_COPYOUT; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(pad) :: "scc");
            if (pad == 0xDEADBEEFu) hv = pad;
        }
#endif
        if (align_now) {
            // ---- 7a. aligned copy-out (per-step launches; tapes whose step stride is a multiple of 16 bytes): LDS byte x of the buffer <-> global byte g_al + x, both sides
            //          16-byte aligned; wave instruction i covers the i-th KiB counted from the 128-byte line the image
            //          starts in, so every store instruction writes whole lines (WRITE_SIZE 1.00-1.02 x the image; the
            //          byte-aligned stores of 7 below write the sectors at their instruction boundaries twice: 1.07 x).
            //          The <= 15 bytes in front of the first whole chunk and behind the last one go singly.  Streaming (nt)
            //          or plain stores by obs_store_policy.  Measured (round 3, same box each): nt, 262 144 envs 195-200 vs
            //          214-219 us, 131 072 envs 94.5 vs 110, 65 536 envs 48.6 vs 55.5; plain, 32 768 envs 23.2-23.7 vs
            //          24.9-26.2, 16 384 envs 13.96 vs 15.21; nt, 4 096 envs: no difference.
            uint8_t* obs_env = obs_t + (size_t)e * S;
            const uint32_t obs_shift = obs_shift_of();
            uint8_t* g_al = obs_env - obs_shift;
            const int lead = (int)(((uint32_t)(uintptr_t)g_al >> 4) & 7u);  // chunks between the start of the 128-byte line and g_al
            const int end = (int)obs_shift + DBG_NO_OBS_STORES(S);           // the image = buffer bytes [shift, end)
            // whole chunks: k in [kmin, kmin + nk); lane l of instruction i holds chunk 64 i + l - lead: ONE address pair per
            // side, the instruction offset steps through the KiB, ONE unsigned compare per instruction
            const int kmin = (int)((obs_shift + 15u) >> 4);
            const uint32_t nk = (uint32_t)((end >> 4) - kmin);
            const int k0 = lane - lead;
            const uint32_t rel = (uint32_t)(k0 - kmin);
            const uint8_t* lsrc = img + 16 * k0;
            uint8_t* gdst = g_al + 16 * k0;
            // (two copies of the loop, one per store kind: written as `if (nt) nt-store else store` on one value and one
            //  address, the compiler merges the two stores into a plain one -- the nontemporal hint is droppable metadata --
            //  and a "streaming" launch then thrashes the L2s like a plain one: 374 instead of 196 us at 262 144 envs)
            auto chunks = [&](auto NT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (rel + 64u * i < nk DBG_FEWER_STORES(i)) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(lsrc + 1024 * i);
                        if constexpr (decltype(NT)::value) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(gdst + 1024 * i));
                        else *reinterpret_cast<u32x4*>(gdst + 1024 * i) = v;
                    }
                }
                if (nk + (uint32_t)(kmin + lead) > 256u)                     // (images beyond 4 KiB)
                    for (uint32_t r = rel + 256u; r < nk; r += 64u) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(lsrc + 16 * (r - rel));
                        if constexpr (decltype(NT)::value) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(gdst + 16 * (size_t)(r - rel)));
                        else *reinterpret_cast<u32x4*>(gdst + 16 * (size_t)(r - rel)) = v;
                    }
            };
            if (pk2 & (MODE == 3 ? PK2_STREAM_TAPE : PK2_STREAM_OBS)) chunks(std::true_type{});
            else chunks(std::false_type{});
            // head bytes [shift, 16) on lanes 0..15 (when shift > 0), tail bytes [end & ~15, end) on lanes 16..31: lane l is on
            // iff l - first < count with (first, count) = (shift, (16 - shift) & 15) resp. (16, end & 15) -- vector selects and ONE
            // compare (as a chain of bools this predicate was 17 scalar mask instructions; images are at least 32 bytes long)
            const bool head = lane < 16;
            const int byte = head ? lane : (end & ~15) + (lane - 16);
            const uint32_t first = head ? obs_shift : 16u;
            const uint32_t count = head ? ((16u - obs_shift) & 15u) : ((uint32_t)end & 15u);
            if ((uint32_t)lane - first < count) g_al[byte] = img[byte];
        } else if (K == 1) {
            // ---- 7. the persistent tape kernel's copy-out (and, until round 3, every launch's): LDS image -> HBM, 16 bytes
            //         per lane, 1 KiB contiguous per wave instruction.  The 3969-byte images are not 16-byte multiples,
            //         so the global side is byte-aligned (the hardware splits the lines that straddle; the sectors at the
            //         instruction boundaries go out twice: WRITE_SIZE 1.07 x the image); the last S%16 bytes go singly.
            //         The notes below were measured on per-step launches in rounds 1-2, which now take step 7a.
            uint8_t* obs_env = obs_t + (size_t)e * S;
            const int nfull = DBG_NO_OBS_STORES(S) >> 4;
            // One LDS read -> one store per KiB, in a plain loop.  (Measured and rejected, same box:
            // all LDS reads first and then the stores back to back -- fewer instructions, but 7.65
            // instead of 6.9 us per launch at 4 096 envs: a wave's burst of stores sits in its CU's
            // memory pipe ahead of the state loads of the waves that start later; pausing between
            // the stores (s_sleep) or storing the last KiB first changes nothing.)
            // (Streaming stores for the interior and plain ones for the first / last 128 bytes of each
            //  image, so that the edge cache lines two neighbouring images share meet in L2: WRITE_SIZE
            //  drops from 4 424 to 4 189 B per env at 262 144 envs, and the launch takes 387 instead of
            //  206-232 us -- "mixing the two store kinds on one region is ruinous", round 2 concluded.  387 us is what plain
            //  stores take there: most likely the compiler had merged the nt store of that experiment into the plain one, as
            //  it did in round 3's first aligned copy-out, see step 7a.)
            // (Round 2, same box each: the LDS read of KiB i+1 in flight while KiB i is stored -- 6.52-6.58 vs
            //  6.52-6.58 us, no change; other cache policies for the 16-byte stores than `nt`: sc1 6.2-6.5,
            //  sc0 sc1 6.1-6.4, sc0 7.0, nt sc1 8.3, plain 7.6 against 5.91 us for `nt`.  A quarter of the stores
            //  takes as long as all of them (diagnostic build, stage bit 9): the tail is their latency.)
            // (the first four KiB unrolled: one address pair and instruction offsets instead of loop arithmetic)
            const uint8_t* lsrc = img + 16 * lane;
            uint8_t* gdst = obs_env + 16 * lane;
            if (pk2 & (MODE == 3 ? PK2_STREAM_TAPE : PK2_STREAM_OBS)) {  // streaming (nt) stores, see msnake_capi.hip: obs_store_policy
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (lane < nfull - 64 * i DBG_FEWER_STORES(i))
                        __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(lsrc + 1024 * i), reinterpret_cast<u32x4_unaligned*>(gdst + 1024 * i));
                for (int k = lane + 256; k < nfull; k += 64)
                    __builtin_nontemporal_store(reinterpret_cast<const u32x4*>(img)[k], reinterpret_cast<u32x4_unaligned*>(obs_env + 16 * k));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (lane < nfull - 64 * i)
                        *reinterpret_cast<uint4_unaligned*>(gdst + 1024 * i) = *reinterpret_cast<const uint4*>(lsrc + 1024 * i);
                for (int k = lane + 256; k < nfull; k += 64)
                    *reinterpret_cast<uint4_unaligned*>(obs_env + 16 * k) = reinterpret_cast<const uint4*>(img)[k];
            }
            const int tail = (nfull << 4) + lane;
            if (tail < DBG_NO_OBS_STORES(S)) obs_env[tail] = img[tail];
        } else {
            // ---- 7'. fused WarpFrame: the LDS image is already K-fold wide; every row of it is
            //          stored to K consecutive output rows.  Rows are W*K*C bytes = a whole number of
            //          dwords (84 pixels), so each lane moves aligned dwords: one LDS read feeds K
            //          stores of 256 contiguous bytes per wave instruction.
            uint32_t* out = reinterpret_cast<uint32_t*>(obs_t + (size_t)e * S * (K * K));
            const uint32_t* src = reinterpret_cast<const uint32_t*>(img);
            const int rowdw = (W * K * C) >> 2;
            if (MODE != 3 && (pk2 & PK2_STREAM_OBS)) {
                // ---- 7b. streaming launches of the fused frames (>= 400 MiB per launch, far beyond the Infinity Cache; the
                //          launch glue sets the bit only when every frame is 16-byte aligned): the frame is ONE contiguous run
                //          of W*K rows of rowB bytes and leaves as 16-byte-per-lane nt stores, every wave instruction writing
                //          1 KiB of whole 128-byte lines (the first one starts at the line the frame starts in: its leading
                //          lanes sit out).  A chunk inside one output row is 16 contiguous bytes of the LDS row it replicates
                //          (dword-aligned: rows are whole dwords, not 16-byte multiples); a chunk that straddles two rows holds
                //          the last <= 12 bytes of one and the first <= 12 of the next, wall pixels (255) in every row.
                //          Measured, same box: 398 vs 418 us at 32 768 envs (2 GB), 105.8 vs 108.6 at 8 192; at 4 096 envs
                //          (248 MiB, still cache resident) the dword stores below win, 37.7 vs 42.7 (plain) / 55.7 (nt).
                const uint32_t rowB = (uint32_t)(W * K * C), OB = rowB * (uint32_t)(W * K);
                uint8_t* g0 = obs_t + (size_t)e * OB;
                const int32_t head = (int32_t)((uint32_t)(uintptr_t)g0 & 127u);
                const float inv = 1.0f / (float)rowB;
                for (int32_t o = 16 * lane - head; o < (int32_t)OB; o += 1024) {
                    if (o < 0) continue;  // (first instruction only)
                    // output row and column of the chunk: o / rowB by a float reciprocal (o < 2^24: exact in float) + one fix-up
                    int32_t orow = (int32_t)((float)o * inv);
                    int32_t c = o - orow * (int32_t)rowB;
                    if (c < 0) { orow -= 1; c += (int32_t)rowB; }
                    if (c >= (int32_t)rowB) { orow += 1; c -= (int32_t)rowB; }
                    const bool straddle = c > (int32_t)rowB - 16;
                    const int32_t cc = straddle ? (int32_t)rowB - 16 : c;
                    u32x4 v = *reinterpret_cast<const u32x4_dw*>(img + (uint32_t)(orow / K) * rowB + (uint32_t)cc);
                    if (straddle) v = u32x4{~0u, ~0u, ~0u, ~0u};
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(g0 + o));
                }
            } else {
                for (int r = 0; r < W; ++r)
                    for (int q = lane; q < rowdw; q += 64) {
                        const uint32_t v = src[r * rowdw + q];
#pragma unroll
                        for (int rr = 0; rr < K; ++rr) out[(size_t)(r * K + rr) * rowdw + q] = v;
                    }
            }
        }
    }
    STAMP(6);
    }  // step loop

    // ---- 5. state write-back (once per launch): the record; the body rings went out slot by slot --
    if (MODE != 2) {
#ifdef MSNAKE_LATE_REFILL
        // Refill the parked Philox draws BEHIND the wave's observation stores, by the waves that have time for it: a wave
        // that stayed on the fast path in this launch (no respawn, no episode end: 94 % of them, and they finish ~1 us
        // ahead of the launch's last waves) tops the cache up once MSNAKE_LATE_REFILL draws (default build: off) are used,
        // so that a later respawn or reset finds its draws parked instead of evaluating Philox ahead of its logic.
        // MSNAKE_LATE_REFILL = 0: every wave refills once fewer than 5*NS draws remain (measured: slower).
        if (PCACHE && !short_rec && MODE != 3) {
            const uint32_t ctr_lo = rdlane(hv, HDR_CTR_LO);
            bool late;
            if (refilled != 0)  // Philox already ran (ahead of some logic): again only if what it left cannot fill the parking words
                late = !(draws_n == 64 && ctr_lo - draw_base + HDR_PC_N <= 64u);
            else if (MSNAKE_LATE_REFILL == 0)
                late = rdlane(hv, HDR_PC_VALID) != 1u || ctr_lo - rdlane(hv, HDR_PC_BASE) > HDR_PC_N - 5u * NS;  // (mod 2^32, like ensure_draws)
            else
                late = slow_step == 0 && (rdlane(hv, HDR_PC_VALID) != 1u || ctr_lo - rdlane(hv, HDR_PC_BASE) >= (uint32_t)(MSNAKE_LATE_REFILL));
            if (late) {
                SPAN_FLAG(8u);
                refill_draws(ctr_lo, rdlane(hv, HDR_CTR_HI));
            }
        }
#endif
        if (PCACHE && (!NUMFLAGS ? (refilled != 0 && !short_rec) : uni(refilled) == 1u)) {
            // Philox ran in this launch: its unused draws go into the record for the launches to come
            upper_is_dirty();
            const uint32_t ctr_lo = rdlane(hv, HDR_CTR_LO);
            const uint32_t off = ctr_lo - draw_base;
            if (draws_n == 64 && off + HDR_PC_N <= 64u) {
                const uint32_t sh = (uint32_t)__shfl((int)draws, lane - HDR_PC_FIRST + (int)off);
                hv = lane >= HDR_PC_FIRST ? sh : hv;
                HV_SET_C(HDR_PC_BASE, ctr_lo);
                HV_SET_C(HDR_PC_VALID, 1u);
            } else {
                HV_SET_C(HDR_PC_VALID, 0u);
            }
        }
        // (the number of words as ONE scalar select and one lane compare, not an or of lane masks)
        const uint32_t rec_words = uni((uint32_t)MSNAKE_HDR_SHORT_WORDS + ((!NUMFLAGS && short_rec) ? 0u : rec_extra));
        if ((uint32_t)lane < rec_words) {
            uint32_t ee = (uint32_t)e;
            if (MODE == 3) asm volatile("" : "+s"(ee));  // (the pointer is not held across the step loop)
            // (round 3: the nt hint on this store, on the ring-sector stores and on the reward store -- nothing left dirty in the
            //  L2s for the end-of-kernel release -- changes nothing: 5.77-5.79 / 5.76-5.79 vs 5.75-5.78 us, and costs with the
            //  4-byte reward stores in: 5.92-5.95)
            (reinterpret_cast<uint32_t*>(state) + (size_t)ee * MSNAKE_HDR_WORDS)[lane] = hv;
        }
        if (RULES == MSNAKE_RULES_ADVERSARIAL && fr_dirty) fl0_of()[lane] = (uint16_t)fr;
    }
#ifdef MSNAKE_END_WAIT  // experiment: every wave stays until its stores have been acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef MSNAKE_HAVE_SPAN
    if (p.dbg_span) {
        SPAN(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SPAN(5);
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) p.dbg_span[(size_t)e * 8 + 6] = dbgf | ((xcc & 15u) << 8);
    }
#endif
}

// Sum the per-env logging totals into stats[0..4] (msnake_get_stats; off the step path).
__global__ __launch_bounds__(256) void msnake_stats_kernel(uint32_t* __restrict__ hdr, int nenv,
                                                           unsigned long long* __restrict__ stats, int clear) {
    long long ep = 0, ln = 0, rt = 0, er = 0;
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < nenv; e += (int)(gridDim.x * blockDim.x)) {
        uint32_t* h = hdr + (size_t)e * MSNAKE_HDR_WORDS;
        ep += h[HDR_ACC_EPISODES]; ln += (long long)(((unsigned long long)h[HDR_ACC_LEN_HI] << 32) | h[HDR_ACC_LEN]);
        rt += (int)h[HDR_ACC_RETURN]; er += h[HDR_ACC_ERRORS];
        if (clear) h[HDR_ACC_EPISODES] = h[HDR_ACC_LEN] = h[HDR_ACC_LEN_HI] = h[HDR_ACC_RETURN] = h[HDR_ACC_ERRORS] = 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ep += __shfl_down(ep, o); ln += __shfl_down(ln, o); rt += __shfl_down(rt, o); er += __shfl_down(er, o);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&stats[0], (unsigned long long)ep);
        atomicAdd(&stats[1], (unsigned long long)ln);
        atomicAdd(&stats[2], (unsigned long long)rt);
        atomicAdd(&stats[4], (unsigned long long)er);
    }
}

hipError_t launch_stats(uint32_t* hdr, int nenv, unsigned long long* stats, int clear, hipStream_t stream) {
    const int blocks = nenv < 256 * 64 ? (nenv + 255) / 256 : 64;
    hipLaunchKernelGGL(msnake_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, hdr, nenv, stats, clear);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// canonical state export / import (msnake_get_state[_all] / msnake_set_state[_all]; checkpoint and
// test path, off the step path).  Word layout per env: include/msnake.h.  One wave per env.
// ------------------------------------------------------------------------------------------------
struct StateView {
    uint32_t* hdr; uint16_t* body0; uint16_t* ovf; uint16_t* fl0; uint16_t* flist;
    int32_t nenv, dim, ns, nf, cap, fcap, rules;
};

static StateView state_view(const StepParams& p, int rules) {
    return StateView{p.hdr, p.body0, p.ring, p.fl0, p.flist, p.nenv, p.dim, p.n_snakes, p.n_fruits, p.rest.cap, p.fcap, rules};
}

__device__ __forceinline__ int state_fruit_count(const StateView& v, const uint32_t* h) {
    return v.rules == MSNAKE_RULES_ADVERSARIAL ? (int)h[HDR_NLIST] : v.nf;
}

__global__ __launch_bounds__(256) void msnake_state_sizes_kernel(StateView v, int env0, int count, uint32_t* __restrict__ need) {
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= count) return;
    const uint32_t* h = v.hdr + (size_t)(env0 + w) * MSNAKE_HDR_WORDS;
    uint32_t n = 8u + 2u * (uint32_t)state_fruit_count(v, h);
    for (int s = 0; s < v.ns; ++s) n += 6u + 2u * (h[SN_A(s)] >> 16);
    need[w] = n;
}

__global__ __launch_bounds__(256) void msnake_state_pack_kernel(StateView v, int env0, int count,
                                                                const uint64_t* __restrict__ offsets, int32_t* __restrict__ words) {
    const int lane = (int)(threadIdx.x & 63u);
    const int w = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w >= count) return;
    const int e = env0 + w;
    const uint32_t* h = v.hdr + (size_t)e * MSNAKE_HDR_WORDS;
    int32_t* out = words + offsets[w];
    const bool adv = v.rules == MSNAKE_RULES_ADVERSARIAL, nw = v.rules == MSNAKE_RULES_NEW_WORLD;
    const int nfr = state_fruit_count(v, h), fr0 = nw ? HDR_FRUIT0_N : HDR_FRUIT0_S;
    if (lane == 0) {
        out[0] = (int32_t)h[HDR_T]; out[1] = (int32_t)h[HDR_CTR_LO]; out[2] = (int32_t)h[HDR_CTR_HI];
        out[3] = (int32_t)h[HDR_SPARE]; out[4] = (int32_t)h[HDR_EP_LEN]; out[5] = (int32_t)h[HDR_EP_RETURN];
        out[6] = nfr; out[7] = v.ns | ((h[HDR_FLAGS] & HDR_FLAG_FINISHED) ? 0x100 : 0);
    }
    size_t k = 8;
    for (int f = lane; f < nfr; f += 64) {
        const uint32_t c = adv ? (uint32_t)v.flist[(size_t)e * v.fcap + f] : (h[fr0 + f] & 0xFFFFu);
        out[k + 2 * f] = (int32_t)(c >> 8) - 1;
        out[k + 2 * f + 1] = (int32_t)(c & 255u) - 1;
    }
    k += 2 * (size_t)nfr;
    for (int s = 0; s < v.ns; ++s) {
        const uint32_t wA = h[SN_A(s)], wC = h[SN_C(s)];
        const int len = (int)(wA >> 16), ohp = (int)(wA & 0xFFFFu), hp0 = (int)((wC >> SN_C_HP0_SHIFT) & 63u);
        const int vel = (int)((wC >> 16) & 7u);
        if (lane == 0) {
            out[k] = len;
            out[k + 1] = vel == 1 ? 1 : vel == 3 ? -1 : 0;
            out[k + 2] = vel == 2 ? 1 : vel == 4 ? -1 : 0;
            out[k + 3] = (int32_t)h[SN_B(s)];
            out[k + 4] = nw ? (int32_t)((h[HDR_FLAGS] >> s) & 1u) : 1;
            out[k + 5] = nw ? (int32_t)((h[HDR_FLAGS] >> (4 + s)) & 1u) : 0;
        }
        for (int i = lane; i < len; i += 64) {
            uint32_t c;
            if (i < 64) {
                c = v.body0[((size_t)e * v.ns + s) * 64 + ((hp0 + i) & 63)];
            } else {
                int idx = ohp + i - 64;
                idx = idx >= v.cap ? idx - v.cap : idx;
                c = v.ovf[((size_t)e * v.ns + s) * v.cap + idx];
            }
            out[k + 6 + 2 * (size_t)i] = (int32_t)(c >> 8) - 1;
            out[k + 6 + 2 * (size_t)i + 1] = (int32_t)(c & 255u) - 1;
        }
        k += 6 + 2 * (size_t)len;
    }
}

// status[0]: number of rejected envs, status[1]: min over them of (local index + 1) << 8 | MSNAKE_ST_* reason (ONE
// atomicMin on ~0u: index and reason of the first rejected env cannot come from two different waves)
#define MSNAKE_ST_SHORT 1      // buffer too short / truncated
#define MSNAKE_ST_SNAKES 2     // snake count differs from the handle
#define MSNAKE_ST_FRUITS 3     // fruit count differs from the handle / exceeds the list capacity
#define MSNAKE_ST_CELL 4       // a cell outside [-1, dim]
#define MSNAKE_ST_LEN 5        // a body length outside [0, cap - 2]
__global__ __launch_bounds__(256) void msnake_state_unpack_kernel(StateView v, int env0, int count,
                                                                  const uint64_t* __restrict__ offsets,
                                                                  const int32_t* __restrict__ words, uint32_t* __restrict__ status) {
    const int lane = (int)(threadIdx.x & 63u);
    const int w = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w >= count) return;
    const int e = env0 + w;
    const int32_t* in = words + offsets[w];
    const long long n = (long long)(offsets[w + 1] - offsets[w]);
    const bool adv = v.rules == MSNAKE_RULES_ADVERSARIAL, nw = v.rules == MSNAKE_RULES_NEW_WORLD;
    auto cell_ok = [&](int c0, int c1) { return c0 >= -1 && c0 <= v.dim && c1 >= -1 && c1 <= v.dim; };
    // ---- pass 1: validate everything before anything is written
    int bad = 0;
    int nfr = 0;
    if (n < 8) bad = MSNAKE_ST_SHORT;
    else if ((in[7] & ~0x100) != v.ns) bad = MSNAKE_ST_SNAKES;  // (bit 8: the episode-finished flag)
    else {
        nfr = in[6];
        if (adv ? (nfr < 0 || nfr > v.fcap) : nfr != v.nf) bad = MSNAKE_ST_FRUITS;
        else if (n < 8 + 2LL * nfr) bad = MSNAKE_ST_SHORT;
    }
    long long k = 8;
    if (!bad) {
        bool okc = true;
        for (int f = lane; f < nfr; f += 64) {
            const int c0 = in[k + 2 * f], c1 = in[k + 2 * f + 1];
            // the adversarial fruit list may hold the off-grid head of a dead snake ([A]:183-185); the inline
            // fruits of the other rule sets are only ever placed inside the grid
            okc = okc && (adv ? cell_ok(c0, c1) : (c0 >= 0 && c0 < v.dim && c1 >= 0 && c1 < v.dim));
        }
        if (__builtin_amdgcn_ballot_w64(!okc) != 0) bad = MSNAKE_ST_CELL;
        k += 2LL * nfr;
    }
    for (int s = 0; s < v.ns && !bad; ++s) {
        if (n < k + 6) { bad = MSNAKE_ST_SHORT; break; }
        const int len = in[k];
        if (len < 0 || len > v.cap - 2) { bad = MSNAKE_ST_LEN; break; }
        if (n < k + 6 + 2LL * len) { bad = MSNAKE_ST_SHORT; break; }
        bool okc = true;
        for (int i = lane; i < len; i += 64) {
            const int c0 = in[k + 6 + 2LL * i], c1 = in[k + 6 + 2LL * i + 1];
            okc = okc && cell_ok(c0, c1);
            // only a head may be one step outside the grid (where a reference snake can be, for one step)
            if (i > 0) okc = okc && c0 >= 0 && c0 < v.dim && c1 >= 0 && c1 < v.dim;
        }
        if (__builtin_amdgcn_ballot_w64(!okc) != 0) { bad = MSNAKE_ST_CELL; break; }
        k += 6 + 2LL * len;
    }
    if (bad) {
        if (lane == 0) {
            atomicAdd(&status[0], 1u);
            atomicMin(&status[1], (((uint32_t)w + 1u) << 8) | (uint32_t)bad);
        }
        return;
    }
    // ---- pass 2: the record (lane l builds word l; logging totals are not part of the canonical
    //      state and stay; every other word, the parked Philox draws included, is cleared)
    uint32_t* h = v.hdr + (size_t)e * MSNAKE_HDR_WORDS;
    uint32_t hv = (lane >= HDR_ACC_EPISODES && lane <= HDR_ACC_LEN_HI) ? h[lane] : 0u;
    HV_SET_C(HDR_T, in[0]); HV_SET_C(HDR_CTR_LO, in[1]); HV_SET_C(HDR_CTR_HI, in[2]); HV_SET_C(HDR_SPARE, in[3]);
    HV_SET_C(HDR_EP_LEN, in[4]); HV_SET_C(HDR_EP_RETURN, in[5]);
    if (adv) HV_SET_C(HDR_NLIST, nfr);
    k = 8;
    const int fr0 = nw ? HDR_FRUIT0_N : HDR_FRUIT0_S;
    for (int f = lane; f < nfr; f += 64) {
        const uint32_t c = ((uint32_t)(in[k + 2 * f] + 1) << 8) | (uint32_t)(in[k + 2 * f + 1] + 1);
        if (adv) {
            v.flist[(size_t)e * v.fcap + f] = (uint16_t)c;
            if (f < 64) v.fl0[(size_t)e * 64 + f] = (uint16_t)c;
        }
    }
    if (!adv)
        for (int f = 0; f < nfr; ++f)
            HV_SET_DYN(fr0 + f, ((uint32_t)(in[k + 2 * f] + 1) << 8) | (uint32_t)(in[k + 2 * f + 1] + 1));
    k += 2LL * nfr;
    uint32_t flags = 0;
    for (int s = 0; s < v.ns; ++s) {
        const int len = in[k], v0 = in[k + 1], v1 = in[k + 2];
        const int vel = (v0 == 1 && v1 == 0) ? 1 : (v0 == 0 && v1 == 1) ? 2 : (v0 == -1 && v1 == 0) ? 3 : (v0 == 0 && v1 == -1) ? 4 : 0;
        if (in[k + 4]) flags |= 1u << s;
        if (in[k + 5]) flags |= 16u << s;
        uint32_t headc = 0;
        if (len > 0) headc = ((uint32_t)(in[k + 6] + 1) << 8) | (uint32_t)(in[k + 7] + 1);
        for (int i = lane; i < len; i += 64) {
            const uint32_t c = ((uint32_t)(in[k + 6 + 2LL * i] + 1) << 8) | (uint32_t)(in[k + 6 + 2LL * i + 1] + 1);
            if (i < 64) v.body0[((size_t)e * v.ns + s) * 64 + i] = (uint16_t)c;          // head in slot 0
            else v.ovf[((size_t)e * v.ns + s) * v.cap + (i - 64)] = (uint16_t)c;         // overflow from position 0
        }
        HV_SET(SN_A(s), (uint32_t)len << 16);
        HV_SET(SN_B(s), in[k + 3]);
        HV_SET(SN_C(s), headc | ((uint32_t)vel << 16));
        k += 6 + 2LL * len;
    }
    if (in[7] & 0x100) flags |= HDR_FLAG_FINISHED;  // restored as exported: the episode is not counted a second time
    HV_SET_C(HDR_FLAGS, flags);
    h[lane] = hv;
}

hipError_t launch_state_sizes(const StepParams& p, int rules, int env0, int count, uint32_t* need_words, hipStream_t stream) {
    hipLaunchKernelGGL(msnake_state_sizes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       state_view(p, rules), env0, count, need_words);
    return hipGetLastError();
}

hipError_t launch_state_pack(const StepParams& p, int rules, int env0, int count, const uint64_t* offsets, int32_t* words,
                             hipStream_t stream) {
    hipLaunchKernelGGL(msnake_state_pack_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, stream,
                       state_view(p, rules), env0, count, offsets, words);
    return hipGetLastError();
}

hipError_t launch_state_unpack(const StepParams& p, int rules, int env0, int count, const uint64_t* offsets,
                               const int32_t* words, uint32_t* status, hipStream_t stream) {
    hipLaunchKernelGGL(msnake_state_unpack_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, stream,
                       state_view(p, rules), env0, count, offsets, words, status);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launch glue (called from the C-ABI in msnake_capi.hip)
// ------------------------------------------------------------------------------------------------
// M of the kernel's division-free x / dim (safe_cell): floor(2^(14 + bits(dim)) / dim) + 1 <= 2^15 + 1
static uint32_t div_magic(uint32_t dim) {
    uint32_t bits = 0;
    while ((dim >> bits) != 0) ++bits;
    return (uint32_t)((1ull << (14u + bits)) / dim) + 1u;
}

template <int RULES, int NS, int K>
static hipError_t launch_k(const StepParams& p, int mode, int epb, hipStream_t stream) {
    const uint32_t pk0 = (uint32_t)p.dim | ((uint32_t)p.n_fruits << 6) | ((uint32_t)p.action_stride << 12) |
                         ((uint32_t)(p.auto_reset ? 1 : 0) << 15) | ((uint32_t)p.rest.max_steps << 16);
    const uint32_t pk1 = (uint32_t)p.S | ((uint32_t)p.rest.cap << 16);

    size_t lds_wave = (size_t)p.lds_per_wave;
    if (mode == 3 && K == 1) {  // + the pristine background copy; fewer envs per workgroup if that needs it
        lds_wave += (size_t)p.img_bytes;
        while (epb > 1 && lds_wave * (size_t)epb > 64 * 1024) epb >>= 1;
        if (lds_wave > 64 * 1024) return hipErrorInvalidValue;
    }
    const size_t lds = lds_wave * (size_t)epb;
    // fused x4 / x7 frames stream only in the flat 16-byte shape, which needs every frame 16-byte aligned (the dword stores
    // never stream: 256 B per wave instruction, 2.5x slower with nt); the persistent kernel's fused copy-out never does
    bool nt_obs = p.rest.stream_obs != 0;
    if (K > 1) {
        const size_t OB = (size_t)p.S * K * K;
        nt_obs = nt_obs && mode != 3 && p.obs && OB % 16 == 0 && ((uintptr_t)p.obs & 15) == 0 && OB < (1u << 24);
    }
    const uint32_t pk2 = (p.short_rec ? PK2_SHORT_REC : 0u) | (nt_obs ? PK2_STREAM_OBS : 0u) |
                         (p.stream_tape && K == 1 ? PK2_STREAM_TAPE : 0u) |
                         (mode == 3 && K == 1 && p.rest.obs_step_stride % 16 == 0 ? PK2_TAPE_ALIGNED : 0u) | ((uint32_t)epb << PK2_EPB_SHIFT) |
                         (div_magic((uint32_t)p.dim) << PK2_DIVM_SHIFT);
    const dim3 grid((unsigned)((((p.nenv + epb - 1) / epb) + 63) & ~63));  // whole groups of 64: see the kernel's XCD swap
    const dim3 block(64u * (unsigned)epb);
#define MSNAKE_LAUNCH(M)                                                                                       \
    hipLaunchKernelGGL((msnake_step_kernel<RULES, NS, M, K>), grid, block, lds, stream, p.state, p.obs, p.actions, \
                       p.rest.rew, p.rest.done, p.nenv, pk0, pk1, pk2, p.rest)
    switch (mode) {
        case 0: MSNAKE_LAUNCH(0); break;
        case 1: MSNAKE_LAUNCH(1); break;
        case 3: MSNAKE_LAUNCH(3); break;
        default: MSNAKE_LAUNCH(2); break;
    }
#undef MSNAKE_LAUNCH
    return hipGetLastError();
}

template <int RULES, int NS>
static hipError_t launch_ns(const StepParams& p, int mode, int epb, hipStream_t stream) {
    switch (p.obs_scale) {
        case 1: return launch_k<RULES, NS, 1>(p, mode, epb, stream);
        case 4: return launch_k<RULES, NS, 4>(p, mode, epb, stream);
        case 7: return launch_k<RULES, NS, 7>(p, mode, epb, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int RULES>
static hipError_t launch_rules(const StepParams& p, int mode, int epb, hipStream_t stream) {
    switch (p.n_snakes) {
        case 1: return launch_ns<RULES, 1>(p, mode, epb, stream);
        case 2: return launch_ns<RULES, 2>(p, mode, epb, stream);
        case 3: return launch_ns<RULES, 3>(p, mode, epb, stream);
        case 4:
            if (RULES == MSNAKE_RULES_NEW_WORLD) return launch_ns<MSNAKE_RULES_NEW_WORLD, 4>(p, mode, epb, stream);
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(const StepParams& p, int rules, int mode, int epb, hipStream_t stream) {
    if (epb < 1 || epb > MSNAKE_MAX_ENVS_PER_BLOCK) return hipErrorInvalidValue;
    switch (rules) {
        case MSNAKE_RULES_SNAKE_ENV: return launch_rules<MSNAKE_RULES_SNAKE_ENV>(p, mode, epb, stream);
        case MSNAKE_RULES_NEW_WORLD: return launch_rules<MSNAKE_RULES_NEW_WORLD>(p, mode, epb, stream);
        case MSNAKE_RULES_ADVERSARIAL: return launch_rules<MSNAKE_RULES_ADVERSARIAL>(p, mode, epb, stream);
        default: return hipErrorInvalidValue;
    }
}

void step_kernel_name(int rules, int n_snakes, int obs_scale, char* out, size_t n) {
    snprintf(out, n, "msnake_step_kernel<%d, %d, 0, %d>", rules, n_snakes, obs_scale);
}

}  // namespace msnake

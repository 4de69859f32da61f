#!/usr/bin/env python3
"""Long parity soak on the GPU box: N envs x T steps, HIP env vs CPU oracle on the same actions.
rew/done/info every step, observation every `--obs-every` steps, full state at the end."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--dim", type=int, default=19)
    ap.add_argument("--snakes", type=int, default=3)
    ap.add_argument("--fruits", type=int, default=None)
    ap.add_argument("--rules", default="snake_env")
    ap.add_argument("--keep", type=float, default=0.6, help="probability of action 0 (snakes live longer)")
    ap.add_argument("--obs-every", type=int, default=25)
    ap.add_argument("--max-steps", type=int, default=2000)
    ap.add_argument("--tape", type=int, default=0, help="> 0: step through msnake_rollout_tape in chunks of this many steps (persistent kernel)")
    ap.add_argument("--store-policy", default="auto", choices=["auto", "plain", "stream"], help="plain = the aligned copy-out")
    args = ap.parse_args()
    import torch
    import msnake
    from oracle.snake_oracle import Oracle

    n, ns = args.envs, args.snakes
    env = msnake.MultiSnakeVecEnv(n, dim=args.dim, n_snakes=ns, n_fruits=args.fruits, rules=args.rules, seed=77,
                                  max_steps=args.max_steps, obs_store_policy=args.store_policy)
    ora = Oracle(n, dim=args.dim, n_snakes=ns, n_fruits=args.fruits, rules=args.rules, seed=77, max_steps=args.max_steps)
    assert np.array_equal(env.reset(), ora.reset())
    rs = np.random.default_rng(5)
    t0 = time.time()
    episodes = maxlen = 0
    def check(t, rew_h, done_h, info_h, obs_d, act):
        nonlocal episodes
        want_obs = obs_d is not None
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(act, threads=16, want_obs=want_obs)
        assert np.array_equal(rew_h, o_rew), t
        assert np.array_equal(done_h, o_done), t
        assert np.array_equal(info_h[:, 2], o_ns) and np.array_equal(info_h[:, 1], o_el), t
        assert np.array_equal(info_h[:, 0].copy().view(np.float32), o_er), t
        if want_obs:
            assert np.array_equal(obs_d.cpu().numpy(), o_obs), t
        episodes += int(o_done.sum())
        if t % 500 == 0:
            print(f"step {t}: ok, episodes so far {episodes}, {time.time() - t0:.0f}s", flush=True)

    def actions():
        act = rs.integers(0, 5, (n, ns)).astype(np.int32)
        return np.where(rs.random((n, ns)) < args.keep, 0, act).astype(np.int32)

    if args.tape > 0:  # the persistent kernel: chunks of args.tape steps, every step's scalars, observations at obs-every
        t = 0
        while t < args.steps:
            m = min(args.tape, args.steps - t)
            acts = np.stack([actions() for _ in range(m)])
            obs_t, rew_t, done_t, info_t = env.rollout_device(torch.from_numpy(acts).to(env.device))
            rew_h, done_h, info_h = rew_t.cpu().numpy(), done_t.cpu().numpy(), info_t.cpu().numpy()
            for j in range(m):
                check(t + j, rew_h[j], done_h[j], info_h[j], obs_t[j] if ((t + j) % args.obs_every == 0 or t + j == args.steps - 1) else None, acts[j])
            t += m
    else:
        for t in range(args.steps):
            act = actions()
            obs_d, rew, done, info = env.step_device(torch.from_numpy(act).to(env.device))
            want_obs = t % args.obs_every == 0 or t == args.steps - 1
            check(t, rew.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy(), obs_d if want_obs else None, act)
    from oracle.snake_oracle import flat_to_state
    for e in range(0, n, max(1, n // 256)):
        st = flat_to_state(env.get_state_words(e))
        assert st == ora.get_state(e), e
        maxlen = max(maxlen, max(len(b) for b in st["snakes"]))
    st = env.stats()
    assert st["errors"] == 0 and st["episodes"] == episodes
    print(f"SOAK OK: {args.rules} {n} envs x {args.steps} steps (store policy {args.store_policy}, tape {args.tape}), {episodes} episodes, "
          f"longest body at the end {maxlen}, {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()

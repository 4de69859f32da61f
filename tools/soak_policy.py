#!/usr/bin/env python3
"""Parity soak under a TRAINED policy: self-play PPO for --train-steps env-steps, then the learnt
policy drives every snake of --envs envs for --steps steps on the HIP env and on the CPU oracle with
the same actions (long bodies, frequent eating: what random play never reaches).  rew/done/info
every step, observation every --obs-every steps, full state at the end."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--snakes", type=int, default=2)
    ap.add_argument("--dim", type=int, default=19)
    ap.add_argument("--train-steps", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--obs-every", type=int, default=10)
    ap.add_argument("--time-envs", type=int, default=0, help="also time msnake_step at this batch size under the policy")
    args = ap.parse_args()
    import torch
    import msnake
    from msnake import selfplay
    from oracle.snake_oracle import Oracle

    n, ns = args.envs, args.snakes
    env = msnake.MultiSnakeVecEnv(n, dim=args.dim, n_snakes=ns, seed=0)
    t0 = time.time()
    model, hist = selfplay.learn(env, total_timesteps=args.train_steps, log_fn=None)
    print(f"trained {hist[-1]['total_timesteps']} steps in {time.time() - t0:.0f}s: eprewmean {hist[-1]['eprewmean 100']:.2f}, "
          f"eplenmean {hist[-1]['eplenmean']:.0f}", flush=True)
    env.close()

    env = msnake.MultiSnakeVecEnv(n, dim=args.dim, n_snakes=ns, seed=123)
    ora = Oracle(n, dim=args.dim, n_snakes=ns, rules="snake_env", seed=123)
    obs = env.reset_device()
    assert np.array_equal(obs.cpu().numpy(), ora.reset())
    longest, eats, episodes = 0, 0, 0
    for t in range(args.steps):
        with torch.no_grad():
            acts = torch.stack([model.step(obs[..., 3 * s:3 * s + 3])[0] for s in range(ns)], 1).to(torch.int32)
        obs, rew, done, info = env.step_device(acts)
        o_obs, o_rew, o_done, o_ns, o_er, o_el = ora.step(acts.cpu().numpy(), threads=8, want_obs=(t % args.obs_every == 0))
        assert np.array_equal(rew.cpu().numpy(), o_rew), t
        assert np.array_equal(done.cpu().numpy().astype(bool), o_done.astype(bool)), t
        ih = info.cpu().numpy()
        assert np.array_equal(ih[:, 2], o_ns) and np.array_equal(ih[:, 1], o_el), t
        if t % args.obs_every == 0:
            assert np.array_equal(obs.cpu().numpy(), o_obs), t
        eats += int((o_rew > 0).sum()); episodes += int(o_done.sum())
        if t % 500 == 0:
            longest = max(longest, max(max(len(b) for b in ora.get_state(e)["snakes"]) for e in range(0, n, 8)))
            print(f"step {t}: ok, episodes {episodes}, eat-steps(main) {eats}, longest sampled body {longest}", flush=True)
    from oracle.snake_oracle import flat_to_state
    for e in range(n):
        assert flat_to_state(env.get_state_words(e)) == ora.get_state(e), e
    assert env.stats()["errors"] == 0
    if args.time_envs:
        # what one msnake_step launch costs under the learnt policy's state distribution
        m = args.time_envs
        env2 = msnake.MultiSnakeVecEnv(m, dim=args.dim, n_snakes=ns, seed=5)
        obs = env2.reset_device()
        us = []
        for t in range(400):
            with torch.no_grad():
                acts = torch.stack([model.step(obs[..., 3 * s:3 * s + 3])[0] for s in range(ns)], 1).to(torch.int32)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            obs, rew, done, info = env2.step_device(acts)
            e1.record()
            torch.cuda.synchronize()
            if t >= 200:
                us.append(e0.elapsed_time(e1) * 1e3)
        st = env2.stats()
        print(f"msnake_step under the learnt policy, {m} envs: median {np.median(us):.2f} us per launch "
              f"(HIP events around single launches; mean episode length {st['ep_len_sum'] / max(1, st['episodes']):.0f})", flush=True)
        env2.close()
        env2 = msnake.MultiSnakeVecEnv(m, dim=args.dim, n_snakes=ns, seed=5)  # same method, random actions
        env2.reset_device()
        us = []
        for t in range(400):
            acts = torch.randint(0, 5, (m, ns), dtype=torch.int32, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env2.step_device(acts)
            e1.record()
            torch.cuda.synchronize()
            if t >= 200:
                us.append(e0.elapsed_time(e1) * 1e3)
        st = env2.stats()
        print(f"msnake_step under uniform random actions, same method: median {np.median(us):.2f} us per launch "
              f"(mean episode length {st['ep_len_sum'] / max(1, st['episodes']):.0f})", flush=True)
        env2.close()
    print(f"POLICY SOAK OK: {n} envs x {args.steps} steps under the learnt policy, {episodes} episodes, "
          f"{eats} eat-steps of the main snake, longest sampled body {longest}")


if __name__ == "__main__":
    main()

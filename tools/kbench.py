#!/usr/bin/env python3
"""Kernel timing sweep on one GPU: step / reset / render-only kernels at several batch sizes.
Times K back-to-back launches with HIP events on the launch stream (python tools/kbench.py)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[4096, 32768, 262144])
    ap.add_argument("--dim", type=int, default=19)
    ap.add_argument("--snakes", type=int, default=3)
    ap.add_argument("--rules", default="snake_env")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--store-policy", default="auto", choices=["auto", "plain", "stream"])
    ap.add_argument("--record-policy", default="auto", choices=["auto", "full", "short"])
    ap.add_argument("--epb", type=int, default=0)
    args = ap.parse_args()
    import torch
    import msnake

    dev = torch.device("cuda", 0)
    for n in args.envs:
        env = msnake.MultiSnakeVecEnv(n, dim=args.dim, n_snakes=args.snakes, rules=args.rules, seed=0, device=dev, obs_scale=args.scale,
                                       obs_store_policy=args.store_policy, record_policy=args.record_policy, envs_per_block=args.epb)
        env.reset_device()
        T = 64
        while T > 8 and T * n * args.snakes * 4 > (32 << 20):  # (the action tape stays below 32 MiB: see bench.py)
            T //= 2
        tape = torch.randint(0, 5, (T, n, args.snakes), dtype=torch.int32, device=dev)
        L, h = env._L, env._h
        obs, rew, done, info = env._obs, env._rew, env._done, env._info

        def timeit(fn, iters):
            fn(20)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(iters)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters

        def tape_steps(obs_ptr):
            def run(k):
                done_k = 0
                while done_k < k:
                    m = min(T, k - done_k)
                    msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), args.snakes, m, obs_ptr, 0,
                                                          rew.data_ptr(), done.data_ptr(), info.data_ptr(), 0,
                                                          env._stream()))
                    done_k += m
            return run

        def rollout(k):
            done_k = 0
            while done_k < k:
                m = min(T, k - done_k)
                msnake._capi.check(L.msnake_rollout_tape(h, tape.data_ptr(), args.snakes, m, obs.data_ptr(), 0,
                                                         rew.data_ptr(), done.data_ptr(), info.data_ptr(), 0,
                                                         env._stream()))
                done_k += m

        def renders(k):
            for _ in range(k):
                env.render_device()

        def resets(k):
            for _ in range(k):
                env.reset_device()

        B = env.algorithmic_bytes_per_env_step() * n
        res = {"envs": n}
        for name, fn in [("step", tape_steps(obs.data_ptr())), ("step_noobs", tape_steps(None)), ("rollout_step", rollout),
                         ("render", renders), ("reset", resets)]:
            us = timeit(fn, args.iters)
            res[name + "_us"] = round(us, 2)
            if name == "step":
                res["Msteps_s"] = round(n / us, 1)
                res["alg_GBs"] = round(B / us / 1e3, 1)
        print(json.dumps(res), flush=True)
        env.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Time the REFERENCE itself (its own Python files, loaded by path) and the C oracle side by side
in this container, same workload shape and action distribution as bench.py.
Runs only where /root/reference exists; prints one JSON line.  Not used on the GPU box."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as gg  # noqa: E402  (loads the reference with the gym stub)
from oracle import snake_oracle  # noqa: E402


def time_reference(rules, dim, n_snakes, n_fruits, steps, seed=0):
    vec = gg.VecHarness(rules, dim, n_snakes, n_fruits, seed, num_envs=1)
    vec.reset()
    rs = np.random.default_rng(1234)
    acts = rs.integers(0, 5, (steps, 1, n_snakes))
    t0 = time.perf_counter()
    for t in range(steps):
        vec.step(acts[t])
    return steps / (time.perf_counter() - t0)


def time_oracle(rules, dim, n_snakes, n_fruits, envs, steps, threads):
    ora = snake_oracle.Oracle(envs, dim=dim, n_snakes=n_snakes, n_fruits=n_fruits, rules=rules, seed=0)
    ora.reset()
    rs = np.random.default_rng(1234)
    acts = rs.integers(0, 5, (64, envs, n_snakes)).astype(np.int32)
    ora.step(acts[0], threads=threads)
    t0 = time.perf_counter()
    for t in range(steps):
        ora.step(acts[t % 64], threads=threads)
    return envs * steps / (time.perf_counter() - t0)


def main():
    cores = len(os.sched_getaffinity(0))
    out = {"host": "build container", "cores": cores,
           "reference_py_19x19x3_env_steps_per_s_1core": round(time_reference(0, 19, 3, 3, 4000), 1),
           "reference_py_new_world_10x10x1_env_steps_per_s_1core": round(time_reference(1, 10, 1, 1, 4000), 1),
           "oracle_c_19x19x3_env_steps_per_s_1core": round(time_oracle("snake_env", 19, 3, 3, 4096, 40, 1), 1),
           f"oracle_c_19x19x3_env_steps_per_s_{cores}core": round(time_oracle("snake_env", 19, 3, 3, 4096, 200, cores), 1)}
    out["oracle_vs_reference_1core"] = round(out["oracle_c_19x19x3_env_steps_per_s_1core"] /
                                            out["reference_py_19x19x3_env_steps_per_s_1core"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

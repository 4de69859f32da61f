#!/usr/bin/env python3
"""Plain vs streaming observation stores at 32 768 envs (130 MB of observations per launch: the plain-store range) with a
small and with a 96 MB action tape cycling through the Infinity Cache beside the launch: does the policy choice survive
the pollution?  (Round 3, one box: plain 23.1 / 23.8 us, streaming 26.9 / 26.6 us.)"""
import os, sys, ctypes, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, msnake
n, NS = 32768, 3
for T in (16, 256):
    for pol in ("plain", "stream"):
        env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, obs_store_policy=pol)
        env.reset_device()
        tape = torch.from_numpy(np.random.default_rng(1).integers(0, 5, (T, n, NS)).astype(np.int32)).cuda()
        L, h = env._L, env._h
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        def run(m):
            k = 0
            while k < m:
                c = min(T, m - k)
                msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, c, env._obs.data_ptr(), 0, env._rew.data_ptr(), env._done.data_ptr(), env._info.data_ptr(), 0, st))
                k += c
        run(64); torch.cuda.synchronize()
        us = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(256); e1.record(); torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / 256)
        print("tape steps", T, "(%d MB)" % (T * n * NS * 4 >> 20), pol, round(statistics.median(us), 2), flush=True)
        env.close(); del env, tape; torch.cuda.empty_cache()

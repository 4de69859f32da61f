#!/usr/bin/env python3
"""One kernel variant, a few hundred launches (run under rocprofv3 --pmc by tools/pmc_decompose.sh).
usage: pmc_variant.py {step|noobs|render|reset} n_snakes [envs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, msnake
what, ns = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=ns, seed=0, device="cuda:0")
env.reset_device()
tape = torch.randint(0, 5, (64, n, ns), dtype=torch.int32, device="cuda:0")
L, h = env._L, env._h
for rep in range(6):  # 64 warm-up steps advance the envs to their steady state, then 320 more
    if what in ("step", "noobs") or rep == 0:
        obs = env._obs.data_ptr() if (what != "noobs" or rep == 0) else None
        msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), ns, 64, obs, 0, env._rew.data_ptr(), env._done.data_ptr(),
                                              env._info.data_ptr(), 0, env._stream()))
    elif what == "render":
        for _ in range(64):
            env.render_device()
    elif what == "reset":
        for _ in range(64):
            env.reset_device()
torch.cuda.synchronize()

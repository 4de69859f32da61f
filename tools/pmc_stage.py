#!/usr/bin/env python3
"""Dynamic instruction count of the step kernel up to a debug stage exit (diagnostic build, MSNAKE_DBG_STAGES):
a normal handle reaches the steady state, its state is copied into a handle whose launches exit at stage
MSNAKE_DBG_STAGE (those launches never write state back, so every one sees the same steady state).
Run under rocprofv3 --pmc by tools/pmc_stages.sh.   usage: pmc_stage.py <stage> [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stage = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
import torch, msnake
os.environ.pop("MSNAKE_DBG_STAGE", None)
a = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=3, seed=0, device="cuda:0")
a.reset_device()
tape = torch.randint(0, 5, (64, n, 3), dtype=torch.int32, device="cuda:0")
a.rollout_device(tape, persistent=False, keep_obs=False)
blob = a.get_state_all()
os.environ["MSNAKE_DBG_STAGE"] = stage
b = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=3, seed=0, device="cuda:0")
b.reset_device()
b.set_state_all(blob)
for t in range(32):
    b.step_device(tape[t])
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Where does a self-play PPO update spend its time? (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
import msnake
from msnake import selfplay

def T():
    torch.cuda.synchronize(); return time.time()

env = msnake.MultiSnakeVecEnv(1024, dim=19, n_snakes=2, seed=0)
dev = env.device
model = selfplay.CnnPolicy((21, 21, 3)).to(dev)
opp = selfplay.CnnPolicy((21, 21, 3)).to(dev)
runner = selfplay.Runner(env, model, [opp], 32, 0.99, 0.95)
for rep in range(2):
    t0 = T(); out = runner.run(); t1 = T()
    print(f"rollout 32 steps x 1024 envs: {t1-t0:.2f}s", flush=True)
obs, returns, masks, actions, values, nlps, epinfos = out
t0 = T()
for _ in range(32):
    a, v, nlp, full = runner.multi_step()
t1 = T(); print(f"  32 x multi_step only: {t1-t0:.2f}s")
t0 = T()
for _ in range(32):
    env.step_device(full)
t1 = T(); print(f"  32 x env.step_device only: {t1-t0:.4f}s")
opt = torch.optim.Adam(model.parameters(), lr=2.5e-4, eps=1e-5)
for B in (1024, 4096):
    mb = torch.randperm(obs.shape[0], device=dev)[:B]
    for rep in range(3):
        t0 = T(); x = obs[mb]; t1 = T()
        logits, vpred = model(x); t2 = T()
        loss, parts = selfplay.ppo_loss(logits, vpred, actions[mb], returns[mb], values[mb], nlps[mb], 0.1); t3 = T()
        opt.zero_grad(); loss.backward(); t4 = T()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5); opt.step(); t5 = T()
        print(f"B={B} gather {t1-t0:.3f} fwd {t2-t1:.3f} loss {t3-t2:.3f} bwd {t4-t3:.3f} opt {t5-t4:.3f}", flush=True)
print("--- which part of the loss makes backward slow? (B=1024)")
mb = torch.randperm(obs.shape[0], device=dev)[:1024]
x = obs[mb]
def timed(name, lossfn):
    for rep in range(2):
        logits, vpred = model(x)
        loss = lossfn(logits, vpred)
        t0 = T(); opt.zero_grad(); loss.backward(); t1 = T()
    print(f"  {name}: bwd {t1-t0:.4f}s", flush=True)
timed("square-mean", lambda lg, v: lg.square().mean() + v.square().mean())
timed("cross-entropy only", lambda lg, v: selfplay.neglogp(lg, actions[mb]).mean())
timed("entropy only", lambda lg, v: selfplay.entropy(lg).mean())
timed("value only", lambda lg, v: ((v - returns[mb]) ** 2).mean())
timed("ppo_loss", lambda lg, v: selfplay.ppo_loss(lg, v, actions[mb], returns[mb], values[mb], nlps[mb], 0.1)[0])
x2 = torch.randint(0, 256, (1024, 21, 21, 3), dtype=torch.uint8, device=dev)
x = x2
timed("square-mean on random frames", lambda lg, v: lg.square().mean() + v.square().mean())

import sys, time, numpy as np, os
sys.path.insert(0,'/root/repo')
from oracle import snake_oracle as so
tp=np.random.default_rng(1).integers(0,5,(256,4096,3)).astype(np.int32)
print('cpu.max:', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a', 'affinity', len(os.sched_getaffinity(0)))
for th in (1,4,8,16,32,64,128):
    big=so.Oracle(4096, dim=19, n_snakes=3, seed=0); big.reset()
    big.rollout(tp[:8], th)
    t0=time.time(); reps=0
    while time.time()-t0 < 2.0:
        big.rollout(tp, th); reps+=1
    dt=time.time()-t0
    print(th,'threads:', round(4096*256*reps/dt/1e6,2), 'M env-steps/s', flush=True)

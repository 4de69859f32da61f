import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, msnake
views = len(sys.argv) > 1
env = msnake.MultiSnakeVecEnv(4096, dim=19, n_snakes=3, seed=0, host_views=views)
print("host_views =", views)
env.reset()
rs = np.random.default_rng(0)
acts = [list(map(tuple, rs.integers(0, 5, (4096, 3)))) for _ in range(4)]
arr = [rs.integers(0, 5, (4096, 3)).astype(np.int32) for _ in range(4)]
for name, A in (("list-of-tuples actions", acts), ("ndarray actions", arr)):
    for _ in range(3): env.step(A[0])
    t0 = time.time(); n = 30
    for i in range(n): obs, rew, done, infos = env.step(A[i % 4])
    dt = (time.time() - t0) / n
    print(f"VecEnv.step ({name}): {dt*1e3:.2f} ms/step -> {4096/dt/1e6:.2f} M env-steps/s (obs {obs.nbytes/1e6:.1f} MB to host per step)")
t0 = time.time()
for i in range(30): eps = [i.get('episode') for i in infos]
print(f"scan of 4096 lazy infos: {(time.time()-t0)/30*1e3:.2f} ms; infos.episodes(): ", end="")
t0 = time.time()
for i in range(30): eps = infos.episodes()
print(f"{(time.time()-t0)/30*1e3:.3f} ms")

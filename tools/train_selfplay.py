#!/usr/bin/env python3
"""Self-play PPO on the HIP env (BASELINE.json configs[4]: 4096 envs, 2 snakes, 19x19).
    python tools/train_selfplay.py --envs 4096 --snakes 2 --timesteps 2000000 --csv ppo.csv"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")  # no exhaustive convolution search on first use


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--snakes", type=int, default=2)
    ap.add_argument("--dim", type=int, default=19)
    ap.add_argument("--rules", default="snake_env")
    ap.add_argument("--nsteps", type=int, default=64)
    ap.add_argument("--timesteps", type=int, default=int(2e6))
    ap.add_argument("--scale", type=int, default=1, help="4 = 84x84 frames (nature_cnn) at dim 19")
    ap.add_argument("--bf16", action="store_true", help="policy trunk under bf16 autocast (reference: fp32)")
    ap.add_argument("--csv", default=None)
    ap.add_argument("--monitor", default=None)
    ap.add_argument("--json", default=None, help="progress.json (baselines JSON log format)")
    ap.add_argument("--tensorboard", default=None, help="directory for a TensorBoard events file")
    ap.add_argument("--save", default=None, metavar="DIR", help="checkpoint directory (the reference's saved_models/<expr>/)")
    ap.add_argument("--save-interval", type=int, default=0, help="also save snake_model_num<N>_<k>.pt every this many updates")
    ap.add_argument("--load", default=None, metavar="FILE", help="initialise learner and opponents from a weights file")
    ap.add_argument("--resume", action="store_true", help="continue from <--save DIR>/trainer_state.pt")
    args = ap.parse_args()
    import torch
    import msnake
    from msnake import selfplay

    env = msnake.MultiSnakeVecEnv(args.envs, dim=args.dim, n_snakes=args.snakes, rules=args.rules, seed=0,
                                  obs_scale=args.scale)
    selfplay.learn(env, nsteps=args.nsteps, total_timesteps=args.timesteps, csv_path=args.csv,
                   monitor_path=args.monitor, json_path=args.json, tb_dir=args.tensorboard,
                   amp_dtype=torch.bfloat16 if args.bf16 else None, save_dir=args.save, save_interval=args.save_interval,
                   load_path=args.load, resume=args.resume)
    print(env.stats())
    env.close()


if __name__ == "__main__":
    main()

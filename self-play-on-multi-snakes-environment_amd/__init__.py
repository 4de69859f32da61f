"""msnake -- MI355X-native batched multi-snake environment step behind the reference's VecEnv API.

The directory name follows the repository convention (`self-play-on-multi-snakes-environment_amd`);
import it as `msnake` (the top-level msnake.py shim loads this package under that name).
Only the hot path lives here: csrc/ (HIP kernels + C-ABI -> libmsnake.so), the ctypes binding and
the VecEnv-compatible host class. Anything under oracle/ is test infrastructure and is never
imported from here.
"""
from . import _capi
from .spaces import Box, Discrete
from .dist import gather_stats, make_sharded, shard_range
from .vec_env import GYM_IDS, LazyInfos, MultiSnakeVecEnv, make, normalize_actions

__all__ = ["MultiSnakeVecEnv", "make", "GYM_IDS", "LazyInfos", "normalize_actions", "Box", "Discrete", "_capi",
           "shard_range", "gather_stats", "make_sharded"]
# msnake.selfplay (PyTorch self-play PPO driver) is imported on demand: `from msnake import selfplay`

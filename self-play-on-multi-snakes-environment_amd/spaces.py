"""Minimal gym.spaces stand-ins, used only when `gym` itself is not importable.

ppo_multi_agent.py needs `action_space` to satisfy isinstance(., gym.spaces.Discrete) with .n == 5
(baselines/common/distributions.py:243) and `observation_space.shape` (utils.py:51-55); when gym is
installed the real classes are used so that check passes unchanged.
"""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box, Discrete  # type: ignore
except Exception:  # noqa: BLE001

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"

        def __eq__(self, other):
            return isinstance(other, Discrete) and other.n == self.n

    class Box:
        def __init__(self, low, high, shape, dtype=np.uint8):
            self.low, self.high = low, high
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)

        def sample(self):
            return np.random.randint(self.low, self.high + 1, self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

        def __eq__(self, other):
            return isinstance(other, Box) and other.shape == self.shape and other.dtype == self.dtype

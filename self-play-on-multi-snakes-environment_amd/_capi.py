"""ctypes binding of include/msnake.h (libmsnake.so: HIP kernels + C-ABI, gfx950).

There is deliberately no fallback: if the shared library has not been built, or no MI355X is
visible, the env cannot be created and says so.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSNAKE_LIB: another build of the same library (kernel A/B runs on one box); never a different backend
LIB_PATH = os.environ.get("MSNAKE_LIB") or os.path.join(_HERE, "libmsnake.so")

RULES = {"snake_env": 0, "new_world": 1, "adversarial": 2}
RULE_NAMES = {v: k for k, v in RULES.items()}

# every extern "C" symbol include/msnake.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "msnake_abi_version", "msnake_last_error", "msnake_create", "msnake_destroy", "msnake_obs_shape",
    "msnake_reset", "msnake_step", "msnake_step_tape", "msnake_rollout_tape", "msnake_get_state", "msnake_set_state",
    "msnake_get_state_all", "msnake_set_state_all", "msnake_state_blob_info",
    "msnake_render", "msnake_get_stats", "msnake_kernel_name", "msnake_algorithmic_bytes_per_env_step",
]


class MsnakeConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("num_envs", ctypes.c_int32),
                ("dim", ctypes.c_int32), ("n_snakes", ctypes.c_int32), ("n_fruits", ctypes.c_int32),
                ("rules", ctypes.c_int32), ("max_steps", ctypes.c_int32), ("auto_reset", ctypes.c_int32),
                ("obs_scale", ctypes.c_int32), ("seed", ctypes.c_uint64), ("env_id_base", ctypes.c_uint64),
                # ABI 3: launch tuning, 0 = the library decides
                ("envs_per_block", ctypes.c_int32), ("record_policy", ctypes.c_int32),
                ("obs_store_policy", ctypes.c_int32), ("tape_store_policy", ctypes.c_int32)]


ABI_VERSION = 3
CONFIG_SIZE_V2 = 56
RECORD_POLICY = {"auto": 0, "full": 1, "short": 2}
STORE_POLICY = {"auto": 0, "plain": 1, "stream": 2}


class MsnakeBlobInfo(ctypes.Structure):
    _fields_ = [("version", ctypes.c_int32), ("num_envs", ctypes.c_int32), ("dim", ctypes.c_int32),
                ("n_snakes", ctypes.c_int32), ("n_fruits", ctypes.c_int32), ("rules", ctypes.c_int32),
                ("total_words", ctypes.c_int64)]


class MsnakeStats(ctypes.Structure):
    _fields_ = [("episodes", ctypes.c_int64), ("ep_len_sum", ctypes.c_int64), ("ep_return_sum", ctypes.c_int64),
                ("env_steps", ctypes.c_int64), ("errors", ctypes.c_int64), ("reserved", ctypes.c_int64 * 3)]


_lib = None


def load():
    """Load libmsnake.so, declare prototypes. Raises RuntimeError if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.basename(LIB_PATH).startswith("libmsnake"):
        raise RuntimeError(f"MSNAKE_LIB={LIB_PATH}: only another build of libmsnake*.so may be named (kernel A/B runs)")
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C <pkg>/csrc). "
            "There is no CPU fallback.")
    # torch first: its wheel bundles a HIP runtime with the SONAME libmsnake.so links against
    # (libamdhip64.so.7), and the device pointers / streams handed to the library come from it.
    # Loading libmsnake.so before torch would pull /opt/rocm's copy in as a SECOND runtime in the
    # process, and that one finds no device (msnake_create -> MSNAKE_E_NOGPU).
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, u8p = ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p
    L.msnake_abi_version.restype = ctypes.c_int
    L.msnake_last_error.restype = ctypes.c_char_p
    L.msnake_create.argtypes = [ctypes.POINTER(MsnakeConfig), ctypes.POINTER(vp)]
    L.msnake_destroy.argtypes = [vp]
    L.msnake_obs_shape.argtypes = [vp] + [ctypes.POINTER(i32)] * 3
    L.msnake_reset.argtypes = [vp, u8p, vp]
    L.msnake_render.argtypes = [vp, u8p, vp]
    L.msnake_step.argtypes = [vp, vp, i32, u8p, vp, vp, vp, vp]
    L.msnake_step_tape.argtypes = [vp, vp, i32, i32, u8p, ctypes.c_size_t, vp, vp, vp, ctypes.c_size_t, vp]
    L.msnake_rollout_tape.argtypes = L.msnake_step_tape.argtypes
    L.msnake_get_state.argtypes = [vp, i32, vp, i32]
    L.msnake_set_state.argtypes = [vp, i32, vp, i32]
    L.msnake_get_state_all.argtypes = [vp, vp, ctypes.c_size_t]
    L.msnake_get_state_all.restype = ctypes.c_int64
    L.msnake_set_state_all.argtypes = [vp, vp, ctypes.c_size_t]
    L.msnake_state_blob_info.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(MsnakeBlobInfo)]
    L.msnake_get_stats.argtypes = [vp, ctypes.POINTER(MsnakeStats), i32]
    L.msnake_kernel_name.argtypes = [vp]
    L.msnake_kernel_name.restype = ctypes.c_char_p
    L.msnake_algorithmic_bytes_per_env_step.argtypes = [vp]
    L.msnake_algorithmic_bytes_per_env_step.restype = ctypes.c_int64
    for name in ("msnake_create", "msnake_destroy", "msnake_obs_shape", "msnake_reset", "msnake_render",
                 "msnake_step", "msnake_step_tape", "msnake_rollout_tape", "msnake_get_state", "msnake_set_state", "msnake_set_state_all",
                 "msnake_state_blob_info", "msnake_get_stats"):
        getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def check(rc, what="msnake call"):
    """Mirror of the reference's error behaviour: failures surface as Python exceptions."""
    if rc < 0:
        msg = load().msnake_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
    return rc

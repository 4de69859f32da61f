"""Helpers to read tests/golden/ (written by tools/gen_golden.py from the running reference)."""
import json
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tape_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("tape_") and f.endswith(".npz"))


def load_tape(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, z


def unpack_state(z, prefix, t, e, rules):
    """Padded arrays -> canonical dict for (step t, env e). prefix 'st_' (per step) or 's0_'."""
    idx = (e,) if prefix == "s0_" else (t, e)
    lens = z[prefix + "len"][idx]
    body = z[prefix + "body"][idx]
    nf = int(z[prefix + "nfruits"][idx])
    st = {
        "snakes": [[[int(c[0]), int(c[1])] for c in body[s, :int(lens[s])]] for s in range(len(lens))],
        "fruits": [[int(c[0]), int(c[1])] for c in z[prefix + "fruits"][idx][:nf]],
        "vels": [[int(v[0]), int(v[1])] for v in z[prefix + "vels"][idx]],
        "grow_to": [int(g) for g in z[prefix + "grow"][idx]],
        "t": int(z[prefix + "t"][idx]),
        "ctr": int(z[prefix + "ctr"][idx]),
    }
    if rules == 1:
        st["alive"] = [bool(a) for a in z[prefix + "alive"][idx]]
        st["in_dead"] = [bool(a) for a in z[prefix + "in_dead"][idx]]
    if rules == 2:
        st["spare_fruits"] = int(z[prefix + "spare"][idx])
    return st


def state_view(st, rules):
    """Project a get_state() dict onto the fields the reference defines for this rule set."""
    keys = ["snakes", "fruits", "vels", "grow_to", "t", "ctr"]
    if rules == 1:
        keys += ["alive", "in_dead"]
    if rules == 2:
        keys += ["spare_fruits"]
    return {k: st[k] for k in keys}


def edge_cases():
    with open(os.path.join(GOLDEN, "edge_cases.json")) as f:
        return json.load(f)


def blob_to_obs(blob):
    return np.frombuffer(zlib.decompress(bytes.fromhex(blob["z"])), np.uint8).reshape(blob["shape"])


def crc_rows(obs):
    return np.array([zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in obs], np.uint32)

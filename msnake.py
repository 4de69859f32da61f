"""`import msnake` -> the package in ./self-play-on-multi-snakes-environment_amd/ (its directory
name is not a valid Python identifier, so it is loaded by path and registered under this name)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "self-play-on-multi-snakes-environment_amd")
_spec = importlib.util.spec_from_file_location("msnake", os.path.join(_PKG_DIR, "__init__.py"),
                                               submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["msnake"] = _mod
_spec.loader.exec_module(_mod)

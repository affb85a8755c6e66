"""Import alias: the package directory name (humanoid-navigation-using-mpc-ldcbf_amd) is not a
Python identifier, so ``import lipmpc`` loads it through importlib."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("humanoid-navigation-using-mpc-ldcbf_amd")
sys.modules[__name__] = _pkg

"""Importable alias for the package directory ``sibrar---single-branch-recommender_amd`` (its name is not a Python
identifier): ``import sibrar_amd`` loads that package and registers it under this name."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('sibrar---single-branch-recommender_amd')
sys.modules[__name__] = _pkg

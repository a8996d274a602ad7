"""Importable alias of the `raytrace-miniapp_amd` package directory (a hyphen is
not a Python identifier).  `import raytrace_miniapp_amd as rt` gives the very
same module objects as importlib.import_module("raytrace-miniapp_amd")."""
import importlib
import sys
from pathlib import Path

_root = str(Path(__file__).resolve().parent.parent)
if _root not in sys.path:
    sys.path.insert(0, _root)
_real = importlib.import_module("raytrace-miniapp_amd")
sys.modules[__name__] = _real

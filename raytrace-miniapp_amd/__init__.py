"""MI355X (gfx950) backend for the XRayTrace `CreateImage` ray-trace imaging path.

The package directory carries the name the project was given
(`raytrace-miniapp_amd`); `import raytrace_miniapp_amd` is an alias module for
code that needs a Python identifier.
"""
from . import cabi, datfile, problem  # noqa: F401
from .problem import Beam, Gain, Problem, Seed, SeedBeam, scale_problem  # noqa: F401

__all__ = ["cabi", "datfile", "problem", "Beam", "Gain", "Problem", "Seed", "SeedBeam",
           "scale_problem"]

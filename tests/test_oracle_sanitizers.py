"""The oracle's C restatement under AddressSanitizer + UBSan (CPU build only): a full
pass over a reduced ASE problem and a seeded one, plus the path tracer, in a child
process that preloads the sanitizer runtime."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

CHILD = r"""
import importlib, sys
sys.path.insert(0, %r)
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
from oracle.binding import Oracle
o = Oracle(%r)
for name, scale in (("ASE_small", 0.05), ("seed_small", 0.0005)):
    p = rt.scale_problem(rt.datfile.load(%r + "/tests/golden/" + name + ".dat.xz"), scale)
    out = o.image_loop(p, n_threads=2)
    rays = p.build_rays(np.arange(0, p.n_rays_total, 7, dtype=np.int64))
    o.probe(p, rays)
    o.calc_ray_path(p, rays)
    assert out["failure_code"] == 0 and np.isfinite(out["image"]).all()
print("sanitized run ok")
"""


def test_oracle_under_asan_ubsan():
    so = ROOT / "oracle" / "librt_oracle_asan.so"
    r = subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "librt_oracle_asan.so"], capture_output=True, text=True)
    if r.returncode != 0 or not so.exists():
        pytest.skip("sanitizer build not available: " + r.stderr[-200:])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not Path(libasan).exists():
        pytest.skip("libasan not found")
    env = {"LD_PRELOAD": libasan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1",
           "PATH": "/usr/bin:/bin"}
    code = CHILD % (str(ROOT), str(so), str(ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sanitized run ok" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr

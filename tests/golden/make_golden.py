#!/usr/bin/env python3
"""Generate the fixtures under tests/golden/ from the reference itself.

Run in the build container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Writes, for each shipped input (ASE_small, seed_small):
  <name>.dat.xz          the reference's own input file, xz-compressed DATA
                         (problem tables + the golden image/I_ang its harness
                         checks against, src/CreateImageHelpers.cpp:66-100)
  <name>_ref_cpu.npz     image / I_ang computed by the UNMODIFIED reference
                         `cpu` method (RayTrace::create_image ->
                         RayTraceImageCPULoop), compiled by oracle/Makefile
  <name>_ref_rays.npz    RayTrace::calc_ray outputs (Iv, exit ray, error) of
                         every STRIDE-th ray of the reference's own ray list
  <name>_ref_path.npz    RayTrace::calc_ray_path outputs (x, y, I per sub-segment
                         boundary) on a small sub-grid, step factors 0.5 and 0.25
Only data is stored -- no reference source text in any encoding.
"""
import hashlib
import importlib
import lzma
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
rt = importlib.import_module("raytrace-miniapp_amd")
from oracle.binding import Reference, build  # noqa: E402

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
STRIDE = {"ASE_small": 997, "seed_small": 19501}
SHA256 = {
    "ASE_small": "d3f7614e7f89554caa6b209b4d1f4ead32943eb5ec1131cc4f93be0cb018e4f8",
    "seed_small": "5bf6597176b291dffe9b2ca71f43f0a935388981c4855058fc8b005e92daff5d",
}


def main():
    build(ref=True)
    ref = Reference()
    for name in ("ASE_small", "seed_small"):
        src = REF / f"{name}.dat"
        raw = src.read_bytes()
        assert hashlib.sha256(raw).hexdigest() == SHA256[name], f"{name}: unexpected input file"
        (OUT / f"{name}.dat.xz").write_bytes(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))
        r = ref.create_image_file(src, "cpu")
        np.savez_compressed(OUT / f"{name}_ref_cpu.npz", image=r["image"], I_ang=r["I_ang"],
                            dims=np.array([r["dims"][k] for k in ("nx", "ny", "na", "nb", "nv")]))
        n = rt.datfile.load(src).n_rays_total // STRIDE[name]
        q = ref.calc_rays_file(src, STRIDE[name], n)
        np.savez_compressed(OUT / f"{name}_ref_rays.npz", stride=STRIDE[name], Iv=q["Iv"],
                            ray2=q["ray2"], rays=q["rays"], err=q["err"])
        # RayTrace::calc_ray_path (the debug tracer) on a small sub-grid, two step factors
        sub = {"ASE_small": ((20, 3, 2, 1), (5, 4, 6, 5)), "seed_small": ((40, 2, 10, 12), (4, 3, 5, 6))}[name]
        paths = {}
        for c in (0.5, 0.25):
            t = ref.calc_ray_path_file(src, sub[0], sub[1], c)
            paths[f"x_c{c}"], paths[f"y_c{c}"], paths[f"I_c{c}"] = t["x"], t["y"], t["I"]
            paths[f"nerr_c{c}"] = t["n_errors"]
        np.savez_compressed(OUT / f"{name}_ref_path.npz", i0=np.array(sub[0]), n=np.array(sub[1]), **paths)
        print(f"{name}: |image|={np.linalg.norm(r['image']):.12g} |I_ang|={np.linalg.norm(r['I_ang']):.12g} "
              f"ref cpu {r['seconds']:.2f}s, {n} probe rays")


if __name__ == "__main__":
    main()

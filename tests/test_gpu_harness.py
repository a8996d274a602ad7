"""Drop-in check at the reference's own boundary: the C++ adapter
RayTraceImageHipLoop (raytrace-miniapp_amd/host/RayTraceImageHip.cpp), compiled
against the reference's headers and driven by a CreateImage-style harness that
links the reference's structs, .dat unpacking, check_ans and CPU loop.  The
binary is built in the container where the reference tree exists
(`make -C oracle hipharness`) and travels with oracle/_ref/."""
import lzma
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
BIN = ROOT / "oracle" / "_ref" / "CreateImageHip"
pytestmark = pytest.mark.gpu


def need(exe: Path):
    """The binaries of oracle/_ref/ exist only where the reference tree was present at build time
    (__graft_entry__.build() records that in oracle/ref_build_stamp.json).  Built there but missing here is a
    FAILURE (the snapshot lost them, and these boundary tests would silently not run); never built is a skip."""
    if exe.exists():
        return
    import json
    import os
    stamp = ROOT / "oracle" / "ref_build_stamp.json"
    if stamp.exists() and json.loads(stamp.read_text()).get("reference_present"):
        pytest.fail(f"{exe.relative_to(ROOT)} is missing although build() ran with the reference tree present: the "
                    "boundary tests cannot run")
    if not stamp.exists() and os.environ.get("GRAFT_REPO_ROOT"):
        # a GPU box runs a snapshot of a tree that build() has been run on where the reference exists: neither the
        # binaries nor the (git-ignored) stamp arriving means the snapshot lost oracle/_ref/, not that it was never built
        pytest.fail(f"{exe.relative_to(ROOT)} and oracle/ref_build_stamp.json are both missing on a GPU box "
                    "(GRAFT_REPO_ROOT is set): the snapshot lost the reference binaries; the boundary tests cannot run")
    pytest.skip(f"{exe.relative_to(ROOT)} was not built (no reference tree where build() ran)")


@pytest.mark.parametrize("name", ["ASE_small", "seed_small"])
def test_reference_harness_with_hip_backend(tmp_path, name):
    need(BIN)
    dat = tmp_path / f"{name}.dat"
    dat.write_bytes(lzma.decompress((ROOT / "tests" / "golden" / f"{name}.dat.xz").read_bytes()))
    methods = "cpu,Hip,Hip-MultiGPU" if name == "ASE_small" else "Hip"
    r = subprocess.run([str(BIN), f"-methods={methods}", "-iterations=5", str(dat)],
                       capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    # the answers must be right; the reference's 10 % / 15 % timing gates (CreateImage.cpp:174-181) are
    # evaluated and counted by the harness as the reference counts them, but ms-scale GPU calls on a shared
    # box may trip them, so the test reads the two counts apart
    assert "correctness errors: 0," in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("timing gates: std dev / avg =") == len(methods.split(","))
    assert "ray-steps/s" in r.stdout and "%HBM-peak" in r.stdout
    row = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("Hip ")][0].split()
    assert len(row) == 9 and float(row[6]) > 1e8 and 0.0 < float(row[8]) < 100.0   # ray-steps/s, % of peak
    if "timing-gate errors: 0" in r.stdout:
        assert r.returncode == 0 and "All tests passed" in r.stdout
    if name == "ASE_small":
        assert r.stdout.count("two-sided rel-L2 vs cpu") == 2


def test_config3_through_the_references_own_scale_problem_and_cpu_loop(tmp_path):
    """BASELINE config 3 against the reference itself: `-scale=16` makes the harness apply the reference's own
    scale_problem (src/CreateImageHelpers.cpp:104-150) to ASE_small.dat -- 6,384,000 rays, the ASE_medium stand-in -- and
    run its own RayTraceImageCPULoop beside the HIP loop; the two-sided rel-L2 it prints must be within 1e-5 for image
    and I_ang.  (check_ans does not run at scale != 1, src/CreateImage.cpp:156: the file's golden arrays are for scale 1.)"""
    import re
    need(BIN)
    dat = tmp_path / "ASE_small.dat"
    dat.write_bytes(lzma.decompress((ROOT / "tests" / "golden" / "ASE_small.dat.xz").read_bytes()))
    r = subprocess.run([str(BIN), "-methods=cpu,Hip", "-iterations=1", "-scale=16", str(dat)],
                       capture_output=True, text=True, timeout=900)
    print(r.stdout, r.stderr)
    assert "correctness errors: 0," in r.stdout, r.stdout + r.stderr
    m = re.search(r"two-sided rel-L2 vs cpu: image (\S+)\s+I_ang (\S+)", r.stdout)
    assert m, r.stdout
    assert float(m.group(1)) <= 1e-5 and float(m.group(2)) <= 1e-5
    row = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("Hip ")][0].split()
    # ray-steps/s over rays/s = cell steps per ray of the 6,384,000-ray stand-in (75,601,675 ray-steps: SURVEY.md 8(d))
    assert abs(float(row[6]) / float(row[5]) - 75601675 / 6384000) < 1e-2


def _write_dat(path, name, mutate=None):
    import importlib
    rt = importlib.import_module("raytrace-miniapp_amd")
    p = rt.datfile.load(ROOT / "tests" / "golden" / f"{name}.dat.xz")
    if mutate is not None:
        p = mutate(p, rt)
    rt.datfile.save(path, p)
    return p


@pytest.mark.parametrize("name", ["ASE_small", "seed_small"])
def test_the_real_dispatcher_with_the_hip_arms(tmp_path, name):
    """RayTrace::create_image itself (src/RayTraceImage.cpp:227-434: limit and grid checks, ray list,
    string dispatch, failure abort) with the arms of INTEGRATION.md section 2 compiled in, driven by the
    reference's own CreateImage (run_tests, check_ans): `make -C oracle dispatcher` applies the documented
    edits to scratch copies at build time; only the binary travels."""
    exe = ROOT / "oracle" / "_ref" / "CreateImage"
    need(exe)
    dat = tmp_path / f"{name}.dat"
    dat.write_bytes(lzma.decompress((ROOT / "tests" / "golden" / f"{name}.dat.xz").read_bytes()))
    methods = "cpu,Hip,Hip-MultiGPU,auto" if name == "ASE_small" else "Hip,Hip-MultiGPU"
    r = subprocess.run([str(exe), f"-methods={methods}", "-iterations=3", str(dat)], capture_output=True, text=True, timeout=900)
    print(r.stdout, r.stderr)
    out = r.stdout
    # check_ans prints nothing when the answer passes and "Answers do not match" lines when it fails
    assert "do not match" not in out + r.stderr, out + r.stderr
    assert out.count("Running ") == len(methods.split(",")) + 1      # + "Running tests for <file>"
    for m in methods.split(","):
        assert any(ln.split()[:1] == [m] for ln in out.splitlines()), f"no timing row for {m}"
    # run_tests counts its timing gates as errors too; anything else that went wrong would have aborted
    gate_msgs = out.count("exceeded")
    assert ("All tests passed" in out) == (gate_msgs == 0)


def test_the_real_dispatcher_aborts_on_failing_rays(tmp_path):
    """create_image's failure path (src/RayTraceImage.cpp:426-430): NaNs in a lineshape table make rays
    fail with error -3; the hip arm hands the code back and create_image ends the process with RAY_ERROR."""
    import numpy as np
    exe = ROOT / "oracle" / "_ref" / "CreateImage"
    need(exe)

    def poison(p, rt):
        g = p.gain[2]
        gv = g.gv.copy()
        gv[::7] = np.nan
        p.gain = p.gain[:2] + [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, gv, g.Nv)]
        return p

    dat = tmp_path / "bad.dat"
    _write_dat(dat, "ASE_small", poison)
    r = subprocess.run([str(exe), "-methods=Hip", "-scale=0.5", str(dat)], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode != 0
    assert "Some rays failed" in (r.stdout + r.stderr)
    assert "NaN" in (r.stdout + r.stderr)


def test_the_real_dispatcher_rejects_what_create_image_rejects(tmp_path):
    """The checks in front of the dispatch (src/RayTraceImage.cpp:229-264) run for the hip arm like for any
    other: a non-uniform euv_beam grid is refused before any ray is traced."""
    exe = ROOT / "oracle" / "_ref" / "CreateImage"
    need(exe)

    def bend(p, rt):
        import copy
        b = copy.copy(p.beam)
        x = b.x.copy()
        x[3] += 0.3 * b.dx
        b.x = x
        p.beam = b
        return p

    dat = tmp_path / "bent.dat"
    _write_dat(dat, "ASE_small", bend)
    r = subprocess.run([str(exe), "-methods=Hip", str(dat)], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode != 0
    assert "uniform grid" in (r.stdout + r.stderr)

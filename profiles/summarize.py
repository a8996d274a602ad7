#!/usr/bin/env python3
"""Condense rocprofv3 CSV output of profiles/run_profile.sh into the small
summaries that are committed under profiles/ (the raw CSVs stay in gpurun_out/).

  profiles/<tag>_kernel_stats.csv   per-kernel calls / total / avg ns (kernel-trace --stats)
  profiles/<tag>_pmc.json           per-launch counter averages of the path's kernels, with the
                                    gfx950 FETCH_SIZE correction (x2, MI355X_MICROARCH.md "HBM")
  profiles/traffic_latest.json      {"kernels": {kernel: hbm_bytes_per_launch}} read by bench.py
"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

KERNELS = ("rt_fused_kernel", "rt_march_kernel", "rt_freq_kernel")


def find(root, pattern):
    hits = sorted(glob.glob(str(Path(root) / "**" / pattern), recursive=True))
    return hits[0] if hits else None


def which(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def timed_calls(root, sub, steps):
    """Average duration of the LAST `steps` launches of each kernel of the path in a kernel-trace pass: the launches
    bench.py times (its warm-up and clock-ramp launches come first)."""
    f = find(Path(root) / sub, "*kernel_trace.csv")
    if f is None or not steps:
        return {}
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = which(r.get("Kernel_Name", ""))
        if k:
            per[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for k, v in per.items():
        last = [d for _, d in sorted(v)[-steps:]]
        out[k] = {"calls": len(last), "avg_ns": sum(last) / len(last), "min_ns": min(last), "max_ns": max(last)}
    return out


def kernel_stats(root, out_csv, sub="trace"):
    f = find(Path(root) / sub, "*kernel_stats.csv")
    if f is None:
        return {}
    rows = list(csv.DictReader(open(f)))
    with open(out_csv, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    out = {}
    for r in rows:
        k = which(r.get("Name", ""))
        if k:
            out[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]),
                      "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "percent": float(r["Percentage"])}
    return out


def pmc(root, sub):
    f = find(Path(root) / sub, "*counter_collection.csv")
    out = {}
    if f is None:
        return out
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = which(r.get("Kernel_Name", ""))
        if not k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {m: r[m] for m in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                     "Grid_Size", "Workgroup_Size") if m in r}
    for k, d in acc.items():
        out[k] = {"avg": {c: sum(v) / len(v) for c, v in d.items()}, "launches": {c: len(v) for c, v in d.items()},
                  "meta": meta[k]}
    return out


def kernel_source_hash():
    """The same hash bench.py computes: sha1 over the device code (bench.py quotes these counters only while the
    kernel sources are the ones they were measured on)."""
    import hashlib

    h = hashlib.sha1()
    csrc = Path(__file__).resolve().parent.parent / "raytrace-miniapp_amd" / "csrc"
    for name in ("rt_device.h", "rt_freq.hip", "rt_fused.hip", "rt_march.hip", "rt_math.h", "rt_path.hip"):
        h.update(name.encode())
        h.update((csrc / name).read_bytes())
    return h.hexdigest()[:16]


def workload_traffic(root, here, tag, prefix, name):
    """kernel times and HBM bytes per step of a side workload (bench.py --workload <name>)"""
    t = kernel_stats(root, here / f"{tag}_{name}_kernel_stats.csv", f"{prefix}_trace")
    f, w = pmc(root, f"{prefix}_fetch"), pmc(root, f"{prefix}_write")
    if not t:
        return None
    rec = {"kernel_trace": t, "kernels": {}}
    tot = 0.0
    for k in KERNELS:
        fs = f.get(k, {}).get("avg", {}).get("FETCH_SIZE")
        ws = w.get(k, {}).get("avg", {}).get("WRITE_SIZE")
        if fs is None or ws is None:
            continue
        rd, wr = fs * 1024.0 * 2.0, ws * 1024.0
        rec["kernels"][k] = {"read_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr,
                             "hbm_bytes_per_launch": rd + wr}
        tot += rd + wr
    if rec["kernels"]:
        rec["hbm_bytes_per_step"] = tot
    sq = pmc(root, f"{prefix}_sq")  # instruction counts per launch: what share of the issue rate the workload's kernels use
    if sq:
        rec["pmc_sq"] = sq
    return rec


def main():
    root, tag = sys.argv[1], sys.argv[2]
    here = Path(__file__).resolve().parent
    out = {"tag": tag, "git_head": sys.argv[3] if len(sys.argv) > 3 else None, "kernel_source_hash": kernel_source_hash(),
           "kernels": list(KERNELS), "kernel_trace": kernel_stats(root, here / f"{tag}_kernel_stats.csv")}
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out["kernel_trace_timed_steps"] = timed_calls(root, "trace", steps)
    fetch, write, sq = pmc(root, "pmc_fetch"), pmc(root, "pmc_write"), pmc(root, "pmc_sq")
    out["pmc_fetch"], out["pmc_write"], out["pmc_sq"] = fetch, write, sq
    hbm = {}
    for k in KERNELS:
        fs = fetch.get(k, {}).get("avg", {}).get("FETCH_SIZE")
        ws = write.get(k, {}).get("avg", {}).get("WRITE_SIZE")
        if fs is None or ws is None:
            continue
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts half the
        # bytes of wide coalesced reads (MI355X_MICROARCH.md "HBM"): doubled here.  Narrow
        # (4-16 B/lane) gathers are uncalibrated, so the read side is an upper bound.
        rd, wr = fs * 1024.0 * 2.0, ws * 1024.0
        hbm[k] = {"read_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    out["hbm"] = hbm
    out["hbm_note"] = "FETCH_SIZE x1024 x2 (gfx950 correction), WRITE_SIZE x1024; separate --pmc passes"
    # second SQ pass: instruction mix, and the shader clock from GRBM_GUI_ACTIVE (summed over the 8 XCDs)
    sq2 = pmc(root, "pmc_sq2")
    out["pmc_sq2"] = sq2
    kt = out["kernel_trace"]
    clocks = []
    for k in KERNELS:
        gui = sq2.get(k, {}).get("avg", {}).get("GRBM_GUI_ACTIVE")
        if gui and k in kt:
            clocks.append(gui / 8.0 / (kt[k]["avg_ns"] * 1e-9))
    if clocks:
        out["shader_clock_hz"] = sum(clocks) / len(clocks)
    # the same step as two launches (RT_HIP_FUSED=2): kernel-trace pass and SQ pass, for the per-kernel figures
    t2 = kernel_stats(root, here / f"{tag}_two_kernel_kernel_stats.csv", "trace2")
    if t2:
        out["two_kernel"] = {"kernel_trace": t2, "pmc_sq": pmc(root, "pmc_sq_2k")}
    # BASELINE config 5 and the seeded half of config 3 (bench.py --workload config5 / seed_medium)
    for prefix, name in (("c5", "config5"), ("sm", "seed_medium")):
        rec = workload_traffic(root, here, tag, prefix, name)
        if rec:
            out[name] = rec
    (here / f"{tag}_pmc.json").write_text(json.dumps(out, indent=1))
    if hbm:
        (here / "traffic_latest.json").write_text(json.dumps(
            {"tag": tag, "git_head": out["git_head"], "kernel_source_hash": out["kernel_source_hash"],
             "kernels": {k: v["hbm_bytes_per_launch"] for k, v in hbm.items()}, "detail": hbm}, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""Experiment: how much do the march and the frequency kernel gain from running concurrently (two queues)?
Plan A runs only its march, plan B only its frequency pass over the records of an earlier full run; both
are launched back to back on two streams and the wall time is compared with the sequential sum."""
import importlib, os, sys, time
sys.path.insert(0, '.')
import torch
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = rt.scale_problem(base, 16.0)
if os.environ.get('SHARD'):   # e.g. SHARD=8: the rank-0 shard of an 8-rank run (its march has a long idle tail)
    mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
    p = mg.shard(p, 0, int(os.environ['SHARD']))
for thr in sys.argv[1:] or ["default"]:
    if thr != "default":
        os.environ["RT_HIP_MARCH_THREADS"] = thr
    full = be.Plan(p); full.set_ray_grid()
    A = be.Plan(p); A.set_ray_grid(); A.set_debug(1)
    B = be.Plan(p); B.set_ray_grid(); B.run(); B.fetch(want_image=False); B.set_debug(2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def t(fn, n=12):
        best = 1e9
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best
    def both():
        A.run(s1.cuda_stream); B.run(s2.cuda_stream)
    def both_rev():
        B.run(s2.cuda_stream); A.run(s1.cuda_stream)
    print("march threads/CU", thr)
    print("  full run (march then freq) ", round(t(lambda: full.run(s1.cuda_stream)), 3))
    print("  march only                 ", round(t(lambda: A.run(s1.cuda_stream)), 3))
    print("  freq only                  ", round(t(lambda: B.run(s2.cuda_stream)), 3))
    print("  march || freq (march first)", round(t(both), 3))
    print("  march || freq (freq first) ", round(t(both_rev), 3))
    for q in (full, A, B):
        q.close()

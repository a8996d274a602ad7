"""Seeded deposit: distinct image pixels per 64-ray tile and per 2 ... 16 consecutive tiles of seed_small (how much a
row cache kept across tiles could save)."""
import importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
p = rt.datfile.load('tests/golden/seed_small.dat.xz')
b = p.beam; sb = p.seed_beam
print("seed beam grid", len(sb.x), len(sb.y), len(sb.a), len(sb.b), "image", b.nx, b.ny, b.nv)
with be.Plan(p) as plan:
    plan.set_ray_grid().enable_probe().run(); plan.fetch(); pr = plan.fetch_probe()
r2 = pr["ray2"]; fl = pr["flags"]
x = r2["x"].astype(np.float64); y = np.abs(r2["y"].astype(np.float64))
ix = np.floor((x - (b.x[0] - 0.5 * b.dx)) / b.dx).astype(np.int64); iy = np.floor((y - (b.y[0] - 0.5 * b.dy)) / b.dy).astype(np.int64)
ok = (ix >= 0) & (ix < b.nx) & (iy >= 0) & (iy < b.ny) & ((fl & 1) == 0)
pix = np.where(ok, ix + iy * b.nx, -1)
n = len(pix) // 64 * 64
t = pix[:n].reshape(-1, 64)
def distinct(rows):
    return np.array([len(set(r[r >= 0])) for r in rows])
d1 = distinct(t[::37][:3000])
print("distinct pixels per tile: mean %.2f max %d" % (d1.mean(), d1.max()))
for B in (2, 4, 8, 16):
    tb = pix[:len(pix) // (64 * B) * 64 * B].reshape(-1, 64 * B)
    dB = distinct(tb[::37][:1500])
    print(f"per {B} consecutive tiles: distinct mean {dB.mean():.2f} max {dB.max()}  (vs {B} x {d1.mean():.2f} = {B * d1.mean():.1f} row flushes now)")

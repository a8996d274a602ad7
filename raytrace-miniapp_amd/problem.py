"""Host-side problem records of the create_image call and the host logic that
surrounds the back-end loop.

Mirrors (does not copy) the data model of the reference:
  Beam      <- EUV_beam_struct   (src/RayTraceStructures.h:26-52, fields the path reads)
  SeedBeam  <- seed_beam_struct  (src/RayTraceStructures.h:141-170, ray grid only)
  Gain      <- ray_gain_struct   (src/RayTraceStructures.h:218-228)
  Seed      <- ray_seed_struct   (src/RayTraceStructures.h:276-281)
  Problem   <- create_image_struct (src/RayTraceStructures.h:323-340)
and the host steps of RayTrace::create_image (src/RayTraceImage.cpp:227-330):
limit/grid checks, mode select, scale, ray-list construction.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .cabi import RAY_DTYPE

N_MAX = 20   # src/common/RayTraceImageHelper.h:29
K_MAX = 100  # src/common/RayTraceImageHelper.h:30 (reference limit; ours is runtime)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


@dataclass
class Beam:
    x: np.ndarray
    y: np.ndarray
    a: np.ndarray
    b: np.ndarray
    dv: np.ndarray
    dx: float
    dy: float
    da: float
    db: float
    dz: float
    v: Optional[np.ndarray] = None
    z: Optional[np.ndarray] = None
    extra: dict = field(default_factory=dict)  # carried-but-unused header fields

    def __post_init__(self):
        self.x, self.y, self.a, self.b, self.dv = map(_f64, (self.x, self.y, self.a, self.b, self.dv))

    nx = property(lambda s: int(s.x.shape[0]))
    ny = property(lambda s: int(s.y.shape[0]))
    na = property(lambda s: int(s.a.shape[0]))
    nb = property(lambda s: int(s.b.shape[0]))
    nv = property(lambda s: int(s.dv.shape[0]))


@dataclass
class SeedBeam:
    x: np.ndarray
    y: np.ndarray
    a: np.ndarray
    b: np.ndarray
    dx: float
    dy: float
    da: float
    db: float
    extra: dict = field(default_factory=dict)

    def __post_init__(self):
        self.x, self.y, self.a, self.b = map(_f64, (self.x, self.y, self.a, self.b))

    nx = property(lambda s: int(s.x.shape[0]))
    ny = property(lambda s: int(s.y.shape[0]))
    na = property(lambda s: int(s.a.shape[0]))
    nb = property(lambda s: int(s.b.shape[0]))


@dataclass
class Gain:
    x: np.ndarray   # [Nx]
    y: np.ndarray   # [Ny]
    n: np.ndarray   # [Ny*Nx] flat, ix fastest
    g0: np.ndarray  # [Ny*Nx]
    E0: Optional[np.ndarray]
    gv: np.ndarray  # [Ny*Nx*Nv] flat, k fastest
    Nv: int
    gv0: Optional[np.ndarray] = None  # carried, never read on the path

    def __post_init__(self):
        self.x, self.y, self.n = map(_f64, (self.x, self.y, self.n.reshape(-1)))
        self.g0 = _f32(self.g0.reshape(-1))
        self.E0 = None if self.E0 is None else _f32(self.E0.reshape(-1))
        self.gv = _f32(self.gv.reshape(-1))

    Nx = property(lambda s: int(s.x.shape[0]))
    Ny = property(lambda s: int(s.y.shape[0]))


@dataclass
class Seed:
    x: List[np.ndarray]  # 5 grids (x, y, a, b, v)
    f: List[np.ndarray]
    f0: float

    def __post_init__(self):
        self.x = [_f64(v) for v in self.x]
        self.f = [_f64(v) for v in self.f]


@dataclass
class Problem:
    beam: Beam
    gain: List[Gain]                 # N lengths; gain[0] only decides use_emis
    seed_beam: Optional[SeedBeam] = None
    seed: Optional[Seed] = None
    N_start: int = 0
    N_parallel: int = 1
    golden_image: Optional[np.ndarray] = None  # arrays embedded in the .dat, if any
    golden_I_ang: Optional[np.ndarray] = None
    label: str = ""

    N = property(lambda s: len(s.gain))

    # ---- RayTraceImage.cpp:283-299 -------------------------------------
    @property
    def method(self) -> int:
        return 2 if self.seed is not None else 1

    @property
    def scale(self) -> float:
        if self.seed is None:
            return 1.0
        sb, eb = self.seed_beam, self.beam
        return (sb.dx * sb.dy * sb.da * sb.db) / (eb.dx * eb.dy)

    @property
    def ray_grid(self):
        g = self.seed_beam if self.seed_beam is not None else self.beam
        return g.x, g.y, g.a, g.b

    @property
    def n_rays_total(self) -> int:
        gx, gy, ga, gb = self.ray_grid
        return len(gx) * len(gy) * len(ga) * len(gb)

    def ray_ids(self) -> np.ndarray:
        """Flat ids ijkm of the rays this call owns (RayTraceImage.cpp:300-313)."""
        nt = self.n_rays_total
        if self.N_parallel <= 0:
            raise ValueError("N_parallel must be >= 1 (the reference divides by it)")
        return np.arange(self.N_start, nt, self.N_parallel, dtype=np.int64)

    def build_rays(self, ids: Optional[np.ndarray] = None) -> np.ndarray:
        """The ray list of create_image (RayTraceImage.cpp:300-328): b fastest,
        then a, y, x; coordinates rounded to float."""
        gx, gy, ga, gb = self.ray_grid
        if ids is None:
            ids = self.ray_ids()
        nb, na, ny = len(gb), len(ga), len(gy)
        m = ids % nb
        k = (ids // nb) % na
        j = (ids // (na * nb)) % ny
        i = ids // (ny * na * nb)
        rays = np.empty(ids.shape[0], dtype=RAY_DTYPE)
        rays["x"] = gx[i].astype(np.float32)
        rays["y"] = gy[j].astype(np.float32)
        rays["a"] = ga[k].astype(np.float32)
        rays["b"] = gb[m].astype(np.float32)
        return rays

    # ---- RayTraceImage.cpp:220-264 -------------------------------------
    def validate(self, enforce_reference_limits: bool = False) -> None:
        """The argument checks create_image performs before tracing."""
        if self.N < 2:
            raise ValueError("need at least 2 lengths")
        if enforce_reference_limits:
            if self.N > N_MAX:
                raise ValueError("Exceeded maximum number of length segments")
            if self.beam.nv >= K_MAX:
                raise ValueError("Exceeded maximum number of frequencies")

        def uneven(g, d):
            return bool(np.any(np.abs(np.diff(g) - d) > 1e-12 * d))

        b = self.beam
        if uneven(b.x, b.dx) or uneven(b.y, b.dy) or uneven(b.a, b.da) or uneven(b.b, b.db):
            raise ValueError("Only uniform grid spacings are currently supported (euv_beam)")
        if self.seed_beam is not None:
            s = self.seed_beam
            if uneven(s.x, s.dx) or uneven(s.y, s.dy) or uneven(s.a, s.da) or uneven(s.b, s.db):
                raise ValueError("Only uniform grid spacings are currently supported (seed_beam)")
            if (b.y[0] >= 0.0) != (s.y[0] >= 0.0):
                raise ValueError("Negitive y positions in seed_beam or euv_beam, but not both")
        if (self.seed is None) != (self.seed_beam is None):
            raise ValueError("seed and seed_beam must be given together")
        for g in self.gain[1:]:
            if g.Nv != b.nv:
                raise ValueError("gain.Nv must equal euv_beam.nv")

    @property
    def use_emis(self) -> bool:
        # Helper.h:402
        return self.gain[0].E0 is not None and self.seed is None


# ---------------------------------------------------------------------------
# scale_problem (src/CreateImageHelpers.cpp:104-150): cell-centred regrid of
# the four beam axes by scale**0.25 over the same physical extents.
# ---------------------------------------------------------------------------
def _scale_axes(beam, s: float):
    out = copy.copy(beam)
    for ax, dn in (("x", "dx"), ("y", "dy"), ("a", "da"), ("b", "db")):
        g = getattr(beam, ax)
        d = getattr(beam, dn)
        lo, hi = g[0] - 0.5 * d, g[-1] + 0.5 * d
        n = int(len(g) * s)
        nd = (hi - lo) / n
        setattr(out, dn, nd)
        setattr(out, ax, _f64(lo + (0.5 + np.arange(n)) * nd))
    return out


def scale_problem(p: Problem, scale: float) -> Problem:
    s = scale ** 0.25
    q = copy.copy(p)
    q.beam = _scale_axes(p.beam, s)
    if p.seed_beam is not None:
        q.seed_beam = _scale_axes(p.seed_beam, s)
    q.golden_image = None
    q.golden_I_ang = None
    q.label = f"{p.label}x{scale:g}(scale_problem)"
    return q


def regrid_beam(p: Problem, nx=None, ny=None, na=None, nb=None, a_centre=None, b_centre=None) -> Problem:
    """Cell-centred regrid of chosen beam axes to explicit sizes over the same
    extents (the scale_beam formula applied per axis); na/nb = 1 with a centre
    value collapses the angular axis (BASELINE config 5)."""
    q = copy.copy(p)
    b = copy.copy(p.beam)
    for ax, dn, n in (("x", "dx", nx), ("y", "dy", ny), ("a", "da", na), ("b", "db", nb)):
        if n is None:
            continue
        g, d = getattr(p.beam, ax), getattr(p.beam, dn)
        lo, hi = g[0] - 0.5 * d, g[-1] + 0.5 * d
        nd = (hi - lo) / n
        setattr(b, dn, nd)
        setattr(b, ax, _f64(lo + (0.5 + np.arange(n)) * nd))
    if a_centre is not None:
        b.a = _f64([a_centre])
    if b_centre is not None:
        b.b = _f64([b_centre])
    q.beam = b
    q.golden_image = q.golden_I_ang = None
    return q


def regrid_seed_beam(p: Problem, nx=None, ny=None, na=None, nb=None) -> Problem:
    """The same cell-centred regrid for the seed beam (the ray grid of the seeded mode)."""
    q = copy.copy(p)
    s = copy.copy(p.seed_beam)
    for ax, dn, n in (("x", "dx", nx), ("y", "dy", ny), ("a", "da", na), ("b", "db", nb)):
        if n is None:
            continue
        g, d = getattr(p.seed_beam, ax), getattr(p.seed_beam, dn)
        lo, hi = g[0] - 0.5 * d, g[-1] + 0.5 * d
        nd = (hi - lo) / n
        setattr(s, dn, nd)
        setattr(s, ax, _f64(lo + (0.5 + np.arange(n)) * nd))
    q.seed_beam = s
    q.golden_image = q.golden_I_ang = None
    return q


def resample_frequency(p: Problem, nv: int) -> Problem:
    """Linear resampling of the frequency axis of every gain table (gv rows)
    and of dv, preserving sum(dv) -- the synthetic nv=512 workload of
    BASELINE config 5."""
    q = copy.copy(p)
    K = p.beam.nv
    src = np.arange(K, dtype=np.float64)
    dst = np.linspace(0.0, K - 1.0, nv)
    lo = np.clip(np.floor(dst).astype(np.int64), 0, K - 2)
    w = (dst - lo).astype(np.float32)
    b = copy.copy(p.beam)
    dv = np.interp(dst, src, p.beam.dv)
    b.dv = _f64(dv * (p.beam.dv.sum() / dv.sum()))
    q.beam = b
    gains = []
    for g in p.gain:
        rows = g.gv.reshape(-1, K)
        new = rows[:, lo] * (1.0 - w)[None, :] + rows[:, lo + 1] * w[None, :]
        gains.append(Gain(g.x, g.y, g.n, g.g0, g.E0, new.astype(np.float32), nv))
    q.gain = gains
    if p.seed is not None:
        f4 = np.interp(dst, src, p.seed.f[4])
        x4 = np.interp(dst, src, p.seed.x[4])
        q.seed = Seed(p.seed.x[:4] + [x4], p.seed.f[:4] + [f4], p.seed.f0)
    q.golden_image = q.golden_I_ang = None
    q.label = f"{p.label}-nv{nv}"
    return q


def shard_columns(p: Problem, rank: int, world: int) -> Problem:
    """Pixel-tile shard for one GPU.

    ASE: image columns i = rank, rank+world, ... of the euv_beam (every rank
    samples the whole x range, so the load is balanced by construction).  The
    tile's deposit grid is the tile's own columns, so its image is a compact
    [ny][nx_local][nv] tile that the assembly step interleaves.
    Seeded: source columns of the seed_beam are sharded the same way; the
    deposit grid stays the full euv_beam and the assembly is a sum-reduce.
    """
    if world == 1:
        return p
    q = copy.copy(p)
    if p.seed is None:
        b = copy.copy(p.beam)
        b.x = _f64(p.beam.x[rank::world])
        q.beam = b
    else:
        s = copy.copy(p.seed_beam)
        s.x = _f64(p.seed_beam.x[rank::world])
        q.seed_beam = s
    q.golden_image = q.golden_I_ang = None
    return q

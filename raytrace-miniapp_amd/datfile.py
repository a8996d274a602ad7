"""Reader / writer of the miniapp's `.dat` problem files.

File  = uint64 N_bytes + payload                      (src/CreateImage.cpp:35-38)
payload = create_image_struct::pack                   (src/RayTraceStructures.cpp:2159-2222)
  int N, N_start, N_parallel; double dz;
  uint32 len + euv_beam blob      (EUV_beam_struct::pack,  :441-511)
  uint32 len + seed_beam blob     (seed_beam_struct::pack, :1028-1140; len 0 => none)
  N x (uint32 len + gain blob)    (ray_gain_struct::pack,  :1987-2017)
  uint32 len + seed blob          (ray_seed_struct::pack,  :1393-1412; len 0 => none)
  bool + image[nx*ny*nv] doubles, bool + I_ang[na*nb] doubles   (golden outputs)
Blobs of euv_beam / seed_beam start with the 16-byte byte_array_header
(src/RayTraceStructures.h:470-483: id 237, sizeof(int), sizeof(double),
version, type, 2 unused, 40-bit length at bytes 7..11, 4 flag bytes).
Little-endian, unaligned.  Written from the format description in SURVEY.md
8(f-1); this is host-side plumbing, not part of the timed path.
"""
from __future__ import annotations

import lzma
import struct
from pathlib import Path

import numpy as np

from .problem import Beam, Gain, Problem, Seed, SeedBeam

HEADER_ID = 237


class _Cur:
    def __init__(self, buf: bytes, pos: int = 0):
        self.buf = buf
        self.pos = pos

    def take(self, fmt: str):
        v = struct.unpack_from("<" + fmt, self.buf, self.pos)
        self.pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def arr(self, dtype, n: int) -> np.ndarray:
        dt = np.dtype(dtype).newbyteorder("<")
        a = np.frombuffer(self.buf, dtype=dt, count=n, offset=self.pos).copy()
        self.pos += n * dt.itemsize
        return a

    def blob(self) -> bytes:
        n = self.take("I")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b


def _read_header(c: _Cur, want_type: int) -> dict:
    """byte_array_header; old headerless blobs (id != 237) are version 0."""
    if c.buf[c.pos] != HEADER_ID:
        return {"version": 0, "type": 0, "n_bytes": 0, "flags": (0, 0, 0, 0)}
    hid, s_int, s_dbl, version, typ = c.take("5B")
    c.take("2B")
    hi = c.take("B")
    lo = c.take("I")
    flags = c.take("4B")
    if s_int != 4 or s_dbl != 8:
        raise ValueError("byte array written with foreign int/double sizes")
    if version > 0 and typ != want_type:
        raise ValueError(f"byte array has type {typ}, expected {want_type}")
    return {"version": version, "type": typ, "n_bytes": hi * 2**32 + lo, "flags": flags}


def _unpack_euv_beam(b: bytes) -> Beam:
    c = _Cur(b)
    head = _read_header(c, 2)
    run_ASE, run_sat, run_refract = c.take("3?")
    nx, ny, nz, na, nb, nv, _old = c.take("7i")
    if min(nx, ny, nz, na, nb, nv) < 1:
        raise ValueError("euv_beam: non-positive dimension")
    R_scale, G_scale, lam, Nc, dx, dy, dz, da, db, v0 = c.take("10d")
    x, y, z = c.arr("f8", nx), c.arr("f8", ny), c.arr("f8", nz)
    a, bb = c.arr("f8", na), c.arr("f8", nb)
    v, dv = c.arr("f8", nv), c.arr("f8", nv)
    if head["version"] >= 2 and head["n_bytes"] not in (0, c.pos):
        raise ValueError("euv_beam: byte count does not match header")
    extra = dict(run_ASE=run_ASE, run_sat=run_sat, run_refract=run_refract, R_scale=R_scale,
                 G_scale=G_scale, **{"lambda": lam}, Nc=Nc, v0=v0, version=head["version"])
    return Beam(x, y, a, bb, dv, dx, dy, da, db, dz, v=v, z=z, extra=extra)


def _unpack_seed_beam(b: bytes) -> SeedBeam:
    c = _Cur(b)
    head = _read_header(c, 3)
    nx, ny, na, nb = c.take("4i")
    vals = c.take("18d")
    dx, dy, da, db = vals[:4]
    names = ["Wx", "Wy", "Wa", "Wb", "Wv", "Wt", "x0", "y0", "a0", "b0", "t0", "E", "target", "chirp"]
    extra = dict(zip(names, vals[4:]))
    x, y, a, bb = c.arr("f8", nx), c.arr("f8", ny), c.arr("f8", na), c.arr("f8", nb)
    if head["version"] < 2:
        raise ValueError("seed_beam: only the version>=2 layout is supported")
    n_shape = c.take("i")
    if n_shape > 0:
        # tau, use_transform and seed_beam_shape blobs: not touched by create_image
        # (out of scope) -- kept as opaque bytes.
        extra["tau"] = c.arr("f8", n_shape)
        extra["use_transform"] = c.arr("?", n_shape)
        shapes = []
        for _ in range(n_shape):
            n = c.take("i")
            shapes.append(c.buf[c.pos:c.pos + n])
            c.pos += n
        extra["seed_shape_blobs"] = shapes
    extra["compression"] = head["flags"][0]
    if head["n_bytes"] not in (0, c.pos):
        raise ValueError("seed_beam: byte count does not match header")
    return SeedBeam(x, y, a, bb, dx, dy, da, db, extra=extra)


def _unpack_gain(b: bytes) -> Gain:
    c = _Cur(b)
    Nx, Ny, Nv = c.take("3i")
    x, y = c.arr("f8", Nx), c.arr("f8", Ny)
    n = c.arr("f8", Nx * Ny)
    g0 = c.arr("f4", Nx * Ny)
    E0 = c.arr("f4", Nx * Ny)  # unpack always allocates E0 (RayTraceStructures.cpp:2038)
    gv = c.arr("f4", Nx * Ny * Nv)
    gv0 = c.arr("f4", Nx * Ny)
    if c.pos != len(b):
        raise ValueError("gain: trailing bytes")
    return Gain(x, y, n, g0, E0, gv, Nv, gv0=gv0)


def _unpack_seed(b: bytes) -> Seed:
    c = _Cur(b)
    dim = c.take("5i")
    xs, fs = [], []
    for d in dim:
        xs.append(c.arr("f8", d))
        fs.append(c.arr("f8", d))
    f0 = c.take("d")
    if c.pos != len(b):
        raise ValueError("seed: trailing bytes")
    return Seed(xs, fs, f0)


def loads(payload: bytes, label: str = "") -> Problem:
    c = _Cur(payload)
    N, N_start, N_parallel = c.take("3i")
    c.take("d")  # dz duplicate, ignored by unpack (RayTraceStructures.cpp:2238-2239)
    eb = c.blob()
    if not eb:
        raise ValueError("file has no euv_beam")
    beam = _unpack_euv_beam(eb)
    sb = c.blob()
    seed_beam = _unpack_seed_beam(sb) if sb else None
    gains = [_unpack_gain(c.blob()) for _ in range(N)]
    sd = c.blob()
    seed = _unpack_seed(sd) if sd else None
    img = ang = None
    if c.take("?"):
        img = c.arr("f8", beam.nx * beam.ny * beam.nv)
    if c.take("?"):
        ang = c.arr("f8", beam.na * beam.nb)
    if c.pos != len(payload):
        raise ValueError("create_image payload: trailing bytes")
    return Problem(beam, gains, seed_beam, seed, N_start, N_parallel, img, ang, label)


def read_bytes(path) -> bytes:
    """Raw file bytes; `.xz` fixtures are decompressed transparently."""
    path = Path(path)
    raw = path.read_bytes()
    if path.suffix == ".xz":
        raw = lzma.decompress(raw)
    return raw


def load(path) -> Problem:
    raw = read_bytes(path)
    (n,) = struct.unpack_from("<Q", raw, 0)
    if n != len(raw) - 8:
        raise ValueError("Failed to read desired count")  # fread2, CreateImageHelpers.cpp:35-42
    name = Path(path).name.replace(".xz", "").replace(".dat", "")
    return loads(raw[8:], label=name)


# --------------------------------------------------------------------------- writer
def _header(typ: int, n_bytes: int, flag0: int = 0) -> bytes:
    return struct.pack("<5B2BBI4B", HEADER_ID, 4, 8, 2, typ, 0, 0, n_bytes >> 32,
                       n_bytes & 0xFFFFFFFF, flag0, 0, 0, 0)


def _pack_euv_beam(b: Beam) -> bytes:
    e = b.extra
    z = b.z if b.z is not None else np.zeros(1)
    v = b.v if b.v is not None else np.zeros(b.nv)
    body = struct.pack("<3?", e.get("run_ASE", True), e.get("run_sat", True), e.get("run_refract", True))
    body += struct.pack("<7i", b.nx, b.ny, len(z), b.na, b.nb, b.nv, 0)
    body += struct.pack("<10d", e.get("R_scale", -1.0), e.get("G_scale", -1.0), e.get("lambda", 0.0),
                        e.get("Nc", 0.0), b.dx, b.dy, b.dz, b.da, b.db, e.get("v0", 0.0))
    for arr in (b.x, b.y, z, b.a, b.b, v, b.dv):
        body += np.asarray(arr, "<f8").tobytes()
    return _header(2, 16 + len(body)) + body


def _pack_seed_beam(s: SeedBeam) -> bytes:
    e = s.extra
    names = ["Wx", "Wy", "Wa", "Wb", "Wv", "Wt", "x0", "y0", "a0", "b0", "t0", "E", "target", "chirp"]
    body = struct.pack("<4i", s.nx, s.ny, s.na, s.nb)
    body += struct.pack("<18d", s.dx, s.dy, s.da, s.db, *[e.get(k, 0.0) for k in names])
    for arr in (s.x, s.y, s.a, s.b):
        body += np.asarray(arr, "<f8").tobytes()
    shapes = e.get("seed_shape_blobs", [])
    body += struct.pack("<i", len(shapes))
    if shapes:
        body += np.asarray(e["tau"], "<f8").tobytes() + np.asarray(e["use_transform"], "?").tobytes()
        for sh in shapes:
            body += struct.pack("<i", len(sh)) + sh
    return _header(3, 16 + len(body), e.get("compression", 0)) + body


def _pack_gain(g: Gain) -> bytes:
    npix = g.Nx * g.Ny
    E0 = g.E0 if g.E0 is not None else np.zeros(npix, "<f4")
    gv0 = g.gv0 if g.gv0 is not None else np.zeros(npix, "<f4")
    return (struct.pack("<3i", g.Nx, g.Ny, g.Nv) + g.x.tobytes() + g.y.tobytes() + g.n.tobytes()
            + g.g0.tobytes() + np.asarray(E0, "<f4").tobytes() + g.gv.tobytes()
            + np.asarray(gv0, "<f4").tobytes())


def _pack_seed(s: Seed) -> bytes:
    out = struct.pack("<5i", *[len(v) for v in s.x])
    for xi, fi in zip(s.x, s.f):
        out += xi.tobytes() + fi.tobytes()
    return out + struct.pack("<d", s.f0)


def dumps(p: Problem, image=None, I_ang=None) -> bytes:
    def blob(b: bytes) -> bytes:
        return struct.pack("<I", len(b)) + b

    out = struct.pack("<3id", p.N, p.N_start, p.N_parallel, p.beam.dz)
    out += blob(_pack_euv_beam(p.beam))
    out += blob(_pack_seed_beam(p.seed_beam) if p.seed_beam is not None else b"")
    for g in p.gain:
        out += blob(_pack_gain(g))
    out += blob(_pack_seed(p.seed) if p.seed is not None else b"")
    image = p.golden_image if image is None else image
    I_ang = p.golden_I_ang if I_ang is None else I_ang
    for arr in (image, I_ang):
        out += struct.pack("<?", arr is not None)
        if arr is not None:
            out += np.asarray(arr, "<f8").tobytes()
    return out


def save(path, p: Problem, image=None, I_ang=None) -> None:
    payload = dumps(p, image, I_ang)
    raw = struct.pack("<Q", len(payload)) + payload
    path = Path(path)
    if path.suffix == ".xz":
        raw = lzma.compress(raw, preset=9)
    path.write_bytes(raw)

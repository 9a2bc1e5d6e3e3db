#!/usr/bin/env python3
"""Convert the reference's sample particle files (data fixtures, MIT licensed) to plain float32 .npy
position arrays under tests/data/.  Build container only (reads /root/reference/data).

Formats handled (positions only):
  * legacy VTK, BINARY, `POINTS <n> float` stored big-endian (io/vtk_format.rs in the reference);
  * gzip-compressed BGEO v5 (io/bgeo_format.rs): big-endian header, per point 4 floats (x,y,z,w)
    followed by the declared point attributes.
"""
import gzip, os, struct, sys
import numpy as np

SRC = "/root/reference/data"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "data")

def read_vtk_points(path):
    raw = open(path, "rb").read()
    key = b"POINTS "
    at = raw.index(key)
    eol = raw.index(b"\n", at)
    parts = raw[at:eol].split()
    ascii_mode = b"\nASCII" in raw[:at]
    n = int(parts[1]); assert parts[2] in (b"float", b"double"), parts
    dt = ">f4" if parts[2] == b"float" else ">f8"  # doubles are cast to f32 like the reference's f32 pipeline does
    if ascii_mode:
        vals = raw[eol + 1:].split()[:3 * n]
        return np.array([float(v) for v in vals], dtype=np.float64).astype(np.float32).reshape(n, 3)
    data = np.frombuffer(raw, dtype=dt, count=3 * n, offset=eol + 1)
    return data.astype(np.float32).reshape(n, 3)

def read_bgeo_points(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:5] == b"BgeoV" and struct.unpack(">i", raw[5:9])[0] == 5
    n_points, n_prims, n_pg, n_prg, n_pattr, n_vattr, n_prattr, n_attr = struct.unpack(">8i", raw[9:41])
    off = 41
    psize = 4
    for _ in range(n_pattr):
        (ln,) = struct.unpack(">H", raw[off:off + 2]); off += 2 + ln
        (size,) = struct.unpack(">H", raw[off:off + 2]); off += 2
        (_ty,) = struct.unpack(">i", raw[off:off + 4]); off += 4
        off += 4 * size  # default value
        psize += size
    data = np.frombuffer(raw, dtype=">f4", count=psize * n_points, offset=off).reshape(n_points, psize)
    return data[:, :3].astype(np.float32)

FILES = {
    "double_dam_break_frame_26_4732_particles": ("double_dam_break_frame_26_4732_particles.vtk", read_vtk_points),
    "hilbert_46843_particles": ("hilbert_46843_particles.bgeo", read_bgeo_points),
    "cube_8_particles": ("cube_8_particles.vtk", read_vtk_points),
    "cube_2366_particles": ("cube_2366_particles.vtk", read_vtk_points),
    "free_particles_125_particles": ("free_particles_125_particles.vtk", read_vtk_points),
    "bunny_frame_14_7705_particles": ("bunny_frame_14_7705_particles.vtk", read_vtk_points),
    # data sets of the reference's test_full.rs
    "pentagonal_hexecontahedron_32286_particles": ("pentagonal_hexecontahedron_32286_particles.bgeo", read_bgeo_points),
    "hilbert2_7954_particles": ("hilbert2_7954_particles.vtk", read_vtk_points),
    "octocat_32614_particles": ("octocat_32614_particles.bgeo", read_bgeo_points),
    "sailors_knot_19539_particles": ("sailors_knot_19539_particles.vtk", read_vtk_points),
    "free_particles_1000_particles": ("free_particles_1000_particles.vtk", read_vtk_points),
}

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for name, (fn, reader) in FILES.items():
        p = reader(os.path.join(SRC, fn))
        assert np.isfinite(p).all()
        np.save(os.path.join(DST, name + ".npy"), p)
        print(name, p.shape, p.min(0), p.max(0))

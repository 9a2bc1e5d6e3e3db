"""File formats: splashsurf_amd.io against files written by the reference itself (tests/golden/io/,
produced with the reference's CLI: `convert --particles` and `reconstruct -o mesh.{vtk,ply,obj}`)."""
import os

import numpy as np
import pytest

import mesh_compare as MC
from conftest import load_golden, load_points

IO_DIR = os.path.join(os.path.dirname(__file__), "golden", "io")


def _g(name):
    return os.path.join(IO_DIR, name)


def test_particle_readers_agree_with_the_reference_files():
    from splashsurf_amd import io as IO
    ref = load_points("free_particles_125_particles.npy")
    for ext in ("json", "vtk", "bgeo"):  # the same particles written by the reference's CLI in three formats
        p = IO.particles_from_file(_g("free_particles_125_particles_out." + ext))
        assert p.dtype == np.float32 and np.array_equal(p.view(np.uint32), ref.view(np.uint32)), ext
    cube = load_points("cube_8_particles.npy")
    assert np.array_equal(IO.particles_from_file(_g("cube8.xyz")), cube)
    assert np.array_equal(IO.particles_from_file(_g("cube8.json")), cube)
    assert IO.particles_from_file(_g("cube8.json"), dtype=np.float64).dtype == np.float64
    with pytest.raises(ValueError):
        IO.particles_from_file(_g("mesh_plain.obj"))


def test_particle_writers_reproduce_the_reference_files_byte_for_byte(tmp_path):
    from splashsurf_amd import io as IO
    p = IO.particles_from_file(_g("free_particles_125_particles_out.bgeo"))
    for ext in ("vtk", "json"):
        out = str(tmp_path / ("p." + ext))
        IO.particles_to_file(p, out)
        assert open(out, "rb").read() == open(_g("free_particles_125_particles_out." + ext), "rb").read(), ext
    out = str(tmp_path / "p.xyz")
    IO.particles_to_file(p, out)
    assert np.array_equal(IO.particles_from_file(out), p)


@pytest.mark.parametrize("kind", ["plain", "attr"])
def test_mesh_formats_round_trip_and_match_reference_bytes(tmp_path, kind):
    """The reference wrote the same mesh as vtk, ply and obj: the readers agree on it, and each writer reproduces the
    reference's file byte for byte from what another format's reader returned."""
    from splashsurf_amd import io as IO
    mv, mp, mo = (IO.mesh_from_file(_g("mesh_%s.%s" % (kind, e))) for e in ("vtk", "ply", "obj"))
    assert mv.vertices.dtype == np.float32 and mv.triangles.dtype == np.uint64
    for m in (mp, mo):
        assert np.array_equal(m.vertices, mv.vertices) and np.array_equal(m.triangles, mv.triangles)
    assert list(mv.point_attributes) == list(mp.point_attributes) == (["wnn", "sw", "normals"] if kind == "attr" else [])
    for k in mv.point_attributes:
        assert np.array_equal(mv.point_attributes[k], mp.point_attributes[k])
    if kind == "attr":
        assert np.array_equal(mo.point_attributes["normals"], mv.point_attributes["normals"])
    for ext, src in (("vtk", mp), ("ply", mv), ("obj", mp)):
        out = str(tmp_path / ("m." + ext))
        IO.mesh_to_file(src, out)
        assert open(out, "rb").read() == open(_g("mesh_%s.%s" % (kind, ext)), "rb").read(), ext


def test_reference_cli_mesh_equals_library_goldens():
    """The mesh the reference's CLI wrote for cube_8 (global strategy) is the mesh of the `global_cube_8` golden."""
    from splashsurf_amd import io as IO
    m = IO.mesh_from_file(_g("mesh_plain.vtk"))
    g = load_golden("global_cube_8")
    cmp = MC.compare_geometric(g["vertices"], g["triangles"], m.vertices, m.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] == 0.0


def test_display_formatting_follows_rust():
    from splashsurf_amd.io import _fmt_display, _fmt_json
    assert [_fmt_display(np.float32(x)) for x in (1.0, 0.5, -0.0, 1e-7, 123456790.0, 0.1)] == ["1", "0.5", "-0", "0.0000001", "123456790", "0.1"]
    assert [_fmt_json(x) for x in (1.0, 1e-7, 1e16, 0.1, float(np.float32(0.1)))] == ["1.0", "1e-7", "1e16", "0.1", "0.10000000149011612"]


@pytest.mark.gpu
def test_gpu_file_to_file_matches_the_reference_cli(gpu_ctx, tmp_path):
    """particles file -> reconstruction on the GPU -> mesh file: equal to what the reference's CLI wrote for the same
    input and parameters (`reconstruct cube8.xyz -r 0.025 -l 2.0 -c 1.0 --subdomain-grid=off`)."""
    import splashsurf_amd as S
    from splashsurf_amd import io as IO
    pts = IO.particles_from_file(_g("cube8.xyz"))
    res = S.reconstruct_surface(pts, particle_radius=0.025, smoothing_length=2.0, cube_size=1.0, subdomain_grid=False, context=gpu_ctx)
    out = str(tmp_path / "mesh.ply")
    IO.mesh_to_file(res.mesh, out)
    mine, ref = IO.mesh_from_file(out), IO.mesh_from_file(_g("mesh_plain.ply"))
    g = res.grid
    cmp = MC.compare_geometric(ref.vertices, ref.triangles, mine.vertices, mine.triangles, g.aabb.min, g.cell_size, g.npoints_per_dim)
    assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] == 0.0, cmp


def test_reference_reader_unit_tests():
    """ply_format.rs:270-312 (cube.ply, cube_normals.ply written by Blender, ASCII) and obj_format.rs:167-192 (icosphere.obj)."""
    from splashsurf_amd import io as IO
    m = IO.mesh_from_file(_g("cube.ply"))
    assert m.vertices.shape == (24, 3) and m.triangles.shape == (12, 3)
    m = IO.mesh_from_file(_g("cube_normals.ply"))
    assert m.vertices.shape == (24, 3) and m.triangles.shape == (12, 3) and m.point_attributes["normals"].shape == (24, 3)
    m = IO.mesh_from_file(_g("icosphere.obj"))
    assert m.vertices.shape == (42, 3) and m.triangles.shape == (80, 3)
    assert MC.mesh_is_closed_manifold(m.triangles)


def test_bgeo_writer_reproduces_the_reference_content(tmp_path):
    """particles_to_bgeo (bgeo_format.rs:108-257): the binary gzips with flate2, this writer with zlib -- the deflate
    streams differ, the decompressed files are identical byte for byte."""
    import gzip
    from splashsurf_amd import io as IO
    p = IO.particles_from_file(_g("free_particles_125_particles_out.bgeo"))
    out = str(tmp_path / "p.bgeo")
    IO.particles_to_file(p, out)
    assert gzip.decompress(open(out, "rb").read()) == gzip.decompress(open(_g("free_particles_125_particles_out.bgeo"), "rb").read())
    assert np.array_equal(IO.particles_from_file(out), p)


def test_bgeo_attributes_match_the_reference_cli(oracle):
    """Point attributes of a BGEO file (splashsurf/src/io.rs:138-190, bgeo_format.rs:54-75, 332-350) on one of the
    reference's own data files: the values read here, interpolated to the vertices the reference's CLI produced for
    `reconstruct -a density -a velocity`, reproduce the attributes the CLI wrote (SPH summation-order tolerance)."""
    from splashsurf_amd import io as IO
    path = os.path.join(os.path.dirname(__file__), "data", "dam_break_frame_9_6859_particles.bgeo")
    pts = IO.particles_from_file(path)
    attrs = IO.particle_attributes_from_file(path, ["id", "density", "velocity"])
    assert pts.shape == (6859, 3) and attrs["id"].dtype == np.uint64 and attrs["density"].dtype == np.float32 and attrs["velocity"].shape == (6859, 3)
    assert np.array_equal(np.sort(attrs["id"]), np.arange(6859, dtype=np.uint64))  # a permutation of the particle ids
    with pytest.raises(ValueError):
        IO.particle_attributes_from_file(path, ["pressure"])
    with pytest.raises(ValueError):
        IO.particle_attributes_from_file(_g("cube8.xyz"), ["density"])
    g = np.load(_g("bgeo_attributes_reference.npz"))
    r, l = np.float32(0.025), np.float32(2.0)
    res = oracle.reconstruct_surface(pts, oracle.make_params_relative(0.025, 2.0, 1.0, iso_surface_threshold=0.6))
    assert res.vertices.shape[0] == int(g["n_vertices"])
    h = np.float32(2.0) * l * r
    mass = np.float32(4.0) * np.float32(np.pi / 3.0) * (r * r * r) * np.float32(1000.0)  # reconstruct.rs:1126-1129
    V = g["vertices"]
    dens = oracle.post_sph_interpolate(pts, res.particle_densities, mass, h, attrs["density"], V, True)
    vel = oracle.post_sph_interpolate(pts, res.particle_densities, mass, h, attrs["velocity"], V, True)
    assert np.max(np.abs(dens - g["density"]) / np.abs(g["density"])) < 2e-5
    assert np.max(np.abs(vel - g["velocity"])) < 2e-5 * max(1.0, float(np.abs(g["velocity"]).max()))


def test_vtu_particle_files_of_the_reference():
    """XML VTK (`.vtu`): the reference's own test files (vtk_format.rs:425-449 checks the particle counts 8 / 250 / 250):
    appended raw data with zlib-compressed blocks and 64-bit headers, and the base64-encoded uncompressed variant of the
    same 250 particles."""
    from splashsurf_amd import io as IO
    d = os.path.join(os.path.dirname(__file__), "data")
    cube = IO.particles_from_file(os.path.join(d, "cube_8_particles.vtu"))
    assert cube.shape == (8, 3) and cube.dtype == np.float32
    assert np.array_equal(cube, load_points("cube_8_particles.npy"))  # the same particles as the reference's legacy .vtk
    a = IO.particles_from_file(os.path.join(d, "fluid_250_particles.vtu"), dtype=np.float64)
    b = IO.particles_from_file(os.path.join(d, "fluid_encoded_250_particles.vtu"), dtype=np.float64)
    assert a.shape == (250, 3) and a.dtype == np.float64 and np.array_equal(a, b)
    names = ["velocity", "pressure", "density", "index"]
    xa = IO.particle_attributes_from_file(os.path.join(d, "fluid_250_particles.vtu"), names)
    xb = IO.particle_attributes_from_file(os.path.join(d, "fluid_encoded_250_particles.vtu"), names)
    for n in names:
        assert np.array_equal(xa[n], xb[n]), n
    assert xa["velocity"].shape == (250, 3) and np.array_equal(np.sort(xa["index"]), np.arange(1, 251))
    assert 520.0 < xa["density"].min() and xa["density"].max() < 1022.0  # RangeMin / RangeMax stated in the encoded file
    with pytest.raises(ValueError):
        IO.particle_attributes_from_file(os.path.join(d, "cube_8_particles.vtu"), ["temperature"])


@pytest.mark.parametrize("binary", [False, True])
def test_legacy_vtk_with_cell_attributes(tmp_path, binary):
    """A legacy VTK mesh that carries CELL_DATA arrays (one tuple per cell, as ParaView writes them) between the geometry and the
    point attributes: the cell arrays are skipped with the right length and the point attributes after them are still found."""
    import struct
    from splashsurf_amd import io as IO
    pts = np.arange(12, dtype=np.float32).reshape(4, 3)
    cells = [3, 0, 1, 2, 3, 1, 2, 3]
    cell_scalar = [7.0, 9.0]
    cell_vec = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    point_scalar = [0.5, 1.5, 2.5, 3.5]
    path = tmp_path / ("cells_%s.vtk" % ("bin" if binary else "ascii"))
    with open(path, "wb") as f:
        f.write(b"# vtk DataFile Version 4.2\ncell attributes\n" + (b"BINARY\n" if binary else b"ASCII\n") + b"DATASET UNSTRUCTURED_GRID\n")

        def block(header, values, fmt):
            f.write(header.encode() + b"\n")
            if binary:
                f.write(struct.pack(">%d%s" % (len(values), fmt), *values) + b"\n")
            else:
                f.write((" ".join(str(v) for v in values) + "\n").encode())
        block("POINTS 4 float", [float(v) for v in pts.ravel()], "f")
        block("CELLS 2 8", cells, "i")
        block("CELL_TYPES 2", [5, 5], "i")
        f.write(b"CELL_DATA 2\n")
        block("SCALARS quality float 1\nLOOKUP_TABLE default", cell_scalar, "f")
        block("VECTORS facing float", cell_vec, "f")
        f.write(b"POINT_DATA 4\n")
        block("SCALARS density float 1\nLOOKUP_TABLE default", point_scalar, "f")
    d = IO._read_vtk(str(path))
    assert np.array_equal(d["points"], pts)
    assert d["cells"][0] == 2 and list(d["cells"][1]) == cells
    assert list(d["point_data"]) == ["density"]
    assert np.allclose(d["point_data"]["density"], point_scalar)

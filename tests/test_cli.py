"""CLI shim with the reference binary's `reconstruct` flags (splashsurf/src/reconstruct.rs:36-380, 604-965)."""
import os

import numpy as np
import pytest

from splashsurf_amd import cli

HERE = os.path.dirname(os.path.abspath(__file__))


def _parse(*argv):
    return cli.build_parser().parse_args(["reconstruct", *argv])


def test_flag_spelling_and_defaults_follow_the_reference():
    a = _parse("in.vtk", "-r=0.025", "-l", "2.0", "-c=0.5")
    assert (a.particle_radius, a.smoothing_length, a.cube_size) == (0.025, 2.0, 0.5)
    # defaults: reconstruct.rs:60 (rest density), :69 (threshold), :160 (subdomain cubes), :255 (normalization)
    assert a.rest_density == 1000.0 and a.surface_threshold == 0.6 and a.subdomain_cubes == 64
    assert a.mesh_smoothing_weights_normalization == 13.0
    assert a.subdomain_grid is True and a.subdomain_grid_auto_disable is True and a.mt_particles is True and a.simd is True
    assert a.double_precision is False and a.normals is False and a.mesh_smoothing_weights is False and a.mesh_cleanup is None
    b = _parse("in.vtk", "-r", "0.01", "-l", "2", "-c", "1", "-d=on", "--normals=ON", "--sph-normals=on", "--mesh-smoothing-iters=25",
               "--mesh-smoothing-weights=on", "--mesh-cleanup=off", "--particle-aabb-min", "-1", "0", "0", "--particle-aabb-max", "1", "1", "1",
               "-a", "velocity", "--interpolate_attribute", "density", "-n", "4", "--mt-files=on", "-t", "0.55")
    assert b.double_precision and b.normals and b.sph_normals and b.mesh_smoothing_iters == 25 and b.mesh_smoothing_weights
    assert b.particle_aabb_min == [-1.0, 0.0, 0.0] and b.interpolate_attributes == ["velocity", "density"] and b.surface_threshold == 0.55
    with pytest.raises(SystemExit):
        _parse("in.vtk", "-r", "0.01", "-l", "2", "-c", "1", "--normals=maybe")


def test_parameter_conversion_mirrors_the_binary():
    a = _parse("in.vtk", "-r", "0.01", "-l", "2", "-c", "1", "--subdomain-grid-auto-disable=off", "--mesh-smoothing-weights=on")
    kw = cli.pipeline_kwargs(a)
    # reconstruct.rs:633-636: the binary hands `!flag` to GridDecompositionParameters::auto_disable
    assert kw["subdomain_grid_auto_disable"] is True
    assert cli.pipeline_kwargs(_parse("in.vtk", "-r", "0.01", "-l", "2", "-c", "1"))["subdomain_grid_auto_disable"] is False
    assert kw["mesh_smoothing_weights"] is True and kw["compute_normals"] is False and kw["mesh_smoothing_iters"] is None
    # radius-relative lengths go through unchanged (the pipeline scales them as reconstruct.rs:627-628 does)
    assert (kw["particle_radius"], kw["smoothing_length"], kw["cube_size"]) == (0.01, 2.0, 1.0)


def test_unprovided_stages_fail_loudly():
    base = ["in.vtk", "-r", "0.01", "-l", "2", "-c", "1"]
    for extra in (["--decimate-barnacles=on"], ["--generate-quads=on"]):
        with pytest.raises(cli.CliError):
            cli.pipeline_kwargs(_parse(*base, *extra))
    kw = cli.pipeline_kwargs(_parse(*base, "--check-mesh=on"))  # reconstruct.rs:660-666: the three checks together
    assert kw["check_mesh_closed"] and kw["check_mesh_manifold"] and kw["check_mesh_orientation"] and not kw["check_mesh_debug"]
    kw = cli.pipeline_kwargs(_parse(*base, "--check-mesh-closed=on"))
    assert kw["check_mesh_closed"] and not kw["check_mesh_manifold"] and not kw["check_mesh_orientation"]
    # mesh cleanup (a sequential host stage of the reference, not provided): the binary's default is "on" as soon as
    # --mesh-smoothing-iters is present and not 0 (reconstruct.rs:201-214), so such a command line must switch it off
    assert cli.pipeline_kwargs(_parse(*base))["mesh_cleanup"] is False
    assert cli.pipeline_kwargs(_parse(*base, "--mesh-smoothing-iters=0"))["mesh_cleanup"] is False
    assert cli.pipeline_kwargs(_parse(*base, "--mesh-smoothing-iters=5", "--mesh-cleanup=off"))["mesh_cleanup"] is False
    for extra in (["--mesh-cleanup=on"], ["--mesh-cleanup=on", "--mesh-cleanup-snap-dist", "0.5"], ["--mesh-smoothing-iters=5", "--mesh-cleanup=on"]):
        with pytest.raises(cli.CliError):  # explicitly requested: refused
            cli.pipeline_kwargs(_parse(*base, *extra))
    # the binary's implicit default (the README's recipe) would switch the cleanup on: refused, never skipped silently
    with pytest.raises(cli.CliError) as ei:
        cli.pipeline_kwargs(_parse(*base, "--mesh-smoothing-iters=5"))
    assert "--mesh-cleanup=off" in str(ei.value)
    assert cli.pipeline_kwargs(_parse(*base, "--keep-verts=on"))["keep_vertices"] is True
    with pytest.raises(cli.CliError):
        cli.pipeline_kwargs(_parse(*base, "--mesh-aabb-min", "1", "0", "0", "--mesh-aabb-max", "0", "1", "1"))


def test_output_names_and_sequences(tmp_path):
    d = tmp_path / "in"
    d.mkdir()
    for i in (1, 2, 10, 11):
        (d / ("frame_%d.xyz" % i)).write_bytes(b"")
    (d / "frame_x.xyz").write_bytes(b"")
    (d / "single.xyz").write_bytes(b"")
    base = ["-r", "0.01", "-l", "2", "-c", "1"]
    # single file: "<stem>_surface.vtk" in the working directory unless --output-dir is given (reconstruct.rs:930-945)
    assert cli.collect_paths(_parse(str(d / "single.xyz"), *base)) == [(str(d / "single.xyz"), "single_surface.vtk")]
    out = tmp_path / "o" / "deep"
    pairs = cli.collect_paths(_parse(str(d / "single.xyz"), *base, "--output-dir", str(out), "-o", "m.ply"))
    assert pairs == [(str(d / "single.xyz"), str(out / "m.ply"))] and out.is_dir()
    # sequences: natural order, index range inclusive, default pattern "<stem with surface_{}>.vtk" (reconstruct.rs:903-927, 783-842)
    seq = cli.collect_paths(_parse(str(d / "frame_{}.xyz"), *base, "-s", "2", "-e=10", "--output-dir", str(out)))
    assert seq == [(str(d / "frame_2.xyz"), str(out / "frame_surface_2.vtk")), (str(d / "frame_10.xyz"), str(out / "frame_surface_10.vtk"))]
    allf = cli.collect_paths(_parse(str(d / "frame_{}.xyz"), *base, "-o", str(out / "s{}.obj")))
    assert [os.path.basename(b) for _, b in allf] == ["s1.obj", "s2.obj", "s10.obj", "s11.obj"]
    with pytest.raises(cli.CliError):
        cli.collect_paths(_parse(str(d / "frame_{}.xyz"), *base, "-o", "fixed.vtk"))
    with pytest.raises(cli.CliError):
        cli.collect_paths(_parse(str(d / "frame_{}.xyz"), *base, "-s", "5", "-e", "2"))
    with pytest.raises(cli.CliError):
        cli.collect_paths(_parse(str(d / "missing.xyz"), *base))
    with pytest.raises(cli.CliError):
        cli.collect_paths(_parse(str(tmp_path / "nodir" / "a.xyz"), *base))


@pytest.mark.gpu
def test_cli_end_to_end_matches_the_library_call(tmp_path, capfd):
    import splashsurf_amd as S
    from splashsurf_amd import io
    p = np.load(os.path.join(HERE, "data", "double_dam_break_frame_26_4732_particles.npy")).astype(np.float32)
    src = tmp_path / "dam_7.xyz"
    io.particles_to_file(p, str(src))
    rc = cli.main(["reconstruct", str(tmp_path / "dam_{}.xyz"), "-r=0.025", "-l=2.0", "-c=1.1", "--output-dir", str(tmp_path / "out"),
                   "--normals=on", "--mesh-smoothing-iters=3", "--mesh-smoothing-weights=on", "--mesh-cleanup=off", "--output-raw-mesh=on"])
    assert rc == 0
    raw = io.mesh_from_file(str(tmp_path / "out" / "raw_dam_surface_7.vtk"))
    out = io.mesh_from_file(str(tmp_path / "out" / "dam_surface_7.vtk"))
    rec = S.reconstruct_surface(p, particle_radius=0.025, smoothing_length=2.0, cube_size=1.1, subdomain_grid_auto_disable=False)  # the binary's default (see the inversion above)
    assert np.array_equal(raw.vertices, rec.mesh.vertices) and np.array_equal(raw.triangles, rec.mesh.triangles)
    assert raw.vertices.shape == (33026, 3) and raw.triangles.shape == (66220, 3)
    assert out.vertices.shape == raw.vertices.shape and "normals" in out.point_attributes
    assert not np.array_equal(out.vertices, raw.vertices)  # smoothed
    # the binary's default recipe (smoothing switches the mesh cleanup on) is refused with exit status 1 and writes nothing; with the
    # explicit opt-out it runs; asking for the cleanup explicitly is refused
    assert cli.main(["reconstruct", str(src), "-r=0.025", "-l=2.0", "-c=1.1", "-o", str(tmp_path / "recipe.obj"), "--mesh-smoothing-iters=3", "--check-mesh=on"]) == 1
    assert "--mesh-cleanup=off" in capfd.readouterr().err and not (tmp_path / "recipe.obj").exists()
    assert cli.main(["reconstruct", str(src), "-r=0.025", "-l=2.0", "-c=1.1", "-o", str(tmp_path / "recipe.obj"), "--mesh-smoothing-iters=3", "--mesh-cleanup=off",
                     "--check-mesh=on"]) == 0
    assert (tmp_path / "recipe.obj").exists()
    assert cli.main(["reconstruct", str(src), "-r=0.025", "-l=2.0", "-c=1.1", "-o", str(tmp_path / "clean.obj"), "--mesh-smoothing-iters=3", "--mesh-cleanup=on"]) == 1
    assert not (tmp_path / "clean.obj").exists()
    # error path: exit code 1, nothing written
    assert cli.main(["reconstruct", str(src), "-r=0.025", "-l=2.0", "-c=1.1", "--decimate-barnacles=on"]) == 1


@pytest.mark.gpu
def test_cli_sequence_equals_frame_by_frame_calls(tmp_path):
    """A three-frame sequence through the overlapped frame loop writes, per frame, the file a single-file call writes."""
    from splashsurf_amd import io
    p = np.load(os.path.join(HERE, "data", "double_dam_break_frame_26_4732_particles.npy")).astype(np.float32)
    for k in (1, 2, 3):
        io.particles_to_file((p[: len(p) - 500 * k] + np.float32(0.01 * k)).astype(np.float32), str(tmp_path / ("dam_%d.xyz" % k)))
    common = ["-r=0.025", "-l=2.0", "-c=1.1", "--normals=on"]
    assert cli.main(["reconstruct", str(tmp_path / "dam_{}.xyz")] + common + ["--output-dir", str(tmp_path / "seq")]) == 0
    for k in (1, 2, 3):
        assert cli.main(["reconstruct", str(tmp_path / ("dam_%d.xyz" % k))] + common + ["-o", str(tmp_path / ("one_%d.vtk" % k))]) == 0
        a = io.mesh_from_file(str(tmp_path / "seq" / ("dam_surface_%d.vtk" % k)))
        b = io.mesh_from_file(str(tmp_path / ("one_%d.vtk" % k)))
        assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.triangles, b.triangles)
        assert np.array_equal(a.point_attributes["normals"], b.point_attributes["normals"])
    assert len({io.mesh_from_file(str(tmp_path / "seq" / ("dam_surface_%d.vtk" % k))).vertices.shape for k in (1, 2, 3)}) == 3


def test_sequence_frames_overlap_file_io_in_order(tmp_path, monkeypatch):
    """The frame loop of a sequence (reconstruct.rs:380-470) reads the next file and writes the previous mesh on host threads while the current frame is
    reconstructed: outputs are written in frame order, each from its own frame's data, and an error of a read or a write fails the command at that frame.
    (The device stage is replaced by a stub here; tests -m gpu run the real one.)"""
    import threading
    import time
    import types
    from splashsurf_amd import io, postprocessing
    for k in (3, 4, 5, 6):
        io.particles_to_file(np.full((k, 3), float(k), np.float32), str(tmp_path / ("f_%d.xyz" % k)))
    events, lock = [], threading.Lock()

    def note(kind, what):
        with lock:
            events.append((kind, what))

    real_read = cli.read_particles_with_attributes

    def slow_read(path, names, dtype):
        note("read", os.path.basename(path))
        time.sleep(0.05)
        return real_read(path, names, dtype)

    def stub_pipeline(particles, attributes_to_interpolate=None, **kw):
        note("reconstruct", int(particles.shape[0]))
        time.sleep(0.1)
        n = int(particles.shape[0])
        m = types.SimpleNamespace(mesh=types.SimpleNamespace(vertices=np.full((n, 3), n, np.float32), triangles=np.zeros((1, 3), np.uint64)), point_attributes={})
        return m, m

    def slow_write(data, path):
        note("write", (os.path.basename(path), int(data.vertices.shape[0])))
        time.sleep(0.05)
        if "surface_5" in path and os.environ.get("FAIL_WRITE_5"):
            raise OSError("disk full")
        open(path, "w").write("%d" % data.vertices.shape[0])

    monkeypatch.setattr(cli, "read_particles_with_attributes", slow_read)
    monkeypatch.setattr(postprocessing, "reconstruction_pipeline", stub_pipeline)
    monkeypatch.setattr(io, "mesh_to_file", slow_write)
    args = _parse(str(tmp_path / "f_{}.xyz"), "-r=0.025", "-l=2.0", "-c=1.0", "--output-dir", str(tmp_path / "out"))
    os.makedirs(tmp_path / "out")
    written = cli.run_reconstruct(args, log=lambda m: None)
    assert [os.path.basename(w) for w in written] == ["f_surface_%d.vtk" % k for k in (3, 4, 5, 6)]
    for k in (3, 4, 5, 6):
        assert open(tmp_path / "out" / ("f_surface_%d.vtk" % k)).read() == str(k)  # every file from its own frame
    assert [e[1] for e in events if e[0] == "reconstruct"] == [3, 4, 5, 6]
    assert [e[1][0] for e in events if e[0] == "write"] == ["f_surface_%d.vtk" % k for k in (3, 4, 5, 6)]
    # overlap: frame k + 1 was read before frame k's reconstruction ended, i.e. before frame k was written
    order = [e for e in events if e[0] in ("read", "write")]
    assert order.index(("read", "f_4.xyz")) < order.index(("write", ("f_surface_3.vtk", 3)))
    # a failing write fails the command (at the next frame's hand-over at the latest); earlier frames are complete
    monkeypatch.setenv("FAIL_WRITE_5", "1")
    for k in (3, 4, 5, 6):
        os.remove(tmp_path / "out" / ("f_surface_%d.vtk" % k))
    with pytest.raises(OSError):
        cli.run_reconstruct(args, log=lambda m: None)
    assert (tmp_path / "out" / "f_surface_4.vtk").exists() and not (tmp_path / "out" / "f_surface_5.vtk").exists()


def test_convert_subcommand(tmp_path):
    """splashsurf/src/convert.rs: particle and mesh files between formats, domain filter, overwrite guard -- host only."""
    from splashsurf_amd import io
    gold = os.path.join(HERE, "golden", "io")
    src = os.path.join(gold, "free_particles_125_particles_out.vtk")
    out = str(tmp_path / "p.json")
    assert cli.main(["convert", "--particles", src, "-o", out]) == 0
    assert open(out, "rb").read() == open(os.path.join(gold, "free_particles_125_particles_out.json"), "rb").read()  # what the reference's convert wrote
    assert cli.main(["convert", "--particles", src, "-o", out]) == 1          # exists, no --overwrite
    p_all = io.particles_from_file(src)
    lo, hi = p_all.min(axis=0), np.median(p_all, axis=0)  # a box that keeps part of the particles; the upper bounds are exclusive
    assert cli.main(["convert", "--particles", src, "-o", out, "--overwrite", "--domain-min", *[repr(float(x)) for x in lo],
                     "--domain-max", *[repr(float(x)) for x in hi]]) == 0
    p_box = io.particles_from_file(out)
    keep = np.all(p_all >= lo, axis=1) & np.all(p_all < hi, axis=1)
    assert 0 < p_box.shape[0] < p_all.shape[0] and np.array_equal(p_box, p_all[keep])
    mesh_out = str(tmp_path / "m.ply")
    assert cli.main(["convert", "--mesh", os.path.join(gold, "mesh_plain.vtk"), "-o", mesh_out]) == 0
    assert open(mesh_out, "rb").read() == open(os.path.join(gold, "mesh_plain.ply"), "rb").read()
    assert cli.main(["convert", "-o", str(tmp_path / "x.vtk")]) == 1            # no input

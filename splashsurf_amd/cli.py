"""`python -m splashsurf_amd reconstruct ...` -- shim with the flags of the reference binary's `reconstruct` subcommand
(splashsurf/src/reconstruct.rs:36-380), routed to the file formats of `splashsurf_amd.io` and to
`postprocessing.reconstruction_pipeline` (reconstruction and post-processing stages on the MI355X).

The `reconstruct` subcommand and the small `convert` subcommand (splashsurf/src/convert.rs) exist.  Flags are spelled as in the reference (`--normals=on`, `-r 0.025`, ...).
Differences, all of them loud:
  * `--decimate-barnacles`, `--generate-quads` and `--mesh-cleanup` (the reference's sequential mesh stages) are not provided;
    switching one of them on is an error.  The binary switches the cleanup on by default as soon as `--mesh-smoothing-iters` is
    given and not 0 (reconstruct.rs:201-214): such a command line needs an explicit `--mesh-cleanup=off` here.  The
    `--check-mesh*` options run on the host; a finding fails the frame with the reference's message.
  * `--mt-files`, `--mt-particles`, `-n/--num-threads` and `--simd` are accepted and ignored: the work runs on the GPU (a sequence always overlaps
    the next frame's file read and the previous frame's file write with the current frame's reconstruction).
"""
import argparse
import os
import re
import sys

import numpy as np


def _switch(v):
    s = str(v).lstrip("=").lower()
    if s not in ("on", "off"):
        raise argparse.ArgumentTypeError("expected on|off")
    return s == "on"


def _opt_usize(v):
    return int(str(v).lstrip("="))


def _real(v):
    return float(str(v).lstrip("="))


def build_parser():
    ap = argparse.ArgumentParser(prog="splashsurf_amd", description="MI355X surface reconstruction (reference CLI flags)")
    sub = ap.add_subparsers(dest="command", required=True)
    p = sub.add_parser("reconstruct", help="reconstruct a surface from particle data (reconstruct.rs:36-380)")
    sw = dict(type=_switch, metavar="off|on")
    p.add_argument("input_file_or_sequence")
    p.add_argument("-o", "--output-file")
    p.add_argument("--output-dir")
    p.add_argument("-s", "--start-index", type=_opt_usize)
    p.add_argument("-e", "--end-index", type=_opt_usize)
    p.add_argument("-r", "--particle-radius", type=_real, required=True)
    p.add_argument("--rest-density", type=_real, default=1000.0)
    p.add_argument("-l", "--smoothing-length", type=_real, required=True)
    p.add_argument("-c", "--cube-size", type=_real, required=True)
    p.add_argument("-t", "--surface-threshold", type=_real, default=0.6)
    p.add_argument("-d", "--double-precision", default=False, **sw)
    p.add_argument("--particle-aabb-min", type=float, nargs=3, metavar=("X_MIN", "Y_MIN", "Z_MIN"))
    p.add_argument("--particle-aabb-max", type=float, nargs=3, metavar=("X_MAX", "Y_MAX", "Z_MAX"))
    p.add_argument("--mt-files", default=False, **sw)
    p.add_argument("--mt-particles", default=True, **sw)
    p.add_argument("-n", "--num-threads", type=_opt_usize)
    p.add_argument("--simd", default=True, **sw)
    p.add_argument("--subdomain-grid", default=True, **sw)
    p.add_argument("--subdomain-grid-auto-disable", default=True, **sw)
    p.add_argument("--subdomain-cubes", type=int, default=64)
    p.add_argument("--normals", default=False, **sw)
    p.add_argument("--sph-normals", default=False, **sw)
    p.add_argument("--normals-smoothing-iters", type=_opt_usize)
    p.add_argument("--output-raw-normals", default=False, **sw)
    p.add_argument("-a", "--interpolate_attribute", "--interpolate-attribute", dest="interpolate_attributes", action="append", default=[],
                   metavar="ATTRIBUTE_NAME")
    p.add_argument("--mesh-cleanup", default=None, **sw)
    p.add_argument("--mesh-cleanup-snap-dist", type=float)
    p.add_argument("--decimate-barnacles", default=False, **sw)
    p.add_argument("--keep-verts", default=False, **sw)
    p.add_argument("--mesh-smoothing-iters", type=_opt_usize)
    p.add_argument("--mesh-smoothing-weights", default=False, **sw)
    p.add_argument("--mesh-smoothing-weights-normalization", type=float, default=13.0)
    p.add_argument("--output-smoothing-weights", default=False, **sw)
    p.add_argument("--generate-quads", default=False, **sw)
    p.add_argument("--quad-max-edge-diag-ratio", type=float, default=1.75)
    p.add_argument("--quad-max-normal-angle", type=float, default=10.0)
    p.add_argument("--quad-max-interior-angle", type=float, default=135.0)
    p.add_argument("--mesh-aabb-min", type=float, nargs=3, metavar=("X_MIN", "Y_MIN", "Z_MIN"))
    p.add_argument("--mesh-aabb-max", type=float, nargs=3, metavar=("X_MAX", "Y_MAX", "Z_MAX"))
    p.add_argument("--mesh-aabb-clamp-verts", default=False, **sw)
    p.add_argument("--output-raw-mesh", default=False, **sw)
    for name in ("--check-mesh", "--check-mesh-closed", "--check-mesh-manifold", "--check-mesh-orientation", "--check-mesh-debug"):
        p.add_argument(name, default=False, **sw)
    cv = sub.add_parser("convert", help="convert particle or mesh files between the supported formats (convert.rs:13-47)")
    src = cv.add_mutually_exclusive_group()
    src.add_argument("--particles", dest="input_particles")
    src.add_argument("--mesh", dest="input_mesh")
    cv.add_argument("-o", dest="output_file", required=True)
    cv.add_argument("--overwrite", action="store_true")
    cv.add_argument("--domain-min", type=float, nargs=3, metavar=("X_MIN", "Y_MIN", "Z_MIN"))
    cv.add_argument("--domain-max", type=float, nargs=3, metavar=("X_MAX", "Y_MAX", "Z_MAX"))
    return ap


class CliError(Exception):
    pass


def _natural_key(name):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def collect_paths(args):
    """Input/output path pairs (reconstruct.rs:700-965): a `{}` in the input file name marks a sequence; the default
    output name is `<input stem>_surface.vtk` (`<stem with {} -> surface_{}>.vtk` for sequences), relative to
    `--output-dir` if given (created on demand)."""
    inp = args.input_file_or_sequence
    in_dir, in_name = os.path.split(inp)
    if not in_name:
        raise CliError('The input file path "%s" does not end with a filename' % inp)
    if in_dir and not os.path.isdir(in_dir):
        raise CliError('The parent directory "%s" of the input file path "%s" does not exist' % (in_dir, inp))
    stem = os.path.splitext(in_name)[0]
    is_sequence = "{}" in in_name
    if is_sequence:
        if args.output_file is not None:
            if "{}" not in args.output_file:
                raise CliError('The output filename "%s" does not contain a place holder "{}"' % args.output_file)
            out = args.output_file
        else:
            out = stem.replace("{}", "surface_{}") + ".vtk"
    else:
        if not os.path.isfile(inp):
            raise CliError('Input file does not exist: "%s"' % inp)
        out = args.output_file if args.output_file is not None else "%s_surface.vtk" % stem
    if args.start_index is not None and args.end_index is not None and args.start_index > args.end_index:
        raise CliError('Invalid input sequence range: "%d to %d"' % (args.start_index, args.end_index))
    if args.output_dir is not None:
        out = os.path.join(args.output_dir, out)
        out_parent = os.path.dirname(out)
        if out_parent and not os.path.exists(out_parent):
            os.makedirs(out_parent)
    if not is_sequence:
        return [(inp, out)]
    prefix, suffix = in_name.split("{}", 1)
    rx = re.compile(re.escape(prefix) + r"(\d+)" + re.escape(suffix))
    out_dir, out_pattern = os.path.split(out)
    pairs = []
    root = in_dir or "."
    for entry in sorted(os.listdir(root), key=_natural_key):
        m = rx.search(entry)
        if not m or not os.path.isfile(os.path.join(root, entry)):
            continue
        idx = int(m.group(1))
        if args.start_index is not None and idx < args.start_index:
            continue
        if args.end_index is not None and idx > args.end_index:
            continue
        pairs.append((os.path.join(in_dir, entry), os.path.join(out_dir, out_pattern.replace("{}", m.group(1)))))
    return pairs


def _aabb(lo, hi, what):
    if (lo is None) != (hi is None):
        raise CliError("both corners of the %s have to be given" % what)
    if lo is None:
        return None, None
    if any(a > b for a, b in zip(lo, hi)):  # reconstruct.rs try_aabb_from_min_max
        raise CliError("Failed to parse %s: a min coordinate is larger than the max coordinate" % what)
    return list(lo), list(hi)


MESH_CLEANUP_REFUSAL = ("--mesh-smoothing-iters switches the reference binary's mesh cleanup ON by default (marching_cubes_cleanup, reconstruct.rs:201-214), "
                        "and this build does not provide that stage: the command would produce a different mesh than the reference with a success exit "
                        "code.  Pass --mesh-cleanup=off to run the smoothing on the raw marching-cubes mesh (the reference does the same with that flag), "
                        "or run the reference's cleanup on the raw mesh (--output-raw-mesh=on)")


def pipeline_kwargs(args, warn=None):
    """reconstruct.rs:604-698 (ReconstructionRunnerArgs::try_from) in terms of `reconstruction_pipeline`'s keywords.
    `warn`: kept for callers of earlier versions (nothing is downgraded to a warning any more)."""
    unsupported = []
    if args.mesh_cleanup:  # explicitly requested: refused
        unsupported.append("--mesh-cleanup=on (not provided by this build; see INTEGRATION.md, \"Differences\")")
    elif args.mesh_cleanup is None and args.mesh_smoothing_iters not in (None, 0):
        # reconstruct.rs:201-214: the binary's default is "off" for 0 iterations and "on" as soon as smoothing is requested.  The reference
        # README's recipe (--mesh-smoothing-iters=25 ...) relies on that default: REFUSED here (exit status 1, nothing written) unless the
        # caller opts out of the cleanup explicitly -- a silent skip would be a non-parity mesh behind a success status.
        raise CliError(MESH_CLEANUP_REFUSAL)
    if args.decimate_barnacles:
        unsupported.append("--decimate-barnacles")
    if args.generate_quads:
        unsupported.append("--generate-quads")
    if unsupported:
        raise CliError("not provided by this build: " + ", ".join(unsupported))
    pmin, pmax = _aabb(args.particle_aabb_min, args.particle_aabb_max, "particle AABB")
    mmin, mmax = _aabb(args.mesh_aabb_min, args.mesh_aabb_max, "mesh AABB")
    return dict(
        particle_radius=args.particle_radius, rest_density=args.rest_density, smoothing_length=args.smoothing_length, cube_size=args.cube_size,
        iso_surface_threshold=args.surface_threshold, aabb_min=pmin, aabb_max=pmax, multi_threading=args.mt_particles, simd=args.simd,
        subdomain_grid=args.subdomain_grid,
        # reconstruct.rs:633-636 passes `auto_disable: !subdomain_grid_auto_disable`; kept as the binary behaves
        subdomain_grid_auto_disable=not args.subdomain_grid_auto_disable,
        subdomain_num_cubes_per_dim=args.subdomain_cubes, compute_normals=args.normals, sph_normals=args.sph_normals,
        normals_smoothing_iters=args.normals_smoothing_iters, mesh_smoothing_iters=args.mesh_smoothing_iters,
        mesh_smoothing_weights=args.mesh_smoothing_weights, mesh_smoothing_weights_normalization=args.mesh_smoothing_weights_normalization,
        output_mesh_smoothing_weights=args.output_smoothing_weights, output_raw_normals=args.output_raw_normals, output_raw_mesh=args.output_raw_mesh,
        mesh_aabb_min=mmin, mesh_aabb_max=mmax, mesh_aabb_clamp_vertices=args.mesh_aabb_clamp_verts, keep_vertices=args.keep_verts,
        mesh_cleanup=False,
        # reconstruct.rs:660-666: --check-mesh switches the three checks on together
        check_mesh_closed=args.check_mesh or args.check_mesh_closed, check_mesh_manifold=args.check_mesh or args.check_mesh_manifold,
        check_mesh_orientation=args.check_mesh or args.check_mesh_orientation, check_mesh_debug=args.check_mesh_debug)


def read_particles_with_attributes(path, names, dtype):
    """splashsurf/src/io.rs:68-190 read_particle_positions_with_attributes: attributes come from VTK point data or BGEO."""
    from . import io
    particles = io.particles_from_file(path, dtype=dtype)
    attrs = {}
    if names:
        try:
            attrs = io.particle_attributes_from_file(path, list(names))
        except ValueError as e:
            raise CliError(str(e))
    return particles, attrs


def run_reconstruct(args, log=None):
    from . import io, postprocessing
    log = log or (lambda m: print(m, file=sys.stderr))
    kwargs = pipeline_kwargs(args, warn=log)
    pairs = collect_paths(args)
    dtype = np.float64 if args.double_precision else np.float32
    written = []
    # A sequence is a chain per frame -- read the file, reconstruct on the GPU, write the mesh -- of which only the middle runs on the device: the next
    # frame's file is read and the previous frame's mesh written on two host threads while the current frame reconstructs (the reference overlaps frames with
    # --mt-files, reconstruct.rs:380-470; here the flag is accepted and the overlap is always on).  Files are written in order, one at a time; a failing read or
    # write surfaces at the frame it belongs to.
    from concurrent.futures import ThreadPoolExecutor
    reader, writer = ThreadPoolExecutor(max_workers=1), ThreadPoolExecutor(max_workers=1)
    try:
        pending_write = None
        ahead = reader.submit(read_particles_with_attributes, pairs[0][0], args.interpolate_attributes, dtype) if pairs else None
        for k, (src, dst) in enumerate(pairs):
            particles, attrs = ahead.result()
            ahead = reader.submit(read_particles_with_attributes, pairs[k + 1][0], args.interpolate_attributes, dtype) if k + 1 < len(pairs) else None
            log('Reconstructing "%s" (%d particles) -> "%s"' % (src, particles.shape[0], dst))
            mesh, rec = postprocessing.reconstruction_pipeline(particles, attributes_to_interpolate=attrs, **kwargs)
            jobs = []
            if args.output_raw_mesh:  # reconstruct.rs:1622-1654: raw_<output file name> next to the output file
                raw = os.path.join(os.path.dirname(dst), "raw_" + os.path.basename(dst))
                jobs.append((io.MeshWithData(rec.mesh.vertices, rec.mesh.triangles), raw))  # (owning copies: the next frame reuses nothing of them)
            jobs.append((io.MeshWithData(mesh.mesh.vertices, mesh.mesh.triangles, mesh.point_attributes), dst))
            if pending_write is not None:
                pending_write.result()  # (the previous frame's files are complete -- or its error is raised -- before this frame's are queued)

            def write_all(jobs=jobs):
                for data, path in jobs:
                    io.mesh_to_file(data, path)

            pending_write = writer.submit(write_all)
            written.extend(path for _, path in jobs)
        if pending_write is not None:
            pending_write.result()
    finally:
        reader.shutdown(wait=True)
        writer.shutdown(wait=True)
    return written


def run_convert(args):
    """splashsurf/src/convert.rs:49-147: f32 throughout, particles optionally filtered by a half-open box (aabb.rs:220-222)."""
    from . import io
    if not args.overwrite and os.path.exists(args.output_file):
        raise CliError('Aborting: Output file "%s" already exists. Use overwrite flag to ignore this.' % args.output_file)
    if (args.domain_min is None) != (args.domain_max is None):
        raise CliError("--domain-min and --domain-max have to be given together")
    try:
        if args.input_particles is not None:
            p = io.particles_from_file(args.input_particles, dtype=np.float32)
            if args.domain_min is not None:
                lo, hi = np.asarray(args.domain_min, np.float32), np.asarray(args.domain_max, np.float32)
                p = p[np.all(p >= lo, axis=1) & np.all(p < hi, axis=1)]
            io.particles_to_file(p, args.output_file)
        elif args.input_mesh is not None:
            io.mesh_to_file(io.mesh_from_file(args.input_mesh, dtype=np.float32), args.output_file)
        else:
            raise CliError("Aborting: No input file specified, either a particle or mesh input file has to be specified.")
    except (ValueError, NotImplementedError, OSError) as e:
        raise CliError(str(e))
    return [args.output_file]


def main(argv=None):
    args = build_parser().parse_args(argv)
    from .postprocessing import MeshCheckError
    try:
        if args.command == "convert":
            run_convert(args)
            return 0
        run_reconstruct(args)
    except (CliError, MeshCheckError) as e:
        print("error: %s" % e, file=sys.stderr)
        return 1
    return 0

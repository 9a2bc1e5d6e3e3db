"""splashsurf_amd -- MI355X-native drop-in for splashsurf_lib's subdomain-grid surface reconstruction.

The compute path lives in the C-ABI shared library `libsplashsurf_hip.so` (splashsurf_amd/csrc, built by
`__graft_entry__.build()`); this package is the thin Python host mirroring
`pysplashsurf.reconstruct_surface` (pysplashsurf/src/reconstruction.rs:135-207 of the reference) and, in
`splashsurf_amd.postprocessing`, its post-processing API (`reconstruction_pipeline`, `SphInterpolator`,
`laplacian_smoothing_parallel`, ...).
The library is loaded lazily, so importing the package (e.g. for `workloads`) works without a GPU.
"""
from .api import (  # noqa: F401
    Parameters,
    SurfaceReconstruction,
    FramePipeline,
    reconstruct_surface,
    reconstruct_surface_abs,
    grid_for_reconstruction,
    marching_cubes,
    neighborhood_search_spatial_hashing_parallel,
    Aabb3d,
    library_path,
    load_library,
)

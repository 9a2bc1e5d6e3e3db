"""Locate and import the reference's own pre-built wheel (build container only).

The wheel is vendored by the reference at
  /root/reference/splashsurf_studio/src/wheels/pysplashsurf-0.14.0.0-cp310-abi3-manylinux_2_17_x86_64.manylinux2014_x86_64.whl
It is extracted to a temp dir OUTSIDE the repo and imported from there; nothing of it is copied
into the repo.  Only tools/ (golden generation) uses this module; tests/bench never do.
"""
import os, sys, zipfile

WHEEL = ("/root/reference/splashsurf_studio/src/wheels/"
         "pysplashsurf-0.14.0.0-cp310-abi3-manylinux_2_17_x86_64.manylinux2014_x86_64.whl")
DEST = os.environ.get("SPLASH_ORACLE_DIR", "/tmp/oracle_whl")

if not os.path.exists(os.path.join(DEST, "pysplashsurf", "__init__.py")):
    if not os.path.exists(WHEEL):
        raise ImportError("reference wheel not available (this only works in the build container)")
    os.makedirs(DEST, exist_ok=True)
    zipfile.ZipFile(WHEEL).extractall(DEST)
if DEST not in sys.path:
    sys.path.insert(0, DEST)
import pysplashsurf  # noqa: E402,F401

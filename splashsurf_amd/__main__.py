"""`python -m splashsurf_amd reconstruct <input> -r <radius> -l <smoothing length> -c <cube size> [...]` (see cli.py)."""
import sys

from .cli import main

sys.exit(main())

"""Developer tools only: point the package at another BUILD of the library (tools/build_variant.sh) before it is loaded.

    import _devlib          # honours SYN3R_LIB_OVERRIDE=<path> for THIS tool run
    _devlib.use(path)       # or name the build explicitly

The product loader (syn3r_amd/_lib.py) reads no environment variable; the override is a tool-side, explicit call."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def use(path) -> None:
    from syn3r_amd import _lib
    _lib.set_library_path(path)


if os.environ.get("SYN3R_LIB_OVERRIDE"):
    use(os.environ["SYN3R_LIB_OVERRIDE"])
    print(f"[tools] library override: {os.environ['SYN3R_LIB_OVERRIDE']}", file=sys.stderr)

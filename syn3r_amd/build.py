"""Build libsyn3r_hip.so (gfx950 code objects) with hipcc, in-tree.

`python -m syn3r_amd.build` compiles every `csrc/*.hip` to an object and links
`syn3r_amd/lib/libsyn3r_hip.so`.  Objects are rebuilt only when their source or
a header is newer.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "libsyn3r_hip.so"
ARCH = "gfx950"

COMMON_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-fPIC",
    "-std=c++17",
    "-munsafe-fp-atomics",
    "-Wno-unused-result",
    "-fno-gpu-rdc",
]
# files whose arithmetic mirrors torch/numpy elementwise op order: no fma contraction
STRICT_FP = {"warp.hip", "sched.hip", "train.hip", "knn.hip"}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _newer(src: Path, dst: Path, headers: list[Path]) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return src.stat().st_mtime > t or any(h.stat().st_mtime > t for h in headers)


def build(verbose: bool = False, force: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))
    sources = sorted(CSRC.glob("*.hip"))
    if not sources:
        raise RuntimeError(f"no sources under {CSRC}")

    extra = os.environ.get("SYN3R_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DSYN3R_TIMING (developer builds)
    stamp = OBJ / "flags.txt"                                        # objects of another flag set are stale
    if (stamp.read_text() if stamp.exists() else "") != " ".join(extra):
        force = True
    stamp.write_text(" ".join(extra))

    def compile_one(src: Path) -> Path:
        obj = OBJ / (src.stem + ".o")
        if force or _newer(src, obj, headers):
            flags = list(COMMON_FLAGS) + extra
            if src.name in STRICT_FP:
                flags.append("-ffp-contract=off")
            cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr)
        return obj

    workers = max(1, min(6, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(workers) as ex:
        objs = list(ex.map(compile_one, sources))

    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    p = build(verbose=True, force="--force" in sys.argv)
    print(p)

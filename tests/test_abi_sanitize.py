"""Host side of the C-ABI under AddressSanitizer + UBSan: a CPU-only build of the library with the HOST
code instrumented (`-Xarch_host -fsanitize=address,undefined`; the device code is compiled as usual, GPU sanitizers are
not available on this pool) and a generated C driver that calls EVERY entry point of include/syn3r_hip.h with null
pointers / zero sizes, then with negative and absurd sizes: each call must come back with a status (or a size) - no
crash, no out-of-bounds access, no undefined behaviour in the argument checks, size computations and launch set-up that
run before the first kernel.  No GPU is needed (and none is used: on a GPU box the calls fail at their argument checks
just the same)."""
import ctypes
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and Path(c).exists():
            return c
    return None


def _ctype_name(t):
    from syn3r_amd import _lib as L
    if t is L.c_i: return "int"
    if t is L.c_f: return "float"
    if t is L.c_d: return "double"
    if t is L.c_sz: return "size_t"
    if t is L.c_ll: return "long long"
    if t is L.c_p: return "void*"
    if t is ctypes.c_char_p: return "const char*"
    if t is None: return "void"
    return "void*"      # POINTER(...)


def _driver_source():
    from syn3r_amd import _lib as L
    decl, calls = [], []
    for name, (res, args) in L.SIGNATURES.items():
        decl.append(f'extern "C" {_ctype_name(res)} {name}({", ".join(_ctype_name(a) for a in args) or "void"});')
        for variant in range(3):
            vals = []
            for a in args:
                c = _ctype_name(a)
                if c == "int": vals.append(["0", "-1", "2147483647"][variant])
                elif c == "long long": vals.append(["0", "-1", "4611686018427387904LL"][variant])
                elif c == "size_t": vals.append(["0", "1", "(size_t)-1"][variant])
                elif c == "float": vals.append(["0.0f", "-1.0f", "3.0e38f"][variant])
                elif c == "double": vals.append(["0.0", "-1.0", "1.0e300"][variant])
                elif c == "const char*": vals.append(["(const char*)0", "\"k_\"", "\"\""][variant])
                elif a not in (L.c_p,) and c == "void*": vals.append(["(void*)0", "(void*)scratch", "(void*)scratch"][variant])   # out-pointers
                else: vals.append(["(void*)0", "(void*)0", "(void*)scratch"][variant])
            if name == "syn3r_trace_report" and variant:
                vals = ["(const char*)0", "0"]      # its first argument is an OUTPUT buffer: only the (null, 0) form is safe to fabricate
            calls.append(f'    {name}({", ".join(vals)}); ++ncalls;')
    return ("#include <stddef.h>\n#include <stdio.h>\n" + "\n".join(decl) +
            "\nstatic char scratch[1 << 16] __attribute__((aligned(256)));\nint main() {\n    int ncalls = 0;\n" + "\n".join(calls) +
            '\n    const char* e = syn3r_last_error();\n    printf("abi driver: %d calls, last error: %s\\n", ncalls, e ? e : "(none)");\n    return 0;\n}\n')


@pytest.mark.timeout(900)
def test_host_side_of_the_abi_under_asan_ubsan(tmp_path):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not found")
    from syn3r_amd import build as B
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined"]
    flags = [f for f in B.COMMON_FLAGS if f != "-O3"] + ["-O1", "-g"]
    objs = []
    procs = []
    for src in sorted(B.CSRC.glob("*.hip")):
        obj = tmp_path / (src.stem + ".o")
        extra = ["-ffp-contract=off"] if src.name in B.STRICT_FP else []
        procs.append((src, subprocess.Popen([hipcc, *flags, *extra, *san, "-c", str(src), "-o", str(obj)], stdout=subprocess.PIPE,
                                            stderr=subprocess.PIPE, text=True)))
        objs.append(obj)
    for src, pr in procs:
        out, err = pr.communicate()
        assert pr.returncode == 0, f"{src.name}: {err[-2000:]}"
    lib = tmp_path / "libsyn3r_hip_san.so"
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-fsanitize=address,undefined", "-o", str(lib), *map(str, objs)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    drv = tmp_path / "driver.cpp"
    drv.write_text(_driver_source())
    exe = tmp_path / "driver"
    r = subprocess.run([hipcc, "-x", "c++", str(drv), "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-o", str(exe),
                        f"-L{tmp_path}", "-lsyn3r_hip_san", f"-Wl,-rpath,{tmp_path}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # no device for this process: the calls that pass their checks must stop at the launch, never run a kernel on the
    # driver's dummy buffers (matters only when the suite is run on a GPU box)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               HIP_VISIBLE_DEVICES="-1", CUDA_VISIBLE_DEVICES="-1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    assert "abi driver:" in r.stdout

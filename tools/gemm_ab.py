"""Per-shape comparison of the dense contraction kernels inside a real UNet unit (developer tool):
python tools/gemm_ab.py [VAR [A [B]]]  -> runs the unit with VAR=A and VAR=B ("-" = unset) in two subprocesses;
default: SYN3R_GEMM_WIDE=0 against unset.
NOTE (round 5): the library's dispatch switches (SYN3R_GEMM_Z, SYN3R_GEMM_WIDE, SYN3R_CONV_Z, SYN3R_Z_BAND, SYN3R_TCONV_ORDER, ...) are
compiled in only with -DSYN3R_TUNING: build the variant first (`tools/build_variant.sh tune -DSYN3R_TUNING`) and point
SYN3R_LIB_OVERRIDE at it; the shipped library ignores the environment."""
import os, re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]


VAR = sys.argv[1] if len(sys.argv) > 1 else "SYN3R_GEMM_WIDE"
VAL_A = sys.argv[2] if len(sys.argv) > 2 else "0"
VAL_B = sys.argv[3] if len(sys.argv) > 3 else "-"


def run(env_val):
    env = dict(os.environ)
    env.pop(VAR, None)
    if env_val != "-":
        env[VAR] = env_val
    out = subprocess.run([sys.executable, str(ROOT / "tools/unet_breakdown.py"), "14", "detail"], env=env,
                         capture_output=True, text=True).stdout
    d = {}
    for l in out.splitlines():
        m = re.match(r"(\S+)\[(\S+)\]\s+(\d+) launches\s+([\d.]+) ms", l)
        if m and m.group(1).startswith("k_gemm"):
            mm = re.search(r"<(\d)", m.group(1))
            mode = mm.group(1) if mm else "0"
            d["cdt"[int(mode)] + ":" + m.group(2)] = (m.group(1), int(m.group(3)), float(m.group(4)))
    tot = [l for l in out.splitlines() if l.startswith("total")]
    return d, tot[-1] if tot else ""


b, tb = run(VAL_A)
w, tw = run(VAL_B)
sb = sw = 0.0
for k in sorted(b, key=lambda k: -b[k][2]):
    if k in w:
        sb += b[k][2]; sw += w[k][2]
        print(f"{k:28s} n={b[k][1]:3d} base {b[k][2]:7.3f}  now {w[k][2]:7.3f} ({w[k][0][:11]})  {100 * (w[k][2] / b[k][2] - 1):+6.1f}%")
print("total", round(sb, 2), round(sw, 2), "|", tb, "|", tw)

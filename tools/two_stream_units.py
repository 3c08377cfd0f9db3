"""The two-stream step (pipeline/svd_2pass.py: _streamed_replace, _merged_post) against the one-stream order, same box:
python tools/two_stream_units.py [F]      (SYN3R_TWO_STREAMS=0 in a second process is the one-stream figure)"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import tuning; tuning.from_env()      # developer tool: host-graph switches from SYN3R_* variables
from syn3r_amd.pipeline.svd_step import SvdStepBench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device("cuda", 0)
b = SvdStepBench(F, dev)


def wall_ms(fn, n=3):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print(f"F={F} SYN3R_TWO_STREAMS={os.environ.get('SYN3R_TWO_STREAMS', '1')}: ms per (step, pass) unit")
print("replace  both passes of a step   %.1f" % (wall_ms(lambda: b.step_both("replace")) / 2))
print("post     both passes of a step   %.1f" % (wall_ms(lambda: b.step_both("post")) / 2))

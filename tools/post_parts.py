"""Developer tool: stacked against per-stream forms of a step's UNet calls at F = 25, INTERLEAVED in one process (the chip's clock
drifts between back-to-back measurements): the Replace step, the bare CFG forwards, the Post step."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.pipeline.svd_step import SvdStepBench
dev = torch.device("cuda", 0)
b = SvdStepBench(25, dev)
for v in ("replace", "replace", "post", "post"):
    b.step_both(v)
stR, stP = b._both_replace, b._both_post
pr, pp = stR["pipe"], stP["pipe"]
i, t = 5, b.sch.timesteps[5]
lat = (b.latents, b.latents.flip(dims=[1]))
x = torch.cat([pr._model_input(i, t, lat[k], stR["img4"][2 * k:2 * k + 2], True) for k in range(2)])
s = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
def once(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
def rep(two):
    pr.two_streams = two
    pr._streamed_replace(i, t, lat, stR["img4"], stR["ehs4"], stR["added4"], stR["ops2"], True)
def post(two):
    pp.two_streams = two
    pp._merged_post(i, t, lat, stP["img4"], stP["ehs4"], stP["added4"], stP["ops2"], True, stP["tile_ctx"])
def cfg(two):
    if not two:
        pr._unet(x, t, stR["ehs4"], stR["added4"], ctx_group=2)
        return
    cur = torch.cuda.current_stream(dev)
    for k in range(2):
        s[k].wait_stream(cur)
        with torch.cuda.stream(s[k]):
            pr._unet(x[2 * k:2 * k + 2], t, stR["ehs4"][2 * k:2 * k + 2], stR["added4"][2 * k:2 * k + 2])
    for k in range(2): cur.wait_stream(s[k])
for name, fn in (("replace step", rep), ("bare CFG forwards", cfg), ("post step", post)):
    fn(False); fn(True)
    a, c = [], []
    for _ in range(5):
        a.append(once(lambda: fn(False))); c.append(once(lambda: fn(True)))
    a.sort(); c.sort()
    print(f"{name:20s} stacked / one stream {a[2]:7.1f} ms (min {a[0]:.1f})   per-pass streams {c[2]:7.1f} ms (min {c[0]:.1f})   {100 * (c[2] / a[2] - 1):+.1f} %")

# where a Post step's time goes with and without the per-pass CFG streams
tiles, ov_y, ov_x, _ = stP["ops2"][0][3]
ehs_t, added_t, grp = stP["tile_ctx"]
sch = b.sch
def post_parts(two):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    xx = torch.cat([pp._model_input(i, t, lat[k], stP["img4"][2 * k:2 * k + 2], True) for k in range(2)])
    ev[0].record()
    for pair in ((0, 2), (1, 3)):
        xb = torch.cat([xx[2 * k:2 * k + 1, :, :, tiles[q][0], tiles[q][1]] for k in range(2) for q in pair], dim=0).contiguous()
        noise = pp._unet(xb, t, ehs_t, added_t, ctx_group=grp)
        for k in range(2):
            tops = stP["ops2"][k][3][3]
            for n, q in enumerate(pair):
                sch.step_interp(noise[2 * k + n:2 * k + n + 1], t, lat[k][0:1, :, :, tiles[q][0], tiles[q][1]].contiguous(), tops[q][0], tops[q][1],
                                stP["ops2"][k][2], step_i=i, lr=0.02, compute_grad=True)
    ev[1].record()
    if two:
        cur = torch.cuda.current_stream(dev)
        for k in range(2):
            s[k].wait_stream(cur)
            with torch.cuda.stream(s[k]):
                pp._unet(xx[2 * k:2 * k + 2], t, stP["ehs4"][2 * k:2 * k + 2], stP["added4"][2 * k:2 * k + 2])
        for k in range(2): cur.wait_stream(s[k])
    else:
        pp._unet(xx, t, stP["ehs4"], stP["added4"], ctx_group=2)
    ev[2].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
post_parts(False); post_parts(True)
for _ in range(3):
    a = post_parts(False); c = post_parts(True)
    print("post step: tiles %.1f ms + CFG stacked %.1f ms   |   tiles %.1f ms + CFG per-pass streams %.1f ms" % (a[0], a[1], c[0], c[1]))

# the pipeline's own _merged_post with events at its fork / join
marks = {}
orig_fork, orig_join = pp._fork, pp._join
def fork(key):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks["fork"] = e
    return orig_fork(key)
def join(streams):
    orig_join(streams)
    e = torch.cuda.Event(enable_timing=True); e.record(); marks["join"] = e
pp._fork, pp._join = fork, join
def post_marked(two):
    pp.two_streams = two
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    pp._merged_post(i, t, lat, stP["img4"], stP["ehs4"], stP["added4"], stP["ops2"], True, stP["tile_ctx"])
    e1.record(); torch.cuda.synchronize()
    if two:
        return e0.elapsed_time(marks["fork"]), marks["fork"].elapsed_time(marks["join"]), marks["join"].elapsed_time(e1)
    return e0.elapsed_time(marks["fork"]), marks["fork"].elapsed_time(e1), 0.0
for _ in range(3):
    print("pipeline post step, one stream: before fork %.1f, rest %.1f | streams: before fork %.1f, fork..join %.1f, after %.1f" % (post_marked(False)[:2] + post_marked(True)))
pp._side = s
for _ in range(2):
    print("with the script's streams: one stream: before fork %.1f, rest %.1f | streams: before fork %.1f, fork..join %.1f, after %.1f" % (post_marked(False)[:2] + post_marked(True)))
print("side stream priorities / ids:", [x.cuda_stream for x in s], [x.priority for x in s])
for _ in range(2):
    a = post_parts(False); c = post_parts(True)
    print("replica again: tiles %.1f + CFG stacked %.1f | tiles %.1f + CFG streams %.1f" % (a[0], a[1], c[0], c[1]))

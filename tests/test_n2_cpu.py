"""Point-cloud densification slice, host logic on the CPU: the key-frame / input-frame bookkeeping of
`DiffusionGS.densify_views` and the frame filter / pair graph / intrinsics of `densify_pcds` against what the REFERENCE's
own methods produced on the same seeded stand-ins (tests/golden/n2_bookkeeping.npz, oracle/gen_golden.py n2), and the
oracle of the cloud filter against brute force."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import golden_inputs as GI
from oracle import pcd_oracle as PO


class _Captured(Exception):
    pass


def _run_densify_views(name, tmp_path):
    from syn3r_amd import orchestrator as O
    from syn3r_amd.diffusionGS import DiffusionGS
    V, dtype, fps, nkey = GI.N2_CASES[name]
    vposes = GI.n2_view_poses(V)
    cap = {}

    def interp(i, j, replace=True, perturb_interp_poses=False):
        poses = list(O.pose_interpolation(vposes[i], vposes[j]))
        return [torch.full((3, 4, 6), GI.n2_frame_id(i, k), dtype=torch.float32) for k in range(25)], poses, None

    def densify_pcds(frames, poses, key_frame_mask=None, input_flags=None, win_samples=-1):
        cap.update(frames=np.array([float(f[0, 0, 0]) for f in frames], np.float32), poses=np.array(poses),
                   key_frame_mask=np.array(key_frame_mask), input_flags=np.array(input_flags), win_samples=win_samples)
        raise _Captured

    me = SimpleNamespace(num_input_views=V, save_dir=str(tmp_path), fps_keyframe_sampling=fps,
                         _interpolate_between_gs_v3=interp, densify_pcds=densify_pcds, device="cpu")
    with pytest.raises(_Captured):
        DiffusionGS.densify_views(me, 0, down_sample_rate=1, densify_type=dtype, num_views_for_pcd_densification=nkey)
    return cap


@pytest.mark.parametrize("name", list(GI.N2_CASES))
def test_key_frame_bookkeeping_matches_reference(name, golden_dir, tmp_path):
    """diffusionGS.py:221-308: which frames, poses and input flags reach densify_pcds (farthest-pose and evenly spaced key
    frames, both densify types, 3 / 4 / 9 input views)."""
    g = np.load(golden_dir / "n2_bookkeeping.npz")
    cap = _run_densify_views(name, tmp_path)
    np.testing.assert_array_equal(cap["frames"], g[f"{name}_sel_frames"])
    np.testing.assert_array_equal(cap["poses"], g[f"{name}_sel_poses"])
    np.testing.assert_array_equal(cap["key_frame_mask"], g[f"{name}_sel_key_frame_mask"])
    np.testing.assert_array_equal(cap["input_flags"], g[f"{name}_sel_input_flags"])
    assert cap["win_samples"] == -1


@pytest.mark.parametrize("name", list(GI.N2_CASES))
def test_densify_pcds_filter_and_pair_graph_match_reference(name, golden_dir):
    """diffusionGS.py:347-435 with recorder networks: keep rule (mask mean > 0.3 or input view), w2c -> c2w, K * 512 / W,
    key-frame indices of the kept list, complete scene graph, dust3r moved on and off the device."""
    from syn3r_amd.diffusionGS import DiffusionGS
    g = np.load(golden_dir / "n2_bookkeeping.npz")
    frames_id, poses = g[f"{name}_sel_frames"], g[f"{name}_sel_poses"]
    kmask, flags = list(g[f"{name}_sel_key_frame_mask"]), list(g[f"{name}_sel_input_flags"])
    means = GI.n2_mask_means(len(frames_id))
    calls, rec = [], {}

    def render_GS(idx=None, pose=None, return_alpha=False):
        calls.append(np.array(pose))
        return pose, np.zeros((3, 4, 6), np.float32), np.ones((4, 6), np.float32), np.ones((4, 6), np.float32)

    def corresp(gs_renderings, svd_outputs, dist_thresh, desc_only):
        assert dist_thresh == 3 and desc_only is False and tuple(svd_outputs[0].shape) == (3, 4, 6)
        return [torch.full((1, 4, 6), float(means[len(calls) - 1]))], None

    class Dust3r:
        def to(self, dev):
            rec.setdefault("to", []).append(dev)

        def make_pairs(self, imgs, scene_graph, global_image_inds):
            rec.update(pair_frames=np.array([float(f[0, 0, 0]) / 255.0 for f in imgs], np.float32), scene_graph=scene_graph,
                       pair_inds=np.array(global_image_inds, np.int64))
            return "pairs"

        def run(self, frames, c2w_poses, intrinsics, preset_pairs):
            assert preset_pairs == "pairs"
            rec.update(run_frames=np.array([float(f[0, 0, 0]) / 255.0 for f in frames], np.float32), c2w=np.array(c2w_poses),
                       K=np.array(intrinsics))
            return None, "trimesh_scene"

    K = np.array([[500.0, 0, 320.0], [0, 510.0, 240.0], [0, 0, 1]], np.float32)
    me = SimpleNamespace(render_GS=render_GS, gsTrainer=SimpleNamespace(generate_corresp_mask=corresp), dust3r=Dust3r(),
                         gs_intrinsics=K, gs_width=640)
    frames = [torch.full((3, 4, 6), float(v)) for v in frames_id]
    out = DiffusionGS.densify_pcds(me, frames, list(poses), key_frame_mask=kmask, input_flags=flags, win_samples=-1)
    assert out == "trimesh_scene" and rec["scene_graph"] == "complete" and rec["to"] == ["cuda", "cpu"]
    assert len(calls) == len(frames_id)                                           # every candidate is rendered once
    for k in ("pair_frames", "pair_inds", "run_frames", "c2w", "K"):
        np.testing.assert_array_equal(rec[k], g[f"{name}_pcd_{k}"], err_msg=k)


def test_view_selection_properties():
    """farthest-pose sampling: starts at pose 0, never repeats, and on a straight equally spaced path with 3 picks
    takes the far end, then the middle."""
    from syn3r_amd import orchestrator as O
    poses = []
    for k in range(9):
        p = np.eye(4, dtype=np.float32)
        p[0, 3] = -0.5 * k                      # w2c translation: centres at x = 0.5 k
        poses.append(p)
    sel = O.view_selection_for_pcd_densification(poses, 3)
    assert sel == [0, 8, 4]
    with pytest.raises(AssertionError):
        O.view_selection_for_pcd_densification(poses[:3], 3)
    assert O.complete_pair_graph([0, 3, 5]) == [(0, 3), (0, 5), (3, 5)]
    t = O.key_frame_template(poses, 9, 4, fps=False)       # linspace(0, 8, 4) -> 0, 2, 5, 8; the last one dropped
    assert list(np.nonzero(t)[0]) == [0, 2, 5] and t.shape == (8,)


def test_outlier_oracle_kdtree_equals_brute_force():
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.standard_normal((1500, 3)), 12.0 + 3.0 * rng.standard_normal((12, 3))])   # a cloud + far strays
    a, b = PO.knn_mean_distance(pts, 20), PO.knn_mean_distance_brute(pts, 20)
    np.testing.assert_array_equal(a, b)
    ind, avg, (mean, std, thr) = PO.remove_statistical_outlier(pts, 20, 3.0)
    assert thr == mean + 3.0 * std and 1450 < len(ind) < 1512 and not set(range(1500, 1512)) & set(ind.tolist())
    small = rng.standard_normal((7, 3))                        # fewer points than neighbours: the mean runs over all 7
    np.testing.assert_array_equal(PO.knn_mean_distance(small, 20), PO.knn_mean_distance_brute(small, 20))
    with pytest.raises(ValueError):
        PO.uniform_down_sample(pts, pts, 0)


def test_flow_cycle_oracle_basics():
    H, W = 12, 20
    fw = np.zeros((2, H, W), np.float32)
    fw[0] = 2.0
    bw = -fw
    m, d = PO.flow_cycle_mask(fw, bw, 3.0)
    assert m[:, : W - 2].all() and not m[:, W - 2:].any() and np.isinf(d[:, W - 1]).all()      # landing outside: rejected
    assert np.all(d[:, : W - 2] == 0)
    bw2 = bw.copy()
    bw2[0, :, 10:] -= 4.0                                    # an inconsistent region: cycle error 4 px > 3 px
    m2, d2 = PO.flow_cycle_mask(fw, bw2, 3.0)
    assert not m2[:, 8: W - 2].any() and m2[:, :7].all()

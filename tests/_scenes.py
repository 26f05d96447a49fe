"""Seeded scenes shared by the parity tests and tests/golden/make_fixtures.py (test infrastructure)."""
import numpy as np


def temporal_scene(seed, H=32, W=1024, n_pts=30000, K_=5):
    """First-frame dict like sample_and_save_temporal.py:281-289, built with the oracle."""
    from lidarcrafter_amd.testing import synth_boxes, synth_points, synth_temporal_inputs
    from oracle import temporal as OT

    trajs, _ = synth_temporal_inputs(seed, K=K_)
    pts = synth_points(n_pts, seed=seed + 100)
    pts[:, 3] = np.floor(pts[:, 3])
    boxes = synth_boxes(K_, pts, seed=seed + 200)
    boxes[:, 3:6] += 3.0                                   # big enough to own some pixels
    names = ["ego"] + [OT.CLASS_NAMES[i % 8] for i in range(K_)]
    gt_boxes = np.concatenate([np.zeros((1, 7), np.float32), boxes]).astype(np.float64)
    item = OT.custom_item(pts, gt_boxes, names, H, W)
    first = dict(gt_fut_trajs=trajs, xyz=item["xyz"], reflectance=item["reflectance"],
                 gt_boxes=gt_boxes, gt_names=names, condition_mask=item["condition_mask"])
    return first, pts, item

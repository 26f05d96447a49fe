"""Autoregressive (temporal) generation glue, resident on the MI355X.

Mirror of the reference's `tools/vis_tools/utils/pipe_related.py` (get_temporal_boxes_3d :28-95,
get_next_frame_points :243-272, refine_next_frame_points :274-283, delete_fg_points :285-291,
remove_ego_points :11-13, interp_trajs_numpy :226-240) and `tools/vis_tools/utils/common.py`
(warp_lidar_future :59-112, warp_boxes_future :115-172, compute_inter_frame_transforms :174-222)
-- the per-frame loop of `tools/evaluation/sample_and_save_temporal.py:262-331`.  Same function
names, argument meaning and return structure, with two differences:

  * point sets are [N,4] float32 CUDA tensors and never leave the device (the reference moves
    every generated frame GPU -> numpy -> GPU); each step is a HIP kernel from temporal.hip /
    geometry.hip: affine transform, masked image -> point list, points-in-boxes, order-preserving
    compaction.  Box / trajectory arithmetic (<= 13 boxes x <= 16 steps of scalars) stays on the
    host in float64 numpy exactly as in the reference.
  * dtype: point sets the reference holds in float32 (frames, `warp_lidar_future` and
    `rotate_points_along_z` results -- float32 library matmuls) are float32 rows, each transform
    evaluated in float64 and rounded once; the sets it holds in FLOAT64 (`Ts @ homo`, the
    [background | re-posed objects] concatenation) are float64 here too and are projected in
    float64 (round 3; pinned on the reference's own outputs, tests/golden/pipe_next.npz).
Also `conduct_obj_data_dict` :200-202 and `get_mask_cond_single` :220-227 (the condition items of
the object branch / the layout-conditioned sampler from user boxes).
"""
from __future__ import annotations

import numpy as np
import torch

from lidarcrafter_amd import ops as K
from lidargen.dataset.custom_dataset import CustomDataset, CustomNuscObjectDataset


# ------------------------------------------------------------------------------ host (trajectories)
def interp_trajs_numpy(trajs: np.ndarray, M: int) -> np.ndarray:
    K_, N, D = trajs.shape
    assert D == 2
    t_orig, t_new = np.linspace(0.0, 1.0, N), np.linspace(0.0, 1.0, M)
    out = np.zeros((K_, M, 2), dtype=trajs.dtype)
    for k in range(K_):
        for d in range(2):
            out[k, :, d] = np.interp(t_new, t_orig, trajs[k, :, d])
    return out


def _step_yaws(future_xy: np.ndarray) -> np.ndarray:
    offsets = np.vstack((future_xy[0:1], future_xy[1:] - future_xy[:-1]))
    yaws = np.arctan2(offsets[:, 1], offsets[:, 0]) - np.pi / 2
    yaws[np.linalg.norm(offsets, axis=1) < 1e-1] = 0.0
    return yaws


def _pose(x, y, z, yaw) -> np.ndarray:
    c, s = np.cos(yaw), np.sin(yaw)
    P = np.eye(4, dtype=float)
    P[:3, :3] = [[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]
    P[:3, 3] = [x, y, z]
    return P


def compute_inter_frame_transforms(future_xy: np.ndarray, z0: float = 0.0) -> np.ndarray:
    """(T,2) ego track -> (T,4,4): M_i maps frame i to frame i+1 (frame 0 = identity pose)."""
    yaws = _step_yaws(future_xy)
    poses = [np.eye(4, dtype=float)] + [_pose(future_xy[i, 0], future_xy[i, 1], z0, yaws[i])
                                        for i in range(future_xy.shape[0])]
    return np.stack([np.linalg.inv(poses[i + 1]) @ poses[i] for i in range(future_xy.shape[0])])


def warp_boxes_future(boxes0: np.ndarray, traj_obj: np.ndarray, traj_ego: np.ndarray,
                      z_e: float) -> np.ndarray:
    """(K,7) boxes, (K,N,2) object tracks, (N,2) ego track -> (K,N,7) boxes in each LiDAR_i frame."""
    K_, N = traj_obj.shape[0], traj_obj.shape[1]
    out = np.zeros((K_, N, 7), dtype=boxes0.dtype)
    yaw_ego = _step_yaws(traj_ego)
    for k in range(K_):
        x0, y0, z0, w, h, l, yaw0 = boxes0[k]
        steps = traj_obj[k, 1:] - traj_obj[k, :-1]
        heading = np.arctan2(steps[:, 1], steps[:, 0])
        still = np.linalg.norm(steps, axis=1) < 1e-3
        yaw_obj = np.empty(N, dtype=boxes0.dtype)
        yaw_obj[0] = yaw0
        for i in range(1, N):                       # a standing object keeps its last heading
            yaw_obj[i] = yaw_obj[i - 1] if still[i - 1] else heading[i - 1]
        for i in range(N):
            d = np.array([x0 + traj_obj[k, i, 0] - traj_ego[i, 0],
                          y0 + traj_obj[k, i, 1] - traj_ego[i, 1], z0 - z_e], dtype=boxes0.dtype)
            c, s = np.cos(yaw_ego[i]), np.sin(yaw_ego[i])
            out[k, i, 0] = c * d[0] + s * d[1]
            out[k, i, 1] = -s * d[0] + c * d[1]
            out[k, i, 2] = d[2]
            out[k, i, 3:6] = [w, h, l]
            out[k, i, 6] = yaw_obj[i] - yaw_ego[i]
    return out


def _warp_matrix(future_xy: np.ndarray, yaws: np.ndarray, i: int, z0: float) -> np.ndarray:
    """p' = R(yaw_i)^T (p - t_i): `(xyz - t).dot(R)` of common.py:100-104 as one 4x4."""
    c, s = np.cos(yaws[i]), np.sin(yaws[i])
    Rt = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
    T = np.eye(4)
    T[:3, :3] = Rt
    T[:3, 3] = -Rt @ np.array([future_xy[i, 0], future_xy[i, 1], z0])
    return T


def warp_lidar_future(P: torch.Tensor, future_xy: np.ndarray, z0: float = 0.0) -> torch.Tensor:
    """[M,4] device points of frame 0 -> [N,M,4] device points in each future LiDAR frame."""
    yaws = _step_yaws(future_xy)
    out = torch.empty((future_xy.shape[0],) + tuple(P.shape), device=P.device, dtype=P.dtype)
    for i in range(future_xy.shape[0]):
        K.transform_points(P, _warp_matrix(future_xy, yaws, i, z0), out=out[i])
    return out


def _box_frame(box7, inverse: bool) -> np.ndarray:
    """LiDAR -> box frame `rotate_points_along_z(p - c, -yaw)` (pipe_related.py:61-65), or back
    `rotate_points_along_z(p, yaw) + c` (:263-266), as one 4x4."""
    x, y, z, yaw = (float(box7[i]) for i in (0, 1, 2, 6))
    T = np.eye(4)
    if inverse:
        c, s = np.cos(yaw), np.sin(yaw)
        T[:3, :3] = [[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]
        T[:3, 3] = [x, y, z]
    else:
        c, s = np.cos(-yaw), np.sin(-yaw)
        R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        T[:3, :3] = R
        T[:3, 3] = -R @ np.array([x, y, z])
    return T


# ------------------------------------------------------------------------------ device (point sets)
def _boxes_dev(boxes_3d, dev) -> torch.Tensor:
    """float32 copy inflated by 0.2 m like roiaware_pool3d_utils.points_in_boxes_cpu."""
    b = torch.as_tensor(np.array(boxes_3d, dtype=np.float32)[:, :7]).clone()
    b[:, 3:6] += 0.2
    return b.to(dev)


def remove_ego_points(points: torch.Tensor, center_radius: float = 2.0) -> torch.Tensor:
    flags = ((points[:, 0].abs() < center_radius) & (points[:, 1].abs() < center_radius)).to(torch.int32)
    return K.compact_points(points.contiguous(), flags, keep_if_zero=True)


def delete_fg_points(points: torch.Tensor, boxes_3d) -> torch.Tensor:
    """Rows of `points` inside none of the boxes (points_in_boxes_cpu semantics)."""
    if len(boxes_3d) == 0 or points.shape[0] == 0:
        return points
    _, cnt = K.points_in_boxes_mask4(points, _boxes_dev(boxes_3d, points.device), 1e-2,
                                     want_mask=False)
    return K.compact_points(points, cnt, keep_if_zero=True)


def _background_points(xyz, reflectance, condition_mask, refl_scale=255.0) -> torch.Tensor:
    pts, keep = K.image_to_points(xyz, reflectance, condition_mask[0], refl_scale=refl_scale,
                                  min_norm=1e-2)
    return K.compact_points(pts, keep)


def refine_next_frame_points(cond_mask_dict_list):
    """Re-project the warped background, drop what the next frame's boxes cover."""
    d = CustomDataset(custom_box_infos=cond_mask_dict_list).__getitem__(0)
    return _background_points(d["xyz"], d["reflectance"], d["condition_mask"])


def conduct_obj_data_dict(unscaled_boxes_list):
    """pipe_related.py:200-202: [{gt_boxes [1+K,7], gt_names}] -> the object branch's item."""
    return CustomNuscObjectDataset(custom_box_infos=unscaled_boxes_list).__getitem__(0)


def get_mask_cond_single(cond_mask_dict_list, temporal=False, inpaint=False):
    """pipe_related.py:220-227: user boxes (+ points) -> the conditioning item of one frame."""
    dataset = CustomDataset(custom_box_infos=cond_mask_dict_list)
    if temporal:
        setattr(dataset, "task", "autoregressive_generation")
    if inpaint:
        setattr(dataset, "inpaint_mode", True)
    return dataset.__getitem__(0)


def get_temporal_boxes_3d(first_frame_data_dict, M=None):
    """-> (curr_background_points, fut_background_points [T,N,4], curr_boxes_3d, fut_boxes_3d
    [K,T,7], Ts [T,4,4], align_obj_points list[K] of [n_k,3], align_obj_intensity list[K])."""
    f = first_frame_data_dict
    a = np.insert(f["gt_fut_trajs"], 0, 0, axis=1)
    acc = np.cumsum(a, axis=1)
    if M is not None:
        acc = interp_trajs_numpy(acc, M=M)
    a = acc[:, 1:] - acc[:, :-1]
    ego_xy = np.cumsum(a[0], axis=0)
    obj_xy = np.cumsum(a[1:], axis=1)

    xyz, refl, cond = f["xyz"], f["reflectance"], f["condition_mask"]
    boxes = np.asarray(f["gt_boxes"])[1:, :7]
    assert obj_xy.shape[0] == boxes.shape[0]
    pts, keep = K.image_to_points(xyz, refl, None, refl_scale=255.0, ego_radius=2.0)
    cur = K.compact_points(pts, keep)
    mask, _ = K.points_in_boxes_mask4(cur, _boxes_dev(boxes, cur.device), 1e-2, want_count=False)
    obj_pts, obj_int = [], []
    for k, box in enumerate(boxes):
        p = K.compact_points(cur, mask[k].contiguous())
        local = K.transform_points(p, _box_frame(box, inverse=False))
        obj_pts.append(local[:, :3])
        obj_int.append(p[:, 3])
    bg = _background_points(xyz, refl, cond)
    fut_boxes = warp_boxes_future(boxes0=boxes, traj_obj=obj_xy, traj_ego=ego_xy, z_e=0.0)
    fut_bg = warp_lidar_future(P=bg, future_xy=ego_xy, z0=0.0)
    Ts = compute_inter_frame_transforms(future_xy=ego_xy, z0=0.0)
    return bg, fut_bg, boxes, fut_boxes, Ts, obj_pts, obj_int


def get_next_frame_points(curr_background_points, align_obj_points, align_obj_intensity,
                          fut_boxes_3d, fut_boxes_names, Ts):
    """Background moved by Ts and re-projected + every object re-posed in its future box --
    pipe_related.py:243-269 with the reference's dtypes: the moved background is a FLOAT64 set
    (`Ts @ homo`) that `refine_next_frame_points` projects in float64 (float32 rows come back);
    the objects are `float32 rotation + float64 centre`; the result is the float64 concatenation."""
    fut_bg = K.transform_points(curr_background_points.float().contiguous(), Ts, f64="keep")
    fut_bg = refine_next_frame_points([{
        "points": fut_bg, "gt_boxes": np.concatenate([np.zeros((1, 7)), fut_boxes_3d]),
        "gt_names": fut_boxes_names}])
    parts = [fut_bg.double()]
    for k, box in enumerate(fut_boxes_3d):
        if align_obj_points[k].shape[0] == 0:
            continue
        p = torch.cat([align_obj_points[k], align_obj_intensity[k][:, None]], dim=1).contiguous()
        parts.append(K.transform_points(p, _box_frame(box, inverse=True), f64="rot32"))
    return torch.cat(parts, dim=0)


# ------------------------------------------------------------------------------ the per-batch loop
@torch.inference_mode()
def generate_sequence(ddpm, auto_ddpm, lidar_utils, batch: dict, num_frames: int = 5,
                      num_steps: int = 256, mode: str = "ddpm", first_mode: str = None,
                      traj_length: int = 16, rng=None, num_classes: int = 9,
                      auto_uses_reflectance: bool = False, data_cfg=None, progress: bool = False,
                      trace: list = None):
    """Frame 0 from the layout-conditioned sampler, frames 1.. autoregressively -- the loop of
    tools/evaluation/sample_and_save_temporal.py:198-331 without its file output, every point set
    resident on the device.

    batch: first-frame conditioning as the reference's collate_fn leaves it -- tensors
    `scaled_gt_boxes`, `gt_boxes_2d`, `is_valid_obj`, `condition_mask` [B,2,H,W] (+ whatever else
    the condition model reads) and per-sample lists `gt_boxes` ([1+K, >=7], ego row first),
    `gt_names`, `gt_fut_trajs` ([1+K, T, 2] per-step offsets, ego row first).
    Returns (frames, points): frames[t] = [B,5,H,W] (depth, x, y, z, reflectance) and
    points[t][b] = the [H*W,4] (x,y,z,reflectance) rows of sample b at frame t.
    `trace` (a list) receives one dict per autoregressive frame with the hand-overs of the loop
    (`next_points` per sample, the collated `condition_mask` / raw `autoregressive_cond`, the
    background sets the frame started from): what the parity tests feed to the oracle glue.
    `data_cfg.resolution` is the resolution of the temporal CustomDataset items (the reference
    always builds them at 32x1024 and resizes `autoregressive_cond` to the model's resolution with
    nearest-exact interpolation, sample_and_save_temporal.py:133-148,177-179 -- done here too when
    the two differ)."""
    dev = ddpm.device
    B = batch["condition_mask"].shape[0]
    cond_batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    cond_batch["concat_cond"] = lidar_utils.preprocess_condition_mask(cond_batch["condition_mask"],
                                                                     num_classes)
    x = ddpm.sample(cond_batch, B, num_steps, progress=progress, rng=rng,
                    mode=first_mode or mode).clamp(-1, 1)
    frame = lidar_utils.postprocess(x)                       # [B,5,H,W]: depth, xyz, reflectance
    H, W = frame.shape[-2:]

    def rows(fr, b):                                         # samples[b,[1,2,3,4]].reshape(4,-1).T
        pts, _ = K.image_to_points(fr[b, 1:4].contiguous(), fr[b, 4].contiguous(), None, 1.0)
        return pts

    frames, points = [frame], [[rows(frame, b) for b in range(B)]]
    cur_bg = list(points[0])
    per = []
    for b in range(B):
        a = np.insert(np.asarray(batch["gt_fut_trajs"][b]), 0, 0, axis=1)
        acc = interp_trajs_numpy(np.cumsum(a, axis=1), M=traj_length)
        first = dict(gt_fut_trajs=acc[:, 1:] - acc[:, :-1], xyz=frame[b, 1:4].contiguous(),
                     reflectance=frame[b, 4:5].contiguous(), gt_boxes=batch["gt_boxes"][b],
                     gt_names=batch["gt_names"][b], condition_mask=cond_batch["condition_mask"][b])
        _, fut_bg, _, fut_boxes, Ts, obj_pts, obj_int = get_temporal_boxes_3d(first)
        per.append(dict(fut_bg=fut_bg, fut_boxes=fut_boxes, Ts=Ts, obj_pts=obj_pts, obj_int=obj_int,
                        names=batch["gt_names"][b]))
    n_fut = min(num_frames - 1, per[0]["Ts"].shape[0]) if B else 0
    for t in range(n_fut):
        infos = []
        for b in range(B):
            p = per[b]
            nxt = get_next_frame_points(cur_bg[b], p["obj_pts"], p["obj_int"], p["fut_boxes"][:, t],
                                        p["names"], p["Ts"][t])
            gt_boxes = np.concatenate([np.zeros((1, 7), np.float32), p["fut_boxes"][:, t]], axis=0)
            infos.append(dict(points=nxt, gt_boxes=gt_boxes, gt_names=p["names"]))
        ds = CustomDataset(infos, cfg=data_cfg)
        ds.task = "autoregressive_generation"
        tb = ds.collate_fn([ds[i] for i in range(B)])
        tb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
        tb["concat_cond"] = lidar_utils.preprocess_condition_mask(tb["condition_mask"], num_classes)
        ar = tb["autoregressive_cond"]                       # [B,2,H,W]: metric depth, reflectance
        if trace is not None:
            trace.append(dict(t=t, next_points=[i["points"] for i in infos], cur_bg=list(cur_bg),
                              gt_boxes=[i["gt_boxes"] for i in infos], Ts=[p["Ts"][t] for p in per],
                              condition_mask=tb["condition_mask"].clone(),
                              autoregressive_cond=ar.clone()))
        chans = [lidar_utils.convert_depth(ar[:, 0:1])]
        if auto_uses_reflectance:
            chans.append(ar[:, 1:2])
        arc = lidar_utils.normalize(torch.cat(chans, dim=1))
        if tuple(arc.shape[-2:]) != (H, W):
            arc = torch.nn.functional.interpolate(arc, size=(H, W), mode="nearest-exact")
        tb["autoregressive_cond"] = arc.contiguous()
        x = auto_ddpm.sample(tb, B, num_steps, progress=progress, rng=rng, mode=mode).clamp(-1, 1)
        frame = lidar_utils.postprocess(x)
        frames.append(frame)
        points.append([rows(frame, b) for b in range(B)])
        for b in range(B):
            comb = torch.cat([per[b]["fut_bg"][t], points[-1][b]], dim=0).contiguous()
            cur_bg[b] = delete_fg_points(comb, infos[b]["gt_boxes"][1:, :7])
    return frames, points

"""`CustomDataset`: user-supplied boxes (+ points) -> one conditioning item, on the device.

Mirror of the reference's `lidargen/dataset/custom_dataset.py:43-89` together with the parts of
`NuscDataset` it inherits and that need no nuScenes files: `pre_process`
(nuscenes_dataset.py:375-421), `scale_boxes_3d` :145-159, `allign_box_num` :175-193,
`encoding_boxes_3d` :195-216, `distille_local_boxes` :247-258 and `DatasetBase.collate_fn`
(base_dataset.py:38-71).  Images (`xyz`, `reflectance`, `depth`, `mask`, `autoregressive_cond`,
`condition_mask`, `scene_loss_weight_map`) are float32 CUDA tensors produced by the projection /
layout kernels; the per-box scalars (<= 13 rows) are float numpy like the reference.
Tasks: 'layout_cond' and 'autoregressive_generation' (the scene-graph task 'layout_generation'
needs the CLIP text encoder: out of scope)."""
from __future__ import annotations

from collections import defaultdict

import numpy as np
import torch

from lidarcrafter_amd import ops as K

from .transforms_3d import common


class DataConfig:
    """custom_dataset.py:9-24 defaults."""
    dataset = "custom"
    class_names = ("car", "truck", "construction_vehicle", "bus", "trailer", "motorcycle", "bicycle",
                   "pedestrian")
    resolution = (32, 1024)
    min_depth, max_depth = 1.45, 80.0
    fov_up, fov_down = 10.0, -30.0
    scan_unfolding = False
    split = "train"
    task = "layout_cond"


class CustomDataset:
    def __init__(self, custom_box_infos, cfg=None):
        self.cfg = cfg if cfg is not None else DataConfig()
        self.task = getattr(self.cfg, "task", "layout_cond")
        self.points_range = [-80, -80, -8, 80, 80, 8]
        self.custom_box_infos = custom_box_infos
        self.data = custom_box_infos
        if self.cfg.scan_unfolding:
            raise NotImplementedError("scan_unfolding=True is not on the path (nuScenes configs: False)")

    def __len__(self):
        return len(self.data)

    # ---- per-box scalar encodings (host, float numpy like the reference) ------------------------
    def scale_boxes_3d(self, boxes_3d):
        out = np.zeros([boxes_3d.shape[0], boxes_3d.shape[-1] + 1])
        x_min, y_min, z_min = self.points_range[:3]
        boxes_3d[:, 0] = boxes_3d[:, 0] / (0 - x_min)
        boxes_3d[:, 1] = boxes_3d[:, 1] / (0 - y_min)
        boxes_3d[:, 2] = boxes_3d[:, 2] / (0 - z_min)
        boxes_3d[:, 3:6] = np.log(boxes_3d[:, 3:6] + 1e-6)
        out[:, :6] = boxes_3d[:, :6]
        out[:, 6] = np.sin(boxes_3d[:, 6])
        out[:, 7] = np.cos(boxes_3d[:, 6])
        if boxes_3d.shape[-1] > 7:
            out[:, 8:] = boxes_3d[:, 7:]
        return out

    def allign_box_num(self, bbox_3d, bbox_2d, fg_encoding_box, expet_box_num=13):
        n = bbox_3d.shape[0]
        if n > expet_box_num:
            return (bbox_3d[:expet_box_num], bbox_2d[:expet_box_num], fg_encoding_box[:expet_box_num],
                    np.ones([expet_box_num]))
        b3 = np.zeros([expet_box_num, bbox_3d.shape[-1]])
        b2 = np.zeros([expet_box_num, bbox_2d.shape[-1]])
        enc = np.zeros([expet_box_num, fg_encoding_box.shape[-1]])
        b3[:n], b2[:n], enc[:n] = bbox_3d, bbox_2d, fg_encoding_box
        valid = np.zeros([expet_box_num])
        valid[:n] = 1
        return b3, b2, enc, valid

    def encoding_boxes_3d(self, boxes_3d, unique_mode=True):
        cb = np.zeros((8), dtype=np.float32)
        x, y, z, w, h, l, yaw = boxes_3d
        x_min, y_min, z_min = self.points_range[:3]
        cb[0] = np.linalg.norm(np.array([x / (0 - x_min), y / (0 - y_min)]), ord=2, axis=0)
        cb[1] = z / (0 - z_min)
        cb[2:5] = np.log(np.array([w, h, l]) + 1e-6)
        if unique_mode:
            cb[5] = yaw - np.arctan2(y, x)
            return cb[:6]
        cb[5] = (-np.arctan2(y, x) / np.pi + 1) / 2 % 1
        cb[6], cb[7] = np.sin(yaw), np.cos(yaw)
        return cb

    def distille_local_boxes(self, data_dict, unique_mode=True):
        fg = np.asarray(data_dict["gt_boxes"])[1:]
        names = data_dict["gt_names"][1:]
        data_dict["fg_encoding_box"] = np.stack(
            [self.encoding_boxes_3d(b[:7], unique_mode) for b in fg], axis=0)
        data_dict["fg_class"] = np.array([self.cfg.class_names.index(n) for n in names])
        return data_dict

    # ---- item ---------------------------------------------------------------------------------
    def pre_process(self, data_dict):
        if self.task not in ("layout_cond", "autoregressive_generation"):
            raise NotImplementedError(f"task {self.task!r} is out of scope of the MI355X hot path")
        H, W = self.cfg.resolution
        data_dict = self.distille_local_boxes(data_dict, unique_mode=False)
        data_dict.pop("fg_class", None)
        class_names = ["ego"] + list(self.cfg.class_names)
        cls = np.array([class_names.index(n) for n in data_dict["gt_names"]], dtype=np.int32)
        gt_boxes = np.concatenate((np.asarray(data_dict["gt_boxes"]),
                                   cls.reshape(-1, 1).astype(np.float32)), axis=1)
        data_dict["gt_boxes"] = gt_boxes
        dev = torch.device("cuda", torch.cuda.current_device())
        # the reference's dtype flow follows the boxes: float64 `gt_boxes` (every caller on the path:
        # python lists / float64 arrays + the float32 class column) -> all-float64 geometry
        bdt = np.float64 if gt_boxes.dtype == np.float64 else np.float32
        b2d, cond, wmap = common.convert_boxes_to_2d(
            torch.from_numpy(np.ascontiguousarray(gt_boxes, bdt)).to(dev), H=H, W=W,
            min_depth=self.cfg.min_depth, max_depth=self.cfg.max_depth, fov_up=self.cfg.fov_up,
            fov_down=self.cfg.fov_down)
        scaled = self.scale_boxes_3d(gt_boxes.copy())
        b3, b2, enc, valid = self.allign_box_num(scaled[1:], b2d[1:].double().cpu().numpy(),
                                                 data_dict["fg_encoding_box"])
        data_dict.update(scaled_gt_boxes=b3, fg_encoding_box=enc, gt_boxes_2d=b2, is_valid_obj=valid,
                         condition_mask=cond, scene_loss_weight_map=wmap)
        data_dict.pop("points", None)
        return data_dict

    def __getitem__(self, idx, inpaint=False):
        d = dict(self.data[idx])
        if "points" in d:
            H, W = self.cfg.resolution
            pts = d["points"]
            if not isinstance(pts, torch.Tensor):
                pts = torch.from_numpy(np.ascontiguousarray(
                    pts, np.float64 if pts.dtype == np.float64 else np.float32)).cuda()
            elif pts.dtype != torch.float64:
                pts = pts.float()
            # float64 point sets (the temporal glue's) are projected in float64, like the reference
            img = common.load_points_as_images(points=pts.contiguous(), scan_unfolding=False,
                                               H=H, W=W, min_depth=self.cfg.min_depth,
                                               max_depth=self.cfg.max_depth, fov_up=self.cfg.fov_up,
                                               fov_down=self.cfg.fov_down)
            x = img.permute(2, 0, 1).contiguous()
            x = x * x[5:6]
            # true division like numpy (a Python-scalar divisor would become a reciprocal multiply)
            d.update(xyz=x[:3].contiguous(), reflectance=x[3:4] / torch.full((), 255.0, device=x.device),
                     depth=x[4:5], mask=x[5:6])
            if self.task == "autoregressive_generation":
                d["autoregressive_cond"] = torch.cat([d["depth"], d["reflectance"]], dim=0)
                if not getattr(self, "inpaint_mode", False):
                    for k in ("depth", "reflectance", "mask", "xyz"):
                        d.pop(k)
        return self.pre_process(d)

    def unscaled_objs_3d(self, sample_index, custom_data_dict, generated_object_points,
                         w_semantic=False):
        """Generated foreground objects [n_obj, N, 4] (unit box frame, intensity in [-1, 1]) ->
        scene-frame rows [n_obj * N, 4 (+ class)] -- reference nuscenes_dataset.py:215-243: scale by
        the half extents, intensity 255 (i + 1) / 2, rotate about z by the box yaw, translate to the
        centre.  CUDA tensor in -> CUDA tensor out: scale, rotation and translation of one object
        are ONE 4x4 affine (lc_transform_points, evaluated in fp64, rounded once); numpy in ->
        numpy out through the same kernel."""
        info = custom_data_dict if custom_data_dict is not None else self.data[sample_index]
        gt_boxes = np.asarray(info["gt_boxes"], np.float64)[1:, :7]
        is_numpy = isinstance(generated_object_points, np.ndarray)
        pts = (torch.from_numpy(np.ascontiguousarray(generated_object_points, np.float32)).cuda()
               if is_numpy else generated_object_points.float())
        assert gt_boxes.shape[0] == pts.shape[0]
        classes = None
        if w_semantic:
            names = list(info["gt_names"])[1:]
            classes = [self.cfg.class_names.index(n) + 1 for n in names]
        rows = []
        for k, box in enumerate(gt_boxes):
            c, s_ = np.cos(box[6]), np.sin(box[6])
            sx, sy, sz = box[3] / 2.0, box[4] / 2.0, box[5] / 2.0
            T = np.array([[c * sx, -s_ * sy, 0.0, box[0]], [s_ * sx, c * sy, 0.0, box[1]],
                          [0.0, 0.0, sz, box[2]], [0.0, 0.0, 0.0, 1.0]])
            p = K.transform_points(pts[k].contiguous(), T)
            p[:, 3] = 255 * (p[:, 3] + 1) / 2
            if classes is not None:
                p = torch.cat([p, torch.full((p.shape[0], 1), float(classes[k]), device=p.device)], 1)
            rows.append(p)
        out = torch.cat(rows, 0)
        return out.cpu().numpy() if is_numpy else out

    def collate_fn(self, batch_list, _unused=False):
        data = defaultdict(list)
        for cur in batch_list:
            for key, val in cur.items():
                data[key].append(val)
        skip = ["points", "gt_names", "gt_boxes", "gt_box_relationships", "gt_fut_trajs",
                "gt_fut_masks", "gt_fut_states", "token", "custom_tokens"]
        ret = {}
        for key, val in data.items():
            if key not in skip:
                if isinstance(val[0], torch.Tensor):
                    ret[key] = torch.stack(val, dim=0).float()
                else:
                    ret[key] = torch.from_numpy(np.stack(val, axis=0)).float()
            if key in ("token", "custom_tokens", "gt_boxes", "gt_names", "gt_fut_trajs"):
                ret[key] = val
        ret["batch_size"] = len(batch_list)
        return ret


class NuscObjectConfig(DataConfig):
    """custom_dataset.py:26-41 defaults."""
    task = "object_generation"


class CustomNuscObjectDataset(CustomDataset):
    """custom_dataset.py:91-108: user-supplied boxes -> the foreground-object branch's condition
    item (`fg_encoding_box` [K,6] in the unique-yaw encoding, `fg_class` [K]) -- host scalars only,
    like the reference (<= 13 rows)."""

    def __init__(self, custom_box_infos, cfg=None):
        super().__init__(custom_box_infos, cfg if cfg is not None else NuscObjectConfig())

    def __getitem__(self, idx):
        return self.distille_local_boxes(self.data[idx])       # mutates + returns the entry, :105-108

"""CPU restatement of the foreground-object branch.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Follows /root/reference/lidargen/models/unets/point_unet.py:14-71 (PCNet / PointUNet),
encoders/object_gen_encoder.py:7-88 + encoders/embedder.py:5-57 (ObjectGenEncoder), and
lidargen/dataset/nuscenes_dataset.py:215-243 (unscaled_objs_3d).  Pinned on outputs of the
reference modules (tests/golden/object.npz)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def fourier_embed(x: torch.Tensor, num_freqs: int = 4) -> torch.Tensor:
    parts = [x]
    for f in (2.0 ** torch.linspace(0.0, num_freqs - 1, steps=num_freqs)).tolist():
        parts += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(parts, -1)


def object_encoder_forward(sd: dict, batch: dict, text_feats: dict, classes) -> torch.Tensor:
    pos = fourier_embed(batch["fg_encoding_box"])
    cls = torch.stack([text_feats[classes[i]] for i in batch["fg_class"].flatten().long().tolist()])
    h = F.silu(F.linear(pos, sd["bbox_proj.weight"], sd["bbox_proj.bias"]))
    h = torch.cat([h, cls], -1)
    h = F.silu(F.linear(h, sd["second_linear.0.weight"], sd["second_linear.0.bias"]))
    h = F.silu(F.linear(h, sd["second_linear.2.weight"], sd["second_linear.2.bias"]))
    return F.linear(h, sd["second_linear.4.weight"], sd["second_linear.4.bias"])


def point_unet_forward(sd: dict, coords: torch.Tensor, lam: torch.Tensor, cond: torch.Tensor,
                       residual: bool = True) -> torch.Tensor:
    B = coords.shape[0]
    beta = lam.view(B, 1, 1)
    emb = torch.cat([beta, torch.sin(beta), torch.cos(beta), cond.view(B, 1, -1)], -1)
    out = coords
    for i in range(6):
        p = f"layers.{i}."
        gate = torch.sigmoid(F.linear(emb, sd[p + "cond_gate.weight"], sd[p + "cond_gate.bias"]))
        bias = F.linear(emb, sd[p + "cond_bias.weight"])
        out = F.linear(out, sd[p + "fea_layer.weight"], sd[p + "fea_layer.bias"]) * gate + bias
        if i < 5:
            out = F.leaky_relu(out)
    return coords + out if residual else out


def unscaled_objs_3d(gt_boxes: np.ndarray, generated: np.ndarray, classes=None) -> np.ndarray:
    """nuscenes_dataset.py:215-243: generated [n_obj, N, 4] points in the unit box frame ->
    scene frame: scale by the half extents, intensity 255 (i + 1) / 2, rotate about z by the box
    yaw (dataset/utils.py rotate_points_along_z), translate to the box centre; optional class
    column; vstack."""
    outs = []
    for k, box in enumerate(gt_boxes[:, :7]):
        p = generated[k].copy()
        p[:, 0] = p[:, 0] * box[3] / 2.0
        p[:, 1] = p[:, 1] * box[4] / 2.0
        p[:, 2] = p[:, 2] * box[5] / 2.0
        p[:, 3] = 255 * (p[:, 3] + 1) / 2
        c, s = np.cos(box[6]), np.sin(box[6])
        rot = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]], dtype=p.dtype)
        p[:, :3] = p[:, :3] @ rot
        p[:, :3] = p[:, :3] + box[:3].reshape(1, 3)
        if classes is not None:
            p = np.hstack((p, np.full((p.shape[0], 1), classes[k])))
        outs.append(p)
    return np.vstack(outs)

"""Condition encoder of the foreground-object branch -- API / state_dict mirror of the reference's
lidargen/models/unets/encoders/object_gen_encoder.py:7-88 (`ObjectGenEncoder`): Fourier embedding
of the object's box code `fg_encoding_box` [B, 6] -> Linear -> SiLU, concatenated with the CLIP text
feature of its class (512-d, from `../data/clips/nuscenes/obj_text_feat.pkl`), then a 3-layer MLP
-> the 768-d condition of `PointUNet`.  Runs once per batch of objects; the dense layers go through
lc_linear_fwd (SiLU fused).  The text features are data, not parameters (not in the state_dict):
`prepare()` loads the reference's pickle; `set_text_features()` injects a dict directly (tests,
synthetic runs -- no CLIP features ship with this repository)."""
from __future__ import annotations

import pickle

import torch
import torch.nn.functional as F
from torch import nn

from lidarcrafter_amd import autograd as AG
from lidarcrafter_amd import ops as K

from .embedder import get_embedder

_NUSC = ["car", "truck", "construction_vehicle", "bus", "trailer", "motorcycle", "bicycle",
         "pedestrian"]


class ObjectGenEncoder(nn.Module):
    def __init__(self, num_class, input_dim=6, embedder_num_freq=4, class_token_dim=512,
                 use_text_encoder_init=True, output_num=1, proj_dims=(768, 512, 512, 768),
                 object_classes=tuple(_NUSC)):
        super().__init__()
        self.prepare_called = False
        self.num_class = num_class
        self.fourier_embedder = get_embedder(input_dim, embedder_num_freq)
        self.use_text_encoder_init = use_text_encoder_init
        self.object_classes = list(object_classes)
        self.bbox_proj = nn.Linear(self.fourier_embedder.out_dim * output_num, proj_dims[0])
        self.second_linear = nn.Sequential(
            nn.Linear(proj_dims[0] + class_token_dim, proj_dims[1]), nn.SiLU(),
            nn.Linear(proj_dims[1], proj_dims[2]), nn.SiLU(),
            nn.Linear(proj_dims[2], proj_dims[3]))
        self.obj_text_feat = {}

    # ---- class text features -------------------------------------------------------------------
    def set_text_features(self, feats: dict, device=None) -> None:
        """{class name: [512] tensor / array}; marks the encoder prepared."""
        self.obj_text_feat = {k: torch.as_tensor(v).float().squeeze().to(device or "cpu")
                              for k, v in feats.items()}
        self.prepare_called = True

    def prepare(self, device="cuda", path="../data/clips/nuscenes/obj_text_feat.pkl"):
        if self.use_text_encoder_init:
            with open(path, "rb") as f:
                self.set_text_features(pickle.load(f), device)
        self.prepare_called = True

    def _class_tokens(self, classes: torch.Tensor, device) -> torch.Tensor:
        names = [self.object_classes[i] for i in classes.flatten().long().tolist()]
        return torch.stack([self.obj_text_feat[n].to(device) for n in names], dim=0)

    # ---- forward -------------------------------------------------------------------------------
    def forward_feature(self, pos_emb: torch.Tensor, cls_emb: torch.Tensor) -> torch.Tensor:
        lead = pos_emb.shape[:-1]
        p = pos_emb.reshape(-1, pos_emb.shape[-1]).float().contiguous()
        c = cls_emb.reshape(-1, cls_emb.shape[-1]).float()
        sl = self.second_linear
        if AG.training_active(self, p, c):           # training (tools/train/train_object.py): torch ops
            h = F.silu(F.linear(p, self.bbox_proj.weight, self.bbox_proj.bias))
            h = sl(torch.cat([h, c], dim=-1))
            return h.reshape(*lead, h.shape[-1])
        h = K.linear(p, self.bbox_proj.weight, self.bbox_proj.bias, act_out=True)
        h = torch.cat([h, c], dim=-1).contiguous()
        h = K.linear(h, sl[0].weight, sl[0].bias, act_out=True)
        h = K.linear(h, sl[2].weight, sl[2].bias, act_out=True)
        h = K.linear(h, sl[4].weight, sl[4].bias)
        return h.reshape(*lead, h.shape[-1])

    def forward(self, input_dict: dict) -> torch.Tensor:
        if not self.prepare_called:
            self.prepare()
        boxes = input_dict["fg_encoding_box"]                    # [B, 6]
        pos = self.fourier_embedder(boxes)
        return self.forward_feature(pos, self._class_tokens(input_dict["fg_class"], boxes.device))

    def forward_scene(self, input_dict: dict) -> torch.Tensor:
        if not self.prepare_called:
            self.prepare()
        boxes = input_dict["fg_encoding_box"]                    # [B, n_obj, 6]
        B = boxes.shape[0]
        cls = self._class_tokens(input_dict["fg_class"], boxes.device)
        return self.forward_feature(self.fourier_embedder(boxes), cls.reshape(B, -1, cls.shape[-1]))

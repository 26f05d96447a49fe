"""Layout condition encoder -- API / state_dict mirror of the reference's
lidargen/models/unets/layout_encoder.py (Transformer :61-137, LayoutTransformerEncoder :140-303).

Runs ONCE per batch (not per denoising step): 13 object tokens, width 64, 6 layers = 9 MFLOP.
SURVEY.md §8a-16 keeps it in plain tensor ops; it is device agnostic (the reference hard-codes
`.cuda()` at :217) and emits the same condition dict the denoiser consumes."""
from __future__ import annotations

import math
import os

import torch
import torch as th
import torch.nn as nn

CORE_GRAPH = os.environ.get("LC_ENCODER_GRAPH", "1") != "0"   # the layout-dependent core as one replayed HIP graph inside sampling runs


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return super().forward(x.float()).to(x.dtype)


class QKVMultiheadAttention(nn.Module):
    def __init__(self, n_heads: int, n_ctx: int):
        super().__init__()
        self.n_heads, self.n_ctx = n_heads, n_ctx

    def forward(self, qkv, key_padding_mask=None):
        B, T, width = qkv.shape
        ch = width // self.n_heads // 3
        scale = 1 / math.sqrt(math.sqrt(ch))
        q, k, v = qkv.view(B, T, self.n_heads, -1).split(ch, dim=-1)
        w = th.einsum("bthc,bshc->bhts", q * scale, k * scale)
        if key_padding_mask is not None:
            w = w.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        w = th.softmax(w.float(), dim=-1).type(w.dtype)
        return th.einsum("bhts,bshc->bthc", w, v).reshape(B, T, -1)


class MultiheadAttention(nn.Module):
    def __init__(self, n_ctx, width, heads):
        super().__init__()
        self.c_qkv = nn.Linear(width, width * 3)
        self.c_proj = nn.Linear(width, width)
        self.attention = QKVMultiheadAttention(heads, n_ctx)

    def forward(self, x, key_padding_mask=None):
        return self.c_proj(self.attention(self.c_qkv(x), key_padding_mask))


class MLP(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.c_fc = nn.Linear(width, width * 4)
        self.c_proj = nn.Linear(width * 4, width)
        self.gelu = nn.GELU()

    def forward(self, x):
        return self.c_proj(self.gelu(self.c_fc(x)))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, n_ctx, width, heads):
        super().__init__()
        self.attn = MultiheadAttention(n_ctx, width, heads)
        self.ln_1 = LayerNorm(width)
        self.mlp = MLP(width)
        self.ln_2 = LayerNorm(width)

    def forward(self, x, key_padding_mask=None):
        x = x + self.attn(self.ln_1(x), key_padding_mask)
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, n_ctx, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList(
            [ResidualAttentionBlock(n_ctx, width, heads) for _ in range(layers)])

    def forward(self, x, key_padding_mask=None):
        for blk in self.resblocks:
            x = blk(x, key_padding_mask)
        return x


class LayoutTransformerEncoder(nn.Module):
    def __init__(self, feature_map_size: list, layout_length: int, hidden_dim: int,
                 output_dim: int, num_layers: int, num_heads: int, use_final_ln: bool,
                 num_classes_for_layout_object: int, mask_size_for_layout_object: int,
                 used_condition_types=("obj_class", "obj_bbox", "obj_mask"),
                 use_positional_embedding=True, resolution_to_attention=(),
                 use_key_padding_mask=False, not_use_layout_fusion_module=False, fov_up=10,
                 fov_down=-30, **kwargs):
        super().__init__()
        self.feature_map_size = feature_map_size
        self.not_use_layout_fusion_module = not_use_layout_fusion_module
        self.use_key_padding_mask = use_key_padding_mask
        self.used_condition_types = list(used_condition_types)
        if not not_use_layout_fusion_module:
            self.transform = Transformer(layout_length, hidden_dim, num_layers, num_heads)
        self.use_positional_embedding = use_positional_embedding
        if use_positional_embedding:
            self.positional_embedding = nn.Parameter(th.empty(layout_length, hidden_dim))
            nn.init.normal_(self.positional_embedding, std=0.01)
        self.transformer_proj = nn.Linear(hidden_dim, output_dim)
        if "obj_class" in self.used_condition_types:
            self.obj_class_embedding = nn.Embedding(num_classes_for_layout_object, hidden_dim)
        if "obj_bbox" in self.used_condition_types:
            self.obj_bbox_2d_embedding = nn.Linear(4, hidden_dim)
            self.obj_bbox_embedding = nn.Linear(8, hidden_dim)
        if "obj_mask" in self.used_condition_types:
            self.obj_mask_embedding = nn.Linear(mask_size_for_layout_object ** 2, hidden_dim)
        self.final_ln = LayerNorm(hidden_dim) if use_final_ln else None
        self.dtype = torch.float32
        self.resolution_to_attention = list(resolution_to_attention)
        # normalised cell corners (x0, y0, x1, y1) of every feature-map cell, per attention level
        self.image_patch_bbox_embedding = {}
        for r in self.resolution_to_attention:
            nh, nw = int(feature_map_size[0] / r), int(feature_map_size[1] / r)
            di, dj = 1.0 / (feature_map_size[0] / r), 1.0 / (feature_map_size[1] / r)
            self.image_patch_bbox_embedding[f"resolution{nh}"] = torch.FloatTensor(
                [(dj * j, di * i, dj * (j + 1), di * (i + 1)) for i in range(nh) for j in range(nw)])
        self.out_channels = kwargs.get("out_channels", 10)

    def _patch_embedding(self, key, dev):
        """Embedding of the feature-map cells of one attention level: a function of `obj_bbox_2d_embedding` alone (the
        reference recomputes it per forward, layout_encoder.py:228-237).  Kept per (weights, device) outside grad mode."""
        lin = self.obj_bbox_2d_embedding
        if torch.is_grad_enabled() and (lin.weight.requires_grad or lin.bias.requires_grad):
            cells = self.image_patch_bbox_embedding[key].to(dev, self.dtype)
            return lin(cells).t().contiguous(), None
        tag = (key, str(dev), lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version)
        cache = self.__dict__.setdefault("_patch_cache", {})
        ent = cache.get(key)
        if ent is None or ent[0] != tag:
            cells = self.image_patch_bbox_embedding[key].to(dev, self.dtype)
            with torch.no_grad():
                ent = cache[key] = (tag, lin(cells).t().contiguous())
        return ent[1], tag

    def forward(self, condition_dict, obj_class=None, obj_bbox=None, obj_mask=None,
                is_valid_obj=None, image_patch_bbox=None):
        boxes = condition_dict["scaled_gt_boxes"]
        dev = boxes.device
        core = self._core_by_graph(boxes, condition_dict["gt_boxes_2d"], condition_dict["is_valid_obj"])
        out = core if core is not None else \
            self._core(boxes, condition_dict["gt_boxes_2d"], condition_dict["is_valid_obj"], obj_mask)
        if "obj_bbox" in self.used_condition_types:
            for r in self.resolution_to_attention:
                key = f"resolution{int(self.feature_map_size[0] / r)}"
                emb, tag = self._patch_embedding(key, dev)                        # [hidden, L]
                # same rows for every sample: a stride-0 batch view, not B copies
                view = emb[None].expand(boxes.shape[0], -1, -1)
                # (the rows depend on this module's weights only, not on the condition: a consumer that has
                #  derived something from a view with the same tag need not derive it again --
                #  ObjectAwareCrossAttention.condition_operands)
                view._lc_weights_only = tag
                out["image_patch_bbox_embedding_for_" + key] = view
        if "concat_cond" in condition_dict:
            cc = condition_dict["concat_cond"]
            if "autoregressive_cond" in condition_dict:
                cc = torch.cat([cc, condition_dict["autoregressive_cond"]], dim=1)
            out["concat_cond"] = cc
        return out

    def _core(self, boxes, obj_bbox_2d, is_valid_obj, obj_mask=None):
        """Everything that depends on the layout (reference layout_encoder.py:205-262): ~100 small launches on 13 tokens."""
        obj_bbox, obj_class = boxes[..., :8], boxes[..., -1]
        out, xf_in = {}, None
        if self.use_positional_embedding:
            xf_in = self.positional_embedding[None]
        if "obj_class" in self.used_condition_types:
            e = self.obj_class_embedding(obj_class.long())
            xf_in = e if xf_in is None else xf_in + e
            out["obj_class_embedding"] = e.permute(0, 2, 1).contiguous()
        if "obj_bbox" in self.used_condition_types:
            e3 = self.obj_bbox_embedding(obj_bbox.to(self.dtype))
            e2 = self.obj_bbox_2d_embedding(obj_bbox_2d.to(self.dtype))
            xf_in = e3 if xf_in is None else xf_in + e3 + e2
            out["obj_bbox_embedding"] = e2.permute(0, 2, 1).contiguous()
        if "obj_mask" in self.used_condition_types:
            m = self.obj_mask_embedding(obj_mask.view(*obj_mask.shape[:2], -1).to(self.dtype))
            xf_in = m if xf_in is None else xf_in + m
        if "is_valid_obj" in self.used_condition_types:
            out["key_padding_mask"] = (1 - is_valid_obj).bool()
        kpm = out["key_padding_mask"] if self.use_key_padding_mask else None
        xf_out = xf_in.to(self.dtype)
        if not self.not_use_layout_fusion_module:
            xf_out = self.transform(xf_out, kpm)
        if self.final_ln is not None:
            xf_out = self.final_ln(xf_out)
        out["xf_proj"] = self.transformer_proj(xf_out[:, 0])
        out["xf_out"] = xf_out.permute(0, 2, 1).contiguous()
        return out

    # The core is host-bound (1.8 ms for ~9 MFLOP) and runs once per `sample()` call while the GPU idles: inside a sampling
    # run (inference mode) its inputs go through static copies and, from the second call of a shape on, ONE replayed HIP
    # graph does the work; the results are handed out as fresh copies (a caller may hold several conditions at once).
    def _core_by_graph(self, boxes, obj_bbox_2d, is_valid_obj):
        if not (CORE_GRAPH and torch.is_inference_mode_enabled() and boxes.is_cuda and not self.use_key_padding_mask
                and "obj_mask" not in self.used_condition_types and not torch.cuda.is_current_stream_capturing()
                and not torch.is_autocast_enabled()):
            return None
        ins = (boxes, obj_bbox_2d, is_valid_obj)
        fp = []
        for t in self.parameters():
            fp.append((t.data_ptr(), t._version))
        sig = (tuple((tuple(t.shape), t.dtype, t.device) for t in ins), tuple(fp))
        st = self.__dict__.get("_core_graph")
        if st is None or st["sig"] != sig:
            st = dict(sig=sig, inp=[t.clone() for t in ins], graph=None, out=None, runs=0)
            self.__dict__["_core_graph"] = st
        else:
            for d, t in zip(st["inp"], ins):
                d.copy_(t)
        if st["graph"] is None and st["runs"] >= 1:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    st["out"] = self._core(*st["inp"])
                st["graph"] = g
            except Exception as e:                  # an optimisation only
                import warnings

                warnings.warn(f"HIP graph capture of the layout encoder failed ({e!r}); staying eager")
                st["graph"] = False
        if st["graph"]:
            st["graph"].replay()
            return {k: v.clone() for k, v in st["out"].items()}
        st["runs"] += 1
        return self._core(*st["inp"])

    def __deepcopy__(self, memo):
        import copy

        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_core_graph" else copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_core_graph"] = None
        return d

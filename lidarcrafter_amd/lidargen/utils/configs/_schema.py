"""Experiment configs: one pydantic dataclass per registry name of the reference
(lidargen/utils/configs/__init__.py:17-32).  The reference keeps one module per experiment with
five near-identical nested dataclasses each; here the sections are declared once and every
experiment is a small set of overrides.  Field names / defaults follow option_unet_nusc.py,
option_nusc_box_layout_v6.py and option_nusc_auto_reg_v2.py (the three configs on the hot path);
`cfg.<section>.<field>` access, `Cfg()` and `Cfg(**ckpt["cfg"])` behave as in the reference.
"""
from __future__ import annotations

import copy
from dataclasses import field
from typing import Any, Dict, Optional, Tuple

from pydantic.dataclasses import dataclass

_NUSC_CLASSES = ("car", "truck", "construction_vehicle", "bus", "trailer", "motorcycle",
                 "bicycle", "pedestrian")


@dataclass
class DiffusionConfig:
    num_training_steps: Optional[int] = None
    num_sampling_steps: int = 1024
    prediction_type: str = "eps"
    loss_type: str = "l2"
    noise_schedule: str = "cosine"
    timestep_type: str = "continuous"
    cond_mode: Optional[str] = None
    w_loss_weight: bool = False
    clip_sample: bool = True


@dataclass
class TrainingConfig:
    batch_size_train: int = 2
    batch_size_eval: int = 8
    num_workers: int = 4
    num_steps: int = 300_000
    steps_save_image: int = 5_000
    steps_save_model: int = 10_000
    gradient_accumulation_steps: int = 1
    lr: float = 1e-4
    lr_warmup_steps: int = 10_000
    adam_beta1: float = 0.9
    adam_beta2: float = 0.99
    adam_weight_decay: float = 0.0
    adam_epsilon: float = 1e-8
    ema_decay: float = 0.995
    ema_update_every: int = 10
    mixed_precision: str = "fp16"
    dynamo_backend: str = "inductor"
    output_dir: str = "logs/diffusion"
    seed: int = 0


@dataclass
class DataConfig:
    dataset: str = "nuscenes"
    task: str = "uncond"
    class_names: Tuple[str, ...] = _NUSC_CLASSES
    custom_collate_fn: bool = False
    data_root: str = "../data/nuscenes"
    pkl_path: str = "../data/infos/nuscenes_infos_lidargen_train.pkl"
    depth_format: str = "log_depth"
    scan_unfolding: bool = False
    projection: str = "spherical-1024"
    train_depth: bool = True
    train_reflectance: bool = True
    resolution: Tuple[int, int] = (32, 1024)
    min_depth: float = 1.45
    max_depth: float = 80.0
    fov_up: float = 10.0
    fov_down: float = -30.0
    delete_ground: bool = False
    split: str = "train"


@dataclass
class ModelConfig:
    architecture: str = "efficient_unet"
    params: Dict[str, Any] = field(default_factory=dict)


@dataclass
class ConditionModelConfig:
    architecture: str = "layout_encoder"
    params: Dict[str, Any] = field(default_factory=dict)


UNCOND_UNET = dict(base_channels=64, temb_channels=None, channel_multiplier=(1, 2, 4, 8),
                   num_residual_blocks=(3, 3, 3, 3), gn_num_groups=8, gn_eps=1e-6,
                   attn_num_heads=8, coords_encoding="fourier_features", ring=True)

LAYOUT_UNET_V1 = dict(image_size=32, use_fp16=False, use_scale_shift_norm=True, out_channels=2,
                      model_channels=64, encoder_channels=64, num_head_channels=32, num_heads=-1,
                      num_heads_upsample=-1, num_res_blocks=2, num_attention_blocks=1,
                      resblock_updown=True, attention_ds=[4, 8], channel_mult=[1, 2, 4, 8],
                      dropout=0.1, use_checkpoint=False,
                      use_positional_embedding_for_attention=True,
                      attention_block_type="ObjectAwareCrossAttention")

LAYOUT_ENCODER = dict(feature_map_size=[32, 1024],
                      used_condition_types=["obj_class", "obj_bbox", "is_valid_obj"],
                      layout_length=13, num_classes_for_layout_object=9,
                      mask_size_for_layout_object=32, hidden_dim=64, output_dim=256, num_layers=6,
                      num_heads=4, use_final_ln=True, use_positional_embedding=False,
                      not_use_layout_fusion_module=False, resolution_to_attention=[4, 8],
                      use_key_padding_mask=False, out_channels=10)


def _merge(base: dict, **over) -> dict:
    d = copy.deepcopy(base)
    d.update(over)
    return d


def make_config(cls_name: str, *, model_arch: str, model_params: dict,
                cond_arch: Optional[str] = None, cond_params: Optional[dict] = None,
                data: Optional[dict] = None, diffusion: Optional[dict] = None,
                training: Optional[dict] = None):
    """Build a pydantic dataclass `cls_name` with sections data/model/[condition_model]/
    diffusion/training (+ `resume`, read by inference.load_model_duffusion_training)."""
    data, diffusion, training = data or {}, diffusion or {}, training or {}
    ann: Dict[str, Any] = {}
    ns: Dict[str, Any] = {"__annotations__": ann, "__module__": __name__}

    def add(name, typ, factory):
        ann[name] = typ
        ns[name] = field(default_factory=factory)

    add("data", DataConfig, lambda: DataConfig(**copy.deepcopy(data)))
    add("model", ModelConfig,
        lambda: ModelConfig(architecture=model_arch, params=copy.deepcopy(model_params)))
    if cond_arch is not None:
        add("condition_model", ConditionModelConfig,
            lambda: ConditionModelConfig(architecture=cond_arch,
                                         params=copy.deepcopy(cond_params)))
    add("diffusion", DiffusionConfig, lambda: DiffusionConfig(**copy.deepcopy(diffusion)))
    add("training", TrainingConfig, lambda: TrainingConfig(**copy.deepcopy(training)))
    ann["resume"] = Optional[str]
    ns["resume"] = None
    return dataclass(type(cls_name, (), ns))


_LAYOUT_DATA = dict(task="layout_cond", custom_collate_fn=True)
_CONCAT = dict(cond_mode="concat")

NUSC_Config = make_config("NUSC_Config", model_arch="efficient_unet", model_params=UNCOND_UNET)


def _layout_cfg(name, *, enc_arch="layout_encoder", enc_over=None, unet_arch="layout_unet_v1",
                unet_over=None, data=None, diffusion=None, training=None):
    return make_config(name, model_arch=unet_arch,
                       model_params=_merge(LAYOUT_UNET_V1, **(unet_over or {})),
                       cond_arch=enc_arch, cond_params=_merge(LAYOUT_ENCODER, **(enc_over or {})),
                       data=_merge(_LAYOUT_DATA, **(data or {})), diffusion=diffusion or {},
                       training=training or {})


_T50 = dict(steps_save_model=50_000)
_no_oc = {k: v for k, v in LAYOUT_ENCODER.items() if k != "out_channels"}

NUSC_Box_Layout_Config = make_config(
    "NUSC_Box_Layout_Config", model_arch="layout_unet",
    model_params=_merge(LAYOUT_UNET_V1, model_channels=256, encoder_channels=256,
                        num_head_channels=64, attention_ds=[4], channel_mult=[1, 1, 2]),
    cond_arch="layout_encoder",
    cond_params=_merge(_no_oc, hidden_dim=256, output_dim=1024, num_heads=8,
                       resolution_to_attention=[1, 2, 4]),
    data=_LAYOUT_DATA)
NUSC_Box_Layout_V1_Config = make_config(
    "NUSC_Box_Layout_V1_Config", model_arch="layout_unet_v1", model_params=LAYOUT_UNET_V1,
    cond_arch="layout_encoder", cond_params=_no_oc, data=_LAYOUT_DATA, training=_T50)
NUSC_Box_Layout_V2_Config = _layout_cfg("NUSC_Box_Layout_V2_Config", diffusion=_CONCAT,
                                        training=_T50)
NUSC_Box_Layout_V3_Config = _layout_cfg("NUSC_Box_Layout_V3_Config", diffusion=_CONCAT,
                                        training=_T50)
NUSC_Box_Layout_V4_Config = _layout_cfg(
    "NUSC_Box_Layout_V4_Config", diffusion=dict(cond_mode="concat", w_loss_weight=True),
    training=dict(num_steps=500_000, steps_save_model=50_000))
NUSC_Box_Layout_V5_Config = _layout_cfg(
    "NUSC_Box_Layout_V5_Config", enc_arch="layout_encoder_v5",
    diffusion=dict(cond_mode="concat", w_loss_weight=True),
    training=dict(num_steps=500_000, steps_save_model=50_000))
NUSC_Box_Layout_V6_Config = _layout_cfg(
    "NUSC_Box_Layout_V6_Config", diffusion=_CONCAT, data=dict(delete_ground=True),
    training=dict(num_steps=500_000, steps_save_model=100_000))
NUSC_Auto_Reg_Config = _layout_cfg(
    "NUSC_Auto_Reg_Config", enc_over=dict(out_channels=12), diffusion=_CONCAT,
    data=dict(task="autoregressive_generation"),
    training=dict(num_steps=500_000, steps_save_model=50_000))
NUSC_Auto_Reg_V2_Config = _layout_cfg(
    "NUSC_Auto_Reg_V2_Config", enc_over=dict(out_channels=11), diffusion=_CONCAT,
    data=dict(task="autoregressive_generation"),
    training=dict(num_steps=500_000, steps_save_model=50_000))

# Registry names whose generators are OUT OF SCOPE (rows 3c, 4, 5, stale KITTI
# config).  They resolve to config objects; building their models raises NotImplementedError.
KITTI_Config_ = make_config("KITTI_Config_", model_arch="efficient_unet",
                            model_params=_merge(UNCOND_UNET, base_channels=128),
                            data=dict(dataset="kitti_360", resolution=(64, 1024),
                                      fov_up=3.0, fov_down=-25.0))
NUSC_HDIT_Config = make_config("NUSC_HDIT_Config", model_arch="hdit", model_params={})
MeanFlow_NUSC_Config = make_config("MeanFlow_NUSC_Config", model_arch="mf_efficient_unet",
                                   model_params={})
NUSC_Layout_Config = make_config("NUSC_Layout_Config", model_arch="unet_1d", model_params={},
                                 cond_arch="scene_graph", cond_params={},
                                 data=dict(task="layout_generation"))
# foreground-object branch (option_nusc_object.py): PointUNet over [1024, 4] object points
NUSC_Object_Config = make_config(
    "NUSC_Object_Config", model_arch="point_unet", model_params=dict(point_dim=4, cond_dims=768),
    cond_arch="object_gen_encoder", cond_params=dict(num_class=8),
    data=dict(task="object_generation", dataset="nuscenes-object", custom_collate_fn=True,
              pkl_path="../data/infos/nuscenes_dbinfos_10sweeps_withvelo.pkl"),
    diffusion=dict(clip_sample=False),
    training=dict(num_steps=1_000_000, steps_save_model=100_000))

"""Denoiser registry: the 14 names of the reference (lidargen/models/unets/__init__.py:15-30).
On the hot path: efficient_unet, layout_unet_v1, layout_encoder, and the foreground-object branch
of SURVEY.md §8f-3 (point_unet, object_gen_encoder).  Everything else is OUT OF SCOPE
 and resolves to a stub whose constructor says so."""
from .efficient_unet import EfficientUNet


def _stub(name, why):
    class _OutOfScope:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name}: {why} -- OUT OF SCOPE of the MI355X hot path "
                                      "(SURVEY.md §2)")
    _OutOfScope.__name__ = name
    return _OutOfScope


try:
    from .layout_encoder import LayoutTransformerEncoder
    from .layout_unet_v1 import LayoutUnetV1
except ImportError:  # pragma: no cover - only while the conditional path is being built
    LayoutTransformerEncoder = _stub("LayoutTransformerEncoder", "not built yet")
    LayoutUnetV1 = _stub("LayoutUnetV1", "not built yet")

LayoutTransformerEncoderV5 = _stub("LayoutTransformerEncoderV5", "CLIP-text box encoder variant")
LayoutUnet = _stub("LayoutUnet", "older variant of LayoutUnetV1 (no ring conv)")
EfficientUNetCond = _stub("EfficientUNetCond", "dict-style time-arg variant of EfficientUNet")
MFEfficientUNet = _stub("MFEfficientUNet", "MeanFlow generator (needs timm)")
UNet1DModel = _stub("UNet1DModel", "1-D layout generator")
SceneGraph = _stub("SceneGraph", "scene-graph GCN of the layout generator")
SpatialRescaler = _stub("SpatialRescaler", "LDM helper")
Identity = _stub("Identity", "LDM helper")
OpenAIUNetModel = _stub("OpenAIUNetModel", "LDM-style UNet")
from .encoders.object_gen_encoder import ObjectGenEncoder  # noqa: E402
from .point_unet import PointUNet  # noqa: E402

__all__ = {
    "layout_encoder": LayoutTransformerEncoder,
    "layout_encoder_v5": LayoutTransformerEncoderV5,
    "layout_unet": LayoutUnet,
    "layout_unet_v1": LayoutUnetV1,
    "efficient_unet": EfficientUNet,
    "efficient_unet_cond": EfficientUNetCond,
    "mf_efficient_unet": MFEfficientUNet,
    "unet_1d": UNet1DModel,
    "scene_graph": SceneGraph,
    "easy_unet": SpatialRescaler,
    "openai_unet": OpenAIUNetModel,
    "identity": Identity,
    "object_gen_encoder": ObjectGenEncoder,
    "point_unet": PointUNet,
}

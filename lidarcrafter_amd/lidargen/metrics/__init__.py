"""Weight-free evaluation front-end on the device (SURVEY.md section 8f-3 ii): `bev`.
The learned-feature metrics of the reference (FRID / FSVD / FPVD: RangeNet, MinkowskiNet, SPVCNN,
PTv3 backbones + checkpoints) are out of scope."""
from . import bev, chamfer  # noqa: F401

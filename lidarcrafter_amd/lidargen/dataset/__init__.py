"""Dataset registry.  The reference's nuScenes datasets need nuScenes on disk plus loguru / clip /
pyquaternion (OUT OF SCOPE); the names resolve so `from lidargen.dataset
import __all__` works, and constructing one says why it is unavailable.  `CustomDataset`
(user-supplied boxes / points, the item builder of the temporal loop) and the
projection every dataset calls per frame (transforms_3d.common.load_points_as_images) ARE on the
path and run on the GPU."""
from .custom_dataset import CustomDataset, CustomNuscObjectDataset  # noqa: F401


def _stub(name):
    class _Dataset:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"dataset {name!r} is OUT OF SCOPE of the MI355X hot path "
                                      "(needs nuScenes + absent dependencies, SURVEY.md §2-11)")
    _Dataset.__name__ = name
    return _Dataset


__all__ = {
    "nuscenes": _stub("NuscDataset"),
    "nuscenes-object": _stub("NuscObjectDataset"),
    "nuscenes-temporal": _stub("NuscTemporalDataset"),
    "custom": CustomDataset,
}

from .base import GaussianDiffusion
from .continuous_time import ContinuousTimeGaussianDiffusion
from .continuous_time_cond import CondContinuousTimeGaussianDiffusion
from .discrete_time import DiscreteTimeGaussianDiffusion


def _out_of_scope(name, why):
    class _Stub:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name}: {why} (SURVEY.md §2 marks it OUT OF SCOPE)")
    _Stub.__name__ = name
    return _Stub


# registry names of the reference (models/diffusion/__init__.py:1-6) must resolve
CondContinuousLayoutGaussianDiffusion = _out_of_scope(
    "CondContinuousLayoutGaussianDiffusion", "diffusion over per-object layout vectors")
from .continuous_time_1d_cond import CondContinuousLayoutGaussianDiffusion1D  # noqa: E402

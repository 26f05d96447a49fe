"""`lidargen.utils.common.to_device` of the reference (lidargen/utils/common.py:4-9): move every
tensor of a (nested) batch dict to `device`; the dict is updated IN PLACE and returned, values that
are neither tensors nor dicts (lists of names, numpy arrays, ints) are left alone."""
from __future__ import annotations

import torch


def to_device(data_dict: dict, device):
    moved = {k: (v.to(device) if isinstance(v, torch.Tensor) else to_device(v, device))
             for k, v in data_dict.items() if isinstance(v, (torch.Tensor, dict))}
    data_dict.update(moved)
    return data_dict

"""NeRF-style Fourier embedding of box parameters -- mirror of the reference's
lidargen/models/unets/encoders/embedder.py:5-57 (`Embedder`, `get_embedder`): the input followed
by sin / cos of the input times each frequency band, concatenated on the last axis
(out_dim = d * (1 + 2 * num_freqs) with include_input)."""
from __future__ import annotations

import torch


class Embedder:
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        k = self.kwargs
        d, n, top = k["input_dims"], k["num_freqs"], k["max_freq_log2"]
        if k["log_sampling"]:
            bands = 2.0 ** torch.linspace(0.0, top, steps=n)
        else:
            bands = torch.linspace(2.0 ** 0.0, 2.0 ** top, steps=n)
        self.include_input = bool(k["include_input"])
        self.freq_bands = bands
        self.periodic_fns = list(k["periodic_fns"])
        self.out_dim = d * (int(self.include_input) + len(bands) * len(self.periodic_fns))

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        parts = [inputs] if self.include_input else []
        for f in self.freq_bands.tolist():
            for fn in self.periodic_fns:
                parts.append(fn(inputs * f))
        return torch.cat(parts, -1)


def get_embedder(input_dims, num_freqs, include_input=True, log_sampling=True):
    return Embedder(input_dims=input_dims, num_freqs=num_freqs, max_freq_log2=num_freqs - 1,
                    include_input=include_input, log_sampling=log_sampling,
                    periodic_fns=[torch.sin, torch.cos])

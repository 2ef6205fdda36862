"""Continuous tokenizer -- host-side mirror of gato/policy/input_tokenizers.py:5-42.

``encode``/``decode`` keep the reference's semantics; on device tensors the binning runs in the
HIP kernel (neko_tokenize_continuous), which is the same code the fused packing kernel uses.
"""
import math

import torch

from .. import ops


def mu_law(tensor, mu=100, M=256):
    """input_tokenizers.py:5-6 (host helper; the device path fuses this into the packing kernel)."""
    return torch.sign(tensor) * torch.log(1 + mu * torch.abs(tensor)) / math.log(1 + mu * M)


class ContinuousTokenizer:
    def __init__(self, use_mu_law=True, mu=100, M=256, n_bins=1024, offset=None):
        self.use_mu_law = use_mu_law
        self.mu = mu
        self.M = M
        self.n_bins = n_bins
        self.offset = offset

    def encode(self, tensor):
        """input_tokenizers.py:17-30: optional mu-law, clamp [-1,1], (x+1)*(n_bins/2), int32, +offset."""
        if not tensor.is_cuda:
            raise RuntimeError("neko_amd.ContinuousTokenizer.encode needs a device tensor (no CPU path)")
        return ops.tokenize_continuous(tensor.to(torch.float32), self.use_mu_law, self.mu, self.M, self.n_bins,
                                       self.offset)

    def decode(self, tensor):
        """input_tokenizers.py:32-42."""
        if self.use_mu_law:
            raise Exception("mu-law encoding only expected with values which are not predicted")
        if self.offset is not None:
            tensor = tensor - self.offset
        return (2 * tensor) / self.n_bins - 1

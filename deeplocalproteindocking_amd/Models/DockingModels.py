"""Scoring model with the reference's plugin surface.

Surface kept from /root/reference/src/Models/DockingModels.py:23-84 so that checkpoints and callers
carry over: ``SimpleFilter(inputs_sizes)`` owns ``fc = [Linear(C, C//2), ReLU, Linear(C//2, 1)]``
(state-dict keys ``fc.0.*`` / ``fc.2.*``, Xavier-uniform weights); ``GlobalDockingModel`` exposes
``.representation``, ``.filter``, ``.threshold_clash``, ``forward(receptor_volumes, ligand_volumes)``
and ``save/load(directory, epoch)`` writing ``<name>_repr_epochN.th`` / ``<name>_filter_epochN.th``.

What differs is where the arithmetic runs: ``forward`` is one VolumeConvolution(clip) per resolution
(HIP K1+K2+K3) and ONE kernel for nearest-upsample + concat + MLP, instead of interpolate / cat /
three transposes / two GEMMs (DockingModels.py:74-83).
"""
import os

import torch
from torch import nn

from deeplocalproteindocking_amd.ops import VolumeConvolution, VolumeRotation, filter_volumes


def init_weights(module):
    """Xavier-uniform on every Linear / Conv3d weight (DockingModels.py:17-21)."""
    if isinstance(module, (nn.Linear, nn.Conv3d)):
        nn.init.xavier_uniform_(module.weight)


class SimpleFilter(nn.Module):
    """Per-translation scorer: concatenated correlation channels -> hidden (half as wide) -> score."""

    def __init__(self, inputs_sizes):
        super().__init__()
        width = int(sum(int(c) for c in inputs_sizes))
        hidden = width // 2
        self.fc_input_size = width
        self.fc = nn.Sequential(nn.Linear(width, hidden), nn.ReLU(), nn.Linear(hidden, 1))
        self.fc.apply(init_weights)

    def forward(self, input):
        return self.fc(input)

    def parameters_tuple(self):
        """(W1 (H,C), b1 (H), W2 (1,H), b2 (1)), detached, for the fused kernels."""
        first, last = self.fc[0], self.fc[2]
        return tuple(t.detach() for t in (first.weight, first.bias, last.weight, last.bias))


class GlobalDockingModel(nn.Module):
    FILES = ("%s_repr_epoch%d.th", "%s_filter_epoch%d.th")

    def __init__(self, representation, filter, threshold_clash=300, normalize=False, rotate_ligand=False,
                 exclude_clashes=True, clip=5.0):
        super().__init__()
        self.representation, self.filter = representation, filter
        self.threshold_clash, self.clip = threshold_clash, clip
        self.normalize, self.rotate_ligand, self.exclude = normalize, rotate_ligand, exclude_clashes
        self.convolve = VolumeConvolution(clip=clip)
        self.vol_rotate = VolumeRotation()

    def _paths(self, directory, epoch, model_name):
        return [os.path.join(directory, pattern % (model_name, epoch)) for pattern in self.FILES]

    def save(self, directory, epoch, model_name="DPD_Model"):
        for part, path in zip((self.representation, self.filter), self._paths(directory, epoch, model_name)):
            torch.save(part.state_dict(), path)

    def load(self, directory, epoch, model_name="DPD_Model"):
        for part, path in zip((self.representation, self.filter), self._paths(directory, epoch, model_name)):
            part.load_state_dict(torch.load(path))

    def forward(self, receptor_volumes, ligand_volumes):
        """Lists of (B, C_i, L_i, L_i, L_i) per resolution -> scores (B, 2L_0, 2L_0, 2L_0)."""
        correlations = [self.convolve(rec, lig) for rec, lig in zip(receptor_volumes, ligand_volumes)]
        W1, b1, W2, b2 = self.filter.parameters_tuple()
        return filter_volumes(correlations, W1, b1, W2, float(b2.reshape(-1)[0]))

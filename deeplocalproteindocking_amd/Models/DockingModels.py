"""Scoring model with the reference's plugin surface.

Mirrors /root/reference/src/Models/DockingModels.py:23-84: ``SimpleFilter`` (Linear(C,C/2) ->
ReLU -> Linear(C/2,1), Xavier-uniform weights) and ``GlobalDockingModel`` with
``.representation``, ``.filter``, ``.threshold_clash``, ``.forward(receptor_volumes,
ligand_volumes)``, ``.load/.save`` (two state_dict files per epoch).  ``forward`` runs on the HIP
ops: per-resolution VolumeConvolution(clip=5.0) and one fused upsample+concat+MLP kernel instead of
interpolate / cat / 3 transposes / 2 GEMMs (DockingModels.py:74-83).
"""
import os

import numpy as np
import torch
from torch import nn
from torch.nn.modules.module import Module

from deeplocalproteindocking_amd.ops import VolumeConvolution, VolumeRotation, filter_volumes


def init_weights(m):
    if type(m) == nn.Conv3d:
        torch.nn.init.xavier_uniform_(m.weight)
    if type(m) == nn.Linear:
        torch.nn.init.xavier_uniform_(m.weight)


class SimpleFilter(Module):
    def __init__(self, inputs_sizes):
        super(SimpleFilter, self).__init__()
        self.fc_input_size = int(np.sum(inputs_sizes))
        self.fc = nn.Sequential(
            nn.Linear(self.fc_input_size, int(self.fc_input_size / 2), bias=True),
            nn.ReLU(),
            nn.Linear(int(self.fc_input_size / 2), 1, bias=True),
        )
        self.fc.apply(init_weights)

    def forward(self, input):
        return self.fc(input)

    def parameters_tuple(self):
        """(W1 (H,C), b1 (H), W2 (1,H), b2 (1)) for the fused kernels."""
        return (self.fc[0].weight.detach(), self.fc[0].bias.detach(),
                self.fc[2].weight.detach(), self.fc[2].bias.detach())


class GlobalDockingModel(Module):
    def __init__(self, representation, filter, threshold_clash=300, normalize=False, rotate_ligand=False,
                 exclude_clashes=True, clip=5.0):
        super(GlobalDockingModel, self).__init__()
        self.threshold_clash = threshold_clash
        self.representation = representation
        self.filter = filter
        self.clip = clip
        self.convolve = VolumeConvolution(clip=clip)
        self.vol_rotate = VolumeRotation()
        self.rotate_ligand = rotate_ligand
        self.normalize = normalize
        self.exclude = exclude_clashes

    def save(self, directory, epoch, model_name="DPD_Model"):
        torch.save(self.representation.state_dict(), os.path.join(directory, '%s_repr_epoch%d.th' % (model_name, epoch)))
        torch.save(self.filter.state_dict(), os.path.join(directory, '%s_filter_epoch%d.th' % (model_name, epoch)))

    def load(self, directory, epoch, model_name="DPD_Model"):
        self.representation.load_state_dict(torch.load(os.path.join(directory, '%s_repr_epoch%d.th' % (model_name, epoch))))
        self.filter.load_state_dict(torch.load(os.path.join(directory, '%s_filter_epoch%d.th' % (model_name, epoch))))

    def forward(self, receptor_volumes, ligand_volumes):
        """lists of (B,C_i,L_i,L_i,L_i) -> (B,2L_0,2L_0,2L_0); DockingModels.py:63-84."""
        convolved = [self.convolve(r, l) for r, l in zip(receptor_volumes, ligand_volumes)]
        W1, b1, W2, b2 = self.filter.parameters_tuple()
        return filter_volumes(convolved, W1, b1, W2, float(b2.reshape(-1)[0]))

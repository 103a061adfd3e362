"""Representation plugins: only the SURFACE is part of the accelerated path
(``forward(volume (B,11,L,L,L)) -> [vol_res0, vol_res1]`` and ``get_num_outputs()``,
/root/reference/src/Models/ProteinRepresentationModels.py:69-76,123-129).  Any user nn.Module
with those two members plugs into GlobalDockingModel / Docker.

``SyntheticRepr`` produces seeded synthetic representation volumes for the BASELINE configs
that have no atoms.  ``E3MultiResRepr4x4`` restates the reference's plain-Conv3d plugin
(:85-114) in torch (it is a caller of the path, SURVEY.md 8(f) row 2, not a measured kernel).
``SE3MultiResReprScalar`` needs the third-party se3cnn package (unpinned, absent here).
"""
import torch
from torch import nn
from torch.nn.modules.module import Module


class SyntheticRepr(Module):
    """Seeded random smooth volumes: get_num_outputs() -> [C] or [C0, C1]; ignores the input
    density except for its batch size and box size."""

    def __init__(self, num_outputs=(48,), seed=0, amplitude=0.05):
        super().__init__()
        self.num_outputs = [int(c) for c in num_outputs]
        self.seed = seed
        self.amplitude = amplitude

    def get_num_outputs(self):
        return list(self.num_outputs)

    def make(self, L, tag, device="cpu"):
        g = torch.Generator().manual_seed(self.seed * 7919 + sum(ord(ch) for ch in tag))
        vols = []
        for i, c in enumerate(self.num_outputs):
            Li = L // (2 ** i)
            ar = (torch.arange(Li, dtype=torch.float32) - (Li - 1) / 2.0) / (Li / 2.0)
            r2 = ar[:, None, None] ** 2 + ar[None, :, None] ** 2 + ar[None, None, :] ** 2
            env = torch.exp(-1.5 * r2)
            v = torch.randn(1, c, Li, Li, Li, generator=g) * self.amplitude * env
            vols.append(v.to(device))
        return vols

    def forward(self, volume):
        B, L = volume.shape[0], volume.shape[2]
        return [v.repeat(B, 1, 1, 1, 1) for v in self.make(L, "fwd", volume.device)]


class E3MultiResRepr4x4(Module):
    def __init__(self, num_input_channels=11, multiplier=16):
        super(E3MultiResRepr4x4, self).__init__()
        m = multiplier
        self.num_outputs_res0 = m * 2
        self.num_outputs_res1 = m * 4
        self.conv1 = nn.Sequential(
            nn.Conv3d(num_input_channels, m * 2, kernel_size=5, padding=2, bias=False), nn.ReLU(),
            nn.Conv3d(m * 2, m * 2, kernel_size=3, padding=1, bias=False), nn.ReLU(),
            nn.Conv3d(m * 2, m * 2, kernel_size=3, padding=1, bias=False), nn.ReLU(),
            nn.Conv3d(m * 2, m * 2, kernel_size=3, padding=1, bias=False), nn.ReLU(),
            nn.Conv3d(m * 2, m * 2, kernel_size=5, padding=2, bias=False))
        self.conv2 = nn.Sequential(
            torch.nn.MaxPool3d(kernel_size=5, stride=2, padding=2),
            nn.Conv3d(m * 2, m * 4, kernel_size=5, padding=2, bias=False), nn.ReLU(),
            nn.Conv3d(m * 4, m * 4, kernel_size=3, padding=1, bias=False), nn.ReLU(),
            nn.Conv3d(m * 4, m * 4, kernel_size=3, padding=1, bias=False), nn.ReLU(),
            nn.Conv3d(m * 4, m * 4, kernel_size=3, padding=1, bias=False))

    def get_num_outputs(self):
        return [self.num_outputs_res0, self.num_outputs_res1]

    def forward(self, volume):
        vol1 = self.conv1(volume)
        vol2 = self.conv2(vol1)
        return [vol1, vol2]


class SE3MultiResReprScalar(Module):
    def __init__(self, num_input_channels=11, multiplier=16):
        super(SE3MultiResReprScalar, self).__init__()
        try:
            import se3cnn  # noqa: F401
        except Exception as e:
            raise Exception("SE3MultiResReprScalar needs the se3cnn package (reference README.md:6), "
                            "which is not part of this build; plug in any module with forward()/get_num_outputs()", e)
        raise Exception("SE3MultiResReprScalar: se3cnn binding is SURVEY.md 8(f) row 2 (next)")

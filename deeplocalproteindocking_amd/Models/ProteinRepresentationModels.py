"""Representation plugins: only the SURFACE is part of the accelerated path
(``forward(volume (B,11,L,L,L)) -> [vol_res0, vol_res1]`` and ``get_num_outputs()``,
/root/reference/src/Models/ProteinRepresentationModels.py:69-76,123-129).  Any user nn.Module
with those two members plugs into GlobalDockingModel / Docker.

``SyntheticRepr`` produces seeded synthetic representation volumes for the BASELINE configs
that have no atoms.  ``E3MultiResRepr4x4`` is the reference's plain-Conv3d plugin (:85-114, same
state-dict keys).  ``SE3MultiResReprScalar`` keeps the reference's layer plan with build-defined
isotropic kernels (the third-party se3cnn package is unpinned and absent here).

Where the convolutions run: on a GPU, in inference (no grad, float32), EVERY Conv3d / max-pool of
both plugins runs on the HIP kernels (``ops.conv3d`` -- f32 matrix cores, stride 1 and 2 --,
``ops.maxpool3d_5s2``).  A layer shape those kernels do not cover is an error there, not a silent
switch to torch/MIOpen: set ``DLPD_ALLOW_TORCH_CONV=1`` to allow it (a warning says which layer).
On CPU tensors and under autograd the modules are plain torch (the reference implementation the
tests compare against; training is outside this build).

Loading reference checkpoints: ``E3MultiResRepr4x4`` loads ``*_repr_epochN.th`` of the reference
as they are (identical module tree: ``conv1.{0,2,4,6,8}.weight``, ``conv2.{1,3,5,7}.weight``).
``SE3MultiResReprScalar`` cannot: se3cnn stores, per layer, coefficients on ITS radial basis, and
neither that basis nor its version is in the reference tree.  ``SE3MultiResReprScalar.load_dense_kernels`` is the
hand-over for a user who has se3cnn: the eight dense (cout, cin, 5, 5, 5) kernels it evaluates, by the reference's own
module names (``dense_kernel_contract``), used as they are -- the scalar-field SE3Convolution is a plain conv3d with that
kernel, so the representation then IS the reference's.
"""
import os
import warnings

import torch
from torch import nn
from torch.nn.modules.module import Module


def _torch_conv_allowed(what):
    """GPU inference reached a layer the HIP kernels do not cover: loud by default."""
    if os.environ.get("DLPD_ALLOW_TORCH_CONV", "") == "1":
        warnings.warn("dlpd: %s has no HIP kernel -- running it on torch/MIOpen (DLPD_ALLOW_TORCH_CONV=1)" % what)
        return True
    raise RuntimeError("dlpd: %s has no HIP kernel (supported: cubic float32 volumes up to 80^3, kernel 3 or 5 with "
                       "padding k//2, stride 1 or 2, no bias, output channels a multiple of 16; MaxPool3d(5, 2, 2)). "
                       "Set DLPD_ALLOW_TORCH_CONV=1 to run it on torch/MIOpen instead." % what)


def _hip_inference(x, hip_lib):
    """The HIP kernels are the path: GPU tensors (or the emulated library of the test-suite), no autograd."""
    return (x.is_cuda or hip_lib is not None) and not torch.is_grad_enabled() and x.dtype == torch.float32


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

    # lib: None -> the product library on a GPU; tests pass the emulated one
    hip_lib = None
    use_hip_conv = True
    # skip the all-zero tiles of the bias-free convolutions (same bits; ops.conv3d); DLPD_TILE_OCCUPANCY=0: compute everywhere
    use_tile_occupancy = os.environ.get("DLPD_TILE_OCCUPANCY", "1") != "0"
    # Unwritten activations (round 6): inside ``with repr.outputs_with_maps():`` the tile-occupancy layers do not WRITE the
    # tiles they skip either (that was most of the plugin's time per dockE3 batch) and never read a cell their input's map
    # marks empty; the returned volumes then carry their maps (``v.dlpd_occupancy``, uint8 (B, ceil(D/4)^3)) and are
    # undefined where the map is 0 -- only for a consumer that goes by the map (the fused engine's volumes path,
    # DockingEngine.step(occupancy=...); Docker._dockE3_fused is the one caller).  Outside the context every voxel is written.
    supports_unwritten_outputs = True
    _unwritten = False

    def outputs_with_maps(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._unwritten = self._unwritten, True
            try:
                yield self
            finally:
                self._unwritten = old
        return ctx()

    def _run(self, seq, x, occ=None, return_occupancy=False, unwritten=False):
        """The Sequential.  GPU inference: Conv3d(+ReLU) pairs and the max-pool on the HIP kernels (exact f32 on
        the matrix cores), anything they cannot take is an error unless DLPD_ALLOW_TORCH_CONV=1; CPU / autograd:
        plain torch."""
        from deeplocalproteindocking_amd import ops
        mods = list(seq)
        native = self.use_hip_conv and _hip_inference(x, self.hip_lib)
        # tile occupancy of x (ops.conv3d): the layers have no bias, so what lies outside the protein's neighbourhood stays
        # exactly zero from layer to layer and is not computed; one map is made of the input, every convolution writes its
        # output's and so does the pooling (None: unknown -- after a torch module -- and made again when a layer needs it);
        # ``occ`` = the map of x if the caller has it (the second Sequential takes the first one's)
        sparse = native and self.use_tile_occupancy and ops.CONV_PRECISION == "split_bf16"
        unwritten = bool(unwritten and sparse)
        i = 0
        while i < len(mods):
            m = mods[i]
            cubic = x.dim() == 5 and x.shape[2] == x.shape[3] == x.shape[4]
            if native and isinstance(m, nn.Conv3d):
                ok = (m.bias is None and m.stride in ((1, 1, 1), (2, 2, 2)) and m.dilation == (1, 1, 1) and m.groups == 1
                      and m.padding == tuple(k // 2 for k in m.kernel_size) and cubic
                      and ops.conv3d_supported(m.weight, x.shape[2], self.hip_lib))
                if ok or not _torch_conv_allowed("Conv3d%s on %s" % (tuple(m.weight.shape), tuple(x.shape))):
                    relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                    if sparse:
                        if occ is None:
                            occ = ops.tile_occupancy(x, lib=self.hip_lib)
                        x, occ = ops.conv3d(x, m.weight, relu=relu, lib=self.hip_lib, stride=m.stride[0], occupancy=occ,
                                            return_occupancy=True, unwritten=unwritten and m.stride[0] == 1)
                    else:
                        x = ops.conv3d(x, m.weight, relu=relu, lib=self.hip_lib, stride=m.stride[0])
                    i += 2 if relu else 1
                    continue
            elif native and isinstance(m, nn.MaxPool3d):
                ok = (m.kernel_size == 5 and m.stride == 2 and m.padding == 2 and m.dilation == 1 and not m.ceil_mode
                      and not m.return_indices and cubic)
                if ok or not _torch_conv_allowed("MaxPool3d(%s, %s, %s)" % (m.kernel_size, m.stride, m.padding)):
                    if sparse:
                        if occ is None:
                            occ = ops.tile_occupancy(x, lib=self.hip_lib)
                        x, occ = ops.maxpool3d_5s2(x, lib=self.hip_lib, occupancy=occ, return_occupancy=True, unwritten=unwritten)
                    else:
                        x = ops.maxpool3d_5s2(x, lib=self.hip_lib)
                    i += 1
                    continue
            x = m(x)
            occ = None
            i += 1
        return (x, occ if sparse else None) if return_occupancy else x

    def forward(self, volume):
        uw = self._unwritten
        # an input that comes with its map (CoordsBackend.project(cells=True)): the first layer takes the map instead of
        # making one, and -- the input being unwritten outside its cells -- must go by it
        occ0 = getattr(volume, "dlpd_occupancy", None)
        from deeplocalproteindocking_amd import ops
        if getattr(volume, "dlpd_unwritten", False) and not (uw and self.use_hip_conv and self.use_tile_occupancy and
                                                              ops.CONV_PRECISION == "split_bf16" and
                                                              _hip_inference(volume, self.hip_lib)):
            raise RuntimeError("dlpd: a cell-wise projected volume (unwritten outside its occupied cells) can only enter the plugin "
                               "inside outputs_with_maps() on the tile-occupancy kernels")
        vol1, occ1 = self._run(self.conv1, volume, occ0, True, unwritten=uw)
        vol2, occ2 = self._run(self.conv2, vol1, occ1, True, unwritten=uw and occ1 is not None)
        if occ1 is not None:                       # (tile-occupancy path: the maps travel with the volumes)
            vol1.dlpd_occupancy = occ1
        if occ2 is not None:
            vol2.dlpd_occupancy = occ2
        return [vol1, vol2]


class IsotropicConv3d(Module):
    """Scalar-field (l = 0 -> l = 0) SE(3)-equivariant convolution: every (out, in) kernel is a
    radial function, K(r) = sum_k w[out, in, k] * phi_k(|r|), phi_k Gaussian shells at radii
    0 .. size//2.  That is the function class of se3cnn's SE3Convolution([(n,0)], [(m,0)], size)
    used by the reference (ProteinRepresentationModels.py:38-61); se3cnn itself is absent, so the
    shell profile (sigma 0.6) and the initialisation are build-defined (parity unpinned)."""

    def __init__(self, cin, cout, size=5, padding=2, stride=1, sigma=0.6):
        super().__init__()
        self.padding, self.stride = padding, stride
        nr = size // 2 + 1
        ax = torch.arange(size, dtype=torch.float32) - (size - 1) / 2.0
        r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
        shells = torch.stack([torch.exp(-0.5 * ((r - k) / sigma) ** 2) for k in range(nr)])
        shells = shells * (r <= size // 2 + 0.5)                      # spherical support
        self.register_buffer("shells", shells / shells.flatten(1).norm(dim=1)[:, None, None, None])
        # se3cnn's dense kernel once load_dense_kernel(exact=True) has run.  A NON-persistent attribute, kept out of the
        # state dict on purpose: checkpoints stay loadable into a freshly constructed model (a strict load would reject
        # an unexpected 'dense' key), the shell coefficients in ``weight`` always hold the kernel's projection, and a later
        # load_state_dict -- new shell coefficients -- drops the dense kernel instead of being silently ignored.
        self.dense = None
        self.weight = nn.Parameter(torch.empty(cout, cin, nr))
        # He-style scale for inputs that vary slowly across the 5^3 window (atom densities are smooth
        # and non-negative): the gain that matters is the kernel SUM, not its norm; without this the
        # eight bias-free layers amplify the mean by orders of magnitude each
        dc2 = float((self.shells.flatten(1).sum(dim=1) ** 2).sum())
        nn.init.normal_(self.weight, std=(2.0 / (cin * dc2)) ** 0.5)

    def kernel(self):
        if self.dense is not None:                 # se3cnn's own kernel, handed over as it is (load_dense_kernel)
            if self.dense.device != self.weight.device or self.dense.dtype != self.weight.dtype:
                self.dense = self.dense.to(device=self.weight.device, dtype=self.weight.dtype)   # follows .to() / .half()
            return self.dense
        return torch.einsum("oik,kxyz->oixyz", self.weight, self.shells)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # reloaded shell coefficients replace a handed-over dense kernel -- loudly: the layer goes back to the shell
        # PROJECTION of whatever is loaded, which is not se3cnn's kernel (non-zero residual)
        stale = state_dict.pop(prefix + "dense", None)   # (checkpoints written while 'dense' was a registered buffer)
        if self.dense is not None or stale is not None:
            warnings.warn("dlpd: %sweight loaded from a state dict: the exact dense kernel of this layer (load_dense_kernel) is "
                          "dropped and the layer uses its radial-shell projection again -- call load_dense_kernels after load()"
                          % prefix)
        self.dense = None
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def load_dense_kernel(self, dense_kernel, exact=True):
        """The dense (cout, cin, size, size, size) kernel se3cnn evaluates for this layer.  exact: it is used AS IS from now
        on (the layer stops being parametrised by its own shells -- inference with reference weights); otherwise it is
        projected onto the shells (load_radial_profile).  Returns the relative residual of that projection either way."""
        K = torch.as_tensor(dense_kernel, dtype=torch.float32)
        want = (self.weight.shape[0], self.weight.shape[1]) + tuple(self.shells.shape[1:])
        if tuple(K.shape) != want:
            raise Exception("Dense kernel shape mismatch", tuple(K.shape), want)
        res = self.load_radial_profile(K)
        if exact:
            self.dense = K.to(self.weight.device).contiguous()
        return res

    hip_lib = None                     # tests: the emulated library

    def load_radial_profile(self, dense_kernels):
        """For users who have se3cnn: its SE3Convolution([(cin, 0)], [(cout, 0)], size=5) of a reference checkpoint
        (``sequence_res{0,1}.{0,2,4,6}`` of ``*_repr_epochN.th``, ProteinRepresentationModels.py:38-61) evaluates to a
        dense (cout, cin, 5, 5, 5) kernel (``module.kernel()`` there).  Pass that tensor: it is projected onto this
        layer's radial shells by least squares; the residual (returned, relative) says how far se3cnn's radial basis
        is from the Gaussian shells used here -- 0 for any kernel that is a function of |r| sampled on the shells."""
        K = torch.as_tensor(dense_kernels, dtype=torch.float32)
        A = self.shells.flatten(1).t()                                   # (125, nr)
        sol = torch.linalg.lstsq(A, K.flatten(2).permute(2, 0, 1).reshape(A.shape[0], -1)).solution
        w = sol.reshape(self.shells.shape[0], K.shape[0], K.shape[1]).permute(1, 2, 0).contiguous()
        with torch.no_grad():
            self.weight.copy_(w)
        return float((torch.einsum("oik,kxyz->oixyz", w, self.shells) - K).norm() / K.norm().clamp_min(1e-30))

    def forward(self, x):
        from deeplocalproteindocking_amd import ops
        k = self.kernel()
        if _hip_inference(x, self.hip_lib):
            ok = (self.stride in (1, 2) and self.padding == k.shape[2] // 2 and x.dim() == 5
                  and x.shape[2] == x.shape[3] == x.shape[4] and ops.conv3d_supported(k, x.shape[2], self.hip_lib))
            if ok or not _torch_conv_allowed("IsotropicConv3d%s stride %d on %s" % (tuple(k.shape), self.stride, tuple(x.shape))):
                return ops.conv3d(x, k, lib=self.hip_lib, stride=self.stride)      # f32 matrix cores
        return nn.functional.conv3d(x, k, padding=self.padding, stride=self.stride)


class SE3MultiResReprScalar(Module):
    """Same layer plan as the reference's SE3MultiResReprScalar (ProteinRepresentationModels.py:23-76):
    four isotropic 5^3 convolutions with ReLU between them at full resolution, then four more whose
    first has stride 2; outputs [2m @ L^3, 4m @ (L/2)^3]."""

    def __init__(self, num_input_channels=11, multiplier=16):
        super().__init__()
        m = multiplier
        self.num_outputs_res0, self.num_outputs_res1 = 2 * m, 4 * m

        def stack(cin, c, stride):
            return nn.Sequential(IsotropicConv3d(cin, c, stride=stride), nn.ReLU(),
                                 IsotropicConv3d(c, c), nn.ReLU(),
                                 IsotropicConv3d(c, c), nn.ReLU(),
                                 IsotropicConv3d(c, c))
        self.sequence_res0 = stack(num_input_channels, 2 * m, 1)
        self.sequence_res1 = stack(2 * m, 4 * m, 2)

    def get_num_outputs(self):
        return [self.num_outputs_res0, self.num_outputs_res1]

    # ---- hand-over of a reference checkpoint by users who have se3cnn --------------------------------------------
    def dense_kernel_contract(self):
        """{tensor name: shape} of the dense kernels ``load_dense_kernels`` expects: one per SE3Convolution of the
        reference module tree (ProteinRepresentationModels.py:38-61: ``sequence_res0.{0,2,4,6}``,
        ``sequence_res1.{0,2,4,6}``; the odd indices are the ScalarActivations, which have no parameters with
        ``bias=False``), each (cout, cin, 5, 5, 5) -- what ``SE3Convolution.kernel()`` returns there for scalar
        (l = 0) fields, where it is the convolution weight passed to ``conv3d`` (stride 2 on ``sequence_res1.0``)."""
        out = {}
        for seq_name in ("sequence_res0", "sequence_res1"):
            for i, m in enumerate(getattr(self, seq_name)):
                if isinstance(m, IsotropicConv3d):
                    out["%s.%d" % (seq_name, i)] = (m.weight.shape[0], m.weight.shape[1]) + tuple(m.shells.shape[1:])
        return out

    def load_dense_kernels(self, kernels, exact=True):
        """kernels: {name: tensor} exactly as ``dense_kernel_contract()`` lists them (missing / unknown names and wrong
        shapes raise).  With se3cnn installed next to the reference:
            ref = SE3MultiResReprScalar(multiplier=8); ref.load_state_dict(torch.load('DPD_Model_repr_epoch299.th'))
            kernels = {'%s.%d' % (s, i): getattr(ref, s)[i].kernel().detach() for s in ('sequence_res0', 'sequence_res1')
                       for i in (0, 2, 4, 6)}
        Returns {name: relative residual of the projection onto this build's radial shells} (exact=True keeps the dense
        kernels themselves, so the residual is only a diagnostic then)."""
        want = self.dense_kernel_contract()
        if set(kernels) != set(want):
            raise Exception("Dense kernel names do not match the contract", sorted(set(kernels) ^ set(want)))
        res = {}
        for name, shape in want.items():
            seq_name, idx = name.split(".")
            res[name] = getattr(self, seq_name)[int(idx)].load_dense_kernel(kernels[name], exact=exact)
        return res

    def forward(self, volume):
        vol1 = self.sequence_res0(volume)
        return [vol1, self.sequence_res1(vol1)]

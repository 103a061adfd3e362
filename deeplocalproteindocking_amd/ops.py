"""torch-facing operator objects with the call signatures the reference uses for
TorchProteinLibrary's volume ops, backed by libdlpd.so.

    VolumeRotation()(volume (B,C,L,L,L) f32 cuda, R (B,3,3) f32 cuda) -> (B,C,L,L,L)
        reference call: src/Docker/Docker.py:40,218
    VolumeConvolution(clip=None)(v1 (B,C,L,L,L), v2 same) -> (B,C,2L,2L,2L)
        reference calls: src/Docker/Docker.py:32,225 ; src/Models/DockingModels.py:48,71

Inference only (the docking search runs under torch.no_grad(), local_test.py:67).
Box sizes 32 / 40 / 64 / 80 run the compiled FFT pipeline, any other box (<= 128) a plan-free slow path.
Build-defined conventions (TPL source absent, parity unpinned): rotation about index L/2 with
trilinear interpolation and zeros outside; ``clip`` clamps the correlation OUTPUT to +-clip.
"""
import torch
from torch import nn

from ._lib import get_lib
from .engine import _ptr, _stream
from .Utils.Conventions import CLIP_MODES, kernel_matrices, rotation_scale


def _check(t, name, lib=None):
    """lib: None -> the product library, GPU tensors only (no CPU path); the test-suite passes the
    emulated library, which takes host pointers."""
    if not (isinstance(t, torch.Tensor) and (t.is_cuda or lib is not None) and t.dtype == torch.float32):
        raise RuntimeError("dlpd: %s must be a float32 CUDA (ROCm) tensor; there is no CPU path" % name)
    return t.contiguous()


class VolumeRotation(nn.Module):
    """``center`` / ``scale`` / ``axis_order``: the conventions of Utils/Conventions.py (pivot index; stretch of the
    sample offset -- a number or "(L-1)/L" / "L/(L-1)"; "xyz" | "zyx"), folded into the 3x3 maps the kernel samples with."""

    def __init__(self, center=None, lib=None, scale=None, axis_order="xyz", transpose=False):
        super().__init__()
        self.center = center
        self.lib = lib
        self.scale = scale
        self.axis_order = axis_order
        self.transpose = bool(transpose)

    def forward(self, volume, R):
        volume, R = _check(volume, "volume", self.lib), _check(R, "R", self.lib)
        B, C, L = volume.shape[0], volume.shape[1], volume.shape[2]
        if R.shape[0] != B:
            raise RuntimeError("dlpd: VolumeRotation batch mismatch: volume %d vs R %d" % (B, R.shape[0]))
        out = torch.empty_like(volume)
        c0 = float(L) / 2.0 if self.center is None else float(self.center)
        if self.scale is not None or self.axis_order != "xyz" or self.transpose:
            R = kernel_matrices(R, rotation_scale(self.scale, L), self.axis_order, self.transpose)
        (self.lib or get_lib()).call("dlpd_rotate_trilinear", _ptr(volume), _ptr(R), _ptr(out), B, C, L, C * L ** 3, c0,
                       _stream(volume.device))
        return out


class VolumeConvolution(nn.Module):
    """Per-channel circular cross-correlation on the 2L zero-padded grid:
    out[b,c,t mod 2L] = sum_r v1[b,c,r+t] * v2[b,c,r]  (semantics: MultiplyVolumes.py:13-47)."""

    def __init__(self, clip=None, lib=None, embed=True, clip_mode="output"):
        """clip_mode (Utils/Conventions.py): "output" clamps the correlation to +-clip (this build's definition),
        "input" clamps both input volumes instead, "none" ignores ``clip``."""
        super().__init__()
        if clip_mode not in CLIP_MODES:
            raise RuntimeError("dlpd: clip_mode must be one of %s" % (CLIP_MODES,))
        self.clip_mode = clip_mode
        self.clip = clip
        self.lib = lib
        self.embed = embed          # boxes without a compiled plan: inside the next compiled box (False: plan-free transforms)

    @property
    def out_clip(self):
        """The clamp the kernels apply to the correlation OUTPUT (None: no clamp)."""
        return self.clip if self.clip_mode == "output" else None

    def forward(self, input_volume1, input_volume2):
        v1, v2 = _check(input_volume1, "volume1", self.lib), _check(input_volume2, "volume2", self.lib)
        if v1.shape != v2.shape:
            raise RuntimeError("dlpd: VolumeConvolution shape mismatch %s vs %s" % (tuple(v1.shape), tuple(v2.shape)))
        if self.clip is not None and self.clip_mode == "input":
            c = float(self.clip)
            v1, v2 = v1.clamp(-c, c), v2.clamp(-c, c)
        B, C, L = v1.shape[0], v1.shape[1], v1.shape[2]
        lib = self.lib or get_lib()
        if not lib.call("dlpd_grid_supported", L):
            Lc = next((c for c in (32, 40, 64, 80) if c > L and lib.call("dlpd_grid_supported", c)), None) if self.embed else None
            if Lc is None:
                return self._forward_generic(v1, v2, lib)
            # the L^3 volumes in the corner of the next compiled box: the correlation of two L-sized volumes is linear
            # for every |t| < L on any grid of >= 2L points, so the (2L)^3 result sits inside the (2Lc)^3 one at index t
            # (0 <= t <= L; t = L: no overlap, zero) and 2Lc + t (-L < t < 0)
            e1, e2 = (torch.zeros(B, C, Lc, Lc, Lc, dtype=torch.float32, device=v1.device) for _ in range(2))
            e1[:, :, :L, :L, :L], e2[:, :, :L, :L, :L] = v1, v2
            big = self.forward(e1, e2)
            idx = torch.tensor(list(range(0, L + 1)) + list(range(2 * Lc - (L - 1), 2 * Lc)), dtype=torch.long, device=v1.device)
            return big.index_select(2, idx).index_select(3, idx).index_select(4, idx).contiguous()
        N, NZ, nvol = 2 * L, L + 1, B * C
        dev, st = v1.device, _stream(v1.device)
        wsA = torch.empty(nvol * NZ * L * L * 2, dtype=torch.float32, device=dev)
        spec = torch.empty(nvol * NZ * N * N * 2, dtype=torch.float32, device=dev)
        lib.call("dlpd_rfft3d_padded", _ptr(v1), _ptr(spec), _ptr(wsA), nvol, L, 1.0 / float(N) ** 3, st)
        lib.call("dlpd_zfft", _ptr(v2), 0, _ptr(wsA), 1, nvol, L, 0, 0, 0.0, st)
        wsB = torch.empty(nvol * NZ * N * N * 2, dtype=torch.float32, device=dev)
        lib.call("dlpd_xy_correlate", _ptr(wsA), _ptr(spec), _ptr(wsB), 1, nvol, L, 0, st)
        out = torch.empty(B, C, N, N, N, dtype=torch.float32, device=dev)
        oc = self.out_clip
        lib.call("dlpd_zifft_real", _ptr(wsB), _ptr(out), 1, nvol, L, 0 if oc is None else 1, float(oc or 0.0), st)
        return out


def _vc_generic(self, v1, v2, lib):
    """Any other box size (the reference's ``box_size`` is free, Docker.py:18): the plan-free correlation of
    dlpd_correlate_generic, in chunks of volumes that keep the scratch below ~4 GB."""
    B, C, L = v1.shape[0], v1.shape[1], v1.shape[2]
    if not lib.call("dlpd_generic_box_supported", L):
        raise RuntimeError("dlpd: VolumeConvolution box size %d exceeds the generic path (box <= 128)" % L)
    N, nvol = 2 * L, B * C
    dev, st = v1.device, _stream(v1.device)
    out = torch.empty(B, C, N, N, N, dtype=torch.float32, device=dev)
    per = lib.call("dlpd_correlate_generic_ws_bytes", 1, L)
    chunk = max(1, min(nvol, (4 << 30) // per, 65535 // N))
    ws = torch.empty(per * chunk, dtype=torch.uint8, device=dev)
    a, b, o = v1.reshape(nvol, -1), v2.reshape(nvol, -1), out.reshape(nvol, -1)
    for beg in range(0, nvol, chunk):
        n = min(chunk, nvol - beg)
        lib.call("dlpd_correlate_generic", _ptr(a[beg]), _ptr(b[beg]), _ptr(o[beg]), n, L, 0 if self.out_clip is None else 1,
                 float(self.out_clip or 0.0), _ptr(ws), st)
    return out


VolumeConvolution._forward_generic = _vc_generic


def filter_volumes(conv_list, W1, b1, W2, b2, mask_norm=None, threshold=0.0, lib=None):
    """Nearest-upsample + concat + SimpleFilter MLP (+ optional clash mask) on materialised
    correlation volumes -- DockingModels.py:74-83, Docker.py:226,232.  conv_list: one or two
    tensors (B,C_i,N_i,N_i,N_i), N_0 the finest."""
    c0 = _check(conv_list[0], "conv0", lib)
    B, C0, N0 = c0.shape[0], c0.shape[1], c0.shape[2]
    if len(conv_list) > 2:
        raise RuntimeError("dlpd: at most two resolutions are supported")
    c1 = _check(conv_list[1], "conv1", lib) if len(conv_list) == 2 else None
    C1, N1 = (c1.shape[1], c1.shape[2]) if c1 is not None else (0, 0)
    dev = c0.device
    H = W1.shape[0]
    W1t = W1.detach().to(dev, torch.float32).t().contiguous()       # (C, H)
    b1 = b1.detach().to(dev, torch.float32).contiguous()
    W2 = W2.detach().to(dev, torch.float32).reshape(-1).contiguous()
    V = torch.empty(B, N0, N0, N0, dtype=torch.float32, device=dev)
    has_clash = mask_norm is not None
    if has_clash:
        mask_norm = _check(mask_norm, "mask_norm", lib)
    (lib or get_lib()).call("dlpd_filter_mask", _ptr(c0), C0, N0, _ptr(c1), C1, N1, _ptr(mask_norm), float(threshold),
                   int(has_clash), _ptr(W1t), _ptr(b1), _ptr(W2), float(b2), H, _ptr(V), B, _stream(dev))
    return V


def conv3d_supported(weight, D, lib=None):
    cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
    cubic = weight.dim() == 5 and weight.shape[2] == weight.shape[3] == weight.shape[4]
    return bool(cubic and (lib or get_lib()).call("dlpd_conv3d_supported", int(cin), int(cout), int(ks), int(D)))


CONV_PRECISION = "split_bf16"      # default arithmetic of conv3d: "split_bf16" (3 x bf16 terms, six products: f32-grade) | "f32"


def tile_occupancy(x, lib=None):
    """Which 4 x 4 x 4 cells of x (B, C, D, D, D) hold a non-zero value: uint8 (B, ceil(D/4), ceil(D/4), ceil(D/4)), the
    ``occupancy`` argument of conv3d."""
    lib = lib or get_lib()
    x = x.contiguous()
    B, cin, D = x.shape[0], x.shape[1], x.shape[2]
    occ = torch.empty(B, (D + 3) // 4, (D + 3) // 4, (D + 3) // 4, dtype=torch.uint8, device=x.device)
    assert occ.numel() == lib.call("dlpd_conv3d_tile_occupancy_bytes", B, D)
    lib.call("dlpd_conv3d_tile_occupancy", _ptr(x), _ptr(occ), B, cin, D, _stream(x.device))
    return occ


def conv3d(x, weight, relu=False, lib=None, stride=1, precision=None, occupancy=None, return_occupancy=False, unwritten=False):
    """[relu] Conv3d(x, weight, padding=k//2, stride=1|2, bias=None) of the representation plugins
    (ProteinRepresentationModels.py:38-61,85-114) on the matrix cores (inference only: no autograd).
    x (B, cin, D, D, D) float32; weight (cout, cin, k, k, k).
    precision: "f32" = exact f32 products on the f32-input matrix instruction; "split_bf16" = every value as three
    bf16 terms, six bf16 products per f32 product, f32 accumulation (equal to the f32 form to a few 1e-7 relative,
    2-3x faster); None = ``ops.CONV_PRECISION``.
    occupancy (split_bf16 only): ``tile_occupancy(x)`` -- output tiles whose neighbouring input tiles are all empty are
    written as zeros without being computed (there is no bias: they ARE zero; same bits).  return_occupancy: -> (y, the
    occupancy of y or None), which the next layer takes -- a representation network pays for one map, of its input.
    unwritten (needs occupancy, return_occupancy, stride 1): the tensors travel WITH their maps -- cells that ``occupancy``
    marks empty are never read (they count as zeros, whatever the memory holds) and skipped output tiles are not written:
    y is undefined wherever the returned map is 0.  Only for consumers that go by the map (the next layer, maxpool3d_5s2,
    the engine's volumes path)."""
    lib = lib or get_lib()
    x = x.contiguous()
    if unwritten and not (occupancy is not None and return_occupancy and stride == 1 and (precision or CONV_PRECISION) == "split_bf16"):
        raise RuntimeError("dlpd: conv3d(unwritten=True) needs the input's occupancy, return_occupancy, stride 1 and split_bf16")
    if x.dtype != torch.float32 or x.dim() != 5 or not (x.shape[2] == x.shape[3] == x.shape[4]):
        raise RuntimeError("dlpd: conv3d expects (B, C, D, D, D) float32, got %s %s" % (x.dtype, tuple(x.shape)))
    B, cin, D = x.shape[0], x.shape[1], x.shape[2]
    w = weight.detach().to(device=x.device, dtype=torch.float32).contiguous()
    cout, ks = w.shape[0], w.shape[2]
    if w.shape[1] != cin:
        raise RuntimeError("dlpd: conv3d channel mismatch %d vs %d" % (w.shape[1], cin))
    precision = precision or CONV_PRECISION
    if precision not in ("f32", "split_bf16"):
        raise RuntimeError("dlpd: conv3d precision %r" % (precision,))
    split = precision == "split_bf16"
    wp = _packed_weights(weight, w, lib, x.device, split)
    if stride not in (1, 2):
        raise RuntimeError("dlpd: conv3d stride %r not supported (1 or 2)" % (stride,))
    Do = (D - 1) // stride + 1
    y = torch.empty(B, cout, Do, Do, Do, dtype=torch.float32, device=x.device)
    occ_out = None
    if split and (occupancy is not None or return_occupancy):
        if occupancy is not None and (occupancy.dtype != torch.uint8 or occupancy.device != x.device or
                                      occupancy.numel() != lib.call("dlpd_conv3d_tile_occupancy_bytes", B, D)):
            raise RuntimeError("dlpd: conv3d occupancy does not belong to this input (shape %s)" % (tuple(occupancy.shape),))
        if return_occupancy and stride == 1:
            occ_out = torch.empty(B, (D + 3) // 4, (D + 3) // 4, (D + 3) // 4, dtype=torch.uint8, device=x.device)
        lib.call("dlpd_conv3d_split_sparse", _ptr(x), _ptr(wp), _ptr(y),
                 _ptr(occupancy.contiguous()) if occupancy is not None else None, _ptr(occ_out) if occ_out is not None else None,
                 B, cin, cout, D, ks, int(bool(relu)), int(stride), int(bool(unwritten)), _stream(x.device))
    else:
        lib.call("dlpd_conv3d_split" if split else "dlpd_conv3d_strided", _ptr(x), _ptr(wp), _ptr(y), B, cin, cout, D, ks,
                 int(bool(relu)), int(stride), _stream(x.device))
    return (y, occ_out) if return_occupancy else y


_PACKED = {}


def _packed_weights(weight, w, lib, device, split=False):
    """dlpd_conv3d_pack; cached per (parameter, version) for nn.Parameters -- inference weights do not
    change between batches.  Other tensors (e.g. kernels composed on the fly) are packed per call: their
    storage may be recycled with new contents under the same address."""
    cacheable = isinstance(weight, torch.nn.Parameter)
    key = (id(weight), weight.data_ptr(), weight._version, tuple(weight.shape), str(device), id(lib), bool(split))
    if cacheable and key in _PACKED:
        return _PACKED[key]
    cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    if split:
        wp = torch.empty(lib.call("dlpd_conv3d_split_packed_bytes", cin, cout, ks), dtype=torch.uint8, device=device)
        lib.call("dlpd_conv3d_split_pack", _ptr(w), _ptr(wp), cin, cout, ks, _stream(device))
    else:
        wp = torch.empty(lib.call("dlpd_conv3d_packed_floats", cin, cout, ks), dtype=torch.float32, device=device)
        lib.call("dlpd_conv3d_pack", _ptr(w), _ptr(wp), cin, cout, ks, _stream(device))
    if cacheable:
        if device.type == "cuda":
            # the packed copy outlives this call and may next be used from ANOTHER stream (a sweep prepares the next target
            # on a stream of its own, Docker.prepare): it must be complete before it is published -- once per weight
            torch.cuda.current_stream(device).synchronize()
        if len(_PACKED) > 64:
            _PACKED.clear()
        _PACKED[key] = wp
    return wp


def maxpool3d_5s2(x, lib=None, occupancy=None, return_occupancy=False, unwritten=False):
    """MaxPool3d(kernel_size=5, stride=2, padding=2) of the E3 plugin, (B, C, D, D, D) float32 (inference).
    occupancy: ``tile_occupancy(x)`` (or what the convolution that made x handed on) -- output tiles whose inputs lie in
    empty cells are zeros and are written without reading them; return_occupancy: -> (y, the occupancy of y).
    unwritten (needs occupancy and return_occupancy): as conv3d -- empty input cells are not read, empty output tiles not written."""
    lib = lib or get_lib()
    if unwritten and not (occupancy is not None and return_occupancy):
        raise RuntimeError("dlpd: maxpool3d_5s2(unwritten=True) needs the input's occupancy and return_occupancy")
    x = x.contiguous()
    B, C, D = x.shape[0], x.shape[1], x.shape[2]
    Do = (D - 1) // 2 + 1
    y = torch.empty(B, C, Do, Do, Do, dtype=torch.float32, device=x.device)
    if occupancy is None and not return_occupancy:
        lib.call("dlpd_maxpool3d_5s2", _ptr(x), _ptr(y), B * C, D, _stream(x.device))
        return y
    if occupancy is not None and (occupancy.dtype != torch.uint8 or occupancy.device != x.device or
                                  occupancy.numel() != lib.call("dlpd_conv3d_tile_occupancy_bytes", B, D)):
        raise RuntimeError("dlpd: maxpool3d_5s2 occupancy does not belong to this input (shape %s)" % (tuple(occupancy.shape),))
    occ_out = torch.empty(B, (Do + 3) // 4, (Do + 3) // 4, (Do + 3) // 4, dtype=torch.uint8, device=x.device) if return_occupancy else None
    lib.call("dlpd_maxpool3d_5s2_sparse", _ptr(x), _ptr(y), _ptr(occupancy.contiguous()) if occupancy is not None else None,
             _ptr(occ_out), B, C, D, int(bool(unwritten)), _stream(x.device))
    return (y, occ_out) if return_occupancy else y

"""ctypes binding of the C-ABI library (include/dlpd.h).

The product path has NO fallback: if ``csrc/libdlpd.so`` is missing or an entry point is absent
the import of the ops fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "csrc", "libdlpd.so")

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_ll = ctypes.c_longlong
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/dlpd.h exactly
SIGNATURES = {
    "dlpd_version": (_i, []),
    "dlpd_source_hash": (ctypes.c_char_p, []),
    "dlpd_debug_poison_lds": (_i, [_i]),
    "dlpd_debug_poison_selfcheck": (_i, [_p, _p]),
    "dlpd_grid_supported": (_i, [_i]),
    "dlpd_orientation_supported": (_i, [_i]),
    "dlpd_hidden_pad": (_i, [_i]),
    "dlpd_fused_hidden_pad": (_i, [_i, _i, _i]),
    "dlpd_generic_box_supported": (_i, [_i]),
    "dlpd_correlate_generic_ws_bytes": (_sz, [_i, _i]),
    "dlpd_correlate_generic": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p]),
    "dlpd_rotate_trilinear": (_i, [_p, _p, _p, _i, _i, _i, _ll, _f, _p]),
    "dlpd_zfft": (_i, [_p, _p, _p, _i, _i, _i, _ll, _i, _f, _p]),
    "dlpd_zfft_into": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _ll, _i, _f, _p]),
    "dlpd_zfft_volumes_occ": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _ll, _i, _p]),
    "dlpd_project_atoms": (_i, [_p, _p, _p, _p, _f, _f, _f, _p, _i, _i, _i, _i, _f, _i, _p]),
    "dlpd_project_atoms_ext": (_i, [_p, _p, _p, _p, _f, _f, _f, _p, _i, _i, _i, _i, _f, _i, _f, _i, _f, _f, _p]),
    "dlpd_project_atoms_cells": (_i, [_p, _p, _p, _p, _f, _f, _f, _p, _p, _i, _i, _i, _i, _f, _f, _i, _f, _f, _p]),
    "dlpd_rfft3d_padded": (_i, [_p, _p, _p, _i, _i, _f, _p]),
    "dlpd_xy_correlate": (_i, [_p, _p, _p, _i, _i, _i, _ll, _p]),
    "dlpd_receptor_packed_floats": (_ll, [_i, _i]),
    "dlpd_receptor_pack": (_i, [_p, _p, _i, _i, _p]),
    "dlpd_xy_correlate_packed": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "dlpd_quads_floats": (ctypes.c_size_t, [_i, _i]),
    "dlpd_make_quads": (_i, [_p, _p, _i, _i, _p]),
    "dlpd_zfft_quads": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dlpd_channels_last_floats": (ctypes.c_size_t, [_i, _i]),
    "dlpd_make_channels_last": (_i, [_p, _p, _i, _i, _p]),
    "dlpd_zfft_channels_last": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "dlpd_zfft_oriented": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _ll, _i, _f, _i, _p]),
    "dlpd_zfft_oriented_ext": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _ll, _i, _f, _i, _i, _p]),
    "dlpd_zfft_channels_last_ext": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dlpd_zfft_channels_last_form": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "dlpd_k1_form_supported": (_i, [_i, _i]),
    "dlpd_rotated_occupancy": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "dlpd_zfft_channels_last_occ": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "dlpd_pencil_map_supported": (_i, [_i]),
    "dlpd_pencil_bits": (_i, [_p, _p, _i, _i, _p]),
    "dlpd_xy_correlate_packed_occ": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "dlpd_xy_correlate_oriented": (_i, [_p, _p, _p, _i, _i, _i, _ll, _i, _p]),
    "dlpd_score_rotations_oriented": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _f, _i, _i, _f, _f,
                                           _p, _p, _p, _i, _p]),
    "dlpd_zifft_real": (_i, [_p, _p, _i, _i, _i, _i, _f, _p]),
    "dlpd_zifft_filter": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _i, _f, _f, _p]),
    "dlpd_zifft_filter_cand": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _i, _f, _f, _p, _i, _i, _p, _p, _p, _i, _p]),
    "dlpd_zifft_filter_form": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _i, _f, _f, _p, _i, _i, _p, _p, _p, _i, _i, _p]),
    "dlpd_zifft_filter_aux": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _i, _f, _f, _p, _i, _i, _p]),
    "dlpd_score_rotations": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _f, _i, _i, _f, _f,
                                  _p, _p, _p, _p]),
    "dlpd_filter_mask": (_i, [_p, _i, _i, _p, _i, _i, _p, _f, _i, _p, _p, _p, _f, _i, _p, _i, _p]),
    "dlpd_filter_preact": (_i, [_p, _i, _i, _p, _p, _i, _p, _i, _p]),
    "dlpd_filter_volumes": (_i, [_p, _i, _ll, _i, _p, _i, _i, _i, _p, _ll, _f, _i, _p, _p, _p, _f, _i, _p, _i, _p]),
    "dlpd_zifft_preact": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _f, _p]),
    "dlpd_zifft_preact_form": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _f, _i, _p]),
    "dlpd_zifft_real_part": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "dlpd_maxpool3d_5s2": (_i, [_p, _p, _i, _i, _p]),
    "dlpd_maxpool3d_5s2_sparse": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "dlpd_conv3d_supported": (_i, [_i, _i, _i, _i]),
    "dlpd_conv3d_packed_floats": (ctypes.c_size_t, [_i, _i, _i]),
    "dlpd_conv3d_pack": (_i, [_p, _p, _i, _i, _i, _p]),
    "dlpd_conv3d": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dlpd_conv3d_strided": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dlpd_conv3d_split_packed_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "dlpd_conv3d_split_pack": (_i, [_p, _p, _i, _i, _i, _p]),
    "dlpd_conv3d_split": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dlpd_conv3d_tile_occupancy_bytes": (_sz, [_i, _i]),
    "dlpd_conv3d_tile_occupancy": (_i, [_p, _p, _i, _i, _i, _p]),
    "dlpd_conv3d_split_sparse": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dlpd_topk_workspace_bytes": (_sz, [_i, _i]),
    "dlpd_topk_select": (_i, [_p, _i, _ll, _i, _p, _p, _p, _p]),
    "dlpd_topk_select_cand": (_i, [_p, _i, _ll, _i, _p, _p, _p, _p, _p, _i, _p]),
    "dlpd_topk_merge_tau": (_i, [_p, _p, _p, _i, _i, _p, _p, _p]),
    "dlpd_topk_glist_bytes": (_sz, [_i]),
    "dlpd_topk_glist_reset": (_i, [_p, _i, _p]),
    "dlpd_topk_merge": (_i, [_p, _p, _p, _i, _i, _p, _p]),
}

ERRORS = {1: "DLPD_ERR_ARG (bad pointer/size)", 2: "DLPD_ERR_UNSUPPORTED (size not compiled)",
          3: "DLPD_ERR_LAUNCH (HIP launch error)"}


class DlpdLib:
    def __init__(self, path=None):
        path = path or os.environ.get("DLPD_LIB_PATH") or DEFAULT_PATH   # (override: A/B builds)
        if not os.path.exists(path):
            raise RuntimeError(
                "dlpd: native library %s not found -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback." % path)
        self.path = path
        # torch must be imported BEFORE the library is loaded: it brings its own libamdhip64, and
        # libdlpd.so's DT_NEEDED entry then binds to that already-loaded runtime.  Loaded the other
        # way round the process holds two HIP runtimes and ours sees "no ROCm-capable device".
        import torch  # noqa: F401
        self._dll = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._dll, name)          # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)

    def source_hash(self):
        return self._dlpd_source_hash().decode()

    def call(self, name, *args):
        rc = getattr(self, "_" + name)(*args)
        if SIGNATURES[name][0] is _i and name not in ("dlpd_version", "dlpd_grid_supported", "dlpd_conv3d_supported", "dlpd_orientation_supported",
                                                      "dlpd_hidden_pad", "dlpd_fused_hidden_pad", "dlpd_generic_box_supported", "dlpd_debug_poison_selfcheck", "dlpd_k1_form_supported", "dlpd_pencil_map_supported") and rc != 0:
            raise RuntimeError("dlpd: %s failed: %s" % (name, ERRORS.get(rc, rc)))
        return rc


_LIB = None


def get_lib():
    global _LIB
    if _LIB is None:
        _LIB = DlpdLib()
    return _LIB

"""Parity tests proper: the gfx950 build, called through the C ABI (ctypes), against the oracle,
the golden fixtures recorded from the reference, and size-independent properties at full size.

Tolerance (BASELINE north_star: <= 1e-4 relative fp32): |V_hip - V_oracle| <= 1e-4 * max|V| per
rotation; ranked lists must agree in score to that band and in pose except for swaps/replacements
inside the band.  Voxels whose clash correlation is within 1e-3*thr of the threshold may flip mask.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import docking_oracle as orc

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs a GPU"
    import __graft_entry__ as entry
    entry.build()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def variants():
    """tests/variants/libdlpd_variants.so: the formulations libdlpd.so does not ship because no Docker path reaches them
    (channel-owning K3 at N = 128 / 160, transposed-slab K2 at N = 160) -- the bit-exactness references."""
    import __graft_entry__ as entry
    from deeplocalproteindocking_amd._lib import DlpdLib
    return DlpdLib(entry.build_test_variants())


def _pair(L, C, seed=0, amp=0.1):
    g = torch.Generator().manual_seed(seed)
    rec = torch.randn(C, L, L, L, generator=g) * amp
    lig = torch.randn(C, L, L, L, generator=g) * amp
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    H = C // 2
    W1, b1 = torch.randn(H, C, generator=g) * 0.3, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    return rec, lig, recf, ligf, W1, b1, W2, b2


def _rots(n, seed=1):
    ang = np.random.RandomState(seed).uniform(-np.pi, np.pi, size=(n, 3))
    return orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])


def _oracle_V(rec, lig, recf, ligf, W, R1, thr, clip):
    Rb = torch.from_numpy(R1[None]).float()
    lr = orc.rotate_volume(lig[None], Rb)
    lfr = orc.rotate_volume(ligf[None, None], Rb)
    mask, norm = orc.clash_mask(recf[None, None], lfr, thr)
    V = (mask * orc.score_volumes([rec[None]], [lr], *W, clip=clip))[0]
    return V, norm[0]


@pytest.mark.parametrize("L,C,nrot,clip", [(32, 4, 5, 5.0), (32, 4, 3, 0.3), (32, 48, 2, 5.0), (64, 48, 2, 5.0),
                                           (64, 6, 3, None), (40, 32, 2, 5.0), (80, 16, 2, 5.0)])
def test_scores_match_oracle(dev, L, C, nrot, clip):
    from deeplocalproteindocking_amd.engine import DockingEngine
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, C, seed=L + C, amp=0.05)
    thr = 0.25 * L ** 3 * 0.5
    R = _rots(nrot)
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=clip, threshold_clash=thr, max_conf=100, batch=3, device=dev)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    Rd = torch.from_numpy(R).float().to(dev).contiguous()
    for beg in range(0, nrot, 3):
        V = eng.score_batch(Rd[beg:beg + 3]).cpu()
        for j in range(V.shape[0]):
            Vo, norm = _oracle_V(rec, lig, recf, ligf, (W1, b1, W2, b2), R[beg + j], thr, clip)
            sure = (norm - thr).abs() > 1e-3 * thr
            assert 0.3 < (Vo != 0).float().mean() <= 1.0
            err = (V[j] - Vo).abs()[sure].max().item()
            assert err <= TOL * Vo.abs().max().item(), (err, Vo.abs().max().item())


def test_full_search_ranked_list_matches_oracle(dev):
    """BASELINE config 1 shape (4ch, 32^3), 40 rotations, batch 7 (odd tail), K=300."""
    from deeplocalproteindocking_amd.engine import DockingEngine
    L, C, K, nrot = 32, 4, 300, 40
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, C, seed=9)
    thr = 4000.0
    R = _rots(nrot, seed=3)
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=thr, max_conf=K, batch=7, device=dev)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    eng.reset_top()
    eng.search(R)
    got = eng.top_list()
    want, Vs = orc.dock_volumes([rec[None]], [lig[None]], recf[None, None], ligf[None, None], R, W1, b1, W2, b2,
                                thr, K, clip=5.0, faithful_topk=False, return_V=True)
    scale = max(float(v.abs().max()) for v in Vs)
    band = TOL * scale
    assert len(got) == len(want) == K
    assert max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= band
    want_set = {w[:4] for w in want}
    kth = want[-1][4]
    for a, b in zip(got, want):
        if a[:4] != b[:4]:                      # allowed only for near-ties inside the band
            assert abs(a[4] - b[4]) <= band and (a[:4] in want_set or abs(a[4] - kth) <= 2 * band)
    assert sum(a[:4] == b[:4] for a, b in zip(got, want)) >= int(0.97 * K)
    # the device merge of the per-rotation picks is EXACT given the device's own V
    eng.reset_top()
    eng2 = []
    Rd = torch.from_numpy(R).float().to(dev).contiguous()
    flags = DockingEngine.prefers_transposed(R)          # search() scores each rotation in this slab orientation
    quads = DockingEngine.prefers_quads(R)               # ... and from this gather layout
    for r in range(nrot):
        V = eng.score_batch(Rd[r:r + 1], transposed=bool(flags[r]), quads=bool(quads[r])).cpu()
        idx, sc = orc.rotation_picks_fast(V[0].numpy(), K)
        x, y, z = orc.flat_to_xyz(idx, 2 * L)
        eng2 += [(r, int(x[i]), int(y[i]), int(z[i]), float(sc[i])) for i in range(K)]
        eng2.sort(key=lambda t: t[4])
        eng2 = eng2[:K]
    assert got == eng2


@pytest.mark.parametrize("case", ["randn8_k5", "randn16_k40", "onehot_k4", "allpos_k4", "ties_k30", "fewneg_k6",
                                  "masked_k12"])
def test_update_top_reproduces_reference_fixtures(dev, golden, case):
    """Docker.update_top on the GPU against outputs recorded from the reference's update_top."""
    from deeplocalproteindocking_amd.Docker import Docker
    g = golden("g3_update_top.npz")
    V = torch.from_numpy(g[case + "_V"].copy()).to(dev)
    dk = Docker(None, box_size=V.shape[0] // 2, max_conf=int(g[case + "_K"]), rotations=np.tile(np.eye(3), (8, 1, 1)),
                device=dev)
    dk.top_list = []
    dk.update_top(V, 7)
    assert dk.top_list == [(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(b[4])) for b in g[case + "_top"]]
    np.testing.assert_array_equal(V.cpu().numpy(), g[case + "_Vafter"])       # V mutated like Docker.py:98


def test_update_top_sequence_reproduces_reference(dev, golden):
    from deeplocalproteindocking_amd.Docker import Docker
    g = golden("g3_update_top.npz")
    dk = Docker(None, box_size=3, max_conf=int(g["seq_K"]), rotations=np.tile(np.eye(3), (8, 1, 1)), device=dev)
    dk.top_list = []
    for r, V in enumerate(g["seq_V"]):
        dk.update_top(torch.from_numpy(V.copy()).to(dev), r)
    assert dk.top_list == [(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(b[4])) for b in g["seq_top"]]


def test_topk_full_size_exact(dev):
    """K=2000 over 128^3 (BASELINE size): exact vs the vectorised oracle, incl. a half-masked volume."""
    from deeplocalproteindocking_amd.engine import DeviceTopList
    from deeplocalproteindocking_amd._lib import get_lib
    N, K, nb = 128, 2000, 3
    g = torch.Generator().manual_seed(4)
    V = torch.randn(nb, N, N, N, generator=g)
    V[1] *= (torch.rand(N, N, N, generator=g) < 0.5).float()                 # half masked -> +-0.0
    V[2] = torch.round(V[2] * 4) / 4                                         # heavy ties
    V[2][V[2] < -0.4] = 0.0                                                  # few negatives + zero fill
    top = DeviceTopList(K, nb, dev, get_lib())
    cs, ci = top.select(V.to(dev).reshape(nb, -1), nb)
    cs, ci = cs.cpu().numpy(), ci.cpu().numpy()
    for j in range(nb):
        idx, sc = orc.rotation_picks_fast(V[j].numpy(), K)
        assert np.array_equal(ci[j], idx) and np.array_equal(cs[j].view(np.uint32), sc.view(np.uint32))


def get_lib_for_tests():
    from deeplocalproteindocking_amd._lib import get_lib
    return get_lib()


def test_topk_and_merge_with_4096_conformations(dev):
    """max_conf = 4096 (the largest list sorted in LDS; the reference uses 2000): per-rotation picks exact against the vectorised oracle
    at 128^3 and the merged running list of three rotations against the faithful sequence of update_top calls' closed
    form (sort of all picks by (score, rotation, pick))."""
    from deeplocalproteindocking_amd.engine import DeviceTopList
    from deeplocalproteindocking_amd._lib import get_lib
    N, K, nb = 128, 4096, 3
    g = torch.Generator().manual_seed(14)
    V = torch.randn(nb, N, N, N, generator=g)
    V[1] = torch.round(V[1] * 8) / 8                                         # ties
    top = DeviceTopList(K, nb, dev, get_lib())
    top.reset()
    cs, ci = top.select(V.to(dev).reshape(nb, -1), nb)
    cs, ci = cs.cpu().numpy().copy(), ci.cpu().numpy().copy()
    picks = []
    for j in range(nb):
        idx, sc = orc.rotation_picks_fast(V[j].numpy(), K)
        assert np.array_equal(ci[j], idx) and np.array_equal(cs[j].view(np.uint32), sc.view(np.uint32))
        picks += [(float(sc[i]), j, i, int(idx[i])) for i in range(K)]
    top.merge(torch.arange(nb, dtype=torch.int32, device=dev), nb)
    rot, idx, score, pick = top.entries()
    picks.sort(key=lambda p: (p[0], p[1], p[2]))
    want = picks[:K]
    assert rot.tolist() == [p[1] for p in want] and idx.tolist() == [p[3] for p in want]
    assert np.array_equal(np.asarray(score, dtype=np.float32), np.asarray([p[0] for p in want], dtype=np.float32))
    with pytest.raises(RuntimeError):
        DeviceTopList(65537, 1, dev, get_lib()).select(V.to(dev).reshape(nb, -1)[:1], 1)


def test_topk_and_merge_with_20000_conformations(dev):
    """max_conf above 4096: the same kernels with their sorts in global scratch (128^3 grid, K = 20,000, three rotations
    with ties and a zero-fill case)."""
    from test_kernels_emu import _topk_large_k
    _topk_large_k(get_lib_for_tests(), dev, 128, 20000, 3)


@pytest.mark.parametrize("tag,nres", [("multires", 2), ("single", 1)])
def test_filter_kernel_reproduces_reference_forward(dev, golden, tag, nres):
    from deeplocalproteindocking_amd.ops import filter_volumes
    g = golden("g5_global_forward.npz")
    rec = [torch.from_numpy(g["%s_rec%d" % (tag, i)]) for i in range(nres)]
    lig = [torch.from_numpy(g["%s_lig%d" % (tag, i)]) for i in range(nres)]
    conv = [orc.correlate_fft(r, l, clip=float(g[tag + "_clip"])).contiguous().to(dev) for r, l in zip(rec, lig)]
    V = filter_volumes(conv, torch.from_numpy(g[tag + "_W1"]), torch.from_numpy(g[tag + "_b1"]),
                       torch.from_numpy(g[tag + "_W2"]), float(g[tag + "_b2"][0]))
    np.testing.assert_allclose(V.cpu().numpy(), g[tag + "_V"], rtol=1e-5, atol=1e-5)


def test_volume_ops_properties_at_full_size(dev):
    """Size-independent properties at L=64 (N=128): DC checksum, shift recovery, clip, linearity,
    |t|=L planes empty; rotation identity / exact quarter turn."""
    from deeplocalproteindocking_amd.ops import VolumeConvolution, VolumeRotation
    L, N = 64, 128
    g = torch.Generator().manual_seed(6)
    v1 = torch.randn(1, 2, L, L, L, generator=g)
    v2 = torch.zeros(1, 2, L, L, L)
    shift = (5, -7, 11)
    src = v1[0, :, 10:40, 20:50, 15:45]
    v2[0, :, 10 - shift[0]:40 - shift[0], 20 - shift[1]:50 - shift[1], 15 - shift[2]:45 - shift[2]] = src
    conv = VolumeConvolution()
    out = conv(v1.to(dev), v2.to(dev)).cpu()
    assert out.shape == (1, 2, N, N, N)
    dc = out.double().sum(dim=(2, 3, 4))
    want = v1.double().sum(dim=(2, 3, 4)) * v2.double().sum(dim=(2, 3, 4))
    assert ((dc - want).abs() <= 1e-3 * out.double().abs().sum(dim=(2, 3, 4)) * 1e-3 + 1e-2 * want.abs()).all()
    for c in range(2):
        peak = np.unravel_index(int(out[0, c].argmax()), (N, N, N))
        assert tuple(int(p) for p in peak) == tuple(s % N for s in shift)      # out[t]: v1[r+t] v2[r]
    assert out[0, :, L].abs().max() < 1e-2 and out[0, :, :, :, L].abs().max() < 1e-2
    ref = orc.correlate_fft(v1, v2)
    assert (out - ref).abs().max() <= TOL * ref.abs().max()
    clipped = VolumeConvolution(clip=5.0)(v1.to(dev), v2.to(dev)).cpu()
    assert clipped.abs().max() <= 5.0 and (clipped - ref.clamp(-5, 5)).abs().max() <= TOL * ref.abs().max()
    a, b = torch.randn(1, 1, L, L, L, generator=g), torch.randn(1, 1, L, L, L, generator=g)
    w = torch.randn(1, 1, L, L, L, generator=g)
    lin = conv((2 * a - 3 * b).to(dev), w.to(dev)) - (2 * conv(a.to(dev), w.to(dev)) - 3 * conv(b.to(dev), w.to(dev)))
    assert lin.abs().max().item() <= 1e-2
    # rotation
    rot = VolumeRotation()
    vol = torch.randn(2, 3, L, L, L, generator=g).to(dev)
    eye = torch.eye(3).repeat(2, 1, 1).to(dev)
    assert torch.equal(rot(vol, eye), vol)
    Rz = torch.tensor([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]]).repeat(2, 1, 1).to(dev)   # quarter turn about z
    q = rot(vol, Rz).cpu()
    volc = vol.cpu()
    # out(x,y,z) = vol(c0 + R^T (i - c0)) = vol(y, L - x, z); x = 0 samples index L -> zero
    assert torch.equal(q[:, :, 1:], volc.transpose(2, 3).flip(2)[:, :, :L - 1])
    assert q[:, :, 0].abs().max() == 0
    Rr = torch.from_numpy(_rots(2, seed=8)).float()
    got = rot(vol, Rr.to(dev)).cpu()
    assert (got - orc.rotate_volume(volc, Rr)).abs().max() < 1e-4


def test_reference_model_shapes_multires(dev):
    """The reference's real layout [16 @ 80^3, 32 @ 40^3] -> 160^3 (local_train.py:23,30;
    ProteinRepresentationModels.py:35-36): GlobalDockingModel.forward on the stand-alone ops and
    Docker.dock_volumes (multi-resolution path) against the oracle."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
    L, K = 80, 50
    torch.manual_seed(31)
    repr_ = SyntheticRepr(num_outputs=(16, 32), seed=5, amplitude=0.12)
    filt = SimpleFilter(repr_.get_num_outputs())
    model = GlobalDockingModel(repr_, filt, threshold_clash=0.02 * L ** 3).to(dev)
    rec, lig = repr_.make(L, "rec"), repr_.make(L, "lig")
    assert [tuple(v.shape) for v in rec] == [(1, 16, 80, 80, 80), (1, 32, 40, 40, 40)]
    W = [w.cpu() for w in filt.parameters_tuple()]
    V = model([v.to(dev) for v in rec], [v.to(dev) for v in lig]).cpu()
    Vo = orc.score_volumes(rec, lig, *W, clip=5.0)
    assert V.shape == (1, 160, 160, 160)
    assert (V - Vo).abs().max() <= TOL * Vo.abs().max()
    R = _rots(3, seed=15)
    g = torch.Generator().manual_seed(32)
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    dk = Docker(model, box_size=L, max_conf=K, rotations=R, device=dev)
    got = dk.dock_volumes(rec, lig, recf, ligf, batch_size=2, write=False)
    want = orc.dock_volumes(rec, lig, recf[None, None], ligf[None, None], R, *W, 0.02 * L ** 3, K, clip=5.0,
                            faithful_topk=False)
    scale = float(Vo.abs().max())
    assert max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= TOL * scale
    assert sum(a[:4] == b[:4] for a, b in zip(got, want)) >= K - 3


def test_global_docking_model_forward_and_docker_dock_volumes(dev):
    """The plugin surface end to end on the GPU: GlobalDockingModel.forward (stand-alone ops)
    equals the oracle, and Docker.dock_volumes (fused path) equals the same search done with the
    stand-alone ops path."""
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SimpleFilter, SyntheticRepr
    L, C, K = 32, 6, 64
    torch.manual_seed(12)
    filt = SimpleFilter([C])
    repr_ = SyntheticRepr(num_outputs=(C,), seed=3, amplitude=0.2)
    model = GlobalDockingModel(repr_, filt, threshold_clash=0.02 * L ** 3).to(dev)
    rec = repr_.make(L, "rec")[0]
    lig = repr_.make(L, "lig")[0]
    W = filt.parameters_tuple()
    V = model([rec.to(dev)], [lig.to(dev)]).cpu()
    Vo = orc.score_volumes([rec], [lig], *[w.cpu() for w in W], clip=5.0)
    assert (V - Vo).abs().max() <= TOL * Vo.abs().max()
    R = _rots(9, seed=5)
    g = torch.Generator().manual_seed(13)
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    dk = Docker(model, box_size=L, max_conf=K, rotations=R, device=dev)
    fused = dk.dock_volumes([rec], [lig], recf, ligf, batch_size=4, write=False)
    entries = dk._dock_volumes_multires([rec[0]], [lig[0]], recf, ligf, 4, np.arange(9))
    from deeplocalproteindocking_amd.engine import DeviceTopList
    generic = DeviceTopList.to_top_list(entries, 2 * L)
    scale = float(Vo.abs().max())
    assert max(abs(a[4] - b[4]) for a, b in zip(fused, generic)) <= TOL * scale
    assert sum(a[:4] == b[:4] for a, b in zip(fused, generic)) >= K - 3
    want = orc.dock_volumes([rec], [lig], recf[None, None], ligf[None, None], R, *[w.cpu() for w in W],
                            0.02 * L ** 3, K, clip=5.0, faithful_topk=False)
    assert max(abs(a[4] - b[4]) for a, b in zip(fused, want)) <= TOL * scale
    assert sum(a[:4] == b[:4] for a, b in zip(fused, want)) >= K - 3


def test_baseline_config1_full_rotation_set(dev):
    """BASELINE config 1 end to end: synthetic 4-channel 32^3 pair, the COMPLETE 20-degree rotation set
    (1,854 rotations, SOI-sized; the generated substitute when the licensed file is absent), K=1000:
    the final ranked list of the GPU search against the oracle run over the same rotations
    ("config 1 runs in full on CPU and its complete ranked list is the parity check")."""
    import bench
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.Models import GlobalDockingModel, SyntheticRepr
    from deeplocalproteindocking_amd.Utils.Rotations import Rotations
    C, L, K = 4, 32, 1000
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    thr = bench.clash_threshold(recf, ligf)
    rot = Rotations(20, allow_generated=True, verbose=False)
    assert rot.R.shape[0] == 1854
    model = GlobalDockingModel(SyntheticRepr((C,)), filt, threshold_clash=thr).to(dev)
    dk = Docker(model, angle_inc=20, box_size=L, max_conf=K, rotations=rot.R.numpy(), device=dev)
    got = dk.dock_volumes([rec], [lig], recf, ligf, batch_size=16, write=False)
    W = [w.cpu() for w in filt.parameters_tuple()]
    want = orc.dock_volumes([rec[None]], [lig[None]], recf[None, None], ligf[None, None], rot.R.numpy(), *W, thr,
                            K, clip=5.0, faithful_topk=False)
    scale = max(abs(w[4]) for w in want)          # <= max|V|: a stricter band than the stated tolerance
    band = TOL * scale
    assert len(got) == len(want) == K
    assert max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= band
    same = sum(a[:4] == b[:4] for a, b in zip(got, want))
    assert same >= int(0.98 * K), same
    assert {g[0] for g in got} == {w[0] for w in want} or same >= int(0.99 * K)    # same rotations win


def test_full_size_search_is_batch_size_and_order_independent(dev):
    """BASELINE config 2 size (48 ch, 64^3, K = 2000): properties that need no oracle.  The ranked list of a
    search must not depend on how rotations are batched (16 vs 7 per launch, odd tail) nor on the order
    in which the same rotations are presented (merge key = (score, rotation, pick)); it is sorted, its
    rotation ids are those searched, and a second run is bit-identical."""
    import bench
    from deeplocalproteindocking_amd.engine import DockingEngine
    C, L, K, nrot = 48, 64, 2000, 75
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    thr = bench.clash_threshold(recf, ligf)
    R = _rots(nrot, seed=77)
    lists = []
    for nb, order in ((16, np.arange(nrot)), (7, np.arange(nrot)), (16, np.arange(nrot)[::-1].copy()), (16, np.arange(nrot))):
        eng = DockingEngine(L, C, *filt.parameters_tuple(), clip=5.0, threshold_clash=thr, max_conf=K, batch=nb, device=dev)
        eng.set_receptor(rec, recf)
        eng.set_ligand(lig, ligf)
        eng.reset_top()
        if order[0] == 0:
            eng.search(R[order], rot_ids=order)
        else:   # another visiting order: the groups of search() reversed, each group in descending set order
            flags, quads = DockingEngine.prefers_transposed(R), DockingEngine.prefers_quads(R)
            for tr, qd in ((True, True), (True, False), (False, True), (False, False)):
                grp = order[(flags[order] == tr) & (quads[order] == qd)]
                for beg in range(0, len(grp), nb):
                    ids = np.sort(grp[beg:beg + nb])
                    eng.step(torch.from_numpy(R[ids]).float().to(dev).contiguous(),
                             torch.from_numpy(ids.astype(np.int32)).to(dev), transposed=tr, quads=qd)
            eng.finish()
        lists.append(eng.top_list())
        del eng
    ref = lists[0]
    assert len(ref) == K and all(a[4] <= b[4] for a, b in zip(ref[:-1], ref[1:]))
    assert {t[0] for t in ref} <= set(range(nrot)) and len({t[0] for t in ref}) > 10
    assert all(0 <= v < 2 * L for t in ref for v in t[1:4])
    assert lists[1] == ref, "batch size changed the result"
    assert lists[2] == ref, "presentation order changed the result"
    assert lists[3] == ref, "rerun is not bit-identical"


def test_full_size_rank_shards_merge_to_the_single_process_list(dev):
    """SURVEY 8(e) at BASELINE config 2 size on one GPU: the interleaved shards of W = 3 ranks, searched
    separately and merged with the deterministic (score, rotation, pick) key, give exactly the list of the
    unsharded search (the all-gather itself is covered by the gloo test)."""
    import bench
    from deeplocalproteindocking_amd.engine import DeviceTopList, DockingEngine
    C, L, K, nrot, W = 48, 64, 2000, 50, 3
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    thr = bench.clash_threshold(recf, ligf)
    R = _rots(nrot, seed=78)

    def run(ids):
        eng = DockingEngine(L, C, *filt.parameters_tuple(), clip=5.0, threshold_clash=thr, max_conf=K, batch=16, device=dev)
        eng.set_receptor(rec, recf)
        eng.set_ligand(lig, ligf)
        eng.reset_top()
        eng.search(R[ids], rot_ids=ids)
        return eng.top_entries()
    whole = run(np.arange(nrot))
    parts = [run(np.arange(r, nrot, W)) for r in range(W)]
    merged = DeviceTopList.merge_entries(parts, K)
    assert DeviceTopList.to_top_list(merged, 2 * L) == DeviceTopList.to_top_list(whole, 2 * L)


@pytest.mark.parametrize("cin,cout,ks,D", [(11, 16, 5, 80), (16, 16, 3, 80), (16, 32, 5, 40), (32, 32, 3, 40), (11, 32, 3, 37), (32, 64, 5, 40), (11, 48, 3, 21)])
def test_conv3d_matches_torch_at_plugin_shapes(dev, cin, cout, ks, D):
    """dlpd_conv3d (f32 MFMA implicit GEMM) against torch's conv3d on the layer shapes of the reference's
    representation plugins (ProteinRepresentationModels.py:85-114) and an awkward size; exact-f32 products,
    so the difference is summation order only."""
    from deeplocalproteindocking_amd import ops
    g = torch.Generator().manual_seed(100 + cin + D)
    x = torch.randn(2, cin, D, D, D, generator=g).to(dev)
    w = (torch.randn(cout, cin, ks, ks, ks, generator=g) * 0.05).to(dev)
    want0 = torch.nn.functional.conv3d(x.cpu().double(), w.cpu().double(), padding=ks // 2)
    for precision in ("f32", "split_bf16"):          # exact f32 products / three bf16 terms and six products per product
        for relu in (False, True):
            got = ops.conv3d(x, w, relu=relu, precision=precision)
            want = torch.relu(want0) if relu else want0
            err = float((got.cpu().double() - want).abs().max() / want.abs().max())
            assert err <= 1e-5, (precision, relu, err)
    assert ops.CONV_PRECISION == "split_bf16"         # what the plugins run by default


@pytest.mark.parametrize("cin,cmid,cout,ks,D,lo,hi", [(11, 16, 16, 5, 80, 22, 50), (16, 32, 64, 3, 40, 0, 9), (11, 16, 32, 5, 37, 30, 37)])
def test_conv3d_tile_occupancy_skips_empty_tiles_with_the_same_bits(dev, cin, cmid, cout, ks, D, lo, hi):
    """The bias-free convolutions with their empty tiles skipped (ops.conv3d(occupancy=...), what the E3 plugin runs per
    batch of rotations) against the same convolutions computed everywhere, at the plugin's box sizes: same bits, and the
    occupancy map every layer hands on equals the map of its output."""
    from test_kernels_emu import _sparse_conv_checks
    _sparse_conv_checks(None, dev, cin, cmid, cout, ks, D, lo, hi, B=3)


@pytest.mark.parametrize("C,D,lo,hi", [(16, 80, 22, 50), (5, 37, 0, 9), (32, 40, 30, 40), (12, 37, 3, 20)])
def test_maxpool_tiled_with_occupancy_equals_torch(dev, C, D, lo, hi):
    """The E3 plugin's pooling on the tiled kernel (separable maximum through LDS), with and without occupancy maps, at the
    plugin's sizes and an odd one: torch's values exactly, and the map it hands on equals the map of its output."""
    from test_kernels_emu import _sparse_pool_checks
    _sparse_pool_checks(None, dev, C, D, lo, hi, B=3)


@pytest.mark.gpu
@pytest.mark.parametrize("L,C,lo,hi", [(64, 48, (20, 25, 12), (45, 40, 33)), (80, 16, (10, 30, 22), (38, 61, 50)), (40, 32, (22, 3, 14), (31, 12, 26)),
                                       (32, 20, (9, 12, 6), (20, 19, 15))])
def test_k1_by_occupancy_maps_gives_the_same_spectra(dev, L, C, lo, hi):
    """Round 6, search side: the channels-last K1 of every compiled box going by per-rotation occupancy maps -- the same
    spectra bit for bit as without, conservative maps, and cells really left out (16 oblique rotations per launch)."""
    from test_kernels_emu import _k1_occupancy_checks
    fill = _k1_occupancy_checks(None, dev, L, C, 16, lo, hi)
    assert fill < 0.8
    _k1_occupancy_checks(None, dev, L, C, 5, lo, hi, seed=3, scale=1.07)      # a scaled sample map (Utils/Conventions)


@pytest.mark.gpu
@pytest.mark.parametrize("L,C,lo,hi", [(80, 16, (30, 41, 22), (52, 62, 45)), (40, 32, (22, 3, 14), (31, 12, 26))])
def test_k2_by_the_pencil_map_reads_no_unwritten_pencil(dev, L, C, lo, hi):
    """Round 6: K1 leaves the blocks without an occupied cell UNWRITTEN and K2 (boxes 80 / 40, 16 rotations per launch) goes by
    the pencil map -- the same K2 output bit for bit although the workspace between them is NaN wherever nothing was written."""
    from test_kernels_emu import _pencil_map_checks
    _pencil_map_checks(None, dev, L, C, 16, lo, hi)


@pytest.mark.gpu
def test_search_of_a_protein_shaped_pair_is_the_same_with_and_without_occupancy_maps(dev):
    """The reference's real shapes [16 @ 80^3, 32 @ 40^3] with blob-shaped ligand volumes: the engine picks K1 by occupancy
    maps + K2 by pencil maps on its own; the ranked list of 48 rotations equals the list of the dense kernels entry for
    entry, with the K1 -> K2 workspaces refilled with NaNs before every launch."""
    from deeplocalproteindocking_amd.engine import DockingEngine
    g = torch.Generator().manual_seed(31)
    L, C, C1, H = 80, 16, 32, 24
    blob = lambda c, l, a, b: torch.nn.functional.pad(torch.randn(c, b - a, b - a, b - a, generator=g) * 0.3, (a, l - b) * 3)
    rec, lig = blob(C, L, 18, 60), blob(C, L, 26, 52)
    rec1, lig1 = blob(C1, 40, 7, 32), blob(C1, 40, 10, 28)
    recf, ligf = blob(1, L, 22, 56)[0].abs(), blob(1, L, 30, 48)[0].abs()
    W1, b1 = torch.randn(H, C + C1, generator=g) * 0.3, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.zeros(1)
    b2 = b2 - (W2 @ torch.relu(b1).reshape(-1, 1)).reshape(-1)            # a pose without contact scores 0
    R = torch.from_numpy(_rots(48, seed=12)).float()
    lists = {}
    for mode in (None, False):
        eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=200.0, max_conf=2000, batch=16, device=dev, coarse_channels=C1,
                            sparse_k1=mode)
        eng.set_receptor(rec, recf, rec1)
        eng.set_ligand(lig, ligf, lig1)
        sw = eng.switches()["k1_occupancy_maps"]
        assert sw["fine"] == sw["coarse"] == (mode is None) and sw["k2_pencil_map"] == {"fine": mode is None, "coarse": mode is None}
        eng.reset_top()
        Rd = R.to(dev).contiguous()
        ids = torch.arange(48, dtype=torch.int32, device=dev)
        for beg in range(0, 48, 16):
            if mode is None:
                torch.cuda.synchronize()
                eng.wsA.fill_(float("nan"))
                eng.wsA1.fill_(float("nan"))
            eng.step(Rd[beg:beg + 16], ids[beg:beg + 16])
        lists[mode] = eng.top_list()
        del eng
        torch.cuda.empty_cache()
    assert lists[None] == lists[False] and len(lists[None]) == 2000
    assert all(t[4] == t[4] for t in lists[None]) and min(t[4] for t in lists[None]) < 0        # no NaN; real (negative) scores


@pytest.mark.gpu
def test_unwritten_activations_stand_for_the_same_tensors(dev):
    """Round 6 (Docker.dockE3's representation, Docker.py:163-167): convolution / pooling layers that neither compute nor
    WRITE their empty tiles and never read an empty cell, and the engine's K1 for given volumes going by the map
    (dlpd_zfft_volumes_occ) -- same bits wherever a map marks a cell, same maps, same spectra; every output buffer starts as NaNs."""
    from test_kernels_emu import _unwritten_chain_checks
    _unwritten_chain_checks(None, dev, D=45, lo=18, hi=27, B=3)



def test_conv3d_stride2_and_se3_plugin_never_touch_torch_convolutions(dev, monkeypatch):
    """The stride-2 5^3 layer (ProteinRepresentationModels.py:51) on the matrix-core kernel at the reference's
    size (16 -> 32 channels, 80^3 -> 40^3), and the whole SE3MultiResReprScalar(8) forward with torch's conv3d
    made to raise: GPU inference of the plugin runs on the HIP kernels only."""
    from deeplocalproteindocking_amd import ops
    from deeplocalproteindocking_amd.Models import SE3MultiResReprScalar
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 16, 80, 80, 80, generator=g).to(dev)
    w = (torch.randn(32, 16, 5, 5, 5, generator=g) * 0.05).to(dev)
    want = torch.nn.functional.conv3d(x.cpu(), w.cpu(), padding=2, stride=2)
    for precision in ("f32", "split_bf16"):
        got = ops.conv3d(x, w, stride=2, precision=precision)
        assert got.shape == (2, 32, 40, 40, 40) and (got.cpu() - want).abs().max() <= 1e-5 * want.abs().max(), precision
    torch.manual_seed(3)
    model = SE3MultiResReprScalar(multiplier=8).eval()
    vol = torch.rand(1, 11, 80, 80, 80, generator=g)
    with torch.no_grad():
        ref = model(vol)                                          # CPU tensors: the torch reference
        model = model.to(dev)

        def boom(*a, **k):
            raise AssertionError("torch conv3d called on the GPU inference path")
        monkeypatch.setattr(torch.nn.functional, "conv3d", boom)
        out = model(vol.to(dev))
    for a, b in zip(out, ref):
        assert a.shape == b.shape and (a.cpu() - b).abs().max() <= 2e-5 * b.abs().max()


@pytest.mark.parametrize("L,C,H,has_clash,clip,nb", [(32, 1, 1, True, 5.0, 1), (32, 3, 3, False, None, 2), (32, 5, 6, True, 0.2, 4),
                                                      (32, 9, 13, False, 5.0, 3), (40, 2, 20, True, None, 2), (32, 7, 32, True, 1.0, 5)])
def test_scores_match_oracle_odd_shapes(dev, L, C, H, has_clash, clip, nb):
    """Corner configurations of the fused pipeline: single channel, hidden widths that need zero padding
    (1, 3, 6, 13, 20 -> 2, 4, 8, 16, 24) or sit at the limit (32), no clash channel, no clip, odd batch sizes."""
    from deeplocalproteindocking_amd.engine import DockingEngine
    g = torch.Generator().manual_seed(1000 + 7 * C + H)
    rec, lig = torch.randn(C, L, L, L, generator=g) * 0.08, torch.randn(C, L, L, L, generator=g) * 0.08
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    W1, b1 = torch.randn(H, C, generator=g) * 0.4, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    thr = 0.125 * L ** 3
    R = _rots(nb, seed=C + H)
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=clip, threshold_clash=thr, has_clash=has_clash, max_conf=50,
                        batch=nb, device=dev)
    eng.set_receptor(rec, recf if has_clash else None)
    eng.set_ligand(lig, ligf if has_clash else None)
    V = eng.score_batch(torch.from_numpy(R).float().to(dev).contiguous()).cpu()
    for j in range(nb):
        Rb = torch.from_numpy(R[j:j + 1]).float()
        S = orc.score_volumes([rec[None]], [orc.rotate_volume(lig[None], Rb)], W1, b1, W2, b2, clip=clip)[0]
        if has_clash:
            mask, norm = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
            S, sure = mask[0] * S, (norm[0] - thr).abs() > 1e-3 * thr
        else:
            sure = torch.ones_like(S, dtype=torch.bool)
        assert (V[j] - S).abs()[sure].max().item() <= TOL * S.abs().max().item()


def test_topk_fuzz_against_faithful_update_top(dev):
    """The same fuzz as tests/test_kernels_emu.py on the real kernels: ties, signed zeros, too few negatives,
    no zero at all, K up to the voxel count, several rotations per batch -- tuple-for-tuple equality with the
    restated reference loop (Docker.py:86-105), including the sign of recorded zeros."""
    from deeplocalproteindocking_amd._lib import get_lib
    from deeplocalproteindocking_amd.engine import DeviceTopList
    rng = np.random.RandomState(777)
    pools = [np.array([-2.0, -1.0, -1.0, -0.5, 0.0, 0.0, 1.0, 3.0], dtype=np.float32),
             np.array([-1.0, 0.0, -0.0, 2.0], dtype=np.float32), np.array([0.5, 1.0, 2.0], dtype=np.float32),
             np.array([-3.0, -3.0, -3.0, 4.0, 0.0], dtype=np.float32)]
    for trial in range(60):
        N = int(rng.choice([4, 6, 8, 12]))
        K = int(rng.randint(1, min(N ** 3, 64) + 1))
        nrot, batch = int(rng.randint(1, 6)), int(rng.randint(1, 4))
        pool = pools[trial % len(pools)]
        Vs = pool[rng.randint(0, len(pool), size=(nrot, N, N, N))]
        if trial % 5 == 0:
            Vs = (Vs + rng.randn(*Vs.shape).astype(np.float32) * 0.01).astype(np.float32)
        want = []
        for r in range(nrot):
            want = orc.update_top(want, torch.from_numpy(Vs[r].copy()), r, K)
        top = DeviceTopList(K, batch, dev, get_lib())
        top.reset()
        for beg in range(0, nrot, batch):
            nb = min(batch, nrot - beg)
            top.select(torch.from_numpy(Vs[beg:beg + nb].reshape(nb, -1).copy()).to(dev), nb)
            top.merge(torch.arange(beg, beg + nb, dtype=torch.int32, device=dev), nb)
        got = DeviceTopList.to_top_list(top.entries(), N)
        assert got == want, (trial, N, K, nrot, batch)
        assert [np.signbit(a[4]) for a in got] == [np.signbit(b[4]) for b in want], trial
        order = rng.permutation(nrot)                  # visiting order must not matter, exact ties included
        top = DeviceTopList(K, batch, dev, get_lib())
        top.reset()
        for beg in range(0, nrot, batch):
            ids = np.sort(order[beg:beg + batch])
            top.select(torch.from_numpy(Vs[ids].reshape(len(ids), -1).copy()).to(dev), len(ids))
            top.merge(torch.from_numpy(ids.astype(np.int32)).to(dev), len(ids))
        assert DeviceTopList.to_top_list(top.entries(), N) == want, (trial, "permuted", order)


def test_quad_gather_equals_plain_gather(dev):
    """dlpd_zfft_quads (two 16-byte gathers per sample from the quad layout) against dlpd_zfft_oriented (four
    8-byte gathers from the plain volume): same weights and products, so the z-spectra agree to round-off
    of the compiler's FMA contraction (<= 1e-6 of the largest coefficient), for both slab orientations."""
    from deeplocalproteindocking_amd._lib import get_lib
    from deeplocalproteindocking_amd.engine import _ptr, _stream
    lib = get_lib()
    L, CT, nb = 64, 3, 4
    g = torch.Generator().manual_seed(4)
    vol = torch.randn(CT, L, L, L, generator=g).to(dev)
    quads = torch.empty(lib.call("dlpd_quads_floats", CT, L), dtype=torch.float32, device=dev)
    lib.call("dlpd_make_quads", _ptr(vol), _ptr(quads), CT, L, _stream(dev))
    q = quads.reshape(CT, L, L - 1, L - 1, 4)
    assert torch.equal(q[..., 0], vol[:, :, :-1, :-1]) and torch.equal(q[..., 3], vol[:, :, 1:, 1:])
    assert torch.equal(q[..., 1], vol[:, :, :-1, 1:]) and torch.equal(q[..., 2], vol[:, :, 1:, :-1])
    R = torch.from_numpy(_rots(nb, seed=5)).float().to(dev).contiguous()
    n = nb * CT * (L + 1) * L * L * 2
    for tr in (0, 1):
        a, b = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        lib.call("dlpd_zfft_oriented", _ptr(vol), _ptr(R), _ptr(a), nb, CT, CT, 0, L, 0, 1, L / 2.0, tr, _stream(dev))
        lib.call("dlpd_zfft_quads", _ptr(quads), _ptr(R), _ptr(b), nb, CT, CT, 0, L, L / 2.0, tr, _stream(dev))
        assert (a - b).abs().max() <= 1e-6 * a.abs().max()


@pytest.mark.parametrize("D", [80, 37, 6])
def test_maxpool3d_matches_torch(dev, D):
    """dlpd_maxpool3d_5s2 against torch's MaxPool3d(5, stride 2, padding 2) (bit-exact: a max of the same values)."""
    from deeplocalproteindocking_amd import ops
    x = torch.randn(2, 3, D, D, D, generator=torch.Generator().manual_seed(D)).to(dev)
    want = torch.nn.functional.max_pool3d(x.cpu(), kernel_size=5, stride=2, padding=2)
    assert torch.equal(ops.maxpool3d_5s2(x).cpu(), want)


def _rotations_by_group(need_each=1, seed=2024):
    """Rotations drawn at random and sorted into the four search groups of DockingEngine.search
    (slab orientation x gather layout): ``need_each`` per group."""
    from deeplocalproteindocking_amd.engine import DockingEngine
    R = _rots(400, seed=seed)
    tr, qd = DockingEngine.prefers_transposed(R), DockingEngine.prefers_quads(R)
    out = {}
    for t in (False, True):
        for q in (False, True):
            sel = np.nonzero((tr == t) & (qd == q))[0][:need_each]
            assert len(sel) == need_each
            out[(t, q)] = R[sel]
    return out


def _assert_scores_match(V, Vo, norm, thr):
    sure = (norm - thr).abs() > 1e-3 * thr
    assert 0.2 < (Vo != 0).float().mean() <= 1.0
    err = (V - Vo).abs()[sure].max().item()
    assert err <= TOL * Vo.abs().max().item(), (err, Vo.abs().max().item())
    return err / Vo.abs().max().item()


def test_all_four_search_paths_match_oracle_at_baseline_config2_size(dev):
    """BASELINE config 2 (48 ch, 64^3 -> 128^3) through every K1/K2 variant the search actually runs:
    slab orientation (transposed: K1 swaps x/y, K2 un-transposes while staging, N = 128 template) x gather
    layout (quads).  Each variant is fed (i) a rotation the search WOULD route to it and (ii) one it would
    not (the flags are speed-only: any rotation must give the same scores), and compared with the oracle
    under the stated rule |V_hip - V_oracle| <= 1e-4 max|V| (Docker.py:211-236)."""
    import bench
    from deeplocalproteindocking_amd.engine import DockingEngine
    C, L = 48, 64
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    thr = bench.clash_threshold(recf, ligf)
    W = [w.cpu() for w in filt.parameters_tuple()]
    groups = _rotations_by_group(1)
    oracle = {}
    for key, Rg in groups.items():
        oracle[key] = _oracle_V(rec, lig, recf, ligf, W, Rg[0], thr, 5.0)
    worst = 0.0
    # the default K1: channels-last gather (one kernel for every rotation), on a rotation of each group
    eng = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, max_conf=100, batch=4, device=dev)
    assert eng.use_cl and not eng.orient and not eng.use_quads
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    keys = list(groups)
    V = eng.score_batch(torch.from_numpy(np.stack([groups[k][0] for k in keys])).float().to(dev).contiguous()).cpu()
    for j, key in enumerate(keys):
        worst = max(worst, _assert_scores_match(V[j], oracle[key][0], oracle[key][1], thr))
    del eng
    # the per-channel K1 (ligands with few channels, DLPD_NO_CHANNELS_LAST) and its four launch variants
    eng = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, max_conf=100, batch=2, device=dev, channels_last=False)
    assert eng.orient and eng.use_quads
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    for (tr, qd) in groups:                                      # the launch variant under test
        prefers, other = groups[(tr, qd)][0], groups[(not tr, not qd)][0]
        Rd = torch.from_numpy(np.stack([prefers, other])).float().to(dev).contiguous()
        V = eng.score_batch(Rd, transposed=tr, quads=qd).cpu()
        for j, key in enumerate(((tr, qd), (not tr, not qd))):
            worst = max(worst, _assert_scores_match(V[j], oracle[key][0], oracle[key][1], thr))
    print("config 2, channels-last K1 + four per-channel K1/K2 variants x preferred/non-preferred rotations: "
          "worst error %.2e of max|V|" % worst)


def test_baseline_config5_shape_48ch_80cube_matches_oracle(dev, variants):
    """BASELINE config 5's literal shape: 48 channels at 80^3, single resolution -> 160^3 (channels-last K1, the
    four-sub-problem K2 with 49 slabs per kz, fused K3 on 8-row tiles; the unfused z-inverse + k_filter_vec pair and
    the per-channel K1 with transposed slabs as variants), clip active, clash channel on; two rotations."""
    import bench
    from deeplocalproteindocking_amd.engine import DockingEngine
    C, L = 48, 80
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    thr = bench.clash_threshold(recf, ligf)
    W = [w.cpu() for w in filt.parameters_tuple()]
    groups = _rotations_by_group(1, seed=5)
    eng = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, max_conf=100, batch=2, device=dev)
    assert eng.CT == 49
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    worst = 0.0
    for key in ((False, False), (True, True)):
        R1 = groups[key][0]
        Vo, norm = _oracle_V(rec, lig, recf, ligf, W, R1, thr, 5.0)
        assert float((Vo.abs() > 0).float().mean()) > 0.2
        Rd = torch.from_numpy(R1[None]).float().to(dev).contiguous()
        V = eng.score_batch(Rd, transposed=key[0], quads=key[1]).cpu()
        worst = max(worst, _assert_scores_match(V[0], Vo, norm, thr))
        if key == (False, False):                            # the unfused pair (engine option) on the same rotation
            eng2 = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, max_conf=100, batch=1, device=dev, fine_unfused=True)
            eng2.set_receptor(rec, recf)
            eng2.set_ligand(lig, ligf)
            worst = max(worst, _assert_scores_match(eng2.score_batch(Rd).cpu()[0], Vo, norm, thr))
            del eng2
        else:                                                # the per-channel K1 with transposed slabs + quad layout
            # (K2's transposed-slab reader at N = 160 is a test variant: the product visits box 80 in one orientation)
            eng3 = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, max_conf=100, batch=1, device=dev, channels_last=False,
                                 lib=variants)
            assert eng3.orient
            prod = DockingEngine(L, 4, torch.zeros(2, 4), torch.zeros(2), torch.zeros(1, 2), torch.zeros(1), max_conf=10, batch=1,
                                 device=dev, channels_last=False)
            assert not prod.orient and prod.use_quads       # libdlpd.so: no orientation at box 80
            del prod
            eng3.set_receptor(rec, recf)
            eng3.set_ligand(lig, ligf)
            worst = max(worst, _assert_scores_match(eng3.score_batch(Rd, transposed=True, quads=True).cpu()[0], Vo, norm, thr))
            del eng3
    print("config 5 shape (48 ch @ 80^3): worst error %.2e of max|V|" % worst)
    # the clip must really bite at this amplitude, otherwise the clamp path is not exercised
    c = orc.correlate_fft(rec[None], orc.rotate_volume(lig[None], torch.from_numpy(groups[(False, False)][0][None]).float()))
    assert float((c.abs() > 5.0).float().mean()) > 1e-3


def test_volume_convolution_at_uncompiled_boxes(dev):
    """The stand-alone correlation op at box 50 (inside the 64 plan) and 72 (inside 80), and the plan-free route."""
    from test_kernels_emu import _volume_convolution_uncompiled_box
    _volume_convolution_uncompiled_box(None, dev, 50)
    _volume_convolution_uncompiled_box(None, dev, 72, B=1, C=2)


def test_two_rank_sharded_search_on_one_gpu_equals_single_process(dev, tmp_path):
    """SURVEY 8(e) end to end in two PROCESSES on the one GPU of this box: interleaved rotation shards searched by the
    HIP pipeline in each rank, one all-gather of the per-rank lists (gloo transport here -- RCCL needs one device per
    rank and is covered by the next test where two GPUs exist), the deterministic merge on every rank; the result must
    equal the single-process list entry for entry."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = str(tmp_path / "lists.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29643", os.path.join(root, "scripts", "shard_check.py"), "--out", out,
           "--backend", "gloo", "--same_device", "1", "--nrot", "50"]
    r = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(out))
    assert res["world"] == 2 and res["backend"] == "gloo"
    assert res["sharded"] == res["single"] and len(res["single"]) == res["K"]


def test_rccl_all_gather_of_the_top_list_runs_on_this_gpu(dev, tmp_path):
    """The collective of SURVEY 8(e) on the ``nccl`` (= RCCL) backend on the one GPU a box has: a process group of one
    rank, the same ``all_gather_top_entries`` Docker and bench.py call, with the early return for a single rank switched
    off -- RCCL initialises, gathers the device-side pack and the merge returns the rank's own list bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29645", os.path.join(root, "scripts", "rccl_one_rank_check.py"), "--out", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(out))
    assert res == {"world": 1, "backend": "nccl", "entries": 2000, "identical": True}


def test_two_rank_nccl_search_equals_single_process(dev, tmp_path):
    """SURVEY 8(e) on hardware: two ranks (one per GPU, RCCL) search interleaved shards of the rotation set
    and all-gather their lists once; the merged list must equal the single-process list.  Needs two GPUs
    (the driver's multi-GPU node); skipped on the one-GPU boxes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL all-gather)")
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = str(tmp_path / "lists.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29641", os.path.join(root, "scripts", "shard_check.py"), "--out", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(out))
    assert res["world"] == 2 and res["backend"] == "nccl"
    assert res["sharded"] == res["single"] and len(res["single"]) == res["K"]


def _bench_line(extra, timeout=1500):
    """`python bench.py ...` as the driver starts it (plain python, no torch.distributed environment) -> its JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0])


def test_bench_multi_rank_leg_runs_the_hip_pipeline_with_two_ranks_on_this_gpu(dev):
    """The code the driver's scaling sweep runs -- bench.py's world > 1 branches: process-group initialisation, the timed
    region ending in the all-gather + merge, the max-over-ranks reduction, `gather_check`, `strong` -- executed with the
    HIP kernels by TWO ranks on the one GPU of this box (gloo transport; RCCL wants one device per rank).  The merged
    list of the gather check must equal the single-process one (same sha256), and the line must carry the contract."""
    common = ["--steps", "4", "--warmup", "2", "--cpu_rotations", "0", "--no_real_shapes", "--sustained_s", "0",
              "--gather_rotations", "256"]
    two = _bench_line(["--gpus", "2", "--backend", "gloo", "--same_device", "--strong_s", "2"] + common)
    one = _bench_line(["--gpus", "1", "--strong_s", "0"] + common)
    assert two["n_gpus"] == 2 and two["steps"] == 4 and two["warmup"] == 2 and two["scaling"] == "weak"
    assert two["config"]["world_size_seen_by_the_collective"] == 2 and two["config"]["collective_backend"] == "gloo"
    assert two["gather_check"]["world_size_seen"] == 2 and one["gather_check"]["world_size_seen"] == 1
    assert two["gather_check"]["list_entries"] == 2000
    assert two["gather_check"]["list_sha256"] == one["gather_check"]["list_sha256"]
    assert two["strong"]["world_size"] == 2 and two["strong"]["rotations"] >= 16 and two["strong"]["list_entries"] == 2000
    assert two["value"] > 0 and two["roofline"]["frac"] > 0 and two["top_entries"] == 2000
    # weak scaling on ONE device: the two ranks share it, so the aggregate stays near the single-rank rate
    assert 0.5 < two["value"] / one["value"] < 1.6
    for line in (one, two):
        assert set(line["setup"]) >= {"host_inputs_s", "device_setup_s"}
        assert abs(line["setup"]["host_inputs_s"] + line["setup"]["device_setup_s"] - line["per_rank_setup_s"]) < 0.5


def test_bench_collectives_run_on_rccl_with_one_rank(dev):
    """bench.py's nccl (= RCCL) branches on the one GPU of this box: `--force_group` makes a one-rank communicator, so
    the process-group initialisation with `device_id`, the barriers, the all-gather of the device-resident list and the
    max-over-ranks reduction on a device tensor all run through RCCL itself; the merged list must equal the ungrouped
    run's (same sha256)."""
    common = ["--gpus", "1", "--steps", "4", "--warmup", "2", "--cpu_rotations", "0", "--no_real_shapes", "--sustained_s", "0",
              "--gather_rotations", "256", "--strong_s", "0"]
    rccl = _bench_line(["--force_group", "--backend", "nccl"] + common)
    plain = _bench_line(common)
    assert rccl["config"]["collective_backend"] == "nccl" and rccl["config"]["world_size_seen_by_the_collective"] == 1
    assert plain["config"]["collective_backend"] is None
    assert rccl["gather_check"]["backend"] == "nccl" and rccl["gather_check"]["list_entries"] == 2000
    assert rccl["gather_check"]["list_sha256"] == plain["gather_check"]["list_sha256"]
    assert rccl["n_gpus"] == 1 and rccl["top_entries"] == 2000 and rccl["value"] > 0


def test_bench_refuses_same_device_with_rccl(dev):
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")},
               WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same_device"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--same_device needs --backend gloo" in (r.stdout + r.stderr)


def test_k1_role_split_equals_the_phased_k1(dev, variants):
    """k_rotate_zfft_cl_rs (gather waves + transform / store waves, one block per CU over a range of work items) bit for
    bit against k_rotate_zfft_cl on the hardware, at the sizes of the BASELINE configs: 48 channels x 64^3 and the
    reference's 16 channels x 80^3 with 16 rotations per launch, 48 x 80^3, a partly filled chunk and an embedded box."""
    from test_kernels_emu import _k1_both_formulations
    from deeplocalproteindocking_amd._lib import get_lib
    lib = variants                   # (the role-split K1 is a test variant: measured slower, kept as the cross-check)
    _k1_both_formulations(lib, dev, 64, 48, 16)
    _k1_both_formulations(lib, dev, 80, 16, 16)
    _k1_both_formulations(lib, dev, 80, 48, 5)
    _k1_both_formulations(lib, dev, 64, 9, 3)
    _k1_both_formulations(lib, dev, 80, 20, 2, extent=50)
    # ... and through the engine: the scores of a batch do not depend on the formulation
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(64, 48)
    from deeplocalproteindocking_amd.engine import DockingEngine
    R = torch.from_numpy(_rots(4)).float().to(dev).contiguous()
    Vs = []
    for form in (1, 2):
        eng = DockingEngine(64, 48, W1, b1, W2, b2, clip=5.0, threshold_clash=4000.0, max_conf=50, batch=4, device=dev, k1_form=form,
                            lib=variants)
        eng.set_receptor(rec, recf)
        eng.set_ligand(lig, ligf)
        assert eng.switches()["k1_form"] == {1: "phased", 2: "role-split"}[form]
        Vs.append(eng.score_batch(R).clone())
    assert torch.equal(Vs[0], Vs[1])
    # the product library holds the phased K1 only: the other formulation is refused, not silently replaced
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        get_lib().call("dlpd_zfft_channels_last_form", 1, 1, 1, 1, 4, 5, 0, 64, 32.0, 0, 2, 0)


def test_no_kernel_writes_outside_its_output_buffers(dev, tmp_path):
    """Every device buffer the host code hands to a kernel as an output (engine workspaces, score volumes, candidate
    lists, top-K buffers, projection and convolution outputs, channels-last / packed copies) sits between two 1 MB guard
    bands here (tests/guard_alloc.py); after whole searches -- dockSE3 and dockE3 from PDB files at box 80 with the
    reference's layer plans, a 48-channel 64^3 engine, an embedded box -- no guard byte may have changed.  (A store past the
    end of a buffer is invisible on one stream and a timing-dependent corruption once a second stream has live data there.)"""
    from guard_alloc import GuardedAllocations
    from synth_pdb import write_protein_like_pdb
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.engine import DockingEngine
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4, GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
    rec_pdb, lig_pdb = str(tmp_path / "rec.pdb"), str(tmp_path / "lig.pdb")
    write_protein_like_pdb(rec_pdb, 150, 21)
    write_protein_like_pdb(lig_pdb, 90, 22)
    R = _rots(40, seed=5)
    problems = []
    for plugin, method in ((SE3MultiResReprScalar, "dockSE3"), (E3MultiResRepr4x4, "dockE3")):
        for seed in range(11, 60):                 # a randomly initialised filter whose scores go NEGATIVE on this pair (an
            torch.manual_seed(seed)                # all-positive one leaves nothing but masked zeros in every list)
            repr_ = plugin(multiplier=8)
            model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=40.0).to(dev)
            with torch.no_grad():
                probe = Docker(model, box_size=80, resolution=1.25, max_conf=50, rotations=R[:2], device=dev, randomize_rot=True,
                               rotation_seed=3)
                getattr(probe, method)(rec_pdb, lig_pdb, batch_size=2)
            best = probe.top_list[0][4]
            probe.release_engine()
            del probe
            if best < -0.05:
                break
        lists = []
        # ... and no result may depend on what an ``empty`` buffer held before its first kernel: NaN bytes, zeros, 0xA5
        for empty_byte in (0xA5, 0xFF, 0x00):
            with GuardedAllocations(empty_byte=empty_byte) as g, torch.no_grad():
                dk = Docker(model, box_size=80, resolution=1.25, max_conf=2000, rotations=R, device=dev, randomize_rot=True, rotation_seed=3)
                getattr(dk, method)(rec_pdb, lig_pdb, batch_size=2)
                assert dk.path == "fused" and len(dk.top_list) == 2000 and len(g.items) > 30
                problems += ["%s: %s" % (method, b) for b in g.check()]
                lists.append(list(dk.top_list))
                dk.release_engine()
            del dk
            torch.cuda.empty_cache()
        assert min(t[4] for t in lists[0]) < 0 and all(t[4] == t[4] for t in lists[1])
        if not (lists[0] == lists[1] == lists[2]):
            problems.append("%s: the ranked list depends on the initial content of an uninitialised buffer (%d / %d of 2000 entries differ)"
                            % (method, sum(a != b for a, b in zip(lists[0], lists[1])), sum(a != b for a, b in zip(lists[0], lists[2]))))
    # the synthetic configuration of the metric (48 ch x 64^3, stored forbidden volume) and an embedded box (50 inside 64)
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(64, 48)
    with GuardedAllocations() as g:
        eng = DockingEngine(64, 48, W1, b1, W2, b2, clip=5.0, threshold_clash=4000.0, max_conf=2000, batch=16, device=dev)
        eng.set_receptor(rec, recf)
        eng.set_ligand(lig, ligf)
        eng.reset_top()
        eng.search(torch.from_numpy(_rots(48, seed=6)))
        assert len(eng.top_entries()[0]) == 2000
        problems += ["engine 48 x 64^3: %s" % b for b in g.check()]
    del eng
    torch.cuda.empty_cache()
    from deeplocalproteindocking_amd.Models import SyntheticRepr
    rec, lig, recf, ligf = _pair(50, 4)[:4]
    torch.manual_seed(6)
    model = GlobalDockingModel(SyntheticRepr((4,)), SimpleFilter([4]), threshold_clash=4000.0).to(dev)
    with GuardedAllocations() as g:
        dk = Docker(model, box_size=50, max_conf=100, rotations=_rots(20, seed=7), device=dev)
        dk.dock_volumes([rec], [lig], recf, ligf, write=False)
        assert dk.path == "embedded"
        problems += ["embedded box 50: %s" % b for b in g.check()]
    assert not problems, "\n".join(problems)


def test_no_result_depends_on_lds_a_kernel_never_wrote(dev, tmp_path):
    """dlpd_debug_poison_lds(1): a kernel that fills the LDS of every CU with NaNs runs before EVERY launch of the library.
    A kernel that reads LDS it did not write itself normally sees the left-over of its own previous block -- stable
    from run to run, small enough to hide inside the 1e-4 tolerance against the oracle, and different as soon as another
    stream's kernels share the CUs (a sweep preparing the next target).  With the poison such a read turns every score
    into a NaN; here the ranked lists of whole searches must be the same entries with and without it."""
    from synth_pdb import write_protein_like_pdb
    from deeplocalproteindocking_amd._lib import get_lib
    from deeplocalproteindocking_amd.Docker import Docker
    from deeplocalproteindocking_amd.engine import DockingEngine
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4, GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
    lib = get_lib()
    # the hook itself: a kernel launched behind the poison finds (nearly) every LDS word it did not write poisoned
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    seen = lib.call("dlpd_debug_poison_selfcheck", counter.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    assert seen >= 900, seen
    rec_pdb, lig_pdb = str(tmp_path / "rec.pdb"), str(tmp_path / "lig.pdb")
    write_protein_like_pdb(rec_pdb, 150, 21)
    write_protein_like_pdb(lig_pdb, 90, 22)
    R = _rots(40, seed=5)

    def both(run):
        out = []
        try:
            for on in (0, 1, 0):
                lib.call("dlpd_debug_poison_lds", on)
                out.append(run())
                torch.cuda.synchronize()
        finally:
            lib.call("dlpd_debug_poison_lds", 0)
        return out

    problems = []
    for plugin, method, seed in ((SE3MultiResReprScalar, "dockSE3", 7), (E3MultiResRepr4x4, "dockE3", 79)):
        torch.manual_seed(seed)
        repr_ = plugin(multiplier=8)
        model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=40.0).to(dev)

        def run():
            with torch.no_grad():
                dk = Docker(model, box_size=80, resolution=1.25, max_conf=2000, rotations=R, device=dev, randomize_rot=True, rotation_seed=3)
                getattr(dk, method)(rec_pdb, lig_pdb, batch_size=2)
            top = list(dk.top_list)
            dk.release_engine()
            return top
        a, b, c = both(run)
        assert a == c and len(a) == 2000
        if a != b:
            nan = sum(1 for t in b if t[4] != t[4])
            problems.append("%s at box 80: %d of 2000 entries differ under the poison (%d NaN scores)" % (method, sum(x != y for x, y in zip(a, b)), nan))
    for L, C in ((64, 48), (32, 4), (40, 8)):
        rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, C)

        def run():
            eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=0.12 * L ** 3, max_conf=500, batch=16, device=dev)
            eng.set_receptor(rec, recf)
            eng.set_ligand(lig, ligf)
            eng.reset_top()
            eng.search(torch.from_numpy(_rots(32, seed=6)))
            ent = eng.top_entries()
            return [tuple(np.asarray(x).tolist()) for x in ent]
        a, b, c = both(run)
        assert a == c
        if a != b:
            problems.append("engine %d ch x %d^3: the list differs under the poison" % (C, L))
    assert not problems, "\n".join(problems)


@pytest.mark.parametrize("co_runner", ["se3_dense_convolution", "e3_tile_occupancy_convolution"])
def test_scores_do_not_change_beside_the_plugin_and_the_radix_select(dev, co_runner):
    """Regression test AND canary of round 5's cross-stream finding (EXPERIMENTS.md R5).  One batch of the reference's real
    shapes is scored again and again while two more streams of the process are kept busy: a representation plugin's bf16 x 3
    convolutions and the engine's own full radix select.
    * co_runner = the SE3 plugin (dense convolution kernel): with `ds_add_u32` in the select's histogram kernel this pair
      changed the coarse grid's K1 / K2 output in 259 of 300 scorings; the histogram now counts without LDS atomics and this
      combination measured clean (0 of 300, profiles/r05_occ_probe3.log).  A HARD assertion: any difference fails.
    * co_runner = the E3 plugin on a protein-like input (the tile-occupancy kernel): the KNOWN-BAD pair -- 298 of 300
      scorings differ on today's hardware (low mantissa bits in lanes 48-63 of a pipeline wave; not root-caused, tracked as an
      open defect in DESIGN.md section 8; stand-alone reproducer: scripts/micro/coresidency_repro.hip).  The product never
      creates this co-residency (one stream for plugin and search), so a difference here is reported as an EXPECTED failure;
      if it stops differing the test passes and the defect entry can be closed.
    The check runs in a FRESH process (scripts/coresidency_canary.py): late in a long session new streams tend to share the
    default stream's hardware queue, the pair is then never co-resident and the known-bad case passed for the wrong reason."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "coresidency_canary.py"), co_runner], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["co_runner"] == co_runner and rec["scorings"] == 80
    report = rec["changed"]
    if co_runner == "e3_tile_occupancy_convolution":
        if report:
            pytest.xfail("KNOWN DEFECT (open, DESIGN.md section 8): the E3 plugin's convolution co-resident with the pipeline "
                         "perturbed %s of 80 scorings per stage -- not a product path (one stream for plugin and search)" % report)
        return                                                     # did not reproduce on this box: passes
    assert not report, "the SE3 plugin's convolution + radix select beside the pipeline changed %s (measured clean in round 5)" % report


def test_k3_role_split_equals_the_channel_owning_k3(dev, variants):
    """k_zifft_filter_rs (dedicated transform / filter waves, the default) bit for bit against k_zifft_filter[_tiles]
    (every wave owns a channel) and against the oracle: 48 channels x 64^3 (13 groups of 4, the last one the clash
    channel alone), the reference's real shapes [16 @ 80^3, 32 @ 40^3] with the coarse pre-activation planes through
    both forms, and 48 channels x 80^3 (five groups of ten)."""
    from test_kernels_emu import _k3_both_formulations
    _k3_both_formulations(variants, dev, 64, 48, 24, 5.0, 5, nb=2)
    _k3_both_formulations(variants, dev, 80, 16, 24, 5.0, 6, C1=32, nb=2)
    _k3_both_formulations(variants, dev, 80, 48, 24, 5.0, 7)
    # the product library holds ONE formulation per box: the other one is refused, not silently replaced
    from deeplocalproteindocking_amd.engine import DockingEngine
    eng = DockingEngine(64, 2, torch.zeros(1, 2), torch.zeros(1), torch.zeros(1, 1), torch.zeros(1), max_conf=4, batch=1, device=dev, k3_form=1)
    eng.set_receptor(torch.zeros(2, 64, 64, 64), torch.zeros(64, 64, 64))
    eng.set_ligand(torch.zeros(2, 64, 64, 64), torch.zeros(64, 64, 64))
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        eng.score_batch(torch.eye(3, device=dev).reshape(1, 3, 3).contiguous())


def test_reference_class_default_hidden_width_48_is_fused(dev):
    """The reference CLASS default (multiplier = 16: [32 @ 80^3, 64 @ 40^3] channels, SimpleFilter hidden width 48,
    ProteinRepresentationModels.py:24,35-36 / DockingModels.py:25-27) on the fused pipeline -- fine K3 and the coarse
    grid's pre-activation kernel with two voxels per filter thread -- against the oracle; and 48 ch x 64^3 at hidden
    width 40 (8-row tiles at N = 128)."""
    from test_kernels_emu import _fused_wide_hidden
    _fused_wide_hidden(None, dev, 80, 32, 64, 48, 2, 93)
    _fused_wide_hidden(None, dev, 64, 48, 0, 40, 2, 94)


def test_topk_candidate_lists_from_k3_equal_the_full_select(dev, monkeypatch):
    """The candidate path of the top-K stage at BASELINE config 2 size (48 ch, 64^3, K = 2000, 60 rotations in batches
    of 16) and on the N = 160 tile-walking K3 (16 ch at 80^3): identical ranked lists with and without it."""
    from test_kernels_emu import _search_with_and_without_candidate_lists
    _search_with_and_without_candidate_lists(None, dev, 64, 48, 2000, 60, 16, monkeypatch)
    _search_with_and_without_candidate_lists(None, dev, 80, 16, 500, 20, 8, monkeypatch, seed=4)
    # max_conf above 4096: K3's lists (4096 slots per rotation) overflow until the threshold has tightened, the large-list
    # select and merge run in global scratch; both routes must still agree entry for entry
    _search_with_and_without_candidate_lists(None, dev, 64, 48, 6000, 60, 16, monkeypatch, seed=6)


def test_four_degree_rotation_set_head_and_tail_match_the_oracle(dev):
    """BASELINE config 3's rotation set (``angle_inc=4``: data/oim04.eul is absent from the reference tree,
    .MISSING_LARGE_BLOBS:1, and from the table at src/Utils/Rotations.py:42-55; the SOI-sized generated substitute of
    232,020 rotations stands in) on config 3's 48-channel 64^3 pair: the first 32 and the last 32 rotations of the set,
    searched under their GLOBAL indices as a rank of the sharded search would (rot_ids), K = 200, against the oracle."""
    import bench
    from deeplocalproteindocking_amd.engine import DockingEngine
    from deeplocalproteindocking_amd.Utils.Rotations import Rotations, generated_set_size
    rot = Rotations(4, allow_generated=True, verbose=False)
    nsphere, nphi = generated_set_size(4)
    nrot = rot.R.shape[0]
    assert nrot == nsphere * nphi == 232020 or rot.source != "generated"
    ids = np.concatenate([np.arange(32), np.arange(nrot - 32, nrot)])
    R = rot.R[ids].numpy()
    L, C, K = 64, 48, 200
    rec, lig, recf, ligf, filt = bench.synthetic_pair(C, L)
    W = filt.parameters_tuple()
    thr = bench.clash_threshold(recf, ligf)
    eng = DockingEngine(L, C, *W, clip=5.0, threshold_clash=thr, max_conf=K, batch=16, device=dev)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    eng.reset_top()
    eng.search(rot.R[ids], rot_ids=ids)
    got = eng.top_list()
    want, Vs = orc.dock_volumes([rec[None]], [lig[None]], recf[None, None], ligf[None, None], R, *[w.cpu() for w in W], thr, K,
                                clip=5.0, faithful_topk=False, return_V=True)
    want = [(int(ids[w[0]]),) + w[1:] for w in want]            # oracle indices are positions in R: -> global indices
    scale = max(float(v.abs().max()) for v in Vs)
    band = TOL * scale
    assert len(got) == len(want) == K and {g[0] for g in got} <= set(ids.tolist())
    assert max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= band
    assert sum(a[:4] == b[:4] for a, b in zip(got, want)) >= int(0.97 * K)
    assert any(g[0] >= nrot - 32 for g in got) and any(g[0] < 32 for g in got)

def test_packed_receptor_is_invisible_in_k2(dev):
    """K2 of boxes 80 and 40 on the receptor spectrum in the natural layout and in the packed order of its column phase
    (dlpd_receptor_pack / dlpd_xy_correlate_packed, what the engine launches): the correlation spectra bit for bit, at the
    real shapes' channel counts and batch; and an engine with the packed copies switched off returns the same scores."""
    from test_kernels_emu import _packed_receptor_equals_natural
    from deeplocalproteindocking_amd._lib import get_lib
    from deeplocalproteindocking_amd.engine import DockingEngine
    lib = get_lib()
    _packed_receptor_equals_natural(lib, dev, 80, 17, 16)
    _packed_receptor_equals_natural(lib, dev, 40, 32, 16)
    L, C, C1, H = 80, 4, 6, 8
    g = torch.Generator().manual_seed(12)
    rec, lig = torch.randn(C, L, L, L, generator=g) * 0.05, torch.randn(C, L, L, L, generator=g) * 0.05
    rec1, lig1 = torch.randn(C1, 40, 40, 40, generator=g) * 0.1, torch.randn(C1, 40, 40, 40, generator=g) * 0.1
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    W1, b1 = torch.randn(H, C + C1, generator=g) * 0.4, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    R = torch.from_numpy(orc.euler_to_matrix([0.5, -2.1], [1.1, 0.3], [-0.4, 1.7])).float().to(dev).contiguous()
    out = []
    for packed in (True, False):
        eng = DockingEngine(L, C, W1, b1, W2, b2, clip=0.6, threshold_clash=0.125 * L ** 3, max_conf=10, batch=2, device=dev,
                            coarse_channels=C1, packed_receptor=packed)
        assert eng.switches()["k2_packed_receptor"] == {"fine": packed, "coarse": packed}
        eng.set_receptor(rec, recf, rec1)
        eng.set_ligand(lig, ligf, lig1)
        out.append(eng.score_batch(R).clone())
    assert torch.equal(out[0], out[1])

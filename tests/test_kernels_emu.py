"""The HIP kernel SOURCES executed on the CPU by the fiber emulator (tests/emu) and checked against
the oracle and the golden fixtures.  This validates index logic / barrier structure without a GPU;
the `-m gpu` tests repeat the checks on the real gfx950 build."""
import numpy as np
import pytest
import torch

from oracle import docking_oracle as orc
from deeplocalproteindocking_amd.engine import DeviceTopList, DockingEngine, _ptr


def _pair(L, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    rec = torch.randn(C, L, L, L, generator=g) * 0.1
    lig = torch.randn(C, L, L, L, generator=g) * 0.1
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    H = C // 2
    W1, b1 = torch.randn(H, C, generator=g), torch.randn(H, generator=g)
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    return rec, lig, recf, ligf, W1, b1, W2, b2


def test_fused_pipeline_matches_oracle(emu):
    L, C = 32, 4
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, C)
    thr = 4000.0
    R = orc.euler_to_matrix([0.3, -1.0], [1.1, 0.4], [-2.0, 2.5])
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=0.5, threshold_clash=thr, max_conf=16, batch=2, device="cpu", lib=emu)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    ref = torch.fft.rfftn(torch.cat([rec, recf[None]]), s=(2 * L,) * 3, dim=(1, 2, 3)) / (2 * L) ** 3
    mine = torch.view_as_complex(eng.recF).permute(0, 2, 3, 1)
    assert (mine - ref).abs().max() < 1e-6 * ref.abs().max() + 1e-8
    V = eng.score_batch(torch.from_numpy(R).float().contiguous()).clone()
    for i in range(2):
        Rb = torch.from_numpy(R[i:i + 1]).float()
        lr = orc.rotate_volume(lig[None], Rb)
        lfr = orc.rotate_volume(ligf[None, None], Rb)
        mask, norm = orc.clash_mask(recf[None, None], lfr, thr)
        Vo = (mask * orc.score_volumes([rec[None]], [lr], W1, b1, W2, b2, clip=0.5))[0]
        sure = (norm[0] - thr).abs() > 1e-3 * thr                       # away from the threshold
        assert 0.5 < mask.mean() < 0.999                                 # the mask is exercised
        assert ((V[i] - Vo).abs()[sure]).max() <= 1e-4 * Vo.abs().max()


def test_fused_pipeline_radix5_grid(emu):
    """L = 40 (N = 80 = 10 x 8): radix-5/10 butterflies, partly idle passes, no register hand-over."""
    L, C = 40, 3
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, 4, seed=7)
    rec, lig, W1 = rec[:C], lig[:C], W1[:, :C]
    thr = 0.125 * L ** 3
    R = orc.euler_to_matrix([0.9], [0.7], [-1.4])
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=thr, max_conf=16, batch=1, device="cpu", lib=emu)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    V = eng.score_batch(torch.from_numpy(R).float().contiguous()).clone()
    Rb = torch.from_numpy(R).float()
    mask, norm = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
    Vo = (mask * orc.score_volumes([rec[None]], [orc.rotate_volume(lig[None], Rb)], W1, b1, W2, b2, clip=5.0))[0]
    sure = (norm[0] - thr).abs() > 1e-3 * thr
    assert ((V[0] - Vo).abs()[sure]).max() <= 1e-4 * Vo.abs().max()


def test_pipeline_real_box_size(emu):
    """L = 80 (N = 160, the reference's box_size): the slab does not fit LDS, so K2 runs the
    decimation-in-frequency kernel (two half-width passes, G0 parked in registers) and the filter is
    unfused; two rotations exercise the persistent loop's register prefetch of the next A slab."""
    L, C = 80, 1
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, 4, seed=9)
    rec, lig, W1 = rec[:C], lig[:C], W1[:, :C]
    thr = 0.125 * L ** 3
    R = orc.euler_to_matrix([0.9, -0.3], [0.7, 1.9], [-1.4, 0.2])
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=thr, max_conf=16, batch=2, device="cpu", lib=emu)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    V = eng.score_batch(torch.from_numpy(R).float().contiguous()).clone()
    for i in range(2):
        Rb = torch.from_numpy(R[i:i + 1]).float()
        mask, norm = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
        Vo = (mask * orc.score_volumes([rec[None]], [orc.rotate_volume(lig[None], Rb)], W1, b1, W2, b2, clip=5.0))[0]
        sure = (norm[0] - thr).abs() > 1e-3 * thr
        assert ((V[i] - Vo).abs()[sure]).max() <= 1e-4 * Vo.abs().max()


def test_search_with_odd_tail_matches_oracle_list(emu):
    L, C, K = 32, 4, 40
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, C, seed=1)
    R = orc.euler_to_matrix([0.3, -1.0, 2.0], [1.1, 0.4, 2.2], [-2.0, 2.5, 0.1])
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=4000.0, max_conf=K, batch=2, device="cpu", lib=emu)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    eng.reset_top()
    eng.search(R)
    got = eng.top_list()
    want, Vs = orc.dock_volumes([rec[None]], [lig[None]], recf[None, None], ligf[None, None], R, W1, b1, W2, b2,
                                4000.0, K, clip=5.0, faithful_topk=False, return_V=True)
    scale = max(float(v.abs().max()) for v in Vs)
    assert len(got) == K
    assert max(abs(a[4] - b[4]) for a, b in zip(got, want)) <= 1e-4 * scale
    assert sum(a[:4] == b[:4] for a, b in zip(got, want)) >= K - 4      # swaps only inside the error band


@pytest.mark.parametrize("case", ["randn8_k5", "randn16_k40", "onehot_k4", "allpos_k4", "ties_k30", "fewneg_k6",
                                  "masked_k12"])
def test_topk_kernels_reproduce_reference_update_top(emu, golden, case):
    g = golden("g3_update_top.npz")
    V = torch.from_numpy(g[case + "_V"].copy())
    K, N = int(g[case + "_K"]), V.shape[0]
    top = DeviceTopList(K, 1, "cpu", emu)
    top.reset()
    top.select(V.reshape(1, -1), 1)
    top.merge(torch.tensor([7], dtype=torch.int32), 1)
    got = DeviceTopList.to_top_list(top.entries(), N)
    ref = [(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(b[4])) for b in g[case + "_top"]]
    assert got == ref
    assert [np.signbit(a[4]) for a in got] == [np.signbit(b[4]) for b in ref]


def test_topk_merge_sequence_reproduces_reference(emu, golden):
    g = golden("g3_update_top.npz")
    K = int(g["seq_K"])
    for batch in (1, 3):
        top = DeviceTopList(K, batch, "cpu", emu)
        top.reset()
        Vs = torch.from_numpy(g["seq_V"].copy())
        for beg in range(0, Vs.shape[0], batch):
            nb = min(batch, Vs.shape[0] - beg)
            top.select(Vs[beg:beg + nb].reshape(nb, -1).contiguous(), nb)
            top.merge(torch.arange(beg, beg + nb, dtype=torch.int32), nb)
        got = DeviceTopList.to_top_list(top.entries(), Vs.shape[1])
        assert got == [(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(b[4])) for b in g["seq_top"]]


def test_topk_many_rotations_with_flushes(emu):
    """More candidates than the merge kernel's staging capacity: forces mid-batch flushes."""
    K, N, nrot = 100, 12, 7
    g = torch.Generator().manual_seed(11)
    Vs = torch.randn(nrot, N, N, N, generator=g)
    Vs[3] = torch.randint(-2, 2, (N, N, N), generator=g).float()
    top = DeviceTopList(K, nrot, "cpu", emu)
    top.reset()
    top.select(Vs.reshape(nrot, -1).contiguous(), nrot)
    top.merge(torch.arange(nrot, dtype=torch.int32), nrot)
    got = DeviceTopList.to_top_list(top.entries(), N)
    want = []
    for r in range(nrot):
        want = orc.update_top(want, Vs[r].clone(), r, K)
    assert got == [(r, x, y, z, float(np.float32(s))) for r, x, y, z, s in want]


def _topk_large_k(lib, device, N, K, nrot):
    """max_conf above 4096 (Docker.py:18 accepts any): the sorts of select and merge run in global scratch instead of
    LDS -- per-rotation picks against the vectorised oracle, the merged list against the closed form of the update_top
    sequence (all picks sorted by (score, rotation, pick)), with ties and a rotation that forces zero-fill."""
    g = torch.Generator().manual_seed(19)
    V = torch.randn(nrot, N, N, N, generator=g)
    V[1] = torch.round(V[1] * 4) / 4                                         # ties
    if nrot > 2:
        V[2] = V[2].abs() * (torch.rand(N, N, N, generator=g) > 0.999)       # almost no negative score: zero-fill
    top = DeviceTopList(K, nrot, device, lib)
    top.reset()
    cs, ci = top.select(V.to(device).reshape(nrot, -1).contiguous(), nrot)
    cs, ci = cs.cpu().numpy().copy(), ci.cpu().numpy().copy()
    picks = []
    for j in range(nrot):
        idx, sc = orc.rotation_picks_fast(V[j].numpy(), K)
        assert np.array_equal(ci[j], idx) and np.array_equal(cs[j].view(np.uint32), sc.view(np.uint32))
        picks += [(float(sc[i]), j, i, int(idx[i])) for i in range(K)]
    top.merge(torch.arange(nrot, dtype=torch.int32, device=device), nrot)
    # a second batch whose scores mostly lose against the full list: few survivors, the steady-state merge (sorted run of
    # new entries ranked into the sorted list)
    V2 = torch.randn(nrot, N, N, N, generator=g) + 2.0
    cs, ci = top.select(V2.to(device).reshape(nrot, -1).contiguous(), nrot)
    cs, ci = cs.cpu().numpy().copy(), ci.cpu().numpy().copy()
    for j in range(nrot):
        idx, sc = orc.rotation_picks_fast(V2[j].numpy(), K)
        assert np.array_equal(ci[j], idx) and np.array_equal(cs[j].view(np.uint32), sc.view(np.uint32))
        picks += [(float(sc[i]), nrot + j, i, int(idx[i])) for i in range(K)]
    top.merge(torch.arange(nrot, 2 * nrot, dtype=torch.int32, device=device), nrot)
    rot, idx, score, pick = top.entries()
    picks.sort(key=lambda p: (p[0], p[1], p[2]))
    want = picks[:K]
    assert 0 < sum(p[1] >= nrot for p in want) < K // 4                      # some, but few, of the second batch made it
    assert rot.tolist() == [p[1] for p in want] and idx.tolist() == [p[3] for p in want]
    assert np.array_equal(np.asarray(score, dtype=np.float32), np.asarray([p[0] for p in want], dtype=np.float32))


def test_topk_with_more_than_4096_conformations_emulated(emu):
    _topk_large_k(emu, "cpu", 20, 4500, 3)


def test_rotate_kernel_matches_oracle(emu):
    torch.manual_seed(2)
    B, C, L = 2, 3, 10
    vol = torch.randn(B, C, L, L, L)
    R = torch.from_numpy(orc.euler_to_matrix([0.4, -1.3], [0.9, 2.0], [1.7, -0.2])).float().contiguous()
    out = torch.empty_like(vol)
    emu.call("dlpd_rotate_trilinear", _ptr(vol), _ptr(R), _ptr(out), B, C, L, C * L ** 3, L / 2.0, 0)
    assert (out - orc.rotate_volume(vol, R)).abs().max() < 1e-5
    ident = torch.eye(3).repeat(B, 1, 1).contiguous()
    emu.call("dlpd_rotate_trilinear", _ptr(vol), _ptr(ident), _ptr(out), B, C, L, C * L ** 3, L / 2.0, 0)
    assert torch.equal(out, vol)


@pytest.mark.parametrize("L,C", [(32, 20), (40, 16), (64, 9)])
def test_channels_last_rotation_equals_the_per_channel_kernel(emu, L, C):
    """dlpd_zfft_channels_last == dlpd_zfft_into(do_rotate=1), bit for bit, on oblique rotations; channel counts that
    are not a multiple of the 16-channel block (zero-padded copy) and a workspace with more channels than written."""
    torch.manual_seed(12)
    nb, NZ, CT = (2 if L < 64 else 1), L + 1, C + 1       # L = 64: the 16-row x 8-channel block shape of N = 128
    vol = torch.randn(C, L, L, L)
    R = torch.from_numpy(orc.euler_to_matrix([0.4, -1.3], [0.9, 2.0], [1.7, -0.2])).float()[:nb].contiguous()
    want = torch.zeros(nb * CT * NZ * L * L * 2)
    got = torch.zeros_like(want)
    emu.call("dlpd_zfft_into", _ptr(vol), _ptr(R), _ptr(want), nb, C, CT, 0, L, 0, 1, L / 2.0, 0)
    cl = torch.empty(emu.call("dlpd_channels_last_floats", C, L))
    emu.call("dlpd_make_channels_last", _ptr(vol), _ptr(cl), C, L, 0)
    Cp = cl.numel() // L ** 3
    assert Cp % 16 == 0 and Cp >= C
    assert torch.equal(cl.view(L, L, L, Cp)[..., :C], vol.permute(1, 2, 3, 0)) and not cl.view(L, L, L, Cp)[..., C:].any()
    emu.call("dlpd_zfft_channels_last", _ptr(cl), _ptr(R), _ptr(got), nb, C, CT, 0, L, L / 2.0, 0)
    assert torch.equal(got, want)


def _k1_both_formulations(lib, device, L, C, nb, extent=0, seed=13):
    """K1 of the channels-last path through both kernel formulations (include/dlpd.h, dlpd_zfft_channels_last_form):
    1 = every wave gathers, transforms and stores in turn; 2 = gather waves + transform / store waves (one block per CU
    walking a range of work items).  Same samples, same butterflies: the spectra must be the same BITS."""
    g = torch.Generator().manual_seed(seed)
    NZ, CT = L + 1, C + 1
    vol = torch.randn(C, L, L, L, generator=g).to(device)
    ang = np.random.RandomState(seed).uniform(-np.pi, np.pi, size=(nb, 3))
    R = torch.from_numpy(orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])).float().contiguous().to(device)
    st = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    cl = torch.empty(lib.call("dlpd_channels_last_floats", C, L), device=device)
    lib.call("dlpd_make_channels_last", _ptr(vol), _ptr(cl), C, L, st)
    outs = []
    for form, fill in ((1, 7.0), (2, 7.0), (2, -3.0)):
        out = torch.full((nb * CT * NZ * L * L * 2,), fill, device=device)      # (channel C of the workspace stays untouched)
        lib.call("dlpd_zfft_channels_last_form", _ptr(cl), _ptr(R), _ptr(out), nb, C, CT, 0, L, L / 2.0 if not extent else extent / 2.0,
                 extent, form, st)
        outs.append(out.view(nb, CT, NZ, L, L, 2))
    assert torch.equal(outs[0], outs[1])
    # every element of channels [0, C) is written (the result does not depend on what the buffer held), channel C is not
    assert torch.equal(outs[1][:, :C], outs[2][:, :C]) and float(outs[1][:, :C].abs().max()) > 1.0
    assert bool((outs[1][:, C] == 7.0).all()) and bool((outs[2][:, C] == -3.0).all())
    return outs[1]


@pytest.mark.parametrize("L,C,nb,extent", [(64, 9, 1, 0), (80, 16, 1, 0), (80, 20, 1, 50)])
def test_k1_role_split_equals_the_phased_k1(emu, L, C, nb, extent):
    """Boxes 64 (two input buffers, one barrier per work item) and 80 (one input buffer, two barriers; the radix-16 x 10
    plan run on 8 lanes per pencil), channel counts that leave a partly filled last chunk, an embedded box (crop)."""
    _k1_both_formulations(emu, "cpu", L, C, nb, extent)
    if L == 64:
        with pytest.raises(RuntimeError, match="UNSUPPORTED"):                  # no role-split kernel for the small boxes
            emu.call("dlpd_zfft_channels_last_form", 1, 1, 1, 1, 4, 5, 0, 32, 16.0, 0, 2, 0)


@pytest.mark.parametrize("L,nvol", [(32, 2), (64, 1)])
def test_volume_convolution_stages_match_oracle_and_definition(emu, L, nvol):
    """rfft3d_padded + zfft + xy_correlate + zifft_real == VolumeConvolution; spot-checked against
    the MultiplyVolumes definition (sum_r v1[r+t] v2[r]).  L = 64: the N = 128 slab kernel of the headline
    configuration (last inverse pass written straight to global memory)."""
    torch.manual_seed(4)
    N, NZ = 2 * L, L + 1
    v1, v2 = torch.randn(nvol, L, L, L), torch.randn(nvol, L, L, L)
    wsA = torch.empty(nvol * NZ * L * L * 2)
    spec = torch.empty(nvol * NZ * N * N * 2)
    wsB = torch.empty(nvol * NZ * N * N * 2)
    out = torch.empty(nvol, N, N, N)
    emu.call("dlpd_rfft3d_padded", _ptr(v1), _ptr(spec), _ptr(wsA), nvol, L, 1.0 / N ** 3, 0)
    emu.call("dlpd_zfft", _ptr(v2), 0, _ptr(wsA), 1, nvol, L, 0, 0, 0.0, 0)
    emu.call("dlpd_xy_correlate", _ptr(wsA), _ptr(spec), _ptr(wsB), 1, nvol, L, 0, 0)
    emu.call("dlpd_zifft_real", _ptr(wsB), _ptr(out), 1, nvol, L, 0, 0.0, 0)
    ref = orc.correlate_fft(v1[None], v2[None], dtype=torch.float64)[0]
    assert (out.double() - ref).abs().max() < 1e-5 * ref.abs().max()
    for t in [(0, 0, 0), (3, -2, 5), (1 - L, L - 1, 0), (7, 7, -7)]:
        sl1 = tuple(slice(max(d, 0), L + min(d, 0)) for d in t)
        sl2 = tuple(slice(max(-d, 0), L + min(-d, 0)) for d in t)
        direct = (v1[(slice(None),) + sl1].double() * v2[(slice(None),) + sl2].double()).sum(dim=(1, 2, 3))
        got = out[:, t[0] % N, t[1] % N, t[2] % N].double()
        assert (got - direct).abs().max() < 1e-3
    assert out[:, L, :, :].abs().max() < 1e-3          # |t| = L: no overlap
    # clip variant
    emu.call("dlpd_zifft_real", _ptr(wsB), _ptr(out), 1, nvol, L, 1, 2.0, 0)
    assert (out.double() - ref.clamp(-2.0, 2.0)).abs().max() < 1e-4 + 1e-6 * ref.abs().max() and out.abs().max() <= 2.0


def _volume_convolution_uncompiled_box(lib, device, L, B=2, C=3):
    """ops.VolumeConvolution at a box size without a compiled plan, both routes -- inside the next compiled box (default)
    and the plan-free transforms -- against the oracle and the MultiplyVolumes definition at a few translations."""
    from deeplocalproteindocking_amd.ops import VolumeConvolution
    torch.manual_seed(14)
    N = 2 * L
    v1, v2 = torch.randn(B, C, L, L, L), torch.randn(B, C, L, L, L)
    ref = orc.correlate_fft(v1, v2, dtype=torch.float64)
    outs = {}
    for name, embed in (("embedded", True), ("plan-free", False)):
        out = VolumeConvolution(clip=None, lib=lib, embed=embed)(v1.to(device), v2.to(device)).cpu()
        assert out.shape == (B, C, N, N, N)
        assert (out.double() - ref).abs().max() < 1e-5 * ref.abs().max(), name
        assert out[:, :, L].abs().max() < 1e-3 and out[:, :, :, :, L].abs().max() < 1e-3      # |t| = L: no overlap
        outs[name] = out
    for t in [(0, 0, 0), (3, -2, 5), (1 - L, L - 1, 0), (-1, -1, -1)]:
        sl1 = tuple(slice(max(d, 0), L + min(d, 0)) for d in t)
        sl2 = tuple(slice(max(-d, 0), L + min(-d, 0)) for d in t)
        direct = (v1[(slice(None), slice(None)) + sl1].double() * v2[(slice(None), slice(None)) + sl2].double()).sum(dim=(2, 3, 4))
        assert (outs["embedded"][:, :, t[0] % N, t[1] % N, t[2] % N].double() - direct).abs().max() < 1e-3
    clipped = VolumeConvolution(clip=2.0, lib=lib)(v1.to(device), v2.to(device)).cpu()
    assert (clipped.double() - ref.clamp(-2.0, 2.0)).abs().max() < 1e-4 + 1e-6 * ref.abs().max() and clipped.abs().max() <= 2.0


def test_volume_convolution_at_an_uncompiled_box_emulated(emu):
    _volume_convolution_uncompiled_box(emu, "cpu", 10, B=1, C=2)
    # an ODD box: N = 2 L = 18 is not a multiple of the plan-free transform's 4-wide unroll (its spare lanes used to
    # step past the twiddle table)
    _volume_convolution_uncompiled_box(emu, "cpu", 9, B=1, C=1)


@pytest.mark.parametrize("tag,nres", [("multires", 2), ("single", 1)])
def test_generic_filter_kernel_reproduces_reference_forward(emu, golden, tag, nres):
    """dlpd_filter_mask on oracle correlations == reference GlobalDockingModel.forward output (G5):
    pins nearest-upsample index, channel concat order and the MLP."""
    g = golden("g5_global_forward.npz")
    rec = [torch.from_numpy(g["%s_rec%d" % (tag, i)]) for i in range(nres)]
    lig = [torch.from_numpy(g["%s_lig%d" % (tag, i)]) for i in range(nres)]
    conv = [orc.correlate_fft(r, l, clip=float(g[tag + "_clip"])).contiguous() for r, l in zip(rec, lig)]
    W1 = torch.from_numpy(g[tag + "_W1"])
    W1t = W1.t().contiguous()
    b1, W2 = torch.from_numpy(g[tag + "_b1"]), torch.from_numpy(g[tag + "_W2"]).reshape(-1).contiguous()
    B, N0 = conv[0].shape[0], conv[0].shape[2]
    V = torch.empty(B, N0, N0, N0)
    c1 = conv[1] if nres == 2 else None
    emu.call("dlpd_filter_mask", _ptr(conv[0]), conv[0].shape[1], N0, _ptr(c1), c1.shape[1] if nres == 2 else 0,
             c1.shape[2] if nres == 2 else 0, 0, 0.0, 0, _ptr(W1t), _ptr(b1), _ptr(W2), float(g[tag + "_b2"][0]),
             W1.shape[0], _ptr(V), B, 0)
    np.testing.assert_allclose(V.numpy(), g[tag + "_V"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("unfused", [False, True])
def test_fused_two_resolution_pipeline(emu, unfused):
    """[C0 @ 64^3, C1 @ 32^3] -> 128^3 through the fused engine (coarse correlations enter K3 as
    auxiliary real channels, nearest-upsampled by index) against the oracle's GlobalDockingModel.forward.
    unfused: the N = 160 route (real volumes + dlpd_filter_volumes), forced here at N = 128."""
    L, C0, C1 = 64, 2, 3
    g = torch.Generator().manual_seed(41)
    rec0, lig0 = torch.randn(C0, L, L, L, generator=g) * 0.05, torch.randn(C0, L, L, L, generator=g) * 0.05
    rec1 = torch.randn(C1, L // 2, L // 2, L // 2, generator=g) * 0.1
    lig1 = torch.randn(C1, L // 2, L // 2, L // 2, generator=g) * 0.1
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    H = 2
    W1, b1 = torch.randn(H, C0 + C1, generator=g), torch.randn(H, generator=g)
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    thr = 0.125 * L ** 3
    R = orc.euler_to_matrix([0.9], [0.7], [-1.4])
    eng = DockingEngine(L, C0, W1, b1, W2, b2, clip=0.8, threshold_clash=thr, max_conf=16, batch=1, device="cpu",
                        lib=emu, coarse_channels=C1, fine_unfused=unfused)
    eng.set_receptor(rec0, recf, rec1)
    eng.set_ligand(lig0, ligf, lig1)
    V = eng.score_batch(torch.from_numpy(R).float().contiguous()).clone()
    Rb = torch.from_numpy(R).float()
    mask, norm = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
    Vo = (mask * orc.score_volumes([rec0[None], rec1[None]],
                                   [orc.rotate_volume(lig0[None], Rb), orc.rotate_volume(lig1[None], Rb)],
                                   W1, b1, W2, b2, clip=0.8))[0]
    sure = (norm[0] - thr).abs() > 1e-3 * thr
    assert ((V[0] - Vo).abs()[sure]).max() <= 1e-4 * Vo.abs().max()


@pytest.mark.parametrize("precision", ["f32", "split_bf16"])
@pytest.mark.parametrize("cin,cout,ks,D,relu", [(11, 16, 5, 6, True), (16, 32, 3, 9, False), (8, 32, 3, 17, True), (5, 48, 3, 5, False)])
def test_conv3d_mfma_kernel_matches_torch(emu, cin, cout, ks, D, relu, precision):
    """The representation plugin's Conv3d (+ReLU) on the emulated matrix cores, both arithmetic forms -- exact f32
    products (16x16x4 f32) and three bf16 terms per value with six bf16 products per f32 product (16x16x32 bf16) --:
    channel counts that are not multiples of the chunk (zero-padded), box sizes that are not multiples of the 4 x 4
    patch or of the 16-voxel z tile, both kernel sizes, tap counts that are not multiples of the 4-tap group."""
    from deeplocalproteindocking_amd import ops
    g = torch.Generator().manual_seed(5 + cin)
    x = torch.randn(1, cin, D, D, D, generator=g)
    w = torch.randn(cout, cin, ks, ks, ks, generator=g) * 0.1
    assert ops.conv3d_supported(w, D, emu)
    y = ops.conv3d(x, w, relu=relu, lib=emu, precision=precision)
    want = torch.nn.functional.conv3d(x.double(), w.double(), padding=ks // 2)
    want = torch.relu(want) if relu else want
    assert (y.double() - want).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.parametrize("precision", ["f32", "split_bf16"])
@pytest.mark.parametrize("cin,cout,ks,D", [(8, 32, 5, 8), (5, 16, 3, 9), (8, 16, 5, 7)])
def test_conv3d_stride2_matches_torch(emu, cin, cout, ks, D, precision):
    """The stride-2 layer of SE3MultiResReprScalar (ProteinRepresentationModels.py:51): even and odd box sizes."""
    from deeplocalproteindocking_amd import ops
    g = torch.Generator().manual_seed(50 + cin)
    x = torch.randn(2, cin, D, D, D, generator=g)
    w = torch.randn(cout, cin, ks, ks, ks, generator=g) * 0.1
    y = ops.conv3d(x, w, lib=emu, stride=2, precision=precision)
    want = torch.nn.functional.conv3d(x.double(), w.double(), padding=ks // 2, stride=2)
    assert y.shape == want.shape and (y.double() - want).abs().max() <= 1e-5 * want.abs().max()


def test_three_bf16_terms_carry_a_float(emu):
    """The split the bf16 convolution rests on: x = h + m + l with each term a bf16 reproduces a float to 2^-24 relative
    (checked on the host with the same rounding rule the kernels use: round to nearest even on the upper 16 bits)."""
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(20000, generator=g) * 10.0 ** torch.randint(-6, 6, (20000,), generator=g).float(),
                   torch.tensor([0.0, 1.0, -1.0, 3.0e38, 1e-30, 0.1, 255.99998])])

    def bf(v):
        u = v.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
        return (u & 0xFFFFFFFF).to(torch.int64).to(torch.int32).view(torch.float32) if False else \
            torch.from_numpy((u.numpy() & 0xFFFFFFFF).astype("uint32").view("float32").copy())
    h = bf(x)
    m = bf(x - h)
    l = bf(x - h - m)
    err = ((h.double() + m.double() + l.double()) - x.double()).abs()
    assert (err <= 2.0 ** -24 * x.double().abs() + 1e-45).all()


def test_plugins_refuse_a_silent_torch_convolution(emu, monkeypatch):
    """GPU inference (here: the emulated library) must not fall back to torch/MIOpen silently: a layer the HIP
    kernel cannot take raises, unless DLPD_ALLOW_TORCH_CONV=1 (then it warns and runs on torch)."""
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4
    torch.manual_seed(2)
    m = E3MultiResRepr4x4(multiplier=3).eval()                # 6 / 12 output channels: not multiples of 16
    m.hip_lib = emu
    x = torch.rand(1, 11, 4, 4, 4)
    monkeypatch.delenv("DLPD_ALLOW_TORCH_CONV", raising=False)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no HIP kernel"):
            m(x)
        monkeypatch.setenv("DLPD_ALLOW_TORCH_CONV", "1")
        with pytest.warns(UserWarning, match="torch/MIOpen"):
            got = m(x)
        m.hip_lib = None
        want = m(x)
    assert all(torch.equal(a, b) for a, b in zip(got, want))


def test_isotropic_layer_takes_dense_radial_kernels():
    """IsotropicConv3d.load_radial_profile: the hand-over for se3cnn users (dense kernels of a scalar-field
    SE3Convolution): a kernel that is radial on the shells is reproduced exactly, a non-radial one reports
    its residual."""
    from deeplocalproteindocking_amd.Models.ProteinRepresentationModels import IsotropicConv3d
    torch.manual_seed(4)
    src, dst = IsotropicConv3d(3, 4), IsotropicConv3d(3, 4)
    K = src.kernel().detach()
    assert dst.load_radial_profile(K) < 1e-5
    assert torch.allclose(dst.kernel(), K, atol=1e-6)
    assert dst.load_radial_profile(K + 0.05 * torch.randn_like(K)) > 1e-3


def test_se3_representation_takes_the_reference_checkpoint_as_dense_kernels():
    """The tensor contract for loading a reference SE3 checkpoint through se3cnn-evaluated kernels
    (ProteinRepresentationModels.py:38-61): eight tensors named after the reference's module tree, (cout, cin, 5, 5, 5);
    used as they are, the plugin computes exactly conv3d / ReLU with those kernels (stride 2 on sequence_res1.0)."""
    import pytest
    from deeplocalproteindocking_amd.Models import SE3MultiResReprScalar
    for mult in (8, 16):
        m = SE3MultiResReprScalar(multiplier=mult)
        c0, c1 = 2 * mult, 4 * mult
        want = {"sequence_res0.0": (c0, 11, 5, 5, 5), "sequence_res0.2": (c0, c0, 5, 5, 5), "sequence_res0.4": (c0, c0, 5, 5, 5),
                "sequence_res0.6": (c0, c0, 5, 5, 5), "sequence_res1.0": (c1, c0, 5, 5, 5), "sequence_res1.2": (c1, c1, 5, 5, 5),
                "sequence_res1.4": (c1, c1, 5, 5, 5), "sequence_res1.6": (c1, c1, 5, 5, 5)}
        assert m.dense_kernel_contract() == want
    torch.manual_seed(9)
    m = SE3MultiResReprScalar(multiplier=1)                     # [2 @ L^3, 4 @ (L/2)^3]: small enough for torch on the CPU
    kernels = {k: torch.randn(*shape) * 0.05 for k, shape in m.dense_kernel_contract().items()}   # NOT radial
    res = m.load_dense_kernels(kernels)
    assert set(res) == set(kernels) and min(res.values()) > 0.1           # far from the shells: kept exactly anyway
    x = torch.rand(1, 11, 12, 12, 12)
    with torch.no_grad():
        got = m(x)
        y = x
        for i in (0, 2, 4, 6):
            y = torch.nn.functional.conv3d(y, kernels["sequence_res0.%d" % i], padding=2)
            y = torch.relu(y) if i < 6 else y
        z = y
        for i in (0, 2, 4, 6):
            z = torch.nn.functional.conv3d(z, kernels["sequence_res1.%d" % i], padding=2, stride=2 if i == 0 else 1)
            z = torch.relu(z) if i < 6 else z
    assert torch.allclose(got[0], y, atol=1e-6) and torch.allclose(got[1], z, atol=1e-6)
    assert got[0].shape == (1, 2, 12, 12, 12) and got[1].shape == (1, 4, 6, 6, 6)
    with pytest.raises(Exception, match="names do not match"):
        m.load_dense_kernels({k: v for k, v in kernels.items() if k != "sequence_res1.6"})
    with pytest.raises(Exception, match="shape mismatch"):
        m.load_dense_kernels(dict(kernels, **{"sequence_res0.0": torch.zeros(2, 11, 3, 3, 3)}))
    # the handed-over dense kernels are not part of the state dict: a checkpoint of this model loads into a fresh one
    # (strict), and reloading shell coefficients drops the dense kernels instead of being ignored
    sd = m.state_dict()
    assert not any(k.endswith("dense") for k in sd)
    fresh = SE3MultiResReprScalar(multiplier=1)
    fresh.load_state_dict(sd)
    m.load_state_dict(fresh.state_dict())
    with torch.no_grad():
        assert torch.allclose(m(x)[0], fresh(x)[0], atol=1e-6) and not torch.allclose(m(x)[0], y, atol=1e-3)
    # radial kernels survive the projection mode too (exact=False re-parametrises on the shells)
    src = SE3MultiResReprScalar(multiplier=1)
    radial = {k: getattr(src, k.split(".")[0])[int(k.split(".")[1])].kernel().detach() for k in src.dense_kernel_contract()}
    m2 = SE3MultiResReprScalar(multiplier=1)
    assert max(m2.load_dense_kernels(radial, exact=False).values()) < 1e-5
    with torch.no_grad():
        a, b = src(x), m2(x)
    assert torch.allclose(a[1], b[1], atol=1e-5)


def test_topk_kernels_fuzz_against_faithful_update_top(emu):
    """Randomised small volumes built to provoke every corner of Docker.update_top (Docker.py:86-105): many
    exact ties, zeros of both signs, fewer negatives than K, no zero at all, K up to the voxel count, batches
    of several rotations merged into a running list -- the device select + merge must reproduce the reference
    loop tuple for tuple (including the sign of recorded zeros)."""
    rng = np.random.RandomState(20240)
    pools = [np.array([-2.0, -1.0, -1.0, -0.5, 0.0, 0.0, 1.0, 3.0], dtype=np.float32),
             np.array([-1.0, 0.0, -0.0, 2.0], dtype=np.float32),
             np.array([0.5, 1.0, 2.0], dtype=np.float32),                      # no zero, no negative
             np.array([-3.0, -3.0, -3.0, 4.0, 0.0], dtype=np.float32)]
    for trial in range(24):
        N = int(rng.choice([4, 5, 8]))
        K = int(rng.randint(1, min(N ** 3, 40) + 1))
        nrot, batch = int(rng.randint(1, 5)), int(rng.randint(1, 4))
        pool = pools[trial % len(pools)]
        Vs = pool[rng.randint(0, len(pool), size=(nrot, N, N, N))]
        if trial % 5 == 0:
            Vs = (Vs + rng.randn(*Vs.shape).astype(np.float32) * 0.01).astype(np.float32)     # mostly distinct values
        want = []
        for r in range(nrot):
            want = orc.update_top(want, torch.from_numpy(Vs[r].copy()), r, K)
        top = DeviceTopList(K, batch, "cpu", emu)
        top.reset()
        for beg in range(0, nrot, batch):
            nb = min(batch, nrot - beg)
            top.select(torch.from_numpy(Vs[beg:beg + nb].reshape(nb, -1).copy()), nb)
            top.merge(torch.arange(beg, beg + nb, dtype=torch.int32), nb)
        got = DeviceTopList.to_top_list(top.entries(), N)
        assert got == want, (trial, N, K, nrot, batch)
        assert [np.signbit(a[4]) for a in got] == [np.signbit(b[4]) for b in want], trial
        # the same rotations visited in another order (DockingEngine.search groups them by slab orientation):
        # exact ties must still go to the lower rotation id
        order = rng.permutation(nrot)
        top = DeviceTopList(K, batch, "cpu", emu)
        top.reset()
        for beg in range(0, nrot, batch):
            ids = np.sort(order[beg:beg + batch])
            top.select(torch.from_numpy(Vs[ids].reshape(len(ids), -1).copy()), len(ids))
            top.merge(torch.from_numpy(ids.astype(np.int32)), len(ids))
        assert DeviceTopList.to_top_list(top.entries(), N) == want, (trial, "permuted", order)


def test_representation_plugins_route_their_convolutions_through_the_kernel(emu):
    """E3MultiResRepr4x4 / SE3MultiResReprScalar forward with the (emulated) HIP kernels equals the plain torch
    modules: Conv3d+ReLU pairs fused, MaxPool3d and the stride-2 layer on their kernels too."""
    from deeplocalproteindocking_amd.Models import E3MultiResRepr4x4
    from deeplocalproteindocking_amd.Models.ProteinRepresentationModels import IsotropicConv3d
    torch.manual_seed(11)
    x = torch.rand(1, 11, 4, 4, 4)
    m = E3MultiResRepr4x4(multiplier=8).eval()
    with torch.no_grad():
        want = m(x)                                   # CPU tensor, no library: torch path
        m.hip_lib = emu
        got = m(x)
    assert [tuple(v.shape) for v in got] == [(1, 16, 4, 4, 4), (1, 32, 2, 2, 2)]
    for g, w in zip(got, want):
        assert (g - w).abs().max() <= 1e-5 * w.abs().max()
    layer = IsotropicConv3d(11, 16).eval()                     # the SE3 plugin's building block
    with torch.no_grad():
        want1 = layer(x)
        layer.hip_lib = emu
        got1 = layer(x)
    assert (got1 - want1).abs().max() <= 1e-5 * want1.abs().max()
    strided = IsotropicConv3d(16, 32, stride=2).eval()         # ProteinRepresentationModels.py:51
    x2 = torch.rand(1, 16, 6, 6, 6)
    with torch.no_grad():
        want2 = strided(x2)
        strided.hip_lib = emu
        got2 = strided(x2)
    assert got2.shape == want2.shape == (1, 32, 3, 3, 3) and (got2 - want2).abs().max() <= 1e-5 * want2.abs().max()


def _axis_rot(axis, deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


@pytest.mark.parametrize("L", [32, 40, 80])
def test_slab_orientation_is_invisible_in_the_scores(emu, L):
    """A launch may process its rotations with transposed slabs (K1 swaps the roles of x and y, K2
    un-transposes while staging: include/dlpd.h 'slab orientation'): purely a memory-access choice.  The same
    two rotations scored both ways, on the plain (N = 64), radix-5 (N = 80) and half-slab (N = 160) K2
    kernels, must give the oracle's scores; search() picks the orientation per rotation and must return the
    same list as a search with the switch off."""
    C = 2 if L < 80 else 1
    rec, lig, recf, ligf, W1, b1, W2, b2 = _pair(L, 4, seed=21 + L)
    rec, lig, W1 = rec[:C], lig[:C], W1[:, :C]
    thr = 0.125 * L ** 3
    R = np.stack([_axis_rot("x", 77.0) @ _axis_rot("z", 20.0), _axis_rot("y", 80.0) @ _axis_rot("z", -35.0)])
    assert DockingEngine.prefers_transposed(R).tolist() == [False, True]
    if L == 80:
        R = R[1:]                                  # the big grid is slow to emulate: the oblique rotation only
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=thr, max_conf=16, batch=2, device="cpu", lib=emu)
    eng.set_receptor(rec, recf)
    eng.set_ligand(lig, ligf)
    Rt = torch.from_numpy(R).float().contiguous()
    Vn = eng.score_batch(Rt, transposed=False).clone()
    Vt = eng.score_batch(Rt, transposed=True).clone()
    for i in range(R.shape[0]):
        Rb = torch.from_numpy(R[i:i + 1]).float()
        mask, norm = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
        Vo = (mask * orc.score_volumes([rec[None]], [orc.rotate_volume(lig[None], Rb)], W1, b1, W2, b2, clip=5.0))[0]
        sure = (norm[0] - thr).abs() > 1e-3 * thr
        for V in (Vn, Vt):
            assert ((V[i] - Vo).abs()[sure]).max() <= 1e-4 * Vo.abs().max()
    if L == 80:
        return
    # the clash channel supplied separately (dockSE3's re-projection path) follows the same orientation
    eng.clash_provider = lambda Rq: torch.cat([orc.rotate_volume(ligf[None, None], Rq[i:i + 1]) for i in range(Rq.shape[0])])
    V2 = eng.score_batch(Rt, transposed=True).clone()
    assert (V2 - Vt).abs().max() <= 1e-5 * Vt.abs().max()
    eng.clash_provider = None
    if L == 32:                                    # search(): mixed set, orientation chosen per rotation
        eng.reset_top()
        eng.search(R)
        got = eng.top_list()
        eng.orient = False
        eng.reset_top()
        eng.search(R)
        plain = eng.top_list()
        assert {t[0] for t in got} == {0, 1} or len({t[0] for t in got}) >= 1
        assert max(abs(a[4] - b[4]) for a, b in zip(got, plain)) <= 1e-4 * float(Vn.abs().max())
        assert sum(a[:4] == b[:4] for a, b in zip(got, plain)) >= len(got) - 2


def _k3_both_formulations(lib, device, L, C, H, clip, seed, C1=0, nb=1):
    """The two formulations of the fused K3 (dlpd_zifft_filter_form: 1 = channel-owning waves with barrier-separated
    transform / filter phases, 2 = role-split transform / filter waves) on the SAME K2 output: the same fmaf chains in
    the same order, so V must agree bit for bit; form 2 (the default) is also compared with the oracle.
    C1 > 0: the reference's two-resolution layout (the coarse grid's pre-activation planes through both forms too)."""
    g = torch.Generator().manual_seed(seed)
    rec, lig = torch.randn(C, L, L, L, generator=g) * 0.05, torch.randn(C, L, L, L, generator=g) * 0.05
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    rec1 = lig1 = None
    if C1:
        rec1 = torch.randn(C1, L // 2, L // 2, L // 2, generator=g) * 0.1
        lig1 = torch.randn(C1, L // 2, L // 2, L // 2, generator=g) * 0.1
    W1, b1 = torch.randn(H, C + C1, generator=g) * 0.4, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    thr = 0.125 * L ** 3
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=clip, threshold_clash=thr, max_conf=10, batch=nb, device=device, lib=lib,
                        coarse_channels=C1)
    eng.set_receptor(rec, recf, rec1)
    eng.set_ligand(lig, ligf, lig1)
    R = orc.euler_to_matrix([0.5, -2.1][:nb], [1.1, 0.3][:nb], [-0.4, 1.7][:nb])
    out = {}
    for form in (1, 2):
        eng.k3_form = form
        out[form] = eng.score_batch(torch.from_numpy(R).float().to(device).contiguous()).cpu().clone()
        if C1:
            out[(form, "pre")] = eng.pre[:nb].cpu().clone()
    if C1:
        assert torch.equal(out[(1, "pre")], out[(2, "pre")])
    assert torch.equal(out[1], out[2])
    for i in range(nb):
        Rb = torch.from_numpy(R[i:i + 1]).float()
        rr, ll = [rec[None]], [orc.rotate_volume(lig[None], Rb)]
        if C1:
            rr.append(rec1[None])
            ll.append(orc.rotate_volume(lig1[None], Rb))
        S = orc.score_volumes(rr, ll, W1, b1, W2, b2, clip=clip)[0]
        mask, norm = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
        sure = (norm[0] - thr).abs() > 1e-3 * thr
        assert ((out[2][i] - mask[0] * S).abs()[sure]).max() <= 1e-4 * S.abs().max()
        assert float((mask[0] == 0).float().mean()) > 0.01


def test_k3_role_split_equals_the_channel_owning_k3_emulated(emu):
    """k_zifft_filter_rs on the fibre emulator at N = 128: 5 score channels + clash = 6 channels (two groups of the
    4 transform waves, the second partial), hidden width 20 (padded to 24), clip biting, one rotation."""
    assert emu.call("dlpd_hidden_pad", 20) == 24
    _k3_both_formulations(emu, "cpu", 64, 5, 20, 0.4, 77)


def test_k3_two_pencil_buffers_at_box_80_emulated(emu):
    """The N = 160 formulation (16-row tiles, five transform + ten filter waves, TWO pencil buffers read in place, raw staging
    of exactly one channel per wave with its last DMA instruction on eight lanes): 6 score channels + clash = 7 channels = two
    groups (4 + 3: the fifth transform wave idles, the clash channel closes the second group), hidden width 20 (padded to 24),
    clip biting -- bit for bit against the channel-owning K3 (a test variant) and within tolerance of the oracle."""
    _k3_both_formulations(emu, "cpu", 80, 6, 20, 0.4, 5)


def _fused_wide_hidden(lib, device, L, C, C1, H, nb, seed):
    """Hidden widths 33..48 on the fused pipeline (role-split K3, two voxels per thread): V against the oracle."""
    g = torch.Generator().manual_seed(seed)
    rec, lig = torch.randn(C, L, L, L, generator=g) * 0.05, torch.randn(C, L, L, L, generator=g) * 0.05
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    rec1 = lig1 = None
    if C1:
        rec1 = torch.randn(C1, L // 2, L // 2, L // 2, generator=g) * 0.1
        lig1 = torch.randn(C1, L // 2, L // 2, L // 2, generator=g) * 0.1
    W1, b1 = torch.randn(H, C + C1, generator=g) * 0.4, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    thr = 0.125 * L ** 3
    eng = DockingEngine(L, C, W1, b1, W2, b2, clip=0.6, threshold_clash=thr, max_conf=10, batch=nb, device=device, lib=lib,
                        coarse_channels=C1)
    assert eng.HP == 48
    eng.set_receptor(rec, recf, rec1)
    eng.set_ligand(lig, ligf, lig1)
    R = orc.euler_to_matrix([0.5, -2.1][:nb], [1.1, 0.3][:nb], [-0.4, 1.7][:nb])
    V = eng.score_batch(torch.from_numpy(R).float().to(device).contiguous()).cpu().clone()
    for i in range(nb):
        Rb = torch.from_numpy(R[i:i + 1]).float()
        rr, ll = [rec[None]], [orc.rotate_volume(lig[None], Rb)]
        if C1:
            rr.append(rec1[None])
            ll.append(orc.rotate_volume(lig1[None], Rb))
        S = orc.score_volumes(rr, ll, W1, b1, W2, b2, clip=0.6)[0]
        mask, norm = orc.clash_mask(recf[None, None], orc.rotate_volume(ligf[None, None], Rb), thr)
        sure = (norm[0] - thr).abs() > 1e-3 * thr
        assert ((V[i] - mask[0] * S).abs()[sure]).max() <= 1e-4 * S.abs().max()
        assert float((mask[0] == 0).float().mean()) > 0.01


def test_hidden_width_48_is_fused_emulated(emu):
    """Hidden width 40 (padded to 48) at N = 128: 8-row tiles, two voxels per filter thread, 3 + 1 channels (one group of
    the four two-channel transform waves)."""
    assert emu.call("dlpd_fused_hidden_pad", 40, 64, 0) == 48 and emu.call("dlpd_fused_hidden_pad", 40, 32, 0) == -1
    assert emu.call("dlpd_fused_hidden_pad", 49, 64, 0) == -1 and emu.call("dlpd_fused_hidden_pad", 24, 32, 0) == 24
    _fused_wide_hidden(emu, "cpu", 64, 3, 0, 40, 1, 91)


def _search_with_and_without_candidate_lists(lib, device, L, C, K, nrot, batch, monkeypatch, seed=3):
    """The top-K candidate path (K3 appends every score below the running K-th score, the select takes that list
    instead of a radix select over V) must give exactly the list of the full select path."""
    from oracle import docking_oracle as orc2
    g = torch.Generator().manual_seed(seed)
    rec, lig = torch.randn(C, L, L, L, generator=g) * 0.1, torch.randn(C, L, L, L, generator=g) * 0.1
    recf, ligf = torch.rand(L, L, L, generator=g), torch.rand(L, L, L, generator=g)
    H = max(C // 2, 1)
    W1, b1 = torch.randn(H, C, generator=g) * 0.3, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    ang = np.random.RandomState(seed).uniform(-np.pi, np.pi, size=(nrot, 3))
    R = orc2.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])
    lists, taus = [], []
    for off in (False, True):
        eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=0.3 * L ** 3, max_conf=K, batch=batch,
                            device=device, lib=lib, prefilter=not off)
        assert eng.prefilter == (not off)
        eng.set_receptor(rec, recf)
        eng.set_ligand(lig, ligf)
        eng.reset_top()
        eng.search(R)
        lists.append(eng.top_list())
        taus.append(int(eng.top.tau[0].item()))
    assert len(lists[0]) == K and lists[0][-1][4] < 0.0          # list full, K-th score negative: the filter was live
    assert 0 < taus[0] < 0x80000000
    assert lists[0] == lists[1]
    return lists[0]


def test_topk_candidate_lists_from_k3_equal_the_full_select_emulated(emu, monkeypatch):
    _search_with_and_without_candidate_lists(emu, "cpu", 32, 3, 25, 7, 2, monkeypatch)

def _packed_receptor_equals_natural(lib, device, L, CT, nb, seed=3):
    """K2 of the boxes that re-read the receptor spectrum for every rotation (80, 40), fed the spectrum in the natural layout
    and in the order its column phase consumes it (include/dlpd.h: dlpd_receptor_pack, dlpd_xy_correlate_packed): the same
    arithmetic on the same values, so the correlation spectra must agree bit for bit."""
    N, NZ = 2 * L, L + 1
    g = torch.Generator().manual_seed(seed)
    wsA = torch.randn(nb, CT, NZ, L, L, 2, generator=g).to(device)
    rec = torch.randn(CT, NZ, N, N, 2, generator=g).to(device)
    n = lib.call("dlpd_receptor_packed_floats", CT, L)
    assert n == rec.numel()
    packed = torch.zeros(n, dtype=torch.float32, device=device)
    B0 = torch.zeros(nb, CT, NZ, N, N, 2, device=device)
    B1 = torch.zeros_like(B0)
    st = 0 if device == "cpu" else torch.cuda.current_stream().cuda_stream
    lib.call("dlpd_xy_correlate", _ptr(wsA), _ptr(rec), _ptr(B0), nb, CT, L, 0, st)
    lib.call("dlpd_receptor_pack", _ptr(rec), _ptr(packed), CT, L, st)
    lib.call("dlpd_xy_correlate_packed", _ptr(wsA), _ptr(packed), _ptr(B1), nb, CT, L, st)
    if device != "cpu":
        torch.cuda.synchronize()
    assert sorted(packed.tolist()) == sorted(rec.flatten().tolist()) if rec.numel() < 2_000_000 else True
    assert float(B0.abs().max()) > 0
    assert torch.equal(B0, B1)


@pytest.mark.parametrize("L,CT,nb", [(40, 2, 2), (80, 1, 1)])
def test_packed_receptor_is_invisible_in_k2_emulated(emu, L, CT, nb):
    _packed_receptor_equals_natural(emu, "cpu", L, CT, nb)


def test_packed_receptor_exists_only_where_k2_rereads_it(emu):
    """Boxes 32 / 64 hold their receptor values in registers across the batch: no packed form, and asking for one fails."""
    for L in (32, 64):
        assert emu.call("dlpd_receptor_packed_floats", 4, L) == 0
        x = torch.zeros(16)
        with pytest.raises(RuntimeError):
            emu.call("dlpd_xy_correlate_packed", _ptr(x), _ptr(x), _ptr(x), 1, 1, L, 0)
    assert emu.call("dlpd_receptor_packed_floats", 3, 40) == 3 * 41 * 80 * 80 * 2
    assert emu.call("dlpd_receptor_packed_floats", 3, 80) == 3 * 81 * 160 * 160 * 2


def _blob_input(B, cin, D, seed, lo, hi):
    """(B, cin, D^3) input that is zero outside the box [lo, hi)^3 -- what a density splat of a protein looks like."""
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(B, cin, D, D, D)
    x[:, :, lo:hi, lo:hi, lo:hi] = torch.randn(B, cin, hi - lo, hi - lo, hi - lo, generator=g)
    return x, g


def _sparse_conv_checks(lib, device, cin, cmid, cout, ks, D, lo, hi, B=1):
    """Two chained bias-free convolutions with tile occupancy (empty tiles written as zeros, not computed) against the same
    convolutions without: the same bits; the occupancy map a layer hands on equals the map computed from its output; and
    tiles are really skipped (the map of the input has empty tiles)."""
    from deeplocalproteindocking_amd import ops
    x, g = _blob_input(B, cin, D, 7 + D, lo, hi)
    w1 = torch.randn(cmid, cin, ks, ks, ks, generator=g) * 0.1
    w2 = torch.randn(cout, cmid, 3, 3, 3, generator=g) * 0.1
    x, w1, w2 = x.to(device), w1.to(device), w2.to(device)
    occ0 = ops.tile_occupancy(x, lib=lib)
    assert occ0.shape == (B, (D + 3) // 4, (D + 3) // 4, (D + 3) // 4) and 0 < int(occ0.sum()) < occ0.numel()
    y1, occ1 = ops.conv3d(x, w1, relu=True, lib=lib, precision="split_bf16", occupancy=occ0, return_occupancy=True)
    y2, occ2 = ops.conv3d(y1, w2, relu=False, lib=lib, precision="split_bf16", occupancy=occ1, return_occupancy=True)
    d1 = ops.conv3d(x, w1, relu=True, lib=lib, precision="split_bf16")
    d2 = ops.conv3d(d1, w2, relu=False, lib=lib, precision="split_bf16")
    assert torch.equal(y1, d1) and torch.equal(y2, d2)
    assert torch.equal(occ1, ops.tile_occupancy(d1, lib=lib)) and torch.equal(occ2, ops.tile_occupancy(d2, lib=lib))
    # a stride-2 layer takes a map and hands none on (its output has another tiling)
    y3, occ3 = ops.conv3d(x, w1, lib=lib, stride=2, precision="split_bf16", occupancy=occ0, return_occupancy=True)
    assert occ3 is None and torch.equal(y3, ops.conv3d(x, w1, lib=lib, stride=2, precision="split_bf16"))
    # a map of the wrong size is refused
    with pytest.raises(RuntimeError, match="occupancy"):
        ops.conv3d(x, w1, lib=lib, precision="split_bf16", occupancy=occ0[:, :1])
    want = torch.relu(torch.nn.functional.conv3d(x.cpu().double(), w1.cpu().double(), padding=ks // 2))
    assert (y1.cpu().double() - want).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.parametrize("cin,cmid,cout,ks,D,lo,hi", [(11, 16, 16, 5, 13, 1, 4), (5, 32, 16, 3, 17, 9, 15), (8, 16, 48, 5, 12, 0, 3)])
def test_conv3d_tile_occupancy_skips_empty_tiles_with_the_same_bits(emu, cin, cmid, cout, ks, D, lo, hi):
    _sparse_conv_checks(emu, "cpu", cin, cmid, cout, ks, D, lo, hi)


def _sparse_pool_checks(lib, device, C, D, lo, hi, B=1):
    """MaxPool3d(5, 2, 2) on the tiled kernel: equal to torch's (a maximum of the same values), with an occupancy map of the
    input the same bits as without, and the map it hands on equals the map of its output."""
    from deeplocalproteindocking_amd import ops
    x, g = _blob_input(B, C, D, 11 + D, lo, hi)
    x[:, :, lo:hi, lo:hi, lo:hi] -= 0.3                             # (negative values too: the layer in front of the pooling has no ReLU)
    x = x.to(device)
    want = torch.nn.functional.max_pool3d(x.cpu(), kernel_size=5, stride=2, padding=2)
    y0 = ops.maxpool3d_5s2(x, lib=lib)
    assert torch.equal(y0.cpu(), want)
    occ = ops.tile_occupancy(x, lib=lib)
    y1, occ1 = ops.maxpool3d_5s2(x, lib=lib, occupancy=occ, return_occupancy=True)
    assert torch.equal(y1, y0) and 0 < int(occ.sum()) < occ.numel()
    assert torch.equal(occ1, ops.tile_occupancy(y0, lib=lib))
    y2, occ2 = ops.maxpool3d_5s2(x, lib=lib, return_occupancy=True)
    assert torch.equal(y2, y0) and torch.equal(occ2, occ1)


@pytest.mark.parametrize("C,D,lo,hi", [(3, 13, 1, 5), (2, 40, 22, 31), (11, 9, 1, 5)])        # 11 channels: a group of 8 and one of 3
def test_maxpool_tiled_with_occupancy_equals_torch(emu, C, D, lo, hi):
    _sparse_pool_checks(emu, "cpu", C, D, lo, hi)


class _NanEmpty(object):
    """``torch.empty`` hands out NaN-filled float buffers (and 0xFF bytes) inside the block: a kernel that reads a cell
    nobody wrote shows up as a NaN in its result."""

    def __enter__(self):
        self.orig = torch.empty

        def empty(*a, **kw):
            t = self.orig(*a, **kw)
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype == torch.uint8:
                t.fill_(255)
            return t
        torch.empty = empty
        return self

    def __exit__(self, *exc):
        torch.empty = self.orig
        return False


def _cells(occ, D):
    """uint8 (B, nc, nc, nc) cell map -> bool (B, 1, D, D, D) voxel mask"""
    m = occ.bool().repeat_interleave(4, 1).repeat_interleave(4, 2).repeat_interleave(4, 3)[:, :D, :D, :D]
    return m[:, None]


def _unwritten_chain_checks(lib, device, D=21, lo=9, hi=15, B=1):
    """conv k5 + ReLU -> conv k3 -> MaxPool(5, 2, 2) -> conv k3 with UNWRITTEN activations (empty tiles neither computed nor
    written, empty input cells never read; every output buffer starts as NaNs) against the same chain writing everything:
    the same bits in every cell the maps mark, the same maps, and the dense values outside the marked cells are zeros --
    so (tensor, map) stands for the same tensor.  Then the engine's K1 for given volumes (dlpd_zfft_volumes_occ) on the
    unwritten tensor + map: the same spectra as dlpd_zfft_into on the dense tensor."""
    from deeplocalproteindocking_amd import ops
    x, g = _blob_input(B, 11, D, 5, lo, hi)
    w1 = torch.randn(16, 11, 5, 5, 5, generator=g) * 0.1
    w2 = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.1
    w3 = torch.randn(32, 16, 3, 3, 3, generator=g) * 0.1
    x, w1, w2, w3 = x.to(device), w1.to(device), w2.to(device), w3.to(device)
    occ0 = ops.tile_occupancy(x, lib=lib)
    kw = dict(lib=lib, precision="split_bf16", return_occupancy=True)
    d1, m1 = ops.conv3d(x, w1, relu=True, occupancy=occ0, **kw)
    d2, m2 = ops.conv3d(d1, w2, occupancy=m1, **kw)
    d3, m3 = ops.maxpool3d_5s2(d2, lib=lib, occupancy=m2, return_occupancy=True)
    d4, m4 = ops.conv3d(d3, w3, occupancy=m3, **kw)
    with _NanEmpty():
        u1, n1 = ops.conv3d(x, w1, relu=True, occupancy=occ0, unwritten=True, **kw)
        u2, n2 = ops.conv3d(u1, w2, occupancy=n1, unwritten=True, **kw)
        u3, n3 = ops.maxpool3d_5s2(u2, lib=lib, occupancy=n2, return_occupancy=True, unwritten=True)
        u4, n4 = ops.conv3d(u3, w3, occupancy=n3, unwritten=True, **kw)
    Dp = (D - 1) // 2 + 1
    for d, m, u, n, Dx, skipped in ((d1, m1, u1, n1, D, True), (d2, m2, u2, n2, D, False), (d3, m3, u3, n3, Dp, False),
                                    (d4, m4, u4, n4, Dp, False)):
        assert torch.equal(m, n) and 0 < int(m.sum()) < m.numel() + (0 if skipped else 1)
        live = _cells(m, Dx).expand_as(d)
        assert torch.equal(d[live], u[live]) and not torch.isnan(u[live]).any()
        assert (d[~live] == 0).all()
        if skipped:                                                 # (behind the first layer every tile of this small box has an occupied neighbour)
            assert torch.isnan(u[~live]).any()                      # tiles were really left unwritten
    # the engine's K1 on given volumes (a box with a compiled plan): unwritten tensor + map == dense tensor
    L = 32
    x, g = _blob_input(B, 11, L, 6, 10, 19)
    x = x.to(device)
    occ0 = ops.tile_occupancy(x, lib=lib)
    d1, m1 = ops.conv3d(x, w1, relu=False, occupancy=occ0, **kw)
    with _NanEmpty():
        u1, n1 = ops.conv3d(x, w1, relu=False, occupancy=occ0, unwritten=True, **kw)
    call = (lib or __import__("deeplocalproteindocking_amd._lib", fromlist=["get_lib"]).get_lib()).call
    NZ, C = L + 1, 16
    A_d = torch.zeros(B, C + 1, NZ, L, L, 2, device=device)
    A_u = torch.full((B, C + 1, NZ, L, L, 2), float("nan"), device=device)
    st = torch.cuda.current_stream(device).cuda_stream if str(device) != "cpu" else 0
    call("dlpd_zfft_into", d1.data_ptr(), 0, A_d.data_ptr(), B, C, C + 1, 0, L, C * L ** 3, 0, 0.0, st)
    call("dlpd_zfft_volumes_occ", u1.data_ptr(), n1.data_ptr(), A_u.data_ptr(), B, C, C + 1, 0, L, C * L ** 3, 0, st)
    assert torch.equal(A_d[:, :C], A_u[:, :C]) and torch.isnan(A_u[:, C]).all()      # (the other channel is not touched)
    assert int(n1.sum()) < n1.numel() // 2                          # ... with most of the box empty
    # misuse is refused: unwritten needs the maps
    with pytest.raises(RuntimeError, match="unwritten"):
        ops.conv3d(x, w1, lib=lib, precision="split_bf16", unwritten=True)
    with pytest.raises(RuntimeError, match="unwritten"):
        ops.maxpool3d_5s2(x, lib=lib, unwritten=True)


def test_unwritten_activations_stand_for_the_same_tensors_emulated(emu):
    _unwritten_chain_checks(emu, "cpu")


def _k1_occupancy_checks(lib, device, L, C, nb, lo, hi, seed=17, scale=1.0):
    """The channels-last K1 going by per-rotation occupancy maps (dlpd_rotated_occupancy + dlpd_zfft_channels_last_occ, round 6:
    the rotation of Docker.py:218 for a ligand that is zero away from the protein) against the same kernel without maps:
    the same spectra bit for bit on oblique rotations; the maps are CONSERVATIVE -- every non-zero voxel of the really
    rotated volume (dlpd_rotate_trilinear) lies in a marked cell -- and they do leave cells out."""
    from deeplocalproteindocking_amd import ops
    if lib is None:
        from deeplocalproteindocking_amd._lib import get_lib
        lib = get_lib()
    g = torch.Generator().manual_seed(seed)
    NZ, CT, nc = L + 1, C + 1, (L + 3) // 4
    vol = torch.zeros(C, L, L, L)
    vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = torch.randn(C, hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], generator=g)
    vol = vol.to(device)
    ang = np.random.RandomState(seed).uniform(-np.pi, np.pi, size=(nb, 3))
    Rm = orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2]) * scale
    Rm[0] = np.eye(3) * scale                                        # (an axis-aligned one: cells map onto cells)
    R = torch.from_numpy(Rm).float().contiguous().to(device)
    st = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    c0 = L / 2.0
    cl = torch.empty(lib.call("dlpd_channels_last_floats", C, L), device=device)
    lib.call("dlpd_make_channels_last", _ptr(vol), _ptr(cl), C, L, st)
    occ_src = ops.tile_occupancy(vol.unsqueeze(0), lib=lib)
    assert occ_src.shape == (1, nc, nc, nc) and 0 < int(occ_src.sum()) < occ_src.numel() // 2
    occ = torch.full((nb, nc, nc, nc), 255, dtype=torch.uint8, device=device)
    lib.call("dlpd_rotated_occupancy", _ptr(occ_src), _ptr(R), _ptr(occ), 0, nb, L, c0, st)
    assert int((occ > 1).sum()) == 0 and 0 < int(occ.sum()) < occ.numel()
    rot = torch.empty(nb, C, L, L, L, device=device)
    lib.call("dlpd_rotate_trilinear", _ptr(vol), _ptr(R), _ptr(rot), nb, C, L, 0, c0, st)
    nz = (rot != 0).any(dim=1, keepdim=True)
    assert not bool((nz & ~_cells(occ, L)).any())                    # conservative
    assert bool((occ[0].bool() | ~occ_src[0].bool()).all()) or scale != 1.0  # identity: at least the ligand's own cells
    want = torch.full((nb * CT * NZ * L * L * 2,), 7.0, device=device)
    got = torch.full_like(want, float("nan"))
    lib.call("dlpd_zfft_channels_last_ext", _ptr(cl), _ptr(R), _ptr(want), nb, C, CT, 0, L, c0, 0, st)
    lib.call("dlpd_zfft_channels_last_occ", _ptr(cl), _ptr(R), _ptr(occ), _ptr(got), nb, C, CT, 0, L, c0, 0, 0, st)
    want, got = want.view(nb, CT, NZ, L, L, 2), got.view(nb, CT, NZ, L, L, 2)
    assert torch.equal(got[:, :C], want[:, :C]) and bool(torch.isnan(got[:, C]).all())
    return float(occ.float().mean())


@pytest.mark.parametrize("L,C,lo,hi", [(32, 20, (9, 12, 6), (20, 19, 15)), (40, 16, (22, 3, 14), (31, 12, 26))])
def test_k1_by_occupancy_maps_gives_the_same_spectra_emulated(emu, L, C, lo, hi):
    _k1_occupancy_checks(emu, "cpu", L, C, 3, lo, hi)


def test_engine_search_with_k1_occupancy_maps_gives_the_same_list_emulated(emu):
    """DockingEngine decides per ligand (cells occupied < SPARSE_K1_MAX_FILL): a blob-shaped two-resolution ligand is searched
    with the maps, a dense one without; switching the maps off gives the same list entry for entry."""
    from deeplocalproteindocking_amd.engine import DockingEngine
    g = torch.Generator().manual_seed(23)
    L, C, C1, H = 64, 8, 8, 4
    blob = lambda c, l, a, b: torch.nn.functional.pad(torch.randn(c, b - a, b - a, b - a, generator=g) * 0.3, (a, l - b) * 3)
    rec, lig = torch.randn(C, L, L, L, generator=g) * 0.05, blob(C, L, 22, 40)
    rec1, lig1 = torch.randn(C1, 32, 32, 32, generator=g) * 0.05, blob(C1, 32, 10, 21)
    recf, ligf = torch.rand(L, L, L, generator=g), blob(1, L, 24, 38)[0].abs()
    W1, b1 = torch.randn(H, C + C1, generator=g) * 0.3, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(1, H, generator=g), torch.randn(1, generator=g)
    R = torch.from_numpy(orc.euler_to_matrix([0.4, -1.3, 2.2], [0.9, 2.0, 0.3], [1.7, -0.2, -2.5])).float()
    lists = {}
    for mode in (None, False):
        eng = DockingEngine(L, C, W1, b1, W2, b2, clip=5.0, threshold_clash=50.0, max_conf=60, batch=2, device="cpu", lib=emu,
                            coarse_channels=C1, sparse_k1=mode)
        eng.set_receptor(rec, recf, rec1)
        eng.set_ligand(lig, ligf, lig1)
        sw = eng.switches()["k1_occupancy_maps"]
        assert sw["fine"] == sw["coarse"] == (mode is not False)
        assert mode is False or sw["ligand_cells_occupied"]["fine"] < 0.1
        eng.reset_top()
        eng.search(R)
        lists[mode] = eng.top_list()
    assert lists[None] == lists[False] and len(lists[None]) == 60
    eng.set_ligand(rec, ligf, rec1)                                  # a dense ligand on the last engine (maps off anyway) ...
    eng2 = DockingEngine(L, C, W1, b1, W2, b2, max_conf=4, batch=2, device="cpu", lib=emu, coarse_channels=C1)
    eng2.set_receptor(rec, recf, rec1)
    eng2.set_ligand(rec, ligf, rec1)                                 # ... and on an undecided one: no maps
    assert eng2.switches()["k1_occupancy_maps"]["fine"] is False and eng2.lig_fill == 1.0


def _pencil_map_checks(lib, device, L, C, nb, lo, hi, seed=29):
    """K1 that does not WRITE its empty blocks (dlpd_zfft_channels_last_occ, skip_empty) + K2 going by the per-rotation pencil
    map (dlpd_xy_correlate_packed_occ; the packed-receptor boxes 80 / 40) against K1 + K2 on everything: the same K2 output
    bit for bit, with the workspace between them full of NaNs wherever nothing was written; the clash channel behind the
    masked channels is dense and read as it is."""
    from deeplocalproteindocking_amd import ops
    if lib is None:
        from deeplocalproteindocking_amd._lib import get_lib
        lib = get_lib()
    assert lib.call("dlpd_pencil_map_supported", L) == 1 and lib.call("dlpd_pencil_map_supported", 64) == 0
    g = torch.Generator().manual_seed(seed)
    N, NZ, CT, nc = 2 * L, L + 1, C + 1, (L + 3) // 4
    lig = torch.zeros(C, L, L, L)
    lig[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = torch.randn(C, hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], generator=g)
    forb = torch.rand(nb, L, L, L, generator=g)                    # the clash channel: dense, per rotation (as from re-projected atoms)
    rec = torch.randn(CT, L, L, L, generator=g) * 0.1
    lig, forb, rec = lig.to(device), forb.to(device), rec.to(device)
    ang = np.random.RandomState(seed).uniform(-np.pi, np.pi, size=(nb, 3))
    R = torch.from_numpy(orc.euler_to_matrix(ang[:, 0], np.abs(ang[:, 1]), ang[:, 2])).float().contiguous().to(device)
    st = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    c0 = L / 2.0
    cl = torch.empty(lib.call("dlpd_channels_last_floats", C, L), device=device)
    lib.call("dlpd_make_channels_last", _ptr(lig), _ptr(cl), C, L, st)
    spec = torch.empty(CT * NZ * N * N * 2, device=device)
    scratch = torch.empty(CT * NZ * L * L * 2, device=device)
    lib.call("dlpd_rfft3d_padded", _ptr(rec), _ptr(spec), _ptr(scratch), CT, L, 1.0 / N ** 3, st)
    packed = torch.empty(lib.call("dlpd_receptor_packed_floats", CT, L), device=device)
    lib.call("dlpd_receptor_pack", _ptr(spec), _ptr(packed), CT, L, st)
    occ_src = ops.tile_occupancy(lig.unsqueeze(0), lib=lib)
    occ = torch.full((nb, nc, nc, nc), 255, dtype=torch.uint8, device=device)
    pen = torch.full((nb, nc), -1, dtype=torch.int32, device=device)
    lib.call("dlpd_rotated_occupancy", _ptr(occ_src), _ptr(R), _ptr(occ), _ptr(pen), nb, L, c0, st)
    want_bits = (occ.bool().any(dim=3).to(torch.int64) << torch.arange(nc, device=occ.device)).sum(dim=2).to(torch.int32)
    assert torch.equal(pen, want_bits) and 0 < int(occ.bool().any(dim=3).sum()) < nb * nc * nc
    pen2 = torch.full_like(pen, -1)
    lib.call("dlpd_pencil_bits", _ptr(occ), _ptr(pen2), nb, L, st)                 # (the same words from given maps)
    assert torch.equal(pen2, pen)
    wsA_d = torch.zeros(nb * CT * NZ * L * L * 2, device=device)
    wsA_s = torch.full_like(wsA_d, float("nan"))
    lib.call("dlpd_zfft_channels_last_ext", _ptr(cl), _ptr(R), _ptr(wsA_d), nb, C, CT, 0, L, c0, 0, st)
    lib.call("dlpd_zfft_channels_last_occ", _ptr(cl), _ptr(R), _ptr(occ), _ptr(wsA_s), nb, C, CT, 0, L, c0, 0, 1, st)
    for w in (wsA_d, wsA_s):
        lib.call("dlpd_zfft_into", _ptr(forb), 0, _ptr(w), nb, 1, CT, C, L, L ** 3, 0, 0.0, st)
    A_d, A_s = wsA_d.view(nb, CT, NZ, L, L, 2), wsA_s.view(nb, CT, NZ, L, L, 2)
    assert bool(torch.isnan(A_s[:, :C]).any()) and not bool(torch.isnan(A_s[:, C]).any())      # blocks really left unwritten
    written = ~torch.isnan(A_s)
    assert torch.equal(A_s[written], A_d[written]) and bool((A_d[~written] == 0).all())
    wsB_d = torch.empty(nb * CT * NZ * N * N * 2, device=device)
    wsB_s = torch.full_like(wsB_d, float("nan"))
    lib.call("dlpd_xy_correlate_packed", _ptr(wsA_d), _ptr(packed), _ptr(wsB_d), nb, CT, L, st)
    lib.call("dlpd_xy_correlate_packed_occ", _ptr(wsA_s), _ptr(packed), _ptr(wsB_s), nb, CT, L, _ptr(pen), C, st)
    assert torch.equal(wsB_d, wsB_s) and not bool(torch.isnan(wsB_s).any()) and float(wsB_d.abs().max()) > 0


@pytest.mark.parametrize("L,C,nb,lo,hi", [(40, 2, 2, (22, 3, 14), (31, 12, 26)), (80, 1, 1, (30, 41, 22), (44, 58, 35))])
def test_k2_by_the_pencil_map_reads_no_unwritten_pencil_emulated(emu, L, C, nb, lo, hi):
    _pencil_map_checks(emu, "cpu", L, C, nb, lo, hi)

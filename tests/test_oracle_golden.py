"""Pin the CPU oracle to the reference: every check compares oracle/docking_oracle.py with outputs
recorded from the reference's own code (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import docking_oracle as orc


def test_g1_correlation_definition_matches_multiply_volumes(golden):
    g = golden("g1_multiply_volumes.npz")
    for L in (4, 6):
        v1, v2 = g["v1_L%d" % L], g["v2_L%d" % L]
        direct = orc.correlate_direct(v1, v2)[0]                 # (C, N, N, N)
        N = 2 * L
        T = g["T_L%d" % L]
        got = direct[:, T[:, 0] % N, T[:, 1] % N, T[:, 2] % N].T   # (nT, C)
        np.testing.assert_allclose(got, g["out_L%d" % L], rtol=1e-5, atol=1e-5)
        # FFT form == direct form on the whole 2L grid (incl. the |t| = L zeros)
        fft = orc.correlate_fft(v1, v2, dtype=torch.float64).numpy()[0]
        np.testing.assert_allclose(fft, direct, atol=1e-10)
        assert np.all(direct[:, L, :, :] == 0) and np.all(direct[:, :, :, L] == 0)


def test_g1_fractional_translation_truncates_toward_zero(golden):
    g = golden("g1_multiply_volumes.npz")
    direct = orc.correlate_direct(g["v1_L6"], g["v2_L6"])[0]
    for b, t in enumerate(g["Tfrac"]):
        d = [int(x) for x in t]                                 # int(): -3.9 -> -3
        np.testing.assert_allclose(direct[:, d[0] % 12, d[1] % 12, d[2] % 12], g["out_frac"][b], rtol=1e-5, atol=1e-5)


def test_g2_euler_convention(golden):
    g = golden("g2_rotations.npz")
    for inc in (20, 15, 12, 10):
        for part in ("first8", "last8"):
            ang = g["ang_%s_%d" % (part, inc)]
            R = orc.euler_to_matrix(ang[:, 0], ang[:, 1], ang[:, 2])
            np.testing.assert_allclose(R, g["%s_%d" % (part, inc)], atol=1e-15)


@pytest.mark.parametrize("case", ["randn8_k5", "randn16_k40", "onehot_k4", "allpos_k4", "ties_k30", "fewneg_k6",
                                  "masked_k12"])
def test_g3_update_top(golden, case):
    g = golden("g3_update_top.npz")
    V = torch.from_numpy(g[case + "_V"].copy())
    K = int(g[case + "_K"])
    top = orc.update_top([], V, 7, K)
    ref = g[case + "_top"]
    assert len(top) == len(ref)
    for a, b in zip(top, ref):
        assert tuple(a[:4]) == tuple(int(x) for x in b[:4]) and a[4] == b[4]
    np.testing.assert_array_equal(V.numpy(), g[case + "_Vafter"])
    # vectorised equivalent (used to check the GPU at full size)
    idx, sc = orc.rotation_picks_fast(g[case + "_V"], K)
    N = V.shape[0]
    x, y, z = orc.flat_to_xyz(idx, N)
    picks = sorted(zip(sc.tolist(), range(K)), key=lambda t: t[0])       # stable sort by score
    got = [(7, int(x[i]), int(y[i]), int(z[i]), s) for s, i in picks]
    assert got == [(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(b[4])) for b in ref]


def test_g3_update_top_sequence_is_stable_across_rotations(golden):
    g = golden("g3_update_top.npz")
    top = []
    for r, V in enumerate(g["seq_V"]):
        top = orc.update_top(top, torch.from_numpy(V.copy()), r, int(g["seq_K"]))
    assert [tuple(t) for t in top] == [(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(b[4])) for b in g["seq_top"]]


def test_g4_dat_format(golden):
    g = golden("g4_write_conformations.npz")
    Rall = {int(i): g["R_used"][k] for k, i in enumerate(g["rot_ids"])}
    top = [(int(t[0]), int(t[1]), int(t[2]), int(t[3]), float(t[4])) for t in g["top_list"]]
    text = orc.format_conformations(top, Rall, int(g["box_size"]), float(g["resolution"]))
    assert text == bytes(g["text"]).decode()
    text = orc.format_conformations(top, Rall, int(g["box_size"]), float(g["resolution"]), randR=g["randR"])
    assert text == bytes(g["text_rand"]).decode()
    assert all(len(line.split("\t")) == 13 for line in text.strip().split("\n"))


@pytest.mark.parametrize("tag,nres", [("multires", 2), ("single", 1)])
def test_g5_global_forward_order(golden, tag, nres):
    g = golden("g5_global_forward.npz")
    rec = [torch.from_numpy(g["%s_rec%d" % (tag, i)]) for i in range(nres)]
    lig = [torch.from_numpy(g["%s_lig%d" % (tag, i)]) for i in range(nres)]
    V = orc.score_volumes(rec, lig, g[tag + "_W1"], g[tag + "_b1"], g[tag + "_W2"], g[tag + "_b2"],
                          clip=float(g[tag + "_clip"]))
    np.testing.assert_allclose(V.numpy(), g[tag + "_V"], rtol=1e-5, atol=1e-5)


def test_rotation_matches_grid_sample_and_geometry():
    """Build-defined trilinear rotation: independent check against torch grid_sample with an
    explicit grid (zeros padding) and against an analytic Gaussian blob rotated about the box
    centre L/2 (the geometric contract of Docker.py:221-223)."""
    torch.manual_seed(3)
    L = 12
    vol = torch.randn(2, 3, L, L, L)
    R = torch.from_numpy(orc.euler_to_matrix([0.4, -1.3], [0.9, 2.0], [1.7, -0.2])).float()
    out = orc.rotate_volume(vol, R)
    c0 = L / 2.0
    ar = torch.arange(L, dtype=torch.float32) - c0
    d = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), -1)            # (L,L,L,3) in (x,y,z)
    for b in range(2):
        p = d @ R[b] + c0                                                     # source index coords
        norm = 2.0 * p / (L - 1) - 1.0                                        # align_corners=True
        grid = norm[..., [2, 1, 0]][None]                                     # grid_sample wants (z,y,x)->(W,H,D)
        ref = torch.nn.functional.grid_sample(vol[b:b + 1], grid, mode="bilinear", padding_mode="zeros",
                                              align_corners=True)[0]
        assert (out[b] - ref).abs().max() < 1e-4
    # analytic blob: rotating the volume == evaluating the blob at rotated coordinates
    L = 24
    c0 = L / 2.0
    ar = torch.arange(L, dtype=torch.float64)
    X = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), -1)
    centre = torch.tensor([c0 + 4.0, c0 - 2.0, c0 + 1.5], dtype=torch.float64)
    blob = torch.exp(-((X - centre) ** 2).sum(-1) / (2 * 5.0 ** 2))
    Rm = torch.from_numpy(orc.euler_to_matrix(0.7, 1.2, -0.5))
    rotated = orc.rotate_volume(blob[None, None].float(), Rm[None].float())[0, 0]
    new_centre = Rm @ (centre - c0) + c0                                      # atom moved by R about the centre
    expect = torch.exp(-((X - new_centre) ** 2).sum(-1) / (2 * 5.0 ** 2))
    inner = ((X - c0) ** 2).sum(-1) < (L / 2.0 - 2.0) ** 2                    # sources stay inside the box
    assert (rotated.double() - expect)[inner].abs().max() < 2e-2              # trilinear smoothing only
    wrong = Rm.T @ (centre - c0) + c0                                         # negative control: inverse rotation
    wrong = torch.exp(-((X - wrong) ** 2).sum(-1) / (2 * 5.0 ** 2))
    assert (rotated.double() - wrong)[inner].abs().max() > 0.2


def test_dock_volumes_faithful_equals_fast():
    torch.manual_seed(5)
    L, C = 8, 4
    rec, lig = torch.randn(1, C, L, L, L) * 0.3, torch.randn(1, C, L, L, L) * 0.3
    rf, lf = torch.rand(1, 1, L, L, L), torch.rand(1, 1, L, L, L)
    W1, b1, W2, b2 = torch.randn(2, C), torch.randn(2), torch.randn(1, 2), torch.randn(1)
    R = orc.euler_to_matrix([0.1, 1.0, -2.0], [0.5, 1.5, 2.5], [0.0, -1.0, 2.0])
    a = orc.dock_volumes([rec], [lig], rf, lf, R, W1, b1, W2, b2, 40.0, 25, faithful_topk=True)
    b = orc.dock_volumes([rec], [lig], rf, lf, R, W1, b1, W2, b2, 40.0, 25, faithful_topk=False)
    assert a == b


def test_vectorised_projection_equals_the_loop_definition():
    """project_atoms_fast (used for protein-sized inputs) against the per-atom loop it restates."""
    rng = np.random.RandomState(4)
    counts = np.array([3, 0, 2, 1, 0, 4, 1, 0, 2, 3, 5])
    offs = np.cumsum(counts) - counts
    xyz = rng.uniform(-9.0, 9.0, size=(int(counts.sum()), 3))
    xyz[0] = [-14.9, 0.0, 14.9]                                     # splats clipped by the box faces
    R = orc.euler_to_matrix(0.4, 1.0, -0.7)
    for kw in (dict(), dict(R=R, shift=[15.0, 15.0, 15.0]), dict(R=R, shift=[15.0, 15.0, 15.0], sum_types=True)):
        a = orc.project_atoms(xyz.reshape(-1), counts, offs, 24, 1.25, **kw)
        b = orc.project_atoms_fast(xyz.reshape(-1), counts, offs, 24, 1.25, **kw)
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-12 and a.sum() > 1.0

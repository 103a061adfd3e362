"""Generate the golden fixtures in this directory by IMPORTING the reference's own code.

Run in the authoring container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py              # rewrite the fixtures
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --check      # regenerate elsewhere, compare array for array
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --out DIR    # write them into DIR

The reference's hot-path arithmetic lives in TorchProteinLibrary / se3cnn, which are not
installed, so those imports are replaced by MagicMock modules; every object whose OUTPUT is
recorded below is the reference's real code:

  G1  src/Models/MultiplyVolumes.py   MultiplyVolumes.forward     -> g1_multiply_volumes.npz
  G2  src/Utils/Rotations.py          Rotations(20).R             -> g2_rotations.npz
  G3  src/Docker/Docker.py            Docker.update_top           -> g3_update_top.npz
  G4  src/Docker/Docker.py            Docker.write_conformations  -> g4_write_conformations.npz
  G5  src/Models/DockingModels.py     GlobalDockingModel.forward, SimpleFilter
                                      (with .convolve := oracle FFT correlation)
                                                                  -> g5_global_forward.npz
  G6  scripts/Results/Benchmark/DockerParser.py   DockerParser.parse_output (on G4's text)
      src/Dataset/SplitComplexBenchmark.py        read_pdb_list, read_dataset_list (synthetic tables)
                                                                  -> g6_consumers.npz
  G7  src/Models/ProteinRepresentationModels.py   E3MultiResRepr4x4 (plain torch; se3cnn mocked): seeded state_dict,
                                      a seeded (1, 11, 12^3) input and both outputs, multiplier 1 and 8
      src/local_train.py                          select_model, E3 branch (``.cuda()`` made a no-op: no GPU here)
                                                                  -> g7_e3_plugin.npz
Only data (inputs + expected outputs) is written; no reference source is copied.
"""
import hashlib
import io
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
import numpy as np
import torch

FIXTURES = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(FIXTURES, "..", ".."))
HERE = FIXTURES            # where the .npz files are written (--out DIR / --check: a directory of their own)
REF = "/root/reference"
sys.path.insert(0, REPO)


def install_stubs():
    for name in ["TorchProteinLibrary", "TorchProteinLibrary.FullAtomModel",
                 "TorchProteinLibrary.FullAtomModel.CoordsTransform",
                 "TorchProteinLibrary.Volume", "_Volume", "se3cnn", "se3cnn.non_linearities",
                 "se3cnn.blocks", "se3cnn.filter"]:
        sys.modules[name] = MagicMock()
    src = types.ModuleType("src")
    src.__path__ = [os.path.join(REF, "src")]
    src.REPOSITORY_DIR = REF
    sys.modules["src"] = src


def check(fresh_dir):
    """Every array of every committed fixture equals the freshly generated one (dtype, shape and values)."""
    bad = 0
    for f in sorted(os.listdir(FIXTURES)):
        if not f.endswith(".npz"):
            continue
        a, b = np.load(os.path.join(FIXTURES, f), allow_pickle=False), np.load(os.path.join(fresh_dir, f), allow_pickle=False)
        same = sorted(a.files) == sorted(b.files) and all(
            a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f")
            for k in a.files)
        print("%-28s %3d arrays  %s" % (f, len(a.files), "identical" if same else "DIFFERENT"))
        bad += 0 if same else 1
    return bad


def main():
    global HERE
    if len(sys.argv) > 1 and sys.argv[1] in ("--out", "--check"):
        # --out DIR: write the fixtures there instead of over the committed ones; --check: regenerate into a temporary
        # directory and compare array for array with the committed ones (exit code = number of differing files)
        import tempfile
        HERE = sys.argv[2] if sys.argv[1] == "--out" else tempfile.mkdtemp(prefix="dlpd_golden_")
        os.makedirs(HERE, exist_ok=True)
    install_stubs()
    import importlib
    from oracle import docking_oracle as orc

    # ---------------- G1: MultiplyVolumes ----------------
    MV = importlib.import_module("src.Models.MultiplyVolumes").MultiplyVolumes()
    g = torch.Generator().manual_seed(101)
    g1 = {}
    for L in (4, 6):
        C = 3
        v1 = torch.randn(1, C, L, L, L, generator=g)
        v2 = torch.randn(1, C, L, L, L, generator=g)
        ts = [(dx, dy, dz) for dx in range(-(L - 1), L) for dy in range(-(L - 1), L)
              for dz in range(-(L - 1), L)]
        out = np.zeros((len(ts), C), dtype=np.float32)
        for i, t in enumerate(ts):
            T = torch.tensor([t], dtype=torch.float32)
            out[i] = MV(v1, v2, T).numpy()[0]
        g1["v1_L%d" % L] = v1.numpy(); g1["v2_L%d" % L] = v2.numpy()
        g1["T_L%d" % L] = np.array(ts, dtype=np.int32); g1["out_L%d" % L] = out
    # fractional translations: int() truncation toward zero (MultiplyVolumes.py:56-58)
    L = 6
    v1 = torch.from_numpy(g1["v1_L6"]).repeat(2, 1, 1, 1, 1)
    v2 = torch.from_numpy(g1["v2_L6"]).repeat(2, 1, 1, 1, 1)
    Tf = torch.tensor([[1.7, -2.2, 0.4], [-3.9, 2.2, 5.0]])
    g1["Tfrac"] = Tf.numpy(); g1["out_frac"] = MV(v1, v2, Tf).numpy()
    np.savez_compressed(os.path.join(HERE, "g1_multiply_volumes.npz"), **g1)

    # ---------------- G2: Rotations ----------------
    Rmod = importlib.import_module("src.Utils.Rotations")
    g2 = {}
    for inc in (20, 15, 12, 10):
        R = Rmod.Rotations(inc).R.numpy()
        g2["n_%d" % inc] = np.int64(R.shape[0])
        g2["first8_%d" % inc] = R[:8].copy()
        g2["last8_%d" % inc] = R[-8:].copy()
        ang = np.loadtxt(os.path.join(REF, "data", "oim%02d.eul" % inc)).reshape(-1, 3)
        g2["ang_first8_%d" % inc] = ang[:8].copy()      # 16 input rows so the Euler-convention
        g2["ang_last8_%d" % inc] = ang[-8:].copy()      # check is self-contained
        g2["sum_%d" % inc] = R.sum(axis=0)
        g2["abs_sum_%d" % inc] = np.abs(R).sum(axis=0)
        # order-sensitive checksum: sum_i (i+1) * R_i
        w = np.arange(1, R.shape[0] + 1, dtype=np.float64)[:, None, None]
        g2["wsum_%d" % inc] = (w * R).sum(axis=0)
    np.savez_compressed(os.path.join(HERE, "g2_rotations.npz"), **g2)

    # ---------------- G3 / G4: Docker.update_top, write_conformations ----------------
    Dmod = importlib.import_module("src.Docker.Docker")
    RefDocker = Dmod.Docker
    dk = RefDocker(docking_model=None, angle_inc=20, box_size=4, resolution=1.25, max_conf=5,
                   randomize_rot=False)
    g3 = {}
    cases = {}
    g = torch.Generator().manual_seed(202)
    cases["randn8_k5"] = (torch.randn(8, 8, 8, generator=g), 5)
    cases["randn16_k40"] = (torch.randn(16, 16, 16, generator=g), 40)
    z = torch.zeros(8, 8, 8); z[1, 2, 3] = -1.0
    cases["onehot_k4"] = (z, 4)
    pos = torch.rand(6, 6, 6, generator=g) + 0.5
    cases["allpos_k4"] = (pos, 4)
    ties = torch.randint(-3, 2, (8, 8, 8), generator=g).float()
    cases["ties_k30"] = (ties, 30)
    fewneg = torch.rand(8, 8, 8, generator=g) + 0.1
    fewneg[7, 0, 1] = -2.0; fewneg[0, 5, 5] = -2.0; fewneg[3, 3, 3] = -0.5
    cases["fewneg_k6"] = (fewneg, 6)
    masked = torch.randn(8, 8, 8, generator=g) * (torch.rand(8, 8, 8, generator=g) < 0.02).float()
    cases["masked_k12"] = (masked, 12)            # mostly +-0.0 entries, few negatives
    for name, (V, K) in cases.items():
        dk.max_conf = K
        dk.top_list = []
        Vin = V.clone()
        dk.update_top(V, 7)
        g3[name + "_V"] = Vin.numpy()
        g3[name + "_K"] = np.int64(K)
        g3[name + "_top"] = np.array(dk.top_list, dtype=np.float64)
        g3[name + "_Vafter"] = V.numpy()
    # multi-rotation accumulation (stable order across rotations)
    dk.max_conf = 10
    dk.top_list = []
    seq = []
    for r in range(4):
        V = torch.randint(-4, 3, (6, 6, 6), generator=g).float()
        seq.append(V.clone().numpy())
        dk.update_top(V, r)
    g3["seq_V"] = np.stack(seq); g3["seq_K"] = np.int64(10)
    g3["seq_top"] = np.array(dk.top_list, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g3_update_top.npz"), **g3)

    # G4
    dk.max_conf = 6
    dk.box_size = 4
    dk.top_list = [(0, 0, 0, 0, -3.5), (5, 7, 3, 4, -1.25), (1853, 3, 4, 7, -0.000001),
                   (17, 1, 2, 5, 0.0), (100, 4, 4, 4, 2.5)]
    buf = io.StringIO()
    dk.log = buf
    dk.write_conformations()
    text = buf.getvalue()
    dk.log = None
    g4 = {"top_list": np.array(dk.top_list, dtype=np.float64), "box_size": np.int64(4),
          "resolution": np.float64(1.25),
          "R_used": dk.rot.R.numpy()[[0, 5, 1853, 17, 100]],
          "rot_ids": np.array([0, 5, 1853, 17, 100]),
          "text": np.frombuffer(text.encode(), dtype=np.uint8)}
    # with randomize_rot
    dk.randomize_rot = True
    randR = orc.euler_to_matrix(0.3, 1.1, -2.0)
    dk.randR = torch.from_numpy(randR).unsqueeze(0)
    buf = io.StringIO(); dk.log = buf
    dk.write_conformations()
    g4["randR"] = randR
    g4["text_rand"] = np.frombuffer(buf.getvalue().encode(), dtype=np.uint8)
    dk.log = None
    np.savez_compressed(os.path.join(HERE, "g4_write_conformations.npz"), **g4)

    # ---------------- G5: GlobalDockingModel.forward ----------------
    import warnings
    warnings.filterwarnings("ignore")
    DM = importlib.import_module("src.Models.DockingModels")
    g5 = {}
    for tag, sizes, L in (("multires", [16, 32], 8), ("single", [4], 6)):
        torch.manual_seed(303)
        filt = DM.SimpleFilter(sizes)
        model = DM.GlobalDockingModel(representation=None, filter=filt, threshold_clash=300)
        clip = 5.0
        model.convolve = lambda a, b: orc.correlate_fft(a, b, clip=clip)
        g = torch.Generator().manual_seed(304)
        rec, lig = [], []
        for i, c in enumerate(sizes):
            Li = L // (2 ** i)
            rec.append(torch.randn(2, c, Li, Li, Li, generator=g) * 0.3)
            lig.append(torch.randn(2, c, Li, Li, Li, generator=g) * 0.3)
        with torch.no_grad():
            V = model(rec, lig)
        for i in range(len(sizes)):
            g5["%s_rec%d" % (tag, i)] = rec[i].numpy()
            g5["%s_lig%d" % (tag, i)] = lig[i].numpy()
        g5[tag + "_W1"] = filt.fc[0].weight.detach().numpy()
        g5[tag + "_b1"] = filt.fc[0].bias.detach().numpy()
        g5[tag + "_W2"] = filt.fc[2].weight.detach().numpy()
        g5[tag + "_b2"] = filt.fc[2].bias.detach().numpy()
        g5[tag + "_clip"] = np.float64(clip)
        g5[tag + "_V"] = V.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_global_forward.npz"), **g5)

    # ---------------- G6: .dat consumer + benchmark table loader ----------------
    import tempfile, json
    import importlib.util
    for name in ["TorchProteinLibrary.RMSD", "Dataset", "Dataset.processing_utils", "DockingBenchmark",
                 "VisualizeBenchmark", "matplotlib", "matplotlib.pylab"]:
        sys.modules[name] = MagicMock()
    sys.modules["src"].LOG_DIR = sys.modules["src"].DATA_DIR = "/nonexistent"
    bdir = os.path.join(REF, "scripts", "Results", "Benchmark")
    sys.path.insert(0, bdir)
    spec = importlib.util.spec_from_file_location("ref_DockerParser", os.path.join(bdir, "DockerParser.py"))
    DP = importlib.util.module_from_spec(spec); spec.loader.exec_module(DP)
    g6 = {}
    with tempfile.TemporaryDirectory() as td:
        dat = bytes(g4["text"]).decode() + "0.5 0 0 0 0.5 0 0 0 1\t-3.75\t2.5\t-0.99\t-1.5e-3\n"
        open(os.path.join(td, "T1.dat"), "w").write(dat)
        res = DP.DockerParser(td).parse_output("T1", header_only=False)
        g6["dat_text"] = np.frombuffer(dat.encode(), dtype=np.uint8)
        g6["parsed_R"] = np.stack([c[0].numpy()[0] for c in res["conformations"]])
        g6["parsed_t"] = np.stack([c[1].numpy()[0] for c in res["conformations"]])
        g6["parsed_score"] = np.array([c[2] for c in res["conformations"]])
        assert DP.DockerParser(td).parse_output("missing") is None
        # benchmark tables
        del sys.modules["Dataset"], sys.modules["Dataset.processing_utils"]
        SB = importlib.import_module("src.Dataset.SplitComplexBenchmark")
        table = ("Complex\tCat.\tPDB ID 1\n1ABC_A:B\tE\tx\n"          # before any section: ignored
                 "Rigid-body (151)\t\t\n1AHW_AB:C\tA\t1FGN_LH\n1BVK_DE:F\tA\t1BVL_BA\n"
                 "Medium Difficulty (45)\t\t\n1BGX_HL:T\tA\t1AY1_HL\n"
                 "Difficult (34)\t\t\n1E4K_AB:C\tO\t2DTQ_AB\n2HMI_AB:CD\tO\t1S6P_AB\n")
        open(os.path.join(td, "Table.csv"), "w").write(table)
        t1 = SB.read_pdb_list(td, os.path.join(td, "Table.csv"), struct_folder="structs")
        os.mkdir(os.path.join(td, "Description"))
        open(os.path.join(td, "Description", "set.dat"), "w").write("1AAA extra\n2BBB\n")
        open(os.path.join(td, "Description", "1AAA.dat"), "w").write("Receptor Ligand\n/p/r1.pdb /p/l1.pdb 3\n")
        open(os.path.join(td, "Description", "2BBB.dat"), "w").write("hdr\nr2.pdb l2.pdb\nmore\n")
        t2 = SB.read_dataset_list(td, os.path.join(td, "Description", "set.dat"))
        rel = lambda rows: [[(v.replace(td, "<D>") if isinstance(v, str) else v) for v in r] for r in rows]
        g6["table_text"] = np.frombuffer(table.encode(), dtype=np.uint8)
        g6["pdb_list_json"] = np.frombuffer(json.dumps(rel(t1)).encode(), dtype=np.uint8)
        g6["dataset_list_json"] = np.frombuffer(json.dumps(rel(t2)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g6_consumers.npz"), **g6)

    # ---------------- G7: the E3 representation plugin + select_model ----------------
    PR = importlib.import_module("src.Models.ProteinRepresentationModels")
    g7 = {}
    for m in (1, 8):
        torch.manual_seed(700 + m)
        net = PR.E3MultiResRepr4x4(multiplier=m).eval()
        g = torch.Generator().manual_seed(710 + m)
        x = torch.relu(torch.randn(1, 11, 12, 12, 12, generator=g)) * 0.5          # density-like: non-negative
        with torch.no_grad():
            v1, v2 = net(x)
        sd = net.state_dict()
        g7["m%d_keys" % m] = np.frombuffer(json.dumps(list(sd.keys())).encode(), dtype=np.uint8)
        for k, v in sd.items():
            g7["m%d_sd_%s" % (m, k)] = v.numpy()
        g7["m%d_input" % m] = x.numpy()
        g7["m%d_out0" % m], g7["m%d_out1" % m] = v1.numpy(), v2.numpy()
        g7["m%d_num_outputs" % m] = np.array(net.get_num_outputs(), dtype=np.int64)
    # select_model (local_train.py:19-43): which classes, which multiplier, how the filter is sized; error behaviour
    for name in ("Training", "Dataset"):
        sys.modules[name] = MagicMock()
    sys.modules["Models"] = importlib.import_module("src.Models")
    sys.modules["src"].MODELS_DIR = "/nonexistent"
    LT = importlib.import_module("src.local_train")
    real_cuda = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, device=None: self
    try:
        args = types.SimpleNamespace(group="E3", model="E3MultiResRepr4x4", filter="SimpleFilter")
        pm, cf = LT.select_model(args)
        sel = {"repr_class": type(pm).__name__, "filter_class": type(cf).__name__, "num_outputs": pm.get_num_outputs(),
               "repr_shapes": {k: list(v.shape) for k, v in pm.state_dict().items()},
               "filter_shapes": {k: list(v.shape) for k, v in cf.state_dict().items()}, "errors": {}}
        for tag, bad in (("group", dict(group="XX", model="E3MultiResRepr4x4", filter="SimpleFilter")),
                         ("model", dict(group="E3", model="Nope", filter="SimpleFilter")),
                         ("filter", dict(group="E3", model="E3MultiResRepr4x4", filter="Nope"))):
            try:
                LT.select_model(types.SimpleNamespace(**bad))
                sel["errors"][tag] = None
            except Exception as exc:
                sel["errors"][tag] = [type(exc).__name__] + [str(a) for a in exc.args]
    finally:
        torch.nn.Module.cuda = real_cuda
    g7["select_model_json"] = np.frombuffer(json.dumps(sel, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g7_e3_plugin.npz"), **g7)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            h = hashlib.sha256(open(os.path.join(HERE, f), "rb").read()).hexdigest()[:16]
            print(f, os.path.getsize(os.path.join(HERE, f)), h)
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(check(HERE))


if __name__ == "__main__":
    main()

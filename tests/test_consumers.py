"""benchmark table loader and the .dat consumer, against fixtures captured
from the reference's own functions (tests/golden/make_golden.py, G6)."""
import json
import os

import numpy as np
import torch

from deeplocalproteindocking_amd.Dataset import get_benchmark_stream, read_dataset_list, read_pdb_list
from deeplocalproteindocking_amd.Results import DockerParser, kabsch_rmsd
from oracle import docking_oracle as orc


def _text(a):
    return bytes(np.asarray(a, dtype=np.uint8)).decode()


def test_parse_output_matches_reference(golden, tmp_path):
    g6 = golden("g6_consumers.npz")
    (tmp_path / "T1.dat").write_text(_text(g6["dat_text"]))
    res = DockerParser(str(tmp_path)).parse_output("T1", header_only=False)
    R = np.stack([c[0].numpy()[0] for c in res["conformations"]])
    t = np.stack([c[1].numpy()[0] for c in res["conformations"]])
    sc = np.array([c[2] for c in res["conformations"]])
    assert np.array_equal(R, g6["parsed_R"]) and np.array_equal(t, g6["parsed_t"]) and np.array_equal(sc, g6["parsed_score"])
    assert res["conformations"][0][0].shape == (1, 3, 3) and res["conformations"][0][1].dtype == torch.double
    assert t[-1].tolist() == [-3.0, 2.0, 0.0]                # int(float()) truncation toward zero
    assert DockerParser(str(tmp_path)).parse_output("missing") is None


def test_benchmark_tables_match_reference(golden, tmp_path):
    g6 = golden("g6_consumers.npz")
    (tmp_path / "Table.csv").write_text(_text(g6["table_text"]))
    D = str(tmp_path)
    rel = lambda rows: [[(v.replace(D, "<D>") if isinstance(v, str) else v) for v in r] for r in rows]
    got = read_pdb_list(D, os.path.join(D, "Table.csv"), struct_folder="structs")
    assert rel(got) == json.loads(_text(g6["pdb_list_json"]))
    assert [r[6] for r in got] == [1, 1, 2, 3, 3]
    os.mkdir(os.path.join(D, "Description"))
    open(os.path.join(D, "Description", "set.dat"), "w").write("1AAA extra\n2BBB\n")
    open(os.path.join(D, "Description", "1AAA.dat"), "w").write("Receptor Ligand\n/p/r1.pdb /p/l1.pdb 3\n")
    open(os.path.join(D, "Description", "2BBB.dat"), "w").write("hdr\nr2.pdb l2.pdb\nmore\n")
    assert rel(read_dataset_list(D, os.path.join(D, "Description", "set.dat"))) == json.loads(_text(g6["dataset_list_json"]))
    # the stream local_test.py:57-63 consumes: batch-1 collated 7-tuples
    item = next(iter(get_benchmark_stream(D, struct_folder="structs", subset="Table.csv")))
    assert item[0][0] == "1AHW" and item[2][0].endswith("structs/1AHW_r_u.pdb") and int(item[6][0]) == 1


def test_kabsch_rmsd_and_pose_reconstruction(tmp_path):
    g = torch.Generator().manual_seed(5)
    X = torch.randn(40, 3, generator=g, dtype=torch.double) * 8
    R = torch.from_numpy(orc.euler_to_matrix(0.4, 1.0, -2.2))
    assert float(kabsch_rmsd(X @ R.t() + torch.tensor([3.0, -1.0, 7.0]), X)) < 1e-9       # rigid motion: 0
    noise = torch.randn(40, 3, generator=g, dtype=torch.double) * 0.3
    r = float(kabsch_rmsd(X @ R.t() + noise, X))
    assert 0.2 < r <= float(torch.sqrt((noise ** 2).sum(dim=1).mean())) + 1e-12          # never above the unfitted RMSD
    assert float(kabsch_rmsd(X * torch.tensor([1.0, 1.0, -1.0]), X)) > 1.0                # reflections are not allowed
    # pose line -> transform_ligand / interface_rmsd: ligand placed exactly by the written pose
    Rp = orc.euler_to_matrix(1.1, 0.5, 0.3)
    line = "\t".join("%f" % v for v in Rp.reshape(-1)) + "\t5.000000\t-10.000000\t2.500000\t-1.0\n"
    (tmp_path / "X.dat").write_text(line)
    dp = DockerParser(str(tmp_path))
    dp.parse_output("X")
    lig = X[:15]
    coords = torch.zeros(1, 45, dtype=torch.double); coords[0] = lig.reshape(-1)
    out = dp.transform_ligand((coords, None, None, None, None, torch.tensor([15], dtype=torch.int32)), 0)
    Rw = torch.tensor([float("%f" % v) for v in Rp.reshape(-1)], dtype=torch.double).reshape(3, 3)
    want = lig @ Rw.t() + torch.tensor([5.0, -10.0, 2.0], dtype=torch.double)
    assert torch.allclose(out[0][0].reshape(15, 3), want, atol=1e-12)
    rec = X[15:]
    native = torch.cat([rec, want]) @ R.t() + 4.0                # the same complex in another frame
    assert dp.interface_rmsd([(rec, lig)], [native], 0) < 1e-9

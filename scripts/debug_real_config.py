"""Diagnostic: the reference's real configuration (SE3 repr, box 80) -- GPU V per rotation vs the oracle."""
import os as _os; _os.environ.setdefault("DLPD_ALLOW_GENERATED_ROTATIONS", "1")   # diagnostic script: SOI-sized generated set when the licensed files are absent
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from oracle import docking_oracle as orc
from test_atoms import _typed, write_fake_pdb
from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
from deeplocalproteindocking_amd.Docker import Docker
from deeplocalproteindocking_amd.Models import GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
import pathlib
tmp = pathlib.Path(tempfile.mkdtemp())
dev = torch.device("cuda:0")
L, res, K = 80, 1.25, 40
frec, _, _, _ = _typed(tmp, 40, seed=15)
flig, _, _, _ = _typed(tmp, 25, seed=16)
torch.manual_seed(80)
repr_ = SE3MultiResReprScalar(multiplier=8)
model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=3.0)
R = orc.euler_to_matrix([0.3, -1.0], [1.1, 0.4], [-2.0, 2.5])
be = CoordsBackend()
centre = torch.full((1, 3), L * res / 2.0, dtype=torch.double)
def load(f):
    c, ch, rn, ri, an, nat = be.pdb2coords([f])
    t, cnt, off = be.assign_types(c, rn, an, nat)
    a, b = be.get_bbox(t, be.last_num_typed)
    return be.translate(t, -(a + b) * 0.5, be.last_num_typed), cnt, off
rc, rn_, ro = load(frec); lc, ln_, lo = load(flig)
rec = torch.from_numpy(orc.project_atoms(rc[0].numpy(), rn_[0].numpy(), ro[0].numpy(), L, res, shift=centre[0].numpy())).float()[None]
lig = torch.from_numpy(orc.project_atoms(lc[0].numpy(), ln_[0].numpy(), lo[0].numpy(), L, res, shift=centre[0].numpy())).float()[None]
with torch.no_grad():
    rv, lv = model.representation(rec), model.representation(lig)
W = [w.cpu() for w in model.filter.parameters_tuple()]
dk = Docker(model.to(dev), box_size=L, resolution=res, max_conf=K, rotations=R, device=dev, coords_backend=be)
with torch.no_grad():
    dk.dockSE3(frec, flig, batch_size=2)
    eng = dk.engine
    grv = [v.cpu() for v in model.representation(rec.to(dev))]
    print("repr GPU vs CPU:", [float((a - b).abs().max() / b.abs().max()) for a, b in zip(grv, rv)])
    Rd = torch.from_numpy(R).float().to(dev).contiguous()
    V = eng.score_batch(Rd).cpu()
    forb_gpu = eng.clash_provider(Rd).cpu()
for ri in range(2):
    Rb = torch.from_numpy(R[ri:ri + 1]).float()
    lrot = [orc.rotate_volume(v, Rb) for v in lv]
    lforb = torch.from_numpy(orc.project_atoms(lc[0].numpy(), ln_[0].numpy(), lo[0].numpy(), L, res, R=R[ri], shift=centre[0].numpy(), sum_types=True)).float()[None]
    print(ri, "forb err", float((forb_gpu[ri] - lforb[0]).abs().max()), float(lforb.abs().max()))
    mask, norm = orc.clash_mask(rec.sum(dim=1, keepdim=True), lforb, 3.0)
    S = orc.score_volumes(rv, lrot, *W, clip=5.0)
    Vo = (mask * S)[0]
    same = (V[ri] == 0) == (Vo == 0)
    print(ri, "mask flips", int((~same).sum()), "masked frac", float((Vo == 0).float().mean()),
          "V err (agreeing voxels)", float((V[ri] - Vo).abs()[same].max()), "scale", float(Vo.abs().max()),
          "unmasked score err", float(((V[ri] - S[0]).abs() * (V[ri] != 0)).max()))
    near = ((norm[0] - 3.0).abs() < 1e-3 * 3.0).sum()
    print(ri, "voxels within 1e-3 of threshold:", int(near), " min V gpu/oracle:", float(V[ri].min()), float(Vo.min()))

from deeplocalproteindocking_amd.engine import DockingEngine
for scale_in in (1.0, 0.05):
    rv2, lv2 = [v * scale_in for v in rv], [v * scale_in for v in lv]
    eng2 = DockingEngine(L, 16, *W, clip=5.0, threshold_clash=3.0, has_clash=True, max_conf=K, batch=2, device=dev,
                         coarse_channels=32)
    eng2.set_receptor(rv2[0][0], rec.sum(dim=1)[0], rv2[1][0])
    eng2.set_ligand(lv2[0][0], lig.sum(dim=1)[0], lv2[1][0])
    V2 = eng2.score_batch(Rd).cpu()
    for ri in range(2):
        Rb = torch.from_numpy(R[ri:ri + 1]).float()
        lrot = [orc.rotate_volume(v, Rb) for v in lv2]
        S = orc.score_volumes(rv2, lrot, *W, clip=5.0)
        Snc = orc.score_volumes(rv2, lrot, *W, clip=None)
        nz = V2[ri] != 0
        print("scale_in", scale_in, "rot", ri, "no-provider: score err", float(((V2[ri] - S[0]).abs() * nz).max()),
              "scale", float(S.abs().max()), " (err vs unclipped oracle", float(((V2[ri] - Snc[0]).abs() * nz).max()), ")")
    c0 = orc.correlate_fft(rv2[0], orc.rotate_volume(lv2[0], torch.from_numpy(R[:1]).float()), clip=None)
    c1 = orc.correlate_fft(rv2[1], orc.rotate_volume(lv2[1], torch.from_numpy(R[:1]).float()), clip=None)
    print("   max |corr| fine / coarse:", float(c0.abs().max()), float(c1.abs().max()))

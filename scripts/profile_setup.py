"""Where the per-target set-up time of dockSE3 goes (host side): cProfile of the SECOND target of a two-target replay
(the first one pays imports, the library load and the engine's workspaces)."""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DLPD_ALLOW_GENERATED_ROTATIONS", "1")
import torch
import __graft_entry__ as entry
entry.build()
from test_replay_local_test import make_benchmark
from deeplocalproteindocking_amd.Docker import Docker
from deeplocalproteindocking_amd.Models import GlobalDockingModel, SE3MultiResReprScalar, SimpleFilter
root = tempfile.mkdtemp(prefix="dlpd_setup_")
bench = make_benchmark(root, targets=(("1SYN", 230, 120, 21), ("2SYN", 310, 95, 33)))
torch.manual_seed(3)
repr_ = SE3MultiResReprScalar(multiplier=8)
model = GlobalDockingModel(repr_, SimpleFilter(repr_.get_num_outputs()), threshold_clash=40.0).cuda()
angle = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dk = Docker(model, angle_inc=angle, box_size=80, resolution=1.25, max_conf=2000)
def run(name):
    assert dk.new_log(os.path.join(root, name + ".dat"))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        dk.dockSE3(os.path.join(bench, "Matched", name + "_r_u.pdb"), os.path.join(bench, "Matched", name + "_l_u.pdb"), batch_size=2)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
print("first target: %.3f s" % run("1SYN"))
pr = cProfile.Profile()
pr.enable()
dt = run("2SYN")
pr.disable()
n = dk.rot.R.shape[0]
print("second target: %.3f s for %d rotations = %.0f rot/s" % (dt, n, n / dt))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

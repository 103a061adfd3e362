"""Where the time of one replay of local_test.py goes (cProfile of scripts/replay_local_test.py on a synthetic target)."""
import cProfile
import os
import pstats
import runpy
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
t0 = time.time()
from test_replay_local_test import make_benchmark
root = tempfile.mkdtemp(prefix="dlpd_time_replay_")
make_benchmark(root)
os.environ.update({"DLPD_DATA_DIR": os.path.join(root, "data"), "DLPD_MODELS_DIR": os.path.join(root, "models"),
                   "DLPD_LOG_DIR": os.path.join(root, "log"), "DLPD_ALLOW_GENERATED_ROTATIONS": "1"})
os.makedirs(os.path.join(root, "log", "LocalDebugSE3"), exist_ok=True)
print("benchmark written in %.1f s" % (time.time() - t0), flush=True)
sys.argv = ["replay_local_test.py", "-angle_inc", "20", "-seed", "7", "-init_weights", "1", "-report", "1", "-threshold_clash", "40.0",
            "-rewrite", "1"] + sys.argv[1:]
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "scripts", "replay_local_test.py"), run_name="__main__")
finally:
    pr.disable()
    print("replay took %.1f s" % (time.time() - t0), flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

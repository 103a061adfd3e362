"""Diagnostic: bench.py's dockE3 measurement alone (projection + E3MultiResRepr4x4(8) + engine per launch of 16 rotations), meant
to be run under `rocprofv3 --kernel-trace --stats` for the per-kernel split of the plugin's half."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.build()
import bench
d = bench.e3_measurement(torch.device("cuda:0"), 16)
print(json.dumps({k: v for k, v in d.items() if k != "workload"}))

#!/bin/bash
# the faster atomic-free histogram under the probes that found the LDS-atomics interaction
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python scripts/search_race_probe.py 50 repr 2>&1 | tail -3
timeout 900 python scripts/stage_race_probe.py 300 repr topk 2>&1 | tail -4

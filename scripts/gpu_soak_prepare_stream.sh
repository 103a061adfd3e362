#!/bin/bash
# after the LDS atomics went: N sweeps (three targets, 20-degree set) with the next target prepared on a stream of its own --
# the configuration that first showed the perturbation (about 1 sweep in 20) -- against the plain sweep
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 2400 python scripts/sweep_diff_probe.py ${1:-100} stream 2>&1 | tail -6

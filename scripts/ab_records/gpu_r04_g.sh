#!/bin/bash
# round 4, call g: K3<160> with ten half-size filter waves (15 waves, 128 registers), roles dealt over the SIMDs (k3m10s) or
# transform waves first (k3m10), against the 5 + 5 default; parity of the variants against the channel-owning reference
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r04_g
mkdir -p $OUT
for v in k3m10s k3m10; do
  DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "role_split or real or hidden or config5 or candidate" > $OUT/pytest_$v.log 2>&1
  echo $v; tail -2 $OUT/pytest_$v.log
done
for wl in real c48l80; do
  bash scripts/gpu_ab_now.sh r04_g_$wl 60 --workload $wl --no_pmc --gather_rotations 0 --strong_s 0 -- default k3m10s k3m10
done
timeout 300 python scripts/conv_bench.py > $OUT/conv_bench.txt 2>&1; tail -7 $OUT/conv_bench.txt

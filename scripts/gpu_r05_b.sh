#!/bin/bash
# round 5, call B: why one replay took minutes in call A; K1 formulations A/B; counter units; the new tests
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r05_b
mkdir -p $OUT
(time timeout 900 python scripts/time_replay.py) > $OUT/time_replay.txt 2>&1; grep -n "replay took\|benchmark written\|real" $OUT/time_replay.txt; sed -n '/cumulative/,+22p' $OUT/time_replay.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_atoms.py -q -m gpu -x -k "k1_role_split or e3_plugin or dockE3" --durations=5 > $OUT/pytest_new.log 2>&1; tail -12 $OUT/pytest_new.log
bench() { timeout 300 python bench.py --steps 40 --warmup 5 --cpu_rotations 0 --no_real_shapes --sustained_s 0 --strong_s 0 --gather_rotations 0 "$@" 2> $OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*'.ljust(44), 'ms/step %.3f | ' % d['ms_per_step'] + ' '.join('%s=%.3f' % (k, v['ms_per_launch']) for k, v in d['stages'].items()))
" || tail -5 $OUT/err.txt; }
for rep in a b; do
  for wl in config2 real c48l80; do
    bench --workload $wl --k1_form 1
    bench --workload $wl --k1_form 2
  done
done 2>&1 | tee $OUT/k1_ab.txt
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k1r_stamps.so timeout 300 python scripts/stamps_k1r.py config2 > $OUT/stamps_k1r_config2.txt 2>&1; cat $OUT/stamps_k1r_config2.txt | tail -16
DLPD_LIB_PATH=$ROOT/build_variants/libdlpd_k1r_stamps.so timeout 300 python scripts/stamps_k1r.py real > $OUT/stamps_k1r_real.txt 2>&1; cat $OUT/stamps_k1r_real.txt | tail -16
bash scripts/pmc_units_probe.sh r05_b 2>&1 | tail -14
timeout 1500 python -m pytest tests/test_replay_local_test.py -q -m gpu -x --durations=5 -s > $OUT/pytest_replay.log 2>&1; tail -15 $OUT/pytest_replay.log

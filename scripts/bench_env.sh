#!/bin/bash
# A/B bench over env settings: each arg is "NAME=VALUE[,NAME=VALUE...]"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() { timeout 200 python bench.py --steps 40 --warmup 5 --cpu_rotations 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s poses/s %.4g rot/s %.1f | '%(sys.argv[1], d['value'], d['rot_per_s']) + ' '.join('%s=%.3f'%(k.split('_')[0],v['ms_per_launch']) for k,v in d['stages'].items()))" "$1"; }
run default
for cfg in "$@"; do ( IFS=','; for kv in $cfg; do export "$kv"; done; run "$cfg" ); done

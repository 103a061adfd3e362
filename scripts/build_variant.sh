#!/bin/bash
# build a variant library: scripts/build_variant.sh <name> "<flags for dlpd_corr>" "<flags for dlpd_k2>"
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/deeplocalproteindocking_amd/csrc
mkdir -p $ROOT/build_variants /tmp/dlpdv
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -I $C"
hipcc $F -I $ROOT/include $2 -c $C/dlpd_corr.hip -o /tmp/dlpdv/c.o 2>&1 | grep -E "error|VGPRs:|Spill:" || true
hipcc $F -I $ROOT/include -fno-slp-vectorize $3 -c $C/dlpd_k2.hip -o /tmp/dlpdv/k.o 2>&1 | grep -E "error|VGPRs:|Spill:" || true
hipcc $F -c $C/dlpd_topk.hip -o /tmp/dlpdv/t.o 2>&1 | grep -E "error" || true
hipcc $F -c $C/dlpd_atoms.hip -o /tmp/dlpdv/a.o 2>&1 | grep -E "error" || true
hipcc $F -c $C/dlpd_conv.hip -o /tmp/dlpdv/v.o 2>&1 | grep -E "error" || true
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_variants/libdlpd_$1.so /tmp/dlpdv/c.o /tmp/dlpdv/k.o /tmp/dlpdv/t.o /tmp/dlpdv/a.o /tmp/dlpdv/v.o
ls -la $ROOT/build_variants/libdlpd_$1.so | awk '{print $5,$9}'

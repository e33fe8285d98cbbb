#!/bin/bash
# A second libdrx built with extra compiler flags, for A/B runs on one box:
#   bash scripts/build_variant.sh <name> "<flags>"   ->  drecpy_amd/csrc/build/libdrx_<name>.so
#   DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_<name>.so python bench.py ...     (drecpy_amd/_lib.py loads that file instead)
set -eu
NAME=$1; FLAGS=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/drecpy_amd/csrc/build/var_$NAME
mkdir -p $OBJ
SRCS="drx_cdae.hip drx_sort.hip drx_topk.hip drx_idmap.hip drx_sampler.hip drx_shard.hip drx_comm.hip drx_generic.hip drx_caser.hip drx_dmf.hip"
pids=""
for s in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I $ROOT/include -I $ROOT/drecpy_amd/csrc $FLAGS -c $ROOT/drecpy_amd/csrc/$s -o $OBJ/$s.o &
  pids="$pids $!"
done
for s in drx_host.cpp drx_shard_phase.cpp; do
  g++ -pthread -O3 -fPIC -std=c++17 -I $ROOT/include -I $ROOT/drecpy_amd/csrc -c $ROOT/drecpy_amd/csrc/$s -o $OBJ/$s.o
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/drecpy_amd/csrc/build/libdrx_$NAME.so $OBJ/*.o -ldl
echo $ROOT/drecpy_amd/csrc/build/libdrx_$NAME.so

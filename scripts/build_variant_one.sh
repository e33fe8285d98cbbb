#!/bin/bash
# A variant of libdrx that differs from the shipped one in ONE translation unit (the others are linked from the regular build):
#   bash scripts/build_variant_one.sh <name> <source.hip> "<flags>"   ->  drecpy_amd/csrc/build/libdrx_<name>.so
# (scripts/build_variant.sh rebuilds everything: 9 compilations where a kernel experiment touches one file)
set -eu
NAME=$1; SRC=$2; FLAGS=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/drecpy_amd/csrc/build
mkdir -p $OBJ/var_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I $ROOT/include -I $ROOT/drecpy_amd/csrc $FLAGS -c $ROOT/drecpy_amd/csrc/$SRC -o $OBJ/var_$NAME/$SRC.o
OTHERS=$(ls $OBJ/*.o | grep -v "/$SRC.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $OBJ/libdrx_$NAME.so $OBJ/var_$NAME/$SRC.o $OTHERS
echo $OBJ/libdrx_$NAME.so

#!/bin/bash
# usage: tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]: a second build of the library with ONE source file compiled with
# extra flags (ablations, A/B timing through I64H_LIBRARY=tools/ab/libi64hip_<name>.so); the other objects are the in-tree ones
set -e
NAME=$1; SRC=$2; shift 2
HERE=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $HERE/tools/ab
OBJ=$HERE/tools/ab/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I $HERE/include -I $HERE/vllm_i64_amd/csrc "$@" -c $HERE/vllm_i64_amd/csrc/$SRC -o $OBJ
OTHERS=$(ls $HERE/vllm_i64_amd/build/*.o | grep -vF "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJ -o $HERE/tools/ab/libi64hip_$NAME.so
echo $HERE/tools/ab/libi64hip_$NAME.so

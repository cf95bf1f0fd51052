#!/bin/bash
# A/B builds of ONE kernel file with extra -D flags, linked against the tree's other objects:
#   scripts/diag/build_variant.sh <name> <file.hip> [flags...]   ->  ab/lib_<name>.so   (select with NERF_SOS_HIP_LIB=...)
# (ab/ is git-ignored but travels to the GPU box; exp/ does not)
set -e
NAME=$1; FILE=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/nerf-sos_amd/csrc
mkdir -p $ROOT/ab
OBJ=$ROOT/ab/${NAME}_$(basename $FILE .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I$ROOT/include -I$CS -Wall -Wno-unused-function "$@" -c $CS/$FILE -o $OBJ
OTHERS=$(ls $CS/*.o | grep -v "/$(basename $FILE .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/lib_$NAME.so $OBJ $OTHERS
echo built ab/lib_$NAME.so

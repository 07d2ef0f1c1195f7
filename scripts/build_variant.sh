#!/bin/bash
# A/B builds of one source file: bash scripts/build_variant.sh <name> <file.hip> [-DFOO=1 ...]  ->  madrigal_amd/lib/ab/lib<name>.so
# (the other objects come from the regular build; load a variant with MDG_AB_LIB=<path> in the scripts that honour it)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p madrigal_amd/lib/ab
obj=madrigal_amd/lib/ab/${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off -I include "$@" -c madrigal_amd/csrc/$src -o $obj
others=$(ls madrigal_amd/lib/obj/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o madrigal_amd/lib/ab/lib${name}.so $obj $others
echo built madrigal_amd/lib/ab/lib${name}.so

#!/bin/bash
# Builds a VARIANT of libclx.so for an A/B measurement: recompiles one source with extra -D flags and links it with
# the regular objects.  Usage: tools/build_variant.sh <tag> <source.hip> [-DFLAG ...]
#   -> cellulus_amd/libclx.so.<tag>   (load it with CLX_LIB=cellulus_amd/libclx.so.<tag> in the tools that honour it)
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
python -c "import cellulus_amd._build as b; b.build()" >/dev/null
obj=/tmp/clx_variant_$tag.o
extra=""
case "$src" in meanshift.hip|seeds.hip) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c cellulus_amd/csrc/$src -o $obj
objs=$(ls cellulus_amd/csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cellulus_amd/libclx.so.$tag $objs $obj
echo cellulus_amd/libclx.so.$tag

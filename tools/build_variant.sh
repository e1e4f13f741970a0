#!/bin/bash
# Build a variant of ONE kernel source into its own shared library for A/B timing on the GPU box:
#   tools/build_variant.sh <source.hip> <tag> <extra hipcc flags...>   ->  vip-cup-2022_amd/variants/libvipcup_<tag>.so  (use with VIP_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
SRC=$1; TAG=$2; shift 2
P=vip-cup-2022_amd
python $P/build.py > /dev/null
mkdir -p vip-cup-2022_amd/variants
OBJ=/tmp/variant_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$P/csrc -Wno-unused-result -ffp-contract=fast "$@" -x hip -c $P/csrc/$SRC -o $OBJ
OBJS=$(ls $P/build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o vip-cup-2022_amd/variants/libvipcup_$TAG.so $OBJS $OBJ -lpthread
echo vip-cup-2022_amd/variants/libvipcup_$TAG.so

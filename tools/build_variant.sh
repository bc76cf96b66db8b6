#!/bin/bash
# build a variant of libspt_hip.so: tools/build_variant.sh <name> <file.hip> <extra flags...>
# -> gpurun_variants/lib_<name>.so = the in-tree objects with <file> recompiled under the flags
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
OBJ=superpoint_transformer_amd/lib/obj
mkdir -p gpurun_variants /tmp/variant_$NAME
EXTRA=""
case $SRC in edge_attn_mfma.hip|edge_attn_el.hip|fused_mlp.hip|fused_mlp_dma.hip) EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $EXTRA "$@" \
  -c superpoint_transformer_amd/csrc/$SRC -o /tmp/variant_$NAME/${SRC%.hip}.o
OBJS=$(ls $OBJ/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/lib_$NAME.so $OBJS /tmp/variant_$NAME/${SRC%.hip}.o
echo built gpurun_variants/lib_$NAME.so

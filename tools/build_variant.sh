#!/bin/bash
# dev tool: a second build of the library with extra compiler flags -> tools/bin/libea_hip_<name>.so (A/B with EA_HIP_LIB=...)
#   tools/build_variant.sh <name> [-DEA_NT_STORES ...]
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
mkdir -p $R/tools/bin/$NAME
OBJS=""
for f in $R/efficient-attention_amd/csrc/ea_*.hip; do
  o=$R/tools/bin/$NAME/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -cuid=$(basename ${f%.hip}) "$@" -I$R/include -I$R/efficient-attention_amd/csrc -c $f -o $o &
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/tools/bin/libea_hip_$NAME.so
rm -rf $R/tools/bin/$NAME
echo built $R/tools/bin/libea_hip_$NAME.so

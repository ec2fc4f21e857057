#!/bin/bash
# dev tool: libea_hip with the kernels' phase time stamps compiled in (-DEA_PROFILE)
#   -> tools/bin/libea_hip_prof.so ; use with EA_HIP_LIB=tools/bin/libea_hip_prof.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tools/bin/prof
OBJS=""
for f in $R/efficient-attention_amd/csrc/ea_*.hip; do
  o=$R/tools/bin/prof/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find $R/efficient-attention_amd/csrc -name '*.h' -newer $o)" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DEA_PROFILE -I$R/include -I$R/efficient-attention_amd/csrc -c $f -o $o &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/tools/bin/libea_hip_prof.so
echo built $R/tools/bin/libea_hip_prof.so

#!/bin/bash
# dev tool: libea_hip with the landmark kernel's phase time stamps compiled in -> tools/bin/libea_hip_prof.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()"
mkdir -p $R/tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DEA_LMK_PROFILE -I$R/include -I$R/efficient-attention_amd/csrc \
  -c $R/efficient-attention_amd/csrc/ea_lara_landmark.hip -o $R/tools/bin/ea_lara_landmark_prof.o
OBJS=$(ls $R/efficient-attention_amd/lib/*.o | grep -v ea_lara_landmark.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/tools/bin/ea_lara_landmark_prof.o -o $R/tools/bin/libea_hip_prof.so
echo built $R/tools/bin/libea_hip_prof.so

#!/bin/bash
mkdir -p gpurun_out
for wl in cfg3 cfg2; do
  EA_HIP_LIB=$PWD/tools/bin/libea_hip_prof.so python bench.py --attn lara --workload $wl --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-other-workloads > /dev/null 2> gpurun_out/wgprof2_lara_$wl.err
  grep -A4 "proj_rs mode\|dgrad_fin mode" gpurun_out/wgprof2_lara_$wl.err | tail -24 > gpurun_out/wgprof2_lara_$wl.txt
done
cat gpurun_out/wgprof2_lara_cfg3.txt gpurun_out/wgprof2_lara_cfg2.txt | cut -c1-330

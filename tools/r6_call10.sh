#!/bin/bash
# round 6, GPU call 10: workgroup timelines of the -DEA_PROFILE build (prologue / epilogue shares of the LARA passes)
mkdir -p gpurun_out
for wl in cfg3 cfg2; do
  EA_HIP_LIB=$PWD/tools/bin/libea_hip_prof.so python bench.py --attn lara --workload $wl --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-other-workloads > /dev/null 2> gpurun_out/wgprof_lara_$wl.err
  grep -A4 "mode" gpurun_out/wgprof_lara_$wl.err | tail -120 > gpurun_out/wgprof_lara_$wl.txt
done
EA_HIP_LIB=$PWD/tools/bin/libea_hip_prof.so python bench.py --attn eva --workload cfg3 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-other-workloads > /dev/null 2> gpurun_out/wgprof_eva_cfg3.err
tail -60 gpurun_out/wgprof_eva_cfg3.err > gpurun_out/wgprof_eva_cfg3.txt
tail -30 gpurun_out/wgprof_lara_cfg3.txt | cut -c1-300

#!/bin/bash
cd /root/repo
EA_HIP_LIB=$PWD/tools/bin/libea_hip_prof.so python tools/time_lara.py lara 2 2> gpurun_out/stamps_lmk.txt >/dev/null
grep -i "lmk" gpurun_out/stamps_lmk.txt | tail -2 | cut -c1-800

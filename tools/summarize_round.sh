#!/bin/bash
# after tools/profiles_round.sh <tag> (+ the LM lines of tools/r6_call25.sh) came back in gpurun_out/: the committed summaries
TAG=${1:-r06}
for a in lara eva; do
  python tools/summarize_profile.py gpurun_out/prof_${TAG}_$a $TAG $a profiles
  for wl in cfg2 cfg5; do python tools/summarize_profile.py gpurun_out/prof_${TAG}${wl}_$a ${TAG}$wl $a profiles $wl; done
done
python tools/summarize_profile.py gpurun_out/prof_${TAG}_softmax $TAG softmax profiles
python tools/summarize_profile.py gpurun_out/prof_${TAG}cfg5_softmax ${TAG}cfg5 softmax profiles cfg5
python tools/summarize_profile.py gpurun_out/prof_${TAG}lm_causal_eva ${TAG}lm causal_eva profiles lm
for a in lara eva softmax; do python tools/summarize_sq.py gpurun_out/sq_$a $TAG $a profiles; done
python tools/summarize_sq.py gpurun_out/sq_softmax_cfg5 ${TAG}cfg5 softmax profiles
python tools/summarize_sq.py gpurun_out/sq_causal_eva_lm ${TAG}lm causal_eva profiles
python tools/resource_report.py --out profiles/${TAG}_resources.md

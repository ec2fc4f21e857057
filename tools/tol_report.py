#!/usr/bin/env python3
"""Worst observed parity errors per (test file, variant, dtype) from an EA_TEST_ERR_LOG file."""
import collections, re, sys
worst = collections.defaultdict(lambda: [0.0, 0.0, ""])
for line in open(sys.argv[1]):
    name, mx, rms = line.rstrip("\n").split("\t")
    mx, rms = float(mx), float(rms)
    if mx != mx: continue
    m = re.match(r"tests/(test_[a-z_]+)\.py::[a-z_]+\[(.*)\]", name)
    if not m: continue
    f, pid = m.group(1), m.group(2)
    dt = "fp16" if "fp16" in pid else "bf16"
    attn = next((a for a in ("scatterbrain", "causal_eva", "performer", "softmax", "local", "lara", "eva", "ra") if a in pid), "other")
    k = (f, attn, dt)
    if mx > worst[k][0]: worst[k][0] = mx; worst[k][2] = pid
    if rms > worst[k][1]: worst[k][1] = rms
for k in sorted(worst):
    print("%-28s %-13s %-5s max %.4f rms %.4f   (%s)" % (k + (worst[k][0], worst[k][1], worst[k][2])))

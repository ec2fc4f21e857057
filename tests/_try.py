import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/efficient-attention_amd", "/root/repo/tests", "/root/repo/tests/golden"]
import torch, cases
from gpu_checks import check_module_case
pref = tuple(sys.argv[1].split(",")); bwd = len(sys.argv) < 3 or sys.argv[2] != "fwd"
for name in cases.CASES:
    if not name.startswith(pref): continue
    for mode in ("eval", "train"):
        try:
            e = check_module_case(name, mode, backward=bwd)
            print("OK  ", name, mode, " ".join("%s=%.1e" % (k, max(v)) for k, v in e.items()))
        except Exception as ex:
            import traceback
            msg = str(ex)
            print("FAIL", name, mode, msg[:600] if "tolerance" in msg else traceback.format_exc()[-700:])

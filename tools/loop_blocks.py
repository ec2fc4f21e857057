#!/usr/bin/env python3
"""dev tool: basic-block structure of the loops of every kernel in a hipcc -S output (a loop that falls apart into dozens of
small blocks has runtime branches in it: nothing is scheduled across them).  usage: loop_blocks.py file.s [filter]"""
import re, collections, subprocess, sys
def main(path, flt=""):
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'_ZN2ea.*:\s*;\s*@', l)]
    for si in starts:
        sym = lines[si].split(':')[0]
        ends = [i for i, l in enumerate(lines) if i > si and '.amdhsa_kernel' in l]
        if not ends: continue
        body = lines[si:ends[0]]
        blocks = collections.OrderedDict(); cur = 'entry'; blocks[cur] = []
        for l in body:
            t = l.strip()
            if re.match(r'\.LBB\d+_\d+:', t):
                cur = t.split(':')[0] + (' L' if 'Loop' in t else ''); blocks[cur] = []; continue
            if not t or t.startswith(';') or t.startswith('.'): continue
            blocks[cur].append(t.split()[0])
        lb = [len(v) for k, v in blocks.items() if k.endswith('L')]
        dem = subprocess.run(['c++filt', sym], capture_output=True, text=True).stdout.strip()
        if flt in dem:
            print(dem[:110], '| loop blocks', len(lb), 'instrs', sum(lb))
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

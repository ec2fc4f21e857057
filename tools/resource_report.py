#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in libea_hip.so, read from the code-object metadata.

    python tools/resource_report.py [--lib PATH] [--out profiles/r05_resources.md] [--scratch-only]

`llvm-objdump --offloading` splits the fat binary into one gfx950 code object per translation unit; the
amdhsa metadata notes of each (`llvm-readelf --notes`) carry, per kernel, `.vgpr_count`, `.agpr_count`, `.sgpr_count`,
`.private_segment_fixed_size` (scratch bytes per lane: spills), `.group_segment_fixed_size` (static LDS) and
`.max_flat_workgroup_size`.  A non-zero scratch size means the kernel spills.  Exit code 1 with --fail-on-scratch
when any kernel has scratch (tests/test_resources.py uses the function form)."""
import argparse
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "efficient-attention_amd", "lib", "libea_hip.so")
LLVM = os.environ.get("EA_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:
            pass
    return {n: n for n in names}


def kernels(lib=LIB):
    """-> list of dicts (name, demangled, vgpr, agpr, sgpr, scratch, lds, wg), one per kernel of the library."""
    tmp = tempfile.mkdtemp(prefix="ea_res_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy2(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   stdout=subprocess.PIPE, text=True, check=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'\"")
                if key == "agpr_count":                       # first key of a kernel record (keys are sorted)
                    cur = {"agpr": int(val)}
                    rows.append(cur)
                elif cur is not None:
                    if key == "name":
                        cur["name"] = val
                    elif key == "vgpr_count":
                        cur["vgpr"] = int(val)
                    elif key == "sgpr_count":
                        cur["sgpr"] = int(val)
                    elif key == "private_segment_fixed_size":
                        cur["scratch"] = int(val)
                    elif key == "group_segment_fixed_size":
                        cur["lds"] = int(val)
                    elif key == "max_flat_workgroup_size":
                        cur["wg"] = int(val)
        rows = [r for r in rows if "name" in r]
        dm = _demangle([r["name"] for r in rows])
        for r in rows:
            r["demangled"] = dm[r["name"]]
            for k in ("vgpr", "sgpr", "scratch", "lds", "wg"):
                r.setdefault(k, 0)
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _short(d):
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(.*\)$", "", d)
    return d.replace("ea::", "")


def report(lib=LIB, scratch_only=False):
    rows = kernels(lib)
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    spill = [r for r in rows if r["scratch"] > 0]
    out = ["# Kernel resources of libea_hip.so (code-object metadata, gfx950)", "",
           "library sha256 `%s`; %d kernels, %d with scratch (spills)." % (sha, len(rows), len(spill)), "",
           "| kernel | VGPR | AGPR | SGPR | scratch B/lane | static LDS B | max WG |", "|---|---|---|---|---|---|---|"]
    for r in sorted(rows, key=lambda r: (-r["scratch"], -r["vgpr"], r["demangled"])):
        if scratch_only and not r["scratch"]:
            continue
        out.append("| `%s` | %d | %d | %d | %d | %d | %d |" % (_short(r["demangled"]), r["vgpr"], r["agpr"], r["sgpr"],
                                                          r["scratch"], r["lds"], r["wg"]))
    return "\n".join(out) + "\n", rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--out", default=None)
    ap.add_argument("--scratch-only", action="store_true")
    ap.add_argument("--fail-on-scratch", action="store_true")
    ap.add_argument("--grep", default=None, help="only kernels whose demangled name contains this")
    a = ap.parse_args()
    text, rows = report(a.lib, a.scratch_only)
    if a.grep:
        text = "\n".join(l for l in text.splitlines() if a.grep in l or not l.startswith("| `")) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    else:
        sys.stdout.write(text)
    if a.fail_on_scratch and any(r["scratch"] for r in rows):
        sys.exit(1)

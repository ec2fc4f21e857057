#!/usr/bin/env python3
"""Condense a tools/pmc_sq.sh output directory (gpurun_out/sq_<attn>) into profiles/<tag>_<attn>_sq.md and
profiles/sq_<attn>.json: per C-ABI entry point, averaged over its launches,
  * where the wave cycles go: SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY as fractions of SQ_WAVE_CYCLES
    (issuing / issue-stalled / parked at s_waitcnt or a barrier; MI355X_MICROARCH.md, rocprofv3 PMC slots),
  * MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs),
  * VALU / MFMA / LDS instruction counts per wave, LDS bank-conflict share of the LDS-active cycles, LDS busy share,
    average waves in flight per CU.
usage: summarize_sq.py <src dir> <tag> <attn> [outdir]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import entry_of, lib_sha     # noqa: E402

CUS, SIMDS = 256, 4


def load(src):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub, fn in (("p1", "a"), ("p2", "b"), ("p3", "c")):
        path = os.path.join(src, sub, fn + "_counter_collection.csv")
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            e = entry_of(r["Kernel_Name"])
            if e:
                acc[e][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {e: {c: sum(v) / len(v) for c, v in d.items()} for e, d in acc.items()}


def main(src, tag, attn, outdir="profiles"):
    data = load(src)
    rows, out = [], {}
    for e, c in sorted(data.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        waves = c.get("SQ_WAVES", 0.0)
        if not wc or not waves:
            continue
        # SQ_BUSY_CU_CYCLES = sum over the CUs of their busy cycles (fq: 33.3 M = 256 CUs x 130 k cycles = the 65 us launch at
        # 2.0 GHz); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs and includes the launch overhead, so the CU-busy
        # cycles are the denominator.  Cross-check: SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA for v_mfma_f32_16x16x32.
        gui = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        mfma_busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        mfma_util = mfma_busy / (gui * SIMDS) if gui else None
        lds_act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        rec = {
            "waves": int(waves),
            "active_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
            "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
            "parked_frac": c.get("SQ_WAIT_ANY", 0.0) / wc,
            "mfma_util": mfma_util,
            "valu_per_wave": c.get("SQ_INSTS_VALU", 0.0) / waves,
            "mfma_per_wave": c.get("SQ_INSTS_MFMA", 0.0) / waves,
            "lds_per_wave": c.get("SQ_INSTS_LDS", 0.0) / waves,
            "lds_conflict_share": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds_act) if lds_act else 0.0,
            "lds_busy_frac": (lds_act / gui) if gui else None,
            # quad-cycles of resident waves per busy CU-cycle = average waves in flight per CU
            "waves_in_flight_per_cu": (wc * 4 / gui) if gui else None,
            "busy_cu_cycles": gui,
        }
        out[e] = rec
        f = lambda v, p=2: "-" if v is None else ("%%.%df" % p) % v
        rows.append("| %s | %d | %s | %s | %s | %s | %.0f | %.0f | %.0f | %s | %s | %s |" % (
            e, rec["waves"], f(rec["active_frac"]), f(rec["issue_stall_frac"]), f(rec["parked_frac"]),
            f(None if mfma_util is None else 100 * mfma_util, 1), rec["valu_per_wave"], rec["mfma_per_wave"], rec["lds_per_wave"],
            f(rec["lds_conflict_share"]), f(rec["lds_busy_frac"]), f(rec["waves_in_flight_per_cu"], 1)))
    out["_lib_sha256"] = lib_sha()
    os.makedirs(outdir, exist_ok=True)
    json.dump(out, open(os.path.join(outdir, "sq_%s.json" % attn), "w"), indent=1)
    hdr = ("SQ / MFMA counters per launch, rocprofv3 --pmc (three counter-only passes, tools/pmc_sq.sh) over the %s layer of "
           "bench.py's default workload ([128,28,28,192], fwd+bwd).\n"
           "issuing / issue-stalled / parked = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES; "
           "MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs) (matrix-pipe busy share of the busy CU time; "
           "SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA for the 16x16x32 bf16 MFMA); LDS conflicts = SQ_LDS_BANK_CONFLICT / "
           "SQ_LDS_IDX_ACTIVE; LDS busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; waves/CU = 4 x SQ_WAVE_CYCLES / "
           "SQ_BUSY_CU_CYCLES (quad-cycle counters).\n\n" % attn)
    table = ["| C-ABI entry | waves | issuing | issue-stalled | parked | MFMA util % | VALU/wave | MFMA/wave | LDS/wave | LDS conflicts | LDS busy | waves/CU |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"] + rows
    open(os.path.join(outdir, "%s_%s_sq.md" % (tag, attn)), "w").write(hdr + "\n".join(table) + "\n")
    print("\n".join(table))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "profiles")

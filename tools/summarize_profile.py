#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>_<attn>) into the small,
committed files under profiles/:  <tag>_<attn>_kernel_stats.csv (rocprofv3 --kernel-trace --stats
summary, ea:: kernels + the top library kernels), <tag>_<attn>_hbm.md and pmc_<attn>.json (per
C-ABI entry point: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
MI355X_MICROARCH.md 'HBM': FETCH_SIZE reports half of a 16 B/lane streaming read)."""
import collections
import csv
import json
import os
import sys

# kernel-name substring -> C-ABI entry point (first match wins)
KERNEL_TO_ENTRY = [
    ("lara_fq_kernel<", "ea_lara_bwd_q_fused"), ("lara_fk_kernel<", "ea_lara_bwd_k_fused"),
    ("lara_fin_kernel<", "ea_lara_bwd_finish"),
    ("lmk2_kernel<64, false>", "ea_lara_landmarks_fwd"), ("lmk2_kernel<64, true>", "ea_lara_landmarks_bwd"),
    ("wgrad_kernel<", "ea_wgrad"), ("part_sum_kernel", "ea_part_sum"), ("multi_sum_kernel", "ea_multi_sum"),
    ("stream_copy_kernel", "ea_stream_copy"),
    ("lin_kernel<ea::BF16, 6, 2, true", "ea_linear[fp32 in]"), ("lin_kernel<", "ea_linear"),
    ("colsum_f32_kernel", "ea_colsum_f32 / ea_bias_grad(finish)"),
    ("lara_sample_kernel<", "ea_lara_sample"), ("pool2d_", "ea_adaptive_pool2d"),
    ("sb_fwd_kernel<", "ea_scatter_fwd"), ("sb_bwd_kernel<ea::BF16, false>", "ea_scatter_bwd_window"),
    ("sb_bwd_kernel<ea::BF16, true>", "ea_scatter_bwd_global"),
    ("lara_y_kernel<ea::BF16, 64, 3,", "ea_scatter_kmax / ea_performer_kmax"),
    ("lara_y_kernel<ea::BF16, 64, 4,", "ea_scatter_kv / ea_performer_kv"),
    ("lara_x_kernel<ea::BF16, 64, 4, 0,", "ea_lara_out_fwd"), ("lara_x_kernel<ea::BF16, 64, 4, 7,", "ea_lara_out_fwd"), ("lara_x_kernel<ea::BF16, 64, 4, 1,", "ea_lara_bwd_q"),
    ("lara_x_kernel<ea::BF16, 64, 4, 2,", "ea_lara_bwd_k"), ("lara_x_kernel<ea::BF16, 64, 4, 3,", "ea_lara_bwd_qcorr"),
    ("lara_x_kernel<ea::BF16, 64, 4, 4,", "ea_performer_out"), ("lara_x_kernel<ea::BF16, 64, 4, 5,", "ea_performer_bwd_q"),
    ("lara_x_kernel<ea::BF16, 64, 4, 6,", "ea_performer_bwd_k"),
    ("lara_y_kernel<ea::BF16, 64, 0,", "ea_lara_stats_fwd"), ("lara_y_kernel<ea::BF16, 64, 1,", "ea_lara_bwd_qstats"),
    ("lara_y_kernel<ea::BF16, 64, 2,", "ea_lara_bwd_kstats"), ("lara_y_kernel<ea::BF16, 64, 5,", "ea_performer_bwd_qstats"),
    ("lara_merge_fwd_kernel", "ea_lara_merge_fwd"), ("lara_merge_bwd_kernel", "ea_lara_merge_bwd"),
    ("win_fwd_kernel<", "ea_window_attn_fwd"), ("win_bwd_finish_kernel<", "ea_window_attn_bwd(finish)"),
    ("win_bwd_kernel<", "ea_window_attn_bwd"),
    ("chunk_mean_fwd_r_kernel<", "ea_eva_chunk_mean_fwd"), ("chunk_mean_bwd_r_kernel<", "ea_eva_chunk_mean_bwd"),
    ("beta_fwd_r_kernel<", "ea_eva_beta_fwd"), ("beta_bwd_r_kernel<", "ea_eva_beta_bwd"),
    ("proj_rs_kernel<", "ea_linear[fp32 in]"), ("dgrad_rs_kernel<", "ea_linear_dgrad"), ("dgrad_fin_kernel<", "ea_linear_dgrad_finish"),
    ("chunk_mean_fwd_kernel<", "ea_eva_chunk_mean_fwd"), ("chunk_mean_bwd_kernel<", "ea_eva_chunk_mean_bwd"),
    ("beta_fwd_kernel<", "ea_eva_beta_fwd"), ("beta_bwd_kernel<", "ea_eva_beta_bwd"),
    ("sm_fwd_kernel<ea::BF16, 64", "ea_softmax_attn_fwd"), ("sm_bwd_dq_kernel<ea::BF16, 64", "ea_softmax_attn_bwd(dq)"),
    ("sm_bwd_dkv_kernel<ea::BF16, 64", "ea_softmax_attn_bwd(dkv)"),
    ("colsum_part_kernel", "ea_bias_grad"), ("colsum_f32_kernel", "ea_colsum_f32 / ea_bias_grad(finish)"),
    ("slice_sum_kernel", "ea_slice_sum"),
    ("rows_mlp_fwd_kernel", "ea_rows_mlp_fwd"), ("rows_mlp_bwd_kernel", "ea_rows_mlp_bwd"),
    ("table_bias_fwd_kernel", "ea_table_bias_fwd"), ("table_bias_bwd_kernel", "ea_table_bias_bwd"),
    ("multi_cast_kernel", "ea_multi_cast"),
    ("seglin_col_kernel<ea::BF16, false", "ea_lara_seglin_fwd"), ("seglin_col_kernel<ea::BF16, true, 0>", "ea_lara_seglin_bwd(dq/dk)"),
    ("seglin_col_kernel<ea::BF16, true", "ea_lara_seglin_bwd_fin(dq/dk + finish)"),
    ("seglin_dg_kernel<", "ea_lara_seglin_bwd(dG)"), ("fold_", "ea_lara_fold"),
]


def lib_sha():
    """sha256 of the library the profile was taken with (the in-tree .so travels to the GPU box unchanged)."""
    import hashlib
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficient-attention_amd", "lib", "libea_hip.so")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.exists(path) else None


def entry_of(kname):
    for k, v in KERNEL_TO_ENTRY:
        if k in kname:
            return v
    return None


def main(src, tag, attn, outdir, workload="default workload"):
    os.makedirs(outdir, exist_ok=True)
    stats = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
    keep = [r for r in stats if "ea::" in r["Name"]] + [r for r in stats if "ea::" not in r["Name"]][:12]
    with open(os.path.join(outdir, "%s_%s_kernel_stats.csv" % (tag, attn)), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in sorted(keep, key=lambda r: -float(r["TotalDurationNs"])):
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for cname, sub, fn in (("FETCH_SIZE", "pmc_fetch", "f"), ("WRITE_SIZE", "pmc_write", "w")):
        path = os.path.join(src, sub, fn + "_counter_collection.csv")
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            e = entry_of(r["Kernel_Name"])
            if e == "ea_stream_copy":               # bench.py's bandwidth yardstick, not part of a step
                continue
            if e and r["Counter_Name"] == cname:
                pmc[e][cname].append(float(r["Counter_Value"]))
    avg_ns = {entry_of(r["Name"]): float(r["AverageNs"]) for r in stats if entry_of(r["Name"])}
    out, lines = {}, ["| C-ABI entry | avg us | FETCH_SIZE raw KB | WRITE_SIZE KB | HBM bytes/launch (2xF+W) MB |", "|---|---|---|---|---|"]
    for e, d in sorted(pmc.items()):
        f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
        wv = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
        total = (2 * f + wv) * 1024
        out[e] = total
        lines.append("| %s | %.1f | %.0f | %.0f | %.1f |" % (e, avg_ns.get(e, 0) / 1e3, f, wv, total / 1e6))
    # bytes of one step: every dispatch of every ea:: kernel, summed, / the number of steps the counter pass ran (= the
    # dispatch count of a kernel launched once per step); `attention` = everything but the projection / reduction kernels
    nrows = {e: max(len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"])) for e, d in pmc.items()}
    steps = collections.Counter(nrows.values()).most_common(1)[0][0] if nrows else 0   # most kernels launch once per step
    if steps:
        proj = ("ea_linear", "ea_wgrad", "ea_part_sum", "ea_bias_grad", "ea_multi_sum", "ea_stream_copy")
        tot = {e: (2 * sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) + sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1))
               * 1024 * nrows[e] / steps for e, d in pmc.items()}
        out["_launches_per_step"] = {e: round(nrows[e] / steps, 2) for e in pmc}
        out["_step_traffic_bytes"] = sum(tot.values())
        out["_step_traffic_bytes_attention"] = sum(v for e, v in tot.items() if not e.startswith(proj))
        lines.append("")
        lines.append("Per step (all ea:: kernels x launches per step): %.1f MB; attention kernels only: %.1f MB."
                     % (out["_step_traffic_bytes"] / 1e6, out["_step_traffic_bytes_attention"] / 1e6))
    # kernel durations of the CAPTURED step (the kernel trace of the same command): bench.py reports them next to its own
    # eager HIP-event timings (roofline.rocprof_avg_us / frac_rocprof) -- the eager clock reads ~9 % long on short kernels
    out["_rocprof_avg_us"] = {e: round(v / 1e3, 3) for e, v in avg_ns.items() if e != "ea_stream_copy"}
    out["_lib_sha256"] = lib_sha()            # bench.py ignores the file once the library has changed
    # one counter file per (variant, workload): the default cfg3 (and causal_eva's lm) keep pmc_<attn>.json, cfg2 / cfg5 get a suffix
    sfx = "_" + workload if workload in ("cfg2", "cfg5") else ""
    json.dump(out, open(os.path.join(outdir, "pmc_%s%s.json" % (attn, sfx)), "w"), indent=1)
    open(os.path.join(outdir, "%s_%s_hbm.md" % (tag, attn)), "w").write(
        "HBM traffic per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --attn %s "
        "%s.\nFETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B; "
        "calibrated here on ea_lara_out_fwd whose WRITE_SIZE equals its 38.5 MB output exactly).\n\n" % (attn, workload)
        + "\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "profiles",
         sys.argv[5] if len(sys.argv) > 5 else "default workload")

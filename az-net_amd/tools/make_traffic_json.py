#!/usr/bin/env python3
"""profiles/roofline_traffic.json from the per-launch-shape PMC summaries of profile_round.sh.
usage: make_traffic_json.py <dir with fc_{fetch,write}_by_launch.csv, pmc_hbm.csv [, twopass_*]> <tag> > roofline_traffic.json
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import json
import os
import sys


def by_launch(path):
    return [r for r in csv.DictReader(open(path))]


def per_kernel(path):
    return {r["kernel"]: r for r in csv.DictReader(open(path))}


def main():
    d, tag = sys.argv[1], sys.argv[2]
    def hbm(pfx):
        f = by_launch(os.path.join(d, pfx + "fc_fetch_by_launch.csv"))
        w = by_launch(os.path.join(d, pfx + "fc_write_by_launch.csv"))
        return [int(float(a["FETCH_SIZE"]) * 1024 * 2 + float(b["WRITE_SIZE"]) * 1024) for a, b in zip(f, w)]
    main_b = hbm("")
    rows = 688
    out = {
        "source": "%s (profiles/%s_fc_*_by_launch.csv, %s_pmc_hbm.csv)" % (tag, tag, tag),
        "kernel": "the fc GEMM launches of one search in its whole-tree form (bench.py's `value`): k_fc_splitk12 int6 of the one "
                  "688-row pass, its int7 (k_fc_splitk)",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py <the `main` set of "
                   "az-net_amd/tools/profile_round.sh>",
        "correction": "FETCH_SIZE and WRITE_SIZE are in KB (x1024 here); FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)",
        "hbm_bytes_per_launch": int(sum(main_b) / len(main_b)),
        "per_launch_shape": {"int6, 688 rows (the full tree's unique rois, root last), k_fc_splitk12": main_b[0],
                             "int7_1|int7_2, 688 rows": main_b[1]},
        "algorithmic_bytes_per_launch": {
            "int6 by SURVEY 8(d) (weights 411 041 792 + biases; pool5 and slabs are NOT algorithmic)": 411041792 + 16384,
            "int6 with its un-fused operands (weights + pool5 rows x 100 352 + 16 slabs x rows x 16 384)": 411041792 + rows * 100352 + 16 * rows * 16384,
            "int7 (weights 20 971 520 + int6 rows x 16 384 + 8 slabs x rows x 5 120)": 20971520 + rows * 16384 + 8 * rows * 5120},
    }
    pk = per_kernel(os.path.join(d, "pmc_hbm.csv"))
    def kb(k):
        r = pk.get(k)
        return (float(r["avg_FETCH_SIZE"]) * 2 + float(r["avg_WRITE_SIZE"])) * 1024 if r else 0.0
    parts = {"int6": main_b[0], "int7": main_b[1], "k_fc_reduce": kb("k_fc_reduce"), "k_roi_pool": kb("k_roi_pool"),
             "k_tail_fused": kb("k_tail_fused")}
    out["per_image_all_head_kernels"] = {
        "whole_tree_form_bytes": int(sum(parts.values())),
        "whole_tree_form": " + ".join("%s %.3f GB" % (k, v / 1e9) for k, v in parts.items()),
        "algorithmic_bytes_one_pass_by_survey_8d": 432239616 + 21728 + 4 * 512 * 38 * 63 + rows * 244}
    if os.path.exists(os.path.join(d, "twopass_fc_fetch_by_launch.csv")):
        tp = hbm("twopass_")
        out["per_image_all_head_kernels"]["two_pass_form_fc_launches"] = {
            "int6 48 rows": tp[0], "int7 48 rows": tp[1], "int6 670 rows": tp[2], "int7 670 rows": tp[3]}
    out["why_above_algorithmic"] = ("the 688-row launch re-reads about half of the int6 weights (two m-tiles of <= 12 row strips per "
                                    "weight tile; the two readers share an XCD's L2 at the same time, which catches ~half of the "
                                    "second reads), writes / re-reads 16 split-K slabs; pool5 is materialised by k_roi_pool")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

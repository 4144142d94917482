#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide
prescribes) of `bench.py` into profiles/traffic.json (bytes per launch per pipeline stage) and a
small per-kernel summary CSV.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc -o fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc -o write -- python3 bench.py ...
    python tools/pmc_to_traffic.py gpurun_out/pmc profiles/r01 cfg3_1080p_full_chain

gfx950 corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; FETCH_SIZE reports exactly
half of the bytes of wide coalesced streaming reads, so it is doubled for the kernels whose loads
are 8-16 B per lane streams (checked here: bg kernel, known 548 MB read, raw counter 267 MB);
WRITE_SIZE is exact for streaming stores.
"""
import collections
import csv
import json
import os
import re
import sys

STAGE_OF_KERNEL = {
    "bg_mean_u8_fast_kernel": "bg", "bg_mean_u8_kernel": "bg", "gauss_fused_kernel": "gauss_fused", "gauss_mfma_kernel": "gauss_mfma",
    "morph_fused_kernel": "morph_fused", "morph_stream_kernel": "morph_fused", "ccl_init_kernel": "ccl_init", "ccl_frame_kernel": "ccl_frame", "ccl_link_kernel": "ccl_link",
    "ccl_flatten_kernel": "ccl_flatten", "ccl_rowscan_kernel": "ccl_rowscan",
    "ccl_rank_kernel": "ccl_rank", "ccl_paint_kernel": "ccl_paint",
    "ema_row_f32_kernel": "ema_row_f32", "row_is_f32_kernel": "ema_row_f32",
    "col_march_f32_kernel": "col_f32", "col_sym_f32_kernel": "col_f32", "unpack_bits_kernel": "mask_unpack", "unpack_bits_wide_kernel": "mask_unpack",
}
# 8-16 B/lane coalesced streaming loads: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM)
WIDE_STREAM_READS = {"bg", "gauss_fused", "gauss_mfma", "ema_row_f32", "col_f32"}


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"va::\(anonymous namespace\)::(?:\w+::)*(\w+)", r["Kernel_Name"])
        if m:
            agg[m.group(1)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    src, out_prefix, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, nf = per_kernel(os.path.join(src, "fetch_counter_collection.csv"))
    write, _ = per_kernel(os.path.join(src, "write_counter_collection.csv"))
    traffic = {}
    rows = []
    for k in sorted(set(fetch) | set(write)):
        stage = STAGE_OF_KERNEL.get(k, k)
        f_raw = fetch.get(k, 0.0) * 1024
        w_raw = write.get(k, 0.0) * 1024
        f_corr = f_raw * (2 if stage in WIDE_STREAM_READS else 1)
        traffic[stage] = int(f_corr + w_raw)
        rows.append((k, stage, nf.get(k, 0), int(f_raw), int(f_corr), int(w_raw), int(f_corr + w_raw)))
    with open(out_prefix + "_pmc_summary.csv", "w") as fh:
        fh.write("kernel,stage,launches,FETCH_SIZE_bytes_raw,fetch_bytes_corrected,WRITE_SIZE_bytes,hbm_bytes_per_launch\n")
        for r in rows:
            fh.write(",".join(str(x) for x in r) + "\n")
    tpath = os.path.join(os.path.dirname(out_prefix), "traffic.json")
    allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
    allt[workload] = traffic
    allt.setdefault("_source", {})[workload] = os.path.basename(out_prefix)
    json.dump(allt, open(tpath, "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()

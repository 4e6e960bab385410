#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/:

  profiles/<tag>_<dt>_kernel_stats.csv        rocprofv3 --kernel-trace --stats per-kernel summary
  profiles/<tag>_<dt>_bench_under_rocprof.json bench line of that same run
  profiles/<tag>_<dt>_pmc.json                 mean PMC counters per kernel (separate passes)
  profiles/traffic_<dt>.json                   what bench.py reports as roofline.traffic

HBM traffic follows MI355X_MICROARCH.md section "HBM": FETCH_SIZE and WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane streaming reads (which is
what every load of these kernels is), so it is doubled; WRITE_SIZE is exact for 16-byte stores.

usage: python3 tools/summarize_profiles.py <tag>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_means(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in d2.items()} for k, d2 in acc.items()}


def is_trunk(name, dt):
    # the 33 trunk launches: conv3x3_<dt>_kernel<64, false(shuffle), pool?, res?>
    return (f"conv3x3_{dt}_kernel<64, false" in name or f"conv3x3_{dt}_wino_kernel<64, false" in name
            or (dt == "f32" and "conv3x3_f32s_kernel<64, false" in name))


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    for dt in ("f32", "bf16"):
        stats = glob.glob(os.path.join(src, f"{dt}_trace", "**", "*_kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(dst, f"{tag}_{dt}_kernel_stats.csv"))
        b = os.path.join(src, f"{dt}_bench_under_rocprof.json")
        if os.path.exists(b) and os.path.getsize(b):
            shutil.copy(b, os.path.join(dst, f"{tag}_{dt}_bench_under_rocprof.json"))
        merged = collections.defaultdict(dict)
        for p in ("fetch", "write", "sq"):
            for k, d in pmc_means(os.path.join(src, f"{dt}_pmc_{p}")).items():
                if "srgen" in k:
                    merged[k.split("(")[0]].update(d)
        json.dump(merged, open(os.path.join(dst, f"{tag}_{dt}_pmc.json"), "w"), indent=1, sort_keys=True)
        # trunk conv: launch-weighted mean of HBM bytes per launch
        fetch = write = n_f = n_w = 0.0
        busy = gui = wait = wavec = 0.0
        for k, d in merged.items():
            if is_trunk(k, dt):
                if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
                    n = d["GRBM_GUI_ACTIVE"]["n"]
                    busy += d["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] * n
                    gui += d["GRBM_GUI_ACTIVE"]["mean"] * n
                    if "SQ_WAIT_INST_ANY" in d and "SQ_WAVE_CYCLES" in d:
                        wait += d["SQ_WAIT_INST_ANY"]["mean"] * n
                        wavec += d["SQ_WAVE_CYCLES"]["mean"] * n
                if "FETCH_SIZE" in d:
                    fetch += d["FETCH_SIZE"]["mean"] * d["FETCH_SIZE"]["n"]
                    n_f += d["FETCH_SIZE"]["n"]
                if "WRITE_SIZE" in d:
                    write += d["WRITE_SIZE"]["mean"] * d["WRITE_SIZE"]["n"]
                    n_w += d["WRITE_SIZE"]["n"]
        if n_f and n_w:
            rd = 2.0 * 1024.0 * fetch / n_f        # KB -> B, gfx950 FETCH_SIZE correction x2
            wr = 1024.0 * write / n_w
            path = dt
            if dt == "f32":
                path = ("split" if any("conv3x3_f32s_kernel" in k for k in merged) else
                        "wino" if any("wino_kernel" in k for k in merged) else "direct")
            out = {"path": path, "conv3x3_trunk_hbm_bytes_per_launch": round(rd + wr),
                   "read_bytes": round(rd), "write_bytes": round(wr), "launches_sampled": int(n_f),
                   "batch": 256,
                   # matrix-pipe busy cycles summed over the chip's 1024 SIMDs / (1024 x kernel cycles);
                   # GRBM_GUI_ACTIVE is summed over the 8 XCDs, hence / 8
                   "mfma_busy_frac": round(busy / (1024.0 * gui / 8.0), 4) if gui else None,
                   "wait_inst_frac": round(wait / wavec, 4) if wavec else None,
                   "source": f"profiles/{tag}_{dt}_pmc.json: FETCH_SIZE (KB, x2 gfx950 correction for 16 B/lane "
                             "reads) + WRITE_SIZE (KB), separate rocprofv3 --pmc passes, launch-weighted mean "
                             "over the 33 trunk conv3x3 launches of one batch-256 forward"}
            json.dump(out, open(os.path.join(dst, f"traffic_{dt}.json"), "w"), indent=1)
            print(dt, out)


if __name__ == "__main__":
    main()

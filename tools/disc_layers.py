#!/usr/bin/env python3
"""Per-layer durations of the last discriminator forward in a rocprofv3 kernel trace csv."""
import csv
import glob
import sys

L = [(125, 3, 64), (63, 64, 64), (63, 64, 128), (32, 128, 128), (32, 128, 256), (16, 256, 256),
     (16, 256, 512), (8, 512, 512), (8, 512, 256), (8, 256, 1)]
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rows = [r for r in csv.DictReader(open(path)) if "disc_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = 0.0
for r, (s, ci, co) in zip(rows[-10:], L):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    fl = 2 * s * s * 9 * ci * co * batch
    nm = r["Kernel_Name"].split("::")[-1].split("(")[0]
    print(f"{nm:28s} {s:4d} {ci:4d} {co:4d} {d:8.1f} us {fl / d / 1e6:7.1f} TF")
print(f"forward total {tot:.1f} us")

#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 counter-collection / kernel-trace CSVs (any number of output directories).

    python tools/pmc_summary.py gpurun_out/prof_sq1 gpurun_out/prof_sq2 ... > profiles/rNN_pmc_summary.txt
"""
import collections
import csv
import glob
import os
import sys


def main(dirs):
    for d in dirs:
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
            meta = {}
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"][:60]
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    meta[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                               row.get("Grid_Size"), row.get("Workgroup_Size"))
            print("# %s" % path)
            for k in sorted(acc):
                print("kernel %-60s vgpr=%s agpr=%s sgpr=%s lds=%s grid=%s wg=%s" % ((k,) + meta[k]))
                for c in sorted(acc[k]):
                    s, n = acc[k][c]
                    print("    %-34s n=%-6d mean/launch=%.4g" % (c, n, s / n))
        for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
            print("# %s" % path)
            print(open(path).read())


if __name__ == "__main__":
    main(sys.argv[1:])

#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv -> short table (kernel name without its arguments, calls, average us, total ms)."""
import csv
import sys
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "")
        if name.startswith("spsp::") or "--all" in sys.argv:
            print("%-44s calls %5d  avg %10.1f us  total %9.3f ms" % (name[:44], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))

"""Prints a rocprofv3 kernel_stats.csv compactly: calls, average us, share, short kernel name.  Usage: python scripts/kstats.py <csv> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = re.sub(r"\(anonymous namespace\)::|rgx::|void ", "", r["Name"])
    name = re.sub(r"\(.*", "", name)
    print("%6s calls  %10.1f us avg  %5.1f%%  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), name[:70]))

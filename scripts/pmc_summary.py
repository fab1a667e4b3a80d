"""Per-kernel averages of one rocprofv3 --pmc pass.  Usage: python scripts/pmc_summary.py <counter_collection.csv> [name filter ...]"""
import collections
import csv
import sys


def main():
    rows = csv.DictReader(open(sys.argv[1]))
    filt = sys.argv[2:]
    agg = collections.OrderedDict()
    for r in rows:
        n = r["Kernel_Name"]
        if filt and not any(f in n for f in filt):
            continue
        key = (n.split("(")[0][:70], r.get("Grid_Size", ""), r["Counter_Name"])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for (n, grid, c), (cnt, tot) in agg.items():
        print("%-72s grid %-9s %-12s n %4d  mean %.1f" % (n, grid, c, cnt, tot / cnt))


if __name__ == "__main__":
    main()

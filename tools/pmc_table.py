#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter over dispatches."""
import collections, csv, glob, sys

def main():
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "gemm" not in k and "attn" not in k:
                    continue
                agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, dd in agg.items():
                print(k)
                for c, v in sorted(dd.items()):
                    print("   %-40s n=%-4d mean=%.5g" % (c, len(v), sum(v) / len(v)))

if __name__ == "__main__":
    main()

"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name."""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"][:60]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k].add(row["Dispatch_Id"])
        print("==", os.path.relpath(path, root))
        for k in sorted(agg, key=lambda k: -len(cnt[k])):
            if "gemm" not in k and "attention" not in k and "layernorm" not in k and "rans" not in k:
                continue
            n = len(cnt[k])
            print(f"  {k}  dispatches={n}")
            for c, v in sorted(agg[k].items()):
                print(f"      {c:32s} {v / n:16.1f} per dispatch")


if __name__ == "__main__":
    main()

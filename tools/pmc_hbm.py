#!/usr/bin/env python3
"""Mean FETCH_SIZE / WRITE_SIZE (KB per dispatch) per kernel from rocprofv3 --pmc
output directories -> JSON list (profiles/rNN_pmc_hbm.json).
usage: pmc_hbm.py DIR... > out.json"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(float))       # (counter, kernel) -> dispatch -> sum over instances
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                acc[(row["Counter_Name"], row["Kernel_Name"])][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    out = []
    for (counter, kernel), per in acc.items():
        out.append({"counter": counter, "kernel": kernel, "dispatches": len(per),
                    "mean_value_KB": sum(per.values()) / len(per)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

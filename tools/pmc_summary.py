#!/usr/bin/env python3
"""Mean PMC counter value per dispatch, per kernel, from rocprofv3 --pmc output
directories (…/*_counter_collection.csv).  usage: pmc_summary.py KERNEL_SUBSTR DIR…"""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(name):
    """'void (anonymous namespace)::sre_k_scan<2, 4>(args...)' -> 'sre_k_scan<2, 4>'"""
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()


def main():
    want = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[2:]:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            per = defaultdict(float)
            meta = {}
            for row in csv.DictReader(open(f)):
                if want not in row["Kernel_Name"]:
                    continue
                key = (short(row["Kernel_Name"]), row["Dispatch_Id"], row["Counter_Name"])
                per[key] += float(row["Counter_Value"])
                meta[short(row["Kernel_Name"])] = {
                    "vgpr": int(row["VGPR_Count"]), "lds": int(row["LDS_Block_Size"]),
                    "scratch": int(row["Scratch_Size"]), "grid": int(row["Grid_Size"])}
            for (k, _, c), v in per.items():
                acc[k][c].append(v)
            for k, m in meta.items():
                acc[k]["_meta"] = m
    out = {}
    for k, cs in acc.items():
        out[k] = {c: (v if c == "_meta" else sum(v) / len(v)) for c, v in cs.items()}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

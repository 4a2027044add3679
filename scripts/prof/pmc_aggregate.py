"""Aggregate rocprofv3 --pmc counter_collection CSVs into per-kernel means.

usage: pmc_aggregate.py OUT.json DIR [DIR ...]
Each DIR is the -d directory of one `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv` pass
(one counter per pass: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "PMC slots").
Raw counter units are KiB; no correction is applied here (see DESIGN.md section 7 for the calibration).
"""
import csv
import glob
import json
import os
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0]
                    key = (name, row["Counter_Name"])
                    # rocprofv3 emits one row per dispatch (and per dimension instance); sum per dispatch id
                    acc.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
                    acc[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
    res = {}
    for (name, ctr), per in acc.items():
        vals = list(per.values())
        res.setdefault(name, {})[ctr] = {"mean": sum(vals) / len(vals), "n": len(vals)}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for name, c in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {"mean": 0})["mean"])[:12]:
        print(name[:60], {k: round(v["mean"], 1) for k, v in c.items()})


if __name__ == "__main__":
    main()

"""Per-kernel averages of rocprofv3 --pmc passes.  Usage: python tools/pmc_summary.py DIR [DIR...] [--match SUBSTR]
Each counter value is summed over its hardware instances (XCDs / channels / SEs) within a dispatch and averaged
over the dispatches of a kernel."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> sum
    for d in args:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if match and match not in k:
                    continue
                per[k][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in per.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

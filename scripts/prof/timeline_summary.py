"""kernel_trace.csv of rocprofv3 -> per LM iteration: every launch in order with its duration and the gap to its predecessor.
An iteration = the launches from one cam_pass_kernel<*, 0> (linearisation) to the next.  Prints / stores the mean over the
steady iterations: per kernel name (in launch order) duration and gap, and the totals (kernel time, gap time, span)."""
import csv
import json
import sys


def short(n):
    n = n.replace("void ", "").replace("vgg::", "")
    return n.split("(")[0]


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("cam_pass_kernel") and ", 0>" in r[2]]
    its = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    if not its:
        print("no iterations found")
        return
    # steady iterations: the most common launch sequence
    from collections import Counter
    sig = Counter(tuple(r[2] for r in it) for it in its).most_common(1)[0][0]
    its = [it for it in its if tuple(r[2] for r in it) == sig][2:]
    n = len(its)
    agg = []
    for k, name in enumerate(sig):
        dur = sum(it[k][1] - it[k][0] for it in its) / n / 1e3
        gap = sum((it[k][0] - it[k - 1][1]) if k else 0 for it in its) / n / 1e3
        agg.append(dict(kernel=name[:70], us=round(dur, 2), gap_before_us=round(gap, 2)))
    span = sum(it[-1][1] - it[0][0] for it in its) / n / 1e3
    tot_k = sum(a["us"] for a in agg)
    tot_g = sum(a["gap_before_us"] for a in agg)
    out = dict(iterations=n, launches_per_iteration=len(sig), span_us=round(span, 1), kernel_us=round(tot_k, 1), gaps_us=round(tot_g, 1),
               small_kernels_us=round(sum(a["us"] for a in agg if a["us"] < 40), 1), launches=agg)
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for a in agg:
        print(f"{a['us']:9.2f} us  gap {a['gap_before_us']:6.2f}  {a['kernel']}")
    print({k: v for k, v in out.items() if k != "launches"})


if __name__ == "__main__":
    main()

"""Gaps between consecutive kernels of the decode step, from a rocprofv3 --kernel-trace CSV (start / end timestamps per
dispatch): for every kernel kind, the average duration and the average idle time between the end of its predecessor and
its own start, over the single-sequence decode steps of the trace (the run of dispatches between two sample_advance
kernels that has exactly the step's launch count).
usage: python tools/gap_analysis.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def short(name):
    for key, tag in (("true, 3", "gate_up"), ("false, 1,", "qkv"), ("false, 6,", "attn_out+combine"), ("false, 2,", "attn_out"),
                     ("false, 4,", "down"), ("false, 5,", "lm_head")):
        if "gemv_kernel" in name and key in name:
            return tag
    for key in ("attn_decode_split", "sample_advance", "argmax_slices", "dequant", "rope"):
        if key in name:
            return key
    return name.split("(")[0][-40:]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # steps: segments ending in sample_advance
    steps, cur = [], []
    for s, e, n in rows:
        cur.append((s, e, short(n)))
        if "sample_advance" in n:
            steps.append(cur)
            cur = []
    counts = defaultdict(int)
    for st in steps:
        counts[len(st)] += 1
    nl = max(counts, key=lambda k: counts[k])
    steps = [st for st in steps if len(st) == nl]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    tot_busy = tot_gap = 0.0
    for st in steps[2:]:
        for i in range(1, len(st)):
            k = st[i][2]
            dur[k] += st[i][1] - st[i][0]
            gap[k] += st[i][0] - st[i - 1][1]
            cnt[k] += 1
            tot_busy += st[i][1] - st[i][0]
            tot_gap += st[i][0] - st[i - 1][1]
    n = max(1, len(steps) - 2)
    print(f"{len(steps)} decode steps of {nl} launches; per step: kernels {tot_busy / n / 1e3:.1f} us + gaps {tot_gap / n / 1e3:.1f} us")
    print(f"{'kernel':28s} {'launches/step':>13s} {'avg us':>8s} {'gap before, us':>15s}")
    for k in sorted(cnt, key=lambda k: -dur[k]):
        print(f"{k:28s} {cnt[k] / n:13.1f} {dur[k] / cnt[k] / 1e3:8.2f} {gap[k] / cnt[k] / 1e3:15.2f}")


if __name__ == "__main__":
    main(sys.argv[1])

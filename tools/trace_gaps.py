"""Where a batch's time goes between its kernels: from a rocprofv3 kernel trace of `bench.py --no-overlap`, the chain of
one step (from one gather's end to the next one's end), averaged over the last steps: per launch its duration and the
idle time on the GPU before it.

    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [steps to average, default 10]
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    f = sorted(glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))
            if "ggms" in r["Kernel_Name"]]
    rows.sort()
    # a step ends with the label gather (PlainRows, 8-byte rows) that follows the feature gather
    ends = [i for i, r in enumerate(rows) if "k_gather_rows<8, ggms::PlainRows" in r[2]]
    steps = [rows[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])][-last:]
    n = min(len(s) for s in steps)
    steps = [s for s in steps if len(s) == n]
    print(f"{len(steps)} steps of {n} launches")
    tot_busy = tot_gap = 0.0
    for j in range(n):
        dur = sum(s[j][1] - s[j][0] for s in steps) / len(steps) / 1e3
        gap = sum(s[j][0] - (s[j - 1][1] if j else s[j][0]) for s in steps) / len(steps) / 1e3
        tot_busy += dur
        tot_gap += gap
        print(f"{j:3d} gap {gap:7.2f} us  run {dur:8.2f} us  {steps[0][j][2][:110]}")
    print(f"busy {tot_busy:.1f} us, idle between launches {tot_gap:.1f} us per step")


if __name__ == "__main__":
    main()

"""Per-kernel statistics of the TIMED region of a `rocprofv3 --kernel-trace` run of bench.py.

MIOpen benchmarks candidate solvers during the warm-up steps (torch.backends.cudnn.benchmark with the
shipped find-db), so the raw --stats table of the whole process is dominated by warm-up kernels.
This tool keeps the last K steps only (a step is delimited by consecutive launches of the
once-per-step cost-volume kernel at level 2) and writes the same columns as rocprofv3's stats csv.

    python tools/summarize_trace.py <kernel_trace.csv> <out.csv> [--steps 10]
"""
import collections
import csv
import sys

MARK = 'corr_fwd_ring_kernel<4, 2, 9>'     # launched exactly once per step (level 2)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 10
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [int(r['Start_Timestamp']) for r in rows if MARK in r['Kernel_Name']]
    if len(marks) < steps + 1:
        raise SystemExit('only %d marker launches in the trace' % len(marks))
    t0 = marks[-steps - 1]           # from the marker of the step before the first kept one ...
    t1 = marks[-1]                   # ... to the marker of the last step: exactly `steps` step periods
    agg = collections.defaultdict(list)
    for r in rows:
        s = int(r['Start_Timestamp'])
        if t0 <= s < t1:
            agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - s)
    total = sum(sum(v) for v in agg.values())
    out = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for name, v in out:
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3), min(v), max(v)])
    print('%d steps, wall %.3f ms/step, GPU-busy %.3f ms/step, %d distinct kernels' %
          (steps, (t1 - t0) / 1e6 / steps, total / 1e6 / steps, len(out)))
    for name, v in out[:8]:
        print('  %-80s calls/step %6.1f  %7.3f ms/step  avg %8.1f us' % (name[:80], len(v) / steps, sum(v) / 1e6 / steps, sum(v) / len(v) / 1e3))
    for name, v in out:
        if MARK in name:
            print('  dominant hand-written kernel: %s avg %.2f us over %d launches' % (name[:60], sum(v) / len(v) / 1e3, len(v)))


if __name__ == '__main__':
    main()

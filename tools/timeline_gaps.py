"""Dev tool: from a rocprofv3 --kernel-trace csv of a few C2 steps, how much of the step the device spends with NO kernel running and how
much with one / two running (the side stream), plus the distribution of the idle gaps.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 tools/ab_step.py base      (ROUNDS=1 PER=5)
    python tools/timeline_gaps.py /tmp/tr/run_kernel_trace.csv"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
# last 40 % of the trace = steady-state steps
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * 6 // 10
rows = [r for r in rows if r[0] >= t_lo]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = collections.Counter(); depth = 0; last = ev[0][0]; gaps = []
for t, d in ev:
    busy[min(depth, 2)] += t - last
    if depth == 0 and t > last:
        gaps.append(t - last)
    depth += d; last = t
tot = sum(busy.values())
print("window %.1f ms, %d kernels" % (tot / 1e6, len(rows)))
for k in (0, 1, 2):
    print("  %s kernels in flight: %5.1f %% of the time" % ({0: "no", 1: "one", 2: "two or more"}[k], 100.0 * busy[k] / tot))
gaps.sort()
if gaps:
    n = len(gaps)
    print("  idle gaps: %d, total %.2f ms, median %.1f us, p90 %.1f us, max %.1f us" % (n, sum(gaps) / 1e6, gaps[n // 2] / 1e3, gaps[n * 9 // 10] / 1e3, gaps[-1] / 1e3))
    big = [g for g in gaps if g > 20000]
    print("  gaps > 20 us: %d, total %.2f ms" % (len(big), sum(big) / 1e6))
if len(sys.argv) > 2:          # list the large gaps with the kernels either side
    ends = sorted(rows, key=lambda r: r[1])
    import bisect
    starts = [r[0] for r in rows]
    out = []
    cur_end = rows[0][1]; prev = rows[0]
    for r in rows[1:]:
        if r[0] > cur_end + 20000:
            out.append((r[0] - cur_end, prev[2], r[2]))
        if r[1] > cur_end:
            cur_end = r[1]; prev = r
    cnt = collections.Counter((a, b) for _, a, b in out); tot = collections.Counter()
    for g, a, b in out: tot[(a, b)] += g
    for (a, b), t in tot.most_common(12):
        print("  %3d gaps, %.2f ms total: after %-55s before %s" % (cnt[(a, b)], t / 1e6, a, b))

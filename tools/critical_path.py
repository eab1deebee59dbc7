"""Dev tool: where the wall time of a train step goes, from a rocprofv3 --kernel-trace csv of a few steps.
The step's chain runs on ONE stream (the encoder's weight gradients on a second one): per queue, the busy time by kernel family, the idle
time between consecutive kernels of that queue (launch boundaries + host stalls), and how much of the side queue runs under the main one.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 tools/ab_step.py base      (ROUNDS=1 PER=6)
    python tools/critical_path.py /tmp/tr/run_kernel_trace.csv [steps in the window]"""
import collections
import csv
import re
import sys


def family(name):
    n = name
    m = re.search(r"gemm_glds_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\d+), (\w+)", n)
    if m:
        am, bm = int(m.group(3)), int(m.group(4))
        kind = {(2, 0): "conv_fwd", (3, 3): "conv_dgrad", (1, 2): "conv_wgrad", (0, 0): "nt", (0, 1): "nn", (1, 1): "tn"}.get((am, bm), "?")
        return "glds_%s_%sx%s_%s" % (kind, m.group(1), m.group(2), "bf" if "bf16" in m.group(5) else "f32")
    for key in ("gemm_bf16_kernel", "gemm_f32_kernel", "wgrad3x3", "splitk_reduce", "bn_bwd_apply", "bn_apply", "bn_colstats", "bn_tile", "bn_relu_maxpool",
                "attention_scores", "attention_context", "attention_bwd_dalpha", "attention_bwd_tanh", "lstm_cell_fwd", "lstm_cell_bwd", "ce_rows", "ce_grad",
                "optimizer_step", "embedding", "colsum", "cast", "gather_rows", "normalize", "resize", "avgpool", "fill", "copy"):
        if key in n:
            return key
    return n.split("(")[0].split("<")[0][-40:]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "0"))))
rows.sort()
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * 6 // 10          # last 40 % of the trace = steady state
rows = [r for r in rows if r[0] >= t_lo]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
window = rows[-1][1] - rows[0][0]
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
qs = sorted(byq, key=lambda q: -sum(e - s for s, e, _, _ in byq[q]))
if steps is None:          # count optimizer launches
    steps = max(1, sum(1 for r in rows if "optimizer_step" in r[2]))
print("window %.2f ms, %d kernels, ~%g steps -> %.2f ms per step" % (window / 1e6, len(rows), steps, window / 1e6 / steps))
main = qs[0]
for q in qs[:3]:
    ev = byq[q]
    busy = sum(e - s for s, e, _, _ in ev)
    gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1) if ev[i + 1][0] > ev[i][1]]
    small = [g for g in gaps if g <= 20000]
    print("queue %s: %d kernels, busy %.2f ms/step, idle between kernels %.2f ms/step (gaps <= 20 us: %d, %.2f ms/step, median %.2f us)" % (
        q, len(ev), busy / 1e6 / steps, sum(gaps) / 1e6 / steps, len(small), sum(small) / 1e6 / steps, (sorted(small)[len(small) // 2] / 1e3 if small else 0)))
    fam = collections.Counter(); cnt = collections.Counter()
    for s, e, n, _ in ev:
        f = family(n); fam[f] += e - s; cnt[f] += 1
    for f, t in fam.most_common(28):
        print("    %-44s %6.0f launches/step  %7.3f ms/step  %6.1f us each" % (f, cnt[f] / steps, t / 1e6 / steps, t / 1e3 / cnt[f]))
# overlap of the side queues with the main one
if len(qs) > 1:
    mev = [(s, e) for s, e, _, _ in byq[main]]
    import bisect
    starts = [s for s, _ in mev]
    for q in qs[1:3]:
        under = 0
        for s, e, _, _ in byq[q]:
            i = max(0, bisect.bisect_right(starts, s) - 1)
            while i < len(mev) and mev[i][0] < e:
                under += max(0, min(e, mev[i][1]) - max(s, mev[i][0])); i += 1
        tot = sum(e - s for s, e, _, _ in byq[q])
        print("queue %s: %.2f of its %.2f ms/step run while a main-queue kernel runs" % (q, under / 1e6 / steps, tot / 1e6 / steps))

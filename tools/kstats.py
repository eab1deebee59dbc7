"""Dev tool: per-kernel ms/step from a rocprofv3 --kernel-trace results db (steps inferred from bn_apply launches: 53/step)."""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); cur = c.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute("select s.kernel_name, d.end-d.start from %s d join %s s on d.kernel_id=s.id" % (kd, ks)).fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for n, t in rows:
    agg[n][0] += 1; agg[n][1] += t
steps = [cn / 53 for n, (cn, t) in agg.items() if "bn_apply_kernel" in n][0]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
tot = 0
for n, (cn, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if re.search(pat, n):
        tot += t / steps / 1e6
        print("%-100s %6.1f/step %7.3f ms/step avg %7.1f us" % (re.sub(r"\(.*", "", n)[:100], cn / steps, t / steps / 1e6, t / cn / 1e3))
print("sum %.3f ms/step over %.0f steps; all kernels %.3f ms/step" % (tot, steps, sum(t for _, t in agg.values()) / steps / 1e6))

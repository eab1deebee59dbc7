"""Dev tool: per-launch GEMM durations inside C2 train steps.  Run under
   SAT_LOG_GEMM=1 rocprofv3 --kernel-trace -d <dir> -o run -- python tools/gemm_trace.py run   (writes the launch log)
then  python tools/gemm_trace.py report <dir>/run_results.db <log>  matches the k-th logged launch with the k-th
gemm_bf16 / gemm_glds dispatch and prints time per problem shape against its HBM / MFMA floors."""
import collections, os, re, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import bench
    import sat_amd  # noqa
    from sat_amd import model as M
    hp, T, B, R = bench.hparams("c2")
    torch.manual_seed(42)
    model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
    model.__dict__["_sat_global_step"] = 2
    opt = model.configure_optimizers()
    img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False)
    img, caps = img.cuda(), caps.cuda()
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); opt.step()
    torch.cuda.synchronize()


def report(db, log):
    shapes = [l.strip() for l in open(log, errors="replace") if l.startswith("GEMMLOG")]
    c = sqlite3.connect(db); cur = c.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute("select s.kernel_name, d.end-d.start from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
    rows = [r for r in rows if re.search(r"gemm_bf16_kernel|gemm_glds_kernel", r[0])]
    assert len(rows) == len(shapes), (len(rows), len(shapes))
    per = len(shapes) // 6
    agg = collections.OrderedDict()
    for (name, dur), sh in list(zip(rows, shapes))[-2 * per:]:          # last two steps
        kind = "glds" if "glds" in name else "regs"
        tile = re.search(r"ILi(\d+)ELi(\d+)E", name).group(1)
        key = sh[8:] + " " + kind + tile
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += dur
    out = []
    for key, (n, tot) in agg.items():
        f = dict(kv.split("=") for kv in key.split()[:8])
        Mm, N, K, acc = int(f["M"]), int(f["N"]), int(f["K"]), int(f["acc"])
        ta, tb, tc = [2 if ch == "1" else 4 for ch in f["types"]]
        by = Mm * K * ta + N * K * tb + Mm * N * tc * (2 if acc else 1)
        fl = 2.0 * Mm * N * K
        us = tot / n / 1e3
        out.append((tot / 2e6, key, n // 2, us, by / 4.5e6, fl / 2.5e9))
    out.sort(reverse=True)
    print("%-86s %3s %8s %7s %7s %6s" % ("shape", "n", "us", "hbm_us", "mfma_us", "ms/step"))
    for ms, key, n, us, hb, mf in out:
        print("%-86s %3d %8.1f %7.1f %7.1f %6.3f" % (key, n, us, hb, mf, ms))
    print("total GEMM ms/step: %.2f ; floor %.2f" % (sum(o[0] for o in out), sum(max(o[4], o[5]) * o[2] for o in out) / 1e3))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2], sys.argv[3])

"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, MI355X_MICROARCH.md HBM section) into
per-kernel HBM bytes per launch.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 64 B per 128-B
request for wide coalesced streaming reads, so the read side is doubled.

    python tools/pmc_aggregate.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_bf16_bench_c2_pmc_hbm.json [steps_fetch steps_write]

``steps_*`` = train steps the profiled command ran in each pass (the bench line's ``train_steps_in_process``: warm-up + profile + timed
steps; the warm-up is time-based, so the two passes differ).  Written as a ``__meta__`` row so that
bench.py can state HBM bytes per step.
"""
import collections, csv, glob, json, re, sys

MODE = {("0", "0"): "nt", ("0", "1"): "nn", ("1", "1"): "tn", ("2", "0"): "conv_fwd", ("3", "3"): "conv_dgrad", ("1", "2"): "conv_wgrad"}


def family(name):
    m = re.search(r"gemm_glds_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d)ELi(\d)E", name) or re.search(r"gemm_glds_kernel<(\d+), (\d+), \d+, \d+, (\d), (\d),", name)
    if m:      # direct-to-LDS kernel: bf16 operands in HBM
        return "gemm_glds_%s_%sx%s" % (MODE.get((m.group(3), m.group(4)), "?"), m.group(1), m.group(2))
    m = re.search(r"gemm_bf16_kernelILi(\d+)ELi(\d+)ELi\d+ELi(\d)ELi(\d)E(DF16b|f)", name) or \
        re.search(r"gemm_bf16_kernel<(\d+), (\d+), \d+, (\d), (\d), (float|__bf16)", name)
    if m:
        return "gemm_bf16_%s_%sx%s_%s" % (MODE.get((m.group(3), m.group(4)), "?"), m.group(1), m.group(2), "b" if m.group(5) in ("DF16b", "__bf16") else "f")
    m = re.search(r"gemm_f32_kernel<(\d+), (\d+), (\d), (\d)>", name)
    if m:
        return "gemm_f32_%s_%sx%s" % (MODE.get((m.group(3), m.group(4)), "?"), m.group(1), m.group(2))
    if "wgrad3x3_kernel" in name: return "gemm_wgrad3x3"
    if "bn_bwd_apply_kernel" in name: return "bn_bwd_apply_kernel"
    if "bn_apply_kernel" in name: return "bn_apply_kernel"
    m = re.search(r"bn_colstats_kernelILi(\d)E", name) or re.search(r"bn_colstats_kernel<(\d),", name)
    if m: return "bn_colstats_kernel<%s>" % m.group(1)
    return name.split("(")[0][:80]


def agg(d):
    out = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open((glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0])):
        e = out[family(r["Kernel_Name"])]
        e[0] += 1; e[1] += float(r["Counter_Value"]); e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return out


if __name__ == "__main__":
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    rows = []
    for k, (n, fs, t) in f.items():
        ws = w.get(k, [0, 0.0, 0.0])[1]
        wn = w.get(k, [0, 0.0, 0.0])[0] or n
        rows.append(dict(family=k, launches=n, read_bytes_per_launch=2 * fs * 1024 / n, write_bytes_per_launch=ws * 1024 / wn,
                         hbm_bytes_per_launch=2 * fs * 1024 / n + ws * 1024 / wn, avg_us_under_pmc=t / n / 1e3))
    rows.sort(key=lambda r: -r["avg_us_under_pmc"] * r["launches"])
    if len(sys.argv) > 5:
        sf, sw = int(sys.argv[4]), int(sys.argv[5])
        rd = sum(r["read_bytes_per_launch"] * r["launches"] for r in rows) / sf
        wr = sum(r["write_bytes_per_launch"] * w.get(r["family"], [r["launches"]])[0] for r in rows) / sw
        per_step = {r["family"]: (r["read_bytes_per_launch"] * r["launches"] / sf + r["write_bytes_per_launch"] * w.get(r["family"], [r["launches"]])[0] / sw) / 1e9 for r in rows}
        rows.append(dict(family="__meta__", steps=sf, steps_write=sw, launches=0, hbm_bytes_per_launch=0.0, read_gb_per_step=rd / 1e9, write_gb_per_step=wr / 1e9,
                         hbm_gb_per_step=(rd + wr) / 1e9, gb_per_step_by_family={k: round(v, 3) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])[:24]}))
        print("HBM bytes per step: %.2f GB read + %.2f GB written" % (rd / 1e9, wr / 1e9))
    json.dump(rows, open(sys.argv[3], "w"), indent=1)
    for r in rows[:10]:
        if r["family"] == "__meta__":
            continue
        print("%-40s n=%4d  %8.1f MB/launch  %7.1f us" % (r["family"], r["launches"], r["hbm_bytes_per_launch"] / 1e6, r["avg_us_under_pmc"]))

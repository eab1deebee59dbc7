#!/usr/bin/env python
"""Headline benchmark: captions/sec of one SAT training step (encoder -> attention-LSTM decoder ->
losses -> backward -> gradient all-reduce -> Adam) on synthetic data, BASELINE.json configs[1]:
resnet50, encoder_size=7 (L=49), encoder_dim=512, vocab=6400, T=22, batch 128 images per GPU, R=5 captions
per image (640 caption sequences per GPU-step).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.  `roofline` is measured with HIP events inside the library (sat_profile_*)
on instrumented steps run right after the timed region; `cpu_baseline` times the CPU oracle (a port of the
reference's arithmetic, oracle/sat_oracle.py) on a bounded sample of the same model on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

CONFIGS = {
    # name: (arch, encoder_size, encoder_dim, vocab, T, batch per GPU)
    "c2": ("resnet50", 7, 512, 6400, 22, 128),
    "c1": ("resnet18", 7, 256, 6400, 22, 8),
    "c3": ("resnet101", 14, 512, 6400, 22, 32),
    "c4": ("wide_resnet101_2", 14, 1024, 10000, 32, 64),
}
PEAK = {"f32": 157.3, "bf16": 2500.0}     # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def hparams(cfg, R=5):
    arch, es, D, V, T, B = CONFIGS[cfg]
    stoi = {"<PAD>": 0, "<UNK>": V - 3, "<START>": V - 2, "<END>": V - 1}
    return dict(encoder_arch=arch, pretrained=False, input_size=256, encoder_dim=D, encoder_size=es,
                mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], embed_dim=256, embed_norm=None, attention_dim=128,
                decoder_dim=512, decoder_layers=1, dropout=0.0, embedding_dropout=0.0, label_smoothing=0.0, weight_tying=False,
                deep_output=True, att_gamma=1.0, vocab_size=V, vocab_stoi=stoi, vocab_itos={v: k for k, v in stoi.items()},
                pretrained_embedding=None, decoder_tf="always", decoder_tf_min=0.5, epochs=10, encoder_finetune_after=1,
                opt="adam", decoder_lr=1e-3, embedding_lr=1e-2, encoder_lr=1e-5, weight_decay=0.0, adam_b1=0.9, adam_b2=0.999,
                momentum=0.9, nesterov=False, scheduler=None, lr_warmup_steps=0), T, B, R


def synthetic_batch(B, R, T, V, seed, ragged):
    """img ~ U[0,1); captions [START] + U{1..V-4} + [END] + PAD; lengths all T-1 (headline) or U{8..T-1}."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, 256, 256, generator=g)
    lengths = torch.randint(8, T, (B, R), generator=g) if ragged else torch.full((B, R), T - 1, dtype=torch.int64)
    caps = torch.zeros(B, R, T, dtype=torch.int64)
    toks = torch.randint(1, V - 3, (B, R, T), generator=g)
    pos = torch.arange(T)[None, None, :]
    caps = torch.where(pos < lengths[..., None], toks, caps)
    caps[..., 0] = V - 2
    caps.scatter_(2, lengths[..., None], V - 1)
    return img, caps, lengths


def cpu_baseline(cfg, seconds_budget=30.0):
    """The CPU oracle (kind "port") on B=8 images of the same model: 1 warm-up + timed steps within the budget."""
    from types import SimpleNamespace
    from oracle import sat_oracle as O
    hp, T, _, R = hparams(cfg)
    # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("SAT_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    ns = SimpleNamespace(**hp)
    model = O.OracleSAT(ns, None, seed=42)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    B = 8
    img, caps, lengths = synthetic_batch(B, R, T, hp["vocab_size"], 1234, False)

    def step():
        opt.zero_grad()
        loss, _ = model.step_loss(img, caps, lengths, 1.0)
        loss.backward()
        opt.step()

    step()
    log("cpu baseline warm-up step done")
    t0 = time.time(); n = 0
    while True:
        step(); n += 1
        log("cpu baseline step %d" % n)
        if time.time() - t0 > seconds_budget * 0.5 or n >= 16:      # ~10-15 s of CPU work
            break
    dt = (time.time() - t0) / n
    return {"value": round(B * R / dt, 2), "unit": "captions/s", "cores": cores, "kind": "port",
            "sample": "%d full train steps (fwd+loss+bwd+Adam, fp32) of the same model at batch %d images x %d captions after 1 warm-up; "
                      "oracle/sat_oracle.py on torch-CPU, %d threads" % (n, B, R, cores),
            "s_per_step": round(dt, 3)}


_T0 = time.time()


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


MIN_WARM_S = 1.5
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# in-library profile name -> kernel family of profiles/*_pmc_hbm.json (tools/pmc_aggregate.py)
PMC_FAMILY = {"bn_apply_bwd": "bn_bwd_apply_kernel", "bn_apply_fwd": "bn_apply_kernel", "bn_stats_bwd": "bn_colstats_kernel<1>",
              "bn_stats_fwd": "bn_colstats_kernel<0>"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--ragged", action="store_true", help="lengths ~ U{8..T-1} instead of all T-1")
    ap.add_argument("--decoder-tf", default="always", choices=["always", "none"],
                    help="teacher forcing: always (epsilon = 1, the headline) or none (train.py:63 default: argmax feedback after step 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=None, help="override images per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="bf16: bf16 MFMA + bf16 activations (BASELINE configs[1]); fp32: exact fp32 MFMA parity mode")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SAT hot path has no CPU fallback")
    # SAT_BENCH_REHEARSAL=1 (dev): every rank on GPU 0 with the gloo backend -- rehearses the multi-process flow (buckets, hooks,
    # timing protocol) on a one-GPU box; the numbers it prints are meaningless
    rehearsal = os.environ.get("SAT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    import sat_amd  # noqa: F401
    from sat_amd import _lib, model as M
    from sat_amd.dist import GradSync, broadcast_parameters

    hp, T, B, R = hparams(args.config)
    if args.batch:
        B = args.batch
    if args.decoder_tf == "none":
        hp["decoder_tf"] = None
    torch.manual_seed(42)
    model = M.SAT(**hp).to(dev).train()
    model.set_precision(args.precision)
    broadcast_parameters(model)
    model.__dict__["_sat_global_step"] = 2          # past encoder_finetune_after: the encoder trains (and is in the optimizer)
    opt = model.configure_optimizers()
    sync = GradSync(model)
    img, caps, lengths = synthetic_batch(B, R, T, hp["vocab_size"], 1234 + rank, args.ragged)
    img, caps = img.to(dev), caps.to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model.training_step((img, caps, lengths), 0)
        out["loss"].backward()
        sync.finish()
        opt.step()
        return out

    log("model built (%s, %d images/GPU); warm-up" % (args.config, B))
    # W untimed steps, and in any case MIN_WARM_S seconds of them: a GPU coming out of idle (fresh box / fresh process) needs
    # ~1 s of load before its clocks settle -- with 3 warm-up steps the first 10 timed steps ran 20 % slow (37-42 vs 32 ms)
    t_warm = time.perf_counter(); warm_done = 0
    while True:
        more = warm_done < args.warmup or time.perf_counter() - t_warm < MIN_WARM_S
        if world > 1:        # every rank must run the same number of steps (each step holds collectives): agree on the flag
            flag = torch.tensor([1 if more else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = bool(flag.item())
        if not more:
            break
        out = step()
        torch.cuda.synchronize(); warm_done += 1
        if warm_done <= 3 or warm_done % 10 == 0:
            log("warm-up step %d done" % warm_done)
    # ---- which kernel family dominates: HIP events on the launch stream around every instrumented launch, two untimed steps
    prof_steps = 2
    _lib.profile_start()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    entries = sorted(_lib.profile_stop(), key=lambda e: -e["total_ms"])
    dom_name = entries[0]["name"] if entries else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- the timed region; the dominant family alone stays instrumented inside it (an event pair per launch of that family)
    if os.environ.get("SAT_BENCH_NO_INREGION") == "1":       # dev: A/B the cost of the in-region event pairs
        dom_name = None
    _lib.profile_start(only=dom_name) if dom_name else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timed = {e["name"]: e for e in _lib.profile_stop()} if dom_name else {}
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(out["loss"].item())
    log("timed region: %.1f ms/step" % (dt / args.steps * 1e3))

    if rank == 0:
        ms = dt / args.steps * 1e3
        caps_per_s = world * B * R / (dt / args.steps)
        dom = timed.get(dom_name) or (entries[0] if entries else None)          # the dominant family as measured INSIDE the timed region
        dom_steps = args.steps if dom_name in timed else prof_steps
        roof = None
        if dom:
            hbm_bound = dom["flops"] == 0           # streaming kernels (BatchNorm passes) are instrumented with bytes only
            tf = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
            gbs = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9
            peak = PEAK["bf16"] if ("bf16" in dom["name"] or "glds" in dom["name"]) else PEAK["f32"]
            traffic = None      # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_aggregate.py), same workload only
            pmc = os.path.join(ROOT, "profiles", "r01_bf16_bench_c2_pmc_hbm.json")
            if args.config == "c2" and args.precision == "bf16" and not args.batch and not args.ragged and os.path.exists(pmc):
                for row in json.load(open(pmc)):
                    if row["family"] == PMC_FAMILY.get(dom["name"], dom["name"]):
                        traffic = round(row["hbm_bytes_per_launch"])
            roof = {"bound": "hbm" if hbm_bound else "mfma", "kernel": dom["name"],
                    "achieved": round(gbs if hbm_bound else tf, 2), "peak": PEAK_HBM_GBS if hbm_bound else peak,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round((gbs / PEAK_HBM_GBS) if hbm_bound else (tf / peak), 4), "traffic": traffic,
                    "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2), "launches_per_step": dom["launches"] // dom_steps,
                    "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"], "flops_per_launch": dom["flops"] / dom["launches"],
                    "measured_on": ("HIP events on the launch stream around every launch of this family inside the timed region (%d steps); "
                                    "`top`: every instrumented family over %d untimed steps before it" % (dom_steps, prof_steps)),
                    "top": [{"kernel": e["name"], "ms_per_step": round(e["total_ms"] / prof_steps, 3),
                             "tflops": round(e["flops"] / (e["total_ms"] * 1e-3) / 1e12, 2) if e["total_ms"] > 0 else None,
                             "gbytes_per_s": round(e["bytes"] / (e["total_ms"] * 1e-3) / 1e9, 1) if e["total_ms"] > 0 else None}
                            for e in entries[:8]]}
        line = {"metric": "captions/sec (train step) at B=128, 256px, seq_len=22", "value": round(caps_per_s, 1), "unit": "captions/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": warm_done, "ms_per_step": round(ms, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
                "config": {"workload": "%s: %s encoder_size=%s encoder_dim=%d vocab=%d T=%d, %d images/GPU x R=%d captions, "
                                       "trainable encoder, Adam, %s lengths, decoder_tf=%s" % (args.config.upper(), CONFIGS[args.config][0],
                                                                               CONFIGS[args.config][1], CONFIGS[args.config][2],
                                                                               hp["vocab_size"], T, B, R, "ragged" if args.ragged else "full", args.decoder_tf),
                           "global_batch_images": world * B, "captions_per_step": world * B * R, "parallelism": "dp%d" % world,
                           "images_per_s": round(world * B / (dt / args.steps), 1), "final_loss": round(loss_val, 4)},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            log("CPU baseline (oracle) ...")
            line["cpu_baseline"] = cpu_baseline(args.config)
            log("CPU baseline done")
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: captions/sec of one SAT training step (encoder -> attention-LSTM decoder ->
losses -> backward -> gradient all-reduce -> Adam) on synthetic data, BASELINE.json configs[1]:
resnet50, encoder_size=7 (L=49), encoder_dim=512, vocab=6400, T=22, batch 128 images per GPU, R=5 captions
per image (640 caption sequences per GPU-step).

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ..., or plainly as above --
     the parent then starts the N ranks itself, before anything touches the GPU, and relays rank 0's JSON line)

Prints ONE JSON line on rank 0.  `roofline` is measured with HIP events inside the library (sat_profile_*): two
untimed instrumented steps rank the kernel families (`families` / `top`), then the dominant family alone stays
instrumented inside the timed region; `cpu_baseline` times the CPU oracle (a port of the reference's arithmetic,
oracle/sat_oracle.py) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (arch, encoder_size, encoder_dim, vocab, T, batch per GPU)
    "c2": ("resnet50", 7, 512, 6400, 22, 128),
    "c1": ("resnet18", 7, 256, 6400, 22, 8),
    "c3": ("resnet101", 14, 512, 6400, 22, 32),
    "c4": ("wide_resnet101_2", 14, 1024, 10000, 32, 64),
    # not a BASELINE configuration: the reference CLI's own defaults (train.py:43-63: shufflenet_v2_x0_5 at 224 px, no projection / resize, plain
    # output layer, decoder_tf None) at the batch its encoder table was measured with (dev/encoder_summaries.txt:28: 32 images) - a sub-line only
    "cli": ("shufflenet_v2_x0_5", None, None, 6400, 22, 32, 224),
}
#: images per step of the whole job as BASELINE.json states them (--strong splits these over the ranks)
GLOBAL_BATCH = {"c1": 8, "c2": 128, "c3": 256, "c4": 512}
PEAK = {"f32": 157.3, "bf16": 2500.0}     # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MIN_WARM_S = 1.5
# in-library profile name -> kernel family of profiles/*_pmc_hbm.json (tools/pmc_aggregate.py)
PMC_FAMILY = {"bn_apply_bwd": "bn_bwd_apply_kernel", "bn_apply_fwd": "bn_apply_kernel", "bn_stats_bwd": "bn_colstats_kernel<1>",
              "bn_stats_fwd": "bn_colstats_kernel<0>"}
PMC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r03_bf16_bench_c2_pmc_hbm.json", "r02_bf16_bench_c2_pmc_hbm.json", "r01_bf16_bench_c2_pmc_hbm.json")]


def hparams(cfg, R=5):
    arch, es, D, V, T, B = CONFIGS[cfg][:6]
    stoi = {"<PAD>": 0, "<UNK>": V - 3, "<START>": V - 2, "<END>": V - 1}
    return dict(encoder_arch=arch, pretrained=False, input_size=CONFIGS[cfg][6] if len(CONFIGS[cfg]) > 6 else 256, encoder_dim=D, encoder_size=es,
                mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], embed_dim=256, embed_norm=None, attention_dim=128,
                decoder_dim=512, decoder_layers=1, dropout=0.0, embedding_dropout=0.0, label_smoothing=0.0, weight_tying=False,
                deep_output=True, att_gamma=1.0, vocab_size=V, vocab_stoi=stoi, vocab_itos={v: k for k, v in stoi.items()},
                pretrained_embedding=None, decoder_tf="always", decoder_tf_min=0.5, epochs=10, encoder_finetune_after=1,
                opt="adam", decoder_lr=1e-3, embedding_lr=1e-2, encoder_lr=1e-5, weight_decay=0.0, adam_b1=0.9, adam_b2=0.999,
                momentum=0.9, nesterov=False, scheduler=None, lr_warmup_steps=0), T, B, R


def synthetic_batch(B, R, T, V, seed, ragged, px=256):
    """img ~ U[0,1); captions [START] + U{1..V-4} + [END] + PAD; lengths all T-1 (headline) or U{8..T-1}."""
    import torch
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, px, px, generator=g)
    lengths = torch.randint(8, T, (B, R), generator=g) if ragged else torch.full((B, R), T - 1, dtype=torch.int64)
    caps = torch.zeros(B, R, T, dtype=torch.int64)
    toks = torch.randint(1, V - 3, (B, R, T), generator=g)
    pos = torch.arange(T)[None, None, :]
    caps = torch.where(pos < lengths[..., None], toks, caps)
    caps[..., 0] = V - 2
    caps.scatter_(2, lengths[..., None], V - 1)
    return img, caps, lengths


#: arch -> (block kind, blocks per stage, width per group): torchvision's table (same as sat_amd.encoder.RESNETS)
_RESNETS = {"resnet18": ("basic", (2, 2, 2, 2), 64), "resnet50": ("bottleneck", (3, 4, 6, 3), 64), "resnet101": ("bottleneck", (3, 4, 23, 3), 64),
            "wide_resnet101_2": ("bottleneck", (3, 4, 23, 3), 128)}


def algorithmic_work(cfg, R=5):
    """algorithmic FLOPs (2 x MAC; training = 3 x forward for contraction work) of one caption.
    Encoder: every convolution of the torchvision ResNet at 256 px + the 1x1 projection."""
    arch, es, D, V, T, _ = CONFIGS[cfg]
    kind, depths, wpg = _RESNETS[arch][:3]
    macs = 128 * 128 * 64 * 3 * 49                       # conv1: 7x7, stride 2 on 256 px
    hw, cin = 64, 64
    for si, (planes, nblk) in enumerate(zip((64, 128, 256, 512), depths)):
        for bi in range(nblk):
            stride = 2 if (si > 0 and bi == 0) else 1
            ho = hw // stride
            if kind == "basic":
                cout = planes
                macs += ho * ho * planes * cin * 9 + ho * ho * planes * planes * 9
            else:
                mid = int(planes * (wpg / 64.0)); cout = planes * 4
                macs += hw * hw * mid * cin + ho * ho * mid * mid * 9 + ho * ho * cout * mid
            if stride != 1 or cin != cout:
                macs += ho * ho * cout * cin
            hw, cin = ho, cout
    macs += hw * hw * D * cin                            # model.py:53
    f_enc = 2.0 * macs
    Lc, A, m, n = es * es, 128, 256, 512
    f_pre = 2.0 * Lc * D * A
    f_step = 2.0 * (n * A + Lc * A + Lc * D + n * D + (m + D + n) * 4 * n + n * m + D * m + m * V)
    f_cap = 3.0 * (f_enc / R + f_pre + (T - 1) * f_step)
    return dict(f_enc=f_enc, f_pre=f_pre, f_step=f_step, f_cap=f_cap)


def cpu_baseline(cfg, seconds_budget=24.0, max_steps=16):
    """The CPU oracle (kind "port") on B=8 images of config ``cfg``'s model: 1 warm-up + timed steps within the budget."""
    import torch
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
    log("cpu baseline (%s) warm-up step done" % cfg)
    t0 = time.time(); n = 0
    while True:
        step(); n += 1
        if time.time() - t0 > seconds_budget * 0.5 or n >= max_steps:      # ~10 s of CPU work
            break
    log("cpu baseline (%s): %d steps" % (cfg, n))
    dt = (time.time() - t0) / n
    return {"value": round(B * R / dt, 2), "unit": "captions/s", "cores": cores, "kind": "port",
            "sample": "%d full train steps (fwd+loss+bwd+Adam, fp32) of the %s model (%s) at batch %d images x %d captions after 1 warm-up; "
                      "oracle/sat_oracle.py on torch-CPU, %d threads" % (n, cfg.upper(), CONFIGS[cfg][0], B, R, cores),
            "s_per_step": round(dt, 3)}



def _build_train(cfg, dev, precision="bf16", decoder_tf="always", batch=None, rank=0, ragged=False):
    """model + optimizer + gradient exchange + one resident synthetic batch of config ``cfg`` (the train step of bench.py's headline)"""
    import torch
    from sat_amd import model as M
    from sat_amd.dist import GradSync, broadcast_parameters
    hp, T, B, R = hparams(cfg)
    if batch:
        B = batch
    if decoder_tf == "none":
        hp["decoder_tf"] = None
    if cfg == "cli":
        hp["deep_output"] = False          # train.py:160
    torch.manual_seed(42)
    model = M.SAT(**hp).to(dev).train()
    model.set_precision(precision)
    broadcast_parameters(model)
    model.__dict__["_sat_global_step"] = 2          # past encoder_finetune_after: the encoder trains (and is in the optimizer)
    opt = model.configure_optimizers()
    sync = GradSync(model)
    img, caps, lengths = synthetic_batch(B, R, T, hp["vocab_size"], 1234 + rank, ragged, px=hp["input_size"])
    hp["encoder_dim"] = model.hp.encoder_dim
    return model, opt, sync, (img.to(dev), caps.to(dev), lengths), (hp, T, B, R)


def sub_line_train(cfg, dev, steps=5, warmup=5, decoder_tf="always", attention=False, graph=False):
    """A short line for one more BASELINE configuration (per-GPU shard sizes of CONFIGS): ms per train step after ``warmup`` untimed steps (and at
    least 0.6 s of them).  ``attention``: two more steps with the attention kernels bracketed by HIP events -> their time and algorithmic bytes
    (the annotation stream once per image-step) against the HBM peak.  ``graph``: the same step replayed from a hipGraph."""
    import torch
    from sat_amd import _lib
    model, opt, sync, batch, (hp, T, B, R) = _build_train(cfg, dev, decoder_tf=decoder_tf)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model.training_step(batch, 0)
        out["loss"].backward()
        sync.finish()
        opt.step()

    t0 = time.perf_counter(); n = 0
    while n < warmup or time.perf_counter() - t0 < 0.6:
        step(); torch.cuda.synchronize(); n += 1
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": "%s: %s at %d px encoder_size=%s encoder_dim=%d vocab=%d T=%d, %d images x R=%d captions on this GPU, decoder_tf=%s" % (
               cfg.upper(), CONFIGS[cfg][0], hp["input_size"], CONFIGS[cfg][1], hp["encoder_dim"], hp["vocab_size"], T, B, R, decoder_tf),
           "ms_per_step": round(dt * 1e3, 3), "value": round(B * R / dt, 1), "unit": "captions/s", "steps": steps, "warmup": n, "dtype": "bf16"}
    if attention:
        _lib.profile_start(only="attention*")
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        rows = {}
        for e in _lib.profile_stop():
            us = e["total_ms"] * 1e3 / e["launches"]
            gbs = e["bytes"] / (e["total_ms"] * 1e-3) / 1e9
            rows[e["name"]] = {"us_per_launch": round(us, 2), "launches_per_step": e["launches"] // 2, "algorithmic_mb_per_launch": round(e["bytes"] / e["launches"] / 1e6, 2),
                               "gbytes_per_s": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)}
        out["attention"] = rows
    if graph:
        from sat_amd.graph import GraphedTrainStep
        stepper = GraphedTrainStep(model, opt, sync=sync)
        for _ in range(4):
            stepper(batch, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepper(batch, 0)
        torch.cuda.synchronize()
        out["graph_replay_ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        out["graph_steps"] = dict(stepper.stats)
    sync.remove()
    del model, opt, sync, batch
    torch.cuda.empty_cache()
    return out


def sub_line_decode(model, dev, images=64, max_len=20):
    """BASELINE configs[4] (C5): caption() on ``images`` images - the headline model (resnet50, encoder_size 7) in eval mode, batched beam search
    of width 5 and greedy (width 1), decode-only (annotations resident) eager / replayed from a hipGraph, and end to end with the encoder."""
    import torch
    hp = model.hp
    was = model.training
    model.eval()
    img = torch.rand(images, 3, hp.input_size, hp.input_size, device=dev)
    out = {"workload": "C5: resnet50 encoder, %d images of %d px, max_gen_length=%d, batched beam search (sat_beam_search_batched)" % (images, hp.input_size, max_len),
           "unit": "images/s", "dtype": "bf16"}
    with torch.no_grad():
        ann, hw = model.encode(img)
        ann = ann.contiguous()
        for beamk in (5, 1):
            row = {}
            for name, kw in (("decode_only", {}), ("decode_only_graph", {"graph": True})):
                for _ in range(2):
                    model.beam_decode_batched(ann, hw, beamk=beamk, max_gen_length=max_len, **kw)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    model.beam_decode_batched(ann, hw, beamk=beamk, max_gen_length=max_len, **kw)
                torch.cuda.synchronize()
                row[name] = round(images / ((time.perf_counter() - t0) / 5), 1)
            model.caption(img, beamk=beamk, max_gen_length=max_len)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                model.caption(img, beamk=beamk, max_gen_length=max_len)
            torch.cuda.synchronize()
            row["end_to_end"] = round(images / ((time.perf_counter() - t0) / 3), 1)
            out["beam%d" % beamk] = row
    model.train(was)
    return out


_T0 = time.time()


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as children of this process
    through torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and pass rank 0's JSON line through.
    The parent never touches the GPU (nothing here calls into torch.cuda), and no process that has is ever replaced."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:])))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0 or line is None:
        raise SystemExit("bench.py: the %d-rank run failed (exit code %s)" % (args.gpus, rc))
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="c2", choices=sorted(c for c in CONFIGS if c != "cli"))
    ap.add_argument("--ragged", action="store_true", help="lengths ~ U{8..T-1} instead of all T-1")
    ap.add_argument("--decoder-tf", default="always", choices=["always", "none"],
                    help="teacher forcing: always (epsilon = 1, the headline) or none (train.py:63 default: argmax feedback after step 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 parity-mode sub-line and the second CPU baseline")
    ap.add_argument("--batch", type=int, default=None, help="override images per GPU")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the config's global batch (BASELINE.json: C2 128, C3 256, C4 512 images) is split over the ranks "
                         "instead of every rank taking the per-GPU batch")
    ap.add_argument("--bucket-dtype", default="fp32", choices=["fp32", "bf16"], help="wire format of the gradient all-reduce buckets")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="bf16: bf16 MFMA + bf16 activations (BASELINE configs[1]); fp32: exact fp32 MFMA parity mode")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
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

    import sat_amd  # noqa: F401
    from sat_amd import _lib, model as M
    from sat_amd.dist import GradSync, broadcast_parameters

    hp, T, B, R = hparams(args.config)
    if args.strong:
        gb = GLOBAL_BATCH[args.config]
        if gb % world:
            raise SystemExit("--strong: the global batch %d does not divide over %d ranks" % (gb, world))
        B = gb // world
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
    sync = GradSync(model, bucket_dtype=torch.bfloat16 if args.bucket_dtype == "bf16" else torch.float32)
    img, caps, lengths = synthetic_batch(B, R, T, hp["vocab_size"], 1234 + rank, args.ragged)
    img, caps = img.to(dev), caps.to(dev)

    steps_run = [0]

    def step():
        steps_run[0] += 1
        opt.zero_grad(set_to_none=True)
        out = model.training_step((img, caps, lengths), 0)
        out["loss"].backward()
        sync.finish()
        opt.step()
        return out

    def agree(flag):
        """every rank must run the same number of steps (each step holds collectives)"""
        if world > 1:
            t = torch.tensor([1 if flag else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(t.item())
        return flag

    log("model built (%s, %d images/GPU); warm-up" % (args.config, B))
    # W untimed steps, and in any case MIN_WARM_S seconds of them: a GPU coming out of idle (fresh box / fresh process) needs
    # ~1 s of load before its clocks settle -- with 3 warm-up steps the first 10 timed steps ran 20 % slow (37-42 vs 32 ms)
    t_warm = time.perf_counter(); warm_done = 0
    while agree(warm_done < args.warmup or time.perf_counter() - t_warm < MIN_WARM_S):
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
    # the same two steps with the encoder's weight gradients back on the main stream: in the product (and in the timed region) they run on a
    # second HIP stream beside the data-gradient chain, and two kernels sharing the chip stretch each other's durations - these `alone` figures
    # are each kernel's duration with the chip to itself (kernel quality), the others what actually ran
    from sat_amd import encoder as _enc
    side = _enc._WGRAD_STREAM
    _enc._WGRAD_STREAM = False
    step(); torch.cuda.synchronize()
    _lib.profile_start()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    alone = _lib.profile_stop()
    _enc._WGRAD_STREAM = side
    step(); torch.cuda.synchronize()
    # the contraction core is ONE kernel family launched under ~20 template names (tile shape x operand form): rank it as one entry
    gemm = [e for e in entries if e["name"].startswith("gemm_")]
    fams = [e for e in entries if not e["name"].startswith("gemm_")]
    if gemm:
        fams.append(dict(name="gemm_*", launches=sum(e["launches"] for e in gemm), total_ms=sum(e["total_ms"] for e in gemm),
                         flops=sum(e["flops"] for e in gemm), bytes=sum(e["bytes"] for e in gemm)))
    fams.sort(key=lambda e: -e["total_ms"])
    dom_name = fams[0]["name"] if fams else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- the timed region; the dominant family alone stays instrumented inside it (an event pair per launch of that family)
    if os.environ.get("SAT_BENCH_NO_INREGION") == "1":       # dev: A/B the cost of the in-region event pairs
        dom_name = None
    # an event pair costs the command processor ~2 us; the contraction core has ~280 launches per step, so it is instrumented on
    # every 10th timed step only (all of its launches of that step): the other families have <= 52 launches per step
    instr_every = 10 if dom_name == "gemm_*" else 1
    instr_steps = 0
    if dom_name:
        _lib.profile_start(only=dom_name)
        _lib.profile_pause(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = bool(dom_name) and i % instr_every == 0
        if on:
            _lib.profile_pause(False); instr_steps += 1
        out = step()
        if on:
            _lib.profile_pause(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timed = _lib.profile_stop() if dom_name else []
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(out["loss"].item())
    log("timed region: %.1f ms/step" % (dt / args.steps * 1e3))
    # ---- N > 1: what the exchange costs beyond the backward pass it overlaps (three untimed steps): the time the stream spends between the end of
    # backward() and the end of GradSync.finish() (collectives still running + the divide), and which buckets started inside the encoder backward
    exchange = None
    if world > 1:
        ex = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            o = model.training_step((img, caps, lengths), 0)
            o["loss"].backward()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); sync.finish(); e1.record()
            opt.step()
            torch.cuda.synchronize()
            ex.append(e0.elapsed_time(e1))
        del o
        tt = torch.tensor([sorted(ex)[1]], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        exchange = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "allreduce_exposed_ms": round(float(tt.item()), 3),
                    "buckets": [{"params": len(bk.params), "mbytes": round(bk.flat.numel() * 4 / 1e6, 1), "launched_inside_backward": i in sync.last_early}
                                for i, bk in enumerate(sync.buckets)],
                    "bucket_dtype": args.bucket_dtype,
                    "what": "median over 3 untimed steps, max over ranks, of the stream time between the end of backward() and the end of GradSync.finish()"}

    line = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        caps_per_s = world * B * R / (dt / args.steps)
        dom = None
        if timed:            # the dominant family as measured INSIDE the timed region
            dom = dict(name=dom_name, launches=sum(e["launches"] for e in timed), total_ms=sum(e["total_ms"] for e in timed),
                       flops=sum(e["flops"] for e in timed), bytes=sum(e["bytes"] for e in timed))
        elif fams:
            dom = fams[0]
        dom_steps = instr_steps if timed else prof_steps
        roof = None
        peak = PEAK["bf16"] if args.precision == "bf16" else PEAK["f32"]
        step_tflop = algorithmic_work(args.config, R)["f_cap"] * B * R / 1e12
        if dom:
            hbm_bound = dom["flops"] == 0           # streaming kernels (BatchNorm passes) are instrumented with bytes only
            tf = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
            gbs = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9
            traffic = None      # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_aggregate.py), same workload only
            pmc_rows, pmc_file = [], None
            headline = args.config == "c2" and args.precision == "bf16" and not args.batch and not args.ragged and not args.strong
            for f in PMC_FILES:
                if headline and os.path.exists(f):
                    pmc_rows, pmc_file = json.load(open(f)), os.path.relpath(f, ROOT)
                    break
            if pmc_rows:
                want = PMC_FAMILY.get(dom["name"], dom["name"])
                sel = [r for r in pmc_rows if (r["family"].startswith("gemm_") if want == "gemm_*" else r["family"] == want)]
                if sel:
                    traffic = round(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in sel) / sum(r["launches"] for r in sel))

            def row(e):
                return {"kernel": e["name"], "ms_per_step": round(e["total_ms"] / prof_steps, 3), "launches_per_step": e["launches"] // prof_steps,
                        "tflops": round(e["flops"] / (e["total_ms"] * 1e-3) / 1e12, 2) if e["total_ms"] > 0 else None,
                        "gbytes_per_s": round(e["bytes"] / (e["total_ms"] * 1e-3) / 1e9, 1) if e["total_ms"] > 0 else None}

            sel_alone = [e for e in alone if (e["name"].startswith("gemm_") if dom["name"] == "gemm_*" else e["name"] == dom["name"])]
            alone_row = None
            if sel_alone and sum(e["total_ms"] for e in sel_alone) > 0:
                a_ms = sum(e["total_ms"] for e in sel_alone); a_fl = sum(e["flops"] for e in sel_alone); a_by = sum(e["bytes"] for e in sel_alone)
                a_val = (a_by / (a_ms * 1e-3) / 1e9) if hbm_bound else (a_fl / (a_ms * 1e-3) / 1e12)
                alone_row = {"kernel": dom["name"], "achieved": round(a_val, 2), "frac": round(a_val / (PEAK_HBM_GBS if hbm_bound else peak), 4),
                             "ms_per_step": round(a_ms / prof_steps, 3), "avg_launch_us": round(a_ms * 1e3 / sum(e["launches"] for e in sel_alone), 2),
                             "what": "the same family over %d untimed steps with the second stream off (SAT_WGRAD_STREAM=0): every kernel has the chip to itself" % prof_steps}
            roof = {"bound": "hbm" if hbm_bound else "mfma", "kernel": dom["name"],
                    "achieved": round(gbs if hbm_bound else tf, 2), "peak": PEAK_HBM_GBS if hbm_bound else peak,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round((gbs / PEAK_HBM_GBS) if hbm_bound else (tf / peak), 4), "traffic": traffic, "traffic_from": pmc_file,
                    "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2), "launches_per_step": dom["launches"] // dom_steps,
                    "ms_per_step": round(dom["total_ms"] / dom_steps, 3),
                    "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"], "flops_per_launch": dom["flops"] / dom["launches"],
                    "measured_on": ("HIP events on the launch stream around every launch of this family inside the timed region (on %d of its steps); "
                                    "`families` / `top`: every instrumented family over %d untimed steps before it; `gemm_*` = every template "
                                    "of the contraction core (tile shape x operand form) summed; the encoder's weight-gradient launches run on a second stream beside the "
                                    "data-gradient chain, so durations here include the stretch of sharing the chip (`alone` = without it)" % (dom_steps, prof_steps)),
                    "families": [row(e) for e in fams[:8]],
                    "top": [row(e) for e in entries[:8]],
                    "alone": alone_row,
                    # the step as a whole: algorithmic FLOPs of SURVEY 8d (3 x forward contraction work) over the measured step time,
                    # and the HBM bytes of the committed PMC passes over the same step time
                    "step_tflop": round(step_tflop, 3), "step_tflops": round(step_tflop / (ms * 1e-3), 1),
                    "step_mfma_frac": round(step_tflop / (ms * 1e-3) / peak, 4)}
            meta = [r for r in pmc_rows if r["family"] == "__meta__"]
            if meta:
                gb = meta[0]["hbm_gb_per_step"]
                roof["step_hbm_gb"] = round(gb, 2)
                roof["step_hbm_frac"] = round(gb / (ms * 1e-3) / PEAK_HBM_GBS, 4)
        line = {"metric": "captions/sec (train step) at B=128, 256px, seq_len=22", "value": round(caps_per_s, 1), "unit": "captions/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": warm_done, "train_steps_in_process": steps_run[0], "ms_per_step": round(ms, 3), "higher_is_better": True,
                "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
                "config": {"workload": "%s: %s encoder_size=%s encoder_dim=%d vocab=%d T=%d, %d images/GPU x R=%d captions, "
                                       "trainable encoder, Adam, %s lengths, decoder_tf=%s" % (args.config.upper(), CONFIGS[args.config][0],
                                                                               CONFIGS[args.config][1], CONFIGS[args.config][2],
                                                                               hp["vocab_size"], T, B, R, "ragged" if args.ragged else "full", args.decoder_tf),
                           "global_batch_images": world * B, "captions_per_step": world * B * R, "parallelism": "dp%d" % world,
                           "images_per_s": round(world * B / (dt / args.steps), 1), "final_loss": round(loss_val, 4),
                           "grad_bucket_dtype": args.bucket_dtype},
                "roofline": roof}

    # ---- fp32 parity mode (the mode that meets north_star's 1e-4): a short sub-line of the same workload, after the headline
    if world == 1 and not args.no_extras and args.precision == "bf16":
        model.set_precision("fp32")
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        n32 = 5
        t0 = time.perf_counter()
        for _ in range(n32):
            step()
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t0) / n32
        line["fp32_parity_mode"] = {"ms_per_step": round(d32 * 1e3, 3), "value": round(B * R / d32, 1), "unit": "captions/s", "steps": n32, "warmup": 3,
                                    "dtype": "f32", "note": "exact fp32 MFMA, fp32 activations: logits / alphas within 1e-4 of the reference"}
        log("fp32 parity mode: %.1f ms/step" % (d32 * 1e3))
        model.set_precision("bf16")
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    if line is not None and exchange is not None:
        line["exchange"] = exchange
    # ---- every other BASELINE configuration as a short sub-line: 5 + 5 steps each, one GPU
    if world == 1 and not args.no_extras and args.config == "c2" and args.precision == "bf16" and not args.batch:
        subs = {}
        try:
            from sat_amd.graph import GraphedTrainStep
            stepper = GraphedTrainStep(model, opt, sync=sync)
            for _ in range(4):
                stepper((img, caps, lengths), 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                stepper((img, caps, lengths), 0)
            torch.cuda.synchronize()
            subs["c2_graph_replay"] = {"ms_per_step": round((time.perf_counter() - t0) / 10 * 1e3, 3), "steps": dict(stepper.stats),
                                       "what": "the headline step captured into a hipGraph and replayed (sat_amd/graph.py); bit-equal to the eager step (tests/test_gpu_graph.py)"}
            del stepper
            log("C2 graph replay: %.1f ms/step" % subs["c2_graph_replay"]["ms_per_step"])
            subs["c5_decode"] = sub_line_decode(model, dev)
            log("C5 decode: beam 5 %s images/s, greedy %s images/s (decode only)" % (subs["c5_decode"]["beam5"]["decode_only"], subs["c5_decode"]["beam1"]["decode_only"]))
            sync.remove()
            del model, opt, sync
            torch.cuda.empty_cache()
            subs["c2_decoder_tf_none"] = sub_line_train("c2", dev, decoder_tf="none")
            log("C2 decoder_tf=None: %.1f ms/step" % subs["c2_decoder_tf_none"]["ms_per_step"])
            subs["c1"] = sub_line_train("c1", dev, steps=10, graph=True)
            log("C1: %.2f ms/step" % subs["c1"]["ms_per_step"])
            subs["cli_defaults_shufflenet"] = sub_line_train("cli", dev, steps=10, decoder_tf="none", graph=True)
            log("reference CLI defaults (shufflenet_v2_x0_5): %.2f ms/step" % subs["cli_defaults_shufflenet"]["ms_per_step"])
            subs["c3_shard"] = sub_line_train("c3", dev, graph=True)
            log("C3 shard: %.1f ms/step" % subs["c3_shard"]["ms_per_step"])
            subs["c4_shard"] = sub_line_train("c4", dev, attention=True)
            log("C4 shard: %.1f ms/step" % subs["c4_shard"]["ms_per_step"])
        except Exception as exc:          # a sub-line must never cost the headline
            subs["error"] = "%s: %s" % (type(exc).__name__, exc)
            log("sub-lines stopped: %s" % subs["error"])
        line["sub_lines"] = subs

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            log("CPU baseline (oracle) ...")
            line["cpu_baseline"] = cpu_baseline(args.config)
            if not args.no_extras and args.config != "c1":      # BASELINE.json configs[0], the reference's own CPU-runnable case
                line["cpu_baseline_c1"] = cpu_baseline("c1", seconds_budget=12.0)
            log("CPU baseline done")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

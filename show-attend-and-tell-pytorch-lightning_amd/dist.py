"""Data-parallel gradient exchange for the SAT train step: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no collective of its own: ``--gpus`` is handed to Lightning (train.py:27-28, 272), i.e. DDP's
gradient mean.  Here the exchange is explicit and shaped for xGMI (point-to-point links, per-link bound rings):
few large buckets, launched as soon as their gradients exist --

* the decoder bucket when backward-through-time has finished (a post-accumulate hook on its parameters
  fires before the encoder backward starts, so its all-reduce overlaps the whole conv-stack backward);
* encoder buckets per ResNet stage, last stage first, as the encoder backward produces them.

Every bucket owns ONE persistent flat fp32 buffer.  The backward kernels write each parameter's gradient straight into
its slice of that buffer (``_lib.grad_buffer``: a fresh view per call, which autograd then adopts as ``p.grad``), so a
bucket is all-reduced in place: no ``cat`` before the collective and no copy back after it.  A gradient that arrives
any other way (accumulation into an existing ``p.grad``, a foreign autograd node) is copied into its slice by the hook
and ``p.grad`` re-pointed at the slice.  ``finish()`` waits for the collectives and divides by the world size; the
optimizer reads the slices through ``p.grad``.  ``bucket_dtype=torch.bfloat16`` sends a bf16 copy of each bucket
(half the xGMI bytes; the sum is rounded once per hop) and writes the mean back in fp32.

Per-rank semantics are those of a single-process run on the local shard (F3: InitLSTM mixes rows of the
*local* batch; BatchNorm uses local batch statistics -- no SyncBN in the reference either)."""
import contextlib

import torch
import torch.distributed as dist

from . import _lib as L


def _slice_view(flat, off, p):
    """a fresh view of flat[off : off + p.numel()] with p's shape and memory order (KRSC for channels_last filters)"""
    seg = flat[off:off + p.numel()]
    if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
        k, c, r, s = p.shape
        return seg.view(k, r, s, c).permute(0, 3, 1, 2)
    return seg.view(p.shape)


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "launched", "work", "wire")

    def __init__(self, params):
        self.params = params
        self.offsets, off = {}, 0
        for p in params:
            self.offsets[id(p)] = off
            off += (p.numel() + 63) // 64 * 64                  # 256-byte aligned slices
        self.flat = torch.zeros(off, dtype=torch.float32, device=params[0].device)
        self.pending, self.launched, self.work, self.wire = 0, False, None, None

    def view(self, p):
        return _slice_view(self.flat, self.offsets[id(p)], p)

    def owns(self, p):
        g = p.grad
        return g is not None and g.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[id(p)] and g.dtype == torch.float32


class GradSync:
    """``sync = GradSync(model)`` once; every step: forward, ``backward()``, ``sync.finish()``, ``optimizer.step()``.
    Under gradient accumulation wrap the non-final micro-batches in ``with sync.no_sync():`` (DDP's contract)."""

    def __init__(self, model, buckets=None, bucket_dtype=torch.float32):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.model = model
        self._custom = buckets
        self.bucket_dtype = bucket_dtype
        self.enabled = True
        self._hook_handles = {}          # id(parameter) -> its post-accumulate hook
        self.buckets = []
        self._sig = None
        self._early = set()              # diagnostics: indices of the buckets launched from inside the encoder backward this step
        self.last_early = []
        self._build()

    # ------------------------------------------------------------------ construction
    def _signature(self):
        plist = self.__dict__.get("_plist")
        if plist is None:          # the module tree is walked once (it costs ~1 ms of host time per call on a ResNet-50 model)
            plist = self.__dict__["_plist"] = list(self.model.parameters())
        return tuple([p.requires_grad for p in plist])

    def _layout_stale(self):
        """True when the set of trainable parameters differs from the one the buckets were laid out for.  The cached parameter list is
        checked against the model first (``model.to()`` / ``_apply`` may replace Parameter objects: the first and the last one are looked at)."""
        plist = self.__dict__.get("_plist")
        if plist:
            it = self.model.parameters()
            first = next(it, None)
            if first is not plist[0]:
                self.__dict__.pop("_plist", None)
        return self._signature() != self._sig

    def _build(self):
        """(Re)lay out the buckets.  A bucket whose parameter list is unchanged is KEPT as it is - buffer, slices, and an all-reduce that may be
        in flight on it (``finish()`` re-lays the buckets out in the middle of a step when ``encoder_finetune_after`` unfreezes the encoder:
        the decoder bucket launched by this step's hooks must neither be cloned out of nor reduced a second time).  Returns the ids of the
        parameters whose (changed) bucket had already been reduced this step: their gradients hold the mean."""
        self.__dict__.pop("_plist", None)
        groups = self._custom if self._custom is not None else default_buckets(self.model)
        seen, clean = set(), []
        for grp in groups:                                       # tied weights appear once
            keep = []
            for p in grp:
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p)); keep.append(p)
            if keep:
                clean.append(keep)
        self._sig = self._signature()
        # one process: the same persistent buffers without the collective - gradient addresses then stay put from step to step, which lets
        # FusedOptimizer skip rebuilding / uploading its pointer table (optim.py) and saves ~180 gradient allocations per step
        old = {tuple(id(p) for p in b.params): b for b in self.buckets}
        kept, fresh, done = [], [], set()
        for g in clean:
            b = old.pop(tuple(id(p) for p in g), None)
            if b is None:
                b = _Bucket(g); fresh.append(b)
            kept.append(b)
        for b in old.values():          # buckets whose parameter list changed: finish what is in flight on them, then move the gradients out
            if b.launched:
                self._complete(b)
                done.update(id(p) for p in b.params)
            self._retire(b)
        self.buckets = kept
        self._of = {}
        for b in self.buckets:
            for p in b.params:
                self._of[id(p)] = b
        for b in fresh:
            for p in b.params:
                if self.world > 1:      # (one process: nothing to launch early; finish() adopts whatever did not land in its slice)
                    self._hook_handles[id(p)] = p.register_post_accumulate_grad_hook(self._hook)
                L.register_grad_sink(p, (lambda b=b, p=p: b.view(p)))
        enc = getattr(self.model, "encoder", None)
        if enc is not None and hasattr(enc, "precision"):          # HipEncoder: stage-by-stage notification
            enc.grad_ready = self.encoder_stage_ready if self.world > 1 else None
        return done

    def _retire(self, b):
        """a bucket that goes away: hooks and gradient sinks off, gradients that live in its buffer moved out (nothing may be in flight on it)"""
        for p in b.params:
            h = self._hook_handles.pop(id(p), None)
            if h is not None:
                h.remove()
            L.unregister_grad_sink(p)
            if b.owns(p):
                p.grad = p.grad.clone()                      # the flat buffer is about to go away

    def _complete(self, b):
        """wait for the bucket's all-reduce and leave the mean in its buffer"""
        if b.work is not None:
            b.work.wait()
            if b.wire is not None:
                b.flat.copy_(b.wire)
                b.wire = None
            b.flat.div_(self.world)
        b.work, b.launched, b.pending = None, False, 0

    def remove(self):
        """detach from the model: collectives in flight are completed first, gradients are moved out of the flat buffers"""
        for b in self.buckets:
            if b.launched:
                self._complete(b)
            self._retire(b)
        self.buckets = []

    # ------------------------------------------------------------------ per step
    @contextlib.contextmanager
    def no_sync(self):
        """micro-batches whose gradients only accumulate locally (every one but the last of an accumulation window)"""
        old, self.enabled = self.enabled, False
        try:
            yield
        finally:
            self.enabled = old

    def _adopt(self, b, p):
        """make p.grad the bucket slice (copying a gradient that was produced elsewhere)"""
        if p.grad is None:
            b.view(p).zero_()
        elif not b.owns(p):
            v = b.view(p)
            v.copy_(p.grad)
            p.grad = v

    def _hook(self, p):
        b = self._of.get(id(p))
        if b is None:
            return
        if b.launched:                   # started from inside the encoder backward: the slice already holds (or is receiving) the sum
            if not b.owns(p):
                p.grad = b.view(p)       # autograd kept a copy instead of adopting the slice: point at the slice again
        else:
            self._adopt(b, p)
        b.pending += 1
        if b.pending == len(b.params):
            b.pending = 0
            if self.enabled and not b.launched:
                self._launch(b)

    def encoder_stage_ready(self, grads):
        """Called from inside the encoder backward with {parameter: gradient} of everything computed so far.  A bucket
        whose parameters all wrote their gradient straight into its flat buffer (no earlier ``p.grad`` to add to) starts
        its all-reduce now, overlapping the rest of the conv-stack backward; autograd adopts the same slices afterwards.
        Anything else (accumulation, a gradient living elsewhere) waits for the post-accumulate hooks."""
        if self.world == 1 or not self.enabled:
            return
        for b in self.buckets:
            if b.launched:
                continue
            ok = True
            for p in b.params:
                g = grads.get(p)
                if g is None or p.grad is not None or g.data_ptr() != b.flat.data_ptr() + 4 * b.offsets[id(p)]:
                    ok = False
                    break
            if ok:
                self._early.add(self.buckets.index(b))
                self._launch(b)

    def _launch(self, b):
        b.launched = True
        if self.world == 1:
            b.work = None
            return
        if self.bucket_dtype != torch.float32:
            b.wire = b.flat.to(self.bucket_dtype)
            b.work = dist.all_reduce(b.wire, op=dist.ReduceOp.SUM, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """Wait for every bucket; afterwards every ``p.grad`` holds the mean over ranks.  Call after backward, before
        ``optimizer.step()`` (with accumulation: after the LAST micro-batch's backward)."""
        done = set()
        if self._layout_stale():
            # requires_grad changed since the buckets were laid out (encoder_finetune_after, model.py:584-586): lay the CHANGED buckets out
            # again and reduce this step's gradients from where autograd left them.  Unchanged buckets keep their buffer and whatever
            # all-reduce this step's hooks started on them; a changed bucket that had been launched is completed before its gradients move.
            done = self._build()
        for b in self.buckets:
            if not b.launched:
                if self.world == 1:
                    # one process: no collective.  A gradient produced elsewhere moves into its slice (addresses then repeat from step to
                    # step); a parameter WITHOUT a gradient keeps ``grad is None`` - the optimizer skips it, as torch and the reference do
                    for p in b.params:
                        if p.grad is not None:
                            self._adopt(b, p)
                    b.pending = 0
                    continue
                ids = [id(p) for p in b.params]
                for p in b.params:                               # parameters the hooks did not see this step
                    self._adopt(b, p)
                b.pending = 0
                if ids and all(i in done for i in ids):          # every gradient in it already holds the mean (its old bucket was reduced)
                    continue
                if any(i in done for i in ids):                  # mixed: the reduced ones are equal on every rank; sum / world returns them
                    pass
                self._launch(b)
        for b in self.buckets:
            self._complete(b)
            if self.world > 1:
                for p in b.params:
                    if p.grad is None:                           # unused this step: the mean of the ranks' zeros / gradients
                        p.grad = b.view(p)
        self.last_early = sorted(self._early)          # diagnostics (bench.py): which buckets started from inside the encoder backward this step
        self._early = set()


def default_buckets(model):
    """[decoder] + encoder stages in the order their gradients appear (projection+layer4, layer3, layer2, layer1+stem)."""
    named = list(model.named_parameters())
    dec = [p for k, p in named if not k.startswith("encoder.")]
    if getattr(getattr(model, "encoder", None), "single_bucket", False):          # shufflenet_v2: 0.3 - 2.5 M parameters, one exchange
        return [dec, [p for k, p in named if k.startswith("encoder.")]]
    groups = {"a": [], "b": [], "c": [], "d": []}
    for k, p in named:
        if not k.startswith("encoder."):
            continue
        idx = k.split(".")[1]
        key = {"9": "a", "8": "a", "7": "b", "6": "c"}.get(idx, "d")
        groups[key].append(p)
    return [dec, groups["a"], groups["b"], groups["c"], groups["d"]]


def broadcast_parameters(model, src=0):
    """Start every rank from rank 0's weights (what DDP does at construction)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)

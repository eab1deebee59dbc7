"""Data-parallel gradient exchange for the SAT train step: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no collective of its own: ``--gpus`` is handed to Lightning (train.py:27-28, 272), i.e. DDP's
gradient mean.  Here the exchange is explicit and shaped for xGMI (point-to-point links, per-link bound rings):
few large buckets, launched as soon as their gradients exist --

* the decoder bucket when backward-through-time has finished (a post-accumulate hook on its parameters
  fires before the encoder backward starts, so its all-reduce overlaps the whole conv-stack backward);
* encoder buckets per ResNet stage, last stage first, as the encoder backward produces them.

Each bucket is flattened into one fp32 buffer, all-reduced asynchronously on the communicator's stream and
copied back (divided by the world size) in ``finish()``, which the optimizer step waits on.
Per-rank semantics are those of a single-process run on the local shard (F3: InitLSTM mixes rows of the
*local* batch; BatchNorm uses local batch statistics -- no SyncBN in the reference either)."""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, buckets=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if buckets is None:
            buckets = default_buckets(model)
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if b]
        self._pending = [0] * len(self.buckets)
        self._inflight = []
        self._handles = []
        self._early = set()                # buckets already launched from inside the encoder backward this step
        if self.world > 1:
            for bi, bucket in enumerate(self.buckets):
                for p in bucket:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
            enc = getattr(model, "encoder", None)
            if enc is not None and hasattr(enc, "precision"):          # HipEncoder: stage-by-stage notification
                enc.grad_ready = self.encoder_stage_ready

    def _make_hook(self, bi):
        def hook(param):
            self._pending[bi] += 1
            if self._pending[bi] == len(self.buckets[bi]):
                self._pending[bi] = 0
                if bi not in self._early:
                    self._launch(bi)
        return hook

    def encoder_stage_ready(self, grads):
        """Called from the encoder backward with {parameter: gradient} of everything computed so far: buckets that
        are complete start their all-reduce now, overlapping the rest of the conv-stack backward."""
        if self.world == 1:
            return
        for bi, bucket in enumerate(self.buckets):
            if bi in self._early or not bucket or any(p not in grads for p in bucket):
                continue
            flat = torch.cat([grads[p].reshape(-1) for p in bucket])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            self._inflight.append((work, flat, list(bucket)))
            self._early.add(bi)

    def _launch(self, bi):
        params = [p for p in self.buckets[bi] if p.grad is not None]
        if not params:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._inflight.append((work, flat, params))

    def finish(self):
        """Wait for every bucket and write the averaged gradients back.  Call after backward, before optimizer.step()."""
        if self.world == 1:
            return
        for bi, n in enumerate(self._pending):       # buckets whose parameters did not all receive a gradient
            if n:
                self._pending[bi] = 0
                if bi not in self._early:
                    self._launch(bi)
        self._early = set()
        for work, flat, params in self._inflight:
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in params:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._inflight = []

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def default_buckets(model):
    """[decoder] + encoder stages in the order their gradients appear (projection+layer4, layer3, layer2, layer1+stem)."""
    named = list(model.named_parameters())
    dec = [p for k, p in named if not k.startswith("encoder.")]
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

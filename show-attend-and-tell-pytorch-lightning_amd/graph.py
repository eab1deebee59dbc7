"""Whole train step replayed from a hipGraph ("no per-step host work on the hot loop").

The eager step is ~830 kernel launches issued through ~600 Python -> ctypes calls.  At BASELINE configs[1] (C2) the GPU needs 22 ms for what
the host issues in 8, but the small configurations are HOST-bound: C1 (resnet18, 8 images) and the per-GPU shard of C3 (resnet101, 32
images) step exactly as fast as Python can issue them.  Every entry point of the library enqueues kernels only (include/sat_hip.h,
"Stream capture"), so one step -

    forward (encoder, decoder loop, cross-entropy + doubly stochastic loss)  ->  backward  ->  FusedOptimizer.step_device()

- is captured once into a hipGraph (``torch.cuda.CUDAGraph``: stream capture + the caching allocator's private pool) and replayed afterwards:
one graph launch per step.  What stays on the host each step is what the reference does on the host too (model.py:559-586, 614-626): the
teacher-forcing epsilon and its CPU-generator draws (F7), the learning-rate recipe, the optimizer's step count -> bias corrections, which
reach the replayed update kernel through device memory (``sat_optimizer_step_dev``).

A graph is valid for one launch sequence, so graphs are keyed by everything that shapes it: tensor shapes, the packing plan (caption
lengths), the teacher-forcing flags, precision, the set of trainable parameters.  The first step with a new key runs eagerly (it also warms
every cache the capture must not touch: plans, split-K scratch, kernel attributes), the second one captures, later ones replay.  Anything
the graph cannot express falls back to the eager step for that call: more than one process (the gradient all-reduce is launched from
autograd hooks), gradient accumulation, dropout (its seed is a launch argument), a parameter changed or moved behind the graph's back.

    step = GraphedTrainStep(model, optimizer)          # optionally sync=GradSync(model)
    for batch in loader:
        out = step(batch)                              # = zero_grad; training_step; backward; [sync.finish]; optimizer.step
"""
import collections

import numpy as np
import torch

from . import _lib as L
from . import decoder as Dk
from . import encoder as E


class _Ctx:
    """what the ``torch.autograd.Function`` bodies of this package use of their ``ctx``, without autograd: the captured step calls the static
    ``forward`` / ``backward`` methods of EncoderFn / DecoderTrainFn / the loss functions directly, in the order autograd would.  No engine, no
    AccumulateGrad nodes: those remember the stream they were created on and live as long as ANY tensor of an earlier step's autograd graph
    does (a kept loss); one of them inside a capture pulls the default stream in and hipStreamEndCapture crashes."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tuple(tensors)

    def mark_non_differentiable(self, *tensors):
        pass


class _Entry:
    __slots__ = ("graph", "img", "caps", "lengths", "out", "grads", "keep", "opt_bufs", "opt_sig", "replays")


class GraphedTrainStep:
    def __init__(self, model, optimizer, sync=None, max_graphs=4, enabled=True):
        self.model, self.opt, self.sync = model, optimizer, sync
        self.enabled = bool(enabled)
        self.max_graphs = int(max_graphs)
        self._graphs = collections.OrderedDict()         # key -> _Entry, or None after the eager (warming) step with that key
        self._pool = None
        self._stream = None
        self._current = None
        self._params = [p for g in optimizer.param_groups for p in g["params"]]
        self.stats = collections.Counter()               # eager / captured / replayed steps, and why a step stayed eager

    # ------------------------------------------------------------------ what a graph depends on
    def _why_eager(self, img):
        import torch.distributed as dist
        hp = self.model.hp
        if not self.enabled:
            return "disabled"
        if not img.is_cuda:
            return "cpu batch"
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return "more than one process"                # the all-reduce is launched from autograd hooks, bucket by bucket
        if int(getattr(hp, "accumulate", 1) or 1) > 1:
            return "gradient accumulation"
        if self.model.training and (float(hp.dropout) > 0.0 or float(hp.embedding_dropout) > 0.0):
            return "dropout"                              # the mask seed of the step is a launch argument
        if not hasattr(self.opt, "step_device"):
            return "optimizer is not a FusedOptimizer"
        return None

    def _params_sig(self):
        return tuple([(p.data_ptr(), p.requires_grad) for p in self._params])

    def _stale_copies(self):
        """a bf16 filter copy the encoder would remake (its parameter changed in place behind the optimizer's back)"""
        for p in self._params:
            if getattr(p, "_sat_bf16_shadow", None) is not None and p._sat_shadow_version != p._version:
                return True
        return False

    # ------------------------------------------------------------------ the eager step (also the reference for tests)
    def _eager(self, batch, epsilon, gstep, teacher):
        self._current = None
        self.opt.zero_grad(set_to_none=True)
        loss = self.model._step_losses(batch, epsilon, teacher=teacher)
        acc = self.model.criterion.last_accuracy
        self.model._step_end(gstep)
        loss.backward()
        if self.sync is not None:
            self.sync.finish()
        self.opt.step()
        return {"loss": loss.detach(), "accuracy": acc, "epsilon_tf": float(epsilon)}

    def _capture_stream(self, dev):
        if self._stream is None:
            self._stream = torch.cuda.Stream(dev)
            self._pool = torch.cuda.graph_pool_handle()
        return self._stream

    def _eager_on_capture_stream(self, batch, epsilon, gstep, teacher):
        cur = torch.cuda.current_stream(batch[0].device)
        side = self._capture_stream(batch[0].device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = self._eager(batch, epsilon, gstep, teacher)
        cur.wait_stream(side)
        for t in (batch[0], batch[1]):
            t.record_stream(side)
        return out

    # ------------------------------------------------------------------ the device half of one step without autograd
    def _forward_backward(self, batch, epsilon, teacher):
        """forward + losses + backward of ``SAT._step_losses`` / ``loss.backward()`` as plain calls (see _Ctx); leaves every gradient in ``p.grad``
        (written, not accumulated: the caller has set the gradients to None) and returns (loss, accuracy)"""
        model = self.model
        img, caps, lengths = batch
        hp = model.hp
        enc = model.encoder
        with torch.no_grad():
            enc_params = list(enc.parameters())
            ectx = _Ctx()
            try:
                ann = enc.Fn._forward(ectx, img, enc, *enc_params)                 # (B, D, h, w) view over NHWC memory
            finally:
                E._defer[0] = False
            Bn, D, h, w = ann.shape
            ann_bld = ann.permute(0, 2, 3, 1).reshape(Bn, h * w, D)
            _, R, T = caps.shape
            plan = Dk.PackPlan.cached(lengths.reshape(-1), T, img.device)
            caps2 = caps.reshape(Bn * R, T)
            caps_i32 = caps2.to(dtype=torch.int32).contiguous()
            dec_params = model.param_list()
            dctx = _Ctx()
            logits, alphas = Dk.DecoderTrainFn.forward(dctx, ann_bld, caps_i32, plan, teacher, bool(hp.deep_output), model.pad_idx, R,
                                                       int(model.sat_precision == "bf16"), getattr(hp, "embed_norm", None) or 0.0,
                                                       model._dropout_args(None), *dec_params)
            targets = plan.pack(caps2[:, 1:].unsqueeze(-1)).squeeze(-1)
            cctx, sctx = _Ctx(), _Ctx()
            ce, acc = Dk.LabelSmoothingFn.forward(cctx, logits, targets.to(torch.int32), model.criterion.smoothing)
            model.criterion.last_accuracy = acc
            ds = Dk.DoublyStochasticFn.forward(sctx, alphas, float(hp.att_gamma))
            loss = ce + ds
            one = torch.ones((), dtype=torch.float32, device=img.device)
            dlogits = Dk.LabelSmoothingFn.backward(cctx, one, None)[0]
            dalphas = Dk.DoublyStochasticFn.backward(sctx, one)[0]
            douts = Dk.DecoderTrainFn.backward(dctx, dlogits, dalphas)
            dann = douts[0]
            grads = {}

            def give(p, g):
                if p is None or g is None or not p.requires_grad:
                    return
                if id(p) in grads:                               # a tied weight reaches the decoder twice (embedding and output layer)
                    grads[id(p)][1].add_(g)
                else:
                    grads[id(p)] = (p, g)

            for p, g in zip(dec_params, douts[10:]):
                give(p, g)
            if any(p.requires_grad for p in enc_params):
                eouts = enc.Fn.backward(ectx, dann.reshape(Bn, h, w, D).permute(0, 3, 1, 2))
                for p, g in zip(enc_params, eouts[2:]):
                    give(p, g)
            for p, g in grads.values():
                p.grad = g
        return loss, acc

    # ------------------------------------------------------------------ capture
    def _capture(self, key, batch, epsilon, teacher):
        img, caps, lengths = batch
        dev = img.device
        ent = _Entry()
        ent.img, ent.caps, ent.lengths = img.clone(), caps.clone(), lengths
        self._capture_stream(dev)
        self.opt.zero_grad(set_to_none=True)              # the captured backward WRITES its gradients (nothing to accumulate onto)
        self.opt.prepare_device_step(dev)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._pool, stream=self._stream):
            out_loss, acc = self._forward_backward((ent.img, ent.caps, ent.lengths), epsilon, teacher)      # no autograd inside a capture (see _Ctx)
            if self.sync is not None:
                self.sync.finish()                        # one process: gradients produced elsewhere move into their bucket slices
            self.opt.step_device()
        ent.graph = g
        ent.out = {"loss": out_loss, "accuracy": acc, "epsilon_tf": float(epsilon)}
        ent.grads = [(p, p.grad) for p in self._params if p.grad is not None]
        # module-level caches the launches read through raw pointers: keep THESE tensors alive even if the caches grow and drop them
        ent.keep = [list(E._slabs.values()), list(E._bn_scratch_buf.values()), Dk.PackPlan.cached(lengths.reshape(-1), caps.shape[-1], dev), teacher]
        ent.opt_bufs = self.opt.device_buffers()
        ent.opt_sig = self.opt.table_signature()
        ent.replays = 0
        self._graphs[key] = ent
        while len(self._graphs) > self.max_graphs:
            self._graphs.popitem(last=False)
        self.stats["captured"] += 1
        return ent

    # ------------------------------------------------------------------ one train step
    def __call__(self, batch, batch_idx=0):
        model = self.model
        img, caps, lengths = batch
        lengths = torch.as_tensor(lengths).cpu()
        batch = (img, caps, lengths)
        epsilon, gstep = model._step_begin()
        plan = Dk.PackPlan.cached(lengths.reshape(-1), caps.shape[-1], img.device)
        teacher = plan.teacher_flags(float(epsilon))       # the CPU-generator draws of this step, exactly as the eager step makes them (F7)
        why = self._why_eager(img)
        if why is not None:
            self.stats["eager: " + why] += 1
            return self._eager(batch, epsilon, gstep, teacher)
        key = (tuple(img.shape), tuple(caps.shape), lengths.numpy().tobytes(), np.asarray(teacher, np.int32).tobytes(), float(epsilon),
               model.sat_precision, model.training, self._params_sig())
        stale = self._stale_copies()
        if stale:
            self._graphs.clear()                          # filter copies are about to be remade: every graph points at the old ones
        if key not in self._graphs:
            # warm every cache (plans, split-K scratch, kernel attributes, optimizer tables) with an eager step ON THE CAPTURE STREAM (the
            # per-stream scratch buffers are then the ones the capture uses); the next step with this key captures
            self._graphs[key] = None
            self.stats["eager: first step of a shape / plan"] += 1
            return self._eager_on_capture_stream(batch, epsilon, gstep, teacher)
        ent = self._graphs[key]
        self._graphs.move_to_end(key)
        if ent is None:
            ent = self._capture(key, batch, epsilon, teacher)
        else:
            if img.data_ptr() != ent.img.data_ptr():
                ent.img.copy_(img, non_blocking=True)
            if caps.data_ptr() != ent.caps.data_ptr():
                ent.caps.copy_(caps, non_blocking=True)
        if self._current is not ent:                      # the gradients of THIS graph are what the optimizer's table must point at
            for p, gr in ent.grads:
                p.grad = gr
            self._current = ent
        if self.opt.table_signature() != ent.opt_sig or self.opt.device_buffers() != ent.opt_bufs:
            # the optimizer would rebuild its tables (they are baked into the graph): drop the graph, this step runs eagerly
            del self._graphs[key]
            self.stats["eager: optimizer tables moved"] += 1
            return self._eager(batch, epsilon, gstep, teacher)
        model._step_end(gstep)
        self.opt.step_host()
        ent.graph.replay()
        ent.replays += 1
        self.stats["replayed"] += 1
        return ent.out

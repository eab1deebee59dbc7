"""Multi-tensor optimizer step on the HIP library: the optimizers ``SAT.configure_optimizers`` builds
(reference model.py:723-757: SGD / Adam / AdamW over per-module parameter groups) and Lightning's gradient clipping
(train.py:93-96, 273-274: ``clip_grad_value_`` / ``clip_grad_norm_``) in one launch over every parameter tensor.

``FusedOptimizer`` is a ``torch.optim.Optimizer``: ``param_groups`` (so the reference's LR schedulers drive it unchanged)
and the per-parameter ``state`` keys of torch's own optimizers (``step``, ``exp_avg``, ``exp_avg_sq`` /
``momentum_buffer``), so optimizer checkpoints stay interchangeable."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

KINDS = {"sgd": 0, "adam": 1, "adamw": 2}


def _same_layout(a, b):
    """same memory order: equal strides on every dimension that has more than one element (size-1 dimensions of a
    channels_last 1x1 filter carry arbitrary strides)"""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


class FusedOptimizer(torch.optim.Optimizer):
    def __init__(self, params, kind="adam", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, momentum=0.0, nesterov=False,
                 grad_clip=None, clip_value=0.0):
        if kind not in KINDS:
            raise ValueError("kind=%r (sgd / adam / adamw)" % (kind,))
        if kind == "adamw" and weight_decay == 0.0:
            weight_decay = 1e-2                        # torch.optim.AdamW's default for groups that do not set it
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, momentum=momentum, nesterov=nesterov))
        self.kind = kind
        self.set_clipping(grad_clip, clip_value)
        self._chunks = None
        self._sig = None
        self.last_grad_norm = None                     # device scalar after a step with norm clipping

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._fast_key = None                          # the state tensors were replaced: rebuild the pointer table on the next step

    def set_clipping(self, algorithm, value):
        """train.py:93-96: ``--grad_clip value|norm`` with ``--clip_value`` (0 = off)."""
        if algorithm not in (None, "value", "norm"):
            raise ValueError("grad_clip=%r" % (algorithm,))
        self.grad_clip, self.clip_value = (algorithm, float(value)) if value and value > 0 else (None, 0.0)

    def _tables(self, entries, dev):
        sig = tuple((p.data_ptr(), p.numel()) for p, _ in entries)
        if sig != self._sig:
            ce = L.lib().sat_optimizer_chunk_elems()
            rows = [(ti, 0, s) for ti, (p, _) in enumerate(entries) for s in range(0, p.numel(), ce)]
            arr = np.array(rows, dtype=[("tensor", np.int32), ("reserved", np.int32), ("start", np.int64)])
            self._chunks = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
            self._n_chunks = len(rows)
            # The pointer table is rebuilt on the host every step and uploaded asynchronously.  The host may run a step or more
            # ahead of the GPU, so a pinned table must not be rewritten before its previous upload has executed: a small ring
            # of pinned tables, each with the event of its last upload (and no upload at all while the table is unchanged).
            nbytes = len(entries) * C.sizeof(L.OptTensor)
            self._ring = [[torch.empty(nbytes, dtype=torch.uint8).pin_memory(), None] for _ in range(3)]
            self._ring_pos = 0
            self._uploaded = None
            self._dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._scratch = torch.empty(self._n_chunks, dtype=torch.float64, device=dev)
            self._coef = torch.ones(2, dtype=torch.float32, device=dev)
            self._sig = sig
        return self._chunks

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        host = self._host_step()
        if host is None:
            return loss
        g0, beta1, beta2, step, chunks = host
        return self._launch(L.lib(), g0, beta1, beta2, step, chunks, loss)

    # ---- the step in two halves for a replayed launch (graph.GraphedTrainStep): everything the host does - step counters, the pointer /
    # learning-rate table and its upload, the hyper-parameters of this step written to DEVICE memory - and the kernel launches, which read
    # nothing but device memory and can therefore sit in a captured graph
    @torch.no_grad()
    def step_host(self):
        host = self._host_step()
        if host is None:
            raise RuntimeError("FusedOptimizer.step_host: no parameter has a gradient")
        g0, beta1, beta2, step, _ = host
        hyper = self._hyper(g0, beta1, beta2, step)
        dev = self._dev.device
        self.prepare_device_step(dev)
        self._hyper_pos = (self._hyper_pos + 1) % len(self._hyper_ring)
        slot = self._hyper_ring[self._hyper_pos]
        if slot[1] is not None:
            slot[1].synchronize()                      # the upload that last read this pinned block has executed (the host may run steps ahead)
        C.memmove(slot[0].data_ptr(), C.addressof(hyper), C.sizeof(L.OptHyper))
        self._hyper_dev.copy_(slot[0], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev)); slot[1] = ev

    def step_device(self):
        """the launches of one step (gradient-norm coefficient when clipping by norm, the update), hyper-parameters from device memory"""
        lib = L.lib()
        if getattr(self, "_hyper_dev", None) is None:
            raise RuntimeError("FusedOptimizer.step_device before step_host")
        coef = None
        if self.grad_clip == "norm":
            L.check(lib.sat_grad_clip_coef(L.ptr(self._dev), L.ptr(self._chunks), self._n_chunks, self.clip_value, L.ptr(self._scratch),
                                           L.ptr(self._coef), L.stream_ptr()), "sat_grad_clip_coef")
            coef = self._coef
            self.last_grad_norm = self._coef[1]
        L.check(lib.sat_optimizer_step_dev(L.ptr(self._dev), L.ptr(self._chunks), self._n_chunks, L.ptr(self._hyper_dev), L.ptr(coef), L.stream_ptr()),
                "sat_optimizer_step_dev")

    def prepare_device_step(self, dev):
        """device / pinned blocks of the hyper-parameters (allocated once, before a capture)"""
        if getattr(self, "_hyper_dev", None) is None or self._hyper_dev.device != dev:
            n = C.sizeof(L.OptHyper)
            self._hyper_dev = torch.zeros(n, dtype=torch.uint8, device=dev)
            self._hyper_ring = [[torch.empty(n, dtype=torch.uint8).pin_memory(), None] for _ in range(4)]
            self._hyper_pos = 0

    def table_signature(self):
        """what ``_tables`` keys the chunk / pointer tables by, for the parameters that hold a gradient now"""
        return tuple((p.data_ptr(), p.numel()) for g in self.param_groups for p in g["params"] if p.grad is not None)

    def device_buffers(self):
        """addresses a captured ``step_device`` has baked in (graph.py re-captures when they move)"""
        return tuple(t.data_ptr() for t in (self._dev, self._chunks, self._scratch, self._coef, self._hyper_dev))

    def _host_step(self):
        lib = L.lib()
        entries = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    entries.append((p, group))
        if not entries:
            return None
        L.require_gpu(*[p for p, _ in entries])
        dev = entries[0][0].device
        g0 = self.param_groups[0]
        beta1, beta2 = g0["betas"]
        for g in self.param_groups:
            if tuple(g["betas"]) != (beta1, beta2) or g["eps"] != g0["eps"] or g["momentum"] != g0["momentum"] or g["nesterov"] != g0["nesterov"]:
                raise ValueError("FusedOptimizer: betas / eps / momentum / nesterov must be the same in every group (lr and weight_decay may differ)")
        # Fast path: the same tensors, gradient addresses, learning rates and weight decays as the step before - the table on the device is
        # current, nothing to rebuild or upload (the per-parameter loop below is ~1.2 ms of host time at C2, 2.5 ms for resnet101, and C1 / C3
        # are host-bound).  Gradients living in persistent buckets (dist.GradSync) or reallocated at the same addresses hit it every step.
        # Everything the table on the device points at is part of the key: the bf16 filter copies (remade by encoder.py whenever a parameter
        # changed behind the optimizer's back - load_state_dict, broadcast, clamping - or created later by set_precision) and the state
        # tensors (replaced by anything but this class's load_state_dict) would otherwise be written through stale addresses.
        state = self.state
        mv_key = []
        for p, _ in entries:
            st = state.get(p)
            if not st:
                mv_key = None
                break
            if self.kind == "sgd":
                m = st.get("momentum_buffer"); mv_key.append(0 if m is None else m.data_ptr())
            else:
                mv_key.append(st["exp_avg"].data_ptr()); mv_key.append(st["exp_avg_sq"].data_ptr())
        shadows = [getattr(p, "_sat_bf16_shadow", None) for p, _ in entries]
        fast_key = (tuple([p.grad.data_ptr() for p, _ in entries]), tuple([(g["lr"], g["weight_decay"]) for g in self.param_groups]),
                    tuple([p.data_ptr() for p, _ in entries]), tuple([0 if s is None else s.data_ptr() for s in shadows]),
                    None if mv_key is None else tuple(mv_key))
        if mv_key is not None and fast_key == getattr(self, "_fast_key", None) and self._fast_ok:
            self.fast_path_steps = getattr(self, "fast_path_steps", 0) + 1
            self._fast_steps += 1                      # every state["step"] is a view of this one tensor (see below)
            step = float(self._fast_steps[0])
            for q in self._fast_shadowed:
                q._sat_shadow_version = q._version
            return g0, beta1, beta2, step, self._chunks
        chunks = self._tables(entries, dev)
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        slot = self._ring[self._ring_pos]
        if slot[1] is not None:
            slot[1].synchronize()                      # the upload that last read this pinned table has executed
        host = slot[0]
        table = (L.OptTensor * len(entries)).from_buffer(host.numpy())
        steps = []          # the per-parameter step counters (torch's state layout), advanced together below
        for i, (p, group) in enumerate(entries):
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))      # KRSC filters
            if p.dtype != torch.float32 or not dense:
                raise ValueError("FusedOptimizer: parameters must be dense fp32 (contiguous or channels_last)")
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                if self.kind == "sgd":
                    st["momentum_buffer"] = torch.zeros_like(p) if group["momentum"] != 0 else None
                else:
                    st["exp_avg"] = torch.zeros_like(p); st["exp_avg_sq"] = torch.zeros_like(p)
            steps.append(st["step"])
            grad = p.grad if _same_layout(p.grad, p) else torch.empty_like(p).copy_(p.grad)     # element i of g <-> element i of p
            m = st.get("momentum_buffer") if self.kind == "sgd" else st["exp_avg"]
            v = None if self.kind == "sgd" else st["exp_avg_sq"]
            t = table[i]
            t.p = p.data_ptr(); t.g = grad.data_ptr(); t.m = 0 if m is None else m.data_ptr(); t.v = 0 if v is None else v.data_ptr()
            t.n = p.numel(); t.lr = float(group["lr"]); t.weight_decay = float(group["weight_decay"])
            sh = getattr(p, "_sat_bf16_shadow", None)      # bf16 filter copy of the encoder (encoder.py): kept current by this step
            t.shadow_bf16 = sh.data_ptr() if (sh is not None and _same_layout(sh, p)) else 0
            if t.shadow_bf16:
                p._sat_shadow_version = p._version
            entries[i] = (p, group, grad)              # keep a re-laid-out gradient alive until the launch
        # torch's state layout keeps one 0-dim "step" tensor per parameter; here they are views of ONE host tensor, so the fast path advances all
        # of them with a single add (a foreach add over ~170 host scalars was 0.4 ms a step).  state_dict() / load_state_dict() see ordinary
        # 0-dim tensors; after a load the addresses in the fast key change and this path re-gathers them.
        sv = torch.stack(steps) + 1
        step = float(sv[0])
        if float(sv.min()) != float(sv.max()):
            raise ValueError("FusedOptimizer: parameters are at different step counts %s" % sorted(set((sv - 1).tolist())))
        for i, (p, _, _) in enumerate(entries):
            self.state[p]["step"] = sv[i]
        steps = sv
        raw = host.numpy().tobytes()
        if raw != self._uploaded:                      # pointers / lr / weight decay changed since the table on the device was written
            self._dev.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            slot[1] = ev
            self._uploaded = raw
        # the fast path may take over when every gradient was used where it lies (no re-laid-out copy whose address would be stale)
        if mv_key is None:          # first step: the state tensors were created above
            mv = []
            for p, _, _ in entries:
                st = state[p]
                if self.kind == "sgd":
                    m = st.get("momentum_buffer"); mv.append(0 if m is None else m.data_ptr())
                else:
                    mv.append(st["exp_avg"].data_ptr()); mv.append(st["exp_avg_sq"].data_ptr())
            fast_key = fast_key[:4] + (tuple(mv),)
        self._fast_key, self._fast_steps = fast_key, steps
        self._fast_ok = all(e[2] is e[0].grad for e in entries)
        self._fast_shadowed = [e[0] for i, e in enumerate(entries) if table[i].shadow_bf16]      # only the copies the kernel really rewrites
        return g0, beta1, beta2, step, chunks

    def _hyper(self, g0, beta1, beta2, step):
        return L.OptHyper(kind=KINDS[self.kind], nesterov=int(bool(g0["nesterov"])), first_step=int(step == 1), beta1=beta1, beta2=beta2,
                          eps=g0["eps"], bias_correction1=1.0 - beta1 ** step, bias_correction2_sqrt=math.sqrt(1.0 - beta2 ** step),
                          momentum=float(g0["momentum"]), clip_value=self.clip_value if self.grad_clip == "value" else 0.0)

    def _launch(self, lib, g0, beta1, beta2, step, chunks, loss):
        coef = None
        if self.grad_clip == "norm":
            L.check(lib.sat_grad_clip_coef(L.ptr(self._dev), L.ptr(chunks), self._n_chunks, self.clip_value, L.ptr(self._scratch),
                                           L.ptr(self._coef), L.stream_ptr()), "sat_grad_clip_coef")
            coef = self._coef
            self.last_grad_norm = self._coef[1]
        hyper = self._hyper(g0, beta1, beta2, step)
        L.check(lib.sat_optimizer_step(L.ptr(self._dev), L.ptr(chunks), self._n_chunks, C.byref(hyper), L.ptr(coef), L.stream_ptr()),
                "sat_optimizer_step")
        return loss

"""Data-parallel training step for the native BSI/DiT path: one process per GPU, gradients averaged with an RCCL
all-reduce (torch.distributed backend "nccl" on ROCm) on per-block buckets that start as soon as the HIP backward
has produced a block's gradients, then one fused clip + AdamW + EMA pass over flat fp32 buffers.

Replaces, for this path, what the reference obtains from Lightning: `DistributedDataParallel(model,
static_graph=True)` (bsi/tasks/bsi.py:163-166), `gradient_clip_val: 1.0` (config/train.yaml:40), fused
`torch.optim.AdamW` (config/task/optimizer/adamw.yaml, config/experiment/imagenet32.yaml:27-30), the warm-up/cosine
LR schedule stepped every optimizer step (bsi/lr_scheduler.py:35-58, bsi/tasks/bsi.py:303-308) and `EMA.update()`
after every batch (bsi/tasks/bsi.py:196-198, bsi/tasks/ema_pytorch.py:308-340).
"""
import copy
import ctypes as C
import math
import os

import torch
import torch.distributed as dist

from . import _native as N


def split_batch(batch_size: int, world_size: int, rank: int) -> int:
    """Per-rank share of a global batch (bsi/data/h5image.py:309-312)."""
    return batch_size // world_size + int(rank < (batch_size % world_size))


def rank_indices(n: int, world_size: int, rank: int):
    """Indices a rank evaluates out of n samples (DistributedNonPaddingSampler, bsi/data/sampler.py:63)."""
    return list(range(rank, n, world_size))


def warmup_cosine_lr(step: int, *, base_lr: float, warmup_steps: int, max_steps: int, start_lr: float, end_lr: float) -> float:
    """Learning rate in effect for optimizer step `step` (0-based) under WarmUpCosineAnnealing
    (bsi/lr_scheduler.py:10-58, config/task/lr_scheduler/cosine.yaml): linear from start_lr to base_lr over
    warmup_steps, then cosine from base_lr to end_lr over max_steps - warmup_steps."""
    if step < warmup_steps:
        return start_lr + (base_lr - start_lr) * (step / warmup_steps)
    t_max = max_steps - warmup_steps
    s = min(step - warmup_steps, t_max)
    return end_lr + (base_lr - end_lr) * (1 + math.cos(math.pi * s / t_max)) / 2


def ema_weight(call_index: int, *, beta: float = 0.9999, update_after_step: int = 1000, inv_gamma: float = 1.0,
               power: float = 2.0 / 3.0, min_value: float = 0.0) -> float:
    """Lerp weight (1 - decay) of the `call_index`-th (0-based) call of EMA.update() (ema_pytorch.py:308-340):
    the first call and every call with step <= update_after_step copy the online weights (weight 1.0)."""
    if call_index <= update_after_step:
        return 1.0
    epoch = (call_index + 1) - update_after_step - 1
    value = 1 - (1 + epoch / inv_gamma) ** -power
    decay = max(min_value, min(value, beta))
    return 1.0 - decay


class GradExchange:
    """Gradient exchange of one optimizer step over the flat gradient buffer — the ONE code path `DPTrainer.train_step`
    runs when world > 1, written device-agnostically (RCCL on GPUs, gloo in the CPU tests).

    `plan` is the ordered list of buckets `(begin, end, gate)`: the buckets are sum-all-reduced in this order; `gate` is the
    index of the backward event the bucket has to wait for (block l of the DiT: its gradients are complete when the backward
    has enqueued block l), or None for buckets that need the whole backward (patch encoder, decoder, or the single bucket
    of a model without per-block events).  Replaces DDP's bucketed reducer for this path (bsi/tasks/bsi.py:163-166:
    `DistributedDataParallel(model, static_graph=True)`); the 1/world of DDP's average is applied by the fused optimizer
    kernel (`bsi_clip_adamw_ema`, grad_scale), not here."""

    def __init__(self, plan, group=None):
        self.plan = [(int(b), int(e), g) for b, e, g in plan if e > b]
        self.group = group

    @classmethod
    def for_params(cls, fp, block_prefixes, group=None):
        """Plan for a `FlatParams` layout: one bucket per prefix of `block_prefixes` (in forward order; reduced LAST block
        first, as the backward completes them, gate = block index), then everything before the first block and everything
        after the last one as two ungated buckets.  With no block prefixes: one ungated bucket over the whole buffer."""
        n = fp.flat.numel()
        if not block_prefixes:
            return cls([(0, n, None)], group)
        spans = [fp.span(pre) for pre in block_prefixes]
        for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
            assert e0 == b1, "block parameter spans must be adjacent in the flat buffer"
        plan = [(b, e, l) for l, (b, e) in reversed(list(enumerate(spans)))]
        plan += [(0, spans[0][0], None), (spans[-1][1], n, None)]
        return cls(plan, group)

    def covers(self, n):
        """True when the buckets tile [0, n) exactly once."""
        pos = 0
        for b, e, _ in sorted(self.plan):
            if b != pos:
                return False
            pos = e
        return pos == n

    def run(self, flat_g, wait_gate=None, wait_all=None):
        """All-reduce (sum) every bucket of `flat_g` in plan order.  `wait_gate(l)` is called before the first bucket gated
        on event l, `wait_all()` once before the first ungated bucket (on GPUs: stream waits; in a synchronous backward they
        may be None)."""
        waited_all = False
        for b, e, gate in self.plan:
            if gate is None:
                if not waited_all and wait_all is not None:
                    wait_all()
                waited_all = True
            elif wait_gate is not None:
                wait_gate(gate)
            dist.all_reduce(flat_g[b:e], op=dist.ReduceOp.SUM, group=self.group)


class FlatParams:
    """Re-homes all parameters of a module into one flat fp32 buffer (parameters become views), so that the
    optimizer, the EMA and the gradient all-reduce work on contiguous memory."""

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.names = [n for n, _ in module.named_parameters()]
        params = [p for _, p in module.named_parameters()]
        self.sizes = [p.numel() for p in params]
        self.offsets = [0]
        for s in self.sizes:
            self.offsets.append(self.offsets[-1] + s)
        dev = params[0].device
        self.flat = torch.empty(self.offsets[-1], dtype=torch.float32, device=dev)
        for p, o, s in zip(params, self.offsets, self.sizes):
            self.flat[o:o + s].copy_(p.detach().reshape(-1))
            p.data = self.flat[o:o + s].view_as(p)

    def span(self, prefix: str):
        """[begin, end) of the parameters whose name starts with `prefix` (they are contiguous)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), f"parameters under {prefix} are not contiguous"
        return self.offsets[idx[0]], self.offsets[idx[-1] + 1]


class DPTrainer:
    """Native data-parallel train step around a `bsi_amd.BSI` whose model is a `bsi_amd.models.dit.DenoisingDiT` (per-block
    gradient buckets overlapped with the backward) or a `bsi_amd.models.vdm_unet.DenoisingVDMUNet` (28 M parameters: one
    bucket after the backward)."""

    def __init__(self, bsi, *, lr: float = 5e-4, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: float | None = 1.0, ema: bool = True, ema_beta: float = 0.9999,
                 ema_update_after_step: int = 1000, lr_schedule=None, process_group=None, force_exchange: bool = False,
                 cu_reserve: int | None = None):
        self.bsi = bsi
        self.model = bsi.model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.lr_schedule = lr_schedule  # callable step -> lr, or None for constant lr
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_exchange runs the gradient exchange (events, side stream, RCCL calls) even in a group of one rank, so that
        # the multi-GPU code path can be exercised on a single device
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        # Compute units left to the RCCL kernels for the length of a step.  The GEMM / weight-gradient / attention / convolution
        # kernels of the backward are persistent: one 160-KB-LDS workgroup per CU with a STATIC share of the tiles.  An all-reduce
        # kernel of `_exchange` that holds c CUs while a bucket is in flight would leave c of those workgroups unplaced until the
        # others finish -- the launch takes up to twice as long.  With the reserve the grids are `CUs - reserve` wide and the
        # communication kernels find their CUs free (DDP's reducer in the reference simply shares the GPU, bsi/tasks/bsi.py:163-166).
        # Default: BSI_DP_CU_RESERVE or 16 when the exchange is on (bench.py caps RCCL's channels to the same number), else 0.
        if cu_reserve is None:
            cu_reserve = int(os.environ.get("BSI_DP_CU_RESERVE", "16")) if self.exchange else 0
        self.cu_reserve = int(cu_reserve)
        # measurement hook (bench.py): with time_stages on, every step appends three HIP events (start, after backward + exchange,
        # after the optimizer) to stage_events; `stage_ms()` turns them into (forward + backward + exchange, optimizer) milliseconds
        self.time_stages = False
        self.stage_events = []
        self.step_count = 0
        self.ema_beta, self.ema_after = ema_beta, ema_update_after_step
        self._invalidate(self.model)
        self.model._ws = None
        # (`_flat_grad_only` -- the HIP backward leaves its flat gradient in `_last_flat_grad` and hands autograd no per-parameter
        # gradients -- is set only INSIDE `_backward`: a sticky attribute would silently turn every ordinary `loss.backward()` on
        # this model, and on the deep-copied EMA model, into a no-op for `p.grad`)
        self.model._flat_grad_only = False
        for attr in ("_plan", "_plan_t"):  # persistent shadow buffers + ctypes tables (rebuilt on demand; not deep-copyable)
            if hasattr(self.model, attr):
                setattr(self.model, attr, None)
        self.ema_model = copy.deepcopy(self.model).eval().requires_grad_(False) if ema else None
        self.fp = FlatParams(self.model)
        self.ema_fp = FlatParams(self.ema_model) if ema else None
        dev = self.fp.flat.device
        self._setup_update_state(dev)
        self.bucketed = hasattr(self.model, "dit")
        depth = len(self.model.dit.blocks) if self.bucketed else 0
        self.xchg = GradExchange.for_params(self.fp, [f"dit.blocks.{i}." for i in range(depth)], process_group)
        assert self.xchg.covers(self.fp.flat.numel())
        self.last_grad_norm = None
        self._setup_exchange_state(dev, depth)

    # -- stages of a step: each is one method so that the host logic (order of stages, bucket plan, gates, 1/world) is
    #    testable with world-size-2 gloo processes that substitute the two device stages (tests/test_dp_host.py)
    def _setup_update_state(self, dev):
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sq_ws = torch.empty(N.lib().bsi_sqnorm_workspace_bytes(), dtype=torch.uint8, device=dev)

    def _setup_exchange_state(self, dev, depth):
        self.comm_stream = torch.cuda.Stream(device=dev) if self.exchange else None
        self.events = None
        if self.exchange and self.bucketed:
            self.events = [torch.cuda.Event() for _ in range(depth)]
            for e in self.events:
                e.record()  # instantiate the underlying hipEvent_t
            self._ev_arr = (C.c_void_p * depth)(*[C.c_void_p(e.cuda_event) for e in self.events])

    def _backward(self, x, generator):
        """BSI.train_loss + the HIP backward on this rank's shard: (mean loss, flat fp32 gradient in FlatParams order).
        With the exchange on, the backward records event l when block l's gradients are enqueued."""
        lib = N.lib()
        for p in self.model.parameters():
            p.grad = None
        self.model._last_flat_grad = None
        if self.exchange and self.bucketed:
            N.check(lib.bsi_dit_backward_set_events(self._ev_arr, len(self.events)))
        self.model._flat_grad_only = True
        try:
            loss = self.bsi.train_loss(x, generator).mean()
            loss.backward()
        finally:
            self.model._flat_grad_only = False
            if self.exchange and self.bucketed:
                N.check(lib.bsi_dit_backward_set_events(None, 0))
        flat_g = self.model._last_flat_grad
        assert flat_g is not None, "the HIP training engine did not run (model is not a native denoiser?)"
        return loss, flat_g

    def _gate_wait(self, l):
        self.comm_stream.wait_event(self.events[l])

    def _exchange(self, flat_g):
        """Sum the gradient over the ranks, bucket by bucket (GradExchange.plan).  On GPUs the collectives run on a side
        stream: blocks finish last-to-first and each bucket starts when its event fires, overlapping the rest of the
        backward; encoder / decoder gradients (and the UNet's single bucket) wait for the whole backward."""
        if not flat_g.is_cuda:
            self.xchg.run(flat_g, wait_gate=self._gate_wait, wait_all=None)
            return
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.comm_stream):
            self.xchg.run(flat_g, wait_gate=self._gate_wait, wait_all=lambda: self.comm_stream.wait_stream(cur))
        cur.wait_stream(self.comm_stream)

    def _update(self, flat_g, lr, ema_w):
        """Global-norm clip + AdamW + EMA on the flat buffers; `flat_g` holds the SUM over ranks, the 1/world of the
        average is folded into the kernel's gradient scale."""
        lib = N.lib()
        n = flat_g.numel()
        N.check(lib.bsi_grad_sqnorm(N.ptr(flat_g), n, N.ptr(self.sq), N.ptr(self.sq_ws), N.stream()))
        N.check(lib.bsi_clip_adamw_ema(N.ptr(self.fp.flat), N.ptr(flat_g), N.ptr(self.m), N.ptr(self.v),
                                       N.ptr(self.ema_fp.flat) if self.ema_fp else None, n, N.ptr(self.sq),
                                       float(self.max_grad_norm or 0.0), 1.0 / self.world, lr, self.betas[0],
                                       self.betas[1], self.eps, self.weight_decay, self.step_count, ema_w, N.stream()))
        self.last_grad_norm = self.sq  # squared norm of the summed gradient (device scalar)

    def stage_ms(self):
        """Mean (forward + backward + exchange, optimizer) milliseconds of the steps recorded since the last call."""
        evs, self.stage_events = self.stage_events, []
        if not evs:
            return None
        evs[-1][2].synchronize()
        a = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs)
        b = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs)
        return a, b

    # ------------------------------------------------------------------------------------------------
    def _invalidate(self, model):
        model._pack = None
        model._pack_t = None

    def train_step(self, x: torch.Tensor, generator=None) -> torch.Tensor:
        """One optimizer step on this rank's shard `x`; returns the (local) mean loss (detached)."""
        reserve = self.cu_reserve if x.is_cuda else 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if (self.time_stages and x.is_cuda) else None
        if ev:
            ev[0].record()
        if reserve:
            N.check(N.lib().bsi_set_cu_reserve(reserve))  # grids are sized at launch: in effect for the kernels enqueued below
        try:
            loss, flat_g = self._backward(x, generator)
            if self.exchange:
                self._exchange(flat_g)
        finally:
            if reserve:
                N.check(N.lib().bsi_set_cu_reserve(0))    # sampling / evaluation between steps use every CU
        lr = self.lr_schedule(self.step_count) if self.lr_schedule is not None else self.lr
        w = ema_weight(self.step_count, beta=self.ema_beta, update_after_step=self.ema_after) if self.ema_fp else -1.0
        self.step_count += 1
        if ev:
            ev[1].record()
        self._update(flat_g, lr, w)
        if ev:
            ev[2].record()
            self.stage_events.append(ev)
        self._invalidate(self.model)
        if self.ema_model is not None:
            self._invalidate(self.ema_model)
        return loss.detach()

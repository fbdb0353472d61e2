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


SEG_ALIGN = 4            # elements: every slice of the flat buffers starts and ends on a 16-byte boundary (float4 kernels, RCCL)
SQNORM_CHUNK = 16384     # = BSI_SQNORM_CHUNK (include/bsi_hip.h): elements per partial of the squared gradient norm


class GradExchange:
    """Gradient exchange of one optimizer step over the flat gradient buffer — the ONE code path `DPTrainer.train_step`
    runs when world > 1, written device-agnostically (RCCL on GPUs, gloo in the CPU tests).

    `plan` is the ordered list of buckets `(begin, end, gate)`: the buckets are exchanged in this order; `gate` is the
    index of the backward event the bucket has to wait for (block l of the DiT: its gradients are complete when the backward
    has enqueued block l), or None for buckets that need the whole backward (patch encoder, decoder, or the single bucket
    of a model without per-block events).  Replaces DDP's bucketed reducer for this path (bsi/tasks/bsi.py:163-166:
    `DistributedDataParallel(model, static_graph=True)`); the 1/world of DDP's average is applied by the fused optimizer
    kernel (`bsi_clip_adamw_ema_segments`, grad_scale), not here.

    Two forms of the exchange over the same plan (`world` = the number of slices a bucket is cut into; every bucket length is a
    multiple of world * SEG_ALIGN):
      * `run`: sum-all-reduce of every bucket (what DDP does);
      * `run_reduce_scatter` + `run_all_gather`: rank r receives only the sum of slice r of every bucket (into its compact
        shard buffer), updates that slice of the parameters, and the updated slices are all-gathered (ZeRO-1 style; the same
        bytes on the wire as the all-reduce, the optimizer pass divided by world)."""

    def __init__(self, plan, group=None, world=1):
        self.plan = [(int(b), int(e), g) for b, e, g in plan if e > b]
        self.group = group
        self.world = int(world)
        for b, e, _ in self.plan:
            assert (e - b) % (self.world * SEG_ALIGN) == 0 and b % SEG_ALIGN == 0, "bucket is not a whole number of aligned slices"
        # slice geometry: bucket k is cut into `world` slices of s_k elements; slice (k, r) owns chunks
        # [chunk0_k + r * nch_k, + nch_k) of the squared-norm partial array and, on rank r, elements [goff_k, + s_k) of the shard buffer
        self.layout, chunk, goff = [], 0, 0
        for b, e, _ in self.plan:
            sl = (e - b) // self.world
            nch = -(-sl // SQNORM_CHUNK)
            self.layout.append((b, sl, nch, chunk, goff))
            chunk += nch * self.world
            goff += sl
        self.total_chunks, self.shard_elems = chunk, goff

    @classmethod
    def for_params(cls, fp, block_prefixes, group=None, world=1):
        """Plan for a `FlatParams` layout: one bucket per prefix of `block_prefixes` (in forward order; exchanged LAST block
        first, as the backward completes them, gate = block index), then everything before the first block and everything
        after the last one as two ungated buckets.  With no block prefixes: one ungated bucket over the whole buffer.
        Bucket boundaries are the block boundaries rounded UP to a multiple of world * SEG_ALIGN elements, so that every
        bucket is a whole number of aligned slices: bucket l then ends a few elements inside block l + 1 (complete before
        block l, the backward runs last-to-first) and leaves the first few elements of block l to bucket l - 1 (gated later)."""
        n = fp.flat.numel()
        a = world * SEG_ALIGN
        assert n % a == 0, "FlatParams must be padded to a multiple of world * SEG_ALIGN"
        if not block_prefixes:
            return cls([(0, n, None)], group, world)
        spans = [fp.span(pre) for pre in block_prefixes]
        for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
            assert e0 == b1, "block parameter spans must be adjacent in the flat buffer"
        up = lambda x: min(n, -(-x // a) * a)  # noqa: E731
        cuts = [up(b) for b, _ in spans] + [up(spans[-1][1])]
        plan = [(cuts[l], cuts[l + 1], l) for l in reversed(range(len(spans)))]
        plan += [(0, cuts[0], None), (cuts[-1], n, None)]
        return cls(plan, group, world)

    def covers(self, n):
        """True when the buckets tile [0, n) exactly once."""
        pos = 0
        for b, e, _ in sorted(self.plan):
            if b != pos:
                return False
            pos = e
        return pos == n

    def segments(self, ranks, compact):
        """Rows (p_off, g_off, len, my_chunk, out_chunk) of the `bsi_seg` table for the slices of `ranks` (an iterable of
        slice indices; range(world) = the whole buffer) in plan order.  compact: the gradient of slice (k, r) lies at goff_k of
        the rank's shard buffer (reduce-scatter output; one rank only) instead of at its place in the flat gradient."""
        ranks = list(ranks)
        assert not compact or len(ranks) == 1
        rows, mine = [], 0
        for b, sl, nch, chunk0, goff in self.layout:
            for r in ranks:
                off = b + r * sl
                rows.append((off, goff if compact else off, sl, mine, chunk0 + r * nch))
                mine += nch
        return rows, mine

    def _gated(self, wait_gate, wait_all):
        waited_all = False
        for k, (b, e, gate) in enumerate(self.plan):
            if gate is None:
                if not waited_all and wait_all is not None:
                    wait_all()
                waited_all = True
            elif wait_gate is not None:
                wait_gate(gate)
            yield k, b, e

    def run(self, flat_g, wait_gate=None, wait_all=None):
        """All-reduce (sum) every bucket of `flat_g` in plan order.  `wait_gate(l)` is called before the first bucket gated
        on event l, `wait_all()` once before the first ungated bucket (on GPUs: stream waits; in a synchronous backward they
        may be None)."""
        for _, b, e in self._gated(wait_gate, wait_all):
            dist.all_reduce(flat_g[b:e], op=dist.ReduceOp.SUM, group=self.group)

    def run_reduce_scatter(self, flat_g, shard, wait_gate=None, wait_all=None):
        """Reduce-scatter (sum) every bucket of `flat_g` in plan order: this rank's slice of bucket k arrives at
        shard[goff_k : goff_k + s_k]."""
        for k, b, e in self._gated(wait_gate, wait_all):
            _, sl, _, _, goff = self.layout[k]
            dist.reduce_scatter_tensor(shard[goff:goff + sl], flat_g[b:e], op=dist.ReduceOp.SUM, group=self.group)

    def forward_order(self):
        """Bucket indices in FORWARD order (front bucket, blocks first to last, tail)."""
        return sorted(range(len(self.plan)), key=lambda k: self.plan[k][0])

    def run_all_gather(self, flat, rank, after_bucket=None, communicate=True):
        """All-gather the slices of every bucket of `flat` in place (slice r of a bucket comes from rank r), buckets in FORWARD
        order (front bucket, blocks first to last, tail) -- the order in which the next step's forward needs them.
        `after_bucket(k)` is called when bucket k's collective has been enqueued (the overlapped step records an event there);
        communicate=False walks the buckets without collectives (a one-GPU rehearsal of the schedule)."""
        for k in self.forward_order():
            b, sl = self.layout[k][0], self.layout[k][1]
            if communicate:
                src = flat[b + rank * sl:b + (rank + 1) * sl]
                if not flat.is_cuda:
                    src = src.clone()  # gloo: no in-place guarantee for an input that aliases the output
                dist.all_gather_into_tensor(flat[b:b + self.world * sl], src, group=self.group)
            if after_bucket is not None:
                after_bucket(k)


class FlatParams:
    """Re-homes all parameters of a module into one flat fp32 buffer (parameters become views), so that the
    optimizer, the EMA and the gradient exchange work on contiguous memory.  The buffer is padded with zeros to a multiple
    of `pad_to` elements (the pad belongs to no parameter: zero gradient, zero moments, stays zero)."""

    def __init__(self, module: torch.nn.Module, pad_to: int = 1):
        self.module = module
        self.names = [n for n, _ in module.named_parameters()]
        params = [p for _, p in module.named_parameters()]
        self.sizes = [p.numel() for p in params]
        self.offsets = [0]
        for s in self.sizes:
            self.offsets.append(self.offsets[-1] + s)
        dev = params[0].device
        self.n = self.offsets[-1]
        self.flat = torch.zeros(-(-self.n // pad_to) * pad_to, dtype=torch.float32, device=dev)
        for p, o, s in zip(params, self.offsets, self.sizes):
            self.flat[o:o + s].copy_(p.detach().reshape(-1))
            p.data = self.flat[o:o + s].view_as(p)

    def span(self, prefix: str):
        """[begin, end) of the parameters whose name starts with `prefix` (they are contiguous)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), f"parameters under {prefix} are not contiguous"
        return self.offsets[idx[0]], self.offsets[idx[-1] + 1]


def _check_reserve(r):
    if r < 0 or r % 8 or r > 64:
        raise ValueError(f"cu_reserve = {r}: must be 0 or a multiple of 8 up to 64 (bsi_set_cu_reserve)")
    return r


_TQ_CHECKED = {}


def tile_queue_self_check(device) -> bool:
    """One GEMM of the shape class the tile queue serves (K >= 512, more 256 x 256 tiles than CUs) with tickets against the static
    schedule, bit for bit; cached per device.  Leaves the queue switch as it found it (off)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key in _TQ_CHECKED:
        return _TQ_CHECKED[key]
    lib = N.lib()
    with torch.cuda.device(key):
        g = torch.Generator(dev).manual_seed(7)
        M, Nn, K = 256 * 45, 256 * 13, 512  # 585 tiles, ragged over the 8 XCDs
        a = torch.randn((M, K), device=dev, generator=g).bfloat16()
        w = torch.randn((Nn, K), device=dev, generator=g).bfloat16()
        bias = torch.randn(Nn, device=dev, generator=g)
        outs = []
        for q in (0, 1, 1):
            o = torch.zeros((M, Nn), dtype=torch.bfloat16, device=dev)
            ga = N.GemmArgs()
            ga.A, ga.W, ga.bias, ga.out = a.data_ptr(), w.data_ptr(), bias.data_ptr(), o.data_ptr()
            ga.M, ga.N, ga.K, ga.lda, ga.ldw, ga.ldo, ga.epilogue = M, Nn, K, K, K, Nn, N.EPI_BIAS_BF16
            N.check(lib.bsi_set_tile_queue(q))
            try:
                N.check(lib.bsi_gemm_bf16(C.byref(ga), N.stream()))
            finally:
                N.check(lib.bsi_set_tile_queue(0))
            outs.append(o)
        torch.cuda.synchronize(dev)
        ok = bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))
    _TQ_CHECKED[key] = ok
    return ok


class DPTrainer:
    """Native data-parallel train step around a `bsi_amd.BSI` whose model is a `bsi_amd.models.dit.DenoisingDiT` (per-block
    gradient buckets overlapped with the backward) or a `bsi_amd.models.vdm_unet.DenoisingVDMUNet` (28 M parameters: one
    bucket after the backward).

    At world > 1 the constructor broadcasts rank 0's parameters, buffers and EMA to every rank, as the constructor of
    `DistributedDataParallel` does (bsi/tasks/bsi.py:165): ranks that were seeded differently still train ONE model.

    shard_update=True: the step is reduce-scatter -> clip + AdamW + EMA on this rank's 1/world slice of every bucket -> all-gather
    of the updated parameters (the same parameters as the all-reduce step up to the order in which the collective sums the ranks:
    bit-identical on gloo and for two ranks, equal to fp32 rounding beyond -- a reduce-scatter ring and an all-reduce ring add in
    different orders; same per-chunk norm, same arithmetic).  The EMA is then updated on the owned slices only: `ema_model` raises
    until `gather_ema()` (a collective) has completed it, `ema_state_dict()` gathers first.  rehearse=(world, rank): lay the buckets and slices out as on that rank of that world size WITHOUT any
    communication -- a timing rehearsal of one rank's compute on a single GPU (bench.py); with shard_update the parameters
    outside the rank's slices are then simply not updated."""

    def __init__(self, bsi, *, lr: float = 5e-4, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: float | None = 1.0, ema: bool = True, ema_beta: float = 0.9999,
                 ema_update_after_step: int = 1000, lr_schedule=None, process_group=None, force_exchange: bool = False,
                 cu_reserve: int | None = None, tile_queue: bool | None = None, shard_update: bool = False, rehearse=None,
                 overlap_gather: bool = False):
        self.bsi = bsi
        self.model = bsi.model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.lr_schedule = lr_schedule  # callable step -> lr, or None for constant lr
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # force_exchange runs the gradient exchange (events, side stream, RCCL calls) even in a group of one rank, so that
        # the multi-GPU code path can be exercised on a single device
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        self.shard_update = bool(shard_update)
        # overlap_gather (sharded step of a bucketed model): the all-gather of the updated parameters is NOT waited for at the end of
        # the step; it runs bucket by bucket in forward order on the communication stream while the next step's forward is enqueued,
        # and the forward waits block by block (bsi_dit_train_forward_set_gates) -- what DDP's own exchange gets from hiding behind
        # the backward (bsi/tasks/bsi.py:163-166).  Same arithmetic, same bits as the step that waits.
        self.overlap_gather = bool(overlap_gather) and self.shard_update
        self._gates_pending = False
        self._gate_events, self._gate_used, self._gate_arr = None, None, None
        # the world / rank the buckets and slices are laid out for (a rehearsal lays out another world's without communicating)
        self.lay_world, self.lay_rank = (int(rehearse[0]), int(rehearse[1])) if rehearse else (self.world, self.rank)
        assert not (rehearse and self.exchange), "a rehearsal does not communicate"
        assert 0 <= self.lay_rank < self.lay_world
        self.bucketed = hasattr(self.model, "dit")
        # Compute units left to the RCCL kernels while buckets are in flight.  The weight-gradient GEMMs and convolutions of the
        # backward are persistent with a STATIC share of the work per workgroup (one 160-KB-LDS workgroup per CU): a communication
        # kernel that holds c CUs leaves c of those workgroups unplaced until the others finish -- the launch takes twice as long.
        # With the reserve their grids are `CUs - reserve` wide.  It is in force only for the kernels that can run beside a bucket:
        # while the backward of a BUCKETED model is enqueued (grids are sized at launch; the forward, and a model whose single
        # bucket leaves after the backward, overlap with nothing).  Default: BSI_DP_CU_RESERVE or 0 -- no multi-GPU run has confirmed a
        # gain yet (bench.py --gpus N measures the step under both settings); RCCL's channel count must then be capped to the
        # same number BEFORE init_process_group (NCCL_MAX_NCHANNELS; bench.py does it), which is checked here.
        if cu_reserve is None:
            cu_reserve = int(os.environ.get("BSI_DP_CU_RESERVE", "0"))
        self.cu_reserve = _check_reserve(int(cu_reserve)) if (self.exchange or rehearse) and self.bucketed else 0
        if self.cu_reserve and self.exchange and bsi.model is not None and next(self.model.parameters()).is_cuda:
            ch = os.environ.get("NCCL_MAX_NCHANNELS")
            if ch is None or not ch.isdigit() or int(ch) > self.cu_reserve:
                import warnings
                warnings.warn(f"DPTrainer: cu_reserve = {self.cu_reserve} but NCCL_MAX_NCHANNELS = {ch}: RCCL may launch more "
                              "workgroups than CUs are kept free for it; export NCCL_MAX_NCHANNELS before init_process_group")
        # Tile queue (bsi_set_tile_queue): the persistent GEMMs of the backward that have more tiles than CUs draw their tiles from
        # counters instead of taking a static share, on ALL CUs whatever the reserve says -- a workgroup whose CU an RCCL kernel holds
        # leaves its share to the others, and no CU idles while RCCL is quiet (the reserve then only sizes the kernels that keep a
        # static partition: weight-gradient GEMMs, attention).  Bit-identical results.  In force for the launches of the backward of a
        # bucketed model, like the reserve.
        # Default OFF (BSI_DP_TILE_QUEUE=1 or tile_queue=True switches it on): no multi-GPU run has exercised it beside real RCCL
        # kernels yet.  Whenever it is switched on, `tile_queue_self_check` first runs one ticketed GEMM against the static schedule
        # on this device and refuses (warning, static schedule) if the bits differ -- the ticket pipeline keeps a value in a register
        # the compiler must not touch (gemm_bf16.hip TQ_DRAW; the build's lint checks the generated code, this checks the loaded one).
        if tile_queue is None:
            tile_queue = os.environ.get("BSI_DP_TILE_QUEUE", "0") == "1"
        self.tile_queue = bool(tile_queue) and (self.exchange or bool(rehearse)) and self.bucketed
        if self.tile_queue and next(self.model.parameters()).is_cuda and not tile_queue_self_check(next(self.model.parameters()).device):
            import warnings
            warnings.warn("DPTrainer: the tile queue's self-check failed on this device (ticketed GEMM != static schedule); "
                          "using the static schedule")
            self.tile_queue = False
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
        for attr in ("_plan", "_plan_t", "_grad_buffer"):  # persistent buffers + ctypes tables (rebuilt on demand; not deep-copyable)
            if hasattr(self.model, attr):
                setattr(self.model, attr, None)
        self._ema_model = copy.deepcopy(self.model).eval().requires_grad_(False) if ema else None
        pad = self.lay_world * SEG_ALIGN
        self.fp = FlatParams(self.model, pad)
        self.ema_fp = FlatParams(self._ema_model, pad) if ema else None
        self.ema_complete = True  # False while the EMA of a sharded step is current on the owned slices only
        dev = self.fp.flat.device
        if self.world > 1:
            self._broadcast_start_state()
        depth = len(self.model.dit.blocks) if self.bucketed else 0
        self.xchg = GradExchange.for_params(self.fp, [f"dit.blocks.{i}." for i in range(depth)], process_group, self.lay_world)
        assert self.xchg.covers(self.fp.flat.numel())
        self.last_grad_norm = None
        self._setup_update_state(dev)
        self._setup_exchange_state(dev, depth)

    def _broadcast_start_state(self):
        """Rank 0's parameters, buffers and EMA to every rank (DistributedDataParallel.__init__ -> _sync_module_states,
        bsi/tasks/bsi.py:165; Lightning wraps the module before the first step)."""
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(self.fp.flat, src=src, group=self.group)
        if self.ema_fp is not None:
            dist.broadcast(self.ema_fp.flat, src=src, group=self.group)
        for m in (self.model, self._ema_model):
            for b in (m.buffers() if m is not None else ()):
                dist.broadcast(b, src=src, group=self.group)

    # -- stages of a step: each is one method so that the host logic (order of stages, bucket plan, gates, slices, 1/world) is
    #    testable with world-size-2 gloo processes that substitute the device stages (tests/test_dp_host.py)
    def _seg_table(self, rows, dev):
        arr = (N.Seg * len(rows))(*[N.Seg(*r) for r in rows])
        return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    def _setup_update_state(self, dev):
        n = self.fp.flat.numel()
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        # the flat gradient the HIP backward writes (its pad stays zero) and, for the sharded step, the compact buffer the
        # reduce-scatter delivers this rank's slices into
        self.gbuf = torch.zeros(n, dtype=torch.float32, device=dev)
        self.gshard = torch.zeros(self.xchg.shard_elems, dtype=torch.float32, device=dev) if self.shard_update else None
        # squared-norm partials: one per chunk of every slice of every rank; `part_own` holds this rank's chunks and zeros
        self.part = torch.zeros(self.xchg.total_chunks, dtype=torch.float32, device=dev)
        self.part_own = torch.zeros_like(self.part) if self.shard_update else None
        ranks = [self.lay_rank] if self.shard_update else range(self.lay_world)
        self.seg_rows, self.seg_chunks = self.xchg.segments(ranks, compact=self.shard_update)
        self.seg_tab = self._seg_table(self.seg_rows, dev) if dev.type == "cuda" else None

    def _setup_exchange_state(self, dev, depth):
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.exchange or (self.overlap_gather and dev.type == "cuda")) else None
        self.events = None
        if self.exchange and self.bucketed:
            self.events = [torch.cuda.Event() for _ in range(depth)]
            for e in self.events:
                e.record()  # instantiate the underlying hipEvent_t
            self._ev_arr = (C.c_void_p * depth)(*[C.c_void_p(e.cuda_event) for e in self.events])

    def _backward(self, x, generator):
        """BSI.train_loss + the HIP backward on this rank's shard: (mean loss, flat fp32 gradient in FlatParams order).
        With the exchange on, the backward records event l when block l's gradients are enqueued.  The CU reserve and the tile queue
        are in force for the launches of the backward only."""
        lib = N.lib()
        for p in self.model.parameters():
            p.grad = None
        self.model._last_flat_grad = None
        self.model._grad_buffer = self.gbuf
        if self.exchange and self.bucketed:
            N.check(lib.bsi_dit_backward_set_events(self._ev_arr, len(self.events)))
        self.model._flat_grad_only = True
        try:
            gated = self._install_gates()
            try:
                loss = self.bsi.train_loss(x, generator).mean()
            finally:
                if gated:
                    N.check(lib.bsi_dit_train_forward_set_gates(None, 0))
            # grids are sized when a kernel is LAUNCHED: the reserve and the queue apply to the launches of the backward only (the
            # forward above overlaps with nothing; the exchange only launches RCCL's own kernels).  They are per launching thread and
            # the backward runs on autograd's thread: the training engine applies `_bwd_sched` around its own launches
            # (models/dit_train.py), so nothing else in the process -- an evaluation on another thread or stream -- is affected.
            self.model._bwd_sched = (self.cu_reserve, self.tile_queue)
            loss.backward()
        finally:
            self.model._flat_grad_only = False
            self.model._grad_buffer = None
            self.model._bwd_sched = None
            if self.exchange and self.bucketed:
                N.check(lib.bsi_dit_backward_set_events(None, 0))
        flat_g = self.model._last_flat_grad
        assert flat_g is not None, "the HIP training engine did not run (model is not a native denoiser?)"
        assert flat_g.data_ptr() == self.gbuf.data_ptr(), "the training engine did not use the trainer's gradient buffer"
        return loss, self.gbuf

    def _gate_wait(self, l):
        self.comm_stream.wait_event(self.events[l])

    # -- overlapped parameter all-gather (sharded step) --------------------------------------------------------------------------
    def _gate_of_bucket(self, k):
        """Forward gate of plan bucket k: 0 = front (patch encoder), 1 + l = block l, depth + 1 = tail (decoder)."""
        b, _, gate = self.xchg.plan[k]
        if gate is not None:
            return 1 + gate
        return 0 if b == 0 else len(self.model.dit.blocks) + 1

    def _gather_async(self, flat):
        """Enqueue the all-gather of `flat` on the communication stream WITHOUT making the compute stream wait: one event per
        bucket, which the next forward waits for block by block.  Anything else that reads the parameters first (sampling from the
        model, a checkpoint) goes through `sync_params`, which the model's cast path calls by itself."""
        depth = len(self.model.dit.blocks)
        if self._gate_events is None:
            self._gate_events = [torch.cuda.Event() if flat.is_cuda else None for _ in range(depth + 2)]
        used = [False] * (depth + 2)

        def mark(k):
            g = self._gate_of_bucket(k)
            used[g] = True
            if flat.is_cuda:
                self._gate_events[g].record()  # on the current (= communication) stream

        if flat.is_cuda and self.comm_stream is not None:
            cur = torch.cuda.current_stream()
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                self.xchg.run_all_gather(flat, self.rank, after_bucket=mark, communicate=self.exchange)
        else:
            self.xchg.run_all_gather(flat, self.rank, after_bucket=mark, communicate=self.exchange)
        self._gate_used = used
        self._gates_pending = True
        self.model._params_pending_sync = self.sync_params

    def sync_params(self):
        """Make the current stream wait for an all-gather of the parameters that is still in flight (no-op otherwise)."""
        if self._gates_pending:
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._gates_pending = False
        self.model.__dict__["_params_pending_sync"] = None

    def _install_gates(self):
        """Before the forward of a step whose parameters are still arriving: hand the per-bucket events and the per-part cast
        sub-tables to the training engine (the forward then casts block l's shadows behind block l's event).  False when nothing is
        pending or the model has no gate tables (then the ordinary cast path waits for the whole gather)."""
        if not self._gates_pending:
            return False
        plan = self.model.native_pack_uncast() if hasattr(self.model, "native_pack_uncast") else None
        tabs = plan.get("gate_tables") if plan is not None else None
        if tabs is None or plan.get("pack_t") is None:
            self.sync_params()
            self._invalidate(self.model)
            return False
        raw, spans = tabs
        n = len(spans)
        arr = (N.FwdGate * n)()
        desc = C.sizeof(N.CastDesc)
        for gi, (pos, cnt, tiles) in enumerate(spans):
            ev = self._gate_events[gi] if (self._gate_events and self._gate_used[gi] and self._gate_events[gi] is not None) else None
            arr[gi] = N.FwdGate(ev.cuda_event if ev is not None else None, raw.data_ptr() + pos * desc if cnt else None, cnt, tiles)
        self._gate_arr = arr  # stays alive until cleared
        N.check(N.lib().bsi_dit_train_forward_set_gates(arr, n))
        self._gates_pending = False
        self.model.__dict__["_params_pending_sync"] = None
        return True

    def _exchange(self, flat_g):
        """Sum the gradient over the ranks, bucket by bucket (GradExchange.plan): all-reduce, or reduce-scatter into this rank's
        shard buffer.  On GPUs the collectives run on a side stream: blocks finish last-to-first and each bucket starts when its
        event fires, overlapping the rest of the backward; encoder / decoder gradients (and the UNet's single bucket) wait for the
        whole backward."""
        run = (lambda **kw: self.xchg.run_reduce_scatter(flat_g, self.gshard, **kw)) if self.shard_update else \
              (lambda **kw: self.xchg.run(flat_g, **kw))
        if not flat_g.is_cuda:
            run(wait_gate=self._gate_wait, wait_all=None)
            return
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.comm_stream):
            run(wait_gate=self._gate_wait, wait_all=lambda: self.comm_stream.wait_stream(cur))
        cur.wait_stream(self.comm_stream)

    # device stages of the update (CPU stand-ins in tests/test_dp_host.py follow the same contracts)
    def _sq_partials(self, g, out):
        """out[chunk] = sum of squares of every chunk of this trainer's segments of gradient buffer `g`."""
        N.check(N.lib().bsi_sqnorm_segments(N.ptr(g), N.ptr(self.seg_tab), len(self.seg_rows), self.seg_chunks, N.ptr(out), N.stream()))

    def _sq_finish(self, part):
        N.check(N.lib().bsi_sqnorm_finish(N.ptr(part), part.numel(), N.ptr(self.sq), N.stream()))

    def _apply(self, g, lr, ema_w):
        """clip + AdamW + EMA on this trainer's segments; `g` holds the SUM over ranks, the 1/world of the average is folded into
        the kernel's gradient scale."""
        N.check(N.lib().bsi_clip_adamw_ema_segments(
            N.ptr(self.fp.flat), N.ptr(g), N.ptr(self.m), N.ptr(self.v), N.ptr(self.ema_fp.flat) if self.ema_fp else None,
            N.ptr(self.seg_tab), len(self.seg_rows), self.seg_chunks, N.ptr(self.sq), float(self.max_grad_norm or 0.0),
            1.0 / self.lay_world, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, ema_w, N.stream()))

    def _update(self, flat_g, lr, ema_w):
        """Global-norm clip + AdamW + EMA.  All-reduce step: on the whole flat buffers.  Sharded step: the norm's partials of the
        owned slices are summed over the ranks (the other ranks' entries are exact zeros), the update runs on the owned slices
        and the updated parameter slices are all-gathered."""
        if not self.shard_update:
            self._sq_partials(flat_g, self.part)
            self._sq_finish(self.part)
            self._apply(flat_g, lr, ema_w)
        else:
            self._sq_partials(self.gshard, self.part_own)
            self.part.copy_(self.part_own)
            if self.exchange:
                dist.all_reduce(self.part, op=dist.ReduceOp.SUM, group=self.group)
            self._sq_finish(self.part)
            self._apply(self.gshard, lr, ema_w)
            self.ema_complete = self.ema_fp is None or ema_w < 0
            if self.overlap_gather and self.bucketed and (self.exchange or self.lay_world > 1):
                self._gather_async(self.fp.flat)
            elif self.exchange:
                self._gather(self.fp.flat)
        self.last_grad_norm = self.sq  # squared norm of the summed gradient (device scalar)

    def _gather(self, flat):
        if not flat.is_cuda:
            self.xchg.run_all_gather(flat, self.rank)
            return
        cur = torch.cuda.current_stream()
        self.comm_stream.wait_stream(cur)
        with torch.cuda.stream(self.comm_stream):
            self.xchg.run_all_gather(flat, self.rank)
        cur.wait_stream(self.comm_stream)

    def gather_ema(self):
        """Sharded step: all-gather the EMA slices so that `ema_model` is complete on every rank (no-op otherwise)."""
        if self.shard_update and self.exchange and self.ema_fp is not None and not self.ema_complete:
            self._gather(self.ema_fp.flat)
            self._invalidate(self._ema_model)
        self.ema_complete = True

    @property
    def ema_model(self):
        """The EMA copy of the model (ema_pytorch's `ema_model`, bsi/tasks/ema_pytorch.py:203-217) -- always WHOLE, as there.  After
        sharded steps across ranks each rank's copy is current only on the slices it owns: reading it then raises instead of
        handing out a mix of current and stale slices; `gather_ema()` (a collective: every rank calls it) completes it."""
        if self._ema_model is not None and self.shard_update and self.exchange and not self.ema_complete:
            raise RuntimeError("DPTrainer.ema_model: after sharded steps the EMA is complete only on this rank's slices; call "
                               "gather_ema() on every rank (or ema_state_dict(), which does) before sampling from or saving it")
        return self._ema_model

    def ema_state_dict(self):
        """`ema_model.state_dict()` of a COMPLETE EMA (gathers first; every rank calls it)."""
        self.gather_ema()
        return None if self._ema_model is None else self._ema_model.state_dict()

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
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if (self.time_stages and x.is_cuda) else None
        if ev:
            ev[0].record()
        loss, flat_g = self._backward(x, generator)
        if self.exchange:
            self._exchange(flat_g)
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
        if self._ema_model is not None:
            self._invalidate(self._ema_model)
        return loss.detach()

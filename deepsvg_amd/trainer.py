"""Data-parallel training step for the MI355X build: the body of deepsvg/train.py:92-106
(zero_grad, forward, SVGLoss, backward, clip_grad_norm_, AdamW.step) as one static sequence of HIP launches on
flat buffers, one process per GPU.

  * parameters, gradients and Adam moments each live in ONE contiguous fp32 buffer (ParamStore), so the
    gradient exchange is one RCCL all-reduce (10.3 M floats = 41 MB over xGMI, replacing nn.DataParallel's
    broadcast/scatter/gather/reduce_add of deepsvg/train.py:74), the global-norm clip is one reduction and
    AdamW is one launch;
  * the per-rank mean losses are re-normalised by the GLOBAL selected-element counts (three scalars), so the
    averaged gradient equals the gradient of the global-batch mean exactly as in single-process training
    (SURVEY.md §8(e));
  * every scalar the kernels need (lr, step, seed, norm, counts) lives in device memory, so the whole step can
    be captured in a hipGraph and replayed (use_graph=True) to remove the host launch overhead (~12 ms of Python
    dispatch per step, as long as the GPU work itself).  The two data-dependent layouts of the model (packed
    encoder rows, visible-first decoder prefix) are made static per graph by rounding them up to buckets - extra
    rows / sequences are inert by construction - and one graph is captured per bucket pair on first use; the layout
    plan (a few tiny kernels + one host read) runs eagerly before each replay and is copied into the graph's buffers.
"""
import collections
import os
import time

import torch
import torch.distributed as dist

from . import ops
from .svgtensor import EOS_ID

DEFAULT_WEIGHTS = {   # configs/deepsvg/default_icons.py:65-73 at step 0
    "kl_tolerance": 0.1, "loss_kl_weight": 0.0, "loss_hierarch_weight": 1.0, "loss_cmd_weight": 1.0,
    "loss_args_weight": 2.0, "loss_visibility_weight": 1.0,
}


# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a stream that waits on an event
# holds up every other stream of its queue.  The data-parallel hipGraph step uses five (main, layout plan, loss counts, RCCL's,
# torch's copy stream): with four queues the plan stream ended up behind RCCL's wait for the previous step's graph, the host's
# read of the plan blocked for a whole step and could never run ahead of the GPU (one-rank RCCL group: 7.19 ms/step, 6.85 with
# six queues against 6.71 for the single-GPU step).  The variable only takes effect when it is set before the HIP runtime
# initialises, so it is a LAUNCH setting of the training process (bench.py sets it; `GPU_MAX_HW_QUEUES=6 python train.py`),
# not something this library changes behind the caller's back.
HW_QUEUES_NOTE = ("data-parallel hipGraph step: export GPU_MAX_HW_QUEUES=6 BEFORE the process makes its first device call - the "
                  "runtime reads it once, when it initialises - (with the default of 4 hardware queues the host cannot run ahead "
                  "of the device, ~+7 % per step)")

_SHARED_STREAMS = {}


def _shared_stream(role, device):
    """the plan / count stream of a device: ONE per process and role, shared by every TrainStep.  ROCm maps HIP streams onto
    GPU_MAX_HW_QUEUES hardware queues in creation order; a trainer created later in the process (a second model, a
    validation trainer) would otherwise draw streams further down torch's pool, and a plan stream that lands on the main
    stream's hardware queue serialises the plan behind the previous step - the host can no longer run ahead of the GPU
    (scripts/secondary_bench.py: the second TrainStep of the process ran 29.3 -> 36.8 ms/step)."""
    key = (role, torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class TrainStep:
    def __init__(self, model, loss_fn, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_clip=1.0,
                 weights=None, process_group=None, use_graph=False, exact_global_mean=True, force_ddp=False):
        self.model, self.loss_fn = model, loss_fn
        self.betas, self.eps, self.weight_decay, self.grad_clip = betas, eps, weight_decay, grad_clip
        self.weights = dict(weights or DEFAULT_WEIGHTS)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # force_ddp: run the data-parallel code path (collectives included) on a one-rank group - how the single-GPU
        # development boxes exercise RCCL and the graph / collective / optimiser split below
        self.ddp = self.world > 1 or (force_ddp and dist.is_available() and dist.is_initialized())
        self.use_graph = use_graph
        if self.ddp and use_graph and "GPU_MAX_HW_QUEUES" not in os.environ:
            import warnings
            late = torch.cuda.is_available() and torch.cuda.is_initialized()
            warnings.warn(HW_QUEUES_NOTE + (" - the HIP runtime of THIS process is already initialised: setting the variable now "
                                            "has no effect, restart the process with it exported" if late else ""))
        self.exact_global_mean = exact_global_mean and self.ddp
        self._counts = None             # graph + DDP: the global loss counts, filled before every replay
        self._in_own_step = False       # True only while this trainer's captured / replayed step is being enqueued
        self._lr_value = float(lr)
        self._ready = False
        # bucket key -> (graph, static inputs, static plan, static results), least recently used first.  Real data lands in
        # tens of (packed rows, visible sequences, loss rows, slot range) buckets; each graph keeps its static tensors and
        # its share of the pool alive, so the cache is bounded: beyond `max_graphs` the least recently used one is dropped
        # (its bucket is simply captured again if it ever comes back).
        self._graphs = collections.OrderedDict()
        self.max_graphs = int(os.environ.get("DSVG_MAX_GRAPHS", "24"))
        self.graphs_captured = 0        # statistics: captures so far (re-captures after an eviction included)
        self.graphs_evicted = 0
        self._plan_stream = None
        self.inputs_resident = False    # see step(): set by callers whose input tensors are complete well before step()
        # gradient all-reduce in two buckets, the decoder's overlapped with the encoder's backward (eager launches only)
        self.overlap_allreduce = os.environ.get("DSVG_DDP_OVERLAP", "1") != "0"
        # (rounds 3-4: the step captured as TWO hipGraphs at the bottleneck, the decoder bucket's all-reduce in flight between
        # them, was measurable on a one-rank RCCL group only - 6.90 against 6.85 ms/step for the single graph, a 15 ms/step
        # pathology with GPU_MAX_HW_QUEUES=8 - and is removed: with hipGraph the gradient goes out in one all-reduce behind
        # the graph; the eager path keeps the overlapped decoder bucket)
        # round 5: DSVG_DDP_BF16=1 - the gradient travels as bf16 (20.6 MB instead of 41.2 MB): cast, all-reduce,
        # cast back; off by default until a run on more than one GPU exists (no hardware claim).
        # (Tried and not possible on this stack: sending the decoder bucket out early in hipGraph mode too, from an EXTERNAL
        # event-record node inside the captured backward that a side stream waits for behind every replay - HIP 7.0 answers
        # hipEventRecordWithFlags(..., hipEventRecordExternal) on a capturing stream with "invalid argument", and
        # torch.cuda.Event(external=True) refuses on ROCm for the same reason.  In graph mode the gradient therefore still goes
        # out in one all-reduce behind the graph; the eager path keeps the overlapped decoder bucket.)
        self.allreduce_bf16 = os.environ.get("DSVG_DDP_BF16", "0") == "1"
        self.time_allreduce = False         # record HIP events around the gradient all-reduce of every step (allreduce_events)
        self.allreduce_events = []
        self._bf16_stage = None
        self._pending = None
        self._pool = None
        self._gradless_slots, self._gradless_ids = [], set()
        # DSVG_TRACE_STEP=1: host-side time stamps of every step (entry, plan read done, graph launched) in `host_trace`
        self.host_trace = [] if os.environ.get("DSVG_TRACE_STEP") == "1" else None
        self._count_stream = None
        self.row_bucket, self.seq_bucket = 1024, 64
        self.defer_reductions = os.environ.get("DSVG_DEFER_REDUCE", "1") != "0"
        model._own_seed = False          # the trainer advances the dropout seed once per step
        if self.exact_global_mean:
            # (a loss_fn serves one data-parallel trainer at a time: a newer TrainStep takes the reducer over)
            loss_fn.count_reducer = self._reduce_counts
        elif getattr(loss_fn, "count_reducer", None) is not None and not self.ddp:
            loss_fn.count_reducer = None    # a single-process trainer must not inherit another trainer's reducer

    # ---- lazily created device state -----------------------------------------------------------------
    def _setup(self, device):
        model = self.model
        model.store.ensure(device, model.compute_dtype)
        n = model.store.flat.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device)
        self.lr = torch.full((1,), self._lr_value, dtype=torch.float32, device=device)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=device)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=device)
        self.seed = model.seed_tensor(device)
        self._reset_gradless()
        self._ready = True

    def _reset_gradless(self):
        """forget which parameters had no gradient (keyed by id(p) / views of the flat gradient buffer): called when the
        device state is created and whenever the ParamStore re-flattened.  The cached graphs go too: each baked in the slot set
        and the buffers of its capture"""
        self._gradless_slots, self._gradless_ids = [], set()
        self._store_generation = getattr(self.model.store, "generation", 0)
        if self._graphs:
            self._graphs.clear()

    def set_lr(self, lr):
        self._lr_value = float(lr)
        if self._ready:
            self.lr.fill_(self._lr_value)

    class _Bf16Work:
        """an all-reduce that travels as bf16: `view` (fp32 slice of the flat gradient) was cast into `stage`, which is being
        reduced; wait() = wait for the collective, cast back"""
        def __init__(self, view, stage, work, world=1.0):
            self.view, self.stage, self.work, self.world = view, stage, work, world

        def wait(self):
            if self.work is not None:
                self.work.wait()
            self.view.copy_(self.stage)
            if self.world != 1.0:
                self.view.mul_(self.world)

    def _all_reduce(self, view, async_op=False):
        """sum-all-reduce of a slice of the flat fp32 gradient buffer (in place) -> a work object with wait(), or None"""
        if not self.allreduce_bf16:
            return dist.all_reduce(view, group=self.pg, async_op=async_op)
        flat_g = self.model.store.grad_buffer(0)
        if self._bf16_stage is None or self._bf16_stage.numel() != flat_g.numel() or self._bf16_stage.device != flat_g.device:
            self._bf16_stage = torch.empty(flat_g.numel(), dtype=torch.bfloat16, device=flat_g.device)
        off = view.storage_offset() - flat_g.storage_offset()
        stage = self._bf16_stage[off:off + view.numel()]
        # (the rank AVERAGE travels as bf16, not the rank sum: the 1 / world scale sits in front of the lossy cast - exact for
        # power-of-two worlds - and is undone behind the cast back, so the flat buffer keeps its "sum over ranks" meaning for
        # the norm, the clip and AdamW's grad_scale)
        torch.mul(view, 1.0 / self.world, out=stage)
        w = TrainStep._Bf16Work(view, stage, dist.all_reduce(stage, group=self.pg, async_op=async_op), float(self.world))
        if async_op:
            return w
        w.wait()
        return None

    def _launch_decoder_bucket(self):
        """called from the backward pass when every decoder gradient is final (model.forward registers the hook)"""
        lo, hi = self.model.decoder_param_range()
        ops.flush_deferred()                # the queued split-K / LayerNorm reductions of the decoder's gradients
        flat_g = self.model.store.grad_buffer(0)
        self._pending = (lo, self._all_reduce(flat_g[lo:hi], async_op=True))

    def _reduce_counts(self, counts):
        """[n] local selected-element counts of the cross-entropies -> global counts / world, in ONE all-reduce"""
        if self._in_own_step and self.use_graph and self._counts is not None:
            # hipGraph + DDP, inside this trainer's own captured step: the counts were all-reduced before the replay
            # into a static tensor (see _global_counts).  Every other caller of the loss - a validation loss, a metric,
            # another trainer - gets a real all-reduce of ITS counts, exactly as in eager mode.
            return self._counts
        dist.all_reduce(counts, group=self.pg)
        return counts / self.world

    def _global_counts(self, commands_dec, args_dec, plan):
        """hipGraph + DDP: the cross-entropy normalisers depend on the targets only (deepsvg/model/loss.py:33-57), so their
        one 3-element all-reduce runs eagerly BEFORE the captured forward + backward and lands in a static tensor"""
        N, G = commands_dec.shape[0], commands_dec.shape[1]
        if plan["loss"] is not None:
            targets = plan["loss"]["targets"]
        else:
            tc = commands_dec.to(torch.float32).contiguous().view(N * G, -1)
            ta = args_dec.to(torch.float32).contiguous().view(N * G, tc.shape[1], -1)
            targets = ops.loss_targets(tc, ta, self.loss_fn._cam(tc.device), EOS_ID)
        local = torch.stack([torch.full((), float(N * G), device=commands_dec.device),
                             (targets[1] != 0).sum().float(), (targets[3] != 0).sum().float()])
        dist.all_reduce(local, group=self.pg)
        local /= self.world
        return local

    # ---- one step ------------------------------------------------------------------------------------
    def _step_body(self, commands, args, label=None, dec=None):
        ld = self._step_front(commands, args, label, dec)
        self._step_back()
        return ld

    def _seed_grad(self, loss):
        g = getattr(self, "_one", None)
        if g is None or g.device != loss.device or g.dtype != loss.dtype:
            g = self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        return g

    def _step_front(self, commands, args, label=None, dec=None):
        """forward, loss, backward: everything up to the (local) gradient in the flat buffer.
        dec = (commands_dec, args_dec) when the decoder side takes other tensors than the encoder side (relative
        targets: model_args = [commands, args, commands, args_rel], deepsvg/model/config.py:52-53)"""
        model = self.model
        cd, ad = dec if dec is not None else (commands, args)
        model.store.pending_advance = (self.step_count, self.seed)    # launched with the weight images of the forward below
        for p in model.store.params:
            p.grad = None
        self._pending = None
        # the norm / AdamW / all-reduce read the WHOLE flat gradient buffer: a parameter that receives no gradient in a step
        # must contribute zero, not its gradient of an earlier step.  Which parameters those are is known from the previous
        # step (no shipped config has one); their slots are zeroed BEFORE backward, i.e. before an overlapped decoder
        # all-reduce can be in flight over them (and inside the captured part of a graph step)
        if self.ddp and self.overlap_allreduce and not self.use_graph:
            model._decoder_grads_ready = self._launch_decoder_bucket    # hooked onto the bottleneck output in forward
        # the ~130 partial-sum reductions of the parameter gradients (split-K slices, LayerNorm gamma/beta partials) are
        # queued during backward and performed by ONE launch per 64 right after it (ops.flush_deferred): nothing reads a
        # gradient in between (the overlapped decoder bucket flushes first, see _launch_decoder_bucket)
        model._defer_wgrad = self.defer_reductions
        try:
            out = model(commands, args, cd, ad, label=label, params={})
            if getattr(model.store, "generation", 0) != getattr(self, "_store_generation", 0):
                self._reset_gradless()  # the forward re-flattened the store: ids and gradient views cached here are stale
            for v in self._gradless_slots:      # (before backward: nothing is in flight over the gradient buffer yet)
                v.zero_()
            ld = self.loss_fn(out, label, weights=self.weights)
            ld["loss"].backward(self._seed_grad(ld["loss"]))      # (a static 1: autograd would launch a fill for its own)
        finally:
            model._decoder_grads_ready = None
            model._defer_wgrad = False
            # (a forward that raised before ParamStore.ensure() consumed it must not leave the advance behind for the next,
            # unrelated forward - an evaluation, a sampling call - to apply)
            model.store.pending_advance = None
            rt = getattr(model, "_rt", None)
            if rt is not None:
                rt.stack_group_end()        # (a stack's grouped weight-gradient launch left open by an interrupted backward)
            ops.flush_deferred()
        # parameters without a gradient in THIS call (torch's AdamW skips them; which ones can depend on the call shape - with /
        # without label, relative targets): checked on every eager step and inside every capture (a replay runs no Python).
        # Slots found for the first time are zeroed now, after everything that may be in flight over the buffer has been
        # waited for, and from the next call on before backward (above)
        new = [p for p in model.store.params if p.grad is None and p.requires_grad and id(p) not in self._gradless_ids]
        if new:
            if self._pending is not None:
                self._pending[1].wait()
            for p in new:
                self._gradless_ids.add(id(p))
                v = model.store._grad_view(p, 0)
                if v is not None:
                    v.zero_()
                    self._gradless_slots.append(v)
        return {k: v.detach() for k, v in ld.items()}

    def _step_back(self):
        """gradient all-reduce (data parallel), global-norm clip and AdamW on the flat buffers (train.py:99-106)"""
        model = self.model
        flat_g = model.store.grad_buffer(0)
        if self.ddp:
            ev = None
            if self.time_allreduce and flat_g.is_cuda:      # (bench.py: the exposed time of the gradient exchange, `ddp.allreduce_ms`)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self._pending is not None:
                # two buckets: the decoder half went out while the encoder's backward was running
                lo, work = self._pending
                self._all_reduce(flat_g[:lo])
                work.wait()
            else:
                self._all_reduce(flat_g)
            if ev is not None:
                ev[1].record()
                self.allreduce_events.append(ev)
                del self.allreduce_events[:-256]
        ops.sumsq(flat_g, out=self.gnorm_sq)
        ops.adamw_step_(model.store.flat, flat_g, self.m, self.v, self.lr, self.step_count,
                        beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay,
                        gnorm_sq=self.gnorm_sq if self.grad_clip else None, max_norm=float(self.grad_clip or 0.0),
                        grad_scale=1.0 / self.world)

    def step(self, commands, args, label=None, commands_dec=None, args_dec=None):
        """one training step on a batch (the body of deepsvg/train.py:92-106): model(commands, args, commands_dec,
        args_dec, label); the decoder-side tensors default to the encoder-side ones (every config without relative
        targets); `label` (N,) for label-conditioned configs"""
        if not self._ready:
            self._setup(commands.device)
        dec = None
        if commands_dec is not None or args_dec is not None:
            dec = (commands_dec if commands_dec is not None else commands, args_dec if args_dec is not None else args)
        if not commands.is_cuda:                    # CPU emulation (tests): no streams
            return self._step_body(commands, args, label, dec)
        model = self.model
        main = torch.cuda.current_stream()
        # The layout plan (a few tiny kernels + ONE host read) runs on its own stream so that the host read does not
        # wait for the previous step's work: the GPU keeps executing step k while the host plans and enqueues k+1.
        # Safe by default (the plan stream first waits for everything enqueued so far, i.e. for the inputs);
        # `inputs_resident = True` (inputs were complete before the previous step was enqueued) skips that wait.
        if self._plan_stream is None:
            self._plan_stream = _shared_stream("plan", commands.device)
        ps = self._plan_stream
        t_trace = [time.perf_counter()] if self.host_trace is not None else None
        if not self.inputs_resident:
            ps.wait_stream(main)
        counts = None
        with torch.cuda.stream(ps):
            plan = model.make_plan(commands, args, dec[0] if dec else commands, True, dec[1] if dec else args)
        t_plan = time.perf_counter() if self.host_trace is not None else None
        if self.ddp and self.use_graph:
            # data parallel + hipGraph: the 3-element count all-reduce goes out before the replay, its result is copied into
            # the graph's static tensor.  It runs on a stream of its OWN: RCCL executes a communicator's collectives in
            # issue order, so this one sits behind the previous step's gradient all-reduce - on the plan stream its wait
            # would hold back the NEXT step's plan kernels, the host's read of their result would block until the
            # previous step has finished, and the host could never run ahead of the GPU (measured on a one-rank group:
            # plan + read 6.7 ms instead of 0.45 ms, every graph launch exposed: 7.27 ms/step against 6.74 single-GPU)
            if self._count_stream is None:
                self._count_stream = _shared_stream("count", commands.device)
            cs = self._count_stream
            cs.wait_stream(ps)
            with torch.cuda.stream(cs):
                counts = self._global_counts(dec[0] if dec else commands, dec[1] if dec else args, plan)
            if plan["loss"] is not None:
                for t in plan["loss"]["targets"]:
                    t.record_stream(cs)
            main.wait_stream(cs)
        main.wait_stream(ps)
        if t_trace is not None:
            t_trace.append(time.perf_counter())
        if counts is not None:
            counts.record_stream(main)
        for part in ("enc", "dec", "loss"):        # allocated on the plan stream, read by launches on the main stream
            for v in (plan[part] or {}).values():
                for t in (v if isinstance(v, tuple) else (v,)):
                    if torch.is_tensor(t):
                        t.record_stream(main)
        if not self.use_graph:
            model._forced_plan = plan
            try:
                res = self._step_body(commands, args, label, dec)
            finally:
                model._forced_plan = None
            self._note_layout(plan, commands)
            return res
        if self.ddp:
            # data parallel: the graph holds forward + backward only.  The loss normalisers go out before it, the gradient
            # all-reduce and the optimiser run eagerly behind it (graph -> all-reduce -> clip + AdamW): no collective is
            # ever captured, and a rank that has to capture a new bucket issues exactly the collectives of a replaying one
            if self._counts is None:
                self._counts = counts.clone()
            else:
                self._counts.copy_(counts)
        key, plan = self._bucketed(plan, commands)
        key = key + (label is not None, dec is not None)
        entry, fresh = self._graph_entry(key, lambda: self._capture(key, commands, args, plan, label, dec))
        if not fresh:
            # refresh the graph's static inputs and static layout plan: ONE launch for all of them (dsvg_copy_many)
            graph, (sc, sa, sl, sdec), splan, res = entry
            pairs = [(sc, commands), (sa, args)]
            if sl is not None:
                pairs.append((sl, label))
            if sdec is not None:
                pairs += [(sdec[0], dec[0]), (sdec[1], dec[1])]
            for part in ("enc", "dec", "loss"):
                if splan[part] is not None:
                    for k, v in splan[part].items():
                        if torch.is_tensor(v):
                            pairs.append((v, plan[part][k]))
                        elif isinstance(v, tuple):
                            pairs += list(zip(v, plan[part][k]))
            if commands.is_cuda and all(d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype for d, s_ in pairs):
                ops.copy_many(pairs)
            else:
                for d, s_ in pairs:
                    d.copy_(s_)
        self._note_layout(plan, commands)
        self._pending = None
        entry[0].replay()
        if t_trace is not None:
            t_trace.append(time.perf_counter())
            t_trace.append(t_plan)
            self.host_trace.append(tuple(t_trace))
        if self.ddp:
            self._step_back()
        return entry[3]

    def _graph_entry(self, key, capture):
        """the cached graph of a bucket key (marked most recently used), or a freshly captured one; the cache holds at most
        `max_graphs` graphs, least recently used out first -> (entry, True when it was captured by this call)"""
        entry = self._graphs.get(key)
        if entry is not None:
            self._graphs.move_to_end(key)
            return entry, False
        entry = capture()
        self._graphs[key] = entry
        self.graphs_captured += 1
        while len(self._graphs) > max(1, self.max_graphs):
            # nothing of the evicted graph can still be running: graphs replay one at a time on this stream and a capture
            # synchronises the device; dropping the entry releases its static tensors and its blocks of the shared pool
            self._graphs.popitem(last=False)
            self.graphs_evicted += 1
        return entry, True

    def rccl_ranks(self):
        """number of ranks that take part in this trainer's collectives, measured by one all-reduce of ones (1 without a
        process group): bench.py reports it so that a scaling run can be checked to have used N ranks"""
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        # (before the first step the flat buffers are not on the device yet: the NCCL backend needs a device tensor all the same)
        dev = self.model.store.flat.device if self._ready else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        one = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(one, group=self.pg)
        return int(round(one.item()))

    def _bucketed(self, plan, commands):
        """round the plan's two data-dependent sizes up to buckets -> (graph cache key, plan with the rounded sizes)"""
        n_seq = commands.shape[0] * commands.shape[1]
        key = []
        if plan["enc"] is not None:
            rb = self.row_bucket
            rows = min((plan["enc"]["total"] + rb - 1) // rb * rb, n_seq * commands.shape[2])
            plan["enc"]["rows"] = rows
            key.append(rows)
        else:
            key.append(-1)
        if plan["dec"] is not None:
            sb = self.seq_bucket
            n_live = min((plan["dec"]["n_visible"] + sb - 1) // sb * sb, n_seq)
            plan["dec"]["n_live"] = n_live
            key.append(n_live)
        else:
            key.append(-1)
        if plan["loss"] is not None:
            rb = self.row_bucket
            rows = min((plan["loss"]["n_live"] + rb - 1) // rb * rb, n_seq * (commands.shape[2] - 1))
            plan["loss"]["rows"] = rows
            key.append(rows)
            key.append((plan["loss"].get("slot_lo", 0), plan["loss"].get("slot_hi", -1)))
        else:
            key.append(-1)
        return tuple(key), plan

    def _note_layout(self, plan, commands):
        n_seq = commands.shape[0] * commands.shape[1]
        m = self.model
        m.last_packing = (plan["enc"]["total"], n_seq * commands.shape[2]) if plan["enc"] is not None else None
        m.last_live = (plan["dec"]["n_visible"], n_seq) if plan["dec"] is not None else None
        m.last_head_rows = ((plan["loss"]["n_live"], n_seq * (commands.shape[2] - 1))
                            if plan["loss"] is not None else None)

    def _capture(self, key, commands, args, plan, label=None, dec=None):
        model = self.model
        sc, sa = commands.clone(), args.clone()
        sl = label.clone() if label is not None else None
        sdec = (dec[0].clone(), dec[1].clone()) if dec is not None else None
        def _static(v):
            if torch.is_tensor(v):
                return v.clone()
            if isinstance(v, tuple):
                return tuple(t.clone() for t in v)
            return v
        splan = {part: (None if plan[part] is None else {k: _static(v) for k, v in plan[part].items()})
                 for part in ("enc", "dec", "loss")}
        model._forced_plan = splan
        self._in_own_step = True        # (the warm-up runs and the capture read the pre-reduced loss counts, see _reduce_counts)
        # the warm-up steps (allocator pools, lazy buffers) must not train the model.  Data parallel: the captured part
        # (and its warm-up) is forward + backward only - no collective, see step()
        body = self._step_front if self.ddp else self._step_body
        state = [model.store.flat, self.m, self.v, self.step_count, self.seed]
        saved = [t.clone() for t in state]
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    body(sc, sa, sl, sdec)
                for t, s0 in zip(state, saved):
                    t.copy_(s0)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()     # graphs replay one at a time: one shared pool
            # thread-local capture mode: other threads of the process keep calling the runtime while this one captures -
            # with an initialised process group the RCCL watchdog thread polls its work events (hipEventQuery), which the
            # default global mode answers by invalidating the capture and the watchdog by aborting the process
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                res = body(sc, sa, sl, sdec)
        finally:
            model._forced_plan = None
            self._in_own_step = False
        return (g, (sc, sa, sl, sdec), splan, res)

    def grad_norm(self):
        """global gradient L2 norm of the last step (after the all-reduce averaging)"""
        return (self.gnorm_sq.sqrt() / self.world).item()

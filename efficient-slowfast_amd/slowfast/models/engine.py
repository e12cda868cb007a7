"""Host-side glue between the nn.Module tree (parameter containers with the reference's names) and the
libsfhip kernels: parameter preprocessing caches (packed weights, folded eval-mode BN), the fused
conv(+BN)(+residual)(+ReLU) unit, the stem trick, and the NCTHW <-> NDHWC boundary handling that lets
every top-level child be called on its own with the reference's list-of-NCTHW-tensors convention
(Grad-CAM contract, wdf_visualization/gradcam_video.py:92-105)."""
import os
import threading
import weakref

import torch
import torch.nn as nn

import sfhip
from sfhip import Act

_tls = threading.local()


class internal(object):
    """Context: children exchange NDHWC `Act`s instead of converting back to NCTHW tensors."""

    def __enter__(self):
        self.prev = getattr(_tls, "depth", 0)
        _tls.depth = self.prev + 1

    def __exit__(self, *a):
        _tls.depth = self.prev


def is_internal():
    return getattr(_tls, "depth", 0) > 0


def enter(xs):
    """list of NCTHW tensors or Acts -> list of Acts."""
    return [x if isinstance(x, Act) else sfhip.from_ncthw(x) for x in xs]


def leave(acts):
    """list of Acts -> what the caller expects (Acts inside a model forward, NCTHW tensors otherwise)."""
    if is_internal():
        return acts
    return [sfhip.to_ncthw(a) for a in acts]


# ------------------------------------------------------------------------------------------------ tape
class Tape(object):
    """Reverse-mode tape of ONE training forward.  Activation gradients live in zero-initialised NDHWC
    buffers shaped like their forward buffers (keyed by address), and every backward op ACCUMULATES into
    them — that is autograd's fan-in sum for residual branches and for the two consumers each pathway tensor
    has in the lateral fusions.  Parameter gradients are collected per nn.Parameter."""

    def __init__(self):
        self.ops = []
        self.gbuf = {}
        self.pgrads = {}
        self.out_act = None
        self.sink = None  # {param: its .grad} when parameter gradients are accumulated in place (set_grad_sink)
        self.side = None  # the side stream ops are being recorded for (run_paths), None = the caller's stream
        self.serial = getattr(_tls, "serial", False)  # the model keeps one stream (collective-carrying layers)
        self.input_ids = {}    # id(clip tensor) -> input index, for the inputs whose gradient autograd asked for
        self.input_grads = {}  # input index -> dL/d(clip), NCTHW (written by the stems' backward)
        self.joins = set()  # companion streams with weight-gradient work in flight (joined at the end of backward)
        self.model = None   # the model whose forward this tape records (milestone hooks are bound to it)
        self.early = None   # backward: event a fusion region left for the next region's side stream (run_paths)
        self.capturing = False  # the backward is being captured into a hipGraph (no companion-stream forks then)
        self.arena = None   # backward of a small model: ONE zero-filled tensor the gradient buffers are cut from
        self.arena_off = 0
        self.zero_floats = -1  # >= 0 while the backward runs: floats of zero-initialised buffers it has asked for

    def pgrad_target(self, param):
        """The tensor kernels may accumulate this parameter's gradient into directly, or None."""
        if self.sink is None:
            return None
        g = self.sink.get(param)
        return g if g is not None and g.is_contiguous() else None

    def grad_of(self, act):
        key = act.buf.data_ptr()
        g = self.gbuf.get(key)
        if g is None:
            n = act.buf.numel()
            nal = (n + 63) // 64 * 64
            if self.zero_floats >= 0:
                self.zero_floats += nal
            if self.arena is not None and self.arena_off + nal <= self.arena.numel() and \
                    self.arena.device == act.buf.device:
                g = self.arena[self.arena_off:self.arena_off + n]
                self.arena_off += nal
            else:
                g = torch.zeros(n, dtype=torch.float32, device=act.buf.device)
            self.gbuf[key] = g
        return Act(g.view(act.buf.shape), act.coff, act.C)

    def has_grad(self, act):
        return act.buf.data_ptr() in self.gbuf

    def grad_of_uninitialised(self, act):
        """The gradient buffer of a whole-buffer activation that has no gradient yet, NOT zeroed: the caller must
        overwrite every element (the first — usually only — consumer's data gradient), which saves the zero fill
        and the read of an accumulate.  Returns None when the buffer already exists or `act` is a channel slice."""
        key = act.buf.data_ptr()
        if key in self.gbuf or act.coff != 0 or act.cs != act.C:
            return None
        g = torch.empty(act.buf.numel(), dtype=torch.float32, device=act.buf.device)
        self.gbuf[key] = g
        return Act(g.view(act.buf.shape), 0, act.C)

    def record(self, fn):
        self.ops.append((fn, self.side))

    def add_pgrad(self, param, g):
        g = g.reshape(param.shape)
        tgt = self.sink.get(param) if self.sink is not None else None
        if tgt is not None:
            tgt.add_(g)
            return
        cur = self.pgrads.get(param)
        self.pgrads[param] = g if cur is None else cur + g

    def add_pgrads(self, params, grads):
        """add_pgrad for several parameters at once: with a gradient sink (FlatGradients) the accumulations are ONE
        multi-tensor launch instead of one small add per parameter (the q | k | v projections of an attention: six adds on
        the path between its weight gradient and its data gradient)."""
        tgts = [self.sink.get(p) if self.sink is not None else None for p in params]
        if all(t is not None for t in tgts) and len(tgts) > 1 and os.environ.get("SF_FOREACH_PGRAD", "1") != "0":
            torch._foreach_add_(tgts, [g.reshape(p.shape) for p, g in zip(params, grads)])
            return
        for p, g in zip(params, grads):
            self.add_pgrad(p, g)

    def backward(self):
        """Replay in reverse.  Ops recorded inside run_paths on the side stream run there again; the region's
        join / fork markers become the backward pass's fork / join."""
        self._open_arena()
        self.capturing = bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
        asked = -1
        try:
            for fn, side in reversed(self.ops):
                if side is None:
                    fn()
                else:
                    with torch.cuda.stream(side):
                        fn()
            for wg in self.joins:
                _sync_streams(wg, torch.cuda.current_stream(wg.device))
            asked = self.zero_floats
        finally:
            # also after a failed replay (SfhipError, OOM): no arena stays referenced, and no later grad_of outside a
            # backward is counted against — or cut from — a partly used, no longer zero one
            self.joins = set()
            self.ops = []
            self.zero_floats = -1
            self.gbuf = {}
            self.arena = None
            self.arena_off = 0
            self.capturing = False
        if self.model is not None and asked >= 0:   # published only by a backward that ran to its end
            self.model.__dict__["_sf_arena_floats"] = asked

    def _open_arena(self):
        """Launch-bound models (cfg #1: ~115 gradient buffers of a few KB each): the zero-initialised gradient buffers of
        one backward come out of ONE zero-filled tensor — one fill launch instead of one per buffer.  Sized by what the
        previous backward of the same model asked for; made here, on the stream the backward starts on and before any
        side stream is forked, so every later use is ordered behind the fill.  Only models whose buffers total at most
        64 MB (SF_GRAD_ARENA_MB) take it: a large model keeps one fill per buffer, which overlaps with the backward's
        kernels instead of delaying its first one (an arena for just the SMALL buffers of a large model was tried: cfg #3
        51.98 against 51.60 ms, two alternations on one box; cfg #5 at 2 clips unchanged)."""
        self.zero_floats = 0
        if self.model is None or ARENA_MAX_FLOATS <= 0 or self.out_act is None:
            return
        want = self.model.__dict__.get("_sf_arena_floats", 0)
        if 0 < want <= ARENA_MAX_FLOATS:
            self.arena = torch.zeros(want, dtype=torch.float32, device=self.out_act.buf.device)
            self.arena_off = 0


def tape():
    return getattr(_tls, "tape", None)


_MILESTONE_HOOKS = weakref.WeakKeyDictionary()  # model -> hook: a hook only ever sees the backward of ITS model


def set_milestone_hook(fn, model):
    """fn(child_name, tape) is called DURING the backward pass of `model` at the point where every gradient of the
    top-level child `child_name` and of all children after it has been issued (on the caller's stream, the side stream
    — joined by then — or a companion stream listed in tape.joins).  utils.distributed.FlatGradients uses it to start
    the all-reduce of the finished tail of the flat gradient buffer while the earlier stages' backward still runs (the
    reference gets the same overlap from DistributedDataParallel's buckets, models/build.py:39-43).  fn = None removes
    the model's hook."""
    if fn is None:
        _MILESTONE_HOOKS.pop(model, None)
    else:
        _MILESTONE_HOOKS[model] = fn


def milestone(name):
    """Called by the models' _forward_impl in front of each top-level child."""
    t = tape()
    if t is None or t.model is None:
        return
    hook = _MILESTONE_HOOKS.get(t.model)
    if hook is not None:
        t.record(lambda: hook(name, t))


# ------------------------------------------------------------------------------------------------ two streams
# The Slow and the Fast pathway are independent between lateral fusions, and so are the two directions of a CMDA
# fusion: the Fast pathway is a string of small-channel, HBM-bound kernels (and tiny grids), the Slow pathway of
# MFMA-bound ones.  run_paths issues the second callable on a side stream so the two overlap on the GPU; forward and
# backward regions are delimited by events (fork: side waits for the caller's stream; join: the caller's stream waits
# for side), which is also all the ordering the caching allocator needs: a block is only ever re-used by the stream
# that allocated it, and every region starts / ends by ordering that stream against the other.
OVERLAP_PATHS = os.environ.get("SF_OVERLAP_PATHS", "1") != "0"
_SIDE = {}


def _side_stream(device):
    s = _SIDE.get(device)
    if s is None:
        # priorities: SF_PRIO_SIDE / SF_PRIO_COMP (default 0 = normal).  High priority for this stream gained 0.25 ms of
        # the eager train step but cost 13-70 % of the eval forward's hipGraph replay (442 -> 383 clips/s, cfg #1 2866
        # -> 600): graph nodes captured on a high-priority stream replay through a slower path on this stack
        s = _SIDE[device] = torch.cuda.Stream(device=device, priority=int(os.environ.get("SF_PRIO_SIDE", "0")))
    return s


_COMPANION = {}


def _companion_stream(parent):
    """A stream paired with `parent` (the caller's or the side stream) for work only needed at the end of backward."""
    key = (parent.device, parent.cuda_stream)
    s = _COMPANION.get(key)
    if s is None:
        s = _COMPANION[key] = torch.cuda.Stream(device=parent.device, priority=int(os.environ.get("SF_PRIO_COMP", "0")))
    return s


def _sync_streams(first, then):
    ev = torch.cuda.Event()
    ev.record(first)
    then.wait_event(ev)


_FUSE = {}
DEFER_JOIN = os.environ.get("SF_DEFER_JOIN", "1") != "0"  # forward: the Slow pathway's next stage starts beside the attention
# SF_FUSE_STREAM=1 (default 0): the CMDA fusions' attention direction runs on a side stream of its own, so that in the
# backward pass the Fast pathway's stage k (which does not depend on the attention's gradient) runs BESIDE the
# attention backward of fusion k instead of queueing behind it on the shared side stream.  Measured (cfg #3, 8 clips,
# one box, two alternations): 57.20 / 57.03 ms with it against 56.28 / 56.30 without — the attention sweeps hold 384 of
# the 512 registers per lane, the Fast pathway's wavefronts that squeeze in beside them slow the sweep by more than
# their own kernels were worth next to the Slow pathway's.  Kept as a switch (and in the stream-equivalence test).
FUSE_STREAM = os.environ.get("SF_FUSE_STREAM", "0") == "1"


def _fuse_stream(device):
    s = _FUSE.get(device)
    if s is None:
        s = _FUSE[device] = torch.cuda.Stream(device=device, priority=int(os.environ.get("SF_PRIO_SIDE", "0")))
    return s


def run_paths(fns, device, defer_join=False, fuse=False):
    """[f() for f in fns] with fns[1] issued on the side stream (two callables on a CUDA device; otherwise serial).
    defer_join: do not make the caller's stream wait for the side stream at the end of the FORWARD region — only
    valid when the next work on the caller's stream does not read what fns[1] produced before the next region's
    join (a CMDA fusion followed by a stage: the attention keeps running beside the Slow pathway's next stage).
    fuse: a CMDA fusion's region — fns[1] (the attention direction) gets the fusion stream.  Forward: the next
    region's side stream waits for it (the Fast pathway reads the attention's output).  Backward: the caller's stream
    leaves an event behind fns[0]'s gradient ops BEFORE it waits for the attention's, and the next region's side stream
    (the Fast pathway's previous stage, which needs only those) starts from that event."""
    if not OVERLAP_PATHS or getattr(_tls, "serial", False) or len(fns) != 2 or device.type != "cuda":
        return [f() for f in fns]
    t = tape()
    if t is not None and t.side is not None:
        return [f() for f in fns]  # already inside a region
    fuse = fuse and FUSE_STREAM
    main, side = torch.cuda.current_stream(device), (_fuse_stream(device) if fuse else _side_stream(device))

    def bwd_join():  # last op of the region's backward
        cur = torch.cuda.current_stream(device)
        if fuse:
            ev = torch.cuda.Event()
            ev.record(cur)
            t.early = ev
        _sync_streams(side, cur)

    def bwd_fork():  # first op of the region's backward
        cur = torch.cuda.current_stream(device)
        ev, t.early = t.early, None
        if ev is not None and not fuse:
            side.wait_event(ev)
        else:
            _sync_streams(cur, side)

    if t is not None:
        t.record(bwd_join)
    _sync_streams(main, side)                                                       # forward: fork
    pend = getattr(_tls, "pending_side", None)
    if pend is not None and pend is not side:
        _sync_streams(pend, side)            # ... and behind a deferred region on another stream (its output is read here)
    out0 = fns[0]()
    with torch.cuda.stream(side):
        if t is not None:
            t.side = side
        try:
            out1 = fns[1]()
        finally:
            if t is not None:
                t.side = None
    if not defer_join:
        _sync_streams(side, main)                                                   # forward: join
        _tls.pending_side = None  # (a pending stream was waited for by `side` above: joined transitively)
    else:
        _tls.pending_side = side
    if t is not None:
        t.record(bwd_fork)
    return [out0, out1]


def join_pending(device):
    """Make the caller's stream wait for a region whose forward join was deferred and never followed by another
    region (end of a forward pass)."""
    pend = getattr(_tls, "pending_side", None)
    if pend is not None:
        _sync_streams(pend, torch.cuda.current_stream(device))
        _tls.pending_side = None


class taping(object):
    def __init__(self, t):
        self.t = t

    def __enter__(self):
        self.prev = getattr(_tls, "tape", None)
        _tls.tape = self.t

    def __exit__(self, *a):
        _tls.tape = self.prev


def _colsum(act):
    """per-channel sum over all rows (bias gradients)."""
    mean, _ = sfhip.channel_stats(act)
    return mean * float(act.rows)


def _record_conv(x, conv_weight, conv_bias, wp_shape, gsrc, kernel, stride, padding, dilation, x_needs_grad=True,
                 cin=None, unpack=None, fold_kw=0, x_planes=None):
    """Backward of a dense conv whose output gradient will be found in `gsrc` (an Act).  fold_kw: the stem layout
    (packed channel = (kw, ci)); `unpack` maps the packed gradient to the parameter's layout on the autograd path.
    x_planes: the bf16 piece planes of x the forward conv made (conv_bx.hip), reused by the weight gradient."""
    t = tape()
    if t is None:
        return
    cout = conv_weight.shape[0]

    def wgrad(g, zp):
        tgt = t.pgrad_target(conv_weight)
        if tgt is not None:  # partial sum + un-pack + accumulate into .grad in one kernel
            real_cin = conv_weight.shape[1]
            sfhip.conv_wgrad(x, g, cout, kernel, stride, padding, dilation, cin=cin, cin_pad=wp_shape[2],
                             finish_into=(tgt, real_cin, fold_kw), x_planes=x_planes, dz_planes=zp)
        else:
            dwp = sfhip.conv_wgrad(x, g, cout, kernel, stride, padding, dilation, cin=cin, cin_pad=wp_shape[2],
                                   x_planes=x_planes, dz_planes=zp)
            t.add_pgrad(conv_weight, unpack(dwp) if unpack else sfhip.unpack_conv_weight_grad(dwp, conv_weight.shape))
        if conv_bias is not None:
            t.add_pgrad(conv_bias, _colsum(g))

    def bwd():
        g = gsrc() if callable(gsrc) else gsrc
        dev = x.buf.device
        # dL/dz as bf16 piece planes, made once on this stream for the weight gradient and the data gradient when
        # either runs on conv_bx.hip
        zp = None
        if fold_kw == 0 and sfhip.bx_backward_wants_dz_planes(x, g, cout, kernel, stride, padding, dilation, cin=cin,
                                                               cin_pad=wp_shape[2], dgrad=x_needs_grad):
            zp = sfhip.act_planes(g)
        # The fork costs two cross-stream hand-offs: ~15 us each on the device when launched eagerly
        # (tools/microbench/stream_latency.py) — worth it even for the small layers of cfg #3 (52.9 ms forking every
        # layer, 53.2 forking only those estimated above 30 .. 120 us) — but far more as edges of a captured hipGraph:
        # cfg #1's replay went from 9.65 to 6.03 ms, cfg #5's from 43.3 to 34.9 ms and cfg #3's own from 56.6 to 54.1 ms
        # without them.  So: no companion stream while the backward is being CAPTURED; eagerly, every layer whose
        # estimated weight gradient (2 rows Cin Cout taps FLOP at 80 TFLOP/s + both operands once at 3 TB/s) reaches
        # SF_WGRAD_FORK_US (default 0: all).
        fork = OVERLAP_PATHS and WGRAD_COMPANION and (WGRAD_FORK_IN_GRAPH or not t.capturing) and not t.serial and \
            x_needs_grad and dev.type == "cuda"
        if fork and WGRAD_FORK_US > 0:
            rows_, cin_ = g.rows, (cin or x.C)
            est_us = (2.0 * rows_ * cin_ * cout * (kernel[0] * kernel[1] * kernel[2])) / 80e6 + \
                4.0 * (x.rows * cin_ + rows_ * cout) / 3e6
            fork = est_us >= WGRAD_FORK_US
        if fork:
            # the weight gradient only feeds the parameter's .grad: issue it on a companion stream so that it
            # overlaps the data gradient (both are short-grid GEMMs on the res4 / res5 layers); joined by
            # Tape.backward before the gradients are handed back
            cur = torch.cuda.current_stream(dev)
            wg = _companion_stream(cur)
            _sync_streams(cur, wg)
            with torch.cuda.stream(wg):
                wgrad(g, zp)
            g.buf.record_stream(wg)  # e.g. the masked-gradient temporary of a bare ReLU dies with this closure
            x.buf.record_stream(wg)
            for pl in (zp, x_planes):
                if pl is not None:
                    pl.record_stream(wg)
            t.joins.add(wg)
        else:
            wgrad(g, zp)
        if x_needs_grad:
            if conv_weight.dim() == 5 and tuple(conv_weight.shape[2:]) == tuple(kernel):
                wtp = _packed_pair(conv_weight)[1]
            else:  # nn.Linear weights ([K, C]) and other re-shaped uses
                wtp = _cached_t(conv_weight, "_sf_wtp", _key(conv_weight),
                                lambda: sfhip.pack_conv_weight(conv_weight.detach().reshape(
                                    conv_weight.shape[0], conv_weight.shape[1], *kernel).transpose(0, 1).contiguous()))
            fresh = t.grad_of_uninitialised(x)
            if fresh is not None:  # first consumer of x: write, do not accumulate
                sfhip.conv_dgrad(g, wtp, x, kernel, stride, padding, dilation, out=fresh, accumulate=False,
                                 dz_planes=zp)
            else:
                sfhip.conv_dgrad(g, wtp, x, kernel, stride, padding, dilation, out=t.grad_of(x), accumulate=True,
                                 dz_planes=zp)

    t.record(bwd)


_PCACHE = {}


# Parameter-derived tensors (packed weights, their planes, folded BN ...) that a hipGraph capture has read or written.
# They are allocated EAGERLY (outside the graph's private pool) and live in per-parameter caches; an eager forward after
# the capture that finds them stale (an eval pass between replays) replaces the cache entry, and without this table
# the old tensors would go back to the allocator while the graph still reads and re-packs into them on every replay.
_CAPTURE_KEEP = {}


def _keep_if_capturing(obj):
    if getattr(_tls, "capturing", False):
        _CAPTURE_KEEP[id(obj)] = obj
    return obj


def _cached_t(tensor, slot, key, make):
    """cache keyed on a tensor (parameters are not nn.Modules).  The entry lives ON the tensor object, so it dies
    with it: a global table keyed by id() would hand a new parameter that re-uses a freed one's id, address and
    version the old one's packed weights."""
    store = tensor.__dict__.setdefault("_sf_cache", {})
    c = store.get(slot)
    if c is None or c[0] != key:
        with torch.no_grad():
            c = (key, make())
        store[slot] = c
    return _keep_if_capturing(c[1])


# Everything derived from parameters (packed conv weights and their bf16 planes, folded BN affines, q/k/v stacks, ...)
# is cached per (address, torch version counter) of its sources.  torch's FUSED optimizers (torch.optim.SGD / Adam
# with fused=True: one multi-tensor kernel, torch._fused_sgd_) update the parameters WITHOUT moving their version
# counters, so the counter alone would leave every cache on the pre-update weights — silently: the step still runs,
# on frozen packed weights (found by tests/test_graph_train_gpu.py in round 5: the replayed graph, whose re-pack is a
# captured launch, disagreed with eager steps that skipped it).  Every optimizer step of the process therefore bumps
# an epoch that is part of every cache key (a global post-step hook: fires for any torch.optim.Optimizer, fused or
# not); code that writes parameters through raw pointers calls parameters_changed() itself.
_PARAM_EPOCH = 0


def parameters_changed(*_unused):
    """Invalidate every parameter-derived cache of the HIP path (packed weights, folded BN, ...)."""
    global _PARAM_EPOCH
    _PARAM_EPOCH += 1


def replay(graph):
    """graph.replay() for a hipGraph that holds a TRAINING step (forward, backward, optimizer update) — the form every
    caller of such a graph must use.  A replayed optimizer kernel moves the parameters without running torch's Python
    optimizer hooks and without touching version counters, and the re-pack launches captured in the graph refresh the
    copies in the GRAPH's memory pool, not the ones eager code reads: without the epoch bump an eager forward after the
    replays (an eval pass, a checkpoint's BN fold) would silently reuse packed weights, bf16 planes, folded BN and q/k/v
    stacks from before them (tests/test_graph_train_gpu.py::test_eval_after_replays_sees_the_trained_weights)."""
    graph.replay()
    parameters_changed()


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook
    _STEP_HOOK = _register_step_hook(lambda optimizer, args, kwargs: parameters_changed())
except ImportError:  # pragma: no cover — torch without global optimizer hooks
    _STEP_HOOK = None


def _key(*tensors):
    return (_PARAM_EPOCH,) + tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


def _cached(mod, slot, key, make):
    c = mod.__dict__.get(slot)
    if c is None or c[0] != key:
        with torch.no_grad():
            c = (key, make())
        mod.__dict__[slot] = c
    return _keep_if_capturing(c[1])


def check_bn(bn):
    from .batchnorm_helper import SubBatchNorm3d
    if not isinstance(bn, (nn.BatchNorm3d, SubBatchNorm3d)):
        raise NotImplementedError("unsupported normalisation layer on the HIP path: %s" % type(bn))


def _sub_bn(bn):
    from .batchnorm_helper import SubBatchNorm3d
    return isinstance(bn, SubBatchNorm3d)


def _sync_bn(bn):
    from .batchnorm_helper import NaiveSyncBatchNorm3d
    from slowfast.utils import distributed as du
    return isinstance(bn, NaiveSyncBatchNorm3d) and du.get_local_size() > 1


_NBT = None  # inside run_model: the num_batches_tracked buffers to bump with ONE foreach add at the end


def _bump(counter):
    if _NBT is not None:
        _NBT.append(counter)
    else:
        counter.add_(1)


def _plain_bn(bn):
    return not _sub_bn(bn) and not _sync_bn(bn)


def _bn_train_stats(bn, z, conv_stats=None):
    """(mean, invstd, scale, shift, nsplit, gamma_for_backward, sync_hook) of a training-mode norm layer; updates
    the running statistics in place exactly as the reference layer would.  conv_stats: per-tile statistics the
    producing conv's epilogue left (sfhip.conv(..., stats=True)) — plain BatchNorm3d then merges those instead of
    reading z again."""
    if _sub_bn(bn):
        S, sb = bn.num_splits, bn.split_bn
        if z.N % S != 0:
            raise ValueError("SubBatchNorm3d: batch %d is not divisible by num_splits %d" % (z.N, S))
        dev = z.buf.device
        w = bn.weight.detach().repeat(S) if bn.affine else torch.ones(S * z.C, device=dev)
        b = bn.bias.detach().repeat(S) if bn.affine else torch.zeros(S * z.C, device=dev)
        track = sb.track_running_stats and sb.running_mean is not None
        m = sb.momentum if sb.momentum is not None else 1.0 / float(int(sb.num_batches_tracked) + 1)
        mean, invstd, scale, shift = sfhip.bn_train_stats(
            z, w, b, sb.eps, m, sb.running_mean if track else None, sb.running_var if track else None, nsplit=S)
        if track:
            _bump(sb.num_batches_tracked)
        return mean, invstd, scale, shift, S, w, None
    track = bn.track_running_stats and bn.running_mean is not None
    m = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
    if _sync_bn(bn):
        from .batchnorm_helper import group_gather_sum
        nd, ng = bn.num_sync_devices, bn.num_groups
        lmean, lvar = sfhip.channel_stats(z)
        lm = lmean.double()
        vec = group_gather_sum(torch.cat([lm, lvar.double() + lm * lm]), nd, ng) / nd  # (mean, E[x^2]) in fp64
        mean64, meansqr = vec[:z.C], vec[z.C:]
        var64 = (meansqr - mean64 * mean64).clamp_min(0.0)
        mean, var = mean64.float(), var64.float()
        invstd = torch.rsqrt(var64 + bn.eps).float()
        scale = (bn.weight.detach() * invstd).contiguous()
        shift = (bn.bias.detach() - mean * scale).contiguous()
        if track:  # the reference keeps the BIASED variance here (batchnorm_helper.py:207-208)
            bn.running_mean += m * (mean - bn.running_mean)
            bn.running_var += m * (var - bn.running_var)
            bn.__dict__.pop("_sf_affine", None)

        def sync(dbeta, dgamma):  # GroupGather.backward: the same gather-and-sum, then the 1/num_sync factor
            tot = group_gather_sum(torch.cat([dbeta, dgamma]).double(), nd, ng) / nd
            return tot[:dbeta.numel()].float().contiguous(), tot[dbeta.numel():].float().contiguous()

        return mean, invstd, scale, shift, 1, bn.weight, sync
    if conv_stats is not None:
        mean, invstd, scale, shift = sfhip.bn_train_stats_merge(
            conv_stats, z.C, bn.weight, bn.bias, bn.eps, m, bn.running_mean if track else None,
            bn.running_var if track else None)
    else:
        mean, invstd, scale, shift = sfhip.bn_train_stats(
            z, bn.weight, bn.bias, bn.eps, m, bn.running_mean if track else None, bn.running_var if track else None)
    if track:  # the kernel wrote the running buffers in place: drop the folded eval-mode affine cache
        bn.__dict__.pop("_sf_affine", None)
        _bump(bn.num_batches_tracked)
    return mean, invstd, scale, shift, 1, bn.weight, None


def bn_train_apply(bn, z, res=None, relu=False, rep=1, out=None, out_reserve=(0, 0), keep=None, out_cmul=1,
                   conv_stats=None):
    """Training-mode BatchNorm3d on the raw tensor z: batch statistics (sf_channel_stats), running-stat update
    (momentum, unbiased variance — torch semantics), then ONE normalise(+residual)(+ReLU)(+T-repeat) pass."""
    check_bn(bn)
    with torch.no_grad():
        mean, invstd, scale, shift, nsplit, gamma_b, sync = _bn_train_stats(bn, z, conv_stats)
    zz = z if keep is None else z.slice(0, keep)
    nk = z.C if keep is None else keep
    if keep is not None:
        assert nsplit == 1, "channel-sliced BN (GhostModule) is only defined for plain BatchNorm3d"
        scale, shift = scale[:keep].contiguous(), shift[:keep].contiguous()
    t = tape()
    shuffled = t is not None and out_cmul != 1
    # taped ReLU / ReLU6 layers: the normalise pass also leaves a byte per 4 channels for the backward's mask, which
    # then reads that instead of the activation (1/16 of the bytes, twice per backward)
    mk = {} if (t is not None and relu) else None
    if shuffled:
        # training: a channel-shuffled store is an explicit (taped) strided copy of the dense result
        y = sfhip.affine(zz, scale.contiguous(), shift.contiguous(), res=res, relu=relu, rep=rep, nsplit=nsplit,
                         mask=mk)
    else:
        y = sfhip.affine(zz, scale.contiguous(), shift.contiguous(), res=res, relu=relu, rep=rep, out=out,
                         out_reserve=out_reserve, out_cmul=out_cmul, nsplit=nsplit, mask=mk)
    mask = mk.get("bytes") if mk else None
    if t is not None:
        if _GRAD_SINK:
            dg = db = None  # allocated lazily by the fallback below
        else:
            dg = torch.zeros(z.C, dtype=torch.float32, device=z.buf.device)
            db = torch.zeros(z.C, dtype=torch.float32, device=z.buf.device)
        sel = slice(None) if nsplit > 1 else slice(0, nk)

        def bwd():
            dres, first = None, False
            if res is not None:  # first writer of the residual branch's gradient: write it, no zero fill / read
                fresh = t.grad_of_uninitialised(res) if (nsplit == 1 and res.C == zz.C and rep == 1) else None
                dres, first = (fresh, True) if fresh is not None else (t.grad_of(res), False)
            wt = t.pgrad_target(bn.weight) if (nsplit == 1 and getattr(bn, "weight", None) is not None) else None
            bt = t.pgrad_target(bn.bias) if wt is not None else None
            if wt is not None and bt is not None:  # dgamma / dbeta accumulate into .grad inside the reduction
                sfhip.bn_bwd(t.grad_of(y), y, zz, mean[sel], invstd[sel], gamma_b[sel], relu, rep=rep, dres=dres,
                             dz_out=zz, sync=sync, grad_sink=(wt, bt), mask=mask, dres_overwrite=first)
            else:
                dg_ = dg if dg is not None else torch.zeros(z.C, dtype=torch.float32, device=z.buf.device)
                db_ = db if db is not None else torch.zeros(z.C, dtype=torch.float32, device=z.buf.device)
                sfhip.bn_bwd(t.grad_of(y), y, zz, mean[sel], invstd[sel], gamma_b[sel], relu, rep=rep, dres=dres,
                             dz_out=zz, dgamma_out=(dg_, db_), nsplit=nsplit, sync=sync, mask=mask,
                             dres_overwrite=first)
            if keep is not None and keep < z.C:  # sliced-away channels (GhostModule [:oup]) get no gradient
                rest = z.slice(keep, z.C - keep)
                sfhip.axpy(rest, rest, alpha=0.0, accumulate=False)
            if (wt is None or bt is None) and getattr(bn, "weight", None) is not None:
                t.add_pgrad(bn.weight, dg_)
                t.add_pgrad(bn.bias, db_)

        t.record(bwd)  # afterwards z's buffer holds dL/dz for the producer's backward
    if shuffled:  # recorded AFTER the BN op so that its backward (the gather) runs first
        return copy_channels(y, out, out_cmul=out_cmul)
    return y


def bn_affine(bn, conv_bias=None):
    """Eval-mode BatchNorm3d as per-channel (scale, bias); a preceding conv bias is folded in."""
    check_bn(bn)
    stat = bn.bn if _sub_bn(bn) else bn  # SubBatchNorm3d evaluates with the aggregated `bn` statistics
    w = getattr(bn, "weight", None)
    b = getattr(bn, "bias", None)

    def make():
        inv = 1.0 / torch.sqrt(stat.running_var + stat.eps)
        scale = w * inv if w is not None else inv
        bias = -stat.running_mean * scale
        if b is not None:
            bias = bias + b
        if conv_bias is not None:
            bias = bias + scale * conv_bias
        return scale.detach().contiguous(), bias.detach().contiguous()

    return _cached(bn, "_sf_affine", _key(w, b, stat.running_mean, stat.running_var, conv_bias), make)


def norm_forward(bn, x):
    """A norm layer called on its own (NCTHW tensor or Act): the HIP statistics / normalise kernels."""
    (a,) = enter([x])
    y = bn_train_apply(bn, a) if bn.training else sfhip.affine(a, *bn_affine(bn))
    return leave([y])[0]


ARENA_MAX_FLOATS = int(float(os.environ.get("SF_GRAD_ARENA_MB", "64")) * (1 << 18))  # 0: off
_GROUPS_OF = {}  # id(weight) -> groups, for the grouped convs (1 < groups < channels) packed by _group_pairs
_PAIR_WEIGHTS = {}  # id(weight) -> weakref of the dense conv weights that have been packed as a pair (tensors compare
# elementwise, so no WeakSet): repack_all's candidates
BATCHED_REPACK = os.environ.get("SF_BATCH_REPACK", "1") != "0"
WGRAD_COMPANION = os.environ.get("SF_WGRAD_COMPANION", "1") != "0"  # weight gradients on a companion stream
WGRAD_FORK_US = float(os.environ.get("SF_WGRAD_FORK_US", "0"))       # ... when estimated to take at least this long
WGRAD_FORK_IN_GRAPH = os.environ.get("SF_WGRAD_FORK_IN_GRAPH", "0") == "1"  # A/B: fork also while a hipGraph is captured


def _packed_pair(weight):
    """(forward, data-gradient) packed copies of a dense conv weight, built together by one kernel and cached per
    parameter version (training re-packs every conv once per optimizer step — repack_all does that in one launch)."""
    if not weight.is_cuda:  # CPU construction / state_dict work: torch fallback for the forward layout only
        return sfhip.pack_conv_weight(weight), None
    if id(weight) not in _PAIR_WEIGHTS:
        wid = id(weight)
        _PAIR_WEIGHTS[wid] = weakref.ref(weight, lambda _r, wid=wid: _PAIR_WEIGHTS.pop(wid, None))
    return _cached_t(weight, "_sf_wpair", _key(weight), lambda: sfhip.pack_conv_weight_pair(weight))


def repack_all(model):
    """Re-pack, in ONE launch, every dense, depthwise and grouped conv weight of `model` whose packed pair is stale
    (after an optimizer step: all of them — ~108 launches of ~6 us and 216 allocations otherwise; a grouped conv alone
    was G launches and two allocations).  The packed tensors are overwritten in place:
    the previous step's kernels that read them were joined before the optimizer ran."""
    if not BATCHED_REPACK:
        return
    cand = model.__dict__.get("_sf_pair_params")
    if cand is None or cand[0] != len(_PAIR_WEIGHTS):
        cand = (len(_PAIR_WEIGHTS), [p for p in model.parameters()
                                     if id(p) in _PAIR_WEIGHTS and _PAIR_WEIGHTS[id(p)]() is p])
        model.__dict__["_sf_pair_params"] = cand
    stale, old, slots = [], [], []
    for w in cand[1]:
        store = w.__dict__.get("_sf_cache", {})
        c = store.get("_sf_wpair")
        if c is not None and c[0] != _key(w) and w.is_cuda and w.is_contiguous():
            stale.append(w.detach())
            old.append(c[1])
            slots.append((w, "_sf_wpair", None))
        c = store.get("_sf_wgroups")
        G = _GROUPS_OF.get(id(w))
        if c is not None and G and c[0] != _key(w) and w.is_cuda and w.is_contiguous():
            # a grouped conv's pair = its G per-group packs one after the other: G more records of the same launch,
            # written into the slices of the two tensors the previous version lives in (no allocation)
            wd, (wp, wtp) = w.detach(), c[1]
            cout_g, cin_g = wd.shape[0] // G, wd.shape[1]
            for g in range(G):
                stale.append(wd[g * cout_g:(g + 1) * cout_g])
                old.append((wp[g * cout_g:(g + 1) * cout_g], wtp[g * cin_g:(g + 1) * cin_g]))
                slots.append((w, "_sf_wgroups", (wp, wtp)) if g == 0 else None)
    if len(stale) < 2:
        return
    with torch.no_grad():
        new = sfhip.pack_conv_weight_pairs(stale, old)
    for sl, o, prev in zip(slots, new, old):
        if sl is None:
            continue
        w, name, whole = sl
        if whole is not None and (o[0].data_ptr() != prev[0].data_ptr() or o[1].data_ptr() != prev[1].data_ptr()):
            continue  # a slice was re-allocated (foreign device / dtype): leave the entry stale, _group_pairs re-packs
        w.__dict__["_sf_cache"][name] = (_key(w), _keep_if_capturing(o if whole is None else whole))


def packed_weight(conv):
    if conv.groups == 1:
        return _packed_pair(conv.weight)[0]
    if conv.groups == conv.in_channels and conv.out_channels == conv.in_channels:
        if conv.weight.is_cuda:
            # [taps][pad16(C)]: the transposed half of the pair pack of the [C, 1, kT, kH, kW] parameter — depthwise
            # weights are then re-packed after an optimizer step by repack_all's ONE launch, with the dense ones
            return _packed_pair(conv.weight)[1][0]
        return _cached(conv, "_sf_wp", _key(conv.weight), lambda: sfhip.pack_dw_weight(conv.weight))
    return _group_pairs(conv)


def _group_pairs(conv):
    """(forward, data-gradient) packed weights of a grouped conv (1 < groups < channels) for the one-launch grouped
    kernels, cached per parameter version: wp [Cout][taps][pad(Cin/G)], wtp [Cin][taps][pad(Cout/G)] — group g owns
    output channels [g*Cout/G, (g+1)*Cout/G) and reads input channels [g*Cin/G, (g+1)*Cin/G): nn.Conv3d(groups=G),
    shufflenet_helper.py:48-63 / resnet_helper.py:196-205."""
    w, G = conv.weight, conv.groups
    if conv.in_channels % G or conv.out_channels % G:
        raise ValueError("grouped conv: %d -> %d channels are not divisible by %d groups" % (
            conv.in_channels, conv.out_channels, G))
    if w.is_cuda and id(w) not in _PAIR_WEIGHTS:  # repack_all's candidate list (it re-packs the groups in its one launch)
        wid = id(w)
        _PAIR_WEIGHTS[wid] = weakref.ref(w, lambda _r, wid=wid: (_PAIR_WEIGHTS.pop(wid, None), _GROUPS_OF.pop(wid, None)))
        _GROUPS_OF[wid] = G
    return _cached_t(w, "_sf_wgroups", _key(w), lambda: sfhip.pack_grouped_weight_pair(w, G))


def grouped_conv(x, conv, scale=None, bias=None, relu=False, res=None, out=None, out_reserve=(0, 0), shuffle=False):
    """nn.Conv3d with 1 < groups < channels as a block-diagonal GEMM in ONE launch (sfhip.conv_grouped: the group is
    the grid's z index; input window g -> output window g of the same NDHWC buffers, no tensor is split or
    concatenated).  shuffle: the output is stored channel-shuffled — group g's channel j lands at j*G + g,
    channel_shuffle(., G) of shufflenet_helper.py:22-29 as index math of the stores (eval mode; the taped path shuffles
    after the BN)."""
    k, s, p, d = conv.kernel_size, conv.stride, conv.padding, conv.dilation
    assert x.C == conv.in_channels, (x, conv)
    return sfhip.conv_grouped(x, _group_pairs(conv)[0], conv.groups, k, s, p, d,
                              scale=None if scale is None else scale.contiguous(),
                              bias=None if bias is None else bias.contiguous(), relu=relu, res=res, out=out,
                              shuffle=shuffle, out_reserve=out_reserve)


def _record_grouped_conv(x, conv, gsrc):
    """Backward of grouped_conv (un-shuffled output): ONE weight-gradient launch (+ its finish, straight into the
    parameter's gradient in nn.Conv3d's grouped layout [Cout][Cin/G][k]) and ONE data-gradient launch for all groups."""
    t = tape()
    if t is None:
        return
    G = conv.groups
    k, s, p, d = conv.kernel_size, conv.stride, conv.padding, conv.dilation

    def bwd():
        g_all = gsrc() if callable(gsrc) else gsrc
        wp, wtp = _group_pairs(conv)
        tgt = t.pgrad_target(conv.weight)
        if tgt is not None:
            sfhip.conv_wgrad_grouped(x, g_all, G, k, s, p, d, cin_pad=wp.shape[2], finish_into=tgt)
        else:
            dwp = sfhip.conv_wgrad_grouped(x, g_all, G, k, s, p, d, cin_pad=wp.shape[2])
            t.add_pgrad(conv.weight, sfhip.unpack_conv_weight_grad(dwp, tuple(conv.weight.shape)))
        sfhip.conv_dgrad_grouped(g_all, wtp, G, x, k, s, p, d, out=t.grad_of(x), accumulate=True)
        if conv.bias is not None:
            t.add_pgrad(conv.bias, _colsum(g_all))

    t.record(bwd)


def channel_shuffle(x, groups, out=None, out_reserve=(0, 0)):
    """channel_shuffle(x, G) (shufflenet_helper.py:22-29): channel g*C/G + j -> j*G + g, one launch (taped: the inverse
    permutation — the same kernel with C/G groups — accumulates the gradient in the backward).  The eval path never
    calls it: grouped_conv(shuffle=True) stores so."""
    if groups == 1:
        return x
    if out is None:
        out = sfhip.new_act(x, x.N, x.T, x.H, x.W, x.C, out_reserve[0], out_reserve[1])
    sfhip.channel_shuffle(x, out, groups)
    t = tape()
    if t is not None:
        t.record(lambda: sfhip.channel_shuffle(t.grad_of(out), t.grad_of(x), x.C // groups, accumulate=True))
    return out


def _is_grouped(conv):
    return 1 < conv.groups and not (conv.groups == conv.in_channels and conv.out_channels == conv.in_channels)


def conv_bn_act(x, conv, bn=None, relu=False, res=None, out=None, out_reserve=(0, 0), out_cmul=1, cout=None,
                shuffle=False):
    """nn.Conv3d [+ BatchNorm3d] [+ residual] [+ ReLU].  Eval: ONE kernel launch (BN folded into the
    epilogue).  Training: raw conv -> batch statistics -> one normalise/residual/ReLU pass.
    shuffle (grouped convs only): channel_shuffle(., conv.groups) applied to the result."""
    if _is_grouped(conv):
        assert out_cmul == 1 and cout is None, "grouped convs have their own shuffle"
        if bn is not None and bn.training:
            z = grouped_conv(x, conv, bias=conv.bias)
            _record_grouped_conv(x, conv, z)
            if not shuffle:
                return bn_train_apply(bn, z, res=res, relu=relu, out=out, out_reserve=out_reserve)
            y = bn_train_apply(bn, z, res=res, relu=relu)
            return channel_shuffle(y, conv.groups, out=out, out_reserve=out_reserve)
        scale, bias = bn_affine(bn, conv.bias) if bn is not None else (None, conv.bias)
        if tape() is not None:
            raise NotImplementedError("taped grouped conv with a folded (eval-mode) BN epilogue")
        return grouped_conv(x, conv, scale=scale, bias=bias, relu=relu, res=res, out=out, out_reserve=out_reserve,
                            shuffle=shuffle)
    assert not shuffle or conv.groups == 1, "channel_shuffle of a depthwise conv's output is not used by any model"
    wp = packed_weight(conv)
    if bn is not None and bn.training:
        k, s, p, d = conv.kernel_size, conv.stride, conv.padding, conv.dilation
        st = None
        if conv.groups == 1:
            # plain BatchNorm3d: its batch statistics come out of the conv's epilogue (no extra pass over z)
            want = _plain_bn(bn) and bn.affine
            keep = {} if tape() is not None else None
            z = sfhip.conv(x, wp, k, s, p, d, bias=conv.bias, stats=want, keep=keep)
            z, st = z if want else (z, None)
            _record_conv(x, conv.weight, conv.bias, wp.shape, z, k, s, p, d, x_planes=keep.get("x") if keep else None)
        else:
            ones = _cached(conv, "_sf_ones", (conv.out_channels, str(x.buf.device)),
                           lambda: torch.ones(conv.out_channels, dtype=torch.float32, device=x.buf.device))
            z = sfhip.dwconv(x, wp, k, s, p, scale=ones if conv.bias is not None else None, bias=conv.bias)
            t = tape()
            if t is not None:
                def bwd_dw():  # z's buffer holds dL/dz after the BN backward
                    tgt = t.pgrad_target(conv.weight)
                    wpk = packed_weight(conv)   # (this step's pack: the forward's may have been re-packed in place)
                    if tgt is not None:  # the reduction's final step accumulates into the parameter's gradient
                        sfhip.dwconv_bwd(x, z, wpk, k, s, p, dx=t.grad_of(x), into=tgt)
                    else:
                        dw = sfhip.dwconv_bwd(x, z, wpk, k, s, p, dx=t.grad_of(x))
                        t.add_pgrad(conv.weight, dw[:, :x.C].t().contiguous())
                    if conv.bias is not None:
                        t.add_pgrad(conv.bias, _colsum(z))

                t.record(bwd_dw)
        return bn_train_apply(bn, z, res=res, relu=relu, out=out, out_reserve=out_reserve, keep=cout,
                              out_cmul=out_cmul, conv_stats=st)
    if bn is not None:
        scale, bias = bn_affine(bn, conv.bias)
    else:
        scale, bias = None, conv.bias
    k, s, p, d = conv.kernel_size, conv.stride, conv.padding, conv.dilation
    if conv.groups == 1:
        y = sfhip.conv(x, wp, k, s, p, d, scale=scale, bias=bias, relu=relu, res=res, out=out,
                       out_reserve=out_reserve, out_cmul=out_cmul)
        t = tape()
        if t is not None:
            if bn is not None or res is not None or out_cmul != 1:
                raise NotImplementedError("taped conv with a folded (eval-mode) BN / residual epilogue")
            if relu:  # bare conv + ReLU (no BN): dL/dz = dL/dy masked by the activation's output
                def masked():
                    dz = sfhip.new_act(y, y.N, y.T, y.H, y.W, y.C)
                    return sfhip.act_bwd(t.grad_of(y), y, relu, dz, accumulate=False)
                _record_conv(x, conv.weight, conv.bias, wp.shape, masked, k, s, p, d)
            else:
                _record_conv(x, conv.weight, conv.bias, wp.shape, lambda: t.grad_of(y), k, s, p, d)
        return y
    assert d == (1, 1, 1)
    if scale is None and bias is not None:
        scale = torch.ones_like(bias)
    return sfhip.dwconv(x, wp, k, s, p, scale=scale, bias=bias, relu=relu, res=res, out=out, cout=cout,
                        out_cmul=out_cmul)


def stem_geometry(conv, H, W):
    """(ph, pw, Wp) of the border-padded NDHWC4 input the stem convolution `conv` consumes for H x W frames."""
    kT, kH, kW = conv.kernel_size
    _, sH, sW = conv.stride
    pT, pH, pW = conv.padding
    Wo = (W + 2 * pW - kW) // sW + 1
    wp_need = max(W + 2 * pW, (Wo - 1) * sW + kW)
    return pH, pW, (wp_need + sW - 1) // sW * sW


def stem_conv_bn_relu(x, conv, bn, relu=True):
    """First convolution of a pathway straight from the caller's NCTHW clip (Cin <= 4).

    The clip is converted ONCE to NDHWC with channels padded to 4 and the H/W borders zero padded, so a
    (kT,kH,kW) conv with W-stride sW becomes a (kT,kH,1) conv over 'pixels' of sW*4 floats that consumes
    kW*4 CONTIGUOUS floats per tap: aligned 16-byte loads, no border predicate in H/W, and the K dimension
    is kT*kH taps x 4kW instead of kT*kH*kW taps x 16-padded 3 channels."""
    assert conv.groups == 1 and conv.in_channels <= 4 and conv.dilation == (1, 1, 1) and conv.stride[0] == 1
    kT, kH, kW = conv.kernel_size
    _, sH, sW = conv.stride
    pT, pH, pW = conv.padding
    N, C, T, H, W = x.shape
    Ho = (H + 2 * pH - kH) // sH + 1
    Wo = (W + 2 * pW - kW) // sW + 1
    wp_need = max(W + 2 * pW, (Wo - 1) * sW + kW)
    Wp = (wp_need + sW - 1) // sW * sW
    if isinstance(x, sfhip.PackedClip):  # produced by the GPU input step already in this layout
        if (x.ph, x.pw, x.Wp) != (pH, pW, Wp):
            raise ValueError("PackedClip geometry (ph, pw, Wp) = %s does not match this stem's %s; pack it with "
                             "engine.stem_geometry(conv, H, W)" % ((x.ph, x.pw, x.Wp), (pH, pW, Wp)))
        abuf = x.buf
    else:
        abuf = sfhip.from_ncthw(x, cpad=4, ph=pH, pw=pW, wp=Wp).buf
    view = Act(abuf.view(N, T, H + 2 * pH, Wp // sW, 4 * sW))

    def make():
        w = conv.weight
        w4 = torch.zeros((w.shape[0], 4, kT, kH, kW), dtype=torch.float32, device=w.device)
        w4[:, :C] = w
        cin = 4 * kW
        cin_pad = (cin + 15) // 16 * 16
        wp = torch.zeros((w.shape[0], kT * kH, cin_pad), dtype=torch.float32, device=w.device)
        wp[:, :, :cin] = w4.permute(0, 2, 3, 4, 1).reshape(w.shape[0], kT * kH, cin)
        return wp.contiguous()

    wp = _cached(conv, "_sf_wp_stem", _key(conv.weight), make)
    thw = (T + 2 * pT - kT + 1, Ho, Wo)
    if bn.training:
        want = _plain_bn(bn) and bn.affine
        z = sfhip.conv(view, wp, (kT, kH, 1), (1, sH, 1), (pT, 0, 0), bias=conv.bias, cin=4 * kW, out_thw=thw,
                       stats=want)
        z, st = z if want else (z, None)

        def unpack(dwp):  # [Cout][kT*kH][kW*4 + c] -> [Cout, C, kT, kH, kW]
            co = dwp.shape[0]
            return dwp[:, :, :4 * kW].reshape(co, kT, kH, kW, 4)[..., :C].permute(0, 4, 1, 2, 3).contiguous()

        _record_conv(view, conv.weight, conv.bias, wp.shape, z, (kT, kH, 1), (1, sH, 1), (pT, 0, 0), (1, 1, 1),
                     x_needs_grad=False, cin=4 * kW, unpack=unpack, fold_kw=kW)
        _record_stem_input_grad(x, conv, z, view, (N, C, T, H, W), (pH, pW, Wp), Wo)
        return bn_train_apply(bn, z, relu=relu, conv_stats=st)
    scale, bias = bn_affine(bn, conv.bias)
    return sfhip.conv(view, wp, (kT, kH, 1), (1, sH, 1), (pT, 0, 0), scale=scale, bias=bias, relu=relu,
                      cin=4 * kW, out_thw=thw)


def _record_stem_input_grad(x, conv, z, view, ncthw, geom, Wo):
    """dL/d(clip) when the caller asked for it (x.requires_grad: saliency / adversarial uses; train_net.py never
    does).  In the stem-trick layout a tap reads kW*4 contiguous floats = ceil(kW*4 / (4*sW)) whole pixels of 4*sW
    floats, so the stem IS an ordinary conv with kernel (kT, kH, KP) over pixels with 4*sW channels (weights
    re-arranged, zero where a tap's float index runs past kW*4) and its data gradient is the generic one; the clip's
    gradient is the interior (borders and the pad channel dropped) of that buffer."""
    t = tape()
    if t is None or not isinstance(x, torch.Tensor) or id(x) not in t.input_ids:
        return
    N, C, T, H, W = ncthw
    pH, pW, Wp = geom
    kT, kH, kW = conv.kernel_size
    _, sH, sW = conv.stride
    pT = conv.padding[0]
    pix = 4 * sW
    KP = (kW * 4 + pix - 1) // pix
    if view.W - KP + 1 != Wo:
        raise NotImplementedError("input gradient of a stem whose packed row is wider than its outputs need")
    idx = t.input_ids[id(x)]

    def make():  # [Cout, 3, kT, kH, kW] -> pixel-conv weight [Cout, pix, kT, kH, KP] -> data-gradient packing
        w = conv.weight
        w4 = torch.zeros((w.shape[0], kT, kH, KP * pix), dtype=torch.float32, device=w.device)
        w4[..., :kW * 4].view(w.shape[0], kT, kH, kW, 4)[..., :C] = w.permute(0, 2, 3, 4, 1)
        wpix = w4.view(w.shape[0], kT, kH, KP, pix).permute(0, 4, 1, 2, 3).contiguous()
        return sfhip.pack_conv_weight(wpix.transpose(0, 1).contiguous())

    def bwd():  # z's buffer holds dL/dz after the BN backward
        wtp = _cached_t(conv.weight, "_sf_wtp_stem", _key(conv.weight), make)
        dxp = sfhip.conv_dgrad(z, wtp, view, (kT, kH, KP), (1, sH, 1), (pT, 0, 0))
        full = dxp.buf.view(N, T, H + 2 * pH, Wp, 4)
        t.input_grads[idx] = full[:, :, pH:pH + H, pW:pW + W, :C].permute(0, 4, 1, 2, 3).contiguous()

    t.record(bwd)


def maxpool(x, kernel, stride, padding=(0, 0, 0), out_reserve=(0, 0)):
    """MaxPool3d (+ its backward on the tape)."""
    t = tape()
    if t is None:
        return sfhip.pool(x, kernel, stride, padding, out_reserve=out_reserve)
    y, arg = sfhip.pool(x, kernel, stride, padding, out_reserve=out_reserve, want_arg=True)
    if t is not None:
        def bwd():
            fresh = t.grad_of_uninitialised(x)  # first writer of x's gradient: write, no zero fill / read
            dx = fresh if fresh is not None else t.grad_of(x)
            # the forward's winner map: no x / y reads, no tie search (else: the search form)
            if arg is not None and sfhip.maxpool_bwd_arg(x, arg, t.grad_of(y), dx, kernel, stride, padding,
                                                         overwrite=fresh is not None):
                return
            sfhip.maxpool_bwd(x, y, t.grad_of(y), dx, kernel, stride, padding, overwrite=fresh is not None)

        t.record(bwd)
    return y


_GRAD_SINK = False


def set_grad_sink(enabled):
    """Opt-in fast path for the training step: parameter gradients are ACCUMULATED IN PLACE into the parameters'
    existing .grad tensors by the backward kernels themselves (conv weight gradients: partial-sum + un-pack +
    accumulate in one launch; BatchNorm dgamma/dbeta: inside the reduction's final kernel) instead of being handed
    to autograd, which would add one temporary, one copy and one accumulate launch per parameter.  Use it with
    pre-allocated gradients (utils.distributed.FlatGradients binds every .grad to one flat buffer).  Autograd
    hooks on the parameters (DistributedDataParallel) do NOT fire for sunk gradients: leave it off under DDP —
    run_model raises for a model that build_model wrapped in DistributedDataParallel while the sink is on."""
    global _GRAD_SINK
    _GRAD_SINK = bool(enabled)


class TapedForward(torch.autograd.Function):
    """The whole HIP forward as ONE autograd node: forward records the tape, backward replays it and hands
    the parameter gradients back to autograd (so .grad accumulation and DDP's all-reduce hooks work as for
    any module).  Activations never become autograd tensors."""

    @staticmethod
    def forward(ctx, model, n_in, *args):
        inputs, params = list(args[:n_in]), args[n_in:]
        t = Tape()
        t.model = model
        t.input_ids = {id(x): i for i, x in enumerate(inputs)
                       if isinstance(x, torch.Tensor) and ctx.needs_input_grad[2 + i]}
        with taping(t):
            out = model._forward_impl(inputs)
        if t.out_act is None:
            raise RuntimeError("the head did not register its logits on the tape")
        ctx.tape = t
        ctx.params = params
        ctx.n_in = n_in
        return out

    @staticmethod
    def backward(ctx, dout):
        t = ctx.tape
        if _GRAD_SINK:  # accumulate into the existing .grad tensors ourselves; autograd gets None for those
            t.sink = {p: p.grad for p in ctx.params if p.grad is not None and p.grad.dtype == torch.float32}
        with torch.no_grad(), taping(None):
            g = t.grad_of(t.out_act)
            g.buf.copy_(dout.reshape(g.buf.shape))
            t.backward()
        sink = t.sink or {}
        grads = tuple(None if p in sink else t.pgrads.get(p) for p in ctx.params)
        gin = tuple(t.input_grads.get(i) for i in range(ctx.n_in))
        if any(ctx.needs_input_grad[2 + i] and gin[i] is None for i in range(ctx.n_in)):
            raise NotImplementedError("dL/d(input) was requested for an input this model's stem does not "
                                      "differentiate (PackedClip inputs / non-stem consumers)")
        ctx.tape = None
        return (None, None) + gin + grads


def run_model(model, x):
    """model.forward body shared by all model classes: taped when training with grad enabled."""
    global _NBT
    outer, _NBT = _NBT, []
    # collective-carrying layers (Sync-BN over > 1 local rank): one stream for this model only.  The decision depends on
    # the process group's state, so the cache is keyed on it (a forward before init_process_group must not pin False)
    from slowfast.utils import distributed as du
    local_size = du.get_local_size()
    cached = model.__dict__.get("_sf_serial_streams")
    if cached is None or cached[0] != local_size:
        serial = local_size > 1 and any(getattr(m, "_sf_collective", False) for m in model.modules())
        model.__dict__["_sf_serial_streams"] = (local_size, serial)
        if serial and OVERLAP_PATHS:
            print("[sfhip] %s holds Sync-BN layers over %d local ranks: its pathways run on one stream (collectives "
                  "stay in program order)" % (type(model).__name__, local_size))
    else:
        serial = cached[1]
    outer_serial, _tls.serial = getattr(_tls, "serial", False), serial
    outer_cap = getattr(_tls, "capturing", False)
    _tls.capturing = bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
    try:
        if model.training and torch.is_grad_enabled():
            if _GRAD_SINK and getattr(model, "_sf_ddp_wrapped", False):
                raise RuntimeError("engine.set_grad_sink(True) bypasses autograd's gradient hooks, which "
                                   "DistributedDataParallel's all-reduce depends on: switch the sink off, or use "
                                   "utils.distributed.FlatGradients on the unwrapped model")
            params = [p for p in model.parameters()]
            repack_all(model)
            return TapedForward.apply(model, len(x), *x, *params)
        return model._forward_impl(x)
    finally:
        if x and hasattr(x[0], "device") and x[0].device.type == "cuda":
            join_pending(x[0].device)  # a fusion whose join was deferred and that no later region picked up
        _tls.serial = outer_serial
        _tls.capturing = outer_cap
        counters, _NBT = _NBT, outer
        if counters:
            with torch.no_grad():
                torch._foreach_add_(counters, 1)


def copy_channels(x, out, out_cmul=1):
    """out[.., out.coff + c*out_cmul] = x[.., c] (+ the gather on the tape)."""
    sfhip.copy_channels(x, out, out_cmul=out_cmul)
    t = tape()
    if t is not None:
        t.record(lambda: sfhip.gather_add(t.grad_of(out), out_cmul, t.grad_of(x), accumulate=True))
    return out


def add_into(a, res, out):
    """out = a + res (elementwise, channel slices), taped: both inputs receive dL/dout."""
    sfhip.affine(a, res=res, out=out)
    t = tape()
    if t is not None:
        def bwd():
            g = t.grad_of(out)
            sfhip.axpy(g, t.grad_of(a), 1.0, accumulate=True)
            sfhip.axpy(g, t.grad_of(res), 1.0, accumulate=True)
        t.record(bwd)
    return out


def avgpool(x, kernel, stride, padding):
    """nn.AvgPool3d(kernel, stride, padding) with count_include_pad=True: a depthwise conv whose taps are all
    1/|kernel| (zero padding == counting the pad).  Taped: the data gradient is the transposed gather."""
    taps = kernel[0] * kernel[1] * kernel[2]
    # keyed on the issuing stream too: a constant created inside a run_paths region must not be shared with the
    # other stream before the region's join
    key = ("avgpool", taps, x.C, str(x.buf.device), torch.cuda.current_stream(x.buf.device).cuda_stream)
    w = _PCACHE.get(key)
    if w is None:
        w = torch.full((taps, x.C), 1.0 / taps, dtype=torch.float32, device=x.buf.device)
        _PCACHE[key] = w
    y = sfhip.dwconv(x, w, kernel, stride, padding)
    t = tape()
    if t is not None:
        t.record(lambda: sfhip.dwconv_dgrad(x, t.grad_of(y), w, kernel, stride, padding, t.grad_of(x)))
    return y


def global_mean(x, out=None):
    """[N,T,H,W,C] -> [N,1,1,1,C] mean over T,H,W (taped)."""
    pooled = sfhip.tmax_mean(x, 1)
    a = Act(pooled.view(x.N, 1, 1, 1, x.C))
    res = sfhip.copy_channels(a, out) if out is not None else a
    t = tape()
    if t is not None:
        def bwd():
            g = t.grad_of(res)
            v = g.buf.view(g.N, g.cs)[:, g.coff:g.coff + g.C].contiguous()
            sfhip.bcast_add(t.grad_of(x), v, 1.0 / float(x.T * x.H * x.W))
        t.record(bwd)
    return res


def small_torch_op(inputs, params, fn):
    """A parameter-sized sub-graph ([N, C] tensors: SE excitation, GhostNet's conv_head on pooled features)
    evaluated with torch ops; on the tape its backward is torch.autograd.grad over the tiny graph.
    inputs: list of Acts whose views are [N,1,1,1,C]; fn(list of [N,C] tensors) -> [N,K] tensor.
    Returns an Act [N,1,1,1,K]."""
    t = tape()
    flat = [a.buf.view(a.N, a.cs)[:, a.coff:a.coff + a.C] for a in inputs]
    if t is None:
        with torch.no_grad():
            y = fn([f for f in flat])
        return Act(y.contiguous().view(y.shape[0], 1, 1, 1, y.shape[1]))
    with torch.enable_grad():
        leaves = [f.detach().clone().requires_grad_(True) for f in flat]
        y = fn(leaves)
    out = Act(y.detach().contiguous().view(y.shape[0], 1, 1, 1, y.shape[1]))

    def bwd():
        g = t.grad_of(out).buf.view(y.shape)
        grads = torch.autograd.grad(y, leaves + list(params), g, allow_unused=True)
        for a, ga in zip(inputs, grads[:len(leaves)]):
            if ga is not None:
                ga_act = Act(ga.contiguous().view(a.N, 1, 1, 1, a.C))
                sfhip.axpy(ga_act, t.grad_of(a), 1.0, accumulate=True)
        for p, gp in zip(params, grads[len(leaves):]):
            if gp is not None:
                t.add_pgrad(p, gp)

    t.record(bwd)
    return out

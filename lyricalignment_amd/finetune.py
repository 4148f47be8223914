"""Host plumbing of the fine-tune row (e): the reference's losses on the align logits
(train_multitask.py:587-633), clip_grad_norm_ + AdamW (:337-340, 683-686) and -- new in this build -- the data-parallel
gradient exchange (one flat all-reduce per optimizer step over RCCL / xGMI, placed BEFORE the clip).
Arithmetic = HIP kernels (la_multitask_loss, la_grad_sqnorm_f32, la_adamw_step_f32) and the collective; nothing here
computes on tensors.

FineTuner assembles the whole step: frame_manual_forward under autograd (head_train / encoder_train / decoder_train run
the forward + backward kernels), the loss kernels on the logits, flat gradient buckets, the all-reduce, clip + AdamW.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import head_train
from ._lib import check, lib, ptr, stream_ptr


def pad_frame_labels(frame_labels: torch.Tensor, frames: int) -> torch.Tensor:
    """train_multitask.py:596-603: trim / right-pad the frame labels with -100 to the logits' frame count (data movement)."""
    fl = frame_labels[:, :frames]
    if fl.shape[1] < frames:
        fl = torch.cat((fl, torch.full((fl.shape[0], frames - fl.shape[1]), -100, dtype=fl.dtype, device=fl.device)), dim=1)
    return fl


def multitask_loss(logits: torch.Tensor, frame_labels: Optional[torch.Tensor], ctc_labels: Optional[torch.Tensor],
                   vocab_size: int = 21128, scale: float = 1.0, want_grad: bool = True):
    """logits [B,T,vocab_size+1] f32 (device).  frame_labels [B,<=T] with -100 (or None: no CE/BCE);
    ctc_labels [B,Lmax] with -100 padding (or None: no CTC).
    -> (losses f32[3] = [word CE, silence BCE, CTC] on the device, dlogits or None).
    compute_ce_loss(...) == losses[0] + losses[1]; compute_ctc_loss(...) == losses[2]."""
    _lib.require_gpu()
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 3 or not logits.is_contiguous():
        raise ValueError("multitask_loss: logits must be a contiguous float32 device tensor [B,T,V+1]")
    B, T, W = logits.shape
    if W < vocab_size + 1:
        raise ValueError("multitask_loss: logits need vocab_size + 1 columns (word classes + silence)")
    dev = logits.device
    fl = None
    if frame_labels is not None:
        fl = pad_frame_labels(frame_labels.to(dev), T).to(torch.int32).contiguous()
    lab = nl = None
    Lmax = 1
    if ctc_labels is not None:
        cl = ctc_labels.to(dev)
        nl = (cl != -100).sum(dim=1).to(torch.int32).contiguous()          # target_length (:629); data-dependent count, not arithmetic of the path
        lab = torch.where(cl == -100, torch.zeros_like(cl), cl).to(torch.int32).contiguous()
        Lmax = max(1, lab.shape[1])
    losses = torch.empty((3,), dtype=torch.float32, device=dev)
    # the gradient kernel writes columns 0 .. vocab_size of every row; wider logits get exact zeros in the columns past them
    dlogits = (torch.zeros_like(logits) if W > vocab_size + 1 else torch.empty_like(logits)) if want_grad else None
    need = ctypes.c_size_t(0)
    check(lib().la_multitask_loss_workspace_bytes(B, T, Lmax, ctypes.byref(need)), "multitask_loss_workspace_bytes")
    ws = torch.empty((need.value,), dtype=torch.uint8, device=dev)
    check(lib().la_multitask_loss(ptr(logits), logits.stride(0), logits.stride(1), B, T, vocab_size, ptr(fl), ptr(lab),
                                  lab.stride(0) if lab is not None else 0, ptr(nl), Lmax, 1 if fl is not None else 0,
                                  1 if lab is not None else 0, float(scale), ptr(losses), ptr(dlogits),
                                  dlogits.stride(0) if want_grad else 0, dlogits.stride(1) if want_grad else 0, ptr(ws),
                                  need.value, stream_ptr()), "multitask_loss")
    return losses, dlogits


class FlatAdamW:
    """clip_grad_norm_(all params, max_norm) + AdamW over flat f32 buckets, one bucket per parameter group
    (reference: head lr 5e-3, backbone lr 5e-6, weight_decay 1e-5, betas (0.9, 0.999), eps 1e-8)."""

    def __init__(self, groups: Sequence[dict], weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8):
        _lib.require_gpu()
        self.groups = []
        for g in groups:
            p = g["params"]
            if not (p.is_cuda and p.dtype == torch.float32 and p.dim() == 1 and p.is_contiguous()):
                raise ValueError("FlatAdamW: each group needs a flat contiguous float32 device parameter buffer")
            self.groups.append(dict(params=p, lr=float(g["lr"]), m=torch.zeros_like(p), v=torch.zeros_like(p)))
        self.wd, self.betas, self.eps, self.t = weight_decay, betas, eps, 0
        self._sumsq = torch.zeros((1,), dtype=torch.float64, device=self.groups[0]["params"].device)

    def step(self, grads: Sequence[torch.Tensor], max_norm: Optional[float] = 1.0, grad_prescale: float = 1.0,
             lr_scale: float = 1.0) -> torch.Tensor:
        """grads: one flat buffer per group (already averaged over ranks).  Returns the device scalar sum(grad^2)."""
        self.t += 1
        L, s = lib(), stream_ptr()
        self._sumsq.zero_()
        if max_norm is not None:
            for g in grads:
                check(L.la_grad_sqnorm_f32(ptr(g), g.numel(), ptr(self._sumsq), s), "grad_sqnorm")
        for grp, g in zip(self.groups, grads):
            if g.numel() != grp["params"].numel() or g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("FlatAdamW.step: gradient bucket does not match its parameter bucket")
            check(L.la_adamw_step_f32(ptr(grp["params"]), ptr(g), ptr(grp["m"]), ptr(grp["v"]), g.numel(), grp["lr"] * lr_scale,
                                      self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                      ptr(self._sumsq) if max_norm is not None else 0, float(max_norm or 0.0),
                                      float(grad_prescale), s), "adamw_step")
            # the kernel wrote through a raw pointer: tell torch the buffer (and every plain view of it) changed
            torch.autograd.graph.increment_version(grp["params"])
        # Parameters bound to the buffer with `p.data = flat[...]` keep version counters of their own which the line above does not move
        # (FineTuner.step bumps those it knows): the derived-weight caches are keyed on an epoch as well
        from .encoder_train import invalidate_weight_caches
        invalidate_weight_caches()
        return self._sumsq


def allreduce_mean_(buckets: Sequence[torch.Tensor], world: int) -> None:
    """The data-parallel exchange step: ONE all-reduce (sum) per flat gradient bucket per optimizer step, then the mean is
    folded into the optimizer's grad_prescale (1/world) by the caller -- or applied here for backends without a fused path.
    RCCL over xGMI on the GPU box (backend 'nccl'), gloo in the CPU tests.  No-op for world == 1."""
    if world == 1:
        return
    import torch.distributed as dist
    # both buckets are handed to the backend before either is waited for (RCCL: back-to-back ring passes on its own stream;
    # ~0.1 % of the 3.14 GB is the small second bucket, so this only saves its launch latency)
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in buckets]
    for w in works:
        w.wait()


class OverlappedAllReduce:
    """The same exchange, started DURING the last backward of the optimizer step instead of after it: every flat gradient bucket
    is cut into chunks (parameter boundaries, at least `min_chunks` for the backbone), a chunk's all-reduce (sum, async_op) is
    handed to the backend the moment autograd has accumulated the last gradient that lands in it
    (Tensor.register_post_accumulate_grad_hook), and finish() waits for all of them -- plus, synchronously, any chunk whose
    parameters received no gradient in that backward.  Backward runs the network back to front, so the decoder's and the last
    encoder blocks' chunks cross the links while the first blocks are still being differentiated: at N = 8 the 3.14 GB of
    Whisper-medium gradients are ~36 ms of ring all-reduce at the per-link rate against a 0.8 s step, and all but the last chunk
    (the conv stem + first blocks: 1 / min_chunks of it) hides behind the backward.  Elementwise sums are the same sums whatever the
    chunking: with 2 ranks the result is bit-identical to allreduce_mean_ (tests), with more ranks it differs like any other
    ring order would.  world == 1: nothing is registered, finish() returns 0.

    The ORDER of the collectives is fixed beforehand and the same on every rank -- `order`: the backbone's chunks from the last
    parameters to the first (the order a backward completes them in), then the head's -- whatever each rank's own batch made ready
    first: a ready chunk is handed to the backend only once every chunk before it in `order` has been, and finish() issues the rest
    in that order.  (Collectives of different sizes issued in different orders on different ranks hang or silently mix buffers: a
    rank whose last micro-batch has no frame-labelled clips gives the head no gradient in that backward while its neighbour's does --
    train_multitask.py:299-321 with use_ctc_loss off, :226.)  A rank that never completes a chunk holds everything behind it
    until finish(): correct, merely un-overlapped on that step."""

    def __init__(self, groups, grads, world: int, min_chunks: int = 4):
        self.world = world
        self.chunks = []                    # (bucket view, number of parameters in it)
        self._chunk_of = {}                 # id(param) -> chunk index
        self._hooks = []
        self._armed = False
        self._pending, self._works, self._launched = [], [], []
        self._ready, self._next = [], 0
        self.order: List[int] = []          # chunk indices in the one order every rank issues its collectives in
        self.exposed_ms = 0.0
        if world == 1:
            return
        group_chunks = []
        for params, flat in zip(groups, grads):
            first_chunk = len(self.chunks)
            n_chunks = min(len(params), min_chunks if flat.numel() >= (1 << 16) else 1)
            target = flat.numel() / n_chunks
            off = start = made = 0
            members = []
            for i, p in enumerate(params):
                members.append(p)
                off += p.numel()
                if off >= target * (made + 1) or i == len(params) - 1:        # cut at the first parameter boundary past k / n of the bucket
                    made += 1
                    ci = len(self.chunks)
                    self.chunks.append((flat[start:off], len(members)))
                    for q in members:
                        self._chunk_of[id(q)] = ci
                    start, members = off, []
            group_chunks.append(list(range(first_chunk, len(self.chunks))))
        # groups = [head, backbone] (FineTuner): the backbone's chunks back to front, then the head's
        for cs in reversed(group_chunks):
            self.order += list(reversed(cs))
        for params in groups:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def arm(self) -> None:
        """Call right before the LAST backward of the optimizer step (gradients of earlier micro-steps are already in the buckets)."""
        if self.world == 1:
            return
        if self._armed or self._works:
            # all-reduces of an earlier armed backward are still in flight (the caller skipped finish() / step()): their chunks are
            # already being summed in place, dropping the handles would sum them again or race the caller's zeroing of the bucket
            raise RuntimeError("OverlappedAllReduce.arm(): the previous armed backward was not finished -- call finish() (on every "
                               "rank) before re-arming, also when the optimizer step is skipped")
        self._pending = [n for _, n in self.chunks]
        self._works, self._launched = [], [False] * len(self.chunks)
        self._ready, self._next = [False] * len(self.chunks), 0
        self._armed = True
        # the stream the backward is started from: with the head on a stream of its own (module/align_model.py) a hook may fire on
        # either; a collective is handed to the backend only after BOTH have been waited for (_launch_in_order)
        self._main_stream = torch.cuda.current_stream(self.chunks[0][0].device) if self.chunks and self.chunks[0][0].is_cuda else None

    def _launch_in_order(self, everything: bool = False) -> None:
        """Hand chunks to the backend strictly in `order`: up to the first one that is not ready yet (all of them from finish())."""
        import torch.distributed as dist
        joined = False
        while self._next < len(self.order):
            ci = self.order[self._next]
            if not (everything or self._ready[ci]):
                return
            if not joined and self.chunks[ci][0].is_cuda:
                # the backend orders the collective behind the CURRENT stream only: make that one wait for the other streams gradients
                # were written on (the chunks go out back to front, so by the time one is ready the other stream has long been idle)
                cur = torch.cuda.current_stream(self.chunks[ci][0].device)
                for st in (getattr(self, "_main_stream", None), *self.extra_streams()):
                    if st is not None and st != cur:
                        cur.wait_stream(st)
                joined = True
            self._works.append(dist.all_reduce(self.chunks[ci][0], op=dist.ReduceOp.SUM, async_op=True))
            self._launched[ci] = True
            self._next += 1

    def extra_streams(self):
        """Streams besides the backward's own that gradients may be written on (set by FineTuner: the model's head stream)."""
        f = getattr(self, "_extra_streams_fn", None)
        return [st for st in (f() if f else []) if st is not None]

    def _on_grad(self, p) -> None:
        if not self._armed:
            return
        ci = self._chunk_of[id(p)]
        self._pending[ci] -= 1
        if self._pending[ci] == 0:
            self._ready[ci] = True
            self._launch_in_order()

    def finish(self) -> float:
        """Wait for the chunks in flight and all-reduce the ones the backward never completed.  What this costs on the device's
        current stream (= what the exchange still costs after the overlap) is bracketed with events: last_exposed_ms()."""
        if self.world == 1:
            return 0.0
        import torch.distributed as dist
        on_gpu = self.chunks[0][0].is_cuda
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if not self._armed:                  # never armed on this step: every chunk, in the same fixed order
            self._launched, self._ready, self._next = [False] * len(self.chunks), [False] * len(self.chunks), 0
        self._armed = False
        self._launch_in_order(everything=True)
        for w in self._works:
            w.wait()
        self._works = []
        if on_gpu:
            e1.record()
            self._events = (e0, e1)          # read lazily (last_exposed_ms): no host synchronisation inside the step
        return 0.0

    def launched_any(self) -> bool:
        return self._armed and any(self._launched)

    def last_exposed_ms(self) -> float:
        ev = getattr(self, "_events", None)
        if ev is None:
            return 0.0
        ev[1].synchronize()
        return float(ev[0].elapsed_time(ev[1]))


def linear_warmup_scale(step: int, warmup_steps: int, train_steps: int) -> float:
    """transformers.get_linear_schedule_with_warmup's LambdaLR factor (train_multitask.py:688-690): linear 0 -> 1 over the
    warm-up, then linear 1 -> 0 at train_steps.  `step` counts completed optimizer steps."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(train_steps - step) / float(max(1, train_steps - warmup_steps)))


class FineTuner:
    """The reference's train_step (train_multitask.py:215-342) on the HIP kernels, one process per GPU.

    micro_step(): both sub-batches of the reference's micro-step -- the multitask one (clips with frame labels: frame CE
    [+ silence BCE + CTC when use_ctc_loss] on the align logits, decoder CE; :250-291) and the transcript-only one (clips
    without frame labels: decoder CE [+ CTC on the align logits]; :299-321) -- each through frame_manual_forward under
    autograd (encoder / head / decoder forward + backward kernels) and the loss kernels, divided by accum_grad_steps.
    Gradients land DIRECTLY in two flat float32 buckets (head, backbone): every parameter's .grad is a view of its bucket,
    so autograd's accumulation is the bucket accumulation (no concatenation pass, no second copy of the gradients).
    step(): ONE all-reduce (sum) per bucket over RCCL -- the only exchange of the data-parallel path -- then the fused
    global-norm clip + AdamW over the buckets (mean over ranks folded into the update), linear warm-up schedule.
    The module's parameters are views into the flat parameter buckets, so the update is in place and nothing is scattered
    back.  Parameters that cannot receive a gradient under the model's flags (the decoder without train_transcript, the
    head without train_alignment, anything with requires_grad False) stay out of the buckets: torch.optim.AdamW skips
    parameters whose .grad is None (no weight decay either), and so does this."""

    def __init__(self, model, lr: float = 5e-3, backbone_lr: float = 5e-6, weight_decay: float = 1e-5, warmup_steps: int = 0,
                 train_steps: int = 2000, max_grad_norm: float = 1.0, use_ctc_loss: bool = True, vocab_size: int = 21128,
                 world: Optional[int] = None, allreduce_chunks: int = 4):
        _lib.require_gpu()
        self.model = model
        self.use_ctc_loss, self.vocab_size, self.max_grad_norm = use_ctc_loss, vocab_size, max_grad_norm
        self.warmup_steps, self.train_steps, self.steps_done = warmup_steps, train_steps, 0
        if world is None:
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.world = world
        dev = model._device()
        decoder = getattr(model.whisper_model, "decoder", None)
        dec_ids = {id(p) for p in decoder.parameters()} if decoder is not None else set()
        head = [p for p in model.align_rnn.parameters() if p.requires_grad] if model.train_alignment else []
        backbone = [p for p in model.whisper_model.parameters()
                    if p.requires_grad and (model.train_transcript or id(p) not in dec_ids)]
        self.groups: List[List[torch.nn.Parameter]] = [head, backbone]     # lr (:683), backbone_lr (:684)
        lrs = [lr, backbone_lr]
        self.flat, self.grad = [], []
        opt_groups = []
        for params, group_lr in zip(self.groups, lrs):
            if not params:
                continue
            n = sum(p.numel() for p in params)
            flat = torch.empty((n,), dtype=torch.float32, device=dev)
            gflat = torch.zeros_like(flat)
            off = 0
            for p in params:                                   # parameters and their .grad become views of the buckets
                flat[off: off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[off: off + p.numel()].view_as(p)
                p.grad = gflat[off: off + p.numel()].view_as(p)
                off += p.numel()
            self.flat.append(flat)
            self.grad.append(gflat)
            opt_groups.append({"params": flat, "lr": group_lr})
        self.groups = [g for g in self.groups if g]
        if self.world > 1:                                     # replicas start from rank 0's parameters (the head's initialisation
            import torch.distributed as dist                   # is drawn per process), one broadcast per bucket
            for flat in self.flat:
                dist.broadcast(flat, src=0)
        self.opt = FlatAdamW(opt_groups, weight_decay=weight_decay)
        # the exchange step, overlapped with the last backward (allreduce_chunks = 0: the single blocking all-reduce per bucket)
        self.overlap = OverlappedAllReduce(self.groups, self.grad, self.world, allreduce_chunks) if allreduce_chunks > 0 else None
        # head and decoder branch on two streams (module/align_model.py); with more ranks the gradient chunks leave for the all-reduce
        # from inside the backward on whichever stream produced the last of them: the exchange joins both streams before every collective
        model._branch_streams = True
        if self.overlap is not None:
            self.overlap._extra_streams_fn = lambda: [getattr(model, "_head_stream", None)]

    @property
    def allreduce_exposed_ms(self) -> float:
        """Device time the last step() spent on the gradient exchange after the overlap with the backward (0 for world 1)."""
        return self.overlap.last_exposed_ms() if self.overlap is not None else 0.0

    def _backward(self, roots, grads) -> None:
        """torch.autograd.backward with this tuner as the reader of the head branch's parked GRU time-out flags (no host synchronisation inside
        the backward; check_deferred_flags() follows every call)."""
        prev, head_train.CALLER_CHECKS_FLAGS = head_train.CALLER_CHECKS_FLAGS, True
        try:
            torch.autograd.backward(roots, grads)
        finally:
            head_train.CALLER_CHECKS_FLAGS = prev

    def _check_grad_views(self) -> None:
        """autograd must have accumulated IN PLACE into the bucket views (it does while .grad is defined and grad mode is
        off in backward); a replaced .grad tensor would silently drop gradients from the all-reduce / update."""
        for params, acc in zip(self.groups, self.grad):
            lo, hi = acc.data_ptr(), acc.data_ptr() + acc.numel() * 4
            for p in params:
                if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                    raise RuntimeError("FineTuner: a parameter's .grad was replaced; gradients must accumulate into the flat bucket")

    def _align_losses(self, align_logit, frame_labels, ctc_labels, s: float, out: torch.Tensor, roots, grads) -> None:
        """compute_ce_loss (+ compute_ctc_loss) on the align logits (train_multitask.py:271-281, 587-633), forward + d/dlogits."""
        logits = align_logit.detach().contiguous()
        if self.use_ctc_loss:
            l3, dlog = multitask_loss(logits, frame_labels, ctc_labels, vocab_size=self.vocab_size, scale=s)
            out[:3] += l3
        else:
            # compute_sil == False (:603-605): plain cross-entropy over ALL output columns, labels unshifted, -100 ignored
            if frame_labels is None:
                return
            from .decoder_train import cross_entropy
            fl = pad_frame_labels(frame_labels.to(logits.device), logits.shape[1])
            l, dlog = cross_entropy(logits, fl, scale_grad=s)
            out[0] += l
        roots.append(align_logit); grads.append(dlog)

    def micro_step(self, audios, ctc_labels=None, frame_labels=None, decoder_input=None, decoder_output=None,
                   accum_grad_steps: int = 1, get_orig_len: bool = False, transcript_batch=None, last: bool = False):
        """One micro-batch of train_step: forward, losses, backward.  Labels are pinyin-class ids with -100 padding (the
        caller maps tokens -> classes as train_step :259-268 does; harness.PinyinClassLUT).
        audios / ctc_labels / frame_labels / decoder_input / decoder_output: the MULTITASK sub-batch (clips with frame
        labels; None or empty audios = absent).  transcript_batch = (audios, ctc_labels, decoder_input, decoder_output): the
        TRANSCRIPT-ONLY sub-batch (clips without frame labels, :299-321): decoder CE, plus CTC on its align logits when
        use_ctc_loss and train_alignment.  Returns the device loss vector [align CE (word CE with use_ctc_loss), silence BCE,
        CTC (both sub-batches), decoder CE (both sub-batches)], un-scaled like the reference's logging."""
        m = self.model
        m.train()
        out = torch.zeros((4,), dtype=torch.float32, device=self.flat[0].device)
        s = 1.0 / float(accum_grad_steps)
        from .decoder_train import cross_entropy
        has_main = audios is not None and len(audios) > 0
        has_tr = transcript_batch is not None and len(transcript_batch[0]) > 0
        # last=True: this is the optimizer step's last micro-step -- the gradient chunks leave for the other ranks as its LAST
        # backward completes them (with both sub-batches present that is the transcript-only one's)
        if last and self.overlap is not None and has_main and not has_tr:
            self.overlap.arm()
        if has_main:
            roots, grads = [], []
            y_in = decoder_input if (m.train_transcript and decoder_input is not None) else None
            align_logit, trans_logit = m.frame_manual_forward(audios, y_in, get_orig_len=get_orig_len)
            if align_logit is not None and m.train_alignment:
                self._align_losses(align_logit, frame_labels, ctc_labels, s, out, roots, grads)
            if trans_logit is not None and decoder_output is not None:
                l, dl = cross_entropy(trans_logit.detach().contiguous(), decoder_output, scale_grad=s)
                out[3] += l
                roots.append(trans_logit); grads.append(dl)
            if roots:
                self._backward(roots, grads)
            head_train.check_deferred_flags()                     # GRU time-out flags of a head that ran beside the decoder
        if has_tr:
            if last and self.overlap is not None:
                self.overlap.arm()
            t_audios, t_ctc, t_in, t_out = transcript_batch
            roots, grads = [], []
            align_logit, trans_logit = m.frame_manual_forward(t_audios, t_in if m.train_transcript else None, get_orig_len=get_orig_len)
            if trans_logit is not None and t_out is not None:
                l, dl = cross_entropy(trans_logit.detach().contiguous(), t_out, scale_grad=s)
                out[3] += l
                roots.append(trans_logit); grads.append(dl)
            if self.use_ctc_loss and m.train_alignment and align_logit is not None and t_ctc is not None:
                l3, dlog = multitask_loss(align_logit.detach().contiguous(), None, t_ctc, vocab_size=self.vocab_size, scale=s)
                out[2] += l3[2]
                roots.append(align_logit); grads.append(dlog)
            if roots:
                self._backward(roots, grads)
            head_train.check_deferred_flags()                     # GRU time-out flags of a head that ran beside the decoder
        self._check_grad_views()
        return out

    def accumulate(self, micro_batches, accum_grad_steps: Optional[int] = None, get_orig_len: bool = False, fused: bool = True,
                   decoder_pad_id: int = 0):
        """The accumulation loop of train_step (train_multitask.py:240-326: `accum_grad_steps` micro-batches, each loss divided
        by accum_grad_steps, gradients summed) for a list of MULTITASK micro-batches, each a dict with the keyword arguments of
        micro_step (audios, ctc_labels, frame_labels, decoder_input, decoder_output).
        fused=True runs the micro-batches as ONE forward / backward over all their clips -- the activations of 8 x 2 clips of
        Whisper-medium are ~80 GB of the 288 GB -- with every loss still taken per micro-batch slice (its own means, its own
        1 / accum factor), so the accumulated gradient is the same sum: the GEMMs see 8 x the rows (M = 24000 instead of
        3000: the float32 kernel fills the chip), the two persistent GRU sweeps per layer run once over 16 clips instead of
        eight times over 2, and ~8 x fewer kernels are launched.  Dropout masks are drawn over the fused batch (a different
        random stream than eight separate draws, the same distribution).  fused=False is the loop of micro_step() calls.
        decoder_pad_id: any valid token id (it fills prompt positions whose targets are -100 and which no earlier position
        attends to).  Returns the summed (un-scaled) loss vector like micro_step."""
        accum = int(accum_grad_steps or len(micro_batches))
        m = self.model
        mel = None
        # the fused form takes one loss set for all micro-batches: it needs them homogeneous (all with or all without the decoder
        # pair) and free of the extra transcript batch micro_step accepts; anything else takes the loop, which trains each as given
        has_dec = [mb.get("decoder_input") is not None and mb.get("decoder_output") is not None for mb in micro_batches]
        homogeneous = (all(has_dec) or not any(has_dec)) and not any(mb.get("transcript_batch") is not None for mb in micro_batches)
        if fused and homogeneous and len(micro_batches) > 1 and not get_orig_len and not m._encoder_frozen():
            # the log-mel of every micro-batch on its own (zero-padding to ITS longest clip, clamp at ITS maximum - 8: the reference
            # computes one log-mel per batch, module/align_model.py:78-84), then one batch of 30 s windows
            from .whisper_compat import N_FRAMES, pad_or_trim
            with torch.no_grad():
                mels = [m._mel_of(mb["audios"]) for mb in micro_batches]
            if all(x.shape[-1] <= N_FRAMES for x in mels):
                mel = torch.cat([pad_or_trim(x, N_FRAMES) for x in mels], dim=0)
        if mel is None:                                         # not fusable (ragged frame counts, long-form chunks, frozen encoder): the loop
            out = None
            for i, mb in enumerate(micro_batches):
                l = self.micro_step(accum_grad_steps=accum, get_orig_len=get_orig_len, last=i == len(micro_batches) - 1 and accum == len(micro_batches), **mb)
                out = l if out is None else out + l
            return out
        m.train()
        from .decoder_train import cross_entropy
        dev = self.flat[0].device
        out = torch.zeros((4,), dtype=torch.float32, device=dev)
        s = 1.0 / float(accum)
        sizes = [len(mb["audios"]) for mb in micro_batches]
        want_dec = m.train_transcript and all(mb.get("decoder_input") is not None for mb in micro_batches)
        y_in = None
        n_tok = 0
        if want_dec:                                            # one [sum B, n_tok] prompt: shorter micro-batches are padded on the
            n_tok = max(int(mb["decoder_input"].shape[1]) for mb in micro_batches)       # right (causal decoder: earlier positions are
            rows = []                                                                     # unaffected; the padded targets are -100)
            for mb in micro_batches:
                di = mb["decoder_input"]
                pad = n_tok - int(di.shape[1])
                rows.append(torch.nn.functional.pad(di, (0, pad), value=decoder_pad_id) if pad else di)
            y_in = torch.cat(rows, dim=0)
        align_logit, trans_logit = m.frame_manual_forward(None, y_in, get_orig_len=False, mel=mel)
        roots, grads = [], []
        if align_logit is not None and m.train_alignment:
            al = align_logit.detach()
            parts, b0, any_part = [], 0, False
            # The micro-batches' loss kernels are independent and one slice's CTC lattice is two workgroups walking 1500 dependent
            # frames (2.2 ms with the rest of the chip idle): each slice runs on its own stream, joined before the backward.
            cur = torch.cuda.current_stream(dev)
            if len(getattr(self, "_loss_streams", [])) < len(micro_batches):
                self._loss_streams = [torch.cuda.Stream(device=dev) for _ in micro_batches]
            outs = []
            for i, (mb, nb) in enumerate(zip(micro_batches, sizes)):
                st = self._loss_streams[i]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    sl_roots, sl_grads = [], []
                    o_i = torch.zeros((4,), dtype=torch.float32, device=dev)
                    self._align_losses(al[b0:b0 + nb], mb.get("frame_labels"), mb.get("ctc_labels"), s, o_i, sl_roots, sl_grads)
                    part = sl_grads[0] if sl_grads else torch.zeros_like(al[b0:b0 + nb])
                part.record_stream(cur); o_i.record_stream(cur)
                parts.append(part); outs.append(o_i)
                any_part |= bool(sl_grads)
                b0 += nb
            for st in self._loss_streams[: len(micro_batches)]:
                cur.wait_stream(st)
            for o_i in outs:
                out += o_i
            if any_part:
                roots.append(align_logit); grads.append(torch.cat(parts, dim=0))
        if trans_logit is not None and all(mb.get("decoder_output") is not None for mb in micro_batches):
            tl = trans_logit.detach()
            parts, b0 = [], 0
            for mb, nb in zip(micro_batches, sizes):
                do = mb["decoder_output"]
                pad = n_tok - int(do.shape[1])
                do = torch.nn.functional.pad(do, (0, pad), value=-100) if pad else do
                l, dl = cross_entropy(tl[b0:b0 + nb].contiguous(), do, scale_grad=s)
                out[3] += l
                parts.append(dl)
                b0 += nb
            roots.append(trans_logit); grads.append(torch.cat(parts, dim=0))
        if roots:
            if self.overlap is not None and accum == len(micro_batches):    # the one backward of the whole optimizer step
                self.overlap.arm()
            self._backward(roots, grads)
            head_train.check_deferred_flags()                     # GRU time-out flags of a head that ran beside the decoder
        self._check_grad_views()
        return out

    def step(self, allreduced: bool = False) -> torch.Tensor:
        """All-reduce + clip + AdamW + schedule; returns the device scalar sum(grad^2) over the summed buckets.
        The exchange: whatever chunks the last backward already sent off (OverlappedAllReduce) are waited for, the rest is
        all-reduced now; self.allreduce_exposed_ms = what that cost after the overlap.
        allreduced=True: the caller has already run allreduce_mean_(self.grad, self.world) itself."""
        if allreduced and self.overlap is not None and self.overlap.launched_any():
            raise RuntimeError("FineTuner.step(allreduced=True): the last backward already sent gradient chunks off "
                               "(allreduce_chunks > 0); reducing the buckets again would count them twice")
        if not allreduced:
            if self.overlap is not None:
                self.overlap.finish()
            else:
                allreduce_mean_(self.grad, self.world)
        elif self.overlap is not None:
            self.overlap._armed = False
        sumsq = self.opt.step(self.grad, max_norm=self.max_grad_norm, grad_prescale=1.0 / self.world,
                              lr_scale=linear_warmup_scale(self.steps_done, self.warmup_steps, self.train_steps))
        self.steps_done += 1
        for g in self.grad:
            g.zero_()
        # `p.data = flat[...]` gave every Parameter a version counter of its own: the update above went through the flat buffer's raw
        # pointer, so the Parameters are bumped here -- everything keyed on p._version (encoder_train.cached_for, AlignModel.engine(),
        # module/align_model.encoder_only_engine / decoder_engine_of) sees the new weights
        torch.autograd.graph.increment_version([p for g in self.groups for p in g])
        self.model._engine_key = None        # parameters changed under the packed inference weights: re-pack on next use
        return sumsq

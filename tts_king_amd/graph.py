"""hipGraph capture of the whole FS2 train step (forward, loss, backward, clip, Adam, counters).

One step is ~450 short kernel launches; launched eagerly from Python the host, not the GPU, sets the pace.
Everything the step needs that changes from step to step (dropout counters, Adam step, learning rate, BatchNorm
running statistics) lives in device memory and is advanced by kernels, so a captured graph replays correctly with
no host work besides copying the next batch into the static input buffers.  Shapes are static per graph: batches
are bucketed by (B, L, T) and one graph is kept per bucket (T varies per batch in real training).
"""
import torch

from . import ops


class GraphedTrainStep:
    """Captures `enqueue(batch)` for one (B, L, T) bucket and replays it.

    `enqueue` must only enqueue device work on the current stream (no host reads); it returns device tensors
    (e.g. the losses) that are valid after each replay."""

    def __init__(self, enqueue, example_batch, warmup=2, pool=None, static_packed=None, layout=None):
        # `static_packed` / `layout`: the example batch's tensors are already the graph's own static inputs — views of one buffer laid
        # out like the DeviceFeeder's packed batches (tts_king_amd/dataset.py): `run` then takes a batch with one device copy
        self.static_packed, self.layout = static_packed, layout
        self.static = list(example_batch) if static_packed is not None else [t.clone() if torch.is_tensor(t) else t for t in example_batch]
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                enqueue(self.static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # ProcessGroupNCCL's watchdog thread polls the end events of the eager collectives it has not reaped yet.  Under the default
        # capture mode ("global") any HIP call another thread makes while this thread captures is an error — once RCCL's stream has
        # joined the capture such an event query came back as hipErrorCapturedEvent and the watchdog aborted the process (seen once
        # in bench.py's 1-GPU DP leg; round 2 slept 0.5 s before every capture instead).  "thread_local" restricts the capture's
        # legality checks to THIS thread, so the watchdog's queries stay legal whenever they come.  Everything is complete after
        # the synchronize above, so no captured work depends on un-captured work either.
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local" if dist_on else "global"):
            self.outputs = enqueue(self.static)

    def key(self):
        return tuple(tuple(t.shape) for t in self.static if torch.is_tensor(t))

    def takes_packed(self, batch):
        pk = getattr(batch, "packed", None)
        return (self.static_packed is not None and pk is not None and getattr(batch, "layout", None) == self.layout
                and pk.numel() == self.static_packed.numel())

    def run(self, batch=None):
        """Copy `batch` into the static inputs (when given) and replay.  The replay runs none of the step's Python: host-side
        bookkeeping a step would have done (optimizer._host_step, model.grads_partial) is the caller's."""
        if batch is not None:
            if self.takes_packed(batch):
                self.static_packed.copy_(batch.packed, non_blocking=True)
            else:
                for dst, src in zip(self.static, batch):
                    if torch.is_tensor(dst) and src is not dst:
                        dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.outputs


def make_enqueue(model, optimizer, cfg, Loss, step_is_update=True, reducer=None, grad_scale=None, frame_limit=None, phoneme_limit=None,
                 accumulate=None):
    """The device-side part of `main_train_step` (train.py:24-56) as a capturable closure.  `frame_limit`: device int32[1] holding
    the batch's own longest utterance when the batch tensors are padded to a shape bucket (see FastSpeech2._forward).
    `accumulate`: does this micro-step add to the gradient buffer (True: micro-steps 2.. of a grad_acc_step cycle) or overwrite it
    (False: the first one after an update; the optimizer step of this path does not zero the buffer)?  None decides from
    `model.grads_partial` when the closure RUNS — right for eager launches; a captured graph freezes the choice, so whoever replays
    graphs of a grad_acc_step > 1 cycle keeps one for the first micro-step and one for the later ones (TrainEngine does)."""
    grad_acc = cfg.train_config["optimizer"]["grad_acc_step"]
    gs = (1.0 / grad_acc) if grad_scale is None else grad_scale

    def enqueue(batch):
        dev = model.device
        with torch.no_grad():
            ops.stamp("step.start")
            out, ctx = model._forward(True, batch[2], batch[3], batch[4], int(batch[5]), batch[7], batch[8], batch[9],
                                      batch[10], batch[11], 1.0, 1.0, 1.0, frame_limit=frame_limit, phoneme_limit=phoneme_limit,
                                      defer_pred_join=bool(getattr(model, "split_loss", False)))
            ops.stamp("fwd.done")
            mel, pitch, energy, logd = out[0], out[1], out[2], out[3]
            post = out[8]
            lim = None if frame_limit is None else (frame_limit, 0)
            if model._pred_fwd_pending:
                # The predictors ran on a stream of their own and nobody has waited for it yet.  The frame-level half of the loss — all the
                # PostNet's backward needs — goes out on this stream without that wait; the phoneme-level half and the predictors' backward
                # follow the predictors on THEIR stream at once (backward_native joins it where the variance adaptor's gradients meet);
                # the loss values are made behind that join, off the chain.  One launch for everything put the loss behind a cross-queue
                # wait and the PostNet's backward behind another (~10 us each on a replayed graph) plus the finalize kernel.
                side = model._pred_stream
                losses, dmel_sum, dpost, dp, de, dd, pending = ops.fs2_loss_split(mel, post, batch[6], batch[7], pitch, energy, logd, batch[11], batch[9],
                                                                                  batch[10], batch[4], side, grad_scale=gs, frame_limit=lim)
                model._pred_fwd_pending = False
                model._var_on_pred = True
                model._loss_finalize = lambda: ops.fs2_loss_finalize(losses, pending, batch[4])
            else:
                losses, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(mel, post, batch[6], batch[7], pitch, energy, logd, batch[11],
                                                                   batch[9], batch[10], batch[4], grad_scale=gs, frame_limit=lim)
            if reducer is not None and step_is_update:
                model.backward_native(ctx, dmel_sum, dpost, dp, de, dd, on_bucket=reducer.on_group_done, accumulate=accumulate)
                reducer.finish()
            else:
                model.backward_native(ctx, dmel_sum, dpost, dp, de, dd, accumulate=accumulate)
            if step_is_update:
                # the end-of-step dropout-counter tick rides along; the gradients stay (the next backward overwrites them)
                optimizer.step_and_update_lr(advance_rng=True, keep_grads=True)
            else:
                ops.rng_advance(model._state())
            ops.stamp("step.end")
        return losses, out
    return enqueue

"""ScheduledOptim: Adam + the reference's warm-up/anneal learning-rate rule + gradient clipping, as ONE fused
pass over the model's flat parameter buffer.

reference: fs_two/model/optimizer.py:5-53 (wrapper, `_get_lr_scale`), train.py:52 (clip_grad_norm_), torch.optim.Adam.
Step counters, lr and bias corrections live in a device state block and are advanced by a kernel, so the whole
train step can be captured in a hipGraph and replayed without host work.
"""
import numpy as np
import torch

from . import ops


class ScheduledOptim:
    def __init__(self, model, train_config, model_config, current_step):
        opt = train_config["optimizer"]
        self.model = model
        self.betas = tuple(float(b) for b in opt["betas"])
        self.eps = float(opt["eps"])
        if float(opt["weight_decay"]) != 0.0:
            raise NotImplementedError("weight_decay != 0 (the reference config uses 0.0)")
        self.grad_clip_thresh = float(opt["grad_clip_thresh"])
        self.n_warmup_steps = opt["warm_up_step"]
        self.anneal_steps = list(opt["anneal_steps"])
        self.anneal_rate = float(opt["anneal_rate"])
        self.d_model = model_config["transformer"]["encoder_hidden"]
        self.init_lr = np.power(self.d_model, -0.5)
        flat, grad, _ = model.flat_buffers()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        # the device state block (a plain CPU tensor for a CPU-resident model: checkpoint plumbing works there, kernels do not)
        self.state = ops.optim_state(flat.device, seed=getattr(model, "_seed", 1234), sched_step=int(current_step))
        model.attach_state(self.state)
        self._partials = torch.empty(1024, dtype=torch.float32, device=flat.device)
        self._host_step = int(current_step)

    # -- reference surface ------------------------------------------------------------------------
    @property
    def current_step(self):
        """Number of optimizer updates so far (device counter: stays right under hipGraph replay)."""
        return int(self.state[0])

    def _get_lr_scale(self, step=None):
        s = self.current_step if step is None else step
        lr = np.min([np.power(s, -0.5), np.power(self.n_warmup_steps, -1.5) * s])
        for a in self.anneal_steps:
            if s > a:
                lr = lr * self.anneal_rate
        return lr

    def step_and_update_lr(self, advance_rng=False, keep_grads=False):
        """clip (global norm) -> lr update -> Adam -> bf16 shadow (+ window-kernel weight packs) -> grads zeroed, all on device, in two
        launches.  `advance_rng`: also tick the dropout counter (the train step's end-of-step tick, folded in).  `keep_grads`: do not
        zero the gradient buffer (the sync-free step path: its next backward overwrites, FastSpeech2.backward_native); the default
        leaves zeros, as the reference's `zero_grad()` right after the step does (train.py:54)."""
        flat, grad, shadow = self.model.flat_buffers()
        tables = getattr(self.model, "_adam_tables", None) if getattr(self.model, "window_ffn", False) else None
        if tables is not None:
            # the Adam launch writes the window kernels' fragment-major weight packs itself (tile by tile, from LDS)
            ops.optim_step_packed(flat, grad, self.exp_avg, self.exp_avg_sq, shadow, self.state, self._partials, self.grad_clip_thresh,
                                  self.betas[0], self.betas[1], self.eps, self.d_model, self.n_warmup_steps, self.anneal_steps,
                                  self.anneal_rate, tables, zero_grad=not keep_grads, advance_rng=advance_rng)
            if hasattr(self.model, "refresh_odd_packs"):
                self.model.refresh_odd_packs()     # the few packs that launch does not write (the PostNet's 80-channel ends)
        else:
            ops.optim_step(flat, grad, self.exp_avg, self.exp_avg_sq, shadow, self.state, self._partials, self.grad_clip_thresh,
                           self.betas[0], self.betas[1], self.eps, self.d_model, self.n_warmup_steps, self.anneal_steps, self.anneal_rate,
                           zero_grad=not keep_grads, advance_rng=advance_rng)
            self.model.refresh_packed()      # the step rewrote the bf16 shadow: so are the fragment-major copies read by the window conv
        self._host_step += 1
        self.model.grads_partial = False     # zeroed, or stale and to be overwritten: either way the next backward need not accumulate

    def zero_grad(self):
        """reference: ScheduledOptim.zero_grad (optimizer.py:29-30).  After `step_and_update_lr()` the buffer is already zero; in the
        middle of an accumulation cycle this discards it."""
        if getattr(self.model, "grads_partial", False):
            self.model.flat_buffers()[1].zero_()
            self.model.grads_partial = False

    def lr(self):
        return float(self.init_lr * self._get_lr_scale())

    def grad_norm(self):
        """||g|| of the last step (device read)."""
        return float(self.state[6:7].view(torch.float32)[0])

    # -- checkpointing ----------------------------------------------------------------------------------------
    # reference: train.py:221 saves `optimizer._optimizer.state_dict()` = torch.optim.Adam's dict
    #   {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{lr, betas, eps, weight_decay, amsgrad, ..., "params": [0..n-1]}]}
    # with i = index into `model.parameters()` (tts_king_amd.params.reference_parameter_keys) and tensors in the reference
    # shapes ((Cout, Cin, k) conv weights); only parameters that have received a gradient carry state.  The dict written
    # here is that layout (torch.optim.Adam.load_state_dict accepts it) plus one extra top-level key "ttsk" holding what a
    # bit-exact resume of THIS trainer also needs (scheduler step, dropout counters); torch ignores unknown top-level keys.
    def _ref_keys(self):
        from . import params as P
        return P.reference_parameter_keys(self.model._table)

    def state_dict(self):
        from . import params as P
        table = self.model._table
        m, v = self.exp_avg.cpu(), self.exp_avg_sq.cpu()
        st = self.state.cpu().clone()
        adam_t = int(st[1])
        keys = self._ref_keys()
        state = {}
        if adam_t > 0:
            for i, k in enumerate(keys):
                en = table[k]
                if en.kind != P.TRAIN:
                    continue                 # frozen tables / unused CWT heads never get a gradient: Adam holds no state for them
                a = m[en.offset:en.offset + en.numel].view(en.storage_shape)
                b = v[en.offset:en.offset + en.numel].view(en.storage_shape)
                if en.conv:
                    a, b = a.permute(0, 2, 1), b.permute(0, 2, 1)
                state[i] = {"step": torch.tensor(float(adam_t)), "exp_avg": a.contiguous().clone(), "exp_avg_sq": b.contiguous().clone()}
        group = {"lr": self.lr() if adam_t > 0 else 1e-3, "betas": self.betas, "eps": self.eps, "weight_decay": 0.0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(keys)))}
        return {"state": state, "param_groups": [group],
                "ttsk": {"state": st, "current_step": self._host_step}}

    def load_state_dict(self, sd):
        """Accepts the reference layout (with or without the "ttsk" extra) and round 1's flat layout."""
        from . import params as P
        if "exp_avg" in sd:                   # flat layout (checkpoints written by round 1 of this build)
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.state.copy_(sd["state"])
            self._host_step = int(sd["current_step"])
            return
        if "state" not in sd or "param_groups" not in sd:
            raise KeyError("optimizer state: neither torch.optim.Adam's {'state','param_groups'} nor the flat layout")
        table = self.model._table
        keys = self._ref_keys()
        if len(sd["param_groups"][0]["params"]) != len(keys):
            raise ValueError("optimizer state has %d parameters, the model %d" % (len(sd["param_groups"][0]["params"]), len(keys)))
        m, v = torch.zeros(self.exp_avg.shape), torch.zeros(self.exp_avg_sq.shape)
        adam_t = 0
        for i, ent in sd["state"].items():
            en = table[keys[int(i)]]
            if en.kind != P.TRAIN:
                continue
            a, b = ent["exp_avg"].float(), ent["exp_avg_sq"].float()
            if en.conv:
                a, b = a.permute(0, 2, 1), b.permute(0, 2, 1)
            m[en.offset:en.offset + en.numel] = a.reshape(-1)
            v[en.offset:en.offset + en.numel] = b.reshape(-1)
            adam_t = max(adam_t, int(float(ent["step"])))
        self.exp_avg.copy_(m)
        self.exp_avg_sq.copy_(v)
        extra = sd.get("ttsk")
        if extra is not None and extra.get("state") is not None:
            self.state.copy_(extra["state"])
            self._host_step = int(extra["current_step"])
        else:
            # a checkpoint written by the reference: Adam's step comes from the state, the scheduler step from
            # cfg.tts.restore_step (already in the device block, as in the reference: optimizer.py:19)
            st = self.state.cpu()
            st[1] = adam_t
            self.state.copy_(st)

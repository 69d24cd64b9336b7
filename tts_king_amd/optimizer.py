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
        self.state = ops.optim_state(flat.device, seed=getattr(model, "_seed", 1234), sched_step=int(current_step)) \
            if flat.is_cuda else None
        if self.state is not None:
            model.attach_state(self.state)
        self._partials = torch.empty(1024, dtype=torch.float32, device=flat.device)
        self._host_step = int(current_step)

    # -- reference surface ------------------------------------------------------------------------
    @property
    def current_step(self):
        """Number of optimizer updates so far (device counter: stays right under hipGraph replay)."""
        return int(self.state[0]) if self.state is not None else self._host_step

    def _get_lr_scale(self, step=None):
        s = self.current_step if step is None else step
        lr = np.min([np.power(s, -0.5), np.power(self.n_warmup_steps, -1.5) * s])
        for a in self.anneal_steps:
            if s > a:
                lr = lr * self.anneal_rate
        return lr

    def step_and_update_lr(self):
        """clip (global norm) -> lr update -> Adam -> bf16 shadow refresh -> grads zeroed, all on device."""
        flat, grad, shadow = self.model.flat_buffers()
        ops.optim_advance(self.state, self.d_model, self.n_warmup_steps, self.anneal_steps, self.anneal_rate, *self.betas)
        ops.clip_adam_step(flat, grad, self.exp_avg, self.exp_avg_sq, shadow, self.state, self._partials,
                           self.grad_clip_thresh, self.betas[0], self.betas[1], self.eps, zero_grad=True)
        self._host_step += 1

    def zero_grad(self):
        pass    # the fused step leaves the flat gradient buffer zeroed

    def lr(self):
        return float(self.init_lr * self._get_lr_scale())

    def grad_norm(self):
        """||g|| of the last step (device read)."""
        return float(self.state[6:7].view(torch.float32)[0])

    # -- checkpointing (torch.optim.Adam-shaped dict, reference: train.py:212-227 saves it) ------------------
    def state_dict(self):
        return {"exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "state": self.state.cpu(),
                "current_step": self._host_step}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.state.copy_(sd["state"])
        self._host_step = int(sd["current_step"])

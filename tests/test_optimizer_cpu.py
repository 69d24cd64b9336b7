"""ScheduledOptim.state_dict() is torch.optim.Adam's layout indexed like the reference's `model.parameters()`
(reference: fs_two/model/optimizer.py:10-15, train.py:221), and round-trips through a real torch.optim.Adam."""
import numpy as np
import torch

from tts_king_amd import params as P
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.optimizer import ScheduledOptim


def test_reference_parameter_order_matches_reference_state_dict(cfg, golden_dir):
    spec = np.load(golden_dir + "/fs2_state_dict_spec.npz")
    want = [str(k) for k in spec["keys"] if "running_" not in str(k) and "num_batches" not in str(k)]
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device="cpu")
    assert P.reference_parameter_keys(m._table) == want


def test_adam_state_dict_round_trips_through_torch_adam(cfg):
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device="cpu")
    opt = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
    g = torch.Generator().manual_seed(3)
    opt.exp_avg.copy_(torch.randn(opt.exp_avg.shape, generator=g))
    opt.exp_avg_sq.copy_(torch.rand(opt.exp_avg_sq.shape, generator=g))
    opt.state[0], opt.state[1] = 7, 7
    opt._host_step = 7
    sd = opt.state_dict()
    keys = P.reference_parameter_keys(m._table)
    # a torch Adam over tensors of the reference shapes, in the reference order, accepts the dict as its own
    ref_params = []
    for k in keys:
        en = m._table[k]
        ref_params.append(torch.nn.Parameter(torch.zeros(en.shape), requires_grad=en.kind == P.TRAIN))
    adam = torch.optim.Adam(ref_params, betas=opt.betas, eps=opt.eps, weight_decay=0.0)
    adam.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
    st = adam.state_dict()["state"]
    i = keys.index("decoder.layer_stack.2.pos_ffn.w_1.weight")
    en = m._table[keys[i]]
    assert tuple(st[i]["exp_avg"].shape) == (1024, 256, 9)                  # reference conv layout (Cout, Cin, k)
    flat = opt.exp_avg[en.offset:en.offset + en.numel].view(1024, 9, 256).permute(0, 2, 1)
    assert torch.equal(st[i]["exp_avg"], flat)
    assert float(st[i]["step"]) == 7.0
    assert keys.index("encoder.position_enc") not in st and keys.index("variance_adaptor.pitch_mean.linear.weight") not in st
    # and back: what torch Adam writes (no "ttsk" extra, as a checkpoint saved by the reference) restores the moments
    opt2 = ScheduledOptim(m, cfg.train_config, cfg.model_config, 7)
    opt2.load_state_dict(adam.state_dict())
    def same(a, b):       # every parameter's slice (the flat buffer's 16-byte alignment gaps belong to no parameter)
        return all(torch.equal(a[en.offset:en.offset + en.numel], b[en.offset:en.offset + en.numel])
                   for en in m._table.values() if en.kind == P.TRAIN)
    assert same(opt2.exp_avg, opt.exp_avg) and same(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert int(opt2.state[1]) == 7 and opt2.current_step == 7
    # with the extra block the device counters come back too
    opt3 = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
    opt3.load_state_dict(sd)
    assert torch.equal(opt3.state, opt.state) and same(opt3.exp_avg_sq, opt.exp_avg_sq)

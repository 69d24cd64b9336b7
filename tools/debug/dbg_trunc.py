import os, sys, copy
import torch
sys.path.insert(0, os.getcwd())
from oracle import fs2 as ofs2
from tests.oracle_util import fs2_state_dict, rel_rms
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.synthetic import make_batch
cfg = default_config(); DEV = "cuda:0"
mc0 = copy.deepcopy(cfg.model_config)
mc0["transformer"]["encoder_dropout"] = mc0["transformer"]["decoder_dropout"] = 0.0
mc0["variance_predictor"]["dropout"] = 0.0
ofs2._drop = lambda x, p, train: x
sd = fs2_state_dict(cfg, 7)
for (B, L, dur_hi, seed) in ((2, 200, 12, 77), (2, 200, 9, 77), (2, 100, 12, 5), (1, 200, 12, 77)):
    b = make_batch(B, L, seed=seed, ragged=True, dur_hi=dur_hi)
    with torch.no_grad():
        o = ofs2.fs2_forward(sd, mc0, *b[2:], train=True, bn_buffers={})
    for name, kw in (("default", {}), ("no fused_ln", {"fused_ln": False}), ("no flash attn", {"flash_attention": False}), ("no grouped pred", {"group_predictors": False})):
        m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=DEV)
        m.load_state_dict(sd); m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0; m.train()
        for k, v in kw.items(): setattr(m, k, v)
        dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]
        with torch.no_grad():
            out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
        mel = out[0].float().cpu()
        T = mel.shape[1]
        per_utt = [rel_rms(mel[i, :min(int(b[7][i]), T)], o[0][i, :min(int(b[7][i]), T)]) for i in range(B)]
        print("B=%d L=%d T_full=%d T=%d %-16s mel rel-RMS %.3f%% per-utt valid rows %s  pitch %.3f%% logd %.3f%%" % (
            B, L, int(b[8]), T, name, 100 * rel_rms(mel, o[0]), ["%.2f%%" % (100 * r) for r in per_utt],
            100 * rel_rms(out[1].cpu(), o[1]), 100 * rel_rms(out[3].cpu(), o[3])))

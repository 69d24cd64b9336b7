"""Helpers shared by the tests: build the oracle's state_dict for FS2 / HiFi-GAN from the committed
key/shape specs (tests/golden/*_state_dict_spec.npz) and the deterministic fill."""
import json
import os

import numpy as np
import torch

from oracle import fs2 as ofs2
from tts_king_amd.synthetic import seeded_fill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _shape(s):
    return tuple(int(x) for x in s.split(";")) if s else ()


def fs2_state_dict(cfg, weight_seed, n_speakers=65):
    spec = np.load(os.path.join(GOLDEN, "fs2_state_dict_spec.npz"))
    sd = {}
    for k, s, dt in zip(spec["keys"], spec["shapes"], spec["dtypes"]):
        shape = _shape(str(s))
        sd[str(k)] = torch.zeros(shape, dtype=torch.int64 if "int64" in str(dt) else torch.float32)
    mc = cfg.model_config
    d = mc["transformer"]["encoder_hidden"]
    tab = ofs2.sinusoid_table(mc["max_seq_len"] + 1, d)[None]
    sd["encoder.position_enc"] = tab.clone()
    sd["decoder.position_enc"] = tab.clone()
    with open(os.path.join(cfg.preprocess_config.path.preprocessed_path, "stats.json")) as f:
        stats = json.load(f)
    nb = mc["variance_embedding"]["n_bins"]
    sd["variance_adaptor.pitch_bins"] = torch.linspace(stats["pitch"][0], stats["pitch"][1], nb - 1)
    sd["variance_adaptor.energy_bins"] = torch.linspace(stats["energy"][0], stats["energy"][1], nb - 1)
    if sd["speaker_emb.weight"].shape[0] != n_speakers:
        sd["speaker_emb.weight"] = torch.zeros(n_speakers, d)
    seeded_fill(sd, weight_seed)
    return sd


def hifi_state_dict_wn(weight_seed):
    spec = np.load(os.path.join(GOLDEN, "hifi_state_dict_spec.npz"))
    sd = {str(k): torch.zeros(_shape(str(s))) for k, s in zip(spec["wn_keys"], spec["wn_shapes"])}
    seeded_fill(sd, weight_seed)
    return sd


def rel_rms(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float(((a - b).pow(2).mean() / b.pow(2).mean().clamp_min(1e-30)).sqrt())

"""FSTWOapi — the reference's FastSpeech2 synthesis facade (reference: fsapi.py:9-96) over the MI355X model.

`tts.weights_path: null` (this repo ships no checkpoint) builds a seeded random-init model and leaves
`preprocess_config.path.preprocessed_path` as configured; with a path the reference rule applies (the checkpoint's
folder holds speakers.json / stats.json, fsapi.py:11-17).
"""
import json
import os

import numpy as np
import torch

from tts_king_amd.fastspeech2 import FastSpeech2


class FSTWOapi:
    def __init__(self, config, device=0):
        weights_path = config.tts.weights_path
        if weights_path is not None:
            model_folder = "/".join(weights_path.split("/")[:-1])
            config.preprocess_config.path.preprocessed_path = model_folder
        self.speakers_dict, self.speaker_names = load_speakers_json(config.preprocess_config.path.preprocessed_path)
        if isinstance(device, int):
            device = "cuda:%d" % device
        if not str(device).startswith("cuda"):
            # the reference's shipped default (config.yaml:2 `gpu: 'cpu'`): this build has no CPU path — say so here, not at the first forward
            from tts_king_amd.lib import TtskError
            raise TtskError("gpu: %r — this build runs FastSpeech2 on hand-written MI355X kernels only; set gpu: 'cuda:0' in config.yaml "
                            "(PyTorch-ROCm names HIP devices cuda:N)" % (device,))
        self.model = FastSpeech2(config.preprocess_config, config.model_config, len(self.speaker_names), device=device)
        self.weights_path = weights_path
        if weights_path is not None:
            checkpoint = torch.load(weights_path, map_location="cpu")
            state = checkpoint["model"]
            state["speaker_emb.weight"] = checkpoint["embedding"]
            self.model.load_state_dict(state)
        self.cfg = config
        self.device = device
        self.restore_step = 0
        mi = config.get("mi355x", {}) if hasattr(config, "get") else {}
        self._synth = None
        if mi and mi.get("hip_graph", False) and str(device).startswith("cuda"):
            from tts_king_amd.synth import GraphedSynthesizer
            self._synth = GraphedSynthesizer(self.model)

    def generate(self, phonemes, duration_control=1.0, pitch_control=1.0, energy_control=1.0, speaker_name=None):
        """phonemes: int ndarray (1, L) -> postnet mel (1, T, 80) fp32 on the device.  reference: fsapi.py:38-82."""
        if speaker_name is not None:
            if speaker_name not in self.speakers_dict:
                raise Exception(f"Speaker {speaker_name} was not found in speakers.json")
            speaker_id = self.speakers_dict[speaker_name]
        else:
            speaker_id = 0          # the reference leaves `speaker` unbound here (NameError); default to the first speaker
        speaker = torch.tensor(speaker_id).long().unsqueeze(0).to(self.device)
        self.model.eval()
        phonemes = np.asarray(phonemes)
        if self._synth is not None:      # hipGraph-replayed path (tts_king_amd/synth.py): same kernels, no launch overhead
            post, _ = self._synth.mel(speaker, torch.from_numpy(phonemes).long().to(self.device), pitch_control, energy_control,
                                      duration_control)
            return post
        src_len = np.array([len(phonemes[0])])
        result = self.model(speaker, torch.from_numpy(phonemes).long().to(self.device), torch.from_numpy(src_len).to(self.device),
                            max(src_len), d_control=duration_control, p_control=pitch_control, e_control=energy_control)
        postnet_output = result[9]
        return postnet_output


def load_speakers_json(dir_path):
    """reference: fsapi.py:85-96."""
    json_path = os.path.join(dir_path, "speakers.json")
    if not os.path.exists(json_path):
        raise FileNotFoundError(f"Did not find speakers.json at {dir_path}")
    with open(json_path, "r") as f:
        speakers = json.load(f)
    return speakers, list(speakers.keys())

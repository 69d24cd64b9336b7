"""TTSKing — the reference's top-level synthesis surface (reference: tts_king.py:18-66) on MI355X.

`TTSKing(config_path)`, `.generate_mel(text, d, p, e, speaker)`, `.mel_to_wav(mel)`, `.speakers`, `.text_preprocess`,
`.to_torch_device` keep the reference signatures.  The text frontend (russian_g2p / g2p_en, fs_two/text) is outside this
build's G2P dependency list: `text` may be a phoneme-id array (1, L) (what `text_preprocess` returns in the reference), a
phoneme string "{R A B O0 T ...}" (tts_king_amd/text.py, pinned by the notebook's known-answer vector), or — when the
optional `russian_g2p` package is importable — plain Russian text.
"""
import numpy as np
import torch

from tts_king_amd.config import load_config
from fsapi import FSTWOapi
from hifiapi import HIFIapi


class TTSKing:
    def __init__(self, config_path="./config.yaml"):
        self.cfg = load_config(config_path)
        self.tts = FSTWOapi(self.cfg, self.cfg.gpu)
        self.vocoder = HIFIapi(self.cfg, self.cfg.gpu)
        self.speakers = self.tts.speaker_names

    def generate_mel(self, text, duration_control=1.0, pitch_control=1.0, energy_control=1.0, speaker=0):
        phonemes = text if isinstance(text, np.ndarray) else self.text_preprocess(text)
        if isinstance(speaker, int):
            speaker = self.speakers[speaker]
        return self.tts.generate(phonemes, duration_control, pitch_control, energy_control, speaker_name=speaker)

    def mel_to_wav(self, mel_spec):
        """(1, T, 80) mel -> int16 ndarray (1, 1, 256 T).  reference: tts_king.py:47-49."""
        return self.vocoder.generate(mel_spec.transpose(1, 2))

    def speak(self, text, duration_control=1.0, pitch_control=1.0, energy_control=1.0, speaker=0):
        """reference: tts_king.py:51-57 calls a missing `generate_mel_batch`; here: mel -> float waveform."""
        mel = self.generate_mel(text, duration_control, pitch_control, energy_control, speaker)
        return self.vocoder(mel.transpose(1, 2))

    def text_preprocess(self, text):
        """reference: tts_king.py:59-60 -> input_process.preprocess_rus (needs russian_g2p).  A string that already is in
        the phoneme notation of the reference's frontend, "{R A B O0 T ...}", is converted directly."""
        from input_process import preprocess_rus
        from tts_king_amd.text import text_to_sequence
        if "{" in text:
            return np.array([text_to_sequence(text, [])])
        return np.array([preprocess_rus(text)])

    def text_preprocess_eng(self, text):
        """reference: tts_king.py:62-63."""
        from input_process import preprocess_eng
        return np.array([preprocess_eng(text, self.cfg.preprocess_config)])

    def to_torch_device(self, items):
        return [torch.tensor(t).to(self.cfg.gpu) for t in items]

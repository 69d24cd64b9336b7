"""HIFIapi — the reference's vocoder facade (reference: hifiapi.py:11-52) over the MI355X HiFi-GAN generator.

Same constructor, attributes (`.model .cfg .device`), `__call__(x)` (float waveform, (B,1,256T)) and
`generate(mel_specs)` (int16 ndarray, C truncation toward zero after *MAX_WAV_VALUE).  The generator runs only on a
HIP device: `model_config.vocoder.use_cpu: true` (the reference default) is rejected loudly, there is no CPU path.
"""
import torch

from tts_king_amd import ops
from tts_king_amd.hifigan import Generator


class AttrDict(dict):
    """reference: hifiapi.py:5-8."""

    def __init__(self, *args, **kwargs):
        super(AttrDict, self).__init__(*args, **kwargs)
        self.__dict__ = self


def _device_of(device):
    if device in ("gpu", None):
        return torch.device("cuda:0")
    if isinstance(device, int):
        return torch.device("cuda:%d" % device)
    return torch.device(device)


class HIFIapi:
    def __init__(self, config, device="gpu"):
        if config.model_config["vocoder"]["use_cpu"]:
            raise ops.L.TtskError("model_config.vocoder.use_cpu: true — this build runs the HiFi-GAN generator on hand-written "
                                  "MI355X kernels only; set use_cpu: false and gpu: 'cuda:0'")
        device = _device_of(device)
        if device.type != "cuda":
            raise ops.L.TtskError("HIFIapi needs a HIP device (gpu: 'cuda:0'), got %s" % device)
        weights_path = config.hifi.weights_path
        self.model = Generator(config.hifi)
        if weights_path is not None:
            checkpoint = torch.load(weights_path, map_location="cpu")
            self.model.load_state_dict(checkpoint["generator"])
        else:
            self.model.reset_parameters(int(config.hifi.get("seed", 1234)))      # no checkpoint ships with the repo
        self.cfg = config
        self.device = device
        self.model.to(device)
        self.model.remove_weight_norm()
        self.model.eval()
        mi = config.get("mi355x", {}) if hasattr(config, "get") else {}
        self._synth = None
        if mi and mi.get("hip_graph", False):
            from tts_king_amd.synth import GraphedSynthesizer
            self._synth = GraphedSynthesizer(None, self.model)

    def train(self):
        """reference: hifiapi.py:32-33 raises (`NotImplemented(...)` is not callable -> TypeError there)."""
        raise NotImplementedError(" Train for HiFi was not implemented yet")

    def __call__(self, x):
        x = x.to(self.device)
        return self.model(x)

    def generate(self, mel_specs):
        """mel (B,80,T) -> int16 ndarray (B,1,256T) on the host.  reference: hifiapi.py:40-52."""
        self.model.eval()
        with torch.no_grad():
            mel_specs = mel_specs.to(self.device)
            audio = self._synth.wav(mel_specs.float()) if self._synth is not None else self.model(mel_specs)
            audio = ops.to_int16(audio, float(self.cfg.hifi.MAX_WAV_VALUE))     # scale + truncate toward zero on device
            audio = ops.to_host(audio).numpy()                                     # D2H through a pinned staging buffer
        return audio

"""CPU: the two CPU defaults of the reference's config.yaml (`gpu: 'cpu'`, config.yaml:2; `model_config.vocoder.use_cpu: true`,
config.yaml:127) are rejected at construction with a message that says what to edit — there is no CPU path."""
import copy

import pytest


def test_reference_cpu_defaults_are_rejected_loudly(cfg):
    import fsapi
    import hifiapi
    from tts_king_amd.lib import TtskError
    with pytest.raises(TtskError, match="set gpu: 'cuda:0'"):
        fsapi.FSTWOapi(copy.deepcopy(cfg), "cpu")
    c = copy.deepcopy(cfg)
    c.model_config["vocoder"]["use_cpu"] = True
    with pytest.raises(TtskError, match="use_cpu: false"):
        hifiapi.HIFIapi(c, "cuda:0")

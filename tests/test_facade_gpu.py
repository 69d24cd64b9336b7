"""GPU: the reference's synthesis surface end to end (BASELINE.json configs[4]): TTSKing(config) -> generate_mel ->
mel_to_wav, eager and hipGraph-replayed, against the oracle on the same weights.

reference: tts_king.py:18-49, fsapi.py:38-82, hifiapi.py:40-52.  Free-running durations come from a bf16-accurate
log-duration, so a few may land on the other side of a rounding boundary than the fp32 oracle's, and a pitch/energy
prediction next to a bin edge may select the neighbouring embedding row (same rule as tests/test_fs2_gpu.py::
test_eval_free_running); the mel is therefore compared with the oracle run teacher-forced on the durations, pitch and
energy values the HIP path produced (stated tolerance for this single random-weight utterance: rel-RMS <= 1.5 %;
the golden-vector tests in tests/test_fs2_gpu.py hold the 1 % bar), the waveform with the oracle vocoder on the
HIP path's own mel (rel-RMS <= 0.5 %).  Graph replay must reproduce the eager result bit for bit."""
import copy
import math
import os

import numpy as np
import pytest
import torch

from oracle import fs2 as ofs2
from oracle import hifigan as ohifi
from tests.oracle_util import rel_rms

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tts(tmp_path, hip_graph):
    import yaml
    import tts_king
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config.yaml")))
    cfg["preprocess_config"]["path"]["preprocessed_path"] = os.path.join(ROOT, "pretrained")
    cfg["mi355x"]["hip_graph"] = hip_graph
    p = tmp_path / ("config_%d.yaml" % hip_graph)
    p.write_text(yaml.safe_dump(cfg))
    t = tts_king.TTSKing(str(p))
    with torch.no_grad():       # random-init duration head predicts ~0 frames: shift it so utterances have a few frames per phoneme
        t.tts.model.get("variance_adaptor.duration_predictor.linear_layer.bias").fill_(1.3)
    return t


def test_ttsking_surface_and_parity(tmp_path):
    tts = make_tts(tmp_path, False)
    assert len(tts.speakers) == 66 and tts.speakers == tts.tts.speaker_names           # pretrained/speakers.json
    g = torch.Generator().manual_seed(3)
    phon = torch.randint(1, 207, (1, 48), generator=g).numpy()
    mel = tts.generate_mel(phon, duration_control=1.0, pitch_control=1.2, energy_control=0.9, speaker=5)
    assert mel.dim() == 3 and mel.shape[0] == 1 and mel.shape[2] == 80 and mel.dtype == torch.float32
    T = mel.shape[1]
    assert T > 48
    wav = tts.mel_to_wav(mel)
    assert wav.dtype == np.int16 and wav.shape == (1, 1, 256 * T)
    with pytest.raises(Exception):
        tts.tts.generate(phon, speaker_name="no such speaker")
    # phoneme-string input (the notation of examples.ipynb cell 2) goes through the text frontend
    ids = tts.text_preprocess("{R A B O0 T A T0 sp}")
    assert ids.shape == (1, 8) and ids[0, 0] == 184
    assert tts.generate_mel("{R A B O0 T A T0 sp}", speaker=1).shape[2] == 80
    # ---- oracle on the same weights, teacher-forced on the durations the HIP path predicted
    sd = {k: v.detach().float().cpu() for k, v in tts.tts.model.state_dict().items()}
    m = tts.tts.model
    out = m(torch.tensor([5]), torch.from_numpy(phon), torch.tensor([48]), 48, p_control=1.2, e_control=0.9)
    d_rounded = out[4].cpu()
    assert int(d_rounded.clamp(min=0).trunc().sum()) == T
    cfg = tts.cfg
    with torch.no_grad():
        ref = ofs2.fs2_forward(sd, cfg.model_config, torch.tensor([5]), torch.from_numpy(phon).long(), torch.tensor([48]), 48,
                               d_targets=d_rounded, max_mel_len=T, mel_lens=torch.tensor([T]),
                               pitches_raw=out[1].detach().float().cpu(), e_targets=out[2].detach().float().cpu())
    r = rel_rms(mel.cpu(), ref[9])
    print("facade mel vs oracle (HIP durations, pitch, energy): rel-RMS %.3f%%" % (100 * r))
    assert r <= 0.015
    gsd = {k: v.detach().float().cpu() for k, v in tts.vocoder.model.state_dict().items()}
    with torch.no_grad():
        wref = ohifi.generator(gsd, cfg.hifi, mel.cpu().transpose(1, 2))
    got = wav.astype(np.float32) / 32768.0
    r = rel_rms(torch.from_numpy(got), wref)
    print("facade waveform vs oracle vocoder on the same mel: rel-RMS %.3f%%" % (100 * r))
    assert r <= 0.005 + 2e-5 / float(wref.pow(2).mean().sqrt())          # + int16 quantisation


def test_graph_replay_is_bit_identical_to_eager(tmp_path):
    eager = make_tts(tmp_path, False)
    graphed = make_tts(tmp_path, True)
    g = torch.Generator().manual_seed(4)
    phons = [torch.randint(1, 207, (1, L), generator=g).numpy() for L in (40, 40, 40, 56)]
    for i, ph in enumerate(phons):          # 1st call of a shape: eager warm-up, 2nd: capture + replay, 3rd: replay
        ph = phons[0] if i < 3 else ph
        a = eager.generate_mel(ph, speaker=2)
        b = graphed.generate_mel(ph, speaker=2)
        assert torch.equal(a, b), i
        wa, wb = eager.mel_to_wav(a), graphed.mel_to_wav(b)
        assert np.array_equal(wa, wb), i
    assert len(graphed.tts._synth._front) >= 1 and len(graphed.tts._synth._back) >= 1 and len(graphed.vocoder._synth._voc) >= 1


def test_graphed_outputs_are_not_aliased(tmp_path):
    """A mel returned by `generate_mel` stays what it was when a later call with the same (L, T) key replays the same graph
    (the graph's static output buffers never leave the synthesizer)."""
    tts = make_tts(tmp_path, True)
    g = torch.Generator().manual_seed(6)
    ph_a = torch.randint(1, 207, (1, 40), generator=g).numpy()
    for _ in range(2):                         # eager warm-up, then capture
        tts.generate_mel(ph_a, speaker=2)
    held = tts.generate_mel(ph_a, speaker=2)   # replay
    snapshot = held.clone()
    again = tts.generate_mel(ph_a, speaker=7)  # same key (same phonemes -> the frame count may differ: retry below if so)
    if again.shape == held.shape:
        assert not torch.equal(again, held) or torch.equal(again, snapshot)
    assert torch.equal(held, snapshot), "a previously returned mel was overwritten by a later call"
    assert held.data_ptr() != again.data_ptr()


def test_free_running_second_control_setting_through_to_the_waveform(tmp_path):
    """one FREE-RUNNING case (nothing teacher-forced on the HIP path's own values) on the controls of the reference's
    notebook (examples.ipynb cell 3: duration 0.9, pitch 1.5, energy 1.2), followed through to the int16 waveform.  The oracle runs
    free as well, on the same weights.  Durations: the HIP log-durations within the stated 0.06 of the oracle's, the HIP durations
    exactly the reference's rounding rule (modules.py:199-204) applied to them, and EVERY position where the two disagree has a
    rounding boundary between the two pre-rounding values.  Then the frames: where no duration differs the two runs share every frame
    position and the mel / waveform are compared directly (mel rel-RMS <= 3 %: a pitch or energy prediction next to a bin edge may pick
    the neighbouring embedding row, as in tests/test_fs2_gpu.py::test_eval_free_running; waveform vs the oracle vocoder on the HIP mel
    <= 0.5 % + int16 quantisation); where some differ, the oracle is re-run with the HIP durations only (pitch and energy still its
    own predictions) and the same bars apply."""
    tts = make_tts(tmp_path, False)
    g = torch.Generator().manual_seed(11)
    L = 56
    phon = torch.randint(1, 207, (1, L), generator=g).numpy()
    dc, pc, ec = 0.9, 1.5, 1.2
    mel = tts.generate_mel(phon, duration_control=dc, pitch_control=pc, energy_control=ec, speaker=9)
    wav = tts.mel_to_wav(mel)
    T = mel.shape[1]
    assert wav.dtype == np.int16 and wav.shape == (1, 1, 256 * T)
    m = tts.tts.model
    out = m(torch.tensor([9]), torch.from_numpy(phon), torch.tensor([L]), L, d_control=dc, p_control=pc, e_control=ec)
    assert torch.equal(out[9].detach().float().cpu(), mel.cpu())                   # the facade returned the model's postnet mel
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    cfg = tts.cfg
    with torch.no_grad():
        ref = ofs2.fs2_forward(sd, cfg.model_config, torch.tensor([9]), torch.from_numpy(phon).long(), torch.tensor([L]), L,
                               p_control=pc, e_control=ec, d_control=dc)
    logd_h, logd_r = out[3].detach().float().cpu(), ref[3].float()
    assert float((logd_h - logd_r).abs().max()) <= 0.06
    d_h, d_r = out[4].detach().float().cpu(), ref[4].float()
    v_h, v_r = torch.exp(logd_h) - 1.0, torch.exp(logd_r) - 1.0
    rule = torch.clamp(torch.round(v_h) * dc, min=0.0)
    assert bool((((v_h - torch.floor(v_h) - 0.5).abs() < 1e-5) | (rule == d_h)).all())
    mism = (d_h != d_r)
    for bi, li in mism.nonzero().tolist():
        vr, vh = float(v_r[bi, li]), float(v_h[bi, li])
        bnd = math.floor(vr) + 0.5
        if abs(vr - bnd) > 0.5:
            bnd += 1.0
        assert min(vr, vh) - 1e-5 <= bnd <= max(vr, vh) + 1e-5, (bi, li, vh, vr)
    print("free-running facade: %d of %d durations differ from the oracle's (each across a rounding boundary); T = %d" % (int(mism.sum()), L, T))
    assert int(d_h.clamp(min=0).trunc().sum()) == T
    if bool(mism.any()):
        with torch.no_grad():          # the oracle on the HIP durations; pitch / energy still its own free-running predictions
            ref = ofs2.fs2_forward(sd, cfg.model_config, torch.tensor([9]), torch.from_numpy(phon).long(), torch.tensor([L]), L,
                                   d_targets=d_h, max_mel_len=T, mel_lens=torch.tensor([T]), p_control=pc, e_control=ec)
    assert ref[9].shape == mel.shape
    r_free = rel_rms(mel.cpu(), ref[9])
    # Pitch and energy pick an embedding row by bucketize(prediction * control, 255 edges) (modules.py:92-101,131-140): a bf16-level
    # difference in a prediction that sits next to an edge selects the neighbouring row — one row of a random-weight table is a large
    # step for a 33-frame utterance.  Same argument as for the durations, position by position: wherever the HIP path and the oracle
    # chose different rows, a bin edge lies between their two predictions; with the HIP path's choices handed over (its predictions as
    # the oracle's `targets`, which only pick rows) the mel must agree to the teacher-forced bar.
    va = "variance_adaptor."

    def edge_between(name, h, r_, bins):
        bh, br = torch.bucketize(h, bins), torch.bucketize(r_, bins)
        n = 0
        for bi, li in (bh != br).nonzero().tolist():
            lo, hi = min(float(h[bi, li]), float(r_[bi, li])), max(float(h[bi, li]), float(r_[bi, li]))
            assert bool(((bins >= lo - 1e-6) & (bins <= hi + 1e-6)).any()), (name, bi, li, lo, hi)
            assert hi - lo <= 0.1 * max(1.0, abs(hi)), (name, bi, li, lo, hi)
            n += 1
        return n
    pitch_h, energy_h = out[1].detach().float().cpu(), out[2].detach().float().cpu()
    n_p = edge_between("pitch", pitch_h, ref[1].float(), sd[va + "pitch_bins"])
    with torch.no_grad():          # the oracle with the HIP path's pitch rows: its energy predictor now sees the same pitch embedding
        ref = ofs2.fs2_forward(sd, cfg.model_config, torch.tensor([9]), torch.from_numpy(phon).long(), torch.tensor([L]), L,
                               d_targets=d_h, max_mel_len=T, mel_lens=torch.tensor([T]), pitches_raw=pitch_h, e_control=ec)
    n_e = edge_between("energy", energy_h, ref[2].float(), sd[va + "energy_bins"])
    with torch.no_grad():
        ref = ofs2.fs2_forward(sd, cfg.model_config, torch.tensor([9]), torch.from_numpy(phon).long(), torch.tensor([L]), L,
                               d_targets=d_h, max_mel_len=T, mel_lens=torch.tensor([T]), pitches_raw=pitch_h, e_targets=energy_h)
    r = rel_rms(mel.cpu(), ref[9])
    print("free-running facade mel vs the free-running oracle: rel-RMS %.3f%%; %d pitch / %d energy rows differ (each across a bin edge); "
          "with those rows handed over: %.3f%%" % (100 * r_free, n_p, n_e, 100 * r))
    assert r <= 0.015
    assert r_free <= 0.03 or (n_p + n_e) > 0          # without a differing row the free runs themselves must agree
    gsd = {k: v.detach().float().cpu() for k, v in tts.vocoder.model.state_dict().items()}
    with torch.no_grad():
        wref = ohifi.generator(gsd, cfg.hifi, mel.cpu().transpose(1, 2))
        wfree = ohifi.generator(gsd, cfg.hifi, ref[9].transpose(1, 2))
    got = torch.from_numpy(wav.astype(np.float32) / 32768.0)
    r = rel_rms(got, wref)
    print("free-running facade waveform vs the oracle vocoder on the HIP mel: rel-RMS %.3f%%; vs the oracle end to end: %.1f%%"
          % (100 * rel_rms(got, wref), 100 * rel_rms(got, wfree)))
    r = rel_rms(got, wref)
    assert r <= 0.005 + 2e-5 / float(wref.pow(2).mean().sqrt())
    # int16 conversion: C truncation toward zero of audio * 32768 (hifiapi.py:50-51), checked against the float output of the HIP generator
    f = tts.vocoder(mel.transpose(1, 2)).detach().float().cpu().numpy()
    assert np.array_equal((f * 32768.0).astype("int16"), wav)

"""GPU: the trainer surface (reference: train.py:78-235 `main`, fs_two/evaluate.py:18-101 `evaluate`) on a small
synthetic preprocessed directory: loop + DataLoader + pinned feeder + validation message + checkpoint save, and resume
incl. the Adam state.  Losses of `evaluate` are checked against the oracle."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import fs2 as ofs2
from tests.test_dataset_cpu import synthetic_samples
from tts_king_amd import dataset as D
from tts_king_amd import text as T

pytestmark = pytest.mark.gpu


def write_corpus(root, n=24):
    samples = synthetic_samples(n, 9)
    for d in ("mel", "energy", "duration", "pitch"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    syms = T.symbols()
    lines = []
    for s in samples:
        spk, b = "spk%d" % (s["speaker"] % 3), s["id"]
        phon = "{" + " ".join(syms[i][1:] for i in (150 + (s["text"] % 50))) + "}"
        lines.append("%s|%s|%s|%s" % (b, spk, phon, s["raw_text"]))
        np.save(os.path.join(root, "mel", "%s-mel-%s.npy" % (spk, b)), s["mel"])
        np.save(os.path.join(root, "energy", "%s-energy-%s.npy" % (spk, b)), s["energy"])
        np.save(os.path.join(root, "duration", "%s-duration-%s.npy" % (spk, b)), s["duration"])
        np.save(os.path.join(root, "pitch", "%s-pitch-%s.npy" % (spk, b)), s["pitch_raw"])
        np.save(os.path.join(root, "pitch", "%s-cwt-pitch-%s.npy" % (spk, b)), s["pitch_cwt"])
        np.save(os.path.join(root, "pitch", "%s-pitch-mean-%s.npy" % (spk, b)), s["pitch_mean"])
        np.save(os.path.join(root, "pitch", "%s-pitch-std-%s.npy" % (spk, b)), s["pitch_std"])
    with open(os.path.join(root, "train.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(root, "val.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines[:6]) + "\n")
    with open(os.path.join(root, "speakers.json"), "w") as f:
        f.write('{"spk0": 0, "spk1": 1, "spk2": 2}')
    import shutil
    shutil.copy(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pretrained", "stats.json"),
                os.path.join(root, "stats.json"))


def test_train_loop_eval_checkpoint_resume(cfg, tmp_path):
    import train
    c = copy.deepcopy(cfg)
    root = str(tmp_path / "prep")
    write_corpus(root)
    c.preprocess_config.path.preprocessed_path = root
    c.train_config["optimizer"]["batch_size"] = 2
    c.train_config["optimizer"]["grad_acc_step"] = 1
    c.train_config["path"] = {"ckpt_path": str(tmp_path / "ckpt"), "log_path": str(tmp_path / "log"), "result_path": str(tmp_path / "res")}
    c.train_config["step"].update({"log_step": 2, "val_step": 4, "save_step": 4, "synth_step": 1000, "total_step": 100})
    c.model_config["transformer"]["encoder_layer"] = c.model_config["transformer"]["decoder_layer"] = 1      # keep it quick
    c.gpu = "cuda:0"
    c.mi355x["loader_workers"] = 0
    model, opt = train.main(c, max_steps=4)
    assert opt.current_step == 4
    path = os.path.join(c.train_config["path"]["ckpt_path"], "4.pth.tar")
    ck = torch.load(path)
    assert set(ck) == {"model", "embedding", "optimizer"} and "speaker_emb.weight" not in ck["model"]
    # ---- evaluate: message format and values vs the oracle on the same weights (eval mode, teacher forced)
    msg = train.evaluate(model, 4, c, None, "val", None, "cuda:0")
    assert msg.startswith("Validation Step 4,") and "Mel Loss:" in msg
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ds = D.Dataset("val.txt", c.preprocess_config, c.train_config, sort=False, drop_last=False)
    sums = np.zeros(4)
    for b in ds.collate_fn([ds[i] for i in range(len(ds))]):
        tb = tuple(torch.as_tensor(x) if isinstance(x, np.ndarray) else x for x in b)
        with torch.no_grad():
            out = ofs2.fs2_forward(sd, c.model_config, tb[2].long(), tb[3].long(), tb[4], int(tb[5]), tb[6].float(), tb[7], int(tb[8]),
                                   tb[9].float(), tb[10].long(), tb[11].float(), train=False)
            ls = ofs2.fs2_loss((None, None, tb[2], tb[3], tb[4], tb[5], tb[6].float(), tb[7], tb[8], tb[9].float(), tb[10].long(),
                                tb[11].float()), out)
        sums += np.array([float(l.sum()) for l in ls[1:5]]) * len(b[0])
    want = sums / len(ds)
    got = [float(x.split(":")[1].strip().rstrip(",")) for x in msg.split("\n")[1:4]]
    print("evaluate", got, "oracle total/mel/pitch", [want.sum(), want[0], want[1]])
    np.testing.assert_allclose(got, [want.sum(), want[0], want[1]], rtol=0.02)
    # ---- resume: weights, embedding and Adam moments come back; the next step continues from step 5
    c2 = copy.deepcopy(c)
    c2.tts["load_path"] = path
    c2.tts["restore_step"] = 4
    m2, o2 = train.get_model(c2, "cuda:0", train=True)
    assert torch.equal(m2.flat_buffers()[0], model.flat_buffers()[0])
    assert torch.equal(o2.exp_avg, opt.exp_avg) and torch.equal(o2.exp_avg_sq, opt.exp_avg_sq)
    assert o2.current_step == 4 and abs(o2.lr() - ofs2.lr_at(4)) < 1e-12

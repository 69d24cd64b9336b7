"""CPU: collate / padding / batch-tuple layout against batches produced by the reference's own
`Dataset.collate_fn` on the same seeded samples (tests/golden/collate.npz, tools/make_goldens.py:g10_collate), plus the
Dataset file layout and the device feeder on CPU."""
import os

import numpy as np
import torch

from tests.oracle_util import GOLDEN
from tts_king_amd import dataset as D


def synthetic_samples(n, seed):         # same generator as tools/make_goldens.py:synthetic_samples
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        L = int(rng.randint(5, 40))
        dur = rng.randint(1, 6, size=L)
        T = int(dur.sum())
        out.append({"id": "utt%03d" % i, "speaker": int(rng.randint(0, 65)), "text": rng.randint(1, 207, size=L),
                    "raw_text": "raw %d" % i, "mel": rng.randn(T, 80).astype(np.float32), "energy": rng.randn(L).astype(np.float32),
                    "duration": dur, "pitch_raw": rng.randn(L).astype(np.float32), "pitch_mean": np.float32(rng.randn()),
                    "pitch_std": np.float32(abs(rng.randn()) + 0.1), "pitch_cwt": rng.randn(L, 11).astype(np.float32)})
    return out


def test_collate_matches_reference():
    g = np.load(os.path.join(GOLDEN, "collate.npz"))
    for tag, (n, bs, sort, drop) in {"a": (11, 4, True, True), "b": (11, 4, True, False), "c": (8, 4, False, True)}.items():
        batches = D.collate(synthetic_samples(n, 77), bs, sort, drop)
        assert len(batches) == int(g[tag + "/n"])
        for bi, b in enumerate(batches):
            assert len(b) == 15
            assert list(b[0]) == list(g["%s/%d/ids" % (tag, bi)])
            for fi in (2, 3, 4, 6, 7, 9, 10, 11, 12, 13, 14):
                want = g["%s/%d/%d" % (tag, bi, fi)]
                assert np.asarray(b[fi]).dtype == want.dtype and np.array_equal(np.asarray(b[fi]), want), (tag, bi, fi)
            assert [b[5], b[8]] == g["%s/%d/max" % (tag, bi)].tolist()


def test_dataset_files_and_feeder(tmp_path):
    samples = synthetic_samples(6, 5)
    root = tmp_path / "prep"
    for d in ("mel", "energy", "duration", "pitch"):
        (root / d).mkdir(parents=True)
    from tts_king_amd import text as T
    syms = T.symbols()
    lines = []
    for s in samples:
        spk, b = "spk%d" % (s["speaker"] % 2), s["id"]
        phon = "{" + " ".join(syms[i][1:] for i in (150 + (s["text"] % 50))) + "}"       # '@'-prefixed phoneme entries
        lines.append("%s|%s|%s|%s" % (b, spk, phon, s["raw_text"]))
        np.save(root / "mel" / ("%s-mel-%s.npy" % (spk, b)), s["mel"])
        np.save(root / "energy" / ("%s-energy-%s.npy" % (spk, b)), s["energy"])
        np.save(root / "duration" / ("%s-duration-%s.npy" % (spk, b)), s["duration"])
        np.save(root / "pitch" / ("%s-pitch-%s.npy" % (spk, b)), s["pitch_raw"])
        np.save(root / "pitch" / ("%s-cwt-pitch-%s.npy" % (spk, b)), s["pitch_cwt"])
        np.save(root / "pitch" / ("%s-pitch-mean-%s.npy" % (spk, b)), s["pitch_mean"])
        np.save(root / "pitch" / ("%s-pitch-std-%s.npy" % (spk, b)), s["pitch_std"])
    (root / "train.txt").write_text("\n".join(lines) + "\n", encoding="utf-8")
    (root / "speakers.json").write_text('{"spk0": 0, "spk1": 1}')
    pc = {"path": {"preprocessed_path": str(root)}, "preprocessing": {"text": {"text_cleaners": []}}}
    ds = D.Dataset("train.txt", pc, {"optimizer": {"batch_size": 2}}, sort=True, drop_last=True)
    assert len(ds) == 6
    it = ds[3]
    assert it["id"] == "utt003" and np.array_equal(it["mel"], samples[3]["mel"]) and it["speaker"] in (0, 1)
    assert it["text"].tolist() == (150 + samples[3]["text"] % 50).tolist()
    batches = ds.collate_fn([ds[i] for i in range(6)])
    assert len(batches) == 3 and batches[0][3].shape[0] == 2
    assert batches[0][4][0] >= batches[0][4][1] >= batches[1][4][0]             # descending phoneme count
    dev = list(D.DeviceFeeder(batches, "cpu"))
    assert len(dev) == 3 and torch.is_tensor(dev[0][6]) and dev[0][6].dtype == torch.float32 and dev[0][10].dtype == torch.int64
    assert torch.equal(dev[1][3], torch.from_numpy(batches[1][3]).long())

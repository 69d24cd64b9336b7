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


def test_packed_bucket_staging_equals_pad_to_bucket():
    """The feeder's one-buffer staging (dataset.packed_layout / pack_into with engine.bucket_plan's target shapes: the bucket's zero
    padding written while packing) holds exactly the arrays engine.pad_to_bucket produces."""
    import numpy as np
    import torch
    from tts_king_amd.dataset import _FIELD_DTYPE, pack_into, packed_layout, views_of
    from tts_king_amd.engine import bucket_plan, pad_to_bucket
    from tts_king_amd.synthetic import make_batch
    for seed, L in ((3, 61), (4, 64), (5, 37)):
        b = tuple(x.numpy() if torch.is_tensor(x) else x for x in make_batch(5, L, seed=seed, ragged=True))
        want = pad_to_bucket(b, 8, 32, 1000)
        Lb, Tb, t_true, l_true, axis1 = bucket_plan(b, 8, 32, 1000)
        assert (Lb, Tb, t_true, l_true) == (want[5], want[8], want.t_true, want.l_true)
        arrays = []
        for i, x in enumerate(b):
            if isinstance(x, np.ndarray) and x.dtype != object:
                a = x.astype(_FIELD_DTYPE[i], copy=False) if i in _FIELD_DTYPE else x
                shp = a.shape if i not in axis1 else (a.shape[0], axis1[i]) + tuple(a.shape[2:])
                arrays.append((i, a, tuple(shp)))
        layout, total = packed_layout(arrays)
        assert total % 256 == 0 and all(o % 256 == 0 for _, o, _, _, _ in layout)
        buf = np.full(total, 0xAB, dtype=np.uint8)          # stale bytes of the slot's previous batch must not survive
        pack_into(buf, layout, arrays)
        got = views_of(torch.from_numpy(buf), layout)
        for i, _, _ in arrays:
            w = np.asarray(want[i])
            assert tuple(got[i].shape) == w.shape, (i, got[i].shape, w.shape)
            assert np.array_equal(got[i].numpy(), w.astype(got[i].numpy().dtype)), i

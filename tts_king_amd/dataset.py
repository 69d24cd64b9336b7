"""Dataset, collate and device feeder of the FS2 trainer (host-side, no kernels).

reference: fs_two/dataset.py:32-225 (`Dataset`: metadata file `name|speaker|phonemes|raw text`, per-utterance .npy files
`{kind}/{speaker}-{kind}-{basename}.npy`, `collate_fn` = sort a group of batch_size*group_size utterances by phoneme
count, cut it into batch_size batches, `reprocess` = pad and emit the 15-tuple), fs_two/utils/tools.py:330-366 (pad_1D,
pad_2D), train.py:83-99,139-141 (DataLoader with group_size 4, `to_device`).

What is MI355X-specific: `DeviceFeeder` — the reference moves each batch with blocking `.to(device)` calls from the
training loop; here batches are staged in pinned host memory and copied on a side HIP stream one batch ahead, so the
H2D transfer (2.2 MB per batch) overlaps the previous step's graph replay.
"""
import json
import os

import numpy as np
import torch

from .text import text_to_sequence


def pad_1D(inputs, PAD=0):
    """reference: fs_two/utils/tools.py:330-341."""
    n = max(len(x) for x in inputs)
    return np.stack([np.pad(x, (0, n - x.shape[0]), mode="constant", constant_values=PAD) for x in inputs])


def pad_2D(inputs, maxlen=None):
    """reference: fs_two/utils/tools.py:344-366 (rows padded with zeros; a longer input is an error)."""
    n = maxlen if maxlen else max(np.shape(x)[0] for x in inputs)
    out = []
    for x in inputs:
        if np.shape(x)[0] > n:
            raise ValueError("not max_len")
        out.append(np.pad(x, ((0, n - np.shape(x)[0]), (0, 0)), mode="constant", constant_values=0))
    return np.stack(out)


def reprocess(data, idxs):
    """Samples -> the 15-tuple batch (reference: fs_two/dataset.py:158-204)."""
    g = lambda k: [data[i][k] for i in idxs]
    texts, mels = g("text"), g("mel")
    text_lens = np.array([t.shape[0] for t in texts])
    mel_lens = np.array([m.shape[0] for m in mels])
    return (g("id"), g("raw_text"), np.array(g("speaker")), pad_1D(texts), text_lens, max(text_lens), pad_2D(mels), mel_lens,
            max(mel_lens), pad_1D(g("energy")), pad_1D(g("duration")), pad_1D(g("pitch_raw")), pad_2D(g("pitch_cwt")),
            np.array(g("pitch_mean")), np.array(g("pitch_std")))


def collate(data, batch_size, sort=True, drop_last=True):
    """A group of samples -> list of batches (reference: fs_two/dataset.py:206-225): descending phoneme count (stable
    argsort of the negated lengths), full batches first, the remainder only when drop_last is False."""
    n = len(data)
    idx = np.argsort(-np.array([d["text"].shape[0] for d in data])) if sort else np.arange(n)
    full = n - n % batch_size
    groups = idx[:full].reshape((-1, batch_size)).tolist()
    if not drop_last and n % batch_size:
        groups.append(idx[full:].tolist())
    return [reprocess(data, g) for g in groups]


class Dataset(torch.utils.data.Dataset):
    """reference: fs_two/dataset.py:32-156.  `random_mask` never triggers in the reference (max_masks_per_sentence 0.15
    is not > 1) and is not reproduced."""

    def __init__(self, filename, preprocess_config, train_config, sort=False, drop_last=True):
        self.preprocessed_path = preprocess_config["path"]["preprocessed_path"]
        self.cleaners = preprocess_config["preprocessing"]["text"]["text_cleaners"]
        self.batch_size = train_config["optimizer"]["batch_size"]
        self.sort, self.drop_last = sort, drop_last
        self.basename, self.speaker, self.text, self.raw_text = [], [], [], []
        with open(os.path.join(self.preprocessed_path, filename), encoding="utf-8") as f:
            for line in f:
                n, s, t, r = line.strip("\n").split("|")
                self.basename.append(n); self.speaker.append(s); self.text.append(t); self.raw_text.append(r)
        with open(os.path.join(self.preprocessed_path, "speakers.json")) as f:
            self.speaker_map = json.load(f)

    def __len__(self):
        return len(self.text)

    def _load(self, kind, sub, speaker, basename):
        return np.load(os.path.join(self.preprocessed_path, kind, "%s-%s-%s.npy" % (speaker, sub, basename)))

    def __getitem__(self, idx):
        b, s = self.basename[idx], self.speaker[idx]
        return {"id": b, "speaker": self.speaker_map[s], "raw_text": self.raw_text[idx],
                "text": np.array(text_to_sequence(self.text[idx], self.cleaners)),
                "mel": self._load("mel", "mel", s, b), "energy": self._load("energy", "energy", s, b),
                "duration": self._load("duration", "duration", s, b), "pitch_raw": self._load("pitch", "pitch", s, b),
                "pitch_cwt": self._load("pitch", "cwt-pitch", s, b), "pitch_mean": self._load("pitch", "pitch-mean", s, b),
                "pitch_std": self._load("pitch", "pitch-std", s, b)}

    def collate_fn(self, data):
        return collate(data, self.batch_size, self.sort, self.drop_last)


class _PinnedPool:
    """Reusable pinned staging buffers, `depth` per (field, shape, dtype): pinning fresh memory for every batch (hipHostMalloc) cost
    milliseconds per step.  A buffer is reused `depth` batches later; the H2D copy that read it is normally long done by then,
    but only an event says so to the HOST (the training stream waiting for the copy stream orders the device, not this thread):
    `copied(...)` records one on the copy stream after the slot's `.to()`, and `stage` waits for it before overwriting the slot."""

    def __init__(self, depth=6):
        # depth: a slot comes round again `depth` batches later.  Round 4 measured what 3 costs: the H2D copy of a slot shares a hardware
        # queue with branches of the replayed step graph and can sit behind them for whole steps — the host found the slot's event
        # unfinished in 10 of 33 waits and then blocked 1.3 ms per batch on average (up to 9.5 ms), 0.3 ms of every trainer step.  At 6+
        # every wait found its event complete (tools/debug/loop_ab.py).
        self.depth, self.slots, self.turn = depth, {}, {}

    def stage(self, field, arr):
        arr = np.asarray(arr)
        key = (field, arr.shape, arr.dtype.str)
        lst = self.slots.setdefault(key, [])
        i = self.turn.get(key, 0)
        if len(lst) <= i:
            t = torch.empty(arr.shape, dtype=torch.from_numpy(np.empty(0, dtype=arr.dtype)).dtype).pin_memory()
            lst.append([t, t.numpy(), None])
        self.turn[key] = (i + 1) % self.depth
        ev = lst[i][2]
        if ev is not None:
            ev.synchronize()               # the copy that last read this slot (a no-op when it finished, as it nearly always has)
        np.copyto(lst[i][1], arr)          # a plain memcpy into the pinned pages (Tensor.copy_ into pinned memory took 0.3 ms per call)
        self._last = lst[i]
        return lst[i][0]

    def to_device_packed(self, arrays, device):
        """Every array of a batch through ONE pinned buffer and ONE H2D copy: arrays = [(tag, ndarray, target shape)] — an array
        smaller than its target shape is zero-padded on the way in.  Returns (packed uint8 device
        tensor, {tag: device tensor view}, layout) — layout = ((tag, byte offset, byte length, numpy dtype str, shape), ...) with
        256-byte aligned offsets: the same shapes give the same layout, so a consumer that holds a buffer of this layout (a captured
        graph's static inputs, tts_king_amd/graph.py) takes the next batch with one device copy instead of one per field."""
        layout, total = packed_layout(arrays)
        key = ("packed", total)
        lst = self.slots.setdefault(key, [])
        i = self.turn.get(key, 0)
        if len(lst) <= i:
            t = torch.empty(total, dtype=torch.uint8).pin_memory()
            lst.append([t, t.numpy(), None])
        self.turn[key] = (i + 1) % self.depth
        slot = lst[i]
        if slot[2] is not None:
            if _DEBUG_EVENTS is not None:
                import time as _t
                q = slot[2].query()
                t0 = _t.perf_counter()
                slot[2].synchronize()
                _DEBUG_EVENTS.append((q, _t.perf_counter() - t0))
            else:
                slot[2].synchronize()
        host = slot[1]
        pack_into(host, layout, arrays)
        packed = slot[0].to(device, non_blocking=True)
        if slot[2] is None:
            slot[2] = torch.cuda.Event()
        slot[2].record()
        return packed, views_of(packed, layout), tuple(layout)

    def to_device(self, field, arr, device):
        """stage + the non-blocking H2D copy on the current (copy) stream + the slot's completion event."""
        t = self.stage(field, arr).to(device, non_blocking=True)
        slot = self._last
        if slot[2] is None:
            slot[2] = torch.cuda.Event()
        slot[2].record()
        return t


def packed_layout(arrays):
    """((tag, byte offset, byte length, numpy dtype str, target shape), ...) and the total byte count (a multiple of 256) for
    arrays = [(tag, ndarray, target shape)], 256-byte aligned offsets in the given order."""
    layout, off = [], 0
    for tag, a, shp in arrays:
        nb = int(np.prod(shp)) * a.dtype.itemsize
        layout.append((tag, off, nb, a.dtype.str, tuple(shp)))
        off = (off + nb + 255) // 256 * 256
    return layout, max(off, 256)


def pack_into(host_u8, layout, arrays):
    """Write the arrays into a uint8 numpy buffer at their layout positions; an array smaller than its target shape is zero-padded on the
    way in (the shape bucket's padding: no np.pad copy of the array first)."""
    for (tag, o, nb, dt, shp), (_, a, _s) in zip(layout, arrays):
        if not nb:
            continue
        dst = host_u8[o:o + nb].view(dt).reshape(shp)
        if a.shape == shp:
            np.copyto(dst, a)
        else:
            dst[...] = 0
            dst[tuple(slice(0, n) for n in a.shape)] = a


def views_of(packed, layout):
    """{tag: typed view} into a packed batch buffer (see _PinnedPool.to_device_packed)."""
    out = {}
    for tag, o, nb, dt, shp in layout:
        tdt = torch.from_numpy(np.empty(0, dtype=np.dtype(dt))).dtype
        out[tag] = packed[o:o + nb].view(tdt).view(shp)
    return out


_POOLS = {}
_DEBUG_EVENTS = None

# dtypes `to_device` gives the batch fields (reference: fs_two/utils/tools.py:15-83): speakers / texts / durations long, mels / pitches
# float, pitches_cwt float with NaN -> 0, the others as stored
_FIELD_DTYPE = {2: np.int64, 3: np.int64, 6: np.float32, 10: np.int64, 11: np.float32, 12: np.float32, 13: np.float32, 14: np.float32}


class DeviceFeeder:
    """Iterates device batches (the tuple `to_device` returns) over an iterable of numpy 15-tuples, copying one batch
    ahead on a side stream from pinned staging buffers.  `bucket` = (l_bucket, t_bucket, max_seq_len): batches are padded to
    shape buckets on the host first (tts_king_amd.engine.pad_to_bucket) and carry `frame_limit` / `phoneme_limit`."""

    def __init__(self, batches, device, bucket=None):
        self.it, self.device = iter(batches), torch.device(device)
        self.bucket = bucket
        # (a high-priority copy stream was tried so that the copies would not queue behind the step graph's branches: every slot event
        # was then complete on time, but the replayed step itself took 6.2 ms instead of 2.7 — plain priority)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.pool = _POOLS.setdefault(str(self.device), _PinnedPool()) if self.stream is not None else None
        self._next = None
        self._preload()

    def _preload(self):
        from .engine import PaddedBatch, bucket_plan, pad_to_bucket
        from .train_step import to_device
        try:
            b = next(self.it)
        except StopIteration:
            self._next = None
            return
        t_true = l_true = None
        nb = len(b[0])
        if self.stream is None:
            if self.bucket is not None:
                b = pad_to_bucket(b, *self.bucket)
                t_true, l_true = b.t_true, b.l_true
            dev_b = to_device(b, self.device)
            fl = torch.tensor([t_true], dtype=torch.int32, device=self.device) if t_true is not None else None
            pl = torch.full((nb,), l_true, dtype=torch.int64, device=self.device) if l_true is not None else None
        else:
            # the whole batch (bucket padding included) and its frame / phoneme limits through one pinned buffer and one H2D copy
            axis1, b = {}, list(b)
            if self.bucket is not None:
                Lb, Tb, t_true, l_true, axis1 = bucket_plan(b, *self.bucket)
                b[5], b[8] = Lb, Tb
            arrays = []
            for i, x in enumerate(b):
                if isinstance(x, np.ndarray) and x.dtype != object:
                    a = x.astype(_FIELD_DTYPE[i], copy=False) if i in _FIELD_DTYPE else x
                    if i == 12:
                        a = np.nan_to_num(a, nan=0.0)
                    shp = a.shape if i not in axis1 else (a.shape[0], axis1[i]) + tuple(a.shape[2:])
                    arrays.append((i, a, tuple(shp)))
            if t_true is not None:
                arrays.append(("fl", np.array([t_true], dtype=np.int32), (1,)))
            if l_true is not None:
                arrays.append(("pl", np.full((nb,), l_true, dtype=np.int64), (nb,)))
            with torch.cuda.stream(self.stream):
                packed, views, layout = self.pool.to_device_packed(arrays, self.device)
            dev_b = tuple(views[i] if i in views else x for i, x in enumerate(b))
            fl, pl = views.get("fl"), views.get("pl")
        if self.bucket is not None or self.stream is not None:
            dev_b = PaddedBatch(dev_b)
            dev_b.t_true, dev_b.l_true, dev_b.frame_limit, dev_b.phoneme_limit = t_true, l_true, fl, pl
            if self.stream is not None:
                dev_b.packed, dev_b.layout = packed, layout
        self._next = dev_b

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        batch = self._next
        pk = getattr(batch, "packed", None)
        if pk is not None:
            pk.record_stream(torch.cuda.current_stream())      # (every field is a view of it)
        else:
            for t in tuple(batch) + (getattr(batch, "frame_limit", None), getattr(batch, "phoneme_limit", None)):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(torch.cuda.current_stream())
        self._preload()
        return batch

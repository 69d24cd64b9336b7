"""Text -> phoneme-id frontend (host-side string processing, no kernels).

reference: fs_two/text/__init__.py (`text_to_sequence`, `sequence_to_text`), fs_two/text/symbols.py (inventory).
The symbol inventory is a data asset, `pretrained/symbols.json` (generated from the reference by
tools/make_goldens.py, 206 entries; a symbol's id is its position, the model's vocabulary is 207 with the PAD row,
Models.py:40).  Phonemes are written in braces, `{R A B O0 T ...}`, and map to the `@`-prefixed entries; text outside
braces maps character by character; unknown symbols, `_` and `~` are dropped — the behaviour the reference's notebook
vector pins (examples.ipynb cell 2, tests/golden/text_to_sequence.json).
Cleaners (unidecode / number expansion) are not part of this build: `cleaner_names` must be empty, as it is on the
reference's Russian path (input_process.py: CLEANERS = []).
"""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SYMBOLS_PATH = os.path.join(os.path.dirname(_HERE), "pretrained", "symbols.json")

_symbols = None
_sym2id = None


def symbols(path=None):
    global _symbols, _sym2id
    if _symbols is None or path is not None:
        with open(path or SYMBOLS_PATH, encoding="utf-8") as f:
            _symbols = json.load(f)
        _sym2id = {s: i for i, s in enumerate(_symbols)}
    return _symbols


def _keep(s):
    return s in _sym2id and s != "_" and s != "~"


def _ids(seq):
    return [_sym2id[s] for s in seq if _keep(s)]


def text_to_sequence(text, cleaner_names=()):
    """String -> list of symbol ids; brace-enclosed runs are phoneme names separated by spaces."""
    if cleaner_names:
        raise NotImplementedError("text cleaners are outside this build's scope; pass cleaner_names=[] (the reference's "
                                  "Russian path uses none)")
    symbols()
    out = []
    pos = 0
    while pos < len(text):
        lb = text.find("{", pos)
        rb = text.find("}", lb + 2) if lb >= 0 else -1          # the reference's pattern needs >= 1 char inside the braces
        if lb < 0 or rb < 0:
            out += _ids(text[pos:])
            break
        out += _ids(text[pos:lb])
        out += _ids(["@" + p for p in text[lb + 1:rb].split()])
        pos = rb + 1
    return out


def sequence_to_text(sequence):
    """Ids -> string, phonemes back in braces (reference: fs_two/text/__init__.py:43-53)."""
    syms = symbols()
    res = ""
    for i in sequence:
        if 0 <= i < len(syms):
            s = syms[i]
            if len(s) > 1 and s[0] == "@":
                s = "{%s}" % s[1:]
            res += s
    return res.replace("}{", " ")

"""Text preprocessing entry points with the reference's names (reference: input_process.py:14-86).

`preprocess_lang` (lexicon lookup, `pretrained/rus_all.dict`-style file) works out of the box; `preprocess_rus` needs the
optional `russian_g2p` package exactly as in the reference (it is not installable on the MI355X image);
`preprocess_eng` needs `g2p_en`.  All three end in `tts_king_amd.text.text_to_sequence(phones, [])`.
"""
import re
from string import punctuation

import numpy as np

from tts_king_amd.text import text_to_sequence

CLEANERS = []      # no cleaners for the Russian dataset (reference: input_process.py:10)


def read_lexicon(lex_path):
    """word -> phoneme list; the first entry of a word wins (reference: input_process.py:14-24)."""
    lexicon = {}
    with open(lex_path, encoding="utf-8") as f:
        for line in f:
            parts = re.split(r"\s+", line.strip("\n"))
            word = parts[0].lower()
            if word not in lexicon:
                lexicon[word] = parts[1:]
    return lexicon


def _phones_to_sequence(phones):
    s = "{" + "}{".join(phones) + "}"
    s = re.sub(r"\{[^\w\s]?\}", "{sp}", s)          # punctuation / empty slots become short pauses
    s = s.replace("}{", " ")
    return np.array(text_to_sequence(s, CLEANERS))


def _split_words(text):
    return re.split(r"([,;.\-\?\!\s+])", text.rstrip(punctuation))


def preprocess_lang(text, preprocess_config):
    """Lexicon-only G2P: unknown words become '.', i.e. a pause (reference: input_process.py:49-68)."""
    lexicon = read_lexicon(preprocess_config["path"]["lexicon_path"])
    phones = []
    for w in _split_words(text):
        phones += lexicon[w.lower()] if w.lower() in lexicon else ["."]
    return _phones_to_sequence(phones)


def preprocess_eng(text, preprocess_config):
    """reference: input_process.py:27-46 (lexicon first, g2p_en for the rest)."""
    try:
        from g2p_en import G2p
    except ImportError as e:
        raise ImportError("preprocess_eng needs the optional g2p_en package") from e
    lexicon = read_lexicon(preprocess_config["path"]["lexicon_path"])
    g2p = G2p()
    phones = []
    for w in _split_words(text):
        phones += lexicon[w.lower()] if w.lower() in lexicon else [p for p in g2p(w) if p != " "]
    return _phones_to_sequence(phones)


_transcriptor = None


def preprocess_rus(text):
    """reference: input_process.py:71-86 (russian_g2p transcription, 'sp' after every sentence)."""
    global _transcriptor
    try:
        from russian_g2p.Transcription import Transcription
    except ImportError as e:
        raise ImportError("preprocess_rus needs the optional russian_g2p package; pass phoneme ids or a '{...}' phoneme "
                          "string instead") from e
    if _transcriptor is None:
        _transcriptor = Transcription()
    sentences = _transcriptor.transcribe([text.rstrip(punctuation)])[0]
    return _phones_to_sequence([p for s in sentences for p in s + ["sp"]])

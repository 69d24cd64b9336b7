"""CPU: text frontend against known-answer vectors produced by the reference's text_to_sequence
(tests/golden/text_to_sequence.json, incl. the vector printed in the reference's examples.ipynb cell 2)."""
import json
import os

import numpy as np

from tests.oracle_util import GOLDEN
from tts_king_amd import text as T


def test_known_answer_vectors():
    g = json.load(open(os.path.join(GOLDEN, "text_to_sequence.json"), encoding="utf-8"))
    assert len(T.symbols()) == g["n_symbols"] == 206                     # vocabulary = 207 with PAD (Models.py:40)
    for s, want in g["cases"].items():
        assert T.text_to_sequence(s, []) == want, s
    nb = "{R A B O0 T A T0 I R A B O0 T A T0 sp S K A Z A0 L O0 N sp}"   # examples.ipynb cell 2
    assert T.text_to_sequence(nb, [])[:5] == [184, 151, 153, 181, 190]
    assert T.sequence_to_text(T.text_to_sequence(nb, [])) == nb


def test_lexicon_path(tmp_path):
    import input_process
    lex = tmp_path / "lex.dict"
    lex.write_text("привет P R I0 V E0 T\nмир M I0 R\nпривет X X X\n", encoding="utf-8")
    seq = input_process.preprocess_lang("Привет, мир!", {"path": {"lexicon_path": str(lex)}})
    want = T.text_to_sequence("{P R I0 V E0 T sp sp sp M I0 R}", [])
    assert isinstance(seq, np.ndarray) and seq.tolist() == want

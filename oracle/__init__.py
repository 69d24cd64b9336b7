"""ORACLE — test infrastructure only.

CPU restatements of the reference algorithms on the hot path, used as the checker by `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg.  Nothing under `tts_king_amd/` may import
this package.  Pinned against outputs of the reference itself (tests/golden/, tools/make_goldens.py).
"""

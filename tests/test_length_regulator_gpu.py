"""GPU: LengthRegulator scan/expand kernel — bit-exact against the reference's outputs (golden) and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import fs2 as ofs2
from tests.oracle_util import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(x, d, T):
    from tts_king_amd import ops
    xb = x.to(torch.bfloat16)
    out, idx, cs, ml = ops.length_regulator_fwd(xb.to(DEV), d.to(DEV), T)
    torch.cuda.synchronize()
    return xb, out.cpu(), idx.cpu(), cs.cpu(), ml.cpu()


def test_golden_edge_cases():
    g = np.load(os.path.join(GOLDEN, "length_regulator.npz"))
    x = torch.zeros(3, 7, 8)
    x[:, :, :4] = torch.from_numpy(g["x"])          # kernel rows are multiples of 8 channels
    d = torch.from_numpy(g["d"])
    for tag, T in (("none", 7), ("crop4", 4), ("pad12", 12)):
        xb, out, idx, cs, ml = run(x, d, T)
        ref, ref_len = ofs2.length_regulator(xb.float(), d, T)
        assert torch.equal(out.float(), ref)                                 # oracle, bit-exact
        assert ml.tolist() == g["len_" + tag].tolist() == ref_len.tolist()   # uncropped totals
        gold = torch.from_numpy(g["out_" + tag]).to(torch.bfloat16).float()  # reference output, bf16-rounded
        assert torch.equal(out.float()[:, :, :4], gold)
        ridx, _ = ofs2.length_regulator_index(d, T)
        assert torch.equal(idx.long(), ridx)


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32, torch.float32])
@pytest.mark.parametrize("B,L,D,hi", [(16, 64, 256, 12), (4, 217, 256, 9), (2, 1, 8, 5), (3, 130, 64, 2)])
def test_random_vs_oracle_and_bwd(B, L, D, hi, dtype):
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, L, D, generator=g)
    d = torch.randint(0, hi, (B, L), generator=g)
    if dtype == torch.float32:
        d = d.float() + torch.rand(B, L, generator=g) * 0.9 - 0.3            # fractional and slightly negative
    d = d.to(dtype)
    _, ref_len = ofs2.length_regulator_index(d)
    T = max(int(ref_len.max()), 1)
    xb, out, idx, cs, ml = run(x, d, T)
    ref, _ = ofs2.length_regulator(xb.float(), d, T)
    assert torch.equal(out.float(), ref)
    assert torch.equal(ml, ref_len)
    # position-encoding fusion: out + PE[t] rounded once to bf16
    pe = ofs2.sinusoid_table(T + 1, D)
    out2, _, _, _ = ops.length_regulator_fwd(xb.to(DEV), d.to(DEV), T, pe=pe.to(DEV))
    assert torch.equal(out2.cpu().float(), (ref + pe[:T][None]).to(torch.bfloat16).float())
    # backward = segment sum
    dout = torch.randn(B, T, D, generator=g).to(torch.bfloat16)
    dx = ops.length_regulator_bwd(dout.to(DEV), cs.to(DEV), L).cpu().float()
    ridx, _ = ofs2.length_regulator_index(d, T)
    want = torch.zeros(B, L, D, dtype=torch.float64)
    for b in range(B):
        m = ridx[b] >= 0
        want[b].index_add_(0, ridx[b][m], dout[b][m].double())
    assert float((dx.double() - want).abs().max()) <= 2 ** -7 * float(want.abs().max() + 1)

"""FFT-block w_1 forward (Conv1d 256 -> 1024, k = 9, ReLU): window kernel vs the implicit-GEMM conv, decoder and encoder shapes."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
for B, S in ((16, 423), (16, 64), (16, 448)):
    x = torch.randn(B, S, 256, device=DEV).bfloat16()
    W = (torch.randn(1024, 9, 256, device=DEV) * 0.02).bfloat16()
    b = torch.randn(1024, device=DEV)
    t1 = timeit(lambda: ops.ffn_conv_fwd(x, W, b))
    pk = ops.ffn_pack_weight(W)
    assert torch.equal(ops.ffn_conv_fwd(x, W, b, packed=pk), ops.ffn_conv_fwd(x, W, b))
    t3 = timeit(lambda: ops.ffn_conv_fwd(x, W, b, packed=pk))
    t4 = timeit(lambda: ops.ffn_pack_weight(W, out=pk))
    print("B=%d S=%d: packed weights %.1f us (pack itself %.1f us)" % (B, S, t3, t4))
    t2 = timeit(lambda: ops.conv1d(x, W, b, flags=ops.RELU))
    gf = 2.0 * B * S * 1024 * 256 * 9 / 1e9
    print("B=%d S=%d: window kernel %.1f us (%.0f TF/s) | implicit GEMM %.1f us (%.0f TF/s)" % (B, S, t1, gf / t1 * 1e3, t2, gf / t2 * 1e3))

# PostNet Conv1d(512 -> 512, k = 5): forward (fp32 out for BatchNorm) and input gradient, window kernel vs implicit GEMM
B, S = 16, 423
x = torch.randn(B, S, 512, device=DEV).bfloat16()
W = (torch.randn(512, 5, 512, device=DEV) * 0.02).bfloat16()
b = torch.randn(512, device=DEV)
pk, pkt = torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV), torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV)
ops.win_conv_pack_batch([W], [pk]); ops.win_conv_pack_batch([W], [pkt], transpose=True)
f1 = ops.win_conv(x, pk, 512, 5, bias=b, out_dtype=torch.float32); f2 = ops.conv1d(x, W, b, out_dtype=torch.float32)
g1 = ops.win_conv(x, pkt, 512, 5); g2 = ops.conv1d_dx(x, W)
print("PostNet fwd max diff %.3g of %.3g; dX max diff %.3g of %.3g" % (float((f1 - f2).abs().max()), float(f2.abs().max()), float((g1.float() - g2.float()).abs().max()), float(g2.float().abs().max())))
t1 = timeit(lambda: ops.win_conv(x, pk, 512, 5, bias=b, out_dtype=torch.float32)); t2 = timeit(lambda: ops.conv1d(x, W, b, out_dtype=torch.float32))
t3 = timeit(lambda: ops.win_conv(x, pkt, 512, 5)); t4 = timeit(lambda: ops.conv1d_dx(x, W))
t5 = timeit(lambda: ops.win_conv_pack_batch([W, W, W], [pk, pk, pk])); t6 = timeit(lambda: ops.win_conv_pack_batch([W, W, W], [pkt, pkt, pkt], transpose=True))
print("B=16 S=423 PostNet 512->512 k5: fwd window %.1f us | GEMM %.1f us || dX window %.1f us | GEMM %.1f us || packs of 3 weights: %.1f us, transposed %.1f us" % (t1, t2, t3, t4, t5, t6))

# decoder-side k = 1 projections on the window kernel: QKV (256 -> 768) and fc dX (256 -> 256, transposed pack)
x = torch.randn(16, 423, 256, device=DEV).bfloat16()
Wq = (torch.randn(768, 1, 256, device=DEV) * 0.05).bfloat16(); bq = torch.randn(768, device=DEV)
pq = torch.empty(Wq.numel(), dtype=torch.bfloat16, device=DEV); ops.win_conv_pack_batch([Wq], [pq])
a = ops.win_conv(x, pq, 768, 1, bias=bq); b2 = ops.linear(x.view(-1, 256), Wq.view(768, 256), bq)
print("QKV max diff %.3g" % float((a.view(-1, 768).float() - b2.float()).abs().max()))
t1 = timeit(lambda: ops.win_conv(x, pq, 768, 1, bias=bq)); t2 = timeit(lambda: ops.linear(x.view(-1, 256), Wq.view(768, 256), bq))
Wf = (torch.randn(256, 1, 256, device=DEV) * 0.05).bfloat16()
pf = torch.empty(Wf.numel(), dtype=torch.bfloat16, device=DEV); ops.win_conv_pack_batch([Wf], [pf], transpose=True)
t3 = timeit(lambda: ops.win_conv(x, pf, 256, 1)); t4 = timeit(lambda: ops.linear_dx(x.view(-1, 256), Wf.view(256, 256)))
print("B=16 S=423 QKV 256->768: window %.1f us | GEMM %.1f us || fc dX 256->256: window %.1f us | GEMM %.1f us" % (t1, t2, t3, t4))

# pack kernel: the model's full item list (one launch from a device table)
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
cfg = default_config()
m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=DEV, seed=1).train()
m.sync_shadow(force=True)
tb, n = m._pack_table
nbytes = sum(o.numel() * 2 for _, _, o, _ in m._pack_items)
t = timeit(lambda: ops.win_conv_pack_run(tb, n))
print("pack table: %d items, %.1f MB of packs, %.1f us per launch (%.2f TB/s read + write)" % (n, nbytes / 1e6, t, 2 * nbytes / t / 1e6))

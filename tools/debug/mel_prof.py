"""A few mel-extraction batches for rocprofv3 (diagnostic): B=16 x 423 frames."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd.audio import TacotronSTFT
stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000, device="cuda:0")
y = (torch.rand(16, 422 * 256) - 0.5).to("cuda:0")
for _ in range(10):
    stft._ex(y)
torch.cuda.synchronize()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import ops
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
M, K, N = 217008, 896, 128
A = torch.randn(M, K, device=dev)
pw = ops.PackedWeight().get(torch.randn(K, N, device=dev), 'kn')
ws = ops.workspace(dev)
for _ in range(3):
    ops.gemm(A, pw)
torch.cuda.synchronize()
t = ws[:28 * 5 * 8].view(torch.int64).view(28, 5).cpu()
d = torch.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]], 1).float()
tot = (t[1:, 0] - t[:-1, 0]).float()
print('per-iteration (memtime ticks): issue-loads %.0f  mfma-block %.0f  store-phase(wait+cvt+ds_write) %.0f  barrier %.0f  | total %.0f'
      % (d[2:26, 0].mean(), d[2:26, 1].mean(), d[2:26, 2].mean(), d[2:26, 3].mean(), tot[2:26].mean()))
print(d[2:10])

"""developer tool (GPU box): what the device takes to WRITE the forward's 198 MB (three 66-MB streams) with plain fills and
copies -- the floor of k_skin_fwd16*'s store streams"""
import torch, numpy as np
dev = 'cuda:0'
n = 800 * 6890 * 3
xs = [torch.empty(n, device=dev) for _ in range(3)]
big = torch.empty(3 * n, device=dev)
src = torch.randn(3 * n, device=dev)
def t(fn, k=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))
f1 = t(lambda: big.fill_(1.0))
f3 = t(lambda: [x.fill_(1.0) for x in xs])
c1 = t(lambda: big.copy_(src))
print('fill 198 MB in one launch: %.1f us (%.2f TB/s); three fills of 66 MB: %.1f us; copy 198 MB (read + write): %.1f us (%.2f TB/s written)'
      % (f1, 198.4e6 / f1 / 1e6, f3, c1, 198.4e6 / c1 / 1e6))

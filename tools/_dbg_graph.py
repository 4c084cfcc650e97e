import sys, torch
sys.path.insert(0, '.')
import brevitas_amd.quant as Q
DEV='cuda:0'
torch.manual_seed(123456)
dtype=torch.float32
w = torch.nn.Parameter((torch.randn(64, 32, 3, 3, device=DEV) * 0.1).to(dtype))
g = torch.randn(64, 32, 3, 3, device=DEV).to(dtype)
q = Q.Int8WeightPerChannelFloat(w).to(DEV)
def step():
    w.grad = None
    y, scale, _, _ = q(w)
    y.backward(g)
    return y, scale, w.grad
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = step()
for trial in range(4):
    with torch.no_grad(): w.mul_(1.5).add_(0.01)
    graph.replay(); torch.cuda.synchronize()
    got = [t.clone() for t in out]
    ref = step()
    for name, a, b in zip(('y','scale','dw'), got, ref):
        nd = int((a != b).sum())
        print(trial, name, 'diff', nd, 'maxabs', float((a-b).abs().max()) if nd else 0.0, flush=True)

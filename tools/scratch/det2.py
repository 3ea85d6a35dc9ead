import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.model_utils import build_model, fill
model = build_model("sigma_tiny", 9, 96, 128).cuda().eval()
rgb, x, label = fill.make_inputs(2, 96, 128, 9, seed=5)
runs = []
cur = None
def hook(name):
    def f(m, i, o):
        if torch.is_tensor(o): cur.append((name, o.detach().clone()))
    return f
for n, m in model.named_modules():
    if n: m.register_forward_hook(hook(n))
with torch.no_grad():
    for _ in range(3):
        cur = []
        model(rgb.cuda(), x.cuda())
        runs.append(cur)
for (n0, a), (n1, b), (n2, c) in zip(*runs):
    d01 = float((a - b).abs().max()); d12 = float((b - c).abs().max())
    if d01 > 0 or d12 > 0:
        print("first differing module:", n0, d01, d12, type(dict(model.named_modules())[n0]).__name__, tuple(a.shape))
        break

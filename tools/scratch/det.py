import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigma_amd import selective_scan_cuda_core as core
from sigma_amd.ss2d_fused import ss2d_core
import importlib
vm = importlib.import_module("sigma_amd.models.encoders.vmamba")
torch.manual_seed(0)
dev = "cuda"
for (B, d, H, W, N) in [(2, 192, 24, 32, 16), (2, 384, 12, 16, 16), (2, 768, 6, 8, 16), (2, 192, 24, 32, 4)]:
    blk = vm.SS2D(d_model=d // 2, d_state=N).to(dev)
    x = torch.randn(B, d, H, W, device=dev)
    ps = [blk.x_proj_weight, blk.dt_projs_weight, blk.dt_projs_bias, blk.A_logs, blk.Ds]
    with torch.no_grad():
        ys = [ss2d_core(x, *ps) for _ in range(4)]
    print("core", (B, d, H, W, N), [float((ys[0] - y).abs().max()) for y in ys[1:]])
    # operator alone
    L = H * W
    u = torch.randn(B, 2 * d, L, device=dev); delta = torch.randn(B, 4 * d, L, device=dev) * 0.5
    A = -torch.rand(4 * d, N, device=dev); Bm = torch.randn(B, 4, N, L, device=dev); Cm = torch.randn(B, 4, N, L, device=dev)
    D = torch.ones(4 * d, device=dev); bias = torch.zeros(4 * d, device=dev)
    for need_x in (True, False):
        outs = [core.fwd_ext(u, delta, A, Bm, Cm, D, bias, True, rev_mask=0b1010, u_gshift=1, need_x=need_x)[0] for _ in range(4)]
        print("  op need_x", need_x, [float((outs[0] - o).abs().max()) for o in outs[1:]])
    # matmul alone
    Wst = torch.randn(2, 2 * 38, d, device=dev); xs2 = torch.randn(B, 2, d, L, device=dev)
    ms = [torch.matmul(Wst.unsqueeze(0), xs2) for _ in range(4)]
    print("  matmul", [float((ms[0] - m).abs().max()) for m in ms[1:]])

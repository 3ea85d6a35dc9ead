"""Body of test_graphed_data_parallel_step_matches_eager_data_parallel_step, run in its OWN process
(``python -m tests.graph_ddp_worker {fp32|bf16}``): HIP-graph capture next to a live RCCL communicator is the one
place where a mistake ends in abort() instead of an exception, and an abort inside the pytest process would take the
rest of the GPU suite with it.  Exit code 0 = the graphed data-parallel step equals the eager one."""
import copy
import os
import socket
import sys

import torch
import torch.distributed as dist


def stage(msg):
    print(f"[graph_ddp_worker] {msg}", flush=True)


def main(bf16: bool) -> None:
    from sigma_amd import train_step as ts
    from tests.model_utils import build_model, fill
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    base = build_model("sigma_tiny", 9, 64, 96).to(dev).eval()
    rgb, x, label = fill.make_inputs(2, 64, 96, 9, seed=10)
    batch = (rgb.to(dev), x.to(dev), label.to(dev))
    ref = copy.deepcopy(base)
    opt_r = ts.make_optimizer(ref, capturable=True)

    def eager():
        opt_r.zero_grad(set_to_none=True)
        loss = ref(*batch)
        red = loss.detach().clone()
        dist.all_reduce(red)
        loss.backward()
        for p in ref.parameters():
            dist.all_reduce(p.grad)                   # world size 1: sum == mean
        opt_r.step()
        return loss.detach()

    g_model = copy.deepcopy(base)
    opt_g = ts.make_optimizer(g_model, capturable=True)
    stage("capture")
    step_g, _ = ts.make_graphed_ddp_step(g_model, opt_g, batch, warmup=3, bf16_comm=bf16)
    torch.cuda.synchronize()
    stage("eager warm-up")
    for _ in range(3):
        eager()                                       # the capture warmed up with 3 steps
    torch.cuda.synchronize()
    for i in range(2):
        stage(f"compare step {i}")
        le = eager()
        torch.cuda.synchronize()
        lg = step_g()
        torch.cuda.synchronize()
        torch.testing.assert_close(lg.detach(), le, rtol=1e-4 if not bf16 else 1e-2, atol=1e-5 if not bf16 else 1e-3)
    # 5 AdamW steps x up to 2 lr each (sign flips of near-zero gradients summed with atomics)
    for (n, a), (_, b) in zip(ref.named_parameters(), g_model.named_parameters()):
        torch.testing.assert_close(b, a, rtol=1e-4, atol=6e-4 if not bf16 else 2e-3, msg=lambda m, n=n: f"{n} (bf16={bf16}): {m}")
    stage("parameters equal")
    torch.cuda.synchronize()
    del step_g
    dist.destroy_process_group()
    stage("done")


if __name__ == "__main__":
    main(sys.argv[1:] == ["bf16"])

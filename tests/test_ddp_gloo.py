"""N > 1 path on CPU: two processes, gloo, the same step / timing / grouping code bench.py runs
with RCCL (sigma_amd/train_step.py).  The model here is a small CPU stand-in with the structural
features that matter for the data-parallel path (LayerNorm only, raw nn.Parameters that the
reference's optimizer grouping skips, a loss returned by forward); the HIP model itself cannot run
without a GPU by design."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from sigma_amd import train_step as ts


class TinySeg(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Conv2d(3, 8, 4, 4)
        self.norm = nn.LayerNorm(8)
        self.scale = nn.Parameter(torch.ones(8))          # raw Parameter: in no optimizer group (App. C-4)
        self.head = nn.Linear(16, 5)
        self.criterion = nn.CrossEntropyLoss(ignore_index=255)

    def forward(self, rgb, x, label=None):
        f = torch.cat([self.norm(self.stem(rgb).permute(0, 2, 3, 1)) * self.scale,
                       self.norm(self.stem(x).permute(0, 2, 3, 1))], dim=-1)
        logits = self.head(f).permute(0, 3, 1, 2)
        logits = nn.functional.interpolate(logits, scale_factor=4, mode="bilinear", align_corners=False)
        return logits if label is None else self.criterion(logits, label)


def _batch(rank, n=2):
    g = torch.Generator().manual_seed(100 + rank)
    return (torch.randn(n, 3, 16, 16, generator=g), torch.randn(n, 3, 16, 16, generator=g),
            torch.randint(0, 5, (n, 16, 16), generator=g))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                               # identical replicas, as DDP would broadcast
        model = TinySeg()
        dev = torch.device("cpu")
        net = ts.wrap_ddp(model, dev)
        opt = ts.make_optimizer(model, lr=1e-2)
        step = ts.make_step(net, opt, _batch(rank))
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        elapsed, loss = ts.timed_steps(step, steps=2, warmup=1, device=dev)
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        moved = {n: bool((p.detach() != before[n]).any()) for n, p in model.named_parameters()}
        # every rank must report the same (max) time
        t = torch.tensor([elapsed], dtype=torch.float64)
        ts_all = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts_all, t)
        if rank == 0:
            torch.save(dict(grads=grads, moved=moved, times=[float(v) for v in ts_all], loss=float(loss),
                            params={n: p.detach().clone() for n, p in model.named_parameters()}), out)
        else:
            torch.save(dict(grads=grads, params={n: p.detach().clone() for n, p in model.named_parameters()}),
                       out + ".r1")
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
def test_two_rank_gloo_step_matches_single_process_average(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out), torch.load(out + ".r1")
    # replicas stay identical: same gradients (all-reduced mean) and same parameters on both ranks
    for n in r0["grads"]:
        torch.testing.assert_close(r0["grads"][n], r1["grads"][n], rtol=0, atol=0)
        torch.testing.assert_close(r0["params"][n], r1["params"][n], rtol=0, atol=0)
    assert len(set(r0["times"])) == 1 and r0["times"][0] > 0          # max over ranks, agreed by all
    # reference quirk reproduced: the raw Parameter gets a gradient (and is all-reduced) but is never stepped
    assert r0["moved"]["stem.weight"] and r0["moved"]["head.bias"] and not r0["moved"]["scale"]
    # single-process reference of the LAST step's gradient: mean of the two ranks' batch gradients
    torch.manual_seed(0)
    ref = TinySeg()
    ref.load_state_dict({k: v for k, v in r0["params"].items()})       # params AFTER the last step ...
    # ... so rebuild the params BEFORE it by replaying three steps single-process on the averaged loss
    torch.manual_seed(0)
    ref = TinySeg()
    opt = ts.make_optimizer(ref, lr=1e-2)
    b0, b1 = _batch(0), _batch(1)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        loss = 0.5 * (ref(*b0) + ref(*b1))
        loss.backward()
        last = {n: p.grad.detach().clone() for n, p in ref.named_parameters()}
        opt.step()
    for n in last:
        torch.testing.assert_close(r0["grads"][n], last[n], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(r0["params"][n], dict(ref.named_parameters())[n].detach(), rtol=1e-4, atol=1e-6)


def test_weak_scaling_accounting():
    assert ts.throughput(8, 1, 5, 2.0) == 20.0
    assert ts.throughput(8, 8, 5, 2.0) == 160.0


def test_bench_launch_plan_arithmetic():
    """`python bench.py --gpus N` must become N ranks or refuse -- never silently measure one GPU."""
    argv = ["--gpus", "4", "--steps", "3"]
    kind, cmd = ts.launch_plan(4, {}, 8, argv, "/x/bench.py", 29999)
    assert kind == "spawn"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == ["/x/bench.py", *argv]
    # inside a launcher: this process is a rank, and the rank count must equal --gpus
    assert ts.launch_plan(4, dict(WORLD_SIZE="4", RANK="2", LOCAL_RANK="2"), 8, argv, "b", 1) == ("run", 4, 2, 2)
    assert ts.launch_plan(None, dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), 2, [], "b", 1) == ("run", 2, 1, 1)
    assert ts.launch_plan(None, {}, 1, [], "b", 1) == ("run", 1, 0, 0)
    for bad in (lambda: ts.launch_plan(8, dict(WORLD_SIZE="1", RANK="0"), 8, argv, "b", 1),      # --gpus ignored
                lambda: ts.launch_plan(8, {}, 1, argv, "b", 1),                                  # too few devices
                lambda: ts.launch_plan(2, dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), 1, argv, "b", 1),
                lambda: ts.launch_plan(1, {}, 0, argv, "b", 1),
                lambda: ts.launch_plan(0, {}, 1, argv, "b", 1)):
        with pytest.raises(SystemExit):
            bad()


def test_optimizer_groups_reference_quirk_and_opt_in_fix():
    """group_weight reproduces the reference's dead isinstance(m, nn.Parameter) branch (raw Parameters are never
    stepped); include_raw_params=True puts them into the no-decay group."""
    import torch.nn as nn
    from sigma_amd import train_step as ts

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)
            self.norm = nn.LayerNorm(4)
            self.A_logs = nn.Parameter(torch.zeros(4, 2))
            self.Ds = nn.Parameter(torch.ones(4))

    m = Blk()
    g = ts.group_weight(m, 1e-3)
    assert ts.unoptimized_parameters(m, g) == 2
    g2 = ts.group_weight(m, 1e-3, include_raw_params=True)
    assert ts.unoptimized_parameters(m, g2) == 0 and len(g2[1]["params"]) == len(g[1]["params"]) + 2
    assert g2[1]["weight_decay"] == 0.0


def _flat_worker(rank, world, port, out, bf16):
    """the arithmetic of the GRAPHED data-parallel step (train_step.flatten_grads + _allreduce_mean around a plain
    forward / backward) next to the eager DistributedDataParallel step, on the same replicas and batches"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = TinySeg()
        flat = ts.flatten_grads(model)
        views_ok = all(p.grad.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() < flat.data_ptr() + 4 * flat.numel()
                       for p in model.parameters())
        flat.zero_()
        model(*_batch(rank)).backward()
        ts._allreduce_mean(flat, bf16)
        mine = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        torch.manual_seed(0)
        ref = TinySeg()
        net = ts.wrap_ddp(ref, torch.device("cpu"))
        net(*_batch(rank)).backward()                       # DDP: bucketed all-reduce, mean over the ranks (train.py:107)
        theirs = {n: p.grad.detach().clone() for n, p in ref.named_parameters()}
        torch.save(dict(mine=mine, theirs=theirs, views_ok=views_ok), out + f".r{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("bf16", [False, True], ids=["fp32-wire", "bf16-wire"])
def test_flat_gradient_all_reduce_equals_ddp_average_on_two_ranks(tmp_path, bf16):
    """the world > 1 arithmetic of make_graphed_ddp_step -- every gradient a view of ONE flat buffer,
    one all-reduce (sum) of it, division by the world size, optionally bf16 on the wire -- against DDP's average, on two
    gloo ranks with different batches."""
    out = str(tmp_path / "flat")
    mp.spawn(_flat_worker, args=(2, _free_port(), out, bf16), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".r0"), torch.load(out + ".r1")
    assert r0["views_ok"] and r1["views_ok"]
    for n in r0["mine"]:
        torch.testing.assert_close(r0["mine"][n], r1["mine"][n], rtol=0, atol=0)          # identical on both ranks
        scale = float(r0["theirs"][n].abs().max()) + 1e-12
        tol = 1.6e-2 if bf16 else 1e-6                       # bf16: 8 significant bits per addend (2^-8 relative) on the wire
        assert float((r0["mine"][n] - r0["theirs"][n]).abs().max()) <= tol * scale, n


def test_spawned_ranks_get_the_ipc_switch_and_a_loopback_rendezvous():
    """what `python bench.py --gpus N` hands its N ranks (train_step.spawn_env) and how a rank binds
    its device (launch_plan -> local rank)."""
    env = ts.spawn_env({"PATH": "/usr/bin"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["MASTER_ADDR"] == "127.0.0.1" and env["PATH"] == "/usr/bin"
    assert ts.spawn_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"     # the caller's choice wins
    for rank in range(8):
        kind, world, r, local = ts.launch_plan(8, dict(WORLD_SIZE="8", RANK=str(rank), LOCAL_RANK=str(rank)), 8, [], "b", 1)
        assert (kind, world, r, local) == ("run", 8, rank, rank)
    import inspect
    import bench
    src = inspect.getsource(bench.main)
    assert "ts.spawn_env(os.environ)" in src and "torch.cuda.set_device(local)" in src


def test_capturable_optimizer_takes_the_schedule_through_a_tensor():
    """a replayed optimizer step reads its learning rate at replay time only if the rate is a tensor;
    set_lr writes into it (and replaces plain floats)."""
    m = TinySeg()
    opt = ts.make_optimizer(m, lr=1e-2)
    ts.set_lr(opt, 5e-3)
    assert all(g["lr"] == 5e-3 for g in opt.param_groups)
    g0 = opt.param_groups[0]
    g0["lr"] = torch.tensor(1e-2)
    ts.set_lr(opt, 2.5e-3)
    assert torch.is_tensor(g0["lr"]) and abs(float(g0["lr"]) - 2.5e-3) < 1e-9

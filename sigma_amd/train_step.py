"""The reference's training step and its multi-GPU plumbing, restated for reuse by bench.py and
the CPU (gloo) tests.

Reference: train.py:59-63 (seed = local rank), :82-107 (model, optimizer, DistributedDataParallel
with find_unused_parameters=False), :160-172 (loss = model(imgs, modal_xs, gts); all-reduce of the
loss for logging; zero_grad; backward; step) and utils/init_func.py:33-58 (optimizer groups).

The forward of an image pair is independent of every other pair (LayerNorm only, no SyncBN), so
the path shards per image: one process per GPU, a full replica each, ONE collective on the data
path -- DDP's bucketed gradient all-reduce (RCCL over xGMI when backend == "nccl") -- plus the
4-byte loss all-reduce the reference does for logging.
"""
from __future__ import annotations

import time
from typing import Callable, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


def group_weight(module: nn.Module, lr: float, include_raw_params: bool = False):
    """Optimizer groups of the reference (utils/init_func.py:33-58): Linear/conv weights decay,
    norms and biases do not; raw nn.Parameters owned directly by the Mamba blocks (x_proj_weight, dt_projs_*,
    A_logs, Ds, A_log_*, D_*, scale1/2) end up in no group and are therefore never stepped --
    reproduced on purpose (the benchmark times the reference's step).  ``include_raw_params=True`` (for
    anyone reusing this for real training) puts them into the no-decay group instead."""
    decay, no_decay = [], []
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d)):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif isinstance(m, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            if m.weight is not None:
                no_decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
    if include_raw_params:
        seen = {id(p) for p in decay} | {id(p) for p in no_decay}
        no_decay += [p for p in module.parameters() if id(p) not in seen]
    return [dict(params=decay, lr=lr), dict(params=no_decay, weight_decay=0.0, lr=lr)]


def unoptimized_parameters(module: nn.Module, groups) -> int:
    """number of parameter tensors that receive gradients but sit in no optimizer group"""
    seen = {id(p) for g in groups for p in g["params"]}
    return sum(1 for p in module.parameters() if p.requires_grad and id(p) not in seen)


def make_optimizer(model: nn.Module, lr: float = 6e-5, weight_decay: float = 0.01, capturable: bool = False,
                   include_raw_params: bool = False):
    """AdamW as configured by the reference (train.py:95-100, configs/config_nyu.py:97-100).  On the GPU
    the single-kernel ("fused") implementation of the same update is used: the multi-tensor default spends
    8.5 ms per step on 70 M parameters (profiles/r02_step_profile.txt), ~25x the bytes it has to move."""
    groups = group_weight(model, lr, include_raw_params)
    on_gpu = any(p.is_cuda for g in groups for p in g["params"])
    extra = dict(fused=True, capturable=capturable) if on_gpu else {}
    if on_gpu and capturable:
        # A captured optimizer step replays whatever learning rate it was captured with: a Python float is baked into
        # the graph, and the reference rewrites param_groups[i]['lr'] every iteration (WarmUpPolyLR, train.py:173-177).
        # As a device tensor the rate is read at replay time: ``set_lr`` (below) writes the schedule's value into it.
        dev = next(p.device for g in groups for p in g["params"] if p.is_cuda)
        for g in groups:
            g["lr"] = torch.tensor(float(lr), device=dev, dtype=torch.float32)
        lr = torch.tensor(float(lr), device=dev, dtype=torch.float32)
    return torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, **extra)


def _tensor_lrs(opt) -> None:
    """Before a capture: every group's learning rate becomes a device tensor (a Python float -- an optimizer not built by
    ``make_optimizer(capturable=True)``, a float-lr checkpoint loaded with ``load_state_dict``, a manual
    ``param_groups[i]['lr'] = v`` -- would be baked into the graph and ``set_lr`` would silently do nothing)."""
    for g in opt.param_groups:
        if not torch.is_tensor(g["lr"]):
            dev = next((p.device for p in g["params"] if p.is_cuda), None)
            if dev is None:
                raise RuntimeError("graphed step: optimizer group without GPU parameters")
            g["lr"] = torch.tensor(float(g["lr"]), device=dev, dtype=torch.float32)


def set_lr(opt, lr: float) -> None:
    """The reference's per-iteration ``optimizer.param_groups[i]['lr'] = lr`` (train.py:173-177) for optimizers of
    ``make_optimizer``: tensor rates (capturable, i.e. graph-replayed steps) are written in place so that the next
    replay reads the new value; float rates are replaced."""
    for g in opt.param_groups:
        if torch.is_tensor(g["lr"]):
            g["lr"].fill_(float(lr))
        else:
            g["lr"] = float(lr)


def _distributed() -> bool:
    """collectives are used exactly when a process group exists (world size 1 included: plumbing checks)"""
    return dist.is_available() and dist.is_initialized()


def _params_changed_behind_autograd() -> None:
    """Parameters were just written in a way autograd's version counters do not see (a broadcast through ``p.data``, a
    replayed optimizer graph): drop the SS2D blocks' cached parameter-derived tensors (ss2d_fused._derived_params)."""
    from .ss2d_fused import invalidate_derived_params
    invalidate_derived_params()


def wrap_ddp(model: nn.Module, device: torch.device) -> nn.Module:
    """train.py:107.  find_unused_parameters=False: every parameter must receive a gradient."""
    if not _distributed():
        return model
    ids = [device.index] if device.type == "cuda" else None
    import os
    # gradient_as_bucket_view: gradients live in the all-reduce buckets (no grad -> bucket copy, 279 MB per step for
    # sigma_small); bucket size: xGMI rings are per-link bound, fewer and larger all-reduces amortise their latency.
    # Environment knobs for A/B runs only.
    kw = dict(gradient_as_bucket_view=os.environ.get("SIGMA_DDP_VIEW", "1") == "1",
              bucket_cap_mb=int(os.environ.get("SIGMA_DDP_BUCKET_MB", "25")),
              static_graph=os.environ.get("SIGMA_DDP_STATIC", "0") == "1")
    net = nn.parallel.DistributedDataParallel(model, device_ids=ids, output_device=ids[0] if ids else None,
                                              find_unused_parameters=False, **kw)
    _params_changed_behind_autograd()       # the constructor broadcasts rank 0's parameters into the replicas
    if os.environ.get("SIGMA_DDP_BF16", "0") == "1":
        # opt-in gradient compression: buckets travel as bf16, the optimizer still sees fp32 gradients
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        net.register_comm_hook(None, default_hooks.bf16_compress_hook)
    return net


def make_step(net: nn.Module, opt, batch: Tuple[torch.Tensor, ...]) -> Callable[[], torch.Tensor]:
    """One training step of train.py:160-172 on a fixed (synthetic) batch."""
    def step():
        opt.zero_grad(set_to_none=True)
        loss = net(*batch)
        if _distributed():                                   # train.py:168 (logging all-reduce)
            red = loss.detach().clone()
            dist.all_reduce(red, op=dist.ReduceOp.SUM)
        loss.backward()
        opt.step()
        return loss
    return step


def make_graphed_step(net: nn.Module, opt, batch: Tuple[torch.Tensor, ...], warmup: int = 3):
    """The same step as ``make_step`` (single process), captured ONCE as a HIP graph and replayed:
    forward + backward + AdamW are ~5.5 k kernel launches whose host-side issue time bounds the step
    when a GPU holds one image (the reference's faithful 8-GPU split, dataloader/dataloader.py:79):
    a replay removes it.  Requirements met by this path: static shapes, no host
    synchronisation inside the step, every kernel launched on torch's current stream (the C ABI takes
    the stream), allocations from torch's graph-private pool, optimizer built with capturable=True.

    Returns (step, static_batch): write new data into ``static_batch`` in place, then call ``step()``;
    the returned loss tensor is overwritten by every replay."""
    if _distributed():
        raise RuntimeError("make_graphed_step: single-process only (DDP buckets are not captured)")
    static = tuple(t.clone() for t in batch)
    _tensor_lrs(opt)
    from .gemm import gemm_mode, selftest
    from .selective_scan_cuda_core import rowlane_selftest
    if gemm_mode() == "split3":
        selftest(static[0].device)                           # not inside the capture (warmup=0)
    from .ss2d_fused import rowlane_possible
    if rowlane_possible():                                   # not when SIGMA_CKPT_PITCH forbids the row-lane kernels
        rowlane_selftest(static[0].device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                            # warm-up off the capture stream: lazy inits, LDS caps
        for _ in range(warmup):
            opt.zero_grad(set_to_none=True)
            net(*static).backward()
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = net(*static)
        loss.backward()
        opt.step()

    def step():
        graph.replay()
        _params_changed_behind_autograd()   # the replay stepped the parameters without touching their version counters
        return loss
    step.graph = graph
    step.static = static            # the graph reads these buffers: they live as long as the step does
    step.set_lr = lambda lr: set_lr(opt, lr)      # edits of param_groups[i]['lr'] with a float would be ignored by the replay
    return step, static


def flatten_grads(model: nn.Module) -> torch.Tensor:
    """One contiguous fp32 buffer holding every gradient; each ``p.grad`` becomes a view of it (autograd accumulates
    into an existing ``.grad`` in place, so the views survive backward).  The gradient all-reduce of the step is then
    ONE collective over 279 MB (sigma_small) instead of DDP's 25 MB buckets: xGMI rings are per-link bound, a single
    large all-reduce amortises their latency (the bucket overlap with backward is given up: at one image per GPU the
    all-reduce is ~3 ms of a ~55 ms step).  Never call ``zero_grad(set_to_none=True)`` afterwards: zero the buffer."""
    params = [p for p in model.parameters() if p.requires_grad]
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, device=params[0].device, dtype=torch.float32)
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    return flat


def _allreduce_mean(flat: torch.Tensor, bf16: bool = False) -> None:
    """mean of the flat gradient over the ranks (train.py:107: DDP averages); ``bf16``: the gradient-compression
    variant (bf16 on the wire, fp32 in the optimizer)"""
    world = dist.get_world_size()
    if bf16:
        wire = flat.to(torch.bfloat16)
        dist.all_reduce(wire, op=dist.ReduceOp.SUM)
        flat.copy_(wire)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if world > 1:
        flat.div_(world)


def make_graphed_ddp_step(model: nn.Module, opt, batch: Tuple[torch.Tensor, ...], warmup: int = 3, bf16_comm: bool = False):
    """``make_graphed_step`` for the multi-GPU job (reference: train.py:107, 164-172 with one image per
    GPU, dataloader/dataloader.py:79).  DDP's bucket hooks cannot be replayed from a graph, so the data-parallel step is
    restated around two HIP graphs:

        graph A: zero the flat gradient buffer, forward, backward (gradients land in views of ONE flat buffer)
        eager  : the 4-byte loss all-reduce of train.py:168 and ONE RCCL all-reduce (mean) of the flat buffer
        graph B: fused AdamW step

    `model` is the UNWRAPPED module (its parameters are broadcast from rank 0 here, as DDP's constructor does); the
    optimizer must have been built with capturable=True (``make_optimizer``: its learning rate is then a device tensor,
    ``step.set_lr(v)`` feeds the schedule; float edits of param_groups would be ignored by the replay).  Optimizer state
    from earlier un-synchronised steps is discarded here, so that the replicas start identical.  ``step()`` returns the
    local loss; ``step.reduced_loss`` holds the mean over the ranks the reference logs (train.py:168-170).
    Returns (step, static_batch)."""
    if not _distributed():
        raise RuntimeError("make_graphed_ddp_step needs an initialised process group (world size 1 is fine)")
    for p in model.parameters():
        dist.broadcast(p.data, 0)
    for b in model.buffers():
        dist.broadcast(b.data, 0)
    _params_changed_behind_autograd()
    opt.state.clear()               # exp_avg / exp_avg_sq / step of steps taken before the broadcast are rank-specific
    static = tuple(t.clone() for t in batch)
    flat = flatten_grads(model)
    _tensor_lrs(opt)
    from .gemm import gemm_mode, selftest
    from .selective_scan_cuda_core import rowlane_selftest
    if gemm_mode() == "split3":
        selftest(static[0].device)
    from .ss2d_fused import rowlane_possible
    if rowlane_possible():                                   # not when SIGMA_CKPT_PITCH forbids the row-lane kernels
        rowlane_selftest(static[0].device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            flat.zero_()
            model(*static).backward()
            _allreduce_mean(flat, bf16_comm)
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    # The process group's watchdog thread polls the events of collectives in flight (hipEventQuery); under the default
    # "global" capture mode such a call from another thread invalidates the capture and poisons the communicator (the
    # process then aborts in destroy_process_group).  So: no collective is in flight when the capture starts, and the
    # capture only polices its own thread.
    torch.cuda.synchronize()
    g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_fb, capture_error_mode="thread_local"):
        flat.zero_()
        loss = model(*static)
        loss.backward()
    with torch.cuda.graph(g_opt, pool=g_fb.pool(), capture_error_mode="thread_local"):
        opt.step()

    world = dist.get_world_size()

    def step():
        g_fb.replay()
        red = loss.detach().clone()                          # train.py:168 (logging all-reduce)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        step.reduced_loss = red / world
        _allreduce_mean(flat, bf16_comm)
        g_opt.replay()
        _params_changed_behind_autograd()
        return loss
    step.reduced_loss = None
    step.set_lr = lambda lr: set_lr(opt, lr)
    step.graphs = (g_fb, g_opt)
    step.flat = flat
    step.static = static            # the graphs read these buffers: they live as long as the step does
    return step, static


def _sync(device: torch.device) -> None:
    if _distributed():
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step: Callable[[], torch.Tensor], steps: int, warmup: int, device: torch.device,
                on_timed_start: Callable[[], None] = lambda: None):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize pairs.
    Returns (seconds of the slowest rank, last loss)."""
    loss = None
    for _ in range(warmup):
        loss = step()
    _sync(device)
    on_timed_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    _sync(device)
    elapsed = time.perf_counter() - t0
    if _distributed():
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, loss


def launch_plan(gpus, env, n_devices: int, argv, script: str, port: int):
    """What `python bench.py --gpus N` must do: decide between running in this
    process, re-launching itself as N ranks, or refusing.

    Returns ("run", world, rank, local_rank) when this process IS a rank (world == gpus is asserted:
    a launcher that started fewer ranks than --gpus must not pass for an N-GPU measurement), or
    ("spawn", cmd) with the torch.distributed.run command line (one process per GPU, rendezvous on
    127.0.0.1 -- the container hostname may not resolve) when --gpus > 1 and no launcher environment
    exists.  Raises SystemExit if the node has fewer devices than ranks.  Reference: train.py:59-63,
    engine/engine.py:59-75 (torch.distributed.launch environment)."""
    import sys as _sys
    if gpus is None:                 # flag omitted: whatever the launcher started (1 without a launcher)
        gpus = int(env.get("WORLD_SIZE", "1"))
    if gpus < 1:
        raise SystemExit(f"--gpus must be >= 1 (got {gpus})")
    if "WORLD_SIZE" in env:
        world, rank = int(env["WORLD_SIZE"]), int(env.get("RANK", "0"))
        local = int(env.get("LOCAL_RANK", str(rank)))
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
        if n_devices < world or local >= n_devices:
            raise SystemExit(f"{world} ranks need {world} GPUs on this node, found {n_devices}")
        return ("run", world, rank, local)
    if gpus == 1:
        if n_devices < 1:
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        return ("run", 1, 0, 0)
    if n_devices < gpus:
        raise SystemExit(f"--gpus {gpus} but only {n_devices} GPU(s) are visible on this node")
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]
    return ("spawn", cmd)


def spawn_env(env) -> dict:
    """Environment of the N ranks `python bench.py --gpus N` starts: the caller's, plus the switch without which RCCL /
    device-memory sharing across processes fails on this platform (`hipIpcGetMemHandle: invalid argument`: the host
    driver only supports dmabuf IPC) unless the caller has set it, and a loopback rendezvous (the container hostname
    may not resolve).  Each rank then binds GPU LOCAL_RANK (``launch_plan`` returns it; bench.py calls
    ``torch.cuda.set_device(local)`` before anything touches the device).  Reference: train.py:59-63."""
    out = dict(env)
    out.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out.setdefault("MASTER_ADDR", "127.0.0.1")
    return out


def throughput(per_rank_batch: int, world: int, steps: int, elapsed: float) -> float:
    """Whole-job images/s under weak scaling: every rank processes per_rank_batch pairs per step."""
    return per_rank_batch * world * steps / elapsed

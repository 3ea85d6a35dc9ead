#!/usr/bin/env python3
"""Headline benchmark: images/sec of Sigma's training step (fwd + bwd + AdamW) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1 default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`, configs[2]/[3]): sigma_small, synthetic RGB-X pairs of
3x480x640 (+ labels of 40 classes, NYU shape), fp32 like the reference, one process per GPU,
DistributedDataParallel over RCCL for N > 1 (gradient all-reduce only; the forward is per-image
data parallel).  Per-GPU batch is fixed (weak scaling); at N = 1 the global batch equals the
reference's config batch_size = 8 (configs/config_nyu.py:101).

One step = loss = model(rgb, x, label); zero_grad; backward; AdamW step; the scalar loss
all-reduce of train.py:168.  W untimed steps, then exactly K steps between barrier +
synchronize pairs; the slowest rank's time counts.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- dominant HIP kernel (largest share of scan time in the timed region):
                  achieved = SURVEY 8(d) algorithmic bytes of that launch shape / mean launch
                  duration, measured with HIP events on the launch stream inside the timed steps
  cpu_baseline -- the CPU oracle (C port of the reference's selective_scan_ref + its adjoint)
                  timed on this host's cores on a bounded sample, in images/s
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", "--per-gpu-batch", dest="batch", type=int, default=8,
                    help="per-GPU batch of RGB-X pairs (8 = weak scaling from the reference's batch_size 8; "
                         "1 = the reference's faithful 8-GPU split, dataloader/dataloader.py:79)")
    ap.add_argument("--backbone", default="sigma_small")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--classes", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-autotune", action="store_true", help="torch.backends.cudnn.benchmark = True (slow start)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="single process: still create the RCCL process group and wrap the model in DDP (plumbing check)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step (fwd + bwd + AdamW) as one HIP graph and time replays (single process); the "
                         "roofline object then comes from a few eager steps run before the capture")
    ap.add_argument("--gemm", default="", choices=["", "fp32", "split3"],
                    help="nn.Linear GEMMs: split3 = hand-written split-operand bf16 MFMA kernels (csrc/gemm_split.hip), fp32 = vendor "
                         "fp32 GEMMs; default: sigma_amd.gemm.gemm_mode()")
    ap.add_argument("--strict-tuned", action="store_true",
                    help="exit with an error when PyTorch rejects the committed TunableOp GEMM table (default: warn on stderr and "
                         "report config.tuned_gemms = false)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="(unused: the CPU sample is fixed) kept for compatibility")
    ap.add_argument("--kernel-report", default="", help="write the per-kernel-shape table (json) here")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of every scan launch inside the timed region, on the launch stream."""

    def __init__(self):
        self.enabled = False
        self.records = []       # (kind, shape_key, start_event, end_event)

    def __call__(self, kind, key, launch):
        if not self.enabled:
            return launch()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        out = launch()
        e.record()
        self.records.append((kind, key, s, e))
        return out

    def table(self):
        agg = {}
        for kind, key, s, e in self.records:
            d = agg.setdefault((kind, key), [0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e) * 1e-3
        return agg


def measured_traffic(kind, shape):
    """HBM bytes per launch of a scan kernel from the committed PMC measurement
    (profiles/scan_traffic.json, produced by tools/gpu_pmc.sh + tools/pmc_traffic.py with the
    guide's gfx950 correction: FETCH_SIZE doubled for wide coalesced reads).  None if that shape
    was not measured -- bench.py itself cannot collect PMC counters."""
    path = os.path.join(ROOT, "profiles", "scan_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except OSError:
        return None
    for e in table.get("entries", []):
        if e["kernel"] == f"scan_{kind}" and list(e["shape"]) == list(shape):
            return e["hbm_bytes_per_launch"]
    return None


def scan_bytes(kind, key):
    B, KD, L, N, G, es = key
    if kind == "fwd":       # SURVEY.md 8(d), training: + checkpoint write
        return es * 3 * B * KD * L + es * 2 * B * G * N * L + 4 * (KD * N + 2 * KD) + 4 * B * KD * ((L + 2047) // 2048) * 2 * N
    return es * 5 * B * KD * L + es * 4 * B * G * N * L


def cpu_baseline(backbone, H, W, budget_s):
    """Images/s of the CPU path's selective scans alone (an upper bound of a full CPU step).

    value: oracle/scan_oracle.c (C port of the reference's selective_scan_ref + its adjoint, OpenMP) on a
    FIXED sample -- one whole launch of every distinct scan shape of the model at batch 1, best of two
    runs -- scaled by call counts.  torch_ref: the reference's
    own CPU fallback algorithm (oracle/scan_ref_torch.py = selective_scan_interface.py:86-131, a Python loop
    over the sequence) timed once on the encoder stage-2 launch shape, forward on a quarter of its rows and
    autograd backward on a sixteenth (scaled), extrapolated to the scans of one image."""
    from oracle import scan_oracle as so
    from oracle import scan_ref_torch as rt
    E = 128 if backbone == "sigma_base" else 96
    depths = [2, 2, 9, 2] if backbone == "sigma_tiny" else [2, 2, 27, 2]
    shapes = []            # (calls, KD, L, N, G)
    h, w = H // 4, W // 4
    for i in range(4):
        C = E * 2 ** i
        d, L = 2 * C, h * w
        shapes.append((2 * depths[i], 4 * d, L, 16, 4))          # encoder, both modalities
        shapes.append((2, d, L, 4, 1))                           # CroMB
        shapes.append((1, 2 * d, 2 * L, 4, 2))                   # ConMB
        if i < 3:
            shapes.append((4, 4 * d, L, 4, 4))                   # decoder level
        h, w = (h + 1) // 2, (w + 1) // 2
    total_updates = sum(c * kd * L * N for c, kd, L, N, _ in shapes)
    t_total, sampled = 0.0, 0
    t_begin = time.perf_counter()
    g = torch.Generator().manual_seed(0)
    nthr_omp = so.num_threads()
    for calls, KD, L, N, G in shapes:
        # one whole launch of the shape (every row, batch 1).  The oracle's backward parallelises over
        # (batch, group) only, so the KD rows are handed over as `chunks` independent problems of KD/chunks
        # rows each (B/C replicated per chunk: the arithmetic per row is that of the grouped call plus one
        # N x L dB/dC store per chunk) -- every host core works.
        r = KD
        chunks = max(c for c in range(1, min(r, 2 * nthr_omp) + 1) if r % c == 0)
        rc = r // chunks
        u = torch.randn(chunks, rc, L, generator=g)
        delta = 0.5 * torch.randn(chunks, rc, L, generator=g)
        A = -torch.arange(1, N + 1, dtype=torch.float32).repeat(rc, 1)
        Bm = torch.randn(1, 1, N, L, generator=g).expand(chunks, 1, N, L).contiguous()
        Cm = torch.randn(1, 1, N, L, generator=g).expand(chunks, 1, N, L).contiguous()
        D, bias = torch.ones(rc), torch.full((rc,), -4.0)
        best = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            out = so.selective_scan_oracle(u, delta, A, Bm, Cm, D, bias, True)
            so.selective_scan_oracle_bwd(u, delta, A, Bm, Cm, D, bias, out, True)
            best = min(best, time.perf_counter() - t0)
        t_total += best * (KD / r) * calls
        sampled += r * L * N
    t_oracle = time.perf_counter() - t_begin
    # the reference's torch fallback on the dominant launch shape at batch 1: forward on a quarter of its rows,
    # autograd backward on a sixteenth (select-backward materialises a zero tensor of the whole (B,D,L,N) array
    # at every position, so its time is linear in the rows), 16 threads (more threads are slower on ops this small)
    KD2, L2 = 4 * 2 * E * 4, (H // 16) * (W // 16)
    thr = min(16, os.cpu_count() or 1)
    t_begin = time.perf_counter()
    f_s, _, nthr = rt.time_fwd_bwd(1, KD2 // 4, L2, 16, 1, threads=thr, backward=False)
    _, b_s, _ = rt.time_fwd_bwd(1, KD2 // 16, L2, 16, 1, threads=thr)
    t_torch = time.perf_counter() - t_begin
    per_update = (4.0 * f_s + 16.0 * b_s) / (KD2 * L2 * 16)
    torch_ref = dict(shape=[1, KD2, L2, 16, 4], fwd_s_scaled=round(4.0 * f_s, 3), bwd_s_scaled=round(16.0 * b_s, 3), threads=nthr,
                     images_per_s_scans_only=1.0 / (per_update * total_updates), seconds=round(t_torch, 1),
                     note="oracle/scan_ref_torch.py = the reference's selective_scan_ref (selective_scan_interface.py:86-131); "
                          "fwd on 1/4 of the rows x 4, autograd bwd on 1/16 of the rows x 16; extrapolated by state updates")
    return dict(value=1.0 / t_total, unit="images/s", cores=so.num_threads(), kind="port",
                sample=(f"oracle/scan_oracle.c fwd+bwd (OpenMP, {so.num_threads()} threads), best of 2, on one whole launch (every row) "
                        f"of each of the {len(shapes)} distinct scan shapes of {backbone} @{H}x{W}, batch 1 "
                        f"({sampled / total_updates:.1%} of one image's state updates), scaled by call counts; "
                        "scans only, so an upper bound on the CPU path"),
                seconds=round(t_oracle, 1), torch_ref=torch_ref)


def main():
    a = parse()
    from sigma_amd import train_step as ts
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    plan = ts.launch_plan(a.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0,
                          sys.argv[1:], os.path.abspath(__file__), port)
    if plan[0] == "spawn":           # `python bench.py --gpus N`: become N ranks (one per GPU, RCCL)
        raise SystemExit(subprocess.call(plan[1], env=ts.spawn_env(os.environ)))
    _, world, rank, local = plan
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # train.py:57 sets cudnn.benchmark = True; on ROCm that is MIOpen's exhaustive find mode, which
    # for this model's ~60 convolution configurations runs for more than 15 minutes before the
    # first step returns (measured), so the default here is MIOpen's immediate mode.
    torch.backends.cudnn.benchmark = bool(a.conv_autotune)
    ddp = world > 1 or a.force_ddp
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    if a.gemm:
        os.environ["SIGMA_GEMM"] = a.gemm
    from sigma_amd import selective_scan_cuda_core as core
    from sigma_amd.models.builder import EncoderDecoder
    from sigma_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()                           # vendor GEMMs look their solution up in the committed table (explicit, process-wide)

    timer = KernelTimer()
    core.set_launch_hook(timer)

    cfg = types.SimpleNamespace(backbone=a.backbone, decoder="MambaDecoder", num_classes=a.classes,
                                image_height=a.height, image_width=a.width, pretrained_model=None, bn_eps=1e-3,
                                bn_momentum=0.1)
    torch.manual_seed(rank)                                  # train.py:59-63: seed = local rank
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):      # the (absent) pretrained checkpoint message
            model = EncoderDecoder(cfg, criterion=nn.CrossEntropyLoss(reduction="mean", ignore_index=255),
                                   norm_layer=nn.BatchNorm2d)
    finally:
        os.chdir(cwd)
    model.to(dev).train()
    use_graph = bool(a.graph)
    opt = ts.make_optimizer(model, capturable=use_graph)
    # train.py:107; under --graph the data-parallel step is two HIP graphs around one flat RCCL all-reduce
    net = model if (use_graph and ddp) else ts.wrap_ddp(model, dev)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    rgb = torch.randn(a.batch, 3, a.height, a.width, generator=g).to(dev)
    mx = torch.randn(a.batch, 3, a.height, a.width, generator=g).to(dev)
    label = torch.randint(0, a.classes, (a.batch, a.height, a.width), generator=g).to(dev)
    step = ts.make_step(net, opt, (rgb, mx, label))

    def start_timers():
        timer.enabled = True

    if use_graph:
        # kernel timings (HIP events per launch) cannot live inside a captured graph: take them from two
        # eager steps first, then capture and time replays of the identical step.  (Under a process group these eager
        # steps run on the unwrapped model, i.e. without the gradient all-reduce: make_graphed_ddp_step re-broadcasts the
        # parameters and discards the optimizer state afterwards, so the replicas start the timed steps identical.)
        step(); torch.cuda.synchronize()
        timer.enabled = True
        step(); step(); torch.cuda.synchronize()
        timer.enabled = False
        eager_scan_steps = 2
        core.set_launch_hook(None)
        if ddp:
            gstep, _ = ts.make_graphed_ddp_step(model, opt, (rgb, mx, label), bf16_comm=os.environ.get("SIGMA_DDP_BF16", "0") == "1")
        else:
            gstep, _ = ts.make_graphed_step(net, opt, (rgb, mx, label))
        elapsed, loss = ts.timed_steps(gstep, a.steps, a.warmup, dev)
    else:
        # the timed region runs WITHOUT the per-launch HIP events (two events around each of ~270 scan launches per step
        # perturb what they measure); the kernel table comes from two extra eager steps afterwards
        elapsed, loss = ts.timed_steps(step, a.steps, a.warmup, dev)
        eager_scan_steps = 2
        start_timers()
        step(); step(); torch.cuda.synchronize()
        timer.enabled = False

    from sigma_amd.tuning import tuned_gemms_active
    tuned = tuned_gemms_active()
    if not tuned and os.environ.get("SIGMA_TUNED_GEMMS", "1") != "0":
        msg = ("bench.py: the committed TunableOp GEMM table (sigma_amd/tuning/tunableop_mi355x.csv) is NOT active -- PyTorch "
               "rejected it (ROCm / hipBLASLt / rocBLAS version validators) or it is missing; library-default GEMM solutions "
               "are ~17 % slower on this step")
        if a.strict_tuned:
            raise SystemExit(msg)
        print(msg, file=sys.stderr, flush=True)
    if rank == 0:
        table = timer.table()
        rows = []
        for (kind, key), (n, secs) in table.items():
            by = scan_bytes(kind, key)
            rows.append(dict(kernel=f"scan_{kind}", shape=list(key[:5]), launches=n, total_ms=secs * 1e3,
                             avg_us=secs / n * 1e6, algorithmic_MB=by / 1e6, GBs=by / (secs / n) / 1e9))
        rows.sort(key=lambda r: -r["total_ms"])
        scan_ms = sum(r["total_ms"] for r in rows)
        def valu_floor(kind, shape):
            """The bound behind the HBM fraction: the S6 recurrence is vector-ALU work.  Element-states of the
            launch / (64 lanes x 1024 SIMDs) x the issue cost of the arithmetic alone: forward 4 plain + 1 transcendental,
            backward 12 plain + 1 transcendental + 1 lane swap per element-state.  `nominal`: 2.1 / 8.1 / 8.1 clocks at 2.1
            GHz (profiles/r03_issue_ubench.jsonl); `measured`: wall-clock cost of the same mix inside a loop at three waves
            per SIMD (tools/ubench/stateloop_ubench.hip, profiles/r06_stateloop_ubench.jsonl: plain 1.0 ns, v_exp_f32 ~5 ns,
            lane swap ~4.8 ns per wave instruction and SIMD)."""
            B, KD, L, N = shape[0], shape[1], shape[2], shape[3]
            waves = B * KD * L * N / 64.0 / 1024.0
            if kind == "fwd":
                nominal_clk, measured_ns = 4 * 2.1 + 8.1, 9.97
            else:
                nominal_clk, measured_ns = 12 * 2.1 + 8.1 + 8.1 + 2.1, 12 * 1.0 + 5.0 + 4.8
            return waves * nominal_clk / 2.1e3, waves * measured_ns * 1e-3          # microseconds

        roof = None
        if rows:
            d = rows[0]
            roof = dict(bound="hbm", achieved=round(d["GBs"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(d["GBs"] / HBM_PEAK_GBS, 4), traffic=measured_traffic(d["kernel"][5:], d["shape"]),
                        traffic_source="profiles/scan_traffic.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE of this round's kernels, "
                                       "tools/gpu_pmc.sh; bench.py itself cannot read counters)",
                        algorithmic_bytes=int(d["algorithmic_MB"] * 1e6), kernel=d["kernel"], shape=d["shape"],
                        avg_launch_us=round(d["avg_us"], 1), launches=d["launches"],
                        share_of_scan_time=round(d["total_ms"] / scan_ms, 3),
                        scan_share_of_step=round((scan_ms / eager_scan_steps) / (elapsed / a.steps * 1e3), 3))
            vf_nom, vf_meas = valu_floor(d["kernel"][5:], d["shape"])
            roof.update(valu_floor_us=round(vf_nom, 1), valu_frac=round(vf_nom / d["avg_us"], 4),
                        valu_floor_measured_us=round(vf_meas, 1), valu_frac_measured=round(vf_meas / d["avg_us"], 4))
        roof_fwd = None
        fwd_rows = [r for r in rows if r["kernel"] == "scan_fwd"]
        if fwd_rows:
            d = fwd_rows[0]
            roof_fwd = dict(bound="hbm", achieved=round(d["GBs"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(d["GBs"] / HBM_PEAK_GBS, 4), traffic=measured_traffic("fwd", d["shape"]),
                            algorithmic_bytes=int(d["algorithmic_MB"] * 1e6), kernel=d["kernel"], shape=d["shape"],
                            avg_launch_us=round(d["avg_us"], 1), launches=d["launches"])
            vf_nom, vf_meas = valu_floor("fwd", d["shape"])
            roof_fwd.update(valu_floor_us=round(vf_nom, 1), valu_frac=round(vf_nom / d["avg_us"], 4),
                            valu_floor_measured_us=round(vf_meas, 1), valu_frac_measured=round(vf_meas / d["avg_us"], 4))
        if a.kernel_report:
            os.makedirs(os.path.dirname(os.path.abspath(a.kernel_report)) or ".", exist_ok=True)
            with open(a.kernel_report, "w") as f:
                json.dump(dict(steps=a.steps, elapsed_s=elapsed, scan_ms_total=scan_ms, kernels=rows), f, indent=1)
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a.backbone, a.height, a.width, a.cpu_budget)
        line = dict(metric=f"images/sec fwd+bwd {a.backbone} {a.height}x{a.width}", value=round(ts.throughput(a.batch, world, a.steps, elapsed), 3),
                    unit="images/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                    ms_per_step=round(elapsed / a.steps * 1e3, 2), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=f"{a.backbone} training step (fwd+bwd+AdamW), RGB-X pairs {a.height}x{a.width}, "
                                         f"{a.classes} classes, fp32", per_gpu_batch=a.batch,
                                global_batch=a.batch * world, parallelism=f"dp{world}", hip_graph=use_graph,
                                tuned_gemms=bool(tuned),
                                gemm=getattr(model, "gemm_mode", "fp32"),
                                loss=round(float(loss.item()), 4)),
                    roofline=roof, roofline_fwd=roof_fwd, cpu_baseline=cpu)
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

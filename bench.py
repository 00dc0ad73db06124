"""Benchmark of the CDDMSL training step on MI355X (BASELINE.json metric: images/sec of the train step).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run, one rank per GPU)

Workload (config.workload): configs[1] of BASELINE.json -- VOC(labeled)+Clipart(unlabeled) CLIP RN50-C4 Faster R-CNN +
caption consistency, 16 images/GPU of 800x1333, bf16 MFMA, iteration > 10000 so all three branches of
``SimpleTrainer.run_step`` are live (supervised + image-level + region-level consistency, backward, gradient
all-reduce, per-parameter clip + SGD).  Synthetic inputs and seeded random weights (no datasets/checkpoints offline);
inputs are resident in HBM before the timed region.

One JSON line on rank 0 with ``roofline`` (dominant kernel = the bf16 implicit-GEMM conv; achieved = algorithmic
2*M*N*K FLOPs of its launches / their HIP-event durations, measured live over the timed steps on the launch stream)
and ``cpu_baseline`` (the oracle = PyTorch-CPU restatement of the same step, a bounded sample timed on this box's
host cores, rank 0 / N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # (see cddmsl_amd/__init__.py; set before anything can initialise the HIP runtime)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_cfg(dtype, kd=False):
    from cddmsl_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", dtype, "MODEL.KD_REGULRAZIATION", kd])
    return cfg


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(height, width, threads, batch=2, timed=3, budget_s=150.0):
    """Oracle (kind 'port') as SURVEY.md 8(d) defines the CPU baseline: the full training step (3 branches + backward +
    clip/SGD) on B=2 images, CPU fp32, one warm-up step, then the median of ``timed`` steps -- bounded: timing stops
    early once ``budget_s`` seconds of timed steps are spent (at least one timed step always runs; the sample says how many)."""
    from cddmsl_amd import synthetic
    from oracle import model as om
    torch.set_num_threads(threads)
    sd = synthetic.make_state_dict(0)
    msd = synthetic.make_mapper_state_dict(1)
    cfg = om.Cfg()
    keys = om.trainable_keys(sd, cfg)
    for k in keys:
        sd[k].requires_grad_(True)
    data = synthetic.make_batch(batch, height, width)
    gen = torch.Generator().manual_seed(1)
    mom = {}

    def step():
        for k in keys:
            sd[k].grad = None
        t0 = time.perf_counter()
        ld = om.run_step_losses(sd, msd, cfg, data, 20000, gen)
        sum(ld.values()).backward()
        grads = {k: sd[k].grad for k in keys}
        with torch.no_grad():
            plain = {k: v.detach() for k, v in sd.items()}
            om.sgd_step(plain, grads, mom, cfg, 20000)
        return time.perf_counter() - t0

    warm = step()
    times = []
    while len(times) < timed and (not times or sum(times) + times[-1] <= budget_s):
        times.append(step())
    med = sorted(times)[len(times) // 2]
    return {"value": batch / med, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"B={batch} images {height}x{width}, full step (3 branches, backward, clip+SGD), oracle fp32 on "
                      f"{threads} threads of {cpu_model_name()}; 1 warm-up ({warm:.1f} s) then median of {len(times)} "
                      f"timed steps ({', '.join('%.1f' % t for t in times)} s)"}


def visible_gpu_count(env=os.environ):
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the parent of the ranks must not hold a device): the KFD
    topology's nodes with SIMDs (CPU nodes have simd_count 0), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when set.  Falls back to torch.cuda.device_count() where the sysfs tree is absent."""
    n = None
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(root):
            with open(os.path.join(root, d, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        n += 1 if int(line.split()[1]) > 0 else 0
                        break
    except OSError:
        n = None
    if n is None:
        return torch.cuda.device_count()
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_plan(gpus, argv, env, device_count):
    """engine/launch.py:27-82 for the bench: WHO runs the ranks.

    Returns ("run", None) when this process is itself a rank (or the single-GPU run), ("spawn", cmd) when it has to start
    ``torch.distributed.run`` with one child per GPU -- decided BEFORE anything touches the GPU: ``device_count`` comes from
    ``visible_gpu_count`` (sysfs, not the HIP runtime), so the parent never holds a device; a rank whose device does not exist
    fails at ``set_device`` -- and raises SystemExit with a message when the request cannot be met (fewer devices than --gpus, or a
    launcher-provided WORLD_SIZE that contradicts --gpus).  Pure function of its arguments (tests/test_cabi_host.py)."""
    gpus = max(int(gpus), 1)
    if "RANK" in env:
        world = int(env.get("WORLD_SIZE", "1"))
        if world != gpus:
            raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {gpus}: refusing to report a number for the wrong rank count")
        return "run", None
    if gpus == 1:
        return "run", None
    if not env.get("CDDMSL_SHARE_GPU") and device_count < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} needs {gpus} visible GPUs, this box has {device_count} (one RCCL rank per device)")
    port = env.get("MASTER_PORT", "29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8 = BASELINE.json configs[4]: e4m3 forward GEMMs (MFMA-bound convs of layer3/4, RoI layer4, RPN; region x text "
                         "contraction), bf16 elsewhere and in backward; use with --batch 32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-roofline", action="store_true", help="skip the extra forward-only loop (profiling runs: keeps the kernel mix = the timed steps)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-baseline-steps", type=int, default=3, help="timed oracle steps (median reported), after one warm-up")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal: start the ranks, form the process group (gloo when "
                    "no GPU is visible), barrier, print the rank layout; no model, no timing")
    args = ap.parse_args()

    # one process per GPU: with --gpus N > 1 and no launcher around us, start N ranks as children (before any GPU call)
    mode, cmd = launch_plan(args.gpus, sys.argv[1:], os.environ, visible_gpu_count() if args.gpus > 1 and "RANK" not in os.environ else 1)
    if mode == "spawn":
        sys.exit(subprocess.call(cmd))

    from cddmsl_amd import engine, hip
    rank, world = engine.init_distributed()
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: process group has {world} ranks but --gpus {args.gpus}")
    if args.dry_run:
        ranks = [None] * world
        if world > 1:
            dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "0"))))
            dist.barrier()
        else:
            ranks = [(0, 0)]
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": world, "ranks": ranks,
                              "dist_backend": dist.get_backend() if world > 1 else None}))
        if world > 1:
            dist.destroy_process_group()
        return
    # CDDMSL_SHARE_GPU=1 (rehearsal only): all ranks on device 0 with the gloo backend, to exercise the N>1 code path
    # on a one-GPU box; RCCL itself needs one device per rank.
    dev = torch.device("cuda", 0 if os.environ.get("CDDMSL_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    cfg = make_cfg(args.dtype)
    cfg.MODEL.DEVICE = str(dev)

    from cddmsl_amd import synthetic
    from cddmsl_amd.modeling import TransformerMapper
    tr = engine.build_trainer(cfg, args.batch, args.height, args.width)
    tr.model.load_state_dict(synthetic.make_state_dict(0), strict=False)
    tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
    tr.iter = 20000          # past burn-in (train_loop.py:334): every branch is computed AND contributes gradients
    tr.metrics_period = 0    # no per-step host sync inside the timed region (losses stay on device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events cost: a pair around every one of the step's ~1000 library launches slows the step by ~5 % (142 vs 135 ms),
    # so the timed region carries events only on the launches of the DOMINANT kernel (the roofline object's `achieved`), and
    # only in its last step (every step launches the same shapes);
    # which kernel that is, and the per-kernel table, come from one fully instrumented untimed step (the last warm-up step).
    full = None
    step0 = None                                       # loss dict of the very first step (device tensors; read after the timed region)
    for i in range(args.warmup):
        if i == args.warmup - 1:
            hip.PROFILE.enable()
        ld_ = tr.run_step()
        if step0 is None:
            step0 = ld_
    if args.warmup > 0:
        full = hip.PROFILE.collect()
    gemm_names = ("k_conv_fwd256", "k_conv_fwd2", "k_conv_fwd", "k_wgrad256", "k_conv_wgrad_dma") if args.dtype != "fp8" else ("k_conv_fwd256_fp8",)
    dom_name = "k_conv_fwd256"
    if full:
        rows = [k for k in gemm_names if k in full]
        if rows:
            dom_name = max(rows, key=lambda k: full[k]["ms"])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:                        # the dominant kernel's launches of the LAST timed step carry the events
            hip.PROFILE.enable(only={dom_name})
        last = tr.run_step()
        if step0 is None:
            step0 = last
    barrier()
    dt = time.perf_counter() - t0
    dom_shapes = hip.PROFILE.by_shape()
    prof = hip.PROFILE.collect()
    prof_steps = 1
    if full is None:                                   # --warmup 0: the instrumented step runs after the timed region
        hip.PROFILE.enable()
        tr.run_step()
        full = hip.PROFILE.collect()
    # RN50-C4 supervised FORWARD alone (backbone to res4, RPN, RoIAlign, RoI layer4, attention pool, classifier, losses), the
    # quantity BASELINE.json's roofline target is stated on: algorithmic 1.878 TFLOP per 800x1333 image (SURVEY.md 8(d):
    # 938.8 GMAC, query-0-only attention pool), timed outside the step timing above, rank 0's own clock
    fwd_ms = fwd_kernels = fwd_shapes = None
    if world == 1 and (args.height, args.width) == (800, 1333) and not args.no_forward_roofline:
        data = next(tr._data_loader_iter)
        tr.model.share_source_pass = False
        for _ in range(2):
            tr.model(data)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(5):
            tr.model(data)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - tf0) / 5 * 1e3
        hip.PROFILE.enable()                            # one more, fully instrumented: the forward-only per-kernel table
        tr.model(data)
        fwd_shapes = hip.PROFILE.by_shape()
        fwd_kernels = hip.PROFILE.collect()
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    devs = [torch.cuda.current_device()]
    if world > 1:
        dl = [None] * world
        dist.all_gather_object(dl, torch.cuda.current_device())
        devs = dl
    losses = {k: float(v.detach()) for k, v in last.items()}
    # Outside the timed region: the FIRST step's losses (seeded weights, seeded batch, seeded samplers) against the fixture that
    # tests/test_gpu_bench_gate.py ties to the exact-f32 step at these very launch shapes -- the bench is timing the gated computation.
    gate = None
    fpath = os.path.join(ROOT, "tests", "golden", "bench_step0_losses.json")
    fkey = f"{args.dtype}_b{args.batch}_{args.height}x{args.width}"
    if world == 1 and os.path.exists(fpath):
        fx = json.load(open(fpath)).get(fkey)
        if fx is not None:
            s0 = {k: float(v.detach()) for k, v in step0.items()}
            dev = max(abs(s0[k] - v) / (abs(v) + 5e-2) for k, v in fx.items())
            gate = {"fixture": "tests/golden/bench_step0_losses.json:" + fkey, "step0_losses": s0, "max_rel_dev": dev, "tolerance": 2e-2, "ok": dev <= 2e-2}

    if rank == 0:
        gb = args.batch * world
        # dominant kernel = the single kernel with the most device time in the instrumented step (k_conv_fwd256 on this
        # workload); the C-ABI reports which kernel a conv / GEMM entry point launches (cddmsl_last_kernel / cddmsl_plan_only),
        # so a profiler row is one kernel, and `dom` holds the HIP-event times of that kernel's launches in the TIMED steps
        dom = prof.get(dom_name, {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0})
        # HBM traffic per launch of that kernel from the PMC passes (tools/profile_round.sh; FETCH_SIZE x2 + WRITE_SIZE)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if args.batch == 16 and args.dtype == "bf16" and os.path.exists(tpath):   # (measured for the bf16 workload only)
            traffic = json.load(open(tpath)).get(dom_name, {}).get("hbm_bytes_per_launch")
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        peak = {"bf16": 2500.0, "fp8": 5000.0}.get(args.dtype, 157.3)
        # the dominant kernel serves layers on both sides of the ridge: each launch is booked under ITS bound (the larger of
        # 2MNK / MFMA peak and algorithmic bytes / HBM peak) and the two groups are reported against their own roofline
        split = {"mfma": [0, 0.0, 0.0, 0.0], "hbm": [0, 0.0, 0.0, 0.0]}
        for (kname, _shape), (n_, ms_, fl_, by_) in dom_shapes.items():
            if kname != dom_name:
                continue
            g_ = split["mfma" if fl_ / (peak * 1e12) >= by_ / 8.0e12 else "hbm"]
            g_[0] += n_; g_[1] += ms_; g_[2] += fl_; g_[3] += by_
        by_bound = {}
        for b_, (n_, ms_, fl_, by_) in split.items():
            if ms_ > 0:
                a_ = fl_ / (ms_ * 1e-3) / 1e12 if b_ == "mfma" else by_ / (ms_ * 1e-3) / 1e9
                p_ = peak if b_ == "mfma" else 8000.0
                by_bound[b_ + "_bound_launches"] = {"launches": n_, "ms": ms_, "achieved": a_, "peak": p_, "unit": "TFLOP/s" if b_ == "mfma" else "GB/s",
                                                    "frac": a_ / p_}
        out = {
            "metric": "images/sec (train step) VOC+Clipart RN50-C4", "value": gb * args.steps / dt, "unit": "images/sec",
            "n_gpus": world, "rccl_ranks": world, "rank_devices": devs,
            "dist_backend": (dist.get_backend() if world > 1 else None), "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "faster_rcnn_voc.sh VOC(labeled)+Clipart(unlabeled) CLIP RN50-C4 + caption consistency "
                                   f"(iter>10000: supervised + image-level + region-level), {args.batch} img/GPU "
                                   f"{args.height}x{args.width}, synthetic pixels + seeded random weights",
                       "global_batch": gb, "parallelism": f"dp{world}",
                       "shared_source_pass": bool(tr.share_source_pass), "fused_consistency_mapper_pass": bool(tr.fuse_consistency)},
            "roofline": {"bound": "mfma", "kernel": dom_name + (" (implicit-GEMM conv / linear, forward + input-gradient)" if args.dtype != "fp8" else
                                                                 " (e4m3 implicit-GEMM conv, forward; v_mfma_scale_f32_32x32x64_f8f6f4)"), "achieved": ach,
                         "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/r03_traffic.json)",
                         "by_bound": by_bound,
                         "algorithmic_bytes_per_launch": dom.get("bytes", 0.0) / max(dom["launches"], 1),
                         "algorithmic_flops_per_launch": dom["flops"] / max(dom["launches"], 1),
                         "launches_per_step": dom["launches"] / prof_steps,
                         "kernel_ms_per_step": dom["ms"] / prof_steps,
                         "measured_on": "HIP events around every launch of this kernel in the last timed step"},
            "kernels_ms_per_step": {k: round(v["ms"], 3) for k, v in full.items()},
            "kernels_ms_note": "one fully instrumented untimed step (event pairs on every launch: that step runs ~5 % slower)",
            "kernels_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in full.items() if v["ms"] > 0 and v["flops"] > 0},
            "losses": losses,
            "losses_gate": gate,
        }
        if fwd_ms is not None:
            ftf = 1.878 * args.batch
            out["forward_roofline"] = {"what": "supervised RN50-C4 forward only (autograd recording on), algorithmic 1.878 TFLOP/image",
                                       "ms": fwd_ms, "achieved": ftf / (fwd_ms * 1e-3), "peak": peak, "unit": "TFLOP/s",
                                       "frac": ftf / (fwd_ms * 1e-3) / peak, "images_per_sec_forward": args.batch / (fwd_ms * 1e-3),
                                       "kernels_ms": {k: round(v["ms"], 3) for k, v in fwd_kernels.items()},
                                       "kernels_ms_note": "one instrumented forward (event pairs on every launch); the ms above is un-instrumented"}
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 8, 64)
            out["cpu_baseline"] = cpu_baseline(args.height, args.width, threads, timed=max(args.cpu_baseline_steps, 1))
        print(json.dumps(out))
        if gate is not None and not gate["ok"]:
            raise SystemExit(f"bench.py: first-step losses {gate['step0_losses']} are not the gated computation's (tests/golden/bench_step0_losses.json:{fkey})")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

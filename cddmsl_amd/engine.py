"""Training loop of the hot path -- detectron2/engine/train_loop.py:261-431 (``SimpleTrainer``), engine/defaults.py:60-79
(DDP), engine/launch.py:27-125 (one process per GPU).

``SimpleTrainer.run_step`` keeps the reference's sequence: supervised forward, caption-consistency forward (x0 during
burn-in, iter <= 10000), region-level forward after burn-in, ``sum(losses).backward()``, gradient all-reduce (mean) over
RCCL/xGMI, per-parameter clip + SGD.  Differences that do not change results: world_size 1 runs without a process
group; the per-iteration ``.item()`` sync of ``_write_metrics`` only happens every ``metrics_period`` steps.
"""
import os
import time

import torch
import torch.distributed as dist

from .modeling.clipcap import TransformerMapper
from .hip import same_layout
from .solver import build_optimizer


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_distributed(backend=None):
    """engine/launch.py:98-123 equivalent under torchrun (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.is_available():
            torch.cuda.set_device(0 if os.environ.get("CDDMSL_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0")))
        # (RCCL needs one device per rank: the shared-GPU rehearsal of the N > 1 path runs over gloo)
        backend = backend or os.environ.get("CDDMSL_DIST_BACKEND") or (
            "nccl" if torch.cuda.is_available() and not os.environ.get("CDDMSL_SHARE_GPU") else "gloo")
        opts = None
        if backend == "nccl" and os.environ.get("CDDMSL_RCCL_HIGH_PRIORITY") == "1":
            # opt-in A/B knob for the first multi-GPU runs: RCCL's kernels on a high-priority stream, so that they get the CUs the
            # compute kernels release first (the persistent conv kernels hold every CU for their whole launch: DESIGN.md section 5)
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group(backend, pg_options=opts)
    return get_rank(), get_world_size()


class GradBuckets:
    """DDP-style gradient averaging (engine/defaults.py:72-74): all trainable grads live in one flat f32 buffer (each ``p.grad``
    is a view with the parameter's memory layout), all-reduced in a few large RCCL calls -- xGMI rings are per-link bound, so
    few big messages beat DDP's 25 MB default buckets.

    Overlap with backward (what the reference's DDP reducer does): the weight-gradient kernels ACCUMULATE into the flat buffer,
    and every such launch is announced through ``layers._grad_buf``.  The first step of a given step signature runs the plain
    way and counts the announcements per parameter; later steps count down, and when every parameter overlapping a bucket has
    had its last write enqueued, an event is recorded on the compute stream and the bucket's all-reduce starts on a side stream
    behind it -- while backward continues with the earlier layers.  Buckets still open when backward returns (the backbone's
    first stages, written last) are reduced then.  A write into a bucket that is already on the wire would corrupt the step, so
    it raises (``CDDMSL_OVERLAP_ALLREDUCE=0`` switches the overlap off)."""

    def __init__(self, params, bucket_bytes=64 << 20, compression=None):
        self.params = [p for p in params if p.requires_grad]
        # optional gradient compression on the wire (engine/defaults.py:75-78 registers torch's fp16_compress_hook: cast, divide by
        # the world size, all-reduce, cast back).  "bf16" keeps the f32 exponent range (no loss scaling needed) and halves the
        # 193.6 MB the ring moves; "fp16" is the reference's own choice.  The f32 flat buffer stays the accumulator on each rank.
        compression = compression if compression is not None else (os.environ.get("CDDMSL_GRAD_COMPRESSION") or None)
        assert compression in (None, "bf16", "fp16"), compression
        self.compression = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16}[compression]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        spans = []
        for p in self.params:
            n = p.numel()
            chunk = self.flat[off:off + n]
            if p.dim() == 4:   # channels_last OIHW parameter -> grad with the same strides
                co, ci, kh, kw = p.shape
                g = chunk.view(co, kh, kw, ci).permute(0, 3, 1, 2)
            else:
                g = chunk.view(p.shape)
            assert same_layout(g, p), (g.stride(), p.stride(), p.shape)
            p.grad = g
            spans.append((off, off + n))
            off += n
        per = max(bucket_bytes // 4, 1)
        self.buckets = [self.flat[i:i + per] for i in range(0, total, per)]
        # parameters overlapping each bucket, and the buckets of each parameter
        self._bucket_params = [[] for _ in self.buckets]
        self._param_buckets = {}
        for p, (a, b) in zip(self.params, spans):
            bs = list(range(a // per, (max(b, a + 1) - 1) // per + 1))
            self._param_buckets[id(p)] = bs
            for bi in bs:
                self._bucket_params[bi].append(id(p))
        self.overlap = os.environ.get("CDDMSL_OVERLAP_ALLREDUCE", "1") != "0"
        self._expected = {}           # step signature -> {id(param): number of gradient writes per step}
        self._counting = None         # dict being filled during a counting step
        self._left = None             # per-parameter writes still to come (overlapped step)
        self._open = None             # per-bucket number of parameters not finished yet
        self._launched, self._ready, self._handles = set(), [], []
        self._comm_stream = None
        self._wire = [None] * len(self.buckets)   # compressed copies of the buckets (allocated on first use)
        self.launch_log = []          # (bucket, writes announced so far) per step: when each bucket went on the wire
        self._announced, self._sig = 0, None

    def zero(self):
        self.flat.zero_()

    # ---------------------------------------------------------------- overlap machinery
    def begin_backward(self, signature):
        """Call right before ``backward()``.  world size 1: nothing to do."""
        from . import layers
        self._handles, self._launched, self._ready, self.launch_log, self._announced = [], set(), [], [], 0
        if get_world_size() == 1 or not self.overlap:
            self._left = self._counting = None
            return
        exp = self._expected.get(signature)
        if exp is None:                                   # first step of this kind: count, reduce after backward
            self._counting, self._left = {}, None
            self._sig = signature
        else:
            self._counting = None
            self._left = dict(exp)
            self._open = [sum(1 for q in ps if exp.get(q, 0) > 0) for ps in self._bucket_params]
        layers._TOUCH_HOOK[0] = self._on_touch

    def _on_touch(self, p):
        self._announced += 1
        if self._counting is not None:
            self._counting[id(p)] = self._counting.get(id(p), 0) + 1
            return
        # the launches announced by EARLIER calls are in the queue by now: buckets they completed can go
        self._flush_ready()
        k = id(p)
        # ANY write into a bucket that has left is an error -- also one by a parameter the counting step never saw write
        for bi in self._param_buckets.get(k, ()):
            if bi in self._launched:
                raise RuntimeError("GradBuckets: a gradient was written into a bucket whose all-reduce is already in flight (the "
                                   "step's kernel sequence changed); rerun with CDDMSL_OVERLAP_ALLREDUCE=0")
        if k not in self._left:
            return
        self._left[k] -= 1
        if self._left[k] == 0:
            for bi in self._param_buckets[k]:
                self._open[bi] -= 1
                if self._open[bi] == 0:
                    self._ready.append(bi)

    def _flush_ready(self):
        while self._ready:
            self._launch(self._ready.pop(0))

    def _reduce(self, bi):
        """scale by 1 / world, (compress,) start the all-reduce of one bucket; returns (handle, bucket index)"""
        b = self.buckets[bi]
        b.mul_(1.0 / get_world_size())
        if self.compression is None:
            return dist.all_reduce(b, async_op=True), bi
        if self._wire[bi] is None:
            self._wire[bi] = torch.empty_like(b, dtype=self.compression)
        w = self._wire[bi]
        w.copy_(b)
        return dist.all_reduce(w, async_op=True), bi

    def _launch(self, bi):
        b = self.buckets[bi]
        self._launched.add(bi)
        self.launch_log.append((bi, self._announced))
        if b.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=b.device)
            ev = torch.cuda.Event()
            ev.record()                                   # behind every launch enqueued so far on the compute stream
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                self._handles.append(self._reduce(bi))
        else:
            self._handles.append(self._reduce(bi))

    def all_reduce_mean(self):
        """After ``backward()`` returned: start what is not on the wire yet, wait for everything."""
        from . import layers
        layers._TOUCH_HOOK[0] = None
        ws = get_world_size()
        if ws == 1:
            return
        if self._counting is not None:
            self._expected[self._sig] = self._counting
            self._counting = None
        if self._left is not None and any(v != 0 for v in self._left.values()):
            late = [bi for bi in self._launched if any(self._left.get(q, 0) != 0 for q in self._bucket_params[bi])]
            if late:
                raise RuntimeError("GradBuckets: fewer gradient writes than expected reached buckets already reduced")
        self._ready = []
        for bi in range(len(self.buckets)):
            if bi not in self._launched:
                self._launch(bi)
        for h, bi in self._handles:
            if self.compression is None or not self.buckets[bi].is_cuda:
                h.wait()                                  # (NCCL: the compute stream waits for the collective's stream)
                if self.compression is not None:
                    self.buckets[bi].copy_(self._wire[bi])
            else:                                         # decompress on the side stream, behind the collective
                with torch.cuda.stream(self._comm_stream):
                    h.wait()
                    self.buckets[bi].copy_(self._wire[bi])
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._handles, self._left, self._open = [], None, None
        # (a plain backward() + all_reduce_mean() without begin_backward() must reduce every bucket again: nothing stays "launched")
        self._launched, self._ready = set(), []


class SimpleTrainer:
    def __init__(self, model, data_loader, optimizer, cfg, clipcap_model=None, metrics_period=20):
        model.train()
        self.model, self.cfg = model, cfg
        if clipcap_model is None:   # train_loop.py:281-288: ClipCaptionModel(40, 40).clip_project, frozen
            clipcap_model = TransformerMapper(1024, 768, 40, 40, 8, compute_dtype=model.compute_dtype)
            path = cfg.MODEL.get("VISION_TO_LANG_PATH", "")
            if path and os.path.exists(path):
                sd = torch.load(path, map_location="cpu", weights_only=True)
                clipcap_model.load_state_dict({k[len("clip_project."):]: v for k, v in sd.items() if k.startswith("clip_project.")})
            clipcap_model.to(model.device)
        self.clipcap_model = clipcap_model.eval()
        for p in self.clipcap_model.parameters():
            p.requires_grad = False
        self.data_loader = data_loader
        self._data_loader_iter = iter(data_loader)
        self.optimizer = optimizer
        self.buckets = GradBuckets(optimizer.params)
        self.iter = 0
        self.burn_in = 10000            # train_loop.py:334
        self.metrics_period = metrics_period
        self.storage = {}
        # run_step below issues every forward of a step before its single backward, on one batch object: the model may
        # evaluate backbone+RPN on the source images once for the supervised and the region-level branch (rcnn.py notes)
        self.share_source_pass = os.environ.get("CDDMSL_SHARE_SOURCE_PASS", "1") != "0"
        # ... and run the two caption-consistency branches' rows through the frozen mapper and the projector together
        self.fuse_consistency = os.environ.get("CDDMSL_FUSE_CONSISTENCY", "1") != "0"

    def compute_losses(self, data):
        """train_loop.py:331-365"""
        kd = self.cfg.MODEL.KD_REGULRAZIATION
        if hasattr(self.model, "share_source_pass"):
            self.model.share_source_pass = self.share_source_pass and self.iter > self.burn_in
        loss_dict = self.model(data)
        loss = {}
        if self.iter > self.burn_in and self.fuse_consistency and hasattr(self.model, "forward_consistency"):
            # the two consistency branches with ONE mapper / projector pass (same per-row results; rcnn.py forward_consistency)
            loss.update(self.model(data, clipcap_model=self.clipcap_model, branch="caption_consistency_both", KD_regularization=kd))
        elif self.iter > self.burn_in:
            loss.update(self.model(data, clipcap_model=self.clipcap_model, branch="caption_consistency", KD_regularization=kd))
            loss["cont_region_loss"] = self.model(data, clipcap_model=self.clipcap_model,
                                                  branch="caption_consistency_regionLevel", KD_regularization=kd)
        else:
            cc = self.model(data, clipcap_model=self.clipcap_model, branch="caption_consistency", KD_regularization=False)
            for k in cc:
                loss[k] = cc[k] * 0.0
        loss_dict.update(loss)
        return loss_dict

    def run_step(self):
        assert self.model.training, "[SimpleTrainer] model was changed to eval mode!"
        start = time.perf_counter()
        data = next(self._data_loader_iter)
        data_time = time.perf_counter() - start
        self.buckets.zero()                       # model.zero_grad() / optimizer.zero_grad()
        loss_dict = self.compute_losses(data)
        losses = sum(loss_dict.values())
        # (the kernel sequence of backward depends on which branches are live and how they are composed, nothing else)
        self.buckets.begin_backward((self.iter > self.burn_in, self.share_source_pass, self.fuse_consistency,
                                     bool(self.cfg.MODEL.KD_REGULRAZIATION), len(data), tuple(sorted(loss_dict))))
        try:
            losses.backward()
        except BaseException:
            from . import layers
            layers._TOUCH_HOOK[0] = None              # a failed backward must not leave the announcement hook installed
            raise
        self.buckets.all_reduce_mean()
        self.optimizer.iteration = self.iter
        self.optimizer.step()
        if self.metrics_period and self.iter % self.metrics_period == 0:
            self._write_metrics(loss_dict, data_time)
        self.iter += 1
        return loss_dict

    def _write_metrics(self, loss_dict, data_time):
        """train_loop.py:391-431: one device sync, NaN/Inf check; cross-rank mean via one small all-reduce."""
        keys = sorted(loss_dict.keys())
        vec = torch.stack([loss_dict[k].detach().float() for k in keys])
        if get_world_size() > 1:
            dist.all_reduce(vec)
            vec = vec / get_world_size()
        vals = vec.cpu().tolist()
        metrics = dict(zip(keys, vals))
        total = sum(vals)
        if not all(map(lambda v: v == v and abs(v) != float("inf"), [total])):
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.iter}!\nloss_dict = {metrics}")
        from . import layers
        nf = layers.FP8_SCALES.nonfinite
        if nf is not None and bool(nf.item()):       # fp8 configuration: e4m3 copies saturate, so an Inf / NaN activation can leave the loss finite
            raise FloatingPointError(f"An activation or gradient quantised to e4m3 held Inf / NaN at or before iteration={self.iter} (loss_dict = {metrics})")
        self.storage.update(metrics)
        self.storage["total_loss"] = total
        self.storage["data_time"] = data_time


class SyntheticPairedLoader:
    """Infinite loader of seeded VOC-shaped paired samples (stand-in for build_detection_train_loader; the real
    VOC+Clipart loader is a 'next' row).  Per-rank batch = IMS_PER_BATCH // world (data/build.py:287)."""

    def __init__(self, per_rank_batch, height=800, width=1333, rank=0, device="cuda", num_classes=20, pool=2):
        from . import synthetic
        from .structures import Boxes, Instances
        self.batches = []
        for it in range(pool):
            b = synthetic.make_batch(per_rank_batch, height, width, rank, it, num_classes=num_classes)
            for x in b:
                x["image"], x["image_trgt"] = x["image"].to(device), x["image_trgt"].to(device)
                i = x["instances"]
                x["instances"] = Instances((height, width), gt_boxes=Boxes(i["gt_boxes"].to(device)), gt_classes=i["gt_classes"].to(device))
            self.batches.append(b)
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return b


def limit_host_threads(n=4):
    """The GPU process only does tiny CPU tensor ops (randperm, index bookkeeping); with the default intra-op pool
    (= all 128-256 host threads) every such op pays an OpenMP team wake-up -- measured 365 vs ~500 ms/step."""
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


def build_trainer(cfg, per_rank_batch, height=800, width=1333, seed=1):
    from .modeling import build_model
    limit_host_threads()
    rank, _ = get_rank(), get_world_size()
    model = build_model(cfg)
    model.proposal_generator.sample_generator.manual_seed(seed + rank)
    model.roi_heads.sample_generator.manual_seed(seed + rank + 7919)
    model.region_generator.manual_seed(seed + rank + 104729)
    opt = build_optimizer(cfg, model)
    loader = SyntheticPairedLoader(per_rank_batch, height, width, rank, cfg.MODEL.DEVICE, cfg.MODEL.ROI_HEADS.NUM_CLASSES)
    return SimpleTrainer(model, loader, opt, cfg)

"""CPU, world_size 2 over gloo: the distributed pieces of the hot path (GatherLayer semantics gather.py:5-20, the DDP-style
flat-buffer gradient mean engine/defaults.py:74, cross-rank contrastive batch rcnn.py:305-317) behave as the reference
intends, and match the oracle's simulated two-rank computation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cddmsl_amd.engine import GradBuckets, get_rank, get_world_size
        from cddmsl_amd.modeling.rcnn import gather_cat
        from oracle import model as om
        assert get_world_size() == 2 and get_rank() == rank
        g = torch.Generator().manual_seed(100)
        a_all = torch.randn(2, 3, 8, generator=g)   # [rank][batch][dim]
        b_all = torch.randn(2, 3, 8, generator=g)
        a = a_all[rank].clone().requires_grad_(True)
        b = b_all[rank].clone().requires_grad_(True)
        loss = om.symmetric_ce(gather_cat(a), gather_cat(b))
        loss.backward()
        # oracle's simulated ranks: other rank's tensors detached, own slot differentiable
        a2 = a_all[rank].clone().requires_grad_(True)
        b2 = b_all[rank].clone().requires_grad_(True)
        sim = om.symmetric_ce(om.gather_cat(a2, [a_all[1 - rank]], rank), om.gather_cat(b2, [b_all[1 - rank]], rank))
        sim.backward()
        assert torch.allclose(loss, sim) and torch.allclose(a.grad, a2.grad, atol=1e-7) and torch.allclose(b.grad, b2.grad, atol=1e-7)
        # flat-buffer gradient mean
        w4 = torch.nn.Parameter(torch.zeros(4, 3, 2, 2).contiguous(memory_format=torch.channels_last))
        w1 = torch.nn.Parameter(torch.zeros(5))
        gb = GradBuckets([w4, w1], bucket_bytes=64)
        w4.grad.fill_(float(rank + 1))
        w1.grad.copy_(torch.arange(5.0) * (rank + 1))
        gb.all_reduce_mean()
        assert torch.allclose(w4.grad, torch.full_like(w4.grad, 1.5)) and torch.allclose(w1.grad, torch.arange(5.0) * 1.5)
        assert w4.grad.permute(0, 2, 3, 1).is_contiguous()
        ret[rank] = float(loss)
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert len(ret) == 2 and abs(ret[0] - ret[1]) < 1e-6   # every rank computes the same full-matrix loss


# ---------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N: who starts the ranks (engine/launch.py:27-82 of the reference spawns one process per GPU itself)
def _bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launch_plan():
    b = _bench()
    assert b.launch_plan(1, [], {}, 0) == ("run", None)
    # under a launcher: WORLD_SIZE has to equal --gpus, there is no "world == 1" escape
    assert b.launch_plan(4, [], {"RANK": "2", "WORLD_SIZE": "4"}, 8) == ("run", None)
    with pytest.raises(SystemExit):
        b.launch_plan(8, [], {"RANK": "0", "WORLD_SIZE": "1"}, 8)
    with pytest.raises(SystemExit):
        b.launch_plan(1, [], {"RANK": "0", "WORLD_SIZE": "2"}, 8)
    # no launcher: spawn N ranks; refuse when the box has fewer devices
    mode, cmd = b.launch_plan(8, ["--gpus", "8", "--steps", "5"], {"MASTER_PORT": "29999"}, 8)
    assert mode == "spawn" and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")
    with pytest.raises(SystemExit) as e:
        b.launch_plan(2, [], {}, 1)
    assert "needs 2 visible GPUs" in str(e.value)


def test_bench_gpus2_without_devices_exits_nonzero():
    """`python bench.py --gpus 2` on a box with fewer than 2 devices: non-zero exit and a message, never an n_gpus: 1 line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CDDMSL_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "visible GPUs" in r.stderr


def test_bench_spawns_two_ranks_dry_run():
    """The spawn path end to end (rehearsal: CDDMSL_SHARE_GPU lifts the device-count check, gloo because no GPU here):
    bench.py --gpus 2 --dry-run starts torch.distributed.run itself, both ranks join, rank 0 reports 2 ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(CDDMSL_SHARE_GPU="1", MASTER_PORT=str(_free_port()), CDDMSL_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and sorted(x[0] for x in out["ranks"]) == [0, 1]


# ---------------------------------------------------------------------------------------------------------------------
# --eval-only over several ranks: InferenceSampler-style sharding + gather of the predictions (evaluation/evaluator.py:103,
# pascal_voc_evaluation.py:79): a 2-rank evaluation gives the 1-rank result
class _ToyDetector(torch.nn.Module):
    def forward(self, batched_inputs):
        from cddmsl_amd.structures import Boxes, Instances
        out = []
        for x in batched_inputs:
            h, w = x["image"].shape[-2:]
            i = int(x["image_id"])                      # _make_voc: object k of image i = class (i + k) % 20 at (5+7k, 3+5k, 40+9k, 50+4k)
            boxes = torch.tensor([[4.0, 2.0, 40.0, 50.0], [1.0, 1.0, w / 2.0, h / 2.0]])
            out.append({"instances": Instances((h, w), pred_boxes=Boxes(boxes), scores=torch.tensor([0.9 - 0.01 * i, 0.3 + 0.02 * i]),
                                               pred_classes=torch.tensor([i % 20, (i + 3) % 20]))})
        return out


def _eval_worker(rank, world, port, root, ret):
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import argparse
        from cddmsl_amd import evaluation
        from cddmsl_amd.config import get_cfg
        cfg = get_cfg()
        cfg.merge_from_list(["MODEL.DEVICE", "cpu", "INPUT.MIN_SIZE_TEST", 0])
        args = argparse.Namespace(voc_root=root, voc_split="test", voc_year=2007)
        res = evaluation.run_eval_only(_ToyDetector(), cfg, args, rank, world, return_results=True)
        if rank == 0:
            ret["w%d" % world] = dict(res["bbox"])
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_eval_only_two_ranks_equals_one_rank(tmp_path):
    from test_data_pipeline import _make_voc
    base = _make_voc(str(tmp_path))
    mgr = mp.Manager()
    ret = mgr.dict()
    _eval_worker(0, 1, 0, base, ret)
    mp.spawn(_eval_worker, args=(2, _free_port(), base, ret), nprocs=2, join=True)
    one, two = ret["w1"], ret["w2"]
    assert set(one) == set(two)
    for k in one:
        assert (one[k] != one[k] and two[k] != two[k]) or abs(one[k] - two[k]) < 1e-12, (k, one[k], two[k])
    assert any(v == v and v > 0 for v in one.values()), one


# ---------------------------------------------------------------------------------------------------------------------
# gradient all-reduce overlapped with backward (engine/defaults.py:72-74: the DDP reducer fires a bucket when its last gradient
# is ready): same averaged gradients as the plain post-backward reduction, buckets on the wire before backward has finished
def _overlap_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cddmsl_amd import layers
        from cddmsl_amd.engine import GradBuckets
        g = torch.Generator().manual_seed(7)
        shapes = [(8, 4, 3, 3), (16,), (12, 8), (6, 4, 1, 1), (40,), (5, 5), (3,)]   # (the last one never writes in a "sig" step)
        params = [torch.nn.Parameter(torch.zeros(*s).contiguous(memory_format=torch.channels_last) if len(s) == 4 else torch.zeros(*s)) for s in shapes]
        gb = GradBuckets(params, bucket_bytes=4 * 100)           # 100 floats per bucket: several parameters per bucket, some straddling
        assert len(gb.buckets) >= 4
        # one "backward": parameters written in reverse registration order, two of them twice (a shared backbone)
        order = [5, 4, 3, 2, 3, 1, 0, 1]
        contrib = {r: [torch.randn(*shapes[i], generator=g) for i in order] for r in range(world)}   # both ranks draw the same stream

        def backward():
            for i, c in zip(order, contrib[rank]):
                layers._grad_buf(params[i]).add_(c)

        want = [torch.zeros(*s) for s in shapes]
        for r in range(world):
            for i, c in zip(order, contrib[r]):
                want[i] += c / world
        logs = []
        for step in range(3):
            gb.zero()
            gb.begin_backward(("sig",))
            backward()
            gb.all_reduce_mean()
            for p, w in zip(params, want):
                assert torch.allclose(p.grad, w, atol=1e-6), step
            logs.append(list(gb.launch_log))
        total = len(order)
        assert all(a == total for _, a in logs[0])                                   # counting step: everything after backward
        assert any(a < total for _, a in logs[1]) and logs[1] == logs[2]             # overlapped: buckets leave while writes remain
        assert sorted(b for b, _ in logs[1]) == list(range(len(gb.buckets)))         # every bucket exactly once
        # another step signature is counted afresh; a changed kernel sequence under a known signature fails loudly
        gb.zero()
        gb.begin_backward(("other",))
        backward()
        layers._grad_buf(params[5]).add_(1.0)
        gb.all_reduce_mean()
        gb.zero()
        gb.begin_backward(("sig",))
        backward()
        try:
            layers._grad_buf(params[5]).add_(1.0)                                    # params[5]'s bucket left long ago
            layers._grad_buf(params[5]).add_(1.0)
            raised = False
        except RuntimeError:
            raised = True
        layers._TOUCH_HOOK[0] = None
        for h, _ in gb._handles:
            h.wait()
        assert raised
        # ... and so does a write by a parameter the counting step never saw write, once its bucket has left (the announcement
        # is checked against the buckets in flight BEFORE the per-parameter countdown is consulted)
        gb.all_reduce_mean()
        gb.zero()
        gb.begin_backward(("sig",))
        backward()
        try:
            layers._grad_buf(params[6]).add_(1.0)                                    # shares the last bucket with params[5]
            layers._grad_buf(params[6]).add_(1.0)
            raised = False
        except RuntimeError:
            raised = True
        layers._TOUCH_HOOK[0] = None
        for h, _ in gb._handles:
            h.wait()
        assert raised
        gb.all_reduce_mean()
        # a plain reduction with no begin_backward() after an overlapped step reduces EVERY bucket again (nothing stays "launched")
        for p in params:
            p.grad.fill_(float(rank + 1))
        gb.all_reduce_mean()
        for p in params:
            assert torch.allclose(p.grad, torch.full_like(p.grad, 1.5))
        # gradient compression on the wire (engine/defaults.py:75-78: the reference's optional fp16 hook; bf16 here): the mean of
        # the two ranks' gradients to bf16 precision, overlapped exactly like the f32 buckets
        mag = [torch.zeros(*s) for s in shapes]                                      # (rounding errors scale with sum |contribution|)
        for r in range(world):
            for i, c in zip(order, contrib[r]):
                mag[i] += c.abs() / world
        for comp, tol in (("bf16", 2.0 ** -8), ("fp16", 2.0 ** -11)):
            gc = GradBuckets(params, bucket_bytes=4 * 100, compression=comp)
            for step in range(2):
                gc.zero()
                gc.begin_backward(("sig",))
                backward()
                gc.all_reduce_mean()
                for p, w, m in zip(params, want, mag):
                    assert p.grad.dtype == torch.float32
                    assert bool(((p.grad - w).abs() <= 3 * tol * m + 1e-6).all()), (comp, step)
                assert any(not torch.equal(p.grad, w) for p, w in zip(params, want))   # (it did go through the narrow type)
            assert any(a < total for _, a in gc.launch_log)
        ret[rank] = logs[1]
    finally:
        dist.destroy_process_group()


def test_allreduce_overlaps_backward_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert len(ret) == 2 and ret[0] == ret[1]

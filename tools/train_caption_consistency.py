#!/usr/bin/env python
"""Entry point with the reference's CLI (tools/train_caption_consistency.py:134-179, engine/defaults.py:82-141):

    python tools/train_caption_consistency.py --num-gpus 8 --config-file configs/VOC-Experiments/faster_rcnn_CLIP_R_50_C4.yaml \\
        MODEL.CLIP.TEXT_EMB_PATH voc_20_cls_emb.pth SOLVER.IMS_PER_BATCH 128

One process per GPU (engine/launch.py:67-80): with --num-gpus > 1 this script re-launches itself under
``torch.distributed.run`` (RCCL over xGMI).  Datasets are not available offline, so the loader is the seeded synthetic
paired-batch generator (``--synthetic`` is implied); real VOC+domain-twin loading is a "next" row.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def default_argument_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--config-file", default="", metavar="FILE")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:29511")
    p.add_argument("--max-iter", type=int, default=None, help="stop after this many iterations (default: SOLVER.MAX_ITER)")
    p.add_argument("--height", type=int, default=800)
    p.add_argument("--width", type=int, default=1333)
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


def setup(args):
    from cddmsl_amd.config import get_cfg
    cfg = get_cfg()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts or [])
    return cfg


def main(args):
    import torch
    from cddmsl_amd import engine, synthetic
    rank, world = engine.init_distributed()
    cfg = setup(args)
    assert not args.eval_only, "evaluation is a 'next' row (SURVEY.md 8(f))"
    cfg.MODEL.DEVICE = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    torch.cuda.set_device(cfg.MODEL.DEVICE)
    per_rank = max(cfg.SOLVER.IMS_PER_BATCH // world, 1)       # data/build.py:287
    tr = engine.build_trainer(cfg, per_rank, args.height, args.width, seed=cfg.SEED)
    if not cfg.MODEL.WEIGHTS:
        tr.model.load_state_dict(synthetic.make_state_dict(0, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES), strict=False)
        tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
    else:
        tr.model.load_state_dict(torch.load(cfg.MODEL.WEIGHTS, map_location="cpu", weights_only=True)["model"], strict=False)
    max_iter = args.max_iter or cfg.SOLVER.MAX_ITER
    for it in range(max_iter):
        tr.run_step()
        if rank == 0 and tr.metrics_period and it % tr.metrics_period == 0:
            print(f"iter {it}  " + "  ".join(f"{k}: {v:.4f}" for k, v in sorted(tr.storage.items())), flush=True)


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    if args.num_gpus > 1 and "RANK" not in os.environ:
        port = args.dist_url.rsplit(":", 1)[-1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes", str(args.num_machines), "--node-rank", str(args.machine_rank),
               "--nproc-per-node", str(args.num_gpus), "--master-addr", "127.0.0.1", "--master-port", port] + sys.argv
        sys.exit(subprocess.call(cmd))
    main(args)

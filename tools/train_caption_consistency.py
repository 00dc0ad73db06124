#!/usr/bin/env python
"""Entry point with the reference's CLI (tools/train_caption_consistency.py:134-179, engine/defaults.py:82-141):

    python tools/train_caption_consistency.py --num-gpus 8 --config-file configs/VOC-Experiments/faster_rcnn_CLIP_R_50_C4.yaml \\
        MODEL.CLIP.TEXT_EMB_PATH voc_20_cls_emb.pth SOLVER.IMS_PER_BATCH 128

One process per GPU (engine/launch.py:67-80): with --num-gpus > 1 this script re-launches itself under
``torch.distributed.run`` (RCCL over xGMI).  Datasets are not available offline, so by default the loader is the seeded synthetic
paired-batch generator; ``--voc-root <VOCdevkit/VOC2007> --dt-data <twin dir>`` switches to the real paired VOC pipeline
(cddmsl_amd/data.py), ``--eval-only --voc-root ...`` runs inference + Pascal VOC AP (cddmsl_amd/evaluation.py), and
``MODEL.WEIGHTS`` / ``--resume`` / ``MODEL.PRE_TRAINED_RCLIP_PATH`` go through cddmsl_amd/checkpoint.py.
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
    p.add_argument("--voc-root", default="", help="VOC devkit year directory (Annotations/, ImageSets/Main/, JPEGImages/) for --eval-only")
    p.add_argument("--voc-split", default="test")
    p.add_argument("--dt-data", default="", help="directory name of the domain-translated twins next to the VOC root (e.g. clipart); "
                                                 "with --voc-root, train on the real paired loader instead of synthetic batches")
    p.add_argument("--voc-year", type=int, default=2007)
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
    from cddmsl_amd.config import auto_scale_workers
    cfg = auto_scale_workers(setup(args), world)                              # engine/defaults.py:374
    cfg.MODEL.DEVICE = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    torch.cuda.set_device(cfg.MODEL.DEVICE)
    per_rank = max(cfg.SOLVER.IMS_PER_BATCH // world, 1)       # data/build.py:287
    tr = engine.build_trainer(cfg, per_rank, args.height, args.width, seed=cfg.SEED)
    from cddmsl_amd.checkpoint import DetectionCheckpointer
    ckpt = DetectionCheckpointer(tr.model, cfg.OUTPUT_DIR, optimizer=tr.optimizer, trainer=tr)
    if not cfg.MODEL.WEIGHTS:       # no checkpoints offline: seeded synthetic weights
        tr.model.load_state_dict(synthetic.make_state_dict(0, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES), strict=False)
        tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
    else:                           # defaults.py:396-413 resume_or_load
        inc = ckpt.resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
        if rank == 0:
            print(f"checkpoint: {len(inc.missing_keys)} missing, {len(inc.unexpected_keys)} unexpected, {len(inc.incorrect_shapes)} shape-skipped")
    rclip = cfg.MODEL.get("PRE_TRAINED_RCLIP_PATH", "")
    if rclip and os.path.exists(rclip):
        ckpt.load_offline_backbone(rclip)                                     # train_loop.py:150-161
    elif rclip and rank == 0:
        print(f"MODEL.PRE_TRAINED_RCLIP_PATH {rclip} not found: the offline (teacher) backbone keeps its loaded / synthetic weights")
    if args.eval_only:              # train_caption_consistency.py:143-152: model.eval(); inference over the test sets
        from cddmsl_amd import evaluation
        assert args.voc_root, "--eval-only needs --voc-root (a VOC devkit year directory) : datasets are not shipped"
        raise SystemExit(evaluation.run_eval_only(tr.model, cfg, args, rank, world))
    if args.voc_root and args.dt_data:      # real paired data; default: seeded synthetic batches
        from cddmsl_amd import data
        from cddmsl_amd.evaluation import VOC_CLASS_NAMES
        dicts = data.load_voc_instances(args.voc_root, "trainval", VOC_CLASS_NAMES[: cfg.MODEL.ROI_HEADS.NUM_CLASSES], dt_data=args.dt_data)
        tr.data_loader = data.build_detection_train_loader(cfg, dicts, per_rank, rank, world, cfg.MODEL.DEVICE)
        tr._data_loader_iter = iter(tr.data_loader)
    max_iter = args.max_iter or cfg.SOLVER.MAX_ITER
    period = cfg.SOLVER.get("CHECKPOINT_PERIOD", 0)
    for it in range(tr.iter, max_iter):
        tr.run_step()
        if rank == 0 and tr.metrics_period and it % tr.metrics_period == 0:
            print(f"iter {it}  " + "  ".join(f"{k}: {v:.4f}" for k, v in sorted(tr.storage.items())), flush=True)
        if rank == 0 and period and (it + 1) % period == 0:                   # hooks.PeriodicCheckpointer
            ckpt.save(f"model_{it:07d}", iteration=it)
    if rank == 0 and period:
        ckpt.save("model_final", iteration=max_iter - 1)


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    if args.num_gpus > 1 and "RANK" not in os.environ:
        port = args.dist_url.rsplit(":", 1)[-1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes", str(args.num_machines), "--node-rank", str(args.machine_rank),
               "--nproc-per-node", str(args.num_gpus), "--master-addr", "127.0.0.1", "--master-port", port] + sys.argv
        sys.exit(subprocess.call(cmd))
    main(args)

#!/usr/bin/env python
"""Loss trajectories of the fp8 configuration (BASELINE.json configs[4]) against bf16 on the same data, weights and random
streams: N optimizer steps each (default 20), per-step losses and the relative deviation of the total.  The two runs share every
seed, but the index stages (NMS, sampling) react to last-bit differences, so part of the deviation is different sampled
RoIs, not arithmetic.  usage: python tools/fp8_vs_bf16.py [--steps 20] [--batch 8]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def run(dtype, steps, batch, h, w):
    import bench
    from cddmsl_amd import engine, synthetic
    cfg = bench.make_cfg(dtype)
    cfg.MODEL.DEVICE = "cuda:0"
    tr = engine.build_trainer(cfg, batch, h, w)
    tr.model.load_state_dict(synthetic.make_state_dict(0), strict=False)
    tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
    tr.iter, tr.metrics_period = 20000, 0
    out = []
    for _ in range(steps):
        ld = tr.run_step()
        out.append({k: float(v.detach()) for k, v in ld.items()})
    torch.cuda.synchronize()
    del tr
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    a = ap.parse_args()
    ref = run("bf16", a.steps, a.batch, a.height, a.width)
    f8 = run("fp8", a.steps, a.batch, a.height, a.width)
    keys = sorted(ref[0])
    print(f"# {a.steps} optimizer steps, {a.batch} x {a.height}x{a.width}, same seeds / data / initial weights; columns: step, total bf16, total fp8, rel. deviation, then per-loss (bf16 | fp8)")
    worst = 0.0
    for i, (r, f) in enumerate(zip(ref, f8)):
        tr_, tf_ = sum(r.values()), sum(f.values())
        dev = abs(tf_ - tr_) / abs(tr_)
        worst = max(worst, dev)
        print(f"{i:3d}  {tr_:9.5f} {tf_:9.5f}  {dev:8.5f}   " + "  ".join(f"{k}: {r[k]:.4f}|{f[k]:.4f}" for k in keys))
    mean_r = sum(sum(r.values()) for r in ref) / len(ref)
    mean_f = sum(sum(f.values()) for f in f8) / len(f8)
    print(f"# worst per-step deviation of the total loss {worst:.4f}; mean total loss bf16 {mean_r:.5f} fp8 {mean_f:.5f} (rel. {abs(mean_f - mean_r) / mean_r:.4f})")


if __name__ == "__main__":
    main()

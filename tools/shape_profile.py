#!/usr/bin/env python
"""Per-shape table of the GEMM kernels inside ONE training step of the bench workload (HIP events around every launch of one
instrumented step): launches, ms, algorithmic TFLOP/s and TB/s, and each shape's own bound -- the larger of its MFMA time
(2*M*N*K at the dense bf16 peak) and its HBM time (every operand once at the HBM peak), both from MI355X_MICROARCH.md -- with
the fraction of that bound it reaches.  usage: python tools/shape_profile.py [batch] [--dtype bf16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

MFMA_PEAK = {"bf16": 2.5e15, "fp8": 5.0e15, "f32": 157.3e12}       # dense, FLOP/s (MI355X_MICROARCH.md)
HBM_PEAK = 8.0e12                                                 # B/s spec (6.29e12 measured for a float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("batch", nargs="?", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rows", type=int, default=110)
    args = ap.parse_args()
    import bench
    from cddmsl_amd import engine, hip, synthetic
    cfg = bench.make_cfg(args.dtype)
    cfg.MODEL.DEVICE = "cuda:0"
    tr = engine.build_trainer(cfg, args.batch, 800, 1333)
    tr.model.load_state_dict(synthetic.make_state_dict(0), strict=False)
    tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
    tr.iter, tr.metrics_period = 20000, 0
    for _ in range(3):
        tr.run_step()
    torch.cuda.synchronize()
    hip.PROFILE.enable()
    tr.run_step()
    bs = hip.PROFILE.by_shape()
    tot = hip.PROFILE.collect()
    peak = MFMA_PEAK[args.dtype]
    rows = sorted(bs.items(), key=lambda kv: -kv[1][1])
    print(f"# one instrumented step, {args.batch} x 800x1333 {args.dtype}; bound = max(2MNK / {peak / 1e12:.0f} TFLOP/s, bytes / {HBM_PEAK / 1e12:.0f} TB/s)")
    print(f"{'kernel':18s} {'shape (M,N,K,KH,pool,stride)':40s} {'n':>4s} {'ms':>8s} {'TFLOP/s':>8s} {'TB/s':>6s} {'mfma_ms':>8s} {'hbm_ms':>7s} {'bound':>5s} {'frac':>5s}")
    agg = {}
    for (name, shape), (n, ms, fl, by) in rows[: args.rows]:
        t_m, t_h = fl / peak * 1e3, by / HBM_PEAK * 1e3
        bound = "mfma" if t_m >= t_h else "hbm"
        frac = max(t_m, t_h) / ms if ms > 0 else 0.0
        print(f"{name:18s} {str(shape):40s} {n:4d} {ms:8.3f} {fl / ms / 1e9:8.1f} {by / ms / 1e9:6.2f} {t_m:8.3f} {t_h:7.3f} {bound:>5s} {frac:5.2f}")
    for (name, shape), (n, ms, fl, by) in bs.items():
        t_m, t_h = fl / peak * 1e3, by / HBM_PEAK * 1e3
        a = agg.setdefault((name, "mfma" if t_m >= t_h else "hbm"), [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += n; a[1] += ms; a[2] += fl; a[3] += by; a[4] += max(t_m, t_h)
    print("# per kernel, launches split by their bound: launches, ms, TFLOP/s, TB/s, fraction of the bound")
    for (name, bound), (n, ms, fl, by, tb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"#   {name:18s} {bound:4s}-bound {n:4d} launches {ms:8.3f} ms {fl / ms / 1e9:8.1f} TFLOP/s {by / ms / 1e9:6.2f} TB/s  frac {tb / ms:4.2f}")
    print("# all library kernels, ms per step:", {k: round(v["ms"], 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["ms"])})


if __name__ == "__main__":
    main()

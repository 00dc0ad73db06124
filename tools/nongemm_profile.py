#!/usr/bin/env python
"""The non-GEMM kernels of one training step against the HBM roofline: ms per step (HIP events around every launch of one
instrumented step of the bench workload), algorithmic bytes (every operand once, from the workload's tensor shapes -- listed per
row), GB/s and the fraction of the 8 TB/s peak (MI355X_MICROARCH.md) -- next to what plain streaming kernels sustain on the
same silicon (tools/hbm_probe.hip: read 6.1-6.4, write 4.1-5.7, copy 4.6-5.8 TB/s).

usage (GPU box): python tools/nongemm_profile.py > profiles/r03_nongemm.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

HBM = 8.0e12


def main():
    import bench
    from cddmsl_amd import engine, hip, synthetic
    B, H, W = 16, 800, 1333
    cfg = bench.make_cfg("bf16")
    cfg.MODEL.DEVICE = "cuda:0"
    tr = engine.build_trainer(cfg, B, H, W)
    tr.model.load_state_dict(synthetic.make_state_dict(0), strict=False)
    tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
    tr.iter, tr.metrics_period = 20000, 0
    for _ in range(3):
        tr.run_step()
    torch.cuda.synchronize()
    hip.PROFILE.enable()
    tr.run_step()
    tot = hip.PROFILE.collect()
    # ---- the workload's tensor sizes (bytes; bf16 unless noted)
    hf, wf = 50, 83                                        # res4 of 800 x 1333
    K = B * 512                                            # sampled RoIs of the supervised pass
    Kr = B * 16 * 2                                        # region-level crops (source + target)
    E = 2 * B                                              # 224 x 224 image-level maps riding behind the region crops
    Kp = (K, Kr + E)                                       # the two RoI-head passes
    C4, P = 2048, 49
    rows = 544 * 80                                        # mapper rows: (2B image + 2 * 16B region) sequences x 80 tokens
    res4 = B * hf * wf * 1024 * 2
    est = {}

    def add(name, nbytes, what):
        est[name] = (float(nbytes), what)

    # RoIAlign forward (commuted head): conv1 map (512 ch) -> o1 crops; res4 (1024 ch) -> pooled crops; per pass
    add("roi_align_forward", sum(k * 196 * 512 * 2 + k * 49 * 1024 * 2 for k in (K, Kr)) + 2 * (B * hf * wf * 512 * 2 + res4) + res4,
        "o1 crops [K,14,14,512] + pooled crops [K,7,7,1024] written, the two maps read once")
    add("roi_align_backward", sum(k * 196 * 512 * 2 + k * 49 * 1024 * 2 for k in (K, Kr)) + 2 * (B * hf * wf * 512 * 2) + 3 * res4,
        "crop gradients read, map gradients written")
    tokb = sum(k * (P * C4 * 2 + 56 * C4 * 2 + C4 * 8) for k in Kp)
    add("attn_tokens_fwd", tokb, "map [K,49,2048] read, tokens [K,56,2048] + mask words [K,2048] x 8 B written")
    add("attnpool_dx", sum(k * (64 * C4 * 2 + P * C4 * 2 + C4 * 12 + 64 * 56 * 2) for k in Kp), "[dZ;U] read, dx written, g0 + mask words read")
    big = sum(k * 32 * C4 * 2 for k in Kp)                 # one [K,32,2048] tensor
    tok = sum(k * 56 * C4 * 2 for k in Kp)
    add("gemm_tn_batched", 2 * (tok + big) + 2 * big * 2, "Z = P.tok, dU = dS.tok (tokens read, [K,32,2048] written), dWv / dWk reductions ([K,32,2048] read)")
    add("gemm_nt_batched", 3 * big + 2 * (big + tok) + 3 * big, "U, dZ written; S = U.tok^T, dP = dZ.tok^T (both operands read); o, dq0 ([K,32,2048] read)")
    # avgpool: stem, layer2.0 / layer3.0 (o2 and x), RoI head o2 -> p2; two backbone passes (source, target) + the 224 branch (small)
    bb = 2 * B
    apf = bb * (400 * 667 * 64) * 2 * 1.25 + bb * (200 * 333 * (128 + 256)) * 2 * 1.25 + bb * (100 * 166 * (256 + 512)) * 2 * 1.25 + sum(k * 196 * 512 * 2 * 1.25 for k in Kp)
    add("avgpool2_fwd", apf, "stem, layer2.0, layer3.0 (conv2 output and block input), RoI head conv2 output: read + quarter-size write")
    apb = B * 2 * (200 * 333 * 128 + 100 * 166 * 256) * 2 * 2.25 + sum(k * 196 * 512 * 2 * 2.25 for k in Kp)
    add("avgpool2_bwd", apb, "pooled gradient read, ReLU mask read, full-size gradient written")
    add("layernorm", 16 * rows * 768 * (4 + 2) + 16 * rows * 768 * (2 + 4 + 4 + 4), "16 forward (f32 in, bf16 out) + 16 backward (bf16 dy, f32 x, f32 accumulate read + write) over [43520,768]")
    add("relu_bwd", 0.0, "")
    table = []
    for name, v in sorted(tot.items(), key=lambda kv: -kv[1]["ms"]):
        if name.startswith("k_conv") or name.startswith("k_wgrad"):
            continue
        b, what = est.get(name, (0.0, ""))
        gbs = b / (v["ms"] * 1e-3) / 1e9 if b else None
        table.append((name, v["launches"], v["ms"], b, gbs, what))
    print(f"# one instrumented step, {B} x {H}x{W} bf16: kernels outside the GEMMs; bytes = every operand once; peak {HBM / 1e12:.0f} TB/s")
    print(f"{'kernel':22s} {'n':>4s} {'ms':>7s} {'GB algorithmic':>15s} {'GB/s':>8s} {'of peak':>8s}  operands")
    s = 0.0
    for name, n, ms, b, gbs, what in table:
        s += ms
        print(f"{name:22s} {n:4d} {ms:7.3f} {b / 1e9:15.2f} {'' if gbs is None else '%8.0f' % gbs} {'' if gbs is None else '%8.2f' % (gbs * 1e9 / HBM)}  {what}")
    print(f"# total outside the GEMMs: {s:.2f} ms of the step; GEMM kernels: {sum(v['ms'] for k, v in tot.items() if k.startswith('k_conv') or k.startswith('k_wgrad')):.2f} ms")


if __name__ == "__main__":
    main()

"""Where a 256x256 tile of k_conv_fwd256 spends its time: per wave, s_memrealtime stamps (the constant 100 MHz reference) at kernel entry,
main-loop start, main-loop end and exit (after the wave's stores have left), from a DIAGNOSTIC build of the library (-DCDDMSL_TILE_STAMPS on gemm_conv.hip;
the shipped library carries no stamps).  Reports, per shape, the medians over all waves of prologue / main loop / epilogue in
microseconds and the launch's event time.

  build:  for f in cddmsl_amd/csrc/*.hip: hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off [-DCDDMSL_TILE_STAMPS for gemm_conv] -c ...
          hipcc -shared -fPIC *.o -o scratch/libstamps.so
  run:    python tools/tile_stamps.py scratch/libstamps.so
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
from cddmsl_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from cddmsl_amd import hip  # noqa: E402

dev = "cuda"
SHAPES = [  # (Nimg, H, W, Cin, Cout, K, pad, residual, relu_mask)
    (8192, 7, 7, 512, 2048, 1, 0, True, False),
    (8192, 7, 7, 512, 2048, 1, 0, False, False),
    (8192, 7, 7, 2048, 512, 1, 0, False, False),
    (8192, 7, 7, 2048, 512, 1, 0, False, True),
    (8192, 7, 7, 512, 512, 3, 1, False, False),
    (16, 200, 333, 64, 256, 1, 0, False, False),
    (16, 100, 166, 128, 512, 1, 0, True, False),
    (16, 50, 83, 256, 1024, 1, 0, True, False),
]
L = _lib.lib()
L.cddmsl_debug_tile_stamps.argtypes = [ctypes.c_void_p]
L.cddmsl_debug_tile_stamps.restype = None
g = torch.Generator(device=dev).manual_seed(0)
print("shape (M,N,K)  res mask kernel | launch ms | per tile slot us | prologue  main  epilogue  total (median us per wave) | tiles per CU")
for (N, H, W, Cin, Cout, K, p, res, msk) in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).bfloat16()
    w = (torch.randn(Cout, K, K, Cin, device=dev, generator=g) * (Cin * K * K) ** -0.5).bfloat16()
    r = torch.randn(N, H, W, Cout, device=dev, generator=g).bfloat16() if res else None
    m = torch.randn(N, H, W, Cout, device=dev, generator=g).bfloat16() if msk else None
    sc = torch.ones(Cout, device=dev)
    bi = torch.zeros(Cout, device=dev)
    M = N * H * W
    tiles = ((M + 255) // 256) * (Cout // 256)
    os.environ["CDDMSL_GEMM256"] = "2"
    for persist in (("0", "1") if K == 1 else ("0",)):     # the persistent kernel stamps per tile: entry = the tile's first read
        os.environ["CDDMSL_PERSIST"] = persist
        stamps = torch.zeros(tiles * 8 * 4, dtype=torch.int64, device=dev)
        for _ in range(2):
            hip.conv_fwd(x, w, sc, bi, r, relu=not msk, relu_mask=m, stride=1, pad=p)
        L.cddmsl_debug_tile_stamps(ctypes.c_void_p(stamps.data_ptr()))
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.conv_fwd(x, w, sc, bi, r, relu=not msk, relu_mask=m, stride=1, pad=p)
        e1.record()
        torch.cuda.synchronize()
        L.cddmsl_debug_tile_stamps(ctypes.c_void_p(0))
        ms = e0.elapsed_time(e1)
        s = stamps.view(tiles * 8, 4).double().cpu()
        tick_us = 0.01                                # s_memrealtime: the constant 100 MHz reference clock
        pro = ((s[:, 1] - s[:, 0]) * tick_us).median().item()
        main = ((s[:, 2] - s[:, 1]) * tick_us).median().item()
        epi = ((s[:, 3] - s[:, 2]) * tick_us).median().item()
        tot = ((s[:, 3] - s[:, 0]) * tick_us).median().item()
        kind = "persistent" if persist == "1" else "one-tile  "
        print(f"({M},{Cout},{K*K*Cin}) {int(res)} {int(msk)} {kind} | {ms:7.3f} | {ms * 1e3 / (tiles / 256):6.2f} | {pro:6.2f} {main:6.2f} {epi:6.2f} {tot:6.2f} | {tiles / 256:.1f}", flush=True)
    del x, w, r, m

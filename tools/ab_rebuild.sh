#!/bin/bash
# usage (GPU box, from the repo root): [SRC=roi_align] bash tools/ab_rebuild.sh "<extra hipcc flags>" <tag>  -- bench, rebuild one object with the flags, relinks the library, runs the bench
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-roofline > gpurun_out/ab_base1.json 2> gpurun_out/ab.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -ffp-contract=off $1 -I include -c cddmsl_amd/csrc/${SRC:-gemm_conv}.hip -o /tmp/gemm_conv_ab.o
mkdir -p /tmp/objs && cp build/obj/*.o /tmp/objs/ && cp /tmp/gemm_conv_ab.o /tmp/objs/${SRC:-gemm_conv}.o
cp cddmsl_amd/libcddmsl_hip.so /tmp/lib_base.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/objs/*.o -o cddmsl_amd/libcddmsl_hip.so
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-roofline > gpurun_out/ab_var1.json 2>> gpurun_out/ab.err
cp /tmp/lib_base.so cddmsl_amd/libcddmsl_hip.so
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-roofline > gpurun_out/ab_base2.json 2>> gpurun_out/ab.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/objs/*.o -o cddmsl_amd/libcddmsl_hip.so
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-roofline > gpurun_out/ab_var2.json 2>> gpurun_out/ab.err
python - <<PY
import json
for n in ("ab_base1", "ab_var1", "ab_base2", "ab_var2"):
    d = json.loads([l for l in open("gpurun_out/%s.json" % n) if l.startswith("{")][-1])
    k = d["kernels_ms_per_step"]
    print(n, round(d["value"], 2), round(d["ms_per_step"], 2), "fwd256", k.get("k_conv_fwd256"), "roi", k.get("roi_align_forward"), "gate", d.get("losses_gate", {}).get("ok"), "hbm-bound", d["roofline"].get("by_bound", {}).get("hbm_bound_launches"))
PY

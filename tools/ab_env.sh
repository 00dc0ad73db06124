#!/bin/bash
# usage (GPU box, from the repo root): [EXTRA="--dtype fp8"] bash tools/ab_env.sh VAR v1 v2 ...   -- the bench under VAR=v for each v, twice round-robin
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VAR=$1; shift
for rep in 1 2 3; do for v in "$@"; do env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-roofline ${EXTRA} > gpurun_out/env_${v}_$rep.json 2>> gpurun_out/env.err; done; done
python - "$@" <<PY
import json, sys
for v in sys.argv[1:]:
    for rep in (1, 2, 3):
        d = json.loads([l for l in open("gpurun_out/env_%s_%d.json" % (v, rep)) if l.startswith("{")][-1])
        k = d["kernels_ms_per_step"]
        print("$VAR", v, rep, round(d["value"], 2), round(d["ms_per_step"], 2), "fwd256", k.get("k_conv_fwd256"), "avgf", k.get("avgpool2_fwd"), "avgb", k.get("avgpool2_bwd"), "tok", k.get("attn_tokens_fwd"), "relu", k.get("relu_bwd"), "gate", (d.get("losses_gate") or {}).get("ok"))
PY

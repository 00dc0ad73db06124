#!/bin/bash
# Collects the judged profiling artefacts on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command   -> gpurun_out/prof/<tag>_kernel_stats.csv
#   2. HBM traffic PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs; MI355X_MICROARCH.md: TCC has 4 slots,
#      FETCH_SIZE costs 3, WRITE_SIZE 2) and the per-launch average for the dominant kernel, with the gfx950
#      correction FETCH_SIZE x2 for wide coalesced reads -> gpurun_out/prof/<tag>_traffic.json
# usage: bash tools/profile_round.sh r01
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/stats $OUT/fetch $OUT/write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-forward-roofline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-forward-roofline > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-forward-roofline > /dev/null 2> $OUT/write.err
cd $R
python3 - <<PY
import csv, glob, json, collections
def per_launch(tag, counter, kern):
    f = glob.glob('$OUT/%s/*/*counter_collection.csv' % tag)[0]
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name'] and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value']); ids.add(r['Dispatch_Id'])
    return tot / max(len(ids), 1), len(ids)
out = {}
for kern, name in (("k_conv_fwd256", "k_conv_fwd256"), ("k_conv_fwdI", "k_conv_fwd"), ("k_wgrad256", "k_wgrad256"), ("k_conv_wgrad_dma", "k_conv_wgrad_dma")):
    fk, n1 = per_launch("fetch", "FETCH_SIZE", kern)      # KiB per launch
    wk, n2 = per_launch("write", "WRITE_SIZE", kern)
    out[name] = {"launches_sampled": n1, "fetch_size_kib_raw": fk, "write_size_kib": wk,
                 "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                 "note": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read"}
json.dump(out, open('$OUT/${TAG}_traffic.json', 'w'), indent=1)
print(json.dumps(out))
PY
head -c 600 $OUT/${TAG}_bench_under_rocprof.json; echo

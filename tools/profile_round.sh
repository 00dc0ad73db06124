#!/bin/bash
# Collects the judged profiling artefacts on the GPU box (run through gpurun from the repo root), named per round:
#   1. rocprofv3 --kernel-trace --stats of the default bench command            -> <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   2. PMC passes, each in its own run with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots": TCC has 4 slots,
#      FETCH_SIZE costs 3, WRITE_SIZE 2; SQ and GRBM are independent blocks):
#        FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
#      -> <tag>_traffic.json : per GEMM kernel HBM bytes per launch (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), MFMA-busy share of
#         the kernel's SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)), effective clock
#         (GRBM_GUI_ACTIVE / 8 / duration; rocprofv3 sums the 8 XCDs) of its long dispatches
#   3. tools/shape_profile.py                                                   -> <tag>_shape_profile.txt
#   4. tools/clock_probe.hip (in-kernel clock of an MFMA-dense loop)            -> <tag>_clock_probe.txt
#   5. tools/hbm_probe.hip (what plain streaming kernels sustain on this box)   -> <tag>_hbm_probe.txt
# usage: bash tools/profile_round.sh r03
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
B="--steps 2 --warmup 1 --no-cpu-baseline --no-forward-roofline"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-forward-roofline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py $B > /dev/null 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py $B > /dev/null 2> $OUT/write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py $B > /dev/null 2> $OUT/sq.err
echo "sq done"
cd $R
python3 - <<PY
import csv, glob, json, collections
def table(tag):
    f = glob.glob('$OUT/%s/*/*counter_collection.csv' % tag)[0]
    rows = list(csv.DictReader(open(f)))
    return rows
def per_launch(rows, counter, kern):
    tot, ids = 0.0, set()
    for r in rows:
        if kern in r['Kernel_Name'] and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value']); ids.add(r['Dispatch_Id'])
    return tot / max(len(ids), 1), len(ids)
fetch, write, sq = table("fetch"), table("write"), table("sq")
# durations of the sq pass's dispatches (kernel trace of the same run)
kt = glob.glob('$OUT/sq/*/*kernel_trace.csv')[0]
dur = {r['Dispatch_Id']: (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) for r in csv.DictReader(open(kt))}
out = {}
for kern, name in (("k_conv_fwd256", "k_conv_fwd256"), ("k_conv_fwd2I", "k_conv_fwd2"), ("k_conv_fwdI", "k_conv_fwd"), ("k_wgrad256", "k_wgrad256"), ("k_conv_wgrad_dma", "k_conv_wgrad_dma")):
    fk, n1 = per_launch(fetch, "FETCH_SIZE", kern)      # KiB per launch
    wk, n2 = per_launch(write, "WRITE_SIZE", kern)
    per = collections.defaultdict(dict)
    for r in sq:
        if kern in r['Kernel_Name']:
            per[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    mf = sum(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for v in per.values())
    gui = sum(v.get('GRBM_GUI_ACTIVE', 0.0) for v in per.values())
    sqb = sum(v.get('SQ_BUSY_CYCLES', 0.0) for v in per.values())
    long_ = [(v['GRBM_GUI_ACTIVE'] / 8.0) / dur[d] for d, v in per.items() if d in dur and dur[d] > 1.0e6 and 'GRBM_GUI_ACTIVE' in v]
    out[name] = {"launches_sampled": n1, "fetch_size_kib_raw": fk, "write_size_kib": wk,
                 "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                 "mfma_busy_cycles": mf, "grbm_gui_active": gui, "sq_busy_cycles": sqb,
                 "mfma_busy_share_of_simd_cycles": (mf / (1024.0 * gui / 8.0)) if gui else None,
                 "effective_clock_ghz_dispatches_over_1ms": (sum(long_) / len(long_)) if long_ else None, "long_dispatches": len(long_),
                 "note": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; "
                         "MFMA share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / dispatch duration (ns)"}
json.dump(out, open('$OUT/${TAG}_traffic.json', 'w'), indent=1)
print(json.dumps(out))
# per-shape view of the dominant kernel: dispatches grouped by grid size (= number of 256x256 tiles)
def by_grid(rows, counter, mul):
    d = collections.defaultdict(list)
    for r in rows:
        if "k_conv_fwd256" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            d[int(r["Grid_Size"]) // 512].append(mul * float(r["Counter_Value"]) * 1024 / 1e6)
    return d
fe, wr = by_grid(fetch, "FETCH_SIZE", 2.0), by_grid(write, "WRITE_SIZE", 1.0)
with open('$OUT/${TAG}_traffic_by_grid.txt', 'w') as f:
    f.write("# k_conv_fwd256: fabric-side traffic per dispatch (MB; FETCH_SIZE x2, WRITE_SIZE), dispatches grouped by grid size / 512 (= tile count; the\n# persistent form of the short-reduction layers launches one workgroup per CU: all of those appear under 256).\n"
            "# FETCH_SIZE counts the L2's memory-side requests INCLUDING Infinity-Cache hits (MI355X_MICROARCH.md, HBM): it bounds HBM reads from above.\n"
            "# tiles  dispatches  fetch: min / quartiles / max   write: distinct values\n")
    for t in sorted(fe, key=lambda t: -sum(fe[t])):
        a = sorted(fe[t]); w = sorted(set(round(x) for x in wr.get(t, [0.0])))
        q = [a[0], a[len(a) // 4], a[len(a) // 2], a[(3 * len(a)) // 4], a[-1]]
        f.write("%6d %4d   fetch %s   write %s\n" % (t, len(a), " ".join("%7.0f" % x for x in q), w))
PY
python3 tools/shape_profile.py 16 > $OUT/${TAG}_shape_profile.txt 2> $OUT/shape.err
echo "shape profile done"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/clock_probe.hip -o /tmp/clock_probe 2> /dev/null && /tmp/clock_probe > $OUT/${TAG}_clock_probe.txt
cat $OUT/${TAG}_clock_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result tools/hbm_probe.hip -o /tmp/hbm_probe 2> /dev/null && /tmp/hbm_probe > $OUT/${TAG}_hbm_probe.txt
cat $OUT/${TAG}_hbm_probe.txt
head -c 400 $OUT/${TAG}_bench_under_rocprof.json; echo

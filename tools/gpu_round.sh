#!/usr/bin/env bash
# One GPU-box visit: parity tests, A/B tools, bench, rocprofv3 kernel stats and PMC passes.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag> [steps...]
# Every step is wrapped in its own timeout and logs under gpurun_out/<tag>_*.
set -u
TAG=${1:-rX}; shift || true
STEPS=${*:-"pytest splat sweep bench prof pmc"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
export TMPDIR=/tmp
run() { local name=$1 t=$2; shift 2; ( cd "$R" && timeout "$t" "$@" ) > "$O/${TAG}_${name}.log" 2>&1; echo "rc=$?" >> "$O/${TAG}_${name}.log"; tail -n 4 "$O/${TAG}_${name}.log"; }
for s in $STEPS; do
  case $s in
    pytest) run pytest 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s ;;
    splattest) run splattest 900 python -m pytest tests/test_gpu_splat.py -m gpu -q -x --timeout 600 -p no:cacheprovider -s ;;
    splatprof) for v in ${SPLAT_VARIANTS:-"d" "nolds:splat_lds=0" "i1:splat_items=1" "i1nolds:splat_items=1 splat_lds=0"}; do
              name=${v%%:*}; knobs=""; [ "$v" != "$name" ] && knobs=${v#*:}
              ( cd /tmp && SPLAT_PROBE_STATS=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_sp_$name" -o splat -- python "$R/tools/splat_cells_probe.py" 30000000 $knobs ) > "$O/${TAG}_sp_$name.log" 2>&1
              grep "ms/frame" "$O/${TAG}_sp_$name.log" | sed "s/^/$name: /"
            done
            python "$R/tools/splat_kstats.py" "$O"/${TAG}_sp_*/ | tee "$O/${TAG}_splat_kernels.txt"
            run splatstats 300 python tools/splat_cells_probe.py ;;
    trainprof) ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_trainprof" -o train -- python "$R/bench.py" --config train --steps 3 --warmup 2 --no-cpu-baseline ) > "$O/${TAG}_trainprof.log" 2>&1; echo "rc=$?" >> "$O/${TAG}_trainprof.log"; tail -n 2 "$O/${TAG}_trainprof.log" | cut -c1-300
            python - "$O/${TAG}_trainprof" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:25]:
    print("%8.2f ms %6s calls %9.1f us avg  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
            ;;
    streetprof) for v in "st_d" "st_nolds:splat_lds=0" "st_n48:splat_near=48" "st_i1:splat_items=1"; do
              name=${v%%:*}; knobs=""; [ "$v" != "$name" ] && knobs=${v#*:}
              ( cd /tmp && SPLAT_PROBE_STATS=0 SPLAT_PROBE_SCENE=street SPLAT_PROBE_H=368 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_sp_$name" -o splat -- python "$R/tools/splat_cells_probe.py" 10000000 $knobs ) > "$O/${TAG}_sp_$name.log" 2>&1
              grep "ms/frame" "$O/${TAG}_sp_$name.log" | sed "s/^/$name: /"
            done
            python "$R/tools/splat_kstats.py" "$O"/${TAG}_sp_st_*/ | tee "$O/${TAG}_street_kernels.txt"
            ( cd "$R" && SPLAT_PROBE_SCENE=street SPLAT_PROBE_H=368 timeout 200 python tools/splat_cells_probe.py 10000000 ) 2>&1 | tail -14 ;;
    bencht) run bencht 900 python bench.py --config train ;;
    benchk) run benchk 600 python bench.py --config kitti6_like --detail "$O/${TAG}_detailk.json" ;;
    splat)  run splat 400 python tools/splat_modes.py --out "$O/${TAG}_splat_modes.json" ;;
    sweep)  run sweep 400 python tools/sweep_conv.py --out "$O/${TAG}_sweep.json" ;;
    trace)  run trace 300 bash -c "python tools/trace_conv.py --shape L0 --configs 0,10 --out $O/${TAG}_trace; python tools/trace_conv.py --shape L1 --configs 10 --out $O/${TAG}_trace; python tools/trace_conv.py --shape L2 --configs 4 --out $O/${TAG}_trace; python tools/trace_conv.py --shape L3 --configs 11 --out $O/${TAG}_trace" ;;
    benchw) run benchw 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --tune conv_wave=1 --detail "$O/${TAG}_detailw.json" ;;
    pytestw) run pytestw 600 env READ_CONV_WAVE=1 python -m pytest tests/test_gpu_unet.py tests/test_gpu_api.py -m gpu -q --timeout 600 -p no:cacheprovider -s ;;
    stagger) for t in 0 400 800 1600; do run stagger$t 200 python tools/sweep_conv.py --shapes 4 --only wave --tune conv_stagger=$t --iters 10 --out "$O/${TAG}_stagger$t.json"; done ;;
    bench)  run bench 600 python bench.py --detail "$O/${TAG}_detail.json" ;;
    prof)   ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_prof" -o bench -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline ) > "$O/${TAG}_prof.log" 2>&1; echo "rc=$?" >> "$O/${TAG}_prof.log"; tail -n 3 "$O/${TAG}_prof.log"
            find "$O/${TAG}_prof" -name "*kernel_stats*" | head -3 ;;
    pmcconv) ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$O/${TAG}_pmc_mfma" -o conv -- python "$R/tools/sweep_conv.py" --main-only --iters 2 --out "$O/${TAG}_pmc_sweep.json" ) > "$O/${TAG}_pmc_mfma.log" 2>&1; echo "rc=$?" >> "$O/${TAG}_pmc_mfma.log"; tail -n 2 "$O/${TAG}_pmc_mfma.log" ;;
    pmctraffic) for cnt in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d "$O/${TAG}_pmc_$cnt" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ) > "$O/${TAG}_pmc_$cnt.log" 2>&1; echo "rc=$?" >> "$O/${TAG}_pmc_$cnt.log"; done; find "$O" -name "*counter_collection.csv" | head ;;
    pmc)    ( cd /tmp && timeout 120 rocprofv3 -L > "$O/${TAG}_counters.txt" 2>&1 )
            ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/${TAG}_pmc_fetch" -o splat -- python "$R/tools/splat_modes.py" --out "$O/${TAG}_pmc_splat.json" ) > "$O/${TAG}_pmc_fetch.log" 2>&1; echo "rc=$?" >> "$O/${TAG}_pmc_fetch.log"
            ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/${TAG}_pmc_write" -o splat -- python "$R/tools/splat_modes.py" --out "$O/${TAG}_pmc_splat.json" ) > "$O/${TAG}_pmc_write.log" 2>&1; echo "rc=$?" >> "$O/${TAG}_pmc_write.log" ;;
  esac
done
# keep the merge-back small: drop raw traces, keep csv summaries
find "$O" -name "*.db" -size +20M -delete 2>/dev/null
du -sh "$O" | tail -1

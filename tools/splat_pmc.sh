#!/usr/bin/env bash
# SQ instruction / activity counters of the rasteriser kernels over 11 consecutive poses (two rocprofv3 --pmc passes, no
# other tracing).  Usage on the GPU box:  bash tools/splat_pmc.sh <tag>   -> gpurun_out/<tag>_pmc{1,2}/
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  SPLAT_PROBE_STATS=0 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/${TAG}_pmc$i" -o splat -- \
      python "$R/tools/splat_cells_probe.py" 30000000 > "$O/${TAG}_pmc$i.log" 2>&1
  grep "ms/frame" "$O/${TAG}_pmc$i.log"
done

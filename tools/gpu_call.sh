#!/usr/bin/env bash
# One gpurun visit, steps given as "name:timeout:command" lines on stdin; logs under gpurun_out/<tag>_<name>.log
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
export TMPDIR=/tmp
while IFS= read -r line; do
  [ -z "$line" ] && continue
  name=${line%%:*}; rest=${line#*:}; t=${rest%%:*}; cmd=${rest#*:}
  s=$(date +%s)
  ( cd "$R" && timeout "$t" bash -c "$cmd" ) > "$O/${TAG}_${name}.log" 2>&1
  echo "rc=$? secs=$(( $(date +%s) - s ))" >> "$O/${TAG}_${name}.log"
  echo "== $name: $(tail -n 1 "$O/${TAG}_${name}.log")"
done
find "$O" -name "*.db" -size +20M -delete 2>/dev/null
du -sh "$O" | tail -1

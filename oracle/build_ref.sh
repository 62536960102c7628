#!/usr/bin/env bash
# ORACLE — TEST INFRASTRUCTURE ONLY.
# Builds oracle/_ref/libpcpr_ref.so from the reference's OWN kernel source where it lies
# (/root/reference/MyRender/CloudProjection/point_render.cu, lines 1-167 = structs + DepthProject;
# the host launcher below line 167 uses the <<< >>> launch syntax and torch CUDA tensors and is
# replaced by oracle/ref_driver.inc).  The source is streamed through the compiler — no copy of
# it is written anywhere.  g++ on x86-64 without -mfma never contracts a*b+c, and
# -ffp-contract=off makes that explicit.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF=/root/reference/MyRender/CloudProjection
[ -f "$REF/point_render.cu" ] || { echo "reference not present; keeping prebuilt oracle/_ref" >&2; exit 0; }
mkdir -p "$HERE/_ref"
{ sed -n '1,167p' "$REF/point_render.cu"; cat "$HERE/ref_driver.inc"; } |
  g++ -O2 -ffp-contract=off -std=c++17 -w -shared -fPIC -x c++ \
      -I "$HERE/shim" -I "$REF" -o "$HERE/_ref/libpcpr_ref.so" -
echo "$HERE/_ref/libpcpr_ref.so"

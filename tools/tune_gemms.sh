#!/bin/bash
# Regenerate walk-these-ways_amd/tuning/tunableop_gfx950.csv on an MI355X (run from the repo root):
# a long TunableOp pass over the GEMM shapes of bench.py's configuration, then copy the result into gpurun_out/.
set -e
rm -f /tmp/tunableop_gfx9500.csv
GO1_TUNE_ALL=1 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunableop_gfx950.csv \
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=150 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=20 PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=512 \
  python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-traffic
mkdir -p gpurun_out
cp /tmp/tunableop_gfx9500.csv gpurun_out/tunableop_gfx950.csv
wc -l gpurun_out/tunableop_gfx950.csv

#!/bin/bash
# Collects the profile summaries kept under profiles/ (run on the GPU box from the repository root):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command and of its inference leg alone
#   2. three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters) over tools/pmc_target.py
#   3. the per-layer HIP-event tables (preset s, preset l, one bf16 training step)
# usage: tools/collect_profiles.sh TAG   (outputs gpurun_out/profiles/TAG_*)
set -o pipefail
TAG=${1:-rX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
RAW=$ROOT/gpurun_out/prof_raw
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
cd /tmp || exit 1

timeout -k 10 200 python3 "$ROOT/tools/layer_profile.py" 64 bf16 > "$OUT/${TAG}_layers_bf16_batch64.txt" 2>/dev/null || exit 1
timeout -k 10 200 python3 "$ROOT/tools/layer_profile.py" 64 bf16 l > "$OUT/${TAG}_layers_l_bf16_batch64.txt" 2>/dev/null || exit 1
timeout -k 10 200 python3 "$ROOT/tools/train_layer_profile.py" 1 > "$OUT/${TAG}_train_layers_bf16.txt" 2>/dev/null || exit 1
echo "layers done"

timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/bench" -o bench -- python3 "$ROOT/bench.py" \
    > "$OUT/${TAG}_bench.json" 2> "$RAW/bench.err" || exit 1
cp "$(find "$RAW/bench" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv" || exit 1
echo "bench stats done"

timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/infer" -o infer -- python3 "$ROOT/bench.py" \
    --no-train --no-extra --no-cpu-baseline > "$OUT/${TAG}_bench_infer.json" 2> "$RAW/infer.err" || exit 1
cp "$(find "$RAW/infer" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_infer_kernel_stats.csv" || exit 1
python3 "$ROOT/tools/family_stats.py" "$OUT/${TAG}_bench_infer_kernel_stats.csv" > "$OUT/${TAG}_bench_infer_family_stats.txt" || exit 1
echo "infer stats done"

timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$RAW/fetch" -o fetch -- python3 "$ROOT/tools/pmc_target.py" 3 \
    > "$RAW/fetch.log" 2>&1 || exit 1
echo "fetch pass done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$RAW/write" -o write -- python3 "$ROOT/tools/pmc_target.py" 3 \
    > "$RAW/write.log" 2>&1 || exit 1
echo "write pass done"
python3 "$ROOT/tools/pmc_summary.py" "$(find "$RAW/fetch" -name '*counter_collection.csv' | head -1)" \
    "$(find "$RAW/write" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_traffic.json" || exit 1

timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$RAW/sq" -o sq -- python3 "$ROOT/tools/pmc_target.py" 3 > "$RAW/sq.log" 2>&1 || exit 1
python3 "$ROOT/tools/pmc_sq_summary.py" "$(find "$RAW/sq" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_sq.json" || exit 1
echo "all done"

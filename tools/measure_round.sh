#!/bin/bash
# Full measurement set of one build on the GPU box (run through gpurun from the repo root):
#   bash tools/measure_round.sh r02_g
# -> gpurun_out/<tag>/: bench.json, kernel stats csv, PMC traffic JSONs (GEMM + ln_bwd). Each GPU step has its own time limit.
set -e -o pipefail
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
timeout -k 10 300 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o $TAG -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-h2d --no-roofline > $OUT/bench_under_rocprof.json 2> $OUT/ks.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-h2d --no-roofline --eager > /dev/null 2> $OUT/pmc_f.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-h2d --no-roofline --eager > /dev/null 2> $OUT/pmc_w.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-h2d --no-roofline --eager > /dev/null 2> $OUT/pmc_sq.err
F=$(ls $OUT/pmc_f/*counter_collection.csv | head -1); W=$(ls $OUT/pmc_w/*counter_collection.csv | head -1)
SQ=$(ls $OUT/pmc_sq/*counter_collection.csv | head -1)
python tools/pmc_counters.py $SQ attn_ $OUT/${TAG}_attn_pmc.json > /dev/null
python tools/pmc_traffic.py $F $W gemm_b $OUT/${TAG}_gemm_pmc.json
python tools/pmc_traffic.py $F $W gemm_big_kernel $OUT/${TAG}_gemm_big_pmc.json
python tools/pmc_traffic.py $F $W ln_bwd_kernel $OUT/${TAG}_ln_bwd_pmc.json
rm -rf $OUT/pmc_f $OUT/pmc_w $OUT/pmc_sq
cat $OUT/bench.json

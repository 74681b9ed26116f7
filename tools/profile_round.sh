#!/bin/bash
# Measurement pass of one build on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02_v6
# writes gpurun_out/<tag>.*: the default bench line, the --ssl line, rocprofv3 kernel statistics of the same bench command, and
# the PMC passes of the dominant kernel (separate runs, --kernel-trace only beside --pmc), then tools/summarize_profile.py turns
# them into the files kept under profiles/.
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
K='conv_zs_kernel'
B=${2:-24576}        # boards per forward in the bench: 256 games x 96 leaves
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/$TAG.bench.json 2> $OUT/$TAG.bench.err || exit 1
echo "[profile] bench done"
python bench.py --ssl --no-cpu-baseline > $OUT/$TAG.bench_ssl.json 2> $OUT/$TAG.bench_ssl.err || exit 1
echo "[profile] bench --ssl done (BASELINE configs[3])"
python bench.py --sims 1600 --steps 10 --no-cpu-baseline > $OUT/$TAG.bench_sims1600.json 2> $OUT/$TAG.bench_sims1600.err || exit 1
echo "[profile] bench --sims 1600 done (BASELINE configs[4] search shape on one GPU)"
python bench.py --streams 2 --no-cpu-baseline > $OUT/$TAG.bench_streams2.json 2> $OUT/$TAG.bench_streams2.err || exit 1
echo "[profile] bench --streams 2 done"
python bench.py --half-split --no-cpu-baseline > $OUT/$TAG.bench_half_split.json 2> $OUT/$TAG.bench_half_split.err || exit 1
echo "[profile] bench --half-split done (engine.tail_split = halves; roofline fields not meaningful in this mode)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG.stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval-cache > $OUT/$TAG.bench_prof.json 2> $OUT/$TAG.stats.err || exit 1
echo "[profile] kernel stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG.netstats -- python3 tools/bench_net.py $B > $OUT/$TAG.netstats.log 2>&1 || exit 1
echo "[profile] forward stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$K" --output-format csv -d $OUT/$TAG.pmc_$C -- python3 tools/bench_net.py $B > $OUT/$TAG.pmc_$C.log 2>&1 || exit 1
  echo "[profile] pmc $C done"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-include-regex "$K" --output-format csv -d $OUT/$TAG.pmc_sq -- python3 tools/bench_net.py $B > $OUT/$TAG.pmc_sq.log 2>&1 || exit 1
echo "[profile] pmc sq done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS --kernel-include-regex "$K" --output-format csv -d $OUT/$TAG.pmc_lds -- python3 tools/bench_net.py $B > $OUT/$TAG.pmc_lds.log 2>&1 || echo "[profile] pmc lds failed (optional)"
echo "[profile] all done"

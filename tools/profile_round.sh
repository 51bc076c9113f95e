#!/bin/bash
# One round's rocprofv3 evidence for bench.py's config-3 step, written under gpurun_out/<tag>_*; run on the GPU box:
#   bash tools/profile_round.sh r03
# Separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (kernel trace + stats; FETCH_SIZE; WRITE_SIZE;
# MFMA busy; the gather's memory-side request and stall counters, 4 TCC counters per pass).  The program itself follows `--`.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
B="python3 $ROOT/bench.py --warmup 5 --no-cpu-baseline --no-second-dist --no-extras"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- $B --steps 20 > $OUT/${TAG}_stats_bench.log 2>$OUT/${TAG}_stats_bench.err && echo stats ok
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o run -- $B --steps 6 > /dev/null 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o run -- $B --steps 6 > /dev/null 2>&1 && echo write ok
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma -o run -- $B --steps 6 > /dev/null 2>&1 && echo mfma ok
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_wr -o run -- $B --steps 6 > /dev/null 2>&1 && echo wrstall ok
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_TAG_STALL_sum --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_rd -o run -- $B --steps 6 > /dev/null 2>&1 && echo rdstall ok
cd $ROOT
python3 tools/prof_summary.py $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_stats_bench.log > $OUT/${TAG}_kernel_stats_table.md
python3 tools/pmc_summary.py $(find $OUT/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/${TAG}_pmc_write -name "*counter_collection.csv" | head -1) $OUT/${TAG}_traffic.json > $OUT/${TAG}_pmc_table.md
python3 tools/mfma_busy.py $(find $OUT/${TAG}_pmc_mfma -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_mfma_busy_table.md
python3 tools/pmc_stalls.py embed_fm_planes_fwd_k $(find $OUT/${TAG}_pmc_wr -name "*counter_collection.csv" | head -1) $(find $OUT/${TAG}_pmc_rd -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_gather_stalls_table.md
python3 tools/pmc_stalls.py sparse_apply_k $(find $OUT/${TAG}_pmc_wr -name "*counter_collection.csv" | head -1) $(find $OUT/${TAG}_pmc_rd -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_apply_stalls_table.md
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_wr $OUT/${TAG}_pmc_rd $OUT/${TAG}_stats/*kernel_trace.csv
head -12 $OUT/${TAG}_kernel_stats_table.md; cat $OUT/${TAG}_gather_stalls_table.md

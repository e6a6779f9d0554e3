#!/bin/bash
# Round profile of the bench workload on the GPU box:  bash tools/profile_round.sh TAG [bench args...]
#   1. bench.py (full line, with cpu_baseline)            -> gpurun_out/prof_TAG/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/prof_TAG/trace/
#   3. rocprofv3 --pmc, one pass per counter group (never mixed with tracing) -> gpurun_out/prof_TAG/pmc{a,b,c,d,e}/
#   4. tools/summarize_profile.py -> gpurun_out/prof_TAG/{kernel_stats.csv,pmc.txt}
# Copy what should be judged from gpurun_out/prof_TAG into profiles/.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"; tail -c 600 "$OUT/bench.json"; echo
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/trace.log" 2>&1
echo "trace done"
G_a="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G_b="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
G_c="FETCH_SIZE"
G_d="WRITE_SIZE"
G_e="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
# counter passes at --inflight 1: one launch per pass and step = the whole batch (the last dispatch of a kernel is what is summarised)
for g in a b c d e; do
  v=G_$g
  rocprofv3 --pmc ${!v} --output-format csv -d "$OUT/pmc$g" -o run -- python3 "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/pmc$g.log" 2>&1
  echo "pmc $g done"
done
python3 "$ROOT/tools/summarize_profile.py" "$OUT" "$TAG"

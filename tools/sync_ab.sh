# A/B of the diagnostic builds (see tools/sync_ab.py); run on a GPU box from the repo root.  Builds are made in the container:
#   cd mpc_motion_planning_amd/csrc; hipcc $FLAGS -DMPCB_NOINLINE_SOLVE [-DMPCB_SYNC_BLOCK | -DMPCB_SYNC_WAITCNT] -o ../lib/variants/<name>.so mpcb_api.hip
set -e
mkdir -p gpurun_out
: > gpurun_out/sync_ab.jsonl
timeout -k 10 200 python tools/sync_ab.py default >> gpurun_out/sync_ab.jsonl
for v in mpc_motion_planning_amd/lib/variants/*.so; do
  MPCB_LIB=$PWD/$v timeout -k 10 200 python tools/sync_ab.py $(basename $v .so) >> gpurun_out/sync_ab.jsonl
done
cat gpurun_out/sync_ab.jsonl

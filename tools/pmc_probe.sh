# Dev tool (GPU box): PMC passes over a balanced probe.  usage: bash tools/pmc_probe.sh "<counters pass 1>" "<pass 2>" ...
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
i=0
for c in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/p$i -o run -- python3 tools/gpu_probe.py unmapped 300 256 100 > gpurun_out/pmc/p$i.log 2>&1
  python3 tools/pmc_sum.py gpurun_out/pmc/p$i >> gpurun_out/pmc/summary.txt
done
cat gpurun_out/pmc/summary.txt

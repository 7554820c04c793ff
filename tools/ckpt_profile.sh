set -e
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo bench done > gpurun_out/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/stats -o run -- python3 bench.py --cpu-baseline 0 --partial-pass 0 > gpurun_out/bench_under_rocprof.log 2> gpurun_out/rocprof_stats.err
echo stats done >> gpurun_out/progress.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc_fetch -o run -- python3 bench.py --cpu-baseline 0 --partial-pass 0 --reads-per-gap 100 > gpurun_out/bench_pmc_fetch.log 2> gpurun_out/rocprof_pmc1.err
echo pmc1 done >> gpurun_out/progress.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pmc_write -o run -- python3 bench.py --cpu-baseline 0 --partial-pass 0 --reads-per-gap 100 > gpurun_out/bench_pmc_write.log 2> gpurun_out/rocprof_pmc2.err
echo pmc2 done >> gpurun_out/progress.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc_fetch_def -o run -- python3 bench.py --cpu-baseline 0 --partial-pass 0 > gpurun_out/bench_pmc_fetch_def.log 2> gpurun_out/rocprof_pmc3.err
echo pmc3 done >> gpurun_out/progress.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pmc_write_def -o run -- python3 bench.py --cpu-baseline 0 --partial-pass 0 > gpurun_out/bench_pmc_write_def.log 2> gpurun_out/rocprof_pmc4.err
echo pmc4 done >> gpurun_out/progress.txt
ls -R gpurun_out/prof | head -30

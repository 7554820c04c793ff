#!/bin/bash
# GPU box: evidence for the large-fill question: the same process fills the 512-gap bench batch, then the
# 2048-gap batch of the same recipe with the scheduler's per-round log, then the 512-gap batch again, while rocm-smi samples
# clocks and power every 3 s.  Output under gpurun_out/largefill/.
set -u
out=gpurun_out/largefill; mkdir -p $out
( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" ; sleep 3; done ) > $out/smi.txt 2>&1 &
SMI=$!
FIG_SCHED_LOG=1 timeout -k 10 900 python tools/gpu_probe.py unmapped mix,mix2048,mix 512 > $out/probe.txt 2> $out/sched.txt
kill $SMI
grep -c figsched $out/sched.txt
grep "^{" $out/probe.txt | cut -c1-200

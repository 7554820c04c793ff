#!/bin/bash
# Dev tool (GPU box): what engine clock and power does the card hold while the bench's fill runs?
# Samples `rocm-smi --showclocks --showpower` once a second from a second process (sysfs reads, no HIP context) while one
# 512-gap fill of the bench set is timed; the FP64 peak bench.py prices against assumes 2.4 GHz.
# usage: tools/clock_probe.sh <out dir>
out=${1:-gpurun_out/clock}
mkdir -p "$out"
python3 bench.py --gpus 1 --total-gaps 512 --steps 1 --warmup 1 --cpu-baseline 0 --partial-pass 0 --bracket-probes 0 > "$out/bench_512.json" 2> "$out/bench_512.err" &
bpid=$!
: > "$out/clock_samples.txt"
while kill -0 $bpid 2>/dev/null; do
    { date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "clk|power"; } >> "$out/clock_samples.txt"
    sleep 1
done
wait $bpid
rc=$?
python3 - "$out" <<'PY'
import re, sys, json
out = sys.argv[1]
txt = open(out + "/clock_samples.txt").read()
sclk = [int(m) for m in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
pw = [float(m) for m in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
rec = {"samples": len(sclk), "sclk_mhz_all": sclk, "power_w_all": pw}
busy = [s for s, p in zip(sclk, pw) if p > 0.6 * max(pw)] if pw and len(pw) == len(sclk) else sclk
if busy:
    rec["sclk_mhz_under_load_mean"] = sum(busy) / len(busy); rec["sclk_mhz_under_load_min"] = min(busy); rec["sclk_mhz_under_load_max"] = max(busy)
if pw: rec["power_w_max"] = max(pw)
try:
    line = json.loads(open(out + "/bench_512.json").read().strip().splitlines()[-1])
    rec["bench"] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "frac_at_2400_mhz": line["roofline"]["frac"]}
    if busy:
        rec["bench"]["frac_at_measured_clock"] = line["roofline"]["frac"] * 2400.0 / rec["sclk_mhz_under_load_mean"]
except Exception as e:
    rec["bench_error"] = repr(e)
open(out + "/clock_probe.json", "w").write(json.dumps(rec) + "\n")
print(json.dumps({k: v for k, v in rec.items() if not k.endswith("_all")}))
PY
exit $rc

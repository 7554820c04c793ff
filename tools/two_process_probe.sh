set -e
A="--scaling weak --steps 2 --warmup 1 --cpu-baseline 0 --partial-pass 0 --bracket-probes 0"
python3 bench.py $A > gpurun_out/tp_single.json 2> gpurun_out/tp_single.err
python3 bench.py $A > gpurun_out/tp_a.json 2> gpurun_out/tp_a.err &
P1=$!
python3 bench.py $A > gpurun_out/tp_b.json 2> gpurun_out/tp_b.err &
P2=$!
wait $P1; wait $P2
for f in single a b; do python3 - <<PY
import json
d=json.loads([l for l in open('gpurun_out/tp_$f.json') if l.startswith('{')][-1])
print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done

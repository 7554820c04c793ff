for mc in 4 8 16 32; do for ipw in 6 12 24 48; do
  r=$(FIG_MIN_CHUNK=$mc FIG_ITEMS_PER_WG=$ipw timeout -k 10 100 python tools/gpu_probe.py partial mix8192 1 48 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gaps_per_s'], d['spec_gflop'])")
  echo "minchunk $mc items_per_wg $ipw: $r"
done; done

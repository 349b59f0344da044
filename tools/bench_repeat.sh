#!/bin/bash
# Three default bench.py runs on one box (A/B of a code change needs the same box: boxes differ by +-3 %).
python3 -c "import torch"
for i in 1 2 3; do
  timeout -s KILL 300 python3 bench.py --cpu-seconds 0 > /tmp/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k: round(v['avg_us']) for k, v in d['kernels'].items()})"
done

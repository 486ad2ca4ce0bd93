#!/bin/bash
# Dev helper: interleaved same-box A/B of the train step under environment settings (engine-side switches).
# usage: ab_env.sh ROUNDS "NAME=V [NAME=V ...]" "..." ...      (an empty string "" = defaults)
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[%s]' % '$v', d['value'], d['ms_per_step'])"
  done
done

#!/bin/bash
# A/B of the Tiny train step (hipGraph replay) under environment switches: one line per variant
run() { echo -n "$1: "; env $1 python bench.py --skip-cpu-baseline --no-alt --steps 40 --warmup 10 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for v in "$@"; do run "$v"; done

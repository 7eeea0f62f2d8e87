#!/bin/bash
# ms per step and per filter level of the bench step for a list of RAGRAPH_FILTER_PARTNER_LEAD values (0 = equal priorities)
for s in "$@"; do
  RAGRAPH_FILTER_PARTNER_LEAD=$s python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('partner_lead $s: ms_per_step', d['ms_per_step'], ' levels', [(l['launch'], l['ms']) for l in r.get('levels', [])], ' call', r.get('retrieval_call_ms'))"
done

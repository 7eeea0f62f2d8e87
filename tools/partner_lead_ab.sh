#!/bin/bash
# A/B of the ring kernel's partner-progress priorities (RAGRAPH_FILTER_PARTNER_LEAD = 0 / 1 / 2) over the shapes the ring kernel
# serves: mid batches on the 1M x 256 bank, D = 128, D = 64 (the edge flavour), and the bench step.
for lead in 0 1 2 0 1; do
  echo "== lead $lead"
  RAGRAPH_FILTER_PARTNER_LEAD=$lead python tools/mid_ab.py 512 1024 2048 4096 8192 16384 2>&1 | tail -1
  RAGRAPH_FILTER_PARTNER_LEAD=$lead MID_D=128 python tools/mid_ab.py 1024 8192 32768 2>&1 | tail -1
  RAGRAPH_FILTER_PARTNER_LEAD=$lead MID_D=64 MID_N=4000000 python tools/mid_ab.py 4096 65536 2>&1 | tail -1
  RAGRAPH_FILTER_PARTNER_LEAD=$lead MID_N=60000 python tools/mid_ab.py 2100 8192 2>&1 | tail -1
  bash tools/partner_lead_sweep.sh $lead
done

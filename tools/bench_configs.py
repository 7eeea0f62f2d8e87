#!/usr/bin/env python3
"""Stand-alone run of bench.py's `configs` and `finetune_step` blocks (tools/bench_blocks.py): the other BASELINE.json
configs -- c1 (Cora-shaped node forward), c3 (PROTEINS-style graph batches), the few-shot node forward, the c5-shaped
single-GPU leg (4096- and 256-query slabs against the 4M x 64 bank, generate() once) -- and the fine-tuning steps, one JSON
object per line.   python tools/bench_configs.py [--no-cpu]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import bench_blocks as BB

dev = torch.device("cuda:0")
torch.manual_seed(0)
cpu = "--no-cpu" not in sys.argv
cores = min(len(os.sched_getaffinity(0)), 16)
print(json.dumps({"finetune_node_528": BB.finetune_node(dev, cores, "528", cpu=cpu)}), flush=True)
cfg, m5 = BB.configs_block(dev)
for k, v in cfg.items():
    print(json.dumps({k: v}), flush=True)
print(json.dumps({"finetune_edge_c5": BB.finetune_edge(dev, cores, m5, cpu=cpu)}), flush=True)

#!/usr/bin/env python3
"""The reference's RAGraph_node/finetune-rag.py driver loop on ragraph_amd, on synthetic TU-shaped data (the reference
ships no dataset and no pre-trained weights, .MISSING_LARGE_BLOBS).  Same steps, same call surface:

    pretrain_model = PrePrompt(F, 256, 'prelu', 1, 0.3)            # finetune-rag.py:40
    rag_model = RAGraph(pretrain_model, train_dataset, F, C, 256)   # :57  (builds the toy-graph bank)
    for epoch: for batch: features, adj, labels = process_tu_dataset(batch, F)
                          logits = rag_model(features, adj); loss = CE(logits, labels); backward; Adam.step   # :74-84
    rag_model.toy_graph_base.build_toy_graph(val_dataset)          # :97
    test accuracy                                                  # :103-112

Usage: python examples/finetune_rag.py [--epochs 5] [--graphs 120]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd.data import DataLoader, synthetic_tu_dataset  # noqa: E402
from ragraph_amd.preprompt import PrePrompt  # noqa: E402
from ragraph_amd.RAGraph import RAGraph  # noqa: E402
from ragraph_amd.ragraph_utils import process_tu_dataset, seed_everything  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--graphs", type=int, default=120)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    seed_everything(0)
    F_attr, C = 18, 3  # ENZYMES-shaped: 18 node attributes + 3 one-hot node labels
    dataset = synthetic_tu_dataset(num_graphs=args.graphs, num_node_attributes=F_attr, num_node_labels=C, seed=9,
                                   name="ENZYMES").shuffle()
    n_train, n_val = int(0.5 * len(dataset)), int(0.2 * len(dataset))
    train_ds, val_ds, test_ds = dataset[:n_train], dataset[n_train:n_train + n_val], dataset[n_train + n_val:]
    pretrain_model = PrePrompt(F_attr, 256, "prelu", 1, 0.3).to(dev)
    t0 = time.perf_counter()
    rag_model = RAGraph(pretrain_model, train_ds, F_attr, C, 256, finetune=True, noise_finetune=False, device=dev)
    torch.cuda.synchronize()
    rag_model.toy_graph_base.show()
    print(f"bank build: {time.perf_counter() - t0:.2f} s for {len(train_ds)} resource graphs")
    opt = torch.optim.Adam(rag_model.parameters(), lr=1e-3)
    xent = torch.nn.CrossEntropyLoss()
    for epoch in range(args.epochs):
        rag_model.train()
        tot, nb, t0 = 0.0, 0, time.perf_counter()
        for batch in DataLoader(train_ds, batch_size=16, shuffle=True):
            features, adj, node_labels = process_tu_dataset(batch, F_attr, device=dev)
            opt.zero_grad()
            logits = rag_model(features, adj)
            loss = xent(logits, node_labels.argmax(dim=1))
            loss.backward()
            opt.step()
            tot, nb = tot + float(loss.detach()), nb + 1
        torch.cuda.synchronize()
        print(f"epoch {epoch}: loss {tot / nb:.4f}  ({time.perf_counter() - t0:.2f} s)")
    rag_model.toy_graph_base.build_toy_graph(val_ds)  # the reference appends the validation graphs before testing
    rag_model.eval()
    correct = total = 0
    with torch.no_grad():
        for batch in DataLoader(test_ds, batch_size=16):
            features, adj, node_labels = process_tu_dataset(batch, F_attr, device=dev)
            pred = rag_model(features, adj).argmax(dim=1)
            correct += int((pred == node_labels.argmax(dim=1)).sum())
            total += pred.numel()
    print(f"test accuracy on synthetic labels: {correct / total:.3f} ({total} nodes; labels are random, chance = {1 / C:.2f})")


if __name__ == "__main__":
    main()

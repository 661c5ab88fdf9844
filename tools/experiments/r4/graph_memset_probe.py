"""Round 4: the hipGraph hazard of round 3, from the graph's own nodes and edges.

Run on the GPU box with the library under test copied over pccx/lib/libpccx.so:
    python3 tools/experiments/r4/graph_memset_probe.py <label> [trials]
(the `memset` variant is built here with PCCX_BUILD_TAG=memset PCCX_EXTRA_FLAGS=-DPCCX_ZERO_WITH_MEMSET python -m pccx.build: pccx_zero_async
issues hipMemsetAsync, i.e. memset NODES between the kernel nodes of the captured step, as rounds 2-3 had them).

1. captures the training step with hipGraphDebugDotPrint on and reads the DOT file: node kinds, edges, and for every memset node its
   predecessors / successors (is it ordered after the last reader and before the accumulating kernel?);
2. replays 4x back to back, then leaves an idle gap (a host sync, eager steps of ANOTHER model, a fill_ of the learning rate) and replays
   once more; reports which captured gradient tensors are non-finite after that replay and how far they are from an eager step's."""
import collections
import copy
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np
import torch

import tests.test_train_step as T
from tests import synth
from pccx import families, train

label = sys.argv[1] if len(sys.argv) > 1 else "product"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out_dir = os.path.join(ROOT, "gpurun_out", "r4g")
os.makedirs(out_dir, exist_ok=True)
dot = os.path.join(out_dir, f"graph_{label}.dot")


def read_dot(path):
    txt = open(path).read()
    nodes, edges = {}, []
    for m in re.finditer(r'"?([\w\.]+)"?\s*\[([^\]]*)\]', txt):
        name, attrs = m.group(1), m.group(2)
        lab = re.search(r'label\s*=\s*"([^"]*)"', attrs)
        if lab and name not in ("graph", "node", "edge"):
            nodes[name] = lab.group(1)
    for m in re.finditer(r'"?([\w\.]+)"?\s*->\s*"?([\w\.]+)"?', txt):
        edges.append((m.group(1), m.group(2)))
    return nodes, edges


o = T._models(2048)
for trial in range(trials):
    g1 = families.PointCloudAE(64, 16, 2048)
    g1.load_state_dict(o.state_dict())
    g1 = g1.cuda()
    g2, g0 = copy.deepcopy(g1), copy.deepcopy(g1)
    names = [n for n, _ in g2.named_parameters()]
    x = torch.from_numpy(synth.train_input(2, 2048)).cuda()
    rng = np.random.default_rng(5)
    starts = [[rng.integers(0, 2048, 2), rng.integers(0, 2048, 2)], rng.integers(0, 512, 2), rng.integers(0, 128, 2)]
    lr = 1e-4
    opt1, opt2 = train.Adam(g1.parameters(), lr=lr), train.Adam(g2.parameters(), lr=lr)
    gs = train.GraphedTrainStep(g2, opt2, x, starts, lam=1e-3, warmup=0, debug_dot=dot if trial == 0 else None)
    if trial == 0 and os.path.exists(dot):
        nodes, edges = read_dot(dot)
        kinds = collections.Counter(re.split(r"[\\\n( ]", v.strip())[0] for v in nodes.values())
        indeg, outdeg = collections.Counter(b for _, b in edges), collections.Counter(a for a, _ in edges)
        roots = [n for n in nodes if indeg[n] == 0]
        leaves = [n for n in nodes if outdeg[n] == 0]
        print(f"[{label}] graph: {len(nodes)} nodes, {len(edges)} edges, kinds {dict(kinds)}; roots {len(roots)}, leaves {len(leaves)}, "
              f"max in-degree {max(indeg.values() or [0])}, max out-degree {max(outdeg.values() or [0])}", flush=True)
        ms = [n for n, v in nodes.items() if "MEMSET" in v.upper()]
        bad = [n for n in ms if indeg[n] == 0 or outdeg[n] == 0]
        print(f"[{label}] memset nodes: {len(ms)}; without a predecessor or without a successor: {len(bad)}", flush=True)
        pred = collections.defaultdict(list)
        succ = collections.defaultdict(list)
        for a, b in edges:
            pred[b].append(a)
            succ[a].append(b)
        for n in ms[:6]:
            print("   memset", n, "<-", [nodes.get(a, a)[:60] for a in pred[n]], "->", [nodes.get(b, b)[:60] for b in succ[n]], flush=True)
    for _ in range(4):
        gs(sync=False)
    torch.cuda.synchronize()
    fin4 = all(bool(torch.isfinite(p).all()) for p in g2.parameters())
    for _ in range(4):                                    # the idle gap of the graph's stream position: other work, host syncs
        train.train_step(g1, opt1, x, starts, lam=1e-3)
    opt2.set_lr(5e-5)
    state = copy.deepcopy(g2.state_dict())
    gs(sync=False)
    torch.cuda.synchronize()
    grads = [g.detach().clone() for g in gs._grads]
    bad = [n for n, g in zip(names, grads) if not bool(torch.isfinite(g).all())]
    # the same step eagerly from the same state (BatchNorm buffers included)
    g3 = families.PointCloudAE(64, 16, 2048).cuda()
    g3.load_state_dict(state)
    opt3 = train.Adam(g3.parameters(), lr=lr)
    train.train_step(g3, opt3, x, starts, lam=1e-3)
    ref = [p.grad for p in g3.parameters() if p.grad is not None]
    worst = 0.0
    if len(ref) == len(grads):
        for n, a, b in zip(names, grads, ref):
            if bool(torch.isfinite(a).all()):
                worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
    print(f"[{label}] trial {trial}: finite after 4 back-to-back replays {fin4}; after the gap: {len(bad)} non-finite gradient tensors "
          f"{bad[:4]}; finite ones within {worst:.2e} (relative to each tensor's largest) of the eager step", flush=True)

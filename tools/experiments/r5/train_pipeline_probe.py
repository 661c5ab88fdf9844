#!/usr/bin/env python3
"""Does the selection of batch i+1 (side stream) really run UNDER the graph replay of step i?  Runs the pipelined loop of
GraphedTrainStep(prefetch=True) a few times; under `rocprofv3 --kernel-trace` the trace shows whether fps_kernel overlaps the graph's
kernels (analyse with --analyse <kernel_trace.csv>)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))


def adam_stream(ks):
    return next(k[4] for k in ks if k[2].startswith("adam_multi"))


def analyse(path):
    import csv
    rows = list(csv.DictReader(open(path)))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id"), r.get("Stream_Id")) for r in rows]
    ks.sort()
    fps = [k for k in ks if k[2].startswith("void fps_kernel<8")]
    print("fps launches:", len(fps))
    for f in fps[-6:]:
        over = [k for k in ks if k[0] < f[1] and k[1] > f[0] and k is not f]
        busy = sum(min(k[1], f[1]) - max(k[0], f[0]) for k in over)
        print("fps %.1f us on queue %s stream %s: %d other kernels overlap it, %.1f us of their time (queues %s)" %
              ((f[1] - f[0]) / 1e3, f[3], f[4], len(over), busy / 1e3, sorted({k[3] for k in over})))
    # the neighbourhood of one fps launch in the middle of the pipelined phase
    mid = fps[len(fps) // 2]
    i0 = ks.index(mid)
    t_ref = ks[max(i0 - 12, 0)][0]
    for k in ks[max(i0 - 12, 0):i0 + 30]:
        print("  %9.1f .. %9.1f us  q%s s%s  %s" % ((k[0] - t_ref) / 1e3, (k[1] - t_ref) / 1e3, k[3], k[4], k[2][:60]))
    # per step on the MAIN queue: span of the graph (pccx_zero_kernel .. adam_multi) and the gap to the next one
    mainq = [k for k in ks if k[4] == adam_stream(ks)]
    zero = [k for k in mainq if k[2].startswith("pccx_zero_kernel")]
    ad = [k for k in mainq if k[2].startswith("adam_multi")]
    spans = []
    for a in ad:
        z = [k for k in zero if k[0] < a[0]]
        if z:
            spans.append(((a[1] - z[-1][0]) / 1e3, a))
    for (sp, a), (_, b) in list(zip(spans, spans[1:]))[-44:]:
        nxt = [k for k in zero if k[0] > a[1]]
        print("  graph span %.1f us, then %.1f us to the next graph's first kernel" % (sp, (nxt[0][0] - a[1]) / 1e3 if nxt else -1))
    # gaps: the period between successive adam_multi kernels = the step time on the GPU
    adam = [k for k in ks if k[2].startswith("adam_multi")]
    per = [(b[0] - a[0]) / 1e6 for a, b in zip(adam, adam[1:])]
    print("period between adam_multi kernels (ms):", [round(p, 3) for p in per[-12:]])


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        return analyse(sys.argv[2])
    import numpy as np
    import torch
    import bench
    from pccx import families, synth, train
    N, Bt = 8192, 4
    model = families.PointCloudAE(64, 16, N)
    model.load_state_dict(bench.seeded_state_dict(model, 32))
    for k, v in model.state_dict().items():
        if k.endswith("running_var"):
            v.fill_(1.0)
    model = model.cuda()
    opt = train.Adam(model.parameters(), lr=1e-3)
    x = torch.from_numpy(np.stack([synth.cad_cloud(900 + i, N) for i in range(Bt)])).cuda()
    rng = np.random.default_rng(0)
    starts = [[rng.integers(0, N, Bt), rng.integers(0, N, Bt)], rng.integers(0, 512, Bt), rng.integers(0, 128, Bt)]
    mode = os.environ.get("PROBE_STARTS", "host")
    if mode == "device":            # start indices already on the device: no host-blocking H2D inside prefetch
        starts = [[torch.as_tensor(s).to("cuda", torch.int32) for s in starts[0]], torch.as_tensor(starts[1]).to("cuda", torch.int32),
                  torch.as_tensor(starts[2]).to("cuda", torch.int32)]
    g = train.GraphedTrainStep(model, opt, x, starts, lam=1e-3, autocast=True, warmup=2, prefetch=True)
    if os.environ.get("PROBE_NOSEL") == "1":      # the pipeline's own overhead: the side stream only copies cached tables
        cached = train.selection_tables(model, x, [[torch.as_tensor(s).to("cuda", torch.int32) for s in starts[0]],
                                                   torch.as_tensor(starts[1]).to("cuda", torch.int32), torch.as_tensor(starts[2]).to("cuda", torch.int32)])
        train.selection_tables = lambda *a: cached
    g.prefetch(x, starts)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            if os.environ.get("PROBE_ORDER") == "after":      # the form that does NOT overlap: the side stream's `ready` event lands behind the replay
                g(sync=False)
                g.prefetch(x, starts)
            else:
                g(sync=False, next_batch=(x, starts))
        torch.cuda.synchronize()
        print("pipelined (%s starts, prefetch %s the replay): %.3f ms/step" % (mode, os.environ.get("PROBE_ORDER", "before"), 1e3 * (time.perf_counter() - t0) / 20), flush=True)
    g(sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        g.graph.replay()
    torch.cuda.synchronize()
    print("graph replay alone: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 20), flush=True)


if __name__ == "__main__":
    main()

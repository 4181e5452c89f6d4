"""Experiment (GPU box): does running the batch as two concurrently replayed half-batch graphs (two streams, two model
instances) beat one full-batch graph?  Quantisation relief: kernels of the two halves fill each other's idle CUs."""
import copy, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from prediff_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
steps = 20


def make(Bh):
    ldm = bench.v1_model("bf16", dev)
    zc = torch.randn((Bh, 7, 16, 16, 64), device=dev)
    st = ldm._graph_step("ddim", Bh, zc, dev)
    st["coef"].copy_(torch.tensor([[0.5, 0.6, 0.0]] * Bh, device=dev))
    st["t"].fill_(500)
    st["z"].normal_()
    return ldm, st


def run(sts, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for st, s in zip(sts, streams):
            with torch.cuda.stream(s):
                st["graph"].replay()
                st["z"].copy_(st["out"])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if ns == 1:
    l1, s1 = make(B)
    t = min(run([s1], [torch.cuda.current_stream()]) for _ in range(3))
    print(f"one graph, B={B}: {B * steps / t:.1f} steps/s")
else:
    ms = [make(B // ns) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for off_ms in (0.0, 6.0, 12.0, 18.0):
        def run_off():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(streams[1]):
                if off_ms > 0:
                    torch.cuda._sleep(int(off_ms * 2.1e6))       # initial skew of lane 1 (cycles at ~2.1 GHz)
            for _ in range(steps):
                for (m, st), s in zip(ms, streams):
                    with torch.cuda.stream(s):
                        st["graph"].replay()
                        st["z"].copy_(st["out"])
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        t = min(run_off() for _ in range(3))
        print(f"{ns} lanes x B={B // ns}, lane-1 start skew {off_ms} ms: {(B // ns) * ns * steps / (t - off_ms * 1e-3):.1f} steps/s (skew excluded)")

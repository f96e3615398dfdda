"""Probe (GPU box): does a captured multi-block torch reduction replay correctly?  torch's global reductions (ATen Reduce.cuh)
zero their inter-block semaphores with hipMemsetAsync before every launch; captured, that becomes a memset node of the HIP graph.
The autograd PPO update replayed as a HIP graph gave wrong bias / std gradients at 24576 rows per mini-batch (column sums are
multi-block there) and exact ones at 1024 rows (single-block) — tools/debug/graph_vs_eager_lockstep.py.  This isolates it:
replay `x.sum(0)` with fresh data and compare with the eager result.    python tools/probes/graph_reduce_repro.py"""
import torch


def trial(rows, cols, replays=200, chain=1):
    x = torch.randn(rows, cols, device="cuda")
    ys = [torch.zeros(cols, device="cuda") for _ in range(chain)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for y in ys:
            y.copy_(x.sum(0))
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k, y in enumerate(ys):            # `chain` back-to-back reductions (the update has dozens)
            y.copy_((x * (k + 1)).sum(0))
    bad = 0
    worst = 0.0
    for r in range(replays):
        x.copy_(torch.randn(rows, cols, device="cuda"))
        g.replay()
        for k, y in enumerate(ys):
            ref = (x * (k + 1)).sum(0)
            if not torch.equal(ref, y):
                bad += 1
                worst = max(worst, float(((ref - y).abs() / (ref.abs() + 1e-6)).max()))
    return bad, worst


if __name__ == "__main__":
    for rows, cols in ((1024, 64), (24576, 64), (24576, 12), (24576, 1), (98304, 256)):
        for chain in (1, 8):
            bad, worst = trial(rows, cols, chain=chain)
            print(f"sum over {rows:6d} rows x {cols:3d} cols, {chain} reductions per graph: {bad} of {200 * chain} replayed results differ from eager"
                  + (f" (worst relative error {worst:.2e})" if bad else ""), flush=True)

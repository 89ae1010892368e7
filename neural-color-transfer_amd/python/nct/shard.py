"""Pair-level sharding and the timed-region protocol used by bench.py and the CLI's `-gpus N`.

Pairs are independent units (transfer_single iterates pairs.txt sequentially and rebuilds all per-pair state,
main.cu:456-543), so the multi-GPU form is: one process per GPU, pair i -> rank i mod N, weights replicated, NO tensor
exchange. The only communication is a barrier on both sides of the timed region and a MAX-reduce of the elapsed time
(RCCL on GPUs — backend "nccl" is RCCL on ROCm — gloo in the CPU tests)."""
import time


def shard_pairs(n_pairs, rank, world):
    """Static round-robin: indices of the pairs this rank processes."""
    return list(range(rank, n_pairs, world))


def timed_region(step, steps, warmup, dist=None, sync=None, device=None):
    """Run `warmup` untimed and `steps` timed calls of step(i); barrier + sync on both sides; returns the MAX over ranks of
    the elapsed seconds (so that `value = world * steps / elapsed` is a whole-job throughput)."""
    import torch

    def barrier():
        if dist is not None and dist.is_initialized():
            dist.barrier()
        if sync is not None:
            sync()

    for i in range(warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed

"""Ensemble sampling sharded over the GPUs of one node (one process per GPU, RCCL over xGMI).

Not in the reference: its ``test_step`` draws the ensemble members one after the other in a Python loop
(scripts/prediff/sevirlr/train_sevirlr_prediff.py:916-951) and DDP shards *contexts*, never one ensemble.
Members are independent latent trajectories, so the denoising loop needs no collective at all: rank r owns members
{k : k mod world == r}, every member's noise comes from its own generator seeded by (base_seed, k) -- results do not
depend on the world size -- weights are replicated, the context is VAE-encoded redundantly per rank (cheaper than a
broadcast dependency), and the ONLY exchange is one all-gather of the decoded frames at the end (393 KB per member at
SEVIR-LR).  On the xGMI full mesh that is a single direct all-gather; `torch.distributed` backend "nccl" is RCCL on ROCm.

Caveat: with knowledge alignment the guidance couples the samples of one batch through a global L2
norm (sevir.py:81-82), so `micro_batch` must equal the reference's batch size for bit-comparable guided results.
"""
from typing import Callable, List, Optional, Sequence

import torch


def shard_members(num_members: int, rank: int, world: int) -> List[int]:
    return list(range(rank, num_members, world))


NOISE_CHUNK_STEPS = 32      # draws of one member fetched per generator call (one launch per member per 32 steps instead of one per step)


def member_noise_fn(latent_shape: Sequence[int], members: Sequence[int], base_seed: int, device) -> Callable[[int], torch.Tensor]:
    """noise(step) -> (len(members), *latent_shape): draw `step` of member k comes from Generator(base_seed + k), always in
    the order step 0 (= x_T), 1, 2, ...  so a member's trajectory is the same on any rank / world size / batch split.
    The draws are fetched NOISE_CHUNK_STEPS steps at a time per member (a member's stream is one sequence of chunk-sized draws from its own
    generator, whatever else is in the batch): 64 members cost 64 small launches per 32 steps, not per step."""
    gens = []
    for k in members:
        g = torch.Generator(device=device)
        g.manual_seed(int(base_seed) + int(k))
        gens.append(g)
    state = {"next": 0, "base": 0, "chunk": None}

    def noise(step: int) -> torch.Tensor:
        assert step == state["next"], "member noise must be drawn in step order"
        state["next"] += 1
        if state["chunk"] is None or step >= state["base"] + NOISE_CHUNK_STEPS:
            state["base"] = step
            state["chunk"] = torch.stack([torch.randn((NOISE_CHUNK_STEPS,) + tuple(latent_shape), generator=g, device=device) for g in gens])
        return state["chunk"][:, step - state["base"]].contiguous()
    return noise


def all_gather_members(local: torch.Tensor, num_members: int, rank: int, world: int, group=None,
                       force_collective: bool = False) -> torch.Tensor:
    """local: (n_local, ...) results of members rank, rank+world, ...  ->  (num_members, ...) on every rank, member order.
    `force_collective`: issue the all-gather even in a world of one (exercises the RCCL path on a single GPU)."""
    import torch.distributed as dist
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        return local
    per = (num_members + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
        out = out.reshape((world, per) + tuple(local.shape[1:]))
    else:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad.contiguous(), group=group)
        out = torch.stack(parts)
    # member k lives at [k % world, k // world]
    idx = torch.arange(num_members, device=local.device)
    return out[idx % world, idx // world]


def sample_ensemble(ldm, cond, num_members: int, base_seed: int = 0, sampler: str = "ddim", ddim_steps: int = 50, eta: float = 0.0,
                    timesteps: Optional[int] = None, micro_batch: Optional[int] = None, group=None, return_decoded: bool = True,
                    use_alignment: bool = False, alignment_kwargs=None, sample_fn: Optional[Callable] = None,
                    force_collective: bool = False) -> torch.Tensor:
    """Draw `num_members` samples for ONE context (cond["y"]: (1, T_in, H, W, C)) across all ranks of `group`.

    Returns (num_members, T_out, H, W, C) on every rank.  `sample_fn(cond_batch, batch, noise_fn)` can replace the call into
    `ldm.sample` (used by the CPU gloo tests of the sharding logic)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = shard_members(num_members, rank, world)
    y = cond["y"] if isinstance(cond, dict) else cond
    device = y.device
    latent_shape = tuple(ldm.latent_shape) if ldm is not None else None
    outs = []
    mb = micro_batch or max(1, len(mine))
    for i in range(0, len(mine), mb):
        ks = mine[i:i + mb]
        yb = y.expand(len(ks), *y.shape[1:]).contiguous()
        cb = {"y": yb} if isinstance(cond, dict) else yb
        if sample_fn is not None:
            outs.append(sample_fn(cb, ks))
            continue
        noise = member_noise_fn(latent_shape, ks, base_seed, device)
        kw = dict(batch_size=len(ks), return_decoded=return_decoded, noise_tape=_LazyTape(noise))
        if sampler == "ddim":
            out = ldm.sample(cb, sampler="ddim", ddim_steps=ddim_steps, eta=eta, **kw)
        else:
            ak = None
            if use_alignment and alignment_kwargs is not None:
                ak = {k: (v.expand(len(ks), *v.shape[1:]) if torch.is_tensor(v) and v.shape[0] == 1 else v) for k, v in alignment_kwargs.items()}
            out = ldm.sample(cb, timesteps=timesteps, use_alignment=use_alignment, alignment_kwargs=ak, **kw)
        outs.append(out)
    if outs:
        local = torch.cat(outs)
    else:   # more ranks than members
        probe = shard_members(num_members, 0, world)
        raise ValueError(f"rank {rank} has no ensemble member (num_members={num_members} < world={world}); "
                         f"use num_members >= world (rank 0 would own {probe})")
    return all_gather_members(local, num_members, rank, world, group, force_collective=force_collective)


class _LazyTape:
    """Indexable view over a step-ordered noise function: tape[k] draws step k (each k exactly once, in order)."""

    def __init__(self, fn):
        self.fn = fn
        self.cache = {}

    def __getitem__(self, k):
        if k not in self.cache:
            self.cache = {k: self.fn(k)}
        return self.cache[k]

"""Multi-GPU: one process per GPU, clips sharded across ranks, no communication during sampling,
one all-gather of the results at the end (modelled on mogen/apis/test.py:129-160
`collect_results_gpu`, which the reference defines but never calls).

Clips are independent (no cross-sample op in the VAEs, denoiser, samplers or retrieval), weights and
the retrieval DB are replicated, so scaling is weak by construction.  On MI355X the backend "nccl" is
RCCL over xGMI; the same code runs on CPU tensors with "gloo" (tests).
"""
import torch
import torch.distributed as dist

RESULT_KEYS = ("pred_upper", "pred_lower", "pred_facepose", "pred_hands", "pred_transl", "pred_exps")


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for this rank (first ranks get the remainder)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(data, rank, world):
    """Slice every per-clip tensor/list of a collated batch (mogen/datasets/builder.py:55-92 schema)."""
    B = data["motion_upper"].shape[0]
    lo, hi = shard_range(B, rank, world)
    out = {}
    for k, v in data.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B:
            out[k] = v[lo:hi]
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def pack_results(results):
    return torch.cat([results[k] for k in RESULT_KEYS], dim=-1).contiguous()


def unpack_results(packed, like):
    out, c = {}, 0
    for k in RESULT_KEYS:
        w = like[k].shape[-1]
        out[k] = packed[..., c:c + w]
        c += w
    return out


def gather_results(results, n_total=None, group=None):
    """All-gather the per-rank result tensors into the full batch on every rank.  Shards may be ragged
    (n_total not divisible by world): they are padded to the largest shard for the collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {k: results[k] for k in RESULT_KEYS}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    packed = pack_results(results)
    b_local = packed.shape[0]
    if n_total is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=packed.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([b_local], dtype=torch.int64, device=packed.device), group=group)
        sizes = [int(s.item()) for s in sizes]
    else:
        sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    b_max = max(sizes)
    if b_local < b_max:
        pad = torch.zeros(b_max - b_local, *packed.shape[1:], dtype=packed.dtype, device=packed.device)
        packed = torch.cat([packed, pad], dim=0)
    gathered = torch.empty(world * b_max, *packed.shape[1:], dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    parts = [gathered[r * b_max:r * b_max + sizes[r]] for r in range(world)]
    return unpack_results(torch.cat(parts, dim=0), results)

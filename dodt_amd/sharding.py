"""Multi-GPU sharding of the frame-pair stream.

Frame pairs are independent through NMS #2 (batch size 1, no cross-pair state:
avod/core/models/dt_rpn_model.py:733-735), so they shard round-robin over ranks with
no data-path collective; the only exchange is an all-gather of the fixed-size
detection records, which the sequential temporal module consumes
(avod/core/dt_evaluator_utils.py:212-367).  torch.distributed is used as plumbing
(backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests).
"""
import numpy as np

MAX_DET = 100
REC_COLS = 17


def pairs_for_rank(n_pairs, rank, world):
    """Global pair ids of `rank`: pair i -> rank i mod world (static round-robin)."""
    return list(range(rank, n_pairs, world))


def step_pairs(step, pairs_per_step, rank, world):
    """Global ids of the pairs rank `rank` processes in step `step` (weak scaling:
    every step the ranks together consume world * pairs_per_step consecutive pairs)."""
    base = step * world * pairs_per_step
    return [base + j * world + rank for j in range(pairs_per_step)]


def all_gather_records(dist, rec, cnt, gathered, gathered_cnt):
    """One exchange per step: (pairs,2,MAX_DET,REC_COLS) float32 + (pairs,2) int32 per
    rank -> rank-major concatenation (world*pairs, ...) on every rank (the output form
    both the gloo and the nccl/RCCL backends accept)."""
    dist.all_gather_into_tensor(gathered, rec)
    dist.all_gather_into_tensor(gathered_cnt, cnt)


def merge_step(gathered, gathered_cnt, step, pairs_per_step, world):
    """Records of one step in global pair order: list of (pair_id, frame, (n,17) array)."""
    g = np.asarray(gathered).reshape(world, pairs_per_step, 2, MAX_DET, REC_COLS)
    c = np.asarray(gathered_cnt).reshape(world, pairs_per_step, 2)
    out = []
    for rank in range(world):
        for j, pid in enumerate(step_pairs(step, pairs_per_step, rank, world)):
            for f in range(2):
                out.append((pid, f, g[rank, j, f, :c[rank, j, f]]))
    out.sort(key=lambda t: (t[0], t[1]))
    return out

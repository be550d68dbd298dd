"""Host logic of the episode pool (kinpoly_amd/rollout.py): the ring arithmetic that decides which context rows a top-up rewrites.
Pure torch, runs without a GPU; the device side (kp_pool_advance, the row writes, the sampler loop) is covered by tests/test_gpu_sampler.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _advance(head, ahead, done, D):
    """what kp_pool_advance does on the device (k_pool_advance, kp_rollout_kernels.hpp)"""
    head = torch.where(done, (head + 1) % D, head)
    return head, ahead - done.to(ahead.dtype)


@pytest.mark.parametrize("depth", [1, 2, 4, 7])
def test_ring_never_runs_dry_and_never_rewrites_a_row_in_play(depth):
    """An env ends at most one episode per step and is topped up every `depth` steps, so it always finds a queued clip; a top-up writes only
    used-up slots: never the slot an env is playing, never a queued clip that has not been played (sample_seq per episode, agent_ar.py:518-535)."""
    from kinpoly_amd.rollout import ring_refill_plan
    N, D = 37, depth + 1
    g = torch.Generator().manual_seed(depth)
    head = torch.zeros(N, dtype=torch.int32); ahead = torch.full((N,), D - 1, dtype=torch.int32)
    clip = torch.arange(D * N).clone()              # id of the clip every row holds; fresh ids are handed out by the top-ups
    next_id = D * N
    played = []                                     # (env, clip id) of every episode start
    for e in range(N):
        played.append((e, int(clip[e])))
    for step in range(1, 8 * depth + 1):
        p = [0.0, 0.3, 1.0][step % 3]               # nobody / some / everybody ends an episode on this step
        done = torch.rand(N, generator=g) < p
        head, ahead = _advance(head, ahead, done, D)
        assert int(ahead.min()) >= 0
        for e in done.nonzero().flatten().tolist():
            played.append((e, int(clip[int(head[e]) * N + e])))
        if step % depth == 0:
            deficit = (D - 1) - ahead
            total = int(deficit.sum())
            env_idx, rows = ring_refill_plan(head, ahead, D, total)
            assert rows.numel() == total == env_idx.numel()
            assert len(set(rows.tolist())) == total, "a row was written twice in one top-up"
            in_play = head.long() * N + torch.arange(N)
            assert not set(rows.tolist()) & set(in_play.tolist()), "a top-up overwrote the clip an env is on"
            queued = set()
            for e in range(N):
                for k in range(1, int(ahead[e]) + 1):
                    queued.add(((int(head[e]) + k) % D) * N + e)
            assert not set(rows.tolist()) & queued, "a top-up overwrote a queued clip that was never played"
            assert (rows % N == env_idx).all()
            clip[rows] = torch.arange(next_id, next_id + total); next_id += total
            ahead = ahead + deficit
            assert (ahead == D - 1).all()
    ids = [c for _, c in played]
    assert len(ids) == len(set(ids)), "an env played the same drawn clip twice"
    assert len(played) > N * 4


def test_refill_plan_orders_an_envs_new_clips_behind_its_queue():
    from kinpoly_amd.rollout import ring_refill_plan
    N, D = 3, 5
    head = torch.tensor([4, 0, 2], dtype=torch.int32); ahead = torch.tensor([1, 4, 0], dtype=torch.int32)
    env_idx, rows = ring_refill_plan(head, ahead, D, 3 + 0 + 4)
    np.testing.assert_array_equal(env_idx.numpy(), [0, 0, 0, 2, 2, 2, 2])
    # env 0: playing slot 4, slot 0 queued -> new clips in slots 1, 2, 3; env 2: playing slot 2, nothing queued -> slots 3, 4, 0, 1
    np.testing.assert_array_equal((rows // N).numpy(), [1, 2, 3, 3, 4, 0, 1])


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_lagged_ring_never_runs_dry_and_never_rewrites_a_row_in_play(depth):
    """VectorSampler(lagged=True): the counts a top-up acts on are those of the PREVIOUS period (the host reads them without waiting for the device), so a
    refill restores the ring to the level it had a period ago; with 2 * depth + 1 rows per env an env that ends an episode on every step still always
    finds a queued clip, and the rows a late refill writes -- planned from the old snapshot -- are never the one in play nor a queued, unplayed one."""
    from kinpoly_amd.rollout import ring_refill_plan
    N, D = 29, 2 * depth + 1
    g = torch.Generator().manual_seed(10 + depth)
    head = torch.zeros(N, dtype=torch.int32); ahead = torch.full((N,), D - 1, dtype=torch.int32)
    clip = torch.arange(D * N).clone()
    next_id = D * N
    played = [(e, int(clip[e])) for e in range(N)]
    pending = None
    for step in range(1, 10 * depth + 1):
        p = [1.0, 1.0, 0.3, 0.0][(step // (2 * depth)) % 4]      # stretches where EVERY env ends an episode on every step
        done = torch.rand(N, generator=g) < p
        head, ahead = _advance(head, ahead, done, D)
        assert int(ahead.min()) >= 0, "the ring ran dry"
        for e in done.nonzero().flatten().tolist():
            played.append((e, int(clip[int(head[e]) * N + e])))
        if step % depth == 0:
            if pending is not None:
                head_s, ahead_s, deficit_s = pending
                total = int(deficit_s.sum())
                env_idx, rows = ring_refill_plan(head_s, ahead_s, D, total)
                in_play = head.long() * N + torch.arange(N)
                assert not set(rows.tolist()) & set(in_play.tolist()), "a late refill overwrote the clip an env is on"
                queued = {((int(head[e]) + k) % D) * N + e for e in range(N) for k in range(1, int(ahead[e]) + 1)}
                assert not set(rows.tolist()) & queued, "a late refill overwrote a queued clip that was never played"
                clip[rows] = torch.arange(next_id, next_id + total); next_id += total
                ahead = ahead + deficit_s
            pending = (head.clone(), ahead.clone(), (D - 1) - ahead)
    ids = [c for _, c in played]
    assert len(ids) == len(set(ids)), "an env played the same drawn clip twice"

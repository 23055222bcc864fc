"""Data-parallel path on CPU: 2 processes over gloo.  The trainer / flat-gradient all-reduce are
model-agnostic, so they are exercised here with the CPU oracle model (tests may import oracle/);
on the GPU box the same code runs with the HIP Model_flow over RCCL (bench.py --gpus N).

Property: averaging the gradients of equal per-rank shards reproduces the
reference's DataParallel global-batch mean, so N ranks x B/N samples == 1 process x B samples."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as R
from unopticalflow_amd.parallel import FlatGradients, init_distributed, shard_batch
from unopticalflow_amd.trainer import FlowTrainer

H = W = 64


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(seed_params=1234):
    cfg = R.default_cfg()
    model = R.Model_flow(cfg)
    model.load_state_dict(R.seeded_state_dict(model, seed_params, 0.25))
    return cfg, model


def _worker(rank, world, port, steps, out_path):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, lr, w = init_distributed('gloo')
    assert (r, w) == (rank, world)
    cfg, model = _make(seed_params=1234 if rank == 0 else 999)   # rank 0's weights must win
    trainer = FlowTrainer(cfg, model, distributed=True, fused_adam=False)
    x = R.synthetic_triplets(2 * world, H, W, seed=5, structured=True)
    mine = shard_batch(x, rank, world)
    early = []
    for _ in range(steps):
        trainer.grads.zero()
        loss_pack = trainer.model(mine)                     # the step of FlowTrainer.step, opened up to look at
        trainer.total_loss(loss_pack).backward()            # how many pieces left during backward
        early.append(trainer.grads.launched_early)
        trainer.grads.all_reduce_mean()
        trainer.optimizer.step()
        loss = trainer.total_loss(loss_pack).detach()
    assert trainer.grads.overlap and early == [trainer.grads.chunks] * steps, early   # all pieces sent by the hooks
    trainer.grads.check_views()
    if rank == 0:
        torch.save({'grad': trainer.grads.vector(), 'params': [p.detach().clone() for p in model.parameters()],
                    'loss': loss}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('steps', [1, 2])
def test_two_ranks_match_single_process(tmp_path, steps):
    world = 2
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker, args=(world, _free_port(), steps, out), nprocs=world, join=True)
    got = torch.load(out)

    torch.set_num_threads(4)
    cfg, model = _make()
    trainer = FlowTrainer(cfg, model, distributed=False, fused_adam=False)
    x = R.synthetic_triplets(2 * world, H, W, seed=5, structured=True)
    for _ in range(steps):
        trainer.step(x)
    g_ref = trainer.grads.vector()
    scale = g_ref.abs().max().item()
    np.testing.assert_allclose(got['grad'].numpy(), g_ref.numpy(), rtol=1e-4, atol=1e-5 * scale)
    for a, b in zip(got['params'], model.parameters()):
        np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-4, atol=2e-5)  # Adam: |update| <= lr = 1e-4


def _worker_flat(rank, world, port, steps, out_path):
    """The replayed multi-rank step's exchange without the graphs (they need a GPU): backward assigns, ``pack_all`` copies
    every piece into the flat buffer (what graph A ends with), ONE ``all_reduce_flat``, Adam on the buffer's views (graph B)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    init_distributed('gloo')
    cfg, model = _make(seed_params=1234 if rank == 0 else 999)
    trainer = FlowTrainer(cfg, model, distributed=True, fused_adam=False, use_graph=True)   # graph mode: no hooks
    assert not trainer.grads.overlap and not trainer.grads._hooks
    x = R.synthetic_triplets(2 * world, H, W, seed=5, structured=True)
    mine = shard_batch(x, rank, world)
    for _ in range(steps):
        trainer.grads.zero()
        loss_pack = trainer.model(mine)
        trainer.total_loss(loss_pack).backward()
        assert trainer.grads.launched_early == 0
        trainer.grads.pack_all()
        trainer.grads.check_views()
        trainer.grads.all_reduce_flat()
        trainer.optimizer.step()
    if rank == 0:
        torch.save({'grad': trainer.grads.vector(), 'params': [p.detach().clone() for p in model.parameters()]}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_flat_exchange_matches_single_process(tmp_path):
    world, steps = 2, 2
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker_flat, args=(world, _free_port(), steps, out), nprocs=world, join=True)
    got = torch.load(out)
    torch.set_num_threads(4)
    cfg, model = _make()
    trainer = FlowTrainer(cfg, model, distributed=False, fused_adam=False)
    x = R.synthetic_triplets(2 * world, H, W, seed=5, structured=True)
    for _ in range(steps):
        trainer.step(x)
    g_ref = trainer.grads.vector()
    np.testing.assert_allclose(got['grad'].numpy(), g_ref.numpy(), rtol=1e-4, atol=1e-5 * g_ref.abs().max().item())
    for a, b in zip(got['params'], model.parameters()):
        np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-4, atol=2e-5)


def test_flat_gradients_alias_and_zero():
    cfg, model = _make()
    fg = FlatGradients(model.parameters())
    assert fg.numel == 5134324
    x = R.synthetic_triplets(1, H, W, seed=1)
    R.total_loss(model(x), R.generate_loss_weights_dict(cfg)).backward()
    fg.check_views()
    assert fg.flat.abs().sum() > 0
    n0 = sum(p.grad.abs().sum() for p in model.parameters())
    assert torch.allclose(n0, fg.flat.abs().sum(), rtol=1e-5)
    fg.zero()
    assert all(float(p.grad.abs().sum()) == 0.0 for p in model.parameters())
    model.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        fg.check_views()


def test_flat_gradient_pieces_cover_buffer():
    cfg, model = _make()
    fg = FlatGradients(model.parameters(), chunks=4)
    assert fg.chunks == 4 and fg.pieces[0][0] == 0 and fg.pieces[-1][1] == fg.numel
    assert all(a[1] == b[0] for a, b in zip(fg.pieces[:-1], fg.pieces[1:]))
    assert sum(n for _, _, n in fg.pieces) == 98
    sizes = [b - a for a, b, _ in fg.pieces]
    assert max(sizes) < 0.45 * fg.numel                        # roughly balanced
    many = FlatGradients(model.parameters(), chunks=1000)      # more pieces than parameters: still a partition
    assert 4 < many.chunks <= 98 and many.pieces[-1][1] == many.numel and sum(n for _, _, n in many.pieces) == 98


def test_packed_gradients_single_rank_group():
    """FlatGradients(pack=True), the trainer's mode: backward assigns the gradients, a piece is copied into the buffer when
    its last gradient exists (three hooks per piece, not one per parameter), ``p.grad`` then aliases the buffer; a parameter
    the loss did not reach contributes zeros."""
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    init_distributed('gloo', force=True)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
        unused = torch.nn.Parameter(torch.ones(7))
        params = list(net.parameters()) + [unused]
        fg = FlatGradients(params, chunks=2, overlap=True, single_rank_collectives=True, pack=True)
        assert len(fg._hooks) == 6 and all(p.grad is None for p in params)
        x = torch.randn(3, 6)
        for _ in range(2):                                   # the second step must not see the first one's buffer contents
            fg.zero()
            assert all(p.grad is None for p in params)
            net(x).square().sum().backward()
            want = [p.grad.clone() for p in net.parameters()]
            assert fg.launched_early >= 1                    # the piece without the unused parameter left from a hook
            fg.all_reduce_mean()
            fg.check_views()
            for p, w in zip(net.parameters(), want):
                assert torch.equal(p.grad, w)
            assert torch.equal(unused.grad, torch.zeros(7))
            assert torch.equal(fg.vector(), torch.cat([w.reshape(-1) for w in want] + [torch.zeros(7)]))
    finally:
        dist.destroy_process_group()


def test_shard_batch_requires_divisible():
    with pytest.raises(ValueError):
        shard_batch(torch.zeros(3, 1), 0, 2)
    assert shard_batch(torch.arange(8).view(8, 1), 1, 4).flatten().tolist() == [2, 3]


def test_checkpoint_roundtrip_and_module_prefix(tmp_path):
    """Checkpoint dict of train.py:23-31; DataParallel-style 'module.' keys load too."""
    cfg, model = _make()
    tr = FlowTrainer(cfg, model, fused_adam=False)
    x = R.synthetic_triplets(1, H, W, seed=2)
    tr.step(x)
    path = str(tmp_path / 'last.pth')
    tr.save(path)
    data = torch.load(path)
    assert set(data) == {'iteration', 'model_state_dict', 'optimizer_state_dict'} and data['iteration'] == 1
    data['model_state_dict'] = {'module.' + k: v for k, v in data['model_state_dict'].items()}
    torch.save(data, path)
    cfg2, model2 = _make(seed_params=7)
    tr2 = FlowTrainer(cfg2, model2, fused_adam=False)
    assert tr2.load(path) == 1
    for a, b in zip(model.parameters(), model2.parameters()):
        assert torch.equal(a, b)
    l1, _ = tr.step(x)
    l2, _ = tr2.step(x)
    assert torch.allclose(l1, l2, rtol=1e-6)


# ------------------------------------------------------------------------------------ train.py plumbing (CPU)
@pytest.mark.parametrize('world,batch,iters,start', [(1, 8, 50, 0), (4, 8, 50, 0), (8, 64, 25, 5), (2, 6, 7, 0)])
def test_every_rank_gets_all_scheduled_iterations(tmp_path, world, batch, iters, start):
    """train.py sizes the dataset with the GLOBAL batch (reference train.py:110); DistributedSampler then hands each
    rank 1/world of it, so every rank's loader must yield exactly num_iterations - iter_start batches (sizing it with
    the per-rank batch silently ran 1/world of the schedule)."""
    import types
    from unopticalflow_amd.train import training_items
    from unopticalflow_amd.data import DecodedTriplets
    (tmp_path / 'train.txt').write_text(''.join('seq/%d.png seq/%d_cam.txt\n' % (i, i) for i in range(11)))
    cfg = types.SimpleNamespace(batch_size=batch, num_iterations=iters, iter_start=start)
    n_items, per_rank = training_items(cfg, world)
    assert per_rank * world == batch and n_items == (iters - start) * batch
    ds = DecodedTriplets(str(tmp_path), 3, (64, 128), n_items)
    for rank in range(world):
        sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True) if world > 1 else None
        loader = torch.utils.data.DataLoader(ds, batch_size=per_rank, sampler=sampler, drop_last=False, collate_fn=lambda s: s)
        assert len(loader) == iters - start, (rank, len(loader))
    with pytest.raises(ValueError):
        training_items(types.SimpleNamespace(batch_size=6, num_iterations=4, iter_start=0), 4)


def test_gpu_flag_selects_devices():
    """--gpu is the device list (reference train.py:198: CUDA_VISIBLE_DEVICES = args.gpu): rank r drives its r-th id."""
    from unopticalflow_amd.train import gpu_ids
    assert gpu_ids('0') == [0] and gpu_ids('3') == [3] and gpu_ids('4,5,6,7') == [4, 5, 6, 7] and gpu_ids(2) == [2]
    with pytest.raises(ValueError):
        gpu_ids('')


def test_single_rank_group_runs_the_collectives():
    """A process group of ONE rank (what the RCCL rehearsal on a single-GPU box uses): with single_rank_collectives the
    hooks launch every piece during backward and all_reduce_mean leaves the gradient unchanged (sum over 1 rank / 1)."""
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    r, lr, w = init_distributed('gloo', force=True)
    try:
        assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
        torch.set_num_threads(2)
        cfg, model = _make()
        tr = FlowTrainer(cfg, model, distributed=True, fused_adam=False, single_rank_collectives=True)
        x = R.synthetic_triplets(1, H, W, seed=5, structured=True)
        tr.grads.zero()
        tr.total_loss(tr.model(x)).backward()
        assert tr.grads.launched_early == tr.grads.chunks
        before = tr.grads.flat.clone()
        tr.grads.all_reduce_mean()
        assert torch.equal(before, tr.grads.flat)
        # without the flag a one-rank group stays silent (no collectives for a plain single-process run)
        cfg2, model2 = _make()
        tr2 = FlowTrainer(cfg2, model2, distributed=True, fused_adam=False)
        tr2.grads.zero()
        tr2.total_loss(tr2.model(x)).backward()
        assert tr2.grads.launched_early == 0
    finally:
        dist.destroy_process_group()


def test_eager_step_after_a_graph_capture_packs_fresh_gradients():
    """A hipGraph capture leaves ``remember_sources()`` behind (the graph's static gradient tensors).  A later EAGER step
    (FlowTrainer.step falls through for a differently shaped batch) must exchange what ITS backward produced: zero() drops
    the views of the flat buffer, the pack reads p.grad -- never the graph's tensors, which hold the previous replay's
    gradients.  The capture is simulated on CPU (no graph needed for the bookkeeping): the 'captured' gradients are the
    tensors of a first backward."""
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    init_distributed('gloo', force=True)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
        params = list(net.parameters())
        fg = FlatGradients(params, chunks=2, overlap=False, single_rank_collectives=True, pack=True)   # (graph mode: no hooks)
        xa, xb = torch.randn(3, 6), torch.randn(5, 6)

        fg.zero()                                           # "capture": backward assigns the graph's static tensors
        net(xa).square().sum().backward()
        fg.remember_sources()
        static = [g for g in fg._sources]
        want_a = [g.clone() for g in static]
        fg.all_reduce_mean(from_graph=True)                 # a replay: packs from the static tensors
        fg.check_views()
        for p, w in zip(params, want_a):
            assert torch.equal(p.grad, w)

        fg.zero()                                           # eager step on another batch shape
        assert all(p.grad is None for p in params)
        net(xb).square().sum().backward()
        want_b = [p.grad.clone() for p in params]
        assert not any(torch.equal(a, b) for a, b in zip(want_a, want_b))
        fg.all_reduce_mean()
        fg.check_views()
        for p, w in zip(params, want_b):
            assert torch.equal(p.grad, w)                   # not the stale replay gradients, not a sum with them

        for g, w in zip(static, want_a):                    # the next replay rewrites the static tensors; exchange again
            g.copy_(2 * w)
        fg.all_reduce_mean(from_graph=True)
        for p, w in zip(params, want_a):
            assert torch.equal(p.grad, 2 * w)
        with pytest.raises(RuntimeError):
            FlatGradients(params, chunks=2, pack=True).all_reduce_mean(from_graph=True)
    finally:
        dist.destroy_process_group()


def test_resume_across_memory_formats_relays_adam_state(tmp_path):
    """A checkpoint written with row-major (NCHW) conv weights resumed into channels_last weights, and back: Adam's
    moments must take the parameter's strides (the fused multi-tensor Adam walks parameter, gradient and moments by
    memory offset), values unchanged, and training continues identically to the uninterrupted run."""
    from unopticalflow_amd.core.networks.structures.net_utils import weights_to_channels_last
    from unopticalflow_amd.trainer import relayout_optimizer_state
    x = R.synthetic_triplets(1, H, W, seed=2)

    def trainer_for(cl):
        cfg, model = _make()
        if cl:
            weights_to_channels_last(model)
        return FlowTrainer(cfg, model, fused_adam=False)

    for src_cl, dst_cl in ((False, True), (True, False)):
        tr = trainer_for(src_cl)
        tr.step(x)
        path = str(tmp_path / ('ckpt_%d.pth' % src_cl))
        tr.save(path)
        tr2 = trainer_for(dst_cl)
        tr2.load(path)
        n4 = 0
        for p in tr2.optimizer.param_groups[0]['params']:
            st = tr2.optimizer.state[p]
            for k in ('exp_avg', 'exp_avg_sq'):
                assert st[k].stride() == p.stride(), (k, tuple(p.shape), st[k].stride(), p.stride())
            n4 += p.dim() == 4
        assert n4 == 49
        for (pa, sa), (pb, sb) in zip(tr.optimizer.state.items(), tr2.optimizer.state.items()):
            assert torch.equal(sa['exp_avg'], sb['exp_avg']) and torch.equal(sa['exp_avg_sq'], sb['exp_avg_sq'])
        l1, _ = tr.step(x)
        l2, _ = tr2.step(x)
        assert torch.allclose(l1, l2, rtol=1e-6)
        for a, b in zip(tr.model.parameters(), tr2.model.parameters()):
            assert torch.allclose(a, b, rtol=0, atol=2.5e-4)    # one more Adam step of <= lr; the two layouts take different CPU conv paths
    relayout_optimizer_state(torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))]))      # empty state: nothing to do

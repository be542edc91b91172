"""Host-side training logic on CPU: LR schedule, label assignment, synthetic data, and the bucketed
gradient all-reduce on 2 gloo ranks."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import labels as olabels, patterns


def test_lr_adjust_matches_reference_fixture(golden):
    from objectdetection_amd.train_class import lr_adjust
    g = golden('training.npz')
    for tag, (step_epoch, lr_end_epoch, warm) in (('a', (10, 100, 0.01)), ('b', (400, 3, 0.05))):
        a = argparse.Namespace(lr_start=0.001, lr_end_ratio=0.1, lr_end_epoch=lr_end_epoch, warmup_ratio=warm)
        sched = lr_adjust(a, step_epoch, 0)

        class _Opt:
            param_groups = [{'lr': 0.0}]
        opt, seq = _Opt(), []
        for _ in range(30):
            sched(opt)
            seq.append(opt.param_groups[0]['lr'])
        assert np.array_equal(np.array(seq), g[f'lr.{tag}'])


def test_assign_labels_equals_reference_restatement():
    from objectdetection_amd.train_class import assign_labels, prediction_ranges
    args = patterns.default_args(input_size=320, output_class=4)
    lo, hi = prediction_ranges(args)
    rlo, rhi = olabels.ranges(args)
    assert torch.allclose(lo, rlo[:, 0:4], rtol=1e-6, atol=1e-5) and torch.allclose(hi, rhi[:, 0:4], rtol=1e-6, atol=1e-5)
    g = torch.Generator().manual_seed(5)
    for trial in range(6):
        n = 10
        label = torch.cat([torch.rand(n, 2, generator=g) * 256 + 32, torch.rand(n, 2, generator=g) * 150 + 10,
                           torch.rand(n, 5, generator=g)], 1)
        if trial == 0:
            label[1, 0:4] = label[0, 0:4] + 0.5          # two labels on the same rows
        idx, exp = assign_labels(label, rlo[:, 0:4], rhi[:, 0:4])
        ridx, rexp = olabels.assign(label, rlo[:, 0:4], rhi[:, 0:4])
        assert torch.equal(idx, ridx)
        assert torch.equal(exp, rexp)
        assert len(idx) > 0
    idx, exp = assign_labels(torch.zeros((0, 9)), rlo[:, 0:4], rhi[:, 0:4])
    assert len(idx) == 0 and exp.shape == (0, 9)


def test_synthetic_dataset_batches():
    from objectdetection_amd.train_class import synthetic_dataset
    args = patterns.default_args(input_size=160, output_class=5)
    ds = synthetic_dataset(args, 4)
    img, screen, expend, labels = ds.collate_fn([ds[0], ds[1]])
    assert img.shape == (2, 3, 160, 160) and len(screen) == 2 and expend.shape[1] == 10
    assert sum(len(s) for s in screen) == len(expend)
    a, b = ds[1], ds[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])       # deterministic per index


def _reducer_rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from objectdetection_amd.train_engine import GradReducer
    n = 1003
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradReducer(flat, n_buckets=3)
    assert red.world == world and red.bounds[0][0] == 0 and red.bounds[-1][1] == n
    assert [red.bucket_of(o) for o in (0, 334, 335, 1002)] == [0, 0, 1, 2]
    for i in (2, 0, 1):          # buckets become ready out of order, as in a backward pass
        red.ready(i)
    scale = red.finish()
    got = flat * scale
    want = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1)) / world
    out[rank] = bool(torch.allclose(got, want))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_all_reduce_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_reducer_rank, args=(r, 2, 29517, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out[0] and out[1]


def _parallel_rank(rank, world, port, out):
    """One data-parallel step's collectives on host tensors: the branch a GPU run takes with RCCL (asynchronous
    handles on the flat buffers themselves, no host staging)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from objectdetection_amd import train_engine as te
    from objectdetection_amd.model import layer
    torch.manual_seed(10 + rank)                       # different initial weights and statistics per rank
    block = layer.elan(8, 16, n=1)
    for m in block.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-1, 1)
            m.num_batches_tracked.fill_(3 + rank)
    state = te.FlatState(block)
    dp = te.DataParallel(state, n_buckets=3)
    ok = not dp.reducer.host_staged and dp.world == world
    ref = [torch.empty_like(state.param) for _ in range(world)]
    dist.all_gather(ref, state.param)
    ok = ok and all(torch.equal(r, ref[0]) for r in ref)                        # constructor broadcast: parameters
    ok = ok and int(block.cbs0.bn.num_batches_tracked) == 3                     # ... and buffers, counters included
    # a forward changed the buffers locally; a backward filled the gradients
    want_buf = state.buffers + 1.0                                              # what rank 0 will hold
    state.buffers.add_(rank + 1.0)
    state.counters.add_(rank + 1)
    state.grad.copy_(torch.arange(state.n, dtype=torch.float32) * (rank + 1))
    dp.after_forward()
    for i in (2, 0, 1):
        dp.reducer.ready(i)
    ok = ok and len(dp.reducer.handles) == len(dp.reducer.bounds) == 3 and dp.pending is not None
    scale = dp.finish()
    ok = ok and dp.pending is None and dp.reducer.handles == [] and dp.collectives == 2 + 1 + 3
    want = torch.arange(state.n, dtype=torch.float32) * sum(range(1, world + 1)) / world
    ok = ok and bool(torch.allclose(state.grad * scale, want))
    ok = ok and torch.equal(state.buffers, want_buf) and int(block.cbs5.bn.num_batches_tracked) == 4
    ok = ok and torch.equal(block.cbs1.bn.running_mean, state.buffers[state.buf_names['cbs1.bn.running_mean']:][:4])
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_collectives_two_ranks_gloo_async():
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_parallel_rank, args=(r, 2, 29519, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out[0] and out[1]


def test_flat_state_stays_out_of_pickles():
    """torch.save(model) after the flat training state exists must not carry it (parameters, gradients, Adam
    moments, EMA copies), and a reloaded module must not find a stale state."""
    import io
    from objectdetection_amd import train_engine as te
    from objectdetection_amd.model.yolov7 import yolov7
    args = patterns.default_args(input_size=64, output_class=1)
    net = yolov7(args)

    def size(m):
        f = io.BytesIO()
        torch.save(m, f)
        return f.getbuffer().nbytes, f

    before, _ = size(net)
    st = net.train_state()
    st.enable_ema()
    p = next(net.parameters())
    assert te.flat_state_of(p) is st and '_odamd_flat' not in p.__dict__
    after, f = size(net)
    assert abs(after - before) < 0.02 * before, (before, after)
    f.seek(0)
    twin = torch.load(f, weights_only=False)
    assert twin._state is None
    assert all(te.flat_state_of(q) is None and '_odamd_flat' not in q.__dict__ for q in twin.parameters())
    # a block of the network finds the network's storage; a block on its own does not inherit anything stale
    assert te.flat_state_of(next(net.elan4.parameters())) is st
    assert te.flat_state_of(next(twin.elan4.parameters())) is None


def test_bucket_plan_waits_for_every_tensor_inside_a_bucket():
    """Bucket bounds cut through tensors.  A bucket may only be reduced after the LAST tensor with any element
    in it has its gradient; a tensor that straddles a bound (here a bias forced across it) belongs to both."""
    from objectdetection_amd.train_engine import GradReducer, TrainPlan

    class _Red(GradReducer):
        def __init__(self, bounds):
            self.bounds = bounds

    red = _Red([(0, 100), (100, 200), (200, 300)])
    assert red.buckets_of(0, 100) == [0]
    assert red.buckets_of(96, 8) == [0, 1]            # straddles the first bound
    assert red.buckets_of(100, 100) == [1]
    assert red.buckets_of(50, 200) == [0, 1, 2]
    assert red.buckets_of(299, 1) == [2]

    class _P:
        def __init__(self, n):
            self._n = n

        def numel(self):
            return self._n

    # backward order: tensor c (bucket 2) first, then the straddling bias b, then a
    a, b, c = _P(96), _P(8), _P(196)
    offsets = {id(a): 0, id(b): 96, id(c): 104}

    class _State:
        pass

    class _Plan:
        _bucket_plan = TrainPlan._bucket_plan

    plan = _Plan()
    plan.state = _State()
    plan.state.offsets = offsets
    plan.bwd = [('mark', (c,), 'mark', None), ('mark', (b,), 'mark', None), ('mark', (a,), 'mark', None)]
    got = plan._bucket_plan(red)
    # bucket 2 is final after c (launch 0) ... bucket 1 holds the tail of b as well -> only after launch 1
    assert got == {0: [2], 1: [1], 2: [0]}
    # with the bias marked BEFORE the tensor that fills the rest of its bucket, the bucket waits for that tensor
    plan.bwd = [('mark', (b,), 'mark', None), ('mark', (c,), 'mark', None), ('mark', (a,), 'mark', None)]
    got = plan._bucket_plan(red)
    assert got == {1: [1, 2], 2: [0]}


def test_gradient_write_modes_first_writer_stores():
    """Symbolic walk of the backward pass (train_engine.gradient_write_modes) on the op list of an elan block: the last
    consumer in forward order of every gradient range is its first writer and stores, the other consumers add; a
    gradient whose first writer is an adding kernel (pool / upsample backward) keeps its zero fill."""
    from objectdetection_amd import train_engine as te
    from objectdetection_amd.engine import PlanBuilder, BF16
    from objectdetection_amd.model import layer

    pb = PlanBuilder(2, BF16, torch.device('cpu'))
    mod = layer.elan(in_=32, out_=64, n=2)
    x = pb.from_nchw(32, 16, 16)
    out = mod._emit(pb, x)
    up = pb.upsample2(out)                      # a gradient that only an adding kernel writes
    pb.to_nchw(up)
    stores, need_zero = te.gradient_write_modes(pb.ops, [c for c, _, _ in pb.buffers])
    convs = [op for op in pb.ops if op['kind'] == 'conv']
    # the final 1x1 over the concat is the first writer of the whole concat buffer
    assert id(convs[-1]) in stores
    # cbs1 / cbs2 read the block input: exactly one of them stores (the later one in forward order), the other adds
    first_two = [op for op in convs if op['x'].buf == x.buf]
    assert len(first_two) == 2 and [id(op) in stores for op in first_two] == [False, True]
    # chain convolutions read slices of the concat that its final 1x1 has already written: they add
    chain = [op for op in convs[:-1] if op['x'].buf == convs[-1]['x'].buf]
    assert chain and not any(id(op) in stores for op in chain)
    # the upsample's input gradient is written by upsample2_bwd (adds) first: that buffer keeps its zero fill; nothing else does
    assert need_zero == {out.buf}

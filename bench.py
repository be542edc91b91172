#!/usr/bin/env python3
"""Headline benchmark: YOLOv7-s 640x640 batch-64 bf16 inference (forward + decode + filter +
NMS) on MI355X, images/sec; one process per GPU (replicas, weak scaling: inference has no
cross-GPU step).  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Synthetic data: torch.rand images under generator seed 999, random-init weights of the
architecture under torch.manual_seed(999) (run.py:75 of the reference uses the same seed).
"""
import argparse
import json
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}   # dense peaks


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=100)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--batch', type=int, default=64, help='images per GPU per step')
    p.add_argument('--size', type=int, default=640)
    p.add_argument('--classes', type=int, default=80)
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    p.add_argument('--conf', type=float, default=0.5)
    p.add_argument('--iou', type=float, default=0.5)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--workload', default='infer', choices=['infer', 'train'],
                   help="which leg `value` reports; the other one is reported in a sub-object")
    p.add_argument('--train-batch', type=int, default=32, help='training images per GPU per step')
    p.add_argument('--train-classes', type=int, default=5)
    p.add_argument('--train-steps', type=int, default=None, help='default: --steps, at most 40')
    p.add_argument('--no-train', action='store_true', help='skip the training leg')
    p.add_argument('--no-extra', action='store_true', help='skip the end-to-end and dense-candidate legs')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help='process-group backend (gloo + --share-gpu rehearses the multi-rank path on a one-GPU box)')
    p.add_argument('--share-gpu', action='store_true', help='every rank uses cuda:0 (rehearsal only)')
    p.add_argument('--splits', type=int, default=1, help='concurrent sub-batches of a step (independent images on separate streams)')
    return p.parse_args()


def cpu_baseline(model, a, margs, x_cpu):
    """The CPU restatement of the same path (oracle), bounded sample, timed on this host.  Also the
    batch-1 CPU configuration split by stage (forward / box decode / filter + NMS), for C = 80 and C = 1."""
    from oracle import network, patterns, postprocess as opost
    from objectdetection_amd.model.yolov7 import yolov7
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(cores, 64))
    used = torch.get_num_threads()

    def stages(sd, args_c, x1):
        net = network.Net(sd)
        with torch.no_grad():
            net.trunk(x1, args_c.output_class, args_c.output_layer[0])                      # warm-up
            t0 = time.perf_counter()
            heads = net.trunk(x1, args_c.output_class, args_c.output_layer[0])
            t1 = time.perf_counter()
            pred = network.deploy_sigmoid(network.decode(heads, args_c.input_size, args_c.anchor)).numpy()
            t2 = time.perf_counter()
        rows, kept = opost.predict_decode(pred[0], (a.size, a.size, 3), a.size, a.conf, a.iou)
        t3 = time.perf_counter()
        cand = len(opost.filter_indices(pred[0], a.conf)[1])
        return {'forward_us': (t1 - t0) * 1e6, 'decode_us': (t2 - t1) * 1e6, 'filter_nms_us': (t3 - t2) * 1e6,
                'candidates': cand, 'kept': len(kept)}

    sd = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
    c80 = stages(sd, margs, x_cpu[:1])
    args1 = patterns.default_args(input_size=a.size, output_class=1)
    torch.manual_seed(999)
    sd1 = {k: v.detach().float().clone() for k, v in yolov7(args1).state_dict().items()}
    c1 = stages(sd1, args1, x_cpu[:1])
    t_img = (c80['forward_us'] + c80['decode_us'] + c80['filter_nms_us']) * 1e-6
    return {'value': 1.0 / t_img, 'unit': 'images/sec', 'cores': used, 'kind': 'port',
            'sample': f'1 image, batch 1, float32: forward on {used} torch threads + numpy decode/NMS on 1 core '
                      '(the restated loop indexes boxes; the reference loop, predict.py:86-90, copies whole rows per '
                      'iteration and is several times slower)',
            'forward_ms_per_image': c80['forward_us'] * 1e-3,
            'postprocess_s_per_image': (c80['decode_us'] + c80['filter_nms_us']) * 1e-6,
            'batch1_cpu_by_stage_us': {'C80': c80, 'C1': c1}}


def end_to_end_leg(model, a, margs, x_cpu, steps=12):
    """The same step fed from the HOST and read back to the host: page-locked uint8 HWC images -> H2D ->
    / 255 and HWC -> CHW on the device (one launch per batch) -> network -> decode / NMS -> kept rows of every image
    in page-locked host memory (what predict.py:53-58 hands back).  Reported beside `value`, never as `value`.
    Batch i+1 is uploaded and enqueued before the rows of batch i are read back; two device input buffers alternate
    (the network of batch i may still read its input while batch i+1 is being prepared)."""
    import ctypes as C
    from objectdetection_amd._lib import lib, check
    from objectdetection_amd.postprocess import HostStager
    from objectdetection_amd.predict import detect_pipeline
    dev = next(model.parameters()).device
    img8 = (x_cpu.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous().pin_memory()
    xs = [torch.empty((a.batch, 3, a.size, a.size), dtype=torch.float32, device=dev) for _ in range(2)]
    d8s = [torch.empty(img8.shape, dtype=torch.uint8, device=dev) for _ in range(2)]
    pipe = detect_pipeline(model, a.conf, a.iou, anchor=margs.anchor, input_size=a.size, reuse_outputs=3)
    stager = HostStager(2)
    turn = [0]

    def step():
        i = turn[0]
        turn[0] ^= 1
        d8s[i].copy_(img8, non_blocking=True)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.odamd_preprocess_u8_batch(d8s[i].data_ptr(), a.batch, a.size, a.size, xs[i].data_ptr(), a.size, st),
              'preprocess_u8_batch')
        return pipe.submit(xs[i])

    def collect(res):
        res.done.synchronize()
        return sum(0 if r is None else len(r) for r, _ in res.to_host(stager, copy=False))

    for i in range(3):
        collect(step())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows_out, pending = 0, None
    for i in range(steps):
        cur = step()
        if pending is not None:                 # read batch i-1 back while batch i runs
            rows_out += collect(pending)
        pending = cur
    rows_out += collect(pending)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'images_per_sec': a.batch * steps / dt, 'ms_per_step': dt / steps * 1e3, 'steps': steps,
            'h2d_bytes_per_step': int(img8.numel()), 'kept_rows_per_step': rows_out / steps,
            'd2h_bytes_per_step': int(rows_out / steps * (margs.output_class + 5 + 1) * 4),
            'note': 'uint8 HWC host images in, kept rows in page-locked host memory out; H2D + device-side /255 and '
                    'transpose + D2H inside the timed region'}


def dense_leg(a, dev, steps=3):
    """Dense-candidate configuration: 1280 x 1280, batch 16, 80 classes, confidence 0.001 (random init: nearly
    every one of the 100 800 rows per image is a candidate)."""
    from objectdetection_amd import postprocess
    from objectdetection_amd.model.yolov7 import yolov7
    from oracle import patterns
    size, batch = 1280, 16
    margs = patterns.default_args(input_size=size, output_class=80)
    torch.manual_seed(999)
    model = yolov7(margs).to(dev).eval()
    x = torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(999)).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    net_ms, post_ms = [], []
    for i in range(steps + 1):
        ev[0].record()
        heads = model.head_tensors(x)
        ev[1].record()
        res = postprocess.postprocess_heads(heads, size, margs.anchor, 0.001, 0.5)
        ev[2].record()
        torch.cuda.synchronize()
        if i:
            net_ms.append(ev[0].elapsed_time(ev[1]))
            post_ms.append(ev[1].elapsed_time(ev[2]))
    cand, kept = res.cand.float().mean().item(), res.count.float().mean().item()
    n, p = sorted(net_ms)[len(net_ms) // 2], sorted(post_ms)[len(post_ms) // 2]
    del model, x, heads, res
    return {'workload': 'YOLOv7-s 1280x1280 batch 16 bf16, C=80, conf 0.001, iou 0.5', 'network_ms_per_batch': n,
            'postprocess_ms_per_batch': p, 'ms_per_batch': n + p, 'images_per_sec': batch / ((n + p) * 1e-3),
            'decode_nms_us_per_image': p * 1e3 / batch, 'candidates_per_image': cand, 'kept_per_image': kept}


def preset_leg(a, dev, preset='l', batch=64, steps=5):
    """The network alone for a wider preset (model/yolov7.py:9-12: 'l' = base width 64, three 3x3 convolutions per ELAN
    branch: 416 GFLOP per 640 x 640 image, ~500 FLOP per byte, i.e. the configuration whose convolutions are bound by
    the matrix cores and not by HBM).  Whole forwards timed with HIP events on the launch stream, then one pass of
    per-launch events for the family table and the per-layer bound min(MFMA, HBM)."""
    from objectdetection_amd import engine
    from objectdetection_amd.model.yolov7 import yolov7
    from oracle import patterns
    margs = patterns.default_args(input_size=a.size, output_class=a.classes, model_type=preset)
    torch.manual_seed(999)
    model = yolov7(margs).to(dev).eval()
    x = torch.rand(batch, 3, a.size, a.size, generator=torch.Generator().manual_seed(999)).to(dev)
    for _ in range(2):
        model.head_tensors(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        model.head_tensors(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    prof = model._plan(x).profile(x, repeats=3)
    peak, hbm = MFMA_PEAK_TFLOPS[a.dtype] * 1e12, HBM_PEAK_GBS * 1e9
    flops = sum(m['flops'] for m in prof)
    ideal = sum(max(m['flops'] / peak, m['bytes'] / hbm) for m in prof)
    layer_ms = sum(m['ms'] for m in prof)
    bound = [m for m in prof if m['flops'] / peak > m['bytes'] / hbm]
    b_ms, b_fl = sum(m['ms'] for m in bound), sum(m['flops'] for m in bound)
    fam = {}
    for m in prof:
        f = fam.setdefault(m['kernel'], {'ms': 0.0, 'flops': 0.0, 'launches': 0})
        f['ms'] += m['ms']
        f['flops'] += m['flops']
        f['launches'] += 1
    params = sum(p.numel() for p in model.parameters())
    del model, x
    return {'workload': f'YOLOv7-{preset} {a.size}x{a.size} batch {batch} {a.dtype}, network only (C={a.classes})',
            'params': params, 'ms_per_batch': ms, 'images_per_sec': batch / (ms * 1e-3),
            'conv_gflop_per_image': flops / batch / 1e9, 'network_tflops': flops / (ms * 1e-3) / 1e12,
            'frac_of_mfma_peak': flops / (ms * 1e-3) / peak,
            'frac_of_per_layer_bound': ideal * 1e3 / layer_ms,
            'compute_bound_layers': {'launches': len(bound), 'ms': b_ms, 'tflops': b_fl / max(b_ms, 1e-9) * 1e-9,
                                     'frac_of_mfma_peak': b_fl / max(b_ms, 1e-9) * 1e3 / peak},
            'kernel_families': {k: {'ms': round(v['ms'], 4), 'launches': v['launches'],
                                    'tflops': round(v['flops'] / max(v['ms'], 1e-9) * 1e-9, 1)}
                                for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])}}


def train_leg(a, dev, rank, world, dist, steps, warmup, amp=False):
    """Training step (the single-GPU and data-parallel training configurations): forward (batch-statistics BN) + loss + backward + bucketed
    gradient all-reduce (RCCL when world > 1) + fused Adam/EMA, float32 or (amp) bf16 mixed precision, `--train-batch` synthetic
    images per GPU with 8 synthetic boxes each.  Weak scaling: global batch = world * train-batch."""
    from objectdetection_amd.train_class import train_class, synthetic_dataset
    from oracle import patterns
    targs = patterns.default_args(input_size=a.size, output_class=a.train_classes)
    targs.__dict__.update(device=f'cuda:{dev.index}', batch=a.train_batch, epoch=1, lr_start=0.001, lr_end_ratio=0.1,
                          lr_end_epoch=100, warmup_ratio=0.01, regularization='L2', r_value=0.0005, ema=True,
                          distributed=world > 1 and not os.environ.get('BENCH_NO_REDUCER'), local_rank=rank, weight_path='', weight_again=True, amp=amp,
                          print_info=False, save_path=None)
    torch.manual_seed(999)
    ds = synthetic_dataset(targs, a.train_batch * world, seed=999)
    batch = ds.collate_fn([ds[rank * a.train_batch + i] for i in range(a.train_batch)])
    note('dataset ready')
    tc = train_class(targs, dataloader=[batch] * 1000)
    note('train_class built')
    tc.optimizer.ema = True                      # every rank carries the EMA so that ranks do equal work
    tc.model.train_state().enable_ema()
    tc.model.train()
    image = batch[0].pin_memory().to(dev, non_blocking=True)     # page-locked source: an asynchronous copy on the
    torch.cuda.current_stream().synchronize()                    # current stream, not a blocking default-stream one
    screen, expend = batch[1], batch[2]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first = None
    for i in range(warmup):
        l = tc.train_step(image, screen, expend)
        if first is None:
            first = l.tolist()
    fence()
    note('train warm-up done')
    # The step is ~1000 launches enqueued from Python, close to the device time: a full collection of the interpreter's
    # cyclic garbage collector over everything the earlier legs left alive (plans, descriptors) stalls the enqueue for
    # milliseconds.  Objects alive now are moved out of the collector's sight for the timed steps (gc.freeze is what a
    # long-running training process does after set-up); nothing is skipped: the collector keeps running on new objects.
    if not os.environ.get('BENCH_NO_GC_FREEZE'):
        gc.collect()
        gc.freeze()
    t0 = time.perf_counter()
    for _ in range(steps):
        l = tc.train_step(image, screen, expend)
    fence()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last = l.tolist()
    if not all(v == v and abs(v) < 1e30 for v in last):
        raise RuntimeError(f'training leg diverged (amp={amp}): losses {last}')
    # convolution work of a step: forward + data gradient + weight gradient = 3 x the forward count of the
    # plan (2 FLOP per multiply-add; the first layer has no data gradient, which this overstates by < 1 %)
    from objectdetection_amd import engine as _engine
    pb = _engine.PlanBuilder(1, _engine.F32, dev)      # lowering only: no buffers are allocated
    tc.model._emit(pb, (a.size, a.size))
    fwd_flops = 0.0                                     # per image
    for op in pb.ops:
        if op['kind'] == 'first':
            fwd_flops += 2.0 * op['y'].h * op['y'].w * op['mod'].conv.out_channels * 27
        elif op['kind'] == 'conv':
            cv = op['mod'].conv
            fwd_flops += 2.0 * op['y'].h * op['y'].w * cv.out_channels * op['k'] ** 2 * cv.in_channels
        elif op['kind'] == 'head':
            cv = op['mod'].output
            fwd_flops += 2.0 * op['x'].h * op['x'].w * cv.out_channels * cv.in_channels
    ips = world * a.train_batch * steps / dt
    tflops = 3.0 * fwd_flops * ips / 1e12
    peak = MFMA_PEAK_TFLOPS['bf16' if amp else 'f32']
    return {'images_per_sec': world * a.train_batch * steps / dt, 'ms_per_step': dt / steps * 1e3,
            'conv_tflops': tflops, 'mfma_peak_tflops': peak * world, 'frac_of_mfma_peak': tflops / (peak * world),
            'global_batch': world * a.train_batch, 'dtype': 'bf16 (f32 accumulate / statistics / master weights)' if amp else 'f32',
            'classes': a.train_classes, 'steps': steps,
            'parallelism': f'dp{world}' + (' (RCCL all-reduce, 3 buckets behind backward)' if world > 1 else ''),
            'positives_per_step': int(sum(len(s) for s in screen)),
            'loss_first': first, 'loss_last': last, 'params': tc.model.train_state().n}, tc


def note(msg):
    """Progress line on stderr (BENCH_DEBUG=1): shows which phase a multi-rank run is in."""
    if os.environ.get('BENCH_DEBUG'):
        print(f"[bench rank {os.environ.get('RANK', 0)} +{time.perf_counter():.1f}s] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` run plainly: this process becomes a launcher that never touches the GPU.  It starts
    `python -m torch.distributed.run --nproc-per-node N` on this same script (one rank per GPU, the launch line of
    run.py:15 of the reference), lets rank 0's JSON line pass through and returns the launcher's exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < n and '--share-gpu' not in sys.argv:
        print(f'bench.py: --gpus {n} asked for, {have} GPU(s) visible', file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if os.environ.get('BENCH_DEBUG'):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get('BENCH_DEBUG_DUMP_S', '90')), exit=False)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != a.gpus and rank == 0:
        print(f'# note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE', file=sys.stderr)
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')

    from objectdetection_amd import engine, postprocess
    from objectdetection_amd.model.yolov7 import yolov7
    from oracle import patterns

    margs = patterns.default_args(input_size=a.size, output_class=a.classes)
    torch.manual_seed(999)
    model = yolov7(margs).to(dev).eval()
    engine.set_compute_dtype(a.dtype)
    gen = torch.Generator().manual_seed(999 + rank)
    x_cpu = torch.rand(a.batch, 3, a.size, a.size, generator=gen)
    x = x_cpu.to(dev)

    # two HIP streams: the network of step i+1 overlaps the decode/NMS of step i; every step's
    # network AND post-processing complete inside the timed region (the closing fence waits for both)
    from objectdetection_amd.predict import detect_pipeline
    pipe = detect_pipeline(model, a.conf, a.iou, anchor=margs.anchor, input_size=a.size, splits=a.splits)

    def step():
        return pipe.submit(x)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    note('process group up, model built')
    for _ in range(a.warmup):
        res = step()
    fence()
    note('warm-up done')
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * a.batch * a.steps / dt
    note(f'inference timed: {dt:.3f}s')

    out = None
    if rank == 0:
        # ---- per-kernel device times (HIP events on the launch stream) -> roofline of the dominant kernel
        plan = model._plan(x)
        prof = plan.profile(x, repeats=5)
        fam = {}
        for m in prof:
            f = fam.setdefault(m['kernel'], {'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'launches': 0})
            f['ms'] += m['ms']
            f['flops'] += m['flops']
            f['bytes'] += m['bytes']
            f['launches'] += 1
        net_ms = sum(m['ms'] for m in prof)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # whole forwards back to back (no event between the launches: the per-launch events above cost about 5 us each)
        for _ in range(3):
            model.head_tensors(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            model.head_tensors(x)
        e1.record()
        torch.cuda.synchronize()
        net_b2b_ms = e0.elapsed_time(e1) / 20
        heads = model.head_tensors(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            res = postprocess.postprocess_heads(heads, a.size, margs.anchor, a.conf, a.iou)
        e1.record()
        torch.cuda.synchronize()
        post_ms = e0.elapsed_time(e1) / 5
        dom_name = max((k for k in fam if fam[k]['flops'] > 0), key=lambda k: fam[k]['ms'])
        dom = fam[dom_name]
        t_mfma = dom['flops'] / (MFMA_PEAK_TFLOPS[a.dtype] * 1e12)
        t_hbm = dom['bytes'] / (HBM_PEAK_GBS * 1e9)
        if t_mfma >= t_hbm:
            ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'achieved': ach, 'peak': MFMA_PEAK_TFLOPS[a.dtype], 'unit': 'TFLOP/s'}
        else:
            ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
            roof = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'}
        # HBM bytes per launch of this kernel family from the committed rocprofv3 counter passes
        # (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); null when the file is absent
        traffic = None
        try:
            for name in ('r3_pmc_traffic.json', 'r2_pmc_traffic.json', 'r1_pmc_traffic.json'):
                path = os.path.join(ROOT, 'profiles', name)
                if os.path.exists(path):
                    with open(path) as fh:
                        traffic = json.load(fh)['families'][dom_name]['traffic_bytes_per_launch']
                    break
        except (OSError, KeyError, ValueError):
            pass
        roof.update({'frac': roof['achieved'] / roof['peak'], 'traffic': traffic, 'kernel': dom_name,
                     'launches_per_step': dom['launches'], 'avg_launch_ms': dom['ms'] / dom['launches'],
                     'algorithmic_flops_per_step': dom['flops'], 'algorithmic_bytes_per_step': dom['bytes']})
        conv_flops = sum(f['flops'] for f in fam.values())
        cnt = res.count.cpu()
        cand = res.cand.cpu()
        out = {
            'metric': 'images/sec @640^2 inference (forward + decode + NMS)', 'value': value, 'unit': 'images/sec',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'YOLOv7-s {a.size}x{a.size} batch {a.batch} {a.dtype} inference + GPU decode/NMS '
                                   f'(headline configuration 2 of 5); C={a.classes}, conf {a.conf}, iou {a.iou}',
                       'global_batch': a.batch * world, 'parallelism': f'replicas x{world}',
                       'pipeline': 'three HIP streams: large-map launches of batch i+1, small-map launches of batch i, decode/NMS of batch i-1 or i'},
            'roofline': roof,
            'network_ms_per_step': net_ms, 'network_ms_back_to_back': net_b2b_ms,
            'network_frac_of_per_layer_bound': sum(max(m['flops'] / (MFMA_PEAK_TFLOPS[a.dtype] * 1e12), m['bytes'] / (HBM_PEAK_GBS * 1e9))
                                                   for m in prof) * 1e3 / net_b2b_ms,
            'postprocess_ms_per_step': post_ms,
            'decode_nms_us_per_image': post_ms * 1e3 / a.batch,
            'network_tflops': conv_flops / (net_b2b_ms * 1e-3) / 1e12,
            'candidates_per_image': float(cand.float().mean()), 'kept_per_image': float(cnt.float().mean()),
            'kernel_families_ms': {k: round(v['ms'], 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])},
        }
        def leg(name, fn):
            # a side leg that fails (say, out of memory on a partitioned device) is reported in its own object and never
            # costs the headline line
            try:
                out[name] = fn()
            except Exception as e:                       # noqa: BLE001
                out[name] = {'error': f'{type(e).__name__}: {e}'}
                note(f'{name} leg failed: {e}')
            torch.cuda.empty_cache()
        if world == 1 and not a.no_extra:
            leg('end_to_end', lambda: end_to_end_leg(model, a, margs, x_cpu))
            leg('dense', lambda: dense_leg(a, dev))
            leg('preset_l', lambda: preset_leg(a, dev, 'l', batch=a.batch))
        if world == 1 and not a.no_cpu_baseline:
            leg('cpu_baseline', lambda: cpu_baseline(model, a, margs, x_cpu))

    if not a.no_train:
        # The training leg must never cost the inference line: if a collective stalls, rank 0 prints
        # what it has and every rank leaves.
        import threading

        def bail():
            # the inference line is still printed, but a stalled training leg is a failure of the run
            if rank == 0:
                out['train'] = {'error': 'training leg did not finish within 240 s'}
                print(json.dumps(out), flush=True)
            os._exit(3)
        guard = threading.Timer(240.0, bail)
        guard.daemon = True
        guard.start()
        del model, x
        tsteps = a.train_steps if a.train_steps is not None else min(a.steps, 40)
        try:
            note('training leg f32')
            train, tc_ = train_leg(a, dev, rank, world, dist, tsteps, max(a.warmup, 2))
            note('training leg bf16')
            del tc_
            torch.cuda.empty_cache()
            train_amp, tc_ = train_leg(a, dev, rank, world, dist, tsteps, max(a.warmup, 2), amp=True)
            del tc_
        except Exception as e:                           # noqa: BLE001
            # like a stalled leg: the inference line is printed, the run fails
            if rank == 0:
                out['train'] = {'error': f'{type(e).__name__}: {e}'}
                print(json.dumps(out), flush=True)
            os._exit(3)
        guard.cancel()
        if rank == 0:
            out['train'] = train                 # float32: the reference default (--amp False)
            out['train_amp'] = train_amp         # bf16 mixed precision: the --amp True mode
            if a.workload == 'train':
                infer = {k: out[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'config')}
                out.update({'metric': 'images/sec @640^2 training step (fwd + loss + bwd + all-reduce + Adam/EMA)',
                            'value': train['images_per_sec'], 'ms_per_step': train['ms_per_step'], 'dtype': 'f32',
                            'steps': tsteps,
                            'config': {'workload': f'YOLOv7-s {a.size}x{a.size} training step, batch {a.train_batch} per GPU, '
                                                   f'synthetic labels (training configurations 3 and 4 of 5); C={a.train_classes}',
                                       'global_batch': train['global_batch'], 'parallelism': train['parallelism']},
                            'infer': infer})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()

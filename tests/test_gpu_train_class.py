"""train_class mirror end to end on the GPU: one epoch on synthetic data, first-step losses against
the CPU oracle, optimizer/EMA wiring against the oracle's Adam on the GPU's own gradients."""
import argparse

import numpy as np
import pytest
import torch

from oracle import network, patterns, training as otrain

pytestmark = pytest.mark.gpu


def _args(**kw):
    a = patterns.default_args(input_size=128, output_class=5)
    a.__dict__.update(device='cuda', batch=4, epoch=1, lr_start=0.001, lr_end_ratio=0.1, lr_end_epoch=10,
                      warmup_ratio=0.01, regularization='L2', r_value=0.0005, ema=True, distributed=False,
                      local_rank=0, weight_path='', weight_again=True, print_info=False, synthetic_images=8,
                      save_path=None)
    a.__dict__.update(kw)
    return a


def test_first_step_losses_and_optimizer_wiring():
    from objectdetection_amd.train_class import train_class, synthetic_dataset
    args = _args()
    torch.manual_seed(3)
    tc = train_class(args)
    model = tc.model
    patterns.fill_state_dict(model)          # parameters are already views of the flat buffers
    st = model.train_state()
    st.ema_param.copy_(st.param)
    st.ema_buffers.copy_(st.buffers)
    ds = synthetic_dataset(args, 8)
    image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(4)])
    assert sum(len(s) for s in screen) > 20
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref_pred = network.yolov7_forward(sd, image, args, training=True)
        ref = [float(v) for v in otrain.detection_loss(ref_pred, screen, expend, args.label_smooth, args.output_class)]
    p0 = st.param.clone()
    lr0 = tc.optimizer.param_groups[0]['lr']
    assert lr0 == otrain.LrSchedule(0.001, 0.1, 10, 0.01, len(tc.train_dataloader)).step()
    model.train()
    losses = tc.train_step(image, screen, expend).cpu().numpy()
    np.testing.assert_allclose(losses, ref, rtol=2e-3)
    # Adam + EMA: replay the oracle's update on the gradients the GPU produced
    g = st.grad.clone().cpu()
    p, m, v = p0.cpu().clone(), torch.zeros_like(g), torch.zeros_like(g)
    otrain.adam_step(p, g, m, v, 1, lr0, weight_decay=args.r_value)
    np.testing.assert_allclose(st.param.cpu().numpy(), p.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(st.exp_avg.cpu().numpy(), m.numpy(), rtol=1e-4, atol=1e-9)
    d = otrain.ema_decay(1)
    np.testing.assert_allclose(st.ema_param.cpu().numpy(), (d * p0.cpu() + (1 - d) * p).numpy(), rtol=1e-5, atol=1e-8)
    assert tc.optimizer.param_groups[0]['lr'] != lr0      # schedule advanced for the next step
    assert float(g.abs().sum()) > 0
    # the reference style loop (model -> loss_class -> backward) leaves the same gradients
    tc.optimizer.zero_grad()
    pred = model(image.cuda())
    tc.loss(pred, screen, expend)[0].backward()
    assert float(st.grad.abs().sum()) > 0
    twin = tc.ema.ema_model
    assert not twin.training and len(twin.state_dict()) == len(model.state_dict())


def test_one_epoch_runs_and_loss_is_finite():
    from objectdetection_amd.train_class import train_class
    torch.manual_seed(999)
    tc = train_class(_args(epoch=2))
    hist = tc.train()
    assert len(hist) == 2 and all(np.isfinite(h).all() for h in hist)
    assert hist[1][0] < hist[0][0] * 1.5


def test_loss_with_duplicate_rows_accumulates_gradient():
    """The reference's label assignment can list a grid row twice; the gradient of such a row is the
    sum of both occurrences (autograd of an index gather)."""
    import ctypes as C
    from objectdetection_amd._lib import lib, check
    b, r, k = 1, 40, 7
    pred = patterns.wave('dup.pred', (b, r, k), 2.0)
    pred[:, :, 0:2] = pred[:, :, 0:2] * 20 + 60
    pred[:, :, 2:4] = pred[:, :, 2:4].abs() * 15 + 5
    screen = [torch.tensor([3, 3, 9])]
    lab = patterns.wave('dup.lab', (3, k), 0.45, 0.5)
    lab[:, 0:2] = lab[:, 0:2] * 40 + 40
    lab[:, 2:4] = lab[:, 2:4] * 30 + 6
    lab[1] = lab[0]
    p = pred.clone().requires_grad_(True)
    vals = otrain.detection_loss(p, screen, lab, 0.01, 2)
    vals[0].backward()
    pd = pred.cuda().contiguous()
    dp = torch.empty_like(pd)
    out = torch.zeros(4, device='cuda')
    ws = torch.empty(int(lib.odamd_loss_workspace_bytes()), dtype=torch.uint8, device='cuda')
    idx = torch.tensor([3, 3, 9], dtype=torch.int32).cuda()
    img = torch.zeros(3, dtype=torch.int32).cuda()
    lb = lab.cuda().contiguous()
    check(lib.odamd_detection_loss(pd.data_ptr(), dp.data_ptr(), idx.data_ptr(), img.data_ptr(), lb.data_ptr(), b, r, k,
                                   3, 0.01, out.data_ptr(), ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), [float(v) for v in vals], rtol=2e-5)
    np.testing.assert_allclose(dp.cpu().numpy(), p.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_amp_bf16_step_tracks_float32_oracle():
    """--amp True: bf16 mixed-precision step.  First-step losses against the float32 CPU oracle
    (5e-2: 57 bf16 layers with batch statistics), optimizer state finite, loss falls over an epoch."""
    from objectdetection_amd.train_class import train_class, synthetic_dataset
    args = _args(amp=True, epoch=3)
    torch.manual_seed(3)
    tc = train_class(args)
    model = tc.model
    assert model.train_dtype == 'bf16'
    ds = synthetic_dataset(args, 8)
    image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(4)])
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref_pred = network.yolov7_forward(sd, image, args, training=True)
        ref = [float(v) for v in otrain.detection_loss(ref_pred, screen, expend, args.label_smooth, args.output_class)]
    model.train()
    losses = tc.train_step(image, screen, expend).cpu().numpy()
    np.testing.assert_allclose(losses, ref, rtol=5e-2)
    st = model.train_state()
    assert torch.isfinite(st.param).all() and torch.isfinite(st.grad).all() and float(st.grad.abs().sum()) > 0
    hist = tc.train()
    assert all(np.isfinite(h).all() for h in hist) and hist[-1][0] < hist[0][0]


def test_device_label_assignment_matches_oracle_and_feeds_loss():
    """odamd_assign_labels against the literal restatement of the reference loop (oracle/labels.py):
    same rows in the same order, same label rows; then the loss through the device-side count equals the
    loss through host lists."""
    from oracle import labels as olabels
    from objectdetection_amd.train_class import device_labels, train_class
    args = _args()
    g = torch.Generator().manual_seed(77)
    k = 5 + args.output_class
    labs = []
    for n in (6, 0, 9, 3):
        frame = torch.cat([torch.rand(n, 2, generator=g) * 100 + 14, torch.rand(n, 2, generator=g) * 60 + 8], 1)
        rest = torch.rand(n, k - 4, generator=g)
        labs.append(torch.cat([frame, rest], 1).float())
    labs[2][4] = labs[2][1]            # two labels hitting the same rows: duplicates, last label wins
    labs[2][4, 4:] += 0.125
    dl = device_labels(args, lmax=16)
    res = dl(labs)
    torch.cuda.synchronize()
    got_rows, got_lab = res.per_image(len(labs))
    lo, hi = olabels.ranges(args)
    want_rows, want_lab = [], []
    for lab in labs:
        if len(lab) == 0:
            want_rows.append(np.zeros(0, np.int64))
            want_lab.append(np.zeros((0, k), np.float32))
            continue
        rows, expend = olabels.assign(lab, lo[:, 0:4], hi[:, 0:4])
        want_rows.append(rows.numpy())
        want_lab.append(expend.numpy().reshape(-1, k))
    assert sum(len(r) for r in want_rows) > 40
    for b in range(len(labs)):
        np.testing.assert_array_equal(got_rows[b].numpy(), want_rows[b])
    np.testing.assert_array_equal(got_lab.numpy(), np.concatenate(want_lab))
    # loss through the device-side count == loss through host lists
    tc = train_class(args)
    tc.model.train()
    x = torch.rand(4, 3, 128, 128, generator=g).cuda()
    pred = tc.model(x)
    a = [float(v) for v in tc.loss(pred, res)]
    pred = tc.model(x)
    b_ = [float(v) for v in tc.loss(pred, [r for r in got_rows], got_lab)]
    np.testing.assert_allclose(a, b_, rtol=1e-6)


@pytest.mark.parametrize('preset,amp', [('m', False), ('m', True), ('l', True)])
def test_other_presets_train_step(preset, amp):
    """Width / depth presets 'm' and 'l' (model/yolov7.py:9-12): first-step losses against the CPU
    oracle and a finite second step."""
    from objectdetection_amd.train_class import train_class, synthetic_dataset
    args = _args(amp=amp, model_type=preset, batch=2, synthetic_images=4)
    torch.manual_seed(5)
    tc = train_class(args)
    ds = synthetic_dataset(args, 4)
    image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(2)])
    sd = {k: v.detach().cpu().clone() for k, v in tc.model.state_dict().items()}
    with torch.no_grad():
        ref_pred = network.yolov7_forward(sd, image, args, training=True)
        ref = [float(v) for v in otrain.detection_loss(ref_pred, screen, expend, args.label_smooth, args.output_class)]
    tc.model.train()
    losses = tc.train_step(image, screen, expend).cpu().numpy()
    np.testing.assert_allclose(losses, ref, rtol=5e-2 if amp else 2e-3)
    again = tc.train_step(image, screen, expend).cpu().numpy()
    assert np.isfinite(again).all()
    assert torch.isfinite(tc.model.train_state().param).all()


def test_cached_eval_plan_follows_training_steps():
    """eval forward -> train_step -> eval forward: the optimizer and the train-mode BatchNorm write
    parameters and running buffers through raw pointers, which tensor `_version`s do not see; the cached
    inference plan must still run on the updated weights (compared with the oracle on the new state_dict)."""
    from objectdetection_amd.train_class import train_class, synthetic_dataset
    args = _args(lr_start=0.01)
    torch.manual_seed(11)
    tc = train_class(args)
    model = tc.model
    patterns.fill_state_dict(model)
    ds = synthetic_dataset(args, 8)
    image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(4)])
    x = image.cuda()

    def oracle_eval():
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            return network.yolov7_forward(sd, image, args).numpy()

    def check():
        model.eval()
        with torch.no_grad():
            got = model(x).cpu().numpy()
        ref = oracle_eval()
        for sl in (slice(0, 4), slice(4, None)):
            err = np.abs(got[..., sl] - ref[..., sl]).max() / np.abs(ref[..., sl]).max()
            assert err < 1e-4, err
        return got

    before = check()
    plans = dict(model._plans)
    model.train()
    for _ in range(3):
        tc.train_step(image, screen, expend)
    after = check()
    assert model._plans.keys() == plans.keys() and all(model._plans[k] is plans[k] for k in plans)   # same cached plan
    assert np.abs(after - before).max() > 1e-3 * np.abs(before).max()       # the weights did move

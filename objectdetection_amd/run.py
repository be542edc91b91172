"""Training CLI with the flags of the reference's run.py (run.py:18-59).

    python -m objectdetection_amd.run --batch 32
    python -m torch.distributed.run --nproc-per-node N -m objectdetection_amd.run --distributed True

The rank is taken from --local_rank, --local-rank or the LOCAL_RANK variable, so both the old
torch.distributed.launch and torchrun work.  Without a dataset directory the synthetic dataset is used.
"""
import argparse
import os

import torch


def _bool(v):
    return str(v).lower() not in ('false', '0', 'no', '')


def build_parser():
    p = argparse.ArgumentParser(description='object detection training (MI355X build)')
    p.add_argument('--log', default=True, type=_bool)
    p.add_argument('--tqdm', default=False, type=_bool)
    p.add_argument('--print_info', default=True, type=_bool)
    p.add_argument('--wandb', default=False, type=_bool)
    p.add_argument('--data_path', default='dataset', type=str)
    p.add_argument('--weight_path', default='last.pt', type=str)
    p.add_argument('--weight_again', default=True, type=_bool)
    p.add_argument('--model', default='yolov7', type=str)
    p.add_argument('--model_type', default='s', type=str)
    p.add_argument('--save_epoch', default=5, type=int)
    p.add_argument('--save_path', default='last.pt', type=str)
    p.add_argument('--input_size', default=640, type=int)
    p.add_argument('--output_class', default=5, type=int)
    p.add_argument('--epoch', default=100, type=int)
    p.add_argument('--batch', default=64, type=int, help='per-rank batch, as in train_class.py:145')
    p.add_argument('--warmup_ratio', default=0.01, type=float)
    p.add_argument('--lr_start', default=0.001, type=float)
    p.add_argument('--lr_end_ratio', default=0.1, type=float)
    p.add_argument('--lr_end_epoch', default=100, type=int)
    p.add_argument('--regularization', default='L2', type=str)
    p.add_argument('--r_value', default=0.0005, type=float)
    p.add_argument('--device', default='cuda', type=str)
    p.add_argument('--num_worker', default=0, type=int)
    p.add_argument('--ema', default=True, type=_bool)
    p.add_argument('--distributed', default=False, type=_bool)
    p.add_argument('--local_rank', '--local-rank', default=int(os.environ.get('LOCAL_RANK', 0)), type=int)
    p.add_argument('--label_smooth', default=0.01, type=float)
    p.add_argument('--synthetic_images', default=512, type=int)
    # flags of the reference kept for command-line compatibility
    p.add_argument('--amp', default=False, type=_bool, help='mixed precision: the bf16 training step (run.py:47)')
    p.add_argument('--latch', default=True, type=_bool)
    p.add_argument('--noise', default=0.5, type=float)
    p.add_argument('--save_best', default='best.pt', type=str)
    p.add_argument('--prune_weight_path', default='prune_source.pt', type=str)
    p.add_argument('--prune_ratio', default=0.8, type=float)
    p.add_argument('--confidence_threshold', default=0.5, type=float)
    p.add_argument('--iou_threshold', default=0.5, type=float)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.output_size = (args.input_size // 8, args.input_size // 16, args.input_size // 32)
    args.output_layer = (3, 3, 3)
    args.anchor = (((0.02, 0.03), (0.03, 0.06), (0.06, 0.04)), ((0.06, 0.11), (0.11, 0.08), (0.11, 0.22)),
                   ((0.22, 0.17), (0.30, 0.40), (0.72, 0.62)))
    args.device_number = max(torch.cuda.device_count(), 1)
    if args.distributed:
        torch.cuda.set_device(args.local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        args.device = torch.device('cuda', args.local_rank)
        torch.distributed.init_process_group(backend='nccl', device_id=args.device)
    torch.manual_seed(999)
    torch.cuda.manual_seed_all(999)
    from .train_class import train_class
    train_class(args).train()
    if args.distributed:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

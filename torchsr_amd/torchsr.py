"""``torchsr`` command line -- interface of torchsr/torchsr.py:157-270 (same sub-commands, flags
and defaults; ``python -m torchsr_amd.torchsr train ...``).

Differences, all deliberate: the process group backend stays ``nccl`` (= RCCL over xGMI on ROCm)
but falls back to ``gloo`` when no GPU is visible so the launcher itself is testable; the ``test``
sub-command works (the reference's crashes on ``args.seed``); ``--epochs < 8`` no
longer divides by zero; ``--train-dir synthetic:N`` trains on N seeded random crops.
"""
import os
import random
import sys
from argparse import ArgumentParser, ArgumentTypeError, Namespace
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist

from torchsr_amd.constants import BATCH_SIZE, EPOCHS, MODEL, PRE_EPOCHS, TRAIN_DIR
from torchsr_amd.models import MODELS, select_test_model, select_trainer_model

try:  # optional, as in the reference (torchsr.py:26-29)
    import wandb
except ImportError:
    wandb = None


def positive_integer(value: str) -> int:
    """torchsr.py:36-66."""
    try:
        int_value = int(value)
    except (TypeError, ValueError):
        raise ArgumentTypeError(f'invalid int value: \'{value}\'')
    if int_value < 1:
        raise ArgumentTypeError('value must be a positive integer!')
    return int_value


def get_device(args: Namespace) -> torch.device:
    """torchsr.py:69-98."""
    count = torch.cuda.device_count()
    if args.local_world_size > count:
        print('More processes per node requested than GPUs found')
        print('Assuming CPU-only mode... (torchsr_amd has no CPU compute path: train / test need an MI355X and will stop;'
              ' the CPU restatement of the reference lives in oracle/ for tests only)')
        return torch.device('cpu')
    if count < 1 or not torch.cuda.is_available():
        print('No GPUs found')
        print('Running in CPU-only mode... (torchsr_amd has no CPU compute path: train / test need an MI355X and will stop;'
              ' the CPU restatement of the reference lives in oracle/ for tests only)')
        return torch.device('cpu')
    return torch.device('cuda')


def distributed_params(args: Namespace) -> Tuple[Namespace, bool]:
    """torchsr.py:101-154: torchrun env, then Slurm env, else single process (rank -1)."""
    try:
        args.world_size = int(os.environ['WORLD_SIZE'])
        args.rank = int(os.environ['RANK'])
        args.local_rank = int(os.environ['LOCAL_RANK'])
        args.local_world_size = int(os.environ['LOCAL_WORLD_SIZE'])
        distributed = True
    except (KeyError, ValueError):
        try:
            args.world_size = int(os.environ['SLURM_NTASKS'])
            args.rank = int(os.environ['SLURM_PROCID'])
            args.local_rank = int(os.environ['SLURM_LOCALID'])
            args.local_world_size = int(os.environ['SLURM_NTASKS_PER_NODE'])
            os.environ['RANK'] = str(args.rank)
            os.environ['WORLD_SIZE'] = str(args.world_size)
            distributed = True
        except (KeyError, ValueError):
            distributed = False
    if not distributed:
        args.world_size, args.rank, args.local_rank, args.local_world_size = 1, -1, -1, 1
    if getattr(args, 'seed', 0):
        torch.manual_seed(args.seed + args.rank)
    return args, distributed


def parse_args(argv=None) -> Namespace:
    """torchsr.py:157-236."""
    parser = ArgumentParser(description='Super-resolution training and inference on AMD Instinct MI355X')
    commands = parser.add_subparsers(dest='function', metavar='function', required=True)
    train = commands.add_parser('train', help='Train an SRGAN / ESRGAN model against an HD dataset.')
    train.add_argument('--batch-size', type=int, default=BATCH_SIZE)
    train.add_argument('--data-workers', type=positive_integer, default=16)
    train.add_argument('--dataset-multiplier', type=positive_integer, default=1)
    train.add_argument('--disable-amp', action='store_true')
    train.add_argument('--epochs', type=int, default=EPOCHS)
    train.add_argument('--gan-checkpoint', type=str, default=None)
    train.add_argument('--master-addr', type=str, default=None)
    train.add_argument('--master-port', type=str, default=None)
    train.add_argument('--model', type=str, default=MODEL, choices=MODELS.keys())
    train.add_argument('--pretrain-epochs', type=int, default=PRE_EPOCHS)
    train.add_argument('--psnr-checkpoint', type=str, default=None)
    train.add_argument('--seed', type=int, default=0)
    train.add_argument('--skip-image-save', action='store_true')
    train.add_argument('--train-dir', type=str, default=TRAIN_DIR)
    train.add_argument('--vgg-weights', type=str, default=None,
                       help='path to vgg19-dcbb9e9d.pth (default: $TORCHSR_VGG19_WEIGHTS, then the torch hub cache); '
                            '"random" trains against seeded random VGG19 features (smoke runs only)')
    train.add_argument('--no-graphs', action='store_true', help='run the step eagerly instead of as a hipGraph')
    train.add_argument('--device-data', action='store_true',
                       help='keep the decoded images in HBM and crop / flip / bicubic-downsample on the GPU')
    test = commands.add_parser('test', help='Generate a super resolution image from a trained model.')
    test.add_argument('image', type=str)
    test.add_argument('--model', type=str, default=MODEL, choices=MODELS.keys())
    test.add_argument('--precision', type=str, default='fp32', choices=('fp32', 'bf16'),
                      help='conv arithmetic of the generator forward: exact fp32 (the reference runs no autocast here) or '
                           'bf16 products with fp32 accumulation')
    return parser.parse_args(argv)


def main(argv=None) -> None:
    """torchsr.py:239-270."""
    args = parse_args(argv)
    args, distributed = distributed_params(args)
    model_class = crop_size = None
    if args.function == 'train':
        model_class, crop_size = select_trainer_model(args)
    # checked before wandb.init so that a refused start leaves no open run; a run that never enters the GAN phase
    # (--epochs 0: pre-training only) never evaluates the perceptual loss
    if args.function == 'train' and args.vgg_weights != 'random' and args.epochs > 0:
        # the reference always trains against the pretrained VGG19 (srgan/loss.py:30); a perceptual loss on random
        # features is not that objective, so it has to be asked for
        from torchsr_amd.srgan.loss import VGG19_FILE, _find_weights
        if _find_weights(args.vgg_weights) is None:
            sys.exit(f'torchsr train: {VGG19_FILE} not found (--vgg-weights PATH, $TORCHSR_VGG19_WEIGHTS or the torch hub '
                     'cache); pass "--vgg-weights random" to train against seeded random VGG19 features instead')
    if wandb and args.rank in [-1, 0]:                      # torchsr.py:242-243
        wandb.init(config=args, name='TorchSR', project='torchsr')
    device = get_device(args)
    if args.function == 'test':
        from torchsr_amd.test import test
        test(args, select_test_model(args), device)
        return
    if args.seed:
        random.seed(args.seed)
        np.random.seed(args.seed)
    if distributed:
        if args.master_addr:
            os.environ['MASTER_ADDR'] = args.master_addr
        if args.master_port:
            os.environ['MASTER_PORT'] = args.master_port
        from torchsr_amd.ddp import configure_comm, describe_group
        backend = 'nccl' if device.type == 'cuda' else 'gloo'
        comm = configure_comm(int(args.world_size), backend)  # RCCL channel bounds + CUs left to them, BEFORE the group exists
        args.comm_reserved_cus = comm['reserved_cus_in_comm_window']
        dist.init_process_group(backend=backend)
        if args.rank == 0:
            print(f'process group: {describe_group()}; communication: {comm}')
    args.use_graphs = not args.no_graphs
    from torchsr_amd.dataset import initialize_datasets, initialize_device_datasets
    if args.device_data:
        train_loader, test_loader, train_len, test_len = initialize_device_datasets(
            args.train_dir, device, batch_size=args.batch_size, crop_size=crop_size, upscale_factor=4,
            dataset_multiplier=args.dataset_multiplier, distributed=distributed, seed=args.seed,
            rank=max(int(args.rank), 0) if distributed else 0, world_size=int(args.world_size) if distributed else 1)
    else:
        train_loader, test_loader, train_len, test_len = initialize_datasets(
            args.train_dir, batch_size=args.batch_size, crop_size=crop_size, upscale_factor=4,
            dataset_multiplier=args.dataset_multiplier, workers=args.data_workers, distributed=distributed,
            seed=args.seed)
    trainer = model_class(device, args, train_loader, test_loader, train_len, test_len, distributed)
    trainer.train()
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

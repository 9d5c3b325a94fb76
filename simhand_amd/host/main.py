"""Training entry with the CLI surface of src/experiments/main.py (:36-199).

    python -m simhand_amd.host.main --experiment_type handclr_w --color_jitter --random_crop --rotate --crop \
        -resnet_size 50 --resize -sources ego4d --datasets_scale 1m -epochs 100 -batch_size 8192 ... --synthetic

    # 8 GPUs, one process per GPU (replaces the reference's hard-coded strategy="dp", main.py:152-163):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m simhand_amd.host.main ...

What is kept: flag names / defaults, config overlay order (JSON -> args), lr*sqrt(1024k), seed, experiment
naming, checkpoint monitor + filename template, model registry.  What is replaced: Lightning's fit loop ->
host/lightning.Trainer; DataParallel -> RCCL (global negatives); Comet / TensorBoard loggers -> stdout (no
network here).  What is out of scope: the dataset readers / OpenCV augmenter (src/data_loader/*, SURVEY 2
#10-12): without them only ``--synthetic`` input exists and the script says so.
"""
from __future__ import annotations

import math
import os
from pprint import pformat

import torch

from . import config as C
from . import dist as shdist
from .config import edict, read_json
from .experiments_utils import get_general_args, get_model, model_config_path, prepare_name, update_model_params, update_train_params
from .lightning import ModelCheckpoint, Trainer, seed_everything


class SyntheticPairs:
    """Iterable of collated batches with the schema of SURVEY Appendix B (what Data_Set + the default
    collate emit for handclr_w / peclr_w): this rank's shard of every global batch."""

    def __init__(self, samples: int, global_batch: int, size: int, rank: int, world: int, seed: int, device):
        _, self.b_loc = shdist.shard_pairs(global_batch, rank, world)
        self.steps = max(1, samples // global_batch)
        self.size, self.seed, self.rank, self.device = size, seed, rank, device
        self.samples = samples

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed * 1000 + self.rank)
        b, s, dev = self.b_loc, self.size, self.device
        for _ in range(self.steps):
            j1 = torch.rand(b, 21, 3, generator=g, device=dev) * s
            j1[:, :, 2] = 1.0
            j2 = j1.clone()
            j2[:, :, :2] += torch.randn(b, 21, 2, generator=g, device=dev) * 8.0
            ri = lambda lo, hi: torch.randint(lo, hi, (b,), generator=g, device=dev)  # noqa: E731
            yield {"transformed_image1": torch.randn(b, 3, s, s, generator=g, device=dev),
                   "transformed_image2": torch.randn(b, 3, s, s, generator=g, device=dev),
                   "joints1_aug": j1, "joints2_aug": j2, "joints1_ori": j1 / s, "joints2_ori": j2 / s,
                   "angle_1": ri(-45, 46).to(torch.float64), "angle_2": ri(-45, 46).to(torch.float64),
                   "jitter_x_1": -ri(0, 15), "jitter_x_2": -ri(0, 15), "jitter_y_1": -ri(0, 15), "jitter_y_2": -ri(0, 15)}


def main(argv=None):
    args = get_general_args("Model training script.", argv)
    rank, local, world = shdist.init_from_env()
    say = print if rank == 0 else (lambda *a, **k: None)
    say(f"Model Name: {args.experiment_type}")
    say(f"Dataset Name: {args.sources}")
    say(f"Dataset Scale: {args.datasets_scale}")
    source_scale = [f"{s}-{args.datasets_scale}" for s in args.sources]
    if any(("freihand" in s or "youtube" in s) for s in args.sources):
        source_scale = ["fh_yt3d"]

    train_param = update_train_params(args, edict(read_json(C.TRAINING_CONFIG_PATH)))
    model_param = edict(read_json(model_config_path(args.experiment_type)))  # ValueError for unknown types (main.py:80)
    lr_str = f"{1e-4 * math.sqrt(1024 * train_param.accumulate_grad_batches):.1e}"
    say(f"Train parameters {pformat(dict(train_param))}")
    seed_everything(train_param.seed)

    if not (args.synthetic or args.synthetic_raw):
        raise SystemExit("dataset readers (src/data_loader/*) are outside this build's scope: pass --synthetic (ready batches) or "
                         "--synthetic_raw (raw frames through the GPU batch producer), or feed Trainer.fit() your own iterable of batch dicts")
    size = args.image_size or (train_param.augmentation_params.resize_shape[0] if train_param.augmentation_flags.resize else 224)
    samples = args.synthetic_samples or 4 * train_param.batch_size
    device = torch.device("cuda", torch.cuda.current_device())
    if args.synthetic_raw:
        from .data import GpuAugmenter, SyntheticRawPairs

        if args.image_size:
            train_param.augmentation_params.resize_shape = [args.image_size, args.image_size]
        augmenter = GpuAugmenter(train_param.augmentation_flags, train_param.augmentation_params)
        data = SyntheticRawPairs(augmenter, samples, train_param.batch_size, rank, world, train_param.seed, device)
    else:
        data = SyntheticPairs(samples, train_param.batch_size, size, rank, world, train_param.seed, device)

    model_param = update_model_params(model_param, args, samples, train_param)
    model_param.augmentation = [k for k, v in train_param.augmentation_flags.items() if v]
    say(f"Model parameters {pformat(dict(model_param))}")
    mode = "train" if not args.resume and not args.eval else "eval"
    cls = get_model(args.experiment_type, args.heatmap, args.denoiser)
    if cls is None:
        raise ValueError(f"Model {args.experiment_type} is not supported.")
    model = cls(config=model_param, logger_debug=None, mode=mode)

    experiment_name = prepare_name(f"{args.experiment_type}_", train_param, hybrid_naming=False)
    out_dir = args.out_dir or os.path.join(os.environ.get("SAVED_MODELS_BASE_PATH", "./runs"), experiment_name)
    ckpt = ModelCheckpoint(
        save_top_k=args.save_top_k, monitor="contrastive_loss", mode="min", dirpath=os.path.join(out_dir, "checkpoints"),
        filename=f"{args.experiment_type}_pretrain_{{epoch:02d}}_train_{source_scale}_bs_{args.batch_size and args.batch_size / 1024}_"
                 f"{1024 * train_param.accumulate_grad_batches}_lr_{lr_str}_{{contrastive_loss:.6f}}")
    precision = args.precision or train_param.precision
    trainer = Trainer(max_epochs=train_param.epochs, precision=precision, callbacks=[ckpt], log_every_n_steps=5,
                      default_root_dir=out_dir, max_steps=args.max_steps, sync_batchnorm=getattr(args, "sync_batchnorm", False))
    if args.eval:
        raise SystemExit("--eval drives the reference's visualisation path (simhand_vis), which is out of scope")
    if args.resume and args.resume_path is None:
        raise ValueError(f"{args.resume_path} is empty, please cheack about it!")
    trainer.fit(model, train_dataloaders=data, val_dataloaders=None, ckpt_path=args.resume_path if args.resume else None)
    say(f"For the {args.experiment_type} model contrastive learning, the best checkpoint is: {ckpt.best_model_path}")
    return trainer


if __name__ == "__main__":
    main()

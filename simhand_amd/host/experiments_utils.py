"""CLI / config surface of src/experiments/utils.py (hot-path part): the argparse
flags of ``get_general_args`` (:30-233), ``update_train_params`` (:345),
``update_param`` (:385), ``prepare_name`` (:404), ``get_model`` (:633-700) and
``update_model_params`` (:725-755).  Flag names, defaults and assertion
messages follow the reference; additive, build-only flags are grouped at the
end (``--synthetic`` ...).  Pinned by tests/golden/cli.json, captured from the
reference's own parser.
"""
from __future__ import annotations

import argparse
from typing import List, Optional

from .config import edict

AUG_FLAGS = ["color_drop", "color_jitter", "crop", "cut_out", "flip", "gaussian_blur", "random_crop", "resize", "rotate",
             "sobel_filter", "gaussian_noise"]
SOURCES = ["freihand", "interhand", "mpii", "youtube", "ego4d", "100doh", "ah", "ah-exo", "ah-ego"]


# (flag, kwargs) in the reference parser's order.  Names, types, defaults and choices are the contract
# (tests/golden/cli.json); the help texts are this build's own wording.
_WEIGHT_FLAGS = [
    ("--experiment_type", dict(type=str, help="model registry key: simclr, peclr, simhand, simhand_w / handclr_w, simclr_w, peclr_w, ...")),
    ("--weight_type", dict(type=str, help="how joint distances become loss weights: linear | non_linear")),
    ("--joints_type", dict(type=str, help="which joints feed the weights: original | augmented")),
    ("--diff_type", dict(type=str, help="joint-distance definition: w_o_abs | w_abs | mpjpe")),
    ("--pos_neg", dict(type=str, help="which terms are weighted: pos | neg | pos_neg")),
    ("--non_linear_lambda_pos", dict(type=float, help="sigmoid slope of the positive weights: 5.0 | 2.5 | 1.0")),
    ("--non_linear_lambda_neg", dict(type=float, help="sigmoid slope of the negative weights: 0.05 | 0.01 | 0.005")),
    ("--use_pca", dict(action="store_true", default=False, help="project the joints onto 14 PCA components before measuring distances")),
    ("--resume", dict(action="store_true", help="continue from --resume_path")),
    ("--resume_path", dict(type=str, help="checkpoint file to continue from")),
    ("--eval", dict(action="store_true", help="evaluation / visualisation run (not part of this build)")),
    ("--eval_path", dict(type=str, help="checkpoint file to evaluate")),
    ("--debug", dict(action="store_true", help="verbose logging")),
    ("--vis", dict(action="store_true", help="dump intermediate batches for plotting")),
    ("--vis_save_dir", dict(type=str, default="", help="where --vis dumps go")),
    ("--datasets_scale", dict(type=str, help="size tag of the pre-training subset, e.g. 1m")),
]
_AUG_HELP = {"color_drop": "drop colour channels at random", "color_jitter": "random HSV jitter", "crop": "crop around the hand",
             "cut_out": "erase a random rectangle", "flip": "random horizontal flip", "gaussian_blur": "random Gaussian blur",
             "rotate": "random in-plane rotation", "random_crop": "jittered crop position", "resize": "resize to resize_shape",
             "sobel_filter": "Sobel edge filter", "gaussian_noise": "additive Gaussian noise"}
_AUG_ORDER = ["color_drop", "color_jitter", "crop", "cut_out", "flip", "gaussian_blur", "rotate", "random_crop", "resize",
              "sobel_filter", "gaussian_noise"]
_TRAIN_FLAGS = [
    ("-tag", dict(action="append", default=[], help="free-form run tag (repeatable)")),
    ("-batch_size", dict(type=int, help="global batch in pairs")),
    ("-epochs", dict(type=int, help="epochs to train")),
    ("-seed", dict(type=int, help="random seed")),
    ("--gpus", dict(type=str, default="0", help="device ids (ignored by the reference's Trainer as well)")),
    ("-num_workers", dict(type=int, help="data-loader worker processes")),
    ("-train_ratio", dict(type=float, help="train share of the train/validation split")),
    ("-accumulate_grad_batches", dict(type=int, help="micro-batches per optimizer step")),
    ("-lr", dict(type=float, default=None, help="base learning rate")),
    ("-optimizer", dict(type=str, default=None, choices=["LARS", "adam"], help="optimizer")),
    ("--denoiser", dict(action="store_true", default=False, help="denoiser variant of the model")),
    ("--heatmap", dict(action="store_true", default=False, help="heat-map variant of the model")),
    ("-sources", dict(action="append", default=[], choices=SOURCES, help="dataset to draw from (repeatable)")),
    ("-log_interval", dict(type=str, default="epoch", choices=["step", "epoch"], help="logging granularity")),
    ("-experiment_key", dict(type=str, default=None, help="experiment key of a pre-trained encoder")),
    ("-checkpoint", dict(type=str, default="", help="checkpoint name to restore")),
    ("-meta_file", dict(type=str, default=None, help="file that receives the experiment name")),
    ("-experiment_name", dict(type=str, default="", help="name under which the run is logged")),
    ("-save_period", dict(type=int, default=1, help="epochs between snapshots")),
    ("-save_top_k", dict(type=int, default=3, help="how many best checkpoints to keep")),
    ("--encoder_trainable", dict(action="store_true", default=False, help="fine-tune the encoder in downstream runs")),
    ("-resnet_size", dict(type=str, default="18", choices=["18", "34", "50", "101", "152"], help="backbone depth")),
    ("-lr_max_epochs", dict(type=int, default=None, help="epoch count the cosine schedule is stretched over")),
    ("--use_palm", dict(action="store_true", default=False, help="regress the palm centre instead of the wrist")),
]


def build_parser(description: str = "Script for training baseline supervised model") -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=description)
    for flag, kw in _WEIGHT_FLAGS:
        p.add_argument(flag, **kw)
    for f in _AUG_ORDER:
        p.add_argument(f"--{f}", action="store_true", help=_AUG_HELP[f])
    for flag, kw in _TRAIN_FLAGS:
        p.add_argument(flag, **kw)
    return p


BUILD_ONLY_FLAGS = ("synthetic", "synthetic_raw", "synthetic_samples", "precision", "image_size", "max_steps", "out_dir", "sync_batchnorm")


def add_build_flags(p: argparse.ArgumentParser) -> argparse.ArgumentParser:
    """Additive flags of this build: none of them changes a reference default."""
    p.add_argument("--synthetic", action="store_true", help="train on the SURVEY 8d synthetic batch schema (no dataset on disk)")
    p.add_argument("--synthetic_raw", action="store_true",
                   help="synthetic RAW frames + joints through the GPU batch producer (the reference's augmentation chain on the device)")
    p.add_argument("--synthetic_samples", type=int, default=None, help="samples per synthetic epoch (default 4 global batches)")
    p.add_argument("--precision", type=str, default=None, choices=["32", "bf16", "16", "fp8"],
                   help="kernel dtype: 32 = exact-fp32 MFMA, bf16/16 = bf16 MFMA (overrides training_config.json's 16)")
    p.add_argument("--image_size", type=int, default=None, help="synthetic image side (default: resize_shape or 224)")
    p.add_argument("--max_steps", type=int, default=-1, help="stop after this many optimizer steps")
    p.add_argument("--out_dir", type=str, default=None, help="where checkpoints go (default $SAVED_MODELS_BASE_PATH or ./runs)")
    p.add_argument("--sync_batchnorm", action="store_true",
                   help="multi-GPU: BatchNorm statistics over the global batch instead of per rank (PL's Trainer(sync_batchnorm=True))")
    return p


def get_general_args(description: str = "Script for training baseline supervised model", argv: Optional[List[str]] = None,
                     with_build_flags: bool = True) -> argparse.Namespace:
    p = build_parser(description)
    if with_build_flags:
        add_build_flags(p)
    return p.parse_args(argv)


def update_param(args: argparse.Namespace, config: edict, params: List[str]) -> edict:
    d = vars(args)
    for k in params:
        if d[k] is not None:
            config[k] = d[k]
    return config


def update_train_params(args: argparse.Namespace, train_param: edict) -> edict:
    if args.train_ratio is not None:
        train_param.train_ratio = (args.train_ratio * 100 % 100) / 100.0
    train_param.update(update_param(args, train_param, ["batch_size", "epochs", "train_ratio", "num_workers", "seed", "use_palm"]))
    train_param.augmentation_flags = update_param(args, train_param.augmentation_flags, AUG_FLAGS)
    if args.accumulate_grad_batches is not None:
        train_param.accumulate_grad_batches = args.accumulate_grad_batches
    return train_param


def prepare_name(prefix: str, train_param: edict, hybrid_naming: bool = False) -> str:
    codes = {"color_drop": "CD", "color_jitter": "CJ", "crop": "C", "cut_out": "CO", "flip": "F", "gaussian_blur": "GB",
             "random_crop": "RC", "resize": "Re", "rotate": "Ro", "sobel_filter": "SF", "gaussian_noise": "GN"}
    if hybrid_naming:
        raise NotImplementedError("hybrid (pairwise + contrastive) naming belongs to the downstream experiments (out of scope)")
    aug = "_".join(sorted(codes[k] for k, v in train_param.augmentation_flags.items() if v))
    return f"{prefix}{train_param.batch_size}{aug}"


def update_model_params(model_param: edict, args, data_length: int, train_param: edict) -> edict:
    model_param = update_param(args, model_param, ["optimizer", "lr", "resnet_size", "lr_max_epochs"])
    model_param.num_samples = data_length
    model_param.batch_size = train_param.batch_size
    model_param.num_of_mini_batch = train_param.accumulate_grad_batches
    model_param.vis = args.vis
    model_param.vis_save_dir = args.vis_save_dir
    if args.weight_type is not None:
        assert args.weight_type in ["linear", "non_linear"], "Invalid value for --weight_type"
        assert args.joints_type in ["original", "augmented"], "Invalid value for --joints_type"
        assert args.diff_type in ["w_o_abs", "w_abs", "mpjpe"], "Invalid value for --diff_type"
        assert args.pos_neg in ["pos", "neg", "pos_neg"], "Invalid value for --pos_neg"
        model_param.weight_type = args.weight_type
        model_param.joints_type = args.joints_type
        model_param.diff_type = args.diff_type
        model_param.pos_neg = args.pos_neg
        model_param.use_pca = args.use_pca
        if args.weight_type == "non_linear":
            assert args.non_linear_lambda_pos in [5.0, 2.5, 1.0], "Invalid value for --non_linear_lambda_pos"
            assert args.non_linear_lambda_neg in [0.05, 0.01, 0.005], "Invalid value for --non_linear_lambda_neg"
            model_param.non_linear_lambda_pos = args.non_linear_lambda_pos
            model_param.non_linear_lambda_neg = args.non_linear_lambda_neg
    return model_param


def get_model(experiment_type: str, heatmap_flag: bool = False, denoiser_flag: bool = False):
    """Registry of src/experiments/utils.py:633-700 plus the README's `handclr_w` spelling, which the
    reference's if-chain does not know (it returns None there and main() then fails)."""
    from . import unsupervised as u

    if heatmap_flag:
        raise NotImplementedError("heatmap models are not implemented in the reference either")
    table = {"simclr": u.SimCLR, "peclr": u.PeCLR, "simhand-base": u.SiMHand_BASE, "simhand": u.SiMHand, "simhand_w": u.SiMHand_W,
             "simclr_w": u.SimCLR_W, "peclr_w": u.PeCLR_W, "simhand_vis": u.SiMHand_VIS, "handclr_w": u.HandCLR_W}
    return table.get(experiment_type)


def model_config_path(experiment_type: str) -> str:
    """main.py:73-80 picks the JSON by substring; `handclr_w` matches none of them in the reference
    (ValueError) -- here it resolves to the shipped handclr_config.json."""
    from . import config as C

    if "simclr" in experiment_type:
        return C.SIMCLR_CONFIG
    if "peclr" in experiment_type:
        return C.PECLR_CONFIG
    if "simhand" in experiment_type or "handclr" in experiment_type:
        return C.SIMHAND_CONFIG
    raise ValueError(f"Model {experiment_type} is not supported.")

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


def build_parser(description: str = "Script for training baseline supervised model") -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--experiment_type", type=str, help="The training model name.")
    p.add_argument("--weight_type", type=str, help="Weight type (linear / non_linear)")
    p.add_argument("--joints_type", type=str, help="joints type (original / augmented)")
    p.add_argument("--diff_type", type=str, help="joints_differ (w_o_abs / w_abs / mpjpe)")
    p.add_argument("--pos_neg", type=str, help="pos_neg weighting(pos / neg / pos_neg)")
    p.add_argument("--non_linear_lambda_pos", type=float, help="non_linear_parm (5.0 / 2.5 / 1.0)")
    p.add_argument("--non_linear_lambda_neg", type=float, help="non_linear_parm (0.05 / 0.01 / 0.005)")
    p.add_argument("--use_pca", action="store_true", help="To enable PCA denoise.", default=False)
    p.add_argument("--resume", action="store_true", help="resume the model training.")
    p.add_argument("--resume_path", type=str, help="resume the model checkpoints path")
    p.add_argument("--eval", action="store_true", help="eval the model and visualization.")
    p.add_argument("--eval_path", type=str, help="eval the model checkpoints path")
    p.add_argument("--debug", action="store_true", help="Enable debug logging.")
    p.add_argument("--vis", action="store_true", help="Enable save the intermediate data.")
    p.add_argument("--vis_save_dir", type=str, help="data visualization save dir", default="")
    p.add_argument("--datasets_scale", type=str, help="Usage sacle of the pre-trained data set.")
    helps = {"color_drop": "To enable random color drop", "color_jitter": "To enable random jitter", "crop": "To enable cropping",
             "cut_out": "To enable random cur out", "flip": "To enable random flipping", "gaussian_blur": "To enable gaussina blur",
             "rotate": "To rotate samples randomly", "random_crop": "To enable random cropping", "resize": "To enable resizing",
             "sobel_filter": "To enable sobel filtering", "gaussian_noise": "To add gaussian noise."}
    for f in ["color_drop", "color_jitter", "crop", "cut_out", "flip", "gaussian_blur", "rotate", "random_crop", "resize",
              "sobel_filter", "gaussian_noise"]:
        p.add_argument(f"--{f}", action="store_true", help=helps[f])
    p.add_argument("-tag", action="append", help="Tag for comet", default=[])
    p.add_argument("-batch_size", type=int, help="Batch size")
    p.add_argument("-epochs", type=int, help="Number of epochs")
    p.add_argument("-seed", type=int, help="To add seed")
    p.add_argument("--gpus", type=str, default="0", help="gpu ids")
    p.add_argument("-num_workers", type=int, help="Number of workers for Dataloader.")
    p.add_argument("-train_ratio", type=float, help="Ratio of train:validation split.")
    p.add_argument("-accumulate_grad_batches", type=int, help="Number of batches to accumulate gradient.")
    p.add_argument("-lr", type=float, help="learning rate", default=None)
    p.add_argument("-optimizer", type=str, help="Select optimizer", default=None, choices=["LARS", "adam"])
    p.add_argument("--denoiser", action="store_true", help="To enable denoising", default=False)
    p.add_argument("--heatmap", action="store_true", help="To enable heatmap model", default=False)
    p.add_argument("-sources", action="append", help="Data sources to use.", default=[], choices=SOURCES)
    p.add_argument("-log_interval", type=str, help="To enable denoising", default="epoch", choices=["step", "epoch"])
    p.add_argument("-experiment_key", type=str, help="Experiment key of pretrained encoder", default=None)
    p.add_argument("-checkpoint", type=str, help="checkpoint name to restore.", default="")
    p.add_argument("-meta_file", type=str, help="File to save the name of the experiment.", default=None)
    p.add_argument("-experiment_name", type=str, help="experiment name for logging", default="")
    p.add_argument("-save_period", type=int, help="interval at which experiments should be saved", default=1)
    p.add_argument("-save_top_k", type=int, help="Top snapshots to save", default=3)
    p.add_argument("--encoder_trainable", action="store_true", help="To enable encoder training in SSL", default=False)
    p.add_argument("-resnet_size", type=str, help="Resnet size", default="18", choices=["18", "34", "50", "101", "152"])
    p.add_argument("-lr_max_epochs", type=int, help="Top snapshots to save", default=None)
    p.add_argument("--use_palm", action="store_true", help="To regress plam instead of wrist.", default=False)
    return p


BUILD_ONLY_FLAGS = ("synthetic", "synthetic_samples", "precision", "image_size", "max_steps", "out_dir")


def add_build_flags(p: argparse.ArgumentParser) -> argparse.ArgumentParser:
    """Additive flags of this build: none of them changes a reference default."""
    p.add_argument("--synthetic", action="store_true", help="train on the SURVEY 8d synthetic batch schema (no dataset on disk)")
    p.add_argument("--synthetic_samples", type=int, default=None, help="samples per synthetic epoch (default 4 global batches)")
    p.add_argument("--precision", type=str, default=None, choices=["32", "bf16", "16"],
                   help="kernel dtype: 32 = exact-fp32 MFMA, bf16/16 = bf16 MFMA (overrides training_config.json's 16)")
    p.add_argument("--image_size", type=int, default=None, help="synthetic image side (default: resize_shape or 224)")
    p.add_argument("--max_steps", type=int, default=-1, help="stop after this many optimizer steps")
    p.add_argument("--out_dir", type=str, default=None, help="where checkpoints go (default $SAVED_MODELS_BASE_PATH or ./runs)")
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

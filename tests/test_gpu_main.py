"""GPU: the real entry point (`python -m simhand_amd.host.main`, CLI of src/experiments/main.py:36-199) end to end:
Trainer.fit -> ModelCheckpoint with the reference's filename template (main.py:143-149) -> --resume continues bit for bit
-> the checkpoint exports to a torchvision-keyed ResNet-50 state dict (src/models/port_model.py:7-48, hubconf.py:6-23)."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


def _argv(out_dir, extra):
    return ["--experiment_type", "handclr_w", "--color_jitter", "--random_crop", "--rotate", "--crop", "--resize", "-resnet_size", "50",
            "-sources", "ego4d", "--datasets_scale", "1m", "-epochs", "1", "-batch_size", "8", "-accumulate_grad_batches", "1",
            "-save_top_k", "1", "-num_workers", "0", "--weight_type", "linear", "--joints_type", "augmented", "--diff_type", "mpjpe",
            "--pos_neg", "pos_neg", "--synthetic", "--synthetic_samples", "32", "--image_size", "64", "--precision", "bf16",
            "--out_dir", str(out_dir)] + extra


def test_main_fit_checkpoint_resume_export(tmp_path):
    from simhand_amd.host import export
    from simhand_amd.host.main import main

    # (A) four uninterrupted steps
    ta = main(_argv(tmp_path / "a", ["--max_steps", "4"]))
    la = [float(x) for x in ta.step_losses]
    assert len(la) == 4 and all(l == l for l in la)
    assert len({round(l, 6) for l in la}) > 1, f"the optimizer never moved the loss: {la}"
    # (B) three steps, then the checkpoint callback fires (epoch cut short by --max_steps)
    tb = main(_argv(tmp_path / "b", ["--max_steps", "3"]))
    lb = [float(x) for x in tb.step_losses]
    assert lb == la[:3], (lb, la)  # deterministic kernels: same seed, same losses bit for bit
    ckdir = tmp_path / "b" / "checkpoints"
    names = os.listdir(ckdir)
    assert len(names) == 1, names
    # main.py:143-149: f"{type}_pretrain_{{epoch:02d}}_train_{sources}_bs_{bs/1024}_{1024*acc}_lr_{lr}_{{contrastive_loss:.6f}}"
    pat = r"^handclr_w_pretrain_epoch=00_train_\['ego4d-1m'\]_bs_0\.0078125_1024_lr_3\.2e-03_contrastive_loss=\d+\.\d{6}\.ckpt$"
    assert re.match(pat, names[0]), names[0]
    ck = torch.load(ckdir / names[0], map_location="cpu", weights_only=False)
    assert ck["global_step"] == 3 and ck["epoch"] == 0 and ck["epoch_complete"] is False and ck["batches_seen"] == 3
    assert set(ck) >= {"state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers", "epoch", "global_step"}
    # (C) resume: step 4 of the resumed run == step 4 of the uninterrupted one
    tc = main(_argv(tmp_path / "c", ["--resume", "--resume_path", str(ckdir / names[0]), "--max_steps", "4"]))
    lc = [float(x) for x in tc.step_losses]
    assert len(lc) == 1 and tc.global_step == 4
    assert lc[0] == la[3], (lc, la)
    # (D) export: torchvision-keyed ResNet-50 state dict, tensors = the checkpoint's encoder.features.* by position
    out = tmp_path / "resnet50_simhand.pth"
    sd = export.export_torchvision_state_dict(str(ckdir / names[0]), str(out), 50)
    feats = [(k, v) for k, v in ck["state_dict"].items() if "features" in k]
    assert len(feats) == 318 and len(sd) == 320
    for (dk, dv), (sk, sv) in zip(sd.items(), feats):
        assert dk.split(".")[-1] == sk.split(".")[-1] and torch.equal(dv, sv.cpu()), (dk, sk)
    hub = export.resnet50_simhand(pretrained=True, path=str(out))
    assert torch.equal(hub.state_dict()["layer4.2.conv3.weight"], ck["state_dict"]["encoder.features.7.2.conv3.weight"].cpu())


def test_precision_16_resume_continues_bit_for_bit(tmp_path):
    """--precision 16 (fp16 storage + GradScaler, the reference's default precision): the scaler state travels in the checkpoint and the
    resumed run's next step equals the uninterrupted run's."""
    from simhand_amd.host.main import main

    def argv(d, extra):
        a = _argv(d, extra)
        a[a.index("--precision") + 1] = "16"
        return a

    ta = main(argv(tmp_path / "a", ["--max_steps", "4"]))
    la = [float(x) for x in ta.step_losses]
    tb = main(argv(tmp_path / "b", ["--max_steps", "3"]))
    assert [float(x) for x in tb.step_losses] == la[:3]
    ckdir = tmp_path / "b" / "checkpoints"
    name = os.listdir(ckdir)[0]
    ck = torch.load(ckdir / name, map_location="cpu", weights_only=False)
    # (at random init a ResNet-50's scaled gradients may overflow fp16 once: the scaler then halves the scale and skips that step)
    st = ck["native_amp_scaling_state"]
    assert st["scale"] == tb.scaler.get_scale() and st["scale"] in (65536.0, 32768.0, 16384.0) and st["_growth_tracker"] == tb.scaler._growth_tracker
    tc = main(argv(tmp_path / "c", ["--resume", "--resume_path", str(ckdir / name), "--max_steps", "4"]))
    assert [float(x) for x in tc.step_losses] == la[3:], (list(tc.step_losses), la)
    assert tc.scaler.get_scale() == ta.scaler.get_scale() and tc.scaler._growth_tracker == ta.scaler._growth_tracker


def test_fp8_scaling_state_round_trips_through_the_engine():
    """The delayed-scaling amax rings of the fp8 sites (forward and data gradient) as a checkpointable dict: a fresh engine that loads it
    continues with the same scales."""
    from oracle import step as orc
    from simhand_amd import _lib, ops
    from tests.test_gpu_configs import _oracle, _product

    om = _oracle("simclr", "50", {}, 3, 0.1)
    batch = {k: v.to("cuda") for k, v in orc.synthetic_batch(4, size=224, seed=3).items()}
    _lib.load().simhand_test_igemm256_enable(2)
    m1 = _product("SimCLR", "50", {}, om, torch.bfloat16, 4)
    m1.set_compute_dtype(torch.bfloat16, fp8=True)
    for i in range(2):
        m1.zero_grad()
        m1.training_step(batch, i)["loss"].backward()
    sd = m1.encoder.engine.fp8_state_dict()
    assert len([k for k in sd if k.startswith("fwd:")]) == 9 and len([k for k in sd if k.startswith("bwd:")]) == 9, sorted(sd)
    m2 = _product("SimCLR", "50", {}, om, torch.bfloat16, 4)
    m2.set_compute_dtype(torch.bfloat16, fp8=True)
    m2.encoder.engine.load_fp8_state_dict(sd, torch.device("cuda"))
    l1 = m1.training_step(batch, 2)["loss"]
    l2 = m2.training_step(batch, 2)["loss"]
    assert torch.equal(l1, l2), (float(l1), float(l2))

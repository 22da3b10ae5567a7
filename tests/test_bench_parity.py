"""CPU tests of bench.py's own logic: the parity gate (`_parity`) on synthetic tensors and the `--gpus N` self-launch.

The gate's verdict on the HIP step must follow the HIP side only (losses vs the fp32 oracle, image vs the fp64 evaluation);
the fp32 oracle's distance from its own fp64 evaluation is reported and held to a sanity bound that fp32 noise cannot reach
(round 5's driver run failed on exactly that: a correct HIP step, an oracle 2.6e-5 in relative L2 from fp64, limit 2e-5)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _case(hip_noise=2e-6, oracle_noise=2.6e-5, loss_rel=1e-6, seed=0):
    """A tanh-like image with a few saturated pixels; `*_noise` = relative-L2 distance of that evaluation from the truth."""
    g = torch.Generator().manual_seed(seed)
    truth = torch.tanh(0.6 * torch.randn(2, 3, 64, 64, generator=g, dtype=torch.float64))

    def noisy(rel):
        n = torch.randn(truth.shape, generator=g, dtype=torch.float64)
        return (truth + n * (rel * truth.norm() / n.norm())).float()

    Go = {"total_loss": torch.tensor(3.25), "bbox_pred": torch.tensor(0.125), "bbox_pred_all": torch.linspace(0.1, 0.9, 8)}
    Do = {"total_img_loss": torch.tensor(1.5), "ac_loss_real": torch.tensor(2.0)}
    G0 = {k: (v.clone() if k == "bbox_pred_all" else float(v) * (1 + loss_rel)) for k, v in Go.items()}
    D0 = {k: float(v) * (1 - loss_rel) for k, v in Do.items()}
    return (G0, D0, noisy(hip_noise)), Go, Do, noisy(oracle_noise), truth


def test_good_hip_noisy_oracle_passes():
    """Round 5's driver record, restated: HIP 2e-6 from fp64, fp32 oracle 2.6e-5 from fp64 -> the step is accepted."""
    step, Go, Do, img_o, truth = _case()
    r = bench._parity(step, Go, Do, img_o, "t", img64=truth)
    assert r["ok"] and r["hip_ok"] and r["oracle_sane"]
    assert r["imgs_pred"]["judged_against"] == "fp64 oracle"
    assert r["imgs_pred"]["fp32_oracle_vs_fp64"]["rel_l2"] > 2e-5          # the number that used to fail the run
    assert r["imgs_pred"]["fp32_oracle_vs_fp64"]["sane"]
    # 3x the worst fp32 noise measured on the oracle still is not "broken"
    step, Go, Do, img_o, truth = _case(oracle_noise=8e-5)
    assert bench._parity(step, Go, Do, img_o, "t", img64=truth)["ok"]


def test_bad_hip_image_fails_whatever_the_oracle_does():
    step, Go, Do, img_o, truth = _case(hip_noise=1e-3, oracle_noise=1e-6)
    r = bench._parity(step, Go, Do, img_o, "t", img64=truth)
    assert not r["ok"] and not r["hip_ok"] and r["oracle_sane"]
    # a single pixel off by 1e-3 (relative L2 stays tiny) is caught by the per-pixel rule
    step, Go, Do, img_o, truth = _case(hip_noise=1e-7)
    step[2][0, 0, 3, 3] += 1e-3
    r = bench._parity(step, Go, Do, img_o, "t", img64=truth)
    assert not r["hip_ok"] and r["imgs_pred"]["pixels_over_rtol_plus_atol"] == 1
    # relative L2 alone: every pixel inside rtol + atol, the image as a whole 3e-5 away
    step, Go, Do, img_o, truth = _case(hip_noise=3e-5)
    r = bench._parity(step, Go, Do, img_o, "t", img64=truth)
    assert not r["hip_ok"] and r["imgs_pred"]["rel_l2"] > 2e-5


def test_bad_loss_or_bbox_fails():
    step, Go, Do, img_o, truth = _case(loss_rel=3e-4)
    r = bench._parity(step, Go, Do, img_o, "t", img64=truth)
    assert not r["ok"] and not r["hip_ok"] and r["max_rel"] > 1e-4
    step, Go, Do, img_o, truth = _case()
    step[0]["bbox_pred_all"][2] += 1e-3
    r = bench._parity(step, Go, Do, img_o, "t", img64=truth)
    assert not r["ok"] and not r["bbox_pred_all_ok"]
    step, Go, Do, img_o, truth = _case()
    del step[1]["ac_loss_real"]                                            # a loss the HIP step did not report
    Do2 = dict(Do)
    with pytest.raises(KeyError):
        bench._parity(step, Go, Do2, img_o, "t", img64=truth)


def test_broken_oracle_is_flagged_as_the_oracle():
    """An oracle whose fp32 evaluation is orders of magnitude from its fp64 one fails the run, and the record says which side."""
    step, Go, Do, img_o, truth = _case(oracle_noise=5e-3)
    r = bench._parity(step, Go, Do, img_o, "t", img64=truth)
    assert not r["ok"] and r["hip_ok"] and not r["oracle_sane"]
    assert not r["imgs_pred"]["fp32_oracle_vs_fp64"]["sane"]


def test_without_fp64_the_fp32_oracle_is_the_yardstick():
    step, Go, Do, img_o, truth = _case(hip_noise=2e-6, oracle_noise=0.0)
    r = bench._parity(step, Go, Do, img_o, "t")
    assert r["ok"] and r["imgs_pred"]["judged_against"] == "fp32 oracle" and r["oracle_sane"]
    step, Go, Do, img_o, truth = _case(hip_noise=1e-3, oracle_noise=0.0)
    assert not bench._parity(step, Go, Do, img_o, "t")["ok"]


def test_later_step_index_is_a_constant():
    """The checked step does not follow --warmup: bench.py names a fixed index."""
    assert bench.PARITY_LATER_STEP == 4
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "done[0] == PARITY_LATER_STEP" in src


def test_gpus_n_without_a_launcher_starts_n_ranks(monkeypatch):
    """`python bench.py --gpus 2 ...` with no torchrun environment: one torch.distributed.run child with 2 ranks of this
    file and the same arguments, before torch is imported in the parent; its status is the parent's status."""
    import subprocess
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_must_equal_gpus(monkeypatch):
    """Under a launcher whose world differs from --gpus the run refuses to print a line (n_gpus would lie)."""
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "--gpus 4 but WORLD_SIZE=1" in str(e.value.code)

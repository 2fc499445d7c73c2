"""GPU: the three drop-in CLIs end to end (FSC_finetune_cross.py, FSC_pretrain.py, FSC_test_cross.py) on a tiny on-disk
FSC147-shaped dataset (PIL/scipy loaders -> fused steps -> checkpoint -> test CLI reading that checkpoint) and on the synthetic
fallback.  Subprocesses, as a user would run them."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd=ROOT):
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.fixture(scope="module")
def fake_fsc(tmp_path_factory):
    from PIL import Image
    root = tmp_path_factory.mktemp("fsc")
    (root / "images_384_VarV2").mkdir()
    rs = np.random.RandomState(0)
    anno, names = {}, []
    for k, (w, h) in enumerate([(640, 384), (512, 384), (700, 400), (900, 384)]):
        name = "%d.jpg" % k
        Image.fromarray(rs.randint(0, 255, size=(h, w, 3)).astype(np.uint8)).save(root / "images_384_VarV2" / name)
        pts = np.stack([rs.uniform(5, w - 5, 12), rs.uniform(5, h - 5, 12)], 1).tolist()
        boxes = []
        for b in range(3):
            x1, y1 = int(rs.uniform(0, w - 80)), int(rs.uniform(0, h - 80))
            boxes.append([[x1, y1], [x1, y1 + 40], [x1 + 50, y1 + 40], [x1 + 50, y1]])
        anno[name] = {"points": pts, "box_examples_coordinates": boxes}
        names.append(name)
    json.dump(anno, open(root / "annotation_FSC147_384.json", "w"))
    json.dump({"train": names, "val": names[:2], "test": names[2:]}, open(root / "Train_Test_Val_FSC_147.json", "w"))
    return str(root)


def test_finetune_then_test_cli_on_files(fake_fsc, tmp_path):
    out = str(tmp_path / "ft")
    log = run(["FSC_finetune_cross.py", "--data_path", fake_fsc, "--batch_size", "2", "--epochs", "1", "--warmup_epochs", "0",
               "--num_workers", "0", "--no_do_aug", "--output_dir", out, "--resume", "", "--log_every", "1", "--blr", "1e-3"])
    lines = [json.loads(l) for l in log.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all(np.isfinite(l["loss"]) for l in lines)          # 4 images / batch 2, drop_last
    ckpt = os.path.join(out, "checkpoint__finetuning_last.pth")
    assert os.path.exists(ckpt)
    # validation pass (FSC_finetune_cross.py:329-350, :416-423): MAE / RMSE / NAE line and the model-selection checkpoint
    assert os.path.exists(os.path.join(out, "checkpoint__finetuning_minMAE.pth"))
    val = [l for l in log.splitlines() if l.startswith("[Val Epoch #0]")]
    assert len(val) == 1 and "MAE:" in val[0] and "RMSE:" in val[0] and "NAE:" in val[0]
    assert any(l.startswith("[Train Epoch #0] - MAE:") for l in log.splitlines())
    assert all(l["grad_norm"] is not None and np.isfinite(l["grad_norm"]) and l["grad_norm"] > 0 for l in lines)
    import torch
    opt = torch.load(ckpt, map_location="cpu", weights_only=False)["optimizer"]
    # the 'optimizer' entry is a torch.optim.AdamW state_dict in the reference's parameter order (util/misc.py:312-318)
    assert set(opt) >= {"state", "param_groups"} and len(opt["param_groups"]) == 2 and opt["countr_amd"]["step"] == 2
    assert all(float(st["step"]) in (1.0, 2.0) and st["exp_avg"].shape == st["exp_avg_sq"].shape for st in opt["state"].values())
    # a mistyped --resume must not silently finetune a randomly initialised (frozen) encoder
    r = subprocess.run([sys.executable, "FSC_finetune_cross.py", "--data_path", fake_fsc, "--batch_size", "2", "--epochs", "1",
                        "--output_dir", out, "--resume", "/nonexistent/checkpoint-300.pth"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "does not exist" in r.stderr
    # --do_resume restores epoch and the flat AdamW state and continues with epoch 1
    log = run(["FSC_finetune_cross.py", "--data_path", "/nonexistent", "--synthetic_steps", "2", "--batch_size", "2", "--epochs", "2",
               "--warmup_epochs", "0", "--output_dir", out, "--resume", ckpt, "--do_resume", "--log_every", "1", "--accum_iter", "2"])
    assert "With optim & sched!" in log
    assert [json.loads(l)["epoch"] for l in log.splitlines() if l.startswith("{")] == [1, 1]
    log = run(["FSC_test_cross.py", "--data_path", fake_fsc, "--resume", ckpt, "--split", "test", "--box_bound", "3"])
    res = json.loads([l for l in log.splitlines() if l.startswith("{")][-1])
    assert res["images"] == 2 and np.isfinite(res["MAE"]) and np.isfinite(res["RMSE"])
    # zero-shot path of the same CLI
    log = run(["FSC_test_cross.py", "--data_path", fake_fsc, "--resume", ckpt, "--split", "val", "--box_bound", "0"])
    assert json.loads([l for l in log.splitlines() if l.startswith("{")][-1])["images"] == 2
    # --external (FSC_test_cross(few-shot).py:96-129): the split's own exemplar crops, cut to --box_bound, serve every image; the
    # counts differ from the per-image exemplars', 5 external exemplars run (shot_num = 5), and the UNBOUNDED list of the train split
    # (--box_bound -1, the reference's default: every box of every image, more than the 8 keys rounds 1-4 stopped at) runs too
    few = [l for l in run(["FSC_test_cross.py", "--data_path", fake_fsc, "--resume", ckpt, "--split", "test"]).splitlines() if "pred_cnt" in l]
    for bound in ("3", "5"):
        log = run(["FSC_test_cross.py", "--data_path", fake_fsc, "--resume", ckpt, "--split", "test", "--external", "--box_bound", bound])
        ext = [l for l in log.splitlines() if "pred_cnt" in l]
        assert len(ext) == len(few) == 2 and ext != few
        assert all(np.isfinite(float(l.split("pred_cnt:")[1].split(",")[0])) for l in ext)
    log = run(["FSC_test_cross.py", "--data_path", fake_fsc, "--resume", ckpt, "--split", "train", "--external"])
    many = [l for l in log.splitlines() if "pred_cnt" in l]
    assert many and all(np.isfinite(float(l.split("pred_cnt:")[1].split(",")[0])) for l in many)


def test_finetune_cli_default_batch_size(tmp_path):
    """No --batch_size: the reference's default of 26 images per GPU (FSC_finetune_cross.py:29), bf16, two synthetic iterations."""
    out = str(tmp_path / "ft26")
    log = run(["FSC_finetune_cross.py", "--data_path", "/nonexistent", "--synthetic_steps", "2", "--epochs", "1", "--warmup_epochs", "0",
               "--output_dir", out, "--resume", "", "--log_every", "1"])
    assert "effective batch size: 26" in log
    lines = [json.loads(l) for l in log.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all(np.isfinite(l["loss"]) and l["loss"] > 0 for l in lines)


def test_finetune_cli_with_augmentation(tmp_path):
    """The reference's default: --do_aug on (noise, colour jitter, blur, affine, flip, mosaic through countr_amd/data/fsc147.py) with
    the class file the cross-image mosaic needs; two DataLoader workers, as a user would run it."""
    from oracle import weights as W
    root = str(tmp_path / "data")
    anno_f, split_f, class_f, im_dir, ids = W.write_aug_dataset(root)
    out = str(tmp_path / "ft")
    log = run(["FSC_finetune_cross.py", "--data_path", root, "--anno_file", "anno.json", "--data_split_file", "split.json",
               "--im_dir", "images", "--class_file", "classes.txt", "--batch_size", "2", "--epochs", "2", "--warmup_epochs", "0",
               "--num_workers", "2", "--output_dir", out, "--resume", "", "--log_every", "1", "--blr", "1e-3"])
    lines = [json.loads(l) for l in log.splitlines() if l.startswith("{")]
    assert len(lines) == 6 and all(np.isfinite(l["loss"]) for l in lines)          # 6 images / batch 2, 2 epochs
    assert os.path.exists(os.path.join(out, "checkpoint__finetuning_last.pth"))
    # without the class file the augmented loader refuses to start (no silent fallback to the plain transform)
    r = subprocess.run([sys.executable, "FSC_finetune_cross.py", "--data_path", root, "--anno_file", "anno.json", "--data_split_file",
                        "split.json", "--im_dir", "images", "--class_file", "nope.txt", "--batch_size", "2", "--epochs", "1",
                        "--output_dir", out, "--resume", ""], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "class_file" in r.stderr


def test_finetune_cli_two_ranks_on_one_gpu(tmp_path):
    """FSC_finetune_cross.py under torch.distributed.run with two ranks sharing the GPU (COUNTR_DIST_BACKEND=gloo): env-driven init,
    DistributedSampler shards, the m_flag MAX all-reduce of the mosaic rule, gradient all-reduce inside the step, loss / validation
    metric reductions, rank-0-only checkpoints (util/misc.py:225-257, FSC_finetune_cross.py:178-183,230,276-284)."""
    import socket
    from oracle import weights as W
    root = str(tmp_path / "data")
    W.write_aug_dataset(root)
    out = str(tmp_path / "ft")
    def cmd():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                str(port), "FSC_finetune_cross.py", "--data_path", root, "--anno_file", "anno.json", "--data_split_file", "split.json", "--im_dir",
                "images", "--class_file", "classes.txt", "--batch_size", "1", "--epochs", "2", "--warmup_epochs", "0", "--num_workers", "0",
                "--output_dir", out, "--resume", "", "--log_every", "1", "--blr", "1e-3"]
    for _attempt in range(2):        # one more try on another port if the process group fails to come up
        r = subprocess.run(cmd(), cwd=ROOT, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", COUNTR_DIST_BACKEND="gloo"))
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 6 and all(np.isfinite(l["loss"]) for l in lines)      # 6 images / (2 ranks x batch 1) = 3 steps x 2 epochs, rank 0 logs
    # as in the reference, every rank evaluates ITS shard of the validation split and prints its own line (no reduction;
    # FSC_finetune_cross.py:329-350,421-422), and rank 0's figure selects the minMAE checkpoint
    assert len([l for l in r.stdout.splitlines() if l.startswith("[Val Epoch #")]) == 4
    assert os.path.exists(os.path.join(out, "checkpoint__finetuning_last.pth"))
    assert os.path.exists(os.path.join(out, "checkpoint__finetuning_minMAE.pth"))


def test_demo_zero_cli(tmp_path):
    """demo_zero.py (reference flags): a directory of frames, counts printed per image, viz_<name>.jpg written at the input size;
    the windows of all frames share forwards and give the same counts as one image per forward."""
    import re
    from PIL import Image
    src = tmp_path / "frames"
    src.mkdir()
    rs = np.random.RandomState(3)
    sizes = {"a.jpg": (640, 360), "b.png": (960, 540), "c.png": (500, 384), "d.png": (300, 600)}     # d: narrower than a window
    for name, (w, h) in sizes.items():
        Image.fromarray(rs.randint(0, 255, size=(h, w, 3)).astype(np.uint8)).save(src / name)
    counts = {}
    for grp in ("8", "1"):
        out = tmp_path / ("out" + grp)
        log = run(["demo_zero.py", "--input_path", str(src), "--output_path", str(out), "--model_path", "", "--precision", "fp32",
                   "--group_images", grp])
        got = dict(re.findall(r"\] (\S+):\tcount =\s*([-0-9.]+)", log))
        assert set(got) == set(sizes), log
        counts[grp] = {k: float(v) for k, v in got.items()}
        for name, (w, h) in sizes.items():
            viz = out / ("viz_%s.jpg" % name.split(".")[0])
            assert viz.exists() and Image.open(viz).size == (w, h)
    assert counts["8"]["d.png"] == 0.0                       # 192 px wide after the resize: the window loop never runs
    for k in sizes:
        assert abs(counts["8"][k] - counts["1"][k]) <= 0.01 + 1e-3 * abs(counts["1"][k]), (k, counts)
    # single file + a missing checkpoint fails like the reference's torch.load
    log = run(["demo_zero.py", "--input_path", str(src / "a.jpg"), "--output_path", str(tmp_path / "o"), "--model_path", "", "--no_viz"])
    assert log.startswith("Count:") or "Count:" in log
    r = subprocess.run([sys.executable, "demo_zero.py", "--input_path", str(src / "a.jpg"), "--model_path", "/nonexistent.pth"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def test_pretrain_cli_on_files_and_synthetic_fallback(fake_fsc, tmp_path):
    out = str(tmp_path / "pre")
    log = run(["FSC_pretrain.py", "--data_path", fake_fsc, "--batch_size", "2", "--epochs", "1", "--warmup_epochs", "0",
               "--num_workers", "0", "--output_dir", out, "--resume", "", "--log_every", "1"])
    lines = [json.loads(l) for l in log.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all(np.isfinite(l["loss"]) and l["loss"] > 0 for l in lines)
    assert os.path.exists(os.path.join(out, "checkpoint__pretraining_0.pth")) and os.path.exists(os.path.join(out, "log.txt"))
    # resume from our own checkpoint (flat AdamW state restored), synthetic images
    log = run(["FSC_pretrain.py", "--data_path", "/nonexistent", "--batch_size", "2", "--epochs", "2", "--warmup_epochs", "0",
               "--synthetic_steps", "2", "--output_dir", out, "--resume", os.path.join(out, "checkpoint__pretraining_0.pth"),
               "--log_every", "1", "--accum_iter", "2"])
    assert "With optim & sched!" in log
    lines = [json.loads(l) for l in log.splitlines() if l.startswith("{")]
    assert [l["epoch"] for l in lines] == [1, 1]


def test_bench_contract_line():
    """bench.py prints exactly one JSON line with the driver's contract keys, the roofline and (at N = 1) the CPU baseline."""
    out = run(["bench.py", "--steps", "3", "--warmup", "2"])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 2 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["dtype"] == "bf16" and j["data"] == "synthetic" and j["unit"] == "images/sec"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 8 * 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["peak"] == 2500.0 and 0.02 < r["frac"] < 1.0 and (r["traffic"] is None or r["traffic"] > 0)
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "images/sec" and c["sample"]
    # the pretraining workload is selected explicitly and is labelled as such
    out = run(["bench.py", "--workload", "pretrain", "--steps", "2", "--warmup", "2"])
    p = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert "pretrain" in p["metric"] and p["value"] > 0 and p["n_gpus"] == 1
